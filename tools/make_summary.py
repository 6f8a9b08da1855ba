#!/usr/bin/env python3
"""profiles/round1_summary.md + the measurement tables of DESIGN.md from a tools/profile_round.sh run.
usage: make_summary.py <gpurun_out tag>   (expects gpurun_out/<tag>_stats/, <tag>_pmc_traffic.json, <tag>_bench_line.json)"""
import csv, glob, json, os, re, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G = os.path.join(R, 'gpurun_out')
shutil.copy(glob.glob(f'{G}/{tag}_stats/**/*kernel_stats.csv', recursive=True)[0], f'{R}/profiles/round1_bench_kernel_stats_exclusive.csv')
shutil.copy(f'{G}/{tag}_pmc_traffic.json', f'{R}/profiles/round1_pmc_traffic.json')
shutil.copy(f'{G}/{tag}_bench_line.json', f'{R}/profiles/round1_bench_line.json')
rows = list(csv.DictReader(open(f'{R}/profiles/round1_bench_kernel_stats_exclusive.csv')))
steps = 7


def avg(fn):
    rs = [r for r in rows if fn(r['Name'])]
    c = sum(int(r['Calls']) for r in rs); t = sum(float(r['TotalDurationNs']) for r in rs)
    return c / steps, t / 1e6 / steps, (t / c / 1e3 if c else 0)


tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / steps
groups = [('conv_igemm_split_kernel<bf16x6> (forward)', lambda n: 'conv_igemm_split' in n and 'false, 3' in n),
          ('conv_wgrad_split_kernel<bf16x3>', lambda n: 'conv_wgrad_split' in n),
          ('conv_igemm_split_kernel<bf16x3> (dgrad)', lambda n: 'conv_igemm_split' in n and 'true, 2' in n),
          ('bn_stats_apply_kernel (forward, statistics from the conv epilogue: 83 layers)', lambda n: 'bn_stats_apply' in n and 'bwd' not in n),
          ('bn_bwd_stats_apply_kernel (backward, sums from the dgrad epilogue; split-K dgrads fall back)', lambda n: 'bn_bwd_stats_apply' in n),
          ('bn_fused_fwd_kernel / bn_fused_bwd_kernel (device-wide barrier)', lambda n: 'bn_fused' in n),
          ('bn_* three-kernel path (8 large / odd-width layers)', lambda n: 'bn_' in n and 'bn_fused' not in n and 'stats_apply' not in n),
          ('wgrad_reduce_kernel', lambda n: 'wgrad_reduce' in n),
          ('torch elementwise add (remaining gradient accumulation)', lambda n: 'CUDAFunctor_add' in n),
          ('weight_transpose_batched_kernel (1 launch/step)', lambda n: 'weight_transpose' in n),
          ('splitk_reduce_kernel', lambda n: 'splitk' in n)]
g = {name: avg(fn) for name, fn in groups}
lines = [f'| {name} | {g[name][1]:.2f} | {g[name][0]:.0f} | {g[name][2]:.1f} |' for name, _ in groups]
known = sum(v[1] for v in g.values())
lines.append(f'| everything else (ConvT tail, CE/MSE/FA, bilinear, pools, SGD, NaN check, one-time arena copies) | {tot - known:.2f} | | |')
d = json.load(open(f'{R}/profiles/round1_bench_line.json'))
r = d['roofline']
pm = json.load(open(f'{R}/profiles/round1_pmc_traffic.json'))
ba = d.get('images_per_s_by_conv_arithmetic') or {}
txt = f'''# Round 1 profile summary (1x MI355X, stage 3, B=8, 256x512 -> 512x1024, fp32 tensors, 'mixed' conv arithmetic)

Produced by `tools/profile_round.sh` + `tools/make_summary.py` on the GPU box: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5
--warmup 2 --no-prof --no-cpu-baseline` with `DSRL_OVERLAP_WGRAD=0` (kernels run one at a time; 7 steps in the trace, the first one includes
the one-time arena initialisation copies), then two `--pmc` passes (FETCH_SIZE, WRITE_SIZE) aggregated by `tools/pmc_traffic.py`.
Files: per-kernel table `round1_bench_kernel_stats_exclusive.csv`; bench line `round1_bench_line.json`; HBM-side traffic per conv kernel
family `round1_pmc_traffic.json`; arithmetic modes per layer shape (time and error vs fp64) `round1_precision_modes.txt`; SQ counters of
cat_conv.0 forward for the fp32 kernel, the first split kernel and the pipelined split kernel `round1_pmc_cat0_sq_counters.txt`;
tile / split / K-group sweeps `round1_sweep_*.txt`.

Default bench run of the same build: {d['value']:.1f} images/s, {d['ms_per_step']:.1f} ms per step wall (weight-gradient stream overlapping); the same step with
bf16x6 everywhere {ba.get('bf16x6')} images/s, with exact-product fp32 MFMA {ba.get('fp32')} images/s. Boxes of the pool differ by about +-3 %.

Total kernel time per step, exclusive execution: {tot:.2f} ms

| group | ms/step | launches/step | avg us |
|---|---|---|---|
''' + '\n'.join(lines) + '''

bench.py HIP events, exclusive pass (library bracket = kernel + its slab-reduce launch), `round1_bench_line.json`:
'''
for k, v in r['all_mfma_kernels'].items():
    txt += f"* {k}: {v['ms_per_step']} ms/step, {v['tflops']} algorithmic TFLOP/s = {v['frac']:.3f} of {v['peak']} TF; algorithmic bytes/launch {v['algorithmic_bytes_per_launch']/1e6:.1f} MB\n"
txt += '\nHBM traffic per launch (PMC, FETCH_SIZE x2 + WRITE_SIZE): ' + '; '.join(f"{k}: {v['hbm_bytes_per_launch']/1e6:.1f} MB" for k, v in pm.items()) + '\n'
fw, wg, dg, rd = g[groups[0][0]], g[groups[1][0]], g[groups[2][0]], g['wgrad_reduce_kernel']
txt += f'''
rocprofv3 average durations vs the event brackets: forward {fw[2]:.1f} us (bracket {1e3*r['all_mfma_kernels'][groups[0][0]]['ms_per_step']/115:.1f} us incl. split-K reduces),
wgrad {wg[2]:.1f} us + {rd[2]:.1f} us reduce (bracket {1e3*r['all_mfma_kernels'][groups[1][0]]['ms_per_step']/115:.1f} us), dgrad {dg[2]:.1f} us (bracket {1e3*r['all_mfma_kernels'][groups[2][0]]['ms_per_step']/114:.1f} us).

SQ counters, cat_conv.0 forward (1024 tiles of 128x128, bf16x6), per dispatch (`round1_pmc_cat0_sq_counters.txt`):
* fp32 kernel: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs) = 86 % of all SIMD cycles; LDS conflicts 0.
* first split kernel (single LDS stage, 80-byte rows): MFMA busy 54 %, VALU 37 %, MFMA/VALU co-execution 10 % of MFMA cycles, SQ_LDS_BANK_CONFLICT 33 % of LDS cycles.
* pipelined split kernel (two 16-deep stages, swizzled 32-byte rows, interleaved convert steps): MFMA busy 61 %, co-execution 47 %, bank conflicts 0,
  GRBM cycles per dispatch 1.02 M -> 0.90 M.

Timing-only experiments (wrong results on purpose, not in the tree any more): 5/6 of the MFMAs removed and all split conversions removed
leave 42 of 54 us on layer3's 3x3 forward and 368 of 579 us on cat_conv.0 - operand fetch (L2 -> LDS) and per-half-step latency bound,
not MFMA / VALU bound.
'''
open(f'{R}/profiles/round1_summary.md', 'w').write(txt)

# ---- headline sentence and the roofline table
p = f'{R}/DESIGN.md'
s = open(p).read()
s = re.sub(r'`python bench.py` \(profiles/round1_bench_line.json\): \*\*[0-9.]+ images/s, [0-9.]+ ms per step\*\*',
           f"`python bench.py` (profiles/round1_bench_line.json): **{d['value']:.1f} images/s, {d['ms_per_step']:.1f} ms per step**", s)
i = s.index('| kernel family (launches/step) | ms/step |')
j = s.index('\n\n', i)
tab = '| kernel family (launches/step) | ms/step | algorithmic TFLOP/s | peak | frac | HBM bytes/launch (PMC) vs algorithmic |\n|---|---|---|---|---|---|\n'
for k in (groups[1][0], groups[0][0], groups[2][0]):
    v = r['all_mfma_kernels'][k]
    tab += (f"| `{k}` ({114 if 'dgrad' in k else 115}){' — dominant' if k == r['kernel'] else ''} | {v['ms_per_step']:.2f}{' (incl. slab reduce)' if 'wgrad' in k else ''} | "
            f"{v['tflops']:.1f} | {v['peak']} | **{v['frac']:.3f}** | {pm[k]['hbm_bytes_per_launch']/1e6:.1f} MB vs {v['algorithmic_bytes_per_launch']/1e6:.1f} MB |\n")
s = s[:i] + tab.rstrip('\n') + s[j:]
open(p, 'w').write(s)
print(open(f'{R}/profiles/round1_summary.md').read()[1200:2600])
