#!/usr/bin/env python3
"""profiles/round<N>_summary.md and the committed copies of a tools/profile_round.sh run (run HERE, in the build container, after gpurun merged the files back:
the traffic JSON is stamped with the commit and the conv kernel source hash it was taken at, which bench.py compares with the tree it runs from).
usage: make_summary_round.py <round number> <gpurun_out tag>   (expects gpurun_out/<tag>_stats/p_kernel_stats.csv, <tag>_pmc_traffic.json, <tag>_bench.json)"""
import csv, hashlib, json, os, re, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RN = int(sys.argv[1])
tag = sys.argv[2]
P = f'round{RN}'
G = os.path.join(R, 'gpurun_out')
STEPS = 14                      # profile_round.sh: --steps 10 --warmup 4 (2 eager + capture, then replays; every step's kernels are traced)
shutil.copy(f'{G}/{tag}_stats/p_kernel_stats.csv', f'{R}/profiles/{P}_bench_kernel_stats.csv')
pm = json.load(open(f'{G}/{tag}_pmc_traffic.json'))
pm['_stamp'] = {'commit': subprocess.run(['git', '-C', R, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip(),
                'conv_igemm_sha16': hashlib.sha256(open(f'{R}/dualsuperreslearningforsemseg_amd/csrc/conv_igemm.hip', 'rb').read()).hexdigest()[:16],
                'conv_planes_sha16': hashlib.sha256(open(f'{R}/dualsuperreslearningforsemseg_amd/csrc/conv_planes.hip', 'rb').read()).hexdigest()[:16],
                'conv_wgrad3_sha16': hashlib.sha256(open(f'{R}/dualsuperreslearningforsemseg_amd/csrc/conv_wgrad3.hip', 'rb').read()).hexdigest()[:16],
                'conv_split_kernel_sha16': hashlib.sha256(open(f'{R}/dualsuperreslearningforsemseg_amd/csrc/conv_split_kernel.h', 'rb').read() +
                                                          open(f'{R}/dualsuperreslearningforsemseg_amd/csrc/conv_sk.hip', 'rb').read()).hexdigest()[:16]}
json.dump(pm, open(f'{R}/profiles/{P}_pmc_traffic.json', 'w'), indent=1)
line = open(f'{G}/{tag}_bench.json').read().strip().splitlines()[-1]
open(f'{R}/profiles/{P}_bench_line.json', 'w').write(line + '\n')
d = json.loads(line)
rows = list(csv.DictReader(open(f'{R}/profiles/{P}_bench_kernel_stats.csv')))


def grp(pat, exclude=None):
    rs = [r for r in rows if re.search(pat, r['Name']) and not (exclude and re.search(exclude, r['Name']))]
    c = sum(int(r['Calls']) for r in rs); t = sum(float(r['TotalDurationNs']) for r in rs)
    return c / STEPS, t / 1e6 / STEPS, (t / c / 1e3 if c else 0.0)


tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / STEPS
groups = [
    ('conv forward: conv_igemm_split_kernel<.., false, 2, .., 2> (f16x3, pre-split filters)', r'conv_igemm_split_kernel<.*?, false, \d', None),
    ('conv dgrad: conv_igemm_split_kernel<.., true, 2, .., 2> (f16x3, pre-split filters)', r'conv_igemm_split_kernel<.*?, true, \d', None),
    ('conv forward with fp16-plane operands staged by LDS-DMA: conv_planes_kernel (the 3x3 convs over operands of >= 8 Mi elements: cat_conv.0 / cat_conv.4 / SISR, the three dilated ASPP convs)', r'conv_planes_kernel', None),
    ('plane producers: split_planes_kernel (concat buffer, cat_conv.4 input, layer4 output), filter_planes_batched_kernel (six filters)', r'split_planes_kernel|filter_planes_batched', None),
    ('conv wgrad, grouped, 3x3 stride 1 with all nine taps per block: conv_wgrad3_group_kernel (dilation 1 / 2)', r'conv_wgrad3', None),
    ('conv wgrad, grouped, one tap per block: conv_wgrad_group_kernel (1x1, strided, dilated ASPP)', r'conv_wgrad_group_kernel', None),
    ('conv wgrad, stem (row-folded 7x7) + grouped slab reduce', r'conv_wgrad_split_kernel|wgrad_reduce', None),
    ('split-K reduces of forward / dgrad', r'splitk_reduce', None),
    ('operand magnitudes measured by a launch of their own (amax_kernel; the rest is left by producers)', r'amax_kernel', None),
    ('BatchNorm forward from conv-epilogue statistics: bn_stats_apply_kernel (+ bn_stats_reduce_kernel for > 256 row blocks)', r'bn_stats_apply_kernel|bn_stats_reduce_kernel', r'bwd'),
    ('BatchNorm backward from dgrad-epilogue sums: bn_bwd_stats_apply_kernel (+ bn_bwd_stats_reduce_kernel)', r'bn_bwd_stats_apply|bn_bwd_stats_reduce', None),
    ('BatchNorm single-kernel (device-wide barrier): bn_fused_fwd / bn_fused_bwd', r'bn_fused', None),
    ('BatchNorm three-kernel path (large / odd-width tensors) + eval apply', r'bn_(partial|finalize|apply|bwd_partial|bwd_finalize|bwd_apply)', r'stats_apply'),
    ('fused loss pass: count_valid, ce_fused, mse_fused, fa_sim / fa_pairs / fa_bwd, finalizers, loss_mix', r'ce_fused|mse_fused|count_valid|fa_sim|fa_pairs|fa_bwd|fa_finalize|ce_finalize|mse_finalize|loss_mix', None),
    ('ConvTranspose 19->19 forward / dx / dw', r'convt2x2', None),
    ('bilinear, pixel shuffle, pools, pointwise stride-8, dropout, colsum, concat copies', r'bilinear|pixel_shuffle|maxpool|gap_|pointwise|dropout_kernel|colsum|copyBufferRect|pad_image|nchw|copy2d', None),
    ('NOT per step: parameter upload / arena set-up copies of the process (__amd_rocclr_copyBuffer, ~1300 launches once), shown divided by the step count', r'__amd_rocclr_copyBuffer$', None),
    ('SGD + filter pass (amax + pre-split: weight_transpose_batched / weight_split_batched) + dropout-key advance + NaN check', r'sgd|weight_transpose|weight_split|weight_amax|rng_advance|nan_check', None),
    ('remaining ATen elementwise / fill kernels', r'at::native|fillBuffer|zero_fill_kernel', None),
]
g = {name: grp(pat, ex) for name, pat, ex in groups}
SETUP = [name for name, _, _ in groups if name.startswith('NOT per step')][0]
known = sum(v[1] for v in g.values())
r = d['roofline']
ba = d.get('images_per_s_by_conv_arithmetic') or {}
txt = f'''# Round {RN} profile summary (1x MI355X, stage 3, B=8, 256x512 -> 512x1024, fp32 tensors, default 'f16x3' = fp32-equivalent conv arithmetic)

Produced by `tools/profile_round.sh` + `tools/make_summary_round.py` on the GPU box: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10
--warmup 4 --no-prof --no-cpu-baseline --no-config5` (the default step: a hipGraph replay that runs one kernel at a time, so the per-kernel durations are
exclusive; {STEPS} steps in the trace: 2 eager, the rest replays), then two `--pmc` passes (FETCH_SIZE, WRITE_SIZE) aggregated by `tools/pmc_traffic.py`.
Files: per-kernel table `{P}_bench_kernel_stats.csv`; bench line `{P}_bench_line.json`; HBM-side traffic per conv kernel family
`{P}_pmc_traffic.json`.

Default bench run of the same build: **{d['value']:.1f} images/s, {d['ms_per_step']:.2f} ms per step** (host enqueue {d['host_enqueue_ms_per_step']:.2f} ms per step: one
hipGraphLaunch); the same step timed over the same {d['steps']} steps / {d['warmup']} warm-ups per arithmetic: {ba}.
CPU baselines on the box's host cores: stock torch.nn CPU graph {d.get('cpu_baseline', {}).get('value')} images/s on {d.get('cpu_baseline', {}).get('cores')} threads; numpy oracle
{d.get('cpu_baseline_numpy_port', {}).get('value')} images/s.

Total kernel time per step: {tot:.2f} ms, of which {g[SETUP][1]:.2f} ms is the one-time set-up row (per step without it: {tot - g[SETUP][1]:.2f} ms)

| group | ms/step | launches/step | avg us |
|---|---|---|---|
''' + '\n'.join(f'| {name} | {g[name][1]:.2f} | {g[name][0]:.0f} | {g[name][2]:.1f} |' for name, _, _ in groups) + f'''
| not matched above | {tot - known:.2f} | | |

bench.py HIP events (5 eager steps behind the timed region, every conv launch bracketed on its stream), `{P}_bench_line.json`:
'''
for k, v in r['all_mfma_kernels'].items():
    txt += f"* {k}: {v['ms_per_step']} ms/step, {v['tflops']} algorithmic TFLOP/s = {v['frac']:.3f} of {v['peak']} TF; algorithmic bytes/launch {v['algorithmic_bytes_per_launch'] / 1e6:.1f} MB\n"
ds = r.get('decoder_stack')
if ds:
    txt += ('\nDecoder-head conv stack (north_star: >= 30 % of MFMA peak), each conv launched alone: '
            + '; '.join(f"{k} {ds[k]['achieved']} TF = {ds[k]['frac']:.3f} of {ds[k]['peak']}" for k in ('forward', 'dgrad', 'wgrad'))
            + f"; all passes {ds['all_passes']['achieved']} TF ({ds['all_passes']['frac_of_time_weighted_peak']:.3f}).\n")
txt += '\nHBM-side traffic per launch (PMC, FETCH_SIZE x2 + WRITE_SIZE):\n'
for k, v in pm.items():
    if k.startswith('_'):
        continue
    txt += f"* {k}: {v['hbm_bytes_per_launch'] / 1e6:.1f} MB (fetch {2 * v['FETCH_SIZE_KB_per_launch_raw'] / 1e3:.1f} MB, write {v['WRITE_SIZE_KB_per_launch'] / 1e3:.1f} MB), {v['launches_sampled']} launches sampled\n"
if d.get('roofline_hbm'):
    txt += '\nMemory-bound kernels, each launched alone (algorithmic bytes / time vs 8 TB/s):\n'
    for row in d['roofline_hbm']:
        txt += f"* {row['kernel']}: {row['achieved']} GB/s = {row['frac']:.3f} ({row['avg_launch_ms'] * 1e3:.0f} us)\n"
open(f'{R}/profiles/{P}_summary.md', 'w').write(txt)
print(txt)
