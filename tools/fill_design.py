#!/usr/bin/env python3
"""Fills the measured figures of DESIGN.md section 6 from profiles/round<N>_bench_line.json, round<N>_summary.md and round<N>_pmc_traffic.json (N = argv[1], default 4), so that the text
and the committed profiles cannot drift apart.  Fields are <!--NAME-->value<!--/NAME--> (a first run converts @@NAME@@ placeholders)."""
import json, os, re, sys
RN = int(sys.argv[1]) if len(sys.argv) > 1 else 4
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.loads(open(f'{R}/profiles/round{RN}_bench_line.json').read().strip().splitlines()[-1])
pm = json.load(open(f'{R}/profiles/round{RN}_pmc_traffic.json'))
summ = open(f'{R}/profiles/round{RN}_summary.md').read()
rf = d['roofline']; ds = rf['decoder_stack']
F = {}
F['VALUE'] = f"{d['value']:.1f}"
F['MS'] = f"{d['ms_per_step']:.2f}"
m = d['images_per_s_by_conv_arithmetic']
F['MODES'] = ', '.join(f"`{k}` {v}" for k, v in m.items() if k != 'f16x3') + ' images/s'
F['F16X1'] = str(m.get('f16x1', 'n/a'))
c5 = d['config5']
F['C5'] = f"{c5['value']:.1f}"
c5m = c5.get('images_per_s_by_conv_arithmetic', {})
F['C5X1'] = str(c5m.get('f16x1', 'n/a'))
F['F16X1LINE'] = (f"{m.get('f16x1', 'n/a')} against {d['value']:.1f} images/s at 256×512 ({100 * (m.get('f16x1', d['value']) / d['value'] - 1):+.0f} %), "
                  f"{c5m.get('f16x1', 'n/a')} against {c5['value']:.1f} at config 5's size ({100 * (c5m.get('f16x1', c5['value']) / c5['value'] - 1):+.0f} %)")
F['CPUC1'] = f"{d.get('cpu_baseline_c1', {}).get('value', float('nan')):.2f}"
sqf = f'{R}/profiles/round{RN}_sq_counters_step.json'
fams = json.load(open(sqf)).get('_families', {}) if os.path.isfile(sqf) else {}
F['MFMABUSY'] = ('; '.join(f"`{k}` {v['mfma_busy']:.3f}" for k, v in fams.items()) + f" (dominant: `{rf['kernel']}`)") if fams else 'not collected'
F['CPU'] = f"{d['cpu_baseline']['value']:.2f}"
F['CPUNP'] = f"{d['cpu_baseline_numpy_port']['value']:.2f}"
rows = re.findall(r'^\| (.+?) \| ([\d.]+) \| (\d+) \| ([\d.]+) \|$', summ, re.M)
short = [('conv forward:', 'conv forward (register-staged)'), ('conv dgrad:', 'conv dgrad (register-staged)'), ('conv forward with fp16-plane', 'conv forward with plane operands (LDS-DMA)'), ('plane producers', 'plane producers'), ('conv wgrad, grouped, 3x3', 'grouped weight gradients, 3x3 with all taps per block'), ('conv wgrad, grouped, one tap', 'grouped weight gradients, one tap per block'), ('conv wgrad, stem', 'stem weight gradient + slab reduce'),
         ('split-K reduces', 'split-K reduces'), ('operand magnitudes', 'amax launches'), ('BatchNorm forward from', 'BatchNorm forward from conv statistics'),
         ('BatchNorm backward from', 'BatchNorm backward from dgrad sums'), ('BatchNorm single-kernel', 'BatchNorm barrier kernels'), ('BatchNorm three-kernel', 'BatchNorm large / odd tensors'),
         ('fused loss pass', 'loss pass'), ('ConvTranspose', 'ConvTranspose tail'), ('bilinear', 'bilinear / shuffle / pools / dropout / concat'), ('SGD + filter pass', 'SGD + filter pass'),
         ('remaining ATen', 'remaining ATen elementwise')]
parts = []
for key, name in short:
    for r in rows:
        if r[0].startswith(key):
            parts.append(f'{name} {float(r[1]):.2f} ({int(r[2])} launches)')
tot = re.search(r'per step without it: ([\d.]+) ms', summ)
F['WHERE'] = ', '.join(parts) + (f'; sum {tot.group(1)} ms (the one-time parameter-upload copies of the process are listed apart in the summary).' if tot else '.')
order = ['conv_igemm_split_kernel<f16x3> (dgrad)', 'conv_wgrad_split_kernel<f16x3>', 'conv_igemm_split_kernel<f16x3> (forward)']
lines = []
for k in order:
    v = rf['all_mfma_kernels'][k]
    n = {'conv_igemm_split_kernel<f16x3> (dgrad)': '114', 'conv_wgrad_split_kernel<f16x3>': '2 grouped grids + stem', 'conv_igemm_split_kernel<f16x3> (forward)': '115'}[k]
    dom = ' — dominant' if k == rf['kernel'] else ''
    lines.append(f"| `{k}` ({n}){dom} | {v['ms_per_step']:.2f} | {v['tflops']:.1f} | **{v['frac']:.3f}** | {v['tflops'] / 419.4:.3f} |")
F['ROOF'] = '\n'.join(lines)
L = ds['layers']
c4 = L['cat_conv.4 3x3 256->256']; c0 = L['cat_conv.0 3x3 304->256']
F['DEC'] = (f"forward {ds['forward']['achieved']:.0f} TF = {ds['forward']['frac']:.3f}, dgrad {ds['dgrad']['achieved']:.0f} TF = {ds['dgrad']['frac']:.3f}, weight gradients "
            f"{ds['wgrad']['achieved']:.0f} TF = {ds['wgrad']['frac']:.3f}, all passes {ds['all_passes']['achieved']:.0f} TF = {ds['all_passes']['frac_of_time_weighted_peak']:.3f} of the f16x3 roof "
            f"(cat_conv.4: forward {c4['forward_tflops']:.0f} TF = {c4['forward_tflops'] / 838.9:.2f}, dgrad {c4['dgrad_tflops']:.0f}; cat_conv.0: forward {c0['forward_tflops']:.0f}, dgrad {c0['dgrad_tflops']:.0f}); "
            f"the same times against round 2's bf16x6 roof would read {ds['all_passes']['achieved'] / 419.4:.2f}")
def tr(k):
    v = pm[k]; return f"{v['hbm_bytes_per_launch'] / 1e6:.1f} MB"
F['TRAFFIC'] = (f"dgrad {tr(order[0])} per launch for {rf['all_mfma_kernels'][order[0]]['algorithmic_bytes_per_launch'] / 1e6:.1f} MB algorithmic, forward {tr(order[2])} for "
                f"{rf['all_mfma_kernels'][order[2]]['algorithmic_bytes_per_launch'] / 1e6:.1f} MB, grouped weight gradients {pm[order[1]]['hbm_bytes_per_launch'] / 1e9:.2f} GB for "
                f"{rf['all_mfma_kernels'][order[1]]['algorithmic_bytes_per_launch'] / 1e9:.2f} GB.")
F['HBM'] = '; '.join(f"{h['kernel'].split(' (')[0]} {h['frac']:.2f}" for h in d['roofline_hbm']) + '.'
p = f'{R}/DESIGN.md'
s = open(p).read()
for k, v in F.items():
    s = s.replace(f'@@{k}@@', f'<!--{k}-->{v}<!--/{k}-->')
    s = re.sub(rf'<!--{k}-->.*?<!--/{k}-->', lambda m_: f'<!--{k}-->{v}<!--/{k}-->', s, flags=re.S)
open(p, 'w').write(s)
print({k: (v if len(v) < 80 else v[:77] + '...') for k, v in F.items()})
