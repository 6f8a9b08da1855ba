#!/usr/bin/env python3
"""Per-group table of a rocprofv3 kernel-stats CSV of a bench.py run (any conv arithmetic).  usage: kstats.py <p_kernel_stats.csv> <steps in the trace>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
STEPS = float(sys.argv[2]) if len(sys.argv) > 2 else 14.0
GROUPS = [
    ('conv forward (conv_igemm_split_kernel, DGRAD = false)', r'conv_igemm_split_kernel<.*?, false, \d', None),
    ('conv dgrad (conv_igemm_split_kernel, DGRAD = true)', r'conv_igemm_split_kernel<.*?, true, \d', None),
    ('conv forward, plane operands (conv_planes_kernel, DGRAD = false)', r'conv_planes_kernel<.*?, false, \d', None),
    ('conv dgrad, plane operands (conv_planes_kernel, DGRAD = true)', r'conv_planes_kernel<.*?, true, \d', None),
    ('plane producers (split_planes_kernel, filter_planes_batched_kernel)', r'split_planes_kernel|filter_planes_batched', None),
    ('conv fp32 MFMA kernels', r'conv_igemm_f32_kernel|conv_wgrad_f32_kernel', None),
    ('conv wgrad grouped, one tap per block (conv_wgrad_group_kernel)', r'conv_wgrad_group_kernel', None),
    ('conv wgrad grouped, 3x3 with all taps per block (conv_wgrad3_group_kernel)', r'conv_wgrad3', None),
    ('conv wgrad per layer (stem) + slab reduces', r'conv_wgrad_split_kernel|wgrad_reduce', None),
    ('split-K reduces of forward / dgrad', r'splitk_reduce', None),
    ('operand magnitudes (amax_kernel)', r'amax_kernel', None),
    ('BN forward from conv statistics (bn_stats_apply_kernel)', r'bn_stats_apply_kernel', r'bwd'),
    ('BN backward from dgrad sums (bn_bwd_stats_apply_kernel)', r'bn_bwd_stats_apply', None),
    ('BN single-kernel with device-wide barrier (bn_fused_*)', r'bn_fused', None),
    ('BN three-kernel path', r'bn_(partial|finalize|apply|bwd_partial|bwd_finalize|bwd_apply)', r'stats_apply'),
    ('loss pass', r'ce_fused|mse_fused|count_valid|fa_fwd|fa_bwd|ce_finalize|mse_finalize|loss_mix', None),
    ('ConvTranspose', r'convt2x2', None),
    ('bilinear / shuffle / pools / pointwise / dropout / colsum / copies', r'bilinear|pixel_shuffle|maxpool|gap_|pointwise|dropout_kernel|colsum|copyBufferRect|pad_image|nchw|copy2d', None),
    ('one-time set-up copies (__amd_rocclr_copyBuffer)', r'__amd_rocclr_copyBuffer$', None),
    ('SGD + filter transposes + key advance + NaN check', r'sgd|weight_transpose|weight_split|weight_amax|rng_advance|nan_check', None),
    ('ATen elementwise / fills', r'at::native|fillBuffer', None),
]
tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / STEPS
known = 0.0
print(f'total kernel time per step {tot:.2f} ms ({STEPS:.0f} steps)')
for name, pat, ex in GROUPS:
    rs = [r for r in rows if re.search(pat, r['Name']) and not (ex and re.search(ex, r['Name']))]
    c = sum(int(r['Calls']) for r in rs); t = sum(float(r['TotalDurationNs']) for r in rs)
    known += t / 1e6 / STEPS
    print(f'{t / 1e6 / STEPS:7.3f} ms  {c / STEPS:6.1f} launches  {t / c / 1e3 if c else 0:8.1f} us  {name}')
print(f'{tot - known:7.3f} ms  not matched')
if len(sys.argv) > 3:
    for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:int(sys.argv[3])]:
        print(f"{float(r['TotalDurationNs']) / 1e6 / STEPS:7.3f} ms {int(r['Calls']) / STEPS:6.1f} x {float(r['TotalDurationNs']) / int(r['Calls']) / 1e3:8.1f} us  {r['Name'][:150]}")
