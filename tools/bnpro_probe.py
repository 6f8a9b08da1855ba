#!/usr/bin/env python3
"""PROBE (needs a library built with the probe patch, see profiles/round5_bn_in_operand_path_probe.txt): what does it cost a forward conv to apply a
per-channel scale / shift + ReLU (a BatchNorm's apply) to its activation operand while staging it?  Forward launches of the step's backbone shapes,
back-to-back, HIP events; DSRL_PROBE_BNPRO=1 turns the extra work on inside the same kernels."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualsuperreslearningforsemseg_amd import functional as HF
from dualsuperreslearningforsemseg_amd._lib import call
B = 8
SHAPES = [('l1 1x1 64->64', 64, 64, 128, 64, 1, 0, 1, 3), ('l1 3x3 64', 64, 64, 128, 64, 3, 1, 1, 3), ('l1 1x1 64->256', 64, 64, 128, 256, 1, 0, 1, 3),
          ('l2 3x3 128', 128, 32, 64, 128, 3, 1, 1, 4), ('l2 1x1 128->512', 128, 32, 64, 512, 1, 0, 1, 4), ('l2 1x1 512->128', 512, 32, 64, 128, 1, 0, 1, 3),
          ('l3 3x3 256', 256, 16, 32, 256, 3, 1, 1, 23), ('l3 1x1 256->1024', 256, 16, 32, 1024, 1, 0, 1, 23), ('l3 1x1 1024->256', 1024, 16, 32, 256, 1, 0, 1, 22),
          ('l4 3x3 d2 512', 512, 16, 32, 512, 3, 2, 2, 3), ('l4 1x1 512->2048', 512, 16, 32, 2048, 1, 0, 1, 3), ('l4 1x1 2048->512', 2048, 16, 32, 512, 1, 0, 1, 2)]


def timeit(fn, reps=30):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


dev = 'cuda:0'
tot = [0.0, 0.0]
for name, C, H, W, K, R, pad, dil, cnt in SHAPES:
    x = torch.randn((B, C, H, W), device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((K, C, R, R), device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    y = HF.new_cl((B, K, H, W), x)
    rec, wsp, wtsp, wt = HF.split_filter(w)
    xa = HF.amax_for(x)
    shp = (B, H, W, C, K, R, R, 1, pad, dil)
    ws = torch.empty(256 << 20, device=dev, dtype=torch.uint8)
    st = HF._stream()
    f = lambda: call('dsrl_conv2d_fwd_planes', x.data_ptr(), C, xa.data_ptr(), None, w.data_ptr(), rec.data_ptr(), wsp.data_ptr(), None, None, y.data_ptr(), K, *shp,
                     ws.data_ptr(), ws.numel(), None, 0, st)
    t = []
    for v in ('0', '1'):
        os.environ['DSRL_PROBE_BNPRO'] = v
        t.append(timeit(f))
    tot[0] += t[0] * cnt; tot[1] += t[1] * cnt
    print(f'{name:20s} x{cnt:2d}: {t[0]:6.1f} us -> {t[1]:6.1f} us with the BatchNorm apply in the operand path ({t[1] - t[0]:+5.1f})', flush=True)
print(f'per step over the listed layers: {tot[0] / 1e3:.3f} -> {tot[1] / 1e3:.3f} ms ({(tot[1] - tot[0]):+.0f} us for {sum(s[-1] for s in SHAPES)} launches)')
# sanity: the probe path really runs (the result changes: here scale = x[0:C], shift = x[4096:4096+C] of the input tensor itself)
outs = []
for v in ('0', '1'):
    os.environ['DSRL_PROBE_BNPRO'] = v
    f(); torch.cuda.synchronize(); outs.append(y.clone())
print('last layer, output changed by the probe path:', not torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max()))
