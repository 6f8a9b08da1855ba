#!/usr/bin/env python3
"""Per-shape timing of the MFMA conv kernels (fwd / dgrad / wgrad) through the C ABI: in-bounds TFLOP/s per layer shape of
the stage-3 step at B=8, 256x512.  Usage: python tools/conv_bench.py [--reps 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualsuperreslearningforsemseg_amd import functional as HF  # noqa: E402

# (name, N, C, H, W, K, R, stride, pad, dil, count per step)
B = 8
SHAPES = [
    ('stem 7x7s2', B, 4, 256, 512, 64, 7, 2, 3, 1, 1),
    ('l1 1x1 64->64', B, 64, 64, 128, 64, 1, 1, 0, 1, 1), ('l1 3x3 64', B, 64, 64, 128, 64, 3, 1, 1, 1, 3),
    ('l1 1x1 64->256', B, 64, 64, 128, 256, 1, 1, 0, 1, 4), ('l1 1x1 256->64', B, 256, 64, 128, 64, 1, 1, 0, 1, 2),
    ('l2 1x1 256->128', B, 256, 64, 128, 128, 1, 1, 0, 1, 1), ('l2 3x3s2 128', B, 128, 64, 128, 128, 3, 2, 1, 1, 1),
    ('l2 3x3 128', B, 128, 32, 64, 128, 3, 1, 1, 1, 3), ('l2 1x1 128->512', B, 128, 32, 64, 512, 1, 1, 0, 1, 4),
    ('l2 1x1 512->128', B, 512, 32, 64, 128, 1, 1, 0, 1, 3), ('l2 ds 1x1s2 256->512', B, 256, 64, 128, 512, 1, 2, 0, 1, 1),
    ('l3 1x1 512->256', B, 512, 32, 64, 256, 1, 1, 0, 1, 1), ('l3 3x3s2 256', B, 256, 32, 64, 256, 3, 2, 1, 1, 1),
    ('l3 3x3 256', B, 256, 16, 32, 256, 3, 1, 1, 1, 22), ('l3 1x1 256->1024', B, 256, 16, 32, 1024, 1, 1, 0, 1, 23),
    ('l3 1x1 1024->256', B, 1024, 16, 32, 256, 1, 1, 0, 1, 22), ('l3 ds 1x1s2 512->1024', B, 512, 32, 64, 1024, 1, 2, 0, 1, 1),
    ('l4 1x1 1024->512', B, 1024, 16, 32, 512, 1, 1, 0, 1, 1), ('l4 3x3 d1 512', B, 512, 16, 32, 512, 3, 1, 1, 1, 1),
    ('l4 3x3 d2 512', B, 512, 16, 32, 512, 3, 1, 2, 2, 2), ('l4 1x1 512->2048', B, 512, 16, 32, 2048, 1, 1, 0, 1, 3),
    ('l4 1x1 2048->512', B, 2048, 16, 32, 512, 1, 1, 0, 1, 2), ('l4 ds 1x1 1024->2048', B, 1024, 16, 32, 2048, 1, 1, 0, 1, 1),
    ('aspp 1x1 2048->256', B, 2048, 16, 32, 256, 1, 1, 0, 1, 1), ('aspp 3x3 d6', B, 2048, 16, 32, 256, 3, 1, 6, 6, 1),
    ('aspp 3x3 d12', B, 2048, 16, 32, 256, 3, 1, 12, 12, 1), ('aspp 3x3 d18', B, 2048, 16, 32, 256, 3, 1, 18, 18, 1),
    ('aspp pool 1x1', B, 2048, 1, 1, 256, 1, 1, 0, 1, 1), ('aspp proj 1280->256', B, 1280, 16, 32, 256, 1, 1, 0, 1, 1),
    ('shortcut 256->48', B, 256, 64, 128, 48, 1, 1, 0, 1, 1), ('cat_conv.0 304->256', B, 304, 64, 128, 256, 3, 1, 1, 1, 1),
    ('cat_conv.4 256->256', B, 256, 64, 128, 256, 3, 1, 1, 1, 1), ('cls 256->19', B, 256, 64, 128, 19, 1, 1, 0, 1, 1),
    ('sisr 304->192', B, 304, 64, 128, 192, 3, 1, 1, 1, 1),
]


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser(); ap.add_argument('--reps', type=int, default=5); args = ap.parse_args()
    dev = 'cuda:0'
    tot = {'fwd': 0., 'dgrad': 0., 'wgrad': 0.}; totf = 0.
    print(f"{'layer':26s} {'GF':>7s} | {'fwd ms':>8s} {'TF':>6s} | {'dgrad ms':>8s} {'TF':>6s} | {'wgrad ms':>8s} {'TF':>6s} | x count")
    for name, N, C, H, W, K, R, stride, pad, dil, cnt in SHAPES:
        x = torch.randn((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        w = (torch.randn((K, C, R, R), device=dev) * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = HF.conv2d(x, w, None, stride, pad, dil)
        dy = torch.randn_like(y)
        gf = 2 * HF.conv2d_inbounds_macs(N, H, W, C, K, R, R, stride, pad, dil) / 1e9
        f = timeit(lambda: HF.conv2d(x.detach(), w.detach(), None, stride, pad, dil), args.reps)
        xd = x.detach().requires_grad_(True); wd = w.detach()
        yd = HF.conv2d(xd, wd, None, stride, pad, dil)
        d = timeit(lambda: torch.autograd.grad(yd, xd, dy, retain_graph=True), args.reps)
        xw = x.detach(); ww = w.detach().requires_grad_(True)
        yw = HF.conv2d(xw, ww, None, stride, pad, dil)
        g = timeit(lambda: torch.autograd.grad(yw, ww, dy, retain_graph=True), args.reps)
        print(f'{name:26s} {gf:7.2f} | {f:8.3f} {gf / f:6.1f} | {d:8.3f} {gf / d:6.1f} | {g:8.3f} {gf / g:6.1f} | x{cnt}')
        tot['fwd'] += f * cnt; tot['dgrad'] += d * cnt; tot['wgrad'] += g * cnt; totf += gf * cnt
    print(f"per step: fwd {tot['fwd']:.2f} ms ({totf / tot['fwd']:.1f} TF), dgrad {tot['dgrad']:.2f} ms ({totf / tot['dgrad']:.1f} TF), "
          f"wgrad {tot['wgrad']:.2f} ms ({totf / tot['wgrad']:.1f} TF); {totf:.1f} GFLOP per pass")


if __name__ == '__main__':
    main()
