#!/usr/bin/env python3
"""conv_planes_kernel (operands as fp16 planes, LDS-DMA staging) against conv_igemm_split_kernel (f16x3, pre-split filters) on the step's layer
shapes: bitwise comparison of forward / dgrad outputs and back-to-back launch times.  Usage: python tools/planes_bench.py [--reps 20] [--only l3]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualsuperreslearningforsemseg_amd import functional as HF  # noqa: E402
from dualsuperreslearningforsemseg_amd._lib import call  # noqa: E402

B = 8
SHAPES = [
    ('l1 1x1 64->64', B, 64, 64, 128, 64, 1, 1, 0, 1, 1), ('l1 3x3 64', B, 64, 64, 128, 64, 3, 1, 1, 1, 3),
    ('l1 1x1 64->256', B, 64, 64, 128, 256, 1, 1, 0, 1, 4), ('l1 1x1 256->64', B, 256, 64, 128, 64, 1, 1, 0, 1, 2),
    ('l2 3x3 128', B, 128, 32, 64, 128, 3, 1, 1, 1, 3), ('l2 1x1 128->512', B, 128, 32, 64, 512, 1, 1, 0, 1, 4),
    ('l2 1x1 512->128', B, 512, 32, 64, 128, 1, 1, 0, 1, 3),
    ('l3 3x3 256', B, 256, 16, 32, 256, 3, 1, 1, 1, 22), ('l3 1x1 256->1024', B, 256, 16, 32, 1024, 1, 1, 0, 1, 23),
    ('l3 1x1 1024->256', B, 1024, 16, 32, 256, 1, 1, 0, 1, 22),
    ('l4 3x3 d2 512', B, 512, 16, 32, 512, 3, 1, 2, 2, 2), ('l4 1x1 512->2048', B, 512, 16, 32, 2048, 1, 1, 0, 1, 3),
    ('l4 1x1 2048->512', B, 2048, 16, 32, 512, 1, 1, 0, 1, 2),
    ('aspp 1x1 2048->256', B, 2048, 16, 32, 256, 1, 1, 0, 1, 1), ('aspp 3x3 d12', B, 2048, 16, 32, 256, 3, 1, 12, 12, 1),
    ('aspp proj 1280->256', B, 1280, 16, 32, 256, 1, 1, 0, 1, 1),
    ('shortcut 256->48', B, 256, 64, 128, 48, 1, 1, 0, 1, 1), ('cat_conv.0 304->256', B, 304, 64, 128, 256, 3, 1, 1, 1, 1),
    ('cat_conv.4 256->256', B, 256, 64, 128, 256, 3, 1, 1, 1, 1), ('sisr 304->192', B, 304, 64, 128, 192, 3, 1, 1, 1, 1),
]


def timeit(fn, reps):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    dev = 'cuda:0'
    torch.manual_seed(0)
    tot = [0., 0., 0., 0.]
    print(f"{'layer':24s} | fwd igemm us  planes us  equal | dgrad igemm us  planes us  equal | split x us")
    for name, N, C, H, W, K, R, stride, pad, dil, cnt in SHAPES:
        if args.only and args.only not in name:
            continue
        x = torch.randn((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn((K, C, R, R), device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        Ho = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dil * (R - 1) - 1) // stride + 1
        dy = torch.randn((N, K, Ho, Wo), device=dev).contiguous(memory_format=torch.channels_last)
        shp = (N, H, W, C, K, R, R, stride, pad, dil)
        rec, wsp, wtsp, wt = HF.split_filter(w)
        wp, wtp = HF.filter_planes(w, rec)
        xa = HF.amax_slot(x.device); xa.zero_()
        call('dsrl_amax', x.data_ptr(), C, N * H * W, C, xa.data_ptr(), HF._stream())
        dya = HF.amax_slot(x.device); dya.zero_()
        call('dsrl_amax', dy.data_ptr(), K, N * Ho * Wo, K, dya.data_ptr(), HF._stream())
        xp = HF.planes_of(x, C, xa)
        dyp = HF.planes_of(dy, K, dya)
        ws = HF._ws(HF.cquery('dsrl_conv2d_fwd_workspace_bytes', *shp), x)
        wsd = HF._ws(HF.cquery('dsrl_conv2d_dgrad_workspace_bytes', *shp), x)
        y0, y1 = HF.new_cl((N, K, Ho, Wo), x), HF.new_cl((N, K, Ho, Wo), x)
        dx0, dx1 = HF.new_cl((N, C, H, W), x), HF.new_cl((N, C, H, W), x)
        st = HF._stream()

        def fwd(planes, y):
            call('dsrl_conv2d_fwd_planes', x.data_ptr(), C, xa.data_ptr(), xp.data_ptr() if planes else None, w.data_ptr(), rec.data_ptr(), wsp.data_ptr(),
                 wp.data_ptr() if planes else None, None, y.data_ptr(), K, *shp, ws.data_ptr(), ws.numel(), None, 0, st)

        def dgrad(planes, dx):
            call('dsrl_conv2d_dgrad_planes', dy.data_ptr(), K, dya.data_ptr(), dyp.data_ptr() if planes else None, w.data_ptr(), None, rec.data_ptr(), wtsp.data_ptr(),
                 wtp.data_ptr() if planes else None, dx.data_ptr(), C, *shp, wsd.data_ptr(), wsd.numel(), None, 0, None, 0, None, None, 0, None, 0, 0, st)

        y0.zero_(); y1.fill_(1.0); dx0.zero_(); dx1.fill_(1.0)
        fwd(False, y0); fwd(True, y1); dgrad(False, dx0); dgrad(True, dx1)
        torch.cuda.synchronize()
        eqf, eqd = torch.equal(y0, y1), torch.equal(dx0, dx1)
        errf = (y0 - y1).abs().max().item() / max(y0.abs().max().item(), 1e-30)
        errd = (dx0 - dx1).abs().max().item() / max(dx0.abs().max().item(), 1e-30)
        tf0, tf1 = timeit(lambda: fwd(False, y0), args.reps), timeit(lambda: fwd(True, y1), args.reps)
        td0, td1 = timeit(lambda: dgrad(False, dx0), args.reps), timeit(lambda: dgrad(True, dx1), args.reps)
        tsx = timeit(lambda: HF.planes_of(x, C, xa), args.reps)
        print(f'{name:24s} | {tf0:9.1f} {tf1:10.1f}  {str(eqf):5s} {errf:.1e} | {td0:9.1f} {td1:10.1f}  {str(eqd):5s} {errd:.1e} | {tsx:7.1f}  x{cnt}', flush=True)
        tot[0] += tf0 * cnt; tot[1] += tf1 * cnt; tot[2] += td0 * cnt; tot[3] += td1 * cnt
    print(f'per step (listed layers): fwd {tot[0] / 1e3:.2f} -> {tot[1] / 1e3:.2f} ms, dgrad {tot[2] / 1e3:.2f} -> {tot[3] / 1e3:.2f} ms')


if __name__ == '__main__':
    main()
