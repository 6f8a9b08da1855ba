#!/usr/bin/env python3
"""Projection of a half-precision STORAGE format for BASELINE config 5, from launches that exist today.

Slice: the decoder head (cat_conv.0, cat_conv.4, SISR conv, cls_conv) and one layer3 bottleneck (conv1 / conv2 / conv3 + their three BatchNorms), forward and
backward, at config 5's per-GPU shapes (B = 8, 512x1024 input: decoder tensors 128x256, layer3 32x64).  Three columns per launch:

  f16x3   today's default: fp32 tensors, two fp16 terms per operand (register-staged kernel; the decoder forward convs on planes)
  f16x1   today's O1 / O2: fp32 tensors, ONE fp16 term formed while staging (register-staged kernel)
  fp16st  the conv a 2-byte storage format would run: conv_planes_kernel<NPL = 1> - both operands arrive as ONE fp16 plane (what a producer would have
          written), staged by LDS-DMA with no conversion; same MFMAs as f16x1.  Outputs are still written as fp32 here (the 2-byte epilogue does not exist
          yet: its stores would be half as many bytes, so this column is an upper bound of the conv time).

BatchNorm / element-wise kernels have no 2-byte build; their traffic halves with the format, and the column `half` times the SAME fp32 kernel on a tensor with
half the channels (= the bytes a 2-byte tensor of the full width moves; same launch geometry per byte).  Weight gradients: both operands are activations, the
f16x1 kernel converts them in its loop; a plane-fed weight-gradient kernel does not exist - the slice is reported with and without them.

Accuracy: fp16st forms the products of exactly the operand values f16x1 rounds to (hi = f16(x 2^e)), so the conv results are BIT-identical to the f16x1 register-
staged kernel for the same plan (checked here); what a storage format adds is the rounding of the BatchNorm inputs / outputs, not measured here.
Usage: python tools/fp16_slice.py [--reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualsuperreslearningforsemseg_amd import _lib, functional as HF  # noqa: E402
from dualsuperreslearningforsemseg_amd._lib import call  # noqa: E402

B = 8
# name, N, C, H, W, K, R, stride, pad, dil, has BatchNorm behind it (channels of the BN = K), residual on the BN
CONVS = [
    ('cat_conv.0 3x3 304->256', B, 304, 128, 256, 256, 3, 1, 1, 1, True, False), ('cat_conv.4 3x3 256->256', B, 256, 128, 256, 256, 3, 1, 1, 1, True, False),
    ('SISR 3x3 304->192', B, 304, 128, 256, 192, 3, 1, 1, 1, False, False), ('cls_conv 1x1 256->19', B, 256, 128, 256, 19, 1, 1, 0, 1, False, False),
    ('l3 conv1 1x1 1024->256', B, 1024, 32, 64, 256, 1, 1, 0, 1, True, False), ('l3 conv2 3x3 256->256', B, 256, 32, 64, 256, 3, 1, 1, 1, True, False),
    ('l3 conv3 1x1 256->1024', B, 256, 32, 64, 1024, 1, 1, 0, 1, True, True),
]


def timeit(fn, reps):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def bn_times(P, C, residual, reps, dev):
    """forward from statistics + backward from sums on a [P][C] fp32 tensor: (fwd us, bwd us)"""
    lib = _lib.load(); st = HF._stream()
    parts = 64
    x, y = torch.randn(P * C, device=dev), torch.empty(P * C, device=dev)
    r = torch.randn(P * C, device=dev) if residual else None
    dy, dx = torch.randn(P * C, device=dev), torch.empty(P * C, device=dev)
    dres = torch.empty(P * C, device=dev) if residual else None
    mean, inv, rm, rv, gam, bet, dg, db = [torch.ones(C, device=dev) for _ in range(8)]
    stats = torch.rand(int(lib.dsrl_bn_stats_floats(3, parts, C)), device=dev) + 1.0

    def fwd():
        _lib.check(lib.dsrl_bn_train_fwd_from_stats(x.data_ptr(), C, y.data_ptr(), C, P, C, 1e-5, 0.1, mean.data_ptr(), inv.data_ptr(), rm.data_ptr(), rv.data_ptr(), gam.data_ptr(),
                                                    bet.data_ptr(), r.data_ptr() if r is not None else None, C, 1, 0.0, 0, 0, stats.data_ptr(), parts, None, st), 'bn fwd')

    def bwd():
        _lib.check(lib.dsrl_bn_bwd_from_stats(x.data_ptr(), C, y.data_ptr(), C, dy.data_ptr(), C, dx.data_ptr(), C, dres.data_ptr() if dres is not None else None, C, P, C,
                                              mean.data_ptr(), inv.data_ptr(), gam.data_ptr(), dg.data_ptr(), db.data_ptr(), 1, 1, stats.data_ptr(), parts, None, st), 'bn bwd')
    return timeit(fwd, reps), timeit(bwd, reps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    args = ap.parse_args()
    dev = 'cuda:0'
    torch.manual_seed(0)
    tot = {k: 0.0 for k in ('conv f16x3', 'conv f16x1', 'conv fp16st', 'wgrad f16x3', 'wgrad f16x1', 'bn fp32', 'bn half')}
    print(f"{'launch':26s} |  fwd f16x3  f16x1 fp16st | dgrad f16x3  f16x1 fp16st | wgrad f16x3  f16x1 | bn fwd+bwd fp32   half | fp16st == f16x1")
    for name, N, C, H, W, K, R, stride, pad, dil, has_bn, res in CONVS:
        x = torch.randn((N, C, H, W), device=dev).clamp_(min=0).contiguous(memory_format=torch.channels_last)
        w = (torch.randn((K, C, R, R), device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        Ho, Wo = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1, (W + 2 * pad - dil * (R - 1) - 1) // stride + 1
        Kp = (K + 3) & ~3
        dy = torch.randn((N, Kp, Ho, Wo), device=dev).contiguous(memory_format=torch.channels_last)[:, :K]
        shp = (N, H, W, C, K, R, R, stride, pad, dil)
        st = HF._stream()
        ws = torch.empty(1 << 30, device=dev, dtype=torch.uint8)
        t = {}
        outs = {}
        planes_ok = C % 8 == 0 and K % 8 == 0
        for mode in ('f16x3', 'f16x1'):
            HF.set_conv_precision(mode)
            rec, wsp, wtsp, wt = HF.split_filter(w)
            xa = HF.amax_slot(x.device); xa.zero_(); call('dsrl_amax', x.data_ptr(), C, N * H * W, C, xa.data_ptr(), st)
            dya = HF.amax_slot(x.device); dya.zero_(); call('dsrl_amax', dy.data_ptr(), Kp, N * Ho * Wo, K, dya.data_ptr(), st)
            npl = 2 if mode == 'f16x3' else 1
            xp = dyp = wp = wtp = None
            if planes_ok:
                wp, wtp = HF.filter_planes(w, rec)
                xp, dyp = HF.planes_of(x, C, xa, npl), HF.planes_of(dy, Kp, dya, npl)
            y = HF.new_cl((N, K, Ho, Wo), x); dx = HF.new_cl((N, C, H, W), x); dw = torch.empty_like(w)
            p_ = lambda b: None if b is None else b.data_ptr()       # noqa: E731

            def fwd(pl):
                call('dsrl_conv2d_fwd_planes', x.data_ptr(), C, xa.data_ptr(), p_(xp) if pl else None, w.data_ptr(), rec.data_ptr(), wsp.data_ptr(), p_(wp) if pl else None, None,
                     y.data_ptr(), K, *shp, ws.data_ptr(), ws.numel(), None, 0, st)

            def dgrad(pl):
                call('dsrl_conv2d_dgrad_planes', dy.data_ptr(), Kp, dya.data_ptr(), p_(dyp) if pl else None, w.data_ptr(), None, rec.data_ptr(), wtsp.data_ptr(), p_(wtp) if pl else None,
                     dx.data_ptr(), C, *shp, ws.data_ptr(), ws.numel(), None, 0, None, 0, None, None, 0, None, 0, 0, st)

            def wgrad():
                call('dsrl_conv2d_wgrad_amax', x.data_ptr(), C, xa.data_ptr(), dy.data_ptr(), Kp, dya.data_ptr(), dw.data_ptr(), *shp, ws.data_ptr(), ws.numel(), st)
            # today's launch of the mode: register-staged, except that the f16x3 decoder forward convs take planes (DSRL_PLANES_MODE=auto)
            today_planes = mode == 'f16x3' and planes_ok and R == 3 and N * H * W * C >= (8 << 20)
            t[(mode, 'fwd')] = timeit(lambda: fwd(today_planes), args.reps)
            t[(mode, 'dgrad')] = timeit(lambda: dgrad(False), args.reps)
            t[(mode, 'wgrad')] = timeit(wgrad, args.reps)
            if mode == 'f16x1':
                fwd(False); dgrad(False); torch.cuda.synchronize()
                outs['reg'] = (y.clone(), dx.clone())
                if planes_ok:
                    t[('fp16st', 'fwd')] = timeit(lambda: fwd(True), args.reps)
                    t[('fp16st', 'dgrad')] = timeit(lambda: dgrad(True), args.reps)
                    fwd(True); dgrad(True); torch.cuda.synchronize()
                    outs['planes'] = (y.clone(), dx.clone())
                else:
                    t[('fp16st', 'fwd')], t[('fp16st', 'dgrad')] = t[('f16x1', 'fwd')], t[('f16x1', 'dgrad')]
        HF.set_conv_precision(None)
        same = 'n/a (channels not a multiple of 8)' if 'planes' not in outs else str(torch.equal(outs['reg'][0], outs['planes'][0]) and torch.equal(outs['reg'][1], outs['planes'][1]))
        bn32 = bnh = 0.0
        if has_bn:
            P = N * Ho * Wo
            f32, b32 = bn_times(P, K, res, args.reps, dev)
            fh, bh = bn_times(P, K // 2, res, args.reps, dev)
            bn32, bnh = f32 + b32, fh + bh
        print(f"{name:26s} | {t[('f16x3', 'fwd')]:10.1f} {t[('f16x1', 'fwd')]:6.1f} {t[('fp16st', 'fwd')]:6.1f} | {t[('f16x3', 'dgrad')]:11.1f} {t[('f16x1', 'dgrad')]:6.1f} {t[('fp16st', 'dgrad')]:6.1f} |"
              f" {t[('f16x3', 'wgrad')]:11.1f} {t[('f16x1', 'wgrad')]:6.1f} | {bn32:15.1f} {bnh:6.1f} | {same}", flush=True)
        tot['conv f16x3'] += t[('f16x3', 'fwd')] + t[('f16x3', 'dgrad')]; tot['conv f16x1'] += t[('f16x1', 'fwd')] + t[('f16x1', 'dgrad')]
        tot['conv fp16st'] += t[('fp16st', 'fwd')] + t[('fp16st', 'dgrad')]
        tot['wgrad f16x3'] += t[('f16x3', 'wgrad')]; tot['wgrad f16x1'] += t[('f16x1', 'wgrad')]
        tot['bn fp32'] += bn32; tot['bn half'] += bnh
        del x, w, dy, ws
        torch.cuda.empty_cache()
    print('slice totals, us:', {k: round(v, 1) for k, v in tot.items()})
    a = tot['conv f16x1'] + tot['bn fp32']; b = tot['conv fp16st'] + tot['bn half']
    print(f"forward + dgrad + BatchNorm:            f16x1 on fp32 storage {a:8.1f} us   fp16 storage (projected) {b:8.1f} us   -> {a / b:.2f}x")
    a2 = a + tot['wgrad f16x1']; b2 = b + tot['wgrad f16x1']
    print(f"... + weight gradients (unchanged):     f16x1 on fp32 storage {a2:8.1f} us   fp16 storage (projected) {b2:8.1f} us   -> {a2 / b2:.2f}x")
    c = tot['conv f16x3'] + tot['bn fp32'] + tot['wgrad f16x3']
    print(f"today's default f16x3 (fp32-equivalent) on the same slice: {c:8.1f} us")


if __name__ == '__main__':
    main()
