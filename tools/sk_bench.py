#!/usr/bin/env python3
"""128x128 tiles / two K groups / split-K across workgroups reduced inside the launch (conv_sk.hip) against the planner's current pick on the M = 4096
layer shapes of the step: forward and dgrad, back-to-back launch times (HIP events), results compared (cooperative == slabs + reduce bitwise; against the
current plan to fp32 summation-order tolerance).  Usage: python tools/sk_bench.py [--reps 30] [--only l3] [--splits 1,2,4,8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualsuperreslearningforsemseg_amd import functional as HF  # noqa: E402
from dualsuperreslearningforsemseg_amd._lib import call  # noqa: E402

B = 8
SHAPES = [
    ('l3 3x3 256', B, 256, 16, 32, 256, 3, 1, 1, 1, 23), ('l3 1x1 256->1024', B, 256, 16, 32, 1024, 1, 1, 0, 1, 23),
    ('l3 1x1 1024->256', B, 1024, 16, 32, 256, 1, 1, 0, 1, 22),
    ('l4 3x3 d2 512', B, 512, 16, 32, 512, 3, 1, 2, 2, 3), ('l4 1x1 512->2048', B, 512, 16, 32, 2048, 1, 1, 0, 1, 3),
    ('l4 1x1 2048->512', B, 2048, 16, 32, 512, 1, 1, 0, 1, 2), ('l4 1x1 1024->512', B, 1024, 16, 32, 512, 1, 1, 0, 1, 1),
    ('l4 ds 1024->2048', B, 1024, 16, 32, 2048, 1, 1, 0, 1, 1),
    ('aspp 1x1 2048->256', B, 2048, 16, 32, 256, 1, 1, 0, 1, 1), ('aspp 3x3 d6', B, 2048, 16, 32, 256, 3, 1, 6, 6, 1), ('aspp 3x3 d12', B, 2048, 16, 32, 256, 3, 1, 12, 12, 1),
    ('aspp 3x3 d18', B, 2048, 16, 32, 256, 3, 1, 18, 18, 1), ('aspp proj 1280->256', B, 1280, 16, 32, 256, 1, 1, 0, 1, 1),
    ('l2 3x3 128', B, 128, 32, 64, 128, 3, 1, 1, 1, 4), ('l2 1x1 128->512', B, 128, 32, 64, 512, 1, 1, 0, 1, 4), ('l2 1x1 512->128', B, 512, 32, 64, 128, 1, 1, 0, 1, 3),
]
KNOBS = ('DSRL_FORCE_CFG', 'DSRL_FORCE_KG', 'DSRL_FORCE_SPLITS', 'DSRL_SK_COOP')


def timeit(fn, reps):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def setenv(**kw):
    for k in KNOBS:
        os.environ.pop(k, None)
    for k, v in kw.items():
        os.environ[k] = str(v)
    HF._query_cache.clear()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--only', default='')
    ap.add_argument('--splits', default='1,2,4,8')
    args = ap.parse_args()
    splits = [int(v) for v in args.splits.split(',')]
    dev = 'cuda:0'
    torch.manual_seed(0)
    tot = {}
    print('us per launch: auto = the planner\'s pick; sN = 128x128 / 2 K groups / split-K N cooperative (in-launch reduction), sNr = the same with slabs + reduce launch')
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
        xa = HF.amax_slot(x.device); xa.zero_()
        call('dsrl_amax', x.data_ptr(), C, N * H * W, C, xa.data_ptr(), HF._stream())
        dya = HF.amax_slot(x.device); dya.zero_()
        call('dsrl_amax', dy.data_ptr(), K, N * Ho * Wo, K, dya.data_ptr(), HF._stream())
        ws = torch.empty(512 << 20, device=dev, dtype=torch.uint8)
        st = HF._stream()

        def fwd(y):
            call('dsrl_conv2d_fwd_planes', x.data_ptr(), C, xa.data_ptr(), None, w.data_ptr(), rec.data_ptr(), wsp.data_ptr(), None, None, y.data_ptr(), K, *shp,
                 ws.data_ptr(), ws.numel(), None, 0, st)

        def dgrad(dx):
            call('dsrl_conv2d_dgrad_planes', dy.data_ptr(), K, dya.data_ptr(), None, w.data_ptr(), None, rec.data_ptr(), wtsp.data_ptr(), None, dx.data_ptr(), C, *shp,
                 ws.data_ptr(), ws.numel(), None, 0, None, 0, None, None, 0, None, 0, 0, st)

        for what, fn, shape_out in (('fwd', fwd, (N, K, Ho, Wo)), ('dgrad', dgrad, (N, C, H, W))):
            like = x
            ref = HF.new_cl(shape_out, like); out = HF.new_cl(shape_out, like); out2 = HF.new_cl(shape_out, like)
            setenv()
            ref.zero_(); fn(ref); torch.cuda.synchronize()
            t_auto = timeit(lambda: fn(ref), args.reps)
            cols = [f'auto {t_auto:6.1f}']
            tot.setdefault((what, 'auto'), 0.0); tot[(what, 'auto')] += t_auto * cnt
            best = t_auto
            for sp in splits:
                setenv(DSRL_FORCE_CFG=0, DSRL_FORCE_KG=2, DSRL_FORCE_SPLITS=sp, DSRL_SK_COOP=1)
                out.fill_(7.0); fn(out); torch.cuda.synchronize()
                err = (out - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
                t_c = timeit(lambda: fn(out), args.reps)
                tag = f's{sp} {t_c:6.1f}'
                if sp > 1:
                    setenv(DSRL_FORCE_CFG=0, DSRL_FORCE_KG=2, DSRL_FORCE_SPLITS=sp, DSRL_SK_COOP=0)
                    out2.fill_(3.0); fn(out2); torch.cuda.synchronize()
                    t_r = timeit(lambda: fn(out2), args.reps)
                    tag += f' (r {t_r:6.1f} {"==" if torch.equal(out, out2) else "!="})'
                tag += f' e{err:.0e}'
                cols.append(tag)
                best = min(best, t_c)
                tot.setdefault((what, f's{sp}'), 0.0); tot[(what, f's{sp}')] += t_c * cnt
            tot.setdefault((what, 'best'), 0.0); tot[(what, 'best')] += best * cnt
            print(f'{name:22s} {what:5s} x{cnt:2d} | ' + ' | '.join(cols), flush=True)
        setenv()
    for k in sorted(tot):
        print(f'per step (listed layers) {k[0]:5s} {k[1]:5s}: {tot[k] / 1e3:.3f} ms')


if __name__ == '__main__':
    main()
