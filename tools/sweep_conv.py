#!/usr/bin/env python3
"""Sweeps the split / tile-config knobs of the MFMA conv kernels on one shape through the raw C ABI (HIP-event timed)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualsuperreslearningforsemseg_amd import _lib  # noqa: E402

lib = _lib.load()


def t_ms(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def run(N, C, H, W, K, R, stride, pad, dil, what):
    dev = 'cuda:0'
    Ho = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1; Wo = (W + 2 * pad - dil * (R - 1) - 1) // stride + 1
    x = torch.randn(N * H * W * C, device=dev); w = torch.randn(K * R * R * C, device=dev) * 0.05
    y = torch.randn(N * Ho * Wo * K, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w)
    shp = (N, H, W, C, K, R, R, stride, pad, dil)
    ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    gf = 2 * lib.dsrl_conv2d_inbounds_macs(*shp) / 1e9
    # f16x3: operand magnitudes and the pre-split filter as the training step has them (left by the producers / the per-step filter pass)
    from dualsuperreslearningforsemseg_amd import functional as HF
    xa = ya = wa = wsp = wtsp = wtr = None
    if HF.get_conv_precision() == 'f16x3':
        rec, sp, tsp, tr = HF.split_filter(w.view(K, R, R, C).permute(0, 3, 1, 2))
        xs, ys = HF.amax_slot(x.device), HF.amax_slot(x.device)
        HF.call('dsrl_amax', x.data_ptr(), C, N * H * W, C, xs.data_ptr(), st)
        HF.call('dsrl_amax', y.data_ptr(), K, N * Ho * Wo, K, ys.data_ptr(), st)
        keep = (rec, sp, tsp, tr, xs, ys)
        xa, ya, wa, wsp, wtsp, wtr = xs.data_ptr(), ys.data_ptr(), rec.data_ptr(), sp.data_ptr(), tsp.data_ptr(), tr.data_ptr()
    if what == 'fwd':
        f = lambda: _lib.check(lib.dsrl_conv2d_fwd_amax(x.data_ptr(), C, xa, w.data_ptr(), wa, wsp, None, y.data_ptr(), K, *shp, ws.data_ptr(), ws.numel(), None, 0, st), 'fwd')
    elif what == 'dgrad':
        f = lambda: _lib.check(lib.dsrl_conv2d_dgrad_amax(y.data_ptr(), K, ya, w.data_ptr(), wtr, wa, wtsp, dx.data_ptr(), C, *shp, ws.data_ptr(), ws.numel(),
                                                          None, 0, None, 0, None, None, 0, None, 0, 0, st), 'dgrad')
    else:
        f = lambda: _lib.check(lib.dsrl_conv2d_wgrad_amax(x.data_ptr(), C, xa, y.data_ptr(), K, ya, dw.data_ptr(), *shp, ws.data_ptr(), ws.numel(), st), 'wgrad')
    ms = t_ms(f)
    return ms, gf / ms


SHAPES = {'l3_3x3': (8, 256, 16, 32, 256, 3, 1, 1, 1), 'l3_1x1_up': (8, 256, 16, 32, 1024, 1, 1, 0, 1), 'l3_1x1_dn': (8, 1024, 16, 32, 256, 1, 1, 0, 1),
          'cat0': (8, 304, 64, 128, 256, 3, 1, 1, 1), 'l2_3x3': (8, 128, 32, 64, 128, 3, 1, 1, 1), 'l1_3x3': (8, 64, 64, 128, 64, 3, 1, 1, 1)}

if __name__ == '__main__':
    for name, shp in SHAPES.items():
        for what, knob, vals in (('fwd', 'DSRL_FORCE_SPLITS', [0, 1, 2, 4, 8, 16]), ('dgrad', 'DSRL_FORCE_SPLITS', [0, 1, 2, 4, 8]),
                                 ('wgrad', 'DSRL_FORCE_PSPLITS', [0, 1, 2, 4, 8, 16, 32])):
            res = []
            for v in vals:
                if v:
                    os.environ[knob] = str(v)
                else:
                    os.environ.pop(knob, None)
                ms, tf = run(*shp, what)
                res.append(f'{v}:{ms * 1e3:.0f}us/{tf:.0f}TF')
            os.environ.pop(knob, None)
            print(f'{name:10s} {what:6s} {knob[11:]:8s} ' + '  '.join(res), flush=True)
        for cfg in (0, 1, 2):
            os.environ['DSRL_FORCE_CFG'] = str(cfg)
            res = [f"{what}:{run(*shp, what)[1]:.0f}TF" for what in ('fwd', 'dgrad', 'wgrad')]
            print(f'{name:10s} cfg{cfg} (0=128x128,1=256x64,2=256x32) ' + '  '.join(res), flush=True)
        os.environ.pop('DSRL_FORCE_CFG', None)
