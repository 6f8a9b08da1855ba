#!/usr/bin/env python3
"""A/B of two builds of the library on the conv layer shapes of the step: time of forward / dgrad per shape and bitwise comparison of the results.
usage: ab_probe.py <reference .so name inside the package dir> (the new build is libdsrl_hip.so)"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualsuperreslearningforsemseg_amd import _lib
ref_name = sys.argv[1] if len(sys.argv) > 1 else 'libdsrl_hip_REF.so'
libs = {'ref': ctypes.CDLL(_lib.LIB_PATH.replace('libdsrl_hip.so', ref_name)), 'new': ctypes.CDLL(_lib.LIB_PATH)}
for l in libs.values():
    for name, (res, args) in _lib.PROTOTYPES.items():
        if hasattr(l, name):
            fn = getattr(l, name); fn.restype = res; fn.argtypes = args
SHAPES = {'l3_3x3': (8, 256, 16, 32, 256, 3, 1, 1, 1), 'l3_1x1_up': (8, 256, 16, 32, 1024, 1, 1, 0, 1), 'l3_1x1_dn': (8, 1024, 16, 32, 256, 1, 1, 0, 1),
          'l2_3x3': (8, 128, 32, 64, 128, 3, 1, 1, 1), 'l1_3x3': (8, 64, 64, 128, 64, 3, 1, 1, 1), 'l1_1x1_up': (8, 64, 64, 128, 256, 1, 1, 0, 1), 'l4_1x1_up': (8, 512, 16, 32, 2048, 1, 1, 0, 1),
          'l4_3x3': (8, 512, 16, 32, 512, 3, 1, 2, 2), 'aspp_d6': (8, 2048, 16, 32, 256, 3, 1, 6, 6), 'aspp_d18': (8, 2048, 16, 32, 256, 3, 1, 18, 18), 'cat0': (8, 304, 64, 128, 256, 3, 1, 1, 1),
          'cat4': (8, 256, 64, 128, 256, 3, 1, 1, 1), 'sisr': (8, 304, 64, 128, 192, 3, 1, 1, 1), 'cls': (8, 256, 64, 128, 19, 1, 1, 0, 1), 'odd': (3, 68, 33, 47, 100, 3, 2, 1, 1),
          'l2_ds': (8, 256, 64, 128, 512, 1, 2, 0, 1)}
dev = 'cuda:0'
def t_us(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
tot = {'ref': [0.0, 0.0], 'new': [0.0, 0.0]}
for name, (N, C, H, W, K, R, stride, pad, dil) in SHAPES.items():
    Ho = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1; Wo = (W + 2 * pad - dil * (R - 1) - 1) // stride + 1
    Kp = (K + 3) & ~3
    torch.manual_seed(0)
    x = torch.randn(N * H * W * C, device=dev); w = torch.randn(K * R * R * C, device=dev) * 0.05; dy = torch.randn(N * Ho * Wo * Kp, device=dev)
    shp = (N, H, W, C, K, R, R, stride, pad, dil)
    ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev); st = torch.cuda.current_stream().cuda_stream
    out, res = [], {}
    for tag, l in libs.items():
        y = torch.empty(N * Ho * Wo * K, device=dev); dx = torch.empty(N * H * W * C, device=dev)
        f = t_us(lambda: l.dsrl_conv2d_fwd(x.data_ptr(), C, w.data_ptr(), None, y.data_ptr(), K, *shp, ws.data_ptr(), ws.numel(), st))
        d = t_us(lambda: l.dsrl_conv2d_dgrad(dy.data_ptr(), Kp, w.data_ptr(), None, dx.data_ptr(), C, *shp, ws.data_ptr(), ws.numel(), st))
        res[tag] = (y.clone(), dx.clone())
        tot[tag][0] += f; tot[tag][1] += d
        out.append(f'{tag}: fwd {f:.1f} dgrad {d:.1f}')
    same = torch.equal(res['ref'][0], res['new'][0]) and torch.equal(res['ref'][1], res['new'][1])
    print(f'{name:10s}', ' | '.join(out), '==' if same else 'DIFFERENT', flush=True)
print('sum       ', ' | '.join(f'{k}: fwd {v[0]:.0f} dgrad {v[1]:.0f}' for k, v in tot.items()))
