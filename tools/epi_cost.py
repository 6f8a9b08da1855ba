#!/usr/bin/env python3
"""What the epilogues of the M = 4096 conv launches cost: forward with / without the BatchNorm-statistics epilogue, data gradient with / without the
BatchNorm-sum epilogue and the accumulate read-modify-write, launched back to back on warm operands (tools/sweep_conv.py conventions)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dualsuperreslearningforsemseg_amd import _lib, functional as HF
from sweep_conv import t_ms
lib = _lib.load()
SH = {'l3_3x3': (8, 256, 16, 32, 256, 3, 1, 1, 1), 'l3_1x1_dn': (8, 1024, 16, 32, 256, 1, 1, 0, 1), 'l3_1x1_up': (8, 256, 16, 32, 1024, 1, 1, 0, 1)}
dev = 'cuda:0'
for name, (N, C, H, W, K, R, stride, pad, dil) in SH.items():
    x = torch.randn(N * H * W * C, device=dev); w = torch.randn(K * R * R * C, device=dev) * 0.05
    y = torch.randn(N * H * W * K, device=dev); dx = torch.zeros_like(x)
    shp = (N, H, W, C, K, R, R, stride, pad, dil)
    ws = torch.empty(1 << 28, dtype=torch.uint8, device=dev); st = torch.cuda.current_stream().cuda_stream
    rec, sp, tsp, tr = HF.split_filter(w.view(K, R, R, C).permute(0, 3, 1, 2))
    xs, ys = HF.amax_slot(x.device), HF.amax_slot(x.device)
    HF.call('dsrl_amax', x.data_ptr(), C, N * H * W, C, xs.data_ptr(), st); HF.call('dsrl_amax', y.data_ptr(), K, N * H * W, K, ys.data_ptr(), st)
    parts = int(lib.dsrl_conv2d_fwd_stats_parts(*shp)); bparts = int(lib.dsrl_conv2d_dgrad_stats_parts(*shp))
    stats = torch.empty(int(lib.dsrl_bn_stats_floats(3, max(parts, 1), K)), device=dev); bstats = torch.empty(int(lib.dsrl_bn_stats_floats(2, max(bparts, 1), C)), device=dev)
    mean = torch.zeros(C, device=dev); inv = torch.ones(C, device=dev); bnx = torch.randn_like(x)
    def fwd(s):
        return lambda: _lib.check(lib.dsrl_conv2d_fwd_amax(x.data_ptr(), C, xs.data_ptr(), w.data_ptr(), rec.data_ptr(), sp.data_ptr(), None, y.data_ptr(), K, *shp, ws.data_ptr(), ws.numel(),
                                                           stats.data_ptr() if s else None, parts if s else 0, st), 'fwd')
    def dg(b, acc):
        return lambda: _lib.check(lib.dsrl_conv2d_dgrad_amax(y.data_ptr(), K, ys.data_ptr(), w.data_ptr(), tr.data_ptr(), rec.data_ptr(), tsp.data_ptr(), dx.data_ptr(), C, *shp, ws.data_ptr(), ws.numel(),
                                                             bnx.data_ptr() if b else None, C, x.data_ptr() if b else None, C, mean.data_ptr() if b else None, inv.data_ptr() if b else None, 1,
                                                             bstats.data_ptr() if b else None, bparts if b else 0, acc, st), 'dgrad')
    r = [t_ms(f, 50) * 1e3 for f in (fwd(False), fwd(True), dg(False, 0), dg(True, 0), dg(False, 1))]
    print(f'{name:10s} parts {parts}/{bparts}  fwd {r[0]:.1f} us, +stats epilogue {r[1]:.1f}; dgrad {r[2]:.1f}, +BN sums {r[3]:.1f}, accumulate {r[4]:.1f}', flush=True)
