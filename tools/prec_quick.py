#!/usr/bin/env python3
"""Timing of the forward / dgrad kernels of a few shapes in one precision mode (DSRL_CONV_PRECISION and tuning variables from the environment)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dualsuperreslearningforsemseg_amd import _lib
from sweep_conv import SHAPES, t_ms
lib = _lib.load()
SHAPES['l4_3x3'] = (8, 512, 16, 32, 512, 3, 1, 2, 2)
SHAPES['aspp_d6'] = (8, 2048, 16, 32, 256, 3, 1, 6, 6)
SHAPES['sisr'] = (8, 304, 64, 128, 192, 3, 1, 1, 1)
dev = 'cuda:0'
names = sys.argv[1:] or list(SHAPES)
out = []
for name in names:
    N, C, H, W, K, R, stride, pad, dil = SHAPES[name]
    Ho = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1; Wo = (W + 2 * pad - dil * (R - 1) - 1) // stride + 1
    torch.manual_seed(0)
    x = torch.relu(torch.randn(N * H * W * C, device=dev)); w = torch.randn(K * R * R * C, device=dev) * (2.0 / (C * R * R)) ** 0.5
    dy = torch.randn(N * Ho * Wo * K, device=dev)
    shp = (N, H, W, C, K, R, R, stride, pad, dil)
    ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    gf = 2 * lib.dsrl_conv2d_inbounds_macs(*shp) / 1e9
    o = torch.empty(N * Ho * Wo * K, device=dev); o2 = torch.empty(N * H * W * C, device=dev); o3 = torch.empty(K * R * R * C, device=dev)
    tf = t_ms(lambda: _lib.check(lib.dsrl_conv2d_fwd(x.data_ptr(), C, w.data_ptr(), None, o.data_ptr(), K, *shp, ws.data_ptr(), ws.numel(), st), 'fwd'), 10)
    td = t_ms(lambda: _lib.check(lib.dsrl_conv2d_dgrad(dy.data_ptr(), K, w.data_ptr(), None, o2.data_ptr(), C, *shp, ws.data_ptr(), ws.numel(), st), 'dgrad'), 10)
    tw = t_ms(lambda: _lib.check(lib.dsrl_conv2d_wgrad(x.data_ptr(), C, dy.data_ptr(), K, o3.data_ptr(), *shp, ws.data_ptr(), ws.numel(), st), 'wgrad'), 10)
    out.append(f'{name} f{tf*1e3:.0f}/d{td*1e3:.0f}/w{tw*1e3:.0f}us')
print(' '.join(out), flush=True)
