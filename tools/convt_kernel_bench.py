#!/usr/bin/env python3
"""Kernel-only time of the ConvTranspose2d 19 -> 19 backward at the step's last layer (8 x 256 x 512 input) through the C ABI: HIP events around
back-to-back launches (kernel + finalize).  Usage: python tools/convt_kernel_bench.py [VAR=value ...]"""
import os, sys, torch
for kv in sys.argv[1:]:
    k, v = kv.split('='); os.environ[k] = v
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualsuperreslearningforsemseg_amd import functional as HF
from dualsuperreslearningforsemseg_amd._lib import call, query
dev = 'cuda:0'
for (N, H, W) in ((8, 128, 256), (8, 256, 512)):
    x = torch.randn(N, H, W, 19, device=dev); w = torch.randn(19, 19, 2, 2, device=dev); dy = torch.randn(N, 2 * H, 2 * W, 19, device=dev)
    dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty(19, device=dev)
    nbytes = query("dsrl_convt2x2_bwd_workspace_bytes", N, H, W, 19, 19)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    st = HF._stream()
    f = lambda: call('dsrl_convt2x2_bwd', x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), N, H, W, 19, 19, ws.data_ptr(), nbytes, st)
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / 20
    print(f'{N}x{H}x{W}: {t * 1e3:.1f} us per backward (kernel + finalize), {(2 * x.numel() + dy.numel()) * 4 / t / 1e6:.0f} GB/s algorithmic', flush=True)
