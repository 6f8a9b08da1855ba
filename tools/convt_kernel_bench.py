#!/usr/bin/env python3
"""Kernel-only times of the logits tail at the step's last layer (8 x 256 x 512 input, 19 -> 19) through the C ABI, HIP events around back-to-back
launches: ConvTranspose backward (kernel + finalize), the loss pass with and without the gradient write, and the backward that forms the CE gradient
inside the kernel (dsrl_convt2x2_bwd_ce).  Usage: python tools/convt_kernel_bench.py [VAR=value ...]"""
import os, sys, torch
for kv in sys.argv[1:]:
    k, v = kv.split('='); os.environ[k] = v
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualsuperreslearningforsemseg_amd import functional as HF
from dualsuperreslearningforsemseg_amd._lib import call, query
dev = 'cuda:0'


def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for (N, H, W) in ((8, 128, 256), (8, 256, 512)):
    C = 19
    x = torch.randn(N, H, W, C, device=dev); w = torch.randn(C, C, 2, 2, device=dev); dy = torch.randn(N, 2 * H, 2 * W, C, device=dev)
    tg = torch.randint(0, C, (N, 2 * H, 2 * W), device=dev, dtype=torch.uint8)
    dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty(C, device=dev); dl = torch.empty_like(dy)
    nbytes = query('dsrl_convt2x2_bwd_workspace_bytes', N, H, W, C, C)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    P = N * 4 * H * W
    wsc = torch.empty(query('dsrl_ce_fused_workspace_bytes', P), dtype=torch.uint8, device=dev)
    scal = torch.zeros(8, device=dev); flag = torch.zeros(1, dtype=torch.int32, device=dev)
    ftg = torch.randn(N, H // 4, W // 4, device=dev); ftw = torch.randn(C, device=dev)
    st = HF._stream()
    t_b = timeit(lambda: call('dsrl_convt2x2_bwd', x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), N, H, W, C, C, ws.data_ptr(), nbytes, st))
    t_ce = timeit(lambda: call('dsrl_ce_fused', dy.data_ptr(), C, tg.data_ptr(), P, C, 255, dl.data_ptr(), C, scal.data_ptr(), flag.data_ptr(), wsc.data_ptr(), wsc.numel(), st))
    t_cef = timeit(lambda: call('dsrl_ce_fused', dy.data_ptr(), C, tg.data_ptr(), P, C, 255, None, C, scal.data_ptr(), flag.data_ptr(), wsc.data_ptr(), wsc.numel(), st))
    line = f'{N}x{H}x{W}: convT backward {t_b:.1f} us ({(2 * x.numel() + dy.numel()) * 4 / t_b / 1e6:.0f} GB/s) | CE value + gradient {t_ce:.1f} us | CE value only {t_cef:.1f} us'
    if query('dsrl_convt2x2_bwd_ce_supported', x.data_ptr(), dy.data_ptr(), tg.data_ptr(), N, H, W, C, C):
        t_f = timeit(lambda: call('dsrl_convt2x2_bwd_ce', x.data_ptr(), w.data_ptr(), dy.data_ptr(), tg.data_ptr(), 255, scal.data_ptr() + 4, ftg.data_ptr(), ftw.data_ptr(), 8,
                                  dx.data_ptr(), dw.data_ptr(), db.data_ptr(), N, H, W, C, C, ws.data_ptr(), nbytes, st))
        line += f' | convT backward with the CE gradient inside {t_f:.1f} us  => {t_ce + t_b:.1f} -> {t_cef + t_f:.1f} us'
    print(line, flush=True)
