#!/usr/bin/env python3
"""ConvTranspose2d 19 -> 19 (k2 s2) forward / backward at the two tail shapes of the step: time and algorithmic GB/s, fused backward vs the
separate dx / dw kernels (DSRL_CONVT_FUSED_BWD)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dualsuperreslearningforsemseg_amd as D
from dualsuperreslearningforsemseg_amd import functional as HF
dev = torch.device('cuda:0')
for (N, H, W) in ((8, 128, 256), (8, 256, 512)):
    x = torch.randn(N, 19, H, W, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.randn(19, 19, 2, 2, device=dev).requires_grad_(True); b = torch.randn(19, device=dev).requires_grad_(True)
    dy = torch.randn(N, 19, 2 * H, 2 * W, device=dev).contiguous(memory_format=torch.channels_last)
    xb, yb = N * H * W * 19 * 4, N * 4 * H * W * 19 * 4
    for fused, dma in (('1', '1'), ('1', '0'), ('0', '0')):
        os.environ['DSRL_CONVT_FUSED_BWD'] = fused; os.environ['DSRL_CONVT_DMA'] = dma
        tf = tb = 0.0
        for it in range(12):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            x.grad = w.grad = b.grad = None
            e[0].record(); y = HF.conv_transpose2d_k2s2(x, w, b); e[1].record(); y.backward(dy); e[2].record()
            torch.cuda.synchronize()
            if it >= 2: tf += e[0].elapsed_time(e[1]) / 10; tb += e[1].elapsed_time(e[2]) / 10
        print(f'{N}x19x{H}x{W} fused_bwd={fused} dma={dma}: fwd {tf*1e3:.0f} us ({(xb+yb)/tf/1e6:.0f} GB/s)  bwd {tb*1e3:.0f} us ({(2*xb+yb)/tb/1e6:.0f} GB/s algorithmic)', flush=True)
