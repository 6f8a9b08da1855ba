#!/usr/bin/env python3
"""Latency of the feature-affinity loss kernels (forward = similarity + all-pairs + finalize launches, backward) at the step's shapes:
B = 8 maps of 64x128 (n = 256 pairs per side) and of 128x256 (config 5: n = 1024)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualsuperreslearningforsemseg_amd import functional as HF
dev = 'cuda:0'
for H, W in ((64, 128), (128, 256)):
    a = torch.rand(8, 1, H, W, device=dev).requires_grad_(True); b = torch.rand(8, 1, H, W, device=dev).requires_grad_(True)
    def fwd():
        return HF.fa_loss(a, b, 8)
    l = fwd()
    def bwd():
        return torch.autograd.grad(l, (a, b), retain_graph=True)
    for name, f in (('forward', fwd), ('backward', bwd)):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record(); torch.cuda.synchronize()
        print(f'fa {name} {H}x{W} (n = {(W // 8) ** 2}): {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (host-enqueued back to back, incl. the small allocations of the wrapper)', flush=True)
