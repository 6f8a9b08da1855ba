#!/usr/bin/env python3
"""BatchNorm forward from conv-epilogue statistics (dsrl_bn_train_fwd_from_stats) launched alone on the step's tensor shapes: warm operands vs operand
sets rotated beyond the Infinity Cache; GB/s of algorithmic bytes (x read, residual read, y write)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualsuperreslearningforsemseg_amd import _lib
lib = _lib.load()
dev = 'cuda:0'; st = torch.cuda.current_stream().cuda_stream
for P, C, res, parts in ((4096, 256, 0, 128), (4096, 1024, 1, 64), (4096, 512, 0, 64), (16384, 128, 0, 256), (16384, 512, 1, 256), (65536, 64, 0, 1024), (65536, 256, 1, 1024), (65536, 256, 0, 1024)):
    for NB in (1, max(2, int(700e6 / (P * C * 4 * (3 if res else 2))))):
        sets = [[torch.randn(P * C, device=dev), torch.empty(P * C, device=dev), torch.randn(P * C, device=dev) if res else None] for _ in range(NB)]
        mean, inv, rm, rv, gam, bet = [torch.ones(C, device=dev) for _ in range(6)]
        stats = torch.rand(int(lib.dsrl_bn_stats_floats(3, parts, C)), device=dev) + 1.0
        def run(i):
            x, y, r = sets[i % NB]
            _lib.check(lib.dsrl_bn_train_fwd_from_stats(x.data_ptr(), C, y.data_ptr(), C, P, C, 1e-5, 0.1, mean.data_ptr(), inv.data_ptr(), rm.data_ptr(), rv.data_ptr(), gam.data_ptr(), bet.data_ptr(),
                                                        r.data_ptr() if r is not None else None, C, 1, 0.0, 0, 0, stats.data_ptr(), parts, None, st), 'bn')
        evs = []
        for i in range(80):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(i); b.record(); evs.append((a, b))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs[8:]); us = ts[len(ts) // 2] * 1e3
        by = P * C * 4 * (3 if res else 2)
        print(f'P {P:6d} C {C:5d} residual {res} parts {parts:5d} {"warm" if NB == 1 else f"{NB} sets"}: {us:6.1f} us  {by / us / 1e3:7.0f} GB/s', flush=True)
