#!/usr/bin/env python3
"""What one more launch costs inside a replayed hipGraph: N dependent launches of a kernel that does (almost) nothing, captured on one stream and replayed;
time per launch from HIP events around the replay.  The step's trace shows every kernel at >= 4.7 us whatever it does: this measures that floor alone."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualsuperreslearningforsemseg_amd import functional as HF
from dualsuperreslearningforsemseg_amd._lib import call
dev = torch.device('cuda:0')
flag = torch.zeros(1, dtype=torch.int32, device=dev)
for n_el, label in ((4, '4 floats, one block'), (1 << 16, '256 KB'), (1 << 22, '16 MB')):
    t = torch.zeros(n_el, device=dev)
    for N in (200, 800):
        def f():
            st = HF._stream()
            for _ in range(N):
                call('dsrl_nan_check', t.data_ptr(), n_el, flag.data_ptr(), st)
        f(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            f()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): g.replay()
        b.record(); torch.cuda.synchronize()
        print(f'nan_check over {label}: {N} dependent launches per replay: {a.elapsed_time(b) / 5 / N * 1e3:.2f} us per launch', flush=True)
