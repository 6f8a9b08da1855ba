#!/usr/bin/env python3
"""How much of a from-statistics BatchNorm forward launch is its statistics merge?  The same launch with 128 / 64 row blocks of partials and with ONE (a merge of
nothing: what a launch would cost if the producing conv had already finalised scale / shift), back to back in a hipGraph of 40 launches (kernel boundary included)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualsuperreslearningforsemseg_amd import _lib
lib = _lib.load()
dev = 'cuda:0'
for P, C, res, parts in ((4096, 256, 0, 128), (4096, 1024, 1, 64), (4096, 512, 0, 64), (16384, 128, 0, 256), (65536, 64, 0, 1024)):
    out = []
    for np_ in (parts, 1):
        x, y = torch.randn(P * C, device=dev), torch.empty(P * C, device=dev)
        r = torch.randn(P * C, device=dev) if res else None
        mean, inv, rm, rv, gam, bet = [torch.ones(C, device=dev) for _ in range(6)]
        stats = torch.rand(int(lib.dsrl_bn_stats_floats(3, np_, C)), device=dev) + 1.0
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            st = s.cuda_stream
            def run():
                _lib.check(lib.dsrl_bn_train_fwd_from_stats(x.data_ptr(), C, y.data_ptr(), C, P, C, 1e-5, 0.1, mean.data_ptr(), inv.data_ptr(), rm.data_ptr(), rv.data_ptr(), gam.data_ptr(), bet.data_ptr(),
                                                            r.data_ptr() if r is not None else None, C, 1, 0.0, 0, 0, stats.data_ptr(), np_, None, st), 'bn')
            run(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(40):
                    run()
            g.replay(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s)
            for _ in range(5):
                g.replay()
            b.record(s); torch.cuda.synchronize()
            out.append(a.elapsed_time(b) / 200 * 1e3)
    print(f'P {P:6d} C {C:5d} residual {res}: {parts:4d} row blocks of partials {out[0]:6.2f} us per launch, 1 row block {out[1]:6.2f} us  -> merge prologue {out[0] - out[1]:5.2f} us', flush=True)
