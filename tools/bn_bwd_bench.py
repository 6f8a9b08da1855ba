#!/usr/bin/env python3
"""BatchNorm backward of a bottleneck's bn3 (P = 4096 x C = 1024, residual + ReLU) launched alone: barrier kernel with 128 / 256 blocks, three-kernel path;
operands rotated over more sets than the Infinity Cache holds unless WARM=1."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualsuperreslearningforsemseg_amd import _lib
lib = _lib.load()
dev = 'cuda:0'; st = torch.cuda.current_stream().cuda_stream
for P, C in ((4096, 1024), (4096, 256), (16384, 512)):
    NB = 1 if os.environ.get('WARM') else 12
    sets = [[torch.randn(P * C, device=dev) for _ in range(3)] + [torch.empty(P * C, device=dev) for _ in range(2)] for _ in range(NB)]
    mean, inv, gam = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.ones(C, device=dev)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    ws = torch.empty(int(lib.dsrl_bn_workspace_bytes(P, C)), dtype=torch.uint8, device=dev)
    def run(i):
        x, y, dy, dx, dr = sets[i % NB]
        _lib.check(lib.dsrl_bn_bwd(x.data_ptr(), C, y.data_ptr(), C, dy.data_ptr(), C, dx.data_ptr(), C, dr.data_ptr(), C, P, C, mean.data_ptr(), inv.data_ptr(), gam.data_ptr(),
                                   dg.data_ptr(), db.data_ptr(), 1, 0.0, 1, ws.data_ptr(), ws.numel(), None, st), 'bn_bwd')
    def timed(reps=60):
        evs = []
        for i in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(i); b.record(); evs.append((a, b))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs[6:])
        return ts[len(ts) // 2] * 1e3
    out = []
    for label, mb in (('barrier kernel, budget 128', 128), ('budget 256', 256), ('three-kernel path', 0)):
        lib.dsrl_bn_fused_max_blocks(mb)
        out.append(f'{label} {timed():.1f} us')
    lib.dsrl_bn_fused_max_blocks(-1)
    print(f'P {P} C {C} ({5 * P * C * 4 / 1e6:.0f} MB moved): ' + ', '.join(out), flush=True)
