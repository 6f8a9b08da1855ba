#!/usr/bin/env python3
"""Why a conv launch takes longer inside the step than launched back to back: the same launch with (a) warm operands, (b) operands rotated over more
buffers than the Infinity Cache holds (cold data, warm code), (c) warm operands with two other kernels launched in between (code evicted?)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dualsuperreslearningforsemseg_amd import _lib, functional as HF
lib = _lib.load()
SH = {'l3_3x3': (8, 256, 16, 32, 256, 3, 1, 1, 1), 'l3_1x1_dn': (8, 1024, 16, 32, 256, 1, 1, 0, 1), 'l3_1x1_up': (8, 256, 16, 32, 1024, 1, 1, 0, 1)}
dev = 'cuda:0'
NB = 48
st = torch.cuda.current_stream().cuda_stream
for name, (N, C, H, W, K, R, stride, pad, dil) in SH.items():
    shp = (N, H, W, C, K, R, R, stride, pad, dil)
    ws = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
    sets = []
    for b in range(NB):
        x = torch.randn(N * H * W * C, device=dev); w = torch.randn(K * R * R * C, device=dev) * 0.05; y = torch.empty(N * H * W * K, device=dev)
        rec, sp, tsp, tr = HF.split_filter(w.view(K, R, R, C).permute(0, 3, 1, 2))
        xs = HF.amax_slot(x.device); HF.call('dsrl_amax', x.data_ptr(), C, N * H * W, C, xs.data_ptr(), st)
        sets.append((x, w, y, rec, sp, xs))
    big = torch.empty(600 << 20, dtype=torch.uint8, device=dev)
    def launch(s):
        x, w, y, rec, sp, xs = s
        _lib.check(lib.dsrl_conv2d_fwd_amax(x.data_ptr(), C, xs.data_ptr(), w.data_ptr(), rec.data_ptr(), sp.data_ptr(), None, y.data_ptr(), K, *shp, ws.data_ptr(), ws.numel(), None, 0, st), 'fwd')
    def timed(fn, reps=96):
        # per-launch events so that what runs between the launches is not counted
        evs = []
        for i in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn(i, a, b); evs.append((a, b))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs[8:])
        return ts[len(ts) // 2] * 1e3
    def warm(i, a, b):
        a.record(); launch(sets[0]); b.record()
    def cold(i, a, b):
        a.record(); launch(sets[i % NB]); b.record()
    small = torch.randn(1 << 16, device=dev)
    def code(i, a, b):
        small.mul_(1.0001); torch.relu_(small)             # two other kernels between the conv launches
        a.record(); launch(sets[0]); b.record()
    def flushed(i, a, b):
        big.zero_()                                         # 600 MB written: nothing of the operands is left in L2 / Infinity Cache
        a.record(); launch(sets[0]); b.record()
    def cold_w(i, a, b):
        s0, si = sets[0], sets[i % NB]
        a.record(); launch((s0[0], si[1], s0[2], si[3], si[4], s0[5])); b.record()
    def cold_x(i, a, b):
        s0, si = sets[0], sets[i % NB]
        a.record(); launch((si[0], s0[1], s0[2], s0[3], s0[4], si[5])); b.record()
    touch = [s_[4].view(torch.int32)[::32] for s_ in sets]          # one word per 128-byte line of the pre-split filter
    acc = torch.zeros((), dtype=torch.int64, device=dev)
    def cold_w_prefetched(i, a, b):
        s0, si = sets[0], sets[i % NB]
        acc.add_(touch[i % NB].sum())                       # a small kernel reads every line of the filter first: Infinity-Cache resident at the launch
        a.record(); launch((s0[0], si[1], s0[2], si[3], si[4], s0[5])); b.record()
    print(f'{name:10s} rotating filters only {timed(cold_w):.1f} us, rotating activations only {timed(cold_x):.1f}, rotating filters touched by a small kernel just before {timed(cold_w_prefetched):.1f}', flush=True)
    print(f'{name:10s} (event pair around each launch, median) warm {timed(warm):.1f} us, rotating {NB} operand sets {timed(cold):.1f}, other kernels in between {timed(code):.1f}, '
          f'caches flushed by a 600 MB fill {timed(flushed, 40):.1f}', flush=True)
    del sets
