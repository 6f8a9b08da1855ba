#!/usr/bin/env python3
"""The N > 1 training path against real RCCL kernels on a one-GPU box: a 1-rank 'nccl' process group, with ddp.FlatParams told that
the world size is 2 (so gradient hooks, chunked asynchronous all-reduces on RCCL's stream, buffer broadcasts, the 1/world factor
and the 128-block fused-BN budget are all active).  A 1-rank all-reduce is an identity, so the losses must match a plain run whose
learning rate is halved (the SGD kernel divides the gradient by the claimed world size)."""
import os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29531')
from dualsuperreslearningforsemseg_amd import functional as HF
from dualsuperreslearningforsemseg_amd import ddp
from dualsuperreslearningforsemseg_amd.models import DSRL
from dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume import TrainStep, SyntheticCityscapes
from dualsuperreslearningforsemseg_amd.datasets.Cityscapes import settings as cs

dev = torch.device('cuda:0')
torch.cuda.set_device(dev)


def run(distributed, steps=int(os.environ.get('REH_STEPS', '44'))):
    torch.manual_seed(54321)
    model = DSRL(3, cs).to(dev).to(memory_format=torch.channels_last).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    if distributed:
        real = dist.get_world_size
        ddp.dist.get_world_size = lambda group=None: 2          # claim two ranks: the collectives still run on the 1-rank group
        try:
            flat = ddp.FlatParams(model, broadcast_buffers=os.environ.get('REH_NO_BCAST') is None)
        finally:
            ddp.dist.get_world_size = real
        assert flat.world == 2 and len(flat._hooks) > 0
    else:
        flat = ddp.FlatParams(model)
    step = TrainStep(model, flat, 3, 0.1, 1.0, cs.IGNORE_CLASS_LABEL)
    (img, org), (tgt, _) = next(iter(SyntheticCityscapes(8, (256, 512), dev, rank=0, length=1)))
    hist = []
    torch.cuda.synchronize(); t0 = None
    for i in range(steps):
        if i == 4:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        step.enqueue(img, org, tgt, 0.006 if distributed else 0.003, 0.9, 5e-4, True)
        while step.pending() > 1:
            hist.append(step.collect())
    while step.pending():
        hist.append(step.collect())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / (steps - 4) * 1e3
    return hist, ms


plain, ms_plain = run(False)
dist.init_process_group('nccl', rank=0, world_size=1)
rccl, ms_rccl = run(True)
for i in (0, 3, 7, 11, len(plain) - 1):
    print(f'step {i}: plain total {plain[i][3]:.5f} | with RCCL reduction {rccl[i][3]:.5f}')
rel = max(abs(a[3] - b[3]) / abs(a[3]) for a, b in zip(plain, rccl))
print(f'max relative difference {rel:.2e}; ms/step plain {ms_plain:.1f}, with RCCL {ms_rccl:.1f}; fused-BN barrier timeouts {HF.bn_fused_barrier_timeouts()}')
assert rel < 2e-3 and HF.bn_fused_barrier_timeouts() == 0
dist.destroy_process_group()
print('OK')
