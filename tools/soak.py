#!/usr/bin/env python3
"""Soak: many training steps of the bench configuration; step time and allocator statistics per block of 50 steps."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualsuperreslearningforsemseg_amd import functional as HF
from dualsuperreslearningforsemseg_amd.models import DSRL
from dualsuperreslearningforsemseg_amd.ddp import FlatParams
from dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume import TrainStep, SyntheticCityscapes
from dualsuperreslearningforsemseg_amd.datasets.Cityscapes import settings as cs
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device('cuda:0')
torch.manual_seed(54321)
model = DSRL(3, cs).to(dev).to(memory_format=torch.channels_last).train()
flat = FlatParams(model)
step = TrainStep(model, flat, 3, 0.1, 1.0, cs.IGNORE_CLASS_LABEL)
(img, org), (tgt, _) = next(iter(SyntheticCityscapes(8, (256, 512), dev, rank=0, length=1)))
last = None
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(steps):
    step.enqueue(img, org, tgt, 0.006, 0.9, 5e-4, True)
    while step.pending() > 1:
        last = step.collect()
    if (i + 1) % 50 == 0:
        while step.pending():
            last = step.collect()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f'steps {i - 48:4d}-{i + 1:4d}: {(t1 - t0) / 50 * 1e3:6.2f} ms/step  allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB  '
              f'reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB  total loss {last[3]:.4f}', flush=True)
        t0 = time.perf_counter()
print('fused-BN barrier timeouts:', HF.bn_fused_barrier_timeouts())
