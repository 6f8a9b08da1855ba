#!/usr/bin/env python3
"""Which host ops issue device-to-device copies / small torch kernels during one training step (torch.profiler, grouped by stack)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualsuperreslearningforsemseg_amd.models import DSRL
from dualsuperreslearningforsemseg_amd.ddp import FlatParams
from dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume import TrainStep, SyntheticCityscapes
from dualsuperreslearningforsemseg_amd.datasets.Cityscapes import settings as cs
dev = torch.device('cuda:0')
model = DSRL(3, cs).to(dev).to(memory_format=torch.channels_last).train()
flat = FlatParams(model)
step = TrainStep(model, flat, 3, 0.1, 1.0, cs.IGNORE_CLASS_LABEL)
(img, org), (tgt, _) = next(iter(SyntheticCityscapes(8, (256, 512), dev, rank=0, length=1)))
for _ in range(3):
    step(img, org, tgt, 0.006, 0.9, 5e-4, True)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(img, org, tgt, 0.006, 0.9, 5e-4, True)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=6)
rows = [e for e in ka if e.key in ('aten::copy_', 'aten::add_', 'aten::add', 'aten::zero_', 'aten::fill_', 'aten::clone', 'aten::contiguous', 'aten::mul', 'aten::sum', 'aten::cat')]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    print(e.key, e.count, 'dev_us', round(e.device_time_total), '|', ' <- '.join(s.split('/')[-1] for s in e.stack[:5]))
