#!/usr/bin/env python3
"""Which ATen kernels (copies, fills) does a stage-3 step still launch, and from where?  One eager step under torch.profiler with stacks."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dualsuperreslearningforsemseg_amd as D
from dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume import SyntheticCityscapes, TrainStep
from dualsuperreslearningforsemseg_amd.datasets.Cityscapes import settings as cs
from dualsuperreslearningforsemseg_amd.ddp import FlatParams
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda', 0)
model = D.DSRL(3, cs).to(dev).to(memory_format=torch.channels_last).train()
flat = FlatParams(model)
step = TrainStep(model, flat, 3, 0.1, 1.0, cs.IGNORE_CLASS_LABEL, graph=False)
(img, org), (tgt, _) = next(iter(SyntheticCityscapes(8, (256, 512), dev, rank=0, length=1)))
for it in range(2):
    step.enqueue(img, org, tgt, 0.006, 0.9, 5e-4, True); step.collect()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step.enqueue(img, org, tgt, 0.006, 0.9, 5e-4, True); step.collect()
for ev in prof.events():
    if ev.name in ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::clone', 'aten::contiguous', 'aten::cat', 'aten::add_', 'aten::mul', 'aten::add', 'aten::sum') or ev.name.startswith('aten::_to'):
        st = [s for s in ev.stack if 'dualsuperres' in s or 'bench' in s][:3]
        shapes = getattr(ev, 'input_shapes', None)
        print(f'{ev.name:18s} dev {ev.device_time_total:8.1f} us  {" < ".join(s.split("/")[-1] for s in st)}')
