#!/usr/bin/env python3
"""Which host-side calls produce the device-to-device copy / fill / elementwise launches of a training step: torch profiler over eager
steps, copy-like device activity grouped by the Python frames that issued it."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import dualsuperreslearningforsemseg_amd as D
from dualsuperreslearningforsemseg_amd import functional as HF, settings
from dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume import SyntheticCityscapes, TrainStep
from dualsuperreslearningforsemseg_amd.datasets.Cityscapes import settings as cs
from dualsuperreslearningforsemseg_amd.ddp import FlatParams

dev = torch.device('cuda', 0)
torch.manual_seed(settings.RANDOM_SEED)
model = D.DSRL(3, cs).to(dev).to(memory_format=torch.channels_last).train()
flat = FlatParams(model)
HF.set_dropout_seed(99)
step = TrainStep(model, flat, 3, 0.1, 1.0, cs.IGNORE_CLASS_LABEL, graph=False)
(img, org), (tgt, _) = next(iter(SyntheticCityscapes(8, (256, 512), dev, length=1)))


def run(n):
    for _ in range(n):
        step.enqueue(img, org, tgt, 0.006, 0.9, 5e-4, True)
        while step.pending():
            step.collect()


run(3)
torch.cuda.synchronize()
N = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    run(N)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    name = ev.name
    if not any(k in name for k in ('copy_', 'aten::fill_', 'aten::zero_', 'aten::add', 'aten::mul', 'aten::cat', 'aten::clone', 'aten::contiguous', 'aten::sum', 'aten::div')):
        continue
    if ev.device_time_total <= 0 and not getattr(ev, 'cuda_time_total', 0):
        continue
    frames = [f for f in (ev.stack or []) if 'dualsuperres' in f or 'bench' in f][:2]
    key = (name, str(getattr(ev, 'input_shapes', ''))[:60], ' <- '.join(f.split('/')[-1] for f in frames))
    a = agg[key]; a[0] += 1; a[1] += (ev.device_time_total or 0.0)
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = 0.0
for (name, shp, where), (cnt, us) in rows[:45]:
    tot += us
    print(f'{us / N:8.1f} us/step {cnt / N:6.1f} calls  {name:18s} {shp:60s} {where}')
print('listed total per step: %.1f us' % (tot / N))
