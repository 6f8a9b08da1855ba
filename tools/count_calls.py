#!/usr/bin/env python3
"""C-ABI calls of one eager stage-3 training step by entry point (DSRL_CONV_PRECISION from the environment): how many operand magnitudes
of the f16x3 arithmetic come from producers and how many from dsrl_amax launches of their own."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dualsuperreslearningforsemseg_amd as D
from dualsuperreslearningforsemseg_amd import functional as HF, settings
from dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume import SyntheticCityscapes, TrainStep
from dualsuperreslearningforsemseg_amd.datasets.Cityscapes import settings as cs
from dualsuperreslearningforsemseg_amd.ddp import FlatParams
dev = torch.device('cuda', 0)
torch.manual_seed(settings.RANDOM_SEED)
model = D.DSRL(3, cs).to(dev).to(memory_format=torch.channels_last).train()
flat = FlatParams(model)
step = TrainStep(model, flat, 3, 0.1, 1.0, cs.IGNORE_CLASS_LABEL, graph=False)
(img, org), (tgt, _) = next(iter(SyntheticCityscapes(8, (256, 512), dev, rank=0, length=1)))
for _ in range(2):
    step.enqueue(img, org, tgt, 0.006, 0.9, 5e-4, True); step.collect()
counts = collections.Counter()
orig = HF.call
def counting(name, *a):
    counts[name] += 1
    return orig(name, *a)
HF.call = counting
step.enqueue(img, org, tgt, 0.006, 0.9, 5e-4, True); vals = step.collect()
HF.call = orig
print('mode', HF.get_conv_precision(), 'losses', [round(float(v), 5) for v in vals])
for k, v in sorted(counts.items(), key=lambda kv: -kv[1]):
    print(f'{v:5d} {k}')
