#!/usr/bin/env python3
"""cProfile of the host side of a training step on a tiny input (the GPU is never the bottleneck): where the launch path spends its time."""
import cProfile, os, pstats, sys
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
(img, org), (tgt, _) = next(iter(SyntheticCityscapes(2, (64, 128), dev, rank=0, length=1)))
for _ in range(5):
    step(img, org, tgt, 0.006, 0.9, 5e-4, True)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step.enqueue(img, org, tgt, 0.006, 0.9, 5e-4, True)
    while step.pending() > 1:
        step.collect()
while step.pending():
    step.collect()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
