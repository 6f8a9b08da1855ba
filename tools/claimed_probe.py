#!/usr/bin/env python3
"""Which parameters of a stage-3 step receive their gradient through the arena's sink protocol (claimed: the producing kernel overwrites the slot) and
which through autograd's accumulation (needs a zeroed slot)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dualsuperreslearningforsemseg_amd as D
from dualsuperreslearningforsemseg_amd import functional as HF
from dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume import SyntheticCityscapes, TrainStep
from dualsuperreslearningforsemseg_amd.datasets.Cityscapes import settings as cs
from dualsuperreslearningforsemseg_amd.ddp import FlatParams
dev = torch.device('cuda', 0)
model = D.DSRL(3, cs).to(dev).to(memory_format=torch.channels_last).train()
flat = FlatParams(model)
step = TrainStep(model, flat, 3, 0.1, 1.0, cs.IGNORE_CLASS_LABEL, graph=False)
(img, org), (tgt, _) = next(iter(SyntheticCityscapes(8, (256, 512), dev, rank=0, length=1)))
names = {id(p): n for n, p in model.named_parameters()}
for it in range(2):
    step.enqueue(img, org, tgt, 0.006, 0.9, 5e-4, True); step.collect()
    un = [names[id(p)] for i, p in enumerate(flat.params) if i not in flat._claimed]
    print(f'step {it}: {len(flat.params)} parameters, {len(flat._claimed)} claimed, {len(un)} not:', un[:12])
