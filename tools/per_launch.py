#!/usr/bin/env python3
"""Every conv launch of one eager training step with its shape, time and algorithmic TFLOP/s (DSRL_PROF_DUMP: the library appends one line per
event-bracketed launch when the profile is read), sorted by time: where the conv time of the step goes, layer by layer."""
import os, sys, collections
dump = '/tmp/dsrl_prof_dump.txt'
if os.path.exists(dump):
    os.remove(dump)
os.environ['DSRL_PROF_DUMP'] = dump
os.environ['DSRL_GRAPH'] = '0'
os.environ['DSRL_WGRAD_GROUP'] = os.environ.get('DSRL_WGRAD_GROUP', '0')      # per-layer weight gradients: the grouped launch is one record
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dualsuperreslearningforsemseg_amd as D
from dualsuperreslearningforsemseg_amd import functional as HF, settings, _lib
from dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume import SyntheticCityscapes, TrainStep
from dualsuperreslearningforsemseg_amd.datasets.Cityscapes import settings as cs
from dualsuperreslearningforsemseg_amd.ddp import FlatParams
dev = torch.device('cuda', 0)
torch.manual_seed(settings.RANDOM_SEED)
model = D.DSRL(3, cs).to(dev).to(memory_format=torch.channels_last).train()
flat = FlatParams(model)
step = TrainStep(model, flat, 3, 0.1, 1.0, cs.IGNORE_CLASS_LABEL, graph=False)
(img, org), (tgt, _) = next(iter(SyntheticCityscapes(8, (256, 512), dev, length=1)))
lib = _lib.load()
def run(n):
    for _ in range(n):
        step.enqueue(img, org, tgt, 0.006, 0.9, 5e-4, True)
        while step.pending():
            step.collect()
run(3); torch.cuda.synchronize()
lib.dsrl_prof_enable(1)
run(1); torch.cuda.synchronize()
import ctypes
for fam in range(16):
    n = ctypes.c_int64(); ms = ctypes.c_double(); fl = ctypes.c_double()
    lib.dsrl_prof_read(fam, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl))
lib.dsrl_prof_enable(0)
rows = []
for l in open(dump):
    fam, ms, fl, by, tag = l.rstrip('\n').split('\t')
    rows.append((float(ms), float(fl), tag or f'family {fam}'))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for ms, fl, tag in rows:
    a = agg[tag]; a[0] += 1; a[1] += ms; a[2] += fl
tot = sum(v[1] for v in agg.values())
print(f'{len(rows)} bracketed launches, {tot:.2f} ms')
for tag, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('TOP', 60))]:
    print(f'{ms * 1e3:8.1f} us  x{n:<3d} {fl / ms / 1e9:7.1f} TF  {tag}')
