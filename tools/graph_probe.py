"""Eager vs hipGraph-replayed training step: ms per step and loss trajectories (both start from the same weights / dropout key)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dualsuperreslearningforsemseg_amd as D
from dualsuperreslearningforsemseg_amd import functional as HF, settings
from dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume import SyntheticCityscapes, TrainStep
from dualsuperreslearningforsemseg_amd.datasets.Cityscapes import settings as cs
from dualsuperreslearningforsemseg_amd.ddp import FlatParams

dev = torch.device('cuda', 0)
B, H, W = int(os.environ.get('B', 8)), int(os.environ.get('H', 256)), int(os.environ.get('W', 512))
steps = int(os.environ.get('STEPS', 30))


def make(graph):
    torch.manual_seed(settings.RANDOM_SEED)
    model = D.DSRL(3, cs)
    with torch.no_grad():
        for m in model.modules():
            if hasattr(m, 'bn3'):
                m.bn3.weight.fill_(0.5)
    model = model.to(dev).to(memory_format=torch.channels_last).train()
    flat = FlatParams(model)
    HF.set_dropout_seed(99)
    return model, flat, TrainStep(model, flat, 3, 0.1, 1.0, cs.IGNORE_CLASS_LABEL, graph=graph)


data = SyntheticCityscapes(B, (H, W), dev, length=1)
(img, org), (tgt, _) = next(iter(data))
res = {}
for graph in ((True,) if os.environ.get('GRAPH_ONLY') else (False, True)):
    model, flat, step = make(graph)
    hist = []
    def run(n):
        for _ in range(n):
            step.enqueue(img, org, tgt, 0.006, 0.9, 5e-4, True)
            while step.pending() > 1:
                hist.append(step.collect())
        while step.pending():
            hist.append(step.collect())
    run(5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[graph] = hist
    print(f'graph={graph}: {1e3 * dt / steps:.2f} ms/step, {B * steps / dt:.1f} img/s, host enqueue {1e3 * step.host_enqueue_s:.2f} ms, replays {step.graph_replays}', flush=True)
    print('  losses first/last', [round(v, 5) for v in hist[0]], [round(v, 5) for v in hist[-1]], flush=True)
    step.release()
    del model, flat, step
    torch.cuda.empty_cache()
if False not in res: sys.exit(0)
a, b = res[False], res[True]
worst = max(abs(x - y) / max(abs(x), 1e-9) for u, v in zip(a, b) for x, y in zip(u, v))
print(f'max relative difference of the loss trajectories eager vs graph: {worst:.3e}')
