#!/usr/bin/env python3
"""End-to-end sanity: N training steps on ONE fixed synthetic batch (dropout off so that runs are comparable), default fast
configuration vs the conservative one (exact-product fp32 convs, three-kernel BN, no shared gradient buffers, per-layer filter
transposes, no K groups).  The loss must fall and the two trajectories must track each other."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def run(conservative):
    for k in ('DSRL_BN_FUSED', 'DSRL_BATCHED_TRANSPOSE', 'DSRL_FORCE_KG', 'DSRL_WGRAD_KG', 'DSRL_SPLIT_PLAN'):
        os.environ.pop(k, None)
    from dualsuperreslearningforsemseg_amd import functional as HF
    if conservative:
        os.environ.update(DSRL_BN_FUSED='0', DSRL_BATCHED_TRANSPOSE='0', DSRL_FORCE_KG='1', DSRL_WGRAD_KG='0', DSRL_SPLIT_PLAN='0')
    HF.set_conv_precision('fp32' if conservative else None)
    HF.grad_slots_enabled = not conservative
    from dualsuperreslearningforsemseg_amd.models import DSRL
    from dualsuperreslearningforsemseg_amd.ddp import FlatParams
    from dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume import TrainStep, SyntheticCityscapes
    from dualsuperreslearningforsemseg_amd.datasets.Cityscapes import settings as cs
    dev = torch.device('cuda:0')
    torch.manual_seed(54321)
    model = DSRL(3, cs).to(dev).to(memory_format=torch.channels_last).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    flat = FlatParams(model)
    step = TrainStep(model, flat, 3, 0.1, 1.0, cs.IGNORE_CLASS_LABEL)
    (img, org), (tgt, _) = next(iter(SyntheticCityscapes(4, (128, 256), dev, rank=0, length=1)))
    hist = []
    for _ in range(steps):
        vals, _ = step(img, org, tgt, 0.006, 0.9, 5e-4, True)
        hist.append(vals)
    return hist


fast, slow = run(False), run(True)
for i in (0, 1, 2, 5, 10, 20, steps - 1):
    if i < steps:
        print(f'step {i:3d}  fast CE {fast[i][0]:.5f} MSE {fast[i][1]:.5f} FA {fast[i][2]:.5f} total {fast[i][3]:.5f} | conservative CE {slow[i][0]:.5f} total {slow[i][3]:.5f}')
assert fast[-1][3] < fast[0][3] - 0.1 and all(b[0] < a[0] + 1e-4 for a, b in zip(fast, fast[1:])), 'the loss does not fall steadily'    # CE monotone; the small FA term is not
rel = [abs(a[3] - b[3]) / abs(b[3]) for a, b in zip(fast, slow)]
print('max relative difference of the total loss: first 5 steps %.2e, all steps %.2e' % (max(rel[:5]), max(rel)))
assert max(rel[:5]) < 2e-3, 'trajectories diverge from the start'
print('OK')
