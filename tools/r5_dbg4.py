import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dualsuperreslearningforsemseg_amd as D
from dualsuperreslearningforsemseg_amd import ddp, functional as HF
from dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume import SyntheticCityscapes, TrainStep
from dualsuperreslearningforsemseg_amd.datasets.Cityscapes import settings as cs
DEV = 'cuda:0'
for size, B in (((64, 128), 2), ((256, 512), 4)):
    torch.manual_seed(54321)
    model = D.DSRL(3, cs)
    with torch.no_grad():
        for m in model.modules():
            if hasattr(m, 'bn3'):
                m.bn3.weight.fill_(0.5)
    model = model.to(DEV).to(memory_format=torch.channels_last).train()
    flat = ddp.FlatParams(model)
    (img, org), (tgt, _) = next(iter(SyntheticCityscapes(B, size, torch.device(DEV), length=1)))
    p0, b0 = flat.p_flat.clone(), flat.b_flat.clone()
    out = {}
    for shared in (False, True, False):
        HF.bn_bwd_stats_shared = shared
        flat.p_flat.copy_(p0); flat.b_flat.copy_(b0); flat.m_flat.zero_()
        HF.set_dropout_seed(4242)
        step = TrainStep(model, flat, 3, 0.1, 1.0, 255, graph=False)
        losses, _ = step(img, org, tgt, 0.0, 0.9, 0.0, True)
        torch.cuda.synchronize()
        g = flat.g_flat.clone()
        if shared in out:
            print(size, 'repeat identical:', bool(torch.equal(out[shared][1], g)))
        out[shared] = (losses, g)
    a, b = out[False][1].double(), out[True][1].double()
    print(size, 'losses', out[False][0], out[True][0])
    print(size, 'gradient arena shared vs not: rel L2 %.3e, max abs diff %.3e (max abs %.3e)' % (float((a - b).norm() / a.norm()), float((a - b).abs().max()), float(a.abs().max())))
    # per parameter
    worst = []
    for p_, o in zip(flat.params, flat.offsets):
        n = p_.numel(); x, y = a[o:o + n], b[o:o + n]
        if float(x.norm()) > 0:
            worst.append((float((x - y).norm() / x.norm()), n))
    worst.sort(reverse=True)
    print('   worst per-parameter rel L2:', ['%.2e (%d)' % w for w in worst[:6]])
