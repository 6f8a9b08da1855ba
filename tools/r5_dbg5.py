import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dualsuperreslearningforsemseg_amd as D
from dualsuperreslearningforsemseg_amd import ddp, functional as HF
from dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume import SyntheticCityscapes, TrainStep
from dualsuperreslearningforsemseg_amd.datasets.Cityscapes import settings as cs
DEV = 'cuda:0'
size, B = (256, 512), int(sys.argv[1]) if len(sys.argv) > 1 else 4
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
OFF = {'DSRL_BNSTATS_FAST': '0', 'DSRL_WGRAD_BIG_CFG': '-1', 'DSRL_SK_AUTO': '0'}
def run(env, shared):
    for k, v in OFF.items():
        os.environ[k] = v
    os.environ.update(env)
    HF._query_cache.clear()
    HF.bn_bwd_stats_shared = shared
    flat.p_flat.copy_(p0); flat.b_flat.copy_(b0); flat.m_flat.zero_()
    HF.set_dropout_seed(4242)
    step = TrainStep(model, flat, 3, 0.1, 1.0, 255, graph=False)
    losses, _ = step(img, org, tgt, 0.0, 0.9, 0.0, True)
    torch.cuda.synchronize()
    return losses, flat.g_flat.clone().double()
ref = run({}, False)
ref2 = run({}, False)
print('B', B, 'reference repeat identical:', bool(torch.equal(ref[1], ref2[1])))
for name, env, shared in (('fast epilogue', {'DSRL_BNSTATS_FAST': '1'}, False), ('wgrad 128x128', {'DSRL_WGRAD_BIG_CFG': '0'}, False), ('sk auto', {'DSRL_SK_AUTO': '1'}, False),
                          ('shared', {}, True), ('shared + fast', {'DSRL_BNSTATS_FAST': '1'}, True), ('all', {'DSRL_BNSTATS_FAST': '1', 'DSRL_WGRAD_BIG_CFG': '0', 'DSRL_SK_AUTO': '1'}, True)):
    r = run(env, shared)
    r2 = run(env, shared)
    print('%-16s rel L2 vs reference %.3e  repeat identical %s  losses equal %s' % (name, float((r[1] - ref[1]).norm() / ref[1].norm()), bool(torch.equal(r[1], r2[1])), r[0] == ref[0]), flush=True)
for name, env in (('split plan off (other tiles, benign)', {'DSRL_SPLIT_PLAN': '0'}), ('big tiles off (benign)', {'DSRL_BIG_TILES': '0'}), ('bn fused off (benign)', {'DSRL_BN_FUSED': '0'})):
    r = run(env, False)
    for k in env:
        os.environ.pop(k)
    print('%-40s rel L2 vs reference %.3e  losses equal %s' % (name, float((r[1] - ref[1]).norm() / ref[1].norm()), r[0] == ref[0]), flush=True)
