import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
import gen
from hip_helpers import dev, host, make_head, hip_losses, rel_err
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'head_small.npz'))
for mode in ('eval', 'train'):
    head, _ = make_head(gen.SMALL, 3, 101, mode == 'train')
    x16, x4, target, org = gen.make_head_inputs(202, 2, 2, 4, gen.SMALL)
    outs = head(dev(x16), dev(x4))
    L = hip_losses(outs, dev(target), dev(org), 3)
    L[3].backward()
    errs = {k: rel_err(host(p.grad), g[f'{mode}.grad.{k}']) for k, p in head.named_parameters()}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    print(mode, 'fwd', f"{rel_err(host(outs[0]), g[f'{mode}.SSSR']):.1e}", [(k[-32:], f'{v:.1e}') for k, v in worst], flush=True)
