#!/usr/bin/env python3
"""Static inspection of the two captured graphs of the multi-rank step (1-rank RCCL group that claims world size 2, as tests/test_rccl_gpu.py does) with
the zero fills issued as hipMemsetAsync again (DSRL_ZERO_FILL_MEMSET=1: what round 3 replaced by zero_fill_kernel after a zeroing ran behind the
kernels that max into the zeroed words in ~40 % of the two-graph replays).  For every memset node of the DOT dumps: its predecessors and successors
in the captured dependency graph - does the memset have an edge to the kernel that consumes the zeroed words?
usage: DSRL_ZERO_FILL_MEMSET=1 python tools/graph_memset_edges.py [outdir]"""
import os
import re
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/graph_dot'
os.environ['DSRL_GRAPH_DEBUG_DUMP'] = out
os.environ.setdefault('DSRL_ZERO_FILL_MEMSET', '1')
os.environ.setdefault('DSRL_GRAPH_SPLIT', '1')            # the two-graph schedule (opt-in since round 4)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import dualsuperreslearningforsemseg_amd as D  # noqa: E402
from dualsuperreslearningforsemseg_amd import ddp, functional as HF  # noqa: E402
from dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume import SyntheticCityscapes, TrainStep  # noqa: E402
from dualsuperreslearningforsemseg_amd.datasets.Cityscapes import settings as cs  # noqa: E402

DEV = 'cuda:0'
s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=torch.device(DEV))
torch.manual_seed(54321)
model = D.DSRL(3, cs).to(DEV).to(memory_format=torch.channels_last).train()
real = ddp.dist.get_world_size
ddp.dist.get_world_size = lambda group=None: 2
try:
    flat = ddp.FlatParams(model, chunk_bytes=16 << 20)
finally:
    ddp.dist.get_world_size = real
step = TrainStep(model, flat, 3, 0.1, 1.0, cs.IGNORE_CLASS_LABEL, graph=True)
(img, org), (tgt, _) = next(iter(SyntheticCityscapes(2, (64, 128), torch.device(DEV), length=1)))
for _ in range(step.GRAPH_WARMUP + 2):
    step(img, org, tgt, 0.003, 0.9, 0.0, True)
torch.cuda.synchronize()
print('captured, replays:', step.graph_replays, 'split:', step.split)
for name in ('graph_a.dot', 'graph_b.dot'):
    path = os.path.join(out, name)
    if not os.path.isfile(path):
        print(name, 'missing'); continue
    txt = open(path).read()
    nodes = dict(re.findall(r'"?(\w+)"?\s*\[[^\]]*label="([^"]*)"', txt))
    edges = re.findall(r'"?(\w+)"?\s*->\s*"?(\w+)"?', txt)
    succ, pred = {}, {}
    for a, b in edges:
        succ.setdefault(a, []).append(b); pred.setdefault(b, []).append(a)
    short = lambda n: re.sub(r'\\[ln]', ' | ', nodes.get(n, n))[:110]          # noqa: E731
    ms = [n for n, lab in nodes.items() if 'MEMSET' in lab.upper() or 'memset' in lab]
    print(f'{name}: {len(nodes)} nodes, {len(edges)} edges, {len(ms)} memset nodes')
    for n in ms[:12]:
        print('  memset', short(n))
        print('     <-', [short(p) for p in pred.get(n, [])][:3])
        print('     ->', [short(q) for q in succ.get(n, [])][:3])
    no_succ = [n for n in ms if not succ.get(n)]
    print(f'  memset nodes without a successor edge: {len(no_succ)} of {len(ms)}')
dist.destroy_process_group()
