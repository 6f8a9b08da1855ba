import os, socket, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch, numpy as np
import torch.distributed as dist
import test_rccl_gpu as T
from dualsuperreslearningforsemseg_amd import functional as HF
hist0, flat0, step0 = T._make(1, False, 0.003)
p0 = flat0.p_flat.clone(); step0.release()
print('plain', hist0[-1])
hist1, flat1, step1 = T._make(1, True, 0.003)
print('plain graph', hist1[-1], float((flat1.p_flat - p0).norm() / p0.norm())); step1.release()
s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=torch.device('cuda:0'))
for graph, split in ((False, '1'), (True, '0'), (True, '1')):
    os.environ['DSRL_GRAPH_SPLIT'] = split
    hist, flat, step = T._make(2, graph, 0.006)
    worst = max(abs(a - b) / max(abs(a), 1e-6) for u, v in zip(hist0, hist) for a, b in zip(u, v))
    print(graph, split, 'worst', worst, hist[-1], 'param rel', float((flat.p_flat - p0).norm() / p0.norm()), flush=True)
    for i, (u, v) in enumerate(zip(hist0, hist)):
        print('   step', i, ['%.5f' % a for a in u], ['%.5f' % b for b in v])
    step.release()
dist.destroy_process_group()
