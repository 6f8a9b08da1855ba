import os, socket, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch, numpy as np
import torch.distributed as dist
import test_rccl_gpu as T
from dualsuperreslearningforsemseg_amd import functional as HF
from dualsuperreslearningforsemseg_amd.command_handlers.train_or_resume import TrainStep
s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=torch.device('cuda:0'))
os.environ['DSRL_GRAPH_SPLIT'] = '1'
res = {}
for warm in (2, 100):
    TrainStep.GRAPH_WARMUP = warm
    hist, flat, step = T._make(2, True, 0.006)
    res[warm] = (hist, flat.p_flat.clone())
    print('warmup', warm, 'replays', step.graph_replays, flush=True)
    for i, v in enumerate(hist):
        print('   step', i, ['%.6f' % b for b in v])
    step.release()
print('replayed == eager:', res[2][0] == res[100][0], bool(torch.equal(res[2][1], res[100][1])))
dist.destroy_process_group()
