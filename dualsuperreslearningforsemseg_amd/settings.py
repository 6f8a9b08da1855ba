"""Training defaults of the reference's settings.py:19-80 that the hot path reads (restated as data; the reference
module itself imports apex/torchvision and cannot be imported on the target)."""
import os

from .datasets import Cityscapes

SUPPORTED_DEVICES = ['cpu', 'gpu']
SUPPORTED_DISTRIBUTED_BACKENDS = ['gloo', 'mpi', 'nccl']       # 'nccl' is RCCL on ROCm
RANDOM_SEED = 54321                                            # settings.py:25
DEFAULT_BATCH_SIZE = 4
DEFAULT_LEARNING_RATE = 0.01
DEFAULT_END_LEARNING_RATE = 0.001
DEFAULT_MOMENTUM = 0.9
DEFAULT_WEIGHTS_DECAY = 0.0005
DEFAULT_POLY_POWER = 0.9
DEFAULT_LOSS_WEIGHTS = [0.1, 1.0]
WEIGHTS_ROOT_DIR = 'weights'
WEIGHTS_DIR = os.path.join(WEIGHTS_ROOT_DIR, 'stage{stage}')
FINAL_WEIGHTS_FILE = 'final.weights'
CHECKPOINTS_DIR = os.path.join(WEIGHTS_DIR, 'checkpoints')
CHECKPOINT_FILE = 'epoch{epoch}.checkpoint'
STAGES = [1, 2, 3]
MODEL_INPUT_SIZE = (256, 512)                                   # settings.py:62 (a parameter here, not a constant)
MODEL_OUTPUT_SIZE = tuple(x * 2 for x in MODEL_INPUT_SIZE)
DATASETS = {'cityscapes': {'path': os.path.join('datasets', 'Cityscapes', 'data'), 'splits': ['train', 'val', 'test'],
                           'settings': Cityscapes.settings}}
