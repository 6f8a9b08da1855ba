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
# settings.py:76-80: every key main.py:51-52 / inspect_checkpoint read back from a .checkpoint file
VARIABLES_IN_CHECKPOINT = \
    ['device', 'mixed_precision', 'amp_state_dict', 'disable_cudnn_benchmark', 'num_workers', 'val_interval', 'checkpoint_interval', 'checkpoint_history',
     'init_weights', 'batch_size', 'epochs', 'learning_rate', 'end_learning_rate', 'momentum', 'weights_decay', 'poly_power', 'stage', 'w1', 'w2',
     'freeze_batch_norm', 'experiment_id', 'description', 'early_stopping', 'CE_train_avg_loss', 'MSE_train_avg_loss', 'FA_train_avg_loss',
     'Avg_train_loss', 'CE_val_avg_loss', 'MSE_val_avg_loss', 'FA_val_avg_loss', 'Avg_val_loss', 'epoch', 'best_validation_dict', 'model_state_dict',
     'optimizer_state_dict', 'amp_state_dict']
# `mixed_precision` (apex opt levels in the reference, train_or_resume.py:68-72) selects the arithmetic of the MFMA conv kernels here
# O0 = fp32 behaviour ('f16x3': fp32-equivalent products); O1 / O2 / O3 = the reference's fp16 tensor-core arithmetic (one fp16 MMA per product, fp32
# accumulation: 'f16x1'; the per-tensor operand scales of the kernels stand in for apex's loss scaling) - never slower than O0
MIXED_PRECISION_TO_CONV_ARITHMETIC = {None: None, '': None, 'O0': 'f16x3', 'O1': 'f16x1', 'O2': 'f16x1', 'O3': 'f16x1'}
STAGES = [1, 2, 3]
MODEL_INPUT_SIZE = (256, 512)                                   # settings.py:62 (a parameter here, not a constant)
MODEL_OUTPUT_SIZE = tuple(x * 2 for x in MODEL_INPUT_SIZE)
DATASETS = {'cityscapes': {'path': os.path.join('datasets', 'Cityscapes', 'data'), 'splits': ['train', 'val', 'test'],
                           'settings': Cityscapes.settings}}
