import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
    import numpy as np
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + '.npz'))
        return cache[name]
    return load


@pytest.fixture(autouse=True)
def _unbind_device_rng():
    """A graph-mode TrainStep binds the device-resident dropout key (dsrl_rng_bind_device_key); parity tests pass explicit seeds, so
    every test starts and ends with the key unbound."""
    yield
    if _has_gpu():
        from dualsuperreslearningforsemseg_amd import _lib, functional as HF
        _lib.call('dsrl_rng_bind_device_key', None)
        HF.DeviceRng._active.clear()
        HF.wgrad_queue = None
