"""Counterpart of the reference's models/BaseModel.py:4-6."""
from abc import abstractmethod

import torch as t


class BaseModel(t.nn.Module):
    @abstractmethod
    def initialize_with_pretrained_weights(self, weights_dir, map_location=t.device('cpu')):
        raise NotImplementedError
