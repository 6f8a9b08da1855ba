"""Feature-affinity loss on the fused HIP kernel - same surface as the reference's models/losses/FALoss.py:5-34."""
import torch as t

from ... import functional as HF


class FALoss(t.nn.modules.loss._Loss):
    __constants__ = ['reduction']

    def __init__(self, subsample_factor: int = 8, size_average=None, reduce=None, reduction='mean') -> None:
        super().__init__(size_average=None, reduce=None, reduction=reduction)
        self.subsample_factor = subsample_factor

    def forward(self, feature_map1: t.Tensor, feature_map2: t.Tensor) -> t.Tensor:
        # the reference's two BUG CHECK asserts, FALoss.py:19-20
        assert len(feature_map1.shape) == 4, "BUG CHECK: Feature map inputs to FALoss.forward() must have 4 dimensions (B, C, H, W)."
        assert feature_map1.shape == feature_map2.shape, "BUG CHECK: Feature map inputs to FALoss.forward() should be of same size."
        return HF.fa_loss(feature_map1, feature_map2, self.subsample_factor, self.reduction)
