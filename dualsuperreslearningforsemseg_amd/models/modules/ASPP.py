"""Atrous spatial pyramid pooling on the HIP kernels - same surface as the reference's models/modules/ASPP.py:4-44
(constructor arguments, `branches` ModuleList of Conv/BN/ReLU triples, `avg`, state_dict keys)."""
import torch as t

from ... import functional as HF
from ...nn_modules import HipAdaptiveAvgPool2d, HipBatchNorm2d, HipConv2d, HipReLU, HipSequential


class ASPP(t.nn.Module):
    def __init__(self, in_channels: int, out_channels: int, rate: int = 1, init_weights=True, BatchNorm2d=HipBatchNorm2d):
        super().__init__()
        # (kernel, padding, dilation) of the six branches, ASPP.py:8-16
        cfg = [(in_channels, 1, 0, 1 * rate), (in_channels, 3, 6 * rate, 6 * rate), (in_channels, 3, 12 * rate, 12 * rate),
               (in_channels, 3, 18 * rate, 18 * rate), (in_channels, 1, 0, 1), (5 * out_channels, 1, 0, 1)]
        self.branches = t.nn.ModuleList()
        for cin, k, pad, dil in cfg:
            self.branches.append(HipSequential(HipConv2d(cin, out_channels, kernel_size=k, padding=pad, dilation=dil, bias=False),
                                               BatchNorm2d(num_features=out_channels), HipReLU()))
        self.avg = HipAdaptiveAvgPool2d(output_size=(1, 1))
        if init_weights:
            self._init_weights(BatchNorm2d)

    @t.no_grad()
    def _init_weights(self, BatchNorm2d):
        # ASPP.py:27-34: kaiming_normal(fan_out, relu) for convs, BN gamma=1 beta=0
        for m in self.modules():
            if isinstance(m, t.nn.Conv2d):
                t.nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, BatchNorm2d):
                m.weight.fill_(1.0)
                m.bias.zero_()

    def forward(self, x: t.Tensor):
        # the four conv branches read the same features: their data gradients accumulate in one buffer (HF.GradSlot behind a fork)
        xs, slot = x, None
        if HF.grad_slots_enabled and x.requires_grad and t.is_grad_enabled() and all(isinstance(self.branches[i], HipSequential) for i in range(4)):
            xs, slot = HF.fork(x), HF.GradSlot()
        outs = [self.branches[i](xs, grad_slot=slot) if slot is not None else self.branches[i](xs) for i in range(4)]     # ASPP.py:37
        # the pooled branch reads the same features: last consumer in the forward = first in the backward, its gradient becomes the shared buffer
        g = self.branches[4](HF.global_avg_pool(xs, slot) if slot is not None else self.avg(x))        # ASPP.py:38-39 (train-mode BN needs batch >= 2)
        outs.append(HF.upsample_bilinear_ac(g, x.shape[-2:]))                  # ASPP.py:40 (1x1 -> HxW, align_corners)
        return self.branches[5](HF.cat_channels(outs))                         # ASPP.py:44
