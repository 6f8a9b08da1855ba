"""ResNet-101 (output stride 16) - counterpart of the reference's models/modules/backbone/ResNet101.py:6-107.

The reference builds it from torchvision 0.8.1's `Bottleneck` (ResNet101.py:13,76), which is absent on the target;
the bottleneck below restates that block (1x1 -> 3x3(stride, dilation) -> 1x1 expansion 4, stride on the 3x3) with
torchvision's attribute names, so the state_dict keys equal torchvision's resnet101 and pretrained weights load.
Every convolution / BN / pooling runs on the HIP kernels.
"""
import os

import torch as t

from ....nn_modules import HipBatchNorm2d, HipConv2d, HipMaxPool2d, HipReLU, HipSequential
from .... import functional as HF


class Bottleneck(t.nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or HipBatchNorm2d
        width = int(planes * (base_width / 64.)) * groups
        self.conv1 = HipConv2d(inplanes, width, kernel_size=1, stride=1, bias=False)
        self.bn1 = norm_layer(width)
        self.conv2 = HipConv2d(width, width, kernel_size=3, stride=stride, padding=dilation, groups=groups, dilation=dilation, bias=False)
        self.bn2 = norm_layer(width)
        self.conv3 = HipConv2d(width, planes * self.expansion, kernel_size=1, stride=1, bias=False)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = HipReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # the block input feeds conv1 and the residual branch: both gradient contributions land in one buffer (HF.GradSlot) instead of
        # being summed by a separate elementwise kernel; fork() isolates the pair from any other user of x
        slot = None
        # bn3 of the previous block left a link on its output: the data gradient that completes our input's shared buffer carries its sums
        lin = getattr(x, '_dsrl_bnlink', None) if HF.bn_bwd_stats_enabled else None
        outer = getattr(x, '_dsrl_outer_slot', None)       # x is already a fork with a slot that a consumer outside this block shares (ResNet101.forward)
        if outer is not None and HF.grad_slots_enabled and t.is_grad_enabled():
            slot = outer
        elif HF.grad_slots_enabled and x.requires_grad and t.is_grad_enabled():
            x, slot = HF.fork(x), HF.GradSlot()
        if slot is None:
            lin = None
        elif lin is not None:
            slot.link = lin
        lds = None
        if self.downsample is None:
            identity = x
        elif len(self.downsample) == 2 and isinstance(self.downsample[0], HipConv2d) and self.downsample[0].bias is None:
            ds = self.downsample[0]
            # the downsample BatchNorm's output feeds only bn3's residual add: bn3's backward writes its output gradient and leaves its sums (HF.BNLink)
            lds = HF.BNLink() if (HF.bn_bwd_stats_enabled and HF.bn_res_stats_enabled and t.is_grad_enabled()) else None
            identity = HF.conv2d_bn_act(x, ds.weight, None, ds.stride[0], ds.padding[0], ds.dilation[0], self.downsample[1], grad_slot=slot, in_link=lin, out_link=lds)
        else:
            if slot is not None:
                slot.closed = True
            identity = self.downsample(x)
        c1, c2, c3 = self.conv1, self.conv2, self.conv3
        # bn1's output feeds only conv2 and bn2's only conv3: their data-gradient kernels leave the BN-backward sums behind (HF.BNLink)
        l1 = l2 = l3 = None
        if HF.bn_bwd_stats_enabled and t.is_grad_enabled():
            l1, l2, l3 = HF.BNLink(), HF.BNLink(), HF.BNLink(shared=True)
        out = HF.conv2d_bn_act(x, c1.weight, None, c1.stride[0], c1.padding[0], c1.dilation[0], self.bn1, relu=True, grad_slot=slot, out_link=l1, in_link=lin)
        out = HF.conv2d_bn_act(out, c2.weight, None, c2.stride[0], c2.padding[0], c2.dilation[0], self.bn2, relu=True, in_link=l1, out_link=l2)
        out = HF.conv2d_bn_act(out, c3.weight, None, c3.stride[0], c3.padding[0], c3.dilation[0], self.bn3, relu=True, residual=identity,   # bn3 + identity, ReLU
                               residual_grad_slot=slot if self.downsample is None else None, in_link=l2, out_link=l3, res_link=lds)
        if l3 is not None and l3.valid and HF.bn_bwd_stats_shared:
            out._dsrl_bnlink = l3       # picked up by the next block, whose conv1 / downsample conv complete this tensor's gradient
        return out


def _cut(x, cut):
    d = x.detach().requires_grad_(True)
    a = HF.carried_amax(x)
    if a is not None:
        HF.set_amax(d, a)               # same values: the detached leaf keeps the operand-magnitude record of the tensor (functional.amax_for)
    cut.append((x, d))
    return d


class ResNet101(t.nn.Module):
    PRETRAINED_WEIGHTS_URL = "https://download.pytorch.org/models/resnet101-5d3b4d8f.pth"
    PRETRAINED_WEIGHTS_FILE = 'resnet101_pretrained.pth'

    def __init__(self, groups=1, width_per_group=64, replace_stride_with_dilation=None, init_weights=True, BatchNorm2d=HipBatchNorm2d):
        super().__init__()
        layers = [3, 4, 23, 3]
        self._norm_layer = BatchNorm2d
        self.inplanes = 64
        self.dilation = 1
        if replace_stride_with_dilation is None:
            replace_stride_with_dilation = [False, False, False]
        assert len(replace_stride_with_dilation) == 3, \
            "replace_stride_with_dilation should be None or a 3-element tuple, got {}".format(replace_stride_with_dilation)
        self.groups = groups
        self.base_width = width_per_group
        self.conv1 = HipConv2d(3, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(self.inplanes)
        self.relu = HipReLU(inplace=True)
        self.maxpool = HipMaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2, dilate=replace_stride_with_dilation[0])
        self.layer3 = self._make_layer(256, layers[2], stride=2, dilate=replace_stride_with_dilation[1])
        self.layer4 = self._make_layer(512, layers[3], stride=2, dilate=replace_stride_with_dilation[2])
        if init_weights:
            self._init_weights(BatchNorm2d)

    @t.no_grad()
    def _init_weights(self, BatchNorm2d):
        # ResNet101.py:45-55: kaiming for convs, BN (1,0), zero-init of every bottleneck's last BN
        for m in self.modules():
            if isinstance(m, t.nn.Conv2d):
                t.nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, BatchNorm2d):
                m.weight.fill_(1.0)
                m.bias.zero_()
        for m in self.modules():
            if isinstance(m, Bottleneck):
                t.nn.init.constant_(m.bn3.weight, 0)

    def initialize_with_pretrained_weights(self, weights_dir, map_location=t.device('cpu')):
        """ResNet101.py:58-65 downloads the torchvision checkpoint; the target has no network, so the file must already
        sit at <weights_dir>/resnet101_pretrained.pth (the name the reference caches it under)."""
        path = os.path.join(weights_dir, self.PRETRAINED_WEIGHTS_FILE)
        if not os.path.isfile(path):
            raise FileNotFoundError(f"'{path}' not found; fetch {self.PRETRAINED_WEIGHTS_URL} to that path (no network access from here)")
        state = t.load(path, map_location=map_location)
        missing_keys, _ = self.load_state_dict(state, strict=False)
        assert len(missing_keys) == 0, "BUG CHECK: Pretrained weights from model zoo for ResNet101 has missing keys: {}.".format(missing_keys)

    def _make_layer(self, planes, blocks, stride=1, dilate=False):
        norm_layer = self._norm_layer
        downsample = None
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        if stride != 1 or self.inplanes != planes * Bottleneck.expansion:
            downsample = HipSequential(HipConv2d(self.inplanes, planes * Bottleneck.expansion, kernel_size=1, stride=stride, bias=False),
                                       norm_layer(planes * Bottleneck.expansion))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample, self.groups, self.base_width, previous_dilation, norm_layer)]
        self.inplanes = planes * Bottleneck.expansion
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes, groups=self.groups, base_width=self.base_width, dilation=self.dilation,
                                     norm_layer=norm_layer))
        return t.nn.Sequential(*layers)

    def forward(self, x: t.Tensor):
        x = HF.batch_norm_act(self.conv1(x), self.bn1, relu=True)        # ResNet101.py:92-94
        x = self.maxpool(x)
        x = self.layer1(x)
        # layer1's output has three consumers: conv1 and the downsample conv of layer2's first block, and the decoder's shortcut conv (DSRL.py:164).
        # One fork + slot for all three (the block adopts it, DSRL.forward_head hands it to the shortcut conv): their data gradients land in one
        # buffer instead of two tensors and an add pass over 67 MB.  Not with the two-phase backward (the tensor is cut into a leaf there).
        if (HF.grad_slots_enabled and HF.outer_grad_slot and t.is_grad_enabled() and x.requires_grad and getattr(self, '_dsrl_cut', None) is None
                and isinstance(self.layer2[0], Bottleneck)):
            link = getattr(x, '_dsrl_bnlink', None)
            x = HF.fork(x)
            x._dsrl_outer_slot = HF.GradSlot()
            if link is not None:
                x._dsrl_bnlink = link
        low_level_features = x                                            # ResNet101.py:98
        x = self.layer2(x)
        x = self.layer3(x)
        # Two-phase backward (command_handlers/train_or_resume.TrainStep with more than one rank): `_dsrl_cut` is a list the training step
        # hangs on this module; the two tensors that leave layers 1-3 are handed on as detached leaves and recorded, so that
        # total.backward() stops there (head, ASPP and layer4: the first ~54 % of the gradient arena in backward order, whose all-reduce
        # then runs beside the second phase) and torch.autograd.backward(originals, leaf gradients) finishes layers 3..1 and the stem.
        cut = getattr(self, '_dsrl_cut', None)
        if cut is not None and t.is_grad_enabled() and x.requires_grad:
            x, low_level_features = _cut(x, cut), _cut(low_level_features, cut)
        x = self.layer4(x)
        return x, low_level_features
