"""The DSRL graph assembled from STOCK torch.nn CPU modules - TEST INFRASTRUCTURE (see oracle/__init__.py).

This is what the reference's `--device cpu` path executes (utils.py:259-260 -> ATen/oneDNN CPU kernels): the same layer
graph as models/DSRL.py:11-186 + models/modules/ASPP.py:5-44 + a ResNet-101 (output stride 16) of torchvision-style
bottlenecks (models/modules/backbone/ResNet101.py:6-107), the loss mix of command_handlers/train_or_resume.py:435-438 with
the all-pairs FA loss of models/losses/FALoss.py:8-34, backward and torch.optim.SGD (:63-66).  Sub-module names equal the
product's, so its state_dict loads here unchanged.  Used only by bench.py's `cpu_baseline` leg (timed) and by
tests/test_oracle_vs_golden.py (checked against the reference-generated head goldens); it never runs in the product.
"""
import time

import torch
from torch import nn
import torch.nn.functional as F


class Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + identity)


class ResNet101(nn.Module):
    def __init__(self):
        super().__init__()
        self.inplanes, self.dilation = 64, 1
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, 3)
        self.layer2 = self._make_layer(128, 4, stride=2)
        self.layer3 = self._make_layer(256, 23, stride=2)
        self.layer4 = self._make_layer(512, 3, stride=2, dilate=True)       # replace_stride_with_dilation=[False, False, True]

    def _make_layer(self, planes, blocks, stride=1, dilate=False):
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample, previous_dilation)]
        self.inplanes = planes * 4
        layers += [Bottleneck(self.inplanes, planes, dilation=self.dilation) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        low = x = self.layer1(x)
        return self.layer4(self.layer3(self.layer2(x))), low


def _cbr(cin, cout, k, pad=0, dil=1):
    return nn.Sequential(nn.Conv2d(cin, cout, k, padding=pad, dilation=dil, bias=False), nn.BatchNorm2d(cout), nn.ReLU())


class ASPP(nn.Module):
    def __init__(self, in_channels=2048, out_channels=256, rate=1):
        super().__init__()
        self.branches = nn.ModuleList([
            _cbr(in_channels, out_channels, 1),
            _cbr(in_channels, out_channels, 3, 6 * rate, 6 * rate), _cbr(in_channels, out_channels, 3, 12 * rate, 12 * rate),
            _cbr(in_channels, out_channels, 3, 18 * rate, 18 * rate),
            nn.Sequential(nn.Conv2d(in_channels, out_channels, 1, bias=False), nn.BatchNorm2d(out_channels), nn.ReLU()),
            _cbr(out_channels * 5, out_channels, 1)])
        self.pool = nn.AdaptiveAvgPool2d(1)

    def forward(self, x):
        h, w = x.shape[-2:]
        outs = [self.branches[i](x) for i in range(4)]
        g = F.interpolate(self.branches[4](self.pool(x)), size=(h, w), mode='bilinear', align_corners=True)
        return self.branches[5](torch.cat(outs + [g], dim=1))


class TorchCpuDSRL(nn.Module):
    def __init__(self, stage=3, num_classes=19):
        super().__init__()
        self.stage = stage
        self.feature_extractor = nn.ModuleDict({'backbone': ResNet101(), 'aspp': ASPP(), 'shortcut_conv': _cbr(256, 48, 1)})
        nc = num_classes
        self.SSSR_decoder = nn.ModuleDict({
            'cat_conv': nn.Sequential(nn.Conv2d(304, 256, 3, padding=1, bias=False), nn.BatchNorm2d(256), nn.ReLU(), nn.Dropout(0.2),
                                      nn.Conv2d(256, 256, 3, padding=1, bias=False), nn.BatchNorm2d(256), nn.ReLU(), nn.Dropout(0.2)),
            'cls_conv': nn.Conv2d(256, nc, 1),
            'upsample16_pred': nn.Sequential(nn.UpsamplingBilinear2d(scale_factor=2.0), nn.Dropout(0.2),
                                             nn.ConvTranspose2d(nc, nc, 2, stride=2, bias=False), nn.BatchNorm2d(nc), nn.ReLU(), nn.Dropout(0.2),
                                             nn.ConvTranspose2d(nc, nc, 2, stride=2, bias=True))})
        if stage > 1:
            self.SISR_decoder = nn.Sequential(nn.Conv2d(304, 3 * 64, 3, padding=1), nn.PixelShuffle(8))
        if stage > 2:
            self.SSSR_feature_transformer = nn.Sequential(nn.Conv2d(nc, 1, 1, stride=8, bias=False), nn.BatchNorm2d(1), nn.ReLU())
            self.SISR_feature_transformer = nn.Sequential(nn.Conv2d(3, 1, 1, stride=8, bias=False), nn.BatchNorm2d(1), nn.ReLU())

    def forward_head(self, backbone_features, low):
        fe = self.feature_extractor
        a = fe['aspp'](backbone_features)
        a = F.interpolate(a, scale_factor=4, mode='bilinear', align_corners=True)
        cat = torch.cat([a, fe['shortcut_conv'](low)], dim=1)
        sssr = self.SSSR_decoder['upsample16_pred'](self.SSSR_decoder['cls_conv'](self.SSSR_decoder['cat_conv'](cat)))
        sisr = sssr_ft = sisr_ft = None
        if self.stage > 1:
            sisr = self.SISR_decoder(cat)
        if self.stage > 2:
            sssr_ft, sisr_ft = self.SSSR_feature_transformer(sssr), self.SISR_feature_transformer(sisr)
        return sssr, sisr, sssr_ft, sisr_ft

    def forward(self, x):
        return self.forward_head(*self.feature_extractor['backbone'](x))


def fa_loss(fm1, fm2, k=8):
    """FALoss.py:8-34 with stock torch ops: spectral-norm normalised X^T X per (b, c) slice, all-pairs L1 mean."""
    def sim(x):
        x = x / torch.linalg.matrix_norm(x, ord=2, dim=(-2, -1), keepdim=True)
        return x.transpose(-2, -1) @ x
    s1, s2 = sim(F.avg_pool2d(fm1, k)).flatten(2), sim(F.avg_pool2d(fm2, k)).flatten(2)
    n = s1.shape[-1]
    return F.l1_loss(s1.repeat_interleave(n, dim=-1), s2.repeat(1, 1, n))


def total_loss(outs, target, input_org, stage=3, w1=0.1, w2=1.0, ignore_index=255):
    sssr, sisr, a, b = outs
    ce = F.cross_entropy(sssr, target.long(), ignore_index=ignore_index)
    ms = w1 * F.mse_loss(sisr, input_org) if stage > 1 else sssr.new_zeros(())
    fa = w2 * fa_loss(a, b) if stage > 2 else sssr.new_zeros(())
    return ce, ms, fa, ce + ms + fa


def time_train_step(state_dict, batch=2, height=256, width=512, stage=3, threads=None, repeats=1, budget_s=12.0):
    """One warm-up + up to `repeats` timed stage-`stage` training steps (forward, losses, backward, SGD) of the stock-torch CPU
    graph with the given weights; returns (images/s of the best step, threads, seconds of that step, steps timed, median seconds).  The warm-up
    is skipped (and the first step reported) when it alone exceeds the budget."""
    if threads:
        torch.set_num_threads(int(threads))
    model = TorchCpuDSRL(stage).train()
    own = set(model.state_dict().keys())           # a lower stage owns a subset of a stage-3 model's tensors (DSRL.py:172-184)
    missing, unexpected = model.load_state_dict({k: v.detach().float().cpu() for k, v in state_dict.items() if k in own}, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    opt = torch.optim.SGD(model.parameters(), lr=0.006, momentum=0.9, weight_decay=5e-4)
    g = torch.Generator().manual_seed(1234)
    org = torch.randn((batch, 3, 2 * height, 2 * width), generator=g)
    x = F.interpolate(org, size=(height, width), mode='bilinear', align_corners=True)
    tgt = torch.randint(0, 19, (batch, 2 * height, 2 * width), generator=g, dtype=torch.uint8)
    tgt[torch.rand(tgt.shape, generator=g) < 0.1] = 255
    times, t_start = [], time.perf_counter()
    for i in range(1 + repeats):
        t0 = time.perf_counter()
        opt.zero_grad()
        total_loss(model(x), tgt, org, stage)[3].backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s:
            break
    timed = times[1:] if len(times) > 1 else times
    best = min(timed)
    return batch / best, torch.get_num_threads(), best, len(timed), sorted(timed)[len(timed) // 2]
