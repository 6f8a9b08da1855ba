"""Shared helpers of the GPU parity tests."""
import numpy as np
import torch

import gen
import dualsuperreslearningforsemseg_amd as D
from dualsuperreslearningforsemseg_amd import functional as HF
from dualsuperreslearningforsemseg_amd.models.DSRL import DSRL
from dualsuperreslearningforsemseg_amd.models.modules.ASPP import ASPP
from dualsuperreslearningforsemseg_amd.nn_modules import HipBatchNorm2d, HipConv2d, HipReLU, HipSequential

DEV = 'cuda:0'


def dev(a, cl=True):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    if cl and t.dim() == 4 and t.dtype == torch.float32:
        t = t.contiguous(memory_format=torch.channels_last)
    return t


def host(t):
    return t.detach().float().cpu().contiguous().numpy()


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(np.isnan(a), np.isnan(b)), 'NaN pattern differs'
    d = np.nanmax(np.abs(a - b)) if a.size else 0.0
    return float(0.0 if np.isnan(d) else d) / max(float(np.nanmax(np.abs(b))) if b.size else 1.0, 1e-30)


def check(a, b, tol, name=''):
    e = rel_err(a, b)
    assert e <= tol, f'{name}: relative error {e:.3e} > {tol:.1e}'
    return e


def check_elementwise(a, b, tol, floor=1e-2, name=''):
    """Element-by-element relative error |a - b| / |b| over the elements with |b| above `floor` x the range of b (north_star's "1e-3 relative"
    read per element).  The floor is 1 % of the range: two fp32 evaluations of the same 2700-term dot products in different summation orders
    (the reference on ATen / oneDNN vs any other order) differ by ~5e-6 of the range, which IS 5e-3 relative on an element at 1e-3 of the
    range - below the floor the range-relative check() is the meaningful statement."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    m = np.abs(b) > floor * np.nanmax(np.abs(b))
    e = float(np.max(np.abs(a[m] - b[m]) / np.abs(b[m]))) if m.any() else 0.0
    assert e <= tol, f'{name}: element-wise relative error {e:.3e} > {tol:.1e} ({int(m.sum())} of {m.size} elements above the floor)'
    return e


class Head(torch.nn.Module):
    """The non-backbone part of DSRL at arbitrary widths, assembled from DSRL's own static constructors."""

    def __init__(self, widths, stage):
        super().__init__()
        w = widths
        self.stage = stage
        self.feature_extractor = torch.nn.ModuleDict({
            'aspp': ASPP(in_channels=w['c16'], out_channels=w['aspp'], rate=1),
            'shortcut_conv': HipSequential(HipConv2d(w['c4'], w['low'], kernel_size=1, padding=0, bias=False), HipBatchNorm2d(w['low']), HipReLU())})
        self.SSSR_decoder = DSRL._define_SSSR_decoder(w['aspp'], w['low'], w['mid'], gen.NUM_CLASSES)
        if stage > 1:
            self.SISR_decoder = DSRL._define_SISR_decoder(w['aspp'] + w['low'], 3, 8)
        if stage > 2:
            self.SSSR_feature_transformer = DSRL._define_feature_transformer(gen.NUM_CLASSES, 1)
            self.SISR_feature_transformer = DSRL._define_feature_transformer(3, 1)

    def forward(self, x16, x4):
        HF.begin_forward()
        return DSRL.forward_head(self, x16, x4)


def make_head(widths, stage, pseed, training, dropout=False):
    P = gen.make_head_params(pseed, widths, stage)
    head = Head(widths, stage)
    missing, unexpected = head.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()}, strict=False)
    assert not unexpected and all(k.endswith('num_batches_tracked') for k in missing), (missing, unexpected)
    head = head.to(DEV).to(memory_format=torch.channels_last)
    head.train(training)
    if not dropout:
        for m in head.modules():
            if isinstance(m, torch.nn.Dropout):
                m.eval()
    return head, P


def hip_losses(outs, target, org, stage, w1=0.1, w2=1.0):
    """train_or_resume.py:435-438 on the HIP loss kernels."""
    ce = HF.cross_entropy(outs[0], target, gen.IGNORE)
    ms = w1 * HF.mse_loss(outs[1], org) if stage > 1 else torch.zeros((), device=DEV)
    fa = w2 * D.FALoss()(outs[2], outs[3]) if stage > 2 else torch.zeros((), device=DEV)
    return ce, ms, fa, ce + ms + fa
