#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REFERENCE's own modules (imported from /root/reference)
on seeded synthetic inputs.  Runs only in the build container; the reference never travels to the GPU box,
only the vectors written here do.

Harness shims: `torch.Assert = torch._assert` (FALoss.py:19-20 uses the torch-1.7 name) and an
empty stub `torchvision` module (DSRL.py:2 / ResNet101.py:2 import it at module import time).  The head is
built from the reference's static constructors (DSRL._define_SSSR_decoder / _define_SISR_decoder /
_define_feature_transformer, ASPP(...)) and DSRL.forward lines 162-184 are replayed on synthetic backbone
features, because constructing DSRL(...) needs torchvision's Bottleneck which is absent here.

    python tests/golden/make_golden.py            # writes the .npz files next to this script
"""
import os
import sys
import types

import numpy as np
import torch as t

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen  # noqa: E402

REF = '/root/reference'
t.Assert = t._assert
sys.path.insert(0, REF)
sys.modules.setdefault('torchvision', types.ModuleType('torchvision'))
from models.DSRL import DSRL            # noqa: E402
from models.modules.ASPP import ASPP    # noqa: E402
from models.losses import FALoss        # noqa: E402

t.manual_seed(0)
t.set_num_threads(8)


class RefHead(t.nn.Module):
    """Everything of DSRL except the backbone, assembled from the reference's own constructors."""

    def __init__(self, w, stage):
        super().__init__()
        self.stage = stage
        self.feature_extractor = t.nn.ModuleDict({
            'aspp': ASPP(in_channels=w['c16'], out_channels=w['aspp'], rate=1),                 # DSRL.py:18
            'shortcut_conv': t.nn.Sequential(t.nn.Conv2d(w['c4'], w['low'], kernel_size=1, padding=0, bias=False),
                                             t.nn.BatchNorm2d(w['low']), t.nn.ReLU())})        # DSRL.py:19-25
        self.SSSR_decoder = DSRL._define_SSSR_decoder(w['aspp'], w['low'], w['mid'], gen.NUM_CLASSES)
        if stage > 1:
            self.SISR_decoder = DSRL._define_SISR_decoder(w['aspp'] + w['low'], 3, 8)
        if stage > 2:
            self.SSSR_feature_transformer = DSRL._define_feature_transformer(gen.NUM_CLASSES, 1)
            self.SISR_feature_transformer = DSRL._define_feature_transformer(3, 1)

    def forward(self, backbone_features, lowlevel_features):     # DSRL.py:162-184 verbatim in structure
        aspp_features = self.feature_extractor['aspp'](backbone_features)
        aspp_features = t.nn.UpsamplingBilinear2d(scale_factor=4.0)(aspp_features)
        lowlevel_features = self.feature_extractor['shortcut_conv'](lowlevel_features)
        cat_features = t.cat([aspp_features, lowlevel_features], dim=1)
        SSSR_output = self.SSSR_decoder['cat_conv'](cat_features)
        SSSR_output = self.SSSR_decoder['cls_conv'](SSSR_output)
        SSSR_output = self.SSSR_decoder['upsample16_pred'](SSSR_output)
        SISR_output = SSSR_t = SISR_t = None
        if self.stage > 1:
            SISR_output = self.SISR_decoder(cat_features)
            if self.stage > 2:
                SSSR_t = self.SSSR_feature_transformer(SSSR_output)
                SISR_t = self.SISR_feature_transformer(SISR_output)
        return SSSR_output, SISR_output, SSSR_t, SISR_t


def load_params(head, P):
    sd = {k: t.from_numpy(v.copy()) for k, v in P.items()}
    missing, unexpected = head.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith('num_batches_tracked') for k in missing), missing


def set_mode(head, bn_training):
    """train(): BN uses batch statistics; Dropout modules are individually put in eval (the same trick the
    reference uses to freeze BN, train_or_resume.py:379-382) because torch's RNG stream is not reproducible."""
    head.train(bn_training)
    for m in head.modules():
        if isinstance(m, t.nn.Dropout):
            m.eval()


def losses(head, outs, target, input_org, w1=0.1, w2=1.0):
    """train_or_resume.py:116-119, 435-438."""
    CE = t.nn.CrossEntropyLoss(ignore_index=gen.IGNORE)(outs[0], t.from_numpy(target).long())
    MSE = w1 * t.nn.MSELoss()(outs[1], t.from_numpy(input_org)) if head.stage > 1 else t.tensor(0.)
    FA = w2 * FALoss()(outs[2], outs[3]) if head.stage > 2 else t.tensor(0.)
    return CE, MSE, FA, CE + MSE + FA


def np_(x):
    return x.detach().cpu().numpy().copy()      # copy: never alias live parameter memory


# ----------------------------------------------------------------------------------------------
def golden_fa():
    out = {}
    rs = np.random.RandomState(11)
    cases = {
        'rand': (rs.uniform(size=(2, 1, 64, 128)), rs.uniform(size=(2, 1, 64, 128))),
        'c3': (rs.uniform(size=(2, 3, 32, 64)), rs.uniform(size=(2, 3, 32, 64))),
        'big': (rs.uniform(size=(1, 1, 128, 256)), rs.uniform(size=(1, 1, 128, 256))),     # 512x1024-input map, n=1024
        'signed': (rs.standard_normal((2, 1, 64, 128)), rs.standard_normal((2, 1, 64, 128))),
        'relu_like': (np.maximum(rs.standard_normal((2, 1, 64, 128)), 0), np.maximum(rs.standard_normal((2, 1, 64, 128)) + 0.3, 0)),
    }
    a = rs.uniform(size=(2, 1, 64, 128))
    cases['same'] = (a, a.copy())
    z = rs.uniform(size=(2, 1, 64, 128)); z[1] = 0.0
    cases['zero_sample'] = (z, rs.uniform(size=(2, 1, 64, 128)))
    # near-degenerate top singular values: pooled map = sigma*(u1 v1^T) + (sigma*(1-1e-3))*(u2 v2^T) + small
    q, _ = np.linalg.qr(rs.standard_normal((8, 8))); r, _ = np.linalg.qr(rs.standard_normal((16, 16)))
    sv = np.array([1.0, 0.999, 0.3, 0.2, 0.1, 0.05, 0.02, 0.01])
    pooled = (q * sv) @ r[:8]
    deg = np.repeat(np.repeat(pooled, 8, axis=0), 8, axis=1)[None, None]
    cases['near_degenerate'] = (np.concatenate([deg, deg[:, :, ::-1].copy()]), rs.uniform(size=(2, 1, 64, 128)))
    for name, (f1, f2) in cases.items():
        f1 = f1.astype(np.float32); f2 = f2.astype(np.float32)
        x1 = t.from_numpy(f1).requires_grad_(True); x2 = t.from_numpy(f2).requires_grad_(True)
        out[f'{name}.fm1'] = f1; out[f'{name}.fm2'] = f2
        for red in ('mean', 'sum'):
            out[f'{name}.{red}'] = np_(FALoss(reduction=red)(x1, x2))
        none = np_(FALoss(reduction='none')(x1, x2))
        out[f'{name}.none_shape'] = np.array(none.shape)
        out[f'{name}.none_sample'] = gen.strided_sample(none)
        loss = FALoss()(x1, x2)
        if name != 'zero_sample':
            loss.backward()
            out[f'{name}.g1'] = np_(x1.grad); out[f'{name}.g2'] = np_(x2.grad)
    np.savez_compressed(os.path.join(HERE, 'fa_loss.npz'), **out)
    print('fa_loss.npz', {k: float(v) for k, v in out.items() if k.endswith('.mean')})


def golden_ops():
    out = {}
    rs = np.random.RandomState(21)
    F = t.nn.functional

    def rec(name, **kw):
        for k, v in kw.items():
            out[f'{name}.{k}'] = v if isinstance(v, np.ndarray) else np_(v)

    # dilated 3x3 convs of ASPP.py:11-13 on a 16x32 map (reduced channels), + 1x1, + 3x3 pad1, + strided
    for name, (cin, cout, k, stride, pad, dil, h, w, bias) in {
        'conv_d6': (32, 16, 3, 1, 6, 6, 16, 32, False), 'conv_d12': (32, 16, 3, 1, 12, 12, 16, 32, False),
        'conv_d18': (32, 16, 3, 1, 18, 18, 16, 32, False), 'conv_1x1': (64, 19, 1, 1, 0, 1, 8, 16, True),
        'conv_3x3': (40, 24, 3, 1, 1, 1, 8, 16, True), 'conv_s8': (19, 1, 1, 8, 0, 1, 64, 64, False),
        'conv_s2': (16, 32, 3, 2, 1, 1, 16, 32, False), 'conv_7x7s2': (3, 8, 7, 2, 3, 1, 32, 64, False),
        'conv_1x1s2': (16, 32, 1, 2, 0, 1, 16, 32, False), 'conv_d2': (16, 16, 3, 1, 2, 2, 8, 16, False),
    }.items():
        x = t.from_numpy(rs.standard_normal((2, cin, h, w)).astype(np.float32)).requires_grad_(True)
        wt = t.from_numpy((rs.standard_normal((cout, cin, k, k)) * 0.1).astype(np.float32)).requires_grad_(True)
        b = t.from_numpy(rs.standard_normal(cout).astype(np.float32)).requires_grad_(True) if bias else None
        y = F.conv2d(x, wt, b, stride=stride, padding=pad, dilation=dil)
        dy = t.from_numpy(rs.standard_normal(tuple(y.shape)).astype(np.float32))
        y.backward(dy)
        rec(name, x=x, w=wt, y=y, dy=dy, dx=x.grad, dw=wt.grad, cfg=np.array([stride, pad, dil]))
        if bias:
            rec(name, b=b, db=b.grad)

    # ConvTranspose2d k2 s2 (DSRL.py:55-60,64-69)
    m = t.nn.ConvTranspose2d(19, 19, kernel_size=2, stride=2, padding=0, bias=True)
    x = t.from_numpy(rs.standard_normal((2, 19, 8, 16)).astype(np.float32)).requires_grad_(True)
    y = m(x); dy = t.from_numpy(rs.standard_normal(tuple(y.shape)).astype(np.float32)); y.backward(dy)
    rec('convT', x=x, w=m.weight, b=m.bias, y=y, dy=dy, dx=x.grad, dw=m.weight.grad, db=m.bias.grad)

    # bilinear align_corners (DSRL.py:53,163; ASPP.py:41)
    for name, (shape, size) in {'up2': ((2, 19, 8, 16), (16, 32)), 'up4': ((2, 8, 4, 8), (16, 32)),
                                'up_bcast': ((2, 8, 1, 1), (4, 8)), 'up_odd': ((1, 3, 5, 7), (13, 9))}.items():
        x = t.from_numpy(rs.standard_normal(shape).astype(np.float32)).requires_grad_(True)
        y = F.interpolate(x, size=size, mode='bilinear', align_corners=True)
        dy = t.from_numpy(rs.standard_normal(tuple(y.shape)).astype(np.float32)); y.backward(dy)
        rec(name, x=x, y=y, dy=dy, dx=x.grad)
    x = t.from_numpy(rs.standard_normal((1, 4, 6, 10)).astype(np.float32))
    rec('up2_module', x=x, y=t.nn.UpsamplingBilinear2d(scale_factor=2.0)(x))

    # PixelShuffle(8) (DSRL.py:84)
    x = t.from_numpy(rs.standard_normal((2, 192, 4, 6)).astype(np.float32)).requires_grad_(True)
    y = t.nn.PixelShuffle(8)(x); dy = t.from_numpy(rs.standard_normal(tuple(y.shape)).astype(np.float32)); y.backward(dy)
    rec('pixel_shuffle', x=x, y=y, dy=dy, dx=x.grad)

    # BatchNorm2d train / eval
    for mode in ('train', 'eval'):
        bn = t.nn.BatchNorm2d(19)
        with t.no_grad():
            bn.weight.copy_(t.from_numpy(rs.uniform(0.5, 1.5, 19).astype(np.float32)))
            bn.bias.copy_(t.from_numpy(rs.standard_normal(19).astype(np.float32)))
            bn.running_mean.copy_(t.from_numpy(rs.standard_normal(19).astype(np.float32) * 0.1))
            bn.running_var.copy_(t.from_numpy(rs.uniform(0.5, 1.5, 19).astype(np.float32)))
        rec(f'bn_{mode}', gamma=bn.weight, beta=bn.bias, rm0=bn.running_mean.clone(), rv0=bn.running_var.clone())
        bn.train(mode == 'train')
        x = t.from_numpy((rs.standard_normal((2, 19, 8, 16)) * 2 + 1).astype(np.float32)).requires_grad_(True)
        y = bn(x); dy = t.from_numpy(rs.standard_normal(tuple(y.shape)).astype(np.float32)); y.backward(dy)
        rec(f'bn_{mode}', x=x, y=y, dy=dy, dx=x.grad, dgamma=bn.weight.grad, dbeta=bn.bias.grad, rm1=bn.running_mean, rv1=bn.running_var)

    # pools
    x = t.from_numpy(rs.standard_normal((2, 5, 16, 32)).astype(np.float32)).requires_grad_(True)
    y = t.nn.AdaptiveAvgPool2d((1, 1))(x); dy = t.from_numpy(rs.standard_normal(tuple(y.shape)).astype(np.float32)); y.backward(dy)
    rec('gap', x=x, y=y, dy=dy, dx=x.grad)
    x = t.from_numpy(rs.standard_normal((2, 5, 16, 32)).astype(np.float32)).requires_grad_(True)
    y = t.nn.MaxPool2d(kernel_size=3, stride=2, padding=1)(x); dy = t.from_numpy(rs.standard_normal(tuple(y.shape)).astype(np.float32)); y.backward(dy)
    rec('maxpool', x=x, y=y, dy=dy, dx=x.grad)
    x = t.from_numpy(rs.standard_normal((2, 1, 64, 128)).astype(np.float32))
    rec('avgpool8', x=x, y=t.nn.AvgPool2d(8)(x))

    # CE with ignore_index, MSE (train_or_resume.py:116-117)
    lg = t.from_numpy((rs.standard_normal((2, 19, 16, 32)) * 3).astype(np.float32)).requires_grad_(True)
    tg = rs.randint(0, 19, (2, 16, 32)).astype(np.uint8); tg[rs.uniform(size=tg.shape) < 0.15] = 255
    ce = t.nn.CrossEntropyLoss(ignore_index=255)(lg, t.from_numpy(tg).long()); ce.backward()
    rec('ce', logits=lg, target=tg, loss=ce, dlogits=lg.grad)
    a = t.from_numpy(rs.standard_normal((2, 3, 16, 32)).astype(np.float32)).requires_grad_(True)
    b = t.from_numpy(rs.standard_normal((2, 3, 16, 32)).astype(np.float32))
    ms = t.nn.MSELoss()(a, b); ms.backward()
    rec('mse', a=a, b=b, loss=ms, da=a.grad)

    # SGD(momentum .9, wd 5e-4) two steps (train_or_resume.py:63-66)
    p = t.nn.Parameter(t.from_numpy(rs.standard_normal(1000).astype(np.float32)))
    opt = t.optim.SGD([p], lr=0.006, momentum=0.9, weight_decay=5e-4)
    rec('sgd', p0=p.detach().clone())
    for step in range(2):
        g = t.from_numpy(rs.standard_normal(1000).astype(np.float32))
        p.grad = g.clone(); opt.step()
        rec('sgd', **{f'g{step}': g, f'p{step + 1}': p.detach().clone()})
    np.savez_compressed(os.path.join(HERE, 'ops_micro.npz'), **out)
    print('ops_micro.npz', len(out), 'arrays')


def run_head(widths, stage, pseed, iseed, batch, h16, w16, bn_training, want_grads):
    P = gen.make_head_params(pseed, widths, stage)
    x16, x4, target, org = gen.make_head_inputs(iseed, batch, h16, w16, widths)
    head = RefHead(widths, stage)
    load_params(head, P)
    set_mode(head, bn_training)
    a = t.from_numpy(x16).requires_grad_(want_grads); b = t.from_numpy(x4).requires_grad_(want_grads)
    with t.set_grad_enabled(want_grads):
        outs = head(a, b)
        L = losses(head, outs, target, org)
    grads = None
    if want_grads:
        L[3].backward()
        grads = {k: np_(p.grad) for k, p in head.named_parameters()}
        grads['backbone_features'] = np_(a.grad); grads['lowlevel_features'] = np_(b.grad)
    return head, outs, L, grads


def golden_head_small():
    out = {}
    W = gen.SMALL
    for mode in ('eval', 'train'):
        head, outs, L, grads = run_head(W, 3, 101, 202, 2, 2, 4, mode == 'train', True)
        for n, o in zip(('SSSR', 'SISR', 'SSSR_ft', 'SISR_ft'), outs):
            out[f'{mode}.{n}'] = np_(o)
        out[f'{mode}.losses'] = np.array([float(x.detach()) for x in L], dtype=np.float64)
        for k, g in grads.items():
            out[f'{mode}.grad.{k}'] = g
        if mode == 'train':
            for k, v in head.state_dict().items():
                if 'running_' in k:
                    out[f'train.new.{k}'] = np_(v)
    # stage 1 / stage 2 gating (DSRL.py:172-184)
    for stage in (1, 2):
        head, outs, L, grads = run_head(W, stage, 101, 202, 2, 2, 4, True, True)
        out[f'stage{stage}.SSSR'] = np_(outs[0])
        out[f'stage{stage}.losses'] = np.array([float(x.detach()) for x in L], dtype=np.float64)
        out[f'stage{stage}.grad.cls_w'] = grads['SSSR_decoder.cls_conv.weight']
    np.savez_compressed(os.path.join(HERE, 'head_small.npz'), **out)
    print('head_small.npz', out['eval.losses'], out['train.losses'])


SMALL_GRAD_KEYS = ('cls_conv', 'upsample16_pred', 'feature_transformer', '.1.weight', '.1.bias', '.5.weight', '.5.bias',
                   'SISR_decoder.0.bias')


def golden_head_fullwidth():
    """Full channel widths (2048/256/48/304) at a small spatial size: features 4x8 and 16x32 (64x128 input)."""
    out = {}
    for mode in ('eval', 'train'):
        head, outs, L, grads = run_head(gen.FULL, 3, 303, 404, 2, 4, 8, mode == 'train', True)
        out[f'{mode}.SSSR_sample'] = gen.strided_sample(np_(outs[0]), 65536)
        out[f'{mode}.SSSR_argmax'] = np_(outs[0]).argmax(axis=1).astype(np.uint8)
        out[f'{mode}.SISR_sample'] = gen.strided_sample(np_(outs[1]), 16384)
        out[f'{mode}.SSSR_ft'] = np_(outs[2]); out[f'{mode}.SISR_ft'] = np_(outs[3])
        out[f'{mode}.losses'] = np.array([float(x.detach()) for x in L], dtype=np.float64)
        for k, g in grads.items():
            if any(s in k for s in SMALL_GRAD_KEYS):
                out[f'{mode}.grad.{k}'] = g
            else:
                out[f'{mode}.gradsum.{k}'] = gen.checksum(g)
                out[f'{mode}.gradsample.{k}'] = gen.strided_sample(g, 4096)
    np.savez_compressed(os.path.join(HERE, 'head_fullwidth.npz'), **out)
    print('head_fullwidth.npz', out['eval.losses'], out['train.losses'])


def golden_head_256x512():
    """BASELINE config size (256x512 input -> 512x1024 logits), eval forward, B=2: argmax map, top-2 margins,
    strided logits sample, checksums, loss values."""
    out = {}
    head, outs, L, _ = run_head(gen.FULL, 3, 505, 606, 2, 16, 32, False, False)
    sssr = np_(outs[0])
    out['SSSR_argmax'] = sssr.argmax(axis=1).astype(np.uint8)
    top2 = np.sort(sssr, axis=1)[:, -2:]
    margin = (top2[:, 1] - top2[:, 0])
    out['margin_q'] = np.quantile(margin, [0, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 0.5]).astype(np.float64)
    out['margin_u8'] = np.minimum(margin / 1e-4, 255).astype(np.uint8)     # margin in units of 1e-4, saturating
    out['SSSR_sample'] = gen.strided_sample(sssr, 1 << 17)
    out['SSSR_sum'] = gen.checksum(sssr)
    out['SISR_sample'] = gen.strided_sample(np_(outs[1]), 1 << 15)
    out['SISR_sum'] = gen.checksum(np_(outs[1]))
    out['SSSR_ft'] = np_(outs[2]); out['SISR_ft'] = np_(outs[3])
    out['losses'] = np.array([float(x.detach()) for x in L], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, 'head_256x512.npz'), **out)
    print('head_256x512.npz', out['losses'], out['margin_q'])


def golden_head_train(fname='head_train_256x512.npz', pseed=909, iseed=1010, h16=16, w16=32):
    """BASELINE's actual workload shape in TRAIN mode (round 3): 256x512 input -> 512x1024 logits (h16 x w16 = 16 x 32; `train512`: config 5's
    512x1024 -> 1024x2048, 32 x 64), B=2, BatchNorm batch statistics, Dropout modules in eval (set_mode), stage 3: the loss tuple, a strided
    logits / SISR sample, and for EVERY head parameter and both backbone-feature inputs the gradient of Total = CE + 0.1 MSE + FA (small tensors
    whole, large ones as a strided 4096-sample plus the order-independent checksum)."""
    out = {}
    head, outs, L, grads = run_head(gen.FULL, 3, pseed, iseed, 2, h16, w16, True, True)
    out['losses'] = np.array([float(x.detach()) for x in L], dtype=np.float64)
    out['SSSR_sample'] = gen.strided_sample(np_(outs[0]), 1 << 16)
    out['SISR_sample'] = gen.strided_sample(np_(outs[1]), 1 << 14)
    out['SSSR_ft'] = np_(outs[2]); out['SISR_ft'] = np_(outs[3])
    for k, g in grads.items():
        if g.size <= 8192:
            out[f'grad.{k}'] = g
        else:
            out[f'gradsample.{k}'] = gen.strided_sample(g, 4096)
            out[f'gradsum.{k}'] = gen.checksum(g)
    # The SAME reference modules in float64 on the same inputs: the backward pass through batch-statistics BatchNorm cancels heavily, and the
    # reference's own fp32 gradients are 1e-3 .. 2e-2 of their range away from these (err32.*) - a bound on "as accurate as the reference"
    # needs the exact values next to the fp32 ones.
    P = gen.make_head_params(pseed, gen.FULL, 3)
    x16, x4, target, org = gen.make_head_inputs(iseed, 2, h16, w16, gen.FULL)
    head64 = RefHead(gen.FULL, 3)
    load_params(head64, P)
    head64 = head64.double()
    set_mode(head64, True)
    a = t.from_numpy(x16).double().requires_grad_(True); b = t.from_numpy(x4).double().requires_grad_(True)
    outs64 = head64(a, b)
    L64 = losses(head64, outs64, target, org.astype(np.float64))
    L64[3].backward()
    g64 = {k: p.grad.numpy() for k, p in head64.named_parameters()}
    g64['backbone_features'], g64['lowlevel_features'] = a.grad.numpy(), b.grad.numpy()
    out['losses64'] = np.array([float(x.detach()) for x in L64], dtype=np.float64)
    out['SSSR_sample64'] = gen.strided_sample(outs64[0].detach().numpy(), 1 << 16)
    for k, g in g64.items():
        s64 = g if g.size <= 8192 else gen.strided_sample(g, 4096)
        s32 = out[f'grad.{k}'] if g.size <= 8192 else out[f'gradsample.{k}']
        out[f'grad64.{k}'] = np.asarray(s64, np.float64)
        out[f'err32.{k}'] = np.array(np.abs(np.asarray(s32, np.float64) - s64).max() / max(np.abs(s64).max(), 1e-300))
    for k, v in head.state_dict().items():
        if 'running_' in k and v.numel() <= 256:
            out[f'new.{k}'] = np_(v)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, out['losses'], len(out), 'arrays')


def golden_head_train_256x512():
    golden_head_train()


def golden_head_train_512x1024():
    """The same at BASELINE config 5's size (round 3): 512x1024 input -> 1024x2048 logits, B=2, train mode."""
    golden_head_train('head_train_512x1024.npz', 1111, 1212, 32, 64)


def golden_head_512x1024():
    """BASELINE config 5 size (512x1024 input -> 1024x2048 logits), eval forward, B=1: argmax map, top-2 margins, strided logits
    sample, checksums, feature-transformer maps (FA on 32x32 similarity matrices, n = 1024) and loss values."""
    out = {}
    head, outs, L, _ = run_head(gen.FULL, 3, 707, 808, 1, 32, 64, False, False)
    sssr = np_(outs[0])
    out['SSSR_argmax'] = sssr.argmax(axis=1).astype(np.uint8)
    top2 = np.sort(sssr, axis=1)[:, -2:]
    margin = (top2[:, 1] - top2[:, 0])
    out['margin_q'] = np.quantile(margin, [0, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 0.5]).astype(np.float64)
    out['margin_u8'] = np.minimum(margin / 1e-4, 255).astype(np.uint8)
    out['SSSR_sample'] = gen.strided_sample(sssr, 1 << 17)
    out['SSSR_sum'] = gen.checksum(sssr)
    out['SISR_sample'] = gen.strided_sample(np_(outs[1]), 1 << 15)
    out['SISR_sum'] = gen.checksum(np_(outs[1]))
    out['SSSR_ft'] = np_(outs[2]); out['SISR_ft'] = np_(outs[3])
    out['losses'] = np.array([float(x.detach()) for x in L], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, 'head_512x1024.npz'), **out)
    print('head_512x1024.npz', out['losses'], out['margin_q'])


def golden_train_steps():
    """Two SGD steps of the small head (train-mode BN, dropout off): losses per step, a few parameters after
    each step (train_or_resume.py:63-66, 435-445)."""
    out = {}
    W = gen.SMALL
    P = gen.make_head_params(101, W, 3)
    head = RefHead(W, 3); load_params(head, P); set_mode(head, True)
    opt = t.optim.SGD(head.parameters(), lr=0.006, momentum=0.9, weight_decay=5e-4)
    for step in range(2):
        x16, x4, target, org = gen.make_head_inputs(700 + step, 2, 2, 4, W)
        opt.zero_grad()
        outs = head(t.from_numpy(x16), t.from_numpy(x4))
        L = losses(head, outs, target, org)
        L[3].backward(); opt.step()
        out[f'step{step}.losses'] = np.array([float(x.detach()) for x in L], dtype=np.float64)
        for k, v in head.state_dict().items():
            if 'num_batches' not in k:
                out[f'step{step}.{k}'] = np_(v)
    np.savez_compressed(os.path.join(HERE, 'train_steps.npz'), **out)
    print('train_steps.npz', out['step0.losses'], out['step1.losses'])


def golden_pipeline_and_metrics():
    """metrices/mIoU.py + Accuracy.py on seeded class maps; JointScaledImage.py on a normalised float crop (the other two
    transforms of the deterministic tail need torchvision/PIL and are restated formulas)."""
    import importlib.util
    out = {}
    from metrices import mIoU, Accuracy                      # numpy only
    rs = np.random.RandomState(31)
    m, a = mIoU(num_classes=19), Accuracy()
    for b in range(3):
        nc = (19, 7, 12)[b]                                  # some batches miss classes -> nan handling
        pred = rs.randint(0, nc, (2, 24, 40)); target = rs.randint(0, nc, (2, 24, 40)).astype(np.uint8)
        agree = rs.uniform(size=pred.shape) < 0.6
        pred = np.where(agree, target, pred)
        target[rs.uniform(size=target.shape) < 0.1] = 255
        valid = target != 255
        m.update(pred, target, valid); a.update(pred, target, valid)
        out[f'metrics.pred{b}'] = pred.astype(np.uint8); out[f'metrics.target{b}'] = target
        out[f'metrics.batch_miou{b}'] = np.float64(m.ious[-1]); out[f'metrics.batch_acc{b}'] = np.float64(a.accuracies[-1])
    out['metrics.miou'] = np.float64(m()); out['metrics.acc'] = np.float64(a())
    spec = importlib.util.spec_from_file_location('JointScaledImage', os.path.join(REF, 'models', 'transforms', 'JointScaledImage.py'))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    rgb = rs.randint(0, 256, (2, 37, 53, 3)).astype(np.uint8); lab = rs.randint(0, 34, (2, 37, 53)).astype(np.uint8)
    mean, std = (0.28690, 0.32513, 0.28389), (0.17614, 0.18099, 0.17772)
    # the label table of the REFERENCE (datasets/Cityscapes/settings.py:9-17: pure constants, `from consts import *` resolves through REF on sys.path),
    # loaded by file path because the top-level name `datasets` is taken by the HuggingFace package
    cspec = importlib.util.spec_from_file_location('ref_cityscapes_settings', os.path.join(REF, 'datasets', 'Cityscapes', 'settings.py'))
    cs = importlib.util.module_from_spec(cspec); cspec.loader.exec_module(cs)
    mean, std = cs.MEAN, cs.STD
    lut = np.full(256, 255, np.uint8)
    for k, v in cs.LABEL_MAPPING_DICT.items():
        if 0 <= k < 256:
            lut[k] = v
    tr = mod.JointScaledImage(new_img_sizes=((16, 32), (32, 64)), new_seg_size=(32, 64))
    imgs_in, imgs_org, segs = [], [], []
    for n in range(2):
        x = (t.from_numpy(rgb[n]).permute(2, 0, 1).float() / 255. - t.tensor(mean).view(3, 1, 1)) / t.tensor(std).view(3, 1, 1)    # ToTensor + Normalize
        (i1, i2), (sg, _) = tr(x, t.from_numpy(lut[lab[n]]))
        imgs_in.append(np_(i1)); imgs_org.append(np_(i2)); segs.append(np_(sg))
    out.update({'prep.rgb': rgb, 'prep.labels': lab, 'prep.img_in': np.stack(imgs_in), 'prep.img_org': np.stack(imgs_org), 'prep.target': np.stack(segs)})
    np.savez_compressed(os.path.join(HERE, 'pipeline_metrics.npz'), **out)
    print('pipeline_metrics.npz', float(out['metrics.miou']), float(out['metrics.acc']))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'pipeline':
        golden_pipeline_and_metrics(); sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'c5':
        golden_head_512x1024(); sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'train256':
        golden_head_train_256x512(); sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'train512':
        golden_head_train_512x1024(); sys.exit(0)
    golden_pipeline_and_metrics()
    golden_fa()
    golden_ops()
    golden_head_small()
    golden_head_fullwidth()
    golden_head_256x512()
    golden_head_512x1024()
    golden_head_train_256x512()
    golden_head_train_512x1024()
    golden_train_steps()
