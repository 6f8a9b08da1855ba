"""Deterministic synthetic parameters / inputs shared by make_golden.py (which feeds them to the imported
reference) and by the tests (which feed them to the oracle and to the HIP path).

numpy only - this file travels to the GPU box; the reference does not.  `np.random.RandomState` is
used because its streams are frozen across numpy versions.
"""
import numpy as np

NUM_CLASSES = 19            # datasets/Cityscapes/settings.py:3
IGNORE = 255                # datasets/Cityscapes/settings.py:8

FULL = dict(c16=2048, c4=256, aspp=256, low=48, mid=256)        # DSRL.py:105-113
SMALL = dict(c16=64, c4=32, aspp=32, low=8, mid=32)             # reduced-width head for small fixtures


def _conv_w(rs, k, c, r, s):
    std = np.sqrt(2.0 / (k * r * s))                              # kaiming_normal_(fan_out, relu) scale
    return (rs.standard_normal((k, c, r, s)) * std).astype(np.float32)


def _bn(rs, prefix, c, P):
    P[f'{prefix}.weight'] = rs.uniform(0.5, 1.5, c).astype(np.float32)
    P[f'{prefix}.bias'] = (rs.standard_normal(c) * 0.1).astype(np.float32)
    P[f'{prefix}.running_mean'] = (rs.standard_normal(c) * 0.1).astype(np.float32)
    P[f'{prefix}.running_var'] = rs.uniform(0.5, 1.5, c).astype(np.float32)


def make_head_params(seed, widths=FULL, stage=3, nc=NUM_CLASSES):
    """Parameters of the non-backbone part of DSRL keyed by the reference's state_dict names."""
    rs = np.random.RandomState(seed)
    w = widths
    P = {}
    a = 'feature_extractor.aspp.branches'
    for i, (cin, k) in enumerate([(w['c16'], 1), (w['c16'], 3), (w['c16'], 3), (w['c16'], 3), (w['c16'], 1), (5 * w['aspp'], 1)]):
        P[f'{a}.{i}.0.weight'] = _conv_w(rs, w['aspp'], cin, k, k)
        _bn(rs, f'{a}.{i}.1', w['aspp'], P)
    P['feature_extractor.shortcut_conv.0.weight'] = _conv_w(rs, w['low'], w['c4'], 1, 1)
    _bn(rs, 'feature_extractor.shortcut_conv.1', w['low'], P)
    cc = w['aspp'] + w['low']
    P['SSSR_decoder.cat_conv.0.weight'] = _conv_w(rs, w['mid'], cc, 3, 3)
    _bn(rs, 'SSSR_decoder.cat_conv.1', w['mid'], P)
    P['SSSR_decoder.cat_conv.4.weight'] = _conv_w(rs, w['mid'], w['mid'], 3, 3)
    _bn(rs, 'SSSR_decoder.cat_conv.5', w['mid'], P)
    P['SSSR_decoder.cls_conv.weight'] = _conv_w(rs, nc, w['mid'], 1, 1)
    P['SSSR_decoder.cls_conv.bias'] = (rs.standard_normal(nc) * 0.05).astype(np.float32)
    u = 'SSSR_decoder.upsample16_pred'
    P[f'{u}.2.weight'] = (rs.standard_normal((nc, nc, 2, 2)) * np.sqrt(2.0 / (nc * 4))).astype(np.float32)
    _bn(rs, f'{u}.3', nc, P)
    P[f'{u}.6.weight'] = (rs.standard_normal((nc, nc, 2, 2)) * np.sqrt(2.0 / (nc * 4))).astype(np.float32)
    P[f'{u}.6.bias'] = (rs.standard_normal(nc) * 0.05).astype(np.float32)
    if stage > 1:
        P['SISR_decoder.0.weight'] = _conv_w(rs, 3 * 64, cc, 3, 3)
        P['SISR_decoder.0.bias'] = (rs.standard_normal(3 * 64) * 0.05).astype(np.float32)
    if stage > 2:
        P['SSSR_feature_transformer.0.weight'] = (rs.standard_normal((1, nc, 1, 1)) * 0.5).astype(np.float32)
        _bn(rs, 'SSSR_feature_transformer.1', 1, P)
        P['SISR_feature_transformer.0.weight'] = (rs.standard_normal((1, 3, 1, 1)) * 0.5).astype(np.float32)
        _bn(rs, 'SISR_feature_transformer.1', 1, P)
        # keep the 1-channel maps away from the all-zero (NaN) corner of FALoss
        P['SSSR_feature_transformer.1.bias'] = np.array([0.3], np.float32)
        P['SISR_feature_transformer.1.bias'] = np.array([0.3], np.float32)
    return P


def make_head_inputs(seed, batch, h16, w16, widths=FULL):
    """Synthetic backbone outputs (post-ReLU statistics), target with 10 % ignore pixels, HR image."""
    rs = np.random.RandomState(seed)
    x16 = np.maximum(rs.standard_normal((batch, widths['c16'], h16, w16)), 0).astype(np.float32)
    x4 = np.maximum(rs.standard_normal((batch, widths['c4'], 4 * h16, 4 * w16)), 0).astype(np.float32)
    Ho, Wo = 32 * h16, 32 * w16
    target = rs.randint(0, NUM_CLASSES, (batch, Ho, Wo)).astype(np.uint8)
    target[rs.uniform(size=target.shape) < 0.1] = IGNORE
    input_org = rs.standard_normal((batch, 3, Ho, Wo)).astype(np.float32)
    return x16, x4, target, input_org


def checksum(a, seed=7):
    """Order-independent summary of a big tensor: sum, abs-sum, seeded random projection, strided sample."""
    a = np.asarray(a, dtype=np.float64).ravel()
    rs = np.random.RandomState(seed)
    proj = rs.standard_normal(min(a.size, 1 << 16))
    idx = np.linspace(0, a.size - 1, proj.size).astype(np.int64)
    return np.array([a.sum(), np.abs(a).sum(), float(a[idx] @ proj)], dtype=np.float64)


def strided_sample(a, n=4096):
    a = np.asarray(a).ravel()
    idx = np.linspace(0, a.size - 1, min(n, a.size)).astype(np.int64)
    return a[idx]
