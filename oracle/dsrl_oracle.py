"""numpy restatement of the DSRL head, its losses and their gradients (TEST INFRASTRUCTURE ONLY).

Every function cites the reference file:line (paths under /root/reference) whose arithmetic it restates.
The reference delegates all arithmetic to torch.nn (PyTorch 1.7 pinned; semantics unchanged in 2.10), so
the formulas below are the published definitions of those torch ops; they are pinned by the golden
vectors in tests/golden (generated from the imported reference modules by tests/golden/make_golden.py).

Layout: logical NCHW numpy arrays (the reference's layout).  dtype follows the inputs (float64 for
checking, float32 for the timed CPU baseline).
"""
import numpy as np
from . import philox

BN_EPS = 1e-5          # torch.nn.BatchNorm2d default (DSRL.py:24,40,48,61,94; ASPP.py:20)
BN_MOMENTUM = 0.1

__all__ = [
    'conv2d', 'conv2d_bwd', 'batchnorm_train', 'batchnorm_eval', 'batchnorm_train_bwd', 'batchnorm_eval_bwd',
    'relu', 'relu_bwd', 'upsample_bilinear_ac', 'upsample_bilinear_ac_bwd', 'conv_transpose2d_k2s2',
    'conv_transpose2d_k2s2_bwd', 'pixel_shuffle', 'pixel_shuffle_bwd', 'avg_pool2d', 'avg_pool2d_bwd',
    'global_avg_pool', 'global_avg_pool_bwd', 'max_pool3x3s2', 'max_pool3x3s2_bwd', 'fa_similarity', 'fa_loss', 'fa_loss_bwd',
    'cross_entropy', 'cross_entropy_bwd', 'mse', 'mse_bwd', 'sgd_step', 'dropout_mask',
    'Tape', 'Var', 'head_forward', 'total_loss', 'HeadOutputs', 'miou_batch', 'backbone_forward', 'model_forward', 'seg_metrics_batch', 'prepare_batch',
]


# --------------------------------------------------------------------------------------------------
# primitive ops
# --------------------------------------------------------------------------------------------------
def _out_size(n, k, stride, pad, dil):
    return (n + 2 * pad - dil * (k - 1) - 1) // stride + 1


def conv2d(x, w, b=None, stride=1, pad=0, dil=1):
    """torch.nn.Conv2d forward (cross-correlation). Call sites: ASPP.py:10-15,19; DSRL.py:19-23,34-38,
    42-46,50,78-83,88-93.  x (N,C,H,W), w (K,C,R,S) -> (N,K,Ho,Wo)."""
    N, C, H, W = x.shape
    K, _, R, S = w.shape
    Ho, Wo = _out_size(H, R, stride, pad, dil), _out_size(W, S, stride, pad, dil)
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad))) if pad else x
    y = np.zeros((N, K, Ho, Wo), dtype=x.dtype)
    for r in range(R):
        for s in range(S):
            xs = xp[:, :, r * dil: r * dil + stride * (Ho - 1) + 1: stride, s * dil: s * dil + stride * (Wo - 1) + 1: stride]
            y += np.tensordot(w[:, :, r, s], xs, axes=([1], [1])).transpose(1, 0, 2, 3)
    if b is not None:
        y += b.reshape(1, K, 1, 1)
    return y


def conv2d_bwd(x, w, dy, stride=1, pad=0, dil=1, has_bias=False):
    """Gradients of conv2d (what torch autograd derives for DSRL.forward)."""
    N, C, H, W = x.shape
    K, _, R, S = w.shape
    Ho, Wo = dy.shape[2], dy.shape[3]
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad))) if pad else x
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    for r in range(R):
        for s in range(S):
            sl = (slice(None), slice(None), slice(r * dil, r * dil + stride * (Ho - 1) + 1, stride),
                  slice(s * dil, s * dil + stride * (Wo - 1) + 1, stride))
            xs = xp[sl]
            dw[:, :, r, s] = np.tensordot(dy, xs, axes=([0, 2, 3], [0, 2, 3]))
            dxp[sl] += np.tensordot(w[:, :, r, s], dy, axes=([0], [1])).transpose(1, 0, 2, 3)
    dx = dxp[:, :, pad:pad + H, pad:pad + W] if pad else dxp
    db = dy.sum(axis=(0, 2, 3)) if has_bias else None
    return np.ascontiguousarray(dx), dw, db


def batchnorm_train(x, gamma, beta, running_mean=None, running_var=None, eps=BN_EPS, momentum=BN_MOMENTUM):
    """torch.nn.BatchNorm2d in training mode: batch statistics over (N,H,W), biased variance for the
    normalisation, unbiased for the running estimate. Returns y, (mean, invstd), new running stats."""
    n = x.shape[0] * x.shape[2] * x.shape[3]
    mean = x.mean(axis=(0, 2, 3))
    var = x.var(axis=(0, 2, 3))
    invstd = 1.0 / np.sqrt(var + eps)
    y = (x - mean.reshape(1, -1, 1, 1)) * (invstd * gamma).reshape(1, -1, 1, 1) + beta.reshape(1, -1, 1, 1)
    new_rm = new_rv = None
    if running_mean is not None:
        new_rm = (1 - momentum) * running_mean + momentum * mean
        new_rv = (1 - momentum) * running_var + momentum * var * (n / max(n - 1, 1))
    return y, (mean, invstd), (new_rm, new_rv)


def batchnorm_eval(x, gamma, beta, running_mean, running_var, eps=BN_EPS):
    invstd = 1.0 / np.sqrt(running_var + eps)
    y = (x - running_mean.reshape(1, -1, 1, 1)) * (invstd * gamma).reshape(1, -1, 1, 1) + beta.reshape(1, -1, 1, 1)
    return y, (running_mean, invstd)


def batchnorm_train_bwd(x, gamma, mean, invstd, dy):
    n = x.shape[0] * x.shape[2] * x.shape[3]
    xhat = (x - mean.reshape(1, -1, 1, 1)) * invstd.reshape(1, -1, 1, 1)
    dbeta = dy.sum(axis=(0, 2, 3))
    dgamma = (dy * xhat).sum(axis=(0, 2, 3))
    dx = (gamma * invstd / n).reshape(1, -1, 1, 1) * (n * dy - dbeta.reshape(1, -1, 1, 1) - xhat * dgamma.reshape(1, -1, 1, 1))
    return dx, dgamma, dbeta


def batchnorm_eval_bwd(x, gamma, mean, invstd, dy):
    xhat = (x - mean.reshape(1, -1, 1, 1)) * invstd.reshape(1, -1, 1, 1)
    dbeta = dy.sum(axis=(0, 2, 3))
    dgamma = (dy * xhat).sum(axis=(0, 2, 3))
    dx = dy * (gamma * invstd).reshape(1, -1, 1, 1)
    return dx, dgamma, dbeta


def relu(x):
    return np.maximum(x, 0)


def relu_bwd(y, dy):
    return dy * (y > 0)


def dropout_mask(shape_nchw, p, seed, stream):
    """Keep-mask of the device Philox generator (oracle/philox.py); torch's own Dropout stream cannot be
    reproduced, so train-mode parity vs the *reference* runs with Dropout modules in eval."""
    return philox.dropout_keep_mask_nchw(shape_nchw, p, seed, stream)


def _ac_coords(n_in, n_out, dtype):
    """align_corners=True source coordinates (torch area_pixel_compute_scale): scale=(in-1)/(out-1)."""
    if n_out > 1:
        scale = dtype.type(n_in - 1) / dtype.type(n_out - 1)
    else:
        scale = dtype.type(0)
    src = scale * np.arange(n_out, dtype=dtype)
    i0 = np.floor(src).astype(np.int64)
    i0 = np.minimum(i0, n_in - 1)
    i1 = np.minimum(i0 + 1, n_in - 1)
    l1 = (src - i0).astype(dtype)
    return i0, i1, l1


def upsample_bilinear_ac(x, out_hw):
    """nn.UpsamplingBilinear2d / F.interpolate(mode='bilinear', align_corners=True).
    Call sites: ASPP.py:41 (1x1 -> HxW broadcast), DSRL.py:53 (x2), DSRL.py:163 (x4)."""
    Ho, Wo = out_hw
    h0, h1, lh = _ac_coords(x.shape[2], Ho, x.dtype)
    w0, w1, lw = _ac_coords(x.shape[3], Wo, x.dtype)
    lh = lh.reshape(1, 1, Ho, 1); lw = lw.reshape(1, 1, 1, Wo)
    top = x[:, :, h0][:, :, :, w0] * (1 - lw) + x[:, :, h0][:, :, :, w1] * lw
    bot = x[:, :, h1][:, :, :, w0] * (1 - lw) + x[:, :, h1][:, :, :, w1] * lw
    return top * (1 - lh) + bot * lh


def upsample_bilinear_ac_bwd(in_hw, dy):
    H, W = in_hw
    N, C, Ho, Wo = dy.shape
    h0, h1, lh = _ac_coords(H, Ho, dy.dtype)
    w0, w1, lw = _ac_coords(W, Wo, dy.dtype)
    # separable: first reduce along W then along H
    tmp = np.zeros((N, C, Ho, W), dtype=dy.dtype)
    np.add.at(tmp, (slice(None), slice(None), slice(None), w0), dy * (1 - lw))
    np.add.at(tmp, (slice(None), slice(None), slice(None), w1), dy * lw)
    dx = np.zeros((N, C, H, W), dtype=dy.dtype)
    np.add.at(dx, (slice(None), slice(None), h0), tmp * (1 - lh).reshape(1, 1, Ho, 1))
    np.add.at(dx, (slice(None), slice(None), h1), tmp * lh.reshape(1, 1, Ho, 1))
    return dx


def conv_transpose2d_k2s2(x, w, b=None):
    """nn.ConvTranspose2d(kernel_size=2, stride=2, padding=0) (DSRL.py:55-60, 64-69).
    w is (Cin, Cout, 2, 2); kernel==stride so every output pixel is one Cin-long dot product."""
    N, Ci, H, W = x.shape
    Co = w.shape[1]
    y = np.empty((N, Co, 2 * H, 2 * W), dtype=x.dtype)
    for i in range(2):
        for j in range(2):
            y[:, :, i::2, j::2] = np.tensordot(w[:, :, i, j], x, axes=([0], [1])).transpose(1, 0, 2, 3)
    if b is not None:
        y += b.reshape(1, Co, 1, 1)
    return y


def conv_transpose2d_k2s2_bwd(x, w, dy, has_bias=False):
    dx = np.zeros_like(x)
    dw = np.zeros_like(w)
    for i in range(2):
        for j in range(2):
            dys = dy[:, :, i::2, j::2]
            dx += np.tensordot(w[:, :, i, j], dys, axes=([1], [1])).transpose(1, 0, 2, 3)
            dw[:, :, i, j] = np.tensordot(x, dys, axes=([0, 2, 3], [0, 2, 3]))
    db = dy.sum(axis=(0, 2, 3)) if has_bias else None
    return dx, dw, db


def pixel_shuffle(x, r):
    """nn.PixelShuffle(r) (DSRL.py:84): out[n,c,h*r+i,w*r+j] = in[n,c*r*r+i*r+j,h,w]."""
    N, C, H, W = x.shape
    c = C // (r * r)
    return np.ascontiguousarray(x.reshape(N, c, r, r, H, W).transpose(0, 1, 4, 2, 5, 3).reshape(N, c, H * r, W * r))


def pixel_shuffle_bwd(dy, r):
    N, c, Hr, Wr = dy.shape
    H, W = Hr // r, Wr // r
    return np.ascontiguousarray(dy.reshape(N, c, H, r, W, r).transpose(0, 1, 3, 5, 2, 4).reshape(N, c * r * r, H, W))


def avg_pool2d(x, k):
    """nn.AvgPool2d(k) (FALoss.py:23-24); floor mode, stride=k."""
    N, C, H, W = x.shape
    Ho, Wo = H // k, W // k
    return x[:, :, :Ho * k, :Wo * k].reshape(N, C, Ho, k, Wo, k).mean(axis=(3, 5))


def avg_pool2d_bwd(in_hw, dy, k):
    H, W = in_hw
    N, C, Ho, Wo = dy.shape
    dx = np.zeros((N, C, H, W), dtype=dy.dtype)
    dx[:, :, :Ho * k, :Wo * k] = np.repeat(np.repeat(dy, k, axis=2), k, axis=3) / (k * k)
    return dx


def global_avg_pool(x):
    """nn.AdaptiveAvgPool2d((1,1)) (ASPP.py:22,38)."""
    return x.mean(axis=(2, 3), keepdims=True)


def global_avg_pool_bwd(in_hw, dy):
    H, W = in_hw
    return np.broadcast_to(dy / (H * W), dy.shape[:2] + (H, W)).copy()


def max_pool3x3s2(x):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (ResNet101.py:32) - backbone, row f1."""
    N, C, H, W = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)), constant_values=-np.inf)
    wins = np.stack([xp[:, :, r: r + 2 * (Ho - 1) + 1: 2, s: s + 2 * (Wo - 1) + 1: 2] for r in range(3) for s in range(3)], axis=0)
    arg = wins.argmax(axis=0)
    return wins.max(axis=0), arg


def max_pool3x3s2_bwd(in_hw, arg, dy):
    H, W = in_hw
    N, C, Ho, Wo = dy.shape
    dxp = np.zeros((N, C, H + 2, W + 2), dtype=dy.dtype)
    for t in range(9):
        r, s = divmod(t, 3)
        dxp[:, :, r: r + 2 * (Ho - 1) + 1: 2, s: s + 2 * (Wo - 1) + 1: 2] += dy * (arg == t)
    return dxp[:, :, 1:-1, 1:-1]


# --------------------------------------------------------------------------------------------------
# Feature-affinity loss (FALoss.py)
# --------------------------------------------------------------------------------------------------
def fa_similarity(x):
    """FALoss._calculate_matrix_similarity (FALoss.py:8-11): Xn = X / sigma_1(X) per (b,c) slice,
    S = Xn^T Xn.  sigma_1 = matrix 2-norm = largest singular value of the (H',W') slice."""
    u, s, vt = np.linalg.svd(x, full_matrices=False)
    sigma = s[..., 0]
    with np.errstate(divide='ignore', invalid='ignore'):
        xn = x / sigma[..., None, None]
        S = np.matmul(np.swapaxes(xn, -1, -2), xn)
    return S, xn, sigma, u[..., :, 0], vt[..., 0, :]


def fa_loss(fm1, fm2, subsample_factor=8, reduction='mean'):
    """FALoss.forward (FALoss.py:18-34): avg-pool, similarity of each map, then the L1 distance between
    EVERY element of S1 and EVERY element of S2 (repeat_interleave vs repeat, :27-30)."""
    assert fm1.ndim == 4 and fm1.shape == fm2.shape          # FALoss.py:19-20
    p1 = avg_pool2d(fm1, subsample_factor)
    p2 = avg_pool2d(fm2, subsample_factor)
    S1 = fa_similarity(p1)[0]
    S2 = fa_similarity(p2)[0]
    B, C = S1.shape[:2]
    a = S1.reshape(B, C, -1)
    b = S2.reshape(B, C, -1)
    d = np.abs(a[:, :, :, None] - b[:, :, None, :])           # (B,C,n,n): [i,j] = |S1_i - S2_j|
    if reduction == 'mean':
        return d.mean()
    if reduction == 'sum':
        return d.sum()
    return d.reshape(B, C, -1)                                # 'none': (B,C,n*n) in repeat_interleave order


def fa_loss_bwd(fm1, fm2, subsample_factor=8, reduction='mean', dloss=1.0):
    """Gradient of fa_loss w.r.t. both feature maps (autograd of FALoss.py:8-34; sign(0)=0 as
    torch's l1_loss; d sigma_1 / dX = u1 v1^T)."""
    k = subsample_factor
    p = [avg_pool2d(fm1, k), avg_pool2d(fm2, k)]
    sims = [fa_similarity(q) for q in p]
    B, C = p[0].shape[:2]
    n = sims[0][0].shape[-1] ** 2
    a = sims[0][0].reshape(B, C, n)
    b = sims[1][0].reshape(B, C, n)
    sg = np.sign(a[:, :, :, None] - b[:, :, None, :])
    scale = dloss / (B * C * n * n) if reduction == 'mean' else dloss
    dS = [sg.sum(axis=3) * scale, -sg.sum(axis=2) * scale]
    out = []
    for q, (S, xn, sigma, u1, v1), dSf, fm in zip(p, sims, dS, (fm1, fm2)):
        dSm = dSf.reshape(S.shape)
        dxn = np.matmul(xn, dSm + np.swapaxes(dSm, -1, -2))
        sig = sigma[..., None, None]
        dsigma = -(dxn * q).sum(axis=(-1, -2), keepdims=True) / (sig * sig)
        dq = dxn / sig + dsigma * (u1[..., :, None] * v1[..., None, :])
        out.append(avg_pool2d_bwd(fm.shape[2:], dq, k))
    return out[0], out[1]


# --------------------------------------------------------------------------------------------------
# CE / MSE / SGD (train_or_resume.py:116-119, 435-445, 63-66)
# --------------------------------------------------------------------------------------------------
def cross_entropy(logits, target, ignore_index=255):
    """nn.CrossEntropyLoss(ignore_index=255), mean over non-ignored pixels (train_or_resume.py:116,435)."""
    m = logits.max(axis=1, keepdims=True)
    lse = m + np.log(np.exp(logits - m).sum(axis=1, keepdims=True))
    logp = logits - lse
    valid = target != ignore_index
    tgt = np.where(valid, target, 0).astype(np.int64)
    picked = np.take_along_axis(logp, tgt[:, None], axis=1)[:, 0]
    nvalid = valid.sum()
    return -(picked * valid).sum() / nvalid


def cross_entropy_bwd(logits, target, ignore_index=255, dloss=1.0):
    m = logits.max(axis=1, keepdims=True)
    e = np.exp(logits - m)
    sm = e / e.sum(axis=1, keepdims=True)
    valid = target != ignore_index
    tgt = np.where(valid, target, 0).astype(np.int64)
    onehot = np.zeros_like(sm)
    np.put_along_axis(onehot, tgt[:, None], 1.0, axis=1)
    return (sm - onehot) * valid[:, None] * (dloss / valid.sum())


def mse(a, b):
    """nn.MSELoss() mean (train_or_resume.py:117,436)."""
    return ((a - b) ** 2).mean()


def mse_bwd(a, b, dloss=1.0):
    return 2.0 * (a - b) * (dloss / a.size)


def sgd_step(p, g, buf, lr, momentum, weight_decay):
    """torch.optim.SGD(momentum, weight_decay), dampening 0, no nesterov (train_or_resume.py:63-66,445).
    buf=None on the first step (torch clones the decayed grad)."""
    d = g + weight_decay * p
    buf = d.copy() if buf is None else momentum * buf + d
    return p - lr * buf, buf


def miou_batch(pred, target, num_classes=19, ignore_index=255):
    """metrices/mIoU.py:15-41 semantics on one batch: returns (sum_inter/sum_union %, mean-of-IoU %)."""
    valid = target != ignore_index
    inter = np.zeros(num_classes); union = np.zeros(num_classes)
    for c in range(num_classes):
        p = (pred == c) & valid; t = (target == c) & valid
        inter[c] = (p & t).sum(); union[c] = (p | t).sum()
    with np.errstate(divide='ignore', invalid='ignore'):
        iou = inter / union
    return 100.0 * inter.sum() / max(union.sum(), 1), 100.0 * np.nanmean(iou)


def seg_metrics_batch(pred, target, num_classes=19, ignore_index=255):
    """One update() of metrices/mIoU.py:15-41 and metrices/Accuracy.py:13-30: (nan-mean IoU over classes, correct/valid)."""
    valid = target != ignore_index
    p1 = (pred.astype(np.int64) + 1) * valid                     # mIoU.py:22-25
    t1 = np.where(valid, target.astype(np.int64) + 1, 0)          # uint8 255+1 wraps to 0 in the reference: outside the histogram range
    inter = p1 * (p1 == t1)
    bins = dict(bins=num_classes, range=(1, num_classes))
    area_pred = np.histogram(p1, **bins)[0]; area_inter = np.histogram(inter, **bins)[0]; area_target = np.histogram(t1, **bins)[0]
    union = area_pred + area_target - area_inter
    with np.errstate(divide='ignore', invalid='ignore'):
        miou = np.nanmean(area_inter / union)
    return miou, ((pred == target) * valid).sum() / valid.sum()


def prepare_batch(rgb_u8, labels_u8, label_mapping, mean, std, model_input_size, ignore_label=255, dtype=np.float64):
    """ToTensor + Normalize (JointNormalize.py:11), label-id remap (JointImageAndLabelTensor.py:9-16) and the dual-scale resize of
    JointScaledImage.py:27-32 on a decoded (N,Hs,Ws,3) uint8 crop. Returns img_in (N,3,H,W), img_org (N,3,2H,2W), target (N,2H,2W)."""
    H, W = model_input_size
    x = (np.transpose(rgb_u8, (0, 3, 1, 2)).astype(dtype) / 255.0 - np.asarray(mean, dtype).reshape(1, 3, 1, 1)) / np.asarray(std, dtype).reshape(1, 3, 1, 1)
    lut = np.full(256, ignore_label, np.uint8)
    for k, v in label_mapping.items():
        if 0 <= k < 256:
            lut[k] = v
    seg = lut[labels_u8]
    Hs, Ws = labels_u8.shape[1:]
    hs = np.minimum(np.floor(np.arange(2 * H) * (Hs / (2 * H))).astype(np.int64), Hs - 1)      # torch 'nearest'
    ws = np.minimum(np.floor(np.arange(2 * W) * (Ws / (2 * W))).astype(np.int64), Ws - 1)
    return upsample_bilinear_ac(x, (H, W)), upsample_bilinear_ac(x, (2 * H, 2 * W)), seg[:, hs][:, :, ws]


# --------------------------------------------------------------------------------------------------
# a tiny reverse-mode tape so the composite head (DSRL.forward) gets its gradients from the
# primitive backward functions above
# --------------------------------------------------------------------------------------------------
class Var:
    __slots__ = ('v', 'g', 'bw', 'name')

    def __init__(self, v, bw=None, name=None):
        self.v = v; self.g = None; self.bw = bw; self.name = name

    def acc(self, g):
        if g is None:
            return
        self.g = g if self.g is None else self.g + g


class Tape:
    def __init__(self):
        self.nodes = []

    def leaf(self, v, name=None):
        return Var(np.asarray(v), None, name)

    def node(self, v, bw):
        n = Var(v, bw)
        self.nodes.append(n)
        return n

    def backward(self):
        for n in reversed(self.nodes):
            if n.g is not None and n.bw is not None:
                n.bw(n.g)

    # ---- differentiable ops ------------------------------------------------------------------
    def conv(self, x, w, b=None, stride=1, pad=0, dil=1):
        y = conv2d(x.v, w.v, None if b is None else b.v, stride, pad, dil)

        def bw(dy):
            dx, dw, db = conv2d_bwd(x.v, w.v, dy, stride, pad, dil, b is not None)
            x.acc(dx); w.acc(dw)
            if b is not None:
                b.acc(db)
        return self.node(y, bw)

    def bn(self, x, gamma, beta, rm, rv, training, stats_out=None, key=None):
        if training:
            y, (mean, invstd), new = batchnorm_train(x.v, gamma.v, beta.v, rm, rv)
            if stats_out is not None:
                stats_out[key] = new

            def bw(dy):
                dx, dg, db = batchnorm_train_bwd(x.v, gamma.v, mean, invstd, dy)
                x.acc(dx); gamma.acc(dg); beta.acc(db)
        else:
            y, (mean, invstd) = batchnorm_eval(x.v, gamma.v, beta.v, rm, rv)

            def bw(dy):
                dx, dg, db = batchnorm_eval_bwd(x.v, gamma.v, mean, invstd, dy)
                x.acc(dx); gamma.acc(dg); beta.acc(db)
        return self.node(y, bw)

    def relu(self, x):
        y = relu(x.v)
        return self.node(y, lambda dy: x.acc(relu_bwd(y, dy)))

    def dropout(self, x, p, seed, stream):
        if seed is None or p == 0.0:
            return x
        keep = dropout_mask(x.v.shape, p, seed, stream)
        sc = x.v.dtype.type(1.0 / (1.0 - p))
        return self.node(x.v * keep * sc, lambda dy: x.acc(dy * keep * sc))

    def upsample(self, x, out_hw):
        y = upsample_bilinear_ac(x.v, out_hw)
        return self.node(y, lambda dy: x.acc(upsample_bilinear_ac_bwd(x.v.shape[2:], dy)))

    def cat(self, xs):
        y = np.concatenate([x.v for x in xs], axis=1)
        sizes = [x.v.shape[1] for x in xs]

        def bw(dy):
            o = 0
            for x, c in zip(xs, sizes):
                x.acc(dy[:, o:o + c]); o += c
        return self.node(y, bw)

    def convT(self, x, w, b=None):
        y = conv_transpose2d_k2s2(x.v, w.v, None if b is None else b.v)

        def bw(dy):
            dx, dw, db = conv_transpose2d_k2s2_bwd(x.v, w.v, dy, b is not None)
            x.acc(dx); w.acc(dw)
            if b is not None:
                b.acc(db)
        return self.node(y, bw)

    def pixel_shuffle(self, x, r):
        return self.node(pixel_shuffle(x.v, r), lambda dy: x.acc(pixel_shuffle_bwd(dy, r)))

    def gap(self, x):
        return self.node(global_avg_pool(x.v), lambda dy: x.acc(global_avg_pool_bwd(x.v.shape[2:], dy)))

    def maxpool(self, x):
        y, arg = max_pool3x3s2(x.v)
        return self.node(y, lambda dy: x.acc(max_pool3x3s2_bwd(x.v.shape[2:], arg, dy)))

    def add(self, a, b):
        def bw(dy):
            a.acc(dy); b.acc(dy)
        return self.node(a.v + b.v, bw)


class HeadOutputs:
    """The 4-tuple DSRL.forward returns (DSRL.py:186) plus the tape to back-propagate through."""
    def __init__(self):
        self.SSSR = self.SISR = self.SSSR_ft = self.SISR_ft = None
        self.tape = None; self.params = None; self.inputs = None; self.new_running = {}


# dropout stream ids (one per Dropout module, DSRL.py:41,49,54,63) - shared with the device code
DROPOUT_STREAMS = {'cat_conv.3': 1, 'cat_conv.7': 2, 'upsample16_pred.1': 3, 'upsample16_pred.5': 4}


def _cbr(tp, P, R, x, prefix, conv_i, bn_i, training, new_running, stride=1, pad=0, dil=1):
    """Conv(bias=False) -> BN -> ReLU, the unit ASPP.py:19-21 / DSRL.py:19-25,34-48,88-95 build."""
    z = tp.conv(x, P[f'{prefix}.{conv_i}.weight'], None, stride, pad, dil)
    z = tp.bn(z, P[f'{prefix}.{bn_i}.weight'], P[f'{prefix}.{bn_i}.bias'], R[f'{prefix}.{bn_i}.running_mean'],
              R[f'{prefix}.{bn_i}.running_var'], training, new_running, f'{prefix}.{bn_i}')
    return tp.relu(z)


def head_forward(params, backbone_features, lowlevel_features, stage=3, bn_training=False, dropout_seed=None,
                 aspp_rate=1, tape=None, leaves=None, new_running=None):
    """Replays DSRL.forward lines DSRL.py:162-184 on given backbone outputs.

    params: dict keyed by the reference's state_dict names (non-backbone entries; numpy arrays).
    dropout_seed None  => the four Dropout modules are identity (eval / parity runs).
    Returns HeadOutputs; call `.tape.backward()` after seeding `.SSSR.g` etc. to get gradients in
    `.params[name].g` and `.inputs[i].g`.
    """
    tp = tape if tape is not None else Tape()
    out = HeadOutputs(); out.tape = tp
    if leaves is not None:
        P, R = leaves
    else:
        P = {k: tp.leaf(v, k) for k, v in params.items() if not (k.endswith('running_mean') or k.endswith('running_var')
                                                                 or k.endswith('num_batches_tracked'))}
        R = {k: np.asarray(v) for k, v in params.items() if k.endswith('running_mean') or k.endswith('running_var')}
    out.params = P
    x16 = backbone_features if isinstance(backbone_features, Var) else tp.leaf(backbone_features, 'backbone_features')
    x4 = lowlevel_features if isinstance(lowlevel_features, Var) else tp.leaf(lowlevel_features, 'lowlevel_features')
    out.inputs = (x16, x4)
    if new_running is not None:
        out.new_running = new_running
    nr = out.new_running
    hw16 = x16.v.shape[2:]

    # ---- ASPP.forward (ASPP.py:36-44) ----
    pre = 'feature_extractor.aspp.branches'
    rates = [(1, 0, 1), (3, 6 * aspp_rate, 6 * aspp_rate), (3, 12 * aspp_rate, 12 * aspp_rate), (3, 18 * aspp_rate, 18 * aspp_rate)]
    branches = [_cbr(tp, P, R, x16, f'{pre}.{i}', 0, 1, bn_training, nr, 1, pad, dil) for i, (_, pad, dil) in enumerate(rates)]
    g = tp.gap(x16)                                                     # ASPP.py:38
    g = _cbr(tp, P, R, g, f'{pre}.4', 0, 1, bn_training, nr)             # ASPP.py:39
    g = tp.upsample(g, hw16)                                            # ASPP.py:40
    aspp = _cbr(tp, P, R, tp.cat(branches + [g]), f'{pre}.5', 0, 1, bn_training, nr)   # ASPP.py:44

    # ---- DSRL.py:163-165 ----
    hw4 = x4.v.shape[2:]
    aspp_up = tp.upsample(aspp, (hw16[0] * 4, hw16[1] * 4))             # UpsamplingBilinear2d(scale_factor=4)
    assert tuple(aspp_up.v.shape[2:]) == tuple(hw4)
    low = _cbr(tp, P, R, x4, 'feature_extractor.shortcut_conv', 0, 1, bn_training, nr)
    cat = tp.cat([aspp_up, low])

    # ---- SSSR decoder (DSRL.py:168-170) ----
    s = _cbr(tp, P, R, cat, 'SSSR_decoder.cat_conv', 0, 1, bn_training, nr, 1, 1, 1)
    s = tp.dropout(s, 0.2, dropout_seed, DROPOUT_STREAMS['cat_conv.3'])
    s = _cbr(tp, P, R, s, 'SSSR_decoder.cat_conv', 4, 5, bn_training, nr, 1, 1, 1)
    s = tp.dropout(s, 0.2, dropout_seed, DROPOUT_STREAMS['cat_conv.7'])
    s = tp.conv(s, P['SSSR_decoder.cls_conv.weight'], P['SSSR_decoder.cls_conv.bias'])
    u = 'SSSR_decoder.upsample16_pred'
    s = tp.upsample(s, (hw4[0] * 2, hw4[1] * 2))                        # DSRL.py:53
    s = tp.dropout(s, 0.2, dropout_seed, DROPOUT_STREAMS['upsample16_pred.1'])
    s = tp.convT(s, P[f'{u}.2.weight'])                                 # DSRL.py:55-60
    s = tp.bn(s, P[f'{u}.3.weight'], P[f'{u}.3.bias'], R[f'{u}.3.running_mean'], R[f'{u}.3.running_var'],
              bn_training, nr, f'{u}.3')
    s = tp.relu(s)
    s = tp.dropout(s, 0.2, dropout_seed, DROPOUT_STREAMS['upsample16_pred.5'])
    out.SSSR = tp.convT(s, P[f'{u}.6.weight'], P[f'{u}.6.bias'])        # DSRL.py:64-69

    if stage > 1:                                                       # DSRL.py:175-177
        z = tp.conv(cat, P['SISR_decoder.0.weight'], P['SISR_decoder.0.bias'], 1, 1, 1)
        out.SISR = tp.pixel_shuffle(z, 8)
        if stage > 2:                                                   # DSRL.py:179-184
            out.SSSR_ft = _cbr(tp, P, R, out.SSSR, 'SSSR_feature_transformer', 0, 1, bn_training, nr, 8, 0, 1)
            out.SISR_ft = _cbr(tp, P, R, out.SISR, 'SISR_feature_transformer', 0, 1, bn_training, nr, 8, 0, 1)
    return out


def backbone_forward(tp, P, R, x, bn_training, new_running, prefix='feature_extractor.backbone'):
    """ResNet-101, output stride 16 (ResNet101.py:91-104 with torchvision's Bottleneck: 1x1 -> 3x3(stride, dilation) -> 1x1,
    expansion 4, stride on the 3x3; layers [3,4,23,3], replace_stride_with_dilation=[F,F,T]).  Returns (layer4, layer1)."""
    def cbr(x, conv, bn, stride=1, pad=0, dil=1, relu=True):
        z = tp.conv(x, P[f'{prefix}.{conv}.weight'], None, stride, pad, dil)
        z = tp.bn(z, P[f'{prefix}.{bn}.weight'], P[f'{prefix}.{bn}.bias'], R[f'{prefix}.{bn}.running_mean'], R[f'{prefix}.{bn}.running_var'],
                  bn_training, new_running, f'{prefix}.{bn}')
        return tp.relu(z) if relu else z

    x = cbr(x, 'conv1', 'bn1', 2, 3, 1)
    x = tp.maxpool(x)
    low = None
    dilation = 1
    for li, (blocks, stride, dilate) in enumerate([(3, 1, False), (4, 2, False), (23, 2, False), (3, 2, True)], start=1):
        prev_dil = dilation
        if dilate:
            dilation *= stride; stride = 1
        for b in range(blocks):
            name = f'layer{li}.{b}'
            st = stride if b == 0 else 1
            dl = prev_dil if b == 0 else dilation
            identity = x
            if f'{prefix}.{name}.downsample.0.weight' in P:
                identity = cbr(x, f'{name}.downsample.0', f'{name}.downsample.1', st, 0, 1, relu=False)
            out = cbr(x, f'{name}.conv1', f'{name}.bn1')
            out = cbr(out, f'{name}.conv2', f'{name}.bn2', st, dl, dl)
            out = cbr(out, f'{name}.conv3', f'{name}.bn3', relu=False)
            x = tp.relu(tp.add(out, identity))
        if li == 1:
            low = x
    return x, low


def model_forward(params, image, stage=3, bn_training=False, dropout_seed=None):
    """Whole DSRL.forward (DSRL.py:158-186): ResNet-101 backbone + head, on an NCHW image."""
    tp = Tape()
    P = {k: tp.leaf(v, k) for k, v in params.items() if not (k.endswith('running_mean') or k.endswith('running_var') or k.endswith('num_batches_tracked'))}
    R = {k: np.asarray(v) for k, v in params.items() if k.endswith('running_mean') or k.endswith('running_var')}
    x = tp.leaf(image, 'image')
    nr = {}
    x16, x4 = backbone_forward(tp, P, R, x, bn_training, nr)
    out = head_forward(params, x16, x4, stage, bn_training, dropout_seed, tape=tp, leaves=(P, R), new_running=nr)
    return out


def total_loss(out, target, input_org, stage=3, w1=0.1, w2=1.0, ignore_index=255, backward=True):
    """Loss mix of train_or_resume.py:435-438; optionally seeds the output grads and runs the tape."""
    ce = cross_entropy(out.SSSR.v, target, ignore_index)
    ms = w1 * mse(out.SISR.v, input_org) if stage > 1 else 0.0
    fa = w2 * fa_loss(out.SSSR_ft.v, out.SISR_ft.v) if stage > 2 else 0.0
    if backward:
        out.SSSR.acc(cross_entropy_bwd(out.SSSR.v, target, ignore_index))
        if stage > 1:
            out.SISR.acc(mse_bwd(out.SISR.v, input_org, w1))
        if stage > 2:
            g1, g2 = fa_loss_bwd(out.SSSR_ft.v, out.SISR_ft.v, dloss=w2)
            out.SSSR_ft.acc(g1); out.SISR_ft.acc(g2)
        out.tape.backward()
    return ce, ms, fa, ce + ms + fa
