"""Philox4x32-10 counter-based RNG, numpy restatement of the device generator.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The HIP kernels in
dualsuperreslearningforsemseg_amd/csrc/common.h (`philox4x32_10`) draw dropout masks from exactly
this function, so train-mode parity (reference Dropout(p=0.2), /root/reference/models/DSRL.py:41,49,54,63)
can be checked bit-for-bit between the oracle and the GPU path.

Convention shared with the device code:
  element e (= linear index of the element in the *NHWC* physical buffer) draws
  word  (e & 3)  of  philox(key=(seed_lo, seed_hi), counter=(e>>2 lo32, e>>2 hi32, stream, 0))
  u = (word >> 8) * 2**-24           in [0, 1)
  keep = u >= p ; scale = 1/(1-p)
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """All arguments uint32 arrays (broadcastable). Returns 4 uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint32); c1 = np.asarray(c1, dtype=np.uint32)
    c2 = np.asarray(c2, dtype=np.uint32); c3 = np.asarray(c3, dtype=np.uint32)
    k0 = np.uint32(k0); k1 = np.uint32(k1)
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32); lo0 = (p0 & _MASK32).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32); lo1 = (p1 & _MASK32).astype(np.uint32)
            n0 = hi1 ^ c1 ^ k0
            n1 = lo1
            n2 = hi0 ^ c3 ^ k1
            n3 = lo0
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32(k0 + _W0)
            k1 = np.uint32(k1 + _W1)
    return c0, c1, c2, c3


def uniform_for_elements(numel, seed, stream):
    """u[e] for e in [0, numel): float32 uniforms in [0,1) following the convention above."""
    e = np.arange(numel, dtype=np.uint64)
    q = e >> np.uint64(2)
    c0 = (q & _MASK32).astype(np.uint32)
    c1 = (q >> np.uint64(32)).astype(np.uint32)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    r = philox4x32_10(c0, c1, np.uint32(stream), np.uint32(0), seed & 0xFFFFFFFF, seed >> 32)
    words = np.stack(r, axis=1)                      # (numel, 4)
    w = words[np.arange(numel), (e & np.uint64(3)).astype(np.int64)]
    return ((w >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24))


def dropout_keep_mask_nchw(shape_nchw, p, seed, stream):
    """Boolean keep-mask in NCHW logical layout; element index is the NHWC linear index."""
    N, C, H, W = shape_nchw
    u = uniform_for_elements(N * C * H * W, seed, stream).reshape(N, H, W, C)
    return np.ascontiguousarray(np.transpose(u >= np.float32(p), (0, 3, 1, 2)))
