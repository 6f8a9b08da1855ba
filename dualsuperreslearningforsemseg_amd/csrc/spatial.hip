// Memory-bound spatial operators on pixel-major fp32 tensors: align-corners bilinear resize, global average /
// 3x3s2 max pooling, ConvTranspose2d(k2,s2), PixelShuffle, the 1x1 stride-s single-output conv of the feature
// transformers and the NCHW->pixel-major import.  Consecutive lanes always touch consecutive channels / pixels.
#include "common.h"
#include <algorithm>

namespace dsrl {

static unsigned flat_grid(long long total, int per_block = 256, int cap = 8192) {
    return (unsigned)std::max<long long>(1, std::min<long long>(ceil_div(total, per_block), cap));
}

// ---------------------------------------------------------------------------------------------- bilinear
// torch upsample_bilinear2d, align_corners=True: src = dst * (in-1)/(out-1) evaluated in fp32.
__device__ inline void ac_src(int dst, float scale, int n_in, int& i0, int& ip, float& l1) {
    const float r = scale * (float)dst;
    i0 = min((int)r, n_in - 1);
    ip = (i0 < n_in - 1) ? 1 : 0;
    l1 = r - (float)i0;
}

// I: index type of the element loop - unsigned when the tensors have fewer than 2^31 elements (host side): the three divisions per element are 32-bit ones
// (with 64-bit indices the 19-channel upsample of DSRL.py:54 took 26.8 us for 20 MB; same arithmetic per element, bit-identical results)
template <typename I>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                            int N, int H, int W, int C, int Ho, int Wo, float sh, float sw) {
    const I total = (I)N * Ho * Wo * C;
    for (I e = (I)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (I)gridDim.x * blockDim.x) {
        const int c = (int)(e % (I)C);
        I pix = e / (I)C;
        const int wo = (int)(pix % Wo); pix /= Wo;
        const int ho = (int)(pix % Ho); const int n = (int)(pix / Ho);
        int h0, hp, w0, wp; float lh, lw;
        ac_src(ho, sh, H, h0, hp, lh); ac_src(wo, sw, W, w0, wp, lw);
        const float* b = x + ((long long)(n * H + h0) * W + w0) * ldx + c;
        const float x00 = b[0], x01 = b[(long long)wp * ldx], x10 = b[(long long)hp * W * ldx], x11 = b[((long long)hp * W + wp) * ldx];
        const float v = (1.f - lh) * ((1.f - lw) * x00 + lw * x01) + lh * ((1.f - lw) * x10 + lw * x11);
        y[((long long)(n * Ho + ho) * Wo + wo) * ldy + c] = v;
    }
}

// four channels per thread: the index arithmetic (and, in the backward passes, the tap weights) is paid once per group of four and in 32 bits; the arithmetic
// per element is the scalar kernel's, so the results are bit-identical.  VEC: C, the pixel strides % 4 == 0, 16-byte aligned bases - one 16-byte access per
// tensor and tap.  !VEC (round 5): any width (the 19-channel upsample of DSRL.py:54) - the last group is short and the accesses are 4-byte ones.
template <bool VEC> __device__ __forceinline__ float4 ldg4(const float* p, int nv) {
    if constexpr (VEC) return *reinterpret_cast<const float4*>(p);
    float4 v = make_float4(p[0], 0.f, 0.f, 0.f);
    if (nv > 1) v.y = p[1];
    if (nv > 2) v.z = p[2];
    if (nv > 3) v.w = p[3];
    return v;
}
template <bool VEC> __device__ __forceinline__ void stg4(float* p, const float4 v, int nv) {
    if constexpr (VEC) { *reinterpret_cast<float4*>(p) = v; return; }
    p[0] = v.x;
    if (nv > 1) p[1] = v.y;
    if (nv > 2) p[2] = v.z;
    if (nv > 3) p[3] = v.w;
}
template <bool VEC>
__global__ __launch_bounds__(256) void bilinear_fwd4_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                             int N, int H, int W, int C, int C4, int Ho, int Wo, float sh, float sw) {
    const unsigned total = (unsigned)N * Ho * Wo * C4;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const unsigned q = e % C4; unsigned pix = e / C4;
        const int wo = (int)(pix % Wo); pix /= Wo;
        const int ho = (int)(pix % Ho), n = (int)(pix / Ho);
        const int nv = min(4, C - 4 * (int)q);
        int h0, hp, w0, wp; float lh, lw;
        ac_src(ho, sh, H, h0, hp, lh); ac_src(wo, sw, W, w0, wp, lw);
        const float* b = x + ((long long)(n * H + h0) * W + w0) * ldx + 4 * q;
        const float4 x00 = ldg4<VEC>(b, nv), x01 = ldg4<VEC>(b + (long long)wp * ldx, nv);
        const float4 x10 = ldg4<VEC>(b + (long long)hp * W * ldx, nv), x11 = ldg4<VEC>(b + ((long long)hp * W + wp) * ldx, nv);
        float4 v;
        v.x = (1.f - lh) * ((1.f - lw) * x00.x + lw * x01.x) + lh * ((1.f - lw) * x10.x + lw * x11.x);
        v.y = (1.f - lh) * ((1.f - lw) * x00.y + lw * x01.y) + lh * ((1.f - lw) * x10.y + lw * x11.y);
        v.z = (1.f - lh) * ((1.f - lw) * x00.z + lw * x01.z) + lh * ((1.f - lw) * x10.z + lw * x11.z);
        v.w = (1.f - lh) * ((1.f - lw) * x00.w + lw * x01.w) + lh * ((1.f - lw) * x10.w + lw * x11.w);
        stg4<VEC>(y + ((long long)(n * Ho + ho) * Wo + wo) * ldy + 4 * q, v, nv);
    }
}

__device__ inline float ac_weight(int dst, float scale, int n_in, int src) {
    int i0, ip; float l1;
    ac_src(dst, scale, n_in, i0, ip, l1);
    float w = 0.f;
    if (i0 == src) w += 1.f - l1;
    if (i0 + ip == src) w += l1;
    return w;
}
__device__ inline void ac_range(int src, float scale, int n_out, int& lo, int& hi) {
    if (scale <= 0.f) { lo = 0; hi = n_out - 1; return; }
    lo = max(0, (int)floorf((float)(src - 1) / scale) - 1);
    hi = min(n_out - 1, (int)ceilf((float)(src + 1) / scale) + 1);
}

// gather form of the backward (deterministic, no atomics), separable: first along W into tmp (N,Ho,W,C), then along H
template <typename I>
__global__ __launch_bounds__(256) void bilinear_bwd_w_kernel(const float* __restrict__ dy, int lddy, float* __restrict__ tmp,
                                                              int N, int W, int C, int Ho, int Wo, float sw) {
    const I total = (I)N * Ho * W * C;
    for (I e = (I)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (I)gridDim.x * blockDim.x) {
        const int c = (int)(e % (I)C);
        I pix = e / (I)C;
        const int w = (int)(pix % (I)W); const long long row = (long long)(pix / (I)W);      // row = n*Ho + ho
        int wlo, whi;
        ac_range(w, sw, Wo, wlo, whi);
        const float* r = dy + row * Wo * lddy + c;
        float acc = 0.f;
        for (int wo = wlo; wo <= whi; ++wo) {
            const float ww = ac_weight(wo, sw, W, w);
            if (ww != 0.f) acc += ww * r[(long long)wo * lddy];
        }
        tmp[e] = acc;
    }
}
template <typename I>
__global__ __launch_bounds__(256) void bilinear_bwd_h_kernel(const float* __restrict__ tmp, float* __restrict__ dx, int lddx,
                                                              int N, int H, int W, int C, int Ho, float sh) {
    const I total = (I)N * H * W * C;
    for (I e = (I)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (I)gridDim.x * blockDim.x) {
        const int c = (int)(e % (I)C);
        I pix = e / (I)C;
        const int w = (int)(pix % (I)W); pix /= (I)W;
        const int h = (int)(pix % (I)H); const int n = (int)(pix / (I)H);
        int hlo, hhi;
        ac_range(h, sh, Ho, hlo, hhi);
        float acc = 0.f;
        for (int ho = hlo; ho <= hhi; ++ho) {
            const float wh = ac_weight(ho, sh, H, h);
            if (wh != 0.f) acc += wh * tmp[(((long long)n * Ho + ho) * W + w) * C + c];
        }
        dx[((long long)(n * H + h) * W + w) * lddx + c] = acc;
    }
}

// tmp is (N, Ho, W, C) dense
template <bool VEC>
__global__ __launch_bounds__(256) void bilinear_bwd_w4_kernel(const float* __restrict__ dy, int lddy, float* __restrict__ tmp,
                                                               int N, int W, int C, int C4, int Ho, int Wo, float sw) {
    const unsigned total = (unsigned)N * Ho * W * C4;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const unsigned q = e % C4, pix = e / C4;
        const int w = (int)(pix % W); const unsigned row = pix / W;          // row = n*Ho + ho
        const int nv = min(4, C - 4 * (int)q);
        int wlo, whi;
        ac_range(w, sw, Wo, wlo, whi);
        const float* r = dy + (long long)row * Wo * lddy + 4 * q;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int wo = wlo; wo <= whi; ++wo) {
            const float ww = ac_weight(wo, sw, W, w);
            if (ww != 0.f) {
                const float4 v = ldg4<VEC>(r + (long long)wo * lddy, nv);
                acc.x += ww * v.x; acc.y += ww * v.y; acc.z += ww * v.z; acc.w += ww * v.w;
            }
        }
        stg4<VEC>(tmp + (long long)pix * C + 4 * q, acc, nv);
    }
}
template <bool VEC>
__global__ __launch_bounds__(256) void bilinear_bwd_h4_kernel(const float* __restrict__ tmp, float* __restrict__ dx, int lddx,
                                                               int N, int H, int W, int C, int C4, int Ho, float sh) {
    const unsigned total = (unsigned)N * H * W * C4;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const unsigned q = e % C4; unsigned pix = e / C4;
        const int w = (int)(pix % W); pix /= W;
        const int h = (int)(pix % H), n = (int)(pix / H);
        const int nv = min(4, C - 4 * (int)q);
        int hlo, hhi;
        ac_range(h, sh, Ho, hlo, hhi);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ho = hlo; ho <= hhi; ++ho) {
            const float wh = ac_weight(ho, sh, H, h);
            if (wh != 0.f) {
                const float4 v = ldg4<VEC>(tmp + (((long long)n * Ho + ho) * W + w) * C + 4 * q, nv);
                acc.x += wh * v.x; acc.y += wh * v.y; acc.z += wh * v.z; acc.w += wh * v.w;
            }
        }
        stg4<VEC>(dx + ((long long)(n * H + h) * W + w) * lddx + 4 * q, acc, nv);
    }
}

// ---------------------------------------------------------------------------------------------- pools
// one block = 64 channels x 4 pixel slots of one image (N * C/64 blocks)
__global__ __launch_bounds__(256) void gap_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int HW, int C) {
    __shared__ float sh[4][64];
    const int n = blockIdx.x, cl = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    float s = 0.f;
    if (c < C) {
#pragma unroll 4
        for (int p = slot; p < HW; p += 4) s += x[((long long)n * HW + p) * ldx + c];
    }
    sh[slot][cl] = s;
    __syncthreads();
    if (slot == 0 && c < C) y[(long long)n * C + c] = (sh[0][cl] + sh[1][cl] + sh[2][cl] + sh[3][cl]) / (float)HW;
}
__global__ __launch_bounds__(256) void gap_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int lddx, int N, int HW, int C) {
    const long long total = (long long)N * HW * C;
    const float inv = 1.f / (float)HW;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C); const long long p = e / C; const int n = (int)(p / HW);
        dx[p * lddx + c] = dy[(long long)n * C + c] * inv;
    }
}

// four channels per lane (C % 4 == 0, 16-byte aligned rows): one 16-byte store per lane instead of four 4-byte ones (21 -> 7 us on ASPP's 33 MB)
template <typename I>      // unsigned when N * HW * C4 < 2^31 (host side): 32-bit divisions
__global__ __launch_bounds__(256) void gap_bwd4_kernel(const float* __restrict__ dy, float* __restrict__ dx, int lddx, int N, int HW, int C4) {
    const I total = (I)N * HW * C4;
    const float inv = 1.f / (float)HW;
    for (I e = (I)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (I)gridDim.x * blockDim.x) {
        const int c = (int)(e % (I)C4); const long long p = (long long)(e / (I)C4); const int n = (int)((I)p / (I)HW);
        const float4 g = reinterpret_cast<const float4*>(dy)[(long long)n * C4 + c];
        *reinterpret_cast<float4*>(dx + p * lddx + 4 * c) = make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv);
    }
}

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx,
                                                           int N, int H, int W, int C, int Ho, int Wo) {
    const long long total = (long long)N * Ho * Wo * C;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        long long pix = e / C;
        const int wo = (int)(pix % Wo); pix /= Wo;
        const int ho = (int)(pix % Ho); const int n = (int)(pix / Ho);
        float best = -INFINITY; int bi = 0;
        for (int r = 0; r < 3; ++r) {
            const int h = 2 * ho - 1 + r;
            if (h < 0 || h >= H) continue;
            for (int s = 0; s < 3; ++s) {
                const int w = 2 * wo - 1 + s;
                if (w < 0 || w >= W) continue;
                const float v = x[((long long)(n * H + h) * W + w) * C + c];
                if (v > best || v != v) { best = v; bi = r * 3 + s; }       // first maximum in scan order (torch max_pool2d)
            }
        }
        y[e] = best;
        if (idx) idx[e] = (unsigned char)bi;
    }
}
// the gradient of a window goes to the tap its forward pass recorded
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const unsigned char* __restrict__ idx, const float* __restrict__ dy, float* __restrict__ dx,
                                                           int N, int H, int W, int C, int Ho, int Wo) {
    const long long total = (long long)N * H * W * C;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        long long pix = e / C;
        const int w = (int)(pix % W); pix /= W;
        const int h = (int)(pix % H); const int n = (int)(pix / H);
        float acc = 0.f;
        for (int ho = max(0, h / 2); ho <= min(Ho - 1, (h + 1) / 2); ++ho)
            for (int wo = max(0, w / 2); wo <= min(Wo - 1, (w + 1) / 2); ++wo) {
                const long long o = ((long long)(n * Ho + ho) * Wo + wo) * C + c;
                const int tap = (h - (2 * ho - 1)) * 3 + (w - (2 * wo - 1));
                if (idx[o] == tap) acc += dy[o];
            }
        dx[e] = acc;
    }
}

// four channels per thread (C % 4 == 0): float4 data, the four argmax bytes as one word; per-element arithmetic as above (bit-identical)
__global__ __launch_bounds__(256) void maxpool_fwd4_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx,
                                                            int N, int H, int W, int C4, int Ho, int Wo) {
    const unsigned total = (unsigned)N * Ho * Wo * C4;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const unsigned q = e % C4; unsigned pix = e / C4;
        const int wo = (int)(pix % Wo); pix /= Wo;
        const int ho = (int)(pix % Ho), n = (int)(pix / Ho);
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}; unsigned bi[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int h = 2 * ho - 1 + r;
            if (h < 0 || h >= H) continue;
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_) {
                const int w = 2 * wo - 1 + s_;
                if (w < 0 || w >= W) continue;
                const float4 v4 = *reinterpret_cast<const float4*>(x + (((long long)(n * H + h) * W + w) * C4 + q) * 4);
                const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (v[k] > best[k] || v[k] != v[k]) { best[k] = v[k]; bi[k] = (unsigned)(r * 3 + s_); }
            }
        }
        *reinterpret_cast<float4*>(y + 4ll * e) = make_float4(best[0], best[1], best[2], best[3]);
        if (idx) *reinterpret_cast<unsigned*>(idx + 4ll * e) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    }
}
__global__ __launch_bounds__(256) void maxpool_bwd4_kernel(const unsigned char* __restrict__ idx, const float* __restrict__ dy, float* __restrict__ dx,
                                                            int N, int H, int W, int C4, int Ho, int Wo) {
    const unsigned total = (unsigned)N * H * W * C4;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const unsigned q = e % C4; unsigned pix = e / C4;
        const int w = (int)(pix % W); pix /= W;
        const int h = (int)(pix % H), n = (int)(pix / H);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ho = max(0, h / 2); ho <= min(Ho - 1, (h + 1) / 2); ++ho)
            for (int wo = max(0, w / 2); wo <= min(Wo - 1, (w + 1) / 2); ++wo) {
                const long long o = (((long long)(n * Ho + ho) * Wo + wo) * C4 + q) * 4;
                const unsigned tap = (unsigned)((h - (2 * ho - 1)) * 3 + (w - (2 * wo - 1)));
                const unsigned ib = *reinterpret_cast<const unsigned*>(idx + o);
                const float4 g = *reinterpret_cast<const float4*>(dy + o);
                if ((ib & 255u) == tap) acc[0] += g.x;
                if (((ib >> 8) & 255u) == tap) acc[1] += g.y;
                if (((ib >> 16) & 255u) == tap) acc[2] += g.z;
                if ((ib >> 24) == tap) acc[3] += g.w;
            }
        *reinterpret_cast<float4*>(dx + 4ll * e) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// ---------------------------------------------------------------------------------------------- ConvTranspose2d k2 s2
// One block = 128 consecutive input pixels of one input row = 256 output pixels of BOTH output rows 2h, 2h+1 (round 4: the input segment is
// read once for the two rows, every global access is a 16-byte one).  The input segment, the four filter slices and the output tile of one row
// go through LDS so that all global traffic is contiguous; the tile is written back as float4 (the row segment starts on a 16-byte boundary when
// W % 4 == 0: 256 * CO floats per row segment).
template <int CI, int CO>
__global__ __launch_bounds__(256) void convt2x2_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ y, int N, int H, int W) {
    constexpr int COP = (CO + 3) & ~3;
    __shared__ __attribute__((aligned(16))) float wsh[2][2][CI][COP];      // [i][j][ci][co]
    __shared__ __attribute__((aligned(16))) float xs[128 * CI];
    __shared__ __attribute__((aligned(16))) float ys[256 * CO];
    const int Wo = 2 * W;
    const int row = blockIdx.y;                 // n*H + h
    const int wo0 = blockIdx.x * 256;
    const int npix = min(256, Wo - wo0), nin = (npix + 1) / 2;
    for (int t = threadIdx.x; t < 4 * CI * COP; t += 256) {
        const int co = t % COP, ci = (t / COP) % CI, ij = t / (COP * CI);
        wsh[ij >> 1][ij & 1][ci][co] = co < CO ? w[(ci * CO + co) * 4 + ij] : 0.f;
    }
    const float* xrow = x + ((long long)row * W + wo0 / 2) * CI;
    const bool vec = (W & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
    if (vec && ((nin * CI) & 3) == 0) {
        for (int t = threadIdx.x; t < (nin * CI) >> 2; t += 256) reinterpret_cast<float4*>(xs)[t] = reinterpret_cast<const float4*>(xrow)[t];
    } else {
        for (int t = threadIdx.x; t < nin * CI; t += 256) xs[t] = xrow[t];
    }
    __syncthreads();
    const int j = threadIdx.x & 1;
    float xin[CI];
    if ((int)threadIdx.x < npix) {
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) xin[ci] = xs[(threadIdx.x >> 1) * CI + ci];
    }
    const int n = row / H, h = row - n * H;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if ((int)threadIdx.x < npix) {
            float acc[COP];
#pragma unroll
            for (int co = 0; co < COP; ++co) acc[co] = (bias != nullptr && co < CO) ? bias[co] : 0.f;
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) {
#pragma unroll
                for (int q = 0; q < COP / 4; ++q) {
                    const float4 wv = *reinterpret_cast<const float4*>(&wsh[i][j][ci][q * 4]);
                    acc[q * 4 + 0] = fmaf(xin[ci], wv.x, acc[q * 4 + 0]);
                    acc[q * 4 + 1] = fmaf(xin[ci], wv.y, acc[q * 4 + 1]);
                    acc[q * 4 + 2] = fmaf(xin[ci], wv.z, acc[q * 4 + 2]);
                    acc[q * 4 + 3] = fmaf(xin[ci], wv.w, acc[q * 4 + 3]);
                }
            }
#pragma unroll
            for (int co = 0; co < CO; ++co) ys[threadIdx.x * CO + co] = acc[co];
        }
        __syncthreads();
        float* yrow = y + ((long long)((n * 2 * H + 2 * h + i)) * Wo + wo0) * CO;
        if (vec && ((npix * CO) & 3) == 0) {
            for (int t = threadIdx.x; t < (npix * CO) >> 2; t += 256) reinterpret_cast<float4*>(yrow)[t] = reinterpret_cast<const float4*>(ys)[t];
        } else {
            for (int t = threadIdx.x; t < npix * CO; t += 256) yrow[t] = ys[t];
        }
        __syncthreads();
    }
}

// The same forward on the matrix pipe (round 4): y [px x 4*CO] = X [px x (CI + 1)] . W' [(CI + 1) x 4*CO] with column (tap, co), the extra row of W' holding
// the bias against a column of ones in X - v_mfma_f32_32x32x2_f32 (exact fp32 products).  A block walks row segments of 128 input pixels (grid-stride, the
// next segment's input in flight); a wave forms the [32 px x 4*CO] tile of its pixels in 3 x 10 MFMAs (CI = CO = 19; the filter sits in 30 registers),
// the tile goes through LDS in the [px][tap][co] order in which BOTH output rows are contiguous, and leaves as 16-byte stores.
// CE = true (dsrl_convt2x2_fwd_ce): y is the logits of nn.CrossEntropyLoss(ignore_index) and `ce.target` is known: every thread also evaluates the
// loss of two output pixels from the tile in LDS (max, exp, sum, log in ce_fused_kernel's arithmetic; NaN logit -> flag bit 0, label outside
// [0, CO) -> NaN loss + flag bit 1) while the tile is being stored: the 319 MB read of a loss pass of its own disappears.  Per-block partials
// (sum of the pixel losses, number of pixels that count) as two doubles in ce.part; ce_finalize_kernel merges them.
struct ConvtFwdCe { const unsigned char* target; double* part; int* nan_flag; int ignore_index; };
using f32x16_f = __attribute__((ext_vector_type(16))) float;
template <int CI, int CO, bool CE = false>
__global__ __launch_bounds__(256, 2) void convt2x2_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                 float* __restrict__ y, int N, int H, int W, int nseg_per_row, int nseg, ConvtFwdCe ce) {
    constexpr int TP = 128, COLS = 4 * CO, GS = COLS + 1, NK = (CI + 2) / 2, XS = 2 * NK, NJ = (COLS + 31) / 32;
    static_assert(CI + 1 <= XS && NJ <= 3 && (TP * CI) % 4 == 0 && (2 * TP * CO) % 4 == 0, "tile does not fit");
    constexpr int XV = TP * CI / 4, DV = 2 * TP * CO / 4;                   // float4 per x segment / per output row segment
    constexpr int XR = (XV + 255) / 256, DR = (DV + 255) / 256;
    __shared__ __attribute__((aligned(16))) float buf[TP * GS];            // outputs [px][tap][co], row stride 77
    __shared__ __attribute__((aligned(16))) float xa[TP * XS];             // inputs [px][ci], slot ci = CI holds 1.0 (bias row), the rest of the pad 0
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    for (int t = tid; t < TP * XS; t += 256) xa[t] = (t % XS) == CI ? 1.f : 0.f;
    float wreg[NJ][NK];                                                     // B operand: k = ci (CI: bias), n = column (tap, co)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            const int k = 2 * kk + lh, col = 32 * j + l31, tap = col / CO, co = col - tap * CO;
            wreg[j][kk] = col < COLS ? (k < CI ? w[(k * CO + co) * 4 + tap] : (k == CI && bias != nullptr ? bias[co] : 0.f)) : 0.f;
        }
    auto xword = [](int t) -> int { const int px = t / CI; return px * XS + (t - px * CI); };
    auto dword = [](int t) -> int { const int px = t / (2 * CO); return px * GS + (t - px * (2 * CO)); };
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
    float4 RX[XR];
    auto gload = [&](int seg) {
        const int row = seg / nseg_per_row;
        const int w0 = (seg - row * nseg_per_row) * TP, npx = min(TP, W - w0);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(x + ((long long)row * W + w0) * CI), 0, npx * CI * 4, 0x00020000);
#pragma unroll
        for (int k = 0; k < XR; ++k) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(tid + 256 * k < XV ? (unsigned)(tid + 256 * k) * 16u : 0x80000000u), 0, 0);
            RX[k] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
    };
    const int px0 = 32 * wv;
    double ce_loss = 0.0, ce_cnt = 0.0;
    bool ce_bad = false, ce_bad_label = false;
    int seg = (int)blockIdx.x;
    if (seg < nseg) gload(seg);
    for (; seg < nseg; seg += (int)gridDim.x) {
        const int row = seg / nseg_per_row;
        const int w0 = (seg - row * nseg_per_row) * TP, npx = min(TP, W - w0);
        const int n = row / H, h = row - n * H;
        __syncthreads();                        // the previous segment's stores (and loss reads) are done with buf
#pragma unroll
        for (int k = 0; k < XR; ++k)
            if (tid + 256 * k < XV) { const int t = 4 * (tid + 256 * k); xa[xword(t)] = RX[k].x; xa[xword(t + 1)] = RX[k].y; xa[xword(t + 2)] = RX[k].z; xa[xword(t + 3)] = RX[k].w; }
        __syncthreads();
        if (seg + (int)gridDim.x < nseg) gload(seg + (int)gridDim.x);
        const float* xp = xa + (px0 + l31) * XS + lh;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f32x16_f acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xp[2 * kk], wreg[j][kk], acc, 0, 0, 0);
            const int col = 32 * j + l31;
            if (col < COLS) {
#pragma unroll
                for (int e = 0; e < 16; ++e) buf[(px0 + (e & 3) + 8 * (e >> 2) + 4 * lh) * GS + col] = acc[e];
            }
        }
        __syncthreads();
        float* d = y + (((long long)(n * 2 * H + 2 * h)) * (2 * W) + 2 * w0) * CO;
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)d, 0, 2 * npx * CO * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(d + 2ll * W * CO), 0, 2 * npx * CO * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < DR; ++k)
                if (tid + 256 * k < DV) {
                    const float* b = buf + i * 2 * CO;
                    const int t = 4 * (tid + 256 * k);
                    const u32x4 v = {__float_as_uint(b[dword(t)]), __float_as_uint(b[dword(t + 1)]), __float_as_uint(b[dword(t + 2)]), __float_as_uint(b[dword(t + 3)])};
                    __builtin_amdgcn_raw_buffer_store_b128(v, i == 0 ? r0 : r1, (int)((unsigned)(tid + 256 * k) * 16u), 0, 0);     // past a ragged segment's end: dropped
                }
        if (CE) {
#pragma unroll 1
            for (int u = 0; u < 2; ++u) {
                const int o = tid + 256 * u, px = o >> 2, tap = o & 3;             // output pixel (2h + (tap >> 1), 2 (w0 + px) + (tap & 1))
                if (px >= npx) continue;
                const int tg = ce.target[((long long)(n * 2 * H + 2 * h + (tap >> 1))) * (2 * W) + 2 * (w0 + px) + (tap & 1)];
                const float* v = buf + px * GS + tap * CO;
                ce_bad_label |= tg != ce.ignore_index && tg >= CO;
                float m = v[0];
#pragma unroll
                for (int c = 1; c < CO; ++c) m = fmaxf(m, v[c]);
                const float vt = v[min(tg == ce.ignore_index ? 0 : tg, CO - 1)];
                float sum = 0.f;
#pragma unroll
                for (int c = 0; c < CO; ++c) sum += exp_nonpos(v[c] - m);
                ce_bad |= !(sum == sum);            // any NaN logit poisons the sum (fmaxf alone would skip it)
                if (tg != ce.ignore_index) { ce_loss += (double)(m + logf(sum) - vt); ce_cnt += 1.0; }
            }
        }
    }
    if (CE) {
        __shared__ double shd[4];
        if (ce_bad_label) ce_loss = __builtin_nan("");
        double l = wave_sum_d(ce_loss), c = wave_sum_d(ce_cnt);
        __syncthreads();
        if (lane == 0) shd[wv] = l;
        __syncthreads();
        l = shd[0] + shd[1] + shd[2] + shd[3];
        __syncthreads();
        if (lane == 0) shd[wv] = c;
        __syncthreads();
        c = shd[0] + shd[1] + shd[2] + shd[3];
        if (tid == 0) { ce.part[2 * blockIdx.x] = l; ce.part[2 * blockIdx.x + 1] = c; }
        if (ce.nan_flag && __any(ce_bad) && lane == 0) atomicOr(ce.nan_flag, 1);
        if (ce.nan_flag && __any(ce_bad_label) && lane == 0) atomicOr(ce.nan_flag, 2);
    }
}

// dx[n,h,w,ci] = sum_{i,j,co} dy[n,2h+i,2w+j,co] * w[ci,co,i,j]: one block = 128 input pixels of one row.
template <int CI, int CO>
__global__ __launch_bounds__(256) void convt2x2_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                           int N, int H, int W) {
    constexpr int CIP = (CI + 3) & ~3;
    __shared__ __attribute__((aligned(16))) float wsh[4][CO][CIP];     // [i*2+j][co][ci]
    __shared__ float dys[2][256 * CO];
    __shared__ float part[2][128 * CI];
    const int row = blockIdx.y;                 // n*H + h
    const int w0 = blockIdx.x * 128;
    const int npx = min(128, W - w0);
    for (int t = threadIdx.x; t < 4 * CO * CIP; t += 256) {
        const int ci = t % CIP, co = (t / CIP) % CO, ij = t / (CIP * CO);
        wsh[ij][co][ci] = ci < CI ? w[(ci * CO + co) * 4 + ij] : 0.f;
    }
    const int n = row / H, h = row % H;
    for (int i = 0; i < 2; ++i) {
        const float* r = dy + (((long long)(n * 2 * H + 2 * h + i)) * (2 * W) + 2 * w0) * CO;
        for (int t = threadIdx.x; t < 2 * npx * CO; t += 256) dys[i][t] = r[t];
    }
    __syncthreads();
    const int px = threadIdx.x & 127, i = threadIdx.x >> 7;
    if (px < npx) {
        float acc[CIP];
#pragma unroll
        for (int ci = 0; ci < CIP; ++ci) acc[ci] = 0.f;
        for (int j = 0; j < 2; ++j)
            for (int co = 0; co < CO; ++co) {
                const float g = dys[i][(2 * px + j) * CO + co];
#pragma unroll
                for (int q = 0; q < CIP / 4; ++q) {
                    const float4 wv = *reinterpret_cast<const float4*>(&wsh[i * 2 + j][co][q * 4]);
                    acc[q * 4 + 0] = fmaf(g, wv.x, acc[q * 4 + 0]);
                    acc[q * 4 + 1] = fmaf(g, wv.y, acc[q * 4 + 1]);
                    acc[q * 4 + 2] = fmaf(g, wv.z, acc[q * 4 + 2]);
                    acc[q * 4 + 3] = fmaf(g, wv.w, acc[q * 4 + 3]);
                }
            }
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) part[i][px * CI + ci] = acc[ci];
    }
    __syncthreads();
    float* o = dx + ((long long)row * W + w0) * CI;
    for (int t = threadIdx.x; t < npx * CI; t += 256) o[t] = part[0][t] + part[1][t];
}

// dw[ci,co,i,j] = sum_pixels x[p,ci] * dy[p@(i,j),co]; db[co] = sum dy.  Each block walks row segments of 64 input
// pixels, every thread owns a fixed set of the CI*CO*4 (+CO) outputs; block partials are merged by a second kernel.
template <int CI, int CO>
__global__ __launch_bounds__(256) void convt2x2_dw_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                           int N, int H, int W, int nseg_per_row, long long nseg) {
    constexpr int NOUT = CI * CO * 4, PER = (NOUT + CO + 255) / 256, TP = 64;
    __shared__ float xs[TP * CI];
    __shared__ float dys[TP * 4 * CO];      // [px][i][j][co]
    float acc[PER];
    int oci[PER], ocol[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        acc[k] = 0.f;
        const int o = threadIdx.x + 256 * k;
        if (o < NOUT) { oci[k] = o / (4 * CO); ocol[k] = o % (4 * CO); }       // col = (i*2+j)*CO + co
        else if (o < NOUT + CO) { oci[k] = -1; ocol[k] = o - NOUT; }            // bias gradient for channel co
        else { oci[k] = -2; ocol[k] = 0; }
    }
    for (long long seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        const long long row = seg / nseg_per_row;       // n*H + h
        const int w0 = (int)(seg % nseg_per_row) * TP;
        const int npx = min(TP, W - w0);
        const int n = (int)(row / H), h = (int)(row % H);
        __syncthreads();
        const float* xr = x + (row * W + w0) * CI;
        for (int t = threadIdx.x; t < npx * CI; t += 256) xs[t] = xr[t];
        for (int i = 0; i < 2; ++i) {
            const float* r = dy + (((long long)(n * 2 * H + 2 * h + i)) * (2 * W) + 2 * w0) * CO;
            for (int t = threadIdx.x; t < 2 * npx * CO; t += 256) {
                const int px = t / (2 * CO), rem = t % (2 * CO);
                dys[(px * 2 + i) * 2 * CO + rem] = r[t];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (oci[k] >= 0) {
                float s = 0.f;
                for (int px = 0; px < npx; ++px) s = fmaf(xs[px * CI + oci[k]], dys[px * 4 * CO + ocol[k]], s);
                acc[k] += s;
            } else if (oci[k] == -1) {
                float s = 0.f;
                for (int px = 0; px < npx; ++px)
                    for (int ij = 0; ij < 4; ++ij) s += dys[px * 4 * CO + ij * CO + ocol[k]];
                acc[k] += s;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int o = threadIdx.x + 256 * k;
        if (o < NOUT + CO) part[(long long)blockIdx.x * (NOUT + CO) + o] = acc[k];
    }
}
template <int CI, int CO>
__global__ __launch_bounds__(256) void convt2x2_dw_finalize_kernel(const float* __restrict__ part, int nblocks, float* __restrict__ dw, float* __restrict__ db) {
    constexpr int NOUT = CI * CO * 4;
    __shared__ double sh[8][32];
    const int ol = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int o = blockIdx.x * 32 + ol;
    double s = 0;
    if (o < NOUT + CO) {
        int b = sl;
        for (; b + 56 < nblocks; b += 64) {        // eight loads in flight; the order of the additions is the plain loop's
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(long long)(b + 8 * u) * (NOUT + CO) + o];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < nblocks; b += 8) s += part[(long long)b * (NOUT + CO) + o];
    }
    sh[sl][ol] = s;
    __syncthreads();
    if (sl != 0 || o >= NOUT + CO) return;
    for (int s2 = 1; s2 < 8; ++s2) s += sh[s2][ol];
    if (o < NOUT) {
        const int ci = o / (4 * CO), col = o % (4 * CO), ij = col / CO, co = col % CO;
        dw[(ci * CO + co) * 4 + ij] = (float)s;
    } else if (db) db[o - NOUT] = (float)s;
}

// Fused backward (round 2): dy is read ONCE for dx, dw and db (the separate dx / dw kernels above each stream it, and the dw kernel spends
// two LDS reads per multiply-add).  A block walks row segments of 64 input pixels (grid-stride; the next segment's 16-byte buffer loads are
// in flight while the current one is computed from LDS; rows past the end of a ragged last segment read as zeros through the descriptor):
//   * dx: thread (tap, 4 pixels, group of 5 input channels) keeps a 4 x 5 register tile: per output channel four gradient words and one
//     b128 + b32 filter read for 20 multiply-adds; the four taps of a pixel sit in one lane quad and are summed with two DPP quad permutes
//     (191 us on the 8 x 256 x 512 input of the step's last ConvTranspose; one lane per (pixel, tap) with five b128 filter reads per output channel: 206 us);
//   * dw: thread (group of 5 input channels, 4 adjacent columns of [tap][co], pixel sub-range) keeps a 5 x 4 register tile: b128 + b128 + b32
//     per 20 multiply-adds. The unused sixth-row slot ci = CI of the input tile holds 1.0, so its row of the tile is the column sum: db for free;
//   * the per-block partials keep the layout convt2x2_dw_finalize_kernel merges.
// Requires W % 4 == 0 (16-byte aligned segments); the host falls back to the two-kernel path otherwise.
__device__ __forceinline__ float quad_xor_add(float v, int which) {      // v + v of lane ^ 1 (which = 1) / lane ^ 2 (which = 2)
    const int i = __float_as_int(v);
    const int o = which == 1 ? __builtin_amdgcn_update_dpp(i, i, 0xB1, 0xF, 0xF, false) : __builtin_amdgcn_update_dpp(i, i, 0x4E, 0xF, 0xF, false);
    return v + __int_as_float(o);
}
template <int CI, int CO>
__global__ __launch_bounds__(256, 4) void convt2x2_bwd_fused_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dy,
                                                                  float* __restrict__ dx, float* __restrict__ part, int N, int H, int W,
                                                                  int nseg_per_row, int nseg) {
    constexpr int TP = 64, COLS = 4 * CO, NCOLG = CO, RG = 5, NCG = (CI + RG - 1) / RG, XA = NCG * 8;
    constexpr int TPS = NCG * NCOLG;                                        // threads per pixel sub-range of the dw tile
    constexpr int NSUB = 256 / TPS < 8 ? 256 / TPS : 8;
    static_assert(NSUB >= 1 && NCG <= 4 && CI < NCG * RG && (TP * CI) % 4 == 0 && (2 * TP * CO) % 4 == 0, "tile does not fit the block");
    constexpr int NOUT = CI * COLS;
    constexpr int XV = TP * CI / 4, DV = 2 * TP * CO / 4;                   // float4 per x segment / per dy row segment
    constexpr int XR = (XV + 255) / 256, DR = (DV + 255) / 256;             // load rounds
    constexpr int BUF = TP * COLS > NSUB * NCG * RG * COLS ? TP * COLS : NSUB * NCG * RG * COLS;
    __shared__ __attribute__((aligned(16))) float wsh[4 * CO * NCG * 8];   // [tap][co][ci group][8]
    __shared__ __attribute__((aligned(16))) float buf[BUF];                // gradients [px][tap][co]; at the end the dw tiles of the sub-ranges
    __shared__ __attribute__((aligned(16))) float xa[TP * XA];             // inputs [px][ci group][8] (slot ci = CI holds 1.0)
    const int tid = threadIdx.x;
    for (int t = tid; t < 4 * CO * NCG * 8; t += 256) {
        const int r = t % 8, g = (t / 8) % NCG, co = (t / (8 * NCG)) % CO, ij = t / (8 * NCG * CO), ci = g * RG + r;
        wsh[t] = (r < RG && ci < CI) ? w[(ci * CO + co) * 4 + ij] : 0.f;
    }
    for (int t = tid; t < TP * XA; t += 256) xa[t] = (t % XA) == (CI / RG) * 8 + CI % RG ? 1.f : 0.f;

    // per-thread constants of the staging: byte offsets of its float4 in the segment (beyond the descriptor when the round has no work for it)
    // and the four LDS word indices each float4 scatters to
    unsigned xoff[XR], doff[DR];
    int xl[XR][4], dl[DR][4];
#pragma unroll
    for (int k = 0; k < XR; ++k) {
        const int v = tid + 256 * k;
        xoff[k] = v < XV ? (unsigned)v * 16u : 0x80000000u;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int t = (4 * v + e) % (TP * CI), px = t / CI, ci = t % CI; xl[k][e] = px * XA + (ci / RG) * 8 + ci % RG; }
    }
#pragma unroll
    for (int k = 0; k < DR; ++k) {
        const int v = tid + 256 * k;
        doff[k] = v < DV ? (unsigned)v * 16u : 0x80000000u;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int t = (4 * v + e) % (2 * TP * CO), px = t / (2 * CO), rem = t % (2 * CO); dl[k][e] = px * COLS + rem; }
    }
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
    float4 RX[XR], RD[2][DR];
    auto ld4 = [](__amdgpu_buffer_rsrc_t r, unsigned off) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    auto gload = [&](int seg) {
        const int row = seg / nseg_per_row;                                  // n*H + h
        const int w0 = (seg - row * nseg_per_row) * TP, npx = min(TP, W - w0);
        const int n = row / H, h = row - n * H;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(x + ((long long)row * W + w0) * CI), 0, npx * CI * 4, 0x00020000);
        const float* d = dy + (((long long)(n * 2 * H + 2 * h)) * (2 * W) + 2 * w0) * CO;
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)d, 0, 2 * npx * CO * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(d + 2ll * W * CO), 0, 2 * npx * CO * 4, 0x00020000);
#pragma unroll
        for (int k = 0; k < XR; ++k) RX[k] = ld4(xr, xoff[k]);
#pragma unroll
        for (int k = 0; k < DR; ++k) { RD[0][k] = ld4(r0, doff[k]); RD[1][k] = ld4(r1, doff[k]); }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int k = 0; k < XR; ++k)
            if (tid + 256 * k < XV) { xa[xl[k][0]] = RX[k].x; xa[xl[k][1]] = RX[k].y; xa[xl[k][2]] = RX[k].z; xa[xl[k][3]] = RX[k].w; }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < DR; ++k)
                if (tid + 256 * k < DV) {
                    float* b = buf + i * 2 * CO;
                    b[dl[k][0]] = RD[i][k].x; b[dl[k][1]] = RD[i][k].y; b[dl[k][2]] = RD[i][k].z; b[dl[k][3]] = RD[i][k].w;
                }
    };

    float dwacc[RG][4];
#pragma unroll
    for (int r = 0; r < RG; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) dwacc[r][c] = 0.f;
    // dx: lane quad = the four taps; 16 pixel groups of 4; wave = input-channel group (waves beyond NCG idle in this phase)
    const int tap = tid & 3, pxg = (tid >> 2) & 15, ciq = tid >> 6;
    const bool dw_thread = tid < NSUB * TPS;
    const int sub = tid / TPS, cg = (tid % TPS) / NCOLG, colg = tid % NCOLG;

    int seg = (int)blockIdx.x;
    if (seg < nseg) gload(seg);
    for (; seg < nseg; seg += (int)gridDim.x) {
        const int row = seg / nseg_per_row;
        const int w0 = (seg - row * nseg_per_row) * TP, npx = min(TP, W - w0);
        __syncthreads();                        // the previous segment's readers are done
        lstore();
        __syncthreads();
        if (seg + (int)gridDim.x < nseg) gload(seg + (int)gridDim.x);
        // ---- dx
        if (ciq < NCG) {
            float acc[4][RG];
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int r = 0; r < RG; ++r) acc[p][r] = 0.f;
            const float* g = buf + (pxg * 4) * COLS + tap * CO;
            const float* wq = wsh + (tap * CO * NCG + ciq) * 8;
#pragma unroll
            for (int co = 0; co < CO; ++co) {
                const float4 w4 = *reinterpret_cast<const float4*>(wq + co * NCG * 8);
                const float w5 = wq[co * NCG * 8 + 4];
                const float wv[RG] = {w4.x, w4.y, w4.z, w4.w, w5};
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float gv = g[p * COLS + co];
#pragma unroll
                    for (int r = 0; r < RG; ++r) acc[p][r] = fmaf(gv, wv[r], acc[p][r]);
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int r = 0; r < RG; ++r) acc[p][r] = quad_xor_add(quad_xor_add(acc[p][r], 1), 2);
            // lane `tap` of the quad stores pixel pxg*4 + tap: five consecutive channels
            const int px = pxg * 4 + tap;
            if (px < npx) {
                float* o = dx + ((long long)row * W + w0 + px) * CI + ciq * RG;
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    const float v = tap == 0 ? acc[0][r] : tap == 1 ? acc[1][r] : tap == 2 ? acc[2][r] : acc[3][r];
                    if (ciq * RG + r < CI) o[r] = v;
                }
            }
        }
        // ---- dw (+ db through the ones slot)
        if (dw_thread) {
#pragma unroll 2
            for (int px = sub; px < TP; px += NSUB) {
                const float4 dv = *reinterpret_cast<const float4*>(buf + px * COLS + colg * 4);
                const float4 x4 = *reinterpret_cast<const float4*>(xa + px * XA + cg * 8);
                const float x5 = xa[px * XA + cg * 8 + 4];
                const float xv[RG] = {x4.x, x4.y, x4.z, x4.w, x5};
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    dwacc[r][0] = fmaf(xv[r], dv.x, dwacc[r][0]); dwacc[r][1] = fmaf(xv[r], dv.y, dwacc[r][1]);
                    dwacc[r][2] = fmaf(xv[r], dv.z, dwacc[r][2]); dwacc[r][3] = fmaf(xv[r], dv.w, dwacc[r][3]);
                }
            }
        }
    }
    // ---- merge the pixel sub-ranges (fixed order) and leave the block's partials; row CI of the tile = column sums of dy = db per tap
    __syncthreads();
    if (dw_thread) {
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) buf[sub * (NCG * RG * COLS) + (cg * RG + r) * COLS + colg * 4 + c] = dwacc[r][c];
    }
    __syncthreads();
    float* po = part + (long long)blockIdx.x * (NOUT + CO);
    for (int o = tid; o < NOUT + CO; o += 256) {
        float s = 0.f;
        if (o < NOUT) {
#pragma unroll
            for (int u = 0; u < NSUB; ++u) s += buf[u * (NCG * RG * COLS) + o];
        } else {
#pragma unroll
            for (int u = 0; u < NSUB; ++u)
#pragma unroll
                for (int ij = 0; ij < 4; ++ij) s += buf[u * (NCG * RG * COLS) + CI * COLS + ij * CO + (o - NOUT)];
        }
        po[o] = s;
    }
}

// The same backward on the matrix pipe (round 4).  Per input pixel the VALU kernel above issues 2 x 1444 multiply-adds and sits at 0.28 of the HBM rate
// (191 us for 478 MB).  Both halves are small GEMMs with the pixels as the long dimension:
//   dx [px x CI]        = G [px x 4*CO] . Wt [4*CO x CI]      G = the pixel's four output gradients, column (tap, co)
//   dw [CI (+1) x 4*CO] = X^T [CI (+1) x px] . G [px x 4*CO]  row CI of X^T is all ones: its row of the product is db per tap
// and v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate - the arithmetic class of the VALU kernel) runs them at 4096 flop per 64 cycles and
// SIMD.  A block walks row segments of 128 input pixels (grid-stride, next segment's loads in flight); each of its four waves owns 32 pixels: 38 MFMAs
// for their dx tile (the transposed filter stays in 38 registers) and 16 x 3 for its share of dw, accumulated over all segments of the block.
using f32x16_t = __attribute__((ext_vector_type(16))) float;
template <int CI, int CO>
__global__ __launch_bounds__(256, 2) void convt2x2_bwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dy,
                                                                 float* __restrict__ dx, float* __restrict__ part, int N, int H, int W,
                                                                 int nseg_per_row, int nseg) {
    constexpr int TP = 128, COLS = 4 * CO, GS = COLS + 1, XS = CI + 1, NK = COLS / 2, NJ = (COLS + 31) / 32;
    static_assert(CI < 32 && COLS % 2 == 0 && NJ <= 3 && (TP * CI) % 4 == 0 && (2 * TP * CO) % 4 == 0, "tile does not fit");
    constexpr int NOUT = CI * COLS;
    constexpr int XV = TP * CI / 4, DV = 2 * TP * CO / 4;                   // float4 per x segment / per dy row segment
    constexpr int XR = (XV + 255) / 256, DR = (DV + 255) / 256;
    __shared__ __attribute__((aligned(16))) float buf[TP * GS];            // gradients [px][tap][co], row stride 77: the 32 rows of a column read 32 banks
    __shared__ __attribute__((aligned(16))) float xa[TP * XS];             // inputs [px][ci], slot ci = CI holds 1.0
    static_assert(3 * NJ * 16 * 64 <= TP * GS, "the final merge of the waves' dw tiles reuses buf");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    for (int t = tid; t < TP * XS; t += 256) xa[t] = (t % XS) == CI ? 1.f : 0.f;
    // transposed filter as the B operand of the dx GEMM: k = tap * CO + co (two per MFMA), n = ci
    float wreg[NK];
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
        const int k = 2 * kk + lh, tap = k / CO, co = k - tap * CO;
        wreg[kk] = l31 < CI ? w[(l31 * CO + co) * 4 + tap] : 0.f;
    }
    // LDS word of element t of the x segment / of a gradient row segment (formed when stored)
    auto xword = [](int t) -> int { const int px = t / CI; return px * XS + (t - px * CI); };
    auto dword = [](int t) -> int { const int px = t / (2 * CO); return px * GS + (t - px * (2 * CO)); };
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
    float4 RX[XR], RD[2][DR];
    auto ld4 = [](__amdgpu_buffer_rsrc_t r, unsigned off) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    auto gload = [&](int seg) {
        const int row = seg / nseg_per_row;                                  // n*H + h
        const int w0 = (seg - row * nseg_per_row) * TP, npx = min(TP, W - w0);
        const int n = row / H, h = row - n * H;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(x + ((long long)row * W + w0) * CI), 0, npx * CI * 4, 0x00020000);
        const float* d = dy + (((long long)(n * 2 * H + 2 * h)) * (2 * W) + 2 * w0) * CO;
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)d, 0, 2 * npx * CO * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(d + 2ll * W * CO), 0, 2 * npx * CO * 4, 0x00020000);
#pragma unroll
        for (int k = 0; k < XR; ++k) RX[k] = ld4(xr, tid + 256 * k < XV ? (unsigned)(tid + 256 * k) * 16u : 0x80000000u);
#pragma unroll
        for (int k = 0; k < DR; ++k) { const unsigned off = tid + 256 * k < DV ? (unsigned)(tid + 256 * k) * 16u : 0x80000000u; RD[0][k] = ld4(r0, off); RD[1][k] = ld4(r1, off); }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int k = 0; k < XR; ++k)
            if (tid + 256 * k < XV) { const int t = 4 * (tid + 256 * k); xa[xword(t)] = RX[k].x; xa[xword(t + 1)] = RX[k].y; xa[xword(t + 2)] = RX[k].z; xa[xword(t + 3)] = RX[k].w; }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < DR; ++k)
                if (tid + 256 * k < DV) {
                    float* b = buf + i * 2 * CO;
                    const int t = 4 * (tid + 256 * k);
                    b[dword(t)] = RD[i][k].x; b[dword(t + 1)] = RD[i][k].y; b[dword(t + 2)] = RD[i][k].z; b[dword(t + 3)] = RD[i][k].w;
                }
    };
    f32x16_t accw[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) accw[j][e] = 0.f;
    const int px0 = 32 * wv;

    int seg = (int)blockIdx.x;
    if (seg < nseg) gload(seg);
    for (; seg < nseg; seg += (int)gridDim.x) {
        const int row = seg / nseg_per_row;
        const int w0 = (seg - row * nseg_per_row) * TP, npx = min(TP, W - w0);
        __syncthreads();                        // the previous segment's readers are done
        lstore();
        __syncthreads();
        if (seg + (int)gridDim.x < nseg) gload(seg + (int)gridDim.x);
        // ---- dx tile of the wave's 32 pixels: A = gradients (row = pixel, two k per MFMA), B = transposed filter
        {
            f32x16_t acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            const float* g = buf + (px0 + l31) * GS + lh;
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g[2 * kk], wreg[kk], acc, 0, 0, 0);
            if (l31 < CI) {
                float* o = dx + ((long long)row * W + w0) * CI + l31;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int px = px0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    if (px < npx) o[(long long)px * CI] = acc[e];
                }
            }
        }
        // ---- dw (+ db through the ones row): A = inputs transposed (row = ci, k = pixel), B = gradients (k = pixel, column = (tap, co))
#pragma unroll 4
        for (int sp = 0; sp < 16; ++sp) {
            const int px = px0 + 2 * sp + lh;
            const float av = l31 <= CI ? xa[px * XS + l31] : 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int col = 32 * j + l31;
                const float bv = col < COLS ? buf[px * GS + col] : 0.f;
                accw[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accw[j], 0, 0, 0);
            }
        }
    }
    // ---- merge the four waves' dw tiles (fixed order) and leave the block's partials in the layout convt2x2_dw_finalize_kernel merges
    __syncthreads();
    if (wv > 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) buf[(((wv - 1) * NJ + j) * 16 + e) * 64 + lane] = accw[j][e];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) accw[j][e] += buf[((u * NJ + j) * 16 + e) * 64 + lane];
    }
    __syncthreads();
    if (wv == 0) {                              // D[row = ci][col = (tap, co)] -> LDS as a dense [CI + 1][COLS] matrix
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ci = (e & 3) + 8 * (e >> 2) + 4 * lh, col = 32 * j + l31;
                if (ci <= CI && col < COLS) xa[ci * COLS + col] = accw[j][e];
            }
    }
    static_assert((CI + 1) * COLS <= TP * XS, "the dw matrix is staged in xa");
    __syncthreads();
    float* po = part + (long long)blockIdx.x * (NOUT + CO);
    for (int o = tid; o < NOUT + CO; o += 256) {
        float sum;
        if (o < NOUT) sum = xa[o];
        else {
            sum = 0.f;
#pragma unroll
            for (int ij = 0; ij < 4; ++ij) sum += xa[CI * COLS + ij * CO + (o - NOUT)];
        }
        po[o] = sum;
    }
}

// ---------------------------------------------------------------------------------------------- PixelShuffle
__global__ __launch_bounds__(256) void pixel_shuffle_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int H, int W, int c, int r, int inverse) {
    // forward: dst (N,H*r,W*r,c) <- src (N,H,W,c*r*r); inverse: dst (N,H,W,c*r*r) <- src (N,H*r,W*r,c)
    const long long total = (long long)N * H * W * c * r * r;
    const int Wr = W * r, Hr = H * r, C = c * r * r;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        // e enumerates the (N,Hr,Wr,c) tensor
        const int ch = (int)(e % c);
        long long pix = e / c;
        const int wr = (int)(pix % Wr); pix /= Wr;
        const int hr = (int)(pix % Hr); const int n = (int)(pix / Hr);
        const int h = hr / r, i = hr % r, w = wr / r, j = wr % r;
        const long long lo = ((long long)(n * H + h) * W + w) * C + ch * r * r + i * r + j;
        if (inverse) dst[lo] = src[e]; else dst[e] = src[lo];
    }
}

// Tiled through LDS: a block takes 32 consecutive pixels of one (n, h) row. Both sides are then contiguous in memory - the 32 records of
// C = c*r*r floats on one side, r row segments of 32*r*c floats on the other - and the permutation happens in LDS ([pixel][channel][r*r + 1]:
// the odd channel stride keeps the c planes of a pixel on different banks). Same values as the gather kernel above, only the traffic differs.
constexpr int kPsTile = 32;
// CT, RT: the channel count / factor as compile-time constants (3, 8: the SISR decoder's PixelShuffle, DSRL.py:84) - the index arithmetic below is
// divisions and remainders by them, per element; 0, 0: any shape, taken from the arguments
template <int CT, int RT>
__global__ __launch_bounds__(256) void pixel_shuffle_tiled_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int H, int W, int c_, int r_, int inverse) {
    extern __shared__ float tile[];                               // [kPsTile][c][r*r + 1]
    const int c = CT ? CT : c_, r = RT ? RT : r_;
    const int rr = r * r, C = c * rr, cs = rr + 1, ps = c * cs;
    const int tiles_w = (W + kPsTile - 1) / kPsTile;
    const int tw = blockIdx.x % tiles_w, row = blockIdx.x / tiles_w;          // row = n*H + h
    const int w0 = tw * kPsTile, npx = min(kPsTile, W - w0);
    const int n = row / H, h = row % H;
    const long long rec = ((long long)row * W + w0) * C;                     // first record of the tile in the (N,H,W,C) tensor
    const int seg = npx * r * c;                                             // floats of one shuffled row segment
    const long long Wr = (long long)W * r;
    if (!inverse) {
        for (int t = threadIdx.x; t < npx * C; t += 256) { const int px = t / C, k = t % C; tile[px * ps + (k / rr) * cs + k % rr] = src[rec + t]; }
    } else {
        for (int i = 0; i < r; ++i) {
            const float* in = src + (((long long)(n * H + h) * r + i) * Wr + (long long)w0 * r) * c;
            for (int t = threadIdx.x; t < seg; t += 256) { const int wl = t / (r * c), rem = t % (r * c), j = rem / c, ch = rem % c; tile[wl * ps + ch * cs + i * r + j] = in[t]; }
        }
    }
    __syncthreads();
    if (!inverse) {
        for (int i = 0; i < r; ++i) {
            float* out = dst + (((long long)(n * H + h) * r + i) * Wr + (long long)w0 * r) * c;
            for (int t = threadIdx.x; t < seg; t += 256) { const int wl = t / (r * c), rem = t % (r * c), j = rem / c, ch = rem % c; out[t] = tile[wl * ps + ch * cs + i * r + j]; }
        }
    } else {
        for (int t = threadIdx.x; t < npx * C; t += 256) { const int px = t / C, k = t % C; dst[rec + t] = tile[px * ps + (k / rr) * cs + k % rr]; }
    }
}

// ---------------------------------------------------------------------------------------------- 1x1 stride-s, single output
__global__ __launch_bounds__(256) void pointwise_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                             int N, int H, int W, int C, int s, int Ho, int Wo) {
    const long long total = (long long)N * Ho * Wo;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int wo = (int)(e % Wo); const long long t = e / Wo; const int ho = (int)(t % Ho), n = (int)(t / Ho);
        const float* p = x + ((long long)(n * H + ho * s) * W + wo * s) * C;
        float acc = 0.f;
        for (int c = 0; c < C; ++c) acc = fmaf(p[c], w[c], acc);
        y[e] = acc;
    }
}
// dx on the stride grid (+= when accumulate), per-block partial dw in part[block][C]
__global__ __launch_bounds__(256) void pointwise_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dy,
                                                             float* __restrict__ dx, float* __restrict__ part, int accumulate,
                                                             int N, int H, int W, int C, int s, int Ho, int Wo) {
    extern __shared__ float sh[];           // [256][C] would be large; reduce channel by channel through [256]
    const long long total = (long long)N * Ho * Wo;
    for (int c = 0; c < C; ++c) {
        float acc = 0.f;
        const float wc = w[c];
        for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
            const int wo = (int)(e % Wo); const long long t = e / Wo; const int ho = (int)(t % Ho), n = (int)(t / Ho);
            const long long o = ((long long)(n * H + ho * s) * W + wo * s) * C + c;
            const float g = dy[e];
            acc = fmaf(g, x[o], acc);
            if (accumulate == 2) continue;          // dw only: the consumer of dy forms dx itself (dsrl_convt2x2_bwd_ce)
            if (accumulate) dx[o] += g * wc; else dx[o] = g * wc;
        }
        sh[threadIdx.x] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
        if (threadIdx.x == 0) part[(long long)blockIdx.x * C + c] = sh[0];
        __syncthreads();
    }
}
// the same for C <= 32 (the 19- and 3-channel feature transformers, DSRL.py:88-95) in ONE pass over the stride grid: a thread keeps the C running sums of
// its pixels in registers and the block reduces them once (wave shuffles, then the four waves in order) - the kernel above walks the grid once per channel
// with a block reduction each (19 rounds: 16.6 us for a 65536-pixel grid).  Same per-thread sums; the block sum is taken in another order.
template <int CMAX>
__global__ __launch_bounds__(256) void pointwise_bwd_regs_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dy,
                                                                  float* __restrict__ dx, float* __restrict__ part, int accumulate,
                                                                  int N, int H, int W, int C, int s, int Ho, int Wo) {
    __shared__ float sh[4][CMAX];
    const unsigned total = (unsigned)N * Ho * Wo;           // host: < 2^31
    float acc[CMAX], wr[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) { acc[c] = 0.f; wr[c] = c < C ? w[c] : 0.f; }
    for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
        const unsigned wo = e % (unsigned)Wo, t = e / (unsigned)Wo, ho = t % (unsigned)Ho, n = t / (unsigned)Ho;
        const long long o = ((long long)(n * H + ho * s) * W + wo * s) * C;
        const float g = dy[e];
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) {
                acc[c] = fmaf(g, x[o + c], acc[c]);
                if (accumulate != 2) { if (accumulate) dx[o + c] += g * wr[c]; else dx[o + c] = g * wr[c]; }
            }
    }
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
        const float v = wave_sum(acc[c]);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][c] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < C) part[(long long)blockIdx.x * C + threadIdx.x] = ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void pointwise_dw_finalize_kernel(const float* __restrict__ part, int nblocks, int C, float* __restrict__ dw) {
    __shared__ double sh[4];            // one block per channel
    const int c = blockIdx.x;
    double s = 0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += part[(long long)b * C + c];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) dw[c] = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
}

// ---------------------------------------------------------------------------------------------- NCHW -> pixel-major
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int HW, int Cpad) {
    const long long total = (long long)N * HW;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long n = e / HW, p = e - n * HW;
        for (int c = 0; c < Cpad; ++c) y[e * Cpad + c] = c < C ? x[(n * C + c) * HW + p] : 0.f;
    }
}

template <typename I>
__global__ __launch_bounds__(256) void pad_image_kernel(const float* __restrict__ x, long long sn, long long sc, long long sh, long long sw,
                                                         float* __restrict__ y, int N, int C, int H, int W, int Cp, int top, int left, int Hp, int Wp) {
    const I total = (I)N * Hp * Wp;
    for (I e = (I)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (I)gridDim.x * blockDim.x) {
        const int wp = (int)(e % (I)Wp); const I t = e / (I)Wp; const int hp = (int)(t % (I)Hp); const long long n = (long long)(t / (I)Hp);
        const int h = hp - top, w = wp - left;
        const bool in = h >= 0 && h < H && w >= 0 && w < W;
        for (int c = 0; c < Cp; ++c) y[(long long)e * Cp + c] = (in && c < C) ? x[n * sn + c * sc + h * sh + w * sw] : 0.f;
    }
}

}  // namespace dsrl
using namespace dsrl;

#define DSRL_PROLOGUE(cond, name)                                                   \
    DSRL_REQUIRE(cond, DSRL_E_BADARG, name ": bad arguments");                      \
    hipStream_t st = (hipStream_t)stream;                                           \
    if (int e_ = bind_stream_device(st)) return e_;

static int env_int_sp(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
static float ac_scale(int n_in, int n_out) { return n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f; }

extern "C" int dsrl_bilinear_ac_fwd(const float* x, int ldx, float* y, int ldy, int N, int H, int W, int C, int Ho, int Wo, dsrl_stream_t stream) {
    DSRL_PROLOGUE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0 && ldx >= C && ldy >= C, "bilinear_ac_fwd")
    if (C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && (long long)N * Ho * Wo * C < (1ll << 32)) {
        hipLaunchKernelGGL(bilinear_fwd4_kernel<true>, dim3(flat_grid((long long)N * Ho * Wo * (C / 4))), dim3(256), 0, st, x, ldx, y, ldy, N, H, W, C, C / 4, Ho, Wo, ac_scale(H, Ho), ac_scale(W, Wo));
        return launch_status("bilinear_fwd4_kernel");
    }
    const int Cg = (C + 3) / 4;           // any width: groups of four channels with 4-byte accesses
    if ((long long)N * Ho * Wo * C < (1ll << 31) && env_int_sp("DSRL_BILINEAR_G4", 1)) {
        hipLaunchKernelGGL(bilinear_fwd4_kernel<false>, dim3(flat_grid((long long)N * Ho * Wo * Cg)), dim3(256), 0, st, x, ldx, y, ldy, N, H, W, C, Cg, Ho, Wo, ac_scale(H, Ho), ac_scale(W, Wo));
        return launch_status("bilinear_fwd4_kernel<any width>");
    }
    if ((long long)N * Ho * Wo * C < (1ll << 31))
        hipLaunchKernelGGL(bilinear_fwd_kernel<unsigned>, dim3(flat_grid((long long)N * Ho * Wo * C)), dim3(256), 0, st, x, ldx, y, ldy, N, H, W, C, Ho, Wo, ac_scale(H, Ho), ac_scale(W, Wo));
    else
        hipLaunchKernelGGL(bilinear_fwd_kernel<long long>, dim3(flat_grid((long long)N * Ho * Wo * C)), dim3(256), 0, st, x, ldx, y, ldy, N, H, W, C, Ho, Wo, ac_scale(H, Ho), ac_scale(W, Wo));
    return launch_status("bilinear_fwd_kernel");
}
extern "C" size_t dsrl_bilinear_ac_bwd_workspace_bytes(int N, int H, int W, int C, int Ho, int Wo) { (void)H; (void)Wo; return (size_t)N * Ho * W * C * sizeof(float); }
extern "C" int dsrl_bilinear_ac_bwd(const float* dy, int lddy, float* dx, int lddx, int N, int H, int W, int C, int Ho, int Wo,
                                    void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    DSRL_PROLOGUE(dy && dx && ws && N > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0 && lddy >= C && lddx >= C, "bilinear_ac_bwd")
    DSRL_REQUIRE(ws_bytes >= dsrl_bilinear_ac_bwd_workspace_bytes(N, H, W, C, Ho, Wo), DSRL_E_WORKSPACE, "bilinear_ac_bwd: workspace too small");
    if (C % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)dx % 16) == 0 && ((uintptr_t)ws % 16) == 0 &&
        (long long)N * Ho * std::max(W, Wo) * C < (1ll << 32)) {
        hipLaunchKernelGGL(bilinear_bwd_w4_kernel<true>, dim3(flat_grid((long long)N * Ho * W * (C / 4))), dim3(256), 0, st, dy, lddy, (float*)ws, N, W, C, C / 4, Ho, Wo, ac_scale(W, Wo));
        if (int e = launch_status("bilinear_bwd_w4_kernel")) return e;
        hipLaunchKernelGGL(bilinear_bwd_h4_kernel<true>, dim3(flat_grid((long long)N * H * W * (C / 4))), dim3(256), 0, st, (const float*)ws, dx, lddx, N, H, W, C, C / 4, Ho, ac_scale(H, Ho));
        return launch_status("bilinear_bwd_h4_kernel");
    }
    if ((long long)N * std::max(H, Ho) * std::max(W, Wo) * C < (1ll << 31) && env_int_sp("DSRL_BILINEAR_G4", 1)) {
        const int Cg = (C + 3) / 4;
        hipLaunchKernelGGL(bilinear_bwd_w4_kernel<false>, dim3(flat_grid((long long)N * Ho * W * Cg)), dim3(256), 0, st, dy, lddy, (float*)ws, N, W, C, Cg, Ho, Wo, ac_scale(W, Wo));
        if (int e = launch_status("bilinear_bwd_w4_kernel<any width>")) return e;
        hipLaunchKernelGGL(bilinear_bwd_h4_kernel<false>, dim3(flat_grid((long long)N * H * W * Cg)), dim3(256), 0, st, (const float*)ws, dx, lddx, N, H, W, C, Cg, Ho, ac_scale(H, Ho));
        return launch_status("bilinear_bwd_h4_kernel<any width>");
    }
    const bool small = (long long)N * std::max(H, Ho) * std::max(W, Wo) * C < (1ll << 31);
    if (small) hipLaunchKernelGGL(bilinear_bwd_w_kernel<unsigned>, dim3(flat_grid((long long)N * Ho * W * C)), dim3(256), 0, st, dy, lddy, (float*)ws, N, W, C, Ho, Wo, ac_scale(W, Wo));
    else hipLaunchKernelGGL(bilinear_bwd_w_kernel<long long>, dim3(flat_grid((long long)N * Ho * W * C)), dim3(256), 0, st, dy, lddy, (float*)ws, N, W, C, Ho, Wo, ac_scale(W, Wo));
    if (int e = launch_status("bilinear_bwd_w_kernel")) return e;
    if (small) hipLaunchKernelGGL(bilinear_bwd_h_kernel<unsigned>, dim3(flat_grid((long long)N * H * W * C)), dim3(256), 0, st, (const float*)ws, dx, lddx, N, H, W, C, Ho, ac_scale(H, Ho));
    else hipLaunchKernelGGL(bilinear_bwd_h_kernel<long long>, dim3(flat_grid((long long)N * H * W * C)), dim3(256), 0, st, (const float*)ws, dx, lddx, N, H, W, C, Ho, ac_scale(H, Ho));
    return launch_status("bilinear_bwd_h_kernel");
}
extern "C" int dsrl_global_avgpool_fwd(const float* x, int ldx, float* y, int N, int HW, int C, dsrl_stream_t stream) {
    DSRL_PROLOGUE(x && y && N > 0 && HW > 0 && C > 0 && ldx >= C, "global_avgpool_fwd")
    hipLaunchKernelGGL(gap_fwd_kernel, dim3(N, (unsigned)ceil_div(C, 64)), dim3(256), 0, st, x, ldx, y, HW, C);
    return launch_status("gap_fwd_kernel");
}
extern "C" int dsrl_global_avgpool_bwd(const float* dy, float* dx, int lddx, int N, int HW, int C, dsrl_stream_t stream) {
    DSRL_PROLOGUE(dy && dx && N > 0 && HW > 0 && C > 0 && lddx >= C, "global_avgpool_bwd")
    if (C % 4 == 0 && lddx % 4 == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)dx % 16) == 0) {
        if ((long long)N * HW * (C / 4) < (1ll << 31)) hipLaunchKernelGGL(gap_bwd4_kernel<unsigned>, dim3(flat_grid((long long)N * HW * (C / 4))), dim3(256), 0, st, dy, dx, lddx, N, HW, C / 4);
        else hipLaunchKernelGGL(gap_bwd4_kernel<long long>, dim3(flat_grid((long long)N * HW * (C / 4))), dim3(256), 0, st, dy, dx, lddx, N, HW, C / 4);
        return launch_status("gap_bwd4_kernel");
    }
    hipLaunchKernelGGL(gap_bwd_kernel, dim3(flat_grid((long long)N * HW * C)), dim3(256), 0, st, dy, dx, lddx, N, HW, C);
    return launch_status("gap_bwd_kernel");
}
extern "C" int dsrl_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* argmax, int N, int H, int W, int C, dsrl_stream_t stream) {
    DSRL_PROLOGUE(x && y && N > 0 && H > 0 && W > 0 && C > 0, "maxpool3x3s2_fwd")
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if (C % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)argmax % 4) == 0 && (long long)N * H * W * C < (1ll << 32)) {
        hipLaunchKernelGGL(maxpool_fwd4_kernel, dim3(flat_grid((long long)N * Ho * Wo * (C / 4))), dim3(256), 0, st, x, y, argmax, N, H, W, C / 4, Ho, Wo);
        return launch_status("maxpool_fwd4_kernel");
    }
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(flat_grid((long long)N * Ho * Wo * C)), dim3(256), 0, st, x, y, argmax, N, H, W, C, Ho, Wo);
    return launch_status("maxpool_fwd_kernel");
}
extern "C" int dsrl_maxpool3x3s2_bwd(const uint8_t* argmax, const float* dy, float* dx, int N, int H, int W, int C, dsrl_stream_t stream) {
    DSRL_PROLOGUE(argmax && dy && dx && N > 0 && H > 0 && W > 0 && C > 0, "maxpool3x3s2_bwd")
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if (C % 4 == 0 && ((uintptr_t)dx % 16) == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)argmax % 4) == 0 && (long long)N * H * W * C < (1ll << 32)) {
        hipLaunchKernelGGL(maxpool_bwd4_kernel, dim3(flat_grid((long long)N * H * W * (C / 4))), dim3(256), 0, st, argmax, dy, dx, N, H, W, C / 4, Ho, Wo);
        return launch_status("maxpool_bwd4_kernel");
    }
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(flat_grid((long long)N * H * W * C)), dim3(256), 0, st, argmax, dy, dx, N, H, W, C, Ho, Wo);
    return launch_status("maxpool_bwd_kernel");
}

// ConvTranspose2d channel counts instantiated: the reference uses NUM_CLASSES -> NUM_CLASSES (19, Cityscapes).
#define DSRL_CONVT_DISPATCH(CI_, CO_, BODY)                         \
    if (Cin == CI_ && Cout == CO_) { constexpr int CI = CI_, CO = CO_; BODY }

// DSRL_CONVT_MAX_BLOCKS: test knob - fewer blocks than segments, so that small shapes exercise the grid-stride loop of the segment kernels
// (prefetch registers and LDS buffers reused across segments)
static int convt_block_cap(int dflt) { const char* v = getenv("DSRL_CONVT_MAX_BLOCKS"); const int n = v ? atoi(v) : 0; return n > 0 ? std::min(n, dflt) : dflt; }
extern "C" int dsrl_convt2x2_fwd(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin, int Cout, dsrl_stream_t stream) {
    DSRL_PROLOGUE(x && w && y && N > 0 && H > 0 && W > 0, "convt2x2_fwd")
    dim3 grid((unsigned)ceil_div(2 * W, 256), (unsigned)(N * H));          // a block writes both output rows of its input row
    {
        const char* mv = getenv("DSRL_CONVT_MFMA");
        const int nseg_per_row = (int)ceil_div(W, 128);
        const long long nseg = (long long)N * H * nseg_per_row;
        if ((!mv || atoi(mv) != 0) && W % 4 == 0 && nseg < (1ll << 31) && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0) {
            const int nb = (int)std::min<long long>(nseg, convt_block_cap(4 * 512));
            DSRL_CONVT_DISPATCH(19, 19, hipLaunchKernelGGL((convt2x2_fwd_mfma_kernel<CI, CO>), dim3(nb), dim3(256), 0, st, x, w, bias, y, N, H, W, nseg_per_row, (int)nseg, ConvtFwdCe{}); return launch_status("convt2x2_fwd_mfma_kernel");)
            DSRL_CONVT_DISPATCH(8, 8, hipLaunchKernelGGL((convt2x2_fwd_mfma_kernel<CI, CO>), dim3(nb), dim3(256), 0, st, x, w, bias, y, N, H, W, nseg_per_row, (int)nseg, ConvtFwdCe{}); return launch_status("convt2x2_fwd_mfma_kernel");)
        }
    }
    DSRL_CONVT_DISPATCH(19, 19, hipLaunchKernelGGL((convt2x2_fwd_kernel<CI, CO>), grid, dim3(256), 0, st, x, w, bias, y, N, H, W); return launch_status("convt2x2_fwd_kernel");)
    DSRL_CONVT_DISPATCH(8, 8, hipLaunchKernelGGL((convt2x2_fwd_kernel<CI, CO>), grid, dim3(256), 0, st, x, w, bias, y, N, H, W); return launch_status("convt2x2_fwd_kernel");)
    set_error("convt2x2_fwd: channel counts %d->%d not instantiated (19->19, 8->8)", Cin, Cout);
    return DSRL_E_UNSUPPORTED;
}
// ce_finalize_kernel lives in losses.hip
int launch_ce_finalize(const double* part, int nb, float* loss_out, hipStream_t st);
extern "C" int dsrl_convt2x2_fwd_ce_supported(const float* x, const float* y, int N, int H, int W, int Cin, int Cout) {
    const char* v = getenv("DSRL_CONVT_CE");
    const char* mv = getenv("DSRL_CONVT_MFMA");
    if ((v && atoi(v) == 0) || (mv && atoi(mv) == 0)) return 0;
    return (x && y && N > 0 && H > 0 && W > 0 && Cin == 19 && Cout == 19 && W % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 &&
            (long long)N * H * ceil_div(W, 128) < (1ll << 31)) ? 1 : 0;
}
extern "C" size_t dsrl_convt2x2_fwd_ce_workspace_bytes(int N, int H, int W) {
    return (size_t)std::min<long long>((long long)N * H * ceil_div(W, 128), convt_block_cap(4 * 512)) * 2 * sizeof(double);
}
extern "C" int dsrl_convt2x2_fwd_ce(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin, int Cout,
                                    const uint8_t* target, int ignore_index, float* loss_out, int* nan_flag, void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    DSRL_PROLOGUE(x && w && y && target && loss_out && ws && N > 0 && H > 0 && W > 0 && ((uintptr_t)ws % 8) == 0, "convt2x2_fwd_ce")
    DSRL_REQUIRE(dsrl_convt2x2_fwd_ce_supported(x, y, N, H, W, Cin, Cout), DSRL_E_UNSUPPORTED,
                 "convt2x2_fwd_ce: needs 19 -> 19 channels, W %% 4 == 0, 16-byte aligned tensors (got %d -> %d, W = %d) and DSRL_CONVT_MFMA / DSRL_CONVT_CE not 0", Cin, Cout, W);
    DSRL_REQUIRE(ws_bytes >= dsrl_convt2x2_fwd_ce_workspace_bytes(N, H, W), DSRL_E_WORKSPACE, "convt2x2_fwd_ce: workspace too small");
    const int nseg_per_row = (int)ceil_div(W, 128);
    const long long nseg = (long long)N * H * nseg_per_row;
    const int nb = (int)std::min<long long>(nseg, convt_block_cap(4 * 512));
    ConvtFwdCe ce{target, (double*)ws, nan_flag, ignore_index};
    hipLaunchKernelGGL((convt2x2_fwd_mfma_kernel<19, 19, true>), dim3(nb), dim3(256), 0, st, x, w, bias, y, N, H, W, nseg_per_row, (int)nseg, ce);
    if (int e = launch_status("convt2x2_fwd_mfma_kernel")) return e;
    return launch_ce_finalize((const double*)ws, nb, loss_out, st);
}
static bool env_flag_convt_fused() { const char* v = getenv("DSRL_CONVT_FUSED_BWD"); return !v || atoi(v) != 0; }   // 0: the separate dx / dw kernels
static int convt_dw_blocks(int N, int H, int W) { return (int)std::min<long long>(convt_block_cap(1024), (long long)N * H * ceil_div(W, 64)); }
extern "C" size_t dsrl_convt2x2_bwd_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
    return (size_t)convt_dw_blocks(N, H, W) * (Cin * Cout * 4 + Cout) * sizeof(float);
}
extern "C" int dsrl_convt2x2_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* dbias,
                                 int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    DSRL_PROLOGUE(x && w && dy && dx && dw && ws && N > 0 && H > 0 && W > 0, "convt2x2_bwd")
    DSRL_REQUIRE(ws_bytes >= dsrl_convt2x2_bwd_workspace_bytes(N, H, W, Cin, Cout), DSRL_E_WORKSPACE, "convt2x2_bwd: workspace too small");
    const int nb = convt_dw_blocks(N, H, W);
    const int nseg_per_row = (int)ceil_div(W, 64);
    const long long nseg = (long long)N * H * nseg_per_row;
    dim3 gdx((unsigned)ceil_div(W, 128), (unsigned)(N * H));
    const bool fused = W % 4 == 0 && nseg < (1ll << 31) && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 && env_flag_convt_fused();
    const char* mv = getenv("DSRL_CONVT_MFMA");
    const bool mfma = fused && (!mv || atoi(mv) != 0);                      // the same pass on the matrix pipe (segments of 128 pixels)
    const int nseg_per_row2 = (int)ceil_div(W, 128);
    const long long nseg2 = (long long)N * H * nseg_per_row2;
    const int nb2 = (int)std::min<long long>(nb, nseg2);
    if (mfma && convt_bwd_dma_supported(x, dy, W, Cin, Cout) && (long long)N * 4 * H * W * Cout * 4 < (1ll << 32)) {            // LDS-DMA staging, one block per CU (convt_dma.hip)
        const int nb3 = convt_bwd_dma_blocks(nseg2, nb);
        if (int e = launch_convt_bwd_dma(x, w, dy, dx, (float*)ws, N, H, W, nb3, st)) return e;
        hipLaunchKernelGGL((convt2x2_dw_finalize_kernel<19, 19>), dim3((unsigned)ceil_div(19 * 19 * 4 + 19, 32)), dim3(256), 0, st, (const float*)ws, nb3, dw, dbias);
        return launch_status("convt2x2_dw_finalize_kernel");
    }
#define DSRL_CONVT_BWD_BODY                                                                                                         \
    if (mfma) {                                                                                                                     \
        hipLaunchKernelGGL((convt2x2_bwd_mfma_kernel<CI, CO>), dim3(nb2), dim3(256), 0, st, x, w, dy, dx, (float*)ws, N, H, W, nseg_per_row2, (int)nseg2); \
        if (int e = launch_status("convt2x2_bwd_mfma_kernel")) return e;                                                            \
        hipLaunchKernelGGL((convt2x2_dw_finalize_kernel<CI, CO>), dim3((unsigned)ceil_div(CI * CO * 4 + CO, 32)), dim3(256), 0, st, \
                           (const float*)ws, nb2, dw, dbias);                                                                       \
        return launch_status("convt2x2_dw_finalize_kernel");                                                                        \
    }                                                                                                                               \
    if (fused) {                                                                                                                    \
        hipLaunchKernelGGL((convt2x2_bwd_fused_kernel<CI, CO>), dim3(nb), dim3(256), 0, st, x, w, dy, dx, (float*)ws, N, H, W, nseg_per_row, (int)nseg); \
        if (int e = launch_status("convt2x2_bwd_fused_kernel")) return e;                                                           \
        hipLaunchKernelGGL((convt2x2_dw_finalize_kernel<CI, CO>), dim3((unsigned)ceil_div(CI * CO * 4 + CO, 32)), dim3(256), 0, st, \
                           (const float*)ws, nb, dw, dbias);                                                                        \
        return launch_status("convt2x2_dw_finalize_kernel");                                                                        \
    }                                                                                                                               \
    hipLaunchKernelGGL((convt2x2_dx_kernel<CI, CO>), gdx, dim3(256), 0, st, dy, w, dx, N, H, W);                                    \
    if (int e = launch_status("convt2x2_dx_kernel")) return e;                                                                      \
    hipLaunchKernelGGL((convt2x2_dw_kernel<CI, CO>), dim3(nb), dim3(256), 0, st, x, dy, (float*)ws, N, H, W, nseg_per_row, nseg);   \
    if (int e = launch_status("convt2x2_dw_kernel")) return e;                                                                      \
    hipLaunchKernelGGL((convt2x2_dw_finalize_kernel<CI, CO>), dim3((unsigned)ceil_div(CI * CO * 4 + CO, 32)), dim3(256), 0, st,     \
                       (const float*)ws, nb, dw, dbias);                                                                            \
    return launch_status("convt2x2_dw_finalize_kernel");
    DSRL_CONVT_DISPATCH(19, 19, DSRL_CONVT_BWD_BODY)
    DSRL_CONVT_DISPATCH(8, 8, DSRL_CONVT_BWD_BODY)
    set_error("convt2x2_bwd: channel counts %d->%d not instantiated (19->19, 8->8)", Cin, Cout);
    return DSRL_E_UNSUPPORTED;
}

extern "C" int dsrl_convt2x2_bwd_ce_supported(const float* x, const float* logits, const uint8_t* target, int N, int H, int W, int Cin, int Cout) {
    const char* v = getenv("DSRL_CONVT_CE");
    if (v && atoi(v) == 0) return 0;
    return (x && logits && target && N > 0 && H > 0 && convt_bwd_dma_supported(x, logits, W, Cin, Cout) && ((uintptr_t)target % 16) == 0 &&
            (long long)N * H * (W / 128) < (1ll << 31) && (long long)N * 4 * H * W * Cout * 4 < (1ll << 32)) ? 1 : 0;
}
extern "C" int dsrl_convt2x2_bwd_ce(const float* x, const float* w, const float* logits, const uint8_t* target, int ignore_index, const float* ce_count,
                                    const float* ft_g, const float* ft_w, int ft_stride, float* dx, float* dw, float* dbias,
                                    int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    DSRL_PROLOGUE(x && w && logits && target && ce_count && dx && dw && ws && N > 0 && H > 0 && W > 0 && (!ft_g || (ft_w && ft_stride > 0)), "convt2x2_bwd_ce")
    DSRL_REQUIRE(dsrl_convt2x2_bwd_ce_supported(x, logits, target, N, H, W, Cin, Cout), DSRL_E_UNSUPPORTED,
                 "convt2x2_bwd_ce: needs 19 -> 19 channels, W %% 128 == 0, 16-byte aligned tensors (got %d -> %d, W = %d) and DSRL_CONVT_DMA / DSRL_CONVT_CE not 0", Cin, Cout, W);
    DSRL_REQUIRE(ws_bytes >= dsrl_convt2x2_bwd_workspace_bytes(N, H, W, Cin, Cout), DSRL_E_WORKSPACE, "convt2x2_bwd_ce: workspace too small");
    const long long nseg = (long long)N * H * (W / 128);
    const int nb3 = convt_bwd_dma_blocks(nseg, convt_dw_blocks(N, H, W));
    if (int e = launch_convt_bwd_dma_ce(x, w, logits, dx, (float*)ws, N, H, W, nb3, target, ignore_index, ce_count, ft_g, ft_w, ft_stride, st)) return e;
    hipLaunchKernelGGL((convt2x2_dw_finalize_kernel<19, 19>), dim3((unsigned)ceil_div(19 * 19 * 4 + 19, 32)), dim3(256), 0, st, (const float*)ws, nb3, dw, dbias);
    return launch_status("convt2x2_dw_finalize_kernel");
}

static size_t pixel_shuffle_tile_bytes(int c, int r) { return (size_t)kPsTile * c * (r * r + 1) * sizeof(float); }
static bool pixel_shuffle_tiled_ok(int N, int H, int W, int c, int r) {
    return pixel_shuffle_tile_bytes(c, r) <= 48 * 1024 && (long long)N * H * ceil_div(W, kPsTile) < (1ll << 31) && (long long)W * r * c < (1ll << 31);
}
extern "C" int dsrl_pixel_shuffle_fwd(const float* x, float* y, int N, int H, int W, int c, int r, dsrl_stream_t stream) {
    DSRL_PROLOGUE(x && y && N > 0 && H > 0 && W > 0 && c > 0 && r > 0, "pixel_shuffle_fwd")
    if (pixel_shuffle_tiled_ok(N, H, W, c, r)) {
        if (c == 3 && r == 8) hipLaunchKernelGGL((pixel_shuffle_tiled_kernel<3, 8>), dim3((unsigned)((long long)N * H * ceil_div(W, kPsTile))), dim3(256), pixel_shuffle_tile_bytes(c, r), st, x, y, N, H, W, c, r, 0);
        else hipLaunchKernelGGL((pixel_shuffle_tiled_kernel<0, 0>), dim3((unsigned)((long long)N * H * ceil_div(W, kPsTile))), dim3(256), pixel_shuffle_tile_bytes(c, r), st, x, y, N, H, W, c, r, 0);
        return launch_status("pixel_shuffle_tiled_kernel");
    }
    hipLaunchKernelGGL(pixel_shuffle_kernel, dim3(flat_grid((long long)N * H * W * c * r * r)), dim3(256), 0, st, x, y, N, H, W, c, r, 0);
    return launch_status("pixel_shuffle_kernel");
}
extern "C" int dsrl_pixel_shuffle_bwd(const float* dy, float* dx, int N, int H, int W, int c, int r, dsrl_stream_t stream) {
    DSRL_PROLOGUE(dy && dx && N > 0 && H > 0 && W > 0 && c > 0 && r > 0, "pixel_shuffle_bwd")
    if (pixel_shuffle_tiled_ok(N, H, W, c, r)) {
        if (c == 3 && r == 8) hipLaunchKernelGGL((pixel_shuffle_tiled_kernel<3, 8>), dim3((unsigned)((long long)N * H * ceil_div(W, kPsTile))), dim3(256), pixel_shuffle_tile_bytes(c, r), st, dy, dx, N, H, W, c, r, 1);
        else hipLaunchKernelGGL((pixel_shuffle_tiled_kernel<0, 0>), dim3((unsigned)((long long)N * H * ceil_div(W, kPsTile))), dim3(256), pixel_shuffle_tile_bytes(c, r), st, dy, dx, N, H, W, c, r, 1);
        return launch_status("pixel_shuffle_tiled_kernel");
    }
    hipLaunchKernelGGL(pixel_shuffle_kernel, dim3(flat_grid((long long)N * H * W * c * r * r)), dim3(256), 0, st, dy, dx, N, H, W, c, r, 1);
    return launch_status("pixel_shuffle_kernel");
}

extern "C" int dsrl_pointwise_strided_fwd(const float* x, const float* w, float* y, int N, int H, int W, int C, int stride, dsrl_stream_t stream) {
    DSRL_PROLOGUE(x && w && y && N > 0 && H > 0 && W > 0 && C > 0 && stride > 0, "pointwise_strided_fwd")
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    hipLaunchKernelGGL(pointwise_fwd_kernel, dim3(flat_grid((long long)N * Ho * Wo)), dim3(256), 0, st, x, w, y, N, H, W, C, stride, Ho, Wo);
    return launch_status("pointwise_fwd_kernel");
}
static int pointwise_blocks(int N, int H, int W, int stride) {
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    return (int)flat_grid((long long)N * Ho * Wo, 256, 256);
}
extern "C" size_t dsrl_pointwise_strided_bwd_workspace_bytes(int N, int H, int W, int C, int stride) {
    return (size_t)pointwise_blocks(N, H, W, stride) * C * sizeof(float);
}
extern "C" int dsrl_pointwise_strided_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, int accumulate,
                                          int N, int H, int W, int C, int stride, void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    DSRL_PROLOGUE(x && w && dy && (dx || accumulate == 2) && dw && ws && N > 0 && H > 0 && W > 0 && C > 0 && stride > 0 && accumulate >= 0 && accumulate <= 2,
                  "pointwise_strided_bwd")
    DSRL_REQUIRE(ws_bytes >= dsrl_pointwise_strided_bwd_workspace_bytes(N, H, W, C, stride), DSRL_E_WORKSPACE, "pointwise_strided_bwd: workspace too small");
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    if (!accumulate) {
        if (int e = launch_zero_fill(dx, (size_t)N * H * W * C * sizeof(float), st)) return e;
    }
    const int nb = pointwise_blocks(N, H, W, stride);
    if (C <= 32 && (long long)N * Ho * Wo < (1ll << 31))
        hipLaunchKernelGGL(pointwise_bwd_regs_kernel<32>, dim3(nb), dim3(256), 0, st, x, w, dy, dx, (float*)ws, accumulate, N, H, W, C, stride, Ho, Wo);
    else
        hipLaunchKernelGGL(pointwise_bwd_kernel, dim3(nb), dim3(256), 256 * sizeof(float), st, x, w, dy, dx, (float*)ws, accumulate, N, H, W, C, stride, Ho, Wo);
    if (int e = launch_status("pointwise_bwd_kernel")) return e;
    hipLaunchKernelGGL(pointwise_dw_finalize_kernel, dim3((unsigned)C), dim3(256), 0, st, (const float*)ws, nb, C, dw);
    return launch_status("pointwise_dw_finalize_kernel");
}

extern "C" int dsrl_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, int Cpad, dsrl_stream_t stream) {
    DSRL_PROLOGUE(x && y && N > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, "nchw_to_nhwc")
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(flat_grid((long long)N * H * W)), dim3(256), 0, st, x, y, N, C, H * W, Cpad);
    return launch_status("nchw_to_nhwc_kernel");
}

extern "C" int dsrl_copy2d(const float* src, int ld_src, float* dst, int ld_dst, int64_t P, int C, dsrl_stream_t stream) {
    DSRL_PROLOGUE(src && dst && P > 0 && C > 0 && ld_src >= C && ld_dst >= C, "copy2d")
    if (hipMemcpy2DAsync(dst, (size_t)ld_dst * sizeof(float), src, (size_t)ld_src * sizeof(float), (size_t)C * sizeof(float), (size_t)P,
                         hipMemcpyDeviceToDevice, st) != hipSuccess)
        return launch_status("hipMemcpy2DAsync");
    return DSRL_OK;
}

// ---- channel concatenation as ONE launch (ASPP.py:44: five 256-channel branches; DSRL.py:165: 256 + 48 channels).  Until round 5 every source was a
// hipMemcpy2DAsync (a rect-copy node per source in the step's graph) and the consumers' operand magnitude a dsrl_amax pass over the finished buffer
// (80 MB read again for the decoder's concat).  Here one kernel writes all sources into their channel ranges and leaves max |value| in the record.
constexpr int kCatMaxSources = 8;
struct CatArgs { const float* src[kCatMaxSources]; int ld4[kCatMaxSources]; int end4[kCatMaxSources]; int n; };      // float4 units; end4 = prefix sums of the widths
__global__ __launch_bounds__(256) void cat_channels_kernel(const CatArgs a, float4* __restrict__ dst, int ld_dst4, unsigned total, unsigned ctot4, unsigned* __restrict__ amax) {
    // 32-bit index arithmetic (host: fewer than 2^31 float4 per tensor span) and ONE division per thread: (pixel, column) advance by the grid stride with a
    // carry.  (A 64-bit division per element made the first version of this kernel slower than the copies it replaces.)  Four elements in flight.
    const unsigned stride = gridDim.x * 256u, dp = stride / ctot4, dq = stride - dp * ctot4;
    unsigned e = blockIdx.x * 256u + threadIdx.x;
    unsigned p = e / ctot4, q = e - p * ctot4;
    unsigned am = 0u;
    auto locate = [&](unsigned pp, unsigned qq, const float4*& sp, float4*& dp_) {
        unsigned s = 0, q0 = 0;
#pragma unroll
        for (int i = 0; i < kCatMaxSources - 1; ++i)
            if (i < a.n - 1 && qq >= (unsigned)a.end4[i]) { s = i + 1; q0 = (unsigned)a.end4[i]; }
        sp = reinterpret_cast<const float4*>(a.src[s]) + (pp * (unsigned)a.ld4[s] + (qq - q0));
        dp_ = dst + (pp * (unsigned)ld_dst4 + qq);
    };
    auto step = [&]() { e += stride; p += dp; q += dq; if (q >= ctot4) { q -= ctot4; ++p; } };
    while (e < total) {
        const float4* sp[4]; float4* dq4[4]; float4 v[4]; bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ok[u] = e < total;
            if (ok[u]) { locate(p, q, sp[u], dq4[u]); v[u] = *sp[u]; }
            step();
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ok[u]) { *dq4[u] = v[u]; am = abs_bits4(am, v[u].x, v[u].y, v[u].z, v[u].w); }
    }
    amax_publish(am, amax);
}

extern "C" int dsrl_cat_channels_supported(const float* const* srcs, const int* lds, const int* cs, int n, const float* dst, int ld_dst, int64_t P) {
    if (!srcs || !lds || !cs || n < 1 || n > kCatMaxSources || !dst || ld_dst % 4 || ((uintptr_t)dst % 16) || P <= 0 || P * (ld_dst / 4) >= (1ll << 31)) return 0;
    for (int i = 0; i < n; ++i)
        if (!srcs[i] || cs[i] <= 0 || cs[i] % 4 || lds[i] % 4 || lds[i] < cs[i] || ((uintptr_t)srcs[i] % 16) || P * (lds[i] / 4) >= (1ll << 31)) return 0;
    return 1;      // the kernel indexes float4 with 32 bits
}

extern "C" int dsrl_cat_channels(const float* const* srcs, const int* lds, const int* cs, int n, float* dst, int ld_dst, int64_t P, uint32_t* amax, dsrl_stream_t stream) {
    DSRL_PROLOGUE(dsrl_cat_channels_supported(srcs, lds, cs, n, dst, ld_dst, P), "cat_channels: 1..8 sources, widths and strides multiples of 4, 16-byte aligned, spans below 2^31 float4")
    CatArgs a{};
    int tot = 0;
    for (int i = 0; i < n; ++i) { a.src[i] = srcs[i]; a.ld4[i] = lds[i] / 4; tot += cs[i] / 4; a.end4[i] = tot; }
    a.n = n;
    DSRL_REQUIRE(4 * tot <= ld_dst, DSRL_E_BADARG, "cat_channels: %d channels do not fit the destination stride %d", 4 * tot, ld_dst);
    hipLaunchKernelGGL(cat_channels_kernel, dim3(flat_grid(P * tot, 256, 4096)), dim3(256), 0, st, a, reinterpret_cast<float4*>(dst), ld_dst / 4, (unsigned)(P * tot), (unsigned)tot,
                       (unsigned*)amax);
    return launch_status("cat_channels_kernel");
}

extern "C" int dsrl_pad_image_nhwc(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, float* y,
                                   int N, int C, int H, int W, int Cp, int top, int left, int Hp, int Wp, dsrl_stream_t stream) {
    DSRL_PROLOGUE(x && y && N > 0 && C > 0 && H > 0 && W > 0 && Cp >= C && top >= 0 && left >= 0 && Hp >= H + top && Wp >= W + left, "pad_image_nhwc")
    if ((long long)N * Hp * Wp < (1ll << 31))
        hipLaunchKernelGGL(pad_image_kernel<unsigned>, dim3(flat_grid((long long)N * Hp * Wp)), dim3(256), 0, st, x, (long long)sn, (long long)sc, (long long)sh, (long long)sw,
                           y, N, C, H, W, Cp, top, left, Hp, Wp);
    else
        hipLaunchKernelGGL(pad_image_kernel<long long>, dim3(flat_grid((long long)N * Hp * Wp)), dim3(256), 0, st, x, (long long)sn, (long long)sc, (long long)sh, (long long)sw,
                           y, N, C, H, W, Cp, top, left, Hp, Wp);
    return launch_status("pad_image_kernel");
}
