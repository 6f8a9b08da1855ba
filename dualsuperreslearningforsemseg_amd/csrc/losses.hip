// Loss kernels of the training step (train_or_resume.py:116-119, 435-438): CrossEntropy(ignore_index), MSE and the
// Feature-Affinity loss (models/losses/FALoss.py), plus the SGD update and the NaN check.  All HBM-bound or tiny.
#include "common.h"
#include <algorithm>

namespace dsrl {

__device__ inline double block_sum_d(double v, double* sh) {     // sh: 4 doubles
    v = wave_sum_d(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// ---------------------------------------------------------------------------------------------- cross entropy
// A block stages 256 pixels x C logits through LDS (contiguous global reads), one thread per pixel.
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, int ld, const unsigned char* __restrict__ target, long long P, int C,
                                                      int ignore_index, double* __restrict__ part) {
    extern __shared__ float tile[];         // [256][C]
    __shared__ double shd[4];
    double loss = 0.0, cnt = 0.0;
    for (long long p0 = (long long)blockIdx.x * 256; p0 < P; p0 += (long long)gridDim.x * 256) {
        const int np = (int)min(256ll, P - p0);
        __syncthreads();
        for (int t = threadIdx.x; t < np * C; t += 256) { const int r = t / C, c = t - r * C; tile[t] = logits[(p0 + r) * ld + c]; }
        __syncthreads();
        if ((int)threadIdx.x < np) {
            const int tg = target[p0 + threadIdx.x];
            if (tg != ignore_index) {
                const float* v = tile + threadIdx.x * C;
                float m = v[0];
                for (int c = 1; c < C; ++c) m = fmaxf(m, v[c]);
                float s = 0.f;
                for (int c = 0; c < C; ++c) s += expf(v[c] - m);
                loss += (double)(m + logf(s) - v[min(tg, C - 1)]);
                cnt += 1.0;
            }
        }
    }
    const double l = block_sum_d(loss, shd);
    const double n = block_sum_d(cnt, shd);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = l; part[2 * blockIdx.x + 1] = n; }
}
__global__ __launch_bounds__(256) void ce_finalize_kernel(const double* __restrict__ part, int nb, float* __restrict__ out) {
    __shared__ double shd[4];
    double l = 0, n = 0;
    for (int i = threadIdx.x; i < nb; i += 256) { l += part[2 * i]; n += part[2 * i + 1]; }
    l = block_sum_d(l, shd); n = block_sum_d(n, shd);
    if (threadIdx.x == 0) {
        out[0] = (float)(l / n);        // 0/0 = NaN when every pixel is ignored, as torch
        out[1] = (float)n;
    }
}
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, int ld, const unsigned char* __restrict__ target, long long P, int C,
                                                      int ignore_index, const float* __restrict__ loss_out, const float* __restrict__ grad_out,
                                                      float* __restrict__ dl, int lddl) {
    extern __shared__ float tile[];
    const float scale = grad_out[0] / loss_out[1];
    for (long long p0 = (long long)blockIdx.x * 256; p0 < P; p0 += (long long)gridDim.x * 256) {
        const int np = (int)min(256ll, P - p0);
        __syncthreads();
        for (int t = threadIdx.x; t < np * C; t += 256) { const int r = t / C, c = t - r * C; tile[t] = logits[(p0 + r) * ld + c]; }
        __syncthreads();
        if ((int)threadIdx.x < np) {
            const int tg = target[p0 + threadIdx.x];
            float* v = tile + threadIdx.x * C;
            if (tg != ignore_index) {
                float m = v[0];
                for (int c = 1; c < C; ++c) m = fmaxf(m, v[c]);
                float s = 0.f;
                for (int c = 0; c < C; ++c) s += expf(v[c] - m);
                const float inv = 1.f / s;
                for (int c = 0; c < C; ++c) v[c] = (expf(v[c] - m) * inv - (c == tg ? 1.f : 0.f)) * scale;
            } else {
                for (int c = 0; c < C; ++c) v[c] = 0.f;
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < np * C; t += 256) { const int r = t / C, c = t - r * C; dl[(p0 + r) * lddl + c] = tile[t]; }
    }
}

// ---------------------------------------------------------------------------------------------- fused loss pass
// One pass over the logits instead of two: the kernel computes the per-pixel CE terms AND writes d(CE)/d(logits) for an upstream
// gradient of 1 (the loss is the root of the backward pass), checks the logits for NaN on the way (the reference's per-output NaN
// asserts, train_or_resume.py:426-433) and stages 256 pixels x C logits through LDS with 16-byte accesses when the tensor is dense.
// The mean's denominator (pixels that are not ignored) comes from a pre-pass over the uint8 target (4 MB at 8 x 512 x 1024).
constexpr int kCountBlocks = 256;
__global__ __launch_bounds__(256) void count_valid_kernel(const unsigned char* __restrict__ target, long long P, int ignore_index, unsigned* __restrict__ part) {
    __shared__ unsigned sh[4];
    unsigned n = 0;
    // 16 labels per load; the head up to the first 16-byte boundary and the tail are counted byte by byte by block 0
    const long long head = min(P, (long long)((16 - ((uintptr_t)target & 15)) & 15));
    const long long nv = (P - head) >> 4;
    const uint4* v = reinterpret_cast<const uint4*>(target + head);
    const unsigned ig = (unsigned)ignore_index & 0xffu;
    const bool can_match = ignore_index >= 0 && ignore_index <= 255;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < nv; e += (long long)gridDim.x * 256) {
        const uint4 q = v[e];
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int b = 0; b < 4; ++b) n += (!can_match || ((w[u] >> (8 * b)) & 0xffu) != ig) ? 1u : 0u;
    }
    if (blockIdx.x == 0) {
        for (long long e = threadIdx.x; e < head; e += 256) n += (target[e] != ignore_index) ? 1u : 0u;
        for (long long e = head + (nv << 4) + threadIdx.x; e < P; e += 256) n += (target[e] != ignore_index) ? 1u : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void ce_fused_kernel(const float* __restrict__ logits, int ld, const unsigned char* __restrict__ target, long long P, int C,
                                                        int ignore_index, const unsigned* __restrict__ cnt_part, float* __restrict__ dl, int lddl,
                                                        double* __restrict__ part, int* __restrict__ nan_flag, int vec_in, int vec_out) {
    extern __shared__ __attribute__((aligned(16))) float tile[];         // [256][C]
    __shared__ double shd[4];
    __shared__ float sh_scale;
    {
        static_assert(kCountBlocks == 256, "one partial count per thread");
        const double n = block_sum_d((double)cnt_part[threadIdx.x], shd);       // exact: counts < 2^53
        if (threadIdx.x == 0) sh_scale = 1.f / (float)n;        // n = 0: every pixel ignored, the loss is 0/0 = NaN as in torch and no gradient element uses the scale
    }
    __syncthreads();
    const float scale = sh_scale;
    double loss = 0.0, cnt = 0.0;
    bool bad = false, bad_label = false;
    for (long long p0 = (long long)blockIdx.x * 256; p0 < P; p0 += (long long)gridDim.x * 256) {
        const int np = (int)min(256ll, P - p0);
        const int nf = np * C;
        __syncthreads();
        if (vec_in) {           // dense rows: 256 pixels are one contiguous run that starts on a 16-byte boundary
            const float4* src = reinterpret_cast<const float4*>(logits + p0 * C);
            for (int t = threadIdx.x; t < (nf >> 2); t += 256) reinterpret_cast<float4*>(tile)[t] = src[t];
            for (int t = (nf & ~3) + threadIdx.x; t < nf; t += 256) tile[t] = logits[p0 * C + t];
        } else {
            for (int t = threadIdx.x; t < nf; t += 256) { const int r = t / C, c = t - r * C; tile[t] = logits[(p0 + r) * ld + c]; }
        }
        __syncthreads();
        if ((int)threadIdx.x < np) {
            const int tg = target[p0 + threadIdx.x];
            bad_label |= tg != ignore_index && (tg < 0 || tg >= C);     // torch's CrossEntropyLoss asserts on such a target: here it poisons the loss and raises flag bit 1
            float* v = tile + threadIdx.x * C;
            float m = v[0];
            for (int c = 1; c < C; ++c) m = fmaxf(m, v[c]);
            const float vt = v[max(min(tg == ignore_index ? 0 : tg, C - 1), 0)];
            float s = 0.f;
            for (int c = 0; c < C; ++c) { const float e = exp_nonpos(v[c] - m); s += e; v[c] = e; }       // the tile keeps exp(v - m): one exp per logit
            bad |= !(s == s);                   // any NaN logit poisons the sum (fmaxf alone would skip it)
            if (tg != ignore_index) {
                loss += (double)(m + logf(s) - vt);
                cnt += 1.0;
                if (dl) {
                    const float inv = scale / s;
                    for (int c = 0; c < C; ++c) v[c] = v[c] * inv - (c == tg ? scale : 0.f);
                }
            } else if (dl) {
                for (int c = 0; c < C; ++c) v[c] = 0.f;
            }
        }
        if (dl) {
            __syncthreads();
            if (vec_out) {
                float4* dst = reinterpret_cast<float4*>(dl + p0 * C);
                for (int t = threadIdx.x; t < (nf >> 2); t += 256) dst[t] = reinterpret_cast<const float4*>(tile)[t];
                for (int t = (nf & ~3) + threadIdx.x; t < nf; t += 256) dl[p0 * C + t] = tile[t];
            } else {
                for (int t = threadIdx.x; t < nf; t += 256) { const int r = t / C, c = t - r * C; dl[(p0 + r) * lddl + c] = tile[t]; }
            }
        }
    }
    if (bad_label) loss = __builtin_nan("");
    const double l = block_sum_d(loss, shd);
    const double n = block_sum_d(cnt, shd);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = l; part[2 * blockIdx.x + 1] = n; }
    if (nan_flag && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(nan_flag, 1);
    if (nan_flag && __any(bad_label) && (threadIdx.x & 63) == 0) atomicOr(nan_flag, 2);
}
// MSE forward and backward in one pass: partial sums of (a-b)^2 and da = (a-b) * 2 * grad_scale / n; NaN check of `a`
__global__ __launch_bounds__(256) void mse_fused_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, float sc, float* __restrict__ da,
                                                         double* __restrict__ part, int* __restrict__ nan_flag, int vec) {
    __shared__ double shd[4];
    double s = 0.0;
    bool bad = false;
    if (vec) {
        const long long n4 = n >> 2;
        const float4* a4 = reinterpret_cast<const float4*>(a); const float4* b4 = reinterpret_cast<const float4*>(b); float4* d4 = reinterpret_cast<float4*>(da);
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long long)gridDim.x * 256) {
            const float4 x = a4[e], y = b4[e];
            const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
            bad |= !(x.x == x.x) | !(x.y == x.y) | !(x.z == x.z) | !(x.w == x.w);
            s += (double)(d0 * d0) + (double)(d1 * d1) + (double)(d2 * d2) + (double)(d3 * d3);
            if (da) d4[e] = make_float4(d0 * sc, d1 * sc, d2 * sc, d3 * sc);
        }
        for (long long e = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
            const float d = a[e] - b[e];
            bad |= !(a[e] == a[e]);
            s += (double)(d * d);
            if (da) da[e] = d * sc;
        }
    } else {
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
            const float d = a[e] - b[e];
            bad |= !(a[e] == a[e]);
            s += (double)(d * d);
            if (da) da[e] = d * sc;
        }
    }
    const double t = block_sum_d(s, shd);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
    if (nan_flag && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(nan_flag, 1);
}
// vals = {CE, w1 * MSE, w2 * FA, their sum, NaN flag}: the five scalars one iteration reads back (train_or_resume.py:435-438, 457-460)
__global__ void loss_mix_kernel(const float* __restrict__ ce, const float* __restrict__ mse, const float* __restrict__ fa, float w1, float w2,
                                const int* __restrict__ nan_flag, float* __restrict__ vals) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float c = ce[0], m = mse ? w1 * mse[0] : 0.f, f = fa ? w2 * fa[0] : 0.f;
        vals[0] = c; vals[1] = m; vals[2] = f; vals[3] = c + m + f; vals[4] = nan_flag ? (float)nan_flag[0] : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------- MSE
__global__ __launch_bounds__(256) void mse_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, double* __restrict__ part) {
    __shared__ double shd[4];
    double s = 0.0;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const float d = a[e] - b[e];
        s += (double)(d * d);
    }
    const double t = block_sum_d(s, shd);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
__global__ __launch_bounds__(256) void mse_finalize_kernel(const double* __restrict__ part, int nb, long long n, float* __restrict__ out) {
    __shared__ double shd[4];
    double s = 0;
    for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
    s = block_sum_d(s, shd);
    if (threadIdx.x == 0) out[0] = (float)(s / (double)n);
}
__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, const float* __restrict__ grad_out,
                                                       float* __restrict__ da) {
    const float sc = 2.f * grad_out[0] / (float)n;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) da[e] = (a[e] - b[e]) * sc;
}

// ---------------------------------------------------------------------------------------------- FA loss
// One block per (b,c) slice pair.  Per map: k x k average pooling -> X (h' x w'), G = X^T X in fp64, the dominant
// eigenpair of G by repeated squaring (20 squarings = power 2^20: converged for any spectral gap a float can
// resolve; the eigenvalue is then the Rayleigh quotient on the original G), sigma_1 = sqrt(lambda), S = G/lambda.
constexpr int FA_MAXW = 32, FA_MAXH = 64, FA_SQUARINGS = 20;

struct FaSmem {
    double g0[FA_MAXW * FA_MAXW];
    double ga[FA_MAXW * FA_MAXW];
    double gb[FA_MAXW * FA_MAXW];
    double red[4];
    double vec[FA_MAXW];
    float x[FA_MAXH * FA_MAXW];
    float s1[FA_MAXW * FA_MAXW];
    float s2[FA_MAXW * FA_MAXW];
    float u[FA_MAXH];
    float scal[4];
};

// k x k average pooling of one map into X (hp x wp).  The usual case (k = 8, unit column stride, 16-byte aligned rows) uses T = 1, 2, 4 or 8
// consecutive lanes per cell - as many as 256 threads allow - each summing k / T rows with two independent 16-byte loads per row (all loads
// of a thread are in flight together; the per-cell loop of 64 dependent scalar loads this replaces took ~30 us per map), the lane partials
// combined by a fixed xor tree: deterministic.
__device__ void fa_pool(const float* __restrict__ fm, long long sh_, long long sw_, int hp, int wp, int k, float* X) {
    const float inv = 1.f / (float)(k * k);
    const int ncells = hp * wp;
    if (k == 8 && sw_ == 1 && (sh_ & 3) == 0 && ((uintptr_t)fm & 15) == 0) {
        int T = 1;
        while (T < 8 && ncells * (T * 2) <= 256) T *= 2;
        const int sub = threadIdx.x & (T - 1), per = 256 / T;
        for (int c0 = 0; c0 < ncells; c0 += per) {
            const int cell = c0 + (int)(threadIdx.x / T);
            float s = 0.f;
            if (cell < ncells) {
                const int i = cell / wp, j = cell - i * wp;
                const float* base = fm + (long long)(i * 8) * sh_ + j * 8;
                float4 v[8][2];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (r < 8 / T) {
                        const float* row = base + (long long)(sub + r * T) * sh_;
                        v[r][0] = *reinterpret_cast<const float4*>(row); v[r][1] = *reinterpret_cast<const float4*>(row + 4);
                    }
                }
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    if (r < 8 / T) s += ((v[r][0].x + v[r][0].y) + (v[r][0].z + v[r][0].w)) + ((v[r][1].x + v[r][1].y) + (v[r][1].z + v[r][1].w));
            }
            for (int o = 1; o < T; o <<= 1) s += __shfl_xor(s, o, 64);
            if (cell < ncells && sub == 0) X[cell] = s * inv;
        }
    } else {
        for (int cell = threadIdx.x; cell < ncells; cell += 256) {
            const int i = cell / wp, j = cell - i * wp;
            float s = 0.f;
            for (int r = 0; r < k; ++r)
                for (int q = 0; q < k; ++q) s += fm[(long long)(i * k + r) * sh_ + (long long)(j * k + q) * sw_];
            X[cell] = s * inv;
        }
    }
    __syncthreads();
}

// wave-wide helpers on doubles: every lane of the wave takes part (wp <= 32 values sit in lanes 0..wp-1, the others carry neutral elements)
__device__ __forceinline__ double shfl_xor_d(double v, int o) {
    const long long b = __double_as_longlong(v);
    const int lo = __shfl_xor((int)(b & 0xffffffffll), o, 64), hi = __shfl_xor((int)(b >> 32), o, 64);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave32_max_d(double v) {        // all 64 lanes receive the result (lanes >= wp carry the neutral element)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, shfl_xor_d(v, o));
    return v;
}
__device__ __forceinline__ double wave32_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += shfl_xor_d(v, o);
    return v;
}

// leaves: sm.g0 = G, sm.vec = unit dominant eigenvector, returns lambda (double, uniform).  Round 3: nothing in the loop is serial any more - the
// largest diagonal entry, the best column, the norm and the Rayleigh quotient used to be loops of wp dependent LDS reads in EVERY thread (the
// kernel took 100 us for two 16x16 problems per slice); they are wave reductions over lanes 0..wp-1 now, done redundantly by each wave.
__device__ double fa_top_eig(FaSmem& sm, int hp, int wp) {
    const int nn = wp * wp, lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < nn; e += 256) {
        const int a = e / wp, b = e - a * wp;
        double s = 0.0;
        for (int i = 0; i < hp; ++i) s += (double)sm.x[i * wp + a] * (double)sm.x[i * wp + b];
        sm.g0[e] = s; sm.ga[e] = s;
    }
    __syncthreads();
    double* cur = sm.ga; double* nxt = sm.gb;
    for (int it = 0; it < FA_SQUARINGS; ++it) {
        // normalise by the largest diagonal entry (G is PSD: max |entry| sits on the diagonal)
        const double mx = wave32_max_d(lane < wp ? cur[lane * wp + lane] : 0.0);
        if (!(mx > 0.0)) break;             // all-zero (or NaN) map: lambda = 0 -> S = 0/0 = NaN exactly like the reference
        const double inv = 1.0 / mx;
        for (int e = threadIdx.x; e < nn; e += 256) {
            const int a = e / wp, b = e - a * wp;
            double s = 0.0;
            for (int q = 0; q < wp; ++q) s += (cur[a * wp + q] * inv) * (cur[q * wp + b] * inv);
            nxt[e] = s;
        }
        __syncthreads();
        double* t = cur; cur = nxt; nxt = t;
    }
    // dominant eigenvector ~ the column of G^(2^s) with the largest diagonal entry (the first such column)
    const double dg = lane < wp ? cur[lane * wp + lane] : -1.0;
    const double bd = wave32_max_d(dg);
    const unsigned long long hit = __ballot(lane < wp && dg == bd);
    const int best = hit ? __builtin_ctzll(hit) : 0;
    const double col = lane < wp ? cur[lane * wp + best] : 0.0;
    const double nrm = sqrt(wave32_sum_d(col * col));
    if ((int)threadIdx.x < wp) sm.vec[threadIdx.x] = nrm > 0.0 ? col / nrm : (threadIdx.x == 0 ? 1.0 : 0.0);
    __syncthreads();
    double row = 0.0;                        // Rayleigh quotient on the original G: lane a holds v_a (G v)_a
    if (lane < wp) {
        for (int b = 0; b < wp; ++b) row += sm.g0[lane * wp + b] * sm.vec[b];
        row *= sm.vec[lane];
    }
    return wave32_sum_d(row);
}

// One (slice, map) per block (round 3: the two maps of a slice used to run one after the other in one block): pooled map -> G = X^T X -> lambda,
// v1 -> S = G / lambda, and what the backward pass needs, into `saved` [S1 n][S2 n][sigma1, sigma2][u1 hp][v1 wp][u2 hp][v2 wp].
__global__ __launch_bounds__(256) void fa_sim_kernel(const float* __restrict__ fm1, const float* __restrict__ fm2, int C, int hp, int wp, int k,
                                                      long long sb, long long sc, long long sh_, long long sw_, float* __restrict__ saved, long long saved_stride) {
    __shared__ FaSmem sm;
    const int slice = blockIdx.x, map = blockIdx.y, b = slice / C, c = slice - b * C;
    const int n = wp * wp;
    float* sv = saved + (long long)slice * saved_stride;
    const float* fm = (map == 0 ? fm1 : fm2) + b * sb + c * sc;
    fa_pool(fm, sh_, sw_, hp, wp, k, sm.x);
    const double lam = fa_top_eig(sm, hp, wp);
    const double sigma = sqrt(lam);
    for (int e = threadIdx.x; e < n; e += 256) sv[map * n + e] = (float)(sm.g0[e] / lam);
    float* uv = sv + 2 * n + 2 + map * (hp + wp);
    for (int i = threadIdx.x; i < hp; i += 256) {
        double s = 0.0;
        for (int a = 0; a < wp; ++a) s += (double)sm.x[i * wp + a] * sm.vec[a];
        uv[i] = (float)(s / sigma);
    }
    if ((int)threadIdx.x < wp) uv[hp + threadIdx.x] = (float)sm.vec[threadIdx.x];
    if (threadIdx.x == 0) sv[2 * n + map] = (float)sigma;
}
// all pairs |S1_i - S2_j| of one slice, rows i dealt over kFaPairBlocks blocks; one fp64 partial per block (summed in a fixed order by fa_finalize_kernel)
constexpr int kFaPairBlocks = 8;
__global__ __launch_bounds__(256) void fa_pairs_kernel(const float* __restrict__ saved, long long saved_stride, int n, int reduction,
                                                        float* __restrict__ out_none, double* __restrict__ part) {
    __shared__ float s2[FA_MAXW * FA_MAXW];
    __shared__ double red[4];
    const int slice = blockIdx.x, chunk = blockIdx.y;
    const float* sv = saved + (long long)slice * saved_stride;
    for (int e = threadIdx.x; e < n; e += 256) s2[e] = sv[n + e];
    __syncthreads();
    const int per = (n + kFaPairBlocks - 1) / kFaPairBlocks, i0 = chunk * per, i1 = min(n, i0 + per);
    double acc = 0.0;
    // a row is shared by 8 lanes (j = sub, sub + 8, ...: 32 rows per pass over 256 threads), so that even n = 256 keeps every thread busy
    const int sub = threadIdx.x & 7;
    for (int i = i0 + (int)(threadIdx.x >> 3); i < i1; i += 32) {
        const float a = sv[i];
        float s = 0.f;
        if (reduction == 2) {
            float* o = out_none + ((long long)slice * n + i) * n;
            for (int j = sub; j < n; j += 8) o[j] = fabsf(a - s2[j]);
        } else {
            for (int j = sub; j < n; j += 8) s += fabsf(a - s2[j]);
        }
        acc += (double)s;
    }
    const double tot = block_sum_d(acc, red);
    if (threadIdx.x == 0) part[slice * kFaPairBlocks + chunk] = tot;
}
__global__ void fa_finalize_kernel(const double* __restrict__ part, int nslices, int n, int reduction, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0;
        for (int i = 0; i < nslices * kFaPairBlocks; ++i) s += part[i];
        out[0] = (float)(reduction == 0 ? s / ((double)nslices * n * n) : s);
    }
}

__global__ __launch_bounds__(256) void fa_bwd_kernel(const float* __restrict__ fm1, const float* __restrict__ fm2, int nslices, int C, int H, int W, int hp, int wp, int k,
                                                      long long sb, long long sc, long long sh_, long long sw_, int reduction,
                                                      const float* __restrict__ grad_out, const float* __restrict__ saved, long long saved_stride,
                                                      float* __restrict__ d1, float* __restrict__ d2) {
    __shared__ float X[FA_MAXH * FA_MAXW];
    __shared__ float S1[FA_MAXW * FA_MAXW], S2[FA_MAXW * FA_MAXW], dS[FA_MAXW * FA_MAXW];
    __shared__ float dXn[FA_MAXH * FA_MAXW];
    __shared__ double red[4];
    const int slice = blockIdx.x, b = slice / C, c = slice - b * C;
    const int n = wp * wp;
    const float* sv = saved + (long long)slice * saved_stride;
    for (int e = threadIdx.x; e < n; e += 256) { S1[e] = sv[e]; S2[e] = sv[n + e]; }
    __syncthreads();
    const float scale = grad_out[0] * (reduction == 0 ? (float)(1.0 / ((double)nslices * n * n)) : 1.f);
    {
        const int map = blockIdx.y;            // round 3: the two maps of a slice in two blocks
        // dS_map
        for (int e = threadIdx.x; e < n; e += 256) {
            int cnt = 0;
            if (map == 0) { const float a = S1[e]; for (int j = 0; j < n; ++j) { const float d = a - S2[j]; cnt += (d > 0.f) - (d < 0.f); } }
            else { const float v = S2[e]; for (int i = 0; i < n; ++i) { const float d = S1[i] - v; cnt -= (d > 0.f) - (d < 0.f); } }
            dS[e] = scale * (float)cnt;
        }
        const float* fm = (map == 0 ? fm1 : fm2) + b * sb + c * sc;
        fa_pool(fm, sh_, sw_, hp, wp, k, X);        // ends with a barrier (dS complete too)
        const float sigma = sv[2 * n + map];
        const float* u1 = sv + 2 * n + 2 + map * (hp + wp);
        const float* v1 = u1 + hp;
        // dXn = (X/sigma) (dS + dS^T);  dot = sum dXn * X
        double dot = 0.0;
        for (int e = threadIdx.x; e < hp * wp; e += 256) {
            const int i = e / wp, a = e - i * wp;
            float s = 0.f;
            for (int q = 0; q < wp; ++q) s += X[i * wp + q] * (dS[q * wp + a] + dS[a * wp + q]);
            s /= sigma;
            dXn[e] = s;
            dot += (double)s * (double)X[e];
        }
        const double tot = block_sum_d(dot, red);
        const float dsigma = (float)(-tot / ((double)sigma * (double)sigma));
        float* dst = (map == 0 ? d1 : d2) + (long long)slice * H * W;
        const float invk2 = 1.f / (float)(k * k);
        for (int e = threadIdx.x; e < H * W; e += 256) {
            const int h = e / W, w = e - h * W;
            const int i = h / k, a = w / k;
            float v = 0.f;
            if (i < hp && a < wp) v = (dXn[i * wp + a] / sigma + dsigma * u1[i] * v1[a]) * invk2;
            dst[e] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------- SGD, NaN check
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long long n,
                                                   float lr, float mom, float wd, float gscale) {
    const long long n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p); const float4* g4 = reinterpret_cast<const float4*>(g); float4* b4 = reinterpret_cast<float4*>(buf);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long long)gridDim.x * blockDim.x) {
        float4 pv = p4[e]; const float4 gv = g4[e]; float4 bv = b4[e];
        bv.x = mom * bv.x + fmaf(wd, pv.x, gv.x * gscale); pv.x -= lr * bv.x;
        bv.y = mom * bv.y + fmaf(wd, pv.y, gv.y * gscale); pv.y -= lr * bv.y;
        bv.z = mom * bv.z + fmaf(wd, pv.z, gv.z * gscale); pv.z -= lr * bv.z;
        bv.w = mom * bv.w + fmaf(wd, pv.w, gv.w * gscale); pv.w -= lr * bv.w;
        p4[e] = pv; b4[e] = bv;
    }
    for (long long e = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const float d = fmaf(wd, p[e], g[e] * gscale);
        const float bv = mom * buf[e] + d;
        buf[e] = bv; p[e] -= lr * bv;
    }
}
// the same update with the four hyper-parameters read from device memory (a captured launch follows the LR schedule without re-capture)
__global__ __launch_bounds__(256) void sgd_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long long n,
                                                       const float* __restrict__ hyper) {
    const float lr = hyper[0], mom = hyper[1], wd = hyper[2], gscale = hyper[3];
    const long long n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p); const float4* g4 = reinterpret_cast<const float4*>(g); float4* b4 = reinterpret_cast<float4*>(buf);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long long)gridDim.x * blockDim.x) {
        float4 pv = p4[e]; const float4 gv = g4[e]; float4 bv = b4[e];
        bv.x = mom * bv.x + fmaf(wd, pv.x, gv.x * gscale); pv.x -= lr * bv.x;
        bv.y = mom * bv.y + fmaf(wd, pv.y, gv.y * gscale); pv.y -= lr * bv.y;
        bv.z = mom * bv.z + fmaf(wd, pv.z, gv.z * gscale); pv.z -= lr * bv.z;
        bv.w = mom * bv.w + fmaf(wd, pv.w, gv.w * gscale); pv.w -= lr * bv.w;
        p4[e] = pv; b4[e] = bv;
    }
    for (long long e = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const float d = fmaf(wd, p[e], g[e] * gscale);
        const float bv = mom * buf[e] + d;
        buf[e] = bv; p[e] -= lr * bv;
    }
}
// The same update over a SEGMENT table that covers the arena (round 5): row {first float, floats (multiple of 4, first float 16-byte aligned), address of an
// amax record or 0}.  A segment lies inside ONE parameter; for the conv filters the block also leaves max |p| of the values it has just written in the
// filter's record (bit patterns, atomicMax: order-independent, so the record equals what dsrl_conv2d_filters_amax_batched measures) - the per-step filter pass
// then starts with the split instead of another 232 MB sweep over the parameters the optimiser wrote a moment ago.  One block per segment.
__global__ __launch_bounds__(256) void sgd_dev_segments_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                                const long long* __restrict__ table, const float* __restrict__ hyper) {
    const float lr = hyper[0], mom = hyper[1], wd = hyper[2], gscale = hyper[3];
    const long long* row = table + 3ll * blockIdx.x;
    const long long off = row[0];
    const int n4 = (int)(row[1] >> 2);
    unsigned* rec = reinterpret_cast<unsigned*>(row[2]);
    float4* p4 = reinterpret_cast<float4*>(p + off); const float4* g4 = reinterpret_cast<const float4*>(g + off); float4* b4 = reinterpret_cast<float4*>(buf + off);
    unsigned m = 0u;
    for (int e = threadIdx.x; e < n4; e += 256) {
        float4 pv = p4[e]; const float4 gv = g4[e]; float4 bv = b4[e];
        bv.x = mom * bv.x + fmaf(wd, pv.x, gv.x * gscale); pv.x -= lr * bv.x;
        bv.y = mom * bv.y + fmaf(wd, pv.y, gv.y * gscale); pv.y -= lr * bv.y;
        bv.z = mom * bv.z + fmaf(wd, pv.z, gv.z * gscale); pv.z -= lr * bv.z;
        bv.w = mom * bv.w + fmaf(wd, pv.w, gv.w * gscale); pv.w -= lr * bv.w;
        p4[e] = pv; b4[e] = bv;
        m = abs_bits4(m, pv.x, pv.y, pv.z, pv.w);
    }
    amax_publish(m, rec);           // every thread of the block reaches it; a null record publishes nothing
}
__global__ __launch_bounds__(256) void nan_check_kernel(const float* __restrict__ x, long long n, int* __restrict__ flag) {
    bool bad = false;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) bad |= (x[e] != x[e]);
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// ---------------------------------------------------------------------------------------------- validation metrics
__global__ __launch_bounds__(256) void seg_metrics_kernel(const float* __restrict__ logits, int ld, const unsigned char* __restrict__ target, long long P, int C,
                                                           int ignore_index, unsigned long long* __restrict__ counts) {
    extern __shared__ float tile[];                     // [256][C] logits, then 3*C+2 unsigned counters
    __shared__ unsigned hist[3 * 64 + 2];
    for (int t = threadIdx.x; t < 3 * C + 2; t += 256) hist[t] = 0u;
    for (long long p0 = (long long)blockIdx.x * 256; p0 < P; p0 += (long long)gridDim.x * 256) {
        const int np = (int)min(256ll, P - p0);
        __syncthreads();
        for (int t = threadIdx.x; t < np * C; t += 256) { const int r = t / C, c = t - r * C; tile[t] = logits[(p0 + r) * ld + c]; }
        __syncthreads();
        if ((int)threadIdx.x < np) {
            const int tg = target[p0 + threadIdx.x];
            if (tg != ignore_index && tg < C) {
                const float* v = tile + threadIdx.x * C;
                int best = 0; float bv = v[0];
                for (int c = 1; c < C; ++c) if (v[c] > bv) { bv = v[c]; best = c; }     // first maximum, as torch.argmax
                atomicAdd(&hist[best], 1u);
                atomicAdd(&hist[2 * C + tg], 1u);
                atomicAdd(&hist[3 * C + 1], 1u);
                if (best == tg) { atomicAdd(&hist[C + tg], 1u); atomicAdd(&hist[3 * C], 1u); }
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 3 * C + 2; t += 256) if (hist[t]) atomicAdd(&counts[t], (unsigned long long)hist[t]);
}

// ---------------------------------------------------------------------------------------------- batch preparation
__device__ inline void ac_src_f(int dst, float scale, int n_in, int& i0, int& ip, float& l1) {
    const float r = scale * (float)dst;
    i0 = min((int)r, n_in - 1); ip = (i0 < n_in - 1) ? 1 : 0; l1 = r - (float)i0;
}
__global__ __launch_bounds__(256) void prepare_image_kernel(const unsigned char* __restrict__ rgb, float* __restrict__ out, int N, int Hs, int Ws, int Ho, int Wo, int Cout,
                                                             float sh, float sw, float m0, float m1, float m2, float i0s, float i1s, float i2s) {
    const long long total = (long long)N * Ho * Wo;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int wo = (int)(e % Wo); const long long t = e / Wo; const int ho = (int)(t % Ho); const long long n = t / Ho;
        int h0, hp, w0, wp; float lh, lw;
        ac_src_f(ho, sh, Hs, h0, hp, lh); ac_src_f(wo, sw, Ws, w0, wp, lw);
        const unsigned char* b = rgb + ((n * Hs + h0) * Ws + w0) * 3;
        const long long dw_ = (long long)wp * 3, dh_ = (long long)hp * Ws * 3;
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x00 = b[c], x01 = b[dw_ + c], x10 = b[dh_ + c], x11 = b[dh_ + dw_ + c];
            v[c] = (1.f - lh) * ((1.f - lw) * x00 + lw * x01) + lh * ((1.f - lw) * x10 + lw * x11);
        }
        float* o = out + e * Cout;
        o[0] = (v[0] * (1.f / 255.f) - m0) * i0s; o[1] = (v[1] * (1.f / 255.f) - m1) * i1s; o[2] = (v[2] * (1.f / 255.f) - m2) * i2s;
        if (Cout == 4) o[3] = 0.f;
    }
}
__global__ __launch_bounds__(256) void prepare_target_kernel(const unsigned char* __restrict__ labels, const unsigned char* __restrict__ lut, unsigned char* __restrict__ target,
                                                              int N, int Hs, int Ws, int Ho, int Wo) {
    const long long total = (long long)N * Ho * Wo;
    const float sh = (float)Hs / (float)Ho, sw = (float)Ws / (float)Wo;             // torch 'nearest': src = floor(dst * in/out)
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int wo = (int)(e % Wo); const long long t = e / Wo; const int ho = (int)(t % Ho); const long long n = t / Ho;
        const int hs = min((int)floorf((float)ho * sh), Hs - 1), ws = min((int)floorf((float)wo * sw), Ws - 1);
        target[e] = lut[labels[(n * Hs + hs) * Ws + ws]];
    }
}

static int loss_blocks(long long work) { return (int)std::max<long long>(1, std::min<long long>(1024, ceil_div(work, 256))); }

}  // namespace dsrl
using namespace dsrl;

extern "C" size_t dsrl_ce_workspace_bytes(int64_t P) { return (size_t)2 * loss_blocks(P) * sizeof(double); }
extern "C" int dsrl_ce_fwd(const float* logits, int ld, const uint8_t* target, int64_t P, int C, int ignore_index, float* loss_out,
                           void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    DSRL_REQUIRE(logits && target && loss_out && ws && P > 0 && C > 0 && C <= 60 && ld >= C, DSRL_E_BADARG, "ce_fwd: bad arguments (C=%d)", C);
    DSRL_REQUIRE(ws_bytes >= dsrl_ce_workspace_bytes(P), DSRL_E_WORKSPACE, "ce_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    const int nb = loss_blocks(P);
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(nb), dim3(256), (size_t)256 * C * sizeof(float), st, logits, ld, target, (long long)P, C, ignore_index, (double*)ws);
    if (int e = launch_status("ce_fwd_kernel")) return e;
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, nb, loss_out);
    return launch_status("ce_finalize_kernel");
}
extern "C" int dsrl_ce_bwd(const float* logits, int ld, const uint8_t* target, int64_t P, int C, int ignore_index, const float* loss_out,
                           const float* grad_out, float* dlogits, int lddl, dsrl_stream_t stream) {
    DSRL_REQUIRE(logits && target && loss_out && grad_out && dlogits && P > 0 && C > 0 && C <= 60 && ld >= C && lddl >= C, DSRL_E_BADARG, "ce_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)std::min<long long>(ceil_div(P, 256), 8192)), dim3(256), (size_t)256 * C * sizeof(float), st,
                       logits, ld, target, (long long)P, C, ignore_index, loss_out, grad_out, dlogits, lddl);
    return launch_status("ce_bwd_kernel");
}

extern "C" size_t dsrl_mse_workspace_bytes(int64_t n) { return (size_t)loss_blocks(n / 4 + 1) * sizeof(double); }
extern "C" int dsrl_mse_fwd(const float* a, const float* b, int64_t n, float* loss_out, void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    DSRL_REQUIRE(a && b && loss_out && ws && n > 0, DSRL_E_BADARG, "mse_fwd: bad arguments");
    DSRL_REQUIRE(ws_bytes >= dsrl_mse_workspace_bytes(n), DSRL_E_WORKSPACE, "mse_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    const int nb = loss_blocks(n / 4 + 1);
    hipLaunchKernelGGL(mse_fwd_kernel, dim3(nb), dim3(256), 0, st, a, b, (long long)n, (double*)ws);
    if (int e = launch_status("mse_fwd_kernel")) return e;
    hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, nb, (long long)n, loss_out);
    return launch_status("mse_finalize_kernel");
}
extern "C" int dsrl_mse_bwd(const float* a, const float* b, int64_t n, const float* grad_out, float* da, dsrl_stream_t stream) {
    DSRL_REQUIRE(a && b && grad_out && da && n > 0, DSRL_E_BADARG, "mse_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    hipLaunchKernelGGL(mse_bwd_kernel, dim3((unsigned)std::min<long long>(ceil_div(n, 1024), 8192)), dim3(256), 0, st, a, b, (long long)n, grad_out, da);
    return launch_status("mse_bwd_kernel");
}

int launch_ce_finalize(const double* part, int nb, float* loss_out, hipStream_t st) {       // for dsrl_convt2x2_fwd_ce (spatial.hip)
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, st, part, nb, loss_out);
    return launch_status("ce_finalize_kernel");
}
extern "C" size_t dsrl_ce_fused_workspace_bytes(int64_t P) { return (size_t)2 * loss_blocks(P) * sizeof(double) + kCountBlocks * sizeof(unsigned); }
extern "C" int dsrl_ce_fused(const float* logits, int ld, const uint8_t* target, int64_t P, int C, int ignore_index, float* dlogits, int lddl,
                             float* loss_out, int* nan_flag, void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    DSRL_REQUIRE(logits && target && loss_out && ws && P > 0 && C > 0 && C <= 60 && ld >= C && (!dlogits || lddl >= C), DSRL_E_BADARG, "ce_fused: bad arguments (C=%d)", C);
    DSRL_REQUIRE(ws_bytes >= dsrl_ce_fused_workspace_bytes(P) && ((uintptr_t)ws % 8) == 0, DSRL_E_WORKSPACE, "ce_fused: workspace too small or misaligned");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    const int nb = loss_blocks(P);
    double* part = (double*)ws;
    unsigned* cnt = (unsigned*)(part + 2 * nb);
    hipLaunchKernelGGL(count_valid_kernel, dim3(kCountBlocks), dim3(256), 0, st, target, (long long)P, ignore_index, cnt);
    if (int e = launch_status("count_valid_kernel")) return e;
    const int vec_in = (ld == C && ((uintptr_t)logits % 16) == 0) ? 1 : 0;          // 256 * C floats per tile: every tile starts 16-byte aligned
    const int vec_out = (dlogits && lddl == C && ((uintptr_t)dlogits % 16) == 0) ? 1 : 0;
    hipLaunchKernelGGL(ce_fused_kernel, dim3(nb), dim3(256), (size_t)256 * C * sizeof(float), st, logits, ld, target, (long long)P, C, ignore_index,
                       (const unsigned*)cnt, dlogits, lddl, part, nan_flag, vec_in, vec_out);
    if (int e = launch_status("ce_fused_kernel")) return e;
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)part, nb, loss_out);
    return launch_status("ce_finalize_kernel");
}
extern "C" int dsrl_mse_fused(const float* a, const float* b, int64_t n, float grad_scale, float* da, float* loss_out, int* nan_flag,
                              void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    DSRL_REQUIRE(a && b && loss_out && ws && n > 0, DSRL_E_BADARG, "mse_fused: bad arguments");
    DSRL_REQUIRE(ws_bytes >= dsrl_mse_workspace_bytes(n), DSRL_E_WORKSPACE, "mse_fused: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    const int nb = loss_blocks(n / 4 + 1);
    const int vec = (((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0 && (!da || ((uintptr_t)da % 16) == 0)) ? 1 : 0;
    hipLaunchKernelGGL(mse_fused_kernel, dim3(nb), dim3(256), 0, st, a, b, (long long)n, 2.f * grad_scale / (float)n, da, (double*)ws, nan_flag, vec);
    if (int e = launch_status("mse_fused_kernel")) return e;
    hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, nb, (long long)n, loss_out);
    return launch_status("mse_finalize_kernel");
}
extern "C" int dsrl_loss_mix(const float* ce, const float* mse, const float* fa, float w1, float w2, const int* nan_flag, float* vals, dsrl_stream_t stream) {
    DSRL_REQUIRE(ce && vals, DSRL_E_BADARG, "loss_mix: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    hipLaunchKernelGGL(loss_mix_kernel, dim3(1), dim3(64), 0, st, ce, mse, fa, w1, w2, nan_flag, vals);
    return launch_status("loss_mix_kernel");
}

static int fa_dims(int H, int W, int k, int& hp, int& wp) {
    DSRL_REQUIRE(k > 0 && H >= k && W >= k, DSRL_E_BADARG, "fa_loss: subsample factor %d larger than the %dx%d map", k, H, W);
    hp = H / k; wp = W / k;
    DSRL_REQUIRE(hp <= FA_MAXH && wp <= FA_MAXW, DSRL_E_UNSUPPORTED, "fa_loss: pooled map %dx%d exceeds %dx%d", hp, wp, FA_MAXH, FA_MAXW);
    return DSRL_OK;
}
static long long fa_saved_stride(int hp, int wp) { return 2ll * wp * wp + 2 + 2ll * (hp + wp); }
extern "C" size_t dsrl_fa_saved_floats(int B, int C, int H, int W, int k) {
    if (k <= 0) return 0;
    return (size_t)B * C * fa_saved_stride(H / k, W / k);
}
extern "C" size_t dsrl_fa_workspace_bytes(int B, int C, int H, int W, int k) { (void)H; (void)W; (void)k; return (size_t)B * C * kFaPairBlocks * sizeof(double); }

extern "C" int dsrl_fa_fwd(const float* fm1, const float* fm2, int B, int C, int H, int W, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                           int k, int reduction, float* out, float* saved, void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    DSRL_REQUIRE(fm1 && fm2 && out && saved && ws && B > 0 && C > 0 && reduction >= 0 && reduction <= 2, DSRL_E_BADARG, "fa_fwd: bad arguments");
    int hp, wp;
    if (int e = fa_dims(H, W, k, hp, wp)) return e;
    DSRL_REQUIRE(ws_bytes >= dsrl_fa_workspace_bytes(B, C, H, W, k), DSRL_E_WORKSPACE, "fa_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    hipLaunchKernelGGL(fa_sim_kernel, dim3(B * C, 2), dim3(256), 0, st, fm1, fm2, C, hp, wp, k, (long long)sb, (long long)sc, (long long)sh, (long long)sw,
                       saved, fa_saved_stride(hp, wp));
    if (int e = launch_status("fa_sim_kernel")) return e;
    hipLaunchKernelGGL(fa_pairs_kernel, dim3(B * C, kFaPairBlocks), dim3(256), 0, st, (const float*)saved, fa_saved_stride(hp, wp), wp * wp, reduction, out, (double*)ws);
    if (int e = launch_status("fa_pairs_kernel")) return e;
    if (reduction != 2) {
        hipLaunchKernelGGL(fa_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, B * C, wp * wp, reduction, out);
        return launch_status("fa_finalize_kernel");
    }
    return DSRL_OK;
}
extern "C" int dsrl_fa_bwd(const float* fm1, const float* fm2, int B, int C, int H, int W, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                           int k, int reduction, const float* grad_out, const float* saved, float* d1, float* d2, void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    (void)ws; (void)ws_bytes;
    DSRL_REQUIRE(fm1 && fm2 && grad_out && saved && d1 && d2 && B > 0 && C > 0, DSRL_E_BADARG, "fa_bwd: bad arguments");
    DSRL_REQUIRE(reduction == 0 || reduction == 1, DSRL_E_UNSUPPORTED, "fa_bwd: only 'mean' and 'sum' reductions have a backward kernel");
    int hp, wp;
    if (int e = fa_dims(H, W, k, hp, wp)) return e;
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    hipLaunchKernelGGL(fa_bwd_kernel, dim3(B * C, 2), dim3(256), 0, st, fm1, fm2, B * C, C, H, W, hp, wp, k, (long long)sb, (long long)sc, (long long)sh, (long long)sw,
                       reduction, grad_out, saved, fa_saved_stride(hp, wp), d1, d2);
    return launch_status("fa_bwd_kernel");
}

extern "C" int dsrl_sgd_step(float* p, const float* g, float* buf, int64_t n, float lr, float momentum, float weight_decay, float grad_scale, dsrl_stream_t stream) {
    DSRL_REQUIRE(p && g && buf && n > 0, DSRL_E_BADARG, "sgd_step: bad arguments");
    DSRL_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)buf % 16) == 0, DSRL_E_BADARG, "sgd_step: arenas must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)std::min<long long>(ceil_div(n, 1024), 4096)), dim3(256), 0, st, p, g, buf, (long long)n, lr, momentum, weight_decay, grad_scale);
    return launch_status("sgd_kernel");
}
extern "C" int dsrl_sgd_step_dev(float* p, const float* g, float* buf, int64_t n, const float* hyper, dsrl_stream_t stream) {
    DSRL_REQUIRE(p && g && buf && hyper && n > 0, DSRL_E_BADARG, "sgd_step_dev: bad arguments");
    DSRL_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)buf % 16) == 0, DSRL_E_BADARG, "sgd_step_dev: arenas must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    hipLaunchKernelGGL(sgd_dev_kernel, dim3((unsigned)std::min<long long>(ceil_div(n, 1024), 4096)), dim3(256), 0, st, p, g, buf, (long long)n, hyper);
    return launch_status("sgd_dev_kernel");
}
extern "C" int dsrl_sgd_step_dev_segments(float* p, const float* g, float* buf, const int64_t* table, int64_t nseg, const float* hyper, dsrl_stream_t stream) {
    DSRL_REQUIRE(p && g && buf && table && hyper && nseg > 0 && nseg < (1ll << 31), DSRL_E_BADARG, "sgd_step_dev_segments: bad arguments");
    DSRL_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)buf % 16) == 0, DSRL_E_BADARG, "sgd_step_dev_segments: arenas must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    hipLaunchKernelGGL(sgd_dev_segments_kernel, dim3((unsigned)nseg), dim3(256), 0, st, p, g, buf, (const long long*)table, hyper);
    return launch_status("sgd_dev_segments_kernel");
}
extern "C" int dsrl_nan_check(const float* x, int64_t n, int* flag, dsrl_stream_t stream) {
    DSRL_REQUIRE(x && flag && n > 0, DSRL_E_BADARG, "nan_check: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    hipLaunchKernelGGL(nan_check_kernel, dim3((unsigned)std::min<long long>(ceil_div(n, 2048), 4096)), dim3(256), 0, st, x, (long long)n, flag);
    return launch_status("nan_check_kernel");
}

extern "C" int dsrl_seg_metrics(const float* logits, int ld, const uint8_t* target, int64_t P, int C, int ignore_index,
                                unsigned long long* counts, dsrl_stream_t stream) {
    DSRL_REQUIRE(logits && target && counts && P > 0 && C > 0 && C <= 60 && ld >= C, DSRL_E_BADARG, "seg_metrics: bad arguments (C=%d)", C);
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    hipLaunchKernelGGL(seg_metrics_kernel, dim3(loss_blocks(P)), dim3(256), (size_t)256 * C * sizeof(float), st, logits, ld, target, (long long)P, C, ignore_index, counts);
    return launch_status("seg_metrics_kernel");
}

extern "C" int dsrl_prepare_batch(const uint8_t* rgb, const uint8_t* labels, const uint8_t* lut, const float* mean, const float* std_,
                                  float* img_in, float* img_org, uint8_t* target, int N, int Hs, int Ws, int H, int W, dsrl_stream_t stream) {
    DSRL_REQUIRE(rgb && mean && std_ && N > 0 && Hs > 0 && Ws > 0 && H > 0 && W > 0, DSRL_E_BADARG, "prepare_batch: bad arguments");
    DSRL_REQUIRE((labels == nullptr) == (target == nullptr) && (labels == nullptr || lut != nullptr), DSRL_E_BADARG, "prepare_batch: labels, lut and target go together");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    auto sc = [](int n_in, int n_out) { return n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f; };
    const float m0 = mean[0], m1 = mean[1], m2 = mean[2], i0 = 1.f / std_[0], i1 = 1.f / std_[1], i2 = 1.f / std_[2];
    if (img_in) {
        hipLaunchKernelGGL(prepare_image_kernel, dim3((unsigned)std::min<long long>(ceil_div((long long)N * H * W, 256), 4096)), dim3(256), 0, st,
                           rgb, img_in, N, Hs, Ws, H, W, 4, sc(Hs, H), sc(Ws, W), m0, m1, m2, i0, i1, i2);
        if (int e = launch_status("prepare_image_kernel")) return e;
    }
    if (img_org) {
        hipLaunchKernelGGL(prepare_image_kernel, dim3((unsigned)std::min<long long>(ceil_div((long long)N * 4 * H * W, 256), 4096)), dim3(256), 0, st,
                           rgb, img_org, N, Hs, Ws, 2 * H, 2 * W, 3, sc(Hs, 2 * H), sc(Ws, 2 * W), m0, m1, m2, i0, i1, i2);
        if (int e = launch_status("prepare_image_kernel")) return e;
    }
    if (target) {
        hipLaunchKernelGGL(prepare_target_kernel, dim3((unsigned)std::min<long long>(ceil_div((long long)N * 4 * H * W, 256), 4096)), dim3(256), 0, st,
                           labels, lut, target, N, Hs, Ws, 2 * H, 2 * W);
        return launch_status("prepare_target_kernel");
    }
    return DSRL_OK;
}
