// BatchNorm2d (+residual, ReLU, Dropout), column sums and standalone Dropout on pixel-major [P][ld] fp32 tensors.
// Memory-bound: every kernel maps consecutive threads to consecutive channels of consecutive pixels (ChanMap in
// common.h), keeps its per-channel constants in registers and reduces deterministically (block partials in the
// workspace, merged by a finalize kernel in fp64; Chan/Welford merge for the variance).
#include "common.h"
#include <algorithm>
#include <initializer_list>
#include <atomic>
#include <mutex>
#include <stdlib.h>

struct SeedArg { unsigned long long value; const unsigned long long* dev; };

namespace dsrl {

constexpr int kMaxRowBlocks = 256;

static int row_blocks(int64_t P) { return (int)std::max<int64_t>(1, std::min<int64_t>(kMaxRowBlocks, ceil_div(P, 32))); }

// ---------------------------------------------------------------------------------------------- statistics
// partial[0][bx][c] = n, [1] = mean, [2] = M2 over the rows of block bx
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, int ld, long long P, int C, long long rows_per_block,
                                                          float* __restrict__ part, int nbx) {
    __shared__ float sh[3][256];
    const ChanMap m = chan_map(C, blockIdx.y);
    const long long row0 = blockIdx.x * rows_per_block, row1 = min(P, row0 + rows_per_block);
    float n = 0.f, s1 = 0.f, s2 = 0.f, k0 = 0.f;
    if (m.c >= 0) {
        long long p = row0 + m.slot;
        if (p < row1) k0 = x[p * ld + m.c];
#pragma unroll 4
        for (; p < row1; p += m.G) {
            const float d = x[p * ld + m.c] - k0;
            s1 += d; s2 += d * d; n += 1.f;
        }
    }
    float mean = 0.f, m2 = 0.f;
    if (n > 0.f) { mean = k0 + s1 / n; m2 = fmaxf(s2 - s1 * s1 / n, 0.f); }
    sh[0][threadIdx.x] = n; sh[1][threadIdx.x] = mean; sh[2][threadIdx.x] = m2;
    __syncthreads();
    if (m.c >= 0 && m.slot == 0) {
        float na = n, ma = mean, qa = m2;
        for (int g = 1; g < m.G; ++g) {
            const int t = g * m.cg + (m.c - m.cg0);
            const float nb = sh[0][t];
            if (nb > 0.f) {
                const float mb = sh[1][t], qb = sh[2][t];
                const float nt = na + nb, d = mb - ma;
                ma += d * (nb / nt);
                qa += qb + d * d * (na * nb / nt);
                na = nt;
            }
        }
        const long long o = (long long)blockIdx.x * C + m.c;
        part[o] = na; part[(long long)nbx * C + o] = ma; part[2ll * nbx * C + o] = qa;
    }
}

// 256 threads = 8 channels x 32 slices of the row blocks; a slice merges its partials in fp32 (Chan), the 32 slice results are
// merged in fp64 in a fixed order (deterministic).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int nbx, int C, float eps, float momentum,
                                   float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ rm, float* __restrict__ rv) {
    __shared__ float sh[3][32][8];
    const int cl = threadIdx.x & 7, sl = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;
    float na = 0.f, ma = 0.f, qa = 0.f;
    if (c < C) {
        float vn[8], vm[8], vq[8];             // nbx <= 256: at most 8 partials per slice, all loads issued before the merge chain
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int b = sl + 32 * j;
            const bool ok = b < nbx;
            const long long o = (long long)(ok ? b : 0) * C + c;
            vn[j] = ok ? part[o] : 0.f; vm[j] = ok ? part[(long long)nbx * C + o] : 0.f; vq[j] = ok ? part[2ll * nbx * C + o] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float nb = vn[j];
            if (nb > 0.f) {
                const float nt = na + nb, d = vm[j] - ma;
                ma += d * (nb / nt);
                qa += vq[j] + d * d * (na * nb / nt);
                na = nt;
            }
        }
    }
    sh[0][sl][cl] = na; sh[1][sl][cl] = ma; sh[2][sl][cl] = qa;
    __syncthreads();
    if (sl != 0 || c >= C) return;
    // the 32 slices relative to slice 0's mean (it always holds row block 0): N = sum n, S = sum n d, T = sum (M2 + n d^2), d = mean_s - mref - the
    // merge of the from-statistics kernels.  (Until round 5 this was Chan's pairwise update, 32 steps with two dependent fp64 divisions each: ~10 us
    // for a launch with nothing else in it.)
    const double mref = sh[1][0][cl];
    double n2 = 0, S = 0, T = 0;
#pragma unroll 8
    for (int s2 = 0; s2 < 32; ++s2) {
        const double nb = sh[0][s2][cl], d = (double)sh[1][s2][cl] - mref;
        n2 += nb; S += nb * d; T += (double)sh[2][s2][cl] + nb * d * d;
    }
    const double m2 = n2 > 0 ? mref + S / n2 : 0.0, q2 = n2 > 0 ? fmax(T - S * S / n2, 0.0) : 0.0;
    const double var = n2 > 0 ? q2 / n2 : 0.0;
    mean[c] = (float)m2;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rm) rm[c] = (float)((1.0 - momentum) * rm[c] + momentum * m2);
    if (rv) rv[c] = (float)((1.0 - momentum) * rv[c] + momentum * (n2 > 1 ? q2 / (n2 - 1) : var));
}

__global__ void invstd_from_var_kernel(const float* __restrict__ var, int C, float eps, float* __restrict__ invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) invstd[c] = (float)(1.0 / sqrt((double)var[c] + (double)eps));
}

// ---------------------------------------------------------------------------------------------- apply
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, long long P, int C,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ res, int ldr, int relu, float drop_p, SeedArg seed_arg, unsigned rng_stream,
                                                        unsigned* __restrict__ y_amax) {
    const unsigned long long seed = seed_arg.dev ? *seed_arg.dev : seed_arg.value;     // device-resident key: replayable from a graph
    const ChanMap m = chan_map(C, blockIdx.y);
    unsigned am = 0u;
    if (m.c >= 0) {
        const float sc = gamma[m.c] * invstd[m.c];
        const float sh = beta[m.c] - mean[m.c] * sc;
        const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
        for (long long p = (long long)blockIdx.x * m.G + m.slot; p < P; p += (long long)gridDim.x * m.G) {
            float v = fmaf(x[p * ldx + m.c], sc, sh);
            if (res) v += res[p * ldr + m.c];
            if (relu) v = fmaxf(v, 0.f);
            if (drop_p > 0.f) v = (philox_uniform((unsigned long long)p * C + m.c, seed, rng_stream) >= drop_p) ? v * keep_scale : 0.f;
            y[p * ldy + m.c] = v;
            am = max(am, abs_bits(v));
        }
    }
    amax_publish(am, y_amax);
}

// ---------------------------------------------------------------------------------------------- backward
__device__ inline float masked_grad(float dy, float y, int relu, float drop_p, float keep_scale) {
    if (relu) return y > 0.f ? dy * keep_scale : 0.f;
    if (drop_p > 0.f) return y != 0.f ? dy * keep_scale : 0.f;
    return dy;
}

// part[0][bx][c] = sum g, part[1][bx][c] = sum g*xhat
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ y, int ldy,
                                                              const float* __restrict__ dy, int lddy, long long P, int C, long long rows_per_block,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              int relu, float drop_p, float* __restrict__ part, int nbx) {
    __shared__ float sh[2][256];
    const ChanMap m = chan_map(C, blockIdx.y);
    const long long row0 = blockIdx.x * rows_per_block, row1 = min(P, row0 + rows_per_block);
    float sg = 0.f, sgx = 0.f;
    if (m.c >= 0) {
        const float mu = mean[m.c], is = invstd[m.c];
        const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
        const bool need_y = relu || drop_p > 0.f;
#pragma unroll 4
        for (long long p = row0 + m.slot; p < row1; p += m.G) {
            const float g = masked_grad(dy[p * lddy + m.c], need_y ? y[p * ldy + m.c] : 1.f, relu, drop_p, ks);
            sg += g; sgx += g * ((x[p * ldx + m.c] - mu) * is);
        }
    }
    sh[0][threadIdx.x] = sg; sh[1][threadIdx.x] = sgx;
    __syncthreads();
    if (m.c >= 0 && m.slot == 0) {
        for (int g = 1; g < m.G; ++g) { const int t = g * m.cg + (m.c - m.cg0); sg += sh[0][t]; sgx += sh[1][t]; }
        const long long o = (long long)blockIdx.x * C + m.c;
        part[o] = sg; part[(long long)nbx * C + o] = sgx;
    }
}

// sums[0][c] = dbeta, sums[1][c] = dgamma (fp32 copies also written to the parameter gradients)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nbx, int C, float* __restrict__ sums,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double sh[2][32][8];
    const int cl = threadIdx.x & 7, sl = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;
    double a = 0, b = 0;
    if (c < C) {
        float va[8], vb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = sl + 32 * j;
            const bool ok = i < nbx;
            va[j] = ok ? part[(long long)i * C + c] : 0.f; vb[j] = ok ? part[(long long)nbx * C + (long long)i * C + c] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { a += va[j]; b += vb[j]; }
    }
    sh[0][sl][cl] = a; sh[1][sl][cl] = b;
    __syncthreads();
    if (sl != 0 || c >= C) return;
    a = 0; b = 0;
    for (int s2 = 0; s2 < 32; ++s2) { a += sh[0][s2][cl]; b += sh[1][s2][cl]; }
    sums[c] = (float)a; sums[C + c] = (float)b;
    if (dbeta) dbeta[c] = (float)a;
    if (dgamma) dgamma[c] = (float)b;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ y, int ldy,
                                                            const float* __restrict__ dy, int lddy, float* __restrict__ dx, int lddx,
                                                            float* __restrict__ dres, int lddr, long long P, int C,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ sums, int relu, float drop_p, int training,
                                                            unsigned* __restrict__ dx_amax) {
    const ChanMap m = chan_map(C, blockIdx.y);
    unsigned am = 0u;
    if (m.c >= 0) {
        const float mu = mean[m.c], is = invstd[m.c], gi = gamma[m.c] * is;
        const float inv_n = 1.f / (float)P;
        const float mb = training ? sums[m.c] * inv_n : 0.f, mg = training ? sums[C + m.c] * inv_n : 0.f;
        const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
        const bool need_y = relu || drop_p > 0.f;
        for (long long p = (long long)blockIdx.x * m.G + m.slot; p < P; p += (long long)gridDim.x * m.G) {
            const float g = masked_grad(dy[p * lddy + m.c], need_y ? y[p * ldy + m.c] : 1.f, relu, drop_p, ks);
            const float xh = (x[p * ldx + m.c] - mu) * is;
            const float d = gi * (g - mb - xh * mg);
            dx[p * lddx + m.c] = d;
            am = max(am, abs_bits(d));
            if (dres) dres[p * lddr + m.c] = g;
        }
    }
    amax_publish(am, dx_amax);
}

// ---------------------------------------------------------------------------------------------- column sums
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int ld, long long P, int C, long long rows_per_block,
                                                              float* __restrict__ part) {
    __shared__ float sh[256];
    const ChanMap m = chan_map(C, blockIdx.y);
    const long long row0 = blockIdx.x * rows_per_block, row1 = min(P, row0 + rows_per_block);
    float s = 0.f;
    if (m.c >= 0) for (long long p = row0 + m.slot; p < row1; p += m.G) s += x[p * ld + m.c];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (m.c >= 0 && m.slot == 0) {
        for (int g = 1; g < m.G; ++g) s += sh[g * m.cg + (m.c - m.cg0)];
        part[(long long)blockIdx.x * C + m.c] = s;
    }
}
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ part, int nbx, int C, float* __restrict__ out) {
    __shared__ double sh[8][32];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double a = 0;
    if (c < C) {
        int i = sl;
        for (; i + 56 < nbx; i += 64) {             // eight loads in flight; the additions keep the plain loop's order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(long long)(i + 8 * u) * C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; i < nbx; i += 8) a += part[(long long)i * C + c];
    }
    sh[sl][cl] = a;
    __syncthreads();
    if (sl != 0 || c >= C) return;
    for (int s2 = 1; s2 < 8; ++s2) a += sh[s2][cl];
    out[c] = (float)a;
}

// ---------------------------------------------------------------------------------------------- dropout
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, long long P, int C,
                                                       float p_drop, SeedArg seed_arg, unsigned rng_stream) {
    const unsigned long long seed = seed_arg.dev ? *seed_arg.dev : seed_arg.value;     // device-resident key: replayable from a graph
    const long long total = P * C;
    const float ks = 1.f / (1.f - p_drop);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long p = e / C; const int c = (int)(e - p * C);
        const float v = x[p * ldx + c];
        y[p * ldy + c] = (philox_uniform((unsigned long long)e, seed, rng_stream) >= p_drop) ? v * ks : 0.f;
    }
}


// ============================================================================================== float4 variants
// C % 4 == 0, ld % 4 == 0, 16-byte aligned bases: a lane owns 4 consecutive channels (one 16-byte access per tensor and
// pixel, one Philox call per 4 dropout draws); G pixels are processed side by side.
struct ChanMap4 { int q0, cq, G, q, slot; };       // q = first channel / 4 (absolute), -1 if idle
__device__ inline ChanMap4 chan_map4(int C, int group_idx) {
    ChanMap4 m;
    const int Q = C >> 2;
    m.q0 = group_idx * 256;
    m.cq = min(256, Q - m.q0);
    m.G = 256 / m.cq;
    const int t = threadIdx.x;
    if (t < m.G * m.cq) { m.q = m.q0 + t % m.cq; m.slot = t / m.cq; }
    else { m.q = -1; m.slot = 0; }
    return m;
}
#define LD4(ptr, p, ld, q) (*reinterpret_cast<const float4*>((ptr) + (p) * (long long)(ld) + 4 * (q)))
#define ST4(ptr, p, ld, q) (*reinterpret_cast<float4*>((ptr) + (p) * (long long)(ld) + 4 * (q)))

__global__ __launch_bounds__(256) void bn_partial4_kernel(const float* __restrict__ x, int ld, long long P, int C, long long rows_per_block,
                                                           float* __restrict__ part, int nbx) {
    __shared__ float sh[3][4][256];
    const ChanMap4 m = chan_map4(C, blockIdx.y);
    const long long row0 = blockIdx.x * rows_per_block, row1 = min(P, row0 + rows_per_block);
    float n = 0.f, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, k0[4] = {0.f, 0.f, 0.f, 0.f};
    if (m.q >= 0) {
        long long p = row0 + m.slot;
        if (p < row1) { const float4 v = LD4(x, p, ld, m.q); k0[0] = v.x; k0[1] = v.y; k0[2] = v.z; k0[3] = v.w; }
#pragma unroll 4
        for (; p < row1; p += m.G) {
            const float4 v = LD4(x, p, ld, m.q);
            const float d0 = v.x - k0[0], d1 = v.y - k0[1], d2 = v.z - k0[2], d3 = v.w - k0[3];
            s1[0] += d0; s2[0] += d0 * d0; s1[1] += d1; s2[1] += d1 * d1; s1[2] += d2; s2[2] += d2 * d2; s1[3] += d3; s2[3] += d3 * d3;
            n += 1.f;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float mean = 0.f, m2 = 0.f;
        if (n > 0.f) { mean = k0[j] + s1[j] / n; m2 = fmaxf(s2[j] - s1[j] * s1[j] / n, 0.f); }
        sh[0][j][threadIdx.x] = n; sh[1][j][threadIdx.x] = mean; sh[2][j][threadIdx.x] = m2;
    }
    __syncthreads();
    if (m.q >= 0 && m.slot == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float na = sh[0][j][threadIdx.x], ma = sh[1][j][threadIdx.x], qa = sh[2][j][threadIdx.x];
            for (int g = 1; g < m.G; ++g) {
                const int t = g * m.cq + (m.q - m.q0);
                const float nb = sh[0][j][t];
                if (nb > 0.f) {
                    const float mb = sh[1][j][t], qb = sh[2][j][t];
                    const float nt = na + nb, d = mb - ma;
                    ma += d * (nb / nt);
                    qa += qb + d * d * (na * nb / nt);
                    na = nt;
                }
            }
            const long long o = (long long)blockIdx.x * C + 4 * m.q + j;
            part[o] = na; part[(long long)nbx * C + o] = ma; part[2ll * nbx * C + o] = qa;
        }
    }
}

__global__ __launch_bounds__(256) void bn_apply4_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, long long P, int C,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ res, int ldr, int relu, float drop_p, SeedArg seed_arg, unsigned rng_stream,
                                                         unsigned* __restrict__ y_amax) {
    const unsigned long long seed = seed_arg.dev ? *seed_arg.dev : seed_arg.value;     // device-resident key: replayable from a graph
    const ChanMap4 m = chan_map4(C, blockIdx.y);
    unsigned am = 0u;
    float sc[4] = {0.f, 0.f, 0.f, 0.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (m.q >= 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int c = 4 * m.q + j; sc[j] = gamma[c] * invstd[c]; sh[j] = beta[c] - mean[c] * sc[j]; }
    }
    const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
#pragma unroll 2
    for (long long p = m.q >= 0 ? (long long)blockIdx.x * m.G + m.slot : P; p < P; p += (long long)gridDim.x * m.G) {
        const float4 xv = LD4(x, p, ldx, m.q);
        float v[4] = {fmaf(xv.x, sc[0], sh[0]), fmaf(xv.y, sc[1], sh[1]), fmaf(xv.z, sc[2], sh[2]), fmaf(xv.w, sc[3], sh[3])};
        if (res) { const float4 r = LD4(res, p, ldr, m.q); v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
        if (relu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (drop_p > 0.f) {
            const unsigned long long e = (unsigned long long)p * C + 4ull * m.q;       // multiple of 4: one Philox block
            unsigned r[4];
            philox4x32_10((unsigned)(e >> 2), (unsigned)(e >> 34), rng_stream, 0u, (unsigned)seed, (unsigned)(seed >> 32), r);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ((float)(r[j] >> 8) * 5.9604644775390625e-08f >= drop_p) ? v[j] * ks : 0.f;
        }
        ST4(y, p, ldy, m.q) = make_float4(v[0], v[1], v[2], v[3]);
        am = abs_bits4(am, v[0], v[1], v[2], v[3]);
    }
    amax_publish(am, y_amax);
}

__global__ __launch_bounds__(256) void bn_bwd_partial4_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ y, int ldy,
                                                               const float* __restrict__ dy, int lddy, long long P, int C, long long rows_per_block,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               int relu, float drop_p, float* __restrict__ part, int nbx) {
    __shared__ float sh[2][4][256];
    const ChanMap4 m = chan_map4(C, blockIdx.y);
    const long long row0 = blockIdx.x * rows_per_block, row1 = min(P, row0 + rows_per_block);
    float sg[4] = {0.f, 0.f, 0.f, 0.f}, sgx[4] = {0.f, 0.f, 0.f, 0.f};
    if (m.q >= 0) {
        float mu[4], is[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { mu[j] = mean[4 * m.q + j]; is[j] = invstd[4 * m.q + j]; }
        const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
        const bool need_y = relu || drop_p > 0.f;
#pragma unroll 2
        for (long long p = row0 + m.slot; p < row1; p += m.G) {
            const float4 dv = LD4(dy, p, lddy, m.q), xv = LD4(x, p, ldx, m.q);
            float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
            if (need_y) yv = LD4(y, p, ldy, m.q);
            const float g0 = masked_grad(dv.x, yv.x, relu, drop_p, ks), g1 = masked_grad(dv.y, yv.y, relu, drop_p, ks);
            const float g2 = masked_grad(dv.z, yv.z, relu, drop_p, ks), g3 = masked_grad(dv.w, yv.w, relu, drop_p, ks);
            sg[0] += g0; sgx[0] += g0 * ((xv.x - mu[0]) * is[0]);
            sg[1] += g1; sgx[1] += g1 * ((xv.y - mu[1]) * is[1]);
            sg[2] += g2; sgx[2] += g2 * ((xv.z - mu[2]) * is[2]);
            sg[3] += g3; sgx[3] += g3 * ((xv.w - mu[3]) * is[3]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { sh[0][j][threadIdx.x] = sg[j]; sh[1][j][threadIdx.x] = sgx[j]; }
    __syncthreads();
    if (m.q >= 0 && m.slot == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = sg[j], b = sgx[j];
            for (int g = 1; g < m.G; ++g) { const int t = g * m.cq + (m.q - m.q0); a += sh[0][j][t]; b += sh[1][j][t]; }
            const long long o = (long long)blockIdx.x * C + 4 * m.q + j;
            part[o] = a; part[(long long)nbx * C + o] = b;
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply4_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ y, int ldy,
                                                             const float* __restrict__ dy, int lddy, float* __restrict__ dx, int lddx,
                                                             float* __restrict__ dres, int lddr, long long P, int C,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ sums, int relu, float drop_p, int training,
                                                             unsigned* __restrict__ dx_amax) {
    const ChanMap4 m = chan_map4(C, blockIdx.y);
    unsigned am = 0u;
    float mu[4] = {}, is[4] = {}, gi[4] = {}, mb[4] = {}, mg[4] = {};
    const float inv_n = 1.f / (float)P;
    if (m.q >= 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * m.q + j;
            mu[j] = mean[c]; is[j] = invstd[c]; gi[j] = gamma[c] * is[j];
            mb[j] = training ? sums[c] * inv_n : 0.f; mg[j] = training ? sums[C + c] * inv_n : 0.f;
        }
    }
    const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const bool need_y = relu || drop_p > 0.f;
#pragma unroll 2
    for (long long p = m.q >= 0 ? (long long)blockIdx.x * m.G + m.slot : P; p < P; p += (long long)gridDim.x * m.G) {
        const float4 dv = LD4(dy, p, lddy, m.q), xv = LD4(x, p, ldx, m.q);
        float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
        if (need_y) yv = LD4(y, p, ldy, m.q);
        const float g0 = masked_grad(dv.x, yv.x, relu, drop_p, ks), g1 = masked_grad(dv.y, yv.y, relu, drop_p, ks);
        const float g2 = masked_grad(dv.z, yv.z, relu, drop_p, ks), g3 = masked_grad(dv.w, yv.w, relu, drop_p, ks);
        const float4 d = make_float4(gi[0] * (g0 - mb[0] - (xv.x - mu[0]) * is[0] * mg[0]), gi[1] * (g1 - mb[1] - (xv.y - mu[1]) * is[1] * mg[1]),
                                     gi[2] * (g2 - mb[2] - (xv.z - mu[2]) * is[2] * mg[2]), gi[3] * (g3 - mb[3] - (xv.w - mu[3]) * is[3] * mg[3]));
        ST4(dx, p, lddx, m.q) = d;
        am = abs_bits4(am, d.x, d.y, d.z, d.w);
        if (dres) ST4(dres, p, lddr, m.q) = make_float4(g0, g1, g2, g3);
    }
    amax_publish(am, dx_amax);
}


// ---------------------------------------------------------------------------------------------- fused BN for small tensors
// The BN layers of the deep stages see 4-16 MB tensors: three dependent kernels (partial / finalize / apply) are launch-latency
// bound there.  One kernel instead: 256 blocks = (channel group of 32 channels = one 128-byte line per pixel) x (row slab); a
// block keeps its whole slab in registers (<= 16 float4 per thread and tensor), writes one partial per channel, crosses ONE
// device-wide barrier, merges the <= 256 slab partials of its own 32 channels (every block of a group computes bit-identical
// statistics: same partials, same order) and applies from registers - the tensor is read once.  The barrier is a self-resetting
// arrival counter + generation word (grid_barrier below): nothing is handed over by the host, so a launch can be replayed from a graph.
// 256 blocks of <= 128 registers are always co-resident on 256 CUs; the spin is bounded so that a bug cannot hang the GPU.
__device__ unsigned int g_grid_count = 0u;               // arrivals of the barrier in progress (back to 0 when it completes)
__device__ unsigned int g_grid_gen = 0u;                 // generation: bumped by the last arriver, which is what the others wait for
__device__ unsigned int g_grid_timeouts = 0u;           // blocks that gave up waiting (dsrl_bn_fused_barrier_timeouts)
constexpr int kFusedBlocks = 128, kFusedBlocksBig = 256, kFusedThreads = 512, kFusedMaxPasses = 16;   // measured: the barrier costs ~20 ns per arriving block,
                                                                                                   // so 128 blocks unless the tensor needs the registers of 256
constexpr int kFusedRL = kFusedThreads / 8, kFusedNW = kFusedThreads / 64, kFusedNS = kFusedThreads / 32;   // row lanes, waves, merge slices

// Hand-off protocol (MI355X_MICROARCH.md, "Valid forms", sc1 stores + sc1 loads): the per-XCD L2s are not coherent with each
// other, so every handed-off value (the slab partials) is written with an agent-scope store (sc1: written through to the
// coherence point); EVERY storing wave drains its stores (s_waitcnt vmcnt(0), explicit inline asm - __syncthreads() alone only
// waits for LDS traffic) before the workgroup barrier behind which lane 0 bumps the arrival counter, and the partials are read
// back with agent-scope loads (st_agent / ld_agent).  No L2-wide write-back / invalidate is needed, which is what an agent-scope
// release/acquire fence pair would cost in every one of the 256 blocks.
// The barrier itself is self-resetting (sense reversal): lane 0 reads the generation, arrives, and either - as the last of the
// `nblocks` arrivers - zeroes the count and bumps the generation, or spins until the generation moves.  No state is handed over by
// the host, so a launch may be replayed from a hipGraph; the fused launches of one device must not overlap each other (the host
// side keeps them on one stream, see dsrl_bn_train_fwd).
__device__ inline bool grid_barrier(unsigned nblocks) {
    __shared__ int ok_sh;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's sc1 partial stores have reached the coherence point
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned gen = __hip_atomic_load(&g_grid_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the generation is read BEFORE this block counts as arrived
        const unsigned prev = __hip_atomic_fetch_add(&g_grid_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = true;
        if (prev + 1u == nblocks) {
            __hip_atomic_store(&g_grid_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the reset lands before anybody is released
            __hip_atomic_fetch_add(&g_grid_gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(&g_grid_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                if (++spins >= (1u << 25)) {                                 // seconds: the other blocks never became resident
                    ok = false;
                    __hip_atomic_fetch_add(&g_grid_timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        ok_sh = ok ? 1 : 0;
    }
    __syncthreads();
    return ok_sh != 0;      // false: the caller poisons its outputs with NaN, which the per-iteration NaN check reports
}
__device__ inline float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ inline void chan_merge(float& na, float& ma, float& qa, float nb, float mb, float qb) {
    if (nb > 0.f) {
        const float nt = na + nb, d = mb - ma;
        ma += d * (nb / nt);
        qa += qb + d * d * (na * nb / nt);
        na = nt;
    }
}

// thread (l8 = tid & 7: float4 column of the 32-channel group, rr = tid >> 3: row lane); rows row0 + rr + kFusedRL i
__global__ __launch_bounds__(kFusedThreads) void bn_fused_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int P, int C,
                                                            int groups, int slabs, int rows_per_slab, float eps, float momentum,
                                                            float* __restrict__ mean_out, float* __restrict__ invstd_out, float* __restrict__ rm, float* __restrict__ rv,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ res, int ldr, int relu, float drop_p, SeedArg seed_arg, unsigned rng_stream,
                                                            float* part, unsigned* __restrict__ y_amax) {
    const unsigned long long seed = seed_arg.dev ? *seed_arg.dev : seed_arg.value;     // device-resident key: replayable from a graph
    __shared__ float shw[kFusedNW][3][4][8];          // per wave: (n, mean, M2) x 4 channels x 8 float4 columns
    __shared__ double shm[kFusedNS][3][32];           // cross-slab merge: 8 slices x 3 sums x 32 channels
    __shared__ float fin[2][32];               // mean, invstd of the group's channels
    const int grp = blockIdx.x % groups, slab = blockIdx.x / groups;
    const int tid = threadIdx.x, l8 = tid & 7, rr = tid >> 3, wave = tid >> 6;
    const int q = grp * 8 + l8;
    const int row0 = slab * rows_per_slab, row1 = min(P, row0 + rows_per_slab);

    float4 xv[kFusedMaxPasses];
#pragma unroll
    for (int i = 0; i < kFusedMaxPasses; ++i) {
        const int p = row0 + rr + kFusedRL * i;
        xv[i] = p < row1 ? LD4(x, p, ldx, q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // ---- slab statistics: per thread shifted sums, then Chan merges over the row lanes (xor tree in the wave, 4 waves via LDS)
    float n = 0.f, mean[4], m2[4];
    {
        const float k0[4] = {xv[0].x, xv[0].y, xv[0].z, xv[0].w};
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < kFusedMaxPasses; ++i) {
            if (row0 + rr + kFusedRL * i < row1) {
                const float d0 = xv[i].x - k0[0], d1 = xv[i].y - k0[1], d2 = xv[i].z - k0[2], d3 = xv[i].w - k0[3];
                s1[0] += d0; s2[0] += d0 * d0; s1[1] += d1; s2[1] += d1 * d1; s1[2] += d2; s2[2] += d2 * d2; s1[3] += d3; s2[3] += d3 * d3;
                n += 1.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            mean[j] = n > 0.f ? k0[j] + s1[j] / n : 0.f;
            m2[j] = n > 0.f ? fmaxf(s2[j] - s1[j] * s1[j] / n, 0.f) : 0.f;
        }
    }
#pragma unroll
    for (int sft = 8; sft < 64; sft <<= 1) {
        const float nb = __shfl_xor(n, sft);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float na = n;                      // the count is shared by the four channels of a thread
            chan_merge(na, mean[j], m2[j], nb, __shfl_xor(mean[j], sft), __shfl_xor(m2[j], sft));
        }
        n += nb;
    }
    if ((tid & 63) < 8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { shw[wave][0][j][l8] = n; shw[wave][1][j][l8] = mean[j]; shw[wave][2][j][l8] = m2[j]; }
    }
    __syncthreads();
    if (tid < 32) {                            // channel tid of the group: merge the 4 wave results, write the slab partial
        const int j = tid & 3, c8 = tid >> 2;
        float na = shw[0][0][j][c8], ma = shw[0][1][j][c8], qa = shw[0][2][j][c8];
#pragma unroll
        for (int w = 1; w < kFusedNW; ++w) chan_merge(na, ma, qa, shw[w][0][j][c8], shw[w][1][j][c8], shw[w][2][j][c8]);
        const long long o = (long long)slab * C + grp * 32 + tid;
        st_agent(part + o, na); st_agent(part + (long long)slabs * C + o, ma); st_agent(part + 2ll * slabs * C + o, qa);
    }
    const bool arrived = grid_barrier(gridDim.x);
    // ---- statistics of the group's 32 channels from all slabs in one pass, fp64, shifted by slab 0's mean (no cancellation):
    //      N = sum n, S = sum n d, T = sum (M2 + n d^2), d = m - m_ref;  mean = m_ref + S / N, M2 = T - S^2 / N.
    //      8 slices of the slabs per channel, 4 slabs (12 independent loads) in flight per thread; slice results added in fixed order
    {
        const int ch = tid & 31, k = tid >> 5;
        const float* pn = part + grp * 32 + ch;
        const float* pm_ = pn + (long long)slabs * C;
        const float* pq = pn + 2ll * slabs * C;
        const float mref = ld_agent(pm_);
        double N = 0, S = 0, T = 0;
        for (int base = k; base < slabs; base += 4 * kFusedNS) {
            float vn[4], vm[4], vq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int sl = min(base + kFusedNS * u, slabs - 1);
                vn[u] = ld_agent(pn + (long long)sl * C); vm[u] = ld_agent(pm_ + (long long)sl * C); vq[u] = ld_agent(pq + (long long)sl * C);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double nb = base + kFusedNS * u < slabs ? (double)vn[u] : 0.0, d = (double)vm[u] - (double)mref;
                N += nb; S += nb * d; T += (base + kFusedNS * u < slabs ? (double)vq[u] : 0.0) + nb * d * d;
            }
        }
        shm[k][0][ch] = N; shm[k][1][ch] = S; shm[k][2][ch] = T;
        if (k == 0) fin[0][ch] = mref;
    }
    __syncthreads();
    if (tid < 32) {
        double N = 0, S = 0, T = 0;
#pragma unroll
        for (int k = 0; k < kFusedNS; ++k) { N += shm[k][0][tid]; S += shm[k][1][tid]; T += shm[k][2][tid]; }
        const double mu = N > 0 ? (double)fin[0][tid] + S / N : 0.0;
        const double Q = N > 0 ? fmax(T - S * S / N, 0.0) : 0.0;
        const double var = N > 0 ? Q / N : 0.0;
        const float is = (float)(1.0 / sqrt(var + (double)eps));
        fin[0][tid] = arrived ? (float)mu : __builtin_nanf(""); fin[1][tid] = is;
        if (slab == 0) {
            const int c = grp * 32 + tid;
            mean_out[c] = fin[0][tid]; invstd_out[c] = is;
            if (rm) rm[c] = (float)((1.0 - momentum) * rm[c] + momentum * mu);
            if (rv) rv[c] = (float)((1.0 - momentum) * rv[c] + momentum * (N > 1 ? Q / (N - 1) : var));
        }
    }
    __syncthreads();
    // ---- apply from registers
    float sc[4], sf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int c = 4 * q + j; sc[j] = gamma[c] * fin[1][4 * l8 + j]; sf[j] = beta[c] - fin[0][4 * l8 + j] * sc[j]; }
    const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    unsigned am = 0u;
#pragma unroll
    for (int i = 0; i < kFusedMaxPasses; ++i) {
        const int p = row0 + rr + kFusedRL * i;
        if (p >= row1) continue;
        float v[4] = {fmaf(xv[i].x, sc[0], sf[0]), fmaf(xv[i].y, sc[1], sf[1]), fmaf(xv[i].z, sc[2], sf[2]), fmaf(xv[i].w, sc[3], sf[3])};
        if (res) { const float4 r = LD4(res, p, ldr, q); v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
        if (relu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (drop_p > 0.f) {
            const unsigned long long e = (unsigned long long)p * C + 4ull * q;       // multiple of 4: one Philox block
            unsigned r[4];
            philox4x32_10((unsigned)(e >> 2), (unsigned)(e >> 34), rng_stream, 0u, (unsigned)seed, (unsigned)(seed >> 32), r);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ((float)(r[j] >> 8) * 5.9604644775390625e-08f >= drop_p) ? v[j] * ks : 0.f;
        }
        ST4(y, p, ldy, q) = make_float4(v[0], v[1], v[2], v[3]);
        am = abs_bits4(am, v[0], v[1], v[2], v[3]);
    }
    amax_publish(am, y_amax);
}

__global__ __launch_bounds__(kFusedThreads) void bn_fused_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ y, int ldy,
                                                            const float* __restrict__ dy, int lddy, float* __restrict__ dx, int lddx,
                                                            float* __restrict__ dres, int lddr, int P, int C, int groups, int slabs, int rows_per_slab,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int relu, float drop_p, int training,
                                                            float* part, unsigned* __restrict__ dx_amax) {
    __shared__ float shw[kFusedNW][2][4][8];
    __shared__ double shm[kFusedNS][2][32];
    __shared__ float fin[2][32];               // sum g / n, sum g*xhat / n
    const int grp = blockIdx.x % groups, slab = blockIdx.x / groups;
    const int tid = threadIdx.x, l8 = tid & 7, rr = tid >> 3, wave = tid >> 6;
    const int q = grp * 8 + l8;
    const int row0 = slab * rows_per_slab, row1 = min(P, row0 + rows_per_slab);
    float mu[4], is[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { mu[j] = mean[4 * q + j]; is[j] = invstd[4 * q + j]; }
    const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const bool need_y = relu || drop_p > 0.f;

    float4 g[kFusedMaxPasses], xh[kFusedMaxPasses];
    float sg[4] = {0.f, 0.f, 0.f, 0.f}, sgx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < kFusedMaxPasses; ++i) {
        const int p = row0 + rr + kFusedRL * i;
        if (p < row1) {
            const float4 dv = LD4(dy, p, lddy, q), xv = LD4(x, p, ldx, q);
            float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
            if (need_y) yv = LD4(y, p, ldy, q);
            g[i] = make_float4(masked_grad(dv.x, yv.x, relu, drop_p, ks), masked_grad(dv.y, yv.y, relu, drop_p, ks),
                               masked_grad(dv.z, yv.z, relu, drop_p, ks), masked_grad(dv.w, yv.w, relu, drop_p, ks));
            xh[i] = make_float4((xv.x - mu[0]) * is[0], (xv.y - mu[1]) * is[1], (xv.z - mu[2]) * is[2], (xv.w - mu[3]) * is[3]);
            sg[0] += g[i].x; sgx[0] += g[i].x * xh[i].x; sg[1] += g[i].y; sgx[1] += g[i].y * xh[i].y;
            sg[2] += g[i].z; sgx[2] += g[i].z * xh[i].z; sg[3] += g[i].w; sgx[3] += g[i].w * xh[i].w;
            if (dres) ST4(dres, p, lddr, q) = g[i];         // the residual branch's gradient does not depend on the sums: written under the read phase
        } else {
            g[i] = make_float4(0.f, 0.f, 0.f, 0.f); xh[i] = g[i];
        }
    }
#pragma unroll
    for (int sft = 8; sft < 64; sft <<= 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { sg[j] += __shfl_xor(sg[j], sft); sgx[j] += __shfl_xor(sgx[j], sft); }
    }
    if ((tid & 63) < 8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { shw[wave][0][j][l8] = sg[j]; shw[wave][1][j][l8] = sgx[j]; }
    }
    __syncthreads();
    if (tid < 32) {
        const int j = tid & 3, c8 = tid >> 2;
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < kFusedNW; ++w) { a += shw[w][0][j][c8]; b += shw[w][1][j][c8]; }
        const long long o = (long long)slab * C + grp * 32 + tid;
        st_agent(part + o, a); st_agent(part + (long long)slabs * C + o, b);
    }
    const bool arrived = grid_barrier(gridDim.x);
    {
        const int ch = tid & 31, k = tid >> 5;
        const float* pa = part + grp * 32 + ch;
        const float* pb = pa + (long long)slabs * C;
        double a = 0, b = 0;
        for (int base = k; base < slabs; base += 4 * kFusedNS) {          // 4 slabs (8 independent loads) in flight per thread
            float va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int sl = min(base + kFusedNS * u, slabs - 1);
                va[u] = ld_agent(pa + (long long)sl * C); vb[u] = ld_agent(pb + (long long)sl * C);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (base + kFusedNS * u < slabs) { a += va[u]; b += vb[u]; }
        }
        shm[k][0][ch] = a; shm[k][1][ch] = b;
    }
    __syncthreads();
    if (tid < 32) {
        double a = 0, b = 0;
#pragma unroll
        for (int k = 0; k < kFusedNS; ++k) { a += shm[k][0][tid]; b += shm[k][1][tid]; }
        const float inv_n = 1.f / (float)P;
        fin[0][tid] = !arrived ? __builtin_nanf("") : (training ? (float)a * inv_n : 0.f); fin[1][tid] = training ? (float)b * inv_n : 0.f;
        if (slab == 0) {
            const int c = grp * 32 + tid;
            if (dbeta) dbeta[c] = (float)a;
            if (dgamma) dgamma[c] = (float)b;
        }
    }
    __syncthreads();
    float gi[4], mb[4], mg[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { gi[j] = gamma[4 * q + j] * is[j]; mb[j] = fin[0][4 * l8 + j]; mg[j] = fin[1][4 * l8 + j]; }
    unsigned am = 0u;
#pragma unroll
    for (int i = 0; i < kFusedMaxPasses; ++i) {
        const int p = row0 + rr + kFusedRL * i;
        if (p >= row1) continue;
        const float4 d = make_float4(gi[0] * (g[i].x - mb[0] - xh[i].x * mg[0]), gi[1] * (g[i].y - mb[1] - xh[i].y * mg[1]),
                                     gi[2] * (g[i].z - mb[2] - xh[i].z * mg[2]), gi[3] * (g[i].w - mb[3] - xh[i].w * mg[3]));
        ST4(dx, p, lddx, q) = d;
        am = abs_bits4(am, d.x, d.y, d.z, d.w);
    }
    amax_publish(am, dx_amax);
}


// ---------------------------------------------------------------------------------------------- BN forward from conv-epilogue partials
constexpr int kMergeRows = 4;      // partial rows a thread requests at once in the merges below
// The producing conv already wrote (n, mean, M2) per (row block, channel) of its output (conv_igemm_split_kernel epilogue), so the
// statistics pass and the device-wide barrier of bn_fused_fwd_kernel disappear: a block = (32-channel group) x (row slab) merges
// the <= 256 partials of its own 32 channels (fp64, fixed order: every block of a group gets bit-identical statistics) and streams
// x -> y once.  Any number of blocks; slab-0 blocks publish mean / invstd and update the running statistics.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void bn_stats_apply_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int P, int C,
                                                              int groups, int rows_per_slab, float eps, float momentum,
                                                              float* __restrict__ mean_out, float* __restrict__ invstd_out, float* __restrict__ rm, float* __restrict__ rv,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ res, int ldr, int relu, float drop_p, SeedArg seed_arg, unsigned rng_stream,
                                                              const float* __restrict__ part, int nparts, unsigned* __restrict__ y_amax) {
    const unsigned long long seed = seed_arg.dev ? *seed_arg.dev : seed_arg.value;     // device-resident key: replayable from a graph
    constexpr int KG = THREADS / 32, RP = THREADS / 8;      // row groups of the merge, tensor rows per pass
    __shared__ double shm[KG][3][32];
    __shared__ float fin[2][32];
    const int grp = blockIdx.x % groups, slab = blockIdx.x / groups;
    const int tid = threadIdx.x, l8 = tid & 7, rr = tid >> 3;
    const int q = grp * 8 + l8;
    // The first row of the slab and the affine parameters are requested before the merge of the partials: a P = 4096 tensor gives a thread
    // one row, and its load would otherwise start only behind the merge's two barriers.
    const int row0 = slab * rows_per_slab, row1 = min(P, row0 + rows_per_slab), p0 = row0 + rr;
    float4 xv0 = make_float4(0.f, 0.f, 0.f, 0.f), rv0 = xv0;
    if (p0 < row1) {
        xv0 = LD4(x, p0, ldx, q);
        if (res) rv0 = LD4(res, p0, ldr, q);
    }
    float gmv[4], btv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { gmv[j] = gamma[4 * q + j]; btv[j] = beta[4 * q + j]; }
    {
        const int ch = tid & 31, k = tid >> 5;
        const float* pn = part + grp * 32 + ch;
        const float* pm_ = pn + (long long)nparts * C;
        const float* pq = pn + 2ll * nparts * C;
        const float mref = pm_[0];
        double N = 0, S = 0, T = 0;
        // rows k, k + KG, k + 2 KG, ... in that order; the next batch of rows is requested before this one is summed
        struct Batch { float n[kMergeRows], m[kMergeRows], q[kMergeRows]; };
        auto fetch = [&](int base, Batch& b) {
#pragma unroll
            for (int u = 0; u < kMergeRows; ++u) {
                const int sl = min(base + KG * u, nparts - 1);
                b.n[u] = pn[(long long)sl * C]; b.m[u] = pm_[(long long)sl * C]; b.q[u] = pq[(long long)sl * C];
            }
        };
        auto sum = [&](int base, const Batch& b) {
#pragma unroll
            for (int u = 0; u < kMergeRows; ++u) {
                const double nb = base + KG * u < nparts ? (double)b.n[u] : 0.0, d = (double)b.m[u] - (double)mref;
                N += nb; S += nb * d; T += (base + KG * u < nparts ? (double)b.q[u] : 0.0) + nb * d * d;
            }
        };
        Batch b0, b1;
        fetch(k, b0);
        for (int base = k; base < nparts; base += 2 * KG * kMergeRows) {
            const int mid = base + KG * kMergeRows;
            if (mid < nparts) fetch(mid, b1);
            sum(base, b0);
            if (mid < nparts) {
                if (mid + KG * kMergeRows < nparts) fetch(mid + KG * kMergeRows, b0);
                sum(mid, b1);
            }
        }
        shm[k][0][ch] = N; shm[k][1][ch] = S; shm[k][2][ch] = T;
        if (k == 0) fin[0][ch] = mref;
    }
    __syncthreads();
    if (tid < 32) {
        double N = 0, S = 0, T = 0;
#pragma unroll
        for (int k = 0; k < KG; ++k) { N += shm[k][0][tid]; S += shm[k][1][tid]; T += shm[k][2][tid]; }
        const double mu = N > 0 ? (double)fin[0][tid] + S / N : 0.0;
        const double Q = N > 0 ? fmax(T - S * S / N, 0.0) : 0.0;
        const double var = N > 0 ? Q / N : 0.0;
        const float is = (float)(1.0 / sqrt(var + (double)eps));
        fin[0][tid] = (float)mu; fin[1][tid] = is;
        if (slab == 0) {
            const int c = grp * 32 + tid;
            mean_out[c] = (float)mu; invstd_out[c] = is;
            if (rm) rm[c] = (float)((1.0 - momentum) * rm[c] + momentum * mu);
            if (rv) rv[c] = (float)((1.0 - momentum) * rv[c] + momentum * (N > 1 ? Q / (N - 1) : var));
        }
    }
    __syncthreads();
    float sc[4], sf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { sc[j] = gmv[j] * fin[1][4 * l8 + j]; sf[j] = btv[j] - fin[0][4 * l8 + j] * sc[j]; }
    const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    unsigned am = 0u;
    auto row = [&](int p, const float4 xv, const float4 r) {
        float v[4] = {fmaf(xv.x, sc[0], sf[0]), fmaf(xv.y, sc[1], sf[1]), fmaf(xv.z, sc[2], sf[2]), fmaf(xv.w, sc[3], sf[3])};
        if (res) { v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
        if (relu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (drop_p > 0.f) {
            const unsigned long long e = (unsigned long long)p * C + 4ull * q;
            unsigned r4[4];
            philox4x32_10((unsigned)(e >> 2), (unsigned)(e >> 34), rng_stream, 0u, (unsigned)seed, (unsigned)(seed >> 32), r4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ((float)(r4[j] >> 8) * 5.9604644775390625e-08f >= drop_p) ? v[j] * ks : 0.f;
        }
        ST4(y, p, ldy, q) = make_float4(v[0], v[1], v[2], v[3]);
        am = abs_bits4(am, v[0], v[1], v[2], v[3]);
    };
    if (p0 < row1) row(p0, xv0, rv0);
#pragma unroll 4
    for (int p = p0 + RP; p < row1; p += RP) {
        const float4 xv = LD4(x, p, ldx, q);
        float4 r = xv;
        if (res) r = LD4(res, p, ldr, q);
        row(p, xv, r);
    }
    amax_publish(am, y_amax);
}


// More than 256 row blocks of partials (the M = 65536 layers: layer1, the decoder convs): kStatsReduced groups of consecutive row blocks are
// merged first, one (32-channel group, reduced row) per block, with the formulas and the fixed order of the merges above; the apply
// kernels then read kStatsReduced partials.  The reduced rows live behind the partials in the same buffer (dsrl_bn_stats_floats).
constexpr int kStatsReduced = 32;
// in [3][nparts][C] (n, mean, M2) -> out [3][kStatsReduced][C]
__global__ __launch_bounds__(256) void bn_stats_reduce_kernel(const float* __restrict__ part, int nparts, int C, float* __restrict__ out) {
    __shared__ double shm[8][3][32];
    __shared__ float ref[32];
    const int grp = blockIdx.x, rrow = blockIdx.y, tid = threadIdx.x, ch = tid & 31, k = tid >> 5;
    const int per = (nparts + kStatsReduced - 1) / kStatsReduced, p0 = rrow * per, p1 = min(nparts, p0 + per);
    const float* pn = part + grp * 32 + ch;
    const float* pm_ = pn + (long long)nparts * C;
    const float* pq = pn + 2ll * nparts * C;
    const float mref = p0 < p1 ? pm_[(long long)p0 * C] : 0.f;
    double N = 0, S = 0, T = 0;
    for (int base = p0 + k; base < p1; base += 32) {        // rows p0 + k, + 8, ... in order, four requested at once
        float vn[4], vm[4], vq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int sl = min(base + 8 * u, p1 - 1);
            vn[u] = pn[(long long)sl * C]; vm[u] = pm_[(long long)sl * C]; vq[u] = pq[(long long)sl * C];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (base + 8 * u < p1) {
                const double nb = (double)vn[u], d = (double)vm[u] - (double)mref;
                N += nb; S += nb * d; T += (double)vq[u] + nb * d * d;
            }
    }
    shm[k][0][ch] = N; shm[k][1][ch] = S; shm[k][2][ch] = T;
    if (k == 0) ref[ch] = mref;
    __syncthreads();
    if (tid < 32) {
        N = S = T = 0;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) { N += shm[kk][0][tid]; S += shm[kk][1][tid]; T += shm[kk][2][tid]; }
        const double mu = N > 0 ? (double)ref[tid] + S / N : 0.0, Q = N > 0 ? fmax(T - S * S / N, 0.0) : 0.0;
        float* o = out + (long long)rrow * C + grp * 32 + tid;
        o[0] = (float)N; o[(long long)kStatsReduced * C] = (float)mu; o[2ll * kStatsReduced * C] = (float)Q;
    }
}
// in [2][nparts][C] (sum g, sum g * xhat) -> out [2][kStatsReduced][C]
__global__ __launch_bounds__(256) void bn_bwd_stats_reduce_kernel(const float* __restrict__ part, int nparts, int C, float* __restrict__ out) {
    __shared__ double shm[8][2][32];
    const int grp = blockIdx.x, rrow = blockIdx.y, tid = threadIdx.x, ch = tid & 31, k = tid >> 5;
    const int per = (nparts + kStatsReduced - 1) / kStatsReduced, p0 = rrow * per, p1 = min(nparts, p0 + per);
    const float* pa = part + grp * 32 + ch;
    const float* pb = pa + (long long)nparts * C;
    double a = 0, b = 0;
    for (int base = p0 + k; base < p1; base += 32) {
        float va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int sl = min(base + 8 * u, p1 - 1);
            va[u] = pa[(long long)sl * C]; vb[u] = pb[(long long)sl * C];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (base + 8 * u < p1) { a += (double)va[u]; b += (double)vb[u]; }
    }
    shm[k][0][ch] = a; shm[k][1][ch] = b;
    __syncthreads();
    if (tid < 32) {
        a = b = 0;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) { a += shm[kk][0][tid]; b += shm[kk][1][tid]; }
        float* o = out + (long long)rrow * C + grp * 32 + tid;
        o[0] = (float)a; o[(long long)kStatsReduced * C] = (float)b;
    }
}

// BN backward from dgrad-epilogue partials: the conv whose data gradient IS this BatchNorm's output gradient already left
// sum(g) and sum(g * xhat) per (row block, channel) (conv_igemm_split_kernel, DGRAD epilogue); merge them for the block's 32
// channels (fp64, fixed order) and stream x, y, dy -> dx once.  No reduction pass, no device-wide barrier.
// RES2 (round 5): the residual of this BatchNorm is the output of ANOTHER BatchNorm without ReLU (the downsample branch of a layer's first bottleneck:
// out = relu(bn3(.) + bn_ds(conv_ds(x)))), whose output gradient is the masked gradient g this kernel writes to dres.  With that BatchNorm's input x2 and
// statistics the block also leaves ITS backward partial sums - sum g, sum g * xhat2 per (slab, channel) in part2 [2][slabs][C] - so that its backward is one
// from-statistics launch instead of the three-kernel / barrier path (a read of x2 here against a pass over x2, dres there).
struct BnRes2 { const float* x; int ldx; const float* mean; const float* invstd; float* part; int nslabs; };
template <int THREADS, bool RES2 = false>
__global__ __launch_bounds__(THREADS) void bn_bwd_stats_apply_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ y, int ldy,
                                                                  const float* __restrict__ dy, int lddy, float* __restrict__ dx, int lddx,
                                                                  float* __restrict__ dres, int lddr, int P, int C, int groups, int rows_per_slab,
                                                                  const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta, int relu, int training,
                                                                  const float* __restrict__ part, int nparts, unsigned* __restrict__ dx_amax, float ks, const BnRes2 r2) {
    constexpr int KG = THREADS / 32, RP = THREADS / 8;
    __shared__ double shm[KG][2][32];
    __shared__ float fin[2][32];
    const int grp = blockIdx.x % groups, slab = blockIdx.x / groups;
    const int tid = threadIdx.x, l8 = tid & 7, rr = tid >> 3;
    const int q = grp * 8 + l8;
    // first row and per-channel constants requested before the merge (see bn_stats_apply_kernel)
    const int row0 = slab * rows_per_slab, row1 = min(P, row0 + rows_per_slab), p0 = row0 + rr;
    float4 dv0 = make_float4(0.f, 0.f, 0.f, 0.f), xv0 = dv0, yv0 = make_float4(1.f, 1.f, 1.f, 1.f), x2v0 = dv0;
    if (p0 < row1) {
        dv0 = LD4(dy, p0, lddy, q); xv0 = LD4(x, p0, ldx, q);
        if (relu) yv0 = LD4(y, p0, ldy, q);
        if constexpr (RES2) x2v0 = LD4(r2.x, p0, r2.ldx, q);
    }
    float mu2[4] = {0.f, 0.f, 0.f, 0.f}, is2[4] = {0.f, 0.f, 0.f, 0.f}, sg2[4] = {0.f, 0.f, 0.f, 0.f}, sgx2[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (RES2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { mu2[j] = r2.mean[4 * q + j]; is2[j] = r2.invstd[4 * q + j]; }
    }
    float mu[4], is[4], gi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { mu[j] = mean[4 * q + j]; is[j] = invstd[4 * q + j]; gi[j] = gamma[4 * q + j]; }
    {
        const int ch = tid & 31, k = tid >> 5;
        const float* pa = part + grp * 32 + ch;
        const float* pb = pa + (long long)nparts * C;
        double a = 0, b = 0;
        struct Batch { float a[kMergeRows], b[kMergeRows]; };      // same schedule as the forward merge: rows k, k + KG, ... in order
        auto fetch = [&](int base, Batch& t) {
#pragma unroll
            for (int u = 0; u < kMergeRows; ++u) {
                const int sl = min(base + KG * u, nparts - 1);
                t.a[u] = pa[(long long)sl * C]; t.b[u] = pb[(long long)sl * C];
            }
        };
        auto sum = [&](int base, const Batch& t) {
#pragma unroll
            for (int u = 0; u < kMergeRows; ++u)
                if (base + KG * u < nparts) { a += t.a[u]; b += t.b[u]; }
        };
        Batch b0, b1;
        fetch(k, b0);
        for (int base = k; base < nparts; base += 2 * KG * kMergeRows) {
            const int mid = base + KG * kMergeRows;
            if (mid < nparts) fetch(mid, b1);
            sum(base, b0);
            if (mid < nparts) {
                if (mid + KG * kMergeRows < nparts) fetch(mid + KG * kMergeRows, b0);
                sum(mid, b1);
            }
        }
        shm[k][0][ch] = a; shm[k][1][ch] = b;
    }
    __syncthreads();
    if (tid < 32) {
        double a = 0, b = 0;
#pragma unroll
        for (int k = 0; k < KG; ++k) { a += shm[k][0][tid]; b += shm[k][1][tid]; }
        const float inv_n = 1.f / (float)P;
        fin[0][tid] = training ? (float)a * inv_n : 0.f; fin[1][tid] = training ? (float)b * inv_n : 0.f;
        if (slab == 0) {
            const int c = grp * 32 + tid;
            if (dbeta) dbeta[c] = (float)a;
            if (dgamma) dgamma[c] = (float)b;
        }
    }
    __syncthreads();
    float mb[4], mg[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { gi[j] *= is[j]; mb[j] = fin[0][4 * l8 + j]; mg[j] = fin[1][4 * l8 + j]; }
    unsigned am = 0u;
    auto row = [&](int p, const float4 dv, const float4 xv, const float4 yv, const float4 x2v) {
        // ks: 1 / (1 - p) of a Dropout behind the ReLU (the combined mask is y > 0), else 1 (exact)
        const float g0 = masked_grad(dv.x, yv.x, relu, 0.f, ks), g1 = masked_grad(dv.y, yv.y, relu, 0.f, ks);
        const float g2 = masked_grad(dv.z, yv.z, relu, 0.f, ks), g3 = masked_grad(dv.w, yv.w, relu, 0.f, ks);
        const float4 d = make_float4(gi[0] * (g0 - mb[0] - (xv.x - mu[0]) * is[0] * mg[0]), gi[1] * (g1 - mb[1] - (xv.y - mu[1]) * is[1] * mg[1]),
                                     gi[2] * (g2 - mb[2] - (xv.z - mu[2]) * is[2] * mg[2]), gi[3] * (g3 - mb[3] - (xv.w - mu[3]) * is[3] * mg[3]));
        ST4(dx, p, lddx, q) = d;
        am = abs_bits4(am, d.x, d.y, d.z, d.w);
        if (dres) ST4(dres, p, lddr, q) = make_float4(g0, g1, g2, g3);
        if constexpr (RES2) {
            sg2[0] += g0; sgx2[0] += g0 * ((x2v.x - mu2[0]) * is2[0]);
            sg2[1] += g1; sgx2[1] += g1 * ((x2v.y - mu2[1]) * is2[1]);
            sg2[2] += g2; sgx2[2] += g2 * ((x2v.z - mu2[2]) * is2[2]);
            sg2[3] += g3; sgx2[3] += g3 * ((x2v.w - mu2[3]) * is2[3]);
        }
    };
    if (p0 < row1) row(p0, dv0, xv0, yv0, x2v0);
#pragma unroll 2
    for (int p = p0 + RP; p < row1; p += RP) {
        const float4 dv = LD4(dy, p, lddy, q), xv = LD4(x, p, ldx, q);
        float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
        if (relu) yv = LD4(y, p, ldy, q);
        float4 x2v = xv;
        if constexpr (RES2) x2v = LD4(r2.x, p, r2.ldx, q);
        row(p, dv, xv, yv, x2v);
    }
    if constexpr (RES2) {
        // the block's sums per channel: lanes that share l8 (same four channels) within a wave, then the waves through LDS in the order 0 .. THREADS/64 - 1
        __shared__ float r2s[THREADS / 64][2][32];
#pragma unroll
        for (int sft = 8; sft < 64; sft <<= 1)
#pragma unroll
            for (int j = 0; j < 4; ++j) { sg2[j] += __shfl_xor(sg2[j], sft); sgx2[j] += __shfl_xor(sgx2[j], sft); }
        if ((tid & 63) < 8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { r2s[tid >> 6][0][4 * l8 + j] = sg2[j]; r2s[tid >> 6][1][4 * l8 + j] = sgx2[j]; }
        }
        __syncthreads();
        if (tid < 64) {
            const int v = tid >> 5, c = tid & 31;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < THREADS / 64; ++w) t += r2s[w][v][c];
            r2.part[((long long)v * r2.nslabs + slab) * C + grp * 32 + c] = t;
        }
    }
    amax_publish(am, dx_amax);
}

// ---------------------------------------------------------------------------------------------- dense tensors, C % 4 != 0 ("flat" kernels)
// A dense [P][C] tensor (ld == C) is one float array: float4 number f holds the elements 4f .. 4f+3 and element e belongs to channel e % C.
// A thread that strides by T float4 with 4T % C == 0 meets the same four channels in every float4 it loads, so the odd channel counts of the
// tail (19 classes at 256x512 and 512x1024, the one-channel feature transformers) stream with 16-byte loads like every other tensor; the
// one-float-per-lane kernels above remain for strided or unaligned operands.  Blocks of 512 threads, T = the largest multiple of
// C / gcd(C, 4) below 512; the per-thread sums share one shift per channel and block, so the block merge is a plain sum by channel.
constexpr int kFlatThreads = 512;
__device__ inline void flat_reduce(float (*sh)[4 * kFlatThreads], float (*sh2)[kFlatThreads], int nv, int C, int T, int phase0, float* out) {
    // sh[v][e], e = 4*thread + slot, belongs to channel (phase0 + e) % C; returns in out[v] (threads < C) the channel totals. Three levels in a
    // fixed order: G = 512 / C row slots per channel, then 16, then one - a one-channel tensor would otherwise leave 512 additions to one thread.
    const int t = threadIdx.x, G = kFlatThreads / C, c = t % C, r = t / C;
    __syncthreads();
    if (r < G) {
        const int e0 = (c - phase0 + C) % C;
        for (int v = 0; v < nv; ++v) {
            float a = 0.f;
            for (int e = e0 + C * r; e < 4 * T; e += C * G) a += sh[v][e];
            sh2[v][r * C + c] = a;
        }
    }
    __syncthreads();
    const int G2 = G > 16 ? 16 : G;
    if (G > 16 && r < 16) {
        for (int v = 0; v < nv; ++v) {
            float a = 0.f;
            for (int q = r; q < G; q += 16) a += sh2[v][q * C + c];
            sh[v][r * C + c] = a;
        }
    }
    __syncthreads();
    if (t < C)
        for (int v = 0; v < nv; ++v) {
            float a = 0.f;
            for (int q = 0; q < G2; ++q) a += (G > 16 ? sh[v][q * C + t] : sh2[v][q * C + t]);
            out[v] = a;
        }
}
__global__ __launch_bounds__(kFlatThreads) void bn_partial_flat_kernel(const float* __restrict__ x, long long total4, long long P, int C, int T,
                                                                      long long per_block4, float* __restrict__ part, int nbx) {
    __shared__ float sh[3][4 * kFlatThreads];
    __shared__ float sh2[3][kFlatThreads];
    const int t = threadIdx.x;
    const long long b0 = min(total4, blockIdx.x * per_block4), b1 = min(total4, b0 + per_block4);
    const long long r0 = min(P - 1, (4 * b0) / C);                         // a row of this block: its values are the shifts
    const int phase0 = (int)((4 * b0) % C);
    float n = 0.f, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (t < T) {
        float k0[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) k0[k] = x[r0 * C + (phase0 + 4 * t + k) % C];
        const float4* x4 = reinterpret_cast<const float4*>(x);
#pragma unroll 4
        for (long long f = b0 + t; f < b1; f += T) {
            const float4 v = x4[f];
            const float d0 = v.x - k0[0], d1 = v.y - k0[1], d2 = v.z - k0[2], d3 = v.w - k0[3];
            s1[0] += d0; s2[0] += d0 * d0; s1[1] += d1; s2[1] += d1 * d1; s1[2] += d2; s2[2] += d2 * d2; s1[3] += d3; s2[3] += d3 * d3;
            n += 1.f;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { sh[0][4 * t + k] = n; sh[1][4 * t + k] = s1[k]; sh[2][4 * t + k] = s2[k]; }
    float tot[3];
    flat_reduce(sh, sh2, 3, C, T, phase0, tot);
    if (t < C) {
        float mean = 0.f, m2 = 0.f;
        if (tot[0] > 0.f) { mean = x[r0 * C + t] + tot[1] / tot[0]; m2 = fmaxf(tot[2] - tot[1] * tot[1] / tot[0], 0.f); }
        const long long o = (long long)blockIdx.x * C + t;
        part[o] = tot[0]; part[(long long)nbx * C + o] = mean; part[2ll * nbx * C + o] = m2;
    }
}
__global__ __launch_bounds__(kFlatThreads) void bn_bwd_partial_flat_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                                                          long long total4, int C, int T, long long per_block4,
                                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                          int relu, float drop_p, float* __restrict__ part, int nbx) {
    __shared__ float sh[3][4 * kFlatThreads];
    __shared__ float sh2[3][kFlatThreads];
    const int t = threadIdx.x;
    const long long b0 = min(total4, blockIdx.x * per_block4), b1 = min(total4, b0 + per_block4);
    const int phase0 = (int)((4 * b0) % C);
    float sg[4] = {0.f, 0.f, 0.f, 0.f}, sgx[4] = {0.f, 0.f, 0.f, 0.f};
    if (t < T) {
        float mu[4], is[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int c = (phase0 + 4 * t + k) % C; mu[k] = mean[c]; is[k] = invstd[c]; }
        const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
        const bool need_y = relu || drop_p > 0.f;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const float4* y4 = reinterpret_cast<const float4*>(y);
        const float4* d4 = reinterpret_cast<const float4*>(dy);
#pragma unroll 2
        for (long long f = b0 + t; f < b1; f += T) {
            const float4 dv = d4[f], xv = x4[f];
            const float4 yv = need_y ? y4[f] : make_float4(1.f, 1.f, 1.f, 1.f);
            const float g0 = masked_grad(dv.x, yv.x, relu, drop_p, ks), g1 = masked_grad(dv.y, yv.y, relu, drop_p, ks);
            const float g2 = masked_grad(dv.z, yv.z, relu, drop_p, ks), g3 = masked_grad(dv.w, yv.w, relu, drop_p, ks);
            sg[0] += g0; sgx[0] += g0 * ((xv.x - mu[0]) * is[0]); sg[1] += g1; sgx[1] += g1 * ((xv.y - mu[1]) * is[1]);
            sg[2] += g2; sgx[2] += g2 * ((xv.z - mu[2]) * is[2]); sg[3] += g3; sgx[3] += g3 * ((xv.w - mu[3]) * is[3]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { sh[0][4 * t + k] = sg[k]; sh[1][4 * t + k] = sgx[k]; }
    float tot[2];
    flat_reduce(sh, sh2, 2, C, T, phase0, tot);
    if (t < C) {
        const long long o = (long long)blockIdx.x * C + t;
        part[o] = tot[0]; part[(long long)nbx * C + o] = tot[1];
    }
}
__global__ __launch_bounds__(kFlatThreads) void bn_apply_flat_kernel(const float* __restrict__ x, float* __restrict__ y, long long total4, int C, int T,
                                                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    const float* __restrict__ res, int relu, float drop_p, SeedArg seed_arg, unsigned rng_stream,
                                                                    unsigned* __restrict__ y_amax) {
    const unsigned long long seed = seed_arg.dev ? *seed_arg.dev : seed_arg.value;
    const int t = threadIdx.x;
    unsigned am = 0u;
    const long long f0 = t < T ? (long long)blockIdx.x * T + t : total4, stride = (long long)gridDim.x * T;      // 4 * stride % C == 0: the channels of a thread never change
    float sc[4], sf[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int c = (int)((4 * f0 + k) % C); sc[k] = gamma[c] * invstd[c]; sf[k] = beta[c] - mean[c] * sc[k]; }
    const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* r4 = reinterpret_cast<const float4*>(res);
    float4* y4 = reinterpret_cast<float4*>(y);
#pragma unroll 2
    for (long long f = f0; f < total4; f += stride) {
        const float4 xv = x4[f];
        float v[4] = {fmaf(xv.x, sc[0], sf[0]), fmaf(xv.y, sc[1], sf[1]), fmaf(xv.z, sc[2], sf[2]), fmaf(xv.w, sc[3], sf[3])};
        if (res) { const float4 r = r4[f]; v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
        if (relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        if (drop_p > 0.f) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (philox_uniform((unsigned long long)(4 * f + k), seed, rng_stream) >= drop_p) ? v[k] * ks : 0.f;
        }
        y4[f] = make_float4(v[0], v[1], v[2], v[3]);
        am = abs_bits4(am, v[0], v[1], v[2], v[3]);
    }
    amax_publish(am, y_amax);
}
// column sums of a [P][C] tensor with C % 4 == 0: one float4 per thread and row (colsum_partial_kernel reads one float per lane)
__global__ __launch_bounds__(256) void colsum_partial4_kernel(const float* __restrict__ x, int ld, long long P, int C, long long rows_per_block,
                                                               float* __restrict__ part) {
    __shared__ float sh[4][256];
    const ChanMap4 m = chan_map4(C, blockIdx.y);
    const long long row0 = blockIdx.x * rows_per_block, row1 = min(P, row0 + rows_per_block);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (m.q >= 0) {
#pragma unroll 4
        for (long long p = row0 + m.slot; p < row1; p += m.G) { const float4 v = LD4(x, p, ld, m.q); s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) sh[j][threadIdx.x] = s[j];
    __syncthreads();
    if (m.q >= 0 && m.slot == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = s[j];
            for (int g = 1; g < m.G; ++g) a += sh[j][g * m.cq + (m.q - m.q0)];
            part[(long long)blockIdx.x * C + 4 * m.q + j] = a;
        }
    }
}

// flat kernels apply to dense operands only: every pixel stride equals C, 16-byte aligned bases, P*C a multiple of 4
static int flat_stride(int C, int64_t P, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs) {
    if (C % 4 == 0 || (P * C) % 4) return 0;
    for (int l : lds) if (l != C) return 0;
    for (const void* p : ptrs) if (p && ((uintptr_t)p % 16)) return 0;
    int g = 1;
    if (C % 2 == 0) g = 2;
    const int cp = C / g;
    return cp <= kFlatThreads / 2 ? kFlatThreads - kFlatThreads % cp : 0;
}
static bool vec4_ok(int C, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs) {
    if (C % 4) return false;
    for (int l : lds) if (l % 4) return false;
    for (const void* p : ptrs) if (p && ((uintptr_t)p % 16)) return false;
    return true;
}
static dim3 apply_grid4(int64_t P, int C) {
    const int Q = C / 4, groups = (int)ceil_div(Q, 256);
    const int cq = std::min(Q, 256), G = 256 / cq;
    int64_t bx = std::min<int64_t>(ceil_div(P, (int64_t)G * 2), std::max(1, 2048 / groups));
    return dim3((unsigned)std::max<int64_t>(1, bx), (unsigned)groups);
}

static dim3 apply_grid(int64_t P, int C) {
    const int groups = (int)ceil_div(C, 256);
    const int cg = std::min(C, 256), G = 256 / cg;
    int64_t bx = std::min<int64_t>(ceil_div(P, (int64_t)G * 4), std::max(1, 2048 / groups));
    return dim3((unsigned)std::max<int64_t>(1, bx), (unsigned)groups);
}

}  // namespace dsrl
using namespace dsrl;

// row_blocks(P) partial triples of the three-kernel path or up to 256 slab partials of the fused path, plus the two sum rows
extern "C" size_t dsrl_bn_workspace_bytes(int64_t P, int C) { return (size_t)(3 * (size_t)std::max(row_blocks(P), 256) + 2) * C * sizeof(float); }

namespace dsrl {
// Dropout key of a launch: the `seed` argument of the call, or - once dsrl_rng_bind_device_key() has bound one for the device - the
// 64-bit word at that device address, read by the kernel when it runs (so that a captured launch sees a fresh key on every replay).
static std::atomic<const unsigned long long*> g_rng_dev_key[64];
static SeedArg seed_arg(uint64_t seed) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    return SeedArg{(unsigned long long)seed, g_rng_dev_key[dev].load()};
}
static int env_int_bn(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
// The from-statistics kernels: every block merges the partials of its channel group again, and what that costs is the bytes a CU pulls through its L1
// (128 partials x 4 blocks of 256 threads per CU: 3.4 us on a 3.7 us launch; 2 blocks of 512: 1.8 us; 1 block of 1024: 0.9 us - but the streaming part
// of a 1024-thread block is 0.3 us slower on a 4 MB tensor and 1.5 us on a 16 MB one, profiles/round5_bn_prologue_probe.txt).  So: 512 threads, and
// 1024 where there are many partials over a small tensor.  DSRL_BN_APPLY_THREADS = 256 | 512 | 1024 forces one size.
static int stats_apply_threads(int64_t P, int C, int nparts) {
    static const int forced = [] { const int v = env_int_bn("DSRL_BN_APPLY_THREADS", 0); return v == 256 || v == 512 || v == 1024 ? v : 0; }();
    if (forced) return forced;
    return nparts >= 128 && P * C <= (2ll << 20) ? 1024 : 512;
}
static int stats_apply_rows(int64_t P, int groups, int threads) {        // rows per slab: ~(4 * 256 * 256 / threads) blocks, at least one pass of rows each
    const int rp = threads / 8;
    const int64_t slabs = std::max<int64_t>(1, std::min<int64_t>(ceil_div(P, rp), ceil_div((int64_t)4 * 256 * 256 / threads, groups)));
    return (int)ceil_div(P, slabs);
}
static std::atomic<int> g_fused_max_blocks{-1};      // dsrl_bn_fused_max_blocks(); -1 = DSRL_BN_FUSED / DSRL_BN_FUSED_BIG from the environment
struct FusedPlan { bool ok; int blocks, groups, slabs, rows_per_slab; };
// eligible: C a power-of-two multiple of 32 and the tensor fits the registers of 128 (P*C <= 4.2 M elements) or 256 blocks (8.4 M)
static FusedPlan fused_plan(int64_t P, int C) {
    FusedPlan f{false, 0, 0, 0, 0};
    int max_blocks = g_fused_max_blocks.load();
    if (max_blocks < 0) max_blocks = !env_int_bn("DSRL_BN_FUSED", 1) ? 0 : (env_int_bn("DSRL_BN_FUSED_BIG", 1) ? kFusedBlocksBig : kFusedBlocks);
    if (max_blocks < kFusedBlocks || C < 32 || C % 32 || P >= (1ll << 30)) return f;
    const int groups = C / 32;
    for (int blocks : {kFusedBlocks, kFusedBlocksBig}) {
        if (blocks > max_blocks || groups > blocks || blocks % groups) continue;
        const int slabs = blocks / groups;
        const int64_t rows = ceil_div(P, (int64_t)slabs);
        if (rows > kFusedRL * kFusedMaxPasses) continue;
        f.ok = true; f.blocks = blocks; f.groups = groups; f.slabs = slabs; f.rows_per_slab = (int)rows;
        return f;
    }
    return f;
}
// The fused kernels share one per-device barrier word pair, so two of them must never be in flight together: they are serialised by
// running on ONE stream per device.  The first fused launch of a device pins its stream; a launch that arrives on another stream
// (outside graph capture, where the capture's own stream stands in for it) takes the three-kernel path instead.
static std::mutex g_fused_mu;
static hipStream_t g_fused_stream[64];
static bool g_fused_stream_set[64] = {false};
static unsigned long long g_fused_capture_id[64] = {0};
static hipStream_t g_fused_capture_stream[64];
static bool fused_stream_ok(hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    if (hipStreamGetCaptureInfo(st, &cs, &id) == hipSuccess && cs == hipStreamCaptureStatusActive) {
        // A capture orders only the nodes of ONE stream: a forked capture (DSRL_GRAPH_OVERLAP=1) could place two barrier launches on parallel
        // branches that share the word pair.  The first fused launch of a capture pins its stream for that capture; launches the same capture
        // records on another stream take the three-kernel path.
        std::lock_guard<std::mutex> lock(g_fused_mu);
        if (g_fused_capture_id[dev] != id) { g_fused_capture_id[dev] = id; g_fused_capture_stream[dev] = st; }
        return g_fused_capture_stream[dev] == st;
    }
    std::lock_guard<std::mutex> lock(g_fused_mu);
    if (!g_fused_stream_set[dev]) { g_fused_stream[dev] = st; g_fused_stream_set[dev] = true; }
    return g_fused_stream[dev] == st;
}
}  // namespace dsrl

extern "C" size_t dsrl_colsum_workspace_bytes(int64_t P, int C) { return (size_t)row_blocks(P) * C * sizeof(float); }
// floats of a partials buffer of `rows` sums (3: forward statistics, 2: backward sums) x `parts` row blocks x C channels, with the room the
// BatchNorm kernels need behind it to reduce more than 256 row blocks
extern "C" size_t dsrl_bn_stats_floats(int rows, int parts, int C) {
    return (size_t)rows * (size_t)(parts + (parts > 256 ? kStatsReduced : 0)) * (size_t)C;
}

extern "C" int dsrl_bn_stats(const float* x, int ldx, int64_t P, int C, float eps, float momentum, float* mean, float* invstd,
                             float* running_mean, float* running_var, void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    DSRL_REQUIRE(x && mean && invstd && ws && P > 0 && C > 0 && ldx >= C, DSRL_E_BADARG, "bn_stats: bad arguments (P=%lld C=%d ldx=%d)", (long long)P, C, ldx);
    DSRL_REQUIRE(ws_bytes >= dsrl_bn_workspace_bytes(P, C), DSRL_E_WORKSPACE, "bn_stats: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    const int nbx = row_blocks(P);
    const long long rpb = ceil_div(P, nbx);
    if (vec4_ok(C, {ldx}, {x}))
        hipLaunchKernelGGL(bn_partial4_kernel, dim3(nbx, (unsigned)ceil_div(C / 4, 256)), dim3(256), 0, st, x, ldx, (long long)P, C, rpb, (float*)ws, nbx);
    else if (const int T = flat_stride(C, P, {ldx}, {x}))
        hipLaunchKernelGGL(bn_partial_flat_kernel, dim3(nbx), dim3(kFlatThreads), 0, st, x, (long long)(P * C / 4), (long long)P, C, T,
                           (long long)ceil_div(P * C / 4, (int64_t)nbx), (float*)ws, nbx);
    else
        hipLaunchKernelGGL(bn_partial_kernel, dim3(nbx, (unsigned)ceil_div(C, 256)), dim3(256), 0, st, x, ldx, (long long)P, C, rpb, (float*)ws, nbx);
    if (int e = launch_status("bn_partial_kernel")) return e;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)ceil_div(C, 8)), dim3(256), 0, st, (const float*)ws, nbx, C, eps, momentum, mean, invstd, running_mean, running_var);
    return launch_status("bn_finalize_kernel");
}

extern "C" int dsrl_bn_invstd_from_var(const float* running_var, int C, float eps, float* invstd, dsrl_stream_t stream) {
    DSRL_REQUIRE(running_var && invstd && C > 0, DSRL_E_BADARG, "bn_invstd_from_var: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    hipLaunchKernelGGL(invstd_from_var_kernel, dim3((unsigned)ceil_div(C, 256)), dim3(256), 0, st, running_var, C, eps, invstd);
    return launch_status("invstd_from_var_kernel");
}

extern "C" int dsrl_bn_apply(const float* x, int ldx, float* y, int ldy, int64_t P, int C, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, const float* residual, int ldr, int relu, float drop_p,
                             uint64_t seed, uint32_t rng_stream, uint32_t* y_amax, dsrl_stream_t stream) {
    DSRL_REQUIRE(x && y && mean && invstd && gamma && beta && P > 0 && C > 0 && ldx >= C && ldy >= C, DSRL_E_BADARG, "bn_apply: bad arguments");
    DSRL_REQUIRE(drop_p >= 0.f && drop_p < 1.f, DSRL_E_BADARG, "bn_apply: dropout p=%f outside [0,1)", drop_p);
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    if (vec4_ok(C, {ldx, ldy, residual ? ldr : 0}, {x, y, residual}))
        hipLaunchKernelGGL(bn_apply4_kernel, apply_grid4(P, C), dim3(256), 0, st, x, ldx, y, ldy, (long long)P, C, mean, invstd, gamma, beta, residual, ldr,
                           relu, drop_p, seed_arg(seed), (unsigned)rng_stream, y_amax);
    else if (const int T = flat_stride(C, P, {ldx, ldy, residual ? ldr : C}, {x, y, residual}))
        hipLaunchKernelGGL(bn_apply_flat_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(2048, ceil_div(P * C / 4, (int64_t)T * 2)))), dim3(kFlatThreads), 0, st,
                           x, y, (long long)(P * C / 4), C, T, mean, invstd, gamma, beta, residual, relu, drop_p, seed_arg(seed), (unsigned)rng_stream, y_amax);
    else
        hipLaunchKernelGGL(bn_apply_kernel, apply_grid(P, C), dim3(256), 0, st, x, ldx, y, ldy, (long long)P, C, mean, invstd, gamma, beta, residual, ldr,
                           relu, drop_p, seed_arg(seed), (unsigned)rng_stream, y_amax);
    return launch_status("bn_apply_kernel");
}

extern "C" int dsrl_bn_fused_max_blocks(int max_blocks) {
    const int prev = g_fused_max_blocks.load();
    if (max_blocks >= -1) g_fused_max_blocks.store(max_blocks);
    return prev;
}

extern "C" int dsrl_bn_fused_barrier_timeouts(int64_t* count) {
    DSRL_REQUIRE(count, DSRL_E_BADARG, "bn_fused_barrier_timeouts: null pointer");
    unsigned int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_grid_timeouts), sizeof(v)) != hipSuccess) return launch_status("hipMemcpyFromSymbol(g_grid_timeouts)");
    *count = (int64_t)v;        // cumulative since the library was loaded
    static unsigned int seen[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (v != seen[dev]) {       // new timeouts since the last query: a block that gave up left the arrival count behind - start the next barrier clean
        const unsigned int zero = 0u;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_grid_count), &zero, sizeof(zero)) != hipSuccess) return launch_status("hipMemcpyToSymbol(g_grid_count)");
        seen[dev] = v;
    }
    return DSRL_OK;
}

extern "C" int dsrl_bn_train_fwd(const float* x, int ldx, float* y, int ldy, int64_t P, int C, float eps, float momentum, float* mean, float* invstd,
                                 float* running_mean, float* running_var, const float* gamma, const float* beta, const float* residual, int ldr,
                                 int relu, float drop_p, uint64_t seed, uint32_t rng_stream, void* ws, size_t ws_bytes, uint32_t* y_amax, dsrl_stream_t stream) {
    DSRL_REQUIRE(x && y && mean && invstd && gamma && beta && ws && P > 0 && C > 0 && ldx >= C && ldy >= C, DSRL_E_BADARG, "bn_train_fwd: bad arguments");
    DSRL_REQUIRE(drop_p >= 0.f && drop_p < 1.f, DSRL_E_BADARG, "bn_train_fwd: dropout p=%f outside [0,1)", drop_p);
    DSRL_REQUIRE(ws_bytes >= dsrl_bn_workspace_bytes(P, C), DSRL_E_WORKSPACE, "bn_train_fwd: workspace too small");
    const FusedPlan f = fused_plan(P, C);
    if (f.ok && vec4_ok(C, {ldx, ldy, residual ? ldr : 0}, {x, y, residual}) && bind_stream_device((hipStream_t)stream) == DSRL_OK &&
        fused_stream_ok((hipStream_t)stream)) {
        hipStream_t st = (hipStream_t)stream;
        hipLaunchKernelGGL(bn_fused_fwd_kernel, dim3(f.blocks), dim3(kFusedThreads), 0, st, x, ldx, y, ldy, (int)P, C, f.groups, f.slabs, f.rows_per_slab, eps, momentum,
                           mean, invstd, running_mean, running_var, gamma, beta, residual, ldr, relu, drop_p, seed_arg(seed), (unsigned)rng_stream,
                           (float*)ws, y_amax);
        return launch_status("bn_fused_fwd_kernel");
    }
    if (int e = dsrl_bn_stats(x, ldx, P, C, eps, momentum, mean, invstd, running_mean, running_var, ws, ws_bytes, stream)) return e;
    return dsrl_bn_apply(x, ldx, y, ldy, P, C, mean, invstd, gamma, beta, residual, ldr, relu, drop_p, seed, rng_stream, y_amax, stream);
}

extern "C" int dsrl_bn_train_fwd_from_stats(const float* x, int ldx, float* y, int ldy, int64_t P, int C, float eps, float momentum, float* mean, float* invstd,
                                            float* running_mean, float* running_var, const float* gamma, const float* beta, const float* residual, int ldr,
                                            int relu, float drop_p, uint64_t seed, uint32_t rng_stream, float* stats, int stats_parts, uint32_t* y_amax,
                                            dsrl_stream_t stream) {
    DSRL_REQUIRE(x && y && mean && invstd && gamma && beta && stats && P > 0 && P < (1ll << 31) && C > 0 && ldx >= C && ldy >= C, DSRL_E_BADARG, "bn_train_fwd_from_stats: bad arguments");
    DSRL_REQUIRE(stats_parts > 0 && stats_parts <= 4096, DSRL_E_BADARG, "bn_train_fwd_from_stats: %d row blocks of partials (1..4096)", stats_parts);
    DSRL_REQUIRE(C % 32 == 0 && vec4_ok(C, {ldx, ldy, residual ? ldr : 0}, {x, y, residual}), DSRL_E_UNSUPPORTED,
                 "bn_train_fwd_from_stats: C (%d) must be a multiple of 32, strides multiples of 4, pointers 16-byte aligned", C);
    DSRL_REQUIRE(drop_p >= 0.f && drop_p < 1.f, DSRL_E_BADARG, "bn_train_fwd_from_stats: dropout p=%f outside [0,1)", drop_p);
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    const int groups = C / 32;
    if (stats_parts > 256) {            // reduced rows behind the partials (dsrl_bn_stats_floats)
        float* red = stats + 3ll * stats_parts * C;
        hipLaunchKernelGGL(bn_stats_reduce_kernel, dim3((unsigned)groups, kStatsReduced), dim3(256), 0, st, (const float*)stats, stats_parts, C, red);
        if (int e = launch_status("bn_stats_reduce_kernel")) return e;
        stats = red; stats_parts = kStatsReduced;
    }
    const int T = stats_apply_threads(P, C, stats_parts), rows_per_slab = stats_apply_rows(P, groups, T), slabs = (int)ceil_div(P, (int64_t)rows_per_slab);
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3((unsigned)(groups * slabs)), dim3(T), 0, st, x, ldx, y, ldy, (int)P, C, groups, rows_per_slab, eps, momentum,
                           mean, invstd, running_mean, running_var, gamma, beta, residual, ldr, relu, drop_p, seed_arg(seed), (unsigned)rng_stream, stats, stats_parts, y_amax);
    };
    if (T == 1024) go(bn_stats_apply_kernel<1024>); else if (T == 512) go(bn_stats_apply_kernel<512>); else go(bn_stats_apply_kernel<256>);
    return launch_status("bn_stats_apply_kernel");
}

extern "C" int dsrl_bn_bwd(const float* x, int ldx, const float* y, int ldy, const float* dy, int lddy, float* dx, int lddx,
                           float* dresidual, int lddr, int64_t P, int C, const float* mean, const float* invstd, const float* gamma,
                           float* dgamma, float* dbeta, int relu, float drop_p, int training, void* ws, size_t ws_bytes, uint32_t* dx_amax, dsrl_stream_t stream) {
    DSRL_REQUIRE(x && dy && dx && mean && invstd && gamma && ws && P > 0 && C > 0, DSRL_E_BADARG, "bn_bwd: bad arguments");
    DSRL_REQUIRE(y || !(relu || drop_p > 0.f), DSRL_E_BADARG, "bn_bwd: forward output needed for the relu/dropout mask");
    DSRL_REQUIRE(ws_bytes >= dsrl_bn_workspace_bytes(P, C), DSRL_E_WORKSPACE, "bn_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    const bool v4 = vec4_ok(C, {ldx, y ? ldy : 0, lddy, lddx, dresidual ? lddr : 0}, {x, y, dy, dx, dresidual});
    const FusedPlan f = fused_plan(P, C);
    if (f.ok && v4 && fused_stream_ok(st)) {
        hipLaunchKernelGGL(bn_fused_bwd_kernel, dim3(f.blocks), dim3(kFusedThreads), 0, st, x, ldx, y, ldy, dy, lddy, dx, lddx, dresidual, lddr, (int)P, C,
                           f.groups, f.slabs, f.rows_per_slab, mean, invstd, gamma, dgamma, dbeta, relu, drop_p, training, (float*)ws, dx_amax);
        return launch_status("bn_fused_bwd_kernel");
    }
    const int nbx = row_blocks(P);
    const long long rpb = ceil_div(P, nbx);
    float* part = (float*)ws;
    float* sums = part + 3ll * nbx * C;
    if (v4)
        hipLaunchKernelGGL(bn_bwd_partial4_kernel, dim3(nbx, (unsigned)ceil_div(C / 4, 256)), dim3(256), 0, st, x, ldx, y, ldy, dy, lddy, (long long)P, C, rpb,
                           mean, invstd, relu, drop_p, part, nbx);
    else if (const int T = flat_stride(C, P, {ldx, y ? ldy : C, lddy}, {x, y, dy}))
        hipLaunchKernelGGL(bn_bwd_partial_flat_kernel, dim3(nbx), dim3(kFlatThreads), 0, st, x, y, dy, (long long)(P * C / 4), C, T,
                           (long long)ceil_div(P * C / 4, (int64_t)nbx), mean, invstd, relu, drop_p, part, nbx);
    else
        hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nbx, (unsigned)ceil_div(C, 256)), dim3(256), 0, st, x, ldx, y, ldy, dy, lddy, (long long)P, C, rpb,
                           mean, invstd, relu, drop_p, part, nbx);
    if (int e = launch_status("bn_bwd_partial_kernel")) return e;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)ceil_div(C, 8)), dim3(256), 0, st, (const float*)part, nbx, C, sums, dgamma, dbeta);
    if (int e = launch_status("bn_bwd_finalize_kernel")) return e;
    if (v4)
        hipLaunchKernelGGL(bn_bwd_apply4_kernel, apply_grid4(P, C), dim3(256), 0, st, x, ldx, y, ldy, dy, lddy, dx, lddx, dresidual, lddr, (long long)P, C,
                           mean, invstd, gamma, (const float*)sums, relu, drop_p, training, dx_amax);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel, apply_grid(P, C), dim3(256), 0, st, x, ldx, y, ldy, dy, lddy, dx, lddx, dresidual, lddr, (long long)P, C,
                           mean, invstd, gamma, (const float*)sums, relu, drop_p, training, dx_amax);
    return launch_status("bn_bwd_apply_kernel");
}

extern "C" int dsrl_bn_bwd_from_stats(const float* x, int ldx, const float* y, int ldy, const float* dy, int lddy, float* dx, int lddx,
                                      float* dresidual, int lddr, int64_t P, int C, const float* mean, const float* invstd, const float* gamma,
                                      float* dgamma, float* dbeta, int relu, int training, float* stats, int stats_parts, uint32_t* dx_amax,
                                      dsrl_stream_t stream) {
    return dsrl_bn_bwd_from_stats_drop(x, ldx, y, ldy, dy, lddy, dx, lddx, dresidual, lddr, P, C, mean, invstd, gamma, dgamma, dbeta, relu, 0.f, training, stats, stats_parts,
                                       dx_amax, stream);
}
// slabs (= row blocks of the residual branch's partial sums) the from-statistics backward of a [P][C] tensor launches when handed stats_parts partials
static int bwd_from_stats_slabs(int64_t P, int C, int stats_parts) {
    const int parts = stats_parts > 256 ? kStatsReduced : stats_parts, groups = C / 32, T = stats_apply_threads(P, C, parts);
    return (int)ceil_div(P, (int64_t)stats_apply_rows(P, groups, T));
}
extern "C" int dsrl_bn_bwd_from_stats_res_parts(int64_t P, int C, int stats_parts) {
    return (P > 0 && P < (1ll << 31) && C > 0 && C % 32 == 0 && stats_parts > 0 && stats_parts <= 4096) ? bwd_from_stats_slabs(P, C, stats_parts) : 0;
}
static int bn_bwd_from_stats_impl(const float* x, int ldx, const float* y, int ldy, const float* dy, int lddy, float* dx, int lddx,
                                  float* dresidual, int lddr, int64_t P, int C, const float* mean, const float* invstd, const float* gamma,
                                  float* dgamma, float* dbeta, int relu, float drop_p, int training, float* stats, int stats_parts, uint32_t* dx_amax,
                                  dsrl_stream_t stream, const BnRes2* res2);
extern "C" int dsrl_bn_bwd_from_stats_drop(const float* x, int ldx, const float* y, int ldy, const float* dy, int lddy, float* dx, int lddx,
                                           float* dresidual, int lddr, int64_t P, int C, const float* mean, const float* invstd, const float* gamma,
                                           float* dgamma, float* dbeta, int relu, float drop_p, int training, float* stats, int stats_parts, uint32_t* dx_amax,
                                           dsrl_stream_t stream) {
    return bn_bwd_from_stats_impl(x, ldx, y, ldy, dy, lddy, dx, lddx, dresidual, lddr, P, C, mean, invstd, gamma, dgamma, dbeta, relu, drop_p, training, stats, stats_parts,
                                  dx_amax, stream, nullptr);
}
extern "C" int dsrl_bn_bwd_from_stats_res(const float* x, int ldx, const float* y, int ldy, const float* dy, int lddy, float* dx, int lddx,
                                          float* dresidual, int lddr, int64_t P, int C, const float* mean, const float* invstd, const float* gamma,
                                          float* dgamma, float* dbeta, int relu, float drop_p, int training, float* stats, int stats_parts, uint32_t* dx_amax,
                                          const float* res_x, int res_ldx, const float* res_mean, const float* res_invstd, float* res_stats, int res_parts,
                                          dsrl_stream_t stream) {
    DSRL_REQUIRE(dresidual && res_x && res_mean && res_invstd && res_stats && res_ldx >= C && res_ldx % 4 == 0 && ((uintptr_t)res_x % 16) == 0, DSRL_E_BADARG,
                 "bn_bwd_from_stats_res: the residual branch's input / statistics / partials buffer are required (stride a multiple of 4, 16-byte aligned)");
    DSRL_REQUIRE(res_parts > 0 && res_parts == dsrl_bn_bwd_from_stats_res_parts(P, C, stats_parts), DSRL_E_BADARG,
                 "bn_bwd_from_stats_res: this launch writes %d row blocks of sums, the caller expects %d (dsrl_bn_bwd_from_stats_res_parts)",
                 dsrl_bn_bwd_from_stats_res_parts(P, C, stats_parts), res_parts);
    const BnRes2 r2{res_x, res_ldx, res_mean, res_invstd, res_stats, res_parts};
    return bn_bwd_from_stats_impl(x, ldx, y, ldy, dy, lddy, dx, lddx, dresidual, lddr, P, C, mean, invstd, gamma, dgamma, dbeta, relu, drop_p, training, stats, stats_parts,
                                  dx_amax, stream, &r2);
}
static int bn_bwd_from_stats_impl(const float* x, int ldx, const float* y, int ldy, const float* dy, int lddy, float* dx, int lddx,
                                  float* dresidual, int lddr, int64_t P, int C, const float* mean, const float* invstd, const float* gamma,
                                  float* dgamma, float* dbeta, int relu, float drop_p, int training, float* stats, int stats_parts, uint32_t* dx_amax,
                                  dsrl_stream_t stream, const BnRes2* res2) {
    DSRL_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || relu), DSRL_E_BADARG, "bn_bwd_from_stats_drop: dropout p=%f needs the ReLU mask (y > 0)", drop_p);
    const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    DSRL_REQUIRE(x && dy && dx && mean && invstd && gamma && stats && P > 0 && P < (1ll << 31) && C > 0, DSRL_E_BADARG, "bn_bwd_from_stats: bad arguments");
    DSRL_REQUIRE(y || !relu, DSRL_E_BADARG, "bn_bwd_from_stats: forward output needed for the relu mask");
    DSRL_REQUIRE(stats_parts > 0 && stats_parts <= 4096, DSRL_E_BADARG, "bn_bwd_from_stats: %d row blocks of partials (1..4096)", stats_parts);
    DSRL_REQUIRE(C % 32 == 0 && vec4_ok(C, {ldx, y ? ldy : 0, lddy, lddx, dresidual ? lddr : 0}, {x, y, dy, dx, dresidual}), DSRL_E_UNSUPPORTED,
                 "bn_bwd_from_stats: C (%d) must be a multiple of 32, strides multiples of 4, pointers 16-byte aligned", C);
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    const int groups = C / 32;
    if (stats_parts > 256) {
        float* red = stats + 2ll * stats_parts * C;
        hipLaunchKernelGGL(bn_bwd_stats_reduce_kernel, dim3((unsigned)groups, kStatsReduced), dim3(256), 0, st, (const float*)stats, stats_parts, C, red);
        if (int e = launch_status("bn_bwd_stats_reduce_kernel")) return e;
        stats = red; stats_parts = kStatsReduced;
    }
    const int T = stats_apply_threads(P, C, stats_parts), rows_per_slab = stats_apply_rows(P, groups, T), slabs = (int)ceil_div(P, (int64_t)rows_per_slab);
    BnRes2 r2{};
    if (res2) { r2 = *res2; DSRL_REQUIRE(r2.nslabs == slabs, DSRL_E_BADARG, "bn_bwd_from_stats_res: %d slabs launched, %d expected", slabs, r2.nslabs); }
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3((unsigned)(groups * slabs)), dim3(T), 0, st, x, ldx, y, ldy, dy, lddy, dx, lddx, dresidual, lddr, (int)P, C,
                           groups, rows_per_slab, mean, invstd, gamma, dgamma, dbeta, relu, training, stats, stats_parts, dx_amax, ks, r2);
    };
    if (res2) { if (T == 1024) go(bn_bwd_stats_apply_kernel<1024, true>); else if (T == 512) go(bn_bwd_stats_apply_kernel<512, true>); else go(bn_bwd_stats_apply_kernel<256, true>); }
    else if (T == 1024) go(bn_bwd_stats_apply_kernel<1024>); else if (T == 512) go(bn_bwd_stats_apply_kernel<512>); else go(bn_bwd_stats_apply_kernel<256>);
    return launch_status("bn_bwd_stats_apply_kernel");
}

extern "C" int dsrl_colsum(const float* x, int ld, int64_t P, int C, float* out, void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    DSRL_REQUIRE(x && out && ws && P > 0 && C > 0 && ld >= C, DSRL_E_BADARG, "colsum: bad arguments");
    DSRL_REQUIRE(ws_bytes >= dsrl_colsum_workspace_bytes(P, C), DSRL_E_WORKSPACE, "colsum: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    const int nbx = row_blocks(P);
    if (vec4_ok(C, {ld}, {x}))
        hipLaunchKernelGGL(colsum_partial4_kernel, dim3(nbx, (unsigned)ceil_div(C / 4, 256)), dim3(256), 0, st, x, ld, (long long)P, C, (long long)ceil_div(P, nbx), (float*)ws);
    else
        hipLaunchKernelGGL(colsum_partial_kernel, dim3(nbx, (unsigned)ceil_div(C, 256)), dim3(256), 0, st, x, ld, (long long)P, C, (long long)ceil_div(P, nbx), (float*)ws);
    if (int e = launch_status("colsum_partial_kernel")) return e;
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((unsigned)ceil_div(C, 32)), dim3(256), 0, st, (const float*)ws, nbx, C, out);
    return launch_status("colsum_finalize_kernel");
}

__global__ void rng_advance_kernel(unsigned long long* state) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { state[1] += 1ull; state[0] = state[2] * 1000003ull + state[1]; }
}
extern "C" int dsrl_rng_bind_device_key(const uint64_t* dev_key) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { set_error("rng_bind_device_key: no current HIP device"); (void)hipGetLastError(); return DSRL_E_LAUNCH; }
    g_rng_dev_key[dev].store((const unsigned long long*)dev_key);
    return DSRL_OK;
}
extern "C" int dsrl_rng_advance_key(uint64_t* state, dsrl_stream_t stream) {
    DSRL_REQUIRE(state && ((uintptr_t)state % 8) == 0, DSRL_E_BADARG, "rng_advance_key: state must be three 8-byte aligned 64-bit words on the device");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, st, (unsigned long long*)state);
    return launch_status("rng_advance_kernel");
}

namespace dsrl {
// dense tensors of any width (the 19-channel Dropout of upsample16_pred, DSRL.py:55): the tensor is one float array, element e draws word e & 3 of
// philox(e >> 2) whatever the channel count - so float4 number i takes the four words of ONE Philox call, with 16-byte accesses and no index division.
// Same draws, same products as dropout_kernel (which pays a Philox call and a 64-bit division per ELEMENT: 18.8 -> 7 us on the 20 MB tensor).
__global__ __launch_bounds__(256) void dropout_flat4_kernel(const float4* __restrict__ x, float4* __restrict__ y, unsigned total4, float p_drop, SeedArg seed_arg,
                                                             unsigned rng_stream) {
    const unsigned long long seed = seed_arg.dev ? *seed_arg.dev : seed_arg.value;
    const float ks = 1.f / (1.f - p_drop);
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total4; i += gridDim.x * 256u) {
        const float4 v = x[i];
        unsigned r[4];
        philox4x32_10(i, 0u, rng_stream, 0u, (unsigned)seed, (unsigned)(seed >> 32), r);
        float4 o;
        o.x = ((float)(r[0] >> 8) * 5.9604644775390625e-08f >= p_drop) ? v.x * ks : 0.f;
        o.y = ((float)(r[1] >> 8) * 5.9604644775390625e-08f >= p_drop) ? v.y * ks : 0.f;
        o.z = ((float)(r[2] >> 8) * 5.9604644775390625e-08f >= p_drop) ? v.z * ks : 0.f;
        o.w = ((float)(r[3] >> 8) * 5.9604644775390625e-08f >= p_drop) ? v.w * ks : 0.f;
        y[i] = o;
    }
}
}  // namespace dsrl

static int dropout_common(const float* x, int ldx, float* y, int ldy, int64_t P, int C, float p, uint64_t seed, uint32_t rng_stream, dsrl_stream_t stream) {
    DSRL_REQUIRE(x && y && P > 0 && C > 0 && p >= 0.f && p < 1.f, DSRL_E_BADARG, "dropout: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    const long long total = (long long)P * C;
    if (ldx == C && ldy == C && total % 4 == 0 && total / 4 < (1ll << 32) && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && env_int_bn("DSRL_DROPOUT_FLAT", 1)) {
        hipLaunchKernelGGL(dropout_flat4_kernel, dim3((unsigned)std::min<long long>(ceil_div(total / 4, 256), 8192)), dim3(256), 0, st, reinterpret_cast<const float4*>(x),
                           reinterpret_cast<float4*>(y), (unsigned)(total / 4), p, seed_arg(seed), (unsigned)rng_stream);
        return launch_status("dropout_flat4_kernel");
    }
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 8192)), dim3(256), 0, st, x, ldx, y, ldy, (long long)P, C, p,
                       seed_arg(seed), (unsigned)rng_stream);
    return launch_status("dropout_kernel");
}
extern "C" int dsrl_dropout_fwd(const float* x, int ldx, float* y, int ldy, int64_t P, int C, float p, uint64_t seed, uint32_t rng_stream, dsrl_stream_t stream) {
    return dropout_common(x, ldx, y, ldy, P, C, p, seed, rng_stream, stream);
}
// the same mask applied to the gradient (the mask is regenerated, never stored)
extern "C" int dsrl_dropout_bwd(const float* dy, int lddy, float* dx, int lddx, int64_t P, int C, float p, uint64_t seed, uint32_t rng_stream, dsrl_stream_t stream) {
    return dropout_common(dy, lddy, dx, lddx, P, C, p, seed, rng_stream, stream);
}
