// Implicit-GEMM conv2d for gfx950 on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s dense peak).
//
//   forward / dgrad:  Y[m][k] = sum_{tap,c} X[pix(m,tap)][c] * Wt[k][tap][c]      (M = output pixels, N = out ch.)
//   wgrad:            dW[k][tap][c] = sum_p dY[p][k] * X[pix(p,tap)][c]           (M = out ch., N = in ch., K = pixels)
//
// Tiling: 256 threads = 4 waves, each wave owns MR x NR MFMA tiles of 32x32; the K loop runs over
// (filter tap, 32-channel chunk) pairs.  Global -> register -> LDS staging with the next chunk's loads in
// flight during the MFMAs of the current one; LDS rows are padded to 36 floats so the ds_read_b128 fragment
// reads are bank-conflict free.  Filter taps that fall completely into the zero padding for a whole tile are
// skipped (dilated ASPP convs, ASPP.py:11-13), so the work done equals the in-bounds MAC count the roofline uses.
#include "common.h"
#include <algorithm>
#include <mutex>
#include <vector>

namespace dsrl {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct ConvArgs {
    const float* x; const float* w; const float* bias; float* y;
    int ldx, ldy;
    int N, H, W, C;          // input tensor of this pass (for dgrad: dy's N,Ho,Wo,K)
    int K;                   // output channels of this pass
    int R, S, Ho, Wo;        // Ho,Wo: output spatial size of this pass
    int stride, pad, dil;
    int M;                   // N*Ho*Wo
    int cchunks;             // ceil(C/32)
    int splits;
    long long slab;          // floats per split slab (M*K) when splits > 1
};

constexpr int BK = 32;
constexpr int LDS_LD = 36;

template <int MR, int NR, int WGM, int WGN, bool DGRAD>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(const ConvArgs a) {
    constexpr int BM = 32 * MR * WGM, BN = 32 * NR * WGN;
    constexpr int A_IT = BM / 32, B_IT = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + BM * LDS_LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, z = blockIdx.z;
    const int HoWo = a.Ho * a.Wo;

    // ---- rows this thread stages: row = r0 + 32*i, 16-byte column c4
    const int c4 = tid & 7, r0 = tid >> 3;
    int a_n[A_IT], a_h[A_IT], a_w[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + r0 + 32 * i;
        a_ok[i] = m < a.M;
        const int mm = a_ok[i] ? m : 0;
        const int n = mm / HoWo, rem = mm - n * HoWo;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        a_n[i] = n;
        if (DGRAD) { a_h[i] = ho + a.pad; a_w[i] = wo + a.pad; }
        else { a_h[i] = ho * a.stride - a.pad; a_w[i] = wo * a.stride - a.pad; }
    }
    const int RS = a.R * a.S;
    const float* b_ptr[B_IT];
    bool b_ok[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int k = n0 + r0 + 32 * i;
        b_ok[i] = k < a.K;
        b_ptr[i] = a.w + (long long)(b_ok[i] ? k : 0) * RS * a.C;
    }

    // ---- taps that touch at least one in-bounds input pixel for this tile (block-uniform)
    unsigned long long tapmask = 0ull;
    {
        const int mf = m0, ml = min(m0 + BM, a.M) - 1;
        const int nf = mf / HoWo, nl = ml / HoWo;
        int hf = 0, hl = a.Ho - 1, wf = 0, wl = a.Wo - 1;
        if (nf == nl) {
            hf = (mf - nf * HoWo) / a.Wo; hl = (ml - nl * HoWo) / a.Wo;
            if (hf == hl) { wf = (mf - nf * HoWo) - hf * a.Wo; wl = (ml - nl * HoWo) - hl * a.Wo; }
        }
        for (int r = 0; r < a.R; ++r)
            for (int s = 0; s < a.S; ++s) {
                bool act;
                if (DGRAD) {
                    act = (hl + a.pad - r * a.dil >= 0) && (hf + a.pad - r * a.dil <= (a.H - 1) * a.stride) &&
                          (wl + a.pad - s * a.dil >= 0) && (wf + a.pad - s * a.dil <= (a.W - 1) * a.stride);
                } else {
                    act = (hl * a.stride - a.pad + r * a.dil >= 0) && (hf * a.stride - a.pad + r * a.dil <= a.H - 1) &&
                          (wl * a.stride - a.pad + s * a.dil >= 0) && (wf * a.stride - a.pad + s * a.dil <= a.W - 1);
                }
                if (act) tapmask |= 1ull << (r * a.S + s);
            }
    }
    const int ntaps = __builtin_popcountll(tapmask);
    const int nq = ntaps * a.cchunks;
    const int q0 = (int)((long long)nq * z / a.splits), q1 = (int)((long long)nq * (z + 1) / a.splits);

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- iterator over (tap, channel chunk)
    int cc = 0, tap = 0;
    unsigned long long rem_mask = tapmask;
    if (q0 < q1) {
        int skip = q0 / a.cchunks;
        cc = q0 - skip * a.cchunks;
        while (skip--) rem_mask &= rem_mask - 1;
        tap = __builtin_ctzll(rem_mask);
    }
    long long a_off[A_IT];     // pixel*ldx of the input row for the current tap, -1 if padding
    auto set_tap = [&](int t) {
        const int r = t / a.S, s = t - r * a.S;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int hi, wi; bool ok = a_ok[i];
            if (DGRAD) {
                const int hn = a_h[i] - r * a.dil, wn_ = a_w[i] - s * a.dil;
                hi = hn / a.stride; wi = wn_ / a.stride;
                ok = ok && hn >= 0 && wn_ >= 0 && hi * a.stride == hn && wi * a.stride == wn_ && hi < a.H && wi < a.W;
            } else {
                hi = a_h[i] + r * a.dil; wi = a_w[i] + s * a.dil;
                ok = ok && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
            }
            a_off[i] = ok ? ((long long)(a_n[i] * a.H + hi) * a.W + wi) * a.ldx : -1;
        }
    };
    float4 ra[A_IT], rb[B_IT];
    auto gload = [&](int t, int ch) {
        const int c = ch * BK + c4 * 4;
        const bool cok = c < a.C;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            ra[i] = (cok && a_off[i] >= 0) ? *reinterpret_cast<const float4*>(a.x + a_off[i] + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            rb[i] = (cok && b_ok[i]) ? *reinterpret_cast<const float4*>(b_ptr[i] + (long long)t * a.C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    if (q0 < q1) { set_tap(tap); gload(tap, cc); }
    const int frag_row = lane & 31, frag_k = (lane >> 5) * 4;
    for (int q = q0; q < q1; ++q) {
        // registers -> LDS
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *reinterpret_cast<float4*>(&As[(r0 + 32 * i) * LDS_LD + c4 * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) *reinterpret_cast<float4*>(&Bs[(r0 + 32 * i) * LDS_LD + c4 * 4]) = rb[i];
        __syncthreads();
        // next chunk's global loads fly during the MFMAs
        if (q + 1 < q1) {
            if (++cc == a.cchunks) { cc = 0; rem_mask &= rem_mask - 1; tap = __builtin_ctzll(rem_mask); set_tap(tap); }
            gload(tap, cc);
        }
#pragma unroll
        for (int ks = 0; ks < BK / 8; ++ks) {
            float4 fa[MR], fb[NR];
#pragma unroll
            for (int i = 0; i < MR; ++i)
                fa[i] = *reinterpret_cast<const float4*>(&As[((wm * MR + i) * 32 + frag_row) * LDS_LD + ks * 8 + frag_k]);
#pragma unroll
            for (int j = 0; j < NR; ++j)
                fb[j] = *reinterpret_cast<const float4*>(&Bs[((wn * NR + j) * 32 + frag_row) * LDS_LD + ks * 8 + frag_k]);
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    // ---- epilogue: D[row][col], col = lane&31 (out channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float* yout = a.y + (a.splits > 1 ? (long long)z * a.slab : 0ll);
    const int col = lane & 31, rq = (lane >> 5) * 4;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int k = n0 + (wn * NR + j) * 32 + col;
        if (k >= a.K) continue;
        const float bv = (a.bias != nullptr) ? a.bias[k] : 0.f;
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            const int mb = m0 + (wm * MR + i) * 32 + rq;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mb + (e & 3) + 8 * (e >> 2);
                if (m < a.M) yout[(long long)m * a.ldy + k] = acc[i][j][e] + bv;
            }
        }
    }
}

// y[m*ldy + k] = sum_z slab[z][m][k] + bias[k]
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, int splits, long long slab, int M, int K,
                                     const float* __restrict__ bias, float* __restrict__ y, int ldy) {
    const long long total = (long long)M * K;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int zz = 0; zz < splits; ++zz) s += slabs[zz * slab + e];
        const int m = (int)(e / K), k = (int)(e - (long long)m * K);
        y[(long long)m * ldy + k] = s + (bias ? bias[k] : 0.f);
    }
}

// wt[c][tap][k] = w[k][tap][c]   (dgrad runs the forward kernel on the transposed filter)
__global__ void weight_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int K, int Kp, int RS, int C) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int k0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 256 threads: 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, c = c0 + tx;
        tile[r][tx] = (k < K && c < C) ? w[((long long)k * RS + tap) * C + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, k = k0 + tx;
        if (c < C && k < Kp) wt[((long long)c * RS + tap) * Kp + k] = tile[tx][r];     // k in [K,Kp): zero padding
    }
}

// ------------------------------------------------------------------------------------------------ wgrad
struct WgradArgs {
    const float* x; const float* dy; float* dw;     // dw or slabs
    int ldx, lddy;
    int N, H, W, C, K, R, S, Ho, Wo, stride, pad, dil;
    long long P;                // N*Ho*Wo
    int ctiles;                 // tiles along C
    int psplits;
    long long slab;             // floats per split slab (K*RS*C) when psplits > 1
    int taps[64]; int ntaps;    // active filter taps (whole-tensor)
};

template <int MR, int NR, int WGM, int WGN>
__global__ __launch_bounds__(256) void conv_wgrad_f32_kernel(const WgradArgs a) {
    constexpr int BM = 32 * MR * WGM, BN = 32 * NR * WGN, BP = 32;
    constexpr int A_V = BM / 4, B_V = BN / 4;                 // float4 per row
    constexpr int A_RP = 256 / A_V, B_RP = 256 / B_V;          // rows per pass
    constexpr int A_IT = BP / A_RP, B_IT = BP / B_RP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // [BP][BM]
    float* Bs = smem + BP * BM;        // [BP][BN]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int kt = blockIdx.x / a.ctiles, ct = blockIdx.x - kt * a.ctiles;
    const int k0 = kt * BM, c0 = ct * BN;
    const int tap = a.taps[blockIdx.y];
    const int r = tap / a.S, s = tap - r * a.S;
    const int dh = r * a.dil - a.pad, dw_ = s * a.dil - a.pad;
    const long long nchunks = (a.P + BP - 1) / BP;
    const long long ch0 = nchunks * blockIdx.z / a.psplits, ch1 = nchunks * (blockIdx.z + 1) / a.psplits;
    const int HoWo = a.Ho * a.Wo;

    const int a_col = (tid % A_V) * 4, a_row = tid / A_V;
    const int b_col = (tid % B_V) * 4, b_row = tid / B_V;
    const bool a_cok = k0 + a_col < a.K, b_cok = c0 + b_col < a.C;

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float4 ra[A_IT], rb[B_IT];
    auto gload = [&](long long ch) {
        const long long pb = ch * BP;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const long long p = pb + a_row + i * A_RP;
            ra[i] = (a_cok && p < a.P) ? *reinterpret_cast<const float4*>(a.dy + p * a.lddy + k0 + a_col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const long long p = pb + b_row + i * B_RP;
            bool ok = b_cok && p < a.P;
            long long off = 0;
            if (ok) {
                const int pi = (int)p;
                const int n = pi / HoWo, rem = pi - n * HoWo;
                const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
                const int hi = ho * a.stride + dh, wi = wo * a.stride + dw_;
                ok = hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
                off = ((long long)(n * a.H + hi) * a.W + wi) * a.ldx + c0 + b_col;
            }
            rb[i] = ok ? *reinterpret_cast<const float4*>(a.x + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (ch0 < ch1) gload(ch0);
    const int fi = lane & 31, fh = lane >> 5;
    for (long long ch = ch0; ch < ch1; ++ch) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *reinterpret_cast<float4*>(&As[(a_row + i * A_RP) * BM + a_col]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) *reinterpret_cast<float4*>(&Bs[(b_row + i * B_RP) * BN + b_col]) = rb[i];
        __syncthreads();
        if (ch + 1 < ch1) gload(ch + 1);
#pragma unroll
        for (int st = 0; st < BP / 2; ++st) {
            float fa[MR], fb[NR];
#pragma unroll
            for (int i = 0; i < MR; ++i) fa[i] = As[(2 * st + fh) * BM + (wm * MR + i) * 32 + fi];
#pragma unroll
            for (int j = 0; j < NR; ++j) fb[j] = Bs[(2 * st + fh) * BN + (wn * NR + j) * 32 + fi];
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    float* out = a.dw + (a.psplits > 1 ? (long long)blockIdx.z * a.slab : 0ll);
    const int RS = a.R * a.S;
    const int col = lane & 31, rq = (lane >> 5) * 4;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int c = c0 + (wn * NR + j) * 32 + col;
        if (c >= a.C) continue;
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            const int kb = k0 + (wm * MR + i) * 32 + rq;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = kb + (e & 3) + 8 * (e >> 2);
                if (k < a.K) out[((long long)k * RS + tap) * a.C + c] = acc[i][j][e];
            }
        }
    }
}

struct TapList { int taps[64]; int n; };
__global__ void wgrad_reduce_kernel(const float* __restrict__ slabs, int psplits, long long slab, float* __restrict__ dw,
                                    int K, int RS, int C, TapList tl) {
    const long long total = (long long)K * tl.n * C;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const long long t = e / C;
        const int ti = (int)(t % tl.n), k = (int)(t / tl.n);
        const long long idx = ((long long)k * RS + tl.taps[ti]) * C + c;
        float s = 0.f;
        for (int zz = 0; zz < psplits; ++zz) s += slabs[zz * slab + idx];
        dw[idx] = s;
    }
}

// ------------------------------------------------------------------------------------------------ host side
static int out_size(int n, int k, int stride, int pad, int dil) { return (n + 2 * pad - dil * (k - 1) - 1) / stride + 1; }

static int valid_count(int n_in, int n_out, int stride, int pad, int off) {
    int c = 0;
    for (int o = 0; o < n_out; ++o) { const int i = o * stride - pad + off; c += (i >= 0 && i < n_in); }
    return c;
}

// ---- launch timing (opt-in)
struct ProfRec { hipEvent_t a, b; int family; double flops; };
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_event_pool;      // events are recycled: recording costs ~1 us, creating them much more
static hipEvent_t prof_event() {
    std::lock_guard<std::mutex> g(g_prof_mu);
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e; hipEventCreate(&e); return e;
}
struct ProfScope {
    bool on; ProfRec r; hipStream_t s;
    ProfScope(int family, double flops, hipStream_t st) : on(g_prof_on), s(st) {
        if (!on) return;
        r.family = family; r.flops = flops;
        r.a = prof_event(); r.b = prof_event();
        hipEventRecord(r.a, s);
    }
    ~ProfScope() {
        if (!on) return;
        hipEventRecord(r.b, s);
        std::lock_guard<std::mutex> g(g_prof_mu);
        g_prof.push_back(r);
    }
};

enum TileCfg { T128x128, T256x64, T256x32 };
static TileCfg pick_cfg(int K) {
    if (K <= 32) return T256x32;
    const int r = K % 128;
    if (K <= 64 || (r > 0 && r <= 64)) return T256x64;
    return T128x128;
}
static void cfg_dims(TileCfg c, int& bm, int& bn) {
    switch (c) { case T128x128: bm = 128; bn = 128; break; case T256x64: bm = 256; bn = 64; break; default: bm = 256; bn = 32; }
}
static int pick_splits(long long tiles, int nq) {
    int s = 1;
    if (tiles < 2 * kNumCU) s = (int)ceil_div(2 * kNumCU, tiles);
    s = (int)std::min<long long>(s, std::max(1, nq / 8));
    return std::max(1, std::min(s, 32));
}

template <bool DGRAD>
static int launch_igemm(const ConvArgs& a, TileCfg cfg, hipStream_t st) {
    int bm, bn; cfg_dims(cfg, bm, bn);
    dim3 grid((unsigned)ceil_div(a.M, bm), (unsigned)ceil_div(a.K, bn), (unsigned)a.splits);
    const size_t lds = (size_t)(bm + bn) * LDS_LD * sizeof(float);
    switch (cfg) {
        case T128x128: hipLaunchKernelGGL((conv_igemm_f32_kernel<2, 2, 2, 2, DGRAD>), grid, dim3(256), lds, st, a); break;
        case T256x64:  hipLaunchKernelGGL((conv_igemm_f32_kernel<2, 2, 4, 1, DGRAD>), grid, dim3(256), lds, st, a); break;
        default:       hipLaunchKernelGGL((conv_igemm_f32_kernel<2, 1, 4, 1, DGRAD>), grid, dim3(256), lds, st, a); break;
    }
    return launch_status("conv_igemm_f32_kernel");
}

static int check_conv(const void* p0, const void* p1, const void* p2, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil) {
    DSRL_REQUIRE(p0 && p1 && p2, DSRL_E_BADARG, "conv2d: null pointer");
    DSRL_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && R > 0 && S > 0 && stride > 0 && dil > 0 && pad >= 0, DSRL_E_BADARG,
                 "conv2d: bad shape N%d H%d W%d C%d K%d R%d S%d stride%d pad%d dil%d", N, H, W, C, K, R, S, stride, pad, dil);
    DSRL_REQUIRE(R * S <= 64, DSRL_E_UNSUPPORTED, "conv2d: filter %dx%d has more than 64 taps", R, S);
    DSRL_REQUIRE(out_size(H, R, stride, pad, dil) > 0 && out_size(W, S, stride, pad, dil) > 0, DSRL_E_BADARG, "conv2d: empty output");
    return 0;
}

struct FwdPlan { int Ho, Wo, M, cchunks, splits; TileCfg cfg; size_t ws; };
static FwdPlan plan_fwd(int N, int Hin, int Win, int Cin, int Kout, int R, int S, int Ho, int Wo) {
    FwdPlan p; p.Ho = Ho; p.Wo = Wo; p.M = N * Ho * Wo; p.cchunks = (int)ceil_div(Cin, BK);
    p.cfg = pick_cfg(Kout);
    int bm, bn; cfg_dims(p.cfg, bm, bn);
    p.splits = pick_splits(ceil_div(p.M, bm) * ceil_div(Kout, bn), R * S * p.cchunks);
    p.ws = p.splits > 1 ? (size_t)p.splits * p.M * Kout * sizeof(float) : 0;
    return p;
}

}  // namespace dsrl

using namespace dsrl;

extern "C" int64_t dsrl_conv2d_inbounds_macs(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil) {
    const int Ho = out_size(H, R, stride, pad, dil), Wo = out_size(W, S, stride, pad, dil);
    if (Ho <= 0 || Wo <= 0) return 0;
    int64_t pix = 0;
    for (int r = 0; r < R; ++r)
        for (int s = 0; s < S; ++s) pix += (int64_t)valid_count(H, Ho, stride, pad, r * dil) * valid_count(W, Wo, stride, pad, s * dil);
    return pix * N * C * K;
}

extern "C" size_t dsrl_conv2d_fwd_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil) {
    const int Ho = out_size(H, R, stride, pad, dil), Wo = out_size(W, S, stride, pad, dil);
    if (Ho <= 0 || Wo <= 0) return 0;
    return plan_fwd(N, H, W, C, K, R, S, Ho, Wo).ws;
}

extern "C" int dsrl_conv2d_fwd(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                               int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                               void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    if (int e = check_conv(x, w, y, N, H, W, C, K, R, S, stride, pad, dil)) return e;
    DSRL_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0, DSRL_E_UNSUPPORTED,
                 "conv2d_fwd: C (%d) and ldx (%d) must be multiples of 4 and x,w 16-byte aligned", C, ldx);
    DSRL_REQUIRE(ldx >= C && ldy >= K, DSRL_E_BADARG, "conv2d_fwd: ld smaller than channel count");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    const int Ho = out_size(H, R, stride, pad, dil), Wo = out_size(W, S, stride, pad, dil);
    const FwdPlan p = plan_fwd(N, H, W, C, K, R, S, Ho, Wo);
    DSRL_REQUIRE(ws_bytes >= p.ws && (p.ws == 0 || ws), DSRL_E_WORKSPACE, "conv2d_fwd: workspace %zu < %zu", ws_bytes, p.ws);
    ConvArgs a{};
    a.x = x; a.w = w; a.ldx = ldx; a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.Ho = Ho; a.Wo = Wo;
    a.stride = stride; a.pad = pad; a.dil = dil; a.M = p.M; a.cchunks = p.cchunks; a.splits = p.splits; a.slab = (long long)p.M * K;
    ProfScope prof(0, 2.0 * (double)dsrl_conv2d_inbounds_macs(N, H, W, C, K, R, S, stride, pad, dil), st);
    if (p.splits > 1) {
        a.y = (float*)ws; a.ldy = K; a.bias = nullptr;
        if (int e = launch_igemm<false>(a, p.cfg, st)) return e;
        const long long total = (long long)p.M * K;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 4096)), dim3(256), 0, st,
                           (const float*)ws, p.splits, a.slab, p.M, K, bias, y, ldy);
        return launch_status("splitk_reduce_kernel");
    }
    a.y = y; a.ldy = ldy; a.bias = bias;
    return launch_igemm<false>(a, p.cfg, st);
}

// dgrad = the same implicit GEMM with dy as the input tensor, the transposed filter wt[c][tap][k] and the
// gather pixel(h,w,tap) = ((h + pad - r*dil)/stride, (w + pad - s*dil)/stride) when divisible.
static int pad4(int v) { return (v + 3) & ~3; }
static size_t dgrad_wt_bytes(int C, int K, int R, int S) { return align_up((size_t)C * pad4(K) * R * S * sizeof(float), 256); }

extern "C" size_t dsrl_conv2d_dgrad_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil) {
    const int Ho = out_size(H, R, stride, pad, dil), Wo = out_size(W, S, stride, pad, dil);
    if (Ho <= 0 || Wo <= 0) return 0;
    return dgrad_wt_bytes(C, K, R, S) + plan_fwd(N, Ho, Wo, pad4(K), C, R, S, H, W).ws;
}

extern "C" int dsrl_conv2d_dgrad(const float* dy, int lddy, const float* w, float* dx, int lddx,
                                 int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                 void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    if (int e = check_conv(dy, w, dx, N, H, W, C, K, R, S, stride, pad, dil)) return e;
    const int Kp = pad4(K);     // K % 4 != 0 (cls_conv, 19 classes): dy must be padded to lddy >= Kp with finite pad values
    DSRL_REQUIRE(lddy % 4 == 0 && ((uintptr_t)dy % 16) == 0, DSRL_E_UNSUPPORTED,
                 "conv2d_dgrad: lddy (%d) must be a multiple of 4 and dy 16-byte aligned", lddy);
    DSRL_REQUIRE(lddy >= Kp && lddx >= C, DSRL_E_BADARG, "conv2d_dgrad: ld smaller than (padded) channel count");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    const int Ho = out_size(H, R, stride, pad, dil), Wo = out_size(W, S, stride, pad, dil);
    const FwdPlan p = plan_fwd(N, Ho, Wo, Kp, C, R, S, H, W);
    const size_t wtb = dgrad_wt_bytes(C, K, R, S);
    DSRL_REQUIRE(ws && ws_bytes >= wtb + p.ws, DSRL_E_WORKSPACE, "conv2d_dgrad: workspace %zu < %zu", ws_bytes, wtb + p.ws);
    float* wt = (float*)ws;
    float* slabs = (float*)((char*)ws + wtb);
    hipLaunchKernelGGL(weight_transpose_kernel, dim3((unsigned)ceil_div(C, 32), (unsigned)ceil_div(Kp, 32), (unsigned)(R * S)), dim3(256), 0, st,
                       w, wt, K, Kp, R * S, C);
    if (int e = launch_status("weight_transpose_kernel")) return e;
    ConvArgs a{};
    a.x = dy; a.w = wt; a.ldx = lddy; a.N = N; a.H = Ho; a.W = Wo; a.C = Kp; a.K = C; a.R = R; a.S = S; a.Ho = H; a.Wo = W;
    a.stride = stride; a.pad = pad; a.dil = dil; a.M = p.M; a.cchunks = p.cchunks; a.splits = p.splits; a.slab = (long long)p.M * C;
    ProfScope prof(0, 2.0 * (double)dsrl_conv2d_inbounds_macs(N, H, W, C, K, R, S, stride, pad, dil), st);
    if (p.splits > 1) {
        a.y = slabs; a.ldy = C; a.bias = nullptr;
        if (int e = launch_igemm<true>(a, p.cfg, st)) return e;
        const long long total = (long long)p.M * C;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 4096)), dim3(256), 0, st,
                           (const float*)slabs, p.splits, a.slab, p.M, C, (const float*)nullptr, dx, lddx);
        return launch_status("splitk_reduce_kernel");
    }
    a.y = dx; a.ldy = lddx; a.bias = nullptr;
    return launch_igemm<true>(a, p.cfg, st);
}

namespace dsrl {
struct WgPlan { int Ho, Wo; long long P; TileCfg cfg; int bm, bn, ktiles, ctiles, psplits; TapList tl; size_t ws; };
static WgPlan plan_wgrad(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil) {
    WgPlan p; p.Ho = out_size(H, R, stride, pad, dil); p.Wo = out_size(W, S, stride, pad, dil);
    p.P = (long long)N * p.Ho * p.Wo;
    // GEMM rows = out channels, columns = in channels
    p.cfg = pick_cfg(C);
    int bm, bn; cfg_dims(p.cfg, bm, bn); p.bm = bm; p.bn = bn;
    p.ktiles = (int)ceil_div(K, bm); p.ctiles = (int)ceil_div(C, bn);
    p.tl.n = 0;
    for (int r = 0; r < R; ++r)
        for (int s = 0; s < S; ++s)
            if (valid_count(H, p.Ho, stride, pad, r * dil) > 0 && valid_count(W, p.Wo, stride, pad, s * dil) > 0) p.tl.taps[p.tl.n++] = r * S + s;
    const long long tiles = (long long)p.ktiles * p.ctiles * std::max(1, p.tl.n);
    const long long chunks = ceil_div(p.P, 32);
    long long sp = tiles < 2 * kNumCU ? ceil_div(2 * kNumCU, tiles) : 1;
    sp = std::min(sp, std::max<long long>(1, chunks / 4));
    p.psplits = (int)std::max<long long>(1, std::min<long long>(sp, 64));
    p.ws = p.psplits > 1 ? (size_t)p.psplits * K * R * S * C * sizeof(float) : 0;
    return p;
}
}  // namespace dsrl

extern "C" size_t dsrl_conv2d_wgrad_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil) {
    if (out_size(H, R, stride, pad, dil) <= 0 || out_size(W, S, stride, pad, dil) <= 0) return 0;
    return plan_wgrad(N, H, W, C, K, R, S, stride, pad, dil).ws;
}

extern "C" int dsrl_conv2d_wgrad(const float* x, int ldx, const float* dy, int lddy, float* dw,
                                 int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                 void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    if (int e = check_conv(x, dy, dw, N, H, W, C, K, R, S, stride, pad, dil)) return e;
    DSRL_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && lddy >= pad4(K) && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0,
                 DSRL_E_UNSUPPORTED, "conv2d_wgrad: C (%d), ldx (%d), lddy (%d) must be multiples of 4 (lddy >= K rounded up to 4), pointers 16-byte aligned", C, ldx, lddy);
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    const WgPlan p = plan_wgrad(N, H, W, C, K, R, S, stride, pad, dil);
    DSRL_REQUIRE(p.P < (1ll << 31), DSRL_E_UNSUPPORTED, "conv2d_wgrad: more than 2^31 output pixels");
    DSRL_REQUIRE(ws_bytes >= p.ws && (p.ws == 0 || ws), DSRL_E_WORKSPACE, "conv2d_wgrad: workspace %zu < %zu", ws_bytes, p.ws);
    const int RS = R * S;
    if (p.tl.n < RS) {      // taps that only ever see zero padding have a zero gradient
        if (hipMemsetAsync(dw, 0, (size_t)K * RS * C * sizeof(float), st) != hipSuccess) return launch_status("hipMemsetAsync(dw)");
    }
    if (p.tl.n == 0) return DSRL_OK;
    WgradArgs a{};
    a.x = x; a.dy = dy; a.ldx = ldx; a.lddy = lddy; a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.Ho = p.Ho; a.Wo = p.Wo;
    a.stride = stride; a.pad = pad; a.dil = dil; a.P = p.P; a.ctiles = p.ctiles; a.psplits = p.psplits; a.slab = (long long)K * RS * C;
    a.ntaps = p.tl.n;
    for (int i = 0; i < p.tl.n; ++i) a.taps[i] = p.tl.taps[i];
    a.dw = p.psplits > 1 ? (float*)ws : dw;
    dim3 grid((unsigned)(p.ktiles * p.ctiles), (unsigned)p.tl.n, (unsigned)p.psplits);
    const size_t lds = (size_t)32 * (p.bm + p.bn) * sizeof(float);
    ProfScope prof(1, 2.0 * (double)dsrl_conv2d_inbounds_macs(N, H, W, C, K, R, S, stride, pad, dil), st);
    switch (p.cfg) {
        case T128x128: hipLaunchKernelGGL((conv_wgrad_f32_kernel<2, 2, 2, 2>), grid, dim3(256), lds, st, a); break;
        case T256x64:  hipLaunchKernelGGL((conv_wgrad_f32_kernel<2, 2, 4, 1>), grid, dim3(256), lds, st, a); break;
        default:       hipLaunchKernelGGL((conv_wgrad_f32_kernel<2, 1, 4, 1>), grid, dim3(256), lds, st, a); break;
    }
    if (int e = launch_status("conv_wgrad_f32_kernel")) return e;
    if (p.psplits > 1) {
        const long long total = (long long)K * p.tl.n * C;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 4096)), dim3(256), 0, st,
                           (const float*)ws, p.psplits, a.slab, dw, K, RS, C, p.tl);
        return launch_status("wgrad_reduce_kernel");
    }
    return DSRL_OK;
}

extern "C" int dsrl_prof_enable(int on) {
    std::lock_guard<std::mutex> g(g_prof_mu);
    g_prof_on = on != 0;
    if (on) {
        for (auto& r : g_prof) { g_event_pool.push_back(r.a); g_event_pool.push_back(r.b); }
        g_prof.clear();
    }
    return DSRL_OK;
}

extern "C" int dsrl_prof_read(int family, int64_t* launches, double* total_ms, double* total_flops) {
    std::lock_guard<std::mutex> g(g_prof_mu);
    int64_t n = 0; double ms = 0, fl = 0;
    for (auto& r : g_prof) {
        if (r.family != family) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) { set_error("prof_read: hipEventSynchronize failed"); return DSRL_E_LAUNCH; }
        float t = 0.f;
        hipEventElapsedTime(&t, r.a, r.b);
        ms += t; fl += r.flops; ++n;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    return DSRL_OK;
}

extern "C" const char* dsrl_prof_kernel_name(int family) {
    return family == 0 ? "conv_igemm_f32_kernel" : (family == 1 ? "conv_wgrad_f32_kernel" : "");
}
