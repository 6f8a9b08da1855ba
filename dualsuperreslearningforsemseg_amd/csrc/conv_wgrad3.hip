// Weight gradient of a 3x3, stride-1 convolution with ALL NINE TAPS in one block (f16x3 / f16x1 arithmetic).
//
// conv_wgrad_split_kernel (conv_igemm.hip) gives every filter tap its own blocks: a 32-pixel chunk of dy is fetched, split into fp16 terms and
// written to LDS by 9 x (tiles along C) blocks, a chunk of x by 9 x (tiles along K) blocks - 13.7 vector instructions per MFMA and 3.5 x the
// algorithmic HBM traffic (profiles/round3_*), with the matrix pipe a third busy.  Here a block owns a (64 out-channel x 64 in-channel) tile of
// dw for all nine taps: per chunk of 32 consecutive output pixels of ONE image row it stages the dy rows once and the three input rows
// ho-d, ho, ho+d once, each 32 + 2d pixels wide (d = dilation = padding), and the tap (r, s) reads its x operand from row r at a pixel offset of
// s*d - the nine shifted windows are nine LDS addresses, not nine trips through the memory pipeline.
// 8 waves = 2 x 2 positions of a 32 x 32 piece x 2 tap groups (taps 0..4 and 5..8; waves w and w + 4 share a SIMD, so every SIMD carries nine taps):
// 80 accumulator registers per lane, 230 VGPRs, two waves per SIMD.  Per 32-pixel chunk a wave issues 2 x 5 (4) x 3 MFMAs (32x32x16 f16) on
// independent accumulators against 0.16 KB of staged operands per MFMA (0.5 KB in the per-tap kernel).  What did not survive register allocation:
// 4 waves with all nine taps each (144 accumulators in AGPRs: one wave per SIMD is issue-bound; 288: shuffled between AGPRs and VGPRs) and 64 x 32
// wave pieces (160 accumulators + staging: 350 spills) - profiles/round4_wgrad3.txt.
//
// Staging mirrors wgrad_split_body: pixel-major fp16 planes in LDS (row stride + 64 B), operands gathered with ds_read_b64_tr_b16, two LDS
// stages of one whole chunk each, ONE block-wide barrier per chunk; the split of chunk i+1 is interleaved with the MFMAs of chunk i and every
// staging register is re-requested (chunk i+2) as soon as its value has been split, so each global load has a whole chunk phase to arrive.
// Results: the same terms in the same order per tap as the per-tap kernel (a0*b1, a1*b0, a0*b0 per 16 pixels, chunks ascending inside a pixel
// range); with the same pixel ranges the two kernels agree bit for bit (tests/test_hip_parity.py).
#include "conv_common.h"
#include <type_traits>

namespace dsrl {

namespace {
constexpr int kW3WGM = 2, kW3WGN = 2, kW3TG = 2;            // waves along K / along C of the 64 x 64 tile, tap groups: 8 waves
constexpr int kW3Threads = 64 * kW3WGM * kW3WGN * kW3TG;
constexpr int kW3Taps0 = 5;                                  // tap group 0: taps 0 .. 4, group 1: taps 5 .. 8 (waves w and w + 4 share a SIMD: 9 taps per SIMD)

template <int NPL, int HALO>
struct W3Geom {
    static constexpr int BM = 32 * kW3WGM, BN = 32 * kW3WGN, NT = kW3Threads;
    static constexpr int XW = 32 + 2 * HALO, XR = 3 * XW;    // staged pixels per input row, staged x rows per chunk
    static constexpr int A_V = BM / 4, B_V = BN / 4;          // float4 per pixel row
    static constexpr int A_RP = NT / A_V, B_RP = NT / B_V;    // pixel rows per staging pass
    static constexpr int A_IT = 32 / A_RP, B_IT = (XR + B_RP - 1) / B_RP;
    static constexpr int SA = BM * 2 + 64, SB = BN * 2 + 64;  // LDS row strides in bytes
    // the x plane has room for every row the staging passes touch (B_IT * B_RP >= XR): the passes write unconditionally, rows past XR are never read
    static constexpr int PLA = 32 * SA, PLB = B_IT * B_RP * SB, OFF_B = NPL * PLA, STAGE = NPL * (PLA + PLB);
    static_assert(32 % A_RP == 0 && A_IT >= 1, "dy staging passes");
};

template <int NPL, int HALO>
__device__ __forceinline__ void wgrad3_body(const WgradArgs& a, const int bid) {
    using G = W3Geom<NPL, HALO>;
    using PT = Plane<true>;
    using pl4 = typename PT::v4; using pl8 = typename PT::v8;
    constexpr int BM = G::BM, BN = G::BN, XW = G::XW, XR = G::XR, A_IT = G::A_IT, B_IT = G::B_IT, A_RP = G::A_RP, B_RP = G::B_RP;
    constexpr int SA = G::SA, SB = G::SB, PLA = G::PLA, PLB = G::PLB, OFF_B = G::OFF_B, STAGE = G::STAGE;
    const unsigned am_a = amax_fetch(a.amax_dy), am_b = amax_fetch(a.amax_x);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const S0 = reinterpret_cast<char*>(smem);
    char* const S1 = S0 + STAGE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tgrp = wave >> 2, wpos = wave & 3;                // tap group; position of the wave's 32 x 32 piece in the tile
    const int wm = wpos / kW3WGN, wn = wpos % kW3WGN;
    const int nb = a.kctiles * a.psplits;
    const int id = a.xcd_remap ? xcd_contiguous(bid, nb) : bid;
    const int zsplit = id / a.kctiles, kc = id - zsplit * a.kctiles;
    const int kt = kc / a.ctiles, ct = kc - kt * a.ctiles;
    const int k0 = kt * BM, c0 = ct * BN;
    const int nchunks = (int)(a.P >> 5);                       // host: Wo % 32 == 0, so P % 32 == 0 and a chunk never leaves its image row
    const int ch0 = (int)((long long)nchunks * zsplit / a.psplits), ch1 = (int)((long long)nchunks * (zsplit + 1) / a.psplits);
    const int g_lddy = a.lddy, g_ldx = a.ldx, g_H = a.H, g_W = a.W, g_Wo = a.Wo, g_dil = a.dil;
    const unsigned g_mHW = a.mHW, g_sHW = a.sHW, g_mW = a.mW, g_sW = a.sW;
    const int HoWo = a.Ho * g_Wo;

    const int a_col = (tid % G::A_V) * 4, a_row = tid / G::A_V;
    const int b_col = (tid % G::B_V) * 4, b_row = tid / G::B_V;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)a.dy_bytes, 0x00020000);

    // ---- global addressing.  offset = (block-uniform part of the chunk) + (per-thread constant of the staging pass); a pass is dropped (out-of-range
    //      offset: the load returns zeros) when one of its flags meets the chunk's mask: bit 0 / 1 the row above / below the image, bit 2 / 3 the
    //      halo pixels left / right of it, bit 4 always (a chunk past the range, a thread past the staged rows or channels).
    unsigned a_rel[A_IT], b_rel[B_IT], b_flag[B_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) a_rel[i] = k0 + a_col < a.K ? (unsigned)((a_row + i * A_RP) * g_lddy + k0 + a_col) * 4u : kOOB;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int t = b_row + i * B_RP, r = t / XW, j = t - r * XW;
        const bool live = t < XR && c0 + b_col < a.C;
        b_rel[i] = (unsigned)((((r - 1) * g_dil) * g_W + (j - HALO)) * g_ldx + c0 + b_col) * 4u;      // may be "negative": added to the chunk's base modulo 2^32
        b_flag[i] = live ? ((r == 0 ? 1u : 0u) | (r == 2 ? 2u : 0u) | (j < HALO ? 4u : 0u) | (j >= 32 + HALO ? 8u : 0u)) : 16u;
    }
    float4 RA[A_IT], RB[B_IT];
    struct ChunkBase { unsigned a, b, mask; };
    auto chunk_base = [&](int ch) -> ChunkBase {        // block-uniform (scalar registers)
        const int cc = ch >= 0 ? ch : 0;
        const int p0 = cc * 32;
        const int n = fast_div(p0, g_mHW, g_sHW), rem = p0 - n * HoWo;
        const int ho = fast_div(rem, g_mW, g_sW), wo0 = rem - ho * g_Wo;
        ChunkBase b;
        b.a = (unsigned)p0 * (unsigned)g_lddy * 4u;
        b.b = (unsigned)((n * g_H + ho) * g_W + wo0) * (unsigned)g_ldx * 4u;
        b.mask = (ho < g_dil ? 1u : 0u) | (ho + g_dil >= g_H ? 2u : 0u) | (wo0 == 0 ? 4u : 0u) | (wo0 + 32 >= g_W ? 8u : 0u) | 16u | (ch >= 0 ? 0u : 32u);
        return b;
    };
    // every load is issued unconditionally (a phase is ONE basic block: the compiler's s_waitcnt then counts the loads in flight instead of draining them)
    auto load_a = [&](int i, const ChunkBase& cb) {
        RA[i] = buf_load4(dr, (cb.mask & 32u) ? kOOB : __builtin_elementwise_add_sat(cb.a, a_rel[i]));
    };
    auto load_b = [&](int i, const ChunkBase& cb) {
        RB[i] = buf_load4(xr, ((b_flag[i] | 32u) & cb.mask) ? kOOB : cb.b + b_rel[i]);
    };

    // ---- LDS addressing (transposed reads as in wgrad_split_body: 16-lane group g -> pixel half g >> 1, column half g & 1; t = 4q + p)
    const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int tr_off_a = (8 * (tg >> 1) + tq) * SA + (16 * (tg & 1) + 4 * tp) * 2 + wm * 64;
    const int tr_off_b = OFF_B + (8 * (tg >> 1) + tq) * SB + (16 * (tg & 1) + 4 * tp) * 2 + wn * 64;
    const int wa_off = a_row * SA + a_col * 2;                               // + i * A_RP * SA
    const int wb_off = OFF_B + b_row * SB + b_col * 2;                       // + i * B_RP * SB
    using lds_bf16x4 = __attribute__((address_space(3))) bf16x4;
    auto tr8 = [&](const char* q, int stride) -> pl8 {
        const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q));
        const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q + 4 * stride));
        return __builtin_bit_cast(pl8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    int sh_a = 0, sh_b = 0;
    float res[4] = {0.f, 0.f, 0.f, 0.f};
    // split step c of the chunk held in RA / RB into stage `st`: plane c % NPL of staged value c / NPL; behind the last plane of a value its register
    // is requested again for the chunk `cb` describes
    auto cstep = [&](char* st, int c, const ChunkBase& cb) {
        const int v = c / NPL, pl = c % NPL;
        if (pl == 0) {
            const float4 x = v < A_IT ? RA[v < A_IT ? v : 0] : RB[v >= A_IT ? v - A_IT : 0];
            const int sh = v < A_IT ? sh_a : sh_b;
            res[0] = __builtin_ldexpf(x.x, sh); res[1] = __builtin_ldexpf(x.y, sh); res[2] = __builtin_ldexpf(x.z, sh); res[3] = __builtin_ldexpf(x.w, sh);
        }
        const pl4 t = PT::cvt(res);
        if (v < A_IT) {
            *reinterpret_cast<pl4*>(st + pl * PLA + wa_off + v * (A_RP * SA)) = t;
        } else {
            *reinterpret_cast<pl4*>(st + pl * PLB + wb_off + (v - A_IT) * (B_RP * SB)) = t;
        }
        if (pl + 1 < NPL) {
            PT::residual(res, t);
        } else {
            if (v < A_IT) load_a(v < A_IT ? v : 0, cb); else load_b(v >= A_IT ? v - A_IT : 0, cb);
        }
    };
    constexpr int NMF = NPL * (NPL + 1) / 2;                  // MFMAs per tap and 16-pixel half
    constexpr int CSTEPS = (A_IT + B_IT) * NPL;
    f32x16 acc[kW3Taps0];
#pragma unroll
    for (int t = 0; t < kW3Taps0; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    // MFMAs of the chunk in `cur` for the taps T0 .. T0 + NT - 1, the split of the chunk in the registers into `nxt` between them
    auto phase = [&](auto t0c, auto ntc, const char* cur, char* nxt, int rch) {
        constexpr int T0 = decltype(t0c)::value, NTAP = decltype(ntc)::value;
        constexpr int NMFMA = 2 * NTAP * NMF, MPS = NMFMA / CSTEPS > 0 ? NMFMA / CSTEPS : 1;
        const ChunkBase cb = chunk_base(rch);
        int m = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            pl8 fa[NPL], fb[2][NPL];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) fa[pl] = tr8(cur + pl * PLA + tr_off_a + 16 * h * SA, SA);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) fb[0][pl] = tr8(cur + pl * PLB + tr_off_b + ((T0 / 3) * XW + 16 * h + (T0 % 3) * HALO) * SB, SB);
#pragma unroll
            for (int j = 0; j < NTAP; ++j) {
                if (j + 1 < NTAP) {
                    const int r = (T0 + j + 1) / 3, s = (T0 + j + 1) % 3;
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) fb[(j + 1) & 1][pl] = tr8(cur + pl * PLB + tr_off_b + (r * XW + 16 * h + s * HALO) * SB, SB);
                }
#pragma unroll
                for (int sum = NPL - 1; sum >= 0; --sum)
#pragma unroll
                    for (int pa = 0; pa <= sum; ++pa) {
                        acc[j] = PT::mfma(fa[pa], fb[j & 1][sum - pa], acc[j]);
                        if (m % MPS == 0 && m / MPS < CSTEPS) cstep(nxt, m / MPS, cb);
                        __builtin_amdgcn_sched_barrier(0);
                        ++m;
                    }
            }
        }
#pragma unroll
        for (int c = (NMFMA + MPS - 1) / MPS; c < CSTEPS; ++c) cstep(nxt, c, cb);
    };
    using I0 = std::integral_constant<int, 0>; using I5 = std::integral_constant<int, kW3Taps0>; using I4 = std::integral_constant<int, 9 - kW3Taps0>;

    if (ch0 < ch1) {
        {
            const ChunkBase cb = chunk_base(ch0);
#pragma unroll
            for (int i = 0; i < A_IT; ++i) load_a(i, cb);
#pragma unroll
            for (int i = 0; i < B_IT; ++i) load_b(i, cb);
        }
        __builtin_amdgcn_sched_barrier(0);
        sh_a = amax_shift_of(am_a); sh_b = amax_shift_of(am_b);
        {
            const ChunkBase cb = chunk_base(ch0 + 1 < ch1 ? ch0 + 1 : -1);
#pragma unroll
            for (int c = 0; c < CSTEPS; ++c) cstep(S0, c, cb);
        }
        __syncthreads();
        // the split of the registers into the other stage runs in every phase: behind the last chunk it moves zeros (out-of-range loads) into a stage
        // nobody reads again
        if (tgrp == 0) {
            for (int ch = ch0; ch < ch1; ch += 2) {
                phase(I0{}, I5{}, S0, S1, ch + 2 < ch1 ? ch + 2 : -1);
                __syncthreads();
                if (ch + 1 >= ch1) break;
                phase(I0{}, I5{}, S1, S0, ch + 3 < ch1 ? ch + 3 : -1);
                __syncthreads();
            }
        } else {
            for (int ch = ch0; ch < ch1; ch += 2) {
                phase(I5{}, I4{}, S0, S1, ch + 2 < ch1 ? ch + 2 : -1);
                __syncthreads();
                if (ch + 1 >= ch1) break;
                phase(I5{}, I4{}, S1, S0, ch + 3 < ch1 ? ch + 3 : -1);
                __syncthreads();
            }
        }
    } else {
        sh_a = amax_shift_of(am_a); sh_b = amax_shift_of(am_b);
    }

    float* out = a.dw + (a.psplits > 1 ? (long long)zsplit * a.slab : 0ll);
    const int col = lane & 31, rq = (lane >> 5) * 4;
    const int sh_out = -(sh_a + sh_b);
    const int c = c0 + wn * 32 + col;
    if (c < a.C) {
        const int kb = k0 + wm * 32 + rq;
        const int tfirst = tgrp == 0 ? 0 : kW3Taps0, ntap = tgrp == 0 ? kW3Taps0 : 9 - kW3Taps0;
#pragma unroll
        for (int j = 0; j < kW3Taps0; ++j)
            if (j < ntap) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = kb + (e & 3) + 8 * (e >> 2);
                    if (k < a.K) out[((long long)k * 9 + tfirst + j) * a.C + c] = __builtin_ldexpf(acc[j][e], sh_out);
                }
            }
    }
}

template <int NPL, int HALO>
__global__ __launch_bounds__(kW3Threads, 1) void conv_wgrad3_kernel(const WgradArgs a) {
    wgrad3_body<NPL, HALO>(a, (int)blockIdx.x);
}
// grouped launch (dsrl_conv2d_wgrad_group_*): see conv_wgrad_group_kernel
template <int NPL, int HALO>
__global__ __launch_bounds__(kW3Threads, 1) void conv_wgrad3_group_kernel(const WgradArgs* __restrict__ table, const int* __restrict__ starts, int nprob) {
    const int b = (int)blockIdx.x;
    int lo = 0, hi = nprob - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (starts[mid] <= b) lo = mid; else hi = mid - 1;
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    const WgradArgs& a = table[lo];
    const int local = b - starts[lo];
    if (local >= a.nblocks) return;
    wgrad3_body<NPL, HALO>(a, local);
}

template <int NPL, int HALO> size_t w3_lds() { return 2 * (size_t)W3Geom<NPL, HALO>::STAGE; }
template <typename Kern> int w3_attr(Kern k, size_t lds) {
    return hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 0 : 1;
}
}  // namespace

bool wgrad3_eligible(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil, int npl, bool f16) {
    if (!f16 || (npl != 1 && npl != 2) || !env_int("DSRL_WGRAD3", 1)) return false;
    if (R != 3 || S != 3 || stride != 1 || pad != dil || (dil != 1 && dil != 2)) return false;
    if (W % 32 != 0 || H <= dil || N <= 0) return false;                    // Ho == H, Wo == W: a 32-pixel chunk is a piece of one image row; all nine taps see pixels
    if (C % 4 != 0 || K < env_int("DSRL_WGRAD3_MIN_K", 64) || C < 32) return false;     // the 64-row tile of out channels is full at least once
    return (long long)N * H * W < (1ll << 31);
}
void wgrad3_tile(int& bm, int& bn) { bm = 32 * kW3WGM; bn = 32 * kW3WGN; }

#define DSRL_W3_DISPATCH(KERN, ...)                                                                                  \
    do {                                                                                                             \
        if (npl == 2 && dil == 1) { static const int at_ = w3_attr(KERN<2, 1>, w3_lds<2, 1>()); (void)at_; hipLaunchKernelGGL((KERN<2, 1>), dim3((unsigned)grid), dim3(kW3Threads), (w3_lds<2, 1>()), st, __VA_ARGS__); } \
        else if (npl == 2)        { static const int at_ = w3_attr(KERN<2, 2>, w3_lds<2, 2>()); (void)at_; hipLaunchKernelGGL((KERN<2, 2>), dim3((unsigned)grid), dim3(kW3Threads), (w3_lds<2, 2>()), st, __VA_ARGS__); } \
        else if (dil == 1)        { static const int at_ = w3_attr(KERN<1, 1>, w3_lds<1, 1>()); (void)at_; hipLaunchKernelGGL((KERN<1, 1>), dim3((unsigned)grid), dim3(kW3Threads), (w3_lds<1, 1>()), st, __VA_ARGS__); } \
        else                      { static const int at_ = w3_attr(KERN<1, 2>, w3_lds<1, 2>()); (void)at_; hipLaunchKernelGGL((KERN<1, 2>), dim3((unsigned)grid), dim3(kW3Threads), (w3_lds<1, 2>()), st, __VA_ARGS__); } \
    } while (0)

int launch_wgrad3(const WgradArgs& a, int npl, hipStream_t st) {
    const int grid = a.kctiles * a.psplits, dil = a.dil;
    DSRL_W3_DISPATCH(conv_wgrad3_kernel, a);
    return launch_status("conv_wgrad3_kernel");
}
int launch_wgrad3_group(const WgradArgs* table, const int* starts, int nprob, int grid, int npl, int dil, hipStream_t st) {
    DSRL_W3_DISPATCH(conv_wgrad3_group_kernel, table, starts, nprob);
    return launch_status("conv_wgrad3_group_kernel");
}

}  // namespace dsrl
