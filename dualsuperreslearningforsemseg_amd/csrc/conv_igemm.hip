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
#include "conv_common.h"
#include <algorithm>
#include <atomic>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>

namespace dsrl {

static std::atomic<int> g_conv_precision{-1};     // dsrl_conv_precision(); -1 = DSRL_CONV_PRECISION, modes in conv_precision_mode()
constexpr int BK = 32;
constexpr int LDS_LD = 36;

// 4 waves per SIMD (<= 128 registers) for the tiles that stage at most 8 rows per thread: 4 blocks of 36.9 KB LDS per CU
// DBUF: two LDS stages - the next chunk is written while the current one feeds the MFMAs, one barrier per chunk (used when
// few blocks share a CU); !DBUF: one stage, two barriers, half the LDS (4 blocks per CU on the big grids).
template <int MR, int NR, int WGM, int WGN, bool DGRAD, int MINW = 2, bool DBUF = false>
__global__ __launch_bounds__(256, MINW) void conv_igemm_f32_kernel(const ConvArgs a) {
    constexpr int BM = 32 * MR * WGM, BN = 32 * NR * WGN;
    constexpr int A_IT = BM / 32, B_IT = BN / 32;
    constexpr int STAGE = (BM + BN) * LDS_LD;                    // floats per LDS stage
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int tile = a.xcd_remap ? xcd_contiguous(blockIdx.x, a.mtiles * a.ntiles) : blockIdx.x;
    const int m0 = (tile / a.ntiles) * BM, n0 = (tile % a.ntiles) * BN, z = blockIdx.z;      // n fastest: the N tiles of one M tile are neighbours
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
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)a.w_bytes, 0x00020000);
    unsigned b_off[B_IT];         // byte offset of filter row k (kOOB past the last output channel)
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int k = n0 + r0 + 32 * i;
        b_off[i] = k < a.K ? (unsigned)k * (unsigned)(RS * a.C) * 4u : kOOB;
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
    unsigned a_off[A_IT];      // byte offset of the input pixel of the current tap, kOOB if it is zero padding
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
            a_off[i] = ok ? (unsigned)((a_n[i] * a.H + hi) * a.W + wi) * (unsigned)a.ldx * 4u : kOOB;
        }
    };
    // PF register sets: the loads of chunk q+PF are in flight while chunk q feeds the MFMAs (PF = 2 for the small tiles, whose
    // per-chunk MFMA phase of ~1000 cycles is shorter than the L2/HBM latency of a load)
    constexpr int PF = 1;      // measured on MI355X: a second prefetch stage (PF = 2) buys nothing on any layer of the step
    float4 ra0[A_IT], rb0[B_IT], ra1[PF == 2 ? A_IT : 1], rb1[PF == 2 ? B_IT : 1];
    auto gload = [&](float4* ra, float4* rb, int t, int ch) {
        const int c = ch * BK + c4 * 4;
        const unsigned coff = c < a.C ? (unsigned)c * 4u : kOOB;          // channel tail of the last chunk reads as zeros
        const unsigned woff = __builtin_elementwise_add_sat(coff, (unsigned)(t * a.C) * 4u);       // saturating: two out-of-range parts must not wrap back into range
#pragma unroll
        for (int i = 0; i < A_IT; ++i) ra[i] = buf_load4(xr, __builtin_elementwise_add_sat(a_off[i], coff));
#pragma unroll
        for (int i = 0; i < B_IT; ++i) rb[i] = buf_load4(wr, __builtin_elementwise_add_sat(b_off[i], woff));
    };

    const int frag_row = lane & 31, frag_k = (lane >> 5) * 4;
    auto lds_store = [&](const float4* ra, const float4* rb, float* As, float* Bs) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *reinterpret_cast<float4*>(&As[(r0 + 32 * i) * LDS_LD + c4 * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) *reinterpret_cast<float4*>(&Bs[(r0 + 32 * i) * LDS_LD + c4 * 4]) = rb[i];
    };
    int issued = q0;           // chunks whose loads have been issued
    auto issue = [&](float4* ra, float4* rb) {
        if (issued >= q1) return;
        if (issued > q0 && ++cc == a.cchunks) { cc = 0; rem_mask &= rem_mask - 1; tap = __builtin_ctzll(rem_mask); set_tap(tap); }
        gload(ra, rb, tap, cc);
        ++issued;
    };
    auto compute = [&](const float* As, const float* Bs) {
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
    };
    if (q0 < q1) set_tap(tap);
    if (DBUF) {
        if (q0 < q1) {
            issue(ra0, rb0);
            lds_store(ra0, rb0, smem, smem + BM * LDS_LD);
            issue(ra0, rb0);
            __syncthreads();
        }
        for (int q = q0; q < q1; ++q) {
            const int cur = (q - q0) & 1;
            float* nA = smem + (cur ^ 1) * STAGE;
            if (q + 1 < q1) {
                lds_store(ra0, rb0, nA, nA + BM * LDS_LD);   // stage q+1 (free since the barrier that ended iteration q-1)
                issue(ra0, rb0);                              // chunk q+2 flies during this iteration's MFMAs
            }
            const float* cA = smem + cur * STAGE;
            compute(cA, cA + BM * LDS_LD);
            __syncthreads();
        }
    } else {
        float* As = smem;
        float* Bs = smem + BM * LDS_LD;
        issue(ra0, rb0);
        if (PF == 2) issue(ra1, rb1);
        for (int q = q0; q < q1; q += PF) {
            lds_store(ra0, rb0, As, Bs);
            __syncthreads();
            issue(ra0, rb0);                                  // chunk q+PF
            compute(As, Bs);
            __syncthreads();
            if (PF == 2 && q + 1 < q1) {
                lds_store(ra1, rb1, As, Bs);
                __syncthreads();
                issue(ra1, rb1);                              // chunk q+3
                compute(As, Bs);
                __syncthreads();
            }
        }
    }

    // ---- epilogue: D[row][col], col = lane&31 (out channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5); bounds by the descriptor
    float* yout = a.y + (a.splits > 1 ? (long long)z * a.slab : 0ll);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)yout, 0, (int)a.y_bytes, 0x00020000);
    const int col = lane & 31, rq = (lane >> 5) * 4;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int k = n0 + (wn * NR + j) * 32 + col;
        const bool kok = k < a.K;
        const float bv = (a.bias != nullptr && kok) ? a.bias[k] : 0.f;
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            const int mb = m0 + (wm * MR + i) * 32 + rq;
            if (a.accumulate && a.splits == 1) {        // y += result: all 16 old values of the tile are fetched before the first store
                float old[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = mb + (e & 3) + 8 * (e >> 2);
                    const unsigned off = (kok && m < a.M) ? ((unsigned)m * (unsigned)a.ldy + (unsigned)k) * 4u : kOOB;
                    old[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yr, (int)off, 0, 0));     // out-of-bounds offsets read 0
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] += old[e];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mb + (e & 3) + 8 * (e >> 2);
                const unsigned off = (kok && m < a.M) ? ((unsigned)m * (unsigned)a.ldy + (unsigned)k) * 4u : kOOB;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][e] + bv), yr, (int)off, 0, 0);
            }
        }
    }
}

}  // namespace dsrl
#include "conv_split_kernel.h"      // conv_igemm_split_kernel
namespace dsrl {


// y[m*ldy + k] = sum_z slab[z][m][k] + bias[k]
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, int splits, long long slab, int M, int K,
                                     const float* __restrict__ bias, float* __restrict__ y, int ldy, int accumulate = 0) {
    const long long total = (long long)M * K;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int zz = 0; zz < splits; ++zz) s += slabs[zz * slab + e];
        const int m = (int)(e / K), k = (int)(e - (long long)m * K);
        float* o = y + (long long)m * ldy + k;
        const float v = s + (bias ? bias[k] : 0.f);
        *o = accumulate ? v + *o : v;
    }
}

// wt[c][tap][k] = w[k][tap][c]   (dgrad runs the forward kernel on the transposed filter)
// The same reduce for a forward conv whose output feeds a BatchNorm (round 2): a block sums the slabs of 64 rows x 32 channels with float4
// loads, stores y and leaves the (n, mean, M2) partial of its 64 rows per channel in stats[3][ceil(M/64)][K] - the layout the conv epilogue
// writes - so that the BatchNorm of a split-K conv (layer4, ASPP) also runs from statistics instead of crossing a device-wide barrier.
// K % 32 == 0, ldy % 4 == 0; Chan merges in a fixed order (row lanes by xor shuffles, the four waves through LDS).
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(const float* __restrict__ slabs, int splits, long long slab, int M, int K,
                                                                   const float* __restrict__ bias, float* __restrict__ y, int ldy, float* __restrict__ stats) {
    __shared__ float shw[4][3][4][8];
    const int groups = K / 32, grp = blockIdx.x % groups, rb = blockIdx.x / groups;
    const int tid = threadIdx.x, l8 = tid & 7, rr = tid >> 3, wave = tid >> 6;
    const int c0 = grp * 32 + 4 * l8, nparts = (M + 63) / 64;
    const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    float v[2][4]; bool ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = rb * 64 + rr + 32 * i;
        ok[i] = m < M;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok[i]) {
            const float* src = slabs + (long long)m * K + c0;
            for (int zz = 0; zz < splits; ++zz) { const float4 t = *reinterpret_cast<const float4*>(src + zz * slab); acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w; }
            acc.x += b4.x; acc.y += b4.y; acc.z += b4.z; acc.w += b4.w;
            *reinterpret_cast<float4*>(y + (long long)m * ldy + c0) = acc;
        }
        v[i][0] = acc.x; v[i][1] = acc.y; v[i][2] = acc.z; v[i][3] = acc.w;
    }
    float n = (ok[0] ? 1.f : 0.f) + (ok[1] ? 1.f : 0.f), mean[4], m2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mean[j] = n > 0.f ? ((ok[0] ? v[0][j] : 0.f) + (ok[1] ? v[1][j] : 0.f)) / n : 0.f;
        const float d0 = v[0][j] - mean[j], d1 = v[1][j] - mean[j];
        m2[j] = (ok[0] ? d0 * d0 : 0.f) + (ok[1] ? d1 * d1 : 0.f);
    }
    auto merge = [](float& na, float& ma, float& qa, float nb, float mb, float qb) {
        if (nb > 0.f) { const float nt = na + nb, d = mb - ma; ma += d * (nb / nt); qa += qb + d * d * (na * nb / nt); na = nt; }
    };
#pragma unroll
    for (int sft = 8; sft < 64; sft <<= 1) {
        const float nb = __shfl_xor(n, sft);
#pragma unroll
        for (int j = 0; j < 4; ++j) { float na = n; merge(na, mean[j], m2[j], nb, __shfl_xor(mean[j], sft), __shfl_xor(m2[j], sft)); }
        n += nb;
    }
    if ((tid & 63) < 8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { shw[wave][0][j][l8] = n; shw[wave][1][j][l8] = mean[j]; shw[wave][2][j][l8] = m2[j]; }
    }
    __syncthreads();
    if (tid < 32) {
        const int j = tid & 3, c8 = tid >> 2;
        float na = shw[0][0][j][c8], ma = shw[0][1][j][c8], qa = shw[0][2][j][c8];
#pragma unroll
        for (int w = 1; w < 4; ++w) merge(na, ma, qa, shw[w][0][j][c8], shw[w][1][j][c8], shw[w][2][j][c8]);
        float* o = stats + (long long)rb * K + grp * 32 + tid;
        o[0] = na; o[(long long)nparts * K] = ma; o[2ll * nparts * K] = qa;
    }
}
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

// The same for every conv filter of the model in ONE launch: table[i] = {w, wt, K, Kp, RS, C, first tile, tiles along C} as int64;
// blockIdx.x = global 32x32 tile index, located in the table by bisection on the first-tile column.
constexpr int kWtTilesPerBlock = 4;
constexpr int kWtRow = 10;      // int64 words per table row: {w, wt, K, Kp, RS, C, first tile, tiles along C, amax word (0: none), unused}
// A block transposes kWtTilesPerBlock consecutive 32x32 tiles (consecutive ids are neighbours along C: contiguous reads): one bisection over
// the rows of the table per block instead of per 4 KB tile, four times the bytes in flight per block.
__global__ __launch_bounds__(256) void weight_transpose_batched_kernel(const long long* __restrict__ table, int n, long long total_tiles) {
    __shared__ float tile[kWtTilesPerBlock][32][33];
    __shared__ unsigned smax[kWtTilesPerBlock];
    if (threadIdx.x < kWtTilesPerBlock) smax[threadIdx.x] = 0u;
    __syncthreads();
    const long long b0 = (long long)blockIdx.x * kWtTilesPerBlock;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * kWtRow + 6] <= b0) lo = mid; else hi = mid - 1;
    }
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    float* dst[kWtTilesPerBlock]; int dC[kWtTilesPerBlock], dKp[kWtTilesPerBlock], dc0[kWtTilesPerBlock], dk0[kWtTilesPerBlock], dRS[kWtTilesPerBlock], dtap[kWtTilesPerBlock];
    unsigned* dam[kWtTilesPerBlock];            // where the filter's max |w| goes (f16x3 operand scale), or null
#pragma unroll
    for (int u = 0; u < kWtTilesPerBlock; ++u) {
        const long long b = b0 + u;
        dst[u] = nullptr; dam[u] = nullptr;
        if (b >= total_tiles) continue;
        while (lo + 1 < n && table[(lo + 1) * kWtRow + 6] <= b) ++lo;            // the next tile may belong to the next filter
        const long long* e = table + lo * kWtRow;
        dam[u] = reinterpret_cast<unsigned*>(e[8]);
        const float* w = reinterpret_cast<const float*>(e[0]);
        const int K = (int)e[2], Kp = (int)e[3], RS = (int)e[4], C = (int)e[5], ct = (int)e[7];
        const int kt = (Kp + 31) / 32;
        int t = (int)(b - e[6]);
        const int tap = t / (ct * kt); t -= tap * ct * kt;
        const int k0 = (t / ct) * 32, c0 = (t % ct) * 32;
        dst[u] = reinterpret_cast<float*>(e[1]); dC[u] = C; dKp[u] = Kp; dc0[u] = c0; dk0[u] = k0; dRS[u] = RS; dtap[u] = tap;
        unsigned mx = 0u;
#pragma unroll
        for (int r = ty; r < 32; r += 8) {
            const int k = k0 + r, c = c0 + tx;
            const float v = (k < K && c < C) ? w[((long long)k * RS + tap) * C + c] : 0.f;
            tile[u][r][tx] = v;
            mx = max(mx, __float_as_uint(v) & 0x7fffffffu);
        }
        if (dam[u] != nullptr) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o, 64));
            if ((threadIdx.x & 63) == 0 && mx) atomicMax(&smax[u], mx);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {                 // one global atomic per (block, filter), into the block's shard of the filter's record
        unsigned* cur = nullptr; unsigned m = 0u;
#pragma unroll
        for (int u = 0; u < kWtTilesPerBlock; ++u) {
            if (dam[u] != cur) { if (cur != nullptr && m) atomicMax(amax_shard(cur), m); cur = dam[u]; m = 0u; }
            m = max(m, smax[u]);
        }
        if (cur != nullptr && m) atomicMax(amax_shard(cur), m);
    }
#pragma unroll
    for (int u = 0; u < kWtTilesPerBlock; ++u) {
        if (dst[u] == nullptr) continue;
#pragma unroll
        for (int r = ty; r < 32; r += 8) {
            const int c = dc0[u] + r, k = dk0[u] + tx;
            if (c < dC[u] && k < dKp[u]) dst[u][((long long)c * dRS[u] + dtap[u]) * dKp[u] + k] = tile[u][tx][r];      // k in [K,Kp): zero padding
        }
    }
}

__global__ __launch_bounds__(256) void zero_fill_kernel(uint4* __restrict__ p, long long n16, unsigned char* __restrict__ tail, int ntail) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}
// max |x| of a pixel-major [P][ld] tensor with C channels as a bit pattern, atomically maxed into the amax record `out` (the caller zeroes
// it; common.h): the operand scale of the f16x3 kernels when the tensor's producer did not leave one.  NaN bit patterns compare above every number.
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, int ld, long long P, int C, int vec, unsigned* __restrict__ out) {
    __shared__ unsigned sm[4];
    unsigned m = 0u;
    const long long stride = (long long)gridDim.x * blockDim.x, t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const int C4 = C >> 2;
        const long long n = P * C4;
        for (long long e = t0; e < n; e += stride) {
            const long long p = ld == C ? 0 : e / C4;
            const float4 v = *reinterpret_cast<const float4*>(ld == C ? x + 4 * e : x + p * ld + 4 * (e - p * C4));
            m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), max(__float_as_uint(v.y) & 0x7fffffffu, max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu)));
        }
    } else {
        const long long n = P * C;
        for (long long e = t0; e < n; e += stride) {
            const long long p = e / C;
            m = max(m, __float_as_uint(x[p * ld + (e - p * C)]) & 0x7fffffffu);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(sm[0], sm[1]), max(sm[2], sm[3]));
        if (m) atomicMax(amax_shard(out), m);
    }
}

// max |w| of every filter of the model in ONE streaming launch (round 3; the batched transpose in its "amax only" form walked 32 x 32 tiles through LDS
// at 1.7 TB/s just to take a maximum): a block takes one segment {pointer, floats (<= kAmaxSegFloats), amax record} of a filter's contiguous
// [K][R][S][C] storage and maxes it into that filter's record (zeroed by the caller).  Table: nseg rows of 3 int64.
constexpr int kAmaxSegFloats = 32768;
__global__ __launch_bounds__(256) void weight_amax_batched_kernel(const long long* __restrict__ table) {
    const long long* e = table + 3ll * blockIdx.x;
    const float* w = reinterpret_cast<const float*>(e[0]);
    const int n = (int)e[1];
    unsigned* rec = reinterpret_cast<unsigned*>(e[2]);
    unsigned m = 0u;
    if ((reinterpret_cast<uintptr_t>(w) & 15) == 0) {
        const int n4 = n >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) {
            const float4 v = reinterpret_cast<const float4*>(w)[i];
            m = abs_bits4(m, v.x, v.y, v.z, v.w);
        }
        for (int i = 4 * n4 + threadIdx.x; i < n; i += 256) m = max(m, abs_bits(w[i]));
    } else {
        for (int i = threadIdx.x; i < n; i += 256) m = max(m, abs_bits(w[i]));
    }
    amax_publish(m, rec);
}

// Filters in "plane" form for the f16x3 kernels (ARITH = 2), all filters of the model in one launch, once per training step behind
// weight_transpose_batched_kernel (which leaves every filter's amax record): for each 32 x 32 (k, c) tile of a tap, w_split [K][RS][C] and
// wt_split [C][RS][Kp] receive, per 4 consecutive elements of the contiguous dimension, the 4 first terms f16(v * 2^e) followed by the 4
// second terms f16(v * 2^e - first) - 16 bytes where the fp32 filter has 16 bytes, so both are indexed exactly like w and wt.
// Table rows as for the transpose, kWtRow int64: {w, wt_split, K, Kp, RS, C, first tile, tiles along C, amax record, w_split}.
__device__ __forceinline__ uint4 split4_f16(float a, float b, float c, float d, int sh) {
    float r[4] = {__builtin_ldexpf(a, sh), __builtin_ldexpf(b, sh), __builtin_ldexpf(c, sh), __builtin_ldexpf(d, sh)};
    const f16x4 t = Plane<true>::cvt(r);
    Plane<true>::residual(r, t);
    const f16x4 u = Plane<true>::cvt(r);
    const uint2 hi = __builtin_bit_cast(uint2, t), lo = __builtin_bit_cast(uint2, u);
    return make_uint4(hi.x, hi.y, lo.x, lo.y);
}
__global__ __launch_bounds__(256) void weight_split_batched_kernel(const long long* __restrict__ table, int n, long long total_tiles) {
    // A block walks kWtTilesPerBlock consecutive 32x32 tiles.  Round 5: the next tile's loads (and its filter's magnitude record, when the filter changes) are
    // in flight while the current tile is split and written - the loop was load -> barrier -> write -> barrier per tile, 3.8 TB/s for 696 MB.
    __shared__ float tile[32][33];
    const long long b0 = (long long)blockIdx.x * kWtTilesPerBlock;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * kWtRow + 6] <= b0) lo = mid; else hi = mid - 1;
    }
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, row = threadIdx.x >> 3, g4 = (threadIdx.x & 7) * 4;
    struct Tile { const long long* e; int tap, k0, c0; };
    auto decode = [&](long long b) -> Tile {
        while (lo + 1 < n && table[(lo + 1) * kWtRow + 6] <= b) ++lo;
        const long long* e = table + lo * kWtRow;
        const int Kp = (int)e[3], ct = (int)e[7], kt = (Kp + 31) / 32;
        int t = (int)(b - e[6]);
        const int tap = t / (ct * kt); t -= tap * ct * kt;
        return Tile{e, tap, (t / ct) * 32, (t % ct) * 32};
    };
    auto fetch = [&](const Tile& T, float (&v)[4]) {
        const float* w = reinterpret_cast<const float*>(T.e[0]);
        const int K = (int)T.e[2], RS = (int)T.e[4], C = (int)T.e[5];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = T.k0 + ty + 8 * i, c = T.c0 + tx;
            v[i] = (k < K && c < C) ? w[((long long)k * RS + T.tap) * C + c] : 0.f;
        }
    };
    Tile cur = decode(b0);
    float v[4];
    fetch(cur, v);
    unsigned am = amax_fetch(reinterpret_cast<const unsigned*>(cur.e[8]));
    const long long* am_of = cur.e;
    for (int u = 0; u < kWtTilesPerBlock; ++u) {
        if (b0 + u >= total_tiles) break;
        __syncthreads();                        // the previous tile has been read out
#pragma unroll
        for (int i = 0; i < 4; ++i) tile[ty + 8 * i][tx] = v[i];
        const int sh = amax_shift_of(am);
        __syncthreads();
        const Tile me = cur;
        if (u + 1 < kWtTilesPerBlock && b0 + u + 1 < total_tiles) {
            cur = decode(b0 + u + 1);
            fetch(cur, v);
            if (cur.e != am_of) { am = amax_fetch(reinterpret_cast<const unsigned*>(cur.e[8])); am_of = cur.e; }
        }
        uint4* wts = reinterpret_cast<uint4*>(me.e[1]);
        uint4* wsp = reinterpret_cast<uint4*>(me.e[9]);
        const int K = (int)me.e[2], Kp = (int)me.e[3], RS = (int)me.e[4], C = (int)me.e[5];
        {   // w_split: row = out channel, 4 consecutive input channels per thread
            const int k = me.k0 + row, c = me.c0 + g4;
            if (wsp != nullptr && k < K && c < C) wsp[(((long long)k * RS + me.tap) * C + c) >> 2] = split4_f16(tile[row][g4], tile[row][g4 + 1], tile[row][g4 + 2], tile[row][g4 + 3], sh);
        }
        {   // wt_split: row = input channel, 4 consecutive out channels per thread (k in [K, Kp): zero padding)
            const int c = me.c0 + row, k = me.k0 + g4;
            if (wts != nullptr && c < C && k < Kp) wts[(((long long)c * RS + me.tap) * Kp + k) >> 2] = split4_f16(tile[g4][row], tile[g4 + 1][row], tile[g4 + 2][row], tile[g4 + 3][row], sh);
        }
    }
}

// ------------------------------------------------------------------------------------------------ wgrad

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
    // 1-D launch: id -> (pixel split z, tap, k tile, c tile), z slowest; with xcd_remap every XCD owns a contiguous range of ids,
    // i.e. whole pixel ranges, so a dy / x chunk is fetched into one L2 instead of eight
    const int per_z = a.ntaps * a.kctiles;
    const int nb = per_z * a.psplits;
    const int id = a.xcd_remap ? xcd_contiguous(blockIdx.x, nb) : blockIdx.x;
    const int zsplit = id / per_z, rem_id = id - zsplit * per_z;
    const int tapi = rem_id / a.kctiles, kc = rem_id - tapi * a.kctiles;
    const int kt = kc / a.ctiles, ct = kc - kt * a.ctiles;
    const int k0 = kt * BM, c0 = ct * BN;
    const int tap = a.taps[tapi];
    const int r = tap / a.S, s = tap - r * a.S;
    const int dh = r * a.dil - a.pad, dw_ = s * a.dil - a.pad;
    const long long nchunks = (a.P + BP - 1) / BP;
    const long long ch0 = nchunks * zsplit / a.psplits, ch1 = nchunks * (zsplit + 1) / a.psplits;
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

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)a.dy_bytes, 0x00020000);
    const unsigned a_coff = a_cok ? (unsigned)(k0 + a_col) * 4u : kOOB, b_coff = b_cok ? (unsigned)(c0 + b_col) * 4u : kOOB;
    const int Pi = (int)a.P;
    float4 ra[A_IT], rb[B_IT];
    auto gload = [&](long long ch) {
        const int pb = (int)ch * BP;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int p = pb + a_row + i * A_RP;
            ra[i] = buf_load4(dr, __builtin_elementwise_add_sat(p < Pi ? (unsigned)p * (unsigned)a.lddy * 4u : kOOB, a_coff));
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int p = pb + b_row + i * B_RP;
            unsigned off = kOOB;
            if (p < Pi) {
                const int n = p / HoWo, rem = p - n * HoWo;
                const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
                const int hi = ho * a.stride + dh, wi = wo * a.stride + dw_;
                if (hi >= 0 && hi < a.H && wi >= 0 && wi < a.W) off = (unsigned)((n * a.H + hi) * a.W + wi) * (unsigned)a.ldx * 4u;
            }
            rb[i] = buf_load4(xr, __builtin_elementwise_add_sat(off, b_coff));
        }
    };
    // a 32-pixel chunk whose pixels all read zero padding for this tap contributes nothing: skip it (block-uniform test)
    auto chunk_live = [&](long long ch) -> bool {
        const int pf = (int)ch * BP, pl = min(pf + BP, Pi) - 1;
        const int nf = pf / HoWo, nl = pl / HoWo;
        if (nf != nl) return true;
        const int hf = (pf - nf * HoWo) / a.Wo, hl = (pl - nl * HoWo) / a.Wo;
        if (hl * a.stride + dh < 0 || hf * a.stride + dh >= a.H) return false;
        if (hf == hl) {
            const int wf = pf - nf * HoWo - hf * a.Wo, wl = pl - nl * HoWo - hl * a.Wo;
            if (wl * a.stride + dw_ < 0 || wf * a.stride + dw_ >= a.W) return false;
        }
        return true;
    };
    auto next_live = [&](long long ch) { while (ch < ch1 && !chunk_live(ch)) ++ch; return ch; };
    long long ch = next_live(ch0);
    if (ch < ch1) gload(ch);
    const int fi = lane & 31, fh = lane >> 5;
    while (ch < ch1) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *reinterpret_cast<float4*>(&As[(a_row + i * A_RP) * BM + a_col]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) *reinterpret_cast<float4*>(&Bs[(b_row + i * B_RP) * BN + b_col]) = rb[i];
        __syncthreads();
        const long long nxt = next_live(ch + 1);
        if (nxt < ch1) gload(nxt);
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
        ch = nxt;
    }
    float* out = a.dw + (a.psplits > 1 ? (long long)zsplit * a.slab : 0ll);
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

// ------------------------------------------------------------------------------------------------ split-precision wgrad

// bf16x3 / bf16x6 weight gradient, software-pipelined like conv_igemm_split_kernel: a 32-pixel chunk is fetched into one of two
// register sets and consumed as two 16-pixel half-steps (= one 16-deep MFMA step each) through two LDS stages; the split /
// convert work of the next half-step is interleaved with the MFMAs of the current one.  LDS keeps the chunks pixel-major
// ([16 px][BM | BN] bf16 per plane, row stride + 64 B) and the K-contiguous MFMA operands are gathered with
// ds_read_b64_tr_b16 (per 16-lane group: lane 4q+p addresses row q / columns 4p..4p+3, lane i receives column i of the 4 rows).
// The pixel -> (n, ho, wo) decomposition of every staged x row uses magic-number division (two mul-hi instead of two divides).
// KG > 1 (pixel groups): KG groups of 4 waves per block, group g takes the 32-pixel chunks g, g+KG, ... of the block's pixel range with
// its own two LDS stages; the KG accumulator sets are summed through LDS in a fixed order - KG times fewer slabs to write and reduce.
__device__ __forceinline__ bool tap_centred_stride1(const WgradArgs& a, int dh, int dw_) {
    return !a.no_ident && dh == 0 && dw_ == 0 && a.stride == 1 && a.Ho == a.H && a.Wo == a.W;
}
template <int MR, int NR, int WGM, int WGN, int NPL, int KG, bool F16 = false>
__device__ __forceinline__ void wgrad_split_body(const WgradArgs& a, const int bid) {
    static_assert(!F16 || NPL <= 2, "f16x3 carries two fp16 terms per operand, f16x1 one");
    using PT = Plane<F16>;
    using pl4 = typename PT::v4; using pl8 = typename PT::v8;
    // f16x3 operand scales: records requested first, consumed behind the first operand loads (see conv_igemm_split_kernel)
    const unsigned am_a = F16 ? amax_fetch(a.amax_dy) : 0u, am_b = F16 ? amax_fetch(a.amax_x) : 0u;
    int sh_a = 0, sh_b = 0;
    auto take_scales = [&]() {
        if (F16) { __builtin_amdgcn_sched_barrier(0); sh_a = amax_shift_of(am_a); sh_b = amax_shift_of(am_b); }
    };
    constexpr int BM = 32 * MR * WGM, BN = 32 * NR * WGN;
    constexpr int A_V = BM / 4, B_V = BN / 4;                 // float4 per pixel row
    constexpr int A_RP = 256 / A_V, B_RP = 256 / B_V;          // pixel rows per staging pass
    constexpr int A_IT = 32 / A_RP, B_IT = 32 / B_RP;          // passes per 32-pixel chunk
    static_assert(A_IT >= 2, "the dy tile is staged in at least two passes per chunk");
    constexpr int A_H = A_IT / 2, B_H = B_IT >= 2 ? B_IT / 2 : 1;      // staged values per half-step
    constexpr int SA = BM * 2 + 64, SB = BN * 2 + 64;          // LDS row strides in bytes
    constexpr int PLA = 16 * SA, PLB = 16 * SB, OFF_B = NPL * PLA, STAGE = NPL * (PLA + PLB);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int grp = KG > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;
    char* const S0 = reinterpret_cast<char*>(smem) + grp * 2 * STAGE;
    char* const S1 = S0 + STAGE;

    const int tid = threadIdx.x & 255, lane = tid & 63;         // thread within its pixel group
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int per_z = a.ntaps * a.kctiles;
    const int nb = per_z * a.psplits;
    const int id = a.xcd_remap ? xcd_contiguous(bid, nb) : bid;
    const int zsplit = id / per_z, rem_id = id - zsplit * per_z;
    const int tapi = rem_id / a.kctiles, kc = rem_id - tapi * a.kctiles;
    const int kt = kc / a.ctiles, ct = kc - kt * a.ctiles;
    const int k0 = kt * BM, c0 = ct * BN;
    const int tap = a.taps[tapi];
    const int r = tap / a.S, s = tap - r * a.S;
    const int dh = r * a.dil - a.pad, dw_ = s * a.dil - a.pad;
    const long long nchunks = (a.P + 31) / 32;
    const int ch0 = (int)(nchunks * zsplit / a.psplits), ch1 = (int)(nchunks * (zsplit + 1) / a.psplits);
    // Loop-invariant fields as values: in a grouped launch `a` is a table entry in global memory, and a field read inside the chunk loop
    // would be a scalar load + wait per use (the stores of the loop keep the compiler from hoisting them).
    const int g_lddy = a.lddy, g_ldx = a.ldx, g_H = a.H, g_W = a.W, g_Wo = a.Wo, g_stride = a.stride;
    const unsigned g_mHW = a.mHW, g_sHW = a.sHW, g_mW = a.mW, g_sW = a.sW;
    const int HoWo = a.Ho * g_Wo;

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

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)a.dy_bytes, 0x00020000);
    const unsigned a_coff = a_cok ? (unsigned)(k0 + a_col) * 4u : kOOB, b_coff = b_cok ? (unsigned)(c0 + b_col) * 4u : kOOB;
    const int Pi = (int)a.P;

    // register sets: RA[pass], RB[pass] of one 32-pixel chunk
    float4 RA0[A_IT], RB0[B_IT], RA1[A_IT], RB1[B_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) RA0[i] = RA1[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < B_IT; ++i) RB0[i] = RB1[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    // A tap that reads input pixel p for output pixel p (every 1x1 / stride-1 conv, the centre tap of a "same" 3x3): no (n, ho, wo) decomposition and
    // no padding test per staged row - the chunk's base is a scalar, the rows' offsets are formed once, rows past P fall outside the descriptor.
    // (SQ counters of the grouped launch, round 5: 43 % of the SIMD cycles in vector instructions against 34 % in MFMAs, and the per-row address
    // arithmetic was about as many instructions as the fp16 split itself.)
    const bool ident = tap_centred_stride1(a, dh, dw_);
    unsigned a_rel[A_IT], b_rel[B_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) a_rel[i] = __builtin_elementwise_add_sat((unsigned)(a_row + i * A_RP) * (unsigned)g_lddy * 4u, a_coff);
#pragma unroll
    for (int i = 0; i < B_IT; ++i) b_rel[i] = __builtin_elementwise_add_sat((unsigned)(b_row + i * B_RP) * (unsigned)g_ldx * 4u, b_coff);
    auto gload = [&](float4* ra, float4* rb, int ch) {
        const int pb = ch * 32;
        if (ident) {
            const unsigned ba = (unsigned)pb * (unsigned)g_lddy * 4u, bb = (unsigned)pb * (unsigned)g_ldx * 4u;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) ra[i] = buf_load4(dr, __builtin_elementwise_add_sat(ba, a_rel[i]));
#pragma unroll
            for (int i = 0; i < B_IT; ++i) rb[i] = buf_load4(xr, __builtin_elementwise_add_sat(bb, b_rel[i]));
            return;
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int p = pb + a_row + i * A_RP;
            ra[i] = buf_load4(dr, __builtin_elementwise_add_sat(p < Pi ? (unsigned)p * (unsigned)g_lddy * 4u : kOOB, a_coff));
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int p = pb + b_row + i * B_RP;
            unsigned off = kOOB;
            if (p < Pi) {
                const int n = fast_div(p, g_mHW, g_sHW), rem = p - n * HoWo;
                const int ho = fast_div(rem, g_mW, g_sW), wo = rem - ho * g_Wo;
                const int hi = ho * g_stride + dh, wi = wo * g_stride + dw_;
                if (hi >= 0 && hi < g_H && wi >= 0 && wi < g_W) off = (unsigned)((n * g_H + hi) * g_W + wi) * (unsigned)g_ldx * 4u;
            }
            rb[i] = buf_load4(xr, __builtin_elementwise_add_sat(off, b_coff));
        }
    };
    // a 32-pixel chunk whose pixels all read zero padding for this tap contributes nothing: skip it (block-uniform test)
    const bool tap_centred = dh == 0 && dw_ == 0;        // 1x1 convs and centre taps never read padding: no liveness test in their K loop
    auto chunk_live = [&](int ch) -> bool {
        if (tap_centred) return true;
        const int pf = ch * 32, pl = min(pf + 32, Pi) - 1;
        const int nf = fast_div(pf, g_mHW, g_sHW), nl = fast_div(pl, g_mHW, g_sHW);
        if (nf != nl) return true;
        const int hf = fast_div(pf - nf * HoWo, g_mW, g_sW), hl = fast_div(pl - nl * HoWo, g_mW, g_sW);
        if (hl * g_stride + dh < 0 || hf * g_stride + dh >= g_H) return false;
        if (hf == hl) {
            const int wf = pf - nf * HoWo - hf * g_Wo, wl = pl - nl * HoWo - hl * g_Wo;
            if (wl * g_stride + dw_ < 0 || wf * g_stride + dw_ >= g_W) return false;
        }
        return true;
    };
    int ci = ch0;                                  // next chunk to fetch
    auto issue = [&](float4* ra, float4* rb) -> bool {
        while (ci < ch1 && !chunk_live(ci)) ++ci;
        if (ci >= ch1) return false;
        gload(ra, rb, ci);
        ++ci;
        return true;
    };

    // ---- LDS addressing.  Transposed reads: 16-lane group g -> k half (g >> 1), channel half (g & 1); t = 4q + p
    const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int tr_off_a = (8 * (tg >> 1) + tq) * SA + (16 * (tg & 1) + 4 * tp) * 2;
    const int tr_off_b = (8 * (tg >> 1) + tq) * SB + (16 * (tg & 1) + 4 * tp) * 2;
    const int wa_off = a_row * SA + a_col * 2;                               // + v * A_RP * SA
    const int b_lrow = B_IT >= 2 ? b_row : (b_row & 15);
    const int wb_off = OFF_B + b_lrow * SB + b_col * 2;                      // + v * B_RP * SB
    const int b_half = B_IT >= 2 ? -1 : (b_row >> 4);                        // B_IT == 1: the single pass spans both halves
    using lds_bf16x4 = __attribute__((address_space(3))) bf16x4;
    float res[4] = {0.f, 0.f, 0.f, 0.f};
    auto cstep = [&](char* nb_, const float4* ra, const float4* rb, int hf, int c) {     // plane c % NPL of staged value c / NPL
        const int v = c / NPL, pl = c % NPL;
        if (pl == 0) {
            const float4 x = v < A_H ? ra[hf * A_H + v] : rb[B_IT >= 2 ? hf * B_H + (v - A_H) : 0];
            res[0] = x.x; res[1] = x.y; res[2] = x.z; res[3] = x.w;
            if (F16) {
                const int sh = v < A_H ? sh_a : sh_b;
#pragma unroll
                for (int e = 0; e < 4; ++e) res[e] = __builtin_ldexpf(res[e], sh);
            }
        }
        const pl4 t = PT::cvt(res);
        if (v < A_H) {
            *reinterpret_cast<pl4*>(nb_ + pl * PLA + wa_off + v * (A_RP * SA)) = t;
        } else if (B_IT >= 2 || b_half == hf) {
            *reinterpret_cast<pl4*>(nb_ + pl * PLB + wb_off + (v - A_H) * (B_RP * SB)) = t;
        }
        if (pl + 1 < NPL) PT::residual(res, t);
    };
    constexpr int NMFMA = MR * NR * (NPL * (NPL + 1) / 2), CSTEPS = (A_H + B_H) * NPL;
    constexpr int MPS = NMFMA / CSTEPS > 0 ? NMFMA / CSTEPS : 1;
    auto tr8 = [&](const char* q, int stride) -> pl8 {
        const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q));
        const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q + 4 * stride));
        return __builtin_bit_cast(pl8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));         // the transposing read moves 16-bit words whatever they hold
    };
    auto pipe = [&](const char* cur, char* nxt, const float4* ra, const float4* rb, int hf) {
        pl8 fa[MR][NPL], fb[NR][NPL];
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) fa[i][pl] = tr8(cur + pl * PLA + tr_off_a + (wm * MR + i) * 64, SA);
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) fb[j][pl] = tr8(cur + OFF_B + pl * PLB + tr_off_b + (wn * NR + j) * 64, SB);
        __builtin_amdgcn_sched_barrier(0);
        int m = 0;
#pragma unroll
        for (int sum = NPL - 1; sum >= 0; --sum)
#pragma unroll
            for (int pa = 0; pa <= sum; ++pa)
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        acc[i][j] = PT::mfma(fa[i][pa], fb[j][sum - pa], acc[i][j]);
                        if (m % MPS == 0 && m / MPS < CSTEPS) cstep(nxt, ra, rb, hf, m / MPS);
                        __builtin_amdgcn_sched_barrier(0);
                        ++m;
                    }
#pragma unroll
        for (int c = (NMFMA + MPS - 1) / MPS; c < CSTEPS; ++c) cstep(nxt, ra, rb, hf, c);
    };

    if constexpr (KG == 1) {
        bool v0 = issue(RA0, RB0);
        bool v1 = v0 && issue(RA1, RB1);
        take_scales();
        if (v0) {
#pragma unroll
            for (int c = 0; c < CSTEPS; ++c) cstep(S0, RA0, RB0, 0, c);
            __syncthreads();
        }
        while (v0) {
            pipe(S0, S1, RA0, RB0, 1);                     // MFMAs of chunk A / half 0, convert chunk A / half 1
            const bool n0 = v1 && issue(RA0, RB0);         // chunk A+2
            __syncthreads();
            pipe(S1, S0, RA1, RB1, 0);                     // MFMAs of chunk A / half 1, convert chunk B / half 0 (stale if !v1: unused)
            __syncthreads();
            if (!v1) break;
            pipe(S0, S1, RA1, RB1, 1);
            const bool n1 = n0 && issue(RA1, RB1);         // chunk B+2
            __syncthreads();
            pipe(S1, S0, RA0, RB0, 0);
            __syncthreads();
            v0 = n0; v1 = n1;
        }
    } else {
        // every group runs the same number of pipeline iterations (the barriers are block-wide); a chunk past the group's share or
        // one that only sees zero padding is fed as zeros instead of being skipped
        const int nloc = (ch1 - ch0 + KG - 1) / KG;
        int itn = 0;                                   // next local chunk of this group
        auto issue_g = [&](float4* ra, float4* rb) {
            const int ch = ch0 + grp + itn * KG;
            ++itn;
            if (ch < ch1 && chunk_live(ch)) { gload(ra, rb, ch); return; }
#pragma unroll
            for (int i = 0; i < A_IT; ++i) ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < B_IT; ++i) rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        };
        if (nloc > 0) {
            issue_g(RA0, RB0);
            issue_g(RA1, RB1);
        }
        take_scales();
        if (nloc > 0) {
#pragma unroll
            for (int c = 0; c < CSTEPS; ++c) cstep(S0, RA0, RB0, 0, c);
            __syncthreads();
        }
        for (int q = 0; q < nloc; q += 2) {
            pipe(S0, S1, RA0, RB0, 1);
            issue_g(RA0, RB0);
            __syncthreads();
            pipe(S1, S0, RA1, RB1, 0);
            __syncthreads();
            if (q + 1 < nloc) {
                pipe(S0, S1, RA1, RB1, 1);
                issue_g(RA1, RB1);
                __syncthreads();
                pipe(S1, S0, RA0, RB0, 0);
                __syncthreads();
            }
        }
        // ---- sum the KG accumulator sets through LDS (stages are free now), fixed order g = 1 .. KG-1; group 0 stores
        float4* red = reinterpret_cast<float4*>(smem);
        constexpr int NQ = MR * NR * 4;
        if (grp > 0) {
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4)
                        red[((grp - 1) * NQ + (i * NR + j) * 4 + e4) * 256 + tid] =
                            make_float4(acc[i][j][4 * e4], acc[i][j][4 * e4 + 1], acc[i][j][4 * e4 + 2], acc[i][j][4 * e4 + 3]);
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll
        for (int g = 1; g < KG; ++g)
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const float4 v = red[((g - 1) * NQ + (i * NR + j) * 4 + e4) * 256 + tid];
                        acc[i][j][4 * e4] += v.x; acc[i][j][4 * e4 + 1] += v.y; acc[i][j][4 * e4 + 2] += v.z; acc[i][j][4 * e4 + 3] += v.w;
                    }
    }

    float* out = a.dw + (a.psplits > 1 ? (long long)zsplit * a.slab : 0ll);
    const int RS = a.R * a.S;
    const int col = lane & 31, rq = (lane >> 5) * 4;
    const int sh_out = -(sh_a + sh_b);          // f16x3: undo the two operand scales (exact)
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
                if (k < a.K) out[((long long)k * RS + tap) * a.C + c] = F16 ? __builtin_ldexpf(acc[i][j][e], sh_out) : acc[i][j][e];
            }
        }
    }
}

template <int MR, int NR, int WGM, int WGN, int NPL, int KG = 1, bool F16 = false>
__global__ __launch_bounds__(256 * KG, KG == 1 ? 2 : 1) void conv_wgrad_split_kernel(const WgradArgs a) {
    wgrad_split_body<MR, NR, WGM, WGN, NPL, KG, F16>(a, (int)blockIdx.x);
}

// Grouped launch: ONE grid covers the weight gradients of many convolutions that share a tile configuration.  `table` holds one
// WgradArgs per problem, `starts` the first block of each (ascending, multiples of 8 so that a problem's local block ids keep their
// XCD round-robin phase); a block finds its problem by bisection (wave-uniform scalar loads) and runs the ordinary kernel body on it.
template <int MR, int NR, int WGM, int WGN, int NPL, bool F16 = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad_group_kernel(const WgradArgs* __restrict__ table, const int* __restrict__ starts, int nprob) {
    const int b = (int)blockIdx.x;
    int lo = 0, hi = nprob - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (starts[mid] <= b) lo = mid; else hi = mid - 1;
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    const WgradArgs& a = table[lo];
    const int local = b - starts[lo];
    if (local >= a.nblocks) return;             // padding block
    wgrad_split_body<MR, NR, WGM, WGN, NPL, 1, F16>(a, local);
}

struct TapList { int taps[64]; int n; };
// dw[k][tap][c] = sum_z slab[z][k][tap][c] over the active taps; C % 4 == 0, one float4 per thread, slabs unrolled by 4
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, int psplits, long long slab, float* __restrict__ dw,
                                    int K, int RS, int C, TapList tl) {
    const int C4 = C >> 2;
    const long long total = (long long)K * tl.n * C4;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % C4);
        const long long t = e / C4;
        const int ti = (int)(t % tl.n), k = (int)(t / tl.n);
        const long long idx = ((long long)k * RS + tl.taps[ti]) * C + 4 * c4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int zz = 0;
        for (; zz + 4 <= psplits; zz += 4) {
            const float4 a = *reinterpret_cast<const float4*>(slabs + (zz + 0) * slab + idx), b = *reinterpret_cast<const float4*>(slabs + (zz + 1) * slab + idx);
            const float4 c = *reinterpret_cast<const float4*>(slabs + (zz + 2) * slab + idx), d = *reinterpret_cast<const float4*>(slabs + (zz + 3) * slab + idx);
            s.x += (a.x + b.x) + (c.x + d.x); s.y += (a.y + b.y) + (c.y + d.y); s.z += (a.z + b.z) + (c.z + d.z); s.w += (a.w + b.w) + (c.w + d.w);
        }
        for (; zz < psplits; ++zz) {
            const float4 a = *reinterpret_cast<const float4*>(slabs + zz * slab + idx);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        *reinterpret_cast<float4*>(dw + idx) = s;
    }
}

// the same reduce for a table of problems: block b of problem j sums 1024 float4 positions (4 per thread) of its slabs
__global__ __launch_bounds__(256) void wgrad_reduce_group_kernel(const WgradArgs* __restrict__ table, const int* __restrict__ starts, int nprob) {
    const int b = (int)blockIdx.x;
    int lo = 0, hi = nprob - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (starts[mid] <= b) lo = mid; else hi = mid - 1;
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    const WgradArgs& a = table[lo];
    const int local = b - starts[lo];
    if (local >= a.rblocks) return;
    const int C4 = a.C >> 2, RS = a.R * a.S;
    const long long total = (long long)a.K * a.ntaps * C4;
    const float* __restrict__ slabs = a.dw;
    float* __restrict__ dw = a.dw_final;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long e = (long long)local * 1024 + u * 256 + threadIdx.x;
        if (e >= total) break;
        const int c4 = (int)(e % C4);
        const long long t = e / C4;
        const int ti = (int)(t % a.ntaps), k = (int)(t / a.ntaps);
        const long long idx = ((long long)k * RS + a.taps[ti]) * a.C + 4 * c4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int zz = 0;
        for (; zz + 4 <= a.psplits; zz += 4) {
            const float4 p = *reinterpret_cast<const float4*>(slabs + (zz + 0) * a.slab + idx), q = *reinterpret_cast<const float4*>(slabs + (zz + 1) * a.slab + idx);
            const float4 r = *reinterpret_cast<const float4*>(slabs + (zz + 2) * a.slab + idx), d = *reinterpret_cast<const float4*>(slabs + (zz + 3) * a.slab + idx);
            s.x += (p.x + q.x) + (r.x + d.x); s.y += (p.y + q.y) + (r.y + d.y); s.z += (p.z + q.z) + (r.z + d.z); s.w += (p.w + q.w) + (r.w + d.w);
        }
        for (; zz < a.psplits; ++zz) {
            const float4 p = *reinterpret_cast<const float4*>(slabs + zz * a.slab + idx);
            s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
        }
        *reinterpret_cast<float4*>(dw + idx) = s;
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
struct ProfRec { hipEvent_t a, b; int family; double flops, bytes; char tag[64]; };
static std::mutex g_prof_mu;
static int g_prof_stride = 0;                    // 0 = off, n = bracket every n-th conv launch with events
static std::atomic<unsigned> g_prof_seq{0};
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_event_pool;      // events are recycled: recording costs ~1 us, creating them much more
static hipEvent_t prof_event() {
    std::lock_guard<std::mutex> g(g_prof_mu);
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e; hipEventCreate(&e); return e;
}
struct ProfScope {
    bool on; ProfRec r; hipStream_t s;
    ProfScope(int family, double flops, double bytes, hipStream_t st) : on(false), s(st) {
        const int stride = g_prof_stride;
        on = stride > 0 && (g_prof_seq.fetch_add(1) % (unsigned)stride) == 0;
        if (!on) return;
        r.family = family; r.flops = flops; r.bytes = bytes; r.tag[0] = 0;
        r.a = prof_event(); r.b = prof_event();
        hipEventRecord(r.a, s);
    }
    void shape(const char* what, int N, int H, int W, int C, int K, int R, int stride, int pad, int dil) {        // for DSRL_PROF_DUMP
        if (on) snprintf(r.tag, sizeof(r.tag), "%s N%d %dx%d C%d K%d R%d s%d p%d d%d", what, N, H, W, C, K, R, stride, pad, dil);
    }
    ~ProfScope() {
        if (!on) return;
        hipEventRecord(r.b, s);
        std::lock_guard<std::mutex> g(g_prof_mu);
        g_prof.push_back(r);
    }
};

// tile configurations <MR,NR,WGM,WGN>: block tile = (32*MR*WGM) x (32*NR*WGN), always 4 waves
// the last two are 8-wave blocks (512 threads) of the split-precision forward / dgrad kernel only: kNumCfg4 counts the 4-wave tiles every kernel has
enum TileCfg { T128x128, T256x64, T256x32, T64x64, T128x64, T64x128, T128x32, T256x128, T256x256, kNumCfg };
constexpr int kNumCfg4 = T256x128;
static const int kCfgDims[kNumCfg][2] = {{128, 128}, {256, 64}, {256, 32}, {64, 64}, {128, 64}, {64, 128}, {128, 32}, {256, 128}, {256, 256}};
#define DSRL_CFG_SWITCH(cfg, LAUNCH)                 \
    switch (cfg) {                                   \
        case T128x128: LAUNCH(2, 2, 2, 2); break;    \
        case T256x64:  LAUNCH(2, 2, 4, 1); break;    \
        case T256x32:  LAUNCH(2, 1, 4, 1); break;    \
        case T64x64:   LAUNCH(1, 1, 2, 2); break;    \
        case T128x64:  LAUNCH(2, 1, 2, 2); break;    \
        case T64x128:  LAUNCH(1, 2, 2, 2); break;    \
        default:       LAUNCH(1, 1, 4, 1); break;    \
    }
static void cfg_dims(TileCfg c, int& bm, int& bn) { bm = kCfgDims[c][0]; bn = kCfgDims[c][1]; }
// forward / dgrad (rows = pixels M, columns = output channels N).  Measured on MI355X (tools/sweep_igemm.py): take the largest
// tile that still yields >= 3 blocks per CU, fall back to 64x64; split K only when the tap/channel loop is very long (ASPP).
static long long cfg_blocks(TileCfg c, long long M, int N) { return ceil_div(M, kCfgDims[c][0]) * ceil_div(N, kCfgDims[c][1]); }
static int conv_precision_mode();
static TileCfg pick_cfg(long long M, int N) {
    const int forced = env_int("DSRL_FORCE_CFG", -1);
    if (forced >= 0 && forced < kNumCfg && (forced < kNumCfg4 || conv_precision_mode() >= 4)) return (TileCfg)forced;
    if (N <= 32) return cfg_blocks(T256x32, M, N) >= 3 * kNumCU ? T256x32 : T128x32;
    const int r = N % 128;
    const bool narrow = N <= 64 || (r > 0 && r <= 64);
    const TileCfg wide[3] = {T128x128, T128x64, T64x64}, nar[3] = {T256x64, T128x64, T64x64};
    const TileCfg* cand = narrow ? nar : wide;
    for (int i = 0; i < 3; ++i)
        if (cfg_blocks(cand[i], M, N) >= 3 * kNumCU) return cand[i];
    return T64x64;
}
// wgrad: rows are output channels K, columns input channels C.  Measured on MI355X (tools/sweep_wgrad.py): 128x64 tiles
// (64x64 when K <= 64) with ~4.5 blocks per CU beat the larger tiles on every layer shape of the step.
static TileCfg pick_cfg_wgrad(int K, int C) {
    int forced = env_int("DSRL_FORCE_CFG", -1);
    if (forced < 0) forced = env_int("DSRL_WGRAD_CFG", -1);
    if (forced >= 0 && forced < kNumCfg4) return (TileCfg)forced;
    // layers with K >= 128 and C >= 128 (the per-tap grid: 1x1, strided and dilated convs of layers 2-4 and ASPP): 128x128 tiles.  Round 4 measured +-1 % for
    // them; with the 3x3 convs gone to conv_wgrad3_kernel the remaining 1x1 problems gain +0.5 .. 1.0 % step throughput on three boxes (profiles/round5_ab.txt):
    // an activation chunk is fetched and split once per 128 output columns instead of 64.  DSRL_WGRAD_BIG_CFG=-1: 128x64 as before
    const int big = env_int("DSRL_WGRAD_BIG_CFG", 0);
    if (big >= 0 && big < kNumCfg4 && K >= 128 && C >= 128) return (TileCfg)big;
    if (C <= 32) return T128x32;
    return K <= 64 ? T64x64 : T128x64;
}
static int pick_psplits(long long tiles, long long chunks) {
    const int forced = env_int("DSRL_FORCE_PSPLITS", 0);
    if (forced > 0) return (int)std::max<long long>(1, std::min<long long>(forced, chunks));
    long long sp = ceil_div(env_int("DSRL_WGRAD_TARGET_BLOCKS", 1152), std::max<long long>(tiles, 1));
    sp = std::min(sp, std::max<long long>(1, chunks / 4));
    return (int)std::max<long long>(1, std::min<long long>(sp, 128));
}
static int pick_splits(long long tiles, int nq) {
    const int forced = env_int("DSRL_FORCE_SPLITS", 0);
    if (forced > 0) return std::max(1, std::min(forced, std::max(1, nq)));
    if (tiles >= 3 * kNumCU) return 1;
    const long long cap = std::min<long long>(8, ceil_div(6 * kNumCU, std::max<long long>(tiles, 1)));
    return (int)std::max<long long>(1, std::min<long long>(nq / 24, cap));
}

// Conv arithmetic, per pass.  Mode (dsrl_conv_precision(), else DSRL_CONV_PRECISION, else the default 4 = f16x3, fp32-equivalent):
//   0  fp32 MFMA everywhere (v_mfma_f32_32x32x2_f32, exact products)
//   1  bf16x3 everywhere   (16 mantissa bits per operand, ~5e-6 relative error per conv)
//   2  bf16x6 everywhere   (24 mantissa bits per operand: fp32-equivalent, measured error vs fp64 equal to mode 0)
//   3  forward bf16x6, dgrad / wgrad bf16x3 (logits keep fp32 accuracy, gradients carry ~5e-6)
//   4  f16x3 everywhere    (two fp16 terms per operand + per-tensor power-of-two scales: 22 mantissa bits, fp32-equivalent, 3 MFMAs)
//   5  f16x1 everywhere    (ONE fp16 term of the scaled operands, one MFMA per product, fp32 accumulate: the arithmetic of apex O1 / O2 - reduced
//                           precision, 11 mantissa bits per operand; the per-tensor scales stand in for loss scaling)
// Returns the number of 16-bit planes per operand for the pass (0 = fp32 kernel); conv_f16(): are they fp16 (mode 4) or bf16.
enum ConvPass { PASS_FWD, PASS_DGRAD, PASS_WGRAD };
static int conv_precision_mode() {
    int prec = g_conv_precision.load();
    if (prec < 0) prec = env_int("DSRL_CONV_PRECISION", 4);
    return prec < 0 ? 0 : (prec > 5 ? 5 : prec);
}
static bool conv_f16() { return conv_precision_mode() >= 4; }
static bool conv_f16x3() { return conv_precision_mode() == 4; }
static int conv_planes(ConvPass pass) {
    switch (conv_precision_mode()) {
        case 0: return 0;
        case 1: return 2;
        case 2: return 3;
        case 4: return 2;
        case 5: return 1;
        default: return pass == PASS_FWD ? 3 : 2;
    }
}

// launch-timer family = 3 * arithmetic (0 fp32, 1 bf16x3, 2 bf16x6, 3 f16x3, 4 f16x1) + pass (0 forward, 1 wgrad, 2 dgrad)
static int prof_arith(int npl, bool f16) { return f16 ? (npl == 1 ? 4 : 3) : (npl ? npl - 1 : 0); }
static int prof_family(ConvPass pass) {
    const int npl = conv_planes(pass);
    return 3 * prof_arith(npl, conv_f16()) + (pass == PASS_FWD ? 0 : (pass == PASS_WGRAD ? 1 : 2));
}

// f16x3 operand scales the caller did not provide: measured into two zeroed words at `scratch` (kAmaxScratch bytes at the end of the
// call's workspace) by one amax launch per missing operand.
constexpr size_t kAmaxScratch = 2 * kAmaxWords * sizeof(unsigned);      // two records
int launch_zero_fill(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return DSRL_OK;
    if ((uintptr_t)p % 16) {                // unaligned head: byte by byte up to the boundary is not worth a kernel of its own - callers pass 16-byte aligned buffers
        set_error("zero_fill: pointer not 16-byte aligned");
        return DSRL_E_BADARG;
    }
    if (env_int("DSRL_ZERO_FILL_MEMSET", 0)) {      // diagnosis only (tools/graph_memset_edges.py): the memset node this function replaced in round 3
        if (hipMemsetAsync(p, 0, bytes, st) != hipSuccess) { set_error("hipMemsetAsync failed"); return DSRL_E_LAUNCH; }
        return DSRL_OK;
    }
    const long long n16 = (long long)(bytes / 16);
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>(ceil_div(n16, 256 * 8), 2048));
    hipLaunchKernelGGL(zero_fill_kernel, dim3(grid), dim3(256), 0, st, (uint4*)p, n16, (unsigned char*)p + n16 * 16, (int)(bytes % 16));
    return launch_status("zero_fill_kernel");
}
int launch_amax(const float* x, int ld, long long P, int C, unsigned* out, hipStream_t st) {
    const int vec = (C % 4 == 0 && ld % 4 == 0 && ((uintptr_t)x % 16) == 0) ? 1 : 0;
    const long long n = vec ? P * (C / 4) : P * C;
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>(ceil_div(n, 256 * 4), 2048));
    hipLaunchKernelGGL(amax_kernel, dim3(grid), dim3(256), 0, st, x, ld, P, C, vec, out);
    return launch_status("amax_kernel");
}
struct OperandAmax { const unsigned* a; const unsigned* b; };
// a: [Pa][lda] with Ca channels, b: [Pb][ldb] with Cb channels
static int resolve_amax(OperandAmax& am, const float* a, int lda, long long Pa, int Ca, const float* b, int ldb, long long Pb, int Cb,
                        void* ws, size_t ws_bytes, size_t ws_used, hipStream_t st, const char* who) {
    if (am.a != nullptr && am.b != nullptr) return DSRL_OK;
    DSRL_REQUIRE(ws != nullptr && ws_bytes >= align_up(ws_used, 64) + kAmaxScratch, DSRL_E_WORKSPACE, "%s: the f16x3 arithmetic measures operand magnitudes the caller did not pass in %zu bytes behind the first %zu of the workspace (got %zu)", who, kAmaxScratch, ws_used, ws_bytes);
    unsigned* scratch = (unsigned*)((char*)ws + align_up(ws_used, 64));
    if (int e = launch_zero_fill(scratch, kAmaxScratch, st)) return e;
    if (am.a == nullptr) { if (int e = launch_amax(a, lda, Pa, Ca, scratch, st)) return e; am.a = scratch; }
    if (am.b == nullptr) { if (int e = launch_amax(b, ldb, Pb, Cb, scratch + kAmaxWords, st)) return e; am.b = scratch + kAmaxWords; }
    return DSRL_OK;
}

static void make_magic(int d, unsigned& m, unsigned& sh) {
    if (d <= 1) { m = 0; sh = 255u; return; }
    int l = 0;
    while ((1ll << l) < d) ++l;
    const unsigned long long k = 31 + l;
    m = (unsigned)((((unsigned long long)1 << k) + (unsigned long long)d - 1) / (unsigned long long)d);
    sh = (unsigned)(l - 1);
}
template <bool DGRAD>
static int launch_igemm(const ConvArgs& a_in, TileCfg cfg, hipStream_t st) {
    int bm, bn; cfg_dims(cfg, bm, bn);
    ConvArgs a = a_in;
    a.mtiles = (int)ceil_div(a.M, bm); a.ntiles = (int)ceil_div(a.K, bn);
    make_magic(a.Ho * a.Wo, a.mHW, a.sHW); make_magic(a.Wo, a.mW, a.sW); make_magic(a.ntiles, a.mNT, a.sNT);
    a.xcd_remap = env_int("DSRL_XCD_REMAP", 1);
    // the fast BatchNorm-sum epilogue of the one-group builds parks the block's fp32 tile in LDS (conv_split_kernel.h): fp16 arithmetics, tiles up to 256x128
    if (a.bn_fast && !(DGRAD && conv_f16() && (size_t)bm * bn * 4 <= 128 * 1024)) a.bn_fast = 0;
    if (a.planes) return launch_planes_igemm(a, (int)cfg, DGRAD, st);          // both operands as fp16 planes, staged by LDS-DMA (conv_planes.hip)
    dim3 grid((unsigned)(a.mtiles * a.ntiles), 1u, (unsigned)a.splits);
    const int npl = conv_planes(DGRAD ? PASS_DGRAD : PASS_FWD);
    const bool f16 = conv_f16();
    const long long nblocks = (long long)grid.x * grid.y * grid.z;
    if (f16 && (a.amax_a == nullptr || a.amax_b == nullptr)) { set_error("conv_igemm_split_kernel<f16x3>: operand magnitudes missing"); return DSRL_E_BADARG; }
    // stride-1 data gradients with pre-split filters run the build without divisibility tests / parity bookkeeping (template STR1; a forward launch
    // passes DGRAD = false in that slot, i.e. the same instantiation as without it)
    const bool s1 = DGRAD && a.stride == 1 && a.par == 0 && env_int("DSRL_DGRAD_S1", 1);
    if (npl) {
        const int kg = a.kg > 1 ? a.kg : 1;
        const size_t stages = (size_t)2 * (bm + bn) * npl * ((kFullStep && f16) ? 64 : 32);      // two stages per K group: 32-byte rows (a 16-channel half-step), or 64-byte rows (a whole step: fp16 arithmetics)
        if (kg > 1) {
            // K groups (64x64 tiles: 2 or 4 groups, 128x64 / 64x128: 2): block of 256*kg threads, LDS = kg stage pairs or the (kg-1)
            // accumulator sets of the final reduction (all kg of them: every group sums, every group stores a share), whichever is larger (up to 72 KiB)
            if (cfg == T128x128) {          // 8 waves, wave tile 64x64, optionally split-K across workgroups reduced inside the launch (conv_sk.hip)
                if (!sk_supported((int)cfg, kg, npl, f16, a.w_split != 0)) { set_error("conv_igemm_split_kernel: 128x128 tiles with two K groups exist for the fp16 arithmetics only"); return DSRL_E_UNSUPPORTED; }
                return launch_sk_igemm(a, DGRAD, s1, npl, st);
            }
            grid = dim3((unsigned)(a.mtiles * a.ntiles), 1u, (unsigned)std::max(1, a.splits));
            const size_t lds = std::max(stages * kg, (size_t)kg * (bm / 32) * (bn / 32) / 4 * 16384 + 8192);      // every group's accumulators + the BatchNorm-sum exchange
#define DSRL_LAUNCH_KG(a_, b_, c_, d_, NPL_, KG_, F16_, S1_)                                                                          \
            {                                                                                                                          \
                static const hipError_t attr = hipFuncSetAttribute((const void*)conv_igemm_split_kernel<a_, b_, c_, d_, DGRAD, NPL_, KG_, F16_, S1_>, \
                                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);            \
                (void)attr;                                                                                                            \
                hipLaunchKernelGGL((conv_igemm_split_kernel<a_, b_, c_, d_, DGRAD, NPL_, KG_, F16_, S1_>), grid, dim3(256 * KG_), lds, st, a); \
            }
#define DSRL_KG_BY_ARITH(a_, b_, c_, d_, KG_) { if (f16 && npl == 1 && a.w_split) DSRL_LAUNCH_KG(a_, b_, c_, d_, 1, KG_, 2, false) else if (f16 && npl == 1) DSRL_LAUNCH_KG(a_, b_, c_, d_, 1, KG_, 1, false) else if (f16 && a.w_split && s1) DSRL_LAUNCH_KG(a_, b_, c_, d_, 2, KG_, 2, DGRAD) else if (f16 && a.w_split) DSRL_LAUNCH_KG(a_, b_, c_, d_, 2, KG_, 2, false) else if (f16) DSRL_LAUNCH_KG(a_, b_, c_, d_, 2, KG_, 1, false) else if (npl == 2) DSRL_LAUNCH_KG(a_, b_, c_, d_, 2, KG_, 0, false) else DSRL_LAUNCH_KG(a_, b_, c_, d_, 3, KG_, 0, false) }
            if (cfg == T64x64) {
                if (kg == 4) DSRL_KG_BY_ARITH(1, 1, 2, 2, 4) else DSRL_KG_BY_ARITH(1, 1, 2, 2, 2)
            } else if (cfg == T128x64) {
                DSRL_KG_BY_ARITH(2, 1, 2, 2, 2)
            } else {
                DSRL_KG_BY_ARITH(1, 2, 2, 2, 2)
            }
#undef DSRL_KG_BY_ARITH
#undef DSRL_LAUNCH_KG
            return launch_status("conv_igemm_split_kernel<K groups>");
        }
        const size_t lds2 = std::max(stages, a.bn_fast ? (size_t)bm * bn * 4 : (size_t)0);
        if (cfg == T256x128 || cfg == T256x256) {               // 8 waves; f16x3 with or without pre-split filters, and bf16x6
#define DSRL_LAUNCH_BIG(a_, b_, c_, d_)                                                                                                   \
            {                                                                                                                              \
                if (f16 && npl == 1 && a.w_split) {                                                                                        \
                    static const hipError_t at_ = hipFuncSetAttribute((const void*)conv_igemm_split_kernel<a_, b_, c_, d_, DGRAD, 1, 1, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); (void)at_; \
                    hipLaunchKernelGGL((conv_igemm_split_kernel<a_, b_, c_, d_, DGRAD, 1, 1, 2, false>), grid, dim3(512), lds2, st, a);      \
                } else if (f16 && npl == 1) {                                                                                              \
                    static const hipError_t at_ = hipFuncSetAttribute((const void*)conv_igemm_split_kernel<a_, b_, c_, d_, DGRAD, 1, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); (void)at_; \
                    hipLaunchKernelGGL((conv_igemm_split_kernel<a_, b_, c_, d_, DGRAD, 1, 1, 1>), grid, dim3(512), lds2, st, a);             \
                } else if (f16 && a.w_split && s1) {                                                                                       \
                    static const hipError_t at_ = hipFuncSetAttribute((const void*)conv_igemm_split_kernel<a_, b_, c_, d_, DGRAD, 2, 1, 2, DGRAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); (void)at_; \
                    hipLaunchKernelGGL((conv_igemm_split_kernel<a_, b_, c_, d_, DGRAD, 2, 1, 2, DGRAD>), grid, dim3(512), lds2, st, a);      \
                } else if (f16 && a.w_split) {                                                                                             \
                    static const hipError_t at_ = hipFuncSetAttribute((const void*)conv_igemm_split_kernel<a_, b_, c_, d_, DGRAD, 2, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); (void)at_; \
                    hipLaunchKernelGGL((conv_igemm_split_kernel<a_, b_, c_, d_, DGRAD, 2, 1, 2>), grid, dim3(512), lds2, st, a);             \
                } else if (f16) {                                                                                                          \
                    static const hipError_t at_ = hipFuncSetAttribute((const void*)conv_igemm_split_kernel<a_, b_, c_, d_, DGRAD, 2, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); (void)at_; \
                    hipLaunchKernelGGL((conv_igemm_split_kernel<a_, b_, c_, d_, DGRAD, 2, 1, 1>), grid, dim3(512), lds2, st, a);             \
                } else {                                                                                                                   \
                    set_error("conv_igemm_split_kernel: the 8-wave tiles exist for the f16x3 arithmetic only"); return DSRL_E_UNSUPPORTED;   \
                }                                                                                                                          \
            }
            if (cfg == T256x128) DSRL_LAUNCH_BIG(2, 2, 4, 2)
            else {
                // 256x256, register-staged: the data-gradient builds needed 76-88 spilled registers and the planner never picked them (rule d takes 256x128
                // for dgrad): removed in round 5.  The forward build (73 spills with two planes) stays for launches whose operands arrive without planes.
                if constexpr (DGRAD) { set_error("conv_igemm_split_kernel: no 256x256 data-gradient build (256x128 is the 8-wave dgrad tile)"); return DSRL_E_UNSUPPORTED; }
                else DSRL_LAUNCH_BIG(4, 2, 2, 4)
            }
#undef DSRL_LAUNCH_BIG
            return launch_status("conv_igemm_split_kernel<f16x3, 8 waves>");
        }
        if (cfg == T128x128 && f16 && a.tickets != nullptr && a.splits > 1)          // the same plan with the split-K reduction inside the launch (DSRL_SK_COOP1)
            return launch_sk_igemm(a, DGRAD, s1, npl, st);
#define DSRL_SPLIT_ONE(...)                                                                                                  \
        { static const hipError_t at_ = hipFuncSetAttribute((const void*)conv_igemm_split_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); (void)at_; \
          hipLaunchKernelGGL((conv_igemm_split_kernel<__VA_ARGS__>), grid, dim3(256), lds2, st, a); }
#define DSRL_LAUNCH_SPLIT(a_, b_, c_, d_)                                                                                    \
        if (f16 && npl == 1 && a.w_split) DSRL_SPLIT_ONE(a_, b_, c_, d_, DGRAD, 1, 1, 2, false)                               \
        else if (f16 && npl == 1) DSRL_SPLIT_ONE(a_, b_, c_, d_, DGRAD, 1, 1, 1)                                              \
        else if (f16 && a.w_split && s1) DSRL_SPLIT_ONE(a_, b_, c_, d_, DGRAD, 2, 1, 2, DGRAD)                                \
        else if (f16 && a.w_split) DSRL_SPLIT_ONE(a_, b_, c_, d_, DGRAD, 2, 1, 2)                                             \
        else if (f16) DSRL_SPLIT_ONE(a_, b_, c_, d_, DGRAD, 2, 1, 1)                                                          \
        else if (npl == 2) hipLaunchKernelGGL((conv_igemm_split_kernel<a_, b_, c_, d_, DGRAD, 2>), grid, dim3(256), lds2, st, a); \
        else hipLaunchKernelGGL((conv_igemm_split_kernel<a_, b_, c_, d_, DGRAD, 3>), grid, dim3(256), lds2, st, a);
        DSRL_CFG_SWITCH(cfg, DSRL_LAUNCH_SPLIT)
#undef DSRL_LAUNCH_SPLIT
#undef DSRL_SPLIT_ONE
        return launch_status(f16 ? (npl == 1 ? "conv_igemm_split_kernel<f16x1>" : "conv_igemm_split_kernel<f16x3>") : (npl == 2 ? "conv_igemm_split_kernel<bf16x3>" : "conv_igemm_split_kernel<bf16x6>"));
    }
    const size_t lds1 = (size_t)(bm + bn) * LDS_LD * sizeof(float);
    const bool dbuf = env_int("DSRL_IGEMM_DBUF", 0) != 0;     // measured: no gain from the two-stage LDS variant; kept selectable
#define DSRL_LAUNCH_IGEMM(a_, b_, c_, d_) hipLaunchKernelGGL((conv_igemm_f32_kernel<a_, b_, c_, d_, DGRAD>), grid, dim3(256), lds1, st, a)
#define DSRL_LAUNCH_IGEMM_DB(a_, b_, c_, d_) hipLaunchKernelGGL((conv_igemm_f32_kernel<a_, b_, c_, d_, DGRAD, 2, true>), grid, dim3(256), 2 * lds1, st, a)
    // 1024+ tiles of 128x128: the <=128-register build keeps 4 blocks per CU resident (one round instead of 1.33)
    if (cfg == T128x128 && !dbuf && env_int("DSRL_IGEMM_OCC4", nblocks > 3 * kNumCU ? 1 : 0)) {
        hipLaunchKernelGGL((conv_igemm_f32_kernel<2, 2, 2, 2, DGRAD, 4>), grid, dim3(256), lds1, st, a);
    } else if (dbuf) {
        static bool attr_set[2] = {false, false};
        if (2 * lds1 > 65536 && !attr_set[DGRAD ? 1 : 0]) {       // > 64 KiB of dynamic LDS needs the opt-in attribute (256x64 / 128x128 tiles)
            hipFuncSetAttribute((const void*)conv_igemm_f32_kernel<2, 2, 2, 2, DGRAD, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
            hipFuncSetAttribute((const void*)conv_igemm_f32_kernel<2, 2, 4, 1, DGRAD, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
            hipFuncSetAttribute((const void*)conv_igemm_f32_kernel<2, 1, 4, 1, DGRAD, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
            attr_set[DGRAD ? 1 : 0] = true;
        }
        DSRL_CFG_SWITCH(cfg, DSRL_LAUNCH_IGEMM_DB)
    } else {
        DSRL_CFG_SWITCH(cfg, DSRL_LAUNCH_IGEMM)
    }
#undef DSRL_LAUNCH_IGEMM
#undef DSRL_LAUNCH_IGEMM_DB
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

static int wgrad_no_ident() { static const int v = [] { const char* e = getenv("DSRL_WGRAD_IDENT"); return (e && atoi(e) == 0) ? 1 : 0; }(); return v; }
static long long span_bytes(long long pixels, int ld, int c) { return ((pixels - 1) * ld + c) * 4ll; }
#define DSRL_REQUIRE_31(bytes, what) DSRL_REQUIRE((bytes) > 0 && (bytes) < (1ll << 31), DSRL_E_UNSUPPORTED, what ": tensor of %lld bytes exceeds the 2 GiB buffer-descriptor range", (long long)(bytes))
struct FwdPlan { int Ho, Wo, M, cchunks, splits, kg; TileCfg cfg; size_t ws; bool coop; };
// K groups per block for the split-precision kernels (measured on the M = 4096 backbone layers): when the grid has at most ~1.5 tiles
// per CU a block runs 2 or 4 groups of 4 waves over interleaved chunks and sums them in LDS - no slabs, no reduce launch.
static int pick_kg(long long tiles, int nq, TileCfg cfg, int npl) {
    if (!npl) return 1;
    const int forced = env_int("DSRL_FORCE_KG", 0);
    const int maxkg = cfg == T64x64 ? 4 : ((cfg == T128x64 || cfg == T64x128 || (cfg == T128x128 && conv_precision_mode() >= 4)) ? 2 : 1);
    if (forced > 0) return std::min(forced >= 4 ? 4 : (forced >= 2 ? 2 : 1), maxkg);
    if (tiles * 2 > 3 * kNumCU) return 1;
    int kg = 1;
    while (kg * 2 <= maxkg && nq >= 6 * kg * 2) kg *= 2;
    return kg;
}
// Tile / split-K / K-group choice for the split-precision kernels, from tools/sweep_igemm2.py on the step's layer shapes
// (profiles/round1_sweep_igemm_split_kernel_v2.txt).  Starting from the generic pick (largest tile with >= 3 blocks per CU, else 64x64):
//   a) a larger tile already pays at >= 2 blocks per CU (layer1 3x3: 128x64, layer4 / layer2 wide 1x1: 128x128);
//   b) when even 64x64 tiles do not fill the chip, 128x128 tiles with split-K <= 4 beat 64x64 K groups if they reach >= 512 blocks
//      (layer4, M = 4096 and N >= 512), or >= 256 blocks on very long K loops (dilated ASPP convs);
//   c) 385..767 tiles of 64x64 (layer2 3x3): a 64x128 / 128x64 tile with two K groups.
//   d) f16x3 only, the 8-wave tiles (profiles/round3_sweep_8wave_tiles.txt): with a long K loop (>= 32 chunks) and at least one block per
//      CU the forward pass takes 256x256 tiles (cat / SISR 3x3 convs: -8..-17 %), dgrad 256x128 at >= 2 blocks per CU (-4..-7 %; its
//      256x256 build spills); the dilated ASPP forward convs (M = 4096, 576 chunks) take 256x128 with split-K 8 (-22 %).
static void pick_split_plan(long long M, int N, int nq, bool dgrad, TileCfg& cfg, int& splits, int& kg) {
    const bool forced = env_int("DSRL_FORCE_CFG", -1) >= 0 || env_int("DSRL_FORCE_SPLITS", 0) > 0 || env_int("DSRL_FORCE_KG", 0) > 0;
    if (forced || N <= 32 || !env_int("DSRL_SPLIT_PLAN", 1)) return;
    if (conv_precision_mode() >= 4 && env_int("DSRL_BIG_TILES", 1) && N >= 192 && nq >= 32) {                                         // d)
        // 256x256 forward: the LDS-DMA kernel's tile (251 registers, no scratch).  The register-staged two-plane build of that tile spills, so with planes
        // switched off altogether (DSRL_PLANES=0, set by functional when DSRL_PLANES_MODE=off) the f16x3 forward takes 256x128 like dgrad; f16x1 (one plane) fits
        const bool t256 = conv_precision_mode() == 5 || env_int("DSRL_PLANES", 1) != 0;
        if (!dgrad && t256 && cfg_blocks(T256x256, M, N) >= kNumCU) { cfg = T256x256; splits = 1; kg = 1; return; }
        if (!dgrad && !t256 && cfg_blocks(T256x128, M, N) >= 2 * kNumCU) { cfg = T256x128; splits = 1; kg = 1; return; }
        if (dgrad && cfg_blocks(T256x128, M, N) >= 2 * kNumCU) { cfg = T256x128; splits = 1; kg = 1; return; }
        if (!dgrad && nq >= 256 && cfg_blocks(T256x128, M, N) * 8 >= kNumCU && cfg_blocks(T128x128, M, N) < kNumCU) { cfg = T256x128; splits = 8; kg = 1; return; }
    }
    // e) round 5 (conv_sk.hip, profiles/round5_sk_tiles_ab.txt): exactly half a chip of 128x128 tiles on a long K loop (layer4: M = 4096, N = 512, the 3x3
    //    conv and the 2048 -> 512 1x1 and its mirror-image dgrad) - two K groups per tile and split-K 2 across workgroups, reduced inside the launch:
    //    -5 .. -10 % against 128x128 tiles with split-K slabs IN ISOLATION.  In the step the rule came out even when it was introduced and 0.4 % behind
    //    at the end of the round (once bn3's sums from the next block's dgrad and the decoder links were in: profiles/round5_ab.txt), and the layer3 shapes
    //    (64 tiles, split-K 4) lose 3 %: the rule is OFF by default.  DSRL_SK_AUTO=1: the layer4 shapes, =2: also the 64-tile shapes (N = 256, nq >= 64).
    if (conv_precision_mode() >= 4 && N % 128 == 0 && M % 128 == 0) {
        const int sk = env_int("DSRL_SK_AUTO", 0);
        const long long t = cfg_blocks(T128x128, M, N);
        if (sk >= 1 && t == kNumCU / 2 && nq >= 64) { cfg = T128x128; splits = 2; kg = 2; return; }
        if (sk >= 2 && t == kNumCU / 4 && nq >= 64) { cfg = T128x128; splits = 4; kg = 2; return; }
    }
    const int r = N % 128;
    const bool narrow = N <= 64 || (r > 0 && r <= 64);
    const TileCfg wide[2] = {T128x128, T128x64}, nar[2] = {T256x64, T128x64};
    for (TileCfg c : (narrow ? nar : wide))                                         // a)
        if (cfg_blocks(c, M, N) >= 2 * kNumCU) { cfg = c; splits = 1; kg = 1; return; }
    const long long t64 = cfg_blocks(T64x64, M, N), t128 = cfg_blocks(T128x128, M, N);
    if (!narrow && t64 < 3 * kNumCU) {                                              // b)
        for (int sp = 2; sp <= 4; sp *= 2)
            if (nq >= 12 * sp && (t128 * sp >= 2 * kNumCU || (nq >= 256 && t128 * sp >= kNumCU))) { cfg = T128x128; splits = sp; kg = 1; return; }
    }
    if (t64 * 2 > 3 * kNumCU && t64 < 3 * kNumCU && nq >= 24) {                     // c)
        cfg = (N % 128 == 0) ? T64x128 : T128x64; splits = 1; kg = 2;
    }
}
static FwdPlan plan_fwd(int N, int Hin, int Win, int Cin, int Kout, int R, int S, int Ho, int Wo, int npl, bool dgrad = false) {
    FwdPlan p; p.Ho = Ho; p.Wo = Wo; p.M = N * Ho * Wo; p.cchunks = (int)ceil_div(Cin, BK);
    p.cfg = pick_cfg(p.M, Kout);
    int bm, bn; cfg_dims(p.cfg, bm, bn);
    const long long tiles = ceil_div(p.M, bm) * ceil_div(Kout, bn);
    const int nq = R * S * p.cchunks;
    p.kg = pick_kg(tiles, nq, p.cfg, npl);
    p.splits = (p.kg > 1 && env_int("DSRL_FORCE_SPLITS", 0) <= 0) ? 1 : pick_splits(tiles, nq);
    if (npl) pick_split_plan(p.M, Kout, nq, dgrad, p.cfg, p.splits, p.kg);
    // split-K across workgroups with the reduction inside the launch (conv_sk.hip): 128x128 tiles with two K groups, at most kCoopMaxTiles tiles
    // (one arrival ticket per tile in the activation's amax record), partial tiles within one buffer descriptor
    cfg_dims(p.cfg, bm, bn);
    const long long t2 = ceil_div(p.M, bm) * ceil_div(Kout, bn);
    p.coop = p.splits > 1 && (p.kg == 2 || (p.kg == 1 && env_int("DSRL_SK_COOP1", 1))) && p.cfg == T128x128 && conv_precision_mode() >= 4 && t2 <= kCoopMaxTiles &&
             (long long)p.splits * t2 * bm * bn * 4 < (1ll << 31) && env_int("DSRL_SK_COOP", 1);
    p.ws = p.splits > 1 ? (p.coop ? (size_t)p.splits * t2 * bm * bn * sizeof(float) : (size_t)p.splits * p.M * Kout * sizeof(float)) : 0;
    return p;
}
// the data gradient's plan: a strided one cannot reduce its split-K inside the launch (parity-ordered rows), so it keeps slabs + reduce
static FwdPlan plan_dgrad(int N, int Ho, int Wo, int Kp, int C, int R, int S, int H, int W, int npl, int stride) {
    FwdPlan p = plan_fwd(N, Ho, Wo, Kp, C, R, S, H, W, npl, true);
    if (stride != 1 && p.coop) { p.coop = false; p.ws = p.splits > 1 ? (size_t)p.splits * p.M * C * sizeof(float) : 0; }
    return p;
}
// workspace queries do not know which arithmetic the launch will run in: the larger of the two plans
static size_t plan_fwd_ws(int N, int Hin, int Win, int Cin, int Kout, int R, int S, int Ho, int Wo) {
    return std::max({plan_fwd(N, Hin, Win, Cin, Kout, R, S, Ho, Wo, 0).ws, plan_fwd(N, Hin, Win, Cin, Kout, R, S, Ho, Wo, 3).ws,
                     plan_fwd(N, Hin, Win, Cin, Kout, R, S, Ho, Wo, 3, true).ws});
}
static size_t with_amax_scratch(size_t ws) { return align_up(ws, 64) + kAmaxScratch; }      // every conv workspace ends with the f16x3 scratch words

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
    return with_amax_scratch(plan_fwd_ws(N, H, W, C, K, R, S, Ho, Wo));
}

// rows blocks of BatchNorm partials a forward launch of this shape writes in the current arithmetic mode (0 = it cannot: fp32 kernels,
// split-K slabs, or more than 256 row blocks)
constexpr int kMaxStatsParts = 4096;
static int fwd_stats_parts(const FwdPlan& p, int npl, bool dgrad = false) {
    if (npl && p.splits > 1 && !p.coop && !dgrad && env_int("DSRL_SPLITK_STATS", 1))      // forward split-K: the slab reduce leaves partials of 64 rows each
        return ((p.ws / ((size_t)p.splits * p.M * sizeof(float))) % 32 == 0 && ceil_div(p.M, 64) <= 256) ? (int)ceil_div(p.M, 64) : 0;
    if (!npl || (p.splits > 1 && !p.coop)) return 0;
    int bm, bn; cfg_dims(p.cfg, bm, bn);
    // forward: ONE partial per block tile (round 5: the wave rows of a tile are merged in the conv epilogue); dgrad sums: one per wave row
    static const int kWGM[kNumCfg] = {2, 4, 4, 2, 2, 2, 4, 4, 2};    // waves along M per block tile, DSRL_CFG_SWITCH order (+ the two 8-wave tiles)
    const long long parts = ceil_div(p.M, bm) * (dgrad ? kWGM[p.cfg] : 1);
    return parts <= kMaxStatsParts ? (int)parts : 0;       // more than 256: the BatchNorm kernels reduce them to 32 first (bn.hip: stats_reduce)
}
extern "C" int dsrl_conv2d_fwd_stats_parts(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil) {
    const int Ho = out_size(H, R, stride, pad, dil), Wo = out_size(W, S, stride, pad, dil);
    if (Ho <= 0 || Wo <= 0 || C % 4) return 0;
    const int npl = conv_planes(PASS_FWD);
    return fwd_stats_parts(plan_fwd(N, H, W, C, K, R, S, Ho, Wo, npl), npl);
}

// can a launch whose reduction runs over C channels of a [P][ld] activation take its operands as fp16 planes?  (DSRL_PLANES=0: never)
static bool planes_usable(int C, int ld, const void* a_planes, const void* b_planes) {
    return C % 8 == 0 && ld % 8 == 0 && ((uintptr_t)a_planes % 16) == 0 && ((uintptr_t)b_planes % 16) == 0 && env_int("DSRL_PLANES", 1) != 0;
}
static int fwd_impl(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                    int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                    void* ws, size_t ws_bytes, dsrl_stream_t stream, float* stats, int stats_parts,
                    const unsigned* x_amax = nullptr, const unsigned* w_amax = nullptr, const void* w_split = nullptr,
                    const void* x_planes = nullptr, const void* w_planes = nullptr) {
    if (int e = check_conv(x, w, y, N, H, W, C, K, R, S, stride, pad, dil)) return e;
    DSRL_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0, DSRL_E_UNSUPPORTED,
                 "conv2d_fwd: C (%d) and ldx (%d) must be multiples of 4 and x,w 16-byte aligned", C, ldx);
    DSRL_REQUIRE(ldx >= C && ldy >= K, DSRL_E_BADARG, "conv2d_fwd: ld smaller than channel count");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    const int Ho = out_size(H, R, stride, pad, dil), Wo = out_size(W, S, stride, pad, dil);
    const FwdPlan p = plan_fwd(N, H, W, C, K, R, S, Ho, Wo, conv_planes(PASS_FWD));
    DSRL_REQUIRE(ws_bytes >= p.ws && (p.ws == 0 || ws), DSRL_E_WORKSPACE, "conv2d_fwd: workspace %zu < %zu", ws_bytes, p.ws);
    ConvArgs a{};
    a.x = x; a.w = w; a.ldx = ldx; a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.Ho = Ho; a.Wo = Wo;
    a.stride = stride; a.pad = pad; a.dil = dil; a.M = p.M; a.cchunks = p.cchunks; a.splits = p.splits; a.kg = p.kg; a.slab = (long long)p.M * K;
    const long long xb = span_bytes((long long)N * H * W, ldx, C), wb = (long long)K * R * S * C * 4, yb = p.splits > 1 ? (long long)p.M * K * 4 : span_bytes(p.M, ldy, K);
    DSRL_REQUIRE_31(xb, "conv2d_fwd(x)"); DSRL_REQUIRE_31(wb, "conv2d_fwd(w)"); DSRL_REQUIRE_31(yb, "conv2d_fwd(y)");
    a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb; a.y_bytes = (unsigned)yb;
    if (conv_f16()) {
        OperandAmax am{x_amax, w_amax};
        if (int e = resolve_amax(am, x, ldx, (long long)N * H * W, C, w, C, (long long)K * R * S, C, ws, ws_bytes, p.ws, st, "conv2d_fwd")) return e;
        a.amax_a = am.a; a.amax_b = am.b;
        if (w_split != nullptr && w_amax != nullptr && env_int("DSRL_PRESPLIT", 1)) {      // the split form carries the scale of ITS record
            DSRL_REQUIRE(((uintptr_t)w_split % 16) == 0, DSRL_E_BADARG, "conv2d_fwd: unaligned pre-split filter");
            a.w = (const float*)w_split; a.w_split = 1;
        }
        // both operands as fp16 planes carrying the scales of THESE records (dsrl_split_planes / dsrl_conv2d_filter_planes_batched): same tile plan,
        // same summation order, staged by LDS-DMA
        // (round 5) f16x1 takes ONE plane per operand - the operand of a half-precision STORAGE format: fp16 activations at their tensor's scale
        // A plan that reduces its split-K inside the launch (p.coop) has no planes build: it keeps the register-staged kernel whether or not planes are
        // on offer, so that the launch, its BatchNorm partials (dsrl_conv2d_fwd_stats_parts) and the bits of its result do not depend on which step first
        // had the planes (round 5: with the planes taking precedence the step-1 and step-2 plans differed, and the slab reduce wrote ceil(M/64) rows of
        // partials into a buffer sized for the cooperative launch's M/128).
        if (x_planes != nullptr && w_planes != nullptr && x_amax != nullptr && w_amax != nullptr && planes_usable(C, ldx, x_planes, w_planes) &&
            planes_cfg_supported((int)p.cfg, p.kg) && !(p.coop && p.splits > 1)) {
            const long long pe = (long long)N * H * W * ldx, we = (long long)K * R * S * C;
            a.planes = conv_planes(PASS_FWD); a.w_split = 0;
            a.x = (const float*)x_planes; a.w = (const float*)w_planes;
            a.x_bytes = (unsigned)(span_bytes((long long)N * H * W, ldx, C) / 2); a.w_bytes = (unsigned)(we * 2);
            a.a_lo = (unsigned)planes_lo_offset(pe); a.b_lo = (unsigned)planes_lo_offset(we);
        }
    }
    ProfScope prof(prof_family(PASS_FWD), 2.0 * (double)dsrl_conv2d_inbounds_macs(N, H, W, C, K, R, S, stride, pad, dil), 4.0 * ((double)N * H * W * C + (double)K * R * S * C + (double)N * out_size(H, R, stride, pad, dil) * out_size(W, S, stride, pad, dil) * K), st);
    prof.shape("fwd", N, H, W, C, K, R, stride, pad, dil);
    if (p.splits > 1 && p.coop && a.amax_a != nullptr && !a.planes) {
        // the partial tiles meet inside the launch: y, bias and the BatchNorm partials as in an unsplit launch
        a.coop_slab = (float*)ws; a.tickets = const_cast<unsigned*>(a.amax_a);
        a.y_bytes = (unsigned)span_bytes(p.M, ldy, K);
        if (stats != nullptr) {
            DSRL_REQUIRE(fwd_stats_parts(p, conv_planes(PASS_FWD)) == stats_parts && stats_parts > 0, DSRL_E_BADARG,
                         "conv2d_fwd_stats: this launch writes %d row blocks of partials, the caller expects %d (dsrl_conv2d_fwd_stats_parts)",
                         fwd_stats_parts(p, conv_planes(PASS_FWD)), stats_parts);
            a.stats = stats;
        }
        a.y = y; a.ldy = ldy; a.bias = bias;
        return launch_igemm<false>(a, p.cfg, st);
    }
    if (p.splits > 1) {
        DSRL_REQUIRE(!(p.coop && stats != nullptr), DSRL_E_UNSUPPORTED, "conv2d_fwd_stats: the cooperative split-K plan of this shape could not be launched (no operand magnitudes)");
        a.y = (float*)ws; a.ldy = K; a.bias = nullptr;
        if (int e = launch_igemm<false>(a, p.cfg, st)) return e;
        const long long total = (long long)p.M * K;
        if (stats != nullptr) {
            DSRL_REQUIRE(fwd_stats_parts(p, conv_planes(PASS_FWD)) == stats_parts && stats_parts > 0 && ldy % 4 == 0 && ((uintptr_t)y % 16) == 0 &&
                         (bias == nullptr || ((uintptr_t)bias % 16) == 0), DSRL_E_BADARG,
                         "conv2d_fwd_stats: this split-K launch writes %d row blocks of partials (ldy a multiple of 4, y 16-byte aligned), the caller expects %d",
                         fwd_stats_parts(p, conv_planes(PASS_FWD)), stats_parts);
            hipLaunchKernelGGL(splitk_reduce_stats_kernel, dim3((unsigned)(stats_parts * (K / 32))), dim3(256), 0, st,
                               (const float*)ws, p.splits, a.slab, p.M, K, bias, y, ldy, stats);
            return launch_status("splitk_reduce_stats_kernel");
        }
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 4096)), dim3(256), 0, st,
                           (const float*)ws, p.splits, a.slab, p.M, K, bias, y, ldy);
        return launch_status("splitk_reduce_kernel");
    }
    if (stats != nullptr) {
        DSRL_REQUIRE(fwd_stats_parts(p, conv_planes(PASS_FWD)) == stats_parts && stats_parts > 0, DSRL_E_BADARG,
                     "conv2d_fwd_stats: this launch writes %d row blocks of partials, the caller expects %d (dsrl_conv2d_fwd_stats_parts)",
                     fwd_stats_parts(p, conv_planes(PASS_FWD)), stats_parts);
        a.stats = stats;
    }
    a.y = y; a.ldy = ldy; a.bias = bias;
    return launch_igemm<false>(a, p.cfg, st);
}

extern "C" int dsrl_conv2d_fwd(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                               int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                               void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    return fwd_impl(x, ldx, w, bias, y, ldy, N, H, W, C, K, R, S, stride, pad, dil, ws, ws_bytes, stream, nullptr, 0);
}
extern "C" int dsrl_conv2d_fwd_stats(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                                     int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                     void* ws, size_t ws_bytes, float* stats, int stats_parts, dsrl_stream_t stream) {
    DSRL_REQUIRE(stats != nullptr, DSRL_E_BADARG, "conv2d_fwd_stats: null statistics buffer");
    return fwd_impl(x, ldx, w, bias, y, ldy, N, H, W, C, K, R, S, stride, pad, dil, ws, ws_bytes, stream, stats, stats_parts);
}

// dgrad = the same implicit GEMM with dy as the input tensor, the transposed filter wt[c][tap][k] and the
// gather pixel(h,w,tap) = ((h + pad - r*dil)/stride, (w + pad - s*dil)/stride) when divisible.
static int pad4(int v) { return (v + 3) & ~3; }
static size_t dgrad_wt_bytes(int C, int K, int R, int S) { return align_up((size_t)C * pad4(K) * R * S * sizeof(float), 256); }

extern "C" size_t dsrl_conv2d_dgrad_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil) {
    const int Ho = out_size(H, R, stride, pad, dil), Wo = out_size(W, S, stride, pad, dil);
    if (Ho <= 0 || Wo <= 0) return 0;
    return with_amax_scratch(dgrad_wt_bytes(C, K, R, S) + plan_fwd_ws(N, Ho, Wo, pad4(K), C, R, S, H, W));
}

extern "C" size_t dsrl_conv2d_transposed_filter_floats(int C, int K, int R, int S) { return (size_t)C * R * S * pad4(K); }
extern "C" int dsrl_conv2d_transpose_filter(const float* w, float* wt, int C, int K, int R, int S, dsrl_stream_t stream) {
    DSRL_REQUIRE(w && wt && C > 0 && K > 0 && R > 0 && S > 0, DSRL_E_BADARG, "conv2d_transpose_filter: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    const int Kp = pad4(K);
    hipLaunchKernelGGL(weight_transpose_kernel, dim3((unsigned)ceil_div(C, 32), (unsigned)ceil_div(Kp, 32), (unsigned)(R * S)), dim3(256), 0, st, w, wt, K, Kp, R * S, C);
    return launch_status("weight_transpose_kernel");
}

extern "C" int dsrl_conv2d_split_filters_batched(const int64_t* table, int n, int64_t total_tiles, dsrl_stream_t stream) {
    DSRL_REQUIRE(table && n > 0 && total_tiles > 0 && total_tiles < (1ll << 31), DSRL_E_BADARG, "conv2d_split_filters_batched: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    hipLaunchKernelGGL(weight_split_batched_kernel, dim3((unsigned)ceil_div(total_tiles, (int64_t)kWtTilesPerBlock)), dim3(256), 0, st, (const long long*)table, n,
                       (long long)total_tiles);
    return launch_status("weight_split_batched_kernel");
}

extern "C" int dsrl_conv2d_transpose_filters_batched(const int64_t* table, int n, int64_t total_tiles, dsrl_stream_t stream) {
    DSRL_REQUIRE(table && n > 0 && total_tiles > 0 && total_tiles < (1ll << 31), DSRL_E_BADARG, "conv2d_transpose_filters_batched: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    hipLaunchKernelGGL(weight_transpose_batched_kernel, dim3((unsigned)ceil_div(total_tiles, (int64_t)kWtTilesPerBlock)), dim3(256), 0, st, (const long long*)table, n,
                       (long long)total_tiles);
    return launch_status("weight_transpose_batched_kernel");
}

extern "C" int dsrl_conv2d_filters_amax_segment_floats(void) { return kAmaxSegFloats; }
extern "C" int dsrl_conv2d_filters_amax_batched(const int64_t* table, int64_t nseg, dsrl_stream_t stream) {
    DSRL_REQUIRE(table && nseg > 0 && nseg < (1ll << 31), DSRL_E_BADARG, "conv2d_filters_amax_batched: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    hipLaunchKernelGGL(weight_amax_batched_kernel, dim3((unsigned)nseg), dim3(256), 0, st, (const long long*)table);
    return launch_status("weight_amax_batched_kernel");
}

struct DgradBn { const float* x; const float* y; const float* mean; const float* invstd; float* stats; int ldx, ldy, relu; float gscale = 1.f; };
static int dgrad_impl(const float* dy, int lddy, const float* w, const float* wt_in, float* dx, int lddx,
                      int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                      void* ws, size_t ws_bytes, dsrl_stream_t stream, int accumulate, const DgradBn* bn = nullptr,
                      const unsigned* dy_amax = nullptr, const unsigned* w_amax = nullptr, const void* wt_split = nullptr,
                      const void* dy_planes = nullptr, const void* wt_planes = nullptr) {
    if (int e = check_conv(dy, w, dx, N, H, W, C, K, R, S, stride, pad, dil)) return e;
    const int Kp = pad4(K);     // K % 4 != 0 (cls_conv, 19 classes): dy must be padded to lddy >= Kp with finite pad values
    DSRL_REQUIRE(lddy % 4 == 0 && ((uintptr_t)dy % 16) == 0, DSRL_E_UNSUPPORTED,
                 "conv2d_dgrad: lddy (%d) must be a multiple of 4 and dy 16-byte aligned", lddy);
    DSRL_REQUIRE(lddy >= Kp && lddx >= C, DSRL_E_BADARG, "conv2d_dgrad: ld smaller than (padded) channel count");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    const int Ho = out_size(H, R, stride, pad, dil), Wo = out_size(W, S, stride, pad, dil);
    const FwdPlan p = plan_dgrad(N, Ho, Wo, Kp, C, R, S, H, W, conv_planes(PASS_DGRAD), stride);
    const size_t wtb = dgrad_wt_bytes(C, K, R, S);
    DSRL_REQUIRE(ws && ws_bytes >= wtb + p.ws, DSRL_E_WORKSPACE, "conv2d_dgrad: workspace %zu < %zu", ws_bytes, wtb + p.ws);
    const float* wt = wt_in;
    float* slabs = (float*)((char*)ws + wtb);
    const bool use_planes = conv_f16() && dy_planes != nullptr && wt_planes != nullptr && dy_amax != nullptr && w_amax != nullptr && stride == 1 && K % 8 == 0 &&
                            planes_usable(K, lddy, dy_planes, wt_planes) && planes_cfg_supported((int)p.cfg, p.kg) && !(p.coop && p.splits > 1);    // see fwd_impl
    const bool use_split = conv_f16() && wt_split != nullptr && w_amax != nullptr && env_int("DSRL_PRESPLIT", 1);
    if (wt == nullptr && use_planes) wt = (const float*)wt_planes;       // replaced below; no fp32 transpose is built for it
    if (wt == nullptr && use_split) wt = (const float*)wt_split;         // replaced below; no fp32 transpose is built for it
    if (wt == nullptr) {
        hipLaunchKernelGGL(weight_transpose_kernel, dim3((unsigned)ceil_div(C, 32), (unsigned)ceil_div(Kp, 32), (unsigned)(R * S)), dim3(256), 0, st,
                           w, (float*)ws, K, Kp, R * S, C);
        if (int e = launch_status("weight_transpose_kernel")) return e;
        wt = (const float*)ws;
    }
    ConvArgs a{};
    a.x = dy; a.w = wt; a.ldx = lddy; a.N = N; a.H = Ho; a.W = Wo; a.C = Kp; a.K = C; a.R = R; a.S = S; a.Ho = H; a.Wo = W;
    a.stride = stride; a.pad = pad; a.dil = dil; a.M = p.M; a.cchunks = p.cchunks; a.splits = p.splits; a.kg = p.kg; a.slab = (long long)p.M * C;
    {
        const long long xb = span_bytes((long long)N * Ho * Wo, lddy, Kp), wb = (long long)C * R * S * Kp * 4, yb = p.splits > 1 ? (long long)p.M * C * 4 : span_bytes(p.M, lddx, C);
        DSRL_REQUIRE_31(xb, "conv2d_dgrad(dy)"); DSRL_REQUIRE_31(wb, "conv2d_dgrad(w)"); DSRL_REQUIRE_31(yb, "conv2d_dgrad(dx)");
        a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb; a.y_bytes = (unsigned)yb;
    }
    if (conv_f16()) {
        OperandAmax am{dy_amax, w_amax};
        if (int e = resolve_amax(am, dy, lddy, (long long)N * Ho * Wo, Kp, w, C, (long long)K * R * S, C, ws, ws_bytes, wtb + p.ws, st, "conv2d_dgrad")) return e;
        a.amax_a = am.a; a.amax_b = am.b;
        if (wt_split != nullptr && w_amax != nullptr && env_int("DSRL_PRESPLIT", 1)) {
            DSRL_REQUIRE(((uintptr_t)wt_split % 16) == 0, DSRL_E_BADARG, "conv2d_dgrad: unaligned pre-split filter");
            a.w = (const float*)wt_split; a.w_split = 1;
        }
        if (use_planes) {
            const long long pe = (long long)N * Ho * Wo * lddy, we = (long long)C * R * S * K;
            a.planes = conv_planes(PASS_DGRAD); a.w_split = 0;
            a.x = (const float*)dy_planes; a.w = (const float*)wt_planes;
            a.x_bytes = (unsigned)(span_bytes((long long)N * Ho * Wo, lddy, K) / 2); a.w_bytes = (unsigned)(we * 2);
            a.a_lo = (unsigned)planes_lo_offset(pe); a.b_lo = (unsigned)planes_lo_offset(we);
        }
    }
    ProfScope prof(prof_family(PASS_DGRAD), 2.0 * (double)dsrl_conv2d_inbounds_macs(N, H, W, C, K, R, S, stride, pad, dil), 4.0 * ((double)N * H * W * C + (double)K * R * S * C + (double)N * out_size(H, R, stride, pad, dil) * out_size(W, S, stride, pad, dil) * K), st);
    prof.shape("dgrad", N, H, W, C, K, R, stride, pad, dil);
    if (p.splits > 1 && !(p.coop && a.amax_a != nullptr && !a.planes && stride == 1)) {
        DSRL_REQUIRE(bn == nullptr, DSRL_E_UNSUPPORTED, "conv2d_dgrad: BatchNorm sums asked from a split-K launch that reduces through slabs");
        a.y = slabs; a.ldy = C; a.bias = nullptr;
        if (int e = launch_igemm<true>(a, p.cfg, st)) return e;
        const long long total = (long long)p.M * C;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 4096)), dim3(256), 0, st,
                           (const float*)slabs, p.splits, a.slab, p.M, C, (const float*)nullptr, dx, lddx, accumulate);
        return launch_status("splitk_reduce_kernel");
    }
    if (p.splits > 1) {             // cooperative split-K: dx, accumulate and the BatchNorm-backward sums as in an unsplit launch
        a.coop_slab = slabs; a.tickets = const_cast<unsigned*>(a.amax_a);
        a.y_bytes = (unsigned)span_bytes(p.M, lddx, C);
    }
    a.y = dx; a.ldy = lddx; a.bias = nullptr; a.accumulate = accumulate;
    if (stride > 1 && conv_planes(PASS_DGRAD) && H % stride == 0 && W % stride == 0 && (W / stride) % 32 == 0 && env_int("DSRL_DGRAD_PARITY", 1)) {
        // rows ordered by parity class (ConvArgs::par): a tile then only runs the taps that divide evenly for its class
        int bm, bn_; cfg_dims(p.cfg, bm, bn_);
        const long long Mc = (long long)N * (H / stride) * (W / stride);
        if (Mc % bm == 0) { a.par = stride; a.Hh = H / stride; a.Wh = W / stride; a.pbm = bm; }
    }
    if (bn) {
        a.bn_x = bn->x; a.bn_y = bn->y; a.bn_mean = bn->mean; a.bn_invstd = bn->invstd; a.bstats = bn->stats; a.bn_ldx = bn->ldx; a.bn_ldy = bn->ldy; a.bn_relu = bn->relu;
        a.bn_gscale = bn->gscale;
        a.bn_fast = a.par == 0 && C % 4 == 0 && bn->ldx % 4 == 0 && (!bn->relu || bn->ldy % 4 == 0) && ((uintptr_t)bn->x % 16) == 0 && (!bn->relu || ((uintptr_t)bn->y % 16) == 0) &&
                    ((uintptr_t)bn->mean % 16) == 0 && ((uintptr_t)bn->invstd % 16) == 0 && ((uintptr_t)bn->stats % 16) == 0 && env_int("DSRL_BNSTATS_FAST", 1);
    }
    return launch_igemm<true>(a, p.cfg, st);
}

extern "C" int dsrl_conv2d_dgrad(const float* dy, int lddy, const float* w, const float* wt_in, float* dx, int lddx,
                                 int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                 void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    return dgrad_impl(dy, lddy, w, wt_in, dx, lddx, N, H, W, C, K, R, S, stride, pad, dil, ws, ws_bytes, stream, 0);
}
// row blocks of BatchNorm-backward partials a dgrad launch of this shape writes in the current arithmetic mode (0 = it cannot)
extern "C" int dsrl_conv2d_dgrad_stats_parts(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil) {
    const int Ho = out_size(H, R, stride, pad, dil), Wo = out_size(W, S, stride, pad, dil);
    if (Ho <= 0 || Wo <= 0) return 0;
    const int npl = conv_planes(PASS_DGRAD);
    return fwd_stats_parts(plan_dgrad(N, Ho, Wo, pad4(K), C, R, S, H, W, npl, stride), npl, true);
}
extern "C" int dsrl_conv2d_dgrad_bnstats(const float* dy, int lddy, const float* w, const float* wt_in, float* dx, int lddx,
                                         int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                         void* ws, size_t ws_bytes, const float* bn_x, int bn_ldx, const float* bn_y, int bn_ldy,
                                         const float* bn_mean, const float* bn_invstd, int bn_relu, float* bstats, int stats_parts, int accumulate,
                                         dsrl_stream_t stream) {
    DSRL_REQUIRE(bn_x && bn_mean && bn_invstd && bstats && (bn_y || !bn_relu) && bn_ldx >= C && (!bn_relu || bn_ldy >= C), DSRL_E_BADARG, "conv2d_dgrad_bnstats: bad BatchNorm arguments");
    DSRL_REQUIRE(dsrl_conv2d_dgrad_stats_parts(N, H, W, C, K, R, S, stride, pad, dil) == stats_parts && stats_parts > 0, DSRL_E_BADARG,
                 "conv2d_dgrad_bnstats: this launch writes %d row blocks of partials, the caller expects %d",
                 dsrl_conv2d_dgrad_stats_parts(N, H, W, C, K, R, S, stride, pad, dil), stats_parts);
    DgradBn bn{bn_x, bn_y, bn_mean, bn_invstd, bstats, bn_ldx, bn_ldy, bn_relu};
    return dgrad_impl(dy, lddy, w, wt_in, dx, lddx, N, H, W, C, K, R, S, stride, pad, dil, ws, ws_bytes, stream, accumulate ? 1 : 0, &bn);
}
extern "C" int dsrl_conv2d_dgrad_accumulate(const float* dy, int lddy, const float* w, const float* wt_in, float* dx, int lddx,
                                            int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                            void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    return dgrad_impl(dy, lddy, w, wt_in, dx, lddx, N, H, W, C, K, R, S, stride, pad, dil, ws, ws_bytes, stream, 1);
}

static void launch_wgrad(const WgradArgs& a_in, TileCfg cfg, int bm, int bn, dim3 grid, hipStream_t st) {
    const int npl = conv_planes(PASS_WGRAD);
    const bool f16 = conv_f16();
    if (npl) {
        WgradArgs a = a_in;
        make_magic(a.Ho * a.Wo, a.mHW, a.sHW);
        make_magic(a.Wo, a.mW, a.sW);
        const size_t stages = (size_t)2 * npl * 16 * ((bm * 2 + 64) + (bn * 2 + 64));     // two stages: <= 60 KiB for every tile
        const int kg = a.kg > 1 ? a.kg : 1;
        if (kg > 1) {
            const size_t lds = std::max(stages * kg, (size_t)kg * (bm / 32) * (bn / 32) / 4 * 16384 + 8192);      // every group's accumulators + the BatchNorm-sum exchange
#define DSRL_LAUNCH_WKG(a_, b_, c_, d_, NPL_, KG_, F16_)                                                                            \
            {                                                                                                                        \
                static const hipError_t attr = hipFuncSetAttribute((const void*)conv_wgrad_split_kernel<a_, b_, c_, d_, NPL_, KG_, F16_>,  \
                                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);          \
                (void)attr;                                                                                                          \
                hipLaunchKernelGGL((conv_wgrad_split_kernel<a_, b_, c_, d_, NPL_, KG_, F16_>), grid, dim3(256 * KG_), lds, st, a);   \
            }
#define DSRL_WKG_BY_NPL(a_, b_, c_, d_, KG_) { if (f16 && npl == 1) DSRL_LAUNCH_WKG(a_, b_, c_, d_, 1, KG_, true) else if (f16) DSRL_LAUNCH_WKG(a_, b_, c_, d_, 2, KG_, true) else if (npl == 2) DSRL_LAUNCH_WKG(a_, b_, c_, d_, 2, KG_, false) else DSRL_LAUNCH_WKG(a_, b_, c_, d_, 3, KG_, false) }
            // two pixel groups only: the four-group builds always lost (tools/wgrad_kg.py) and spilled 8-12 registers; removed in round 5
            if (cfg == T64x64) DSRL_WKG_BY_NPL(1, 1, 2, 2, 2)
            else if (cfg == T128x64) DSRL_WKG_BY_NPL(2, 1, 2, 2, 2)
            else if (cfg == T64x128) DSRL_WKG_BY_NPL(1, 2, 2, 2, 2)
            else DSRL_WKG_BY_NPL(2, 2, 2, 2, 2)
#undef DSRL_WKG_BY_NPL
#undef DSRL_LAUNCH_WKG
            return;
        }
        const size_t lds = stages;
#define DSRL_LAUNCH_WGRAD(a_, b_, c_, d_)                                                                           \
        if (f16 && npl == 1) hipLaunchKernelGGL((conv_wgrad_split_kernel<a_, b_, c_, d_, 1, 1, true>), grid, dim3(256), lds, st, a); \
        else if (f16) hipLaunchKernelGGL((conv_wgrad_split_kernel<a_, b_, c_, d_, 2, 1, true>), grid, dim3(256), lds, st, a); \
        else if (npl == 2) hipLaunchKernelGGL((conv_wgrad_split_kernel<a_, b_, c_, d_, 2>), grid, dim3(256), lds, st, a); \
        else hipLaunchKernelGGL((conv_wgrad_split_kernel<a_, b_, c_, d_, 3>), grid, dim3(256), lds, st, a);
        DSRL_CFG_SWITCH(cfg, DSRL_LAUNCH_WGRAD)
#undef DSRL_LAUNCH_WGRAD
        return;
    }
    const size_t lds = (size_t)32 * (bm + bn) * sizeof(float);
#define DSRL_LAUNCH_WGRAD(a_, b_, c_, d_) hipLaunchKernelGGL((conv_wgrad_f32_kernel<a_, b_, c_, d_>), grid, dim3(256), lds, st, a_in)
    DSRL_CFG_SWITCH(cfg, DSRL_LAUNCH_WGRAD)
#undef DSRL_LAUNCH_WGRAD
}

namespace dsrl {
struct WgPlan { int Ho, Wo; long long P; TileCfg cfg; int bm, bn, ktiles, ctiles, psplits; TapList tl; size_t ws; bool w3; };
// pixel ranges of an all-taps 3x3 launch (conv_wgrad3.hip; one 8-wave block per CU): as many as fit two rounds of blocks, at least 8 chunks each
static int w3_psplits(long long tiles, long long chunks) {
    const int forced = env_int("DSRL_FORCE_PSPLITS", 0);
    if (forced > 0) return (int)std::max<long long>(1, std::min<long long>(forced, chunks));
    long long sp = std::max<long long>(1, env_int("DSRL_WGRAD3_TARGET_BLOCKS", 2 * kNumCU) / std::max<long long>(tiles, 1));     // rounded down: the blocks fit two rounds of one per CU
    sp = std::min(sp, std::max<long long>(1, chunks / 8));
    return (int)std::max<long long>(1, std::min<long long>(sp, 128));
}
static WgPlan plan_wgrad(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil) {
    WgPlan p; p.Ho = out_size(H, R, stride, pad, dil); p.Wo = out_size(W, S, stride, pad, dil);
    p.P = (long long)N * p.Ho * p.Wo;
    // GEMM rows = out channels, columns = in channels
    p.cfg = pick_cfg_wgrad(K, C);
    int bm, bn; cfg_dims(p.cfg, bm, bn); p.bm = bm; p.bn = bn;
    p.ktiles = (int)ceil_div(K, bm); p.ctiles = (int)ceil_div(C, bn);
    p.tl.n = 0;
    for (int r = 0; r < R; ++r)
        for (int s = 0; s < S; ++s)
            if (valid_count(H, p.Ho, stride, pad, r * dil) > 0 && valid_count(W, p.Wo, stride, pad, s * dil) > 0) p.tl.taps[p.tl.n++] = r * S + s;
    const long long chunks = ceil_div(p.P, 32);
    p.w3 = p.tl.n == 9 && wgrad3_eligible(N, H, W, C, K, R, S, stride, pad, dil, conv_planes(PASS_WGRAD), conv_f16());
    if (p.w3) {         // all nine taps in one block (conv_wgrad3.hip)
        wgrad3_tile(p.bm, p.bn);
        p.ktiles = (int)ceil_div(K, p.bm); p.ctiles = (int)ceil_div(C, p.bn);
        p.psplits = w3_psplits((long long)p.ktiles * p.ctiles, chunks);
        p.ws = p.psplits > 1 ? (size_t)p.psplits * K * R * S * C * sizeof(float) : 0;
        return p;
    }
    const long long tiles = (long long)p.ktiles * p.ctiles * std::max(1, p.tl.n);
    p.psplits = pick_psplits(tiles, chunks);
    p.ws = p.psplits > 1 ? (size_t)p.psplits * K * R * S * C * sizeof(float) : 0;
    return p;
}
}  // namespace dsrl

extern "C" size_t dsrl_conv2d_wgrad_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil) {
    if (out_size(H, R, stride, pad, dil) <= 0 || out_size(W, S, stride, pad, dil) <= 0) return 0;
    return with_amax_scratch(plan_wgrad(N, H, W, C, K, R, S, stride, pad, dil).ws);
}

static int wgrad_impl(const float* x, int ldx, const float* dy, int lddy, float* dw,
                      int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                      void* ws, size_t ws_bytes, dsrl_stream_t stream, const unsigned* x_amax, const unsigned* dy_amax) {
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
        if (int e = launch_zero_fill(dw, (size_t)K * RS * C * sizeof(float), st)) return e;
    }
    if (p.tl.n == 0) return DSRL_OK;
    WgradArgs a{};
    a.x = x; a.dy = dy; a.ldx = ldx; a.lddy = lddy; a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.Ho = p.Ho; a.Wo = p.Wo;
    a.stride = stride; a.pad = pad; a.dil = dil; a.P = p.P; a.ctiles = p.ctiles; a.psplits = p.psplits; a.slab = (long long)K * RS * C;
    {
        const long long xb = span_bytes((long long)N * H * W, ldx, C), db = span_bytes(p.P, lddy, pad4(K));
        DSRL_REQUIRE_31(xb, "conv2d_wgrad(x)"); DSRL_REQUIRE_31(db, "conv2d_wgrad(dy)");
        a.x_bytes = (unsigned)xb; a.dy_bytes = (unsigned)db; a.no_ident = wgrad_no_ident();
    }
    a.ntaps = p.tl.n;
    for (int i = 0; i < p.tl.n; ++i) a.taps[i] = p.tl.taps[i];
    if (conv_f16()) {
        OperandAmax am{dy_amax, x_amax};
        if (int e = resolve_amax(am, dy, lddy, p.P, pad4(K), x, ldx, (long long)N * H * W, C, ws, ws_bytes, p.ws, st, "conv2d_wgrad")) return e;
        a.amax_dy = am.a; a.amax_x = am.b;
    }
    if (p.w3) {
        a.psplits = p.psplits; a.kg = 1;
        a.dw = p.psplits > 1 ? (float*)ws : dw;
        a.kctiles = p.ktiles * p.ctiles; a.xcd_remap = env_int("DSRL_XCD_REMAP", 1);
        a.nblocks = a.kctiles * a.psplits;
        make_magic(a.Ho * a.Wo, a.mHW, a.sHW);
        make_magic(a.Wo, a.mW, a.sW);
        ProfScope prof(prof_family(PASS_WGRAD), 2.0 * (double)dsrl_conv2d_inbounds_macs(N, H, W, C, K, R, S, stride, pad, dil), 4.0 * ((double)N * H * W * C + (double)K * R * S * C + (double)p.P * K), st);
        prof.shape("wgrad", N, H, W, C, K, R, stride, pad, dil);
        if (int e = launch_wgrad3(a, conv_planes(PASS_WGRAD), st)) return e;
        if (p.psplits > 1) {
            const long long total = (long long)K * p.tl.n * (C / 4);
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 4096)), dim3(256), 0, st,
                               (const float*)ws, p.psplits, a.slab, dw, K, RS, C, p.tl);
            return launch_status("wgrad_reduce_kernel");
        }
        return DSRL_OK;
    }
    // pixel groups (split kernels): two groups of 4 waves share a block and half of the planned slabs remain.  Measured
    // (tools/wgrad_kg.py): 1x1 convs gain 5-20 %, 3x3 convs lose (each of their taps already has its own blocks), four groups always lose.
    int psplits = p.psplits, kg = 1;
    if (conv_planes(PASS_WGRAD) && env_int("DSRL_WGRAD_KG", 1)) {
        const int npl = conv_planes(PASS_WGRAD);
        const int maxkg = (p.cfg == T64x64 || p.cfg == T128x64 || p.cfg == T64x128 || p.cfg == T128x128) ? 2 : 1;
        const int forced = env_int("DSRL_FORCE_WGRAD_KG", 0);
        kg = forced > 0 ? std::min(forced, maxkg) : ((RS == 1 && psplits >= 4) ? std::min(2, maxkg) : 1);
        if (kg == 3) kg = 2;
        while (kg > 1 && (size_t)kg * 2 * npl * 16 * ((p.bm * 2 + 64) + (p.bn * 2 + 64)) > 144 * 1024) kg /= 2;      // LDS: kg stage pairs
        psplits = (int)ceil_div(psplits, kg);
    }
    a.psplits = psplits; a.kg = kg;
    a.dw = psplits > 1 ? (float*)ws : dw;
    a.kctiles = p.ktiles * p.ctiles; a.xcd_remap = env_int("DSRL_XCD_REMAP", 1);
    dim3 grid((unsigned)(a.kctiles * p.tl.n * psplits));
    ProfScope prof(prof_family(PASS_WGRAD), 2.0 * (double)dsrl_conv2d_inbounds_macs(N, H, W, C, K, R, S, stride, pad, dil), 4.0 * ((double)N * H * W * C + (double)K * R * S * C + (double)N * out_size(H, R, stride, pad, dil) * out_size(W, S, stride, pad, dil) * K), st);
    prof.shape("wgrad", N, H, W, C, K, R, stride, pad, dil);
    launch_wgrad(a, p.cfg, p.bm, p.bn, grid, st);
    if (int e = launch_status("conv_wgrad kernel")) return e;
    if (psplits > 1) {
        const long long total = (long long)K * p.tl.n * (C / 4);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 4096)), dim3(256), 0, st,
                           (const float*)ws, psplits, a.slab, dw, K, RS, C, p.tl);
        return launch_status("wgrad_reduce_kernel");
    }
    return DSRL_OK;
}
extern "C" int dsrl_conv2d_wgrad(const float* x, int ldx, const float* dy, int lddy, float* dw,
                                 int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                 void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    return wgrad_impl(x, ldx, dy, lddy, dw, N, H, W, C, K, R, S, stride, pad, dil, ws, ws_bytes, stream, nullptr, nullptr);
}

// ---- the same three passes with the operand magnitudes of the f16x3 arithmetic handed in by the caller (device words holding max |.| as a
//      bit pattern, e.g. left by the tensor's producer: dsrl_bn_*(..., y_amax), dsrl_conv2d_transpose_filters_batched; dsrl_amax measures one);
//      a null word is measured by the call itself, the other arithmetics ignore them
extern "C" int dsrl_amax(const float* x, int ld, int64_t P, int C, uint32_t* amax, dsrl_stream_t stream) {
    DSRL_REQUIRE(x && amax && ld >= C && C > 0 && P > 0, DSRL_E_BADARG, "amax: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    return launch_amax(x, ld, P, C, amax, st);
}
extern "C" int dsrl_conv2d_fwd_amax(const float* x, int ldx, const uint32_t* x_amax, const float* w, const uint32_t* w_amax, const void* w_split, const float* bias,
                                    float* y, int ldy, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                    void* ws, size_t ws_bytes, float* stats, int stats_parts, dsrl_stream_t stream) {
    return fwd_impl(x, ldx, w, bias, y, ldy, N, H, W, C, K, R, S, stride, pad, dil, ws, ws_bytes, stream, stats, stats_parts, x_amax, w_amax, w_split);
}
extern "C" int dsrl_conv2d_dgrad_amax(const float* dy, int lddy, const uint32_t* dy_amax, const float* w, const float* wt_in, const uint32_t* w_amax, const void* wt_split,
                                      float* dx, int lddx,
                                      int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                      void* ws, size_t ws_bytes, const float* bn_x, int bn_ldx, const float* bn_y, int bn_ldy,
                                      const float* bn_mean, const float* bn_invstd, int bn_relu, float* bstats, int stats_parts, int accumulate,
                                      dsrl_stream_t stream) {
    if (bstats == nullptr)
        return dgrad_impl(dy, lddy, w, wt_in, dx, lddx, N, H, W, C, K, R, S, stride, pad, dil, ws, ws_bytes, stream, accumulate ? 1 : 0, nullptr, dy_amax, w_amax, wt_split);
    DSRL_REQUIRE(bn_x && bn_mean && bn_invstd && (bn_y || !bn_relu) && bn_ldx >= C && (!bn_relu || bn_ldy >= C), DSRL_E_BADARG, "conv2d_dgrad_amax: bad BatchNorm arguments");
    DSRL_REQUIRE(dsrl_conv2d_dgrad_stats_parts(N, H, W, C, K, R, S, stride, pad, dil) == stats_parts && stats_parts > 0, DSRL_E_BADARG,
                 "conv2d_dgrad_amax: this launch writes %d row blocks of partials, the caller expects %d",
                 dsrl_conv2d_dgrad_stats_parts(N, H, W, C, K, R, S, stride, pad, dil), stats_parts);
    DgradBn bn{bn_x, bn_y, bn_mean, bn_invstd, bstats, bn_ldx, bn_ldy, bn_relu};
    return dgrad_impl(dy, lddy, w, wt_in, dx, lddx, N, H, W, C, K, R, S, stride, pad, dil, ws, ws_bytes, stream, accumulate ? 1 : 0, &bn, dy_amax, w_amax, wt_split);
}
// dsrl_conv2d_fwd_amax / _dgrad_amax with the operands ALSO available as fp16 planes (conv_planes.hip): x_planes / dy_planes from dsrl_split_planes (or the
// tensor's producer), w_planes / wt_planes from dsrl_conv2d_filter_planes_batched, each scaled by the record passed beside it.  A launch that cannot take
// planes (null pointers, channel counts that are no multiples of 8, strided data gradients, a tile plan without a planes build) runs as the _amax call.
extern "C" int dsrl_conv2d_fwd_planes(const float* x, int ldx, const uint32_t* x_amax, const void* x_planes, const float* w, const uint32_t* w_amax, const void* w_split,
                                      const void* w_planes, const float* bias, float* y, int ldy, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                      void* ws, size_t ws_bytes, float* stats, int stats_parts, dsrl_stream_t stream) {
    return fwd_impl(x, ldx, w, bias, y, ldy, N, H, W, C, K, R, S, stride, pad, dil, ws, ws_bytes, stream, stats, stats_parts, x_amax, w_amax, w_split, x_planes, w_planes);
}
extern "C" int dsrl_conv2d_dgrad_planes(const float* dy, int lddy, const uint32_t* dy_amax, const void* dy_planes, const float* w, const float* wt_in, const uint32_t* w_amax,
                                        const void* wt_split, const void* wt_planes, float* dx, int lddx,
                                        int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                        void* ws, size_t ws_bytes, const float* bn_x, int bn_ldx, const float* bn_y, int bn_ldy,
                                        const float* bn_mean, const float* bn_invstd, int bn_relu, float* bstats, int stats_parts, int accumulate,
                                        dsrl_stream_t stream) {
    return dsrl_conv2d_dgrad_planes_drop(dy, lddy, dy_amax, dy_planes, w, wt_in, w_amax, wt_split, wt_planes, dx, lddx, N, H, W, C, K, R, S, stride, pad, dil, ws, ws_bytes,
                                         bn_x, bn_ldx, bn_y, bn_ldy, bn_mean, bn_invstd, bn_relu, 0.f, bstats, stats_parts, accumulate, stream);
}
extern "C" int dsrl_conv2d_dgrad_planes_drop(const float* dy, int lddy, const uint32_t* dy_amax, const void* dy_planes, const float* w, const float* wt_in, const uint32_t* w_amax,
                                             const void* wt_split, const void* wt_planes, float* dx, int lddx,
                                             int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                             void* ws, size_t ws_bytes, const float* bn_x, int bn_ldx, const float* bn_y, int bn_ldy,
                                             const float* bn_mean, const float* bn_invstd, int bn_relu, float bn_drop_p, float* bstats, int stats_parts, int accumulate,
                                             dsrl_stream_t stream) {
    DSRL_REQUIRE(bn_drop_p >= 0.f && bn_drop_p < 1.f && (bn_drop_p == 0.f || bn_relu), DSRL_E_BADARG, "conv2d_dgrad_planes_drop: dropout p=%f needs the ReLU mask (y > 0)", bn_drop_p);
    if (bstats == nullptr)
        return dgrad_impl(dy, lddy, w, wt_in, dx, lddx, N, H, W, C, K, R, S, stride, pad, dil, ws, ws_bytes, stream, accumulate ? 1 : 0, nullptr, dy_amax, w_amax, wt_split,
                          dy_planes, wt_planes);
    DSRL_REQUIRE(bn_x && bn_mean && bn_invstd && (bn_y || !bn_relu) && bn_ldx >= C && (!bn_relu || bn_ldy >= C), DSRL_E_BADARG, "conv2d_dgrad_planes: bad BatchNorm arguments");
    DSRL_REQUIRE(dsrl_conv2d_dgrad_stats_parts(N, H, W, C, K, R, S, stride, pad, dil) == stats_parts && stats_parts > 0, DSRL_E_BADARG,
                 "conv2d_dgrad_planes: this launch writes %d row blocks of partials, the caller expects %d",
                 dsrl_conv2d_dgrad_stats_parts(N, H, W, C, K, R, S, stride, pad, dil), stats_parts);
    DgradBn bn{bn_x, bn_y, bn_mean, bn_invstd, bstats, bn_ldx, bn_ldy, bn_relu};
    bn.gscale = bn_drop_p > 0.f ? 1.f / (1.f - bn_drop_p) : 1.f;          // the factor the dropout kernels apply (bn.hip: ks)
    return dgrad_impl(dy, lddy, w, wt_in, dx, lddx, N, H, W, C, K, R, S, stride, pad, dil, ws, ws_bytes, stream, accumulate ? 1 : 0, &bn, dy_amax, w_amax, wt_split,
                      dy_planes, wt_planes);
}
extern "C" int dsrl_conv2d_wgrad_amax(const float* x, int ldx, const uint32_t* x_amax, const float* dy, int lddy, const uint32_t* dy_amax, float* dw,
                                      int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                      void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    return wgrad_impl(x, ldx, dy, lddy, dw, N, H, W, C, K, R, S, stride, pad, dil, ws, ws_bytes, stream, x_amax, dy_amax);
}


// ------------------------------------------------------------------------------------------------ grouped wgrad
// The weight gradients of a backward pass are independent of everything else in it (only the optimiser reads them), and most
// layers are too small to fill 256 CUs on their own.  dsrl_conv2d_wgrad_group_* therefore runs ALL of them as a few grids: one
// per tile configuration, blocks ordered longest first, a problem's pixel range split only where a block would otherwise exceed
// ~4096 pixels (small-weight / many-pixel layers: their slabs are tiny).  Compared with one launch per layer: no launch ramp / tail
// per layer, no pixel splits (slab write + reduce launch) just to fill the chip, one slab reduce for the whole pass.
namespace dsrl {
constexpr unsigned kGroupMagic = 0x44535247u;       // "DSRG"
constexpr int kGroupMaxProblems = 512, kGroupMaxLaunches = kNumCfg + 2;
constexpr int kCfgW3 = kNumCfg;         // pseudo tile configurations kCfgW3 + (dilation - 1): all-taps 3x3 problems (conv_wgrad3.hip)
struct GroupLaunch { int cfg, first, count, grid; double flops, bytes; };
struct GroupHeader {
    unsigned magic; int n, nlaunch, npl, f16;
    GroupLaunch launch[kGroupMaxLaunches];
    int rgrid;                      // blocks of the slab reduce (0: no problem is split)
    long long args_off, starts_off, rstarts_off, used_bytes;      // byte offsets inside the table
};
static size_t group_table_bytes(int n) {
    return align_up(sizeof(GroupHeader), 256) + align_up((size_t)n * sizeof(WgradArgs), 256) + 2 * align_up((size_t)(n + 1) * sizeof(int), 256);
}
static int group_target_px() { return std::max(256, env_int("DSRL_WGRAD_GROUP_PX", 4096)); }
struct GroupItem { WgradArgs a; TileCfg cfg; int bm, bn; double cost, flops, bytes; size_t slab_bytes; };
// Pixel ranges of one problem. Default: one range per ~DSRL_WGRAD_GROUP_PX output pixels. A problem that this leaves with fewer than 8 ranges
// spreads the blocks of a range (taps x tiles) over all 8 XCDs, and every XCD's L2 then fetches the whole of x and dy; with 8 ranges each XCD
// owns one and fetches an eighth. DSRL_WGRAD_XCD_SPLIT=1 takes 8 ranges where the slab traffic this adds (8 slabs written and read back)
// is smaller than the operand re-fetches it removes.
static int group_psplits(long long P, int K, int R, int S, int C, long long x_pixels, int w3_px = 0) {
    const long long chunks = ceil_div(P, 32);
    if (w3_px > 0) {       // a block takes all nine taps of its pixel range: 9 x the work per pixel; pixels per block chosen for the whole launch (w3_group_px)
        const int forced3 = env_int("DSRL_FORCE_PSPLITS", 0), px = w3_px;
        long long sp3 = forced3 > 0 ? forced3 : (P + px / 2) / px;
        sp3 = std::max<long long>(1, std::min<long long>(sp3, std::max<long long>(1, chunks / 8)));
        return (int)std::min<long long>(sp3, 256);
    }
    const int forced = env_int("DSRL_FORCE_PSPLITS", 0);
    long long sp = forced > 0 ? forced : (P + group_target_px() / 2) / group_target_px();
    sp = std::max<long long>(1, std::min<long long>(sp, std::max<long long>(1, chunks / 4)));
    if (forced <= 0 && sp < 8 && chunks >= 64 && env_int("DSRL_WGRAD_XCD_SPLIT", 0)) {
        const double operands = 4.0 * ((double)x_pixels * C + (double)P * K), slabs = 4.0 * (double)K * R * S * C;
        if (16.0 * slabs < 7.0 * operands) sp = 8;
    }
    return (int)std::min<long long>(sp, 256);
}
// Pixels per block of the all-taps launches.  Their blocks run one per CU and are dealt in id order to whichever CU is free, so with equal blocks a
// launch takes ceil(blocks / 256) rounds, and a last round that is nearly full does not fit in practice (1264 blocks of 128 chunks: six rounds, 1.25 ms;
// 1216: five, 1.07 ms).  Measured over the step's 34 problems (tools/sweep_w3_px.sh; launch + slab reduce, ms): 1536 px 1.59, 2048 1.52, 3072 1.58,
// 4096 1.61, 8192 1.56, 16384 1.56 - short blocks balance best and their extra slab traffic costs less than the rounding of long ones.  A list-schedule
// simulation per problem list (block time = chunks + constant, 256 or 240 CUs, slab bytes at 4 TB/s) did not predict the measured order and was dropped.
static int w3_group_px(const dsrl_wgrad_problem*, int, int) { return std::max(256, env_int("DSRL_WGRAD3_PX", 2048)); }
static int group_item(const dsrl_wgrad_problem& q, int npl, GroupItem& it, const int* w3_px) {
    const int N = q.N, H = q.H, W = q.W, C = q.C, K = q.K, R = q.R, S = q.S, stride = q.stride, pad = q.pad, dil = q.dil;
    if (int e = check_conv(q.x, q.dy, q.dw, N, H, W, C, K, R, S, stride, pad, dil)) return e;
    DSRL_REQUIRE(C % 4 == 0 && q.ldx % 4 == 0 && q.lddy % 4 == 0 && q.lddy >= pad4(K) && ((uintptr_t)q.x % 16) == 0 && ((uintptr_t)q.dy % 16) == 0 && ((uintptr_t)q.dw % 16) == 0,
                 DSRL_E_UNSUPPORTED, "conv2d_wgrad_group: C (%d), ldx (%d), lddy (%d) must be multiples of 4 (lddy >= K rounded up to 4), pointers 16-byte aligned", C, q.ldx, q.lddy);
    const WgPlan p = plan_wgrad(N, H, W, C, K, R, S, stride, pad, dil);
    DSRL_REQUIRE(p.P < (1ll << 31), DSRL_E_UNSUPPORTED, "conv2d_wgrad_group: more than 2^31 output pixels");
    WgradArgs a{};
    a.x = q.x; a.dy = q.dy; a.ldx = q.ldx; a.lddy = q.lddy; a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.Ho = p.Ho; a.Wo = p.Wo;
    a.stride = stride; a.pad = pad; a.dil = dil; a.P = p.P; a.ctiles = p.ctiles; a.slab = (long long)K * R * S * C;
    const long long xb = span_bytes((long long)N * H * W, q.ldx, C), db = span_bytes(p.P, q.lddy, pad4(K));
    DSRL_REQUIRE_31(xb, "conv2d_wgrad_group(x)"); DSRL_REQUIRE_31(db, "conv2d_wgrad_group(dy)");
    a.x_bytes = (unsigned)xb; a.dy_bytes = (unsigned)db; a.no_ident = wgrad_no_ident();
    a.ntaps = p.tl.n;
    for (int i = 0; i < p.tl.n; ++i) a.taps[i] = p.tl.taps[i];
    a.psplits = group_psplits(p.P, K, R, S, C, (long long)N * H * W, p.w3 ? w3_px[dil - 1] : 0);
    a.kg = 1; a.kctiles = p.ktiles * p.ctiles; a.xcd_remap = env_int("DSRL_XCD_REMAP", 1);
    a.nblocks = p.w3 ? a.kctiles * a.psplits : a.kctiles * a.ntaps * a.psplits;
    a.dw_final = q.dw; a.dw = q.dw;
    a.rblocks = a.psplits > 1 ? (int)ceil_div((long long)K * a.ntaps * (C / 4), 1024) : 0;
    make_magic(a.Ho * a.Wo, a.mHW, a.sHW);
    make_magic(a.Wo, a.mW, a.sW);
    DSRL_REQUIRE(!conv_f16() || (q.x_amax != nullptr && q.dy_amax != nullptr), DSRL_E_BADARG,
                 "conv2d_wgrad_group: the f16x3 arithmetic needs x_amax / dy_amax of every problem (dsrl_amax measures a tensor)");
    a.amax_x = q.x_amax; a.amax_dy = q.dy_amax;
    it.a = a; it.cfg = p.w3 ? (TileCfg)(kCfgW3 + dil - 1) : p.cfg; it.bm = p.bm; it.bn = p.bn;
    it.cost = (double)p.P / a.psplits * p.bm * p.bn;
    it.flops = 2.0 * (double)dsrl_conv2d_inbounds_macs(N, H, W, C, K, R, S, stride, pad, dil);
    it.bytes = 4.0 * ((double)N * H * W * C + (double)K * R * S * C + (double)p.P * K);
    it.slab_bytes = a.psplits > 1 ? align_up((size_t)a.psplits * a.slab * sizeof(float), 256) : 0;
    (void)npl;
    return DSRL_OK;
}
}  // namespace dsrl

extern "C" size_t dsrl_conv2d_wgrad_group_table_bytes(int n) { return n > 0 ? group_table_bytes(n) : 0; }

extern "C" size_t dsrl_conv2d_wgrad_group_workspace_bytes(const dsrl_wgrad_problem* problems, int n) {
    if (!problems || n <= 0) return 0;
    size_t total = 0;
    const int w3_px[2] = {dsrl::w3_group_px(problems, n, 1), dsrl::w3_group_px(problems, n, 2)};
    for (int i = 0; i < n; ++i) {
        const dsrl_wgrad_problem& q = problems[i];
        if (q.N <= 0 || q.H <= 0 || q.W <= 0 || q.C <= 0 || q.K <= 0 || q.R <= 0 || q.S <= 0 || q.stride <= 0 || q.dil <= 0 || q.pad < 0) continue;
        if (out_size(q.H, q.R, q.stride, q.pad, q.dil) <= 0 || out_size(q.W, q.S, q.stride, q.pad, q.dil) <= 0) continue;
        const long long P = (long long)q.N * out_size(q.H, q.R, q.stride, q.pad, q.dil) * out_size(q.W, q.S, q.stride, q.pad, q.dil);
        const bool w3 = out_size(q.H, q.R, q.stride, q.pad, q.dil) == q.H && plan_wgrad(q.N, q.H, q.W, q.C, q.K, q.R, q.S, q.stride, q.pad, q.dil).w3;
        const int sp = dsrl::group_psplits(P, q.K, q.R, q.S, q.C, (long long)q.N * q.H * q.W, w3 ? w3_px[q.dil - 1] : 0);
        if (sp > 1) total += align_up((size_t)sp * q.K * q.R * q.S * q.C * sizeof(float), 256);
    }
    return total;
}

extern "C" int dsrl_conv2d_wgrad_group_plan(const dsrl_wgrad_problem* problems, int n, void* host_table, size_t table_bytes, const void* dev_table,
                                            void* ws, size_t ws_bytes) {
    DSRL_REQUIRE(problems && host_table && dev_table && n > 0 && n <= kGroupMaxProblems, DSRL_E_BADARG, "conv2d_wgrad_group_plan: bad arguments (n=%d, at most %d problems)", n, kGroupMaxProblems);
    DSRL_REQUIRE(table_bytes >= group_table_bytes(n), DSRL_E_WORKSPACE, "conv2d_wgrad_group_plan: table %zu < %zu bytes", table_bytes, group_table_bytes(n));
    DSRL_REQUIRE(((uintptr_t)dev_table % 16) == 0 && ((uintptr_t)host_table % 16) == 0, DSRL_E_BADARG, "conv2d_wgrad_group_plan: tables must be 16-byte aligned");
    const int npl = conv_planes(PASS_WGRAD);
    DSRL_REQUIRE(npl >= 1 && npl <= 3, DSRL_E_UNSUPPORTED, "conv2d_wgrad_group_plan: the grouped launch exists for the 16-bit MFMA arithmetics only (dsrl_conv_precision 1..5)");
    std::vector<GroupItem> items((size_t)n);
    size_t need_ws = 0;
    const int w3_px[2] = {w3_group_px(problems, n, 1), w3_group_px(problems, n, 2)};
    for (int i = 0; i < n; ++i) {
        if (int e = group_item(problems[i], npl, items[(size_t)i], w3_px)) return e;
        need_ws += items[(size_t)i].slab_bytes;
    }
    DSRL_REQUIRE(ws_bytes >= need_ws && (need_ws == 0 || ws), DSRL_E_WORKSPACE, "conv2d_wgrad_group_plan: workspace %zu < %zu", ws_bytes, need_ws);
    DSRL_REQUIRE(need_ws == 0 || ((uintptr_t)ws % 16) == 0, DSRL_E_BADARG, "conv2d_wgrad_group_plan: workspace must be 16-byte aligned");
    // order: by tile configuration, inside one longest blocks first (the hardware deals blocks in id order: the short ones fill the tail)
    std::stable_sort(items.begin(), items.end(), [](const GroupItem& l, const GroupItem& r) { return l.cfg != r.cfg ? l.cfg < r.cfg : l.cost > r.cost; });
    char* base = (char*)host_table;
    memset(base, 0, group_table_bytes(n));
    GroupHeader* h = (GroupHeader*)base;
    h->magic = kGroupMagic; h->n = n; h->npl = npl; h->nlaunch = 0; h->f16 = conv_f16() ? 1 : 0;
    h->args_off = (long long)align_up(sizeof(GroupHeader), 256);
    h->starts_off = h->args_off + (long long)align_up((size_t)n * sizeof(WgradArgs), 256);
    h->rstarts_off = h->starts_off + (long long)align_up((size_t)(n + 1) * sizeof(int), 256);
    h->used_bytes = (long long)group_table_bytes(n);
    WgradArgs* args = (WgradArgs*)(base + h->args_off);
    int* starts = (int*)(base + h->starts_off);
    int* rstarts = (int*)(base + h->rstarts_off);
    size_t ws_off = 0;
    int rtotal = 0;
    for (int i = 0; i < n; ++i) {
        GroupItem& it = items[(size_t)i];
        if (it.a.psplits > 1) { it.a.dw = (float*)((char*)ws + ws_off); ws_off += it.slab_bytes; }
        args[i] = it.a;
        rstarts[i] = rtotal;
        rtotal += it.a.rblocks;
        if (h->nlaunch == 0 || h->launch[h->nlaunch - 1].cfg != (int)it.cfg) {
            GroupLaunch& L = h->launch[h->nlaunch++];
            L.cfg = (int)it.cfg; L.first = i; L.count = 0; L.grid = 0; L.flops = 0; L.bytes = 0;
        }
        GroupLaunch& L = h->launch[h->nlaunch - 1];
        starts[i] = L.grid;                                       // local to the launch; multiples of 8
        L.grid += (int)align_up((size_t)it.a.nblocks, 8);
        L.count += 1; L.flops += it.flops; L.bytes += it.bytes;
    }
    rstarts[n] = rtotal;
    h->rgrid = rtotal;
    return DSRL_OK;
}

extern "C" int dsrl_conv2d_wgrad_group_launch(const void* host_table, const void* dev_table, dsrl_stream_t stream) {
    DSRL_REQUIRE(host_table && dev_table, DSRL_E_BADARG, "conv2d_wgrad_group_launch: null table");
    const GroupHeader* h = (const GroupHeader*)host_table;
    DSRL_REQUIRE(h->magic == kGroupMagic && h->n > 0 && h->nlaunch > 0 && h->nlaunch <= kGroupMaxLaunches, DSRL_E_BADARG, "conv2d_wgrad_group_launch: the host table was not written by dsrl_conv2d_wgrad_group_plan");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    const WgradArgs* hargs = (const WgradArgs*)((const char*)host_table + h->args_off);
    const WgradArgs* dargs = (const WgradArgs*)((const char*)dev_table + h->args_off);
    const int* dstarts = (const int*)((const char*)dev_table + h->starts_off);
    const int* drstarts = (const int*)((const char*)dev_table + h->rstarts_off);
    for (int i = 0; i < h->n; ++i) {        // taps that only ever see zero padding have a zero gradient
        const WgradArgs& a = hargs[i];
        if (a.ntaps < a.R * a.S)
            if (int e = launch_zero_fill(a.dw_final, (size_t)a.K * a.R * a.S * a.C * sizeof(float), st)) return e;
    }
    const int npl = h->npl;
    for (int l = 0; l < h->nlaunch; ++l) {
        const GroupLaunch& L = h->launch[l];
        if (L.grid <= 0) continue;
        const WgradArgs* t = dargs + L.first;
        const int* s = dstarts + L.first;
        ProfScope prof(3 * prof_arith(npl, h->f16 != 0) + 1, L.flops, L.bytes, st);
        if (L.cfg >= kCfgW3) {
            if (int e = launch_wgrad3_group(t, s, L.count, L.grid, npl, L.cfg - kCfgW3 + 1, st)) return e;
            continue;
        }
        int bm, bn; cfg_dims((TileCfg)L.cfg, bm, bn);
        const size_t lds = (size_t)2 * npl * 16 * ((bm * 2 + 64) + (bn * 2 + 64));
#define DSRL_LAUNCH_WGROUP(a_, b_, c_, d_)                                                                                         \
        if (h->f16 && npl == 1) hipLaunchKernelGGL((conv_wgrad_group_kernel<a_, b_, c_, d_, 1, true>), dim3((unsigned)L.grid), dim3(256), lds, st, t, s, L.count); \
        else if (h->f16) hipLaunchKernelGGL((conv_wgrad_group_kernel<a_, b_, c_, d_, 2, true>), dim3((unsigned)L.grid), dim3(256), lds, st, t, s, L.count); \
        else if (npl == 2) hipLaunchKernelGGL((conv_wgrad_group_kernel<a_, b_, c_, d_, 2>), dim3((unsigned)L.grid), dim3(256), lds, st, t, s, L.count); \
        else hipLaunchKernelGGL((conv_wgrad_group_kernel<a_, b_, c_, d_, 3>), dim3((unsigned)L.grid), dim3(256), lds, st, t, s, L.count);
        DSRL_CFG_SWITCH((TileCfg)L.cfg, DSRL_LAUNCH_WGROUP)
#undef DSRL_LAUNCH_WGROUP
        if (int e = launch_status("conv_wgrad_group_kernel")) return e;
    }
    if (h->rgrid > 0) {
        hipLaunchKernelGGL(wgrad_reduce_group_kernel, dim3((unsigned)h->rgrid), dim3(256), 0, st, dargs, drstarts, h->n);
        return launch_status("wgrad_reduce_group_kernel");
    }
    return DSRL_OK;
}

// ------------------------------------------------------------------------------------------------ row-folded conv (stem)
static int check_rowfold(const void* a, const void* b, const void* c, int ldx, int N, int H, int W, int Cf, int K, int R, int stride, int Ho, int Wo) {
    DSRL_REQUIRE(a && b && c, DSRL_E_BADARG, "conv2d_rowfold: null pointer");
    DSRL_REQUIRE(N > 0 && H > 0 && W > 0 && Cf > 0 && K > 0 && R > 0 && R <= 64 && stride > 0 && Ho > 0 && Wo > 0 && ldx > 0, DSRL_E_BADARG, "conv2d_rowfold: bad shape");
    DSRL_REQUIRE(Cf % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)a % 16) == 0, DSRL_E_UNSUPPORTED, "conv2d_rowfold: Cfold and ldx must be multiples of 4, x 16-byte aligned");
    DSRL_REQUIRE((long long)(Ho - 1) * stride + R <= H && (long long)(Wo - 1) * stride * ldx + Cf <= (long long)W * ldx, DSRL_E_BADARG,
                 "conv2d_rowfold: the folded window leaves the (pre-padded) image");
    return 0;
}
extern "C" size_t dsrl_conv2d_rowfold_fwd_workspace_bytes(int N, int H, int W, int Cfold, int K, int R, int stride, int Ho, int Wo) {
    (void)stride;
    return with_amax_scratch(plan_fwd_ws(N, H, W, Cfold, K, R, 1, Ho, Wo));
}
extern "C" int dsrl_conv2d_rowfold_fwd(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                                       int N, int H, int W, int Cfold, int K, int R, int stride, int Ho, int Wo, int64_t algorithmic_macs,
                                       void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    if (int e = check_rowfold(x, w, y, ldx, N, H, W, Cfold, K, R, stride, Ho, Wo)) return e;
    DSRL_REQUIRE(ldy >= K && ((uintptr_t)w % 16) == 0, DSRL_E_BADARG, "conv2d_rowfold_fwd: bad ldy / unaligned filter");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    const FwdPlan p = plan_fwd(N, H, W, Cfold, K, R, 1, Ho, Wo, conv_planes(PASS_FWD));
    DSRL_REQUIRE(ws_bytes >= p.ws && (p.ws == 0 || ws), DSRL_E_WORKSPACE, "conv2d_rowfold_fwd: workspace %zu < %zu", ws_bytes, p.ws);
    ConvArgs a{};
    a.x = x; a.w = w; a.ldx = ldx; a.N = N; a.H = H; a.W = W; a.C = Cfold; a.K = K; a.R = R; a.S = 1; a.Ho = Ho; a.Wo = Wo;
    a.stride = stride; a.pad = 0; a.dil = 1; a.M = p.M; a.cchunks = p.cchunks; a.splits = p.splits; a.kg = p.kg; a.slab = (long long)p.M * K;
    {
        const long long xb = (long long)N * H * W * ldx * 4, wb = (long long)K * R * Cfold * 4, yb = p.splits > 1 ? (long long)p.M * K * 4 : span_bytes(p.M, ldy, K);
        DSRL_REQUIRE_31(xb, "conv2d_rowfold_fwd(x)"); DSRL_REQUIRE_31(wb, "conv2d_rowfold_fwd(w)"); DSRL_REQUIRE_31(yb, "conv2d_rowfold_fwd(y)");
        a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb; a.y_bytes = (unsigned)yb;
    }
    if (conv_f16()) {       // the image and the folded filter are measured by the call itself
        OperandAmax am{nullptr, nullptr};
        if (int e = resolve_amax(am, x, ldx, (long long)N * H * W, ldx, w, Cfold, (long long)K * R, Cfold, ws, ws_bytes, p.ws, st, "conv2d_rowfold_fwd")) return e;
        a.amax_a = am.a; a.amax_b = am.b;
    }
    ProfScope prof(prof_family(PASS_FWD), 2.0 * (double)algorithmic_macs, 4.0 * ((double)N * H * W * ldx + (double)K * R * Cfold + (double)N * Ho * Wo * K), st);
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

namespace dsrl {
static WgPlan plan_wgrad_rowfold(int N, int Cf, int K, int R, int Ho, int Wo) {
    WgPlan p; p.Ho = Ho; p.Wo = Wo; p.P = (long long)N * Ho * Wo;
    p.cfg = pick_cfg_wgrad(K, Cf);
    cfg_dims(p.cfg, p.bm, p.bn);
    p.ktiles = (int)ceil_div(K, p.bm); p.ctiles = (int)ceil_div(Cf, p.bn);
    p.tl.n = R;
    for (int r = 0; r < R; ++r) p.tl.taps[r] = r;
    const long long tiles = (long long)p.ktiles * p.ctiles * R, chunks = ceil_div(p.P, 32);
    p.psplits = pick_psplits(tiles, chunks);
    p.ws = p.psplits > 1 ? (size_t)p.psplits * K * R * Cf * sizeof(float) : 0;
    return p;
}
}  // namespace dsrl
extern "C" size_t dsrl_conv2d_rowfold_wgrad_workspace_bytes(int N, int H, int W, int Cfold, int K, int R, int stride, int Ho, int Wo) {
    (void)H; (void)W; (void)stride;
    return with_amax_scratch(plan_wgrad_rowfold(N, Cfold, K, R, Ho, Wo).ws);
}
extern "C" int dsrl_conv2d_rowfold_wgrad(const float* x, int ldx, const float* dy, int lddy, float* dw,
                                         int N, int H, int W, int Cfold, int K, int R, int stride, int Ho, int Wo, int64_t algorithmic_macs,
                                         void* ws, size_t ws_bytes, dsrl_stream_t stream) {
    if (int e = check_rowfold(x, dy, dw, ldx, N, H, W, Cfold, K, R, stride, Ho, Wo)) return e;
    DSRL_REQUIRE(lddy % 4 == 0 && lddy >= pad4(K) && ((uintptr_t)dy % 16) == 0, DSRL_E_UNSUPPORTED, "conv2d_rowfold_wgrad: lddy must be a multiple of 4, dy aligned");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    const WgPlan p = plan_wgrad_rowfold(N, Cfold, K, R, Ho, Wo);
    DSRL_REQUIRE(p.P < (1ll << 31), DSRL_E_UNSUPPORTED, "conv2d_rowfold_wgrad: more than 2^31 output pixels");
    DSRL_REQUIRE(ws_bytes >= p.ws && (p.ws == 0 || ws), DSRL_E_WORKSPACE, "conv2d_rowfold_wgrad: workspace %zu < %zu", ws_bytes, p.ws);
    WgradArgs a{};
    a.x = x; a.dy = dy; a.ldx = ldx; a.lddy = lddy; a.N = N; a.H = H; a.W = W; a.C = Cfold; a.K = K; a.R = R; a.S = 1; a.Ho = Ho; a.Wo = Wo;
    a.stride = stride; a.pad = 0; a.dil = 1; a.P = p.P; a.ctiles = p.ctiles; a.psplits = p.psplits; a.slab = (long long)K * R * Cfold;
    {
        const long long xb = (long long)N * H * W * ldx * 4, db = span_bytes(p.P, lddy, pad4(K));
        DSRL_REQUIRE_31(xb, "conv2d_rowfold_wgrad(x)"); DSRL_REQUIRE_31(db, "conv2d_rowfold_wgrad(dy)");
        a.x_bytes = (unsigned)xb; a.dy_bytes = (unsigned)db; a.no_ident = wgrad_no_ident();
    }
    a.ntaps = R;
    for (int r = 0; r < R; ++r) a.taps[r] = r;
    a.dw = p.psplits > 1 ? (float*)ws : dw;
    a.kctiles = p.ktiles * p.ctiles; a.xcd_remap = env_int("DSRL_XCD_REMAP", 1);
    dim3 grid((unsigned)(a.kctiles * R * p.psplits));
    if (conv_f16()) {
        OperandAmax am{nullptr, nullptr};
        if (int e = resolve_amax(am, dy, lddy, p.P, pad4(K), x, ldx, (long long)N * H * W, ldx, ws, ws_bytes, p.ws, st, "conv2d_rowfold_wgrad")) return e;
        a.amax_dy = am.a; a.amax_x = am.b;
    }
    ProfScope prof(prof_family(PASS_WGRAD), 2.0 * (double)algorithmic_macs, 4.0 * ((double)N * H * W * ldx + (double)K * R * Cfold + (double)N * Ho * Wo * K), st);
    launch_wgrad(a, p.cfg, p.bm, p.bn, grid, st);
    if (int e = launch_status("conv_wgrad kernel")) return e;
    if (p.psplits > 1) {
        const long long total = (long long)K * R * (Cfold / 4);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 4096)), dim3(256), 0, st,
                           (const float*)ws, p.psplits, a.slab, dw, K, R, Cfold, p.tl);
        return launch_status("wgrad_reduce_kernel");
    }
    return DSRL_OK;
}

extern "C" int dsrl_conv_precision(int mode) {
    const int prev = g_conv_precision.load();
    if (mode >= -1 && mode <= 5) g_conv_precision.store(mode);
    return prev;
}

extern "C" int dsrl_prof_enable(int on) {
    std::lock_guard<std::mutex> g(g_prof_mu);
    g_prof_stride = on > 0 ? on : 0;
    g_prof_seq.store(0);
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
        if (const char* dump = getenv("DSRL_PROF_DUMP")) {          // diagnosis (tools/per_launch.py): one line per bracketed launch
            if (FILE* f = fopen(dump, "a")) { fprintf(f, "%d\t%.6f\t%.0f\t%.0f\t%s\n", r.family, (double)t, r.flops, r.bytes, r.tag); fclose(f); }
        }
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    return DSRL_OK;
}

extern "C" int dsrl_prof_read_bytes(int family, double* total_bytes) {
    std::lock_guard<std::mutex> g(g_prof_mu);
    double b = 0;
    for (auto& r : g_prof)
        if (r.family == family) b += r.bytes;
    if (total_bytes) *total_bytes = b;
    return DSRL_OK;
}

extern "C" const char* dsrl_prof_kernel_name(int family) {
    static const char* names[15] = {
        "conv_igemm_f32_kernel (forward)", "conv_wgrad_f32_kernel", "conv_igemm_f32_kernel (dgrad)",
        "conv_igemm_split_kernel<bf16x3> (forward)", "conv_wgrad_split_kernel<bf16x3>", "conv_igemm_split_kernel<bf16x3> (dgrad)",
        "conv_igemm_split_kernel<bf16x6> (forward)", "conv_wgrad_split_kernel<bf16x6>", "conv_igemm_split_kernel<bf16x6> (dgrad)",
        "conv_igemm_split_kernel<f16x3> (forward)", "conv_wgrad_split_kernel<f16x3>", "conv_igemm_split_kernel<f16x3> (dgrad)",
        "conv_igemm_split_kernel<f16x1> (forward)", "conv_wgrad_split_kernel<f16x1>", "conv_igemm_split_kernel<f16x1> (dgrad)"};
    return family >= 0 && family < 15 ? names[family] : "";
}
