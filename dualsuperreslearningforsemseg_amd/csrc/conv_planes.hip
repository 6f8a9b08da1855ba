// Implicit-GEMM conv2d forward / data gradient whose operands arrive as fp16 PLANES and are staged by LDS-DMA (round 4).
//
// The f16x3 arithmetic (conv_common.h) multiplies two fp16 terms per operand, hi = f16(v * 2^e), lo = f16(v * 2^e - hi).  conv_igemm.hip forms
// the terms inside the K loop (global -> VGPR -> convert -> ds_write).  Here every operand tensor exists in memory as two fp16 planes of the
// tensor's own shape ([P][ld] for activations, [K][R][S][C] / [C][R][S][K] for filters: dsrl_split_planes, dsrl_conv2d_filter_planes_batched,
// or written by the producing kernel), so a K step is
//
//     buffer_load_dwordx4 ... lds   (16 rows x 64 B of one plane per wave instruction, no VGPR round trip, no conversion, no ds_write)
//     counted s_waitcnt vmcnt(N) + ONE s_barrier per 32-channel step
//     ds_read_b128 fragments -> v_mfma_f32_32x32x16_f16 (a0*b1, a1*b0, a0*b0 per 16 channels: the order of conv_igemm_split_kernel, so the
//     results are bit-identical to it for the same tile / K-group plan)
//
// LDS ring of R slots per K group; a slot holds one 32-channel chunk of the block tile: [plane][row][64 B].  The 16-byte units of a row are stored
// XOR-swizzled by bits 2..3 of the row (unit u of row r at position u ^ ((r >> 2) & 3)): a ds_read_b128 of 16 lanes then covers all 64 banks.  LDS-DMA
// writes lane l at base + 16 l, so the swizzle is applied to the SOURCE address each lane fetches (cdna_hip_programming.md rule 21).  Out-of-range
// offsets (zero padding of the conv, channel / row tails, exhausted K groups) read as zeros through the buffer descriptor's bounds check.
#include "common.h"
#include "conv_common.h"
#include "lds_dma.h"
#include <algorithm>

namespace dsrl {

template <int NPL> struct PlaneMfma;
template <> struct PlaneMfma<2> {
    template <int MR, int NR>
    static __device__ __forceinline__ void run(f32x16 (&acc)[MR][NR], const f16x8 (&fa)[MR][2], const f16x8 (&fb)[NR][2]) {
#pragma unroll
        for (int sum = 1; sum >= 0; --sum)
#pragma unroll
            for (int pa = 0; pa <= sum; ++pa)
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NR; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][pa], fb[j][sum - pa], acc[i][j], 0, 0, 0);
    }
};
template <> struct PlaneMfma<1> {
    template <int MR, int NR>
    static __device__ __forceinline__ void run(f32x16 (&acc)[MR][NR], const f16x8 (&fa)[MR][1], const f16x8 (&fb)[NR][1]) {
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
    }
};

// MR x NR 32x32 MFMA tiles per wave, WGM x WGN waves per K group, KG K groups per block (interleaved chunks, accumulators summed through LDS in the
// order g = 0 .. KG-1), NPL planes per operand (2 = f16x3, 1 = f16x1), R ring slots per group.  DGRAD: stride-1 data gradient (a.x = dy planes,
// a.w = planes of the transposed filter [C][R][S][K]); strided data gradients stay on conv_igemm_split_kernel.
template <int MR, int NR, int WGM, int WGN, int KG, int NPL, bool DGRAD, int R, int DBG = 0>
__global__ __launch_bounds__(64 * WGM * WGN * KG, (64 * WGM * WGN * KG >= 512) ? 1 : 2)
void conv_planes_kernel(const ConvArgs a) {
    constexpr int NW = WGM * WGN, NT = 64 * NW;
    constexpr int BM = 32 * MR * WGM, BN = 32 * NR * WGN;
    constexpr int ABLK = BM / 16, BBLK = BN / 16;           // 16-row DMA pieces per plane
    static_assert(ABLK % NW == 0 && BBLK % NW == 0, "every wave stages the same number of pieces");
    constexpr int A_IT = ABLK / NW, B_IT = BBLK / NW;
    constexpr int PW = (A_IT + B_IT) * NPL;                 // pieces per wave and step
    static_assert((R - 1) * PW <= 60, "vmcnt is a 6-bit counter");
    constexpr int ROWB = 64;                                // bytes per row and plane in a slot: 32 channels
    constexpr int SLOT = (BM + BN) * NPL * ROWB;
    static_assert(R >= 2, "ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (DBG == 4) return;                                   // fixed-cost dissection: the launch alone
    const unsigned am_a = amax_fetch(a.amax_a), am_b = amax_fetch(a.amax_b);      // consumed in the epilogue (the planes already carry the scales)
    const int grp = KG > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x / NT)) : 0;
    char* const ring = smem + grp * R * SLOT;
    const unsigned ring_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring;

    const int tid = threadIdx.x % NT;           // thread within its K group
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int tile = a.xcd_remap ? xcd_contiguous(blockIdx.x, a.mtiles * a.ntiles) : blockIdx.x;
    const int tile_m = fast_div(tile, a.mNT, a.sNT);
    const int m0 = tile_m * BM, n0 = (tile - tile_m * a.ntiles) * BN, z = blockIdx.z;
    const int HoWo = a.Ho * a.Wo;

    // ---- what this lane fetches: row (lane >> 2) of each of its pieces, logical 16-byte unit (lane & 3) ^ swizzle(row)
    const int prow = lane >> 2;
    const int unit = (lane & 3) ^ ((lane >> 4) & 3);        // (row >> 2) & 3 of a piece row = (lane >> 4) & 3: piece bases are multiples of 16
    int a_n[A_IT], a_h[A_IT], a_w[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + (wave + NW * i) * 16 + prow;
        a_ok[i] = m < a.M;
        const int mm = a_ok[i] ? m : 0;
        const int n = fast_div(mm, a.mHW, a.sHW), rem = mm - n * HoWo;
        const int ho = fast_div(rem, a.mW, a.sW), wo = rem - ho * a.Wo;
        a_n[i] = n;
        if (DGRAD) { a_h[i] = ho + a.pad; a_w[i] = wo + a.pad; }
        else { a_h[i] = ho * a.stride - a.pad; a_w[i] = wo * a.stride - a.pad; }
    }
    const int RS = a.R * a.S;
    // the descriptors span BOTH planes: a raw buffer's range check takes the scalar offset into account (offset >= num_records - soffset is out of
    // range), so the second plane, reached through the scalar offset, is checked against the same per-plane extent as the first
    const u32x4 xr = make_rsrc(a.x, a.x_bytes + (NPL > 1 ? a.a_lo : 0u)), wr = make_rsrc(a.w, a.w_bytes + (NPL > 1 ? a.b_lo : 0u));
    unsigned b_off[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int k = n0 + (wave + NW * i) * 16 + prow;
        b_off[i] = k < a.K ? (unsigned)k * (unsigned)(RS * a.C) * 2u : kOOB;
    }

    // ---- taps that touch at least one in-bounds input pixel for this tile (block-uniform)
    unsigned long long tapmask = 0ull;
    if (RS == 1 && a.pad == 0) {
        tapmask = 1ull;
    } else {
        const int mf = m0, ml = min(m0 + BM, a.M) - 1;
        const int nf = fast_div(mf, a.mHW, a.sHW), nl = fast_div(ml, a.mHW, a.sHW);
        int hf = 0, hl = a.Ho - 1, wf = 0, wl = a.Wo - 1;
        if (nf == nl) {
            hf = fast_div(mf - nf * HoWo, a.mW, a.sW); hl = fast_div(ml - nl * HoWo, a.mW, a.sW);
            if (hf == hl) { wf = (mf - nf * HoWo) - hf * a.Wo; wl = (ml - nl * HoWo) - hl * a.Wo; }
        }
        for (int r = 0; r < a.R; ++r)
            for (int s = 0; s < a.S; ++s) {
                bool act;
                if (DGRAD) {
                    act = (hl + a.pad - r * a.dil >= 0) && (hf + a.pad - r * a.dil <= a.H - 1) &&
                          (wl + a.pad - s * a.dil >= 0) && (wf + a.pad - s * a.dil <= a.W - 1);
                } else {
                    act = (hl * a.stride - a.pad + r * a.dil >= 0) && (hf * a.stride - a.pad + r * a.dil <= a.H - 1) &&
                          (wl * a.stride - a.pad + s * a.dil >= 0) && (wf * a.stride - a.pad + s * a.dil <= a.W - 1);
                }
                if (act) tapmask |= 1ull << (r * a.S + s);
            }
    }
    const int ntaps = __builtin_popcountll(tapmask);
    const int nq = ntaps * a.cchunks;
    int q0 = 0, q1 = nq;
    if (a.splits > 1) { q0 = (int)((long long)nq * z / a.splits); q1 = (int)((long long)nq * (z + 1) / a.splits); }

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // iterator over (tap, channel chunk), positioned at this group's first chunk q0 + grp (as in conv_igemm_split_kernel)
    int cc = 0, tap = 0, pos = q0;
    unsigned long long rem_mask = tapmask;
    if (q0 < q1) {
        if (q0 > 0) {
            int skip = q0 / a.cchunks;
            cc = q0 - skip * a.cchunks;
            while (skip--) rem_mask &= rem_mask - 1;
        }
        tap = __builtin_ctzll(rem_mask);
    }
    auto advance = [&](int n) {
        cc += n; pos += n;
        while (cc >= a.cchunks) { cc -= a.cchunks; rem_mask &= rem_mask - 1; tap = __builtin_ctzll(rem_mask); }
    };
    unsigned a_off[A_IT];
    const unsigned inv_s = 65536u / (unsigned)a.S + 1u;
    auto set_tap = [&](int t) {
        const int r = a.S <= 8 ? (int)(((unsigned)t * inv_s) >> 16) : t / a.S, s = t - r * a.S;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int hi, wi;
            if (DGRAD) { hi = a_h[i] - r * a.dil; wi = a_w[i] - s * a.dil; }
            else { hi = a_h[i] + r * a.dil; wi = a_w[i] + s * a.dil; }
            const bool ok = a_ok[i] && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
            a_off[i] = ok ? (unsigned)((a_n[i] * a.H + hi) * a.W + wi) * (unsigned)a.ldx * 2u : kOOB;
        }
    };
    int nextq = q0 + grp, tap_set = -1;
    // Every step issues the same PW pieces - past the end of the group's chunks with out-of-range offsets (zeros land in the slot): the counted
    // waits below stay exact, and an exhausted K group multiplies zeros while the others finish.  prepare() advances the iterator and forms the
    // piece offsets (scalar work + a few vector adds); piece<idx>() is one DMA instruction, so that the pieces can sit between the MFMAs.
    unsigned va[A_IT], vb[B_IT], pbase = 0u;
    auto prepare = [&](int slot) {
        const bool live = nextq < q1;
        if (live) {
            advance(nextq - pos);
            if (tap != tap_set) { set_tap(tap); tap_set = tap; }
            nextq += KG;
        }
        const int c = cc * 32 + unit * 8;
        const unsigned coff = (c < a.C ? (unsigned)c * 2u : kOOB) | (live ? 0u : kOOB);
        const unsigned woff = __builtin_elementwise_add_sat(coff, (unsigned)(tap * a.C) * 2u);
        pbase = ring_lds + (unsigned)(slot * SLOT);
#pragma unroll
        for (int i = 0; i < A_IT; ++i) va[i] = __builtin_elementwise_add_sat(a_off[i], coff);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) vb[i] = __builtin_elementwise_add_sat(b_off[i], woff);
    };
    auto piece = [&](int idx) {              // idx is a compile-time constant at every call site (unrolled loops)
        if (idx < A_IT * NPL) {
            const int i = idx / NPL, pl = idx % NPL;
            lds_dma16(xr, va[i], pl ? a.a_lo : 0u, pbase + (unsigned)((pl * BM + (wave + NW * i) * 16) * ROWB));
        } else {
            const int i = (idx - A_IT * NPL) / NPL, pl = (idx - A_IT * NPL) % NPL;
            lds_dma16(wr, vb[i], pl ? a.b_lo : 0u, pbase + (unsigned)((NPL * BM + pl * BN + (wave + NW * i) * 16) * ROWB));
        }
    };
    auto issue = [&](int slot) {
        prepare(slot);
#pragma unroll
        for (int idx = 0; idx < PW; ++idx) piece(idx);
    };

    // ---- fragment reads: row = lane & 31 of a 32-row tile, k half = lane >> 5 of the 16-channel sub-step `sub`
    const int frag_row = lane & 31;
    const int swz = (lane >> 2) & 3;                        // (row >> 2) & 3: tile bases are multiples of 32
    const int fr_off = frag_row * ROWB;
    int u_off[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) u_off[sub] = (((sub * 2 + (lane >> 5)) ^ swz) << 4) + fr_off;
    // One step: the DMA pieces of the step that refills the slot freed by the last barrier go out first (spreading them between the MFMAs was
    // measured: no gain on long K loops, a loss on short ones - the data of the next step lands later), then the fragments of both 16-channel
    // sub-steps are requested (small tiles; large tiles: one sub-step ahead), then the MFMAs in the order a0*b1, a1*b0, a0*b0 per sub-step.
    constexpr bool ALLFRAGS = MR * NR <= 2;                        // both sub-steps' fragments fit in registers
    auto compute = [&](int slot, int refill) {
        const char* cur = ring + slot * SLOT;
        if (DBG != 2 && DBG != 3 && DBG != 6) prepare(refill);
        f16x8 fa[2][MR][NPL], fb[2][NR][NPL];
        auto rd = [&](int sub) {
            // the order the MFMAs need them: last plane of the filter fragments, first plane of the pixel fragments, then the rest
#pragma unroll
            for (int j = 0; j < NR; ++j) fb[sub][j][NPL - 1] = *reinterpret_cast<const f16x8*>(cur + (NPL * BM + (NPL - 1) * BN + (wn * NR + j) * 32) * ROWB + u_off[sub]);
#pragma unroll
            for (int i = 0; i < MR; ++i) fa[sub][i][0] = *reinterpret_cast<const f16x8*>(cur + ((wm * MR + i) * 32) * ROWB + u_off[sub]);
            if constexpr (NPL == 2) {
#pragma unroll
                for (int i = 0; i < MR; ++i) fa[sub][i][1] = *reinterpret_cast<const f16x8*>(cur + (BM + (wm * MR + i) * 32) * ROWB + u_off[sub]);
#pragma unroll
                for (int j = 0; j < NR; ++j) fb[sub][j][0] = *reinterpret_cast<const f16x8*>(cur + (NPL * BM + (wn * NR + j) * 32) * ROWB + u_off[sub]);
            }
        };
        if (DBG != 2 && DBG != 3 && DBG != 6) {
#pragma unroll
            for (int idx = 0; idx < PW; ++idx) piece(idx);
        }
        if (DBG == 1 || DBG == 3 || DBG == 6) return;
        rd(0);
        if (ALLFRAGS) rd(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            if (!ALLFRAGS && sub == 0) rd(1);                       // requested under the MFMAs of sub-step 0
#pragma unroll
            for (int sum = NPL - 1; sum >= 0; --sum)
#pragma unroll
                for (int pa = 0; pa <= sum; ++pa)
#pragma unroll
                    for (int i = 0; i < MR; ++i)
#pragma unroll
                        for (int j = 0; j < NR; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[sub][i][pa], fb[sub][j][sum - pa], acc[i][j], 0, 0, 0);
        }
    };

    const int nloc = (q1 - q0 + KG - 1) / KG;   // steps: the same for every group (barriers are block-wide)
    if (DBG == 5) {                             // fixed-cost dissection: launch + prologue (everything it computed stays live)
        if (nloc == 0x7fffff && a_off[0] + b_off[0] + u_off[0] + u_off[1] + (unsigned)tap + am_a + am_b == 12345u) a.y[0] = 1.f;
        return;
    }
    if (q0 < q1) {
#pragma unroll
        for (int s = 0; s < R - 1; ++s) issue(s);
    }
    for (int q = 0; q < nloc; q += R) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (q + j < nloc) {
                s_waitcnt_vm<(R - 2) * PW>();       // my pieces of step q + j have landed (R - 2 later steps stay in flight)
                block_barrier();                    // ... and everybody else's; every wave is done reading the slot of step q + j - 1
                compute(j, (j + R - 1) % R);        // ... while step q + j + R - 1 is requested into that slot
            }
        }
    }
    s_waitcnt_vm<0>();                              // the zero-filling pieces of the steps past the end still write LDS
    if (DBG == 6) { if (acc[0][0][0] == 12345.678f) a.y[0] = 1.f; return; }      // fixed-cost dissection: launch + prologue + loop skeleton
    const int sh_a = amax_shift_of(am_a), sh_b = amax_shift_of(am_b);
    constexpr int EPG = 16 / KG;                    // accumulator registers per 32x32 tile that one K group stores in the epilogue
    if constexpr (KG > 1) {
        // ---- sum the KG accumulator sets through LDS in the fixed order g = 0 .. KG-1; EVERY group forms the sums and stores a share of the rows
        block_barrier();
        float4* red = reinterpret_cast<float4*>(smem);
        constexpr int NQ = MR * NR * 4;
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j)
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4)
                    red[(grp * NQ + (i * NR + j) * 4 + e4) * NT + tid] =
                        make_float4(acc[i][j][4 * e4], acc[i][j][4 * e4 + 1], acc[i][j][4 * e4 + 2], acc[i][j][4 * e4 + 3]);
        __syncthreads();
#pragma unroll
        for (int g = 0; g < KG; ++g)
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const float4 v = red[(g * NQ + (i * NR + j) * 4 + e4) * NT + tid];
                        if (g == 0) { acc[i][j][4 * e4] = v.x; acc[i][j][4 * e4 + 1] = v.y; acc[i][j][4 * e4 + 2] = v.z; acc[i][j][4 * e4 + 3] = v.w; }
                        else { acc[i][j][4 * e4] += v.x; acc[i][j][4 * e4 + 1] += v.y; acc[i][j][4 * e4 + 2] += v.z; acc[i][j][4 * e4 + 3] += v.w; }
                    }
    }
    {           // undo the two operand scales (exact: a power of two)
        const int sh = -(sh_a + sh_b);
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int jj = 0; jj < NR; ++jj)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][jj][e] = __builtin_ldexpf(acc[i][jj][e], sh);
    }
    float mine[KG > 1 ? MR : 1][KG > 1 ? NR : 1][KG > 1 ? EPG : 1];
    if constexpr (KG > 1) {
#pragma unroll
        for (int g = 0; g < KG; ++g)
            if (grp == g) {
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NR; ++j)
#pragma unroll
                        for (int k = 0; k < EPG; ++k) mine[i][j][k] = acc[i][j][g * EPG + k];
            }
    }
    auto share = [&](int i, int j, int e) -> float { if constexpr (KG > 1) return mine[i][j][e]; else return acc[i][j][e]; };
    auto share_add = [&](int i, int j, int e, float v) { if constexpr (KG > 1) mine[i][j][e] += v; else acc[i][j][e] += v; };
    const int rowg = 8 * ((grp * EPG) >> 2);
    // ---- epilogue: D[row][col], col = lane&31 (out channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5); bounds by the descriptor
    float* yout = a.y + (a.splits > 1 ? (long long)z * a.slab : 0ll);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)yout, 0, (int)a.y_bytes, 0x00020000);
    const int col = lane & 31, rq = (lane >> 5) * 4 + rowg;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int k = n0 + (wn * NR + j) * 32 + col;
        const bool kok = k < a.K;
        const float bv = (a.bias != nullptr && kok) ? a.bias[k] : 0.f;
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            const int mb32 = m0 + (wm * MR + i) * 32, mb = mb32 + rq;
            if (a.accumulate && a.splits == 1) {
                float old[EPG];
#pragma unroll
                for (int e = 0; e < EPG; ++e) {
                    const int m = mb + (e & 3) + 8 * (e >> 2);
                    const unsigned off = (kok && m < a.M) ? ((unsigned)m * (unsigned)a.ldy + (unsigned)k) * 4u : kOOB;
                    old[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yr, (int)off, 0, 0));
                }
#pragma unroll
                for (int e = 0; e < EPG; ++e) share_add(i, j, e, old[e]);
            }
#pragma unroll
            for (int e = 0; e < EPG; ++e) {
                const int m = mb + (e & 3) + 8 * (e >> 2);
                const unsigned off = (kok && m < a.M) ? ((unsigned)m * (unsigned)a.ldy + (unsigned)k) * 4u : kOOB;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(share(i, j, e) + bv), yr, (int)off, 0, 0);
            }
        }
    }
    bool bstats_done = false;
    if constexpr (DGRAD && KG == 1) {
        if (a.bstats != nullptr && (a.splits == 1) && a.bn_fast) {
            // Round 5: the same sums with 16-byte accesses (the code of conv_split_kernel.h: same order, bit-identical partials).  The accumulator layout (lane = column, register = row) forces one 4-byte load per element of x
            // and y - 64 (128x64 tile) or 128 (128x128) dependent-ish loads per lane, 11-17 us behind the K loop of the wide 1x1 data gradients.  Here a wave
            // parks its 32 MR x 32 NR tile in LDS (wave-private region of the idle stages, row-major), reads it back as float4 rows, and meets x / y with
            // float4 loads: a quarter of the memory instructions, rows of 128 contiguous bytes per 8 lanes.  Host side: K % 4 == 0, strides multiples of 4,
            // 16-byte aligned tensors, stride-1 launch (no parity order).  Same partials layout [2][mtiles * WGM][K]; summation order differs from the path below.
            constexpr int TR = 32 * MR, TC = 32 * NR, CQ = TC / 4, RSTEP = 64 / CQ, NIT = TR / RSTEP, NB = NIT < 8 ? NIT : 8;
            float* wt = reinterpret_cast<float*>(smem) + (size_t)wave * TR * TC;
            __syncthreads();                                            // every wave is done with the LDS stages
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        wt[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * TC + j * 32 + (lane & 31)] = acc[i][j][e];
            const int c4 = lane % CQ, r0_ = lane / CQ;
            const int kk = n0 + wn * TC + 4 * c4;
            const bool kok4 = kk < a.K;
            const float4 mu4 = kok4 ? *reinterpret_cast<const float4*>(a.bn_mean + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 is4 = kok4 ? *reinterpret_cast<const float4*>(a.bn_invstd + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
            float sg[4] = {0.f, 0.f, 0.f, 0.f}, sgx[4] = {0.f, 0.f, 0.f, 0.f};
            const int mrow0 = m0 + wm * TR;
#pragma unroll
            for (int b0 = 0; b0 < NIT; b0 += NB) {
                float4 xv[NB], yv[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int m = mrow0 + r0_ + RSTEP * (b0 + u);
                    const bool ok = kok4 && m < a.M;
                    xv[u] = ok ? *reinterpret_cast<const float4*>(a.bn_x + (long long)m * a.bn_ldx + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
                    yv[u] = (ok && a.bn_relu) ? *reinterpret_cast<const float4*>(a.bn_y + (long long)m * a.bn_ldy + kk) : make_float4(1.f, 1.f, 1.f, 1.f);
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int row = r0_ + RSTEP * (b0 + u);
                    if (kok4 && mrow0 + row < a.M) {
                        const float4 gv = *reinterpret_cast<const float4*>(wt + row * TC + 4 * c4);
                        const float gs = a.bn_gscale;            // 1 / (1 - p) of a Dropout behind the ReLU (its mask is y > 0 as well), else 1: exact
                        const float g0 = (a.bn_relu && !(yv[u].x > 0.f)) ? 0.f : gv.x * gs, g1 = (a.bn_relu && !(yv[u].y > 0.f)) ? 0.f : gv.y * gs;
                        const float g2 = (a.bn_relu && !(yv[u].z > 0.f)) ? 0.f : gv.z * gs, g3 = (a.bn_relu && !(yv[u].w > 0.f)) ? 0.f : gv.w * gs;
                        sg[0] += g0; sgx[0] += g0 * ((xv[u].x - mu4.x) * is4.x);
                        sg[1] += g1; sgx[1] += g1 * ((xv[u].y - mu4.y) * is4.y);
                        sg[2] += g2; sgx[2] += g2 * ((xv[u].z - mu4.z) * is4.z);
                        sg[3] += g3; sgx[3] += g3 * ((xv[u].w - mu4.w) * is4.w);
                    }
                }
            }
#pragma unroll
            for (int sft = CQ; sft < 64; sft <<= 1)
#pragma unroll
                for (int q = 0; q < 4; ++q) { sg[q] += __shfl_xor(sg[q], sft); sgx[q] += __shfl_xor(sgx[q], sft); }
            if (lane < CQ && kok4) {
                const int nparts = a.mtiles * WGM, part = tile_m * WGM + wm;
                float* o = a.bstats + (long long)part * a.K + kk;
                *reinterpret_cast<float4*>(o) = make_float4(sg[0], sg[1], sg[2], sg[3]);
                *reinterpret_cast<float4*>(o + (long long)nparts * a.K) = make_float4(sgx[0], sgx[1], sgx[2], sgx[3]);
            }
            bstats_done = true;
        }
    }
    if (DGRAD && a.bstats != nullptr && a.splits == 1 && !bstats_done) {
        // BatchNorm-backward partials of the gradient tile just written: sum(g), sum(g * xhat) per channel over the wave's 32*MR rows
        // (layout [2][mtiles * WGM][K], dsrl_bn_bwd_from_stats); K groups: shares meet in LDS, group 0 adds them in the order g = 0 .. KG-1
        const int nparts = a.mtiles * WGM, part = tile_m * WGM + wm;
        float* ex = reinterpret_cast<float*>(smem) + (size_t)KG * MR * NR * 4 * NT * 4;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int k = n0 + (wn * NR + j) * 32 + col;
            const bool kok = k < a.K;
            const float mu = kok ? a.bn_mean[k] : 0.f, is = kok ? a.bn_invstd[k] : 0.f;
            float sg = 0.f, sgx = 0.f;
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                float xv[EPG], yv[EPG];
                const int mb32 = m0 + (wm * MR + i) * 32;
#pragma unroll
                for (int e = 0; e < EPG; ++e) {
                    const int m = mb32 + rq + (e & 3) + 8 * (e >> 2);
                    const bool ok = kok && m < a.M;
                    const long long px = m;
                    xv[e] = ok ? a.bn_x[px * a.bn_ldx + k] : 0.f;
                    yv[e] = (ok && a.bn_relu) ? a.bn_y[px * a.bn_ldy + k] : 1.f;
                }
#pragma unroll
                for (int e = 0; e < EPG; ++e) {
                    const int m = mb32 + rq + (e & 3) + 8 * (e >> 2);
                    if (kok && m < a.M) {
                        const float g = (a.bn_relu && !(yv[e] > 0.f)) ? 0.f : share(i, j, e) * a.bn_gscale;
                        sg += g; sgx += g * ((xv[e] - mu) * is);
                    }
                }
            }
            sg += __shfl_xor(sg, 32); sgx += __shfl_xor(sgx, 32);
            if (KG == 1) {
                if (lane < 32 && kok) {
                    float* o = a.bstats + (long long)part * a.K + k;
                    o[0] = sg; o[(long long)nparts * a.K] = sgx;
                }
            } else if (lane < 32) {
                ex[(((grp * NW + wave) * NR + j) * 2 + 0) * 32 + lane] = sg;
                ex[(((grp * NW + wave) * NR + j) * 2 + 1) * 32 + lane] = sgx;
            }
        }
        if constexpr (KG > 1) {
            __syncthreads();
            if (grp == 0 && lane < 32) {
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const int k = n0 + (wn * NR + j) * 32 + col;
                    float sg = 0.f, sgx = 0.f;
#pragma unroll
                    for (int g = 0; g < KG; ++g) {
                        sg += ex[(((g * NW + wave) * NR + j) * 2 + 0) * 32 + lane];
                        sgx += ex[(((g * NW + wave) * NR + j) * 2 + 1) * 32 + lane];
                    }
                    if (k < a.K) {
                        float* o = a.bstats + (long long)part * a.K + k;
                        o[0] = sg; o[(long long)nparts * a.K] = sgx;
                    }
                }
            }
        }
    }
    const bool stats_wave = !(KG > 1 && grp > 0);    // the forward statistics below are taken by group 0 from the whole tile (every wave stays for the block barrier)
    // ---- BatchNorm partials of the tile this block just wrote: (n, mean, M2) per output channel over the wave's 32*MR rows; layout [3][mtiles * WGM][K]
    if (a.stats != nullptr && a.splits == 1) {
        // Round 5: ONE partial per block tile.  The WGM wave rows of a tile used to leave one partial each (128 row blocks for an M = 4096 tensor on 64x64
        // tiles), and every block of the BatchNorm kernel merges all of them for its 32 channels before it streams a row: 2.9 us of a 6.9 us launch at
        // 128 partials, 1.1-1.4 us at 64 (tools/bn_prologue_probe.py).  The wave rows meet in LDS and wave row 0 merges them in the order wm = 1 .. WGM-1
        // (Chan's update: exact counts, the same formulas the BatchNorm kernels use).  Layout [3][mtiles][K].
        const int nparts = a.mtiles, part = tile_m;
        float* mg = reinterpret_cast<float*>(smem) + (KG > 1 ? (size_t)KG * MR * NR * 4 * NT * 4 : 0);   // [WGM][WGN * NR][3][32], behind the K-group reduction area
        float sn[NR], sm[NR], sq[NR];
        if (stats_wave) {
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int k = n0 + (wn * NR + j) * 32 + col;
                const bool kok = k < a.K;
                const float bv = (a.bias != nullptr && kok) ? a.bias[k] : 0.f;
                float n = 0.f, sum = 0.f;
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (m0 + (wm * MR + i) * 32 + (lane >> 5) * 4 + (e & 3) + 8 * (e >> 2) < a.M) { n += 1.f; sum += acc[i][j][e] + bv; }
                n += __shfl_xor(n, 32); sum += __shfl_xor(sum, 32);
                const float mean = n > 0.f ? sum / n : 0.f;
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (m0 + (wm * MR + i) * 32 + (lane >> 5) * 4 + (e & 3) + 8 * (e >> 2) < a.M) { const float d = acc[i][j][e] + bv - mean; q += d * d; }
                q += __shfl_xor(q, 32);
                sn[j] = n; sm[j] = mean; sq[j] = q;
                if (WGM > 1 && wm > 0 && lane < 32) {
                    float* o = mg + ((wm * (WGN * NR) + wn * NR + j) * 3) * 32 + lane;
                    o[0] = n; o[32] = mean; o[64] = q;
                }
            }
        }
        if constexpr (WGM > 1) __syncthreads();
        if (stats_wave && wm == 0 && lane < 32) {
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int k = n0 + (wn * NR + j) * 32 + col;
                float na = sn[j], ma = sm[j], qa = sq[j];
#pragma unroll
                for (int w = 1; w < WGM; ++w) {
                    const float* o = mg + ((w * (WGN * NR) + wn * NR + j) * 3) * 32 + lane;
                    const float nb = o[0], mb = o[32], qb = o[64];
                    if (nb > 0.f) {
                        const float nt = na + nb, d = mb - ma;
                        ma += d * (nb / nt);
                        qa += qb + d * d * (na * nb / nt);
                        na = nt;
                    }
                }
                if (k < a.K) {
                    float* o = a.stats + (long long)part * a.K + k;
                    o[0] = na; o[(long long)nparts * a.K] = ma; o[2ll * nparts * a.K] = qa;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ producers of planes
// planes of a pixel-major fp32 tensor: hi[p][c] = f16(x * 2^e), lo[p][c] = f16(x * 2^e - hi), e from the tensor's amax record - the terms
// conv_igemm_split_kernel forms on the fly (same instructions: ldexp, v_cvt_pk_f16_f32, v_fma_mix_f32 residual).  8 channels per thread.
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, int ld, long long P, int C8, const unsigned* __restrict__ rec,
                                                           uint4* __restrict__ hi, uint4* __restrict__ lo, int ldp8) {
    const int sh = amax_shift(rec);
    const long long n = P * C8, stride = (long long)gridDim.x * 256;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) {
        const long long p = e / C8;
        const int c8 = (int)(e - p * C8);
        const float4 v0 = *reinterpret_cast<const float4*>(x + p * ld + c8 * 8), v1 = *reinterpret_cast<const float4*>(x + p * ld + c8 * 8 + 4);
        float r0[4] = {__builtin_ldexpf(v0.x, sh), __builtin_ldexpf(v0.y, sh), __builtin_ldexpf(v0.z, sh), __builtin_ldexpf(v0.w, sh)};
        float r1[4] = {__builtin_ldexpf(v1.x, sh), __builtin_ldexpf(v1.y, sh), __builtin_ldexpf(v1.z, sh), __builtin_ldexpf(v1.w, sh)};
        const f16x4 t0 = Plane<true>::cvt(r0), t1 = Plane<true>::cvt(r1);
        const uint2 h0 = __builtin_bit_cast(uint2, t0), h1 = __builtin_bit_cast(uint2, t1);
        hi[p * ldp8 + c8] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        if (lo != nullptr) {
            Plane<true>::residual(r0, t0); Plane<true>::residual(r1, t1);
            const uint2 l0 = __builtin_bit_cast(uint2, Plane<true>::cvt(r0)), l1 = __builtin_bit_cast(uint2, Plane<true>::cvt(r1));
            lo[p * ldp8 + c8] = make_uint4(l0.x, l0.y, l1.x, l1.y);
        }
    }
}

// Every conv filter of the model as planes, forward layout [K][RS][C] and transposed [C][RS][K], in one launch per training step behind the amax
// pass (dsrl_conv2d_filters_amax_batched).  Table rows of kWtRow int64 as for weight_split_batched_kernel:
// {w, wt_planes, K, Kp (unused), RS, C, first tile, tiles along C, amax record, w_planes}; each planes pointer addresses the first plane, the second
// lies dsrl_planes_lo_offset(elements) bytes behind it.  K % 8 == 0 and C % 8 == 0 (host-checked).
constexpr int kWtRowP = 10, kWtTilesPerBlockP = 4;
__global__ __launch_bounds__(256) void filter_planes_batched_kernel(const long long* __restrict__ table, int n, long long total_tiles) {
    // software-pipelined like weight_split_batched_kernel (round 5): the next tile's loads are in flight while this one is split and written
    __shared__ float tile[32][33];
    const long long b0 = (long long)blockIdx.x * kWtTilesPerBlockP;
    int lo_ = 0, hi_ = n - 1;
    while (lo_ < hi_) {
        const int mid = (lo_ + hi_ + 1) >> 1;
        if (table[mid * kWtRowP + 6] <= b0) lo_ = mid; else hi_ = mid - 1;
    }
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int row = threadIdx.x >> 2, g8 = (threadIdx.x & 3) * 8;       // 128 threads write 32 rows x 4 units of 8 elements
    struct Tile { const long long* e; int tap, k0, c0; };
    auto decode = [&](long long b) -> Tile {
        while (lo_ + 1 < n && table[(lo_ + 1) * kWtRowP + 6] <= b) ++lo_;
        const long long* e = table + lo_ * kWtRowP;
        const int K = (int)e[2], ct = (int)e[7], kt = (K + 31) / 32;
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
    for (int u = 0; u < kWtTilesPerBlockP; ++u) {
        if (b0 + u >= total_tiles) break;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) tile[ty + 8 * i][tx] = v[i];
        const int sh = amax_shift_of(am);
        __syncthreads();
        const Tile me = cur;
        if (u + 1 < kWtTilesPerBlockP && b0 + u + 1 < total_tiles) {
            cur = decode(b0 + u + 1);
            fetch(cur, v);
            if (cur.e != am_of) { am = amax_fetch(reinterpret_cast<const unsigned*>(cur.e[8])); am_of = cur.e; }
        }
        char* wtp = reinterpret_cast<char*>(me.e[1]);
        char* wp = reinterpret_cast<char*>(me.e[9]);
        const int K = (int)me.e[2], RS = (int)me.e[4], C = (int)me.e[5], tap = me.tap, k0 = me.k0, c0 = me.c0;
        const long long elems = (long long)K * RS * C, lo_off = planes_lo_offset(elems);
        if (threadIdx.x < 128) {
            float r0[4], r1[4];
            {   // forward layout: row = out channel, 8 consecutive input channels
                const int k = k0 + row, c = c0 + g8;
                if (wp != nullptr && k < K && c < C) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { r0[q] = __builtin_ldexpf(tile[row][g8 + q], sh); r1[q] = __builtin_ldexpf(tile[row][g8 + 4 + q], sh); }
                    const f16x4 t0 = Plane<true>::cvt(r0), t1 = Plane<true>::cvt(r1);
                    Plane<true>::residual(r0, t0); Plane<true>::residual(r1, t1);
                    const uint2 h0 = __builtin_bit_cast(uint2, t0), h1 = __builtin_bit_cast(uint2, t1);
                    const uint2 l0 = __builtin_bit_cast(uint2, Plane<true>::cvt(r0)), l1 = __builtin_bit_cast(uint2, Plane<true>::cvt(r1));
                    const long long off = (((long long)k * RS + tap) * C + c) * 2;
                    *reinterpret_cast<uint4*>(wp + off) = make_uint4(h0.x, h0.y, h1.x, h1.y);
                    *reinterpret_cast<uint4*>(wp + lo_off + off) = make_uint4(l0.x, l0.y, l1.x, l1.y);
                }
            }
            {   // transposed: row = input channel, 8 consecutive out channels
                const int c = c0 + row, k = k0 + g8;
                if (wtp != nullptr && c < C && k < K) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { r0[q] = __builtin_ldexpf(tile[g8 + q][row], sh); r1[q] = __builtin_ldexpf(tile[g8 + 4 + q][row], sh); }
                    const f16x4 t0 = Plane<true>::cvt(r0), t1 = Plane<true>::cvt(r1);
                    Plane<true>::residual(r0, t0); Plane<true>::residual(r1, t1);
                    const uint2 h0 = __builtin_bit_cast(uint2, t0), h1 = __builtin_bit_cast(uint2, t1);
                    const uint2 l0 = __builtin_bit_cast(uint2, Plane<true>::cvt(r0)), l1 = __builtin_bit_cast(uint2, Plane<true>::cvt(r1));
                    const long long off = (((long long)c * RS + tap) * K + k) * 2;
                    *reinterpret_cast<uint4*>(wtp + off) = make_uint4(h0.x, h0.y, h1.x, h1.y);
                    *reinterpret_cast<uint4*>(wtp + lo_off + off) = make_uint4(l0.x, l0.y, l1.x, l1.y);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
// R_: ring depth with two planes (f16x3); with ONE plane (f16x1: a.planes == 1) a slot is half as large and the ring takes R1_ slots
#define DSRL_PLANES_LAUNCH_N(MR_, NR_, WGM_, WGN_, KG_, NPL_, R_)                                                                          \
    {                                                                                                                                       \
        constexpr int threads = 64 * WGM_ * WGN_ * KG_;                                                                                     \
        constexpr size_t slot = (size_t)(32 * MR_ * WGM_ + 32 * NR_ * WGN_) * NPL_ * 64;                                                   \
        constexpr size_t redb = KG_ > 1 ? (size_t)KG_ * MR_ * NR_ * 4 * (threads / KG_) * 16 + 8192 : 0;                                    \
        constexpr size_t tileb = (KG_ == 1 && DGRAD) ? (size_t)(32 * MR_ * WGM_) * (32 * NR_ * WGN_) * 4 : 0;      /* fast BatchNorm-sum epilogue: the fp32 tile */ \
        constexpr size_t lds0 = slot * R_ * KG_ > redb ? slot * R_ * KG_ : redb;                                                            \
        constexpr size_t lds = (tileb > lds0 && tileb <= 160 * 1024) ? tileb : lds0;                                                        \
        static_assert(lds <= 160 * 1024, "LDS");                                                                                            \
        static const hipError_t attr = hipFuncSetAttribute((const void*)conv_planes_kernel<MR_, NR_, WGM_, WGN_, KG_, NPL_, DGRAD, R_>,      \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
        (void)attr;                                                                                                                         \
        hipLaunchKernelGGL((conv_planes_kernel<MR_, NR_, WGM_, WGN_, KG_, NPL_, DGRAD, R_>), grid, dim3(threads), lds, st, a);               \
        return launch_status("conv_planes_kernel");                                                                                         \
    }
#define DSRL_PLANES_LAUNCH2(MR_, NR_, WGM_, WGN_, KG_, R_, R1_) { if (a.planes == 1) DSRL_PLANES_LAUNCH_N(MR_, NR_, WGM_, WGN_, KG_, 1, R1_) else DSRL_PLANES_LAUNCH_N(MR_, NR_, WGM_, WGN_, KG_, 2, R_) }
#define DSRL_PLANES_LAUNCH(MR_, NR_, WGM_, WGN_, KG_, R_) DSRL_PLANES_LAUNCH2(MR_, NR_, WGM_, WGN_, KG_, R_, R_)

// cfg: enum TileCfg of conv_igemm.hip {T128x128, T256x64, T256x32, T64x64, T128x64, T64x128, T128x32, T256x128, T256x256}
template <bool DGRAD>
static int launch_planes_t(const ConvArgs& a, int cfg, hipStream_t st) {
    const dim3 grid((unsigned)(a.mtiles * a.ntiles), 1u, (unsigned)(a.kg > 1 ? 1 : a.splits));
    const int kg = a.kg > 1 ? a.kg : 1;
    static const int r = env_int("DSRL_PLANES_R", 0);       // ring depth override (tools/planes_bench.py), read once
    switch (cfg) {
        case 3:     // 64x64
#ifdef DSRL_PLANES_ABLATION      // timing-only builds (wrong results): make CXXFLAGS+=-DDSRL_PLANES_ABLATION, profiles/round4_planes_kernel_ab.txt
            if (const int dbg = env_int("DSRL_PLANES_DBG", 0)) {         // timing-only ablations (wrong results): 1 = no MFMA / fragment reads, 2 = no DMA, 3 = neither
                if constexpr (!DGRAD) {
                    constexpr size_t lds = 128 * 1024;
                    hipFuncSetAttribute((const void*)conv_planes_kernel<1, 1, 2, 2, 4, 2, false, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    hipFuncSetAttribute((const void*)conv_planes_kernel<1, 1, 2, 2, 4, 2, false, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    hipFuncSetAttribute((const void*)conv_planes_kernel<1, 1, 2, 2, 4, 2, false, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    hipFuncSetAttribute((const void*)conv_planes_kernel<1, 1, 2, 2, 4, 2, false, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    hipFuncSetAttribute((const void*)conv_planes_kernel<1, 1, 2, 2, 4, 2, false, 2, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    hipFuncSetAttribute((const void*)conv_planes_kernel<1, 1, 2, 2, 4, 2, false, 2, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    if (kg == 4 && dbg == 1) hipLaunchKernelGGL((conv_planes_kernel<1, 1, 2, 2, 4, 2, false, 2, 1>), grid, dim3(1024), lds, st, a);
                    if (kg == 4 && dbg == 2) hipLaunchKernelGGL((conv_planes_kernel<1, 1, 2, 2, 4, 2, false, 2, 2>), grid, dim3(1024), lds, st, a);
                    if (kg == 4 && dbg == 3) hipLaunchKernelGGL((conv_planes_kernel<1, 1, 2, 2, 4, 2, false, 2, 3>), grid, dim3(1024), lds, st, a);
                    if (kg == 4 && dbg == 4) hipLaunchKernelGGL((conv_planes_kernel<1, 1, 2, 2, 4, 2, false, 2, 4>), grid, dim3(1024), lds, st, a);
                    if (kg == 4 && dbg == 5) hipLaunchKernelGGL((conv_planes_kernel<1, 1, 2, 2, 4, 2, false, 2, 5>), grid, dim3(1024), lds, st, a);
                    if (kg == 4 && dbg == 6) hipLaunchKernelGGL((conv_planes_kernel<1, 1, 2, 2, 4, 2, false, 2, 6>), grid, dim3(1024), lds, st, a);
                    if (kg == 4) return launch_status("conv_planes_kernel<debug>");
                    if (kg == 2 && dbg == 1) {          // DMA only, two K groups, ring depth r
                        hipFuncSetAttribute((const void*)conv_planes_kernel<1, 1, 2, 2, 2, 2, false, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                        hipFuncSetAttribute((const void*)conv_planes_kernel<1, 1, 2, 2, 2, 2, false, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                        hipFuncSetAttribute((const void*)conv_planes_kernel<1, 1, 2, 2, 2, 2, false, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                        if (r == 2) hipLaunchKernelGGL((conv_planes_kernel<1, 1, 2, 2, 2, 2, false, 2, 1>), grid, dim3(512), lds, st, a);
                        else if (r == 3) hipLaunchKernelGGL((conv_planes_kernel<1, 1, 2, 2, 2, 2, false, 3, 1>), grid, dim3(512), lds, st, a);
                        else hipLaunchKernelGGL((conv_planes_kernel<1, 1, 2, 2, 2, 2, false, 4, 1>), grid, dim3(512), lds, st, a);
                        return launch_status("conv_planes_kernel<debug>");
                    }
                }
            }
#endif
            if (kg == 4) DSRL_PLANES_LAUNCH2(1, 1, 2, 2, 4, 2, 4)
            if (kg == 2) { if (r == 2) DSRL_PLANES_LAUNCH(1, 1, 2, 2, 2, 2) else if (r == 3) DSRL_PLANES_LAUNCH(1, 1, 2, 2, 2, 3) else DSRL_PLANES_LAUNCH(1, 1, 2, 2, 2, 4) }
            if (r == 2) DSRL_PLANES_LAUNCH(1, 1, 2, 2, 1, 2) else DSRL_PLANES_LAUNCH(1, 1, 2, 2, 1, 3)
        case 4:     // 128x64
            if (kg == 2) { if (r == 2) DSRL_PLANES_LAUNCH(2, 1, 2, 2, 2, 2) else DSRL_PLANES_LAUNCH(2, 1, 2, 2, 2, 3) }
            if (kg == 1) { if (r == 2) DSRL_PLANES_LAUNCH(2, 1, 2, 2, 1, 2) else DSRL_PLANES_LAUNCH(2, 1, 2, 2, 1, 3) }
            break;
        case 5:     // 64x128
            if (kg == 2) { if (r == 2) DSRL_PLANES_LAUNCH(1, 2, 2, 2, 2, 2) else DSRL_PLANES_LAUNCH(1, 2, 2, 2, 2, 3) }
            if (kg == 1) { if (r == 2) DSRL_PLANES_LAUNCH(1, 2, 2, 2, 1, 2) else DSRL_PLANES_LAUNCH(1, 2, 2, 2, 1, 3) }
            break;
        case 0:     // 128x128
            if (kg == 1) { if (r == 3) DSRL_PLANES_LAUNCH(2, 2, 2, 2, 1, 3) else DSRL_PLANES_LAUNCH2(2, 2, 2, 2, 1, 2, 4) }
            break;
        case 1:     // 256x64
            if (kg == 1) { if (r == 3) DSRL_PLANES_LAUNCH(2, 2, 4, 1, 1, 3) else DSRL_PLANES_LAUNCH(2, 2, 4, 1, 1, 2) }
            break;
        case 7:     // 256x128, 8 waves
            if (kg == 1) { if (r == 2) DSRL_PLANES_LAUNCH(2, 2, 4, 2, 1, 2) else DSRL_PLANES_LAUNCH(2, 2, 4, 2, 1, 3) }
            break;
        case 8:     // 256x256, 8 waves
            if (kg == 1) DSRL_PLANES_LAUNCH2(4, 2, 2, 4, 1, 2, 3)
            break;
        default: break;
    }
    set_error("conv_planes_kernel: no build for tile configuration %d with %d K groups", cfg, kg);
    return DSRL_E_UNSUPPORTED;
}
#undef DSRL_PLANES_LAUNCH
#undef DSRL_PLANES_LAUNCH2
#undef DSRL_PLANES_LAUNCH_N

bool planes_cfg_supported(int cfg, int kg) {
    switch (cfg) {
        case 3: return kg == 1 || kg == 2 || kg == 4;
        case 4: case 5: return kg == 1 || kg == 2;
        case 0: case 1: case 7: case 8: return kg <= 1;
        default: return false;
    }
}
int launch_planes_igemm(const ConvArgs& a, int cfg, bool dgrad, hipStream_t st) {
    return dgrad ? launch_planes_t<true>(a, cfg, st) : launch_planes_t<false>(a, cfg, st);
}

}  // namespace dsrl

using namespace dsrl;

extern "C" size_t dsrl_planes_lo_offset(int64_t elems) { return (size_t)planes_lo_offset(elems); }
extern "C" size_t dsrl_planes_bytes(int64_t elems, int nplanes) { return (size_t)planes_lo_offset(elems) * (size_t)(nplanes > 1 ? 2 : 1); }

extern "C" int dsrl_split_planes(const float* x, int ld, int64_t P, int C, const uint32_t* amax, void* planes, int nplanes, dsrl_stream_t stream) {
    DSRL_REQUIRE(x && amax && planes && P > 0 && C > 0 && (nplanes == 1 || nplanes == 2), DSRL_E_BADARG, "split_planes: bad arguments");
    DSRL_REQUIRE(C % 8 == 0 && ld % 8 == 0 && ld >= C && ((uintptr_t)x % 16) == 0 && ((uintptr_t)planes % 16) == 0, DSRL_E_UNSUPPORTED,
                 "split_planes: C (%d) and ld (%d) must be multiples of 8, x and planes 16-byte aligned", C, ld);
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    const long long n = (long long)P * (C / 8);
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>(ceil_div(n, 256 * 2), 4096));
    char* hi = (char*)planes;
    hipLaunchKernelGGL(split_planes_kernel, dim3(grid), dim3(256), 0, st, x, ld, (long long)P, C / 8, (const unsigned*)amax, (uint4*)hi,
                       nplanes > 1 ? (uint4*)(hi + planes_lo_offset((long long)P * ld)) : (uint4*)nullptr, ld / 8);
    return launch_status("split_planes_kernel");
}

extern "C" int dsrl_conv2d_filter_planes_batched(const int64_t* table, int n, int64_t total_tiles, dsrl_stream_t stream) {
    DSRL_REQUIRE(table && n > 0 && total_tiles > 0 && total_tiles < (1ll << 31), DSRL_E_BADARG, "conv2d_filter_planes_batched: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (int e = bind_stream_device(st)) return e;
    hipLaunchKernelGGL(filter_planes_batched_kernel, dim3((unsigned)ceil_div(total_tiles, (int64_t)kWtTilesPerBlockP)), dim3(256), 0, st,
                       (const long long*)table, n, (long long)total_tiles);
    return launch_status("filter_planes_batched_kernel");
}
