// conv_igemm_split_kernel: the split-precision implicit-GEMM conv forward / data gradient (bf16x3, bf16x6, f16x3, f16x1), shared by
// conv_igemm.hip (the tile / K-group builds of rounds 1-4) and conv_sk.hip (round 5: 128x128 tiles with two K groups and split-K across CUs
// with an in-kernel last-arriver reduction).
#pragma once
#include "common.h"
#include "conv_common.h"

namespace dsrl {

// ------------------------------------------------------------------------------------------------ split-precision forward / dgrad
// bf16x3 (NPL = 2) / bf16x6 (NPL = 3) implicit GEMM, software-pipelined for two waves per SIMD:
//   * a 32-channel chunk is fetched as before (two float4 per staged row: k = 4*c4.. and 16 + 4*c4..) into one of two register
//     sets and consumed as two 16-deep half-steps, each with its own LDS stage (two stages, ping-pong);
//   * one half-step = fragment reads of the current stage, then its MFMAs with the split/convert + LDS writes of the NEXT
//     half-step interleaved between them (one convert step = one bf16 plane of one staged float4), one barrier;
//   * LDS rows are 32 B (16 bf16) per plane, unpadded, with the two 16-byte halves swapped on rows with bit 3 set: the b64
//     writes (8 rows x 32 B per 32 lanes) and the b128 fragment reads (16 rows per 16 lanes) are both bank-conflict-free;
//   * the loads of chunk q+2 are issued when the registers of chunk q are drained: two half-steps of flight time;
//   * KG > 1 ("K groups", for grids of at most ~1.5 tiles per CU): the block has KG groups of 4 waves, group g runs the same pipeline
//     on chunks g, g+KG, ... with its own two LDS stages, and the KG accumulator sets are summed through LDS in a fixed order at the
//     end - the latency-hiding of split-K (more waves per SIMD) without slab traffic or a reduce launch.
// ARITH: 0 = bf16 terms; 1 = f16x3, both operands split here; 2 = f16x3 with the filter operand already split (ConvArgs::w then points at the
// filter in "plane" form - per 4 consecutive channels 4 fp16 first terms followed by the 4 second terms, 16 bytes for 16 bytes of fp32, same
// indexing - written once per training step by weight_split_batched_kernel with the scale of a.amax_b): its staged float4 are stored to
// LDS as they come, which removes half of the split work of a chunk - work that every one of the M / BM row tiles used to repeat.
// WGM * WGN = 4 waves (256 threads per K group) or, round 3, 8 waves (512 threads, KG = 1: the 256x128 and 256x256 tiles - a wave still owns
// MR x NR tiles of 32x32, but the block stages 0.75x / 0.5x the bytes of 128x128 tiles per multiply-add, through LDS and from L2).
// Full-step LDS stages (round 4, fp16 arithmetics): a stage holds a whole 32-channel step (64-byte rows) instead of a 16-channel half, so a K group
// crosses ONE block barrier per step and the fragment reads of the second half overlap the MFMAs of the first (profiles/round4_staging_ablation.txt:
// with no operand staging at all the half-step loop still took 27.6 of 30.9 us).  Same MFMAs in the same order.  -DDSRL_FULLSTEP=0: half-step stages.
#ifndef DSRL_FULLSTEP
#define DSRL_FULLSTEP 1
#endif
constexpr bool kFullStep = DSRL_FULLSTEP != 0;
// COOP (round 5, conv_sk.hip): split-K across workgroups reduced INSIDE the launch.  The `splits` blocks of a tile each run their share of the
// (tap, chunk) loop; a block publishes its (unscaled) partial tile with write-through stores, drains them, takes a ticket; the last arriver reads the
// other partials back (sc1 loads: L2-served, never a stale L1 line), adds them in the fixed order z = 0 .. splits-1 - its own contribution from
// registers, same values - and runs the ordinary epilogue on y, so bias, accumulate, the BatchNorm partials of both passes and the amax-free
// hand-over all work as in an unsplit launch and no reduce launch follows.  Same sums in the same order as slabs + splitk_reduce_kernel.
// (MI355X_MICROARCH.md, "Valid forms": every hand-off store sc1, every storing wave waits vmcnt(0), a workgroup barrier, ONE lane's agent-scope add,
// the block whose add returns splits-1 loads behind a barrier that lane joins, every hand-off load sc1; one 512-thread workgroup per CU.)
template <int MR, int NR, int WGM, int WGN, bool DGRAD, int NPL, int KG = 1, int ARITH = 0, bool STR1 = false, bool COOP = false>
__global__ __launch_bounds__(64 * WGM * WGN * KG, KG == 1 ? 2 : 1)
void conv_igemm_split_kernel(const ConvArgs a) {
    // STR1: the launch has stride 1 (110 of the step's 114 data gradients): no divisibility tests, no parity order - the gather is a forward conv's
    const int stride = STR1 ? 1 : a.stride, par = STR1 ? 0 : a.par;
    constexpr bool F16 = ARITH != 0, PREB = ARITH == 2;
    constexpr int NT = 64 * WGM * WGN, RP = NT / 4;          // threads of one K group; rows per staging pass (4 lanes per row)
    static_assert(NT == 256 || (NT == 512 && KG == 1), "4 waves per K group, or one group of 8 waves");
    static_assert(!F16 || NPL <= 2, "f16x3 carries two fp16 terms per operand, f16x1 one");
    using PT = Plane<F16>;
    using pl4 = typename PT::v4; using pl8 = typename PT::v8;
    // f16x3 operand scales: both records are requested before anything else and consumed behind the first operand loads (in-order return:
    // waiting for them does not wait for the tiles)
    const unsigned am_a = F16 ? amax_fetch(a.amax_a) : 0u, am_b = F16 ? amax_fetch(a.amax_b) : 0u;
    int sh_a = 0, sh_b = 0;
    constexpr int BM = 32 * MR * WGM, BN = 32 * NR * WGN;
    constexpr int A_IT = (BM + RP - 1) / RP, B_IT = (BN + RP - 1) / RP;
    constexpr int NV = A_IT + B_IT;
    constexpr bool FS = kFullStep && F16;                            // a stage = a whole 32-channel step
    constexpr int ROWB = FS ? 64 : 32;
    constexpr int STAGE = (BM + BN) * NPL * ROWB;                     // bytes
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int grp = KG > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;
    char* const S0 = reinterpret_cast<char*>(smem) + grp * 2 * STAGE;
    char* const S1 = S0 + STAGE;

    const int tid = threadIdx.x & (NT - 1);     // thread within its K group
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int tile = a.xcd_remap ? xcd_contiguous(blockIdx.x, a.mtiles * a.ntiles) : blockIdx.x;
    const int tile_m = fast_div(tile, a.mNT, a.sNT);           // the prologue runs on every wave of the block at once: no integer divides in it
    const int m0 = tile_m * BM, n0 = (tile - tile_m * a.ntiles) * BN, z = blockIdx.z;
    const int HoWo = a.Ho * a.Wo;

    const int c4 = tid & 3, r0 = tid >> 2;
    const bool b_rows = (BN % RP == 0) || r0 < BN;                   // BN < rows per pass: only the first waves stage filter rows
    int a_n[A_IT], a_h[A_IT], a_w[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + r0 + RP * i;
        a_ok[i] = m < a.M && (BM % RP == 0 || r0 + RP * i < BM);
        const int mm = a_ok[i] ? ((DGRAD && par) ? dgrad_pix(a, m) : m) : 0;
        const int n = fast_div(mm, a.mHW, a.sHW), rem = mm - n * HoWo;
        const int ho = fast_div(rem, a.mW, a.sW), wo = rem - ho * a.Wo;
        a_n[i] = n;
        if (DGRAD) { a_h[i] = ho + a.pad; a_w[i] = wo + a.pad; }
        else { a_h[i] = ho * stride - a.pad; a_w[i] = wo * stride - a.pad; }
    }
    const int RS = a.R * a.S;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)a.w_bytes, 0x00020000);
    unsigned b_off[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int k = n0 + r0 + RP * i;
        b_off[i] = (k < a.K && b_rows) ? (unsigned)k * (unsigned)(RS * a.C) * 4u : kOOB;
    }

    // ---- taps that touch at least one in-bounds input pixel for this tile (block-uniform)
    unsigned long long tapmask = 0ull;
    if (a.R * a.S == 1 && a.pad == 0) {
        tapmask = 1ull;                         // 1x1 without padding: the one tap is always in bounds
    } else {
        const int mf = m0, ml = min(m0 + BM, a.M) - 1;
        const int nf = fast_div(mf, a.mHW, a.sHW), nl = fast_div(ml, a.mHW, a.sHW);
        int hf = 0, hl = a.Ho - 1, wf = 0, wl = a.Wo - 1;
        if (nf == nl) {
            hf = fast_div(mf - nf * HoWo, a.mW, a.sW); hl = fast_div(ml - nl * HoWo, a.mW, a.sW);
            if (hf == hl) { wf = (mf - nf * HoWo) - hf * a.Wo; wl = (ml - nl * HoWo) - hl * a.Wo; }
        }
        const int pcls = (DGRAD && par) ? (m0 / BM) % (par * par) : 0;  // parity class of this tile
        const int ph = (DGRAD && par) ? pcls / par : 0, pw = (DGRAD && par) ? pcls - ph * par : 0;
        if (DGRAD && par) { hf = 0; hl = a.Ho - 1; wf = 0; wl = a.Wo - 1; }     // the row range of a parity-ordered tile is not an interval: no bounds pruning
        for (int r = 0; r < a.R; ++r)
            for (int s = 0; s < a.S; ++s) {
                bool act;
                if (DGRAD) {
                    act = (hl + a.pad - r * a.dil >= 0) && (hf + a.pad - r * a.dil <= (a.H - 1) * stride) &&
                          (wl + a.pad - s * a.dil >= 0) && (wf + a.pad - s * a.dil <= (a.W - 1) * stride);
                    if (par) act = act && ((ph + a.pad - r * a.dil) % par == 0) && ((pw + a.pad - s * a.dil) % par == 0);
                } else {
                    act = (hl * stride - a.pad + r * a.dil >= 0) && (hf * stride - a.pad + r * a.dil <= a.H - 1) &&
                          (wl * stride - a.pad + s * a.dil >= 0) && (wf * stride - a.pad + s * a.dil <= a.W - 1);
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

    // iterator over (tap, channel chunk), positioned at this group's first chunk q0 + grp
    int cc = 0, tap = 0, pos = q0;
    unsigned long long rem_mask = tapmask;
    if (q0 < q1) {
        if (q0 > 0) {                           // split-K launches only
            int skip = q0 / a.cchunks;
            cc = q0 - skip * a.cchunks;
            while (skip--) rem_mask &= rem_mask - 1;
        }
        tap = __builtin_ctzll(rem_mask);
    }
    auto advance = [&](int n) {                 // wave-uniform; only ever asked to step onto an existing chunk.  O(taps crossed), not O(n): a K-group
        cc += n; pos += n;                      // block steps KG chunks at a time, and this sits between a wave's last MFMA and its barrier
        while (cc >= a.cchunks) { cc -= a.cchunks; rem_mask &= rem_mask - 1; tap = __builtin_ctzll(rem_mask); }
    };
    unsigned a_off[A_IT];
    const unsigned inv_s = 65536u / (unsigned)a.S + 1u;        // t / S without a divide for t < 64, S <= 8 (checked exhaustively): set_tap runs inside the K loop
    auto set_tap = [&](int t) {
        const int r = a.S <= 8 ? (int)(((unsigned)t * inv_s) >> 16) : t / a.S, s = t - r * a.S;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int hi, wi; bool ok = a_ok[i];
            if (DGRAD) {
                const int hn = a_h[i] - r * a.dil, wn_ = a_w[i] - s * a.dil;
                hi = hn / stride; wi = wn_ / stride;
                ok = ok && hn >= 0 && wn_ >= 0 && hi * stride == hn && wi * stride == wn_ && hi < a.H && wi < a.W;
            } else {
                hi = a_h[i] + r * a.dil; wi = a_w[i] + s * a.dil;
                ok = ok && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
            }
            a_off[i] = ok ? (unsigned)((a_n[i] * a.H + hi) * a.W + wi) * (unsigned)a.ldx * 4u : kOOB;
        }
    };
    // register sets: R[v][half], v < A_IT: pixel rows, v >= A_IT: filter rows
    float4 R0[NV][2], R1[NV][2];
#pragma unroll
    for (int v = 0; v < NV; ++v) { R0[v][0] = R0[v][1] = R1[v][0] = R1[v][1] = make_float4(0.f, 0.f, 0.f, 0.f); }
    // Offsets are added with unsigned saturation: a padding pixel (a_off = kOOB) in the channel tail (coff = kOOB) must stay out of range instead of
    // wrapping to offset 0 (its product is multiplied by a zero filter value, but a NaN at x[0] would have leaked into border outputs).
    auto gload = [&](float4 (*R)[2], int t, int ch, unsigned dead) {          // dead: 0, or kOOB = every load of the set out of range (zeros)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int c = ch * 32 + hf * 16 + c4 * 4;
            const unsigned coff = (c < a.C ? (unsigned)c * 4u : kOOB) | dead;   // channel tail of the last chunk reads as zeros
            const unsigned woff = __builtin_elementwise_add_sat(coff, (unsigned)(t * a.C) * 4u);
#pragma unroll
            for (int i = 0; i < A_IT; ++i) R[i][hf] = buf_load4(xr, __builtin_elementwise_add_sat(a_off[i], coff));
#pragma unroll
            for (int i = 0; i < B_IT; ++i) R[A_IT + i][hf] = buf_load4(wr, __builtin_elementwise_add_sat(b_off[i], woff));
        }
    };
    int nextq = q0 + grp, tap_set = -1;         // next chunk of this group; tap whose a_off[] is current
    // Every call issues the same NV * 2 loads - past the end of the group's (or the block's) chunks with out-of-range offsets, which return zeros: an
    // exhausted K group keeps iterating with the others on zero operands, and the compiler can count the loads in flight (with a path that issues
    // none it must assume the fewest: its s_waitcnt in front of the older register set then also drained the set issued a moment ago).
    auto issue = [&](float4 (*R)[2]) {
        const bool live = nextq < q1;
        if (live) {
            advance(nextq - pos);
            if (tap != tap_set) { set_tap(tap); tap_set = tap; }
            nextq += KG;
        }
        gload(R, tap, cc, live ? 0u : kOOB);
    };

    // ---- LDS addressing (swizzle: 16-byte half h of row r lives at half h ^ bit3(r))
    // half-step stages: 16-byte half h of row r lives at half h ^ bit3(r).  Full-step stages: 16-byte unit u (of 4) of row r lives at unit u ^ bits 2..3
    // of r - 16 consecutive rows then read (and write) 16 different bank groups
    const int frag_row = lane & 31;
    auto wswz = [&](int hf) -> int { return FS ? (((((hf << 1) | (c4 >> 1)) ^ ((r0 >> 2) & 3)) << 4) + ((c4 & 1) << 3)) : ((((c4 >> 1) ^ ((r0 >> 3) & 1)) << 4) + ((c4 & 1) << 3)); };
    auto rswz = [&](int hf) -> int { return FS ? ((((hf << 1) | (lane >> 5)) ^ ((frag_row >> 2) & 3)) << 4) : (((lane >> 5) ^ ((frag_row >> 3) & 1)) << 4); };
    float res[4] = {0.f, 0.f, 0.f, 0.f};
    auto cstep = [&](char* nb, float4 (*R)[2], int hf, int c) {        // plane c % NPL of staged value c / NPL
        const int v = c / NPL, pl = c % NPL;
        if (PREB && v >= A_IT) {                // filter rows arrive split: first terms in .x .y, second terms in .z .w
            if (pl == 0 && b_rows) {
                const float4 x = R[v][hf];
                char* q = nb + (NPL * BM + r0 + RP * (v - A_IT)) * ROWB + wswz(hf);
                *reinterpret_cast<uint2*>(q) = make_uint2(__float_as_uint(x.x), __float_as_uint(x.y));
                if (NPL > 1) *reinterpret_cast<uint2*>(q + BN * ROWB) = make_uint2(__float_as_uint(x.z), __float_as_uint(x.w));       // f16x1 uses the first terms only
            }
            return;
        }
        if (pl == 0) {
            const float4 x = R[v][hf]; res[0] = x.x; res[1] = x.y; res[2] = x.z; res[3] = x.w;
            if (F16) {
                const int sh = v < A_IT ? sh_a : sh_b;
#pragma unroll
                for (int e = 0; e < 4; ++e) res[e] = __builtin_ldexpf(res[e], sh);
            }
        }
        const pl4 t = PT::cvt(res);
        if (v < A_IT) {
            *reinterpret_cast<pl4*>(nb + (pl * BM + r0 + RP * v) * ROWB + wswz(hf)) = t;
        } else if (b_rows) {
            *reinterpret_cast<pl4*>(nb + (NPL * BM + pl * BN + r0 + RP * (v - A_IT)) * ROWB + wswz(hf)) = t;
        }
        if (pl + 1 < NPL) PT::residual(res, t);
    };
    constexpr int NMFMA = MR * NR * (NPL * (NPL + 1) / 2), CSTEPS = NV * NPL;
    constexpr int MPS = NMFMA / CSTEPS > 0 ? NMFMA / CSTEPS : 1;
    auto pipe = [&](const char* cur, char* nxt, float4 (*R)[2], int hf) {
        // every fragment read first (the compiler cannot prove the two stages disjoint: a read placed after a write would wait)
        // in the order the MFMAs below need them (LDS reads return in order, so the first MFMA waits for two reads, not for all of them): the last
        // plane of the filter fragments, the first plane of the pixel fragments, then the rest
        pl8 fa[MR][NPL], fb[NR][NPL];
        const int r_swz = rswz(0);
        auto rd_a = [&](int i, int pl) { fa[i][pl] = *reinterpret_cast<const pl8*>(cur + (pl * BM + (wm * MR + i) * 32 + frag_row) * ROWB + r_swz); };
        auto rd_b = [&](int j, int pl) { fb[j][pl] = *reinterpret_cast<const pl8*>(cur + (NPL * BM + pl * BN + (wn * NR + j) * 32 + frag_row) * ROWB + r_swz); };
#pragma unroll
        for (int j = 0; j < NR; ++j) rd_b(j, NPL - 1);
#pragma unroll
        for (int i = 0; i < MR; ++i) rd_a(i, 0);
#pragma unroll
        for (int pl = 1; pl < NPL; ++pl)
#pragma unroll
            for (int i = 0; i < MR; ++i) rd_a(i, pl);
#pragma unroll
        for (int pl = NPL - 2; pl >= 0; --pl)
#pragma unroll
            for (int j = 0; j < NR; ++j) rd_b(j, pl);
        __builtin_amdgcn_sched_barrier(0);
        int m = 0;
        // small terms first; consecutive MFMAs go to different accumulator tiles; a convert step after every MPS-th MFMA
#pragma unroll
        for (int sum = NPL - 1; sum >= 0; --sum)
#pragma unroll
            for (int pa = 0; pa <= sum; ++pa)
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        acc[i][j] = PT::mfma(fa[i][pa], fb[j][sum - pa], acc[i][j]);
                        if (m % MPS == 0 && m / MPS < CSTEPS) cstep(nxt, R, hf, m / MPS);
                        __builtin_amdgcn_sched_barrier(0);
                        ++m;
                    }
#pragma unroll
        for (int c = (NMFMA + MPS - 1) / MPS; c < CSTEPS; ++c) cstep(nxt, R, hf, c);
    };

    // full-step stages: the MFMAs of BOTH halves of the step in `cur`, the split of both halves of register set R (the next step) into `nxt` between them.
    // Small tiles read the fragments of both halves up front; tiles with four 32x32 pieces per wave read the second half's behind the first half's MFMAs.
    auto pipe_full = [&](const char* cur, char* nxt, float4 (*R)[2]) {
        constexpr bool BOTH = MR * NR <= 2;
        pl8 fa[BOTH ? 2 : 1][MR][NPL], fb[BOTH ? 2 : 1][NR][NPL];
        auto rd = [&](int h) {
            const int sw = rswz(h), s_ = BOTH ? h : 0;
#pragma unroll
            for (int j = 0; j < NR; ++j) fb[s_][j][NPL - 1] = *reinterpret_cast<const pl8*>(cur + (NPL * BM + (NPL - 1) * BN + (wn * NR + j) * 32 + frag_row) * ROWB + sw);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int i = 0; i < MR; ++i) fa[s_][i][pl] = *reinterpret_cast<const pl8*>(cur + (pl * BM + (wm * MR + i) * 32 + frag_row) * ROWB + sw);
#pragma unroll
            for (int pl = NPL - 2; pl >= 0; --pl)
#pragma unroll
                for (int j = 0; j < NR; ++j) fb[s_][j][pl] = *reinterpret_cast<const pl8*>(cur + (NPL * BM + pl * BN + (wn * NR + j) * 32 + frag_row) * ROWB + sw);
        };
        constexpr int NM2 = 2 * NMFMA, CS2 = 2 * CSTEPS, MPS2 = NM2 / CS2 > 0 ? NM2 / CS2 : 1;
        rd(0);
        if (BOTH) rd(1);
        __builtin_amdgcn_sched_barrier(0);
        int m = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!BOTH && h == 1) { rd(1); __builtin_amdgcn_sched_barrier(0); }
            const int s_ = BOTH ? h : 0;
#pragma unroll
            for (int sum = NPL - 1; sum >= 0; --sum)
#pragma unroll
                for (int pa = 0; pa <= sum; ++pa)
#pragma unroll
                    for (int i = 0; i < MR; ++i)
#pragma unroll
                        for (int j = 0; j < NR; ++j) {
                            acc[i][j] = PT::mfma(fa[s_][i][pa], fb[s_][j][sum - pa], acc[i][j]);
                            if (m % MPS2 == 0 && m / MPS2 < CS2) cstep(nxt, R, (m / MPS2) / CSTEPS, (m / MPS2) % CSTEPS);
                            __builtin_amdgcn_sched_barrier(0);
                            ++m;
                        }
        }
#pragma unroll
        for (int c = (NM2 + MPS2 - 1) / MPS2; c < CS2; ++c) cstep(nxt, R, c / CSTEPS, c % CSTEPS);
    };

    const int nloc = (q1 - q0 + KG - 1) / KG;   // pipeline iterations: the same for every group (barriers are block-wide)
    if (q0 < q1) {
        issue(R0);
        issue(R1);
    }
    if (F16) {
        __builtin_amdgcn_sched_barrier(0);
        sh_a = amax_shift_of(am_a); sh_b = amax_shift_of(am_b);
    }
    if constexpr (FS) {
        // S0 / S1 hold whole steps.  Phase q: the set whose step sits in `cur` is free - it is requested again (step q + 2) first; the other set
        // (step q + 1, requested one phase ago) is split into the other stage between the MFMAs of step q; one barrier.
        if (q0 < q1) {
#pragma unroll
            for (int c = 0; c < 2 * CSTEPS; ++c) cstep(S0, R0, c / CSTEPS, c % CSTEPS);
            __syncthreads();
        }
        for (int q = 0; q < nloc; q += 2) {
            issue(R0);                      // local step q + 2
            pipe_full(S0, S1, R1);          // MFMAs of step q, split of step q + 1 (past the end: zeros)
            __syncthreads();
            if (q + 1 < nloc) {
                issue(R1);                  // local step q + 3
                pipe_full(S1, S0, R0);
                __syncthreads();
            }
        }
    } else {
    if (q0 < q1) {
#pragma unroll
        for (int c = 0; c < CSTEPS; ++c) cstep(S0, R0, 0, c);
        __syncthreads();
    }
    for (int q = 0; q < nloc; q += 2) {
        pipe(S0, S1, R0, 1);            // MFMAs of local chunk q / half 0, convert chunk q / half 1
        issue(R0);                      // local chunk q+2
        __syncthreads();
        pipe(S1, S0, R1, 0);            // MFMAs of chunk q / half 1, convert chunk q+1 / half 0 (past the end: stale for KG = 1 and unused, zeros for KG > 1)
        __syncthreads();
        if (q + 1 < nloc) {
            pipe(S0, S1, R1, 1);
            issue(R1);                  // local chunk q+3
            __syncthreads();
            pipe(S1, S0, R0, 0);
            __syncthreads();
        }
    }
    }
    constexpr int EPG = 16 / KG;                // accumulator registers per 32x32 tile that one K group stores in the epilogue
    if constexpr (KG > 1) {
        // ---- sum the KG accumulator sets through LDS (the stages are free now) in the fixed order g = 0 .. KG-1.  EVERY group forms the sums, so
        //      all of them hold the finished tile and share the epilogue: group g stores (and, dgrad, takes the BatchNorm sums of) the registers
        //      g*EPG .. g*EPG+EPG-1 of every 32x32 tile = a quarter / half of its rows - 16 waves instead of 4 behind the main loop.
        float4* red = reinterpret_cast<float4*>(smem);
        constexpr int NQ = MR * NR * 4;         // float4 per thread
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j)
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4)
                    red[(grp * NQ + (i * NR + j) * 4 + e4) * 256 + tid] =
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
                        const float4 v = red[(g * NQ + (i * NR + j) * 4 + e4) * 256 + tid];
                        if (g == 0) { acc[i][j][4 * e4] = v.x; acc[i][j][4 * e4 + 1] = v.y; acc[i][j][4 * e4 + 2] = v.z; acc[i][j][4 * e4 + 3] = v.w; }
                        else { acc[i][j][4 * e4] += v.x; acc[i][j][4 * e4 + 1] += v.y; acc[i][j][4 * e4 + 2] += v.z; acc[i][j][4 * e4 + 3] += v.w; }
                    }
    }

    if (F16) {          // undo the two operand scales (exact: a power of two)
        const int sh = -(sh_a + sh_b);
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int jj = 0; jj < NR; ++jj)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][jj][e] = __builtin_ldexpf(acc[i][jj][e], sh);
    }
    // this group's share of the tile (static register indices per group: grp is wave-uniform); one group: the accumulators themselves
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
    const int rowg = 8 * ((grp * EPG) >> 2);    // first row of the group's registers inside a 32-row tile: register e <-> row (e&3) + 8*(e>>2) + 4*(lane>>5)
    const bool coop = COOP && a.splits > 1 && a.tickets != nullptr;     // the reduction over blockIdx.z happens here; from `single` on the block behaves like an unsplit launch
    if constexpr (COOP) {
        if (coop) {
            constexpr int NSH = KG > 1 ? MR * NR * EPG : MR * NR * 16, NV4 = NSH / 4, NTH = NT * KG;      // floats / float4 per thread in its share; threads per block
            static_assert(NSH % 4 == 0, "shares are whole float4");
            const long long tslab = (long long)NV4 * NTH * 4;                                            // floats per partial tile
            const long long ntile = (long long)a.mtiles * a.ntiles;
            const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.coop_slab + (long long)tile * tslab), 0, 0x7ffffff0, 0x00020000);
            const unsigned zstride = (unsigned)(ntile * tslab * 4);                                       // bytes between the partials of z and z + 1 (host: splits * zstride < 2^31)
            const unsigned toff = (unsigned)threadIdx.x * 16u;
            constexpr int kSC1 = 16;                                                                      // cache-policy bit 4 of the buffer builtins = sc1 (write-through / L1 bypass)
#pragma unroll
            for (int v = 0; v < NV4; ++v) {
                u32x4 q;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int f = 4 * v + e;
                    q[e] = __float_as_uint(share(f / (NSH / (MR * NR)) / NR, f / (NSH / (MR * NR)) % NR, f % (NSH / (MR * NR))));
                }
                __builtin_amdgcn_raw_buffer_store_b128(q, sr, (int)(toff + (unsigned)v * (unsigned)(NTH * 16) + (unsigned)z * zstride), 0, kSC1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every storing wave drains its stores ...
            __shared__ int coop_last;
            __syncthreads();                                         // ... before ONE lane signals for all of them
            if (threadIdx.x == 0) {
                unsigned* tk = a.tickets + coop_ticket_word(tile);
                const unsigned old = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = old == (unsigned)(a.splits - 1);
                if (last) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the next launch that uses this record finds zeros
                coop_last = last;
            }
            __syncthreads();
            if (!coop_last) return;
            // the partials of the other blocks, ZB at a time in flight (every load unconditional: an absent or own partial reads through an out-of-range offset)
            float tot[NSH];
#pragma unroll
            for (int f = 0; f < NSH; ++f) tot[f] = 0.f;
            constexpr int ZB = NV4 > 8 ? 1 : 4;          // partials in flight: 4 x 8 or 1 x 16 float4 per thread (one K group: acc + running sum + one partial = 192 registers)
            for (int z0 = 0; z0 < a.splits; z0 += ZB) {
                u32x4 in[ZB][NV4];
#pragma unroll
                for (int d = 0; d < ZB; ++d) {
                    const int zz = z0 + d;
                    const unsigned zoff = (zz < a.splits && zz != z) ? (unsigned)zz * zstride : kOOB;
#pragma unroll
                    for (int v = 0; v < NV4; ++v)
                        in[d][v] = __builtin_amdgcn_raw_buffer_load_b128(sr, (int)__builtin_elementwise_add_sat(zoff, toff + (unsigned)v * (unsigned)(NTH * 16)), 0, kSC1);
                }
#pragma unroll
                for (int d = 0; d < ZB; ++d) {
                    const int zz = z0 + d;
                    if (zz < a.splits) {
#pragma unroll
                        for (int f = 0; f < NSH; ++f) {
                            const float own = share(f / (NSH / (MR * NR)) / NR, f / (NSH / (MR * NR)) % NR, f % (NSH / (MR * NR)));
                            const float v = zz == z ? own : __uint_as_float(in[d][f / 4][f % 4]);
                            tot[f] = zz == 0 ? v : tot[f] + v;
                        }
                    }
                }
            }
#pragma unroll
            for (int f = 0; f < NSH; ++f) {
                const int i = f / (NSH / (MR * NR)) / NR, j = f / (NSH / (MR * NR)) % NR, e = f % (NSH / (MR * NR));
                if constexpr (KG > 1) mine[i][j][e] = tot[f]; else acc[i][j][e] = tot[f];
            }
            if constexpr (KG > 1) {
                // the forward BatchNorm partials below are taken by group 0 from whole 32x32 tiles: the groups' finished shares meet in LDS
                if (!DGRAD && a.stats != nullptr) {
                    float* xch = reinterpret_cast<float*>(smem);
                    __syncthreads();
#pragma unroll
                    for (int i = 0; i < MR; ++i)
#pragma unroll
                        for (int j = 0; j < NR; ++j)
#pragma unroll
                            for (int k = 0; k < EPG; ++k) xch[((i * NR + j) * 16 + grp * EPG + k) * 256 + tid] = mine[i][j][k];
                    __syncthreads();
                    if (grp == 0) {
#pragma unroll
                        for (int i = 0; i < MR; ++i)
#pragma unroll
                            for (int j = 0; j < NR; ++j)
#pragma unroll
                                for (int e = 0; e < 16; ++e) acc[i][j][e] = xch[((i * NR + j) * 16 + e) * 256 + tid];
                    }
                }
            }
        }
    }
    const bool single = coop || a.splits == 1;  // this block writes finished values to y (else: a slab of partials for splitk_reduce_kernel)
    // ---- epilogue: D[row][col], col = lane&31 (out channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5); bounds by the descriptor
    float* yout = a.y + (!single ? (long long)z * a.slab : 0ll);
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
            // pixel of row m of this 32-row block: m itself, or (parity-ordered dgrad) 32 consecutive pixels of one class: stride par apart
            const int pix0 = (DGRAD && par) ? dgrad_pix(a, min(mb32, a.M - 1)) : mb32, pst = (DGRAD && par) ? par : 1;
            if (a.accumulate && single) {        // y += result: all old values of the share are fetched before the first store
                float old[EPG];
#pragma unroll
                for (int e = 0; e < EPG; ++e) {
                    const int m = mb + (e & 3) + 8 * (e >> 2);
                    const unsigned off = (kok && m < a.M) ? ((unsigned)(pix0 + (m - mb32) * pst) * (unsigned)a.ldy + (unsigned)k) * 4u : kOOB;
                    old[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yr, (int)off, 0, 0));     // out-of-bounds offsets read 0
                }
#pragma unroll
                for (int e = 0; e < EPG; ++e) share_add(i, j, e, old[e]);
            }
#pragma unroll
            for (int e = 0; e < EPG; ++e) {
                const int m = mb + (e & 3) + 8 * (e >> 2);
                const unsigned off = (kok && m < a.M) ? ((unsigned)(pix0 + (m - mb32) * pst) * (unsigned)a.ldy + (unsigned)k) * 4u : kOOB;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(share(i, j, e) + bv), yr, (int)off, 0, 0);
            }
        }
    }
    bool bstats_done = false;
    if constexpr (DGRAD && KG == 1) {
        if (a.bstats != nullptr && single && a.bn_fast) {
            // Round 5: the same sums with 16-byte accesses.  The accumulator layout (lane = column, register = row) forces one 4-byte load per element of x
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
    if (DGRAD && a.bstats != nullptr && single && !bstats_done) {
        // BatchNorm-backward partials of the gradient tile just written (values as stored, no bias in a dgrad): sum(g), sum(g * xhat) per channel over
        // the wave's 32*MR rows, g = the gradient behind the ReLU mask; layout [2][mtiles * WGM][K] (dsrl_bn_bwd_from_stats).  K groups: every
        // group sums its share of the rows, the shares meet in LDS (behind the reduction area) and group 0 adds them in the order g = 0 .. KG-1.
        const int nparts = a.mtiles * WGM, part = tile_m * WGM + wm;
        float* ex = reinterpret_cast<float*>(smem) + (size_t)KG * MR * NR * 4 * 256 * 4;
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
                const int pix0 = par ? dgrad_pix(a, min(mb32, a.M - 1)) : mb32, pst = par ? par : 1;
#pragma unroll
                for (int e = 0; e < EPG; ++e) {          // all loads of the share first
                    const int m = mb32 + rq + (e & 3) + 8 * (e >> 2);
                    const bool ok = kok && m < a.M;
                    const long long px = pix0 + (m - mb32) * pst;
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
                ex[(((grp * 4 + wave) * NR + j) * 2 + 0) * 32 + lane] = sg;
                ex[(((grp * 4 + wave) * NR + j) * 2 + 1) * 32 + lane] = sgx;
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
                        sg += ex[(((g * 4 + wave) * NR + j) * 2 + 0) * 32 + lane];
                        sgx += ex[(((g * 4 + wave) * NR + j) * 2 + 1) * 32 + lane];
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
    // ---- BatchNorm partials of the tile this block just wrote: for every output channel (n, mean, M2) over the wave's 32*MR rows,
    //      two passes over the registers (exact centred second moment), the two lanes that share a channel combined with one shuffle.
    //      Layout [3][mtiles * WGM][K] = what bn_partial4_kernel writes, consumed by dsrl_bn_train_fwd_from_stats.
    if (a.stats != nullptr && single) {
        // Round 5: ONE partial per block tile.  The WGM wave rows of a tile used to leave one partial each (128 row blocks for an M = 4096 tensor on 64x64
        // tiles), and every block of the BatchNorm kernel merges all of them for its 32 channels before it streams a row: 2.9 us of a 6.9 us launch at
        // 128 partials, 1.1-1.4 us at 64 (tools/bn_prologue_probe.py).  The wave rows meet in LDS and wave row 0 merges them in the order wm = 1 .. WGM-1
        // (Chan's update: exact counts, the same formulas the BatchNorm kernels use).  Layout [3][mtiles][K].
        const int nparts = a.mtiles, part = tile_m;
        float* mg = reinterpret_cast<float*>(smem) + (KG > 1 ? (size_t)KG * MR * NR * 4 * 256 * 4 : 0);  // [WGM][WGN * NR][3][32], behind the K-group reduction area (the dgrad sums' exchange area)
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
                        if (m0 + (wm * MR + i) * 32 + rq + (e & 3) + 8 * (e >> 2) < a.M) { n += 1.f; sum += acc[i][j][e] + bv; }
                n += __shfl_xor(n, 32); sum += __shfl_xor(sum, 32);
                const float mean = n > 0.f ? sum / n : 0.f;
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (m0 + (wm * MR + i) * 32 + rq + (e & 3) + 8 * (e >> 2) < a.M) { const float d = acc[i][j][e] + bv - mean; q += d * d; }
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

}  // namespace dsrl
