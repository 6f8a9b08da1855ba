// ConvTranspose2d k2 s2 backward with LDS-DMA staging (round 5): the logits tail of the step (models/DSRL.py:53-69 of the reference builds the layer,
// loss.backward() of train_or_resume.py:435-444 runs this pass).
//
// convt2x2_bwd_mfma_kernel (spatial.hip) loads a 128-pixel segment into 52 registers per lane, re-lays it out with ds_write_b32 behind two
// barriers and needs all 256 VGPRs, so a CU holds two blocks whose staging phases leave the matrix pipe half idle (169 us for 478 MB on the step's
// last layer).  Here the segment goes global -> LDS with buffer_load_dwordx4 ... lds in the layout it has in memory:
//     stage = [dy row 2h, pixels 2*w0 .. 2*w0+255][dy row 2h+1, the same pixels][x row h, pixels w0 .. w0+127]      (19 + 19 + 10 KB for 19 channels)
// With an odd channel count the MFMA operand reads of that layout hit distinct banks: the gradient of input pixel p, column k = tap*CO + co of the
// [px x 4 CO] matrix sits at word p * 2 CO + k (k < 2 CO, output row 2h) or one row segment further (k >= 2 CO), and 2 CO * p mod 64 is a bijection
// of 32 pixels onto the even banks (the half-wave with k + 1 reads the odd ones).  No registers, no ds_write, ONE barrier per segment, a ring of three
// stages.
// A block = 8 waves: waves 0-3 form dx [32 px x CI] = G . W^T of their 32 pixels (38 x v_mfma_f32_32x32x2_f32, filter in registers) and store it;
// waves 4-7 issue the DMA pieces (they have no other vector memory traffic, so their s_waitcnt vmcnt sees exactly the pieces) and accumulate
// dw [(CI + 1) x 4 CO] = X^T . G over every segment of the block (row CI multiplies ones: db per tap).  The arithmetic, the accumulation order within
// a segment and the per-block partial layout are those of convt2x2_bwd_mfma_kernel; convt2x2_dw_finalize_kernel merges the partials.
//
// CE = true (dsrl_convt2x2_bwd_ce): the layer's output IS the logits of nn.CrossEntropyLoss and the incoming gradient is never materialised.  The
// DMA brings the LOGITS rows and the target bytes of the segment, and d(loss)/d(logits) is formed in place in LDS - max, exp, sum, scale in the order
// and with the roundings of ce_fused_kernel (losses.hip), plus the stride-s feature transformer's rank-one contribution g * w_c on the sampled pixels
// (pointwise_bwd_kernel's `dx += g * w`) - one segment ahead of the MFMAs that consume it.  TW = true (default): by four more waves (8-11, one per SIMD,
// two output pixels per lane) while waves 0-7 multiply the previous segment; TW = false (DSRL_CONVT_CE_WAVES=8): inside the eight MFMA waves, one
// pixel per lane, the dx waves before their MFMAs and the dw waves after theirs.  The 319 MB gradient write of the loss pass and its 319 MB read
// here disappear; results are bit-identical to ce_fused + pointwise_bwd + this kernel with CE = false.
// Measured (profiles/round5_convt_tail.txt): 135 us without CE, 176-183 us with (TW), 188-195 (TW = false).  The softmax does not hide behind the
// MFMAs: SQ counters read 43 % of the SIMD cycles in MFMAs + 28 % in the other vector instructions with no co-execution counted, i.e. the ~290
// vector instructions per pixel add to the matrix work whichever wave issues them.
#include "common.h"
#include "lds_dma.h"
#include <algorithm>

namespace dsrl {

using f32x16_t = __attribute__((ext_vector_type(16))) float;

struct ConvtCeArgs {
    const unsigned char* target;    // (N, 2H, 2W) labels
    const float* count;             // number of pixels that are not ignore_index (ce_finalize_kernel's out[1])
    const float* ft_g;              // (N, Hf, Wf) incoming gradient of the stride-s 1x1 conv on the logits, or null
    const float* ft_w;              // its CO weights
    int ignore_index, ft_s, ft_shift, Hf, Wf;     // ft_shift = log2(ft_s) when ft_s is a power of two, else -1
};

template <int CI, int CO, bool CE, bool TW = false>
__global__ __launch_bounds__(TW ? 768 : 512, 1) void convt2x2_bwd_dma_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dy,
                                                                 float* __restrict__ dx, float* __restrict__ part, int N, int H, int W,
                                                                 int nseg_per_row, int nseg, ConvtCeArgs ce) {
    constexpr int TP = 128, COLS = 4 * CO, NK = COLS / 2, NJ = (COLS + 31) / 32, NOUT = CI * COLS;
    constexpr int XB = TP * CI * 4, XP = (XB + 1023) / 1024;               // bytes / 1 KB pieces of an x segment
    constexpr int GB = 2 * TP * CO * 4, GP = GB / 1024;                    // of one dy row segment
    constexpr int PIECES = XP + 2 * GP + (CE ? 2 : 0), STAGE = PIECES * 1024, S = 3;
    constexpr int MAXPW = (PIECES + 3) / 4;                                // pieces per DMA wave (wave q takes pieces q, q + 4, ...)
    constexpr int GW = GB / 4;                                             // words between the two gradient rows of a stage
    constexpr int TOFF = (XP + 2 * GP) * 1024;                             // target bytes of the two output rows: 1 KB slots, 2 TP bytes used
    static_assert(CI < 32 && CO % 2 == 1 && NJ <= 3 && GB % 1024 == 0 && MAXPW <= 30 && 2 * TP <= 1024, "tile does not fit this kernel");
    static_assert(S * STAGE <= 160 * 1024, "three stages in LDS");
    static_assert((3 * NJ * 16 * 64 + (CI + 1) * COLS) * 4 <= S * STAGE, "the final merge of the dw tiles reuses the ring");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned ring_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_dw = wave >= 4 && wave < 8, is_tw = wave >= 8;           // waves 0-3: dx, 4-7: DMA + dw, 8-11 (CE): logits -> gradient
    const int wq = wave & 3;                                               // quarter of the segment this wave works on / DMA lane of the dw waves
    const int l31 = lane & 31, lh = lane >> 5;
    const int px0 = 32 * wq;
    const int grid = (int)gridDim.x;

    // Three descriptors for the whole tensors, made once; a piece's position travels in the scalar offset (segment + piece: a few scalar adds per
    // segment instead of three 64-bit descriptors, which cost 40 spilled SGPRs and a scalar preamble in every wave of the CE build).  The scalar offset
    // is not bounds-checked, so only pieces that lie inside their tensor use it; the x piece that crosses the end of its segment and the 1 KB label
    // pieces (2 TP bytes of a row are used) carry their position in the checked vector offset: they fetch neighbouring bytes of the tensor into LDS
    // words nobody reads, and zeros past the end of the tensor.  The host checks that every tensor is below 4 GB.
    const unsigned l16 = (unsigned)lane * 16u;
    const dma_u32x4 rs_dy = make_rsrc(dy, (unsigned)((long long)N * 4 * H * W * CO * 4));
    const dma_u32x4 rs_x = make_rsrc(x, (unsigned)((long long)N * H * W * CI * 4));
    const dma_u32x4 rs_t = make_rsrc(CE ? (const void*)ce.target : (const void*)dy, (unsigned)((long long)N * 4 * H * W));
    auto issue = [&](int seg, int slot) {                                  // dw waves only: this wave's pieces of the segment
        const int row = seg / nseg_per_row;                                // n*H + h
        const int w0 = (seg - row * nseg_per_row) * TP;
        const int n = row / H, h = row - n * H;
        const unsigned opx = (unsigned)((n * 2 * H + 2 * h) * (2 * W) + 2 * w0);         // first output pixel of the segment (row 2h)
        const unsigned o0 = opx * (unsigned)(CO * 4), o1 = o0 + (unsigned)(2 * W * CO * 4);  // byte offsets of the two gradient rows
        const unsigned ox = (unsigned)(row * W + w0) * (unsigned)(CI * 4);
        const unsigned base = ring_lds + (unsigned)(slot * STAGE);
#pragma unroll
        for (int i = 0; i < MAXPW; ++i) {
            const int q = wq + 4 * i;                                       // wave-uniform
            const unsigned lds = base + (unsigned)q * 1024u;
            if (q < GP) lds_dma16(rs_dy, l16, o0 + (unsigned)(q * 1024), lds);
            else if (q < 2 * GP) lds_dma16(rs_dy, l16, o1 + (unsigned)((q - GP) * 1024), lds);
            else if (q < 2 * GP + XP - 1 || XB % 1024 == 0) lds_dma16(rs_x, l16, ox + (unsigned)((q - 2 * GP) * 1024), lds);
            else if (q < 2 * GP + XP) lds_dma16(rs_x, l16 + ox + (unsigned)((XP - 1) * 1024), 0u, lds);           // may cross the end of x: checked offset
            else if (CE && q < PIECES) lds_dma16(rs_t, l16 + opx + (unsigned)((q - (2 * GP + XP)) * 2 * W), 0u, lds);   // 1 KB from a 2 TP byte row: likewise
        }
    };

    // ---- CE: logits -> d(loss)/d(logits) of this wave's 64 output pixels of a landed stage, in place.  ce_fused_kernel's arithmetic, per pixel:
    //      m = max, e_c = exp_nonpos(v_c - m), s = sum e_c (c ascending), dl_c = e_c * (scale / s) - (c == target ? scale : 0); ignored pixel: zeros
    float scale = 0.f;
    __shared__ float ftw_s[32];                                            // the feature transformer's weights (read as a broadcast by the few lanes on its grid)
    if (CE) {
        scale = 1.f / ce.count[0];
        if (tid < 32) ftw_s[tid] = (ce.ft_g && tid < CO) ? ce.ft_w[tid] : 0.f;          // visible after the first barrier below
    }
    auto transform = [&](int seg, int slot) {                              // TW: wave wq of the tw waves takes 128 output pixels, two per lane; else every wave 64
        const int r = TW ? wq >> 1 : (is_dw ? 1 : 0);                       // output row 2h + r of the segment
        // the transformer's gradient g of this lane's pixels, fetched FIRST: the softmax arithmetic covers its latency (fetched where it is used it
        // stalled the wave for a memory round trip in every fourth row: +19 us on the step's last layer).  Odd output rows are never on the grid.
        float gft[TW ? 2 : 1];
        bool gon[TW ? 2 : 1];
#pragma unroll
        for (int u = 0; u < (TW ? 2 : 1); ++u) { gft[u] = 0.f; gon[u] = false; }
        bool any_grid = false;
        if (ce.ft_g && r == 0) {
            const int row = seg / nseg_per_row;
            const int w0 = (seg - row * nseg_per_row) * TP;
            const int n = row / H, oh = 2 * (row - n * H);
            const bool row_on = ce.ft_shift >= 0 ? (oh & (ce.ft_s - 1)) == 0 : oh % ce.ft_s == 0;      // wave-uniform
            if (row_on) {
                any_grid = true;
                const int fh = ce.ft_shift >= 0 ? oh >> ce.ft_shift : oh / ce.ft_s;
#pragma unroll
                for (int u = 0; u < (TW ? 2 : 1); ++u) {
                    const int ow = 2 * w0 + (TW ? (wq & 1) * 128 + u * 64 + lane : wq * 64 + lane);
                    const bool on = ce.ft_shift >= 0 ? (ow & (ce.ft_s - 1)) == 0 : ow % ce.ft_s == 0;
                    const int fw = ce.ft_shift >= 0 ? ow >> ce.ft_shift : ow / ce.ft_s;
                    gon[u] = on;
                    if (on) gft[u] = ce.ft_g[((long long)n * ce.Hf + fh) * ce.Wf + fw];
                }
            }
        }
#pragma unroll 1
        for (int u = 0; u < (TW ? 2 : 1); ++u) {                            // (rolled: two pixels in flight would need 2 x 19 more registers)
            const float g_u = (TW && u) ? gft[TW ? 1 : 0] : gft[0];
            const bool on_u = (TW && u) ? gon[TW ? 1 : 0] : gon[0];
            const int j = TW ? (wq & 1) * 128 + u * 64 + lane : wq * 64 + lane;       // output pixel 2*w0 + j of that row
            float* v = reinterpret_cast<float*>(smem + slot * STAGE) + r * GW + j * CO;
            const int tg = reinterpret_cast<const unsigned char*>(smem + slot * STAGE + TOFF + r * 1024)[j];
            float e[CO];
#pragma unroll
            for (int c = 0; c < CO; ++c) e[c] = v[c];
            float m = e[0];
#pragma unroll
            for (int c = 1; c < CO; ++c) m = fmaxf(m, e[c]);
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < CO; ++c) { e[c] = exp_nonpos(e[c] - m); s += e[c]; }
            // e_c * inv - (c == tg ? scale : 0), zeros for an ignored pixel; then the feature transformer's term: ((e * inv) - scale) + g * w, the
            // roundings of ce_fused_kernel and pointwise_bwd_kernel.  Everything stays in registers between the 19 reads and the 19 writes: a
            // read-modify-write of the target word in LDS instead of the 19 selects measured +11 us
            const bool live = tg != ce.ignore_index;
            const float inv = scale / s;
#pragma unroll
            for (int c = 0; c < CO; ++c) e[c] = live ? e[c] * inv - (c == tg ? scale : 0.f) : 0.f;
            if (any_grid) {                                                 // wave-uniform: every eighth lane of every fourth row at stride 8
#pragma unroll
                for (int c = 0; c < CO; ++c) e[c] = on_u ? e[c] + g_u * ftw_s[c] : e[c];
            }
#pragma unroll
            for (int c = 0; c < CO; ++c) v[c] = e[c];
        }
    };

    // One set of persistent registers per wave: the dw waves' three accumulator tiles; in the dx waves the same registers hold the transposed filter,
    // the B operand of the dx GEMM (k = tap * CO + co, two per MFMA, n = ci) - declared once so that the two roles do not add up in the allocation.
    f32x16_t accw[NJ];
    static_assert(NK <= NJ * 16, "the filter fits the accumulator registers");
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) accw[j][e] = 0.f;
    if (wave < 4) {
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            const int k = 2 * kk + lh, tap = k / CO, co = k - tap * CO;
            accw[kk / 16][kk % 16] = l31 < CI ? w[(l31 * CO + co) * 4 + tap] : 0.f;
        }
    }
    // word of column col = 32 j + l31 of the gradient matrix inside a stage, relative to the pixel's first word
    int colw[NJ];
    bool colok[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int col = 32 * j + l31; colok[j] = col < COLS; colw[j] = !colok[j] ? 0 : col < 2 * CO ? col : GW + col - 2 * CO; }
    const int xl = l31 < CI ? l31 : 0;
    const float aconst = l31 == CI ? 1.f : 0.f;

    // Schedule: at the barrier of iteration t segment t is ready to multiply (CE: transformed during iteration t - 1), segment t + 1 has LANDED (it
    // was issued during iteration t - 1 and the DMA waves drained their counters before the barrier), and the slot of segment t - 1 is free for
    // segment t + 2, issued right after the barrier.
    int seg = (int)blockIdx.x, t = 0;
    if (is_dw && seg < nseg) issue(seg, 0);
    if (CE) {
        if (is_dw) s_waitcnt_vm<0>();
        block_barrier();
        if ((is_tw || !TW) && seg < nseg) transform(seg, 0);
    }
    if (is_dw && seg + grid < nseg) issue(seg + grid, 1);
    for (; seg < nseg; seg += grid, ++t) {
        const int slot = t % S;
        if (is_dw) s_waitcnt_vm<0>();
        block_barrier();
        if (is_dw && seg + 2 * grid < nseg) issue(seg + 2 * grid, (t + 2) % S);
        const float* G = reinterpret_cast<const float*>(smem + slot * STAGE);
        const float* X = G + 2 * GW;
        const bool next = seg + grid < nseg;
        if (is_tw) {
            if (CE && next) transform(seg + grid, (t + 1) % S);
        } else if (!is_dw) {
            if (CE && !TW && next) transform(seg + grid, (t + 1) % S);
            // ---- dx tile of the wave's 32 pixels: A = gradients (row = pixel, two k per MFMA), B = transposed filter
            const int row = seg / nseg_per_row;
            const int w0 = (seg - row * nseg_per_row) * TP;
            f32x16_t acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            const float* g = G + (px0 + l31) * (2 * CO) + lh;
            constexpr int BK = TW ? 10 : 19, NBK = (NK + BK - 1) / BK;     // operand reads in batches; batch b + 1 is read while batch b multiplies
            float ga[2][BK];
            auto fetch = [&](int b, float (&a)[BK]) {
#pragma unroll
                for (int u = 0; u < BK; ++u) { const int kk = b * BK + u; if (kk < NK) a[u] = g[(kk < CO ? 0 : GW - 2 * CO) + 2 * kk]; }
            };
            fetch(0, ga[0]);
#pragma unroll
            for (int b = 0; b < NBK; ++b) {
                if (b + 1 < NBK) fetch(b + 1, ga[(b + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise re-serialises read -> wait -> two MFMAs through one register pair)
#pragma unroll
                for (int u = 0; u < BK; ++u) {
                    const int kk = b * BK + u;
                    if (kk < NK) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[b & 1][u], accw[kk / 16][kk % 16], acc, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (l31 < CI) {
                float* o = dx + ((long long)row * W + w0) * CI + l31;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int px = px0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    o[(long long)px * CI] = acc[e];
                }
            }
        } else {
            // ---- dw (+ db through the ones row): A = inputs transposed (row = ci, k = pixel), B = gradients (k = pixel, column = (tap, co))
            // unconditional reads (lanes past CI / past the last column fetch a valid word and drop it): no branches between the reads, so they are
            // issued ahead of the MFMAs that use them
            constexpr int HB = TW ? 2 : 4, NB = 16 / HB;   // pixel pairs per batch of operand reads; batch b + 1 is read while batch b multiplies
            float av[2][HB], bv[2][HB][NJ];
            auto fetch = [&](int b, float (&a)[HB], float (&v)[HB][NJ]) {
#pragma unroll
                for (int u = 0; u < HB; ++u) {
                    const int px = px0 + 2 * (b * HB + u) + lh;
                    a[u] = X[px * CI + xl];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) v[u][j] = G[px * (2 * CO) + colw[j]];
                }
            };
            fetch(0, av[0], bv[0]);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (b + 1 < NB) fetch(b + 1, av[(b + 1) & 1], bv[(b + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < HB; ++u) {
                    const float a = l31 < CI ? av[b & 1][u] : aconst;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) accw[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, colok[j] ? bv[b & 1][u][j] : 0.f, accw[j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (CE && !TW && next) transform(seg + grid, (t + 1) % S);
        }
    }
    // ---- merge the four dw waves' tiles (fixed order) and leave the block's partials in the layout convt2x2_dw_finalize_kernel merges
    float* buf = reinterpret_cast<float*>(smem);
    float* xa = buf + 3 * NJ * 16 * 64;
    __syncthreads();
    if (is_dw && wq > 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) buf[(((wq - 1) * NJ + j) * 16 + e) * 64 + lane] = accw[j][e];
    }
    __syncthreads();
    if (is_dw && wq == 0) {
#pragma unroll 1
        for (int u = 0; u < 3; ++u)              // (one wave's tiles at a time: unrolled, the 144 reads in flight set the kernel's register count)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) accw[j][e] += buf[((u * NJ + j) * 16 + e) * 64 + lane];
#pragma unroll
        for (int j = 0; j < NJ; ++j)              // D[row = ci][col = (tap, co)] -> LDS as a dense [CI + 1][COLS] matrix
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ci = (e & 3) + 8 * (e >> 2) + 4 * lh, col = 32 * j + l31;
                if (ci <= CI && col < COLS) xa[ci * COLS + col] = accw[j][e];
            }
    }
    __syncthreads();
    float* po = part + (long long)blockIdx.x * (NOUT + CO);
    for (int o = tid; o < NOUT + CO; o += (int)blockDim.x) {
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

// host side: supported = 19 -> 19 channels, W a multiple of the 128-pixel segment, 16-byte aligned tensors
bool convt_bwd_dma_supported(const void* x, const void* dy, int W, int Cin, int Cout) {
    const char* v = getenv("DSRL_CONVT_DMA");
    if (v && atoi(v) == 0) return false;
    return Cin == 19 && Cout == 19 && W % 128 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0;
}
int convt_bwd_dma_blocks(long long nseg, int cap) { return (int)std::min<long long>(std::min(cap, kNumCU), nseg); }

template <bool CE, bool TW>
static int launch_dma(const float* x, const float* w, const float* dy, float* dx, float* part, int N, int H, int W, int nblocks, const ConvtCeArgs& ce, hipStream_t st) {
    constexpr int CI = 19, CO = 19;
    constexpr int kLds = 3 * ((128 * CI * 4 + 1023) / 1024 + 2 * (2 * 128 * CO * 4 / 1024) + (CE ? 2 : 0)) * 1024;
    static const hipError_t attr = hipFuncSetAttribute((const void*)convt2x2_bwd_dma_kernel<CI, CO, CE, TW>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (attr != hipSuccess) { set_error("convt2x2_bwd_dma_kernel: %d bytes of LDS refused (%s)", kLds, hipGetErrorString(attr)); return DSRL_E_LAUNCH; }
    const int nseg_per_row = W / 128;
    const long long nseg = (long long)N * H * nseg_per_row;
    if ((long long)N * 4 * H * W * CO * 4 >= (1ll << 32)) { set_error("convt2x2_bwd_dma_kernel: output gradient of 4 GB or more"); return DSRL_E_UNSUPPORTED; }
    if (nseg >= (1ll << 31) || nblocks < 1 || nblocks > nseg) { set_error("convt2x2_bwd_dma_kernel: %lld segments, %d blocks", nseg, nblocks); return DSRL_E_BADARG; }
    hipLaunchKernelGGL((convt2x2_bwd_dma_kernel<CI, CO, CE, TW>), dim3(nblocks), dim3(TW ? 768 : 512), kLds, st, x, w, dy, dx, part, N, H, W, nseg_per_row, (int)nseg, ce);
    return launch_status("convt2x2_bwd_dma_kernel");
}
int launch_convt_bwd_dma(const float* x, const float* w, const float* dy, float* dx, float* part, int N, int H, int W, int nblocks, hipStream_t st) {
    return launch_dma<false, false>(x, w, dy, dx, part, N, H, W, nblocks, ConvtCeArgs{}, st);
}
int launch_convt_bwd_dma_ce(const float* x, const float* w, const float* logits, float* dx, float* part, int N, int H, int W, int nblocks,
                            const unsigned char* target, int ignore_index, const float* count, const float* ft_g, const float* ft_w, int ft_s, hipStream_t st) {
    ConvtCeArgs ce{};
    ce.target = target; ce.count = count; ce.ft_g = ft_g; ce.ft_w = ft_w; ce.ignore_index = ignore_index; ce.ft_s = ft_s > 0 ? ft_s : 1;
    ce.ft_shift = -1;
    for (int b = 0; b < 31; ++b) if (ce.ft_s == (1 << b)) ce.ft_shift = b;
    ce.Hf = (2 * H - 1) / ce.ft_s + 1; ce.Wf = (2 * W - 1) / ce.ft_s + 1;
    const char* tw = getenv("DSRL_CONVT_CE_WAVES");
    if (tw && atoi(tw) == 8) return launch_dma<true, false>(x, w, logits, dx, part, N, H, W, nblocks, ce, st);      // the transform inside the MFMA waves
    return launch_dma<true, true>(x, w, logits, dx, part, N, H, W, nblocks, ce, st);
}

}  // namespace dsrl
