// ConvTranspose2d k2 s2 backward with LDS-DMA staging (round 5): the logits tail of the step (models/DSRL.py:53-69 of the reference builds the layer,
// loss.backward() of train_or_resume.py:435 runs this pass).
//
// convt2x2_bwd_mfma_kernel (spatial.hip) loads a 128-pixel segment into 52 registers per lane, re-lays it out with ds_write_b32 behind two
// barriers and needs all 256 VGPRs, so a CU holds two blocks whose staging phases leave the matrix pipe half idle (169 us for 478 MB on the step's
// last layer).  Here the segment goes global -> LDS with buffer_load_dwordx4 ... lds in the layout it has in memory:
//     stage = [dy row 2h, pixels 2*w0 .. 2*w0+255][dy row 2h+1, the same pixels][x row h, pixels w0 .. w0+127]      (19 + 19 + 10 KB for 19 channels)
// With an odd channel count the MFMA operand reads of that layout hit distinct banks: the gradient of input pixel p, column k = tap*CO + co of the
// [px x 4 CO] matrix sits at word p * 2 CO + k (k < 2 CO, output row 2h) or one row segment further (k >= 2 CO), and 2 CO * p mod 64 is a bijection
// of 32 pixels onto the even banks (the half-wave with k + 1 reads the odd ones).  No registers, no ds_write, ONE barrier per segment, a ring of three
// stages with two segments in flight per CU.
// A block = 8 waves: waves 0-3 form dx [32 px x CI] = G . W^T of their 32 pixels (38 x v_mfma_f32_32x32x2_f32, filter in registers) and store it;
// waves 4-7 issue the DMA pieces (12 each per segment: they have no other vector memory traffic, so their s_waitcnt vmcnt counts are exact) and
// accumulate dw [(CI + 1) x 4 CO] = X^T . G over every segment of the block (row CI multiplies ones: db per tap).  The arithmetic, the accumulation
// order within a segment and the per-block partial layout are those of convt2x2_bwd_mfma_kernel; convt2x2_dw_finalize_kernel merges the partials.
#include "common.h"
#include "lds_dma.h"
#include <algorithm>

namespace dsrl {

using f32x16_t = __attribute__((ext_vector_type(16))) float;

template <int CI, int CO, int ABL = 0>
__global__ __launch_bounds__(512, 1) void convt2x2_bwd_dma_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dy,
                                                                 float* __restrict__ dx, float* __restrict__ part, int N, int H, int W,
                                                                 int nseg_per_row, int nseg) {
    constexpr int TP = 128, COLS = 4 * CO, NK = COLS / 2, NJ = (COLS + 31) / 32, NOUT = CI * COLS;
    constexpr int XB = TP * CI * 4, XP = (XB + 1023) / 1024;               // bytes / 1 KB pieces of an x segment
    constexpr int GB = 2 * TP * CO * 4, GP = GB / 1024;                    // of one dy row segment
    constexpr int PIECES = XP + 2 * GP, PW = PIECES / 4, STAGE = PIECES * 1024, S = 3;
    constexpr int GW = GB / 4;                                             // words between the two gradient rows of a stage
    static_assert(CI < 32 && CO % 2 == 1 && NJ <= 3 && GB % 1024 == 0 && PIECES % 4 == 0 && PW <= 30, "tile does not fit this kernel");
    static_assert(3 * NJ * 16 * 64 * 4 <= S * STAGE && (CI + 1) * COLS * 4 <= STAGE, "the final merge of the dw tiles reuses the ring");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned ring_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_dw = wave >= 4;
    const int wq = wave & 3;                                               // quarter of the segment this wave works on / DMA lane of the dw waves
    const int l31 = lane & 31, lh = lane >> 5;
    const int px0 = 32 * wq;
    const int grid = (int)gridDim.x;

    auto issue = [&](int seg, int slot) {                                  // dw waves only: the PW pieces of this wave
        const int row = seg / nseg_per_row;                                // n*H + h
        const int w0 = (seg - row * nseg_per_row) * TP;
        const int n = row / H, h = row - n * H;
        const float* d = dy + (((long long)(n * 2 * H + 2 * h)) * (2 * W) + 2 * w0) * CO;
        const dma_u32x4 r0 = make_rsrc(d, GB), r1 = make_rsrc(d + 2ll * W * CO, GB);
        const dma_u32x4 xr = make_rsrc(x + ((long long)row * W + w0) * CI, XB);
        const unsigned base = ring_lds + (unsigned)(slot * STAGE);
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int q = wq + 4 * i;                                       // wave-uniform
            const unsigned lds = base + (unsigned)q * 1024u;
            if (q < GP) lds_dma16(r0, (unsigned)(q * 1024 + lane * 16), 0u, lds);
            else if (q < 2 * GP) lds_dma16(r1, (unsigned)((q - GP) * 1024 + lane * 16), 0u, lds);
            else lds_dma16(xr, (unsigned)((q - 2 * GP) * 1024 + lane * 16), 0u, lds);          // past XB: zeros through the bounds check
        }
    };

    // transposed filter as the B operand of the dx GEMM: k = tap * CO + co (two per MFMA), n = ci
    float wreg[NK];
    f32x16_t accw[NJ];
    if (!is_dw) {
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            const int k = 2 * kk + lh, tap = k / CO, co = k - tap * CO;
            wreg[kk] = l31 < CI ? w[(l31 * CO + co) * 4 + tap] : 0.f;
        }
    } else {
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) wreg[kk] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) accw[j][e] = 0.f;
    // word of column col = 32 j + l31 of the gradient matrix inside a stage, relative to the pixel's first word
    int colw[NJ];
    bool colok[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int col = 32 * j + l31; colok[j] = col < COLS; colw[j] = !colok[j] ? 0 : col < 2 * CO ? col : GW + col - 2 * CO; }
    const int xl = l31 < CI ? l31 : 0;
    const float aconst = l31 == CI ? 1.f : 0.f;

    int seg = (int)blockIdx.x, t = 0;
    if (is_dw) {
        if (seg < nseg) issue(seg, 0);
        if (ABL < 2 && seg + grid < nseg) issue(seg + grid, 1);
    }
    for (; seg < nseg; seg += grid, ++t) {
        const int slot = t % S;
        if (is_dw) {
            if (ABL < 2 && seg + grid < nseg) s_waitcnt_vm<PW>(); else s_waitcnt_vm<0>();     // my pieces of this segment have landed (the next segment's stay in flight)
        }
        block_barrier();                        // every wave's pieces have; and every wave is done reading the slot the next issue overwrites
        if (ABL < 2 && is_dw && seg + 2 * grid < nseg) issue(seg + 2 * grid, (t + 2) % S);
        const float* G = reinterpret_cast<const float*>(smem + slot * STAGE);
        const float* X = G + 2 * GW;
        if (ABL == 1) continue;
        if (ABL == 4 && is_dw) continue;
        if (ABL == 5 && !is_dw) continue;
        if (!is_dw) {
            // ---- dx tile of the wave's 32 pixels: A = gradients (row = pixel, two k per MFMA), B = transposed filter
            const int row = seg / nseg_per_row;
            const int w0 = (seg - row * nseg_per_row) * TP;
            f32x16_t acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            const float* g = G + (px0 + l31) * (2 * CO) + lh;
            float ga[NK];                        // every operand read is issued before the first MFMA waits for one
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) ga[kk] = g[(kk < CO ? 0 : GW - 2 * CO) + 2 * kk];
            __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise re-serialises read -> wait -> two MFMAs through one register pair)
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[kk], wreg[kk], acc, 0, 0, 0);
            if (ABL == 3 ? (l31 < CI && acc[0] == 123.456f) : l31 < CI) {
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
            constexpr int HB = 4, NB = 16 / HB;   // pixel pairs per batch of operand reads; batch b + 1 is read while batch b multiplies
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
        }
    }
    // ---- merge the four dw waves' tiles (fixed order) and leave the block's partials in the layout convt2x2_dw_finalize_kernel merges
    float* buf = reinterpret_cast<float*>(smem);
    float* xa = buf + 3 * NJ * 16 * 64;
    static_assert((3 * NJ * 16 * 64 + (CI + 1) * COLS) * 4 <= S * STAGE, "merge area");
    __syncthreads();
    if (is_dw && wq > 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) buf[(((wq - 1) * NJ + j) * 16 + e) * 64 + lane] = accw[j][e];
    }
    __syncthreads();
    if (is_dw && wq == 0) {
#pragma unroll
        for (int u = 0; u < 3; ++u)
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
    for (int o = tid; o < NOUT + CO; o += 512) {
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

// host side: nonzero = supported (19 -> 19 channels, W a multiple of the 128-pixel segment, 16-byte aligned tensors)
bool convt_bwd_dma_supported(const void* x, const void* dy, int W, int Cin, int Cout) {
    const char* v = getenv("DSRL_CONVT_DMA");
    if (v && atoi(v) == 0) return false;
    return Cin == 19 && Cout == 19 && W % 128 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0;
}
int convt_bwd_dma_blocks(long long nseg, int cap) { return (int)std::min<long long>(std::min(cap, kNumCU), nseg); }
int launch_convt_bwd_dma(const float* x, const float* w, const float* dy, float* dx, float* part, int N, int H, int W, int nblocks, hipStream_t st) {
    constexpr int CI = 19, CO = 19;
    constexpr int kLds = 3 * ((128 * CI * 4 + 1023) / 1024 + 2 * (2 * 128 * CO * 4 / 1024)) * 1024;
    static const hipError_t attr = hipFuncSetAttribute((const void*)convt2x2_bwd_dma_kernel<CI, CO>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (attr != hipSuccess) { set_error("convt2x2_bwd_dma_kernel: %d bytes of LDS refused (%s)", kLds, hipGetErrorString(attr)); return DSRL_E_LAUNCH; }
    const int nseg_per_row = W / 128;
    const long long nseg = (long long)N * H * nseg_per_row;
    if (nseg >= (1ll << 31) || nblocks < 1 || nblocks > nseg) { set_error("convt2x2_bwd_dma_kernel: %lld segments, %d blocks", nseg, nblocks); return DSRL_E_BADARG; }
#ifdef DSRL_CONVT_ABLATION          // timing builds: 1 = DMA + barriers only, 2 = compute on stale LDS (no loads), 3 = 2 without the dx stores, 4 / 5 = 2 with dx / dw only
    const char* ab = getenv("DSRL_CONVT_ABL");
#define ABL_CASE(n) if (ab && atoi(ab) == n) { hipFuncSetAttribute((const void*)convt2x2_bwd_dma_kernel<CI, CO, n>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds); \
        hipLaunchKernelGGL((convt2x2_bwd_dma_kernel<CI, CO, n>), dim3(nblocks), dim3(512), kLds, st, x, w, dy, dx, part, N, H, W, nseg_per_row, (int)nseg); return launch_status("abl"); }
    ABL_CASE(1) ABL_CASE(2) ABL_CASE(3) ABL_CASE(4) ABL_CASE(5)
#endif
    hipLaunchKernelGGL((convt2x2_bwd_dma_kernel<CI, CO>), dim3(nblocks), dim3(512), kLds, st, x, w, dy, dx, part, N, H, W, nseg_per_row, (int)nseg);
    return launch_status("convt2x2_bwd_dma_kernel");
}

}  // namespace dsrl
