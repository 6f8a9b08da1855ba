// Shared host/device helpers of libdsrl_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/dsrl_hip.h"

namespace dsrl {

// ---------------------------------------------------------------- error reporting (thread-local)
void set_error(const char* fmt, ...);
int launch_status(const char* what);

#define DSRL_REQUIRE(cond, code, ...)        \
    do {                                     \
        if (!(cond)) {                       \
            ::dsrl::set_error(__VA_ARGS__);  \
            return (code);                   \
        }                                    \
    } while (0)

// Binds the calling thread to the device that owns `stream` (autograd runs backward on its own thread).
int bind_stream_device(hipStream_t s);

// convt_dma.hip: ConvTranspose2d k2 s2 backward staged by LDS-DMA (19 -> 19 channels, W % 128 == 0)
bool convt_bwd_dma_supported(const void* x, const void* dy, int W, int Cin, int Cout);
int convt_bwd_dma_blocks(long long nseg, int cap);
int launch_convt_bwd_dma(const float* x, const float* w, const float* dy, float* dx, float* part, int N, int H, int W, int nblocks, hipStream_t st);
int launch_convt_bwd_dma_ce(const float* x, const float* w, const float* logits, float* dx, float* part, int N, int H, int W, int nblocks,
                            const unsigned char* target, int ignore_index, const float* count, const float* ft_g, const float* ft_w, int ft_s, hipStream_t st);

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

constexpr int kWave = 64;
constexpr int kNumCU = 256;     // MI355X
constexpr int kNumXCD = 8;

// ---------------------------------------------------------------- Philox4x32-10 (same as oracle/philox.py)
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// uniform in [0,1) for element e: word (e&3) of philox(counter = e>>2, stream), u = (word>>8) * 2^-24
__device__ inline float philox_uniform(uint64_t e, uint64_t seed, uint32_t stream) {
    uint32_t r[4];
    const uint64_t q = e >> 2;
    philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    return (float)(r[e & 3] >> 8) * 5.9604644775390625e-08f;
}

// ---------------------------------------------------------------- exp(x) for x <= 0 (softmax terms exp(v - max))
// v_exp_f32 on t = fl(x * log2 e) with the rounding error of that product and the low part of log2 e put back to first order:
// exp(x) = 2^t * 2^r, r = (x * L2E_HI - t) + x * L2E_LO exactly (one fma each), 2^r = 1 + r ln 2 + O(r^2), |r| < 2^-23 |t|.  Six instructions instead
// of libm's ~20 (range reduction + polynomial + overflow / denormal cases, none of which a non-positive argument needs); within 2 ulp of expf for
// x in [-87, 0], 0 below (the hardware flushes the denormal result).  ce_fused_kernel and the ConvTranspose backward that forms the CE gradient
// itself (convt_dma.hip) both use it: their results are bit-identical.
__device__ __forceinline__ float exp_nonpos(float x) {
    constexpr float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-08f, LN2 = 0.693147182464599609375f;
    const float t = x * L2E_HI;
    const float r = fmaf(x, L2E_LO, fmaf(x, L2E_HI, -t));
    const float e0 = __builtin_amdgcn_exp2f(t);
    return fmaf(e0, r * LN2, e0);
}

// ---------------------------------------------------------------- wave / block reductions
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------- operand magnitudes (f16x3 conv arithmetic, dsrl_amax)
// A kernel that writes a tensor a conv will read can leave max |x| of what it wrote (as a bit pattern; NaN patterns sort above every number)
// in a device record: per thread a running maximum over its stores, then amax_publish - one fire-and-forget atomic per block.
// A record ("amax record", kAmaxWords uint32) holds kAmaxShards partial maxima, one per 64-byte line: the ~1000 blocks of a streaming
// kernel finish together, and their atomics on ONE word would be served one after the other (~12 ns each: measured +9 us on a 7 us
// BatchNorm launch); spread over 16 lines they cost nothing visible.  The reader takes the maximum of the shards (amax_read).
// Every thread of the block must reach amax_publish (it synchronises); `out` may be null.
constexpr int kAmaxShards = 16, kAmaxShardStride = 16, kAmaxWords = kAmaxShards * kAmaxShardStride;     // 1 KiB per record
__device__ inline unsigned abs_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }
__device__ inline unsigned* amax_shard(unsigned* rec) { return rec + (((blockIdx.x + 5u * blockIdx.y) & (kAmaxShards - 1)) * kAmaxShardStride); }
// maximum over the shards of a record, wave-uniform (every lane of the wave must be active).  Two halves so that a kernel can request the
// shards at its very start (amax_fetch: one load per lane, no wait) and consume them where the value is first needed (amax_reduce): the conv
// kernels used to sit through two dependent load latencies - one per operand record - before they computed a single address.
__device__ inline unsigned amax_fetch(const unsigned* rec) {
    const int lane = threadIdx.x & 63;
    return lane < kAmaxShards ? rec[lane * kAmaxShardStride] : 0u;
}
__device__ inline unsigned amax_reduce(unsigned m) {
#pragma unroll
    for (int o = kAmaxShards / 2; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    return __builtin_amdgcn_readfirstlane(m);
}
__device__ inline unsigned amax_read(const unsigned* rec) { return amax_reduce(amax_fetch(rec)); }
__device__ inline unsigned abs_bits4(unsigned m, float a, float b, float c, float d) {
    return max(max(m, abs_bits(a)), max(max(abs_bits(b), abs_bits(c)), abs_bits(d)));
}
__device__ inline void amax_publish(unsigned m, unsigned* out) {
    if (out == nullptr) return;
    __shared__ unsigned sm_amax;
    if (threadIdx.x == 0) sm_amax = 0u;
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(&sm_amax, m);
    __syncthreads();
    if (threadIdx.x == 0 && sm_amax) __hip_atomic_fetch_max(amax_shard(out), sm_amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the same measurement as a launch of its own (conv_igemm.hip)
int launch_amax(const float* x, int ld, long long P, int C, unsigned* out, hipStream_t st);
// Zero-fill as a KERNEL launch (conv_igemm.hip), not hipMemsetAsync.  Round 3: with the 2 KiB amax scratch of the stem's weight gradient zeroed by a
// captured memset node, ~40 % of the replays of the two-graph step produced a NaN there, and replacing that one call by a kernel made it disappear.
// Round 4 dumped the captured graphs of a build with the memsets restored (DSRL_ZERO_FILL_MEMSET=1, tools/graph_memset_edges.py,
// profiles/round4_graph_memset_edges.txt): both graphs are pure chains and every memset node has its edge to the kernel that consumes the zeroed
// words - the capture did NOT drop a dependency.  Why the replay misbehaved is therefore not established (a runtime fault in how memset nodes execute,
// or a cause the substitution only perturbed); the kernel fill is kept because it leaves kernel nodes as the only node type of the step's graphs.
int launch_zero_fill(void* p, size_t bytes, hipStream_t st);

// Per-channel thread mapping for pixel-major [P][ld] tensors with C channels (channel group of <= 256):
// G = 256 / cg pixels are processed side by side, thread t < G*cg owns channel (t % cg) of pixel slot (t / cg).
struct ChanMap {
    int cg0;     // first channel of this block's group
    int cg;      // channels in the group (<= 256)
    int G;       // pixel slots
    int c;       // my channel (absolute), -1 if idle
    int slot;    // my pixel slot
};
__device__ inline ChanMap chan_map(int C, int group_idx) {
    ChanMap m;
    m.cg0 = group_idx * 256;
    m.cg = min(256, C - m.cg0);
    m.G = 256 / m.cg;
    const int t = threadIdx.x;
    if (t < m.G * m.cg) { m.c = m.cg0 + t % m.cg; m.slot = t / m.cg; }
    else { m.c = -1; m.slot = 0; }
    return m;
}

}  // namespace dsrl
