// Device helpers and argument blocks shared by the implicit-GEMM conv kernels (conv_igemm.hip: register-staged operands,
// conv_planes.hip: operands as fp16 planes staged by LDS-DMA).
#pragma once
#include "common.h"
#include <stdlib.h>

namespace dsrl {

static inline int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }   // tuning knobs (tools/sweep_conv.py)

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;

// Split-precision arithmetic ("bf16x3": NPL = 2 planes, "bf16x6": NPL = 3): an fp32 value is carried as NPL bf16 terms
// x = t0 + t1 (+ t2), t0 = bf16(x), t1 = bf16(x - t0), t2 = bf16(x - t0 - t1), i.e. 16 (24) mantissa bits, and a product is the sum
// of the bf16 MFMAs a_i * b_j with i + j < NPL (3 or 6 of them), small terms first, all accumulated in fp32
// (conv_igemm_split_kernel, conv_wgrad_split_kernel).
//
// "f16x3" (F16 = true, NPL = 2, round 3): the two terms are fp16 instead of bf16, t0 = f16(x * 2^e), t1 = f16(x * 2^e - t0), i.e. 22 mantissa
// bits in 3 MFMAs (v_mfma_f32_32x32x16_f16: a0*b1 + a1*b0 + a0*b0) - the accuracy of bf16x6 / fp32 MFMA at half the matrix work and
// two thirds of the LDS bytes.  fp16 has 5 exponent bits, so every operand tensor carries a power-of-two scale: 2^e maps the tensor's
// largest magnitude into [2^14, 2^15) (e from a device word holding max |x| as its bit pattern, written by the tensor's producer or by
// amax_kernel), and the accumulators are multiplied by 2^-(ea + eb) in the epilogue - exact.  Elements more than 2^18 below the
// tensor's maximum lose low-order bits gracefully (absolute error <= 2^-40 of the maximum: invisible in any sum an fp32 kernel forms).
template <bool F16> struct Plane;
template <> struct Plane<false> {
    using v4 = bf16x4; using v8 = bf16x8;
    static __device__ __forceinline__ v4 cvt(const float* r) { return v4{(__bf16)r[0], (__bf16)r[1], (__bf16)r[2], (__bf16)r[3]}; }
    static __device__ __forceinline__ void residual(float* r, v4 t) {
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] -= (float)t[e];
    }
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
// fp16 split of one staged float4 in 8 vector instructions behind the scaling: two v_cvt_pk_f16_f32 per term and the residual r - t as ONE
// v_fma_mix_f32 per element (t * -1.0 + r, t read as the low / high fp16 half of its packed word: exact, like the subtraction it replaces;
// hipcc unpacks t with four v_cvt_f32_f16 and re-converts r twice when the same is written with casts).
__device__ __forceinline__ float f16_resid_lo(unsigned t, float r) { float d; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(t), "v"(r)); return d; }
__device__ __forceinline__ float f16_resid_hi(unsigned t, float r) { float d; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(t), "v"(r)); return d; }
template <> struct Plane<true> {
    using v4 = f16x4; using v8 = f16x8;
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
    static __device__ __forceinline__ v4 cvt(const float* r) { return __builtin_convertvector((f32x4{r[0], r[1], r[2], r[3]}), v4); }
    static __device__ __forceinline__ void residual(float* r, v4 t) {
        const u32x2 u = __builtin_bit_cast(u32x2, t);
        r[0] = f16_resid_lo(u.x, r[0]); r[1] = f16_resid_hi(u.x, r[1]); r[2] = f16_resid_lo(u.y, r[2]); r[3] = f16_resid_hi(u.y, r[3]);
    }
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
// e with max|x| * 2^e in [2^14, 2^15) from the bit pattern of max|x| (zero / subnormal maxima count as 2^-126; Inf / NaN maxima give a
// finite e: such tensors turn into NaN in the split by themselves)
__device__ __forceinline__ int amax_shift_of(unsigned fetched) {      // fetched: this lane's amax_fetch() of the record
    const unsigned u = amax_reduce(fetched);
    int ex = (int)((u >> 23) & 0xffu);
    if (ex == 0) ex = 1;
    return 14 - (ex - 127);
}
__device__ __forceinline__ int amax_shift(const unsigned* p) { return amax_shift_of(amax_fetch(p)); }

// q = n / d for 0 <= n < 2^31: m = ceil(2^(31+l) / d) with l = ceil(log2 d) >= 1, q = mulhi(n, m) >> (l - 1), exact; d = 1 is
// flagged by shift 255 (host side: make_magic)
__device__ __forceinline__ int fast_div(int n, unsigned m, unsigned s) {       // a select, not a branch: it sits inside the K loops (scalar and vector)
    const int q = (int)(__umulhi((unsigned)n, m) >> (s & 31u));
    return s == 255u ? n : q;
}

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
    unsigned x_bytes, w_bytes, y_bytes;   // extents of the three buffers (< 2^31): buffer loads/stores bounds-check against them
    int mtiles, ntiles, xcd_remap;        // 1-D launch of mtiles*ntiles blocks (x splits in z); XCD-aware tile order when xcd_remap
    int kg;                               // split kernels: K groups per block (1, 2 or 4)
    int accumulate;                       // epilogue: y += result (only without slabs: a split-K launch accumulates in its reduce)
    float* stats;                         // split kernels, forward: per (row block, channel) BatchNorm partials (n, mean, M2) of the output, or null
    // split kernels, dgrad: the output is the gradient w.r.t. the output y = relu(bn(x)) of a BatchNorm; the epilogue leaves the
    // partial sums of g = dy * [y > 0] and g * xhat per (row block, channel) in bstats [2][parts][K] (null: off)
    const float* bn_x; const float* bn_y; const float* bn_mean; const float* bn_invstd; float* bstats;
    int bn_ldx, bn_ldy, bn_relu;
    float bn_gscale;                      // factor on the masked gradient: 1 / (1 - p) when a Dropout sits behind the BatchNorm's ReLU (round 5), else 1
    int bn_fast;                          // KG == 1 builds: the BatchNorm-backward sums through LDS with 16-byte loads of x / y (host: alignment, K % 4 == 0, no parity order)
    // split kernels, dgrad of a strided conv (par = stride > 1, else 0): the GEMM rows are ordered by parity class - M-tile t (pbm rows) holds
    // rows (t / par^2) * pbm ... of class (ph, pw) = t % par^2, a class row j being pixel (n, hh*par + ph, wh*par + pw), (n, hh, wh) = j over the
    // Hh x Wh grid of the class - so that all rows of a tile share the taps that divide evenly (a row-major tile multiplies zeros for the
    // other par^2 - 1 of par^2 taps per row), and consecutive tiles cycle through the classes (their tap counts differ: 4, 2, 2, 1 of 9).
    // Host side: Ho % par == Wo % par == 0, pixels per class % pbm == 0 (a tile never straddles classes), Wh % 32 == 0 (32 consecutive rows
    // are 32 consecutive pixels of one image row of the class: the epilogue derives their offsets from the first one).
    int par, Hh, Wh, pbm;
    const unsigned* amax_a; const unsigned* amax_b;     // f16x3: amax records of the input tensor x and of the filter w
    int w_split;                                        // f16x3: `w` is the pre-split filter (kernel ARITH = 2)
    unsigned mHW, sHW, mW, sW, mNT, sNT;                // split kernels: magic numbers of the divisions by Ho*Wo, Wo and ntiles (fast_div; host: make_magic)
    // conv_planes_kernel: x / w point at the FIRST fp16 plane of the operand ([P][ldx] resp. [K][R][S][C] fp16, ldx in fp16 elements, x_bytes / w_bytes =
    // extent of one plane); the second plane lies a_lo / b_lo bytes behind it (unused with one plane)
    unsigned a_lo, b_lo;
    int planes;                                         // host side: route the launch to conv_planes_kernel
    // split-K ACROSS workgroups with the reduction inside the launch (conv_sk.hip, template COOP): the `splits` blocks of a tile (blockIdx.z) publish
    // their partial tiles in coop_slab [z][tile][...], take a ticket, and the block whose ticket is the last one adds them in the order z = 0 .. splits-1
    // and runs the whole epilogue (bias, accumulate, BatchNorm partials) on y.  tickets: one zeroed word per tile, left zero by the last arriver.
    float* coop_slab; unsigned* tickets;
};
// Arrival tickets of the cooperative split-K launches live in the words of the ACTIVATION operand's amax record that carry no shard: a record is
// kAmaxWords words of which every kAmaxShardStride-th holds a partial maximum; the other 240 are zero whenever the record is (arena fill, scratch
// fill) and nothing else touches them.  Ticket t < kCoopMaxTiles sits at word 16 * (t / 15) + 1 + t % 15.  The launches of a stream run one after the
// other and each leaves its tickets zero, so the convs that share an input tensor share its tickets.
constexpr int kCoopMaxTiles = (kAmaxShardStride - 1) * kAmaxShards;
__host__ __device__ inline int coop_ticket_word(int t) { return kAmaxShardStride * (t / (kAmaxShardStride - 1)) + 1 + t % (kAmaxShardStride - 1); }
__device__ __forceinline__ int dgrad_pix(const ConvArgs& a, int m) {        // row of the parity-ordered GEMM -> pixel index (n*Ho + h)*Wo + w
    const int t = m / a.pbm, p2 = a.par * a.par, c = t % p2, j = (t / p2) * a.pbm + (m - t * a.pbm);
    const int hw = a.Hh * a.Wh, n = j / hw, r = j - n * hw, hh = r / a.Wh, wh = r - hh * a.Wh;
    const int ph = c / a.par, pw = c - ph * a.par;
    return (n * a.Ho + hh * a.par + ph) * a.Wo + wh * a.par + pw;
}

// Blocks are dealt round-robin over the 8 XCDs (private L2 each). Remap the linear block id so that every XCD works on a contiguous
// range of tile ids (bijective for any count): tiles that share input rows then hit the same L2. Speed only, never correctness.
__device__ inline int xcd_contiguous(int bid, int n) {
    const int q = n / kNumXCD, r = n % kNumXCD, xcd = bid % kNumXCD, local = bid / kNumXCD;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
constexpr unsigned kOOB = 0x80000000u;      // any offset >= 2^31 is outside every descriptor: loads return 0, stores are dropped
__device__ inline float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

struct WgradArgs {
    const float* x; const float* dy; float* dw;     // dw or slabs
    int ldx, lddy;
    int N, H, W, C, K, R, S, Ho, Wo, stride, pad, dil;
    long long P;                // N*Ho*Wo
    unsigned x_bytes, dy_bytes; // buffer extents (< 2^31)
    int ctiles;                 // tiles along C
    int psplits;
    long long slab;             // floats per split slab (K*RS*C) when psplits > 1
    int taps[64]; int ntaps;    // active filter taps (whole-tensor)
    int xcd_remap;              // pixel-range-major block order per XCD (blocks of one pixel range share dy / x chunks)
    int kctiles;                // ktiles * ctiles
    unsigned mHW, sHW, mW, sW;  // magic multipliers / shifts: p / (Ho*Wo) and rem / Wo for p < 2^31 (fast_div)
    int kg;                     // split kernel: pixel groups per block (1, 2 or 4)
    // grouped launches (dsrl_conv2d_wgrad_group_*): blocks of this problem (the rest up to the next start are padding), where the
    // slab reduce writes, and that reduce's block count
    int nblocks;
    float* dw_final;
    int rblocks;
    const unsigned* amax_dy; const unsigned* amax_x;        // f16x3: max |.| (bit patterns) of dy and of x
    int no_ident;               // DSRL_WGRAD_IDENT=0: the general per-row addressing also for taps that read pixel p for pixel p (A/B knob)
};

// conv_wgrad3.hip: 3x3 / stride-1 weight gradients with all nine taps in one block (f16x3 / f16x1; tile 64 out channels x 64 in channels: wgrad3_tile()).
// WgradArgs as for conv_wgrad_split_kernel with ntaps = 9, kctiles = tiles of that size, nblocks = kctiles * psplits.
bool wgrad3_eligible(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil, int npl, bool f16);
void wgrad3_tile(int& bm, int& bn);
int launch_wgrad3(const WgradArgs& a, int npl, hipStream_t st);
int launch_wgrad3_group(const WgradArgs* table, const int* starts, int nprob, int grid, int npl, int dil, hipStream_t st);

// conv_planes.hip: the same implicit GEMM with both operands as fp16 planes staged by LDS-DMA (a.x / a.w / a.a_lo / a.b_lo as described in ConvArgs);
// cfg is conv_igemm.hip's TileCfg, the tile / K-group / split-K plan (and with it the summation order) is the caller's.
bool planes_cfg_supported(int cfg, int kg);
int launch_planes_igemm(const ConvArgs& a, int cfg, bool dgrad, hipStream_t st);
// conv_sk.hip: 128x128 tiles, two K groups (8 waves), f16x3 / f16x1 with pre-split filters; a.splits > 1 with a.tickets set = cooperative split-K
bool sk_supported(int cfg, int kg, int npl, bool f16, bool w_split);
size_t sk_lds_bytes();
int launch_sk_igemm(const ConvArgs& a, bool dgrad, bool str1, int npl, hipStream_t st);
__host__ __device__ inline long long planes_lo_offset(long long elems) { return (elems * 2 + 255) / 256 * 256; }      // bytes from the first to the second plane

}  // namespace dsrl
