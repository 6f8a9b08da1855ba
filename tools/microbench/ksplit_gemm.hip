// Prototype (not product): "wave-private K slices" bf16x6 GEMM for the M = 4096 layers.
//   y[M][N] = x[M][C] . w[N][C]^T, fp32 in HBM, three bf16 planes per operand formed in registers, six MFMAs per product.
// One block per 64x64 output tile, W waves; wave v owns the 16-deep K steps v, v+W, ... and the WHOLE 64x64 tile for them:
// operands go global -> VGPR in MFMA fragment layout (no LDS, no barrier in the loop; every element is fetched and split exactly once
// per block), NSET register sets deep; the W accumulator sets are summed through LDS in a fixed order at the end.
// Compared with conv_igemm_split_kernel<1,1,2,2,.,3,4> (16 waves, 6 MFMAs per wave between block-wide barriers).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
constexpr unsigned kOOB = 0x80000000u;

__device__ inline float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// three bf16 planes of two float4 (k = 4g.., 8 + 4g..) -> one MFMA operand per plane
__device__ __forceinline__ void split3(const float4 lo, const float4 hi, bf16x8 (&pl)[3]) {
    float r[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        bf16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (__bf16)r[e];
        pl[p] = t;
        if (p < 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] -= (float)t[e];
        }
    }
}

template <int W, int NSET>
__global__ __launch_bounds__(64 * W, 2) void ksplit_gemm(const float* __restrict__ x, int ldx, const float* __restrict__ w, int ldw, float* __restrict__ y,
                                                             int ldy, int M, int N, int C, int ntiles_n) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int row = lane & 31, g = lane >> 5;
    const int m0 = (blockIdx.x / ntiles_n) * 64, n0 = (blockIdx.x % ntiles_n) * 64;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((long long)M * ldx * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)((long long)N * ldw * 4), 0x00020000);
    unsigned off[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + 32 * i + row, k = n0 + 32 * i + row;
        off[i] = m < M ? (unsigned)m * (unsigned)ldx * 4u : kOOB;
        off[2 + i] = k < N ? (unsigned)k * (unsigned)ldw * 4u : kOOB;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nsteps = (C + 15) / 16;
    const int nloc = (nsteps - wave + W - 1) / W;             // steps of this wave
    float4 R[NSET][4][2];
    auto load = [&](int s, int t) {                           // past the end: offsets beyond the descriptor, zeros
        const int c = t * 16 + g * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned coff = (c + 8 * h) < C ? (unsigned)(c + 8 * h) * 4u : kOOB;
#pragma unroll
            for (int v = 0; v < 4; ++v) R[s][v][h] = buf_load4(v < 2 ? xr : wr, off[v] + coff);
        }
    };
    auto compute = [&](int s) {
        bf16x8 fa[2][3], fb[2][3];
        split3(R[s][0][0], R[s][0][1], fa[0]);
        split3(R[s][2][0], R[s][2][1], fb[0]);
        split3(R[s][3][0], R[s][3][1], fb[1]);
        split3(R[s][1][0], R[s][1][1], fa[1]);
#pragma unroll
        for (int sum = 2; sum >= 0; --sum)
#pragma unroll
            for (int pa = 0; pa <= sum; ++pa)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][pa], fb[j][sum - pa], acc[i][j], 0, 0, 0);
    };
#pragma unroll
    for (int s = 0; s < NSET; ++s) load(s, wave + s * W);
    for (int it = 0; it < nloc; it += NSET) {
#pragma unroll
        for (int s = 0; s < NSET; ++s) {
            compute(s);
            load(s, wave + (it + s + NSET) * W);
        }
    }
    // ---- fixed-order sum of the W accumulator sets: red[w][e][lane]; wave v sums registers 8v .. 8v+7 (W = 8) and stores them
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[(wave * 64 + (i * 2 + j) * 16 + e) * 64 + lane] = acc[i][j][e];
    __syncthreads();
    constexpr int PER = 64 / W;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int e64 = wave * PER + q;
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < W; ++v) s += red[(v * 64 + e64) * 64 + lane];
        const int blk = e64 >> 4, e = e64 & 15, i = blk >> 1, j = blk & 1;
        const int m = m0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), k = n0 + 32 * j + (lane & 31);
        if (m < M && k < N) y[(long long)m * ldy + k] = s;
    }
}

template <int W, int NSET>
static float run(const float* x, const float* w, float* y, int M, int N, int C, int reps) {
    const int tm = (M + 63) / 64, tn = (N + 63) / 64;
    const size_t lds = (size_t)W * 64 * 64 * 4;
    hipFuncSetAttribute((const void*)ksplit_gemm<W, NSET>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((ksplit_gemm<W, NSET>), dim3(tm * tn), dim3(64 * W), lds, 0, x, C, w, C, y, N, M, N, C, tn);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (r > 0 && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4096, C = argc > 2 ? atoi(argv[2]) : 1024, N = argc > 3 ? atoi(argv[3]) : 256;
    std::vector<float> hx((size_t)M * C), hw((size_t)N * C), hy((size_t)M * N);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd() * 0.05f;
    float *x, *w, *y;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&w, hw.size() * 4); hipMalloc(&y, hy.size() * 4);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    const double flops = 2.0 * M * N * C;
    struct { const char* name; float ms; } res[4];
    res[0] = {"W=8 NSET=2", run<8, 2>(x, w, y, M, N, C, 6)};
    res[1] = {"W=8 NSET=3", run<8, 3>(x, w, y, M, N, C, 6)};
    res[2] = {"W=4 NSET=3", run<4, 3>(x, w, y, M, N, C, 6)};
    res[3] = {"W=4 NSET=2", run<4, 2>(x, w, y, M, N, C, 6)};
    hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost);       // result of the last configuration
    double maxerr = 0, maxref = 0;
    for (int t = 0; t < 2000; ++t) {
        const int m = (t * 7919) % M, k = (t * 104729) % N;
        double r = 0;
        for (int c = 0; c < C; ++c) r += (double)hx[(size_t)m * C + c] * hw[(size_t)k * C + c];
        maxerr = fmax(maxerr, fabs(r - hy[(size_t)m * N + k])); maxref = fmax(maxref, fabs(r));
    }
    for (auto& r : res) printf("M %d C %d N %d  %s: %.1f us, %.1f TF (%.3f of 417)\n", M, C, N, r.name, r.ms * 1e3, flops / (r.ms * 1e-3) / 1e12, flops / (r.ms * 1e-3) / 417e12);
    printf("max |err| %.3g (max |ref| %.3g) on 2000 samples; %s\n", maxerr, maxref, hipGetLastError() == hipSuccess ? "no HIP error" : "HIP ERROR");
    return maxerr > 1e-4 * fmax(1.0, maxref);
}
