// Microbenchmark (not product): throughput of staging [rows x 128 B] operand tiles from an L2 / Infinity-Cache resident tensor into LDS on gfx950,
// for the access patterns discussed in DESIGN.md section 6 ("what bounds the conv kernels"):
//   mode 0  VGPR staging, 4 lanes per row, the two 64-byte halves of a row in two instructions (what conv_igemm_split_kernel does today);
//   mode 1  VGPR staging, 8 lanes per row: one instruction covers whole 128-byte lines;
//   mode 2  LDS-DMA (global_load_lds_dwordx4), 8 lanes per row, no VGPR round trip;
//   mode 3  LDS-DMA, 4 lanes per row (half lines).
// Every block stages `rows` rows per step for `steps` steps (row stride `ld` floats, the tile start advances through the tensor), sums a few LDS
// words so that nothing is optimised away, and the host reports GB/s per CU and chip-wide.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
typedef __attribute__((address_space(3))) void lds_void;

template <int MODE>
__global__ __launch_bounds__(256, 2) void stage_kernel(const float* __restrict__ x, long long rows_total, int ld, int rows, int steps, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];            // rows * 32 floats per stage, 2 stages
    const int tid = threadIdx.x;
    float acc = 0.f;
    const long long tile_stride = rows;      // consecutive tiles
    long long row0 = ((long long)blockIdx.x * 9973) % (rows_total - (long long)rows * 2);
    for (int s = 0; s < steps; ++s) {
        float* st = smem + (s & 1) * rows * 32;
        if (MODE == 0) {
            const int c4 = tid & 3, r = tid >> 2;               // 64 rows per pass
            for (int p = 0; p < rows; p += 64) {
                const float* src = x + (row0 + p + r) * ld;
                const float4 a = *reinterpret_cast<const float4*>(src + c4 * 4);
                const float4 b = *reinterpret_cast<const float4*>(src + 16 + c4 * 4);
                *reinterpret_cast<float4*>(st + (p + r) * 32 + c4 * 4) = a;
                *reinterpret_cast<float4*>(st + (p + r) * 32 + 16 + c4 * 4) = b;
            }
        } else if (MODE == 1) {
            const int c8 = tid & 7, r = tid >> 3;               // 32 rows per pass
            for (int p = 0; p < rows; p += 32) {
                const float4 a = *reinterpret_cast<const float4*>(x + (row0 + p + r) * ld + c8 * 4);
                *reinterpret_cast<float4*>(st + (p + r) * 32 + c8 * 4) = a;
            }
        } else if (MODE == 2) {
            const int c8 = tid & 7, r = tid >> 3;
            const int wave = tid >> 6;
            for (int p = 0; p < rows; p += 32) {
                // one wave-instruction writes 64 lanes x 16 B = 8 rows x 128 B, lane-linear: LDS base must be wave-uniform
                float* dst = st + (p + wave * 8) * 32;
                __builtin_amdgcn_global_load_lds((const void*)(x + (row0 + p + r) * ld + c8 * 4), (lds_void*)dst, 16, 0, 0);
            }
        } else {
            const int c4 = tid & 3, r = tid >> 2;
            const int wave = tid >> 6;
            for (int p = 0; p < rows; p += 64) {
                float* dst0 = st + (p + wave * 16) * 32;        // lane-linear images: 16 rows x 64 B per instruction, two instructions per 16 rows
                __builtin_amdgcn_global_load_lds((const void*)(x + (row0 + p + r) * ld + c4 * 4), (lds_void*)dst0, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const void*)(x + (row0 + p + r) * ld + 16 + c4 * 4), (lds_void*)(dst0 + 16 * 16), 16, 0, 0);
            }
        }
        if (MODE >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += st[(tid * 33) % (rows * 32)];
        row0 += tile_stride;
        if (row0 + rows * 2 >= rows_total) row0 = 0;
    }
    if (acc == 123.456f) out[blockIdx.x] = acc;
}

int main(int argc, char** argv) {
    const int ld = argc > 1 ? atoi(argv[1]) : 304, rows = argc > 2 ? atoi(argv[2]) : 256, steps = 512;
    const long long rows_total = (long long)(argc > 3 ? atoi(argv[3]) : 64) * 1024 * 1024 / (ld * 4);     // tensor of N MiB
    float *x, *out;
    hipMalloc(&x, rows_total * ld * 4); hipMalloc(&out, 4096 * 4);
    hipMemset(x, 1, rows_total * ld * 4);
    const int blocks = 512;
    const size_t lds = (size_t)2 * rows * 32 * 4;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(a, 0);
            if (mode == 0) hipLaunchKernelGGL(stage_kernel<0>, dim3(blocks), dim3(256), lds, 0, x, rows_total, ld, rows, steps, out);
            if (mode == 1) hipLaunchKernelGGL(stage_kernel<1>, dim3(blocks), dim3(256), lds, 0, x, rows_total, ld, rows, steps, out);
            if (mode == 2) hipLaunchKernelGGL(stage_kernel<2>, dim3(blocks), dim3(256), lds, 0, x, rows_total, ld, rows, steps, out);
            if (mode == 3) hipLaunchKernelGGL(stage_kernel<3>, dim3(blocks), dim3(256), lds, 0, x, rows_total, ld, rows, steps, out);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep > 0 && ms < best) best = ms;
        }
        const double bytes = (double)blocks * steps * rows * 128.0;
        printf("ld %d rows %d tensor %lld MiB mode %d: %.3f ms, %.2f TB/s chip, %.1f GB/s per CU (%s)\n", ld, rows, rows_total * ld * 4 / (1 << 20), mode, best,
               bytes / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e9 / 256,
               mode == 0 ? "VGPR, half lines" : mode == 1 ? "VGPR, full lines" : mode == 2 ? "LDS-DMA, full lines" : "LDS-DMA, half lines");
    }
    return hipGetLastError() != hipSuccess;
}
