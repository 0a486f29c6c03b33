// Issue rate of the 32x32x16 MFMAs of gfx950 by input type (one wave per SIMD, independent accumulators, operands in
// registers):  hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate.hip -o tools/_trace/mfma_rate && tools/_trace/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, float* out, long long* cyc) {
    f32x16 acc[4];
    f32x4 acc4[4];
    for (int i = 0; i < 4; ++i) {
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f16x8 ha, hb;
    bf16x8 ba, bb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(threadIdx.x * 0.001f + i); hb[i] = (_Float16)(1.f + i); ba[i] = (__bf16)(threadIdx.x * 0.001f + i); bb[i] = (__bf16)(1.f + i); }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (MODE == 0) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, acc[u & 3], 0, 0, 0);
            if constexpr (MODE == 1) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[u & 3], 0, 0, 0);
            if constexpr (MODE == 2) acc4[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc4[u & 3], 0, 0, 0);
            if constexpr (MODE == 3) acc4[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc4[u & 3], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) { for (int r = 0; r < 16; ++r) s += acc[i][r]; s += acc4[i][0]; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 64);
    const int iters = 2000;
    const char* names[4] = {"32x32x16 bf16", "32x32x16 f16", "16x16x32 f16", "16x16x32 bf16"};
    for (int grid = 1; grid <= 1024; grid *= 1024) {
        hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, iters, out, cyc);
        hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, iters, out, cyc);
        hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, iters, out, cyc);
        hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, iters, out, cyc);
        hipDeviceSynchronize();
        long long h[8];
        hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        for (int m = 0; m < 4; ++m)
            printf("grid %4d  %-14s %7.2f s_memtime ticks per MFMA (one wave per SIMD)\n", grid, names[m], (double)h[m] / (iters * 8.0));
    }
    // wall-clock rate with the whole chip busy
    for (int m = 0; m < 2; ++m) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (m == 0) hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, iters * 4, out, cyc);
        else hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, iters * 4, out, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2048.0 * 4 * iters * 4 * 8 * 2.0 * 32 * 32 * 16;
        printf("%-14s full chip: %.1f TFLOP/s\n", names[m], flops / (ms * 1e-3) / 1e12);
    }
    return 0;
}
