// What issues for free in the shadow of a v_mfma_f32_32x32x16_bf16 (one wave per SIMD)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fill_probe.hip -o tools/_trace/fill_probe
// loop of 16 x [ MFMA ; NF fillers of one kind ]; prints shader cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int J>
__device__ __forceinline__ void filler(float (&t)[8], f32x2 (&p)[4], unsigned (&u)[8], int lds_addr) {
    if constexpr (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(t[J & 7]) : "v"(t[(J + 4) & 7]));
    if constexpr (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[J & 7]) : "v"(t[J & 7]), "v"(t[(J + 1) & 7]));
    if constexpr (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[J & 3]) : "v"(p[(J + 2) & 3]));
    if constexpr (KIND == 3) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u[J & 7]) : "v"(u[(J + 3) & 7]));
    if constexpr (KIND == 4) asm volatile("ds_write_b64 %0, %1" :: "v"(lds_addr), "v"(p[J & 3]) : "memory");
    if constexpr (KIND == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(t[0]) : "v"(t[1]));          // dependent chain
    if constexpr (KIND == 6) asm volatile("s_nop 0");
    if constexpr (KIND == 7) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[J & 7]) : "v"(u[(J + 3) & 7]));
    if constexpr (KIND == 8) asm volatile("ds_read_b128 %0, %1" : "=v"(*reinterpret_cast<float4*>(&p[0])) : "v"(lds_addr) : "memory");
    if constexpr (KIND == 9) {      // the dependent micro-step of the split: and, shl, pk_sub (reads the previous result)
        if constexpr (J % 3 == 0) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u[1]) : "v"(u[0]));
        if constexpr (J % 3 == 1) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[2]) : "v"(u[0]));
        if constexpr (J % 3 == 2) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[0]) : "v"(*reinterpret_cast<f32x2*>(&u[2])));
    }
}

template <int KIND, int NF>
__global__ __launch_bounds__(256) void fill_kernel(int iters, float* out, long long* cyc) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)1.0f; fb[i] = (__bf16)0.5f; }
    float t[8];
    f32x2 p[4];
    unsigned u[8];
    for (int i = 0; i < 8; ++i) { t[i] = 1.f + threadIdx.x + i; u[i] = threadIdx.x * 77u + i; }
    for (int i = 0; i < 4; ++i) p[i] = f32x2{t[i], t[i + 4]};
    const int lds_addr = (threadIdx.x & 255) * 16;
    lds[threadIdx.x] = 0.f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[m & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NF > 0) filler<KIND, 0>(t, p, u, lds_addr);
            if constexpr (NF > 1) filler<KIND, 1>(t, p, u, lds_addr);
            if constexpr (NF > 2) filler<KIND, 2>(t, p, u, lds_addr);
            if constexpr (NF > 3) filler<KIND, 3>(t, p, u, lds_addr);
            if constexpr (NF > 4) filler<KIND, 4>(t, p, u, lds_addr);
            if constexpr (NF > 5) filler<KIND, 5>(t, p, u, lds_addr);
            if constexpr (NF > 6) filler<KIND, 6>(t, p, u, lds_addr);
            if constexpr (NF > 7) filler<KIND, 7>(t, p, u, lds_addr);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += t[i] + (float)u[i];
    for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int NF>
void run(const char* name, float* out, long long* cyc) {
    const int iters = 500;
    hipLaunchKernelGGL((fill_kernel<KIND, NF>), dim3(256), dim3(256), 0, 0, 10, out, cyc);
    hipLaunchKernelGGL((fill_kernel<KIND, NF>), dim3(256), dim3(256), 0, 0, iters, out, cyc);
    hipDeviceSynchronize();
    long long h;
    hipMemcpy(&h, cyc + 4 * 17, 8, hipMemcpyDeviceToHost);
    printf("%-34s NF=%d  %6.1f cycles per MFMA\n", name, NF, (double)h / iters / 16);
}

template <int KIND>
void run_kind(const char* name, float* out, long long* cyc) {
    run<KIND, 2>(name, out, cyc);
    run<KIND, 4>(name, out, cyc);
    run<KIND, 6>(name, out, cyc);
    run<KIND, 8>(name, out, cyc);
}

int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 4 * 8);
    run<0, 0>("MFMA only", out, cyc);
    run_kind<0>("v_add_f32 independent", out, cyc);
    run_kind<5>("v_add_f32 dependent chain", out, cyc);
    run_kind<1>("v_cvt_pk_bf16_f32", out, cyc);
    run_kind<2>("v_pk_add_f32", out, cyc);
    run_kind<3>("v_and_b32 (literal)", out, cyc);
    run_kind<7>("v_lshlrev_b32", out, cyc);
    run_kind<9>("and/shl/pk_sub dependent", out, cyc);
    run_kind<4>("ds_write_b64", out, cyc);
    run_kind<8>("ds_read_b128", out, cyc);
    run_kind<6>("s_nop 0", out, cyc);
    return 0;
}
