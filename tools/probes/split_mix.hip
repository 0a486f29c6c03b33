// f16x3 operand split (gemm_h3.h store_items_h): the compiler's sequence (v_pk_mul, v_cvt_pk_f16_f32, 2 x v_cvt_f32_f16,
// v_pk_fma, v_cvt_pk_f16_f32 per PAIR = 14 VALU per 4 elements) against a v_fma_mix formulation (h1 = mixlo/mixhi(x s),
// r = fma_mix_f32(x, s, -h1), h2 = mixlo/mixhi(r * 2048): 12 per 4 elements, no conversions back to f32).
// Prints whether the two agree bit for bit and the s_memtime ticks per 4 elements of each (one wave, registers only).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/split_mix.hip -o tools/_trace/split_mix && tools/_trace/split_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_ref(float4 v, float s, uint2& p1, uint2& p2) {
    f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
    lo *= s; hi *= s;
    const f16x2 l1 = __builtin_convertvector(lo, f16x2), h1 = __builtin_convertvector(hi, f16x2);
    p1.x = __builtin_bit_cast(unsigned, l1); p1.y = __builtin_bit_cast(unsigned, h1);
    lo = (lo - __builtin_convertvector(l1, f32x2)) * 2048.f;
    hi = (hi - __builtin_convertvector(h1, f32x2)) * 2048.f;
    const f16x2 l2 = __builtin_convertvector(lo, f16x2), h2 = __builtin_convertvector(hi, f16x2);
    p2.x = __builtin_bit_cast(unsigned, l2); p2.y = __builtin_bit_cast(unsigned, h2);
}

__device__ __forceinline__ unsigned mix_pair_h1(float a, float b, float s) {       // -> (f16(a s), f16(b s))
    unsigned d = 0;
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]\n\t"
                 "v_fma_mixhi_f16 %0, %3, %2, 0 op_sel_hi:[0,0,0]" : "+v"(d) : "v"(a), "v"(s), "v"(b));
    return d;
}
__device__ __forceinline__ unsigned mix_pair_h2(float a, float b, float s, unsigned h1) {
    float r0, r1;
    // r = a * s - h1.lo / h1.hi  (src2 read as f16: op_sel_hi[2] = 1; op_sel[2] picks the half; the minus sign is the neg modifier)
    asm volatile("v_fma_mix_f32 %0, %2, %3, -%4 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
                 "v_fma_mix_f32 %1, %5, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                 : "=&v"(r0), "=&v"(r1) : "v"(a), "v"(s), "v"(h1), "v"(b));
    unsigned d = 0;
    const float k = 2048.f;
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]\n\t"
                 "v_fma_mixhi_f16 %0, %3, %2, 0 op_sel_hi:[0,0,0]" : "+v"(d) : "v"(r0), "v"(k), "v"(r1));
    return d;
}
__device__ __forceinline__ void split_mix(float4 v, float s, uint2& p1, uint2& p2) {
    p1.x = mix_pair_h1(v.x, v.y, s); p1.y = mix_pair_h1(v.z, v.w, s);
    p2.x = mix_pair_h2(v.x, v.y, s, p1.x); p2.y = mix_pair_h2(v.z, v.w, s, p1.y);
}

template <int MODE>
__global__ __launch_bounds__(64) void k(const float4* x, float s, uint2* o, long long* cyc, int iters) {
    float4 v[8];
    for (int i = 0; i < 8; ++i) v[i] = x[threadIdx.x + 64 * i];
    uint2 a1[8], a2[8];
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) split_ref(v[i], s, a1[i], a2[i]); else split_mix(v[i], s, a1[i], a2[i]);
            v[i].x += __uint_as_float(a2[i].x & 1u) * 1e-30f;      // keep the loop from collapsing
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; ++i) { o[(threadIdx.x + 64 * i) * 2] = a1[i]; o[(threadIdx.x + 64 * i) * 2 + 1] = a2[i]; }
    if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}

int main() {
    const int n = 512;
    float4* x; uint2 *o0, *o1; long long* cyc;
    hipMallocManaged(&x, n * sizeof(float4)); hipMallocManaged(&o0, 2 * n * sizeof(uint2));
    hipMallocManaged(&o1, 2 * n * sizeof(uint2)); hipMallocManaged(&cyc, 16);
    srand(1);
    for (int i = 0; i < n * 4; ++i) ((float*)x)[i] = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * (i % 7 == 0 ? 1e-6f : 1.f);
    const float s = 16384.f;
    const int iters = 1;            // 1 for the comparison (the anti-collapse perturbation would diverge), more for timing
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, x, s, o0, cyc, iters);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, x, s, o1, cyc, iters);
    hipDeviceSynchronize();
    int bad = 0;
    for (int i = 0; i < 2 * n; ++i) bad += (o0[i].x != o1[i].x) + (o0[i].y != o1[i].y);
    printf("words that differ between the two formulations: %d of %d\n", bad, 4 * n);
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, x, s, o0, cyc, 2000);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, x, s, o1, cyc, 2000);
    hipDeviceSynchronize();
    printf("compiler sequence: %.1f ticks per 4 elements;  fma_mix sequence: %.1f\n", cyc[0] / (2000.0 * 8), cyc[1] / (2000.0 * 8));
    return 0;
}
