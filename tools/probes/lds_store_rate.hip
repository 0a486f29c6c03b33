// LDS store throughput of one CU by store width (8 waves, every lane its own address, row stride 80 B as in the GEMM
// images):  hipcc --offload-arch=gfx950 -O3 tools/probes/lds_store_rate.hip -o tools/_trace/lds_store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int W>   // bytes per lane and store: 8 or 16
__global__ __launch_bounds__(512) void k(int iters, long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    // GEMM-like addressing: W == 8: row = tid >> 3, k4 = tid & 7 (8 B each);  W == 16: row = tid >> 2, k8 = tid & 3
    const int off = W == 8 ? (tid >> 3) * 80 + (tid & 7) * 8 : ((tid >> 2) % 128) * 80 + (tid & 3) * 16 + (tid >> 9) * 0;
    uint2 a = make_uint2(tid, tid * 3);
    uint4 b = make_uint4(tid, tid * 3, tid * 5, tid * 7);
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (W == 8) *reinterpret_cast<uint2*>(lds + off + u * 5120) = a;
            else *reinterpret_cast<uint4*>(lds + off + (u & 3) * 10240) = b;
            asm volatile("" ::: "memory");
        }
    }
    __syncthreads();
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[W == 8 ? 0 : 1] = t1 - t0;
    sink[tid] = (float)lds[tid];
}
int main() {
    long long* cyc; float* sink;
    hipMalloc(&cyc, 64); hipMalloc(&sink, 4096);
    const int iters = 1000;
    hipLaunchKernelGGL(k<8>, dim3(1), dim3(512), 65536, 0, iters, cyc, sink);
    hipLaunchKernelGGL(k<16>, dim3(1), dim3(512), 65536, 0, iters, cyc, sink);
    hipDeviceSynchronize();
    long long h[2];
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("ds_write_b64 : %.1f B/clk (512 lanes x 8 stores x 8 B per iteration, %lld ticks)\n", 512.0 * 8 * 8 * iters / h[0], h[0]);
    printf("ds_write_b128: %.1f B/clk (512 lanes x 8 stores x 16 B per iteration, %lld ticks)\n", 512.0 * 8 * 16 * iters / h[1], h[1]);
    return 0;
}
