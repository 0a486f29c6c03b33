// Micro-probe of the bf16x6 GEMM's two phases in isolation (no global memory in the loop):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/mfma_probe.hip -o tools/_trace/mfma_probe
// mode 0: 48 MFMAs / iteration, operands in registers
// mode 1: mfma_tile (24 ds_read_b128 + 48 MFMAs) / iteration
// mode 2: store_items x2 (split 8 float4 items into 3 bf16 planes + LDS stores) / iteration
// mode 3: waves 0-3 run mode 1, waves 4-7 run mode 2 concurrently (no barriers)
// mode 4: like 3 with a workgroup barrier per iteration
// modes 5-7: MFMA ordering / shape variants of mode 0
#include "../re-net_amd/csrc/gemm_split.hip"
#include <cstdio>
#include <vector>

namespace {

template <int mode>
__global__ __launch_bounds__(512) void probe_kernel(int iters, float* out, long long* cyc, const float4* src) {
    extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
    const int half = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    __bf16* sA = smem + half * (6 * PLANE);
    __bf16* sB = sA + 3 * PLANE;
    for (int i = threadIdx.x; i < 12 * PLANE / 2; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[4], rb[4];
    for (int i = 0; i < 4; ++i) { ra[i] = src[tid + 256 * i]; rb[i] = src[tid + 256 * i + 1024]; }
    const int arow = (wm * 64 + (lane & 31)) * LDS_ROW;
    const int brow = (wn * 64 + (lane & 31)) * LDS_ROW;
    const int ksel = (lane >> 5) * 8;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc16[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_mfma = mode == 0 || mode == 1 || (mode >= 5 && mode <= 7) || ((mode == 3 || mode == 4 || mode == 13 || mode == 14) && half == 0);
    const bool do_conv = mode == 2 || ((mode == 3 || mode == 4 || mode == 13 || mode == 14) && half == 1);
    if constexpr (mode == 13 || mode == 14) { if (half == 1) __builtin_amdgcn_s_setprio(3); }
    bf16x8 fa[2][3], fb[2][3];
    for (int t = 0; t < 2; ++t)
        for (int p = 0; p < 3; ++p) {
            fa[t][p] = *reinterpret_cast<const bf16x8*>(&sA[p * PLANE + arow + t * 32 * LDS_ROW + ksel]);
            fb[t][p] = *reinterpret_cast<const bf16x8*>(&sB[p * PLANE + brow + t * 32 * LDS_ROW + ksel]);
        }
    FusedCtx<false, true> fc;
    if constexpr (mode >= 10 && mode <= 12) {
        fc.M = 128; fc.N = 128; fc.K = 32; fc.m0 = 0; fc.n0 = 0; fc.tid = tid;
        fc.a_edge = false; fc.b_edge = false;
        fc.frag_a = arow + ksel; fc.frag_b = brow + ksel;
        fc.la.init(reinterpret_cast<const float*>(src), 32, 128, 32, 0, tid);
        fc.lb.init(reinterpret_cast<const float*>(src) + 4096, 32, 128, 32, 0, tid);
        fc.la.load(0, fc.ra);
        fc.lb.load(0, fc.rb);
        for (int i = 0; i < 12; ++i) { fc.F0[i] = fa[0][0]; fc.F1[i] = fb[0][0]; }
    }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (mode >= 10 && mode <= 12) {
            // the production k-step of gemm_split_fused_kernel on an L1/L2-resident 128x32 operand pair
            // 10: full step   11: without the split (MFMAs + fragment reads only)   12: no old-slab MFMAs
            fused_step<false, true, mode != 12, mode != 11, true>(fc, acc, smem + (it & 1) * BUF,
                                                                   smem + ((it & 1) ^ 1) * BUF, 0, 0);
            __syncthreads();
        }
        if constexpr (mode == 8 || mode == 9) {
            // one wave does BOTH: MFMAs on buffer (it & 1) interleaved with the split of the next tile into the
            // other buffer; the sched_group_barrier sequence spells the interleave out for the scheduler
            const __bf16* rA = smem + (it & 1) * (6 * PLANE);
            const __bf16* rB = rA + 3 * PLANE;
            __bf16* wA = smem + ((it & 1) ^ 1) * (6 * PLANE);
            __bf16* wB = wA + 3 * PLANE;
            mfma_tile(rA, rB, arow, brow, ksel, acc);
            store_items<true, false>(wA, 1 << 20, 1 << 20, 0, 0, tid, ra);
            store_items<true, false>(wB, 1 << 20, 1 << 20, 0, 0, tid, rb);
            for (int i = 0; i < 4; ++i) {
                ra[i].x += 1.f; ra[i].y += 1.f; ra[i].z += 1.f; ra[i].w += 1.f;
                rb[i].x += 1.f; rb[i].y += 1.f; rb[i].z += 1.f; rb[i].w += 1.f;
            }
            if constexpr (mode == 8) {
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
                for (int g = 0; g < 48; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    if (g % 2 == 0 && g < 36) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (g % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
            }
            __syncthreads();
        }
        if (do_mfma) {
            if constexpr (mode == 0) {
                constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
                constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                    for (int q = 0; q < 6; ++q)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA[q]], fb[j][PB[q]], acc[i][j], 0, 0, 0);
            } else if constexpr (mode == 5) {            // 12 consecutive MFMAs per accumulator
                constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
                constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                            for (int q = 0; q < 6; ++q)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA[q]], fb[j][PB[q]], acc[i][j], 0, 0, 0);
            } else if constexpr (mode == 6) {            // rotation over 2 accumulators
                constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
                constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                        for (int q = 0; q < 6; ++q)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA[q]], fb[j][PB[q]], acc[i][j], 0, 0, 0);
            } else if constexpr (mode == 7) {            // 16x16x32: 4x4 tiles of 16x16 cover the same 64x64 (8 per product)
#pragma unroll
                for (int rep = 0; rep < 12; ++rep)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i & 1][rep % 3], fb[j & 1][(rep + i) % 3], acc16[i][j], 0, 0, 0);
            } else {
                mfma_tile(sA, sB, arow, brow, ksel, acc);
            }
        }
        if (do_conv) {
            store_items<true, false>(sA, 1 << 20, 1 << 20, 0, 0, tid, ra);
            store_items<true, false>(sB, 1 << 20, 1 << 20, 0, 0, tid, rb);
            for (int i = 0; i < 4; ++i) {                                        // keep the work loop-variant
                ra[i].x += 1.f; ra[i].y += 1.f; ra[i].z += 1.f; ra[i].w += 1.f;
                rb[i].x += 1.f; rb[i].y += 1.f; rb[i].z += 1.f; rb[i].w += 1.f;
            }
        }
        if constexpr (mode == 4 || mode == 14) __syncthreads();
        else asm volatile("" ::: "memory");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    for (int i = 0; i < 4; ++i) s += ra[i].x + rb[i].y;
    if constexpr (mode == 7)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) s += acc16[i][j][0] + acc16[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int M>
void launch1(int blocks, int threads, int iters, float* out, long long* cyc, const float4* src) {
    hipFuncSetAttribute((const void*)probe_kernel<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FUSED_LDS);
    hipLaunchKernelGGL(probe_kernel<M>, dim3(blocks), dim3(threads), FUSED_LDS, 0, iters, out, cyc, src);
}

void launch(int mode, int blocks, int threads, int iters, float* out, long long* cyc, const float4* src) {
    switch (mode) {
        case 0: launch1<0>(blocks, threads, iters, out, cyc, src); break;
        case 1: launch1<1>(blocks, threads, iters, out, cyc, src); break;
        case 2: launch1<2>(blocks, threads, iters, out, cyc, src); break;
        case 3: launch1<3>(blocks, threads, iters, out, cyc, src); break;
        case 4: launch1<4>(blocks, threads, iters, out, cyc, src); break;
        case 5: launch1<5>(blocks, threads, iters, out, cyc, src); break;
        case 6: launch1<6>(blocks, threads, iters, out, cyc, src); break;
        case 7: launch1<7>(blocks, threads, iters, out, cyc, src); break;
        case 8: launch1<8>(blocks, threads, iters, out, cyc, src); break;
        case 9: launch1<9>(blocks, threads, iters, out, cyc, src); break;
        case 10: launch1<10>(blocks, threads, iters, out, cyc, src); break;
        case 11: launch1<11>(blocks, threads, iters, out, cyc, src); break;
        case 12: launch1<12>(blocks, threads, iters, out, cyc, src); break;
        case 13: launch1<13>(blocks, threads, iters, out, cyc, src); break;
        default: launch1<14>(blocks, threads, iters, out, cyc, src); break;
    }
}

}  // namespace

int main() {
    const int iters = 2000;
    float* out; long long* cyc; float4* src;
    hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&cyc, 1024 * 8 * 8); hipMalloc(&src, 2048 * 16);
    std::vector<float> h(2048 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.001f * (float)(i % 977) + 0.5f;
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    struct Cfg { int mode, threads, blocks; const char* name; };
    const Cfg cfgs[] = {
        {0, 256, 256, "48 MFMA regs, 1 wave/SIMD"}, {0, 512, 256, "48 MFMA regs, 2 waves/SIMD"},
        {1, 256, 256, "mfma_tile (LDS reads), 1 wave/SIMD"}, {1, 512, 256, "mfma_tile, 2 waves/SIMD"},
        {2, 256, 256, "convert+LDS stores, 1 wave/SIMD"}, {2, 512, 256, "convert, 2 waves/SIMD"},
        {5, 256, 256, "48 MFMA regs, 12-long chains, 1 wave/SIMD"}, {5, 512, 256, "48 MFMA 12-long chains, 2 waves/SIMD"},
        {6, 256, 256, "48 MFMA regs, rotation 2, 1 wave/SIMD"}, {7, 256, 256, "192 MFMA 16x16x32 (same flops), 1 wave/SIMD"},
        {7, 512, 256, "192 MFMA 16x16x32, 2 waves/SIMD"},
        {10, 256, 256, "fused_step full (L1-resident operands)"}, {11, 256, 256, "fused_step without split"},
        {12, 256, 256, "fused_step without old-slab MFMAs"}, {10, 256, 1, "fused_step full, ONE workgroup"},
        {9, 256, 256, "same wave: tile + convert, compiler order"}, {8, 256, 256, "same wave: tile + convert, interleaved"},
        {3, 512, 256, "mfma_tile || convert, free running"}, {13, 512, 256, "mfma_tile || convert(prio 3), free running"},
        {14, 512, 256, "mfma_tile || convert(prio 3), barrier/iter"}, {4, 512, 256, "mfma_tile || convert, barrier/iter"},
    };
    for (const Cfg& c : cfgs) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        launch(c.mode, c.blocks, c.threads, 10, out, cyc, src);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        launch(c.mode, c.blocks, c.threads, iters, out, cyc, src);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> hc(8);
        hipMemcpy(hc.data(), cyc + 8 * 17, 64, hipMemcpyDeviceToHost);
        const int nw = c.threads / 64;
        printf("%-44s wall %8.1f us/1000it  ticks/iter: w0 %7.1f  w%d %7.1f   (%.2f ticks/ns)\n", c.name,
               ms * 1e3 / iters * 1000, (double)hc[0] / iters, nw - 1, (double)hc[nw - 1] / iters,
               (double)hc[0] / (ms * 1e6));
    }
    return 0;
}
