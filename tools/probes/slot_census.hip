// Which hardware wave slots do the co-resident workgroups of a CU get?  (round 4: is HW_ID.wave_id & 1 a usable parity
// to tell the two workgroups of a CU apart?)   hipcc --offload-arch=gfx950 -O3 -o tools/_trace/slot_census tools/probes/slot_census.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 2) void census(unsigned* out, int spin) {
    __shared__ float big[18000];                       // 72 KB: two workgroups per CU, like the GEMM kernels
    big[threadIdx.x] = 1.f;
    __syncthreads();
    const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);       // HW_REG_HW_ID, all 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);      // HW_REG_XCC_ID
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    float a = big[threadIdx.x];
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;   // stay resident so that 512 workgroups co-reside
    if (a == 12345.f) out[0] = 0;
}
int main() {
    const int nb = 512;
    unsigned* d;
    hipMalloc(&d, nb * 4 * 2 * sizeof(unsigned));
    hipLaunchKernelGGL(census, dim3(nb), dim3(256), 0, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // per (xcc, se, sh?, cu): which wave_ids per simd
    std::map<unsigned, std::vector<int>> cu_blocks;
    std::map<int, int> slot_hist;
    for (int b = 0; b < nb; ++b)
        for (int w = 0; w < 4; ++w) {
            const unsigned hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 0xf;
            const int wave_id = hw & 0xf, simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            slot_hist[wave_id]++;
            if (b < 6 || (b >= 256 && b < 260))
                printf("block %3d wave %d: hw_id %08x xcc %u se %d sh %d cu %2d simd %d wave_id %d\n", b, w, hw, xcc, se, sh, cu, simd, wave_id);
            if (w == 0) cu_blocks[(xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(b * 16 + wave_id);
        }
    printf("wave_id histogram:");
    for (auto& kv : slot_hist) printf(" %d:%d", kv.first, kv.second);
    printf("\ndistinct CUs %zu\n", cu_blocks.size());
    int both = 0, same_parity = 0;
    for (auto& kv : cu_blocks) {
        if (kv.second.size() == 2) { both++; if (((kv.second[0] ^ kv.second[1]) & 1) == 0) same_parity++; }
    }
    printf("CUs with two workgroups %d, of which same wave_id parity (wave 0) %d\n", both, same_parity);
    int shown = 0;
    for (auto& kv : cu_blocks) {
        if (shown++ < 8) { printf("cu key %06x:", kv.first); for (int v : kv.second) printf(" block %d slot %d;", v / 16, v % 16); printf("\n"); }
    }
    return 0;
}
