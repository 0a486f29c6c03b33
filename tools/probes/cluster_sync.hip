// Cost of a per-step rendezvous among the workgroups of a small CLUSTER (the weight-stationary GRU of DESIGN 8b.3: the gate
// columns of W_hh split over G workgroups that keep their slice in LDS for the whole launch and exchange h through L2 once per
// time step).  grid = clusters x G persistent workgroups (one per CU); every step each workgroup does `work` iterations of a
// dependent FMA chain (stand-in for its MFMA + gate work), writes `bytes` of state to a global buffer, publishes its step
// counter with a release store and waits until the G counters of its cluster have reached the step (acquire loads).
// Prints shader cycles per step with and without the rendezvous, for G in {1, 2, 5} -- the difference is what the
// exchange costs per step (to be compared with the ~13 us per step of the current persistent kernel).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/cluster_sync.hip -o tools/_trace/cluster_sync && tools/_trace/cluster_sync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// same_xcd: the members of a cluster are workgroups with EQUAL index mod 8 (the dispatcher deals workgroup b to XCD b % 8), so
// flags and state stay within one L2; otherwise consecutive workgroups = G different XCDs (coherence through the fabric).
__global__ __launch_bounds__(256) void k(int G, int steps, int work, int words, unsigned* flags, float* state,
                                         long long* cyc, float* sink, int sync_on, int same_xcd) {
    const int wg = blockIdx.x;
    int cluster, member;
    if (same_xcd) {
        const int x = wg & 7, local = wg >> 3;                 // position within the XCD
        cluster = x * 64 + local / G;
        member = local - (local / G) * G;
    } else {
        cluster = wg / G;
        member = wg - cluster * G;
    }
    unsigned* f = flags + cluster * 64;                       // one 256-byte line group per cluster
    float* st = state + (size_t)cluster * G * words;
    float acc = threadIdx.x * 1e-3f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 1; s <= steps; ++s) {
        for (int i = 0; i < work; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 1e-7f);          // dependent chain: ~4 cycles each
        for (int w = threadIdx.x; w < words; w += 256) st[member * words + w] = acc + w;      // this member's share of h
        if (sync_on) {
            __threadfence();                                                                   // state before flag
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(&f[member], (unsigned)s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x < G) {
                while (__hip_atomic_load(&f[threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)s)
                    __builtin_amdgcn_s_sleep(1);
            }
            __syncthreads();
            // read the partners' shares (what the next step's matrix product consumes)
            float r = 0.f;
            for (int w = threadIdx.x; w < G * words; w += 256) r += __builtin_nontemporal_load(&st[w]);
            acc += r * 1e-20f;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && wg == 0) cyc[0] = t1 - t0;
    sink[wg * 256 + threadIdx.x] = acc;
}

int main() {
    const int steps = 2000, work = 500;
    unsigned* flags; float *state, *sink; long long* cyc;
    hipMalloc(&flags, 512 * 64 * sizeof(unsigned));
    hipMalloc(&state, (size_t)512 * 8 * 4096 * sizeof(float));
    hipMalloc(&sink, 256 * 256 * sizeof(float));
    hipMallocManaged(&cyc, 64);
    for (int same_xcd = 0; same_xcd <= 1; ++same_xcd)
        for (int G : {1, 2, 5}) {
            // same_xcd: 32 workgroups per XCD -> 6 clusters of 5 (30 workgroups) per XCD: the grid is 8 * (32 / G) * G
            const int wgs = same_xcd ? 8 * ((32 / G) * G) : (255 / G) * G;
            const int words = 80 * 40;                            // 80 sequences x 40 hidden units per member and step
            for (int sync_on = 0; sync_on <= 1; ++sync_on) {
                hipMemset(flags, 0, 512 * 64 * sizeof(unsigned));
                hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, G, steps, work, words, flags, state, cyc, sink, sync_on,
                                   same_xcd);
                if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
                printf("%s, G = %d, %3d workgroups, rendezvous %s: %8.0f cycles per step\n",
                       same_xcd ? "cluster within one XCD" : "cluster across XCDs   ", G, wgs, sync_on ? "on " : "off",
                       (double)cyc[0] / steps);
            }
        }
    return 0;
}
