// Skinny fp32-class GEMM for gfx950:  C[M, N] = alpha * A[M, K] @ op(B) + bias + beta * C  with a SMALL second
// operand (K <= 208, N <= 256): the self-loop products of the RGCN layers (RGCN.py:35 h @ W_loop and its
// backward g_loop @ W_loop^T: [N_nodes, 200] x [200, 200]) and every other "tall activation x small weight" shape.
//
// The general kernels (gemm_split.hip) tile the output 128 x 128 and re-stage BOTH operands through LDS per k-tile;
// on these shapes they run at 24-48 TFLOP/s: two column tiles of which the second is 56 % padding, a 7-step k loop
// that is all prologue / epilogue, and the fp32 -> 3 x bf16 split of the (tiny, constant) weight repeated by every
// workgroup and k-tile.  Here the roles are asymmetric:
//   * B is split ONCE per workgroup and stays RESIDENT IN REGISTERS as MFMA B-operand fragments: wave w of the 8
//     owns output columns [32 w, 32 w + 32) and holds, for all K, the three bf16 planes of its 32 columns
//     (K/16 k-steps x 3 planes x 4 VGPRs = 156 VGPRs at K = 208);
//   * A streams through LDS in 32-row tiles (double buffered, one barrier per tile): 512 threads load the tile
//     with float4 loads, split it on the way into LDS (three planes, [row][k], odd 16-byte row stride), and every
//     wave reads the SAME A fragments (one ds_read_b128 per plane and k-step) against its own B registers;
//   * bf16x6 arithmetic as in gemm_split.hip (x = x1 + x2 + x3, six term pairs on v_mfma_f32_32x32x16_bf16, fp32
//     accumulate, smallest terms first); even and odd k-steps accumulate into two independent chains.
// Workgroups are persistent over row tiles (tile = blockIdx.x, += gridDim.x).  Matrix work per 32-row tile and wave:
// 78 MFMAs (K = 208); a 23 033-row product is 720 tiles, < 3 per CU.
#include "common.h"
#include "gemm_skinny.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int SK_THREADS = 512;
constexpr int SK_ROWS = 32;                 // rows of A per tile (one MFMA row block)

struct SkinnyArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int M, N, K, lda, ldb, ldc;
    float alpha, beta;
    int n_tiles;
};

// x -> three bf16 terms (RNE each, residuals exact): lanes of two packed pairs
__device__ __forceinline__ void split2(f32x2 v, bf16x2 (&out)[3]) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        out[p] = __builtin_convertvector(v, bf16x2);
        if (p < 2) v -= __builtin_convertvector(out[p], f32x2);
    }
}

template <int K16, bool TB>
__global__ __launch_bounds__(SK_THREADS) void gemm_split_skinny_kernel(SkinnyArgs g) {
    constexpr int KP = K16 * 16;                 // padded K
    constexpr int SROW = KP + 8;                 // bf16 per LDS row: (KP + 8) * 2 B = odd multiple of 16 B
    constexpr int PLANE = SK_ROWS * SROW;        // bf16 per plane
    constexpr int BUF = 3 * PLANE;               // one A tile
    constexpr int ITEMS = SK_ROWS * (KP / 4);    // float4 items per tile
    constexpr int NI = (ITEMS + SK_THREADS - 1) / SK_THREADS;
    extern __shared__ __attribute__((aligned(16))) __bf16 smem_sk[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int n = wave * 32 + (lane & 31);       // this lane's output column
    const bool n_ok = n < g.N;

    // ---- B fragments: k-step s, lane (col n, k = 16 s + 8 half + j) -> three planes of 8 bf16 -----------------
    // Branch-free loads (clamped addresses, zeroed afterwards): with `if (ok) v = *p` hipcc branches around every
    // load and drains vmcnt at each merge -- 104 serialised L2 latencies at kernel start.  Issued in batches of
    // BATCH k-steps so that the loads of a batch are in flight together.
    bf16x8 bfr[K16][3];
    const int nc = min(n, g.N - 1);
    constexpr int BATCH = 4;
#pragma unroll
    for (int s0 = 0; s0 < K16; s0 += BATCH) {
        float v[BATCH][8];
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            const int s = s0 + b;
            if (s < K16) {
                const int k0 = s * 16 + half * 8;
                if constexpr (TB) {              // B[n][k]: 8 consecutive k of one row (K % 4 == 0)
                    const float* p = g.B + (size_t)nc * g.ldb;
                    const float4 a = *reinterpret_cast<const float4*>(p + min(k0, g.K - 4));
                    const float4 c = *reinterpret_cast<const float4*>(p + min(k0 + 4, g.K - 4));
                    v[b][0] = a.x; v[b][1] = a.y; v[b][2] = a.z; v[b][3] = a.w;
                    v[b][4] = c.x; v[b][5] = c.y; v[b][6] = c.z; v[b][7] = c.w;
                } else {                         // B[k][n]: a half-wave reads 32 consecutive n per k
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[b][j] = g.B[(size_t)min(k0 + j, g.K - 1) * g.ldb + nc];
                }
            }
        }
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            const int s = s0 + b;
            if (s < K16) {
                const int k0 = s * 16 + half * 8;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // a float4 is either entirely inside K or entirely past it (K % 4 == 0)
                    const bool ok = n_ok && (k0 + 2 * q) < g.K;
                    bf16x2 t[3];
                    split2(f32x2{ok ? v[b][2 * q] : 0.f, ok ? v[b][2 * q + 1] : 0.f}, t);
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        bfr[s][p][2 * q] = t[p][0];
                        bfr[s][p][2 * q + 1] = t[p][1];
                    }
                }
            }
        }
    }

    // ---- A tile staging: item f = tid + 512 i -> (row, 4 consecutive k) -----------------------------------------
    auto load_tile = [&](int tile, float4 (&r)[NI]) {
        const int row0 = tile * SK_ROWS;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int f = min(tid + SK_THREADS * i, ITEMS - 1);
            const int row = f / (KP / 4), k = (f - row * (KP / 4)) * 4;
            // clamped, branch-free (K % 4 == 0 and M >= 1: checked by the launcher); out-of-range items are zeroed
            // (zeroed in store_tile_lds, i.e. BEHIND the MFMAs: a select right here would make the compiler wait for
            // the load before the matrix work it is supposed to hide under)
            r[i] = *reinterpret_cast<const float4*>(g.A + (size_t)min(row0 + row, g.M - 1) * g.lda + min(k, g.K - 4));
        }
    };
    auto store_tile_lds = [&](__bf16* S, int tile, const float4 (&r)[NI]) {
        const int row0 = tile * SK_ROWS;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int f = tid + SK_THREADS * i;
            if (f < ITEMS) {
                const int row = f / (KP / 4), k = (f - row * (KP / 4)) * 4;
                const bool ok = row0 + row < g.M && k < g.K;
                bf16x2 lo[3], hi[3];
                split2(f32x2{ok ? r[i].x : 0.f, ok ? r[i].y : 0.f}, lo);
                split2(f32x2{ok ? r[i].z : 0.f, ok ? r[i].w : 0.f}, hi);
                __bf16* dst = S + row * SROW + k;
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    uint2 u;
                    u.x = __builtin_bit_cast(unsigned, lo[p]);
                    u.y = __builtin_bit_cast(unsigned, hi[p]);
                    *reinterpret_cast<uint2*>(dst + p * PLANE) = u;
                }
            }
        }
    };

    float4 stage[NI];
    int tile = blockIdx.x;
    int cur = 0;
    if (tile < g.n_tiles) {
        load_tile(tile, stage);
        store_tile_lds(smem_sk, tile, stage);
    }
    __syncthreads();
    const int aoff = (lane & 31) * SROW + half * 8;
    const float bv = (g.bias && n_ok) ? g.bias[nc] : 0.f;
    for (; tile < g.n_tiles; tile += gridDim.x) {
        const int nxt = tile + gridDim.x;
        const bool has_next = nxt < g.n_tiles;                  // workgroup-uniform
        // in flight under the MFMAs below.  UNCONDITIONAL (the last iteration re-loads its own tile and drops it): a
        // load inside `if (has_next)` makes hipcc drain vmcnt at the merge point, i.e. before the matrix work
        load_tile(has_next ? nxt : tile, stage);
        __builtin_amdgcn_sched_barrier(0);                      // ... and pinned HERE (the scheduler sinks them to their
                                                                // first use, behind the MFMAs, otherwise)
        const __bf16* S = smem_sk + cur * BUF;
        f32x16 acc[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
        constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int s = 0; s < K16; ++s) {
            bf16x8 a[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const bf16x8*>(S + p * PLANE + aoff + s * 16);
#pragma unroll
            for (int q = 0; q < 6; ++q)
                acc[s & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]], bfr[s][PB[q]], acc[s & 1], 0, 0, 0);
        }
        if (has_next) store_tile_lds(smem_sk + (cur ^ 1) * BUF, nxt, stage);
        // epilogue of this tile: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 half
        {
            const int row0 = tile * SK_ROWS;
            float* cp = g.C + (size_t)(row0 + 4 * half) * g.ldc + nc;      // + ((r & 3) + 8 (r >> 2)) * ldc
            if (row0 + SK_ROWS <= g.M) {                        // interior tile (wave-uniform): ONE exec region
                if (n_ok) {
#pragma unroll
                    for (int h8 = 0; h8 < 2; ++h8) {            // 8 reads of C in flight together, then 8 stores
                        float old[8];
#pragma unroll
                        for (int r8 = 0; r8 < 8; ++r8) old[r8] = 0.f;
                        if (g.beta != 0.f) {
#pragma unroll
                            for (int r8 = 0; r8 < 8; ++r8) {
                                const int r = h8 * 8 + r8;
                                old[r8] = cp[(size_t)((r & 3) + 8 * (r >> 2)) * g.ldc];
                            }
                        }
#pragma unroll
                        for (int r8 = 0; r8 < 8; ++r8) {
                            const int r = h8 * 8 + r8;
                            cp[(size_t)((r & 3) + 8 * (r >> 2)) * g.ldc] =
                                g.alpha * (acc[0][r] + acc[1][r]) + bv + g.beta * old[r8];
                        }
                    }
                }
            } else {                                            // the last, partial tile
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (n_ok && row < g.M) {
                        float* p = g.C + (size_t)row * g.ldc + n;
                        float v = g.alpha * (acc[0][r] + acc[1][r]) + bv;
                        if (g.beta != 0.f) v += g.beta * (*p);
                        *p = v;
                    }
                }
            }
        }
        // LDS-only barrier (next buffer complete, this one fully consumed): __syncthreads() would also wait for
        // the C stores above (s_waitcnt vmcnt(0)), one write latency per tile
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
    }
}

template <int K16, bool TB>
int launch_skinny(const SkinnyArgs& g, hipStream_t st) {
    constexpr size_t lds = (size_t)2 * 3 * SK_ROWS * (K16 * 16 + 8) * sizeof(__bf16);
    static bool attr_set = false;       // benign race: the attribute is idempotent
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_split_skinny_kernel<K16, TB>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int grid = min(g.n_tiles, 256);       // one persistent workgroup per CU
    RENET_LAUNCH((gemm_split_skinny_kernel<K16, TB>), dim3(grid), dim3(SK_THREADS), lds, st, g);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

}  // namespace

bool renet_gemm_skinny_eligible(int ta, int M, int N, int K, const float* A, int lda, const float* B, int ldb, int tb) {
    if (ta || N > 256 || N < 1 || K > 208 || K < 16 || (K & 3) || (lda & 3) || M < 256) return false;
    if (reinterpret_cast<uintptr_t>(A) & 15) return false;
    if (tb && ((ldb & 3) || (reinterpret_cast<uintptr_t>(B) & 15))) return false;
    return true;
}

int renet_gemm_skinny_launch(int tb, int M, int N, int K, float alpha, const float* A, int lda, const float* B, int ldb,
                             float beta, float* C, int ldc, const float* bias, void* stream) {
    SkinnyArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.alpha = alpha; g.beta = beta;
    g.n_tiles = (M + SK_ROWS - 1) / SK_ROWS;
    hipStream_t st = (hipStream_t)stream;
    if (K <= 112) return tb ? launch_skinny<7, true>(g, st) : launch_skinny<7, false>(g, st);
    return tb ? launch_skinny<13, true>(g, st) : launch_skinny<13, false>(g, st);
}
