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
    bf16x8 bfr[K16][3];
#pragma unroll
    for (int s = 0; s < K16; ++s) {
        float v[8];
        const int k0 = s * 16 + half * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (n_ok) {
            if constexpr (TB) {                  // B[n][k]: 8 consecutive k of one row
                const float* p = g.B + (size_t)n * g.ldb + k0;
                if (k0 + 8 <= g.K) {
                    const float4 a = *reinterpret_cast<const float4*>(p);
                    const float4 b = *reinterpret_cast<const float4*>(p + 4);
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (k0 + j < g.K) v[j] = p[j];
                }
            } else {                             // B[k][n]: lanes of a half-wave read 32 consecutive n per k
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (k0 + j < g.K) v[j] = g.B[(size_t)(k0 + j) * g.ldb + n];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bf16x2 t[3];
            split2(f32x2{v[2 * q], v[2 * q + 1]}, t);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                bfr[s][p][2 * q] = t[p][0];
                bfr[s][p][2 * q + 1] = t[p][1];
            }
        }
    }

    // ---- A tile staging: item f = tid + 512 i -> (row, 4 consecutive k) -----------------------------------------
    auto load_tile = [&](int tile, float4 (&r)[NI]) {
        const int row0 = tile * SK_ROWS;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int f = tid + SK_THREADS * i;
            const int row = f / (KP / 4), k = (f - row * (KP / 4)) * 4;
            r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < ITEMS && row0 + row < g.M && k < g.K)       // K % 4 == 0 (checked by the launcher)
                r[i] = *reinterpret_cast<const float4*>(g.A + (size_t)(row0 + row) * g.lda + k);
        }
    };
    auto store_tile_lds = [&](__bf16* S, const float4 (&r)[NI]) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int f = tid + SK_THREADS * i;
            if (f < ITEMS) {
                const int row = f / (KP / 4), k = (f - row * (KP / 4)) * 4;
                bf16x2 lo[3], hi[3];
                split2(f32x2{r[i].x, r[i].y}, lo);
                split2(f32x2{r[i].z, r[i].w}, hi);
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
        store_tile_lds(smem_sk, stage);
    }
    __syncthreads();
    const int aoff = (lane & 31) * SROW + half * 8;
    for (; tile < g.n_tiles; tile += gridDim.x) {
        const int nxt = tile + gridDim.x;
        const bool has_next = nxt < g.n_tiles;                  // workgroup-uniform
        if (has_next) load_tile(nxt, stage);                    // in flight under the MFMAs below
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
        if (has_next) store_tile_lds(smem_sk + (cur ^ 1) * BUF, stage);
        // epilogue of this tile: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 half
        if (n_ok) {
            const float bv = g.bias ? g.bias[n] : 0.f;
            const int row0 = tile * SK_ROWS;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < g.M) {
                    float* p = g.C + (size_t)row * g.ldc + n;
                    float v = g.alpha * (acc[0][r] + acc[1][r]) + bv;
                    if (g.beta != 0.f) v += g.beta * (*p);
                    *p = v;
                }
            }
        }
        __syncthreads();                                        // next buffer complete, this one fully consumed
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
    if (ta || N > 256 || K > 208 || K < 16 || (K & 3) || (lda & 3) || M < 256) return false;
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
