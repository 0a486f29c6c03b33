// fp32-accurate GEMM on the bf16 matrix cores of gfx950 ("bf16x6 split"):
//
//   every fp32 operand value is split EXACTLY-to-rounding into three bf16 terms  x = x1 + x2 + x3
//   (x1 = rne(x), x2 = rne(x - x1), x3 = rne(x - x1 - x2); |x - x1 - x2 - x3| <= 2^-25 |x|), and the
//   product a*b is evaluated with the six term pairs of weight >= 2^-16:
//        a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)
//   each on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  bf16 x bf16 products are exact in fp32;
//   the dropped pairs (a2 b3, a3 b2, a3 b3) are <= 2^-23 |a b|, i.e. at the fp32 rounding level, so the
//   result is an fp32-class GEMM (measured against fp64 in tests/test_gpu_parity.py next to the
//   v_mfma_f32_32x32x2_f32 kernel of gemm.hip) -- while the matrix pipe runs 16x faster per product:
//   6 bf16 MFMAs replace 16/6 = 2.67x their time in f32-input MFMAs.
//
// 128x128x32 workgroup tile, 4 waves 2x2, each wave 2x2 MFMA tiles of 32x32.  Staging: every thread owns
// (row, 4 consecutive k) items: one global_load_dwordx4 when k is contiguous in memory, four dword loads
// (coalesced across lanes along the row index) otherwise; the split happens in registers on the way into
// LDS; three bf16 planes per operand, image [row][k] with an 80-byte row stride (16-byte aligned, rows
// spread over all banks); fragments are one ds_read_b128 per plane (8 consecutive k per lane).
// Out-of-range elements are clamped at load time and zeroed at LDS-store time (never right behind the
// load, see gemm.hip).
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDS_ROW = 40;                     // bf16 per LDS row (80 B)
constexpr int PLANE = BM * LDS_ROW;             // bf16 per plane
constexpr int THREADS = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct SplitArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int M, N, K, lda, ldb, ldc;
    float alpha, beta;
    int k_tiles_per_split;
    int split_k;
    float* partial;
};

// item i of this thread: CONTIG_K: row = f>>3, k = 4*(f&7);  else: row = f&127, k = 4*(f>>7)
template <bool CONTIG_K>
__device__ __forceinline__ void item_pos(int f, int& row, int& k) {
    if constexpr (CONTIG_K) { row = f >> 3; k = (f & 7) << 2; }
    else { row = f & 127; k = (f >> 7) << 2; }
}

// Per-thread load state: the row part of every item's address is computed ONCE (the per-tile work is an
// add); integer multiplies inside the k loop cost more issue slots than the MFMAs they feed.
template <bool CONTIG_K>
struct ItemLoader {
    const float* base[4];      // CONTIG_K: P + row*ld          else: P + row
    int kk[4];                 // k offset of the item inside a tile
    size_t ld;
    int K;

    __device__ __forceinline__ void init(const float* P, int ld_, int rows, int K_, int row0, int tid) {
        ld = (size_t)ld_;
        K = K_;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row, k;
            item_pos<CONTIG_K>(tid + THREADS * i, row, k);
            row = min(row0 + row, rows - 1);
            kk[i] = k;
            base[i] = CONTIG_K ? P + (size_t)row * ld : P + row;
        }
    }

    __device__ __forceinline__ void load(int k0, float4 (&r)[4]) const {
        const bool full = k0 + BK <= K;                     // workgroup-uniform
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + kk[i];
            if constexpr (CONTIG_K) {
                if (K >= 4) {
                    r[i] = *reinterpret_cast<const float4*>(base[i] + (full ? k : min(k, K - 4)));
                } else {
                    const float* p = base[i];
                    r[i] = make_float4(p[min(k, K - 1)], p[min(k + 1, K - 1)], p[min(k + 2, K - 1)], p[min(k + 3, K - 1)]);
                }
            } else {
                if (full) {
                    const float* p = base[i] + (size_t)k * ld;
                    r[i] = make_float4(p[0], p[ld], p[2 * ld], p[3 * ld]);
                } else {
                    const float* p = base[i];
                    r[i].x = p[(size_t)min(k, K - 1) * ld];
                    r[i].y = p[(size_t)min(k + 1, K - 1) * ld];
                    r[i].z = p[(size_t)min(k + 2, K - 1) * ld];
                    r[i].w = p[(size_t)min(k + 3, K - 1) * ld];
                }
            }
        }
    }
};

__device__ __forceinline__ uint2 pack4(bf16x2 lo, bf16x2 hi) {
    uint2 u;
    u.x = __builtin_bit_cast(unsigned, lo);
    u.y = __builtin_bit_cast(unsigned, hi);
    return u;
}

// registers -> three bf16 planes in LDS.  EDGE (workgroup-uniform: the tile touches the end of the matrix
// in either dimension) enables the out-of-range fix-up of the clamped loads; interior tiles -- almost all of
// them -- run the bare split: 6 v_cvt_pk_bf16_f32, 4 packed subtractions and 3 ds_write_b64 per item.
template <bool CONTIG_K, bool EDGE>
__device__ __forceinline__ void store_items(__bf16* __restrict__ S, int rows, int K, int row0, int k0, int tid,
                                            const float4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row, k;
        item_pos<CONTIG_K>(tid + THREADS * i, row, k);
        float4 v = r[i];
        if constexpr (EDGE) {
            const int kg = k0 + k;
            if constexpr (CONTIG_K) {
                if (K >= 4 && kg > K - 4 && kg < K) {      // the float4 was loaded from K-4: shift it back
                    const int d = kg - (K - 4);
                    v = d == 1 ? make_float4(v.y, v.z, v.w, 0.f) : d == 2 ? make_float4(v.z, v.w, 0.f, 0.f)
                                                                         : make_float4(v.w, 0.f, 0.f, 0.f);
                }
            }
            const bool rok = row0 + row < rows;
            if (!rok || kg >= K) v.x = 0.f;
            if (!rok || kg + 1 >= K) v.y = 0.f;
            if (!rok || kg + 2 >= K) v.z = 0.f;
            if (!rok || kg + 3 >= K) v.w = 0.f;
        }
        f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
        __bf16* dst = S + row * LDS_ROW + k;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const bf16x2 blo = __builtin_convertvector(lo, bf16x2);        // v_cvt_pk_bf16_f32 (RNE)
            const bf16x2 bhi = __builtin_convertvector(hi, bf16x2);
            *reinterpret_cast<uint2*>(dst + p * PLANE) = pack4(blo, bhi);
            if (p < 2) {
                lo -= __builtin_convertvector(blo, f32x2);                  // exact residuals
                hi -= __builtin_convertvector(bhi, f32x2);
            }
        }
    }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(THREADS) void gemm_split_kernel(SplitArgs g) {
    __shared__ __attribute__((aligned(16))) __bf16 sA[3 * PLANE];
    __shared__ __attribute__((aligned(16))) __bf16 sB[3 * PLANE];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int z = blockIdx.z;
    const int kt0 = z * g.k_tiles_per_split;
    const int kt_total = (g.K + BK - 1) / BK;
    const int kt1 = min(kt_total, kt0 + g.k_tiles_per_split);
    constexpr bool A_CK = !TA;
    constexpr bool B_CK = TB;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[4], rb[4];
    ItemLoader<A_CK> la;
    ItemLoader<B_CK> lb;
    la.init(g.A, g.lda, g.M, g.K, m0, tid);
    lb.init(g.B, g.ldb, g.N, g.K, n0, tid);
    if (kt0 < kt1) {
        la.load(kt0 * BK, ra);
        lb.load(kt0 * BK, rb);
    }
    const bool a_edge = m0 + BM > g.M, b_edge = n0 + BN > g.N;
    const int arow = (wm * 64 + (lane & 31)) * LDS_ROW;
    const int brow = (wn * 64 + (lane & 31)) * LDS_ROW;
    const int ksel = (lane >> 5) * 8;

    for (int kt = kt0; kt < kt1; ++kt) {
        __syncthreads();                               // previous tile fully consumed
        const bool k_edge = (kt + 1) * BK > g.K;
        if (a_edge || k_edge) store_items<A_CK, true>(sA, g.M, g.K, m0, kt * BK, tid, ra);
        else store_items<A_CK, false>(sA, g.M, g.K, m0, kt * BK, tid, ra);
        if (b_edge || k_edge) store_items<B_CK, true>(sB, g.N, g.K, n0, kt * BK, tid, rb);
        else store_items<B_CK, false>(sB, g.N, g.K, n0, kt * BK, tid, rb);
        __syncthreads();
        if (kt + 1 < kt1) {                            // next tile's global loads fly during the MFMAs
            la.load((kt + 1) * BK, ra);
            lb.load((kt + 1) * BK, rb);
        }
#pragma unroll
        for (int slab = 0; slab < 2; ++slab) {
            const int ko = slab * 16 + ksel;
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a[t][p] = *reinterpret_cast<const bf16x8*>(&sA[p * PLANE + arow + t * 32 * LDS_ROW + ko]);
                    b[t][p] = *reinterpret_cast<const bf16x8*>(&sB[p * PLANE + brow + t * 32 * LDS_ROW + ko]);
                }
            // six term pairs, smallest first; consecutive MFMAs go to DIFFERENT accumulators so that none
            // waits on the previous one's result
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
            constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[q]], b[j][PB[q]], acc[i][j], 0, 0, 0);
        }
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const bool split = g.split_k > 1;
    float* Cout = split ? g.partial + (size_t)z * g.M * g.N : g.C;
    const int ldo = split ? g.N : g.ldc;
    const int half = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
            if (col >= g.N) continue;
            const float bv = (!split && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < g.M) {
                    float* p = Cout + (size_t)row * ldo + col;
                    if (split) *p = acc[i][j][r];
                    else {
                        float v = g.alpha * acc[i][j][r] + bv;
                        if (g.beta != 0.f) v += g.beta * (*p);
                        *p = v;
                    }
                }
            }
        }
}

__global__ __launch_bounds__(256) void split_reduce_kernel(const float* __restrict__ partial, int split_k,
                                                           int M, int N, float alpha, float beta,
                                                           const float* __restrict__ bias,
                                                           float* __restrict__ C, int ldc) {
    const size_t total = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), n = (int)(i % N);
        float s = 0.f;
        for (int zz = 0; zz < split_k; ++zz) s += partial[(size_t)zz * total + i];
        float v = alpha * s + (bias ? bias[n] : 0.f);
        float* p = C + (size_t)m * ldc + n;
        if (beta != 0.f) v += beta * (*p);
        *p = v;
    }
}

}  // namespace

extern "C" {

int renet_gemm_f32_split(int ta, int tb, int M, int N, int K, float alpha, const float* A, int lda,
                         const float* B, int ldb, float beta, float* C, int ldc, const float* bias,
                         int split_k, float* workspace, size_t workspace_bytes, void* stream) {
    if (M < 0 || N < 0 || K < 1 || lda <= 0 || ldb <= 0 || ldc < N) return RENET_ERR_BADARG;
    if (M == 0 || N == 0) return RENET_OK;
    if (split_k < 1) split_k = 1;
    const int kt_total = (K + BK - 1) / BK;
    if (split_k > kt_total) split_k = max(kt_total, 1);
    if (split_k > 1 && workspace_bytes < renet_gemm_workspace(M, N, split_k)) return RENET_ERR_WORKSPACE;
    SplitArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.alpha = alpha; g.beta = beta;
    g.split_k = split_k;
    g.k_tiles_per_split = max(1, (kt_total + split_k - 1) / split_k);
    g.partial = workspace;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, split_k);
    if (!ta && !tb) hipLaunchKernelGGL((gemm_split_kernel<false, false>), grid, dim3(THREADS), 0, st, g);
    else if (!ta && tb) hipLaunchKernelGGL((gemm_split_kernel<false, true>), grid, dim3(THREADS), 0, st, g);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_split_kernel<true, false>), grid, dim3(THREADS), 0, st, g);
    else hipLaunchKernelGGL((gemm_split_kernel<true, true>), grid, dim3(THREADS), 0, st, g);
    RENET_LAUNCH_CHECK();
    if (split_k > 1) {
        const size_t total = (size_t)M * N;
        int blocks = (int)min((size_t)2048, (total + 255) / 256);
        hipLaunchKernelGGL(split_reduce_kernel, dim3(blocks), dim3(256), 0, st, workspace, split_k, M, N, alpha,
                           beta, bias, C, ldc);
        RENET_LAUNCH_CHECK();
    }
    return RENET_OK;
}

}  // extern "C"
