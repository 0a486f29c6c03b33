// fp32 GEMM on the f32-input matrix cores of gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, i.e. a
// k-ordered fmaf chain, at the 157 TF/s vector-equivalent rate).  128x128x32 workgroup tile, 4 waves
// in a 2x2 arrangement, each wave 2x2 MFMA tiles of 32x32 (64 accumulator registers per lane).
//
// Operand staging is global -> registers -> LDS with 16-byte accesses end to end, double-buffered in
// LDS (one barrier per k-tile; the global loads of tile t+2 are in flight during the MFMAs of tile t):
//  * an operand whose k index is contiguous in memory (A of C=A.B, B of C=A.B^T) keeps that order in
//    LDS, image [row][k] with a 36-float row stride: ds_write_b128 in, ds_read_b128 out, both
//    conflict-free.  The MFMA consumes k in a permuted order to make that possible: within a group of
//    8 k the half-wave `ksel` owns k = 8g + 4*ksel + {0..3}, so one float4 feeds 4 consecutive MFMAs.
//    (Any k order is legal as long as A and B agree; the sum over k is the same set of products.)
//  * an operand whose row index is contiguous (A of C=A^T.B, B of C=A.B) is kept [k][row] with a
//    132-float stride: ds_write_b128 in, conflict-free ds_read_b32 out (32 consecutive rows per
//    half-wave), in the same permuted k order.
//
// Used for: the RGCN self-loop h @ W_loop (RGCN.py:35), the GRU input projections (inside nn.GRU,
// model.py:86,94), the score heads (model.py:89-90,98-99) and all their backward GEMMs
// (dX = dY W, dW = dY^T X with deterministic split-K).
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDK = BK + 4;             // [row][k] image: 36-float rows
constexpr int LDR = BM + 4;             // [k][row] image: 132-float rows
constexpr int OPBUF = BM * LDK;         // floats per operand buffer (>= BK * LDR)
constexpr int THREADS = 256;
static_assert(BK * LDR <= OPBUF, "operand buffer too small");

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int M, N, K, lda, ldb, ldc;
    float alpha, beta;
    int k_tiles_per_split;      // in units of BK
    int split_k;
    float* partial;             // [split_k, M, N] when split_k > 1
    int xcd_order;              // 1: XCD-aware virtual tile order (tile_of_block)
};

// Loads one 128 x BK operand tile into 4 float4 registers per thread.
//  CONTIG_K = true : storage is [rows, K] (k contiguous): float4 along k; thread f -> row f/8, kq f%8
//  CONTIG_K = false: storage is [K, rows] (row index contiguous): float4 along rows; f -> k f/32, rq f%32
template <bool CONTIG_K>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int rows, int K, int row0,
                                          int k0, int tid, bool vec_ok, float4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + THREADS * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (CONTIG_K) {
            const int row = row0 + (f >> 3), k = k0 + ((f & 7) << 2);
            if (row < rows) {
                const float* p = P + (size_t)row * ld + k;
                if (vec_ok && k + 3 < K) v = *reinterpret_cast<const float4*>(p);
                else {
                    if (k < K) v.x = p[0];
                    if (k + 1 < K) v.y = p[1];
                    if (k + 2 < K) v.z = p[2];
                    if (k + 3 < K) v.w = p[3];
                }
            }
        } else {
            const int k = k0 + (f >> 5), row = row0 + ((f & 31) << 2);
            if (k < K) {
                const float* p = P + (size_t)k * ld + row;
                if (vec_ok && row + 3 < rows) v = *reinterpret_cast<const float4*>(p);
                else {
                    if (row < rows) v.x = p[0];
                    if (row + 1 < rows) v.y = p[1];
                    if (row + 2 < rows) v.z = p[2];
                    if (row + 3 < rows) v.w = p[3];
                }
            }
        }
        r[i] = v;
    }
}

// Branch-free tile load: every thread always issues its 4 global_load_dwordx4.  Out-of-range rows / k
// are handled by CLAMPING the address into the matrix and fixing the value up afterwards (zero, or a
// component shift for the one float4 that straddles the end of the contiguous dimension), so the
// compiler can issue all loads back to back and overlap them with the MFMAs -- per-element predication
// (the generic loader above) costs 15-25 % on the shapes of this workload, whose N = 600 / 200 always
// has a partial tile.  gfx950 global loads only need dword alignment, so odd leading dimensions
// (dW_lin: lda = 23033) take this path too.  Needs >= 4 elements along the contiguous dimension.
__device__ __forceinline__ float4 shift_tail(float4 v, int d) {
    // element j of the result = v[j + d] if j + d < 4 else 0      (d = 1..3)
    if (d == 1) return make_float4(v.y, v.z, v.w, 0.f);
    if (d == 2) return make_float4(v.z, v.w, 0.f, 0.f);
    return make_float4(v.w, 0.f, 0.f, 0.f);
}

template <bool CONTIG_K>
__device__ __forceinline__ void load_tile_any(const float* __restrict__ P, int ld, int rows, int K, int row0,
                                              int k0, int tid, bool small, float4 (&r)[4]) {
    if (small) {
        load_tile<CONTIG_K>(P, ld, rows, K, row0, k0, tid, false, r);
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + THREADS * i;
        // c = index along the contiguous dimension (extent C), o = index along the other one (extent O)
        const int c = CONTIG_K ? k0 + ((f & 7) << 2) : row0 + ((f & 31) << 2);
        const int o = CONTIG_K ? row0 + (f >> 3) : k0 + (f >> 5);
        const int C = CONTIG_K ? K : rows;
        const int O = CONTIG_K ? rows : K;
        r[i] = *reinterpret_cast<const float4*>(P + (size_t)min(o, O - 1) * ld + min(c, C - 4));
    }
}

// registers -> LDS.  The out-of-range fix-up of the clamped loads happens HERE (one iteration after the
// load was issued), never right behind the load: a select on a just-loaded value would make the compiler
// wait for the load before the MFMA phase instead of after it.
template <bool CONTIG_K>
__device__ __forceinline__ void store_tile(float* __restrict__ S, int rows, int K, int row0, int k0, int tid,
                                           bool small, const float4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + THREADS * i;
        float4 v = r[i];
        if (!small) {
            const int c = CONTIG_K ? k0 + ((f & 7) << 2) : row0 + ((f & 31) << 2);
            const int o = CONTIG_K ? row0 + (f >> 3) : k0 + (f >> 5);
            const int C = CONTIG_K ? K : rows;
            const int O = CONTIG_K ? rows : K;
            if (o >= O || c >= C) v = make_float4(0.f, 0.f, 0.f, 0.f);
            else if (c > C - 4) v = shift_tail(v, c - (C - 4));
        }
        if constexpr (CONTIG_K) *reinterpret_cast<float4*>(&S[(f >> 3) * LDK + ((f & 7) << 2)]) = v;
        else *reinterpret_cast<float4*>(&S[(f >> 5) * LDR + ((f & 31) << 2)]) = v;
    }
}

// the 4 values (MFMA steps tt = 0..3) of k-group g for tile row `row`, half-wave ksel
template <bool CONTIG_K>
__device__ __forceinline__ float4 read_frag(const float* __restrict__ S, int row, int g, int ksel) {
    if constexpr (CONTIG_K) {
        return *reinterpret_cast<const float4*>(&S[row * LDK + 8 * g + 4 * ksel]);
    } else {
        const float* p = S + (8 * g + 4 * ksel) * LDR + row;
        return make_float4(p[0], p[LDR], p[2 * LDR], p[3 * LDR]);
    }
}

// The round-1 kernel: 64-bit addressing and per-element predication for ANY extent.  Since round 4 only the fallback for
// operands the buffer-addressed kernel below cannot take (>= 4 GiB, K < 4, fewer than 4 rows along a contiguous row index).
// TA: A stored [K,M] (A^T);  TB: B stored [N,K] (B^T)
template <bool TA, bool TB>
__global__ __launch_bounds__(THREADS) void gemm_f32_generic_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float smem[2][2][OPBUF];       // [buffer][A|B]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int z = blockIdx.z;
    const int kt0 = z * g.k_tiles_per_split;
    const int kt_total = (g.K + BK - 1) / BK;
    const int kt1 = min(kt_total, kt0 + g.k_tiles_per_split);

    // A is k-contiguous when NOT transposed; B ([K,N] row-major) is n-contiguous when NOT transposed.
    constexpr bool A_CK = !TA;
    constexpr bool B_CK = TB;
    const bool a_vec = ((g.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0);
    const bool b_vec = ((g.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    (void)a_vec; (void)b_vec;
    const bool a_full = (g.K < 4) || (g.M < 4);        // `small`: too short to clamp -> generic loader
    const bool b_full = (g.K < 4) || (g.N < 4);
    float4 ra[4], rb[4];
    if (kt0 < kt1) {
        load_tile_any<A_CK>(g.A, g.lda, g.M, g.K, m0, kt0 * BK, tid, a_full, ra);
        load_tile_any<B_CK>(g.B, g.ldb, g.N, g.K, n0, kt0 * BK, tid, b_full, rb);
        store_tile<A_CK>(smem[0][0], g.M, g.K, m0, kt0 * BK, tid, a_full, ra);
        store_tile<B_CK>(smem[0][1], g.N, g.K, n0, kt0 * BK, tid, b_full, rb);
        if (kt0 + 1 < kt1) {
            load_tile_any<A_CK>(g.A, g.lda, g.M, g.K, m0, (kt0 + 1) * BK, tid, a_full, ra);
            load_tile_any<B_CK>(g.B, g.ldb, g.N, g.K, n0, (kt0 + 1) * BK, tid, b_full, rb);
        }
    }
    __syncthreads();
    const int arow = wm * 64 + (lane & 31);
    const int brow = wn * 64 + (lane & 31);
    const int ksel = lane >> 5;

    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        if (kt + 1 < kt1) {                    // tile t+1: registers -> the other LDS buffer
            store_tile<A_CK>(smem[cur ^ 1][0], g.M, g.K, m0, (kt + 1) * BK, tid, a_full, ra);
            store_tile<B_CK>(smem[cur ^ 1][1], g.N, g.K, n0, (kt + 1) * BK, tid, b_full, rb);
        }
        if (kt + 2 < kt1) {                    // tile t+2: global -> registers, lands during the MFMAs below
            load_tile_any<A_CK>(g.A, g.lda, g.M, g.K, m0, (kt + 2) * BK, tid, a_full, ra);
            load_tile_any<B_CK>(g.B, g.ldb, g.N, g.K, n0, (kt + 2) * BK, tid, b_full, rb);
        }
        const float* As = smem[cur][0];
        const float* Bs = smem[cur][1];
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            const float4 a0 = read_frag<A_CK>(As, arow, kg, ksel);
            const float4 a1 = read_frag<A_CK>(As, arow + 32, kg, ksel);
            const float4 b0 = read_frag<B_CK>(Bs, brow, kg, ksel);
            const float4 b1 = read_frag<B_CK>(Bs, brow + 32, kg, ksel);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b1.x, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0.x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc[1][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b1.y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b0.y, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc[1][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b1.z, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b0.z, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc[1][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b1.w, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b0.w, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc[1][1], 0, 0, 0);
        }
        __syncthreads();                       // buffer `cur` free for tile t+2, buffer cur^1 complete
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const bool split = g.split_k > 1;
    float* Cout = split ? g.partial + (size_t)z * g.M * g.N : g.C;
    const int ldo = split ? g.N : g.ldc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
            if (col >= g.N) continue;
            const float bv = (!split && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * ksel;
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

// ---------------------------------------------------------------------------------------------------------------------
// The production exact-fp32 kernel (round 4).  Same tile, LDS images and permuted k order as above; what changed is
// everything AROUND the 64 MFMAs of a k-tile, which at 64 cycles per v_mfma_f32_32x32x2_f32 leave 16 issue slots each:
//   * operands are addressed through raw buffer descriptors (whole-tensor extent in SGPRs, one 32-bit byte offset per
//     item, advanced by a uniform k-tile step): no 64-bit address arithmetic, no clamping, NO BRANCH in the loop.
//     Dwords beyond the tensor read as 0 in hardware; what lies inside the tensor but outside the tile's logical
//     extent (the next row behind a K tail, the next k line behind a row tail) is zeroed by selects on the way into
//     LDS, one k-tile after the load was issued;
//   * the loop body is unconditional: tile t+1 is stored and tile t+2 is loaded even past the end of the split (zeros
//     or data nobody reads, into the buffer nobody reads);
//   * the order inside a k-tile is pinned (sched_barrier): fragments of k-group g+1 are read behind the first MFMAs of
//     group g, one {select, ds_write_b128, buffer_load} piece follows every seventh MFMA;
//   * the barrier at the end of a k-tile waits for LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier): the global loads
//     of tile t+2 stay in flight across it (__syncthreads() drains vmcnt).
// Two workgroups per CU (73.7 KB of LDS each): while one sits at its barrier or in its epilogue the other owns the
// matrix pipe.  Bound: the f32-input MFMA rate, 157.3 TFLOP/s (MI355X_MICROARCH.md); LDS / L2 traffic is an order
// of magnitude below its ceilings (16 ds_read_b128 + 8 ds_write_b128 + 8 buffer loads per wave and 4096 MFMA cycles).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void tile_of_block(int nbx, int nby, bool xcd_order, int& bx, int& by, int& bz) {
    // same virtual tile order as gemm_split.hip: XCD x owns one contiguous 1/8 of the tile sequence (of the (k-slice, tile)
    // sequence for split-K grids), in which the SHORT grid dimension runs fastest in panels of <= 8 tiles (tiles sharing a
    // slab of the long operand share an L2)
    bz = blockIdx.z;
    if (!xcd_order) { bx = blockIdx.x; by = blockIdx.y; return; }
    const int nb = nbx * nby;
    int t;
    if (gridDim.z == 1) {
        const int per = nb >> 3;
        const int L = blockIdx.x + nbx * blockIdx.y;
        t = L < 8 * per ? (L & 7) * per + (L >> 3) : L;
    } else {
        const int total = nb * (int)gridDim.z, per3 = total >> 3;
        const int L3 = blockIdx.x + nbx * (blockIdx.y + nby * blockIdx.z);
        const int v = L3 < 8 * per3 ? (L3 & 7) * per3 + (L3 >> 3) : L3;
        bz = v / nb;
        t = v - bz * nb;
    }
    const int ns = min(nbx, nby), nl = max(nbx, nby);
    const int w = min(ns, 8);
    const int p = t / (w * nl), r = t - p * (w * nl);
    const int wp = min(w, ns - p * w);
    const int l = r / wp, sh = p * w + (r - l * wp);
    if (nby <= nbx) { bx = l; by = sh; }
    else { by = l; bx = sh; }
}

// One operand tile (128 rows x KT k) of the buffer-addressed kernel.  KT = 32 or 16 floats of k per stage.
//   CONTIG_K (storage [rows, K]): item f -> row f / (KT/4), k 4 * (f % (KT/4)); LDS image [row][KT + 4]
//   else     (storage [K, rows]): item f -> k f / 32, rows 4 * (f % 32) ..;      LDS image [k][132]
template <bool CONTIG_K, int KT>
struct BufLoader {
    static constexpr int NI = KT / 8;              // float4 items per thread and stage
    static constexpr int CPR = KT / 4;             // float4 chunks per tile row (CONTIG_K)
    static constexpr int LDKT = KT + 4;
    __amdgpu_buffer_rsrc_t rs;
    uint32_t off[NI];          // byte offset of item i in k-tile 0
    uint32_t kstep;            // bytes per k-tile
    int lim[NI];               // CONTIG_K: k of the item inside a tile (K-tail fix-up); unused otherwise
    int rows, K;

    __device__ __forceinline__ void init(const float* P, int ld, int rows_, int K_, int row0, int tid) {
        rows = rows_; K = K_;
        const uint32_t extent = CONTIG_K ? (uint32_t)(rows - 1) * (uint32_t)ld + (uint32_t)K
                                         : (uint32_t)(K - 1) * (uint32_t)ld + (uint32_t)rows;
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), (short)0, (int)(extent * 4u), 0x00020000);
        kstep = CONTIG_K ? (uint32_t)KT * 4u : (uint32_t)KT * (uint32_t)ld * 4u;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int f = tid + THREADS * i;
            if constexpr (CONTIG_K) {
                const int row = row0 + f / CPR, k = (f % CPR) << 2;
                off[i] = ((uint32_t)min(row, rows - 1) * (uint32_t)ld + (uint32_t)k) * 4u;
                lim[i] = k;
            } else {
                const int k = f >> 5, row = row0 + ((f & 31) << 2);
                off[i] = ((uint32_t)k * (uint32_t)ld + (uint32_t)min(row, rows - 1)) * 4u;
                lim[i] = 0;
            }
        }
    }

    __device__ __forceinline__ float4 load(int i, int kt) const {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off[i] + (uint32_t)kt * kstep), 0, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }

    // K tail of an operand whose k index is contiguous in memory: the dwords behind K belong to the NEXT ROW (inside the
    // tensor: not zeroed by the range check) -> select zeros.  k0 = first k of the item's tile.  Nothing to do for the
    // other storage order (k >= K lies behind the tensor's extent: hardware zeros), and nothing for rows past the
    // operand in either order (they only feed outputs that are never stored).
    __device__ __forceinline__ float4 fix(int i, int k0, float4 v) const {
        if constexpr (CONTIG_K) {
            const int room = K - k0 - lim[i];           // components j < room are valid
            v.x = room > 0 ? v.x : 0.f;
            v.y = room > 1 ? v.y : 0.f;
            v.z = room > 2 ? v.z : 0.f;
            v.w = room > 3 ? v.w : 0.f;
        }
        return v;
    }

    __device__ __forceinline__ void store(float* __restrict__ S, int i, int tid, float4 v) const {
        const int f = tid + THREADS * i;
        if constexpr (CONTIG_K) *reinterpret_cast<float4*>(&S[(f / CPR) * LDKT + ((f % CPR) << 2)]) = v;
        else *reinterpret_cast<float4*>(&S[(f >> 5) * LDR + ((f & 31) << 2)]) = v;
    }

    // the 4 values (MFMA steps 0..3) of k-group g for tile row `row`, half-wave ksel
    static __device__ __forceinline__ float4 frag(const float* __restrict__ S, int row, int g, int ksel) {
        if constexpr (CONTIG_K) {
            return *reinterpret_cast<const float4*>(&S[row * LDKT + 8 * g + 4 * ksel]);
        } else {
            const float* p = S + (8 * g + 4 * ksel) * LDR + row;
            return make_float4(p[0], p[LDR], p[2 * LDR], p[3 * LDR]);
        }
    }
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ float f4_get(const float4& v, int t) { return t == 0 ? v.x : t == 1 ? v.y : t == 2 ? v.z : v.w; }

// accumulators -> C (or the split-K partial plane).  C/D layout of the 32x32 MFMA: col = lane & 31,
// row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  beta != 0: ALL reads of C before the stores (a store may alias
// the next load as far as the compiler knows: 64 chained round trips per lane otherwise).
__device__ __forceinline__ void store_acc(const GemmArgs& g, int m0, int n0, int z, int wm, int wn, int lane,
                                          const f32x16 (&acc)[2][2]) {
    const bool split = g.split_k > 1;
    float* Cout = split ? g.partial + (size_t)z * g.M * g.N : g.C;
    const int ldo = split ? g.N : g.ldc;
    const int half = lane >> 5;
    const bool accumulate = !split && g.beta != 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
            if (col >= g.N) continue;
            const float bv = (!split && g.bias) ? g.bias[col] : 0.f;
            const int row_base = m0 + wm * 64 + i * 32 + 4 * half;
            float old[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) old[r] = 0.f;
            if (accumulate) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(row_base + (r & 3) + 8 * (r >> 2), g.M - 1);
                    old[r] = Cout[(size_t)row * ldo + col];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row_base + (r & 3) + 8 * (r >> 2);
                if (row < g.M) {
                    float* p = Cout + (size_t)row * ldo + col;
                    if (split) *p = acc[i][j][r];
                    else *p = g.alpha * acc[i][j][r] + bv + (accumulate ? g.beta * old[r] : 0.f);
                }
            }
        }
}

#ifndef F32P_NOBAR
#define F32_LOOP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#else
#define F32_LOOP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif

// KT floats of k per stage; g.k_tiles_per_split counts KT-tiles.  WPE = workgroups per CU the launch is compiled for
// (KT 32: 73.7 KB of LDS, 2 per CU; KT 16: 41 KB, 3 per CU).
template <bool TA, bool TB, int KT, int WPE>
__global__ __launch_bounds__(THREADS, WPE) void gemm_f32_kernel(GemmArgs g) {
    constexpr bool A_CK = !TA;
    constexpr bool B_CK = TB;
    typedef BufLoader<A_CK, KT> LA;
    typedef BufLoader<B_CK, KT> LB;
    constexpr int NI = KT / 8, NG = KT / 8, NM = 16 * NG;              // items / k-groups / MFMAs per stage
    constexpr int OPB = (BM * (KT + 4) > KT * LDR) ? BM * (KT + 4) : KT * LDR;
    __shared__ __attribute__((aligned(16))) float smem[2][2][OPB];          // [buffer][A|B]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bx, by, z;
    tile_of_block(gridDim.x, gridDim.y, g.xcd_order != 0, bx, by, z);
    const int m0 = by * BM, n0 = bx * BN;
    const int kt0 = z * g.k_tiles_per_split;
    const int kt_total = (g.K + KT - 1) / KT;
    const int kt1 = min(kt_total, kt0 + g.k_tiles_per_split);

    // Known loss (profiles/r04_f32_gemm_probes.md): ~17 us per round of 512 tiles on the short-K shapes that is neither
    // k-loop nor tile quantisation (2048 x 23033 x 600: 510 us with an empty loop body around the MFMAs against 414 us
    // of sustained matrix-pipe time).  Tried and measured without gain: a static s_setprio by hardware wave slot, and a
    // one-time start stagger of half a tile for the slot-1 workgroup of every CU (a workgroup alone on its CU does not
    // run at twice the paired rate, so the waiting time is not given back).
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    LA la;
    LB lb;
    la.init(g.A, g.lda, g.M, g.K, m0, tid);
    lb.init(g.B, g.ldb, g.N, g.K, n0, tid);
    float4 ra[NI], rb[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) { ra[i] = la.load(i, kt0); rb[i] = lb.load(i, kt0); }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        la.store(smem[0][0], i, tid, la.fix(i, kt0 * KT, ra[i]));
        lb.store(smem[0][1], i, tid, lb.fix(i, kt0 * KT, rb[i]));
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) { ra[i] = la.load(i, kt0 + 1); rb[i] = lb.load(i, kt0 + 1); }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    const int arow = wm * 64 + (lane & 31);
    const int brow = wn * 64 + (lane & 31);
    const int ksel = lane >> 5;

    // One k-tile: 16 * NG MFMAs; tile kt+1 goes registers -> LDS and tile kt+2 global -> registers on the way.  FIX: the
    // stored tile may hold k >= K (only the last k-tile of the operand, or tiles past it): select zeros there.  Rows
    // past the operand need no fix-up at all: they only feed output rows / columns that are never stored.
    auto k_tile = [&](int kt, auto fixc) {
        constexpr bool FIX = decltype(fixc)::value;
        const int cur = (kt - kt0) & 1;
        const float* As = smem[cur][0];
        const float* Bs = smem[cur][1];
        float* An = smem[cur ^ 1][0];
        float* Bn = smem[cur ^ 1][1];
        float4 fa[2][2], fb[2][2];                       // [fragment buffer][MFMA tile]
        fa[0][0] = LA::frag(As, arow, 0, ksel);
        fb[0][0] = LB::frag(Bs, brow, 0, ksel);
        fa[0][1] = LA::frag(As, arow + 32, 0, ksel);
        fb[0][1] = LB::frag(Bs, brow + 32, 0, ksel);
        static_for<0, NM>([&](auto mc) {
            constexpr int m = mc.value, grp = m >> 4, t = (m >> 2) & 3, i = (m >> 1) & 1, j = m & 1, fbuf = grp & 1;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4_get(fa[fbuf][i], t), f4_get(fb[fbuf][j], t), acc[i][j],
                                                             0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (grp + 1 < NG && (m & 15) < 4) {         // fragments of k-group grp + 1, one per MFMA
                constexpr int q = m & 15;
                if constexpr (q == 0) fa[fbuf ^ 1][0] = LA::frag(As, arow, grp + 1, ksel);
                if constexpr (q == 1) fb[fbuf ^ 1][0] = LB::frag(Bs, brow, grp + 1, ksel);
                if constexpr (q == 2) fa[fbuf ^ 1][1] = LA::frag(As, arow + 32, grp + 1, ksel);
                if constexpr (q == 3) fb[fbuf ^ 1][1] = LB::frag(Bs, brow + 32, grp + 1, ksel);
            }
#ifndef F32P_NOSTAGE
            // staging: piece pc = {store item, reload item}; the store behind MFMA 4 + STRIDE * pc, its reload one MFMA on
            constexpr int STRIDE = (NM - 6) / (2 * NI);
            if constexpr (m >= 4 && (m - 4) % STRIDE == 0 && (m - 4) / STRIDE < 2 * NI) {
                constexpr int pc = (m - 4) / STRIDE, it = pc >> 1;
                if constexpr ((pc & 1) == 0) la.store(An, it, tid, FIX ? la.fix(it, (kt + 1) * KT, ra[it]) : ra[it]);
                else lb.store(Bn, it, tid, FIX ? lb.fix(it, (kt + 1) * KT, rb[it]) : rb[it]);
            }
            if constexpr (m >= 5 && (m - 5) % STRIDE == 0 && (m - 5) / STRIDE < 2 * NI) {
                constexpr int pc = (m - 5) / STRIDE, it = pc >> 1;
                if constexpr ((pc & 1) == 0) ra[it] = la.load(it, kt + 2);
                else rb[it] = lb.load(it, kt + 2);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
        F32_LOOP_BARRIER();                              // LDS only: tile t+2's loads stay in flight
    };
    // tiles kt+1 <= kt_total - 2 are complete along k: no fix-up in the steady state
    const int kt_plain = min(kt1, kt_total - 2);
    int kt = kt0;
    for (; kt < kt_plain; ++kt) k_tile(kt, std::false_type{});
    for (; kt < kt1; ++kt) k_tile(kt, std::true_type{});
    store_acc(g, m0, n0, z, wm, wn, lane, acc);
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int split_k,
                                                            int M, int N, float alpha, float beta,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ C, int ldc) {
    const size_t total = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), n = (int)(i % N);
        float s = 0.f;
        for (int z = 0; z < split_k; ++z) s += partial[(size_t)z * total + i];
        float v = alpha * s + (bias ? bias[n] : 0.f);
        float* p = C + (size_t)m * ldc + n;
        if (beta != 0.f) v += beta * (*p);
        *p = v;
    }
}

// out[g, n] = sum over the rows of group g of X[m, n].  Block = 64 columns x one row group; the 4
// waves take rows g0+w, g0+w+4, ... (4 independent loads in flight each), then a fixed-order LDS
// combine => deterministic.  Two launches (row groups, then the groups) keep every CU busy on the
// tall-skinny bias-gradient shapes ([7.6k, 600], [1024, 23033]).
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, int M, int N, int ldx,
                                                     int rows_per_group, float beta, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const int g = blockIdx.y;
    const int m0 = g * rows_per_group, m1 = min(M, m0 + rows_per_group);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (n < N) {
        int m = m0 + wave;
        for (; m + 12 < m1; m += 16) {
            s0 += (float)X[(size_t)m * ldx + n];
            s1 += (float)X[(size_t)(m + 4) * ldx + n];
            s2 += (float)X[(size_t)(m + 8) * ldx + n];
            s3 += (float)X[(size_t)(m + 12) * ldx + n];
        }
        for (; m < m1; m += 4) s0 += (float)X[(size_t)m * ldx + n];
    }
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && n < N) {
        float r = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        float* o = out + (size_t)g * N + n;
        if (beta != 0.f) r += beta * (*o);
        *o = r;
    }
}

// x *= *scale with the factor in device memory (an upstream autograd gradient); exactly 1 => nothing to do.
// bound_out (optional): |*scale| * bound_in -- the magnitude bound of the scaled tensor for the f16x3 GEMMs.
__global__ __launch_bounds__(256) void scale_dev_kernel(float* __restrict__ x, size_t n,
                                                        const float* __restrict__ scale, float bound_in,
                                                        float* __restrict__ bound_out) {
    const float f = *scale;
    if (bound_out && blockIdx.x == 0 && threadIdx.x == 0) *bound_out = fabsf(f) * bound_in;
    if (f == 1.f) return;
    const size_t n4 = n >> 2;
    float4* x4 = reinterpret_cast<float4*>(x);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = x4[i];
        x4[i] = make_float4(v.x * f, v.y * f, v.z * f, v.w * f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) x[(n4 << 2) + threadIdx.x] *= f;
}

// bf16 matrix (rows x ld elements, contiguous) *= *scale; exactly 1 => nothing to do
__global__ __launch_bounds__(256) void scale_dev_bf16_kernel(__bf16* __restrict__ x, size_t n,
                                                             const float* __restrict__ scale) {
    const float f = *scale;
    if (f == 1.f) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        x[i] = (__bf16)((float)x[i] * f);
}

inline int colsum_groups(int M) { return max(1, min(64, (M + 127) / 128)); }

template <typename T>
int colsum_impl(const T* X, int M, int N, int ldx, float* out, float beta, float* workspace,
                       size_t workspace_bytes, void* stream) {
    if (M < 0 || N <= 0 || ldx < N) return RENET_ERR_BADARG;
    const int G = colsum_groups(M);
    hipStream_t st = (hipStream_t)stream;
    if (G == 1) {
        RENET_LAUNCH((colsum_kernel<T>), dim3((N + 63) / 64, 1), dim3(256), 0, st, X, M, N, ldx, max(M, 1), beta, out);
        RENET_LAUNCH_CHECK();
        return RENET_OK;
    }
    if (workspace_bytes < (size_t)colsum_groups(M) * N * sizeof(float)) return RENET_ERR_WORKSPACE;
    const int rpg = (M + G - 1) / G;
    RENET_LAUNCH((colsum_kernel<T>), dim3((N + 63) / 64, G), dim3(256), 0, st, X, M, N, ldx, rpg, 0.f, workspace);
    RENET_LAUNCH_CHECK();
    RENET_LAUNCH((colsum_kernel<float>), dim3((N + 63) / 64, 1), dim3(256), 0, st, (const float*)workspace, G, N, N, G,
                 beta, out);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}


int tile_order_f32() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("RENET_GEMM_TILE_ORDER");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v;
}

// RENET_GEMM_F32_GENERIC=1: the round-1 kernel for every shape (A/B runs, tests of the fallback)
bool generic_forced() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("RENET_GEMM_F32_GENERIC");
        v = (e && e[0] == '1') ? 1 : 0;
    }
    return v != 0;
}

}  // namespace

extern "C" {

size_t renet_gemm_workspace(int M, int N, int split_k) {
    return split_k > 1 ? (size_t)split_k * (size_t)M * (size_t)N * sizeof(float) : 0;
}

int renet_gemm_f32(int ta, int tb, int M, int N, int K, float alpha, const float* A, int lda,
                   const float* B, int ldb, float beta, float* C, int ldc, const float* bias,
                   int split_k, float* workspace, size_t workspace_bytes, void* stream) {
    if (M < 0 || N < 0 || K < 0 || lda <= 0 || ldb <= 0 || ldc < N) return RENET_ERR_BADARG;
    if (M == 0 || N == 0) return RENET_OK;
    if (split_k < 1) split_k = 1;
    const int kt_total = (K + BK - 1) / BK;
    if (split_k > kt_total) split_k = max(kt_total, 1);
    if (split_k > 1 && workspace_bytes < renet_gemm_workspace(M, N, split_k)) return RENET_ERR_WORKSPACE;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.alpha = alpha; g.beta = beta;
    g.split_k = split_k;
    g.k_tiles_per_split = (kt_total + split_k - 1) / split_k;
    if (g.k_tiles_per_split < 1) g.k_tiles_per_split = 1;
    g.partial = workspace;
    g.xcd_order = tile_order_f32();
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, split_k);
    // the buffer-addressed kernel reaches every element (and up to two k-tiles past the end) with a 32-bit byte offset
    const size_t reach_a = ta ? ((size_t)K + 3 * BK) * lda + M : (size_t)M * lda + K + 3 * BK;
    const size_t reach_b = tb ? (size_t)N * ldb + K + 3 * BK : ((size_t)K + 3 * BK) * ldb + N;
    const bool buffered = K >= 1 && reach_a < ((size_t)1 << 30) && reach_b < ((size_t)1 << 30) && !generic_forced();
    if (buffered) {
        constexpr int kt = 32;      // a 16-deep stage (3 workgroups per CU) was measured: no gain on the step's shapes
        const int ktiles = (K + kt - 1) / kt;
        if (split_k > ktiles) { split_k = max(ktiles, 1); g.split_k = split_k; }
        g.k_tiles_per_split = max(1, (ktiles + split_k - 1) / split_k);
        const dim3 gridb((N + BN - 1) / BN, (M + BM - 1) / BM, split_k);
#define RENET_F32_LAUNCH(KT_, WPE_)                                                                                   \
        do {                                                                                                          \
            if (!ta && !tb) RENET_LAUNCH((gemm_f32_kernel<false, false, KT_, WPE_>), gridb, dim3(THREADS), 0, st, g); \
            else if (!ta && tb) RENET_LAUNCH((gemm_f32_kernel<false, true, KT_, WPE_>), gridb, dim3(THREADS), 0, st, g); \
            else if (ta && !tb) RENET_LAUNCH((gemm_f32_kernel<true, false, KT_, WPE_>), gridb, dim3(THREADS), 0, st, g); \
            else RENET_LAUNCH((gemm_f32_kernel<true, true, KT_, WPE_>), gridb, dim3(THREADS), 0, st, g);              \
        } while (0)
        RENET_F32_LAUNCH(32, 2);
#undef RENET_F32_LAUNCH
    } else {
        g.xcd_order = 0;
        if (!ta && !tb) RENET_LAUNCH((gemm_f32_generic_kernel<false, false>), grid, dim3(THREADS), 0, st, g);
        else if (!ta && tb) RENET_LAUNCH((gemm_f32_generic_kernel<false, true>), grid, dim3(THREADS), 0, st, g);
        else if (ta && !tb) RENET_LAUNCH((gemm_f32_generic_kernel<true, false>), grid, dim3(THREADS), 0, st, g);
        else RENET_LAUNCH((gemm_f32_generic_kernel<true, true>), grid, dim3(THREADS), 0, st, g);
    }
    RENET_LAUNCH_CHECK();
    if (split_k > 1) {
        const size_t total = (size_t)M * N;
        int blocks = (int)min((size_t)2048, (total + 255) / 256);
        RENET_LAUNCH(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, workspace, split_k, M, N,
                           alpha, beta, bias, C, ldc);
        RENET_LAUNCH_CHECK();
    }
    return RENET_OK;
}

size_t renet_colsum_workspace(int M, int N) {
    const int G = colsum_groups(M);
    return G > 1 ? (size_t)G * (size_t)N * sizeof(float) : 0;
}

int renet_scale_by_device_scalar(float* x, size_t n, const float* scale, void* stream) {
    if (!x || !scale || (reinterpret_cast<uintptr_t>(x) & 15)) return RENET_ERR_BADARG;
    if (n == 0) return RENET_OK;
    const int blocks = (int)min((size_t)2048, (n / 4 + 255) / 256 + 1);
    RENET_LAUNCH(scale_dev_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, scale, 0.f, (float*)nullptr);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

int renet_scale_by_device_scalar_bound(float* x, size_t n, const float* scale, float bound_in, float* bound_out,
                                       void* stream) {
    if (!x || !scale || !bound_out || (reinterpret_cast<uintptr_t>(x) & 15)) return RENET_ERR_BADARG;
    const int blocks = (int)min((size_t)2048, (n / 4 + 255) / 256 + 1);
    RENET_LAUNCH(scale_dev_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, scale, bound_in, bound_out);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

int renet_colsum(const float* X, int M, int N, int ldx, float* out, float beta, float* workspace,
                 size_t workspace_bytes, void* stream) {
    return colsum_impl<float>(X, M, N, ldx, out, beta, workspace, workspace_bytes, stream);
}

int renet_colsum_bf16(const void* X, int M, int N, int ldx, float* out, float beta, float* workspace,
                      size_t workspace_bytes, void* stream) {
    return colsum_impl<__bf16>((const __bf16*)X, M, N, ldx, out, beta, workspace, workspace_bytes, stream);
}

int renet_scale_bf16_by_device_scalar(void* x, size_t n, const float* scale, void* stream) {
    if (n == 0) return RENET_OK;
    const int blocks = (int)min((size_t)2048, (n + 255) / 256);
    RENET_LAUNCH(scale_dev_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (__bf16*)x, n, scale);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

}  // extern "C"
