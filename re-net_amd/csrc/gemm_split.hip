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
// Two k-loop structures share the tile (128x128x32, 4 waves 2x2, each wave 2x2 MFMA tiles of 32x32) and the
// LDS image: gemm_split_kernel (two phases per k-tile, two workgroups per CU) and gemm_split_fused_kernel
// (split interleaved with the MFMAs, one workgroup per CU); kernel_choice() picks by grid size.  Staging: every thread owns
// (row, 4 consecutive k) items: one global_load_dwordx4 when k is contiguous in memory, four dword loads
// (coalesced across lanes along the row index) otherwise; the split happens in registers on the way into
// LDS; three bf16 planes per operand, image [row][k] with an 80-byte row stride (16-byte aligned, rows
// spread over all banks); fragments are one ds_read_b128 per plane (8 consecutive k per lane).
// Out-of-range elements are clamped at load time and zeroed at LDS-store time (never right behind the
// load, see gemm.hip).
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "gemm_skinny.h"

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
    int xcd_order;          // 0: plain order; w >= 1: XCD-aware virtual tile order with panels of <= w tiles (tile_of_block)
};


// Optional phase tracing (tools/gemm_trace.py builds a separate library with -DRENET_GEMM_TRACE; the shipped
// library contains none of this): s_memtime stamps per wave and k-step for the first TRACE_BLOCKS workgroups.
#ifdef RENET_GEMM_TRACE
constexpr int TRACE_BLOCKS = 64, TRACE_STEPS = 320;
__device__ unsigned long long* g_trace = nullptr;          // [TRACE_BLOCKS][8 waves][TRACE_STEPS][4]
__device__ __forceinline__ void trace_put(int wave8, int step, int slot, unsigned long long v) {
    const int flat = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    // workgroups 0-31 and 256-287: the second set usually lands on the same CUs as the first
    if (g_trace && flat < 512 && (flat & 255) < 32 && step < TRACE_STEPS && (threadIdx.x & 63) == 0)
        g_trace[(((size_t)((flat >> 8) * 32 + (flat & 255)) * 8 + wave8) * TRACE_STEPS + step) * 4 + slot] = v;
}
#define TRACE_T(wave8, step, slot) trace_put(wave8, step, slot, __builtin_amdgcn_s_memtime())
#define TRACE_V(wave8, step, slot, v) trace_put(wave8, step, slot, (unsigned long long)(v))
#else
#define TRACE_T(wave8, step, slot)
#define TRACE_V(wave8, step, slot, v)
#endif

// Tile of this workgroup.  The dispatcher deals workgroups to the 8 XCDs round-robin (block b -> XCD b % 8, each
// XCD with its own 4 MB L2), so the plain (blockIdx.x, blockIdx.y) order makes every XCD sweep the WHOLE of the
// long operand once per tile row of the short one (PMC: 215 MB of fabric traffic per GEMM launch of the step
// against ~60 MB of operands + output; 127 MB with this order).  Virtual order instead: XCD x owns one contiguous
// 1/8 of the tile sequence, and in that sequence the SHORT grid dimension runs fastest, so the tiles that share
// a slab of the long operand are consecutive on one XCD and the slab is fetched once.  RENET_GEMM_TILE_ORDER=0
// in the environment restores the plain order (tools/gemm_bench.py).
__device__ __forceinline__ void tile_of_block(int nbx, int nby, int xcd_order, int& bx, int& by, int& bz) {
    bz = blockIdx.z;
    if (!xcd_order) { bx = blockIdx.x; by = blockIdx.y; return; }
    const int nb = nbx * nby;
    int t;                                                   // position in the tile sequence of one k-slice
    if (gridDim.z == 1) {
        const int per = nb >> 3;
        const int L = blockIdx.x + nbx * blockIdx.y;
        t = L < 8 * per ? (L & 7) * per + (L >> 3) : L;
    } else {
        // split-K grids (round 4): the dispatcher deals the FLATTENED index (x fastest, then y, then z) to the XCDs, so the
        // 2-D rule above spreads every k-slice over all eight L2s whenever nbx * nby is not a multiple of 8 -- and even
        // when it is, each slice's operand slabs are fetched by all XCDs (PMC, tools/pmc_by_shape.py: 600 x 800 x 16000 / 14
        // read 421 MB for 90 MB of operands, dfeat 2048 x 600 x 23033 / 6 689 MB for 244).  Here XCD x owns one contiguous
        // eighth of the (k-slice, tile) sequence: whole k-slices, read by one L2 (two where a slice straddles).
        const int total = nb * (int)gridDim.z, per3 = total >> 3;
        const int L3 = blockIdx.x + nbx * (blockIdx.y + nby * blockIdx.z);
        const int v = L3 < 8 * per3 ? (L3 & 7) * per3 + (L3 >> 3) : L3;
        bz = v / nb;
        t = v - bz * nb;
    }
    // sequence: panels of <= 8 tiles across the SHORT dimension, the long dimension sweeping each panel
    // (a square problem becomes 8 x 8 blocks of concurrently resident tiles per XCD instead of 2 x 32)
    const int ns = min(nbx, nby), nl = max(nbx, nby);
    const int w = min(ns, xcd_order);
    const int p = t / (w * nl), r = t - p * (w * nl);
    const int wp = min(w, ns - p * w);                       // width of this (possibly last, narrower) panel
    const int l = r / wp, sh = p * w + (r - l * wp);
    if (nby <= nbx) { bx = l; by = sh; }
    else { by = l; bx = sh; }
}

// item i of this thread (f = tid + threads * i) of a ROWS x 32 operand tile:
//   CONTIG_K: row = f>>3, k = 4*(f&7);  else: row = f % ROWS, k = 4*(f / ROWS)
template <bool CONTIG_K, int ROWS = 128>
__device__ __forceinline__ void item_pos(int f, int& row, int& k) {
    if constexpr (CONTIG_K) { row = f >> 3; k = (f & 7) << 2; }
    else { row = f & (ROWS - 1); k = (f / ROWS) << 2; }
}

// Per-thread load state: the row part of every item's address is computed ONCE (the per-tile work is an
// add); integer multiplies inside the k loop cost more issue slots than the MFMAs they feed.
template <bool CONTIG_K, int NT = 256, int ROWS = 128, int NI = 4>
struct ItemLoader {
    const float* base[NI];     // CONTIG_K: P + row*ld          else: P + row
    int kk[NI];                // k offset of the item inside a tile
    size_t ld;
    int K;

    __device__ __forceinline__ void init(const float* P, int ld_, int rows, int K_, int row0, int tid) {
        ld = (size_t)ld_;
        K = K_;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int row, k;
            item_pos<CONTIG_K, ROWS>(tid + NT * i, row, k);
            row = min(row0 + row, rows - 1);
            kk[i] = k;
            base[i] = CONTIG_K ? P + (size_t)row * ld : P + row;
        }
    }

    __device__ __forceinline__ void load(int k0, float4 (&r)[NI]) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) load_item(i, k0, r[i]);
    }

    __device__ __forceinline__ void load_item(int i, int k0, float4& r) const {
        const bool full = k0 + BK <= K;                     // workgroup-uniform
        const int k = k0 + kk[i];
        if constexpr (CONTIG_K) {
            if (K >= 4) {
                r = *reinterpret_cast<const float4*>(base[i] + (full ? k : min(k, K - 4)));
            } else {
                const float* p = base[i];
                r = make_float4(p[min(k, K - 1)], p[min(k + 1, K - 1)], p[min(k + 2, K - 1)], p[min(k + 3, K - 1)]);
            }
        } else {
            if (full) {
                const float* p = base[i] + (size_t)k * ld;
                r = make_float4(p[0], p[ld], p[2 * ld], p[3 * ld]);
            } else {
                const float* p = base[i];
                r.x = p[(size_t)min(k, K - 1) * ld];
                r.y = p[(size_t)min(k + 1, K - 1) * ld];
                r.z = p[(size_t)min(k + 2, K - 1) * ld];
                r.w = p[(size_t)min(k + 3, K - 1) * ld];
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

// Out-of-range fix-up of one clamped item (see ItemLoader::load_item): (row, k) are the item's coordinates
// inside the tile.
template <bool CONTIG_K>
__device__ __forceinline__ float4 fix_item(float4 v, int rows, int K, int row0, int k0, int row, int k) {
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
    return v;
}

// Raw-buffer loader (f16x3 kernels since round 3, bf16x6 two-phase kernels since round 4): raw buffer loads -- a descriptor of the operand (SGPRs) plus a per-lane 32-bit byte
// offset computed once and advanced by the tile's uniform k offset (one v_add per load): no 64-bit address arithmetic and
// NO BRANCH inside the MFMA phase.  (tools/gemm_trace.py: with the generic ItemLoader -- clamped addresses, a
// uniform branch per item -- every load piece cost the issuing wave ~180 cycles between two MFMAs, 2.4x the
// matrix-pipe time of the 24-MFMA phase.)  Nothing is clamped along k: the descriptor's num_records is the operand's
// exact extent, every dword beyond it reads as 0 without touching memory (raw buffers are range-checked per dword:
// tests/test_gpu_parity.py runs K % 4 != 0 with odd row strides, where the last row's last 16-byte load straddles the
// end), reads beyond K inside it (the next row) are zeroed by store_items_h's EDGE path like the clamped rows.
// Requires rows * ld * 4 < 2^32 (checked on the host; larger operands run the bf16x6 kernels).
typedef uint32_t h3_u32x4 __attribute__((ext_vector_type(4)));

template <bool CONTIG_K, int NT, int ROWS, int NI>
struct TileLoaderH {
    __amdgpu_buffer_rsrc_t rs;
    uint32_t off[NI];          // bytes: CONTIG_K: (row * ld + kk) * 4      else: (row + kk * ld) * 4
    uint32_t ldb;              // row stride in bytes

    __device__ __forceinline__ void init(const float* P, int ld, int rows, int K, int row0, int tid) {
        const uint32_t extent = CONTIG_K ? (uint32_t)(rows - 1) * (uint32_t)ld + (uint32_t)K
                                         : (uint32_t)(K - 1) * (uint32_t)ld + (uint32_t)rows;
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), (short)0, (int)(extent * 4u),
                                               0x00020000);
        ldb = (uint32_t)ld * 4u;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int row, k;
            item_pos<CONTIG_K, ROWS>(tid + NT * i, row, k);
            row = min(row0 + row, rows - 1);
            off[i] = (CONTIG_K ? (uint32_t)row * (uint32_t)ld + (uint32_t)k : (uint32_t)row + (uint32_t)k * (uint32_t)ld) * 4u;
        }
    }

    __device__ __forceinline__ void load_item(int i, int k0, float4& r) const {
#ifdef RENET_PROBE_NOLOAD           // probe builds only (tools/gemm_split_probe.py): the k-loop without its global loads
        if (k0 > 0) return;
#endif
        if constexpr (CONTIG_K) {
            const h3_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off[i] + (uint32_t)k0 * 4u), 0, 0);
            r = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        } else {
            const uint32_t b0 = off[i] + (uint32_t)k0 * ldb;
            r.x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)b0, 0, 0));
            r.y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(b0 + ldb), 0, 0));
            r.z = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(b0 + 2u * ldb), 0, 0));
            r.w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(b0 + 3u * ldb), 0, 0));
        }
    }

    __device__ __forceinline__ void load(int k0, float4 (&r)[NI]) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) load_item(i, k0, r[i]);
    }
};

// out-of-range fix-up of one UNCLAMPED item: rows past the operand and k past K become zeros
__device__ __forceinline__ float4 fix_item_h(float4 v, int rows, int K, int row0, int k0, int row, int k) {
    const int kg = k0 + k;
    const bool rok = row0 + row < rows;
    if (!rok || kg >= K) v.x = 0.f;
    if (!rok || kg + 1 >= K) v.y = 0.f;
    if (!rok || kg + 2 >= K) v.z = 0.f;
    if (!rok || kg + 3 >= K) v.w = 0.f;
    return v;
}

// registers -> three bf16 planes in LDS.  EDGE (workgroup-uniform: the tile touches the end of the matrix
// in either dimension) enables the out-of-range fix-up of the clamped loads; interior tiles -- almost all of
// them -- run the bare split: 6 v_cvt_pk_bf16_f32, 4 packed subtractions and 3 ds_write_b64 per item.
template <bool CONTIG_K, bool EDGE, int NT = 256, int ROWS = 128, int NI = 4, bool RAW = false>
__device__ __forceinline__ void store_items(__bf16* __restrict__ S, int rows, int K, int row0, int k0, int tid,
                                            const float4 (&r)[NI]) {
    constexpr int PLANE = ROWS * LDS_ROW;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int row, k;
        item_pos<CONTIG_K, ROWS>(tid + NT * i, row, k);
        float4 v = r[i];
        if constexpr (EDGE) {
            if constexpr (RAW) v = fix_item_h(v, rows, K, row0, k0, row, k);      // unclamped loads (TileLoaderH)
            else v = fix_item<CONTIG_K>(v, rows, K, row0, k0, row, k);
        }
        f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
        __bf16* dst = S + row * LDS_ROW + k;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const bf16x2 blo = __builtin_convertvector(lo, bf16x2);        // v_cvt_pk_bf16_f32 (RNE)
            const bf16x2 bhi = __builtin_convertvector(hi, bf16x2);
#ifdef RENET_PROBE_NOLDSW           // probe builds only: the split without its LDS stores (one plane still written)
            if (p == 2) *reinterpret_cast<uint2*>(dst) = pack4(blo, bhi);
#else
            *reinterpret_cast<uint2*>(dst + p * PLANE) = pack4(blo, bhi);
#endif
#ifdef RENET_PROBE_NOSPLIT          // probe builds only: three roundings, no residual arithmetic
            if (false)
#endif
            if (p < 2) {
                lo -= __builtin_convertvector(blo, f32x2);                  // exact residuals
                hi -= __builtin_convertvector(bhi, f32x2);
            }
        }
    }
}


// One 128x128x32 tile step of a wave: 24 fragment reads (ds_read_b128) and 48 MFMAs.
template <int PLANE_A = PLANE, int PLANE_B = PLANE>
__device__ __forceinline__ void mfma_tile(const __bf16* __restrict__ sA, const __bf16* __restrict__ sB, int arow,
                                          int brow, int ksel, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int slab = 0; slab < 2; ++slab) {
        const int ko = slab * 16 + ksel;
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[t][p] = *reinterpret_cast<const bf16x8*>(&sA[p * PLANE_A + arow + t * 32 * LDS_ROW + ko]);
                b[t][p] = *reinterpret_cast<const bf16x8*>(&sB[p * PLANE_B + brow + t * 32 * LDS_ROW + ko]);
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

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// The MFMA phase of the two-phase kernels with the NEXT tile's global loads spread over it: issued in one
// burst right after the barrier (8 x 1 KB per wave, every wave of the CU at once) they fill the vector-memory
// queue -- the texture-address unit takes 64 B/clk, 16 cycles per dwordx4 wave-instruction -- and the waves
// sit in the load issue for ~1000-1500 cycles before their first MFMA (tools/gemm_trace.py).  Order pinned with
// sched_barrier: slab-0 fragments, then 48 MFMAs with the slab-1 fragment reads behind the first 12 and one
// load piece behind every fifth.
template <int PLANE_A, int PLANE_B, int NPIECES, class LoadFn>
__device__ __forceinline__ void mfma_tile_ld(const __bf16* __restrict__ sA, const __bf16* __restrict__ sB, int arow,
                                             int brow, int ksel, f32x16 (&acc)[2][2], LoadFn&& load_piece) {
    bf16x8 F0[12], F1[12];                       // index = operand + 2 * t + 4 * plane
    auto read = [&](auto frc, bf16x8 (&F)[12], int slab) {
        constexpr int fr = frc.value, op = fr & 1, t = (fr >> 1) & 1, p = fr >> 2;
        const __bf16* base = op ? sB + p * PLANE_B + brow : sA + p * PLANE_A + arow;
        F[fr] = *reinterpret_cast<const bf16x8*>(base + t * 32 * LDS_ROW + slab * 16 + ksel);
    };
    // in the order the term pairs consume them (lgkmcnt retires LDS reads in order: the first MFMA waits for
    // two fragments, not twelve)
    constexpr int ORDER[12] = {8, 1, 3, 10, 4, 5, 7, 6, 0, 9, 11, 2};
    static_for<0, 12>([&](auto n) { read(std::integral_constant<int, ORDER[n.value]>{}, F0, 0); });
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
    static_for<0, 48>([&](auto gc) {
        constexpr int g = gc.value, w = g % 24, q = w >> 2, i = (w >> 1) & 1, j = w & 1;
        if constexpr (g < 24)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F0[2 * i + 4 * PA[q]], F0[1 + 2 * j + 4 * PB[q]], acc[i][j], 0, 0, 0);
        else
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F1[2 * i + 4 * PA[q]], F1[1 + 2 * j + 4 * PB[q]], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g < 12) read(std::integral_constant<int, ORDER[g]>{}, F1, 1);
        if constexpr (g % 5 == 2 && g / 5 < NPIECES) load_piece(std::integral_constant<int, g / 5>{});
        __builtin_amdgcn_sched_barrier(0);
    });
}

// accumulators -> C (or the split-K partial plane).  C/D layout of the 32x32 MFMA: col = lane & 31,
// row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ void store_tile(const SplitArgs& g, int m0, int n0, int z, int wm, int wn, int lane,
                                           const f32x16 (&acc)[2][2]) {
#ifdef RENET_PROBE_NOSTORE          // probe builds only (tools/gemm_split_probe.py): what the C-store epilogue costs
    if (acc[0][0][0] != 12345.678f) return;
#endif
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
            // beta != 0 (in-place gradient accumulation): ALL 16 reads of C first, then the 16 stores.  Written as
            // load / fma / store per element the compiler must assume that a store aliases the next load and chains
            // 64 memory round trips per lane at the end of every tile.
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

// RAW: operands addressed through raw buffer descriptors (TileLoaderH; both operands below 2^30 elements of reach, checked
// on the host) -- the generic ItemLoader otherwise.
// (Round 6 tried a 64-byte-row image with XOR-swizzled chunks here -- no LDS bank conflicts, 48 KB per tile: no change at two
// workgroups per CU, 15-50 % SLOWER at three (168 registers: spills); profiles/r06_a_planes_ablation.md.  Not kept.)
template <bool TA, bool TB, bool RAW>
__global__ __launch_bounds__(THREADS) void gemm_split_kernel(SplitArgs g) {
    constexpr int PL = BM * LDS_ROW;
    __shared__ __attribute__((aligned(16))) __bf16 sA[3 * PL];
    __shared__ __attribute__((aligned(16))) __bf16 sB[3 * PL];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bx, by, z;
    tile_of_block(gridDim.x, gridDim.y, g.xcd_order, bx, by, z);
    const int m0 = by * BM, n0 = bx * BN;
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
    std::conditional_t<RAW, TileLoaderH<A_CK, THREADS, BM, 4>, ItemLoader<A_CK>> la;
    std::conditional_t<RAW, TileLoaderH<B_CK, THREADS, BN, 4>, ItemLoader<B_CK>> lb;
    la.init(g.A, g.lda, g.M, g.K, m0, tid);
    lb.init(g.B, g.ldb, g.N, g.K, n0, tid);
    if (kt0 < kt1) {
        la.load(kt0 * BK, ra);
        lb.load(kt0 * BK, rb);
    }
    const bool a_edge = m0 + BM > g.M, b_edge = n0 + BN > g.N;
    const int ra_ = wm * 64 + (lane & 31), rb_ = wn * 64 + (lane & 31);
    const int arow = ra_ * LDS_ROW, brow = rb_ * LDS_ROW;
    const int ksel = (lane >> 5) * 8;

    TRACE_V(wave, TRACE_STEPS - 1, 0, __builtin_amdgcn_s_getreg((31 << 11) | 4));      // HW_ID
    TRACE_V(wave, TRACE_STEPS - 1, 1, __builtin_amdgcn_s_getreg((31 << 11) | 20));     // XCC_ID
    for (int kt = kt0; kt < kt1; ++kt) {
        __syncthreads();                               // previous tile fully consumed
        TRACE_T(wave, kt - kt0, 0);
        const bool k_edge = (kt + 1) * BK > g.K;
        if (a_edge || k_edge) store_items<A_CK, true, THREADS, BM, 4, RAW>(sA, g.M, g.K, m0, kt * BK, tid, ra);
        else store_items<A_CK, false, THREADS, BM, 4, RAW>(sA, g.M, g.K, m0, kt * BK, tid, ra);
        if (b_edge || k_edge) store_items<B_CK, true, THREADS, BN, 4, RAW>(sB, g.N, g.K, n0, kt * BK, tid, rb);
        else store_items<B_CK, false, THREADS, BN, 4, RAW>(sB, g.N, g.K, n0, kt * BK, tid, rb);
        TRACE_T(wave, kt - kt0, 1);
        __syncthreads();
        TRACE_T(wave, kt - kt0, 2);
        const int k0n = min(kt + 1, kt1 - 1) * BK;     // next tile (the last step reloads its own: harmless)
        mfma_tile_ld<PL, PL, 8>(sA, sB, arow, brow, ksel, acc, [&](auto ic) {
            constexpr int i = ic.value;
            if constexpr (i < 4) la.load_item(i, k0n, ra[i]);
            else lb.load_item(i - 4, k0n, rb[i - 4]);
        });
        TRACE_T(wave, kt - kt0, 3);
    }

    store_tile(g, m0, n0, z, wm, wn, lane, acc);
}

// ------------------------------------------------------------------------------------------------------
// TALL variant of the two-phase kernel: 256 x 128 tile, 8 waves (4 x 2, each 64 x 64 as above), one workgroup per
// CU.  The in-loop split is paid per operand ELEMENT: a 256 x 128 x 32 step splits 12 288 elements for 2 x the MACs of
// a 128 x 128 x 32 step (8 192 elements), i.e. 0.75x the VALU / LDS-store work per flop -- the resource the k-loop
// is bound by (header of the fused variant).  Used for outputs with >= 1000 such tiles (use_tall).
// ------------------------------------------------------------------------------------------------------
constexpr int BMT = 256, THREADS_T = 512;
constexpr int PLANE_T = BMT * LDS_ROW;
constexpr size_t TALL_LDS = (size_t)(3 * PLANE_T + 3 * PLANE) * sizeof(__bf16);

template <bool TA, bool TB, bool RAW>
__global__ __launch_bounds__(THREADS_T) void gemm_split_tall_kernel(SplitArgs g) {
    extern __shared__ __attribute__((aligned(16))) __bf16 smem_t[];
    __bf16* sA = smem_t;
    __bf16* sB = smem_t + 3 * PLANE_T;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bx, by, z;
    tile_of_block(gridDim.x, gridDim.y, g.xcd_order, bx, by, z);
    const int m0 = by * BMT, n0 = bx * BN;
    const int kt0 = z * g.k_tiles_per_split;
    const int kt_total = (g.K + BK - 1) / BK;
    const int kt1 = min(kt_total, kt0 + g.k_tiles_per_split);
    constexpr bool A_CK = !TA;
    constexpr bool B_CK = TB;
    constexpr int NIA = BMT * 8 / THREADS_T, NIB = BN * 8 / THREADS_T;         // 4 and 2 items per thread

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[NIA], rb[NIB];
    std::conditional_t<RAW, TileLoaderH<A_CK, THREADS_T, BMT, NIA>, ItemLoader<A_CK, THREADS_T, BMT, NIA>> la;
    std::conditional_t<RAW, TileLoaderH<B_CK, THREADS_T, BN, NIB>, ItemLoader<B_CK, THREADS_T, BN, NIB>> lb;
    la.init(g.A, g.lda, g.M, g.K, m0, tid);
    lb.init(g.B, g.ldb, g.N, g.K, n0, tid);
    if (kt0 < kt1) {
        la.load(kt0 * BK, ra);
        lb.load(kt0 * BK, rb);
    }
    const bool a_edge = m0 + BMT > g.M, b_edge = n0 + BN > g.N;
    const int arow = (wm * 64 + (lane & 31)) * LDS_ROW;
    const int brow = (wn * 64 + (lane & 31)) * LDS_ROW;
    const int ksel = (lane >> 5) * 8;
    for (int kt = kt0; kt < kt1; ++kt) {
        __syncthreads();                               // previous tile fully consumed
        const bool k_edge = (kt + 1) * BK > g.K;
        if (a_edge || k_edge) store_items<A_CK, true, THREADS_T, BMT, NIA, RAW>(sA, g.M, g.K, m0, kt * BK, tid, ra);
        else store_items<A_CK, false, THREADS_T, BMT, NIA, RAW>(sA, g.M, g.K, m0, kt * BK, tid, ra);
        if (b_edge || k_edge) store_items<B_CK, true, THREADS_T, BN, NIB, RAW>(sB, g.N, g.K, n0, kt * BK, tid, rb);
        else store_items<B_CK, false, THREADS_T, BN, NIB, RAW>(sB, g.N, g.K, n0, kt * BK, tid, rb);
        __syncthreads();
        const int k0n = min(kt + 1, kt1 - 1) * BK;     // next tile (the last step reloads its own: harmless)
        mfma_tile_ld<PLANE_T, PLANE, NIA + NIB>(sA, sB, arow, brow, ksel, acc, [&](auto ic) {
            constexpr int i = ic.value;
            if constexpr (i < NIA) la.load_item(i, k0n, ra[i]);
            else lb.load_item(i - NIA, k0n, rb[i - NIA]);
        });
    }
    store_tile(g, m0, n0, z, wm, wn, lane, acc);
}

// ------------------------------------------------------------------------------------------------------
// FUSED variant: every wave runs its MFMAs and the split of the NEXT k-tile in ONE instruction stream.
//
// Measured on MI355X (tools/mfma_probe.hip, tools/fill_probe.hip; shader cycles per 128x128x32 k-tile):
//   48 v_mfma_f32_32x32x16_bf16 of one wave                       1536   (32.0 each, any accumulator order)
//   ... plus its 24 ds_read_b128                                  1697
//   split of a k-tile (8 items/thread) + LDS stores, alone        1114
//   the same split as a SECOND wave beside an MFMA wave           2657   (s_setprio changes nothing)
//   free in a 32-cycle MFMA shadow of the same wave: ~4 plain VALU, 2 ds_read_b128; v_cvt_pk_bf16_f32 counts
//   double, v_pk_add_f32 and ds_write_b64 do not overlap at all (the LDS store path moves ~85 B/clk/CU: the
//   48 KB of planes of one k-tile cost ~650 cycles whichever wave issues them)
// so neither two independent workgroups per CU (gemm_split_kernel: both drift into lockstep, 3300 cycles per
// k-tile and CU) nor barrier-anti-phased halves (tried: 3650) hide the split behind the matrix pipe.  Here each
// of the 48 MFMAs of a k-step is followed by one "micro-step" of the split (4 independent plain VALU, at times
// one ds_write_b64 / ds_read_b128 / global load), the order pinned with sched_barrier: 2540 cycles per k-tile
// (1623 without the split; the LDS stores are ~650 of the difference).
//
// LDS is double buffered (2 x 61 440 B, one 4-wave workgroup per CU) with ONE barrier per k-tile, and the
// k-loop is rotated by half a tile so that no MFMA waits for LDS after the barrier:
//     barrier(kt): tile kt visible in buf[kt&1]; every wave holds slab 1 of tile kt-1 in registers (F1)
//       reads  F0 <- slab 0 of tile kt                   (first 4 up front, 8 in the first MFMA shadows)
//       MFMAs   0..23 : slab 1 of tile kt-1 (F1)         | split micro-steps 0..23 of tile kt+1 -> buf[~kt&1]
//       MFMAs  24..47 : slab 0 of tile kt   (F0)         | micro-steps 24..47, reads F1 <- slab 1 of tile kt
//     (buf[~kt&1] held tile kt-1, whose last reads -- F1 -- completed before barrier(kt) in every wave)
// The global loads of tile kt+2 are issued from the micro-steps that free their registers.
// Loader of the fused kernel: every item's address is  UNIFORM tile base (SGPRs, advanced per k-tile by the
// scalar unit) + a per-lane 32-bit element offset computed once (global_load saddr form: no VALU address math
// in the k-loop).  Requires rows * ld < 2^31 elements and, for CONTIG_K, K >= 4 (checked on the host).
template <bool CONTIG_K>
struct TileLoader {
    const float* P;
    int off[4];                // CONTIG_K: row * ld + kk       else: row + kk * ld
    int rowc[4], kk[4];        // clamped row / k offset inside a tile (for the partial last k-tile)
    int ld, K;

    __device__ __forceinline__ void init(const float* P_, int ld_, int rows, int K_, int row0, int tid) {
        P = P_; ld = ld_; K = K_;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row, k;
            item_pos<CONTIG_K>(tid + THREADS * i, row, k);
            row = min(row0 + row, rows - 1);
            rowc[i] = row;
            kk[i] = k;
            off[i] = CONTIG_K ? row * ld + k : row + k * ld;
        }
    }

    template <bool FAST>
    __device__ __forceinline__ void load_item(int i, int k0, float4& r) const {
        if (FAST || k0 + BK <= K) {                         // workgroup-uniform: all but the last partial tile
            if constexpr (CONTIG_K) {
                r = *reinterpret_cast<const float4*>(P + k0 + off[i]);
            } else {
                const float* t0 = P + (size_t)k0 * ld;
                r = make_float4(t0[off[i]], (t0 + ld)[off[i]], (t0 + 2 * (size_t)ld)[off[i]],
                                (t0 + 3 * (size_t)ld)[off[i]]);
            }
        } else {
            const int k = k0 + kk[i];
            if constexpr (CONTIG_K) {
                r = *reinterpret_cast<const float4*>(P + (size_t)rowc[i] * ld + min(k, K - 4));
            } else {
                const float* p = P + rowc[i];
                r.x = p[(size_t)min(k, K - 1) * ld];
                r.y = p[(size_t)min(k + 1, K - 1) * ld];
                r.z = p[(size_t)min(k + 2, K - 1) * ld];
                r.w = p[(size_t)min(k + 3, K - 1) * ld];
            }
        }
    }

    __device__ __forceinline__ void load(int k0, float4 (&r)[4]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) load_item<false>(i, k0, r[i]);
    }
};

constexpr int BUF = 6 * PLANE;                              // bf16 per buffer: A planes, then B planes
constexpr size_t FUSED_LDS = (size_t)2 * BUF * sizeof(__bf16);

struct SplitState {
    float x[4];          // the item, then its residuals
    bf16x2 blo, bhi;     // the bf16 terms just split off (pairs x[0..1], x[2..3])
};

__device__ __forceinline__ float bf16_lo_as_f32(bf16x2 b) {
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, b) << 16);
}
__device__ __forceinline__ float bf16_hi_as_f32(bf16x2 b) {
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, b) & 0xffff0000u);
}
__device__ __forceinline__ bf16x2 rne_pair(float lo, float hi) {          // v_cvt_pk_bf16_f32
    return __builtin_convertvector(f32x2{lo, hi}, bf16x2);
}

// Micro-step ST (0..5) of one item: the SAME three round-to-nearest terms as store_items (both kernels feed
// the matrix cores identical planes), cut into pieces that issue beside an MFMA (tools/fill_probe.hip: ~4 plain
// VALU per 32-cycle shadow; v_cvt_pk_bf16_f32 counts double; v_pk_add_f32 does not overlap at all, hence the
// scalar subtractions); dependent instructions sit in different steps.
template <int ST>
__device__ __forceinline__ void split_step(SplitState& t, __bf16* dst) {
    if constexpr (ST == 0) {
        t.blo = rne_pair(t.x[0], t.x[1]);
        t.bhi = rne_pair(t.x[2], t.x[3]);
        *reinterpret_cast<uint2*>(dst) = pack4(t.blo, t.bhi);
    } else if constexpr (ST == 1) {
        t.x[0] -= bf16_lo_as_f32(t.blo);
        t.x[1] -= bf16_hi_as_f32(t.blo);
    } else if constexpr (ST == 2) {
        t.x[2] -= bf16_lo_as_f32(t.bhi);
        t.x[3] -= bf16_hi_as_f32(t.bhi);
    } else if constexpr (ST == 3) {
        t.blo = rne_pair(t.x[0], t.x[1]);
        t.bhi = rne_pair(t.x[2], t.x[3]);
        *reinterpret_cast<uint2*>(dst + PLANE) = pack4(t.blo, t.bhi);
    } else if constexpr (ST == 4) {
        t.x[0] -= bf16_lo_as_f32(t.blo);
        t.x[1] -= bf16_hi_as_f32(t.blo);
        t.x[2] -= bf16_lo_as_f32(t.bhi);
        t.x[3] -= bf16_hi_as_f32(t.bhi);
    } else {
        *reinterpret_cast<uint2*>(dst + 2 * PLANE) = pack4(rne_pair(t.x[0], t.x[1]), rne_pair(t.x[2], t.x[3]));
    }
}

template <bool TA, bool TB>
struct FusedCtx {
    static constexpr bool A_CK = !TA, B_CK = TB;
    TileLoader<A_CK> la;
    TileLoader<B_CK> lb;
    float4 ra[4], rb[4];                 // fp32 items of the tile being split next
    bf16x8 F0[12], F1[12];               // fragments: index = operand + 2 * t + 4 * plane
    SplitState st;
    int M, N, K, m0, n0, tid;
    int frag_a, frag_b;                  // element offsets of this lane's fragment rows inside a plane
    bool a_edge, b_edge;

    // item `it` (0..7): operand it&1 (0 = A), slot it>>1
    // FAST: the tile being split needs no out-of-range fix-up and the tile being loaded is a full one
    template <int IT, bool FAST>
    __device__ __forceinline__ void item_begin(int k0_tile, int k0_next, __bf16* wbuf, __bf16*& dst) {
        constexpr int i = IT >> 1;
        int row, k;
        float4 v;
        if constexpr ((IT & 1) == 0) {
            item_pos<A_CK>(tid + THREADS * i, row, k);
            v = ra[i];
            if (!FAST && (a_edge || k0_tile + BK > K)) v = fix_item<A_CK>(v, M, K, m0, k0_tile, row, k);
            dst = wbuf + row * LDS_ROW + k;
        } else {
            item_pos<B_CK>(tid + THREADS * i, row, k);
            v = rb[i];
            if (!FAST && (b_edge || k0_tile + BK > K)) v = fix_item<B_CK>(v, N, K, n0, k0_tile, row, k);
            dst = wbuf + 3 * PLANE + row * LDS_ROW + k;
        }
        st.x[0] = v.x; st.x[1] = v.y; st.x[2] = v.z; st.x[3] = v.w;
        if constexpr ((IT & 1) == 0) la.template load_item<FAST>(i, k0_next, ra[i]);      // the register is free again
        else lb.template load_item<FAST>(i, k0_next, rb[i]);
    }

    template <int FR>
    __device__ __forceinline__ void read_frag(bf16x8 (&F)[12], const __bf16* rbuf, int slab) {
        constexpr int op = FR & 1, t = (FR >> 1) & 1, p = FR >> 2;
        const __bf16* base = rbuf + (op ? 3 * PLANE + frag_b : frag_a);
        F[FR] = *reinterpret_cast<const bf16x8*>(base + p * PLANE + t * 32 * LDS_ROW + slab * 16);
    }
};

template <int W>
__device__ __forceinline__ void mfma_w(const bf16x8 (&F)[12], f32x16 (&acc)[2][2]) {
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
    constexpr int q = W >> 2, i = (W >> 1) & 1, j = W & 1;
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[2 * i + 4 * PA[q]], F[1 + 2 * j + 4 * PB[q]], acc[i][j], 0, 0, 0);
}

// one rotated k-step (see the header comment).  WITH_OLD: slab 1 of the previous tile is pending in F1;
// WITH_CONV: there is a next tile to split.
template <bool TA, bool TB, bool WITH_OLD, bool WITH_CONV, bool FAST>
__device__ __forceinline__ void fused_step(FusedCtx<TA, TB>& c, f32x16 (&acc)[2][2], const __bf16* rbuf,
                                           __bf16* wbuf, int k0_tile, int k0_next) {
    static_for<0, 4>([&](auto fr) { c.template read_frag<fr.value>(c.F0, rbuf, 0); });
    __bf16* dst = nullptr;
    static_for<0, 48>([&](auto gc) {
        constexpr int g = gc.value;
        if constexpr (g < 24) {
            if constexpr (WITH_OLD) mfma_w<g>(c.F1, acc);
        } else {
            mfma_w<g - 24>(c.F0, acc);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g < 8) c.template read_frag<4 + g>(c.F0, rbuf, 0);
        if constexpr (g >= 24 && g < 36) c.template read_frag<g - 24>(c.F1, rbuf, 1);
        if constexpr (WITH_CONV) {
            constexpr int it = g / 6, stp = g % 6;
            if constexpr (stp == 0) c.template item_begin<it, FAST>(k0_tile, k0_next, wbuf, dst);
            split_step<stp>(c.st, dst);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}

template <bool TA, bool TB>
__global__ __launch_bounds__(THREADS) void gemm_split_fused_kernel(SplitArgs g) {
    extern __shared__ __attribute__((aligned(16))) __bf16 smem[];           // [2][A planes | B planes]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bx, by, z;
    tile_of_block(gridDim.x, gridDim.y, g.xcd_order, bx, by, z);
    const int m0 = by * BM, n0 = bx * BN;
    const int kt0 = z * g.k_tiles_per_split;
    const int kt_total = (g.K + BK - 1) / BK;
    const int kt1 = min(kt_total, kt0 + g.k_tiles_per_split);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kt0 < kt1) {
        FusedCtx<TA, TB> c;
        c.M = g.M; c.N = g.N; c.K = g.K; c.m0 = m0; c.n0 = n0; c.tid = tid;
        c.a_edge = m0 + BM > g.M;
        c.b_edge = n0 + BN > g.N;
        c.frag_a = (wm * 64 + (lane & 31)) * LDS_ROW + (lane >> 5) * 8;
        c.frag_b = (wn * 64 + (lane & 31)) * LDS_ROW + (lane >> 5) * 8;
        c.la.init(g.A, g.lda, g.M, g.K, m0, tid);
        c.lb.init(g.B, g.ldb, g.N, g.K, n0, tid);
        c.la.load(kt0 * BK, c.ra);
        c.lb.load(kt0 * BK, c.rb);
        // prologue: split tile kt0 into buffer 0 (nothing to overlap it with), loads of tile kt0+1
        {
            const int k0_next = min(kt0 + 1, kt1 - 1) * BK;
            __bf16* dst = nullptr;
            static_for<0, 48>([&](auto gc) {
                constexpr int it = gc.value / 6, stp = gc.value % 6;
                if constexpr (stp == 0) c.template item_begin<it, false>(kt0 * BK, k0_next, smem, dst);
                split_step<stp>(c.st, dst);
            });
        }
        __syncthreads();
        // the loop is peeled so that its body is ONE straight-line variant (accumulators stay in place)
        auto step = [&](int kt, auto with_old, auto with_conv, auto fast) {
            const int cur = (kt - kt0) & 1;
            const int k0_tile = (kt + 1) * BK;                        // the tile being split in this step
            const int k0_next = min(kt + 2, kt1 - 1) * BK;            // the tile being loaded (clamped, harmless)
            fused_step<TA, TB, with_old.value, with_conv.value, fast.value>(
                c, acc, smem + cur * BUF, smem + (cur ^ 1) * BUF, k0_tile, k0_next);
            __syncthreads();
        };
        using T = std::true_type;
        using F = std::false_type;
        if (kt0 + 1 < kt1) {
            step(kt0, F{}, T{}, F{});
            int kt = kt0 + 1;
            if (!c.a_edge && !c.b_edge)          // interior tile: steady state without any edge handling
                for (; kt < kt1 - 1 && (min(kt + 2, kt1 - 1) + 1) * BK <= g.K; ++kt) step(kt, T{}, T{}, T{});
            for (; kt < kt1 - 1; ++kt) step(kt, T{}, T{}, F{});
            step(kt1 - 1, T{}, F{}, F{});
        } else {
            step(kt0, F{}, F{}, F{});
        }
        static_for<0, 24>([&](auto w) { mfma_w<w.value>(c.F1, acc); });     // slab 1 of the last tile
    }
    store_tile(g, m0, n0, z, wm, wn, lane, acc);
}

// ------------------------------------------------------------------------------------------------------
// bf16 mode (renet_gemm_bf16; BASELINE config 5 "n_hidden=400 bf16"): the SAME tile, loaders and LDS image with ONE
// bf16 plane per operand -- every fp32 operand value is rounded to bf16 (RNE) on its way into LDS and the product
// is a single v_mfma_f32_32x32x16_bf16 with fp32 accumulation (standard bf16 mixed precision: bf16 multiplicands,
// fp32 sums, fp32 storage of every tensor).  1/6 of the matrix-pipe work of the bf16x6 kernels; the k-loop is then
// bound by staging, so the next tile's global loads are issued before the MFMA phase and land behind it.
// ------------------------------------------------------------------------------------------------------
template <bool CONTIG_K, bool EDGE>
__device__ __forceinline__ void store_items_1(__bf16* __restrict__ S, int rows, int K, int row0, int k0, int tid,
                                              const float4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row, k;
        item_pos<CONTIG_K>(tid + THREADS * i, row, k);
        float4 v = r[i];
        if constexpr (EDGE) v = fix_item<CONTIG_K>(v, rows, K, row0, k0, row, k);
        const f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
        *reinterpret_cast<uint2*>(S + row * LDS_ROW + k) =
            pack4(__builtin_convertvector(lo, bf16x2), __builtin_convertvector(hi, bf16x2));
    }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(THREADS) void gemm_bf16_kernel(SplitArgs g) {
    __shared__ __attribute__((aligned(16))) __bf16 sA[2][PLANE];
    __shared__ __attribute__((aligned(16))) __bf16 sB[2][PLANE];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bx, by, z;
    tile_of_block(gridDim.x, gridDim.y, g.xcd_order, bx, by, z);
    const int m0 = by * BM, n0 = bx * BN;
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
        const int buf = (kt - kt0) & 1;                 // LDS double buffer: ONE barrier per k-tile
        const bool k_edge = (kt + 1) * BK > g.K;
        if (a_edge || k_edge) store_items_1<A_CK, true>(sA[buf], g.M, g.K, m0, kt * BK, tid, ra);
        else store_items_1<A_CK, false>(sA[buf], g.M, g.K, m0, kt * BK, tid, ra);
        if (b_edge || k_edge) store_items_1<B_CK, true>(sB[buf], g.N, g.K, n0, kt * BK, tid, rb);
        else store_items_1<B_CK, false>(sB[buf], g.N, g.K, n0, kt * BK, tid, rb);
        __syncthreads();
        if (kt + 1 < kt1) {                             // in flight behind the MFMAs
            la.load((kt + 1) * BK, ra);
            lb.load((kt + 1) * BK, rb);
        }
#pragma unroll
        for (int slab = 0; slab < 2; ++slab) {
            const int ko = slab * 16 + ksel;
            bf16x8 a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = *reinterpret_cast<const bf16x8*>(&sA[buf][arow + t * 32 * LDS_ROW + ko]);
                b[t] = *reinterpret_cast<const bf16x8*>(&sB[buf][brow + t * 32 * LDS_ROW + ko]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    store_tile(g, m0, n0, z, wm, wn, lane, acc);
}

// ------------------------------------------------------------------------------------------------------
// LDS-DMA ring shared by the bf16-storage GEMM below (the three-plane "planes" GEMM that introduced it in round 2 was
// measured against the in-loop split, not adopted -- DESIGN 4b -- and removed in round 5): operands are bf16 matrices
// [rows][cols] in HBM, both dims padded with zeros to multiples of 128 (no edge handling in the loop); a k-tile is
// global_load_lds_dwordx4 pieces (no VGPRs, no VALU, no ds_write) into a 3-deep LDS ring with ONE raw s_barrier per
// k-tile.  An operand is consumed in one of two roles:
//   K-CONTIGUOUS (tr = 0): rows = the operand's M (N) index, cols = K.  Tile image in LDS: [128 rows][32 k], 64-byte
//       rows, the four 16-byte chunks of a row XOR-swizzled by (row >> 2) & 3 -- applied on the SOURCE address of
//       the DMA (the LDS side of global_load_lds is lane-linear) -- which makes every ds_read_b128 fragment read
//       conflict free.
//   K-STRIDED (tr = 1): rows = K, cols = the operand's M (N) index (the tensor as stored when the contraction runs
//       over its rows: dW = dlogits^T feat, dfeat = dlogits W).  Tile image [32 k][128 cols], 256-byte rows, the
//       sixteen 16-byte chunks XOR-swizzled by 4 * (k & 3); fragments come from two ds_read_b64_tr_b16 each (gfx950's
//       transposing LDS read: in a 16-lane group, lane i receives element i & 3 of the 8-byte chunks addressed by
//       lanes (i >> 2) + 4 j, j = 0..3 -- measured with tools/probes/tr_probe.hip).
// ------------------------------------------------------------------------------------------------------

constexpr int P3_SLOTS = 3;
constexpr int P3_LOADERS = 4;                           // loader waves per workgroup (48 DMA pieces per k-tile)
constexpr int P3_THREADS = 64 * (4 + P3_LOADERS);

typedef short s16x4 __attribute__((ext_vector_type(4)));

// Fragment of MFMA tile t of a wave whose 64 rows (K-contiguous image) / 64 columns (K-strided image) start at `w0`
// inside the operand's 128-wide tile; `tile` = start of the plane's 8 KB image.
template <bool TR>
__device__ __forceinline__ bf16x8 p3_fragment(const char* tile, int lane, int w0, int t, int slab) {
    if constexpr (!TR) {
        const int row = w0 + 32 * t + (lane & 31);
        const int chunk = (2 * slab + (lane >> 5)) ^ ((row >> 2) & 3);
        return *reinterpret_cast<const bf16x8*>(tile + row * 64 + chunk * 16);
    } else {
        const int s = lane & 15;
        const int col = w0 + 32 * t + 16 * ((lane >> 4) & 1) + 4 * (s & 3);  // first of the 4 columns of the chunk
        bf16x8 r;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int kk = slab * 16 + 8 * (lane >> 5) + 4 * q + (s >> 2);
            const int phys16 = (col >> 3) ^ (4 * (kk & 3));
            const char* p = tile + kk * 256 + phys16 * 16 + ((col >> 2) & 1) * 8;
            const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
#pragma unroll
            for (int j = 0; j < 4; ++j) r[4 * q + j] = __builtin_bit_cast(__bf16, (short)v[j]);
        }
        return r;
    }
}

// Wave specialisation: a workgroup is 8 waves -- waves 0..3 (one per SIMD) only read fragments and issue MFMAs, waves
// 4..7 (their SIMD partners) only issue the LDS-DMA.  A global_load_lds costs the issuing wave ~60-100 cycles of issue
// time (12 per k-tile and wave: as much as half the tile's MFMA time when the MFMA wave has to issue them itself,
// measured: 3200 cycles per k-tile against 1536 of matrix-pipe time); from a partner wave it overlaps the MFMAs.
//   ring of 3 slots, ONE s_barrier per round, all 8 waves:
//     loader, round k : issue tile k + 2 into slot (k + 2) % 3   [held tile k - 1: every MFMA wave finished reading it
//                                                                 before barrier k - 1]
//                       s_waitcnt vmcnt(12)                       [tile k + 1 landed; tile k + 2 stays in flight]
//                       barrier k
//     MFMA wave, round k : fragments of tile k (slot k % 3), 48 MFMAs with the slab-1 reads behind the first 12,
//                          s_waitcnt lgkmcnt(0), barrier k
// ------------------------------------------------------------------------------------------------------
// bf16-STORAGE GEMM (renet_gemm_bf16s, BASELINE config 5 "n_hidden=400 bf16"): both operands are bf16 matrices in
// HBM ([Rp][Cp], padded with zeros to multiples of 128), ONE product per fragment
// pair, fp32 accumulation.  The LDS-DMA ring, wave specialisation, images and swizzles described above; a
// ring slot holds TWO 32-wide k sub-tiles per operand (4 x 8 KB = 32 KB per slot, 3 slots), so a stage is 128 x 128 x 64: 16 MFMAs per MFMA wave and barrier.  With one product per
// element pair the k-loop moves 16 KB of operands per 1 MFLOP: the kernel is bound by the L2 -> LDS stream, not by
// the matrix pipe.  Either operand may be consumed K-contiguous or K-strided (ds_read_b64_tr_b16).
// ------------------------------------------------------------------------------------------------------
struct Bf16sArgs {
    const __bf16* A;
    const __bf16* B;
    int lda, ldb;               // row stride (elements)
    SplitArgs out;              // M, N, K, C, ldc, alpha, beta, bias, split-K fields; k_tiles_per_split in 64-wide STAGES
};

constexpr int B1_SUB = 2;                                 // k sub-tiles per operand and ring slot
constexpr int B1_BK = 32 * B1_SUB;                        // k per stage
constexpr int B1_STAGE = 2 * B1_SUB * 8192;
constexpr int B1_SLOTS = 3;
constexpr size_t B1_LDS = (size_t)B1_STAGE * B1_SLOTS;
constexpr size_t B1_LDS_TALL = (size_t)B1_SUB * (256 * 64 + 8192) * B1_SLOTS;

// TALL: 256 x 128 tile, 8 MFMA waves (4 x 2, two per SIMD: one wave's barrier wait is covered by its partner's
// MFMAs) + 4 loader waves; A images are 16 KB (256 rows K-contiguous / 256 columns K-strided).  0.75x the operand
// bytes per flop of the 128 x 128 tile.
template <bool A_TR, bool B_TR, bool TALL>
__global__ __launch_bounds__(TALL ? 768 : P3_THREADS) void gemm_bf16s_kernel(Bf16sArgs pa) {
    constexpr int TBM = TALL ? 256 : 128;
    constexpr int MW = TALL ? 8 : 4;                       // MFMA waves
    constexpr int AIMG = TBM * 64;                         // bytes of one A sub-tile image
    constexpr int APIECES = TBM / 16;                      // 1 KB DMA pieces per A image
    constexpr int STAGE = B1_SUB * (AIMG + 8192);
    constexpr int NPIECE = B1_SUB * (APIECES + 8);
    constexpr int PER = NPIECE / P3_LOADERS;               // 8 (128-row tile) or 12 (256-row tile)
    extern __shared__ __attribute__((aligned(16))) char ring1[];
    const SplitArgs& g = pa.out;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by, z;
    tile_of_block(gridDim.x, gridDim.y, g.xcd_order, bx, by, z);
    const int m0 = by * TBM, n0 = bx * BN;
    const int st_total = (g.K + B1_BK - 1) / B1_BK;
    const int s0 = z * g.k_tiles_per_split;
    const int s1 = min(st_total, s0 + g.k_tiles_per_split);
    const int nk = max(s1 - s0, 0);

    if (wave >= MW) {
        if (nk == 0) return;
        const int lw = wave - MW;
        const __bf16* src[PER];
        int dst[PER];
        bool is_b[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int id = lw + P3_LOADERS * i;
            // pieces 0 .. B1_SUB * APIECES - 1: A (sub-tile major), then B
            const int opnd = id >= B1_SUB * APIECES ? 1 : 0;
            const int idl = opnd ? id - B1_SUB * APIECES : id;
            const int per_sub = opnd ? 8 : APIECES;
            const int sub = idl / per_sub, piece = idl % per_sub;
            is_b[i] = opnd != 0;
            const bool tr = opnd ? B_TR : A_TR;
            const __bf16* base = opnd ? pa.B : pa.A;
            const int ld = opnd ? pa.ldb : pa.lda;
            const int r0 = opnd ? n0 : m0;
            const size_t k0 = (size_t)s0 * B1_BK + sub * 32;
            size_t off;
            int img_off;
            if (!tr) {                                   // [rows][32 k]: piece = 16 rows x 64 B
                const int row = 16 * piece + (lane >> 2);
                const int chunk = (lane & 3) ^ ((row >> 2) & 3);
                off = (size_t)(r0 + row) * ld + k0 + chunk * 8;
                img_off = piece * 1024;
            } else {                                     // [32 k][cols]: 128-column panels of 8 KB, piece = 4 k x 256 B
                const int panel = piece >> 3, pc = piece & 7;
                const int kk = 4 * pc + (lane >> 4);
                const int log16 = (lane & 15) ^ (4 * (kk & 3));
                off = (k0 + kk) * ld + r0 + panel * 128 + log16 * 8;
                img_off = panel * 8192 + pc * 1024;
            }
            src[i] = base + off;
            dst[i] = (opnd ? B1_SUB * AIMG + sub * 8192 : sub * AIMG) + img_off;
        }
        const size_t a_step = A_TR ? (size_t)B1_BK * pa.lda : (size_t)B1_BK;
        const size_t b_step = B_TR ? (size_t)B1_BK * pa.ldb : (size_t)B1_BK;
        auto issue_all = [&](int slot) {
            char* base = ring1 + slot * STAGE;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[i],
                                                 (__attribute__((address_space(3))) void*)(base + dst[i]), 16, 0, 0);
                src[i] += is_b[i] ? b_step : a_step;
            }
        };
        static_assert(PER == 8 || PER == 12, "counted waits below");
        auto wait_one_stage = [&]() {
            if constexpr (PER == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        };
        issue_all(0);
        if (nk > 1) {
            issue_all(1);
            wait_one_stage();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                                     // barrier -1: stage 0 visible
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 2 < nk) {
                issue_all((kt + 2) % B1_SLOTS);
                wait_one_stage();
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                                 // barrier kt: stage kt + 1 visible
        }
        return;
    }

    // ---------------- MFMA waves ----------------
    const int wm = wave >> 1, wn = wave & 1;               // wm: 0..1 (128-row tile) or 0..3 (256-row tile)
    int offA[2][2], offB[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if constexpr (!A_TR) {
                const int row = wm * 64 + 32 * t + (lane & 31);
                offA[t][u] = row * 64 + (((2 * u + (lane >> 5)) ^ ((row >> 2) & 3)) * 16);
            } else {
                const int sl = lane & 15;
                const int colf = wm * 64 + 32 * t + 16 * ((lane >> 4) & 1) + 4 * (sl & 3);
                const int col = colf & 127;                  // inside its 128-column panel
                const int kk = 8 * (lane >> 5) + 4 * u + (sl >> 2);
                offA[t][u] = (colf >> 7) * 8192 + kk * 256 + (((col >> 3) ^ (4 * (kk & 3))) * 16) + ((col >> 2) & 1) * 8;
            }
            if constexpr (!B_TR) {
                const int row = wn * 64 + 32 * t + (lane & 31);
                offB[t][u] = row * 64 + (((2 * u + (lane >> 5)) ^ ((row >> 2) & 3)) * 16);
            } else {
                const int sl = lane & 15;
                const int col = wn * 64 + 32 * t + 16 * ((lane >> 4) & 1) + 4 * (sl & 3);
                const int kk = 8 * (lane >> 5) + 4 * u + (sl >> 2);
                offB[t][u] = kk * 256 + (((col >> 3) ^ (4 * (kk & 3))) * 16) + ((col >> 2) & 1) * 8;
            }
        }
    auto frag_tr = [&](const char* img, const int (&off)[2], int slab) {
        bf16x8 r;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(img + off[q] + slab * 4096));
#pragma unroll
            for (int j = 0; j < 4; ++j) r[4 * q + j] = __builtin_bit_cast(__bf16, (short)v[j]);
        }
        return r;
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nk > 0) __builtin_amdgcn_s_barrier();                             // barrier -1
    for (int kt = 0; kt < nk; ++kt) {
        const char* st = ring1 + (kt % B1_SLOTS) * STAGE;
        bf16x8 fa[B1_SUB][2][2], fb[B1_SUB][2][2];                        // [sub][slab][t]: all 16 reads issued up front
#pragma unroll
        for (int sb = 0; sb < B1_SUB; ++sb)
#pragma unroll
            for (int slab = 0; slab < 2; ++slab)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const char* ia = st + sb * AIMG;
                    const char* ib = st + B1_SUB * AIMG + sb * 8192;
                    if constexpr (!A_TR) fa[sb][slab][t] = *reinterpret_cast<const bf16x8*>(ia + offA[t][slab]);
                    else fa[sb][slab][t] = frag_tr(ia, offA[t], slab);
                    if constexpr (!B_TR) fb[sb][slab][t] = *reinterpret_cast<const bf16x8*>(ib + offB[t][slab]);
                    else fb[sb][slab][t] = frag_tr(ib, offB[t], slab);
                }
#pragma unroll
        for (int sb = 0; sb < B1_SUB; ++sb)
#pragma unroll
            for (int slab = 0; slab < 2; ++slab)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[sb][slab][i], fb[sb][slab][j], acc[i][j], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                     // barrier kt
    }
    store_tile(g, m0, n0, z, wm, wn, lane, acc);
}

// fp32 [R, C] (row stride ldx) -> ONE bf16 matrix [Rp][Cp] (RNE), padding written as zeros
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ X, int R, int C, int ldx, int Rp,
                                                        int Cp, __bf16* __restrict__ P) {
    const size_t total = (size_t)Rp * Cp / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / (Cp / 4)), c = (int)(i % (Cp / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < R) {
            const float* x = X + (size_t)row * ldx + c;
            if (c + 3 < C && ((ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0)) {
                v = *reinterpret_cast<const float4*>(x);
            } else {
                if (c < C) v.x = x[0];
                if (c + 1 < C) v.y = x[1];
                if (c + 2 < C) v.z = x[2];
                if (c + 3 < C) v.w = x[3];
            }
        }
        const bf16x2 lo = __builtin_convertvector(f32x2{v.x, v.y}, bf16x2);
        const bf16x2 hi = __builtin_convertvector(f32x2{v.z, v.w}, bf16x2);
        *reinterpret_cast<uint2*>(P + (size_t)row * Cp + c) = pack4(lo, hi);
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

// The same reduction for SMALL outputs with many slices (dW of the 200x200 / 600x200 weights: 40k-120k elements,
// 39-128 slices): one thread per element leaves most of the chip idle and walks the slices serially, so the 4
// waves of a workgroup split the slices (wave w takes z = w, w+4, ...: every load a coalesced 256-byte row
// segment) and combine through LDS in a fixed order.
__global__ __launch_bounds__(256) void split_reduce4_kernel(const float* __restrict__ partial, int split_k,
                                                            int M, int N, float alpha, float beta,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ C, int ldc) {
    __shared__ float red[4][64];
    const size_t total = (size_t)M * N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f;
    if (i < total) {
        int zz = wave;
        for (; zz + 4 < split_k; zz += 8) {
            s0 += partial[(size_t)zz * total + i];
            s1 += partial[(size_t)(zz + 4) * total + i];
        }
        if (zz < split_k) s0 += partial[(size_t)zz * total + i];
    }
    red[wave][lane] = s0 + s1;
    __syncthreads();
    if (wave == 0 && i < total) {
        const int m = (int)(i / N), n = (int)(i % N);
        const float s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        float v = alpha * s + (bias ? bias[n] : 0.f);
        float* p = C + (size_t)m * ldc + n;
        if (beta != 0.f) v += beta * (*p);
        *p = v;
    }
}

// RENET_GEMM_TILE_ORDER = 0 | 1: plain order | XCD-aware order (default; the SAME boolean gemm.hip reads).  The panel width
// of the XCD-aware order is a separate knob, RENET_GEMM_PANEL_W = 1..64 (default 8; setting it also pins the width, i.e.
// panel_width() below leaves it alone).
bool panel_w_pinned() { return getenv("RENET_GEMM_PANEL_W") != nullptr; }
int tile_order() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("RENET_GEMM_TILE_ORDER");
        if (e && e[0] == '0') v = 0;
        else {
            const char* w = getenv("RENET_GEMM_PANEL_W");
            v = w ? atoi(w) : 8;
            if (v < 1 || v > 64) v = 8;
        }
    }
    return v;
}

// Panel width for an UN-SPLIT grid that takes several rounds of tiles per XCD.  The short operand is swept once per round
// by every XCD; when all of it (ns tiles) does not fit the 4 MB L2 next to the streamed long operand, that cyclic sweep
// misses (PMC, tools/pmc_by_shape.py: the 256-row logits GEMM fetched 313 MB for 60 MB of operands, 258 MB of it the
// 4.9 MB `feat`).  Narrower panels keep one panel of the short operand resident (<= 2.5 MB) and re-read the long operand
// once per extra panel: taken when that costs less than the sweep does (so NOT for dW = dlogits^T feat, whose long
// operand is 189 MB).  An explicit RENET_GEMM_PANEL_W, the plain order, or a split k range leaves the width alone.
int panel_width(int base, int nbx, int nby, int tile_m, int K, int split_k, int slots_per_xcd) {
    if (base != 8 || split_k != 1 || panel_w_pinned()) return base;
    const bool short_is_m = nby <= nbx;
    const int ns = short_is_m ? nby : nbx, nl = short_is_m ? nbx : nby;
    const double slab = (double)(short_is_m ? tile_m : BN) * K * 4.0;           // short-operand bytes of one tile row / column
    const double long_total = (double)(short_is_m ? BN : tile_m) * K * 4.0 * nl;
    const double short_total = slab * ns;
    const double rounds = (double)nbx * nby / (8.0 * slots_per_xcd);
    if (short_total <= 3.0e6 || rounds <= 1.0) return base;
    const int w = max(1, min(8, (int)(2.5e6 / slab)));
    const int w0 = min(ns, 8);
    if (w >= w0) return base;
    const int extra_panels = (ns + w - 1) / w - (ns + w0 - 1) / w0;
    if (long_total * extra_panels >= short_total * rounds * 8.0) return base;
    return w;
}

template <bool TA, bool TB>
int launch_fused(const SplitArgs& g, dim3 grid, hipStream_t st) {
    static bool attr_set = false;      // benign race: the attribute is idempotent
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_split_fused_kernel<TA, TB>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)FUSED_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    RENET_LAUNCH((gemm_split_fused_kernel<TA, TB>), grid, dim3(THREADS), FUSED_LDS, st, g);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

template <bool TA, bool TB, bool RAW>
int launch_tall(const SplitArgs& g, dim3 grid, hipStream_t st) {
    static bool attr_set = false;      // benign race: the attribute is idempotent
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_split_tall_kernel<TA, TB, RAW>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)TALL_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    RENET_LAUNCH((gemm_split_tall_kernel<TA, TB, RAW>), grid, dim3(THREADS_T), TALL_LDS, st, g);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

// 256-row tiles when the output has enough of them to fill the chip a few times (RENET_GEMM_TALL=0 disables,
// =<n> sets the minimum tile count)
bool use_tall(int ta, int M, int nbx, int split_k) {
    static int min_tiles = -1;
    static bool forced = false;
    if (min_tiles < 0) {
        const char* e = getenv("RENET_GEMM_TALL");
        forced = e != nullptr;
        min_tiles = e ? atoi(e) : 1000;             // measured: logits 2048 x 23033 x 600 (1440 tiles) 395 -> 362 us;
                                                    // dW 23033 x 600 x 2048 (450 tiles: 1.76 rounds) 447 -> 480 us
        if (e && min_tiles == 0) min_tiles = 0x7fffffff;
    }
    const long tiles = (long)nbx * ((M + BMT - 1) / BMT) * split_k;
    if (forced) return tiles >= min_tiles;
    // round 4: as for the f16x3 kernels (use_tall_h3), K-contiguous A with a split k range from 200 tiles on (dfeat)
    return tiles >= min_tiles || (!ta && split_k >= 4 && tiles >= 200);
}

// f16x3 kernels: the 256-row tile from RENET_H3_TALL tiles on (default 1000, as for the bf16x6 kernels; 0 disables).
// Measured on the step's shapes (tools/sessions/r03_s20.sh): the tall tile wins for K-contiguous A from 200 tiles on
// when the k range is split (dfeat 2048 x 600 x 23033 / 6: 286 -> 259 us) and loses for K-strided A (dW 23033 x 600 x
// 2048: 293 -> 319 us; the split-K weight gradients 100 -> 178 us).
bool use_tall_h3(int ta, int M, int nbx, int split_k) {
    static int min_tiles = -1;
    static bool forced = false;
    if (min_tiles < 0) {
        const char* e = getenv("RENET_H3_TALL");
        forced = e != nullptr;
        min_tiles = e ? atoi(e) : 1000;
        if (e && min_tiles == 0) min_tiles = 0x7fffffff;
    }
    const long tiles = (long)nbx * ((M + 255) / 256) * split_k;
    if (forced) return tiles >= min_tiles;
    return !ta && (tiles >= min_tiles || (split_k >= 4 && tiles >= 200));
}

// Which k-loop: the fused kernel (one workgroup per CU, 122.9 KB LDS) when the whole grid fits in ONE round of
// 256 workgroups -- there a lone workgroup finishes a k-tile in ~2500 cycles against ~3900 for the two-phase
// kernel (MI355X: 51 vs 35 TFLOP/s on a 64-tile problem, 155 vs 126 on 256 tiles) -- and the two-phase kernel
// with two co-resident workgroups per CU beyond that, where its second workgroup covers the prologue, the
// C-store epilogue and the barrier waits of the first (short K loops: 133 vs 117 TFLOP/s on the 1440-tile
// K=600 logits GEMM).  RENET_GEMM_KERNEL=fused|split forces one of them (tools/gemm_bench.py).
int kernel_choice(int ntiles) {
    static int forced = -2;
    if (forced == -2) {
        const char* e = getenv("RENET_GEMM_KERNEL");
        forced = !e ? -1 : e[0] == 's' ? 1 : e[0] == 'f' ? 0 : -1;
    }
    if (forced >= 0) return forced;
    return ntiles <= 256 ? 0 : 1;
}

#include "gemm_h3.h"
#include "gemm_p6.h"

}  // namespace

namespace {
template <bool A_TR, bool B_TR, bool TALL>
int launch_bf16s(const Bf16sArgs& pa, dim3 grid, hipStream_t st) {
    constexpr size_t lds = TALL ? B1_LDS_TALL : B1_LDS;
    static bool attr_set = false;      // benign race: the attribute is idempotent
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16s_kernel<A_TR, B_TR, TALL>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    RENET_LAUNCH((gemm_bf16s_kernel<A_TR, B_TR, TALL>), grid, dim3(TALL ? 768 : P3_THREADS), lds, st, pa);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

}  // namespace

extern "C" {

#ifdef RENET_GEMM_TRACE
int renet_gemm_trace_set(unsigned long long* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &buf, sizeof(buf));
}
#endif

static bool skinny_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("RENET_GEMM_SKINNY");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v != 0;
}

static bool split_raw_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("RENET_GEMM_SPLIT_RAW");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v != 0;
}

// What the bf16x6 / bf16 launcher does for a problem: ONE function decides, the launcher executes it and
// renet_gemm_split_plan reports it (host logic only -- testable without a GPU).
struct SplitPlan {
    int kernel;            // 0 fused (one workgroup per CU), 1 two-phase 128 x 128, 2 two-phase 256 x 128, 3 weight-resident
                           // (gemm_skinny.hip), 4 single-plane bf16 (renet_gemm_bf16)
    int raw;               // two-phase kernels: operands through raw buffer descriptors
    int xcd_order;         // 0 plain tile order, w: XCD-aware order with panels of <= w tiles
    int gx, gy, gz;        // grid
    int split_k;           // clamped split
};

static SplitPlan plan_split(bool bf16_mode, int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B,
                            int ldb, int split_k) {
    SplitPlan p;
    const int kt_total = (K + BK - 1) / BK;
    if (split_k < 1) split_k = 1;
    if (split_k > kt_total) split_k = max(kt_total, 1);
    p.split_k = split_k;
    p.raw = 0;
    p.xcd_order = tile_order();
    const int nbx = (N + BN - 1) / BN, nby = (M + BM - 1) / BM;
    p.gx = nbx; p.gy = nby; p.gz = split_k;
    // tall activation x small weight (K <= 208, N <= 256): the weight-resident kernel of gemm_skinny.hip
    // (RENET_GEMM_SKINNY=0 in the environment keeps the general kernels, for A/B runs)
    if (!bf16_mode && split_k == 1 && skinny_enabled() && renet_gemm_skinny_eligible(ta, M, N, K, A, lda, B, ldb, tb)) {
        p.kernel = 3;
        return p;
    }
    const int ntiles = nbx * nby * split_k;
    int choice = bf16_mode ? 2 : kernel_choice(ntiles);
    // the fused kernel addresses with 32-bit element offsets and float4 loads along a contiguous K
    if (choice == 0 && (K < 4 || (size_t)(ta ? K : M) * lda >= (1u << 31) || (size_t)(tb ? N : K) * ldb >= (1u << 31)))
        choice = 1;
    if (choice == 0) { p.kernel = 0; return p; }
    if (choice == 2) { p.kernel = 4; return p; }
    // two-phase kernels: raw buffer descriptors (32-bit byte offsets) while both operands reach less than 2^30 elements
    // (RENET_GEMM_SPLIT_RAW=0: the generic 64-bit loader, for A/B runs)
    p.raw = split_raw_enabled() && (size_t)(ta ? K : M) * lda < ((size_t)1 << 30) &&
            (size_t)(tb ? N : K) * ldb < ((size_t)1 << 30);
    if (use_tall(ta, M, nbx, split_k)) {
        p.kernel = 2;
        p.gy = (M + BMT - 1) / BMT;
        p.xcd_order = panel_width(p.xcd_order, nbx, p.gy, BMT, K, split_k, 32);
    } else {
        p.kernel = 1;
        p.xcd_order = panel_width(p.xcd_order, nbx, nby, BM, K, split_k, 64);
    }
    return p;
}

static int gemm_split_launch(bool bf16_mode, int ta, int tb, int M, int N, int K, float alpha, const float* A, int lda,
                              const float* B, int ldb, float beta, float* C, int ldc, const float* bias,
                              int split_k, float* workspace, size_t workspace_bytes, void* stream) {
    if (M < 0 || N < 0 || K < 1 || lda <= 0 || ldb <= 0 || ldc < N) return RENET_ERR_BADARG;
    if (M == 0 || N == 0) return RENET_OK;
    const SplitPlan p = plan_split(bf16_mode, ta, tb, M, N, K, A, lda, B, ldb, split_k);
    split_k = p.split_k;
    const int kt_total = (K + BK - 1) / BK;
    if (split_k > 1 && workspace_bytes < renet_gemm_workspace(M, N, split_k)) return RENET_ERR_WORKSPACE;
    if (p.kernel == 3) return renet_gemm_skinny_launch(tb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, stream);
    SplitArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.alpha = alpha; g.beta = beta;
    g.split_k = split_k;
    g.k_tiles_per_split = max(1, (kt_total + split_k - 1) / split_k);
    g.partial = workspace;
    g.xcd_order = p.xcd_order;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(p.gx, p.gy, p.gz);
    const bool raw = p.raw != 0;
    int e = RENET_OK;
    if (p.kernel == 0) {
        if (!ta && !tb) e = launch_fused<false, false>(g, grid, st);
        else if (!ta && tb) e = launch_fused<false, true>(g, grid, st);
        else if (ta && !tb) e = launch_fused<true, false>(g, grid, st);
        else e = launch_fused<true, true>(g, grid, st);
    } else if (p.kernel == 4) {
        if (!ta && !tb) RENET_LAUNCH((gemm_bf16_kernel<false, false>), grid, dim3(THREADS), 0, st, g);
        else if (!ta && tb) RENET_LAUNCH((gemm_bf16_kernel<false, true>), grid, dim3(THREADS), 0, st, g);
        else if (ta && !tb) RENET_LAUNCH((gemm_bf16_kernel<true, false>), grid, dim3(THREADS), 0, st, g);
        else RENET_LAUNCH((gemm_bf16_kernel<true, true>), grid, dim3(THREADS), 0, st, g);
    } else if (p.kernel == 2) {
#define RENET_TALL_LAUNCH(RAWV)                                                     \
        do {                                                                        \
            if (!ta && !tb) e = launch_tall<false, false, RAWV>(g, grid, st);       \
            else if (!ta && tb) e = launch_tall<false, true, RAWV>(g, grid, st);    \
            else if (ta && !tb) e = launch_tall<true, false, RAWV>(g, grid, st);    \
            else e = launch_tall<true, true, RAWV>(g, grid, st);                    \
        } while (0)
        if (raw) RENET_TALL_LAUNCH(true);
        else RENET_TALL_LAUNCH(false);
#undef RENET_TALL_LAUNCH
    } else {
#define RENET_SPLIT_LAUNCH(RAWV)                                                                                    \
        do {                                                                                                        \
            if (!ta && !tb) RENET_LAUNCH((gemm_split_kernel<false, false, RAWV>), grid, dim3(THREADS), 0, st, g);   \
            else if (!ta && tb) RENET_LAUNCH((gemm_split_kernel<false, true, RAWV>), grid, dim3(THREADS), 0, st, g); \
            else if (ta && !tb) RENET_LAUNCH((gemm_split_kernel<true, false, RAWV>), grid, dim3(THREADS), 0, st, g); \
            else RENET_LAUNCH((gemm_split_kernel<true, true, RAWV>), grid, dim3(THREADS), 0, st, g);                \
        } while (0)
        if (raw) RENET_SPLIT_LAUNCH(true);
        else RENET_SPLIT_LAUNCH(false);
#undef RENET_SPLIT_LAUNCH
    }
    if (e != RENET_OK) return e;
    RENET_LAUNCH_CHECK();
    if (split_k > 1) {
        const size_t total = (size_t)M * N;
        if (total <= (size_t)256 * 1024 && split_k >= 8) {
            RENET_LAUNCH(split_reduce4_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, workspace,
                               split_k, M, N, alpha, beta, bias, C, ldc);
        } else {
            int blocks = (int)min((size_t)2048, (total + 255) / 256);
            RENET_LAUNCH(split_reduce_kernel, dim3(blocks), dim3(256), 0, st, workspace, split_k, M, N, alpha,
                               beta, bias, C, ldc);
        }
        RENET_LAUNCH_CHECK();
    }
    return RENET_OK;
}


int renet_gemm_f32_split(int ta, int tb, int M, int N, int K, float alpha, const float* A, int lda,
                         const float* B, int ldb, float beta, float* C, int ldc, const float* bias,
                         int split_k, float* workspace, size_t workspace_bytes, void* stream) {
    return gemm_split_launch(false, ta, tb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, split_k, workspace,
                              workspace_bytes, stream);
}

int renet_gemm_split_plan(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                          int split_k, int* plan) {
    if (!plan || M < 1 || N < 1 || K < 1 || lda <= 0 || ldb <= 0) return RENET_ERR_BADARG;
    const SplitPlan p = plan_split(false, ta, tb, M, N, K, A, lda, B, ldb, split_k);
    plan[0] = p.kernel; plan[1] = p.raw; plan[2] = p.xcd_order; plan[3] = p.gx; plan[4] = p.gy; plan[5] = p.gz;
    plan[6] = p.split_k;
    plan[7] = 0;
    return RENET_OK;
}

int renet_gemm_bf16(int ta, int tb, int M, int N, int K, float alpha, const float* A, int lda,
                    const float* B, int ldb, float beta, float* C, int ldc, const float* bias,
                    int split_k, float* workspace, size_t workspace_bytes, void* stream) {
    return gemm_split_launch(true, ta, tb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, split_k, workspace,
                              workspace_bytes, stream);
}

int renet_maxabs_blocks(int rows, int cols, int ld) {
    if (rows <= 0 || cols <= 0) return 1;
    const size_t total = (size_t)rows * cols;
    if (ld == cols && (total & 3) == 0) return (int)max((size_t)1, min((size_t)256, total / 4096));
    return min(rows, 256);
}

int renet_maxabs_partials(const float* x, int rows, int cols, int ld, float* part, void* stream) {
    if (rows < 0 || cols < 0 || ld < cols || !part) return RENET_ERR_BADARG;
    if (rows == 0 || cols == 0) {
        hipError_t e = hipMemsetAsync(part, 0, sizeof(float), (hipStream_t)stream);
        return e == hipSuccess ? RENET_OK : (int)e;
    }
    const int blocks = renet_maxabs_blocks(rows, cols, ld);
    const bool flat = ld == cols && (((size_t)rows * cols) & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    if (flat) RENET_LAUNCH(maxabs_partials_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, rows, cols,
                           (size_t)ld, part);
    else RENET_LAUNCH(maxabs_partials_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, rows, cols,
                      (size_t)ld, part);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

int renet_maxabs_partials_multi(const void* jobs, int n_jobs, void* stream) {
    if (n_jobs < 0 || (n_jobs > 0 && !jobs)) return RENET_ERR_BADARG;
    if (n_jobs == 0) return RENET_OK;
    RENET_LAUNCH(maxabs_multi_kernel, dim3(256, n_jobs), dim3(256), 0, (hipStream_t)stream, (const MaxabsJob*)jobs);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

int renet_gemm_f32_h3(int ta, int tb, int M, int N, int K, float alpha, const float* A, int lda, const float* B,
                      int ldb, float beta, float* C, int ldc, const float* bias, int split_k, float* workspace,
                      size_t workspace_bytes, const float* maxA, int nA, const float* maxB, int nB, void* stream) {
    if (M < 0 || N < 0 || K < 1 || lda <= 0 || ldb <= 0 || ldc < N) return RENET_ERR_BADARG;
    if (!maxA || !maxB || nA < 1 || nB < 1 || nA > 1024 || nB > 1024) return RENET_ERR_BADARG;
    if (M == 0 || N == 0) return RENET_OK;
    // the f16x3 loaders address with 32-bit byte offsets; operands of 4 GiB and more and the weight-resident shapes run
    // the bf16x6 kernels (same accuracy class, no bounds needed)
    const bool small_ok = (size_t)(ta ? K : M) * lda < ((size_t)1 << 30) && (size_t)(tb ? N : K) * ldb < ((size_t)1 << 30);
    if (!small_ok || ((split_k <= 1) && skinny_enabled() && renet_gemm_skinny_eligible(ta, M, N, K, A, lda, B, ldb, tb)))
        return gemm_split_launch(false, ta, tb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, split_k, workspace,
                                  workspace_bytes, stream);
    if (split_k < 1) split_k = 1;
    const int kt_total = (K + BK - 1) / BK;
    if (split_k > kt_total) split_k = max(kt_total, 1);
    if (split_k > 1 && workspace_bytes < renet_gemm_workspace(M, N, split_k)) return RENET_ERR_WORKSPACE;
    H3Args h;
    SplitArgs& g = h.g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.alpha = alpha; g.beta = beta;
    g.split_k = split_k;
    g.k_tiles_per_split = max(1, (kt_total + split_k - 1) / split_k);
    g.partial = workspace;
    g.xcd_order = tile_order();
    h.maxA = maxA; h.maxB = maxB; h.nA = nA; h.nB = nB;
    hipStream_t st = (hipStream_t)stream;
    const int nbx = (N + BN - 1) / BN;
#define RENET_H3_LAUNCH(TALLV, THR)                                                                    \
    do {                                                                                               \
        if (!ta && !tb) RENET_LAUNCH((gemm_h3_kernel<false, false, TALLV>), grid, dim3(THR), 0, st, h); \
        else if (!ta && tb) RENET_LAUNCH((gemm_h3_kernel<false, true, TALLV>), grid, dim3(THR), 0, st, h); \
        else if (ta && !tb) RENET_LAUNCH((gemm_h3_kernel<true, false, TALLV>), grid, dim3(THR), 0, st, h); \
        else RENET_LAUNCH((gemm_h3_kernel<true, true, TALLV>), grid, dim3(THR), 0, st, h);              \
    } while (0)
    if (use_tall_h3(ta, M, nbx, split_k)) {
        dim3 grid(nbx, (M + 255) / 256, split_k);
        g.xcd_order = panel_width(g.xcd_order, nbx, (int)grid.y, 256, K, split_k, 32);
        RENET_H3_LAUNCH(true, 512);
    } else {
        dim3 grid(nbx, (M + BM - 1) / BM, split_k);
        g.xcd_order = panel_width(g.xcd_order, nbx, (int)grid.y, BM, K, split_k, 64);
        RENET_H3_LAUNCH(false, 256);
    }
#undef RENET_H3_LAUNCH
    RENET_LAUNCH_CHECK();
    if (split_k > 1) {
        const size_t total = (size_t)M * N;
        if (total <= (size_t)256 * 1024 && split_k >= 8) {
            RENET_LAUNCH(split_reduce4_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, workspace,
                               split_k, M, N, alpha, beta, bias, C, ldc);
        } else {
            int blocks = (int)min((size_t)2048, (total + 255) / 256);
            RENET_LAUNCH(split_reduce_kernel, dim3(blocks), dim3(256), 0, st, workspace, split_k, M, N, alpha,
                               beta, bias, C, ldc);
        }
        RENET_LAUNCH_CHECK();
    }
    return RENET_OK;
}

size_t renet_bf16_bytes(int R, int C) {
    const size_t rp = ((size_t)R + 255) & ~(size_t)255, cp = ((size_t)C + 255) & ~(size_t)255;
    return rp * cp * sizeof(__bf16);
}

int renet_pack_bf16(const float* X, int R, int C, int ldx, void* out, void* stream) {
    if (R < 0 || C < 0 || ldx < C || !out) return RENET_ERR_BADARG;
    const int Rp = (R + 255) & ~255, Cp = (C + 255) & ~255;
    if (Rp == 0 || Cp == 0) return RENET_OK;
    const size_t total = (size_t)Rp * Cp / 4;
    const int blocks = (int)min((size_t)4096, (total + 255) / 256);
    RENET_LAUNCH(pack_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, X, R, C, ldx, Rp, Cp, (__bf16*)out);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

int renet_bf16_zero_padding(void* P, int rows, int C, int ld, int rows_alloc, void* stream) {
    if (!P || rows < 0 || C < 0 || ld < C || rows_alloc < rows) return RENET_ERR_BADARG;
    __bf16* p = (__bf16*)P;
    hipStream_t st = (hipStream_t)stream;
    const int c16 = min((C + 63) & ~63, ld), r16 = min((rows + 63) & ~63, rows_alloc);
    if (c16 > C && rows > 0) {
        hipError_t e = hipMemset2DAsync(p + C, (size_t)ld * sizeof(__bf16), 0, (size_t)(c16 - C) * sizeof(__bf16),
                                        (size_t)rows, st);
        if (e != hipSuccess) return (int)e;
    }
    if (r16 > rows) {
        hipError_t e = hipMemsetAsync(p + (size_t)rows * ld, 0, (size_t)(r16 - rows) * ld * sizeof(__bf16), st);
        if (e != hipSuccess) return (int)e;
    }
    return RENET_OK;
}

int renet_gemm_bf16s(int a_tr, int b_tr, int M, int N, int K, float alpha, const void* Ap, int lda, const void* Bp,
                     int ldb, float beta, float* C, int ldc, const float* bias, int split_k, float* workspace,
                     size_t workspace_bytes, void* stream) {
    if (M < 0 || N < 0 || K < 1 || ldc < N || !Ap || !Bp) return RENET_ERR_BADARG;
    if (M == 0 || N == 0) return RENET_OK;
    const int Mp = (M + 255) & ~255, Np = (N + 255) & ~255, Kp = (K + 255) & ~255;
    if (lda < (a_tr ? Mp : Kp) || ldb < (b_tr ? Np : Kp) || (lda & 7) || (ldb & 7)) return RENET_ERR_BADARG;
    if (split_k < 1) split_k = 1;
    const int st_total = (K + B1_BK - 1) / B1_BK;
    if (split_k > st_total) split_k = max(st_total, 1);
    if (split_k > 1 && workspace_bytes < renet_gemm_workspace(M, N, split_k)) return RENET_ERR_WORKSPACE;
    Bf16sArgs pa;
    pa.A = (const __bf16*)Ap; pa.B = (const __bf16*)Bp;
    pa.lda = lda; pa.ldb = ldb;
    SplitArgs& g = pa.out;
    g.A = nullptr; g.B = nullptr; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K;
    g.lda = 0; g.ldb = 0; g.ldc = ldc; g.alpha = alpha; g.beta = beta;
    g.split_k = split_k;
    g.k_tiles_per_split = max(1, (st_total + split_k - 1) / split_k);
    g.partial = workspace;
    g.xcd_order = tile_order();
    hipStream_t st = (hipStream_t)stream;
    // 256 x 128 tiles when they still fill the chip (>= 256 workgroups); the matrices are padded to multiples of
    // 256 in both dimensions, so a 256-row (or, K-strided, 256-column) A image never leaves the buffer --
    // RENET_BF16S_TALL=0 keeps the 128 x 128 tile
    const int nbx = (N + BN - 1) / BN;
    static int tall_ok = -1;
    if (tall_ok < 0) {
        const char* e_ = getenv("RENET_BF16S_TALL");
        tall_ok = (e_ && e_[0] == '0') ? 0 : 1;
    }
    const bool tall = tall_ok && (size_t)nbx * (Mp / 256) * split_k >= 256;
    dim3 grid(nbx, tall ? Mp / 256 : (M + BM - 1) / BM, split_k);
    int e;
    if (tall) {
        if (!a_tr && !b_tr) e = launch_bf16s<false, false, true>(pa, grid, st);
        else if (!a_tr && b_tr) e = launch_bf16s<false, true, true>(pa, grid, st);
        else if (a_tr && !b_tr) e = launch_bf16s<true, false, true>(pa, grid, st);
        else e = launch_bf16s<true, true, true>(pa, grid, st);
    } else {
        if (!a_tr && !b_tr) e = launch_bf16s<false, false, false>(pa, grid, st);
        else if (!a_tr && b_tr) e = launch_bf16s<false, true, false>(pa, grid, st);
        else if (a_tr && !b_tr) e = launch_bf16s<true, false, false>(pa, grid, st);
        else e = launch_bf16s<true, true, false>(pa, grid, st);
    }
    if (e != RENET_OK) return e;
    if (split_k > 1) {
        const size_t total = (size_t)M * N;
        if (total <= (size_t)256 * 1024 && split_k >= 8) {
            RENET_LAUNCH(split_reduce4_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, workspace,
                               split_k, M, N, alpha, beta, bias, C, ldc);
        } else {
            int blocks = (int)min((size_t)2048, (total + 255) / 256);
            RENET_LAUNCH(split_reduce_kernel, dim3(blocks), dim3(256), 0, st, workspace, split_k, M, N, alpha,
                               beta, bias, C, ldc);
        }
        RENET_LAUNCH_CHECK();
    }
    return RENET_OK;
}

// ---- planes GEMM (gemm_p6.h) -----------------------------------------------------------------------------------
size_t renet_planes_elems(int R, int C) {
    const size_t rp = ((size_t)R + 255) & ~(size_t)255, cp = ((size_t)C + 255) & ~(size_t)255;
    return rp * cp;
}

int renet_pack_planes(const float* X, int R, int C, int ldx, int ones_col, void* out, void* stream) {
    if (R < 0 || C < 0 || ldx < C || !out) return RENET_ERR_BADARG;
    const int cols = ones_col ? C + 1 : C;
    const int Rp = (R + 255) & ~255, Cp = (cols + 255) & ~255;
    if (Rp == 0 || Cp == 0) return RENET_OK;
    const size_t blocks = (size_t)(Rp >> 4) * (size_t)(Cp >> 6);       // one workgroup per tile row x 4 tiles
    if (blocks > 0x7fffffffu) return RENET_ERR_UNSUPPORTED;
    RENET_LAUNCH(pack_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, X, R, C, ldx, Rp, Cp,
                 ones_col ? C : -1, (__bf16*)out, (size_t)Rp * Cp);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

int renet_gemm_planes(int a_tr, int b_tr, int M, int N, int K, float alpha, const float* alpha_dev, const void* Ap,
                      int lda, size_t a_plane, const void* Bp, int ldb, size_t b_plane, float beta, float* C, int ldc,
                      const float* bias, float* col_out, int split_k, float* workspace, size_t workspace_bytes,
                      void* stream) {
    // N counts the logical columns of the product INCLUDING the col_out one
    const int n_main = col_out ? N - 1 : N;
    if (M < 0 || N < 0 || n_main < 0 || K < 1 || ldc < n_main || !Ap || !Bp) return RENET_ERR_BADARG;
    if (M == 0 || N == 0) return RENET_OK;
    // lda / ldb: Cp (columns of the padded stored matrix, a multiple of 16: T16 tiles, gemm_p6.h)
    const int Mp = (M + 255) & ~255, Np = (N + 127) & ~127, Kp = (K + 15) & ~15;
    if (lda < (a_tr ? Mp : Kp) || ldb < (b_tr ? Np : Kp) || (lda & 15) || (ldb & 15)) return RENET_ERR_BADARG;
    // the plane strides must cover the stored (padded) matrix
    if (a_plane < (size_t)(a_tr ? Kp : Mp) * lda || b_plane < (size_t)(b_tr ? Kp : Np) * ldb) return RENET_ERR_BADARG;
    if (split_k < 1) split_k = 1;
    const int st_total = (K + 15) / 16;                       // half-stages of 16 k (gemm_p6.h)
    if (split_k > st_total) split_k = max(st_total, 1);
    if (split_k > 1 && workspace_bytes < renet_gemm_workspace(M, N, split_k)) return RENET_ERR_WORKSPACE;
    P6Args pa;
    pa.A = (const __bf16*)Ap; pa.B = (const __bf16*)Bp;
    pa.a_plane = a_plane; pa.b_plane = b_plane;
    pa.tca = lda >> 4; pa.tcb = ldb >> 4;
    pa.alpha_dev = alpha_dev; pa.col_out = col_out;
    SplitArgs& g = pa.out;
    g.A = nullptr; g.B = nullptr; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K;
    g.lda = 0; g.ldb = 0; g.ldc = ldc; g.alpha = alpha; g.beta = beta;
    g.split_k = split_k;
    g.k_tiles_per_split = (max(1, (st_total + split_k - 1) / split_k) + 1) & ~1;     // even: see gemm_p6_kernel
    g.partial = workspace;
    hipStream_t st = (hipStream_t)stream;
    // tile height: 128 rows (two workgroups per CU) unless RENET_P6_TILE=256 (one 8-wave workgroup per CU)
    static int tile_h = 0;
    if (!tile_h) {
        const char* e_ = getenv("RENET_P6_TILE");
        tile_h = (e_ && atoi(e_) == 256) ? 256 : 128;
    }
    const int nbx = (N + BN - 1) / BN, nby = (M + tile_h - 1) / tile_h;
    g.xcd_order = panel_width(tile_order(), nbx, nby, tile_h, K, split_k, tile_h == 256 ? 32 : 64);
    // persistent grid: at most `slots` workgroups (two per CU for the 128-row tile, one for the 256-row tile), a multiple of 8
    // when the items outnumber it (a workgroup then stays on one XCD's share of the tile sequence); RENET_P6_PERSISTENT=1
    // selects it (default: one workgroup per item)
    pa.nbx = nbx; pa.nby = nby; pa.nbz = split_k;
    static int persistent = -1;
    if (persistent < 0) {
        const char* e_ = getenv("RENET_P6_PERSISTENT");
        persistent = (e_ && e_[0] == '1') ? 1 : 0;      // measured: no gain (profiles/r06_a_planes_ablation.md, v6): off by default
    }
    const long items = (long)nbx * nby * split_k;
    const long slots = tile_h == 256 ? 256 : 512;
    if (items > 0x7fffffffL) return RENET_ERR_UNSUPPORTED;
    dim3 grid((unsigned)((persistent && items > slots) ? slots : items), 1, 1);
    int e;
#define RENET_P6_LAUNCH(WMV)                                                   \
    do {                                                                       \
        if (!a_tr && !b_tr) e = launch_p6<false, false, WMV>(pa, grid, st);    \
        else if (!a_tr && b_tr) e = launch_p6<false, true, WMV>(pa, grid, st); \
        else if (a_tr && !b_tr) e = launch_p6<true, false, WMV>(pa, grid, st); \
        else e = launch_p6<true, true, WMV>(pa, grid, st);                     \
    } while (0)
    if (tile_h == 256) RENET_P6_LAUNCH(4);
    else RENET_P6_LAUNCH(2);
#undef RENET_P6_LAUNCH
    if (e != RENET_OK) return e;
    if (split_k > 1) {
        const size_t total = (size_t)M * N;
        const int blocks = (int)min((size_t)2048, (total + 255) / 256);
        RENET_LAUNCH(p6_reduce_kernel, dim3(blocks), dim3(256), 0, st, workspace, split_k, M, N, alpha, alpha_dev, beta,
                     bias, C, ldc, col_out);
        RENET_LAUNCH_CHECK();
    }
    return RENET_OK;
}

}  // extern "C"
