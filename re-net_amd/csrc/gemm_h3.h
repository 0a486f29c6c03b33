// Fragment of gemm_split.hip's translation unit (included inside its anonymous namespace; uses its tile helpers).
//
// fp32-accurate GEMM on the f16 matrix cores of gfx950 ("f16x3 split", renet_gemm_f32_h3):
//
//   every fp32 operand value, multiplied by a power-of-two scale s of its TENSOR (so that max |x| s lies in
//   [2^14, 2^15)), is split into TWO binary16 terms   x s = h1 + 2^-11 h2,   h1 = rne16(x s),  h2 = rne16((x s - h1) 2^11)
//   (both subtractions / scalings exact in fp32).  Two 11-bit significands: |x s - h1 - 2^-11 h2| <= 2^-22 |x s| while h1
//   is a normal fp16, i.e. for |x| >= 2^-29 max |x|; below that the ABSOLUTE error stays <= 2^-39 max |x|.  The product
//   a*b is evaluated as
//        a1 b1  +  2^-11 (a1 b2 + a2 b1)
//   on v_mfma_f32_32x32x16_f16 with two fp32 accumulators per output element (the 2^-11 is applied once, in the
//   epilogue; the dropped pair a2 b2 2^-22 is <= 2^-22 |a b|): every product is accurate to ~2^-21 (worst case; the
//   errors are unbiased and average out along k -- a K-long dot product comes out within a small multiple of what fp32
//   products with fp32 accumulation give, tests/test_gpu_parity.py).  f16 x f16 products are exact in fp32.  THREE
//   matrix instructions per fragment pair where the bf16x6 split (24 significant bits, worst case 2^-23) needs six,
//   two LDS planes per operand instead of three: the bf16 split spends 16 of a plane's bits on fp32's exponent range,
//   which a GEMM operand does not need per ELEMENT once the tensor's magnitude is factored out.
//
// The tensor scale comes from a bound on max |x|: renet_maxabs_partials (one pass over an activation, <= 256 partial
// maxima; a weight's are cached per optimizer step by the caller) or any upper bound the producer knows (the CE
// gradient is bounded by its scale factor; up to 1024 partials, so that a producer kernel can emit one per workgroup) -- a bound 2^k too large costs k of the 29 binades, nothing else.  The
// kernel reduces the partial maxima in its prologue (no extra launch, no atomics).
//
// Structure: the two-phase k-loop of gemm_split_kernel / gemm_split_tall_kernel (128 x 128 x 32 or 256 x 128 x 32
// tile, split in registers on the way into LDS, the next tile's global loads spread over the MFMAs).

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct H3Args {
    SplitArgs g;
    const float* maxA;         // nA partial maxima of |A| (or one upper bound)
    const float* maxB;
    int nA, nB;
};

// 2^(15 - e) for bound = m 2^e, m in [0.5, 1): bound * scale in [2^14, 2^15).  (0, denormal: the largest scale.)
__device__ __forceinline__ float h3_scale_of(float bound, float& inv) {
    const uint32_t E = (__float_as_uint(bound) >> 23) & 0xffu;
    int f = 268 - (int)E;                      // exponent field of the scale
    f = f > 253 ? 253 : (f < 1 ? 1 : f);
    inv = __uint_as_float((uint32_t)(254 - f) << 23);
    return __uint_as_float((uint32_t)f << 23);
}

__device__ __forceinline__ float h3_wave_max(const float* __restrict__ part, int n, int lane) {
    float m = 0.f;
    for (int i = lane; i < n; i += 64) m = fmaxf(m, fabsf(part[i]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    return m;
}

// (TileLoaderH and fix_item_h -- the raw-buffer loader these kernels were first written for -- now live in gemm_split.hip:
// the bf16x6 two-phase kernels use them too.)

// registers -> two f16 planes in LDS (same [row][k] image and item order as store_items)
template <bool CONTIG_K, bool EDGE, int NT, int ROWS, int NI>
__device__ __forceinline__ void store_items_h(_Float16* __restrict__ S, int rows, int K, int row0, int k0, int tid,
                                              const float4 (&r)[NI], float scale) {
    constexpr int PLANE_H = ROWS * LDS_ROW;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int row, k;
        item_pos<CONTIG_K, ROWS>(tid + NT * i, row, k);
        float4 v = r[i];
        if constexpr (EDGE) v = fix_item_h(v, rows, K, row0, k0, row, k);
        f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
        lo *= scale;
        hi *= scale;
        const f16x2 l1 = __builtin_convertvector(lo, f16x2);                   // v_cvt_pk_f16_f32 (RNE)
        const f16x2 h1 = __builtin_convertvector(hi, f16x2);
        _Float16* dst = S + row * LDS_ROW + k;
        uint2 u;
        u.x = __builtin_bit_cast(unsigned, l1);
        u.y = __builtin_bit_cast(unsigned, h1);
        *reinterpret_cast<uint2*>(dst) = u;
        lo = (lo - __builtin_convertvector(l1, f32x2)) * 2048.f;              // exact residual, exact scaling
        hi = (hi - __builtin_convertvector(h1, f32x2)) * 2048.f;
        const f16x2 l2 = __builtin_convertvector(lo, f16x2);
        const f16x2 h2 = __builtin_convertvector(hi, f16x2);
        u.x = __builtin_bit_cast(unsigned, l2);
        u.y = __builtin_bit_cast(unsigned, h2);
        *reinterpret_cast<uint2*>(dst + PLANE_H) = u;
    }
}

// One k-tile of a wave: 16 fragment reads and 24 MFMAs, the next tile's global loads spread over them.
// Fragment index = operand + 2 * t + 4 * plane.  Per slab: corr += a2 b1, corr += a1 b2, main += a1 b1.
template <int PLANE_A, int PLANE_B, int NPIECES, class LoadFn>
__device__ __forceinline__ void mfma_tile_ld_h(const _Float16* __restrict__ sA, const _Float16* __restrict__ sB, int arow,
                                               int brow, int ksel, f32x16 (&accm)[2][2], f32x16 (&accc)[2][2],
                                               LoadFn&& load_piece) {
    f16x8 F0[8], F1[8];
    auto read = [&](auto frc, f16x8 (&F)[8], int slab) {
        constexpr int fr = frc.value, op = fr & 1, t = (fr >> 1) & 1, p = fr >> 2;
        const _Float16* base = op ? sB + p * PLANE_B + brow : sA + p * PLANE_A + arow;
        F[fr] = *reinterpret_cast<const f16x8*>(base + t * 32 * LDS_ROW + slab * 16 + ksel);
    };
    constexpr int ORDER[8] = {4, 1, 3, 6, 0, 5, 7, 2};        // in the order the products consume them
    static_for<0, 8>([&](auto n) { read(std::integral_constant<int, ORDER[n.value]>{}, F0, 0); });
    static_for<0, 24>([&](auto gc) {
        constexpr int g = gc.value, w = g % 12, q = w >> 2, i = (w >> 1) & 1, j = w & 1;
        constexpr int ia = 2 * i + (q == 0 ? 4 : 0), ib = 1 + 2 * j + (q == 1 ? 4 : 0);
        if constexpr (g < 12) {
            if constexpr (q == 2) accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F0[ia], F0[ib], accm[i][j], 0, 0, 0);
            else accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F0[ia], F0[ib], accc[i][j], 0, 0, 0);
        } else {
            if constexpr (q == 2) accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F1[ia], F1[ib], accm[i][j], 0, 0, 0);
            else accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F1[ia], F1[ib], accc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g < 8) read(std::integral_constant<int, ORDER[g]>{}, F1, 1);
        if constexpr (g % 3 == 1 && g / 3 < NPIECES) load_piece(std::integral_constant<int, g / 3>{});
        __builtin_amdgcn_sched_barrier(0);
    });
}

template <bool TA, bool TB, bool TALL>
__global__ __launch_bounds__(TALL ? 512 : 256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_h3_kernel(H3Args ha) {
    constexpr int TBM = TALL ? 256 : 128, NT = TALL ? 512 : 256;
    constexpr int NIA = TBM * 8 / NT, NIB = BN * 8 / NT;                 // items per thread and k-tile
    constexpr int PLANE_A = TBM * LDS_ROW, PLANE_B = BN * LDS_ROW;
    __shared__ __attribute__((aligned(16))) _Float16 sA[2 * PLANE_A];
    __shared__ __attribute__((aligned(16))) _Float16 sB[2 * PLANE_B];
    const SplitArgs& g = ha.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bx, by, z;
    tile_of_block(gridDim.x, gridDim.y, g.xcd_order, bx, by, z);
    const int m0 = by * TBM, n0 = bx * BN;
    const int kt0 = z * g.k_tiles_per_split;
    const int kt_total = (g.K + BK - 1) / BK;
    const int kt1 = min(kt_total, kt0 + g.k_tiles_per_split);
    constexpr bool A_CK = !TA;
    constexpr bool B_CK = TB;

    f32x16 accm[2][2], accc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accm[i][j][r] = 0.f; accc[i][j][r] = 0.f; }

    float4 ra[NIA], rb[NIB];
    TileLoaderH<A_CK, NT, TBM, NIA> la;
    TileLoaderH<B_CK, NT, BN, NIB> lb;
    la.init(g.A, g.lda, g.M, g.K, m0, tid);
    lb.init(g.B, g.ldb, g.N, g.K, n0, tid);
    if (kt0 < kt1) {
        la.load(kt0 * BK, ra);
        lb.load(kt0 * BK, rb);
    }
    // tensor scales (every wave reduces the <= 256 partial maxima for itself: no LDS, no barrier)
    float inv_a, inv_b;
    const float sc_a = h3_scale_of(h3_wave_max(ha.maxA, ha.nA, lane), inv_a);
    const float sc_b = h3_scale_of(h3_wave_max(ha.maxB, ha.nB, lane), inv_b);

    const bool a_edge = m0 + TBM > g.M, b_edge = n0 + BN > g.N;
    const int arow = (wm * 64 + (lane & 31)) * LDS_ROW;
    const int brow = (wn * 64 + (lane & 31)) * LDS_ROW;
    const int ksel = (lane >> 5) * 8;
    for (int kt = kt0; kt < kt1; ++kt) {
        __syncthreads();                               // previous tile fully consumed
        TRACE_T(wave, kt - kt0, 0);
        const bool k_edge = (kt + 1) * BK > g.K;
        if (a_edge || k_edge) store_items_h<A_CK, true, NT, TBM, NIA>(sA, g.M, g.K, m0, kt * BK, tid, ra, sc_a);
        else store_items_h<A_CK, false, NT, TBM, NIA>(sA, g.M, g.K, m0, kt * BK, tid, ra, sc_a);
        if (b_edge || k_edge) store_items_h<B_CK, true, NT, BN, NIB>(sB, g.N, g.K, n0, kt * BK, tid, rb, sc_b);
        else store_items_h<B_CK, false, NT, BN, NIB>(sB, g.N, g.K, n0, kt * BK, tid, rb, sc_b);
        TRACE_T(wave, kt - kt0, 1);
        __syncthreads();
        TRACE_T(wave, kt - kt0, 2);
        const int k0n = min(kt + 1, kt1 - 1) * BK;     // next tile (the last step reloads its own: harmless)
        mfma_tile_ld_h<PLANE_A, PLANE_B, NIA + NIB>(sA, sB, arow, brow, ksel, accm, accc, [&](auto ic) {
            constexpr int i = ic.value;
            if constexpr (i < NIA) la.load_item(i, k0n, ra[i]);
            else lb.load_item(i - NIA, k0n, rb[i - NIA]);
        });
        TRACE_T(wave, kt - kt0, 3);
    }
    // main + 2^-11 corr, back to the operands' units (both factors are powers of two: exact)
    const float inv = inv_a * inv_b;
    const bool inv_ok = inv > 0.f && inv < __uint_as_float(0x7f000000u);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = fmaf(accc[i][j][r], 0.00048828125f, accm[i][j][r]);
                accm[i][j][r] = inv_ok ? v * inv : (v * inv_a) * inv_b;
            }
    store_tile(g, m0, n0, z, wm, wn, lane, accm);
}

// partial maxima of |x| over a [rows, cols] matrix with row stride ld: part[b], b < gridDim.x <= 256.
// FLAT (contiguous, 16-byte aligned, element count a multiple of 4): one grid-stride sweep with four independent
// 16-byte loads per lane and round; otherwise row by row.
__device__ __forceinline__ float max4(float m, const float4& v) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}

template <bool FLAT>
__global__ __launch_bounds__(256) void maxabs_partials_kernel(const float* __restrict__ x, int rows, int cols, size_t ld,
                                                              float* __restrict__ part) {
    __shared__ float red[4];
    float m = 0.f;
    if constexpr (FLAT) {
        const size_t n4 = (size_t)rows * cols / 4;
        const float4* __restrict__ p = reinterpret_cast<const float4*>(x);
        const size_t stride = (size_t)gridDim.x * 256;
        size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
        float m1 = 0.f, m2 = 0.f, m3 = 0.f;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            const float4 v0 = p[i], v1 = p[i + stride], v2 = p[i + 2 * stride], v3 = p[i + 3 * stride];
            m = max4(m, v0); m1 = max4(m1, v1); m2 = max4(m2, v2); m3 = max4(m3, v3);
        }
        for (; i < n4; i += stride) m = max4(m, p[i]);
        m = fmaxf(fmaxf(m, m1), fmaxf(m2, m3));
    } else {
        const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
        for (int row = blockIdx.x; row < rows; row += gridDim.x) {
            const float* p = x + (size_t)row * ld;
            if (vec) {
                const int c4 = cols >> 2;
                for (int c = threadIdx.x; c < c4; c += 256) m = max4(m, reinterpret_cast<const float4*>(p)[c]);
                for (int c = (c4 << 2) + threadIdx.x; c < cols; c += 256) m = fmaxf(m, fabsf(p[c]));
            } else {
                for (int c = threadIdx.x; c < cols; c += 256) m = fmaxf(m, fabsf(p[c]));
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}


// The same for SEVERAL contiguous fp32 arrays in one launch (the weights of a model, once per optimizer step):
// blockIdx.y = array, table[y] = {pointer, number of float4, pointer to its partials, number of partials}.
struct MaxabsJob { const float* x; unsigned long long n4; float* part; int nblocks; int pad; };
__global__ __launch_bounds__(256) void maxabs_multi_kernel(const MaxabsJob* __restrict__ jobs) {
    __shared__ float red[4];
    const MaxabsJob j = jobs[blockIdx.y];
    if ((int)blockIdx.x >= j.nblocks) return;
    const float4* __restrict__ p = reinterpret_cast<const float4*>(j.x);
    const size_t stride = (size_t)j.nblocks * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float m = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    for (; i + 3 * stride < j.n4; i += 4 * stride) {
        const float4 v0 = p[i], v1 = p[i + stride], v2 = p[i + 2 * stride], v3 = p[i + 3 * stride];
        m = max4(m, v0); m1 = max4(m1, v1); m2 = max4(m2, v2); m3 = max4(m3, v3);
    }
    for (; i < j.n4; i += stride) m = max4(m, p[i]);
    m = fmaxf(fmaxf(m, m1), fmaxf(m2, m3));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) j.part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
