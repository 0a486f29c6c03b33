// "Planes" GEMM (round 6; included by gemm_split.hip inside its anonymous namespace): the bf16x6 arithmetic of
// gemm_split_kernel -- every fp32 operand value as three bf16 terms, the six leading term products on
// v_mfma_f32_32x32x16_bf16, fp32 accumulation -- on operands that ARRIVE split: three bf16 planes per matrix in HBM,
// written by the kernel that produces the tensor (softmax-CE: the CE gradient; Adam: the weights; renet_pack_planes for
// anything else).  The k-loop then contains no conversion at all: no v_cvt / v_sub, no ds_write -- the resource the
// in-loop split is bound by (a SIMD's conversion VALU work and its partner's MFMAs serialise, the matrix
// pipe idles half of every k-tile).  An operand element is split ONCE per step instead of once per output tile that
// reads it (the logits GEMM re-split `feat` 180 times and the weight 8 times).
//
// Structure: LDS-DMA (global_load_lds: no VGPRs, no VALU, no ds_write) into a ring of FOUR half-stage slots, wave
// specialised like gemm_bf16s_kernel<.., TALL> above; one stored matrix serves the K-contiguous and the K-strided role
// (ds_read_b128 / ds_read_b64_tr_b16).
//   * tile 256 x 128, half-stage = 16 k of THREE planes per operand: 3 x (8 KB + 4 KB) = 36 KB, four slots = 144 KB;
//   * 8 MFMA waves (4 x 2, two per SIMD, 64 x 64 each): 24 MFMAs per wave and half-stage = 1536 matrix-pipe cycles per
//     SIMD, against 36 x 1 KB DMA pieces issued by the 4 loader waves (one per SIMD);
//   * one raw s_barrier per half-stage: at barrier j half-stage j is visible and j - 1 is consumed; the loaders then
//     issue half-stage j + 3 into the slot j - 1 held and wait (counted vmcnt) only for half-stage j + 1 -- two to three
//     half-stages (72-108 KB per CU) stay in flight.  (The first version had two 72 KB k32 slots and waited vmcnt(0) per
//     stage: ablation builds, tools/p6_probe.py, showed the DMA stream alone at 240 us and the MFMA stream alone at 294 us
//     on the logits GEMM, the two together at 394 us -- latency-exposed, profiles/r06_a_planes_ablation.md.)
//   * K-contiguous image of a half-stage: [rows][16 k], 32-byte rows, the two 16-byte chunks of a row swapped when
//     (row >> 3) & 1 (on the DMA's SOURCE address): every ds_read_b128 lane group of MI355X_MICROARCH.md's LDS table
//     ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) then covers all 64 banks exactly once.
//     K-strided image: [16 k][128-column panels], 256-byte rows, sixteen 16-byte chunks XOR-swizzled by 4 * (k & 3) -- the
//     image of the bf16-storage kernel, read with two ds_read_b64_tr_b16 per fragment.
// Epilogue extras: `alpha_dev` (a device scalar folded into alpha: the upstream autograd gradient, no pass over the CE
// gradient), and `col_out`: the LAST logical column of the product goes to a vector instead of C -- with a ones column
// appended to B this is the bias gradient (column sums of A^T) for free, inside the padding of the last column tile.

struct P6Args {
    const __bf16* A;
    const __bf16* B;
    size_t a_plane, b_plane;      // elements between consecutive planes of an operand
    int lda, ldb;                 // row stride (elements) of the stored matrices
    const float* alpha_dev;       // optional device scalar multiplied into alpha
    float* col_out;               // optional: logical column N - 1 is stored to col_out[row] (stride 1), columns < N - 1 to C
    SplitArgs out;                // M, N (logical columns incl. the col_out one), K, C, ldc, alpha, beta, bias, split-K fields
};

constexpr int P6_AIMG = 256 * 32;                      // bytes: one plane's A image of a k16 half-stage
constexpr int P6_BIMG = 128 * 32;
constexpr int P6_PIMG = P6_AIMG + P6_BIMG;             // one plane of a half-stage: A image, then B image (12 KB)
constexpr int P6_STAGE = 3 * P6_PIMG;                  // 36 864 B
constexpr int P6_SLOTS = 4;
constexpr size_t P6_LDS = (size_t)P6_STAGE * P6_SLOTS; // 147 456 B
constexpr int P6_MW = 8, P6_LW = 4;
constexpr int P6_THREADS = 64 * (P6_MW + P6_LW);
constexpr int P6_NPIECE = 3 * (8 + 4);
constexpr int P6_PER = P6_NPIECE / P6_LW;              // 9 DMA pieces per loader wave and half-stage

__device__ __forceinline__ void p6_store_tile(const P6Args& pa, int m0, int n0, int z, int wm, int wn, int lane,
                                              const f32x16 (&acc)[2][2]) {
    const SplitArgs& g = pa.out;
    const bool split = g.split_k > 1;
    const int half = lane >> 5;
    const int n_main = pa.col_out ? g.N - 1 : g.N;
    const bool accumulate = !split && g.beta != 0.f;
    const float alpha = pa.alpha_dev ? g.alpha * pa.alpha_dev[0] : g.alpha;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
            if (col >= g.N) continue;
            const int row_base = m0 + wm * 64 + i * 32 + 4 * half;
            float* base;
            size_t rs;                                   // row stride (elements) of this lane's output column
            float bv = 0.f;
            if (split) { base = g.partial + (size_t)z * g.M * g.N + col; rs = (size_t)g.N; }
            else if (col < n_main) { base = g.C + col; rs = (size_t)g.ldc; bv = g.bias ? g.bias[col] : 0.f; }
            else { base = pa.col_out; rs = 1; }
            float old[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) old[r] = 0.f;
            if (accumulate) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(row_base + (r & 3) + 8 * (r >> 2), g.M - 1);
                    old[r] = base[(size_t)row * rs];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row_base + (r & 3) + 8 * (r >> 2);
                if (row < g.M) {
                    float* p = base + (size_t)row * rs;
                    if (split) *p = acc[i][j][r];
                    else *p = alpha * acc[i][j][r] + bv + (accumulate ? g.beta * old[r] : 0.f);
                }
            }
        }
}

template <bool A_TR, bool B_TR>
__global__ __launch_bounds__(P6_THREADS) void gemm_p6_kernel(P6Args pa) {
    extern __shared__ __attribute__((aligned(16))) char ring6[];
    const SplitArgs& g = pa.out;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by, z;
    tile_of_block(gridDim.x, gridDim.y, g.xcd_order, bx, by, z);
    const int m0 = by * 256, n0 = bx * BN;
    const int hs_total = (g.K + 15) / 16;                 // half-stages of 16 k; k_tiles_per_split counts them
    const int s0 = z * g.k_tiles_per_split;
    const int s1 = min(hs_total, s0 + g.k_tiles_per_split);
    const int nh = max(s1 - s0, 0);

    if (wave >= P6_MW) {
        // ---------------- loader waves: 9 x 1 KB LDS-DMA pieces per half-stage each ----------------
        if (nh == 0) return;
        const int lw = wave - P6_MW;
        const __bf16* src[P6_PER];
        int dst[P6_PER];
#pragma unroll
        for (int i = 0; i < P6_PER; ++i) {
            // slot i of loader lw: plane i / 3; piece id lw + 4 (i % 3) in 0 .. 11 of that plane: A pieces 0..7, B pieces 8..11
            // -- for every lw the plane and the operand (i % 3 == 2: B) of slot i are compile-time constants
            const int plane = i / 3, idl = lw + P6_LW * (i % 3);
            const int opnd = (i % 3) == 2 ? 1 : 0;
            const int piece = opnd ? idl - 8 : idl;
            const bool tr = opnd ? B_TR : A_TR;
            const __bf16* base = opnd ? pa.B + (size_t)plane * pa.b_plane : pa.A + (size_t)plane * pa.a_plane;
            const int ld = opnd ? pa.ldb : pa.lda;
            const int r0 = opnd ? n0 : m0;
            const size_t k0 = (size_t)s0 * 16;
            size_t off;
            int img_off;
            if (!tr) {                                   // image [rows][16 k], 32-byte rows: piece = 32 rows
                const int row = 32 * piece + (lane >> 1);
                const int chunk = (lane & 1) ^ ((row >> 3) & 1);
                off = (size_t)(r0 + row) * ld + k0 + chunk * 8;
                img_off = piece * 1024;
            } else {                                     // image [16 k][cols] in 128-column panels of 4 KB: piece = 4 k x 256 B
                const int panel = piece >> 2, pc = piece & 3;
                const int kk = 4 * pc + (lane >> 4);
                const int log16 = (lane & 15) ^ (4 * (kk & 3));
                off = (k0 + kk) * ld + r0 + panel * 128 + log16 * 8;
                img_off = panel * 4096 + pc * 1024;
            }
            src[i] = base + off;
            dst[i] = plane * P6_PIMG + (opnd ? P6_AIMG : 0) + img_off;
        }
        const size_t a_step = A_TR ? (size_t)16 * pa.lda : (size_t)16;
        const size_t b_step = B_TR ? (size_t)16 * pa.ldb : (size_t)16;
        auto issue_all = [&](int slot) {
            char* base = ring6 + slot * P6_STAGE;
#pragma unroll
            for (int i = 0; i < P6_PER; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[i],
                                                 (__attribute__((address_space(3))) void*)(base + dst[i]), 16, 0, 0);
                src[i] += (i % 3) == 2 ? b_step : a_step;
            }
        };
        static_assert(P6_PER == 9, "counted waits below");
        // wait until at most `left` half-stages issued by this wave are still in flight
        auto wait_left = [&](int left) {
            if (left >= 2) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
            else if (left == 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        const int pre = min(nh, 3);
        for (int j = 0; j < pre; ++j) issue_all(j);
        wait_left(pre - 1);
        __builtin_amdgcn_s_barrier();                                     // barrier 0: half-stage 0 visible
        for (int j = 0; j < nh; ++j) {
#ifdef RENET_P6_NODMA                 // probe builds only (tools/p6_probe.py): the k-loop without its DMA stream
            if (j + 3 < nh && j < 1) issue_all((j + 3) & 3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
            if (j + 3 < nh) issue_all((j + 3) & 3);                       // its slot held half-stage j - 1: consumed before barrier j
            wait_left(min(j + 3, nh - 1) - (j + 1));                      // half-stage j + 1 landed; j + 2, j + 3 stay in flight
#endif
            __builtin_amdgcn_s_barrier();                                 // barrier j + 1: half-stage j + 1 visible, j consumed
        }
        return;
    }

    // ---------------- MFMA waves ----------------
    const int wm = wave >> 1, wn = wave & 1;               // 4 x 2 waves, 64 x 64 outputs each
    int offA[2][2], offB[2][2];                            // [t][u]: K-contiguous uses u = 0 only; K-strided: its two tr reads
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if constexpr (!A_TR) {
                const int row = wm * 64 + 32 * t + (lane & 31);
                offA[t][u] = row * 32 + (((lane >> 5) ^ ((row >> 3) & 1)) * 16);
            } else {
                const int sl = lane & 15;
                const int colf = wm * 64 + 32 * t + 16 * ((lane >> 4) & 1) + 4 * (sl & 3);
                const int col = colf & 127;                  // inside its 128-column panel
                const int kk = 8 * (lane >> 5) + 4 * u + (sl >> 2);
                offA[t][u] = (colf >> 7) * 4096 + kk * 256 + (((col >> 3) ^ (4 * (kk & 3))) * 16) + ((col >> 2) & 1) * 8;
            }
            if constexpr (!B_TR) {
                const int row = wn * 64 + 32 * t + (lane & 31);
                offB[t][u] = row * 32 + (((lane >> 5) ^ ((row >> 3) & 1)) * 16);
            } else {
                const int sl = lane & 15;
                const int col = wn * 64 + 32 * t + 16 * ((lane >> 4) & 1) + 4 * (sl & 3);
                const int kk = 8 * (lane >> 5) + 4 * u + (sl >> 2);
                offB[t][u] = kk * 256 + (((col >> 3) ^ (4 * (kk & 3))) * 16) + ((col >> 2) & 1) * 8;
            }
        }
    auto frag_tr = [&](const char* img, const int (&off)[2]) {
        bf16x8 r;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + off[q]));
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

    if (nh > 0) __builtin_amdgcn_s_barrier();                             // barrier 0
    for (int hs = 0; hs < nh; ++hs) {
        const char* st = ring6 + (hs & 3) * P6_STAGE;
#ifdef RENET_P6_NOMFMA                // probe builds only: DMA stream + barriers, no fragment reads / MFMAs
        if (hs >= 0) { __builtin_amdgcn_s_barrier(); continue; }
#endif
        bf16x8 fa[2][3], fb[2][3];                                        // [t][plane]
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const char* ia = st + p * P6_PIMG;
                const char* ib = ia + P6_AIMG;
                if constexpr (!A_TR) fa[t][p] = *reinterpret_cast<const bf16x8*>(ia + offA[t][0]);
                else fa[t][p] = frag_tr(ia, offA[t]);
                if constexpr (!B_TR) fb[t][p] = *reinterpret_cast<const bf16x8*>(ib + offB[t][0]);
                else fb[t][p] = frag_tr(ib, offB[t]);
            }
        // the six term pairs, smallest first (the order of mfma_tile: both kernels sum a k-slab's products alike)
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
        constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA[q]], fb[j][PB[q]], acc[i][j], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                     // barrier hs + 1
    }
    p6_store_tile(pa, m0, n0, z, wm, wn, lane, acc);
}

// split-K reduction of the planes GEMM: C / col_out = alpha * sum_z partial[z] (+ bias) (+ beta * old)
__global__ __launch_bounds__(256) void p6_reduce_kernel(const float* __restrict__ partial, int split_k, int M, int N,
                                                        float alpha, const float* __restrict__ alpha_dev, float beta,
                                                        const float* __restrict__ bias, float* __restrict__ C, int ldc,
                                                        float* __restrict__ col_out) {
    const size_t total = (size_t)M * N;
    const int n_main = col_out ? N - 1 : N;
    const float a = alpha_dev ? alpha * alpha_dev[0] : alpha;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), n = (int)(i % N);
        float s = 0.f;
        for (int zz = 0; zz < split_k; ++zz) s += partial[(size_t)zz * total + i];
        float* p = n < n_main ? C + (size_t)m * ldc + n : col_out + m;
        float v = a * s + ((bias && n < n_main) ? bias[n] : 0.f);
        if (beta != 0.f) v += beta * (*p);
        *p = v;
    }
}

// fp32 X[R, C] (row stride ldx) -> three bf16 planes [3][Rp][Cp] (x = p1 + p2 + p3, each term RNE of the running residual:
// the split of store_items), padding written as zeros; ones_col >= 0: column `ones_col` (>= C) of rows < R is 1.0 (the
// bias-gradient column, see col_out)
__global__ __launch_bounds__(256) void pack_planes_kernel(const float* __restrict__ X, int R, int C, int ldx, int Rp,
                                                          int Cp, int ones_col, __bf16* __restrict__ P, size_t plane) {
    const size_t total = (size_t)Rp * Cp / 4;
    const bool vec_ok = ((ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / (Cp / 4)), c = (int)(i % (Cp / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < R) {
            const float* x = X + (size_t)row * ldx + c;
            if (c + 3 < C && vec_ok) {
                v = *reinterpret_cast<const float4*>(x);
            } else {
                if (c < C) v.x = x[0];
                if (c + 1 < C) v.y = x[1];
                if (c + 2 < C) v.z = x[2];
                if (c + 3 < C) v.w = x[3];
            }
            if (ones_col >= c && ones_col < c + 4) {
                const int d = ones_col - c;
                if (d == 0) v.x = 1.f; else if (d == 1) v.y = 1.f; else if (d == 2) v.z = 1.f; else v.w = 1.f;
            }
        }
        f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
        __bf16* dst = P + (size_t)row * Cp + c;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const bf16x2 blo = __builtin_convertvector(lo, bf16x2);
            const bf16x2 bhi = __builtin_convertvector(hi, bf16x2);
            *reinterpret_cast<uint2*>(dst + (size_t)p * plane) = pack4(blo, bhi);
            if (p < 2) {
                lo -= __builtin_convertvector(blo, f32x2);
                hi -= __builtin_convertvector(bhi, f32x2);
            }
        }
    }
}

template <bool A_TR, bool B_TR>
int launch_p6(const P6Args& pa, dim3 grid, hipStream_t st) {
    static bool attr_set = false;      // benign race: the attribute is idempotent
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_p6_kernel<A_TR, B_TR>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)P6_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    RENET_LAUNCH((gemm_p6_kernel<A_TR, B_TR>), grid, dim3(P6_THREADS), P6_LDS, st, pa);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}
