// "Planes" GEMM (round 6; included by gemm_split.hip inside its anonymous namespace): the bf16x6 arithmetic of
// gemm_split_kernel -- every fp32 operand value as three bf16 terms, the six leading term products on
// v_mfma_f32_32x32x16_bf16, fp32 accumulation -- on operands that ARRIVE split: three bf16 planes per matrix in HBM,
// written by the kernel that produces the tensor (softmax-CE: the CE gradient) or by renet_pack_planes (a weight: once per
// optimizer step).  The k-loop then contains no conversion at all: no v_cvt / v_sub, no ds_write -- the resource the
// in-loop split is bound by (a SIMD's conversion VALU work and its partner's MFMAs serialise, the matrix
// pipe idles half of every k-tile).  An operand element is split ONCE per step instead of once per output tile that
// reads it (the logits GEMM re-split `feat` 180 times and the weight 8 times).
//
// PLANE FORMAT "T16" (renet_t16_off in common.h; tools/p6_layout_sim.py is its executable specification): a plane of a matrix
// [Rp, Cp] (multiples of 256) is stored as 16 x 16 tiles of 512 bytes, tile (tr, tc) at ((tr * Cp / 16) + tc) * 256
// elements, row-major inside with two twists that make the LDS reads of BOTH consumer roles conflict free: row i of a tile
// sits at i ^ 4 when tc is odd, and the two 16-byte halves of a row are swapped when (i >> 3) & 1.  A GEMM consumes a
// stored matrix either K-CONTIGUOUS (contraction over its columns: a half-stage is one tile column, 512-byte segments) or
// K-STRIDED (contraction over its rows: a half-stage is one tile row, one contiguous run); both see every 1 KB LDS-DMA
// piece as two whole tiles.  (Row-major planes, the first two versions of this kernel, fed the K-contiguous role with
// 32- and 64-byte row segments: the DMA stream alone then took 340 / 240 us on the logits GEMM against 195 us for K-strided
// operands -- ablation builds, profiles/r06_a_planes_ablation.md.)
//
// KERNEL (P6Cfg<4>; P6Cfg<2> below is the 128-row form): tile 256 x 128, 8 waves (4 x 2, two per SIMD, 64 x 64 outputs each),
// every wave both loads and multiplies.
//   * half-stage = 16 k of the three planes of both operands: 3 x (8 KB + 4 KB) = 36 KB = 36 LDS-DMA pieces
//     (global_load_lds: no VGPRs, no VALU, no ds_write), 5 / 4 per wave; ring of FOUR slots (144 KB of the 160);
//   * fragments are double buffered in REGISTERS: iteration j multiplies half-stage j (read during j - 1) while the
//     ds_read_b128 / ds_read_b64_tr_b16 of half-stage j + 1 and the DMA of half-stage j + 4 (into the slot j held) are in
//     flight; one raw s_barrier per half-stage, counted vmcnt: three half-stages (108 KB per CU) stay in flight;
//   * 24 MFMAs per wave and half-stage = 1536 matrix-pipe cycles per SIMD.
// Epilogue extras: `alpha_dev` (a device scalar folded into alpha: the upstream autograd gradient, no pass over the CE
// gradient), and `col_out`: the LAST logical column of the product goes to a vector instead of C -- with a ones column
// appended to B this is the bias gradient (column sums of A^T) for free, inside the padding of the last column tile.

struct P6Args {
    const __bf16* A;
    const __bf16* B;
    size_t a_plane, b_plane;      // elements between consecutive planes of an operand
    int tca, tcb;                 // tiles per tile row (Cp / 16) of the stored matrices
    int nbx, nby, nbz;            // output tiles in N, in M, k slices: the kernel is PERSISTENT over nbx * nby * nbz work items
    const float* alpha_dev;       // optional device scalar multiplied into alpha
    float* col_out;               // optional: logical column N - 1 is stored to col_out[row] (stride 1), columns < N - 1 to C
    SplitArgs out;                // M, N (logical columns incl. the col_out one), K, C, ldc, alpha, beta, bias, split-K fields
};

// Tile configuration: WM = rows of 64-row waves.  WM = 4: 256 x 128 tile, 8 waves, 4 ring slots (144 KB: ONE workgroup per
// CU); WM = 2: 128 x 128 tile, 4 waves, 3 ring slots (72 KB: TWO workgroups per CU, whose barriers, prologues and C-store
// epilogues cover each other as in gemm_split_kernel, at 4/3 of the DMA bytes per flop).
template <int WM>
struct P6Cfg {
    static constexpr int NW = 2 * WM;                         // waves
    static constexpr int THREADS = 64 * NW;
    static constexpr int APIECES = 2 * WM;                    // 1 KB DMA pieces of one plane's A image (two tiles each)
    static constexpr int AIMG = APIECES * 1024;               // bytes: one plane's A image of a half-stage
    static constexpr int BIMG = 4 * 1024;
    static constexpr int PP = APIECES + 4;                    // pieces per plane
    static constexpr int PIMG = AIMG + BIMG;
    static constexpr int STAGE = 3 * PIMG;                    // 36 864 / 24 576 B
    static constexpr int SLOTS = WM == 4 ? 4 : 3;
    static constexpr size_t LDS = (size_t)STAGE * SLOTS;      // 147 456 / 73 728 B
    static constexpr int NPIECE = 3 * PP;                     // 36 / 24 per half-stage
    static constexpr int PER = (NPIECE + NW - 1) / NW;        // 5 (waves 0-3; 4 for waves 4-7) / 6
    static constexpr bool UNEVEN = (NPIECE % NW) != 0;
};

// Work item L (the order in which a plain grid would have dispatched its workgroups: x fastest, then y, then z) -> tile.
// The arithmetic of tile_of_block (gemm_split.hip) with the grid passed explicitly: the persistent kernel below walks
// L = blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x a multiple of 8 whenever it is smaller than the item count, so a
// workgroup stays on "its" XCD's eighth of the virtual tile sequence).
__device__ __forceinline__ void p6_tile_of_index(int L, int nbx, int nby, int nbz, int xcd_order, int& bx, int& by, int& bz) {
    const int nb = nbx * nby;
    if (!xcd_order) { bz = L / nb; const int r = L - bz * nb; by = r / nbx; bx = r - by * nbx; return; }
    const int total = nb * nbz, per = total >> 3;
    const int v = L < 8 * per ? (L & 7) * per + (L >> 3) : L;
    bz = v / nb;
    const int t = v - bz * nb;
    const int ns = min(nbx, nby), nl = max(nbx, nby);
    const int w = min(ns, xcd_order);
    const int p = t / (w * nl), r = t - p * (w * nl);
    const int wp = min(w, ns - p * w);
    const int l = r / wp, sh = p * w + (r - l * wp);
    if (nby <= nbx) { bx = l; by = sh; }
    else { by = l; bx = sh; }
}

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

struct P6Frags { bf16x8 a[2][3], b[2][3]; };              // [t][plane]

template <bool A_TR, bool B_TR, int WM>
__global__ __launch_bounds__(P6Cfg<WM>::THREADS, 2) void gemm_p6_kernel(P6Args pa) {
    using Cfg = P6Cfg<WM>;
    constexpr int P6_AIMG = Cfg::AIMG, P6_PIMG = Cfg::PIMG, P6_STAGE = Cfg::STAGE, NS = Cfg::SLOTS, PER = Cfg::PER;
    extern __shared__ __attribute__((aligned(16))) char ring6[];
    const SplitArgs& g = pa.out;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hs_total = (g.K + 15) / 16;                 // half-stages of 16 k; k_tiles_per_split counts them and is EVEN
                                                          //   (the parity of a k block is then the parity of its loop index)
    const int n_items = pa.nbx * pa.nby * pa.nbz;
    int m0 = 0, n0 = 0, z = 0, s0 = 0, nh = 0;            // the tile being computed

    // ---- this wave's DMA pieces: id = wave + NW i < NPIECE; plane id / PP; inside a plane pieces 0 .. APIECES-1 = A (two
    // tiles each), then 4 of B.  (WM = 4: waves 0-3 issue 5 pieces per half-stage, waves 4-7 four; WM = 2: six each.)
    const bool five = !Cfg::UNEVEN || wave < Cfg::NPIECE % Cfg::NW;        // this wave owns a piece in the last slot
    const __bf16* src[PER];
    size_t step[PER];
    int dst[PER];
    auto setup_tile = [&]() {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int id = min(wave + Cfg::NW * i, Cfg::NPIECE - 1);
            const int plane = id / Cfg::PP, idl = id % Cfg::PP;
            const int opnd = idl >= Cfg::APIECES ? 1 : 0;
            const int piece = opnd ? idl - Cfg::APIECES : idl;
            const bool tr = opnd ? B_TR : A_TR;
            const __bf16* base = opnd ? pa.B + (size_t)plane * pa.b_plane : pa.A + (size_t)plane * pa.a_plane;
            const size_t tcn = (size_t)(opnd ? pa.tcb : pa.tca);
            const size_t t0 = (size_t)((opnd ? n0 : m0) >> 4) + 2 * piece + (lane >> 5);    // tile along the operand's M / N index
            const size_t tile = tr ? (size_t)s0 * tcn + t0 : t0 * tcn + (size_t)s0;
            src[i] = base + tile * 256 + (lane & 31) * 8;
            step[i] = tr ? tcn * 256 : (size_t)256;
            dst[i] = plane * P6_PIMG + (opnd ? P6_AIMG : 0) + piece * 1024;
        }
    };
    auto issue_piece = [&](char* base, auto ic) {
        constexpr int i = decltype(ic)::value;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[i],
                                         (__attribute__((address_space(3))) void*)(base + dst[i]), 16, 0, 0);
        src[i] += step[i];
    };
    auto issue = [&](int slot) {
        char* base = ring6 + slot * P6_STAGE;
        static_for<0, PER - 1>([&](auto ic) { issue_piece(base, ic); });
        if (five) issue_piece(base, std::integral_constant<int, PER - 1>{});
    };
    // wait until at most `left` of this wave's issue batches (PER or PER - 1 pieces each) are still in flight
    auto wait_left = [&](int left) {
        if constexpr (WM == 4) {
            if (left >= 2) { if (five) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
            else if (left == 1) { if (five) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            static_assert(WM == 4 || (PER == 6 && !Cfg::UNEVEN), "counted waits");
            if (left >= 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };

    // ---- fragment addresses (bytes inside an operand's image)
    const int wm = wave >> 1, wn = wave & 1;               // WM x 2 waves, 64 x 64 outputs each
    int offA[2][2], offB[2][2];                            // [t][u]: K-contiguous: u = parity of the k block; K-strided: its two tr reads
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if constexpr (!A_TR) {
                const int r = wm * 64 + 32 * t + (lane & 31);
                offA[t][u] = (r >> 4) * 512 + (((r & 15) ^ (4 * u)) * 32) + (((lane >> 5) ^ ((r >> 3) & 1)) * 16);
            } else {
                const int sl = lane & 15;
                const int c = wm * 64 + 32 * t + 16 * ((lane >> 4) & 1) + 4 * (sl & 3);
                const int kk = 8 * (lane >> 5) + 4 * u + (sl >> 2);
                offA[t][u] = (c >> 4) * 512 + ((kk ^ (4 * ((c >> 4) & 1))) * 32) + ((((c >> 3) & 1) ^ ((kk >> 3) & 1)) * 16) +
                             ((c >> 2) & 1) * 8;
            }
            if constexpr (!B_TR) {
                const int r = wn * 64 + 32 * t + (lane & 31);
                offB[t][u] = (r >> 4) * 512 + (((r & 15) ^ (4 * u)) * 32) + (((lane >> 5) ^ ((r >> 3) & 1)) * 16);
            } else {
                const int sl = lane & 15;
                const int c = wn * 64 + 32 * t + 16 * ((lane >> 4) & 1) + 4 * (sl & 3);
                const int kk = 8 * (lane >> 5) + 4 * u + (sl >> 2);
                offB[t][u] = (c >> 4) * 512 + ((kk ^ (4 * ((c >> 4) & 1))) * 32) + ((((c >> 3) & 1) ^ ((kk >> 3) & 1)) * 16) +
                             ((c >> 2) & 1) * 8;
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
    auto read_frags = [&](P6Frags& F, int slot, auto parc) {
        constexpr int par = decltype(parc)::value;
        const char* st = ring6 + slot * P6_STAGE;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const char* ia = st + p * P6_PIMG;
                const char* ib = ia + P6_AIMG;
                if constexpr (!A_TR) F.a[t][p] = *reinterpret_cast<const bf16x8*>(ia + offA[t][par]);
                else F.a[t][p] = frag_tr(ia, offA[t]);
                if constexpr (!B_TR) F.b[t][p] = *reinterpret_cast<const bf16x8*>(ib + offB[t][par]);
                else F.b[t][p] = frag_tr(ib, offB[t]);
            }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // the six term pairs, smallest first (the order of mfma_tile: both kernels sum a k-slab's products alike); group q =
    // the four MFMAs of one term pair
    auto mfma_group = [&](const P6Frags& F, auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
        constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[i][PA[q]], F.b[j][PB[q]], acc[i][j], 0, 0, 0);
    };
    // One half-stage of a wave, order pinned: the 12 fragment reads of the NEXT half-stage first (they complete under the
    // MFMAs), then the six MFMA groups with this wave's DMA pieces of half-stage j + 4 between them, ONE piece per gap -- the
    // two waves of a SIMD (w and w + 4) take different gaps (waves 0-3: behind groups 0..4, waves 4-7: behind groups 2..5):
    // issued as a burst right behind the barrier, both partners sat in the vector-memory issue together and the matrix
    // pipe idled (tools/planes_bench.py: 4096^3 640 us with the burst against 534 us without any DMA).
    auto stage_body = [&](const P6Frags& Fcur, P6Frags& Fnext, int next_slot, auto parc, bool do_issue, char* dbase) {
#ifndef RENET_P6_NOREAD               // (probe builds: MFMAs on whatever the registers hold -- no LDS fragment reads in the loop)
        read_frags(Fnext, next_slot, parc);
#endif
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 6>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
#ifndef RENET_P6_NOMFMA               // (probe builds: DMA stream + barriers only)
            mfma_group(Fcur, qc);
#endif
            __builtin_amdgcn_sched_barrier(0);
            if (do_issue) {
                if constexpr (WM == 4) {
                    if (five) { if constexpr (q < 5) issue_piece(dbase, std::integral_constant<int, q>{}); }
                    else { if constexpr (q >= 2) issue_piece(dbase, std::integral_constant<int, q - 2>{}); }
                } else {
                    issue_piece(dbase, std::integral_constant<int, q>{});           // six pieces, six gaps
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // PERSISTENT over the work items (round 6, v6): a workgroup stores a finished tile with asynchronous stores, zeroes its
    // accumulators and immediately issues the DMA prologue of its next tile -- the stores drain and the prologue lands while
    // the co-resident workgroup multiplies.  (One workgroup per tile paid ~16 us of launch + prologue + C-store per tile: 30 %
    // of the logits GEMM, whose K = 600 is only 38 half-stages.)
    P6Frags F0, F1;
    bool have_tile = false;
    int pm0 = 0, pn0 = 0, pz = 0;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        if (have_tile) {
            p6_store_tile(pa, pm0, pn0, pz, wm, wn, lane, acc);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        int bx, by;
        p6_tile_of_index(item, pa.nbx, pa.nby, pa.nbz, g.xcd_order, bx, by, z);
        m0 = by * (64 * WM); n0 = bx * BN;
        s0 = z * g.k_tiles_per_split;
        nh = max(min(hs_total, s0 + g.k_tiles_per_split) - s0, 0);
        have_tile = true; pm0 = m0; pn0 = n0; pz = z;
        if (nh == 0) continue;
        setup_tile();
        const int npre = min(nh, NS);
        for (int j = 0; j < npre; ++j) issue(j);
        wait_left(max(npre - 2, 0));                                      // half-stages 0 and 1 landed (the stores are older)
        __builtin_amdgcn_s_barrier();
        read_frags(F0, 0, std::integral_constant<int, 0>{});
        __builtin_amdgcn_s_waitcnt(0xC07F);                               // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();                                     // slot 0 has been read by every wave
        // iteration j (parity P): DMA of half-stage j + NS into slot j % NS (read during j - 1 / above), fragment reads of
        // j + 1 into the other register set, MFMAs of j; then half-stage j + 2 must have landed
        int slot = 0;                                                     // j % NS
        auto iter = [&](int j, auto pc) {
            constexpr int P = decltype(pc)::value;
#ifdef RENET_P6_NODMA                 // probe builds only (tools/p6_probe.py): the k-loop without its DMA stream
            const bool do_issue = j + NS < nh && j < 1;
#else
            const bool do_issue = j + NS < nh;
#endif
            char* dbase = ring6 + slot * P6_STAGE;
            slot = slot + 1 == NS ? 0 : slot + 1;
            // (the reads are unconditional: after the last half-stage they fetch a slot nobody uses)
            if constexpr (P == 0) stage_body(F0, F1, slot, std::integral_constant<int, 1>{}, do_issue, dbase);
            else stage_body(F1, F0, slot, std::integral_constant<int, 0>{}, do_issue, dbase);
            wait_left(max(min(nh - 1, j + NS) - (j + 2), 0));
            __builtin_amdgcn_s_waitcnt(0xC07F);                           // lgkmcnt(0), visible to the compiler's scoreboard
            __builtin_amdgcn_s_barrier();
        };
        int j = 0;
        for (; j + 1 < nh; j += 2) {
            iter(j, std::integral_constant<int, 0>{});
            iter(j + 1, std::integral_constant<int, 1>{});
        }
        if (j < nh) iter(j, std::integral_constant<int, 0>{});
    }
    if (have_tile) p6_store_tile(pa, pm0, pn0, pz, wm, wn, lane, acc);
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

// fp32 X[R, C] (row stride ldx) -> three bf16 planes in T16 format (x = p1 + p2 + p3, each term RNE of the running
// residual: the split of store_items), padding written as zeros; ones_col >= 0: column `ones_col` (>= C) of rows < R is 1.0
// (the bias-gradient column, see col_out).  A workgroup = one tile row x four adjacent tiles: wave w writes tile tc0 + w
// WHOLE (512 contiguous bytes per plane; lane = (row of the tile, 4-column group)) and the four waves together read 256
// contiguous bytes of each of the 16 source rows.  (Thread-per-row-segment order wrote 8-byte pieces 512 bytes apart: 3.5x
// slower on the 188 MB CE gradient.)
__global__ __launch_bounds__(256) void pack_planes_kernel(const float* __restrict__ X, int R, int C, int ldx, int Rp,
                                                          int Cp, int ones_col, __bf16* __restrict__ P, size_t plane) {
    const int nbc = Cp >> 6;                               // workgroups per tile row
    const int tr = blockIdx.x / nbc;
    const int tc = (blockIdx.x - tr * nbc) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int row = tr * 16 + (lane >> 2), c = tc * 16 + (lane & 3) * 4;
    const bool vec_ok = ((ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
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
    __bf16* dst = P + renet_t16_off(row, c, Cp >> 4);      // 4 consecutive columns = 4 consecutive elements of a tile row half
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

template <bool A_TR, bool B_TR, int WM>
int launch_p6(const P6Args& pa, dim3 grid, hipStream_t st) {
    static bool attr_set = false;      // benign race: the attribute is idempotent
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_p6_kernel<A_TR, B_TR, WM>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)P6Cfg<WM>::LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    RENET_LAUNCH((gemm_p6_kernel<A_TR, B_TR, WM>), grid, dim3(P6Cfg<WM>::THREADS), P6Cfg<WM>::LDS, st, pa);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}
