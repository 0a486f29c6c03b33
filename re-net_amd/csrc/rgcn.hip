// RGCN block-diagonal gather-SpMM and its backward kernels for gfx950 (MI355X).
//
// Replaces, for RE-Net's RGCNBlockLayer (reference RGCN.py:79-94 + 42-50), the DGL/torch sequence
//   index_select(weight, type) [E, D*si]  ->  bmm (E*100 tiny GEMMs)  ->  fn.sum  ->  h*norm  -> +loop -> act
// with ONE pass: the destination row is the unit of work, the feature dimension lies across the
// lanes (float4 per lane: 50 lanes at D=200), the 1x1 / 2x2 / 4x4 relation block product is
// lane-local, the in-edges of the row are walked serially (rows are short: SURVEY 8, deg<=4 for
// 72-97 % of rows) so no cross-lane reduction and no atomics are needed, and the epilogue
// (norm, self-loop addend with dropout, ReLU) is fused.  HBM-bound integer/gather work: no MFMA.
#include "common.h"

namespace {

constexpr int kWaves = 4;           // waves per workgroup
constexpr int kThreads = 64 * kWaves;

template <int SI, bool TR>
__device__ __forceinline__ void blockmul(const float4 x, const float4* __restrict__ w, float4& acc) {
    if constexpr (SI == 1) {
        const float4 w0 = w[0];
        acc.x = fmaf(x.x, w0.x, acc.x);
        acc.y = fmaf(x.y, w0.y, acc.y);
        acc.z = fmaf(x.z, w0.z, acc.z);
        acc.w = fmaf(x.w, w0.w, acc.w);
    } else if constexpr (SI == 2) {
        const float4 a = w[0], b = w[1];      // block0 = (a.x a.y ; a.z a.w)  block1 = (b.x b.y ; b.z b.w)
        if constexpr (!TR) {
            acc.x = fmaf(x.x, a.x, fmaf(x.y, a.z, acc.x));
            acc.y = fmaf(x.x, a.y, fmaf(x.y, a.w, acc.y));
            acc.z = fmaf(x.z, b.x, fmaf(x.w, b.z, acc.z));
            acc.w = fmaf(x.z, b.y, fmaf(x.w, b.w, acc.w));
        } else {
            acc.x = fmaf(x.x, a.x, fmaf(x.y, a.y, acc.x));
            acc.y = fmaf(x.x, a.z, fmaf(x.y, a.w, acc.y));
            acc.z = fmaf(x.z, b.x, fmaf(x.w, b.y, acc.z));
            acc.w = fmaf(x.z, b.z, fmaf(x.w, b.w, acc.w));
        }
    } else {
        const float4 r0 = w[0], r1 = w[1], r2 = w[2], r3 = w[3];   // rows i = 0..3 of the 4x4 block
        if constexpr (!TR) {
            acc.x = fmaf(x.x, r0.x, fmaf(x.y, r1.x, fmaf(x.z, r2.x, fmaf(x.w, r3.x, acc.x))));
            acc.y = fmaf(x.x, r0.y, fmaf(x.y, r1.y, fmaf(x.z, r2.y, fmaf(x.w, r3.y, acc.y))));
            acc.z = fmaf(x.x, r0.z, fmaf(x.y, r1.z, fmaf(x.z, r2.z, fmaf(x.w, r3.z, acc.z))));
            acc.w = fmaf(x.x, r0.w, fmaf(x.y, r1.w, fmaf(x.z, r2.w, fmaf(x.w, r3.w, acc.w))));
        } else {
            acc.x = fmaf(x.x, r0.x, fmaf(x.y, r0.y, fmaf(x.z, r0.z, fmaf(x.w, r0.w, acc.x))));
            acc.y = fmaf(x.x, r1.x, fmaf(x.y, r1.y, fmaf(x.z, r1.z, fmaf(x.w, r1.w, acc.y))));
            acc.z = fmaf(x.x, r2.x, fmaf(x.y, r2.y, fmaf(x.z, r2.z, fmaf(x.w, r2.w, acc.z))));
            acc.w = fmaf(x.x, r3.x, fmaf(x.y, r3.y, fmaf(x.z, r3.z, fmaf(x.w, r3.w, acc.w))));
        }
    }
}

struct GatherArgs {
    const float* x;
    const int32_t* row_ptr;
    const int32_t* col;
    const int32_t* etype;
    const float* scale;
    const float* W;
    const float* addend;
    float* out;
    const int32_t* heavy;       // rows with in-degree > heavy_thresh, handled by rgcn_gather_heavy_kernel
    int n_heavy, heavy_thresh;
    int src_limit;              // edges whose source row is >= src_limit are skipped (pruned layer-2 backward)
    int addend_rows;            // rows >= addend_rows have no addend
    int N, T, shift, relu;
    DropCfg drop;
};

template <int CH>
__device__ __forceinline__ void gather_epilogue(const GatherArgs& a, int v, int ch, float4 o, float sc) {
    o = f4_scale(o, sc);
    if (a.addend && v < a.addend_rows) {
        float4 ad = reinterpret_cast<const float4*>(a.addend)[(size_t)v * CH + ch];
        ad = f4_mul(ad, renet_drop4(a.drop, (uint64_t)v * CH + ch));
        o = f4_add(o, ad);
    }
    if (a.relu) {
        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
    }
    reinterpret_cast<float4*>(a.out)[(size_t)v * CH + ch] = o;
}

// SI = D/100 (relation block size); NCH = float4 chunks per lane = ceil(D/4/64); UNR = edges whose
// operand loads are in flight together.
//
// A wave owns a GROUP of R = 8 consecutive destination rows: one coalesced fetch brings the group's
// row_ptr slice (and norm), one more the source/type indices of its in-edges (rows are short: the
// group's CSR segment is ~20-40 contiguous edges), so the dependent-load chain is
// {row_ptr} -> {indices} -> {source rows + relation blocks} for 8 rows at once instead of per row.
// Edges are walked in CSR order in batches of UNR with all UNR source-row / weight loads issued
// before the first FMA; a row is flushed (norm, +self-loop addend with dropout, ReLU, store) when the
// walk crosses its row_ptr boundary.  Hub rows (in-degree > heavy_thresh; a Zipf tail of a few hundred
// rows with up to ~300 in-edges) would serialise one wave for the whole launch, so they are skipped
// by the row-group walk and reduced by a whole workgroup each (gather_heavy_row, same launch).  Everything that steers
// control flow is wave-uniform (SGPR).
template <int SI, int NCH, int UNR, bool TR>
__device__ __forceinline__ void gather_heavy_row(const GatherArgs& a, int v);

template <int SI, int NCH, int UNR, bool TR>
__global__ __launch_bounds__(kThreads) void rgcn_gather_kernel(GatherArgs a) {
    constexpr int D = 100 * SI;
    constexpr int CH = D / 4;               // float4 chunks per feature row
    constexpr int WCH = SI;                 // float4 weight loads per chunk
    constexpr int WROW4 = D * SI / 4;       // float4 per relation weight row
    constexpr int R = 8;                    // rows per group
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // the first n_heavy workgroups of the launch each reduce one hub row (longest work items first);
    // the rest walk row groups
    if ((int)blockIdx.x < a.n_heavy) {
        gather_heavy_row<SI, NCH, UNR, TR>(a, a.heavy[blockIdx.x]);
        return;
    }
    const int nb = gridDim.x - a.n_heavy;
    const int vb = renet_xcd_block(blockIdx.x - a.n_heavy, nb);
    const int ngroups = (a.N + R - 1) / R;
    const int gpb = (ngroups + nb - 1) / nb;
    const int g0 = vb * gpb;
    const int g1 = min(ngroups, g0 + gpb);
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(a.x);
    const float4* __restrict__ w4 = reinterpret_cast<const float4*>(a.W);

    for (int grp = g0 + wave; grp < g1; grp += kWaves) {
        const int v0 = grp * R;
        const int nrows = min(R, a.N - v0);
        int my_rp = 0;
        float my_sc = 1.f;
        if (lane <= nrows) my_rp = a.row_ptr[v0 + lane];
        if (a.scale && lane < nrows) my_sc = a.scale[v0 + lane];
        const int e_end = __builtin_amdgcn_readlane(my_rp, nrows);
        int e = __builtin_amdgcn_readlane(my_rp, 0);
        int r = 0;                                       // current row; [row_beg, row_end) its edges
        int row_end = __builtin_amdgcn_readlane(my_rp, 1);
        bool row_heavy = (row_end - e) > a.heavy_thresh;
        float4 acc[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);

        auto flush = [&]() {
            if (!row_heavy) {
                const float sc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_sc), r));
#pragma unroll
                for (int c = 0; c < NCH; ++c)
                    if (lane + 64 * c < CH) gather_epilogue<CH>(a, v0 + r, lane + 64 * c, acc[c], sc);
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            ++r;
            const int beg = row_end;
            row_end = r < nrows ? __builtin_amdgcn_readlane(my_rp, r + 1) : 0x7fffffff;
            row_heavy = r < nrows && (row_end - beg) > a.heavy_thresh;
        };

        int eb = e - 64;                                 // start of the index window held in registers
        int my_col = 0, my_t = 0;
        while (e < e_end) {
            while (e >= row_end) flush();                // row boundary (also empty rows)
            if (row_heavy) { e = row_end; continue; }    // hub row: left to the heavy kernel
            if (e >= eb + 64) {                          // refill the 64-edge index window (coalesced)
                eb = e;
                const int my_e = eb + lane;
                my_col = 0; my_t = 0;
                if (my_e < e_end) {
                    my_col = a.col[my_e];
                    my_t = a.etype[my_e] + a.shift;
                    if (my_t >= a.T) my_t -= a.T;
                }
            }
            const int k0 = e - eb;
            const int cnt = min(64, e_end - eb);
            float4 xv[UNR][NCH];
            float4 wv[UNR][NCH][WCH];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (k0 + u < cnt && __builtin_amdgcn_readlane(my_col, k0 + u) < a.src_limit) {
                    const int src = __builtin_amdgcn_readlane(my_col, k0 + u);   // wave-uniform -> SGPR base
                    const int t = __builtin_amdgcn_readlane(my_t, k0 + u);
                    const float4* xr = x4 + (size_t)src * CH;
                    const float4* wr = w4 + (size_t)t * WROW4;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const int ch = lane + 64 * c;
                        if (ch < CH) {
                            xv[u][c] = xr[ch];
#pragma unroll
                            for (int q = 0; q < WCH; ++q) wv[u][c][q] = wr[ch * WCH + q];
                        }
                    }
                }
            }
            // (a batch may run into the next rows; loads past a hub row's start are simply unused)
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (k0 + u < cnt && e == eb + k0 + u) {
                    while (e >= row_end) flush();
                    if (!row_heavy) {
                        if (__builtin_amdgcn_readlane(my_col, k0 + u) < a.src_limit) {
#pragma unroll
                            for (int c = 0; c < NCH; ++c)
                                if (lane + 64 * c < CH) blockmul<SI, TR>(xv[u][c], wv[u][c], acc[c]);
                        }
                        ++e;
                    }
                }
            }
        }
        while (r < nrows) flush();
    }
}

// One workgroup (4 waves) per hub row: wave w takes in-edges w, w+4, ... in batches of UNR, then a
// fixed-order LDS combine and the same fused epilogue => deterministic.
template <int SI, int NCH, int UNR, bool TR>
__device__ __forceinline__ void gather_heavy_row(const GatherArgs& a, int v) {
    constexpr int D = 100 * SI;
    constexpr int CH = D / 4;
    constexpr int WCH = SI;
    constexpr int WROW4 = D * SI / 4;
    __shared__ float4 red[kWaves][CH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e0 = a.row_ptr[v], e1 = a.row_ptr[v + 1];
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(a.x);
    const float4* __restrict__ w4 = reinterpret_cast<const float4*>(a.W);
    float4 acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = e0 + wave; e < e1; e += kWaves * UNR) {
        float4 xv[UNR][NCH];
        float4 wv[UNR][NCH][WCH];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int ee = e + u * kWaves;
            if (ee < e1 && a.col[ee] < a.src_limit) {
                const int src = a.col[ee];
                int t = a.etype[ee] + a.shift;
                if (t >= a.T) t -= a.T;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int ch = lane + 64 * c;
                    if (ch < CH) {
                        xv[u][c] = x4[(size_t)src * CH + ch];
#pragma unroll
                        for (int q = 0; q < WCH; ++q) wv[u][c][q] = w4[(size_t)t * WROW4 + ch * WCH + q];
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (e + u * kWaves < e1 && a.col[e + u * kWaves] < a.src_limit) {
#pragma unroll
                for (int c = 0; c < NCH; ++c)
                    if (lane + 64 * c < CH) blockmul<SI, TR>(xv[u][c], wv[u][c], acc[c]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (lane + 64 * c < CH) red[wave][lane + 64 * c] = acc[c];
    __syncthreads();
    if (wave == 0) {
        const float sc = a.scale ? a.scale[v] : 1.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = lane + 64 * c;
            if (ch < CH) {
                float4 s = red[0][ch];
#pragma unroll
                for (int w = 1; w < kWaves; ++w) s = f4_add(s, red[w][ch]);
                gather_epilogue<CH>(a, v, ch, s, sc);
            }
        }
    }
}

template <int SI, int NCH, int UNR>
int launch_gather(const GatherArgs& a, bool tr, hipStream_t st) {
    // one 8-row group per wave where possible; multiple of 8 blocks for the XCD remap
    const int ngroups = (a.N + 7) / 8;
    int blocks = (ngroups + kWaves - 1) / kWaves;
    blocks = max(8, min(blocks, 256 * 8));
    blocks = (blocks + 7) & ~7;
    const int grid = blocks + a.n_heavy;         // hub rows first, then the row groups, in ONE launch
    if (tr) RENET_LAUNCH((rgcn_gather_kernel<SI, NCH, UNR, true>), dim3(grid), dim3(kThreads), 0, st, a);
    else RENET_LAUNCH((rgcn_gather_kernel<SI, NCH, UNR, false>), dim3(grid), dim3(kThreads), 0, st, a);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

// ---- backward prologue ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void rgcn_bwd_prep_kernel(const float4* __restrict__ g_out,
                                                            const float4* __restrict__ out,
                                                            const float* __restrict__ norm, int relu,
                                                            DropCfg drop, int N, int CH,
                                                            float4* __restrict__ gn,
                                                            float4* __restrict__ g_loop) {
    const size_t total = (size_t)N * CH;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(i / CH);
        float4 g = g_out[i];
        if (relu) {
            const float4 o = out[i];
            g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f;
            g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
        }
        gn[i] = f4_scale(g, norm[v]);
        g_loop[i] = f4_mul(g, renet_drop4(drop, i));
    }
}

// ---- dW: per-chunk partial outer products -----------------------------------------------------
// one wave per chunk (all edges of the chunk have the same relation type); lane = float4 chunk of
// the feature row; SI*4 accumulators per lane-chunk (the si x so block entries).
template <int SI, int NCH>
__global__ __launch_bounds__(kThreads) void rgcn_bwd_w_partial_kernel(
    const float4* __restrict__ x4, const float4* __restrict__ g4, const int32_t* __restrict__ e_src,
    const int32_t* __restrict__ e_dst, const int32_t* __restrict__ chunk_ptr, int n_chunks,
    float4* __restrict__ partial) {
    constexpr int D = 100 * SI;
    constexpr int CH = D / 4;
    constexpr int WROW4 = D * SI / 4;
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * kWaves + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    const int e0 = chunk_ptr[c], e1 = chunk_ptr[c + 1];
    float4 acc[NCH][SI];
#pragma unroll
    for (int q = 0; q < NCH; ++q)
#pragma unroll
        for (int i = 0; i < SI; ++i) acc[q][i] = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int eb = e0; eb < e1; eb += 64) {
        const int my_e = eb + lane;
        int my_s = 0, my_d = 0;
        if (my_e < e1) { my_s = e_src[my_e]; my_d = e_dst[my_e]; }
        const int cnt = min(64, e1 - eb);
        constexpr int UNR = (SI == 4) ? 2 : 4;              // edges with both row loads in flight together
        for (int k0 = 0; k0 < cnt; k0 += UNR) {
            float4 xv[UNR][NCH], gv[UNR][NCH];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (k0 + u < cnt) {
                    const int s = __builtin_amdgcn_readlane(my_s, k0 + u);
                    const int d = __builtin_amdgcn_readlane(my_d, k0 + u);
#pragma unroll
                    for (int q = 0; q < NCH; ++q) {
                        const int ch = lane + 64 * q;
                        if (ch < CH) {
                            xv[u][q] = x4[(size_t)s * CH + ch];
                            gv[u][q] = g4[(size_t)d * CH + ch];
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (k0 + u < cnt) {
#pragma unroll
                    for (int q = 0; q < NCH; ++q) {
                        if (lane + 64 * q < CH) {
                        if constexpr (SI == 1) {
                            acc[q][0].x = fmaf(xv[u][q].x, gv[u][q].x, acc[q][0].x);
                            acc[q][0].y = fmaf(xv[u][q].y, gv[u][q].y, acc[q][0].y);
                            acc[q][0].z = fmaf(xv[u][q].z, gv[u][q].z, acc[q][0].z);
                            acc[q][0].w = fmaf(xv[u][q].w, gv[u][q].w, acc[q][0].w);
                        } else if constexpr (SI == 2) {
                            // block0: dW[i][j] = x_i g_j (i,j in {0,1}); block1 with elements 2,3
                            acc[q][0].x = fmaf(xv[u][q].x, gv[u][q].x, acc[q][0].x);
                            acc[q][0].y = fmaf(xv[u][q].x, gv[u][q].y, acc[q][0].y);
                            acc[q][0].z = fmaf(xv[u][q].y, gv[u][q].x, acc[q][0].z);
                            acc[q][0].w = fmaf(xv[u][q].y, gv[u][q].y, acc[q][0].w);
                            acc[q][1].x = fmaf(xv[u][q].z, gv[u][q].z, acc[q][1].x);
                            acc[q][1].y = fmaf(xv[u][q].z, gv[u][q].w, acc[q][1].y);
                            acc[q][1].z = fmaf(xv[u][q].w, gv[u][q].z, acc[q][1].z);
                            acc[q][1].w = fmaf(xv[u][q].w, gv[u][q].w, acc[q][1].w);
                        } else {
                            const float xs[4] = {xv[u][q].x, xv[u][q].y, xv[u][q].z, xv[u][q].w};
    #pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                acc[q][i].x = fmaf(xs[i], gv[u][q].x, acc[q][i].x);
                                acc[q][i].y = fmaf(xs[i], gv[u][q].y, acc[q][i].y);
                                acc[q][i].z = fmaf(xs[i], gv[u][q].z, acc[q][i].z);
                                acc[q][i].w = fmaf(xs[i], gv[u][q].w, acc[q][i].w);
                            }
                        }
                    }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
        const int ch = lane + 64 * q;
        if (ch < CH) {
#pragma unroll
            for (int i = 0; i < SI; ++i) partial[(size_t)c * WROW4 + ch * SI + i] = acc[q][i];
        }
    }
}

// dW[t, :] = sum over the chunks of type t (fixed order => deterministic); grid = (T, ceil(WROW4/64)),
// 4 waves split the chunk range, then an LDS tree over the 4 partial sums.
__global__ __launch_bounds__(kThreads) void rgcn_bwd_w_reduce_kernel(
    const float4* __restrict__ partial, const int32_t* __restrict__ type_chunk_ptr, int WROW4,
    int T, int shift, float beta, float4* __restrict__ dW) {
    __shared__ float4 red[kWaves][64];
    const int t = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int colq = blockIdx.y * 64 + lane;
    const int c0 = type_chunk_ptr[t], c1 = type_chunk_ptr[t + 1];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (colq < WROW4) {
        // relation frequencies are Zipf-like: the hottest type owns hundreds of chunks.  Four independent
        // partial-sum loads in flight per wave (fixed association order => still deterministic)
        float4 s1 = s, s2 = s, s3 = s;
        int c = c0 + wave;
        for (; c + 3 * kWaves < c1; c += 4 * kWaves) {
            const float4 v0 = partial[(size_t)c * WROW4 + colq];
            const float4 v1 = partial[(size_t)(c + kWaves) * WROW4 + colq];
            const float4 v2 = partial[(size_t)(c + 2 * kWaves) * WROW4 + colq];
            const float4 v3 = partial[(size_t)(c + 3 * kWaves) * WROW4 + colq];
            s = f4_add(s, v0); s1 = f4_add(s1, v1); s2 = f4_add(s2, v2); s3 = f4_add(s3, v3);
        }
        for (; c < c1; c += kWaves) s = f4_add(s, partial[(size_t)c * WROW4 + colq]);
        s = f4_add(f4_add(s, s1), f4_add(s2, s3));
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && colq < WROW4) {
        float4 r = f4_add(f4_add(red[0][lane], red[1][lane]), f4_add(red[2][lane], red[3][lane]));
        int to = t + shift;
        if (to >= T) to -= T;
        float4* o = dW + (size_t)to * WROW4 + colq;
        if (beta != 0.f) {
            const float4 p = *o;
            r = make_float4(r.x + beta * p.x, r.y + beta * p.y, r.z + beta * p.z, r.w + beta * p.w);
        }
        *o = r;
    }
}

// ---- row gather / segmented add ---------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const float4* __restrict__ table,
                                                          const int32_t* __restrict__ idx, int n, int CH,
                                                          float4* __restrict__ out) {
    const size_t total = (size_t)n * CH;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / CH), c = (int)(i % CH);
        out[i] = table[(size_t)idx[r] * CH + c];
    }
}

__global__ __launch_bounds__(kThreads) void segment_add_kernel(const float4* __restrict__ src,
                                                               const int32_t* __restrict__ order,
                                                               const int32_t* __restrict__ seg_ptr,
                                                               const int32_t* __restrict__ seg_target,
                                                               int U, int CH, float4* __restrict__ dst) {
    // One workgroup per segment: the 4 waves take rows k0+w, k0+w+4, ... (hot entities / relations own
    // hundreds of rows), 4 independent row loads in flight per wave, fixed-order LDS combine.
    __shared__ float4 red[kWaves][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u = blockIdx.x;
    if (u >= U) return;
    const int k0 = seg_ptr[u], k1 = seg_ptr[u + 1];
    const size_t tgt = (size_t)seg_target[u] * CH;
    const bool single = (k1 - k0) <= 1;                  // the common case: no cross-wave combine needed
    for (int ch = lane; ch < CH; ch += 64) {
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
        int k = k0 + wave;
        for (; k + 12 < k1; k += 16) {
            const int r0 = order[k], r1 = order[k + 4], r2 = order[k + 8], r3 = order[k + 12];
            s0 = f4_add(s0, src[(size_t)r0 * CH + ch]);
            s1 = f4_add(s1, src[(size_t)r1 * CH + ch]);
            s2 = f4_add(s2, src[(size_t)r2 * CH + ch]);
            s3 = f4_add(s3, src[(size_t)r3 * CH + ch]);
        }
        for (; k < k1; k += 4) s0 = f4_add(s0, src[(size_t)order[k] * CH + ch]);
        const float4 s = f4_add(f4_add(s0, s1), f4_add(s2, s3));
        if (single) {
            if (wave == 0) dst[tgt + ch] = f4_add(dst[tgt + ch], s);
        } else {
            red[wave][ch] = s;
        }
    }
    if (single) return;
    __syncthreads();
    if (wave == 0) {
        for (int ch = lane; ch < CH; ch += 64) {
            const float4 s = f4_add(f4_add(red[0][ch], red[1][ch]), f4_add(red[2][ch], red[3][ch]));
            dst[tgt + ch] = f4_add(dst[tgt + ch], s);
        }
    }
}

}  // namespace

extern "C" {

int renet_version(void) { return RENET_ABI_VERSION; }

int renet_gather_rows(const float* table, const int32_t* idx, int n, int D, float* out, void* stream) {
    if (n < 0 || D <= 0 || (D & 3)) return RENET_ERR_BADARG;
    if (n == 0) return RENET_OK;
    const int CH = D / 4;
    const size_t total = (size_t)n * CH;
    int blocks = (int)min((size_t)2048, (total + 255) / 256);
    RENET_LAUNCH(gather_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)table, idx, n, CH, (float4*)out);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

int renet_segment_add(const float* src, const int32_t* order, const int32_t* seg_ptr,
                      const int32_t* seg_target, int U, int D, float* dst, void* stream) {
    if (U < 0 || D <= 0 || (D & 3)) return RENET_ERR_BADARG;
    if (U == 0) return RENET_OK;
    if (D > 512) return RENET_ERR_UNSUPPORTED;
    RENET_LAUNCH(segment_add_kernel, dim3(U), dim3(kThreads), 0,
                       (hipStream_t)stream, (const float4*)src, order, seg_ptr, seg_target, U, D / 4,
                       (float4*)dst);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

int renet_rgcn_gather(const float* x, int D, const int32_t* row_ptr, const int32_t* col,
                      const int32_t* etype, const float* scale, const float* W, int T, int type_shift,
                      int transpose_w, const float* addend, float drop_p, uint64_t seed, int relu,
                      float* out, int N, const int32_t* heavy_rows, int n_heavy, int heavy_thresh,
                      int src_limit, int addend_rows, void* stream) {
    if (!renet_dim_ok(D)) return RENET_ERR_UNSUPPORTED;
    if (n_heavy < 0 || (n_heavy > 0 && (!heavy_rows || heavy_thresh < 1))) return RENET_ERR_BADARG;
    if (N < 0 || T <= 0 || type_shift < 0 || type_shift >= T || drop_p < 0.f || drop_p >= 1.f)
        return RENET_ERR_BADARG;
    if (N == 0) return RENET_OK;
    GatherArgs a;
    a.x = x; a.row_ptr = row_ptr; a.col = col; a.etype = etype; a.scale = scale; a.W = W;
    a.addend = addend; a.out = out; a.N = N; a.T = T; a.shift = type_shift; a.relu = relu;
    a.heavy = heavy_rows; a.n_heavy = n_heavy;
    a.src_limit = src_limit > 0 ? src_limit : 0x7fffffff;
    a.addend_rows = addend_rows > 0 ? addend_rows : 0x7fffffff;
    a.heavy_thresh = n_heavy > 0 ? heavy_thresh : 0x7fffffff;
    a.drop = make_drop(drop_p, seed);
    hipStream_t st = (hipStream_t)stream;
    switch (D) {
        case 100: return launch_gather<1, 1, 4>(a, transpose_w != 0, st);
        case 200: return launch_gather<2, 1, 4>(a, transpose_w != 0, st);
        default: return launch_gather<4, 2, 2>(a, transpose_w != 0, st);
    }
}

int renet_rgcn_bwd_prep(const float* g_out, const float* out, const float* norm, int relu,
                        float drop_p, uint64_t seed, int N, int D, float* gn, float* g_loop,
                        void* stream) {
    if (N < 0 || D <= 0 || (D & 3) || drop_p < 0.f || drop_p >= 1.f) return RENET_ERR_BADARG;
    if (N == 0) return RENET_OK;
    const size_t total = (size_t)N * (D / 4);
    int blocks = (int)min((size_t)2048, (total + 255) / 256);
    RENET_LAUNCH(rgcn_bwd_prep_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)g_out, (const float4*)out, norm, relu, make_drop(drop_p, seed), N,
                       D / 4, (float4*)gn, (float4*)g_loop);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

size_t renet_rgcn_bwd_w_workspace(int n_chunks, int D) {
    return (size_t)max(n_chunks, 0) * (size_t)(D * (D / 100)) * sizeof(float);
}

int renet_rgcn_bwd_w(const float* x, const float* gn, const int32_t* e_src, const int32_t* e_dst,
                     const int32_t* chunk_ptr, const int32_t* chunk_type, int n_chunks,
                     const int32_t* type_chunk_ptr, int T, int type_shift, int D, float* dW, float beta,
                     float* workspace, size_t workspace_bytes, void* stream) {
    (void)chunk_type;
    if (!renet_dim_ok(D)) return RENET_ERR_UNSUPPORTED;
    if (n_chunks < 0 || T <= 0 || type_shift < 0 || type_shift >= T) return RENET_ERR_BADARG;
    if (workspace_bytes < renet_rgcn_bwd_w_workspace(n_chunks, D)) return RENET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int SI = D / 100;
    const int WROW4 = D * SI / 4;
    if (n_chunks > 0) {
        dim3 grid((n_chunks + kWaves - 1) / kWaves);
        const float4* x4 = (const float4*)x;
        const float4* g4 = (const float4*)gn;
        float4* p4 = (float4*)workspace;
        switch (D) {
            case 100:
                RENET_LAUNCH((rgcn_bwd_w_partial_kernel<1, 1>), grid, dim3(kThreads), 0, st, x4, g4,
                                   e_src, e_dst, chunk_ptr, n_chunks, p4);
                break;
            case 200:
                RENET_LAUNCH((rgcn_bwd_w_partial_kernel<2, 1>), grid, dim3(kThreads), 0, st, x4, g4,
                                   e_src, e_dst, chunk_ptr, n_chunks, p4);
                break;
            default:
                RENET_LAUNCH((rgcn_bwd_w_partial_kernel<4, 2>), grid, dim3(kThreads), 0, st, x4, g4,
                                   e_src, e_dst, chunk_ptr, n_chunks, p4);
                break;
        }
        RENET_LAUNCH_CHECK();
    }
    RENET_LAUNCH(rgcn_bwd_w_reduce_kernel, dim3(T, (WROW4 + 63) / 64), dim3(kThreads), 0, st,
                       (const float4*)workspace, type_chunk_ptr, WROW4, T, type_shift, beta, (float4*)dW);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

}  // extern "C"
