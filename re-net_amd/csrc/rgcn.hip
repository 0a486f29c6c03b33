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
#include <stdlib.h>

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
    uint32_t x_rowb, w_rowb;    // item kernels: row stride in BYTES of x (fp32: 4 D; bf16: 2 ld) and of the relation table
    const int32_t* row_map;     // layer 1 on the entity table: addend row of output row v = row_map[v] (hub rows; the
                                // item stream carries it inside its flush items); nullptr = v
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


// =================================================================================================
// Item-stream gather (renet_rgcn_gather_items): the row-group kernel above walks CSR rows and pays one
// dependent load chain per row boundary (the epilogue's self-loop addend is fetched only when the row
// is flushed) on top of {row_ptr} -> {indices} -> {rows}: 16+ serialised memory latencies per wave and
// 91-105 VGPRs (4-5 waves per SIMD, a 2048-block launch needs two rounds) -- latency-, not bandwidth-bound.
//
// Here the host planner (graph.plan_gather_items) linearises the light rows (in-degree <= heavy_thresh) into
// ONE item stream: the in-edges of row v as (source row, edge type) followed by a FLUSH item (v, -1), cut
// into groups of <= 64 items (balanced by item count, never straddling the pruned-layer row prefix).  A wave
// takes one group: one coalesced fetch brings all of its items, then the items go through the load pipe
// UNR at a time -- an edge item loads the source row + its relation blocks, a flush item loads the row's
// self-loop addend and norm -- so the addend is just another in-flight load and nothing in the loop depends on
// anything but the item registers.  Chain per wave: {group bounds} -> {items} -> ceil(n/UNR) batches.
// Hub rows (in-degree > heavy_thresh) are not in the stream: the first n_heavy workgroups of the launch reduce
// one each with all their waves (longest work first), prefetching 64 edge indices per coalesced fetch.
// Every branch is wave-uniform; no atomics; results do not depend on the launch geometry.
// =================================================================================================
struct ItemArgs {
    GatherArgs g;
    const int32_t* it_src;      // item stream: source row of an edge item / destination row of a flush item
    const int32_t* it_type;     // edge type (type_s) or -1 for a flush item
    const int32_t* grp_ptr;     // [n_groups + 1] item offsets of the groups
    int n_groups;
};

// Buffer-descriptor loads (raw_buffer_load, hardware bounds check): an access past num_records returns 0 and
// touches no memory.  Every load of the item loop is therefore UNCONDITIONAL -- a skipped item, a lane beyond the
// feature row (lanes 50..63 at D = 200) or a flush item's relation block simply gets a descriptor with
// num_records = 0 / an offset past the row.  This matters more than it looks: with `if (valid) x = *p;` around the
// loads hipcc (ROCm 7.2) branches around each one and drains vmcnt at every merge point, i.e. the "UNR loads in
// flight" of the row-group kernel above were in fact issued and waited for one at a time.
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    // {base[31:0], base[47:32] (stride 0), num_records, dst_sel/format word of gfx9-family raw buffers}
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t voff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float4 buf_load4s(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// Span of a whole-tensor descriptor and the "skip" vector offset.  The hardware compares the offset with
// num_records: whether or not the scalar (row) offset takes part in that comparison, kOob (+ row offset, no 32-bit
// wrap) is out of range and a valid lane's offset (+ row offset) in range as long as the tensor is smaller than
// 2 GiB; the entry points check that.
constexpr uint32_t kBufSpan = 0x80000000u;
constexpr uint32_t kOob = 0x80000000u;
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, uint32_t voff, float4 o) {
    u32x4 v;
    v.x = __float_as_uint(o.x); v.y = __float_as_uint(o.y); v.z = __float_as_uint(o.z); v.w = __float_as_uint(o.w);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, 0, 0);
}

// bf16 STORAGE of the gather operands (BASELINE config 5; MX = 1: relation blocks bf16, MX = 2: source rows too):
// 8-byte loads of 4 bf16 per lane instead of 16-byte loads of 4 floats, widened to fp32 in registers (exact), fp32
// accumulation and fp32 addend / output as before.  At D = 400 the 6.4 KB relation block per edge is the stream that
// bounds the kernel: bf16 blocks halve it.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float4 buf_load4s_bf16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
    return make_float4(bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y));
}
// the WCH float4 of one lane's relation-block slice: fp32 (16 WCH bytes at wo) or bf16 (8 WCH bytes at wo)
template <int WCH, bool B16>
__device__ __forceinline__ void load_wblock(__amdgpu_buffer_rsrc_t rw, uint32_t wo, uint32_t ws, float4 (&w)[WCH]) {
    if constexpr (!B16) {
#pragma unroll
        for (int q = 0; q < WCH; ++q) w[q] = buf_load4s(rw, wo + 16u * q, ws);
    } else if constexpr (WCH == 1) {
        w[0] = buf_load4s_bf16(rw, wo, ws);
    } else {
#pragma unroll
        for (int q = 0; q < WCH / 2; ++q) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)(wo + 16u * q), (int)ws, 0);
            w[2 * q] = make_float4(bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y));
            w[2 * q + 1] = make_float4(bf_lo(v.z), bf_hi(v.z), bf_lo(v.w), bf_hi(v.w));
        }
    }
}

constexpr int kItemFlush = -1;   // it_type of a flush item
constexpr int kItemNop = -2;     // lanes past the end of a group
// it_type <= kItemFlushMap: a flush item whose self-loop addend lives in row (kItemFlushMap - it_type) of the addend
// tensor instead of row it_src (layer 1 on the entity table: addend = (ent_embeds @ W_loop)[entity of the row])
constexpr int kItemFlushMap = -3;

template <int SI, int NCH, bool TR>
__device__ __forceinline__ void row_epilogue(const GatherArgs& g, int row, bool has_ad, float sc, int lane,
                                             const float4 (&acc)[NCH], const float4 (&ad)[NCH]) {
    constexpr int CH = 100 * SI / 4;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(g.out + (size_t)row * (CH * 4), CH * 16);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = lane + 64 * c;
        float4 o = f4_scale(acc[c], sc);
        if (has_ad) o = f4_add(o, f4_mul(ad[c], renet_drop4(g.drop, (uint64_t)row * CH + ch)));
        if (g.relu) {
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        buf_store4(ro, (uint32_t)ch * 16u, o);               // lanes past the row: out of range => dropped
    }
}

// COMPACT (the pruned backward launch, round 4): that launch walks the item stream of ALL rows for the edges whose source
// lies in the row prefix (src < src_limit: 150 k of the 268 k edges of the bench batch), and a skipped edge still
// occupied one of the UNR slots of a batch (no memory traffic, but a third of the loop iterations).  The wave now
// ballots which of its <= 64 items are live (flush items always; edges by their source) and pops the set bits of that
// mask instead of counting 0..n: only live items reach the load pipe; order, and therefore the result, is unchanged.
template <int SI, int NCH, int UNR, bool TR, int MX, bool COMPACT = false>
__device__ __forceinline__ void gather_item_group(const ItemArgs& a, int grp) {
    constexpr bool XB = MX == 2, WB = MX >= 1;
    constexpr int D = 100 * SI;
    constexpr int CH = D / 4;
    constexpr int WCH = SI;
    constexpr uint32_t ROWB = D * 4;                     // bytes of one fp32 feature row (addend, output)
    const uint32_t XROWB = a.g.x_rowb;                   // bytes of one row of x (fp32: ROWB; bf16: 2 * its row stride)
    const uint32_t WROWB = a.g.w_rowb;                   // bytes of one relation's blocks
    const int lane = threadIdx.x & 63;
    const GatherArgs& g = a.g;
    const int i0 = a.grp_ptr[grp];
    const int n = a.grp_ptr[grp + 1] - i0;               // <= 64 by construction of the plan
    int my_src = 0, my_t = kItemNop;
    if (lane < n) { my_src = a.it_src[i0 + lane]; my_t = a.it_type[i0 + lane]; }
    // pin the wait for the item fetch HERE: left to the compiler it becomes an `s_waitcnt vmcnt(0)` at the loop
    // header, which from the second batch on also waits for the previous batch's output stores
    asm volatile("" : "+v"(my_src), "+v"(my_t));
    const float* sp = g.scale ? g.scale : g.x;           // always a readable address
    // whole-tensor descriptors (kernel arguments => provably wave-uniform, no waterfall loops); the row goes into
    // the scalar offset, the lane into the vector offset; kOob in the vector offset = "do not load, return 0"
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(g.x, kBufSpan);
    const __amdgpu_buffer_rsrc_t rad = make_rsrc(g.addend ? g.addend : g.x, kBufSpan);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(g.W, kBufSpan);
    uint32_t xoff[NCH], woff[NCH], aoff[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const uint32_t ch = (uint32_t)(lane + 64 * c);
        xoff[c] = ch < (uint32_t)CH ? ch * (XB ? 8u : 16u) : kOob;
        aoff[c] = ch < (uint32_t)CH ? ch * 16u : kOob;                 // the addend is always fp32
        woff[c] = ch < (uint32_t)CH ? ch * ((WB ? 8u : 16u) * WCH) : kOob;
    }
    float4 acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);

    unsigned long long live = 0ull;
    if constexpr (COMPACT) live = __builtin_amdgcn_ballot_w64(lane < n && (my_t < 0 || my_src < g.src_limit));
    for (int k = 0; COMPACT ? live != 0ull : k < n; k += UNR) {
        float4 xv[UNR][NCH];
        float4 wv[UNR][NCH][WCH];
        float scv[UNR];
        int pick[UNR];                                   // item index of slot u (63 + "nop" when the batch runs short)
        bool have[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if constexpr (COMPACT) {
                have[u] = live != 0ull;
                pick[u] = have[u] ? __builtin_ctzll(live) : 63;
                live &= live - 1ull;                     // (0 stays 0)
            } else {
                have[u] = (k + u) < n;
                pick[u] = min(k + u, 63);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int idx = pick[u];
            const int src = __builtin_amdgcn_readlane(my_src, idx);            // wave-uniform (SGPR)
            const int t = have[u] ? __builtin_amdgcn_readlane(my_t, idx) : kItemNop;       // (n may be 64)
            const bool edge = t >= 0 && src < g.src_limit;
            const bool flush = t == kItemFlush || t <= kItemFlushMap;
            const bool flush_ad = flush && g.addend != nullptr && src < g.addend_rows;
            int tt = t + g.shift;
            if (tt >= g.T) tt -= g.T;
            const int ldrow = t <= kItemFlushMap ? kItemFlushMap - t : src;      // row the x / addend load reads
            const uint32_t xs = edge ? (uint32_t)ldrow * XROWB : flush_ad ? (uint32_t)ldrow * ROWB : 0u;
            const uint32_t ws = edge ? (uint32_t)tt * WROWB : 0u;
            scv[u] = sp[flush ? src : 0];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                // an item that loads nothing gets the out-of-range vector offset (a per-item descriptor with
                // num_records = 0 would do the same in SGPRs, but costs 4 SGPRs per load in flight: > 96 SGPRs
                // and one wave per SIMD less)
                const uint32_t wo = edge ? woff[c] : kOob;
                if constexpr (!XB) {
                    const uint32_t xo = (edge || flush_ad) ? xoff[c] : kOob;
                    xv[u][c] = flush_ad ? buf_load4s(rad, xo, xs) : buf_load4s(rx, xo, xs);
                } else {
                    // bf16 source rows and fp32 addend rows differ in load width: two unconditional loads, the one that
                    // does not apply gets the out-of-range offset (no memory access); exactly one of them is non-zero
                    const float4 xe = buf_load4s_bf16(rx, edge ? xoff[c] : kOob, xs);
                    const float4 xa = buf_load4s(rad, flush_ad ? aoff[c] : kOob, xs);
                    xv[u][c] = f4_add(xe, xa);
                }
                load_wblock<WCH, WB>(rw, wo, ws, wv[u][c]);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int idx = pick[u];
            const int src = __builtin_amdgcn_readlane(my_src, idx);
            const int t = have[u] ? __builtin_amdgcn_readlane(my_t, idx) : kItemNop;
            if (t >= 0) {                                     // (a skipped edge multiplied zeros: harmless)
#pragma unroll
                for (int c = 0; c < NCH; ++c) blockmul<SI, TR>(xv[u][c], wv[u][c], acc[c]);
            } else if (t == kItemFlush || t <= kItemFlushMap) {
                const bool has_ad = g.addend != nullptr && src < g.addend_rows;
                row_epilogue<SI, NCH, TR>(g, src, has_ad, g.scale ? scv[u] : 1.f, lane, acc, xv[u]);
#pragma unroll
                for (int c = 0; c < NCH; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
}

// One workgroup per hub row: 64 edge indices per coalesced fetch, wave w takes entries w, w + WAVES, ... of
// the window UNR at a time (unconditional buffer loads as above); wave 0 prefetches the row's addend;
// fixed-order LDS combine => deterministic.
template <int SI, int NCH, int UNR, bool TR, int WAVES, int MX, bool COMPACT = false>
__device__ __forceinline__ void gather_hub_row(const GatherArgs& a, int v) {
    constexpr bool XB = MX == 2, WB = MX >= 1;
    constexpr int D = 100 * SI;
    constexpr int CH = D / 4;
    constexpr int WCH = SI;
    constexpr uint32_t ROWB = D * 4;
    const uint32_t XROWB = a.x_rowb, WROWB = a.w_rowb;
    __shared__ float4 red[WAVES][CH];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int e0 = a.row_ptr[v], e1 = a.row_ptr[v + 1];
    const bool has_ad = a.addend != nullptr && v < a.addend_rows;
    const int adrow = (has_ad && a.row_map) ? a.row_map[v] : v;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, kBufSpan);
    const __amdgpu_buffer_rsrc_t rad = make_rsrc(a.addend ? a.addend : a.x, kBufSpan);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.W, kBufSpan);
    uint32_t xoff[NCH], woff[NCH];
    float4 adv[NCH];
    float4 acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const uint32_t ch = (uint32_t)(lane + 64 * c);
        xoff[c] = ch < (uint32_t)CH ? ch * (XB ? 8u : 16u) : kOob;
        woff[c] = ch < (uint32_t)CH ? ch * ((WB ? 8u : 16u) * WCH) : kOob;
        acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        adv[c] = buf_load4s(rad, (has_ad && wave == 0 && ch < (uint32_t)CH) ? ch * 16u : kOob,
                            has_ad ? (uint32_t)adrow * ROWB : 0u);
    }
    for (int base = e0; base < e1; base += 64) {
        const int cnt = min(64, e1 - base);
        int my_col = 0x7fffffff, my_t = 0;
        if (lane < cnt) {
            my_col = a.col[base + lane];
            my_t = a.etype[base + lane] + a.shift;
            if (my_t >= a.T) my_t -= a.T;
        }
        // COMPACT (pruned backward): only the window's LIVE edges (source inside the row prefix) are dealt to the waves --
        // lane l's rank among the live lanes is mbcnt(live); wave w takes ranks w, w + WAVES, ...; the lane holding a
        // rank is found with one ballot.  Otherwise: entries w, w + WAVES, ... of the window, skipped ones included.
        const bool lv = COMPACT && lane < cnt && my_col < a.src_limit;
        const unsigned long long live = COMPACT ? __builtin_amdgcn_ballot_w64(lv) : 0ull;
        const int rank = COMPACT ? (int)__builtin_amdgcn_mbcnt_hi((unsigned)(live >> 32),
                                                                 __builtin_amdgcn_mbcnt_lo((unsigned)live, 0u)) : 0;
        const int n_walk = COMPACT ? __builtin_popcountll(live) : cnt;
        for (int k = wave; k < n_walk; k += WAVES * UNR) {
            float4 xv[UNR][NCH];
            float4 wv[UNR][NCH][WCH];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                int kk;
                bool in_walk = (k + u * WAVES) < n_walk;
                if constexpr (COMPACT) {
                    const unsigned long long sel = __builtin_amdgcn_ballot_w64(lv && rank == k + u * WAVES);
                    kk = sel ? __builtin_ctzll(sel) : 63;
                    in_walk = sel != 0ull;
                } else {
                    kk = min(k + u * WAVES, 63);
                }
                const int src = __builtin_amdgcn_readlane(my_col, kk);      // lanes >= cnt hold INT_MAX => skipped
                const int t = __builtin_amdgcn_readlane(my_t, kk);
                const bool ok = in_walk && src < a.src_limit;
                const uint32_t xs = ok ? (uint32_t)src * XROWB : 0u;
                const uint32_t ws = ok ? (uint32_t)t * WROWB : 0u;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if constexpr (XB) xv[u][c] = buf_load4s_bf16(rx, ok ? xoff[c] : kOob, xs);
                    else xv[u][c] = buf_load4s(rx, ok ? xoff[c] : kOob, xs);
                    load_wblock<WCH, WB>(rw, ok ? woff[c] : kOob, ws, wv[u][c]);
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) blockmul<SI, TR>(xv[u][c], wv[u][c], acc[c]);   // zeros when skipped
            }
        }
    }
    if (wave != 0) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (lane + 64 * c < CH) red[wave][lane + 64 * c] = acc[c];
    }
    __syncthreads();
    if (wave == 0) {
        const float sc = a.scale ? a.scale[v] : 1.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = lane + 64 * c;
            if (ch < CH) {
#pragma unroll
                for (int w = 1; w < WAVES; ++w) acc[c] = f4_add(acc[c], red[w][ch]);
            }
        }
        row_epilogue<SI, NCH, TR>(a, v, has_ad, sc, lane, acc, adv);
    }
}

template <int SI, int NCH, int UNR, bool TR, int MX, bool COMPACT = false>
__device__ __forceinline__ void gather_items_body(const ItemArgs& a) {
    if ((int)blockIdx.x < a.g.n_heavy) {
        gather_hub_row<SI, NCH, UNR, TR, kWaves, MX, COMPACT>(a.g, a.g.heavy[blockIdx.x]);
        return;
    }
    const int nb = gridDim.x - a.g.n_heavy;
    const int vb = renet_xcd_block(blockIdx.x - a.g.n_heavy, nb);
    const int grp = vb * kWaves + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // neighbouring rows share an XCD
    if (grp < a.n_groups) gather_item_group<SI, NCH, UNR, TR, MX, COMPACT>(a, grp);
}

// Four entry kernels with distinct names so that a rocprof kernel trace separates the launch classes of a
// training step: forward over the full batch graph (layer 1), forward over the subject-row prefix (layer 2),
// and their backward-wrt-h counterparts (transposed relation blocks).
#define RENET_GATHER_KERNEL(NAME, TRV, COMPACTV)                                                                       \
    template <int SI, int NCH, int UNR, int MX = 0>                                                                    \
    __global__ __launch_bounds__(kThreads) void NAME(ItemArgs a) { gather_items_body<SI, NCH, UNR, TRV, MX, COMPACTV>(a); }
RENET_GATHER_KERNEL(rgcn_gather_fwd_full, false, false)
RENET_GATHER_KERNEL(rgcn_gather_fwd_pruned, false, false)
RENET_GATHER_KERNEL(rgcn_gather_bwdh_full, true, false)
RENET_GATHER_KERNEL(rgcn_gather_bwdh_pruned, true, true)       // the only class with a source limit: compacted item walk
#undef RENET_GATHER_KERNEL

template <int SI, int NCH, int UNR, int MX = 0>
int launch_gather_items(const ItemArgs& a, bool tr, bool pruned, hipStream_t st) {
    int blocks = (a.n_groups + kWaves - 1) / kWaves;
    blocks = max(8, (blocks + 7) & ~7);                    // multiple of 8 for the XCD remap
    const dim3 grid(blocks + a.g.n_heavy), blk(kThreads);  // hub rows first, then the groups, in ONE launch
    if (!tr && !pruned) RENET_LAUNCH((rgcn_gather_fwd_full<SI, NCH, UNR, MX>), grid, blk, 0, st, a);
    else if (!tr) RENET_LAUNCH((rgcn_gather_fwd_pruned<SI, NCH, UNR, MX>), grid, blk, 0, st, a);
    else if (!pruned) RENET_LAUNCH((rgcn_gather_bwdh_full<SI, NCH, UNR, MX>), grid, blk, 0, st, a);
    else RENET_LAUNCH((rgcn_gather_bwdh_pruned<SI, NCH, UNR, MX>), grid, blk, 0, st, a);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

// edges in flight per wave (tuning knob, read once): RENET_GATHER_UNR in {2, 3, 4, 6, 8}; 0 / unset = default
int gather_unr() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("RENET_GATHER_UNR");
        v = e ? atoi(e) : 0;
    }
    return v;
}

// ---- backward prologue ----------------------------------------------------------------------
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4* p) {
    const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store4(float4* p, float4 v) {
    nt_f4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<nt_f4*>(p));
}

__global__ __launch_bounds__(256) void rgcn_bwd_prep_kernel(const float4* __restrict__ g_out,
                                                            const float4* __restrict__ out,
                                                            const float* __restrict__ norm, int relu,
                                                            DropCfg drop, int N, int CH,
                                                            float4* __restrict__ gn,
                                                            float4* __restrict__ g_loop,
                                                            float* __restrict__ bound_part) {
    // bound_part (optional): per-workgroup maxima of |g_loop|, the operand bound of the f16x3 GEMM that consumes it
    __shared__ float red[4];
    float mx = 0.f;
    const size_t total = (size_t)N * CH;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(i / CH);
        // streaming operands (read once / consumed by matrix-bound GEMMs) bypass the caches: what should still be
        // cache resident when this kernel ends is gn, which the bandwidth-bound gather reads next
        float4 g = nt_load4(g_out + i);
        if (relu) {
            const float4 o = nt_load4(out + i);
            g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f;
            g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
        }
        gn[i] = f4_scale(g, norm[v]);
        const float4 gl = f4_mul(g, renet_drop4(drop, i));
        nt_store4(g_loop + i, gl);
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(gl.x), fabsf(gl.y))), fmaxf(fabsf(gl.z), fabsf(gl.w)));
    }
    if (bound_part) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) bound_part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    }
}

// ---- dW: per-chunk partial outer products -----------------------------------------------------
// one wave per chunk (all edges of the chunk have the same relation type); lane = float4 chunk of
// the feature row; SI*4 accumulators per lane-chunk (the si x so block entries).  Row loads are unconditional
// buffer loads (see the item-stream gather): an edge slot past the chunk's end reads nothing and multiplies zeros.
// A workgroup takes kBwdWGroup CONSECUTIVE chunks (chunks are sorted by type) and adds up, through LDS and in chunk
// order, the chunks of one type before anything is written: the partial of a run lands in the slot of the run's first
// chunk -- the group's first chunk or the first chunk of a type -- and the reduce kernel below reads only those slots
// (the hottest relation of a Zipf batch owns > 1000 chunks: 8x fewer dependent rounds on its critical path).
// BIG: x / gmat of 2 GiB and more (renet_rgcn_bwd_w64) -- 64-bit global addressing instead of the 32-bit buffer offsets.
constexpr int kBwdWGroup = 8;
template <int SI, int NCH, bool BIG = false>
__global__ __launch_bounds__(kBwdWGroup * 64) void rgcn_bwd_w_partial_kernel(
    const float* __restrict__ x, const float* __restrict__ gmat, const int32_t* __restrict__ e_src,
    const int32_t* __restrict__ e_dst, const int32_t* __restrict__ chunk_ptr, const int32_t* __restrict__ chunk_type,
    int n_chunks, float4* __restrict__ partial) {
    constexpr int D = 100 * SI;
    constexpr int CH = D / 4;
    constexpr int WROW4 = D * SI / 4;
    constexpr uint32_t ROWB = D * 4;
    constexpr int UNR = (SI == 4) ? 2 : 8;              // edges with both row loads in flight together
    __shared__ float4 red[kBwdWGroup][WROW4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = blockIdx.x * kBwdWGroup + wave;
    const bool live = c < n_chunks;
    const int e0 = live ? chunk_ptr[c] : 0, e1 = live ? chunk_ptr[c + 1] : 0;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, kBufSpan);
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(gmat, kBufSpan);
    uint32_t off[NCH];
#pragma unroll
    for (int q = 0; q < NCH; ++q) off[q] = (uint32_t)(lane + 64 * q) < (uint32_t)CH ? (uint32_t)(lane + 64 * q) * 16u : kOob;
    float4 acc[NCH][SI];
#pragma unroll
    for (int q = 0; q < NCH; ++q)
#pragma unroll
        for (int i = 0; i < SI; ++i) acc[q][i] = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int eb = e0; eb < e1; eb += 64) {
        const int my_e = eb + lane;
        int my_s = 0, my_d = 0;
        if (my_e < e1) { my_s = e_src[my_e]; my_d = e_dst[my_e]; }
        asm volatile("" : "+v"(my_s), "+v"(my_d));          // wait for the indices here, not inside the loop
        const int cnt = min(64, e1 - eb);
        for (int k0 = 0; k0 < cnt; k0 += UNR) {
            float4 xv[UNR][NCH], gv[UNR][NCH];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int kk = min(k0 + u, 63);
                const bool ok = (k0 + u) < cnt;
                const uint32_t so = ok ? (uint32_t)__builtin_amdgcn_readlane(my_s, kk) * ROWB : 0u;
                const uint32_t dof = ok ? (uint32_t)__builtin_amdgcn_readlane(my_d, kk) * ROWB : 0u;
#pragma unroll
                for (int q = 0; q < NCH; ++q) {
                    if constexpr (BIG) {
                        const int ch = lane + 64 * q;
                        const bool on = ok && ch < CH;
                        const size_t sr = (size_t)__builtin_amdgcn_readlane(my_s, kk) * CH + ch;
                        const size_t dr = (size_t)__builtin_amdgcn_readlane(my_d, kk) * CH + ch;
                        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                        xv[u][q] = on ? reinterpret_cast<const float4*>(x)[sr] : z;
                        gv[u][q] = on ? reinterpret_cast<const float4*>(gmat)[dr] : z;
                    } else {
                        xv[u][q] = buf_load4s(rx, ok ? off[q] : kOob, so);
                        gv[u][q] = buf_load4s(rg, ok ? off[q] : kOob, dof);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
#pragma unroll
                for (int q = 0; q < NCH; ++q) {
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
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
        const int ch = lane + 64 * q;
        if (ch < CH) {
#pragma unroll
            for (int i = 0; i < SI; ++i) red[wave][ch * SI + i] = acc[q][i];
        }
    }
    __syncthreads();
    if (!live) return;
    const int ty = chunk_type[c];
    if (wave != 0 && chunk_type[c - 1] == ty) return;         // not the first chunk of its run
    int run = 1;                                              // chunks of this run inside the group
    while (wave + run < kBwdWGroup && c + run < n_chunks && chunk_type[c + run] == ty) ++run;
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
        const int ch = lane + 64 * q;
        if (ch < CH) {
#pragma unroll
            for (int i = 0; i < SI; ++i) {
                float4 r = red[wave][ch * SI + i];
                for (int w = 1; w < run; ++w) r = f4_add(r, red[wave + w][ch * SI + i]);
                partial[(size_t)c * WROW4 + ch * SI + i] = r;
            }
        }
    }
}

// dW[t, :] = sum over the chunks of type t (fixed order => deterministic); grid = (T, ceil(WROW4/64)).
// Relation frequencies are Zipf-like: on the ICEWS18-shaped merged batch the hottest type owns > 1000 of the ~4500
// chunks, and its workgroup is the kernel's critical path (40 us with 4 waves x 4 loads in flight: 80 dependent
// rounds).  16 waves x 4 independent partial-sum loads each walk the chunk range 64 chunks per round; fixed
// association order (per wave, then an LDS tree over the waves) => still deterministic.
constexpr int kRedWaves = 16;
__global__ __launch_bounds__(kRedWaves * 64) void rgcn_bwd_w_reduce_kernel(
    const float4* __restrict__ partial, const int32_t* __restrict__ type_chunk_ptr, int WROW4,
    int T, int shift, float beta, float4* __restrict__ dW) {
    __shared__ float4 red[kRedWaves][64];
    const int t = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int colq = blockIdx.y * 64 + lane;
    const int c0 = type_chunk_ptr[t], c1 = type_chunk_ptr[t + 1];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (colq < WROW4 && c1 > c0) {
        // the slots that hold a run's sum (rgcn_bwd_w_partial_kernel): c0 itself, then every group start inside the type
        if (wave == 0) s = partial[(size_t)c0 * WROW4 + colq];
        const int g0 = c0 / kBwdWGroup + 1;                                   // first group that starts behind c0
        const int g1 = (c1 + kBwdWGroup - 1) / kBwdWGroup;                   // groups starting before c1
        float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, s3 = s1;
        int gq = g0 + wave;
        for (; gq + 3 * kRedWaves < g1; gq += 4 * kRedWaves) {
            const float4 v0 = partial[(size_t)gq * kBwdWGroup * WROW4 + colq];
            const float4 v1 = partial[(size_t)(gq + kRedWaves) * kBwdWGroup * WROW4 + colq];
            const float4 v2 = partial[(size_t)(gq + 2 * kRedWaves) * kBwdWGroup * WROW4 + colq];
            const float4 v3 = partial[(size_t)(gq + 3 * kRedWaves) * kBwdWGroup * WROW4 + colq];
            s = f4_add(s, v0); s1 = f4_add(s1, v1); s2 = f4_add(s2, v2); s3 = f4_add(s3, v3);
        }
        for (; gq < g1; gq += kRedWaves) s = f4_add(s, partial[(size_t)gq * kBwdWGroup * WROW4 + colq]);
        s = f4_add(f4_add(s, s1), f4_add(s2, s3));
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && colq < WROW4) {
        float4 r = red[0][lane];
#pragma unroll
        for (int w = 1; w < kRedWaves; ++w) r = f4_add(r, red[w][lane]);
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

// dst[target[u]] += sum of the rows src[order[k]], k in [seg_ptr[u], seg_ptr[u+1])  -- deterministic (fixed association
// order per segment length), no atomics.  Segment lengths are Zipf-like (most entities own 1-8 rows of a batch, the
// hottest entity / relation hundreds), and both ends are latency problems, not bandwidth problems:
//   * short segments (<= kSegShort rows): ONE WAVE per segment, all its row loads in flight at once (clamped,
//     branch-free) -- 16 segments per 1024-thread workgroup instead of one 256-thread workgroup per 2-row segment;
//   * long segments: the workgroup's 16 waves walk the segment together, 8 independent row loads per wave and round
//     (128 rows in flight), then a fixed-order LDS combine.
// blockIdx.y selects one of two (source, destination) pairs that share the plan (renet_segment_add2).
constexpr int kSegWaves = 16, kSegShort = 16, kSegUnr = 8;
__global__ __launch_bounds__(kSegWaves * 64) void segment_add_kernel(const float4* __restrict__ src0,
                                                                     const float4* __restrict__ src1,
                                                                     const int32_t* __restrict__ order,
                                                                     const int32_t* __restrict__ seg_ptr,
                                                                     const int32_t* __restrict__ seg_target,
                                                                     int U, int CH, float4* __restrict__ dst0,
                                                                     float4* __restrict__ dst1) {
    const float4* __restrict__ src = blockIdx.y ? src1 : src0;
    float4* __restrict__ dst = blockIdx.y ? dst1 : dst0;
    __shared__ float4 red[kSegWaves][128];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // segment j of workgroup b is b + j * gridDim.x: plans are sorted by target id and the hot ids of a Zipf batch are
    // neighbours -- dealt out contiguously, one workgroup would walk all the long segments one after the other
    const int nwg = gridDim.x;
    // ---- phase A: this wave's own segment, if short
    {
        const int u = blockIdx.x + wave * nwg;
        const int k0 = u < U ? seg_ptr[u] : 0, k1 = u < U ? seg_ptr[u + 1] : 0;
        const int len = k1 - k0;
        if (len > 0 && len <= kSegShort) {
            const size_t tgt = (size_t)seg_target[u] * CH;
            for (int ch = lane; ch < CH; ch += 64) {
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int k = k0; k < k1; k += kSegUnr) {
                    float4 v[kSegUnr];
#pragma unroll
                    for (int j = 0; j < kSegUnr; ++j) v[j] = src[(size_t)order[min(k + j, k1 - 1)] * CH + ch];
#pragma unroll
                    for (int j = 0; j < kSegUnr; ++j)
                        if (k + j < k1) s = f4_add(s, v[j]);
                }
                dst[tgt + ch] = f4_add(dst[tgt + ch], s);
            }
        }
    }
    // ---- phase B: the long segments of this workgroup's 16, one after the other, all waves together
    for (int j = 0; j < kSegWaves; ++j) {
        const int u = blockIdx.x + j * nwg;
        if (u >= U) break;
        const int k0 = seg_ptr[u], k1 = seg_ptr[u + 1];                   // workgroup-uniform
        if (k1 - k0 <= kSegShort) continue;
        const size_t tgt = (size_t)seg_target[u] * CH;
        for (int ch = lane; ch < CH; ch += 64) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = k0 + wave * kSegUnr; k < k1; k += kSegWaves * kSegUnr) {
                float4 v[kSegUnr];
#pragma unroll
                for (int q = 0; q < kSegUnr; ++q) v[q] = src[(size_t)order[min(k + q, k1 - 1)] * CH + ch];
#pragma unroll
                for (int q = 0; q < kSegUnr; ++q)
                    if (k + q < k1) s = f4_add(s, v[q]);
            }
            red[wave][ch] = s;
        }
        __syncthreads();
        if (wave == 0) {
            for (int ch = lane; ch < CH; ch += 64) {
                float4 r = red[0][ch];
#pragma unroll
                for (int w = 1; w < kSegWaves; ++w) r = f4_add(r, red[w][ch]);
                dst[tgt + ch] = f4_add(dst[tgt + ch], r);
            }
        }
        __syncthreads();
    }
}

// ---- index composition for the table-addressed first layer --------------------------------------------
__global__ __launch_bounds__(256) void compose_table_items_kernel(const int32_t* __restrict__ row_map,
                                                                  const int32_t* __restrict__ it_src,
                                                                  const int32_t* __restrict__ it_type, int n_items,
                                                                  const int32_t* __restrict__ col,
                                                                  const int32_t* __restrict__ e_src, int E,
                                                                  int32_t* __restrict__ it_src_t,
                                                                  int32_t* __restrict__ it_type_t,
                                                                  int32_t* __restrict__ col_t,
                                                                  int32_t* __restrict__ e_src_t) {
    const int total = n_items + 2 * E;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        if (i < n_items) {
            const int s = it_src[i], t = it_type[i];
            if (t >= 0) { it_src_t[i] = row_map[s]; it_type_t[i] = t; }            // edge item: source -> table row
            else if (t == -1) { it_src_t[i] = s; it_type_t[i] = -3 - row_map[s]; }  // flush: keep the output row, carry
            else { it_src_t[i] = s; it_type_t[i] = t; }                            //        its table row in the type
        } else if (i < n_items + E) {
            col_t[i - n_items] = row_map[col[i - n_items]];
        } else {
            e_src_t[i - n_items - E] = row_map[e_src[i - n_items - E]];
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

// Workgroups of a segmented add over U segments: one per segment while that still fills the chip (a plan over 20
// relations is 20 long segments -- 16 of them dealt to ONE workgroup would be walked one after the other), 16
// segments per workgroup beyond that.
static int seg_grid(int U) { return U <= 512 ? U : max(512, (U + kSegWaves - 1) / kSegWaves); }

int renet_segment_add(const float* src, const int32_t* order, const int32_t* seg_ptr,
                      const int32_t* seg_target, int U, int D, float* dst, void* stream) {
    if (U < 0 || D <= 0 || (D & 3)) return RENET_ERR_BADARG;
    if (U == 0) return RENET_OK;
    if (D > 512) return RENET_ERR_UNSUPPORTED;
    RENET_LAUNCH(segment_add_kernel, dim3(seg_grid(U)), dim3(kSegWaves * 64), 0,
                       (hipStream_t)stream, (const float4*)src, (const float4*)src, order, seg_ptr, seg_target, U, D / 4,
                       (float4*)dst, (float4*)dst);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

int renet_segment_add2(const float* src0, const float* src1, const int32_t* order, const int32_t* seg_ptr,
                       const int32_t* seg_target, int U, int D, float* dst0, float* dst1, void* stream) {
    if (U < 0 || D <= 0 || (D & 3)) return RENET_ERR_BADARG;
    if (U == 0) return RENET_OK;
    if (D > 512) return RENET_ERR_UNSUPPORTED;
    RENET_LAUNCH(segment_add_kernel, dim3(seg_grid(U), 2), dim3(kSegWaves * 64), 0,
                       (hipStream_t)stream, (const float4*)src0, (const float4*)src1, order, seg_ptr, seg_target, U, D / 4,
                       (float4*)dst0, (float4*)dst1);
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

static int gather_items_impl(int mx, int x_ld, int w_ld, const float* x, int x_rows, int D, const int32_t* it_src, const int32_t* it_type,
                             const int32_t* grp_ptr, int n_groups, const int32_t* row_ptr, const int32_t* col,
                             const int32_t* etype, const float* scale, const float* W, int T, int type_shift,
                             int transpose_w, const float* addend, float drop_p, uint64_t seed, int relu,
                             float* out, int N, const int32_t* heavy_rows, int n_heavy, int src_limit,
                             int addend_rows, int pruned, const int32_t* row_map, void* stream) {
    if (!renet_dim_ok(D)) return RENET_ERR_UNSUPPORTED;
    if (n_heavy < 0 || (n_heavy > 0 && !heavy_rows) || n_groups < 0) return RENET_ERR_BADARG;
    if (N < 0 || T <= 0 || type_shift < 0 || type_shift >= T || drop_p < 0.f || drop_p >= 1.f)
        return RENET_ERR_BADARG;
    if (N == 0 || (n_groups == 0 && n_heavy == 0)) return RENET_OK;
    // 32-bit buffer offsets with the skip marker at the span (kBufSpan / kOob): tensors must stay below 2 GiB
    if ((size_t)max(N, x_rows) * D * sizeof(float) >= ((size_t)1 << 31) ||
        (size_t)T * D * (D / 100) * sizeof(float) >= ((size_t)1 << 31))
        return RENET_ERR_UNSUPPORTED;
    ItemArgs a;
    a.g.x = x; a.g.row_ptr = row_ptr; a.g.col = col; a.g.etype = etype; a.g.scale = scale; a.g.W = W;
    a.g.addend = addend; a.g.out = out; a.g.N = N; a.g.T = T; a.g.shift = type_shift; a.g.relu = relu;
    a.g.heavy = heavy_rows; a.g.n_heavy = n_heavy; a.g.heavy_thresh = 0;
    a.g.src_limit = src_limit > 0 ? src_limit : 0x7fffffff;
    a.g.addend_rows = addend_rows > 0 ? addend_rows : 0x7fffffff;
    a.g.row_map = row_map;
    a.g.drop = make_drop(drop_p, seed);
    a.it_src = it_src; a.it_type = it_type; a.grp_ptr = grp_ptr; a.n_groups = n_groups;
    hipStream_t st = (hipStream_t)stream;
    const bool tr = transpose_w != 0, pr = pruned != 0;
    // bf16 storage: mx = 1: W is a bf16 matrix with row stride w_ld elements; mx = 2: x (a table) too, row stride x_ld
    a.g.x_rowb = mx == 2 ? (uint32_t)x_ld * 2u : (uint32_t)D * 4u;
    a.g.w_rowb = mx >= 1 ? (uint32_t)w_ld * 2u : (uint32_t)D * (D / 100) * 4u;
    if (mx < 0 || mx > 2 || (mx >= 1 && w_ld < D * (D / 100)) || (mx == 2 && x_ld < D)) return RENET_ERR_BADARG;
    if (mx == 1) {
        switch (D) {
            case 100: return launch_gather_items<1, 1, 6, 1>(a, tr, pr, st);
            case 200: return launch_gather_items<2, 1, 3, 1>(a, tr, pr, st);
            default: return launch_gather_items<4, 2, 2, 1>(a, tr, pr, st);
        }
    }
    if (mx == 2) {
        switch (D) {
            case 100: return launch_gather_items<1, 1, 6, 2>(a, tr, pr, st);
            case 200: return launch_gather_items<2, 1, 3, 2>(a, tr, pr, st);
            default: return launch_gather_items<4, 2, 2, 2>(a, tr, pr, st);
        }
    }
    const int unr = gather_unr();
    switch (D) {
        case 100:
            if (unr == 8) return launch_gather_items<1, 1, 8>(a, tr, pr, st);
            if (unr == 4) return launch_gather_items<1, 1, 4>(a, tr, pr, st);
            return launch_gather_items<1, 1, 6>(a, tr, pr, st);
        case 200:
            if (unr == 2) return launch_gather_items<2, 1, 2>(a, tr, pr, st);
            if (unr == 4) return launch_gather_items<2, 1, 4>(a, tr, pr, st);
            if (unr == 6) return launch_gather_items<2, 1, 6>(a, tr, pr, st);
            if (unr == 8) return launch_gather_items<2, 1, 8>(a, tr, pr, st);
            return launch_gather_items<2, 1, 3>(a, tr, pr, st);      // 60 VGPRs: 8 waves per SIMD
        default:
            if (unr == 3) return launch_gather_items<4, 2, 3>(a, tr, pr, st);
            if (unr == 4) return launch_gather_items<4, 2, 4>(a, tr, pr, st);
            return launch_gather_items<4, 2, 2>(a, tr, pr, st);
    }
}

int renet_rgcn_gather_items(const float* x, int D, const int32_t* it_src, const int32_t* it_type,
                            const int32_t* grp_ptr, int n_groups, const int32_t* row_ptr, const int32_t* col,
                            const int32_t* etype, const float* scale, const float* W, int T, int type_shift,
                            int transpose_w, const float* addend, float drop_p, uint64_t seed, int relu,
                            float* out, int N, const int32_t* heavy_rows, int n_heavy, int src_limit,
                            int addend_rows, int pruned, void* stream) {
    return gather_items_impl(0, 0, 0, x, N, D, it_src, it_type, grp_ptr, n_groups, row_ptr, col, etype, scale, W, T,
                             type_shift, transpose_w, addend, drop_p, seed, relu, out, N, heavy_rows, n_heavy, src_limit,
                             addend_rows, pruned, nullptr, stream);
}

int renet_rgcn_gather_items_bf16(const float* x, int D, const int32_t* it_src, const int32_t* it_type,
                                 const int32_t* grp_ptr, int n_groups, const int32_t* row_ptr, const int32_t* col,
                                 const int32_t* etype, const float* scale, const void* W_bf16, int w_ld, int T,
                                 int type_shift, int transpose_w, const float* addend, float drop_p, uint64_t seed,
                                 int relu, float* out, int N, const int32_t* heavy_rows, int n_heavy, int src_limit,
                                 int addend_rows, int pruned, void* stream) {
    return gather_items_impl(1, 0, w_ld, x, N, D, it_src, it_type, grp_ptr, n_groups, row_ptr, col, etype, scale,
                             (const float*)W_bf16, T, type_shift, transpose_w, addend, drop_p, seed, relu, out, N,
                             heavy_rows, n_heavy, src_limit, addend_rows, pruned, nullptr, stream);
}

int renet_rgcn_gather_items_table(const float* table, int table_rows, int D, const int32_t* it_src_t,
                                  const int32_t* it_type_t, const int32_t* grp_ptr, int n_groups,
                                  const int32_t* row_ptr, const int32_t* col_t, const int32_t* etype,
                                  const int32_t* row_map, const float* scale, const float* W, int T, int type_shift,
                                  const float* addend_table, float drop_p, uint64_t seed, int relu, float* out, int N,
                                  const int32_t* heavy_rows, int n_heavy, void* stream) {
    if (!row_map || table_rows <= 0) return RENET_ERR_BADARG;
    return gather_items_impl(0, 0, 0, table, table_rows, D, it_src_t, it_type_t, grp_ptr, n_groups, row_ptr, col_t, etype,
                             scale, W, T, type_shift, 0, addend_table, drop_p, seed, relu, out, N, heavy_rows, n_heavy, 0, 0,
                             0, row_map, stream);
}

int renet_rgcn_gather_items_table_bf16(const void* table_bf16, int table_ld, int table_rows, int D,
                                       const int32_t* it_src_t, const int32_t* it_type_t, const int32_t* grp_ptr,
                                       int n_groups, const int32_t* row_ptr, const int32_t* col_t, const int32_t* etype,
                                       const int32_t* row_map, const float* scale, const void* W_bf16, int w_ld, int T,
                                       int type_shift, const float* addend_table, float drop_p, uint64_t seed, int relu,
                                       float* out, int N, const int32_t* heavy_rows, int n_heavy, void* stream) {
    if (!row_map || table_rows <= 0) return RENET_ERR_BADARG;
    return gather_items_impl(2, table_ld, w_ld, (const float*)table_bf16, table_rows, D, it_src_t, it_type_t, grp_ptr,
                             n_groups, row_ptr, col_t, etype, scale, (const float*)W_bf16, T, type_shift, 0, addend_table,
                             drop_p, seed, relu, out, N, heavy_rows, n_heavy, 0, 0, 0, row_map, stream);
}

int renet_compose_table_items(const int32_t* row_map, const int32_t* it_src, const int32_t* it_type, int n_items,
                              const int32_t* col, const int32_t* e_src, int E, int32_t* it_src_t,
                              int32_t* it_type_t, int32_t* col_t, int32_t* e_src_t, void* stream) {
    if (n_items < 0 || E < 0 || !row_map) return RENET_ERR_BADARG;
    const int total = n_items + 2 * E;
    if (total == 0) return RENET_OK;
    RENET_LAUNCH(compose_table_items_kernel, dim3(min(2048, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                 row_map, it_src, it_type, n_items, col, e_src, E, it_src_t, it_type_t, col_t, e_src_t);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

static int bwd_prep_impl(const float* g_out, const float* out, const float* norm, int relu, float drop_p,
                         uint64_t seed, int N, int D, float* gn, float* g_loop, float* bound_part, void* stream) {
    if (N < 0 || D <= 0 || (D & 3) || drop_p < 0.f || drop_p >= 1.f) return RENET_ERR_BADARG;
    if (N == 0) return bound_part ? RENET_ERR_BADARG : RENET_OK;
    const size_t total = (size_t)N * (D / 4);
    int blocks = (int)min((size_t)(bound_part ? 1024 : 2048), (total + 255) / 256);     // = renet_bound_parts(total)
    RENET_LAUNCH(rgcn_bwd_prep_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)g_out, (const float4*)out, norm, relu, make_drop(drop_p, seed), N,
                       D / 4, (float4*)gn, (float4*)g_loop, bound_part);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

int renet_rgcn_bwd_prep(const float* g_out, const float* out, const float* norm, int relu,
                        float drop_p, uint64_t seed, int N, int D, float* gn, float* g_loop,
                        void* stream) {
    return bwd_prep_impl(g_out, out, norm, relu, drop_p, seed, N, D, gn, g_loop, nullptr, stream);
}

int renet_rgcn_bwd_prep_bounds(const float* g_out, const float* out, const float* norm, int relu,
                               float drop_p, uint64_t seed, int N, int D, float* gn, float* g_loop,
                               float* bound_part, void* stream) {
    if (!bound_part) return RENET_ERR_BADARG;
    return bwd_prep_impl(g_out, out, norm, relu, drop_p, seed, N, D, gn, g_loop, bound_part, stream);
}

size_t renet_rgcn_bwd_w_workspace(int n_chunks, int D) {
    return (size_t)max(n_chunks, 0) * (size_t)(D * (D / 100)) * sizeof(float);
}

static int bwd_w_impl(bool big, const float* x, const float* gn, const int32_t* e_src, const int32_t* e_dst,
                      const int32_t* chunk_ptr, const int32_t* chunk_type, int n_chunks,
                      const int32_t* type_chunk_ptr, int T, int type_shift, int D, float* dW, float beta,
                      float* workspace, size_t workspace_bytes, void* stream) {
    if (!renet_dim_ok(D)) return RENET_ERR_UNSUPPORTED;
    if (n_chunks > 0 && !chunk_type) return RENET_ERR_BADARG;
    if (n_chunks < 0 || T <= 0 || type_shift < 0 || type_shift >= T) return RENET_ERR_BADARG;
    if (workspace_bytes < renet_rgcn_bwd_w_workspace(n_chunks, D)) return RENET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int SI = D / 100;
    const int WROW4 = D * SI / 4;
    if (n_chunks > 0) {
        dim3 grid((n_chunks + kBwdWGroup - 1) / kBwdWGroup);
        const float* x4 = x;
        const float* g4 = gn;
        float4* p4 = (float4*)workspace;
        switch (D) {
            case 100:
                if (big) RENET_LAUNCH((rgcn_bwd_w_partial_kernel<1, 1, true>), grid, dim3(kBwdWGroup * 64), 0, st, x4, g4,
                                            e_src, e_dst, chunk_ptr, chunk_type, n_chunks, p4);
                else RENET_LAUNCH((rgcn_bwd_w_partial_kernel<1, 1>), grid, dim3(kBwdWGroup * 64), 0, st, x4, g4,
                                   e_src, e_dst, chunk_ptr, chunk_type, n_chunks, p4);
                break;
            case 200:
                if (big) RENET_LAUNCH((rgcn_bwd_w_partial_kernel<2, 1, true>), grid, dim3(kBwdWGroup * 64), 0, st, x4, g4,
                                            e_src, e_dst, chunk_ptr, chunk_type, n_chunks, p4);
                else RENET_LAUNCH((rgcn_bwd_w_partial_kernel<2, 1>), grid, dim3(kBwdWGroup * 64), 0, st, x4, g4,
                                   e_src, e_dst, chunk_ptr, chunk_type, n_chunks, p4);
                break;
            default:
                if (big) RENET_LAUNCH((rgcn_bwd_w_partial_kernel<4, 2, true>), grid, dim3(kBwdWGroup * 64), 0, st, x4, g4,
                                            e_src, e_dst, chunk_ptr, chunk_type, n_chunks, p4);
                else RENET_LAUNCH((rgcn_bwd_w_partial_kernel<4, 2>), grid, dim3(kBwdWGroup * 64), 0, st, x4, g4,
                                   e_src, e_dst, chunk_ptr, chunk_type, n_chunks, p4);
                break;
        }
        RENET_LAUNCH_CHECK();
    }
    RENET_LAUNCH(rgcn_bwd_w_reduce_kernel, dim3(T, (WROW4 + 63) / 64), dim3(kRedWaves * 64), 0, st,
                       (const float4*)workspace, type_chunk_ptr, WROW4, T, type_shift, beta, (float4*)dW);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

int renet_rgcn_bwd_w(const float* x, const float* gn, const int32_t* e_src, const int32_t* e_dst,
                     const int32_t* chunk_ptr, const int32_t* chunk_type, int n_chunks,
                     const int32_t* type_chunk_ptr, int T, int type_shift, int D, float* dW, float beta,
                     float* workspace, size_t workspace_bytes, void* stream) {
    return bwd_w_impl(false, x, gn, e_src, e_dst, chunk_ptr, chunk_type, n_chunks, type_chunk_ptr, T, type_shift, D, dW,
                      beta, workspace, workspace_bytes, stream);
}

int renet_rgcn_bwd_w64(const float* x, const float* gn, const int32_t* e_src, const int32_t* e_dst,
                       const int32_t* chunk_ptr, const int32_t* chunk_type, int n_chunks,
                       const int32_t* type_chunk_ptr, int T, int type_shift, int D, float* dW, float beta,
                       float* workspace, size_t workspace_bytes, void* stream) {
    return bwd_w_impl(true, x, gn, e_src, e_dst, chunk_ptr, chunk_type, n_chunks, type_chunk_ptr, T, type_shift, D, dW,
                      beta, workspace, workspace_bytes, stream);
}

}  // extern "C"
