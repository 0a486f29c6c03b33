// Batch-graph builder ON THE DEVICE for the merged training batch (graph.build_batch_both; reference
// utils.py:209-244 get_sorted_s_r_embed_rgcn + utils.py:115-131 make_subgraph + dgl.batch, both directions of
// train.py:136-137 in one batch): everything the host builder (graph.py + host_builder.cpp) derives from a batch of
// quadruple indices -- the length sort, the packed sequence layout, the per-(direction, timestamp) node sets, the
// node-induced edges, the CSR by destination with relation-sorted rows, the relation-bucketed edge list and its
// <= 64-edge chunks (full and restricted to the row prefix), hub rows, the gather item stream and its wave groups, and
// the four segmented-add plans -- as kernels over data that is RESIDENT in HBM (quadruples, history index, graph
// store).  The host uploads 4 KB of indices and reads back ~200 bytes of counts; nothing is synchronised in between:
// every stage launches over a capacity and guards on device-side counts.  Output = the arrays of graph.HostBatch,
// bit for bit (tests/test_gpu_builder.py compares every array with the host builder's).
// Integer / index work: rocPRIM radix sorts and scans + small hand-written kernels; HBM-bound, no MFMA.
#include <cstring>
#include "common.h"
#include <rocprim/rocprim.hpp>

namespace {

constexpr int BB_MAXQ = 4096;          // sequences per batch (2 B)
constexpr int BB_MAXL = 32;            // history steps per sequence

// ---- block-wide exclusive scan (1024 threads) ----------------------------------------------------------------
__device__ __forceinline__ int block_excl_scan_1024(int v, int* total, int* wsum /* [16] LDS */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    __syncthreads();                                   // wsum may still be read by the previous call
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int s = wsum[w];
        if (w < wave) base += s;
        tot += s;
    }
    *total = tot;
    return base + inc - v;
}

struct Store {
    const int32_t *q_s, *q_r, *q_o;
    const int32_t* h_first[2];
    const int32_t* h_count[2];
    const int32_t* snap_t[2];
    const int32_t* snap_ptr[2];
    const int32_t* nbr_o[2];
    const int32_t* times;
    const int32_t* trip_ptr;
    const int32_t *trip_s, *trip_r, *trip_o;
    const int32_t* glob_times;
    int T, n_glob, num_ent, num_rels;
};

// ---- stage A: length sort + per-sequence arrays (ONE workgroup) --------------------------------------------------
// sequences q in [0, 2B): q < B = subject side of quadruple idx[q] (entity s, history role 0, relation row r),
// q >= B = object side (entity o, role 1, relation row R + r).  Stable sort by descending history length.
__global__ __launch_bounds__(1024) void bb_seq_kernel(Store st, const int32_t* __restrict__ idx, int B, int seq_len,
                                                      int32_t* __restrict__ perm, int32_t* __restrict__ seq_first,
                                                      int32_t* __restrict__ seq_len_s, int32_t* __restrict__ seq_start,
                                                      int32_t* __restrict__ s_sorted, int32_t* __restrict__ r_sorted,
                                                      int32_t* __restrict__ rel_label, int32_t* __restrict__ ent_label,
                                                      int32_t* __restrict__ step_off, int32_t* __restrict__ counts) {
    __shared__ int lens[BB_MAXQ];
    __shared__ int pos_of[BB_MAXQ];
    __shared__ int wsum[16];
    __shared__ int hist[BB_MAXL + 2];
    const int Q = 2 * B;
    for (int q = threadIdx.x; q < BB_MAXQ; q += 1024) {
        int len = 0;
        if (q < Q) {
            const int role = q >= B, qi = idx[q - role * B];
            len = min(st.h_count[role][qi], seq_len);          // (the index already holds <= history_len snapshots)
        }
        lens[q] = q < Q ? len : -1;
    }
    if (threadIdx.x < BB_MAXL + 2) hist[threadIdx.x] = 0;
    __syncthreads();
    // stable counting sort, longest first: value v from BB_MAXL down to 0, members in index order
    int base = 0;
    for (int v = BB_MAXL; v >= 0; --v) {
        int mine[4], cnt = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { mine[u] = lens[4 * threadIdx.x + u] == v; cnt += mine[u]; }
        int tot;
        int off = block_excl_scan_1024(cnt, &tot, wsum);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (mine[u]) pos_of[4 * threadIdx.x + u] = base + off++;
        if (threadIdx.x == 0) hist[v] = tot;
        base += tot;
    }
    __syncthreads();
    // per sorted position
    for (int q = threadIdx.x; q < Q; q += 1024) {
        const int p = pos_of[q];
        const int role = q >= B, qi = idx[q - role * B];
        const int len = lens[q];
        perm[p] = q;
        seq_len_s[p] = len;
        // the newest `len` snapshots of the window (preprocess.HistoryIndex.take with max_len)
        seq_first[p] = st.h_first[role][qi] + (st.h_count[role][qi] - len);
        const int s = st.q_s[qi], r = st.q_r[qi], o = st.q_o[qi];
        s_sorted[p] = role ? o : s;
        r_sorted[p] = r + (role ? st.num_rels : 0);
        rel_label[p] = r;
        ent_label[p] = role ? s : o;
    }
    __syncthreads();
    // nnz, L, S, step offsets (batch size of step j = #sequences longer than j), sequence-major step starts
    if (threadIdx.x == 0) {
        int nnz = 0, S = 0, L = 0;
        for (int v = 1; v <= BB_MAXL; ++v) { nnz += hist[v]; S += v * hist[v]; if (hist[v]) L = v; }
        counts[RENET_BB_NNZ] = nnz; counts[RENET_BB_S] = S; counts[RENET_BB_L] = L;
        int longer = nnz, off = 0;                       // longer = #sequences with len > j
        for (int j = 0; j <= BB_MAXL; ++j) {
            step_off[j] = off;
            off += longer;
            longer -= hist[j + 1 <= BB_MAXL ? j + 1 : BB_MAXL + 1];
        }
    }
    __syncthreads();
    // seq_start = exclusive scan of the sorted lengths (sequence-major step index of every sequence's first step)
    {
        int v[4], cnt = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = 4 * threadIdx.x + u;
            // sorted length at position p: recover from the histogram (positions are grouped by length, descending)
            int acc = 0, len = 0;
            for (int vv = BB_MAXL; vv >= 1; --vv) { if (p < acc + hist[vv]) { len = vv; break; } acc += hist[vv]; }
            v[u] = p < Q ? len : 0;
            cnt += v[u];
        }
        int tot;
        int off = block_excl_scan_1024(cnt, &tot, wsum);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = 4 * threadIdx.x + u;
            if (p < Q) seq_start[p] = off;
            off += v[u];
        }
    }
}

// ---- stage B: steps -------------------------------------------------------------------------------------------
__device__ __forceinline__ int lower_bound_i32(const int32_t* a, int n, int v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// thread per (sorted sequence i, step j): packed row p = step_off[j] + i, sequence-major k = seq_start[i] + j
__global__ __launch_bounds__(256) void bb_steps_kernel(Store st, int B, const int32_t* __restrict__ perm,
                                                       const int32_t* __restrict__ seq_first,
                                                       const int32_t* __restrict__ seq_len_s,
                                                       const int32_t* __restrict__ seq_start,
                                                       const int32_t* __restrict__ s_sorted,
                                                       const int32_t* __restrict__ r_sorted,
                                                       const int32_t* __restrict__ step_off,
                                                       const int32_t* __restrict__ counts,
                                                       int32_t* __restrict__ step_snap, int32_t* __restrict__ step_dense,
                                                       int32_t* __restrict__ step_packed, int32_t* __restrict__ slot_used,
                                                       int32_t* __restrict__ row_seq, int32_t* __restrict__ row_ent,
                                                       int32_t* __restrict__ row_rel, int32_t* __restrict__ glob_row,
                                                       int32_t* __restrict__ err) {
    const int nnz = counts[RENET_BB_NNZ];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t / BB_MAXL, j = t - i * BB_MAXL;
    if (i >= nnz || j >= seq_len_s[i]) return;
    const int role = perm[i] >= B;
    const int snap = seq_first[i] + j;
    const int k = seq_start[i] + j, p = step_off[j] + i;
    const int tt = st.snap_t[role][snap];
    const int tidx = lower_bound_i32(st.times, st.T, tt);
    if (tidx >= st.T || st.times[tidx] != tt) { atomicOr(err, RENET_BB_ERR_TIME); return; }
    const int dense = role * st.T + tidx;
    step_snap[k] = snap | (role << 30);
    step_dense[k] = dense;
    step_packed[k] = p;
    slot_used[dense] = 1;
    row_seq[p] = i;
    row_ent[p] = s_sorted[i];
    row_rel[p] = r_sorted[i];
    const int gi = lower_bound_i32(st.glob_times, st.n_glob, tt);
    if (gi >= st.n_glob || st.glob_times[gi] != tt) { atomicOr(err, RENET_BB_ERR_GLOB); return; }
    glob_row[p] = gi;
}

// ---- stage C: slots (ONE workgroup): compact the used (direction, timestamp) pairs in (direction, time) order ----
__global__ __launch_bounds__(1024) void bb_slots_kernel(Store st, const int32_t* __restrict__ slot_used,
                                                        int32_t* __restrict__ slot_of_dense, int32_t* __restrict__ slot_ti,
                                                        int32_t* __restrict__ slot_group, int32_t* __restrict__ fact_off,
                                                        int32_t* __restrict__ counts) {
    __shared__ int wsum[16];
    __shared__ int s_tb;
    const int n = 2 * st.T;
    int base = 0;
    for (int c0 = 0; c0 < n; c0 += 1024) {
        const int d = c0 + threadIdx.x;
        const int used = d < n ? slot_used[d] : 0;
        int tot;
        const int off = block_excl_scan_1024(used, &tot, wsum);
        if (used) {
            const int c = base + off;
            slot_of_dense[d] = c;
            slot_ti[c] = d % st.T;
            slot_group[c] = d / st.T;
        }
        base += tot;
    }
    if (threadIdx.x == 0) { s_tb = base; counts[RENET_BB_TB] = base; }
    __syncthreads();
    const int Tb = s_tb;
    base = 0;
    for (int c0 = 0; c0 < Tb; c0 += 1024) {              // fact offsets of the slots' timestamps
        const int c = c0 + threadIdx.x;
        int nf = 0;
        if (c < Tb) { const int ti = slot_ti[c]; nf = st.trip_ptr[ti + 1] - st.trip_ptr[ti]; }
        int tot;
        const int off = block_excl_scan_1024(nf, &tot, wsum);
        if (c < Tb) fact_off[c] = base + off;
        base += tot;
    }
    if (threadIdx.x == 0) { fact_off[Tb] = base; counts[RENET_BB_FACTS] = base; }
}

// ---- stage D: node marking: byte table [slot][entity]: bit 0 = in the node set, bit 1 = a subject (row prefix) ----
__device__ __forceinline__ void mark_byte(uint32_t* table, size_t key, uint32_t bits) {
    atomicOr(&table[key >> 2], bits << (8 * (key & 3)));
}

// one WAVE per step: lane-strided over the snapshot's neighbours
__global__ __launch_bounds__(256) void bb_mark_kernel(Store st, const int32_t* __restrict__ counts,
                                                      const int32_t* __restrict__ step_snap,
                                                      const int32_t* __restrict__ step_dense,
                                                      const int32_t* __restrict__ step_packed,
                                                      const int32_t* __restrict__ row_ent,
                                                      const int32_t* __restrict__ slot_of_dense,
                                                      uint32_t* __restrict__ table) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= counts[RENET_BB_S]) return;
    const int lane = threadIdx.x & 63;
    const int sv = step_snap[k], role = sv >> 30, snap = sv & 0x3FFFFFFF;
    const size_t base = (size_t)slot_of_dense[step_dense[k]] * st.num_ent;
    if (lane == 0) mark_byte(table, base + row_ent[step_packed[k]], 3u);
    const int b = st.snap_ptr[role][snap], e = st.snap_ptr[role][snap + 1];
    for (int n = b + lane; n < e; n += 64) mark_byte(table, base + st.nbr_o[role][n], 1u);
}

// ---- stage E: numbering: rows of the subject keys first (in key order), then the other keys (in key order) ----
constexpr int NUM_TILE = 4096;
__global__ __launch_bounds__(256) void bb_tile_count_kernel(const uint8_t* __restrict__ table, size_t entries_cap,
                                                            const int32_t* __restrict__ counts, int num_ent,
                                                            int2* __restrict__ tile_cnt) {
    __shared__ int ra[4], rb[4];
    const size_t entries = (size_t)counts[RENET_BB_TB] * num_ent;
    const size_t i0 = (size_t)blockIdx.x * NUM_TILE;
    int a = 0, b = 0;
    for (int u = threadIdx.x; u < NUM_TILE; u += 256) {
        const size_t i = i0 + u;
        if (i < entries && i < entries_cap) { const uint8_t v = table[i]; a += (v & 2) != 0; b += v == 1; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if ((threadIdx.x & 63) == 0) { ra[threadIdx.x >> 6] = a; rb[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = make_int2(ra[0] + ra[1] + ra[2] + ra[3], rb[0] + rb[1] + rb[2] + rb[3]);
}

__global__ __launch_bounds__(1024) void bb_tile_scan_kernel(int2* __restrict__ tile_cnt, int n_tiles,
                                                            int32_t* __restrict__ counts, int cap_nodes,
                                                            int32_t* __restrict__ err) {
    __shared__ int wsum[16];
    int base_a = 0, base_b = 0;
    for (int c0 = 0; c0 < n_tiles; c0 += 1024) {
        const int c = c0 + threadIdx.x;
        const int2 v = c < n_tiles ? tile_cnt[c] : make_int2(0, 0);
        int ta, tb;
        const int oa = block_excl_scan_1024(v.x, &ta, wsum);
        const int ob = block_excl_scan_1024(v.y, &tb, wsum);
        if (c < n_tiles) tile_cnt[c] = make_int2(base_a + oa, base_b + ob);
        base_a += ta; base_b += tb;
    }
    if (threadIdx.x == 0) {
        const bool over = base_a + base_b > cap_nodes;
        if (over) atomicOr(err, RENET_BB_ERR_NODES);      // every later stage then sees an EMPTY graph (no OOB access)
        counts[RENET_BB_NA] = over ? 0 : base_a;
        counts[RENET_BB_N] = over ? 0 : base_a + base_b;
    }
}

__global__ __launch_bounds__(256) void bb_number_kernel(const uint8_t* __restrict__ table,
                                                        const int2* __restrict__ tile_off,
                                                        const int32_t* __restrict__ counts, int num_ent, int cap_nodes,
                                                        int32_t* __restrict__ new_id, int32_t* __restrict__ node_ent,
                                                        int32_t* __restrict__ node_slot) {
    __shared__ int wa[4], wb[4];
    const size_t entries = (size_t)counts[RENET_BB_TB] * num_ent;
    const int nA = counts[RENET_BB_NA];
    if (counts[RENET_BB_ERR] & RENET_BB_ERR_NODES) return;
    (void)cap_nodes;
    const size_t i0 = (size_t)blockIdx.x * NUM_TILE;
    if (i0 >= entries) return;
    const int2 toff = tile_off[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int run_a = toff.x, run_b = toff.y;
    // 16 rounds of 256 consecutive entries: order inside the tile = entry order
    for (int r = 0; r < NUM_TILE / 256; ++r) {
        const size_t i = i0 + (size_t)r * 256 + threadIdx.x;
        uint8_t v = 0;
        if (i < entries) v = table[i];
        const int fa = (v & 2) != 0, fb = v == 1;
        int ia = fa, ib = fb;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int ta = __shfl_up(ia, o), tb = __shfl_up(ib, o);
            if (lane >= o) { ia += ta; ib += tb; }
        }
        __syncthreads();
        if (lane == 63) { wa[wave] = ia; wb[wave] = ib; }
        __syncthreads();
        int pa = 0, pb = 0, ta = 0, tb = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) { pa += wa[w]; pb += wb[w]; }
            ta += wa[w]; tb += wb[w];
        }
        if (v) {
            const int id = fa ? run_a + pa + ia - 1 : nA + run_b + pb + ib - 1;
            new_id[i] = id;
            node_ent[id] = (int)(i % (size_t)num_ent);
            node_slot[id] = (int)(i / (size_t)num_ent);
        }
        run_a += ta; run_b += tb;
    }
}

// subject row of every step, in packed order
__global__ __launch_bounds__(256) void bb_subj_row_kernel(int num_ent, const int32_t* __restrict__ counts,
                                                          const int32_t* __restrict__ step_dense,
                                                          const int32_t* __restrict__ step_packed,
                                                          const int32_t* __restrict__ row_ent,
                                                          const int32_t* __restrict__ slot_of_dense,
                                                          const int32_t* __restrict__ new_id,
                                                          int32_t* __restrict__ subj_row) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= counts[RENET_BB_S]) return;
    const int p = step_packed[k];
    subj_row[p] = new_id[(size_t)slot_of_dense[step_dense[k]] * num_ent + row_ent[p]];
}

// ---- stage F: node-induced edges ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bb_fact_flag_kernel(Store st, const int32_t* __restrict__ counts,
                                                           const int32_t* __restrict__ fact_off,
                                                           const int32_t* __restrict__ slot_ti,
                                                           const uint8_t* __restrict__ table, int cap_facts,
                                                           int32_t* __restrict__ flag, int32_t* __restrict__ fslot) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= cap_facts) return;
    int keep = 0, c = 0;
    if (f < counts[RENET_BB_FACTS]) {
        const int Tb = counts[RENET_BB_TB];
        int lo = 0, hi = Tb;                              // last slot with fact_off <= f
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (fact_off[mid] <= f) lo = mid; else hi = mid; }
        c = lo;
        const int j = st.trip_ptr[slot_ti[c]] + (f - fact_off[c]);
        const size_t base = (size_t)c * st.num_ent;
        keep = (table[base + st.trip_s[j]] != 0) && (table[base + st.trip_o[j]] != 0);
    }
    flag[f] = keep;
    fslot[f] = c;
}

// both directions of every kept fact (utils.py:74-76): edge e < E2: ls -> lo with type r; e >= E2: lo -> ls with
// type r + R; the object-side member graphs (group 1) store type_o = (type_s + R) mod 2R (model.py:78).
// Also: sort keys, in-degree and relation histograms.
__global__ __launch_bounds__(256) void bb_edges_kernel(Store st, const int32_t* __restrict__ counts,
                                                       const int32_t* __restrict__ fact_off,
                                                       const int32_t* __restrict__ slot_ti,
                                                       const int32_t* __restrict__ slot_group,
                                                       const int32_t* __restrict__ flag, const int32_t* __restrict__ pos,
                                                       const int32_t* __restrict__ fslot,
                                                       const int32_t* __restrict__ new_id, int cap_facts, int cap_edges,
                                                       int32_t* __restrict__ half_src, int32_t* __restrict__ half_dst,
                                                       int32_t* __restrict__ half_et, int32_t* __restrict__ err) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= cap_facts || f >= counts[RENET_BB_FACTS] || !flag[f]) return;
    const int m = pos[f];
    if (m >= counts[RENET_BB_E2]) return;                  // (E2 was zeroed on overflow / error)
    (void)err; (void)cap_edges;
    const int c = fslot[f];
    const int j = st.trip_ptr[slot_ti[c]] + (f - fact_off[c]);
    const size_t base = (size_t)c * st.num_ent;
    half_src[m] = new_id[base + st.trip_s[j]];
    half_dst[m] = new_id[base + st.trip_o[j]];
    int t = st.trip_r[j];
    if (slot_group[c]) t += st.num_rels;                  // type_o of the forward edge
    half_et[m] = t;
}

__global__ __launch_bounds__(256) void bb_expand_kernel(const int32_t* __restrict__ counts, int num_rels, int cap_edges,
                                                        int key_bits, const int32_t* __restrict__ half_src,
                                                        const int32_t* __restrict__ half_dst,
                                                        const int32_t* __restrict__ half_et,
                                                        int32_t* __restrict__ src, int32_t* __restrict__ dst,
                                                        int32_t* __restrict__ et, uint32_t* __restrict__ key_dt,
                                                        uint32_t* __restrict__ key_t, uint32_t* __restrict__ key_t2,
                                                        int32_t* __restrict__ iota, int32_t* __restrict__ deg,
                                                        int32_t* __restrict__ tc, int32_t* __restrict__ tc2) {
    // relation frequencies are Zipf-like (the hottest type owns a third of the edges): global atomics on the 2R-bin
    // histograms serialise (1.3 ms of a 2.2 ms build); workgroup-local LDS histograms, flushed once, instead
    __shared__ int h1[1024], h2[1024];
    const int T2 = 2 * num_rels;
    for (int i = threadIdx.x; i < T2; i += blockDim.x) { h1[i] = 0; h2[i] = 0; }
    __syncthreads();
    const int E2 = counts[RENET_BB_E2], E = 2 * E2, nA = counts[RENET_BB_NA];
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < cap_edges; e += gridDim.x * blockDim.x) {
        iota[e] = e;
        if (e >= E) {                                     // sentinels: sorted behind every valid key
            key_dt[e] = 1u << key_bits;
            key_t[e] = (uint32_t)T2;
            key_t2[e] = (uint32_t)T2;
            continue;
        }
        const int m = e < E2 ? e : e - E2;
        int s = half_src[m], d = half_dst[m], t = half_et[m];
        if (e >= E2) { const int tmp = s; s = d; d = tmp; t = t + num_rels >= T2 ? t + num_rels - T2 : t + num_rels; }
        src[e] = s; dst[e] = d; et[e] = t;
        key_dt[e] = (uint32_t)d * (uint32_t)T2 + (uint32_t)t;
        key_t[e] = (uint32_t)t;
        key_t2[e] = d < nA ? (uint32_t)t : (uint32_t)T2;
        atomicAdd(&deg[d], 1);
        atomicAdd(&h1[t], 1);
        if (d < nA) atomicAdd(&h2[t], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T2; i += blockDim.x) {
        if (h1[i]) atomicAdd(&tc[i], h1[i]);
        if (h2[i]) atomicAdd(&tc2[i], h2[i]);
    }
}

__global__ void bb_set_e2_kernel(const int32_t* __restrict__ flag, const int32_t* __restrict__ pos,
                                 int32_t* __restrict__ counts, int cap_facts, int cap_edges) {
    const int F = min(counts[RENET_BB_FACTS], cap_facts);
    int e2 = F > 0 ? pos[F - 1] + flag[F - 1] : 0;
    if (counts[RENET_BB_ERR] != 0) e2 = 0;                 // node overflow / bad timestamp: no edges (new_id is not valid)
    if (2 * e2 > cap_edges) { atomicOr(&counts[RENET_BB_ERR], RENET_BB_ERR_EDGES); e2 = 0; }
    counts[RENET_BB_E2] = e2;
    counts[RENET_BB_E] = 2 * e2;
}

// CSR columns / types from the (dst, type)-sorted order; relation-bucketed lists from the type-sorted orders
__global__ __launch_bounds__(256) void bb_apply_orders_kernel(const int32_t* __restrict__ counts, int cap_edges,
                                                              const int32_t* __restrict__ src,
                                                              const int32_t* __restrict__ dst,
                                                              const int32_t* __restrict__ et,
                                                              const int32_t* __restrict__ ord_dt,
                                                              const int32_t* __restrict__ ord_t,
                                                              const int32_t* __restrict__ ord_t2,
                                                              const int32_t* __restrict__ row_ptr,
                                                              int32_t* __restrict__ col, int32_t* __restrict__ etype,
                                                              int32_t* __restrict__ e_src, int32_t* __restrict__ e_dst,
                                                              int32_t* __restrict__ e_src2, int32_t* __restrict__ e_dst2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap_edges) return;
    const int E = counts[RENET_BB_E];
    if (i < E) {
        const int a = ord_dt[i], b = ord_t[i];
        col[i] = src[a]; etype[i] = et[a];
        e_src[i] = src[b]; e_dst[i] = dst[b];
    }
    const int E_out = row_ptr[counts[RENET_BB_NA]];
    if (i < E_out) { const int c = ord_t2[i]; e_src2[i] = src[c]; e_dst2[i] = dst[c]; }
}

// norm = 1 / max(in-degree, 1) (utils.py:126-127), hub flags, light-row item counts
__global__ __launch_bounds__(256) void bb_rows_kernel(const int32_t* __restrict__ counts, int cap_nodes, int heavy_thr,
                                                      const int32_t* __restrict__ deg, float* __restrict__ norm,
                                                      int32_t* __restrict__ heavy_flag, int32_t* __restrict__ item_cnt,
                                                      int32_t* __restrict__ light_id) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v > cap_nodes) return;
    const int N = counts[RENET_BB_N];
    int hf = 0, ic = 0, li = -1;
    if (v < N) {
        const int d = deg[v];
        norm[v] = 1.f / (float)max(d, 1);
        hf = d > heavy_thr;
        if (!hf) { ic = d + 1; li = v; }
    }
    heavy_flag[v] = hf; item_cnt[v] = ic; light_id[v] = li;
}

__global__ __launch_bounds__(256) void bb_heavy_kernel(const int32_t* __restrict__ counts_c, int32_t* __restrict__ counts,
                                                       int cap_nodes, const int32_t* __restrict__ heavy_flag,
                                                       const int32_t* __restrict__ heavy_pos,
                                                       int32_t* __restrict__ heavy_rows) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = counts_c[RENET_BB_N], nA = counts_c[RENET_BB_NA];
    if (v < N && heavy_flag[v]) heavy_rows[heavy_pos[v]] = v;
    if (v == 0) {
        counts[RENET_BB_NHEAVY] = N > 0 ? heavy_pos[N - 1] + heavy_flag[N - 1] : 0;
        counts[RENET_BB_NHEAVY_OUT] = nA > 0 ? heavy_pos[nA - 1] + heavy_flag[nA - 1] : 0;
    }
    (void)cap_nodes;
}

// chunk lists of the relation-bucketed edge list: <= chunk edges of ONE relation per work item (ONE workgroup per
// list; T2 <= 1024 relation types)
__global__ __launch_bounds__(1024) void bb_chunks_kernel(const int32_t* __restrict__ tc_a, const int32_t* __restrict__ tc_b,
                                                         int T2, int chunk, int cap_chunks,
                                                         int32_t* __restrict__ tcp_a, int32_t* __restrict__ tcp_b,
                                                         int32_t* __restrict__ ctype_a, int32_t* __restrict__ cptr_a,
                                                         int32_t* __restrict__ ctype_b, int32_t* __restrict__ cptr_b,
                                                         int32_t* __restrict__ counts, int32_t* __restrict__ err) {
    __shared__ int wsum[16];
    const int which = blockIdx.x;
    const int32_t* tc = which ? tc_b : tc_a;
    int32_t* tcp = which ? tcp_b : tcp_a;
    int32_t* ctype = which ? ctype_b : ctype_a;
    int32_t* cptr = which ? cptr_b : cptr_a;
    const int t = threadIdx.x;
    const int n = t < T2 ? tc[t] : 0;
    const int nch = (n + chunk - 1) / chunk;
    int tot_e, tot_c;
    const int e0 = block_excl_scan_1024(n, &tot_e, wsum);
    const int c0 = block_excl_scan_1024(nch, &tot_c, wsum);
    if (t < T2) tcp[t] = c0;
    if (t == 0) {
        tcp[T2] = tot_c;
        counts[which ? RENET_BB_NCHUNKS2 : RENET_BB_NCHUNKS] = tot_c;
        if (tot_c > cap_chunks) atomicOr(err, RENET_BB_ERR_EDGES);
    }
    if (tot_c > cap_chunks) return;
    for (int w = 0; w < nch; ++w) { ctype[c0 + w] = t; cptr[c0 + w] = e0 + w * chunk; }
    if (t == 0) cptr[tot_c] = tot_e;
}

// ---- stage G: gather item stream + wave groups (graph.plan_gather_items) --------------------------------------------
__global__ __launch_bounds__(256) void bb_items_kernel(const int32_t* __restrict__ counts, int budget,
                                                       const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                                       const int32_t* __restrict__ etype, const int32_t* __restrict__ item_cnt,
                                                       const int32_t* __restrict__ item_start,
                                                       const int32_t* __restrict__ prev_light,
                                                       int32_t* __restrict__ it_src, int32_t* __restrict__ it_type,
                                                       int32_t* __restrict__ first_flag, int32_t* __restrict__ first_out) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = counts[RENET_BB_N], nA = counts[RENET_BB_NA];
    int ff = 0, fo = 0;
    if (v < N && item_cnt[v] > 0) {
        const int st = item_start[v], e0 = row_ptr[v], d = item_cnt[v] - 1;
        for (int q = 0; q < d; ++q) { it_src[st + q] = col[e0 + q]; it_type[st + q] = etype[e0 + q]; }
        it_src[st + d] = v; it_type[st + d] = -1;
        const int pl = prev_light[v];                     // the previous light row, -1 if none
        const int side = v >= nA;
        ff = pl < 0 || (item_start[pl] / budget) != (st / budget) || ((pl >= nA) != side);
        fo = ff && !side;
    }
    if (v <= N) { first_flag[v] = ff; first_out[v] = fo; }
}

__global__ __launch_bounds__(256) void bb_groups_kernel(const int32_t* __restrict__ counts_c, int32_t* __restrict__ counts,
                                                        int cap_nodes, const int32_t* __restrict__ first_flag,
                                                        const int32_t* __restrict__ first_pos,
                                                        const int32_t* __restrict__ first_out_pos,
                                                        const int32_t* __restrict__ first_out,
                                                        const int32_t* __restrict__ item_start,
                                                        const int32_t* __restrict__ item_cnt,
                                                        int32_t* __restrict__ grp_ptr) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = counts_c[RENET_BB_N];
    if (v < N && first_flag[v]) grp_ptr[first_pos[v]] = item_start[v];
    if (v == 0) {
        const int ng = N > 0 ? first_pos[N - 1] + first_flag[N - 1] : 0;
        const int total = N > 0 ? item_start[N - 1] + item_cnt[N - 1] : 0;
        grp_ptr[ng] = total;
        counts[RENET_BB_NGROUPS] = ng;
        counts[RENET_BB_NGROUPS_OUT] = N > 0 ? first_out_pos[N - 1] + first_out[N - 1] : 0;
        counts[RENET_BB_NITEMS] = total;
    }
    (void)cap_nodes;
}

// ---- stage H: segmented-add plans (graph.SegPlan): rows sorted stably by key, segment starts, segment targets ----
__global__ __launch_bounds__(256) void bb_plan_keys_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ n_ptr,
                                                           int n_fixed, int cap, uint32_t sentinel,
                                                           uint32_t* __restrict__ key, int32_t* __restrict__ iota) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const int n = n_ptr ? *n_ptr : n_fixed;
    key[i] = i < n ? (uint32_t)idx[i] : sentinel;
    iota[i] = i;
}

__global__ __launch_bounds__(256) void bb_plan_flags_kernel(const uint32_t* __restrict__ skey, const int32_t* __restrict__ n_ptr,
                                                            int n_fixed, int cap, int32_t* __restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > cap) return;
    const int n = n_ptr ? *n_ptr : n_fixed;
    flag[i] = (i < n && (i == 0 || skey[i] != skey[i - 1])) ? 1 : 0;
}

__global__ __launch_bounds__(256) void bb_plan_segs_kernel(const uint32_t* __restrict__ skey, const int32_t* __restrict__ n_ptr,
                                                           int n_fixed, int cap, const int32_t* __restrict__ flag,
                                                           const int32_t* __restrict__ pos, int32_t* __restrict__ seg_ptr,
                                                           int32_t* __restrict__ target, int32_t* __restrict__ count_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = n_ptr ? *n_ptr : n_fixed;
    if (i < n && flag[i]) { seg_ptr[pos[i]] = i; target[pos[i]] = (int32_t)skey[i]; }
    if (i == 0) {
        const int u = n > 0 ? pos[n - 1] + flag[n - 1] : 0;
        seg_ptr[u] = n;
        *count_out = u;
    }
    (void)cap;
}

__global__ void bb_finish_kernel(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ step_off,
                                 int32_t* __restrict__ counts) {
    const int t = threadIdx.x;
    if (t <= BB_MAXL) counts[RENET_BB_STEP_OFF + t] = step_off[t];
    if (t == 0) counts[RENET_BB_EOUT] = row_ptr[counts[RENET_BB_NA]];
}

inline int bits_for(uint64_t v) { int b = 1; while (b < 63 && (1ull << b) <= v) ++b; return b; }

#define BB_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return (int)e__; } while (0)

struct Carver {
    char* p; size_t left; size_t used = 0; bool ok = true;
    template <class T> T* take(size_t n) {
        const size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
        used += bytes;
        if (bytes > left) { ok = false; left = 0; return nullptr; }
        T* r = reinterpret_cast<T*>(p);
        p += bytes; left -= bytes;
        return r;
    }
};

size_t rocprim_temp_bytes(int cap_nodes, int cap_edges, int cap_facts) {
    size_t m = 0, t = 0;
    (void)rocprim::exclusive_scan(nullptr, t, (int*)nullptr, (int*)nullptr, 0,
                                  (size_t)max(max(cap_nodes + 1, cap_facts), 1), rocprim::plus<int>());
    m = max(m, t);
    (void)rocprim::exclusive_scan(nullptr, t, (int*)nullptr, (int*)nullptr, -1, (size_t)max(cap_nodes + 1, 1),
                                  rocprim::maximum<int>());
    m = max(m, t);
    (void)rocprim::radix_sort_pairs(nullptr, t, (uint32_t*)nullptr, (uint32_t*)nullptr, (int*)nullptr, (int*)nullptr,
                                    (size_t)max(max(cap_edges, cap_nodes), 1), 0, 32);
    m = max(m, t);
    return (m + 255) & ~(size_t)255;
}

struct Bufs {
    int32_t* seq_first;
    int32_t* seq_len_s;
    int32_t* seq_start;
    int32_t* step_snap;
    int32_t* step_dense;
    int32_t* step_packed;
    int32_t* slot_used;
    int32_t* slot_of_dense;
    int32_t* slot_ti;
    int32_t* slot_group;
    int32_t* fact_off;
    uint32_t* table;
    int32_t* new_id;
    int2* tile_cnt;
    int32_t* flag;
    int32_t* pos;
    int32_t* fslot;
    int32_t* half_src;
    int32_t* half_dst;
    int32_t* half_et;
    int32_t* src;
    int32_t* dst;
    int32_t* et;
    uint32_t* key_dt;
    uint32_t* key_t;
    uint32_t* key_t2;
    uint32_t* key_sorted;
    int32_t* iota;
    int32_t* ord_dt;
    int32_t* ord_t;
    int32_t* ord_t2;
    int32_t* deg;
    int32_t* tc;
    int32_t* tc2;
    int32_t* heavy_flag;
    int32_t* heavy_pos;
    int32_t* item_cnt;
    int32_t* item_start;
    int32_t* light_id;
    int32_t* prev_light;
    int32_t* first_flag;
    int32_t* first_pos;
    int32_t* first_out;
    int32_t* first_out_pos;
    uint32_t* pkey;
    int32_t* pflag;
    int32_t* ppos;
    
    void* tmp;
    size_t tmp_bytes;
    // carves every scratch array out of `cv`; false if it does not fit
    bool carve(Carver& cv, const RenetStoreDev* sd, int B, int cap_nodes, int cap_edges) {
        const int cap_facts = 2 * sd->n_facts, cap_steps = 2 * B * BB_MAXL;
        const size_t entries_cap = (size_t)2 * sd->T * sd->num_ent;
        const int n_tiles = (int)((entries_cap + NUM_TILE - 1) / NUM_TILE);
        seq_first = cv.take<int32_t>(BB_MAXQ);
        seq_len_s = cv.take<int32_t>(BB_MAXQ);
        seq_start = cv.take<int32_t>(BB_MAXQ);
        step_snap = cv.take<int32_t>(cap_steps);
        step_dense = cv.take<int32_t>(cap_steps);
        step_packed = cv.take<int32_t>(cap_steps);
        slot_used = cv.take<int32_t>(2 * sd->T + 2);
        slot_of_dense = cv.take<int32_t>(2 * sd->T + 2);
        slot_ti = cv.take<int32_t>(2 * sd->T + 2);
        slot_group = cv.take<int32_t>(2 * sd->T + 2);
        fact_off = cv.take<int32_t>(2 * sd->T + 2);
        table = cv.take<uint32_t>(entries_cap / 4 + 16);
        new_id = cv.take<int32_t>(entries_cap);
        tile_cnt = cv.take<int2>(n_tiles + 1);
        flag = cv.take<int32_t>(cap_facts + 1);
        pos = cv.take<int32_t>(cap_facts + 1);
        fslot = cv.take<int32_t>(cap_facts + 1);
        half_src = cv.take<int32_t>(cap_edges / 2);
        half_dst = cv.take<int32_t>(cap_edges / 2);
        half_et = cv.take<int32_t>(cap_edges / 2);
        src = cv.take<int32_t>(cap_edges);
        dst = cv.take<int32_t>(cap_edges);
        et = cv.take<int32_t>(cap_edges);
        key_dt = cv.take<uint32_t>(cap_edges);
        key_t = cv.take<uint32_t>(cap_edges);
        key_t2 = cv.take<uint32_t>(cap_edges);
        key_sorted = cv.take<uint32_t>(max(cap_edges, max(cap_nodes, cap_steps)));
        iota = cv.take<int32_t>(max(cap_edges, max(cap_nodes, cap_steps)));
        ord_dt = cv.take<int32_t>(cap_edges);
        ord_t = cv.take<int32_t>(cap_edges);
        ord_t2 = cv.take<int32_t>(cap_edges);
        deg = cv.take<int32_t>(cap_nodes + 2);
        tc = cv.take<int32_t>(1024);
        tc2 = cv.take<int32_t>(1024);
        heavy_flag = cv.take<int32_t>(cap_nodes + 2);
        heavy_pos = cv.take<int32_t>(cap_nodes + 2);
        item_cnt = cv.take<int32_t>(cap_nodes + 2);
        item_start = cv.take<int32_t>(cap_nodes + 2);
        light_id = cv.take<int32_t>(cap_nodes + 2);
        prev_light = cv.take<int32_t>(cap_nodes + 2);
        first_flag = cv.take<int32_t>(cap_nodes + 2);
        first_pos = cv.take<int32_t>(cap_nodes + 2);
        first_out = cv.take<int32_t>(cap_nodes + 2);
        first_out_pos = cv.take<int32_t>(cap_nodes + 2);
        pkey = cv.take<uint32_t>(max(cap_nodes, cap_steps));
        pflag = cv.take<int32_t>(max(cap_nodes, cap_steps) + 2);
        ppos = cv.take<int32_t>(max(cap_nodes, cap_steps) + 2);
        tmp_bytes = rocprim_temp_bytes(cap_nodes, cap_edges, cap_facts);
        tmp = cv.take<char>(tmp_bytes);
        return tmp != nullptr && cv.ok;
    }
};

}  // namespace

extern "C" {

size_t renet_build_batch_workspace(const RenetStoreDev* sd, int B, int cap_nodes, int cap_edges) {
    if (!sd || B <= 0 || cap_nodes <= 0 || cap_edges <= 0) return 0;
    Carver dry{reinterpret_cast<char*>(256), ~(size_t)0 >> 2};          // never dereferenced
    Bufs b;
    b.carve(dry, sd, B, cap_nodes, cap_edges & ~1);
    return dry.used + 256;
}

int renet_build_batch_both(const RenetStoreDev* sd, const int32_t* idx_dev, int B, int seq_len, int heavy_thr,
                           int group_budget, int chunk, const RenetBatchOut* out, void* workspace,
                           size_t workspace_bytes, void* stream) {
    if (!sd || !out || !idx_dev || B <= 0 || 2 * B > BB_MAXQ || seq_len <= 0 || seq_len > BB_MAXL) return RENET_ERR_BADARG;
    if (2 * sd->num_rels > 1024 || sd->T <= 0 || group_budget + heavy_thr + 1 > 64 || chunk <= 0) return RENET_ERR_UNSUPPORTED;
    if (workspace_bytes < renet_build_batch_workspace(sd, B, out->cap_nodes, out->cap_edges)) return RENET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Store S;
    S.q_s = sd->q_s; S.q_r = sd->q_r; S.q_o = sd->q_o;
    for (int r = 0; r < 2; ++r) {
        S.h_first[r] = sd->h_first[r]; S.h_count[r] = sd->h_count[r]; S.snap_t[r] = sd->snap_t[r];
        S.snap_ptr[r] = sd->snap_ptr[r]; S.nbr_o[r] = sd->nbr_o[r];
    }
    S.times = sd->times; S.trip_ptr = sd->trip_ptr; S.trip_s = sd->trip_s; S.trip_r = sd->trip_r; S.trip_o = sd->trip_o;
    S.glob_times = sd->glob_times; S.T = sd->T; S.n_glob = sd->n_glob; S.num_ent = sd->num_ent; S.num_rels = sd->num_rels;
    const int cap_nodes = out->cap_nodes, cap_edges = out->cap_edges & ~1;
    const int cap_facts = 2 * sd->n_facts, cap_steps = 2 * B * BB_MAXL;
    const int T2 = 2 * sd->num_rels;
    const size_t entries_cap = (size_t)2 * sd->T * sd->num_ent;
    const int n_tiles = (int)((entries_cap + NUM_TILE - 1) / NUM_TILE);
    const int cap_chunks = cap_edges / chunk + T2 + 1;
    if ((uint64_t)cap_nodes * T2 >= (1ull << 31)) return RENET_ERR_UNSUPPORTED;
    const int key_bits = bits_for((uint64_t)cap_nodes * T2);

    Carver cv{reinterpret_cast<char*>(workspace), workspace_bytes};
    Bufs bf;
    if (!bf.carve(cv, sd, B, cap_nodes, cap_edges)) return RENET_ERR_WORKSPACE;
    int32_t* counts = out->counts;
    int32_t* err = counts + RENET_BB_ERR;

    BB_HIP(hipMemsetAsync(counts, 0, RENET_BB_NCOUNTS * sizeof(int32_t), st));
    BB_HIP(hipMemsetAsync(bf.slot_used, 0, (size_t)(2 * sd->T + 2) * sizeof(int32_t), st));
    BB_HIP(hipMemsetAsync(bf.table, 0, entries_cap + 64, st));
    BB_HIP(hipMemsetAsync(bf.deg, 0, (size_t)(cap_nodes + 2) * sizeof(int32_t), st));
    BB_HIP(hipMemsetAsync(bf.tc, 0, 2 * 1024 * sizeof(int32_t), st));      // bf.tc and bf.tc2 are adjacent 4 KB blocks

    RENET_LAUNCH(bb_seq_kernel, dim3(1), dim3(1024), 0, st, S, idx_dev, B, seq_len, out->perm, bf.seq_first, bf.seq_len_s,
                 bf.seq_start, out->s_sorted, out->r_sorted, out->rel_label, out->ent_label, out->step_off, counts);
    RENET_LAUNCH_CHECK();
    RENET_LAUNCH(bb_steps_kernel, dim3((2 * B * BB_MAXL + 255) / 256), dim3(256), 0, st, S, B, out->perm, bf.seq_first,
                 bf.seq_len_s, bf.seq_start, out->s_sorted, out->r_sorted, out->step_off, counts, bf.step_snap, bf.step_dense,
                 bf.step_packed, bf.slot_used, out->row_seq, out->row_ent, out->row_rel, out->glob_row, err);
    RENET_LAUNCH_CHECK();
    RENET_LAUNCH(bb_slots_kernel, dim3(1), dim3(1024), 0, st, S, bf.slot_used, bf.slot_of_dense, bf.slot_ti, bf.slot_group, bf.fact_off,
                 counts);
    RENET_LAUNCH_CHECK();
    RENET_LAUNCH(bb_mark_kernel, dim3((cap_steps + 3) / 4), dim3(256), 0, st, S, counts, bf.step_snap, bf.step_dense,
                 bf.step_packed, out->row_ent, bf.slot_of_dense, bf.table);
    RENET_LAUNCH_CHECK();
    RENET_LAUNCH(bb_tile_count_kernel, dim3(n_tiles), dim3(256), 0, st, (const uint8_t*)bf.table, entries_cap, counts,
                 sd->num_ent, bf.tile_cnt);
    RENET_LAUNCH_CHECK();
    RENET_LAUNCH(bb_tile_scan_kernel, dim3(1), dim3(1024), 0, st, bf.tile_cnt, n_tiles, counts, cap_nodes, err);
    RENET_LAUNCH_CHECK();
    RENET_LAUNCH(bb_number_kernel, dim3(n_tiles), dim3(256), 0, st, (const uint8_t*)bf.table, bf.tile_cnt, counts, sd->num_ent,
                 cap_nodes, bf.new_id, out->node_ent, out->node_slot);
    RENET_LAUNCH_CHECK();
    RENET_LAUNCH(bb_subj_row_kernel, dim3((cap_steps + 255) / 256), dim3(256), 0, st, sd->num_ent, counts, bf.step_dense,
                 bf.step_packed, out->row_ent, bf.slot_of_dense, bf.new_id, out->subj_row);
    RENET_LAUNCH_CHECK();
    // induced edges
    RENET_LAUNCH(bb_fact_flag_kernel, dim3((cap_facts + 255) / 256), dim3(256), 0, st, S, counts, bf.fact_off, bf.slot_ti,
                 (const uint8_t*)bf.table, cap_facts, bf.flag, bf.fslot);
    RENET_LAUNCH_CHECK();
    size_t tb = bf.tmp_bytes;
    BB_HIP(rocprim::exclusive_scan(bf.tmp, tb, bf.flag, bf.pos, 0, (size_t)cap_facts, rocprim::plus<int>(), st));
    RENET_LAUNCH(bb_set_e2_kernel, dim3(1), dim3(1), 0, st, bf.flag, bf.pos, counts, cap_facts, cap_edges);
    RENET_LAUNCH_CHECK();
    RENET_LAUNCH(bb_edges_kernel, dim3((cap_facts + 255) / 256), dim3(256), 0, st, S, counts, bf.fact_off, bf.slot_ti,
                 bf.slot_group, bf.flag, bf.pos, bf.fslot, bf.new_id, cap_facts, cap_edges, bf.half_src, bf.half_dst, bf.half_et, err);
    RENET_LAUNCH_CHECK();
    RENET_LAUNCH(bb_expand_kernel, dim3(min((cap_edges + 255) / 256, 1024)), dim3(256), 0, st, counts, sd->num_rels, cap_edges,
                 key_bits, bf.half_src, bf.half_dst, bf.half_et, bf.src, bf.dst, bf.et, bf.key_dt, bf.key_t, bf.key_t2, bf.iota, bf.deg, bf.tc, bf.tc2);
    RENET_LAUNCH_CHECK();
    tb = bf.tmp_bytes;
    BB_HIP(rocprim::radix_sort_pairs(bf.tmp, tb, bf.key_dt, bf.key_sorted, bf.iota, bf.ord_dt, (size_t)cap_edges, 0, key_bits + 1, st));
    const int tbits = bits_for((uint64_t)T2) ;
    tb = bf.tmp_bytes;
    BB_HIP(rocprim::radix_sort_pairs(bf.tmp, tb, bf.key_t, bf.key_sorted, bf.iota, bf.ord_t, (size_t)cap_edges, 0, tbits, st));
    tb = bf.tmp_bytes;
    BB_HIP(rocprim::radix_sort_pairs(bf.tmp, tb, bf.key_t2, bf.key_sorted, bf.iota, bf.ord_t2, (size_t)cap_edges, 0, tbits, st));
    // rows
    tb = bf.tmp_bytes;
    BB_HIP(rocprim::exclusive_scan(bf.tmp, tb, bf.deg, out->row_ptr, 0, (size_t)(cap_nodes + 1), rocprim::plus<int>(), st));
    RENET_LAUNCH(bb_rows_kernel, dim3((cap_nodes + 1 + 255) / 256), dim3(256), 0, st, counts, cap_nodes, heavy_thr, bf.deg,
                 out->norm, bf.heavy_flag, bf.item_cnt, bf.light_id);
    RENET_LAUNCH_CHECK();
    RENET_LAUNCH(bb_apply_orders_kernel, dim3((cap_edges + 255) / 256), dim3(256), 0, st, counts, cap_edges, bf.src, bf.dst, bf.et,
                 bf.ord_dt, bf.ord_t, bf.ord_t2, out->row_ptr, out->col, out->etype, out->e_src, out->e_dst, out->e_src2,
                 out->e_dst2);
    RENET_LAUNCH_CHECK();
    tb = bf.tmp_bytes;
    BB_HIP(rocprim::exclusive_scan(bf.tmp, tb, bf.heavy_flag, bf.heavy_pos, 0, (size_t)(cap_nodes + 1), rocprim::plus<int>(), st));
    RENET_LAUNCH(bb_heavy_kernel, dim3((cap_nodes + 255) / 256), dim3(256), 0, st, counts, counts, cap_nodes, bf.heavy_flag,
                 bf.heavy_pos, out->heavy_rows);
    RENET_LAUNCH_CHECK();
    RENET_LAUNCH(bb_chunks_kernel, dim3(2), dim3(1024), 0, st, bf.tc, bf.tc2, T2, chunk, cap_chunks, out->type_chunk_ptr,
                 out->type_chunk_ptr2, out->chunk_type, out->chunk_ptr, out->chunk_type2, out->chunk_ptr2, counts, err);
    RENET_LAUNCH_CHECK();
    // gather item plan
    tb = bf.tmp_bytes;
    BB_HIP(rocprim::exclusive_scan(bf.tmp, tb, bf.item_cnt, bf.item_start, 0, (size_t)(cap_nodes + 1), rocprim::plus<int>(), st));
    tb = bf.tmp_bytes;
    BB_HIP(rocprim::exclusive_scan(bf.tmp, tb, bf.light_id, bf.prev_light, -1, (size_t)(cap_nodes + 1), rocprim::maximum<int>(), st));
    RENET_LAUNCH(bb_items_kernel, dim3((cap_nodes + 1 + 255) / 256), dim3(256), 0, st, counts, group_budget, out->row_ptr,
                 out->col, out->etype, bf.item_cnt, bf.item_start, bf.prev_light, out->it_src, out->it_type, bf.first_flag, bf.first_out);
    RENET_LAUNCH_CHECK();
    tb = bf.tmp_bytes;
    BB_HIP(rocprim::exclusive_scan(bf.tmp, tb, bf.first_flag, bf.first_pos, 0, (size_t)(cap_nodes + 1), rocprim::plus<int>(), st));
    tb = bf.tmp_bytes;
    BB_HIP(rocprim::exclusive_scan(bf.tmp, tb, bf.first_out, bf.first_out_pos, 0, (size_t)(cap_nodes + 1), rocprim::plus<int>(), st));
    RENET_LAUNCH(bb_groups_kernel, dim3((cap_nodes + 255) / 256), dim3(256), 0, st, counts, counts, cap_nodes, bf.first_flag,
                 bf.first_pos, bf.first_out_pos, bf.first_out, bf.item_start, bf.item_cnt, out->grp_ptr);
    RENET_LAUNCH_CHECK();
    // segmented-add plans: 0 node_ent (N rows), 1 subj_row (S rows), 2 s_sorted (2B), 3 r_sorted (2B)
    for (int pl = 0; pl < 4; ++pl) {
        const int32_t* idx = pl == 0 ? out->node_ent : pl == 1 ? out->subj_row : pl == 2 ? out->s_sorted : out->r_sorted;
        const int32_t* n_ptr = pl == 0 ? counts + RENET_BB_N : pl == 1 ? counts + RENET_BB_S : nullptr;
        const int n_fixed = 2 * B;
        const int cap = pl == 0 ? cap_nodes : pl == 1 ? cap_steps : 2 * B;
        const uint64_t bound = pl == 0 ? (uint64_t)sd->num_ent : pl == 1 ? (uint64_t)cap_nodes
                               : pl == 2 ? (uint64_t)sd->num_ent : (uint64_t)T2;
        const int kb = bits_for(bound);
        RENET_LAUNCH(bb_plan_keys_kernel, dim3((cap + 255) / 256), dim3(256), 0, st, idx, n_ptr, n_fixed, cap,
                     (uint32_t)(1u << kb), bf.pkey, bf.iota);
        RENET_LAUNCH_CHECK();
        tb = bf.tmp_bytes;
        BB_HIP(rocprim::radix_sort_pairs(bf.tmp, tb, bf.pkey, bf.key_sorted, bf.iota, out->plan_order[pl], (size_t)cap, 0, kb + 1, st));
        RENET_LAUNCH(bb_plan_flags_kernel, dim3((cap + 1 + 255) / 256), dim3(256), 0, st, bf.key_sorted, n_ptr, n_fixed, cap,
                     bf.pflag);
        RENET_LAUNCH_CHECK();
        tb = bf.tmp_bytes;
        BB_HIP(rocprim::exclusive_scan(bf.tmp, tb, bf.pflag, bf.ppos, 0, (size_t)(cap + 1), rocprim::plus<int>(), st));
        RENET_LAUNCH(bb_plan_segs_kernel, dim3((cap + 255) / 256), dim3(256), 0, st, bf.key_sorted, n_ptr, n_fixed, cap, bf.pflag,
                     bf.ppos, out->plan_seg[pl], out->plan_target[pl], counts + RENET_BB_NSEG0 + pl);
        RENET_LAUNCH_CHECK();
    }
    RENET_LAUNCH(bb_finish_kernel, dim3(1), dim3(64), 0, st, out->row_ptr, out->step_off, counts);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

}  // extern "C"
