// Native HOST half of the batch-graph builder (no device code): the O(#facts of the batch's timestamps)
// passes of graph.build_batch -- node-induced edge filtering, the two stable counting sorts that produce
// the CSR-by-destination and the relation-bucketed edge list, chunking, hub rows, segmented-add plans.
// Replaces the reference's per-batch DGL subgraph / batch calls (utils.py:115-131,158-170,236-241) and the
// numpy versions in graph.py (kept as the executable specification: tests compare the two bit for bit).
// Plain C ABI, caller-owned buffers, no allocation that outlives a call, thread-safe given distinct scratch.
#include <stdint.h>
#include <stddef.h>
#include <vector>
#include <algorithm>

extern "C" {

// Slot by slot (slot c = the graph of timestamp ti[c]; keys = sorted slot * num_ent + entity): marks the slot's
// nodes in `table`, scans every fact of the timestamp, keeps those whose two endpoints are both marked,
// restores the table to -1.  `table` is a caller-owned int32 scratch of at least num_ent entries, all -1 on entry.  Outputs (capacity = total facts of those timestamps): local source /
// destination rows and relation of every kept fact, in (slot, fact) order.  Returns the number kept.
int64_t renet_host_filter_edges(const int64_t* trip_ptr, const int64_t* trip_s, const int64_t* trip_r,
                                const int64_t* trip_o, const int64_t* ti, int64_t Tb, int64_t num_ent,
                                const int64_t* keys, const int32_t* new_id, int64_t N, int32_t* table,
                                int64_t* out_ls, int64_t* out_lo, int64_t* out_rr) {
    // one slot at a time through the FIRST num_ent entries of the table (92 KB at 23k entities: cache resident;
    // a (slot, entity) table of all slots is 22 MB and every endpoint lookup missed)
    int64_t m = 0, k0 = 0;
    for (int64_t slot = 0; slot < Tb; ++slot) {
        int64_t k1 = k0;
        const int64_t base = slot * num_ent, lim = base + num_ent;
        while (k1 < N && keys[k1] < lim) ++k1;                       // keys are sorted: this slot's nodes
        for (int64_t i = k0; i < k1; ++i) table[keys[i] - base] = new_id[i];
        const int64_t b = trip_ptr[ti[slot]], e = trip_ptr[ti[slot] + 1];
        for (int64_t j = b; j < e; ++j) {
            const int32_t ps = table[trip_s[j]];
            const int32_t po = table[trip_o[j]];
            if ((ps | po) >= 0) {                         // both non-negative
                out_ls[m] = ps; out_lo[m] = po; out_rr[m] = trip_r[j];
                ++m;
            }
        }
        for (int64_t i = k0; i < k1; ++i) table[keys[i] - base] = -1;
        k0 = k1;
    }
    return m;
}

// The same filter for MANY small member graphs (batched inference: one slot per (entity, timestamp), node sets
// of a few dozen entities): instead of scanning all facts of every slot's timestamp, walk only the facts whose
// SUBJECT is in the slot's node set, through a per-timestamp subject index (by_subj: fact indices sorted by
// (timestamp, subject), stable; subj_sorted: their subjects).  `table` is an int32 scratch of num_ent entries,
// all -1 on entry and on return.  Output order = (slot, fact), identical to renet_host_filter_edges.
int64_t renet_host_filter_edges_sparse(const int64_t* trip_ptr, const int64_t* trip_s, const int64_t* trip_r,
                                       const int64_t* trip_o, const int64_t* by_subj,
                                       const int64_t* subj_sorted, const int64_t* ti, int64_t Tb,
                                       int64_t num_ent, const int64_t* keys, const int32_t* new_id, int64_t N,
                                       int32_t* table, int64_t* out_ls, int64_t* out_lo, int64_t* out_rr) {
    int64_t m = 0, k0 = 0;
    std::vector<int64_t> facts;
    for (int64_t slot = 0; slot < Tb; ++slot) {
        int64_t k1 = k0;
        while (k1 < N && keys[k1] < (slot + 1) * num_ent) ++k1;          // keys are sorted: this slot's nodes
        const int64_t base = slot * num_ent;
        for (int64_t i = k0; i < k1; ++i) table[keys[i] - base] = new_id[i];
        const int64_t b = trip_ptr[ti[slot]], e = trip_ptr[ti[slot] + 1];
        facts.clear();
        for (int64_t i = k0; i < k1; ++i) {
            const int64_t u = keys[i] - base;
            const int64_t* q = std::lower_bound(subj_sorted + b, subj_sorted + e, u);
            for (; q < subj_sorted + e && *q == u; ++q) {
                const int64_t j = by_subj[q - subj_sorted];
                if (table[trip_o[j]] >= 0) facts.push_back(j);
            }
        }
        std::sort(facts.begin(), facts.end());                           // back to fact order
        for (int64_t j : facts) {
            out_ls[m] = table[trip_s[j]]; out_lo[m] = table[trip_o[j]]; out_rr[m] = trip_r[j];
            ++m;
        }
        for (int64_t i = k0; i < k1; ++i) table[keys[i] - base] = -1;
        k0 = k1;
    }
    return m;
}

// Directed edges (src -> dst, type et in [0,T)) -> CSR by destination with relation-sorted rows, norm,
// hub rows (in-degree > heavy), the relation-bucketed edge list and its <= chunk-edge single-type chunks.
// Stable throughout (ties keep the input edge order) == graph.HostBatch.set_edges.
void renet_host_edge_layouts(int64_t n, int64_t E, const int64_t* src, const int64_t* dst, const int64_t* et,
                             int64_t T, int64_t chunk, int64_t heavy, int32_t* col, int32_t* etype,
                             int32_t* row_ptr, float* norm, int32_t* heavy_rows, int64_t* n_heavy,
                             int32_t* e_src, int32_t* e_dst, int32_t* type_chunk_ptr, int32_t* chunk_type,
                             int32_t* chunk_ptr, int64_t* n_chunks) {
    std::vector<int64_t> tstart(T + 1, 0), by_type(E);
    for (int64_t e = 0; e < E; ++e) ++tstart[et[e] + 1];
    for (int64_t t = 0; t < T; ++t) tstart[t + 1] += tstart[t];
    {
        std::vector<int64_t> cur(tstart.begin(), tstart.end() - 1);
        for (int64_t e = 0; e < E; ++e) by_type[cur[et[e]]++] = e;
    }
    for (int64_t k = 0; k < E; ++k) { e_src[k] = (int32_t)src[by_type[k]]; e_dst[k] = (int32_t)dst[by_type[k]]; }
    std::vector<int64_t> rstart(n + 1, 0);
    for (int64_t e = 0; e < E; ++e) ++rstart[dst[e] + 1];
    int64_t nh = 0;
    for (int64_t v = 0; v < n; ++v) {
        const int64_t d = rstart[v + 1];
        norm[v] = 1.0f / (float)(d > 0 ? d : 1);
        if (d > heavy) heavy_rows[nh++] = (int32_t)v;
        rstart[v + 1] += rstart[v];
    }
    *n_heavy = nh;
    for (int64_t v = 0; v <= n; ++v) row_ptr[v] = (int32_t)rstart[v];
    {
        std::vector<int64_t> cur(rstart.begin(), rstart.end() - 1);
        for (int64_t k = 0; k < E; ++k) {                 // by_type order => rows end up sorted by type, stably
            const int64_t e = by_type[k];
            const int64_t p = cur[dst[e]]++;
            col[p] = (int32_t)src[e];
            etype[p] = (int32_t)et[e];
        }
    }
    int64_t nc = 0;
    for (int64_t t = 0; t < T; ++t) {
        type_chunk_ptr[t] = (int32_t)nc;
        for (int64_t b = tstart[t]; b < tstart[t + 1]; b += chunk) {
            chunk_type[nc] = (int32_t)t;
            chunk_ptr[nc] = (int32_t)b;
            ++nc;
        }
    }
    type_chunk_ptr[T] = (int32_t)nc;
    chunk_ptr[nc] = (int32_t)E;
    *n_chunks = nc;
}

// Sorted plan for renet_segment_add: order = stable argsort(idx); one segment per distinct value.
// idx values in [0, bound).  Returns the number of segments.
int64_t renet_host_segplan(const int64_t* idx, int64_t n, int64_t bound, int32_t* order, int32_t* seg_ptr,
                           int32_t* target) {
    std::vector<int64_t> start(bound + 1, 0);
    for (int64_t i = 0; i < n; ++i) ++start[idx[i] + 1];
    int64_t U = 0;
    for (int64_t v = 0; v < bound; ++v) {
        if (start[v + 1] > 0) { target[U] = (int32_t)v; seg_ptr[U] = (int32_t)start[v]; ++U; }
        start[v + 1] += start[v];
    }
    seg_ptr[U] = (int32_t)n;
    std::vector<int64_t> cur(start.begin(), start.end() - 1);
    for (int64_t i = 0; i < n; ++i) order[cur[idx[i]]++] = (int32_t)i;
    return U;
}


// The relation-bucketed chunk list restricted to edges whose destination is < n_out (evaluating a layer on a
// row prefix: graph.HostBatch.set_out_rows) -- only the second half of renet_host_edge_layouts, on the kept
// edges, without materialising the filtered edge list.  Capacity of e_src / e_dst: E; of the chunk arrays:
// E / chunk + T + 1.  Returns the number of kept edges.
int64_t renet_host_type_chunks(int64_t E, const int64_t* src, const int64_t* dst, const int64_t* et, int64_t T,
                               int64_t chunk, int64_t n_out, int32_t* e_src, int32_t* e_dst,
                               int32_t* type_chunk_ptr, int32_t* chunk_type, int32_t* chunk_ptr,
                               int64_t* n_chunks) {
    std::vector<int64_t> tstart(T + 1, 0);
    for (int64_t e = 0; e < E; ++e)
        if (dst[e] < n_out) ++tstart[et[e] + 1];
    for (int64_t t = 0; t < T; ++t) tstart[t + 1] += tstart[t];
    {
        std::vector<int64_t> cur(tstart.begin(), tstart.end() - 1);
        for (int64_t e = 0; e < E; ++e)
            if (dst[e] < n_out) {
                const int64_t k = cur[et[e]]++;
                e_src[k] = (int32_t)src[e];
                e_dst[k] = (int32_t)dst[e];
            }
    }
    const int64_t kept = tstart[T];
    int64_t nc = 0;
    for (int64_t t = 0; t < T; ++t) {
        type_chunk_ptr[t] = (int32_t)nc;
        for (int64_t b = tstart[t]; b < tstart[t + 1]; b += chunk) {
            chunk_type[nc] = (int32_t)t;
            chunk_ptr[nc] = (int32_t)b;
            ++nc;
        }
    }
    type_chunk_ptr[T] = (int32_t)nc;
    chunk_ptr[nc] = (int32_t)kept;
    *n_chunks = nc;
    return kept;
}

// Node sets of a batch (utils.py:149-156): per slot (member graph) the union of the subjects and the history
// objects of the steps that fall into it.  Step k belongs to slot slot_k[k], has subject subj_ent[k] and the
// objects nbr_o[nbr_begin[k] .. nbr_begin[k] + nbr_cnt[k]).  Outputs, identical to the numpy specification in
// graph.build_batch (np.unique over slot * num_ent + entity, subject rows numbered first):
//   keys[N]      sorted slot * num_ent + entity          new_id[N]   row of key i (rows that are a step's
//   subj_pos[S]  index into keys of every step's subject             subject come first, in key order)
//   node_ent[N], node_slot[N]  entity / slot of every ROW (new order);  *n_a = number of subject rows
// `table`: int32 scratch of num_ent entries, -1 on entry and on return.  Returns N (capacity: S + total objects).
int64_t renet_host_node_sets(int64_t S, const int64_t* slot_k, const int64_t* subj_ent, const int64_t* nbr_begin,
                             const int64_t* nbr_cnt, const int64_t* nbr_o, int64_t Tb, int64_t num_ent,
                             int32_t* table, int64_t* keys, int64_t* subj_pos, int64_t* new_id,
                             int32_t* node_ent, int64_t* node_slot, int64_t* n_a) {
    std::vector<int64_t> start(Tb + 1, 0), steps(S);
    for (int64_t k = 0; k < S; ++k) ++start[slot_k[k] + 1];
    for (int64_t t = 0; t < Tb; ++t) start[t + 1] += start[t];
    {
        std::vector<int64_t> cur(start.begin(), start.end() - 1);
        for (int64_t k = 0; k < S; ++k) steps[cur[slot_k[k]]++] = k;
    }
    std::vector<int64_t> ents;
    std::vector<uint8_t> is_a;
    int64_t N = 0;
    for (int64_t t = 0; t < Tb; ++t) {
        ents.clear();
        for (int64_t q = start[t]; q < start[t + 1]; ++q) {
            const int64_t k = steps[q];
            if (table[subj_ent[k]] < 0) { table[subj_ent[k]] = 0; ents.push_back(subj_ent[k]); }
            for (int64_t j = nbr_begin[k]; j < nbr_begin[k] + nbr_cnt[k]; ++j)
                if (table[nbr_o[j]] < 0) { table[nbr_o[j]] = 0; ents.push_back(nbr_o[j]); }
        }
        std::sort(ents.begin(), ents.end());
        for (size_t i = 0; i < ents.size(); ++i) {
            keys[N + (int64_t)i] = t * num_ent + ents[i];
            table[ents[i]] = (int32_t)(N + (int64_t)i) + 1;              // position + 1 (0 = "seen")
        }
        is_a.resize((size_t)(N + (int64_t)ents.size()), 0);
        for (int64_t q = start[t]; q < start[t + 1]; ++q) {
            const int64_t k = steps[q];
            const int64_t pos = (int64_t)table[subj_ent[k]] - 1;
            subj_pos[k] = pos;
            is_a[(size_t)pos] = 1;
        }
        for (int64_t e : ents) table[e] = -1;
        N += (int64_t)ents.size();
    }
    int64_t na = 0;
    for (int64_t i = 0; i < N; ++i) na += is_a[(size_t)i];
    int64_t ia = 0, ib = na;
    for (int64_t i = 0; i < N; ++i) {
        const int64_t r = is_a[(size_t)i] ? ia++ : ib++;
        new_id[i] = r;
        node_ent[r] = (int32_t)(keys[i] % num_ent);
        node_slot[r] = keys[i] / num_ent;
    }
    *n_a = na;
    return N;
}

// Item stream + wave groups of the gather-SpMM kernels (renet_rgcn_gather_items); numpy specification:
// graph.plan_gather_items.  Light rows (in-degree <= heavy) contribute their in-edges (col, etype) in CSR order
// followed by a flush item (row, -1); a new group starts at the first light row, when a row's first item falls
// into a new `budget`-sized window of the stream, and at the first light row >= n_out.
// Capacity: it_src / it_type E + N entries, grp_ptr N + 2.  Returns the number of groups (-1: budget + heavy + 1 > 64).
int64_t renet_host_gather_items(int64_t N, const int32_t* row_ptr, const int32_t* col, const int32_t* etype,
                                int64_t heavy, int64_t budget, int64_t n_out, int32_t* it_src, int32_t* it_type,
                                int32_t* grp_ptr, int64_t* n_items, int64_t* n_groups_out) {
    if (budget + heavy + 1 > 64 || budget < 1 || heavy < 0) return -1;
    int64_t pos = 0, ng = 0, ngo = 0, prev_key = -1;
    bool any = false, prev_side = false;
    for (int64_t v = 0; v < N; ++v) {
        const int64_t e0 = row_ptr[v], e1 = row_ptr[v + 1];
        if (e1 - e0 > heavy) continue;
        const int64_t key = pos / budget;
        const bool side = v >= n_out;
        if (!any || key != prev_key || side != prev_side) {
            grp_ptr[ng++] = (int32_t)pos;
            if (!side) ++ngo;
        }
        any = true; prev_key = key; prev_side = side;
        for (int64_t e = e0; e < e1; ++e) { it_src[pos] = col[e]; it_type[pos] = etype[e]; ++pos; }
        it_src[pos] = (int32_t)v; it_type[pos] = -1; ++pos;
    }
    grp_ptr[ng] = (int32_t)pos;
    *n_items = pos;
    *n_groups_out = ngo;
    return ng;
}

}  // extern "C"
