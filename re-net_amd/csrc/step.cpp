// The merged training step as ONE launch list issued from C (include/renet_hip.h: renet_step_*; round 6).
//
// One iteration of the reference's training loop body (train.py:136-139: the two model() calls of a batch and
// loss.backward()) is ~55 launches of this library.  From Python every one of them costs an autograd Function, tensor
// allocations and a ctypes call (~37 us each: the launching thread needed 1.97 ms for a step the GPU runs in 2.75 ms); here
// the sequence is a C function per direction of the pass: the arguments are read from three plain structs the caller fills
// once per batch, all intermediates live at fixed offsets of ONE workspace, and the two streams are forked / joined with
// HIP events.  This file contains NO arithmetic: every launch goes through an extern "C" entry point of the library with
// the arguments re-net_amd/ops.py passes on the autograd path (RGCNTableLayerFn, RGCNLayerFn, SeqAssembleFn, MultiGRUFn,
// DualHeadCEFn -- each block below names the Function it mirrors), so the two paths produce bit-identical losses and
// gradients (tests/test_gpu_step_plan.py).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include "../../include/renet_hip.h"

namespace {

// ---- the deterministic split-K factors of re-net_amd/renet_hip.py (auto_split_k / auto_split_k_planes), restated -------
int auto_split_k(int m, int n, int k) {
    const long tiles = (long)((m + 127) / 128) * ((n + 127) / 128);
    const int ktiles = (k + 31) / 32;
    if (tiles >= 256 || ktiles < 8) return 1;
    const int smax = std::max(1, std::min(128, ktiles / 4));
    const double per_slice = (double)m * (double)n * 2.7e-6;
    int best = 1;
    double best_t = -1.0;
    for (int s = 1; s <= smax; ++s) {
        const long rounds = (tiles * s + 511) / 512;
        const double t = (double)rounds * ((double)ktiles / (double)s + 6.0) + (s > 1 ? per_slice * s : 0.0);
        if (best_t < 0.0 || t < best_t - 1e-9) { best = s; best_t = t; }
    }
    return best;
}

int auto_split_k_planes(int m, int n, int k) {
    const long tiles = (long)((m + 255) / 256) * ((n + 127) / 128);
    const int ktiles = (k + 31) / 32;
    if (tiles >= 128 || ktiles < 8) return 1;
    const int smax = std::max(1, std::min(64, ktiles / 4));
    const double per_slice = (double)m * (double)n * 2.7e-6;
    int best = 1;
    double best_t = -1.0;
    for (int s = 1; s <= smax; ++s) {
        const long rounds = (tiles * s + 255) / 256;
        const double t = (double)rounds * ((double)ktiles / (double)s + 6.0) + (s > 1 ? per_slice * s : 0.0);
        if (best_t < 0.0 || t < best_t - 1e-9) { best = s; best_t = t; }
    }
    return best;
}

inline size_t up(size_t x) { return (x + 255) & ~(size_t)255; }
inline size_t fbytes(size_t rows, size_t cols) { return up(rows * cols * sizeof(float)); }

// ---- workspace layout: byte offsets; every region 256-byte aligned; no region is reused inside a step ------------------
struct Lay {
    // activations the backward pass reads
    size_t out1, out2, X, Xr, sv0, sv1, feat1, feat2, dl1, dl2, featp1;
    // forward temporaries
    size_t ew, gi0, gi1, hs0, hs1, logits, featp;
    // backward temporaries
    size_t dfeat1, dfeat2, da1, da2, dh1, dh2, dc1, dgi0, dgi1, dgh0, dgh1, dX, dXr, d_rows, d_ent_seq, d_rel_seq, d_h2, gn2,
        gl2, dhN, gn1, gl1, dh1N, gs;
    // per-stream scratch
    size_t gru_ws, gemm_ws_main, gemm_ws_side, bwdw_ws, colsum_ws;
    size_t gru_ws_bytes, gemm_ws_bytes, bwdw_ws_bytes, colsum_ws_bytes;
    size_t dl_plane, dl_ld, dl_rows, featp_plane, featp_ld, featp1_plane, featp1_ld, logits_ld;
    size_t total;
};

bool use_planes(const RenetStepModel& m) { return m.lin_w_planes != nullptr; }

Lay layout(const RenetStepModel& m, const RenetStepBatch& b) {
    Lay L{};
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += up(bytes); return at; };
    const size_t D = m.D, N = b.N, nA = b.nA, S = b.S, B = b.B, NE = m.num_ent, C2 = m.C2;
    const bool pl = use_planes(m);
    L.out1 = take(fbytes(N, D));
    L.out2 = take(fbytes(nA, D));
    L.X = take(fbytes(S, 4 * D));
    L.Xr = take(fbytes(S, 3 * D));
    L.sv0 = take(fbytes(S, 5 * D));
    L.sv1 = take(fbytes(S, 5 * D));
    L.feat1 = take(fbytes(B, 3 * D));
    L.feat2 = take(fbytes(B, 2 * D));
    L.dl2 = take(fbytes(B, C2));
    if (pl) {
        L.dl_rows = (B + 255) & ~(size_t)255;
        L.dl_ld = (NE + 255) & ~(size_t)255;
        L.dl_plane = L.dl_rows * L.dl_ld;
        L.dl1 = take(3 * L.dl_plane * 2);
        L.featp_ld = (3 * D + 255) & ~(size_t)255;
        L.featp_plane = L.dl_rows * L.featp_ld;
        L.featp1_ld = (3 * D + 1 + 255) & ~(size_t)255;
        L.featp1_plane = L.dl_rows * L.featp1_ld;
        L.featp1 = take(3 * L.featp1_plane * 2);
        L.logits_ld = (NE + 3) & ~(size_t)3;
        L.logits = take(fbytes(B, L.logits_ld));
    } else {
        L.dl1 = take(fbytes(B, NE));
        L.logits_ld = NE;
    }
    L.ew = take(fbytes(NE, D));
    L.gi0 = take(fbytes(S, 3 * D));
    L.gi1 = take(fbytes(S, 3 * D));
    L.hs0 = take(fbytes(B, D));
    L.hs1 = take(fbytes(B, D));
    L.dfeat1 = take(fbytes(B, 3 * D));
    L.dfeat2 = take(fbytes(B, 2 * D));
    L.da1 = take(fbytes(B, D));
    L.da2 = take(fbytes(B, D));
    L.dh1 = take(fbytes(B, D));
    L.dh2 = take(fbytes(B, D));
    L.dc1 = take(fbytes(B, D));
    L.dgi0 = take(fbytes(S, 3 * D));
    L.dgi1 = take(fbytes(S, 3 * D));
    L.dgh0 = take(fbytes(S, 3 * D));
    L.dgh1 = take(fbytes(S, 3 * D));
    L.dX = take(fbytes(S, 4 * D));
    L.dXr = take(fbytes(S, 3 * D));
    L.d_rows = take(fbytes(S, D));
    L.d_ent_seq = take(fbytes(B, D));
    L.d_rel_seq = take(fbytes(B, D));
    L.d_h2 = take(fbytes(nA, D));
    L.gn2 = take(fbytes(nA, D));
    L.gl2 = take(fbytes(nA, D));
    L.dhN = take(fbytes(N, D));
    L.gn1 = take(fbytes(N, D));
    L.gl1 = take(fbytes(N, D));
    L.dh1N = take(fbytes(N, D));
    L.gs = take(fbytes(NE, D));
    const int bmax = b.L > 0 ? b.step_off_host[1] - b.step_off_host[0] : 0;
    L.gru_ws_bytes = 2 * renet_gru_workspace(bmax, m.D);
    L.gru_ws = take(std::max<size_t>(L.gru_ws_bytes, 4));
    // split-K partials: the largest need of any GEMM of the step, once per stream
    size_t need = 4;
    auto g_ = [&](int mm, int nn, int kk) { need = std::max(need, renet_gemm_workspace(mm, nn, auto_split_k(mm, nn, kk))); };
    const int Di = m.D, Si = b.S, Bi = b.B, NEi = m.num_ent, nAi = b.nA, C2i = m.C2;
    g_(NEi, Di, Di); g_(nAi, Di, Di); g_(Si, 3 * Di, 4 * Di); g_(Si, 3 * Di, 3 * Di); g_(Bi, NEi, 3 * Di); g_(Bi, C2i, 2 * Di);
    g_(Bi, 3 * Di, NEi); g_(NEi, 3 * Di, Bi); g_(Bi, 2 * Di, C2i); g_(C2i, 2 * Di, Bi);
    g_(3 * Di, 4 * Di, Si); g_(3 * Di, 3 * Di, Si); g_(3 * Di, Di, Si); g_(Si, 3 * Di, 3 * Di); g_(Si, 2 * Di, 3 * Di);
    g_(Di, Di, nAi); g_(Di, Di, NEi);
    if (pl) {
        need = std::max(need, renet_gemm_workspace(Bi, 3 * Di, auto_split_k_planes(Bi, 3 * Di, NEi)));
        need = std::max(need, renet_gemm_workspace(NEi, 3 * Di + 1, auto_split_k_planes(NEi, 3 * Di + 1, Bi)));
        need = std::max(need, renet_gemm_workspace(Bi, NEi, auto_split_k_planes(Bi, NEi, 3 * Di)));
    }
    L.gemm_ws_bytes = need;
    L.gemm_ws_main = take(need);
    L.gemm_ws_side = take(need);
    L.bwdw_ws_bytes = std::max<size_t>(renet_rgcn_bwd_w_workspace(std::max(b.n_chunks, b.n_chunks2), m.D), 4);
    L.bwdw_ws = take(L.bwdw_ws_bytes);
    size_t cs = 4;
    cs = std::max(cs, renet_colsum_workspace(Bi, NEi));
    cs = std::max(cs, renet_colsum_workspace(Bi, C2i));
    cs = std::max(cs, renet_colsum_workspace(Si, 3 * Di));
    L.colsum_ws_bytes = cs;
    L.colsum_ws = take(cs);
    L.total = o;
    return L;
}

bool args_ok(const RenetStepModel* m, const RenetStepBatch* b) {
    if (!m || !b) return false;
    if (!(m->D == 100 || m->D == 200 || m->D == 400) || m->num_ent < 1 || m->T < 2 || m->C2 < 1) return false;
    if (b->N < 1 || b->nA < 1 || b->nA > b->N || b->S < 1 || b->B < 1 || b->L < 1 || !b->step_off_host) return false;
    // 32-bit buffer offsets of the item kernels / table layer (the Python path falls back to other kernels beyond this)
    const size_t big = (size_t)std::max(b->N, m->num_ent) * m->D * 4;
    return big < ((size_t)1 << 31);
}

// fork / join events: a few per host thread, created once (timing disabled), reused by every call
hipEvent_t step_event(int i) {
    thread_local std::vector<hipEvent_t> ev;
    while ((int)ev.size() <= i) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        ev.push_back(e);
    }
    return ev[i];
}

struct Streams {
    hipStream_t main, side;
    int next_ev = 0, rc = 0;
    bool two() const { return side != main; }
    void order(hipStream_t first, hipStream_t then) {            // `then` waits for everything enqueued on `first` so far
        if (!two() || rc) return;
        hipEvent_t e = step_event(next_ev++ & 7);
        if (!e) { rc = (int)hipErrorOutOfMemory; return; }
        hipError_t x = hipEventRecord(e, first);
        if (x == hipSuccess) x = hipStreamWaitEvent(then, e, 0);
        if (x != hipSuccess) rc = (int)x;
    }
    void fork() { order(main, side); }
    void join() { order(side, main); }
};

#define CK(call)                      \
    do {                              \
        const int rc__ = (call);      \
        ++launches;                   \
        if (rc__ != 0) return rc__;   \
    } while (0)

// C = A op(B) (+ bias) (+ beta C) in the default fp32-class mode with the automatic split-K factor: renet_hip.gemm()
int gemm(hipStream_t st, float* ws, size_t ws_bytes, int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B,
         int ldb, float beta, float* C, int ldc, const float* bias) {
    const int sk = auto_split_k(M, N, K);
    return renet_gemm_f32_split(ta, tb, M, N, K, 1.f, A, lda, B, ldb, beta, C, ldc, bias, sk, sk > 1 ? ws : nullptr,
                                sk > 1 ? ws_bytes : 0, st);
}

}  // namespace

extern "C" {

size_t renet_step_workspace(const RenetStepModel* m, const RenetStepBatch* b) {
    if (!args_ok(m, b)) return 0;
    return layout(*m, *b).total;
}

int renet_step_forward(const RenetStepModel* mp, const RenetStepBatch* bp, const RenetStepRun* r, float* row_loss,
                       int* n_launches) {
    if (!args_ok(mp, bp) || !r || !r->workspace || !row_loss) return RENET_ERR_BADARG;
    const RenetStepModel& m = *mp;
    const RenetStepBatch& b = *bp;
    const Lay L = layout(m, b);
    if (r->workspace_bytes < L.total) return RENET_ERR_WORKSPACE;
    char* W = (char*)r->workspace;
    auto F = [&](size_t off) { return (float*)(W + off); };
    Streams s{(hipStream_t)r->stream, (hipStream_t)(r->side_stream ? r->side_stream : r->stream)};
    int launches = 0;
    const int D = m.D, N = b.N, nA = b.nA, S = b.S, B = b.B, NE = m.num_ent, T = m.T;
    const float p = m.drop_p;
    const bool pruned = nA < N;
    float* wsm = F(L.gemm_ws_main);
    float* wss = F(L.gemm_ws_side);

    // ---- ops.RGCNTableLayerFn.forward (RGCN.py:33-51,79-94 of layer 1 on h0 = ent_embeds[id], utils.py:239)
    CK(gemm(s.main, wsm, L.gemm_ws_bytes, 0, 0, NE, D, D, m.ent, D, m.loop1, D, 0.f, F(L.ew), D, nullptr));
    CK(renet_rgcn_gather_items_table(m.ent, NE, D, b.it_src_t, b.it_type_t, b.grp_ptr, b.n_groups, b.row_ptr, b.col_t, b.etype,
                                     b.node_ent, b.norm, m.w1, T, 0, F(L.ew), p, r->seed_rgcn1, 1, F(L.out1), N,
                                     b.n_heavy ? b.heavy_rows : nullptr, b.n_heavy, s.main));
    // ---- ops.RGCNLayerFn.forward, layer 2 on the row prefix [0, nA) (Aggregator.py:139-140 reads only the subject rows)
    CK(gemm(s.main, wsm, L.gemm_ws_bytes, 0, 0, nA, D, D, F(L.out1), D, m.loop2, D, 0.f, F(L.out2), D, nullptr));
    CK(renet_rgcn_gather_items(F(L.out1), D, b.it_src, b.it_type, b.grp_ptr, pruned ? b.n_groups_out : b.n_groups, b.row_ptr,
                               b.col, b.etype, b.norm, m.w2, T, 0, 0, F(L.out2), p, r->seed_rgcn2, 0, F(L.out2), nA,
                               pruned ? (b.n_heavy_out ? b.heavy_rows_out : nullptr) : (b.n_heavy ? b.heavy_rows : nullptr),
                               pruned ? b.n_heavy_out : b.n_heavy, 0, 0, pruned ? 1 : 0, s.main));
    // ---- ops.SeqAssembleFn.forward (Aggregator.py:139-165)
    CK(renet_seq_assemble_fwd(F(L.out2), m.ent, m.rel, m.glob, b.subj_row, b.row_ent, b.row_rel, b.glob_row, S, D, p, r->seed_x,
                              r->seed_xr, F(L.X), F(L.Xr), s.main));
    // ---- ops.MultiGRUFn.forward (model.py:86,94): encoder_r's input projection on the side stream
    s.fork();
    CK(gemm(s.side, wss, L.gemm_ws_bytes, 0, 1, S, 3 * D, 3 * D, F(L.Xr), 3 * D, m.wih_r, 3 * D, 0.f, F(L.gi1), 3 * D, m.bih_r));
    CK(gemm(s.main, wsm, L.gemm_ws_bytes, 0, 1, S, 3 * D, 4 * D, F(L.X), 4 * D, m.wih, 4 * D, 0.f, F(L.gi0), 3 * D, m.bih));
    s.join();
    {
        const float* gi[2] = {F(L.gi0), F(L.gi1)};
        const int32_t* so[2] = {b.step_off_host, b.step_off_host};
        const int ls[2] = {b.L, b.L};
        const float* whh[2] = {m.whh, m.whh_r};
        const float* bhh[2] = {m.bhh, m.bhh_r};
        float* hl[2] = {F(L.hs0), F(L.hs1)};
        const int bmax = b.step_off_host[1] - b.step_off_host[0];
        const int rows[2] = {std::max(B, bmax), std::max(B, bmax)};
        float* sv[2] = {F(L.sv0), F(L.sv1)};
        if (rows[0] != B) return RENET_ERR_BADARG;
        CK(renet_gru_fwd_layouts(2, gi, so, ls, D, whh, bhh, hl, rows, sv, F(L.gru_ws), L.gru_ws_bytes, s.main));
    }
    // ---- ops.DualHeadCEFn.forward (model.py:89-103): the relation head on the side stream
    s.fork();
    CK(renet_concat3_fwd(m.ent, b.s_idx, F(L.hs1), nullptr, nullptr, B, D, p, r->seed_head2, F(L.feat2), s.side));
    CK(gemm(s.side, wss, L.gemm_ws_bytes, 0, 1, B, m.C2, 2 * D, F(L.feat2), 2 * D, m.linr_w, 2 * D, 0.f, F(L.dl2), m.C2, m.linr_b));
    CK(renet_softmax_ce(F(L.dl2), b.rel_label, B, m.C2, m.C2, r->scale_rel, row_loss + B, F(L.dl2), s.side));
    CK(renet_concat3_fwd(m.ent, b.s_idx, F(L.hs0), m.rel, b.r_idx, B, D, p, r->seed_head1, F(L.feat1), s.main));
    if (use_planes(m)) {
        // ONE split of feat, with the ones column the weight-gradient GEMM wants (its bias column): the logits GEMM reads
        // the same planes with K = 3D -- the ones meet the zero k padding of the weight's planes and add exactly 0
        CK(renet_pack_planes(F(L.feat1), B, 3 * D, 3 * D, 1, W + L.featp1, s.main));
        const int sk = auto_split_k_planes(B, NE, 3 * D);
        CK(renet_gemm_planes(0, 0, B, NE, 3 * D, 1.f, nullptr, W + L.featp1, (int)L.featp1_ld, L.featp1_plane, m.lin_w_planes,
                             m.lin_w_ld, m.lin_w_plane, 0.f, F(L.logits), (int)L.logits_ld, m.lin_b, nullptr, sk,
                             sk > 1 ? wsm : nullptr, sk > 1 ? L.gemm_ws_bytes : 0, s.main));
        CK(renet_softmax_ce_planes(F(L.logits), b.ent_label, B, NE, (int)L.logits_ld, r->scale_ent, row_loss, W + L.dl1, L.dl_plane,
                                   (int)L.dl_ld, (int)L.dl_rows, s.main));
    } else {
        CK(gemm(s.main, wsm, L.gemm_ws_bytes, 0, 1, B, NE, 3 * D, F(L.feat1), 3 * D, m.lin_w, 3 * D, 0.f, F(L.dl1), NE, m.lin_b));
        CK(renet_softmax_ce(F(L.dl1), b.ent_label, B, NE, NE, r->scale_ent, row_loss, F(L.dl1), s.main));
    }
    s.join();
    if (n_launches) *n_launches = launches;
    return s.rc;
}

int renet_step_backward(const RenetStepModel* mp, const RenetStepBatch* bp, const RenetStepRun* r, const float* g,
                        int defer_side, void* ev_head_done, void* ev_gru_done, int* n_launches) {
    if (!args_ok(mp, bp) || !r || !r->workspace || !g) return RENET_ERR_BADARG;
    const RenetStepModel& m = *mp;
    const RenetStepBatch& b = *bp;
    const Lay L = layout(m, b);
    if (r->workspace_bytes < L.total) return RENET_ERR_WORKSPACE;
    char* W = (char*)r->workspace;
    auto F = [&](size_t off) { return (float*)(W + off); };
    Streams s{(hipStream_t)r->stream, (hipStream_t)(r->side_stream ? r->side_stream : r->stream)};
    int launches = 0;
    const int D = m.D, N = b.N, nA = b.nA, S = b.S, B = b.B, NE = m.num_ent, T = m.T, C2 = m.C2;
    const float p = m.drop_p;
    const bool pruned = nA < N;
    const bool pl = use_planes(m);
    float* wsm = F(L.gemm_ws_main);
    float* wss = F(L.gemm_ws_side);
    float* csw = L.colsum_ws_bytes > 4 ? F(L.colsum_ws) : nullptr;
    auto colsum = [&](const float* X, int M_, int N_, float* out, hipStream_t st) {
        const size_t need = renet_colsum_workspace(M_, N_);
        return renet_colsum(X, M_, N_, N_, out, 1.f, need ? csw : nullptr, need, st);
    };

    // ---- ops.DualHeadCEFn.backward: every gradient of a head is linear in its CE gradient; the upstream scalar g is folded
    // into it once (in-loop-split head) or into the planes GEMMs' alpha (planes head)
    if (!pl) CK(renet_scale_by_device_scalar(F(L.dl1), (size_t)B * NE, g, s.main));
    s.fork();
    CK(renet_scale_by_device_scalar(F(L.dl2), (size_t)B * C2, g, s.side));
    CK(gemm(s.side, wss, L.gemm_ws_bytes, 0, 0, B, 2 * D, C2, F(L.dl2), C2, m.linr_w, 2 * D, 0.f, F(L.dfeat2), 2 * D, nullptr));
    CK(gemm(s.side, wss, L.gemm_ws_bytes, 1, 0, C2, 2 * D, B, F(L.dl2), C2, F(L.feat2), 2 * D, 1.f, m.g_linr_w, 2 * D, nullptr));
    CK(colsum(F(L.dl2), B, C2, m.g_linr_b, s.side));
    CK(renet_concat3_bwd(F(L.dfeat2), B, D, 2, p, r->seed_head2, F(L.da2), F(L.dh2), nullptr, s.side));
    if (pl) {
        const int sk1 = auto_split_k_planes(B, 3 * D, NE);
        CK(renet_gemm_planes(0, 1, B, 3 * D, NE, 1.f, g, W + L.dl1, (int)L.dl_ld, L.dl_plane, m.lin_w_planes, m.lin_w_ld,
                             m.lin_w_plane, 0.f, F(L.dfeat1), 3 * D, nullptr, nullptr, sk1, sk1 > 1 ? wsm : nullptr,
                             sk1 > 1 ? L.gemm_ws_bytes : 0, s.main));
        const int sk2 = auto_split_k_planes(NE, 3 * D + 1, B);
        CK(renet_gemm_planes(1, 1, NE, 3 * D + 1, B, 1.f, g, W + L.dl1, (int)L.dl_ld, L.dl_plane, W + L.featp1, (int)L.featp1_ld,
                             L.featp1_plane, 1.f, m.g_lin_w, 3 * D, nullptr, m.g_lin_b, sk2, sk2 > 1 ? wsm : nullptr,
                             sk2 > 1 ? L.gemm_ws_bytes : 0, s.main));
    } else {
        CK(colsum(F(L.dl1), B, NE, m.g_lin_b, s.side));              // (bias column sum next to the matrix-bound GEMMs)
        CK(gemm(s.main, wsm, L.gemm_ws_bytes, 0, 0, B, 3 * D, NE, F(L.dl1), NE, m.lin_w, 3 * D, 0.f, F(L.dfeat1), 3 * D, nullptr));
        CK(gemm(s.main, wsm, L.gemm_ws_bytes, 1, 0, NE, 3 * D, B, F(L.dl1), NE, F(L.feat1), 3 * D, 1.f, m.g_lin_w, 3 * D, nullptr));
    }
    CK(renet_concat3_bwd(F(L.dfeat1), B, D, 3, p, r->seed_head1, F(L.da1), F(L.dh1), F(L.dc1), s.main));
    s.join();
    if (ev_head_done) {                                                  // linear.weight / linear.bias gradients are complete
        const hipError_t e = hipEventRecord((hipEvent_t)ev_head_done, s.main);
        if (e != hipSuccess) return (int)e;
    }
    CK(renet_add_inplace(F(L.da1), F(L.da2), (size_t)B * D, s.main));
    CK(renet_segment_add(F(L.da1), b.plan_s.order, b.plan_s.seg_ptr, b.plan_s.target, b.plan_s.num_segments, D, m.g_ent, s.main));
    CK(renet_segment_add(F(L.dc1), b.plan_r.order, b.plan_r.seg_ptr, b.plan_r.target, b.plan_r.num_segments, D, m.g_rel, s.main));

    // ---- ops.MultiGRUFn.backward: the recurrences and dX on this stream, every parameter gradient on the side stream
    {
        const float* dhl[2] = {F(L.dh1), F(L.dh2)};
        const int32_t* so[2] = {b.step_off_host, b.step_off_host};
        const int ls[2] = {b.L, b.L};
        const float* whh[2] = {m.whh, m.whh_r};
        const float* sv[2] = {F(L.sv0), F(L.sv1)};
        float* dgi[2] = {F(L.dgi0), F(L.dgi1)};
        float* dgh[2] = {F(L.dgh0), F(L.dgh1)};
        CK(renet_gru_bwd_layouts(2, dhl, so, ls, D, whh, sv, dgi, dgh, F(L.gru_ws), L.gru_ws_bytes, s.main));
    }
    // The parameter gradients of both encoders (~0.4 ms of chip-filling GEMMs that feed only the optimizer) run on the side
    // stream; RENET_STEP_LATE_FORK=1 forks them behind the dX GEMMs instead of in front (A/B: profiles/r06_d_step_plan.md).
    static int late_fork = -1;
    if (late_fork < 0) {
        const char* e_ = getenv("RENET_STEP_LATE_FORK");
        late_fork = (e_ && e_[0] == '1') ? 1 : 0;
    }
    auto gru_param_grads = [&]() -> int {
        const float* xs[2] = {F(L.X), F(L.Xr)};
        const int in[2] = {4 * D, 3 * D};
        const size_t dgi[2] = {L.dgi0, L.dgi1}, dgh[2] = {L.dgh0, L.dgh1}, sv[2] = {L.sv0, L.sv1};
        float* gwih[2] = {m.g_wih, m.g_wih_r};
        float* gwhh[2] = {m.g_whh, m.g_whh_r};
        float* gbih[2] = {m.g_bih, m.g_bih_r};
        float* gbhh[2] = {m.g_bhh, m.g_bhh_r};
        for (int k = 0; k < 2; ++k) {
            CK(gemm(s.side, wss, L.gemm_ws_bytes, 1, 0, 3 * D, in[k], S, F(dgi[k]), 3 * D, xs[k], in[k], 1.f, gwih[k], in[k], nullptr));
            CK(gemm(s.side, wss, L.gemm_ws_bytes, 1, 0, 3 * D, D, S, F(dgh[k]), 3 * D, F(sv[k]) + 4 * D, 5 * D, 1.f, gwhh[k], D, nullptr));
            CK(colsum(F(dgi[k]), S, 3 * D, gbih[k], s.side));
            CK(colsum(F(dgh[k]), S, 3 * D, gbhh[k], s.side));
        }
        return 0;
    };
    auto gru_done = [&]() -> int {                     // both encoders' parameter gradients are complete (third all-reduce bucket)
        if (!ev_gru_done) return 0;
        return (int)hipEventRecord((hipEvent_t)ev_gru_done, s.side);
    };
    if (!late_fork) {
        s.fork();
        int rc_ = gru_param_grads();
        if (!rc_) rc_ = gru_done();
        if (rc_) return rc_;
    }
    // dX: only the columns the sequence assembly reads (the trailing D columns are the constant global embedding)
    CK(gemm(s.main, wsm, L.gemm_ws_bytes, 0, 0, S, 3 * D, 3 * D, F(L.dgi0), 3 * D, m.wih, 4 * D, 0.f, F(L.dX), 4 * D, nullptr));
    CK(gemm(s.main, wsm, L.gemm_ws_bytes, 0, 0, S, 2 * D, 3 * D, F(L.dgi1), 3 * D, m.wih_r, 3 * D, 0.f, F(L.dXr), 3 * D, nullptr));
    if (late_fork) {
        s.fork();
        int rc_ = gru_param_grads();
        if (!rc_) rc_ = gru_done();
        if (rc_) return rc_;
    }

    // ---- ops.SeqAssembleFn.backward
    CK(renet_seq_assemble_bwd(F(L.dX), F(L.dXr), b.step_off, b.L, S, B, D, p, r->seed_x, r->seed_xr, F(L.d_rows), F(L.d_ent_seq),
                              F(L.d_rel_seq), s.main));
    CK(renet_zero(F(L.d_h2), (size_t)nA * D, s.main));
    CK(renet_segment_add(F(L.d_rows), b.plan_subj_row.order, b.plan_subj_row.seg_ptr, b.plan_subj_row.target,
                         b.plan_subj_row.num_segments, D, F(L.d_h2), s.main));
    CK(renet_segment_add(F(L.d_ent_seq), b.plan_s.order, b.plan_s.seg_ptr, b.plan_s.target, b.plan_s.num_segments, D, m.g_ent, s.main));
    CK(renet_segment_add(F(L.d_rel_seq), b.plan_r.order, b.plan_r.seg_ptr, b.plan_r.target, b.plan_r.num_segments, D, m.g_rel, s.main));

    // ---- ops.RGCNLayerFn.backward (layer 2, evaluated on the row prefix)
    const int pair_shift = (T / 2) % T;
    CK(renet_rgcn_bwd_prep(F(L.d_h2), F(L.out2), b.norm, 0, p, r->seed_rgcn2, nA, D, F(L.gn2), F(L.gl2), s.main));
    CK(renet_rgcn_gather_items(F(L.gn2), D, b.it_src, b.it_type, b.grp_ptr, b.n_groups, b.row_ptr, b.col, b.etype, nullptr, m.w2, T,
                               pair_shift, 1, nullptr, 0.f, 0, 0, F(L.dhN), N, b.n_heavy ? b.heavy_rows : nullptr, b.n_heavy,
                               pruned ? nA : 0, 0, pruned ? 1 : 0, s.main));
    CK(gemm(s.main, wsm, L.gemm_ws_bytes, 0, 1, nA, D, D, F(L.gl2), D, m.loop2, D, 1.f, F(L.dhN), D, nullptr));
    s.fork();
    if (pruned)
        CK(renet_rgcn_bwd_w(F(L.out1), F(L.gn2), b.e_src2, b.e_dst2, b.chunk_ptr2, b.chunk_type2, b.n_chunks2, b.type_chunk_ptr2, T,
                            0, D, m.g_w2, 1.f, F(L.bwdw_ws), renet_rgcn_bwd_w_workspace(b.n_chunks2, D), s.side));
    else
        CK(renet_rgcn_bwd_w(F(L.out1), F(L.gn2), b.e_src, b.e_dst, b.chunk_ptr, b.chunk_type, b.n_chunks, b.type_chunk_ptr, T, 0, D,
                            m.g_w2, 1.f, F(L.bwdw_ws), renet_rgcn_bwd_w_workspace(b.n_chunks, D), s.side));
    CK(gemm(s.side, wss, L.gemm_ws_bytes, 1, 0, D, D, nA, F(L.out1), D, F(L.gl2), D, 1.f, m.g_loop2, D, nullptr));

    // ---- ops.RGCNTableLayerFn.backward (layer 1): transposed gather on [N, D], then ONE reduction to entity rows
    CK(renet_rgcn_bwd_prep(F(L.dhN), F(L.out1), b.norm, 1, p, r->seed_rgcn1, N, D, F(L.gn1), F(L.gl1), s.main));
    CK(renet_rgcn_gather_items(F(L.gn1), D, b.it_src, b.it_type, b.grp_ptr, b.n_groups, b.row_ptr, b.col, b.etype, nullptr, m.w1, T,
                               pair_shift, 1, nullptr, 0.f, 0, 0, F(L.dh1N), N, b.n_heavy ? b.heavy_rows : nullptr, b.n_heavy, 0, 0,
                               0, s.main));
    s.fork();
    CK(renet_rgcn_bwd_w(m.ent, F(L.gn1), b.e_src_t, b.e_dst, b.chunk_ptr, b.chunk_type, b.n_chunks, b.type_chunk_ptr, T, 0, D,
                        m.g_w1, 1.f, F(L.bwdw_ws), renet_rgcn_bwd_w_workspace(b.n_chunks, D), s.side));
    CK(renet_zero(F(L.gs), (size_t)NE * D, s.main));
    CK(renet_segment_add2(F(L.dh1N), F(L.gl1), b.plan_node_ent.order, b.plan_node_ent.seg_ptr, b.plan_node_ent.target,
                          b.plan_node_ent.num_segments, D, m.g_ent, F(L.gs), s.main));
    CK(gemm(s.main, wsm, L.gemm_ws_bytes, 0, 1, NE, D, D, F(L.gs), D, m.loop1, D, 1.f, m.g_ent, D, nullptr));
    CK(gemm(s.main, wsm, L.gemm_ws_bytes, 1, 0, D, D, NE, m.ent, D, F(L.gs), D, 1.f, m.g_loop1, D, nullptr));
    if (!defer_side) s.join();
    if (n_launches) *n_launches = launches;
    return s.rc;
}

}  // extern "C"
