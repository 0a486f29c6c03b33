/*
 * renet_hip.h -- C ABI of librenet_hip.so: the MI355X (gfx950) kernels behind RE-Net's
 * RENet / RGCNAggregator Python API.
 *
 * The reference (INK-USC/RE-Net) has NO native code and NO FFI: its boundary for this path is the
 * Python class API (RGCN.py, Aggregator.py, model.py, global_model.py) and everything below that is
 * third-party (DGL 0.4 + torch 1.6 CUDA kernels).  Each entry point below therefore names the
 * reference lines whose device work it replaces.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch-ROCm tensors: tensor.data_ptr());
 *     the library allocates nothing and keeps no global mutable state (re-entrant);
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); every
 *     call only enqueues work on it and never synchronises;
 *   - fp32 row-major contiguous matrices, int32 indices;
 *   - D (n_hidden) must be 100, 200 or 400 (the reference hard-codes num_bases = 100, model.py:36,
 *     so the relation blocks are 1x1, 2x2, 4x4); other values return RENET_ERR_UNSUPPORTED;
 *   - return value: 0 = ok, <0 = argument error (below), >0 = hipError_t from the launch.
 */
#ifndef RENET_HIP_H
#define RENET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RENET_OK 0
#define RENET_ERR_BADARG (-1)
#define RENET_ERR_UNSUPPORTED (-2)
#define RENET_ERR_WORKSPACE (-3)

#define RENET_ABI_VERSION 1
int renet_version(void);

/* ------------------------------------------------------------------------------------------------
 * Row gather / deterministic segmented scatter-add.
 * renet_gather_rows : out[i,:] = table[idx[i],:]              (utils.py:239 h0 = ent_embeds[id];
 *                                                              Aggregator.py:139-140 subject rows)
 * renet_segment_add : for u in [0,U): dst[seg_target[u],:] += sum_{k in [seg_ptr[u],seg_ptr[u+1])}
 *                     src[order[k],:]   -- the backward of renet_gather_rows with the index sorted on
 *                     the host (no atomics => bit-reproducible).
 * ---------------------------------------------------------------------------------------------- */
int renet_gather_rows(const float* table, const int32_t* idx, int n, int D, float* out, void* stream);
int renet_segment_add(const float* src, const int32_t* order, const int32_t* seg_ptr,
                      const int32_t* seg_target, int U, int D, float* dst, void* stream);
/* Two segmented adds that share ONE plan in one launch: dst0[target[u]] += sum of src0 rows, dst1[target[u]] += sum
 * of src1 rows (the first RGCN layer's backward reduces both the message gradient and the self-loop gradient of the
 * batch graph's nodes to entity rows, keyed by the same node -> entity map). */
int renet_segment_add2(const float* src0, const float* src1, const int32_t* order, const int32_t* seg_ptr,
                       const int32_t* seg_target, int U, int D, float* dst0, float* dst1, void* stream);

/* ------------------------------------------------------------------------------------------------
 * RGCN block-diagonal gather-SpMM  (RGCN.py:79-94 msg_func + fn.sum + apply_func, fused with the
 * epilogue of RGCN.py:42-50).
 *
 *   out[v,:] = act( scale[v] * sum_{e in [row_ptr[v],row_ptr[v+1])} blockmul(x[col[e],:], W[tau(e)])
 *                   + addend[v,:] * dropmask )
 *   tau(e)   = (etype[e] + type_shift) mod T                       (T = 2 * num_rels)
 *   blockmul : x viewed [nb=100, si], W[tau] viewed [nb, si, so] row-major (RGCN.py:81-87):
 *              transpose_w == 0: y[b,j] = sum_i x[b,i] W[b,i,j]     (forward message)
 *              transpose_w == 1: y[b,i] = sum_j x[b,j] W[b,i,j]     (backward wrt h through W^T)
 *   scale    : per-destination 1/in-degree (`norm`, utils.py:126-127) or NULL (= 1)
 *   addend   : the self-loop message h @ W_loop (RGCN.py:34-37) or NULL; may alias `out`.
 *              drop_p > 0 applies inverted dropout to the addend with the counter-based mask
 *              keep(seed, v*D+c) (RGCN.py:36-37); the same (seed) regenerates the mask in backward.
 *   relu     : 1 = ReLU epilogue (rgcn1, Aggregator.py:119-120), 0 = identity (rgcn2)
 *
 * The graph is CSR by destination (row_ptr[N+1], col[E] = source row, etype[E]).
 * heavy_rows[n_heavy] (optional, may be NULL/0) lists the rows whose in-degree exceeds heavy_thresh:
 * the row-group kernel skips them and a second launch reduces each with a whole workgroup, so a
 * Zipf-tail hub row cannot serialise one wave for the entire launch.
 * src_limit > 0 skips edges whose source row is >= src_limit and addend_rows > 0 restricts the addend to
 * rows < addend_rows (0 = no restriction): used by the backward of a layer that was only evaluated on a
 * prefix of the rows (RE-Net reads the second RGCN layer only at the subject rows, Aggregator.py:139-140).
 * The backward wrt x uses the same CSR: RE-Net graphs hold both directions of every fact with paired
 * types (utils.py:74-76), so the transposed graph is the same structure with type_shift = num_rels.
 * ---------------------------------------------------------------------------------------------- */
int renet_rgcn_gather(const float* x, int D, const int32_t* row_ptr, const int32_t* col,
                      const int32_t* etype, const float* scale, const float* W, int T, int type_shift,
                      int transpose_w, const float* addend, float drop_p, uint64_t seed, int relu,
                      float* out, int N, const int32_t* heavy_rows, int n_heavy, int heavy_thresh,
                      int src_limit, int addend_rows, void* stream);

/* The same operator on the planned ITEM STREAM of the graph (the production path; renet_rgcn_gather above needs no
 * plan and stays as the plain-CSR entry).  The host planner (renet_host_gather_items / graph.plan_gather_items)
 * linearises every LIGHT row (in-degree <= heavy) as its in-edges (it_src = source row, it_type = type_s) followed
 * by a flush item (it_src = the row, it_type = -1) and cuts the stream into n_groups groups of <= 64 items
 * (grp_ptr[n_groups + 1]); one wave takes one group, all of its loads are unconditional buffer loads UNR items at a
 * time (the self-loop addend and norm of a row are loaded by its flush item like any edge's operands), hub rows
 * (heavy_rows[n_heavy], in-degree > heavy, NOT in the stream) are reduced by one workgroup each in the same
 * launch.  row_ptr / col / etype are only read for hub rows.  pruned != 0 only selects the kernel NAME
 * (rgcn_gather_{fwd,bwdh}_{full,pruned}: a kernel trace separates the four launch classes of a training step);
 * the caller passes the group / hub-row prefix that covers rows < N.  x, addend, out, W must each be smaller than 2 GiB (32-bit buffer offsets).
 * Replaces RGCN.py:79-94 + 42-50 exactly as renet_rgcn_gather does. */
int renet_rgcn_gather_items(const float* x, int D, const int32_t* it_src, const int32_t* it_type,
                            const int32_t* grp_ptr, int n_groups, const int32_t* row_ptr, const int32_t* col,
                            const int32_t* etype, const float* scale, const float* W, int T, int type_shift,
                            int transpose_w, const float* addend, float drop_p, uint64_t seed, int relu,
                            float* out, int N, const int32_t* heavy_rows, int n_heavy, int src_limit,
                            int addend_rows, int pruned, void* stream);

/* The FIRST RGCN layer addressed through the entity table (utils.py:239 + RGCN.py:35,79-94): the reference
 * materialises h0 = ent_embeds[id] ([N, D], one row per node of the batch graph) and multiplies it by W_loop; since
 * h0 @ W_loop == (ent_embeds @ W_loop)[id], the layer can read BOTH its source rows and its self-loop addend from
 * [table_rows, D] tables (ent_embeds and ent_embeds @ W_loop, computed once per step on N_ent rows instead of N):
 *   out[v] = act( scale[v] * sum_{e=(u->v)} blockmul(table[row_map[u]], W[type_e]) + drop(addend_table[row_map[v]]) )
 * it_src_t / it_type_t / col_t are the item stream / CSR columns with the source rows already composed through
 * row_map (renet_compose_table_items); flush items carry row_map[v] inside it_type (-3 - row_map[v]).  Forward
 * only (the backward pass of the layer is the ordinary transposed gather on [N, D] gradients). */
int renet_rgcn_gather_items_table(const float* table, int table_rows, int D, const int32_t* it_src_t,
                                  const int32_t* it_type_t, const int32_t* grp_ptr, int n_groups,
                                  const int32_t* row_ptr, const int32_t* col_t, const int32_t* etype,
                                  const int32_t* row_map, const float* scale, const float* W, int T, int type_shift,
                                  const float* addend_table, float drop_p, uint64_t seed, int relu, float* out, int N,
                                  const int32_t* heavy_rows, int n_heavy, void* stream);

/* bf16-STORAGE forms of the two item-stream gathers (BASELINE config 5, bench.py --dtype bf16): the relation-block
 * table W_bf16 [T, w_ld] (row stride w_ld >= D * D/100 bf16 elements; a renet_pack_bf16 matrix) and -- table form --
 * the entity table table_bf16 [table_rows, table_ld] are read as bf16 and widened in registers; x / the gradient rows,
 * the addend and the output stay fp32, accumulation is fp32.  Same arguments and results otherwise (the product
 * rounds W / the table to bf16 exactly as the bf16 GEMMs of this mode do with their operands).
 * Replaces, like the fp32 forms: RGCN.py:53-77,79-94 (msg_func / bmm + fn.sum + apply_func). */
int renet_rgcn_gather_items_bf16(const float* x, int D, const int32_t* it_src, const int32_t* it_type,
                                 const int32_t* grp_ptr, int n_groups, const int32_t* row_ptr, const int32_t* col,
                                 const int32_t* etype, const float* scale, const void* W_bf16, int w_ld, int T,
                                 int type_shift, int transpose_w, const float* addend, float drop_p, uint64_t seed,
                                 int relu, float* out, int N, const int32_t* heavy_rows, int n_heavy, int src_limit,
                                 int addend_rows, int pruned, void* stream);
int renet_rgcn_gather_items_table_bf16(const void* table_bf16, int table_ld, int table_rows, int D,
                                       const int32_t* it_src_t, const int32_t* it_type_t, const int32_t* grp_ptr,
                                       int n_groups, const int32_t* row_ptr, const int32_t* col_t, const int32_t* etype,
                                       const int32_t* row_map, const float* scale, const void* W_bf16, int w_ld, int T,
                                       int type_shift, const float* addend_table, float drop_p, uint64_t seed, int relu,
                                       float* out, int N, const int32_t* heavy_rows, int n_heavy, void* stream);
/* Composes the per-batch index arrays of the table-addressed layer on the device (once per batch):
 * it_src_t / it_type_t [n_items], col_t [E] (CSR columns), e_src_t [E] (relation-bucketed edge list of
 * renet_rgcn_bwd_w) = the plain arrays with every SOURCE row u replaced by row_map[u]. */
int renet_compose_table_items(const int32_t* row_map, const int32_t* it_src, const int32_t* it_type, int n_items,
                              const int32_t* col, const int32_t* e_src, int E, int32_t* it_src_t,
                              int32_t* it_type_t, int32_t* col_t, int32_t* e_src_t, void* stream);

/* Backward prologue of one RGCN layer (element-wise, RGCN.py:42-50 + :93-94 reversed):
 *   g_pre = g_out * (relu ? out > 0 : 1);  gn = g_pre * norm[v];  g_loop = g_pre * dropmask       */
int renet_rgcn_bwd_prep(const float* g_out, const float* out, const float* norm, int relu,
                        float drop_p, uint64_t seed, int N, int D, float* gn, float* g_loop,
                        void* stream);
/* The same with the operand bound of g_loop for renet_gemm_f32_h3: bound_part receives renet_bound_parts(N * D / 4)
 * floats (one maximum of |g_loop| per workgroup). */
int renet_rgcn_bwd_prep_bounds(const float* g_out, const float* out, const float* norm, int relu,
                               float drop_p, uint64_t seed, int N, int D, float* gn, float* g_loop,
                               float* bound_part, void* stream);

/* Gradient of the relation weights (RGCN.py:81-87 backward):
 *   dW[tau, b, i, j] = sum_{e: tau(e) == tau} x[src[e], b*si+i] * gn[dst[e], b*so+j]
 * Edges are pre-sorted by type and cut into chunks of one type each (host side):
 *   chunk c covers sorted edges [chunk_ptr[c], chunk_ptr[c+1]) , all of type chunk_type[c];
 *   type_chunk_ptr[T+1] lists, per type, its range of chunks; the stored type t accumulates into
 *   dW[(t + type_shift) mod T] (same convention as renet_rgcn_gather).  Two deterministic passes:
 *   per-chunk partial sums into `workspace` (n_chunks * D*D/100 floats), then a per-type reduction.
 *   x and gn must each be smaller than 2 GiB (rows are addressed with 32-bit buffer offsets). */
size_t renet_rgcn_bwd_w_workspace(int n_chunks, int D);
int renet_rgcn_bwd_w(const float* x, const float* gn, const int32_t* e_src, const int32_t* e_dst,
                     const int32_t* chunk_ptr, const int32_t* chunk_type, int n_chunks,
                     const int32_t* type_chunk_ptr, int T, int type_shift, int D, float* dW, float beta,
                     float* workspace, size_t workspace_bytes, void* stream);
/* The same with 64-bit global addressing of the rows of x / gn: for feature tensors of 2 GiB and more, which the
 * 32-bit buffer offsets of renet_rgcn_bwd_w cannot reach (the counterpart of renet_rgcn_gather for the gather entry). */
int renet_rgcn_bwd_w64(const float* x, const float* gn, const int32_t* e_src, const int32_t* e_dst,
                       const int32_t* chunk_ptr, const int32_t* chunk_type, int n_chunks,
                       const int32_t* type_chunk_ptr, int T, int type_shift, int D, float* dW, float beta,
                       float* workspace, size_t workspace_bytes, void* stream);   /* dW = beta * dW + ... */

/* ------------------------------------------------------------------------------------------------
 * fp32 GEMM on the f32-input MFMA (exact fp32, v_mfma_f32_32x32x2_f32):
 *   C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] (+ bias[N]) (+ beta * C)
 *   ta == 0: A is [M,K] row-major (lda = K-stride);  ta == 1: A is stored [K,M] (A^T)
 *   tb == 0: B is [K,N] row-major;                    tb == 1: B is stored [N,K] (B^T)
 * Replaces torch.mm (RGCN.py:35), the GRU input projection inside nn.GRU (model.py:86,94), nn.Linear
 * (model.py:89-90,98-99) and their autograd backward GEMMs.
 * split_k > 1 runs the deterministic two-pass split-K (partials in `workspace`,
 * renet_gemm_workspace bytes) for the tall-skinny weight-gradient shapes.
 * Products are EXACT fp32 (no operand rounding: integer-valued operands give integer-exact results), accumulation is
 * fp32 in a fixed k order.  Since round 4 the kernel addresses its operands through raw buffer descriptors (32-bit byte
 * offsets, branch-free k-loop; csrc/gemm.hip); operands it cannot reach that way (>= 4 GiB) run the round-1 kernel
 * with 64-bit addressing -- same contract, same results up to summation order.
 * ---------------------------------------------------------------------------------------------- */
size_t renet_gemm_workspace(int M, int N, int split_k);
int renet_gemm_f32(int ta, int tb, int M, int N, int K, float alpha, const float* A, int lda,
                   const float* B, int ldb, float beta, float* C, int ldc, const float* bias,
                   int split_k, float* workspace, size_t workspace_bytes, void* stream);

/* Same contract as renet_gemm_f32, computed on the bf16 matrix cores with every fp32 operand split into
 * three bf16 terms and the six leading term products accumulated in fp32 ("bf16x6"): fp32-class accuracy
 * (dropped terms <= 2^-23 relative) at 16/6 of the f32-input MFMA rate.  See csrc/gemm_split.hip. */
int renet_gemm_f32_split(int ta, int tb, int M, int N, int K, float alpha, const float* A, int lda,
                         const float* B, int ldb, float beta, float* C, int ldc, const float* bias,
                         int split_k, float* workspace, size_t workspace_bytes, void* stream);

/* What renet_gemm_f32_split launches for a problem -- the launcher executes exactly this plan (csrc/gemm_split.hip,
 * plan_split); pure host logic, callable without a GPU.  A / B may be NULL (only their 16-byte alignment is looked at,
 * for the weight-resident kernel).  plan[8] (host ints):
 *   [0] kernel: 0 fused k-loop (one workgroup per CU; grids of <= 256 tiles), 1 two-phase 128 x 128 tiles, 2 two-phase
 *       256 x 128 tiles, 3 weight-resident kernel (N <= 256, K <= 208, gemm_skinny.hip)
 *   [1] two-phase kernels: 1 = operands through raw buffer descriptors (both reach < 2^30 elements), 0 = 64-bit loader
 *   [2] tile order: 0 plain (blockIdx.x, blockIdx.y); w >= 1: XCD-aware order with panels of <= w tiles (8 by default,
 *       narrowed per shape for un-split grids whose short operand does not fit an L2; split-K grids map whole k-slices to
 *       one XCD)
 *   [3..5] grid (x = column tiles, y = row tiles, z = k-slices)   [6] the split factor after clamping   [7] 0 */
int renet_gemm_split_plan(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                          int split_k, int* plan);

/* Same contract as renet_gemm_f32 on the f16 matrix cores ("f16x3", csrc/gemm_h3.h): every operand value, scaled by
 * a power of two of its tensor so that max |x| lands in [2^14, 2^15), is split into two binary16 terms
 * x s = h1 + 2^-11 h2 (22 significant bits) and a b is evaluated as a1 b1 + 2^-11 (a1 b2 + a2 b1) with fp32
 * accumulation: split error and dropped term are each <= 2^-22 relative for every element with |x| >= 2^-29 max |x|
 * (smaller elements keep an ABSOLUTE error <= 2^-39 max |x|), unbiased, so that a K-long dot product stays within a
 * small multiple of the error of fp32 products with fp32 accumulation (measured next to renet_gemm_f32 and
 * renet_gemm_f32_split in tests/test_gpu_parity.py) -- with three matrix instructions per fragment pair
 * instead of six.  maxA / maxB: nA / nB (1..1024) device floats whose largest magnitude bounds max |A| / max |B|
 * from above -- the output of renet_maxabs_partials, or any bound the caller knows (a bound 2^k too large costs k of
 * the 29 binades).  A bound that is too SMALL overflows binary16: undefined results (inf / NaN).
 *   renet_maxabs_partials : part[b], b < renet_maxabs_blocks(rows, cols, ld) <= 256 (part[0] = 0 for an empty matrix):
 *                           the maxima of |x| over disjoint parts of x[rows, cols] (row stride ld); NaNs are ignored.
 * Replaces the same reference calls as renet_gemm_f32 (torch.mm RGCN.py:35, nn.GRU's input projection model.py:86,94,
 * nn.Linear model.py:89-90,98-99 and their backward GEMMs). */
int renet_maxabs_blocks(int rows, int cols, int ld);
int renet_maxabs_partials(const float* x, int rows, int cols, int ld, float* part, void* stream);
/* renet_maxabs_partials for n_jobs CONTIGUOUS, 16-byte aligned arrays in one launch (a model's weights, once per
 * optimizer step).  jobs: device array of n_jobs records {const float* x; uint64 n4 (number of float4); float* part;
 * int32 nblocks (<= 256 partials written to part); int32 pad}. */
int renet_maxabs_partials_multi(const void* jobs, int n_jobs, void* stream);
int renet_gemm_f32_h3(int ta, int tb, int M, int N, int K, float alpha, const float* A, int lda, const float* B,
                      int ldb, float beta, float* C, int ldc, const float* bias, int split_k, float* workspace,
                      size_t workspace_bytes, const float* maxA, int nA, const float* maxB, int nB, void* stream);

/* Same contract in bf16 mixed precision (BASELINE config 5: "n_hidden=400 bf16"): every fp32 operand value is
 * rounded to bf16 (RNE) on its way into LDS, products on v_mfma_f32_32x32x16_bf16, fp32 accumulation, fp32 C.
 * Relative error of a product ~2^-8 per operand; a K-long dot product of O(1) terms is accurate to ~2^-8/sqrt(K)
 * relative to its magnitude scale (tolerances: tests/test_gpu_bf16.py). */
int renet_gemm_bf16(int ta, int tb, int M, int N, int K, float alpha, const float* A, int lda,
                    const float* B, int ldb, float beta, float* C, int ldc, const float* bias,
                    int split_k, float* workspace, size_t workspace_bytes, void* stream);

/* bf16 STORAGE (BASELINE config 5, "n_hidden=400 bf16"): operands live in HBM as bf16 matrices, [Rp][Cp] row-major
 * with Rp, Cp = R, C rounded up to multiples of 256 and zero padding.
 *   renet_pack_bf16  : fp32 X[R, C] (row stride ldx) -> bf16 (RNE), padding written; renet_bf16_bytes(R, C) bytes.
 *   renet_gemm_bf16s : C[M,N] = alpha * op(A) op(B) (+ bias) (+ beta * C) on such matrices, contraction length K: ONE bf16
 *                      product per element pair (v_mfma_f32_32x32x16_bf16), fp32 accumulation, fp32 C; operands staged by
 *                      LDS-DMA (global_load_lds) through a 3-deep LDS ring.
 *       a_tr == 0: A is stored as [M rows][K cols] (K contiguous);  a_tr == 1: A^T, i.e. [K rows][M cols] (the tensor as
 *                  stored when the contraction runs over its rows): fragments come from ds_read_b64_tr_b16
 *       b_tr == 0: B^T as [N rows][K cols] (nn.Linear's weight layout);  b_tr == 1: B as [K][N]
 *     One stored matrix serves both roles: linear.weight [N_ent, 3D] is op(B) of the logits GEMM (b_tr = 0) and of
 *     dfeat = dlogits W (b_tr = 1); dlogits [B, N_ent] is op(A) of dfeat (a_tr = 0) and of dW = dlogits^T feat (a_tr = 1).
 *     split_k / workspace as renet_gemm_f32.  Replaces torch.mm / nn.Linear /
 *                      nn.GRU's input projection at config 5 (RGCN.py:35, model.py:86-99) -- the reference itself
 *                      has no bf16 path; tolerances in tests/test_gpu_bf16.py. */
size_t renet_bf16_bytes(int R, int C);
int renet_pack_bf16(const float* X, int R, int C, int ldx, void* out, void* stream);
int renet_gemm_bf16s(int a_tr, int b_tr, int M, int N, int K, float alpha, const void* Ap, int lda, const void* Bp,
                     int ldb, float beta, float* C, int ldc, const float* bias, int split_k, float* workspace,
                     size_t workspace_bytes, void* stream);
/* Producers that write the bf16 operand format directly (no fp32 copy of the tensor exists in bf16 mode).  A
 * bf16-storage GEMM reads the contraction dimension in 64-wide stages, so the padding a consumer may touch --
 * columns [C, ceil64(C)) of the valid rows and rows [rows, ceil64(rows)) -- must be zero: each producer below zeroes
 * it, renet_bf16_zero_padding does so for a buffer filled by some other kernel.
 *   renet_seq_assemble_fwd_bf16   : renet_seq_assemble_fwd with X [S, 4D] / Xr [S, 3D] as bf16, row strides ldx / ldxr
 *   renet_softmax_ce_bf16         : renet_softmax_ce with the gradient (softmax - onehot) * grad_scale as bf16 (the
 *                                   fp32 logits are left untouched)
 *   renet_gru_bwd_layouts_bf16out : renet_gru_bwd_layouts_bf16 with dGi / dGh [S, 3H] as bf16, row stride out_ld
 *   renet_colsum_bf16 / renet_scale_bf16_by_device_scalar : the bias-gradient column sums / upstream-gradient scaling
 *                                   on such matrices */
int renet_bf16_zero_padding(void* P, int rows, int C, int ld, int rows_alloc, void* stream);
int renet_seq_assemble_fwd_bf16(const float* h2, const float* ent, const float* rel, const float* glob,
                                const int32_t* subj_row, const int32_t* row_ent, const int32_t* row_rel,
                                const int32_t* glob_row, int S, int D, float drop_p, uint64_t seed_x,
                                uint64_t seed_xr, void* X, int ldx, void* Xr, int ldxr, int rows_alloc,
                                void* stream);
int renet_softmax_ce_bf16(const float* logits, const int32_t* target, int B, int C, int ld, float grad_scale,
                          float* row_loss, void* dlogits_bf16, int ld16, int rows16, void* stream);
int renet_gru_bwd_layouts_bf16out(int n, const float* const* dh_last, const int32_t* const* step_off, const int* L,
                                  int H, const float* const* Whh, const float* const* saved, void* const* dGi16,
                                  void* const* dGh16, int out_ld, float* workspace, size_t workspace_bytes,
                                  void* stream);
int renet_colsum_bf16(const void* X, int M, int N, int ldx, float* out, float beta, float* workspace,
                      size_t workspace_bytes, void* stream);
int renet_scale_bf16_by_device_scalar(void* x, size_t n, const float* scale, void* stream);

/* PLANES (round 6): the bf16x6 arithmetic of renet_gemm_f32_split on operands that are ALREADY split -- a matrix [R, C] is
 * stored as three bf16 matrices ("planes", x = p1 + p2 + p3, p1 = rne(x), p2 = rne(x - p1), p3 = rne(x - p1 - p2): the
 * split the bf16x6 loaders perform per k-tile), each over the padded extent [Rp][Cp] (R, C rounded up to multiples of 256,
 * ZERO padding), `plane` elements apart.  Inside a plane the elements are TILED, not row-major ("T16": 16 x 16 tiles of
 * 512 bytes, tile (tr, tc) at (tr * Cp / 16 + tc) * 256 elements; element (i, j) of a tile at
 * (i ^ 4 * (tc & 1)) * 16 + ((j >> 3) ^ ((i >> 3) & 1)) * 8 + (j & 7) -- csrc/common.h renet_t16_off): every 1 KB
 * LDS-DMA piece of the GEMM is then two whole tiles whether the matrix is contracted over its columns or over its rows,
 * and both kinds of fragment read are free of LDS bank conflicts (tools/p6_layout_sim.py models the data path).
 * The producers write the format directly (renet_softmax_ce_planes: the CE gradient; renet_pack_planes: anything
 * else -- a weight once per optimizer step), so the GEMM k-loop contains no conversion: LDS-DMA staging
 * (global_load_lds) into a 3-slot ring of 128 x 128 x 16 half-stages (two workgroups per CU; RENET_P6_TILE=256: 4 slots of
 * 256 x 128 x 16, one workgroup per CU), fragments double buffered in registers, six
 * v_mfma_f32_32x32x16_bf16 per fragment pair, fp32 accumulation; same result as renet_gemm_f32_split up to fp32
 * summation order.
 *   renet_planes_elems : Rp * Cp, the element count of one plane (the plane stride the packers use)
 *   renet_pack_planes  : fp32 X[R, C] (row stride ldx) -> planes; ones_col != 0 appends a column of ones (C + 1 columns:
 *                        see col_out)
 *   renet_gemm_planes  : C[M, N] = alpha * (*alpha_dev) * op(A) op(B) (+ bias) (+ beta * C);  lda / ldb = Cp of the
 *                        stored matrices
 *       a_tr == 0: A stored [M rows][K cols];  a_tr == 1: stored [K rows][M cols] (contraction over its rows)
 *       b_tr == 0: B^T stored [N rows][K cols] (nn.Linear's weight layout);  b_tr == 1: B stored [K rows][N cols]
 *       alpha_dev : optional DEVICE scalar folded into alpha (the upstream autograd gradient: no pass over the operand)
 *       col_out   : optional; N then counts one EXTRA logical column whose values go to col_out[M] instead of C -- with
 *                   B packed with ones_col this is the bias gradient sum_k op(A)[m, k] inside the weight-gradient GEMM
 *       split_k / workspace as renet_gemm_f32 (renet_gemm_workspace(M, N, split_k) with this N).
 * Replaces nn.Linear's forward and backward GEMMs of the entity score head (model.py:89-90 and autograd), like
 * renet_gemm_f32_split. */
size_t renet_planes_elems(int R, int C);
int renet_pack_planes(const float* X, int R, int C, int ldx, int ones_col, void* out, void* stream);
int renet_gemm_planes(int a_tr, int b_tr, int M, int N, int K, float alpha, const float* alpha_dev, const void* Ap,
                      int lda, size_t a_plane, const void* Bp, int ldb, size_t b_plane, float beta, float* C, int ldc,
                      const float* bias, float* col_out, int split_k, float* workspace, size_t workspace_bytes,
                      void* stream);
/* renet_softmax_ce with the gradient (softmax - onehot) * grad_scale written as T16 planes over [rows16][ld16] (`plane`
 * elements apart; ld16 % 16 == 0, rows16 >= ceil16(B); columns [C, ld16) of the rows < B and the rows [B, ceil16(B))
 * zeroed); the fp32 logits are left untouched.
 * Replaces F.cross_entropy's backward (model.py:91,100) as renet_softmax_ce does. */
int renet_softmax_ce_planes(const float* logits, const int32_t* target, int B, int C, int ld, float grad_scale,
                            float* row_loss, void* dl_planes, size_t plane, int ld16, int rows16, void* stream);

/* column sums: out[n] = beta * out[n] + sum_m X[m,n]  (bias gradients; beta = 1 accumulates straight into
 * an existing .grad); two deterministic passes over row groups, `workspace` = renet_colsum_workspace(M, N)
 * bytes (0 for short matrices). */
size_t renet_colsum_workspace(int M, int N);
int renet_colsum(const float* X, int M, int N, int ldx, float* out, float beta, float* workspace,
                 size_t workspace_bytes, void* stream);

/* x[0..n) *= *scale, with the factor read from DEVICE memory (an upstream autograd gradient: no host sync);
 * a factor of exactly 1 returns without touching x. */
int renet_scale_by_device_scalar(float* x, size_t n, const float* scale, void* stream);
/* The same, additionally writing *bound_out = |*scale| * bound_in: the caller's bound on max |x| carried through the
 * scaling (operand bound of renet_gemm_f32_h3) without a pass over x. */
int renet_scale_by_device_scalar_bound(float* x, size_t n, const float* scale, float bound_in, float* bound_out,
                                       void* stream);

/* ------------------------------------------------------------------------------------------------
 * Sequence assembly (Aggregator.py:142-165): builds the GRU inputs directly in PACKED time-major
 * layout (the .data of the PackedSequence the reference returns), with the dropout of
 * Aggregator.py:157-158 fused.  Row p of the packed layout belongs to sequence seq[p] (position in
 * the length-sorted batch):
 *   X [p,:] = drop([ h2[subj_row[p]] | ent[s[seq[p]]] | rel[r[seq[p]]] | glob[glob_row[p]] ])   (4D)
 *   Xr[p,:] = drop([ h2[subj_row[p]] | ent[s[seq[p]]] |                  glob[glob_row[p]] ])   (3D)
 * Masks are counter-based: keep(seed_x, p*4D+c), keep(seed_xr, p*3D+c).
 * Backward: dRows[p,:]  = grad wrt the gathered h2 row of packed row p (X and Xr parts);
 *           dEntSeq[i,:], dRelSeq[i,:] (i < B) = grad wrt ent[s_i] / rel[r_i] summed over the steps of
 *           sequence i (packed row of step j = step_off[j] + i; step_off is a DEVICE array [L+1]);
 *           rows of sequences without history come out zero.  The caller scatters the three with
 *           renet_segment_add.
 * ---------------------------------------------------------------------------------------------- */
int renet_seq_assemble_fwd(const float* h2, const float* ent, const float* rel, const float* glob,
                           const int32_t* subj_row, const int32_t* row_ent, const int32_t* row_rel,
                           const int32_t* glob_row, int S, int D, float drop_p, uint64_t seed_x,
                           uint64_t seed_xr, float* X, float* Xr, void* stream);
/* The same, additionally emitting the operand bounds of renet_gemm_f32_h3 for X and Xr: partX / partXr receive
 * renet_bound_parts(S * D) floats each (one maximum of |value written| per workgroup), so that no separate
 * renet_maxabs_partials pass over the 51 + 38 MB is needed.  S > 0. */
int renet_bound_parts(size_t total4);
int renet_seq_assemble_fwd_bounds(const float* h2, const float* ent, const float* rel, const float* glob,
                                  const int32_t* subj_row, const int32_t* row_ent, const int32_t* row_rel,
                                  const int32_t* glob_row, int S, int D, float drop_p, uint64_t seed_x,
                                  uint64_t seed_xr, float* X, float* Xr, float* partX, float* partXr, void* stream);
int renet_seq_assemble_bwd(const float* dX, const float* dXr, const int32_t* step_off, int L, int S,
                           int B, int D, float drop_p, uint64_t seed_x, uint64_t seed_xr, float* dRows,
                           float* dEntSeq, float* dRelSeq, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GRU over the packed <= seq_len window (torch.nn.GRU semantics, 1 layer, h0 = 0, gate order r,z,n;
 * model.py:28-29,86,94; global_model.py:25,49).  Gi = X @ W_ih^T + b_ih is computed by the caller
 * with renet_gemm_f32; these entry points run the recurrence.
 *   Gi      [S, 3H]  packed time-major (step j occupies rows [off[j], off[j+1]), sequence i of the
 *                    length-sorted batch is row off[j]+i; batch_sizes non-increasing)
 *   step_off[L+1]    HOST array of row offsets (off[0] = 0, off[L] = S)
 *   Whh [3H,H], bhh[3H]
 *   h_last  [out_rows,H] state of every sequence after its own last step (h_n of nn.GRU), B = off[1];
 *                    rows [B, out_rows) are written as zeros (the reference pads h_n for the empty
 *                    histories, model.py:88): out_rows >= B
 *   saved   [S, 5H]  per packed row: r, z, n, (W_hn h + b_hn), h_prev   (consumed by backward)
 * Backward: given dh_last[B,H] produces dGi[S,3H], dGh[S,3H] (caller forms dW_ih, dW_hh, biases,
 * dX with renet_gemm_f32 / renet_colsum).
 * `workspace` = renet_gru_workspace(B, H) bytes per GRU (B = the largest first-step batch size of the
 * GRUs of the call), for both directions: the bf16 planes of W_hh (forward) / W_hh^T and its planes
 * (backward), then the state that travels between the per-step launches (bf16 planes of h / dGh, ping-pong,
 * and the fp32 dh) -- the recurrent products run as "bf16x6" like renet_gemm_f32_split unless RENET_GEMM=f32
 * selects the exact-fp32 MFMA kernels (one persistent launch, no per-step state: B may then be 0).
 * ---------------------------------------------------------------------------------------------- */
size_t renet_gru_workspace(int B, int H);
int renet_gru_fwd(const float* Gi, const int32_t* step_off, int L, int H, const float* Whh,
                  const float* bhh, float* h_last, int out_rows, float* saved, float* workspace,
                  size_t workspace_bytes, void* stream);
int renet_gru_bwd(const float* dh_last, const int32_t* step_off, int L, int H, const float* Whh,
                  const float* saved, float* dGi, float* dGh, float* workspace,
                  size_t workspace_bytes, void* stream);
/* n (1..4) independent GRUs over the SAME packed layout in one launch (RE-Net's `encoder` and
 * `encoder_r`, model.py:86,94): every pointer argument is a HOST array of n device pointers;
 * the workspace is n * renet_gru_workspace bytes. */
int renet_gru_fwd_multi(int n, const float* const* Gi, const int32_t* step_off, int L, int H,
                        const float* const* Whh, const float* const* bhh, float* const* h_last, int out_rows,
                        float* const* saved, float* workspace, size_t workspace_bytes, void* stream);
int renet_gru_bwd_multi(int n, const float* const* dh_last, const int32_t* step_off, int L, int H,
                        const float* const* Whh, const float* const* saved, float* const* dGi,
                        float* const* dGh, float* workspace, size_t workspace_bytes, void* stream);
/* n (<= 4) GRUs of up to TWO packed layouts in one launch: step_off[k] / L[k] / out_rows[k] describe problem k
 * (problems that pass the same step_off pointer share a layout).  The subject and the object pass of a training
 * step (train.py:136-137) are independent until their losses are added: their four recurrences -- ~60 workgroups
 * each -- then run side by side on the 256 CUs.  Equal Whh pointers are split into planes once. */
int renet_gru_fwd_layouts(int n, const float* const* Gi, const int32_t* const* step_off, const int* L, int H,
                          const float* const* Whh, const float* const* bhh, float* const* h_last,
                          const int* out_rows, float* const* saved, float* workspace, size_t workspace_bytes,
                          void* stream);
int renet_gru_bwd_layouts(int n, const float* const* dh_last, const int32_t* const* step_off, const int* L, int H,
                          const float* const* Whh, const float* const* saved, float* const* dGi,
                          float* const* dGh, float* workspace, size_t workspace_bytes, void* stream);
/* renet_gru_bwd_layouts that additionally emits the operand bound of dGi for renet_gemm_f32_h3: bounds[k] receives
 * renet_gru_bound_parts(max over the problems of their sequence count) floats, one maximum of |dGi| per workgroup
 * (|dGh| <= |dGi| elementwise: the same bound serves both).  Only the persistent bf16x6 recurrence emits them;
 * RENET_ERR_UNSUPPORTED otherwise (RENET_GEMM=f32, RENET_GRU=steps) -- call the plain entry then. */
int renet_gru_bound_parts(int max_rows);
int renet_gru_bwd_layouts_bounds(int n, const float* const* dh_last, const int32_t* const* step_off, const int* L, int H,
                                 const float* const* Whh, const float* const* saved, float* const* dGi,
                                 float* const* dGh, float* const* bounds, float* workspace, size_t workspace_bytes,
                                 void* stream);
/* The same recurrences with EXACT fp32 products (v_mfma_f32_16x16x4_f32) whatever RENET_GEMM says: the per-model form of the
 * exact mode (round 5; `net.gemm_mode = 'f32'`).  RENET_GEMM=f32 makes the plain entries above behave like these. */
int renet_gru_fwd_layouts_f32(int n, const float* const* Gi, const int32_t* const* step_off, const int* L, int H,
                              const float* const* Whh, const float* const* bhh, float* const* h_last,
                              const int* out_rows, float* const* saved, float* workspace, size_t workspace_bytes,
                              void* stream);
int renet_gru_bwd_layouts_f32(int n, const float* const* dh_last, const int32_t* const* step_off, const int* L, int H,
                              const float* const* Whh, const float* const* saved, float* const* dGi,
                              float* const* dGh, float* workspace, size_t workspace_bytes, void* stream);
/* The same recurrences in bf16 mode (BASELINE config 5): W_hh and the hidden state / gate gradients are rounded to
 * bf16 (RNE) as MFMA operands -- ONE v_mfma_f32_16x16x32_bf16 product per fragment pair instead of the six of the
 * fp32-class split, a third of the W_hh stream -- with fp32 accumulation, fp32 gate math, fp32 state and outputs. */
int renet_gru_fwd_layouts_bf16(int n, const float* const* Gi, const int32_t* const* step_off, const int* L, int H,
                               const float* const* Whh, const float* const* bhh, float* const* h_last,
                               const int* out_rows, float* const* saved, float* workspace, size_t workspace_bytes,
                               void* stream);
int renet_gru_bwd_layouts_bf16(int n, const float* const* dh_last, const int32_t* const* step_off, const int* L, int H,
                               const float* const* Whh, const float* const* saved, float* const* dGi,
                               float* const* dGh, float* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Score head (model.py:89-91, 98-100).
 * renet_concat3_fwd/bwd : feat[b,:] = drop([ a[ia[b]] | hmid[b] | c[ic[b]] ])   (c may be NULL: 2 parts)
 * renet_softmax_ce      : per-row loss_b = logsumexp(logits[b,:]) - logits[b,target[b]];
 *                         loss_sum += sum_b loss_b (one float, zeroed by the caller);
 *                         if dlogits != NULL: dlogits = (softmax - onehot) * grad_scale  (may alias logits)
 * ---------------------------------------------------------------------------------------------- */
int renet_concat3_fwd(const float* a, const int32_t* ia, const float* hmid, const float* c,
                      const int32_t* ic, int B, int D, float drop_p, uint64_t seed, float* feat,
                      void* stream);
/* The same with the operand bound of feat: bound_part receives renet_bound_parts(B * parts * D / 4) floats. */
int renet_concat3_fwd_bounds(const float* a, const int32_t* ia, const float* hmid, const float* c,
                             const int32_t* ic, int B, int D, float drop_p, uint64_t seed, float* feat,
                             float* bound_part, void* stream);
int renet_concat3_bwd(const float* dfeat, int B, int D, int parts, float drop_p, uint64_t seed,
                      float* da_rows, float* dhmid, float* dc_rows, void* stream);
/* y = x * keepmask(seed) / (1 - p) on float4 groups (n % 4 == 0); its own backward (apply to dy).
 * Used for the dropout of Aggregator.py:69 (global model sequence tensor). */
int renet_dropout(const float* x, size_t n, float drop_p, uint64_t seed, float* y, void* stream);
int renet_softmax_ce(const float* logits, const int32_t* target, int B, int C, int ld,
                     float grad_scale, float* row_loss, float* dlogits, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Per-graph readout of the global model (Aggregator.py:58-61 dgl.max_nodes / mean_nodes):
 *   out[g,:] = max / mean over rows [seg_ptr[g], seg_ptr[g+1]) of h; argmax[g,:] saved for backward. */
int renet_segment_pool_fwd(const float* h, const int32_t* seg_ptr, int G, int D, int is_max,
                           float* out, int32_t* argmax, void* stream);
int renet_segment_pool_bwd(const float* dout, const int32_t* seg_ptr, const int32_t* argmax, int G,
                           int D, int is_max, int N, float* dh, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused clip_grad_norm_ + Adam (+ weight decay) + zero_grad over flat buffers (train.py:140-142).
 * p, g, m, v: n floats each (16-byte aligned); `step` is the 1-based step count (bias correction);
 * max_norm <= 0 disables clipping; grad_norm_out (optional device float) receives the pre-clip norm.
 * torch.optim.Adam semantics (L2 weight decay added to the clipped gradient). */
size_t renet_adam_workspace(size_t n);
int renet_adam_step(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, float max_norm, int step, int zero_grad, float* workspace,
                    size_t workspace_bytes, float* grad_norm_out, void* stream);
/* The same step on the gradient g * grad_scale (norm, clip coefficient and update all see the scaled gradient):
 * grad_scale = 1 / world_size turns the SUM all-reduce of the data-parallel exchange into the mean without a
 * separate pass over the 81 MB buffer (the reference has no distributed step; train.py:140-142 on the averaged
 * gradient is what gradient accumulation over world_size batches would do). */
int renet_adam_step_scaled(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                           float eps, float weight_decay, float max_norm, float grad_scale, int step,
                           int zero_grad, float* workspace, size_t workspace_bytes, float* grad_norm_out,
                           void* stream);

/* The same step with the sum of squares ALREADY accumulated per region of the flat gradient (data-parallel runs: each
 * all-reduce bucket's partials are computed on the reducer's stream as the bucket arrives, so that only this scalar combine
 * is left of clip_grad_norm_, train.py:140, when the last bucket lands):
 *   renet_sumsq_partials      : partial[0..n_slots) = per-workgroup sums of g[i]^2 over g[0..n) (fixed order: deterministic;
 *                               slots without elements receive 0); g 16-byte aligned; n_slots <= 2048
 *   renet_adam_step_presummed : renet_adam_step_scaled without its own norm pass: ||g||^2 = sum of partial[0..n_partial) */
int renet_sumsq_partials(const float* g, size_t n, float* partial, int n_slots, void* stream);
int renet_adam_step_presummed(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, float max_norm, float grad_scale, int step, int zero_grad,
                              const float* partial, int n_partial, float* grad_norm_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Inference: the joint (relation, object) distribution of pred_r_rank2 and its top-k (model.py:205-209,239).
 *   renet_joint_softmax : logits [n * R, N] (row stride ld) is overwritten, row (e, r) with
 *                           softmax(logits[e, r, :]) * softmax(logits_r[e, :])[r] * prob_e[e]
 *                         (the reference: torch.softmax x2 + two broadcast multiplies); N * 4 <= 128 KB, R <= 1024.
 *   renet_topk_positive : the k largest elements of every row of x [n, M] (row stride ldx), x >= 0: values and int64
 *                         column indices, in NO particular order (torch.topk(..., sorted=False)); exact (radix select
 *                         on the bit pattern); ties at the threshold are broken arbitrarily.
 *                         workspace: renet_topk_workspace(n) bytes. */
int renet_joint_softmax(float* logits, int ld, int n, int R, int N, const float* logits_r, int ld_r,
                        const float* prob_e, void* stream);
size_t renet_topk_workspace(int n);
int renet_topk_positive(const float* x, size_t ldx, int n, int M, int k, float* out_val, int64_t* out_idx,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * DEVICE batch-graph builder for the merged training batch (both directions of train.py:136-137 as one batch of 2B
 * sequences: graph.build_batch_both; replaces utils.py:209-244 + 115-131 + dgl.batch and this library's own HOST
 * builder for that case).  The dataset is resident in HBM (RenetStoreDev: quadruples, the per-role history index of
 * preprocess.HistoryIndex, the per-timestamp fact lists of graph.GraphStore); one call takes the B quadruple indices of
 * a batch (device int32) and fills the caller's output arrays with EXACTLY the arrays the host builder produces
 * (graph.HostBatch; tests/test_gpu_builder.py compares them bit for bit).  No host synchronisation inside: every stage
 * launches over the capacities and guards on the device-side counts, which the caller copies back (RENET_BB_NCOUNTS
 * int32) when it needs the sizes.  counts[RENET_BB_ERR] != 0 => a capacity was exceeded (or a timestamp was not in the
 * store): the outputs are invalid, rebuild with larger capacities / on the host. */
typedef struct {
    const int32_t *q_s, *q_r, *q_o;            /* [n_quads] the stream's quadruples */
    const int32_t* h_first[2];                 /* role 0 = subject histories, 1 = object histories: per quadruple the */
    const int32_t* h_count[2];                 /*   first snapshot of its window and the number of snapshots in it    */
    const int32_t* snap_t[2];                  /* [n_snap] timestamp of every snapshot                                */
    const int32_t* snap_ptr[2];                /* [n_snap + 1] neighbour range of every snapshot                      */
    const int32_t* nbr_o[2];                   /* [n_quads] neighbour entities                                        */
    const int32_t* times;                      /* [T] sorted timestamps of graph_dict                                 */
    const int32_t* trip_ptr;                   /* [T + 1] fact range of every timestamp                               */
    const int32_t *trip_s, *trip_r, *trip_o;   /* [n_facts]                                                           */
    const int32_t* glob_times;                 /* [n_glob] sorted timestamps of the global-embedding table            */
    int T, n_glob, n_facts, num_ent, num_rels;
} RenetStoreDev;

typedef struct {
    int32_t *node_ent, *node_slot, *row_ptr, *col, *etype;
    float* norm;
    int32_t *heavy_rows, *e_src, *e_dst, *chunk_ptr, *chunk_type, *type_chunk_ptr;
    int32_t *e_src2, *e_dst2, *chunk_ptr2, *chunk_type2, *type_chunk_ptr2;
    int32_t *it_src, *it_type, *grp_ptr;
    int32_t *subj_row, *row_seq, *row_ent, *row_rel, *glob_row;
    int32_t *s_sorted, *r_sorted, *rel_label, *ent_label, *perm, *step_off;
    int32_t* plan_order[4];                    /* 0: node_ent, 1: subj_row, 2: s_sorted, 3: r_sorted */
    int32_t* plan_seg[4];
    int32_t* plan_target[4];
    int32_t* counts;                           /* [RENET_BB_NCOUNTS] */
    int cap_nodes, cap_edges;
} RenetBatchOut;

enum {
    RENET_BB_NNZ = 0, RENET_BB_S, RENET_BB_L, RENET_BB_TB, RENET_BB_FACTS, RENET_BB_N, RENET_BB_NA, RENET_BB_E2,
    RENET_BB_E, RENET_BB_NHEAVY, RENET_BB_NHEAVY_OUT, RENET_BB_NCHUNKS, RENET_BB_NCHUNKS2, RENET_BB_NITEMS,
    RENET_BB_NGROUPS, RENET_BB_NGROUPS_OUT, RENET_BB_NSEG0, RENET_BB_NSEG1, RENET_BB_NSEG2, RENET_BB_NSEG3,
    RENET_BB_ERR, RENET_BB_EOUT,             /* E_out = edges into the row prefix [0, nA) */
    RENET_BB_STEP_OFF = 24,                  /* counts[24 .. 56]: copy of step_off[0 .. 32] (the GRU launches need it on the host) */
    RENET_BB_NCOUNTS = 64
};
enum { RENET_BB_ERR_TIME = 1, RENET_BB_ERR_GLOB = 2, RENET_BB_ERR_NODES = 4, RENET_BB_ERR_EDGES = 8 };

size_t renet_build_batch_workspace(const RenetStoreDev* store, int B, int cap_nodes, int cap_edges);
int renet_build_batch_both(const RenetStoreDev* store, const int32_t* idx_dev, int B, int seq_len, int heavy_thresh,
                           int group_budget, int chunk, const RenetBatchOut* out, void* workspace,
                           size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * THE MERGED TRAINING STEP AS ONE LAUNCH LIST (round 6; csrc/step.cpp): one iteration of train.py:136-139 --
 *     loss = model(batch, ..., subject=True) + model(batch, ..., subject=False);  loss.backward()
 * on the merged batch of both directions (renet_build_batch_both / graph.build_batch_both) -- issued from C: the forward
 * call enqueues the ~25 launches of  RGCN x2 (RGCN.py:33-51,79-94) -> sequence assembly (Aggregator.py:139-165) -> GRU x2
 * (model.py:86,94) -> both score heads + CE (model.py:89-103), the backward call the ~35 launches of their gradients, on two
 * HIP streams with the fork / join events between them (the relation head and encoder_r's projection next to the entity
 * head / encoder; parameter-gradient kernels that feed nothing but the optimizer on the side stream).  Every launch goes
 * through the entry points of this header with exactly the arguments the Python autograd path (re-net_amd/ops.py) passes,
 * so losses and gradients are BIT-identical to that path (tests/test_gpu_step_plan.py); the host cost of a step drops from
 * ~55 Python-level C-ABI calls to two.
 *   RenetStepModel : parameters and their gradient buffers (gradients ACCUMULATE: beta = 1 everywhere, as into an existing
 *                    .grad), the global-embedding matrix, optionally the planes of the entity head's weight
 *   RenetStepBatch : the device arrays of one merged batch (graph.DeviceGraph / gpu_builder.DeviceBatch) + its sizes
 *   RenetStepRun   : dropout seeds, loss scales, the workspace, the two streams
 *   renet_step_workspace : bytes of workspace for (model, batch); activations live there between forward and backward
 *   renet_step_forward   : writes the 2 B per-row losses (entity head rows, then relation head rows) to row_loss
 *   renet_step_backward  : upstream scalar g (device); defer_side != 0 leaves the side stream UN-joined (the caller joins it
 *                          before it reads a parameter gradient: parallel.HipAdam.step); ev_head_done (optional hipEvent_t)
 *                          is recorded when the entity head's weight / bias gradients are complete (early all-reduce bucket),
 *                          ev_gru_done (optional) when both encoders' parameter gradients are (the middle bucket)
 *   both return the number of C-ABI launches they issued in *n_launches (optional).
 * fp32-class default mode only (bf16x6 GEMMs, bf16x6 recurrences); tensors below 2 GiB. */
typedef struct { const int32_t *order, *seg_ptr, *target; int num_segments; } RenetSegPlan;
typedef struct {
    int D, num_ent, T, C2;                     /* n_hidden; entities; 2 * num_rels relation types; classes of the relation head */
    float drop_p;                              /* 0 in eval mode */
    const float *ent, *rel;                    /* [num_ent, D], [T, D] */
    const float *w1, *loop1, *w2, *loop2;      /* rgcn1 / rgcn2: weight [T, D * D / 100], loop_weight [D, D] */
    const float *wih, *whh, *bih, *bhh;        /* encoder   (4D -> D): [3D, 4D], [3D, D], [3D], [3D] */
    const float *wih_r, *whh_r, *bih_r, *bhh_r;/* encoder_r (3D -> D): [3D, 3D], ... */
    const float *lin_w, *lin_b;                /* linear   [num_ent, 3D], [num_ent] */
    const float *linr_w, *linr_b;              /* linear_r [C2, 2D], [C2] */
    float *g_ent, *g_rel, *g_w1, *g_loop1, *g_w2, *g_loop2, *g_wih, *g_whh, *g_bih, *g_bhh, *g_wih_r, *g_whh_r, *g_bih_r,
          *g_bhh_r, *g_lin_w, *g_lin_b, *g_linr_w, *g_linr_b;
    const float* glob;                         /* [n_glob, D] global embeddings (a constant of the step) */
    const void* lin_w_planes;                  /* optional: renet_pack_planes(lin_w); NULL = the in-loop-split head */
    size_t lin_w_plane;                        /*   elements per plane */
    int lin_w_ld;                              /*   Cp of the planes */
} RenetStepModel;
typedef struct {
    int N, E, nA, S, B, L, n_items;            /* nodes, edges, rows of layer 2, packed rows, sequences, steps, items */
    int n_groups, n_groups_out, n_heavy, n_heavy_out, n_chunks, n_chunks2;
    const int32_t* step_off_host;              /* [L + 1] HOST copy of step_off */
    const int32_t *node_ent, *row_ptr, *col, *etype;
    const float* norm;
    const int32_t *it_src, *it_type, *grp_ptr, *heavy_rows, *heavy_rows_out;
    const int32_t *it_src_t, *it_type_t, *col_t, *e_src_t;        /* renet_compose_table_items */
    const int32_t *e_src, *e_dst, *chunk_ptr, *chunk_type, *type_chunk_ptr;
    const int32_t *e_src2, *e_dst2, *chunk_ptr2, *chunk_type2, *type_chunk_ptr2;
    const int32_t *subj_row, *row_ent, *row_rel, *glob_row, *step_off;
    const int32_t *s_idx, *r_idx, *ent_label, *rel_label;
    RenetSegPlan plan_node_ent, plan_subj_row, plan_s, plan_r;
} RenetStepBatch;
typedef struct {
    uint64_t seed_rgcn1, seed_rgcn2, seed_x, seed_xr, seed_head1, seed_head2;
    float scale_ent, scale_rel;                /* CE gradient scales: loss_scale / B and loss_scale * 0.1 / B (model.py:103) */
    void* workspace;
    size_t workspace_bytes;
    void *stream, *side_stream;                /* hipStream_t; side_stream may equal stream (everything in stream order) */
} RenetStepRun;
size_t renet_step_workspace(const RenetStepModel* m, const RenetStepBatch* b);
int renet_step_forward(const RenetStepModel* m, const RenetStepBatch* b, const RenetStepRun* r, float* row_loss,
                       int* n_launches);
int renet_step_backward(const RenetStepModel* m, const RenetStepBatch* b, const RenetStepRun* r, const float* g,
                        int defer_side, void* ev_head_done, void* ev_gru_done, int* n_launches);
/* x[0..n) += y[0..n) (the two heads' gradients wrt the same gathered rows ent[s], model.py:89,98, before ONE scatter-add) */
int renet_add_inplace(float* x, const float* y, size_t n, void* stream);
/* x[0..n) = 0 as an ordinary kernel launch (the zero-initialised scatter-add targets of the backward pass: hipMemsetAsync goes
 * through the runtime's blit path, whose packets do not pipeline with the neighbouring kernel dispatches) */
int renet_zero(float* x, size_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * HOST-side batch-graph builder passes (no device work; pointers are HOST arrays): the native form of
 * graph.build_batch's heavy middle, replacing the reference's per-batch DGL subgraph/batch calls
 * (utils.py:115-131,158-170,236-241).  See csrc/host_builder.cpp for the contracts. */
int64_t renet_host_filter_edges(const int64_t* trip_ptr, const int64_t* trip_s, const int64_t* trip_r,
                                const int64_t* trip_o, const int64_t* ti, int64_t Tb, int64_t num_ent,
                                const int64_t* keys, const int32_t* new_id, int64_t N, int32_t* table,
                                int64_t* out_ls, int64_t* out_lo, int64_t* out_rr);
/* The same result for MANY small member graphs (batched inference: one slot per (entity, timestamp)): walks only
 * the facts whose subject is in the slot's node set through a per-timestamp subject index -- by_subj = fact
 * indices sorted stably by (timestamp, subject), subj_sorted = their subjects; `table`: num_ent entries, -1. */
int64_t renet_host_filter_edges_sparse(const int64_t* trip_ptr, const int64_t* trip_s, const int64_t* trip_r,
                                       const int64_t* trip_o, const int64_t* by_subj,
                                       const int64_t* subj_sorted, const int64_t* ti, int64_t Tb,
                                       int64_t num_ent, const int64_t* keys, const int32_t* new_id, int64_t N,
                                       int32_t* table, int64_t* out_ls, int64_t* out_lo, int64_t* out_rr);
/* The relation-bucketed chunk list of renet_host_edge_layouts restricted to edges with dst < n_out (a layer
 * evaluated on a row prefix); returns the number of kept edges. */
int64_t renet_host_type_chunks(int64_t E, const int64_t* src, const int64_t* dst, const int64_t* et, int64_t T,
                               int64_t chunk, int64_t n_out, int32_t* e_src, int32_t* e_dst,
                               int32_t* type_chunk_ptr, int32_t* chunk_type, int32_t* chunk_ptr,
                               int64_t* n_chunks);
/* Node sets of a batch (utils.py:149-156) -- per member graph the union of the subjects and history objects of its
 * steps -- as sorted keys slot * num_ent + entity, with the rows that are some step's subject numbered first
 * (graph.build_batch documents the outputs); `table`: num_ent entries, -1.  Returns the number of nodes. */
int64_t renet_host_node_sets(int64_t S, const int64_t* slot_k, const int64_t* subj_ent, const int64_t* nbr_begin,
                             const int64_t* nbr_cnt, const int64_t* nbr_o, int64_t Tb, int64_t num_ent,
                             int32_t* table, int64_t* keys, int64_t* subj_pos, int64_t* new_id,
                             int32_t* node_ent, int64_t* node_slot, int64_t* n_a);
void renet_host_edge_layouts(int64_t n, int64_t E, const int64_t* src, const int64_t* dst, const int64_t* et,
                             int64_t T, int64_t chunk, int64_t heavy, int32_t* col, int32_t* etype,
                             int32_t* row_ptr, float* norm, int32_t* heavy_rows, int64_t* n_heavy,
                             int32_t* e_src, int32_t* e_dst, int32_t* type_chunk_ptr, int32_t* chunk_type,
                             int32_t* chunk_ptr, int64_t* n_chunks);
int64_t renet_host_segplan(const int64_t* idx, int64_t n, int64_t bound, int32_t* order, int32_t* seg_ptr,
                           int32_t* target);
/* Item stream + wave groups consumed by renet_rgcn_gather_items, from the CSR of renet_host_edge_layouts (its
 * contract is stated there and in graph.plan_gather_items, the numpy specification).  Capacities: it_src / it_type
 * E + N, grp_ptr N + 2.  Returns the number of groups, or -1 if budget + heavy + 1 > 64. */
int64_t renet_host_gather_items(int64_t N, const int32_t* row_ptr, const int32_t* col, const int32_t* etype,
                                int64_t heavy, int64_t budget, int64_t n_out, int32_t* it_src, int32_t* it_type,
                                int32_t* grp_ptr, int64_t* n_items, int64_t* n_groups_out);

#ifdef __cplusplus
}
#endif
#endif /* RENET_HIP_H */
