// Inference-side selection kernels for gfx950 (reference model.py:205-209,239: pred_r_rank2 + the per-entity top-k of
// RENet.predict): for every sampled entity e the reference forms the joint distribution
//     joint[e, r, o] = softmax_o(logits[e, r, :])[o] * softmax_r(logits_r[e, :])[r] * prob[e]        (R x N_ent values)
// with torch.softmax x2, two broadcast multiplies and torch.topk(k = num_k, sorted=False) over the flattened R * N_ent
// axis -- ~10 passes over a [n * R, N_ent] fp32 block (260 MB for 11 entities at ICEWS18 sizes) plus temporaries.
// Here:
//   renet_joint_softmax : ONE read + ONE write of the block: the row is staged in LDS (as softmax_ce_lds_kernel), max /
//                         sum / normalise run out of LDS, and the relation and entity factors are folded into the
//                         write.  Same operation order as the reference's expression (exp(x - max) / sum, then * p_r,
//                         then * prob).
//   renet_topk_positive : exact top-k of every row of a [n, M] matrix of POSITIVE floats by radix select on the bit
//                         pattern (monotone for positive floats): three histogram passes (12 + 12 + 8 bits, LDS
//                         histograms, integer atomics: deterministic counts), a threshold pick per row, one collect pass.
//                         The k results come back in no particular order (the reference asks for sorted=False);
//                         values equal to the threshold are taken in arbitrary order, as torch.topk does.
// HBM / L2-bound integer and exp work: no MFMA.
#include "common.h"
#include <math.h>

namespace {

__device__ __forceinline__ float tk_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float tk_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// one workgroup (1024 threads) per row (e, r) of the [n * R, N] block
__global__ __launch_bounds__(1024) void joint_softmax_kernel(float* x, int ld, int N, int R,
                                                             const float* __restrict__ logits_r, int ld_r,
                                                             const float* __restrict__ prob_e) {
    extern __shared__ float row[];
    __shared__ float red[16];
    __shared__ float s_pr;
    const int rowid = blockIdx.x;
    const int e = rowid / R, r = rowid - e * R;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* xr = x + (size_t)rowid * ld;
    // relation factor p_r = softmax(logits_r[e, :])[r] (R <= 1024 values: wave 0 alone)
    if (wave == 0) {
        const float* lr = logits_r + (size_t)e * ld_r;
        float m = -INFINITY;
        for (int c = lane; c < R; c += 64) m = fmaxf(m, lr[c]);
        m = tk_wave_max(m);
        float s = 0.f;
        for (int c = lane; c < R; c += 64) s += expf(lr[c] - m);
        s = tk_wave_sum(s);
        if (lane == 0) s_pr = expf(lr[r] - m) / s;
    }
    float m = -INFINITY;
    int c = threadIdx.x;
    for (; c + 7 * 1024 < N; c += 8 * 1024) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = xr[c + q * 1024];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            row[c + q * 1024] = v[q];
            m = fmaxf(m, v[q]);
        }
    }
    for (; c < N; c += 1024) {
        const float v = xr[c];
        row[c] = v;
        m = fmaxf(m, v);
    }
    m = tk_wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
    __syncthreads();
    float s = 0.f;
    for (int cc = threadIdx.x; cc < N; cc += 1024) {
        const float ex = expf(row[cc] - m);
        row[cc] = ex;
        s += ex;
    }
    s = tk_wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += red[w];
    const float pr = s_pr, pe = prob_e[e];
    for (int cc = threadIdx.x; cc < N; cc += 1024) xr[cc] = ((row[cc] / s) * pr) * pe;
}

// ---- radix select ---------------------------------------------------------------------------------------------
// Order-preserving key of a float: unsigned comparison of keys == comparison of the values for EVERY finite input
// (the raw bit pattern ranks negatives, and -0.0, above all positives).  Non-negative values get the sign
// bit set, negative values are inverted; -0.0 is treated as +0.0 and a NaN as the SMALLEST key (never selected
// before any number) -- the joint probabilities this kernel is used on are >= 0, for which the key is the raw bit
// pattern with the top bit set, i.e. the selection is what it was.
__device__ __forceinline__ unsigned topk_key(float v) {
    unsigned u = __float_as_uint(v);
    if (v != v) return 0u;
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct SelState {          // per row of the [n, M] matrix
    unsigned prefix;       // the key bits decided so far (high bits), 0 before pass 1
    int remaining;         // results still to be taken from the undecided elements
    int cnt_gt;            // collect pass: write cursor of the elements above the threshold
    int cnt_eq;            // ... and of the elements equal to it
};

constexpr int TK_BINS = 4096;
constexpr int TK_THREADS = 512;

// PASS 0: bins = key >> 20 (12 bits); PASS 1: keys whose top 12 bits == prefix: (key >> 8) & 0xFFF; PASS 2: keys whose
// top 24 bits == prefix: key & 0xFF.  grid = (slices, n); a workgroup histograms its slice of row blockIdx.y in LDS
// and flushes the non-empty bins with integer atomics.
template <int PASS>
__global__ __launch_bounds__(TK_THREADS) void topk_hist_kernel(const float* __restrict__ x, size_t ldx, int M,
                                                               const SelState* __restrict__ st,
                                                               unsigned* __restrict__ hist) {
    __shared__ unsigned h[TK_BINS];
    for (int i = threadIdx.x; i < TK_BINS; i += TK_THREADS) h[i] = 0u;
    __syncthreads();
    const int rowid = blockIdx.y;
    const unsigned prefix = st[rowid].prefix;
    const float* xr = x + (size_t)rowid * ldx;
    const int per = (M + gridDim.x - 1) / gridDim.x;
    const int i0 = blockIdx.x * per, i1 = min(M, i0 + per);
    for (int i = i0 + threadIdx.x; i < i1; i += TK_THREADS) {
        const unsigned key = topk_key(xr[i]);
        if (PASS == 0) atomicAdd(&h[key >> 20], 1u);
        else if (PASS == 1) { if ((key >> 20) == prefix) atomicAdd(&h[(key >> 8) & 0xFFFu], 1u); }
        else { if ((key >> 8) == prefix) atomicAdd(&h[key & 0xFFu], 1u); }
    }
    __syncthreads();
    unsigned* hg = hist + (size_t)rowid * TK_BINS;
    for (int i = threadIdx.x; i < TK_BINS; i += TK_THREADS)
        if (h[i]) atomicAdd(&hg[i], h[i]);
}

// one workgroup per row: the highest bin b with (count of bins > b) < remaining <= (count of bins >= b); the prefix
// grows by b's bits, remaining shrinks by the count above b; the histogram is cleared for the next pass.
template <int PASS>
__global__ __launch_bounds__(256) void topk_pick_kernel(SelState* __restrict__ st, unsigned* __restrict__ hist) {
    __shared__ unsigned part[256];
    const int rowid = blockIdx.x;
    unsigned* hg = hist + (size_t)rowid * TK_BINS;
    constexpr int NB = PASS == 2 ? 256 : TK_BINS;
    constexpr int PER = NB / 256;                       // bins per thread, thread t owns bins [t * PER, (t + 1) * PER)
    unsigned loc[PER == 0 ? 1 : PER];
    unsigned tot = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) { loc[j] = hg[threadIdx.x * PER + j]; tot += loc[j]; }
    part[threadIdx.x] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int need = st[rowid].remaining;
        unsigned above = 0;
        int t = 255;
        for (; t > 0; --t) {
            if (above + part[t] >= (unsigned)need) break;
            above += part[t];
        }
        part[0] = (unsigned)t;                          // owner thread of the threshold bin
        part[1] = above;
    }
    __syncthreads();
    const int owner = (int)part[0];
    if ((int)threadIdx.x == owner) {
        unsigned above = part[1];
        const int need = st[rowid].remaining;
        int j = PER - 1;
        for (; j > 0; --j) {
            if (above + loc[j] >= (unsigned)need) break;
            above += loc[j];
        }
        const unsigned bin = (unsigned)(owner * PER + j);
        SelState s = st[rowid];
        s.prefix = PASS == 0 ? bin : PASS == 1 ? ((s.prefix << 12) | bin) : ((s.prefix << 8) | bin);
        s.remaining = need - (int)above;
        s.cnt_gt = 0; s.cnt_eq = 0;
        st[rowid] = s;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) hg[threadIdx.x * PER + j] = 0u;
}

// after pass 2: prefix = the threshold's full bit pattern T, remaining = how many elements EQUAL to T belong to the
// result (>= 1).  Elements > T go to slots [0, k - remaining), elements == T to the rest (first come, first served).
__global__ __launch_bounds__(TK_THREADS) void topk_collect_kernel(const float* __restrict__ x, size_t ldx, int M, int k,
                                                                  SelState* __restrict__ st,
                                                                  float* __restrict__ out_val,
                                                                  int64_t* __restrict__ out_idx) {
    const int rowid = blockIdx.y;
    const unsigned T = st[rowid].prefix;
    const int n_eq = st[rowid].remaining, n_gt = k - n_eq;
    const float* xr = x + (size_t)rowid * ldx;
    const int per = (M + gridDim.x - 1) / gridDim.x;
    const int i0 = blockIdx.x * per, i1 = min(M, i0 + per);
    for (int i = i0 + threadIdx.x; i < i1; i += TK_THREADS) {
        const float v = xr[i];
        const unsigned key = topk_key(v);
        if (key > T) {
            const int slot = atomicAdd(&st[rowid].cnt_gt, 1);
            if (slot < n_gt) { out_val[(size_t)rowid * k + slot] = v; out_idx[(size_t)rowid * k + slot] = i; }
        } else if (key == T) {
            const int slot = atomicAdd(&st[rowid].cnt_eq, 1);
            if (slot < n_eq) { out_val[(size_t)rowid * k + n_gt + slot] = v; out_idx[(size_t)rowid * k + n_gt + slot] = i; }
        }
    }
}

__global__ void topk_init_kernel(SelState* st, int n, int k) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { st[i].prefix = 0u; st[i].remaining = k; st[i].cnt_gt = 0; st[i].cnt_eq = 0; }
}

}  // namespace

extern "C" {

int renet_joint_softmax(float* logits, int ld, int n, int R, int N, const float* logits_r, int ld_r,
                        const float* prob_e, void* stream) {
    if (n < 0 || R <= 0 || N <= 0 || ld < N || ld_r < R || !logits || !logits_r || !prob_e) return RENET_ERR_BADARG;
    if (n == 0) return RENET_OK;
    const size_t lds = (size_t)N * sizeof(float);
    if (lds > 128 * 1024 || R > 1024) return RENET_ERR_UNSUPPORTED;
    static bool attr_set = false;          // benign race: the attribute is idempotent
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)joint_softmax_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           128 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    RENET_LAUNCH(joint_softmax_kernel, dim3((unsigned)(n * R)), dim3(1024), lds, (hipStream_t)stream, logits, ld, N, R,
                 logits_r, ld_r, prob_e);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

size_t renet_topk_workspace(int n) {
    return (size_t)max(n, 1) * (TK_BINS * sizeof(unsigned) + sizeof(SelState));
}

int renet_topk_positive(const float* x, size_t ldx, int n, int M, int k, float* out_val, int64_t* out_idx,
                        void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 0 || M <= 0 || k <= 0 || k > M || ldx < (size_t)M || !x || !out_val || !out_idx) return RENET_ERR_BADARG;
    if (n == 0) return RENET_OK;
    if (!workspace || workspace_bytes < renet_topk_workspace(n)) return RENET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    unsigned* hist = (unsigned*)workspace;
    SelState* sel = (SelState*)((char*)workspace + (size_t)n * TK_BINS * sizeof(unsigned));
    hipError_t he = hipMemsetAsync(hist, 0, (size_t)n * TK_BINS * sizeof(unsigned), st);
    if (he != hipSuccess) return (int)he;
    RENET_LAUNCH(topk_init_kernel, dim3((n + 255) / 256), dim3(256), 0, st, sel, n, k);
    RENET_LAUNCH_CHECK();
    // slices per row: enough workgroups to fill the chip, at least ~8 k elements each
    int slices = max(1, min(1024, (int)(((size_t)M + 16383) / 16384)));
    while ((size_t)slices * n < 1024 && slices < 1024 && (size_t)M / slices > 4096) slices *= 2;
    const dim3 grid(slices, n);
    RENET_LAUNCH((topk_hist_kernel<0>), grid, dim3(TK_THREADS), 0, st, x, ldx, M, sel, hist);
    RENET_LAUNCH_CHECK();
    RENET_LAUNCH((topk_pick_kernel<0>), dim3(n), dim3(256), 0, st, sel, hist);
    RENET_LAUNCH_CHECK();
    RENET_LAUNCH((topk_hist_kernel<1>), grid, dim3(TK_THREADS), 0, st, x, ldx, M, sel, hist);
    RENET_LAUNCH_CHECK();
    RENET_LAUNCH((topk_pick_kernel<1>), dim3(n), dim3(256), 0, st, sel, hist);
    RENET_LAUNCH_CHECK();
    RENET_LAUNCH((topk_hist_kernel<2>), grid, dim3(TK_THREADS), 0, st, x, ldx, M, sel, hist);
    RENET_LAUNCH_CHECK();
    RENET_LAUNCH((topk_pick_kernel<2>), dim3(n), dim3(256), 0, st, sel, hist);
    RENET_LAUNCH_CHECK();
    RENET_LAUNCH(topk_collect_kernel, grid, dim3(TK_THREADS), 0, st, x, ldx, M, k, sel, out_val, out_idx);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

}  // extern "C"
