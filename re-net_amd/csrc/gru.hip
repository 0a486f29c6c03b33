// Fused GRU over the packed <= seq_len history window (torch.nn.GRU semantics: 1 layer, h0 = 0,
// gate order r,z,n; reference model.py:28-29,86,94 and global_model.py:25,49).
//
// The input projection Gi = X W_ih^T + b_ih is one large GEMM done by the caller.  The recurrence is ONE
// persistent launch per direction of time: a workgroup owns 16 sequences (sequences are independent,
// so there is no inter-workgroup traffic) and walks all their steps with the hidden state resident in
// LDS.  Per step the 16 x 3H recurrent product h W_hh^T runs on the f32-input MFMA
// (v_mfma_f32_16x16x4_f32, exact fp32): a wave owns blocks of 16 hidden units and accumulates the r, z
// and n gates of the same (sequence, unit) pairs in three accumulators that share one C layout, so the
// whole gate non-linearity is lane-local -- no G_h tensor, no per-step kernel boundary.
// W_hh (<= 1.9 MB) is streamed from L2 every step as float4 B-fragments: a group of 4 MFMA k-steps
// covers 16 consecutive k, lane (j, kq) holding k = 16*kg + 4*kq + {0..3} for both operands, which
// turns both fragment fetches into 16-byte loads (ds_read_b128 for h, global_load_dwordx4 for W_hh).
// The batch is length-sorted, so the sequences alive at step j are a prefix; rows past it are masked.
//
// Default arithmetic ("bf16x6", as in gemm_split.hip): W_hh is split ONCE per launch into three bf16 planes
// (k-padded to a multiple of 32), h / dGh are split into planes in LDS as they are produced, and every 16x16x32
// product runs as the six leading term pairs on v_mfma_f32_16x16x32_bf16 with fp32 accumulation -- fp32-class
// results at 2.7x the f32-input MFMA rate (the recurrence was matrix-pipe bound: 13 unit blocks x 156 f32 MFMAs
// per step on 4 SIMDs).  RENET_GEMM=f32 selects the exact-fp32 kernels (v_mfma_f32_16x16x4_f32).
#include <cstdlib>
#include <cstring>
#include "common.h"
#include <utility>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MT = 16;           // sequences per workgroup (MFMA M)
#ifndef RENET_GRU_NW
#define RENET_GRU_NW 8
#endif
constexpr int NW = RENET_GRU_NW; // waves per workgroup: 2 per SIMD so that W_hh fetch latency hides behind the partner's MFMAs
constexpr int NT = NW * 64;
constexpr int MAXL = 32;         // max packed steps (seq_len is 10 / 15 in the reference configs)

struct StepOff {
    int off[MAXL + 1];
};

// Up to MAXP independent GRUs run in ONE launch (blockIdx.y selects the problem); they may belong to up to MAXLAY
// different packed layouts: RE-Net's `encoder` and `encoder_r` consume the same batch (model.py:86,94), and the
// subject and object passes of a training step (train.py:136-137) are independent until their losses are added, so
// a step can run all four recurrences -- each only ~60 workgroups -- side by side on the 256 CUs.
constexpr int MAXP = 4;
constexpr int MAXLAY = 2;
struct Layouts {
    StepOff so[MAXLAY];
    int L[MAXLAY];
    int rows[MAXLAY];           // forward: rows of h_last (>= B); backward: B
    int lay_of[MAXP];
    int rot_mod;                // backward: number of distinct starting k groups of the W stream (0 = default; RENET_GRU_ROT)
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// Optional phase tracing of the persistent forward kernel (tools/gru_trace.py builds a separate library with
// -DRENET_GRU_TRACE; the shipped library contains none of this): s_memtime stamps per wave and step for the first
// GT_BLOCKS workgroups of problem 0.
#ifdef RENET_GRU_TRACE
constexpr int GT_BLOCKS = 32, GT_SLOTS = 8;
__device__ unsigned long long* g_gru_trace = nullptr;      // [GT_BLOCKS][NW][MAXL][GT_SLOTS]
__device__ __forceinline__ void gt_put(int wave, int step, int slot, unsigned long long v) {
    if (g_gru_trace && wave < NW && blockIdx.y == 0 && blockIdx.x < GT_BLOCKS && (threadIdx.x & 63) == 0)
        g_gru_trace[(((size_t)blockIdx.x * NW + wave) * MAXL + step) * GT_SLOTS + slot] = v;
}
#define GT_NOW() __builtin_amdgcn_s_memtime()
#define GT_PUT(wave, step, slot, v) gt_put(wave, step, slot, (unsigned long long)(v))
#else
#define GT_NOW() 0ull
#define GT_PUT(wave, step, slot, v)
#endif

template <int H>
struct Cfg {
    static constexpr int NUB = (H + 15) / 16;          // blocks of 16 hidden units
    static constexpr int KG = (H + 15) / 16;           // groups of 16 k over K = H
    static constexpr int LDH = NUB * 16 + 4;           // LDS row stride of the h / dh tile (16 B aligned)
    static constexpr int K3 = 3 * H;
    static constexpr int KG3 = (K3 + 15) / 16;         // groups of 16 k over K = 3H (backward)
    static constexpr int LDG = KG3 * 16 + 4;
};

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
struct FwdProb { const float* Gi; const float* Whh; const float* bhh; float* h_last; float* saved; };
struct FwdProbs { FwdProb p[MAXP]; };
struct BwdProb { const float* dh_last; const float* WhhT; const float* saved; float* dGi; float* dGh; };
struct BwdProbs { BwdProb p[MAXP]; };

template <int H>
__global__ __launch_bounds__(NT) void gru_fwd_kernel(FwdProbs ps, Layouts ly) {
    const int lay = ly.lay_of[blockIdx.y];
    const StepOff& so = ly.so[lay];
    const int L = ly.L[lay], out_rows = ly.rows[lay];
    if ((int)blockIdx.x * MT >= out_rows) return;
    using C = Cfg<H>;
    const float* __restrict__ Gi = ps.p[blockIdx.y].Gi;
    const float* __restrict__ Whh = ps.p[blockIdx.y].Whh;
    const float* __restrict__ bhh = ps.p[blockIdx.y].bhh;
    float* __restrict__ h_last = ps.p[blockIdx.y].h_last;
    float* __restrict__ saved = ps.p[blockIdx.y].saved;
    __shared__ __attribute__((aligned(16))) float Hs[MT * C::LDH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * MT;
    const int B = so.off[1] - so.off[0];
    for (int t = tid; t < MT * C::LDH; t += NT) Hs[t] = 0.f;          // h0 = 0 (and zero k-padding)
    __syncthreads();

    const int jj = lane & 15;          // B column / C column: hidden unit within the block
    const int kq = lane >> 4;          // k quad within a 16-k group; C rows 4*kq .. 4*kq+3
    const int ai = lane & 15;          // A row: sequence within the tile

    for (int j = 0; j < L; ++j) {
        const int p0 = so.off[j];
        const int bs = so.off[j + 1] - p0;
        if (i0 >= bs) break;                                            // whole tile finished (sorted batch)
        f32x4 hnew[(C::NUB + NW - 1) / NW];
#pragma unroll
        for (int q = 0; q < (C::NUB + NW - 1) / NW; ++q) {
            const int ub = wave + NW * q;
            if (ub < C::NUB) {
                const int u = ub * 16 + jj;                             // this lane's hidden unit
                const bool uok = u < H;
                f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = ar, an = ar;
                const float* wr = Whh + (size_t)(uok ? u : 0) * H;
                const float* wz = wr + (size_t)H * H;
                const float* wn = wz + (size_t)H * H;
#pragma unroll 4
                for (int kg = 0; kg < C::KG; ++kg) {
                    const int k = kg * 16 + 4 * kq;
                    const float4 a = *reinterpret_cast<const float4*>(&Hs[ai * C::LDH + k]);
                    float4 br = make_float4(0.f, 0.f, 0.f, 0.f), bz = br, bn = br;
                    if (uok && k < H) {
                        br = *reinterpret_cast<const float4*>(wr + k);
                        bz = *reinterpret_cast<const float4*>(wz + k);
                        bn = *reinterpret_cast<const float4*>(wn + k);
                    }
                    ar = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, br.x, ar, 0, 0, 0);
                    az = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bz.x, az, 0, 0, 0);
                    an = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bn.x, an, 0, 0, 0);
                    ar = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, br.y, ar, 0, 0, 0);
                    az = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bz.y, az, 0, 0, 0);
                    an = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bn.y, an, 0, 0, 0);
                    ar = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, br.z, ar, 0, 0, 0);
                    az = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bz.z, az, 0, 0, 0);
                    an = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bn.z, an, 0, 0, 0);
                    ar = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, br.w, ar, 0, 0, 0);
                    az = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bz.w, az, 0, 0, 0);
                    an = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bn.w, an, 0, 0, 0);
                }
                // C layout: column = lane & 15 (unit u), row = 4 * (lane >> 4) + reg (sequence)
                const float b_r = uok ? bhh[u] : 0.f, b_z = uok ? bhh[H + u] : 0.f, b_n = uok ? bhh[2 * H + u] : 0.f;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int i = 4 * kq + reg;
                    const float hp = Hs[i * C::LDH + (uok ? u : 0)];
                    float hv = hp;
                    if (uok && i0 + i < bs) {
                        const size_t p = (size_t)(p0 + i0 + i);
                        const float* gi = Gi + p * C::K3;
                        const float hn = an[reg] + b_n;
                        const float r = sigmoidf_(gi[u] + ar[reg] + b_r);
                        const float z = sigmoidf_(gi[H + u] + az[reg] + b_z);
                        const float n = tanhf(gi[2 * H + u] + r * hn);
                        hv = (1.f - z) * n + z * hp;
                        float* sv = saved + p * 5 * H;
                        sv[u] = r; sv[H + u] = z; sv[2 * H + u] = n; sv[3 * H + u] = hn; sv[4 * H + u] = hp;
                    }
                    hnew[q][reg] = hv;
                }
            }
        }
        __syncthreads();                                                // every wave is done reading Hs
#pragma unroll
        for (int q = 0; q < (C::NUB + NW - 1) / NW; ++q) {
            const int ub = wave + NW * q;
            const int u = ub * 16 + jj;
            if (ub < C::NUB && u < H) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) Hs[(4 * kq + reg) * C::LDH + u] = hnew[q][reg];
            }
        }
        __syncthreads();
    }
    (void)B;
    for (int t = tid; t < MT * H; t += NT) {                  // rows >= B were never touched: still h0 = 0
        const int i = t / H, u = t - i * H;
        if (i0 + i < out_rows) h_last[(size_t)(i0 + i) * H + u] = Hs[i * C::LDH + u];
    }
}

// ---------------------------------------------------------------------------------------------
// backward (BPTT): dh lives in LDS; per step the gate gradients are formed element-wise, written out
// as dGi / dGh rows (the caller turns them into dW_ih, dW_hh, dX with large GEMMs) and dGh is kept in
// LDS as the A operand of   dh_prev = dh * z + dGh W_hh   (K = 3H, B fragments from W_hh^T [H, 3H]).
// ---------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(NT) void gru_bwd_kernel(BwdProbs ps, Layouts ly) {
    const int lay = ly.lay_of[blockIdx.y];
    const StepOff& so = ly.so[lay];
    const int L = ly.L[lay];
    if ((int)blockIdx.x * MT >= ly.rows[lay]) return;
    using C = Cfg<H>;
    const float* __restrict__ dh_last = ps.p[blockIdx.y].dh_last;
    const float* __restrict__ WhhT = ps.p[blockIdx.y].WhhT;          // [H, 3H]
    const float* __restrict__ saved = ps.p[blockIdx.y].saved;
    float* __restrict__ dGi = ps.p[blockIdx.y].dGi;
    float* __restrict__ dGh = ps.p[blockIdx.y].dGh;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dHs = smem;                         // [MT][LDH]
    float* Gs = smem + MT * C::LDH;            // [MT][LDG]  dGh tile (k-padded with zeros)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * MT;
    const int B = so.off[1] - so.off[0];
    for (int t = tid; t < MT * C::LDH; t += NT) {
        const int i = t / C::LDH, u = t - i * C::LDH;
        dHs[t] = (u < H && i0 + i < B) ? dh_last[(size_t)(i0 + i) * H + u] : 0.f;
    }
    for (int t = tid; t < MT * C::LDG; t += NT) Gs[t] = 0.f;
    __syncthreads();
    const int jj = lane & 15, kq = lane >> 4, ai = lane & 15;

    for (int j = L - 1; j >= 0; --j) {
        const int p0 = so.off[j];
        const int bs = so.off[j + 1] - p0;
        if (i0 >= bs) continue;                                         // tile not alive yet at this step
        // phase 1: gate gradients of the live rows
        for (int t = tid; t < MT * H; t += NT) {
            const int i = t / H, u = t - i * H;
            float gr = 0.f, gz = 0.f, gn = 0.f;
            if (i0 + i < bs) {
                const size_t p = (size_t)(p0 + i0 + i);
                const float* sv = saved + p * 5 * H;
                const float r = sv[u], z = sv[H + u], n = sv[2 * H + u], hn = sv[3 * H + u], hp = sv[4 * H + u];
                const float g = dHs[i * C::LDH + u];
                const float dan = g * (1.f - z) * (1.f - n * n);
                const float daz = g * (hp - n) * z * (1.f - z);
                const float dar = dan * hn * r * (1.f - r);
                float* gi = dGi + p * C::K3;
                float* gh = dGh + p * C::K3;
                gi[u] = dar; gi[H + u] = daz; gi[2 * H + u] = dan;
                gr = dar; gz = daz; gn = dan * r;
                gh[u] = gr; gh[H + u] = gz; gh[2 * H + u] = gn;
                dHs[i * C::LDH + u] = g * z;                            // direct path h_prev -> h
            }
            Gs[i * C::LDG + u] = gr; Gs[i * C::LDG + H + u] = gz; Gs[i * C::LDG + 2 * H + u] = gn;
        }
        __syncthreads();
        if (j > 0) {
            // phase 2: dh_prev += dGh W_hh  (rows of dead sequences have dGh = 0 and keep their dh)
#pragma unroll
            for (int q = 0; q < (C::NUB + NW - 1) / NW; ++q) {
                const int ub = wave + NW * q;
                if (ub < C::NUB) {
                    const int u = ub * 16 + jj;
                    const bool uok = u < H;
                    f32x4 acc;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) acc[reg] = dHs[(4 * kq + reg) * C::LDH + (uok ? u : 0)];
                    const float* wt = WhhT + (size_t)(uok ? u : 0) * C::K3;
#pragma unroll 4
                    for (int kg = 0; kg < C::KG3; ++kg) {
                        const int k = kg * 16 + 4 * kq;
                        const float4 a = *reinterpret_cast<const float4*>(&Gs[ai * C::LDG + k]);
                        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (uok && k < C::K3) b = *reinterpret_cast<const float4*>(wt + k);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
                    }
                    if (uok) {
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) dHs[(4 * kq + reg) * C::LDH + u] = acc[reg];
                    }
                }
            }
            __syncthreads();
        }
    }
}


// ---------------------------------------------------------------------------------------------
// bf16x6 variants
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Planes3 {
    __bf16 p[3];
};

// x = p0 + p1 + p2 (round-to-nearest terms, as gemm_split.hip)
__device__ __forceinline__ Planes3 split3(float x) {
    Planes3 r;
    r.p[0] = (__bf16)x;
    const float r1 = x - (float)r.p[0];
    r.p[1] = (__bf16)r1;
    r.p[2] = (__bf16)(r1 - (float)r.p[1]);
    return r;
}

template <int H>
struct BCfg {
    static constexpr int KG = (H + 31) / 32;           // groups of 32 k over K = H (forward)
    static constexpr int KP = KG * 32;                 // padded K of the W_hh planes
    static constexpr int LDP = KP + 8;                 // bf16 row stride of the h planes in LDS (16 B aligned)
    static constexpr int KG3 = (3 * H + 31) / 32;      // K = 3H (backward)
    static constexpr int KP3 = KG3 * 32;
    static constexpr int LDP3 = KP3 + 8;
};

// W -> bf16 planes in FRAGMENT order: chunk (ub, kg, g, p) = the 64 x 16 bytes that the 64 lanes of a wave load
// as the B operand (16 units x 32 k) of unit block ub, k group kg, gate g, plane p -- one fully coalesced 1 KB
// global_load_dwordx4 per chunk (row-major planes made every such load touch 16 half-used 128-byte lines, and
// the kernels were bound by the L1's line rate).  Element (unit u, k) of gate g = in[g * sg + u * su + k * sk];
// units >= U and k >= K are zero.   out index = (((ub * KG + kg) * G + g) * 3 + p) * 64 + lane  (x 8 bf16)
__global__ __launch_bounds__(256) void split_frag_kernel(const float* __restrict__ in, int U, int K, int G, size_t sg,
                                                         size_t su, size_t sk, int NUBk, int KGk, int NPL,
                                                         bf16x8* __restrict__ out) {
    const int total = NUBk * KGk * G * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63;
        const int g = (i >> 6) % G;
        const int kg = ((i >> 6) / G) % KGk;
        const int ub = ((i >> 6) / G) / KGk;
        const int u = ub * 16 + (lane & 15), k0 = kg * 32 + (lane >> 4) * 8;
        bf16x8 o[3];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (u < U && k0 + e < K) ? in[g * sg + u * su + (size_t)(k0 + e) * sk] : 0.f;
            const Planes3 t = split3(x);
#pragma unroll
            for (int p = 0; p < 3; ++p) o[p][e] = t.p[p];
        }
        // NPL = 3: the bf16x6 planes; NPL = 1 (bf16 mode): plane 0 only = rne(x)
        const size_t base = ((size_t)((ub * KGk + kg) * G + g) * NPL) * 64 + lane;
#pragma unroll
        for (int p = 0; p < 3; ++p)
            if (p < NPL) out[base + (size_t)p * 64] = o[p];
    }
}

// six leading term pairs of a (16 x 32) x (32 x 16) product, smallest first
__device__ __forceinline__ f32x4 mfma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
    return acc;
}

// NPL = 3: the six leading term pairs (fp32-class); NPL = 1: ONE bf16 product (bf16 mode, BASELINE config 5)
template <int NPL>
__device__ __forceinline__ f32x4 mfma_p(const bf16x8 (&a)[NPL], const bf16x8 (&b)[NPL], f32x4 acc) {
    if constexpr (NPL == 3) return mfma6(a, b, acc);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
}

struct FwdProbB { const float* Gi; const bf16x8* Wp; const float* bhh; float* h_last; float* saved; };
struct FwdProbsB { FwdProbB p[MAXP]; };
struct BwdProbB { const float* dh_last; const bf16x8* WTp; const float* saved; float* dGi; float* dGh; int out_ld;
                  float* bound; /* optional: bound[blockIdx.x] = this workgroup's max |dGi| (f16x3 GEMM operand bound) */ };
struct BwdProbsB { BwdProbB p[MAXP]; };

// Wp: bf16 planes of W_hh in fragment order (split_frag_kernel with G = 3 gates)
// WV: waves per workgroup (default NW = 8; the one-plane bf16 kernels run 16: their steps are latency-, not
// register-bound, and a 256-workgroup launch has ONE workgroup per CU)
template <int H, int NPL, int WV = NW>
__global__ __launch_bounds__(WV * 64) void gru_fwd_bf_kernel(FwdProbsB ps, Layouts ly) {
    constexpr int NTW = WV * 64;
    const int lay = ly.lay_of[blockIdx.y];
    const StepOff& so = ly.so[lay];
    const int L = ly.L[lay], out_rows = ly.rows[lay];
    if ((int)blockIdx.x * MT >= out_rows) return;
    using C = Cfg<H>;
    using Bc = BCfg<H>;
    const float* __restrict__ Gi = ps.p[blockIdx.y].Gi;
    const bf16x8* __restrict__ Wp = ps.p[blockIdx.y].Wp;
    const float* __restrict__ bhh = ps.p[blockIdx.y].bhh;
    float* __restrict__ h_last = ps.p[blockIdx.y].h_last;
    float* __restrict__ saved = ps.p[blockIdx.y].saved;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Hs = smem;                                                    // [MT][LDH] h of the current step (fp32)
    float* Hn = smem + MT * C::LDH;                                      // [MT][LDH] h being produced
    __bf16* Hp = reinterpret_cast<__bf16*>(smem + 2 * MT * C::LDH);      // [NPL][MT][LDP] bf16 planes of Hs
    // (the wave index through readfirstlane: everything derived from it -- unit block, fragment addresses -- is then
    // known to be wave-uniform and lives in scalar registers)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i0 = blockIdx.x * MT;
    for (int t = tid; t < 2 * MT * C::LDH; t += NTW) Hs[t] = 0.f;      // h0 = 0
    for (int t = tid; t < NPL * MT * Bc::LDP / 2; t += NTW) reinterpret_cast<unsigned*>(Hp)[t] = 0u;   // and its planes
    __syncthreads();
    const int jj = lane & 15, kq = lane >> 4, ai = lane & 15;
    // Every workgroup streams the SAME W_hh planes from L2 every step, and workgroups that start together run the
    // same schedule: their requests for one chunk arrive at one L2 channel together (traced: the k loop of a unit
    // block took 14 k cycles against 2 k of MFMA time).  The 32 workgroups that share an XCD's L2 (the dispatcher
    // deals workgroups round-robin: XCD = linear block id mod 8) therefore start at different (k group, unit block)
    // positions of the same cyclic order: 41.0 k -> 31.1 k cycles per step at H = 200, 124.5 k -> 91.8 k at H = 400.
    // (The order of the fp32 accumulation over k depends on the workgroup: deterministic, not row-order invariant.)
    const int rot_id = (int)((blockIdx.x + gridDim.x * blockIdx.y) >> 3);
    // H = 400: the planes of the launch's GRUs (2 x 3 MB) exceed an XCD's 4 MB L2 -- there only the k position is
    // rotated (workgroups stay on the same unit block, whose chunks are then fetched into L2 once): rotating the
    // unit blocks as well measured 101.8 k instead of 91.8 k cycles per step
    const int rot_k = rot_id % Bc::KG, rot_u = H <= 200 ? (rot_id / Bc::KG) % C::NUB : 0;
    // W_hh fragments through a register ring, PF chunks ahead of the MFMAs that consume them (left to itself the compiler
    // re-uses four register quads and keeps 1-3 loads in flight: the loop then runs at one L2 latency per k group,
    // 8.5 k cycles per unit block against 2 k of MFMA time).  A chunk = the NPL plane fragments of one (k group, gate).
    // Round 6: the ring runs CONTINUOUSLY over the unit blocks and time steps of a wave -- the stream does not depend on
    // h, so the first PF chunks of the wave's NEXT unit block (of the next time step after its last one) are requested
    // during the last chunks of the current one and fly under the gate epilogue, the barriers and the plane split.  A
    // unit block is VP = roundup(CH, RS) virtual chunks long (the surplus ones carry neither loads nor MFMAs), which
    // keeps every ring slot index a compile-time constant.
    // Ring depth: (PF_kgroups + 1) * 3 chunks with <= 8 waves (256 registers per wave); 5 chunks when the workgroup has
    // one wave per unit block (13 waves at H = 200: 4 waves on a SIMD, 128 registers each).
    constexpr int CH = Bc::KG * 3;                                      // chunks per unit block
    constexpr int RS = (NPL == 3 && WV > 8) ? 5 : ((Bc::KG > 8 ? 2 : (Bc::KG > 3 ? 3 : Bc::KG - 1)) + 1) * 3;
    constexpr int PF = RS - 1, VP = (CH + RS - 1) / RS * RS;
    static_assert(PF <= CH, "ring deeper than a unit block");
    bf16x8 wb[RS][NPL];                                                 // [slot][plane]
    auto rotc = [&](int c) {                                            // chunk -> its position in the fragment stream
        const int kg = c / 3 + rot_k;
        return (kg >= Bc::KG ? kg - Bc::KG : kg) * 3 + c % 3;
    };
    // fragment order (split_frag_kernel): chunk = NPL consecutive 1 KB fragments.  Buffer loads: the descriptor and the
    // chunk's byte offset are wave-uniform (scalar registers), the only address VGPR is lane * 16 -- with flat pointers
    // every chunk in flight kept its own 64-bit address pair alive (74 spilled registers in the 13-wave kernel)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16x8*>(Wp), (short)0, (int)((size_t)C::NUB * CH * NPL * 1024), 0x00020000);
    const int lane16 = lane * 16;
    auto frag_base = [&](int ub0) {                                     // byte offset of a unit block's fragments
        const int ub = ub0 + rot_u >= C::NUB ? ub0 + rot_u - C::NUB : ub0 + rot_u;
        return ub * (CH * NPL * 1024);
    };
    auto frag = [&](int base, int c, int p) {
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        const u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16 + p * 1024, base + rotc(c) * (NPL * 1024), 0);
        return __builtin_bit_cast(bf16x8, v);
    };
    if (wave < C::NUB) {
        const int wf0 = frag_base(wave);
#pragma unroll
        for (int q = 0; q < PF; ++q)
#pragma unroll
            for (int p = 0; p < NPL; ++p) wb[q][p] = frag(wf0, q, p);
    }

    for (int j = 0; j < L; ++j) {
        const int p0 = so.off[j];
        const int bs = so.off[j + 1] - p0;
        if (i0 >= bs) break;                                            // whole tile finished (sorted batch)
        GT_PUT(wave, j, 0, GT_NOW());
        unsigned long long gt_mfma = 0, gt_epi = 0;
        (void)gt_mfma; (void)gt_epi;
#pragma unroll 1
        for (int ub0 = wave; ub0 < C::NUB; ub0 += WV) {
            const unsigned long long gt_a = GT_NOW();
            const int ub = ub0 + rot_u >= C::NUB ? ub0 + rot_u - C::NUB : ub0 + rot_u;
            const int u = ub * 16 + jj;                                 // this lane's hidden unit
            const bool uok = u < H;
            f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = ar, an = ar;
            // input-gate pre-activations of this lane's four (sequence, unit) pairs: requested BEFORE the matrix work
            // and unconditionally (dead rows read the tile's last live row).  Loaded inside the `row is alive` branch
            // of the epilogue they cost one full memory latency per row (hipcc drains vmcnt at every branch merge):
            // 8 serialised latencies per wave and step
            const int uc = uok ? u : 0;
            float gr[4], gz[4], gn[4];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = min(i0 + 4 * kq + reg, bs - 1);
                const float* gi = Gi + (size_t)(p0 + row) * C::K3 + uc;
                gr[reg] = gi[0]; gz[reg] = gi[H]; gn[reg] = gi[2 * H];
            }
            const float b_r = bhh[uc], b_z = bhh[H + uc], b_n = bhh[2 * H + uc];
            const int wf = frag_base(ub0);
            const int wfn = frag_base(ub0 + WV < C::NUB ? ub0 + WV : wave);         // this wave's next unit block
            const __bf16* ha = Hp + ai * Bc::LDP + kq * 8;
            bf16x8 a[NPL];
#pragma unroll
            for (int v = 0; v < VP; ++v) {
                const int tq = v + PF;                                  // virtual chunk requested now
#ifndef RENET_GRU_NOLOAD                                                // (ablation builds of tools/gru_trace.py)
                if (tq < CH) {
#pragma unroll
                    for (int p = 0; p < NPL; ++p) wb[tq % RS][p] = frag(wf, tq, p);
                } else if (tq >= VP) {
#pragma unroll
                    for (int p = 0; p < NPL; ++p) wb[tq % RS][p] = frag(wfn, tq - VP, p);
                }
#endif
                if (v < CH) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (v % 3 == 0) {
                        const int kg = v / 3 + rot_k >= Bc::KG ? v / 3 + rot_k - Bc::KG : v / 3 + rot_k;
#pragma unroll
                        for (int p = 0; p < NPL; ++p) a[p] = *reinterpret_cast<const bf16x8*>(ha + p * MT * Bc::LDP + kg * 32);
                    }
#ifdef RENET_GRU_NOMFMA
                    f32x4& acc_ = v % 3 == 0 ? ar : (v % 3 == 1 ? az : an);
                    acc_[0] += (float)a[0][v % 3] * (float)wb[v % RS][0][0] + (float)wb[v % RS][NPL - 1][7];
#else
                    if (v % 3 == 0) ar = mfma_p<NPL>(a, wb[v % RS], ar);
                    else if (v % 3 == 1) az = mfma_p<NPL>(a, wb[v % RS], az);
                    else an = mfma_p<NPL>(a, wb[v % RS], an);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // C layout: column = lane & 15 (unit u), row = 4 * (lane >> 4) + reg (sequence)
#ifdef RENET_GRU_TRACE
            asm volatile("" : "+v"(ar), "+v"(az), "+v"(an));            // the MFMA results exist here
#endif
            const unsigned long long gt_b = GT_NOW();
            gt_mfma += gt_b - gt_a;
            if (uok) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int i = 4 * kq + reg;
                    const float hp = Hs[i * C::LDH + u];
                    float hv = hp;
                    if (i0 + i < bs) {
                        const size_t p = (size_t)(p0 + i0 + i);
                        const float hn = an[reg] + b_n;
                        const float r = sigmoidf_(gr[reg] + ar[reg] + b_r);
                        const float z = sigmoidf_(gz[reg] + az[reg] + b_z);
                        const float n = tanhf(gn[reg] + r * hn);
                        hv = (1.f - z) * n + z * hp;
                        float* sv = saved + p * 5 * H;
                        sv[u] = r; sv[H + u] = z; sv[2 * H + u] = n; sv[3 * H + u] = hn; sv[4 * H + u] = hp;
                    }
                    Hn[i * C::LDH + u] = hv;
                }
            }
            gt_epi += GT_NOW() - gt_b;
        }
        GT_PUT(wave, j, 1, gt_mfma);
        GT_PUT(wave, j, 2, gt_epi);
        GT_PUT(wave, j, 3, GT_NOW());
        __syncthreads();                                                // every wave is done reading Hs / Hp
        GT_PUT(wave, j, 4, GT_NOW());
        for (int t = tid; t < MT * H; t += NTW) {
            const int i = t / H, u = t - i * H;
            const float v = Hn[i * C::LDH + u];
            Hs[i * C::LDH + u] = v;
            const Planes3 s = split3(v);
#pragma unroll
            for (int p = 0; p < NPL; ++p) Hp[p * MT * Bc::LDP + i * Bc::LDP + u] = s.p[p];
        }
        GT_PUT(wave, j, 5, GT_NOW());
        __syncthreads();
        GT_PUT(wave, j, 6, GT_NOW());
    }
    for (int t = tid; t < MT * H; t += NTW) {                  // rows >= B were never touched: still h0 = 0
        const int i = t / H, u = t - i * H;
        if (i0 + i < out_rows) h_last[(size_t)(i0 + i) * H + u] = Hs[i * C::LDH + u];
    }
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for_n(F&& f) {                   // f(integral_constant<int, 0>) ... f(<N - 1>)
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// ---------------------------------------------------------------------------------------------
// Round 6: the bf16x6 forward recurrence as ONE continuous W_hh stream (H <= 200).
//
// What bounded gru_fwd_bf_kernel (tools/gru_trace.py + its NOLOAD / NOMFMA builds, profiles/r06_f_gru.md): a workgroup
// pulls its 0.8 MB of W_hh planes through the CU's L1 at exactly the L1 fill rate (64 B/clk: 12.8 k cycles per step,
// the same with 16 or with 256 workgroups on the chip -- it is not an L2 limit), but only while its waves are inside
// their k loops; the gate epilogues, the two barriers and the plane split (another ~12 k cycles per step) ran with the
// L1 idle, and with 13 unit blocks on 8 waves the step waited for the five waves that own two blocks.
// Here (a) the work items are (unit block, gate) pairs -- 39 at H = 200, five per wave -- so every wave streams the
// same number of bytes; the gate pre-activations are exchanged through LDS and the gate epilogue is one flat pass of
// all 512 threads that also writes the next step's planes (the old copy + split pass); (b) the fragments go through
// a 12-chunk register ring that never drains: a step's chunk list is the same at every step, so the ring simply
// wraps -- the first 11 chunks of step t + 1 are requested during the last chunks of step t and fly under its
// epilogue and barriers.  Per step: one k loop of 7 k groups x 5 items (5 independent accumulator chains per wave).
// ---------------------------------------------------------------------------------------------
template <int H>
struct XCfg {
    static constexpr int WV = 8;
    static constexpr int NTW = WV * 64;
    static constexpr int NIT = (3 * Cfg<H>::NUB + WV - 1) / WV;        // items per wave (5 at H = 200; the last wave: 4)
    static constexpr int KG = BCfg<H>::KG;
    static constexpr int CHW = KG * NIT;                                // chunks per wave and step
    static constexpr int RS = 8, PF = RS - 1;                           // 7 chunks = 21 KB in flight per wave (7 or >= 9 slots: spills, and every spill reload drains vmcnt)
    static constexpr int VPW = (CHW + RS - 1) / RS * RS;                // virtual chunks per step (ring period)
    static constexpr int HP = Cfg<H>::NUB * 16;                         // padded units per gate in the exchange tile
    static constexpr int LDG = 3 * HP + 4;                              // fp32 row stride of the exchange tile
    static constexpr size_t lds_bytes() {
        return ((size_t)MT * Cfg<H>::LDH + (size_t)MT * LDG + 3 * H) * sizeof(float) +
               (size_t)3 * MT * BCfg<H>::LDP * sizeof(__bf16);
    }
};

template <int H>
__global__ __launch_bounds__(XCfg<H>::NTW) void gru_fwd_x_kernel(FwdProbsB ps, Layouts ly) {
    using C = Cfg<H>;
    using Bc = BCfg<H>;
    using X = XCfg<H>;
    constexpr int NPL = 3, NTW = X::NTW, NIT = X::NIT, RS = X::RS, PF = X::PF, CHW = X::CHW, VPW = X::VPW;
    static_assert(PF <= CHW, "ring deeper than a step");
    const int lay = ly.lay_of[blockIdx.y];
    const StepOff& so = ly.so[lay];
    const int L = ly.L[lay], out_rows = ly.rows[lay];
    if ((int)blockIdx.x * MT >= out_rows) return;
    const float* __restrict__ Gi = ps.p[blockIdx.y].Gi;
    const bf16x8* __restrict__ Wp = ps.p[blockIdx.y].Wp;
    const float* __restrict__ bhh = ps.p[blockIdx.y].bhh;
    float* __restrict__ h_last = ps.p[blockIdx.y].h_last;
    float* __restrict__ saved = ps.p[blockIdx.y].saved;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Hs = smem;                                                    // [MT][LDH]  h (fp32), updated in place
    float* Gh = Hs + MT * C::LDH;                                        // [MT][LDG]  h W_hh^T of the step (r | z | n)
    float* Bs = Gh + MT * X::LDG;                                        // [3H]       b_hh
    __bf16* Hp = reinterpret_cast<__bf16*>(Bs + 3 * H);                  // [3][MT][LDP] bf16 planes of Hs
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i0 = blockIdx.x * MT;
    for (int t = tid; t < MT * C::LDH; t += NTW) Hs[t] = 0.f;           // h0 = 0
    for (int t = tid; t < 3 * H; t += NTW) Bs[t] = bhh[t];
    for (int t = tid; t < NPL * MT * Bc::LDP / 2; t += NTW) reinterpret_cast<unsigned*>(Hp)[t] = 0u;
    __syncthreads();
    const int jj = lane & 15, kq = lane >> 4;
    // workgroups that share an XCD's L2 start at different k groups / unit blocks of the same cyclic order (see
    // gru_fwd_bf_kernel: L2 channel hot-spotting)
    const int rot_id = (int)((blockIdx.x + gridDim.x * blockIdx.y) >> 3);
    const int rot_k = rot_id % X::KG, rot_u = (rot_id / X::KG) % C::NUB;
    // this wave's items: (unit block, gate) pairs wave, wave + 8, ...; item_off = byte offset of the item's first fragment,
    // item_col = its first column in the exchange tile
    int item_off[NIT], item_col[NIT];
    bool item_on[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int id = wave + X::WV * it;
        item_on[it] = id < 3 * C::NUB;
        const int idc = item_on[it] ? id : 3 * C::NUB - 1;              // a surplus slot streams (and discards) a real item:
        const int ub0 = idc / 3, g = idc % 3;                           // no branch inside the k loop
        const int ub = ub0 + rot_u >= C::NUB ? ub0 + rot_u - C::NUB : ub0 + rot_u;
        item_off[it] = ((ub * X::KG) * 3 + g) * (NPL * 1024);           // fragment order of split_frag_kernel
        item_col[it] = g * X::HP + ub * 16;
    }
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16x8*>(Wp), (short)0, (int)((size_t)C::NUB * X::KG * 3 * NPL * 1024), 0x00020000);
    const int lane16 = lane * 16;
    auto rotk = [&](int q) { return q + rot_k >= X::KG ? q + rot_k - X::KG : q + rot_k; };
    bf16x8 wb[RS][NPL];
    // chunk c of a step = (k group c / NIT in rotated order, item c % NIT)
    auto request = [&](auto slot, int c) {
        constexpr int S = decltype(slot)::value;
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        const int soff = item_off[c % NIT] + rotk(c / NIT) * (3 * NPL * 1024);
#pragma unroll
        for (int p = 0; p < NPL; ++p)
            wb[S][p] = __builtin_bit_cast(bf16x8, (u32x4_)__builtin_amdgcn_raw_buffer_load_b128(wrs, lane16 + p * 1024, soff, 0));
    };
    {
        auto pro = [&](auto q) { request(q, decltype(q)::value); };
        static_for_n<PF>(pro);
    }
    constexpr int P1 = (MT * H + NTW - 1) / NTW;                        // (sequence, unit) pairs per thread in the epilogue

    for (int j = 0; j < L; ++j) {
        const int p0 = so.off[j];
        const int bs = so.off[j + 1] - p0;
        if (i0 >= bs) break;                                            // whole tile finished (sorted batch)
        GT_PUT(wave, j, 0, GT_NOW());
        // the input-gate pre-activations of this thread's epilogue pairs: requested before the matrix work (dead rows
        // read the tile's last live row).  (Staging them in LDS by DMA instead cost a drained ring per step: the
        // waitcnt pass orders the next LDS access behind vmcnt(0) whatever the DMA's destination object.)
        float gir[P1], giz[P1], gin[P1];
#pragma unroll
        for (int q = 0; q < P1; ++q) {
            const int e = min(tid + NTW * q, MT * H - 1);
            const int i = e / H, u = e - i * H;
            const float* gi = Gi + (size_t)(p0 + min(i0 + i, bs - 1)) * C::K3 + u;
            gir[q] = gi[0]; giz[q] = gi[H]; gin[q] = gi[2 * H];
        }
        f32x4 acc[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) acc[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        const __bf16* ha = Hp + jj * Bc::LDP + kq * 8;
        bf16x8 a[NPL];
        auto body = [&](auto vc) {
            constexpr int v = decltype(vc)::value;
            constexpr int tq = v + PF;                                  // virtual chunk requested now
#ifndef RENET_GRU_NOLOAD
            if constexpr (tq < CHW) request(std::integral_constant<int, tq % RS>{}, tq);
            else if constexpr (tq >= VPW && tq - VPW < CHW) request(std::integral_constant<int, tq % RS>{}, tq - VPW);
#endif
            if constexpr (v < CHW) {
                constexpr int it = v % NIT;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (it == 0) {
                    const int kg = rotk(v / NIT);
#pragma unroll
                    for (int p = 0; p < NPL; ++p) a[p] = *reinterpret_cast<const bf16x8*>(ha + p * MT * Bc::LDP + kg * 32);
                }
#ifdef RENET_GRU_NOMFMA
                acc[it][0] += (float)a[0][it] * (float)wb[v % RS][0][0] + (float)wb[v % RS][NPL - 1][7];
#else
                acc[it] = mfma6(a, wb[v % RS], acc[it]);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        static_for_n<VPW>(body);
        GT_PUT(wave, j, 1, GT_NOW());
        // C layout: column = lane & 15 (unit), row = 4 * (lane >> 4) + reg (sequence)
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (item_on[it]) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) Gh[(4 * kq + reg) * X::LDG + item_col[it] + jj] = acc[it][reg];
            }
        GT_PUT(wave, j, 3, GT_NOW());
        __syncthreads();
        GT_PUT(wave, j, 4, GT_NOW());
#pragma unroll
        for (int q = 0; q < P1; ++q) {
            const int e = tid + NTW * q;
            if (e < MT * H) {
                const int i = e / H, u = e - i * H;
                if (i0 + i < bs) {
                    const float* gh = Gh + i * X::LDG + u;
                    const float hp = Hs[i * C::LDH + u];
                    const float hn = gh[2 * X::HP] + Bs[2 * H + u];
                    const float r = sigmoidf_(gir[q] + gh[0] + Bs[u]);
                    const float z = sigmoidf_(giz[q] + gh[X::HP] + Bs[H + u]);
                    const float n = tanhf(gin[q] + r * hn);
                    const float hv = (1.f - z) * n + z * hp;
                    float* sv = saved + (size_t)(p0 + i0 + i) * 5 * H;
                    sv[u] = r; sv[H + u] = z; sv[2 * H + u] = n; sv[3 * H + u] = hn; sv[4 * H + u] = hp;
                    Hs[i * C::LDH + u] = hv;
                    const Planes3 s = split3(hv);
#pragma unroll
                    for (int p = 0; p < NPL; ++p) Hp[p * MT * Bc::LDP + i * Bc::LDP + u] = s.p[p];
                }
            }
            if (q % 4 == 3) __builtin_amdgcn_sched_barrier(0);      // a few pairs at a time: the ring's registers stay live through this pass
        }
        GT_PUT(wave, j, 5, GT_NOW());
        __syncthreads();
        GT_PUT(wave, j, 6, GT_NOW());
    }
    for (int t = tid; t < MT * H; t += NTW) {                  // rows >= B were never touched: still h0 = 0
        const int i = t / H, u = t - i * H;
        if (i0 + i < out_rows) h_last[(size_t)(i0 + i) * H + u] = Hs[i * C::LDH + u];
    }
}

// WTp: bf16 planes of W_hh^T (unit = hidden unit, k over the 3H gate columns) in fragment order (G = 1)
// OUT16: dGi / dGh are written as bf16 (RNE) into matrices with row stride out_ld (elements): the operand format of
// the bf16-storage GEMMs that consume them (dW_ih, dX, dW_hh); the recurrence itself keeps its fp32 values in LDS.
template <int H, int NPL, bool OUT16 = false, int WV = NW>
__global__ __launch_bounds__(WV * 64) void gru_bwd_bf_kernel(BwdProbsB ps, Layouts ly) {
    constexpr int NTW = WV * 64;
    const int lay = ly.lay_of[blockIdx.y];
    const StepOff& so = ly.so[lay];
    const int L = ly.L[lay];
    float* __restrict__ bound = ps.p[blockIdx.y].bound;
    if ((int)blockIdx.x * MT >= ly.rows[lay]) {
        if (bound && threadIdx.x == 0) bound[blockIdx.x] = 0.f;
        return;
    }
    constexpr bool TRACK = NPL == 3;              // only the bf16x6 kernels feed f16x3 GEMMs (the bf16-storage ones don't)
    __shared__ float gred[16];
    float gmax = 0.f;                                                     // max |dGi| written by this thread
    using C = Cfg<H>;
    using Bc = BCfg<H>;
    const float* __restrict__ dh_last = ps.p[blockIdx.y].dh_last;
    const bf16x8* __restrict__ WTp = ps.p[blockIdx.y].WTp;
    const float* __restrict__ saved = ps.p[blockIdx.y].saved;
    float* __restrict__ dGi = ps.p[blockIdx.y].dGi;
    float* __restrict__ dGh = ps.p[blockIdx.y].dGh;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dHs = smem;                                                   // [MT][LDH] fp32
    __bf16* Gp = reinterpret_cast<__bf16*>(smem + MT * C::LDH);          // [3][MT][LDP3] planes of the dGh tile
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: scalar registers)
    const int i0 = blockIdx.x * MT;
    const int B = so.off[1] - so.off[0];
    for (int t = tid; t < MT * C::LDH; t += NTW) {
        const int i = t / C::LDH, u = t - i * C::LDH;
        dHs[t] = (u < H && i0 + i < B) ? dh_last[(size_t)(i0 + i) * H + u] : 0.f;
    }
    for (int t = tid; t < NPL * MT * Bc::LDP3 / 2; t += NTW) reinterpret_cast<unsigned*>(Gp)[t] = 0u;
    __syncthreads();
    const int jj = lane & 15, kq = lane >> 4, ai = lane & 15;
    const int rot_id = (int)((blockIdx.x + gridDim.x * blockIdx.y) >> 3);      // see gru_fwd_bf_kernel
    // H = 400 (planes larger than the L2): a few distinct starting positions beat all 38 (sweep over RENET_GRU_ROT,
    // average of the forward and backward launch at config 5: 1 -> 910, 4 -> 754, 8 -> 756, 13 -> 802, 38 -> 818 us)
    const int rmod = ly.rot_mod > 0 ? min(ly.rot_mod, Bc::KG3) : (H > 200 ? 6 : Bc::KG3);
    const int rot_k = (rot_id % rmod) * (Bc::KG3 / rmod), rot_u = H <= 200 ? (rot_id / Bc::KG3) % C::NUB : 0;
    constexpr int PLG = MT * Bc::LDP3;
    // W_hh^T fragments through a register ring, PFB k groups ahead; continuous over the unit blocks and time steps of a
    // wave (see gru_fwd_bf_kernel): the first PFB k groups of the next unit block fly under phase 1 and the barriers
    // (buffer loads: descriptor and chunk offset in scalar registers, one address VGPR -- see gru_fwd_bf_kernel; with one
    // wave per unit block, 128 registers per wave, the ring is 4 k groups deep: the L2 -> L1 stream of a CU saturates
    // at ~45 B/clk whatever the depth, profiles/r06_f_gru.md)
    constexpr int PFB = Bc::KG3 > 5 ? (WV > 8 && NPL == 3 ? 3 : 5) : Bc::KG3 - 1, RB = PFB + 1, VPB = (Bc::KG3 + RB - 1) / RB * RB;
    bf16x8 wb[RB][NPL];
    auto rotk = [&](int q) { return q + rot_k >= Bc::KG3 ? q + rot_k - Bc::KG3 : q + rot_k; };
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16x8*>(WTp), (short)0, (int)((size_t)C::NUB * Bc::KG3 * NPL * 1024), 0x00020000);
    const int lane16 = lane * 16;
    auto frag_base = [&](int ub0) {                                     // byte offset of a unit block's fragments
        const int ub = ub0 + rot_u >= C::NUB ? ub0 + rot_u - C::NUB : ub0 + rot_u;
        return ub * (Bc::KG3 * NPL * 1024);
    };
    auto frag = [&](int base, int kgr, int p) {
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        const u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16 + p * 1024, base + kgr * (NPL * 1024), 0);
        return __builtin_bit_cast(bf16x8, v);
    };
    if (wave < C::NUB) {
        const int wf0 = frag_base(wave);
#pragma unroll
        for (int q = 0; q < PFB; ++q)
#pragma unroll
            for (int p = 0; p < NPL; ++p) wb[q][p] = frag(wf0, rotk(q), p);
    }

    for (int j = L - 1; j >= 0; --j) {
        const int p0 = so.off[j];
        const int bs = so.off[j + 1] - p0;
        if (i0 >= bs) continue;                                         // tile not alive yet at this step
        // phase 1: gate gradients of the live rows.  The saved activations are requested for ALL of this thread's
        // elements first, unconditionally (dead rows read the tile's last live row): inside the `row is alive` branch
        // every element paid its own memory latency (7 in a row per thread and step)
        constexpr int P1 = (MT * H + NTW - 1) / NTW;
        constexpr int PC = P1 > 7 ? 7 : P1;                             // elements requested together (35 registers)
#pragma unroll 1
        for (int q0 = 0; q0 < P1; q0 += PC) {
        float s_r[PC], s_z[PC], s_n[PC], s_hn[PC], s_hp[PC];
#pragma unroll
        for (int q = 0; q < PC; ++q) {
            const int t = min(tid + NTW * (q0 + q), MT * H - 1);
            const int i = t / H, u = t - i * H;
            const float* sv = saved + (size_t)(p0 + min(i0 + i, bs - 1)) * 5 * H + u;
            s_r[q] = sv[0]; s_z[q] = sv[H]; s_n[q] = sv[2 * H]; s_hn[q] = sv[3 * H]; s_hp[q] = sv[4 * H];
        }
#pragma unroll
        for (int q = 0; q < PC; ++q) {
            const int t = tid + NTW * (q0 + q);
            if (q0 + q < P1 && t < MT * H) {
                const int i = t / H, u = t - i * H;
                float gr = 0.f, gz = 0.f, gn = 0.f;
                if (i0 + i < bs) {
                    const size_t p = (size_t)(p0 + i0 + i);
                    const float r = s_r[q], z = s_z[q], n = s_n[q], hn = s_hn[q], hp = s_hp[q];
                    const float g = dHs[i * C::LDH + u];
                    const float dan = g * (1.f - z) * (1.f - n * n);
                    const float daz = g * (hp - n) * z * (1.f - z);
                    const float dar = dan * hn * r * (1.f - r);
                    gr = dar; gz = daz; gn = dan * r;
                    if constexpr (TRACK) gmax = fmaxf(gmax, fmaxf(fabsf(dar), fmaxf(fabsf(daz), fabsf(dan))));
                    if constexpr (OUT16) {
                        __bf16* gi = reinterpret_cast<__bf16*>(dGi) + p * ps.p[blockIdx.y].out_ld;
                        __bf16* gh = reinterpret_cast<__bf16*>(dGh) + p * ps.p[blockIdx.y].out_ld;
                        gi[u] = (__bf16)dar; gi[H + u] = (__bf16)daz; gi[2 * H + u] = (__bf16)dan;
                        gh[u] = (__bf16)gr; gh[H + u] = (__bf16)gz; gh[2 * H + u] = (__bf16)gn;
                    } else {
                        float* gi = dGi + p * C::K3;
                        float* gh = dGh + p * C::K3;
                        gi[u] = dar; gi[H + u] = daz; gi[2 * H + u] = dan;
                        gh[u] = gr; gh[H + u] = gz; gh[2 * H + u] = gn;
                    }
                    dHs[i * C::LDH + u] = g * z;                        // direct path h_prev -> h
                }
                if (j > 0) {
                    const Planes3 sr = split3(gr), sz = split3(gz), sn = split3(gn);
                    __bf16* row = Gp + i * Bc::LDP3;
#pragma unroll
                    for (int p = 0; p < NPL; ++p) {
                        row[p * PLG + u] = sr.p[p];
                        row[p * PLG + H + u] = sz.p[p];
                        row[p * PLG + 2 * H + u] = sn.p[p];
                    }
                }
            }
        }
        }
        __syncthreads();
        if (j > 0) {
            // phase 2: dh_prev += dGh W_hh  (rows of dead sequences have dGh = 0 and keep their dh)
#pragma unroll 1
            for (int ub0 = wave; ub0 < C::NUB; ub0 += WV) {
                {
                    const int ub = ub0 + rot_u >= C::NUB ? ub0 + rot_u - C::NUB : ub0 + rot_u;
                    const int u = ub * 16 + jj;
                    const bool uok = u < H;
                    f32x4 acc;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) acc[reg] = dHs[(4 * kq + reg) * C::LDH + (uok ? u : 0)];
                    const int wf = frag_base(ub0);
                    const int wfn = frag_base(ub0 + WV < C::NUB ? ub0 + WV : wave);
                    const __bf16* ga = Gp + ai * Bc::LDP3 + kq * 8;
                    f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};                  // two chains: no MFMA waits on the previous one
#pragma unroll 1
                    for (int base = 0; base < VPB; base += RB) {
#pragma unroll
                        for (int r = 0; r < RB; ++r) {
                            const int kg0 = base + r;
                            const int tq = kg0 + PFB;                   // virtual k group requested now
                            if (tq < Bc::KG3) {
                                const int kn = rotk(tq);
#pragma unroll
                                for (int p = 0; p < NPL; ++p) wb[(r + PFB) % RB][p] = frag(wf, kn, p);
                            } else if (tq >= VPB) {
                                const int kn = rotk(tq - VPB);
#pragma unroll
                                for (int p = 0; p < NPL; ++p) wb[(r + PFB) % RB][p] = frag(wfn, kn, p);
                            }
                            if (kg0 < Bc::KG3) {
                                const int kg = rotk(kg0);
                                __builtin_amdgcn_sched_barrier(0);
                                bf16x8 a[NPL];
#pragma unroll
                                for (int p = 0; p < NPL; ++p) a[p] = *reinterpret_cast<const bf16x8*>(ga + p * PLG + kg * 32);
                                if (r & 1) acc2 = mfma_p<NPL>(a, wb[r], acc2);
                                else acc = mfma_p<NPL>(a, wb[r], acc);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                    if (uok) {
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) dHs[(4 * kq + reg) * C::LDH + u] = acc[reg] + acc2[reg];
                    }
                }
            }
            __syncthreads();
        }
    }
    if (TRACK && bound) {                                                 // kernel-uniform per problem
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off));
        if (lane == 0) gred[wave] = gmax;
        __syncthreads();
        if (tid == 0) {
            float m = 0.f;
            for (int w = 0; w < WV; ++w) m = fmaxf(m, gred[w]);
            bound[blockIdx.x] = m;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Step kernels (the default bf16x6 path): ONE launch per time step, GEMM-shaped tiles.
//
// The persistent kernels above give a workgroup 16 sequences and ALL hidden units, so every workgroup re-streams
// the whole of W_hh (0.8 MB of planes at H = 200, 3 MB at H = 400) from L2 every step: 213 MB of L2 -> L1 traffic per
// step at H = 200 with 256 workgroups, and the 64 B/clk L1 fill rate of a CU -- not the matrix pipe -- bounds the step
// (PMC: r02 DESIGN 4).  Here a workgroup owns 64 sequences x 64 hidden units (4 waves, one block of 16 units each,
// 4 MFMA row tiles per wave): a W fragment is used for 4 row tiles, the A operand (bf16 planes of h, or of dGh in the
// backward pass) is staged through LDS in double-buffered chunks of 128 k and shared by the 4 waves, and the state
// travels between steps through L2 (fp32 h / dh in place, bf16 planes ping-pong) -- the kernel boundary is the
// grid-wide barrier the step needs.  W traffic per step drops 4x (16x per sequence tile), the step becomes
// matrix-pipe / epilogue-traffic bound, and later steps launch only the row tiles that are still alive.
// The epilogue is the gate math on the MFMA C layout exactly as above; backward launch j forms dh(j-1) and, in the
// same epilogue, the gate gradients of step j-1 (they need dh(j-1) at the lane's own (sequence, unit) pairs only).
// ---------------------------------------------------------------------------------------------
constexpr int SR = 64;                          // sequences per workgroup: 4 MFMA row tiles
constexpr int SRT = SR / 16;
constexpr int SW = 4;                           // waves per workgroup, one block of 16 hidden units each
constexpr int SNT = SW * 64;
constexpr int CKG = 4;                          // k groups (32 k each) per LDS chunk
constexpr int CK = CKG * 32;
constexpr int LDC = CK + 8;                     // bf16 row stride of a chunk in LDS (ds_read_b128 conflict free)
constexpr int CHUNK_ELEMS = 3 * SR * LDC;       // one chunk buffer: three planes
constexpr int STAGE_V = 3 * SR * (CK / 8) / SNT;        // 16-byte vectors per thread and chunk (12)
constexpr size_t STEP_LDS = (size_t)2 * CHUNK_ELEMS * sizeof(__bf16);

struct StageRegs { uint4 v[STAGE_V]; };

// A operand in global memory: [3][rows_pad][KPg] bf16 planes, rows_pad a multiple of SR, k padding zero
template <int KPg>
__device__ __forceinline__ void stage_load(const __bf16* __restrict__ A, size_t plane_stride, int r0, int kbase, int tid,
                                           StageRegs& s) {
#pragma unroll
    for (int q = 0; q < STAGE_V; ++q) {
        const int idx = tid + SNT * q;
        const int plane = idx / (SR * (CK / 8));
        const int rem = idx - plane * (SR * (CK / 8));
        const int row = rem / (CK / 8), seg = rem - row * (CK / 8);
        const int k = kbase + seg * 8;
        const uint4 v = *reinterpret_cast<const uint4*>(A + plane * plane_stride + (size_t)(r0 + row) * KPg + (k < KPg ? k : 0));
        s.v[q] = k < KPg ? v : make_uint4(0u, 0u, 0u, 0u);
    }
}

__device__ __forceinline__ void stage_store(__bf16* __restrict__ buf, int tid, const StageRegs& s) {
#pragma unroll
    for (int q = 0; q < STAGE_V; ++q) {
        const int idx = tid + SNT * q;
        const int plane = idx / (SR * (CK / 8));
        const int rem = idx - plane * (SR * (CK / 8));
        const int row = rem / (CK / 8), seg = rem - row * (CK / 8);
        *reinterpret_cast<uint4*>(buf + plane * (SR * LDC) + row * LDC + seg * 8) = s.v[q];
    }
}

// LDS-only barrier: the global loads that prefetch the next chunk / the next W fragments stay in flight across it
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

struct StepF {
    const float* Gi; const bf16x8* Wp; const float* bhh; float* h; float* saved;
    const __bf16* Ain; __bf16* Aout;
    int p0, bs;                         // packed row of sequence 0 at this step, sequences alive at this step
    size_t plane_stride;
};
struct StepsF { StepF p[MAXP]; };

template <int H, bool GEMM>
__global__ __launch_bounds__(SNT) void gru_step_fwd_kernel(StepsF ps) {
    using C = Cfg<H>;
    using Bc = BCfg<H>;
    const StepF& P = ps.p[blockIdx.z];
    const int r0 = blockIdx.x * SR;
    if (r0 >= P.bs) return;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* bufs = reinterpret_cast<__bf16*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ub = blockIdx.y * SW + wave;
    const bool wave_on = ub < C::NUB;                                   // wave-uniform
    const int jj = lane & 15, kq = lane >> 4, ai = lane & 15;
    const int u = ub * 16 + jj;
    const bool uok = wave_on && u < H;
    const int uc = uok ? u : 0;
    const int bs = P.bs;

    // operands of the epilogue, requested before the matrix work (unconditional loads from clamped rows)
    float gr[SRT][4], gz[SRT][4], gn[SRT][4], hp[SRT][4];
#pragma unroll
    for (int t = 0; t < SRT; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = min(r0 + 16 * t + 4 * kq + reg, bs - 1);
            const float* gi = P.Gi + (size_t)(P.p0 + row) * C::K3 + uc;
            gr[t][reg] = gi[0]; gz[t][reg] = gi[H]; gn[t][reg] = gi[2 * H];
            hp[t][reg] = GEMM ? P.h[(size_t)row * H + uc] : 0.f;
        }
    const float b_r = P.bhh[uc], b_z = P.bhh[H + uc], b_n = P.bhh[2 * H + uc];

    f32x4 ar[SRT], az[SRT], an[SRT];
#pragma unroll
    for (int t = 0; t < SRT; ++t) { ar[t] = {0.f, 0.f, 0.f, 0.f}; az[t] = ar[t]; an[t] = ar[t]; }

    if constexpr (GEMM) {
        constexpr int NC = (Bc::KG + CKG - 1) / CKG;
        const bf16x8* wf = P.Wp + (size_t)(wave_on ? ub : 0) * Bc::KG * 9 * 64 + lane;      // fragment order
        bf16x8 wcur[3][3], wnext[3][3];                                   // [gate][plane]
#pragma unroll
        for (int f = 0; f < 9; ++f) wcur[f / 3][f % 3] = wf[f * 64];
        StageRegs sr;
        stage_load<Bc::KP>(P.Ain, P.plane_stride, r0, 0, tid, sr);
        stage_store(bufs, tid, sr);
        if (NC > 1) stage_load<Bc::KP>(P.Ain, P.plane_stride, r0, CK, tid, sr);
        lds_barrier();
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const __bf16* cur = bufs + (c & 1) * CHUNK_ELEMS;
            if (c + 1 < NC) {
                stage_store(bufs + ((c + 1) & 1) * CHUNK_ELEMS, tid, sr);
                if (c + 2 < NC) stage_load<Bc::KP>(P.Ain, P.plane_stride, r0, (c + 2) * CK, tid, sr);
            }
            if (wave_on) {
                const __bf16* ha = cur + ai * LDC + kq * 8;
#pragma unroll
                for (int kk = 0; kk < CKG; ++kk) {
                    const int kg = c * CKG + kk;
                    if (kg < Bc::KG) {
                        if (kg + 1 < Bc::KG) {
#pragma unroll
                            for (int f = 0; f < 9; ++f) wnext[f / 3][f % 3] = wf[((kg + 1) * 9 + f) * 64];
                        }
#pragma unroll
                        for (int t = 0; t < SRT; ++t) {
                            bf16x8 a[3];
#pragma unroll
                            for (int p = 0; p < 3; ++p)
                                a[p] = *reinterpret_cast<const bf16x8*>(ha + p * (SR * LDC) + t * 16 * LDC + kk * 32);
                            ar[t] = mfma6(a, wcur[0], ar[t]);
                            az[t] = mfma6(a, wcur[1], az[t]);
                            an[t] = mfma6(a, wcur[2], an[t]);
                        }
                        if (kg + 1 < Bc::KG) {
#pragma unroll
                            for (int f = 0; f < 9; ++f) wcur[f / 3][f % 3] = wnext[f / 3][f % 3];
                        }
                    }
                }
            }
            lds_barrier();
        }
    }

    if (uok) {
        // C layout: column = lane & 15 (unit u), row = 4 * (lane >> 4) + reg (sequence of the row tile)
#pragma unroll
        for (int t = 0; t < SRT; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = r0 + 16 * t + 4 * kq + reg;
                if (row < bs) {
                    const size_t p = (size_t)(P.p0 + row);
                    const float hn = an[t][reg] + b_n;
                    const float r = sigmoidf_(gr[t][reg] + ar[t][reg] + b_r);
                    const float z = sigmoidf_(gz[t][reg] + az[t][reg] + b_z);
                    const float n = tanhf(gn[t][reg] + r * hn);
                    const float hpv = hp[t][reg];
                    const float hv = (1.f - z) * n + z * hpv;
                    float* sv = P.saved + p * 5 * H;
                    sv[u] = r; sv[H + u] = z; sv[2 * H + u] = n; sv[3 * H + u] = hn; sv[4 * H + u] = hpv;
                    P.h[(size_t)row * H + u] = hv;
                    const Planes3 sp = split3(hv);
                    __bf16* dst = P.Aout + (size_t)row * Bc::KP + u;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) dst[pl * P.plane_stride] = sp.p[pl];
                }
            }
    }
}

struct StepB {
    const float* saved; const bf16x8* WTp; float* dh; float* dGi; float* dGh;
    const __bf16* Ain; __bf16* Aout;
    int bs_cur;                         // sequences alive at the step whose dGh is contracted (0: none)
    int p0_prev, bs_prev;               // packed row 0 / sequences alive at the step whose gate gradients are formed
    size_t plane_stride;
};
struct StepsB { StepB p[MAXP]; };

// launch for step j:  dh(j-1) = dh(j) z(j) [already in dh] + dGh(j) W_hh  for the sequences alive at step j, then the
// gate gradients of step j-1 for the sequences alive at step j-1 (a superset: sequences whose last step is j-1 enter
// with dh = dh_last).  GEMM = false is the first launch (step L-1's gate gradients from dh_last alone).
template <int H, bool GEMM>
__global__ __launch_bounds__(SNT) void gru_step_bwd_kernel(StepsB ps) {
    using C = Cfg<H>;
    using Bc = BCfg<H>;
    const StepB& P = ps.p[blockIdx.z];
    const int r0 = blockIdx.x * SR;
    if (r0 >= P.bs_prev) return;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* bufs = reinterpret_cast<__bf16*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ub = blockIdx.y * SW + wave;
    const bool wave_on = ub < C::NUB;
    const int jj = lane & 15, kq = lane >> 4, ai = lane & 15;
    const int u = ub * 16 + jj;
    const bool uok = wave_on && u < H;
    const int uc = uok ? u : 0;
    const int bsp = P.bs_prev;

    float sv_r[SRT][4], sv_z[SRT][4], sv_n[SRT][4], sv_hn[SRT][4], sv_hp[SRT][4], gin[SRT][4];
#pragma unroll
    for (int t = 0; t < SRT; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = min(r0 + 16 * t + 4 * kq + reg, bsp - 1);
            const float* sv = P.saved + (size_t)(P.p0_prev + row) * 5 * H + uc;
            sv_r[t][reg] = sv[0]; sv_z[t][reg] = sv[H]; sv_n[t][reg] = sv[2 * H]; sv_hn[t][reg] = sv[3 * H];
            sv_hp[t][reg] = sv[4 * H];
            gin[t][reg] = P.dh[(size_t)row * H + uc];
        }

    f32x4 acc[SRT];
#pragma unroll
    for (int t = 0; t < SRT; ++t) acc[t] = {0.f, 0.f, 0.f, 0.f};

    if (GEMM && r0 < P.bs_cur) {                                        // workgroup-uniform
        constexpr int NC = (Bc::KG3 + CKG - 1) / CKG;
        const bf16x8* wf = P.WTp + (size_t)(wave_on ? ub : 0) * Bc::KG3 * 3 * 64 + lane;
        bf16x8 wcur[3], wnext[3];
#pragma unroll
        for (int f = 0; f < 3; ++f) wcur[f] = wf[f * 64];
        StageRegs sr;
        stage_load<Bc::KP3>(P.Ain, P.plane_stride, r0, 0, tid, sr);
        stage_store(bufs, tid, sr);
        if (NC > 1) stage_load<Bc::KP3>(P.Ain, P.plane_stride, r0, CK, tid, sr);
        lds_barrier();
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const __bf16* cur = bufs + (c & 1) * CHUNK_ELEMS;
            if (c + 1 < NC) {
                stage_store(bufs + ((c + 1) & 1) * CHUNK_ELEMS, tid, sr);
                if (c + 2 < NC) stage_load<Bc::KP3>(P.Ain, P.plane_stride, r0, (c + 2) * CK, tid, sr);
            }
            if (wave_on) {
                const __bf16* ga = cur + ai * LDC + kq * 8;
#pragma unroll
                for (int kk = 0; kk < CKG; ++kk) {
                    const int kg = c * CKG + kk;
                    if (kg < Bc::KG3) {
                        if (kg + 1 < Bc::KG3) {
#pragma unroll
                            for (int f = 0; f < 3; ++f) wnext[f] = wf[((kg + 1) * 3 + f) * 64];
                        }
#pragma unroll
                        for (int t = 0; t < SRT; ++t) {
                            bf16x8 a[3];
#pragma unroll
                            for (int p = 0; p < 3; ++p)
                                a[p] = *reinterpret_cast<const bf16x8*>(ga + p * (SR * LDC) + t * 16 * LDC + kk * 32);
                            acc[t] = mfma6(a, wcur, acc[t]);
                        }
                        if (kg + 1 < Bc::KG3) {
#pragma unroll
                            for (int f = 0; f < 3; ++f) wcur[f] = wnext[f];
                        }
                    }
                }
            }
            lds_barrier();
        }
    }

    if (uok) {
#pragma unroll
        for (int t = 0; t < SRT; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = r0 + 16 * t + 4 * kq + reg;
                if (row < bsp) {
                    const size_t p = (size_t)(P.p0_prev + row);
                    const float r = sv_r[t][reg], z = sv_z[t][reg], n = sv_n[t][reg], hn = sv_hn[t][reg];
                    const float hpv = sv_hp[t][reg];
                    const float g = gin[t][reg] + ((GEMM && row < P.bs_cur) ? acc[t][reg] : 0.f);
                    const float dan = g * (1.f - z) * (1.f - n * n);
                    const float daz = g * (hpv - n) * z * (1.f - z);
                    const float dar = dan * hn * r * (1.f - r);
                    float* gi = P.dGi + p * C::K3;
                    float* gh = P.dGh + p * C::K3;
                    gi[u] = dar; gi[H + u] = daz; gi[2 * H + u] = dan;
                    gh[u] = dar; gh[H + u] = daz; gh[2 * H + u] = dan * r;
                    P.dh[(size_t)row * H + u] = g * z;                  // direct path h_prev -> h
                    const Planes3 s0 = split3(dar), s1 = split3(daz), s2 = split3(dan * r);
                    __bf16* dst = P.Aout + (size_t)row * Bc::KP3 + u;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        dst[pl * P.plane_stride] = s0.p[pl];
                        dst[pl * P.plane_stride + H] = s1.p[pl];
                        dst[pl * P.plane_stride + 2 * H] = s2.p[pl];
                    }
                }
            }
    }
}

// W_hh [3H, H] -> W_hh^T [H, 3H]  (tiny; once per backward call)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int rows, int cols,
                                                        float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    for (int r = ty; r < 32; r += 8)
        if (by + r < rows && bx + tx < cols) tile[r][tx] = in[(size_t)(by + r) * cols + bx + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (bx + r < cols && by + tx < rows) out[(size_t)(bx + r) * rows + by + tx] = tile[tx][r];
}

bool use_f32() {                        // RENET_GEMM=f32: exact-fp32 products everywhere (gemm.hip too)
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("RENET_GEMM");
        v = (e && strcmp(e, "f32") == 0) ? 1 : 0;
    }
    return v == 1;
}

template <class KernelT>
int set_lds(KernelT kernel, size_t lds, bool& done) {
    if (!done && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    done = true;                       // benign race: the attribute is idempotent
    return RENET_OK;
}

inline int max_rows(const Layouts& ly) { return ly.rows[0] > ly.rows[1] ? ly.rows[0] : ly.rows[1]; }

template <int H>
int launch_fwd(const FwdProbs& ps, int np, const Layouts& ly, hipStream_t st) {
    RENET_LAUNCH((gru_fwd_kernel<H>), dim3((max_rows(ly) + MT - 1) / MT, np), dim3(NT), 0, st, ps, ly);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

// waves per workgroup of the one-plane (bf16 mode) recurrences: 16 (128 VGPRs) up to H = 200; 12 (170 VGPRs) at H = 400,
// where 16 waves spill 17-39 registers
template <int H> constexpr int nw1() { return H > 200 ? 12 : 16; }

// waves per workgroup of the bf16x6 forward recurrence: ONE wave per unit block where the blocks fit a workgroup (H = 200:
// 13 waves) -- with 8 waves five of them own two blocks and the step waits for those (profiles/r06_f_gru.md)
template <int H> constexpr int nw3() { return (Cfg<H>::NUB > 8 && Cfg<H>::NUB <= 16) ? Cfg<H>::NUB : NW; }

template <int H>
int launch_fwd_x(const FwdProbsB& ps, int np, const Layouts& ly, hipStream_t st) {
    const size_t lds = XCfg<H>::lds_bytes();
    static bool attr_set = false;
    const int e = set_lds(gru_fwd_x_kernel<H>, lds, attr_set);
    if (e != RENET_OK) return e;
    RENET_LAUNCH((gru_fwd_x_kernel<H>), dim3((max_rows(ly) + MT - 1) / MT, np), dim3(XCfg<H>::NTW), lds, st, ps, ly);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

// RENET_GRU_FWD=ring selects the per-unit-block kernel (gru_fwd_bf_kernel) at H <= 200 as well, for A/B runs
inline bool fwd_stream_kernel() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("RENET_GRU_FWD");
        v = (e && strcmp(e, "ring") == 0) ? 0 : 1;
    }
    return v == 1;
}

template <int H, int NPL = 3>
int launch_fwd_bf(const FwdProbsB& ps, int np, const Layouts& ly, hipStream_t st) {
    using C = Cfg<H>;
    if constexpr (NPL == 3 && H <= 200) {
        if (fwd_stream_kernel()) return launch_fwd_x<H>(ps, np, ly, st);
    }
    constexpr int WV = NPL == 1 ? nw1<H>() : nw3<H>();
    const size_t lds = (size_t)2 * MT * C::LDH * sizeof(float) + (size_t)3 * MT * BCfg<H>::LDP * sizeof(__bf16);
    static bool attr_set = false;
    const int e = set_lds(gru_fwd_bf_kernel<H, NPL, WV>, lds, attr_set);
    if (e != RENET_OK) return e;
    RENET_LAUNCH((gru_fwd_bf_kernel<H, NPL, WV>), dim3((max_rows(ly) + MT - 1) / MT, np), dim3(WV * 64), lds, st, ps, ly);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

template <int H>
int launch_bwd(const BwdProbs& ps, int np, const Layouts& ly, hipStream_t st) {
    using C = Cfg<H>;
    const size_t lds = (size_t)MT * (C::LDH + C::LDG) * sizeof(float);
    static bool attr_set = false;
    const int e = set_lds(gru_bwd_kernel<H>, lds, attr_set);
    if (e != RENET_OK) return e;
    RENET_LAUNCH((gru_bwd_kernel<H>), dim3((max_rows(ly) + MT - 1) / MT, np), dim3(NT), lds, st, ps, ly);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

template <int H, int NPL = 3, bool OUT16 = false>
int launch_bwd_bf(const BwdProbsB& ps, int np, const Layouts& ly, hipStream_t st) {
    using C = Cfg<H>;
    const size_t lds = (size_t)MT * C::LDH * sizeof(float) + (size_t)3 * MT * BCfg<H>::LDP3 * sizeof(__bf16);
    static bool attr_set = false;
    constexpr int WV = NPL == 1 ? nw1<H>() : nw3<H>();      // (H = 200: 13 waves, 118 -> 99 us per launch; profiles/r06_f_gru.md)
    const int e = set_lds(gru_bwd_bf_kernel<H, NPL, OUT16, WV>, lds, attr_set);
    if (e != RENET_OK) return e;
    RENET_LAUNCH((gru_bwd_bf_kernel<H, NPL, OUT16, WV>), dim3((max_rows(ly) + MT - 1) / MT, np), dim3(WV * 64), lds, st, ps, ly);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

// ---- step-kernel launch sequences ---------------------------------------------------------------
inline int rows_pad_of(int rows) { return (rows + SR - 1) / SR * SR; }
inline size_t kp_of(int K) { return (size_t)((K + 31) / 32) * 32; }

// Which bf16x6 recurrence runs.  Measured on MI355X (merged step, 2 x 2048 sequences; profiles/r02_gru_steps.md):
//   H = 200: persistent 189 / 167 us (fwd / bwd) per launch;  step kernels 10 x 20.8 / 10 x 28.3 us
//   H = 400, L = 15: persistent 858 us average;                step kernels 814 us
// The step kernels cut the W_hh stream 4x, but a launch whose workgroups all load, multiply and store in lockstep
// leaves the memory system idle during the MFMA phase and the matrix pipe idle during the epilogue (the epilogue
// alone -- Gi in, saved / h / planes out, 46 B per element and step -- is 12.6 us of a 20.8 us forward step),
// while the persistent workgroups drift apart and overlap the two.  With the W_hh stream of the persistent workgroups
// de-synchronised (rot_k / rot_u in the kernels: 188 -> 142 us at H = 200, 858 -> ~790 us at H = 400) the persistent
// kernels are the default everywhere; RENET_GRU=steps selects the per-step launches.
bool use_persistent(int H) {
    const char* e = getenv("RENET_GRU");
    if (e && strcmp(e, "persistent") == 0) return true;
    if (e && strcmp(e, "steps") == 0) return false;
    (void)H;
    return true;
}

struct StepState {                      // per problem: bf16 plane ping-pong of the A operand + fp32 dh
    __bf16* A[2];
    float* dh;
    size_t plane_stride;                // elements between two planes of one buffer
};

template <int H>
int run_steps_fwd(int n, const Layouts& ly, const StepF* base, const StepState* stt, hipStream_t st) {
    using C = Cfg<H>;
    static bool a0 = false, a1 = false;
    int e = set_lds(gru_step_fwd_kernel<H, false>, STEP_LDS, a0);
    if (e != RENET_OK) return e;
    e = set_lds(gru_step_fwd_kernel<H, true>, STEP_LDS, a1);
    if (e != RENET_OK) return e;
    int maxL = 0;
    for (int k = 0; k < n; ++k) maxL = ly.L[ly.lay_of[k]] > maxL ? ly.L[ly.lay_of[k]] : maxL;
    for (int j = 0; j < maxL; ++j) {
        StepsF ps;
        int maxbs = 0;
        for (int i = 0; i < MAXP; ++i) {
            const int k = i < n ? i : 0;
            const int lay = ly.lay_of[k];
            ps.p[i] = base[k];
            ps.p[i].p0 = 0; ps.p[i].bs = 0;
            if (i < n && j < ly.L[lay]) {
                ps.p[i].p0 = ly.so[lay].off[j];
                ps.p[i].bs = ly.so[lay].off[j + 1] - ly.so[lay].off[j];
            }
            ps.p[i].Aout = stt[k].A[j & 1];
            ps.p[i].Ain = stt[k].A[(j + 1) & 1];
            ps.p[i].plane_stride = stt[k].plane_stride;
            maxbs = ps.p[i].bs > maxbs ? ps.p[i].bs : maxbs;
        }
        if (maxbs == 0) continue;
        const dim3 grid((maxbs + SR - 1) / SR, (C::NUB + SW - 1) / SW, n);
        if (j == 0) RENET_LAUNCH((gru_step_fwd_kernel<H, false>), grid, dim3(SNT), STEP_LDS, st, ps);
        else RENET_LAUNCH((gru_step_fwd_kernel<H, true>), grid, dim3(SNT), STEP_LDS, st, ps);
        RENET_LAUNCH_CHECK();
    }
    return RENET_OK;
}

template <int H>
int run_steps_bwd(int n, const Layouts& ly, const StepB* base, const StepState* stt, hipStream_t st) {
    using C = Cfg<H>;
    static bool a0 = false, a1 = false;
    int e = set_lds(gru_step_bwd_kernel<H, false>, STEP_LDS, a0);
    if (e != RENET_OK) return e;
    e = set_lds(gru_step_bwd_kernel<H, true>, STEP_LDS, a1);
    if (e != RENET_OK) return e;
    int maxL = 0;
    for (int k = 0; k < n; ++k) maxL = ly.L[ly.lay_of[k]] > maxL ? ly.L[ly.lay_of[k]] : maxL;
    for (int s = 0; s < maxL; ++s) {                 // launch s handles step L-1-s of every problem that has one
        StepsB ps;
        int maxbs = 0;
        for (int i = 0; i < MAXP; ++i) {
            const int k = i < n ? i : 0;
            const int lay = ly.lay_of[k];
            const int Lk = ly.L[lay];
            ps.p[i] = base[k];
            ps.p[i].bs_cur = 0; ps.p[i].p0_prev = 0; ps.p[i].bs_prev = 0;
            if (i < n && s < Lk) {
                const int jp = Lk - 1 - s;
                ps.p[i].p0_prev = ly.so[lay].off[jp];
                ps.p[i].bs_prev = ly.so[lay].off[jp + 1] - ly.so[lay].off[jp];
                if (s > 0) ps.p[i].bs_cur = ly.so[lay].off[jp + 2] - ly.so[lay].off[jp + 1];
            }
            ps.p[i].Aout = stt[k].A[s & 1];
            ps.p[i].Ain = stt[k].A[(s + 1) & 1];
            ps.p[i].dh = stt[k].dh;
            ps.p[i].plane_stride = stt[k].plane_stride;
            maxbs = ps.p[i].bs_prev > maxbs ? ps.p[i].bs_prev : maxbs;
        }
        if (maxbs == 0) continue;
        const dim3 grid((maxbs + SR - 1) / SR, (C::NUB + SW - 1) / SW, n);
        if (s == 0) RENET_LAUNCH((gru_step_bwd_kernel<H, false>), grid, dim3(SNT), STEP_LDS, st, ps);
        else RENET_LAUNCH((gru_step_bwd_kernel<H, true>), grid, dim3(SNT), STEP_LDS, st, ps);
        RENET_LAUNCH_CHECK();
    }
    return RENET_OK;
}

bool fill_offsets(const int32_t* step_off, int L, StepOff& so, int& B) {
    if (L < 1 || L > MAXL || !step_off) return false;
    for (int j = 0; j <= L; ++j) so.off[j] = step_off[j];
    for (int j = L + 1; j <= MAXL; ++j) so.off[j] = step_off[L];
    B = step_off[1] - step_off[0];
    for (int j = 1; j < L; ++j)                                   // batch sizes must be non-increasing
        if (step_off[j + 1] - step_off[j] > step_off[j] - step_off[j - 1]) return false;
    return B >= 0;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int nub(int H) { return (H + 15) / 16; }
inline size_t fwd_plane_bytes(int H) { return (size_t)nub(H) * ((H + 31) / 32) * 9 * 1024; }          // 1 KB chunks
inline size_t bwd_t_bytes(int H) { return align256((size_t)3 * H * H * sizeof(float)); }
inline size_t bwd_plane_bytes(int H) { return (size_t)nub(H) * ((3 * H + 31) / 32) * 3 * 1024; }

int split_frag(const float* in, int U, int K, int G, size_t sg, size_t su, size_t sk, bf16x8* out, hipStream_t st,
               int npl = 3) {
    const int NUBk = (U + 15) / 16, KGk = (K + 31) / 32;
    const int total = NUBk * KGk * G * 64;
    RENET_LAUNCH(split_frag_kernel, dim3((total + 255) / 256), dim3(256), 0, st, in, U, K, G, sg, su, sk, NUBk,
                       KGk, npl, out);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

}  // namespace

extern "C" {

#ifdef RENET_GRU_TRACE
int renet_gru_trace_set(unsigned long long* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_gru_trace), &buf, sizeof(buf));
}
#endif

// per GRU: forward = the bf16 planes of W_hh; backward = those of W_hh^T (bf16x6) or W_hh^T in fp32 (RENET_GEMM=f32)
size_t renet_gru_workspace(int B, int H) {
    size_t m = fwd_plane_bytes(H);
    if (bwd_t_bytes(H) > m) m = bwd_t_bytes(H);
    if (bwd_plane_bytes(H) > m) m = bwd_plane_bytes(H);
    m = align256(m);
    if (B > 0) {                        // step kernels: A-operand planes (ping-pong, sized for K = 3H) + dh
        const size_t rp = (size_t)rows_pad_of(B);
        m += align256((size_t)2 * 3 * rp * kp_of(3 * H) * sizeof(__bf16)) + align256(rp * H * sizeof(float));
    }
    return m;
}

}  // extern "C"

namespace {

// Groups the n problems by packed layout (equal step_off POINTERS = same layout) and validates each layout.
// rows_in[k]: forward = rows of h_last of problem k, backward = ignored (B of the layout is used).
int make_layouts(int n, const int32_t* const* step_off, const int* Ls, const int* rows_in, bool fwd, Layouts& ly,
                 int* B_of) {
    const int32_t* seen[MAXLAY] = {nullptr, nullptr};
    int nl = 0;
    for (int i = 0; i < MAXLAY; ++i) {
        ly.L[i] = 0; ly.rows[i] = 0;
        for (int j = 0; j <= MAXL; ++j) ly.so[i].off[j] = 0;
    }
    for (int k = 0; k < MAXP; ++k) ly.lay_of[k] = 0;
    {
        const char* e = getenv("RENET_GRU_ROT");
        ly.rot_mod = e ? atoi(e) : 0;
    }
    for (int k = 0; k < n; ++k) {
        int l = -1;
        for (int i = 0; i < nl; ++i)
            if (seen[i] == step_off[k] && ly.L[i] == Ls[k]) l = i;
        if (l < 0) {
            if (nl == MAXLAY) return RENET_ERR_BADARG;
            l = nl++;
            seen[l] = step_off[k];
            ly.L[l] = Ls[k];
            int B = 0;
            if (Ls[k] > 0 && !fill_offsets(step_off[k], Ls[k], ly.so[l], B)) return RENET_ERR_BADARG;
            ly.rows[l] = fwd ? 0 : B;
        }
        const int B = ly.L[l] > 0 ? ly.so[l].off[1] - ly.so[l].off[0] : 0;
        if (fwd) {
            if (rows_in[k] < B) return RENET_ERR_BADARG;
            if (ly.rows[l] != 0 && ly.rows[l] != rows_in[k]) return RENET_ERR_BADARG;   // one h_last height per layout
            ly.rows[l] = rows_in[k];
        }
        ly.lay_of[k] = l;
        B_of[k] = B;
    }
    return RENET_OK;
}

// state region of problem k inside its workspace slice (behind the weight planes)
StepState carve_state(char* slice, int H, int Bmax, size_t kp) {
    StepState t;
    const size_t rp = (size_t)rows_pad_of(Bmax);
    char* base = slice + align256(renet_gru_workspace(0, H));
    t.plane_stride = rp * kp;
    t.A[0] = reinterpret_cast<__bf16*>(base);
    t.A[1] = t.A[0] + 3 * t.plane_stride;
    t.dh = reinterpret_cast<float*>(base + align256((size_t)2 * 3 * rp * kp_of(3 * H) * sizeof(__bf16)));
    return t;
}

// W_hh planes are split once per DISTINCT weight pointer (the subject and object passes share the encoders)
int plane_slot(int k, const float* const* W) {
    for (int i = 0; i < k; ++i)
        if (W[i] == W[k]) return i;
    return k;
}

}  // namespace

extern "C" {

static int gru_fwd_impl(int npl, int n, const float* const* Gi, const int32_t* const* step_off, const int* L, int H,
                        const float* const* Whh, const float* const* bhh, float* const* h_last,
                        const int* out_rows, float* const* saved, float* workspace, size_t workspace_bytes,
                        void* stream);

int renet_gru_fwd_layouts(int n, const float* const* Gi, const int32_t* const* step_off, const int* L, int H,
                          const float* const* Whh, const float* const* bhh, float* const* h_last,
                          const int* out_rows, float* const* saved, float* workspace, size_t workspace_bytes,
                          void* stream) {
    return gru_fwd_impl(3, n, Gi, step_off, L, H, Whh, bhh, h_last, out_rows, saved, workspace, workspace_bytes, stream);
}

int renet_gru_fwd_layouts_f32(int n, const float* const* Gi, const int32_t* const* step_off, const int* L, int H,
                              const float* const* Whh, const float* const* bhh, float* const* h_last,
                              const int* out_rows, float* const* saved, float* workspace, size_t workspace_bytes,
                              void* stream) {
    return gru_fwd_impl(0, n, Gi, step_off, L, H, Whh, bhh, h_last, out_rows, saved, workspace, workspace_bytes, stream);
}

int renet_gru_fwd_layouts_bf16(int n, const float* const* Gi, const int32_t* const* step_off, const int* L, int H,
                               const float* const* Whh, const float* const* bhh, float* const* h_last,
                               const int* out_rows, float* const* saved, float* workspace, size_t workspace_bytes,
                               void* stream) {
    return gru_fwd_impl(1, n, Gi, step_off, L, H, Whh, bhh, h_last, out_rows, saved, workspace, workspace_bytes, stream);
}

static int gru_fwd_impl(int npl, int n, const float* const* Gi, const int32_t* const* step_off, const int* L, int H,
                        const float* const* Whh, const float* const* bhh, float* const* h_last,
                        const int* out_rows, float* const* saved, float* workspace, size_t workspace_bytes,
                        void* stream) {
    if (n < 1 || n > MAXP) return RENET_ERR_BADARG;
    Layouts ly;
    int B_of[MAXP];
    const int e0 = make_layouts(n, step_off, L, out_rows, true, ly, B_of);
    if (e0 != RENET_OK) return e0;
    if (max_rows(ly) == 0) return RENET_OK;
    if (H != 100 && H != 200 && H != 400) return RENET_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (npl == 0 || (use_f32() && npl == 3)) {              // npl 0: the exact-fp32 kernels asked for by the caller (per model)
        FwdProbs ps;
        for (int i = 0; i < MAXP; ++i) {
            const int k = i < n ? i : 0;
            ps.p[i].Gi = Gi[k]; ps.p[i].Whh = Whh[k]; ps.p[i].bhh = bhh[k]; ps.p[i].h_last = h_last[k];
            ps.p[i].saved = saved[k];
        }
        switch (H) {
            case 100: return launch_fwd<100>(ps, n, ly, st);
            case 200: return launch_fwd<200>(ps, n, ly, st);
            default: return launch_fwd<400>(ps, n, ly, st);
        }
    }
    const bool steps = npl == 3 && !use_persistent(H);
    int Bmax = 0;
    for (int k = 0; k < n; ++k) Bmax = B_of[k] > Bmax ? B_of[k] : Bmax;
    const size_t per = renet_gru_workspace(steps ? Bmax : 0, H);
    if (!workspace || workspace_bytes < (size_t)n * per) return RENET_ERR_WORKSPACE;
    FwdProbsB ps;
    StepF sf[MAXP];
    StepState stt[MAXP];
    for (int i = 0; i < MAXP; ++i) {
        const int k = i < n ? i : 0;
        const int slot = plane_slot(k, Whh);
        bf16x8* planes = reinterpret_cast<bf16x8*>(reinterpret_cast<char*>(workspace) + (size_t)slot * per);
        if (i < n && slot == k) {                   // gate g of unit u, input k: W_hh[g*H + u][k]
            const int e = split_frag(Whh[k], H, H, 3, (size_t)H * H, (size_t)H, 1, planes, st, npl);
            if (e != RENET_OK) return e;
        }
        ps.p[i].Gi = Gi[k]; ps.p[i].Wp = planes; ps.p[i].bhh = bhh[k]; ps.p[i].h_last = h_last[k];
        ps.p[i].saved = saved[k];
        if (steps && i < n) {
            const size_t kp = kp_of(H);
            stt[i] = carve_state(reinterpret_cast<char*>(workspace) + (size_t)i * per, H, Bmax, kp);
            sf[i].Gi = Gi[i]; sf[i].Wp = planes; sf[i].bhh = bhh[i]; sf[i].h = h_last[i]; sf[i].saved = saved[i];
            sf[i].Ain = sf[i].Aout = nullptr; sf[i].p0 = sf[i].bs = 0; sf[i].plane_stride = stt[i].plane_stride;
            // h0 = 0 (also the rows of the empty histories past B); k padding of the A planes = 0
            const int lay = ly.lay_of[i];
            hipError_t he = hipMemsetAsync(h_last[i], 0, (size_t)ly.rows[lay] * H * sizeof(float), st);
            if (he != hipSuccess) return (int)he;
            if (kp > (size_t)H) {
                he = hipMemset2DAsync(stt[i].A[0] + H, kp * sizeof(__bf16), 0, (kp - H) * sizeof(__bf16),
                                      (size_t)2 * 3 * rows_pad_of(Bmax), st);
                if (he != hipSuccess) return (int)he;
            }
        }
    }
    if (steps) {
        switch (H) {
            case 100: return run_steps_fwd<100>(n, ly, sf, stt, st);
            case 200: return run_steps_fwd<200>(n, ly, sf, stt, st);
            default: return run_steps_fwd<400>(n, ly, sf, stt, st);
        }
    }
    if (npl == 1) {
        switch (H) {
            case 100: return launch_fwd_bf<100, 1>(ps, n, ly, st);
            case 200: return launch_fwd_bf<200, 1>(ps, n, ly, st);
            default: return launch_fwd_bf<400, 1>(ps, n, ly, st);
        }
    }
    switch (H) {
        case 100: return launch_fwd_bf<100>(ps, n, ly, st);
        case 200: return launch_fwd_bf<200>(ps, n, ly, st);
        default: return launch_fwd_bf<400>(ps, n, ly, st);
    }
}

static int gru_bwd_impl(int npl, int n, const float* const* dh_last, const int32_t* const* step_off, const int* L, int H,
                        const float* const* Whh, const float* const* saved, float* const* dGi,
                        float* const* dGh, float* workspace, size_t workspace_bytes, void* stream, int out_ld = 0,
                        float* const* bounds = nullptr);

int renet_gru_bwd_layouts(int n, const float* const* dh_last, const int32_t* const* step_off, const int* L, int H,
                          const float* const* Whh, const float* const* saved, float* const* dGi,
                          float* const* dGh, float* workspace, size_t workspace_bytes, void* stream) {
    return gru_bwd_impl(3, n, dh_last, step_off, L, H, Whh, saved, dGi, dGh, workspace, workspace_bytes, stream);
}

int renet_gru_bwd_layouts_f32(int n, const float* const* dh_last, const int32_t* const* step_off, const int* L, int H,
                              const float* const* Whh, const float* const* saved, float* const* dGi,
                              float* const* dGh, float* workspace, size_t workspace_bytes, void* stream) {
    return gru_bwd_impl(0, n, dh_last, step_off, L, H, Whh, saved, dGi, dGh, workspace, workspace_bytes, stream);
}

int renet_gru_bound_parts(int max_rows) { return max(1, (max_rows + MT - 1) / MT); }

int renet_gru_bwd_layouts_bounds(int n, const float* const* dh_last, const int32_t* const* step_off, const int* L, int H,
                                 const float* const* Whh, const float* const* saved, float* const* dGi,
                                 float* const* dGh, float* const* bounds, float* workspace, size_t workspace_bytes,
                                 void* stream) {
    if (!bounds) return RENET_ERR_BADARG;
    return gru_bwd_impl(3, n, dh_last, step_off, L, H, Whh, saved, dGi, dGh, workspace, workspace_bytes, stream, 0,
                        bounds);
}

int renet_gru_bwd_layouts_bf16(int n, const float* const* dh_last, const int32_t* const* step_off, const int* L, int H,
                               const float* const* Whh, const float* const* saved, float* const* dGi,
                               float* const* dGh, float* workspace, size_t workspace_bytes, void* stream) {
    return gru_bwd_impl(1, n, dh_last, step_off, L, H, Whh, saved, dGi, dGh, workspace, workspace_bytes, stream);
}

int renet_gru_bwd_layouts_bf16out(int n, const float* const* dh_last, const int32_t* const* step_off, const int* L,
                                  int H, const float* const* Whh, const float* const* saved, void* const* dGi16,
                                  void* const* dGh16, int out_ld, float* workspace, size_t workspace_bytes,
                                  void* stream) {
    if (out_ld < 3 * H) return RENET_ERR_BADARG;
    return gru_bwd_impl(1, n, dh_last, step_off, L, H, Whh, saved, reinterpret_cast<float* const*>(dGi16),
                        reinterpret_cast<float* const*>(dGh16), workspace, workspace_bytes, stream, out_ld);
}

static int gru_bwd_impl(int npl, int n, const float* const* dh_last, const int32_t* const* step_off, const int* L, int H,
                        const float* const* Whh, const float* const* saved, float* const* dGi,
                        float* const* dGh, float* workspace, size_t workspace_bytes, void* stream, int out_ld,
                        float* const* bounds) {
    if (n < 1 || n > MAXP) return RENET_ERR_BADARG;
    Layouts ly;
    int B_of[MAXP];
    const int e0 = make_layouts(n, step_off, L, nullptr, false, ly, B_of);
    if (e0 != RENET_OK) return e0;
    if (max_rows(ly) == 0) return RENET_OK;
    if (H != 100 && H != 200 && H != 400) return RENET_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const bool f32 = npl == 0 || (use_f32() && npl == 3);
    const bool steps = npl == 3 && !f32 && !use_persistent(H);
    if (bounds && (f32 || steps || npl != 3)) return RENET_ERR_UNSUPPORTED;   // only the persistent bf16x6 kernel emits them
    int Bmax = 0;
    for (int k = 0; k < n; ++k) Bmax = B_of[k] > Bmax ? B_of[k] : Bmax;
    const size_t per = renet_gru_workspace(steps ? Bmax : 0, H);
    if (!workspace || workspace_bytes < (size_t)n * per) return RENET_ERR_WORKSPACE;
    BwdProbs ps;
    BwdProbsB pb;
    StepB sb[MAXP];
    StepState stt[MAXP];
    for (int i = 0; i < MAXP; ++i) {
        const int k = i < n ? i : 0;
        const int slot = plane_slot(k, Whh);
        char* base = reinterpret_cast<char*>(workspace) + (size_t)slot * per;
        float* WhhT = reinterpret_cast<float*>(base);
        bf16x8* planes = reinterpret_cast<bf16x8*>(base);
        if (i < n && slot == k) {
            if (f32) {
                RENET_LAUNCH(transpose_kernel, dim3((H + 31) / 32, (3 * H + 31) / 32), dim3(256), 0, st, Whh[k],
                                   3 * H, H, WhhT);
                RENET_LAUNCH_CHECK();
            } else {                                // W_hh^T: unit u = hidden unit, k = gate column c: W_hh[c][u]
                const int e = split_frag(Whh[k], H, 3 * H, 1, 0, 1, (size_t)H, planes, st, npl);
                if (e != RENET_OK) return e;
            }
        }
        ps.p[i].dh_last = dh_last[k]; ps.p[i].WhhT = WhhT; ps.p[i].saved = saved[k]; ps.p[i].dGi = dGi[k];
        ps.p[i].dGh = dGh[k];
        pb.p[i].dh_last = dh_last[k]; pb.p[i].WTp = planes; pb.p[i].saved = saved[k]; pb.p[i].dGi = dGi[k];
        pb.p[i].dGh = dGh[k]; pb.p[i].out_ld = out_ld; pb.p[i].bound = bounds ? bounds[k] : nullptr;
        if (steps && i < n) {
            const size_t kp = kp_of(3 * H);
            stt[i] = carve_state(reinterpret_cast<char*>(workspace) + (size_t)i * per, H, Bmax, kp);
            sb[i].saved = saved[i]; sb[i].WTp = planes; sb[i].dh = stt[i].dh; sb[i].dGi = dGi[i]; sb[i].dGh = dGh[i];
            sb[i].Ain = sb[i].Aout = nullptr; sb[i].bs_cur = sb[i].p0_prev = sb[i].bs_prev = 0;
            sb[i].plane_stride = stt[i].plane_stride;
            if (B_of[i] > 0) {
                hipError_t he = hipMemcpyAsync(stt[i].dh, dh_last[i], (size_t)B_of[i] * H * sizeof(float),
                                               hipMemcpyDeviceToDevice, st);
                if (he != hipSuccess) return (int)he;
                if (kp > (size_t)3 * H) {
                    he = hipMemset2DAsync(stt[i].A[0] + 3 * H, kp * sizeof(__bf16), 0, (kp - 3 * H) * sizeof(__bf16),
                                          (size_t)2 * 3 * rows_pad_of(Bmax), st);
                    if (he != hipSuccess) return (int)he;
                }
            }
        }
    }
    if (steps) {
        switch (H) {
            case 100: return run_steps_bwd<100>(n, ly, sb, stt, st);
            case 200: return run_steps_bwd<200>(n, ly, sb, stt, st);
            default: return run_steps_bwd<400>(n, ly, sb, stt, st);
        }
    }
    if (f32) {
        switch (H) {
            case 100: return launch_bwd<100>(ps, n, ly, st);
            case 200: return launch_bwd<200>(ps, n, ly, st);
            default: return launch_bwd<400>(ps, n, ly, st);
        }
    }
    if (npl == 1 && out_ld > 0) {
        switch (H) {
            case 100: return launch_bwd_bf<100, 1, true>(pb, n, ly, st);
            case 200: return launch_bwd_bf<200, 1, true>(pb, n, ly, st);
            default: return launch_bwd_bf<400, 1, true>(pb, n, ly, st);
        }
    }
    if (npl == 1) {
        switch (H) {
            case 100: return launch_bwd_bf<100, 1>(pb, n, ly, st);
            case 200: return launch_bwd_bf<200, 1>(pb, n, ly, st);
            default: return launch_bwd_bf<400, 1>(pb, n, ly, st);
        }
    }
    switch (H) {
        case 100: return launch_bwd_bf<100>(pb, n, ly, st);
        case 200: return launch_bwd_bf<200>(pb, n, ly, st);
        default: return launch_bwd_bf<400>(pb, n, ly, st);
    }
}

// n (<= 4) GRUs over ONE packed layout
int renet_gru_fwd_multi(int n, const float* const* Gi, const int32_t* step_off, int L, int H,
                        const float* const* Whh, const float* const* bhh, float* const* h_last, int out_rows,
                        float* const* saved, float* workspace, size_t workspace_bytes, void* stream) {
    if (n < 1 || n > MAXP) return RENET_ERR_BADARG;
    const int32_t* so[MAXP];
    int Ls[MAXP], rows[MAXP];
    for (int k = 0; k < n; ++k) { so[k] = step_off; Ls[k] = L; rows[k] = out_rows; }
    return renet_gru_fwd_layouts(n, Gi, so, Ls, H, Whh, bhh, h_last, rows, saved, workspace, workspace_bytes, stream);
}

int renet_gru_fwd(const float* Gi, const int32_t* step_off, int L, int H, const float* Whh,
                  const float* bhh, float* h_last, int out_rows, float* saved, float* workspace,
                  size_t workspace_bytes, void* stream) {
    return renet_gru_fwd_multi(1, &Gi, step_off, L, H, &Whh, &bhh, &h_last, out_rows, &saved, workspace,
                               workspace_bytes, stream);
}

int renet_gru_bwd_multi(int n, const float* const* dh_last, const int32_t* step_off, int L, int H,
                        const float* const* Whh, const float* const* saved, float* const* dGi,
                        float* const* dGh, float* workspace, size_t workspace_bytes, void* stream) {
    if (n < 1 || n > MAXP) return RENET_ERR_BADARG;
    const int32_t* so[MAXP];
    int Ls[MAXP];
    for (int k = 0; k < n; ++k) { so[k] = step_off; Ls[k] = L; }
    return renet_gru_bwd_layouts(n, dh_last, so, Ls, H, Whh, saved, dGi, dGh, workspace, workspace_bytes, stream);
}

int renet_gru_bwd(const float* dh_last, const int32_t* step_off, int L, int H, const float* Whh,
                  const float* saved, float* dGi, float* dGh, float* workspace,
                  size_t workspace_bytes, void* stream) {
    return renet_gru_bwd_multi(1, &dh_last, step_off, L, H, &Whh, &saved, &dGi, &dGh, workspace,
                               workspace_bytes, stream);
}

}  // extern "C"
