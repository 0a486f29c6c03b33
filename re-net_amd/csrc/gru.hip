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

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MT = 16;           // sequences per workgroup (MFMA M)
constexpr int NW = 8;            // waves per workgroup: 2 per SIMD so that W_hh fetch latency hides behind the partner's MFMAs
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
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

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
                                                         size_t su, size_t sk, int NUBk, int KGk,
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
        const size_t base = ((size_t)((ub * KGk + kg) * G + g) * 3) * 64 + lane;
#pragma unroll
        for (int p = 0; p < 3; ++p) out[base + (size_t)p * 64] = o[p];
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

struct FwdProbB { const float* Gi; const bf16x8* Wp; const float* bhh; float* h_last; float* saved; };
struct FwdProbsB { FwdProbB p[MAXP]; };
struct BwdProbB { const float* dh_last; const bf16x8* WTp; const float* saved; float* dGi; float* dGh; };
struct BwdProbsB { BwdProbB p[MAXP]; };

// Wp: bf16 planes of W_hh in fragment order (split_frag_kernel with G = 3 gates)
template <int H>
__global__ __launch_bounds__(NT) void gru_fwd_bf_kernel(FwdProbsB ps, Layouts ly) {
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
    __bf16* Hp = reinterpret_cast<__bf16*>(smem + 2 * MT * C::LDH);      // [3][MT][LDP] bf16 planes of Hs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * MT;
    for (int t = tid; t < 2 * MT * C::LDH; t += NT) Hs[t] = 0.f;      // h0 = 0
    for (int t = tid; t < 3 * MT * Bc::LDP / 2; t += NT) reinterpret_cast<unsigned*>(Hp)[t] = 0u;   // and its planes
    __syncthreads();
    const int jj = lane & 15, kq = lane >> 4, ai = lane & 15;

    for (int j = 0; j < L; ++j) {
        const int p0 = so.off[j];
        const int bs = so.off[j + 1] - p0;
        if (i0 >= bs) break;                                            // whole tile finished (sorted batch)
#pragma unroll 1
        for (int ub = wave; ub < C::NUB; ub += NW) {
            const int u = ub * 16 + jj;                                 // this lane's hidden unit
            const bool uok = u < H;
            f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = ar, an = ar;
            const bf16x8* wf = Wp + (size_t)ub * Bc::KG * 9 * 64 + lane;       // fragment order (split_frag_kernel)
            const __bf16* ha = Hp + ai * Bc::LDP + kq * 8;
#pragma unroll 2
            for (int kg = 0; kg < Bc::KG; ++kg) {
                bf16x8 a[3], br[3], bz[3], bn[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a[p] = *reinterpret_cast<const bf16x8*>(ha + p * MT * Bc::LDP + kg * 32);
                    br[p] = wf[(kg * 9 + p) * 64];
                    bz[p] = wf[(kg * 9 + 3 + p) * 64];
                    bn[p] = wf[(kg * 9 + 6 + p) * 64];
                }
                ar = mfma6(a, br, ar);
                az = mfma6(a, bz, az);
                an = mfma6(a, bn, an);
            }
            // C layout: column = lane & 15 (unit u), row = 4 * (lane >> 4) + reg (sequence)
            if (uok) {
                const float b_r = bhh[u], b_z = bhh[H + u], b_n = bhh[2 * H + u];
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int i = 4 * kq + reg;
                    const float hp = Hs[i * C::LDH + u];
                    float hv = hp;
                    if (i0 + i < bs) {
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
                    Hn[i * C::LDH + u] = hv;
                }
            }
        }
        __syncthreads();                                                // every wave is done reading Hs / Hp
        for (int t = tid; t < MT * H; t += NT) {
            const int i = t / H, u = t - i * H;
            const float v = Hn[i * C::LDH + u];
            Hs[i * C::LDH + u] = v;
            const Planes3 s = split3(v);
#pragma unroll
            for (int p = 0; p < 3; ++p) Hp[p * MT * Bc::LDP + i * Bc::LDP + u] = s.p[p];
        }
        __syncthreads();
    }
    for (int t = tid; t < MT * H; t += NT) {                  // rows >= B were never touched: still h0 = 0
        const int i = t / H, u = t - i * H;
        if (i0 + i < out_rows) h_last[(size_t)(i0 + i) * H + u] = Hs[i * C::LDH + u];
    }
}

// WTp: bf16 planes of W_hh^T (unit = hidden unit, k over the 3H gate columns) in fragment order (G = 1)
template <int H>
__global__ __launch_bounds__(NT) void gru_bwd_bf_kernel(BwdProbsB ps, Layouts ly) {
    const int lay = ly.lay_of[blockIdx.y];
    const StepOff& so = ly.so[lay];
    const int L = ly.L[lay];
    if ((int)blockIdx.x * MT >= ly.rows[lay]) return;
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
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * MT;
    const int B = so.off[1] - so.off[0];
    for (int t = tid; t < MT * C::LDH; t += NT) {
        const int i = t / C::LDH, u = t - i * C::LDH;
        dHs[t] = (u < H && i0 + i < B) ? dh_last[(size_t)(i0 + i) * H + u] : 0.f;
    }
    for (int t = tid; t < 3 * MT * Bc::LDP3 / 2; t += NT) reinterpret_cast<unsigned*>(Gp)[t] = 0u;
    __syncthreads();
    const int jj = lane & 15, kq = lane >> 4, ai = lane & 15;
    constexpr int PLG = MT * Bc::LDP3;

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
            if (j > 0) {
                const Planes3 sr = split3(gr), sz = split3(gz), sn = split3(gn);
                __bf16* row = Gp + i * Bc::LDP3;
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    row[p * PLG + u] = sr.p[p];
                    row[p * PLG + H + u] = sz.p[p];
                    row[p * PLG + 2 * H + u] = sn.p[p];
                }
            }
        }
        __syncthreads();
        if (j > 0) {
            // phase 2: dh_prev += dGh W_hh  (rows of dead sequences have dGh = 0 and keep their dh)
#pragma unroll 1
            for (int ub = wave; ub < C::NUB; ub += NW) {
                {
                    const int u = ub * 16 + jj;
                    const bool uok = u < H;
                    f32x4 acc;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) acc[reg] = dHs[(4 * kq + reg) * C::LDH + (uok ? u : 0)];
                    const bf16x8* wf = WTp + (size_t)ub * Bc::KG3 * 3 * 64 + lane;
                    const __bf16* ga = Gp + ai * Bc::LDP3 + kq * 8;
                    f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};                  // two chains: no MFMA waits on the previous one
#pragma unroll 2
                    for (int kg = 0; kg < Bc::KG3; ++kg) {
                        bf16x8 a[3], b[3];
#pragma unroll
                        for (int p = 0; p < 3; ++p) {
                            a[p] = *reinterpret_cast<const bf16x8*>(ga + p * PLG + kg * 32);
                            b[p] = wf[(kg * 3 + p) * 64];
                        }
                        if (kg & 1) acc2 = mfma6(a, b, acc2);
                        else acc = mfma6(a, b, acc);
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

template <int H>
int launch_fwd_bf(const FwdProbsB& ps, int np, const Layouts& ly, hipStream_t st) {
    using C = Cfg<H>;
    const size_t lds = (size_t)2 * MT * C::LDH * sizeof(float) + (size_t)3 * MT * BCfg<H>::LDP * sizeof(__bf16);
    static bool attr_set = false;
    const int e = set_lds(gru_fwd_bf_kernel<H>, lds, attr_set);
    if (e != RENET_OK) return e;
    RENET_LAUNCH((gru_fwd_bf_kernel<H>), dim3((max_rows(ly) + MT - 1) / MT, np), dim3(NT), lds, st, ps, ly);
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

template <int H>
int launch_bwd_bf(const BwdProbsB& ps, int np, const Layouts& ly, hipStream_t st) {
    using C = Cfg<H>;
    const size_t lds = (size_t)MT * C::LDH * sizeof(float) + (size_t)3 * MT * BCfg<H>::LDP3 * sizeof(__bf16);
    static bool attr_set = false;
    const int e = set_lds(gru_bwd_bf_kernel<H>, lds, attr_set);
    if (e != RENET_OK) return e;
    RENET_LAUNCH((gru_bwd_bf_kernel<H>), dim3((max_rows(ly) + MT - 1) / MT, np), dim3(NT), lds, st, ps, ly);
    RENET_LAUNCH_CHECK();
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

int split_frag(const float* in, int U, int K, int G, size_t sg, size_t su, size_t sk, bf16x8* out, hipStream_t st) {
    const int NUBk = (U + 15) / 16, KGk = (K + 31) / 32;
    const int total = NUBk * KGk * G * 64;
    RENET_LAUNCH(split_frag_kernel, dim3((total + 255) / 256), dim3(256), 0, st, in, U, K, G, sg, su, sk, NUBk,
                       KGk, out);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

}  // namespace

extern "C" {

// per GRU: forward = the bf16 planes of W_hh; backward = those of W_hh^T (bf16x6) or W_hh^T in fp32 (RENET_GEMM=f32)
size_t renet_gru_workspace(int B, int H) {
    (void)B;
    size_t m = fwd_plane_bytes(H);
    if (bwd_t_bytes(H) > m) m = bwd_t_bytes(H);
    if (bwd_plane_bytes(H) > m) m = bwd_plane_bytes(H);
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

// W_hh planes are split once per DISTINCT weight pointer (the subject and object passes share the encoders)
int plane_slot(int k, const float* const* W) {
    for (int i = 0; i < k; ++i)
        if (W[i] == W[k]) return i;
    return k;
}

}  // namespace

extern "C" {

int renet_gru_fwd_layouts(int n, const float* const* Gi, const int32_t* const* step_off, const int* L, int H,
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
    if (use_f32()) {
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
    const size_t per = renet_gru_workspace(0, H);
    if (!workspace || workspace_bytes < (size_t)n * per) return RENET_ERR_WORKSPACE;
    FwdProbsB ps;
    for (int i = 0; i < MAXP; ++i) {
        const int k = i < n ? i : 0;
        const int slot = plane_slot(k, Whh);
        bf16x8* planes = reinterpret_cast<bf16x8*>(reinterpret_cast<char*>(workspace) + (size_t)slot * per);
        if (i < n && slot == k) {                   // gate g of unit u, input k: W_hh[g*H + u][k]
            const int e = split_frag(Whh[k], H, H, 3, (size_t)H * H, (size_t)H, 1, planes, st);
            if (e != RENET_OK) return e;
        }
        ps.p[i].Gi = Gi[k]; ps.p[i].Wp = planes; ps.p[i].bhh = bhh[k]; ps.p[i].h_last = h_last[k];
        ps.p[i].saved = saved[k];
    }
    switch (H) {
        case 100: return launch_fwd_bf<100>(ps, n, ly, st);
        case 200: return launch_fwd_bf<200>(ps, n, ly, st);
        default: return launch_fwd_bf<400>(ps, n, ly, st);
    }
}

int renet_gru_bwd_layouts(int n, const float* const* dh_last, const int32_t* const* step_off, const int* L, int H,
                          const float* const* Whh, const float* const* saved, float* const* dGi,
                          float* const* dGh, float* workspace, size_t workspace_bytes, void* stream) {
    if (n < 1 || n > MAXP) return RENET_ERR_BADARG;
    Layouts ly;
    int B_of[MAXP];
    const int e0 = make_layouts(n, step_off, L, nullptr, false, ly, B_of);
    if (e0 != RENET_OK) return e0;
    if (max_rows(ly) == 0) return RENET_OK;
    if (H != 100 && H != 200 && H != 400) return RENET_ERR_UNSUPPORTED;
    const size_t per = renet_gru_workspace(0, H);
    if (!workspace || workspace_bytes < (size_t)n * per) return RENET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const bool f32 = use_f32();
    BwdProbs ps;
    BwdProbsB pb;
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
                const int e = split_frag(Whh[k], H, 3 * H, 1, 0, 1, (size_t)H, planes, st);
                if (e != RENET_OK) return e;
            }
        }
        ps.p[i].dh_last = dh_last[k]; ps.p[i].WhhT = WhhT; ps.p[i].saved = saved[k]; ps.p[i].dGi = dGi[k];
        ps.p[i].dGh = dGh[k];
        pb.p[i].dh_last = dh_last[k]; pb.p[i].WTp = planes; pb.p[i].saved = saved[k]; pb.p[i].dGi = dGi[k];
        pb.p[i].dGh = dGh[k];
    }
    if (f32) {
        switch (H) {
            case 100: return launch_bwd<100>(ps, n, ly, st);
            case 200: return launch_bwd<200>(ps, n, ly, st);
            default: return launch_bwd<400>(ps, n, ly, st);
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
