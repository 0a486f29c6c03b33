// Sequence assembly, score-head feature concat, fused softmax cross-entropy and the global model's
// per-graph readout.  All element-wise / row-wise HBM-bound kernels: float4 lanes, fused dropout.
#include "common.h"

namespace {

// Magnitude bounds for the f16x3 GEMMs (gemm_h3.h), produced by the kernel that WRITES the operand instead of a
// separate pass over it: every workgroup's maximum of |value written| goes to part[blockIdx.x] (grid <= 1024 = the
// number of partial maxima renet_gemm_f32_h3 accepts).  The maximum over the partials is the tensor's exact maximum,
// so the GEMMs scale -- and round -- exactly as with renet_maxabs_partials.
__device__ __forceinline__ float sq_max4(float m, const float4& v) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
__device__ __forceinline__ void sq_block_max_write(float m, float* __restrict__ part, float* red /* [4] LDS */) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
}

// ---- Aggregator.py:142-165: packed GRU inputs X [S,4D], Xr [S,3D] ----------------------------
__global__ __launch_bounds__(256) void seq_assemble_fwd_kernel(
    const float4* __restrict__ h2, const float4* __restrict__ ent, const float4* __restrict__ rel,
    const float4* __restrict__ glob, const int32_t* __restrict__ subj_row,
    const int32_t* __restrict__ row_ent, const int32_t* __restrict__ row_rel,
    const int32_t* __restrict__ glob_row, int S, int CH, DropCfg dx, DropCfg dxr,
    float4* __restrict__ X, float4* __restrict__ Xr, float* __restrict__ partX, float* __restrict__ partXr) {
    __shared__ float red[4];
    float mx = 0.f, mxr = 0.f;
    const size_t total = (size_t)S * 4 * CH;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i / (4 * CH));
        const int c = (int)(i - (size_t)p * 4 * CH);
        const int part = c / CH, cc = c - part * CH;
        float4 v;
        if (part == 0) v = h2[(size_t)subj_row[p] * CH + cc];
        else if (part == 1) v = ent[(size_t)row_ent[p] * CH + cc];
        else if (part == 2) v = rel[(size_t)row_rel[p] * CH + cc];
        else v = glob[(size_t)glob_row[p] * CH + cc];
        const float4 o = f4_mul(v, renet_drop4(dx, i));
        X[i] = o;
        mx = sq_max4(mx, o);
        if (part != 2) {                                   // Xr = [h2 | ent | glob]
            const int cr = (part == 3 ? 2 : part) * CH + cc;
            const size_t ir = (size_t)p * 3 * CH + cr;
            const float4 orr = f4_mul(v, renet_drop4(dxr, ir));
            Xr[ir] = orr;
            mxr = sq_max4(mxr, orr);
        }
    }
    if (partX) {                                           // kernel-uniform
        sq_block_max_write(mx, partX, red);
        sq_block_max_write(mxr, partXr, red);
    }
}

// The same rows written as bf16 (RNE) straight into the zero-padded matrices the bf16-storage GEMMs consume (row
// strides ldx / ldxr in elements): no fp32 X / Xr exists in bf16 mode.  A thread converts one float4 -> 8 bytes.
typedef float sq_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sq_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 sq_pack4(float4 v) {
    const sq_bf16x2 lo = __builtin_convertvector(sq_f32x2{v.x, v.y}, sq_bf16x2);
    const sq_bf16x2 hi = __builtin_convertvector(sq_f32x2{v.z, v.w}, sq_bf16x2);
    uint2 u;
    u.x = __builtin_bit_cast(unsigned, lo);
    u.y = __builtin_bit_cast(unsigned, hi);
    return u;
}
__global__ __launch_bounds__(256) void seq_assemble_fwd_bf16_kernel(
    const float4* __restrict__ h2, const float4* __restrict__ ent, const float4* __restrict__ rel,
    const float4* __restrict__ glob, const int32_t* __restrict__ subj_row,
    const int32_t* __restrict__ row_ent, const int32_t* __restrict__ row_rel,
    const int32_t* __restrict__ glob_row, int S, int CH, DropCfg dx, DropCfg dxr,
    __bf16* __restrict__ X, int ldx, __bf16* __restrict__ Xr, int ldxr) {
    const size_t total = (size_t)S * 4 * CH;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i / (4 * CH));
        const int c = (int)(i - (size_t)p * 4 * CH);
        const int part = c / CH, cc = c - part * CH;
        float4 v;
        if (part == 0) v = h2[(size_t)subj_row[p] * CH + cc];
        else if (part == 1) v = ent[(size_t)row_ent[p] * CH + cc];
        else if (part == 2) v = rel[(size_t)row_rel[p] * CH + cc];
        else v = glob[(size_t)glob_row[p] * CH + cc];
        // the dropout mask is keyed by the float4-group index of the UNPADDED [S, 4D] / [S, 3D] layouts, exactly as in
        // the fp32 kernel (the backward kernels regenerate it from the same index)
        *reinterpret_cast<uint2*>(X + (size_t)p * ldx + c * 4) = sq_pack4(f4_mul(v, renet_drop4(dx, i)));
        if (part != 2) {
            const int cr = (part == 3 ? 2 : part) * CH + cc;
            const size_t ir = (size_t)p * 3 * CH + cr;
            *reinterpret_cast<uint2*>(Xr + (size_t)p * ldxr + cr * 4) = sq_pack4(f4_mul(v, renet_drop4(dxr, ir)));
        }
    }
}

// backward, part 1: gradient wrt the gathered h2 row of every packed row (X and Xr segments)
__global__ __launch_bounds__(256) void seq_assemble_bwd_rows_kernel(const float4* __restrict__ dX,
                                                                    const float4* __restrict__ dXr, int S,
                                                                    int CH, DropCfg dx, DropCfg dxr,
                                                                    float4* __restrict__ dRows) {
    const size_t total = (size_t)S * CH;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i / CH), cc = (int)(i - (size_t)p * CH);
        const size_t bx = (size_t)p * 4 * CH, br = (size_t)p * 3 * CH;
        const float4 a = f4_mul(dX[bx + cc], renet_drop4(dx, bx + cc));
        const float4 b = f4_mul(dXr[br + cc], renet_drop4(dxr, br + cc));
        dRows[i] = f4_add(a, b);
    }
}

// backward, part 2: ent[s_i] and rel[r_i] were broadcast to every step of sequence i
// (Aggregator.py:150-155), so their gradients are summed over the sequence's steps here: packed row of
// step j of sequence i is off[j] + i while i < off[j+1] - off[j].  Rows i >= nnz come out zero.
__global__ __launch_bounds__(256) void seq_assemble_bwd_seq_kernel(const float4* __restrict__ dX,
                                                                   const float4* __restrict__ dXr,
                                                                   const int32_t* __restrict__ off, int L,
                                                                   int B, int CH, DropCfg dx, DropCfg dxr,
                                                                   float4* __restrict__ dEntSeq,
                                                                   float4* __restrict__ dRelSeq) {
    const size_t total = (size_t)B * CH;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(t / CH), cc = (int)(t - (size_t)i * CH);
        float4 se = make_float4(0.f, 0.f, 0.f, 0.f), sr = se;
        for (int j = 0; j < L; ++j) {
            const int o0 = off[j], o1 = off[j + 1];
            if (i >= o1 - o0) break;
            const size_t p = (size_t)(o0 + i);
            const size_t bx = p * 4 * CH, br = p * 3 * CH;
            se = f4_add(se, f4_mul(dX[bx + CH + cc], renet_drop4(dx, bx + CH + cc)));
            se = f4_add(se, f4_mul(dXr[br + CH + cc], renet_drop4(dxr, br + CH + cc)));
            sr = f4_add(sr, f4_mul(dX[bx + 2 * CH + cc], renet_drop4(dx, bx + 2 * CH + cc)));
        }
        dEntSeq[t] = se;
        dRelSeq[t] = sr;
    }
}

// ---- model.py:89-90 / 98-99: feat = drop([a[ia] | hmid | c[ic]]) -------------------------------
__global__ __launch_bounds__(256) void concat3_fwd_kernel(const float4* __restrict__ a,
                                                          const int32_t* __restrict__ ia,
                                                          const float4* __restrict__ hmid,
                                                          const float4* __restrict__ c,
                                                          const int32_t* __restrict__ ic, int B, int CH,
                                                          int parts, DropCfg d, float4* __restrict__ feat,
                                                          float* __restrict__ bound_part) {
    __shared__ float red[4];
    float mx = 0.f;
    const size_t total = (size_t)B * parts * CH;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / (parts * CH));
        const int q = (int)(i - (size_t)b * parts * CH);
        const int part = q / CH, cc = q - part * CH;
        float4 v;
        if (part == 0) v = a[(size_t)ia[b] * CH + cc];
        else if (part == 1) v = hmid[(size_t)b * CH + cc];
        else v = c[(size_t)ic[b] * CH + cc];
        const float4 o = f4_mul(v, renet_drop4(d, i));
        feat[i] = o;
        mx = sq_max4(mx, o);
    }
    if (bound_part) sq_block_max_write(mx, bound_part, red);
}

__global__ __launch_bounds__(256) void concat3_bwd_kernel(const float4* __restrict__ dfeat, int B, int CH,
                                                          int parts, DropCfg d, float4* __restrict__ da,
                                                          float4* __restrict__ dh, float4* __restrict__ dc) {
    const size_t total = (size_t)B * parts * CH;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / (parts * CH));
        const int q = (int)(i - (size_t)b * parts * CH);
        const int part = q / CH, cc = q - part * CH;
        const float4 v = f4_mul(dfeat[i], renet_drop4(d, i));
        const size_t o = (size_t)b * CH + cc;
        if (part == 0) da[o] = v;
        else if (part == 1) dh[o] = v;
        else dc[o] = v;
    }
}

// ---- fused log-softmax + NLL (+ gradient) : one workgroup per row -----------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// `dlogits` may ALIAS `logits` (the training path turns the logits into their gradient in place): neither pointer
// may be __restrict__ -- with it hipcc is free to sink thread 0's read of x[target] below the barrier, past other
// waves' stores to the same row (seen as a rare wrong LOSS with correct gradients).
// dl16 (optional, instead of dlogits): the gradient as bf16 (RNE) into a matrix with row stride ld16 (elements), the
// columns [C, C16) of every row written as zeros (C16 = C rounded up to 64: the k padding a bf16-storage GEMM reads).
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* logits,
                                                         const int32_t* __restrict__ target, int C, int ld,
                                                         float grad_scale, float* __restrict__ row_loss,
                                                         float* dlogits, __bf16* __restrict__ dl16, int ld16) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* x = logits + (size_t)b * ld;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < C; c += 256) m = fmaxf(m, x[c]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) s += expf(x[c] - m);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    const int t = target[b];
    float xt = x[t];                            // read before dlogits may overwrite (alias allowed)
    // pin the read HERE: hipcc turns the block-uniform x[t] into a scalar load and sinks it into the thread-0 branch
    // below the barrier (checked in the ISA), i.e. behind other waves' in-place stores to the row -- seen as a rare
    // wrong LOSS with correct gradients
    asm volatile("" : "+v"(xt) :: "memory");
    __syncthreads();
    if (threadIdx.x == 0) row_loss[b] = logf(s) + m - xt;
    if (dlogits) {
        float* dx = dlogits + (size_t)b * ld;
        const float inv = 1.f / s;
        for (int c = threadIdx.x; c < C; c += 256) {
            const float p = expf(x[c] - m) * inv;
            dx[c] = (p - (c == t ? 1.f : 0.f)) * grad_scale;
        }
    }
    if (dl16) {
        __bf16* dx = dl16 + (size_t)b * ld16;
        const float inv = 1.f / s;
        const int C16 = min((C + 63) & ~63, ld16);
        for (int c = threadIdx.x; c < C16; c += 256)
            dx[c] = c < C ? (__bf16)((expf(x[c] - m) * inv - (c == t ? 1.f : 0.f)) * grad_scale) : (__bf16)0.f;
    }
}

// Same for fp32 output with the row held in REGISTERS (round 4): 512 threads, thread t owns columns t, t + 512, ...
// (NQ <= 48 of them), every load of the row is in flight at once, max / exp / sum / write run on registers, LDS only
// carries the two 8-value reductions.  No 92 KB of LDS per workgroup and <= 128 VGPRs => TWO workgroups (rows) per CU,
// so the load of one row overlaps the exp / store of another (the LDS kernel below runs its phases back to back, one
// row per CU at a time).  In place (dlogits == logits) is safe by construction: a thread reads all of its own columns
// before it writes any, and nobody else touches them.
template <int NQ>
__global__ __launch_bounds__(512, 4) void softmax_ce_reg_kernel(const float* logits, const int32_t* __restrict__ target,
                                                                int C, int ld, float grad_scale,
                                                                float* __restrict__ row_loss, float* dlogits) {
    __shared__ float red_m[8], red_s[8];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = target[b];
    // the row through a raw buffer descriptor of exactly C floats: ONE vector offset (thread * 4) for all NQ accesses, the
    // column block in the scalar offset; columns >= C read 0 and their stores are dropped by the range check
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(logits + (size_t)b * ld), (short)0, C * 4, 0x00020000);
    const int voff = (int)threadIdx.x * 4;
    float v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        v[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rin, voff, 2048 * q, 0));
    float m = -INFINITY, xt = 0.f;
    const int tq = t >> 9;                              // column t is element tq of thread t & 511
    const bool mine = (int)threadIdx.x == (t & 511);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        // column threadIdx.x + 512 q < C, written against a SCALAR bound (the compiler otherwise keeps all NQ column
        // indices alive, and spills them); the only use of the tail mask: exp(-inf) = 0 below
        v[q] = (int)threadIdx.x < C - 512 * q ? v[q] : -INFINITY;
        m = fmaxf(m, v[q]);
        xt = (mine && q == tq) ? v[q] : xt;
    }
    m = wave_max(m);
    if (lane == 0) red_m[wave] = m;
    __syncthreads();
    m = red_m[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) m = fmaxf(m, red_m[w]);
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        v[q] = expf(v[q] - m);
        s += v[q];
        // eight expf at a time: scheduled all at once the NQ expansions' temporaries overflow the 128-register budget
        if ((q & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    s = wave_sum(s);
    if (lane == 0) red_s[wave] = s;
    __syncthreads();
    s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += red_s[w];
    if (mine) row_loss[b] = logf(s) + m - xt;                                   // the owner of column t
    if (dlogits) {
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(dlogits + (size_t)b * ld, (short)0, C * 4,
                                                                             0x00020000);
        const float inv = 1.f / s;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float o = (v[q] * inv - ((mine && q == tq) ? 1.f : 0.f)) * grad_scale;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), rout, voff, 2048 * q, 0);
        }
    }
}

// Same, with the row staged in LDS (C * 4 <= 128 KB: 23 033 entities = 92 KB): the three passes of the kernel above
// re-read the row from the fabric (all 1024 rows are in flight at once, 94 MB against 32 MB of L2); here the row is
// read ONCE by 1024 threads (16 waves keep enough loads in flight for one workgroup per CU) and the max / sum / write
// passes run out of LDS.
__global__ __launch_bounds__(1024) void softmax_ce_lds_kernel(const float* logits,
                                                              const int32_t* __restrict__ target, int C, int ld,
                                                              float grad_scale, float* __restrict__ row_loss,
                                                              float* dlogits, __bf16* __restrict__ dl16, int ld16) {
    extern __shared__ float row[];
    __shared__ float red[16];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* x = logits + (size_t)b * ld;
    float m = -INFINITY;
    // eight row loads in flight per thread (written as one load + wait + LDS store per iteration the compiler kept
    // exactly one: 22 serialised HBM latencies per thread for a 23 033-wide row)
    int c = threadIdx.x;
    for (; c + 7 * 1024 < C; c += 8 * 1024) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = x[c + q * 1024];
        __builtin_amdgcn_sched_barrier(0);           // all eight requested before the first is consumed
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            row[c + q * 1024] = v[q];
            m = fmaxf(m, v[q]);
        }
    }
    for (; c < C; c += 1024) {
        const float v = x[c];
        row[c] = v;
        m = fmaxf(m, v);
    }
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
    const int t = target[b];
    const float xt = row[t];                         // the staged copy, before the exp pass overwrites it
    __syncthreads();
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 1024) {
        const float e = expf(row[c] - m);
        row[c] = e;                                  // each thread revisits only its own elements
        s += e;
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += red[w];
    if (threadIdx.x == 0) row_loss[b] = logf(s) + m - xt;
    if (dlogits) {
        float* dx = dlogits + (size_t)b * ld;
        const float inv = 1.f / s;
        for (int c = threadIdx.x; c < C; c += 1024) dx[c] = (row[c] * inv - (c == t ? 1.f : 0.f)) * grad_scale;
    }
    if (dl16) {                                      // two columns per thread: 4-byte stores
        unsigned* dx = reinterpret_cast<unsigned*>(dl16 + (size_t)b * ld16);
        const float inv = 1.f / s;
        const int C16 = min((C + 63) & ~63, ld16);
        for (int c = 2 * threadIdx.x; c < C16; c += 2048) {
            const float v0 = c < C ? (row[c] * inv - (c == t ? 1.f : 0.f)) * grad_scale : 0.f;
            const float v1 = c + 1 < C ? (row[c + 1] * inv - (c + 1 == t ? 1.f : 0.f)) * grad_scale : 0.f;
            typedef float f32x2_ __attribute__((ext_vector_type(2)));
            typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
            const bf16x2_ pk = __builtin_convertvector(f32x2_{v0, v1}, bf16x2_);
            dx[c >> 1] = __builtin_bit_cast(unsigned, pk);
        }
    }
}

// ---- the CE gradient as THREE bf16 PLANES (round 6): the operand format of renet_gemm_planes (gemm_p6.h) -----------
// dl = (softmax - onehot) * grad_scale is never stored as fp32: each value leaves this kernel as p1 = rne(dl),
// p2 = rne(dl - p1), p3 = rne(dl - p1 - p2) -- the split the bf16x6 GEMM loaders (gemm_split.hip: store_items) do per k-tile
// -- into planes [3][rows16][ld16]; 6 bytes per element instead of 4, and the backward GEMMs that consume it (dfeat, dW)
// run without any conversion work in their k-loops.  Columns [C, C64) of every row are written as zeros.
typedef uint32_t sq_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t sq_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3_f4(float4 v, sq_u32x2 (&out)[3]) {
    sq_f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const sq_bf16x2 blo = __builtin_convertvector(lo, sq_bf16x2);
        const sq_bf16x2 bhi = __builtin_convertvector(hi, sq_bf16x2);
        out[p].x = __builtin_bit_cast(uint32_t, blo);
        out[p].y = __builtin_bit_cast(uint32_t, bhi);
        if (p < 2) {
            lo -= __builtin_convertvector(blo, sq_f32x2);
            hi -= __builtin_convertvector(bhi, sq_f32x2);
        }
    }
}

// Round 6 (late): FOUR rows per workgroup, whole 128-byte lines out.  The 16 rows of a T16 tile share every line of the
// output (a row owns 32 bytes of each 512-byte tile); with one row per workgroup every store instruction wrote 16 separate
// 32-byte pieces and the lines were only completed in L2 by the workgroups of the neighbouring rows (137 us in the step for
// 2048 x 23033 against 68 us for the fp32 in-place kernel).  Here a workgroup keeps rows 4 j .. 4 j + 3 of a tile in
// registers (4 x 12 float4 per thread), takes the four row statistics, and in the store phase stages each 2048-column
// slab -- 128 tiles x 3 planes x (4 rows x 32 bytes) -- through LDS so that every lane stores 16 bytes and 8 lanes one
// whole line.  Rows >= B of the last quad write zeros.
constexpr int SP4_STRIDE = 144;                         // bytes between two tiles' 128-byte blocks in the staging buffer
constexpr int SP4_PLANE = 128 * SP4_STRIDE;             // one plane of a slab

template <int NQ4>
__global__ __launch_bounds__(512) void softmax_ce_planes4_kernel(const float* __restrict__ logits,
                                                                 const int32_t* __restrict__ target, int B, int C, int ld,
                                                                 float grad_scale, float* __restrict__ row_loss,
                                                                 __bf16* __restrict__ P, size_t plane, int ld16) {
    __shared__ float red_m[8][4], red_s[8][4];
    __shared__ __attribute__((aligned(16))) char stage[3 * SP4_PLANE];
    // block L -> XCD L & 7: the four quads of a tile row take consecutive slots of one XCD (see softmax_ce_planes_kernel)
    const int L = (int)blockIdx.x;
    const int tile_row = (L >> 5) * 8 + (L & 7), quad = (L >> 3) & 3;
    const int b0 = tile_row * 16 + quad * 4;
    if (b0 >= B) return;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int voff = tid * 16, c0 = tid * 4;
    float4 v[4][NQ4];
    int tgt[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = min(b0 + r, B - 1);               // (a dead row re-reads the last live one; its outputs are zeroed)
        tgt[r] = target[b];
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(logits + (size_t)b * ld), (short)0, C * 4, 0x00020000);
#pragma unroll
        for (int q = 0; q < NQ4; ++q) {
            const sq_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rin, voff, 8192 * q, 0);
            v[r][q] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
        }
    }
    float m[4], xt[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int tg = tgt[r] >> 2, tc = tgt[r] & 3;
        const bool mine = tid == (tg & 511);
        const int tq = tg >> 9;
        m[r] = -INFINITY;
#pragma unroll
        for (int q = 0; q < NQ4; ++q) {
            const int lim = C - 2048 * q;
            float4& x = v[r][q];
            x.x = c0 < lim ? x.x : -INFINITY;
            x.y = c0 + 1 < lim ? x.y : -INFINITY;
            x.z = c0 + 2 < lim ? x.z : -INFINITY;
            x.w = c0 + 3 < lim ? x.w : -INFINITY;
            m[r] = fmaxf(fmaxf(m[r], fmaxf(x.x, x.y)), fmaxf(x.z, x.w));
            if (mine && q == tq) xt[r] = tc == 0 ? x.x : tc == 1 ? x.y : tc == 2 ? x.z : x.w;
        }
        m[r] = wave_max(m[r]);
        if (lane == 0) red_m[wave][r] = m[r];
    }
    __syncthreads();
    float s[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float mm = red_m[0][r];
#pragma unroll
        for (int w = 1; w < 8; ++w) mm = fmaxf(mm, red_m[w][r]);
        m[r] = mm;
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < NQ4; ++q) {
            float4& x = v[r][q];
            x.x = expf(x.x - mm); x.y = expf(x.y - mm); x.z = expf(x.z - mm); x.w = expf(x.w - mm);
            acc += (x.x + x.y) + (x.z + x.w);
            if ((q & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        acc = wave_sum(acc);
        if (lane == 0) red_s[wave][r] = acc;
    }
    __syncthreads();
    float inv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) acc += red_s[w][r];
        s[r] = acc;
        const bool live = b0 + r < B;
        if (live && tid == ((tgt[r] >> 2) & 511)) row_loss[b0 + r] = logf(acc) + m[r] - xt[r];
        inv[r] = live ? grad_scale / acc : 0.f;         // (a dead row: zeros)
    }
    // store phase.  T16 (common.h): rows 4 quad + r of tile column tc are the 128-byte block (quad ^ (tc & 1)) of the tile,
    // row r at r * 32, its 8-column halves swapped when quad >= 2.
    const int tcl = tid >> 2;                                           // this thread's tile within a slab of 128 tiles
    const int hq = (quad >> 1) & 1;
    char* wr = stage + tcl * SP4_STRIDE + ((((tid >> 1) & 1) ^ hq) * 16) + (tid & 1) * 8;
    // read side: 16-byte piece `pc` of tile `rt`, plane `rp` for k = 0 .. 5: linear index tid + 512 k over (plane, tile, piece)
    __amdgpu_buffer_rsrc_t rout[3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
        rout[p] = __builtin_amdgcn_make_buffer_rsrc(P + (size_t)p * plane + (size_t)tile_row * ld16 * 16, (short)0,
                                                    ld16 * 32, 0x00020000);
#pragma unroll
    for (int q = 0; q < NQ4; ++q) {
        if (q) __syncthreads();                                         // the previous slab has been read
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tg = tgt[r] >> 2, tc = tgt[r] & 3;
            float4 o = make_float4(v[r][q].x * inv[r], v[r][q].y * inv[r], v[r][q].z * inv[r], v[r][q].w * inv[r]);
            if (tid == (tg & 511) && q == (tg >> 9) && b0 + r < B) {
                const float one = grad_scale;
                if (tc == 0) o.x -= one; else if (tc == 1) o.y -= one; else if (tc == 2) o.z -= one; else o.w -= one;
            }
            sq_u32x2 pk[3];
            split3_f4(o, pk);
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<sq_u32x2*>(wr + p * SP4_PLANE + r * 32) = pk[p];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int n = tid + 512 * k;
            const int rp = n >> 10, rt = (n >> 3) & 127, pc = n & 7;
            const sq_u32x4 d = *reinterpret_cast<const sq_u32x4*>(stage + rp * SP4_PLANE + rt * SP4_STRIDE + pc * 16);
            const int tcg = q * 128 + rt;
            if (tcg * 16 < ld16) {                                      // (columns in [C, ld16) were written as exact zeros)
                const int vo = rt * 512 + ((quad ^ (rt & 1)) * 128) + pc * 16;
                if (rp == 0) __builtin_amdgcn_raw_buffer_store_b128(d, rout[0], vo, 65536 * q, 0);
                else if (rp == 1) __builtin_amdgcn_raw_buffer_store_b128(d, rout[1], vo, 65536 * q, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(d, rout[2], vo, 65536 * q, 0);
            }
            if (k & 1) __builtin_amdgcn_sched_barrier(0);               // two pieces in flight (register budget: 4 x 48 row values)
        }
    }
}

// Row in registers (the structure of softmax_ce_reg_kernel): 512 threads, thread t owns the float4 groups at columns
// 4 t + 2048 q, q < NQ4 (<= 12: C <= 24 576); needs 16-byte aligned rows (ld % 4 == 0).
template <int NQ4>
__global__ __launch_bounds__(512, 4) void softmax_ce_planes_kernel(const float* __restrict__ logits,
                                                                   const int32_t* __restrict__ target, int B, int C, int ld,
                                                                   float grad_scale, float* __restrict__ row_loss,
                                                                   __bf16* __restrict__ P, size_t plane, int ld16) {
    __shared__ float red_m[8], red_s[8];
    // row of this workgroup: the 16 rows of a T16 tile row share every 128-byte line of the output (a row contributes 32
    // bytes per tile), and the dispatcher deals workgroups to the XCDs round-robin (block L -> XCD L & 7, each with its own
    // write-back L2) -- so the 16 workgroups of a tile row are given CONSECUTIVE slots of ONE XCD: its L2 merges their
    // pieces into whole lines before they leave for HBM.  (With row = blockIdx.x eight L2s each held an eighth of every
    // line: 236 us instead of 95 for the 2048 x 23033 gradient.)
    const int b = ((((int)blockIdx.x >> 7) * 8 + ((int)blockIdx.x & 7)) << 4) + (((int)blockIdx.x >> 3) & 15);
    if (b >= B) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = target[b];
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(logits + (size_t)b * ld), (short)0, C * 4, 0x00020000);
    const int voff = (int)threadIdx.x * 16;
    float4 v[NQ4];
#pragma unroll
    for (int q = 0; q < NQ4; ++q) {
        const sq_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rin, voff, 8192 * q, 0);
        v[q] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
    }
    float m = -INFINITY, xt = 0.f;
    const int tg = t >> 2;                              // column t is component t & 3 of float4 group tg
    const bool mine = (int)threadIdx.x == (tg & 511);
    const int tq = tg >> 9, tc = t & 3;
    const int c0 = (int)threadIdx.x * 4;
#pragma unroll
    for (int q = 0; q < NQ4; ++q) {
        const int lim = C - 2048 * q;                   // scalar bound: columns c0 + j < lim are real
        v[q].x = c0 < lim ? v[q].x : -INFINITY;
        v[q].y = c0 + 1 < lim ? v[q].y : -INFINITY;
        v[q].z = c0 + 2 < lim ? v[q].z : -INFINITY;
        v[q].w = c0 + 3 < lim ? v[q].w : -INFINITY;
        m = fmaxf(fmaxf(m, fmaxf(v[q].x, v[q].y)), fmaxf(v[q].z, v[q].w));
        if (mine && q == tq) xt = tc == 0 ? v[q].x : tc == 1 ? v[q].y : tc == 2 ? v[q].z : v[q].w;
    }
    m = wave_max(m);
    if (lane == 0) red_m[wave] = m;
    __syncthreads();
    m = red_m[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) m = fmaxf(m, red_m[w]);
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NQ4; ++q) {
        v[q].x = expf(v[q].x - m); v[q].y = expf(v[q].y - m); v[q].z = expf(v[q].z - m); v[q].w = expf(v[q].w - m);
        s += (v[q].x + v[q].y) + (v[q].z + v[q].w);
        if ((q & 1) == 1) __builtin_amdgcn_sched_barrier(0);       // eight expf at a time (register budget, as above)
    }
    s = wave_sum(s);
    if (lane == 0) red_s[wave] = s;
    __syncthreads();
    s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += red_s[w];
    if (mine) row_loss[b] = logf(s) + m - xt;
    // stores in T16 format (common.h): row b lives in tile row b >> 4; this thread's float4 group q covers columns
    // 4 t + 2048 q = tile column (t >> 2) + 128 q, tile-row half (t >> 1) & 1, 8 bytes at (t & 1) * 8 -- all per-thread
    // constants but the 64 KB step per q.  One descriptor per plane over the tile row (ld16 / 16 tiles of 512 bytes).
    const int i16 = b & 15;
    const int tcl = (int)threadIdx.x >> 2;
    const int vo = tcl * 512 + ((i16 ^ ((tcl & 1) << 2)) * 32) + (((((int)threadIdx.x >> 1) & 1) ^ ((i16 >> 3) & 1)) * 16) +
                   ((int)threadIdx.x & 1) * 8;
    __amdgpu_buffer_rsrc_t rout[3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
        rout[p] = __builtin_amdgcn_make_buffer_rsrc(P + (size_t)p * plane + (size_t)(b >> 4) * ld16 * 16, (short)0,
                                                    ld16 * 32, 0x00020000);
    const float inv = 1.f / s;
#pragma unroll
    for (int q = 0; q < NQ4; ++q) {
        float4 o = make_float4(v[q].x * inv, v[q].y * inv, v[q].z * inv, v[q].w * inv);
        if (mine && q == tq) {
            if (tc == 0) o.x -= 1.f; else if (tc == 1) o.y -= 1.f; else if (tc == 2) o.z -= 1.f; else o.w -= 1.f;
        }
        o = f4_scale(o, grad_scale);
        sq_u32x2 pk[3];
        split3_f4(o, pk);
        if (c0 < ld16 - 2048 * q) {                     // (columns in [C, ld16) are exact zeros: the k / tile padding)
#pragma unroll
            for (int p = 0; p < 3; ++p) __builtin_amdgcn_raw_buffer_store_b64(pk[p], rout[p], vo, 65536 * q, 0);
        }
    }
}

// any width / alignment: three passes over the row (the structure of softmax_ce_kernel), planes out
__global__ __launch_bounds__(256) void softmax_ce_planes_generic_kernel(const float* __restrict__ logits,
                                                                        const int32_t* __restrict__ target, int C, int ld,
                                                                        float grad_scale, float* __restrict__ row_loss,
                                                                        __bf16* __restrict__ P, size_t plane, int ld16) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* x = logits + (size_t)b * ld;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < C; c += 256) m = fmaxf(m, x[c]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) s += expf(x[c] - m);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    const int t = target[b];
    if (threadIdx.x == 0) row_loss[b] = logf(s) + m - x[t];
    const float inv = 1.f / s;
    const int tcn = ld16 >> 4;
    for (int c = threadIdx.x; c < ld16; c += 256) {
        float r = c < C ? (expf(x[c] - m) * inv - (c == t ? 1.f : 0.f)) * grad_scale : 0.f;
        __bf16* dx = P + renet_t16_off(b, c, tcn);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const __bf16 h = (__bf16)r;
            dx[(size_t)p * plane] = h;
            r -= (float)h;
        }
    }
}

__global__ __launch_bounds__(256) void zero_kernel(float4* __restrict__ x, size_t n4, float* __restrict__ xt, int tail) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) x[i] = z;
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) xt[threadIdx.x] = 0.f;
}

__global__ __launch_bounds__(256) void add_inplace_kernel(float4* __restrict__ x, const float4* __restrict__ y, size_t n4,
                                                          float* __restrict__ xt, const float* __restrict__ yt, int tail) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) x[i] = f4_add(x[i], y[i]);
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) xt[threadIdx.x] += yt[threadIdx.x];
}

// ---- dgl.max_nodes / mean_nodes (Aggregator.py:58-61) ------------------------------------------
__global__ __launch_bounds__(256) void segment_pool_fwd_kernel(const float* __restrict__ h,
                                                               const int32_t* __restrict__ seg_ptr, int D,
                                                               int is_max, float* __restrict__ out,
                                                               int32_t* __restrict__ argmax) {
    __shared__ float rv[4][64];
    __shared__ int ri[4][64];
    const int g = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    const int r0 = seg_ptr[g], r1 = seg_ptr[g + 1];
    float best = is_max ? -INFINITY : 0.f;
    int bi = r0;
    if (c < D) {
        for (int r = r0 + wave; r < r1; r += 4) {
            const float v = h[(size_t)r * D + c];
            if (is_max) { if (v > best) { best = v; bi = r; } }
            else best += v;
        }
    }
    rv[wave][lane] = best; ri[wave][lane] = bi;
    __syncthreads();
    if (wave == 0 && c < D) {
        if (is_max) {
            // first maximum in row order (wave w holds rows r0+w, r0+w+4, ...)
            for (int w = 1; w < 4; ++w) {
                const float v = rv[w][lane];
                const int i = ri[w][lane];
                if (v > best || (v == best && i < bi)) { best = v; bi = i; }
            }
            out[(size_t)g * D + c] = best;
            argmax[(size_t)g * D + c] = bi;
        } else {
            const float s = (rv[0][lane] + rv[1][lane]) + (rv[2][lane] + rv[3][lane]);
            out[(size_t)g * D + c] = s / (float)max(r1 - r0, 1);
        }
    }
}

__global__ __launch_bounds__(256) void segment_pool_bwd_kernel(const float* __restrict__ dout,
                                                               const int32_t* __restrict__ seg_ptr,
                                                               const int32_t* __restrict__ argmax, int D,
                                                               int is_max, float* __restrict__ dh) {
    const int g = blockIdx.x;
    const int r0 = seg_ptr[g], r1 = seg_ptr[g + 1];
    const int n = (r1 - r0) * D;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int r = r0 + i / D, c = i % D;
        const float go = dout[(size_t)g * D + c];
        float v;
        if (is_max) v = (argmax[(size_t)g * D + c] == r) ? go : 0.f;
        else v = go / (float)(r1 - r0);
        dh[(size_t)r * D + c] = v;
    }
}

__global__ __launch_bounds__(256) void dropout_kernel(const float4* __restrict__ x, size_t n4, DropCfg d,
                                                      float4* __restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        y[i] = f4_mul(x[i], renet_drop4(d, i));
}

inline int grid_for(size_t total) { return (int)max((size_t)1, min((size_t)2048, (total + 255) / 256)); }

}  // namespace

extern "C" {

int renet_seq_assemble_fwd(const float* h2, const float* ent, const float* rel, const float* glob,
                           const int32_t* subj_row, const int32_t* row_ent, const int32_t* row_rel,
                           const int32_t* glob_row, int S, int D, float drop_p, uint64_t seed_x,
                           uint64_t seed_xr, float* X, float* Xr, void* stream) {
    if (S < 0 || D <= 0 || (D & 3) || drop_p < 0.f || drop_p >= 1.f) return RENET_ERR_BADARG;
    if (S == 0) return RENET_OK;
    const int CH = D / 4;
    RENET_LAUNCH(seq_assemble_fwd_kernel, dim3(grid_for((size_t)S * 4 * CH)), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)h2, (const float4*)ent, (const float4*)rel,
                       (const float4*)glob, subj_row, row_ent, row_rel, glob_row, S, CH,
                       make_drop(drop_p, seed_x), make_drop(drop_p, seed_xr), (float4*)X, (float4*)Xr,
                       (float*)nullptr, (float*)nullptr);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

int renet_bound_parts(size_t total4) { return (int)max((size_t)1, min((size_t)1024, (total4 + 255) / 256)); }

int renet_seq_assemble_fwd_bounds(const float* h2, const float* ent, const float* rel, const float* glob,
                                  const int32_t* subj_row, const int32_t* row_ent, const int32_t* row_rel,
                                  const int32_t* glob_row, int S, int D, float drop_p, uint64_t seed_x,
                                  uint64_t seed_xr, float* X, float* Xr, float* partX, float* partXr, void* stream) {
    if (S <= 0 || D <= 0 || (D & 3) || drop_p < 0.f || drop_p >= 1.f || !partX || !partXr) return RENET_ERR_BADARG;
    const int CH = D / 4;
    RENET_LAUNCH(seq_assemble_fwd_kernel, dim3(renet_bound_parts((size_t)S * 4 * CH)), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)h2, (const float4*)ent, (const float4*)rel,
                       (const float4*)glob, subj_row, row_ent, row_rel, glob_row, S, CH,
                       make_drop(drop_p, seed_x), make_drop(drop_p, seed_xr), (float4*)X, (float4*)Xr, partX, partXr);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

// zero the padding a bf16-storage GEMM may read: columns [C, ceil64(C)) of the first `rows` rows and rows
// [rows, ceil64(rows)) entirely
static int zero_bf16_padding(__bf16* P, int rows, int C, int ld, int rows_alloc, hipStream_t st) {
    const int c16 = min((C + 63) & ~63, ld), r16 = min((rows + 63) & ~63, rows_alloc);
    if (c16 > C && rows > 0) {
        hipError_t e = hipMemset2DAsync(P + C, (size_t)ld * sizeof(__bf16), 0, (size_t)(c16 - C) * sizeof(__bf16),
                                        (size_t)rows, st);
        if (e != hipSuccess) return (int)e;
    }
    if (r16 > rows) {
        hipError_t e = hipMemsetAsync(P + (size_t)rows * ld, 0, (size_t)(r16 - rows) * ld * sizeof(__bf16), st);
        if (e != hipSuccess) return (int)e;
    }
    return RENET_OK;
}

int renet_seq_assemble_fwd_bf16(const float* h2, const float* ent, const float* rel, const float* glob,
                                const int32_t* subj_row, const int32_t* row_ent, const int32_t* row_rel,
                                const int32_t* glob_row, int S, int D, float drop_p, uint64_t seed_x,
                                uint64_t seed_xr, void* X, int ldx, void* Xr, int ldxr, int rows_alloc,
                                void* stream) {
    if (S < 0 || D <= 0 || (D & 3) || drop_p < 0.f || drop_p >= 1.f || ldx < 4 * D || ldxr < 3 * D || (ldx & 3) ||
        (ldxr & 3) || rows_alloc < S) return RENET_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    int e = zero_bf16_padding((__bf16*)X, S, 4 * D, ldx, rows_alloc, st);
    if (e != RENET_OK) return e;
    e = zero_bf16_padding((__bf16*)Xr, S, 3 * D, ldxr, rows_alloc, st);
    if (e != RENET_OK) return e;
    if (S == 0) return RENET_OK;
    const int CH = D / 4;
    RENET_LAUNCH(seq_assemble_fwd_bf16_kernel, dim3(grid_for((size_t)S * 4 * CH)), dim3(256), 0, st,
                       (const float4*)h2, (const float4*)ent, (const float4*)rel, (const float4*)glob, subj_row,
                       row_ent, row_rel, glob_row, S, CH, make_drop(drop_p, seed_x), make_drop(drop_p, seed_xr),
                       (__bf16*)X, ldx, (__bf16*)Xr, ldxr);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

int renet_seq_assemble_bwd(const float* dX, const float* dXr, const int32_t* step_off, int L, int S,
                           int B, int D, float drop_p, uint64_t seed_x, uint64_t seed_xr, float* dRows,
                           float* dEntSeq, float* dRelSeq, void* stream) {
    if (S < 0 || B < 0 || L < 0 || D <= 0 || (D & 3) || drop_p < 0.f || drop_p >= 1.f) return RENET_ERR_BADARG;
    const int CH = D / 4;
    const DropCfg dx = make_drop(drop_p, seed_x), dxr = make_drop(drop_p, seed_xr);
    if (S > 0) {
        RENET_LAUNCH(seq_assemble_bwd_rows_kernel, dim3(grid_for((size_t)S * CH)), dim3(256), 0,
                           (hipStream_t)stream, (const float4*)dX, (const float4*)dXr, S, CH, dx, dxr,
                           (float4*)dRows);
        RENET_LAUNCH_CHECK();
    }
    if (B > 0) {
        RENET_LAUNCH(seq_assemble_bwd_seq_kernel, dim3(grid_for((size_t)B * CH)), dim3(256), 0,
                           (hipStream_t)stream, (const float4*)dX, (const float4*)dXr, step_off, L, B, CH, dx,
                           dxr, (float4*)dEntSeq, (float4*)dRelSeq);
        RENET_LAUNCH_CHECK();
    }
    return RENET_OK;
}

static int concat3_fwd_impl(const float* a, const int32_t* ia, const float* hmid, const float* c,
                            const int32_t* ic, int B, int D, float drop_p, uint64_t seed, float* feat,
                            float* bound_part, void* stream) {
    if (B < 0 || D <= 0 || (D & 3) || drop_p < 0.f || drop_p >= 1.f) return RENET_ERR_BADARG;
    if (B == 0) return bound_part ? RENET_ERR_BADARG : RENET_OK;
    const int CH = D / 4, parts = c ? 3 : 2;
    RENET_LAUNCH(concat3_fwd_kernel, dim3(bound_part ? renet_bound_parts((size_t)B * parts * CH)
                                                     : grid_for((size_t)B * parts * CH)), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)a, ia, (const float4*)hmid, (const float4*)c, ic,
                       B, CH, parts, make_drop(drop_p, seed), (float4*)feat, bound_part);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

int renet_concat3_fwd(const float* a, const int32_t* ia, const float* hmid, const float* c,
                      const int32_t* ic, int B, int D, float drop_p, uint64_t seed, float* feat,
                      void* stream) {
    return concat3_fwd_impl(a, ia, hmid, c, ic, B, D, drop_p, seed, feat, nullptr, stream);
}

int renet_concat3_fwd_bounds(const float* a, const int32_t* ia, const float* hmid, const float* c,
                             const int32_t* ic, int B, int D, float drop_p, uint64_t seed, float* feat,
                             float* bound_part, void* stream) {
    if (!bound_part) return RENET_ERR_BADARG;
    return concat3_fwd_impl(a, ia, hmid, c, ic, B, D, drop_p, seed, feat, bound_part, stream);
}

int renet_concat3_bwd(const float* dfeat, int B, int D, int parts, float drop_p, uint64_t seed,
                      float* da_rows, float* dhmid, float* dc_rows, void* stream) {
    if (B < 0 || D <= 0 || (D & 3) || (parts != 2 && parts != 3) || drop_p < 0.f || drop_p >= 1.f)
        return RENET_ERR_BADARG;
    if (B == 0) return RENET_OK;
    const int CH = D / 4;
    RENET_LAUNCH(concat3_bwd_kernel, dim3(grid_for((size_t)B * parts * CH)), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)dfeat, B, CH, parts, make_drop(drop_p, seed),
                       (float4*)da_rows, (float4*)dhmid, (float4*)dc_rows);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

int renet_dropout(const float* x, size_t n, float drop_p, uint64_t seed, float* y, void* stream) {
    if ((n & 3) || drop_p < 0.f || drop_p >= 1.f) return RENET_ERR_BADARG;
    if (n == 0) return RENET_OK;
    RENET_LAUNCH(dropout_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)x, n / 4, make_drop(drop_p, seed), (float4*)y);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

static int softmax_ce_impl(const float* logits, const int32_t* target, int B, int C, int ld, float grad_scale,
                           float* row_loss, float* dlogits, __bf16* dl16, int ld16, void* stream);

int renet_softmax_ce(const float* logits, const int32_t* target, int B, int C, int ld,
                     float grad_scale, float* row_loss, float* dlogits, void* stream) {
    return softmax_ce_impl(logits, target, B, C, ld, grad_scale, row_loss, dlogits, nullptr, 0, stream);
}

int renet_softmax_ce_bf16(const float* logits, const int32_t* target, int B, int C, int ld, float grad_scale,
                          float* row_loss, void* dlogits_bf16, int ld16, int rows16, void* stream) {
    if (!dlogits_bf16 || ld16 < C || (ld16 & 1) || rows16 < B) return RENET_ERR_BADARG;
    // rows [B, rows16) of the padded matrix: the k padding of dW = dlogits^T feat (contraction over the rows)
    if (rows16 > B) {
        hipError_t e = hipMemsetAsync((__bf16*)dlogits_bf16 + (size_t)B * ld16, 0,
                                      (size_t)(rows16 - B) * ld16 * sizeof(__bf16), (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    return softmax_ce_impl(logits, target, B, C, ld, grad_scale, row_loss, nullptr, (__bf16*)dlogits_bf16, ld16, stream);
}

static int softmax_ce_impl(const float* logits, const int32_t* target, int B, int C, int ld, float grad_scale,
                           float* row_loss, float* dlogits, __bf16* dl16, int ld16, void* stream) {
    if (B < 0 || C <= 0 || ld < C) return RENET_ERR_BADARG;
    if (B == 0) return RENET_OK;
    const size_t lds = (size_t)C * sizeof(float);
    static int reg_ok = -1;                 // RENET_SOFTMAX_REG=0: the LDS-staged kernel for every wide row (A/B runs)
    if (reg_ok < 0) {
        const char* e_ = getenv("RENET_SOFTMAX_REG");
        reg_ok = (e_ && e_[0] == '0') ? 0 : 1;
    }
    if (reg_ok && !dl16 && C >= 4096 && C <= 48 * 512) {
        const dim3 grid(B), blk(512);
        hipStream_t st = (hipStream_t)stream;
        if (C <= 16 * 512) RENET_LAUNCH((softmax_ce_reg_kernel<16>), grid, blk, 0, st, logits, target, C, ld, grad_scale, row_loss, dlogits);
        else if (C <= 32 * 512) RENET_LAUNCH((softmax_ce_reg_kernel<32>), grid, blk, 0, st, logits, target, C, ld, grad_scale, row_loss, dlogits);
        else RENET_LAUNCH((softmax_ce_reg_kernel<48>), grid, blk, 0, st, logits, target, C, ld, grad_scale, row_loss, dlogits);
    } else if (lds <= 128 * 1024 && C >= 4096) {
        static bool attr_set = false;      // benign race: the attribute is idempotent
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)softmax_ce_lds_kernel,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        RENET_LAUNCH(softmax_ce_lds_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, logits, target, C, ld,
                     grad_scale, row_loss, dlogits, dl16, ld16);
    } else {
        RENET_LAUNCH(softmax_ce_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, target, C, ld,
                     grad_scale, row_loss, dlogits, dl16, ld16);
    }
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

int renet_softmax_ce_planes(const float* logits, const int32_t* target, int B, int C, int ld, float grad_scale,
                            float* row_loss, void* dl_planes, size_t plane, int ld16, int rows16, void* stream) {
    if (B < 0 || C <= 0 || ld < C || !dl_planes || ld16 < C || (ld16 & 15) || rows16 < ((B + 15) & ~15)) return RENET_ERR_BADARG;
    if (plane < (size_t)rows16 * ld16) return RENET_ERR_BADARG;
    if (B == 0) return RENET_OK;
    hipStream_t st = (hipStream_t)stream;
    __bf16* P = (__bf16*)dl_planes;
    // rows [B, ceil16(B)): the k padding of dW = dl^T feat (contraction over the rows, 16 per half-stage) must be zero in
    // every plane: with T16 tiles that is the rest of tile row B >> 4 -- zeroed here, BEFORE the kernel writes its rows < B
    if ((B & 15) != 0) {
        for (int p = 0; p < 3; ++p) {
            hipError_t e = hipMemsetAsync(P + (size_t)p * plane + (size_t)(B >> 4) * ld16 * 16, 0,
                                          (size_t)ld16 * 16 * sizeof(__bf16), st);
            if (e != hipSuccess) return (int)e;
        }
    }
    const bool aligned = (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(dl_planes) & 7) == 0 && (plane & 3) == 0;
    static int rows4 = -1;                 // RENET_SOFTMAX_ROWS=1: one row per workgroup (the first planes writer), for A/B runs
    if (rows4 < 0) {
        const char* e_ = getenv("RENET_SOFTMAX_ROWS");
        rows4 = (e_ && e_[0] == '1') ? 0 : 1;
    }
    if (aligned && rows4 && C >= 2048 && ld16 <= 12 * 2048 && (reinterpret_cast<uintptr_t>(dl_planes) & 15) == 0 &&
        (plane & 7) == 0) {
        const int quads = (B + 3) / 4;
        const dim3 grid((unsigned)((quads + 31) & ~31)), blk(512);   // whole groups of 8 XCDs x 4 quads
        if (ld16 <= 4 * 2048) RENET_LAUNCH((softmax_ce_planes4_kernel<4>), grid, blk, 0, st, logits, target, B, C, ld, grad_scale, row_loss, P, plane, ld16);
        else if (ld16 <= 8 * 2048) RENET_LAUNCH((softmax_ce_planes4_kernel<8>), grid, blk, 0, st, logits, target, B, C, ld, grad_scale, row_loss, P, plane, ld16);
        else RENET_LAUNCH((softmax_ce_planes4_kernel<12>), grid, blk, 0, st, logits, target, B, C, ld, grad_scale, row_loss, P, plane, ld16);
    } else if (aligned && C >= 2048 && ld16 <= 12 * 2048) {
        const dim3 grid((unsigned)((B + 127) & ~127)), blk(512);     // whole groups of 8 XCDs x 16 rows (see the kernel)
        if (ld16 <= 4 * 2048) RENET_LAUNCH((softmax_ce_planes_kernel<4>), grid, blk, 0, st, logits, target, B, C, ld, grad_scale, row_loss, P, plane, ld16);
        else if (ld16 <= 8 * 2048) RENET_LAUNCH((softmax_ce_planes_kernel<8>), grid, blk, 0, st, logits, target, B, C, ld, grad_scale, row_loss, P, plane, ld16);
        else RENET_LAUNCH((softmax_ce_planes_kernel<12>), grid, blk, 0, st, logits, target, B, C, ld, grad_scale, row_loss, P, plane, ld16);
    } else {
        RENET_LAUNCH(softmax_ce_planes_generic_kernel, dim3(B), dim3(256), 0, st, logits, target, C, ld, grad_scale,
                     row_loss, P, plane, ld16);
    }
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

int renet_zero(float* x, size_t n, void* stream) {
    if (n == 0) return RENET_OK;
    if (!x || (reinterpret_cast<uintptr_t>(x) & 15)) return RENET_ERR_BADARG;
    const size_t n4 = n / 4;
    const int blocks = (int)max((size_t)1, min((size_t)2048, (n4 + 255) / 256));
    RENET_LAUNCH(zero_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float4*)x, n4, x + n4 * 4, (int)(n & 3));
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

int renet_add_inplace(float* x, const float* y, size_t n, void* stream) {
    if (n == 0) return RENET_OK;
    if (!x || !y || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15)) return RENET_ERR_BADARG;
    const size_t n4 = n / 4;
    const int blocks = (int)max((size_t)1, min((size_t)2048, (n4 + 255) / 256));
    RENET_LAUNCH(add_inplace_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float4*)x, (const float4*)y, n4,
                 x + n4 * 4, y + n4 * 4, (int)(n & 3));
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

int renet_segment_pool_fwd(const float* h, const int32_t* seg_ptr, int G, int D, int is_max,
                           float* out, int32_t* argmax, void* stream) {
    if (G < 0 || D <= 0) return RENET_ERR_BADARG;
    if (G == 0) return RENET_OK;
    RENET_LAUNCH(segment_pool_fwd_kernel, dim3(G, (D + 63) / 64), dim3(256), 0, (hipStream_t)stream, h,
                       seg_ptr, D, is_max, out, argmax);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

int renet_segment_pool_bwd(const float* dout, const int32_t* seg_ptr, const int32_t* argmax, int G,
                           int D, int is_max, int N, float* dh, void* stream) {
    (void)N;
    if (G < 0 || D <= 0) return RENET_ERR_BADARG;
    if (G == 0) return RENET_OK;
    RENET_LAUNCH(segment_pool_bwd_kernel, dim3(G), dim3(256), 0, (hipStream_t)stream, dout, seg_ptr,
                       argmax, D, is_max, dh);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

}  // extern "C"
