// Fused gradient clipping + Adam over ONE flat parameter / gradient / moment buffer.
// Replaces train.py:140-142 (clip_grad_norm_, Adam.step with weight_decay, zero_grad), i.e. torch's
// multi-tensor norm + ~8 multi_tensor_apply kernels + the 81 MB gradient memset, by two launches:
//   1. per-workgroup partial sums of g^2 (fixed order => deterministic norm);
//   2. every workgroup re-reduces the (<= 2048) partials, derives the clip coefficient, updates its slice
//      (torch.optim.Adam semantics: L2 weight decay added to the clipped gradient, bias-corrected moments)
//      and zeroes the gradient it just consumed.
// HBM-bound: 28 B of traffic per parameter.
#include "common.h"

namespace {

constexpr int kBlocks = 2048;

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float4* __restrict__ g, size_t n4,
                                                            const float* __restrict__ g_tail, int tail,
                                                            float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = g[i];
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) s += g_tail[threadIdx.x] * g_tail[threadIdx.x];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

struct AdamCfg {
    float lr, beta1, beta2, eps, wd, max_norm, bc1, bc2_sqrt;
    float grad_scale;          // the gradient is g * grad_scale (1 / world_size after a SUM all-reduce)
    int zero_grad;
};

__device__ __forceinline__ float adam_one(float& p, float g, float& m, float& v, const AdamCfg& c, float coef) {
    g = g * coef + c.wd * p;
    m = c.beta1 * m + (1.f - c.beta1) * g;
    v = c.beta2 * v + (1.f - c.beta2) * g * g;
    const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
    p -= (c.lr / c.bc1) * (m / denom);
    return 0.f;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, size_t n,
                                                   const float* __restrict__ partial, int n_partial, AdamCfg c,
                                                   float* __restrict__ norm_out) {
    __shared__ float red[4];
    __shared__ float s_coef;
    float s = 0.f;
    for (int i = threadIdx.x; i < n_partial; i += 256) s += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = c.grad_scale * sqrtf((red[0] + red[1]) + (red[2] + red[3]));
        float coef = 1.f;
        if (c.max_norm > 0.f) coef = fminf(c.max_norm / (norm + 1e-6f), 1.f);      // clip_grad_norm_
        s_coef = coef * c.grad_scale;
        if (blockIdx.x == 0 && norm_out) *norm_out = norm;
    }
    __syncthreads();
    const float coef = s_coef;
    const size_t n4 = n / 4;
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* g4 = reinterpret_cast<float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
        adam_one(pp.x, gg.x, mm.x, vv.x, c, coef);
        adam_one(pp.y, gg.y, mm.y, vv.y, c, coef);
        adam_one(pp.z, gg.z, mm.z, vv.z, c, coef);
        adam_one(pp.w, gg.w, mm.w, vv.w, c, coef);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
        if (c.zero_grad) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        float pp = p[i], mm = m[i], vv = v[i];
        adam_one(pp, g[i], mm, vv, c, coef);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (c.zero_grad) g[i] = 0.f;
    }
}

}  // namespace

extern "C" {

size_t renet_adam_workspace(size_t n) {
    (void)n;
    return kBlocks * sizeof(float);
}

int renet_adam_step(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, float max_norm, int step, int zero_grad, float* workspace,
                    size_t workspace_bytes, float* grad_norm_out, void* stream) {
    return renet_adam_step_scaled(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, max_norm, 1.f, step, zero_grad,
                                  workspace, workspace_bytes, grad_norm_out, stream);
}

int renet_sumsq_partials(const float* g, size_t n, float* partial, int n_slots, void* stream) {
    if (!g || !partial || n_slots < 1 || n_slots > kBlocks) return RENET_ERR_BADARG;
    if (reinterpret_cast<uintptr_t>(g) & 15) return RENET_ERR_BADARG;
    const size_t n4 = n / 4;
    // exactly n_slots workgroups: the ones without elements write 0 (every slot of the region is defined)
    RENET_LAUNCH(sumsq_partial_kernel, dim3(n_slots), dim3(256), 0, (hipStream_t)stream, (const float4*)g, n4, g + n4 * 4,
                       (int)(n & 3), partial);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

static int adam_launch(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                       float weight_decay, float max_norm, float grad_scale, int step, int zero_grad, const float* partial,
                       int n_partial, float* grad_norm_out, hipStream_t st) {
    const size_t n4 = n / 4;
    const int blocks = (int)max((size_t)1, min((size_t)kBlocks, (n4 + 255) / 256));
    AdamCfg c;
    c.lr = lr; c.beta1 = beta1; c.beta2 = beta2; c.eps = eps; c.wd = weight_decay; c.max_norm = max_norm;
    c.bc1 = 1.f - powf(beta1, (float)step);
    c.bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    c.zero_grad = zero_grad;
    c.grad_scale = grad_scale;
    RENET_LAUNCH(adam_kernel, dim3(blocks), dim3(256), 0, st, p, g, m, v, n, partial, n_partial, c, grad_norm_out);
    RENET_LAUNCH_CHECK();
    return RENET_OK;
}

int renet_adam_step_presummed(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, float max_norm, float grad_scale, int step, int zero_grad,
                              const float* partial, int n_partial, float* grad_norm_out, void* stream) {
    if (step < 1 || lr < 0.f || beta1 < 0.f || beta1 >= 1.f || beta2 < 0.f || beta2 >= 1.f) return RENET_ERR_BADARG;
    if (n == 0) return RENET_OK;
    if (!partial || n_partial < 1 || n_partial > kBlocks) return RENET_ERR_BADARG;
    if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
         reinterpret_cast<uintptr_t>(v)) & 15) return RENET_ERR_BADARG;
    return adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, max_norm, grad_scale, step, zero_grad, partial,
                       n_partial, grad_norm_out, (hipStream_t)stream);
}

int renet_adam_step_scaled(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                           float eps, float weight_decay, float max_norm, float grad_scale, int step, int zero_grad,
                           float* workspace, size_t workspace_bytes, float* grad_norm_out, void* stream) {
    if (step < 1 || lr < 0.f || beta1 < 0.f || beta1 >= 1.f || beta2 < 0.f || beta2 >= 1.f) return RENET_ERR_BADARG;
    if (n == 0) return RENET_OK;
    if (workspace_bytes < renet_adam_workspace(n)) return RENET_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
         reinterpret_cast<uintptr_t>(v)) & 15) return RENET_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const size_t n4 = n / 4;
    const int blocks = (int)max((size_t)1, min((size_t)kBlocks, (n4 + 255) / 256));
    RENET_LAUNCH(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)g, n4, g + n4 * 4,
                       (int)(n & 3), workspace);
    RENET_LAUNCH_CHECK();
    return adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, max_norm, grad_scale, step, zero_grad, workspace,
                       blocks, grad_norm_out, st);
}


}  // extern "C"
