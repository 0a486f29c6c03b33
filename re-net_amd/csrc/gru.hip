// GRU recurrence over the packed <= seq_len history window (torch.nn.GRU semantics: 1 layer,
// h0 = 0, gate order r,z,n; reference model.py:28-29,86,94 and global_model.py:25,49).
//
// The input projection Gi = X W_ih^T + b_ih is one large GEMM done by the caller; here each step j
// runs   Gh = H[0:bs_j] W_hh^T + b_hh   (MFMA GEMM, bs_j = sequences still alive at step j; the
// batch is sorted by length so the live sequences are a prefix) followed by ONE fused gate kernel that
// produces h_j in place and stashes (r, z, n, W_hn h + b_hn, h_prev) for the backward pass.
// step_off lives on the host: the per-step loop is enqueued from C, not from Python.
#include "common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

__global__ __launch_bounds__(256) void gru_gate_fwd_kernel(const float* __restrict__ Gi,   // [bs,3H] rows of this step
                                                           const float* __restrict__ Gh,   // [bs,3H]
                                                           float* __restrict__ Hcur,       // [bs,H] in/out
                                                           float* __restrict__ saved,      // [bs,5H]
                                                           int bs, int H) {
    const int total = bs * H;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / H, c = i - b * H;
        const float* gi = Gi + (size_t)b * 3 * H;
        const float* gh = Gh + (size_t)b * 3 * H;
        const float r = sigmoidf_(gi[c] + gh[c]);
        const float z = sigmoidf_(gi[H + c] + gh[H + c]);
        const float hn = gh[2 * H + c];
        const float n = tanhf(gi[2 * H + c] + r * hn);
        const float hp = Hcur[i];
        const float h = (1.f - z) * n + z * hp;
        float* sv = saved + (size_t)b * 5 * H;
        sv[c] = r; sv[H + c] = z; sv[2 * H + c] = n; sv[3 * H + c] = hn; sv[4 * H + c] = hp;
        Hcur[i] = h;
    }
}

__global__ __launch_bounds__(256) void gru_gate_bwd_kernel(const float* __restrict__ saved,  // [bs,5H]
                                                           float* __restrict__ dh,           // [bs,H] in/out
                                                           float* __restrict__ dGi,          // [bs,3H]
                                                           float* __restrict__ dGh,          // [bs,3H]
                                                           int bs, int H) {
    const int total = bs * H;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / H, c = i - b * H;
        const float* sv = saved + (size_t)b * 5 * H;
        const float r = sv[c], z = sv[H + c], n = sv[2 * H + c], hn = sv[3 * H + c], hp = sv[4 * H + c];
        const float g = dh[i];
        const float dn = g * (1.f - z);
        const float dz = g * (hp - n);
        const float dan = dn * (1.f - n * n);
        const float daz = dz * z * (1.f - z);
        const float dar = dan * hn * r * (1.f - r);
        float* gi = dGi + (size_t)b * 3 * H;
        float* gh = dGh + (size_t)b * 3 * H;
        gi[c] = dar; gi[H + c] = daz; gi[2 * H + c] = dan;
        gh[c] = dar; gh[H + c] = daz; gh[2 * H + c] = dan * r;
        dh[i] = g * z;                       // direct path h_prev -> h ; the W_hh path is added by the GEMM
    }
}

inline int grid_for(int total) { return max(1, min(2048, (total + 255) / 256)); }

}  // namespace

extern "C" {

size_t renet_gru_workspace(int B, int H) {
    return ((size_t)B * 3 * H + (size_t)B * H) * sizeof(float);
}

int renet_gru_fwd(const float* Gi, const int32_t* step_off, int L, int H, const float* Whh,
                  const float* bhh, float* h_last, float* saved, float* workspace,
                  size_t workspace_bytes, void* stream) {
    if (L < 0 || H <= 0 || !step_off) return RENET_ERR_BADARG;
    if (L == 0) return RENET_OK;
    const int B = step_off[1] - step_off[0];
    if (B <= 0) return RENET_OK;
    if (workspace_bytes < renet_gru_workspace(B, H)) return RENET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* Gh = workspace;
    hipError_t e = hipMemsetAsync(h_last, 0, (size_t)B * H * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    for (int j = 0; j < L; ++j) {
        const int p0 = step_off[j], bs = step_off[j + 1] - step_off[j];
        if (bs <= 0) break;
        if (bs > B) return RENET_ERR_BADARG;      // batch sizes must be non-increasing
        int rc = renet_gemm_f32(0, 1, bs, 3 * H, H, 1.f, h_last, H, Whh, H, 0.f, Gh, 3 * H, bhh, 1,
                                nullptr, 0, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(gru_gate_fwd_kernel, dim3(grid_for(bs * H)), dim3(256), 0, st,
                           Gi + (size_t)p0 * 3 * H, Gh, h_last, saved + (size_t)p0 * 5 * H, bs, H);
        RENET_LAUNCH_CHECK();
    }
    return RENET_OK;
}

int renet_gru_bwd(const float* dh_last, const int32_t* step_off, int L, int H, const float* Whh,
                  const float* saved, float* dGi, float* dGh, float* workspace,
                  size_t workspace_bytes, void* stream) {
    if (L < 0 || H <= 0 || !step_off) return RENET_ERR_BADARG;
    if (L == 0) return RENET_OK;
    const int B = step_off[1] - step_off[0];
    if (B <= 0) return RENET_OK;
    if (workspace_bytes < renet_gru_workspace(B, H)) return RENET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* dh = workspace + (size_t)B * 3 * H;
    hipError_t e = hipMemcpyAsync(dh, dh_last, (size_t)B * H * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
    for (int j = L - 1; j >= 0; --j) {
        const int p0 = step_off[j], bs = step_off[j + 1] - step_off[j];
        if (bs <= 0) continue;
        hipLaunchKernelGGL(gru_gate_bwd_kernel, dim3(grid_for(bs * H)), dim3(256), 0, st,
                           saved + (size_t)p0 * 5 * H, dh, dGi + (size_t)p0 * 3 * H,
                           dGh + (size_t)p0 * 3 * H, bs, H);
        RENET_LAUNCH_CHECK();
        if (j > 0) {   // dh_prev += dGh W_hh   (h_prev of step 0 is the constant h0 = 0)
            int rc = renet_gemm_f32(0, 0, bs, H, 3 * H, 1.f, dGh + (size_t)p0 * 3 * H, 3 * H, Whh, H, 1.f,
                                    dh, H, nullptr, 1, nullptr, 0, stream);
            if (rc) return rc;
        }
    }
    return RENET_OK;
}

}  // extern "C"
