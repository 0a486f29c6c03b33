// Shared device/host helpers for librenet_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/renet_hip.h"

// hipGetLastError() reports the last error of ANY earlier runtime call on this thread -- including calls made
// by the host framework that are expected to fail and are never cleared (PyTorch probing a host pointer with
// hipPointerGetAttributes leaves hipErrorInvalidValue behind).  Every launch therefore clears the slot first,
// so that RENET_LAUNCH_CHECK() sees the status of OUR launch only.
#define RENET_LAUNCH(...)                         \
    do {                                          \
        (void)hipGetLastError();                  \
        hipLaunchKernelGGL(__VA_ARGS__);          \
    } while (0)

#define RENET_LAUNCH_CHECK()                      \
    do {                                          \
        hipError_t e__ = hipGetLastError();       \
        if (e__ != hipSuccess) return (int)e__;   \
    } while (0)

static inline bool renet_dim_ok(int D) { return D == 100 || D == 200 || D == 400; }

// ---- counter-based dropout ------------------------------------------------------------------
// One splitmix64 draw per group of 4 consecutive elements (every dropout site works on float4
// groups); element j of group g is kept iff bits [16j, 16j+16) of the draw are >= p * 65536.
// The same (seed, g) regenerates the mask in the backward pass: no mask tensor is stored.
__device__ __forceinline__ uint64_t renet_mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

struct DropCfg {
    uint32_t thresh;   // keep iff draw16 >= thresh ; 0 => keep everything
    float scale;       // 1 / (1 - p)
    uint64_t seed;
};

static inline DropCfg make_drop(float p, uint64_t seed) {
    DropCfg d;
    if (p <= 0.f) { d.thresh = 0; d.scale = 1.f; }
    else { d.thresh = (uint32_t)(p * 65536.f + 0.5f); d.scale = 1.f / (1.f - p); }
    d.seed = seed;
    return d;
}

// returns the 4 multipliers (0 or scale) of group g
__device__ __forceinline__ float4 renet_drop4(const DropCfg& d, uint64_t g) {
    if (d.thresh == 0) return make_float4(1.f, 1.f, 1.f, 1.f);
    uint64_t r = renet_mix64(d.seed * 0xD1342543DE82EF95ull + g);
    float4 m;
    m.x = ((uint32_t)(r) & 0xFFFFu) >= d.thresh ? d.scale : 0.f;
    m.y = ((uint32_t)(r >> 16) & 0xFFFFu) >= d.thresh ? d.scale : 0.f;
    m.z = ((uint32_t)(r >> 32) & 0xFFFFu) >= d.thresh ? d.scale : 0.f;
    m.w = ((uint32_t)(r >> 48) & 0xFFFFu) >= d.thresh ? d.scale : 0.f;
    return m;
}

__device__ __forceinline__ float4 f4_mul(float4 a, float4 b) {
    return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 f4_scale(float4 a, float s) {
    return make_float4(a.x * s, a.y * s, a.z * s, a.w * s);
}

// XCD-aware virtual block id: the dispatcher places block b on XCD b % 8 (observed, speed only);
// remapping gives every XCD one contiguous slice of the row space so that neighbouring rows
// (same member graph => shared source rows) hit the same per-XCD L2.
__device__ __forceinline__ int renet_xcd_block(int b, int nb) {
    return (nb & 7) == 0 ? (b & 7) * (nb >> 3) + (b >> 3) : b;
}

// ---- T16: the tiled storage format of bf16 operand PLANES (csrc/gemm_p6.h; tools/p6_layout_sim.py is the executable
// specification).  Element offset of (row, col) inside one plane of a padded matrix with `tc_count` = Cp / 16 tiles per
// tile row: 16 x 16 tiles of 512 bytes, row i of a tile at i ^ 4 when the tile column is odd, the two 16-byte halves of a
// tile row swapped when (i >> 3) & 1.  Four consecutive columns starting at a multiple of 4 are consecutive elements.
__host__ __device__ __forceinline__ size_t renet_t16_off(int row, int col, int tc_count) {
    const int i = row & 15, j = col & 15, tc = col >> 4;
    const int ip = i ^ ((tc & 1) << 2);
    const int hh = (j >> 3) ^ ((i >> 3) & 1);
    return ((size_t)(row >> 4) * tc_count + tc) * 256 + ip * 16 + hh * 8 + (j & 7);
}
