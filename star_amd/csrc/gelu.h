// Exact-erf GELU (torch.nn.GELU, timm Mlp act) and its derivative for kernel epilogues.
#pragma once
#include "common.h"

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. float32 round-off level) on the v_rcp / v_exp units:
// about a dozen VALU instructions instead of libm's ~50, which matters because the GELU epilogue of a
// 128x128 tile with K = 384 would otherwise cost more cycles than the tile's MFMAs.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
    float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    const float r = 1.0f - p * t * e;
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f)); }
// gelu(u) and gelu'(u) together: the derivative reuses Phi(u) and exp(-u^2/2) of the erf evaluation (3 extra
// instructions), so the backward pass multiplies by a stored factor instead of evaluating erf again
__device__ __forceinline__ void gelu_both(float u, float& v, float& dv) {
    const float z = fabsf(u) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);   // exp(-u^2 / 2)
    const float hy = 0.5f * p * t * e;                                      // erfc(|u| / sqrt 2) / 2
    const float phi = u >= 0.f ? 1.0f - hy : hy;
    v = u * phi;
    dv = __builtin_fmaf(u * 0.3989422804014327f, e, phi);
}
__device__ __forceinline__ float dgelu_f(float x) {
    return 0.5f * (1.0f + erf_as(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);
}
