// mi_math.h - branch-free fp32 sin / sincos for the fused field-MLP epilogues (gfx950).
//
// The SIREN / FiLM activations sin(30*v) (nerf/nerf.py:112, pi_GAN/modules.py:25) and the
// positional encoding sin/cos(2^i x) (nerf/nerf.py:47-48) run once per register of a 256-wide
// layer, i.e. 128 times per lane per layer.  libm's sinf drags its Payne-Hanek slow path (and a
// private-memory array) into every call site; this version is ~20 VALU ops, no branches, no
// scratch: 3-constant Cody-Waite reduction by pi/2 with FMA, then the classic degree-7 / degree-8
// minimax kernels on [-pi/4, pi/4].  Max error vs float64 on |x| <= 1e5 is < 2.5e-7 absolute
// (measured by tests/test_gpu_math.py); beyond 2^23 quadrants the reduction loses exactness.
#pragma once
#include <hip/hip_runtime.h>

namespace mi {

struct SinCos { float s, c; };

__device__ __forceinline__ SinCos fast_sincos(float x) {
    const float k = rintf(x * 0.636619772367581343f);                 // x * 2/pi
    float r = fmaf(k, -1.57079637050628662109375f, x);                // pi/2 split in three floats
    r = fmaf(k, 4.37113900018624283e-8f, r);
    r = fmaf(k, 1.71512449044e-15f, r);
    const float s2 = r * r;
    float ps = fmaf(s2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(ps, s2, -1.6666654611e-1f);
    const float sn = fmaf(ps * s2, r, r);
    float pc = fmaf(s2, 2.44331571e-5f, -1.38873163e-3f);
    pc = fmaf(pc, s2, 4.16666457e-2f);
    pc = fmaf(pc, s2, -0.5f);
    const float cs = fmaf(pc, s2, 1.0f);
    const int q = (int)k;
    const bool swap = q & 1;
    float so = swap ? cs : sn;
    float co = swap ? sn : cs;
    so = (q & 2) ? -so : so;
    co = ((q + 1) & 2) ? -co : co;
    return {so, co};
}

__device__ __forceinline__ float fast_sin(float x) { return fast_sincos(x).s; }

// sin / cos of 30*u on the hardware transcendental unit (v_sin_f32 / v_cos_f32 take revolutions), for the SIREN /
// FiLM activations sin(30 u) (nerf/nerf.py:112, pi_GAN/modules.py:25).
//  * t = fl(30 u) FIRST: the reference evaluates torch.sin(30 * x), i.e. the sine of the ROUNDED product; at
//    |30 u| ~ 100 (first layers on raw coordinates) that rounding moves the sine by up to 4e-6, so reducing the exact
//    product instead (as the first version of this file did) is closer to real arithmetic but 10x further from
//    the reference than the transcendental unit's own error;
//  * revolutions = t / (2 pi) as a two-float product, reduced to [-1/2, 1/2] (hi - rndne(hi) is exact, the low
//    part is added to a value whose ulp is <= 3e-8) before v_sin.
// Measured on MI355X (tools/probes/sin_variants.hip) against sin(fl(30 u)) in fp64 over |u| < 40: max abs error
// 1.8e-7, rms 4.2e-8 (v_fract instead of the centred reduction: 4.2e-7 / 6.9e-8; libm sinf: 6.9e-8 / 1.8e-8), at
// 8 instructions instead of libm's ~40 with a Payne-Hanek slow path at every call site.
// hipcc contracts a*b-c into an fma ACROSS statements by default, which would replace the rounded `hi` below by the
// exact product and count its low part twice: contraction is off inside these helpers.
__device__ __forceinline__ float hw_turns(float t) {
#pragma clang fp contract(off)
    const float c_hi = 0.15915494309189535f;                                    // 1 / (2 pi)
    const float c_lo = (float)(0.15915494309189533577 - (double)0.15915494309189535f);
    const float hi = t * c_hi;
    const float lo = fmaf(t, c_lo, fmaf(t, c_hi, -hi));
    return (hi - rintf(hi)) + lo;
}
__device__ __forceinline__ float hw_turns30(float u) {
#pragma clang fp contract(off)
    return hw_turns(30.f * u);
}
__device__ __forceinline__ float hw_sin30(float u) { return __builtin_amdgcn_sinf(hw_turns30(u)); }
__device__ __forceinline__ SinCos hw_sincos30(float u) {
    const float f = hw_turns30(u);
    return {__builtin_amdgcn_sinf(f), __builtin_amdgcn_cosf(f)};
}

}  // namespace mi
