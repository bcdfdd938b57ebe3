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

// sin / cos of 30*u on the hardware transcendental unit (v_sin_f32 / v_cos_f32 take revolutions):
// t = u * 30/(2 pi) as a two-float product so the fractional part keeps 24 good bits whatever |u|, then
// v_fract + v_sin.  Measured on MI355X against fp64 over |u| < 40: max abs error 3.8e-7 (the polynomial above:
// 7e-8) at ~6 instructions instead of ~22.  Used for the SIREN / FiLM activations sin(30 u)
// (nerf/nerf.py:112, pi_GAN/modules.py:25); the positional encoding keeps the polynomial.
__device__ __forceinline__ float hw_frac30(float u) {
    const float c_hi = 4.77464829275686f;                                       // 30 / (2 pi)
    const float c_lo = (float)(4.774648292756860073 - (double)4.77464829275686f);
    const float hi = u * c_hi;
    const float lo = fmaf(u, c_hi, -hi) + u * c_lo;
    return __builtin_amdgcn_fractf(hi) + lo;
}
__device__ __forceinline__ float hw_sin30(float u) { return __builtin_amdgcn_sinf(hw_frac30(u)); }
__device__ __forceinline__ SinCos hw_sincos30(float u) {
    const float f = hw_frac30(u);
    return {__builtin_amdgcn_sinf(f), __builtin_amdgcn_cosf(f)};
}

}  // namespace mi
