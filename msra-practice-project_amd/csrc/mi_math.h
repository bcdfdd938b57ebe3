// mi_math.h - fp32 sin for the fused field-MLP epilogues (gfx950): the SIREN / FiLM activations sin(30 v)
// (nerf/nerf.py:112, pi_GAN/modules.py:25) and the positional encoding sin / cos(2^i x) (nerf/nerf.py:47-48) run once per
// register of a 256-wide layer, i.e. 128 times per lane per layer.  libm's sinf drags its Payne-Hanek slow path (and a
// private-memory array) into every call site; here: a two-float range reduction in revolutions + v_sin_f32, branch-free.
#pragma once
#include <hip/hip_runtime.h>

namespace mi {

// sin / cos of 30*u on the hardware transcendental unit (v_sin_f32 / v_cos_f32 take revolutions), for the SIREN /
// FiLM activations sin(30 u) (nerf/nerf.py:112, pi_GAN/modules.py:25).
//  * t = fl(30 u) FIRST: the reference evaluates torch.sin(30 * x), i.e. the sine of the ROUNDED product; at
//    |30 u| ~ 100 (first layers on raw coordinates) that rounding moves the sine by up to 4e-6, so reducing the exact
//    product instead (as the first version of this file did) is closer to real arithmetic but 10x further from
//    the reference than the transcendental unit's own error;
//  * revolutions = t / (2 pi) with 1 / (2 pi) as a two-float constant, reduced to [-1/2, 1/2] before v_sin:
//    n = rndne(fl(t c_hi)), r = fma(t, c_lo, fma(t, c_hi, -n)) - four instructions.
// Measured on MI355X (tools/probes/sin_variants.hip) against sin(fl(30 u)) in fp64 over |u| < 40: max abs error
// 2.5e-7, rms 4.9e-8 (the six-instruction form hi = fl(t c_hi), lo = fma(t, c_lo, fma(t, c_hi, -hi)),
// r = (hi - rndne(hi)) + lo that this replaced: 1.8e-7 / 4.2e-8; v_fract instead of the centred reduction:
// 4.2e-7 / 6.9e-8; libm sinf: 6.9e-8 / 1.8e-8): two VALU instructions per activation are 1.5 % of a sin layer's time
// (every instruction between two MFMAs costs its issue cycles), the 7e-9 of rms error is 1/20 of what the layer's
// own 256-term fp32 dot product contributes.
// hipcc contracts a*b-c into an fma ACROSS statements by default, which would replace the rounded `hi` below by the
// exact product and count its low part twice: contraction is off inside these helpers.
__device__ __forceinline__ float hw_turns_fast(float t) {
#pragma clang fp contract(off)
    const float c_hi = 0.15915494309189535f;                                    // 1 / (2 pi)
    const float c_lo = (float)(0.15915494309189533577 - (double)0.15915494309189535f);
    // n = the nearest whole revolution of the ROUNDED product; both fmas then run against n: t c_hi - n with one
    // rounding (the difference is small, so that rounding is at the result's own ulp), plus the constant's low part
    const float n = rintf(t * c_hi);
    return fmaf(t, c_lo, fmaf(t, c_hi, -n));
}
// The six-instruction form (1.8e-7 / 4.2e-8): kept for the positional encoding, which runs 84 times per POINT (not per
// layer) and feeds an error-amplifying network - the NeRF field's closest parity case sits at 0.96 of its 1e-4 gate.
__device__ __forceinline__ float hw_turns(float t) {
#pragma clang fp contract(off)
    const float c_hi = 0.15915494309189535f;
    const float c_lo = (float)(0.15915494309189533577 - (double)0.15915494309189535f);
    const float hi = t * c_hi;
    const float lo = fmaf(t, c_lo, fmaf(t, c_hi, -hi));
    return (hi - rintf(hi)) + lo;
}
// The range reduction of the MFMA layers' epilogues (hw_turns30_x4 below), priced in round 3 on MI355X against the parity
// records (tools/sin_variants_round.sh, profiles/r03_sin_variants.log; VALU instructions per element before the v_sin):
//   0  t = fl(30 u); n = rndne(fl(t c_hi)); r = fma(t, c_lo, fma(t, c_hi, -n))    5   SirenNeRF 88.3 %  FiLM 86.3 % of peak
//   1  t = fl(30 u); n = rndne(fl(t c_hi)); r = fma(t, c_hi, -n)   (no c_lo)      4             88.8         86.9
//   2  t = fl(30 u); r = fract(fl(t c_hi))                                        3             89.3         87.5
//   3  r = fract(fl(u K)), K = fl(30 / 2 pi)                                      2             90.2         88.3
// Variant 0 follows the reference's fp32 expression torch.sin(30 * x) to 2.5e-7 (rms 5e-8) whatever |u|: it forms the
// ROUNDED product fl(30 u) like the reference and reduces it exactly (the c_lo term removes the 4e-8 relative error of
// fl(1 / 2 pi), which is a systematic scaling of every layer's frequency).  Every cheaper form either skips that
// rounding or rounds the revolutions once more: rms 2.4e-7 .. 2.8e-7 already for |u| < 0.5 (five times variant 0's), and
// growing with |u|.  On the committed fixtures that is almost invisible (worst SIREN / FiLM record on rgb / acc / depth
// 0.034 of the flat 1e-4 gate against 0.012), but a sin stack amplifies what its activations carry, point by point: with
// variant 3 in the product the x50-head sigma of one 128-point case left its bound - 1.07e-4 from the fp64 evaluation
// where the fp32 oracle sits at 4.5e-5 (with variant 0: 2.4e-5), i.e. further from exact arithmetic than 2x the
// reference's own fp32 path, which is the one thing the parity rules of this repository do not allow.  Parity is the
// first gate: the product keeps variant 0 and pays 1.9 / 2.0 points of MFMA time for it; -DMI_SIN_VARIANT=k builds the
// others (tools/diag_build.sh sin<k>).
#ifndef MI_SIN_VARIANT
#define MI_SIN_VARIANT 0
#endif
// w0: the layer's frequency - the literal 30 of nerf/nerf.py:112 (Siren hard-codes it) or FilmSiren's constructor argument
// (pi_GAN/modules.py:11,73), which the FiLM kernels read from the packed stream's trailer (field_layout.h:kTrailer).
// With the literal the code is what it was: t = fl(w0 u) is formed first, like torch.sin(w_0 * x).
__device__ __forceinline__ float hw_turns_w(float u, float w0) {
#pragma clang fp contract(off)
    return hw_turns_fast(w0 * u);
}
__device__ __forceinline__ float hw_turns30(float u) { return hw_turns_w(u, 30.f); }
#if defined(MI_DIAG_SIN) && MI_DIAG_SIN == 1     // diagnostic builds only (tools/diag_build.sh): no activation work at all
__device__ __forceinline__ float hw_sin_w(float u, float w0) { return u; }
#elif defined(MI_DIAG_SIN) && MI_DIAG_SIN == 2   // ... the transcendental alone, no range reduction
__device__ __forceinline__ float hw_sin_w(float u, float w0) { return __builtin_amdgcn_sinf(u); }
#elif defined(MI_DIAG_SIN) && MI_DIAG_SIN == 3   // ... the reduction alone, no transcendental
__device__ __forceinline__ float hw_sin_w(float u, float w0) { return hw_turns_w(u, w0); }
#else
__device__ __forceinline__ float hw_sin_w(float u, float w0) { return __builtin_amdgcn_sinf(hw_turns_w(u, w0)); }
#endif
__device__ __forceinline__ float hw_sin30(float u) { return hw_sin_w(u, 30.f); }

// Two elements at a time on the packed fp32 VALU ops (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32, full rate on
// gfx90a+): the wave runs ONE instruction stream per SIMD, and measured (tools/probes/mfma_valu_overlap.hip) every
// VALU instruction issued between two MFMAs ADDS ~3 cycles to the 64 of the MFMA - there is no second wave on the
// SIMD whose matrix work could cover it - so the activation's cost is its instruction count, and packing halves the
// seven arithmetic instructions of the range reduction (v_rndne and v_sin have no packed form).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 hw_turns_x2(f32x2 t) {
#pragma clang fp contract(off)
    const f32x2 c_hi = {0.15915494309189535f, 0.15915494309189535f};
    const float cl = (float)(0.15915494309189533577 - (double)0.15915494309189535f);
    const f32x2 c_lo = {cl, cl};
    const f32x2 hi = t * c_hi;
    const f32x2 n = {rintf(hi.x), rintf(hi.y)};
    return __builtin_elementwise_fma(t, c_lo, __builtin_elementwise_fma(t, c_hi, -n));
}
__device__ __forceinline__ f32x2 hw_turns30_x2(f32x2 u) {
#pragma clang fp contract(off)
    const f32x2 w0 = {30.f, 30.f};
    return hw_turns_x2(u * w0);
}
__device__ __forceinline__ f32x2 hw_sin30_x2(f32x2 u) {
    const f32x2 r = hw_turns30_x2(u);
    return f32x2{__builtin_amdgcn_sinf(r.x), __builtin_amdgcn_sinf(r.y)};
}

// Training: what the backward needs of a sin layer is its output X = sin(30 u) (the next layer's dW operand) and the
// derivative factor C = 30 cos(30 u).  Only X is kept, with the SIGN of the cosine in its lowest mantissa bit
// (the reduced angle r is at hand: cos < 0 iff |r| > 1/4 turn, see cos_sign_into), and the backward rebuilds
// C = +-30 sqrt(1 - X^2).  Halves the saved bytes per point (18.5 -> 9.3 KB for the FiLM field) and saves the v_cos.
// Cost in accuracy: X' differs from X by <= 1 ulp (6e-8 relative, below the dW GEMM's own rounding), and the rebuilt
// |cos| carries X's 2e-7 error amplified by |X|/|cos|: more than 2e-4 absolute only where |cos| < 1e-3, i.e. for
// 0.06 % of the units - 2e-5 of the factor's RMS, against the 5e-4 relative gate on every gradient tensor.
// The bit without a compare: cos(2 pi r) < 0 iff rndne(2 r) is odd (r in [-1/2, 1/2]); adding 1.5 * 2^23 rounds 2 r to
// that integer in the low mantissa bits, and one v_bfi_b32 moves its parity into X - two instructions instead of a
// compare, a select and an and-or.
__device__ __forceinline__ float cos_sign_into(float sn, float r) {
    const float y = fmaf(r, 2.f, 12582912.f);
    return __uint_as_float((__float_as_uint(sn) & ~1u) | (__float_as_uint(y) & 1u));
}
struct SinSaved { float s, saved; };
__device__ __forceinline__ SinSaved hw_sin_w_saved(float u, float w0) {
    const float r = hw_turns_w(u, w0);
    const float sn = __builtin_amdgcn_sinf(r);
    return {sn, cos_sign_into(sn, r)};
}
__device__ __forceinline__ SinSaved hw_sin30_saved(float u) { return hw_sin_w_saved(u, 30.f); }
// w0sq = fl(w0 w0) (900 for the literal): the derivative factor is +-sqrt(w0^2 (1 - X^2)) = w0 |cos(w0 u)|, w0 > 0
__device__ __forceinline__ float dsin_w_from_saved(float xs, float w0sq) {
    // w0^2 (1 - X^2); |X| <= 1 makes it non-negative, and the |.| (a free source modifier of v_sqrt) keeps a
    // transcendental-unit result one ulp above 1 from turning into a NaN
    // (the root is non-negative, so the sign goes in with an OR: one v_lshl_or_b32 instead of a shift and an xor)
    const float y = fmaf(xs * -w0sq, xs, w0sq);
    return __uint_as_float((__float_as_uint(xs) << 31) | __float_as_uint(__builtin_amdgcn_sqrtf(fabsf(y))));
}
__device__ __forceinline__ float dsin30_from_saved(float xs) { return dsin_w_from_saved(xs, 900.f); }
// Four saved values (one row quarter) at a time: the squares and the fma as two packed instructions each - 3 instead of 4
// VALU instructions per element, same roundings as the scalar form (fl(fl(x * -900) * x + 900)).
typedef float f32x4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4d dsin_w_from_saved_x4(f32x4d xs, float w0sq) {
#pragma clang fp contract(off)
    const f32x4d k = {w0sq, w0sq, w0sq, w0sq};
    const f32x4d y = __builtin_elementwise_fma(xs * -w0sq, xs, k);
    f32x4d o;
    o.x = __uint_as_float((__float_as_uint(xs.x) << 31) | __float_as_uint(__builtin_amdgcn_sqrtf(fabsf(y.x))));
    o.y = __uint_as_float((__float_as_uint(xs.y) << 31) | __float_as_uint(__builtin_amdgcn_sqrtf(fabsf(y.y))));
    o.z = __uint_as_float((__float_as_uint(xs.z) << 31) | __float_as_uint(__builtin_amdgcn_sqrtf(fabsf(y.z))));
    o.w = __uint_as_float((__float_as_uint(xs.w) << 31) | __float_as_uint(__builtin_amdgcn_sqrtf(fabsf(y.w))));
    return o;
}
__device__ __forceinline__ f32x4d dsin30_from_saved_x4(f32x4d xs) { return dsin_w_from_saved_x4(xs, 900.f); }
struct SinSaved2 { f32x2 s, saved; };
__device__ __forceinline__ SinSaved2 hw_sin30_saved_x2(f32x2 u) {
    const f32x2 r = hw_turns30_x2(u);
    const f32x2 sn = {__builtin_amdgcn_sinf(r.x), __builtin_amdgcn_sinf(r.y)};
    SinSaved2 o;
    o.s = sn;
    o.saved.x = cos_sign_into(sn.x, r.x);
    o.saved.y = cos_sign_into(sn.y, r.y);
    return o;
}

// Four elements (one epilogue quarter) at a time.  A dependent chain of packed fp32 ops costs a wait state between any
// two of them (the ISA listing of the two-wide form shows an `s_nop 0` after every v_pk_mul / v_pk_fma: an issue slot
// each, on a wave whose every issue slot between two MFMAs is paid in full); written four wide, each step is TWO
// independent packed instructions back to back and the wait states disappear: 4.5 instead of 6.5-7 issue slots per
// element for the default reduction.
typedef float f32x4m __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4m hw_turns_w_x4(f32x4m u, float w0) {
#pragma clang fp contract(off)
    const float c_hi = 0.15915494309189535f;
#if MI_SIN_VARIANT == 0
    const float c_lo = (float)(0.15915494309189533577 - (double)0.15915494309189535f);
    const f32x4m t = u * w0, hi = t * c_hi;
    const f32x4m n = {rintf(hi.x), rintf(hi.y), rintf(hi.z), rintf(hi.w)};
    const f32x4m k_hi = {c_hi, c_hi, c_hi, c_hi}, k_lo = {c_lo, c_lo, c_lo, c_lo};
    return __builtin_elementwise_fma(t, k_lo, __builtin_elementwise_fma(t, k_hi, -n));
#elif MI_SIN_VARIANT == 1
    const f32x4m t = u * w0, hi = t * c_hi;
    const f32x4m n = {rintf(hi.x), rintf(hi.y), rintf(hi.z), rintf(hi.w)};
    const f32x4m k_hi = {c_hi, c_hi, c_hi, c_hi};
    return __builtin_elementwise_fma(t, k_hi, -n);
#elif MI_SIN_VARIANT == 2
    const f32x4m hi = (u * w0) * c_hi;
    return f32x4m{__builtin_amdgcn_fractf(hi.x), __builtin_amdgcn_fractf(hi.y), __builtin_amdgcn_fractf(hi.z), __builtin_amdgcn_fractf(hi.w)};
#else
    const f32x4m hi = u * (w0 * 0.15915494309189535f);
    return f32x4m{__builtin_amdgcn_fractf(hi.x), __builtin_amdgcn_fractf(hi.y), __builtin_amdgcn_fractf(hi.z), __builtin_amdgcn_fractf(hi.w)};
#endif
}
__device__ __forceinline__ f32x4m hw_sin4(f32x4m r) {
#if defined(MI_DIAG_SIN) && MI_DIAG_SIN == 3
    return r;
#else
    return f32x4m{__builtin_amdgcn_sinf(r.x), __builtin_amdgcn_sinf(r.y), __builtin_amdgcn_sinf(r.z), __builtin_amdgcn_sinf(r.w)};
#endif
}
__device__ __forceinline__ f32x4m hw_turns30_x4(f32x4m u) { return hw_turns_w_x4(u, 30.f); }
__device__ __forceinline__ f32x4m hw_sin_w_x4(f32x4m u, float w0) {
#if defined(MI_DIAG_SIN) && MI_DIAG_SIN == 1
    return u;
#elif defined(MI_DIAG_SIN) && MI_DIAG_SIN == 2
    return hw_sin4(u);
#else
    return hw_sin4(hw_turns_w_x4(u, w0));
#endif
}
__device__ __forceinline__ f32x4m hw_sin30_x4(f32x4m u) { return hw_sin_w_x4(u, 30.f); }
struct SinSaved4 { f32x4m s, saved; };
__device__ __forceinline__ SinSaved4 hw_sin_w_saved_x4(f32x4m u, float w0) {
    const f32x4m r = hw_turns_w_x4(u, w0);
    const f32x4m sn = hw_sin4(r);
    SinSaved4 o;
    o.s = sn;
    o.saved = f32x4m{cos_sign_into(sn.x, r.x), cos_sign_into(sn.y, r.y), cos_sign_into(sn.z, r.z), cos_sign_into(sn.w, r.w)};
    return o;
}
__device__ __forceinline__ SinSaved4 hw_sin30_saved_x4(f32x4m u) { return hw_sin_w_saved_x4(u, 30.f); }

}  // namespace mi
