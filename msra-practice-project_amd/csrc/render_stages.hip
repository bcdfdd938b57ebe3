// render_stages.hip - the HBM-bound stages around the fused field MLP (gfx950).
//
//   gen_rays_kernel       get_rays                       reference nerf/render.py:7-23
//   sample_coarse_kernel  stratified depths              nerf/render.py:123-132
//   composite_kernel      raw_to_outputs                 nerf/render.py:78-103
//   sample_fine_kernel    sample_pdf + detach/cat/sort   nerf/render.py:27-56, 140-142
//
// All arithmetic is fp32 in the reference's operation order (this file is compiled with
// -ffp-contract=off so a*b+c stays two roundings like the un-fused torch/NumPy ops).
// These stages move 20-30 B per sample against 1.2 MFLOP per sample in the MLP: they are
// coalesced streaming kernels, not worth fusing further.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mi_common.h"

namespace mi {

// ---------------------------------------------------------------------------------------
// get_rays: ray `idx` of the row-major H*W list; dirs = ((i-W/2)/f, -(j-H/2)/f, -1);
// d = sum_k dirs[k]*c2w[c][k] accumulated left to right like np.sum over 3 elements.
// ---------------------------------------------------------------------------------------
struct Cam { float m[12]; };

// T = float reproduces NumPy when `focal` is a Python float (weak scalar -> all-fp32 math);
// T = double reproduces it when `focal` is an np.float64 scalar (pi_GAN/modules.py:127: the
// whole expression is promoted to fp64 and rounded to fp32 by torch.tensor(..., dtype=float)).
template <class T>
__global__ void gen_rays_kernel(int width, T half_w, T half_h, T focal, Cam cam, int64_t ray0, int64_t n,
                                float* __restrict__ rays) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int64_t idx = ray0 + t;
    const T i = (T)(float)(idx % width), j = (T)(float)(idx / width);
    const T d0 = (i - half_w) / focal;
    const T d1 = -((j - half_h) / focal);
    const T d2 = (T)-1;
    float* o = rays + t * 6;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o[c] = cam.m[4 * c + 3];
        o[3 + c] = (float)((d0 * (T)cam.m[4 * c + 0] + d1 * (T)cam.m[4 * c + 1]) + d2 * (T)cam.m[4 * c + 2]);
    }
}

// ---------------------------------------------------------------------------------------
// linspace(start, end, steps)[i] as ATen's scalar formula (RangeFactories: symmetric halves).
// The host may instead pass the table torch.linspace produced so z matches the CPU oracle
// bit for bit (its vectorised path differs from this formula by 1 ulp on a few entries).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i) {
    if (steps <= 1) return start;
    const float step = (end - start) / (float)(steps - 1);
    return i < steps / 2 ? start + step * (float)i : end - step * (float)(steps - i - 1);
}

// Philox4x32-10 (Salmon et al. 2011), counter = (ray_lo, ray_hi, sample>>2, 0), key = seed.
__device__ __forceinline__ uint32_t philox_u32(uint64_t seed, uint64_t ray, uint32_t sample) {
    uint32_t c0 = (uint32_t)ray, c1 = (uint32_t)(ray >> 32), c2 = sample >> 2, c3 = 0;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const uint32_t s = sample & 3;
    return s == 0 ? c0 : (s == 1 ? c1 : (s == 2 ? c2 : c3));
}

__global__ void sample_coarse_kernel(int64_t n, float near_, float far_, int nc, const float* __restrict__ z_lin,
                                     const float* __restrict__ t_rand, uint64_t seed, uint64_t ray0,
                                     float* __restrict__ z) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * nc) return;
    const int k = (int)(t % nc);
    const int64_t ray = t / nc;
    auto lin = [&](int i) { return z_lin ? z_lin[i] : linspace_at(near_, far_, nc, i); };
    const float zk = lin(k);
    const float lower = k == 0 ? zk : 0.5f * (zk + lin(k - 1));
    const float upper = k == nc - 1 ? zk : 0.5f * (lin(k + 1) + zk);
    const float u = t_rand ? t_rand[t] : (float)(philox_u32(seed, ray0 + (uint64_t)ray, (uint32_t)k) >> 8) * 0x1p-24f;
    z[t] = lower + (upper - lower) * u;
}

// ---------------------------------------------------------------------------------------
// composite: G lanes per ray (G = 16/32/64), lane = sample, passes of G samples with the
// transmittance carried between passes.  alpha = 1-exp(-sigma*delta*|d|),
// T = exclusive prod(1-alpha+1e-10), w = alpha*T; rgb = sum w c + (1-acc) (white background).
// ---------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void composite_kernel(int64_t n, int S, const float* __restrict__ raw,
                                                        const float* __restrict__ z, const float* __restrict__ rays,
                                                        float* __restrict__ rgb, float* __restrict__ depth,
                                                        float* __restrict__ acc, float* __restrict__ weights) {
    const int lane = threadIdx.x & 63;
    const int sub = lane & (G - 1);
    const int64_t groups_per_block = 256 / G;
    const int64_t ray = (int64_t)blockIdx.x * groups_per_block + threadIdx.x / G;
    const bool live = ray < n;
    const int64_t rc = live ? ray : n - 1;
    const float* rd = rays + rc * 6 + 3;
    const float nrm = sqrtf((rd[0] * rd[0] + rd[1] * rd[1]) + rd[2] * rd[2]);
    double T = 1.0;
    float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
    for (int k0 = 0; k0 < S; k0 += G) {
        const int k = k0 + sub;
        const bool in = k < S;
        const int kc = in ? k : S - 1;
        const float4 c = reinterpret_cast<const float4*>(raw)[rc * S + kc];
        const float zk = z[rc * S + kc];
        const float zn = kc + 1 < S ? z[rc * S + kc + 1] : 0.f;
        float delta = kc + 1 < S ? zn - zk : 1e10f;
        delta = delta * nrm;
        const float alpha = in ? 1.0f - expf(-c.w * delta) : 0.f;
        const float f = in ? (1.0f - alpha) + 1e-10f : 1.f;
        // inclusive product scan inside the G-lane group.  The running product is kept in fp64 and
        // rounded to fp32 per sample, as ATen's CPU cumprod does (accumulate type of float is double).
        double p = (double)f;
#pragma unroll
        for (int o = 1; o < G; o <<= 1) {
            const double q = __shfl_up(p, o, G);
            if (sub >= o) p *= q;
        }
        double excl = __shfl_up(p, 1, G);
        if (sub == 0) excl = 1.0;
        const float w = alpha * (float)(T * excl);
        T = T * __shfl(p, G - 1, G);
        if (in) {
            sr += w * c.x; sg += w * c.y; sb += w * c.z; sd += w * zk; sa += w;
            if (weights && live) weights[rc * S + k] = w;
        }
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
        sr += __shfl_xor(sr, o, G); sg += __shfl_xor(sg, o, G); sb += __shfl_xor(sb, o, G);
        sd += __shfl_xor(sd, o, G); sa += __shfl_xor(sa, o, G);
    }
    if (live && sub == 0) {
        const float bg = 1.0f - sa;
        rgb[ray * 3 + 0] = sr + bg; rgb[ray * 3 + 1] = sg + bg; rgb[ray * 3 + 2] = sb + bg;
        depth[ray] = sd;
        acc[ray] = sa;
    }
}

// ---------------------------------------------------------------------------------------
// composite backward (autograd of raw_to_outputs, render.py:91-101): G lanes per ray, lane = sample.
//   Gk = sum_c g_rgb[c]*(c_k[c]-1) + g_depth*z_k + g_acc        (dL/dw_k; rgb carries +1-acc)
//   dL/dalpha_k = T_k * (Gk - S_k),  S_k = sum_{j>k} Gj alpha_j prod_{k<i<j} f_i = R_{k+1} with the
//                 division-free first-order recurrence R_k = Gk alpha_k + f_k R_{k+1}
//   dL/dsigma_k = dL/dalpha_k * delta_k * (1-alpha_k);   dL/dc_k = g_rgb * w_k
// f_k = 1-alpha_k+1e-10.  z and rays carry no gradient (z_samples is detached, render.py:141).
// Sweep 1 (passes of G samples, front to back): the transmittance T_k as in composite_kernel (fp64 product
// scan), parked in the output's .w slot.  Sweep 2 (back to front): R by a reverse scan of the affine maps
// (f_k, Gk alpha_k) inside the pass, carried between passes.
// ---------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void composite_bwd_kernel(int64_t n, int S, const float* __restrict__ raw,
                                                            const float* __restrict__ z,
                                                            const float* __restrict__ rays,
                                                            const float* __restrict__ g_rgb,
                                                            const float* __restrict__ g_depth,
                                                            const float* __restrict__ g_acc,
                                                            const float* __restrict__ g_w,
                                                            float* __restrict__ g_raw) {
    const int lane = threadIdx.x & 63;
    const int sub = lane & (G - 1);
    const int64_t ray = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G;
    const bool live = ray < n;
    const int64_t rc = live ? ray : n - 1;
    const float* rd = rays + rc * 6 + 3;
    const float nrm = sqrtf((rd[0] * rd[0] + rd[1] * rd[1]) + rd[2] * rd[2]);
    const float gr = g_rgb ? g_rgb[rc * 3 + 0] : 0.f, gg = g_rgb ? g_rgb[rc * 3 + 1] : 0.f,
                gb = g_rgb ? g_rgb[rc * 3 + 2] : 0.f;
    const float gd = g_depth ? g_depth[rc] : 0.f, ga = g_acc ? g_acc[rc] : 0.f;
    const float4* rw = reinterpret_cast<const float4*>(raw) + rc * S;
    const float* zr = z + rc * S;
    float4* out = reinterpret_cast<float4*>(g_raw) + rc * S;
    const int passes = (S + G - 1) / G;
    double T = 1.0;
    for (int pass = 0; pass < passes; ++pass) {
        const int k = pass * G + sub;
        const bool in = k < S;
        const int kc = in ? k : S - 1;
        const float delta = (kc + 1 < S ? zr[kc + 1] - zr[kc] : 1e10f) * nrm;
        const float alpha = in ? 1.0f - expf(-rw[kc].w * delta) : 0.f;
        double p = in ? (double)((1.0f - alpha) + 1e-10f) : 1.0;
#pragma unroll
        for (int o = 1; o < G; o <<= 1) {
            const double q = __shfl_up(p, o, G);
            if (sub >= o) p *= q;
        }
        double excl = __shfl_up(p, 1, G);
        if (sub == 0) excl = 1.0;
        if (in && live) out[k].w = (float)(T * excl);
        T = T * __shfl(p, G - 1, G);
    }
    float carry = 0.f;                                   // R at the first sample of the pass behind this one
    for (int pass = passes - 1; pass >= 0; --pass) {
        const int k = pass * G + sub;
        const bool in = k < S;
        const int kc = in ? k : S - 1;
        const float4 c = rw[kc];
        const float zk = zr[kc];
        const float delta = (kc + 1 < S ? zr[kc + 1] - zk : 1e10f) * nrm;
        const float e = expf(-c.w * delta);
        const float alpha = 1.0f - e;
        const float Tk = in && live ? out[kc].w : 0.f;
        // dL/dw_k; g_w = the cotangent of the returned weights themselves (raw_to_outputs hands them out, render.py:103)
        float Gk = gr * (c.x - 1.f) + gg * (c.y - 1.f) + gb * (c.z - 1.f) + gd * zk + ga;
        if (g_w) Gk += g_w[rc * S + kc];
        // affine map of this sample, R_k = A + M * R_{k+1}; composed towards higher lanes
        float M = in ? (1.0f - alpha) + 1e-10f : 1.f, A = in ? Gk * alpha : 0.f;
#pragma unroll
        for (int o = 1; o < G; o <<= 1) {
            const float M2 = __shfl_down(M, o, G), A2 = __shfl_down(A, o, G);
            if (sub + o < G) { A = A + M * A2; M = M * M2; }
        }
        const float R = A + M * carry;                   // R_k
        float Snext = __shfl_down(R, 1, G);              // S_k = R_{k+1}
        if (sub == G - 1) Snext = carry;
        carry = __shfl(R, 0, G);
        if (in && live) {
            const float w = alpha * Tk;
            const float dalpha = Tk * (Gk - Snext);
            out[k] = make_float4(gr * w, gg * w, gb * w, dalpha * delta * e);
        }
    }
}

// ---------------------------------------------------------------------------------------
// cdf[j] = fp32(sum_{i<j} pdf[i]) with the running sum in fp64, j = 0..nb-1, pdf[0..nb-2] in LDS: ATen's CPU cumsum
// keeps the running sum in the accumulate type of float, i.e. DOUBLE, and rounds every output to fp32
// (cpu_cum_base_kernel).  A double holds every partial sum of these floats EXACTLY whenever the terms' exponents span
// few enough bits (largest sum's exponent - smallest term's last mantissa bit <= 52), and exact sums do not depend
// on the order of the additions: the wave then takes a shuffle scan (6 steps) instead of nb dependent additions.
// Terms that do not qualify (inf / NaN, or a dynamic range beyond 2^(28 - log2 nb)) take the sequential order itself.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void cdf_like_cumsum(const float* pdf, float* cdf, int nb, int lane) {
    const int nw = nb - 1;
    int fmin = 255, fmax = 0;
    for (int j = lane; j < nw; j += 64) {
        const int f = (int)((__float_as_uint(pdf[j]) >> 23) & 0xffu);
        const bool zero = (__float_as_uint(pdf[j]) << 1) == 0u;
        fmax = f > fmax ? f : fmax;
        if (!zero) { const int g = f > 1 ? f : 1; fmin = g < fmin ? g : fmin; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int a = __shfl_xor(fmin, o), b = __shfl_xor(fmax, o);
        fmin = a < fmin ? a : fmin;
        fmax = b > fmax ? b : fmax;
    }
    const int log2n = 32 - __builtin_clz((unsigned)(nb > 1 ? nb - 1 : 1));
    const bool exact = fmax < 255 && (fmin > fmax || fmax - fmin <= 28 - log2n);
    if (exact) {
        double carry = 0.0;
        for (int base = 0; base < nb; base += 64) {
            const int j = base + lane;
            double incl = j < nw ? (double)pdf[j] : 0.0;
            const double own = incl;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const double up = __shfl_up(incl, o);
                if (lane >= o) incl += up;
            }
            if (j < nb) cdf[j] = (float)(carry + (incl - own));      // exclusive prefix (exact, like every sum here)
            carry += __shfl(incl, 63);
        }
    } else {
        for (int j = lane; j < nb; j += 64) {
            double sum = 0.0;
            for (int i = 0; i < j; ++i) sum += (double)pdf[i];
            cdf[j] = (float)sum;
        }
    }
}

// (value, index) as one unsigned key: float order for everything that is not a NaN (with -0 before +0), ties by
// index - the rank sort below then needs one 64-bit compare per pair instead of two float compares and an index test
__device__ __forceinline__ uint64_t sort_key(float v, int i) {
    uint32_t b = __float_as_uint(v);
    b ^= (b >> 31) ? 0xffffffffu : 0x80000000u;
    return ((uint64_t)b << 32) | (uint32_t)i;
}

// ---------------------------------------------------------------------------------------
// sample_fine: one wave per ray.  bins = mids of the coarse linspace, w = weights[1:-1]+1e-5,
// pdf = w/sum(w), cdf = [0, cumsum(pdf)] with the running sum in fp64 rounded per entry like
// torch.cumsum on CPU; u = linspace(0,1,Nf); idx = #(cdf <= u) (searchsorted right=True); guarded lerp;
// then z_fine = sort(cat(z_coarse, z_samples)) by full rank counting (no sortedness assumed; -0 ranks before +0).
// LDS per wave: cdf[Nc] | bins[Nc] | zall[Nc+Nf] | zout[Nc+Nf] (the sorted row, stored coalesced)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sample_fine_kernel(int64_t n, float near_, float far_, int nc, int nf,
                                                          const float* __restrict__ z_lin,
                                                          const float* __restrict__ u_lin,
                                                          const float* __restrict__ z_coarse,
                                                          const float* __restrict__ weights,
                                                          float* __restrict__ z_samples, float* __restrict__ z_fine,
                                                          int* __restrict__ pos) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = nc + nf;
    const int per_wave = 2 * nc + 2 * S;
    float* cdf = smem + wave * per_wave;
    float* bins = cdf + nc;
    float* zall = bins + nc;
    float* zout = zall + S;
    const int nb = nc - 1;      // bins / cdf entries
    const int nw = nc - 2;      // interior weights
    auto lin = [&](int i) { return z_lin ? z_lin[i] : linspace_at(near_, far_, nc, i); };
    for (int j = lane; j < nb; j += 64) bins[j] = 0.5f * (lin(j + 1) + lin(j));

    for (int64_t ray = (int64_t)blockIdx.x * 4 + wave; ray < n; ray += (int64_t)gridDim.x * 4) {
        const float* wr = weights + ray * nc;
        const float* zc = z_coarse + ray * nc;
        // sum of (w + 1e-5)
        float part = 0.f;
        for (int j = lane; j < nw; j += 64) part += wr[j + 1] + 1e-5f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        const float total = part;
        // pdf into zall (scratch), then per-entry sequential prefix = torch.cumsum order
        for (int j = lane; j < nw; j += 64) zall[j] = (wr[j + 1] + 1e-5f) / total;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        cdf_like_cumsum(zall, cdf, nb, lane);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // coarse depths first, then inverse-CDF samples
        for (int j = lane; j < nc; j += 64) zall[j] = zc[j];
        for (int s = lane; s < nf; s += 64) {
            const float u = u_lin ? u_lin[s] : linspace_at(0.f, 1.f, nf, s);
            int lo_i = 0, hi_i = nb;                 // count of cdf entries <= u  (cdf ascending)
            while (lo_i < hi_i) {
                const int mid = (lo_i + hi_i) >> 1;
                if (cdf[mid] <= u) lo_i = mid + 1; else hi_i = mid;
            }
            const int idx = lo_i;
            const int below = idx - 1 > 0 ? idx - 1 : 0;
            const int above = idx < nb - 1 ? idx : nb - 1;
            const float c0 = cdf[below], c1 = cdf[above];
            float denom = c1 - c0;
            if (denom < 1e-5f) denom = 1.f;
            const float t = (u - c0) / denom;
            const float b0 = bins[below], b1 = bins[above];
            const float zs = b0 + t * (b1 - b0);
            zall[nc + s] = zs;
            if (z_samples) z_samples[ray * nf + s] = zs;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // Merge by rank: position of an element = #(keys below its own), key = (value, index).  Both lists are
        // sorted on every call the render path makes (stratified depths; inverse-CDF samples of an ascending u), and
        // then a rank is the element's own index in its list plus one binary search in the other list.  A ray whose
        // lists are not sorted (the stage-level entry point takes any depths) is ranked by full counting instead.
        bool sorted = true;
        for (int e = lane; e + 1 < S; e += 64)
            if (e + 1 != nc) sorted = sorted && !(zall[e + 1] < zall[e]) && zall[e] == zall[e] && zall[e + 1] == zall[e + 1];
        if (__builtin_amdgcn_ballot_w64(!sorted) == 0ull) {
            for (int e = lane; e < S; e += 64) {
                const float v = zall[e];
                const bool coarse = e < nc;
                // coarse element: #(samples < v) (an equal sample has the higher index); sample: #(coarse <= v)
                const float* other = coarse ? zall + nc : zall;
                int lo_i = 0, hi_i = coarse ? nf : nc;
                while (lo_i < hi_i) {
                    const int mid = (lo_i + hi_i) >> 1;
                    const float o = other[mid];
                    if (coarse ? o < v : o <= v) lo_i = mid + 1; else hi_i = mid;
                }
                zout[(coarse ? e : e - nc) + lo_i] = v;
                if (pos) pos[ray * S + e] = (coarse ? e : e - nc) + lo_i;
            }
        } else {
            for (int e0 = lane; e0 < S; e0 += 256) {                 // up to four elements per sweep over the list
                uint64_t key[4];
                int rank[4] = {0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < 4; ++k) key[k] = e0 + 64 * k < S ? sort_key(zall[e0 + 64 * k], e0 + 64 * k) : ~0ull;
#pragma unroll 8
                for (int i = 0; i < S; ++i) {
                    const uint64_t o = sort_key(zall[i], i);
#pragma unroll
                    for (int k = 0; k < 4; ++k) rank[k] += o < key[k] ? 1 : 0;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (e0 + 64 * k < S) {
                        zout[rank[k]] = zall[e0 + 64 * k];
                        if (pos) pos[ray * S + e0 + 64 * k] = rank[k];
                    }
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int e = lane; e < S; e += 64) z_fine[ray * S + e] = zout[e];        // coalesced rows out of LDS
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// ---------------------------------------------------------------------------------------
// sample_pdf as a free-standing function (render.py:27-56) on arbitrary per-ray bins: one wave per ray,
// same arithmetic as sample_fine_kernel without the merge.  bins [n,nb], weights [n,nb-1] -> out [n,ns].
// LDS per wave: cdf[nb] | pdf[nb]
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sample_pdf_kernel(int64_t n, int nb, int ns, const float* __restrict__ bins,
                                                         const float* __restrict__ weights,
                                                         const float* __restrict__ u_lin, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* cdf = smem + wave * 2 * nb;
    float* pdf = cdf + nb;
    const int nw = nb - 1;
    for (int64_t ray = (int64_t)blockIdx.x * 4 + wave; ray < n; ray += (int64_t)gridDim.x * 4) {
        const float* wr = weights + ray * nw;
        const float* br = bins + ray * nb;
        float part = 0.f;
        for (int j = lane; j < nw; j += 64) part += wr[j] + 1e-5f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        for (int j = lane; j < nw; j += 64) pdf[j] = (wr[j] + 1e-5f) / part;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        cdf_like_cumsum(pdf, cdf, nb, lane);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int s = lane; s < ns; s += 64) {
            const float u = u_lin ? u_lin[s] : linspace_at(0.f, 1.f, ns, s);
            int lo_i = 0, hi_i = nb;
            while (lo_i < hi_i) {
                const int mid = (lo_i + hi_i) >> 1;
                if (cdf[mid] <= u) lo_i = mid + 1; else hi_i = mid;
            }
            const int below = lo_i - 1 > 0 ? lo_i - 1 : 0;
            const int above = lo_i < nb - 1 ? lo_i : nb - 1;
            const float c0 = cdf[below], c1 = cdf[above];
            float denom = c1 - c0;
            if (denom < 1e-5f) denom = 1.f;
            const float t = (u - c0) / denom;
            const float b0 = br[below], b1 = br[above];
            out[ray * ns + s] = b0 + t * (b1 - b0);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// ---- host launchers ------------------------------------------------------------------------
int launch_gen_rays(int width, int height, double focal, const float* c2w, int64_t ray0, int64_t n, float* rays,
                    int compute_f64, hipStream_t stream) {
    if (n <= 0) return 0;
    Cam cam;
    for (int i = 0; i < 12; ++i) cam.m[i] = c2w[i];
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (compute_f64)
        hipLaunchKernelGGL(gen_rays_kernel<double>, dim3(blocks), dim3(256), 0, stream, width, width * 0.5,
                           height * 0.5, focal, cam, ray0, n, rays);
    else
        hipLaunchKernelGGL(gen_rays_kernel<float>, dim3(blocks), dim3(256), 0, stream, width, (float)(width * 0.5),
                           (float)(height * 0.5), (float)focal, cam, ray0, n, rays);
    return check_launch("gen_rays");
}

int launch_sample_coarse(int64_t n, float near_, float far_, int nc, const float* z_lin, const float* t_rand,
                         uint64_t seed, uint64_t ray0, float* z, hipStream_t stream) {
    if (n <= 0) return 0;
    const int64_t total = n * nc;
    hipLaunchKernelGGL(sample_coarse_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, n, near_,
                       far_, nc, z_lin, t_rand, seed, ray0, z);
    return check_launch("sample_coarse");
}

int launch_composite(int64_t n, int S, const float* raw, const float* z, const float* rays, float* rgb, float* depth,
                     float* acc, float* weights, hipStream_t stream) {
    if (n <= 0) return 0;
    if (S > 32) {
        hipLaunchKernelGGL(composite_kernel<64>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, n, S, raw, z,
                           rays, rgb, depth, acc, weights);
    } else if (S > 16) {
        hipLaunchKernelGGL(composite_kernel<32>, dim3((unsigned)((n + 7) / 8)), dim3(256), 0, stream, n, S, raw, z,
                           rays, rgb, depth, acc, weights);
    } else {
        hipLaunchKernelGGL(composite_kernel<16>, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, stream, n, S, raw, z,
                           rays, rgb, depth, acc, weights);
    }
    return check_launch("composite");
}

int launch_composite_bwd(int64_t n, int S, const float* raw, const float* z, const float* rays, const float* g_rgb,
                         const float* g_depth, const float* g_acc, const float* g_w, float* g_raw, hipStream_t stream) {
    if (n <= 0) return 0;
    const int G = S <= 16 ? 16 : (S <= 32 ? 32 : 64);
    const dim3 grid((unsigned)((n * G + 255) / 256)), block(256);
    if (G == 16) hipLaunchKernelGGL((composite_bwd_kernel<16>), grid, block, 0, stream, n, S, raw, z, rays, g_rgb, g_depth, g_acc, g_w, g_raw);
    else if (G == 32) hipLaunchKernelGGL((composite_bwd_kernel<32>), grid, block, 0, stream, n, S, raw, z, rays, g_rgb, g_depth, g_acc, g_w, g_raw);
    else hipLaunchKernelGGL((composite_bwd_kernel<64>), grid, block, 0, stream, n, S, raw, z, rays, g_rgb, g_depth, g_acc, g_w, g_raw);
    return check_launch("composite_bwd");
}

int launch_sample_pdf(int64_t n, int nb, int ns, const float* bins, const float* weights, const float* u_lin, float* out,
                      hipStream_t stream) {
    if (n <= 0 || ns <= 0) return 0;
    const size_t lds = (size_t)4 * 2 * nb * sizeof(float);
    if (lds > 160 * 1024) { set_error("sample_pdf: %d bins exceed LDS", nb); return -1; }
    int64_t blocks = (n + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, n, nb, ns, bins, weights, u_lin,
                       out);
    return check_launch("sample_pdf");
}

// ---------------------------------------------------------------------------------------
// One field for both passes (pi_GAN renders with coarse_model is fine_model, pi_GAN/modules.py:160-161; nerf with
// use_fine_model off, nerf/train_nerf.py:91,94): the fine pass of render_rays (render.py:143-144) evaluates the field at
// sort(cat(z_coarse, z_samples)) - and Nc of those Nc + Nf points are the very points the coarse pass evaluated, with the
// same field: identical inputs, identical outputs.  The renderer then evaluates the Nf NEW points only and puts the
// two sets of raw values into the sorted order with the positions the merge computed (`pos` of sample_fine_kernel:
// pos[e] = index in z_fine of input element e, e < Nc coarse, else sample e - Nc).  Generalises the Nf = 0 alias of
// SURVEY.md 8d C2; bit-identical to evaluating all Nc + Nf points because a point's value does not depend on which
// launch or lane computed it.
//   merge_raw:  raw_f[pos[e]] = e < Nc ? raw_c[e] : raw_s[e - Nc]
//   split_grad: the transpose, for the backward pass: g_c[e] (+)= g_f[pos[e]], g_s[i] = g_f[pos[Nc + i]]
// thread per (ray, element); float4 per point.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void merge_raw_kernel(int64_t n, int nc, int nf, const float4* __restrict__ raw_c,
                                                        const float4* __restrict__ raw_s, const int* __restrict__ pos,
                                                        float4* __restrict__ raw_f) {
    const int S = nc + nf;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * S) return;
    const int64_t ray = i / S;
    const int e = (int)(i - ray * S);
    raw_f[ray * S + pos[i]] = e < nc ? raw_c[ray * nc + e] : raw_s[ray * nf + (e - nc)];
}

__global__ __launch_bounds__(256) void split_grad_kernel(int64_t n, int nc, int nf, const float4* __restrict__ g_f,
                                                         const int* __restrict__ pos, float4* __restrict__ g_c,
                                                         int accumulate_coarse, float4* __restrict__ g_s) {
    const int S = nc + nf;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * S) return;
    const int64_t ray = i / S;
    const int e = (int)(i - ray * S);
    const float4 g = g_f[ray * S + pos[i]];
    if (e < nc) {
        float4* d = g_c + ray * nc + e;
        if (accumulate_coarse) { const float4 o = *d; *d = make_float4(o.x + g.x, o.y + g.y, o.z + g.z, o.w + g.w); }
        else *d = g;
    } else {
        g_s[ray * nf + (e - nc)] = g;
    }
}

int launch_merge_raw(int64_t n, int nc, int nf, const float* raw_c, const float* raw_s, const int* pos, float* raw_f,
                     hipStream_t stream) {
    const int64_t total = n * (nc + nf);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(merge_raw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, n, nc, nf,
                       (const float4*)raw_c, (const float4*)raw_s, pos, (float4*)raw_f);
    return check_launch("merge_raw");
}

int launch_split_grad(int64_t n, int nc, int nf, const float* g_f, const int* pos, float* g_c, int accumulate_coarse,
                      float* g_s, hipStream_t stream) {
    const int64_t total = n * (nc + nf);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(split_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, n, nc, nf,
                       (const float4*)g_f, pos, (float4*)g_c, accumulate_coarse, (float4*)g_s);
    return check_launch("split_grad");
}

int launch_sample_fine(int64_t n, float near_, float far_, int nc, int nf, const float* z_lin, const float* u_lin,
                       const float* z_coarse, const float* weights, float* z_samples, float* z_fine, int* pos,
                       hipStream_t stream) {
    if (n <= 0) return 0;
    const size_t lds = (size_t)4 * (2 * nc + 2 * (nc + nf)) * sizeof(float);
    if (lds > 64 * 1024) { set_error("sample_fine: Nc=%d Nf=%d exceed LDS", nc, nf); return -1; }
    int64_t blocks = (n + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(sample_fine_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, n, near_, far_, nc, nf,
                       z_lin, u_lin, z_coarse, weights, z_samples, z_fine, pos);
    return check_launch("sample_fine");
}

}  // namespace mi
