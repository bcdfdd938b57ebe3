// Diagnostic (not part of the product): accuracy of sin(30 u) variants on the transcendental unit vs fp64.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/sin_variants.hip -o gpurun_tools/sin_variants
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
// t = fl(30 u) as torch.sin(30 * x) sees it; revolutions = t / (2 pi) as a two-float product (no contraction of hi)
// hipcc contracts a*b-c into fma across statements by default, and __fmul_rn / __fsub_rn are header functions compiled
// WITH contraction (their operations carry the flag wherever they are inlined): the reductions below need the ROUNDED
// product, so they use plain operators inside bodies that switch contraction off.
// (the pragma only takes effect INSIDE a function body)
#define NO_CONTRACT _Pragma("clang fp contract(off)")
__device__ __forceinline__ void two_prod(float u, float& hi, float& lo) {
    NO_CONTRACT
    const float c_hi = 0.15915494309189535f;
    const float c_lo = (float)(0.15915494309189533577 - (double)0.15915494309189535f);
    const float t = (30.f * u);
    hi = (t * c_hi);
    lo = fmaf(t, c_lo, fmaf(t, c_hi, -hi));
}
__device__ float v0(float u) { NO_CONTRACT float hi, lo; two_prod(u, hi, lo); return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(hi) + lo); }
__device__ float v1(float u) { NO_CONTRACT float hi, lo; two_prod(u, hi, lo); return __builtin_amdgcn_sinf(((hi - rintf(hi)) + lo)); }
__device__ float v2(float u) {
    NO_CONTRACT
    float hi, lo; two_prod(u, hi, lo);
    const float r = ((hi - rintf(hi)) + lo);
    return __builtin_amdgcn_sinf(__builtin_amdgcn_fmed3f(r, 0.5f - r, -0.5f - r));
}
__device__ float v3(float u) {   // fold + polynomial for sin(2 pi r), |r| <= 1/4
    NO_CONTRACT
    float hi, lo; two_prod(u, hi, lo);
    float r = ((hi - rintf(hi)) + lo);
    r = __builtin_amdgcn_fmed3f(r, 0.5f - r, -0.5f - r);
    const float s = r * r;
    float p = fmaf(s, 39.76049716f, -76.58125642f);      // least-squares fit on Chebyshev nodes: 6.7e-9 max error
    p = fmaf(p, s, 81.60247959f);
    p = fmaf(p, s, -41.34168065f);
    p = fmaf(p, s, 6.28318528f);
    return p * r;
}
__device__ float v4(float u) { return sinf(30.f * u); }   // libm on the product rounded to fp32 (what torch.sin(30*x) sees)
__device__ float v5(float u) {   // n from the rounded product, both fmas against n: mul, rndne, fma, fma (4 instead of 6)
    NO_CONTRACT
    const float c_hi = 0.15915494309189535f;
    const float c_lo = (float)(0.15915494309189533577 - (double)0.15915494309189535f);
    const float t = (30.f * u);
    const float n = rintf((t * c_hi));
    return __builtin_amdgcn_sinf(fmaf(t, c_lo, fmaf(t, c_hi, -n)));
}
__device__ float v6(float u) { NO_CONTRACT const float c_hi = 0.15915494309189535f; const float t = (30.f * u); return __builtin_amdgcn_sinf(fmaf(t, c_hi, -rintf((t * c_hi)))); }
__device__ float v7(float u) { NO_CONTRACT return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(((30.f * u) * 0.15915494309189535f))); }
__device__ float v8(float u) { NO_CONTRACT return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf((u * 4.774648292756860f))); }
__device__ float v9(float u) { NO_CONTRACT return __builtin_amdgcn_sinf(((30.f * u) * 0.15915494309189535f)); }   // no reduction: v_sin's own
__global__ void run(const float* u, float* o, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    o[0 * n + i] = v0(u[i]); o[1 * n + i] = v1(u[i]); o[2 * n + i] = v2(u[i]); o[3 * n + i] = v3(u[i]); o[4 * n + i] = v4(u[i]); o[5 * n + i] = v5(u[i]); o[6 * n + i] = v6(u[i]); o[7 * n + i] = v7(u[i]); o[8 * n + i] = v8(u[i]); o[9 * n + i] = v9(u[i]);
}
int main() {
    const int n = 1 << 22, nv = 10;
    std::vector<float> u(n), o((size_t)nv * n);
    srand(1);
    for (int i = 0; i < n; ++i) u[i] = (float)((rand() / (double)RAND_MAX * 2 - 1) * (i % 4 == 0 ? 40.0 : i % 4 == 1 ? 4.0 : i % 4 == 2 ? 0.5 : 0.05));
    float *du, *dout;
    (void)hipMalloc(&du, n * 4); (void)hipMalloc(&dout, (size_t)nv * n * 4);
    (void)hipMemcpy(du, u.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(run, dim3(n / 256), dim3(256), 0, 0, du, dout, n);
    (void)hipMemcpy(o.data(), dout, (size_t)nv * n * 4, hipMemcpyDeviceToHost);
    const char* names[] = {"v0 fract", "v1 centred (two-float)", "v2 centred+fold", "v3 fold+poly", "v4 libm sinf(30u)", "v5 fma against n (4 ops)",
                           "v6 rndne + one fma, no c_lo", "v7 fract(fl(30u) c_hi)", "v8 fract(u * 30/2pi)", "v9 v_sin(fl(30u) c_hi)"};
    for (int v = 0; v < nv; ++v) {
        double mx = 0, ss = 0;
        for (int i = 0; i < n; ++i) {
            // reference: sin of the fp32-rounded product 30*u evaluated exactly (what the fp32 oracle computes) and of the exact product
            const double e = (double)o[(size_t)v * n + i] - sin(30.0 * (double)u[i]);
            mx = fmax(mx, fabs(e)); ss += e * e;
        }
        double mx2 = 0, ss2 = 0;
        for (int i = 0; i < n; ++i) {
            const double e = (double)o[(size_t)v * n + i] - sin((double)(30.f * u[i]));
            mx2 = fmax(mx2, fabs(e)); ss2 += e * e;
        }
        for (int rgn = 0; rgn < 4; ++rgn) {          // |u| < 40 | 4 | 0.5 | 0.05 (i % 4 selects the range above)
            double m3 = 0, s3 = 0; int cnt = 0;
            for (int i = rgn; i < n; i += 4) { const double e = (double)o[(size_t)v * n + i] - sin((double)(30.f * u[i])); m3 = fmax(m3, fabs(e)); s3 += e * e; ++cnt; }
            printf("    |u| < %-5g vs sin(fl(30u)): max %.3e rms %.3e\n", rgn == 0 ? 40.0 : rgn == 1 ? 4.0 : rgn == 2 ? 0.5 : 0.05, m3, sqrt(s3 / cnt));
        }
        printf("%-22s vs sin(30u exact): max %.3e rms %.3e | vs sin(fl(30u)): max %.3e rms %.3e\n", names[v], mx, sqrt(ss / n), mx2, sqrt(ss2 / n));
    }
    return 0;
}
