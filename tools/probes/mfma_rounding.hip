// Diagnostic (not part of the product): how does v_mfma_f32_32x32x2_f32 round?
//   case 0: c=1, a0*b0 = 2^-24 + 2^-30 (just over half an ulp of 1)      RN -> 1+2^-23, RZ -> 1
//   case 1: c=1, a0*b0 = a1*b1 = 2^-24 (each a tie; exact sum = one ulp)  fused -> 1+2^-23, separate RN-even -> 1
//   case 2: c=1, a0*b0 = -(2^-25 + 2^-30)                                 RN -> 1, RZ -> 1-2^-24
//   case 3: c=1, a0*b0 = 3*2^-25 (1.5 half-ulps)                          RN -> 1+2^-23
//   case 4: product exactness: a0 = 1+2^-23, b0 = 1+2^-23, c = -(1+2^-22) exact result 2^-46
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_rounding.hip -o gpurun_tools/mfma_rounding
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(const float* a, const float* b, const float* c, float* out, int ncase) {
    const int lane = threadIdx.x;
    for (int k = 0; k < ncase; ++k) {
        // A[i][kk]: lane = i + 32*kk holds A[i][kk];  B[kk][j]: lane = j + 32*kk holds B[kk][j]
        const float av = a[2 * k + (lane >> 5)], bv = b[2 * k + (lane >> 5)];
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = c[k];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        if (lane == 0) out[k] = acc[0];
    }
}
int main() {
    const int n = 5;
    float ha[2 * n], hb[2 * n], hc[n], ho[n];
    const float t24 = ldexpf(1.f, -24), t30 = ldexpf(1.f, -30), t25 = ldexpf(1.f, -25);
    ha[0] = t24 + t30; hb[0] = 1.f; ha[1] = 0.f; hb[1] = 0.f; hc[0] = 1.f;
    ha[2] = t24; hb[2] = 1.f; ha[3] = t24; hb[3] = 1.f; hc[1] = 1.f;
    ha[4] = -(t25 + t30); hb[4] = 1.f; ha[5] = 0.f; hb[5] = 0.f; hc[2] = 1.f;
    ha[6] = 3 * t25; hb[6] = 1.f; ha[7] = 0.f; hb[7] = 0.f; hc[3] = 1.f;
    ha[8] = 1.f + ldexpf(1.f, -23); hb[8] = 1.f + ldexpf(1.f, -23); ha[9] = 0.f; hb[9] = 0.f; hc[4] = -(1.f + ldexpf(1.f, -22));
    float *da, *db, *dc, *dout;
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dc, sizeof(hc)); hipMalloc(&dout, sizeof(ho));
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipMemcpy(dc, hc, sizeof(hc), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dc, dout, n);
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    for (int k = 0; k < n; ++k) printf("case %d: result %.10e  (result-1)/2^-24 = %.4f\n", k, ho[k], (ho[k] - 1.f) / t24);
    printf("case 4 exact 2^-46 = %.10e\n", ldexp(1.0, -46));
    return 0;
}
