// Diagnostic (not part of the product): how many VALU instructions fit between two v_mfma_f32_32x32x2_f32 of ONE wave
// per SIMD before the matrix pipe starts to idle?  Kernel<N, DEP, KIND>: a loop of 64 MFMAs per iteration with N VALU
// instructions after each; DEP = 1: all MFMAs accumulate into one block (dependent chain), 0: eight blocks round-robin;
// KIND 0: v_fma_f32 on private VGPRs, 1: v_accvgpr_read of another block + v_add, 2: v_sin_f32.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap.hip -o gpurun_tools/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int N, int DEP, int KIND>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float seed) {
    f32x16 acc[8];
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = seed * (m + r);
    float a = seed + threadIdx.x, b = seed * 2.f;
    float t[8];
    for (int i = 0; i < 8; ++i) t[i] = seed * (i + 1) + threadIdx.x;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const int m = DEP ? 0 : (j & 7);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < N; ++v) {
                if (KIND == 0) t[v & 7] = __builtin_fmaf(t[v & 7], 1.0001f, 0.5f);
                else if (KIND == 3) t[0] = __builtin_fmaf(t[0], 1.0001f, 0.5f);                 // one dependent chain
                else if (KIND == 4) { typedef float f2 __attribute__((ext_vector_type(2)));      // packed, independent
                    f2 x = {t[(2 * v) & 7], t[(2 * v + 1) & 7]}; x = x * f2{1.0001f, 1.0002f} + f2{0.5f, 0.25f};
                    t[(2 * v) & 7] = x.x; t[(2 * v + 1) & 7] = x.y; }
                else if (KIND == 5) { typedef float f2 __attribute__((ext_vector_type(2)));      // packed, dependent
                    f2 x = {t[0], t[1]}; x = x * f2{1.0001f, 1.0002f} + f2{0.5f, 0.25f}; t[0] = x.x; t[1] = x.y; }
                else if (KIND == 1) t[v & 7] = t[v & 7] + acc[7][(v + j) & 15];
                else t[v & 7] = __builtin_amdgcn_sinf(t[v & 7]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    for (int i = 0; i < 8; ++i) s += t[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[256 * gridDim.x] = (float)(t1 - t0) / (iters * 64.f);
}

static int g_blocks = 256;       // 256: one wave per SIMD; 512: two (the kernel needs few registers, so two workgroups fit a CU)
template <int N, int DEP, int KIND>
void run(float* d, const char* tag) {
    const int blocks = g_blocks, iters = 200;
    hipLaunchKernelGGL((k<N, DEP, KIND>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<N, DEP, KIND>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0, cyc = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&cyc, d + 256 * blocks, 4, hipMemcpyDeviceToHost);
    printf("%-28s N=%2d  waves/SIMD %d  %.1f ns per MFMA of one wave, %.1f ns per MFMA per SIMD (clock64 ticks per MFMA of one wave %.1f)\n",
           tag, N, blocks / 256, ms * 1e6 / (iters * 64.0), ms * 1e6 / (iters * 64.0) / (blocks / 256), cyc);
}

int main(int argc, char** argv) {
    if (argc > 1) g_blocks = atoi(argv[1]);
    float* d;
    (void)hipMalloc(&d, (1024 * 256 + 16) * 4);
#define ROW(DEP, KIND, TAG) run<0, DEP, KIND>(d, TAG); run<2, DEP, KIND>(d, TAG); run<4, DEP, KIND>(d, TAG); run<6, DEP, KIND>(d, TAG); \
    run<8, DEP, KIND>(d, TAG); run<10, DEP, KIND>(d, TAG); run<12, DEP, KIND>(d, TAG); run<14, DEP, KIND>(d, TAG); run<16, DEP, KIND>(d, TAG); run<20, DEP, KIND>(d, TAG);
    ROW(0, 0, "8 independent v_fma chains")
    ROW(0, 3, "1 dependent v_fma chain")
    ROW(0, 4, "v_pk_fma, independent")
    ROW(0, 5, "v_pk_fma, dependent")
    ROW(0, 1, "accread + add")
    ROW(0, 2, "v_sin, 8 chains")
    return 0;
}
