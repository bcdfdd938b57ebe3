// Diagnostic (not part of the product): what does a row store cost next to a stream of MFMAs, by access pattern?
//   pattern 0: none; 1: contiguous (lane L -> 16 B at L*16: 1 KiB per instruction)
//   pattern 2: [point][feature] rows as the training kernels write them (lane (j, h) -> point j's row + 16 h: 64 pieces of 16 B,
//              32 B contiguous per point);  3: the same for loads;  4: point pairs on adjacent lanes (32 requests of 32 B)
// one store (load) per 8 MFMAs, 4 waves per CU, 256 CUs.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_store_pattern.hip -o gpurun_tools/mfma_store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(256, 1) void k(float* buf, float* out, int iters) {
    f32x16 acc[8];
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 1.f * (m + r);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float a = 1.f + lane, b = 2.f;
    // each wave owns a 32-point x 256-feature tile per iteration: 32 KiB; 64 quarters of 1 KiB
    float* tile = buf + ((size_t)blockIdx.x * 4 + wave) * (size_t)iters * 8192;
    f32x4 v = {1.f, 2.f, 3.f, 4.f}, sink = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        float* t = tile + (size_t)it * 8192;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            acc[j & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j & 7], 0, 0, 0);
            if ((j & 7) == 7) {
                const int qd = j >> 3;                         // 8 quarters per 64 MFMAs (the real kernels: 32 per 1024)
                __builtin_amdgcn_sched_barrier(0);
                if (PAT == 1) *reinterpret_cast<f32x4*>(t + qd * 256 + lane * 4) = v;
                if (PAT == 2) *reinterpret_cast<f32x4*>(t + (lane & 31) * 256 + qd * 8 + (lane >> 5) * 4) = v;
                if (PAT == 3) sink += *reinterpret_cast<const f32x4*>(t + (lane & 31) * 256 + qd * 8 + (lane >> 5) * 4);
                if (PAT == 4) *reinterpret_cast<f32x4*>(t + (lane >> 1) * 256 + qd * 8 + (lane & 1) * 4) = v;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const long long t1 = clock64();
    float s = sink.x + sink.y + sink.z + sink.w;
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[256 * gridDim.x] = (float)(t1 - t0) / (iters * 64.f);
}
template <int PAT>
void run(float* buf, float* out, const char* tag) {
    const int iters = 64;
    hipLaunchKernelGGL((k<PAT>), dim3(256), dim3(256), 0, 0, buf, out, iters);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((k<PAT>), dim3(256), dim3(256), 0, 0, buf, out, iters);
    (void)hipDeviceSynchronize();
    float cyc = 0;
    (void)hipMemcpy(&cyc, out + 256 * 256, 4, hipMemcpyDeviceToHost);
    printf("%-44s %.1f cycles per MFMA (one memory instruction per 8 MFMAs: %.0f cycles each over the 64 baseline)\n", tag, cyc, (cyc - 64.2) * 8);
}
int main() {
    float *buf, *out;
    (void)hipMalloc(&buf, (size_t)256 * 4 * 64 * 8192 * 4);     // 2 GiB
    (void)hipMemset(buf, 0, (size_t)256 * 4 * 64 * 8192 * 4);
    (void)hipMalloc(&out, (256 * 256 + 16) * 4);
    run<0>(buf, out, "no memory traffic");
    run<1>(buf, out, "store, 1 KiB contiguous per instruction");
    run<2>(buf, out, "store, [point][feature] rows (64 x 16 B)");
    run<4>(buf, out, "store, rows with point pairs on adjacent lanes");
    run<3>(buf, out, "load,  [point][feature] rows (64 x 16 B)");
    return 0;
}
