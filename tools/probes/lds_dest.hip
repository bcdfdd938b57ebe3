// Diagnostic (not part of the product): what does a ds_read_b128 cost a wave that issues MFMAs back to back, and does it
// matter where the data lands?  mfma_store_mix.hip priced an A-operand read at ~13 cycles of matrix-pipe time even when it
// is issued a whole group of 16 MFMAs ahead - 16 B/clk per CU, nowhere near the LDS bandwidth - which smells of a register
// file port: an LDS return writes 4 registers x 64 lanes.  Variants (4 reads + 16 MFMAs per group, operands one group ahead):
//   V    destination ArchVGPRs (what the compiler emits for the kernels)
//   A    destination AGPRs (inline asm, "=a"), the MFMA takes its A operand from the AGPR
//   B64  eight ds_read_b64 instead of four b128 (same bytes)
//   N    no reads at all (operands constant)
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/lds_dest.hip -o gpurun_tools/lds_dest
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x16 acc[8];
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 1.f * (m + r);
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 256) smem[i] = 1.f + (i & 7);
    __syncthreads();
    const float b = 2.f;
    const unsigned base = (unsigned)(size_t)(smem) + lane * 16;        // LDS byte address of this lane's float4
    f32x4 nxt[4];
    for (int t = 0; t < 4; ++t) nxt[t] = f32x4{1.f, 2.f, 3.f, 4.f};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {                          // 8 groups of 16 MFMAs = one K block
            f32x4 a4[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) a4[t] = nxt[t];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j == 4 && MODE != 3) {
                    if (MODE == 0) {
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            nxt[t] = *reinterpret_cast<const f32x4*>(smem + (((g + 1) & 7) * 4 + t) * 256 + lane * 4);
                    } else if (MODE == 1) {
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(nxt[t]) : "v"(base), "n"(0) : "memory");
                    } else if (MODE == 4) {                  // the same instruction stream with ArchVGPR destinations
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(nxt[t]) : "v"(base), "n"(0) : "memory");
                    } else if (MODE == 2) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const f32x2 lo = *reinterpret_cast<const f32x2*>(smem + (((g + 1) & 7) * 4 + t) * 256 + lane * 2);
                            const f32x2 hi = *reinterpret_cast<const f32x2*>(smem + (((g + 1) & 7) * 4 + t) * 256 + 128 + lane * 2);
                            nxt[t] = f32x4{lo.x, lo.y, hi.x, hi.y};
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (j == 15 && (MODE == 1 || MODE == 4)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                acc[j & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j >> 2][j & 3], b, acc[j & 7], 0, 0, 0);
            }
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[256 * gridDim.x] = (float)(t1 - t0) / iters;
}

template <int MODE>
void run(float* d, const char* tag) {
    (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 65536, 0, d, 64);
        (void)hipDeviceSynchronize();
    }
    float cyc = 0;
    (void)hipMemcpy(&cyc, d + 256 * 256, 4, hipMemcpyDeviceToHost);
    printf("%-44s %8.0f cycles per K block of 128 MFMAs (+%5.0f over 8192, %5.1f per ds_read_b128-equivalent)\n", tag, cyc, cyc - 8192.f,
           (cyc - 8192.f) / 32.f);
}
int main() {
    float* d;
    (void)hipMalloc(&d, (256 * 256 + 16) * 4);
    run<3>(d, "N   no LDS reads");
    run<0>(d, "V   ds_read_b128 -> ArchVGPR");
    run<4>(d, "V'  ds_read_b128 -> ArchVGPR, inline asm like A");
    run<1>(d, "A   ds_read_b128 -> AGPR (MFMA srcA from AGPR)");
    run<2>(d, "B64 two ds_read_b64 per operand quad");
    return 0;
}
