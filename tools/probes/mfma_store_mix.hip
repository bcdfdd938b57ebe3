// Diagnostic (not part of the product): why do row stores cost ~190 cycles each inside the backward chain when
// mfma_store_pattern.hip prices them at 16-19?  This probe rebuilds the chain's K block around the MFMA stream one
// ingredient at a time:
//   ST   4 row stores  per row in rows 1-2 of a K block (odd slots; a slot = 4 MFMAs), [point][feature] rows
//   LD   4 row loads   per row in rows 1-2 (even slots)
//   BAR  s_waitcnt vmcnt(0) + s_barrier at the end of every K block (128 MFMAs), as the weight stage hand-over needs
//   DS   8 ds_read_b128 per 16 MFMAs feeding the A operands
//   DMA  8 x 1 KiB buffer_load ... lds per wave in row 0 (the next weight stage)
// 4 waves per CU, 256 CUs, 32 K blocks per wave.  Prints cycles per K block (8192 = pure MFMA issue).
// Findings (profiles/r02_probe_mfma_store_mix.log): 8 stores or 8 loads per K block alone cost 17 / 33 cycles apiece, both
// together 80-110 apiece wherever they sit in the K block (with every CU at that density: 4.9 TB/s of demand against
// 4.1-4.4 TB/s of mixed traffic delivered), half the density a quarter of that; nontemporal
// hints make it far worse; reading the first A operand of a group ahead of time recovers about half of the LDS cost.
// The combined variants overstate what the real kernels pay (their rows run at 2 200-2 400 cycles): use them for
// differences, not for absolute numbers.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_store_mix.hip -o gpurun_tools/mfma_store_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS_PTR __attribute__((address_space(3)))

enum { ST = 1, LD = 2, BAR = 4, DS = 8, DMA = 16, DSP = 32, DSP2 = 64 };

template <int F, int MODE = 0, int PH = 0>
__global__ __launch_bounds__(256, 1) void k(float* rows_out, const float* rows_in, const float* wts, float* out, int iters) {
    extern __shared__ float smem[];
    f32x16 acc[8];
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 1.f * (m + r);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += 256) smem[i] = 1.f + (i & 7);
    __syncthreads();
    const float b = 2.f;
    const size_t tile_floats = 8192;                            // 32 points x 256 features
    float* tile_o = rows_out + ((size_t)blockIdx.x * 4 + wave) * (size_t)iters * tile_floats;
    const float* tile_i = rows_in + ((size_t)blockIdx.x * 4 + wave) * (size_t)iters * tile_floats;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)wts, 0, 0x7fffffff, 0x00020000);
    const int voff = wave * 1024 + lane * 16;
    f32x4 v = {1.f, 2.f, 3.f, 4.f}, sink = {0.f, 0.f, 0.f, 0.f};
    const int row_off = (lane & 31) * 256 + (lane >> 5) * 4;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        float* to = tile_o + (size_t)it * tile_floats + row_off;
        const float* ti = tile_i + (size_t)it * tile_floats + row_off;
        const int q0 = (it & 3) * 8;                             // 32 quarters per layer = 4 K blocks x 8
        f32x4 got[8] = {};
        f32x4 nxt[4];
        if (F & DSP) for (int t = 0; t < 4; ++t) nxt[t] = *reinterpret_cast<const f32x4*>(smem + t * 256 + lane * 4);                                          // consumed after the K block, like the chain's saved rows
#pragma unroll
        for (int row = 0; row < 4; ++row) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {                        // 16 MFMAs per group
                f32x4 a4[4];
                if (F & DSP) {                                   // operands read one group ahead (8 MFMAs before the group ends)
#pragma unroll
                    for (int t = 0; t < 4; ++t) a4[t] = nxt[t];
                } else if (F & DS) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        a4[t] = *reinterpret_cast<const f32x4*>(smem + ((row * 2 + g) * 4 + t) * 256 + lane * 4);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) a4[t] = f32x4{1.f + lane, 2.f, 3.f, 4.f};
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if ((F & DSP) && j == 8) {
                        constexpr int NP = (F & DSP2) ? 1 : 4;   // DSP2: only the first operand ahead, the rest at the group start
#pragma unroll
                        for (int t = 0; t < NP; ++t)
                            nxt[t] = *reinterpret_cast<const f32x4*>(smem + (((row * 2 + g + 1) & 7) * 4 + t) * 256 + lane * 4);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if ((F & DSP2) && j == 0) {
#pragma unroll
                        for (int t = 1; t < 4; ++t)
                            a4[t] = *reinterpret_cast<const f32x4*>(smem + ((row * 2 + g) * 4 + t) * 256 + lane * 4);
                    }
                    acc[j & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j >> 2][j & 3], b, acc[j & 7], 0, 0, 0);
                    if ((j & 3) == 3) {
                        const int slot = g * 4 + (j >> 2);       // 0..7 within the row
                        __builtin_amdgcn_sched_barrier(0);
                        // PH 0: 8 + 8 row instructions per K block; 3: half that density (the same bytes over twice the time)
                        const bool on = PH == 0 ? true : slot < 4;
                        if ((row == 1 || row == 2) && on) {
                            // MODE 0: load / store alternate; 1: all loads in row 1, all stores in row 2; 2: per row, four loads
                            // then four stores; 3: as 0 with nontemporal stores; 4: as 0 with nontemporal loads and stores
                            bool ld, st; int q;
                            if (MODE == 1) { ld = row == 1; st = row == 2; q = q0 + slot; }
                            else if (MODE == 2) { ld = slot < 4; st = slot >= 4; q = q0 + (row - 1) * 4 + (slot & 3); }
                            else { ld = (slot & 1) == 0; st = (slot & 1) == 1; q = q0 + (row - 1) * 4 + slot / 2; }
                            if (MODE == 5) {   // tile layout [quarter][point][8 features]: every instruction moves one contiguous KiB
                                const int off = q * 256 + ((lane & 31) * 2 + (lane >> 5)) * 4 - row_off - q * 8;
                                if ((F & LD) && ld) got[q - q0] = *reinterpret_cast<const f32x4*>(ti + q * 8 + off);
                                if ((F & ST) && st) *reinterpret_cast<f32x4*>(to + q * 8 + off) = v;
                            } else
                            if ((F & LD) && ld) {
                                if (MODE == 4) got[q - q0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ti + q * 8));
                                else got[q - q0] = *reinterpret_cast<const f32x4*>(ti + q * 8);
                            }
                            if (MODE != 5 && (F & ST) && st) {
                                if (MODE >= 3) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(to + q * 8));
                                else *reinterpret_cast<f32x4*>(to + q * 8) = v;
                            }
                        }
                        if ((F & DMA) && row == 0)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_PTR void*)(smem + 16384 + (slot * 4 + wave) * 256), 16, voff,
                                                                     ((it & 63) * 8 + slot) * 4096, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        if (F & LD) {
#pragma unroll
            for (int t = 0; t < 8; ++t) sink += got[t];       // (PH != 0: stale registers on idle K blocks; values are irrelevant)
        }
        if (F & BAR) __syncthreads();
    }
    const long long t1 = clock64();
    float s = sink.x + sink.y + sink.z + sink.w + smem[16384 + threadIdx.x];
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[256 * gridDim.x] = (float)(t1 - t0) / iters;
}

static float *g_o, *g_i, *g_w, *g_out;
template <int F, int MODE = 0, int PH = 0>
void run(const char* tag) {
    const int iters = 32;
    (void)hipFuncSetAttribute((const void*)k<F, MODE, PH>, hipFuncAttributeMaxDynamicSharedMemorySize, 24576 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<F, MODE, PH>), dim3(256), dim3(256), 24576 * 4, 0, g_o, g_i, g_w, g_out, iters);
        (void)hipDeviceSynchronize();
    }
    float cyc = 0;
    (void)hipMemcpy(&cyc, g_out + 256 * 256, 4, hipMemcpyDeviceToHost);
    printf("%-34s %8.0f cycles per K block (+%5.0f over 8192)\n", tag, cyc, cyc - 8192.f);
}
int main() {
    const size_t bytes = (size_t)256 * 4 * 32 * 8192 * 4;       // 1 GiB each way
    (void)hipMalloc(&g_o, bytes);
    (void)hipMalloc(&g_i, bytes);
    (void)hipMemset(g_o, 0, bytes);
    (void)hipMemset(g_i, 0, bytes);
    (void)hipMalloc(&g_w, 4 << 20);
    (void)hipMemset(g_w, 0, 4 << 20);
    (void)hipMalloc(&g_out, (256 * 256 + 16) * 4);
    run<0>("mfma only");
    run<ST>("ST");
    run<LD>("LD");
    run<ST | LD>("ST LD");
    run<ST | LD, 1>("ST LD, loads row 1 / stores row 2");
    run<ST | LD, 2>("ST LD, 4 loads then 4 stores per row");
    run<ST | LD, 5>("ST LD, contiguous KiB quarters");
    run<ST | LD | BAR, 5>("ST LD BAR, contiguous");
    run<DS | DMA | ST | LD | BAR, 5>("DS DMA ST LD BAR, contiguous");
    run<ST | LD, 0, 3>("ST LD, half density everywhere");
    run<DS | DMA | ST | LD | BAR, 0, 3>("chain, half density everywhere");
    run<BAR>("BAR");
    run<ST | BAR>("ST BAR");
    run<ST | LD | BAR>("ST LD BAR");
    run<DS>("DS");
    run<DS | DSP>("DS, operands one group ahead");
    run<DS | DSP | DSP2>("DS, first operand ahead only");
    run<DS | DSP | BAR>("DS ahead, BAR");
    run<DS | DSP | DMA | ST | LD | BAR, 0, 3>("chain half density, DS ahead");
    run<DS | ST>("DS ST");
    run<DS | BAR>("DS BAR");
    run<DS | ST | LD | BAR>("DS ST LD BAR");
    run<DMA | BAR>("DMA BAR");
    run<DMA | ST | BAR>("DMA ST BAR");
    run<DS | DMA | BAR>("DS DMA BAR");
    run<DS | DMA | ST | BAR>("DS DMA ST BAR");
    run<DS | DMA | LD | BAR>("DS DMA LD BAR");
    run<DS | DMA | ST | LD | BAR>("DS DMA ST LD BAR (the chain)");
    run<DS | DMA | ST | LD>("DS DMA ST LD, no barrier");
    return 0;
}
