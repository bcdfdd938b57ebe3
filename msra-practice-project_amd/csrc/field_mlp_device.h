// field_mlp_device.h - device building blocks shared by the fused field-MLP forward
// (field_mlp.hip) and backward-chain (field_mlp_bwd.hip) kernels: LDS map, DMA stages, the 32-wide
// K-block MFMA step, register<->row-major activation I/O.  See field_mlp.hip for the design.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "field_layout.h"
#include "mi_common.h"
#include "mi_math.h"

namespace mi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MI_LDS __attribute__((address_space(3)))
#define MI_GLB __attribute__((address_space(1)))
typedef const MI_LDS f32x4* lds4_t;

// Per-lane LDS base made opaque to the optimiser.  The aux / film regions sit above 64 KiB, out
// of reach of the 16-bit ds_read immediate from LDS address 0; without this hipcc materialises
// one address VGPR per read (hundreds, spilled to scratch).  With it: one VGPR + immediates.
__device__ __forceinline__ lds4_t lds_base(const float* p) {
    lds4_t q = (lds4_t)(const MI_LDS float*)p;
    asm volatile("" : "+v"(q));
    return q;
}

// ---- LDS map (floats) ----------------------------------------------------------------------
constexpr int kLdsChunk = 16384;                         // 64 KiB stage buffer = two 32-wide K blocks (MB = 8)
constexpr int kLdsAux = kMaxAuxPieces * kPiece;          // 8 KiB per-layer aux slot
constexpr int kLdsChunk0 = 0;
constexpr int kLdsAux0 = 2 * kLdsChunk;
constexpr int kLdsFilm0 = kLdsAux0 + 2 * kLdsAux;
constexpr int kLdsFloats = kLdsFilm0 + 2 * kFilmRow;     // 37888 floats = 148 KiB of the CU's 160 KiB

enum Act : int { ACT_LINEAR = 0, ACT_RELU = 1, ACT_SIN30 = 2, ACT_FILM = 3 };

struct Ctx {
    float* smem;
    __amdgpu_buffer_rsrc_t rsrc;    // buffer descriptor of the packed weight stream
    __amdgpu_buffer_rsrc_t frsrc;   // ... of this group's FiLM table [9][512] (FiLM kinds)
    int soff;                       // byte offset of the next unread piece of the stream (wave-uniform, SGPR)
    int buf;                        // stage buffer (0/1) holding the next stage to consume
    int voff;                       // per-lane byte offset inside a round of 4 pieces: wave * 1024 + lane * 16
    int lane, wave, h;
    float w0, w0sq;                 // FiLM kinds: the layers' frequency w_0 and fl(w_0^2) from the packed stream's trailer (uniform)
#ifdef MI_PROFILE_STAMPS
    unsigned long long* rowst;    // diagnostic build: next per-row stamp slot of this block (null = off)
#endif
};

#ifdef MI_PROFILE_STAMPS
// diagnostic build: one s_memtime per row of MFMAs (4 per K block) while c.rowst is set
#define MI_ROW_STAMP(c) do { if ((c).rowst) { if (threadIdx.x == 0) *(c).rowst = __builtin_amdgcn_s_memtime(); ++(c).rowst; } } while (0)
#else
#define MI_ROW_STAMP(c) do { } while (0)
#endif

// LDS-DMA of 1 KiB pieces (buffer_load_dwordx4 ... lds): per piece one SALU add for M0 and one VMEM issue -
// SGPR soffset selects the piece, the VGPR offset is constant - so a piece can be slipped between two MFMAs.
__device__ __forceinline__ void dma_piece(const Ctx& c, __amdgpu_buffer_rsrc_t r, int lds_float_off, int soff_bytes) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (MI_LDS void*)(c.smem + lds_float_off), 16, c.voff, soff_bytes, 0, 0);
}

// N consecutive pieces starting at stream offset `soff` -> LDS float offset lds_off, split over the 4 waves.
template <int N>
__device__ __forceinline__ void dma_lump(const Ctx& c, __amdgpu_buffer_rsrc_t r, int soff, int lds_off) {
#pragma unroll
    for (int t0 = 0; t0 < N; t0 += 4)
        if (t0 + 4 <= N || t0 + c.wave < N) dma_piece(c, r, lds_off + (t0 + c.wave) * kPiece, soff + t0 * 1024);
}

// Issue the DMA of one stage in one go: optional aux pieces (+ FiLM row) of a layer, then one K block.
template <int N_AUX, int N_CHUNK_PIECES, bool FILM>
__device__ __forceinline__ void issue_stage(Ctx& c, int aux_slot, int chunk_buf, int film_layer) {
    if constexpr (N_AUX > 0) {
        dma_lump<N_AUX>(c, c.rsrc, c.soff, kLdsAux0 + aux_slot * kLdsAux);
        c.soff += N_AUX * 1024;
        if constexpr (FILM) dma_lump<2>(c, c.frsrc, film_layer * (kFilmRow * 4), kLdsFilm0 + aux_slot * kFilmRow);
    }
    if constexpr (N_CHUNK_PIECES > 0) {
        dma_lump<N_CHUNK_PIECES>(c, c.rsrc, c.soff, kLdsChunk0 + chunk_buf * kLdsChunk);
        c.soff += N_CHUNK_PIECES * 1024;
    }
}

// First stage of a layer, issued in one go (kernel start, or after a VALU-only first layer): its aux pieces and
// its first TWO K blocks (every MFMA layer has at least two) into the current stage buffer.
template <int N_AUX, int BLOCK_PIECES, bool FILM>
__device__ __forceinline__ void issue_first_stage(Ctx& c, int aux_slot, int /*unused*/, int film_layer) {
    issue_stage<N_AUX, 2 * BLOCK_PIECES, FILM>(c, aux_slot, c.buf, film_layer);
}

// The same stage issued piecewise from NSLOT call sites spread over a K block's first row of MFMAs:
// slot 0 also issues the (rare) aux pieces; slot s issues this wave's chunk pieces s, s + NSLOT, ...
template <int N_AUX, int N_CHUNK_PIECES, bool FILM, int NSLOT, int S>
__device__ __forceinline__ void issue_stage_slot(Ctx& c, int aux_slot, int chunk_buf, int film_layer) {
    if constexpr (S == 0 && N_AUX > 0) {
        dma_lump<N_AUX>(c, c.rsrc, c.soff, kLdsAux0 + aux_slot * kLdsAux);
        c.soff += N_AUX * 1024;
    }
    if constexpr (S == 0 && FILM) dma_lump<2>(c, c.frsrc, film_layer * (kFilmRow * 4), kLdsFilm0 + aux_slot * kFilmRow);
    constexpr int NP = N_CHUNK_PIECES / 4;        // pieces per wave
    static_assert(N_CHUNK_PIECES % 4 == 0, "K blocks are whole rounds of 4 pieces");
#pragma unroll
    for (int j = S; j < NP; j += NSLOT)
        dma_piece(c, c.rsrc, kLdsChunk0 + chunk_buf * kLdsChunk + (4 * j + c.wave) * kPiece, c.soff + 4096 * j);
    if constexpr (S == NSLOT - 1) c.soff += N_CHUNK_PIECES * 1024;
}

// One row of fragments (all MB output blocks of one rg) is read first, then its 4*MB MFMAs are issued.
// Nothing overlaps an MFMA on this SIMD (tools/probes/mfma_valu_overlap.hip: every VALU / LDS / VMEM instruction between
// two MFMAs adds its own issue time to the 64 cycles of the MFMA, with one wave per SIMD or with two), so the hooks
// below do not hide work - they ORDER it: no lump delays a DMA piece or bunches row stores, and each piece of work sits
// where its operands are ready:
//   * slot(s), s = 0..MB-1: after every 4th MFMA of the first row - the next stage's DMA pieces;
//   * ORDER bit 0 (a layer's first K block): the first row runs m-major and pre(m+1, part) - the start value of the
//     NEXT accumulator block, a quarter at a time (K = 3 products; the backward chain's scaled head row) - sits after
//     each MFMA of block m's chain;  ORDER bit 3: the chains start at zero, the first MFMA takes srcC = 0;
//   * ORDER bit 1 (a layer's last K block): the last row runs m-major and post(m-1, part) - bias, activation and
//     training stores of the block whose chain has just completed - follows one MFMA behind.
// Other rows run q-major (MB independent accumulators round-robin); rows 1-3 offer mid(kb, s), s = 0..3MB-1, one
// slot per 4 MFMAs, for global loads / stores that must not arrive in a burst (the training kernels' row traffic:
// 32 x 1 KiB per wave in one 2048-cycle row saturates the CU's 64 B/clk vector-memory path).
struct NoHook {
    template <class A, class B> __device__ __forceinline__ void operator()(A, B) const {}
    template <class A> __device__ __forceinline__ void operator()(A) const {}
};
template <int I> using ic = std::integral_constant<int, I>;

template <int MB, int ORDER, int KBI = 0, class Slot, class Pre, class Post, class Mid = NoHook>
__device__ __forceinline__ void mma_chunk(Ctx& c, const float* chunk, const f32x16& b, f32x16 (&acc)[8], Slot slot,
                                          Pre pre, Post post, Mid mid = Mid{}) {
    const f32x4* a4 = reinterpret_cast<const f32x4*>(chunk) + c.lane;
    // Rows whose hooks are fenced with sched_barriers keep the compiler from starting the next row's operand reads
    // early, and the row's first MFMA then waits a full LDS latency (tools/probes/mfma_store_mix.hip: ~90 cycles per
    // 2048-cycle row).  Reading only the FIRST operand of the next row a few MFMAs ahead hides nearly all of it
    // for 4 registers; the other reads follow at the row start, behind that MFMA.
    f32x4 a_first = a4[0];
    if constexpr (ORDER & 16) {
        // A layer's LAST K block run m-major over all four rows (ORDER bit 4): block m's chain completes after its own 16
        // MFMAs, and post(m - 1, part) rides behind every 4th MFMA of block m - the epilogue is spread over the whole K
        // block (128 MFMAs) instead of its last row (32).  For epilogues that STORE a row quarter per part (the training
        // forward of the sin layers, whose encoded rows cannot be deferred to the next layer) that is one 1 KiB store per
        // 256 cycles and wave instead of one per 64: 32 of them inside one 2 048-cycle row ask for the CU's whole
        // 64 B/clk vector-memory path.  Same MFMAs, same A reads (4 per block), same rounding (a chain is still summed in
        // k order).
        static_for<MB>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            f32x4 a[4];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) a[rg] = (m == 0 && rg == 0) ? a_first : a4[(rg * MB + m) * 64];
            static_for<4>([&](auto rgc) {
                constexpr int rg = decltype(rgc)::value;
                static_for<4>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rg][q], b[4 * rg + q], acc[m], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (m >= 1 && q == 1) post(ic<(m >= 1 ? m - 1 : 0)>{}, rgc);
                    if constexpr (rg == 0 && q == 3) slot(mc);
                    // the 2 MB mid slots a layer's last K block offers (the previous layer's deferred row quarters)
                    if constexpr (!std::is_same<Mid, NoHook>::value && (rg & 1) == 1 && q == 3) mid(ic<KBI>{}, ic<2 * m + rg / 2>{});
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        });
        static_for<4>([&](auto pc) { post(ic<MB - 1>{}, pc); });
        return;
    }
    static_for<4>([&](auto rgc) {
        constexpr int rg = decltype(rgc)::value;
        MI_ROW_STAMP(c);
        f32x4 a[MB];
        a[0] = a_first;
#pragma unroll
        for (int m = 1; m < MB; ++m) a[m] = a4[(rg * MB + m) * 64];
        constexpr bool first_mm = (ORDER & 1) && rg == 0;
        constexpr bool last_mm = (ORDER & 2) && rg == 3;
        const auto ahead = [&]() { if constexpr (rg < 3) a_first = a4[((rg + 1) * MB) * 64]; };
        if constexpr (first_mm || last_mm) {
            if constexpr (first_mm) static_for<4>([&](auto pc) { pre(ic<0>{}, pc); });
            static_for<MB>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                if constexpr (m == (MB >= 2 ? MB - 2 : 0)) ahead();
                static_for<4>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    if constexpr (first_mm && q == 0 && (ORDER & 8)) {
                        // the chain starts at zero: srcC = 0 is an inline constant, no accumulator to initialise
                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][q], b[4 * rg + q], zero, 0, 0, 0);
                    } else {
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][q], b[4 * rg + q], acc[m], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (first_mm && m + 1 < MB) pre(ic<m + 1>{}, qc);
                    if constexpr (last_mm) {
                        if constexpr (q >= 1 && m >= 1) post(ic<(m >= 1 ? m - 1 : 0)>{}, ic<(q >= 1 ? q - 1 : 0)>{});
                        if constexpr (q == 0 && m >= 2) post(ic<(m >= 2 ? m - 2 : 0)>{}, ic<3>{});
                    }
                    if constexpr (rg == 0 && q == 3) slot(ic<m>{});
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            if constexpr (last_mm) {
                if constexpr (MB >= 2) post(ic<(MB >= 2 ? MB - 2 : 0)>{}, ic<3>{});
                static_for<4>([&](auto pc) { post(ic<MB - 1>{}, pc); });
            }
        } else {
            static_for<4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                static_for<MB / 4>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    if constexpr (q == 3 && g == 0) ahead();
#pragma unroll
                    for (int m = 4 * g; m < 4 * g + 4; ++m)
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][q], b[4 * rg + q], acc[m], 0, 0, 0);
                    if constexpr (rg == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        slot(ic<q * (MB / 4) + g>{});
                        __builtin_amdgcn_sched_barrier(0);
                    } else if constexpr (!std::is_same<Mid, NoHook>::value) {
                        __builtin_amdgcn_sched_barrier(0);
                        mid(ic<KBI>{}, ic<(rg - 1) * 4 * (MB / 4) + q * (MB / 4) + g>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            });
        }
    });
}

// Rounding order of a linear layer.  The reference's x W^T + b (torch F.linear on MKL) is an FMA chain over k
// starting from ZERO with the bias added to the finished sum; measured against fp64, preloading the bias into the
// accumulator instead (one add less) is 1.8x less accurate for the SIREN-family layers, whose pre-activations
// (|u| ~ 0.03) are smaller than their biases (|b| <= 0.06): every partial sum then rounds at the bias's ulp.
// v_mfma_f32_32x32x2_f32 is two chained IEEE FMAs (tools/probes/mfma_rounding.hip), so starting the accumulators
// at zero (or at the K = 3 products, the first columns of the chain) and adding the bias in the epilogue gives the
// reference's error, not 1.8x it.

// acc = W3 * xyz for the K = 3 inputs of Siren/FiLM nets (zero otherwise); the bias is added by the epilogue.
template <int MB, bool K3>
__device__ __forceinline__ void init_acc(const float* aux, int h, int k3_piece, float x, float y, float z,
                                         f32x16 (&acc)[8]) {
    const lds4_t pb = lds_base(aux + h * 16);       // VEC piece entry ((m*2+h)*4 + rg) in f32x4 units
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if constexpr (K3) {
                const f32x4 w0 = pb[(k3_piece + 0) * 64 + m * 8 + rg];
                const f32x4 w1 = pb[(k3_piece + 1) * 64 + m * 8 + rg];
                const f32x4 w2 = pb[(k3_piece + 2) * 64 + m * 8 + rg];
                t.x = fmaf(w2.x, z, fmaf(w1.x, y, w0.x * x));
                t.y = fmaf(w2.y, z, fmaf(w1.y, y, w0.y * x));
                t.z = fmaf(w2.z, z, fmaf(w1.z, y, w0.z * x));
                t.w = fmaf(w2.w, z, fmaf(w1.w, y, w0.w * x));
            }
            acc[m][4 * rg + 0] = t.x; acc[m][4 * rg + 1] = t.y; acc[m][4 * rg + 2] = t.z; acc[m][4 * rg + 3] = t.w;
        }
    }
}

// Activation epilogue: X = act(acc + bias).  FiLM reads gamma|beta of this layer from the film slot.
template <int MB, int ACT>
__device__ __forceinline__ void activate(const f32x16 (&acc)[8], f32x16 (&X)[8], const float* film_row, int h,
                                         const float* aux, float w0 = 30.f) {
    const lds4_t pb = lds_base(aux + h * 16);
    lds4_t pf = nullptr;
    if constexpr (ACT == ACT_FILM) pf = lds_base(film_row + h * 4);   // gamma at f, beta at 256 + f
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            f32x4 g, b;
            const f32x4 bias = pb[m * 8 + rg];
            if constexpr (ACT == ACT_FILM) {
                g = pf[m * 8 + rg * 2];
                b = pf[64 + m * 8 + rg * 2];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = __fadd_rn(acc[m][4 * rg + q], bias[q]);
                float o;
                if constexpr (ACT == ACT_RELU) o = fmaxf(v, 0.f);
                else if constexpr (ACT == ACT_SIN30) o = hw_sin30(v);
                else if constexpr (ACT == ACT_FILM) o = hw_sin_w(__fadd_rn(__fmul_rn(g[q], v), b[q]), w0);
                else o = v;
                X[m][4 * rg + q] = o;
            }
        }
    }
}

// Training flavour of the sin activations: besides X = sin(30 u) it stores, as [point][feature] rows, X with the
// sign of cos(30 u) in its lowest mantissa bit (mi_math.h:hw_sin30_saved) - all the backward needs of the layer.
template <int MB, int ACT>
__device__ __forceinline__ void activate_train(const f32x16 (&acc)[8], f32x16 (&X)[8], const float* film_row, int h,
                                               const float* aux, float* __restrict__ x_rows, int64_t ld, int64_t p,
                                               bool valid, float w0 = 30.f) {
    static_assert(ACT == ACT_SIN30 || ACT == ACT_FILM, "sin activations only");
    const lds4_t pb = lds_base(aux + h * 16);
    lds4_t pf = nullptr;
    if constexpr (ACT == ACT_FILM) pf = lds_base(film_row + h * 4);
    f32x4* xrow = reinterpret_cast<f32x4*>(x_rows + p * ld + 4 * h);
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            f32x4 g, b, xo;
            const f32x4 bias = pb[m * 8 + rg];
            if constexpr (ACT == ACT_FILM) {
                g = pf[m * 8 + rg * 2];
                b = pf[64 + m * 8 + rg * 2];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = __fadd_rn(acc[m][4 * rg + q], bias[q]);
                float u = v;
                if constexpr (ACT == ACT_FILM) u = __fadd_rn(__fmul_rn(g[q], v), b[q]);
                const SinSaved sc = hw_sin_w_saved(u, w0);
                X[m][4 * rg + q] = sc.s;
                xo[q] = sc.saved;
            }
            xrow[m * 8 + rg * 2] = xo;                   // unguarded on purpose: see fwd_layer
        }
    }
}

// One MFMA layer: KB K blocks; bsel(kb) yields the B-operand register block of K block kb.
// On entry the layer's first stage (aux + K block 0) has been issued into aux slot
// `aux_slot` and chunk buffer PAR0.  NEXT_* describe the stage to issue while the last K block
// computes (the next layer's first stage), 0/0 for none.
// A stage = up to two K blocks behind one barrier (39 barriers per NeRF tile instead of 77).  Stage buffers
// alternate (c.buf); while stage i computes, stage i+1's pieces (or the next layer's aux pieces + its first
// two K blocks: NEXT_AUX, NEXT_BLOCK = pieces of one of its K blocks, 0 for none) are DMA'd into the other one.
// HOOKS: pre(m, part) / post(m, part) are the sliced accumulator start / epilogue of mma_chunk; without them
// init(acc) runs as one lump before the first MFMA and the caller applies its epilogue after the call.
template <int KB, int MB, int PAR0_UNUSED, int NEXT_AUX, int NEXT_BLOCK, bool FILM, bool HOOKS = false, bool ZERO_START = false,
          bool SPREAD_LAST = false, class Init, class BSel, class Pre = NoHook, class Post = NoHook, class Mid = NoHook>
__device__ __forceinline__ void mma_layer_fn(Ctx& c, int aux_slot, int next_film_layer, Init init, BSel bsel,
                                             f32x16 (&acc)[8], Pre pre = Pre{}, Post post = Post{}, Mid mid = Mid{}) {
    static_assert(KB >= 2, "every MFMA layer has at least two K blocks");
    auto stage = [&](auto ic_) {
        constexpr int i = decltype(ic_)::value;
        constexpr int kb0 = 2 * i;
        constexpr bool two = kb0 + 1 < KB;
        constexpr int left = KB - 2 * (i + 1);                       // K blocks after this stage
        constexpr int next_blocks = left >= 2 ? 2 : (left > 0 ? left : 0);
        constexpr int zs = ZERO_START ? 8 : 0;         // ORDER bit 3: the layer's first MFMAs take srcC = 0
        constexpr int last = SPREAD_LAST ? 16 : 2;      // the last K block: epilogue in its last row, or spread over all of it
        constexpr int order0 = HOOKS ? ((kb0 == 0 ? 1 | zs : 0) | (kb0 == KB - 1 ? last : 0)) : 0;
        constexpr int order1 = HOOKS ? (kb0 + 1 == KB - 1 ? last : 0) : 0;
        __syncthreads();
        if constexpr (i == 0 && !HOOKS) init(acc);
        const float* buf = c.smem + kLdsChunk0 + c.buf * kLdsChunk;
        mma_chunk<MB, order0, kb0>(c, buf, bsel(ic<kb0>{}), acc, [&](auto sc) {
            constexpr int S = decltype(sc)::value;
            if constexpr (next_blocks > 0) issue_stage_slot<0, next_blocks * MB * 4, false, MB, S>(c, 0, c.buf ^ 1, 0);
            else issue_stage_slot<NEXT_AUX, 2 * NEXT_BLOCK, FILM, MB, S>(c, aux_slot ^ 1, c.buf ^ 1, next_film_layer);
        }, pre, post, mid);
        if constexpr (two) mma_chunk<MB, order1, kb0 + 1>(c, buf + MB * 1024, bsel(ic<kb0 + 1>{}), acc, NoHook{}, pre, post, mid);
        c.buf ^= 1;
    };
    static_for<(KB + 1) / 2>(stage);
}

// FiLM's gamma * A + beta (pi_GAN/modules.py:24): a rounded product, then a rounded sum, four elements at a time
__device__ __forceinline__ f32x4 film_affine(f32x4 g, f32x4 v, f32x4 b) {
#pragma clang fp contract(off)
    return g * v + b;
}

// Rows the training forward stores per layer: X = the layer's activation, [point][ld] row-major (sin layers: with
// the cosine's sign in the lowest mantissa bit, mi_math.h).
struct SaveRows {
    float* x;
    int64_t ld, p;
    bool valid;
    float* mask = nullptr;      // ReLU layers: the region of the layer's switch bits (field_layout.h nerf_acts()), or null
};

// A ReLU layer's switches for the backward chain, one bit per unit, shifted into a lane's words as the epilogue produces the
// units (posts run in ascending (block m, quarter rg) order in every mma_chunk ordering, elements q = 0..3 inside a post):
// word m >> 1 receives 32 bits, unit (m, rg, q) ends at bit 31 - (16 (m & 1) + 4 rg + q).  v_cmp + v_addc: two VALU
// instructions per unit, exactly torch's relu' = [output > 0].
__device__ __forceinline__ void relu_switch_in(uint32_t& w, float o) {
    asm("v_cmp_lt_f32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(w) : "v"(o) : "vcc");
}
// ... and read back: 0xffffffff if the unit was on, else 0 (v_bfe_i32), to be ANDed onto the incoming gradient
template <int M, int RG, int Q>
__device__ __forceinline__ uint32_t relu_switch_of(const uint32_t (&w)[4]) {
    return (uint32_t)__builtin_amdgcn_sbfe(w[M >> 1], 31 - (16 * (M & 1) + 4 * RG + Q), 1);
}

// One forward layer with its accumulator start (K = 3 products, else none: srcC = 0) and its epilogue (bias,
// activation, training stores) sliced between the MFMAs of its first / last K block.  X <- act(W . [inputs] + b); the
// layer's inputs may be X itself (in place: block m of X is rewritten only after every K block that reads it has been
// consumed).
// Training stores do not burst either: with DEFER_X this layer's X rows are written by the NEXT layer's mid slots
// (slot 4(j%4)+1 of K block j/4 for quarter j when that layer has 8 K blocks, else 2(j%PPC)+1 of K block j/PPC; X is
// that layer's B operand and unchanged until its last row) - that
// layer receives them as `prev` (PREV_MB blocks).  Sin layers save an ENCODING of X (cosine sign in the lowest bit)
// that is not what the registers carry on, so their rows are stored by the activation hook itself, one quarter per
// hook.
template <int KB, int MB, bool K3, int NEXT_AUX, int NEXT_BLOCK, bool FILM, int ACT, bool SAVE, bool DEFER_X = false,
          int PREV_MB = 0, class BSel>
__device__ __forceinline__ void fwd_layer(Ctx& c, int aux_slot, int next_film_layer, int k3_piece, float x, float y,
                                          float z, BSel bsel, f32x16 (&acc)[8], f32x16 (&X)[8], const float* film_row,
                                          const SaveRows& sv, const SaveRows& prev = SaveRows{nullptr, 0, 0, false}) {
    const int h = c.h;
    const lds4_t pb = lds_base(c.smem + kLdsAux0 + aux_slot * kLdsAux + h * 16);
    lds4_t pf = nullptr;
    if constexpr (ACT == ACT_FILM) pf = lds_base(film_row + h * 4);
    f32x4 bias_q[2], g_q[2], bb_q[2];
    uint32_t mw[4] = {0u, 0u, 0u, 0u};           // ReLU + SAVE: the layer's switch bits (relu_switch_in)
    constexpr bool kSinAct = ACT == ACT_SIN30 || ACT == ACT_FILM;
    // epilogues that store their own rows (training forward of a sin layer: the saved rows are an ENCODING of X that the
    // registers do not carry on, so they cannot be deferred to the next layer's slots) are spread over the whole last K
    // block (mma_chunk ORDER bit 4); K = 3 layers start their accumulators in that same K block's first hooks - not them
    constexpr bool kSpread = SAVE && kSinAct;
    const auto pre = [&](auto mc, auto pc) {
        constexpr int m = decltype(mc)::value, rg = decltype(pc)::value;
        // The chain starts at zero and post adds the bias (see init_acc).  Plain layers need nothing here: their first
        // MFMAs take srcC = 0 (mma_chunk ORDER bit 3).  K = 3 layers start from the three products.
        if constexpr (K3) {
            const f32x4 w0 = pb[(k3_piece + 0) * 64 + m * 8 + rg];
            const f32x4 w1 = pb[(k3_piece + 1) * 64 + m * 8 + rg];
            const f32x4 w2 = pb[(k3_piece + 2) * 64 + m * 8 + rg];
            acc[m][4 * rg + 0] = fmaf(w2.x, z, fmaf(w1.x, y, w0.x * x));
            acc[m][4 * rg + 1] = fmaf(w2.y, z, fmaf(w1.y, y, w0.y * x));
            acc[m][4 * rg + 2] = fmaf(w2.z, z, fmaf(w1.z, y, w0.z * x));
            acc[m][4 * rg + 3] = fmaf(w2.w, z, fmaf(w1.w, y, w0.w * x));
        }
        if constexpr (m == 0 && rg == 0) {           // first hook after the layer's first barrier: the aux slot has landed
            bias_q[0] = pb[0];
            if constexpr (ACT == ACT_FILM) { g_q[0] = pf[0]; bb_q[0] = pf[64]; }
        }
    };
    // The epilogue's LDS operands (bias, FiLM gamma|beta) are read ONE HOOK AHEAD into a two-deep register queue: a
    // ds_read consumed in the hook that issues it puts its whole round trip (~100 cycles > one MFMA's 64) on the
    // wave's in-order issue path, between two MFMAs.
    // post(m, rg): one epilogue quarter = 4 registers, as two pairs on the packed fp32 ops (mi_math.h says why the
    // instruction count is what matters).  LDS operands come from the one-quarter-ahead queue.
    const auto post = [&](auto mc, auto pc) {
        constexpr int m = decltype(mc)::value, rg = decltype(pc)::value, idx = m * 4 + rg;
        if constexpr (idx + 1 < MB * 4) {
            constexpr int m1 = (idx + 1) / 4, rg1 = (idx + 1) % 4;
            bias_q[(idx + 1) & 1] = pb[m1 * 8 + rg1];
            if constexpr (ACT == ACT_FILM) {
                g_q[(idx + 1) & 1] = pf[m1 * 8 + rg1 * 2];
                bb_q[(idx + 1) & 1] = pf[64 + m1 * 8 + rg1 * 2];
            }
        }
        const f32x4 bias = bias_q[idx & 1];
        // the whole quarter as ONE four-wide value: every step is two independent packed instructions back to back, so no
        // dependent packed op waits a state on its predecessor (mi_math.h:hw_turns30_x4)
        f32x4 v = f32x4{acc[m][4 * rg + 0], acc[m][4 * rg + 1], acc[m][4 * rg + 2], acc[m][4 * rg + 3]} + bias;
        f32x4 o, xo;
        if constexpr (ACT == ACT_RELU) {
            o = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
            xo = o;
            if constexpr (SAVE) {
                relu_switch_in(mw[m >> 1], o.x); relu_switch_in(mw[m >> 1], o.y);
                relu_switch_in(mw[m >> 1], o.z); relu_switch_in(mw[m >> 1], o.w);
            }
        }
        else if constexpr (ACT == ACT_LINEAR) { o = v; xo = o; }
        else {
            if constexpr (ACT == ACT_FILM) v = film_affine(g_q[idx & 1], v, bb_q[idx & 1]);
            // FiLM layers: w_0 is the module's (pi_GAN/modules.py:11,73), a wave-uniform scalar; Siren's is the literal 30
            float w0 = 30.f;
            if constexpr (ACT == ACT_FILM) w0 = c.w0;
            if constexpr (SAVE) {
                const SinSaved4 sc = hw_sin_w_saved_x4(v, w0);
                o = sc.s;
                xo = sc.saved;
            } else {
                o = hw_sin_w_x4(v, w0);
                xo = o;
            }
        }
        X[m][4 * rg + 0] = o.x; X[m][4 * rg + 1] = o.y; X[m][4 * rg + 2] = o.z; X[m][4 * rg + 3] = o.w;
        if constexpr (SAVE && (kSinAct || !DEFER_X)) {
            // float4 index m*8 + rg*2 inside the row (h folded into the row pointer)
            // NOT guarded by sv.valid: lanes past the end of a partial tile are clamped to its last point, compute
            // exactly what that point's own lane computes and store the same bytes to the same address - while a
            // guard costs a saveexec / branch / restore around every row store, and each of those branches cost the
            // in-order wave hundreds of cycles (stamped profile of the chain: 2048-cycle rows took up to 5400)
            reinterpret_cast<f32x4*>(sv.x + sv.p * sv.ld + 4 * h)[m * 8 + rg * 2] = xo;
        }
    };
    if constexpr (SAVE && PREV_MB > 0) {
        // a K block offers 3 * MB mid slots (2 * MB in the layer's last one): PPC quarters per K block on its odd slots
        // (as thinly as the layer's K blocks allow: every fourth slot when 4 quarters per K block are enough - dense row
        // traffic costs several times more per instruction, see tools/probes/mfma_store_mix.hip)
        constexpr int PPC = MB >= 8 && KB * 4 < PREV_MB * 4 ? 8 : 4;
        constexpr int STRIDE = MB >= 8 && PPC == 4 ? 4 : 2;
        static_assert(KB * PPC >= PREV_MB * 4, "not enough K blocks to carry the previous layer's row quarters");
        f32x4* prow = reinterpret_cast<f32x4*>(prev.x + prev.p * prev.ld + 4 * h);
        const auto mid = [&](auto kbc, auto sc) {
            constexpr int kb = decltype(kbc)::value, slot = decltype(sc)::value, j = kb * PPC + slot / STRIDE;
            if constexpr (slot < STRIDE * PPC && slot % STRIDE == 1 && j < PREV_MB * 4) {
                constexpr int m = j / 4, rg = j % 4;
                prow[m * 8 + rg * 2] = f32x4{X[m][4 * rg + 0], X[m][4 * rg + 1], X[m][4 * rg + 2], X[m][4 * rg + 3]};
            }
        };
        mma_layer_fn<KB, MB, 0, NEXT_AUX, NEXT_BLOCK, FILM, true, !K3, kSpread>(c, aux_slot, next_film_layer, NoHook{}, bsel, acc, pre, post, mid);
    } else {
        mma_layer_fn<KB, MB, 0, NEXT_AUX, NEXT_BLOCK, FILM, true, !K3, kSpread>(c, aux_slot, next_film_layer, NoHook{}, bsel, acc, pre, post);
    }
    if constexpr (ACT == ACT_RELU && SAVE) {
        // one 16-byte (MB = 8) or 8-byte (MB = 4) store per lane: the wave's 32 points x 2 halves are contiguous
        if (sv.mask) {
            uint32_t* dst = reinterpret_cast<uint32_t*>(sv.mask) + (sv.p * 2 + h) * (MB / 2);
            if constexpr (MB == 8) *reinterpret_cast<uint4*>(dst) = uint4{mw[0], mw[1], mw[2], mw[3]};
            else { static_assert(MB == 4, "ReLU layers are 256 or 128 wide"); *reinterpret_cast<uint2*>(dst) = uint2{mw[0], mw[1]}; }
        }
    }
}

// sigma / rgb heads: dot products over the features a lane holds + one cross-half add.
template <int MB>
__device__ __forceinline__ float head_dot(const f32x16 (&X)[8], const float* aux, int piece, int h) {
    const lds4_t p = lds_base(aux + h * 16);
    float s0 = 0.f, s1 = 0.f;
    // The weights of block m + 1 are read while block m's 16 FMAs run (left to itself the compiler keeps ONE read in
    // flight and waits for it after four FMAs: 32 half-hidden LDS latencies per head, tools/isa_lds_waits.py).
    f32x4 w[4], wn[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) w[rg] = p[piece * 64 + rg];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        if (m + 1 < MB) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) wn[rg] = p[piece * 64 + (m + 1) * 8 + rg];
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            s0 = fmaf(w[rg].x, X[m][4 * rg + 0], s0);
            s1 = fmaf(w[rg].y, X[m][4 * rg + 1], s1);
            s0 = fmaf(w[rg].z, X[m][4 * rg + 2], s0);
            s1 = fmaf(w[rg].w, X[m][4 * rg + 3], s1);
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) w[rg] = wn[rg];
    }
    float s = s0 + s1;
    s += __shfl_xor(s, 32);
    return s;
}

__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + expf(-v)); }

// Positional-encoding blocks, fully unrolled.  Feature f of the encoding of (x,y,z): i = f/6, c = f%6,
// c < 3 -> sin(2^i x_c), else cos(2^i x_{c-3}) (nerf/nerf.py:44-49).  Register r of block blk holds feature
// f0 = 32 blk + (r&3) + 8 (r>>2) on lane half 0 and f0 + 4 on half 1, so (frequency, component, sin|cos) are
// compile-time constants per half and only a select on h remains at run time.  One v_sin per feature:
// the argument goes to revolutions with a two-float product (hw_turns, as for the activations) and cos is sin
// shifted by 1/4 turn.
__device__ __forceinline__ float pe_feature(float xv, float scale, float quarter) {
#pragma clang fp contract(off)
    const float r = hw_turns(xv * scale) + quarter;                             // xv * scale exact: a power of two
    return __builtin_amdgcn_sinf(r - rintf(r));                                 // back to [-1/2, 1/2] after the 1/4 turn
}

template <int NBLK>
__device__ __forceinline__ void posenc_blocks(float* /*scr*/, int /*lane*/, int h, float x, float y, float z, int nfeat,
                                              f32x16* out) {
    const float xyz[3] = {x, y, z};
    static_for<NBLK * 16>([&](auto sc) {
        constexpr int slot = decltype(sc)::value;
        constexpr int r = slot & 15, blk = slot >> 4;
        constexpr int f0 = 32 * blk + (r & 3) + 8 * (r >> 2), f1 = f0 + 4;
        constexpr int i0 = f0 / 6, c0 = f0 % 6, i1 = f1 / 6, c1 = f1 % 6;
        const bool up = h != 0;
        const int f = up ? f1 : f0;
        const float xv = up ? xyz[c1 % 3] : xyz[c0 % 3];
        const float scale = up ? (float)(1 << i1) : (float)(1 << i0);
        const float quarter = up ? (c1 >= 3 ? 0.25f : 0.f) : (c0 >= 3 ? 0.25f : 0.f);
        out[blk][r] = f < nfeat ? pe_feature(xv, scale, quarter) : 0.f;
    });
}

struct PointIn {
    float px, py, pz, dx, dy, dz;
    int64_t p;      // global point index (row of the output)
    bool valid;
};

// mode 0: x[M,6] points; mode 1: rays[N,2,3] + z[N,S].
__device__ __forceinline__ PointIn load_point(int mode, const float* __restrict__ a, const float* __restrict__ zv,
                                               int64_t group, int64_t ppg, int64_t rpg, int S, int64_t local) {
    PointIn o;
    o.valid = local < ppg;
    const int64_t lc = o.valid ? local : ppg - 1;
    o.p = group * ppg + lc;
    if (mode == 0) {
        const float* r = a + o.p * 6;
        o.px = r[0]; o.py = r[1]; o.pz = r[2]; o.dx = r[3]; o.dy = r[4]; o.dz = r[5];
    } else {
        const int64_t ray = group * rpg + lc / S;
        const float* r = a + ray * 6;
        const float zz = zv[o.p];
        const float d0 = r[3], d1 = r[4], d2 = r[5];
        // render.py:134 pts = o + d*z (separate mul/add), :122 view = d / ||d||
        o.px = __fadd_rn(r[0], __fmul_rn(d0, zz));
        o.py = __fadd_rn(r[1], __fmul_rn(d1, zz));
        o.pz = __fadd_rn(r[2], __fmul_rn(d2, zz));
        const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2)));
        o.dx = d0 / nrm; o.dy = d1 / nrm; o.dz = d2 / nrm;
    }
    return o;
}

__device__ __forceinline__ Ctx make_ctx_raw(float* smem, const float* packed, const float* film_group) {
    Ctx c;
    c.smem = smem;
    c.lane = threadIdx.x & 63;
    c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.h = c.lane >> 5;
    // descriptors are built from wave-uniform values only; no bounds are relied on (num_records = 2 GiB)
    c.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)packed, 0, 0x7fffffff, 0x00020000);
    c.frsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(film_group ? film_group : packed), 0, 0x7fffffff, 0x00020000);
    c.soff = 0;
    c.buf = 0;
    c.voff = c.wave * 1024 + c.lane * 16;
    c.w0 = 30.f;
    c.w0sq = 900.f;
#ifdef MI_PROFILE_STAMPS
    c.rowst = nullptr;
#endif
    return c;
}

__device__ __forceinline__ Ctx make_ctx(float* smem, const MlpArgs& a, int64_t group) {
    return make_ctx_raw(smem, a.packed, a.film ? a.film + group * (kFilmLayers * kFilmRow) : nullptr);
}

__device__ __forceinline__ void store_out(const MlpArgs& a, const PointIn& pt, int h, float r, float g, float b,
                                          float s) {
    if (pt.valid && h == 0) reinterpret_cast<f32x4*>(a.out)[pt.p] = f32x4{r, g, b, s};
}


// ---- activations <-> HBM in [point][feature] row-major (training: saved layer inputs, gradients) --------
// Register r of block m on lane (col j, half h) is feature 32m + 8(r>>2) + 4h + (r&3) of point j, so each
// (m, rg) group of 4 registers is one aligned float4 of the point's row.
template <int MB>
__device__ __forceinline__ void store_rows(float* __restrict__ base, int64_t ld, int64_t p, bool valid, int h,
                                           const f32x16 (&X)[8]) {
    if (!valid) return;
    f32x4* row = reinterpret_cast<f32x4*>(base + p * ld + 4 * h);
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
            row[m * 8 + rg * 2] = f32x4{X[m][4 * rg + 0], X[m][4 * rg + 1], X[m][4 * rg + 2], X[m][4 * rg + 3]};
}


}  // namespace mi
