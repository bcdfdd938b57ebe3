// field_mlp.hip - fused radiance-field MLP forward for gfx950 (MI355X), plus the weight packer.
//
// What it replaces: the `network(inputs[M,6]) -> [M,4]` call inside run_network
// (reference nerf/render.py:59-75) for NeRF / SirenNeRF (nerf/nerf.py:52-170), FilmSirenNeRF
// (pi_GAN/modules.py:70-118) and the build-defined TinyNeRF, including positional encoding
// (nerf/nerf.py:44-49), every Dense/Siren/FiLM layer, the sigma/rgb heads and - in ray mode -
// the point generation pts = o + d*z, view = d/|d| of render_rays (render.py:122,134).
//
// Structure (one workgroup = 4 waves = 128 points, one wave per SIMD, whole register file):
//   * features on MFMA rows, points on MFMA columns (v_mfma_f32_32x32x2_f32, exact fp32), so
//     the 8x16 accumulator registers of a 256-wide layer ARE the next layer's B operands:
//     activations never leave registers between layers;
//   * weights stream HBM/L2 -> LDS by global_load_lds_dwordx4 (1 KiB pieces) in the order the
//     MFMAs consume them (field_layout.h), double-buffered per 32-wide K block (32 KiB), one
//     barrier per K block; A fragments are ds_read_b128 (4 MFMAs per read);
//   * bias / head weights / K=3 input columns ride the same stream as 1 KiB VEC pieces into a
//     per-layer LDS aux slot; FiLM gamma|beta rows are DMA'd per image into a film slot;
//   * sigma / rgb heads (1 and 3 outputs) are VALU dot products + one cross-half shuffle.
// Roofline: fp32 MFMA (157 TFLOP/s); HBM traffic is 24-28 B in + 16 B out per point.
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

// ---- packed tables in constant memory (built at compile time) -----------------------------
__constant__ PackTable c_tab_nerf = build_nerf();
__constant__ PackTable c_tab_siren = build_siren_nerf();
__constant__ PackTable c_tab_film = build_film(true);
__constant__ PackTable c_tab_film_nodir = build_film(false);
__constant__ PackTable c_tab_tiny = build_tiny_nerf();

static constexpr PackTable h_tab_nerf = build_nerf();
static constexpr PackTable h_tab_siren = build_siren_nerf();
static constexpr PackTable h_tab_film = build_film(true);
static constexpr PackTable h_tab_film_nodir = build_film(false);
static constexpr PackTable h_tab_tiny = build_tiny_nerf();

const PackTable* host_table(int kind) {
    switch (kind) {
        case 0: return &h_tab_nerf;
        case 1: return &h_tab_siren;
        case 2: return &h_tab_film;
        case 3: return &h_tab_film_nodir;
        case 4: return &h_tab_tiny;
    }
    return nullptr;
}

struct ParamPtrs { const float* p[24]; };

__device__ __forceinline__ const PackTable& dev_table(int kind) {
    switch (kind) {
        case 0: return c_tab_nerf;
        case 1: return c_tab_siren;
        case 2: return c_tab_film;
        case 3: return c_tab_film_nodir;
        default: return c_tab_tiny;
    }
}

// grid: (32, n_items); block 256.  One block row per item.
__global__ void pack_kernel(int kind, ParamPtrs pp, float* __restrict__ dst) {
    const PackTable& t = dev_table(kind);
    const int it = blockIdx.y;
    if (it >= t.n_items) return;
    const PackItem item = t.item[it];
    float* out = dst + t.dst_off[it];
    const float* src = pp.p[item.param];
    if (item.type == ITEM_CHUNK) {
        const int total = item.mb * 1024;
        for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += 32 * 256) {
            int q = idx & 3, lane = (idx >> 2) & 63, rm = idx >> 8;
            int m = rm % item.mb, rg = rm / item.mb;
            int row = 32 * m + (lane & 31);
            int c = 8 * rg + 4 * (lane >> 5) + q;
            float v = 0.f;
            if (row < item.rows_valid && c < item.n_valid) v = src[(int64_t)row * item.ld + item.offset + c];
            out[idx] = v;
        }
    } else if (blockIdx.x == 0) {
        const int f = threadIdx.x;  // 256 features per piece
        if (item.type == ITEM_VEC) {
            float v = f < item.n_valid ? src[item.offset + (int64_t)f * item.stride] : 0.f;
            out[vec_slot(f)] = v;
        } else {
            out[f] = f < item.n_valid ? src[f] : 0.f;
        }
    }
}

// ---- LDS map (floats) ----------------------------------------------------------------------
constexpr int kLdsChunk = 8192;                          // 32 KiB K-block buffer (MB = 8)
constexpr int kLdsAux = kMaxAuxPieces * kPiece;          // 8 KiB per-layer aux slot
constexpr int kLdsChunk0 = 0;
constexpr int kLdsAux0 = 2 * kLdsChunk;
constexpr int kLdsFilm0 = kLdsAux0 + 2 * kLdsAux;
constexpr int kLdsFloats = kLdsFilm0 + 2 * kFilmRow;     // 21504 floats = 84 KiB

enum Act : int { ACT_LINEAR = 0, ACT_RELU = 1, ACT_SIN30 = 2, ACT_FILM = 3 };

struct Ctx {
    float* smem;
    const float* wp;        // next unread piece of the packed stream (wave-uniform)
    const float* film;      // this group's FiLM table [9][512] or nullptr
    int lane, wave, h;
};

// DMA n consecutive 1 KiB pieces g -> lds, split over the 4 waves.
__device__ __forceinline__ void dma_pieces(const float* g, float* lds, int n, int wave, int lane) {
#pragma unroll
    for (int t0 = 0; t0 < n; t0 += 4) {
        const int t = t0 + wave;
        if (t < n)
            __builtin_amdgcn_global_load_lds((const MI_GLB void*)(g + t * kPiece + lane * 4),
                                             (MI_LDS void*)(lds + t * kPiece), 16, 0, 0);
    }
}

// Issue the DMA of one stage: optional aux pieces (+ FiLM row) of a layer, then one K block.
template <int N_AUX, int N_CHUNK_PIECES, bool FILM>
__device__ __forceinline__ void issue_stage(Ctx& c, int aux_slot, int chunk_buf, int film_layer) {
    if constexpr (N_AUX > 0) {
        dma_pieces(c.wp, c.smem + kLdsAux0 + aux_slot * kLdsAux, N_AUX, c.wave, c.lane);
        c.wp += N_AUX * kPiece;
        if constexpr (FILM)
            dma_pieces(c.film + film_layer * kFilmRow, c.smem + kLdsFilm0 + aux_slot * kFilmRow, 2, c.wave, c.lane);
    }
    if constexpr (N_CHUNK_PIECES > 0) {
        dma_pieces(c.wp, c.smem + kLdsChunk0 + chunk_buf * kLdsChunk, N_CHUNK_PIECES, c.wave, c.lane);
        c.wp += N_CHUNK_PIECES * kPiece;
    }
}

// 32-wide K block: acc[m] += W[32m.., kblock] * B, A fragments from LDS (4 MFMAs per b128 read).
template <int MB>
__device__ __forceinline__ void mma_chunk(const float* chunk, int lane, const f32x16& b, f32x16 (&acc)[8]) {
    const f32x4* a4 = reinterpret_cast<const f32x4*>(chunk) + lane;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const f32x4 a = a4[(rg * MB + m) * 64];
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[4 * rg + 0], acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[4 * rg + 1], acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[4 * rg + 2], acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[4 * rg + 3], acc[m], 0, 0, 0);
        }
    }
}

// acc = bias (+ W3 * xyz for the K=3 inputs of Siren/FiLM nets), from the layer's aux slot.
template <int MB, bool K3>
__device__ __forceinline__ void init_acc(const float* aux, int h, int k3_piece, float x, float y, float z,
                                         f32x16 (&acc)[8]) {
    const lds4_t pb = lds_base(aux + h * 16);       // VEC piece entry ((m*2+h)*4 + rg) in f32x4 units
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            f32x4 t = pb[m * 8 + rg];
            if constexpr (K3) {
                const f32x4 w0 = pb[(k3_piece + 0) * 64 + m * 8 + rg];
                const f32x4 w1 = pb[(k3_piece + 1) * 64 + m * 8 + rg];
                const f32x4 w2 = pb[(k3_piece + 2) * 64 + m * 8 + rg];
                t.x = fmaf(w2.x, z, fmaf(w1.x, y, fmaf(w0.x, x, t.x)));
                t.y = fmaf(w2.y, z, fmaf(w1.y, y, fmaf(w0.y, x, t.y)));
                t.z = fmaf(w2.z, z, fmaf(w1.z, y, fmaf(w0.z, x, t.z)));
                t.w = fmaf(w2.w, z, fmaf(w1.w, y, fmaf(w0.w, x, t.w)));
            }
            acc[m][4 * rg + 0] = t.x; acc[m][4 * rg + 1] = t.y; acc[m][4 * rg + 2] = t.z; acc[m][4 * rg + 3] = t.w;
        }
    }
}

// Activation epilogue: X = act(acc).  FiLM reads gamma|beta of this layer from the film slot.
template <int MB, int ACT>
__device__ __forceinline__ void activate(const f32x16 (&acc)[8], f32x16 (&X)[8], const float* film_row, int h) {
    lds4_t pf = nullptr;
    if constexpr (ACT == ACT_FILM) pf = lds_base(film_row + h * 4);   // gamma at f, beta at 256 + f
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            f32x4 g, b;
            if constexpr (ACT == ACT_FILM) {
                g = pf[m * 8 + rg * 2];
                b = pf[64 + m * 8 + rg * 2];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = acc[m][4 * rg + q];
                float o;
                if constexpr (ACT == ACT_RELU) o = fmaxf(v, 0.f);
                else if constexpr (ACT == ACT_SIN30) o = fast_sin(__fmul_rn(30.f, v));
                else if constexpr (ACT == ACT_FILM) o = fast_sin(__fmul_rn(30.f, __fadd_rn(__fmul_rn(g[q], v), b[q])));
                else o = v;
                X[m][4 * rg + q] = o;
            }
        }
    }
}

// One MFMA layer: KB K blocks; bsel(kb) yields the B-operand register block of K block kb.
// On entry the layer's first stage (aux + K block 0) has been issued into aux slot
// `aux_slot` and chunk buffer PAR0.  NEXT_* describe the stage to issue while the last K block
// computes (the next layer's first stage), 0/0 for none.
template <int KB, int MB, int PAR0, bool K3, int NEXT_AUX, int NEXT_CHUNK, bool FILM, class BSel>
__device__ __forceinline__ void mma_layer(Ctx& c, int aux_slot, int next_film_layer, int k3_piece, float x, float y,
                                          float z, BSel bsel, f32x16 (&acc)[8]) {
    auto stage = [&](auto kbc) {
        constexpr int kb = decltype(kbc)::value;
        constexpr int cur = (PAR0 + kb) & 1;
        __syncthreads();
        if constexpr (kb + 1 < KB) issue_stage<0, MB * 4, false>(c, 0, cur ^ 1, 0);
        else issue_stage<NEXT_AUX, NEXT_CHUNK, FILM>(c, aux_slot ^ 1, cur ^ 1, next_film_layer);
        if constexpr (kb == 0)
            init_acc<MB, K3>(c.smem + kLdsAux0 + aux_slot * kLdsAux, c.h, k3_piece, x, y, z, acc);
        mma_chunk<MB>(c.smem + kLdsChunk0 + cur * kLdsChunk, c.lane, bsel(kbc), acc);
    };
    static_for<KB>(stage);
}

// sigma / rgb heads: dot products over the features a lane holds + one cross-half add.
template <int MB>
__device__ __forceinline__ float head_dot(const f32x16 (&X)[8], const float* aux, int piece, int h) {
    const lds4_t p = lds_base(aux + h * 16);
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const f32x4 w = p[piece * 64 + m * 8 + rg];
            s0 = fmaf(w.x, X[m][4 * rg + 0], s0);
            s1 = fmaf(w.y, X[m][4 * rg + 1], s1);
            s0 = fmaf(w.z, X[m][4 * rg + 2], s0);
            s1 = fmaf(w.w, X[m][4 * rg + 3], s1);
        }
    float s = s0 + s1;
    s += __shfl_xor(s, 32);
    return s;
}

__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + expf(-v)); }

// Positional-encoding blocks through an LDS scratch so sincosf is not inlined per register.
// Feature f of the encoding of (x,y,z) with L frequencies: i=f/6, c=f%6: c<3 sin(2^i x_c) else cos(2^i x_{c-3}).
template <int NBLK>
__device__ __forceinline__ void posenc_blocks(float* scr, int lane, int h, float x, float y, float z, int nfeat,
                                              f32x16* out) {
#pragma unroll 1
    for (int slot = 0; slot < NBLK * 16; ++slot) {
        const int r = slot & 15, blk = slot >> 4;
        const int f = 32 * blk + (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = 0.f;
        if (f < nfeat) {
            const int i = f / 6, cc = f - 6 * i;
            const int comp = cc >= 3 ? cc - 3 : cc;
            const float xv = comp == 0 ? x : (comp == 1 ? y : z);
            const SinCos sc = fast_sincos(ldexpf(xv, i));
            v = cc >= 3 ? sc.c : sc.s;
        }
        scr[slot * 64 + lane] = v;
    }
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[blk][r] = scr[(blk * 16 + r) * 64 + lane];
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

__device__ __forceinline__ Ctx make_ctx(float* smem, const MlpArgs& a, int64_t group) {
    Ctx c;
    c.smem = smem;
    c.wp = a.packed;
    c.film = a.film ? a.film + group * (kFilmLayers * kFilmRow) : nullptr;
    c.lane = threadIdx.x & 63;
    c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.h = c.lane >> 5;
    return c;
}

__device__ __forceinline__ void store_out(const MlpArgs& a, const PointIn& pt, int h, float r, float g, float b,
                                          float s) {
    if (pt.valid && h == 0) reinterpret_cast<f32x4*>(a.out)[pt.p] = f32x4{r, g, b, s};
}

// =========================================================================================
// NeRF (nerf/nerf.py:75-94) and TinyNeRF
// =========================================================================================
template <bool TINY>
__global__ __launch_bounds__(256, 1) void nerf_fwd_kernel(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int64_t group = blockIdx.x / a.tiles_per_group;
    const int64_t tile = blockIdx.x % a.tiles_per_group;
    Ctx c = make_ctx(smem, a, group);
    issue_stage<1, 32, false>(c, 0, 0, 0);   // layer 0: bias + K block 0 (PE features 0..31)

    const PointIn pt = load_point(a.mode, a.a, a.z, group, a.points_per_group, a.rays_per_group, a.n_samples,
                                  tile * 128 + c.wave * 32 + (c.lane & 31));
    f32x16 pe[2], pd[1], X[8], acc[8];
    float* scr = smem + kLdsChunk0 + kLdsChunk + c.wave * 2048;   // chunk buffer 1 is idle until stage 1
    posenc_blocks<2>(scr, c.lane, c.h, pt.px, pt.py, pt.pz, 60, pe);
    posenc_blocks<1>(scr, c.lane, c.h, pt.dx, pt.dy, pt.dz, 24, pd);

    const auto sel_pe = [&](auto kb) -> const f32x16& { return pe[decltype(kb)::value]; };
    const auto sel_x = [&](auto kb) -> const f32x16& { return X[decltype(kb)::value]; };
    const auto sel_skip = [&](auto kb) -> const f32x16& {
        constexpr int k = decltype(kb)::value;
        if constexpr (k < 2) return pe[k]; else return X[k - 2];
    };
    const auto sel_dir = [&](auto kb) -> const f32x16& {
        constexpr int k = decltype(kb)::value;
        if constexpr (k < 8) return X[k]; else return pd[0];
    };

    int slot = 0;
    // layers_pos[0]: 60 -> 256
    mma_layer<2, 8, 0, false, 1, 32, false>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_pe, acc);
    activate<8, ACT_RELU>(acc, X, nullptr, c.h); slot ^= 1;
    float sigma;
    if constexpr (!TINY) {
        // layers_pos[1..4]
#pragma unroll 1
        for (int l = 1; l <= 4; ++l) {
            mma_layer<8, 8, 0, false, 1, 32, false>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc);
            activate<8, ACT_RELU>(acc, X, nullptr, c.h); slot ^= 1;
        }
        // layers_pos[5]: [PE(60) | h(256)] -> 256
        mma_layer<10, 8, 0, false, 1, 32, false>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_skip, acc);
        activate<8, ACT_RELU>(acc, X, nullptr, c.h); slot ^= 1;
        // layers_pos[6]
        mma_layer<8, 8, 0, false, 3, 32, false>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc);
        activate<8, ACT_RELU>(acc, X, nullptr, c.h); slot ^= 1;
        // layers_pos[7] (+ sigma head pieces)
        mma_layer<8, 8, 0, false, 1, 32, false>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc);
        activate<8, ACT_RELU>(acc, X, nullptr, c.h);
        {
            const float* aux = smem + kLdsAux0 + slot * kLdsAux;
            sigma = fmaxf(head_dot<8>(X, aux, 1, c.h) + aux[2 * kPiece], 0.f);
        }
        slot ^= 1;
        // layers_dir[0]: linear
        mma_layer<8, 8, 0, false, 5, 16, false>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc);
        activate<8, ACT_LINEAR>(acc, X, nullptr, c.h); slot ^= 1;
    } else {
        // layers_pos[1], [2], [3] (+ sigma head pieces), then the dir layer's 5 aux pieces
        mma_layer<8, 8, 0, false, 1, 32, false>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc);
        activate<8, ACT_RELU>(acc, X, nullptr, c.h); slot ^= 1;
        mma_layer<8, 8, 0, false, 3, 32, false>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc);
        activate<8, ACT_RELU>(acc, X, nullptr, c.h); slot ^= 1;
        mma_layer<8, 8, 0, false, 5, 16, false>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc);
        activate<8, ACT_RELU>(acc, X, nullptr, c.h);
        {
            const float* aux = smem + kLdsAux0 + slot * kLdsAux;
            sigma = fmaxf(head_dot<8>(X, aux, 1, c.h) + aux[2 * kPiece], 0.f);
        }
        slot ^= 1;
    }
    // layers_dir[1] (TinyNeRF: layers_dir[0]): [h(256) | PE_dir(24)] -> 128, relu; then rgb head
    mma_layer<9, 4, 0, false, 0, 0, false>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_dir, acc);
    activate<4, ACT_RELU>(acc, X, nullptr, c.h);
    const float* aux = smem + kLdsAux0 + slot * kLdsAux;
    const float r = sigmoidf(head_dot<4>(X, aux, 1, c.h) + aux[4 * kPiece + 0]);
    const float g = sigmoidf(head_dot<4>(X, aux, 2, c.h) + aux[4 * kPiece + 1]);
    const float b = sigmoidf(head_dot<4>(X, aux, 3, c.h) + aux[4 * kPiece + 2]);
    store_out(a, pt, c.h, r, g, b, sigma);
}

// =========================================================================================
// SirenNeRF (nerf/nerf.py:153-170): sin(30 * linear) layers on raw xyz / dir
// =========================================================================================
__global__ __launch_bounds__(256, 1) void siren_fwd_kernel(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int64_t group = blockIdx.x / a.tiles_per_group;
    const int64_t tile = blockIdx.x % a.tiles_per_group;
    Ctx c = make_ctx(smem, a, group);
    issue_stage<4, 0, false>(c, 0, 0, 0);   // layers_pos[0]: bias + 3 weight columns (K = 3, VALU)

    const PointIn pt = load_point(a.mode, a.a, a.z, group, a.points_per_group, a.rays_per_group, a.n_samples,
                                  tile * 128 + c.wave * 32 + (c.lane & 31));
    f32x16 X[8], acc[8];
    const auto sel_x = [&](auto kb) -> const f32x16& { return X[decltype(kb)::value]; };

    int slot = 0;
    __syncthreads();
    issue_stage<1, 32, false>(c, 1, 0, 0);
    init_acc<8, true>(smem + kLdsAux0, c.h, 1, pt.px, pt.py, pt.pz, acc);
    activate<8, ACT_SIN30>(acc, X, nullptr, c.h); slot ^= 1;
#pragma unroll 1
    for (int l = 1; l <= 3; ++l) {
        mma_layer<8, 8, 0, false, 1, 32, false>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc);
        activate<8, ACT_SIN30>(acc, X, nullptr, c.h); slot ^= 1;
    }
    mma_layer<8, 8, 0, false, 4, 32, false>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc);     // layers_pos[4]
    activate<8, ACT_SIN30>(acc, X, nullptr, c.h); slot ^= 1;
    mma_layer<8, 8, 0, true, 1, 32, false>(c, slot, 0, 1, pt.px, pt.py, pt.pz, sel_x, acc);  // [5]: [pos | h]
    activate<8, ACT_SIN30>(acc, X, nullptr, c.h); slot ^= 1;
    mma_layer<8, 8, 0, false, 3, 32, false>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc);     // [6]
    activate<8, ACT_SIN30>(acc, X, nullptr, c.h); slot ^= 1;
    mma_layer<8, 8, 0, false, 1, 32, false>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc);     // [7] + sigma head
    activate<8, ACT_SIN30>(acc, X, nullptr, c.h);
    float sigma;
    {
        const float* aux = smem + kLdsAux0 + slot * kLdsAux;
        sigma = fmaxf(head_dot<8>(X, aux, 1, c.h) + aux[2 * kPiece], 0.f);
    }
    slot ^= 1;
    mma_layer<8, 8, 0, false, 8, 16, false>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc);     // layers_dir[0] linear
    activate<8, ACT_LINEAR>(acc, X, nullptr, c.h); slot ^= 1;
    mma_layer<8, 4, 0, true, 0, 0, false>(c, slot, 0, 1, pt.dx, pt.dy, pt.dz, sel_x, acc);   // layers_dir[1]: [h | dir]
    activate<4, ACT_SIN30>(acc, X, nullptr, c.h);
    const float* aux = smem + kLdsAux0 + slot * kLdsAux;
    const float r = sigmoidf(head_dot<4>(X, aux, 4, c.h) + aux[7 * kPiece + 0]);
    const float g = sigmoidf(head_dot<4>(X, aux, 5, c.h) + aux[7 * kPiece + 1]);
    const float b = sigmoidf(head_dot<4>(X, aux, 6, c.h) + aux[7 * kPiece + 2]);
    store_out(a, pt, c.h, r, g, b, sigma);
}

// =========================================================================================
// FilmSirenNeRF (pi_GAN/modules.py:101-118): sin(30 * (gamma * linear + beta)), one FiLM
// table per group (image)
// =========================================================================================
template <bool USE_DIR>
__global__ __launch_bounds__(256, 1) void film_fwd_kernel(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int64_t group = blockIdx.x / a.tiles_per_group;
    const int64_t tile = blockIdx.x % a.tiles_per_group;
    Ctx c = make_ctx(smem, a, group);
    issue_stage<4, 0, true>(c, 0, 0, 0);   // input_layer: bias + 3 columns, FiLM row 0

    const PointIn pt = load_point(a.mode, a.a, a.z, group, a.points_per_group, a.rays_per_group, a.n_samples,
                                  tile * 128 + c.wave * 32 + (c.lane & 31));
    f32x16 X[8], acc[8];
    const auto sel_x = [&](auto kb) -> const f32x16& { return X[decltype(kb)::value]; };
    const auto film_row = [&](int s) { return smem + kLdsFilm0 + s * kFilmRow; };

    int slot = 0;
    __syncthreads();
    issue_stage<1, 32, true>(c, 1, 0, 1);
    init_acc<8, true>(smem + kLdsAux0, c.h, 1, pt.px, pt.py, pt.pz, acc);
    activate<8, ACT_FILM>(acc, X, film_row(0), c.h); slot ^= 1;
#pragma unroll 1
    for (int l = 1; l <= 5; ++l) {                                                   // hidden_layers[0..4]
        mma_layer<8, 8, 0, false, 1, 32, true>(c, slot, l + 1, 0, 0.f, 0.f, 0.f, sel_x, acc);
        activate<8, ACT_FILM>(acc, X, film_row(slot), c.h); slot ^= 1;
    }
    mma_layer<8, 8, 0, false, 3, 32, true>(c, slot, 7, 0, 0.f, 0.f, 0.f, sel_x, acc);  // hidden_layers[5]
    activate<8, ACT_FILM>(acc, X, film_row(slot), c.h); slot ^= 1;
    mma_layer<8, 8, 0, false, USE_DIR ? 8 : 5, 32, true>(c, slot, 8, 0, 0.f, 0.f, 0.f, sel_x, acc);  // hidden_layers[6]
    activate<8, ACT_FILM>(acc, X, film_row(slot), c.h);
    float sigma;
    {
        const float* aux = smem + kLdsAux0 + slot * kLdsAux;
        sigma = fmaxf(head_dot<8>(X, aux, 1, c.h) + aux[2 * kPiece], 0.f);
    }
    slot ^= 1;
    mma_layer<8, 8, 0, USE_DIR, 0, 0, false>(c, slot, 0, 1, pt.dx, pt.dy, pt.dz, sel_x, acc);   // hidden_layer_rgb
    activate<8, ACT_FILM>(acc, X, film_row(slot), c.h);
    const float* aux = smem + kLdsAux0 + slot * kLdsAux;
    constexpr int hp = USE_DIR ? 4 : 1;
    const float r = sigmoidf(head_dot<8>(X, aux, hp + 0, c.h) + aux[(hp + 3) * kPiece + 0]);
    const float g = sigmoidf(head_dot<8>(X, aux, hp + 1, c.h) + aux[(hp + 3) * kPiece + 1]);
    const float b = sigmoidf(head_dot<8>(X, aux, hp + 2, c.h) + aux[(hp + 3) * kPiece + 2]);
    store_out(a, pt, c.h, r, g, b, sigma);
}

// ---- host side ---------------------------------------------------------------------------
int launch_pack(int kind, const float* const* params, int n_params, float* packed, hipStream_t stream) {
    const PackTable* t = host_table(kind);
    ParamPtrs pp{};
    for (int i = 0; i < n_params && i < 24; ++i) pp.p[i] = params[i];
    hipLaunchKernelGGL(pack_kernel, dim3(32, t->n_items), dim3(256), 0, stream, kind, pp, packed);
    return check_launch("pack_kernel");
}

int launch_mlp(int kind, const MlpArgs& a, int64_t n_groups, hipStream_t stream) {
    const int64_t blocks = n_groups * a.tiles_per_group;
    if (blocks <= 0) return 0;
    if (blocks > 0x7fffffffLL) { set_error("too many point tiles (%lld)", (long long)blocks); return -1; }
    const dim3 grid((unsigned)blocks), block(256);
    const size_t lds = kLdsFloats * sizeof(float);
    // 84 KiB of dynamic LDS: raise the per-kernel limit once (host-side attribute, no device work)
    static bool attr_done = false;
    if (!attr_done) {
        const void* fns[] = {(const void*)nerf_fwd_kernel<false>, (const void*)siren_fwd_kernel,
                             (const void*)film_fwd_kernel<true>, (const void*)film_fwd_kernel<false>,
                             (const void*)nerf_fwd_kernel<true>};
        for (const void* f : fns) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return -2; }
        }
        attr_done = true;
    }
    switch (kind) {
        case 0: hipLaunchKernelGGL(nerf_fwd_kernel<false>, grid, block, lds, stream, a); break;
        case 1: hipLaunchKernelGGL(siren_fwd_kernel, grid, block, lds, stream, a); break;
        case 2: hipLaunchKernelGGL(film_fwd_kernel<true>, grid, block, lds, stream, a); break;
        case 3: hipLaunchKernelGGL(film_fwd_kernel<false>, grid, block, lds, stream, a); break;
        case 4: hipLaunchKernelGGL(nerf_fwd_kernel<true>, grid, block, lds, stream, a); break;
        default: set_error("unknown field kind %d", kind); return -1;
    }
    return check_launch("field_mlp_fwd");
}

}  // namespace mi
