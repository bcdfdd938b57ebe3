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

#include "field_mlp_device.h"

namespace mi {

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
__global__ void pack_kernel(int kind, ParamPtrs pp, float* __restrict__ dst, float w0) {
    const PackTable& t = dev_table(kind);
    const int it = blockIdx.y;
    if (it >= t.n_items) return;
    if (it == 0 && blockIdx.x == 0) {                        // the trailer piece: hyper-parameters (field_layout.h:kTrailer)
        float* tr = dst + packed_body_floats(t);
        tr[threadIdx.x] = threadIdx.x == 0 ? w0 : (threadIdx.x == 1 ? w0 * w0 : 0.f);
    }
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
            if (row < item.rows_valid && c < item.n_valid)
                v = src[(int64_t)row * item.ld + item.offset + (int64_t)c * item.stride];
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

// =========================================================================================
// NeRF (nerf/nerf.py:75-94) and TinyNeRF
// =========================================================================================
// SAVE = training forward: every linear layer's input is also written to HBM ([point][feature] rows,
// region table nerf_acts()/tiny_acts() in field_layout.h) for the backward pass.
template <bool TINY, bool SAVE>
__global__ __launch_bounds__(256, 1) void nerf_fwd_kernel(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int64_t group = blockIdx.x / a.tiles_per_group;
    const int64_t tile = blockIdx.x % a.tiles_per_group;
    Ctx c = make_ctx(smem, a, group);
    MI_STAMP(a, 0);
    issue_first_stage<1, 32, false>(c, 0, 0, 0);   // layer 0: bias + K block 0 (PE features 0..31)

    const PointIn pt = load_point(a.mode, a.a, a.z, group, a.points_per_group, a.rays_per_group, a.n_samples,
                                  tile * 128 + c.wave * 32 + (c.lane & 31));
    f32x16 pe[2], pd[1], X[8], acc[8];
    float* scr = smem + kLdsChunk0 + kLdsChunk + c.wave * 2048;   // chunk buffer 1 is idle until stage 1
    posenc_blocks<2>(scr, c.lane, c.h, pt.px, pt.py, pt.pz, 60, pe);
    posenc_blocks<1>(scr, c.lane, c.h, pt.dx, pt.dy, pt.dz, 24, pd);

    constexpr RegionLayout RL = TINY ? tiny_acts() : nerf_acts();
    const int64_t SP = a.save_points;
    // `sw`: the region of the layer's ReLU switch bits (nerf_acts() 12.. / tiny_acts() 7..), -1 for none
    const auto rows = [&](int off_floats_per_point, int width, int sw = -1) {
        return SaveRows{SAVE ? a.save + (int64_t)off_floats_per_point * SP : nullptr, width, pt.p, pt.valid,
                        SAVE && sw >= 0 ? a.save + (int64_t)region_offset(RL, sw) * SP : nullptr};
    };
    constexpr int SW0 = TINY ? 7 : 12;                     // switch region of H1; H_l: SW0 + l - 1; H_d: the last region
    if constexpr (SAVE) {
        f32x16 tmp[8];
        tmp[0] = pe[0]; tmp[1] = pe[1];
        store_rows<2>(a.save, 64, pt.p, pt.valid, c.h, tmp);
        tmp[0] = pd[0];
        store_rows<1>(a.save + (int64_t)region_offset(RL, TINY ? 5 : 10) * SP, 32, pt.p, pt.valid, c.h, tmp);
    }

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
    MI_STAMP(a, 1);
    // layers_pos[0]: 60 -> 256
    fwd_layer<2, 8, false, 1, 32, false, ACT_RELU, SAVE, true>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_pe, acc, X, nullptr, rows(64, 256, SW0));
    slot ^= 1;                                                          // H1
    MI_STAMP(a, 3);
    float sigma;
    if constexpr (!TINY) {
        // layers_pos[1..4]
#pragma unroll 1
        for (int l = 1; l <= 4; ++l) {
#ifdef MI_PROFILE_STAMPS
            if (l == 2 && a.stamps) c.rowst = a.stamps + (int64_t)blockIdx.x * 128 + 32;   // rows of layers_pos[2]: 32..64
#endif
            fwd_layer<8, 8, false, 1, 32, false, ACT_RELU, SAVE, true, 8>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc, X, nullptr,
                                                                          rows(64 + 256 * l, 256, SW0 + l), rows(64 + 256 * (l - 1), 256));   // H2..H5
            slot ^= 1;
            MI_ROW_STAMP(c);
#ifdef MI_PROFILE_STAMPS
            c.rowst = nullptr;
#endif
            MI_STAMP(a, 3 + 2 * l);
        }
        // layers_pos[5]: [PE(60) | h(256)] -> 256
        fwd_layer<10, 8, false, 1, 32, false, ACT_RELU, SAVE, true, 8>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_skip, acc, X, nullptr,
                                                                       rows(region_offset(RL, 6), 256, SW0 + 5), rows(region_offset(RL, 5), 256));   // H6
        slot ^= 1;
        MI_STAMP(a, 13);
        // layers_pos[6]
        fwd_layer<8, 8, false, 3, 32, false, ACT_RELU, SAVE, true, 8>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc, X, nullptr,
                                                                      rows(region_offset(RL, 7), 256, SW0 + 6), rows(region_offset(RL, 6), 256));    // H7
        slot ^= 1;
        MI_STAMP(a, 15);
        // layers_pos[7] (+ sigma head pieces)
        fwd_layer<8, 8, false, 1, 32, false, ACT_RELU, SAVE, true, 8>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc, X, nullptr,
                                                                      rows(region_offset(RL, 8), 256, SW0 + 7), rows(region_offset(RL, 7), 256));    // H8
        {
            const float* aux = smem + kLdsAux0 + slot * kLdsAux;
            sigma = fmaxf(head_dot<8>(X, aux, 1, c.h) + aux[2 * kPiece], 0.f);
        }
        slot ^= 1;
        MI_STAMP(a, 17);
        // layers_dir[0]: linear
        fwd_layer<8, 8, false, 5, 16, false, ACT_LINEAR, SAVE, true, 8>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc, X, nullptr,
                                                                        rows(region_offset(RL, 9), 256), rows(region_offset(RL, 8), 256));  // G
        slot ^= 1;
        MI_STAMP(a, 19);
    } else {
        // layers_pos[1], [2], [3] (+ sigma head pieces), then the dir layer's 5 aux pieces
        fwd_layer<8, 8, false, 1, 32, false, ACT_RELU, SAVE, true, 8>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc, X, nullptr,
                                                                      rows(region_offset(RL, 2), 256, SW0 + 1), rows(region_offset(RL, 1), 256));
        slot ^= 1;
        fwd_layer<8, 8, false, 3, 32, false, ACT_RELU, SAVE, true, 8>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc, X, nullptr,
                                                                      rows(region_offset(RL, 3), 256, SW0 + 2), rows(region_offset(RL, 2), 256));
        slot ^= 1;
        fwd_layer<8, 8, false, 5, 16, false, ACT_RELU, SAVE, true, 8>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc, X, nullptr,
                                                                      rows(region_offset(RL, 4), 256, SW0 + 3), rows(region_offset(RL, 3), 256));
        {
            const float* aux = smem + kLdsAux0 + slot * kLdsAux;
            sigma = fmaxf(head_dot<8>(X, aux, 1, c.h) + aux[2 * kPiece], 0.f);
        }
        slot ^= 1;
    }
    // layers_dir[1] (TinyNeRF: layers_dir[0]): [h(256) | PE_dir(24)] -> 128, relu; then rgb head
    fwd_layer<9, 4, false, 0, 0, false, ACT_RELU, SAVE, false, 8>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_dir, acc, X, nullptr,
                                                                  rows(region_offset(RL, TINY ? 6 : 11), 128, RL.n - 1),
                                                                  rows(region_offset(RL, TINY ? 4 : 9), 256));   // H_d; G / H4 from X
    MI_STAMP(a, 20);
    const float* aux = smem + kLdsAux0 + slot * kLdsAux;
    const float r = sigmoidf(head_dot<4>(X, aux, 1, c.h) + aux[4 * kPiece + 0]);
    const float g = sigmoidf(head_dot<4>(X, aux, 2, c.h) + aux[4 * kPiece + 1]);
    const float b = sigmoidf(head_dot<4>(X, aux, 3, c.h) + aux[4 * kPiece + 2]);
    store_out(a, pt, c.h, r, g, b, sigma);
    MI_STAMP(a, 21);
}

// =========================================================================================
// SirenNeRF (nerf/nerf.py:153-170): sin(30 * linear) layers on raw xyz / dir
// =========================================================================================
template <bool SAVE>
__global__ __launch_bounds__(256, 1) void siren_fwd_kernel(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int64_t group = blockIdx.x / a.tiles_per_group;
    const int64_t tile = blockIdx.x % a.tiles_per_group;
    Ctx c = make_ctx(smem, a, group);
    issue_first_stage<4, 0, false>(c, 0, 0, 0);   // layers_pos[0]: bias + 3 weight columns (K = 3, VALU)

    const PointIn pt = load_point(a.mode, a.a, a.z, group, a.points_per_group, a.rays_per_group, a.n_samples,
                                  tile * 128 + c.wave * 32 + (c.lane & 31));
    f32x16 X[8], acc[8];
    const auto sel_x = [&](auto kb) -> const f32x16& { return X[decltype(kb)::value]; };
    constexpr RegionLayout RL = siren_acts();
    const int64_t SP = a.save_points;
    const auto region = [&](int idx) { return a.save + (int64_t)region_offset(RL, idx) * SP; };
    // sin layer l (1..8): X_l (cosine sign in the lowest bit) -> region l
    const auto sin_rows = [&](int l) {
        return SaveRows{SAVE ? a.save + (int64_t)(8 + 256 * (l - 1)) * SP : nullptr, 256, pt.p, pt.valid};
    };
    const SaveRows none{nullptr, 0, 0, false};
    const auto sin_act = [&](int l) {
        if constexpr (SAVE)
            activate_train<8, ACT_SIN30>(acc, X, nullptr, c.h, smem + kLdsAux0, a.save + (int64_t)(8 + 256 * (l - 1)) * SP, 256,
                                         pt.p, pt.valid);
        else
            activate<8, ACT_SIN30>(acc, X, nullptr, c.h, smem + kLdsAux0);
    };
    if constexpr (SAVE) {
        if (pt.valid && c.h == 0) {
            f32x4* xin = reinterpret_cast<f32x4*>(a.save + pt.p * 8);
            xin[0] = f32x4{pt.px, pt.py, pt.pz, pt.dx};
            xin[1] = f32x4{pt.dy, pt.dz, 0.f, 0.f};
        }
    }

    int slot = 0;
    __syncthreads();
    issue_first_stage<1, 32, false>(c, 1, 0, 0);
    init_acc<8, true>(smem + kLdsAux0, c.h, 1, pt.px, pt.py, pt.pz, acc);
    sin_act(1); slot ^= 1;
    // sin layers store their own (encoded) rows from the activation hook: nothing is deferred to the next layer
#pragma unroll 1
    for (int l = 1; l <= 3; ++l) {
        fwd_layer<8, 8, false, 1, 32, false, ACT_SIN30, SAVE, false, 0>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc, X, nullptr, sin_rows(l + 1), none);
        slot ^= 1;
    }
    fwd_layer<8, 8, false, 4, 32, false, ACT_SIN30, SAVE, false, 0>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc, X, nullptr, sin_rows(5), none);  // layers_pos[4]
    slot ^= 1;
    fwd_layer<8, 8, true, 1, 32, false, ACT_SIN30, SAVE, false, 0>(c, slot, 0, 1, pt.px, pt.py, pt.pz, sel_x, acc, X, nullptr, sin_rows(6), none);  // [5]: [pos | h]
    slot ^= 1;
    fwd_layer<8, 8, false, 3, 32, false, ACT_SIN30, SAVE, false, 0>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc, X, nullptr, sin_rows(7), none);  // [6]
    slot ^= 1;
    fwd_layer<8, 8, false, 1, 32, false, ACT_SIN30, SAVE, false, 0>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc, X, nullptr, sin_rows(8), none);  // [7] + sigma head
    float sigma;
    {
        const float* aux = smem + kLdsAux0 + slot * kLdsAux;
        sigma = fmaxf(head_dot<8>(X, aux, 1, c.h) + aux[2 * kPiece], 0.f);
    }
    slot ^= 1;
    const SaveRows g_rows{SAVE ? region(9) : nullptr, 256, pt.p, pt.valid};
    fwd_layer<8, 8, false, 8, 16, false, ACT_LINEAR, SAVE, true, 0>(c, slot, 0, 0, 0.f, 0.f, 0.f, sel_x, acc, X, nullptr, g_rows,
                                                                    none);  // layers_dir[0] linear: G (stored by the next layer)
    slot ^= 1;
    fwd_layer<8, 4, true, 0, 0, false, ACT_SIN30, SAVE, false, 8>(c, slot, 0, 1, pt.dx, pt.dy, pt.dz, sel_x, acc, X, nullptr,
                                                                  SaveRows{SAVE ? region(10) : nullptr, 128, pt.p, pt.valid},
                                                                  g_rows);  // layers_dir[1]: [h | dir]; G from X
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
template <bool USE_DIR, bool SAVE>
__global__ __launch_bounds__(256, 1) void film_fwd_kernel(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int64_t group = blockIdx.x / a.tiles_per_group;
    const int64_t tile = blockIdx.x % a.tiles_per_group;
    Ctx c = make_ctx(smem, a, group);
    // FilmSiren's w_0 (pi_GAN/modules.py:11,73): the first float of the stream's trailer, a uniform (scalar) load
    constexpr int kBody = USE_DIR ? packed_body_floats(h_tab_film) : packed_body_floats(h_tab_film_nodir);
    c.w0 = a.packed[kBody];
    issue_first_stage<4, 0, true>(c, 0, 0, 0);   // input_layer: bias + 3 columns, FiLM row 0

    const PointIn pt = load_point(a.mode, a.a, a.z, group, a.points_per_group, a.rays_per_group, a.n_samples,
                                  tile * 128 + c.wave * 32 + (c.lane & 31));
    f32x16 X[8], acc[8];
    const auto sel_x = [&](auto kb) -> const f32x16& { return X[decltype(kb)::value]; };
    const auto film_row = [&](int s) { return smem + kLdsFilm0 + s * kFilmRow; };
    const int64_t SP = a.save_points;
    // FiLM layer l (0..8): X_l (cosine sign in the lowest bit) -> region 1+l (256 wide, after the 8-wide xin)
    const auto film_rows = [&](int l) {
        return SaveRows{SAVE ? a.save + (int64_t)(8 + 256 * l) * SP : nullptr, 256, pt.p, pt.valid};
    };
    const SaveRows none{nullptr, 0, 0, false};
    const auto film_act = [&](int l, int slot_) {
        if constexpr (SAVE)
            activate_train<8, ACT_FILM>(acc, X, film_row(slot_), c.h, smem + kLdsAux0, a.save + (int64_t)(8 + 256 * l) * SP, 256,
                                        pt.p, pt.valid, c.w0);
        else
            activate<8, ACT_FILM>(acc, X, film_row(slot_), c.h, smem + kLdsAux0, c.w0);
    };
    if constexpr (SAVE) {
        if (pt.valid && c.h == 0) {
            f32x4* xin = reinterpret_cast<f32x4*>(a.save + pt.p * 8);
            xin[0] = f32x4{pt.px, pt.py, pt.pz, pt.dx};
            xin[1] = f32x4{pt.dy, pt.dz, 0.f, 0.f};
        }
    }

    int slot = 0;
    __syncthreads();
    issue_first_stage<1, 32, true>(c, 1, 0, 1);
    init_acc<8, true>(smem + kLdsAux0, c.h, 1, pt.px, pt.py, pt.pz, acc);
    film_act(0, 0); slot ^= 1;
#pragma unroll 1
    for (int l = 1; l <= 5; ++l) {                                                   // hidden_layers[0..4]
        fwd_layer<8, 8, false, 1, 32, true, ACT_FILM, SAVE, false, 0>(c, slot, l + 1, 0, 0.f, 0.f, 0.f, sel_x, acc, X, film_row(slot), film_rows(l), none);
        slot ^= 1;
    }
    fwd_layer<8, 8, false, 3, 32, true, ACT_FILM, SAVE, false, 0>(c, slot, 7, 0, 0.f, 0.f, 0.f, sel_x, acc, X, film_row(slot), film_rows(6), none);  // hidden_layers[5]
    slot ^= 1;
    fwd_layer<8, 8, false, USE_DIR ? 8 : 5, 32, true, ACT_FILM, SAVE, false, 0>(c, slot, 8, 0, 0.f, 0.f, 0.f, sel_x, acc, X, film_row(slot), film_rows(7), none);  // hidden_layers[6]
    float sigma;
    {
        const float* aux = smem + kLdsAux0 + slot * kLdsAux;
        sigma = fmaxf(head_dot<8>(X, aux, 1, c.h) + aux[2 * kPiece], 0.f);
    }
    slot ^= 1;
    fwd_layer<8, 8, USE_DIR, 0, 0, false, ACT_FILM, SAVE, false, 0>(c, slot, 0, 1, pt.dx, pt.dy, pt.dz, sel_x, acc, X, film_row(slot), film_rows(8), none);   // hidden_layer_rgb
    const float* aux = smem + kLdsAux0 + slot * kLdsAux;
    constexpr int hp = USE_DIR ? 4 : 1;
    const float r = sigmoidf(head_dot<8>(X, aux, hp + 0, c.h) + aux[(hp + 3) * kPiece + 0]);
    const float g = sigmoidf(head_dot<8>(X, aux, hp + 1, c.h) + aux[(hp + 3) * kPiece + 1]);
    const float b = sigmoidf(head_dot<8>(X, aux, hp + 2, c.h) + aux[(hp + 3) * kPiece + 2]);
    store_out(a, pt, c.h, r, g, b, sigma);
}

// ---- host side ---------------------------------------------------------------------------
int launch_pack(int kind, const float* const* params, int n_params, float w0, float* packed, hipStream_t stream) {
    const PackTable* t = host_table(kind);
    ParamPtrs pp{};
    for (int i = 0; i < n_params && i < 24; ++i) pp.p[i] = params[i];
    hipLaunchKernelGGL(pack_kernel, dim3(32, t->n_items), dim3(256), 0, stream, kind, pp, packed, w0);
    return check_launch("pack_kernel");
}

int launch_mlp(int kind, const MlpArgs& a, int64_t n_groups, hipStream_t stream) {
    const int64_t blocks = n_groups * a.tiles_per_group;
    if (blocks <= 0) return 0;
    if (blocks > 0x7fffffffLL) { set_error("too many point tiles (%lld)", (long long)blocks); return -1; }
    const dim3 grid((unsigned)blocks), block(256);
    const size_t lds = kLdsFloats * sizeof(float);
    // 148 KiB of dynamic LDS: raise the per-kernel limit once per device (host-side attribute, no device work)
    static PerDeviceOnce attr_once;
    const int arc = attr_once.run([&]() {
        const void* fns[] = {(const void*)nerf_fwd_kernel<false, false>, (const void*)nerf_fwd_kernel<false, true>,
                             (const void*)nerf_fwd_kernel<true, false>, (const void*)nerf_fwd_kernel<true, true>,
                             (const void*)siren_fwd_kernel<false>, (const void*)siren_fwd_kernel<true>,
                             (const void*)film_fwd_kernel<true, false>, (const void*)film_fwd_kernel<true, true>,
                             (const void*)film_fwd_kernel<false, false>, (const void*)film_fwd_kernel<false, true>};
        for (const void* f : fns) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return -2; }
        }
        return 0;
    });
    if (arc) return arc;
    const bool sv = a.save != nullptr;
#define MI_LAUNCH(K) hipLaunchKernelGGL((K), grid, block, lds, stream, a)
    switch (kind) {
        case 0: if (sv) MI_LAUNCH((nerf_fwd_kernel<false, true>)); else MI_LAUNCH((nerf_fwd_kernel<false, false>)); break;
        case 1: if (sv) MI_LAUNCH((siren_fwd_kernel<true>)); else MI_LAUNCH((siren_fwd_kernel<false>)); break;
        case 2: if (sv) MI_LAUNCH((film_fwd_kernel<true, true>)); else MI_LAUNCH((film_fwd_kernel<true, false>)); break;
        case 3: if (sv) MI_LAUNCH((film_fwd_kernel<false, true>)); else MI_LAUNCH((film_fwd_kernel<false, false>)); break;
        case 4: if (sv) MI_LAUNCH((nerf_fwd_kernel<true, true>)); else MI_LAUNCH((nerf_fwd_kernel<true, false>)); break;
        default: set_error("unknown field kind %d", kind); return -1;
    }
#undef MI_LAUNCH
    return check_launch("field_mlp_fwd");
}

}  // namespace mi
