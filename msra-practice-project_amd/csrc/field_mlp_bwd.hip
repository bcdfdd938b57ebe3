// field_mlp_bwd.hip - backward of the fused field MLP (gfx950): what autograd does for
// `network(inputs)` inside run_network (reference nerf/render.py:72-74) when train_nerf.py:168 /
// pi_GAN/train.py call loss.backward().
//
// Three kinds of kernels, all fp32:
//   1. backward CHAIN (one per field kind): same structure as the forward - 128 points per workgroup,
//      features on MFMA rows, the accumulators of dX_l = W_l^T dA_l are, after the activation derivative,
//      the B operands of the next (earlier) layer; transposed weights stream through LDS from a second
//      packed stream (field_layout.h build_*_bwd).  It reads the saved layer inputs for the activation
//      derivative and writes dA of every linear layer as [point][feature] rows.
//   2. dW GEMM: dW_l[M,K] = sum_p dA_l[p,:]^T X_l[p,:] - a plain GEMM whose contraction is over POINTS.
//      Features sit on the MFMA lanes (one float4 load per lane gives the operands of four 32-wide blocks
//      straight from the row-major rows, no LDS), each workgroup reduces a slab of points into a full
//      256x256 tile held in accumulators (256 registers per lane), slabs are combined by a second kernel in
//      a fixed order (deterministic, no atomics).  Bias gradients ride along as column sums of dA.
//   3. head gradients (1- and 3-row weights): VALU reduction over point slabs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "field_mlp_device.h"

#include <type_traits>

namespace mi {

__constant__ PackTable c_tabb_nerf = build_nerf_bwd();
__constant__ PackTable c_tabb_siren = build_siren_nerf_bwd();
__constant__ PackTable c_tabb_film = build_film_bwd(true);
__constant__ PackTable c_tabb_film_nodir = build_film_bwd(false);
__constant__ PackTable c_tabb_tiny = build_tiny_nerf_bwd();

static constexpr PackTable h_tabb_nerf = build_nerf_bwd();
static constexpr PackTable h_tabb_siren = build_siren_nerf_bwd();
static constexpr PackTable h_tabb_film = build_film_bwd(true);
static constexpr PackTable h_tabb_film_nodir = build_film_bwd(false);
static constexpr PackTable h_tabb_tiny = build_tiny_nerf_bwd();

const PackTable* host_table_bwd(int kind) {
    switch (kind) {
        case 0: return &h_tabb_nerf;
        case 1: return &h_tabb_siren;
        case 2: return &h_tabb_film;
        case 3: return &h_tabb_film_nodir;
        case 4: return &h_tabb_tiny;
    }
    return nullptr;
}

struct ParamPtrsB { const float* p[24]; };

__device__ __forceinline__ const PackTable& dev_table_bwd(int kind) {
    switch (kind) {
        case 0: return c_tabb_nerf;
        case 1: return c_tabb_siren;
        case 2: return c_tabb_film;
        case 3: return c_tabb_film_nodir;
        default: return c_tabb_tiny;
    }
}

__global__ void pack_bwd_kernel(int kind, ParamPtrsB pp, float* __restrict__ dst, float w0) {
    const PackTable& t = dev_table_bwd(kind);
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
        const int f = threadIdx.x;
        if (item.type == ITEM_VEC) out[vec_slot(f)] = f < item.n_valid ? src[item.offset + (int64_t)f * item.stride] : 0.f;
        else out[f] = f < item.n_valid ? src[f] : 0.f;
    }
}

struct BwdArgs {
    const float* packed;     // backward stream
    const float* acts;       // saved layer inputs (regions x points)
    float* grads;            // dA regions x points (written here)
    const float* raw;        // [P,4] forward outputs (rgb post-sigmoid, sigma post-ReLU)
    const float* g_raw;      // [P,4] dL/d(raw)
    int64_t points;
    // FiLM kinds: one table [9][512] per group of points_per_group consecutive points
    const float* film;
    float* film_partial;     // unused by the chain (kept so the argument block stays stable)
    int64_t points_per_group, tiles_per_group, n_tiles;
    unsigned long long* stamps;   // diagnostic build (-DMI_PROFILE_STAMPS) only: [block][128] s_memtime values (MI_STAMP)
};

#ifdef MI_PROFILE_STAMPS
unsigned long long* g_bwd_stamps = nullptr;       // set by mi_debug_set_stamps (api.hip), diagnostic build only
#endif

// f32x4 element i of a lane's row in a [point][width] region: a UNIFORM base pointer (SGPR pair) + one per-lane 32-bit
// byte offset (+ 16 i as the instruction's immediate) - the `global_load/store v, v_off, s[base:base+1] offset:imm` form.
template <class V>
struct RowRef {
    using Byte = std::conditional_t<std::is_const_v<V>, const char, char>;
    Byte* base;
    uint32_t off;
    __device__ __forceinline__ V& operator[](int i) const {
        return *reinterpret_cast<V*>(base + (uint64_t)off + (uint64_t)((uint32_t)i * 16u));
    }
};

// The saved row of the chain's first epilogue (the 128-wide dir layer), loaded at the top of the kernel: its latency
// overlaps the first weight stage's DMA and the head gradients instead of following them.
template <int MB>
__device__ __forceinline__ void load_saved_row(f32x4 (&sv)[MB * 4], const float* __restrict__ rows, int64_t ld, int64_t p, int h) {
    const f32x4* row = reinterpret_cast<const f32x4*>(rows + p * ld + 4 * h);
#pragma unroll
    for (int j = 0; j < MB * 4; ++j) sv[j] = row[(j / 4) * 8 + (j % 4) * 2];
}

// A lane's switch words of a ReLU layer (field_layout.h nerf_acts() 12..: MB / 2 dwords per (point, half)), one load
template <int MB>
__device__ __forceinline__ void load_switches(uint32_t (&mw)[4], const float* __restrict__ sw, int64_t p, int h) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(sw) + (p * 2 + h) * (MB / 2);
    if constexpr (MB == 8) { const uint4 v = *reinterpret_cast<const uint4*>(src); mw[0] = v.x; mw[1] = v.y; mw[2] = v.z; mw[3] = v.w; }
    else { const uint2 v = *reinterpret_cast<const uint2*>(src); mw[0] = v.x; mw[1] = v.y; mw[2] = 0u; mw[3] = 0u; }
}

// dA = dX (.) relu'(H): the layer's switch bits (`mw`, load_switches) ANDed onto dX; stores dA rows and leaves them in X.
template <int MB>
__device__ __forceinline__ void relu_bwd_store(const f32x16 (&dX)[8], f32x16 (&X)[8], const uint32_t (&mw)[4],
                                               float* __restrict__ dA, int64_t ld, int64_t p, bool valid, int h) {
    f32x4* drow = reinterpret_cast<f32x4*>(dA + p * ld + 4 * h);
    static_for<MB>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        static_for<4>([&](auto rc) {
            constexpr int rg = decltype(rc)::value;
            f32x4 o;
            o.x = __uint_as_float(__float_as_uint(dX[m][4 * rg + 0]) & relu_switch_of<m, rg, 0>(mw));
            o.y = __uint_as_float(__float_as_uint(dX[m][4 * rg + 1]) & relu_switch_of<m, rg, 1>(mw));
            o.z = __uint_as_float(__float_as_uint(dX[m][4 * rg + 2]) & relu_switch_of<m, rg, 2>(mw));
            o.w = __uint_as_float(__float_as_uint(dX[m][4 * rg + 3]) & relu_switch_of<m, rg, 3>(mw));
            X[m][4 * rg + 0] = o.x; X[m][4 * rg + 1] = o.y; X[m][4 * rg + 2] = o.z; X[m][4 * rg + 3] = o.w;
            drow[m * 8 + rg * 2] = o;
        });
    });
}


// One backward-chain layer with everything but the MFMAs sliced between them (mma_chunk's hooks):
//   pre(m, part)  : accumulator start (0, or s * aux row `piece`) and the global load of the saved row quarter
//                   the epilogue of this block needs (H for ReLU, C for sin) - issued a whole layer ahead, so its
//                   latency sits under ~1000 MFMAs instead of stalling the one resident wave per SIMD;
//   post(m, part) : dA = dX (.) act'(saved), written to HBM and left in X as the next layer's B operand.
// B operands come from bsel; when the last K block reads X[j] with j < MB-1 pass a copy (post overwrites X[j]).
// Row stores are NOT guarded by `valid`: lanes past the end of a partial tile are clamped to its last point, compute what
// that point's own lane computes and store the same bytes to the same address; a guard costs a saveexec / branch /
// restore around every store and stalled the in-order wave for hundreds of cycles each (field_mlp_device.h:fwd_layer).
enum BwdEpi : int { EPI_LINEAR = 0, EPI_RELU = 1, EPI_SIN = 2, EPI_FILM = 3 };

// EPI_SIN / EPI_FILM: saved = the layer's X rows with the cosine's sign in the lowest mantissa bit; the derivative
// factor C = 30 cos(30 u) is rebuilt from them (mi_math.h:dsin30_from_saved) one K block after the quarter's load
// was issued, in the mid slot that issues a later quarter's load.
// EPI_FILM: writes dL/du = dX (.) C rows and leaves dA = gamma (.) dL/du in X; gamma is
// this layer's FiLM row in LDS (`film_row`).  FILM layers also DMA the next epilogue's FiLM row (`next_film_layer`)
// into the other film slot; `issue_slot` is the slot pair index handed to the stage issue (its aux / film
// target is issue_slot ^ 1), `aux_slot` the slot the SCALED start row is read from.
// Row traffic is spread over the layer instead of bursting in one row (32 x 1 KiB per wave inside 2048 cycles
// saturates the CU's vector-memory path and stalls the in-order wave): the saved-row quarters are loaded one per mid
// slot - 8-K-block layers: quarter j in slot 4(j%6) of K block j/6; 4-K-block layers: slot 2(j%8) of K block j/8 - and
// with DEFER the dA rows this layer produces are not stored by its own epilogue but by the NEXT layer's mid slots
// (slot 4(j%4)+2 of K block j/4, resp. 2(j%8)+1 of K block j/8; they sit unchanged in X, that layer's B operand, until
// its last row) - PREV_MB blocks to `prev_dA`.  FiLM layers cannot defer (they store dL/du but carry gamma dL/du).
template <int KB, int MB, int NEXT_AUX, int NEXT_BLOCK, int EPI, bool SCALED, bool FILM = false, bool DEFER = false,
          int PREV_MB = 0, class BSel>
__device__ __forceinline__ void bwd_layer(Ctx& c, int aux_slot, int piece, float s, BSel bsel, f32x16 (&acc)[8],
                                          f32x16 (&X)[8], const float* __restrict__ saved, float* __restrict__ dA,
                                          int64_t ld, int64_t p, bool valid, int issue_slot = -1,
                                          int next_film_layer = 0, const float* film_row = nullptr,
                                          float* __restrict__ prev_dA = nullptr, int64_t prev_ld = 0) {
    static_assert(KB >= 4, "four K blocks are the fewest whose mid slots carry the 32 row quarters");
    static_assert(KB != 8 || MB == 8, "the 8-K-block slot mapping below counts on 24 mid slots per K block (MB = 8)");
    static_assert(!(DEFER && EPI == EPI_FILM), "FiLM layers store dL/du, not what they carry on");
    const int h = c.h;
    // only SCALED layers read the start row: lds_base's opaque asm would otherwise keep a dead address alive across a
    // layer loop (the SirenNeRF chain spilled and reloaded exactly that, once per layer, in front of a stage barrier)
    lds4_t pv = nullptr;
    if constexpr (SCALED) pv = lds_base(c.smem + kLdsAux0 + aux_slot * kLdsAux + h * 16);
    lds4_t pg = nullptr;
    if constexpr (EPI == EPI_FILM) pg = lds_base(film_row + h * 4);
    // Row addresses as a UNIFORM base (the tile's first point: SGPRs) plus one 32-bit byte offset per lane (the lane's
    // point inside the tile: < 128 rows), so a row instruction is `global_load/store ..., v_off, s[base]` and the three
    // row pointers of a layer cost one VGPR, not three 64-bit pairs: round 4 - the SirenNeRF chain kept its per-lane
    // 64-bit row pointers across the layer loop in scratch (6 spilled registers, 3 reloads per layer, each followed by an
    // s_waitcnt vmcnt(0) that drains the wave's row traffic).
    const int64_t tile0 = (int64_t)blockIdx.x * 128;
    const uint32_t lrow = (uint32_t)(p - tile0);
    const RowRef<const f32x4> srow{reinterpret_cast<const char*>(saved + tile0 * ld), (lrow * (uint32_t)ld + 4u * h) * 4u};
    const RowRef<f32x4> drow{reinterpret_cast<char*>(dA + tile0 * ld), (lrow * (uint32_t)ld + 4u * h) * 4u};
    const RowRef<f32x4> prow{reinterpret_cast<char*>(prev_dA + tile0 * prev_ld), (lrow * (uint32_t)prev_ld + 4u * h) * 4u};
    // ReLU layers: `saved` is the layer's SWITCH region (one bit per unit, MB / 2 dwords per lane: one load in the layer's
    // first mid slot) - round 4; sin layers: the saved X rows, a quarter per slot, decoded one K block later
    f32x4 sv[(EPI == EPI_LINEAR || EPI == EPI_RELU) ? 1 : MB * 4];
    uint32_t mw[4] = {0u, 0u, 0u, 0u};
    const auto pre = [&](auto mc, auto pc) {
        constexpr int m = decltype(mc)::value, rg = decltype(pc)::value;
        if constexpr (SCALED) {
            const f32x4 w = pv[piece * 64 + m * 8 + rg];
            acc[m][4 * rg + 0] = w.x * s; acc[m][4 * rg + 1] = w.y * s; acc[m][4 * rg + 2] = w.z * s; acc[m][4 * rg + 3] = w.w * s;
        }       // else: nothing to write, the layer's first MFMAs take srcC = 0 (mma_layer_fn ZERO_START)
    };
    static_assert(!(EPI == EPI_SIN || EPI == EPI_FILM) || KB >= 5, "sin rows are decoded one K block after their load");
    const auto mid = [&](auto kbc, auto sc) {
        constexpr int kb = decltype(kbc)::value, slot = decltype(sc)::value, j = kb * 8 + slot / 2;
#if defined(MI_BWD_PHASES) && MI_BWD_PHASES
        if constexpr (KB == 8) {
            // EXPERIMENT (round 4): loads and stores in separate phases of the layer instead of interleaved in every K block
            // (tools/probes/mfma_store_mix.hip: a row instruction costs 17 / 33 cycles among its own kind, 80-110 in a mix):
            // the previous layer's dA rows go out in K blocks 0..2 (even slots 0..20: 11 per K block), the saved rows come in
            // during K blocks 3..6 (slots 0, 3, .., 21: 8 per K block) and sin rows are decoded one K block later.
            if constexpr (kb <= 2 && (slot & 1) == 0 && slot < 22) {
                constexpr int js = kb * 11 + slot / 2;
                if constexpr (js < PREV_MB * 4) {
                    constexpr int m = js / 4, rg = js % 4;
                    prow[m * 8 + rg * 2] = f32x4{X[m][4 * rg + 0], X[m][4 * rg + 1], X[m][4 * rg + 2], X[m][4 * rg + 3]};
                }
            }
            if constexpr (kb >= 3 && slot % 3 == 0) {
                constexpr int jl = (kb - 3) * 8 + slot / 3;
                if constexpr ((EPI == EPI_SIN || EPI == EPI_FILM) && kb >= 4 && jl - 8 >= 0 && jl - 8 < MB * 4) {
                    sv[jl - 8] = dsin30_from_saved_x4(sv[jl - 8]);
                }
                if constexpr ((EPI == EPI_SIN || EPI == EPI_FILM) && kb <= 6 && jl < MB * 4) sv[jl] = srow[(jl / 4) * 8 + (jl % 4) * 2];
                if constexpr (EPI == EPI_RELU && kb == 3 && slot == 0) load_switches<MB>(mw, saved, p, h);
            }
        } else
#endif
        if constexpr (KB == 8) {
            // 8 K blocks: the row traffic is spread over the whole layer (tools/probes/mfma_store_mix.hip: 8 + 8 quarters
            // per K block run into a mixed read/write ceiling - 80-110 cycles per instruction instead of 17-33 - and
            // half that density costs a quarter as much).  Loads: slots 0, 4, .., 20 of K blocks 1..6, decoded one K block
            // later (K block 7: slots 0 and 4); deferred stores: slots 2, 6, 10, 14 of K blocks 0..7.
            if constexpr ((slot & 3) == 0) {
                // K blocks 1..6 (not 0..5): the saved rows are needed by the layer's LAST row only, and every K block
                // they arrive later is a K block in which 24 registers stay free while all of X is still live (round 3:
                // siren_bwd_kernel spilled 6 registers, and each scratch reload drains the wave's row traffic)
                constexpr int jl = (kb - 1) * 6 + slot / 4;
                if constexpr ((EPI == EPI_SIN || EPI == EPI_FILM) && kb >= 2 && jl - 6 >= 0 && jl - 6 < MB * 4) {
                    sv[jl - 6] = dsin30_from_saved_x4(sv[jl - 6]);
                }
                if constexpr ((EPI == EPI_SIN || EPI == EPI_FILM) && kb >= 1 && kb < 7 && jl < MB * 4) sv[jl] = srow[(jl / 4) * 8 + (jl % 4) * 2];
                if constexpr (EPI == EPI_RELU && kb == 1 && slot == 0) load_switches<MB>(mw, saved, p, h);
            } else if constexpr ((slot & 3) == 2 && slot < 16) {
                constexpr int js = kb * 4 + slot / 4;
                if constexpr (js < PREV_MB * 4) {
                    constexpr int m = js / 4, rg = js % 4;
                    prow[m * 8 + rg * 2] = f32x4{X[m][4 * rg + 0], X[m][4 * rg + 1], X[m][4 * rg + 2], X[m][4 * rg + 3]};
                }
            }
        } else
        if constexpr ((EPI == EPI_SIN || EPI == EPI_FILM) && kb >= 1 && kb < 5 && slot < 16 && (slot & 1) == 0 && j - 8 < MB * 4) {
            sv[j - 8] = dsin30_from_saved_x4(sv[j - 8]);
        }
        if constexpr (KB != 8 && kb < 4 && slot < 16) {          // 4 K blocks: all 16 mid slots of rows 1-2 are needed
            if constexpr ((slot & 1) == 0) {
                if constexpr ((EPI == EPI_SIN || EPI == EPI_FILM) && j < MB * 4) sv[j] = srow[(j / 4) * 8 + (j % 4) * 2];
                if constexpr (EPI == EPI_RELU && j == 0) load_switches<MB>(mw, saved, p, h);
            } else if constexpr (j < PREV_MB * 4) {
                constexpr int m = j / 4, rg = j % 4;
                prow[m * 8 + rg * 2] = f32x4{X[m][4 * rg + 0], X[m][4 * rg + 1], X[m][4 * rg + 2], X[m][4 * rg + 3]};
            }
        }
    };
    const auto post = [&](auto mc, auto pc) {
        constexpr int m = decltype(mc)::value, rg = decltype(pc)::value;
        f32x4 o, g;
        if constexpr (EPI == EPI_FILM) g = pg[m * 8 + rg * 2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float d = acc[m][4 * rg + q];
            if constexpr (EPI == EPI_RELU) {
                const uint32_t on = q == 0 ? relu_switch_of<m, rg, 0>(mw) : q == 1 ? relu_switch_of<m, rg, 1>(mw)
                                  : q == 2 ? relu_switch_of<m, rg, 2>(mw) : relu_switch_of<m, rg, 3>(mw);
                o[q] = __uint_as_float(__float_as_uint(d) & on);
            } else if constexpr (EPI == EPI_SIN || EPI == EPI_FILM) o[q] = sv[m * 4 + rg][q] * d;
            else o[q] = d;
            if constexpr (EPI == EPI_FILM) X[m][4 * rg + q] = o[q] * g[q];
            else X[m][4 * rg + q] = o[q];
        }
        if constexpr (!DEFER) drow[m * 8 + rg * 2] = o;
    };
    // a layer that stores its own dA rows (the chain's last one: there is no next layer to do it) runs its last K block
    // m-major, so the 32 row stores of the epilogue are spread over 128 MFMAs instead of bursting out of the last 32
    // (stamped profile, round 3: the FiLM chain's last layer took 89.5 k cycles where the others take 76-78 k)
    mma_layer_fn<KB, MB, 0, NEXT_AUX, NEXT_BLOCK, FILM, true, !SCALED, !DEFER>(c, issue_slot < 0 ? aux_slot : issue_slot, next_film_layer,
                                                                               NoHook{}, bsel, acc, pre, post, mid);
}

// =========================================================================================
// NeRF / TinyNeRF backward chain (reverse of nerf/nerf.py:75-94)
// =========================================================================================
template <bool TINY>
__global__ __launch_bounds__(256, 1) void nerf_bwd_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    Ctx c = make_ctx_raw(smem, a.packed, nullptr);
    constexpr RegionLayout AL = TINY ? tiny_acts() : nerf_acts();
    constexpr RegionLayout GL = TINY ? tiny_grads() : nerf_grads();
    constexpr int kHeadAux = TINY ? 4 : 3;                 // rgb rows (+ sigma row for TinyNeRF)
    const int64_t P = a.points;
    issue_first_stage<kHeadAux, 32, false>(c, 0, 0, 0);

    const int64_t local = (int64_t)blockIdx.x * 128 + c.wave * 32 + (c.lane & 31);
    const bool valid = local < P;
    const int64_t p = valid ? local : P - 1;
    const f32x4 g = reinterpret_cast<const f32x4*>(a.g_raw)[p];
    const f32x4 o = reinterpret_cast<const f32x4*>(a.raw)[p];
    // heads: rgb = sigmoid(pre), sigma = relu(pre)
    const float d0 = g.x * o.x * (1.f - o.x), d1 = g.y * o.y * (1.f - o.y), d2 = g.z * o.z * (1.f - o.z);
    const float ds = o.w > 0.f ? g.w : 0.f;
    if (valid && c.h == 0)
        reinterpret_cast<f32x4*>(a.grads + (int64_t)region_offset(GL, GL.n - 1) * P)[p] = f32x4{d0, d1, d2, ds};

    f32x16 X[8], acc[8];
    const auto sel_x = [&](auto kb) -> const f32x16& { return X[decltype(kb)::value]; };
    const auto acts = [&](int region) { return a.acts + (int64_t)region_offset(AL, region) * P; };
    const auto grads = [&](int region) { return a.grads + (int64_t)region_offset(GL, region) * P; };
    // ReLU switches (one bit per unit) of layer H_l: region SW0 + l - 1, of the dir layer H_d: the last region
    constexpr int SW0 = TINY ? 7 : 12;
    const auto sw = [&](int l) { return a.acts + ((int64_t)region_offset(AL, SW0) + 8 * (l - 1)) * P; };
    uint32_t hsw[4];
    load_switches<4>(hsw, acts(AL.n - 1), p, c.h);
    MI_STAMP(a, 0);

    __syncthreads();                                        // head rows (and K block 0) have landed
    {   // dH_d = W_rgb^T d_pre_rgb   (128 features)
        const lds4_t pw = lds_base(smem + kLdsAux0 + c.h * 16);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x4 w0 = pw[0 * 64 + m * 8 + rg], w1 = pw[1 * 64 + m * 8 + rg], w2 = pw[2 * 64 + m * 8 + rg];
                acc[m][4 * rg + 0] = fmaf(w2.x, d2, fmaf(w1.x, d1, w0.x * d0));
                acc[m][4 * rg + 1] = fmaf(w2.y, d2, fmaf(w1.y, d1, w0.y * d0));
                acc[m][4 * rg + 2] = fmaf(w2.z, d2, fmaf(w1.z, d1, w0.z * d0));
                acc[m][4 * rg + 3] = fmaf(w2.w, d2, fmaf(w1.w, d1, w0.w * d0));
            }
    }
    relu_bwd_store<4>(acc, X, hsw, grads(TINY ? 4 : 9), 128, p, valid, c.h);                    // dA of the dir layer
    MI_STAMP(a, 1);

    int slot = 0;
    if constexpr (!TINY) {
        // layers_dir[1]^T: dG = W[:, :256]^T dA (K = 128); layers_dir[0] is linear: dA = dG.  The B operand is a
        // copy: the epilogue rewrites X[0..3] while the last K block still reads X[3].
        f32x16 Bd[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) Bd[m] = X[m];
        const auto sel_d = [&](auto kb) -> const f32x16& { return Bd[decltype(kb)::value]; };
        bwd_layer<4, 8, 1, 32, EPI_LINEAR, false, false, true, 0>(c, slot, 0, 0.f, sel_d, acc, X, nullptr, grads(8), 256, p, valid);
        MI_STAMP(a, 2);
        slot ^= 1;
        // layers_dir[0]^T, plus the sigma head's contribution to dH8; dA7 = dH8 (.) [H8>0]
        bwd_layer<8, 8, 0, 32, EPI_RELU, true, false, true, 8>(c, slot, 0, ds, sel_x, acc, X, sw(8), grads(7), 256, p, valid, -1, 0, nullptr, grads(8), 256);
        MI_STAMP(a, 3);
        slot ^= 1;
#ifdef MI_PROFILE_STAMPS
        if (a.stamps) c.rowst = a.stamps + (int64_t)blockIdx.x * 128 + 32;                     // rows of L7^T: 32..64
#endif
        bwd_layer<8, 8, 0, 32, EPI_RELU, false, false, true, 8>(c, slot, 0, 0.f, sel_x, acc, X, sw(7), grads(6), 256, p, valid, -1, 0, nullptr, grads(7), 256);  // L7^T
#ifdef MI_PROFILE_STAMPS
        if (c.rowst) { MI_ROW_STAMP(c); }
        c.rowst = nullptr;
#endif
        MI_STAMP(a, 4);
        bwd_layer<8, 8, 0, 32, EPI_RELU, false, false, true, 8>(c, slot, 0, 0.f, sel_x, acc, X, sw(6), grads(5), 256, p, valid, -1, 0, nullptr, grads(6), 256);  // L6^T
        MI_STAMP(a, 5);
        bwd_layer<8, 8, 0, 32, EPI_RELU, false, false, true, 8>(c, slot, 0, 0.f, sel_x, acc, X, sw(5), grads(4), 256, p, valid, -1, 0, nullptr, grads(5), 256);  // L5^T (h part)
        MI_STAMP(a, 6);
#pragma unroll 1
        for (int l = 4; l >= 2; --l)                                                           // L4^T .. L2^T
            bwd_layer<8, 8, 0, 32, EPI_RELU, false, false, true, 8>(c, slot, 0, 0.f, sel_x, acc, X, sw(l),
                                                                    a.grads + (int64_t)(256 * (l - 1)) * P, 256, p, valid, -1, 0, nullptr,
                                                                    a.grads + (int64_t)(256 * l) * P, 256);
        MI_STAMP(a, 7);                                                                        // after L4^T .. L2^T
        bwd_layer<8, 8, 0, 0, EPI_RELU, false, false, false, 8>(c, slot, 0, 0.f, sel_x, acc, X, sw(1), grads(0), 256, p, valid, -1, 0, nullptr, grads(1), 256);   // L1^T
        MI_STAMP(a, 8);
    } else {
        // dir layer^T with the sigma head's contribution to dH4 (sigma row is aux piece 3 of slot 0)
        f32x16 Bd[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) Bd[m] = X[m];
        const auto sel_d = [&](auto kb) -> const f32x16& { return Bd[decltype(kb)::value]; };
        bwd_layer<4, 8, 0, 32, EPI_RELU, true, false, true, 0>(c, 0, 3, ds, sel_d, acc, X, sw(4), grads(3), 256, p, valid);
        bwd_layer<8, 8, 0, 32, EPI_RELU, false, false, true, 8>(c, slot, 0, 0.f, sel_x, acc, X, sw(3), grads(2), 256, p, valid, -1, 0, nullptr, grads(3), 256);  // L3^T
        bwd_layer<8, 8, 0, 32, EPI_RELU, false, false, true, 8>(c, slot, 0, 0.f, sel_x, acc, X, sw(2), grads(1), 256, p, valid, -1, 0, nullptr, grads(2), 256);  // L2^T
        bwd_layer<8, 8, 0, 0, EPI_RELU, false, false, false, 8>(c, slot, 0, 0.f, sel_x, acc, X, sw(1), grads(0), 256, p, valid, -1, 0, nullptr, grads(1), 256);   // L1^T
    }
}

// dA = dX (.) C with C = 30 cos(30 A) rebuilt from the saved (sign-encoded) X rows; stores dA rows, leaves them in X.
template <int MB>
__device__ __forceinline__ void sin_bwd_store(const f32x16 (&dX)[8], f32x16 (&X)[8], const f32x4 (&xsv)[MB * 4],
                                              float* __restrict__ dA, int64_t ld, int64_t p, bool valid, int h) {
    f32x4* drow = reinterpret_cast<f32x4*>(dA + p * ld + 4 * h);
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const f32x4 o = dsin30_from_saved_x4(xsv[m * 4 + rg]) *
                            f32x4{dX[m][4 * rg + 0], dX[m][4 * rg + 1], dX[m][4 * rg + 2], dX[m][4 * rg + 3]};
            X[m][4 * rg + 0] = o.x; X[m][4 * rg + 1] = o.y; X[m][4 * rg + 2] = o.z; X[m][4 * rg + 3] = o.w;
            drow[m * 8 + rg * 2] = o;
        }
}

// =========================================================================================
// SirenNeRF backward chain (reverse of nerf/nerf.py:153-170); same shape as NeRF's, sin derivatives
// =========================================================================================
__global__ __launch_bounds__(256, 1) void siren_bwd_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    Ctx c = make_ctx_raw(smem, a.packed, nullptr);
    constexpr RegionLayout AL = siren_acts();
    constexpr RegionLayout GL = siren_grads();
    const int64_t P = a.points;
    issue_first_stage<3, 32, false>(c, 0, 0, 0);

    const int64_t local = (int64_t)blockIdx.x * 128 + c.wave * 32 + (c.lane & 31);
    const bool valid = local < P;
    const int64_t p = valid ? local : P - 1;
    const f32x4 g = reinterpret_cast<const f32x4*>(a.g_raw)[p];
    const f32x4 o = reinterpret_cast<const f32x4*>(a.raw)[p];
    const float d0 = g.x * o.x * (1.f - o.x), d1 = g.y * o.y * (1.f - o.y), d2 = g.z * o.z * (1.f - o.z);
    const float ds = o.w > 0.f ? g.w : 0.f;
    if (valid && c.h == 0)
        reinterpret_cast<f32x4*>(a.grads + (int64_t)region_offset(GL, GL.n - 1) * P)[p] = f32x4{d0, d1, d2, ds};

    f32x16 X[8], acc[8];
    const auto sel_x = [&](auto kb) -> const f32x16& { return X[decltype(kb)::value]; };
    const auto acts = [&](int region) { return a.acts + (int64_t)region_offset(AL, region) * P; };
    const auto grads = [&](int region) { return a.grads + (int64_t)region_offset(GL, region) * P; };
    f32x4 xsv[16];
    load_saved_row<4>(xsv, acts(10), 128, p, c.h);

    __syncthreads();
    {   // dX_d = W_rgb^T d_pre_rgb (128 features)
        const lds4_t pw = lds_base(smem + kLdsAux0 + c.h * 16);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x4 w0 = pw[0 * 64 + m * 8 + rg], w1 = pw[1 * 64 + m * 8 + rg], w2 = pw[2 * 64 + m * 8 + rg];
                acc[m][4 * rg + 0] = fmaf(w2.x, d2, fmaf(w1.x, d1, w0.x * d0));
                acc[m][4 * rg + 1] = fmaf(w2.y, d2, fmaf(w1.y, d1, w0.y * d0));
                acc[m][4 * rg + 2] = fmaf(w2.z, d2, fmaf(w1.z, d1, w0.z * d0));
                acc[m][4 * rg + 3] = fmaf(w2.w, d2, fmaf(w1.w, d1, w0.w * d0));
            }
    }
    sin_bwd_store<4>(acc, X, xsv, grads(9), 128, p, valid, c.h);                               // dA layers_dir.1 (X_d rows)
    int slot = 0;
    {   // layers_dir.1^T (h part); layers_dir.0 is linear: dA = dG.  B operand copied (see nerf_bwd_kernel).
        f32x16 Bd[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) Bd[m] = X[m];
        const auto sel_d = [&](auto kb) -> const f32x16& { return Bd[decltype(kb)::value]; };
        bwd_layer<4, 8, 1, 32, EPI_LINEAR, false, false, true, 0>(c, slot, 0, 0.f, sel_d, acc, X, nullptr, grads(8), 256, p, valid);
    }
    slot ^= 1;
    // layers_dir.0^T + sigma head; dA7 = dX8 (.) C8, C_l rebuilt from the saved X_l rows (acts region l)
    bwd_layer<8, 8, 0, 32, EPI_SIN, true, false, true, 8>(c, slot, 0, ds, sel_x, acc, X, acts(8), grads(7), 256, p, valid, -1, 0, nullptr, grads(8), 256);
#pragma unroll 1
    for (int l = 7; l >= 2; --l)                                                                // L7^T .. L2^T: dA_{l-1} = dX_l (.) C_l
        bwd_layer<8, 8, 0, 32, EPI_SIN, false, false, true, 8>(c, slot, 0, 0.f, sel_x, acc, X, a.acts + (int64_t)(8 + 256 * (l - 1)) * P,
                                                               a.grads + (int64_t)(256 * (l - 1)) * P, 256, p, valid, -1, 0, nullptr,
                                                               a.grads + (int64_t)(256 * l) * P, 256);
    bwd_layer<8, 8, 0, 0, EPI_SIN, false, false, false, 8>(c, slot, 0, 0.f, sel_x, acc, X, acts(1), grads(0), 256, p, valid, -1, 0, nullptr, grads(1), 256);   // L1^T: dA0 = dX1 (.) C1
}

// =========================================================================================
// FilmSirenNeRF backward chain (reverse of pi_GAN/modules.py:101-118).  Layer l: A = W x + b,
// u = gamma A + beta, X = sin(30 u).  dL/du = dX (.) C is what the chain writes (grads region l); it carries
// dA = gamma (.) dL/du to the next layer in registers.  d gamma / d beta / dW / db all come out of the per-image
// sums T_g = sum_p dL/du X_{l-1}^T and s_g = sum_p dL/du (launch_field_backward), so the chain needs neither
// the linear output A nor any cross-lane reduction.
// =========================================================================================
// dU = dX (.) C with C = 30 cos(30 u) rebuilt from the saved (sign-encoded) X rows; leaves dU in X.  Nothing is stored here:
// the first chain layer's mid slots write these rows (film_chain_layer), like every other layer's.
// `cv`: the point's saved X_8 row, loaded by the caller at the top of the kernel (its latency then overlaps the first
// weight stage's DMA and the head gradients instead of following them).
template <int MB>
__device__ __forceinline__ void film_bwd_first(const f32x16 (&dX)[8], f32x16 (&X)[8], const f32x4 (&cv)[32], float w0sq) {
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const f32x4 c = dsin_w_from_saved_x4(cv[m * 4 + rg], w0sq);
            const f32x4 o = c * f32x4{dX[m][4 * rg + 0], dX[m][4 * rg + 1], dX[m][4 * rg + 2], dX[m][4 * rg + 3]};
            X[m][4 * rg + 0] = o.x; X[m][4 * rg + 1] = o.y; X[m][4 * rg + 2] = o.z; X[m][4 * rg + 3] = o.w;
        }
}

// One FiLM chain layer j: dX_j = W_{j+1}^T (gamma_{j+1} (.) dU_{j+1}), dU_j = dX_j (.) C_j.
//
// On entry X holds dL/du_{j+1} itself (round 2 carried gamma (.) dL/du in X and parked dL/du in the C registers until the
// next layer stored it: at the layer boundary 128 + 128 such registers were live next to the A fragments, more than the
// 256 architectural VGPRs - the allocator shuffled ~330 values per layer through AGPRs (v_accvgpr_write / read pairs) and
// spilled 25 registers to scratch, and every scratch reload is followed by an s_waitcnt vmcnt(0) that drains ALL the
// wave's outstanding row loads and stores: tools/isa_stats.py).  Now, one K block ahead of its use as
// the B operand, each block of X is first STORED (the dU_{j+1} rows, a quarter per mid slot) and then scaled in place by
// gamma_{j+1} (the FiLM row in LDS slot `gamma_row`, read one slot earlier): block kb + 1 in K block kb's slots
// 1|2|3, 5|6|7, 9|10|11, 13|14|15 (read gamma | store | scale), block 0 in the layer's first hooks.  The same 128
// multiplies per layer as before, no second copy of anything; X[kb] is dead after K block kb, so X shrinks by a block
// per K block while the C quarters arrive (slots 0, 4, .., 20 of K blocks 1..6, decoded one K block later).
// The epilogue leaves dU_j = C_j (.) dX_j in X (LAST: and stores it, there being no next layer to do it).
// FILM_NEXT: the last stage also DMAs FiLM row `next_film_layer` (= j, what layer j - 1 multiplies by) into the film
// slot `issue_slot ^ 1`, the one not being read.
template <int NEXT_BLOCK, bool SCALED, bool FILM_NEXT, bool LAST>
__device__ __forceinline__ void film_chain_layer(Ctx& c, int piece, float s, f32x16 (&acc)[8], f32x16 (&X)[8],
                                                 f32x4 (&ring)[32], const float* __restrict__ C_rows,
                                                 float* __restrict__ dU, float* __restrict__ prev_dU, int64_t p,
                                                 int issue_slot, int next_film_layer, const float* gamma_row) {
    const int h = c.h;
    const lds4_t pv = lds_base(c.smem + kLdsAux0 + h * 16);          // aux slot 0 (the sigma head's row for SCALED)
    const lds4_t pg = lds_base(gamma_row + h * 4);
    const f32x4* srow = reinterpret_cast<const f32x4*>(C_rows + p * 256 + 4 * h);
    f32x4* drow = reinterpret_cast<f32x4*>(dU + p * 256 + 4 * h);
    f32x4* prow = reinterpret_cast<f32x4*>(prev_dU + p * 256 + 4 * h);
    f32x4 gq;                                                        // gamma quarter read one slot ahead of its use
    const auto pre = [&](auto mc, auto pc) {
        constexpr int m = decltype(mc)::value, rg = decltype(pc)::value;
        if constexpr (SCALED) {
            const f32x4 w = pv[piece * 64 + m * 8 + rg];
            acc[m][4 * rg + 0] = w.x * s; acc[m][4 * rg + 1] = w.y * s; acc[m][4 * rg + 2] = w.z * s; acc[m][4 * rg + 3] = w.w * s;
        }       // else: nothing to write, the layer's first MFMAs take srcC = 0 (mma_layer_fn ZERO_START)
        if constexpr (m == 0) {                                      // before the layer's first MFMA: block 0 of X
            const f32x4 g = pg[rg * 2];
            prow[rg * 2] = f32x4{X[0][4 * rg + 0], X[0][4 * rg + 1], X[0][4 * rg + 2], X[0][4 * rg + 3]};
            const f32x4 t = f32x4{X[0][4 * rg + 0], X[0][4 * rg + 1], X[0][4 * rg + 2], X[0][4 * rg + 3]} * g;
            X[0][4 * rg + 0] = t.x; X[0][4 * rg + 1] = t.y; X[0][4 * rg + 2] = t.z; X[0][4 * rg + 3] = t.w;
        }
    };
    const auto mid = [&](auto kbc, auto sc) {
        constexpr int kb = decltype(kbc)::value, slot = decltype(sc)::value;
        if constexpr ((slot & 3) == 0) {
            // C quarters: quarter j = 6 (kb - 1) + slot / 4 in slots 0, 4, .., 20 of K blocks 1..6, decoded one K block later
            // (K block 7 offers slots 0..15: the last two quarters, loaded in K block 6, are decoded in its slots 0 and 4)
            constexpr int j = (kb - 1) * 6 + slot / 4;
            if constexpr (kb >= 2 && j - 6 >= 0 && j - 6 < 32) {
                ring[j - 6] = dsin_w_from_saved_x4(ring[j - 6], c.w0sq);
            }
            if constexpr (kb >= 1 && kb < 7 && j < 32) ring[j] = srow[(j / 4) * 8 + (j % 4) * 2];
        } else if constexpr (kb < 7 && slot < 16) {
            // block kb + 1 of X, quarter rg = slot / 4: gamma read | dU row store | scale in place
            constexpr int m = kb + 1, rg = slot / 4;
            if constexpr ((slot & 3) == 1) gq = pg[m * 8 + rg * 2];
            if constexpr ((slot & 3) == 2)
                prow[m * 8 + rg * 2] = f32x4{X[m][4 * rg + 0], X[m][4 * rg + 1], X[m][4 * rg + 2], X[m][4 * rg + 3]};
            if constexpr ((slot & 3) == 3) {
                const f32x4 t = f32x4{X[m][4 * rg + 0], X[m][4 * rg + 1], X[m][4 * rg + 2], X[m][4 * rg + 3]} * gq;
                X[m][4 * rg + 0] = t.x; X[m][4 * rg + 1] = t.y; X[m][4 * rg + 2] = t.z; X[m][4 * rg + 3] = t.w;
            }
        }
    };
    const auto post = [&](auto mc, auto pc) {
        constexpr int m = decltype(mc)::value, rg = decltype(pc)::value, j = m * 4 + rg;
        const f32x4 o = ring[j] * f32x4{acc[m][4 * rg + 0], acc[m][4 * rg + 1], acc[m][4 * rg + 2], acc[m][4 * rg + 3]};
        X[m][4 * rg + 0] = o.x; X[m][4 * rg + 1] = o.y; X[m][4 * rg + 2] = o.z; X[m][4 * rg + 3] = o.w;
        if constexpr (LAST) drow[m * 8 + rg * 2] = o;
    };
    const auto sel_x = [&](auto kb) -> const f32x16& { return X[decltype(kb)::value]; };
    mma_layer_fn<8, 8, 0, 0, NEXT_BLOCK, FILM_NEXT, true, !SCALED, LAST>(c, issue_slot, next_film_layer, NoHook{}, sel_x, acc, pre, post, mid);
}

template <bool USE_DIR>
__global__ __launch_bounds__(256, 1) void film_bwd_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int64_t group = blockIdx.x / a.tiles_per_group;
    const int64_t tile = blockIdx.x % a.tiles_per_group;
    Ctx c = make_ctx_raw(smem, a.packed, a.film + group * (kFilmLayers * kFilmRow));
    // fl(w_0^2) of the module's w_0 (pi_GAN/modules.py:11,73) from the backward stream's trailer: the rebuilt derivative
    // factor is +-sqrt(w_0^2 (1 - X^2)); a uniform (scalar) load
    constexpr int kBody = USE_DIR ? packed_body_floats(h_tabb_film) : packed_body_floats(h_tabb_film_nodir);
    c.w0sq = a.packed[kBody + 1];
    const int64_t P = a.points;
    // rgb head rows x3, sigma row; K blocks 0-1 of hidden_layer_rgb^T; FiLM row 8 -> film slot 0 (row r lives in slot (8 - r) & 1)
    issue_first_stage<4, 32, true>(c, 0, 0, 8);

    const int64_t local = tile * 128 + c.wave * 32 + (c.lane & 31);
    const bool valid = local < a.points_per_group;
    const int64_t p = group * a.points_per_group + (valid ? local : a.points_per_group - 1);
    const f32x4 g = reinterpret_cast<const f32x4*>(a.g_raw)[p];
    const f32x4 o = reinterpret_cast<const f32x4*>(a.raw)[p];
    const float d0 = g.x * o.x * (1.f - o.x), d1 = g.y * o.y * (1.f - o.y), d2 = g.z * o.z * (1.f - o.z);
    const float ds = o.w > 0.f ? g.w : 0.f;
    if (valid && c.h == 0) reinterpret_cast<f32x4*>(a.grads + (int64_t)(9 * 256) * P)[p] = f32x4{d0, d1, d2, ds};

    f32x16 X[8], acc[8];
    const auto film_row = [&](int r) { return smem + kLdsFilm0 + ((8 - r) & 1) * kFilmRow; };       // FiLM row r's LDS slot
    const auto C = [&](int l) { return a.acts + (int64_t)(8 + 256 * l) * P; };    // saved X_l rows: C_l is rebuilt from them
    const auto dU = [&](int l) { return a.grads + (int64_t)(256 * l) * P; };
    f32x4 ring[32];                                            // a layer's C quarters, loaded a K block or more ahead
    {   // the first epilogue's rows (X_8) now, while the first weight stage is still on its way
        const f32x4* crow = reinterpret_cast<const f32x4*>(C(8) + p * 256 + 4 * c.h);
#pragma unroll
        for (int j = 0; j < 32; ++j) ring[j] = crow[(j / 4) * 8 + (j % 4) * 2];
    }

    __syncthreads();
    {   // dX_8 = W_rgb^T d_pre_rgb (256 features)
        const lds4_t pw = lds_base(smem + kLdsAux0 + c.h * 16);
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x4 w0 = pw[0 * 64 + m * 8 + rg], w1 = pw[1 * 64 + m * 8 + rg], w2 = pw[2 * 64 + m * 8 + rg];
                acc[m][4 * rg + 0] = fmaf(w2.x, d2, fmaf(w1.x, d1, w0.x * d0));
                acc[m][4 * rg + 1] = fmaf(w2.y, d2, fmaf(w1.y, d1, w0.y * d0));
                acc[m][4 * rg + 2] = fmaf(w2.z, d2, fmaf(w1.z, d1, w0.z * d0));
                acc[m][4 * rg + 3] = fmaf(w2.w, d2, fmaf(w1.w, d1, w0.w * d0));
            }
    }
    MI_STAMP(a, 0);
    film_bwd_first<8>(acc, X, ring, c.w0sq);                                                             // hidden_layer_rgb: X = dU_8
    MI_STAMP(a, 1);
    // Chain layer j multiplies by FiLM row j + 1 (slot (7 - j) & 1) and DMAs row j - what layer j - 1 multiplies by -
    // into the other slot.  j = 7 starts from the sigma head's row (aux slot 0, piece 3).
#if defined(MI_PROFILE_STAMPS) && defined(MI_STAMP_LAYER) && MI_STAMP_LAYER == 7
    if (a.stamps) c.rowst = a.stamps + (int64_t)blockIdx.x * 128 + 32;                          // rows of layer 7: 32..64
#endif
    film_chain_layer<32, true, true, false>(c, 3, ds, acc, X, ring, C(7), dU(7), dU(8), p, 0, 7, film_row(8));
#if defined(MI_PROFILE_STAMPS) && defined(MI_STAMP_LAYER) && MI_STAMP_LAYER == 7
    if (c.rowst) { MI_ROW_STAMP(c); }
    c.rowst = nullptr;
#endif
    MI_STAMP(a, 2);
#pragma unroll 1
    for (int j = 6; j >= 1; --j) {                                                               // hidden_layers[5..0]
#ifdef MI_PROFILE_STAMPS
#if !defined(MI_STAMP_LAYER) || MI_STAMP_LAYER != 7
        if (j == 4 && a.stamps) c.rowst = a.stamps + (int64_t)blockIdx.x * 128 + 32;            // rows of layer j = 4: 32..64
#endif
#endif
        film_chain_layer<32, false, true, false>(c, 0, 0.f, acc, X, ring, C(j), dU(j), dU(j + 1), p, (7 - j) & 1, j, film_row(j + 1));
#ifdef MI_PROFILE_STAMPS
        if (c.rowst) { MI_ROW_STAMP(c); }
        c.rowst = nullptr;
        if (threadIdx.x == 0 && a.stamps) a.stamps[(int64_t)blockIdx.x * 128 + 3 + (6 - j)] = __builtin_amdgcn_s_memtime();   // 3..8
#endif
    }
    film_chain_layer<0, false, false, true>(c, 0, 0.f, acc, X, ring, C(0), dU(0), dU(1), p, 1, 0, film_row(1));   // input_layer
    MI_STAMP(a, 9);
}

// FiLM layer finishing for ONE image (group) g.  T[f][k] = sum_{p in g} dL/du[p][f] X[p][k] (tk = 256, or the 3
// raw-input columns), s[f] = sum_{p in g} dL/du[p][f].  Row f per workgroup, column k per thread:
//   dW[f][col0+k]  (+)= gamma[f] T[f][k]                       (+= for g > 0: images are summed in order)
//   d gamma_g[f]   (+)= sum_k W[f][col0+k] T[f][k]  (+ b[f] s[f] with the bias part: A = W x + b)
//   d beta_g[f]      = s[f];   db[f] (+)= gamma[f] s[f]        (bias part only)
constexpr int kMaxFinishJobs = 10;          // an image's nine FiLM layers + the dir columns of hidden_layer_rgb
struct FinishJob {
    const float* T; const float* s; const float* W; const float* b; const float* film_row;
    float* dW; float* db; float* dfilm_row;
    int tk, w_ld, col0, bias_part;
};
struct FinishBatch { FinishJob job[kMaxFinishJobs]; int first_group; };

// grid = (256 rows, jobs): every finishing job of one image in ONE launch (round 2: ten launches of 4.6 us per image,
// 320 per C4 step).  A job that adds to a d gamma another job of the same launch WRITES (hidden_layer_rgb's dir columns
// after its h columns: bias_part = 0) runs in the same workgroup right after it - see launch_field_backward.
__global__ __launch_bounds__(256) void film_finish_kernel(FinishBatch fb) {
    __shared__ float red[4];
    const int f = blockIdx.x, k = threadIdx.x;
    const int first_group = fb.first_group;
    // blockIdx.y indexes the jobs with a bias part; a job without one (it ADDS to the d gamma row of the job before it
    // in the table) is chained behind its predecessor here, so the two never race
    int j = 0;
    for (int seen = -1; j < kMaxFinishJobs; ++j)
        if (fb.job[j].T && fb.job[j].bias_part && ++seen == (int)blockIdx.y) break;
    if (j >= kMaxFinishJobs) return;
    for (; j < kMaxFinishJobs && fb.job[j].T; ++j) {
        const FinishJob& q = fb.job[j];
        const float gamma = q.film_row[f];
        float dot = 0.f;
        if (k < q.tk) {
            const float t = q.T[f * q.tk + k];
            float* w = q.dW + (int64_t)f * q.w_ld + q.col0 + k;
            *w = first_group ? gamma * t : *w + gamma * t;
            dot = q.W[(int64_t)f * q.w_ld + q.col0 + k] * t;
        }
        for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
        __syncthreads();                                   // red[] of the previous job of this chain has been consumed
        if ((k & 63) == 0) red[k >> 6] = dot;
        __syncthreads();
        if (k == 0) {
            float dg = ((red[0] + red[1]) + red[2]) + red[3];
            if (q.bias_part) {
                dg += q.b[f] * q.s[f];
                q.dfilm_row[f] = dg;
                q.dfilm_row[256 + f] = q.s[f];
                q.db[f] = first_group ? gamma * q.s[f] : q.db[f] + gamma * q.s[f];
            } else {
                q.dfilm_row[f] += dg;
            }
        }
        if (j + 1 < kMaxFinishJobs && fb.job[j + 1].T && fb.job[j + 1].bias_part) break;     // the next job is another chain's head
    }
}

// =========================================================================================
// dW GEMM over points: dW[f][k] = sum_p dA[p][f] X[p][k].  Features sit on the MFMA lanes: lane i of wave-tile
// (wm, wk) owns dA features 128*wm + 4*i + c (c = the four 32-wide row blocks) and X features 32*CB*wk + CB*i + d,
// acc[c][d] (+)= A_c (x) B_d per point pair.  Workgroup = 4 waves = tiles (WM x WK) x k-split KS = 4 / (WM*WK).
// partial record (slab) = [TM][TK] row-major tile, then (with_bias) the TM column sums of dA.
// Rows are staged through LDS by DMA (buffer_load ... lds) rather than loaded per lane (the first version: 4 ms
// of a 60 ms step slower):
// a stage = 32 points = one contiguous run of rows of each operand ([point][feature] rows with lda == TM and
// ldx == TK, which holds for every job), double-buffered; the next stage's 1 KiB pieces are issued one per 8
// MFMAs during the first half of the current stage, so they have half a stage (>= 3 us) to land.  Rows past the
// slab's end read as zero (buffer bounds), so ragged slabs need no masking.  Operand fetch is one ds_read_b128
// (conflict-free: 16 consecutive lanes read 16 consecutive float4) per operand and point pair.
constexpr int kGemmStagePts = 32;

// Up to kMaxGemmJobs GEMMs of one tile shape in ONE launch (grid.y = job): every job contracts the same P points, so
// a training step's nine 256x256 weight gradients are one launch of 9 x slabs workgroups instead of nine launches
// of `slabs` - at the 1024-ray batch of nerf/train_nerf.py the step is launch-bound otherwise.
constexpr int kMaxGemmJobs = 12;
struct GemmBatch {
    const float* dA[kMaxGemmJobs];
    const float* X[kMaxGemmJobs];
    float* partial[kMaxGemmJobs];      // one record per slab of TM*TK (+ TM bias sums) floats
    int with_bias[kMaxGemmJobs];
};

// LDS floats one k-split wave parks for the merge (its 4 x CB accumulator blocks + the bias sums), and the dynamic LDS a
// launch needs: the two stage buffers, or the merge area of the (KS - 1) x TILES parked waves if that is larger.
constexpr int gemm_merge_wave_floats(int CB) { return (4 * CB * 16 + 4) * 64; }
// Stage buffers in the ring.  A 256x256 tile runs 256 MFMAs per wave and stage (6.9 us) on 64 KiB of rows: one stage in
// flight while the other is consumed hides the memory latency, and two buffers are all that fit.  The NARROW tiles run 16 -
// 128 MFMAs (0.4 - 3.4 us) on 20 - 48 KiB: with one stage in flight a CU has ~40 KiB outstanding against a loaded HBM latency
// of ~2 us - Little's law caps the chip at ~2.5 TB/s, which is what round 3 measured for them (2.0 - 2.5 TB/s of algorithmic
// bytes at 0.25 - 0.65 MFMA-busy) while they ask for 3.6 - 6 TB/s.  Their stages are small, so the ring holds one or two more.
template <int CB, int WM, int WK>
constexpr int gemm_ring() {
    constexpr size_t stage = (size_t)kGemmStagePts * (128 * WM + 32 * CB * WK) * sizeof(float);
    return stage >= 64 * 1024 ? 2 : (stage > 20 * 1024 ? 3 : 4);
}
template <int CB, int WM, int WK>
constexpr size_t gemm_lds_bytes() {
    constexpr int TILES = WM * WK, KS = 4 / TILES, TM = 128 * WM, TK = 32 * CB * WK;
    constexpr size_t stage = (size_t)gemm_ring<CB, WM, WK>() * kGemmStagePts * (TM + TK) * sizeof(float);
    constexpr size_t merge = (size_t)(KS - 1) * TILES * gemm_merge_wave_floats(CB) * sizeof(float);
    return stage > merge ? stage : merge;
}

template <int CB, int WM, int WK>
__global__ __launch_bounds__(256, 1) void dw_gemm_kernel(GemmBatch jobs, int64_t P, int slab_pts) {
    const float* __restrict__ dA = jobs.dA[blockIdx.y];
    const float* __restrict__ X = jobs.X[blockIdx.y];
    float* __restrict__ partial = jobs.partial[blockIdx.y];
    const int with_bias = jobs.with_bias[blockIdx.y];
    constexpr int TILES = WM * WK, KS = 4 / TILES, TM = 128 * WM, TK = 32 * CB * WK, T = kGemmStagePts;
    constexpr int PA = T * TM / 256, PB = T * TK / 256;              // 1 KiB pieces per stage and operand
    constexpr int NPW = (PA + PB + 3) / 4;                           // pieces per wave and stage
    constexpr int STEPS = (T / 2) / KS;                              // point pairs per wave and stage
    constexpr int PPS = (NPW + STEPS / 2 - 1) / (STEPS / 2);         // pieces issued per step in the first half
    constexpr int STAGE = T * (TM + TK);                             // floats per stage buffer
    typedef float bvec __attribute__((ext_vector_type(CB)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int tile = wave % TILES, ks = wave / TILES;
    const int wm = tile / WK, wk = tile % WK;
    const int64_t p0 = (int64_t)blockIdx.x * slab_pts;
    const int64_t p1 = p0 + slab_pts < P ? p0 + slab_pts : P;
    const int n_stages = (int)((p1 - p0 + T - 1) / T);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(dA + p0 * TM), 0, (int)((p1 - p0) * TM * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(X + p0 * TK), 0, (int)((p1 - p0) * TK * 4), 0x00020000);
    // piece q (0..PA+PB-1) of stage st into buffer b; wave w owns pieces w, w+4, ...
    const auto issue = [&](int st, int b, int k) {
        const int q = wave + 4 * k;
        if (q < PA)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (MI_LDS void*)(smem + b * STAGE + q * 256), 16, lane * 16,
                                                     st * (T * TM * 4) + q * 1024, 0, 0);
        else if (q < PA + PB)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (MI_LDS void*)(smem + b * STAGE + T * TM + (q - PA) * 256), 16,
                                                     lane * 16, st * (T * TK * 4) + (q - PA) * 1024, 0, 0);
    };

    f32x16 acc[4][CB];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < CB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][d][r] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

    // Ring of NB stage buffers: stages st+1 .. st+NB-2 are in flight while stage st is consumed, and stage st+NB-1 is
    // issued during it (into the buffer stage st-1 just left).  Every wave issues exactly NPW pieces per stage, ALWAYS -
    // stages past the slab's end read zeros through the buffer bounds and are never consumed - so "stage st has landed" is
    // `s_waitcnt vmcnt((NB - 2) * NPW)`: the wave's vector-memory operations complete in order and nothing else is issued
    // in the loop.
    constexpr int NB = gemm_ring<CB, WM, WK>();
    static_assert((PA + PB) % 4 == 0, "every wave must issue the same number of pieces per stage (the vmcnt arithmetic)");
    static_assert((NB - 2) * NPW < 64, "vmcnt is a 6-bit counter");
#pragma unroll
    for (int sp = 0; sp < NB - 1; ++sp)
#pragma unroll
        for (int k = 0; k < NPW; ++k) issue(sp, sp, k);
    int b = 0;
    for (int st = 0; st < n_stages; ++st) {
        // stage st has landed in every wave's view: own pieces by the count, the others' by the barrier
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NB - 2) * NPW) : "memory");
        const int bn = b == 0 ? NB - 1 : b - 1;                      // the buffer stage st - 1 used = (st + NB - 1) % NB
        // this wave's first point pair of the stage; step sidx reads rows 2 KS sidx further on (an immediate offset)
        const float* As = smem + b * STAGE + 128 * wm + 4 * i + (2 * ks + h) * TM;
        const float* Bs = smem + b * STAGE + T * TM + 32 * CB * wk + CB * i + (2 * ks + h) * TK;
        // Operands one step ahead: the reads of step s + 1 are issued at the top of step s and pinned there, so their LDS
        // latency runs under the 4 CB MFMAs of step s.  (Left to the scheduler, the A operand of a step was read three MFMAs
        // before the step began with an s_waitcnt lgkmcnt(0) right behind it - the matrix pipe drained at every step: the
        // ISA of round 2's kernel, tools/isa_stats.py.)  Only the first step of a stage waits for LDS.
        f32x4 av = *reinterpret_cast<const f32x4*>(As);
        bvec bv = *reinterpret_cast<const bvec*>(Bs);
        static_for<STEPS>([&](auto sc) {
            constexpr int sidx = decltype(sc)::value;
            f32x4 av_n = av;
            bvec bv_n = bv;
            if constexpr (sidx + 1 < STEPS) {
                av_n = *reinterpret_cast<const f32x4*>(As + 2 * KS * (sidx + 1) * TM);
                bv_n = *reinterpret_cast<const bvec*>(Bs + 2 * KS * (sidx + 1) * TK);
                __builtin_amdgcn_sched_barrier(0);
            }
            bsum += av;
            static_for<4>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
#pragma unroll
                for (int d = 0; d < CB; ++d)
                    acc[c][d] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], bv[d], acc[c][d], 0, 0, 0);
                // first half of the stage: up to PPS pieces of the next stage per step, one after each group of CB MFMAs
                constexpr int slot = sidx * PPS + c;
                if constexpr (sidx < STEPS / 2 && c < PPS && slot < NPW) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue(st + NB - 1, bn, slot);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            av = av_n;
            bv = bv_n;
        });
        b = b + 1 == NB ? 0 : b + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the zero stages issued past the slab's end
    // The k-split waves of a workgroup (KS = 2 or 4 for the narrow tiles) hold partial sums of the SAME tile: they are
    // added here, through the stage buffers, in k order - one record per slab instead of KS (the narrow GEMMs of a
    // NeRF step wrote and re-read 2-4x the partial bytes of the wide ones for a fraction of their work).
    if constexpr (KS > 1) {
        constexpr int WF = gemm_merge_wave_floats(CB);
        __syncthreads();                                             // every wave has left the stage buffers
        if (ks > 0) {
            float* mine = smem + ((ks - 1) * TILES + tile) * WF + lane;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int d = 0; d < CB; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mine[((c * CB + d) * 16 + r) * 64] = acc[c][d][r];
#pragma unroll
            for (int q = 0; q < 4; ++q) mine[(4 * CB * 16 + q) * 64] = bsum[q];
        }
        __syncthreads();
        if (ks > 0) return;
#pragma unroll
        for (int k = 1; k < KS; ++k) {
            const float* o = smem + ((k - 1) * TILES + tile) * WF + lane;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int d = 0; d < CB; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[c][d][r] += o[((c * CB + d) * 16 + r) * 64];
#pragma unroll
            for (int q = 0; q < 4; ++q) bsum[q] += o[(4 * CB * 16 + q) * 64];
        }
    }
    // write the partial tile: D[row i'][col j] on lane (j, h), reg r: i' = (r&3) + 8*(r>>2) + 4*h
    float* out = partial + (int64_t)blockIdx.x * (TM * TK + (with_bias ? TM : 0));
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ip = (r & 3) + 8 * (r >> 2) + 4 * h;
            bvec v;
#pragma unroll
            for (int d = 0; d < CB; ++d) v[d] = acc[c][d][r];
            *reinterpret_cast<bvec*>(out + (int64_t)(128 * wm + 4 * ip + c) * TK + 32 * CB * wk + CB * i) = v;
        }
    if (with_bias && wk == 0) {
        bsum.x += __shfl_xor(bsum.x, 32); bsum.y += __shfl_xor(bsum.y, 32);
        bsum.z += __shfl_xor(bsum.z, 32); bsum.w += __shfl_xor(bsum.w, 32);
        if (h == 0) *reinterpret_cast<f32x4*>(out + TM * TK + 128 * wm + 4 * i) = bsum;
    }
}

// Thin gradients: out[c][f] = sum_p S[p][c0+c] * H[p][f] for c < nc <= 3, f < F <= 256, plus sum_p S[p][c0+c], plus - in
// the record's fourth row, free once the rows are being read - the column sums sum_p H[p][f] (the bias gradient of a
// layer whose weights are all K = 3 columns).
//   heads:        S = head pre-activation grads [P,4], H = the head's input  -> dW_head[c][f], db_head[c]
//   K = 3 inputs: S = xin [P,8] (xyz | dir),           H = dA of the layer   -> dW[f][col0 + c]
// grid = (point slabs, jobs), block = 256 threads = features; partial[slab][4][256], bias_partial[slab][4] per job.
constexpr int kMaxThinJobs = 6;
struct ThinBatch {
    const float* S[kMaxThinJobs]; const float* H[kMaxThinJobs];
    float* partial[kMaxThinJobs]; float* bias_partial[kMaxThinJobs];
    int lds_[kMaxThinJobs], c0[kMaxThinJobs], nc[kMaxThinJobs], ldh[kMaxThinJobs], F[kMaxThinJobs];
};
__global__ __launch_bounds__(256) void thin_grad_kernel(ThinBatch tb, int64_t P, int slab_pts) {
    const int job = blockIdx.y;
    const float* __restrict__ S = tb.S[job];
    const float* __restrict__ H = tb.H[job];
    float* __restrict__ partial = tb.partial[job];
    float* __restrict__ bias_partial = tb.bias_partial[job];
    const int lds_ = tb.lds_[job], c0 = tb.c0[job], nc = tb.nc[job], ldh = tb.ldh[job], F = tb.F[job];
    // wave w takes points p0 + w, p0 + w + 4, ...; lane l the features 4l..4l+3 (float4 rows, 4 points in flight);
    // the four waves' sums are added in wave order at the end, so the result does not depend on timing
    __shared__ float red[4][4][256];
    __shared__ float bred[4][4];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t p0 = (int64_t)blockIdx.x * slab_pts;
    const int64_t p1 = p0 + slab_pts < P ? p0 + slab_pts : P;
    const bool live = 4 * lane < F;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f;
    const auto row = [&](int64_t p) {
        return live ? *reinterpret_cast<const f32x4*>(H + p * ldh + 4 * lane) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    const auto fma_row = [&](int64_t p, const f32x4& hv) {
        const float* sp = S + p * lds_ + c0;
        const float s0 = sp[0], s1 = nc > 1 ? sp[1] : 0.f, s2 = nc > 2 ? sp[2] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a0[q] = fmaf(s0, hv[q], a0[q]); a1[q] = fmaf(s1, hv[q], a1[q]); a2[q] = fmaf(s2, hv[q], a2[q]);
        }
        a3 += hv;
        b0 += s0; b1 += s1; b2 += s2;
    };
    // eight rows in flight per wave (and four workgroups per CU: BwdBatcher::flush_thin): this kernel is a stream over
    // P x 1 KiB rows, and with four rows per wave and one workgroup per CU only 16 KiB per CU were in flight - 2.9 TB/s
    int64_t p = p0 + w;
    for (; p + 28 < p1; p += 32) {
        const f32x4 h0 = row(p), h1 = row(p + 4), h2 = row(p + 8), h3 = row(p + 12);
        const f32x4 h4 = row(p + 16), h5 = row(p + 20), h6 = row(p + 24), h7 = row(p + 28);
        fma_row(p, h0); fma_row(p + 4, h1); fma_row(p + 8, h2); fma_row(p + 12, h3);
        fma_row(p + 16, h4); fma_row(p + 20, h5); fma_row(p + 24, h6); fma_row(p + 28, h7);
    }
    for (; p < p1; p += 4) fma_row(p, row(p));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        red[w][0][4 * lane + q] = a0[q]; red[w][1][4 * lane + q] = a1[q]; red[w][2][4 * lane + q] = a2[q];
        red[w][3][4 * lane + q] = a3[q];
    }
    if (lane == 0) { bred[w][0] = b0; bred[w][1] = b1; bred[w][2] = b2; }
    __syncthreads();
    const int f = threadIdx.x;
    float* out = partial + (int64_t)blockIdx.x * 4 * 256;
#pragma unroll
    for (int c = 0; c < 4; ++c) out[c * 256 + f] = ((red[0][c][f] + red[1][c][f]) + red[2][c][f]) + red[3][c][f];
    if (f < 4) {
        float* bo = bias_partial + (int64_t)blockIdx.x * 4;
        bo[f] = f < 3 ? ((bred[0][f] + bred[1][f]) + bred[2][f]) + bred[3][f] : 0.f;
    }
}

// Every cross-slab reduction of a backward pass in ONE launch (grid.y = job), fixed order, deterministic.  A job sums
// n records of `rec` floats element-wise; element idx < TM*TK is tile entry (row, col) = (idx / TK, idx % TK), the
// tail (when bias_dst) the TM bias sums.  Placement: dst[row*ld + col0 + col], or transposed (K = 3 weight columns:
// record [c][f] -> dst[f*ld + col0 + c]).
constexpr int kMaxReduceJobs = 24;
struct ReduceJob {
    const float* src;
    float* dst;
    float* bias_dst;
    int n, rec, TM, TK, ld, col0, rows_valid, cols_valid, transposed;
    int row0 = 0;               // first record row the job places (rows below it belong to another job over the same records)
};
struct ReduceBatch { ReduceJob job[kMaxReduceJobs]; };

__global__ __launch_bounds__(256) void reduce_jobs_kernel(ReduceBatch rb) {
    // Block = 32 float4 entries x 8 k-lanes: lane g sums records g, g+8, g+16, ... (four 16-byte loads in flight), the
    // eight lane sums are added in lane order through LDS - a fixed order whatever the timing.  A job's critical path
    // is n/8 dependent-latency loads instead of n: the 128 x 256 GEMM of a NeRF's dir layer leaves 512 records
    // (256 slabs x 2 k-splits), and with one thread walking all of them the whole launch waited 130 us for it.
    __shared__ f32x4 part[8][32];
    const ReduceJob& j = rb.job[blockIdx.y];
    const int x = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int idx = (blockIdx.x * 32 + x) * 4;
    const bool live = idx < j.rec;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (live) {
        const f32x4* p = reinterpret_cast<const f32x4*>(j.src + idx);
        const int64_t stride4 = j.rec / 4;
        int k = g;
        for (; k + 24 < j.n; k += 32) {
            s0 += p[(int64_t)k * stride4]; s1 += p[(int64_t)(k + 8) * stride4];
            s2 += p[(int64_t)(k + 16) * stride4]; s3 += p[(int64_t)(k + 24) * stride4];
        }
        for (; k < j.n; k += 8) s0 += p[(int64_t)k * stride4];
    }
    part[g][x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g != 0 || !live) return;
    f32x4 s = part[0][x];
#pragma unroll
    for (int t = 1; t < 8; ++t) s += part[t][x];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = idx + q;
        const bool is_bias = e >= j.TM * j.TK;
        const int row = is_bias ? e - j.TM * j.TK : e / j.TK, col = is_bias ? 0 : e % j.TK;
        if (row >= j.rows_valid || (!is_bias && (col >= j.cols_valid || row < j.row0)) || (is_bias && !j.bias_dst)) continue;
        if (is_bias) j.bias_dst[row] = s[q];
        else if (j.transposed) j.dst[(int64_t)col * j.ld + j.col0 + row] = s[q];
        else j.dst[(int64_t)(row - j.row0) * j.ld + j.col0 + col] = s[q];
    }
}

// ---- host orchestration ---------------------------------------------------------------------
int launch_pack_bwd(int kind, const float* const* params, int n_params, float w0, float* packed, hipStream_t stream) {
    const PackTable* t = host_table_bwd(kind);
    ParamPtrsB pp{};
    for (int i = 0; i < n_params && i < 24; ++i) pp.p[i] = params[i];
    hipLaunchKernelGGL(pack_bwd_kernel, dim3(32, t->n_items), dim3(256), 0, stream, kind, pp, packed, w0);
    return check_launch("pack_bwd_kernel");
}

int64_t train_acts_floats(int kind) {
    switch (kind) {
        case 0: return region_total(nerf_acts());
        case 1: return region_total(siren_acts());
        case 2: case 3: return region_total(film_acts());
        case 4: return region_total(tiny_acts());
    }
    return -1;
}
int64_t train_grads_floats(int kind) {
    switch (kind) {
        case 0: return region_total(nerf_grads());
        case 1: return region_total(siren_grads());
        case 2: case 3: return region_total(film_grads());
        case 4: return region_total(tiny_grads());
    }
    return -1;
}
// FiLM scratch: one image's T_l (256x256) and column sums s_l (256) per 256-wide layer, its raw-input columns (256x3, padded)
constexpr int64_t kFilmLayerScratch = 256 * 256 + 256;       // T_l and s_l of one 256-wide FiLM layer
int64_t film_partial_floats(int64_t /*n_groups*/, int64_t /*points_per_group*/) {
    return 8 * kFilmLayerScratch + 2 * 256 * 4 + 256;        // eight layers, two K = 3 blocks, s_0 (launch_field_backward)
}

static int64_t batched_partial_floats(int kind, int64_t P);
int64_t bwd_partial_floats(int64_t P) {
    // FiLM kinds: one image at a time - eight GEMM jobs of at most 32 slabs each, two thin jobs (or the two head
    // jobs over all images): an upper bound for any image of at most P points.  launch_field_backward checks every
    // pass's real plan against this figure before it launches anything.
    const int64_t slabs256 = (P + 255) / 256 > 0 ? (P + 255) / 256 : 1;          // BwdBatcher::slabs_for never exceeds this
    int64_t most = 8 * (slabs256 < 32 ? slabs256 : 32) * kFilmLayerScratch + 2 * (slabs256 < 512 ? slabs256 : 512) * 1280 + 4096;
    for (int kind : {0, 1, 4}) {                                       // the others: every job of the pass at once
        const int64_t n = batched_partial_floats(kind, P);
        if (n > most) most = n;
    }
    return most;
}


// ---- batched orchestration (NeRF, TinyNeRF, SirenNeRF) -----------------------------------------------------------
// A backward pass is: the chain kernel, then every weight-gradient GEMM of one tile shape in ONE launch (grid.y =
// job), the thin (1-/3-row) gradients in one launch, and ONE reduction launch that sums every job's slab partials
// into the gradient tensors - 8 launches for a NeRF (it was 41: at the 1024-ray batch of nerf/train_nerf.py the step
// is launch-bound).  With several jobs per launch a job needs fewer slabs to fill the chip (slabs x jobs ~ 2 per
// CU), so the partial tiles written and re-read shrink by the same factor.
struct BwdBatcher {
    int64_t P;
    float* partial;            // scratch base (null: plan only - used to size the scratch)
    int64_t used = 0;          // floats of scratch handed out
    hipStream_t stream;
    struct Group { GemmBatch b{}; int n = 0; ReduceJob red[kMaxGemmJobs]; } g422, g221, g412, g111;
    ThinBatch thin{};
    int n_thin = 0;
    ReduceJob thin_red[3 * kMaxThinJobs];
    ReduceBatch all{};
    int n_red = 0, max_rec = 0;

    static int slabs_for(int64_t P, int njobs, int per_cu = 1) {
        // one workgroup per CU over the whole launch: the workgroups of a launch do equal work, so a single wave of
        // them has no tail, and fewer slabs mean fewer partial tiles to write and to sum.  (per_cu = 4 for the thin
        // gradients: a 256-thread streaming kernel with no LDS to speak of needs several workgroups per CU to keep
        // enough loads in flight, and its records are 4 KiB, not 256 KiB.)
        int64_t s = (256 * per_cu + njobs - 1) / njobs;
        if (s > 256 * per_cu) s = 256 * per_cu;
        const int64_t most = (P + 255) / 256;              // slabs of at least 256 points
        if (s > most) s = most;
        return (int)(s < 1 ? 1 : s);
    }
    static int slab_pts_for(int64_t P, int slabs) { return (int)(((P + slabs - 1) / slabs + 31) / 32 * 32); }

    float* take(int64_t floats) {
        float* p = partial ? partial + used : nullptr;
        used += (floats + 255) / 256 * 256;
        return p;
    }
    void add_reduce(const ReduceJob& j) {
        all.job[n_red++] = j;
        if (j.rec > max_rec) max_rec = j.rec;
    }
    // dW[M rows][w_ld] at column w_col0 (+ bias gradient gb) = dA^T X over the P points
    template <int CB, int WM, int WK>
    void gemm(Group& g, const float* dA, const float* X, float* gw, int w_ld, int w_col0, int rows_valid, int cols_valid,
              float* gb) {
        constexpr int TM = 128 * WM, TK = 32 * CB * WK;
        g.b.dA[g.n] = dA; g.b.X[g.n] = X; g.b.with_bias[g.n] = gb ? 1 : 0;
        g.red[g.n] = ReduceJob{nullptr, gw, gb, 0, TM * TK + (gb ? TM : 0), TM, TK, w_ld, w_col0, rows_valid, cols_valid, 0};
        ++g.n;
    }
    template <int CB, int WM, int WK>
    int flush(Group& g) {
        if (!g.n) return 0;
        const int slabs = slabs_for(P, g.n), slab = slab_pts_for(P, slabs);
        const int n_slabs = (int)((P + slab - 1) / slab);
        for (int i = 0; i < g.n; ++i) {
            g.red[i].n = n_slabs;                             // one record per slab (k-split waves merge in the kernel)
            g.b.partial[i] = take((int64_t)g.red[i].n * g.red[i].rec);
            g.red[i].src = g.b.partial[i];
            add_reduce(g.red[i]);
        }
        if (!partial) return 0;
        constexpr size_t lds = gemm_lds_bytes<CB, WM, WK>();
        static PerDeviceOnce attr_once;                                               // one per template instance
        const int arc = attr_once.run([&]() {
            if (hipFuncSetAttribute((const void*)dw_gemm_kernel<CB, WM, WK>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess) { set_error("hipFuncSetAttribute(dw_gemm) failed"); return -2; }
            return 0;
        });
        if (arc) return arc;
        hipLaunchKernelGGL((dw_gemm_kernel<CB, WM, WK>), dim3(n_slabs, g.n), dim3(256), lds, stream, g.b, P, slab);
        return check_launch("dw_gemm (batched)");
    }
    // out[c][f] = sum_p S[p][c0+c] H[p][f]: heads (dst [nc][F], bias gb[nc]) or K = 3 weight columns (transposed into
    // dst[f*ld + col0 + c], no bias)
    // colsum_dst: also dst[f] = sum_p H[p][f] (the record's fourth row)
    void thin_job(const float* S, int lds_, int c0, int nc, const float* H, int ldh, int F, float* dst, int ld, int col0,
                  bool transposed, float* gb, float* colsum_dst = nullptr) {
        const int i = n_thin++;
        thin.S[i] = S; thin.lds_[i] = lds_; thin.c0[i] = c0; thin.nc[i] = nc; thin.H[i] = H; thin.ldh[i] = ldh; thin.F[i] = F;
        // record [4][256]: row c, col f
        thin_red[3 * i] = transposed ? ReduceJob{nullptr, dst, nullptr, 0, 1024, 4, 256, ld, col0, nc, F, 1}
                                     : ReduceJob{nullptr, dst, nullptr, 0, 1024, 4, 256, F, 0, nc, F, 0};
        thin_red[3 * i + 1] = gb ? ReduceJob{nullptr, gb, nullptr, 0, 4, 1, 4, 4, 0, 1, nc, 0}
                                 : ReduceJob{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        thin_red[3 * i + 2] = colsum_dst ? ReduceJob{nullptr, colsum_dst, nullptr, 0, 1024, 4, 256, 256, 0, 4, F, 0, 3}
                                         : ReduceJob{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    }
    int flush_thin() {
        if (!n_thin) return 0;
        const int slabs = slabs_for(P, n_thin, 4), slab = slab_pts_for(P, slabs);
        const int n_slabs = (int)((P + slab - 1) / slab);
        for (int i = 0; i < n_thin; ++i) {
            thin.partial[i] = take((int64_t)n_slabs * 1024);
            thin.bias_partial[i] = take((int64_t)n_slabs * 4);
            ReduceJob a = thin_red[3 * i], b = thin_red[3 * i + 1], c = thin_red[3 * i + 2];
            a.n = n_slabs; a.src = thin.partial[i];
            add_reduce(a);
            if (b.dst) { b.n = n_slabs; b.src = thin.bias_partial[i]; add_reduce(b); }
            if (c.dst) { c.n = n_slabs; c.src = thin.partial[i]; add_reduce(c); }
        }
        if (!partial) return 0;
        hipLaunchKernelGGL(thin_grad_kernel, dim3(n_slabs, n_thin), dim3(256), 0, stream, thin, P, slab);
        return check_launch("thin_grad (batched)");
    }
    int reduce_all() {
        if (!partial || !n_red) return 0;
        hipLaunchKernelGGL(reduce_jobs_kernel, dim3((max_rec / 4 + 31) / 32, n_red), dim3(256), 0, stream, all);
        return check_launch("reduce_jobs");
    }
};

// The jobs of one backward pass, per kind.  partial == nullptr: plan only (b.used = scratch floats needed).
static int batched_backward(int kind, BwdBatcher& b, const float* acts, float* grads, float* const* gp) {
    const int64_t P = b.P;
    int rc;
    if (kind == 0 || kind == 4) {
        const bool tiny = kind == 4;
        const RegionLayout AL = tiny ? tiny_acts() : nerf_acts(), GL = tiny ? tiny_grads() : nerf_grads();
        const auto A = [&](int r) { return acts + (int64_t)region_offset(AL, r) * P; };
        const auto G = [&](int r) { return grads + (int64_t)region_offset(GL, r) * P; };
        b.gemm<2, 2, 1>(b.g221, G(0), A(0), gp[0], 60, 0, 256, 60, gp[1]);                        // layers_pos.0: dA0 x E_pos
        if (!tiny) {
            for (int l = 1; l <= 7; ++l)
                b.gemm<4, 2, 2>(b.g422, G(l), A(l), gp[2 * l], l == 5 ? 316 : 256, l == 5 ? 60 : 0, 256, 256, gp[2 * l + 1]);
            b.gemm<2, 2, 1>(b.g221, G(5), A(0), gp[10], 316, 0, 256, 60, nullptr);                // skip layer's E_pos columns
            b.gemm<4, 2, 2>(b.g422, G(8), A(8), gp[16], 256, 0, 256, 256, gp[17]);                // layers_dir.0 x H8
            b.gemm<4, 1, 2>(b.g412, G(9), A(9), gp[18], 280, 0, 128, 256, gp[19]);                // layers_dir.1 x [G |
            b.gemm<1, 1, 1>(b.g111, G(9), A(10), gp[18], 280, 256, 128, 24, nullptr);             //                E_dir]
            b.thin_job(G(10), 4, 3, 1, A(8), 256, 256, gp[20], 256, 0, false, gp[21]);            // sigma head x H8
            b.thin_job(G(10), 4, 0, 3, A(11), 128, 128, gp[22], 128, 0, false, gp[23]);           // rgb head x H_d
        } else {
            for (int l = 1; l <= 3; ++l) b.gemm<4, 2, 2>(b.g422, G(l), A(l), gp[2 * l], 256, 0, 256, 256, gp[2 * l + 1]);
            b.gemm<4, 1, 2>(b.g412, G(4), A(4), gp[8], 280, 0, 128, 256, gp[9]);
            b.gemm<1, 1, 1>(b.g111, G(4), A(5), gp[8], 280, 256, 128, 24, nullptr);
            b.thin_job(G(5), 4, 3, 1, A(4), 256, 256, gp[10], 256, 0, false, gp[11]);
            b.thin_job(G(5), 4, 0, 3, A(6), 128, 128, gp[12], 128, 0, false, gp[13]);
        }
    } else {                                                                                      // SirenNeRF
        const RegionLayout AL = siren_acts(), GL = siren_grads();
        const auto A = [&](int r) { return acts + (int64_t)region_offset(AL, r) * P; };
        const auto G = [&](int r) { return grads + (int64_t)region_offset(GL, r) * P; };
        for (int l = 1; l <= 7; ++l)                                                              // input of layer l: X_l
            b.gemm<4, 2, 2>(b.g422, G(l), A(l), gp[2 * l], l == 5 ? 259 : 256, l == 5 ? 3 : 0, 256, 256, gp[2 * l + 1]);
        b.gemm<4, 2, 2>(b.g422, G(8), A(8), gp[16], 256, 0, 256, 256, gp[17]);                    // layers_dir.0 x X8
        b.gemm<4, 1, 2>(b.g412, G(9), A(9), gp[18], 259, 0, 128, 256, gp[19]);                    // layers_dir.1 x G
        b.thin_job(A(0), 8, 0, 3, G(0), 256, 256, gp[0], 3, 0, true, nullptr, gp[1]);             // layers_pos.0 (K = 3) + its bias
        b.thin_job(A(0), 8, 0, 3, G(5), 256, 256, gp[10], 259, 0, true, nullptr);                 // skip layer's xyz columns
        b.thin_job(A(0), 8, 3, 3, G(9), 128, 128, gp[18], 259, 256, true, nullptr);               // layers_dir.1's dir columns
        b.thin_job(G(10), 4, 3, 1, A(8), 256, 256, gp[20], 256, 0, false, gp[21]);                // sigma head x X8
        b.thin_job(G(10), 4, 0, 3, A(10), 128, 128, gp[22], 128, 0, false, gp[23]);               // rgb head x X_d
    }
    if ((rc = b.flush<4, 2, 2>(b.g422))) return rc;
    if ((rc = b.flush<2, 2, 1>(b.g221))) return rc;
    if ((rc = b.flush<4, 1, 2>(b.g412))) return rc;
    if ((rc = b.flush<1, 1, 1>(b.g111))) return rc;
    if ((rc = b.flush_thin())) return rc;
    return b.reduce_all();
}

// Backward of a NeRF / TinyNeRF field over P points.  grad_params[2i], [2i+1]: device pointers to the weight /
// bias gradient tensors (torch layout), overwritten.
int launch_field_backward(int kind, const float* packed_bwd, const float* acts, float* grads, const float* raw,
                          const float* g_raw, int64_t n_groups, int64_t points_per_group, const float* film,
                          float* film_partial, float* grad_film, float* partial, float* const* gp,
                          const float* const* params, hipStream_t stream) {
    const int64_t P = n_groups * points_per_group;
    if (P <= 0) return 0;
    if (kind < 0 || kind > 4) { set_error("unknown field kind %d", kind); return -1; }
    const size_t lds = kLdsFloats * sizeof(float);
    static PerDeviceOnce attr_once;
    const int arc = attr_once.run([&]() {
        const void* fns[] = {(const void*)nerf_bwd_kernel<false>, (const void*)nerf_bwd_kernel<true>,
                             (const void*)siren_bwd_kernel, (const void*)film_bwd_kernel<true>,
                             (const void*)film_bwd_kernel<false>};
        for (const void* f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                set_error("hipFuncSetAttribute failed"); return -2;
            }
        return 0;
    });
    if (arc) return arc;
    const int64_t tpg = (points_per_group + 127) / 128;
#ifdef MI_PROFILE_STAMPS
    unsigned long long* stamps = g_bwd_stamps;
#else
    unsigned long long* stamps = nullptr;
#endif
    BwdArgs a{packed_bwd, acts, grads, raw, g_raw, P, film, film_partial, points_per_group, tpg, n_groups * tpg, stamps};
    const unsigned blocks = (unsigned)((kind == 2 || kind == 3) ? n_groups * tpg : (P + 127) / 128);
    int rc;
    if (kind == 0 || kind == 1 || kind == 4) {
        if (kind == 0) hipLaunchKernelGGL(nerf_bwd_kernel<false>, dim3(blocks), dim3(256), lds, stream, a);
        else if (kind == 4) hipLaunchKernelGGL(nerf_bwd_kernel<true>, dim3(blocks), dim3(256), lds, stream, a);
        else hipLaunchKernelGGL(siren_bwd_kernel, dim3(blocks), dim3(256), lds, stream, a);
        if ((rc = check_launch("backward chain"))) return rc;
        // dry run first (no launches): the scratch this very pass needs must fit what mi_field_bwd_partial_floats told the
        // caller to allocate - a planning / execution mismatch would otherwise be an out-of-bounds write by the GEMMs
        BwdBatcher plan{P, nullptr, 0, stream};
        (void)batched_backward(kind, plan, acts, grads, gp);
        if (plan.used > bwd_partial_floats(P)) {
            set_error("backward scratch plan (%lld floats) exceeds mi_field_bwd_partial_floats (%lld)", (long long)plan.used,
                      (long long)bwd_partial_floats(P));
            return -1;
        }
        BwdBatcher bb{P, partial, 0, stream};
        if ((rc = batched_backward(kind, bb, acts, grads, gp))) return rc;
    } else if (kind == 2 || kind == 3) {
        const bool use_dir = kind == 2;
        if (!film || !film_partial || !grad_film || !params) {
            set_error("FiLM backward needs film, the FiLM scratch, grad_film and the parameter pointers");
            return -1;
        }
        if (use_dir) hipLaunchKernelGGL(film_bwd_kernel<true>, dim3(blocks), dim3(256), lds, stream, a);
        else hipLaunchKernelGGL(film_bwd_kernel<false>, dim3(blocks), dim3(256), lds, stream, a);
        if ((rc = check_launch("film_bwd_kernel"))) return rc;
        constexpr RegionLayout AL = film_acts();
        const int64_t ppg = points_per_group;
        // FiLM scratch of the current image: T_l [256][256] + s_l [256] for the eight 256-wide FiLM layers, the K = 3
        // blocks of layer 0 (xyz) and of layer 8 (dir) as [256][3] (padded to 4), s_0 [256]
        const auto Tl = [&](int l) { return film_partial + (int64_t)(l - 1) * kFilmLayerScratch; };
        const auto sl = [&](int l) { return Tl(l) + 256 * 256; };
        float* T3_0 = film_partial + 8 * kFilmLayerScratch;
        float* T3_8 = T3_0 + 256 * 4;
        float* s0 = T3_8 + 256 * 4;
        const int ld9 = use_dir ? 259 : 256;
        // Per image g (fixed order, so the sums over images are deterministic): T_g, s_g of every FiLM layer - the eight
        // 256 x 256 GEMMs of an image are ONE launch (grid.y = layer, a job needs only 32 slabs to give every CU a
        // workgroup, so 8x fewer partial tiles are written and re-read than with a launch per layer), the K = 3 blocks
        // one thin launch, one reduction launch for all of them - then per layer
        // dW += gamma_g (.) T_g, db += gamma_g (.) s_g, d gamma_g = <W, T_g> + b (.) s_g, d beta_g = s_g.
        for (int64_t g = 0; g < n_groups; ++g) {
            const auto A = [&](int r) { return acts + (int64_t)region_offset(AL, r) * P + g * ppg * AL.width[r]; };
            const auto G = [&](int l) { return grads + (int64_t)(256 * l) * P + g * ppg * 256; };
            const float* frow = film + (g * kFilmLayers) * kFilmRow;
            float* dfrow = grad_film + (g * kFilmLayers) * kFilmRow;
            const int first = g == 0;
            const auto jobs = [&](BwdBatcher& bb) -> int {
                for (int l = 1; l <= 8; ++l)                                  // FiLM layer l: input X_{l-1} = acts region l
                    bb.gemm<4, 2, 2>(bb.g422, G(l), A(l), Tl(l), 256, 0, 256, 256, sl(l));
                bb.thin_job(A(0), 8, 0, 3, G(0), 256, 256, T3_0, 3, 0, true, nullptr, s0);          // input_layer (K = 3: xyz) + s_0
                if (use_dir) bb.thin_job(A(0), 8, 3, 3, G(8), 256, 256, T3_8, 3, 0, true, nullptr); // hidden_layer_rgb's dir columns
                int r;
                if ((r = bb.flush<4, 2, 2>(bb.g422))) return r;
                if ((r = bb.flush_thin())) return r;
                return bb.reduce_all();
            };
            if (first) {                             // dry run: this image's scratch plan against what the caller was told to allocate
                BwdBatcher plan{ppg, nullptr, 0, stream};
                (void)jobs(plan);
                if (plan.used > bwd_partial_floats(P)) {
                    set_error("FiLM backward scratch plan (%lld floats) exceeds mi_field_bwd_partial_floats (%lld)",
                              (long long)plan.used, (long long)bwd_partial_floats(P));
                    return -1;
                }
            }
            BwdBatcher bb{ppg, partial, 0, stream};
            if ((rc = jobs(bb))) return rc;
            FinishBatch fb{};
            int n_fin = 0, n_heads = 0;
            fb.first_group = first;
            const auto finish = [&](const float* t, int tk, const float* sg, int l, int wp, int w_ld, int col0, int bias_part) {
                fb.job[n_fin++] = FinishJob{t, sg, params[2 * wp], params[2 * wp + 1], frow + l * kFilmRow, gp[2 * wp], gp[2 * wp + 1],
                                            dfrow + l * kFilmRow, tk, w_ld, col0, bias_part};
                n_heads += bias_part;
            };
            finish(T3_0, 3, s0, 0, 0, 3, 0, 1);                               // input_layer: FiLM layer 0, parameter pair 0
            for (int l = 1; l <= 7; ++l) finish(Tl(l), 256, sl(l), l, l, 256, 0, 1);      // hidden_layers[l-1]: pair l
            finish(Tl(8), 256, sl(8), 8, 9, ld9, 0, 1);                       // hidden_layer_rgb: FiLM layer 8, pair 9: [X_7 | dir]
            if (use_dir) finish(T3_8, 3, sl(8), 8, 9, 259, 256, 0);           // ... its dir columns: chained behind the job above
            hipLaunchKernelGGL(film_finish_kernel, dim3(256, n_heads), dim3(256), 0, stream, fb);
        }
        if ((rc = check_launch("film_finish_kernel"))) return rc;
        // heads: sigma (param pair 8) on X_7, rgb (pair 10) on X_8 - no FiLM in between, all images at once
        const auto A = [&](int r) { return acts + (int64_t)region_offset(AL, r) * P; };
        const float* dpre = grads + (int64_t)(9 * 256) * P;
        for (int pass = 0; pass < 2; ++pass) {                        // 0: plan only (scratch check), 1: launch
            BwdBatcher hb{P, pass ? partial : nullptr, 0, stream};
            hb.thin_job(dpre, 4, 3, 1, A(8), 256, 256, gp[16], 256, 0, false, gp[17]);                        // sigma x X_7
            hb.thin_job(dpre, 4, 0, 3, A(9), 256, 256, gp[20], 256, 0, false, gp[21]);                        // rgb x X_8
            if ((rc = hb.flush_thin())) return rc;
            if (!pass && hb.used > bwd_partial_floats(P)) { set_error("FiLM head scratch plan exceeds mi_field_bwd_partial_floats"); return -1; }
            if ((rc = hb.reduce_all())) return rc;
        }
    }
    return 0;
}

static int64_t batched_partial_floats(int kind, int64_t P) {
    // Plan only (partial == nullptr: nothing is launched, nothing dereferenced).  The gradient pointers must be NON-null
    // here: a job's record is TM*TK floats plus TM bias sums when it has a bias destination, and a plan made with null
    // destinations comes out 256 floats per record short of what the real pass writes.
    static float sink;
    float* gp[24];
    for (float*& g : gp) g = &sink;
    BwdBatcher bb{P, nullptr, 0, nullptr};
    (void)batched_backward(kind, bb, nullptr, nullptr, gp);
    return bb.used + 1024;
}

}  // namespace mi
