// field_layout.h - packed weight-stream layout shared by the pack kernel and the fused MLP
// kernel (gfx950).  One source of truth for both sides.
//
// The fused kernel computes every 256-wide layer as  Y[f_out][p] = sum_k W[f_out][k] X[k][p]
// on v_mfma_f32_32x32x2_f32 with OUTPUT FEATURES ON MFMA ROWS and POINTS ON MFMA COLUMNS, so a
// layer's accumulator registers are directly the next layer's B operands (no LDS round trip
// for activations).  Register r (0..15) of accumulator block m on lane (col j, half h) holds
// feature  32*m + 8*(r>>2) + 4*h + (r&3).  As a B operand, register r of block kb therefore
// carries k = 32*kb + 8*(r>>2) + 4*h + (r&3), and the A operand of that MFMA must hold
// W[32*m + i][k] on lane (i, h).
//
// Stream = concatenation, in consumption order, of
//   VEC piece   256 floats: a per-feature vector v[f] stored at ((f/32)*2 + h)*16 + r,
//               with (h, r) from f%32 as above (bias, head weights, K=3 input columns)
//   PLAIN piece 256 floats: up to 4 scalars at [0..3] (head biases)
//   CHUNK       MB*1024 floats: one 32-wide K block of a layer for all MB output blocks:
//               index (((rg*MB + m)*64 + lane)*4 + q) = W[32*m + (lane&31)][col0 + 8*rg + 4*(lane>>5) + q]
// Each 256-float piece is one global_load_lds_dwordx4 wave-instruction (1 KiB).
#pragma once
#include <stdint.h>

namespace mi {

constexpr int kPiece = 256;          // floats per DMA piece
constexpr int kMaxAuxPieces = 8;     // per-layer VEC/PLAIN pieces (LDS aux slot = 8 KiB)
constexpr int kHidden = 256;
constexpr int kFilmLayers = 9;
constexpr int kFilmRow = 512;        // gamma[256] | beta[256]

enum ItemType : int { ITEM_VEC = 0, ITEM_PLAIN = 1, ITEM_CHUNK = 2 };

struct PackItem {
    int type;          // ItemType
    int param;         // index into params[] (weight or bias tensor)
    int ld;            // CHUNK: source stride between MFMA rows (forward: in_features; transposed: 1)
    int offset;        // VEC: start offset into source; CHUNK: offset of element (row 0, col 0)
    int stride;        // VEC: stride between consecutive features; CHUNK: stride between K columns
    int n_valid;       // VEC/PLAIN: valid features; CHUNK: valid K columns (<=32)
    int rows_valid;    // CHUNK: valid MFMA rows
    int mb;            // CHUNK: MFMA row blocks (8 or 4)
    int row0, col0;    // coordinates, in the [out, in] weight matrix (bias: [n, 1]), of the item's first element: the
                       // inverse map (parameter element -> stream position) of adam_step.hip needs no division by them
};

constexpr int kMaxItems = 112;

struct PackTable {
    int n_items;
    int dst_off[kMaxItems];   // float offset of each item in the packed stream
    PackItem item[kMaxItems];
};

// feature -> slot inside a VEC piece
__host__ __device__ inline int vec_slot(int f) {
    int m = f >> 5, w = f & 31;
    int rg = w >> 3, h = (w >> 2) & 1, q = w & 3;
    return (m * 2 + h) * 16 + rg * 4 + q;
}

}  // namespace mi

// ---------------------------------------------------------------------------------------
// Per-kind recipes.  params[2*i] / params[2*i+1] = weight / bias of linear layer i in the
// state-dict order listed per kind below (same order as oracle/fields.py SPECS).
// The fused kernels in field_mlp.hip consume the stream in exactly this order.
// ---------------------------------------------------------------------------------------
namespace mi {

struct TableBuilder {
    PackTable t{};
    int off = 0;

    constexpr void push(PackItem it, int floats) {
        t.item[t.n_items] = it;
        t.dst_off[t.n_items] = off;
        t.n_items++;
        off += floats;
    }
    // bias of layer i (n outputs)
    constexpr void bias(int layer, int n) { push({ITEM_VEC, 2 * layer + 1, 0, 0, 1, n, 0, 0, 0, 0}, kPiece); }
    // row `row` of weight of layer i as a vector over its n input features starting at col0
    constexpr void wrow(int layer, int ld, int row, int col0, int n) {
        push({ITEM_VEC, 2 * layer, ld, row * ld + col0, 1, n, 0, 0, row, col0}, kPiece);
    }
    // column `col` of weight of layer i as a vector over its n output features
    constexpr void wcol(int layer, int ld, int col, int n) { push({ITEM_VEC, 2 * layer, ld, col, ld, n, 0, 0, 0, col}, kPiece); }
    // first n entries of bias of layer i, unpacked at [0..n)
    constexpr void scalars(int layer, int n) { push({ITEM_PLAIN, 2 * layer + 1, 0, 0, 1, n, 0, 0, 0, 0}, kPiece); }
    // K blocks of weight of layer i covering columns [col0, col0+ncols)
    constexpr void chunks(int layer, int ld, int col0, int ncols, int rows, int mb) {
        for (int c = 0; c < ncols; c += 32) {
            int nv = ncols - c < 32 ? ncols - c : 32;
            push({ITEM_CHUNK, 2 * layer, ld, col0 + c, 1, nv, rows, mb, 0, col0 + c}, mb * 1024);
        }
    }
    // K blocks of the TRANSPOSED weight of layer i for the backward chain dX = W^T dA:
    // MFMA rows = input features [hcol0, hcol0+rows) of W, K = W's n_out output features
    constexpr void chunks_t(int layer, int ld, int hcol0, int rows, int n_out, int mb) {
        for (int c = 0; c < n_out; c += 32) {
            int nv = n_out - c < 32 ? n_out - c : 32;
            push({ITEM_CHUNK, 2 * layer, 1, hcol0 + c * ld, ld, nv, rows, mb, c, hcol0}, mb * 1024);
        }
    }
};

// NeRF (nerf/nerf.py:59-73): 0..7 layers_pos, 8,9 layers_dir, 10 output_layer_sigma, 11 output_layer_rgb
constexpr PackTable build_nerf() {
    TableBuilder b;
    b.bias(0, 256); b.chunks(0, 60, 0, 60, 256, 8);
    for (int l = 1; l <= 4; ++l) { b.bias(l, 256); b.chunks(l, 256, 0, 256, 256, 8); }
    b.bias(5, 256); b.chunks(5, 316, 0, 60, 256, 8); b.chunks(5, 316, 60, 256, 256, 8);
    b.bias(6, 256); b.chunks(6, 256, 0, 256, 256, 8);
    b.bias(7, 256); b.wrow(10, 256, 0, 0, 256); b.scalars(10, 1); b.chunks(7, 256, 0, 256, 256, 8);
    b.bias(8, 256); b.chunks(8, 256, 0, 256, 256, 8);
    b.bias(9, 128); b.wrow(11, 128, 0, 0, 128); b.wrow(11, 128, 1, 0, 128); b.wrow(11, 128, 2, 0, 128); b.scalars(11, 3);
    b.chunks(9, 280, 0, 256, 128, 4); b.chunks(9, 280, 256, 24, 128, 4);
    return b.t;
}

// TinyNeRF (build-defined, BASELINE C1): 0..3 layers_pos, 4 layers_dir.0 [128,280], 5 sigma, 6 rgb
constexpr PackTable build_tiny_nerf() {
    TableBuilder b;
    b.bias(0, 256); b.chunks(0, 60, 0, 60, 256, 8);
    for (int l = 1; l <= 2; ++l) { b.bias(l, 256); b.chunks(l, 256, 0, 256, 256, 8); }
    b.bias(3, 256); b.wrow(5, 256, 0, 0, 256); b.scalars(5, 1); b.chunks(3, 256, 0, 256, 256, 8);
    b.bias(4, 128); b.wrow(6, 128, 0, 0, 128); b.wrow(6, 128, 1, 0, 128); b.wrow(6, 128, 2, 0, 128); b.scalars(6, 3);
    b.chunks(4, 280, 0, 256, 128, 4); b.chunks(4, 280, 256, 24, 128, 4);
    return b.t;
}

// SirenNeRF (nerf/nerf.py:123-150): same indices as NeRF; layer 0 is K=3 (VALU), layer 5 = [pos(3) | h(256)],
// layer 9 = [h(256) | dir(3)]
constexpr PackTable build_siren_nerf() {
    TableBuilder b;
    b.bias(0, 256); b.wcol(0, 3, 0, 256); b.wcol(0, 3, 1, 256); b.wcol(0, 3, 2, 256);
    for (int l = 1; l <= 4; ++l) { b.bias(l, 256); b.chunks(l, 256, 0, 256, 256, 8); }
    b.bias(5, 256); b.wcol(5, 259, 0, 256); b.wcol(5, 259, 1, 256); b.wcol(5, 259, 2, 256);
    b.chunks(5, 259, 3, 256, 256, 8);
    b.bias(6, 256); b.chunks(6, 256, 0, 256, 256, 8);
    b.bias(7, 256); b.wrow(10, 256, 0, 0, 256); b.scalars(10, 1); b.chunks(7, 256, 0, 256, 256, 8);
    b.bias(8, 256); b.chunks(8, 256, 0, 256, 256, 8);
    b.bias(9, 128); b.wcol(9, 259, 256, 128); b.wcol(9, 259, 257, 128); b.wcol(9, 259, 258, 128);
    b.wrow(11, 128, 0, 0, 128); b.wrow(11, 128, 1, 0, 128); b.wrow(11, 128, 2, 0, 128); b.scalars(11, 3);
    b.chunks(9, 259, 0, 256, 128, 4);
    return b.t;
}

// FilmSirenNeRF (pi_GAN/modules.py:76-94): 0 input_layer, 1..7 hidden_layers, 8 output_layer_sigma.0,
// 9 hidden_layer_rgb ([256,259] with dir, [256,256] without), 10 output_layer_rgb.0
constexpr PackTable build_film(bool use_dir) {
    TableBuilder b;
    b.bias(0, 256); b.wcol(0, 3, 0, 256); b.wcol(0, 3, 1, 256); b.wcol(0, 3, 2, 256);
    for (int l = 1; l <= 6; ++l) { b.bias(l, 256); b.chunks(l, 256, 0, 256, 256, 8); }
    b.bias(7, 256); b.wrow(8, 256, 0, 0, 256); b.scalars(8, 1); b.chunks(7, 256, 0, 256, 256, 8);
    const int ld = use_dir ? 259 : 256;
    b.bias(9, 256);
    if (use_dir) { b.wcol(9, ld, 256, 256); b.wcol(9, ld, 257, 256); b.wcol(9, ld, 258, 256); }
    b.wrow(10, 256, 0, 0, 256); b.wrow(10, 256, 1, 0, 256); b.wrow(10, 256, 2, 0, 256); b.scalars(10, 3);
    b.chunks(9, ld, 0, 256, 256, 8);
    return b.t;
}

// ---- backward-chain streams (dX = W^T dA per layer, reverse order), consumed by field_mlp_bwd.hip ----
// NeRF: [rgb head rows x3] | layers_dir.1^T (h part) | [sigma head row] layers_dir.0^T | layers_pos.7^T .. 1^T
constexpr PackTable build_nerf_bwd() {
    TableBuilder b;
    b.wrow(11, 128, 0, 0, 128); b.wrow(11, 128, 1, 0, 128); b.wrow(11, 128, 2, 0, 128);
    b.chunks_t(9, 280, 0, 256, 128, 8);
    b.wrow(10, 256, 0, 0, 256);
    b.chunks_t(8, 256, 0, 256, 256, 8);
    b.chunks_t(7, 256, 0, 256, 256, 8);
    b.chunks_t(6, 256, 0, 256, 256, 8);
    b.chunks_t(5, 316, 60, 256, 256, 8);
    for (int l = 4; l >= 1; --l) b.chunks_t(l, 256, 0, 256, 256, 8);
    return b.t;
}

// TinyNeRF: [rgb rows x3] | layers_dir.0^T (h part) with [sigma row] | layers_pos.3^T .. 1^T
constexpr PackTable build_tiny_nerf_bwd() {
    TableBuilder b;
    b.wrow(6, 128, 0, 0, 128); b.wrow(6, 128, 1, 0, 128); b.wrow(6, 128, 2, 0, 128); b.wrow(5, 256, 0, 0, 256);
    b.chunks_t(4, 280, 0, 256, 128, 8);
    for (int l = 3; l >= 1; --l) b.chunks_t(l, 256, 0, 256, 256, 8);
    return b.t;
}

// SirenNeRF: same chain as NeRF (K=3 inputs need no dX); layers_pos.5's h part starts at column 3
constexpr PackTable build_siren_nerf_bwd() {
    TableBuilder b;
    b.wrow(11, 128, 0, 0, 128); b.wrow(11, 128, 1, 0, 128); b.wrow(11, 128, 2, 0, 128);
    b.chunks_t(9, 259, 0, 256, 128, 8);
    b.wrow(10, 256, 0, 0, 256);
    b.chunks_t(8, 256, 0, 256, 256, 8);
    b.chunks_t(7, 256, 0, 256, 256, 8);
    b.chunks_t(6, 256, 0, 256, 256, 8);
    b.chunks_t(5, 259, 3, 256, 256, 8);
    for (int l = 4; l >= 1; --l) b.chunks_t(l, 256, 0, 256, 256, 8);
    return b.t;
}

// FilmSirenNeRF: [rgb rows x3 over 256 features] | hidden_layer_rgb^T (h part) with [sigma row] | hidden 6..0 ^T
constexpr PackTable build_film_bwd(bool use_dir) {
    TableBuilder b;
    b.wrow(10, 256, 0, 0, 256); b.wrow(10, 256, 1, 0, 256); b.wrow(10, 256, 2, 0, 256); b.wrow(8, 256, 0, 0, 256);
    b.chunks_t(9, use_dir ? 259 : 256, 0, 256, 256, 8);
    for (int l = 7; l >= 1; --l) b.chunks_t(l, 256, 0, 256, 256, 8);
    return b.t;
}

// ---- training buffers: per-point row-major regions [points][width], region r at offset_r * points ------
// acts  = layer inputs saved by the forward (what dW = dA X^T and the activation derivatives need)
// grads = dA of every linear layer written by the backward chain (+ 4 head pre-activation grads)
struct RegionLayout { int n; int width[40]; };
constexpr int region_offset(const RegionLayout& L, int idx) {
    int o = 0;
    for (int i = 0; i < idx; ++i) o += L.width[i];
    return o;
}
constexpr int region_total(const RegionLayout& L) { return region_offset(L, L.n); }

// NeRF acts: 0 E_pos(64) | 1..8 H1..H8 (post-ReLU outputs of layers_pos.0..7) | 9 G (layers_dir.0 out) |
//            10 E_dir(32) | 11 H_d (post-ReLU layers_dir.1, 128)
//            12..19 ReLU switches of H1..H8 (8 dwords = 256 bits per point) | 20 of H_d (4 dwords): what the backward CHAIN
//            needs of a ReLU layer is one bit per unit (the dW GEMMs read the rows themselves); lane (point, half h) owns
//            dwords [4h, 4h + 4) (H_d: [2h, 2h + 2)), the bit of block m, quarter rg, element q = 31 - (16 (m & 1) + 4 rg + q)
//            of dword m >> 1 (relu_switch_in / relu_switch_of in field_mlp_device.h)
constexpr RegionLayout nerf_acts() { return {21, {64, 256, 256, 256, 256, 256, 256, 256, 256, 256, 32, 128, 8, 8, 8, 8, 8, 8, 8, 8, 4}}; }
// NeRF grads: 0..7 dA of layers_pos.0..7 | 8 dA layers_dir.0 | 9 dA layers_dir.1 (128) | 10 head pre-act grads (4)
constexpr RegionLayout nerf_grads() { return {11, {256, 256, 256, 256, 256, 256, 256, 256, 256, 128, 4}}; }
// TinyNeRF acts: 0 E_pos | 1..4 H1..H4 | 5 E_dir(32) | 6 H_d(128); grads: 0..3 | 4 dA dir (128) | 5 heads (4)
//                7..10 ReLU switches of H1..H4 (8 dwords per point) | 11 of H_d (4 dwords), as in nerf_acts()
constexpr RegionLayout tiny_acts() { return {12, {64, 256, 256, 256, 256, 32, 128, 8, 8, 8, 8, 4}}; }
constexpr RegionLayout tiny_grads() { return {6, {256, 256, 256, 256, 128, 4}}; }

// SirenNeRF acts: 0 xin(8: xyz, dir, 0, 0) | l = 1..8: X_l = sin(30 A_{l-1}) with the sign of cos(30 A_{l-1}) in its
//                 lowest mantissa bit (mi_math.h: the derivative factor 30 cos is rebuilt from it, not stored) |
//                 9 G (layers_dir.0 out) | 10 X_d (128, same encoding).   grads: as NeRF.
constexpr RegionLayout siren_acts() { return {11, {8, 256, 256, 256, 256, 256, 256, 256, 256, 256, 128}}; }
constexpr RegionLayout siren_grads() { return nerf_grads(); }
// FilmSirenNeRF acts: 0 xin(8) | FiLM layer l = 0..8 (input, hidden 0..6, rgb hidden): 1+l X_l = sin(30 u),
//                     u = gamma*A + beta, encoded like SirenNeRF's (the linear output A is not kept either: the FiLM
//                     table gradient comes out of the per-image dW sums, see launch_field_backward).
// grads: 0..8 dL/du_l (256) | 9 head pre-act grads (4)
constexpr RegionLayout film_acts() {
    RegionLayout L{10, {}};
    L.width[0] = 8;
    for (int i = 1; i < 10; ++i) L.width[i] = 256;
    return L;
}
constexpr RegionLayout film_grads() { return {10, {256, 256, 256, 256, 256, 256, 256, 256, 256, 4}}; }

// The stream ends with a trailer piece of hyper-parameters the kernels read as wave-uniform scalars: [0] = w_0, the sin layers'
// frequency (FilmSiren's constructor argument, pi_GAN/modules.py:11,73; 30 for every other kind), [1] = fl(w_0^2) for the
// backward's rebuilt derivative.  Written by the pack kernels, left alone by the fused Adam's scatter refresh.
constexpr int kTrailer = 256;
constexpr int packed_body_floats(const PackTable& t) {
    const PackItem& last = t.item[t.n_items - 1];
    return t.dst_off[t.n_items - 1] + (last.type == ITEM_CHUNK ? last.mb * 1024 : kPiece);
}
constexpr int packed_floats(const PackTable& t) { return packed_body_floats(t) + kTrailer; }

}  // namespace mi
