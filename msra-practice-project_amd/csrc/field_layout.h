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
    int ld;            // row stride of the source (weights: in_features; bias: 0)
    int offset;        // VEC: start offset into source; CHUNK: first column (col0)
    int stride;        // VEC: element stride between consecutive features
    int n_valid;       // VEC/PLAIN: valid features; CHUNK: valid columns from col0 (<=32)
    int rows_valid;    // CHUNK: valid output rows
    int mb;            // CHUNK: output blocks (8 or 4)
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
    constexpr void bias(int layer, int n) { push({ITEM_VEC, 2 * layer + 1, 0, 0, 1, n, 0, 0}, kPiece); }
    // row `row` of weight of layer i as a vector over its n input features starting at col0
    constexpr void wrow(int layer, int ld, int row, int col0, int n) {
        push({ITEM_VEC, 2 * layer, ld, row * ld + col0, 1, n, 0, 0}, kPiece);
    }
    // column `col` of weight of layer i as a vector over its n output features
    constexpr void wcol(int layer, int ld, int col, int n) { push({ITEM_VEC, 2 * layer, ld, col, ld, n, 0, 0}, kPiece); }
    // first n entries of bias of layer i, unpacked at [0..n)
    constexpr void scalars(int layer, int n) { push({ITEM_PLAIN, 2 * layer + 1, 0, 0, 1, n, 0, 0}, kPiece); }
    // K blocks of weight of layer i covering columns [col0, col0+ncols)
    constexpr void chunks(int layer, int ld, int col0, int ncols, int rows, int mb) {
        for (int c = 0; c < ncols; c += 32) {
            int nv = ncols - c < 32 ? ncols - c : 32;
            push({ITEM_CHUNK, 2 * layer, ld, col0 + c, 0, nv, rows, mb}, mb * 1024);
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

constexpr int packed_floats(const PackTable& t) {
    const PackItem& last = t.item[t.n_items - 1];
    return t.dst_off[t.n_items - 1] + (last.type == ITEM_CHUNK ? last.mb * 1024 : kPiece);
}

}  // namespace mi
