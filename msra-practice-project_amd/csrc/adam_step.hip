// adam_step.hip - the optimiser step of nerf/train_nerf.py:98,168 (torch.optim.Adam, betas (0.9, 0.999), no weight
// decay, no amsgrad) for the parameters of up to two fields, FUSED with the refresh of their packed MFMA weight
// streams (gfx950).
//
// After a torch optimiser step the renderer has to repack both streams of each model (forward order, and the
// transposed order of the backward chain: field_layout.h) from the updated parameters: two optimiser-side launches
// and four pack launches per step, on a step that is launch-bound at the reference's 1024-ray batch.  Here one
// launch does all of it: thread = one parameter element; it applies Adam to (p, m, v) and SCATTERS the new value to
// every position of the two streams that holds it, by inverting the item tables (an element sits in at most a few
// items: its K block of the forward stream, its K block of the transposed stream, a VEC / PLAIN piece for biases,
// head rows and K = 3 columns).  Padding entries of the streams never change, so they are written once by the
// ordinary pack kernels and left alone.
//
// Arithmetic follows torch's multi-tensor Adam op by op (torch/optim/adam.py:_multi_tensor_adam): lerp of exp_avg,
// mul + addcmul of exp_avg_sq, sqrt / bias_correction2_sqrt + eps, addcdiv with -lr / bias_correction1; the scalars
// are computed on the host in double like torch does and handed over as floats.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "field_layout.h"
#include "mi_common.h"

namespace mi {

__constant__ PackTable c_fwd[5] = {build_nerf(), build_siren_nerf(), build_film(true), build_film(false), build_tiny_nerf()};
__constant__ PackTable c_bwd[5] = {build_nerf_bwd(), build_siren_nerf_bwd(), build_film_bwd(true), build_film_bwd(false),
                                   build_tiny_nerf_bwd()};

constexpr int kAdamMaxParams = 48;      // two fields of up to 24 tensors
constexpr int kAdamMaxHits = 16;        // stream items per parameter tensor the kernel's hit lists hold

// The hit lists are sized by a constant, the tables by field_layout.h: tie the two at compile time, so a layout change
// that puts a tensor into more items than the lists hold cannot leave stream positions silently stale.
constexpr int max_items_per_param(const PackTable& t) {
    int worst = 0;
    for (int prm = 0; prm < kAdamMaxParams; ++prm) {       // a field has at most 24 tensors
        int n = 0;
        for (int i = 0; i < t.n_items; ++i) n += t.item[i].param == prm;
        worst = n > worst ? n : worst;
    }
    return worst;
}
constexpr int max_items_per_param_all() {
    const PackTable all[10] = {build_nerf(), build_siren_nerf(), build_film(true), build_film(false), build_tiny_nerf(),
                               build_nerf_bwd(), build_siren_nerf_bwd(), build_film_bwd(true), build_film_bwd(false),
                               build_tiny_nerf_bwd()};
    int worst = 0;
    for (int k = 0; k < 10; ++k) { const int n = max_items_per_param(all[k]); worst = n > worst ? n : worst; }
    return worst;
}
static_assert(max_items_per_param_all() <= kAdamMaxHits, "adam_pack_kernel: a parameter sits in more stream items than hit_f / hit_b hold");
struct AdamArgs {
    float* p[kAdamMaxParams];
    const float* g[kAdamMaxParams];
    float* m[kAdamMaxParams];
    float* v[kAdamMaxParams];
    int numel[kAdamMaxParams];
    int in_f[kAdamMaxParams];           // row length of the tensor ([out, in] weight: in; bias: 1)
    int field[kAdamMaxParams];          // which field the tensor belongs to (0 / 1)
    int index[kAdamMaxParams];          // its index in that field's parameter list (the `param` of PackItem)
    int kind[2];
    float* packed_fwd[2];
    float* packed_bwd[2];               // null: the transposed stream does not exist yet (built on first backward)
    float step_size, w1, beta2, w2, eps, bc2_sqrt;   // -lr / bc1, 1 - beta1, beta2, 1 - beta2
};

// Position of the parameter element at (row, col) of its [out, in] matrix (a bias: (f, 0)) in the item's packed image,
// or -1 (see pack_kernel for the forward mapping).  No division: the items carry their first element's coordinates.
__device__ __forceinline__ int packed_index(const PackItem& it, int row, int col) {
    if (it.type == ITEM_PLAIN) return row < it.n_valid ? row : -1;                    // bias scalars at [0..n)
    int r = row - it.row0, c = col - it.col0;
    if (it.type == ITEM_VEC) {
        int f;
        if (it.ld == 0) f = row;                                                      // a bias vector
        else if (it.stride == 1) { if (r != 0) return -1; f = c; }                    // a row of the weight
        else { if (c != 0) return -1; f = r; }                                        // a column of the weight
        return (f >= 0 && f < it.n_valid) ? vec_slot(f) : -1;
    }
    if (it.stride != 1) { const int t = r; r = c; c = t; }                            // transposed K block: MFMA rows = inputs
    if (r < 0 || c < 0 || r >= it.rows_valid || c >= it.n_valid) return -1;
    const int m = r >> 5, lane = (r & 31) + 32 * ((c >> 2) & 1), rg = c >> 3, q = c & 3;
    return (((rg * it.mb + m) * 64 + lane) << 2) + q;
}

__global__ __launch_bounds__(256) void adam_pack_kernel(AdamArgs a) {
    const int t = blockIdx.y;
    const int n = a.numel[t], in_f = a.in_f[t];
    const int fld = a.field[t], prm = a.index[t];
    const PackTable& tf = c_fwd[a.kind[fld]];
    const PackTable& tb = c_bwd[a.kind[fld]];
    float* pf = a.packed_fwd[fld];
    float* pb = a.packed_bwd[fld];
    // the items that hold this tensor, found once per block (a tensor sits in <= 10 forward and <= 9 transposed items;
    // walking both 100-item tables per ELEMENT cost 130 us on a 1.2 M-parameter step)
    // (thread i looks at item i of each table; the order of the hits does not matter, every hit writes its own positions)
    if (blockIdx.x * 256 >= n) return;                           // the grid is sized for the largest tensor
    __shared__ int hit_f[kAdamMaxHits], hit_b[kAdamMaxHits], n_hit[2];
    if (threadIdx.x < 2) n_hit[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < tf.n_items; i += 256)
        if (tf.item[i].param == prm) { const int k = atomicAdd(&n_hit[0], 1); if (k < kAdamMaxHits) hit_f[k] = i; }
    if (pb)
        for (int i = threadIdx.x; i < tb.n_items; i += 256)
            if (tb.item[i].param == prm) { const int k = atomicAdd(&n_hit[1], 1); if (k < kAdamMaxHits) hit_b[k] = i; }
    __syncthreads();
    const int nf = n_hit[0], nb = n_hit[1];                      // <= kAdamMaxHits by the static_assert above
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        const float g = a.g[t][e];
        float m = a.m[t][e], v = a.v[t][e], p = a.p[t][e];
        m = m + a.w1 * (g - m);                                  // exp_avg.lerp_(grad, 1 - beta1)
        v = v * a.beta2;                                         // exp_avg_sq.mul_(beta2)
        v = v + (a.w2 * g) * g;                                  //            .addcmul_(grad, grad, value = 1 - beta2)
        const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
        p = p + a.step_size * (m / denom);                       // param.addcdiv_(exp_avg, denom, value = -lr / bc1)
        a.m[t][e] = m; a.v[t][e] = v; a.p[t][e] = p;
        const int row = e / in_f, col = e - row * in_f;
        for (int h = 0; h < nf; ++h) {
            const int i = hit_f[h], k = packed_index(tf.item[i], row, col);
            if (k >= 0) pf[tf.dst_off[i] + k] = p;
        }
        for (int h = 0; h < nb; ++h) {
            const int i = hit_b[h], k = packed_index(tb.item[i], row, col);
            if (k >= 0) pb[tb.dst_off[i] + k] = p;
        }
    }
}

int launch_adam_step(int n_fields, const int* kinds, const int* n_params, float* const* params, const float* const* grads,
                     float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel, float step_size,
                     float one_minus_beta1, float beta2, float one_minus_beta2, float eps, float bc2_sqrt,
                     float* const* packed_fwd, float* const* packed_bwd, hipStream_t stream) {
    AdamArgs a{};
    int t = 0;
    int64_t most = 0;
    for (int f = 0; f < n_fields; ++f) {
        a.kind[f] = kinds[f];
        a.packed_fwd[f] = packed_fwd[f];
        a.packed_bwd[f] = packed_bwd ? packed_bwd[f] : nullptr;
        for (int i = 0; i < n_params[f]; ++i, ++t) {
            if (t >= kAdamMaxParams) { set_error("mi_adam_step: more than %d tensors", kAdamMaxParams); return -1; }
            if (numel[t] > 0x7fffffff) { set_error("mi_adam_step: tensor too large"); return -1; }
            a.p[t] = params[t]; a.g[t] = grads[t]; a.m[t] = exp_avg[t]; a.v[t] = exp_avg_sq[t];
            a.numel[t] = (int)numel[t]; a.field[t] = f; a.index[t] = i;
            if (!(i & 1) && (numel[t + 1] <= 0 || numel[t] % numel[t + 1] != 0)) {
                set_error("mi_adam_step: tensor %d (%lld elements) is not a [out, in] weight of the bias that follows it (%lld)",
                          t, (long long)numel[t], (long long)numel[t + 1]);
                return -1;
            }
            a.in_f[t] = (i & 1) ? 1 : (int)(numel[t] / numel[t + 1]);          // weight [out, in] is followed by its bias [out]
            if (numel[t] > most) most = numel[t];
        }
    }
    a.step_size = step_size; a.w1 = one_minus_beta1; a.beta2 = beta2; a.w2 = one_minus_beta2; a.eps = eps; a.bc2_sqrt = bc2_sqrt;
    if (t == 0) return 0;
    const unsigned bx = (unsigned)((most + 255) / 256 < 64 ? (most + 255) / 256 : 64);
    hipLaunchKernelGGL(adam_pack_kernel, dim3(bx ? bx : 1, t), dim3(256), 0, stream, a);
    return check_launch("adam_pack_kernel");
}

}  // namespace mi
