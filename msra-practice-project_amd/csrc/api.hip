// api.hip - extern "C" surface of libmirender.so (see include/mi_render.h).
// Argument validation lives here; kernels assume validated shapes.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/mi_render.h"
#include "field_layout.h"
#include "mi_common.h"

namespace mi {

static thread_local char g_err[512] = "";
static thread_local hipEvent_t g_mlp_ev[4] = {nullptr, nullptr, nullptr, nullptr};
#ifdef MI_PROFILE_STAMPS
static unsigned long long* g_stamps = nullptr;
extern unsigned long long* g_bwd_stamps;          // field_mlp_bwd.hip
#endif

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return MI_EHIP;
    }
    return MI_OK;
}

// implemented in field_mlp.hip / field_mlp_bwd.hip
const PackTable* host_table(int kind);
const PackTable* host_table_bwd(int kind);
static const int kNumLayers[MI_FIELD_KINDS] = {12, 12, 11, 11, 7};
// multiply-accumulates of the linear layers per point (a6, a7, a8)
static const int64_t kMacs[MI_FIELD_KINDS] = {591488, 559616, 526848, 526080, 248448};
// (out, in) of every linear layer in mi_field_pack order (nerf/nerf.py:59-73, 128-146; pi_GAN/modules.py:76-94)
static const int kLayerDims[MI_FIELD_KINDS][12][2] = {
    {{256, 60}, {256, 256}, {256, 256}, {256, 256}, {256, 256}, {256, 316}, {256, 256}, {256, 256}, {256, 256}, {128, 280}, {1, 256}, {3, 128}},
    {{256, 3}, {256, 256}, {256, 256}, {256, 256}, {256, 256}, {256, 259}, {256, 256}, {256, 256}, {256, 256}, {128, 259}, {1, 256}, {3, 128}},
    {{256, 3}, {256, 256}, {256, 256}, {256, 256}, {256, 256}, {256, 256}, {256, 256}, {256, 256}, {1, 256}, {256, 259}, {3, 256}, {0, 0}},
    {{256, 3}, {256, 256}, {256, 256}, {256, 256}, {256, 256}, {256, 256}, {256, 256}, {256, 256}, {1, 256}, {256, 256}, {3, 256}, {0, 0}},
    {{256, 60}, {256, 256}, {256, 256}, {256, 256}, {128, 280}, {1, 256}, {3, 128}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}}};

static bool bad_kind(int kind) {
    if (kind < 0 || kind >= MI_FIELD_KINDS) { set_error("unknown field kind %d", kind); return true; }
    return false;
}
static bool is_film(int kind) { return kind == MI_FIELD_FILM_SIREN_NERF || kind == MI_FIELD_FILM_SIREN_NERF_NODIR; }

static int eval_common(int kind, const float* packed, const float* film, const float* a, const float* z,
                       int64_t n_groups, int64_t ppg, int64_t rpg, int S, int mode, float* out, hipStream_t s,
                       float* save = nullptr) {
    if (bad_kind(kind)) return MI_EINVAL;
    if (!packed || !a || !out || (mode == 1 && !z)) { set_error("null pointer argument"); return MI_EINVAL; }
    if (is_film(kind) && !film) { set_error("FiLM kind needs a film table"); return MI_EINVAL; }
    if (n_groups < 0 || ppg < 0) { set_error("negative size"); return MI_EINVAL; }
    if (n_groups == 0 || ppg == 0) return MI_OK;
    MlpArgs args;
    args.packed = packed; args.film = is_film(kind) ? film : nullptr; args.a = a; args.z = z; args.out = out;
    args.points_per_group = ppg; args.rays_per_group = rpg; args.tiles_per_group = (ppg + 127) / 128;
    args.n_samples = S; args.mode = mode; args.save = save; args.save_points = n_groups * ppg;
#ifdef MI_PROFILE_STAMPS
    args.stamps = g_stamps;
#else
    args.stamps = nullptr;
#endif
    return launch_mlp(kind, args, n_groups, s);
}

}  // namespace mi

using namespace mi;

extern "C" {

int mi_abi_version(void) { return 4; }
const char* mi_last_error(void) { return g_err; }

int mi_field_num_params(int kind) { return bad_kind(kind) ? MI_EINVAL : 2 * kNumLayers[kind]; }
int64_t mi_field_packed_floats(int kind) { return bad_kind(kind) ? MI_EINVAL : packed_floats(*host_table(kind)); }
int64_t mi_field_macs(int kind) { return bad_kind(kind) ? MI_EINVAL : kMacs[kind]; }
int mi_field_param_shape(int kind, int index, int64_t* rows, int64_t* cols) {
    if (bad_kind(kind)) return MI_EINVAL;
    if (index < 0 || index >= 2 * kNumLayers[kind] || !rows || !cols) { set_error("mi_field_param_shape: bad arguments"); return MI_EINVAL; }
    *rows = kLayerDims[kind][index / 2][0];
    *cols = (index & 1) ? 1 : kLayerDims[kind][index / 2][1];
    return MI_OK;
}

// w_0 of the kind's sin layers: any finite w_0 > 0 for the FiLM kinds (FilmSiren's constructor argument, pi_GAN/modules.py:11,73);
// every other kind has no such parameter (Siren hard-codes 30, nerf/nerf.py:112; the ReLU kinds have no sin) and takes 30 only.
static int check_w0(int kind, float w_0, const char* fn) {
    if (is_film(kind)) {
        if (!(w_0 > 0.f) || !(w_0 < 1e30f)) { set_error("%s: w_0 = %g (FiLM kinds need a finite w_0 > 0)", fn, (double)w_0); return MI_EINVAL; }
    } else if (w_0 != 30.f) {
        set_error("%s: w_0 = %g, but only the FiLM kinds have a w_0 (pass 30 for kind %d)", fn, (double)w_0, kind);
        return MI_EINVAL;
    }
    return MI_OK;
}

int mi_field_pack(int kind, const float* const* params, int n_params, float w_0, float* packed, void* stream) {
    if (bad_kind(kind)) return MI_EINVAL;
    if (!params || !packed) { set_error("null pointer argument"); return MI_EINVAL; }
    if (n_params != 2 * kNumLayers[kind]) {
        set_error("kind %d expects %d parameter tensors, got %d", kind, 2 * kNumLayers[kind], n_params);
        return MI_EINVAL;
    }
    for (int i = 0; i < n_params; ++i)
        if (!params[i]) { set_error("parameter %d is null", i); return MI_EINVAL; }
    if (int rc = check_w0(kind, w_0, "mi_field_pack")) return rc;
    return launch_pack(kind, params, n_params, w_0, packed, (hipStream_t)stream);
}

int mi_field_eval_points(int kind, const float* packed, const float* film, const float* x, int64_t n_groups,
                         int64_t points_per_group, float* out, void* stream) {
    return eval_common(kind, packed, film, x, nullptr, n_groups, points_per_group, 0, 1, 0, out, (hipStream_t)stream);
}

int mi_field_eval_rays(int kind, const float* packed, const float* film, const float* rays, const float* z,
                       int64_t n_groups, int64_t rays_per_group, int n_samples, float* raw, void* stream) {
    if (n_samples <= 0) { set_error("n_samples must be positive"); return MI_EINVAL; }
    return eval_common(kind, packed, film, rays, z, n_groups, rays_per_group * n_samples, rays_per_group, n_samples, 1,
                       raw, (hipStream_t)stream);
}

int mi_gen_rays(int width, int height, double focal, const float* c2w_host, int64_t ray0, int64_t n, float* rays,
                int compute_f64, void* stream) {
    if (width <= 0 || height <= 0 || !c2w_host || !rays || ray0 < 0 || n < 0 || ray0 + n > (int64_t)width * height) {
        set_error("mi_gen_rays: bad arguments");
        return MI_EINVAL;
    }
    return launch_gen_rays(width, height, focal, c2w_host, ray0, n, rays, compute_f64, (hipStream_t)stream);
}

int mi_sample_coarse(int64_t n, float near_, float far_, int n_coarse, const float* z_lin, const float* t_rand,
                     uint64_t seed, uint64_t ray0, float* z, void* stream) {
    if (n < 0 || n_coarse < 1 || !z) { set_error("mi_sample_coarse: bad arguments"); return MI_EINVAL; }
    return launch_sample_coarse(n, near_, far_, n_coarse, z_lin, t_rand, seed, ray0, z, (hipStream_t)stream);
}

int mi_composite(int64_t n, int n_samples, const float* raw, const float* z, const float* rays, float* rgb,
                 float* depth, float* acc, float* weights, void* stream) {
    if (n < 0 || n_samples < 1 || !raw || !z || !rays || !rgb || !depth || !acc) {
        set_error("mi_composite: bad arguments");
        return MI_EINVAL;
    }
    return launch_composite(n, n_samples, raw, z, rays, rgb, depth, acc, weights, (hipStream_t)stream);
}

int mi_sample_fine(int64_t n, float near_, float far_, int n_coarse, int n_fine, const float* z_lin,
                   const float* u_lin, const float* z_coarse, const float* weights, float* z_samples, float* z_fine,
                   void* stream) {
    // sample_pdf is called with mids (Nc-1 bins) and weights[1:-1] (Nc-2): needs Nc >= 3
    if (n < 0 || n_coarse < 3 || n_fine < 0 || !z_coarse || !weights || !z_fine) {
        set_error("mi_sample_fine: bad arguments (need Nc >= 3)");
        return MI_EINVAL;
    }
    return launch_sample_fine(n, near_, far_, n_coarse, n_fine, z_lin, u_lin, z_coarse, weights, z_samples, z_fine, nullptr,
                              (hipStream_t)stream);
}

int mi_sample_fine_pos(int64_t n, float near_, float far_, int n_coarse, int n_fine, const float* z_lin,
                       const float* u_lin, const float* z_coarse, const float* weights, float* z_samples, float* z_fine,
                       int* pos, void* stream) {
    if (n < 0 || n_coarse < 3 || n_fine < 0 || !z_coarse || !weights || !z_fine || !pos) {
        set_error("mi_sample_fine_pos: bad arguments (need Nc >= 3 and a pos table)");
        return MI_EINVAL;
    }
    return launch_sample_fine(n, near_, far_, n_coarse, n_fine, z_lin, u_lin, z_coarse, weights, z_samples, z_fine, pos,
                              (hipStream_t)stream);
}

int mi_merge_raw(int64_t n, int n_coarse, int n_fine, const float* raw_coarse, const float* raw_samples, const int* pos,
                 float* raw_fine, void* stream) {
    if (n < 0 || n_coarse < 1 || n_fine < 0 || !raw_coarse || (n_fine > 0 && !raw_samples) || !pos || !raw_fine) {
        set_error("mi_merge_raw: bad arguments");
        return MI_EINVAL;
    }
    return launch_merge_raw(n, n_coarse, n_fine, raw_coarse, raw_samples, pos, raw_fine, (hipStream_t)stream);
}

int mi_split_grad(int64_t n, int n_coarse, int n_fine, const float* g_raw_fine, const int* pos, float* g_raw_coarse,
                  int accumulate_coarse, float* g_raw_samples, void* stream) {
    if (n < 0 || n_coarse < 1 || n_fine < 0 || !g_raw_fine || !pos || !g_raw_coarse || (n_fine > 0 && !g_raw_samples)) {
        set_error("mi_split_grad: bad arguments");
        return MI_EINVAL;
    }
    return launch_split_grad(n, n_coarse, n_fine, g_raw_fine, pos, g_raw_coarse, accumulate_coarse, g_raw_samples,
                             (hipStream_t)stream);
}

int mi_sample_pdf(int64_t n, int n_bins, int n_samples, const float* bins, const float* weights, const float* u_lin,
                  float* samples, void* stream) {
    if (n < 0 || n_bins < 2 || n_samples < 0 || !bins || !weights || (n_samples > 0 && !samples)) {
        set_error("mi_sample_pdf: bad arguments (need at least 2 bins)");
        return MI_EINVAL;
    }
    return launch_sample_pdf(n, n_bins, n_samples, bins, weights, u_lin, samples, (hipStream_t)stream);
}

int64_t mi_render_workspace_bytes(int64_t n, int n_coarse, int n_fine) {
    const int64_t S = (int64_t)n_coarse + n_fine;
    // z_c[n,Nc] raw_c[n,Nc,4] w_c[n,Nc] z_f[n,S] raw_f[n,S,4]; each region 256-byte aligned
    int64_t f = 0;
    auto add = [&](int64_t x) { f += (x + 63) / 64 * 64; };
    add(n * n_coarse); add(n * n_coarse * 4); add(n * n_coarse); add(n * S); add(n * S * 4);
    return f * (int64_t)sizeof(float);
}

int64_t mi_render_shared_field_extra_bytes(int64_t n, int n_coarse, int n_fine) {
    // z_samples[n,Nf] raw_samples[n,Nf,4] pos[n,Nc+Nf] behind the regions of mi_render_workspace_bytes
    int64_t f = 0;
    auto add = [&](int64_t x) { f += (x + 63) / 64 * 64; };
    add(n * n_fine); add(n * n_fine * 4); add(n * ((int64_t)n_coarse + n_fine));
    return f * (int64_t)sizeof(float);
}

int mi_render_rays(int kind_coarse, const float* packed_coarse, int kind_fine, const float* packed_fine,
                   const float* film, const float* rays, int64_t n_groups, int64_t rays_per_group, float near_,
                   float far_, int n_coarse, int n_fine, const float* z_lin, const float* u_lin, const float* t_rand,
                   uint64_t seed, uint64_t ray0, float* rgb_c, float* depth_c, float* acc_c, float* rgb_f,
                   float* depth_f, float* acc_f, void* workspace, int64_t workspace_bytes, void* stream) {
    if (n_groups < 0 || rays_per_group < 0 || n_coarse < 1 || n_fine < 0) { set_error("mi_render_rays: bad sizes"); return MI_EINVAL; }
    if (n_groups * rays_per_group == 0) return MI_OK;          // no rays: nothing to launch, no buffer is touched
    if (!workspace || !rays || !rgb_c || !depth_c || !acc_c || !rgb_f || !depth_f || !acc_f) {
        set_error("mi_render_rays: null pointer argument");
        return MI_EINVAL;
    }
    const int64_t n = n_groups * rays_per_group;
    const int S = n_coarse + n_fine;
    const int64_t base_bytes = mi_render_workspace_bytes(n, n_coarse, n_fine);
    if (workspace_bytes < base_bytes) {
        set_error("mi_render_rays: workspace of %lld bytes, mi_render_workspace_bytes says %lld", (long long)workspace_bytes,
                  (long long)base_bytes);
        return MI_EINVAL;
    }
    float* ws = (float*)workspace;
    auto take = [&](int64_t x) { float* p = ws; ws += (x + 63) / 64 * 64; return p; };
    float* z_c = take(n * n_coarse);
    float* raw_c = take(n * n_coarse * 4);
    float* w_c = take(n * n_coarse);
    float* z_f = take(n * (int64_t)S);
    float* raw_f = take(n * (int64_t)S * 4);
    int rc;
    if ((rc = mi_sample_coarse(n, near_, far_, n_coarse, z_lin, t_rand, seed, ray0, z_c, stream))) return rc;
    hipStream_t hs = (hipStream_t)stream;
    if (g_mlp_ev[0]) (void)hipEventRecord(g_mlp_ev[0], hs);
    if ((rc = mi_field_eval_rays(kind_coarse, packed_coarse, film, rays, z_c, n_groups, rays_per_group, n_coarse, raw_c,
                                 stream))) return rc;
    if (g_mlp_ev[1]) (void)hipEventRecord(g_mlp_ev[1], hs);
    if ((rc = mi_composite(n, n_coarse, raw_c, z_c, rays, rgb_c, depth_c, acc_c, w_c, stream))) return rc;
    if (n_fine == 0 && kind_fine == kind_coarse && packed_fine == packed_coarse) {
        // render.py:140-145 with Nf = 0 and one model: sort(z_coarse) == z_coarse, so the second pass would
        // re-evaluate identical inputs; alias its outputs instead.
        hipStream_t s = (hipStream_t)stream;
        if (hipMemcpyAsync(rgb_f, rgb_c, n * 3 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess ||
            hipMemcpyAsync(depth_f, depth_c, n * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess ||
            hipMemcpyAsync(acc_f, acc_c, n * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
            set_error("mi_render_rays: output alias copy failed");
            return MI_EHIP;
        }
        return MI_OK;
    }
    if (kind_fine == kind_coarse && packed_fine == packed_coarse &&
        workspace_bytes >= base_bytes + mi_render_shared_field_extra_bytes(n, n_coarse, n_fine)) {
        // One field for both passes: Nc of the fine pass's Nc + Nf points are the coarse pass's points - evaluate the Nf
        // new ones only and merge (render_stages.hip: merge_raw_kernel).  Needs the extra workspace regions; a caller
        // that did not provide them gets the plain path below (same results).
        float* z_s = take(n * n_fine);
        float* raw_s = take(n * (int64_t)n_fine * 4);
        int* pos = (int*)take(n * (int64_t)S);
        if ((rc = mi_sample_fine_pos(n, near_, far_, n_coarse, n_fine, z_lin, u_lin, z_c, w_c, z_s, z_f, pos, stream))) return rc;
        if (g_mlp_ev[2]) (void)hipEventRecord(g_mlp_ev[2], hs);
        if ((rc = mi_field_eval_rays(kind_fine, packed_fine, film, rays, z_s, n_groups, rays_per_group, n_fine, raw_s, stream)))
            return rc;
        if (g_mlp_ev[3]) (void)hipEventRecord(g_mlp_ev[3], hs);
        if ((rc = mi_merge_raw(n, n_coarse, n_fine, raw_c, raw_s, pos, raw_f, stream))) return rc;
        return mi_composite(n, S, raw_f, z_f, rays, rgb_f, depth_f, acc_f, nullptr, stream);
    }
    if ((rc = mi_sample_fine(n, near_, far_, n_coarse, n_fine, z_lin, u_lin, z_c, w_c, nullptr, z_f, stream))) return rc;
    if (g_mlp_ev[2]) (void)hipEventRecord(g_mlp_ev[2], hs);
    if ((rc = mi_field_eval_rays(kind_fine, packed_fine, film, rays, z_f, n_groups, rays_per_group, S, raw_f, stream)))
        return rc;
    if (g_mlp_ev[3]) (void)hipEventRecord(g_mlp_ev[3], hs);
    return mi_composite(n, S, raw_f, z_f, rays, rgb_f, depth_f, acc_f, nullptr, stream);
}

int mi_composite_bwd(int64_t n, int n_samples, const float* raw, const float* z, const float* rays,
                     const float* g_rgb, const float* g_depth, const float* g_acc, const float* g_weights, float* g_raw,
                     void* stream) {
    if (n < 0 || n_samples < 1 || !raw || !z || !rays || !g_raw) { set_error("mi_composite_bwd: bad arguments"); return MI_EINVAL; }
    return launch_composite_bwd(n, n_samples, raw, z, rays, g_rgb, g_depth, g_acc, g_weights, g_raw, (hipStream_t)stream);
}

int64_t mi_field_packed_bwd_floats(int kind) { return bad_kind(kind) ? MI_EINVAL : packed_floats(*host_table_bwd(kind)); }

int mi_field_pack_bwd(int kind, const float* const* params, int n_params, float w_0, float* packed_bwd, void* stream) {
    if (bad_kind(kind)) return MI_EINVAL;
    if (!params || !packed_bwd || n_params != 2 * kNumLayers[kind]) { set_error("mi_field_pack_bwd: bad arguments"); return MI_EINVAL; }
    if (int rc = check_w0(kind, w_0, "mi_field_pack_bwd")) return rc;
    return launch_pack_bwd(kind, params, n_params, w_0, packed_bwd, (hipStream_t)stream);
}

int64_t mi_field_train_acts_floats(int kind) { return bad_kind(kind) ? MI_EINVAL : train_acts_floats(kind); }
int64_t mi_field_train_grads_floats(int kind) { return bad_kind(kind) ? MI_EINVAL : train_grads_floats(kind); }
int64_t mi_field_bwd_partial_floats(int64_t points) { return bwd_partial_floats(points); }

int mi_field_eval_rays_train(int kind, const float* packed, const float* film, const float* rays, const float* z,
                             int64_t n_groups, int64_t rays_per_group, int n_samples, float* raw, float* acts,
                             void* stream) {
    if (n_samples <= 0 || !acts) { set_error("mi_field_eval_rays_train: bad arguments"); return MI_EINVAL; }
    if (bad_kind(kind)) return MI_EINVAL;
    if (train_acts_floats(kind) < 0) { set_error("kind %d has no training path yet", kind); return MI_EINVAL; }
    return eval_common(kind, packed, film, rays, z, n_groups, rays_per_group * n_samples, rays_per_group, n_samples, 1,
                       raw, (hipStream_t)stream, acts);
}

int mi_field_eval_points_train(int kind, const float* packed, const float* film, const float* x, int64_t n_groups,
                               int64_t points_per_group, float* out, float* acts, void* stream) {
    if (!acts) { set_error("mi_field_eval_points_train: bad arguments"); return MI_EINVAL; }
    if (bad_kind(kind)) return MI_EINVAL;
    if (train_acts_floats(kind) < 0) { set_error("kind %d has no training path yet", kind); return MI_EINVAL; }
    return eval_common(kind, packed, film, x, nullptr, n_groups, points_per_group, 0, 1, 0, out, (hipStream_t)stream, acts);
}

int64_t mi_field_film_partial_floats(int64_t n_groups, int64_t points_per_group) {
    return film_partial_floats(n_groups, points_per_group);
}

int mi_field_backward(int kind, const float* packed_bwd, const float* film, const float* acts, float* grads_ws,
                      const float* raw, const float* g_raw, int64_t n_groups, int64_t points_per_group,
                      float* partial_ws, float* film_partial_ws, float* const* grad_params,
                      const float* const* params, int n_params, float* grad_film, void* stream) {
    if (bad_kind(kind)) return MI_EINVAL;
    if (!packed_bwd || !acts || !grads_ws || !raw || !g_raw || !partial_ws || !grad_params ||
        n_params != 2 * kNumLayers[kind]) { set_error("mi_field_backward: bad arguments"); return MI_EINVAL; }
    const bool film_kind = kind == 2 || kind == 3;
    for (int i = 0; i < n_params; ++i) {
        if (!grad_params[i]) { set_error("gradient pointer %d is null", i); return MI_EINVAL; }
        if (film_kind && (!params || !params[i])) { set_error("FiLM kinds need parameter pointer %d", i); return MI_EINVAL; }
    }
    return launch_field_backward(kind, packed_bwd, acts, grads_ws, raw, g_raw, n_groups, points_per_group, film,
                                 film_partial_ws, grad_film, partial_ws, grad_params, params, (hipStream_t)stream);
}

#ifdef MI_PROFILE_STAMPS
void mi_debug_set_stamps(void* p) { g_stamps = (unsigned long long*)p; g_bwd_stamps = (unsigned long long*)p; }
#endif

int64_t mi_image_metrics_workspace_floats(int images, int channels, int height, int width) {
    if (images <= 0 || channels <= 0 || height <= 0 || width <= 0) return 0;
    return image_metrics_workspace_floats(images, channels, height, width);
}

int mi_image_metrics(const float* img1, const float* img2, int images, int channels, int height, int width,
                     const float* window, int window_size, float* workspace, float* out, void* stream) {
    if (!img1 || !img2 || !window || !workspace || !out || images <= 0 || channels <= 0 || height <= 0 || width <= 0 ||
        window_size < 1 || window_size > 31 || window_size % 2 == 0) {
        set_error("mi_image_metrics: bad arguments (window must be odd and <= 31)");
        return MI_EINVAL;
    }
    return launch_image_metrics(img1, img2, images, channels, height, width, window, window_size, workspace, out,
                                (hipStream_t)stream);
}

int mi_grid_points(int n, const float* voxel_origin, float voxel_size, int64_t head, int64_t count, float* points,
                   void* stream) {
    if (n <= 0 || !voxel_origin || head < 0 || count < 0 || head + count > (int64_t)n * n * n || (count > 0 && !points)) {
        set_error("mi_grid_points: bad arguments (need 0 <= head, head + count <= n^3)");
        return MI_EINVAL;
    }
    return launch_grid_points(n, voxel_origin, voxel_size, head, count, points, (hipStream_t)stream);
}

int64_t mi_nerf_loss_workspace_floats(int64_t n) { return n > 0 ? nerf_loss_workspace_floats(n) : 0; }

int mi_nerf_loss(int64_t n, const float* rgb_c, const float* acc_c, const float* rgb_f, const float* acc_f,
                 const float* target, int use_alpha, int use_fine_model, float* g_rgb_c, float* g_acc_c, float* g_rgb_f,
                 float* g_acc_f, float* workspace, float* out, void* stream) {
    if (n <= 0 || !rgb_c || !acc_c || !rgb_f || !acc_f || !target || !g_rgb_c || !g_acc_c || !g_rgb_f || !g_acc_f ||
        !workspace || !out) { set_error("mi_nerf_loss: bad arguments"); return MI_EINVAL; }
    return launch_nerf_loss(n, rgb_c, acc_c, rgb_f, acc_f, target, use_alpha, use_fine_model, g_rgb_c, g_acc_c, g_rgb_f,
                            g_acc_f, workspace, out, (hipStream_t)stream);
}

int mi_ray_bank(int width, int height, double focal, const float* poses, const float* rgba, int white_bkgd, int64_t images,
                float* rays_rgba, int compute_f64, void* stream) {
    if (width <= 0 || height <= 0 || images < 0 || !poses || !rgba || (images > 0 && !rays_rgba)) {
        set_error("mi_ray_bank: bad arguments");
        return MI_EINVAL;
    }
    return launch_ray_bank(width, height, focal, poses, rgba, white_bkgd, images, rays_rgba, compute_f64, (hipStream_t)stream);
}

int mi_adam_step(int n_fields, const int* kinds, float* const* params, const float* const* grads, float* const* exp_avg,
                 float* const* exp_avg_sq, const int64_t* numel, float step_size, float one_minus_beta1, float beta2,
                 float one_minus_beta2, float eps, float bias_correction2_sqrt, float* const* packed_fwd,
                 float* const* packed_bwd, void* stream) {
    if (n_fields < 1 || n_fields > 2 || !kinds || !params || !grads || !exp_avg || !exp_avg_sq || !numel || !packed_fwd) {
        set_error("mi_adam_step: bad arguments (1 or 2 fields, non-null tables)");
        return MI_EINVAL;
    }
    int n_params[2] = {0, 0}, total = 0;
    for (int f = 0; f < n_fields; ++f) {
        if (bad_kind(kinds[f])) return MI_EINVAL;
        if (!packed_fwd[f]) { set_error("mi_adam_step: field %d has no packed stream", f); return MI_EINVAL; }
        n_params[f] = 2 * kNumLayers[kinds[f]];
        total += n_params[f];
    }
    for (int t = 0; t < total; ++t)
        if (!params[t] || !grads[t] || !exp_avg[t] || !exp_avg_sq[t] || numel[t] < 0) {
            set_error("mi_adam_step: tensor %d has a null pointer", t);
            return MI_EINVAL;
        }
    return launch_adam_step(n_fields, kinds, n_params, params, grads, exp_avg, exp_avg_sq, numel, step_size, one_minus_beta1,
                            beta2, one_minus_beta2, eps, bias_correction2_sqrt, packed_fwd, packed_bwd, (hipStream_t)stream);
}

void* mi_event_create(void) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { set_error("hipEventCreate failed"); return nullptr; }
    return (void*)e;
}
void mi_event_destroy(void* ev) { if (ev) (void)hipEventDestroy((hipEvent_t)ev); }
int mi_event_record(void* ev, void* stream) {
    if (hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) != hipSuccess) { set_error("hipEventRecord failed"); return MI_EHIP; }
    return MI_OK;
}
int mi_event_elapsed_ms(void* start, void* stop, float* ms) {
    if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess ||
        hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) {
        set_error("hipEventElapsedTime failed");
        return MI_EHIP;
    }
    return MI_OK;
}
void mi_render_set_mlp_events(void* start_coarse, void* stop_coarse, void* start_fine, void* stop_fine) {
    g_mlp_ev[0] = (hipEvent_t)start_coarse; g_mlp_ev[1] = (hipEvent_t)stop_coarse;
    g_mlp_ev[2] = (hipEvent_t)start_fine; g_mlp_ev[3] = (hipEvent_t)stop_fine;
}

}  // extern "C"
