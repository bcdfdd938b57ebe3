/*
 * mi_render.h - C ABI of libmirender.so: the MI355X (gfx950) implementation of the
 * ray-march -> field-MLP -> alpha-composite path of JeffreyXiang/MSRA-practice-project.
 *
 * The reference has NO plugin/FFI layer: its boundary is the Python
 * module namespace `render` (nerf/render.py, pi_GAN/render.py).  Each entry point below
 * therefore cites the reference *Python function* whose device work it replaces; the
 * Python drop-in modules (msra-practice-project_amd/nerf/render.py, .../pi_GAN/render.py)
 * keep the reference call surface and bind these symbols with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless stated otherwise;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - no allocation, no synchronisation inside: callers pass outputs and workspace;
 *   - return 0 on success, negative MI_E* on error; mi_last_error() gives the message
 *     (thread-local);
 *   - all launches are asynchronous on `stream`.
 */
#ifndef MI_RENDER_H
#define MI_RENDER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_OK 0
#define MI_EINVAL (-1)   /* bad argument (shape, kind, null pointer) */
#define MI_EHIP (-2)     /* HIP runtime error at launch */

/* Field MLP kinds (state-dict layouts: SURVEY.md §8a a6-a8). */
#define MI_FIELD_NERF 0                   /* nerf/nerf.py:52-94   NeRF            */
#define MI_FIELD_SIREN_NERF 1             /* nerf/nerf.py:120-170 SirenNeRF       */
#define MI_FIELD_FILM_SIREN_NERF 2        /* pi_GAN/modules.py:70-118 use_dir=True  */
#define MI_FIELD_FILM_SIREN_NERF_NODIR 3  /* pi_GAN/modules.py:70-118 use_dir=False */
#define MI_FIELD_TINY_NERF 4              /* build-defined 4-layer net for BASELINE C1 */
#define MI_FIELD_KINDS 5

/* Library / build identification. */
int mi_abi_version(void);
const char* mi_last_error(void);

/* ---- field weights -------------------------------------------------------------- */

/* Number of (weight, bias) tensors the kind's state dict holds, in forward order
 * (weight_0, bias_0, weight_1, bias_1, ...). */
int mi_field_num_params(int kind);
/* Floats in the packed MFMA-ordered weight stream of the kind. */
int64_t mi_field_packed_floats(int kind);
/* Multiply-accumulates of the kind's linear layers per point (roofline accounting). */
int64_t mi_field_macs(int kind);
/* Shape of parameter tensor `index` (0 <= index < mi_field_num_params(kind)): weights [rows = out, cols = in], biases
 * [rows = out, cols = 1].  What a host that does not hold an nn.Module needs in order to lay out the tensors it hands to
 * mi_field_pack (the shapes of nerf/nerf.py:59-73, 128-146 and pi_GAN/modules.py:76-94). */
int mi_field_param_shape(int kind, int index, int64_t* rows, int64_t* cols);
/* Repack torch-layout parameters ([out,in] row-major weights, [out] biases; `params` is a
 * HOST array of n_params device pointers in state-dict order) into the packed stream the
 * fused MLP kernel consumes.  Replaces nothing in the reference: it is the hand-off from
 * nn.Module parameters (nerf/nerf.py:59-73, pi_GAN/modules.py:76-94) to the kernel and is
 * called whenever the optimiser has changed the weights.
 * w_0: the frequency of the kind's sin layers, sin(w_0 (gamma (W x + b) + beta)) - FilmSiren's constructor argument
 * (pi_GAN/modules.py:11,22-25,73), any finite w_0 > 0, for the two FiLM kinds; every other kind has no such parameter
 * (Siren hard-codes 30, nerf/nerf.py:112) and takes 30 only.  It travels in the stream (a trailer piece the kernels read as a
 * wave-uniform scalar), so every entry point that consumes the stream evaluates the field it was packed for. */
int mi_field_pack(int kind, const float* const* params, int n_params, float w_0, float* packed, void* stream);

/* ---- fused field MLP forward ---------------------------------------------------- */

/* network(inputs[M,6]) -> [M,4] = (r,g,b,sigma): replaces the model call inside
 * run_network (nerf/render.py:72-74) for a known kind.
 *   x      [n_groups*points_per_group, 6]  (xyz, view_dir)
 *   film   [n_groups, 9, 512] FiLM table (gamma|beta per layer; pi_GAN/modules.py:96-99)
 *          for FILM kinds, else NULL; group g uses film[g]
 *   out    [n_groups*points_per_group, 4] */
int mi_field_eval_points(int kind, const float* packed, const float* film, const float* x,
                         int64_t n_groups, int64_t points_per_group, float* out, void* stream);

/* run_network fused with point generation (nerf/render.py:134-135 / :143-144):
 * pts = o + d*z, view = d/|d|, raw = network([pts, view]).
 *   rays   [n_groups*rays_per_group, 2, 3] (origin, direction)
 *   z      [n_groups*rays_per_group, S]
 *   raw    [n_groups*rays_per_group, S, 4] */
int mi_field_eval_rays(int kind, const float* packed, const float* film, const float* rays,
                       const float* z, int64_t n_groups, int64_t rays_per_group, int n_samples,
                       float* raw, void* stream);

/* ---- sampling / compositing stages ---------------------------------------------- */

/* get_rays (nerf/render.py:7-23) on device, written in render_image's ray-list order
 * (render.py:151-154): rays[ray0 .. ray0+n) of the H*W list, c2w = 12 floats (3x4 row-major,
 * HOST pointer).  compute_f64 = 0 reproduces NumPy with a Python-float focal (all fp32);
 * compute_f64 = 1 reproduces an np.float64 focal (pi_GAN/modules.py:127: fp64 math, rounded to
 * fp32 at the end).  out rays [n,2,3]. */
int mi_gen_rays(int width, int height, double focal, const float* c2w_host, int64_t ray0, int64_t n,
                float* rays, int compute_f64, void* stream);

/* stratified depths (nerf/render.py:123-132): z = lower + (upper-lower)*t_rand.
 * z_lin  [Nc] optional table = linspace(near,far,Nc) (pass torch.linspace's CPU output for
 *        bit parity with the CPU reference; NULL = ATen's scalar formula computed in-kernel)
 * t_rand [n,Nc] device pointer, or NULL to draw U[0,1) from Philox4x32-10 keyed by
 *        (seed, ray0 + ray index, sample index): ray0 = the index of this call's first ray in the
 *        caller's larger ray list, so a frame's jitter does not depend on how it is split over calls
 *        or GPUs.  out z [n,Nc]. */
int mi_sample_coarse(int64_t n, float near_, float far_, int n_coarse, const float* z_lin,
                     const float* t_rand, uint64_t seed, uint64_t ray0, float* z, void* stream);

/* raw_to_outputs (nerf/render.py:78-103).  raw [n,S,4], z [n,S], rays [n,2,3] (direction
 * used for |d|).  out rgb [n,3], depth [n], acc [n], weights [n,S] (weights may be NULL). */
int mi_composite(int64_t n, int n_samples, const float* raw, const float* z, const float* rays,
                 float* rgb, float* depth, float* acc, float* weights, void* stream);

/* sample_pdf + sort (nerf/render.py:140-142): bins = mids of linspace(near,far,Nc),
 * weights[...,1:-1] of the coarse pass, deterministic u = linspace(0,1,Nf), merged with
 * the coarse depths and sorted.  z_lin [Nc] / u_lin [Nf]: optional tables as above.
 * out z_fine [n,Nc+Nf]; z_samples [n,Nf] optional (NULL). */
int mi_sample_fine(int64_t n, float near_, float far_, int n_coarse, int n_fine, const float* z_lin,
                   const float* u_lin, const float* z_coarse, const float* weights, float* z_samples,
                   float* z_fine, void* stream);

/* The same, also returning where every input element went: pos [n, Nc+Nf] int32, pos[e] = index in z_fine of
 * z_coarse[e] (e < Nc) or of z_samples[e - Nc].  What a renderer needs to evaluate ONE field shared by both passes
 * (pi_GAN/modules.py:160-161 passes the same model twice; nerf/train_nerf.py:91,94 with use_fine_model off) at the Nf new
 * depths only: the fine pass of nerf/render.py:143-144 re-evaluates the Nc coarse points with the same field. */
int mi_sample_fine_pos(int64_t n, float near_, float far_, int n_coarse, int n_fine, const float* z_lin,
                       const float* u_lin, const float* z_coarse, const float* weights, float* z_samples,
                       float* z_fine, int* pos, void* stream);

/* raw_fine [n,Nc+Nf,4] in sorted order out of the coarse pass's raw_coarse [n,Nc,4] and raw_samples [n,Nf,4] (the field at
 * z_samples): raw_fine[pos[e]] = e < Nc ? raw_coarse[e] : raw_samples[e - Nc].  Replaces the second field call of
 * nerf/render.py:144 when both passes share one field; bit-identical to it. */
int mi_merge_raw(int64_t n, int n_coarse, int n_fine, const float* raw_coarse, const float* raw_samples, const int* pos,
                 float* raw_fine, void* stream);

/* The transpose of mi_merge_raw for the backward pass: g_raw_coarse[e] (+)= g_raw_fine[pos[e]] (+= when
 * accumulate_coarse: onto the coarse outputs' own gradient), g_raw_samples[i] = g_raw_fine[pos[Nc + i]]. */
int mi_split_grad(int64_t n, int n_coarse, int n_fine, const float* g_raw_fine, const int* pos, float* g_raw_coarse,
                  int accumulate_coarse, float* g_raw_samples, void* stream);

/* sample_pdf as a free-standing function (nerf/render.py:27-56; star-imported by the reference's scripts):
 * bins [n,n_bins], weights [n,n_bins-1] -> samples [n,n_samples]; u_lin [n_samples] optional table as above. */
int mi_sample_pdf(int64_t n, int n_bins, int n_samples, const float* bins, const float* weights,
                  const float* u_lin, float* samples, void* stream);

/* ---- whole path ----------------------------------------------------------------- */

/* render_rays (nerf/render.py:106-147) for known field kinds, all stages on `stream`.
 *   rays [n,2,3]; n = n_groups*rays_per_group; film tables as in mi_field_eval_rays
 *   outs: rgb_c[n,3] depth_c[n] acc_c[n] rgb_f[n,3] depth_f[n] acc_f[n]
 *   seed, ray0: as in mi_sample_coarse (used when t_rand is NULL)
 *   workspace: mi_render_workspace_bytes(n, Nc, Nf) bytes; workspace_bytes = what the caller really provided.
 *   When both passes share one field (same kind, same packed pointer) and the workspace also holds
 *   mi_render_shared_field_extra_bytes(n, Nc, Nf) more, the fine pass evaluates the Nf new depths only (mi_merge_raw);
 *   with Nf = 0 it aliases the coarse outputs.  Results are bit-identical either way. */
int64_t mi_render_workspace_bytes(int64_t n, int n_coarse, int n_fine);
int64_t mi_render_shared_field_extra_bytes(int64_t n, int n_coarse, int n_fine);
int mi_render_rays(int kind_coarse, const float* packed_coarse, int kind_fine, const float* packed_fine,
                   const float* film, const float* rays, int64_t n_groups, int64_t rays_per_group,
                   float near_, float far_, int n_coarse, int n_fine, const float* z_lin, const float* u_lin,
                   const float* t_rand, uint64_t seed, uint64_t ray0, float* rgb_c, float* depth_c, float* acc_c,
                   float* rgb_f, float* depth_f, float* acc_f, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- training: what autograd does for the reference (train_nerf.py:151-168) ------------------ */

/* Backward of raw_to_outputs (nerf/render.py:91-103): dL/d(rgb, depth, acc, weights) -> dL/d(raw) [n,S,4].
 * Any of g_rgb [n,3], g_depth [n], g_acc [n], g_weights [n,S] may be NULL (= zero); g_weights is the cotangent of the
 * weights the function also returns (render.py:103; render_rays itself detaches what it derives from them, :141).
 * z and rays carry no gradient (z_samples is detached at render.py:141). */
int mi_composite_bwd(int64_t n, int n_samples, const float* raw, const float* z, const float* rays,
                     const float* g_rgb, const float* g_depth, const float* g_acc, const float* g_weights, float* g_raw,
                     void* stream);

/* Transposed weight stream for the backward chain (second packed buffer, refreshed with the weights). */
int64_t mi_field_packed_bwd_floats(int kind);
int mi_field_pack_bwd(int kind, const float* const* params, int n_params, float w_0, float* packed_bwd, void* stream);

/* Per-point sizes (floats) of the training buffers, or -1 if the kind has no backward yet:
 * acts = layer inputs saved by the training forward; grads = per-layer dA written by the backward chain.
 * mi_field_bwd_partial_floats(points) = scratch for the dW slab partial sums. */
int64_t mi_field_train_acts_floats(int kind);
int64_t mi_field_train_grads_floats(int kind);
int64_t mi_field_bwd_partial_floats(int64_t points);

/* mi_field_eval_rays that also saves every linear layer's input into `acts`
 * [mi_field_train_acts_floats(kind) * points] (points = n_groups*rays_per_group*n_samples). */
int mi_field_eval_rays_train(int kind, const float* packed, const float* film, const float* rays, const float* z,
                             int64_t n_groups, int64_t rays_per_group, int n_samples, float* raw, float* acts,
                             void* stream);

/* mi_field_eval_points that also saves every linear layer's input (the training forward of `network(x)` called on
 * its own, nerf/render.py:73 outside render_rays): acts [mi_field_train_acts_floats(kind) * n_groups*points_per_group]. */
int mi_field_eval_points_train(int kind, const float* packed, const float* film, const float* x, int64_t n_groups,
                               int64_t points_per_group, float* out, float* acts, void* stream);

/* Backward of network(inputs) over n_groups*points_per_group points: g_raw [points,4] = dL/d(raw) ->
 * parameter gradients (and, for FiLM kinds, the gradient of the FiLM table).
 *   film            [n_groups,9,512] FiLM table of the forward (FiLM kinds) or NULL
 *   grad_params     HOST array of n_params device pointers (torch layouts, state-dict order), OVERWRITTEN
 *   params          HOST array of the n_params parameter tensors themselves (FiLM kinds: the FiLM-table
 *                   gradient is formed from the per-image weight-gradient sums, d gamma = <W, dW_image> +
 *                   b . db_image); NULL for the other kinds
 *   grad_film       [n_groups,9,512] (FiLM kinds) or NULL, OVERWRITTEN
 *   grads_ws        [mi_field_train_grads_floats(kind) * points]
 *   partial_ws      [mi_field_bwd_partial_floats(points)]
 *   film_partial_ws [mi_field_film_partial_floats(n_groups, points_per_group)] FiLM scratch (FiLM kinds) or NULL */
int64_t mi_field_film_partial_floats(int64_t n_groups, int64_t points_per_group);
int mi_field_backward(int kind, const float* packed_bwd, const float* film, const float* acts, float* grads_ws,
                      const float* raw, const float* g_raw, int64_t n_groups, int64_t points_per_group,
                      float* partial_ws, float* film_partial_ws, float* const* grad_params,
                      const float* const* params, int n_params, float* grad_film, void* stream);

/* ---- evaluation stages (frame metrics, density grids) --------------------- */

/* Frame metrics of nerf/test_nerf.py:102-105: mean squared error (psnr = -10 log10 mse) and
 * pytorch_ssim.ssim (nerf/pytorch_ssim/__init__.py:12-37, 66-73: gaussian window, zero padding, groups =
 * channels, C1 = 0.01^2, C2 = 0.03^2) of two image batches [images][channels][height][width] (planar, as the
 * reference passes them: NCHW).  window: HOST array of window_size (odd, <= 31) normalised 1-D gaussian taps
 * (the reference's gaussian(window_size, 1.5), __init__.py:7-10); the 2-D window is its outer product.
 *   out [images][2] = {mse, mean ssim} per image (size_average=True is their mean: images have equal size)
 *   workspace [mi_image_metrics_workspace_floats(...)] */
int64_t mi_image_metrics_workspace_floats(int images, int channels, int height, int width);
int mi_image_metrics(const float* img1, const float* img2, int images, int channels, int height, int width,
                     const float* window, int window_size, float* workspace, float* out, void* stream);

/* Query points of create_mesh's voxel grid (pi_GAN/utils.py:57-75): for overall index i in [head, head+count):
 * (x,y,z) = (i/N/N%N, i/N%N, i%N) * voxel_size + (origin[2], origin[1], origin[0]) - the reference's axis
 * order - and a zero view direction.  voxel_origin: HOST array of 3 floats.  out points [count,6], ready for
 * mi_field_eval_points. */
int mi_grid_points(int n, const float* voxel_origin, float voxel_size, int64_t head, int64_t count, float* points,
                   void* stream);

/* ---- nerf training-loop data path ------------------------------------ */

/* nerf/train_nerf.py:158-167 in one pass over the batch: loss_x = mean((rgb_x - rgb)^2) [+ 0.1 mean((acc_x -
 * alpha)^2) if use_alpha]; loss = loss_fine [+ loss_coarse if use_fine_model].  target [n,4] = (r,g,b,alpha)
 * (batch[:, -4:]).  Writes the gradient seeds d loss / d(rgb_c [n,3], acc_c [n], rgb_f [n,3], acc_f [n]) that
 * render_rays' backward consumes, and out[4] = {loss, mean((rgb_fine - rgb)^2) (psnr = -10 log10 of it),
 * loss_coarse, loss_fine}.  workspace [mi_nerf_loss_workspace_floats(n)]. */
int64_t mi_nerf_loss_workspace_floats(int64_t n);
int mi_nerf_loss(int64_t n, const float* rgb_c, const float* acc_c, const float* rgb_f, const float* acc_f,
                 const float* target, int use_alpha, int use_fine_model, float* g_rgb_c, float* g_acc_c,
                 float* g_rgb_f, float* g_acc_f, float* workspace, float* out, void* stream);

/* The rays_rgba batching table of nerf/train_nerf.py:78-82 built on the device: row image*H*W + pixel =
 * (rays_o, rays_d of get_rays(width, height, focal, pose[image]), r, g, b, a); white_bkgd != 0 composites rgb on
 * white first (rgb*a + 1 - a, train_nerf.py:64-68).  poses [images,12] = c2w[:3,:4] row-major (device),
 * rgba [images,height,width,4] (device); out rays_rgba [images*height*width,10].  compute_f64 as in mi_gen_rays:
 * nonzero when the script's focal is an np.float64 scalar (nerf/data_loader.py:151 returns one), which makes
 * NumPy >= 2 evaluate get_rays in fp64 before train_nerf.py:84 rounds to fp32. */
int mi_ray_bank(int width, int height, double focal, const float* poses, const float* rgba, int white_bkgd,
                int64_t images, float* rays_rgba, int compute_f64, void* stream);

/* One optimiser step of torch.optim.Adam(betas, eps; no weight decay, no amsgrad) - nerf/train_nerf.py:98,168 - on the
 * parameters of 1 or 2 fields, fused with the refresh of their packed weight streams: every tensor's (param, exp_avg,
 * exp_avg_sq) is updated in place and each new parameter value is written to the positions of the field's forward
 * stream (mi_field_pack layout) and, when packed_bwd[f] is non-null, its transposed stream (mi_field_pack_bwd) that
 * hold it.  kinds[n_fields]; params / grads / exp_avg / exp_avg_sq / numel: one entry per tensor, field 0's
 * mi_field_num_params(kind) tensors first, in mi_field_pack order.  The scalars are torch's, computed by the caller in
 * double: step_size = -lr / (1 - beta1^t), bias_correction2_sqrt = sqrt(1 - beta2^t).  The streams must already
 * hold the current parameters (their padding entries are not rewritten). */
int mi_adam_step(int n_fields, const int* kinds, float* const* params, const float* const* grads, float* const* exp_avg,
                 float* const* exp_avg_sq, const int64_t* numel, float step_size, float one_minus_beta1, float beta2,
                 float one_minus_beta2, float eps, float bias_correction2_sqrt, float* const* packed_fwd,
                 float* const* packed_bwd, void* stream);

/* ---- measurement hooks (bench.py) ---------------------------------------------------- */

/* HIP events owned by the library's HIP runtime (the one the kernels launch on), so a host
 * program can time individual launches without linking HIP itself.  mi_event_elapsed_ms
 * synchronises on `stop`. */
void* mi_event_create(void);
void mi_event_destroy(void* ev);
int mi_event_record(void* ev, void* stream);
int mi_event_elapsed_ms(void* start, void* stop, float* ms);
/* While set (thread-local; pass NULLs to clear), mi_render_rays records these events on its
 * stream immediately before/after the coarse and the fine field-MLP launch. */
void mi_render_set_mlp_events(void* start_coarse, void* stop_coarse, void* start_fine, void* stop_fine);

#ifdef __cplusplus
}
#endif
#endif /* MI_RENDER_H */
