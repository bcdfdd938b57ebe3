// mi_common.h - small shared helpers for the libmirender kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>
#include <type_traits>
#include <utility>

namespace mi {

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
// f(integral_constant<int,0>) ... f(integral_constant<int,N-1>), fully unrolled at compile time
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// error plumbing for the C ABI (api.hip)
void set_error(const char* fmt, ...);
int check_launch(const char* what);

// hipFuncSetAttribute configures the CURRENT device's copy of a kernel, so "once" means once per device ordinal:
// a process that drives several GPUs (DataParallel's thread per replica, pi_GAN/train.py:50) sets it on each.
// run(f) calls f() the first time the calling thread's current device is seen (serialised); lock-free afterwards.
class PerDeviceOnce {
    std::atomic<uint64_t> done_[4] = {};          // 256 device ordinals
    std::mutex mu_;
public:
    template <class F>
    int run(F&& f) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 256) { set_error("hipGetDevice failed"); return -2; }
        const uint64_t bit = 1ull << (dev & 63);
        if (done_[dev >> 6].load(std::memory_order_acquire) & bit) return 0;
        std::lock_guard<std::mutex> g(mu_);
        if (done_[dev >> 6].load(std::memory_order_relaxed) & bit) return 0;
        const int rc = f();
        if (rc == 0) done_[dev >> 6].fetch_or(bit, std::memory_order_release);
        return rc;
    }
};

// arguments of the fused field-MLP kernels (field_mlp.hip)
struct MlpArgs {
    const float* packed;    // packed weight stream (field_layout.h)
    const float* film;      // [groups][9][512] or null
    const float* a;         // points x[M,6] (mode 0) or rays [N,2,3] (mode 1)
    const float* z;         // [N,S] (mode 1)
    float* out;             // [M,4]
    int64_t points_per_group;
    int64_t rays_per_group;
    int64_t tiles_per_group;
    int n_samples;
    int mode;
    float* save;            // training: saved layer inputs, region r = save + act_offset(r) * save_points
    int64_t save_points;    // points in this launch (row count of every saved region)
    unsigned long long* stamps;   // diagnostic build (-DMI_PROFILE_STAMPS) only: [block][128] s_memtime values
};

// In-kernel cycle stamps for the diagnostic build (csrc/build.py --profile -> gpurun_tools/libmirender_prof.so);
// the product build compiles them out.
#ifdef MI_PROFILE_STAMPS
#define MI_STAMP(a, i) do { if (threadIdx.x == 0 && (a).stamps) (a).stamps[(int64_t)blockIdx.x * 128 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MI_STAMP(a, i) do { } while (0)
#endif

// host launchers (field_mlp.hip, render_stages.hip)
int launch_pack(int kind, const float* const* params, int n_params, float w0, float* packed, hipStream_t stream);
int launch_mlp(int kind, const MlpArgs& a, int64_t n_groups, hipStream_t stream);
int launch_pack_bwd(int kind, const float* const* params, int n_params, float w0, float* packed, hipStream_t stream);
int64_t train_acts_floats(int kind);
int64_t train_grads_floats(int kind);
int64_t bwd_partial_floats(int64_t P);
int64_t film_partial_floats(int64_t n_groups, int64_t points_per_group);
int launch_field_backward(int kind, const float* packed_bwd, const float* acts, float* grads, const float* raw,
                          const float* g_raw, int64_t n_groups, int64_t points_per_group, const float* film,
                          float* film_partial, float* grad_film, float* partial, float* const* gp,
                          const float* const* params, hipStream_t stream);
int launch_gen_rays(int width, int height, double focal, const float* c2w, int64_t ray0, int64_t n, float* rays,
                    int compute_f64, hipStream_t stream);
int launch_sample_coarse(int64_t n, float near_, float far_, int nc, const float* z_lin, const float* t_rand,
                         uint64_t seed, uint64_t ray0, float* z, hipStream_t stream);
int launch_composite(int64_t n, int S, const float* raw, const float* z, const float* rays, float* rgb, float* depth,
                     float* acc, float* weights, hipStream_t stream);
int launch_composite_bwd(int64_t n, int S, const float* raw, const float* z, const float* rays, const float* g_rgb,
                         const float* g_depth, const float* g_acc, const float* g_w, float* g_raw, hipStream_t stream);
int64_t image_metrics_workspace_floats(int images, int channels, int H, int W);
int launch_image_metrics(const float* img1, const float* img2, int images, int channels, int H, int W,
                         const float* window, int window_size, float* workspace, float* out, hipStream_t stream);
int launch_grid_points(int N, const float* origin, float voxel_size, int64_t head, int64_t count, float* pts,
                       hipStream_t stream);
int64_t nerf_loss_workspace_floats(int64_t n);
int launch_nerf_loss(int64_t n, const float* rgb_c, const float* acc_c, const float* rgb_f, const float* acc_f,
                     const float* target, int use_alpha, int use_fine, float* g_rgb_c, float* g_acc_c, float* g_rgb_f,
                     float* g_acc_f, float* workspace, float* out, hipStream_t stream);
int launch_ray_bank(int width, int height, double focal, const float* poses, const float* rgba, int white_bkgd,
                    int64_t images, float* out, int compute_f64, hipStream_t stream);
int launch_adam_step(int n_fields, const int* kinds, const int* n_params, float* const* params, const float* const* grads,
                     float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel, float step_size,
                     float one_minus_beta1, float beta2, float one_minus_beta2, float eps, float bc2_sqrt,
                     float* const* packed_fwd, float* const* packed_bwd, hipStream_t stream);
int launch_sample_pdf(int64_t n, int nb, int ns, const float* bins, const float* weights, const float* u_lin, float* out,
                      hipStream_t stream);
int launch_sample_fine(int64_t n, float near_, float far_, int nc, int nf, const float* z_lin, const float* u_lin,
                       const float* z_coarse, const float* weights, float* z_samples, float* z_fine, int* pos,
                       hipStream_t stream);
int launch_merge_raw(int64_t n, int nc, int nf, const float* raw_c, const float* raw_s, const int* pos, float* raw_f,
                     hipStream_t stream);
int launch_split_grad(int64_t n, int nc, int nf, const float* g_f, const int* pos, float* g_c, int accumulate_coarse,
                      float* g_s, hipStream_t stream);

}  // namespace mi
