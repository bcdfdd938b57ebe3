// eval_stages.hip - consumers/producers either side of the render path (gfx950), SURVEY.md 8(f) ranks 2-4.
//
//   image_metrics_kernel   mse + pytorch_ssim.ssim        reference nerf/test_nerf.py:102-105,
//                                                          nerf/pytorch_ssim/__init__.py:12-37
//   grid_points_kernel     voxel-grid query points        pi_GAN/utils.py:57-75 (create_mesh)
//   nerf_loss_kernel       training loss + its gradient   nerf/train_nerf.py:158-167
//   ray_bank_kernel        rays_rgba batching table       nerf/train_nerf.py:64-68, 78-82
//
// All are HBM-bound streaming kernels next to a frame's 1.9e14 FLOP; they exist so a rendered frame is scored, a
// density grid is sampled and a training batch is assembled and scored without leaving the device.  Compiled with -ffp-contract=off (un-fused mul/add
// like the torch ops they restate).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mi_common.h"

namespace mi {

constexpr int kMaxWindow = 31;
struct Window { float g[kMaxWindow + 1]; int size; };

constexpr int kTile = 32;                       // output pixels per workgroup side

// One workgroup = one 32x32 output tile of one (image, channel) plane.
// Stage 1: the (32+2r)^2 halo of both images into LDS (zero outside the plane = F.conv2d's zero padding).
// Stage 2: horizontal 1-D gaussian of the five planes a, b, a*a, b*b, a*b  -> hbuf[5][32+2r][32].
// Stage 3: vertical 1-D gaussian, the SSIM map value, block sums of ssim and (a-b)^2 -> partial[block][2].
// The reference convolves with the 2-D window g g^T (121 taps); the separable form differs from it by fp32
// rounding only (measured against the oracle in tests/test_gpu_metrics.py).
__global__ __launch_bounds__(256) void image_metrics_kernel(const float* __restrict__ img1,
                                                            const float* __restrict__ img2, int H, int W, Window win,
                                                            float* __restrict__ partial) {
    extern __shared__ float lds[];
    const int r = win.size / 2, span = kTile + 2 * r;
    float* a = lds;                         // [span][span]
    float* b = a + span * span;             // [span][span]
    float* hb = b + span * span;            // [5][span][kTile]
    __shared__ float red[2][4];
    const int64_t plane = blockIdx.z;
    const float* p1 = img1 + plane * H * W;
    const float* p2 = img2 + plane * H * W;
    const int x0 = blockIdx.x * kTile - r, y0 = blockIdx.y * kTile - r;
    for (int i = threadIdx.x; i < span * span; i += 256) {
        const int y = y0 + i / span, x = x0 + i % span;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        a[i] = in ? p1[(int64_t)y * W + x] : 0.f;
        b[i] = in ? p2[(int64_t)y * W + x] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < span * kTile; i += 256) {
        const int row = i / kTile, col = i % kTile;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
        for (int t = 0; t < win.size; ++t) {
            const float g = win.g[t], va = a[row * span + col + t], vb = b[row * span + col + t];
            s0 += g * va; s1 += g * vb; s2 += g * (va * va); s3 += g * (vb * vb); s4 += g * (va * vb);
        }
        hb[(0 * span + row) * kTile + col] = s0; hb[(1 * span + row) * kTile + col] = s1;
        hb[(2 * span + row) * kTile + col] = s2; hb[(3 * span + row) * kTile + col] = s3;
        hb[(4 * span + row) * kTile + col] = s4;
    }
    __syncthreads();
    float ssim_sum = 0.f, se_sum = 0.f;
    for (int i = threadIdx.x; i < kTile * kTile; i += 256) {
        const int row = i / kTile, col = i % kTile;
        const int y = blockIdx.y * kTile + row, x = blockIdx.x * kTile + col;
        if (y >= H || x >= W) continue;
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
        for (int t = 0; t < win.size; ++t) {
            const float g = win.g[t];
            mu1 += g * hb[(0 * span + row + t) * kTile + col]; mu2 += g * hb[(1 * span + row + t) * kTile + col];
            e11 += g * hb[(2 * span + row + t) * kTile + col]; e22 += g * hb[(3 * span + row + t) * kTile + col];
            e12 += g * hb[(4 * span + row + t) * kTile + col];
        }
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu1_mu2 = mu1 * mu2;
        const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu1_mu2;
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        ssim_sum += ((2.f * mu1_mu2 + C1) * (2.f * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2));
        const float d = a[(row + r) * span + col + r] - b[(row + r) * span + col + r];
        se_sum += d * d;
    }
    for (int off = 32; off > 0; off >>= 1) {
        ssim_sum += __shfl_xor(ssim_sum, off);
        se_sum += __shfl_xor(se_sum, off);
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ssim_sum; red[1][threadIdx.x >> 6] = se_sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int64_t blk = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[blk * 2 + 0] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        partial[blk * 2 + 1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    }
}

// out[n] = {mean squared error, mean ssim} of image n: fixed-order fp64 sum of its blocks' partials.
__global__ __launch_bounds__(256) void image_metrics_reduce_kernel(const float* __restrict__ partial,
                                                                   int64_t blocks_per_image, double inv_count,
                                                                   float* __restrict__ out) {
    __shared__ double red[2][256];
    const float* src = partial + (int64_t)blockIdx.x * blocks_per_image * 2;
    double s = 0.0, e = 0.0;
    for (int64_t i = threadIdx.x; i < blocks_per_image; i += 256) { s += src[i * 2]; e += src[i * 2 + 1]; }
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = e;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            red[0][threadIdx.x] += red[0][threadIdx.x + w];
            red[1][threadIdx.x] += red[1][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2 + 0] = (float)(red[1][0] * inv_count);
        out[blockIdx.x * 2 + 1] = (float)(red[0][0] * inv_count);
    }
}

int64_t image_metrics_workspace_floats(int images, int channels, int H, int W) {
    const int64_t bx = (W + kTile - 1) / kTile, by = (H + kTile - 1) / kTile;
    return (int64_t)images * channels * bx * by * 2;
}

int launch_image_metrics(const float* img1, const float* img2, int images, int channels, int H, int W,
                         const float* window, int window_size, float* workspace, float* out, hipStream_t stream) {
    Window win{};
    win.size = window_size;
    for (int i = 0; i < window_size; ++i) win.g[i] = window[i];
    const int span = kTile + 2 * (window_size / 2);
    const size_t lds = (size_t)(2 * span * span + 5 * span * kTile) * sizeof(float);
    if (lds > 64 * 1024) { set_error("image_metrics: window %d needs %zu B of LDS", window_size, lds); return -1; }
    const unsigned bx = (W + kTile - 1) / kTile, by = (H + kTile - 1) / kTile;
    const int64_t planes = (int64_t)images * channels;
    if (planes > 65535) { set_error("image_metrics: too many planes (%lld)", (long long)planes); return -1; }
    hipLaunchKernelGGL(image_metrics_kernel, dim3(bx, by, (unsigned)planes), dim3(256), lds, stream, img1, img2, H, W, win,
                       workspace);
    hipLaunchKernelGGL(image_metrics_reduce_kernel, dim3(images), dim3(256), 0, stream, workspace,
                       (int64_t)channels * bx * by, 1.0 / ((double)channels * H * W), out);
    return check_launch("image_metrics");
}

// create_mesh's sample grid (pi_GAN/utils.py:57-75): overall index i -> (i / N^2 % N, i / N % N, i % N) scaled by
// voxel_size and shifted by voxel_origin[2], [1], [0] respectively (the reference's axis order), view dir = 0.
__global__ void grid_points_kernel(int N, float o0, float o1, float o2, float voxel_size, int64_t head, int64_t count,
                                   float* __restrict__ pts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int64_t idx = head + i;
    const float fx = (float)(idx / N / N % N), fy = (float)(idx / N % N), fz = (float)(idx % N);
    float* p = pts + i * 6;
    p[0] = fx * voxel_size + o2;
    p[1] = fy * voxel_size + o1;
    p[2] = fz * voxel_size + o0;
    p[3] = 0.f; p[4] = 0.f; p[5] = 0.f;
}

int launch_grid_points(int N, const float* origin, float voxel_size, int64_t head, int64_t count, float* pts,
                       hipStream_t stream) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL(grid_points_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, N, origin[0],
                       origin[1], origin[2], voxel_size, head, count, pts);
    return check_launch("grid_points");
}

// ---------------------------------------------------------------------------------------
// nerf/train_nerf.py:158-167: loss_x = mean((rgb_x - rgb)^2) [+ 0.1 mean((acc_x - alpha)^2) if use_alpha],
// loss = loss_fine [+ loss_coarse if use_fine_model]; psnr = -10 log10(mean((rgb_fine - rgb)^2)).
// One pass writes the four gradient seeds d loss / d (rgb_c, acc_c, rgb_f, acc_f) and per-block partial sums
// {se_rgb_c, se_acc_c, se_rgb_f, se_acc_f}; a one-block kernel adds them in fixed order (fp64) into
// out = {loss, mse_rgb_fine, loss_coarse, loss_fine}.  target [n,4] = (r, g, b, alpha), as batch[:, -4:].
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nerf_loss_kernel(int64_t n, const float* __restrict__ rgb_c,
                                                        const float* __restrict__ acc_c, const float* __restrict__ rgb_f,
                                                        const float* __restrict__ acc_f, const float* __restrict__ target,
                                                        int use_alpha, int use_fine, float* __restrict__ g_rgb_c,
                                                        float* __restrict__ g_acc_c, float* __restrict__ g_rgb_f,
                                                        float* __restrict__ g_acc_f, float* __restrict__ partial) {
    __shared__ float red[4][4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < n) {
        const float w_rgb = 2.f / (3.f * (float)n), w_acc = use_alpha ? 0.1f * 2.f / (float)n : 0.f;
        const float wc = use_fine ? 1.f : 0.f;       // the coarse loss only counts with a separate fine model
        const float a = target[i * 4 + 3];
        const float dac = acc_c[i] - a, daf = acc_f[i] - a;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float t = target[i * 4 + k];
            const float dc = rgb_c[i * 3 + k] - t, df = rgb_f[i * 3 + k] - t;
            s[0] += dc * dc; s[2] += df * df;
            g_rgb_c[i * 3 + k] = wc * w_rgb * dc;
            g_rgb_f[i * 3 + k] = w_rgb * df;
        }
        s[1] = dac * dac; s[3] = daf * daf;
        g_acc_c[i] = wc * w_acc * dac;
        g_acc_f[i] = w_acc * daf;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_xor(s[k], off);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s[k];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        partial[(int64_t)blockIdx.x * 4 + k] = ((red[k][0] + red[k][1]) + red[k][2]) + red[k][3];
    }
}

__global__ __launch_bounds__(256) void nerf_loss_reduce_kernel(const float* __restrict__ partial, int64_t blocks, int64_t n,
                                                               int use_alpha, int use_fine, float* __restrict__ out) {
    __shared__ double red[4][256];
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t b = threadIdx.x; b < blocks; b += 256)
        for (int k = 0; k < 4; ++k) s[k] += partial[b * 4 + k];
    for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = s[k];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mc = red[0][0] / (3.0 * n), mf = red[2][0] / (3.0 * n);
        const double lc = mc + (use_alpha ? 0.1 * red[1][0] / n : 0.0), lf = mf + (use_alpha ? 0.1 * red[3][0] / n : 0.0);
        out[0] = (float)(lf + (use_fine ? lc : 0.0));
        out[1] = (float)mf;
        out[2] = (float)lc;
        out[3] = (float)lf;
    }
}

int64_t nerf_loss_workspace_floats(int64_t n) { return ((n + 255) / 256) * 4; }

int launch_nerf_loss(int64_t n, const float* rgb_c, const float* acc_c, const float* rgb_f, const float* acc_f,
                     const float* target, int use_alpha, int use_fine, float* g_rgb_c, float* g_acc_c, float* g_rgb_f,
                     float* g_acc_f, float* workspace, float* out, hipStream_t stream) {
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(nerf_loss_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, n, rgb_c, acc_c, rgb_f, acc_f, target,
                       use_alpha, use_fine, g_rgb_c, g_acc_c, g_rgb_f, g_acc_f, workspace);
    hipLaunchKernelGGL(nerf_loss_reduce_kernel, dim3(1), dim3(256), 0, stream, workspace, blocks, n, use_alpha, use_fine, out);
    return check_launch("nerf_loss");
}

// ---------------------------------------------------------------------------------------
// The batching table of nerf/train_nerf.py:78-82: row r = image*H*W + pixel holds (rays_o, rays_d, r, g, b, a)
// with the rays of get_rays(width, height, focal, pose[image]) (all-fp32 arithmetic: focal is a Python float
// there) and, as train_nerf.py:64-68 does for the training set, rgb composited on white: rgb*a + (1 - a).
// poses [images][12] = c2w[:3,:4] row-major, rgba [images][H][W][4].  out [images*H*W][10].
// ---------------------------------------------------------------------------------------
// T = float: `focal` is a Python float (all-fp32 NumPy math); T = double: `focal` is an np.float64 scalar - what
// nerf/data_loader.py:151 returns and train_nerf.py:78 passes - so NumPy >= 2 evaluates get_rays in fp64 and the
// script rounds to fp32 afterwards (train_nerf.py:84); same switch as gen_rays_kernel (render_stages.hip).
template <class T>
__global__ __launch_bounds__(256) void ray_bank_kernel(int width, int height, T focal, const float* __restrict__ poses,
                                                       const float* __restrict__ rgba, int white_bkgd, int64_t n,
                                                       float* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int64_t hw = (int64_t)width * height;
    const int64_t img = r / hw, pix = r % hw;
    const float* m = poses + img * 12;
    const T half_w = (T)width / (T)2, half_h = (T)height / (T)2;                 // get_rays: width / 2 in Python floats
    const T x = ((T)(float)(pix % width) - half_w) / focal;
    const T y = -((T)(float)(pix / width) - half_h) / focal;
    float* o = out + r * 10;
    o[0] = m[3]; o[1] = m[7]; o[2] = m[11];
#pragma unroll
    for (int c = 0; c < 3; ++c) o[3 + c] = (float)((x * (T)m[4 * c + 0] + y * (T)m[4 * c + 1]) + (T)-1 * (T)m[4 * c + 2]);
    const float* px = rgba + r * 4;
    const float a = px[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) o[6 + c] = white_bkgd ? px[c] * a + (1.f - a) : px[c];
    o[9] = a;
}

int launch_ray_bank(int width, int height, double focal, const float* poses, const float* rgba, int white_bkgd,
                    int64_t images, float* out, int compute_f64, hipStream_t stream) {
    const int64_t n = images * width * height;
    if (n <= 0) return 0;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (compute_f64)
        hipLaunchKernelGGL(ray_bank_kernel<double>, grid, block, 0, stream, width, height, focal, poses, rgba, white_bkgd, n, out);
    else
        hipLaunchKernelGGL(ray_bank_kernel<float>, grid, block, 0, stream, width, height, (float)focal, poses, rgba, white_bkgd, n, out);
    return check_launch("ray_bank");
}

}  // namespace mi
