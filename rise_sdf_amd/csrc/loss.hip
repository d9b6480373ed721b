// N1: the loss tail on the path's outputs (systems/split_occ.py:163-215, systems/criterions.py:155-159) as two
// reduction kernels forward and two elementwise kernels backward, instead of ~40 masked-select / elementwise /
// reduce launches over [N,3] and [S,.] tensors:
//   per ray     masked RGB MSE + L1 on comp_rgb_full (and comp_rgb_phys_full at stage 1), mask BCE and "opaque"
//               BCE on opacity clamped to [1e-3, 1 - 1e-3]
//   per sample  eikonal (||grad|| - 1)^2, sparsity exp(-scale |sdf|), curvature |laplace|
// Forward writes raw sums (fp64, block tree + one atomic per block); the host-side mirror turns them into the
// reference's means with the lambda weights on device (no host read).  Backward reads its six coefficients
// (lambda_k * dL / count_k) from a device array, so the whole tail stays asynchronous.
#include "common.h"

namespace {

constexpr int THREADS = 256;

template <int K>
__device__ __forceinline__ void block_add(double (&v)[K], double *__restrict__ sums)
{
    __shared__ double part[THREADS / 64][K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double a = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][k] = a;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        double t = 0.0;
        for (int w = 0; w < THREADS / 64; ++w) t += part[w][threadIdx.x];
        atomicAdd(sums + threadIdx.x, t);
    }
}

__device__ __forceinline__ float clamp_op(float o) { return fminf(fmaxf(o, 1e-3f), 1.0f - 1e-3f); }

// sums: 0 sum d^2, 1 sum |d|, 2 sum dp^2, 3 sum |dp|, 4 valid rays, 5 mask BCE terms, 6 opaque BCE terms
__global__ void __launch_bounds__(THREADS)
loss_rays_fwd_kernel(const float *__restrict__ rgb, const float *__restrict__ rgb_phys,
                     const float *__restrict__ target, const uint8_t *__restrict__ valid,
                     const float *__restrict__ opacity, const float *__restrict__ fg_mask, int64_t n,
                     double *__restrict__ sums)
{
    double v[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS) {
        if (valid[i]) {
            v[4] += 1.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float d = rgb[3 * i + c] - target[3 * i + c];
                v[0] += (double)(d * d);
                v[1] += (double)fabsf(d);
                if (rgb_phys != nullptr) {
                    const float dp = rgb_phys[3 * i + c] - target[3 * i + c];
                    v[2] += (double)(dp * dp);
                    v[3] += (double)fabsf(dp);
                }
            }
        }
        const float o = clamp_op(opacity[i]);
        const float lo = logf(o), l1o = logf(1.0f - o);
        if (fg_mask != nullptr) {
            const float t = fg_mask[i];
            v[5] += (double)(-(t * lo + (1.0f - t) * l1o));
        }
        v[6] += (double)(-(o * lo + (1.0f - o) * l1o));
    }
    block_add<7>(v, sums);
}

// coef: 0 mse, 1 l1, 2 phys mse, 3 phys l1 (each already / (3 valid)), 4 mask / N, 5 opaque / N
__global__ void __launch_bounds__(THREADS)
loss_rays_bwd_kernel(const float *__restrict__ rgb, const float *__restrict__ rgb_phys,
                     const float *__restrict__ target, const uint8_t *__restrict__ valid,
                     const float *__restrict__ opacity, const float *__restrict__ fg_mask,
                     const float *__restrict__ coef, int64_t n, float *__restrict__ d_rgb,
                     float *__restrict__ d_rgb_phys, float *__restrict__ d_opacity)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const bool ok = valid[i] != 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float t = target[3 * i + c];
        const float d = rgb[3 * i + c] - t;
        const float sg = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
        d_rgb[3 * i + c] = ok ? 2.0f * d * coef[0] + sg * coef[1] : 0.0f;
        if (d_rgb_phys != nullptr) {
            const float dp = rgb_phys[3 * i + c] - t;
            const float sp = dp > 0.0f ? 1.0f : (dp < 0.0f ? -1.0f : 0.0f);
            d_rgb_phys[3 * i + c] = ok ? 2.0f * dp * coef[2] + sp * coef[3] : 0.0f;
        }
    }
    const float raw = opacity[i];
    float g = 0.0f;
    if (raw >= 1e-3f && raw <= 1.0f - 1e-3f) {   // torch.clamp passes the gradient inside [min, max]
        const float o = raw;
        if (fg_mask != nullptr) {
            const float t = fg_mask[i];
            g += coef[4] * (-(t / o - (1.0f - t) / (1.0f - o)));
        }
        g += coef[5] * (logf(1.0f - o) - logf(o));   // d/do -(o log o + (1-o) log(1-o))
    }
    d_opacity[i] = g;
}

// sums: 0 eikonal, 1 sparsity, 2 curvature
__global__ void __launch_bounds__(THREADS)
loss_samples_fwd_kernel(const float *__restrict__ sdf, const float *__restrict__ grad,
                        const float *__restrict__ laplace, float sparsity_scale, int64_t n,
                        double *__restrict__ sums)
{
    double v[3] = {0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS) {
        const float gx = grad[3 * i], gy = grad[3 * i + 1], gz = grad[3 * i + 2];
        const float e = sqrtf(gx * gx + gy * gy + gz * gz) - 1.0f;
        v[0] += (double)(e * e);
        v[1] += (double)expf(-sparsity_scale * fabsf(sdf[i]));
        if (laplace != nullptr) v[2] += (double)fabsf(laplace[i]);
    }
    block_add<3>(v, sums);
}

// coef: 0 eikonal / S, 1 sparsity / S, 2 curvature / S
__global__ void __launch_bounds__(THREADS)
loss_samples_bwd_kernel(const float *__restrict__ sdf, const float *__restrict__ grad,
                        const float *__restrict__ laplace, float sparsity_scale,
                        const float *__restrict__ coef, int64_t n, float *__restrict__ d_sdf,
                        float *__restrict__ d_grad, float *__restrict__ d_laplace)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const float gx = grad[3 * i], gy = grad[3 * i + 1], gz = grad[3 * i + 2];
    const float nrm = sqrtf(gx * gx + gy * gy + gz * gz);
    const float k = nrm > 0.0f ? coef[0] * 2.0f * (nrm - 1.0f) / nrm : 0.0f;
    d_grad[3 * i] = k * gx; d_grad[3 * i + 1] = k * gy; d_grad[3 * i + 2] = k * gz;
    const float s = sdf[i];
    const float sg = s > 0.0f ? 1.0f : (s < 0.0f ? -1.0f : 0.0f);
    d_sdf[i] = -coef[1] * sparsity_scale * sg * expf(-sparsity_scale * fabsf(s));
    if (d_laplace != nullptr) {
        const float l = laplace[i];
        d_laplace[i] = coef[2] * (l > 0.0f ? 1.0f : (l < 0.0f ? -1.0f : 0.0f));
    }
}

unsigned red_blocks(int64_t n) { const int64_t b = (n + THREADS * 4 - 1) / (THREADS * 4); return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b)); }

}  // namespace

extern "C" {

int rsdf_loss_rays_fwd(const float *comp_rgb, const float *comp_rgb_phys, const float *target,
                       const uint8_t *rays_valid, const float *opacity, const float *fg_mask, int64_t n,
                       double *sums7, void *stream)
{
    if (n <= 0) return 0;
    loss_rays_fwd_kernel<<<red_blocks(n), THREADS, 0, (hipStream_t)stream>>>(comp_rgb, comp_rgb_phys, target,
                                                                           rays_valid, opacity, fg_mask, n, sums7);
    RSDF_RETURN_LAUNCH();
}

int rsdf_loss_rays_bwd(const float *comp_rgb, const float *comp_rgb_phys, const float *target,
                       const uint8_t *rays_valid, const float *opacity, const float *fg_mask, const float *coef6,
                       int64_t n, float *d_comp_rgb, float *d_comp_rgb_phys, float *d_opacity, void *stream)
{
    if (n <= 0) return 0;
    loss_rays_bwd_kernel<<<rsdf_blocks(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        comp_rgb, comp_rgb_phys, target, rays_valid, opacity, fg_mask, coef6, n, d_comp_rgb, d_comp_rgb_phys,
        d_opacity);
    RSDF_RETURN_LAUNCH();
}

int rsdf_loss_samples_fwd(const float *sdf, const float *sdf_grad, const float *laplace, float sparsity_scale,
                          int64_t n, double *sums3, void *stream)
{
    if (n <= 0) return 0;
    loss_samples_fwd_kernel<<<red_blocks(n), THREADS, 0, (hipStream_t)stream>>>(sdf, sdf_grad, laplace,
                                                                              sparsity_scale, n, sums3);
    RSDF_RETURN_LAUNCH();
}

int rsdf_loss_samples_bwd(const float *sdf, const float *sdf_grad, const float *laplace, float sparsity_scale,
                          const float *coef3, int64_t n, float *d_sdf, float *d_sdf_grad, float *d_laplace,
                          void *stream)
{
    if (n <= 0) return 0;
    loss_samples_bwd_kernel<<<rsdf_blocks(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        sdf, sdf_grad, laplace, sparsity_scale, coef3, n, d_sdf, d_sdf_grad, d_laplace);
    RSDF_RETURN_LAUNCH();
}

}  // extern "C"
