// P1 / H4 / A1: sample positions + finite-difference taps, FD gradient assembly, NeuS alpha.
//
// Specification followed (paths relative to the upstream RISE-SDF tree):
//   models/split_mixed_occ.py:229-231   positions = o[ri] + d[ri] * (t0 + t1)[...,None] / 2
//   models/geometry.py:229-244          six taps x +- eps e_k, clamp(-r, r), contraction, 0.5*(f+ - f-)/eps
//   models/geometry.py:17-19, models/utils.py:109-114   AABB contraction (x - (-r)) / (r - (-r))
//   models/split_mixed_occ.py:237       normal = F.normalize(grad, p=2, dim=-1, eps=1e-6)
//   models/split_mixed_occ.py:151-177   get_alpha (== models/neus.py:128-150)
//   models/split_mixed_occ.py:21-56     VarianceNetwork: inv_s = exp(10 v), clipped to [1e-6, 1e6]
// The reference runs this as ~40 elementwise torch kernels; here it is one kernel each way.
#include "common.h"

namespace {

constexpr int THREADS = 256;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void __launch_bounds__(THREADS)
fd_points_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                 const int64_t *__restrict__ ri, const float *__restrict__ ts,
                 const float *__restrict__ te, int64_t n, float radius, float eps,
                 float *__restrict__ xu, float *__restrict__ positions, int tap_major)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const int64_t r = ri[i];
    // (t0 + t1) / 2 as the reference writes it: the sum first, then d * sum, then / 2
    const float tsum = ts[i] + te[i];
    float p[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = rays_o[3 * r + k] + rays_d[3 * r + k] * tsum / 2.0f;
    if (positions) {
#pragma unroll
        for (int k = 0; k < 3; ++k) positions[3 * i + k] = p[k];
    }
    if (xu == nullptr) return;       // positions only: the x2 stencil kernels derive the taps themselves (hashgrid_fd7.hip)
    const float two_r = radius - (-radius);
    // interleaved [n][7][3] or tap-major [7][n][3]
    const int64_t tstride = tap_major ? n * 3 : 3;
    float *o = xu + (tap_major ? i * 3 : i * 21);
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = (p[k] - (-radius)) / two_r;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int axis = t >> 1;
        const float off = (t & 1) ? -eps : eps;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float q = p[k] + (k == axis ? off : 0.0f);
            q = fminf(fmaxf(q, -radius), radius);
            o[(t + 1) * tstride + k] = (q - (-radius)) / two_r;
        }
    }
}

// taps from explicit world-space points (the VolumeSDF.forward(points) signature, geometry.py:206)
__global__ void __launch_bounds__(THREADS)
fd_taps_kernel(const float *__restrict__ points, int64_t n, float radius, float eps,
               float *__restrict__ xu)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const float p[3] = {points[3 * i], points[3 * i + 1], points[3 * i + 2]};
    const float two_r = radius - (-radius);
    float *o = xu + i * 21;
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = (p[k] - (-radius)) / two_r;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int axis = t >> 1;
        const float off = (t & 1) ? -eps : eps;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float q = p[k] + (k == axis ? off : 0.0f);
            q = fminf(fmaxf(q, -radius), radius);
            o[3 + 3 * t + k] = (q - (-radius)) / two_r;
        }
    }
}

// grad = 0.5 * (f+ - f-) / eps   (geometry.py:243)
__global__ void __launch_bounds__(THREADS)
fd_gradient_fwd_kernel(const float *__restrict__ sdf7, int ld, float eps, int64_t n,
                       float *__restrict__ sdf, float *__restrict__ grad)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const int64_t tstr = ld < 0 ? n : (int64_t)ld, ss = ld < 0 ? 1 : 7 * (int64_t)ld;
    const float *s = sdf7 + i * ss;
    if (sdf) sdf[i] = s[0];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        grad[3 * i + k] = 0.5f * (s[(1 + 2 * k) * tstr] - s[(2 + 2 * k) * tstr]) / eps;
}

__global__ void __launch_bounds__(THREADS)
fd_gradient_bwd_kernel(const float *__restrict__ d_sdf, const float *__restrict__ d_grad, float eps,
                       int64_t n, float *__restrict__ d_sdf7, int ld)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const int64_t tstr = ld < 0 ? n : (int64_t)ld, ss = ld < 0 ? 1 : 7 * (int64_t)ld;
    float *o = d_sdf7 + i * ss;
    o[0] = d_sdf ? d_sdf[i] : 0.0f;
    const float c = 0.5f / eps;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float g = d_grad ? d_grad[3 * i + k] : 0.0f;
        o[(1 + 2 * k) * tstr] = c * g;
        o[(2 + 2 * k) * tstr] = -c * g;
    }
}

struct AlphaTerms {
    float inv_s, cosv, u, v, iter_cos, half, e_prev, e_next, pc, nc, q, denom;
};

__device__ __forceinline__ float inv_s_from(const float *variance)
{
    return fminf(fmaxf(expf(variance[0] * 10.0f), 1e-6f), 1e6f);
}

__device__ __forceinline__ float alpha_core(float sdf, float cosv, float dist, float inv_s, float r,
                                            AlphaTerms &t)
{
    t.inv_s = inv_s;
    t.cosv = cosv;
    t.u = -cosv * 0.5f + 0.5f;
    t.v = -cosv;
    t.iter_cos = -(fmaxf(t.u, 0.0f) * (1.0f - r) + fmaxf(t.v, 0.0f) * r);
    t.half = t.iter_cos * dist * 0.5f;
    t.e_next = sdf + t.half;
    t.e_prev = sdf - t.half;
    t.pc = sigmoidf_(t.e_prev * inv_s);
    t.nc = sigmoidf_(t.e_next * inv_s);
    const float p = t.pc - t.nc;
    t.denom = t.pc + 1e-5f;
    t.q = (p + 1e-5f) / t.denom;
    return fminf(fmaxf(t.q, 0.0f), 1.0f);
}

// d_alpha -> (d_sdf, d_cos, d_inv_s)
__device__ __forceinline__ void alpha_core_bwd(const AlphaTerms &t, float dist, float r, float d_alpha,
                                               float &d_sdf, float &d_cos, float &d_inv_s)
{
    const float dq = (t.q >= 0.0f && t.q <= 1.0f) ? d_alpha : 0.0f;
    const float dp = dq / t.denom;
    const float dc = -dq * t.q / t.denom;
    const float d_pc = dp + dc, d_nc = -dp;
    const float A = d_pc * t.pc * (1.0f - t.pc);
    const float B = d_nc * t.nc * (1.0f - t.nc);
    const float d_eprev = A * t.inv_s, d_enext = B * t.inv_s;
    d_inv_s = A * t.e_prev + B * t.e_next;
    d_sdf = d_eprev + d_enext;
    const float d_half = d_enext - d_eprev;
    const float d_iter = d_half * dist * 0.5f;
    // iter_cos = -(relu(u)(1-r) + relu(v) r), u = -cos/2 + 1/2, v = -cos
    d_cos = d_iter * ((t.u > 0.0f ? 0.5f * (1.0f - r) : 0.0f) + (t.v > 0.0f ? r : 0.0f));
}

__global__ void __launch_bounds__(THREADS)
alpha_fd_fwd_kernel(const float *__restrict__ sdf7, int ld, const float *__restrict__ rays_d,
                    const int64_t *__restrict__ ri, const float *__restrict__ ts,
                    const float *__restrict__ te, const float *__restrict__ variance, float r,
                    float eps, int64_t n, float *__restrict__ sdf_o, float *__restrict__ grad_o,
                    float *__restrict__ normal_o, float *__restrict__ alpha_o)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const int64_t tstr = ld < 0 ? n : (int64_t)ld, ss = ld < 0 ? 1 : 7 * (int64_t)ld;
    const float *s = sdf7 + i * ss;
    const float sdf = s[0];
    float g[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) g[k] = 0.5f * (s[(1 + 2 * k) * tstr] - s[(2 + 2 * k) * tstr]) / eps;
    const float nrm = fmaxf(sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]), 1e-6f);
    const float nx = g[0] / nrm, ny = g[1] / nrm, nz = g[2] / nrm;
    const int64_t ray = ri[i];
    const float cosv = rays_d[3 * ray] * nx + rays_d[3 * ray + 1] * ny + rays_d[3 * ray + 2] * nz;
    AlphaTerms t;
    const float a = alpha_core(sdf, cosv, te[i] - ts[i], inv_s_from(variance), r, t);
    if (sdf_o) sdf_o[i] = sdf;
    if (grad_o) { grad_o[3 * i] = g[0]; grad_o[3 * i + 1] = g[1]; grad_o[3 * i + 2] = g[2]; }
    if (normal_o) { normal_o[3 * i] = nx; normal_o[3 * i + 1] = ny; normal_o[3 * i + 2] = nz; }
    alpha_o[i] = a;
}

__global__ void __launch_bounds__(THREADS)
alpha_fd_bwd_kernel(const float *__restrict__ sdf7, int ld, const float *__restrict__ rays_d,
                    const int64_t *__restrict__ ri, const float *__restrict__ ts,
                    const float *__restrict__ te, const float *__restrict__ variance, float r,
                    float eps, int64_t n, const float *__restrict__ d_alpha,
                    const float *__restrict__ d_normal, const float *__restrict__ d_sdf_in,
                    const float *__restrict__ d_grad_in, float *__restrict__ d_sdf7, int ld_out,
                    float *__restrict__ d_variance)
{
    // grid-stride: a workgroup walks many tiles and ends with ONE atomic on d_variance.  With a workgroup per tile the
    // 74 k same-address float atomics of an 18.9 M-sample chunk serialise at the memory side and were the kernel's
    // whole duration (1.0 ms against the forward's 0.23 for the same bytes).
    float d_inv_s = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS) {
        const int64_t tstr = ld < 0 ? n : (int64_t)ld, ss = ld < 0 ? 1 : 7 * (int64_t)ld;
        const float *s = sdf7 + i * ss;
        const float sdf = s[0];
        float g[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            g[k] = 0.5f * (s[(1 + 2 * k) * tstr] - s[(2 + 2 * k) * tstr]) / eps;
        const float len = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
        const float nrm = fmaxf(len, 1e-6f);
        const float nv[3] = {g[0] / nrm, g[1] / nrm, g[2] / nrm};
        const int64_t ray = ri[i];
        const float dv[3] = {rays_d[3 * ray], rays_d[3 * ray + 1], rays_d[3 * ray + 2]};
        const float cosv = dv[0] * nv[0] + dv[1] * nv[1] + dv[2] * nv[2];
        const float dist = te[i] - ts[i];
        AlphaTerms t;
        alpha_core(sdf, cosv, dist, inv_s_from(variance), r, t);
        float d_sdf = 0.0f, d_cos = 0.0f, d_is = 0.0f;
        if (d_alpha) alpha_core_bwd(t, dist, r, d_alpha[i], d_sdf, d_cos, d_is);
        d_inv_s += d_is;
        if (d_sdf_in) d_sdf += d_sdf_in[i];
        float dn[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) dn[k] = d_cos * dv[k] + (d_normal ? d_normal[3 * i + k] : 0.0f);
        // n = g / max(|g|, 1e-6)
        float dg[3];
        if (len > 1e-6f) {
            const float ndn = nv[0] * dn[0] + nv[1] * dn[1] + nv[2] * dn[2];
#pragma unroll
            for (int k = 0; k < 3; ++k) dg[k] = (dn[k] - nv[k] * ndn) / len;
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) dg[k] = dn[k] / 1e-6f;
        }
        if (d_grad_in) {
#pragma unroll
            for (int k = 0; k < 3; ++k) dg[k] += d_grad_in[3 * i + k];
        }
        const int64_t tso = ld_out < 0 ? n : (int64_t)ld_out, sso = ld_out < 0 ? 1 : 7 * (int64_t)ld_out;
        float *o = d_sdf7 + i * sso;
        o[0] = d_sdf;
        const float c = 0.5f / eps;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o[(1 + 2 * k) * tso] = c * dg[k];
            o[(2 + 2 * k) * tso] = -c * dg[k];
        }
    }
    if (d_variance) {
        // d inv_s / d v = 10 inv_s inside the clip range
        const float e = expf(variance[0] * 10.0f);
        const float scale = (e >= 1e-6f && e <= 1e6f) ? 10.0f * e : 0.0f;
        float part = wave_sum(d_inv_s * scale);
        __shared__ float red[THREADS / 64];
        if (lane_id() == 0) red[threadIdx.x >> 6] = part;
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = 0.0f;
#pragma unroll
            for (int k = 0; k < THREADS / 64; ++k) tot += red[k];
            atomicAdd(d_variance, tot);
        }
    }
}

__global__ void __launch_bounds__(THREADS)
alpha_fwd_kernel(const float *__restrict__ sdf, const float *__restrict__ normal,
                 const float *__restrict__ dirs, const float *__restrict__ dists,
                 const float *__restrict__ variance, float r, int64_t n, float *__restrict__ alpha)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const float cosv = dirs[3 * i] * normal[3 * i] + dirs[3 * i + 1] * normal[3 * i + 1] +
                       dirs[3 * i + 2] * normal[3 * i + 2];
    AlphaTerms t;
    alpha[i] = alpha_core(sdf[i], cosv, dists[i], inv_s_from(variance), r, t);
}

// A2: occ_eval_fn (models/split_mixed_occ.py:108-119, models/neus.py:101-111): the same formula with cos == -1 and
// dist == render_step_size, on the SDF of the occupancy grid's cell points (no gradient: the update runs under no_grad)
__global__ void __launch_bounds__(THREADS)
occ_alpha_kernel(const float *__restrict__ sdf, const float *__restrict__ variance, float step, int64_t n,
                 float *__restrict__ alpha)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    AlphaTerms t;
    alpha[i] = alpha_core(sdf[i], -1.0f, step, inv_s_from(variance), 1.0f, t);
}

__global__ void __launch_bounds__(THREADS)
alpha_bwd_kernel(const float *__restrict__ sdf, const float *__restrict__ normal,
                 const float *__restrict__ dirs, const float *__restrict__ dists,
                 const float *__restrict__ variance, float r, int64_t n,
                 const float *__restrict__ d_alpha, float *__restrict__ d_sdf,
                 float *__restrict__ d_normal, float *__restrict__ d_variance)
{
    float d_inv_s = 0.0f;   // grid-stride, one d_variance atomic per workgroup (see alpha_fd_bwd_kernel)
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS) {
        const float dv[3] = {dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]};
        const float cosv = dv[0] * normal[3 * i] + dv[1] * normal[3 * i + 1] + dv[2] * normal[3 * i + 2];
        AlphaTerms t;
        alpha_core(sdf[i], cosv, dists[i], inv_s_from(variance), r, t);
        float ds, dc, d_is;
        alpha_core_bwd(t, dists[i], r, d_alpha[i], ds, dc, d_is);
        d_inv_s += d_is;
        if (d_sdf) d_sdf[i] = ds;
        if (d_normal) {
#pragma unroll
            for (int k = 0; k < 3; ++k) d_normal[3 * i + k] = dc * dv[k];
        }
    }
    if (d_variance) {
        const float e = expf(variance[0] * 10.0f);
        const float scale = (e >= 1e-6f && e <= 1e6f) ? 10.0f * e : 0.0f;
        float part = wave_sum(d_inv_s * scale);
        __shared__ float red[THREADS / 64];
        if (lane_id() == 0) red[threadIdx.x >> 6] = part;
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = 0.0f;
#pragma unroll
            for (int k = 0; k < THREADS / 64; ++k) tot += red[k];
            atomicAdd(d_variance, tot);
        }
    }
}

}  // namespace

extern "C" {

int rsdf_fd_points(const float *rays_o, const float *rays_d, const int64_t *ray_indices,
                   const float *t_starts, const float *t_ends, int64_t n, float radius, float eps,
                   float *x_unit, float *positions, int tap_major, void *stream)
{
    RSDF_CHECK_ARG(radius > 0.f, "fd_points: radius must be > 0");
    RSDF_CHECK_ARG(x_unit != nullptr || positions != nullptr, "fd_points: x_unit and positions are both NULL");
    if (n <= 0) return 0;
    fd_points_kernel<<<rsdf_blocks(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        rays_o, rays_d, ray_indices, t_starts, t_ends, n, radius, eps, x_unit, positions, tap_major);
    RSDF_RETURN_LAUNCH();
}

int rsdf_fd_taps(const float *points, int64_t n, float radius, float eps, float *x_unit, void *stream)
{
    RSDF_CHECK_ARG(radius > 0.f, "fd_taps: radius must be > 0");
    if (n <= 0) return 0;
    fd_taps_kernel<<<rsdf_blocks(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(points, n, radius, eps,
                                                                              x_unit);
    RSDF_RETURN_LAUNCH();
}

int rsdf_fd_gradient_fwd(const float *sdf7, int ld, float eps, int64_t n, float *sdf, float *grad,
                         void *stream)
{
    RSDF_CHECK_ARG(eps > 0.f && ld != 0, "fd_gradient_fwd: bad eps or ld");
    if (n <= 0) return 0;
    fd_gradient_fwd_kernel<<<rsdf_blocks(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(sdf7, ld, eps, n,
                                                                                      sdf, grad);
    RSDF_RETURN_LAUNCH();
}

int rsdf_fd_gradient_bwd(const float *d_sdf, const float *d_grad, float eps, int64_t n, float *d_sdf7,
                         int ld, void *stream)
{
    RSDF_CHECK_ARG(eps > 0.f && ld != 0, "fd_gradient_bwd: bad eps or ld");
    if (n <= 0) return 0;
    fd_gradient_bwd_kernel<<<rsdf_blocks(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(d_sdf, d_grad, eps,
                                                                                      n, d_sdf7, ld);
    RSDF_RETURN_LAUNCH();
}

int rsdf_neus_alpha_fd_fwd(const float *sdf7, int ld, const float *rays_d, const int64_t *ray_indices,
                           const float *t_starts, const float *t_ends, const float *variance,
                           float cos_anneal_ratio, float eps, int64_t n, float *sdf, float *grad,
                           float *normal, float *alpha, void *stream)
{
    RSDF_CHECK_ARG(eps > 0.f && ld != 0, "neus_alpha_fd_fwd: bad eps or ld");
    if (n <= 0) return 0;
    alpha_fd_fwd_kernel<<<rsdf_blocks(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        sdf7, ld, rays_d, ray_indices, t_starts, t_ends, variance, cos_anneal_ratio, eps, n, sdf, grad,
        normal, alpha);
    RSDF_RETURN_LAUNCH();
}

int rsdf_neus_alpha_fd_bwd(const float *sdf7, int ld, const float *rays_d, const int64_t *ray_indices,
                           const float *t_starts, const float *t_ends, const float *variance,
                           float cos_anneal_ratio, float eps, int64_t n, const float *d_alpha,
                           const float *d_normal, const float *d_sdf, const float *d_grad,
                           float *d_sdf7, int ld_out, float *d_variance, void *stream)
{
    RSDF_CHECK_ARG(eps > 0.f && ld != 0 && ld_out != 0, "neus_alpha_fd_bwd: bad eps or ld");
    if (n <= 0) return 0;
    const unsigned tiles = rsdf_blocks(n, THREADS);
    alpha_fd_bwd_kernel<<<tiles < 4096u ? tiles : 4096u, THREADS, 0, (hipStream_t)stream>>>(
        sdf7, ld, rays_d, ray_indices, t_starts, t_ends, variance, cos_anneal_ratio, eps, n, d_alpha,
        d_normal, d_sdf, d_grad, d_sdf7, ld_out, d_variance);
    RSDF_RETURN_LAUNCH();
}

int rsdf_neus_alpha_fwd(const float *sdf, const float *normal, const float *dirs, const float *dists,
                        const float *variance, float cos_anneal_ratio, int64_t n, float *alpha,
                        void *stream)
{
    if (n <= 0) return 0;
    alpha_fwd_kernel<<<rsdf_blocks(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        sdf, normal, dirs, dists, variance, cos_anneal_ratio, n, alpha);
    RSDF_RETURN_LAUNCH();
}

int rsdf_neus_occ_alpha(const float *sdf, const float *variance, float render_step_size, int64_t n, float *alpha,
                        void *stream)
{
    if (n <= 0) return 0;
    occ_alpha_kernel<<<rsdf_blocks(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(sdf, variance, render_step_size, n,
                                                                                 alpha);
    RSDF_RETURN_LAUNCH();
}

int rsdf_neus_alpha_bwd(const float *sdf, const float *normal, const float *dirs, const float *dists,
                        const float *variance, float cos_anneal_ratio, int64_t n, const float *d_alpha,
                        float *d_sdf, float *d_normal, float *d_variance, void *stream)
{
    if (n <= 0) return 0;
    const unsigned tiles = rsdf_blocks(n, THREADS);
    alpha_bwd_kernel<<<tiles < 4096u ? tiles : 4096u, THREADS, 0, (hipStream_t)stream>>>(
        sdf, normal, dirs, dists, variance, cos_anneal_ratio, n, d_alpha, d_sdf, d_normal, d_variance);
    RSDF_RETURN_LAUNCH();
}

}  // extern "C"
