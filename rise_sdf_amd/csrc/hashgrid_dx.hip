// H1 input gradient and its backward: what analytic normals (models/geometry.py:224-228, grad_type
// 'analytic') and the curvature term (geometry.py:246-282) reach through tcnn's autograd.
//
// With pos = scale_l x + 0.5, w = frac(pos), omega_k(0) = 1 - w_k, omega_k(1) = w_k and corner weight
// w_c = prod_k omega_k(c_k):
//   d w_c / d x_d       = scale_l * sgn_d(c) * prod_{k != d} omega_k(c_k)
//   d2 w_c / d x_d d x_e = scale_l^2 * sgn_d(c) sgn_e(c) * omega_k(c_k), k the remaining axis (0 for d == e)
//
//   dx kernel      dx_d = sum_l sum_f dy[l,f] sum_c T_l[idx_c][f] * d w_c / d x_d
//   dx_bwd kernel  given g = dL/d(dx):  m_c = sum_d g_d d w_c / d x_d
//                  ddy[l,f]        = sum_c T_l[idx_c][f] m_c
//                  dtable[idx_c,f] += dy[l,f] m_c                       (float atomics)
//                  gx_e           = sum_l sum_f dy[l,f] sum_c T_l[idx_c][f] sum_{d != e} g_d d2 w_c / d x_d d x_e
// One thread per sample walks the active levels in order, so dx / gx / ddy are deterministic; only the
// table scatter uses atomics.  These kernels serve the c4-only curvature row and analytic-normal configs;
// the finite-difference hot path never calls them.
#include "common.h"
#include "hashgrid_common.h"

namespace {

constexpr int THREADS = 256;

template <int F>
struct Gather {
    float v[8][F];
    __device__ __forceinline__ void load(const float *table, const LevelInfo &li, const CellFrac &cf, uint32_t *idx)
    {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            idx[c] = grid_index(cf.c[0] + (c & 1), cf.c[1] + ((c >> 1) & 1), cf.c[2] + ((c >> 2) & 1), li);
            const float *p = table + ((size_t)li.offset + idx[c]) * F;
#pragma unroll
            for (int f = 0; f < F; ++f) v[c][f] = p[f];
        }
    }
};

__device__ __forceinline__ float omega(const CellFrac &cf, int c, int k)
{
    return ((c >> k) & 1) ? cf.w[k] : 1.0f - cf.w[k];
}
__device__ __forceinline__ float sgn(int c, int k) { return ((c >> k) & 1) ? 1.0f : -1.0f; }

template <int F>
__global__ void __launch_bounds__(THREADS)
hashgrid_dx_kernel(const float *__restrict__ x, const float *__restrict__ table, const rsdf_grid_meta meta,
                   int64_t n, int n_active, const float *__restrict__ dy, int ld_dy, int col_off,
                   float *__restrict__ dx)
{
    const int64_t s = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (s >= n) return;
    const float px = x[3 * s], py = x[3 * s + 1], pz = x[3 * s + 2];
    float acc[3] = {0.f, 0.f, 0.f};
    for (int l = 0; l < n_active; ++l) {
        const LevelInfo li = level_info(meta, l);
        const CellFrac cf = cell_frac(px, py, pz, li.scale);
        Gather<F> g;
        uint32_t idx[8];
        g.load(table, li, cf, idx);
        const float *d = dy + s * ld_dy + col_off + l * F;
        float part[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float t = 0.f;
#pragma unroll
            for (int f = 0; f < F; ++f) t = fmaf(d[f], g.v[c][f], t);
            const float o0 = omega(cf, c, 0), o1 = omega(cf, c, 1), o2 = omega(cf, c, 2);
            part[0] = fmaf(t, sgn(c, 0) * o1 * o2, part[0]);
            part[1] = fmaf(t, sgn(c, 1) * o0 * o2, part[1]);
            part[2] = fmaf(t, sgn(c, 2) * o0 * o1, part[2]);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] = fmaf(li.scale, part[k], acc[k]);
    }
    dx[3 * s] = acc[0]; dx[3 * s + 1] = acc[1]; dx[3 * s + 2] = acc[2];
}

template <int F>
__global__ void __launch_bounds__(THREADS)
hashgrid_dx_bwd_kernel(const float *__restrict__ x, const float *__restrict__ table, const rsdf_grid_meta meta,
                       int64_t n, int n_active, const float *__restrict__ dy, int ld_dy, int col_off,
                       const float *__restrict__ gdx, float *__restrict__ ddy, int ld_ddy, int col_off_ddy,
                       float *__restrict__ dtable, float *__restrict__ gx)
{
    const int64_t s = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (s >= n) return;
    const float px = x[3 * s], py = x[3 * s + 1], pz = x[3 * s + 2];
    const float g0 = gdx[3 * s], g1 = gdx[3 * s + 1], g2 = gdx[3 * s + 2];
    float acc[3] = {0.f, 0.f, 0.f};
    const int L = meta.n_levels;
    for (int l = 0; l < L; ++l) {
        float *od = ddy ? ddy + s * ld_ddy + col_off_ddy + l * F : nullptr;
        if (l >= n_active) {
            if (od)
#pragma unroll
                for (int f = 0; f < F; ++f) od[f] = 0.f;
            continue;
        }
        const LevelInfo li = level_info(meta, l);
        const CellFrac cf = cell_frac(px, py, pz, li.scale);
        Gather<F> g;
        uint32_t idx[8];
        g.load(table, li, cf, idx);
        const float *d = dy + s * ld_dy + col_off + l * F;
        float dyv[F], dd[F];
#pragma unroll
        for (int f = 0; f < F; ++f) { dyv[f] = d[f]; dd[f] = 0.f; }
        float part[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float o0 = omega(cf, c, 0), o1 = omega(cf, c, 1), o2 = omega(cf, c, 2);
            const float s0 = sgn(c, 0), s1 = sgn(c, 1), s2 = sgn(c, 2);
            // m_c / scale
            const float m = g0 * s0 * o1 * o2 + g1 * s1 * o0 * o2 + g2 * s2 * o0 * o1;
            float t = 0.f;
#pragma unroll
            for (int f = 0; f < F; ++f) {
                dd[f] = fmaf(g.v[c][f], m, dd[f]);
                t = fmaf(dyv[f], g.v[c][f], t);
            }
            if (dtable) {
                float *tp = dtable + ((size_t)li.offset + idx[c]) * F;
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    const float a = li.scale * dyv[f] * m;
                    if (a != 0.0f) atomicAdd(tp + f, a);
                }
            }
            // mixed second derivatives / scale^2
            part[0] = fmaf(t, s0 * (g1 * s1 * o2 + g2 * s2 * o1), part[0]);
            part[1] = fmaf(t, s1 * (g0 * s0 * o2 + g2 * s2 * o0), part[1]);
            part[2] = fmaf(t, s2 * (g0 * s0 * o1 + g1 * s1 * o0), part[2]);
        }
        if (od)
#pragma unroll
            for (int f = 0; f < F; ++f) od[f] = li.scale * dd[f];
        const float s2c = li.scale * li.scale;
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] = fmaf(s2c, part[k], acc[k]);
    }
    if (gx) { gx[3 * s] = acc[0]; gx[3 * s + 1] = acc[1]; gx[3 * s + 2] = acc[2]; }
}

}  // namespace

extern "C" {

int rsdf_hashgrid_dx(const float *x, const float *table, const rsdf_grid_meta *meta, int64_t n, int n_active_levels,
                     const float *dy, int ld_dy, int col_off, float *dx, void *stream)
{
    if (!meta || meta->n_levels > RSDF_MAX_LEVELS || n_active_levels > (int)meta->n_levels) return hipErrorInvalidValue;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((n + THREADS - 1) / THREADS));
    switch (meta->n_features) {
    case 1: hashgrid_dx_kernel<1><<<grid, THREADS, 0, st>>>(x, table, *meta, n, n_active_levels, dy, ld_dy, col_off, dx); break;
    case 2: hashgrid_dx_kernel<2><<<grid, THREADS, 0, st>>>(x, table, *meta, n, n_active_levels, dy, ld_dy, col_off, dx); break;
    case 4: hashgrid_dx_kernel<4><<<grid, THREADS, 0, st>>>(x, table, *meta, n, n_active_levels, dy, ld_dy, col_off, dx); break;
    default: return hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

int rsdf_hashgrid_dx_bwd(const float *x, const float *table, const rsdf_grid_meta *meta, int64_t n,
                         int n_active_levels, const float *dy, int ld_dy, int col_off, const float *g_dx,
                         float *d_dy, int ld_ddy, int col_off_ddy, float *dtable, float *g_x, void *stream)
{
    if (!meta || meta->n_levels > RSDF_MAX_LEVELS || n_active_levels > (int)meta->n_levels) return hipErrorInvalidValue;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((n + THREADS - 1) / THREADS));
    switch (meta->n_features) {
    case 1: hashgrid_dx_bwd_kernel<1><<<grid, THREADS, 0, st>>>(x, table, *meta, n, n_active_levels, dy, ld_dy, col_off, g_dx, d_dy, ld_ddy, col_off_ddy, dtable, g_x); break;
    case 2: hashgrid_dx_bwd_kernel<2><<<grid, THREADS, 0, st>>>(x, table, *meta, n, n_active_levels, dy, ld_dy, col_off, g_dx, d_dy, ld_ddy, col_off_ddy, dtable, g_x); break;
    case 4: hashgrid_dx_bwd_kernel<4><<<grid, THREADS, 0, st>>>(x, table, *meta, n, n_active_levels, dy, ld_dy, col_off, g_dx, d_dy, ld_ddy, col_off_ddy, dtable, g_x); break;
    default: return hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

}  // extern "C"
