// S2 / S3: bilinear 2-D grid_sample with first- AND second-order gradients.
//
// Operator contract (paths relative to the upstream RISE-SDF tree):
//   lib/grid_sample_grad2/gridsample_cuda.cpp:26-56   grad2_2d(g2_input, g2_grid, grad_output, input, grid,
//                                                     padding_mode, align_corners) -> [ggOut, gInput, gGrid]
//   lib/grid_sample_grad2/gridsample_cuda.cu:27-210   the double-backward kernel being replaced
//   utils/cuda_gridsample.py:25-73                    autograd nesting (forward = aten grid_sample,
//                                                     backward = aten grid_sampler_2d_backward)
//   models/texture.py:338-341                         the live FG-LUT lookup: dr.texture(LUT[1,256,256,2],
//                                                     uv, 'linear', 'clamp') == grid_sample(LUT NCHW, 2uv-1,
//                                                     bilinear, border, align_corners=False)
// Coordinate conventions are aten's (GridSampler.h): unnormalise, then for border padding clip to
// [0, size-1] with a zero gradient multiplier outside; zeros padding skips out-of-range corners.
// One thread per output location, looping over channels; gradient w.r.t. the input is an atomic scatter
// (the LUT is 512 KiB: its gradient image stays cache resident).
#include "common.h"

namespace {

constexpr int THREADS = 256;

struct GSDims {
    int N, C, H, W, Ho, Wo;
    int border, align;
};

__device__ __forceinline__ float unnorm(float g, int size, int align, int border, float &mult)
{
    float x;
    if (align) { x = (g + 1.0f) * 0.5f * (float)(size - 1); mult = 0.5f * (float)(size - 1); }
    else { x = ((g + 1.0f) * (float)size - 1.0f) * 0.5f; mult = 0.5f * (float)size; }
    if (border) {
        const float hi = (float)(size - 1);
        if (x <= 0.0f) { x = 0.0f; mult = 0.0f; }          // aten: clip + zero gradient when clipped
        else if (x >= hi) { x = hi; mult = 0.0f; }
    }
    return x;
}

struct Taps {
    int x0, y0;
    float w[4], dwx[4], dwy[4];  // nw, ne, sw, se ; d/dix, d/diy
    bool ok[4];
    float mx, my;
};

__device__ __forceinline__ Taps make_taps(float gx, float gy, const GSDims &d)
{
    Taps t;
    const float ix = unnorm(gx, d.W, d.align, d.border, t.mx);
    const float iy = unnorm(gy, d.H, d.align, d.border, t.my);
    const float fx = floorf(ix), fy = floorf(iy);
    t.x0 = (int)fx;
    t.y0 = (int)fy;
    const float tx = ix - fx, ty = iy - fy;
    t.w[0] = (1.f - tx) * (1.f - ty); t.w[1] = tx * (1.f - ty); t.w[2] = (1.f - tx) * ty; t.w[3] = tx * ty;
    t.dwx[0] = -(1.f - ty); t.dwx[1] = (1.f - ty); t.dwx[2] = -ty; t.dwx[3] = ty;
    t.dwy[0] = -(1.f - tx); t.dwy[1] = -tx; t.dwy[2] = (1.f - tx); t.dwy[3] = tx;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = t.x0 + (k & 1), y = t.y0 + (k >> 1);
        t.ok[k] = x >= 0 && x < d.W && y >= 0 && y < d.H;
    }
    return t;
}

__device__ __forceinline__ int64_t tap_off(const Taps &t, int k, const GSDims &d)
{
    return (int64_t)(t.y0 + (k >> 1)) * d.W + (t.x0 + (k & 1));
}

__global__ void __launch_bounds__(THREADS)
gs_fwd_kernel(const float *__restrict__ input, const float *__restrict__ grid, GSDims d,
              float *__restrict__ output)
{
    const int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    const int64_t per = (int64_t)d.Ho * d.Wo;
    if (idx >= d.N * per) return;
    const int n = (int)(idx / per);
    const int64_t o = idx - n * per;
    const Taps t = make_taps(grid[2 * idx], grid[2 * idx + 1], d);
    const int64_t plane = (int64_t)d.H * d.W;
    for (int c = 0; c < d.C; ++c) {
        const float *I = input + ((int64_t)n * d.C + c) * plane;
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (t.ok[k]) acc += t.w[k] * I[tap_off(t, k, d)];
        output[((int64_t)n * d.C + c) * per + o] = acc;
    }
}

__global__ void __launch_bounds__(THREADS)
gs_bwd_kernel(const float *__restrict__ gout, const float *__restrict__ input,
              const float *__restrict__ grid, GSDims d, float *__restrict__ ginput,
              float *__restrict__ ggrid)
{
    const int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    const int64_t per = (int64_t)d.Ho * d.Wo;
    if (idx >= d.N * per) return;
    const int n = (int)(idx / per);
    const int64_t o = idx - n * per;
    const Taps t = make_taps(grid[2 * idx], grid[2 * idx + 1], d);
    const int64_t plane = (int64_t)d.H * d.W;
    float gx = 0.0f, gy = 0.0f;
    for (int c = 0; c < d.C; ++c) {
        const float g = gout[((int64_t)n * d.C + c) * per + o];
        const float *I = input + ((int64_t)n * d.C + c) * plane;
        float *gI = ginput ? ginput + ((int64_t)n * d.C + c) * plane : nullptr;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!t.ok[k]) continue;
            const int64_t off = tap_off(t, k, d);
            const float v = I[off];
            gx += g * t.dwx[k] * v;
            gy += g * t.dwy[k] * v;
            if (gI) atomicAdd(gI + off, g * t.w[k]);
        }
    }
    if (ggrid) {
        ggrid[2 * idx] = gx * t.mx;
        ggrid[2 * idx + 1] = gy * t.my;
    }
}

// second order: see the derivation in DESIGN.md ("grid_sample double backward")
__global__ void __launch_bounds__(THREADS)
gs_bwd2_kernel(const float *__restrict__ g2in, const float *__restrict__ g2grid,
               const float *__restrict__ gout, const float *__restrict__ input,
               const float *__restrict__ grid, GSDims d, float *__restrict__ ggout,
               float *__restrict__ ginput, float *__restrict__ ggrid)
{
    const int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    const int64_t per = (int64_t)d.Ho * d.Wo;
    if (idx >= d.N * per) return;
    const int n = (int)(idx / per);
    const int64_t o = idx - n * per;
    const Taps t = make_taps(grid[2 * idx], grid[2 * idx + 1], d);
    const int64_t plane = (int64_t)d.H * d.W;
    const float a = g2grid ? g2grid[2 * idx] * t.mx : 0.0f;      // g2_grid.x * d ix / d gx
    const float b = g2grid ? g2grid[2 * idx + 1] * t.my : 0.0f;
    // mixed second derivative of the bilinear weights: nw +1, ne -1, sw -1, se +1
    const float dxy[4] = {1.f, -1.f, -1.f, 1.f};
    float ggx = 0.0f, ggy = 0.0f;
    for (int c = 0; c < d.C; ++c) {
        const int64_t po = ((int64_t)n * d.C + c) * per + o;
        const float g = gout[po];
        const float *I = input + ((int64_t)n * d.C + c) * plane;
        const float *G2 = g2in ? g2in + ((int64_t)n * d.C + c) * plane : nullptr;
        float *gI = ginput ? ginput + ((int64_t)n * d.C + c) * plane : nullptr;
        float s = 0.0f;        // ggOut accumulator
        float dx2 = 0.f, dy2 = 0.f, mix = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!t.ok[k]) continue;
            const int64_t off = tap_off(t, k, d);
            const float v = I[off];
            const float coef = a * t.dwx[k] + b * t.dwy[k];
            s += coef * v;
            mix += dxy[k] * v;
            if (G2) {
                const float q = G2[off];
                s += t.w[k] * q;
                dx2 += t.dwx[k] * q;
                dy2 += t.dwy[k] * q;
            }
            if (gI) atomicAdd(gI + off, g * coef);
        }
        if (ggout) ggout[po] = s;
        ggx += g * (dx2 + b * mix);
        ggy += g * (dy2 + a * mix);
    }
    if (ggrid) {
        ggrid[2 * idx] = ggx * t.mx;
        ggrid[2 * idx + 1] = ggy * t.my;
    }
}

int check_dims(const GSDims &d)
{
    return d.N >= 0 && d.C >= 1 && d.H >= 1 && d.W >= 1 && d.Ho >= 0 && d.Wo >= 0;
}

}  // namespace

extern "C" {

int rsdf_grid_sample2d_fwd(const float *input, const float *grid, int N, int C, int H, int W, int Ho,
                           int Wo, int padding_border, int align_corners, float *output, void *stream)
{
    const GSDims d{N, C, H, W, Ho, Wo, padding_border, align_corners};
    RSDF_CHECK_ARG(check_dims(d), "grid_sample2d_fwd: bad dimensions");
    const int64_t n = (int64_t)N * Ho * Wo;
    if (n <= 0) return 0;
    gs_fwd_kernel<<<rsdf_blocks(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(input, grid, d, output);
    RSDF_RETURN_LAUNCH();
}

int rsdf_grid_sample2d_bwd(const float *grad_output, const float *input, const float *grid, int N, int C,
                           int H, int W, int Ho, int Wo, int padding_border, int align_corners,
                           float *grad_input, float *grad_grid, void *stream)
{
    const GSDims d{N, C, H, W, Ho, Wo, padding_border, align_corners};
    RSDF_CHECK_ARG(check_dims(d), "grid_sample2d_bwd: bad dimensions");
    const int64_t n = (int64_t)N * Ho * Wo;
    if (n <= 0) return 0;
    gs_bwd_kernel<<<rsdf_blocks(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(grad_output, input, grid, d,
                                                                                grad_input, grad_grid);
    RSDF_RETURN_LAUNCH();
}

int rsdf_grid_sample2d_bwd2(const float *g2_input, const float *g2_grid, const float *grad_output,
                            const float *input, const float *grid, int N, int C, int H, int W, int Ho,
                            int Wo, int padding_border, int align_corners, float *gg_out, float *g_input,
                            float *g_grid, void *stream)
{
    const GSDims d{N, C, H, W, Ho, Wo, padding_border, align_corners};
    RSDF_CHECK_ARG(check_dims(d), "grid_sample2d_bwd2: bad dimensions");
    const int64_t n = (int64_t)N * Ho * Wo;
    if (n <= 0) return 0;
    gs_bwd2_kernel<<<rsdf_blocks(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        g2_input, g2_grid, grad_output, input, grid, d, gg_out, g_input, g_grid);
    RSDF_RETURN_LAUNCH();
}

}  // extern "C"
