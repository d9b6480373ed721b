// T1 / T2 / S1 (stage 0): the small per-sample kernels of the radiance branch.
//
// Specification followed (paths relative to the upstream RISE-SDF tree):
//   models/network_utils.py:14-40    VanillaFrequency: [sin(2^k x), cos(2^k x)] k = 0..n-1, freq-major
//   models/network_utils.py:98-99    tcnn.Encoding(otype SphericalHarmonics, degree 5): tiny-cuda-nn is absent
//                                    upstream; the published real-SH polynomial basis of (2 d - 1) is used
//                                    (parity unpinned, identical to oracle/texture.py)
//   models/texture.py:295-297        wi = -d, wo = 2 (wi.n) n - wi, NoV = n.wi
//   models/texture.py:303-327        sigmoid activations + blend (stage 0 output [diff(3), spec(3), blend])
//   lib/pbr/utils/nvdiffrecmc_util.py:95-103   rgb_to_srgb
#include "common.h"

namespace {

constexpr int THREADS = 256;

// A workgroup's rows [row0, row0 + THREADS) x m columns, one row per thread in LDS ([THREADS][m | 1]: odd stride), written to
// out[row][col_off .. col_off + m) by consecutive threads along each row (round 6: one thread storing its own row put every
// store instruction on 64 different rows at the row stride -- the frequency encoder ran at 1 TB/s of its 168 B per row).
__device__ __forceinline__ void store_rows_staged(const float *s_rows, int m, int64_t row0, int64_t n, float *__restrict__ out, int ld,
                                                  int col_off)
{
    __syncthreads();
    const int rows = (int)(n - row0 < THREADS ? n - row0 : THREADS), lds = m | 1;
    for (int e = threadIdx.x; e < rows * m; e += THREADS) {
        const int r = e / m, c = e - r * m;
        out[(row0 + r) * ld + col_off + c] = s_rows[r * lds + c];
    }
}

__global__ void __launch_bounds__(THREADS)
freq_encode_kernel(const float *__restrict__ x, int64_t n, int n_freq, float x_scale, float x_offset,
                   const float *__restrict__ mask, float *__restrict__ out, int ld, int col_off)
{
    extern __shared__ float s_rows[];      // [THREADS][6 n_freq | 1]
    const int64_t row0 = (int64_t)blockIdx.x * THREADS, i = row0 + threadIdx.x;
    const int m6 = 6 * n_freq;
    if (i < n) {
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = x[3 * i + c] * x_scale + x_offset;
        float *o = s_rows + threadIdx.x * (m6 | 1);
        float f = 1.0f;
        for (int k = 0; k < n_freq; ++k, f *= 2.0f) {
            const float m = mask ? mask[k] : 1.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float a = f * v[c];
                o[k * 6 + c] = sinf(a) * m;
                o[k * 6 + 3 + c] = cosf(a) * m;
            }
        }
    }
    store_rows_staged(s_rows, m6, row0, n, out, ld, col_off);
}

// real SH basis, up to 5 bands; u = 2 d01 - 1
__device__ __forceinline__ void sh_eval(float x, float y, float z, int degree, float *o)
{
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    const float x4 = x2 * x2, y4 = y2 * y2, z4 = z2 * z2;
    o[0] = 0.28209479177387814f;
    if (degree <= 1) return;
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
    if (degree <= 2) return;
    o[4] = 1.0925484305920792f * xy;
    o[5] = -1.0925484305920792f * yz;
    o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o[7] = -1.0925484305920792f * xz;
    o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    if (degree <= 3) return;
    o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    o[10] = 2.8906114426405538f * xy * z;
    o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    o[14] = 1.4453057213202769f * z * (x2 - y2);
    o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
    if (degree <= 4) return;
    o[16] = 2.5033429417967046f * xy * (x2 - y2);
    o[17] = 1.7701307697799304f * yz * (-3.0f * x2 + y2);
    o[18] = 0.94617469575756008f * xy * (7.0f * z2 - 1.0f);
    o[19] = 0.66904654355728921f * yz * (3.0f - 7.0f * z2);
    o[20] = -3.1735664074561294f * z2 + 3.7024941420321507f * z4 + 0.31735664074561293f;
    o[21] = 0.66904654355728921f * xz * (3.0f - 7.0f * z2);
    o[22] = 0.47308734787878004f * (x2 - y2) * (7.0f * z2 - 1.0f);
    o[23] = 1.7701307697799304f * xz * (-x2 + 3.0f * y2);
    o[24] = -3.7550144126950569f * x2 * y2 + 0.62583573544917614f * x4 + 0.62583573544917614f * y4;
}

// (dx,dy,dz) += sum_i g[i] * d o_i / d(x,y,z)
__device__ __forceinline__ void sh_grad(float x, float y, float z, int degree, const float *g, float &dx,
                                        float &dy, float &dz)
{
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    dx = dy = dz = 0.0f;
    if (degree <= 1) return;
    const float c1 = 0.48860251190291987f;
    dy += -c1 * g[1];
    dz += c1 * g[2];
    dx += -c1 * g[3];
    if (degree <= 2) return;
    const float c2 = 1.0925484305920792f, c3 = 0.54627421529603959f;
    dx += c2 * y * g[4];  dy += c2 * x * g[4];
    dy += -c2 * z * g[5]; dz += -c2 * y * g[5];
    dz += 2.0f * 0.94617469575755997f * z * g[6];
    dx += -c2 * z * g[7]; dz += -c2 * x * g[7];
    dx += 2.0f * c3 * x * g[8]; dy += -2.0f * c3 * y * g[8];
    if (degree <= 3) return;
    const float a = 0.59004358992664352f, b = 2.8906114426405538f, c = 0.45704579946446572f,
                d = 0.3731763325901154f, e = 1.4453057213202769f;
    dx += -6.0f * a * xy * g[9];            dy += a * (-3.0f * x2 + 3.0f * y2) * g[9];
    dx += b * yz * g[10];                   dy += b * xz * g[10];              dz += b * xy * g[10];
    dy += c * (1.0f - 5.0f * z2) * g[11];   dz += -10.0f * c * yz * g[11];
    dz += d * (15.0f * z2 - 3.0f) * g[12];
    dx += c * (1.0f - 5.0f * z2) * g[13];   dz += -10.0f * c * xz * g[13];
    dx += 2.0f * e * xz * g[14];            dy += -2.0f * e * yz * g[14];      dz += e * (x2 - y2) * g[14];
    dx += a * (-3.0f * x2 + 3.0f * y2) * g[15]; dy += 6.0f * a * xy * g[15];
    if (degree <= 4) return;
    const float f = 2.5033429417967046f, gg = 1.7701307697799304f, h = 0.94617469575756008f,
                ii = 0.66904654355728921f, j = 0.47308734787878004f, k = 0.62583573544917614f;
    dx += f * (3.0f * x2 * y - y2 * y) * g[16];      dy += f * (x2 * x - 3.0f * x * y2) * g[16];
    dx += -6.0f * gg * xy * z * g[17];               dy += gg * z * (-3.0f * x2 + 3.0f * y2) * g[17];
    dz += gg * y * (-3.0f * x2 + y2) * g[17];
    dx += h * y * (7.0f * z2 - 1.0f) * g[18];        dy += h * x * (7.0f * z2 - 1.0f) * g[18];
    dz += 14.0f * h * xy * z * g[18];
    dy += ii * z * (3.0f - 7.0f * z2) * g[19];       dz += ii * y * (3.0f - 21.0f * z2) * g[19];
    dz += (-6.3471328149122588f * z + 14.809976568128603f * z2 * z) * g[20];
    dx += ii * z * (3.0f - 7.0f * z2) * g[21];       dz += ii * x * (3.0f - 21.0f * z2) * g[21];
    dx += 2.0f * j * x * (7.0f * z2 - 1.0f) * g[22]; dy += -2.0f * j * y * (7.0f * z2 - 1.0f) * g[22];
    dz += 14.0f * j * z * (x2 - y2) * g[22];
    dx += gg * z * (-3.0f * x2 + 3.0f * y2) * g[23]; dy += 6.0f * gg * xy * z * g[23];
    dz += gg * x * (-x2 + 3.0f * y2) * g[23];
    dx += (-7.5100288253901138f * x * y2 + 4.0f * k * x2 * x) * g[24];
    dy += (-7.5100288253901138f * x2 * y + 4.0f * k * y2 * y) * g[24];
}

__global__ void __launch_bounds__(THREADS)
sh_fwd_kernel(const float *__restrict__ d01, int64_t n, int degree, float *__restrict__ out, int ld,
              int col_off)
{
    extern __shared__ float s_rows[];      // [THREADS][degree^2 | 1]
    const int64_t row0 = (int64_t)blockIdx.x * THREADS, i = row0 + threadIdx.x;
    const int m = degree * degree;
    if (i < n) {
        float o[25];
        sh_eval(d01[3 * i] * 2.0f - 1.0f, d01[3 * i + 1] * 2.0f - 1.0f, d01[3 * i + 2] * 2.0f - 1.0f, degree, o);
        float *dst = s_rows + threadIdx.x * (m | 1);
#pragma unroll
        for (int k = 0; k < 25; ++k)
            if (k < m) dst[k] = o[k];
    }
    store_rows_staged(s_rows, m, row0, n, out, ld, col_off);
}

__global__ void __launch_bounds__(THREADS)
sh_bwd_kernel(const float *__restrict__ d01, const float *__restrict__ dout, int64_t n, int degree, int ld,
              int col_off, float *__restrict__ dd01)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    float g[25];
    const float *src = dout + i * ld + col_off;
    const int m = degree * degree;
#pragma unroll
    for (int k = 0; k < 25; ++k) g[k] = k < m ? src[k] : 0.0f;
    float dx, dy, dz;
    sh_grad(d01[3 * i] * 2.0f - 1.0f, d01[3 * i + 1] * 2.0f - 1.0f, d01[3 * i + 2] * 2.0f - 1.0f, degree, g,
            dx, dy, dz);
    dd01[3 * i] = 2.0f * dx;
    dd01[3 * i + 1] = 2.0f * dy;
    dd01[3 * i + 2] = 2.0f * dz;
}

// wo01 = (wo + 1) / 2 with wo = 2 (wi.n) n - wi, wi = -d ; nov = n.wi
__global__ void __launch_bounds__(THREADS)
reflect_fwd_kernel(const float *__restrict__ dirs, const float *__restrict__ normals, int64_t n,
                   float *__restrict__ wo01, float *__restrict__ nov)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const float w[3] = {-dirs[3 * i], -dirs[3 * i + 1], -dirs[3 * i + 2]};
    const float nn[3] = {normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]};
    const float dot = w[0] * nn[0] + w[1] * nn[1] + w[2] * nn[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) wo01[3 * i + c] = ((dot * nn[c] * 2.0f - w[c]) + 1.0f) / 2.0f;
    if (nov) nov[i] = dot;
}

__global__ void __launch_bounds__(THREADS)
reflect_bwd_kernel(const float *__restrict__ dirs, const float *__restrict__ normals, int64_t n,
                   const float *__restrict__ d_wo01, const float *__restrict__ d_nov,
                   float *__restrict__ d_normals)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const float w[3] = {-dirs[3 * i], -dirs[3 * i + 1], -dirs[3 * i + 2]};
    const float nn[3] = {normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]};
    const float dot = w[0] * nn[0] + w[1] * nn[1] + w[2] * nn[2];
    float g[3] = {0.f, 0.f, 0.f};
    if (d_wo01) { g[0] = d_wo01[3 * i] * 0.5f; g[1] = d_wo01[3 * i + 1] * 0.5f; g[2] = d_wo01[3 * i + 2] * 0.5f; }
    // wo_c = 2 dot n_c - w_c ; d wo_c / d n_k = 2 w_k n_c + 2 dot delta_ck
    const float gn = g[0] * nn[0] + g[1] * nn[1] + g[2] * nn[2];
    const float dn = d_nov ? d_nov[i] : 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) d_normals[3 * i + k] = 2.0f * w[k] * gn + 2.0f * dot * g[k] + dn * w[k];
}

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// stage 0: colors[7] = [(1-blend) sig(albedo6[0:3]), blend sig(spec3), blend], blend = sig(metallic2[0])
__global__ void __launch_bounds__(THREADS)
split_color0_fwd_kernel(const float *__restrict__ albedo6, const float *__restrict__ metallic2,
                        const float *__restrict__ spec3, int64_t n, float *__restrict__ colors)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const float blend = sigm(metallic2[2 * i]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        colors[7 * i + c] = (1.0f - blend) * sigm(albedo6[6 * i + c]);
        colors[7 * i + 3 + c] = blend * sigm(spec3[3 * i + c]);
    }
    colors[7 * i + 6] = blend;
}

__global__ void __launch_bounds__(THREADS)
split_color0_bwd_kernel(const float *__restrict__ albedo6, const float *__restrict__ metallic2,
                        const float *__restrict__ spec3, const float *__restrict__ d_colors, int64_t n,
                        float *__restrict__ d_albedo6, float *__restrict__ d_metallic2,
                        float *__restrict__ d_spec3)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const float blend = sigm(metallic2[2 * i]);
    float d_blend = d_colors[7 * i + 6];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float sd = sigm(albedo6[6 * i + c]), ss = sigm(spec3[3 * i + c]);
        const float gd = d_colors[7 * i + c], gs = d_colors[7 * i + 3 + c];
        d_albedo6[6 * i + c] = gd * (1.0f - blend) * sd * (1.0f - sd);
        d_albedo6[6 * i + 3 + c] = 0.0f;
        d_spec3[3 * i + c] = gs * blend * ss * (1.0f - ss);
        d_blend += -gd * sd + gs * ss;
    }
    d_metallic2[2 * i] = d_blend * blend * (1.0f - blend);
    d_metallic2[2 * i + 1] = 0.0f;
}

// stage 1 (split-sum shading, models/texture.py:329-345) on ACTIVATED material values:
//   a6 = sigmoid(albedo net) [diff_rgb(3) | albedo(3)], m2 = sigmoid(metallic net) [blend | metallic],
//   r = sigmoid(roughness net), s3 = sigmoid(env net); Ld/Ls diffuse / specular light, fg the LUT lookup.
// colors24 = [diff_rf(3), spec_rf(3), blend, diff_pbr(3), spec_pbr(3), spec_ref(3), Ls(3), albedo(3), metallic, r]
__global__ void __launch_bounds__(THREADS)
split_shade1_fwd_kernel(const float *__restrict__ a6, const float *__restrict__ r1,
                        const float *__restrict__ m2, const float *__restrict__ s3,
                        const float *__restrict__ Ld, const float *__restrict__ Ls,
                        const float *__restrict__ fg, int64_t n, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const float blend = m2[2 * i], metal = m2[2 * i + 1];
    const float fgx = fg[2 * i], fgy = fg[2 * i + 1];
    float *o = out + 24 * i;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float alb = a6[6 * i + 3 + c];
        const float sref = (0.04f * (1.0f - metal) + metal * alb) * fgx + fgy;
        o[c] = (1.0f - blend) * a6[6 * i + c];
        o[3 + c] = blend * s3[3 * i + c];
        o[7 + c] = (1.0f - metal) * alb * Ld[3 * i + c];
        o[10 + c] = sref * Ls[3 * i + c];
        o[13 + c] = sref;
        o[16 + c] = Ls[3 * i + c];
        o[19 + c] = alb;
    }
    o[6] = blend;
    o[22] = metal;
    o[23] = r1[i];
}

__global__ void __launch_bounds__(THREADS)
split_shade1_bwd_kernel(const float *__restrict__ a6, const float *__restrict__ m2,
                        const float *__restrict__ s3, const float *__restrict__ Ld,
                        const float *__restrict__ Ls, const float *__restrict__ fg,
                        const float *__restrict__ g, int64_t n, float *__restrict__ d_a6,
                        float *__restrict__ d_r1, float *__restrict__ d_m2, float *__restrict__ d_s3,
                        float *__restrict__ d_Ld, float *__restrict__ d_Ls, float *__restrict__ d_fg)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const float blend = m2[2 * i], metal = m2[2 * i + 1];
    const float fgx = fg[2 * i], fgy = fg[2 * i + 1];
    const float *go = g + 24 * i;
    float d_blend = go[6], d_metal = go[22], d_fgx = 0.f, d_fgy = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float alb = a6[6 * i + 3 + c], ld = Ld[3 * i + c], ls = Ls[3 * i + c];
        const float salb = 0.04f * (1.0f - metal) + metal * alb;
        const float sref = salb * fgx + fgy;
        // d wrt sref: spec_pbr (x ls) + its own channel
        const float d_sref = go[10 + c] * ls + go[13 + c];
        d_Ls[3 * i + c] = go[10 + c] * sref + go[16 + c];
        d_Ld[3 * i + c] = go[7 + c] * (1.0f - metal) * alb;
        d_fgx += d_sref * salb;
        d_fgy += d_sref;
        const float d_salb = d_sref * fgx;
        d_a6[6 * i + c] = go[c] * (1.0f - blend);
        d_a6[6 * i + 3 + c] = go[7 + c] * (1.0f - metal) * ld + d_salb * metal + go[19 + c];
        d_s3[3 * i + c] = go[3 + c] * blend;
        d_blend += -go[c] * a6[6 * i + c] + go[3 + c] * s3[3 * i + c];
        d_metal += -go[7 + c] * alb * ld + d_salb * (alb - 0.04f);
    }
    d_m2[2 * i] = d_blend;
    d_m2[2 * i + 1] = d_metal;
    d_r1[i] = go[23];
    d_fg[2 * i] = d_fgx;
    d_fg[2 * i + 1] = d_fgy;
}

__global__ void __launch_bounds__(THREADS)
srgb_fwd_kernel(const float *__restrict__ x, int64_t n, float *__restrict__ y)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const float f = x[i];
    y[i] = f <= 0.0031308f ? f * 12.92f : powf(fmaxf(f, 0.0031308f), 1.0f / 2.4f) * 1.055f - 0.055f;
}

__global__ void __launch_bounds__(THREADS)
srgb_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy, int64_t n, float *__restrict__ dx)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const float f = x[i];
    // torch.where + clamp: the power branch's clamp passes gradient for f >= 0.0031308
    dx[i] = dy[i] * (f <= 0.0031308f ? 12.92f : 1.055f / 2.4f * powf(f, 1.0f / 2.4f - 1.0f));
}

// O1 (models/split_mixed_occ.py:405-436): y = clamp(rgb_to_srgb(comp + bg (1 - opacity)), 0, 1) per ray, one pass each way
// instead of rsub, mul, add, the sRGB kernel and clamp (and clamp's five-kernel backward).  x is formed as torch forms it
// (this file is built with -ffp-contract=off): same values as the unfused chain, bit for bit.
__device__ __forceinline__ float srgb_of(float f)
{
    return f <= 0.0031308f ? f * 12.92f : powf(fmaxf(f, 0.0031308f), 1.0f / 2.4f) * 1.055f - 0.055f;
}

__global__ void __launch_bounds__(THREADS)
compose_srgb_fwd_kernel(const float *__restrict__ comp, const float *__restrict__ bg, const float *__restrict__ opacity,
                        int64_t n, float *__restrict__ y)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= 3 * n) return;
    const int64_t row = i / 3;
    const int c = (int)(i - 3 * row);
    const float x = comp[i] + bg[c] * (1.0f - opacity[row]);
    y[i] = fminf(fmaxf(srgb_of(x), 0.0f), 1.0f);
}

__global__ void __launch_bounds__(THREADS)
compose_srgb_bwd_kernel(const float *__restrict__ comp, const float *__restrict__ bg, const float *__restrict__ opacity,
                        const float *__restrict__ dy, int64_t n, float *__restrict__ d_comp, float *__restrict__ d_opacity)
{
    const int64_t row = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (row >= n) return;
    const float om = 1.0f - opacity[row];
    float dop = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float x = comp[3 * row + c] + bg[c] * om;
        const float yp = srgb_of(x);
        // clamp passes the gradient on [0, 1] (bounds included), then the sRGB derivative (srgb_bwd_kernel)
        const float g = (yp >= 0.0f && yp <= 1.0f)
                            ? dy[3 * row + c] * (x <= 0.0031308f ? 12.92f : 1.055f / 2.4f * powf(x, 1.0f / 2.4f - 1.0f))
                            : 0.0f;
        d_comp[3 * row + c] = g;
        dop -= g * bg[c];
    }
    if (d_opacity != nullptr) d_opacity[row] = dop;
}

// H4 (models/geometry.py:224-228 through autograd; here rise_sdf_amd/geometry.py::field_with_analytic_grad): a hidden layer of the
// analytic-gradient sweep needs the activation AND its slope, h = softplus(z, beta = 100) (torch: z for 100 z > 20, else
// log1p(exp(100 z)) / 100) and s = sigmoid(100 z) = dh/dz.  One kernel each way instead of torch's softplus + mul + sigmoid and
// their four backward kernels on [n, 128] rows (0.3 ms of an 18 ms training step):  dz = dh s + ds 100 s (1 - s).
__global__ void __launch_bounds__(THREADS)
softplus100_slope_fwd_kernel(const float *__restrict__ z, int64_t n, float *__restrict__ h, float *__restrict__ s)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const float bz = z[i] * 100.0f;
    h[i] = bz > 20.0f ? z[i] : log1pf(expf(bz)) / 100.0f;
    s[i] = 1.0f / (1.0f + expf(-bz));
}

__global__ void __launch_bounds__(THREADS)
softplus100_slope_bwd_kernel(const float *__restrict__ s, const float *__restrict__ dh, const float *__restrict__ ds, int64_t n,
                             float *__restrict__ dz)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const float si = s[i];
    float g = dh != nullptr ? dh[i] * si : 0.0f;
    if (ds != nullptr) g += ds[i] * (100.0f * (si * (1.0f - si)));
    dz[i] = g;
}

}  // namespace

#define LAUNCH1D(kern, n, ...) kern<<<rsdf_blocks(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(__VA_ARGS__)

extern "C" {

int rsdf_freq_encode(const float *x, int64_t n, int n_frequencies, float x_scale, float x_offset,
                     const float *mask, float *out, int ld_out, int col_off, void *stream)
{
    RSDF_CHECK_ARG(n_frequencies >= 1 && ld_out >= col_off + 6 * n_frequencies, "freq_encode: bad sizes");
    if (n <= 0) return 0;
    RSDF_CHECK_ARG(n_frequencies <= 24, "freq_encode: at most 24 frequencies (the row staging's LDS)");
    const size_t lds = (size_t)THREADS * ((6 * n_frequencies) | 1) * sizeof(float);
    if (lds > 65536)
        if (int rc = rsdf_func_lds(reinterpret_cast<const void *>(freq_encode_kernel), lds)) return rc;
    freq_encode_kernel<<<rsdf_blocks(n, THREADS), THREADS, lds, (hipStream_t)stream>>>(x, n, n_frequencies, x_scale, x_offset, mask, out,
                                                                                       ld_out, col_off);
    RSDF_RETURN_LAUNCH();
}

int rsdf_sh_encode_fwd(const float *d01, int64_t n, int degree, float *out, int ld_out, int col_off,
                       void *stream)
{
    RSDF_CHECK_ARG(degree >= 1 && degree <= 5 && ld_out >= col_off + degree * degree, "sh_encode_fwd: bad sizes");
    if (n <= 0) return 0;
    sh_fwd_kernel<<<rsdf_blocks(n, THREADS), THREADS, (size_t)THREADS * ((degree * degree) | 1) * sizeof(float), (hipStream_t)stream>>>(
        d01, n, degree, out, ld_out, col_off);
    RSDF_RETURN_LAUNCH();
}

int rsdf_sh_encode_bwd(const float *d01, const float *dout, int64_t n, int degree, int ld_dout, int col_off,
                       float *d_d01, void *stream)
{
    RSDF_CHECK_ARG(degree >= 1 && degree <= 5 && ld_dout >= col_off + degree * degree, "sh_encode_bwd: bad sizes");
    if (n <= 0) return 0;
    LAUNCH1D(sh_bwd_kernel, n, d01, dout, n, degree, ld_dout, col_off, d_d01);
    RSDF_RETURN_LAUNCH();
}

int rsdf_reflect_fwd(const float *dirs, const float *normals, int64_t n, float *wo01, float *nov,
                     void *stream)
{
    if (n <= 0) return 0;
    LAUNCH1D(reflect_fwd_kernel, n, dirs, normals, n, wo01, nov);
    RSDF_RETURN_LAUNCH();
}

int rsdf_reflect_bwd(const float *dirs, const float *normals, int64_t n, const float *d_wo01,
                     const float *d_nov, float *d_normals, void *stream)
{
    if (n <= 0) return 0;
    LAUNCH1D(reflect_bwd_kernel, n, dirs, normals, n, d_wo01, d_nov, d_normals);
    RSDF_RETURN_LAUNCH();
}

int rsdf_split_color0_fwd(const float *albedo6, const float *metallic2, const float *spec3, int64_t n,
                          float *colors7, void *stream)
{
    if (n <= 0) return 0;
    LAUNCH1D(split_color0_fwd_kernel, n, albedo6, metallic2, spec3, n, colors7);
    RSDF_RETURN_LAUNCH();
}

int rsdf_split_color0_bwd(const float *albedo6, const float *metallic2, const float *spec3,
                          const float *d_colors7, int64_t n, float *d_albedo6, float *d_metallic2,
                          float *d_spec3, void *stream)
{
    if (n <= 0) return 0;
    LAUNCH1D(split_color0_bwd_kernel, n, albedo6, metallic2, spec3, d_colors7, n, d_albedo6, d_metallic2, d_spec3);
    RSDF_RETURN_LAUNCH();
}

int rsdf_split_shade1_fwd(const float *albedo6, const float *roughness, const float *metallic2,
                          const float *spec3, const float *diffuse_light, const float *specular_light,
                          const float *fg, int64_t n, float *colors24, void *stream)
{
    if (n <= 0) return 0;
    LAUNCH1D(split_shade1_fwd_kernel, n, albedo6, roughness, metallic2, spec3, diffuse_light, specular_light, fg,
             n, colors24);
    RSDF_RETURN_LAUNCH();
}

int rsdf_split_shade1_bwd(const float *albedo6, const float *metallic2, const float *spec3,
                          const float *diffuse_light, const float *specular_light, const float *fg,
                          const float *d_colors24, int64_t n, float *d_albedo6, float *d_roughness,
                          float *d_metallic2, float *d_spec3, float *d_diffuse_light,
                          float *d_specular_light, float *d_fg, void *stream)
{
    if (n <= 0) return 0;
    LAUNCH1D(split_shade1_bwd_kernel, n, albedo6, metallic2, spec3, diffuse_light, specular_light, fg, d_colors24,
             n, d_albedo6, d_roughness, d_metallic2, d_spec3, d_diffuse_light, d_specular_light, d_fg);
    RSDF_RETURN_LAUNCH();
}

int rsdf_rgb_to_srgb_fwd(const float *x, int64_t n, float *y, void *stream)
{
    if (n <= 0) return 0;
    LAUNCH1D(srgb_fwd_kernel, n, x, n, y);
    RSDF_RETURN_LAUNCH();
}

int rsdf_compose_srgb_fwd(const float *comp, const float *bg, const float *opacity, int64_t n, float *y, void *stream)
{
    if (n <= 0) return 0;
    LAUNCH1D(compose_srgb_fwd_kernel, 3 * n, comp, bg, opacity, n, y);
    RSDF_RETURN_LAUNCH();
}

int rsdf_compose_srgb_bwd(const float *comp, const float *bg, const float *opacity, const float *dy, int64_t n,
                          float *d_comp, float *d_opacity, void *stream)
{
    if (n <= 0) return 0;
    LAUNCH1D(compose_srgb_bwd_kernel, n, comp, bg, opacity, dy, n, d_comp, d_opacity);
    RSDF_RETURN_LAUNCH();
}

int rsdf_rgb_to_srgb_bwd(const float *x, const float *dy, int64_t n, float *dx, void *stream)
{
    if (n <= 0) return 0;
    LAUNCH1D(srgb_bwd_kernel, n, x, dy, n, dx);
    RSDF_RETURN_LAUNCH();
}

int rsdf_softplus100_slope_fwd(const float *z, int64_t n, float *h, float *slope, void *stream)
{
    if (n <= 0) return 0;
    LAUNCH1D(softplus100_slope_fwd_kernel, n, z, n, h, slope);
    RSDF_RETURN_LAUNCH();
}

int rsdf_softplus100_slope_bwd(const float *slope, const float *dh, const float *dslope, int64_t n, float *dz, void *stream)
{
    RSDF_CHECK_ARG(dh != nullptr || dslope != nullptr, "softplus100_slope_bwd: both gradients are NULL");
    if (n <= 0) return 0;
    LAUNCH1D(softplus100_slope_bwd_kernel, n, slope, dh, dslope, n, dz);
    RSDF_RETURN_LAUNCH();
}

}  // extern "C"
