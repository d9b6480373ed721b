// C1 / C2: per-ray transmittance compositing -- one wavefront per ray, wave64 scans.
//
// Specification followed (paths relative to the upstream RISE-SDF tree):
//   lib/nerfacc/cuda/csrc/render_weight.cu:86-153           w_i = a_i T_i and its closed-form backward
//   lib/nerfacc/cuda/csrc/render_transmittance.cu:85-145    T_i = prod_{j<i}(1-a_j), backward
//   lib/nerfacc/cuda/csrc/render_transmittance_cub.cu:111-166  (the scan-by-key formulation replaced)
//   lib/nerfacc/vol_rendering.py:174-198                    accumulate_along_rays
//   lib/nerfacc/vol_rendering.py:503-520                    render_visibility
//
// The reference runs one THREAD per ray (serial loop) or a device-wide CUB scan keyed by int64 ray
// ids.  Here a ray's samples are contiguous, so a wavefront walks its ray in 64-sample chunks,
// scanning each chunk across lanes and carrying the running product / suffix sum in a scalar.
#include "common.h"

namespace {

constexpr int THREADS = 256;  // 4 rays per workgroup

__device__ __forceinline__ int64_t ray_of_wave() { return ((int64_t)blockIdx.x * THREADS + threadIdx.x) >> 6; }

// ---- C1 in the REFERENCE'S ORDER OF OPERATIONS ---------------------------------------------------------------------------
// render_weight.cu:86-153 runs one thread per ray: T *= (1 - a) left to right, and in the backward a running total that is
// first summed and then DECREMENTED left to right,
//     accum = sum_j gw_j w_j;   ga_j = (gw_j T - accum) / max(1 - a_j, 1e-10);   accum -= gw_j w_j;   T *= (1 - a_j).
// Once a ray has saturated (late training: inv_s in the hundreds to thousands, alpha exactly 1 at the surface crossing) accum
// should be 0 and IS the rounding residue of that subtraction chain, and max(1 - a_j, 1e-10) multiplies it by up to 1e10: the
// reference's gradient there is a deterministic function of its order of operations.  A suffix scan (rounds 1-4 here) evaluates
// the same closed form without the residue -- and therefore differs from the reference by exactly that term (measured with the
// oracle, 48 x 48 rays, visibility-pruned, inv_s 1808: |d_alpha| up to 9.7e3 where the closed form gives 10; 1.6e-3 of the
// largest table-gradient row).  Parity with the reference means its order: these kernels keep one LANE per ray for the
// arithmetic (the identical sequence of fp32 roundings: bit-exact against oracle/risesdf_oracle.c, no contraction) and the
// whole wave for the memory side -- a wavefront owns 64 consecutive rays, stages 64 samples of each through LDS with
// coalesced 256-byte runs, and every lane then walks its own ray's row.  The dependent chain is ~2 x steps x a few cycles per
// wave and all waves run at once (448 for a 28672-ray chunk): 20-40 us per launch, below the kernels' memory time.
constexpr int SEQ_LD = 65;                       // row stride of a [64 rays][64 samples] LDS tile (conflict-free both ways)

struct SeqRays {
    int base, steps, max_steps;
};
__device__ __forceinline__ SeqRays seq_rays(const int32_t *__restrict__ packed, int64_t n_rays)
{
    const int64_t r = (int64_t)blockIdx.x * 64 + threadIdx.x;
    SeqRays s;
    s.base = r < n_rays ? packed[2 * r] : 0;
    s.steps = r < n_rays ? packed[2 * r + 1] : 0;
    int m = s.steps;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
    s.max_steps = m;
    return s;
}
// tile_k[rr][lane] <- src_k[base_rr + c + lane] for the wave's 64 rays and N arrays at once: ALL 64 N loads are issued before
// the first is consumed (rows past a ray's end are left as they are: never read).  A first version staged with an 8-deep
// unrolled load -> LDS-write loop: eight global-memory round trips per array and 64-sample step, ~30 us per step and wave
// while every wave of the launch runs at once -- 1.6 ms per 28672-ray launch for the backward, 0.5 for the forward.
template <int N>
__device__ __forceinline__ void seq_stage(float *const (&tile)[N], const float *const (&src)[N], const SeqRays &s, int c)
{
    const int lane = threadIdx.x;
    float v[N][64];
#pragma unroll
    for (int rr = 0; rr < 64; ++rr) {
        const int b = __builtin_amdgcn_readlane(s.base, rr), n = __builtin_amdgcn_readlane(s.steps, rr);
        const bool ok = c + lane < n;
#pragma unroll
        for (int k = 0; k < N; ++k) v[k][rr] = ok ? src[k][(int64_t)b + c + lane] : 0.0f;
    }
#pragma unroll
    for (int rr = 0; rr < 64; ++rr)
#pragma unroll
        for (int k = 0; k < N; ++k) tile[k][rr * SEQ_LD + lane] = v[k][rr];
}
__device__ __forceinline__ void seq_unstage(float *__restrict__ dst, const float *tile, const SeqRays &s, int c)
{
    const int lane = threadIdx.x;
#pragma unroll
    for (int rr = 0; rr < 64; ++rr) {
        const int b = __builtin_amdgcn_readlane(s.base, rr), n = __builtin_amdgcn_readlane(s.steps, rr);
        if (c + lane < n) dst[(int64_t)b + c + lane] = tile[rr * SEQ_LD + lane];
    }
}

// split staging: issue the loads of a step (seq_load), write them to LDS later (seq_put) -- the forward prefetches step c + 64
// while it works on step c
template <int N>
__device__ __forceinline__ void seq_load(float (&v)[N][64], const float *const (&src)[N], const SeqRays &s, int c)
{
    const int lane = threadIdx.x;
#pragma unroll
    for (int rr = 0; rr < 64; ++rr) {
        const int b = __builtin_amdgcn_readlane(s.base, rr), n = __builtin_amdgcn_readlane(s.steps, rr);
        const bool ok = c + lane < n;
#pragma unroll
        for (int k = 0; k < N; ++k) v[k][rr] = ok ? src[k][(int64_t)b + c + lane] : 0.0f;
    }
}
template <int N>
__device__ __forceinline__ void seq_put(float *const (&tile)[N], const float (&v)[N][64])
{
    const int lane = threadIdx.x;
#pragma unroll
    for (int rr = 0; rr < 64; ++rr)
#pragma unroll
        for (int k = 0; k < N; ++k) tile[k][rr * SEQ_LD + lane] = v[k][rr];
}

// MODE 0: weights + trans; MODE 1: visibility mask (lib/nerfacc/vol_rendering.py:503-520 on the same sequential T)
template <int MODE>
__global__ void __launch_bounds__(64)
weight_fwd_kernel(const int32_t *__restrict__ packed, const float *__restrict__ alphas,
                  int64_t n_rays, float *__restrict__ weights, float *__restrict__ trans,
                  float eps, float alpha_thre, uint8_t *__restrict__ keep)
{
    __shared__ float s_a[64 * SEQ_LD], s_t[64 * SEQ_LD];
    const SeqRays s = seq_rays(packed, n_rays);
    const int lane = threadIdx.x;
    float T = 1.0f;
    float nxt[1][64];
    seq_load<1>(nxt, {alphas}, s, 0);
    for (int c = 0; c < s.max_steps; c += 64) {
        seq_put<1>({s_a}, nxt);
        __syncthreads();
        if (c + 64 < s.max_steps) seq_load<1>(nxt, {alphas}, s, c + 64);      // in flight while this step's chain runs
        const int n = s.steps - c;
        float av[64];                                            // the lane's own row: the chain runs on registers
#pragma unroll
        for (int j = 0; j < 64; ++j) av[j] = s_a[lane * SEQ_LD + j];
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const float a = av[j];
            if (j < n) {
                if (MODE == 0) {
                    s_a[lane * SEQ_LD + j] = a * T;              // weights[j] = alpha * T      (render_weight.cu:108-111)
                    s_t[lane * SEQ_LD + j] = T;
                } else {
                    bool v = T >= eps;
                    if (alpha_thre > 0.0f) v = v && (a >= alpha_thre);
                    s_t[lane * SEQ_LD + j] = v ? 1.0f : 0.0f;
                }
                T *= (1.0f - a);                                 // T *= (1.f - alpha)
            }
        }
        __syncthreads();
        if (MODE == 0) {
            seq_unstage(weights, s_a, s, c);
            if (trans) seq_unstage(trans, s_t, s, c);
        } else {
#pragma unroll
            for (int rr = 0; rr < 64; ++rr) {
                const int b = __builtin_amdgcn_readlane(s.base, rr), nn = __builtin_amdgcn_readlane(s.steps, rr);
                if (c + lane < nn) keep[(int64_t)b + c + lane] = s_t[rr * SEQ_LD + lane] != 0.0f ? 1 : 0;
            }
        }
        __syncthreads();
    }
}

// render_weight.cu:114-153, statement for statement per lane (``trans`` of the forward is not read: the reference recomputes T).
// FMAD: the reference BINARY is built by nvcc -O3 with its default --fmad=true (lib/nerfacc/cuda/_backend.py:43-44 passes no
// -fmad flag), which contracts each single-use multiply with the add / subtract that consumes it:
//     accum += gw*w            ->  accum = fma(gw, w, accum)
//     gw*T - accum             ->  fma(gw, T, -accum)
//     accum -= gw*w            ->  accum = fma(-gw, w, accum)
// (T *= (1 - a) and w = a*T have no multiply-add pair).  In the saturated regime the value of ``accum`` IS its rounding
// residue, so the two sequences give different d_alpha there; FMAD = true (the default) is the contracted sequence, FMAD =
// false the source's operations one rounding each (RSDF_C1_FMAD=0).  Both are bit-exact against oracle/risesdf_oracle.c's
// orc_weight_from_alpha_bwd(..., fmad).
template <bool FMAD>
__global__ void __launch_bounds__(64)
weight_bwd_kernel(const int32_t *__restrict__ packed, const float *__restrict__ alphas,
                  const float *__restrict__ weights, const float *__restrict__ gw, int64_t n_rays, float *__restrict__ ga)
{
    __shared__ float s_g[64 * SEQ_LD], s_w[64 * SEQ_LD], s_a[64 * SEQ_LD];
    const SeqRays s = seq_rays(packed, n_rays);
    const int lane = threadIdx.x;
    float accum = 0.0f;
    {
        float nxt[2][64];
        seq_load<2>(nxt, {gw, weights}, s, 0);
        for (int c = 0; c < s.max_steps; c += 64) {              // accum += grad_weights[j] * weights[j]
            seq_put<2>({s_g, s_w}, nxt);
            __syncthreads();
            if (c + 64 < s.max_steps) seq_load<2>(nxt, {gw, weights}, s, c + 64);
            const int n = s.steps - c;
            if (FMAD) {
                float g_[64], w_[64];
#pragma unroll
                for (int j = 0; j < 64; ++j) {
                    g_[j] = s_g[lane * SEQ_LD + j];
                    w_[j] = s_w[lane * SEQ_LD + j];
                }
#pragma unroll
                for (int j = 0; j < 64; ++j)
                    if (j < n) accum = __fmaf_rn(g_[j], w_[j], accum);
            } else {
                float pv[64];
#pragma unroll
                for (int j = 0; j < 64; ++j) pv[j] = s_g[lane * SEQ_LD + j] * s_w[lane * SEQ_LD + j];
#pragma unroll
                for (int j = 0; j < 64; ++j)
                    if (j < n) accum += pv[j];
            }
            __syncthreads();
        }
    }
    float T = 1.0f;
    for (int c = 0; c < s.max_steps; c += 64) {
        seq_stage<3>({s_g, s_w, s_a}, {gw, weights, alphas}, s, c);
        __syncthreads();
        const int n = s.steps - c;
        float av[64], gv[64], pv[64];
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            av[j] = s_a[lane * SEQ_LD + j];
            gv[j] = s_g[lane * SEQ_LD + j];
            pv[j] = FMAD ? s_w[lane * SEQ_LD + j] : gv[j] * s_w[lane * SEQ_LD + j];      // FMAD: the weight itself
        }
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            if (j < n) {
                if (FMAD) {
                    s_a[lane * SEQ_LD + j] = __fmaf_rn(gv[j], T, -accum) / fmaxf(1.0f - av[j], 1e-10f);
                    accum = __fmaf_rn(-gv[j], pv[j], accum);
                } else {
                    s_a[lane * SEQ_LD + j] = (gv[j] * T - accum) / fmaxf(1.0f - av[j], 1e-10f);
                    accum -= pv[j];
                }
                T *= (1.0f - av[j]);
            }
        }
        __syncthreads();
        seq_unstage(ga, s_a, s, c);
        __syncthreads();
    }
}

// ga_j = (sum_{k>j} -gT_k T_k) / max(1 - a_j, 1e-10)   (render_transmittance.cu:137-142)
__global__ void __launch_bounds__(THREADS)
trans_bwd_kernel(const int32_t *__restrict__ packed, const float *__restrict__ alphas,
                 const float *__restrict__ trans, const float *__restrict__ gt, int64_t n_rays,
                 float *__restrict__ ga)
{
    const int64_t r = ray_of_wave();
    if (r >= n_rays) return;
    const int base = packed[2 * r], steps = packed[2 * r + 1];
    const int lane = lane_id();
    float carry = 0.0f;
    const int n_chunks = (steps + 63) >> 6;
    for (int ci = n_chunks - 1; ci >= 0; --ci) {
        const int j = ci * 64 + lane;
        const bool ok = j < steps;
        const float v = ok ? -gt[base + j] * trans[base + j] : 0.0f;
        const float suf = wave_suffix_sum(v);  // inclusive of j
        if (ok) ga[base + j] = (suf - v + carry) / fmaxf(1.0f - alphas[base + j], 1e-10f);
        carry += __shfl(suf, 0, 64);
    }
}

// out[r,d] = sum_i w_i v[i,d]
template <int D_TILE>
__global__ void __launch_bounds__(THREADS)
accumulate_fwd_kernel(const int32_t *__restrict__ packed, const float *__restrict__ weights,
                      const float *__restrict__ values, int64_t n_rays, int D,
                      float *__restrict__ out)
{
    const int64_t r = ray_of_wave();
    if (r >= n_rays) return;
    const int base = packed[2 * r], steps = packed[2 * r + 1];
    const int lane = lane_id();
    for (int d0 = 0; d0 < D; d0 += D_TILE) {
        float acc[D_TILE];
#pragma unroll
        for (int d = 0; d < D_TILE; ++d) acc[d] = 0.0f;
        for (int j = lane; j < steps; j += 64) {
            const float w = weights[base + j];
            if (values) {
                const float *v = values + (int64_t)(base + j) * D + d0;
#pragma unroll
                for (int d = 0; d < D_TILE; ++d)
                    if (d0 + d < D) acc[d] = fmaf(w, v[d], acc[d]);
            } else {
                acc[0] += w;
            }
        }
#pragma unroll
        for (int d = 0; d < D_TILE; ++d) {
            const float s = wave_sum(acc[d]);
            if (lane == 0 && d0 + d < D) out[r * D + d0 + d] = s;
        }
    }
}

// gw_i = sum_d go[r,d] v[i,d];  gv[i,d] = w_i go[r,d]
__global__ void __launch_bounds__(THREADS)
accumulate_bwd_kernel(const int32_t *__restrict__ packed, const float *__restrict__ weights,
                      const float *__restrict__ values, const float *__restrict__ go,
                      int64_t n_rays, int D, float *__restrict__ gw, float *__restrict__ gv)
{
    const int64_t r = ray_of_wave();
    if (r >= n_rays) return;
    const int base = packed[2 * r], steps = packed[2 * r + 1];
    const int lane = lane_id();
    const float *g = go + r * D;
    for (int j = lane; j < steps; j += 64) {
        const int64_t s = base + j;
        if (values) {
            const float w = weights[s];
            float acc = 0.0f;
            for (int d = 0; d < D; ++d) {
                const float gd = g[d];
                acc = fmaf(gd, values[s * D + d], acc);
                if (gv) gv[s * D + d] = w * gd;
            }
            if (gw) gw[s] = acc;
        } else if (gw) {
            gw[s] = g[0];
        }
    }
}

// Opacity and depth of a ray in one pass (models/volrend.py:878-885: accumulate_along_rays(weights, None) and
// accumulate_along_rays(weights, (t_starts + t_ends)[..., None] / 2.0)): the midpoint is formed here, as torch forms it (an
// add, then the exact halving), and the two sums run in the kernels' own order above -- bit-identical to the two calls,
// without the midpoint tensor (two elementwise passes over all samples), one of the two launches each way and autograd's
// add of the two weight gradients.
// NORMALS (round 4): the ray's normal map, accumulate_along_rays(weights, normals [S,3]) (models/volrend.py:875-877), in
// the same pass: per channel the same fmaf chain in the same lane order and the same wave reduction as
// accumulate_fwd_kernel<4> runs for D = 3 -- bit-identical to the separate call, one launch less each way.
template <bool NORMALS>
__global__ void __launch_bounds__(THREADS)
opacity_depth_fwd_kernel(const int32_t *__restrict__ packed, const float *__restrict__ weights,
                         const float *__restrict__ ts, const float *__restrict__ te, const float *__restrict__ normals,
                         int64_t n_rays, float *__restrict__ opacity, float *__restrict__ depth,
                         float *__restrict__ normal_map, float *__restrict__ mid_out)
{
    const int64_t r = ray_of_wave();
    if (r >= n_rays) return;
    const int base = packed[2 * r], steps = packed[2 * r + 1];
    const int lane = lane_id();
    float a0 = 0.0f, a1 = 0.0f, nx = 0.0f, ny = 0.0f, nz = 0.0f;
    for (int j = lane; j < steps; j += 64) {
        const float w = weights[base + j];
        const float mid = (ts[base + j] + te[base + j]) / 2.0f;
        if (mid_out) mid_out[base + j] = mid;            // (the training outputs' "points": models/neus.py:303)
        a0 += w;
        a1 = fmaf(w, mid, a1);
        if (NORMALS) {
            const float *n = normals + (int64_t)(base + j) * 3;
            nx = fmaf(w, n[0], nx);
            ny = fmaf(w, n[1], ny);
            nz = fmaf(w, n[2], nz);
        }
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    if (NORMALS) {
        nx = wave_sum(nx);
        ny = wave_sum(ny);
        nz = wave_sum(nz);
    }
    if (lane == 0) {
        opacity[r] = a0;
        depth[r] = a1;
        if (NORMALS) {
            normal_map[3 * r] = nx;
            normal_map[3 * r + 1] = ny;
            normal_map[3 * r + 2] = nz;
        }
    }
}

// gw_i = g_opacity[r] + g_depth[r] * mid_i   (either gradient may be NULL = zero)
// with normals: gw_i += sum_d g_normal[r,d] n[i,d] (the chain accumulate_bwd_kernel runs, added last, as autograd adds the
// two calls' weight gradients); gn[i,d] = w_i g_normal[r,d]
__global__ void __launch_bounds__(THREADS)
opacity_depth_bwd_kernel(const int32_t *__restrict__ packed, const float *__restrict__ ts, const float *__restrict__ te,
                         const float *__restrict__ g_opacity, const float *__restrict__ g_depth, int64_t n_rays,
                         float *__restrict__ gw, const float *__restrict__ weights, const float *__restrict__ normals,
                         const float *__restrict__ g_normal, float *__restrict__ gn)
{
    const int64_t r = ray_of_wave();
    if (r >= n_rays) return;
    const int base = packed[2 * r], steps = packed[2 * r + 1];
    const float go = g_opacity ? g_opacity[r] : 0.0f, gd = g_depth ? g_depth[r] : 0.0f;
    float g3[3] = {0.0f, 0.0f, 0.0f};
    if (g_normal) {
        g3[0] = g_normal[3 * r];
        g3[1] = g_normal[3 * r + 1];
        g3[2] = g_normal[3 * r + 2];
    }
    for (int j = lane_id(); j < steps; j += 64) {
        const int64_t s = base + j;
        float v = 0.0f;
        if (g_opacity || g_depth) {
            const float mid = (ts[s] + te[s]) / 2.0f;
            const float d = gd * mid;
            v = g_opacity ? (g_depth ? go + d : go) : d;
        }
        if (g_normal) {
            float acc = 0.0f;
#pragma unroll
            for (int d = 0; d < 3; ++d) acc = fmaf(g3[d], normals[s * 3 + d], acc);
            v = (g_opacity || g_depth) ? v + acc : acc;
            if (gn) {
                const float w = weights[s];
#pragma unroll
                for (int d = 0; d < 3; ++d) gn[s * 3 + d] = w * g3[d];
            }
        }
        if (gw) gw[s] = v;
    }
}

// ---- channel-parallel forms for 4 <= D <= 64 (the 24-channel maps of the split-sum stage) ------------------------------
// In the kernels above a lane is a sample and walks its D channels: at D = 24 every load / store instruction touches 64
// rows 96 bytes apart (one 4-byte piece of 64 different lines), 24 times per sample tile.  Here a lane is a (sample slot,
// channel) pair -- SPW = 64 / D samples per wavefront iteration, channel = lane % D fixed for the lane -- so the values
// and their gradients move as contiguous runs; the per-sample sum over channels of the backward is a segmented shuffle
// reduction.  One wavefront per ray as before.
__global__ void __launch_bounds__(THREADS)
accumulate_fwd_ch_kernel(const int32_t *__restrict__ packed, const float *__restrict__ weights,
                         const float *__restrict__ values, int64_t n_rays, int D, float *__restrict__ out)
{
    const int64_t r = ray_of_wave();
    if (r >= n_rays) return;
    const int base = packed[2 * r], steps = packed[2 * r + 1];
    const int lane = lane_id();
    const int spw = 64 / D, q = lane / D, d = lane - q * D;
    const bool on = q < spw;
    float acc = 0.0f;
    for (int j = q; j < steps && on; j += spw) {
        const int64_t s = base + j;
        acc = fmaf(weights[s], values[s * D + d], acc);
    }
    // sum over the sample slots: lanes d, d + D, d + 2 D, ...
    for (int o = 1; o < spw; ++o) {
        const float t = __shfl(acc, d + o * D, 64);
        if (q == 0) acc += t;
    }
    if (lane < D) out[r * D + d] = acc;
}

// gw_i = sum_d go[r,d] v[i,d];  gv[i,d] = w_i go[r,d]
__global__ void __launch_bounds__(THREADS)
accumulate_bwd_ch_kernel(const int32_t *__restrict__ packed, const float *__restrict__ weights,
                         const float *__restrict__ values, const float *__restrict__ go, int64_t n_rays, int D,
                         float *__restrict__ gw, float *__restrict__ gv)
{
    const int64_t r = ray_of_wave();
    if (r >= n_rays) return;
    const int base = packed[2 * r], steps = packed[2 * r + 1];
    const int lane = lane_id();
    const int spw = 64 / D, q = lane / D, d = lane - q * D;
    const bool on = q < spw;
    const float gd = on ? go[r * D + d] : 0.0f;
    for (int j0 = 0; j0 < steps; j0 += spw) {      // uniform trip count: the shuffles below need every lane
        const int j = j0 + q;
        const bool live = on && j < steps;
        const int64_t s = base + (live ? j : 0);
        float p = 0.0f;
        if (live) {
            p = gd * values[s * D + d];
            if (gv) gv[s * D + d] = weights[s] * gd;
        }
        if (gw) {
            for (int o = 1; o < D; o <<= 1) {
                const float t = __shfl_down(p, o, 64);
                if (d + o < D) p += t;
            }
            if (live && d == 0) gw[s] = p;
        }
    }
}

}  // namespace

extern "C" {

int rsdf_weight_from_alpha_fwd(const int32_t *packed_info, const float *alphas, int64_t n_rays,
                               float *weights, float *trans, void *stream)
{
    if (n_rays <= 0) return 0;
    weight_fwd_kernel<0><<<rsdf_blocks(n_rays, 64), 64, 0, (hipStream_t)stream>>>(
        packed_info, alphas, n_rays, weights, trans, 0.f, 0.f, nullptr);
    RSDF_RETURN_LAUNCH();
}

int rsdf_visibility_from_alpha(const int32_t *packed_info, const float *alphas, int64_t n_rays,
                               float early_stop_eps, float alpha_thre, uint8_t *keep, void *stream)
{
    if (n_rays <= 0) return 0;
    weight_fwd_kernel<1><<<rsdf_blocks(n_rays, 64), 64, 0, (hipStream_t)stream>>>(
        packed_info, alphas, n_rays, nullptr, nullptr, early_stop_eps, alpha_thre, keep);
    RSDF_RETURN_LAUNCH();
}

int rsdf_weight_from_alpha_bwd(const int32_t *packed_info, const float *alphas,
                               const float *weights, const float *trans,
                               const float *grad_weights, int64_t n_rays, float *grad_alphas,
                               void *stream)
{
    if (n_rays <= 0) return 0;
    (void)trans;          // (the reference's backward recomputes T in its own order, render_weight.cu:141-151)
    return rsdf_weight_from_alpha_bwd_seq(packed_info, alphas, weights, grad_weights, n_rays, 1, grad_alphas, stream);
}

int rsdf_weight_from_alpha_bwd_seq(const int32_t *packed_info, const float *alphas, const float *weights,
                                   const float *grad_weights, int64_t n_rays, int fmad, float *grad_alphas, void *stream)
{
    if (n_rays <= 0) return 0;
    if (fmad)
        weight_bwd_kernel<true><<<rsdf_blocks(n_rays, 64), 64, 0, (hipStream_t)stream>>>(
            packed_info, alphas, weights, grad_weights, n_rays, grad_alphas);
    else
        weight_bwd_kernel<false><<<rsdf_blocks(n_rays, 64), 64, 0, (hipStream_t)stream>>>(
            packed_info, alphas, weights, grad_weights, n_rays, grad_alphas);
    RSDF_RETURN_LAUNCH();
}

int rsdf_transmittance_from_alpha_bwd(const int32_t *packed_info, const float *alphas,
                                      const float *trans, const float *grad_trans, int64_t n_rays,
                                      float *grad_alphas, void *stream)
{
    if (n_rays <= 0) return 0;
    trans_bwd_kernel<<<rsdf_blocks(n_rays * 64, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        packed_info, alphas, trans, grad_trans, n_rays, grad_alphas);
    RSDF_RETURN_LAUNCH();
}

int rsdf_accumulate_fwd(const int32_t *packed_info, const float *weights, const float *values,
                        int64_t n_rays, int D, float *out, void *stream)
{
    RSDF_CHECK_ARG(D >= 1, "accumulate_fwd: D must be >= 1");
    RSDF_CHECK_ARG(values || D == 1, "accumulate_fwd: values == NULL requires D == 1");
    if (n_rays <= 0) return 0;
    const unsigned grid = rsdf_blocks(n_rays * 64, THREADS);
    hipStream_t st = (hipStream_t)stream;
    if (values != nullptr && D >= 4 && D <= 64)
        accumulate_fwd_ch_kernel<<<grid, THREADS, 0, st>>>(packed_info, weights, values, n_rays, D, out);
    else if (D <= 1) accumulate_fwd_kernel<1><<<grid, THREADS, 0, st>>>(packed_info, weights, values, n_rays, D, out);
    else if (D <= 4) accumulate_fwd_kernel<4><<<grid, THREADS, 0, st>>>(packed_info, weights, values, n_rays, D, out);
    else accumulate_fwd_kernel<8><<<grid, THREADS, 0, st>>>(packed_info, weights, values, n_rays, D, out);
    RSDF_RETURN_LAUNCH();
}

int rsdf_opacity_depth_fwd(const int32_t *packed_info, const float *weights, const float *t_starts, const float *t_ends,
                           int64_t n_rays, float *opacity, float *depth, float *midpoints, void *stream)
{
    if (n_rays <= 0) return 0;
    opacity_depth_fwd_kernel<false><<<rsdf_blocks(n_rays * 64, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        packed_info, weights, t_starts, t_ends, nullptr, n_rays, opacity, depth, nullptr, midpoints);
    RSDF_RETURN_LAUNCH();
}

int rsdf_opacity_depth_bwd(const int32_t *packed_info, const float *t_starts, const float *t_ends, const float *grad_opacity,
                           const float *grad_depth, int64_t n_rays, float *grad_weights, void *stream)
{
    RSDF_CHECK_ARG(grad_opacity != nullptr || grad_depth != nullptr, "opacity_depth_bwd: both gradients are NULL");
    if (n_rays <= 0) return 0;
    opacity_depth_bwd_kernel<<<rsdf_blocks(n_rays * 64, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        packed_info, t_starts, t_ends, grad_opacity, grad_depth, n_rays, grad_weights, nullptr, nullptr, nullptr, nullptr);
    RSDF_RETURN_LAUNCH();
}

int rsdf_opacity_depth_normal_fwd(const int32_t *packed_info, const float *weights, const float *t_starts,
                                  const float *t_ends, const float *normals, int64_t n_rays, float *opacity, float *depth,
                                  float *normal_map, float *midpoints, void *stream)
{
    RSDF_CHECK_ARG(normals != nullptr && normal_map != nullptr, "opacity_depth_normal_fwd: normals / normal_map are NULL");
    if (n_rays <= 0) return 0;
    opacity_depth_fwd_kernel<true><<<rsdf_blocks(n_rays * 64, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        packed_info, weights, t_starts, t_ends, normals, n_rays, opacity, depth, normal_map, midpoints);
    RSDF_RETURN_LAUNCH();
}

int rsdf_opacity_depth_normal_bwd(const int32_t *packed_info, const float *weights, const float *t_starts,
                                  const float *t_ends, const float *normals, const float *grad_opacity,
                                  const float *grad_depth, const float *grad_normal_map, int64_t n_rays,
                                  float *grad_weights, float *grad_normals, void *stream)
{
    RSDF_CHECK_ARG(grad_opacity != nullptr || grad_depth != nullptr || grad_normal_map != nullptr,
                   "opacity_depth_normal_bwd: all three gradients are NULL");
    RSDF_CHECK_ARG(grad_normal_map == nullptr || (normals != nullptr && weights != nullptr),
                   "opacity_depth_normal_bwd: the normal gradient needs weights and normals");
    RSDF_CHECK_ARG(grad_weights != nullptr || grad_normals != nullptr, "opacity_depth_normal_bwd: no output requested");
    if (n_rays <= 0) return 0;
    opacity_depth_bwd_kernel<<<rsdf_blocks(n_rays * 64, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        packed_info, t_starts, t_ends, grad_opacity, grad_depth, n_rays, grad_weights, weights, normals, grad_normal_map,
        grad_normals);
    RSDF_RETURN_LAUNCH();
}

int rsdf_accumulate_bwd(const int32_t *packed_info, const float *weights, const float *values,
                        const float *grad_out, int64_t n_rays, int D, float *grad_weights,
                        float *grad_values, void *stream)
{
    RSDF_CHECK_ARG(D >= 1, "accumulate_bwd: D must be >= 1");
    if (n_rays <= 0) return 0;
    if (values != nullptr && D >= 4 && D <= 64)
        accumulate_bwd_ch_kernel<<<rsdf_blocks(n_rays * 64, THREADS), THREADS, 0, (hipStream_t)stream>>>(
            packed_info, weights, values, grad_out, n_rays, D, grad_weights, grad_values);
    else
        accumulate_bwd_kernel<<<rsdf_blocks(n_rays * 64, THREADS), THREADS, 0, (hipStream_t)stream>>>(
            packed_info, weights, values, grad_out, n_rays, D, grad_weights, grad_values);
    RSDF_RETURN_LAUNCH();
}

}  // extern "C"
