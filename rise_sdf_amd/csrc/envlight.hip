// S4 / E1: environment light -- cube-map prefilters (diffuse, GGX specular, mip chain) and cube texture
// lookups (bilinear, explicit-mip trilinear) with their backward passes.
//
// Specification followed (paths relative to the upstream RISE-SDF tree):
//   lib/renderutils/c_src/cubemap.cu:17-46      pixel_area, cube_to_dir (face orientation)
//   lib/renderutils/c_src/cubemap.cu:110-169    DiffuseCubemapFwd/Bwd
//   lib/renderutils/c_src/cubemap.cu:181-244    SpecularBoundsKernel (per-texel, per-face bbox of the lobe
//                                               cone, 16x16-tile interval culling)
//   lib/renderutils/c_src/cubemap.cu:246-350    SpecularCubemapFwd/Bwd (+ the /wsum of lib/renderutils/ops.py:458)
//   lib/pbr/utils/light_utils.py:94-109         cubemap_mip (2x2 average; backward = cube-linear lookup of dout/4)
//   lib/pbr/light.py:188-206                    eval_mip: dr.texture(..., boundary_mode='cube'), linear or
//                                               linear-mipmap-linear with mip_level_bias
// nvdiffrast is absent upstream: the cube lookup is this build's definition (oracle/envlight.py docstring).
//
// MI355X notes.  The reference's backward prefilters scatter with atomicAdd from every output texel.  The
// prefilter weights depend on (L.V) and on the INPUT texel's solid angle only, so the backward is written
// as a gather over the same lobe window (the cone around L is the set of outputs that saw L): no atomics,
// deterministic, same traffic as the forward.  Per-texel directions are recomputed (one rsqrt) instead of
// fetched; the separable solid-angle factor comes from a small LDS table.
#include "common.h"

namespace {

constexpr int THREADS = 256;

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 normalize3(V3 a)
{
    const float l = sqrtf(fmaxf(dot3(a, a), 1e-20f));
    return v3(a.x / l, a.y / l, a.z / l);
}
__device__ __forceinline__ V3 face_vec(int s, float fx, float fy)
{
    switch (s) {
    case 0: return v3(1.f, -fy, -fx);
    case 1: return v3(-1.f, -fy, fx);
    case 2: return v3(fx, 1.f, fy);
    case 3: return v3(fx, -1.f, -fy);
    case 4: return v3(fx, -fy, 1.f);
    default: return v3(-fx, -fy, -1.f);
    }
}
__device__ __forceinline__ V3 cube_to_dir(int x, int y, int s, int N)
{
    const float fx = 2.0f * (((float)x + 0.5f) / (float)N) - 1.0f;
    const float fy = 2.0f * (((float)y + 0.5f) / (float)N) - 1.0f;
    return normalize3(face_vec(s, fx, fy));
}
// separable factor of pixel_area: area(x,y) = side(x) * side(y)
__device__ __forceinline__ float area_side(int i, int N)
{
    if (N <= 1) return 1.0f;
    const int H = N / 2;
    const int a = abs(i - H);
    // atan((a+1)/H) - atan(a/H) (cubemap.cu:24) as one atan: no cancellation at large H
    return atanf((float)H / (float)(H * H + a * (a + 1)));
}

// ------------------------------------------------------------------------------------------------
// diffuse (R is small: 16)
// ------------------------------------------------------------------------------------------------
template <bool BWD>
__global__ void __launch_bounds__(THREADS)
diffuse_kernel(const float *__restrict__ src, int R, float *__restrict__ dst)
{
    extern __shared__ float s_side[];  // [R]
    for (int i = threadIdx.x; i < R; i += THREADS) s_side[i] = area_side(i, R);
    __syncthreads();
    // one wavefront per output texel; the lanes stride over the 6 R^2 input texels and reduce (R is 16: a thread
    // per texel would be 1536 threads of 1536 serial iterations)
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
    if (idx >= 6 * R * R) return;
    const int s0 = idx / (R * R), y0 = (idx / R) % R, x0 = idx % R;
    const V3 A = cube_to_dir(x0, y0, s0, R);
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    for (int i = lane; i < 6 * R * R; i += 64) {
        const int s = i / (R * R), y = (i / R) % R, x = i % R;
        const V3 B = cube_to_dir(x, y, s, R);
        const float ct = fminf(fmaxf(dot3(A, B), 0.0f), 0.999f);
        // forward: weight uses the INPUT texel's area; backward (gather at the input texel A): the
        // area factor is A's own and is applied after the loop
        const float w = BWD ? ct : ct * s_side[x] * s_side[y] / 3.141592f;
        const float *p = src + i * 3;
        c0 += p[0] * w; c1 += p[1] * w; c2 += p[2] * w;
    }
    c0 = wave_sum(c0); c1 = wave_sum(c1); c2 = wave_sum(c2);
    if (lane != 0) return;
    if (BWD) {
        const float a = s_side[x0] * s_side[y0] / 3.141592f;
        c0 *= a; c1 *= a; c2 *= a;
    }
    float *o = dst + idx * 3;
    o[0] = c0; o[1] = c1; o[2] = c2;
}

// ------------------------------------------------------------------------------------------------
// specular bounds
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(THREADS)
bounds_kernel(int R, float cos_cutoff, float *__restrict__ bounds)
{
    const int idx = blockIdx.x * THREADS + threadIdx.x;
    if (idx >= 6 * R * R) return;
    const int pz = idx / (R * R), py = (idx / R) % R, px = idx % R;
    const V3 V = cube_to_dir(px, py, pz, R);
    constexpr int TILE = 16;
    const int nt = (R + TILE - 1) / TILE;
    for (int s = 0; s < 6; ++s) {
        int mnx = R - 1, mxx = 0, mny = R - 1, mxy = 0;
        for (int tx = 0; tx < nt; ++tx)
            for (int ty = 0; ty < nt; ++ty) {
                const int tsx = tx * TILE, tsy = ty * TILE;
                const int tex = min((tx + 1) * TILE, R), tey = min((ty + 1) * TILE, R);
                const V3 L0 = cube_to_dir(tsx, tsy, s, R), L1 = cube_to_dir(tex, tsy, s, R);
                const V3 L2 = cube_to_dir(tsx, tey, s, R), L3 = cube_to_dir(tex, tey, s, R);
                const float minx = fminf(fminf(L0.x, L1.x), fminf(L2.x, L3.x)), maxx = fmaxf(fmaxf(L0.x, L1.x), fmaxf(L2.x, L3.x));
                const float miny = fminf(fminf(L0.y, L1.y), fminf(L2.y, L3.y)), maxy = fmaxf(fmaxf(L0.y, L1.y), fmaxf(L2.y, L3.y));
                const float minz = fminf(fminf(L0.z, L1.z), fminf(L2.z, L3.z)), maxz = fmaxf(fmaxf(L0.z, L1.z), fmaxf(L2.z, L3.z));
                const float maxdp = fmaxf(minx * V.x, maxx * V.x) + fmaxf(miny * V.y, maxy * V.y) +
                                    fmaxf(minz * V.z, maxz * V.z);
                if (maxdp >= cos_cutoff) {
                    for (int y = tsy; y < tey; ++y)
                        for (int x = tsx; x < tex; ++x)
                            if (dot3(cube_to_dir(x, y, s, R), V) >= cos_cutoff) {
                                mnx = min(mnx, x); mxx = max(mxx, x);
                                mny = min(mny, y); mxy = max(mxy, y);
                            }
                }
            }
        float *o = bounds + (size_t)idx * 24 + s * 4;
        o[0] = (float)mnx; o[1] = (float)mxx; o[2] = (float)mny; o[3] = (float)mxy;
    }
}

// GGX NDF at cos = V.H with H = normalize(A + B), A and B unit (bsdf.h ndf_ggx: a2 / (pi d^2), d = (c a2 - c) c + 1).
// d = 1 - c^2 (1 - a2) cancels catastrophically in fp32 for narrow lobes (a2 ~ 4e-5 at roughness 0.08), so it is
// evaluated as d = sin^2 (1 - a2) + a2 with sin^2 = sin^2(angle(A, B) / 2) = |A - B|^2 / 4: the same function, no
// cancellation (A - B is exact for nearby unit vectors).  The window loops are bound by the vector instructions of this
// function (per-level times are the same whether a source texel comes from L2 per output texel or from an LDS tile
// shared by 64 outputs: measured), so it is 9 instructions with one reciprocal (v_rcp_f32, 1 ulp); the first form,
// |A x B|^2 / |A + B|^2 with two IEEE divisions, cost 35 per pair: 4.6 -> 3.6 (one division) -> 2.x ms per prefilter.
__device__ __forceinline__ float ndf_ggx_pair(float a2, V3 A, V3 B)
{
    const V3 e = v3(A.x - B.x, A.y - B.y, A.z - B.z);
    const float s2 = 0.25f * dot3(e, e);
    const float d = fmaf(s2, 1.0f - a2, a2);
    return a2 * __builtin_amdgcn_rcpf(d * d * 3.14159265358979323846f);
}

// forward: out4 = [sum w c, sum w];  backward (gather): dcube[L] = area(L)/4 * sum_V g[V] (L.V) D(V.H)
// G lanes per output texel (launch_specular): 1 = a thread walks its window (large maps: millions of threads); 64 = one
// wavefront per texel, the lanes stride over the window and reduce -- the coarse levels have few texels (6144 at 32^2) but
// windows of thousands, which a thread per texel leaves latency bound.
//
// The window loops are bound by the vector instructions per (output texel, source texel) pair (measured: the same per-level
// times with the source texels in an LDS tile).  One pair, with the cached texel table (direction + solid angle / 4 in one
// 16-byte load): L.V, |V - L|^2 as 3 sub + mul + 2 fma, d = fma(|V - L|^2, (1 - a2) / 4, a2), one v_rcp_f32 of d^2 pi / a2, a
// select on the cutoff and 4 accumulating fma -- the fused forms are written out because this file is built with
// -ffp-contract=off -- and 32-bit texel indices instead of 64-bit multiply-adds per load.  The pair is branch-free (a texel
// outside the cone gets weight 0; cc = max(cos_cutoff, 0): a texel with L.V in [cos_cutoff, 0) has weight max(L.V, 0) = 0
// anyway), so a loop trip holds the four loads of TWO pairs in flight.  L.V is formed exactly as bounds_kernel forms it
// (unfused, left to right), so the set of texels with a non-zero weight is the reference's.  Round 3: 512^2 1.08 -> 0.68
// ms, 256^2 1.16 -> 1.03, 128^2 0.46 -> 0.42 per prefilter.  Measured and dropped: the two texels of a trip on packed fp32
// instructions from a per-texel-pair table (v_pk_fma_f32 & co: 0.73 / 1.20 / 0.50 ms -- no faster than two plain
// instructions, plus the even-x alignment of the window); 16 lanes per texel (1.10 / 0.40 at 256^2 / 128^2).
struct PairAcc { float c0, c1, c2, wsum; };
// -DSPEC_DIAG=1: no source-texel loads; 2: no table loads; 3: neither (A/B diagnostics of what bounds the window loops:
// tools/bench_prefilter.py under tools/ab_unit.sh; results are wrong by construction)
#ifndef SPEC_DIAG
#define SPEC_DIAG 0
#endif
__device__ __forceinline__ float4 spec_ld_table(const float4 *p, const V3 &A, int i)
{
    if (SPEC_DIAG & 2) return make_float4(A.x, A.y - 1e-3f * (float)(i & 15), A.z, 1e-6f);
    return *p;
}
__device__ __forceinline__ float spec_ld_src(const float *p, int k, int i)
{
    if (SPEC_DIAG & 1) return 0.25f * (float)(k + 1) + (float)(i & 1);
    return p[k];
}

template <bool BWD>
__device__ __forceinline__ void spec_pair(PairAcc &acc, const V3 &A, float cc, float k1, float a2, float cpi,
                                          const float4 t, float p0, float p1, float p2)
{
#ifdef SPEC_UNFUSED_DOT
    const float d = t.x * A.x + t.y * A.y + t.z * A.z;
#else
    // (round 6: two fma -- the reference's dot(L, VNR) is built by nvcc with its default contraction too; a texel within an ulp
    // of the cutoff may fall on the other side than bounds_kernel's unfused form puts it: the tests bracket exactly that)
    const float d = fmaf(t.z, A.z, fmaf(t.y, A.y, t.x * A.x));
#endif
    const float ex = A.x - t.x, ey = A.y - t.y, ez = A.z - t.z;
    const float e2 = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
    const float dd = fmaf(e2, k1, a2);                                  // sin^2 (1 - a2) + a2, sin^2 = |A - B|^2 / 4
    const float geom = d * __builtin_amdgcn_rcpf(dd * dd * cpi);        // (L.V) a2 / (pi d^2)
    const float w = d < cc ? 0.0f : (BWD ? geom : geom * t.w);
    acc.c0 = fmaf(p0, w, acc.c0);
    acc.c1 = fmaf(p1, w, acc.c1);
    acc.c2 = fmaf(p2, w, acc.c2);
    acc.wsum += w;
}

template <bool BWD, int G>     // G lanes per output texel: 1 or 64
__global__ void __launch_bounds__(THREADS)
specular_kernel(const float *__restrict__ src, int src_ch, const float *__restrict__ bounds,
                const float4 *__restrict__ table, int R, float roughness, float cos_cutoff,
                float *__restrict__ dst, float *__restrict__ wsum_out)
{
    // table (nullable) = per texel (unit direction, solid angle / 4) from texel_table_kernel: one 16-byte load
    // replaces the direction normalisation and the two area factors of every window iteration
    extern __shared__ float s_side[];
    if (table == nullptr) {   // (with the table no workgroup pays for R atanf before it starts)
        for (int i = threadIdx.x; i < R; i += THREADS) s_side[i] = area_side(i, R);
        __syncthreads();
    }
    const int sub = threadIdx.x & (G - 1);
    const int idx = blockIdx.x * (THREADS / G) + threadIdx.x / G;
    if (idx >= 6 * R * R) return;      // (uniform over the G lanes of a texel)
    const int s0 = idx / (R * R), y0 = (idx / R) % R, x0 = idx % R;
    const V3 A = cube_to_dir(x0, y0, s0, R);
    const float alpha = roughness * roughness, a2 = alpha * alpha;
    const float k1 = 0.25f * (1.0f - a2), cpi = 3.14159265358979323846f / a2, cc = fmaxf(cos_cutoff, 0.0f);
    PairAcc acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < 6; ++s) {
        const float4 bb = *reinterpret_cast<const float4 *>(bounds + (size_t)idx * 24 + s * 4);
        const int xmin = (int)bb.x, xmax = (int)bb.y, ymin = (int)bb.z, ymax = (int)bb.w;
        if (xmin > xmax) continue;
        const int bw = xmax - xmin + 1, bh = ymax - ymin + 1;
        if (table != nullptr) {
            if (G == 1) {
                for (int yo = 0; yo < bh; ++yo) {
                    const int row = (s * R + ymin + yo) * R + xmin;       // < 6 * 4096^2 < 2^31
                    const float4 *tr = table + row;
                    const float *pr = src + (size_t)row * src_ch;
                    int xo = 0;
                    for (; xo + 1 < bw; xo += 2) {                        // two pairs per trip: four loads in flight
                        const float4 t0 = spec_ld_table(tr + xo, A, xo), t1 = spec_ld_table(tr + xo + 1, A, xo + 1);
                        const float *q0 = pr + xo * src_ch, *q1 = q0 + src_ch;
                        const float u0 = spec_ld_src(q0, 0, xo), u1 = spec_ld_src(q0, 1, xo), u2 = spec_ld_src(q0, 2, xo);
                        const float v0 = spec_ld_src(q1, 0, xo), v1 = spec_ld_src(q1, 1, xo), v2 = spec_ld_src(q1, 2, xo);
                        spec_pair<BWD>(acc, A, cc, k1, a2, cpi, t0, u0, u1, u2);
                        spec_pair<BWD>(acc, A, cc, k1, a2, cpi, t1, v0, v1, v2);
                    }
                    if (xo < bw) {
                        const float *q0 = pr + xo * src_ch;
                        spec_pair<BWD>(acc, A, cc, k1, a2, cpi, tr[xo], q0[0], q0[1], q0[2]);
                    }
                }
            } else {
#ifndef SPEC_G64_BLOCKS
                // i / bw without an integer division per pair: (i + 0.5) / bw is at least 0.5 / bw away from an integer and
                // i < 2^18, so the float product truncates to the exact quotient.
                const float inv_bw = 1.0f / (float)bw;
                const int cnt = bw * bh, base = (s * R + ymin) * R + xmin, wrap = R - bw;
                auto texel = [&](int i) {
                    const int yo = (int)(((float)i + 0.5f) * inv_bw);
                    return base + i + (int)__mul24(yo, wrap);            // base + yo R + (i - yo bw)
                };
                int i = sub;
                for (; i + G < cnt; i += 2 * G) {                         // two pairs per trip: four loads in flight
                    const int ta = texel(i), tb = texel(i + G);
                    const float4 t0 = spec_ld_table(table + ta, A, ta), t1 = spec_ld_table(table + tb, A, tb);
                    const float *q0 = src + (size_t)ta * src_ch, *q1 = src + (size_t)tb * src_ch;
                    const float u0 = spec_ld_src(q0, 0, ta), u1 = spec_ld_src(q0, 1, ta), u2 = spec_ld_src(q0, 2, ta);
                    const float v0 = spec_ld_src(q1, 0, tb), v1 = spec_ld_src(q1, 1, tb), v2 = spec_ld_src(q1, 2, tb);
                    spec_pair<BWD>(acc, A, cc, k1, a2, cpi, t0, u0, u1, u2);
                    spec_pair<BWD>(acc, A, cc, k1, a2, cpi, t1, v0, v1, v2);
                }
                if (i < cnt) {
                    const int ta = texel(i);
                    const float *q0 = src + (size_t)ta * src_ch;
                    spec_pair<BWD>(acc, A, cc, k1, a2, cpi, table[ta], q0[0], q0[1], q0[2]);
                }
#else
                // Round 6 experiment (-DSPEC_G64_BLOCKS), NOT the default: the wave's 64 lanes walk the window as an 8 x 8 block
                // (lane = (ly, lx)), two blocks per trip -- 28 instead of ~46 vector instructions per pair (no reciprocal
                // quotient, no mul24, 32-bit offsets), and SLOWER: 2.26 + 2.12 ms against 2.14 + 2.00 ms for the five levels
                // (tools/bench_prefilter.py, profiles/r06*/spec_ab.log).  With the diagnostics of the same file (no loads at
                // all: 1.57 + 1.38 ms) this says the window loops are bound by neither instruction count nor bytes but by the
                // short dependent chains of a 27-pair-per-lane loop between wave reductions.
                const int lx = sub & 7, ly = sub >> 3;
                const unsigned tb_bytes = 16u, sb_bytes = 4u * (unsigned)src_ch;
                const char *tbase = reinterpret_cast<const char *>(table);
                const char *sbase = reinterpret_cast<const char *>(src);
                for (int y0 = 0; y0 < bh; y0 += 8) {
                    const int yy = y0 + ly;
                    const bool yok = yy < bh;
                    const int rb = (s * R + ymin + min(yy, bh - 1)) * R + xmin;           // < 6 * 4096^2 < 2^31
                    int x0 = 0;
                    for (; x0 + 8 < bw; x0 += 16) {
                        const int xa = x0 + lx, xb = xa + 8;
                        const unsigned ta = (unsigned)(rb + min(xa, bw - 1)), tb = (unsigned)(rb + min(xb, bw - 1));
                        const float4 t0 = spec_ld_table(reinterpret_cast<const float4 *>(tbase + (size_t)(ta * tb_bytes)), A, (int)ta);
                        const float4 t1 = spec_ld_table(reinterpret_cast<const float4 *>(tbase + (size_t)(tb * tb_bytes)), A, (int)tb);
                        const float *q0 = reinterpret_cast<const float *>(sbase + (size_t)(ta * sb_bytes));
                        const float *q1 = reinterpret_cast<const float *>(sbase + (size_t)(tb * sb_bytes));
                        const float u0 = spec_ld_src(q0, 0, (int)ta), u1 = spec_ld_src(q0, 1, (int)ta), u2 = spec_ld_src(q0, 2, (int)ta);
                        const float v0 = spec_ld_src(q1, 0, (int)tb), v1 = spec_ld_src(q1, 1, (int)tb), v2 = spec_ld_src(q1, 2, (int)tb);
                        spec_pair<BWD>(acc, A, (yok && xa < bw) ? cc : 2.0f, k1, a2, cpi, t0, u0, u1, u2);
                        spec_pair<BWD>(acc, A, (yok && xb < bw) ? cc : 2.0f, k1, a2, cpi, t1, v0, v1, v2);
                    }
                    if (x0 < bw) {
                        const int xa = x0 + lx;
                        const unsigned ta = (unsigned)(rb + min(xa, bw - 1));
                        const float4 t0 = spec_ld_table(reinterpret_cast<const float4 *>(tbase + (size_t)(ta * tb_bytes)), A, (int)ta);
                        const float *q0 = reinterpret_cast<const float *>(sbase + (size_t)(ta * sb_bytes));
                        const float u0 = spec_ld_src(q0, 0, (int)ta), u1 = spec_ld_src(q0, 1, (int)ta), u2 = spec_ld_src(q0, 2, (int)ta);
                        spec_pair<BWD>(acc, A, (yok && xa < bw) ? cc : 2.0f, k1, a2, cpi, t0, u0, u1, u2);
                    }
                }
#endif
            }
        } else {   // no cached table: direction and solid angle per pair (API completeness; the mirrors always pass one)
            const float inv_bw = 1.0f / (float)bw;
            const int cnt = bw * bh;
            for (int i = sub; i < cnt; i += G) {
                const int yo = (int)(((float)i + 0.5f) * inv_bw);   // exact: (i + 0.5) / bw is >= 0.5 / bw from an integer
                const int y = ymin + yo, x = xmin + (i - yo * bw);
                const V3 B = cube_to_dir(x, y, s, R);
                const float4 t = make_float4(B.x, B.y, B.z, s_side[x] * s_side[y] / 4.0f);
                const float *q0 = src + (size_t)((s * R + y) * R + x) * src_ch;
                spec_pair<BWD>(acc, A, cc, k1, a2, cpi, t, q0[0], q0[1], q0[2]);
            }
        }
    }
    float c0 = acc.c0, c1 = acc.c1, c2 = acc.c2, wsum = acc.wsum;
    if (G > 1) {
        if (G == 64) {
            c0 = wave_sum(c0); c1 = wave_sum(c1); c2 = wave_sum(c2); wsum = wave_sum(wsum);
        } else {                                   // G lanes of a wave share a texel: xor-shuffles inside the group
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) {
                c0 += __shfl_xor(c0, o, 64); c1 += __shfl_xor(c1, o, 64); c2 += __shfl_xor(c2, o, 64); wsum += __shfl_xor(wsum, o, 64);
            }
        }
        if (sub != 0) return;
    }
    if (BWD) {
        const float a = table != nullptr ? table[idx].w : s_side[x0] * s_side[y0] / 4.0f;
        float *o = dst + (size_t)idx * 3;
        o[0] = c0 * a; o[1] = c1 * a; o[2] = c2 * a;
    } else if (wsum_out != nullptr) {
        // lib/renderutils/ops.py:458 (out[..., 0:3] / out[..., 3:]) here: contiguous [6,R,R,3] + the weight sums for the backward
        float *o = dst + (size_t)idx * 3;
        o[0] = c0 / wsum; o[1] = c1 / wsum; o[2] = c2 / wsum;
        wsum_out[idx] = wsum;
    } else {
        float *o = dst + (size_t)idx * 4;
        o[0] = c0; o[1] = c1; o[2] = c2; o[3] = wsum;
    }
}

// lanes per output texel by map size: a thread per texel for the large maps (millions of threads), a wavefront per texel
// up to 256^2 (few texels, windows of thousands; measured per level at the yaml's 512^2 light: a thread per texel takes
// 1.9 instead of 1.0 ms at 256^2)
#ifndef SPEC_G64_MAX_R
#define SPEC_G64_MAX_R 256
#endif
template <bool BWD>
void launch_specular(const float *src, int src_ch, const float *bounds, const float4 *table, int R, float roughness,
                     float cos_cutoff, float *dst, hipStream_t st, float *wsum_out = nullptr)
{
    const size_t lds = R * sizeof(float);
    const int64_t n = (int64_t)6 * R * R;
#ifndef SPEC_G
#define SPEC_G 64      // lanes per output texel up to SPEC_G64_MAX_R (A/B: 32, 16)
#endif
    if (R <= SPEC_G64_MAX_R)
        specular_kernel<BWD, SPEC_G><<<rsdf_blocks(n, THREADS / SPEC_G), THREADS, lds, st>>>(src, src_ch, bounds, table, R, roughness, cos_cutoff, dst, wsum_out);
    else
        specular_kernel<BWD, 1><<<rsdf_blocks(n, THREADS), THREADS, lds, st>>>(src, src_ch, bounds, table, R, roughness, cos_cutoff, dst, wsum_out);
}

__global__ void __launch_bounds__(THREADS)
texel_table_kernel(int R, float4 *__restrict__ table)
{
    const int idx = blockIdx.x * THREADS + threadIdx.x;
    if (idx >= 6 * R * R) return;
    const int s = idx / (R * R), y = (idx / R) % R, x = idx % R;
    const V3 d = cube_to_dir(x, y, s, R);
    table[idx] = make_float4(d.x, d.y, d.z, area_side(x, R) * area_side(y, R) / 4.0f);
}

// ------------------------------------------------------------------------------------------------
// 2x2 average pool of a [6,R,R,C] cube map
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(THREADS)
avgpool_kernel(const float *__restrict__ src, int R, int C, float *__restrict__ dst)
{
    const int Rh = R / 2;
    const int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (idx >= (int64_t)6 * Rh * Rh * C) return;
    const int c = idx % C;
    const int x = (idx / C) % Rh, y = (idx / ((int64_t)C * Rh)) % Rh, s = idx / ((int64_t)C * Rh * Rh);
    const float *p = src + ((size_t)(s * R + 2 * y) * R + 2 * x) * C + c;
    dst[idx] = (p[0] + p[C] + p[(size_t)R * C] + p[(size_t)R * C + C]) * 0.25f;
}

// ------------------------------------------------------------------------------------------------
// cube texture lookup
// ------------------------------------------------------------------------------------------------
struct FaceUV { int face; float fx, fy, ma; };

__device__ __forceinline__ FaceUV dir_to_face(float x, float y, float z)
{
    const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
    FaceUV r;
    if (ax >= ay && ax >= az) {
        r.face = x > 0.f ? 0 : 1; r.ma = fmaxf(ax, 1e-30f);
        r.fx = (x > 0.f ? -z : z) / r.ma; r.fy = -y / r.ma;
    } else if (ay >= az) {
        r.face = y > 0.f ? 2 : 3; r.ma = fmaxf(ay, 1e-30f);
        r.fx = x / r.ma; r.fy = (y > 0.f ? z : -z) / r.ma;
    } else {
        r.face = z > 0.f ? 4 : 5; r.ma = fmaxf(az, 1e-30f);
        r.fx = (z > 0.f ? x : -x) / r.ma; r.fy = -y / r.ma;
    }
    return r;
}

// resolve a (possibly out-of-face) texel to its storage offset (in texels) inside a [6,R,R] image
__device__ __forceinline__ int texel_offset(int face, int xi, int yi, int R)
{
    if (xi >= 0 && xi < R && yi >= 0 && yi < R) return (face * R + yi) * R + xi;
    const float fx = 2.0f * (((float)xi + 0.5f) / (float)R) - 1.0f;
    const float fy = 2.0f * (((float)yi + 0.5f) / (float)R) - 1.0f;
    const V3 d = face_vec(face, fx, fy);
    const FaceUV f2 = dir_to_face(d.x, d.y, d.z);
    const int x2 = min(max((int)floorf((f2.fx + 1.0f) * 0.5f * (float)R), 0), R - 1);
    const int y2 = min(max((int)floorf((f2.fy + 1.0f) * 0.5f * (float)R), 0), R - 1);
    return (f2.face * R + y2) * R + x2;
}

struct Bilin {
    int off[4];
    float w[4], dwx[4], dwy[4];
};

__device__ __forceinline__ Bilin bilin_setup(const FaceUV &f, int R)
{
    Bilin b;
    const float px = (f.fx + 1.0f) * 0.5f * (float)R - 0.5f;
    const float py = (f.fy + 1.0f) * 0.5f * (float)R - 0.5f;
    const float flx = floorf(px), fly = floorf(py);
    const int x0 = (int)flx, y0 = (int)fly;
    const float tx = px - flx, ty = py - fly;
    b.w[0] = (1.f - tx) * (1.f - ty); b.w[1] = tx * (1.f - ty); b.w[2] = (1.f - tx) * ty; b.w[3] = tx * ty;
    b.dwx[0] = -(1.f - ty); b.dwx[1] = (1.f - ty); b.dwx[2] = -ty; b.dwx[3] = ty;
    b.dwy[0] = -(1.f - tx); b.dwy[1] = -tx; b.dwy[2] = (1.f - tx); b.dwy[3] = tx;
#pragma unroll
    for (int k = 0; k < 4; ++k) b.off[k] = texel_offset(f.face, x0 + (k & 1), y0 + (k >> 1), R);
    return b;
}

struct MipStack {
    const float *tex[8];
    float *grad[8];
    int n, R0, C;
};

__global__ void __launch_bounds__(THREADS)
cube_sample_fwd_kernel(MipStack m, const float *__restrict__ dirs, const float *__restrict__ level, int64_t n,
                       float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const FaceUV f = dir_to_face(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
    float lv = level ? fminf(fmaxf(level[i], 0.0f), (float)(m.n - 1)) : 0.0f;
    const int l0 = min((int)floorf(lv), m.n - 1), l1 = min(l0 + 1, m.n - 1);
    const float t = lv - (float)l0;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int l = pass == 0 ? l0 : l1;
        const float wl = pass == 0 ? 1.0f - t : t;
        if (pass == 1 && (l1 == l0 || wl == 0.0f)) break;
        const int R = m.R0 >> l;
        const Bilin b = bilin_setup(f, R);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float *p = m.tex[l] + (size_t)b.off[k] * m.C;
            for (int c = 0; c < m.C; ++c) acc[c] += wl * b.w[k] * p[c];
        }
    }
    for (int c = 0; c < m.C; ++c) out[i * m.C + c] = acc[c];
}

// d_out -> d_tex, d_dirs, d_level.
// Texture gradients: every sample adds to 4 texels x C channels of one or two levels.  The small levels (the 16^2
// diffuse map, the coarse specular mips) receive ALL samples on a few thousand texels: global float atomics on
// them serialise at the memory side (measured 1.9 s for 7e7 samples).  Levels of at most LDS_LEVEL_FLOATS floats
// are therefore accumulated in an LDS copy per workgroup (ds_add_f32) and flushed once; only the large levels,
// where collisions are rare, use global atomics directly.
constexpr int BWD_THREADS = 1024;
constexpr int LDS_BUDGET_FLOATS = 24 * 1024;   // 96 KiB: 6*32^2*3 + 6*16^2*3 = 23040 floats

struct LdsPlan {
    int off[8];   // float offset of the level's LDS copy, -1 = global atomics
    int total;
};

__global__ void __launch_bounds__(BWD_THREADS)
cube_sample_bwd_kernel(MipStack m, LdsPlan plan, const float *__restrict__ dirs, const float *__restrict__ level,
                       int64_t n, const float *__restrict__ dout, float *__restrict__ d_dirs,
                       float *__restrict__ d_level)
{
    extern __shared__ float s_grad[];
    for (int e = threadIdx.x; e < plan.total; e += BWD_THREADS) s_grad[e] = 0.0f;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * BWD_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * BWD_THREADS) {
        const float dx = dirs[3 * i], dy = dirs[3 * i + 1], dz = dirs[3 * i + 2];
        const FaceUV f = dir_to_face(dx, dy, dz);
        const float lraw = level ? level[i] : 0.0f;
        const float lv = fminf(fmaxf(lraw, 0.0f), (float)(m.n - 1));
        const int l0 = min((int)floorf(lv), m.n - 1), l1 = min(l0 + 1, m.n - 1);
        const float t = lv - (float)l0;
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < m.C; ++c) g[c] = dout[i * m.C + c];
        float dfx = 0.f, dfy = 0.f, s0 = 0.f, s1 = 0.f;  // d/d(face coords), <g, sample(l0)>, <g, sample(l1)>
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int l = pass == 0 ? l0 : l1;
            const float wl = pass == 0 ? 1.0f - t : t;
            if (pass == 1 && l1 == l0) break;
            const int R = m.R0 >> l;
            const Bilin b = bilin_setup(f, R);
            float sv = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float *p = m.tex[l] + (size_t)b.off[k] * m.C;
                float gp = 0.f;
                for (int c = 0; c < m.C; ++c) gp += g[c] * p[c];
                sv += b.w[k] * gp;
                // pixel coords px = (fx + 1) R/2 - 1/2  =>  d px / d fx = R/2
                dfx += wl * b.dwx[k] * gp * 0.5f * (float)R;
                dfy += wl * b.dwy[k] * gp * 0.5f * (float)R;
                if (m.grad[l] && wl != 0.0f) {
                    if (plan.off[l] >= 0) {
                        float *q = s_grad + plan.off[l] + b.off[k] * m.C;
                        for (int c = 0; c < m.C; ++c) atomicAdd(q + c, wl * b.w[k] * g[c]);
                    } else {
                        float *q = m.grad[l] + (size_t)b.off[k] * m.C;
                        for (int c = 0; c < m.C; ++c) atomicAdd(q + c, wl * b.w[k] * g[c]);
                    }
                }
            }
            if (pass == 0) s0 = sv; else s1 = sv;
        }
        if (d_level) d_level[i] = (level && lraw > 0.0f && lraw < (float)(m.n - 1) && l1 != l0) ? (s1 - s0) : 0.0f;
        if (d_dirs) {
            // fx = sx * comp_u / ma, fy = sy * comp_v / ma with ma = |major component|
            float gx = 0.f, gy = 0.f, gz = 0.f;
            const float ima = 1.0f / f.ma;
            switch (f.face) {
            case 0: gz = -dfx * ima; gy = -dfy * ima; gx = -(f.fx * dfx + f.fy * dfy) * ima; break;           // ma = x
            case 1: gz = dfx * ima;  gy = -dfy * ima; gx = (f.fx * dfx + f.fy * dfy) * ima; break;            // ma = -x
            case 2: gx = dfx * ima;  gz = dfy * ima;  gy = -(f.fx * dfx + f.fy * dfy) * ima; break;           // ma = y
            case 3: gx = dfx * ima;  gz = -dfy * ima; gy = (f.fx * dfx + f.fy * dfy) * ima; break;            // ma = -y
            case 4: gx = dfx * ima;  gy = -dfy * ima; gz = -(f.fx * dfx + f.fy * dfy) * ima; break;           // ma = z
            default: gx = -dfx * ima; gy = -dfy * ima; gz = (f.fx * dfx + f.fy * dfy) * ima; break;           // ma = -z
            }
            d_dirs[3 * i] = gx; d_dirs[3 * i + 1] = gy; d_dirs[3 * i + 2] = gz;
        }
    }
    if (plan.total == 0) return;
    __syncthreads();
    for (int l = 0; l < m.n; ++l) {
        if (plan.off[l] < 0 || !m.grad[l]) continue;
        const int R = m.R0 >> l, cnt = 6 * R * R * m.C;
        for (int e = threadIdx.x; e < cnt; e += BWD_THREADS) {
            const float v = s_grad[plan.off[l] + e];
            if (v != 0.0f) atomicAdd(m.grad[l] + e, v);
        }
    }
}

MipStack make_stack(const float *const *tex, float *const *grad, int n, int R0, int C)
{
    MipStack m;
    for (int l = 0; l < 8; ++l) {
        m.tex[l] = l < n ? tex[l] : nullptr;
        m.grad[l] = (grad && l < n) ? grad[l] : nullptr;
    }
    m.n = n; m.R0 = R0; m.C = C;
    return m;
}

}  // namespace

extern "C" {

int rsdf_diffuse_cubemap_fwd(const float *cubemap, int R, float *out, void *stream)
{
    RSDF_CHECK_ARG(R >= 1 && R <= 64, "diffuse_cubemap: R must be in [1,64] (all-pairs filter)");
    diffuse_kernel<false><<<rsdf_blocks(6 * R * R, THREADS / 64), THREADS, R * sizeof(float), (hipStream_t)stream>>>(cubemap, R, out);
    RSDF_RETURN_LAUNCH();
}

int rsdf_diffuse_cubemap_bwd(const float *grad_out, int R, float *grad_cubemap, void *stream)
{
    RSDF_CHECK_ARG(R >= 1 && R <= 64, "diffuse_cubemap: R must be in [1,64] (all-pairs filter)");
    diffuse_kernel<true><<<rsdf_blocks(6 * R * R, THREADS / 64), THREADS, R * sizeof(float), (hipStream_t)stream>>>(grad_out, R, grad_cubemap);
    RSDF_RETURN_LAUNCH();
}

int rsdf_specular_bounds(int R, float cos_cutoff, float *bounds, void *stream)
{
    RSDF_CHECK_ARG(R >= 1, "specular_bounds: bad resolution");
    bounds_kernel<<<rsdf_blocks(6 * R * R, THREADS), THREADS, 0, (hipStream_t)stream>>>(R, cos_cutoff, bounds);
    RSDF_RETURN_LAUNCH();
}

int rsdf_cubemap_texel_table(int R, float *table, void *stream)
{
    RSDF_CHECK_ARG(R >= 1 && table != nullptr, "cubemap_texel_table: bad arguments");
    texel_table_kernel<<<rsdf_blocks(6 * R * R, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        R, reinterpret_cast<float4 *>(table));
    RSDF_RETURN_LAUNCH();
}

int rsdf_specular_cubemap_fwd(const float *cubemap, const float *bounds, const float *texel_table, int R,
                              float roughness, float cos_cutoff, float *out4, void *stream)
{
    RSDF_CHECK_ARG(R >= 1 && R <= 4096, "specular_cubemap_fwd: bad resolution");
    launch_specular<false>(cubemap, 3, bounds, reinterpret_cast<const float4 *>(texel_table), R, roughness, cos_cutoff, out4,
                           (hipStream_t)stream);
    RSDF_RETURN_LAUNCH();
}

int rsdf_specular_cubemap_fwd_norm(const float *cubemap, const float *bounds, const float *texel_table, int R,
                                   float roughness, float cos_cutoff, float *out3, float *wsum, void *stream)
{
    RSDF_CHECK_ARG(R >= 1 && R <= 4096 && wsum != nullptr, "specular_cubemap_fwd_norm: bad arguments");
    launch_specular<false>(cubemap, 3, bounds, reinterpret_cast<const float4 *>(texel_table), R, roughness, cos_cutoff, out3,
                           (hipStream_t)stream, wsum);
    RSDF_RETURN_LAUNCH();
}

int rsdf_specular_cubemap_bwd(const float *grad_out, int grad_channels, const float *bounds,
                              const float *texel_table, int R, float roughness, float cos_cutoff,
                              float *grad_cubemap, void *stream)
{
    RSDF_CHECK_ARG(R >= 1 && R <= 4096 && grad_channels >= 3, "specular_cubemap_bwd: bad arguments");
    launch_specular<true>(grad_out, grad_channels, bounds, reinterpret_cast<const float4 *>(texel_table), R, roughness,
                          cos_cutoff, grad_cubemap, (hipStream_t)stream);
    RSDF_RETURN_LAUNCH();
}

int rsdf_cubemap_avgpool(const float *cubemap, int R, int C, float *out, void *stream)
{
    RSDF_CHECK_ARG(R >= 2 && (R % 2) == 0 && C >= 1, "cubemap_avgpool: R must be even");
    const int64_t n = (int64_t)6 * (R / 2) * (R / 2) * C;
    avgpool_kernel<<<rsdf_blocks(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(cubemap, R, C, out);
    RSDF_RETURN_LAUNCH();
}

int rsdf_cube_sample_fwd(const float *const *mips, int n_mips, int R0, int C, const float *dirs,
                         const float *level, int64_t n, float *out, void *stream)
{
    RSDF_CHECK_ARG(n_mips >= 1 && n_mips <= 8 && C >= 1 && C <= 4 && (R0 >> (n_mips - 1)) >= 1,
                   "cube_sample_fwd: bad mip stack");
    if (n <= 0) return 0;
    cube_sample_fwd_kernel<<<rsdf_blocks(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        make_stack(mips, nullptr, n_mips, R0, C), dirs, level, n, out);
    RSDF_RETURN_LAUNCH();
}

int rsdf_cube_sample_bwd(const float *const *mips, float *const *grad_mips, int n_mips, int R0, int C,
                         const float *dirs, const float *level, int64_t n, const float *grad_out,
                         float *grad_dirs, float *grad_level, void *stream)
{
    RSDF_CHECK_ARG(n_mips >= 1 && n_mips <= 8 && C >= 1 && C <= 4 && (R0 >> (n_mips - 1)) >= 1,
                   "cube_sample_bwd: bad mip stack");
    if (n <= 0) return 0;
    const MipStack st = make_stack(mips, grad_mips, n_mips, R0, C);
    LdsPlan plan;
    plan.total = 0;
    for (int l = 0; l < 8; ++l) plan.off[l] = -1;
    for (int l = n_mips - 1; l >= 0; --l) {   // smallest levels first
        const int R = R0 >> l, cnt = 6 * R * R * C;
        if (!st.grad[l] || plan.total + cnt > LDS_BUDGET_FLOATS) break;
        plan.off[l] = plan.total;
        plan.total += cnt;
    }
    const size_t lds = (size_t)plan.total * sizeof(float);
    if (int rc = rsdf_func_lds(reinterpret_cast<const void *>(cube_sample_bwd_kernel), LDS_BUDGET_FLOATS * sizeof(float)))
        return rc;
    unsigned grid = rsdf_blocks(n, BWD_THREADS);
    if (plan.total > 0 && grid > 512) grid = 512;   // persistent: one LDS copy (and one flush) per workgroup
    cube_sample_bwd_kernel<<<grid, BWD_THREADS, lds, (hipStream_t)stream>>>(st, plan, dirs, level, n, grad_out,
                                                                           grad_dirs, grad_level);
    RSDF_RETURN_LAUNCH();
}

}  // extern "C"
