// M1 / M3 / M4 / M5 / M6: ray-AABB test, occupancy-grid marcher, scans, pack/unpack, compaction.
//
// Built with -ffp-contract=off: the occupancy cell index is an integer function of fp32
// arithmetic (hard part 1) and must match the CPU oracle bit for bit, so no
// multiply-add may be fused behind our back.  Division is IEEE (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt).
//
// Specification followed (paths relative to the upstream RISE-SDF tree):
//   lib/nerfacc/cuda/csrc/intersection.cu:16-91       slab test
//   lib/nerfacc/cuda/csrc/ray_marching.cu:9-75        calc_dt, grid_idx_at, grid_occupied_at,
//                                                     distance_to_next_voxel, advance_to_next_voxel
//   lib/nerfacc/cuda/csrc/ray_marching.cu:81-192      the marching loop (count pass / write pass)
//   lib/nerfacc/cuda/csrc/include/helpers_contraction.h:16-21   roi_to_unit
//   lib/nerfacc/cuda/csrc/pack.cu:7-28                unpack_info
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// M1
// ------------------------------------------------------------------------------------------------
__global__ void aabb_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                            const float *__restrict__ aabb, int64_t n, float *__restrict__ t_min,
                            float *__restrict__ t_max)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float ox = rays_o[3 * i], oy = rays_o[3 * i + 1], oz = rays_o[3 * i + 2];
    const float dx = rays_d[3 * i], dy = rays_d[3 * i + 1], dz = rays_d[3 * i + 2];
    float tmin = (aabb[0] - ox) / dx, tmax = (aabb[3] - ox) / dx;
    if (tmin > tmax) { float c = tmin; tmin = tmax; tmax = c; }
    float tymin = (aabb[1] - oy) / dy, tymax = (aabb[4] - oy) / dy;
    if (tymin > tymax) { float c = tymin; tymin = tymax; tymax = c; }
    float near_ = 1e10f, far_ = 1e10f;
    if (!(tmin > tymax || tymin > tmax)) {
        if (tymin > tmin) tmin = tymin;
        if (tymax < tmax) tmax = tymax;
        float tzmin = (aabb[2] - oz) / dz, tzmax = (aabb[5] - oz) / dz;
        if (tzmin > tzmax) { float c = tzmin; tzmin = tzmax; tzmax = c; }
        if (!(tmin > tzmax || tzmin > tmax)) {
            if (tzmin > tmin) tmin = tzmin;
            if (tzmax < tmax) tmax = tzmax;
            near_ = tmin;
            far_ = tmax;
        }
    }
    t_min[i] = near_ > 0.f ? near_ : 0.f;
    t_max[i] = far_;
}

// ------------------------------------------------------------------------------------------------
// M3
// ------------------------------------------------------------------------------------------------
struct Grid {
    float roi[6];
    int res[3];
    const uint8_t *binary;
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ int cell_index(float x, float y, float z, const Grid &g)
{
    if (x < g.roi[0] || x > g.roi[3] || y < g.roi[1] || y > g.roi[4] || z < g.roi[2] || z > g.roi[5])
        return -1;
    const float ux = (x - g.roi[0]) / (g.roi[3] - g.roi[0]);
    const float uy = (y - g.roi[1]) / (g.roi[4] - g.roi[1]);
    const float uz = (z - g.roi[2]) / (g.roi[5] - g.roi[2]);
    const int ix = clampi((int)(ux * (float)g.res[0]), 0, g.res[0] - 1);
    const int iy = clampi((int)(uy * (float)g.res[1]), 0, g.res[1] - 1);
    const int iz = clampi((int)(uz * (float)g.res[2]), 0, g.res[2] - 1);
    return ix * (g.res[1] * g.res[2]) + iy * g.res[2] + iz;
}

__device__ __forceinline__ bool occupied_at(float x, float y, float z, const Grid &g)
{
    const int c = cell_index(x, y, z, g);
    return c >= 0 && g.binary[c] != 0;
}

// The roi arrives as six floats in DEVICE memory (a torch tensor at the reference's boundary), so
// each thread loads it instead of the host reading it back.
struct GridDev {
    const float *roi;
    int res[3];
    const uint8_t *binary;
};
__device__ __forceinline__ Grid load_grid(const GridDev &d)
{
    Grid g;
#pragma unroll
    for (int k = 0; k < 6; ++k) g.roi[k] = d.roi[k];
    g.res[0] = d.res[0]; g.res[1] = d.res[1]; g.res[2] = d.res[2];
    g.binary = d.binary;
    return g;
}

// ------------------------------------------------------------------------------------------------
// M4
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }
__device__ __forceinline__ float calc_dt(float t, float cone, float dt_min, float dt_max)
{
    return clampf(t * cone, dt_min, dt_max);
}
__device__ __forceinline__ float axis_dist(float p, float dir, float inv_dir, float rmin, float rmax,
                                           int res)
{
    const float r = (float)res;
    const float u = (p - rmin) / (rmax - rmin) * r;
    const float s = copysignf(1.0f, dir);
    return ((floorf(u + 0.5f + 0.5f * s) - u) * inv_dir) / r * (rmax - rmin);
}

// ------------------------------------------------------------------------------------------------
// int32 exclusive scan, three phases.  Tile = 256 threads x 8 items.
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ int block_excl_scan(int v, int *total)
{
    __shared__ int wsum[SCAN_THREADS / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int incl = wave_incl_sum_i(v);
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < SCAN_THREADS / 64; ++k) {
        if (k < w) off += wsum[k];
        tot += wsum[k];
    }
    __syncthreads();
    *total = tot;
    return off + incl - v;
}

__global__ void __launch_bounds__(SCAN_THREADS)
scan_tile_sums(const int32_t *__restrict__ in, int64_t n, int32_t *__restrict__ tile_sums)
{
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) s += in[base + k];
    int tot;
    block_excl_scan(s, &tot);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SCAN_THREADS)
scan_tile_offsets(int32_t *__restrict__ tile_sums, int64_t n_tiles, int32_t *__restrict__ total)
{
    int carry = 0;
    for (int64_t b = 0; b < n_tiles; b += SCAN_THREADS) {
        const int64_t i = b + threadIdx.x;
        const int v = i < n_tiles ? tile_sums[i] : 0;
        int tot;
        const int ex = block_excl_scan(v, &tot);
        if (i < n_tiles) tile_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}

// MODE 0: packed_info[i] = {offset, count};  MODE 1: out[i] = offset
template <int MODE>
__global__ void __launch_bounds__(SCAN_THREADS)
scan_apply(const int32_t *in, int64_t n, const int32_t *__restrict__ tile_offsets, int32_t *out)
{
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        s += v[k];
    }
    int tot;
    int run = block_excl_scan(s, &tot) + tile_offsets[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) {
            if (MODE == 0) {
                out[2 * (base + k)] = run;
                out[2 * (base + k) + 1] = v[k];
            } else {
                out[base + k] = run;
            }
        }
        run += v[k];
    }
}

__global__ void u8_to_i32(const uint8_t *__restrict__ in, int64_t n, int32_t *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// M6 / M5
// ------------------------------------------------------------------------------------------------
// counts[r] = number of samples of ray r, from SORTED ray_indices (zero-initialised counts): the
// last sample of a run adds its end position, the first subtracts its start position.
__global__ void run_bounds_kernel(const int64_t *__restrict__ ri, int64_t n,
                                  int32_t *__restrict__ counts)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t r = ri[i];
    // The last sample of the run adds (i+1), the first subtracts i: counts[r] = end - start.
    if (i == n - 1 || ri[i + 1] != r) atomicAdd(&counts[r], (int32_t)(i + 1));
    if (i == 0 || ri[i - 1] != r) atomicAdd(&counts[r], -(int32_t)i);
}

__global__ void unpack_info_kernel(const int32_t *__restrict__ packed, int64_t n_rays,
                                   int64_t *__restrict__ ri)
{
    // one wavefront per ray: coalesced int64 stores along the ray's segment
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= n_rays) return;
    const int base = packed[2 * r], steps = packed[2 * r + 1];
    for (int j = lane_id(); j < steps; j += 64) ri[base + j] = r;
}

__global__ void compact_kernel(const uint8_t *__restrict__ keep, const int32_t *__restrict__ off,
                               const int64_t *__restrict__ ri, const float *__restrict__ ts,
                               const float *__restrict__ te, int64_t n, int64_t *__restrict__ ri_o,
                               float *__restrict__ ts_o, float *__restrict__ te_o, const float *__restrict__ extra,
                               float *__restrict__ extra_o)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !keep[i]) return;
    const int o = off[i];
    ri_o[o] = ri[i];
    ts_o[o] = ts[i];
    te_o[o] = te[i];
    if (extra != nullptr) extra_o[o] = extra[i];
}

int scan_i32(const int32_t *in, int64_t n, int32_t *out, int mode, int32_t *total, void *scratch,
             hipStream_t st)
{
    const int64_t n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    int32_t *tile = (int32_t *)scratch;
    if (n_tiles > 0) scan_tile_sums<<<(unsigned)n_tiles, SCAN_THREADS, 0, st>>>(in, n, tile);
    scan_tile_offsets<<<1, SCAN_THREADS, 0, st>>>(tile, n_tiles, total);
    if (n_tiles > 0) {
        if (mode == 0) scan_apply<0><<<(unsigned)n_tiles, SCAN_THREADS, 0, st>>>(in, n, tile, out);
        else scan_apply<1><<<(unsigned)n_tiles, SCAN_THREADS, 0, st>>>(in, n, tile, out);
    }
    return 0;
}

}  // namespace

extern "C" {

int rsdf_ray_aabb_intersect(const float *rays_o, const float *rays_d, const float *aabb,
                            int64_t n_rays, float *t_min, float *t_max, void *stream)
{
    RSDF_CHECK_ARG(n_rays >= 0, "ray_aabb_intersect: n_rays < 0");
    if (n_rays == 0) return 0;
    aabb_kernel<<<rsdf_blocks(n_rays, 256), 256, 0, (hipStream_t)stream>>>(rays_o, rays_d, aabb,
                                                                          n_rays, t_min, t_max);
    RSDF_RETURN_LAUNCH();
}

int64_t rsdf_scan_scratch_bytes(int64_t n)
{
    return ((n + SCAN_TILE - 1) / SCAN_TILE + 1) * (int64_t)sizeof(int32_t);
}

int rsdf_pack_from_counts(const int32_t *counts, int64_t n, int32_t *packed_info, int32_t *total,
                          void *scratch, void *stream)
{
    RSDF_CHECK_ARG(n >= 0 && total && scratch, "pack_from_counts: bad arguments");
    scan_i32(counts, n, packed_info, 0, total, scratch, (hipStream_t)stream);
    RSDF_RETURN_LAUNCH();
}

int rsdf_counts_from_ray_indices(const int64_t *ray_indices, int64_t n_samples, int64_t n_rays,
                                 int32_t *counts, void *stream)
{
    RSDF_CHECK_ARG(n_samples >= 0 && n_rays >= 0, "counts_from_ray_indices: negative size");
    hipStream_t st = (hipStream_t)stream;
    if (n_rays > 0) (void)hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)n_rays, st);
    if (n_samples > 0)
        run_bounds_kernel<<<rsdf_blocks(n_samples, 256), 256, 0, st>>>(ray_indices, n_samples, counts);
    RSDF_RETURN_LAUNCH();
}

int rsdf_unpack_info(const int32_t *packed_info, int64_t n_rays, int64_t *ray_indices, void *stream)
{
    if (n_rays <= 0) return 0;
    unpack_info_kernel<<<rsdf_blocks(n_rays * 64, 256), 256, 0, (hipStream_t)stream>>>(
        packed_info, n_rays, ray_indices);
    RSDF_RETURN_LAUNCH();
}

int rsdf_compact_samples(const uint8_t *keep, const int64_t *ray_indices, const float *t_starts,
                         const float *t_ends, int64_t n, int32_t *offsets, int32_t *n_kept,
                         void *scan_scratch, int64_t *ray_indices_out, float *t_starts_out,
                         float *t_ends_out, const float *extra, float *extra_out, void *stream)
{
    RSDF_CHECK_ARG(n >= 0 && n_kept && scan_scratch, "compact_samples: bad arguments");
    RSDF_CHECK_ARG((extra == nullptr) == (extra_out == nullptr), "compact_samples: extra and extra_out go together");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        (void)hipMemsetAsync(n_kept, 0, sizeof(int32_t), st);
        RSDF_RETURN_LAUNCH();
    }
    // offsets doubles as the int32 copy of keep, scanned in place (scan_apply reads before it writes
    // within a tile and tiles are disjoint).
    u8_to_i32<<<rsdf_blocks(n, 256), 256, 0, st>>>(keep, n, offsets);
    scan_i32(offsets, n, offsets, 1, n_kept, scan_scratch, st);
    compact_kernel<<<rsdf_blocks(n, 256), 256, 0, st>>>(keep, offsets, ray_indices, t_starts, t_ends,
                                                        n, ray_indices_out, t_starts_out, t_ends_out, extra, extra_out);
    RSDF_RETURN_LAUNCH();
}

}  // extern "C"

// ---- marcher entry points (roi read from device memory) ---------------------------------------
namespace {
// The reference marches one ray per thread (ray_marching.cu:81-192); its sample set is an integer function of a SERIAL fp32
// recurrence (t1 = t0 + dt; the do { _t += dt } while skip), so the arithmetic cannot be re-associated -- but the mapping
// can change.  Here ONE WAVEFRONT marches a ray.  Per trip, lane k runs the "occupied" recurrence k steps ahead of the
// ray's current state (k dependent adds: the only serial part, two vector instructions per step for the whole wave), all
// lanes look their cell up at once (one load latency per up-to-64 steps instead of one per step), a ballot finds the
// first lane whose step the reference would NOT have taken as occupied (outside [near, far) or an empty cell), the
// lanes before it emit their samples with one coalesced store each, and the ray continues from exactly that lane's
// state -- into the reference's skip if the cell was empty.  Same (t0, t1) bit for bit, same order.  Inside empty space a
// trip speculates a single step; after a fully occupied trip the depth grows 8x up to 64.
// Round 2's thread-per-ray kernel was pure dependent-load latency: 0.45 ms per pass whatever the ray count (143 ms of a
// step at the reference's 4096-ray chunks); round 3's first cut (8 speculative steps per thread) 0.24.
constexpr int MARCH_THREADS = 256;     // four rays per workgroup

// STAGED marching (round 5): the reference's two passes march every ray TWICE (count, then write: ray_marching.cu:257-289), and
// a pass is bound by the serial recurrence, not by its stores (0.37 + 0.83 ms per 28672-ray chunk, 2 x 0.21 ms per 4096-ray
// training batch).  With ``stage`` the COUNT pass also parks each ray's (t0, t1) pairs in a per-ray slot of ``stride`` samples;
// the WRITE pass of a ray whose count fits its slot is then a coalesced copy into the packed arrays -- the same values, produced
// by the same instruction sequence, once.  A ray with more samples than its slot (the caller's stride is a hint, not a
// bound) is marched again as before.
struct MarchStage {
    float *t0, *t1;          // [n_rays][stride] each (nullptr: plain two-pass marching)
    int64_t stride;
    const int32_t *counts;   // WRITE pass: the count pass's per-ray sample counts
};

template <bool WRITE>
__global__ void __launch_bounds__(MARCH_THREADS)
march_entry(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
            const float *__restrict__ t_min, const float *__restrict__ t_max, GridDev gd,
            float step_size, float cone_angle, int64_t n_rays,
            const int32_t *__restrict__ packed_info, int32_t *__restrict__ num_steps,
            int64_t *__restrict__ ray_indices, float *__restrict__ t_starts,
            float *__restrict__ t_ends, const MarchStage stage)
{
    const Grid g = load_grid(gd);
    const int64_t i = ((int64_t)blockIdx.x * MARCH_THREADS + threadIdx.x) >> 6;     // ray = wavefront
    if (i >= n_rays) return;                                                        // (wave-uniform)
    const int lane = lane_id();
    if (WRITE && stage.t0 != nullptr && (int64_t)stage.counts[i] <= stage.stride) {
        // every sample of this ray is parked: copy (at most packed_info's clamped count of them)
        const int64_t base = packed_info[2 * i];
        const int limit = packed_info[2 * i + 1];
        const float *p0 = stage.t0 + i * stage.stride, *p1 = stage.t1 + i * stage.stride;
        for (int k = lane; k < limit; k += 64) {
            t_starts[base + k] = p0[k];
            t_ends[base + k] = p1[k];
            ray_indices[base + k] = i;
        }
        return;
    }
    float *const s0 = (!WRITE && stage.t0 != nullptr) ? stage.t0 + i * stage.stride : nullptr;
    float *const s1 = (!WRITE && stage.t0 != nullptr) ? stage.t1 + i * stage.stride : nullptr;
    const float ox = rays_o[3 * i], oy = rays_o[3 * i + 1], oz = rays_o[3 * i + 2];
    const float dx = rays_d[3 * i], dy = rays_d[3 * i + 1], dz = rays_d[3 * i + 2];
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    const float near_ = t_min[i], far_ = t_max[i];
    const float dt_min = step_size, dt_max = 1e10f;
    int64_t base = 0;
    int limit = 0;
    if (WRITE) {
        base = packed_info[2 * i];
        // never write more than packed_info says (normally exactly the recount below).  A caller that sized its buffers
        // from an earlier step and clamped packed_info to that capacity (OccGridEstimator capacity mode) stays in bounds.
        limit = packed_info[2 * i + 1];
    }

    int j = 0;
    float t0 = near_;
    float dt = calc_dt(t0, cone_angle, dt_min, dt_max);
    float t1 = t0 + dt;
    float t_mid = (t0 + t1) * 0.5f;
    int n_spec = 64;
    while (t_mid < far_) {
        // lane k: the state after k occupied steps (the reference's "occupied" branch: t0 = t1; t1 = t0 + dt(t0))
        float l0 = t0, l1 = t1;
        for (int k = 0; k + 1 < n_spec; ++k) {
            if (lane > k) {
                l0 = l1;
                l1 = l0 + calc_dt(l0, cone_angle, dt_min, dt_max);
            }
        }
        // lane 0 tests the ray's own t_mid (after a skip that is the accumulated _t, NOT (t0 + t1) / 2 re-rounded);
        // the others the midpoint the occupied branch would have formed
        const float lm = lane == 0 ? t_mid : (l0 + l1) * 0.5f;
        bool take = false;
        if (lane < n_spec && lm < far_) take = occupied_at(ox + lm * dx, oy + lm * dy, oz + lm * dz, g);
        const unsigned long long mask = __ballot(take);
        const int m = mask == ~0ull ? 64 : __ffsll((long long)~mask) - 1;     // leading accepted steps (<= n_spec)
        if (WRITE && lane < m && j + lane < limit) {
            t_starts[base + j + lane] = l0;
            t_ends[base + j + lane] = l1;
            ray_indices[base + j + lane] = i;
        }
        if (!WRITE && s0 != nullptr && lane < m && (int64_t)(j + lane) < stage.stride) {
            s0[j + lane] = l0;
            s1[j + lane] = l1;
        }
        j += m;
        if (m > 0) {   // the ray's state after m occupied steps = the successor of lane m - 1's state
            const float n0 = l1, n1 = n0 + calc_dt(n0, cone_angle, dt_min, dt_max), nm = (n0 + n1) * 0.5f;
            t0 = __shfl(n0, m - 1, 64);
            t1 = __shfl(n1, m - 1, 64);
            t_mid = __shfl(nm, m - 1, 64);
        }
        if (m == n_spec) {                        // every speculated step was taken: go deeper
            n_spec = n_spec >= 8 ? 64 : n_spec * 8;
            continue;
        }
        if (!(t_mid < far_)) break;
        // the next step's cell is empty (or outside the box): the reference's skip, computed by every lane alike
        const float x = ox + t_mid * dx, y = oy + t_mid * dy, z = oz + t_mid * dz;
        const float tx = axis_dist(x, dx, ix, g.roi[0], g.roi[3], g.res[0]);
        const float ty = axis_dist(y, dy, iy, g.roi[1], g.roi[4], g.res[1]);
        const float tz = axis_dist(z, dz, iz, g.roi[2], g.roi[5], g.res[2]);
        float t_target = t_mid + fmaxf(fminf(fminf(tx, ty), tz), 0.0f);
        t_target = fminf(t_target, far_);
        float _t = t_mid;
        do { _t += dt_min; } while (_t < t_target);
        t_mid = _t;
        dt = calc_dt(t_mid, cone_angle, dt_min, dt_max);
        t0 = t_mid - dt * 0.5f;
        t1 = t_mid + dt * 0.5f;
        n_spec = (m == 0) ? 1 : 64;               // still in empty space: look at one cell next time
    }
    if (!WRITE && lane == 0) num_steps[i] = j;
}

__global__ void query_occ_entry(const float *__restrict__ xyz, GridDev gd, int64_t n,
                                uint8_t *__restrict__ occ, int32_t *__restrict__ cell)
{
    const Grid g = load_grid(gd);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = cell_index(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], g);
    occ[i] = (c >= 0 && g.binary[c] != 0) ? 1 : 0;
    if (cell) cell[i] = c;
}
}  // namespace

extern "C" {

int rsdf_march_count(const float *rays_o, const float *rays_d, const float *t_min,
                     const float *t_max, const float *roi, const uint8_t *binary, int res_x,
                     int res_y, int res_z, float step_size, float cone_angle, int64_t n_rays,
                     int32_t *num_steps, void *stream)
{
    RSDF_CHECK_ARG(res_x > 0 && res_y > 0 && res_z > 0, "march_count: bad grid resolution");
    RSDF_CHECK_ARG(step_size > 0.f, "march_count: step_size must be > 0");
    if (n_rays <= 0) return 0;
    GridDev gd{roi, {res_x, res_y, res_z}, binary};
    march_entry<false><<<rsdf_blocks(n_rays * 64, MARCH_THREADS), MARCH_THREADS, 0, (hipStream_t)stream>>>(
        rays_o, rays_d, t_min, t_max, gd, step_size, cone_angle, n_rays, nullptr, num_steps, nullptr,
        nullptr, nullptr, MarchStage{nullptr, nullptr, 0, nullptr});
    RSDF_RETURN_LAUNCH();
}

int rsdf_march_count_staged(const float *rays_o, const float *rays_d, const float *t_min, const float *t_max,
                            const float *roi, const uint8_t *binary, int res_x, int res_y, int res_z, float step_size,
                            float cone_angle, int64_t n_rays, int32_t *num_steps, int64_t stride, float *stage_t0,
                            float *stage_t1, void *stream)
{
    RSDF_CHECK_ARG(res_x > 0 && res_y > 0 && res_z > 0, "march_count_staged: bad grid resolution");
    RSDF_CHECK_ARG(step_size > 0.f, "march_count_staged: step_size must be > 0");
    RSDF_CHECK_ARG(stride >= 1 && stage_t0 != nullptr && stage_t1 != nullptr, "march_count_staged: bad staging buffers");
    if (n_rays <= 0) return 0;
    GridDev gd{roi, {res_x, res_y, res_z}, binary};
    march_entry<false><<<rsdf_blocks(n_rays * 64, MARCH_THREADS), MARCH_THREADS, 0, (hipStream_t)stream>>>(
        rays_o, rays_d, t_min, t_max, gd, step_size, cone_angle, n_rays, nullptr, num_steps, nullptr, nullptr, nullptr,
        MarchStage{stage_t0, stage_t1, stride, nullptr});
    RSDF_RETURN_LAUNCH();
}

int rsdf_march_write(const float *rays_o, const float *rays_d, const float *t_min,
                     const float *t_max, const float *roi, const uint8_t *binary, int res_x,
                     int res_y, int res_z, float step_size, float cone_angle, int64_t n_rays,
                     const int32_t *packed_info, int64_t *ray_indices, float *t_starts,
                     float *t_ends, void *stream)
{
    RSDF_CHECK_ARG(res_x > 0 && res_y > 0 && res_z > 0, "march_write: bad grid resolution");
    RSDF_CHECK_ARG(step_size > 0.f, "march_write: step_size must be > 0");
    if (n_rays <= 0) return 0;
    GridDev gd{roi, {res_x, res_y, res_z}, binary};
    march_entry<true><<<rsdf_blocks(n_rays * 64, MARCH_THREADS), MARCH_THREADS, 0, (hipStream_t)stream>>>(
        rays_o, rays_d, t_min, t_max, gd, step_size, cone_angle, n_rays, packed_info, nullptr,
        ray_indices, t_starts, t_ends, MarchStage{nullptr, nullptr, 0, nullptr});
    RSDF_RETURN_LAUNCH();
}

int rsdf_march_write_staged(const float *rays_o, const float *rays_d, const float *t_min, const float *t_max,
                            const float *roi, const uint8_t *binary, int res_x, int res_y, int res_z, float step_size,
                            float cone_angle, int64_t n_rays, const int32_t *packed_info, const int32_t *num_steps,
                            int64_t stride, const float *stage_t0, const float *stage_t1, int64_t *ray_indices,
                            float *t_starts, float *t_ends, void *stream)
{
    RSDF_CHECK_ARG(res_x > 0 && res_y > 0 && res_z > 0, "march_write_staged: bad grid resolution");
    RSDF_CHECK_ARG(step_size > 0.f, "march_write_staged: step_size must be > 0");
    RSDF_CHECK_ARG(stride >= 1 && stage_t0 != nullptr && stage_t1 != nullptr && num_steps != nullptr,
                   "march_write_staged: bad staging buffers");
    if (n_rays <= 0) return 0;
    GridDev gd{roi, {res_x, res_y, res_z}, binary};
    march_entry<true><<<rsdf_blocks(n_rays * 64, MARCH_THREADS), MARCH_THREADS, 0, (hipStream_t)stream>>>(
        rays_o, rays_d, t_min, t_max, gd, step_size, cone_angle, n_rays, packed_info, nullptr, ray_indices, t_starts, t_ends,
        MarchStage{const_cast<float *>(stage_t0), const_cast<float *>(stage_t1), stride, num_steps});
    RSDF_RETURN_LAUNCH();
}

int rsdf_query_occ(const float *samples, const float *roi, const uint8_t *binary, int res_x,
                   int res_y, int res_z, int64_t n, uint8_t *occ, int32_t *cell, void *stream)
{
    if (n <= 0) return 0;
    GridDev gd{roi, {res_x, res_y, res_z}, binary};
    query_occ_entry<<<rsdf_blocks(n, 256), 256, 0, (hipStream_t)stream>>>(samples, gd, n, occ, cell);
    RSDF_RETURN_LAUNCH();
}

}  // extern "C"
