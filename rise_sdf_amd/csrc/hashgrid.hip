// H1 / H1b / H2: multiresolution hash-grid encoding, forward gather and backward scatter.
//
// Replaces tcnn.Encoding(3, {otype: HashGrid}) as constructed at models/network_utils.py:47-50 and
// called at :59 (and the progressive level mask of :58-68, fused here as n_active_levels).
// tiny-cuda-nn is not part of the reference tree; the definition implemented here is the build's
// own statement of the Instant-NGP encoding, identical to oracle/risesdf_oracle.c:
//   pos = fmaf(scale_l, x, 0.5); cell = floor(pos); w = pos - cell
//   index = x + y*res + z*res^2 if res^3 <= size_l else x ^ y*2654435761 ^ z*805459861; mod size_l
//   out_f = sum_{corner 0..7} fmaf(w_c, table[index_c][f], .)   w_c = prod_d (bit_d ? w_d : 1-w_d)
// With -ffp-contract=off and the explicit fmaf chain the forward is bit-identical to the oracle.
//
// Mapping: blockIdx.y = level, threads along samples.  A wavefront therefore gathers 64 neighbouring
// samples of one level: at coarse levels (and for the 7 finite-difference taps of one sample, which
// the caller lays out adjacently) most lanes hit the same few table lines and the texture-address
// unit merges them; at the hashed fine levels every lane is a random 8-byte read and the kernel is
// bound by L2 / Infinity-Cache line traffic (the whole 55 MiB table is MALL resident).
#include "common.h"
#include "hashgrid_common.h"

namespace {

constexpr int THREADS = 256;

template <int F>
struct Feat;
template <>
struct Feat<1> { using T = float; };
template <>
struct Feat<2> { using T = float2; };
template <>
struct Feat<4> { using T = float4; };

#ifndef RSDF_GEN_FWD_GROUP
#define RSDF_GEN_FWD_GROUP 1024
#endif
constexpr int FWD_GROUP = RSDF_GEN_FWD_GROUP;   // tiles per sample group

template <int F>
__global__ void __launch_bounds__(THREADS)
hashgrid_fwd_kernel(const float *__restrict__ x, const float *__restrict__ table,
                    const rsdf_grid_meta meta, int64_t n, int n_levels, int n_active, float *__restrict__ out,
                    int ld_out, int col_off, int write_xyz, float xyz_scale, float xyz_offset)
{
    // 1-D grid in sample-group-major order (hashgrid_fd7.hip, fd7_fwd_kernel): all levels of FWD_GROUP consecutive tiles,
    // then the next group.  A level writes F floats into each sample's [ld_out] row; with the level as the slow grid
    // dimension the 16 partial writes of an output line were whole passes over the batch apart (read-modify-write in
    // HBM each time), inside a group they meet in the L2 / MALL.
    const int64_t per_group = (int64_t)FWD_GROUP * n_levels;
    const int64_t grp = blockIdx.x / per_group, r = blockIdx.x - grp * per_group;
    const int l = (int)(r / FWD_GROUP);
    const int64_t s = (grp * FWD_GROUP + (r - (int64_t)l * FWD_GROUP)) * THREADS + threadIdx.x;
    if (s >= n) return;
    float *o = out + s * ld_out + col_off + l * F;
    const float px = x[3 * s], py = x[3 * s + 1], pz = x[3 * s + 2];
    if (write_xyz && l == 0) {
        // CompositeEncoding include_xyz: x * xyz_scale + xyz_offset (separate mul, add)
        float *ox = out + s * ld_out;
        ox[0] = px * xyz_scale + xyz_offset;
        ox[1] = py * xyz_scale + xyz_offset;
        ox[2] = pz * xyz_scale + xyz_offset;
    }
    if (l >= n_active) {
#pragma unroll
        for (int f = 0; f < F; ++f) o[f] = 0.0f;
        return;
    }
    const LevelInfo li = level_info(meta, l);
    const float posx = fmaf(li.scale, px, 0.5f), posy = fmaf(li.scale, py, 0.5f),
                posz = fmaf(li.scale, pz, 0.5f);
    const float fx = floorf(posx), fy = floorf(posy), fz = floorf(posz);
    const uint32_t cx = (uint32_t)(int32_t)fx, cy = (uint32_t)(int32_t)fy, cz = (uint32_t)(int32_t)fz;
    const float wx = posx - fx, wy = posy - fy, wz = posz - fz;
    using V = typename Feat<F>::T;
    const V *tl = reinterpret_cast<const V *>(table) + li.offset;

    V v[8];
    float wc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float w = 1.0f;
        w *= (c & 1) ? wx : 1.0f - wx;
        w *= (c & 2) ? wy : 1.0f - wy;
        w *= (c & 4) ? wz : 1.0f - wz;
        wc[c] = w;
        const uint32_t idx = grid_index(cx + (c & 1), cy + ((c >> 1) & 1), cz + ((c >> 2) & 1), li);
        v[c] = tl[idx];
    }
    float acc[F];
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float *vf = reinterpret_cast<const float *>(&v[c]);
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f] = fmaf(wc[c], vf[f], acc[f]);
    }
#pragma unroll
    for (int f = 0; f < F; ++f) o[f] = acc[f];
}

// Staged form of the same gather for large batches (rsdf_hashgrid_fwd_staged).  The sample-group-major order above buys
// whole-line output writes with table locality: the 8 XCDs change level every FWD_GROUP tiles and every change refills a
// 4 MiB level into each 4 MiB L2 (a third of the line requests miss at 1024 tiles).  Here the two concerns are separated:
//   1. planes kernel, PURE level-major (blockIdx.y = level): every workgroup in flight reads the same L2-resident level
//      and writes its F floats per sample as one coalesced [n][F] plane -- no partial rows anywhere;
//   2. rows kernel: 128 samples x all levels of planes -> an LDS tile -> whole [ld_out] rows (xyz columns and the zeros of
//      the inactive levels included), 2 x 4 F L bytes per evaluation of extra streaming traffic.
// The arithmetic is the kernel's above, statement for statement: bit-identical outputs.
template <int F>
__global__ void __launch_bounds__(THREADS)
hashgrid_fwd_planes_kernel(const float *__restrict__ x, const float *__restrict__ table, const rsdf_grid_meta meta,
                           int64_t n, float *__restrict__ planes)
{
    const int l = blockIdx.y;
    const int64_t s = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (s >= n) return;
    const float px = x[3 * s], py = x[3 * s + 1], pz = x[3 * s + 2];
    const LevelInfo li = level_info(meta, l);
    const float posx = fmaf(li.scale, px, 0.5f), posy = fmaf(li.scale, py, 0.5f),
                posz = fmaf(li.scale, pz, 0.5f);
    const float fx = floorf(posx), fy = floorf(posy), fz = floorf(posz);
    const uint32_t cx = (uint32_t)(int32_t)fx, cy = (uint32_t)(int32_t)fy, cz = (uint32_t)(int32_t)fz;
    const float wx = posx - fx, wy = posy - fy, wz = posz - fz;
    using V = typename Feat<F>::T;
    const V *tl = reinterpret_cast<const V *>(table) + li.offset;
    V v[8];
    float wc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float w = 1.0f;
        w *= (c & 1) ? wx : 1.0f - wx;
        w *= (c & 2) ? wy : 1.0f - wy;
        w *= (c & 4) ? wz : 1.0f - wz;
        wc[c] = w;
        const uint32_t idx = grid_index(cx + (c & 1), cy + ((c >> 1) & 1), cz + ((c >> 2) & 1), li);
        v[c] = tl[idx];
    }
    V acc;
    float *af = reinterpret_cast<float *>(&acc);
#pragma unroll
    for (int f = 0; f < F; ++f) af[f] = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float *vf = reinterpret_cast<const float *>(&v[c]);
#pragma unroll
        for (int f = 0; f < F; ++f) af[f] = fmaf(wc[c], vf[f], af[f]);
    }
    // (plain loads and stores: non-temporal hints on x and on the planes made the pass 5 % slower)
    reinterpret_cast<V *>(planes)[(int64_t)l * n + s] = acc;
}

constexpr int ROWS_TILE = 128;   // samples per workgroup of the rows kernel

template <int F>
__global__ void __launch_bounds__(THREADS)
planes_to_rows_kernel(const float *__restrict__ planes, const float *__restrict__ x, int64_t n, int n_levels,
                      int n_active, float *__restrict__ out, int ld_out, int col_off, int write_xyz, float xyz_scale,
                      float xyz_offset)
{
    // tile [ROWS_TILE][tw] with an odd row stride (conflict-free column writes); tw = (3 +) n_levels F columns that
    // start at column c0 of the output row
    extern __shared__ float s_tile[];
    const int nx = write_xyz ? 3 : 0;
    const int tw = nx + n_levels * F, stride = tw | 1, c0 = col_off - nx;
    const int64_t s0 = (int64_t)blockIdx.x * ROWS_TILE;
    const int r = threadIdx.x & (ROWS_TILE - 1), half = threadIdx.x / ROWS_TILE;   // two threads per sample
    const int64_t s = s0 + r;
    using V = typename Feat<F>::T;
    if (s < n) {
        float *row = s_tile + r * stride;
        if (write_xyz && half == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) row[d] = x[3 * s + d] * xyz_scale + xyz_offset;
        }
        for (int l = half; l < n_levels; l += THREADS / ROWS_TILE) {
            V v;
            float *vf = reinterpret_cast<float *>(&v);
            if (l < n_active) {
                v = reinterpret_cast<const V *>(planes)[(int64_t)l * n + s];
            } else {
#pragma unroll
                for (int f = 0; f < F; ++f) vf[f] = 0.0f;
            }
#pragma unroll
            for (int f = 0; f < F; ++f) row[nx + l * F + f] = vf[f];
        }
    }
    __syncthreads();
    const int rows = (int)min((int64_t)ROWS_TILE, n - s0);
    const int total = rows * tw;
    // i / tw without an integer division: (i + 0.5) / tw is at least 0.5 / tw from an integer and i < 2^15
    const float inv_tw = 1.0f / (float)tw;
    for (int i = threadIdx.x; i < total; i += THREADS) {
        const int rr = (int)(((float)i + 0.5f) * inv_tw), c = i - rr * tw;
        out[(s0 + rr) * ld_out + c0 + c] = s_tile[rr * stride + c];
    }
}

// Backward scatter.  RUNS: lanes whose neighbours target the same table entry (the FD taps of one
// sample, or consecutive samples of a ray at a coarse level) first combine their contributions with
// a segmented wave scan; only the last lane of each run issues the atomic.  Atomic traffic is the
// bound for this kernel (MI355X float atomics: ~1.3 TB/s of added bytes when well shaped, an order
// of magnitude less for 64 scattered rows per instruction), so every merged add is a direct win.
template <int F, bool RUNS>
__global__ void __launch_bounds__(THREADS)
hashgrid_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dout,
                    const rsdf_grid_meta meta, int64_t n, int ld_dout, int col_off,
                    float *__restrict__ dtable, int level_begin)
{
    const int64_t s = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    const int l = level_begin + blockIdx.y;
    const bool valid = s < n;
    if (!RUNS && !valid) return;
    const LevelInfo li = level_info(meta, l);
    float g[F];
    float px = 0.f, py = 0.f, pz = 0.f;
    if (valid) {
        px = x[3 * s]; py = x[3 * s + 1]; pz = x[3 * s + 2];
        const float *d = dout + s * ld_dout + col_off + l * F;
#pragma unroll
        for (int f = 0; f < F; ++f) g[f] = d[f];
    } else {
#pragma unroll
        for (int f = 0; f < F; ++f) g[f] = 0.0f;
    }
    const float posx = fmaf(li.scale, px, 0.5f), posy = fmaf(li.scale, py, 0.5f),
                posz = fmaf(li.scale, pz, 0.5f);
    const float fx = floorf(posx), fy = floorf(posy), fz = floorf(posz);
    const uint32_t cx = (uint32_t)(int32_t)fx, cy = (uint32_t)(int32_t)fy, cz = (uint32_t)(int32_t)fz;
    const float wx = posx - fx, wy = posy - fy, wz = posz - fz;
    float *tl = dtable + (size_t)li.offset * F;
    const int lane = lane_id();

#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float w = 1.0f;
        w *= (c & 1) ? wx : 1.0f - wx;
        w *= (c & 2) ? wy : 1.0f - wy;
        w *= (c & 4) ? wz : 1.0f - wz;
        uint32_t idx = grid_index(cx + (c & 1), cy + ((c >> 1) & 1), cz + ((c >> 2) & 1), li);
        float v[F];
#pragma unroll
        for (int f = 0; f < F; ++f) v[f] = w * g[f];
        if (RUNS) {
            if (!valid) idx = 0xffffffffu;  // never equals a real index (size <= 2^31)
            const uint32_t prev = __shfl_up(idx, 1, 64);
            int head = (lane == 0 || prev != idx) ? 1 : 0;
            // segmented inclusive sum over runs of equal idx
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                float u[F];
#pragma unroll
                for (int f = 0; f < F; ++f) u[f] = __shfl_up(v[f], o, 64);
                const int hu = __shfl_up(head, o, 64);
                if (lane >= o && !head) {
#pragma unroll
                    for (int f = 0; f < F; ++f) v[f] += u[f];
                    head = hu;
                }
            }
            const uint32_t next = __shfl_down(idx, 1, 64);
            const bool tail = (lane == 63 || next != idx);
            if (valid && tail) {
#pragma unroll
                for (int f = 0; f < F; ++f) atomicAdd(tl + (size_t)idx * F + f, v[f]);
            }
        } else {
#pragma unroll
            for (int f = 0; f < F; ++f) atomicAdd(tl + (size_t)idx * F + f, v[f]);
        }
    }
}

}  // namespace

extern "C" {

int64_t rsdf_grid_meta_init(rsdf_grid_meta *meta, int n_levels, int n_features, int log2_hashmap_size,
                            int base_resolution, double per_level_scale)
{
    if (!meta || n_levels < 1 || n_levels > RSDF_MAX_LEVELS) return -1;
    meta->n_levels = (uint32_t)n_levels;
    meta->n_features = (uint32_t)n_features;
    uint64_t off = 0;
    for (int l = 0; l < n_levels; ++l) {
        // fp64 on the host, rounded once to fp32
        const double sc = exp2((double)l * log2(per_level_scale)) * (double)base_resolution - 1.0;
        const float scale = (float)sc;
        const uint32_t res = (uint32_t)ceilf(scale) + 1u;
        uint64_t size = (uint64_t)res * res * res;
        size = (size + 7) / 8 * 8;
        const uint64_t cap = 1ull << log2_hashmap_size;
        if (size > cap) size = cap;
        meta->scale[l] = scale;
        meta->res[l] = res;
        meta->offset[l] = (uint32_t)off;
        meta->size[l] = (uint32_t)size;
        off += size;
    }
    return (int64_t)off * n_features;
}

int rsdf_hashgrid_fwd(const float *x, const float *table, const rsdf_grid_meta *meta, int64_t n,
                      int n_active_levels, float *out, int ld_out, int col_off, int write_xyz,
                      float xyz_scale, float xyz_offset, void *stream)
{
    RSDF_CHECK_ARG(meta != nullptr, "hashgrid_fwd: meta is NULL");
    const int L = (int)meta->n_levels, F = (int)meta->n_features;
    RSDF_CHECK_ARG(F == 1 || F == 2 || F == 4, "hashgrid_fwd: n_features must be 1, 2 or 4");
    RSDF_CHECK_ARG(ld_out >= col_off + L * F, "hashgrid_fwd: ld_out too small");
    RSDF_CHECK_ARG(!write_xyz || col_off >= 3, "hashgrid_fwd: write_xyz needs col_off >= 3");
    if (n <= 0) return 0;
    if (n_active_levels < 0 || n_active_levels > L) n_active_levels = L;
    const unsigned tiles = (rsdf_blocks(n, THREADS) + FWD_GROUP - 1) / FWD_GROUP * FWD_GROUP;
    // HIP launches in threads: grid.x * block.x must stay below 2^32
    RSDF_CHECK_ARG((uint64_t)tiles * L * THREADS < (1ull << 32), "hashgrid_fwd: too many points for one launch");
    const unsigned grid = tiles * L;
    hipStream_t st = (hipStream_t)stream;
    switch (F) {
    case 1: hashgrid_fwd_kernel<1><<<grid, THREADS, 0, st>>>(x, table, *meta, n, L, n_active_levels, out, ld_out, col_off, write_xyz, xyz_scale, xyz_offset); break;
    case 2: hashgrid_fwd_kernel<2><<<grid, THREADS, 0, st>>>(x, table, *meta, n, L, n_active_levels, out, ld_out, col_off, write_xyz, xyz_scale, xyz_offset); break;
    default: hashgrid_fwd_kernel<4><<<grid, THREADS, 0, st>>>(x, table, *meta, n, L, n_active_levels, out, ld_out, col_off, write_xyz, xyz_scale, xyz_offset); break;
    }
    RSDF_RETURN_LAUNCH();
}

// points per pass of the staged gather: x (12 B per point) is read once per level and the planes once by the rows
// kernel, so a pass is sized for both to stay in the 256 MB MALL beside the table (tools/bench_gather.py, 15 M points:
// 1 M per pass 2.73e9 evaluations/s, 2 M 2.77, 4 M 2.87, 8 M 2.89, one pass 2.56)
#ifndef RSDF_STAGED_BATCH
#define RSDF_STAGED_BATCH (1 << 22)
#endif
static int64_t staged_batch()
{
    static const int64_t b = [] {
        const char *e = getenv("RSDF_STAGED_BATCH");
        const long long v = e ? atoll(e) : 0;
        return (int64_t)(v > 0 ? v : RSDF_STAGED_BATCH);
    }();
    return b;
}

int64_t rsdf_hashgrid_fwd_staged_scratch_bytes(const rsdf_grid_meta *meta, int64_t n, int n_active_levels)
{
    if (!meta || n < 0) return -1;
    const int L = (int)meta->n_levels, F = (int)meta->n_features;
    if (n_active_levels < 0 || n_active_levels > L) n_active_levels = L;
    return (int64_t)n_active_levels * std::min(n, staged_batch()) * F * (int64_t)sizeof(float);
}

int rsdf_hashgrid_fwd_staged(const float *x, const float *table, const rsdf_grid_meta *meta, int64_t n,
                             int n_active_levels, float *out, int ld_out, int col_off, int write_xyz,
                             float xyz_scale, float xyz_offset, void *scratch, int64_t scratch_bytes, void *stream)
{
    RSDF_CHECK_ARG(meta != nullptr, "hashgrid_fwd_staged: meta is NULL");
    const int L = (int)meta->n_levels, F = (int)meta->n_features;
    RSDF_CHECK_ARG(F == 1 || F == 2 || F == 4, "hashgrid_fwd_staged: n_features must be 1, 2 or 4");
    RSDF_CHECK_ARG(ld_out >= col_off + L * F, "hashgrid_fwd_staged: ld_out too small");
    RSDF_CHECK_ARG(!write_xyz || col_off == 3, "hashgrid_fwd_staged: write_xyz needs col_off == 3");
    if (n <= 0) return 0;
    if (n_active_levels < 0 || n_active_levels > L) n_active_levels = L;
    RSDF_CHECK_ARG(scratch_bytes >= rsdf_hashgrid_fwd_staged_scratch_bytes(meta, n, n_active_levels) &&
                       (scratch != nullptr || n_active_levels == 0),
                   "hashgrid_fwd_staged: scratch too small");
    hipStream_t st = (hipStream_t)stream;
    float *planes = static_cast<float *>(scratch);
    const int tw = (write_xyz ? 3 : 0) + L * F;
    const size_t lds = (size_t)ROWS_TILE * (tw | 1) * sizeof(float);
    if (lds > 48 * 1024) {   // 32 levels of 4 features: 67 KB
        const void *k = F == 1 ? reinterpret_cast<const void *>(planes_to_rows_kernel<1>)
                      : F == 2 ? reinterpret_cast<const void *>(planes_to_rows_kernel<2>)
                               : reinterpret_cast<const void *>(planes_to_rows_kernel<4>);
        if (int rc = rsdf_func_lds(k, lds)) return rc;
    }
    const int64_t batch = staged_batch();
    for (int64_t b0 = 0; b0 < n; b0 += batch) {
        const int64_t nb = std::min(batch, n - b0);
        const float *xb = x + 3 * b0;
        float *ob = out + b0 * ld_out;
        if (n_active_levels > 0) {
            const dim3 grid(rsdf_blocks(nb, THREADS), n_active_levels);
            switch (F) {
            case 1: hashgrid_fwd_planes_kernel<1><<<grid, THREADS, 0, st>>>(xb, table, *meta, nb, planes); break;
            case 2: hashgrid_fwd_planes_kernel<2><<<grid, THREADS, 0, st>>>(xb, table, *meta, nb, planes); break;
            default: hashgrid_fwd_planes_kernel<4><<<grid, THREADS, 0, st>>>(xb, table, *meta, nb, planes); break;
            }
        }
        const unsigned tiles = rsdf_blocks(nb, ROWS_TILE);
        switch (F) {
        case 1: planes_to_rows_kernel<1><<<tiles, THREADS, lds, st>>>(planes, xb, nb, L, n_active_levels, ob, ld_out, col_off, write_xyz, xyz_scale, xyz_offset); break;
        case 2: planes_to_rows_kernel<2><<<tiles, THREADS, lds, st>>>(planes, xb, nb, L, n_active_levels, ob, ld_out, col_off, write_xyz, xyz_scale, xyz_offset); break;
        default: planes_to_rows_kernel<4><<<tiles, THREADS, lds, st>>>(planes, xb, nb, L, n_active_levels, ob, ld_out, col_off, write_xyz, xyz_scale, xyz_offset); break;
        }
    }
    RSDF_RETURN_LAUNCH();
}

int rsdf_hashgrid_bwd(const float *x, const float *dout, const rsdf_grid_meta *meta, int64_t n,
                      int n_active_levels, int ld_dout, int col_off, float *dtable, void *stream)
{
    RSDF_CHECK_ARG(meta != nullptr, "hashgrid_bwd: meta is NULL");
    const int L = (int)meta->n_levels, F = (int)meta->n_features;
    RSDF_CHECK_ARG(F == 1 || F == 2 || F == 4, "hashgrid_bwd: n_features must be 1, 2 or 4");
    RSDF_CHECK_ARG(ld_dout >= col_off + L * F, "hashgrid_bwd: ld_dout too small");
    if (n <= 0) return 0;
    if (n_active_levels < 0 || n_active_levels > L) n_active_levels = L;
    if (n_active_levels == 0) return 0;
    dim3 grid(rsdf_blocks(n, THREADS), n_active_levels);
    hipStream_t st = (hipStream_t)stream;
    switch (F) {
    case 1: hashgrid_bwd_kernel<1, true><<<grid, THREADS, 0, st>>>(x, dout, *meta, n, ld_dout, col_off, dtable, 0); break;
    case 2: hashgrid_bwd_kernel<2, true><<<grid, THREADS, 0, st>>>(x, dout, *meta, n, ld_dout, col_off, dtable, 0); break;
    default: hashgrid_bwd_kernel<4, true><<<grid, THREADS, 0, st>>>(x, dout, *meta, n, ld_dout, col_off, dtable, 0); break;
    }
    RSDF_RETURN_LAUNCH();
}

}  // extern "C"
