// H1b specialised for the finite-difference stencil: backward scatter of the 7 taps of each sample
// (centre, +-x, +-y, +-z; models/geometry.py:229-244) WITHOUT per-corner global atomics.
//
// Why: MI355X executes float atomics at the memory side at ~2e10 64-byte requests/s chip-wide
// (MI355X_MICROARCH.md "Global float atomics"); a scattered 4-byte add costs a whole request, and the
// stencil produces 7 x 16 x 8 x 2 = 1792 adds per sample.  The first version of this kernel (one
// atomic per corner, r01a profile) spent 49 % of the whole step here at 0.04 TB/s of added bytes.
//
// How (hashed levels; dense coarse levels keep the run-merging atomic kernel of hashgrid.hip):
//   1. PRODUCE: one thread per (sample, level).  The six taps sit eps away from the centre, at most
//      one cell (eps = one finest-level cell under the progressive schedule), so their corners are
//      merged in registers into the centre cell's 8 corners plus at most 4 new corners per displaced
//      tap.  Each merged (entry, value) record is binned by table range (16384 entries = one LDS-
//      sized slice), counting-sorted inside the workgroup through LDS and appended to its bin's
//      queue in HBM with coalesced stores.
//   2. REDUCE: one workgroup per (level, bin, split) streams its queue, accumulates into a 128 KiB
//      LDS image of the table slice with ds_add_f32, then adds the slice to dtable with contiguous
//      (full-rate) global atomics.
// HBM traffic: ~200 records x 12 B per sample each way instead of ~400 x 64-byte atomic requests.
#include "hashgrid_common.h"

extern "C" int rsdf_internal_hashgrid_bwd_levels(const float *x, const float *dout,
                                                 const rsdf_grid_meta *meta, int64_t n, int ld_dout,
                                                 int col_off, float *dtable, int level_begin,
                                                 int level_count, void *stream);

namespace {

constexpr int BIN_SHIFT = 14;
constexpr int BIN_ENTRIES = 1 << BIN_SHIFT;  // 16384 entries x 2 floats = 128 KiB of LDS
constexpr int MAX_BINS = 64;                 // log2_hashmap_size <= 20
constexpr int P_THREADS = 256;               // producer: samples per workgroup
constexpr int ROUND_RECS = 8;                // records a thread may stage per round
constexpr int STAGE_CAP = P_THREADS * ROUND_RECS;
constexpr int R_THREADS = 1024;              // reducer

struct Record {
    uint32_t idx;  // entry index within the level
    float v0, v1;
};

struct HashedLevels {
    int count;
    int level[RSDF_MAX_LEVELS];
    int n_bins[RSDF_MAX_LEVELS];
    int64_t cap[RSDF_MAX_LEVELS];        // records per bin queue
    int64_t queue_off[RSDF_MAX_LEVELS];  // first record of this level's bin 0 (in records)
    int counter_off[RSDF_MAX_LEVELS];    // first counter of this level
};

// hashed levels have size = 2^log2_hashmap_size, so "mod size" is a mask (checked in plan())
__device__ __forceinline__ uint32_t hash_index(uint32_t x, uint32_t y, uint32_t z, uint32_t size)
{
    return ((x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u)) & (size - 1u);
}

// Stage up to ROUND_RECS records per thread, counting-sort them by bin in LDS, append to the queues.
__device__ __forceinline__ void emit_round(const uint32_t (&ridx)[ROUND_RECS],
                                           const float2 (&rval)[ROUND_RECS], uint32_t valid_mask,
                                           int n_bins, int64_t cap, Record *__restrict__ queue,
                                           int *__restrict__ qcount, float *__restrict__ dlevel,
                                           int *s_cnt, int *s_off, int *s_gbase, Record *s_stage)
{
    const int tid = threadIdx.x;
    // does any thread of the workgroup have a record this round?
    if (!__syncthreads_or(valid_mask != 0)) return;
    if (tid < n_bins) s_cnt[tid] = 0;
    __syncthreads();
    int slot[ROUND_RECS];
#pragma unroll
    for (int r = 0; r < ROUND_RECS; ++r) {
        slot[r] = 0;
        if (valid_mask & (1u << r)) slot[r] = atomicAdd(&s_cnt[ridx[r] >> BIN_SHIFT], 1);
    }
    __syncthreads();
    if (tid < 64) {  // first wavefront: exclusive scan of the bin counts + global reservations
        const int c = tid < n_bins ? s_cnt[tid] : 0;
        const int incl = wave_incl_sum_i(c);
        if (tid < n_bins) {
            s_off[tid] = incl - c;
            s_gbase[tid] = c > 0 ? atomicAdd(&qcount[tid], c) : 0;
        }
        if (tid == 63) s_off[MAX_BINS] = incl;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUND_RECS; ++r) {
        if (valid_mask & (1u << r)) {
            const int pos = s_off[ridx[r] >> BIN_SHIFT] + slot[r];
            s_stage[pos] = Record{ridx[r], rval[r].x, rval[r].y};
        }
    }
    __syncthreads();
    const int total = s_off[MAX_BINS];
    for (int i = tid; i < total; i += P_THREADS) {
        const Record rec = s_stage[i];
        const int b = rec.idx >> BIN_SHIFT;
        const int64_t g = (int64_t)s_gbase[b] + (i - s_off[b]);
        if (g < cap) {
            queue[(int64_t)b * cap + g] = rec;
        } else {  // queue full (capacity is sized with slack; never silently drop)
            atomicAdd(dlevel + 2 * (size_t)rec.idx, rec.v0);
            atomicAdd(dlevel + 2 * (size_t)rec.idx + 1, rec.v1);
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(P_THREADS)
fd7_produce_kernel(const float *__restrict__ x7, const float *__restrict__ dout,
                   const rsdf_grid_meta meta, const HashedLevels hl, int64_t n_samples, int ld,
                   int col_off, Record *__restrict__ queues, int *__restrict__ counters,
                   float *__restrict__ dtable)
{
    __shared__ int s_cnt[MAX_BINS];
    __shared__ int s_off[MAX_BINS + 1];
    __shared__ int s_gbase[MAX_BINS];
    __shared__ Record s_stage[STAGE_CAP];

    const int h = blockIdx.y;
    const int l = hl.level[h];
    const float scale = meta.scale[l];
    const uint32_t size = meta.size[l];
    float *dlevel = dtable + (size_t)meta.offset[l] * 2;
    Record *queue = queues + hl.queue_off[h];
    int *qcount = counters + hl.counter_off[h];
    const int n_bins = hl.n_bins[h];
    const int64_t cap = hl.cap[h];

    const int64_t s = (int64_t)blockIdx.x * P_THREADS + threadIdx.x;
    const bool active = s < n_samples;

    float2 acc[8];
    float2 ex[3][8];  // [axis][0..3: +side new corners, 4..7: -side new corners]
    bool plus[3] = {false, false, false}, minus[3] = {false, false, false};
    CellFrac c0;
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = make_float2(0.f, 0.f);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 8; ++c) ex[a][c] = make_float2(0.f, 0.f);
    c0.c[0] = c0.c[1] = c0.c[2] = 0;

    if (active) {
        const float *xs = x7 + s * 21;
        const float *ds = dout + (s * 7) * (int64_t)ld + col_off + 2 * l;
        c0 = cell_frac(xs[0], xs[1], xs[2], scale);
        {
            const float2 g = make_float2(ds[0], ds[1]);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float w = corner_weight(c0, c);
                acc[c].x = w * g.x;
                acc[c].y = w * g.y;
            }
        }
#pragma unroll
        for (int t = 1; t < 7; ++t) {
            const int a = (t - 1) >> 1;
            const CellFrac ct = cell_frac(xs[3 * t], xs[3 * t + 1], xs[3 * t + 2], scale);
            const float2 g = make_float2(ds[(int64_t)t * ld], ds[(int64_t)t * ld + 1]);
            const int32_t da = (int32_t)(ct.c[a] - c0.c[a]);
            if (da == 0) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float w = corner_weight(ct, c);
                    acc[c].x += w * g.x;
                    acc[c].y += w * g.y;
                }
            } else if (da == 1) {
                // tap cell = centre cell + 1 along a: its low face is the centre's high face
                plus[a] = true;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float w = corner_weight(ct, c);
                    if (((c >> a) & 1) == 0) {
                        acc[c | (1 << a)].x += w * g.x;
                        acc[c | (1 << a)].y += w * g.y;
                    } else {
                        const int o1 = (a == 0) ? 1 : 0, o2 = (a == 2) ? 1 : 2;
                        const int k = ((c >> o1) & 1) | (((c >> o2) & 1) << 1);
                        ex[a][k].x += w * g.x;
                        ex[a][k].y += w * g.y;
                    }
                }
            } else if (da == -1) {
                minus[a] = true;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float w = corner_weight(ct, c);
                    if (((c >> a) & 1) == 1) {
                        acc[c & ~(1 << a)].x += w * g.x;
                        acc[c & ~(1 << a)].y += w * g.y;
                    } else {
                        const int o1 = (a == 0) ? 1 : 0, o2 = (a == 2) ? 1 : 2;
                        const int k = ((c >> o1) & 1) | (((c >> o2) & 1) << 1);
                        ex[a][4 + k].x += w * g.x;
                        ex[a][4 + k].y += w * g.y;
                    }
                }
            } else {
                // tap more than one cell away (eps larger than a cell): rare slow path
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float w = corner_weight(ct, c);
                    const uint32_t idx = hash_index(ct.c[0] + (c & 1), ct.c[1] + ((c >> 1) & 1),
                                                    ct.c[2] + ((c >> 2) & 1), size);
                    atomicAdd(dlevel + 2 * (size_t)idx, w * g.x);
                    atomicAdd(dlevel + 2 * (size_t)idx + 1, w * g.y);
                }
            }
        }
    }

    uint32_t ridx[ROUND_RECS];
    // round 0: the centre cell's 8 corners
#pragma unroll
    for (int c = 0; c < 8; ++c)
        ridx[c] = hash_index(c0.c[0] + (c & 1), c0.c[1] + ((c >> 1) & 1), c0.c[2] + ((c >> 2) & 1), size);
    emit_round(ridx, acc, active ? 0xffu : 0u, n_bins, cap, queue, qcount, dlevel, s_cnt, s_off,
               s_gbase, s_stage);
    // rounds 1..3: new corners of displaced taps, one axis per round
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int o1 = (a == 0) ? 1 : 0, o2 = (a == 2) ? 1 : 2;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t cc[3] = {c0.c[0], c0.c[1], c0.c[2]};
            cc[a] += (k < 4) ? 2u : 0xffffffffu;  // +2 or -1
            cc[o1] += (uint32_t)(k & 1);
            cc[o2] += (uint32_t)((k >> 1) & 1);
            ridx[k] = hash_index(cc[0], cc[1], cc[2], size);
        }
        const uint32_t mask = (active && plus[a] ? 0x0fu : 0u) | (active && minus[a] ? 0xf0u : 0u);
        emit_round(ridx, ex[a], mask, n_bins, cap, queue, qcount, dlevel, s_cnt, s_off, s_gbase,
                   s_stage);
    }
}

__global__ void __launch_bounds__(R_THREADS)
fd7_reduce_kernel(const rsdf_grid_meta meta, const HashedLevels hl, const Record *__restrict__ queues,
                  const int *__restrict__ counters, int n_split, float *__restrict__ dtable)
{
    extern __shared__ __attribute__((aligned(16))) float s_acc[];  // [BIN_ENTRIES][2]
    const int h = blockIdx.y;
    const int b = blockIdx.x / n_split, part = blockIdx.x % n_split;
    if (b >= hl.n_bins[h]) return;
    const int l = hl.level[h];
    const int64_t cap = hl.cap[h];
    int64_t count = counters[hl.counter_off[h] + b];
    if (count > cap) count = cap;
    const int64_t per = (count + n_split - 1) / n_split;
    const int64_t r0 = part * per;
    int64_t r1 = r0 + per;
    if (r1 > count) r1 = count;
    if (r0 >= r1) return;
    const uint32_t size = meta.size[l];
    const int entries = size < (uint32_t)BIN_ENTRIES ? (int)size : BIN_ENTRIES;

    for (int i = threadIdx.x; i < entries * 2; i += R_THREADS) s_acc[i] = 0.0f;
    __syncthreads();
    const Record *q = queues + hl.queue_off[h] + (int64_t)b * cap;
    for (int64_t i = r0 + threadIdx.x; i < r1; i += R_THREADS) {
        const Record rec = q[i];
        const uint32_t e = rec.idx & (BIN_ENTRIES - 1);
        atomicAdd(&s_acc[2 * e], rec.v0);
        atomicAdd(&s_acc[2 * e + 1], rec.v1);
    }
    __syncthreads();
    float *dst = dtable + ((size_t)meta.offset[l] + (size_t)b * BIN_ENTRIES) * 2;
    for (int i = threadIdx.x; i < entries * 2; i += R_THREADS) {
        const float v = s_acc[i];
        if (v != 0.0f) atomicAdd(dst + i, v);
    }
}

// Expected records per (sample, level): 8 for the centre cell + 4 per displaced tap, where a tap is
// displaced with probability ~min(1, eps_unit * scale).
double expected_records(float scale, float eps_unit)
{
    double p = (double)eps_unit * (double)scale;
    if (p > 1.0) p = 1.0;
    return 8.0 + 24.0 * p;
}

int plan(const rsdf_grid_meta *meta, int64_t n_samples, int n_active, float eps_unit, HashedLevels *hl,
         int *n_dense, int64_t *total_records, int *total_counters)
{
    hl->count = 0;
    *n_dense = 0;
    int64_t qoff = 0;
    int coff = 0;
    for (int l = 0; l < n_active; ++l) {
        const uint64_t dense = (uint64_t)meta->res[l] * meta->res[l] * meta->res[l];
        if (dense <= meta->size[l]) {
            if (hl->count != 0) return -1;  // dense levels must precede hashed ones
            ++*n_dense;
            continue;
        }
        if (meta->size[l] & (meta->size[l] - 1)) return -3;  // hashed level sizes are powers of two
        const int h = hl->count++;
        const int nb = (int)((meta->size[l] + BIN_ENTRIES - 1) >> BIN_SHIFT);
        if (nb > MAX_BINS) return -2;
        hl->level[h] = l;
        hl->n_bins[h] = nb;
        const double per_bin = (double)n_samples * expected_records(meta->scale[l], eps_unit) / nb;
        hl->cap[h] = (int64_t)(per_bin * 1.15) + 16384;
        hl->queue_off[h] = qoff;
        hl->counter_off[h] = coff;
        qoff += hl->cap[h] * nb;
        coff += nb;
    }
    *total_records = qoff;
    *total_counters = coff;
    return 0;
}

}  // namespace

extern "C" {

int64_t rsdf_hashgrid_bwd_fd7_scratch_bytes(const rsdf_grid_meta *meta, int64_t n_samples,
                                            int n_active_levels, float eps_unit)
{
    if (!meta) return -1;
    HashedLevels hl;
    int n_dense, n_cnt;
    int64_t n_rec;
    int na = n_active_levels;
    if (na < 0 || na > (int)meta->n_levels) na = (int)meta->n_levels;
    if (plan(meta, n_samples, na, eps_unit, &hl, &n_dense, &n_rec, &n_cnt) != 0) return -1;
    return n_rec * (int64_t)sizeof(Record) + (int64_t)(n_cnt + 64) * (int64_t)sizeof(int) + 256;
}

int rsdf_hashgrid_bwd_fd7(const float *x7, const float *dout, const rsdf_grid_meta *meta,
                          int64_t n_samples, int n_active_levels, int ld_dout, int col_off,
                          float eps_unit, float *dtable, void *scratch, int64_t scratch_bytes,
                          void *stream)
{
    RSDF_CHECK_ARG(meta != nullptr, "hashgrid_bwd_fd7: meta is NULL");
    RSDF_CHECK_ARG(meta->n_features == 2, "hashgrid_bwd_fd7: n_features must be 2");
    const int L = (int)meta->n_levels;
    RSDF_CHECK_ARG(ld_dout >= col_off + 2 * L, "hashgrid_bwd_fd7: ld_dout too small");
    if (n_samples <= 0) return 0;
    int na = n_active_levels;
    if (na < 0 || na > L) na = L;
    HashedLevels hl;
    int n_dense, n_cnt;
    int64_t n_rec;
    RSDF_CHECK_ARG(plan(meta, n_samples, na, eps_unit, &hl, &n_dense, &n_rec, &n_cnt) == 0,
                   "hashgrid_bwd_fd7: unsupported level layout");
    const int64_t need = n_rec * (int64_t)sizeof(Record) + (int64_t)(n_cnt + 64) * (int64_t)sizeof(int) + 256;
    RSDF_CHECK_ARG(scratch != nullptr && scratch_bytes >= need, "hashgrid_bwd_fd7: scratch too small");
    hipStream_t st = (hipStream_t)stream;

    if (n_dense > 0) {
        int rc = rsdf_internal_hashgrid_bwd_levels(x7, dout, meta, n_samples * 7, ld_dout, col_off, dtable,
                                                   0, n_dense, stream);
        if (rc) return rc;
    }
    if (hl.count == 0) return 0;
    // scratch layout: [counters (n_cnt ints, padded)] [records]
    int *counters = (int *)scratch;
    const size_t cbytes = (((size_t)n_cnt * sizeof(int)) + 255) / 256 * 256;
    Record *queues = (Record *)((char *)scratch + cbytes);
    (void)hipMemsetAsync(counters, 0, cbytes, st);
    dim3 pgrid(rsdf_blocks(n_samples, P_THREADS), hl.count);
    fd7_produce_kernel<<<pgrid, P_THREADS, 0, st>>>(x7, dout, *meta, hl, n_samples, ld_dout, col_off,
                                                    queues, counters, dtable);
    int max_bins = 0;
    for (int h = 0; h < hl.count; ++h) max_bins = hl.n_bins[h] > max_bins ? hl.n_bins[h] : max_bins;
    const int n_split = 2;
    const size_t lds = (size_t)BIN_ENTRIES * 2 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fd7_reduce_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 rgrid(max_bins * n_split, hl.count);
    fd7_reduce_kernel<<<rgrid, R_THREADS, lds, st>>>(*meta, hl, queues, counters, n_split, dtable);
    RSDF_RETURN_LAUNCH();
}

}  // extern "C"
