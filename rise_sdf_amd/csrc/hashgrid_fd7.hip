// H1 / H1b specialised for the finite-difference stencil (models/geometry.py:229-244): the centre
// of a sample and its six taps x +- eps e_k are encoded / back-propagated together.
//
// Layouts (tap-major structure-of-arrays, so that every kernel reads and writes full cache lines):
//   x7t    [7][S][3]        tap t of sample s (tap 0 = centre, then +x,-x,+y,-y,+z,-z), unit cube
//   planes [L][7][S][2]     encoding (or its gradient) of level l, tap t, sample s, F = 2 features
//
// FORWARD.  One thread per (sample, level).  Under the progressive-eps schedule eps is one cell of
// the finest ACTIVE level, so a tap is at most one cell away from the centre: the thread gathers
// the centre cell's 8 corners plus 4 new corners per displaced tap (8..32 gathers instead of 56) and
// evaluates the 7 trilinear interpolations from registers, each as the same corner-ordered fmaf
// chain as the generic kernel (bit-identical results).
//
// BACKWARD.  MI355X executes float atomics at the memory side at ~2e10 64-byte requests/s chip-wide
// (MI355X_MICROARCH.md "Global float atomics"): a scattered 4-byte add costs a whole request and
// the stencil produces 1792 of them per sample (the r01a profile: 49 % of the step).  Instead:
//   1. PRODUCE: one thread per (sample, level) merges the taps' contributions in registers into
//      the same 8..32 corners, bins the (entry, value) records by table slice (<= 8192 entries, one
//      LDS image), counting-sorts them by bin inside the workgroup through LDS and appends each bin's
//      run to its queue in HBM as 16-byte elements of two records (8 bytes per record: 13 bits of entry
//      within the bin + a block-float pair of the two feature gradients, see PairRec).
//   2. REDUCE: one workgroup per (level, bin, split) streams its queue (4 records in flight per
//      thread), accumulates into a 128 KiB fp64 LDS image with ds_add_f64 (measured 18x the rate of
//      ds_add_f32 on gfx950, tools/lds_atomic_bench.hip) and adds the image to dtable
//      with contiguous (full-rate) global atomics.
// Hashed levels bin by the high index bits; dense levels bin by idx % n_bins (n_bins a power of two) so that a spatially
// compact batch of rays still loads all bins evenly.
#include <stdlib.h>

#include "hashgrid_common.h"

namespace {

constexpr int BIN_SHIFT = 13;
constexpr int BIN_ENTRIES = 1 << BIN_SHIFT;  // 8192 entries x 2 doubles = 128 KiB of LDS
constexpr int MAX_BINS = 64;
constexpr int F_THREADS = 256;   // forward: samples per workgroup
// producer: samples per workgroup.  Large on purpose: a workgroup appends one run per bin per round,
// and HBM write efficiency follows the run length (256 threads: ~0.4 KB runs, 1.6 TB/s; 1024 threads:
// ~1.5 KB runs) -- measured with the RSDF ablation in profiles/README.md.
#ifndef P_THREADS_CFG
#define P_THREADS_CFG 1024
#endif
constexpr int P_THREADS = P_THREADS_CFG;
constexpr int ROUND_RECS = 8;
// LDS staging of one copy-out pass.  A round holds up to P_THREADS * ROUND_RECS records; staging HALF of that (a full
// round then takes two passes) keeps the producer at 62 KB of LDS, i.e. two 1024-thread workgroups per CU: the kernel is
// bound by barrier and memory latency, not by any one unit (PMC: vector issue 24 %, LDS 22 %, 2 TB/s), so the second
// workgroup's work fills the first one's stalls.
#ifndef RSDF_STAGE_RECS
#define RSDF_STAGE_RECS 4
#endif
constexpr int STAGE_CAP = P_THREADS * RSDF_STAGE_RECS;
#ifndef MERGE_MAX_RUNS
#define MERGE_MAX_RUNS 40
#endif
constexpr int R_THREADS = 1024;  // reducer
#ifndef RSDF_R_UNROLL
#define RSDF_R_UNROLL 8     // 16-byte elements in flight per thread (4: 6.67, 8: 6.56 ms per 18.4 M-sample launch)
#endif
constexpr int R_UNROLL = RSDF_R_UNROLL;

struct Record {   // LDS staging form
    uint32_t idx;  // entry index within the level
    float v0, v1;
};
// Queue form.  Round 3: TWO contributions of one bin per 20-byte element (inside a bin an entry needs 13 bits, so the two
// indices share a dword: 10 bytes per contribution instead of 12; -DRSDF_REC_FP32 keeps that form for A/B runs).
// Round 4 (default): 8 bytes per contribution, two per 16-BYTE element -- one aligned dwordx4 store / load per lane instead
// of a 16 + 4 byte pair at a 20-byte stride.  A contribution is (entry, v0, v1) with v0, v1 the two features' gradients of
// ONE table entry (the same corner weight times the two plane gradients): they are stored as a block-float pair,
//     bits 0..12 entry within the bin | 13..20 E = the larger value's biased fp32 exponent | 21..41 m0 | 42..62 m1,
//     v_i = m_i 2^(E - 127 - 19), m_i = rint(v_i 2^(19 - (E - 127))) clamped to +-(2^20 - 1)  (21-bit two's complement),
// i.e. the larger value keeps 20 significant bits (relative error <= 2^-20 = 9.5e-7 per record, against 2^-24 in fp32) and
// the smaller one the same ABSOLUTE step.  The reducer still sums in fp64, so an entry that collects N records of one sign
// ends ~2^-20 / sqrt(3 N) from the exact sum -- closer than an fp32 atomic accumulation from N ~ 20 up, and three orders of
// magnitude closer than the reference's own table gradient (tiny-cuda-nn accumulates it with __half2 atomics).  E = 255
// (inf / NaN in either value) decodes to NaN so that a non-finite gradient still poisons its table row; values below 2^-107
// flush to zero.  A (bin, round) run with an odd record count ends in a half-empty element (second record all zero: it adds
// 0.0 to entry 0; fp32 form: e1 == PAIR_NONE).
// BOTH forms are compiled; rsdf_set_record_format() picks one at run time (round 6; the reference's table gradient is plain
// fp32 atomics, models/network_utils.py:47-59, so the fp32-valued records are the conservative choice and the block-float
// records the fast default; -DRSDF_REC_FP32 only changes the library's initial setting).
struct RecF32 {   // fp32 values, 20 bytes per element
    uint32_t e;    // entry within the bin: first contribution bits 0..15, second 16..31
    float a0, a1, b0, b1;
};
constexpr uint32_t PAIR_NONE = 0xffffu;
struct __attribute__((aligned(16))) RecBF {   // block-float pairs, 16 bytes per element
    unsigned long long r0, r1;
};
__device__ __forceinline__ unsigned long long pack_rec(uint32_t entry, float v0, float v1)
{
    const uint32_t b0 = __float_as_uint(v0) & 0x7fffffffu, b1 = __float_as_uint(v1) & 0x7fffffffu;
    const uint32_t E = (b0 > b1 ? b0 : b1) >> 23;
    if (E < 20u) return (unsigned long long)entry;            // both below 2^-107 (or zero): m0 = m1 = 0, E = 0
    const float sc = __uint_as_float((273u - E) << 23);       // 2^(19 - (E - 127)); E = 255 -> 2^-109 (inf / NaN stay)
    int m0 = (int)rintf(v0 * sc), m1 = (int)rintf(v1 * sc);   // cvt saturates; NaN -> 0 (E = 255 carries it)
    const int lim = (1 << 20) - 1;
    m0 = m0 < -lim ? -lim : (m0 > lim ? lim : m0);
    m1 = m1 < -lim ? -lim : (m1 > lim ? lim : m1);
    return (unsigned long long)(entry | (E << 13) | ((uint32_t)m0 << 21))            // m0's low 11 bits
           | ((unsigned long long)(((uint32_t)m0 >> 11 & 0x3ffu) | ((uint32_t)m1 & 0x1fffffu) << 10) << 32);
}
__device__ __forceinline__ void unpack_rec(unsigned long long r, uint32_t &entry, double &d0, double &d1)
{
    const uint32_t lo = (uint32_t)r, hi = (uint32_t)(r >> 32);
    entry = lo & 0x1fffu;
    const uint32_t E = lo >> 13 & 0xffu;
    const int m0 = (int)(((lo >> 21) | (hi << 11)) << 11) >> 11;     // 21-bit two's complement
    const int m1 = (int)(hi << 1) >> 11;
    // 2^(E - 127 - 19) as a double: exponent field E - 146 + 1023; E = 255 -> NaN
    const double sc = E == 255u ? __longlong_as_double(0x7ff8000000000000ll)
                                : __longlong_as_double((long long)(E + 877u) << 52);
    d0 = (double)m0 * sc;
    d1 = (double)m1 * sc;
}

template <typename REC> struct RecOps;
template <> struct RecOps<RecF32> {
    static __device__ __forceinline__ void store(RecF32 *q, uint32_t e0, float a0, float a1, bool two, uint32_t e1, float b0, float b1)
    {
        *q = RecF32{e0 | ((two ? e1 : PAIR_NONE) << 16), a0, a1, two ? b0 : 0.f, two ? b1 : 0.f};
    }
    static __device__ __forceinline__ RecF32 load(const RecF32 *q) { return *q; }
    static __device__ __forceinline__ RecF32 empty() { return RecF32{0u | (PAIR_NONE << 16), 0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ bool decode(const RecF32 &r, uint32_t &e0, double &a0, double &a1, uint32_t &e1, double &b0,
                                                  double &b1)
    {
        e0 = r.e & 0xffffu, e1 = r.e >> 16;   // entries within the bin (the producer's entry_of)
        a0 = (double)r.a0, a1 = (double)r.a1, b0 = (double)r.b0, b1 = (double)r.b1;
        return e1 != PAIR_NONE;
    }
};
template <> struct RecOps<RecBF> {
    static __device__ __forceinline__ void store(RecBF *q, uint32_t e0, float a0, float a1, bool two, uint32_t e1, float b0, float b1)
    {
        const RecBF pr{pack_rec(e0, a0, a1), two ? pack_rec(e1, b0, b1) : 0ull};
        // (streaming stores: the queues are read back by the reducer a whole launch later; producer 21.2 -> 20.5 ms)
#ifndef RSDF_Q_PLAIN_STORE
        __builtin_nontemporal_store(pr.r0, &q->r0);
        __builtin_nontemporal_store(pr.r1, &q->r1);
#else
        *q = pr;
#endif
    }
    static __device__ __forceinline__ RecBF load(const RecBF *q)
    {
#ifdef RSDF_Q_NT_LOAD
        RecBF r;
        r.r0 = __builtin_nontemporal_load(&q->r0);
        r.r1 = __builtin_nontemporal_load(&q->r1);
        return r;
#else
        return *q;
#endif
    }
    static __device__ __forceinline__ RecBF empty() { return RecBF{0ull, 0ull}; }
    static __device__ __forceinline__ bool decode(const RecBF &r, uint32_t &e0, double &a0, double &a1, uint32_t &e1, double &b0,
                                                  double &b1)
    {
        unpack_rec(r.r0, e0, a0, a1);
        unpack_rec(r.r1, e1, b0, b1);
        return r.r1 != 0ull;   // a run of odd length ends in a half-empty element
    }
};
constexpr uint32_t PAD_IDX = 0xffffffffu;   // staging slot that pads a run to an even length

struct LevelPlan {
    int count;                       // active levels
    int n_bins[RSDF_MAX_LEVELS];
    int interleaved[RSDF_MAX_LEVELS];  // 0: bin = idx >> 13 (hashed levels); k > 0: n_bins = 2^(k-1), bin = idx & (n_bins-1)
    int n_split[RSDF_MAX_LEVELS];      // reducer workgroups per bin (balances the per-level load)
    int64_t cap[RSDF_MAX_LEVELS];      // queue elements (record PAIRS) per bin
    int64_t queue_off[RSDF_MAX_LEVELS];
    int counter_off[RSDF_MAX_LEVELS];
};

struct LevelGeom {
    float scale;
    uint32_t res, res2, size, mask;
    bool dense;
};

__device__ __forceinline__ LevelGeom level_geom(const rsdf_grid_meta &m, int l)
{
    LevelGeom g;
    g.scale = m.scale[l];
    g.res = m.res[l];
    g.size = m.size[l];
    g.res2 = g.res * g.res;
    g.dense = (uint64_t)g.res * g.res * g.res <= (uint64_t)g.size;
    g.mask = g.size - 1u;
    return g;
}

// identical index function to hashgrid.hip / the oracle; hashed sizes are powers of two (mask)
__device__ __forceinline__ uint32_t entry_index(uint32_t x, uint32_t y, uint32_t z, const LevelGeom &g)
{
    if (g.dense) return (x + y * g.res + z * g.res * g.res) % g.size;
    return ((x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u)) & g.mask;
}

// The 8 corner indices of cell (x, y, z), sharing the multiplications: (y + 1) p == y p + p (mod 2^32), so the hashed
// form needs 2 integer multiplies per cell instead of 16 (v_mul_lo_u32 is a quarter-rate instruction), and the dense form
// replaces "% size" by one conditional subtract: all coordinates are in [0, res], hence x + y res + z res^2 < 2 res^3 <=
// 2 size.  Same values as entry_index() for every corner.
__device__ __forceinline__ void corner_indices(uint32_t x, uint32_t y, uint32_t z, const LevelGeom &g, uint32_t (&idx)[8])
{
    if (g.dense) {
        const uint32_t b = x + y * g.res + z * g.res2;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t v = b + (uint32_t)(c & 1) + ((c >> 1) & 1 ? g.res : 0u) + ((c >> 2) & 1 ? g.res2 : 0u);
            idx[c] = min(v, v - g.size);      // (unsigned: v - size wraps above v when v < size; two instructions, no v_cmp ->
                                              //  s_nop -> v_cndmask chain per corner)
        }
    } else {
        const uint32_t yp0 = y * 2654435761u, zp0 = z * 805459861u;
        const uint32_t xs[2] = {x, x + 1u}, yp[2] = {yp0, yp0 + 2654435761u}, zp[2] = {zp0, zp0 + 805459861u};
#pragma unroll
        for (int c = 0; c < 8; ++c) idx[c] = (xs[c & 1] ^ yp[(c >> 1) & 1] ^ zp[(c >> 2) & 1]) & g.mask;
    }
}

// compact index 0..3 of the two axes other than a
__device__ __forceinline__ int other_bits(int c, int a)
{
    const int o1 = (a == 0) ? 1 : 0, o2 = (a == 2) ? 1 : 2;
    return ((c >> o1) & 1) | (((c >> o2) & 1) << 1);
}

// coordinates of the k-th new corner of axis a: k < 4 -> +side (c0_a + 2), k >= 4 -> -side (c0_a - 1)
__device__ __forceinline__ uint32_t extra_index(const CellFrac &c0, int a, int k, const LevelGeom &g)
{
    const int o1 = (a == 0) ? 1 : 0, o2 = (a == 2) ? 1 : 2;
    uint32_t cc[3] = {c0.c[0], c0.c[1], c0.c[2]};
    cc[a] += (k < 4) ? 2u : 0xffffffffu;
    cc[o1] += (uint32_t)(k & 1);
    cc[o2] += (uint32_t)((k >> 1) & 1);
    return entry_index(cc[0], cc[1], cc[2], g);
}

// Where a sample's seven stencil points come from.
//   x7t != nullptr : read them, x7t [7][S][3] (unit cube), as rsdf_fd_points / rsdf_fd_taps wrote them;
//   x7t == nullptr : DERIVE them from the sample's world-space centre pw [S][3] with exactly rsdf_fd_points' arithmetic
//                    (models/geometry.py:229-244: x +- eps e_k, clamp(-r, r), AABB contraction (q - (-r)) / (r - (-r));
//                    this file is built with -ffp-contract=off and IEEE division like neus.hip), so that the cells and
//                    weights are bit-identical to the x7t form.  Both stencil kernels walk the samples once per level:
//                    16 x 84 B of tap positions per sample become 16 x 12 B (1.3 of the forward's 4.2 KB).
struct TapSrc {
    const float *x7t;
    const float *pw;
    float radius, eps;
    UnitDiv two_r;      // 2 r and RN(1 / 2 r) (hashgrid_common.h unit_div)
};

__device__ __forceinline__ void cell_frac1(float u, float scale, uint32_t &c, float &w)
{
    const float p = fmaf(scale, u, 0.5f);
    const float f = floorf(p);
    c = (uint32_t)(int32_t)f;
    w = p - f;
}

// The centre's unit-cube coordinates u0 (unclamped, as fd_points writes tap 0) and the clamped form uc that every
// displaced tap carries in its two OTHER components (fd_points clamps all three components of a tap; uc differs from u0
// only for a centre that rounding has put outside [-r, r]).  uc needs no second division: (clamp(p, -r, r) + r) / 2r ==
// clamp((p + r) / 2r, 0, 1) bit for bit -- inside the box both are u0; for p > r the left side is 2r / 2r = 1 and
// u0 >= 1 (rounding is monotonic); for p < -r the left side is 0 / 2r = +0 and u0 < 0.
struct CentreUnit {
    float p[3];
    CellFrac c0, cc;
};

__device__ __forceinline__ CentreUnit centre_unit(const TapSrc &src, int64_t s, float scale)
{
    CentreUnit cu;
    const float *pp = src.pw + s * 3;
    const float r = src.radius;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        cu.p[k] = pp[k];
        const float u0 = unit_div(cu.p[k] - (-r), src.two_r);
        cell_frac1(u0, scale, cu.c0.c[k], cu.c0.w[k]);
        cu.cc.c[k] = cu.c0.c[k];
        cu.cc.w[k] = cu.c0.w[k];
        const float uc = fminf(fmaxf(u0, 0.0f), 1.0f);
        if (uc != u0) cell_frac1(uc, scale, cu.cc.c[k], cu.cc.w[k]);   // rare: centre outside the box
    }
    return cu;
}

// tap t = 1..6 (+x,-x,+y,-y,+z,-z) of a derived stencil
__device__ __forceinline__ CellFrac derived_tap(const TapSrc &src, const CentreUnit &cu, int t, float scale)
{
    const int a = (t - 1) >> 1;
    const float off = ((t - 1) & 1) ? -src.eps : src.eps;
    const float r = src.radius;
    CellFrac ct = cu.cc;
    const float q = fminf(fmaxf(cu.p[a] + off, -r), r);
    cell_frac1(unit_div(q - (-r), src.two_r), scale, ct.c[a], ct.w[a]);
    return ct;
}

// the same for a tap index known only at run time (the producer's work items): selects instead of indexed registers
__device__ __forceinline__ CellFrac derived_tap_rt(const TapSrc &src, const CentreUnit &cu, int t, float scale)
{
    const int a = (t - 1) >> 1;
    const float off = ((t - 1) & 1) ? -src.eps : src.eps;
    const float r = src.radius;
    const float pa = a == 0 ? cu.p[0] : (a == 1 ? cu.p[1] : cu.p[2]);
    const float q = fminf(fmaxf(pa + off, -r), r);
    uint32_t c;
    float w;
    cell_frac1(unit_div(q - (-r), src.two_r), scale, c, w);
    CellFrac ct = cu.cc;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (k == a) {
            ct.c[k] = c;
            ct.w[k] = w;
        }
    return ct;
}

template <bool DERIVE>
__device__ __forceinline__ void load_stencil(const TapSrc &src, int64_t S, int64_t s, float scale, CellFrac (&cf)[7])
{
    if (DERIVE) {
        const CentreUnit cu = centre_unit(src, s, scale);
        cf[0] = cu.c0;
#pragma unroll
        for (int t = 1; t < 7; ++t) cf[t] = derived_tap(src, cu, t, scale);
    } else {
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            const float *p = src.x7t + ((int64_t)t * S + s) * 3;
            cf[t] = cell_frac(p[0], p[1], p[2], scale);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// "x2" output of the forward (round 4): the fused SDF-MLP kernels' input image, PRE-SPLIT.  Those kernels (mlp_x2.hip)
// carry every fp32 matrix operand as two fp16 parts, v 2^8 = hi + lo (hi = RNE_f16(v 2^8), lo = RNE_f16(v 2^8 - hi): v to
// 2^-24 relative, 2^-33 absolute near zero).  The split is a deterministic function of the value, so instead of each MLP
// kernel splitting the 36 input columns of all 7 taps again (and the backward running a whole staging pass for it), the
// gather performs it once and writes the parts TILE BY TILE in the layout the MLP kernels keep in LDS -- 2 x 16 bits per
// value, the same bytes as the fp32 planes:
//     x2 [tile = row / 32][tap 7][part 2 (hi, lo)][column 36][32 rows] fp16      (4608 contiguous bytes per tile and tap)
//     column 2 l + f = feature f of level l (levels >= n_active_levels: zeros), columns 32..34 = x, y, z of the tap
//     (unit cube) * xyz_scale + xyz_offset, column 35 = 1 (the bias column), all times 2^8; rows >= n_samples of the last
//     tile are zeros; inside a column with bit 3 of its index set the two 16-row halves are swapped (byte ^ 32): the LDS
//     bank swizzle.
// A column is 64 bytes, a 16-byte unit 8 consecutive rows of one column: the MLP backward lands a tile with linear LDS-DMA
// (no staging pass), and both MLP kernels read layer-1 fragments with ds_read_b64_tr_b16 (columns are the MFMA k
// dimension) and weight-gradient fragments with ds_read_b128 (rows are).  Precondition: |value| < 255 (fp16 range).
// ------------------------------------------------------------------------------------------------
constexpr int X2_COLS = 36;
constexpr int X2_PART_B = X2_COLS * 64;
constexpr float X2_SCALE = 256.0f;
struct X2Out {
    unsigned char *base;
    int64_t Sp;
    float xyz_scale, xyz_offset;
    int parts;          // 2: hi + lo (fp32-equivalent); 1: hi only (the 16-bit mode of mlp_x2.hip)
    int tap_b, tile_b;  // parts * X2_PART_B, 7 * tap_b
};
using x2_f32x2 = __attribute__((ext_vector_type(2))) float;
using x2_f16x2 = __attribute__((ext_vector_type(2))) _Float16;
__device__ __forceinline__ unsigned x2_pack(float a, float b)        // v_cvt_pk_f16_f32 (round to nearest even); a -> low half
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector(x2_f32x2{a, b}, x2_f16x2));
}
// (a, b) already scaled -> hi, lo exactly as mlp_x2.hip's split2_pair forms them
__device__ __forceinline__ void x2_split(float a, float b, unsigned &h, unsigned &l)
{
    h = x2_pack(a, b);
    const x2_f16x2 hh = __builtin_bit_cast(x2_f16x2, h);
    l = x2_pack(a - (float)hh[0], b - (float)hh[1]);
}
// byte address of (row s, tap, part, column)
__device__ __forceinline__ unsigned char *x2_at(const X2Out &o, int64_t s, int tap, int part, int col)
{
    return o.base + (s >> 5) * o.tile_b + tap * o.tap_b + part * X2_PART_B + col * 64 +
           ((2 * (int)(s & 31)) ^ (32 * ((col >> 3) & 1)));
}
// Lane pairs (2 i, 2 i + 1) hold rows (s, s + 1): the even lane ends up with feature 0 of both rows, the odd lane with
// feature 1 of both -- one dword store per lane; the two columns of a level are adjacent, so a wave instruction writes one
// full 128-byte line per tile.
__device__ __forceinline__ unsigned x2_pair_exchange(unsigned own, bool odd)
{
    // lane ^ 1 through DPP quad_perm [1,0,3,2] (a vector-ALU move)
    const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp(0, (int)own, 0xB1, 0xF, 0xF, true);
    // even: {own.lo16, nb.lo16}; odd: {nb.hi16, own.hi16}
    return odd ? __builtin_amdgcn_perm(nb, own, 0x03020706u) : __builtin_amdgcn_perm(nb, own, 0x05040100u);
}
__device__ __forceinline__ void x2_store_features(unsigned char *p, bool odd, float f0, float f1, int parts)
{
    unsigned h, lo;
    x2_split(f0 * X2_SCALE, f1 * X2_SCALE, h, lo);
    // the image (1008 B per sample, 16 GB per launch) is read back by the MLP forward a whole launch later: streaming stores
    // leave the L2 / MALL to the table and the centres (gather 12.9 -> 12.1 ms per launch; -DRSDF_X2_PLAIN_IMAGE for A/B)
#ifndef RSDF_X2_PLAIN_IMAGE
    __builtin_nontemporal_store(x2_pair_exchange(h, odd), reinterpret_cast<unsigned *>(p));
    if (parts == 2) __builtin_nontemporal_store(x2_pair_exchange(lo, odd), reinterpret_cast<unsigned *>(p + X2_PART_B));
#else
    *reinterpret_cast<unsigned *>(p) = x2_pair_exchange(h, odd);
    if (parts == 2) *reinterpret_cast<unsigned *>(p + X2_PART_B) = x2_pair_exchange(lo, odd);    // (uniform)
#endif
}
// unit-cube coordinates of the 7 stencil points, as rsdf_fd_points writes them (neus.hip fd_points_kernel), * scale + offset,
// and the bias column: written as two more "levels" -- (x, y) into columns 32 / 33 and (z, 1) into 34 / 35 -- through the
// same pair exchange as the features (two dword stores per lane, tap and part instead of four 2-byte ones: the level-0
// workgroups that carry these stores were a millisecond of the launch's tail)
template <bool DERIVE>
__device__ __forceinline__ void x2_store_points(const X2Out &o, const TapSrc &src, int64_t S, int64_t s, int64_t sl, bool pad)
{
    const bool odd = (threadIdx.x & 1) != 0;
    const unsigned loff = (threadIdx.x >> 5) * (unsigned)o.tile_b + (unsigned)(32 + (int)(threadIdx.x & 1)) * 64u +
                          (unsigned)(2 * (int)(threadIdx.x & 30));
    unsigned char *bb = o.base + (s - threadIdx.x) / 32 * o.tile_b;
    const float one = pad ? 0.0f : 1.0f;
    float u[7][3];
    if (DERIVE) {
        const float r = src.radius;
        float p[3], uc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            p[k] = src.pw[sl * 3 + k];
            u[0][k] = unit_div(p[k] - (-r), src.two_r);
            uc[k] = unit_div(fminf(fmaxf(p[k], -r), r) - (-r), src.two_r);
        }
#pragma unroll
        for (int t = 1; t < 7; ++t) {
            const int a = (t - 1) >> 1;
            const float off = ((t - 1) & 1) ? -src.eps : src.eps;
#pragma unroll
            for (int k = 0; k < 3; ++k) u[t][k] = k == a ? unit_div(fminf(fmaxf(p[k] + off, -r), r) - (-r), src.two_r) : uc[k];
        }
    } else {
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
            for (int k = 0; k < 3; ++k) u[t][k] = src.x7t[((int64_t)t * S + sl) * 3 + k];
    }
#pragma unroll
    for (int t = 0; t < 7; ++t) {
        float v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = pad ? 0.0f : u[t][k] * o.xyz_scale + o.xyz_offset;
        x2_store_features(bb + t * o.tap_b + loff, odd, v[0], v[1], o.parts);
        x2_store_features(bb + t * o.tap_b + loff + 128, odd, v[2], one, o.parts);
    }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
#ifndef RSDF_FWD_WAVES
#define RSDF_FWD_WAVES 3
#endif
#ifndef RSDF_FWD_GROUP
#define RSDF_FWD_GROUP 4096
#endif
#ifndef RSDF_BWD_GROUP
#define RSDF_BWD_GROUP 64      // producer: 64 x 1024 samples (0.8 MB of centres) per group: the 15 re-reads of a group's centres hit
#endif                         // in the XCD's L2 (256: they come back from the MALL, +150 B per sample of fetch traffic; same time)
using pk_f32x2 = __attribute__((ext_vector_type(2))) float;
// acc += w v on both features: the same two fmas, written as a packed one (this file is built with -fno-slp-vectorize:
// packed fp32 arithmetic that the compiler forms by itself costs more than it saves on gfx950 -- producer 21.8 -> 21.4 ms
// without it -- but the gather's accumulation is faster packed: 12.26 vs 12.40 ms)
__device__ __forceinline__ void fma2(float2 &acc, float w, const float2 &v)
{
#ifdef RSDF_FWD_SCALAR_FMA
    acc.x = fmaf(w, v.x, acc.x);
    acc.y = fmaf(w, v.y, acc.y);
#else
    const pk_f32x2 r = __builtin_elementwise_fma(pk_f32x2{w, w}, pk_f32x2{v.x, v.y}, pk_f32x2{acc.x, acc.y});
    acc.x = r[0];
    acc.y = r[1];
#endif
}

template <bool DERIVE, bool X3>
__device__ __forceinline__ void fd7_fwd_sample(const TapSrc &src, const float2 *__restrict__ tl,
                                               const LevelGeom &g, int64_t S, int64_t s, int l,
                                               float2 *__restrict__ planes, const X2Out &x3)
{
    // x3 form: rows S .. Sp - 1 are padding -- they gather the last sample's cells (valid loads) and store zeros
    const bool pad = X3 && s >= S;
    const int64_t sl = pad ? S - 1 : s;
    CellFrac cf[7];
    load_stencil<DERIVE>(src, S, sl, g.scale, cf);
    const CellFrac &c0 = cf[0];
    int da[7];
    bool plus[3] = {false, false, false}, minus[3] = {false, false, false};
    da[0] = 0;
#pragma unroll
    for (int t = 1; t < 7; ++t) {
        const int a = (t - 1) >> 1;
        da[t] = (int32_t)(cf[t].c[a] - c0.c[a]);
        if (da[t] == 1) plus[a] = true;
        if (da[t] == -1) minus[a] = true;
    }
    float2 v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
        v[c] = tl[entry_index(c0.c[0] + (c & 1), c0.c[1] + ((c >> 1) & 1), c0.c[2] + ((c >> 2) & 1), g)];
    float2 ex[3][8];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            ex[a][k] = make_float2(0.f, 0.f);
            if ((k < 4) ? plus[a] : minus[a]) ex[a][k] = tl[extra_index(c0, a, k, g)];
        }
    }
#pragma unroll
    for (int t = 0; t < 7; ++t) {
        const int a = t == 0 ? 0 : (t - 1) >> 1;
        float2 acc = make_float2(0.f, 0.f);
        if (t == 0 || da[t] == 0) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float w = corner_weight(cf[t], c);
                fma2(acc, w, v[c]);
            }
        } else if (da[t] == 1) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float w = corner_weight(cf[t], c);
                const float2 val = (((c >> a) & 1) == 0) ? v[c | (1 << a)] : ex[a][other_bits(c, a)];
                fma2(acc, w, val);
            }
        } else if (da[t] == -1) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float w = corner_weight(cf[t], c);
                const float2 val = (((c >> a) & 1) == 1) ? v[c & ~(1 << a)] : ex[a][4 + other_bits(c, a)];
                fma2(acc, w, val);
            }
        } else {  // tap more than one cell away: gather its own 8 corners
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float w = corner_weight(cf[t], c);
                const float2 val = tl[entry_index(cf[t].c[0] + (c & 1), cf[t].c[1] + ((c >> 1) & 1),
                                                  cf[t].c[2] + ((c >> 2) & 1), g)];
                fma2(acc, w, val);
            }
        }
        if (X3) {
            // workgroup-uniform base (the block's first tile; a scalar add per tap) + a 32-bit lane offset: the stores
            // need no vector address arithmetic (global_store ... saddr, parts as immediates)
            const unsigned loff = (threadIdx.x >> 5) * (unsigned)x3.tile_b + (unsigned)(2 * l + (int)(threadIdx.x & 1)) * 64u +
                                  (unsigned)((2 * (int)(threadIdx.x & 30)) ^ (32 * ((l >> 2) & 1)));
            unsigned char *bb = x3.base + (s - threadIdx.x) / 32 * x3.tile_b + t * x3.tap_b;
            x2_store_features(bb + loff, (threadIdx.x & 1) != 0, pad ? 0.0f : acc.x, pad ? 0.0f : acc.y, x3.parts);
        }
        else planes[((int64_t)l * 7 + t) * S + s] = acc;
    }
#ifndef RSDF_X2_NOXYZ
    if (X3 && l == 0) x2_store_points<DERIVE>(x3, src, S, s, sl, pad);
#endif
}

// 1-D grid in SAMPLE-GROUP-MAJOR order: all active levels of RSDF_FWD_GROUP consecutive tiles (4096 x 256 = 1 M samples:
// 88 MB of x7t), then the next group.  With the level as the slow grid dimension every level's pass re-read the whole
// 1.6 GB of x7t from HBM (43 % of the kernel's fetch traffic, and the kernel runs at the ~4 TB/s the memory system gives
// it); inside a group the 15 re-reads come from the 256 MB MALL.  18.8 -> 16.0 ms per launch; 1024-tile groups 16.9,
// 8192-tile groups (176 MB + the 64 MB of tables: past the MALL) 18.8.  Not paying, measured: XCD-contiguous sample
// ranges (neighbouring rays in one L2: <= 3 %), several tiles per workgroup, forcing more waves per SIMD.
#ifndef RSDF_X2_WAVES
#define RSDF_X2_WAVES 3     // (4 waves per SIMD = 128 registers spill 10 of them: 13.3 against 12.9 ms per launch)
#endif
template <bool DERIVE, bool X3>
__global__ void __launch_bounds__(F_THREADS, X3 ? RSDF_X2_WAVES : RSDF_FWD_WAVES)
fd7_fwd_kernel(const TapSrc src, const float *__restrict__ table,
               const rsdf_grid_meta meta, int64_t S, int n_active, float2 *__restrict__ planes, const X2Out x3, int group)
{
    // group = min(RSDF_FWD_GROUP, tiles of the launch): a training step's 250 k samples are 1 k tiles, and a grid padded to
    // 4096 tiles x 16 levels spent 1.3 ms per step dispatching empty workgroups
    const int64_t per_group = (int64_t)group * n_active;
    const int64_t grp = blockIdx.x / per_group, r = blockIdx.x - grp * per_group;
    const int l = (int)(r / group);
    const int64_t tile = grp * group + (r - (int64_t)l * group);
    const int64_t s = tile * F_THREADS + threadIdx.x;
    if (s >= (X3 ? x3.Sp : S)) return;      // (Sp is a multiple of 32: lane pairs stay whole)
    const LevelGeom g = level_geom(meta, l);
    const float2 *tl = reinterpret_cast<const float2 *>(table) + meta.offset[l];
    fd7_fwd_sample<DERIVE, X3>(src, tl, g, S, s, l, planes, x3);
}

// zero the columns of the levels that are not active (or do not exist): the MLP kernels read all 32 feature columns.
// One thread per 16-byte unit: (tile, tap, part) blocks of (32 - col0) columns x 4 units.
__global__ void __launch_bounds__(256)
x3_zero_cols_kernel(const X2Out x3, int col0)
{
    const int per = (32 - col0) * 4;
    const int64_t n = x3.Sp / 32 * 7 * x3.parts * per;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t blk = i / per;
        const int u = (int)(i - blk * per);
        *reinterpret_cast<uint4 *>(x3.base + blk * X2_PART_B + col0 * 64 + u * 16) = uint4{0u, 0u, 0u, 0u};
    }
}

// in-kernel stamps (build with -DRSDF_STAMPS; tools/stamps_produce.sh): thread 0 of the workgroups of level RSDF_STAMP_LEVEL
// add the s_memtime cycles between consecutive stamp points to g_pstamps[point] (atomically: many workgroups)
#ifdef RSDF_STAMPS
#ifndef RSDF_STAMP_LEVEL
#define RSDF_STAMP_LEVEL 12
#endif
__device__ unsigned long long g_pstamps[16];
struct StamperP {
    unsigned long long last;
    bool on;
    __device__ __forceinline__ void begin(bool enable) { on = enable; last = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void at(int i)
    {
        const unsigned long long t = __builtin_readcyclecounter();
        if (on) atomicAdd(&g_pstamps[i], t - last);
        last = __builtin_readcyclecounter();
    }
};
#define RSDF_PSTAMP(st, i) (st).at(i)
#else
struct StamperP {
    __device__ __forceinline__ void begin(bool) {}
};
#define RSDF_PSTAMP(st, i)
#endif

// a += w g, one fma per feature
__device__ __forceinline__ void axpy2(float2 &a, float w, pk_f32x2 g)
{
#ifndef RSDF_AXPY_PACKED      // (as a packed fma the producer needs 54 spilled registers instead of 13: 23.9 vs 21.4 ms)
    a.x = fmaf(w, g[0], a.x);
    a.y = fmaf(w, g[1], a.y);
#else
    const pk_f32x2 r = __builtin_elementwise_fma(pk_f32x2{w, w}, g, pk_f32x2{a.x, a.y});
    a.x = r[0];
    a.y = r[1];
#endif
}

// d_planes are read once per launch by phase 1 (the displaced taps of phase 2 re-read a few of them from L2)
__device__ __forceinline__ float2 ld_dplane(const float2 *p)
{
#ifdef RSDF_DPL_NT_LOAD
    const unsigned long long v = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long *>(p));
    return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
#else
    return *p;
#endif
}

// ------------------------------------------------------------------------------------------------
// backward: produce
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int bin_of(uint32_t idx, int n_bins, int interleaved)
{
    return interleaved ? (int)(idx & (uint32_t)(n_bins - 1)) : (int)(idx >> BIN_SHIFT);   // no division on this path
}

__device__ __forceinline__ uint32_t entry_of(uint32_t idx, int interleaved)
{
    return interleaved ? idx >> (interleaved - 1) : (idx & (uint32_t)(BIN_ENTRIES - 1));
}

// ridx is clobbered: a record's rank within its bin (< 8192: 13 bits) rides in the upper bits of its entry index (< 2^19:
// make_plan admits at most MAX_BINS * BIN_ENTRIES entries per level) -- eight registers less across the barriers, in a
// kernel that is held to 64 registers by its occupancy
constexpr int IDX_BITS = 19;
static_assert((1u << IDX_BITS) == (unsigned)MAX_BINS * BIN_ENTRIES, "entry index + rank must fit 32 bits");
constexpr uint32_t IDX_MASK = (1u << IDX_BITS) - 1u;

template <typename REC>
__device__ __forceinline__ void emit_round(uint32_t (&ridx)[ROUND_RECS],
                                           const float2 (&rval)[ROUND_RECS], uint32_t valid_mask,
                                           int n_bins, int interleaved, int64_t cap,
                                           REC *__restrict__ queue, int *__restrict__ qcount,
                                           float *__restrict__ dlevel, int *s_cnt, int *s_off,
                                           int *s_gbase, Record *s_stage, StamperP &stp)
{
    // s_cnt is all zero on entry (zeroed by the kernel prologue / the previous round's scan).
    // Three barriers per non-empty round (four when it needs a second staging pass), two per empty one:
    //   slots -> A -> scan (+ re-zero counts) -> B -> stage -> C -> copy out [-> D -> stage -> C -> copy out].
    // No barrier is needed after the copy-out: the next round's scan (which overwrites s_off /
    // s_gbase) runs after its barrier A, and its stage writes after its barrier B, both of which
    // every thread reaches only after finishing this copy-out.
    const int tid = threadIdx.x;
#pragma unroll
    for (int r = 0; r < ROUND_RECS; ++r)
        if (valid_mask & (1u << r))
            ridx[r] |= (uint32_t)atomicAdd(&s_cnt[bin_of(ridx[r], n_bins, interleaved)], 1) << IDX_BITS;
    RSDF_PSTAMP(stp, 4);   // slot atomics
    __syncthreads();  // A
    RSDF_PSTAMP(stp, 5);   // barrier A
    // First wavefront: exclusive scan of the bin counts + queue reservations.  Every bin's run is padded to an EVEN length
    // in the staging order, so that records 2 p, 2 p + 1 of the staging buffer always belong to one bin (or the second is
    // the pad) and a staging-pass boundary never splits a pair.  The reservation is a RETURNING global
    // atomic (~1-2 us under load); its result is only needed by the copy-out, so it is parked in a register here and
    // published to s_gbase after this wave has staged its own records: the round trip overlaps barrier B and the staging
    // instead of holding all sixteen waves at B.
    int gbase = 0, padpos = -1;
    if (tid < 64) {
        const int c = tid < n_bins ? s_cnt[tid] : 0;
        const int cp = (c + 1) & ~1;
        const int incl = wave_incl_sum_i(cp);
        if (tid < n_bins) {
            s_off[tid] = incl - cp;
            gbase = c > 0 ? atomicAdd(&qcount[tid], cp >> 1) : 0;
            s_cnt[tid] = 0;
            if (c & 1) padpos = incl - 1;
        }
        if (tid == 63) s_off[MAX_BINS] = incl;
    }
    RSDF_PSTAMP(stp, 6);   // scan + queue reservation (first wave)
    __syncthreads();  // B
    RSDF_PSTAMP(stp, 7);   // barrier B
    const int total = s_off[MAX_BINS];   // even
    if (total == 0) return;  // uniform (no reservation was made: every count was zero)
    // (measured and not kept, round 6: the bin's absolute queue offset b * cap + gbase published here as a 64-bit LDS word so
    // that the copy-out forms no 64-bit product per record pair, a 24-bit multiply for the staging stride, a 32 x 32-bit
    // product for the displaced taps' plane index: 20.9 -> 21.3 ms, +0.0, +0.2 -- the kernel is held to 64 registers and
    // every one of these moved its spills)
    if (tid < n_bins) s_gbase[tid] = gbase;
    for (int lo = 0; lo < total; lo += STAGE_CAP) {   // uniform; bins stay contiguous: a pass boundary splits one run
        const int n = min(total - lo, STAGE_CAP);     // even
        if (lo) __syncthreads();  // D: the previous pass has left the staging buffer
#pragma unroll
        for (int r = 0; r < ROUND_RECS; ++r)
            if (valid_mask & (1u << r)) {
                const uint32_t idx = ridx[r] & IDX_MASK;
                const unsigned pos = (unsigned)(s_off[bin_of(idx, n_bins, interleaved)] + (int)(ridx[r] >> IDX_BITS) - lo);
                if (pos < (unsigned)n) s_stage[pos] = Record{idx, rval[r].x, rval[r].y};
            }
        if ((unsigned)(padpos - lo) < (unsigned)n) s_stage[padpos - lo].idx = PAD_IDX;
        RSDF_PSTAMP(stp, 8);   // staging writes
        __syncthreads();  // C
        RSDF_PSTAMP(stp, 9);   // barrier C (+ D)
        for (int i = tid; 2 * i < n; i += P_THREADS) {
            const Record r0 = s_stage[2 * i], r1 = s_stage[2 * i + 1];
            const int b = bin_of(r0.idx, n_bins, interleaved);
            const bool two = r1.idx != PAD_IDX;
            const int64_t gpos = (int64_t)s_gbase[b] + ((lo + 2 * i - s_off[b]) >> 1);
            if (gpos < cap) {      // (as a 32-bit compare: no change, 20.68 vs 20.69 ms)
                RecOps<REC>::store(&queue[(int64_t)b * cap + gpos], entry_of(r0.idx, interleaved), r0.v0, r0.v1, two,
                                   two ? entry_of(r1.idx, interleaved) : 0u, r1.v0, r1.v1);
            } else {  // queue full (capacity carries slack; never drop a contribution)
                atomicAdd(dlevel + 2 * (size_t)r0.idx, r0.v0);
                atomicAdd(dlevel + 2 * (size_t)r0.idx + 1, r0.v1);
                if (two) {
                    atomicAdd(dlevel + 2 * (size_t)r1.idx, r1.v0);
                    atomicAdd(dlevel + 2 * (size_t)r1.idx + 1, r1.v1);
                }
            }
        }
        RSDF_PSTAMP(stp, 10);  // copy-out
    }
}

// Work item of the second phase: a displaced tap.  bits 0..9 sample within the workgroup, 10..12 tap - 1, 13 side (1: +1 cell)
constexpr int MAX_ITEMS = P_THREADS * 6;

template <bool DERIVE, typename REC>
__global__ void __launch_bounds__(P_THREADS, RSDF_STAGE_RECS <= 4 ? 8 : 4)
fd7_produce_kernel(const TapSrc src, const float2 *__restrict__ dplanes,
                   const rsdf_grid_meta meta, const LevelPlan plan, int64_t S, int n_active,
                   REC *__restrict__ queues, int *__restrict__ counters,
                   float *__restrict__ dtable)
{
    __shared__ int s_cnt[MAX_BINS];
    __shared__ int s_off[MAX_BINS + 1];
    __shared__ int s_gbase[MAX_BINS];
    __shared__ int s_nitems;
    __shared__ unsigned short s_items[MAX_ITEMS];
    extern __shared__ __attribute__((aligned(16))) Record s_stage[];  // [STAGE_CAP]
    if (threadIdx.x < MAX_BINS) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_nitems = 0;
    __syncthreads();

    // 1-D grid in sample-group-major order (see the forward): all active levels of RSDF_BWD_GROUP consecutive tiles, then
    // the next group, so that the re-reads of a group's stencil centres come from the L2 (RSDF_BWD_GROUP)
    const int64_t per_group = (int64_t)RSDF_BWD_GROUP * n_active;
    const int64_t grp = blockIdx.x / per_group, rem = blockIdx.x - grp * per_group;
    const int l = (int)(rem / RSDF_BWD_GROUP);
    const int64_t tile_id = grp * RSDF_BWD_GROUP + (rem - (int64_t)l * RSDF_BWD_GROUP);
    const LevelGeom g = level_geom(meta, l);
    float *dlevel = dtable + (size_t)meta.offset[l] * 2;
    REC *queue = queues + plan.queue_off[l];
    int *qcount = counters + plan.counter_off[l];
    const int n_bins = plan.n_bins[l], interleaved = plan.interleaved[l];
    const int64_t cap = plan.cap[l];

    const int64_t s_block = tile_id * P_THREADS;
    if (s_block >= S) return;   // padding tile of the last group (uniform)
    const int64_t s = s_block + threadIdx.x;
    const bool active = s < S;
    StamperP stp;
#ifdef RSDF_STAMPS
    stp.begin(threadIdx.x == 0 && l == RSDF_STAMP_LEVEL);
    if (threadIdx.x == 0 && l == RSDF_STAMP_LEVEL) atomicAdd(&g_pstamps[15], 1ull);
#endif

    // ---- phase 1: one thread per sample.  The centre cell's 8 corners collect the centre tap, every tap that stays in
    // the cell, and the shared face of every tap that moved to a face neighbour.  The 4 NEW corners of such a displaced
    // tap are not computed here: the tap is queued as a work item, so that their cost follows the number of displaced
    // taps (~ eps / cell: a few per cent at the coarse levels) instead of being paid by every thread at every level.
    float2 acc[8];
    CellFrac c0;
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = make_float2(0.f, 0.f);
    c0.c[0] = c0.c[1] = c0.c[2] = 0;
    unsigned items = 0;   // 6 x 2 bits: bit 2(t-1) = displaced, bit 2(t-1)+1 = side

    if (active) {
        const float2 *pg = dplanes + ((int64_t)l * 7) * S + s;          // tap t: pg + t S
        CellFrac cf[7];
        load_stencil<DERIVE>(src, S, s, g.scale, cf);
        {
            c0 = cf[0];
            const float2 gr = ld_dplane(pg);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float w = corner_weight(c0, c);
                acc[c].x = w * gr.x;
                acc[c].y = w * gr.y;
            }
        }
#ifndef RSDF_PRODUCE_PHASE1_R5
        // (round 6) Axis by axis: the two taps of an axis share the weights of the OTHER two axes (every tap carries the
        // clamped centre's fractions there), so a tap's eight weights are four shared products times its own pair -- 8
        // multiplies per tap instead of 15 -- and the accumulation is an fma per feature instead of a multiply and an add
        // (this file is built with -ffp-contract=off: only explicit fmas fuse).  Against the
        // round-5 form (-DRSDF_PRODUCE_PHASE1_R5) a weight is (w_o1 w_o2) w_a instead of (w_x w_y) w_z and the sum is fused:
        // ulp-level differences in a gradient that the records quantise at 2^-20 anyway.  The kernel is vector-issue bound
        // (0.80 busy at 7.5 waves per SIMD, profiles/r06d_final/pmc/pipes.json).
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int o1 = (a == 0) ? 1 : 0, o2 = (a == 2) ? 1 : 2;
            float oth[4];
            {
                const CellFrac &cp = cf[2 * a + 1];
                const float u1[2] = {1.0f - cp.w[o1], cp.w[o1]}, u2[2] = {1.0f - cp.w[o2], cp.w[o2]};
#pragma unroll
                for (int k = 0; k < 4; ++k) oth[k] = u1[k & 1] * u2[k >> 1];
            }
#pragma unroll
            for (int sd = 0; sd < 2; ++sd) {
                const int t = 2 * a + 1 + sd;
                const CellFrac ct = cf[t];
                const float2 gr = ld_dplane(pg + (int64_t)t * S);
                const pk_f32x2 g2 = {gr.x, gr.y};
                const int32_t da = (int32_t)(ct.c[a] - c0.c[a]);
                const float wa[2] = {1.0f - ct.w[a], ct.w[a]};
                if (da == 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int b = 0; b < 2; ++b) axpy2(acc[(b << a) | ((k & 1) << o1) | ((k >> 1) << o2)], oth[k] * wa[b], g2);
                } else if (da == 1) {
                    // the tap's near face (its corners with bit a == 0) coincides with the centre cell's far face (bit a == 1)
                    items |= 3u << (2 * (t - 1));
#pragma unroll
                    for (int k = 0; k < 4; ++k) axpy2(acc[(1 << a) | ((k & 1) << o1) | ((k >> 1) << o2)], oth[k] * wa[0], g2);
                } else if (da == -1) {
                    items |= 1u << (2 * (t - 1));
#pragma unroll
                    for (int k = 0; k < 4; ++k) axpy2(acc[((k & 1) << o1) | ((k >> 1) << o2)], oth[k] * wa[1], g2);
                } else {  // tap more than one cell away (eps larger than a cell): rare slow path
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const float w = corner_weight(ct, c);
                        const uint32_t idx = entry_index(ct.c[0] + (c & 1), ct.c[1] + ((c >> 1) & 1),
                                                         ct.c[2] + ((c >> 2) & 1), g);
                        atomicAdd(dlevel + 2 * (size_t)idx, w * gr.x);
                        atomicAdd(dlevel + 2 * (size_t)idx + 1, w * gr.y);
                    }
                }
            }
        }
#else
#pragma unroll
        for (int t = 1; t < 7; ++t) {
            const int a = (t - 1) >> 1;
            const CellFrac ct = cf[t];
            const float2 gr = ld_dplane(pg + (int64_t)t * S);
            const int32_t da = (int32_t)(ct.c[a] - c0.c[a]);
            if (da == 0) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float w = corner_weight(ct, c);
                    acc[c].x += w * gr.x;
                    acc[c].y += w * gr.y;
                }
            } else if (da == 1 || da == -1) {
                // the tap's near face (its corners with bit a == (da < 0)) coincides with the centre cell's far face
                items |= (da == 1 ? 3u : 1u) << (2 * (t - 1));
                const int near = da == 1 ? 0 : 1;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (((c >> a) & 1) == near) {
                        const float w = corner_weight(ct, c);
                        const int cc = c ^ (1 << a);
                        acc[cc].x += w * gr.x;
                        acc[cc].y += w * gr.y;
                    }
                }
            } else {  // tap more than one cell away (eps larger than a cell): rare slow path
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float w = corner_weight(ct, c);
                    const uint32_t idx = entry_index(ct.c[0] + (c & 1), ct.c[1] + ((c >> 1) & 1),
                                                     ct.c[2] + ((c >> 2) & 1), g);
                    atomicAdd(dlevel + 2 * (size_t)idx, w * gr.x);
                    atomicAdd(dlevel + 2 * (size_t)idx + 1, w * gr.y);
                }
            }
        }
#endif
    }
    RSDF_PSTAMP(stp, 0);   // loads + phase-1 weights
    {   // append this thread's displaced taps to the workgroup's work list
        const int cnt = __popc(items & 0x555u);
        int pos = cnt ? atomicAdd(&s_nitems, cnt) : 0;
#pragma unroll
        for (int t = 0; t < 6; ++t)
            if (items >> (2 * t) & 1u)
                s_items[pos++] = (unsigned short)(threadIdx.x | (t << 10) | ((items >> (2 * t + 1) & 1u) << 13));
    }

    RSDF_PSTAMP(stp, 1);   // work-list append
    // Cross-sample merge at coarse levels: consecutive samples of a ray (= consecutive lanes) share their centre
    // cell for ~cell/step samples (18 at level 0 of the 32..2048 grid, < 2 from level 6 up).  When a wavefront holds
    // few distinct cells, sum each run of equal cells with a segmented wave scan and let only the run's last lane
    // emit: up to 18x fewer records for that level's queue traffic and reducer work.
    bool emit0 = active;
    {
        const int lane = lane_id();
        const uint32_t kx = active ? c0.c[0] : 0xffffffffu, ky = active ? c0.c[1] : (uint32_t)lane, kz = c0.c[2];
        const uint32_t px = __shfl_up(kx, 1, 64), py = __shfl_up(ky, 1, 64), pz = __shfl_up(kz, 1, 64);
        int head = (lane == 0 || px != kx || py != ky || pz != kz) ? 1 : 0;
        const int n_runs = __popcll(__ballot(head));
        if (n_runs <= MERGE_MAX_RUNS) {  // wave-uniform
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                float2 u[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    u[c].x = __shfl_up(acc[c].x, o, 64);
                    u[c].y = __shfl_up(acc[c].y, o, 64);
                }
                const int hu = __shfl_up(head, o, 64);
                if (lane >= o && !head) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        acc[c].x += u[c].x;
                        acc[c].y += u[c].y;
                    }
                    head = hu;
                }
            }
            const uint32_t nx = __shfl_down(kx, 1, 64), ny = __shfl_down(ky, 1, 64), nz = __shfl_down(kz, 1, 64);
            emit0 = active && (lane == 63 || nx != kx || ny != ky || nz != kz);
        }
    }
    RSDF_PSTAMP(stp, 2);   // run merge
    uint32_t ridx[ROUND_RECS];
    corner_indices(c0.c[0], c0.c[1], c0.c[2], g, ridx);
    RSDF_PSTAMP(stp, 3);   // corner indices
    emit_round<REC>(ridx, acc, emit0 ? 0xffu : 0u, n_bins, interleaved, cap, queue, qcount, dlevel, s_cnt, s_off,
               s_gbase, s_stage, stp);   // (its barriers also publish the work list)

    // ---- phase 2: dense over the displaced taps, two per thread and round (4 new corners each)
    const int n_items = s_nitems;
    for (int base = 0; base < n_items; base += 2 * P_THREADS) {
        uint32_t mask = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = base + j * P_THREADS + (int)threadIdx.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ridx[4 * j + k] = 0;
                acc[4 * j + k] = make_float2(0.f, 0.f);
            }
            if (i < n_items) {
                const unsigned it = s_items[i];
                const int64_t s2 = s_block + (it & 1023u);
                const int t = (int)(it >> 10 & 7u) + 1, a = (t - 1) >> 1;
                const bool far = (it >> 13 & 1u) != 0;
                CellFrac ct;
                if (DERIVE) {
                    ct = derived_tap_rt(src, centre_unit(src, s2, g.scale), t, g.scale);
                } else {
                    const float *p = src.x7t + ((int64_t)t * S + s2) * 3;
                    ct = cell_frac(p[0], p[1], p[2], g.scale);
                }
                const float2 gr = dplanes[((int64_t)l * 7 + t) * S + s2];
#ifdef RSDF_PRODUCE_PHASE2_R5
                const int o1 = (a == 0) ? 1 : 0, o2 = (a == 2) ? 1 : 2;
                uint32_t cidx[8];
                corner_indices(ct.c[0], ct.c[1], ct.c[2], g, cidx);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = ((int)far << a) | ((k & 1) << o1) | ((k >> 1) << o2);   // far face: bit a == (da > 0)
                    const float w = corner_weight(ct, c);
                    ridx[4 * j + k] = cidx[c];
                    acc[4 * j + k] = make_float2(w * gr.x, w * gr.y);
                }
#else
                // The four corners of the tap's FAR face only (bit a == far; k enumerates the other two axes o1 < o2), the
                // axis picked by selects on the per-axis terms instead of on eight finished indices / weights: a corner
                // index is a xor (hashed) or a sum (dense) of one term per axis, a weight a product of one factor per axis.
                // Same indices as corner_indices(); weights as phase 1 forms them, (u_o1 u_o2) u_a.
                uint32_t tx[2], ty[2], tz[2];
                if (g.dense) {
                    const uint32_t b = ct.c[0] + ct.c[1] * g.res + ct.c[2] * g.res2;
                    tx[0] = b, tx[1] = b + 1u, ty[0] = 0u, ty[1] = g.res, tz[0] = 0u, tz[1] = g.res2;
                } else {
                    const uint32_t yp = ct.c[1] * 2654435761u, zp = ct.c[2] * 805459861u;
                    tx[0] = ct.c[0], tx[1] = ct.c[0] + 1u, ty[0] = yp, ty[1] = yp + 2654435761u, tz[0] = zp, tz[1] = zp + 805459861u;
                }
                const uint32_t fx = far ? tx[1] : tx[0], fy = far ? ty[1] : ty[0], fz = far ? tz[1] : tz[0];
                const uint32_t A = a == 0 ? fx : (a == 1 ? fy : fz);
                const uint32_t B[2] = {a == 0 ? ty[0] : tx[0], a == 0 ? ty[1] : tx[1]};      // axis o1
                const uint32_t C[2] = {a == 2 ? ty[0] : tz[0], a == 2 ? ty[1] : tz[1]};      // axis o2
                const float w_a = a == 0 ? ct.w[0] : (a == 1 ? ct.w[1] : ct.w[2]);
                const float w_1 = a == 0 ? ct.w[1] : ct.w[0], w_2 = a == 2 ? ct.w[1] : ct.w[2];
                const float ua = far ? w_a : 1.0f - w_a;
                const float u1[2] = {1.0f - w_1, w_1}, u2[2] = {1.0f - w_2, w_2};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t idx;
                    if (g.dense) {
                        const uint32_t v = A + B[k & 1] + C[k >> 1];
                        idx = min(v, v - g.size);
                    } else {
                        idx = (A ^ B[k & 1] ^ C[k >> 1]) & g.mask;
                    }
                    const float w = (u1[k & 1] * u2[k >> 1]) * ua;
                    ridx[4 * j + k] = idx;
                    acc[4 * j + k] = make_float2(w * gr.x, w * gr.y);
                }
#endif
                mask |= 0xfu << (4 * j);
            }
        }
        RSDF_PSTAMP(stp, 11);  // phase-2 item evaluation
        emit_round<REC>(ridx, acc, mask, n_bins, interleaved, cap, queue, qcount, dlevel, s_cnt, s_off, s_gbase, s_stage, stp);
    }
}

// ------------------------------------------------------------------------------------------------
// The same queue machinery for PLAIN points (one evaluation per point, row-major gradients): the table scatter of the
// generic encoder's backward (MODE 0: value = w_c * dy) and of the input-gradient's backward (MODE 1: value = scale * m_c *
// dy with m_c = sum_d g_d dw_c/dx_d, hashgrid_dx.hip).  The curvature term of the training step (models/geometry.py:246-282)
// sends 2.6e5 points per step through both; as per-corner float atomics they were 3.3 + 2.2 ms of a 32 ms step (the memory
// side executes ~2e10 scattered atomics per second), as records through the bins they cost what 8 records per (point,
// level) cost.  x [n,3] unit cube; dy rows [n, ld_dy], level l at columns col_off + 2 l, + 1.
// ------------------------------------------------------------------------------------------------
template <int MODE, typename REC>
__global__ void __launch_bounds__(P_THREADS, RSDF_STAGE_RECS <= 4 ? 8 : 4)
scatter_produce_kernel(const float *__restrict__ x, const float *__restrict__ dy, int ld_dy, int col_off,
                       const float *__restrict__ gdx, const rsdf_grid_meta meta, const LevelPlan plan, int64_t S,
                       int n_active, REC *__restrict__ queues, int *__restrict__ counters, float *__restrict__ dtable)
{
    __shared__ int s_cnt[MAX_BINS];
    __shared__ int s_off[MAX_BINS + 1];
    __shared__ int s_gbase[MAX_BINS];
    extern __shared__ __attribute__((aligned(16))) Record s_stage[];  // [STAGE_CAP]
    if (threadIdx.x < MAX_BINS) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t per_group = (int64_t)RSDF_BWD_GROUP * n_active;
    const int64_t grp = blockIdx.x / per_group, rem = blockIdx.x - grp * per_group;
    const int l = (int)(rem / RSDF_BWD_GROUP);
    const int64_t tile_id = grp * RSDF_BWD_GROUP + (rem - (int64_t)l * RSDF_BWD_GROUP);
    const LevelGeom g = level_geom(meta, l);
    float *dlevel = dtable + (size_t)meta.offset[l] * 2;
    REC *queue = queues + plan.queue_off[l];
    int *qcount = counters + plan.counter_off[l];
    const int n_bins = plan.n_bins[l], interleaved = plan.interleaved[l];
    const int64_t cap = plan.cap[l];
    const int64_t s_block = tile_id * P_THREADS;
    if (s_block >= S) return;   // padding tile of the last group (uniform)
    const int64_t s = s_block + threadIdx.x;
    const bool active = s < S;
    StamperP stp;
    stp.begin(false);

    float2 acc[8];
    CellFrac c0;
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = make_float2(0.f, 0.f);
    c0.c[0] = c0.c[1] = c0.c[2] = 0;
    if (active) {
        c0 = cell_frac(x[3 * s], x[3 * s + 1], x[3 * s + 2], g.scale);
        const float *d = dy + s * ld_dy + col_off + 2 * l;
        const float d0 = d[0], d1 = d[1];
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        if (MODE == 1) { g0 = gdx[3 * s]; g1 = gdx[3 * s + 1]; g2 = gdx[3 * s + 2]; }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float w;
            if (MODE == 0) {
                w = corner_weight(c0, c);
                acc[c] = make_float2(w * d0, w * d1);
            } else {
                // hashgrid_dx.hip: m_c = sum_d g_d sgn_d(c) prod_{k != d} omega_k(c_k); contribution scale * dy * m_c,
                // formed in the atomic kernel's order (scale * dy) * m
                const float o0 = (c & 1) ? c0.w[0] : 1.0f - c0.w[0], o1 = (c & 2) ? c0.w[1] : 1.0f - c0.w[1],
                            o2 = (c & 4) ? c0.w[2] : 1.0f - c0.w[2];
                const float s0 = (c & 1) ? 1.0f : -1.0f, s1 = (c & 2) ? 1.0f : -1.0f, s2 = (c & 4) ? 1.0f : -1.0f;
                w = g0 * s0 * o1 * o2 + g1 * s1 * o0 * o2 + g2 * s2 * o0 * o1;
                acc[c] = make_float2(g.scale * d0 * w, g.scale * d1 * w);
            }
        }
    }
    // cross-sample merge of runs of equal cells (consecutive samples of a ray at coarse levels), as fd7_produce_kernel
    bool emit0 = active;
    {
        const int lane = lane_id();
        const uint32_t kx = active ? c0.c[0] : 0xffffffffu, ky = active ? c0.c[1] : (uint32_t)lane, kz = c0.c[2];
        const uint32_t px = __shfl_up(kx, 1, 64), py = __shfl_up(ky, 1, 64), pz = __shfl_up(kz, 1, 64);
        int head = (lane == 0 || px != kx || py != ky || pz != kz) ? 1 : 0;
        const int n_runs = __popcll(__ballot(head));
        if (n_runs <= MERGE_MAX_RUNS) {  // wave-uniform
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                float2 u[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    u[c].x = __shfl_up(acc[c].x, o, 64);
                    u[c].y = __shfl_up(acc[c].y, o, 64);
                }
                const int hu = __shfl_up(head, o, 64);
                if (lane >= o && !head) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        acc[c].x += u[c].x;
                        acc[c].y += u[c].y;
                    }
                    head = hu;
                }
            }
            const uint32_t nx = __shfl_down(kx, 1, 64), ny = __shfl_down(ky, 1, 64), nz = __shfl_down(kz, 1, 64);
            emit0 = active && (lane == 63 || nx != kx || ny != ky || nz != kz);
        }
    }
    uint32_t ridx[ROUND_RECS];
    corner_indices(c0.c[0], c0.c[1], c0.c[2], g, ridx);
    emit_round<REC>(ridx, acc, emit0 ? 0xffu : 0u, n_bins, interleaved, cap, queue, qcount, dlevel, s_cnt, s_off, s_gbase,
               s_stage, stp);
}

// ------------------------------------------------------------------------------------------------
// backward: reduce
// ------------------------------------------------------------------------------------------------
template <typename REC>
__global__ void __launch_bounds__(R_THREADS)
fd7_reduce_kernel(const rsdf_grid_meta meta, const LevelPlan plan, const REC *__restrict__ queues,
                  const int *__restrict__ counters, float *__restrict__ dtable)
{
    extern __shared__ __attribute__((aligned(16))) double s_acc[];  // [BIN_ENTRIES][2], fp64: see below
    const int l = blockIdx.y;
    const int n_split = plan.n_split[l];
    const int b = blockIdx.x / n_split, part = blockIdx.x % n_split;
    const int n_bins = plan.n_bins[l];
    if (b >= n_bins) return;
    const int interleaved = plan.interleaved[l];
    const int64_t cap = plan.cap[l];
    int64_t count = counters[plan.counter_off[l] + b];
    if (count > cap) count = cap;
    const int64_t per = (count + n_split - 1) / n_split;
    const int64_t r0 = part * per;
    int64_t r1 = r0 + per;
    if (r1 > count) r1 = count;
    if (r0 >= r1) return;
    const uint32_t size = meta.size[l];
    // entries this bin owns
    const int lg = interleaved - 1;   // log2(n_bins) of an interleaved level
    const int entries = interleaved ? (int)((size - (uint32_t)b + (uint32_t)n_bins - 1u) >> lg)
                                    : (int)(size < (uint32_t)BIN_ENTRIES ? size : (uint32_t)BIN_ENTRIES);

    for (int i = threadIdx.x; i < entries * 2; i += R_THREADS) s_acc[i] = 0.0;
    __syncthreads();
    const REC *q = queues + plan.queue_off[l] + (int64_t)b * cap;
    for (int64_t i0 = r0 + threadIdx.x; i0 < r1; i0 += (int64_t)R_THREADS * R_UNROLL) {
        REC rec[R_UNROLL];
#pragma unroll
        for (int u = 0; u < R_UNROLL; ++u) {
            const int64_t i = i0 + (int64_t)u * R_THREADS;
            rec[u] = i < r1 ? RecOps<REC>::load(&q[i]) : RecOps<REC>::empty();
        }
#pragma unroll
        for (int u = 0; u < R_UNROLL; ++u) {
            if (i0 + (int64_t)u * R_THREADS < r1) {
                uint32_t e0, e1;
                double a0, a1, b0, b1;
                const bool two = RecOps<REC>::decode(rec[u], e0, a0, a1, e1, b0, b1);
                atomicAdd(&s_acc[2 * e0], a0);
                atomicAdd(&s_acc[2 * e0 + 1], a1);
                if (two) {
                    atomicAdd(&s_acc[2 * e1], b0);
                    atomicAdd(&s_acc[2 * e1 + 1], b1);
                }
            }
        }
    }
    __syncthreads();
    float *dlevel = dtable + (size_t)meta.offset[l] * 2;
    for (int i = threadIdx.x; i < entries * 2; i += R_THREADS) {
        const float val = (float)s_acc[i];
        if (val != 0.0f) {
            const size_t e = (size_t)(i >> 1);
            const size_t idx = interleaved ? e * (size_t)n_bins + (size_t)b : (size_t)b * BIN_ENTRIES + e;
            atomicAdd(dlevel + 2 * idx + (i & 1), val);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Layout conversions for the reference-shaped entry of the stencil path (tcnn.Encoding.forward on [S, 7] interleaved points,
// models/geometry.py:229-244 through models/network_utils.py:47-59): rows [7 S][ld] <-> tap-major planes [L][7][S][2], and
// interleaved points [S][7][3] -> tap-major [7][S][3].  As permuted torch copies these were 30 % of the per-layer route's
// step (83 launches, 0.9 ms each on a 200 x 200 view); here a workgroup moves 32 samples x 7 taps through an LDS tile
// whose rows have an odd stride: 256-byte plane runs on one side, the tile's contiguous rows on the other.
// ------------------------------------------------------------------------------------------------
constexpr int ST_SAMPLES = 32, ST_ROWS = ST_SAMPLES * 7, ST_THREADS = 256;

__global__ void __launch_bounds__(ST_THREADS)
stencil_planes_to_rows_kernel(const float2 *__restrict__ planes, const float *__restrict__ x7, int64_t S, int n_levels,
                              int n_active, float *__restrict__ out, int ld_out, int col_off, int write_xyz, float xyz_scale,
                              float xyz_offset)
{
    extern __shared__ float s_rows[];
    const int nx = write_xyz ? 3 : 0, tw = nx + 2 * n_levels, stride = tw | 1, c0 = col_off - nx;
    const int64_t s0 = (int64_t)blockIdx.x * ST_SAMPLES;
    const int ns = (int)min((int64_t)ST_SAMPLES, S - s0);
    for (int i = threadIdx.x; i < n_levels * 7 * ST_SAMPLES; i += ST_THREADS) {
        const int s = i & (ST_SAMPLES - 1), lt = i >> 5, l = lt / 7, t = lt - 7 * l;
        float2 v = make_float2(0.f, 0.f);
        if (s < ns && l < n_active) v = planes[((int64_t)l * 7 + t) * S + s0 + s];
        float *row = s_rows + (s * 7 + t) * stride + nx + 2 * l;
        row[0] = v.x;
        row[1] = v.y;
    }
    if (write_xyz)
        for (int i = threadIdx.x; i < ns * 21; i += ST_THREADS) {
            const int r = i / 3, d = i - 3 * r;
            s_rows[r * stride + d] = x7[(s0 * 7 + r) * 3 + d] * xyz_scale + xyz_offset;
        }
    __syncthreads();
    const int total = ns * 7 * tw;
    const float inv_tw = 1.0f / (float)tw;     // i / tw exactly: (i + 0.5) / tw is >= 0.5 / tw from an integer, i < 2^13
    for (int i = threadIdx.x; i < total; i += ST_THREADS) {
        const int rr = (int)(((float)i + 0.5f) * inv_tw), c = i - rr * tw;
        out[(s0 * 7 + rr) * ld_out + c0 + c] = s_rows[rr * stride + c];
    }
}

__global__ void __launch_bounds__(ST_THREADS)
stencil_rows_to_planes_kernel(const float *__restrict__ g, int ld, int col_off, int64_t S, int n_levels,
                              float2 *__restrict__ dplanes)
{
    extern __shared__ float s_rows[];
    const int tw = 2 * n_levels, stride = tw | 1;
    const int64_t s0 = (int64_t)blockIdx.x * ST_SAMPLES;
    const int ns = (int)min((int64_t)ST_SAMPLES, S - s0);
    const int total = ns * 7 * tw;
    const float inv_tw = 1.0f / (float)tw;
    for (int i = threadIdx.x; i < total; i += ST_THREADS) {
        const int rr = (int)(((float)i + 0.5f) * inv_tw), c = i - rr * tw;
        s_rows[rr * stride + c] = g[(s0 * 7 + rr) * ld + col_off + c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_levels * 7 * ST_SAMPLES; i += ST_THREADS) {
        const int s = i & (ST_SAMPLES - 1), lt = i >> 5, l = lt / 7, t = lt - 7 * l;
        if (s < ns) {
            const float *row = s_rows + (s * 7 + t) * stride + 2 * l;
            dplanes[((int64_t)l * 7 + t) * S + s0 + s] = make_float2(row[0], row[1]);
        }
    }
}

__global__ void __launch_bounds__(ST_THREADS)
stencil_points_tap_major_kernel(const float *__restrict__ x7, int64_t S, float *__restrict__ x7t)
{
    const int64_t o = (int64_t)blockIdx.x * ST_THREADS + threadIdx.x;     // output element [t][s][d]
    if (o >= 21 * S) return;
    const int64_t ts = o / 3;
    const int d = (int)(o - 3 * ts);
    const int64_t t = ts / S, s = ts - t * S;
    x7t[o] = x7[(s * 7 + t) * 3 + d];
}

// Expected records per (sample, level): 8 for the centre cell + 4 per displaced tap, a tap being
// displaced with probability ~min(1, eps_unit * scale).
double expected_records(float scale, float eps_unit)
{
    double p = (double)eps_unit * (double)scale;
    if (p > 1.0) p = 1.0;
    return 8.0 + 24.0 * p;
}

int make_plan(const rsdf_grid_meta *meta, int64_t S, int n_active, float eps_unit, LevelPlan *plan,
              int64_t *total_records, int *total_counters)
{
    plan->count = n_active;
    int64_t qoff = 0;
    int coff = 0;
    double total_exp = 0.0;
    for (int l = 0; l < n_active; ++l) total_exp += (double)S * expected_records(meta->scale[l], eps_unit);
    // aim at ~3 reducer workgroups per CU, each with the same number of records
#ifndef RSDF_R_WGS
#define RSDF_R_WGS 1536     // (768: 6.56, 1536: 6.11, 2048: 6.33, 3072: 6.11 ms per 18.4 M-sample launch)
#endif
    double per_wg = total_exp / (double)RSDF_R_WGS;
    if (per_wg < 65536.0) per_wg = 65536.0;
    for (int l = 0; l < n_active; ++l) {
        const uint32_t size = meta->size[l];
        const uint64_t dense = (uint64_t)meta->res[l] * meta->res[l] * meta->res[l];
        const bool is_dense = dense <= size;
        if (!is_dense && (size & (size - 1u))) return -3;  // hashed sizes must be powers of two
        int nb = (int)((size + BIN_ENTRIES - 1) >> BIN_SHIFT);
        int lg = 0;
        if (is_dense) {            // round up to a power of two: bin = idx & (nb - 1), entry = idx >> lg (shifts, no division)
            while ((1 << lg) < nb) ++lg;
            nb = 1 << lg;
        }
        if (nb > MAX_BINS) return -2;
        plan->n_bins[l] = nb;
        plan->interleaved[l] = is_dense ? lg + 1 : 0;
        const double per_bin = (double)S * expected_records(meta->scale[l], eps_unit) / nb;
        // RSDF_FD7_QUEUE_SCALE (test knob, default 1): shrinks the queues so that the overflow path -- a record that
        // finds its queue full goes to the table with a direct atomic, and the reducer clamps its count -- is exercised
        // (tests/test_gpu_regimes.py).  Read here so that scratch_bytes() and the launch agree.
        static const double qscale_env = [] {
            if (const char *e = getenv("RSDF_FD7_QUEUE_SCALE")) {
                const double v = atof(e);
                if (v > 0.0 && v <= 1.0) return v;
            }
            return 1.0;
        }();
        const double qscale = qscale_env;
        // queue elements hold two records; a (workgroup, round, bin) run of odd length leaves half an element unused
        // (< 1 % at the fine levels, covered by the slack)
        plan->cap[l] = (qscale < 1.0 ? (int64_t)(per_bin * qscale) + 64 : (int64_t)(per_bin * 1.15) + 16384) / 2 + 1;
        int ns = (int)(per_bin / per_wg + 0.999);
        plan->n_split[l] = ns < 1 ? 1 : (ns > 256 ? 256 : ns);
        plan->queue_off[l] = qoff;
        plan->counter_off[l] = coff;
        qoff += plan->cap[l] * nb;
        coff += nb;
    }
    *total_records = qoff;
    *total_counters = coff;
    return 0;
}

// the record format of this process (rsdf_set_record_format): 0 = block-float pairs (16-byte elements), 1 = fp32 values (20)
#ifdef RSDF_REC_FP32
int g_rec_fp32 = 1;
#else
int g_rec_fp32 = 0;
#endif
int64_t scratch_need(int64_t n_rec, int n_cnt)
{
    const int64_t cbytes = (((int64_t)n_cnt * (int64_t)sizeof(int)) + 255) / 256 * 256;
    return cbytes + n_rec * (int64_t)(g_rec_fp32 ? sizeof(RecF32) : sizeof(RecBF)) + 256;
}

}  // namespace

#ifdef RSDF_STAMPS
extern "C" int rsdf_debug_read_pstamps(unsigned long long *out16, int reset)
{
    const int rc = (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_pstamps), 16 * sizeof(unsigned long long));
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pstamps), z, sizeof(z));
    }
    return rc;
}
#endif

namespace {

// the stencil's source and the contraction's divisor (2 r = r - (-r), as the kernels form it) with its reciprocal for unit_div
TapSrc tap_src(const float *x7t, const float *points, float radius, float eps)
{
    const float two_r = radius - (-radius);
    return TapSrc{x7t, points, radius, eps, UnitDiv{two_r, two_r > 0.f ? 1.0f / two_r : 0.0f}};
}
bool radius_ok(float radius) { return radius - (-radius) > UNIT_DIV_MIN && radius - (-radius) < UNIT_DIV_MAX; }

int launch_fwd(const TapSrc &src, const float *table, const rsdf_grid_meta *meta, int64_t n_samples,
               int n_active_levels, float *planes, void *stream, const X2Out *x3 = nullptr)
{
    RSDF_CHECK_ARG(meta != nullptr, "hashgrid_fwd_fd7: meta is NULL");
    RSDF_CHECK_ARG(meta->n_features == 2, "hashgrid_fwd_fd7: n_features must be 2");
    if (n_samples <= 0) return 0;
    int na = n_active_levels;
    if (na < 0 || na > (int)meta->n_levels) na = (int)meta->n_levels;
    if (na == 0) return 0;
    for (int l = 0; l < na; ++l) {
        const uint64_t dense = (uint64_t)meta->res[l] * meta->res[l] * meta->res[l];
        RSDF_CHECK_ARG(dense <= meta->size[l] || (meta->size[l] & (meta->size[l] - 1u)) == 0,
                       "hashgrid_fwd_fd7: hashed level sizes must be powers of two");
    }
    const unsigned n_tiles_raw = rsdf_blocks(x3 ? x3->Sp : n_samples, F_THREADS);
    const int group = n_tiles_raw < (unsigned)RSDF_FWD_GROUP ? (int)n_tiles_raw : RSDF_FWD_GROUP;
    const unsigned n_tiles_f = (n_tiles_raw + group - 1) / group * group;
    // HIP launches in threads: grid.x * block.x must stay below 2^32
    RSDF_CHECK_ARG((uint64_t)n_tiles_f * na * F_THREADS < (1ull << 32), "hashgrid_fwd_fd7: too many samples for one launch");
    float2 *pl = reinterpret_cast<float2 *>(planes);
    hipStream_t st = (hipStream_t)stream;
    if (x3) {
        if (na < 16) {
            const int64_t n16 = x3->Sp / 32 * 7 * x3->parts * (int64_t)(32 - 2 * na) * 4;
            const unsigned gx = (unsigned)((n16 + 255) / 256 < 4096 ? (n16 + 255) / 256 : 4096);
            x3_zero_cols_kernel<<<gx, 256, 0, st>>>(*x3, 2 * na);
        }
        if (src.x7t)
            fd7_fwd_kernel<false, true><<<n_tiles_f * na, F_THREADS, 0, st>>>(src, table, *meta, n_samples, na, nullptr, *x3, group);
        else
            fd7_fwd_kernel<true, true><<<n_tiles_f * na, F_THREADS, 0, st>>>(src, table, *meta, n_samples, na, nullptr, *x3, group);
        RSDF_RETURN_LAUNCH();
    }
    const X2Out none{nullptr, 0, 0.f, 0.f, 2, 0, 0};
    if (src.x7t)
        fd7_fwd_kernel<false, false><<<n_tiles_f * na, F_THREADS, 0, st>>>(src, table, *meta, n_samples, na, pl, none, group);
    else
        fd7_fwd_kernel<true, false><<<n_tiles_f * na, F_THREADS, 0, st>>>(src, table, *meta, n_samples, na, pl, none, group);
    RSDF_RETURN_LAUNCH();
}

template <bool DERIVE, typename REC>
void launch_produce(dim3 grid, size_t stage_bytes, hipStream_t st, const TapSrc &src, const float2 *dplanes,
                    const rsdf_grid_meta &meta, const LevelPlan &plan, int64_t n_samples, int na, void *queues,
                    int *counters, float *dtable, int dev)
{
    static thread_local unsigned long long attr_set = 0;      // one bit per device (the attribute is per device and instantiation)
    if (!(attr_set >> (dev & 63) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fd7_produce_kernel<DERIVE, REC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)stage_bytes);
        attr_set |= 1ull << (dev & 63);
    }
    fd7_produce_kernel<DERIVE, REC><<<grid, P_THREADS, stage_bytes, st>>>(src, dplanes, meta, plan, n_samples, na,
                                                                          reinterpret_cast<REC *>(queues), counters, dtable);
}

template <typename REC>
int launch_reduce(const rsdf_grid_meta &meta, const LevelPlan &plan, int na, const void *queues, const int *counters, float *dtable,
                  hipStream_t st)
{
    int max_wgs = 0;
    for (int l = 0; l < na; ++l) {
        const int w = plan.n_bins[l] * plan.n_split[l];
        max_wgs = w > max_wgs ? w : max_wgs;
    }
    const size_t lds = (size_t)BIN_ENTRIES * 2 * sizeof(double);
    if (int rc = rsdf_func_lds(reinterpret_cast<const void *>(fd7_reduce_kernel<REC>), lds)) return rc;
    fd7_reduce_kernel<REC><<<dim3(max_wgs, na), R_THREADS, lds, st>>>(meta, plan, reinterpret_cast<const REC *>(queues), counters,
                                                                      dtable);
    return 0;
}

int launch_bwd(const TapSrc &src, const float *dplanes, const rsdf_grid_meta *meta, int64_t n_samples,
               int n_active_levels, float eps_unit, float *dtable, void *scratch, int64_t scratch_bytes, void *stream)
{
    RSDF_CHECK_ARG(meta != nullptr, "hashgrid_bwd_fd7: meta is NULL");
    RSDF_CHECK_ARG(meta->n_features == 2, "hashgrid_bwd_fd7: n_features must be 2");
    if (n_samples <= 0) return 0;
    int na = n_active_levels;
    if (na < 0 || na > (int)meta->n_levels) na = (int)meta->n_levels;
    if (na == 0) return 0;
    LevelPlan plan;
    int n_cnt;
    int64_t n_rec;
    RSDF_CHECK_ARG(make_plan(meta, n_samples, na, eps_unit, &plan, &n_rec, &n_cnt) == 0,
                   "hashgrid_bwd_fd7: unsupported level layout");
    RSDF_CHECK_ARG(scratch != nullptr && scratch_bytes >= scratch_need(n_rec, n_cnt),
                   "hashgrid_bwd_fd7: scratch too small");
    hipStream_t st = (hipStream_t)stream;
    int *counters = (int *)scratch;
    const size_t cbytes = (((size_t)n_cnt * sizeof(int)) + 255) / 256 * 256;
    void *queues = (char *)scratch + cbytes;
    (void)hipMemsetAsync(counters, 0, cbytes, st);
    const unsigned p_tiles = (rsdf_blocks(n_samples, P_THREADS) + RSDF_BWD_GROUP - 1) / RSDF_BWD_GROUP * RSDF_BWD_GROUP;
    // HIP launches in threads: grid.x * block.x must stay below 2^32
    RSDF_CHECK_ARG((uint64_t)p_tiles * na * P_THREADS < (1ull << 32), "hashgrid_bwd_fd7: too many samples for one launch");
    dim3 pgrid(p_tiles * na, 1);
    const size_t stage_bytes = (size_t)STAGE_CAP * sizeof(Record);
    int dev = 0;
    (void)hipGetDevice(&dev);
    const float2 *dpl = reinterpret_cast<const float2 *>(dplanes);
    const bool f32 = g_rec_fp32 != 0;
    if (src.x7t && f32)
        launch_produce<false, RecF32>(pgrid, stage_bytes, st, src, dpl, *meta, plan, n_samples, na, queues, counters, dtable, dev);
    else if (src.x7t)
        launch_produce<false, RecBF>(pgrid, stage_bytes, st, src, dpl, *meta, plan, n_samples, na, queues, counters, dtable, dev);
    else if (f32)
        launch_produce<true, RecF32>(pgrid, stage_bytes, st, src, dpl, *meta, plan, n_samples, na, queues, counters, dtable, dev);
    else
        launch_produce<true, RecBF>(pgrid, stage_bytes, st, src, dpl, *meta, plan, n_samples, na, queues, counters, dtable, dev);
    if (int rc = f32 ? launch_reduce<RecF32>(*meta, plan, na, queues, counters, dtable, st)
                     : launch_reduce<RecBF>(*meta, plan, na, queues, counters, dtable, st))
        return rc;
    RSDF_RETURN_LAUNCH();
}

template <int MODE, typename REC>
void launch_scatter_produce(dim3 grid, size_t stage_bytes, hipStream_t st, const float *x, const float *dy, int ld_dy,
                            int col_off, const float *gdx, const rsdf_grid_meta &meta, const LevelPlan &plan, int64_t n,
                            int na, void *queues, int *counters, float *dtable)
{
    (void)rsdf_func_lds(reinterpret_cast<const void *>(scatter_produce_kernel<MODE, REC>), stage_bytes);
    scatter_produce_kernel<MODE, REC><<<grid, P_THREADS, stage_bytes, st>>>(x, dy, ld_dy, col_off, gdx, meta, plan, n, na,
                                                                           reinterpret_cast<REC *>(queues), counters, dtable);
}

}  // namespace

extern "C" {

int64_t rsdf_hashgrid_scatter_binned_scratch_bytes(const rsdf_grid_meta *meta, int64_t n, int n_active_levels)
{
    return rsdf_hashgrid_bwd_fd7_scratch_bytes(meta, n, n_active_levels, 0.0f);
}

int rsdf_hashgrid_scatter_binned(int mode, const float *x, const float *dy, int ld_dy, int col_off, const float *g_dx,
                                 const rsdf_grid_meta *meta, int64_t n, int n_active_levels, float *dtable, void *scratch,
                                 int64_t scratch_bytes, void *stream)
{
    RSDF_CHECK_ARG(meta != nullptr && meta->n_features == 2, "hashgrid_scatter_binned: n_features must be 2");
    RSDF_CHECK_ARG(mode == 0 || (mode == 1 && g_dx != nullptr), "hashgrid_scatter_binned: mode 0, or 1 with g_dx");
    if (n <= 0) return 0;
    int na = n_active_levels;
    if (na < 0 || na > (int)meta->n_levels) na = (int)meta->n_levels;
    if (na == 0) return 0;
    LevelPlan plan;
    int n_cnt;
    int64_t n_rec;
    RSDF_CHECK_ARG(make_plan(meta, n, na, 0.0f, &plan, &n_rec, &n_cnt) == 0, "hashgrid_scatter_binned: unsupported level layout");
    RSDF_CHECK_ARG(scratch != nullptr && scratch_bytes >= scratch_need(n_rec, n_cnt), "hashgrid_scatter_binned: scratch too small");
    hipStream_t st = (hipStream_t)stream;
    int *counters = (int *)scratch;
    const size_t cbytes = (((size_t)n_cnt * sizeof(int)) + 255) / 256 * 256;
    void *queues = (char *)scratch + cbytes;
    (void)hipMemsetAsync(counters, 0, cbytes, st);
    const unsigned p_tiles = (rsdf_blocks(n, P_THREADS) + RSDF_BWD_GROUP - 1) / RSDF_BWD_GROUP * RSDF_BWD_GROUP;
    RSDF_CHECK_ARG((uint64_t)p_tiles * na * P_THREADS < (1ull << 32), "hashgrid_scatter_binned: too many points for one launch");
    const dim3 pgrid(p_tiles * na, 1);
    const size_t stage_bytes = (size_t)STAGE_CAP * sizeof(Record);
    const bool f32 = g_rec_fp32 != 0;
#define RSDF_SCATTER_LAUNCH(M, R) launch_scatter_produce<M, R>(pgrid, stage_bytes, st, x, dy, ld_dy, col_off, g_dx, *meta, plan, n, na, queues, counters, dtable)
    if (mode == 0 && f32) RSDF_SCATTER_LAUNCH(0, RecF32);
    else if (mode == 0) RSDF_SCATTER_LAUNCH(0, RecBF);
    else if (f32) RSDF_SCATTER_LAUNCH(1, RecF32);
    else RSDF_SCATTER_LAUNCH(1, RecBF);
#undef RSDF_SCATTER_LAUNCH
    if (int rc = f32 ? launch_reduce<RecF32>(*meta, plan, na, queues, counters, dtable, st)
                     : launch_reduce<RecBF>(*meta, plan, na, queues, counters, dtable, st))
        return rc;
    RSDF_RETURN_LAUNCH();
}

/* Run-time choice of the hash backward's queue record format for this process (round 6): 0 = 16-byte elements of two
 * block-float contributions (20 significant bits; the default), 1 = 20-byte elements of fp32 values.  Not thread safe against
 * concurrent launches: set it once (rise_sdf_amd reads RSDF_REC=fp32 at import); scratch sizes follow the current setting. */
int rsdf_set_record_format(int fp32_values)
{
    g_rec_fp32 = fp32_values ? 1 : 0;
    return 0;
}
int rsdf_get_record_format(void) { return g_rec_fp32; }

int rsdf_hashgrid_fwd_fd7(const float *x7t, const float *table, const rsdf_grid_meta *meta,
                          int64_t n_samples, int n_active_levels, float *planes, void *stream)
{
    RSDF_CHECK_ARG(x7t != nullptr || n_samples <= 0, "hashgrid_fwd_fd7: x7t is NULL");
    return launch_fwd(tap_src(x7t, nullptr, 0.f, 0.f), table, meta, n_samples, n_active_levels, planes, stream);
}

int rsdf_hashgrid_fwd_fd7_pts(const float *points, float radius, float eps, const float *table,
                              const rsdf_grid_meta *meta, int64_t n_samples, int n_active_levels, float *planes,
                              void *stream)
{
    RSDF_CHECK_ARG(points != nullptr || n_samples <= 0, "hashgrid_fwd_fd7_pts: points is NULL");
    RSDF_CHECK_ARG(radius_ok(radius), "hashgrid_fwd_fd7_pts: radius must be in (2^-101, 2^99)");
    return launch_fwd(tap_src(nullptr, points, radius, eps), table, meta, n_samples, n_active_levels, planes, stream);
}

int rsdf_stencil_points_tap_major(const float *x7, int64_t n_samples, float *x7t, void *stream)
{
    if (n_samples <= 0) return 0;
    RSDF_CHECK_ARG(x7 != nullptr && x7t != nullptr, "stencil_points_tap_major: NULL pointer");
    RSDF_CHECK_ARG((uint64_t)n_samples * 21 < (1ull << 40), "stencil_points_tap_major: too many samples");
    stencil_points_tap_major_kernel<<<rsdf_blocks(21 * n_samples, ST_THREADS), ST_THREADS, 0, (hipStream_t)stream>>>(x7, n_samples,
                                                                                                                       x7t);
    RSDF_RETURN_LAUNCH();
}

int rsdf_stencil_planes_to_rows(const float *planes, const float *x7, int64_t n_samples, int n_levels, int n_active_levels,
                                float *out, int ld_out, int col_off, int write_xyz, float xyz_scale, float xyz_offset,
                                void *stream)
{
    if (n_samples <= 0) return 0;
    RSDF_CHECK_ARG(n_levels >= 1 && n_levels <= RSDF_MAX_LEVELS, "stencil_planes_to_rows: bad level count");
    RSDF_CHECK_ARG(planes != nullptr && out != nullptr && (!write_xyz || x7 != nullptr), "stencil_planes_to_rows: NULL pointer");
    RSDF_CHECK_ARG(col_off >= (write_xyz ? 3 : 0) && ld_out >= col_off + 2 * n_levels, "stencil_planes_to_rows: bad row layout");
    if (n_active_levels < 0 || n_active_levels > n_levels) n_active_levels = n_levels;
    const int tw = (write_xyz ? 3 : 0) + 2 * n_levels;
    const size_t lds = (size_t)ST_ROWS * (tw | 1) * sizeof(float);
    stencil_planes_to_rows_kernel<<<rsdf_blocks(n_samples, ST_SAMPLES), ST_THREADS, lds, (hipStream_t)stream>>>(
        reinterpret_cast<const float2 *>(planes), x7, n_samples, n_levels, n_active_levels, out, ld_out, col_off, write_xyz,
        xyz_scale, xyz_offset);
    RSDF_RETURN_LAUNCH();
}

int rsdf_stencil_rows_to_planes(const float *g, int ld, int col_off, int64_t n_samples, int n_levels, float *dplanes,
                                void *stream)
{
    if (n_samples <= 0) return 0;
    RSDF_CHECK_ARG(n_levels >= 1 && n_levels <= RSDF_MAX_LEVELS, "stencil_rows_to_planes: bad level count");
    RSDF_CHECK_ARG(g != nullptr && dplanes != nullptr && col_off >= 0 && ld >= col_off + 2 * n_levels,
                   "stencil_rows_to_planes: bad arguments");
    const size_t lds = (size_t)ST_ROWS * ((2 * n_levels) | 1) * sizeof(float);
    stencil_rows_to_planes_kernel<<<rsdf_blocks(n_samples, ST_SAMPLES), ST_THREADS, lds, (hipStream_t)stream>>>(
        g, ld, col_off, n_samples, n_levels, reinterpret_cast<float2 *>(dplanes));
    RSDF_RETURN_LAUNCH();
}

int64_t rsdf_x2_rows(int64_t n_samples) { return (n_samples + 31) / 32 * 32; }
int64_t rsdf_x2_bytes(int64_t n_samples, int parts)        // + slack: see mlp_x2.hip fetch_x2
{
    return rsdf_x2_rows(n_samples) / 32 * (int64_t)(7 * (parts == 1 ? 1 : 2) * X2_PART_B) + 4096;
}

int rsdf_hashgrid_fwd_fd7_x2(const float *x7t, const float *points, float radius, float eps, const float *table,
                             const rsdf_grid_meta *meta, int64_t n_samples, int n_active_levels, float xyz_scale,
                             float xyz_offset, int parts, void *x3, void *stream)
{
    RSDF_CHECK_ARG((x7t != nullptr) != (points != nullptr) || n_samples <= 0, "hashgrid_fwd_fd7_x2: give x7t or points");
    RSDF_CHECK_ARG(points == nullptr || radius_ok(radius), "hashgrid_fwd_fd7_x2: radius must be in (2^-101, 2^99)");
    RSDF_CHECK_ARG(x3 != nullptr || n_samples <= 0, "hashgrid_fwd_fd7_x2: x3 is NULL");
    RSDF_CHECK_ARG(parts == 1 || parts == 2, "hashgrid_fwd_fd7_x2: parts must be 1 or 2");
    const X2Out o{reinterpret_cast<unsigned char *>(x3), rsdf_x2_rows(n_samples), xyz_scale, xyz_offset, parts, parts * X2_PART_B,
                  7 * parts * X2_PART_B};
    return launch_fwd(x7t ? tap_src(x7t, nullptr, 0.f, 0.f) : tap_src(nullptr, points, radius, eps), table, meta, n_samples,
                      n_active_levels, nullptr, stream, &o);
}

int64_t rsdf_hashgrid_bwd_fd7_scratch_bytes(const rsdf_grid_meta *meta, int64_t n_samples,
                                            int n_active_levels, float eps_unit)
{
    if (!meta) return -1;
    LevelPlan plan;
    int n_cnt;
    int64_t n_rec;
    int na = n_active_levels;
    if (na < 0 || na > (int)meta->n_levels) na = (int)meta->n_levels;
    if (make_plan(meta, n_samples, na, eps_unit, &plan, &n_rec, &n_cnt) != 0) return -1;
    return scratch_need(n_rec, n_cnt);
}

int rsdf_hashgrid_bwd_fd7(const float *x7t, const float *dplanes, const rsdf_grid_meta *meta,
                          int64_t n_samples, int n_active_levels, float eps_unit, float *dtable,
                          void *scratch, int64_t scratch_bytes, void *stream)
{
    RSDF_CHECK_ARG(x7t != nullptr || n_samples <= 0, "hashgrid_bwd_fd7: x7t is NULL");
    return launch_bwd(tap_src(x7t, nullptr, 0.f, 0.f), dplanes, meta, n_samples, n_active_levels, eps_unit, dtable,
                      scratch, scratch_bytes, stream);
}

int rsdf_hashgrid_bwd_fd7_pts(const float *points, float radius, float eps, const float *dplanes,
                              const rsdf_grid_meta *meta, int64_t n_samples, int n_active_levels, float eps_unit,
                              float *dtable, void *scratch, int64_t scratch_bytes, void *stream)
{
    RSDF_CHECK_ARG(points != nullptr || n_samples <= 0, "hashgrid_bwd_fd7_pts: points is NULL");
    RSDF_CHECK_ARG(radius_ok(radius), "hashgrid_bwd_fd7_pts: radius must be in (2^-101, 2^99)");
    return launch_bwd(tap_src(nullptr, points, radius, eps), dplanes, meta, n_samples, n_active_levels, eps_unit, dtable,
                      scratch, scratch_bytes, stream);
}

}  // extern "C"
