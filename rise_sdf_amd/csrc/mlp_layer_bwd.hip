// T3 / H3: backward of one 128-wide VanillaMLP layer in ONE pass over its rows (models/network_utils.py:109-157;
// the texture networks of models/texture.py:237-327 are stacks of such layers).
//
//   dz = dy * act'(y)                       (never leaves the CU)
//   dx[:, 0:Kout] = dz @ W[:, k0:k0+Kout]   (nullable)
//   dW += dz^T x,   db += colsum(dz)
//
// The two-kernel form (mlp.hip: linear_bwd_input writes dz, linear_bwd_weight reads dz and x again) moves
// 16 N + 4 K + 4 (K + N) bytes per row and reads its operands as scalar dwords in fragment order, every value split
// to bf16 parts once per consuming wave.  Here a workgroup of 8 waves takes 64-row tiles:
//   * all 512 threads stage the tile with 16-byte loads: dz and x are split ONCE into two LDS images
//     [part h/m/l][8-column chunk][64 rows][8 columns] (chunk stride 1088 B: the layout of mlp_coop.hip with 64 rows);
//   * dx: wave (jt = w & 3, row half = w >> 2) owns a 32-column x 32-row output tile; its W^T fragments (8 k-steps x 3
//     parts = 96 registers) stay in registers for the whole kernel, dz fragments come from the image (ds_read_b128);
//   * dW: wave (nt = w & 3, k pair = w >> 2) owns two 32 x 32 tiles of dW in registers for the whole row loop; both
//     operands are transposed fragments of the images (ds_read_b64_tr_b16: rows are the MFMA k dimension);
//   * db: the staging threads keep running column sums (a thread always stages the same 4 columns);
//   * the next tile's rows are fetched into registers while the current tile computes; the two workgroup barriers per
//     tile are s_waitcnt lgkmcnt(0) + s_barrier (a __syncthreads() would also wait for that prefetch).
// Traffic: 8 N + 4 K (+ 4 Kout) bytes per row.  All products are the 6-term split-bf16 products of split_bf16.h.
#include "common.h"
#include "split_bf16.h"
#include "act.h"

namespace {

typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;

constexpr int LB_WAVES = 8;
constexpr int LB_THREADS = LB_WAVES * 64;
constexpr int LB_ROWS = 64;
constexpr int LB_N = 128;                      // layer width this kernel is built for
constexpr int CS64 = LB_ROWS * 16 + 64;        // chunk stride: (CS64 / 4) % 64 == 16, as the 32-row layout (tools/lds_bank_sim.py)
constexpr int IMG_PART = (LB_N / 8) * CS64;    // one bf16 part of a [64][128] image
constexpr int IMG_BYTES = 3 * IMG_PART;
constexpr int PASSES = LB_ROWS * LB_N / 4 / LB_THREADS;   // float4 per thread and operand: 4

__device__ __forceinline__ void lds_barrier()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);      // s_waitcnt lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ u32x4 lds_b128(const unsigned char *p) { return *reinterpret_cast<const u32x4 *>(p); }

// B fragment with k = columns: lane (row, half hf) reads columns 16 ns + 8 hf .. +7
__device__ __forceinline__ Frag3 row_frag64(const unsigned char *img, int ns, int row, int hf)
{
    const unsigned char *p = img + (2 * ns + hf) * CS64 + row * 16;
    Frag3 f;
    f.h = lds_b128(p);
    f.m = lds_b128(p + IMG_PART);
    f.l = lds_b128(p + 2 * IMG_PART);
    return f;
}

__device__ __forceinline__ void tr2(const unsigned char *p, unsigned &a, unsigned &b)
{
    const v4i16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16 *)p);
    const unsigned long long u = __builtin_bit_cast(unsigned long long, r);
    a = (unsigned)u;
    b = (unsigned)(u >> 32);
}
// Fragment with k = ROWS: lane (column 32 tile + (lane & 31), h = lane >> 5), k-step ks covers rows 16 ks + 8 h .. +7
// (the lane map of mlp_coop.hip::tr_frag, checked on the box by tools/tr_read_check.hip).
__device__ __forceinline__ Frag3 tr_frag64(const unsigned char *img, int tile, int ks, int lane)
{
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int row = 16 * ks + 8 * (g >> 1) + q;
    const unsigned char *a = img + (4 * tile + 2 * (g & 1) + (p >> 1)) * CS64 + row * 16 + (p & 1) * 8;
    Frag3 f;
    unsigned x0, x1, y0, y1;
    tr2(a, x0, x1);
    tr2(a + 64, y0, y1);                                  // rows + 4
    f.h = u32x4{x0, x1, y0, y1};
    tr2(a + IMG_PART, x0, x1);
    tr2(a + IMG_PART + 64, y0, y1);
    f.m = u32x4{x0, x1, y0, y1};
    tr2(a + 2 * IMG_PART, x0, x1);
    tr2(a + 2 * IMG_PART + 64, y0, y1);
    f.l = u32x4{x0, x1, y0, y1};
    return f;
}

// four consecutive columns col .. col+3 (col % 4 == 0) of one row -> the three parts of an image
__device__ __forceinline__ void put4(unsigned char *img, int row, int col, float v0, float v1, float v2, float v3)
{
    unsigned h0, m0, l0, h1, m1, l1;
    split3_pair(v0, v1, h0, m0, l0);
    split3_pair(v2, v3, h1, m1, l1);
    unsigned char *p = img + (col >> 3) * CS64 + row * 16 + (col & 7) * 2;
    *reinterpret_cast<uint2 *>(p) = uint2{h0, h1};
    *reinterpret_cast<uint2 *>(p + IMG_PART) = uint2{m0, m1};
    *reinterpret_cast<uint2 *>(p + 2 * IMG_PART) = uint2{l0, l1};
}

// RAW load of 4 consecutive floats on an always-valid address (masking happens at use: a load inside a divergent
// branch is waited for on the spot, which serialises the prefetch)
__device__ __forceinline__ float4 load4(const float *__restrict__ p, bool vec)
{
    if (vec) return *reinterpret_cast<const float4 *>(p);
    return make_float4(p[0], p[1], p[2], p[3]);
}

// ACT is a template parameter: with a run-time activation id the derivative's switch is re-evaluated for each of the
// 16 staged values (the first build spent 620 scalar and 980 vector instructions per tile on it)
// PREV_RELU: the layer's input x is the ReLU output of the previous layer of the chain; dx is then written as that layer's
// dz = dx * (x > 0), the mask read from the high bf16 part of the X image (same sign as x, and zero exactly when x is zero
// or a positive value below half the smallest bf16 subnormal, 4.6e-41 -- not a value an activation takes): the previous
// layer's backward needs no activation pass and does not read its y at all.
// TAIL: the layer above is the network's narrow output layer (N2 <= 4 columns, weights w2 [N2][128]); instead of reading its
// input gradient dy [n][128] this kernel forms it while staging, dy[row][col] = sum_q dzo[row][q] w2[q][col], from that
// layer's dz (dzo [n][N2], 4-16 bytes per row): the output layer's input-gradient kernel and 512 bytes per row go away.
struct TailArgs {
    const float *dzo, *w2;
    int N2;
};
template <bool HAS_DX, int ACT, bool PREV_RELU, bool TAIL>
__global__ void __launch_bounds__(LB_THREADS, 2)
layer_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y, int lddy, const float *__restrict__ x,
                 int ldx, const float *__restrict__ w, int64_t n, int K, int k0, int Kout, const TailArgs tail,
                 float *__restrict__ dx, int lddx, float *__restrict__ dw, float *__restrict__ db, float *__restrict__ part)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char *dzi = smem_b, *xi = smem_b + IMG_BYTES;
    float *w2s = reinterpret_cast<float *>(smem_b + 2 * IMG_BYTES);   // TAIL: [4][128]
    if (TAIL) {
        for (int e = threadIdx.x; e < 4 * LB_N; e += LB_THREADS) w2s[e] = e < tail.N2 * LB_N ? tail.w2[e] : 0.0f;
        __syncthreads();   // once, before any prefetch is in flight
    }
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, c = lane & 31, hf = lane >> 5;
    const int KT = (K + 31) >> 5;

    // ---- per-wave constant state
    const int jt = wave & 3, rh = wave >> 2;                 // dx: column tile, row half
    const bool dx_on = HAS_DX && 32 * jt < Kout;
    Frag3 wt[HAS_DX ? 8 : 1];
    if (HAS_DX) {
        const int kc = 32 * jt + c;
#pragma unroll
        for (int ns = 0; ns < 8; ++ns) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = kc < Kout ? w[(16 * ns + 8 * hf + j) * K + k0 + kc] : 0.0f;
            wt[ns] = split_frag(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
        }
    }
    const int nt = wave & 3, kt0 = 2 * (wave >> 2), kt1 = kt0 + 1;   // dW: n tile, the two k tiles
    const bool w0_on = kt0 < KT, w1_on = kt1 < KT;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = 0.0f, acc1[r] = 0.0f;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};

    // ---- staging map: float4 (row (t >> 5) + 16 i, columns col4 .. col4 + 3), i < PASSES
    const int srow = t >> 5, col4 = (t & 31) * 4;
    const bool vz = (lddy & 3) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 &&
                    (y == nullptr || (reinterpret_cast<uintptr_t>(y) & 15) == 0);
    const bool vxl = (ldx & 3) == 0 && (K & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const bool vxs = HAS_DX && (lddx & 3) == 0 && (reinterpret_cast<uintptr_t>(dx) & 15) == 0;
    // x columns: a clamped (valid) column for the raw load, masked at use
    const int xcol = col4 + 4 <= K ? col4 : (K >= 4 ? K - 4 : 0);
    const int64_t last = n - 1;
    float4 pdy[PASSES], py[PASSES], px[PASSES];
    auto fetch = [&](int64_t tile) {
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int64_t row = tile * LB_ROWS + srow + 16 * i;
            const int64_t rc = row <= last ? row : last;
            if (TAIL) {
                const float *p = tail.dzo + rc * tail.N2;
                pdy[i] = make_float4(p[0], tail.N2 > 1 ? p[1] : 0.f, tail.N2 > 2 ? p[2] : 0.f, tail.N2 > 3 ? p[3] : 0.f);
            } else {
                pdy[i] = load4(dy + rc * lddy + col4, vz);
            }
            if (ACT != RSDF_ACT_NONE) py[i] = load4(y + rc * lddy + col4, vz);
            if (K >= 4) px[i] = load4(x + rc * ldx + xcol, vxl);
            else px[i] = make_float4(x[rc * ldx], K > 1 ? x[rc * ldx + 1] : 0.f, K > 2 ? x[rc * ldx + 2] : 0.f, 0.f);
        }
    };

    const int64_t n_tiles = (n + LB_ROWS - 1) / LB_ROWS;
    int64_t tile = blockIdx.x;
    if (tile < n_tiles) fetch(tile);
    for (; tile < n_tiles; tile += gridDim.x) {
        const int64_t row0 = tile * LB_ROWS;
        // ---- stage: dz and x, split once
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int row = srow + 16 * i;
            const bool ok = row0 + row < n;
            float4 g = pdy[i];
            if (TAIL) {
                const float4 d = pdy[i];
                const float4 a0 = *reinterpret_cast<const float4 *>(w2s + col4), a1 = *reinterpret_cast<const float4 *>(w2s + LB_N + col4);
                const float4 a2 = *reinterpret_cast<const float4 *>(w2s + 2 * LB_N + col4),
                             a3 = *reinterpret_cast<const float4 *>(w2s + 3 * LB_N + col4);
                g.x = fmaf(d.w, a3.x, fmaf(d.z, a2.x, fmaf(d.y, a1.x, d.x * a0.x)));
                g.y = fmaf(d.w, a3.y, fmaf(d.z, a2.y, fmaf(d.y, a1.y, d.x * a0.y)));
                g.z = fmaf(d.w, a3.z, fmaf(d.z, a2.z, fmaf(d.y, a1.z, d.x * a0.z)));
                g.w = fmaf(d.w, a3.w, fmaf(d.z, a2.w, fmaf(d.y, a1.w, d.x * a0.w)));
            }
            if (ACT != RSDF_ACT_NONE) {
                g.x *= act_bwd_from_y(py[i].x, ACT);
                g.y *= act_bwd_from_y(py[i].y, ACT);
                g.z *= act_bwd_from_y(py[i].z, ACT);
                g.w *= act_bwd_from_y(py[i].w, ACT);
            }
            if (!ok) g = make_float4(0.f, 0.f, 0.f, 0.f);
            bs[0] += g.x, bs[1] += g.y, bs[2] += g.z, bs[3] += g.w;
            put4(dzi, row, col4, g.x, g.y, g.z, g.w);
            float4 v = px[i];
            if (K >= 4 && xcol != col4) {   // the clamped load covers columns xcol .. xcol+3: pick the ones of this slot
                const float s[4] = {v.x, v.y, v.z, v.w};
                float o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int src = col4 + q - xcol;
                    o[q] = (col4 + q < K && src >= 0 && src < 4) ? s[src & 3] : 0.0f;
                }
                v = make_float4(o[0], o[1], o[2], o[3]);
            } else if (K < 4) {
                if (col4 != 0) v = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            put4(xi, row, col4, v.x, v.y, v.z, v.w);
        }
        lds_barrier();   // images complete
        if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);

        // ---- dx tile [32 columns of the window][32 rows]
        if (dx_on) {
            f32x16 a;
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = 0.0f;
#pragma unroll
            for (int ns = 0; ns < 8; ++ns) a = mma6r(wt[HAS_DX ? ns : 0], row_frag64(dzi, ns, 32 * rh + c, hf), a);
            const int64_t row = row0 + 32 * rh + c;
            if (PREV_RELU) {   // (k0 == 0: window column = input column)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = 32 * jt + 8 * g + 4 * hf;
                    const uint2 hx = *reinterpret_cast<const uint2 *>(xi + (col >> 3) * CS64 + (32 * rh + c) * 16 + (col & 7) * 2);
                    if ((short)(hx.x & 0xffffu) <= 0) a[4 * g] = 0.0f;
                    if ((short)(hx.x >> 16) <= 0) a[4 * g + 1] = 0.0f;
                    if ((short)(hx.y & 0xffffu) <= 0) a[4 * g + 2] = 0.0f;
                    if ((short)(hx.y >> 16) <= 0) a[4 * g + 3] = 0.0f;
                }
            }
            if (row < n) {
                float *xr = dx + row * (int64_t)lddx;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = 32 * jt + 8 * g + 4 * hf;   // registers 4g .. 4g+3 = window columns col .. col+3
                    if (vxs && col + 4 <= Kout) {
                        *reinterpret_cast<float4 *>(xr + col) = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (col + q < Kout) xr[col + q] = a[4 * g + q];
                    }
                }
            }
        }
        // ---- dW tiles: k = the tile's 64 rows
        if (w0_on) {
#pragma unroll
            for (int ks = 0; ks < LB_ROWS / 16; ++ks) {
                const Frag3 az = tr_frag64(dzi, nt, ks, lane);
                acc0 = mma6r(az, tr_frag64(xi, kt0, ks, lane), acc0);
                if (w1_on) acc1 = mma6r(az, tr_frag64(xi, kt1, ks, lane), acc1);
            }
        }
        lds_barrier();   // images free
    }

    // ---- flush: D[i = n within tile][j = k within tile]
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int kcol = 32 * (tt ? kt1 : kt0) + c;
        if ((tt ? w1_on : w0_on) && kcol < K) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = tt ? acc1[r] : acc0[r];
                const int nrow = 32 * nt + (r & 3) + 8 * (r >> 2) + 4 * hf;
                if (part != nullptr) part[(size_t)blockIdx.x * (LB_N * K + LB_N) + nrow * K + kcol] = v;
                else if (v != 0.0f) atomicAdd(&dw[nrow * K + kcol], v);
            }
        }
    }
    if (db) {
        // Bias sums: sixteen staging threads hold partial sums of the same four columns.  They meet in LDS (the images are
        // free) and the workgroup issues 128 adds instead of 1024: per launch every db address used to receive 8 x 256
        // same-address float atomics, which the memory side executes one after the other -- 0.19 ms whatever the row
        // count (tools/layer_bwd_fixed_cost.py), more than the rest of the call at a training step's 250 k rows.
        float *red = reinterpret_cast<float *>(smem_b);           // [16][128]
#pragma unroll
        for (int q = 0; q < 4; ++q) red[srow * LB_N + col4 + q] = bs[q];
        __syncthreads();
        if (t < LB_N) {
            float s = 0.0f;
#pragma unroll
            for (int r = 0; r < LB_THREADS / 32; ++r) s += red[r * LB_N + t];
            if (part != nullptr) part[(size_t)blockIdx.x * (LB_N * K + LB_N) + LB_N * K + t] = s;
            else if (s != 0.0f) atomicAdd(&db[t], s);
        }
    } else if (part != nullptr && t < LB_N) {
        part[(size_t)blockIdx.x * (LB_N * K + LB_N) + LB_N * K + t] = 0.0f;
    }
}

// The flush above is N K float atomics per workgroup onto the SAME N K addresses from up to 256 workgroups (4.2 M
// memory-side atomic operations per launch), and as many same-address adds per db entry as there are workgroups.  With a
// workspace the workgroups store their dW tiles and bias sums as plain [workgroup][N K + N] partials and this kernel adds
// their sums to dw / db (which keep their "accumulates: zero first" contract).
__global__ void __launch_bounds__(256)
dw_reduce_kernel(const float *__restrict__ part, int n_parts, int NK, int N, float *__restrict__ dw, float *__restrict__ db)
{
    const int i = blockIdx.x * 256 + threadIdx.x, stride = NK + N;     // a partial: [N K] of dW, then [N] of db
    if (i >= stride) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int g = 0;
    for (; g + 3 < n_parts; g += 4) {
        s0 += part[(size_t)g * stride + i];
        s1 += part[(size_t)(g + 1) * stride + i];
        s2 += part[(size_t)(g + 2) * stride + i];
        s3 += part[(size_t)(g + 3) * stride + i];
    }
    for (; g < n_parts; ++g) s0 += part[(size_t)g * stride + i];
    const float s = (s0 + s1) + (s2 + s3);
    if (i < NK) dw[i] += s;
    else if (db != nullptr) db[i - NK] += s;
}

struct LaunchArgs {
    unsigned grid;
    size_t lds;
    hipStream_t st;
    const float *dy, *y;
    int lddy;
    const float *x;
    int ldx;
    const float *w;
    int64_t n;
    int K, k0, Kout;
    TailArgs tail;
    float *dx;
    int lddx;
    float *dw, *db;
    float *part;     // workspace for the per-workgroup dW partials (nullptr: atomic flush)
};

template <bool HAS_DX, int ACT, bool PREV_RELU, bool TAIL>
int launch(const LaunchArgs &a)
{
    if (int rc = rsdf_func_lds(reinterpret_cast<const void *>(layer_bwd_kernel<HAS_DX, ACT, PREV_RELU, TAIL>), a.lds))
        return rc;   // per (kernel, device, host thread): autograd calls this from its own thread
    layer_bwd_kernel<HAS_DX, ACT, PREV_RELU, TAIL><<<a.grid, LB_THREADS, a.lds, a.st>>>(
        a.dy, a.y, a.lddy, a.x, a.ldx, a.w, a.n, a.K, a.k0, a.Kout, a.tail, a.dx, a.lddx, a.dw, a.db, a.part);
    if (a.part != nullptr)
        dw_reduce_kernel<<<(LB_N * a.K + LB_N + 255) / 256, 256, 0, a.st>>>(a.part, (int)a.grid, LB_N * a.K, LB_N, a.dw, a.db);
    return 0;
}

template <bool HAS_DX, bool PREV_RELU>
int launch_act(int act, const LaunchArgs &a)
{
    switch (act) {
    case RSDF_ACT_NONE: return launch<HAS_DX, RSDF_ACT_NONE, PREV_RELU, false>(a);
    case RSDF_ACT_RELU: return launch<HAS_DX, RSDF_ACT_RELU, PREV_RELU, false>(a);
    case RSDF_ACT_SOFTPLUS100: return launch<HAS_DX, RSDF_ACT_SOFTPLUS100, PREV_RELU, false>(a);
    case RSDF_ACT_SIGMOID: return launch<HAS_DX, RSDF_ACT_SIGMOID, PREV_RELU, false>(a);
    default: rsdf_set_error("linear_bwd_fused: unknown activation"); return 1;
    }
}

int check_common(const float *y, int lddy, const float *x, int ldx, int K, int N, int act, int k0, int Kout, const float *dx,
                 int lddx, int prev_act, const float *dw)
{
    RSDF_CHECK_ARG(N == LB_N && K >= 1 && K <= 128, "linear_bwd_fused: needs N == 128 and K in [1,128]");
    RSDF_CHECK_ARG(lddy >= N && ldx >= K, "linear_bwd_fused: row stride smaller than the row");
    RSDF_CHECK_ARG(prev_act == RSDF_ACT_NONE || (prev_act == RSDF_ACT_RELU && dx != nullptr && k0 == 0),
                   "linear_bwd_fused: prev_act must be NONE, or RELU with dx and k0 == 0");
    RSDF_CHECK_ARG(dw != nullptr, "linear_bwd_fused: dw is NULL");
    if (dx) {
        RSDF_CHECK_ARG(k0 >= 0 && Kout >= 1 && k0 + Kout <= K, "linear_bwd_fused: bad column window");
        RSDF_CHECK_ARG(lddx >= Kout, "linear_bwd_fused: lddx < Kout");
    }
    return 0;
}

}  // namespace

extern "C" {

int RSDF_P(rsdf_linear_bwd_fused_supported)(int K, int N) { return N == LB_N && K >= 1 && K <= 128; }

// workspace: >= rsdf_linear_bwd_fused_workspace_bytes(n, K, N) bytes, or NULL (atomic flush); used only from 64 tiles up
static float *pick_part(void *workspace, int64_t workspace_bytes, int64_t tiles, unsigned grid, int K)
{
    if (workspace == nullptr || tiles < 64) return nullptr;
    return workspace_bytes >= (int64_t)grid * (LB_N * K + LB_N) * (int64_t)sizeof(float) ? static_cast<float *>(workspace) : nullptr;
}

#ifndef RSDF_BF16
int64_t rsdf_linear_bwd_fused_workspace_bytes(int64_t n, int K, int N)
{
    if (N != LB_N || K < 1 || K > 128 || n <= 0) return 0;
    const int64_t tiles = (n + LB_ROWS - 1) / LB_ROWS;
    return (tiles < 256 ? tiles : 256) * (int64_t)(LB_N * K + LB_N) * (int64_t)sizeof(float);
}
#endif

int RSDF_P(rsdf_linear_bwd_fused_ws)(const float *dy, const float *y, int lddy, const float *x, int ldx, const float *w,
                             int64_t n, int K, int N, int act, int k0, int Kout, float *dx, int lddx, int prev_act,
                             float *dw, float *db, void *workspace, int64_t workspace_bytes, void *stream)
{
    if (int rc = check_common(y, lddy, x, ldx, K, N, act, k0, Kout, dx, lddx, prev_act, dw)) return rc;
    if (n <= 0) return 0;   // an empty batch carries no pointers
    RSDF_CHECK_ARG(act == RSDF_ACT_NONE || y != nullptr, "linear_bwd_fused: activation needs y");
    const int64_t tiles = (n + LB_ROWS - 1) / LB_ROWS;
    const unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
    LaunchArgs a{grid, 2 * (size_t)IMG_BYTES, (hipStream_t)stream, dy, y, lddy, x, ldx, w, n, K,
                 dx ? k0 : 0, dx ? Kout : 0, TailArgs{nullptr, nullptr, 0}, dx, dx ? lddx : 0, dw, db,
                 pick_part(workspace, workspace_bytes, tiles, grid, K)};
    const int rc = !dx ? launch_act<false, false>(act, a)
                   : prev_act == RSDF_ACT_RELU ? launch_act<true, true>(act, a) : launch_act<true, false>(act, a);
    if (rc) return rc;
    RSDF_RETURN_LAUNCH();
}

int RSDF_P(rsdf_linear_bwd_fused)(const float *dy, const float *y, int lddy, const float *x, int ldx, const float *w,
                          int64_t n, int K, int N, int act, int k0, int Kout, float *dx, int lddx, int prev_act,
                          float *dw, float *db, void *stream)
{
    return RSDF_P(rsdf_linear_bwd_fused_ws)(dy, y, lddy, x, ldx, w, n, K, N, act, k0, Kout, dx, lddx, prev_act, dw, db,
                                            nullptr, 0, stream);
}

int RSDF_P(rsdf_linear_bwd_fused_tail_ws)(const float *dz_out, int N2, const float *w2, const float *y, int lddy,
                                  const float *x, int ldx, const float *w, int64_t n, int K, int N, int act, int k0,
                                  int Kout, float *dx, int lddx, int prev_act, float *dw, float *db, void *workspace,
                                  int64_t workspace_bytes, void *stream)
{
    if (int rc = check_common(y, lddy, x, ldx, K, N, act, k0, Kout, dx, lddx, prev_act, dw)) return rc;
    RSDF_CHECK_ARG(N2 >= 1 && N2 <= 4, "linear_bwd_fused_tail: the output layer must have 1..4 columns");
    RSDF_CHECK_ARG(act == RSDF_ACT_RELU && dx != nullptr, "linear_bwd_fused_tail: built for ReLU layers with an input gradient");
    if (n <= 0) return 0;
    RSDF_CHECK_ARG(dz_out != nullptr && w2 != nullptr && y != nullptr, "linear_bwd_fused_tail: NULL argument");
    const int64_t tiles = (n + LB_ROWS - 1) / LB_ROWS;
    const unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
    LaunchArgs a{grid, 2 * (size_t)IMG_BYTES + 4 * LB_N * sizeof(float), (hipStream_t)stream,
                 nullptr, y, lddy, x, ldx, w, n, K, k0, Kout, TailArgs{dz_out, w2, N2}, dx, lddx, dw, db,
                 pick_part(workspace, workspace_bytes, tiles, grid, K)};
    const int rc = prev_act == RSDF_ACT_RELU ? launch<true, RSDF_ACT_RELU, true, true>(a) : launch<true, RSDF_ACT_RELU, false, true>(a);
    if (rc) return rc;
    RSDF_RETURN_LAUNCH();
}

int RSDF_P(rsdf_linear_bwd_fused_tail)(const float *dz_out, int N2, const float *w2, const float *y, int lddy, const float *x,
                               int ldx, const float *w, int64_t n, int K, int N, int act, int k0, int Kout, float *dx,
                               int lddx, int prev_act, float *dw, float *db, void *stream)
{
    return RSDF_P(rsdf_linear_bwd_fused_tail_ws)(dz_out, N2, w2, y, lddy, x, ldx, w, n, K, N, act, k0, Kout, dx, lddx, prev_act,
                                                 dw, db, nullptr, 0, stream);
}

}  // extern "C"
