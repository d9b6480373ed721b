// H3 fused, cooperative form: the SDF VanillaMLP (Linear -> Softplus(100) -> Linear -> Softplus(100) -> Linear,
// models/network_utils.py:109-157, n_hidden_layers = 2) for hidden widths 32 / 64 / 128 on the finite-difference
// stencil layout (x7t [7][S][3], planes [L][7][S][2]).
//
// Why a second form next to mlp_fused.hip.  There every wave owns whole 32-row tiles and every wave needs every
// weight; at H = 128 (configs/split-mixed-occ-tensoir.yaml:73-84, the width the reference actually ships) the
// 3-way split bf16 weights are 135 KB and neither LDS nor one wave's registers hold them.  Here a workgroup is
// NT = H/32 waves and wave w OWNS output features 32w..32w+31 of every layer:
//   * its weight fragments (W1 rows, W1^T rows, W0^T slice) live in ITS registers for the whole kernel -- the 512 KB
//     register file of a CU is the only on-chip memory large enough -- and are read from HBM once per workgroup;
//   * the activations of the one 32-row tile in flight are shared through LDS as split bf16 images
//     [part h/m/l][8-column chunk][row][8 columns] (chunk stride 576 B): a wave splits the 32x32 tile it produced ONCE,
//     and every consumer reads fragments with ds_read_b128 (layer products: rows are the MFMA n dimension) or
//     ds_read_b64_tr_b16 (weight gradients: rows are the MFMA k dimension) -- both conflict-free on this layout
//     (tools/lds_bank_sim.py);
//   * weight gradients are split-bf16 products of those transposed fragments (no fp32 MFMA, which blocks the SIMD's
//     vector issue for its whole duration: DESIGN.md 3.5), accumulated in registers over the workgroup's whole row
//     loop and flushed once with 128-byte-segment atomics;
//   * d(hash features) = sum over the waves' feature slices: partial 32x32 tiles meet in LDS, each wave reduces and
//     stores its share of the level planes as full 256-byte lines.
// All products are the 6-term split-bf16 products of split_bf16.h (fp32-equivalent, 1e-7 relative).
#include "common.h"
#include "split_bf16.h"

namespace {

typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glob_void;

constexpr int CS = 576;        // image chunk stride: 32 rows x 16 B + 64 B so that (CS / 4) % 64 == 16
constexpr int XCH = 8;         // X image: 64 columns (36 used: 32 hash features, xyz, 1)
constexpr int X_PART = XCH * CS;
constexpr int W0CH = 6;        // W0 image: 48 k columns
constexpr int DXS = 33;        // fp32 partial-tile row stride

__device__ __forceinline__ float softplus100c(float z)
{
    const float e = __builtin_amdgcn_exp2f(-144.26950408889634f * fabsf(z));
    return fmaf(__builtin_amdgcn_logf(1.0f + e), 0.0069314718055994531f, max0(z));
}
__device__ __forceinline__ float softplus100c_grad(float h) { return 1.0f - __builtin_amdgcn_exp2f(-144.26950408889634f * h); }

__device__ __forceinline__ int n_of(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }

template <int NT>
struct G {
    static constexpr int H = 32 * NT;
    static constexpr int THREADS = 64 * NT;
    static constexpr int HCH = H / 8;
    static constexpr int H_PART = HCH * CS;
    static constexpr int W0CS = H * 16 + 64;          // (W0CS / 4) % 64 == 16 for H = 32, 64, 128
    static constexpr int W0_PART = W0CH * W0CS;
    // Layer-1 weights: a shared LDS image at H = 128 (one workgroup per CU, registers are full), per-wave register
    // fragments below that (two or four workgroups per CU must share the 160 KB of LDS: <= 80 KB each at H = 64)
    static constexpr bool W0_LDS = NT == 4;
    static constexpr int LPW = 16 / NT;               // hash levels staged per wave
    // byte offsets into dynamic LDS
    static constexpr int XI = 0;                                  // two X images (tile parity): the weight-gradient
    static constexpr int H1I = XI + 2 * 3 * X_PART;               // reads of tile t overlap the staging of tile t + 1
    static constexpr int DZI = H1I + 3 * H_PART;
    static constexpr int W0I = DZI + 3 * H_PART;
    static constexpr int DXP = W0I + (W0_LDS ? 3 * W0_PART : 0);  // fp32 [NT][32][DXS]
    static constexpr int TAB = DXP + NT * 32 * DXS * 4;           // fp32 b1 [NT][2][16], w2r0 [NT][2][16]
    static constexpr int SP = TAB + 2 * H * 4;                    // fp32 [NT][32] partial SDF sums (forward)
    static constexpr int RAW = SP + NT * 32 * 4;                  // fp32 [2][18][64]: LDS-DMA landing zone of the next tile
    static constexpr int END = RAW + 2 * 18 * 256;
    static constexpr int N_XI = 2;
};

// Forward-only layout ("lean"): one X image, the H1 image, tables, partial sums, landing zone.  52 KB at H = 128, so two
// workgroups share a CU (two waves per SIMD: the second hides the first one's barrier and LDS latencies, and a SIMD
// issues a vector instruction every 2 cycles with two waves against every 4 with one).  W0 fragments are per-wave
// registers; the last layer's feature rows are NOT in this kernel (the caller runs them as one per-layer product on h2c).
template <int NT>
struct GF {
    static constexpr int H = 32 * NT;
    static constexpr int THREADS = 64 * NT;
    static constexpr int HCH = H / 8;
    static constexpr int H_PART = HCH * CS;
    static constexpr int W0CS = 0, W0_PART = 0, W0I = 0;
    static constexpr bool W0_LDS = false;
    static constexpr int LPW = 16 / NT;
    static constexpr int XI = 0;
    static constexpr int H1I = XI + 3 * X_PART;
    static constexpr int TAB = H1I + 3 * H_PART;
    static constexpr int SP = TAB + 2 * H * 4;                    // fp32 [2][NT][32] partial SDF sums (tile parity)
    static constexpr int RAW = SP + 2 * NT * 32 * 4;
    static constexpr int END = RAW + 2 * 18 * 256;
    static constexpr int N_XI = 1;
};

// ---- image addressing -------------------------------------------------------------------------------------------
__device__ __forceinline__ int img_off(int row, int col) { return (col >> 3) * CS + row * 16 + (col & 7) * 2; }

__device__ __forceinline__ u32x4 lds_b128(const unsigned char *p) { return *reinterpret_cast<const u32x4 *>(p); }

// B fragment of a layer product: lane (row c, half hf) reads columns 16 ks2 + 8 hf .. +7 of k tile kt (natural order)
__device__ __forceinline__ Frag3 row_frag(const unsigned char *img, int part_stride, int kt, int s, int c, int hf)
{
    const unsigned char *p = img + (4 * kt + 2 * s + hf) * CS + c * 16;
    Frag3 f;
    f.h = lds_b128(p);
    f.m = lds_b128(p + part_stride);
    f.l = lds_b128(p + 2 * part_stride);
    return f;
}

__device__ __forceinline__ void tr2(const unsigned char *p, unsigned &a, unsigned &b)
{
    const v4i16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16 *)p);
    const unsigned long long u = __builtin_bit_cast(unsigned long long, r);
    a = (unsigned)u;
    b = (unsigned)(u >> 32);
}
// Fragment whose k dimension is the tile's ROWS (weight-gradient products): lane (column 32 tile + (lane & 31), h = lane >> 5),
// k-step ks covers rows 16 ks + 8 h .. +7.  Checked on the box by tools/tr_read_check.hip.
__device__ __forceinline__ Frag3 tr_frag(const unsigned char *img, int part_stride, int tile, int ks, int lane)
{
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int row = 16 * ks + 8 * (g >> 1) + q;
    const unsigned char *a = img + (4 * tile + 2 * (g & 1) + (p >> 1)) * CS + row * 16 + (p & 1) * 8;
    Frag3 f;
    unsigned x0, x1, y0, y1;
    tr2(a, x0, x1);
    tr2(a + 64, y0, y1);                                  // rows + 4
    f.h = u32x4{x0, x1, y0, y1};
    tr2(a + part_stride, x0, x1);
    tr2(a + part_stride + 64, y0, y1);
    f.m = u32x4{x0, x1, y0, y1};
    tr2(a + 2 * part_stride, x0, x1);
    tr2(a + 2 * part_stride + 64, y0, y1);
    f.l = u32x4{x0, x1, y0, y1};
    return f;
}

// A operand of the d(hash features) product when the layer-1 weights live in the LDS image (H = 128):
// A[i = hash column c][k' = n_local] = W0[32 w + n_local][c], n_local in the order of a register-resident accumulator tile
// used as B: element j of k-step s <-> n_local = 16 s + 8 (j >> 2) + 4 hf + (j & 3).  Transposed read of rows n, columns k.
template <int NT>
__device__ __forceinline__ Frag3 w0t_frag(const unsigned char *w0i, int w, int s, int lane)
{
    using L = G<NT>;
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int row = 32 * w + 16 * s + 4 * (g >> 1) + q;
    const unsigned char *a = w0i + (2 * (g & 1) + (p >> 1)) * L::W0CS + row * 16 + (p & 1) * 8;
    Frag3 f;
    unsigned x0, x1, y0, y1;
    tr2(a, x0, x1);
    tr2(a + 128, y0, y1);                                 // rows + 8
    f.h = u32x4{x0, x1, y0, y1};
    tr2(a + L::W0_PART, x0, x1);
    tr2(a + L::W0_PART + 128, y0, y1);
    f.m = u32x4{x0, x1, y0, y1};
    tr2(a + 2 * L::W0_PART, x0, x1);
    tr2(a + 2 * L::W0_PART + 128, y0, y1);
    f.l = u32x4{x0, x1, y0, y1};
    return f;
}

// The producing wave's 32x32 accumulator tile (features 32 w + n_of(r, hf), row c) -> split once -> image.  Returns the
// two fragments (k-steps of the PERMUTED feature order) for products that consume the tile straight from registers.
__device__ __forceinline__ void split_tile(const f32x16 &v, Frag3 (&f)[2])
{
#pragma unroll
    for (int s = 0; s < 2; ++s)
        f[s] = split_frag(v[8 * s], v[8 * s + 1], v[8 * s + 2], v[8 * s + 3], v[8 * s + 4], v[8 * s + 5], v[8 * s + 6],
                          v[8 * s + 7]);
}
__device__ __forceinline__ void store_tile(unsigned char *img, int part_stride, int w, int c, int hf, const Frag3 (&f)[2])
{
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        // elements 0..3: features 16 s + 4 hf + 0..3 (chunk 4 w + 2 s), elements 4..7: + 8 (next chunk)
        unsigned char *p = img + (4 * w + 2 * s) * CS + c * 16 + 8 * hf;
        *reinterpret_cast<uint2 *>(p) = uint2{f[s].h[0], f[s].h[1]};
        *reinterpret_cast<uint2 *>(p + CS) = uint2{f[s].h[2], f[s].h[3]};
        if (!RSDF_SPLIT3) continue;      // bf16 build: the middle / low images are never read
        *reinterpret_cast<uint2 *>(p + part_stride) = uint2{f[s].m[0], f[s].m[1]};
        *reinterpret_cast<uint2 *>(p + part_stride + CS) = uint2{f[s].m[2], f[s].m[3]};
        *reinterpret_cast<uint2 *>(p + 2 * part_stride) = uint2{f[s].l[0], f[s].l[1]};
        *reinterpret_cast<uint2 *>(p + 2 * part_stride + CS) = uint2{f[s].l[2], f[s].l[3]};
    }
}

__device__ __forceinline__ f32x16 mma6f(const Frag3 &a, const Frag3 &b, f32x16 c) { return mma6r(a, b, c); }
// first product of a chain: the matrix instruction takes the constant 0 as its C operand (no register zeroing)
__device__ __forceinline__ f32x16 mma6z(const Frag3 &a, const Frag3 &b)
{
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!RSDF_SPLIT3) return mma_bf16(a.h, b.h, z);
    f32x16 c = mma_bf16(a.l, b.h, z);
    c = mma_bf16(a.h, b.l, c);
    c = mma_bf16(a.m, b.m, c);
    c = mma_bf16(a.m, b.h, c);
    c = mma_bf16(a.h, b.m, c);
    c = mma_bf16(a.h, b.h, c);
    return c;
}
// "all fragment reads of a batch, then its products": measured faster at H <= 64 (46.3 vs 47.1 ms per launch); at H = 128
// the pinned order costs registers (spills) and time (179 vs 109 ms), so the compiler schedules freely there.
#ifdef RSDF_NO_FENCE
#define RSDF_SCHED_FENCE()
#else
#define RSDF_SCHED_FENCE()                                   \
    do {                                                     \
        if (NT <= 2) __builtin_amdgcn_sched_barrier(0);      \
    } while (0)
#endif

__device__ __forceinline__ f32x16 zero16()
{
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = 0.0f;
    return v;
}

struct Src {
    const float *x7t;
    const float *planes;
    int64_t S;
    int n_levels, n_active;
    float xyz_scale, xyz_offset;
};

// Stage the per-workgroup constants: zeroed images, the "1" column, the W0 image (columns in X order), b1 / w2r0 tables.
//   X column order: 2 l + f = hash feature f of level l (0..31), 32..34 = xyz, 35 = 1 (carries b0), 36..63 = 0.
template <typename L>
__device__ __forceinline__ void stage_common(unsigned char *smem, const float *__restrict__ w0, const float *__restrict__ b0,
                                             const float *__restrict__ b1, const float *__restrict__ w2, int n_levels)
{
    const int K0 = 3 + 2 * n_levels;
    for (int e = threadIdx.x; e < L::END / 4; e += L::THREADS) reinterpret_cast<unsigned *>(smem)[e] = 0u;
    __syncthreads();
    if (threadIdx.x < 32 * L::N_XI)                               // 1.0 (h part) in every X image
        *reinterpret_cast<unsigned short *>(smem + L::XI + (threadIdx.x >> 5) * 3 * X_PART + img_off(threadIdx.x & 31, 35)) = 0x3F80;
    unsigned short *w0i = reinterpret_cast<unsigned short *>(smem + L::W0I);
    for (int e = threadIdx.x; L::W0_LDS && e < L::H * 48; e += L::THREADS) {
        const int n = e / 48, k = e - n * 48;
        float w = 0.0f;
        if (k < 32) w = k < 2 * n_levels ? w0[n * K0 + 3 + k] : 0.0f;
        else if (k < 35) w = w0[n * K0 + (k - 32)];
        else if (k == 35) w = b0[n];
        const int idx = ((k >> 3) * L::W0CS + n * 16 + (k & 7) * 2) / 2;
        store3(w0i, L::W0_PART / 2, idx, w);
    }
    float *tab = reinterpret_cast<float *>(smem + L::TAB);
    for (int e = threadIdx.x; e < L::H; e += L::THREADS) {
        const int w = e >> 5, hf = (e >> 4) & 1, r = e & 15;
        const int n = 32 * w + n_of(r, hf);
        tab[e] = b1[n];
        tab[L::H + e] = w2[n];          // row 0 of W2 [N2, H]
    }
}

// ---- X tile: planes / x7t -> registers (prefetch) -> split bf16 image ------------------------------------------------
// Prefetch of the NEXT tile's inputs by LDS-DMA (global_load_lds_dword: 4 bytes per lane straight into LDS, no registers,
// completion tracked by the wave's vmcnt).  A register prefetch does not survive here: with ~500 live registers the
// compiler parks each loaded value in an accumulator register, which needs the value, i.e. a vmcnt wait, right behind
// every load -- one memory latency per level and tile.  Addresses are clamped (always valid); masking (row beyond S, level
// >= n_active) happens at staging time.  Landing zone: raw[level 0..15 | xyz 16, 17][64 lanes] fp32, each wave DMAs and
// later reads back only its own levels, so its own vmcnt wait orders the reads (no barrier).
template <int NT>
__device__ __forceinline__ void dma_x(unsigned char *raw, const Src &src, int64_t s0, int tap, int w, int lane)
{
    constexpr int LPW = 16 / NT;
    const int64_t last = src.S - 1;
    if (s0 + 32 <= src.S) {
        // Full tile: 16-byte DMAs, four levels each (an LDS-DMA instruction costs 60-185 cycles of issue whatever its width;
        // the in-kernel stamps of the lean forward put input staging at 18 % of a tile): lane (level l0 + lane / 16, chunk
        // lane % 16) fetches rows 2 chunk, 2 chunk + 1; same [level][row][2] landing image.  dwordx4 needs dword alignment
        // only.  The last, partial tile of a launch takes the 4-byte path below.
#pragma unroll
        for (int i = 0; i < LPW; i += 4) {
            const int l = w * LPW + i + (lane >> 4);
            const float *g = src.planes + (((int64_t)(l < src.n_active ? l : 0) * 7 + tap) * src.S + s0 + 2 * (lane & 15)) * 2;
            __builtin_amdgcn_global_load_lds((glob_void *)g, (lds_void *)(raw + (w * LPW + i) * 256), 16, 0, 0);
        }
        if (w == 0 && lane < 24) {
            const float *xb = src.x7t + (int64_t)tap * src.S * 3 + s0 * 3 + 4 * lane;
            __builtin_amdgcn_global_load_lds((glob_void *)xb, (lds_void *)(raw + 16 * 256), 16, 0, 0);
        }
        return;
    }
    const int64_t r = s0 + (lane >> 1);
    const int64_t rc = r <= last ? r : last;
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int l = w * LPW + i;
        const float *g = src.planes + (((int64_t)(l < src.n_active ? l : 0) * 7 + tap) * src.S + rc) * 2 + (lane & 1);
        __builtin_amdgcn_global_load_lds((glob_void *)g, (lds_void *)(raw + l * 256), 4, 0, 0);
    }
    if (w == 0) {
        const float *xb = src.x7t + (int64_t)tap * src.S * 3;
        const int64_t e0 = s0 * 3 + lane, e1 = s0 * 3 + 64 + (lane & 31), emax = src.S * 3 - 1;
        __builtin_amdgcn_global_load_lds((glob_void *)(xb + (e0 <= emax ? e0 : emax)), (lds_void *)(raw + 16 * 256), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((glob_void *)(xb + (e1 <= emax ? e1 : emax)), (lds_void *)(raw + 17 * 256), 4, 0, 0);
    }
}
__device__ __forceinline__ void wait_vm0() { __builtin_amdgcn_s_waitcnt(0x0F70); }     // s_waitcnt vmcnt(0)
// Workgroup barrier for LDS hand-offs only.  __syncthreads() carries a workgroup-scope release fence, and with LDS-DMA in
// flight (it writes LDS) the compiler must implement that as s_waitcnt vmcnt(0): every barrier would wait for the prefetch
// of the next tile.  The hand-offs here are plain ds_write -> ds_read across waves: lgkmcnt(0) retires this wave's LDS
// operations, s_barrier does the rest.  (Each wave reads back only its OWN DMA data, behind its own vmcnt wait.)
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);      // s_waitcnt lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void put3(unsigned char *xi, int off, float v)
{
    unsigned h, m, l;
    split3_pair(v, 0.0f, h, m, l);
    *reinterpret_cast<unsigned short *>(xi + off) = (unsigned short)h;
    if (!RSDF_SPLIT3) return;
    *reinterpret_cast<unsigned short *>(xi + off + X_PART) = (unsigned short)m;
    *reinterpret_cast<unsigned short *>(xi + off + 2 * X_PART) = (unsigned short)l;
}
// Landing zone -> split bf16 X image.  Lane (row = lane >> 1, f = lane & 1) holds feature f of LPW levels.  Lanes of a
// pair swap one value per two levels: the even lane then owns both features of the even level, the odd lane both of the
// odd level, and each writes one dword per part (columns 2 l, 2 l + 1 are adjacent bf16).
template <int NT>
__device__ __forceinline__ void store_x(unsigned char *xi, const unsigned char *raw, const Src &src, int64_t s0, int w, int lane)
{
    constexpr int LPW = 16 / NT;
    const int row = lane >> 1, f = lane & 1;
    const bool ok = s0 + row < src.S;
    const float *rf = reinterpret_cast<const float *>(raw);
    float pre[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) pre[i] = rf[(w * LPW + i) * 64 + lane];
#pragma unroll
    for (int i = 0; i < LPW; i += 2) {
        const float p0 = (ok && w * LPW + i < src.n_active) ? pre[i] : 0.0f;
        const float p1 = (ok && w * LPW + i + 1 < src.n_active) ? pre[i + 1] : 0.0f;
        const float got = __shfl_xor(f ? p0 : p1, 1, 64);
        const float v0 = f ? got : p0, v1 = f ? p1 : got;                   // features 0, 1 of level w LPW + i + f
        unsigned h, m, l;
        split3_pair(v0, v1, h, m, l);
        unsigned char *p = xi + img_off(row, 2 * (w * LPW + i + f));
        *reinterpret_cast<unsigned *>(p) = h;
        *reinterpret_cast<unsigned *>(p + X_PART) = m;
        *reinterpret_cast<unsigned *>(p + 2 * X_PART) = l;
    }
    if (w == 0) {
        const float r0 = rf[16 * 64 + lane], r1 = rf[17 * 64 + lane];
        const float x0 = s0 + lane / 3 < src.S ? r0 : 0.5f, x1 = s0 + (lane + 64) / 3 < src.S ? r1 : 0.5f;
        put3(xi, img_off(lane / 3, 32 + lane % 3), x0 * src.xyz_scale + src.xyz_offset);
        if (lane < 32) put3(xi, img_off((lane + 64) / 3, 32 + (lane + 64) % 3), x1 * src.xyz_scale + src.xyz_offset);
    }
}

// ---- per-wave weight fragments (registers, loaded once) ---------------------------------------------------------------
// w0f[s]: A[i = n = 32 w + c][k = X column 16 s + 8 hf + j]  (layer-1 forward; X column order of stage_common)
__device__ __forceinline__ void load_w0f(Frag3 (&f)[3], const float *__restrict__ w0, const float *__restrict__ b0,
                                         int n_levels, int n)
{
    const int K0 = 3 + 2 * n_levels;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * s + 8 * (int)(threadIdx.x >> 5 & 1) + j;
            float x = 0.0f;
            if (k < 32) x = k < 2 * n_levels ? w0[(size_t)n * K0 + 3 + k] : 0.0f;
            else if (k < 35) x = w0[(size_t)n * K0 + (k - 32)];
            else if (k == 35) x = b0[n];
            v[j] = x;
        }
        f[s] = split_frag(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    }
}
// w1f[kt][s]: A[i = n = 32 w + c][k = 32 kt + 16 s + 8 hf + j] = W1[n][k]                 (layer-2 forward)
template <int NT>
__device__ __forceinline__ void load_w1f(Frag3 (&f)[NT][2], const float *__restrict__ w1, int w, int c, int hf)
{
    constexpr int H = 32 * NT;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float *p = w1 + (size_t)(32 * w + c) * H + 32 * kt + 16 * s + 8 * hf;
            const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
            f[kt][s] = split_frag(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
        }
}
// w1t[nt][s]: A[i = k1 = 32 w + c][k' = n = 32 nt + 16 s + 8 hf + j] = W1[n][k1]           (layer-2 backward)
template <int NT>
__device__ __forceinline__ void load_w1t(Frag3 (&f)[NT][2], const float *__restrict__ w1, int w, int c, int hf)
{
    constexpr int H = 32 * NT;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = w1[(size_t)(32 * nt + 16 * s + 8 * hf + j) * H + 32 * w + c];
            f[nt][s] = split_frag(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
        }
}

// ---- in-kernel stamps (build with -DRSDF_STAMPS; tools/stamps_coop.py): wave 0 of workgroup 0 accumulates the s_memtime
// cycles between consecutive stamp points of the lean forward into g_stamps[point]; g_stamps[15] counts the tiles
#ifdef RSDF_STAMPS
__device__ unsigned long long g_stamps[16];
struct Stamper {
    unsigned long long last, acc[16];
    bool on;
    __device__ __forceinline__ void begin(bool enable) { on = enable; for (int i = 0; i < 16; ++i) acc[i] = 0; last = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void at(int i) { const unsigned long long t = __builtin_readcyclecounter(); acc[i] += t - last; last = t; }
    __device__ __forceinline__ void flush() { if (on) for (int i = 0; i < 16; ++i) g_stamps[i] = acc[i]; }
};
#define RSDF_STAMP(st, i) (st).at(i)
#else
struct Stamper {
    __device__ __forceinline__ void begin(bool) {}
    __device__ __forceinline__ void flush() {}
};
#define RSDF_STAMP(st, i)
#endif

// ---- the two hidden layers of the tile in flight ---------------------------------------------------------------------
// On return h1 / h2 hold this wave's feature tile (activated), the H1 image is complete and visible to every wave.
template <int NT, typename L>
__device__ __forceinline__ void hidden_layers(unsigned char *smem, const unsigned char *xi, const Frag3 (&w0f)[3],
                                              const Frag3 (&w1f)[NT][2], int w, int c, int hf, f32x16 &h1, f32x16 &h2,
                                              Stamper &stp)
{
    f32x16 acc;
    {   // layer 1: all 18 fragment reads in flight, then the 18 products
        Frag3 a[3], b[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (L::W0_LDS) {
                const unsigned char *wa = smem + L::W0I + (2 * s + hf) * L::W0CS + (32 * w + c) * 16;
                a[s].h = lds_b128(wa);
                a[s].m = lds_b128(wa + L::W0_PART);
                a[s].l = lds_b128(wa + 2 * L::W0_PART);
            } else {
                a[s] = w0f[s];
            }
            b[s] = row_frag(xi, X_PART, 0, s, c, hf);                            // X chunks 2 s + hf
        }
        RSDF_SCHED_FENCE();
        acc = mma6z(a[0], b[0]);
        acc = mma6f(a[1], b[1], acc);
        acc = mma6f(a[2], b[2], acc);
    }
    RSDF_STAMP(stp, 3);    // layer-1 products issued
#pragma unroll
    for (int r = 0; r < 16; ++r) h1[r] = softplus100c(acc[r]);
    {
        Frag3 f[2];
        split_tile(h1, f);
        store_tile(smem + L::H1I, L::H_PART, w, c, hf, f);
    }
    RSDF_STAMP(stp, 4);    // Softplus + split + image store
    lds_barrier();
    RSDF_STAMP(stp, 5);    // H1 barrier
    const float *tab = reinterpret_cast<const float *>(smem + L::TAB) + (2 * w + hf) * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 t = *reinterpret_cast<const float4 *>(tab + 4 * q);        // b1
        acc[4 * q] = t.x; acc[4 * q + 1] = t.y; acc[4 * q + 2] = t.z; acc[4 * q + 3] = t.w;
    }
    constexpr int KB = NT == 2 ? 2 : 1;                                         // k tiles per batch of fragment reads (registers)
#pragma unroll
    for (int k0 = 0; k0 < NT; k0 += KB) {
        Frag3 b[KB][2];
#pragma unroll
        for (int kk = 0; kk < KB; ++kk)
#pragma unroll
            for (int s = 0; s < 2; ++s) b[kk][s] = row_frag(smem + L::H1I, L::H_PART, k0 + kk, s, c, hf);
        RSDF_SCHED_FENCE();
#pragma unroll
        for (int kk = 0; kk < KB; ++kk)
#pragma unroll
            for (int s = 0; s < 2; ++s) acc = mma6f(w1f[k0 + kk][s], b[kk][s], acc);
    }
    RSDF_STAMP(stp, 6);    // layer-2 products issued
#pragma unroll
    for (int r = 0; r < 16; ++r) h2[r] = softplus100c(acc[r]);
    RSDF_STAMP(stp, 7);    // Softplus
}

// ------------------------------------------------------------------------------------------------------------------
// forward: sdf7t [7][S]; centre taps: feature [S, N2] (nullable), h2c [S, H] (nullable)
// ------------------------------------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(64 * NT)
coop_fwd_kernel(const Src src, const float *__restrict__ w0, const float *__restrict__ b0, const float *__restrict__ w1,
                const float *__restrict__ b1, const float *__restrict__ w2, const float *__restrict__ b2, int N2,
                float *__restrict__ sdf7, float *__restrict__ feature, float *__restrict__ h2c)
{
    using L = G<NT>;
    constexpr int H = L::H;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, hf = lane >> 5;
    stage_common<G<NT>>(smem, w0, b0, b1, w2, src.n_levels);
    Frag3 w1f[NT][2], w0f[3];
    load_w1f<NT>(w1f, w1, w, c, hf);
    if (!L::W0_LDS) load_w0f(w0f, w0, b0, src.n_levels, 32 * w + c);
    // last layer (centre taps with features): wave t < ceil(N2 / 32) owns output tile t
    const int n2_tiles = (N2 + 31) / 32;
    Frag3 w2f[NT][2];
    if (feature != nullptr && w < n2_tiles) {
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    v[j] = (32 * w + c < N2) ? w2[(size_t)(32 * w + c) * H + 32 * kt + 16 * s + 8 * hf + j] : 0.0f;
                w2f[kt][s] = split_frag(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            }
    }
    const float b2_0 = b2[0];
    lds_barrier();
    const float *tab = reinterpret_cast<const float *>(smem + L::TAB);
    float *sp = reinterpret_cast<float *>(smem + L::SP);

    const int64_t n_groups = (src.S + 31) / 32;
    Stamper stp;
    stp.begin(false);
    if ((int64_t)blockIdx.x < n_groups) dma_x<NT>(smem + L::RAW, src, (int64_t)blockIdx.x * 32, 0, w, lane);
    int parity = 0;
    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int64_t s0 = g * 32;
        for (int tap = 0; tap < 7; ++tap) {
            unsigned char *xi = smem + L::XI + parity * 3 * X_PART;
            wait_vm0();                                                          // this tile's inputs have landed
            store_x<NT>(xi, smem + L::RAW + parity * 18 * 256, src, s0, w, lane);
            parity ^= 1;
            {
                const int ntap = tap == 6 ? 0 : tap + 1;
                const int64_t ng = tap == 6 ? g + gridDim.x : g;
                if (ng < n_groups) dma_x<NT>(smem + L::RAW + parity * 18 * 256, src, ng * 32, ntap, w, lane);
            }
            lds_barrier();                                                     // X image complete
            f32x16 h1, h2;
            hidden_layers<NT, G<NT>>(smem, xi, w0f, w1f, w, c, hf, h1, h2, stp);
            // SDF = W2[0,:] . h2 + b2[0]: this wave's 32 features, then across the waves through LDS
            float part = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) part = fmaf(tab[H + (2 * w + hf) * 16 + r], h2[r], part);
            part += __shfl_xor(part, 32, 64);
            if (hf == 0) sp[w * 32 + c] = part;
            const bool centre = tap == 0 && feature != nullptr;
            if (centre) {
                Frag3 f[2];
                split_tile(h2, f);
                store_tile(smem + L::DZI, L::H_PART, w, c, hf, f);               // h2 image (the dz slot is free in this kernel)
                if (h2c != nullptr && s0 + c < src.S) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4 *>(h2c + (s0 + c) * H + 32 * w + 8 * q + 4 * hf) =
                            float4{h2[4 * q], h2[4 * q + 1], h2[4 * q + 2], h2[4 * q + 3]};
                }
            }
            lds_barrier();                                                     // partial sums (+ h2 image) visible
            if (w == 0 && hf == 0 && s0 + c < src.S) {
                float acc = b2_0;
#pragma unroll
                for (int t = 0; t < NT; ++t) acc += sp[t * 32 + c];
                sdf7[(int64_t)tap * src.S + s0 + c] = acc;
            }
            if (centre && w < n2_tiles) {
                f32x16 o;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n2 = 32 * w + n_of(r, hf);
                    o[r] = n2 < N2 ? b2[n2] : 0.0f;
                }
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int s = 0; s < 2; ++s) o = mma6f(w2f[kt][s], row_frag(smem + L::DZI, L::H_PART, kt, s, c, hf), o);
                if (s0 + c < src.S) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n2 = 32 * w + n_of(r, hf);
                        if (n2 < N2) feature[(s0 + c) * N2 + n2] = o[r];
                    }
                }
            }
            // the next tile's X / H1 / h2 image writes come after barriers that every wave reaches only once it has
            // issued all reads of this tile; LDS executes a wave's operations in order
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// forward, lean form (layout GF): sdf7t [7][S]; centre taps: h2c [S, H] (nullable).  Two workgroups per CU.
// ------------------------------------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(64 * NT, 2)
coop_fwd_lean_kernel(const Src src, const float *__restrict__ w0, const float *__restrict__ b0, const float *__restrict__ w1,
                     const float *__restrict__ b1, const float *__restrict__ w2, const float *__restrict__ b2,
                     float *__restrict__ sdf7, float *__restrict__ h2c)
{
    using L = GF<NT>;
    constexpr int H = L::H;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, hf = lane >> 5;
    stage_common<L>(smem, w0, b0, b1, w2, src.n_levels);
    Frag3 w1f[NT][2], w0f[3];
    load_w1f<NT>(w1f, w1, w, c, hf);
    load_w0f(w0f, w0, b0, src.n_levels, 32 * w + c);
    const float b2_0 = b2[0];
    __syncthreads();
    const float *tab = reinterpret_cast<const float *>(smem + L::TAB);
    float *sp = reinterpret_cast<float *>(smem + L::SP);
    unsigned char *xi = smem + L::XI;

    const int64_t n_groups = (src.S + 31) / 32;
    if ((int64_t)blockIdx.x < n_groups) dma_x<NT>(smem + L::RAW, src, (int64_t)blockIdx.x * 32, 0, w, lane);
    int parity = 0;
    Stamper stp;
    stp.begin(blockIdx.x == 0 && w == 0);
    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int64_t s0 = g * 32;
        for (int tap = 0; tap < 7; ++tap) {
            RSDF_STAMP(stp, 9);                                                  // tail of the previous tile (sdf store, loop)
            wait_vm0();                                                          // this tile's inputs have landed
            RSDF_STAMP(stp, 0);                                                  // wait for the DMA
            // (every wave is past barrier (b) of the previous tile, i.e. past its reads of the X image and of sp)
            store_x<NT>(xi, smem + L::RAW + parity * 18 * 256, src, s0, w, lane);
            parity ^= 1;
            {
                const int ntap = tap == 6 ? 0 : tap + 1;
                const int64_t ng = tap == 6 ? g + gridDim.x : g;
                if (ng < n_groups) dma_x<NT>(smem + L::RAW + parity * 18 * 256, src, ng * 32, ntap, w, lane);
            }
            RSDF_STAMP(stp, 1);                                                  // X image store + next DMA issue
            lds_barrier();                                                     // (a) X image complete
            RSDF_STAMP(stp, 2);                                                  // barrier (a)
            f32x16 h1, h2;
            hidden_layers<NT, L>(smem, xi, w0f, w1f, w, c, hf, h1, h2, stp);    // barrier inside: H1 image complete
            float part = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) part = fmaf(tab[H + (2 * w + hf) * 16 + r], h2[r], part);
            part += __shfl_xor(part, 32, 64);
            // two partial-sum buffers by tile parity: wave 0 reads sp of this tile after (b) while the others may already
            // be writing the next tile's sums
            if (hf == 0) sp[(parity * NT + w) * 32 + c] = part;
            if (tap == 0 && h2c != nullptr && s0 + c < src.S) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4 *>(h2c + (s0 + c) * H + 32 * w + 8 * q + 4 * hf) =
                        float4{h2[4 * q], h2[4 * q + 1], h2[4 * q + 2], h2[4 * q + 3]};
            }
            RSDF_STAMP(stp, 8);                                                  // SDF partial + h2c store
            lds_barrier();                                                     // (b) partial sums visible
            RSDF_STAMP(stp, 10);                                                 // barrier (b)
            if (w == 0 && hf == 0 && s0 + c < src.S) {
                float acc = b2_0;
#pragma unroll
                for (int t = 0; t < NT; ++t) acc += sp[(parity * NT + t) * 32 + c];
                sdf7[(int64_t)tap * src.S + s0 + c] = acc;
            }
#ifdef RSDF_STAMPS
            stp.acc[15] += 1;
#endif
        }
    }
    stp.flush();
}

// ------------------------------------------------------------------------------------------------------------------
// backward: d_sdf7t [7][S] (+ dh2c [S, H]: d(loss)/d(h2) of the centre taps through the feature rows of the last layer,
//           nullable) -> d_planes [L][7][S][2] (nullable), dW0, db0, dW1, db1, dW2 row 0, db2[0] (atomically accumulated)
// ------------------------------------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(64 * NT)
coop_bwd_kernel(const Src src, const float *__restrict__ w0, const float *__restrict__ b0, const float *__restrict__ w1,
                const float *__restrict__ b1, const float *__restrict__ w2, const float *__restrict__ d_sdf7,
                const float *__restrict__ dh2c, float *__restrict__ d_planes, float *__restrict__ dw0,
                float *__restrict__ db0, float *__restrict__ dw1, float *__restrict__ db1, float *__restrict__ dw2,
                float *__restrict__ db2, const unsigned *__restrict__ run_if)
{
    using L = G<NT>;
    constexpr int H = L::H;
    // (mlp_x2.hip's range guard launches this kernel as the range-free route of a backward call: it runs iff the word is set)
    if (run_if != nullptr && *run_if == 0u) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, hf = lane >> 5;
    const int K0 = 3 + 2 * src.n_levels;
    stage_common<G<NT>>(smem, w0, b0, b1, w2, src.n_levels);
    Frag3 w1f[NT][2], w1t[NT][2], w0t[2], w0f[3];
    load_w1f<NT>(w1f, w1, w, c, hf);
    if (!L::W0_LDS) load_w0f(w0f, w0, b0, src.n_levels, 32 * w + c);
    load_w1t<NT>(w1t, w1, w, c, hf);
    // w0t[s]: A[i = hash column c][k' = n_local = 16 s + 8 (j >> 2) + 4 hf + (j & 3)] = W0[32 w + n_local][3 + c]
    //         (the order of a register-resident accumulator tile used as the B operand)
#pragma unroll
    for (int s = 0; s < 2 && !L::W0_LDS; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = 32 * w + 16 * s + 8 * (j >> 2) + 4 * hf + (j & 3);
            v[j] = c < 2 * src.n_levels ? w0[(size_t)n * K0 + 3 + c] : 0.0f;
        }
        w0t[s] = split_frag(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    }
    __syncthreads();
    const float *tab = reinterpret_cast<const float *>(smem + L::TAB);
    float *dxp = reinterpret_cast<float *>(smem + L::DXP);

    f32x16 gw1[NT], gw0[2];
#pragma unroll
    for (int b = 0; b < NT; ++b) gw1[b] = zero16();
    gw0[0] = zero16();
    gw0[1] = zero16();
    f32x16 gw2p = zero16(), gb1p = zero16();      // per-lane (= per row slot) partials of dW2[0][n] and db1[n]
    float gb2 = 0.0f;

    const int64_t n_groups = (src.S + 31) / 32;
    Stamper stp;
    stp.begin(false);
    if ((int64_t)blockIdx.x < n_groups) dma_x<NT>(smem + L::RAW, src, (int64_t)blockIdx.x * 32, 0, w, lane);
    int parity = 0;
    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int64_t s0 = g * 32;
        for (int tap = 0; tap < 7; ++tap) {
            unsigned char *xi = smem + L::XI + parity * 3 * X_PART;
            wait_vm0();                            // this tile's inputs have landed (and the previous tile's stores retired)
            store_x<NT>(xi, smem + L::RAW + parity * 18 * 256, src, s0, w, lane);
            parity ^= 1;
            const bool row_ok = s0 + c < src.S;
            const int64_t rowc = row_ok ? s0 + c : src.S - 1;                    // clamped: loads stay unconditional
            const float dsdf_raw = d_sdf7[(int64_t)tap * src.S + rowc];          // masked where it is used
            f32x16 dz;                             // centre taps: d(h2) through the feature rows (raw), else 0
            if (tap == 0 && dh2c != nullptr) {     // (uniform branch)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4 *>(dh2c + rowc * H + 32 * w + 8 * q + 4 * hf);
                    dz[4 * q] = v.x; dz[4 * q + 1] = v.y; dz[4 * q + 2] = v.z; dz[4 * q + 3] = v.w;
                }
            } else {
                dz = zero16();
            }
            {   // next tile's inputs: issued behind this tile's own loads, lands while the tile computes
                const int ntap = tap == 6 ? 0 : tap + 1;
                const int64_t ng = tap == 6 ? g + gridDim.x : g;
                if (ng < n_groups) dma_x<NT>(smem + L::RAW + parity * 18 * 256, src, ng * 32, ntap, w, lane);
            }
            lds_barrier();                                                     // (1) X image complete
            f32x16 h1, h2;
            hidden_layers<NT, G<NT>>(smem, xi, w0f, w1f, w, c, hf, h1, h2, stp);             // (2) inside: H1 image complete
            // ---- layer 3: d(h2) = W2[0,:] d_sdf (+ feature part); dW2[0,:] += d_sdf h2; dz2 = d(h2) sigma'(z2)
            const float dsdf = row_ok ? dsdf_raw : 0.0f;
            if (w == 0 && hf == 0) gb2 += dsdf;
            {
                float w2r[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 t = *reinterpret_cast<const float4 *>(tab + H + (2 * w + hf) * 16 + 4 * q);
                    w2r[4 * q] = t.x; w2r[4 * q + 1] = t.y; w2r[4 * q + 2] = t.z; w2r[4 * q + 3] = t.w;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    gw2p[r] = fmaf(dsdf, h2[r], gw2p[r]);
                    dz[r] = row_ok ? fmaf(w2r[r], dsdf, dz[r]) * softplus100c_grad(h2[r]) : 0.0f;
                    gb1p[r] += dz[r];
                }
            }
            {
                Frag3 f[2];
                split_tile(dz, f);
                store_tile(smem + L::DZI, L::H_PART, w, c, hf, f);
            }
            lds_barrier();                                                     // (3) dz2 image complete
            // ---- layer 2.  Critical path first: dz1[own k1] = (W1^T dz2) sigma'(z1); the weight-gradient products
            //      dW1[own n][all k] += dz2^T h1 (rows = MFMA k) only accumulate, so they are issued behind it and run on
            //      the matrix pipe while the vector ALU finishes dz1.
            f32x16 acc;
            {
                constexpr int KB = NT == 2 ? 2 : 1;
#pragma unroll
                for (int n0 = 0; n0 < NT; n0 += KB) {
                    Frag3 bz[KB][2];
#pragma unroll
                    for (int nn = 0; nn < KB; ++nn)
#pragma unroll
                        for (int s = 0; s < 2; ++s) bz[nn][s] = row_frag(smem + L::DZI, L::H_PART, n0 + nn, s, c, hf);
                    RSDF_SCHED_FENCE();
#pragma unroll
                    for (int nn = 0; nn < KB; ++nn)
#pragma unroll
                        for (int s = 0; s < 2; ++s)
                            acc = (n0 + nn + s == 0) ? mma6z(w1t[0][0], bz[0][0]) : mma6f(w1t[n0 + nn][s], bz[nn][s], acc);
                }
            }
            Frag3 a1[2];
            a1[0] = tr_frag(smem + L::DZI, L::H_PART, w, 0, lane);
            a1[1] = tr_frag(smem + L::DZI, L::H_PART, w, 1, lane);
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                Frag3 bh[2];
                bh[0] = tr_frag(smem + L::H1I, L::H_PART, b, 0, lane);
                bh[1] = tr_frag(smem + L::H1I, L::H_PART, b, 1, lane);
                RSDF_SCHED_FENCE();
                gw1[b] = mma6f(a1[0], bh[0], gw1[b]);
                gw1[b] = mma6f(a1[1], bh[1], gw1[b]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) dz[r] = acc[r] * softplus100c_grad(h1[r]);
            Frag3 f1[2];
            split_tile(dz, f1);
            lds_barrier();                                                     // (4) every wave has read dz2 / h1
            store_tile(smem + L::DZI, L::H_PART, w, c, hf, f1);
            // d(hash features): this wave's feature slice, straight from the registers
            {
                f32x16 dx;
                if (L::W0_LDS) {                          // (columns >= 2 n_levels of the image are zero)
                    const Frag3 t0 = w0t_frag<NT>(smem + L::W0I, w, 0, lane), t1 = w0t_frag<NT>(smem + L::W0I, w, 1, lane);
                    dx = mma6z(t0, f1[0]);
                    dx = mma6f(t1, f1[1], dx);
                } else {
                    dx = mma6z(w0t[0], f1[0]);
                    dx = mma6f(w0t[1], f1[1], dx);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) dxp[(w * 32 + n_of(r, hf)) * DXS + c] = dx[r];
            }
            lds_barrier();                                                     // (5) dz1 image + partial tiles complete
            // (the stores go first: they retire under the products below, before the next tile's vmcnt wait)
            if (d_planes != nullptr && s0 + (lane >> 1) < src.S) {
                // level planes [L][7][S][2]: per level 32 samples x float2 = one 256-byte line per wave store
#pragma unroll
                for (int i = 0; i < L::LPW; ++i) {
                    const int l = w * L::LPW + i;
                    if (l < src.n_active) {
                        const int col = 2 * l + (lane & 1), row = lane >> 1;
                        float v = 0.0f;
#pragma unroll
                        for (int t = 0; t < NT; ++t) v += dxp[(t * 32 + col) * DXS + row];
                        d_planes[(((int64_t)l * 7 + tap) * src.S + s0) * 2 + lane] = v;
                    }
                }
            }
            // ---- layer 1: dW0[own n][X columns] += dz1^T X (column 35 = 1 gives db0)
            {
                Frag3 a[2];
                a[0] = tr_frag(smem + L::DZI, L::H_PART, w, 0, lane);
                a[1] = tr_frag(smem + L::DZI, L::H_PART, w, 1, lane);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    Frag3 bx[2];
                    bx[0] = tr_frag(xi, X_PART, ct, 0, lane);
                    bx[1] = tr_frag(xi, X_PART, ct, 1, lane);
                    RSDF_SCHED_FENCE();
                    gw0[ct] = mma6f(a[0], bx[0], gw0[ct]);
                    gw0[ct] = mma6f(a[1], bx[1], gw0[ct]);
                }
            }
            // No barrier here: the next tile stages the OTHER X image; its H1 / dz images and partial tiles are written
            // behind its barriers (1), (2), (4), which every wave reaches only after issuing all reads of this tile.
        }
    }

    // ---- flush ------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) atomicAdd(&dw1[(size_t)(32 * w + n_of(r, hf)) * H + 32 * b + c], gw1[b][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = 32 * w + n_of(r, hf);
        if (c < 2 * src.n_levels) atomicAdd(&dw0[(size_t)n * K0 + 3 + c], gw0[0][r]);
        if (c < 3) atomicAdd(&dw0[(size_t)n * K0 + c], gw0[1][r]);
        if (c == 3) atomicAdd(&db0[n], gw0[1][r]);
    }
    // per-lane partials -> sums over the 32 row slots of each lane half
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float a = gw2p[r], b = gb1p[r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            a += __shfl_xor(a, o, 64);
            b += __shfl_xor(b, o, 64);
        }
        if (c == 0) {
            atomicAdd(&dw2[32 * w + n_of(r, hf)], a);
            atomicAdd(&db1[32 * w + n_of(r, hf)], b);
        }
    }
    if (w == 0) {
        gb2 = wave_sum(gb2);
        if (lane == 0) atomicAdd(&db2[0], gb2);
    }
}

template <typename K>
int coop_set_lds(K kern, size_t bytes)
{
    static thread_local const void *done[8] = {};
    static thread_local int done_dev[8] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (int i = 0; i < 8; ++i)
        if (done[i] == reinterpret_cast<const void *>(kern) && done_dev[i] == dev + 1) return 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)bytes);
    if (e != hipSuccess) { rsdf_set_error(hipGetErrorString(e)); return (int)e; }
    for (int i = 0; i < 8; ++i)
        if (done[i] == nullptr) { done[i] = reinterpret_cast<const void *>(kern); done_dev[i] = dev + 1; break; }
    return 0;
}

template <int NT>
unsigned coop_grid(int64_t n_samples)
{
    const int64_t groups = (n_samples + 31) / 32;
    const int64_t cap = 256 * (4 / NT);             // one wave per SIMD: 4 / NT workgroups per CU
    return (unsigned)(groups < cap ? (groups > 0 ? groups : 1) : cap);
}

}  // namespace

#ifdef RSDF_STAMPS
extern "C" int rsdf_debug_read_stamps(unsigned long long *out16)
{
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamps), 16 * sizeof(unsigned long long));
}
#endif

// entry points used by mlp_fused.hip's dispatchers
__attribute__((visibility("hidden"))) int RSDF_P(rsdf_coop_fwd)(int NT, const float *x7t, const float *planes, int n_levels, int n_active, float xyz_scale,
                  float xyz_offset, int N2, const float *w0, const float *b0, const float *w1, const float *b1,
                  const float *w2, const float *b2, int64_t n_samples, float *sdf7t, float *feature, float *h2c,
                  hipStream_t st)
{
    const Src src{x7t, planes, n_samples, n_levels, n_active, xyz_scale, xyz_offset};
    int rc;
    if (feature == nullptr || h2c != nullptr) {
        // lean kernel (two workgroups per CU); the feature rows of the last layer are one per-layer product on h2c
#define RSDF_COOP_FWD_LEAN(N)                                                                                        \
    if ((rc = coop_set_lds(coop_fwd_lean_kernel<N>, GF<N>::END))) return rc;                                         \
    coop_fwd_lean_kernel<N><<<2 * coop_grid<N>(n_samples), 64 * N, GF<N>::END, st>>>(src, w0, b0, w1, b1, w2, b2,      \
                                                                                      sdf7t, feature ? h2c : nullptr)
        if (NT == 4) { RSDF_COOP_FWD_LEAN(4); }
        else if (NT == 2) { RSDF_COOP_FWD_LEAN(2); }
        else { RSDF_COOP_FWD_LEAN(1); }
#undef RSDF_COOP_FWD_LEAN
        {
            hipError_t e_ = hipGetLastError();
            if (e_ != hipSuccess) { rsdf_set_error(hipGetErrorString(e_)); return (int)e_; }
        }
        if (feature != nullptr)
            return RSDF_P(rsdf_linear_fwd)(h2c, 32 * NT, w2, b2, n_samples, 32 * NT, N2, RSDF_ACT_NONE, feature, N2, (void *)st);
        return 0;
    }
#define RSDF_COOP_FWD(N)                                                                                             \
    if ((rc = coop_set_lds(coop_fwd_kernel<N>, G<N>::END))) return rc;                                               \
    coop_fwd_kernel<N><<<coop_grid<N>(n_samples), 64 * N, G<N>::END, st>>>(src, w0, b0, w1, b1, w2, b2, N2, sdf7t,    \
                                                                            feature, h2c)
    if (NT == 4) { RSDF_COOP_FWD(4); }
    else if (NT == 2) { RSDF_COOP_FWD(2); }
    else { RSDF_COOP_FWD(1); }
#undef RSDF_COOP_FWD
    RSDF_RETURN_LAUNCH();
}

__attribute__((visibility("hidden"))) int RSDF_P(rsdf_coop_bwd)(int NT, const float *x7t, const float *planes, int n_levels, int n_active, float xyz_scale,
                  float xyz_offset, const float *w0, const float *b0, const float *w1, const float *b1, const float *w2,
                  int64_t n_samples, const float *d_sdf7t, const float *dh2c, float *d_planes, float *dw0, float *db0,
                  float *dw1, float *db1, float *dw2, float *db2, hipStream_t st, const unsigned *run_if)
{
    const Src src{x7t, planes, n_samples, n_levels, n_active, xyz_scale, xyz_offset};
    int rc;
#define RSDF_COOP_BWD(N)                                                                                             \
    if ((rc = coop_set_lds(coop_bwd_kernel<N>, G<N>::END))) return rc;                                               \
    coop_bwd_kernel<N><<<coop_grid<N>(n_samples), 64 * N, G<N>::END, st>>>(src, w0, b0, w1, b1, w2, d_sdf7t, dh2c,    \
                                                                            d_planes, dw0, db0, dw1, db1, dw2, db2, run_if)
    if (NT == 4) { RSDF_COOP_BWD(4); }
    else if (NT == 2) { RSDF_COOP_BWD(2); }
    else { RSDF_COOP_BWD(1); }
#undef RSDF_COOP_BWD
    RSDF_RETURN_LAUNCH();
}
