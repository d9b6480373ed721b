// H3 fused backward, "quad" form: the cooperative design of mlp_coop.hip re-cut for TWO waves per SIMD.
//
// mlp_coop.hip's backward is bound by the vector issue of ONE wave per SIMD: a lone wave issues a vector
// instruction every 4 cycles where the SIMD takes one every 2, and nothing covers its barrier / LDS / MFMA-result
// latencies.  A second wave per SIMD needs the per-wave state -- weight fragments AND weight-gradient accumulators -- in
// half the register file, and with 32-feature ownership (v_mfma_f32_32x32x16_bf16 tiles) that state alone is 252
// registers at H = 64.  Here a wave owns 16 features and every product is a v_mfma_f32_16x16x32_bf16:
//   * weight fragments per wave: W1 rows, W1^T rows, W0 rows, W0^T rows = 4 x 2 k-blocks x 12 = 96 registers;
//     weight-gradient accumulators dW1 [16 x 64] + dW0 [16 x 48] = 28 registers (a 16x16 tile is 4 registers) -> the
//     kernel fits 256 registers, a workgroup is H/16 = 4 waves, two workgroups share a CU;
//   * K = 32 per instruction covers all 32 rows of a tile at once in the weight-gradient products;
//   * d(hash features) needs no cross-wave reduction: wave w computes the (16 columns x 16 rows) sub-tile
//     (w & 1, w >> 1) of the 32 x 32 result over all 64 features, B fragments from the dz1 image.
// Activation images as in mlp_coop.hip ([part][8-column chunk][row][8 columns]) but with 512-byte chunks and the row
// index of odd chunks XORed with 12: conflict-free for this lane pattern's ds_read_b128 and ds_read_b64_tr_b16
// (tools/lds_bank_sim.py, report16 / swizzle search).
// Same numerics as the other fused kernels: 6-term split-bf16 products, fp32 accumulation.
#include "common.h"
#include "split_bf16.h"

namespace {

typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glob_void;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int QCS = 512;               // chunk: 32 rows x 16 B
constexpr int QX_PART = 8 * QCS;       // X image: 64 columns (32 hash features, xyz, 1, zeros)

__device__ __forceinline__ int qoff(int row, int col)
{
    const int ch = col >> 3;
    return ch * QCS + ((row ^ ((ch & 1) * 12)) << 4) + (col & 7) * 2;
}

__device__ __forceinline__ float softplus100q(float z)
{
    const float e = __builtin_amdgcn_exp2f(-144.26950408889634f * fabsf(z));
    return fmaf(__builtin_amdgcn_logf(1.0f + e), 0.0069314718055994531f, max0(z));
}
__device__ __forceinline__ float softplus100q_grad(float h) { return 1.0f - __builtin_amdgcn_exp2f(-144.26950408889634f * h); }

__device__ __forceinline__ f32x4 mma16(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma6q(const Frag3 &a, const Frag3 &b, f32x4 c)
{
    if (!RSDF_SPLIT3) return mma16(a.h, b.h, c);
    c = mma16(a.l, b.h, c);
    c = mma16(a.h, b.l, c);
    c = mma16(a.m, b.m, c);
    c = mma16(a.m, b.h, c);
    c = mma16(a.h, b.m, c);
    c = mma16(a.h, b.h, c);
    return c;
}

template <int NW>
struct Q {
    static constexpr int H = 16 * NW;
    static constexpr int THREADS = 64 * NW;
    static constexpr int H_PART = (H / 8) * QCS;
    static constexpr int LPW = 16 / NW;
    static constexpr int XI = 0;                                  // two X images (tile parity)
    static constexpr int H1I = XI + 2 * 3 * QX_PART;
    static constexpr int DZI = H1I + 3 * H_PART;                  // dz2
    static constexpr int DZ1 = DZI + 3 * H_PART;                  // dz1 in an image of its own: no barrier between the last
    static constexpr int RAW = DZ1 + 3 * H_PART;                  //   read of dz2 and the write of dz1.  fp32 [2][18][64] landing zone
    static constexpr int END = RAW + 2 * 18 * 256;
};

struct SrcQ {
    const float *x7t;
    const float *planes;
    int64_t S;
    int n_levels, n_active;
    float xyz_scale, xyz_offset;
};

__device__ __forceinline__ u32x4 ldq128(const unsigned char *p) { return *reinterpret_cast<const u32x4 *>(p); }

// Per-lane address constants.  With the swizzle keyed on the chunk parity, every fragment address is
//   (lane constant) + (compile-time constant): the loop body carries no address arithmetic, only ds_* immediates.
struct LaneQ {
    int row;     // B fragment of a layer product (rowq): chunk 4 kb + g, row 16 rh + c16
    int tr0;     // transposed fragment (trfq): rows 8 g + q, columns 16 ft + 4 p ..
    int tr1;     //   rows + 4
    int st;      // store_q: row 16 rh + c16, columns 16 w + 4 g ..
};
__device__ __forceinline__ LaneQ lane_consts(int w, int lane)
{
    const int g = lane >> 4, c16 = lane & 15, q = c16 >> 2, p = lane & 3;
    LaneQ c;
    c.row = g * QCS + ((c16 ^ (12 * (g & 1))) << 4);
    c.tr0 = (p >> 1) * QCS + (((8 * g + q) ^ (12 * (p >> 1))) << 4) + (p & 1) * 8;
    c.tr1 = (p >> 1) * QCS + (((8 * g + 4 + q) ^ (12 * (p >> 1))) << 4) + (p & 1) * 8;
    c.st = (2 * w + (g >> 1)) * QCS + ((c16 ^ (12 * (g >> 1))) << 4) + (g & 1) * 8;
    return c;
}

// B fragment of a layer product: lane (k-group g, sample row 16 rh + c16) reads features 32 kb + 8 g .. +7
//   = qoff(16 rh + c16, 32 kb + 8 g): the row offset 16 rh does not touch the swizzled bits
__device__ __forceinline__ Frag3 rowq(const unsigned char *img, int part, int kb, int rh, const LaneQ &c)
{
    const unsigned char *p = img + c.row + kb * (4 * QCS) + rh * 256;
    Frag3 f;
    f.h = ldq128(p);
    f.m = ldq128(p + part);
    f.l = ldq128(p + 2 * part);
    return f;
}

__device__ __forceinline__ void trq(const unsigned char *p, unsigned &a, unsigned &b)
{
    const v4i16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16 *)p);
    const unsigned long long u = __builtin_bit_cast(unsigned long long, r);
    a = (unsigned)u;
    b = (unsigned)(u >> 32);
}
// Fragment whose k dimension is the tile's 32 ROWS: lane (k-group g: rows 8 g .. 8 g + 7, column 16 ft + c16).  Serves as the
// A operand (A[i = column][k = row]) and as the B operand (B[k = row][j = column]) of the weight-gradient products.
__device__ __forceinline__ Frag3 trfq(const unsigned char *img, int part, int ft, const LaneQ &c)
{
    const unsigned char *a0 = img + c.tr0 + ft * (2 * QCS);      // = qoff(8 g + q, 16 ft + 4 p)
    const unsigned char *a1 = img + c.tr1 + ft * (2 * QCS);      // = qoff(8 g + 4 + q, 16 ft + 4 p)
    Frag3 f;
    unsigned x0, x1, y0, y1;
    trq(a0, x0, x1);
    trq(a1, y0, y1);
    f.h = u32x4{x0, x1, y0, y1};
    trq(a0 + part, x0, x1);
    trq(a1 + part, y0, y1);
    f.m = u32x4{x0, x1, y0, y1};
    trq(a0 + 2 * part, x0, x1);
    trq(a1 + 2 * part, y0, y1);
    f.l = u32x4{x0, x1, y0, y1};
    return f;
}

// this wave's 16 x 16 result (features 16 w + 4 g + r, sample row 16 rh + c16) -> split once -> image
__device__ __forceinline__ void store_q(unsigned char *img, int part, int rh, const LaneQ &c, const f32x4 &v)
{
    unsigned h0, m0, l0, h1, m1, l1;
    split3_pair(v[0], v[1], h0, m0, l0);
    split3_pair(v[2], v[3], h1, m1, l1);
    unsigned char *p = img + c.st + rh * 256;                    // = qoff(16 rh + c16, 16 w + 4 g)
    *reinterpret_cast<uint2 *>(p) = uint2{h0, h1};
    *reinterpret_cast<uint2 *>(p + part) = uint2{m0, m1};
    *reinterpret_cast<uint2 *>(p + 2 * part) = uint2{l0, l1};
}

__device__ __forceinline__ void wait_vm0q() { __builtin_amdgcn_s_waitcnt(0x0F70); }     // s_waitcnt vmcnt(0)
__device__ __forceinline__ void lds_barrier_q()                                          // see mlp_coop.hip lds_barrier()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);      // s_waitcnt lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// in-kernel stamps (build with -DRSDF_STAMPS; tools/stamps_quad.sh): wave 0 of workgroup 0 accumulates the s_memtime
// cycles between consecutive stamp points into g_qstamps[point]; g_qstamps[15] counts the tiles
#ifdef RSDF_STAMPS
__device__ unsigned long long g_qstamps[16];
struct StamperQ {
    unsigned long long last, acc[16];
    bool on;
    __device__ __forceinline__ void begin(bool enable) { on = enable; for (int i = 0; i < 16; ++i) acc[i] = 0; last = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void at(int i) { const unsigned long long t = __builtin_readcyclecounter(); acc[i] += t - last; last = t; }
    __device__ __forceinline__ void flush() { if (on) for (int i = 0; i < 16; ++i) g_qstamps[i] = acc[i]; }
};
#define RSDF_QSTAMP(i) stq.at(i)
#else
struct StamperQ {
    __device__ __forceinline__ void begin(bool) {}
    __device__ __forceinline__ void flush() {}
};
#define RSDF_QSTAMP(i)
#endif

// LDS-DMA prefetch of the next tile's inputs (mlp_coop.hip dma_x): wave w lands its LPW levels (and wave 0 the points)
template <int NW>
__device__ __forceinline__ void dma_q(unsigned char *raw, const SrcQ &src, int64_t s0, int tap, int w, int lane)
{
    constexpr int LPW = 16 / NW;
    const int64_t last = src.S - 1;
    if (LPW == 4 && s0 + 32 <= src.S) {
        // Full tile: one 16-byte DMA per wave instead of four 4-byte ones (an LDS-DMA instruction costs 60-185 cycles of
        // issue whatever its width): lane (level w*4 + lane / 16, chunk lane % 16) fetches rows 2 chunk, 2 chunk + 1 of
        // its level; the landing zone is the same [level][row][2] image.  (Global addresses are 8-byte aligned only
        // when S is odd: dwordx4 needs dword alignment.)  The last, partial tile of a launch takes the narrow path below.
        const int l = w * 4 + (lane >> 4);
        const int64_t r = s0 + 2 * (lane & 15);
        const float *gp = src.planes + (((int64_t)(l < src.n_active ? l : 0) * 7 + tap) * src.S + r) * 2;
        __builtin_amdgcn_global_load_lds((glob_void *)gp, (lds_void *)(raw + w * 4 * 256), 16, 0, 0);
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
        const float *gp = src.planes + (((int64_t)(l < src.n_active ? l : 0) * 7 + tap) * src.S + rc) * 2 + (lane & 1);
        __builtin_amdgcn_global_load_lds((glob_void *)gp, (lds_void *)(raw + l * 256), 4, 0, 0);
    }
    if (w == 0) {
        const float *xb = src.x7t + (int64_t)tap * src.S * 3;
        const int64_t e0 = s0 * 3 + lane, e1 = s0 * 3 + 64 + (lane & 31), emax = src.S * 3 - 1;
        __builtin_amdgcn_global_load_lds((glob_void *)(xb + (e0 <= emax ? e0 : emax)), (lds_void *)(raw + 16 * 256), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((glob_void *)(xb + (e1 <= emax ? e1 : emax)), (lds_void *)(raw + 17 * 256), 4, 0, 0);
    }
}
__device__ __forceinline__ void put3q(unsigned char *xi, int off, float v)
{
    unsigned h, m, l;
    split3_pair(v, 0.0f, h, m, l);
    *reinterpret_cast<unsigned short *>(xi + off) = (unsigned short)h;
    if (!RSDF_SPLIT3) return;
    *reinterpret_cast<unsigned short *>(xi + off + QX_PART) = (unsigned short)m;
    *reinterpret_cast<unsigned short *>(xi + off + 2 * QX_PART) = (unsigned short)l;
}
// landing zone -> split bf16 X image (column order: 2 l + f hash features 0..31, xyz 32..34, 1 at 35, zeros after)
template <int NW>
__device__ __forceinline__ void stage_q(unsigned char *xi, const unsigned char *raw, const SrcQ &src, int64_t s0, int w, int lane)
{
    constexpr int LPW = 16 / NW;
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
#ifdef RSDF_NO_DPP
        const float got = __shfl_xor(f ? p0 : p1, 1, 64);
#else
        // lane ^ 1 through DPP quad_perm [1,0,3,2]: a vector-ALU move, no LDS round trip (ds_bpermute)
        const float got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f ? p0 : p1), 0xB1, 0xF, 0xF, true));
#endif
        const float v0 = f ? got : p0, v1 = f ? p1 : got;                   // features 0, 1 of level w LPW + i + f
        unsigned h, m, l;
        split3_pair(v0, v1, h, m, l);
        unsigned char *p = xi + qoff(row, 2 * (w * LPW + i + f));
        *reinterpret_cast<unsigned *>(p) = h;
        *reinterpret_cast<unsigned *>(p + QX_PART) = m;
        *reinterpret_cast<unsigned *>(p + 2 * QX_PART) = l;
    }
    if (w == 0) {
        const float r0 = rf[16 * 64 + lane], r1 = rf[17 * 64 + lane];
        const float x0 = s0 + lane / 3 < src.S ? r0 : 0.5f, x1 = s0 + (lane + 64) / 3 < src.S ? r1 : 0.5f;
        put3q(xi, qoff(lane / 3, 32 + lane % 3), x0 * src.xyz_scale + src.xyz_offset);
        if (lane < 32) put3q(xi, qoff((lane + 64) / 3, 32 + (lane + 64) % 3), x1 * src.xyz_scale + src.xyz_offset);
    }
}

__device__ __forceinline__ Frag3 split8(const float (&v)[8])
{
    return split_frag(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
}

// ------------------------------------------------------------------------------------------------------------------
// backward (H = 64): same contract as coop_bwd_kernel
// ------------------------------------------------------------------------------------------------------------------
template <int NW>
__global__ void __launch_bounds__(64 * NW, 2)
quad_bwd_kernel(const SrcQ src, const float *__restrict__ w0, const float *__restrict__ b0, const float *__restrict__ w1,
                const float *__restrict__ b1, const float *__restrict__ w2, const float *__restrict__ d_sdf7,
                const float *__restrict__ dh2c, float *__restrict__ d_planes, float *__restrict__ dw0,
                float *__restrict__ db0, float *__restrict__ dw1, float *__restrict__ db1, float *__restrict__ dw2,
                float *__restrict__ db2, const unsigned *__restrict__ run_if)
{
    using L = Q<NW>;
    constexpr int H = L::H;
    constexpr int KB = H / 32;                     // k-blocks of a hidden-layer product
    // (mlp_x2.hip's range guard launches this kernel as the range-free route of a backward call: it runs iff the word is set)
    if (run_if != nullptr && *run_if == 0u) return;
    static_assert(NW == 4, "the d(hash features) sub-tile assignment below is written for four waves (H = 64)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
    const int K0 = 3 + 2 * src.n_levels;
    const int fw = 16 * w + c16;                   // the feature this lane addresses in an A fragment of its wave
    const LaneQ lc = lane_consts(w, lane);

    for (int e = threadIdx.x; e < L::END / 4; e += L::THREADS) reinterpret_cast<unsigned *>(smem)[e] = 0u;
    __syncthreads();
    if (threadIdx.x < 64)                          // 1.0 (h part) in column 35 of both X images
        *reinterpret_cast<unsigned short *>(smem + L::XI + (threadIdx.x >> 5) * 3 * QX_PART + qoff(threadIdx.x & 31, 35)) = 0x3F80;

    // ---- weight fragments (A operands: lane = (row c16 of the wave's 16-row block, k-group g), 8 consecutive k)
    Frag3 w1f[KB], w1t[KB], w0f[2], w0t[KB];
    const int mt = w & 1, rhx = w >> 1;            // this wave's d(hash features) sub-tile: columns 16 mt.., rows 16 rhx..
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        float v[8];
        const float *p = w1 + (size_t)fw * H + 32 * kb + 8 * g;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[j];
        w1f[kb] = split8(v);                                                        // W1[fw][k]
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = w1[(size_t)(32 * kb + 8 * g + j) * H + fw];
        w1t[kb] = split8(v);                                                        // W1[n][fw]
        const int col = 16 * mt + c16;                                              // hash column of the dx sub-tile
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = col < 2 * src.n_levels ? w0[(size_t)(32 * kb + 8 * g + j) * K0 + 3 + col] : 0.0f;
        w0t[kb] = split8(v);                                                        // W0[n][3 + col]
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 32 * kb + 8 * g + j;                                      // X column
            float x = 0.0f;
            if (k < 32) x = k < 2 * src.n_levels ? w0[(size_t)fw * K0 + 3 + k] : 0.0f;
            else if (k < 35) x = w0[(size_t)fw * K0 + (k - 32)];
            else if (k == 35) x = b0[fw];
            v[j] = x;
        }
        w0f[kb] = split8(v);
    }
    f32x4 b1r, w2r;                                // bias of layer 2 / row 0 of W2 for features 16 w + 4 g + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        b1r[r] = b1[16 * w + 4 * g + r];
        w2r[r] = w2[16 * w + 4 * g + r];
    }
    __syncthreads();

    f32x4 gw1[H / 16], gw0[3], gw2p = {0.f, 0.f, 0.f, 0.f}, gb1p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < H / 16; ++n) gw1[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 3; ++n) gw0[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float gb2 = 0.0f;

    const int64_t n_groups = (src.S + 31) / 32;
    if ((int64_t)blockIdx.x < n_groups) dma_q<NW>(smem + L::RAW, src, (int64_t)blockIdx.x * 32, 0, w, lane);
    int parity = 0;
    StamperQ stq;
    stq.begin(blockIdx.x == 0 && w == 0);
    for (int64_t gi = blockIdx.x; gi < n_groups; gi += gridDim.x) {
        const int64_t s0 = gi * 32;
        for (int tap = 0; tap < 7; ++tap) {
            unsigned char *xi = smem + L::XI + parity * 3 * QX_PART;
            RSDF_QSTAMP(0);                        // loop overhead
            wait_vm0q();                           // this tile's inputs have landed (and the previous tile's stores retired)
            RSDF_QSTAMP(1);                        // vmcnt wait
            stage_q<NW>(xi, smem + L::RAW + parity * 18 * 256, src, s0, w, lane);
            parity ^= 1;
            bool row_ok[2];
            float dsdf_raw[2];
            f32x4 dz[2];
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                const int64_t row = s0 + 16 * rh + c16;
                row_ok[rh] = row < src.S;
                const int64_t rowc = row_ok[rh] ? row : src.S - 1;
                dsdf_raw[rh] = d_sdf7[(int64_t)tap * src.S + rowc];
                if (tap == 0 && dh2c != nullptr) {     // (uniform) centre taps: d(h2) through the feature rows
                    const float4 v = *reinterpret_cast<const float4 *>(dh2c + rowc * H + 16 * w + 4 * g);
                    dz[rh] = f32x4{v.x, v.y, v.z, v.w};
                } else {
                    dz[rh] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            {
                const int ntap = tap == 6 ? 0 : tap + 1;
                const int64_t ng = tap == 6 ? gi + gridDim.x : gi;
                if (ng < n_groups) dma_q<NW>(smem + L::RAW + parity * 18 * 256, src, ng * 32, ntap, w, lane);
            }
            RSDF_QSTAMP(2);                        // X image staging, d_sdf loads, next DMA issue
            lds_barrier_q();                                                     // (1) X image complete
            RSDF_QSTAMP(3);                        // barrier 1
            // ---- recompute layer 1
            f32x4 h1[2], h2[2];
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = mma6q(w0f[0], rowq(xi, QX_PART, 0, rh, lc), acc);
                acc = mma6q(w0f[1], rowq(xi, QX_PART, 1, rh, lc), acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) h1[rh][r] = softplus100q(acc[r]);
                store_q(smem + L::H1I, L::H_PART, rh, lc, h1[rh]);
            }
            RSDF_QSTAMP(4);                        // layer-1 recompute + Softplus + H1 store
            lds_barrier_q();                                                     // (2) H1 image complete
            RSDF_QSTAMP(5);                        // barrier 2
            // ---- recompute layer 2, then layer 3 backward: dz2 = (W2[0,:] d_sdf + feature part) sigma'(z2)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                f32x4 acc = b1r;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) acc = mma6q(w1f[kb], rowq(smem + L::H1I, L::H_PART, kb, rh, lc), acc);
                const float dsdf = row_ok[rh] ? dsdf_raw[rh] : 0.0f;
                if (w == 0 && g == 0) gb2 += dsdf;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    h2[rh][r] = softplus100q(acc[r]);
                    gw2p[r] = fmaf(dsdf, h2[rh][r], gw2p[r]);
                    dz[rh][r] = row_ok[rh] ? fmaf(w2r[r], dsdf, dz[rh][r]) * softplus100q_grad(h2[rh][r]) : 0.0f;
                    gb1p[r] += dz[rh][r];
                }
                store_q(smem + L::DZI, L::H_PART, rh, lc, dz[rh]);
            }
            RSDF_QSTAMP(6);                        // layer-2 recompute + dz2 + store
            lds_barrier_q();                                                     // (3) dz2 image complete
            RSDF_QSTAMP(7);                        // barrier 3
            // ---- layer 2 backward: dz1[own k1] = (W1^T dz2) sigma'(z1) ; dW1[own n][all k] += dz2^T h1 (K = the 32 rows)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) acc = mma6q(w1t[kb], rowq(smem + L::DZI, L::H_PART, kb, rh, lc), acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) dz[rh][r] = acc[r] * softplus100q_grad(h1[rh][r]);
            }
            {
                const Frag3 a = trfq(smem + L::DZI, L::H_PART, w, lc);
#pragma unroll
                for (int n = 0; n < H / 16; ++n) gw1[n] = mma6q(a, trfq(smem + L::H1I, L::H_PART, n, lc), gw1[n]);
            }
            store_q(smem + L::DZ1, L::H_PART, 0, lc, dz[0]);
            store_q(smem + L::DZ1, L::H_PART, 1, lc, dz[1]);
            RSDF_QSTAMP(8);                        // dz1 + dW1 products + dz1 store
            lds_barrier_q();                                                     // (4) dz1 image complete
            RSDF_QSTAMP(9);                        // barrier 4
            // ---- layer 1 backward: d(hash features) sub-tile (16 columns x 16 rows, all 64 features); dW0 += dz1^T X
            {
                f32x4 dx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) dx = mma6q(w0t[kb], rowq(smem + L::DZ1, L::H_PART, kb, rhx, lc), dx);
                // result rows = hash columns 16 mt + 4 g + r = (level 8 mt + 2 g + (r >> 1), feature r & 1); lane column =
                // sample row 16 rhx + c16: two float2 stores, 16 lanes cover 128 contiguous bytes of a level plane
                const int64_t row = s0 + 16 * rhx + c16;
                if (d_planes != nullptr && row < src.S) {
                    const int lev = 8 * mt + 2 * g;
                    if (lev < src.n_active)
                        *reinterpret_cast<float2 *>(d_planes + (((int64_t)lev * 7 + tap) * src.S + row) * 2) = float2{dx[0], dx[1]};
                    if (lev + 1 < src.n_active)
                        *reinterpret_cast<float2 *>(d_planes + (((int64_t)(lev + 1) * 7 + tap) * src.S + row) * 2) = float2{dx[2], dx[3]};
                }
            }
            {
                const Frag3 a = trfq(smem + L::DZ1, L::H_PART, w, lc);
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) gw0[ct] = mma6q(a, trfq(xi, QX_PART, ct, lc), gw0[ct]);
            }
            RSDF_QSTAMP(10);                       // dx + store + dW0 products
#ifdef RSDF_STAMPS
            stq.acc[15] += 1;
#endif
            // no barrier: the next tile stages the other X image; its H1 / dz2 / dz1 writes sit behind its barriers (1) .. (3)
        }
    }
    stq.flush();

    // ---- flush: gw1[n][r] = dW1[16 w + 4 g + r][16 n + c16]; gw0[ct][r] = dW0 image [feature][X column 16 ct + c16]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int f = 16 * w + 4 * g + r;
#pragma unroll
        for (int n = 0; n < H / 16; ++n) atomicAdd(&dw1[(size_t)f * H + 16 * n + c16], gw1[n][r]);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
            if (16 * ct + c16 < 2 * src.n_levels) atomicAdd(&dw0[(size_t)f * K0 + 3 + 16 * ct + c16], gw0[ct][r]);
        if (c16 < 3) atomicAdd(&dw0[(size_t)f * K0 + c16], gw0[2][r]);
        if (c16 == 3) atomicAdd(&db0[f], gw0[2][r]);
        float a = gw2p[r], b = gb1p[r];            // per-lane partials -> sum over the 16 sample columns
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            a += __shfl_xor(a, o, 64);
            b += __shfl_xor(b, o, 64);
        }
        if (c16 == 0) {
            atomicAdd(&dw2[f], a);
            atomicAdd(&db1[f], b);
        }
    }
    if (w == 0) {
        gb2 = wave_sum(gb2);
        if (lane == 0) atomicAdd(&db2[0], gb2);
    }
}

}  // namespace

#ifdef RSDF_STAMPS
extern "C" int rsdf_debug_read_qstamps(unsigned long long *out16)
{
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_qstamps), 16 * sizeof(unsigned long long));
}
#endif

__attribute__((visibility("hidden"))) int RSDF_P(rsdf_quad_bwd)(int H, const float *x7t, const float *planes, int n_levels, int n_active, float xyz_scale, float xyz_offset,
                  const float *w0, const float *b0, const float *w1, const float *b1, const float *w2, int64_t n_samples,
                  const float *d_sdf7t, const float *dh2c, float *d_planes, float *dw0, float *db0, float *dw1, float *db1,
                  float *dw2, float *db2, hipStream_t st, const unsigned *run_if)
{
    RSDF_CHECK_ARG(H == 64, "quad backward: H must be 64");
    const SrcQ src{x7t, planes, n_samples, n_levels, n_active, xyz_scale, xyz_offset};
    static thread_local unsigned long long attr_set = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!(attr_set >> (dev & 63) & 1ull)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(quad_bwd_kernel<4>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Q<4>::END);
        if (e != hipSuccess) { rsdf_set_error(hipGetErrorString(e)); return (int)e; }
        attr_set |= 1ull << (dev & 63);
    }
    const int64_t groups = (n_samples + 31) / 32;
    const unsigned grid = (unsigned)(groups < 512 ? (groups > 0 ? groups : 1) : 512);     // two workgroups per CU
    quad_bwd_kernel<4><<<grid, 256, Q<4>::END, st>>>(src, w0, b0, w1, b1, w2, d_sdf7t, dh2c, d_planes, dw0, db0, dw1, db1,
                                                      dw2, db2, run_if);
    RSDF_RETURN_LAUNCH();
}
