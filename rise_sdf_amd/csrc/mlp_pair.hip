// T3, layer PAIRS: two 128-wide ReLU layers of a radiance network (models/texture.py:237-327: albedo / env / secondary
// 4 hidden layers, roughness / metallic 2; models/network_utils.py:109-157 without weight_norm) per kernel, forward and
// backward, in the two-part fp16 ("x2") number format of mlp_x2.hip.
//
//     forward :  x -> ha = relu(Wa x + ba) -> hb = relu(Wb ha + bb)           ha never leaves the CU
//     backward:  (x, d hb) -> recompute ha, hb -> dz_b = d hb (hb > 0) -> dWb, dbb, d ha -> dz_a = d ha (ha > 0)
//                -> dWa, dba, dx [ (x > 0) when x is itself a ReLU output ]     nothing but x and d hb is read
//
// The per-layer kernels (mlp.hip, mlp_layer_bwd.hip) are at the memory system's rate: forward 1 KB per row and layer,
// one-pass backward 1.5-2 KB.  A 4-hidden-layer network moves 12.5 KB per row through HBM forward + backward, of which 1 KB is
// its input and output.  Here the odd activation of a pair stays on the chip (forward) or is recomputed from the pair's input
// (backward: two more products on a matrix pipe that the per-layer kernels leave 70 % idle), and a pair's input crosses HBM
// once each way: 1 KB per row and pair forward, 1.5 KB backward.
//
// Form: mlp_x2.hip's backward (a workgroup is 8 waves, wave w owns features 16 w .. 16 w + 15 of every layer, every product
// is three v_mfma_f32_16x16x32_f16 of two-part operands with fp32 accumulation) generalised from the 36-column stencil image
// to a 128-column input.  What had to move: Wb^T (the A operand of d ha = Wb^T dz_b) lives in LDS -- with Wa, Wa^T, Wb and
// both weight-gradient accumulators the register file holds no fourth fragment set -- as a [feature][k] image whose 16-byte
// units are XOR-swizzled with the feature index (a 16-lane pass of a ds_read_b128 then covers all 64 banks once).
//
// Activation images, in LDS and in HBM alike ("the pair image"):
//     [tile = row / 32][part 2 (hi, lo)][chunk = column / 8 (16)][row ^ 12 (chunk & 1)][8 columns] fp16, values x 2^6
// 16 KB per 32-row tile, the same 4 bytes per value as fp32 rows; a tile lands in LDS by linear LDS-DMA.  rsdf_pair_pack
// writes it from fp32 rows (the network's input), the forward writes hb in it for the next pair; the last pair of a network
// writes fp32 rows for the (narrow) output layer's per-layer kernels.  Rows >= n of the last tile hold finite values that
// no result depends on (their gradients are masked to zero).
//
// Ranges (class scales 2^6 for activations and weights): |x|, |activation| < 1023, |weight| < 1023; a violation shows as inf /
// nan in the forward's outputs and is counted in status[RSDF_STATUS_X2_FWD_NONFINITE].  The backward's gradient images share
// one power-of-two scale per launch derived from a caller-supplied bound on |d hb| (rsdf_pair_bound*).
#include "common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glob_void;

constexpr int H = 128, NW = 8, NTHR = 64 * NW, KB = H / 32;
constexpr float SA = 64.0f, SW = 64.0f, T = SA * SW;     // class scales: activations (incl. the input), weights; accumulator
constexpr int QCS = 512;                                  // chunk: 32 rows x 16 B
constexpr int PART = (H / 8) * QCS;                       // one part of an image: 8 KB
constexpr int IMG = 2 * PART;                             // 16 KB per 32-row tile
// LDS map
constexpr int XI = 0;                                     // two X images (tile parity)
constexpr int H1I = XI + 2 * IMG;
constexpr int DZI = H1I + IMG;                            // dz_b (backward) / the output image (forward)
constexpr int DZ1 = DZI + IMG;                            // dz_a
constexpr int WTI = DZ1 + IMG;                            // Wb^T image: [part 2][feature 128][k 128] fp16, swizzled 16-byte units
constexpr int WT_PART = H * H * 2;
constexpr int RED = WTI + 2 * WT_PART;                    // 16 floats of scratch
// the network's narrow output layer folded into its last pair (N2 <= 8): W_out [8][128] + b_out [8] fp32 in LDS; the backward
// also keeps the tile's dz_out as a 16-column image (two chunks x two parts) for the dW_out product
constexpr int OUT_W = 8 * H * 4 + 64;
constexpr int FWD_WO = DZ1;
constexpr int LDS_FWD = FWD_WO + OUT_W;                   // the forward needs X x 2, H1, O (+ the output layer's bias)
constexpr int BWD_WO = RED + 64;
constexpr int DZO = BWD_WO + OUT_W;                       // [part 2][chunk 2][32 rows][8 columns] fp16
constexpr int DZO_PART = 2 * QCS;
constexpr int DZS = DZO + 2 * DZO_PART;                   // dz_out rows of a tile [32][N2 <= 8] fp32, two buffers (parity): by DMA
constexpr int LDS_BWD = DZS + 2 * 1024;

struct Frag2 { u32x4 h, l; };

__device__ __forceinline__ unsigned pack_f16(float a, float b)
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, f16x2));
}
__device__ __forceinline__ float resid_lo(float a, unsigned hi)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi), "v"(a));
    return r;
}
__device__ __forceinline__ float resid_hi(float b, unsigned hi)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi), "v"(b));
    return r;
}
// NP = 2: v S = hi + lo, three products per fp32 product (the fp32-equivalent default).  NP = 1: the 16-bit mode of a network
// (``precision: fp16 / bf16`` in its config node; BASELINE.json configs[4] "bf16 MLP on MFMA") -- every matrix operand is rounded
// ONCE to fp16 at its class scale (11 significant bits, three more than bf16) and a product is ONE v_mfma_f32_16x16x32_f16 with
// fp32 accumulation; the lo parts of the images are neither written nor read.
template <int NP = 2>
__device__ __forceinline__ void split2_pair(float a, float b, unsigned &h, unsigned &l)
{
    h = pack_f16(a, b);
    l = NP == 2 ? pack_f16(resid_lo(a, h), resid_hi(b, h)) : 0u;
}
template <int NP = 2>
__device__ __forceinline__ Frag2 split2_frag(const float *v)
{
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    split2_pair<NP>(v[0], v[1], h0, l0);
    split2_pair<NP>(v[2], v[3], h1, l1);
    split2_pair<NP>(v[4], v[5], h2, l2);
    split2_pair<NP>(v[6], v[7], h3, l3);
    Frag2 f;
    f.h = u32x4{h0, h1, h2, h3};
    f.l = u32x4{l0, l1, l2, l3};
    return f;
}
__device__ __forceinline__ f32x4 mma16(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <int NP = 2>
__device__ __forceinline__ f32x4 mma3q(const Frag2 &a, const Frag2 &b, f32x4 c)     // small terms first
{
    if (NP == 2) {
        c = mma16(a.l, b.h, c);
        c = mma16(a.h, b.l, c);
    }
    c = mma16(a.h, b.h, c);
    return c;
}
__device__ __forceinline__ u32x4 ld128(const unsigned char *p) { return *reinterpret_cast<const u32x4 *>(p); }
__device__ __forceinline__ void tr64(const unsigned char *p, unsigned &a, unsigned &b)
{
    const v4i16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16 *)p);
    const unsigned long long u = __builtin_bit_cast(unsigned long long, r);
    a = (unsigned)u;
    b = (unsigned)(u >> 32);
}
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);      // s_waitcnt lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wait_vm0() { __builtin_amdgcn_s_waitcnt(0x0F70); }      // s_waitcnt vmcnt(0)

// byte offset of (row, col) inside one part of an image
__host__ __device__ __forceinline__ int qoff(int row, int col)
{
    const int ch = col >> 3;
    return ch * QCS + ((row ^ ((ch & 1) * 12)) << 4) + (col & 7) * 2;
}
struct LaneQ {
    int row;     // B fragment of a layer product: chunk 4 kb + g, row 16 rh + c16
    int tr0;     // transposed fragment: rows 8 g + q, columns 16 ft + 4 p ..
    int tr1;     //   rows + 4
    int st;      // this wave's result slab: row 16 rh + c16, columns 16 w + 4 g ..
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
// B fragment of a layer product: lane (k-group g, sample row 16 rh + c16) reads columns 32 kb + 8 g .. + 7
template <int NP = 2>
__device__ __forceinline__ Frag2 rowq(const unsigned char *img, int kb, int rh, const LaneQ &c)
{
    const unsigned char *p = img + c.row + kb * (4 * QCS) + rh * 256;
    Frag2 f;
    f.h = ld128(p);
    f.l = NP == 2 ? ld128(p + PART) : u32x4{0u, 0u, 0u, 0u};
    return f;
}
// fragment whose k dimension is the tile's 32 ROWS: lane (rows 8 g .. 8 g + 7, column 16 ft + c16)
template <int NP = 2>
__device__ __forceinline__ Frag2 trfq(const unsigned char *img, int ft, const LaneQ &c, int part = PART)
{
    const unsigned char *a0 = img + c.tr0 + ft * (2 * QCS), *a1 = img + c.tr1 + ft * (2 * QCS);
    Frag2 f;
    unsigned x0, x1, y0, y1;
    tr64(a0, x0, x1);
    tr64(a1, y0, y1);
    f.h = u32x4{x0, x1, y0, y1};
    if (NP == 2) {
        tr64(a0 + part, x0, x1);
        tr64(a1 + part, y0, y1);
        f.l = u32x4{x0, x1, y0, y1};
    } else {
        f.l = u32x4{0u, 0u, 0u, 0u};
    }
    return f;
}
// this wave's 16 x 16 result (columns 16 w + 4 g + r, sample row 16 rh + c16), already scaled -> split once -> image
template <int NP = 2>
__device__ __forceinline__ void store_q(unsigned char *img, int rh, const LaneQ &c, const f32x4 &v)
{
    unsigned h0, l0, h1, l1;
    split2_pair<NP>(v[0], v[1], h0, l0);
    split2_pair<NP>(v[2], v[3], h1, l1);
    unsigned char *p = img + c.st + rh * 256;
    *reinterpret_cast<uint2 *>(p) = uint2{h0, h1};
    if (NP == 2) *reinterpret_cast<uint2 *>(p + PART) = uint2{l0, l1};
}
// one tile (16 KB, linear) by LDS-DMA: 16 instructions of 1 KB, two per wave
// (block ws of a part = chunks 2 ws, 2 ws + 1 = columns 16 ws .. 16 ws + 15: blocks past the input's last 32-column group hold
// zeros that nothing reads -- K = 84: 12 of 16 KB)
template <int NP = 2>
__device__ __forceinline__ void dma_tile(unsigned char *img, const unsigned char *x, int64_t tile, int ws, int lane, int kba)
{
    if (ws >= 2 * kba) return;
    const unsigned char *tb = x + tile * IMG + lane * 16;
    __builtin_amdgcn_global_load_lds((glob_void *)(tb + ws * 1024), (lds_void *)(img + ws * 1024), 16, 0, 0);
    if (NP == 2)
        __builtin_amdgcn_global_load_lds((glob_void *)(tb + (ws + 8) * 1024), (lds_void *)(img + (ws + 8) * 1024), 16, 0, 0);
}
// 2^e with |v| 2^e < 2^14 for every |v| <= bound (bound = 0, inf or nan: 1)
__device__ __forceinline__ float grad_scale(float bound)
{
    if (!(bound > 0.0f) || !(bound < 3.0e38f)) return 1.0f;
    int e = 13 - ilogbf(bound);
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    return ldexpf(1.0f, e);
}
__device__ __forceinline__ bool f16_pos(unsigned h) { return (h & 0x8000u) == 0u && (h & 0x7fffu) != 0u; }

// ---- 16-bit mode only (NP = 1): the backward's fp32 ROW stream of a tile -- d hb (g) of a lower pair, or the forward's hb rows
// that give the top pair its ReLU mask -- lands in LDS by DMA ONE TILE AHEAD, like the X image, instead of per-lane global loads
// at the top of the tile that are needed a few hundred cycles later (round 6: with a third of the matrix instructions the
// 16-bit backward's tile is 4.5 us, of which such a load's ~1 us round trip was exposed).  The buffer lives in the second
// part of the Wb^T image, which the one-part form does not use: two 16 KB tiles (parity).  Slot s (16 bytes) of row r holds
// column chunk s ^ (r & 31): a 16-lane pass of ds_read_b128 over 16 rows of one chunk then covers all 64 banks once.
constexpr int GST_TILE = 32 * H * 4;                    // 16 KB
__device__ __forceinline__ void dma_rows(unsigned char *gst, const float *src, int64_t s0, int64_t n, int ws, int lane)
{
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = ws + 8 * k;                        // instruction i: rows 2 i, 2 i + 1 of the tile (1 KB)
        const int r = 2 * i + (lane >> 5), slot = lane & 31;
        int64_t row = s0 + r;
        row = row < n ? row : n - 1;
        const float *gp = src + row * H + 4 * (slot ^ (r & 31));
        __builtin_amdgcn_global_load_lds((glob_void *)gp, (lds_void *)(gst + i * 1024), 16, 0, 0);
    }
}
// the tile's dz_out rows [32][n_out] are one contiguous span of the [n][n_out] tensor: four 256-byte DMA instructions (waves 0..3)
__device__ __forceinline__ void dma_dzo(unsigned char *dst, const float *dz_out, int64_t s0, int64_t n, int n_out, int ws, int lane)
{
    if (ws >= 4) return;
    const int e = 64 * ws + lane;
    if (e >= 32 * n_out) return;
    int64_t idx = s0 * n_out + e;
    const int64_t last = n * n_out - 1;
    idx = idx < last ? idx : last;                       // (rows past the end: masked by row_ok at the point of use)
    __builtin_amdgcn_global_load_lds((glob_void *)(dz_out + idx), (lds_void *)(dst + ws * 256), 4, 0, 0);
}
__device__ __forceinline__ float4 staged_row4(const unsigned char *gst, int r, int chunk)     // columns 4 chunk .. + 3 of tile row r
{
    return *reinterpret_cast<const float4 *>(gst + r * (H * 4) + ((chunk ^ (r & 31)) << 4));
}

struct PairArgs {
    const unsigned char *x;          // input image
    int64_t n, tiles;
    const float *wa, *ba;            // [128][K], [128]
    const float *wb, *bb;            // [128][128], [128]
    int K;                           // real columns of x / of Wa (<= 128; image columns >= K hold zeros)
    // forward outputs
    unsigned char *out_img;          // nullable
    float *out_rows;                 // nullable, [n][128]
    int *status;
    // backward
    const float *g;                  // d hb, [n][128] fp32 rows
    int g_masked;                    // g is already dz_b (the producer applied the ReLU mask) ...
    const float *hmask;              // ... or becomes it here: hb [n][128] fp32 rows (nullable), dz_b = g (hb > 0)
    const unsigned *bound;           // bits of a bound on |g|
    float *dx;                       // nullable, [n][lddx]: columns [0, kout)
    int lddx, kout, x_relu;          // x_relu: write dx (x > 0) -- x is the ReLU output of the layer below
    float *dx2;                      // nullable: columns [k1, kout) go to dx2 [n][ld2] (column - k1) instead -- the gradients of a
    int ld2, k1;                     // two-source input as two contiguous tensors (k1 a multiple of 4)
    unsigned *dx_absmax;             // nullable: atomicMax of |dx| (the next launch's bound)
    float *dwa, *dba, *dwb, *dbb;
    // the narrow output layer on top of this pair (N2 <= 8), folded in
    const float *w_out, *b_out;      // [N2][128], [N2]
    int n_out, out_act;              // forward: y = act(W_out hb + b_out) -> y_out [n][N2] (RSDF_ACT_NONE / RSDF_ACT_SIGMOID)
    float *y_out;
    const float *dz_out;             // backward: g = dz_out [n][N2] @ W_out formed here instead of being read
    float *dw_out;                   // backward (nullable): dW_out [N2][128] += dz_out^T hb, from the hb rows that supply the mask
};

// MASKED (backward only): g is already dz_b, so hb is not recomputed and Wb's own fragments are not held.
// TOP (backward, MASKED): the pair under the network's narrow output layer -- g is formed from dz_out, the mask comes from the
// forward's hb rows, dW_out is accumulated here (a variant of its own: the pairs below keep the lean register budget)
template <bool BWD, bool MASKED, bool TOP = false, int NP = 2>
__global__ void __launch_bounds__(NTHR, 1)
pair_kernel(const PairArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
    const int ws = __builtin_amdgcn_readfirstlane(w);
    const int fw = 16 * w + c16;                   // the feature (or dx column) this lane addresses in an A fragment of its wave
    const LaneQ lc = lane_consts(w, lane);
    const int K = a.K;
    const int kba = (K + 31) >> 5, cta = (K + 15) >> 4;      // 32-column k-blocks / 16-column tiles of the input that are not all zero

    // ---- weight fragments (A operands: lane = (row c16 of the wave's 16-row block, k-group g), 8 consecutive k), x SW
    bool bad0 = false;                             // a weight beyond the class range
    Frag2 waf[KB], wbf[MASKED ? 1 : KB], wat[BWD ? KB : 1];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 32 * kb + 8 * g + j;
            v[j] = k < K ? a.wa[(size_t)fw * K + k] * SW : 0.0f;
        }
        if (!BWD) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bad0 |= !(fabsf(v[j]) < 65504.0f);
        }
        waf[kb] = split2_frag<NP>(v);                                                   // Wa[fw][k]
        if (!MASKED) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = a.wb[(size_t)fw * H + 32 * kb + 8 * g + j] * SW;
            if (!BWD) {
#pragma unroll
                for (int j = 0; j < 8; ++j) bad0 |= !(fabsf(v[j]) < 65504.0f);
            }
            wbf[kb] = split2_frag<NP>(v);                                               // Wb[fw][k]
        }
        if (BWD) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fw < K ? a.wa[(size_t)(32 * kb + 8 * g + j) * K + fw] * SW : 0.0f;
            wat[kb] = split2_frag<NP>(v);                                               // Wa[n][fw]: dx column fw
        }
    }
    float *s_wo = reinterpret_cast<float *>(smem + (BWD ? BWD_WO : FWD_WO));
    if (a.n_out > 0) {
        for (int e = threadIdx.x; e < 8 * H; e += NTHR) s_wo[e] = e < a.n_out * H ? a.w_out[e] : 0.0f;
        if (threadIdx.x < 8) s_wo[8 * H + threadIdx.x] = (!BWD && threadIdx.x < a.n_out) ? a.b_out[threadIdx.x] : 0.0f;
    }
    Frag2 wof[(!BWD) ? KB : 1];                    // forward fold: W_out rows q = c16 (zero rows >= N2), x SW
    if (!BWD && a.n_out > 0) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = c16 < a.n_out ? a.w_out[(size_t)c16 * H + 32 * kb + 8 * g + j] * SW : 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) bad0 |= !(fabsf(v[j]) < 65504.0f);
            wof[kb] = split2_frag<NP>(v);
        }
    }
    f32x4 bar, bbr;                                // biases (x T) of features 16 w + 4 g + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        bar[r] = a.ba[16 * w + 4 * g + r] * T;
        bbr[r] = a.bb[16 * w + 4 * g + r] * T;
    }
    float G2 = 1.0f, G1 = 1.0f, GZ = 1.0f;
    if (TOP) {
        for (int e = threadIdx.x; e < 2 * DZO_PART / 4; e += NTHR) reinterpret_cast<unsigned *>(smem + DZO)[e] = 0u;
    }
    if (BWD) {
        // Wb^T image: unit u (8 consecutive k of Wb[k][f], i.e. of row f of Wb^T) of feature f at unit u ^ (f & 15)
        unsigned short *e16 = reinterpret_cast<unsigned short *>(smem + WTI);
        for (int e = threadIdx.x; e < H * H; e += NTHR) {
            const int f = e & (H - 1), k = e >> 7;                                  // coalesced read of Wb[k][f]
            unsigned hh, ll;
            split2_pair<NP>(a.wb[(size_t)k * H + f] * SW, 0.0f, hh, ll);
            const int idx = f * H + ((((k >> 3) ^ (f & 15)) << 3) | (k & 7));
            e16[idx] = (unsigned short)(hh & 0xffffu);
            if (NP == 2) e16[idx + WT_PART / 2] = (unsigned short)(ll & 0xffffu);
        }
        // gradient-image scales: |dz_b| <= bound, |dz_a| <= max_k sum_n |Wb[n][k]| bound
        float cs = 0.0f;
        float *s_red = reinterpret_cast<float *>(smem + RED);
        if (threadIdx.x < H)
            for (int n = 0; n < H; ++n) cs += fabsf(a.wb[(size_t)n * H + threadIdx.x]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cs = fmaxf(cs, __shfl_xor(cs, o, 64));
        if (lane == 0 && w < 2) s_red[w] = cs;
        __syncthreads();
        const float bound2 = __uint_as_float(a.bound[0]);
        G2 = grad_scale(bound2);
        G1 = grad_scale(fmaxf(s_red[0], s_red[1]) * bound2);
        // the dz_out image of the folded dW_out product has its OWN scale, from max|dz_out| (bound[1], written next to bound[0] by
        // rsdf_pair_bound_from_out_layer): bound[0] = max|dz_out| x (largest column sum of |W_out|) is BELOW max|dz_out| when
        // the output weights are small (a fresh 1 x 128 roughness layer: column sums ~0.05), and dz_out x G2 then left fp16's
        // range -- inf in the image, NaN in dW_out after one training step (the bench's c3_step, round 5)
        if (TOP) GZ = grad_scale(__uint_as_float(a.bound[1]));
    }
    __syncthreads();

    f32x4 gwb[H / 16], gwa[H / 16], gbbp = {0.f, 0.f, 0.f, 0.f}, gbap = {0.f, 0.f, 0.f, 0.f};
    f32x4 gwo = {0.f, 0.f, 0.f, 0.f};             // folded output layer: GZ SA dW_out[4 g + r][16 w + c16]
    if (BWD) {
#pragma unroll
        for (int n = 0; n < H / 16; ++n) gwb[n] = gwa[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float k_dz1 = G1 / (SW * G2), k_dx = 1.0f / (SW * G1);
    float dxmax = 0.0f;
    bool bad = bad0;

    // (16-bit backward: the tile's fp32 row stream by DMA one tile ahead, see dma_rows)
    const float *rows_src = nullptr;
    if (BWD && NP == 1) rows_src = TOP ? a.hmask : ((a.g != nullptr && a.hmask == nullptr) ? a.g : nullptr);
    unsigned char *gst = smem + WTI + WT_PART;
    if ((int64_t)blockIdx.x < a.tiles) {
        dma_tile<NP>(smem + XI, a.x, (int64_t)blockIdx.x, ws, lane, kba);
        if (BWD && NP == 1 && rows_src != nullptr) dma_rows(gst, rows_src, (int64_t)blockIdx.x * 32, a.n, ws, lane);
        if (TOP && a.dz_out != nullptr) dma_dzo(smem + DZS, a.dz_out, (int64_t)blockIdx.x * 32, a.n, a.n_out, ws, lane);
    }
    // ---- layer a backward, input side: the dx slab (columns 16 w .. 16 w + 15, all 32 rows) of the tile whose dz_a image is in
    // LDS.  DEFERRED by one tile (round 6; -DRSDF_PAIR_NO_DEFER_DX for A/B): it runs right after barrier (1) of the NEXT tile,
    // which removes the fourth barrier of a tile (dz_a's only cross-wave reader is this product) and gives the 16 KB of dx
    // stores a whole tile to retire before the next s_waitcnt vmcnt(0) instead of being issued just before it (they are then
    // OLDER than the next tile's DMA, which that wait is for).  ``pxi``: that
    // tile's X image (the ReLU mask of dx comes from this wave's OWN slab of it, which only this wave's share of the next
    // DMA overwrites -- issued after this call in program order).
    auto emit_dx = [&](int64_t ps0, const unsigned char *pxi) {
        if (a.dx != nullptr && 16 * ws < a.kout) {
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                f32x4 dx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) dx = mma3q<NP>(wat[kb], rowq<NP>(smem + DZ1, kb, rh, lc), dx);
                const int64_t row = ps0 + 16 * rh + c16;
                const int col = 16 * w + 4 * g;
                float o[4];
                if (a.x_relu) {                // x > 0 from the hi part of the X image (same sign; zero only for x < 2^-31)
                    const uint2 xh = *reinterpret_cast<const uint2 *>(pxi + lc.st + rh * 256);
                    o[0] = f16_pos(xh.x) ? dx[0] * k_dx : 0.0f;
                    o[1] = f16_pos(xh.x >> 16) ? dx[1] * k_dx : 0.0f;
                    o[2] = f16_pos(xh.y) ? dx[2] * k_dx : 0.0f;
                    o[3] = f16_pos(xh.y >> 16) ? dx[3] * k_dx : 0.0f;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = dx[r] * k_dx;
                }
                if (row < a.n) {
                    // (a group of four columns never straddles k1: both are multiples of 4)
                    const bool second = a.dx2 != nullptr && col >= a.k1;
                    float *p = second ? a.dx2 + row * a.ld2 + (col - a.k1) : a.dx + row * a.lddx + col;
                    const int ld = second ? a.ld2 : a.lddx;
                    const int lim = (a.dx2 != nullptr && !second) ? a.k1 : a.kout;
                    if (col + 3 < lim && (ld & 3) == 0) {
                        *reinterpret_cast<float4 *>(p) = float4{o[0], o[1], o[2], o[3]};
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (col + r < lim) p[r] = o[r];
                    }
                    dxmax = fmaxf(fmaxf(dxmax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
                }
            }
        }
    };
    int parity = 0;
    for (int64_t ti = blockIdx.x; ti < a.tiles; ti += gridDim.x) {
        const int64_t s0 = ti * 32;
        const unsigned char *xi = smem + XI + parity * IMG;
        // this wave's share of the tile has landed.  (Vector-memory operations retire in issue order, and the forward's stores
        // of the previous tile are younger than this tile's DMA: a counted s_waitcnt vmcnt(n) that leaves them in flight was
        // measured in round 6 -- rsdf_pair_fwd 510 -> 524 ms per config[2] step, i.e. nothing: the plain wait stays.)
        wait_vm0();
        lds_barrier();                         // (1) every share has landed; the other X image and the H1 / DZ images are free
        parity ^= 1;
#ifndef RSDF_PAIR_NO_DEFER_DX
        if (BWD && ti != (int64_t)blockIdx.x) {
            emit_dx(s0 - (int64_t)gridDim.x * 32, smem + XI + parity * IMG);
            // TOP parks this tile's hb image in the dz_a buffer right below: every wave must have finished reading the previous
            // tile's dz_a first (the barrier count of that variant stays four; its stores still retire a tile earlier)
            if (TOP && a.dw_out != nullptr) lds_barrier();
        }
#endif
        f32x4 dz[2];
        bool row_ok[2];
        // ---- backward: the tile's per-row gradient inputs.  Round 6: the global loads are ISSUED here and CONSUMED after the
        // layer-a recompute below (which does not depend on them): at the top of the tile their ~1 us round trip was exposed
        // (one workgroup per CU: nothing else runs).  The 16-bit form takes the row stream from the DMA staging buffer instead
        // (dma_rows), and the top pair's small dz_out rows come by DMA one tile ahead in both forms (dma_dzo).
        float4 vg[2], vm[2];
        const bool top_fold = TOP && a.dz_out != nullptr;
        const bool need_mask = TOP ? true : (MASKED && a.hmask != nullptr);
        const bool staged = NP == 1 && rows_src != nullptr;
        if (BWD) {
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                const int64_t row = s0 + 16 * rh + c16;
                row_ok[rh] = row < a.n;
                const int64_t rowc = row_ok[rh] ? row : a.n - 1;
                if (!top_fold && !staged) {
                    vg[rh] = *reinterpret_cast<const float4 *>(a.g + rowc * H + 16 * w + 4 * g);
                }
                if (need_mask && !(staged && TOP)) vm[rh] = *reinterpret_cast<const float4 *>(a.hmask + rowc * H + 16 * w + 4 * g);
            }
        }
        if (ti + gridDim.x < a.tiles) {
            dma_tile<NP>(smem + XI + parity * IMG, a.x, ti + gridDim.x, ws, lane, kba);
            if (BWD && NP == 1 && rows_src != nullptr)
                dma_rows(gst + parity * GST_TILE, rows_src, (ti + gridDim.x) * 32, a.n, ws, lane);
            if (TOP && a.dz_out != nullptr)
                dma_dzo(smem + DZS + parity * 1024, a.dz_out, (ti + gridDim.x) * 32, a.n, a.n_out, ws, lane);
        }
        // ---- layer a: C = T za -> SA ha
        // (round 6: the matrix products of BOTH row halves are issued before the vector work of either; written per half, the
        // compiler keeps "product, activation, split, store" in that order and the second half's chain waits behind it)
        // (measured per variant: forward 510 -> 483 ms per config[2] step, 16-bit backward 749 -> 726; the fp32 backward, which
        // has no register to spare, 1131 -> 1176: it keeps the per-half order)
        constexpr bool BOTH_FIRST = !BWD || NP == 1;
        f32x4 ha[2];
        {
            f32x4 acca[2];
            auto prod_a = [&](int rh) {
                acca[rh] = bar;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
                    if (kb < kba) acca[rh] = mma3q<NP>(waf[kb], rowq<NP>(xi, kb, rh, lc), acca[rh]);
            };
            auto act_a = [&](int rh) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ha[rh][r] = max0(acca[rh][r]) * (SA / T);
                    // (range guard at the split point: fmaxf would swallow the NaN an overflowed operand makes downstream)
                    if (!BWD) bad |= !(acca[rh][r] < 65504.0f * (T / SA));
                }
                store_q<NP>(smem + H1I, rh, lc, ha[rh]);
            };
            if (BOTH_FIRST) { prod_a(0); prod_a(1); act_a(0); act_a(1); }
            else { prod_a(0); act_a(0); prod_a(1); act_a(1); }
        }
        if (BWD) {
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                if (top_fold) {                        // d hb = dz_out W_out, formed from 4 N2 bytes per row instead of 512 read
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    unsigned short *pz = reinterpret_cast<unsigned short *>(smem + DZO + ((16 * rh + c16) << 4));
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        if (q < a.n_out) {
                            const float dqv = row_ok[rh] ? reinterpret_cast<const float *>(smem + DZS + (parity ^ 1) * 1024)
                                                               [(16 * rh + c16) * a.n_out + q] : 0.0f;
                            if (a.dw_out != nullptr && ws == 0 && g == 0) {    // dz_out as a 16-column image (x GZ)
                                unsigned hh, ll;
                                split2_pair<NP>(dqv * GZ, 0.0f, hh, ll);
                                pz[q] = (unsigned short)(hh & 0xffffu);
                                if (NP == 2) pz[DZO_PART / 2 + q] = (unsigned short)(ll & 0xffffu);
                            }
                            const float4 wq = *reinterpret_cast<const float4 *>(s_wo + q * H + 16 * w + 4 * g);
                            acc[0] = fmaf(dqv, wq.x, acc[0]);
                            acc[1] = fmaf(dqv, wq.y, acc[1]);
                            acc[2] = fmaf(dqv, wq.z, acc[2]);
                            acc[3] = fmaf(dqv, wq.w, acc[3]);
                        }
                    }
                    dz[rh] = acc;
                } else {
                    const float4 v = staged ? staged_row4(gst + (parity ^ 1) * GST_TILE, 16 * rh + c16, 4 * w + g) : vg[rh];
                    dz[rh] = f32x4{v.x, v.y, v.z, v.w};
                }
                if (need_mask) {                       // the mask from the forward's own hb rows instead of a recompute
                    const float4 m = (staged && TOP) ? staged_row4(gst + (parity ^ 1) * GST_TILE, 16 * rh + c16, 4 * w + g)
                                                     : vm[rh];
                    dz[rh][0] = m.x > 0.0f ? dz[rh][0] : 0.0f;
                    dz[rh][1] = m.y > 0.0f ? dz[rh][1] : 0.0f;
                    dz[rh][2] = m.z > 0.0f ? dz[rh][2] : 0.0f;
                    dz[rh][3] = m.w > 0.0f ? dz[rh][3] : 0.0f;
                    if (TOP && a.dw_out != nullptr) {
                        // dW_out += dz_out^T hb needs hb as a k = rows operand: its image goes where dz_a will go later in this
                        // tile (free until barrier (3))
                        const f32x4 hs = row_ok[rh] ? f32x4{m.x * SA, m.y * SA, m.z * SA, m.w * SA} : f32x4{0.f, 0.f, 0.f, 0.f};
                        store_q<NP>(smem + DZ1, rh, lc, hs);
                    }
                }
            }
        }
        lds_barrier();                         // (2) H1 image complete
        // ---- layer b: C = T zb
        f32x4 accb[2];
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) {
            accb[rh] = bbr;
            if (!MASKED) {                     // (backward on an already masked gradient: hb is not needed at all)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) accb[rh] = mma3q<NP>(wbf[kb], rowq<NP>(smem + H1I, kb, rh, lc), accb[rh]);
            }
        }
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) {
            const f32x4 acc = accb[rh];
            if (!BWD) {
                f32x4 hb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    hb[r] = max0(acc[r]) * (SA / T);
                    bad |= !(acc[r] < 65504.0f * (T / SA));
                }
                if (a.out_img != nullptr || a.n_out > 0) store_q<NP>(smem + DZI, rh, lc, hb);
                const int64_t row = s0 + 16 * rh + c16;
                if (a.out_rows != nullptr && row < a.n)
                    *reinterpret_cast<float4 *>(a.out_rows + row * H + 16 * w + 4 * g) =
                        float4{hb[0] * (1.0f / SA), hb[1] * (1.0f / SA), hb[2] * (1.0f / SA), hb[3] * (1.0f / SA)};
            } else {
                f32x4 dzs;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool on = row_ok[rh] && (MASKED || acc[r] > 0.0f);
                    dz[rh][r] = on ? dz[rh][r] : 0.0f;
                    gbbp[r] += dz[rh][r];
                    dzs[r] = dz[rh][r] * G2;
                }
                store_q<NP>(smem + DZI, rh, lc, dzs);
            }
        }
        if (TOP && a.dw_out != nullptr)            // dW_out += dz_out^T hb (k = the 32 rows): both images were written before (2)
            gwo = mma3q<NP>(trfq<NP>(smem + DZO, 0, lc, DZO_PART), trfq<NP>(smem + DZ1, w, lc), gwo);
        lds_barrier();                         // (3) dz_b image (forward: the output image) complete
        if (!BWD) {
            if (a.n_out > 0 && ws < 2) {       // y = act(W_out hb + b_out): waves 0 / 1 take the tile's row halves, hb from its image
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) acc = mma3q<NP>(wof[kb], rowq<NP>(smem + DZI, kb, ws, lc), acc);
                const int64_t row = s0 + 16 * ws + c16;
                if (row < a.n) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int q = 4 * g + r;               // C[i = q][j = row c16]
                        if (q < a.n_out) {
                            float y = acc[r] * (1.0f / T) + s_wo[8 * H + q];
                            if (a.out_act == RSDF_ACT_SIGMOID)
                                y = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y * -1.4426950408889634f));
                            a.y_out[row * a.n_out + q] = y;
                        }
                    }
                }
            }
            if (a.out_img != nullptr) {        // 16 KB, linear: 32 bytes per thread
                unsigned char *ob = a.out_img + ti * IMG + threadIdx.x * 16;
                const unsigned char *ib = smem + DZI + threadIdx.x * 16;
                *reinterpret_cast<u32x4 *>(ob) = ld128(ib);
                if (NP == 2) *reinterpret_cast<u32x4 *>(ob + NTHR * 16) = ld128(ib + NTHR * 16);      // (NTHR * 16 = PART: the lo part)
            }
            continue;                          // (the next tile's barrier (1) orders these reads before its stores)
        }
        // ---- layer b backward: G1 dz_a[own k] = (Wb^T dz_b) (ha > 0) ; dWb[own n][all k] += dz_b^T ha (K = the 32 rows)
        {
            const unsigned char *wt = smem + WTI + fw * (H * 2);
            Frag2 wbt[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int u = ((4 * kb + g) ^ c16) << 4;
                wbt[kb].h = ld128(wt + u);
                wbt[kb].l = NP == 2 ? ld128(wt + WT_PART + u) : u32x4{0u, 0u, 0u, 0u};
            }
            f32x4 acct[2];
            auto prod_t = [&](int rh) {
                acct[rh] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) acct[rh] = mma3q<NP>(wbt[kb], rowq<NP>(smem + DZI, kb, rh, lc), acct[rh]);
            };
            auto act_t = [&](int rh) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dz[rh][r] = ha[rh][r] > 0.0f ? acct[rh][r] * k_dz1 : 0.0f;         // x G1
                    gbap[r] += dz[rh][r];
                }
            };
            if (BOTH_FIRST) { prod_t(0); prod_t(1); act_t(0); act_t(1); }
            else { prod_t(0); act_t(0); prod_t(1); act_t(1); }
        }
        {
            const Frag2 af = trfq<NP>(smem + DZI, w, lc);
#pragma unroll
            for (int n = 0; n < H / 16; ++n) gwb[n] = mma3q<NP>(af, trfq<NP>(smem + H1I, n, lc), gwb[n]);       // x G2 SA
        }
        store_q<NP>(smem + DZ1, 0, lc, dz[0]);
        store_q<NP>(smem + DZ1, 1, lc, dz[1]);
#ifdef RSDF_PAIR_NO_DEFER_DX
        lds_barrier();                         // (4) dz_a image complete
        emit_dx(s0, xi);
#else
        // no barrier (4): the dWa product below reads only this wave's OWN slab of the dz_a image (what store_q just wrote; LDS
        // operations of one wave execute in order, the wait makes it explicit); the product that needs every wave's slab runs
        // after the next tile's barrier (1) (emit_dx)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);    // s_waitcnt lgkmcnt(0)
        asm volatile("" ::: "memory");
#endif
        // ---- layer a backward, weight side: dWa += dz_a^T X
        {
            const Frag2 af = trfq<NP>(smem + DZ1, w, lc);
#pragma unroll
            for (int ct = 0; ct < H / 16; ++ct)
                if (ct < cta) gwa[ct] = mma3q<NP>(af, trfq<NP>(xi, ct, lc), gwa[ct]);                          // x G1 SA
        }
        // no barrier: the next tile's barrier (1) separates these reads from the DMA that overwrites this X image
    }

#ifndef RSDF_PAIR_NO_DEFER_DX
    if (BWD && (int64_t)blockIdx.x < a.tiles) {    // the last tile's input gradient
        lds_barrier();
        const int64_t last_ti = (int64_t)blockIdx.x + ((a.tiles - 1 - (int64_t)blockIdx.x) / gridDim.x) * gridDim.x;
        emit_dx(last_ti * 32, smem + XI + (parity ^ 1) * IMG);
    }
#endif
    if (!BWD) {
        if (__builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0 && a.status != nullptr) {
            atomicAdd(&a.status[RSDF_STATUS_X2_FWD_NONFINITE], 1);
            atomicAdd(&a.status[RSDF_STATUS_PAIR_FWD_NONFINITE], 1);
        }
        return;
    }
    // ---- flush: gwb[n][r] = G2 SA dWb[16 w + 4 g + r][16 n + c16]; gwa[ct][r] = G1 SA dWa[..][16 ct + c16]
    const float ub = 1.0f / (G2 * SA), ua = 1.0f / (G1 * SA);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int f = 16 * w + 4 * g + r;
#pragma unroll
        for (int n = 0; n < H / 16; ++n) atomicAdd(&a.dwb[(size_t)f * H + 16 * n + c16], gwb[n][r] * ub);
#pragma unroll
        for (int ct = 0; ct < H / 16; ++ct)
            if (16 * ct + c16 < K) atomicAdd(&a.dwa[(size_t)f * K + 16 * ct + c16], gwa[ct][r] * ua);
        float s = gbbp[r], t = gbap[r];            // per-lane partials -> sum over the 16 sample columns
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            s += __shfl_xor(s, o, 64);
            t += __shfl_xor(t, o, 64);
        }
        if (c16 == 0) {
            atomicAdd(&a.dbb[f], s);
            atomicAdd(&a.dba[f], t * (1.0f / G1));
        }
    }
    if (TOP && a.dw_out != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * g + r < a.n_out)
                atomicAdd(&a.dw_out[(size_t)(4 * g + r) * H + 16 * w + c16], gwo[r] * (1.0f / (GZ * SA)));
    }
    if (a.dx_absmax != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dxmax = fmaxf(dxmax, __shfl_xor(dxmax, o, 64));
        if (lane == 0 && dxmax > 0.0f) atomicMax(a.dx_absmax, __float_as_uint(dxmax));
    }
}

// fp32 rows [n][ld] (columns [0, K)) -> the pair image (columns >= K and rows >= n: zeros)
__global__ void __launch_bounds__(256)
pack_kernel(const float *__restrict__ x, int ld, int K1, const float *__restrict__ x2, int ld2, int K, int64_t n, int64_t tiles,
            unsigned char *__restrict__ img, int *__restrict__ status)
{
    // columns [0, K1) from x, [K1, K) from x2 (the reference's torch.cat([feature, encoding], -1), models/texture.py:299-313,
    // never materialised); chunks past the last 32-column group that holds a column are not written: nothing reads them
    // (round 6) Sources that allow it (K1 a multiple of 8, 16-byte-aligned rows) are read as two float4 per chunk.  Measured and
    // not kept (-DRSDF_PACK_ROWS_OVER_LANES): a tile's 32 rows over the lanes and the chunk over the half-waves, so that a
    // half-wave's stores fill one contiguous 512-byte chunk of the image -- 28.0 vs 24.0 ms per 400 x 400 config[2] step: with
    // the chunk over the lanes sixteen lanes READ one contiguous row, and that matters more than the scattered 16-byte stores.
    const int64_t total = tiles * 32 * 16;                                         // (row, chunk) pairs
    const int n_chunks = ((K + 31) >> 5) * 4;
#ifdef RSDF_PACK_SCALAR
    const bool vec1 = false, vec2 = false;
#else
    const bool vec1 = ((K1 | ld) & 3) == 0 && (K1 & 7) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const bool vec2 = x2 != nullptr && ((ld2 & 3) == 0) && (K1 & 7) == 0 && (reinterpret_cast<uintptr_t>(x2) & 15) == 0;
#endif
    bool bad = false;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
#ifndef RSDF_PACK_ROWS_OVER_LANES
        const int ch = (int)(e & 15);
        const int64_t row = e >> 4;
#else
        const int ch = (int)(e >> 5 & 15);
        const int64_t row = (e >> 9) * 32 + (e & 31);
#endif
        if (ch >= n_chunks) continue;
        float v[8];
        const int c0 = 8 * ch;
        if (row < n && vec1 && c0 + 8 <= K1) {                       // whole chunk from x (uniform per half-wave but for row < n)
            const float4 a = *reinterpret_cast<const float4 *>(x + row * ld + c0), b = *reinterpret_cast<const float4 *>(x + row * ld + c0 + 4);
            v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
        } else if (row < n && vec2 && c0 >= K1 && c0 + 8 <= K) {     // whole chunk from x2
            const float *q = x2 + row * ld2 + (c0 - K1);
            const float4 a = *reinterpret_cast<const float4 *>(q), b = *reinterpret_cast<const float4 *>(q + 4);
            v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int col = c0 + j;
                float t = 0.0f;
                if (row < n && col < K) t = col < K1 ? x[row * ld + col] : x2[row * ld2 + (col - K1)];
                v[j] = t;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] *= SA;
            bad |= !(fabsf(v[j]) < 65504.0f);
        }
        const Frag2 f = split2_frag(v);
        unsigned char *p = img + (row >> 5) * IMG + ch * QCS + ((((int)(row & 31)) ^ ((ch & 1) * 12)) << 4);
        *reinterpret_cast<u32x4 *>(p) = f.h;
        *reinterpret_cast<u32x4 *>(p + PART) = f.l;
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0 && status != nullptr) {
        atomicAdd(&status[RSDF_STATUS_X2_FWD_NONFINITE], 1);
        atomicAdd(&status[RSDF_STATUS_PAIR_PACK_NONFINITE], 1);
    }
}

// the pair image -> fp32 rows [n][128] (tests; callers that need an even activation as rows)
__global__ void __launch_bounds__(256)
unpack_kernel(const unsigned char *__restrict__ img, int64_t n, float *__restrict__ rows)
{
    const int64_t total = n * 16;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int ch = (int)(e & 15);
        const int64_t row = e >> 4;
        const unsigned char *p = img + (row >> 5) * IMG + ch * QCS + ((((int)(row & 31)) ^ ((ch & 1) * 12)) << 4);
        const f16x8 h = *reinterpret_cast<const f16x8 *>(p), l = *reinterpret_cast<const f16x8 *>(p + PART);
#pragma unroll
        for (int j = 0; j < 8; ++j) rows[row * H + 8 * ch + j] = ((float)h[j] + (float)l[j]) * (1.0f / SA);
    }
}

// bound on |d h| of the layer below a narrow output layer: max|dz_out| * max_k sum_q |W_out[q][k]|
__global__ void __launch_bounds__(128)
out_bound_kernel(const unsigned *__restrict__ absmax_dz, const float *__restrict__ w_out, int N2, unsigned *__restrict__ bound)
{
    float cs = 0.0f;
    for (int q = 0; q < N2; ++q) cs += fabsf(w_out[(size_t)q * H + threadIdx.x]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cs = fmaxf(cs, __shfl_xor(cs, o, 64));
    __shared__ float s[2];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = cs;
    __syncthreads();
    if (threadIdx.x == 0) bound[0] = __float_as_uint(fmaxf(s[0], s[1]) * __uint_as_float(absmax_dz[0]));
}

__global__ void __launch_bounds__(256)
absmax_kernel(const float *__restrict__ v, int64_t n, unsigned *__restrict__ out)
{
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(v[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(out, __float_as_uint(m));
}

// max |dz_out| and the column sums of dz_out [n][N2] (N2 <= 8) in one pass: a thread walks rows, a workgroup adds its sums once
__global__ void __launch_bounds__(256)
absmax_colsum_kernel(const float *__restrict__ v, int64_t n, int N2, unsigned *__restrict__ out, float *__restrict__ colsum)
{
    float m = 0.0f, s[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) s[q] = 0.0f;
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < n; row += (int64_t)gridDim.x * 256) {
        const float *p = v + row * N2;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q < N2) {
                const float t = p[q];
                m = fmaxf(m, fabsf(t));
                s[q] += t;
            }
    }
    __shared__ float red[4][8];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m = fmaxf(m, __shfl_xor(m, o, 64));
#pragma unroll
        for (int q = 0; q < 8; ++q) s[q] += __shfl_xor(s[q], o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (m > 0.0f) atomicMax(out, __float_as_uint(m));
#pragma unroll
        for (int q = 0; q < 8; ++q) red[threadIdx.x >> 6][q] = s[q];
    }
    __syncthreads();
    if (threadIdx.x < N2) {
        const int q = threadIdx.x;
        atomicAdd(&colsum[q], (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]));
    }
}

}  // namespace

extern "C" {

int rsdf_pair_supported(int K, int Na, int Nb) { return (K >= 1 && K <= 128 && Na == 128 && Nb == 128) ? 1 : 0; }

int64_t rsdf_pair_image_bytes(int64_t n_rows) { return ((n_rows + 31) / 32) * (int64_t)IMG + 1024; }

int rsdf_pair_pack(const float *x, int ldx, int K, int64_t n, void *image, int *status, void *stream)
{
    return rsdf_pair_pack2(x, ldx, K, nullptr, 0, 0, n, image, status, stream);
}

int rsdf_pair_pack2(const float *x1, int ld1, int K1, const float *x2, int ld2, int K2, int64_t n, void *image, int *status,
                    void *stream)
{
    const int K = K1 + K2;
    RSDF_CHECK_ARG(K1 >= 1 && K2 >= 0 && K <= 128 && ld1 >= K1 && (K2 == 0 || (x2 != nullptr && ld2 >= K2)),
                   "pair_pack: 1 <= K1, K1 + K2 <= 128, ld >= K");
    if (n <= 0) return 0;
    const int64_t tiles = (n + 31) / 32, work = tiles * 32 * 16;
    const unsigned grid = (unsigned)((work + 255) / 256 < 65536 ? (work + 255) / 256 : 65536);
    pack_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x1, ld1, K1, x2, ld2, K, n, tiles, reinterpret_cast<unsigned char *>(image),
                                                      status);
    RSDF_RETURN_LAUNCH();
}

int rsdf_pair_unpack(const void *image, int64_t n, float *rows, void *stream)
{
    if (n <= 0) return 0;
    const int64_t work = n * 16;
    const unsigned grid = (unsigned)((work + 255) / 256 < 65536 ? (work + 255) / 256 : 65536);
    unpack_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const unsigned char *>(image), n, rows);
    RSDF_RETURN_LAUNCH();
}

static int pair_fwd_impl(int parts, const void *x_image, int K, const float *wa, const float *ba, const float *wb, const float *bb,
                         int64_t n, void *out_image, float *out_rows, const float *w_out, const float *b_out, int N2, int out_act,
                         float *y_out, int *status, void *stream)
{
    RSDF_CHECK_ARG(K >= 1 && K <= 128, "pair_fwd: K must be in [1,128]");
    RSDF_CHECK_ARG(out_image != nullptr || out_rows != nullptr || y_out != nullptr, "pair_fwd: no output");
    RSDF_CHECK_ARG(y_out == nullptr || (w_out != nullptr && b_out != nullptr && N2 >= 1 && N2 <= 8 &&
                                        (out_act == RSDF_ACT_NONE || out_act == RSDF_ACT_SIGMOID)),
                   "pair_fwd: the folded output layer needs W_out, b_out, 1 <= N2 <= 8 and activation none / sigmoid");
    if (n <= 0) return 0;
    PairArgs a{};
    a.x = reinterpret_cast<const unsigned char *>(x_image);
    a.n = n;
    a.tiles = (n + 31) / 32;
    a.wa = wa, a.ba = ba, a.wb = wb, a.bb = bb, a.K = K;
    a.out_img = reinterpret_cast<unsigned char *>(out_image);
    a.out_rows = out_rows;
    a.status = status;
    if (y_out != nullptr) a.w_out = w_out, a.b_out = b_out, a.n_out = N2, a.out_act = out_act, a.y_out = y_out;
    // (persistent workgroups, one per CU; the 16-bit forward needs 108 registers and 68 KB of LDS: two fit a CU)
    const int64_t wgs = parts == 1 ? 512 : 256;
    const unsigned grid = (unsigned)(a.tiles < wgs ? a.tiles : wgs);
    if (parts == 1) {
        if (int rc = rsdf_func_lds(reinterpret_cast<const void *>(pair_kernel<false, false, false, 1>), LDS_FWD)) return rc;
        pair_kernel<false, false, false, 1><<<grid, NTHR, LDS_FWD, (hipStream_t)stream>>>(a);
    } else {
        if (int rc = rsdf_func_lds(reinterpret_cast<const void *>(pair_kernel<false, false>), LDS_FWD)) return rc;
        pair_kernel<false, false><<<grid, NTHR, LDS_FWD, (hipStream_t)stream>>>(a);
    }
    RSDF_RETURN_LAUNCH();
}

int rsdf_pair_fwd(const void *x_image, int K, const float *wa, const float *ba, const float *wb, const float *bb, int64_t n,
                  void *out_image, float *out_rows, const float *w_out, const float *b_out, int N2, int out_act, float *y_out,
                  int *status, void *stream)
{
    return pair_fwd_impl(2, x_image, K, wa, ba, wb, bb, n, out_image, out_rows, w_out, b_out, N2, out_act, y_out, status, stream);
}

int rsdf_pair_fwd16(const void *x_image, int K, const float *wa, const float *ba, const float *wb, const float *bb, int64_t n,
                    void *out_image, float *out_rows, const float *w_out, const float *b_out, int N2, int out_act, float *y_out,
                    int *status, void *stream)
{
    return pair_fwd_impl(1, x_image, K, wa, ba, wb, bb, n, out_image, out_rows, w_out, b_out, N2, out_act, y_out, status, stream);
}

int rsdf_pair_bound_from_rows(const float *g, int64_t count, void *bound, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(bound, 0, 4, st);
    if (count > 0) absmax_kernel<<<1024, 256, 0, st>>>(g, count, reinterpret_cast<unsigned *>(bound));
    RSDF_RETURN_LAUNCH();
}

int rsdf_pair_bound_from_out_layer(const float *dz_out, int64_t n, int N2, const float *w_out, void *bound, float *db_out,
                                   void *stream)
{
    RSDF_CHECK_ARG(N2 >= 1, "pair_bound_from_out_layer: N2 must be >= 1");
    RSDF_CHECK_ARG(db_out == nullptr || N2 <= 8, "pair_bound_from_out_layer: db_out needs N2 <= 8");
    hipStream_t st = (hipStream_t)stream;
    unsigned *b = reinterpret_cast<unsigned *>(bound);
    (void)hipMemsetAsync(b, 0, 8, st);
    if (n > 0 && db_out != nullptr) absmax_colsum_kernel<<<256, 256, 0, st>>>(dz_out, n, N2, b + 1, db_out);
    else if (n > 0) absmax_kernel<<<256, 256, 0, st>>>(dz_out, n * N2, b + 1);
    out_bound_kernel<<<1, 128, 0, st>>>(b + 1, w_out, N2, b);
    RSDF_RETURN_LAUNCH();
}

static int pair_bwd_impl(int parts, const void *x_image, int K, const float *wa, const float *ba, const float *wb, const float *bb, int64_t n,
                         const float *g, int g_masked, const float *hb_rows, const float *dz_out, const float *w_out, int N2,
                         float *dw_out, const void *bound, float *dx, int lddx, int kout, float *dx2, int ld2, int k1, int x_relu,
                         void *dx_absmax, float *dwa, float *dba, float *dwb, float *dbb, void *stream)
{
    RSDF_CHECK_ARG(dx2 == nullptr || (dx != nullptr && k1 >= 4 && (k1 & 3) == 0 && k1 < kout && ld2 >= kout - k1),
                   "pair_bwd: the second dx output needs dx, 0 < k1 < kout, k1 a multiple of 4, ld2 >= kout - k1");
    RSDF_CHECK_ARG(K >= 1 && K <= 128, "pair_bwd: K must be in [1,128]");
    RSDF_CHECK_ARG((g != nullptr || dz_out != nullptr) && bound != nullptr, "pair_bwd: g (or dz_out) and its bound are required");
    RSDF_CHECK_ARG(dz_out == nullptr || (hb_rows != nullptr && w_out != nullptr && N2 >= 1 && N2 <= 8),
                   "pair_bwd: the folded output layer needs hb_rows, W_out and 1 <= N2 <= 8");
    RSDF_CHECK_ARG(dx == nullptr || (kout >= 1 && kout <= K && lddx >= (dx2 != nullptr ? k1 : kout)), "pair_bwd: bad dx window");
    if (n <= 0) return 0;
    PairArgs a{};
    a.x = reinterpret_cast<const unsigned char *>(x_image);
    a.n = n;
    a.tiles = (n + 31) / 32;
    a.wa = wa, a.ba = ba, a.wb = wb, a.bb = bb, a.K = K;
    a.g = g, a.g_masked = (g_masked || hb_rows != nullptr) ? 1 : 0, a.hmask = g_masked ? nullptr : hb_rows;
    a.bound = reinterpret_cast<const unsigned *>(bound);
    if (dz_out != nullptr) a.dz_out = dz_out, a.w_out = w_out, a.n_out = N2, a.dw_out = dw_out;
    a.dx = dx, a.lddx = lddx, a.kout = kout, a.x_relu = x_relu;
    a.dx2 = dx2, a.ld2 = ld2, a.k1 = k1;
    a.dx_absmax = reinterpret_cast<unsigned *>(dx_absmax);
    a.dwa = dwa, a.dba = dba, a.dwb = dwb, a.dbb = dbb;
    const unsigned grid = (unsigned)(a.tiles < 256 ? a.tiles : 256);
#define RSDF_PAIR_BWD_LAUNCH(NPP)                                                                                                  \
    do {                                                                                                                          \
        if (a.dz_out != nullptr) {                                                                                                \
            if (int rc = rsdf_func_lds(reinterpret_cast<const void *>(pair_kernel<true, true, true, NPP>), LDS_BWD)) return rc;   \
            pair_kernel<true, true, true, NPP><<<grid, NTHR, LDS_BWD, (hipStream_t)stream>>>(a);                                  \
        } else if (a.g_masked) {                                                                                                  \
            if (int rc = rsdf_func_lds(reinterpret_cast<const void *>(pair_kernel<true, true, false, NPP>), LDS_BWD)) return rc;  \
            pair_kernel<true, true, false, NPP><<<grid, NTHR, LDS_BWD, (hipStream_t)stream>>>(a);                                 \
        } else {                                                                                                                  \
            if (int rc = rsdf_func_lds(reinterpret_cast<const void *>(pair_kernel<true, false, false, NPP>), LDS_BWD)) return rc; \
            pair_kernel<true, false, false, NPP><<<grid, NTHR, LDS_BWD, (hipStream_t)stream>>>(a);                                \
        }                                                                                                                         \
    } while (0)
    if (parts == 1) RSDF_PAIR_BWD_LAUNCH(1);
    else RSDF_PAIR_BWD_LAUNCH(2);
#undef RSDF_PAIR_BWD_LAUNCH
    RSDF_RETURN_LAUNCH();
}

int rsdf_pair_bwd(const void *x_image, int K, const float *wa, const float *ba, const float *wb, const float *bb, int64_t n,
                  const float *g, int g_masked, const float *hb_rows, const float *dz_out, const float *w_out, int N2,
                  float *dw_out, const void *bound, float *dx, int lddx, int kout, float *dx2, int ld2, int k1, int x_relu,
                  void *dx_absmax, float *dwa, float *dba, float *dwb, float *dbb, void *stream)
{
    return pair_bwd_impl(2, x_image, K, wa, ba, wb, bb, n, g, g_masked, hb_rows, dz_out, w_out, N2, dw_out, bound, dx, lddx, kout,
                            dx2, ld2, k1, x_relu, dx_absmax, dwa, dba, dwb, dbb, stream);
}

int rsdf_pair_bwd16(const void *x_image, int K, const float *wa, const float *ba, const float *wb, const float *bb, int64_t n,
                    const float *g, int g_masked, const float *hb_rows, const float *dz_out, const float *w_out, int N2,
                    float *dw_out, const void *bound, float *dx, int lddx, int kout, float *dx2, int ld2, int k1, int x_relu,
                    void *dx_absmax, float *dwa, float *dba, float *dwb, float *dbb, void *stream)
{
    return pair_bwd_impl(1, x_image, K, wa, ba, wb, bb, n, g, g_masked, hb_rows, dz_out, w_out, N2, dw_out, bound, dx, lddx, kout,
                            dx2, ld2, k1, x_relu, dx_absmax, dwa, dba, dwb, dbb, stream);
}

}  // extern "C"
