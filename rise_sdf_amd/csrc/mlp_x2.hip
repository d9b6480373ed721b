// H3 fused, "x2" form (round 4): the SDF VanillaMLP (models/network_utils.py:109-157, n_hidden_layers = 2, Softplus(100)) of the
// finite-difference stencil (models/geometry.py:229-244) with every fp32 matrix operand carried as TWO fp16 parts and
// every product evaluated as THREE v_mfma_f32_*_f16 instructions with fp32 accumulation:
//
//     v S = hi + lo,   hi = RNE_f16(v S),   lo = RNE_f16(v S - hi)        (S: a power of two per operand class)
//     a b  ~  a_lo b_hi + a_hi b_lo + a_hi b_hi                            (dropped: a_lo b_lo <= 2^-24 |a||b|)
//
// hi carries 11 significant bits, the residual |v S - hi| <= 2^-12 |v S| is rounded to 11 more: hi + lo = v S to 2^-24
// relative -- half an fp32 ulp, what one fp32 rounding of the operand would cost -- as long as lo is a normal fp16 number
// (|v S| >= 0.5), and to 2^-25 ABSOLUTE below that (fp16 subnormals: the matrix cores keep them, measured with
// tools/mfma_f16_subnormal.hip; the vector ALU's f16 conversions keep them by the default mode).  The class scales put the
// values that matter well above that floor: inputs (hash features, xyz, 1) x 2^8, weights x 2^6, activations x 100 log2(e)
// (= 144.27, round 6: the scale that leaves the Softplus no multiply after its logarithm; 2^6 before), and the
// backward's gradient images x a power of two derived per LAUNCH from max|d_sdf|, max|d h2| and the weights' norms (the
// weight-gradient accumulators live across the whole row loop, so the scale must not change inside a launch).  All scales
// but the activations' are powers of two: scaling and unscaling are exact (the activations': one fp32 rounding).
// Preconditions (overflow to inf otherwise): |input| < 255, |weight| < 1023, |activation| < 454.
//
// Against the round 1-3 form (three bf16 parts, six products; split_bf16.h): half the matrix instructions, 2 instead of
// 5.5 vector instructions per split value, 2/3 of the LDS traffic -- and the pre-split input image costs the SAME bytes as
// the fp32 planes it replaces (2 x 16 bits), so the stencil gather can write it for free (hashgrid_fd7.hip "x2").
// Accuracy (tools/accuracy_split.py, tests/test_gpu_x2.py): a layer's algorithmic error is ~1e-7 of the largest output,
// the level of an fp32 GEMM's own accumulation rounding (the six-product bf16 form: 1e-8, below it).
//
// Input image, written by rsdf_hashgrid_fwd_fd7_x2:  x2 [tile = row / 32][tap 7][part 2][column 36][32 rows] fp16, column
// 2 l + f = feature f of level l, 32..34 = xyz * xyz_scale + xyz_offset, 35 = 1, all times 2^8; the row halves of columns
// with bit 3 set are swapped (the LDS bank swizzle).  4608 contiguous bytes per tile and tap land in LDS as they are.
#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glob_void;

constexpr float K100 = 144.26950408889634f;                 // 100 log2(e)
#ifdef RSDF_X2_SH64
constexpr float LN2_100 = 0.0069314718055994531f;           // ln(2) / 100
#endif
// class scales: inputs, weights, activations.  Round 6: the activations' scale is 100 log2(e) instead of 2^6 -- with
// t = 100 log2(e) z the scaled Softplus SH softplus(z) is max(t, 0) + log2(1 + 2^-|t|) with no multiply left after the
// logarithm, and sigmoid(100 z) = 1 - 2^(-hs) needs none before the exponential: 6 instead of 7 and 2 instead of 3 vector
// instructions per value in kernels that are bound by their vector issue.  The fp16 parts are relative, so a scale that is not
// a power of two costs nothing but the exactness of the (fp32) scaling multiplications; the activations' range is
// 65504 / 144.27 = 454 instead of 1023.  -DRSDF_X2_SH64: the round-4..6 scale.
#ifdef RSDF_X2_SH64
constexpr float SX = 256.0f, SW = 64.0f, SH = 64.0f;
#else
constexpr float SX = 256.0f, SW = 64.0f, SH = K100;
#endif
constexpr float T1 = SW * SX, T2 = SW * SH;                 // accumulator scale of layer 1 / of layers 2 and 3

// words of a backward launch's guard scratch ("the backward's range guard" below)
constexpr int GUARD_MAX_DSDF = 0, GUARD_MAX_DH2C = 1, GUARD_NONZERO = 2, GUARD_WITHIN = 3, GUARD_DECISION = 4, GUARD_TICKET = 5;
constexpr int GUARD_WORDS = 8;

struct Frag2 { u32x4 h, l; };

__device__ __forceinline__ unsigned pack_f16(float a, float b)       // v_cvt_pk_f16_f32 (round to nearest even); a -> low half
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, f16x2));
}
// a - (float)hi.lo / b - (float)hi.hi in ONE instruction each (v_fma_mix_f32 reads the f16 half directly); exact: the
// difference of a value and its 11-bit rounding has at most 13 significant bits
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
// NP = 2: the fp32-equivalent form.  NP = 1: ONE part -- operands rounded once to fp16 (11 significant bits), one matrix
// instruction per product: the "16-bit MLP on MFMA" mode of BASELINE.json configs[4] for the fused SDF field (precision
// 'fp16' on the Python side); same kernels, the lo parts compile away.
template <int NP = 2>
__device__ __forceinline__ void split2_pair(float a, float b, unsigned &h, unsigned &l)
{
    h = pack_f16(a, b);
    l = NP == 2 ? pack_f16(resid_lo(a, h), resid_hi(b, h)) : 0u;
}
template <int NP = 2>
__device__ __forceinline__ Frag2 split2_frag(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7)
{
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    split2_pair<NP>(v0, v1, h0, l0);
    split2_pair<NP>(v2, v3, h1, l1);
    split2_pair<NP>(v4, v5, h2, l2);
    split2_pair<NP>(v6, v7, h3, l3);
    Frag2 f;
    f.h = u32x4{h0, h1, h2, h3};
    f.l = u32x4{l0, l1, l2, l3};
    return f;
}
template <int NP = 2>
__device__ __forceinline__ void store2(unsigned short *base, size_t part_stride_elems, size_t idx, float w)
{
    unsigned h, l;
    split2_pair<NP>(w, 0.0f, h, l);
    base[idx] = (unsigned short)(h & 0xffffu);
    if (NP == 2) base[idx + part_stride_elems] = (unsigned short)(l & 0xffffu);
}

__device__ __forceinline__ f32x16 mma32(u32x4 a, u32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma16(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// a (weights in LDS, the two parts PART_U4 16-byte units apart) x b, small terms first
template <int PART_U4, int NP = 2>
__device__ __forceinline__ f32x16 mma3(const u32x4 *__restrict__ wa, const Frag2 &b, f32x16 c)
{
    const u32x4 ah = wa[0];
    if (NP == 2) {
        const u32x4 al = wa[PART_U4];
        c = mma32(al, b.h, c);
        c = mma32(ah, b.l, c);
    }
    c = mma32(ah, b.h, c);
    return c;
}
template <int NP = 2>
__device__ __forceinline__ f32x4 mma3q(const Frag2 &a, const Frag2 &b, f32x4 c)
{
    if (NP == 2) {
        c = mma16(a.l, b.h, c);
        c = mma16(a.h, b.l, c);
    }
    c = mma16(a.h, b.h, c);
    return c;
}

// SH * Softplus(beta = 100)(C / T) from the accumulator C = T z:  SH max(z, 0) + SH ln2/100 log2(1 + 2^(-100 log2(e) |z|))
template <int LAYER>
__device__ __forceinline__ float softplus_scaled(float C)
{
    constexpr float T = LAYER == 1 ? T1 : T2;
#ifdef RSDF_X2_SH64
    const float e = __builtin_amdgcn_exp2f(fabsf(C) * (-K100 / T));
    return fmaf(max0(C), SH / T, __builtin_amdgcn_logf(1.0f + e) * (LN2_100 * SH));
#else
    const float t = C * (K100 / T);                 // 100 log2(e) z  (layer 2: K100 / T2 = 1 / SW, exact)
    return max0(t) + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(t)));
#endif
}
// sigmoid(100 z) = 1 - 2^(-100 log2(e) h) from hs = SH h, h = softplus(z)
// k sigmoid(100 z) = k - k 2^(-hs) as ONE fma after the exponential (the scale k of the gradient image folded in: a multiply
// and a subtraction less per value than k * (1 - e))
__device__ __forceinline__ float softplus_grad_times(float hs, float k)
{
#ifdef RSDF_X2_SH64
    return fmaf(-k, __builtin_amdgcn_exp2f(hs * (-K100 / SH)), k);
#else
    return fmaf(-k, __builtin_amdgcn_exp2f(-hs), k);
#endif
}
__device__ __forceinline__ float softplus_grad_scaled(float hs)
{
#ifdef RSDF_X2_SH64
    return 1.0f - __builtin_amdgcn_exp2f(hs * (-K100 / SH));
#else
    return 1.0f - __builtin_amdgcn_exp2f(-hs);
#endif
}

__device__ __forceinline__ u32x4 ld128(const unsigned char *p) { return *reinterpret_cast<const u32x4 *>(p); }
__device__ __forceinline__ void tr64(const unsigned char *p, unsigned &a, unsigned &b)
{
    const v4i16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16 *)p);
    const unsigned long long u = __builtin_bit_cast(unsigned long long, r);
    a = (unsigned)u;
    b = (unsigned)(u >> 32);
}
// two transposed reads 4 image "rows" (= 256 bytes in a column-major 64-byte-per-column image) apart -> 8 k-elements
__device__ __forceinline__ u32x4 tr128(const unsigned char *a, int second)
{
    unsigned x0, x1, y0, y1;
    tr64(a, x0, x1);
    tr64(a + second, y0, y1);
    return u32x4{x0, x1, y0, y1};
}

// the HBM image (NP parts per tap)
constexpr int X2_PART_B = 36 * 64;
template <int NP> constexpr int x2_tap_b() { return NP * X2_PART_B; }
template <int NP> constexpr int x2_tile_b() { return 7 * NP * X2_PART_B; }
struct SrcX2 {
    const unsigned char *x2;
    int64_t S, Sp;
    int n_levels, n_active;
};

// ==================================================================================================================
// forward (H = 32 NT <= 64): one 32-row tile per wave and iteration, whole MLP per tile, activations in registers
// (transposed chaining: a layer's accumulator tile, split, IS the next layer's B operand -- mlp_fused.hip), weights of
// all layers in LDS, the tile's input image per wave in LDS ([part 2][column 48][32 rows]: columns 36..47 constant zero).
// ==================================================================================================================
#ifndef RSDF_X2_FWD_WAVES
#define RSDF_X2_FWD_WAVES 12     // 168 registers: three waves per SIMD (8 waves: 7.49 against 7.23 ms per launch)
#endif
#ifndef RSDF_X2_BWD_OCC
#define RSDF_X2_BWD_OCC 2
#endif
constexpr int FWD_WAVES = RSDF_X2_FWD_WAVES;
constexpr int KS0 = 3;                 // layer-1 k-steps of 16 columns
constexpr int LDFS = 33;               // [row][32 features] transpose of the centre rows' outputs
constexpr int XF_COL_B = 64, XF_PART_B = 48 * XF_COL_B;
template <int NP> constexpr int xf_img_b() { return (NP * XF_PART_B > 32 * 33 * 4) ? NP * XF_PART_B : 32 * 33 * 4 + 128; }   // (>= the [32][33] fp32 transpose)

// H = 128: W0 + W1 alone are 88 KB of the 160: the last layer's feature rows are not staged -- the kernel writes the centre
// rows' second hidden layer (h2c) and the entry point runs one per-layer product on it (mlp_coop.hip's lean forward did
// the same).  With three bf16 parts per weight (135 KB) this width needed the cooperative register partition of
// mlp_coop.hip; with two parts every wave can again hold the whole network through LDS.
template <int H, int NP>
struct SmemF {
    static constexpr int NT = H / 32;
    static constexpr bool W2_IN_LDS = H <= 64;
    static constexpr int W0_PART = NT * KS0 * 2 * 32;            // [nt][s][hf][c]                (units: 16 bytes)
    static constexpr int W1_PART = NT * NT * 2 * 2 * 32;         // [nt][kt][s][hf][c]
    static constexpr int W2_PART = W2_IN_LDS ? 2 * NT * 2 * 2 * 32 : 0;   // [n2 tile (2)][kt][s][hf][c]
    static constexpr int W0 = 0;
    static constexpr int W1 = W0 + NP * W0_PART;
    static constexpr int W2 = W1 + NP * W1_PART;
    static constexpr int END_U4 = W2 + NP * W2_PART;
    static constexpr int B1 = 0;          // [H]   b1 * T2
    static constexpr int B2 = B1 + H;     // [64]  b2 (unscaled)
    static constexpr int W2R0 = B2 + 64;  // [H]   row 0 of W2 / SH: the taps' SDF dot on the vector ALU
    static constexpr int TAIL_F = W2R0 + H;
    static constexpr size_t SHARED_BYTES = (size_t)END_U4 * 16 + (size_t)TAIL_F * 4;
};
template <int H>
constexpr int fwd_waves() { return H <= 64 ? FWD_WAVES : 8; }       // H = 128: 88 KB of weights + 8 x 6 KB of images, ~230 registers
template <int H, int NP>
size_t fwd_lds() { return SmemF<H, NP>::SHARED_BYTES + (size_t)fwd_waves<H>() * xf_img_b<NP>(); }

// k of element j of lane half hf in k-step s of a 32-feature activation tile (the accumulator's register order)
__device__ __forceinline__ int frag_k(int s, int hf, int j) { return 16 * s + 8 * (j >> 2) + 4 * hf + (j & 3); }

template <int H, int NP>
__device__ __forceinline__ void stage_weights_fwd(unsigned char *smem, const float *__restrict__ w0, const float *__restrict__ b0,
                                                  const float *__restrict__ w1, const float *__restrict__ b1,
                                                  const float *__restrict__ w2, const float *__restrict__ b2, int K0, int N2)
{
    using S = SmemF<H, NP>;
    constexpr int NT = S::NT;
    unsigned short *e16 = reinterpret_cast<unsigned short *>(smem);
    const int NTHR = blockDim.x;
    // W0: [nt][s][hf][c][j], X column k = 16 s + 8 hf + j: hash features 0..31, xyz 32..34, bias (the 1 column) at 35
    for (int e = threadIdx.x; e < NT * KS0 * 2 * 32 * 8; e += NTHR) {
        const int j = e & 7, c = (e >> 3) & 31, hf = (e >> 8) & 1, s = (e >> 9) % KS0, nt = (e >> 9) / KS0;
        const int n = 32 * nt + c, k = 16 * s + 8 * hf + j;
        const float w = k < 32 ? (k < K0 - 3 ? w0[n * K0 + 3 + k] : 0.0f) : (k < 35 ? w0[n * K0 + (k - 32)] : (k == 35 ? b0[n] : 0.0f));
        store2<NP>(e16 + (size_t)S::W0 * 8, (size_t)S::W0_PART * 8, e, w * SW);
    }
    for (int e = threadIdx.x; e < NT * NT * 2 * 2 * 32 * 8; e += NTHR) {
        const int j = e & 7, c = (e >> 3) & 31, hf = (e >> 8) & 1, s = (e >> 9) & 1, kt = (e >> 10) % NT, nt = (e >> 10) / NT;
        const int n = 32 * nt + c, k = 32 * kt + frag_k(s, hf, j);
        store2<NP>(e16 + (size_t)S::W1 * 8, (size_t)S::W1_PART * 8, e, w1[n * H + k] * SW);
    }
    for (int e = threadIdx.x; S::W2_IN_LDS && e < 2 * NT * 2 * 2 * 32 * 8; e += NTHR) {
        const int j = e & 7, c = (e >> 3) & 31, hf = (e >> 8) & 1, s = (e >> 9) & 1, kt = (e >> 10) % NT, nt = (e >> 10) / NT;
        const int n = 32 * nt + c, k = 32 * kt + frag_k(s, hf, j);
        store2<NP>(e16 + (size_t)S::W2 * 8, (size_t)S::W2_PART * 8, e, n < N2 ? w2[n * H + k] * SW : 0.0f);
    }
    float *tail = reinterpret_cast<float *>(smem + (size_t)S::END_U4 * 16);
    for (int e = threadIdx.x; e < H; e += NTHR) {
        tail[S::B1 + e] = b1[e] * T2;
        tail[S::W2R0 + e] = w2[e] * (1.0f / SH);
    }
    for (int e = threadIdx.x; e < 64; e += NTHR) tail[S::B2 + e] = e < N2 ? b2[e] : 0.0f;
}

// 6 x 16 bytes per lane, every load contiguous: part p = bytes 1024 j + 16 lane (j = 0, 1, 2) of the part's 2304; the
// third load of a part needs 16 lanes only (columns 32..35) -- the other lanes read on into what follows (the image is
// allocated with 1 KB of slack for the very last one) and do not store
struct PreX2 { u32x4 a0, a1, a2, b0, b1, b2; };
template <int NP>
__device__ __forceinline__ void fetch_x2(PreX2 &pre, const SrcX2 &src, int64_t tile, int tap, int lane)
{
    const unsigned char *tb = src.x2 + tile * x2_tile_b<NP>() + tap * x2_tap_b<NP>() + lane * 16;
// (non-temporal loads of the image, here and in the backward's LDS-DMA: measured, no effect -- the streaming STORES are what matter)
#ifdef RSDF_X2_NT_LOAD
#define RSDF_LDX2(p) __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p))
#else
#define RSDF_LDX2(p) (*reinterpret_cast<const u32x4 *>(p))
#endif
    pre.a0 = RSDF_LDX2(tb);
    pre.a1 = RSDF_LDX2(tb + 1024);
    pre.a2 = RSDF_LDX2(tb + 2048);
    if (NP == 2) {
        pre.b0 = RSDF_LDX2(tb + X2_PART_B);
        pre.b1 = RSDF_LDX2(tb + X2_PART_B + 1024);
        pre.b2 = RSDF_LDX2(tb + X2_PART_B + 2048);
    }
#undef RSDF_LDX2
}
template <int NP>
__device__ __forceinline__ void store_x2(unsigned char *img, const PreX2 &pre, int lane)
{
    unsigned char *p = img + lane * 16;
    *reinterpret_cast<u32x4 *>(p) = pre.a0;
    *reinterpret_cast<u32x4 *>(p + 1024) = pre.a1;
    if (NP == 2) {
        *reinterpret_cast<u32x4 *>(p + XF_PART_B) = pre.b0;
        *reinterpret_cast<u32x4 *>(p + XF_PART_B + 1024) = pre.b1;
    }
    if (lane < 16) {
        *reinterpret_cast<u32x4 *>(p + 2048) = pre.a2;
        if (NP == 2) *reinterpret_cast<u32x4 *>(p + XF_PART_B + 2048) = pre.b2;
    }
}
// B fragment of layer-1 k-step s (lane = row c, k = 16 s + 8 hf + j): block rows = columns 16 s + 8 hf + q (+ 4), block
// columns = tile rows 16 (g & 1) + 4 p ..; bit 3 of the column index is hf, and such columns hold their row halves swapped
__device__ __forceinline__ int lane_tr_fwd(int lane)
{
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3, h = g >> 1;
    return (8 * h + q) * XF_COL_B + (((16 * (g & 1) + 4 * p) * 2) ^ (32 * h));
}
template <int NP>
__device__ __forceinline__ Frag2 x_frag_fwd(const unsigned char *img, int lctr, int s)
{
    const unsigned char *a = img + s * (16 * XF_COL_B) + lctr;
    Frag2 f;
    f.h = tr128(a, 4 * XF_COL_B);
    f.l = NP == 2 ? tr128(a + XF_PART_B, 4 * XF_COL_B) : u32x4{0u, 0u, 0u, 0u};
    return f;
}

template <int H, int NP>
__global__ void __launch_bounds__(64 * fwd_waves<H>(), fwd_waves<H>() / 4)
fwd_x2_kernel(const SrcX2 src, const float *__restrict__ w0, const float *__restrict__ b0, const float *__restrict__ w1,
              const float *__restrict__ b1, const float *__restrict__ w2, const float *__restrict__ b2, int N2,
              float *__restrict__ sdf7, float *__restrict__ feature, float *__restrict__ h2c, int *__restrict__ status)
{
    const int64_t n_samples = src.S;
    // range guard: an operand beyond its fp16 class range (header) becomes inf in a hi part and reaches the SDF output as
    // inf / nan (every product and the vector-ALU dot keep non-finite values non-finite); nothing in the steady state but a
    // compare and a scalar branch per tap, one atomic per offending tile (no state carried: the kernel sits at its register cap)
    const int K0 = 3 + 2 * src.n_levels;
    using S = SmemF<H, NP>;
    constexpr int NT = S::NT;
    constexpr int XF_IMG_B = xf_img_b<NP>();
    static_assert(XF_IMG_B >= 32 * LDFS * 4, "the feature transpose overlays the image");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, hf = lane >> 5;
    float *tail = reinterpret_cast<float *>(smem_b + (size_t)S::END_U4 * 16);
    unsigned char *img = smem_b + S::SHARED_BYTES + (size_t)wave * XF_IMG_B;
    float *Fs = reinterpret_cast<float *>(img);   // [32][LDFS] over the image, which is dead once the first layer has read it
    stage_weights_fwd<H, NP>(smem_b, w0, b0, w1, b1, w2, b2, K0, N2);
    for (int e = lane; e < XF_IMG_B / 4; e += 64) reinterpret_cast<unsigned *>(img)[e] = 0u;
    __syncthreads();
    const u32x4 *wl = reinterpret_cast<const u32x4 *>(smem_b);
    const float b2_0 = tail[S::B2];
    const int lctr = lane_tr_fwd(lane);

    const int64_t n_groups = src.Sp / 32;
    const int64_t g_first = (int64_t)blockIdx.x * fwd_waves<H>() + wave, g_step = (int64_t)gridDim.x * fwd_waves<H>();
    PreX2 pre;
    if (g_first < n_groups) fetch_x2<NP>(pre, src, g_first, 0, lane);
    for (int64_t g = g_first; g < n_groups; g += g_step) {
        const int64_t s0 = g * 32;
        for (int tap = 0; tap < 7; ++tap) {
            store_x2<NP>(img, pre, lane);
            {   // prefetch the next tile of this wave
                const int ntap = tap == 6 ? 0 : tap + 1;
                const int64_t ng = tap == 6 ? g + g_step : g;
                if (ng < n_groups) fetch_x2<NP>(pre, src, ng, ntap, lane);
            }
            // ---- layer 1: C = T1 z1
            f32x16 h1[NT], h2[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) h1[t][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < KS0; ++s) {
                const Frag2 xb = x_frag_fwd<NP>(img, lctr, s);
#pragma unroll
                for (int t = 0; t < NT; ++t) h1[t] = mma3<S::W0_PART, NP>(wl + S::W0 + ((t * KS0 + s) * 2 + hf) * 32 + c, xb, h1[t]);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) h1[t][r] = softplus_scaled<1>(h1[t][r]);
            // ---- layer 2: C = T2 z2 (bias pre-scaled)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b = *reinterpret_cast<const float4 *>(tail + S::B1 + 32 * t + 8 * q + 4 * hf);
                    h2[t][4 * q] = b.x, h2[t][4 * q + 1] = b.y, h2[t][4 * q + 2] = b.z, h2[t][4 * q + 3] = b.w;
                }
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const Frag2 hb = split2_frag<NP>(h1[kt][8 * s], h1[kt][8 * s + 1], h1[kt][8 * s + 2], h1[kt][8 * s + 3],
                                                 h1[kt][8 * s + 4], h1[kt][8 * s + 5], h1[kt][8 * s + 6], h1[kt][8 * s + 7]);
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        h2[t] = mma3<S::W1_PART, NP>(wl + S::W1 + (((t * NT + kt) * 2 + s) * 2 + hf) * 32 + c, hb, h2[t]);
                }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) h2[t][r] = softplus_scaled<2>(h2[t][r]);
            const int64_t s = s0 + c;
            // ---- SDF: dot(W2[0,:], h2[:,row]) on the vector ALU (W2 row pre-divided by SH), features split over the lane halves
            float acc = 0.0f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w = *reinterpret_cast<const float4 *>(tail + S::W2R0 + 32 * t + 8 * q + 4 * hf);
                    acc = fmaf(w.x, h2[t][4 * q], acc);
                    acc = fmaf(w.y, h2[t][4 * q + 1], acc);
                    acc = fmaf(w.z, h2[t][4 * q + 2], acc);
                    acc = fmaf(w.w, h2[t][4 * q + 3], acc);
                }
            acc += __shfl_xor(acc, 32, 64);
            if (__builtin_amdgcn_ballot_w64(!(fabsf(acc) < 3.0e38f)) != 0ull && lane == 0 && status != nullptr)
                atomicAdd(&status[RSDF_STATUS_X2_FWD_NONFINITE], 1);
            if (hf == 0 && s < n_samples) sdf7[(int64_t)tap * n_samples + s] = acc + b2_0;
            if (tap == 0 && (S::W2_IN_LDS ? feature != nullptr : h2c != nullptr)) {
                if (h2c != nullptr) {   // second hidden layer of the centre rows (unscaled), for the dW2 of the feature rows
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) Fs[c * LDFS + (r & 3) + 8 * (r >> 2) + 4 * hf] = h2[t][r] * (1.0f / SH);
                        for (int e = lane; e < 32 * 32; e += 64) {
                            const int r = e >> 5, cc = e & 31;
                            if (s0 + r < n_samples) h2c[(s0 + r) * H + 32 * t + cc] = Fs[r * LDFS + cc];
                        }
                    }
                }
                if constexpr (S::W2_IN_LDS) {
                // full last layer on the matrix cores: C = T2 (W2 h2 + b2)
                f32x16 o[2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[t][r] = tail[S::B2 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf] * T2;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int ss = 0; ss < 2; ++ss) {
                        const Frag2 hb = split2_frag<NP>(h2[kt][8 * ss], h2[kt][8 * ss + 1], h2[kt][8 * ss + 2], h2[kt][8 * ss + 3],
                                                     h2[kt][8 * ss + 4], h2[kt][8 * ss + 5], h2[kt][8 * ss + 6], h2[kt][8 * ss + 7]);
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            if (t * 32 < N2)
                                o[t] = mma3<S::W2_PART, NP>(wl + S::W2 + (((t * NT + kt) * 2 + ss) * 2 + hf) * 32 + c, hb, o[t]);
                    }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int nc = N2 - 32 * t < 32 ? N2 - 32 * t : 32;   // columns of this half (wave-uniform)
                    if (nc <= 0) break;
#pragma unroll
                    for (int r = 0; r < 16; ++r) Fs[c * LDFS + (r & 3) + 8 * (r >> 2) + 4 * hf] = o[t][r] * (1.0f / T2);
                    bool fbad = false;         // (h2 >= 454 overflows the split of THIS product only: the SDF dot reads fp32 h2)
                    for (int e = lane; e < 32 * nc; e += 64) {
                        const int r = e / nc, cc = e - r * nc;
                        const float v = Fs[r * LDFS + cc];
                        fbad |= !(fabsf(v) < 3.0e38f);
                        if (s0 + r < n_samples) feature[(s0 + r) * N2 + 32 * t + cc] = v;
                    }
                    if (__builtin_amdgcn_ballot_w64(fbad) != 0ull && lane == 0 && status != nullptr)
                        atomicAdd(&status[RSDF_STATUS_X2_FWD_NONFINITE], 1);
                }
                }
                // the transpose has overwritten the hi part (and the head of the lo part, which the next tile's store
                // rewrites): its zero columns 36..47 must be zero again (12 x 64 B; columns 0..35 are rewritten per tile)
                if (lane < 48) *reinterpret_cast<u32x4 *>(img + 36 * XF_COL_B + lane * 16) = u32x4{0u, 0u, 0u, 0u};
            }
        }
    }
}

// ==================================================================================================================
// backward (H = 64), the "quad" form of mlp_quad.hip: a workgroup is 4 waves, wave w owns features 16 w .. 16 w + 15 of
// every layer, every product is a v_mfma_f32_16x16x32_f16; the layers are recomputed, weight gradients accumulate in
// registers over the workgroup's whole row loop.  Images: X column-major (lands by LDS-DMA as the gather wrote it),
// H1 / dz2 / dz1 in mlp_quad.hip's chunked [8-column chunk][row ^ swizzle][8 columns] form, two fp16 parts each.
// ==================================================================================================================
constexpr int QCS = 512;               // chunk: 32 rows x 16 B
constexpr int QX_PART = 64 * 64;       // X image part: 64 columns x 64 B (columns 36..63 constant zero)

template <int NW, int NP>
struct QL {
    static constexpr int H = 16 * NW;
    static constexpr int H_PART = (H / 8) * QCS;                  // activation image part
    static constexpr int XI = 0;                                  // two X images (tile parity)
    static constexpr int H1I = XI + 2 * NP * QX_PART;
    static constexpr int DZI = H1I + NP * H_PART;                 // dz2
    static constexpr int DZ1 = DZI + NP * H_PART;                 // dz1 in an image of its own (no barrier between the last
    static constexpr int RED = DZ1 + NP * H_PART;                 //   read of dz2 and the write of dz1); 8 floats of scratch
    // round 6: the tile's per-row gradient inputs land by DMA ONE TILE AHEAD (see dma_grads): d_sdf of the next tile-tap
    // (32 floats, two buffers) and, for the centre tap, the [32][H] rows of d(h2) (one buffer: consumed six tiles before it
    // is refilled)
    static constexpr int DSD = RED + 32;                          // 2 x 128 B
    static constexpr int DH2 = DSD + 256;                         // 32 rows x H fp32, 16-byte chunks XOR-swizzled by row
    static constexpr int END = DH2 + 32 * H * 4;
};

__device__ __forceinline__ int qoff(int row, int col)
{
    const int ch = col >> 3;
    return ch * QCS + ((row ^ ((ch & 1) * 12)) << 4) + (col & 7) * 2;
}
struct LaneQ {
    int row;     // B fragment of a layer product: chunk 4 kb + g, row 16 rh + c16
    int tr0;     // transposed fragment: rows 8 g + q, columns 16 ft + 4 p ..
    int tr1;     //   rows + 4
    int st;      // store: row 16 rh + c16, columns 16 w + 4 g ..
    int xtr[2];  // X image, layer-1 B fragment, tile rows 16 rh ..: column 8 g + q, bytes (32 rh + 8 p) ^ 32 (g & 1)
    int xrow;    // X image, dW0 operand: column c16, rows 8 g .. 8 g + 7
};
__device__ __forceinline__ LaneQ lane_consts(int w, int lane)
{
    const int g = lane >> 4, c16 = lane & 15, q = c16 >> 2, p = lane & 3;
    LaneQ c;
    c.row = g * QCS + ((c16 ^ (12 * (g & 1))) << 4);
    c.tr0 = (p >> 1) * QCS + (((8 * g + q) ^ (12 * (p >> 1))) << 4) + (p & 1) * 8;
    c.tr1 = (p >> 1) * QCS + (((8 * g + 4 + q) ^ (12 * (p >> 1))) << 4) + (p & 1) * 8;
    c.st = (2 * w + (g >> 1)) * QCS + ((c16 ^ (12 * (g >> 1))) << 4) + (g & 1) * 8;
    c.xtr[0] = (8 * g + q) * 64 + ((8 * p) ^ (32 * (g & 1)));
    c.xtr[1] = (8 * g + q) * 64 + ((32 + 8 * p) ^ (32 * (g & 1)));
    c.xrow = c16 * 64 + ((16 * g) ^ (32 * ((c16 >> 3) & 1)));
    return c;
}
// B fragment of a layer product: lane (k-group g, sample row 16 rh + c16) reads features 32 kb + 8 g .. + 7
template <int PART, int NP>
__device__ __forceinline__ Frag2 rowq(const unsigned char *img, int kb, int rh, const LaneQ &c)
{
    const unsigned char *p = img + c.row + kb * (4 * QCS) + rh * 256;
    Frag2 f;
    f.h = ld128(p);
    f.l = NP == 2 ? ld128(p + PART) : u32x4{0u, 0u, 0u, 0u};
    return f;
}
// fragment whose k dimension is the tile's 32 ROWS: lane (rows 8 g .. 8 g + 7, column 16 ft + c16); A operand (A[i = column]
// [k = row]) and B operand (B[k = row][j = column]) of the weight-gradient products
template <int PART, int NP>
__device__ __forceinline__ Frag2 trfq(const unsigned char *img, int ft, const LaneQ &c)
{
    const unsigned char *a0 = img + c.tr0 + ft * (2 * QCS), *a1 = img + c.tr1 + ft * (2 * QCS);
    Frag2 f;
    unsigned x0, x1, y0, y1;
    tr64(a0, x0, x1);
    tr64(a1, y0, y1);
    f.h = u32x4{x0, x1, y0, y1};
    if (NP == 2) {
        tr64(a0 + PART, x0, x1);
        tr64(a1 + PART, y0, y1);
        f.l = u32x4{x0, x1, y0, y1};
    } else {
        f.l = u32x4{0u, 0u, 0u, 0u};
    }
    return f;
}
// this wave's 16 x 16 result (features 16 w + 4 g + r, sample row 16 rh + c16), already scaled -> split once -> image
template <int PART, int NP>
__device__ __forceinline__ void store_q(unsigned char *img, int rh, const LaneQ &c, const f32x4 &v)
{
    unsigned h0, l0, h1, l1;
    split2_pair<NP>(v[0], v[1], h0, l0);
    split2_pair<NP>(v[2], v[3], h1, l1);
    unsigned char *p = img + c.st + rh * 256;
    *reinterpret_cast<uint2 *>(p) = uint2{h0, h1};
    if (NP == 2) *reinterpret_cast<uint2 *>(p + PART) = uint2{l0, l1};
}
template <int NP>
__device__ __forceinline__ Frag2 x_col(const unsigned char *xi, int kb, int rh, const LaneQ &c)
{
    const unsigned char *a = xi + c.xtr[rh] + kb * (32 * 64);
    Frag2 f;
    f.h = tr128(a, 4 * 64);
    f.l = NP == 2 ? tr128(a + QX_PART, 4 * 64) : u32x4{0u, 0u, 0u, 0u};
    return f;
}
template <int NP>
__device__ __forceinline__ Frag2 x_rows(const unsigned char *xi, int ct, const LaneQ &c)
{
    const unsigned char *a = xi + c.xrow + ct * (16 * 64);
    Frag2 f;
    f.h = ld128(a);
    f.l = NP == 2 ? ld128(a + QX_PART) : u32x4{0u, 0u, 0u, 0u};
    return f;
}
__device__ __forceinline__ void wait_vm0() { __builtin_amdgcn_s_waitcnt(0x0F70); }      // s_waitcnt vmcnt(0)
__device__ __forceinline__ void lds_barrier()                                            // see mlp_coop.hip lds_barrier()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);      // s_waitcnt lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// 6 linear DMA instructions per tile (1024 + 1024 + 256 bytes per part): i = 3 part + block; with four waves wave w issues
// i = w and, for w < 2, i = w + 4; with eight, waves 0..5 one each (ws: the wave index as a SCALAR: scalar branches)
__device__ __forceinline__ void dma_one(unsigned char *img, const unsigned char *tb, int i, int lane)
{
    const int part = i / 3, b = i - 3 * part;
    const unsigned char *gp = tb + part * X2_PART_B + b * 1024 + lane * 16;
    unsigned char *dst = img + part * QX_PART + b * 1024;
#ifdef RSDF_X2_NT_DMA
    constexpr int AUX = 2;     // cache policy nt (the SLC position of the aux operand): the image is read once
#else
    constexpr int AUX = 0;
#endif
    if (b < 2) __builtin_amdgcn_global_load_lds((glob_void *)gp, (lds_void *)dst, 16, 0, AUX);
    else if (lane < 16) __builtin_amdgcn_global_load_lds((glob_void *)gp, (lds_void *)dst, 16, 0, AUX);
}
template <int NW, int NP>
__device__ __forceinline__ void dma_tile(unsigned char *img, const SrcX2 &src, int64_t tile, int tap, int ws, int lane)
{
    const unsigned char *tb = src.x2 + tile * x2_tile_b<NP>() + tap * x2_tap_b<NP>();
    if (ws < 3 * NP) dma_one(img, tb, ws, lane);
    if (NW == 4 && NP == 2 && ws < 2) dma_one(img, tb, ws + 4, lane);
}
// The per-row gradient inputs of tile (group, tap) by DMA into LDS: d_sdf7[tap][32 rows] (wave 0, 32 lanes x 4 bytes) and, for
// tap 0, dh2c[32 rows][H] (1 KB per instruction: 1024 / (4 H) rows each; chunk c of row r at slot c ^ (r & (H / 4 - 1)) so that
// the 16 rows a lane group reads for one chunk cover all banks).  Rows past the end are clamped to the last sample, as the
// per-lane loads did; their values are masked by row_ok at the point of use.
#ifndef RSDF_X2_GRAD_DMA_H128
#define RSDF_X2_GRAD_DMA_H128 0
#endif
template <int NW, bool WITH_DH2 = true>
__device__ __forceinline__ void dma_grads(unsigned char *dsd, unsigned char *dh2, const float *__restrict__ d_sdf7,
                                          const float *__restrict__ dh2c, int64_t S, int64_t s0, int tap, int ws, int lane)
{
    constexpr int H = 16 * NW, CPR = H / 4;                 // 16-byte chunks per row
    if (ws == 0 && lane < 32) {
        int64_t row = s0 + lane;
        row = row < S ? row : S - 1;
        __builtin_amdgcn_global_load_lds((glob_void *)(d_sdf7 + (int64_t)tap * S + row), (lds_void *)dsd, 4, 0, 0);
    }
    if (WITH_DH2 && tap == 0 && dh2c != nullptr) {
        constexpr int RPI = 64 / CPR;                       // rows per 1 KB instruction: 4 (H = 64) / 2 (H = 128)
        constexpr int N_INSTR = 32 / RPI;                   // 8 / 16
#pragma unroll
        for (int k = 0; k < N_INSTR / NW; ++k) {
            const int i = ws + NW * k;
            const int r = RPI * i + lane / CPR, slot = lane % CPR;
            int64_t row = s0 + r;
            row = row < S ? row : S - 1;
            __builtin_amdgcn_global_load_lds((glob_void *)(dh2c + row * H + 4 * (slot ^ (r & (CPR - 1)))),
                                             (lds_void *)(dh2 + i * 1024), 16, 0, 0);
        }
    }
}

// d_planes (896 B per sample, 13-17 GB per launch) are written once here and read once by the hash backward's producer, a
// whole launch later: a streaming (non-temporal) store keeps them out of the caches' way (22.0 -> 21.6 ms per launch;
// -DRSDF_X2_PLAIN_DPLANES for A/B)
__device__ __forceinline__ void st_dplane(float *p, float a, float b)
{
#ifndef RSDF_X2_PLAIN_DPLANES
    __builtin_nontemporal_store(((unsigned long long)__float_as_uint(b) << 32) | (unsigned long long)__float_as_uint(a),
                                reinterpret_cast<unsigned long long *>(p));
#else
    *reinterpret_cast<float2 *>(p) = float2{a, b};
#endif
}
// 2^e with |v| 2^e < 2^14 for every |v| <= bound (bound = 0, inf or nan: 1)
__device__ __forceinline__ float grad_scale(float bound)
{
    if (!(bound > 0.0f) || !(bound < 3.0e38f)) return 1.0f;
    int e = 13 - ilogbf(bound);
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    return ldexpf(1.0f, e);
}

// NW = 4 (H = 64): two workgroups per CU.  NW = 8 (H = 128): one workgroup of eight waves per CU; the four 16 x 16
// sub-tiles of d(hash features) go to waves 0..3.
template <int NW, int NP>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? RSDF_X2_BWD_OCC : 1)
bwd_x2_kernel(const SrcX2 src, const float *__restrict__ w0, const float *__restrict__ b0, const float *__restrict__ w1,
              const float *__restrict__ b1, const float *__restrict__ w2, const float *__restrict__ d_sdf7,
              const float *__restrict__ dh2c, const unsigned *__restrict__ absmax /* the guard words: bits of max|d_sdf7|, max|dh2c|, .., decision */,
              float *__restrict__ d_planes, float *__restrict__ dw0, float *__restrict__ db0, float *__restrict__ dw1,
              float *__restrict__ db1, float *__restrict__ dw2, float *__restrict__ db2)
{
    using L = QL<NW, NP>;
    constexpr int H = L::H, KB = H / 32, HP = L::H_PART, NTHR = 64 * NW;
    if (absmax[GUARD_DECISION] != 0u) return;      // this launch runs on the range-free kernels (spread_kernel below)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *s_red = reinterpret_cast<float *>(smem + L::RED);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
    const int ws = __builtin_amdgcn_readfirstlane(w);
    const int K0 = 3 + 2 * src.n_levels;
    const int fw = 16 * w + c16;                   // the feature this lane addresses in an A fragment of its wave
    const LaneQ lc = lane_consts(w, lane);

    for (int e = threadIdx.x; e < L::RED / 4; e += NTHR) reinterpret_cast<unsigned *>(smem)[e] = 0u;
    // ---- gradient-image scales of this launch: |dz2| <= max|W2 row 0| max|d_sdf| + max|dh2c|, |dz1| <= max_k sum_n |W1[n][k]| |dz2|
    float m2 = 0.0f, cs = 0.0f;
    if (threadIdx.x < H) {
        m2 = fabsf(w2[threadIdx.x]);
        for (int n = 0; n < H; ++n) cs += fabsf(w1[(size_t)n * H + threadIdx.x]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m2 = fmaxf(m2, __shfl_xor(m2, o, 64));
        cs = fmaxf(cs, __shfl_xor(cs, o, 64));
    }
    if (lane == 0 && w < 2) { s_red[2 * w] = m2; s_red[2 * w + 1] = cs; }      // (H <= 128: the first two waves)
    __syncthreads();
    const float m2a = H > 64 ? fmaxf(s_red[0], s_red[2]) : s_red[0], csa = H > 64 ? fmaxf(s_red[1], s_red[3]) : s_red[1];
    const float bound2 = m2a * __uint_as_float(absmax[0]) + __uint_as_float(absmax[1]);
    const float G2 = grad_scale(bound2), G1 = grad_scale(csa * bound2);

    // ---- weight fragments (A operands: lane = (row c16 of the wave's 16-row block, k-group g), 8 consecutive k), x SW
    Frag2 w1f[KB], w1t[KB], w0f[2], w0t[KB];
    const int mt = w & 1, rhx = (w >> 1) & 1;      // this wave's d(hash features) sub-tile: columns 16 mt.., rows 16 rhx.. (waves 0..3)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        float v[8];
        const float *p = w1 + (size_t)fw * H + 32 * kb + 8 * g;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[j] * SW;
        w1f[kb] = split2_frag<NP>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);                                                   // W1[fw][k]
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = w1[(size_t)(32 * kb + 8 * g + j) * H + fw] * SW;
        w1t[kb] = split2_frag<NP>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);                                                   // W1[n][fw]
        const int col = 16 * mt + c16;                                              // hash column of the dx sub-tile
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = col < 2 * src.n_levels ? w0[(size_t)(32 * kb + 8 * g + j) * K0 + 3 + col] * SW : 0.0f;
        w0t[kb] = split2_frag<NP>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);                                                   // W0[n][3 + col]
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
            v[j] = x * SW;
        }
        w0f[kb] = split2_frag<NP>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    }
    f32x4 b1r, w2r;                                // bias of layer 2 (x T2) / row 0 of W2 for features 16 w + 4 g + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        b1r[r] = b1[16 * w + 4 * g + r] * T2;
        w2r[r] = w2[16 * w + 4 * g + r];
    }
    __syncthreads();

    f32x4 gw1[H / 16], gw0[3], gw2p = {0.f, 0.f, 0.f, 0.f}, gb1p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < H / 16; ++n) gw1[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 3; ++n) gw0[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float gb2 = 0.0f;
    const float k_dz1 = G1 / (SW * G2), k_dx = 1.0f / (SW * G1);

    const int64_t n_groups = src.Sp / 32;
    // (H = 64 only: measured 22.07 -> 21.54 ms per launch; at H = 128 the eight-wave kernel has no register to spare -- seven
    // spills, 52.1 -> 52.4 ms -- and keeps the per-lane loads.  -DRSDF_X2_NO_GRAD_DMA for A/B.)
#ifndef RSDF_X2_NO_GRAD_DMA
    constexpr bool GRAD_DMA = NW == 4 || RSDF_X2_GRAD_DMA_H128;     // d_sdf of the next tile-tap
    constexpr bool GRAD_DMA_DH2 = NW == 4;                          // ... and the centre tap's d(h2) rows
#else
    constexpr bool GRAD_DMA = false, GRAD_DMA_DH2 = false;
#endif
    if ((int64_t)blockIdx.x < n_groups) {
        dma_tile<NW, NP>(smem + L::XI, src, (int64_t)blockIdx.x, 0, ws, lane);
        if (GRAD_DMA)
            dma_grads<NW, GRAD_DMA_DH2>(smem + L::DSD, smem + L::DH2, d_sdf7, dh2c, src.S, (int64_t)blockIdx.x * 32, 0, ws, lane);
    }
    // ---- layer 1 backward, input side: d(hash features) sub-tile (16 columns x 16 rows, all features) of the tile whose dz1
    // image is in LDS.  DEFERRED by one tile (round 6, -DRSDF_X2_NO_DEFER_DX for A/B): it runs right after barrier (1) of the
    // NEXT tile, which (a) removes the fourth barrier of a tile -- dz1's only cross-wave reader is this product -- and (b)
    // gives the d_planes stores a whole tile to retire before the next s_waitcnt vmcnt(0) (they used to be issued just before
    // it, younger than the next tile's DMA only by accident of placement; stores count in vmcnt like loads, in issue order).
    auto emit_dx = [&](int64_t ps0, int ptap) {
        if (NW == 4 || ws < 4) {
            f32x4 dx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) dx = mma3q<NP>(w0t[kb], rowq<HP, NP>(smem + L::DZ1, kb, rhx, lc), dx);
            // result rows = hash columns 16 mt + 4 g + r = (level 8 mt + 2 g + (r >> 1), feature r & 1); lane column =
            // sample row 16 rhx + c16: two float2 stores, 16 lanes cover 128 contiguous bytes of a level plane
            const int64_t row = ps0 + 16 * rhx + c16;
            if (d_planes != nullptr && row < src.S) {
                const int lev = 8 * mt + 2 * g;
                if (lev < src.n_active)
                    st_dplane(d_planes + (((int64_t)lev * 7 + ptap) * src.S + row) * 2, dx[0] * k_dx, dx[1] * k_dx);
                if (lev + 1 < src.n_active)
                    st_dplane(d_planes + (((int64_t)(lev + 1) * 7 + ptap) * src.S + row) * 2, dx[2] * k_dx, dx[3] * k_dx);
            }
        }
    };
    int parity = 0;
    for (int64_t gi = blockIdx.x; gi < n_groups; gi += gridDim.x) {
        const int64_t s0 = gi * 32;
        for (int tap = 0; tap < 7; ++tap) {
            const unsigned char *xi = smem + L::XI + parity * NP * QX_PART;
            wait_vm0();                            // this wave's share of the tile has landed (and the previous tile's stores retired)
            // every wave's share has landed once all have passed their wait; the OTHER image is free once all have finished
            // the previous tile's dW0 reads; the previous tile's dz1 image is complete: one barrier serves all three
            lds_barrier();                                                       // (1)
            parity ^= 1;
#ifndef RSDF_X2_NO_DEFER_DX
            if (tap > 0) emit_dx(s0, tap - 1);
            else if (gi != (int64_t)blockIdx.x) emit_dx(s0 - (int64_t)gridDim.x * 32, 6);
#endif
            bool row_ok[2];
            float dsdf_raw[2];
            f32x4 dz[2];
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                const int64_t row = s0 + 16 * rh + c16;
                row_ok[rh] = row < src.S;
                const int64_t rowc = row_ok[rh] ? row : src.S - 1;
                // (GRAD_DMA: after the flip ``parity`` names the NEXT tile's buffers: this tile's d_sdf is in the other one)
                dsdf_raw[rh] = GRAD_DMA ? reinterpret_cast<const float *>(smem + L::DSD + (parity ^ 1) * 128)[16 * rh + c16]
                                        : d_sdf7[(int64_t)tap * src.S + rowc];
                if (tap == 0 && dh2c != nullptr) {     // (uniform) centre taps: d(h2) through the feature rows
                    const int r = 16 * rh + c16;
                    const float4 v = GRAD_DMA_DH2 ? *reinterpret_cast<const float4 *>(smem + L::DH2 + r * (H * 4) +
                                                                                  (((4 * w + g) ^ (r & (H / 4 - 1))) << 4))
                                              : *reinterpret_cast<const float4 *>(dh2c + rowc * H + 16 * w + 4 * g);
                    dz[rh] = f32x4{v.x, v.y, v.z, v.w};
                } else {
                    dz[rh] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            {
                const int ntap = tap == 6 ? 0 : tap + 1;
                const int64_t ng = tap == 6 ? gi + gridDim.x : gi;
                if (ng < n_groups) {
                    dma_tile<NW, NP>(smem + L::XI + parity * NP * QX_PART, src, ng, ntap, ws, lane);
                    if (GRAD_DMA)
                        dma_grads<NW, GRAD_DMA_DH2>(smem + L::DSD + parity * 128, smem + L::DH2, d_sdf7, dh2c, src.S, ng * 32, ntap,
                                                    ws, lane);
                }
            }
            // ---- recompute layer 1 (C = T1 z1) -> SH h1
            // (round 6, -DRSDF_X2_SERIAL_RH for A/B: the matrix products of BOTH row halves are issued before the vector work of
            // either -- written as "product, activation, store" per half the compiler keeps that order, and the second half's
            // dependent MFMA chain then waits behind the first half's exp2 / log2 instead of running under it)
            f32x4 h1[2], h2[2];
#ifndef RSDF_X2_SERIAL_RH
            {
                f32x4 acc[2];
#pragma unroll
                for (int rh = 0; rh < 2; ++rh) {
                    acc[rh] = f32x4{0.f, 0.f, 0.f, 0.f};
                    acc[rh] = mma3q<NP>(w0f[0], x_col<NP>(xi, 0, rh, lc), acc[rh]);
                    acc[rh] = mma3q<NP>(w0f[1], x_col<NP>(xi, 1, rh, lc), acc[rh]);
                }
#pragma unroll
                for (int rh = 0; rh < 2; ++rh) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) h1[rh][r] = softplus_scaled<1>(acc[rh][r]);
                    store_q<HP, NP>(smem + L::H1I, rh, lc, h1[rh]);
                }
            }
#else
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = mma3q<NP>(w0f[0], x_col<NP>(xi, 0, rh, lc), acc);
                acc = mma3q<NP>(w0f[1], x_col<NP>(xi, 1, rh, lc), acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) h1[rh][r] = softplus_scaled<1>(acc[r]);
                store_q<HP, NP>(smem + L::H1I, rh, lc, h1[rh]);
            }
#endif
            lds_barrier();                                                       // (2) H1 image complete
            // ---- recompute layer 2 (C = T2 z2), then layer 3 backward: dz2 = (W2[0,:] d_sdf + feature part) sigma'(z2)
            {
                f32x4 acc2[2];
#ifndef RSDF_X2_SERIAL_RH
#pragma unroll
                for (int rh = 0; rh < 2; ++rh) {
                    acc2[rh] = b1r;
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb) acc2[rh] = mma3q<NP>(w1f[kb], rowq<HP, NP>(smem + L::H1I, kb, rh, lc), acc2[rh]);
                }
#endif
#pragma unroll
                for (int rh = 0; rh < 2; ++rh) {
#ifdef RSDF_X2_SERIAL_RH
                    acc2[rh] = b1r;
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb) acc2[rh] = mma3q<NP>(w1f[kb], rowq<HP, NP>(smem + L::H1I, kb, rh, lc), acc2[rh]);
#endif
                    const f32x4 acc = acc2[rh];
                    const float dsdf = row_ok[rh] ? dsdf_raw[rh] : 0.0f;
                    if (w == 0 && g == 0) gb2 += dsdf;
                    f32x4 dzs;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        h2[rh][r] = softplus_scaled<2>(acc[r]);
                        gw2p[r] = fmaf(dsdf, h2[rh][r], gw2p[r]);                    // (x SH: unscaled at the flush)
#ifdef RSDF_X2_GRAD_R5
                        dz[rh][r] = row_ok[rh] ? fmaf(w2r[r], dsdf, dz[rh][r]) * softplus_grad_scaled(h2[rh][r]) : 0.0f;
                        gb1p[r] += dz[rh][r] * G2;
                        dzs[r] = dz[rh][r] * G2;
#else
                        // (the image's scale G2 rides in the derivative's fma; the bias gradient sums the scaled values and is
                        // unscaled at the flush: G2 is a power of two)
                        dzs[r] = row_ok[rh] ? fmaf(w2r[r], dsdf, dz[rh][r]) * softplus_grad_times(h2[rh][r], G2) : 0.0f;
                        gb1p[r] += dzs[r];
#endif
                    }
                    store_q<HP, NP>(smem + L::DZI, rh, lc, dzs);
                }
            }
            lds_barrier();                                                       // (3) dz2 image complete
            // ---- layer 2 backward: G1 dz1[own k1] = (W1^T dz2) sigma'(z1) ; dW1[own n][all k] += dz2^T h1 (K = the 32 rows)
            {
                f32x4 acc3[2];
#pragma unroll
                for (int rh = 0; rh < 2; ++rh) {
                    acc3[rh] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb) acc3[rh] = mma3q<NP>(w1t[kb], rowq<HP, NP>(smem + L::DZI, kb, rh, lc), acc3[rh]);
                }
#pragma unroll
                for (int rh = 0; rh < 2; ++rh)
#pragma unroll
#ifdef RSDF_X2_GRAD_R5
                    for (int r = 0; r < 4; ++r) dz[rh][r] = acc3[rh][r] * k_dz1 * softplus_grad_scaled(h1[rh][r]);
#else
                    for (int r = 0; r < 4; ++r) dz[rh][r] = acc3[rh][r] * softplus_grad_times(h1[rh][r], k_dz1);
#endif
            }
            {
                const Frag2 a = trfq<HP, NP>(smem + L::DZI, w, lc);
#pragma unroll
                for (int n = 0; n < H / 16; ++n) gw1[n] = mma3q<NP>(a, trfq<HP, NP>(smem + L::H1I, n, lc), gw1[n]);   // x G2 SH
            }
            store_q<HP, NP>(smem + L::DZ1, 0, lc, dz[0]);
            store_q<HP, NP>(smem + L::DZ1, 1, lc, dz[1]);
#ifdef RSDF_X2_NO_DEFER_DX
            lds_barrier();                                                       // (4) dz1 image complete
            emit_dx(s0, tap);
#else
            // no barrier (4): the dW0 product below reads only this wave's OWN slab of the dz1 image (columns 16 w ..: what
            // store_q just wrote; LDS operations of one wave execute in order, the wait makes it explicit); the product
            // that needs every wave's slab runs after the next tile's barrier (1) (emit_dx above)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xC07F);      // s_waitcnt lgkmcnt(0)
            asm volatile("" ::: "memory");
#endif
            // ---- layer 1 backward, weight side: dW0 += dz1^T X
            {
                const Frag2 a = trfq<HP, NP>(smem + L::DZ1, w, lc);
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) gw0[ct] = mma3q<NP>(a, x_rows<NP>(xi, ct, lc), gw0[ct]);           // x G1 SX
            }
            // no barrier: the next tile's barrier (1) separates these reads from the DMA that overwrites this X image, and
            // its H1 / dz2 / dz1 writes sit behind its barriers (1) .. (3)
        }
    }

#ifndef RSDF_X2_NO_DEFER_DX
    if ((int64_t)blockIdx.x < n_groups) {          // the last tile's input gradient
        lds_barrier();
        const int64_t last_gi = (int64_t)blockIdx.x + ((n_groups - 1 - (int64_t)blockIdx.x) / gridDim.x) * gridDim.x;
        emit_dx(last_gi * 32, 6);
    }
#endif
    // ---- flush: gw1[n][r] = G2 SH dW1[16 w + 4 g + r][16 n + c16]; gw0[ct][r] = G1 SX dW0 image [feature][X column 16 ct + c16]
    const float u1 = 1.0f / (G2 * SH), u0 = 1.0f / (G1 * SX);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int f = 16 * w + 4 * g + r;
#pragma unroll
        for (int n = 0; n < H / 16; ++n) atomicAdd(&dw1[(size_t)f * H + 16 * n + c16], gw1[n][r] * u1);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
            if (16 * ct + c16 < 2 * src.n_levels) atomicAdd(&dw0[(size_t)f * K0 + 3 + 16 * ct + c16], gw0[ct][r] * u0);
        if (c16 < 3) atomicAdd(&dw0[(size_t)f * K0 + c16], gw0[2][r] * u0);
        if (c16 == 3) atomicAdd(&db0[f], gw0[2][r] * u0);
        float a = gw2p[r], b = gb1p[r];            // per-lane partials -> sum over the 16 sample columns
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            a += __shfl_xor(a, o, 64);
            b += __shfl_xor(b, o, 64);
        }
        if (c16 == 0) {
            atomicAdd(&dw2[f], a * (1.0f / SH));
            atomicAdd(&db1[f], b * (1.0f / G2));      // (gb1p sums G2 dz2)
        }
    }
    if (w == 0) {
        gb2 = wave_sum(gb2);
        if (lane == 0) atomicAdd(&db2[0], gb2);
    }
}

// max |v| over n floats -> *out (bits of a non-negative float: unsigned order == float order); *out zeroed by the caller
__global__ void __launch_bounds__(256)
absmax_kernel(const float *__restrict__ v, int64_t n, unsigned *__restrict__ out)
{
    float m = 0.0f;
    const int64_t n4 = n / 4;
    const float4 *v4 = reinterpret_cast<const float4 *>(v);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 t = v4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(t.x), fabsf(t.y))), fmaxf(fabsf(t.z), fabsf(t.w)));
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(v[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(out, __float_as_uint(m));
}

// ---- the backward's range guard -------------------------------------------------------------------------------------
// The gradient images share ONE power-of-two scale per launch (grad_scale of the launch bound): a row keeps the full two-part
// precision down to 2^-15 of the bound, 11 bits down to 2^-28, nothing below 2^-38 -- an ABSOLUTE floor of bound 2^-38 where
// the reference's fp32 network (models/network_utils.py:109-157) keeps 24 bits of every row.  That is harmless while the
// rows that carry the gradient sit within ~2^20 of the largest one (every configuration measured: the spread inside a chunk
// is 2^10 .. 2^20) and wrong when a few outliers set the bound -- one saturated sample behind render_weight.cu:139-151's
// 1 / max(1 - alpha, 1e-10) would flush the d(hash features) of every ordinary row.  The guard measures it per launch, on the
// device: spread_kernel counts the non-zero rows of d_sdf7t and those within 2^-SPREAD of the launch bound; when fewer than
// 1 / BULK_INV of the non-zero rows are (the maximum is an outlier relative to the bulk), the launch's decision word is set,
// bwd_x2_kernel returns at once and the same launch runs on the range-free round-3 kernels (three bf16 parts: fp32's
// exponent range) from planes rebuilt out of the x2 image -- no host read, two idle launches in the steady state.
#ifndef RSDF_X2_SPREAD_LOG2
#define RSDF_X2_SPREAD_LOG2 20
#endif
#ifndef RSDF_X2_BULK_INV
#define RSDF_X2_BULK_INV 1024u
#endif

// mode 1: decide from the counts; mode 2: reroute unconditionally (tests, RSDF_X2_REROUTE=force)
__global__ void __launch_bounds__(256)
spread_kernel(const float *__restrict__ v, int64_t n, const float *__restrict__ w2, int H, int spread_log2, int mode,
              unsigned *__restrict__ guard, int *__restrict__ status)
{
    __shared__ float s_m2[4];
    __shared__ unsigned s_cnt[8];
    float m2 = 0.0f;
    for (int e = threadIdx.x; e < H; e += 256) m2 = fmaxf(m2, fabsf(w2[e]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m2 = fmaxf(m2, __shfl_xor(m2, o, 64));
    if ((threadIdx.x & 63) == 0) s_m2[threadIdx.x >> 6] = m2;
    __syncthreads();
    m2 = fmaxf(fmaxf(s_m2[0], s_m2[1]), fmaxf(s_m2[2], s_m2[3]));
    // a tap row's own bound is m2 |d_sdf|; the launch bound is m2 max|d_sdf| + max|dh2c| (bwd_x2_kernel's bound2)
    const float bound = m2 * __uint_as_float(guard[GUARD_MAX_DSDF]) + __uint_as_float(guard[GUARD_MAX_DH2C]);
    const float thr = (bound > 0.0f && bound < 3.0e38f && m2 > 0.0f) ? ldexpf(bound, -spread_log2) / m2 : 0.0f;
    unsigned nz = 0, in = 0;
    const int64_t n4 = n / 4;
    const float4 *v4 = reinterpret_cast<const float4 *>(v);
    // the decision is a FRACTION of the rows: large launches are sampled -- one 16 KB block of every eight (whole cache lines:
    // a strided 16 bytes of every 128 would still move every line), 1.4e7 of a bench chunk's 1.1e8 rows; the maximum above is exact
    const bool sampled = n4 > (int64_t)1 << 22;
    const int64_t work = sampled ? (n4 >> 13) << 10 : n4;        // float4 groups visited
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < work; j += (int64_t)gridDim.x * 256) {
        const int64_t i = sampled ? ((j >> 10) << 13) + (j & 1023) : j;
        const float4 t = v4[i];
        const float a = fabsf(t.x), b = fabsf(t.y), c = fabsf(t.z), d = fabsf(t.w);
        nz += (a > 0.0f) + (b > 0.0f) + (c > 0.0f) + (d > 0.0f);
        in += (a > 0.0f && a >= thr) + (b > 0.0f && b >= thr) + (c > 0.0f && c >= thr) + (d > 0.0f && d >= thr);
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n && !sampled; i += (int64_t)gridDim.x * 256) {
        const float a = fabsf(v[i]);
        nz += a > 0.0f;
        in += a > 0.0f && a >= thr;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        nz += __shfl_xor(nz, o, 64);
        in += __shfl_xor(in, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { s_cnt[threadIdx.x >> 6] = nz; s_cnt[4 + (threadIdx.x >> 6)] = in; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&guard[GUARD_NONZERO], s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]);
        atomicAdd(&guard[GUARD_WITHIN], s_cnt[4] + s_cnt[5] + s_cnt[6] + s_cnt[7]);
        __threadfence();
        if (atomicAdd(&guard[GUARD_TICKET], 1u) == gridDim.x - 1) {          // the last workgroup decides for the launch
            const unsigned tnz = atomicAdd(&guard[GUARD_NONZERO], 0u), tin = atomicAdd(&guard[GUARD_WITHIN], 0u);
            const bool reroute = mode == 2 || (tnz > 0u && (unsigned long long)tin * RSDF_X2_BULK_INV < (unsigned long long)tnz);
            if (reroute) {
                atomicExch(&guard[GUARD_DECISION], 1u);
                if (status != nullptr) atomicAdd(&status[RSDF_STATUS_X2_BWD_REROUTED], 1);
            }
            if (status != nullptr) atomicAdd(&status[RSDF_STATUS_X2_BWD_GUARDED], 1);
        }
    }
}

// the rerouted launch's inputs: planes [L][7][S][2] and x7t [7][S][3] (already scaled: the round-3 kernels get xyz_scale 1,
// offset 0) back out of the image, (hi + lo) / 2^8 -- the value the fp32 gather would have written, to 2^-24.  One workgroup
// per tile and iteration; runs only when the launch's decision word is set.
__global__ void __launch_bounds__(256)
x2_to_planes_kernel(const SrcX2 src, const unsigned *__restrict__ run_if, float *__restrict__ planes, float *__restrict__ x7t)
{
    if (*run_if == 0u) return;
    const int64_t n_tiles = src.Sp / 32;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const unsigned char *tb = src.x2 + tile * x2_tile_b<2>();
        for (int e = threadIdx.x; e < 7 * 18 * 32; e += 256) {
            const int r = e & 31, cp = (e >> 5) % 18, tap = (e >> 5) / 18;      // column pair cp = columns 2 cp, 2 cp + 1
            const int64_t s = tile * 32 + r;
            if (s >= src.S) continue;
            float val[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int col = 2 * cp + f;
                const int rr = r ^ (((col >> 3) & 1) * 16);
                const unsigned char *p = tb + tap * x2_tap_b<2>() + col * 64 + rr * 2;
                const float hi = (float)*reinterpret_cast<const _Float16 *>(p);
                const float lo = (float)*reinterpret_cast<const _Float16 *>(p + X2_PART_B);
                val[f] = (hi + lo) * (1.0f / SX);
            }
            if (cp < 16) {
                if (cp < src.n_levels)
                    *reinterpret_cast<float2 *>(planes + (((int64_t)cp * 7 + tap) * src.S + s) * 2) = float2{val[0], val[1]};
            } else if (cp == 16) {
                x7t[((int64_t)tap * src.S + s) * 3 + 0] = val[0];
                x7t[((int64_t)tap * src.S + s) * 3 + 1] = val[1];
            } else {
                x7t[((int64_t)tap * src.S + s) * 3 + 2] = val[0];
            }
        }
    }
}

}  // namespace

// the range-free route (mlp_quad.hip, mlp_coop.hip)
__attribute__((visibility("hidden"))) int rsdf_coop_bwd(int NT, const float *x7t, const float *planes, int n_levels, int n_active, float xyz_scale,
                  float xyz_offset, const float *w0, const float *b0, const float *w1, const float *b1, const float *w2,
                  int64_t n_samples, const float *d_sdf7t, const float *dh2c, float *d_planes, float *dw0, float *db0,
                  float *dw1, float *db1, float *dw2, float *db2, hipStream_t st, const unsigned *run_if);
__attribute__((visibility("hidden"))) int rsdf_quad_bwd(int H, const float *x7t, const float *planes, int n_levels, int n_active, float xyz_scale, float xyz_offset,
                  const float *w0, const float *b0, const float *w1, const float *b1, const float *w2, int64_t n_samples,
                  const float *d_sdf7t, const float *dh2c, float *d_planes, float *dw0, float *db0, float *dw1, float *db1,
                  float *dw2, float *db2, hipStream_t st, const unsigned *run_if);

extern "C" {

int rsdf_sdfmlp_fd7_x2_supported(int K0, int H, int N2)
{
    return (K0 >= 5 && K0 <= 35 && (K0 - 3) % 2 == 0 && (H == 32 || H == 64 || H == 128) && N2 >= 1 && N2 <= 64) ? 1 : 0;
}

int rsdf_sdfmlp_fd7_fwd_x2(const void *x2, int parts, int n_levels, int H, int N2, const float *w0, const float *b0, const float *w1,
                           const float *b1, const float *w2, const float *b2, int64_t n_samples, float *sdf7t, float *feature,
                           float *h2c, int *status, void *stream)
{
    const int K0 = 3 + 2 * n_levels;
    RSDF_CHECK_ARG(n_levels >= 1 && n_levels <= 16, "sdfmlp_fd7_fwd_x2: n_levels must be in [1,16]");
    RSDF_CHECK_ARG(h2c == nullptr || feature != nullptr, "sdfmlp_fd7_fwd_x2: h2c needs feature");
    RSDF_CHECK_ARG(rsdf_sdfmlp_fd7_x2_supported(K0, H, N2), "sdfmlp_fd7_fwd_x2: unsupported layer sizes");
    RSDF_CHECK_ARG(parts == 1 || parts == 2, "sdfmlp_fd7_fwd_x2: parts must be 1 or 2");
    if (n_samples <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int64_t Sp = (n_samples + 31) / 32 * 32;
    const SrcX2 src{reinterpret_cast<const unsigned char *>(x2), n_samples, Sp, n_levels, n_levels};
    auto grid_of = [&](int waves) {
        const int64_t want = (Sp / 32 + waves - 1) / waves;
        return (unsigned)(want < 512 ? want : 512);
    };
#define RSDF_X2_FWD(HH, NPP)                                                                                                  \
    do {                                                                                                                      \
        if (int rc = rsdf_func_lds(reinterpret_cast<const void *>(fwd_x2_kernel<HH, NPP>), fwd_lds<HH, NPP>())) return rc;     \
        fwd_x2_kernel<HH, NPP><<<grid_of(fwd_waves<HH>()), 64 * fwd_waves<HH>(), fwd_lds<HH, NPP>(), st>>>(                    \
            src, w0, b0, w1, b1, w2, b2, N2, sdf7t, feature, h2c, status);                                                    \
    } while (0)
    if (H == 128) {
        // the feature rows of the last layer are one per-layer product on the centre rows' h2 (see SmemF)
        RSDF_CHECK_ARG(feature == nullptr || h2c != nullptr, "sdfmlp_fd7_fwd_x2: at H = 128 the feature output needs h2c");
        if (parts == 2) RSDF_X2_FWD(128, 2); else RSDF_X2_FWD(128, 1);
        if (hipGetLastError() != hipSuccess) { rsdf_set_error("sdfmlp_fd7_fwd_x2: launch failed"); return RSDF_EINVAL; }
        if (feature != nullptr) return rsdf_linear_fwd(h2c, H, w2, b2, n_samples, H, N2, RSDF_ACT_NONE, feature, N2, stream);
        return 0;
    }
    if (H == 64) {
        if (parts == 2) RSDF_X2_FWD(64, 2); else RSDF_X2_FWD(64, 1);
    } else {
        if (parts == 2) RSDF_X2_FWD(32, 2); else RSDF_X2_FWD(32, 1);
    }
#undef RSDF_X2_FWD
    RSDF_RETURN_LAUNCH();
}

int rsdf_sdfmlp_fd7_bwd_x2(const void *x2, int parts, int n_levels, int n_active_levels, int H, int N2, const float *w0, const float *b0,
                           const float *w1, const float *b1, const float *w2, const float *b2, int64_t n_samples,
                           const float *d_sdf7t, const float *d_feature, float *dh2c_scratch, void *guard_scratch,
                           float *x7t_scratch, int reroute, float *d_planes, float *dw0, float *db0, float *dw1, float *db1,
                           float *dw2, float *db2, int *status, void *stream)
{
    const int K0 = 3 + 2 * n_levels;
    (void)b2;
    RSDF_CHECK_ARG(n_levels >= 1 && n_levels <= 16, "sdfmlp_fd7_bwd_x2: n_levels must be in [1,16]");
    RSDF_CHECK_ARG((H == 64 || H == 128) && rsdf_sdfmlp_fd7_x2_supported(K0, H, N2), "sdfmlp_fd7_bwd_x2: unsupported layer sizes (H must be 64 or 128)");
    RSDF_CHECK_ARG(guard_scratch != nullptr, "sdfmlp_fd7_bwd_x2: the 32-byte guard scratch is required");
    RSDF_CHECK_ARG(parts == 1 || parts == 2, "sdfmlp_fd7_bwd_x2: parts must be 1 or 2");
    RSDF_CHECK_ARG(reroute >= 0 && reroute <= 2, "sdfmlp_fd7_bwd_x2: reroute must be 0 (never), 1 (guarded) or 2 (always)");
    RSDF_CHECK_ARG(reroute == 0 || (parts == 2 && x7t_scratch != nullptr && d_planes != nullptr),
                   "sdfmlp_fd7_bwd_x2: the range-free route needs parts = 2, d_planes and the x7t scratch");
    if (n_samples <= 0) return 0;
    if (n_active_levels < 0 || n_active_levels > n_levels) n_active_levels = n_levels;
    hipStream_t st = (hipStream_t)stream;
    unsigned *am = reinterpret_cast<unsigned *>(guard_scratch);
    (void)hipMemsetAsync(am, 0, GUARD_WORDS * 4, st);
    absmax_kernel<<<1024, 256, 0, st>>>(d_sdf7t, 7 * n_samples, am);
    if (d_feature != nullptr) {
        RSDF_CHECK_ARG(dh2c_scratch != nullptr, "sdfmlp_fd7_bwd_x2: d_feature needs the [n, H] dh2c scratch");
        const int rc = rsdf_linear_bwd_input(d_feature, nullptr, N2, w2, n_samples, H, N2, RSDF_ACT_NONE, 0, H, nullptr,
                                             dh2c_scratch, H, stream);
        if (rc) return rc;
        absmax_kernel<<<1024, 256, 0, st>>>(dh2c_scratch, (int64_t)H * n_samples, am + 1);
    }
    const int64_t Sp = (n_samples + 31) / 32 * 32;
    const SrcX2 src{reinterpret_cast<const unsigned char *>(x2), n_samples, Sp, n_levels, n_active_levels};
    const int64_t groups = Sp / 32;
    const float *dh = d_feature != nullptr ? dh2c_scratch : nullptr;
    // the range guard: only the fp32-labelled form with a table gradient has a parity contract the launch scale can break,
    // and the range-free route needs somewhere to rebuild its inputs (planes in place of d_planes, x7t in the scratch)
    const int mode = reroute;
    if (mode != 0)
        spread_kernel<<<1024, 256, 0, st>>>(d_sdf7t, 7 * n_samples, w2, H, RSDF_X2_SPREAD_LOG2, mode, am, status);
#define RSDF_X2_BWD(NWW, NPP, GRID)                                                                                           \
    do {                                                                                                                      \
        if (int rc = rsdf_func_lds(reinterpret_cast<const void *>(bwd_x2_kernel<NWW, NPP>), QL<NWW, NPP>::END)) return rc;     \
        bwd_x2_kernel<NWW, NPP><<<(unsigned)(GRID), 64 * NWW, QL<NWW, NPP>::END, st>>>(                                        \
            src, w0, b0, w1, b1, w2, d_sdf7t, dh, am, d_planes, dw0, db0, dw1, db1, dw2, db2);                                \
    } while (0)
    const int64_t max_wgs = 256 * RSDF_X2_BWD_OCC;                       // H = 64: RSDF_X2_BWD_OCC workgroups per CU
    const int64_t g4 = groups < max_wgs ? groups : max_wgs, g8 = groups < 256 ? groups : 256;
    if (H == 64) {
        if (parts == 2) RSDF_X2_BWD(4, 2, g4); else RSDF_X2_BWD(4, 1, g4);
    } else {
        if (parts == 2) RSDF_X2_BWD(8, 2, g8); else RSDF_X2_BWD(8, 1, g8);
    }
#undef RSDF_X2_BWD
    if (mode != 0) {
        if (hipGetLastError() != hipSuccess) { rsdf_set_error("sdfmlp_fd7_bwd_x2: launch failed"); return RSDF_EINVAL; }
        // (both return at once unless the decision word is set; the round-3 kernels read the planes of a (tile, tap) before they
        // write its d_planes and no other workgroup touches that (tile, tap): in place)
        x2_to_planes_kernel<<<(unsigned)(groups < 2048 ? groups : 2048), 256, 0, st>>>(src, am + GUARD_DECISION, d_planes, x7t_scratch);
        if (H == 64)
            return rsdf_quad_bwd(64, x7t_scratch, d_planes, n_levels, n_active_levels, 1.0f, 0.0f, w0, b0, w1, b1, w2, n_samples,
                                 d_sdf7t, dh, d_planes, dw0, db0, dw1, db1, dw2, db2, st, am + GUARD_DECISION);
        return rsdf_coop_bwd(4, x7t_scratch, d_planes, n_levels, n_active_levels, 1.0f, 0.0f, w0, b0, w1, b1, w2, n_samples,
                             d_sdf7t, dh, d_planes, dw0, db0, dw1, db1, dw2, db2, st, am + GUARD_DECISION);
    }
    RSDF_RETURN_LAUNCH();
}

}  // extern "C"
