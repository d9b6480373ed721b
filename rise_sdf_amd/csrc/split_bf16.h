// Split-bf16 matrix products: fp32-equivalent products on v_mfma_f32_32x32x16_bf16 (see mlp_fused.hip,
// "Split-bf16 matrix products", and DESIGN.md 3.5).  Shared by the fused SDF-MLP kernels and the per-layer kernels.
#pragma once
#include "common.h"

// Two builds of the MLP translation units share this header (Makefile):
//   default      fp32-equivalent: every operand is the exact sum of three bf16 parts, six partial products;
//   -DRSDF_BF16  config[4]'s "bf16 MLP on MFMA" (BASELINE.json configs[4]; SURVEY H3 "c4 optional bf16"): operands
//                rounded ONCE to bf16 (v_cvt_pk_bf16_f32, round to nearest even), ONE v_mfma_f32_*_bf16 product per
//                k-step, fp32 accumulation, fp32 master weights and fp32 weight-gradient accumulators.  The middle and
//                low parts are compile-time zeros there: their conversions, LDS traffic and five of the six matrix
//                instructions disappear.  Entry points of that build carry the suffix _bf16 (RSDF_P), so both
//                precisions live in one library and the choice is per call (opt-in; the f32 headline never uses it).
#ifdef RSDF_BF16
#define RSDF_P(name) name##_bf16
#else
#define RSDF_P(name) name
#endif

namespace {

#ifdef RSDF_BF16
constexpr bool RSDF_SPLIT3 = false;
#else
constexpr bool RSDF_SPLIT3 = true;
#endif

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

struct Frag3 { u32x4 h, m, l; };

__device__ __forceinline__ unsigned pack_bf16(float a, float b)  // v_cvt_pk_bf16_f32 (RNE); a -> low half
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

__device__ __forceinline__ void split3_pair(float a, float b, unsigned &h, unsigned &m, unsigned &l)
{
    h = pack_bf16(a, b);
    if (!RSDF_SPLIT3) {
        m = l = 0u;
        return;
    }
    const float ra = a - bf16_lo(h), rb = b - bf16_hi(h);   // exact
    m = pack_bf16(ra, rb);
    l = pack_bf16(ra - bf16_lo(m), rb - bf16_hi(m));
}
// 8 consecutive fragment elements -> 3-part fragment
__device__ __forceinline__ Frag3 split_frag(float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                            float v7)
{
    unsigned h[4], m[4], l[4];
    split3_pair(v0, v1, h[0], m[0], l[0]);
    split3_pair(v2, v3, h[1], m[1], l[1]);
    split3_pair(v4, v5, h[2], m[2], l[2]);
    split3_pair(v6, v7, h[3], m[3], l[3]);
    Frag3 f;
    f.h = u32x4{h[0], h[1], h[2], h[3]};
    f.m = u32x4{m[0], m[1], m[2], m[3]};
    f.l = u32x4{l[0], l[1], l[2], l[3]};
    return f;
}
__device__ __forceinline__ f32x16 mma_bf16(u32x4 a, u32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c,
                                                    0, 0, 0);
}
// a (weights: LDS, three parts PART_STRIDE bytes apart) x b (activations), six partial products
template <int PART_STRIDE_U4>
__device__ __forceinline__ f32x16 mma6(const u32x4 *__restrict__ wa, const Frag3 &b, f32x16 c)
{
    if (!RSDF_SPLIT3) return mma_bf16(wa[0], b.h, c);
    const u32x4 ah = wa[0], am = wa[PART_STRIDE_U4], al = wa[2 * PART_STRIDE_U4];
    c = mma_bf16(al, b.h, c);
    c = mma_bf16(ah, b.l, c);
    c = mma_bf16(am, b.m, c);
    c = mma_bf16(am, b.h, c);
    c = mma_bf16(ah, b.m, c);
    c = mma_bf16(ah, b.h, c);
    return c;
}

__device__ __forceinline__ void store3(unsigned short *base, size_t part_stride_elems, size_t idx, float w)
{
    unsigned h, m, l;
    split3_pair(w, 0.0f, h, m, l);
    base[idx] = (unsigned short)(h & 0xffffu);
    if (!RSDF_SPLIT3) return;
    base[idx + part_stride_elems] = (unsigned short)(m & 0xffffu);
    base[idx + 2 * part_stride_elems] = (unsigned short)(l & 0xffffu);
}
// six partial products with the weight fragment in registers
__device__ __forceinline__ f32x16 mma6r(const Frag3 &a, const Frag3 &b, f32x16 c)
{
    if (!RSDF_SPLIT3) return mma_bf16(a.h, b.h, c);
    c = mma_bf16(a.l, b.h, c);
    c = mma_bf16(a.h, b.l, c);
    c = mma_bf16(a.m, b.m, c);
    c = mma_bf16(a.m, b.h, c);
    c = mma_bf16(a.h, b.m, c);
    c = mma_bf16(a.h, b.h, c);
    return c;
}

}  // namespace
