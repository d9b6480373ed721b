// Device helpers shared by the hash-grid kernels (hashgrid.hip, hashgrid_fd7.hip).
#pragma once
#include "common.h"

namespace {

struct LevelInfo {
    float scale;
    uint32_t res, offset, size;
    bool dense, pow2;
};

__device__ __forceinline__ LevelInfo level_info(const rsdf_grid_meta &m, int l)
{
    LevelInfo li;
    li.scale = m.scale[l];
    li.res = m.res[l];
    li.offset = m.offset[l];
    li.size = m.size[l];
    li.dense = (uint64_t)li.res * li.res * li.res <= (uint64_t)li.size;
    li.pow2 = (li.size & (li.size - 1u)) == 0u;
    return li;
}

__device__ __forceinline__ uint32_t grid_index(uint32_t x, uint32_t y, uint32_t z, const LevelInfo &li)
{
    // idx % size without the 32-bit division where it can be avoided (8 of them per point and level were half of the
    // generic gather's vector instructions): same value in every case
    if (li.dense) {
        // corners of points in [0,1] give idx < size (1 + 1/res + 1/res^2): one conditional subtraction; anything else
        // (out-of-range input) takes the division
        uint32_t idx = x + y * li.res + z * li.res * li.res;
        if (idx >= li.size) {
            idx -= li.size;
            if (idx >= li.size) idx %= li.size;
        }
        return idx;
    }
    const uint32_t idx = (x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u);
    return li.pow2 ? (idx & (li.size - 1u)) : idx % li.size;   // hashed levels hold 2^log2_hashmap_size entries
}


struct CellFrac {
    uint32_t c[3];
    float w[3];
};

// x / d for a divisor d that is the same for a whole launch (the AABB contraction's 2 r: models/geometry.py:229-244 through
// neus.hip fd_points_kernel), CORRECTLY ROUNDED -- the same bits as the IEEE division those kernels perform -- in five
// multiply-adds instead of the compiler's ~11-instruction division (v_div_scale x 2, v_rcp, five fma, v_div_fmas,
// v_div_fixup; nine of them per (sample, level)).  With y = RN(1 / d) (computed on the host, one IEEE division per launch):
// q0 = RN(x y) is within 2 ulp of x / d; r0 = x - q0 d is exact (fma); q1 = RN(q0 + r0 y) is a faithful rounding; r1 = x - q1 d
// exact; q2 = RN(q1 + r1 y) is the correctly rounded quotient (Markstein's theorem: a faithful q and a y within half an ulp
// of 1 / d).  Valid where nothing over- or underflows: 2^-100 < |x|, d < 2^100 or x == 0 (the entry points reject other
// radii; x = p + r with p inside or near the box).  Known differences: x = -0 gives +0 (the same cell and weight); a
// non-finite x gives NaN instead of +-inf (both index garbage).  tools/unit_div_check.hip sweeps all 2^32 x for 18 divisors
// against the IEEE division: no other in-contract difference.  -DRSDF_IEEE_UNIT_DIV: the division itself (A/B: gather
// 12.39 -> 12.30 ms, hash backward 21.9 -> 21.6 per launch; as a launch-uniform BRANCH between the two forms it was slower
// than either -- nine branches per thread cut the gather's loads out of the schedule).
struct UnitDiv {
    float d, y;
};
constexpr float UNIT_DIV_MIN = 0x1p-100f, UNIT_DIV_MAX = 0x1p100f;
__device__ __forceinline__ float unit_div(float x, const UnitDiv &u)
{
#ifdef RSDF_IEEE_UNIT_DIV
    return x / u.d;
#else
    const float q0 = x * u.y;
    const float r0 = fmaf(-q0, u.d, x);
    const float q1 = fmaf(r0, u.y, q0);
    const float r1 = fmaf(-q1, u.d, x);
    return fmaf(r1, u.y, q1);
#endif
}

// pos = fmaf(scale, x, 0.5); cell = floor(pos); w = pos - cell   (identical to the oracle)
__device__ __forceinline__ CellFrac cell_frac(float px, float py, float pz, float scale)
{
    CellFrac r;
    const float p[3] = {fmaf(scale, px, 0.5f), fmaf(scale, py, 0.5f), fmaf(scale, pz, 0.5f)};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float f = floorf(p[d]);
        r.c[d] = (uint32_t)(int32_t)f;
        r.w[d] = p[d] - f;
    }
    return r;
}

__device__ __forceinline__ float corner_weight(const CellFrac &cf, int c)
{
    float w = 1.0f;
    w *= (c & 1) ? cf.w[0] : 1.0f - cf.w[0];
    w *= (c & 2) ? cf.w[1] : 1.0f - cf.w[1];
    w *= (c & 4) ? cf.w[2] : 1.0f - cf.w[2];
    return w;
}

}  // namespace
