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
