// Exhaustive check of hashgrid_common.h's unit_div (five multiply-adds) against the IEEE division it replaces: all 2^32 bit
// patterns of x for each divisor of a list (2 r of the yamls' radius 1.5 first), the quotients compared bit for bit.
// (models/geometry.py:229-244's contraction (q - (-r)) / (r - (-r)), which neus.hip fd_points_kernel performs as a division.)
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I rise_sdf_amd/csrc -I include -o /tmp/unit_div_check tools/unit_div_check.hip
//   /tmp/unit_div_check
// Prints, per divisor: mismatches among the x the fast path is specified for (x == +0 or 2^-100 < |x| < 2^100) and among the
// rest (-0, tiny, huge, non-finite: documented as out of contract).  Exit code 1 if any in-contract quotient differs.
// Round 6, MI355X: 0 in-contract mismatches for all 18 divisors (gpurun_out -> profiles/r06d_final/unit_div_check.txt).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "hashgrid_common.h"

namespace {

__global__ void sweep_kernel(UnitDiv u, unsigned long long *bad_in, unsigned long long *bad_out, unsigned *first_bad)
{
    const uint64_t n_threads = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long in = 0, out = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += n_threads) {
        const uint32_t bits = (uint32_t)i;
        const float x = __uint_as_float(bits);
        const float want = x / u.d;
        const float got = unit_div(x, u);
        const bool same = __float_as_uint(want) == __float_as_uint(got) || (want != want && got != got);
        if (!same) {
            const float ax = fabsf(x);
            if (bits == 0x80000000u) {      // -0 / d = -0, the multiply-adds give +0: documented, same cell and weight
                ++out;
            } else if (x == 0.0f || (ax > 0x1p-100f && ax < 0x1p100f)) {
                ++in;
                atomicMin(first_bad, bits);
            } else {
                ++out;
            }
        }
    }
    if (in) atomicAdd(bad_in, in);
    if (out) atomicAdd(bad_out, out);
}

}  // namespace

int main()
{
    const float radii[] = {1.5f, 1.0f, 0.5f, 0.7f, 1.1f, 2.3f, 3.7f, 0.123f, 10.0f, 1.9999999f, 1.0000001f, 0.33333334f, 1e-3f, 1e3f,
                           0.999999f, 1.25f, 6.0f, 47.11f};
    unsigned long long *d_in, *d_out;
    unsigned *d_first;
    hipMalloc(&d_in, 8);
    hipMalloc(&d_out, 8);
    hipMalloc(&d_first, 4);
    int rc = 0;
    for (float r : radii) {
        const float two_r = r - (-r);
        const UnitDiv u{two_r, 1.0f / two_r};      // (as tap_src() in hashgrid_fd7.hip)
        hipMemset(d_in, 0, 8);
        hipMemset(d_out, 0, 8);
        hipMemset(d_first, 0xff, 4);
        sweep_kernel<<<4096, 256>>>(u, d_in, d_out, d_first);
        unsigned long long in = 0, out = 0;
        unsigned first = 0;
        hipMemcpy(&in, d_in, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&out, d_out, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost);
        std::printf("radius %.9g (2 r = %.9g, y = %.9g): in-contract mismatches %llu", r, two_r, u.y, in);
        if (in) std::printf(" (first x bits 0x%08x)", first);
        std::printf(", out-of-contract mismatches %llu of 2^32\n", out);
        if (in) rc = 1;
    }
    return rc;
}
