// Shared host/device helpers for librisesdf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/risesdf_hip.h"

#define RSDF_WAVE 64

// library-internal (hidden: not part of the C ABI in include/risesdf_hip.h)
extern "C" __attribute__((visibility("hidden"))) void rsdf_set_error(const char *msg);

#define RSDF_CHECK_ARG(cond, msg)        \
    do {                                 \
        if (!(cond)) {                   \
            rsdf_set_error(msg);         \
            return RSDF_EINVAL;          \
        }                                \
    } while (0)

// Launch-error check only: never synchronises (hipGetLastError is host-side state).
#define RSDF_RETURN_LAUNCH()                              \
    do {                                                  \
        hipError_t e_ = hipGetLastError();                \
        if (e_ != hipSuccess) {                           \
            rsdf_set_error(hipGetErrorString(e_));        \
            return (int)e_;                               \
        }                                                 \
        return 0;                                         \
    } while (0)

static inline unsigned rsdf_blocks(int64_t n, int threads) { return (unsigned)((n + threads - 1) / threads); }

// hipFuncAttributeMaxDynamicSharedMemorySize for kernels that want more than 64 KB of LDS.  The attribute is per DEVICE
// (and autograd runs backward on another host thread), so the "already set" cache is keyed on (kernel, device, bytes) and
// thread-local; returns 0 or the hipError_t (rsdf_last_error() set).  Library-internal: not exported.
__attribute__((visibility("hidden"))) int rsdf_func_lds(const void *kernel, size_t bytes);
// getenv() once per process for the A/B knobs (DESIGN.md "Knobs"): the value cannot change under a running library
__attribute__((visibility("hidden"))) bool rsdf_env_is(const char *name, const char *value);

// ---- wave64 primitives -------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// max(x, 0) in ONE instruction.  fmaxf() is llvm.maxnum, which under the IEEE mode bit quiets a possibly-signalling input
// first: `v_max_f32 x, x, x` in front of every `v_max_f32 0, x` -- 64 of the ~520 vector instructions of an x2 forward tile.
// As a signed-integer maximum of the bit pattern: negative floats (and -0) are negative integers -> +0, positive ones are
// unchanged.  Same value as fmaxf(x, 0) for every non-NaN input; a NaN with a clear sign bit stays a NaN (torch's relu
// propagates it too; fmaxf returned 0).  (Not inline assembly: the compiler does not know the wait states between a matrix
// instruction's result and an instruction it cannot see into, and the first version of this read accumulators too early.)
__device__ __forceinline__ float max0(float x)
{
#ifdef RSDF_NO_MAX0      // A/B: the two-instruction form
    return fmaxf(x, 0.0f);
#else
    const int i = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, i > 0 ? i : 0);
#endif
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// inclusive scans across the 64 lanes (Hillis-Steele over ds_bpermute shuffles)
__device__ __forceinline__ float wave_incl_prod(float v)
{
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float u = __shfl_up(v, o, 64);
        if (l >= o) v *= u;
    }
    return v;
}
__device__ __forceinline__ float wave_incl_sum(float v)
{
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float u = __shfl_up(v, o, 64);
        if (l >= o) v += u;
    }
    return v;
}
__device__ __forceinline__ int wave_incl_sum_i(int v)
{
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int u = __shfl_up(v, o, 64);
        if (l >= o) v += u;
    }
    return v;
}
// inclusive SUFFIX sum: lane l gets sum over lanes >= l
__device__ __forceinline__ float wave_suffix_sum(float v)
{
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float u = __shfl_down(v, o, 64);
        if (l + o < 64) v += u;
    }
    return v;
}
