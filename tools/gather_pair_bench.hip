// Is a 16-byte gather of an aligned entry pair as cheap as ONE 8-byte gather?  (hash-grid corner pairs along x)
// Random entries of a 4 MiB table (one hashed level), 64 independent gathers in flight per lane.
//   mode 0: two float2 loads (idx, idx ^ 1)      mode 1: one float4 load (idx & ~1)      mode 2: one float2 load
// Build: hipcc -O2 --offload-arch=gfx950 tools/gather_pair_bench.hip -o tools/_build/gather_pair_bench
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float2 *__restrict__ t, int mode, unsigned mask, int iters, float *out)
{
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            const unsigned idx = (s >> 8) & mask;
            if (mode == 0) {
                const float2 a = t[idx], b = t[idx ^ 1u];
                acc += a.x + a.y + b.x + b.y;
            } else if (mode == 1) {
                const float4 v = *reinterpret_cast<const float4 *>(t + (idx & ~1u));
                acc += v.x + v.y + v.z + v.w;
            } else {
                const float2 a = t[idx];
                acc += a.x + a.y;
            }
        }
    }
    if (acc == 123.456f) out[0] = acc;
}
int main()
{
    const unsigned n = 1u << 19;
    float2 *t; float *o;
    hipMalloc(&t, n * sizeof(float2)); hipMalloc(&o, 4);
    hipMemset(t, 0, n * sizeof(float2));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            k<<<4096, 256>>>(t, mode, n - 1, 64, o);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        const double lanes = 4096.0 * 256 * 64 * 8;
        printf("mode %d: %.3f ms  %.3e lane-gathers/s\n", mode, ms, lanes / (ms * 1e-3));
    }
    return 0;
}
