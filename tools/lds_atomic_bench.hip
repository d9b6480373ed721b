// Micro-benchmark: LDS atomic throughput on gfx950 for random addresses (run on the GPU box).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_bench.hip -o /tmp/lds_bench && /tmp/lds_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

constexpr int ENTRIES = 16384;
constexpr int ITERS = 512;

template <int MODE>
__global__ void __launch_bounds__(1024) k(float *out, int seed)
{
    extern __shared__ float s[];
    for (int i = threadIdx.x; i < ENTRIES * 2; i += 1024) s[i] = 0.f;
    __syncthreads();
    uint32_t x = (threadIdx.x + 1) * 2654435761u + blockIdx.x * 97u + seed;
    float acc = 0.f;
    for (int it = 0; it < ITERS; ++it) {
        x = x * 1664525u + 1013904223u;
        const uint32_t e = (x >> 10) & (ENTRIES - 1);
        const float v = (float)(x & 255);
        if (MODE == 0) { atomicAdd(&s[2 * e], v); atomicAdd(&s[2 * e + 1], v); }                 // 2x ds_add_f32
        if (MODE == 1) { atomicAdd((unsigned *)&s[2 * e], (unsigned)v); atomicAdd((unsigned *)&s[2 * e + 1], (unsigned)v); }  // 2x ds_add_u32
        if (MODE == 2) { atomicAdd((unsigned long long *)&s[2 * e], (unsigned long long)v); }      // 1x ds_add_u64
        if (MODE == 3) { acc += (float)atomicAdd((unsigned *)&s[2 * e], 1u); }                       // ds_add_rtn_u32 random
        if (MODE == 4) { acc += (float)atomicAdd((unsigned *)&s[(e & 31)], 1u); }                    // rtn, 32 hot addresses
        if (MODE == 5) { float2 *p = (float2 *)&s[2 * e]; float2 t = *p; t.x += v; t.y += v; *p = t; }  // racy plain RMW b64
        if (MODE == 6) { acc += __hip_atomic_fetch_add(&s[2 * e], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } // rtn f32
        if (MODE == 7) { atomicAdd((double *)&s[2 * e], (double)v); }                                 // ds_add_f64
    }
    __syncthreads();
    float t = acc;
    for (int i = threadIdx.x; i < ENTRIES * 2; i += 1024) t += s[i];
    if (t == 12345.678f) out[0] = t;
}

template <int MODE>
void run(const char *name)
{
    float *out;
    hipMalloc(&out, 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, ENTRIES * 8);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256, 1024, ENTRIES * 8>>>(out, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 4; ++r) k<MODE><<<256, 1024, ENTRIES * 8>>>(out, r);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double ops = 4.0 * 256 * 1024 * ITERS;
    printf("%-34s %8.3f ms  %7.1f G lane-iter/s chip  (%.2f per clk per CU @2.1GHz)\n", name, ms, ops / ms / 1e6,
           ops / ms / 1e6 * 1e9 / 256 / 2.1e9 / 1e9 * 1e0);
    hipFree(out);
}

int main()
{
    run<0>("2x ds_add_f32 random");
    run<1>("2x ds_add_u32 random");
    run<2>("1x ds_add_u64 random");
    run<3>("1x ds_add_rtn_u32 random");
    run<4>("1x ds_add_rtn_u32 32 hot addrs");
    run<5>("plain b64 read+write (racy)");
    run<6>("1x ds_add_rtn_f32 random");
    run<7>("1x ds_add_f64 random");
    return 0;
}
