// Issue-to-issue cost of DEPENDENT bf16 MFMAs on gfx950: C chains of v_mfma_f32_32x32x16_bf16 (or 16x16x32), each
// instruction accumulating into the previous result of its chain.  Prints cycles per MFMA for C = 1, 2, 4.
// Build: hipcc -O2 --offload-arch=gfx950 tools/mfma_chain.hip -o tools/_build/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int C, int SHAPE>
__global__ void __launch_bounds__(512) chain(long long *out, int iters)
{
    f32x16 a[4] = {{0}, {0}, {0}, {0}};
    f32x4 b[4] = {{0}, {0}, {0}, {0}};
    bf16x8 p, q;
    for (int k = 0; k < 8; ++k) p[k] = (__bf16)(1.0f + threadIdx.x * 1e-3f), q[k] = (__bf16)0.5f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (SHAPE == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a[u % C]) : "v"(p), "v"(q));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(b[u % C]) : "v"(p), "v"(q));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int c = 0; c < 4; ++c) { for (int k = 0; k < 16; ++k) s += a[c][k]; for (int k = 0; k < 4; ++k) s += b[c][k]; }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (s == 123.456f) out[1] = (long long)s;
}

template <int C, int SHAPE>
void run(long long *d, int threads)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        chain<C, SHAPE><<<256, threads>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    long long cyc;
    hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
    printf("%s, %d chain(s), %d wave(s)/SIMD: %.1f ns, %.1f clk per MFMA per wave\n", SHAPE ? "16x16x32" : "32x32x16", C,
           threads / 256, ms * 1e6 / (iters * 8.0), (double)cyc / (iters * 8.0));
}

int main()
{
    long long *d;
    hipMalloc(&d, 16);
    run<1, 0>(d, 256); run<2, 0>(d, 256); run<4, 0>(d, 256); run<1, 0>(d, 512); run<2, 0>(d, 512);
    run<1, 1>(d, 256); run<2, 1>(d, 256); run<4, 1>(d, 256); run<1, 1>(d, 512); run<2, 1>(d, 512); run<4, 1>(d, 512);
    return 0;
}
