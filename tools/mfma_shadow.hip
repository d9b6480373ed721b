// How many vector-ALU instructions fit in the shadow of one bf16 MFMA on gfx950?  Exact instruction streams (inline asm):
// per loop step 4 x { v_mfma_f32_32x32x16_bf16 (or 2 x 16x16x32); K x v_fma_f32 }, two independent accumulator chains.
// Prints s_memtime cycles per MFMA for K = 0..12, with one and with two waves per SIMD, and for v_exp_f32 fillers.
// Build: hipcc -O2 --offload-arch=gfx950 tools/mfma_shadow.hip -o tools/_build/mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int K, int EXP, int SHAPE>
__global__ void __launch_bounds__(512) shadow(long long *out, int iters)
{
    f32x16 a0 = {0}, a1 = {0};
    f32x4 b0 = {0}, b1 = {0};
    bf16x8 p, q;
    for (int k = 0; k < 8; ++k) p[k] = (__bf16)(1.0f + threadIdx.x * 1e-3f), q[k] = (__bf16)0.5f;
    float v[12];
    for (int k = 0; k < 12; ++k) v[k] = threadIdx.x * 1e-3f + k;
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    f32x2 w[12], cc = {0.999f, 0.998f};
    for (int k = 0; k < 12; ++k) w[k] = f32x2{v[k], v[k] + 1.0f};
    const float c = 0.999f, d = 0.25f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (SHAPE == 2) {
                // no matrix instruction: raw vector issue rate
            } else if (SHAPE == 0) {
                if (u & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a1) : "v"(p), "v"(q));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a0) : "v"(p), "v"(q));
            } else {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(b0) : "v"(p), "v"(q));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(b1) : "v"(p), "v"(q));
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (EXP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[k]) : "v"(cc));
                else if (EXP == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(w[k]) : "v"(cc));
                else if (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[k]));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(c), "v"(d));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int k = 0; k < 16; ++k) s += a0[k] + a1[k];
    for (int k = 0; k < 4; ++k) s += b0[k] + b1[k];
    for (int k = 0; k < 12; ++k) s += v[k] + w[k][0] + w[k][1];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (s == 123.456f) out[1] = (long long)s;
}

template <int K, int EXP, int SHAPE>
void run(long long *d, int threads)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        shadow<K, EXP, SHAPE><<<256, threads>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    long long cyc;
    hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
    printf("  K=%2d: %6.1f ns, %6.1f clk per 32x32-MFMA-time slot (per wave)\n", K, ms * 1e6 / (iters * 4.0), (double)cyc / (iters * 4.0));
}

template <int EXP, int SHAPE>
void sweep(long long *d, int threads)
{
    const char *fill[4] = {"v_fma_f32", "v_exp_f32", "v_pk_add_f32", "v_pk_fma_f32"};
    printf("%s, filler %s, %d waves per SIMD:\n", SHAPE == 2 ? "no MFMA" : SHAPE ? "2 x 16x16x32 bf16" : "32x32x16 bf16", fill[EXP], threads / 256);
    run<0, EXP, SHAPE>(d, threads); run<2, EXP, SHAPE>(d, threads); run<4, EXP, SHAPE>(d, threads); run<6, EXP, SHAPE>(d, threads);
    run<8, EXP, SHAPE>(d, threads); run<12, EXP, SHAPE>(d, threads);
}

int main()
{
    long long *d;
    hipMalloc(&d, 16);
    sweep<0, 0>(d, 256); sweep<0, 0>(d, 512); sweep<1, 0>(d, 256); sweep<0, 1>(d, 256); sweep<0, 1>(d, 512);
    sweep<2, 0>(d, 256); sweep<2, 0>(d, 512); sweep<2, 1>(d, 512); sweep<3, 0>(d, 256);
    sweep<0, 2>(d, 256); sweep<0, 2>(d, 512); sweep<2, 2>(d, 256); sweep<2, 2>(d, 512); sweep<1, 2>(d, 512); sweep<1, 0>(d, 512); sweep<1, 1>(d, 512);
    return 0;
}
