// Does fp32 MFMA on one SIMD overlap with another wave's VALU work on the same SIMD?  (gfx950)
// 512-thread workgroups: waves w and w+4 share SIMD (tools/wave_simd_map.hip).  Modes:
//   0: every wave: N MFMA                       1: every wave: M VALU ops
//   2: waves 0-3 N MFMA, waves 4-7 M VALU       3: every wave: N MFMA then M VALU (bursts, in phase)
//   4: like 3 but waves 4-7 start with the VALU burst (out of phase)
//   5: every wave: MFMA and VALU finely interleaved (1 MFMA : M/N VALU)
// Prints wall microseconds per mode; shape s = 0: 32x32x2, 1: 16x16x4.
// Build: hipcc -O2 --offload-arch=gfx950 tools/mfma_valu_overlap.hip -o tools/_build/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int SHAPE>
__device__ __forceinline__ void mfma_burst(f32x16 &a0, f32x16 &a1, f32x4 &b0, f32x4 &b1, float x, float y, int n)
{
    for (int i = 0; i < n; i += 2) {
        if (SHAPE == 0) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        } else if (SHAPE == 2) {   // split-bf16 kernels: v_mfma_f32_32x32x16_bf16 (32 cycles each)
            bf16x8 p, q;
            for (int k = 0; k < 8; ++k) p[k] = (__bf16)x, q[k] = (__bf16)y;
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q, p, a1, 0, 0, 0);
        } else if (SHAPE == 3) {   // v_mfma_f32_16x16x32_bf16 (16 cycles each): four = the time of two 32x32x16
            bf16x8 p, q;
            for (int k = 0; k < 8; ++k) p[k] = (__bf16)x, q[k] = (__bf16)y;
            b0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, q, b0, 0, 0, 0);
            b1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q, p, b1, 0, 0, 0);
            b0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, q, b0, 0, 0, 0);
            b1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q, p, b1, 0, 0, 0);
        } else {  // two 16x16x4 = the flops of one 32x32x2
            b0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, b0, 0, 0, 0);
            b1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, b1, 0, 0, 0);
            b0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, b0, 0, 0, 0);
            b1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, b1, 0, 0, 0);
        }
    }
}
__device__ __forceinline__ void valu_burst(float (&v)[8], float c, int m)
{
    for (int i = 0; i < m; i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __builtin_fmaf(v[k], c, 0.25f);
    }
}

template <int SHAPE>
__global__ void __launch_bounds__(512) bench(float *out, int mode, int iters, int N, int M)
{
    const int wave = threadIdx.x >> 6;
    f32x16 a0 = {0}, a1 = {0};
    f32x4 b0 = {0}, b1 = {0};
    float v[8];
    for (int k = 0; k < 8; ++k) v[k] = threadIdx.x * 1e-3f + k;
    const float x = 1.0f + threadIdx.x * 1e-6f, y = 0.5f, c = 0.999f;
    for (int it = 0; it < iters; ++it) {
        switch (mode) {
        case 0: mfma_burst<SHAPE>(a0, a1, b0, b1, x, y, N); break;
        case 1: valu_burst(v, c, M); break;
        case 2: if (wave < 4) mfma_burst<SHAPE>(a0, a1, b0, b1, x, y, N); else valu_burst(v, c, M); break;
        case 3: mfma_burst<SHAPE>(a0, a1, b0, b1, x, y, N); valu_burst(v, c, M); break;
        case 4:
            if (wave < 4) { mfma_burst<SHAPE>(a0, a1, b0, b1, x, y, N); valu_burst(v, c, M); }
            else { valu_burst(v, c, M); mfma_burst<SHAPE>(a0, a1, b0, b1, x, y, N); }
            break;
        default:
            for (int i = 0; i < N; i += 2) { mfma_burst<SHAPE>(a0, a1, b0, b1, x, y, 2); valu_burst(v, c, 2 * M / N); }
        }
    }
    float s = 0;
    for (int k = 0; k < 16; ++k) s += a0[k] + a1[k];
    for (int k = 0; k < 4; ++k) s += b0[k] + b1[k];
    for (int k = 0; k < 8; ++k) s += v[k];
    if (s == 123.456f) out[0] = s;
}

int main()
{
    float *d;
    hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 64, iters = 2000;
    const char *names[4] = {"32x32x2 f32", "16x16x4 f32", "32x32x16 bf16", "16x16x32 bf16"};
    for (int shape = 0; shape < 4; ++shape)
        for (int M : {256, 512, 1024}) {
            printf("shape %s  N=%d MFMA(32x32-equivalents) M=%d VALU per iteration:", names[shape], N, M);
            for (int mode = 0; mode < 6; ++mode) {
                float ms = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    if (shape == 0) bench<0><<<256, 512>>>(d, mode, iters, N, M);
                    else if (shape == 1) bench<1><<<256, 512>>>(d, mode, iters, N, M);
                    else if (shape == 2) bench<2><<<256, 512>>>(d, mode, iters, N, M);
                    else bench<3><<<256, 512>>>(d, mode, iters, N, M);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                printf("  m%d %.0fus", mode, ms * 1e3f);
            }
            printf("\n");
        }
    return 0;
}
