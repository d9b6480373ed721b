// Does v_mfma_f32_32x32x16_f16 / 16x16x32_f16 keep fp16 SUBNORMAL inputs (needed by the two-part fp16 split of fp32 operands:
// the low part of a value near the bottom of the fp16 range is subnormal)?  Also times f16 against bf16 MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f16_subnormal.hip -o tools/_build/mfma_f16_subnormal && tools/_build/mfma_f16_subnormal
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__global__ void probe(float a, float b, float *out)
{
    f16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16)a; B[i] = (_Float16)b; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, d, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = d[0]; out[2] = (float)A[0]; }
}

template <int KIND>
__global__ void rate(float *out, int iters)
{
    f32x16 c[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) c[t][i] = 0.f;
    f16x8 A, B; bf16x8 Ab, Bb;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16)(0.001f * threadIdx.x); B[i] = (_Float16)0.5f; Ab[i] = (__bf16)(0.001f * threadIdx.x); Bb[i] = (__bf16)0.5f; }
    for (int it = 0; it < iters; ++it)
        for (int t = 0; t < 4; ++t)
            c[t] = KIND ? __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c[t], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ab, Bb, c[t], 0, 0, 0);
    float s = 0;
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) s += c[t][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    float *d; hipMalloc(&d, 1 << 22);
    float h[3];
    struct { float a, b; const char *what; } cases[] = {
        {9.5367431640625e-07f /*2^-20*/, 1024.f, "subnormal a = 2^-20 x b = 2^10: sum of 16 (32) terms expected 2^-10 x 16 = 0.015625 (x32: 0.03125)"},
        {6.103515625e-05f /*2^-14 min normal*/, 1024.f, "min normal 2^-14 x 2^10 x16 = 1.0 (x32: 2.0)"},
        {5.9604644775390625e-08f /*2^-24 min subnormal*/, 1024.f, "min subnormal 2^-24 x 2^10 x16 = 0.0009765625"},
    };
    for (auto &c : cases) {
        probe<<<1, 64>>>(c.a, c.b, d); hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("%s\n   32x32x16: %.10g   16x16x32: %.10g   (A as fp16 -> %.10g)\n", c.what, h[0], h[1], h[2]);
    }
    for (int kind = 0; kind < 2; ++kind) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 4096;
        if (kind) rate<1><<<1024, 256>>>(d, 16); else rate<0><<<1024, 256>>>(d, 16);
        hipEventRecord(e0);
        if (kind) rate<1><<<1024, 256>>>(d, iters); else rate<0><<<1024, 256>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f ms, %.1f TFLOP/s\n", kind ? "f16 " : "bf16", ms, 1024.0 * 4 * iters * 4 * 2.0 * 32 * 32 * 16 / ms / 1e9);
    }
    return 0;
}
