// Checks on the box (a) the lane <-> element map of ds_read_b64_tr_b16 as cdna_hip_programming.md T10 states it and
// (b) that two tr-read fragments of [row][feature] bf16 images feed v_mfma_f32_32x32x16_bf16 as
// C[i][j] = sum_r P[r][i] Q[r][j]  (the weight-gradient product of mlp_coop.hip: rows are the MFMA k dimension).
//   hipcc --offload-arch=gfx950 -O3 tools/tr_read_check.hip -o tools/_build/tr_read_check && tools/_build/tr_read_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int LD = 72;   // elements per image row (row stride 144 B: the stride mlp_coop.hip uses for 64-wide images)

__device__ __forceinline__ unsigned long long tr_read(const unsigned short *p)
{
    v4i16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16 *)p);
    return __builtin_bit_cast(unsigned long long, r);
}

// fragment (8 k-elements = rows R0..R0+7 at column col) for lane (col = 32*tile + (lane&31), h = lane>>5), k-step ks
__device__ __forceinline__ u32x4 tr_frag(const unsigned short *img, int col_tile, int ks, int lane)
{
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int h = g >> 1;
    const int R0 = 16 * ks + 8 * h, C0 = 32 * col_tile + 16 * (g & 1);
    const unsigned long long a = tr_read(img + (R0 + q) * LD + C0 + 4 * p);
    const unsigned long long b = tr_read(img + (R0 + 4 + q) * LD + C0 + 4 * p);
    return u32x4{(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
}

__global__ void k(const unsigned short *P, const unsigned short *Q, unsigned long long *raw, float *C)
{
    __shared__ __attribute__((aligned(16))) unsigned short sp[32 * LD], sq[32 * LD];
    for (int i = threadIdx.x; i < 32 * LD; i += 64) { sp[i] = P[i]; sq[i] = Q[i]; }
    __syncthreads();
    const int lane = threadIdx.x;
    {   // (a) raw map: block rows 8..11, columns 16..31 for every 16-lane group
        const int q = (lane & 15) >> 2, p = lane & 3;
        raw[lane] = tr_read(sp + (8 + q) * LD + 16 + 4 * p);
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int ks = 0; ks < 2; ++ks) {
        const u32x4 a = tr_frag(sp, 0, ks, lane), b = tr_frag(sq, 1, ks, lane);   // P columns 0..31, Q columns 32..63
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), j = lane & 31;
        C[i * 32 + j] = acc[r];
    }
}

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

int main()
{
    std::vector<unsigned short> P(32 * LD), Q(32 * LD);
    for (int r = 0; r < 32; ++r)
        for (int c = 0; c < LD; ++c) {
            P[r * LD + c] = f2bf((float)((r * 7 + c * 3) % 17 - 8));          // small integers: exact in bf16
            Q[r * LD + c] = f2bf((float)((r * 5 + c * 11) % 13 - 6) * (c & 1 ? 1.f : 0.5f));   // asymmetric
        }
    unsigned short *dP, *dQ; unsigned long long *dr; float *dC;
    hipMalloc(&dP, P.size() * 2); hipMalloc(&dQ, Q.size() * 2); hipMalloc(&dr, 64 * 8); hipMalloc(&dC, 1024 * 4);
    hipMemcpy(dP, P.data(), P.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dQ, Q.data(), Q.size() * 2, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dP, dQ, dr, dC);
    std::vector<unsigned long long> raw(64); std::vector<float> C(1024);
    hipMemcpy(raw.data(), dr, 64 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(C.data(), dC, 1024 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) {
            const unsigned short got = (unsigned short)(raw[l] >> (16 * e));
            const unsigned short want = P[(8 + e) * LD + 16 + (l & 15)];       // lane i of a group: column i, element q: row q
            if (got != want) { if (bad < 8) printf("raw lane %d elem %d: got %04x want %04x\n", l, e, got, want); ++bad; }
        }
    printf("tr map: %s\n", bad ? "MISMATCH" : "ok");
    int badc = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float ref = 0.f;
            for (int r = 0; r < 32; ++r) ref += bf2f(P[r * LD + i]) * bf2f(Q[r * LD + 32 + j]);
            if (ref != C[i * 32 + j]) { if (badc < 8) printf("C[%d][%d] = %g want %g\n", i, j, C[i * 32 + j], ref); ++badc; }
        }
    printf("P^T Q via tr fragments: %s\n", badc ? "MISMATCH" : "ok");
    return (bad || badc) ? 1 : 0;
}
