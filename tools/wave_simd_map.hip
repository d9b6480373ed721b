// Which SIMD does wave w of a workgroup land on?  (HW_REG_HW_ID: SIMD_ID bits [5:4], CU_ID [11:8] on gfx9-family.)
// Build: hipcc -O2 --offload-arch=gfx950 tools/wave_simd_map.hip -o tools/_build/wave_simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int *out)
{
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);  // HW_ID[15:0]
        out[blockIdx.x * 16 + wave] = (int)hw;
    }
}
int main()
{
    int *d, h[64];
    hipMalloc(&d, sizeof(h));
    for (int threads : {256, 512, 1024}) {
        hipMemset(d, 0, sizeof(h));
        probe<<<2, threads>>>(d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int b = 0; b < 2; ++b) {
            printf("threads %4d block %d: ", threads, b);
            for (int w = 0; w < threads / 64; ++w)
                printf("w%d->simd%d(cu%d) ", w, (h[b * 16 + w] >> 4) & 3, (h[b * 16 + w] >> 8) & 15);
            printf("\n");
        }
    }
    return 0;
}
