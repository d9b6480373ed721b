// H3 fused: the whole SDF VanillaMLP (Linear -> Softplus(100) -> Linear -> Softplus(100) -> Linear,
// models/network_utils.py:109-157 with n_hidden_layers = 2) in one kernel each way, for the
// finite-difference stencil layout (tap-major: x7t [7][S][3], hash features planes [L][7][S][2]).
//
// Why: run layer by layer (mlp.hip) the MLP moves ~5.5 KB of activations per evaluation through HBM
// and is memory bound at ~10 % of the fp32 MFMA rate (r01b profile).  Here activations never leave
// the CU:
//   * every layer is computed TRANSPOSED, D[n][row] = sum_k W[n][k] * act[k][row], so a layer's
//     accumulator registers (lane = row, register = output feature) are directly the next layer's
//     MFMA B operand once packed to bf16 (see "Split-bf16 matrix products" below);
//   * the backward recomputes the two hidden layers instead of reading them back, chains
//     d(act) through the same register trick, and forms the weight gradients from LDS-transposed
//     [feature][row] tiles with MFMA (sum over rows = the MFMA k dimension), accumulating them in
//     registers across the workgroup's whole row loop and flushing once with atomics;
//   * the taps only need output column 0 (the SDF), so their last layer is a 64-term dot product on
//     the vector ALU instead of a matrix tile; the full feature row is produced for centre rows only.
// All results are fp32-equivalent (parity tests: 1e-5 relative); summation order differs from a row-major GEMM.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "split_bf16.h"

namespace {

constexpr int BWD_WAVES = 4;      // backward: one wave per SIMD (register-resident weight fragments + accumulators)
constexpr int BWD_THREADS = BWD_WAVES * 64;
#ifndef RSDF_FWD_WAVES_CFG
#define RSDF_FWD_WAVES_CFG 8
#endif
constexpr int FWD_WAVES = RSDF_FWD_WAVES_CFG;   // forward: waves per workgroup (one workgroup per CU): 8 = two per SIMD, 12 = three
// (12 waves fit since b1 / the SDF row are read from LDS and the wave tile is 4.7 KB, and gained 3 % with the dword prefetch;
//  with the dwordx4 prefetch the 168-register budget spills and 8 waves are as fast: 68.8 vs 72.0 ms per quarter step)
constexpr int FWD_THREADS = FWD_WAVES * 64;
constexpr int LDT = 33;       // [feature][row] transposed tiles

__device__ __forceinline__ int n_lo(int r) { return (r & 3) + 8 * (r >> 2); }

// Input tile of one (32-sample group, tap): [32 rows][36] = [xyz*scale+offset | L*2 hash features | 1 (bias column)]
// read from the tap-major SoA buffers (x7t [7][S][3], planes [L][7][S][2]) with full-line loads:
// per level the 32 samples' float2 are 256 contiguous bytes.  The tile is first fetched into 18
// registers per lane (so the loads of tile i+1 fly while tile i is computed), then written to LDS.
struct TileSrc {
    const float *x7t;
    const float *planes;
    int64_t S;
    int n_levels, n_active;
    float xyz_scale, xyz_offset;
};

__device__ __forceinline__ void fetch_tile(float (&pre)[18], const TileSrc &src, int64_t s0, int tap,
                                           int lane)
{
    // RAW loads on clamped (always valid) addresses; the masking happens in store_tile_f.  A load inside a divergent
    // branch, or a select right behind it, is waited for on the spot (vmcnt), which serialises these 18 loads.
    const int64_t last = src.S - 1;
    const int64_t r = s0 + (lane >> 1);
    const int64_t rc = r <= last ? r : last;
#pragma unroll
    for (int l = 0; l < 16; ++l)
        pre[l] = src.planes[(((int64_t)(l < src.n_active ? l : 0) * 7 + tap) * src.S + rc) * 2 + (lane & 1)];
    const float *xb = src.x7t + (int64_t)tap * src.S * 3;
    const int64_t e0 = s0 * 3 + lane, e1 = s0 * 3 + 64 + (lane & 31), emax = src.S * 3 - 1;
    pre[16] = xb[e0 <= emax ? e0 : emax];
    pre[17] = xb[e1 <= emax ? e1 : emax];
}

// ------------------------------------------------------------------------------------------------
// Split-bf16 matrix products.
//
// Measured on MI355X (tools/mfma_valu_overlap.hip): an fp32 MFMA occupies its SIMD for its whole 64 (32)
// cycles -- vector-ALU work of the same or of another wave on that SIMD does not overlap with it, so a
// kernel that mixes fp32 MFMAs with Softplus / transposes pays their SUM.  bf16 MFMAs run at 16x the rate
// and hold the issue port for 8 of their 32 cycles only.  Every fp32 operand is therefore written as the
// exact sum of three bf16 numbers (x = h + m + l: 8 + 8 + 8 mantissa bits, round-to-nearest at each step) and a
// product a.b is evaluated as the six partial products of weight >= 2^-16,
//     al.bh + ah.bl + am.bm + am.bh + ah.bm + ah.bh        (dropped: am.bl, al.bm, al.bl <= 2^-24 |a||b|)
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16, small terms first.  The
// error per product is ~1e-7 |a||b|, the same order as one fp32 rounding, at 6/16 of the fp32-MFMA cost, and the
// matrix pipe now runs in the shadow of the vector ALU work.
//
// Layouts (cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand"): a 32x32 accumulator
// tile D[n][row] keeps row = lane & 31 on the lane and feature (r & 3) + 8 (r >> 2) + 4 (lane >> 5) in register r.
// Registers 8s .. 8s+7, packed pairwise to bf16, ARE the B fragment of k-step s of the next layer; the weights
// are staged in LDS already permuted to that k order, one 16-byte A fragment per lane:
//     Wp[part][n tile][k tile][s][lane >> 5][lane & 31][j]  =  W[32 nt + (lane & 31)][32 kt + 16 s + 8 (j >> 2) + 4 (lane >> 5) + (j & 3)]
// ------------------------------------------------------------------------------------------------
// Softplus(beta = 100, threshold = 20) on the raw exp2 / log2 units:  max(z, 0) + log2(1 + 2^(-|100 z| log2 e)) ln 2 / 100
__device__ __forceinline__ float softplus100_fast(float z)
{
    const float e = __builtin_amdgcn_exp2f(-144.26950408889634f * fabsf(z));
    return fmaf(__builtin_amdgcn_logf(1.0f + e), 0.0069314718055994531f, max0(z));
}
// sigmoid(100 z) = 1 - 2^(-100 log2(e) h) from h = softplus(z)
__device__ __forceinline__ float softplus100_grad_fast(float h)
{
    return 1.0f - __builtin_amdgcn_exp2f(-144.26950408889634f * h);
}

constexpr int LDXF = 36;   // fp32 X tile row stride: 16-byte aligned rows, conflict-free ds_read_b128 over 8 rows
constexpr int LDFS = 33;   // transposed [row][32 features] staging of the forward's row stores
constexpr int KS0 = 3;     // layer-1 k-steps of 16 (K0 <= 35, column 35 = 1 carries the bias, 36..47 hit zero weights)

// LDS image of the split weights (units: u32x4 = 16 bytes)
template <int H>
struct SmemS {
    static constexpr int NT = H / 32;
    static constexpr int W0_PART = NT * KS0 * 2 * 32;            // [nt][s][hf][c]
    static constexpr int W1_PART = NT * NT * 2 * 2 * 32;         // [nt][kt][s][hf][c]
    static constexpr int W2_PART = 2 * NT * 2 * 2 * 32;          // [n2 tile (2)][kt][s][hf][c]
    static constexpr int W0 = 0;
    static constexpr int W1 = W0 + 3 * W0_PART;
    static constexpr int W2 = W1 + 3 * W1_PART;
    static constexpr int END_U4 = W2 + 3 * W2_PART;
    // fp32 tail (float index from the start of the tail)
    static constexpr int B1 = 0;          // [H]
    static constexpr int B2 = B1 + H;     // [64]
    static constexpr int W2R0 = B2 + 64;  // [H]   row 0 of W2 (SDF) for the vector-ALU dot of the taps
    static constexpr int TAIL_F = W2R0 + H;
    static constexpr size_t SHARED_BYTES = (size_t)END_U4 * 16 + (size_t)TAIL_F * 4;
    // X tile [32][LDXF] fp32 (+ the 12 pad columns the last row's third k-step reads), overlaid by one 32-column half of
    // the feature transpose [32][LDFS]
    static constexpr int PER_WAVE_F = 32 * LDXF + 16;
};

// k of element j of lane half hf in k-step s of a 32-feature activation tile
__device__ __forceinline__ int frag_k(int s, int hf, int j) { return 16 * s + 8 * (j >> 2) + 4 * hf + (j & 3); }

// Stage W0 (+ b0 as column 35), W1, W2 as split, fragment-ordered bf16; b1, b2, W2 row 0 in fp32.
template <int H>
__device__ __forceinline__ void stage_split_weights(unsigned char *smem, const float *__restrict__ w0,
                                                    const float *__restrict__ b0, const float *__restrict__ w1,
                                                    const float *__restrict__ b1, const float *__restrict__ w2,
                                                    const float *__restrict__ b2, int K0, int N2)
{
    using S = SmemS<H>;
    constexpr int NT = S::NT;
    unsigned short *e16 = reinterpret_cast<unsigned short *>(smem);
    const int NTHR = blockDim.x;
    // W0: [nt][s][hf][c][j], natural k = 16 s + 8 hf + j (the X tile is read from LDS in that order)
    for (int e = threadIdx.x; e < NT * KS0 * 2 * 32 * 8; e += NTHR) {
        const int j = e & 7, c = (e >> 3) & 31, hf = (e >> 8) & 1, s = (e >> 9) % KS0, nt = (e >> 9) / KS0;
        const int n = 32 * nt + c, k = 16 * s + 8 * hf + j;
        const float w = k < K0 ? w0[n * K0 + k] : (k == 35 ? b0[n] : 0.0f);
        store3(e16 + (size_t)S::W0 * 8, (size_t)S::W0_PART * 8, e, w);
    }
    for (int e = threadIdx.x; e < NT * NT * 2 * 2 * 32 * 8; e += NTHR) {
        const int j = e & 7, c = (e >> 3) & 31, hf = (e >> 8) & 1, s = (e >> 9) & 1, kt = (e >> 10) % NT,
                  nt = (e >> 10) / NT;
        const int n = 32 * nt + c, k = 32 * kt + frag_k(s, hf, j);
        store3(e16 + (size_t)S::W1 * 8, (size_t)S::W1_PART * 8, e, w1[n * H + k]);
    }
    for (int e = threadIdx.x; e < 2 * NT * 2 * 2 * 32 * 8; e += NTHR) {
        const int j = e & 7, c = (e >> 3) & 31, hf = (e >> 8) & 1, s = (e >> 9) & 1, kt = (e >> 10) % NT,
                  nt = (e >> 10) / NT;
        const int n = 32 * nt + c, k = 32 * kt + frag_k(s, hf, j);
        store3(e16 + (size_t)S::W2 * 8, (size_t)S::W2_PART * 8, e, n < N2 ? w2[n * H + k] : 0.0f);
    }
    float *tail = reinterpret_cast<float *>(smem + (size_t)S::END_U4 * 16);
    for (int e = threadIdx.x; e < H; e += NTHR) {
        tail[S::B1 + e] = b1[e];
        tail[S::W2R0 + e] = w2[e];
    }
    for (int e = threadIdx.x; e < 64; e += NTHR) tail[S::B2 + e] = e < N2 ? b2[e] : 0.0f;
}

__device__ __forceinline__ void store_tile_f(float *Xs, const float (&pre)[18], const TileSrc &src, int64_t s0, int lane)
{
    const int r = lane >> 1, f = lane & 1;
    const bool ok = s0 + r < src.S;
#pragma unroll
    for (int l = 0; l < 16; ++l) Xs[r * LDXF + 3 + 2 * l + f] = (ok && l < src.n_active) ? pre[l] : 0.0f;
    const float x0 = s0 + lane / 3 < src.S ? pre[16] : 0.5f, x1 = s0 + (lane + 64) / 3 < src.S ? pre[17] : 0.5f;
    Xs[(lane / 3) * LDXF + lane % 3] = x0 * src.xyz_scale + src.xyz_offset;
    if (lane < 32) Xs[((lane + 64) / 3) * LDXF + (lane + 64) % 3] = x1 * src.xyz_scale + src.xyz_offset;
    if (lane < 32) Xs[lane * LDXF + 35] = 1.0f;   // bias column
}

// Wide variant for FULL tiles (the forward is bound by its vector instructions: 18 dword loads with their address
// arithmetic per tile become 5 dwordx4 loads).  v[4k .. 4k+3] = level 4k + lane / 16, rows 2 (lane % 16) and + 1, both
// features; v[16 .. 19] = floats 4 lane .. + 3 of the tile's 96 point coordinates (lanes >= 24 re-read a valid address).
// A partial tile (the last one of a launch) keeps the dword path; v is shared storage.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void fetch_tile4(float (&v)[20], const TileSrc &src, int64_t s0, int tap, int lane)
{
    if (s0 + 32 > src.S) {
        float pre[18];
        fetch_tile(pre, src, s0, tap, lane);
#pragma unroll
        for (int i = 0; i < 18; ++i) v[i] = pre[i];
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int l = 4 * k + (lane >> 4);
        const f32x4u t = *reinterpret_cast<const f32x4u *>(
            src.planes + (((int64_t)(l < src.n_active ? l : 0) * 7 + tap) * src.S + s0 + 2 * (lane & 15)) * 2);
        v[4 * k] = t[0]; v[4 * k + 1] = t[1]; v[4 * k + 2] = t[2]; v[4 * k + 3] = t[3];
    }
    // (lanes 24 .. 63 re-read one of the 24 valid 16-byte pieces; `lane - 24` alone sent lanes 48 .. 63 up to 256 bytes past
    // the tile, i.e. past the END of x7t on the last tap of the last full tile: tests/test_gpu_guard_pages.py)
    const int xl = lane < 24 ? lane : (lane < 48 ? lane - 24 : lane - 48);
    const f32x4u t = *reinterpret_cast<const f32x4u *>(src.x7t + ((int64_t)tap * src.S + s0) * 3 + 4 * xl);
    v[16] = t[0]; v[17] = t[1]; v[18] = t[2]; v[19] = t[3];
}
__device__ __forceinline__ void store_tile4(float *Xs, const float (&v)[20], const TileSrc &src, int64_t s0, int lane)
{
    if (s0 + 32 > src.S) {
        float pre[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) pre[i] = v[i];
        store_tile_f(Xs, pre, src, s0, lane);
        return;
    }
    const int r0 = 2 * (lane & 15);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int l = 4 * k + (lane >> 4);
        const bool on = l < src.n_active;
        float *p = Xs + r0 * LDXF + 3 + 2 * l;
        p[0] = on ? v[4 * k] : 0.0f;
        p[1] = on ? v[4 * k + 1] : 0.0f;
        p[LDXF] = on ? v[4 * k + 2] : 0.0f;
        p[LDXF + 1] = on ? v[4 * k + 3] : 0.0f;
    }
    if (lane < 24) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = 4 * lane + i;
            Xs[(e / 3) * LDXF + e % 3] = v[16 + i] * src.xyz_scale + src.xyz_offset;
        }
    }
    if (lane < 32) Xs[lane * LDXF + 35] = 1.0f;   // bias column
}

// hidden layers 1 and 2 of one 32-row tile; h[t][r]: feature 32 t + (r & 3) + 8 (r >> 2) + 4 hf of row c
template <int H>
__device__ __forceinline__ void hidden_forward_s(const unsigned char *smem, const float *Xs, int c, int hf,
                                                 const float *b1s, f32x16 (&h1)[H / 32],
                                                 f32x16 (&h2)[H / 32])
{
    using S = SmemS<H>;
    constexpr int NT = S::NT;
    const u32x4 *wl = reinterpret_cast<const u32x4 *>(smem);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) h1[t][r] = 0.0f;
#pragma unroll
    for (int s = 0; s < KS0; ++s) {
        const float4 x0 = *reinterpret_cast<const float4 *>(Xs + c * LDXF + 16 * s + 8 * hf);
        const float4 x1 = *reinterpret_cast<const float4 *>(Xs + c * LDXF + 16 * s + 8 * hf + 4);
        const Frag3 xb = split_frag(x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w);
#pragma unroll
        for (int t = 0; t < NT; ++t)
            h1[t] = mma6<S::W0_PART>(wl + S::W0 + ((t * KS0 + s) * 2 + hf) * 32 + c, xb, h1[t]);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) h1[t][r] = softplus100_fast(h1[t][r]);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // bias of features 32 t + 8 q + 4 hf .. + 3: one 16-byte LDS read
            const float4 b = *reinterpret_cast<const float4 *>(b1s + 32 * t + 8 * q + 4 * hf);
            h2[t][4 * q] = b.x, h2[t][4 * q + 1] = b.y, h2[t][4 * q + 2] = b.z, h2[t][4 * q + 3] = b.w;
        }
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const Frag3 hb = split_frag(h1[kt][8 * s], h1[kt][8 * s + 1], h1[kt][8 * s + 2], h1[kt][8 * s + 3],
                                        h1[kt][8 * s + 4], h1[kt][8 * s + 5], h1[kt][8 * s + 6], h1[kt][8 * s + 7]);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                h2[t] = mma6<S::W1_PART>(wl + S::W1 + (((t * NT + kt) * 2 + s) * 2 + hf) * 32 + c, hb, h2[t]);
        }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) h2[t][r] = softplus100_fast(h2[t][r]);
}

// ------------------------------------------------------------------------------------------------
// forward: tap-major inputs -> sdf7t [7][S], feature [S, N2] (nullable; centre rows only), h2c (nullable)
// ------------------------------------------------------------------------------------------------
// RSDF_FWD_VGPR_CAP: 192 registers x 2 waves leave 128 of a SIMD's 512 free, i.e. room for one wave of the stencil
// gather (fd7_fwd_kernel, 128 registers) beside this kernel: NeuSModel's pipelined forward runs the next chunk's gather on
// a second stream while this (vector-issue-bound) kernel computes.
// (on gfx90a+ the attribute counts architectural VGPRs of the unified file: LLVM doubles it, so 96 = 192 unified.)
#ifndef RSDF_FWD_VGPR_CAP
#define RSDF_FWD_VGPR_CAP 96
#endif
template <int H>
__global__ void __launch_bounds__(FWD_THREADS, FWD_WAVES / 4) __attribute__((amdgpu_num_vgpr(RSDF_FWD_VGPR_CAP)))
sdfmlp_fwd_kernel(const TileSrc src, const float *__restrict__ w0,
                  const float *__restrict__ b0, const float *__restrict__ w1,
                  const float *__restrict__ b1, const float *__restrict__ w2,
                  const float *__restrict__ b2, int N2, float *__restrict__ sdf7,
                  float *__restrict__ feature, float *__restrict__ h2c)
{
    const int64_t n_samples = src.S;
    const int K0 = 3 + 2 * src.n_levels;
    using S = SmemS<H>;
    constexpr int NT = S::NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, hf = lane >> 5;
    float *tail = reinterpret_cast<float *>(smem_b + (size_t)S::END_U4 * 16);
    float *Xs = reinterpret_cast<float *>(smem_b + S::SHARED_BYTES) + wave * S::PER_WAVE_F;
    float *Fs = Xs;  // [32][LDFS] over the X tile, which is dead once the first layer has read it
    stage_split_weights<H>(smem_b, w0, b0, w1, b1, w2, b2, K0, N2);
    for (int e = lane; e < S::PER_WAVE_F; e += 64) Xs[e] = 0.0f;   // pad columns must be finite
    __syncthreads();
    const u32x4 *wl = reinterpret_cast<const u32x4 *>(smem_b);

    // b1 and the SDF row of W2 stay in LDS (16-byte reads at their use): 64 registers less per wave, which is what lets
    // three waves share a SIMD
    const float b2_0 = tail[S::B2];

    const int64_t n_groups = (n_samples + 31) / 32;  // 32 samples per wave iteration
    const int64_t g_first = (int64_t)blockIdx.x * FWD_WAVES + wave, g_step = (int64_t)gridDim.x * FWD_WAVES;
    float pre[20];
    if (g_first < n_groups) fetch_tile4(pre, src, g_first * 32, 0, lane);
    for (int64_t g = g_first; g < n_groups; g += g_step) {
        const int64_t s0 = g * 32;
        for (int tap = 0; tap < 7; ++tap) {
            store_tile4(Xs, pre, src, s0, lane);
            {   // prefetch the next tile of this wave
                const int ntap = tap == 6 ? 0 : tap + 1;
                const int64_t ng = tap == 6 ? g + g_step : g;
                if (ng < n_groups) fetch_tile4(pre, src, ng * 32, ntap, lane);
            }
            f32x16 h1[NT], h2[NT];
            hidden_forward_s<H>(smem_b, Xs, c, hf, tail + S::B1, h1, h2);
            const int64_t s = s0 + c;
            // SDF: dot(W2[0,:], h2[:,row]) on the vector ALU, features split over the lane halves
            float acc = 0.0f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w = *reinterpret_cast<const float4 *>(tail + S::W2R0 + 32 * t + 8 * q + 4 * hf);
                    acc = fmaf(w.x, h2[t][4 * q], acc);
                    acc = fmaf(w.y, h2[t][4 * q + 1], acc);
                    acc = fmaf(w.z, h2[t][4 * q + 2], acc);
                    acc = fmaf(w.w, h2[t][4 * q + 3], acc);
                }
            acc += __shfl_xor(acc, 32, 64);
            if (hf == 0 && s < n_samples) sdf7[(int64_t)tap * n_samples + s] = acc + b2_0;
            if (tap == 0 && feature != nullptr) {
                if (h2c != nullptr) {  // second hidden layer of the centre rows, for the dW2 of features
#pragma unroll
                    for (int t = 0; t < NT; ++t) {   // 32 features at a time through the wave's LDS tile
#pragma unroll
                        for (int r = 0; r < 16; ++r) Fs[c * LDFS + (r & 3) + 8 * (r >> 2) + 4 * hf] = h2[t][r];
                        for (int e = lane; e < 32 * 32; e += 64) {
                            const int r = e >> 5, cc = e & 31;
                            if (s0 + r < n_samples) h2c[(s0 + r) * H + 32 * t + cc] = Fs[r * LDFS + cc];
                        }
                    }
                }
                // full last layer on the matrix cores: out[n2][row]
                f32x16 o[2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[t][r] = tail[S::B2 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf];
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int ss = 0; ss < 2; ++ss) {
                        const Frag3 hb = split_frag(h2[kt][8 * ss], h2[kt][8 * ss + 1], h2[kt][8 * ss + 2],
                                                    h2[kt][8 * ss + 3], h2[kt][8 * ss + 4], h2[kt][8 * ss + 5],
                                                    h2[kt][8 * ss + 6], h2[kt][8 * ss + 7]);
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            if (t * 32 < N2)
                                o[t] = mma6<S::W2_PART>(wl + S::W2 + (((t * NT + kt) * 2 + ss) * 2 + hf) * 32 + c, hb,
                                                        o[t]);
                    }
                // transpose through LDS for coalesced row stores, 32 columns at a time
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int nc = N2 - 32 * t < 32 ? N2 - 32 * t : 32;   // columns of this half (wave-uniform)
                    if (nc <= 0) break;
#pragma unroll
                    for (int r = 0; r < 16; ++r) Fs[c * LDFS + (r & 3) + 8 * (r >> 2) + 4 * hf] = o[t][r];
                    for (int e = lane; e < 32 * nc; e += 64) {
                        const int r = e / nc, cc = e - r * nc;
                        if (s0 + r < n_samples) feature[(s0 + r) * N2 + 32 * t + cc] = Fs[r * LDFS + cc];
                    }
                }
                // the X tile region now holds feature rows: restore finite pad columns for the next tile
                // (columns 36.. of a row alias the next row's data, which stays finite; nothing to do)
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward: d_sdf7t [7][S] (+ d_feature [S,N2], centre rows) -> d_planes [L][7][S][2], dW0, db0, dW1, db1,
//           dW2 row 0, db2[0]   (atomically accumulated: zero first)
//
// One wave per SIMD (4 per workgroup, up to 512 registers each).  Per 32-row tile:
//   recompute h1, h2                      split-bf16 products, weights from LDS (as the forward)
//   dz2 = (W2^T d_out) sigma'(z2)         SDF column on the vector ALU; feature columns (centre tap) split-bf16 from LDS
//   dz1 = (W1^T dz2) sigma'(z1), dx = W0^T dz1    split-bf16, the transposed weight fragments live in registers
//   dW1 += dz2^T h1, dW0 += dz1^T x       fp32 MFMA over [feature][row] tiles in LDS (sum over rows = MFMA k),
//                                         accumulated in registers across the whole row loop, flushed once
// ------------------------------------------------------------------------------------------------
constexpr int LDF = 68;   // d_feature staging [32][LDF] fp32: 16-byte aligned rows, conflict-free b128 reads

template <int H>
struct SmemSB {
    using F = SmemS<H>;
    static constexpr int NT = H / 32;
    static constexpr int W0 = 0;
    static constexpr int W1 = W0 + 3 * F::W0_PART;
    static constexpr int W2T = W1 + 3 * F::W1_PART;                 // [part][t][s][hf][c], s < steps2
    static constexpr int w2t_part(int steps2) { return NT * steps2 * 2 * 32; }
    static constexpr int end_u4(int steps2) { return W2T + 3 * w2t_part(steps2); }
    // fp32 tail
    static constexpr int B1 = 0;          // [H]
    static constexpr int W2R0 = B1 + H;   // [H]
    static constexpr int TAIL_F = W2R0 + H;
    static constexpr size_t shared_bytes(int steps2) { return (size_t)end_u4(steps2) * 16 + (size_t)TAIL_F * 4; }
    // per wave (floats)
    static constexpr int XS = 0;                         // [32][LDXF]
    static constexpr int TA = XS + 32 * LDXF;            // [H][LDT]
    static constexpr int TD = TA + H * LDT;              // [H][LDT]
    static constexpr int TILES_END = TD + H * LDT + 16;  // + slack for the pad-column over-read of the last row
    static constexpr int FS_END = TA + 32 * LDF;         // d_feature staging overlays Ta|Td
    static constexpr int PER_WAVE_F = TILES_END > FS_END ? TILES_END : FS_END;
};

template <int H>
__device__ __forceinline__ void stage_split_weights_bwd(unsigned char *smem, const float *__restrict__ w0,
                                                        const float *__restrict__ b0, const float *__restrict__ w1,
                                                        const float *__restrict__ b1, const float *__restrict__ w2,
                                                        int K0, int N2, int steps2)
{
    using S = SmemSB<H>;
    using F = SmemS<H>;
    constexpr int NT = S::NT;
    unsigned short *e16 = reinterpret_cast<unsigned short *>(smem);
    const int NTHR = blockDim.x;
    for (int e = threadIdx.x; e < NT * KS0 * 2 * 32 * 8; e += NTHR) {
        const int j = e & 7, c = (e >> 3) & 31, hf = (e >> 8) & 1, s = (e >> 9) % KS0, nt = (e >> 9) / KS0;
        const int n = 32 * nt + c, k = 16 * s + 8 * hf + j;
        const float w = k < K0 ? w0[n * K0 + k] : (k == 35 ? b0[n] : 0.0f);
        store3(e16 + (size_t)S::W0 * 8, (size_t)F::W0_PART * 8, e, w);
    }
    for (int e = threadIdx.x; e < NT * NT * 2 * 2 * 32 * 8; e += NTHR) {
        const int j = e & 7, c = (e >> 3) & 31, hf = (e >> 8) & 1, s = (e >> 9) & 1, kt = (e >> 10) % NT,
                  nt = (e >> 10) / NT;
        const int n = 32 * nt + c, k = 32 * kt + frag_k(s, hf, j);
        store3(e16 + (size_t)S::W1 * 8, (size_t)F::W1_PART * 8, e, w1[n * H + k]);
    }
    // W2T: A[i = h2 feature 32 t + c][k = n2 = 16 s + 8 hf + j] = W2[n2][32 t + c]
    for (int e = threadIdx.x; e < NT * steps2 * 2 * 32 * 8; e += NTHR) {
        const int j = e & 7, c = (e >> 3) & 31, hf = (e >> 8) & 1, s = (e >> 9) % steps2, t = (e >> 9) / steps2;
        const int n2 = 16 * s + 8 * hf + j;
        store3(e16 + (size_t)S::W2T * 8, (size_t)S::w2t_part(steps2) * 8, e, n2 < N2 ? w2[n2 * H + 32 * t + c] : 0.0f);
    }
    float *tail = reinterpret_cast<float *>(smem + (size_t)S::end_u4(steps2) * 16);
    for (int e = threadIdx.x; e < H; e += NTHR) {
        tail[S::B1 + e] = b1[e];
        tail[S::W2R0 + e] = w2[e];
    }
}

__device__ __forceinline__ Frag3 frag_of(const f32x16 &v, int s)
{
    return split_frag(v[8 * s], v[8 * s + 1], v[8 * s + 2], v[8 * s + 3], v[8 * s + 4], v[8 * s + 5], v[8 * s + 6],
                      v[8 * s + 7]);
}

template <int H>
__global__ void __launch_bounds__(BWD_THREADS)
sdfmlp_bwd_kernel(const TileSrc src, const float *__restrict__ w0,
                  const float *__restrict__ b0, const float *__restrict__ w1,
                  const float *__restrict__ b1, const float *__restrict__ w2,
                  const float *__restrict__ b2, int N2, int steps2,
                  const float *__restrict__ d_sdf7, const float *__restrict__ d_feature,
                  float *__restrict__ d_planes,
                  float *__restrict__ dw0, float *__restrict__ db0,
                  float *__restrict__ dw1, float *__restrict__ db1, float *__restrict__ dw2,
                  float *__restrict__ db2)
{
    const int64_t n_samples = src.S;
    const int K0 = 3 + 2 * src.n_levels;
    const int k0w = 3, kw = 2 * src.n_levels;
    using S = SmemSB<H>;
    using F = SmemS<H>;
    constexpr int NT = S::NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, hf = lane >> 5;
    const int li = c, lh = hf;
    const float *tail = reinterpret_cast<const float *>(smem_b + (size_t)S::end_u4(steps2) * 16);
    float *wbase = reinterpret_cast<float *>(smem_b + S::shared_bytes(steps2)) + wave * S::PER_WAVE_F;
    float *Xs = wbase + S::XS, *Ta = wbase + S::TA, *Td = wbase + S::TD;
    stage_split_weights_bwd<H>(smem_b, w0, b0, w1, b1, w2, K0, N2, steps2);
    for (int e = lane; e < S::PER_WAVE_F; e += 64) wbase[e] = 0.0f;   // pad columns must be finite
    __syncthreads();
    const u32x4 *wl = reinterpret_cast<const u32x4 *>(smem_b);

    // transposed weight fragments, register resident for the whole kernel
    //   w1t[t][kt][s]: A[i = k1 = 32 t + c][k2 = 32 kt + frag_k(s, hf, j)] = W1[k2][k1]
    //   w0t[kt][s]:    A[i = hash column c][k1 = 32 kt + frag_k(s, hf, j)] = W0[k1][3 + c]
    Frag3 w1t[NT][NT][2], w0t[NT][2];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float v[8];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = w1[(32 * kt + frag_k(s, hf, j)) * H + 32 * t + c];
                w1t[t][kt][s] = split_frag(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = c < kw ? w0[(32 * kt + frag_k(s, hf, j)) * K0 + k0w + c] : 0.0f;
            w0t[kt][s] = split_frag(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
        }

    // gradient accumulators that live across the whole row loop
    f32x16 gw1[NT][NT];  // dW1[n tile][k tile]
    f32x16 gw0[NT];      // dW0[n tile][input columns 0..31]
    f32x16 gw2p[NT];     // per-lane (= per row slot) partial of dW2[0][feature]
#pragma unroll
    for (int a = 0; a < NT; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { gw0[a][r] = 0.0f; gw2p[a][r] = 0.0f; }
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) gw1[a][b][r] = 0.0f;
    }
    float gb1 = 0.0f, gb0 = 0.0f, gb2 = 0.0f;     // lane f (< H) owns feature f
    float gw0_tail[4] = {0.f, 0.f, 0.f, 0.f};     // dW0[f][32..35]

    const int64_t n_groups = (n_samples + 31) / 32;
    const int64_t g_first = (int64_t)blockIdx.x * BWD_WAVES + wave, g_step = (int64_t)gridDim.x * BWD_WAVES;
    float pre[18];
    if (g_first < n_groups) fetch_tile(pre, src, g_first * 32, 0, lane);
    for (int64_t g = g_first; g < n_groups; g += g_step) {
        const int64_t s0 = g * 32;
        for (int tap = 0; tap < 7; ++tap) {
            store_tile_f(Xs, pre, src, s0, lane);
            {
                const int ntap = tap == 6 ? 0 : tap + 1;
                const int64_t ng = tap == 6 ? g + g_step : g;
                if (ng < n_groups) fetch_tile(pre, src, ng * 32, ntap, lane);
            }
            f32x16 h1[NT], dz2[NT];
            const int64_t s = s0 + c;
            const float dsdf_raw = d_sdf7[(int64_t)tap * n_samples + (s < n_samples ? s : n_samples - 1)];
            {
                f32x16 h2[NT];
                {   // the forward's hidden layers, weights at this kernel's LDS offsets (same image order)
                    static_assert(S::W0 == F::W0 && S::W1 == F::W1, "recompute shares the forward's weight image");
                    hidden_forward_s<H>(smem_b, Xs, c, hf, tail + S::B1, h1, h2);
                }
                const float dsdf = s < n_samples ? dsdf_raw : 0.0f;
                gb2 += (hf == 0) ? dsdf : 0.0f;
                // ---- layer 3: dh2 = W2^T d_out.  Output channel 0 is the SDF: on the centre tap with feature
                // gradients d_sdf joins column 0 of d_feature and the whole product runs on the matrix cores;
                // otherwise it is the rank-1 term W2[0,:] d_sdf on the vector ALU.
                if (tap == 0 && d_feature != nullptr) {
                    float *Fs = Ta;  // [32][LDF] over Ta|Td (both idle here)
                    for (int e = lane; e < 32 * 64; e += 64) {
                        const int r = e >> 6, cc = e & 63;
                        float v = (s0 + r < n_samples && cc < N2) ? d_feature[(s0 + r) * N2 + cc] : 0.0f;
                        Fs[r * LDF + cc] = v;
                    }
                    if (hf == 0) Fs[c * LDF] += dsdf;
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) dz2[t][r] = 0.0f;
                    for (int ss = 0; ss < steps2; ++ss) {
                        const float4 x0 = *reinterpret_cast<const float4 *>(Fs + c * LDF + 16 * ss + 8 * hf);
                        const float4 x1 = *reinterpret_cast<const float4 *>(Fs + c * LDF + 16 * ss + 8 * hf + 4);
                        const Frag3 fb = split_frag(x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w);
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const u32x4 *wa = wl + S::W2T + ((t * steps2 + ss) * 2 + hf) * 32 + c;
                            const int ps = S::w2t_part(steps2);
                            const u32x4 ah = wa[0], am = wa[ps], al = wa[2 * ps];
                            dz2[t] = mma_bf16(al, fb.h, dz2[t]);
                            dz2[t] = mma_bf16(ah, fb.l, dz2[t]);
                            dz2[t] = mma_bf16(am, fb.m, dz2[t]);
                            dz2[t] = mma_bf16(am, fb.h, dz2[t]);
                            dz2[t] = mma_bf16(ah, fb.m, dz2[t]);
                            dz2[t] = mma_bf16(ah, fb.h, dz2[t]);
                        }
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            dz2[t][r] = tail[S::W2R0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hf] * dsdf;
                }
                // dW2[0,:] += d_sdf h2 (per-lane partial) ; dz2 = dh2 sigma'(z2)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        gw2p[t][r] = fmaf(dsdf, h2[t][r], gw2p[t][r]);
                        dz2[t][r] *= softplus100_grad_fast(h2[t][r]);
                    }
            }
            // ---- layer 2: dW1 += dz2^T h1 ; db1 += colsum(dz2) ; dz1 = (W1^T dz2) sigma'(z1) -------------
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    Td[(t * 32 + n_lo(r) + 4 * lh) * LDT + li] = dz2[t][r];
                    Ta[(t * 32 + n_lo(r) + 4 * lh) * LDT + li] = h1[t][r];
                }
            for (int rs = 0; rs < 32; rs += 2) {  // fp32 MFMA, k = row
                float av[NT], bv[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    av[t] = Td[(t * 32 + li) * LDT + rs + lh];
                    bv[t] = Ta[(t * 32 + li) * LDT + rs + lh];
                }
#pragma unroll
                for (int a = 0; a < NT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b)
                        gw1[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], gw1[a][b], 0, 0, 0);
            }
            if (lane < H) {
                float acc = 0.0f;
#pragma unroll 8
                for (int row = 0; row < 32; ++row) acc += Td[lane * LDT + row];
                gb1 += acc;
            }
            f32x16 dz1[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) dz1[t][r] = 0.0f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {
                    const Frag3 zb = frag_of(dz2[kt], ss);
#pragma unroll
                    for (int t = 0; t < NT; ++t) dz1[t] = mma6r(w1t[t][kt][ss], zb, dz1[t]);
                }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) dz1[t][r] *= softplus100_grad_fast(h1[t][r]);

            // ---- layer 1: dW0 += dz1^T x ; db0 += colsum(dz1) ; dx = W0^T dz1 (hash-feature columns) ------
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) Td[(t * 32 + n_lo(r) + 4 * lh) * LDT + li] = dz1[t][r];
            for (int rs = 0; rs < 32; rs += 2) {
                const float bx = Xs[(rs + lh) * LDXF + li];  // B[k = row][j = input column li]
#pragma unroll
                for (int a = 0; a < NT; ++a) {
                    const float av = Td[(a * 32 + li) * LDT + rs + lh];
                    gw0[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bx, gw0[a], 0, 0, 0);
                }
            }
            if (lane < H) {
                float acc = 0.0f, t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll 8
                for (int row = 0; row < 32; ++row) {
                    const float d = Td[lane * LDT + row];
                    acc += d;
                    t0 = fmaf(d, Xs[row * LDXF + 32], t0);
                    t1 = fmaf(d, Xs[row * LDXF + 33], t1);
                    t2 = fmaf(d, Xs[row * LDXF + 34], t2);
                }
                gb0 += acc;
                gw0_tail[0] += t0; gw0_tail[1] += t1; gw0_tail[2] += t2;
            }
            if (d_planes != nullptr) {
                f32x16 dx;
#pragma unroll
                for (int r = 0; r < 16; ++r) dx[r] = 0.0f;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int ss = 0; ss < 2; ++ss) dx = mma6r(w0t[kt][ss], frag_of(dz1[kt], ss), dx);
                // dx[c][row] -> Ta as [row][c] -> coalesced 128-byte row stores
#pragma unroll
                for (int r = 0; r < 16; ++r) Ta[li * LDT + n_lo(r) + 4 * lh] = dx[r];
                // level planes [L][7][S][2]: per level 32 samples x float2 = 256 contiguous bytes
                if (s0 + (lane >> 1) < n_samples) {
                    for (int l = 0; l < src.n_active; ++l)
                        d_planes[(((int64_t)l * 7 + tap) * n_samples + s0) * 2 + lane] =
                            Ta[(lane >> 1) * LDT + 2 * l + (lane & 1)];
                }
            }
        }
    }

    // ---- flush the accumulated parameter gradients ------------------------------------------------
#pragma unroll
    for (int a = 0; a < NT; ++a) {
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = a * 32 + n_lo(r) + 4 * lh, k = b * 32 + li;
                atomicAdd(&dw1[n * H + k], gw1[a][b][r]);
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = a * 32 + n_lo(r) + 4 * lh, k = li;
            if (k < K0) atomicAdd(&dw0[n * K0 + k], gw0[a][r]);
        }
    }
    // dW2[0][f]: sum the per-lane partials over the 32 row slots (transpose through Ta)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) Ta[(t * 32 + n_lo(r) + 4 * lh) * LDT + li] = gw2p[t][r];
    if (lane < H) {
        float acc = 0.0f;
#pragma unroll 8
        for (int row = 0; row < 32; ++row) acc += Ta[lane * LDT + row];
        atomicAdd(&dw2[lane], acc);  // row 0 of dW2 [N2, H]
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
            if (32 + cc < K0) atomicAdd(&dw0[lane * K0 + 32 + cc], gw0_tail[cc]);
        atomicAdd(&db0[lane], gb0);
        atomicAdd(&db1[lane], gb1);
    }
    gb2 = wave_sum(gb2);
    if (lane == 0) atomicAdd(&db2[0], gb2);
}

template <int H>
size_t fwd_lds() { return SmemS<H>::SHARED_BYTES + (size_t)FWD_WAVES * SmemS<H>::PER_WAVE_F * sizeof(float); }
template <int H>
size_t bwd_lds(int steps2)
{
    return SmemSB<H>::shared_bytes(steps2) + (size_t)BWD_WAVES * SmemSB<H>::PER_WAVE_F * sizeof(float);
}

template <typename K>
int set_lds(K kern, size_t bytes)
{
    // the attribute sticks to the function on a device: set it once per (kernel, device, size) and host thread
    static thread_local const void *done[8] = {};
    static thread_local size_t done_key[8] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t key = bytes * 64 + (size_t)dev + 1;
    for (int i = 0; i < 8; ++i)
        if (done[i] == reinterpret_cast<const void *>(kern) && done_key[i] == key) return 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { rsdf_set_error(hipGetErrorString(e)); return (int)e; }
    for (int i = 0; i < 8; ++i)
        if (done[i] == nullptr || done[i] == reinterpret_cast<const void *>(kern)) {
            done[i] = reinterpret_cast<const void *>(kern);
            done_key[i] = key;
            break;
        }
    return 0;
}

unsigned persistent_grid(int64_t n_samples, int waves)
{
    const int64_t groups = (n_samples + 31) / 32;
    const int64_t want = (groups + waves - 1) / waves;
    return (unsigned)(want < 512 ? (want > 0 ? want : 1) : 512);
}

bool env_is(const char *name, const char *value) { return rsdf_env_is(name, value); }   // cached getenv (core.hip)

}  // namespace

// cooperative kernels (mlp_coop.hip): a workgroup of H/32 waves, wave w owns features 32w..32w+31
__attribute__((visibility("hidden"))) int RSDF_P(rsdf_coop_fwd)(int NT, const float *x7t, const float *planes, int n_levels, int n_active, float xyz_scale,
                  float xyz_offset, int N2, const float *w0, const float *b0, const float *w1, const float *b1,
                  const float *w2, const float *b2, int64_t n_samples, float *sdf7t, float *feature, float *h2c,
                  hipStream_t st);
__attribute__((visibility("hidden"))) int RSDF_P(rsdf_coop_bwd)(int NT, const float *x7t, const float *planes, int n_levels, int n_active, float xyz_scale,
                  float xyz_offset, const float *w0, const float *b0, const float *w1, const float *b1, const float *w2,
                  int64_t n_samples, const float *d_sdf7t, const float *dh2c, float *d_planes, float *dw0, float *db0,
                  float *dw1, float *db1, float *dw2, float *db2, hipStream_t st, const unsigned *run_if);

// quad kernel (mlp_quad.hip): H = 64 backward on 16-feature wave tiles, two waves per SIMD
__attribute__((visibility("hidden"))) int RSDF_P(rsdf_quad_bwd)(int H, const float *x7t, const float *planes, int n_levels, int n_active, float xyz_scale, float xyz_offset,
                  const float *w0, const float *b0, const float *w1, const float *b1, const float *w2, int64_t n_samples,
                  const float *d_sdf7t, const float *dh2c, float *d_planes, float *dw0, float *db0, float *dw1, float *db1,
                  float *dw2, float *db2, hipStream_t st, const unsigned *run_if);

extern "C" {

int RSDF_P(rsdf_sdfmlp_fd7_supported)(int K0, int H, int N2)
{
    return (K0 >= 1 && K0 <= 35 && (H == 32 || H == 64 || H == 128) && N2 >= 1 && N2 <= 64) ? 1 : 0;
}

// Kernel selection.  Forward: the per-wave kernel of this file for H <= 64 (every wave holds all weights through LDS),
// the cooperative kernel for H = 128.  Backward: the cooperative kernel for every width.  RSDF_MLP_FWD=coop /
// RSDF_MLP_BWD=legacy force the other form where it exists (A/B timing, parity tests of both).
int RSDF_P(rsdf_sdfmlp_fd7_fwd)(const float *x7t, const float *planes, int n_levels, int n_active_levels,
                        float xyz_scale, float xyz_offset, int H, int N2, const float *w0,
                        const float *b0, const float *w1, const float *b1, const float *w2,
                        const float *b2, int64_t n_samples, float *sdf7t, float *feature, float *h2c,
                        void *stream)
{
    const int K0 = 3 + 2 * n_levels;
    RSDF_CHECK_ARG(n_levels >= 1 && n_levels <= 16, "sdfmlp_fd7_fwd: n_levels must be in [1,16]");
    RSDF_CHECK_ARG(h2c == nullptr || feature != nullptr, "sdfmlp_fd7_fwd: h2c needs feature");
    RSDF_CHECK_ARG(RSDF_P(rsdf_sdfmlp_fd7_supported)(K0, H, N2), "sdfmlp_fd7_fwd: unsupported layer sizes");
    if (n_samples <= 0) return 0;
    if (n_active_levels < 0 || n_active_levels > n_levels) n_active_levels = n_levels;
    hipStream_t st = (hipStream_t)stream;
    if (H == 128 || env_is("RSDF_MLP_FWD", "coop"))
        return RSDF_P(rsdf_coop_fwd)(H / 32, x7t, planes, n_levels, n_active_levels, xyz_scale, xyz_offset, N2, w0, b0, w1, b1,
                             w2, b2, n_samples, sdf7t, feature, h2c, st);
    const unsigned grid = persistent_grid(n_samples, FWD_WAVES);
    const TileSrc src{x7t, planes, n_samples, n_levels, n_active_levels, xyz_scale, xyz_offset};
    int rc;
    if (H == 64) {
        if ((rc = set_lds(sdfmlp_fwd_kernel<64>, fwd_lds<64>()))) return rc;
        sdfmlp_fwd_kernel<64><<<grid, FWD_THREADS, fwd_lds<64>(), st>>>(src, w0, b0, w1, b1, w2, b2, N2, sdf7t,
                                                                     feature, h2c);
    } else {
        if ((rc = set_lds(sdfmlp_fwd_kernel<32>, fwd_lds<32>()))) return rc;
        sdfmlp_fwd_kernel<32><<<grid, FWD_THREADS, fwd_lds<32>(), st>>>(src, w0, b0, w1, b1, w2, b2, N2, sdf7t,
                                                                     feature, h2c);
    }
    RSDF_RETURN_LAUNCH();
}

int RSDF_P(rsdf_sdfmlp_fd7_bwd)(const float *x7t, const float *planes, int n_levels, int n_active_levels,
                        float xyz_scale, float xyz_offset, int H, int N2, const float *w0,
                        const float *b0, const float *w1, const float *b1, const float *w2,
                        const float *b2, int64_t n_samples, const float *d_sdf7t, const float *d_feature,
                        float *dh2c_scratch, float *d_planes, float *dw0, float *db0, float *dw1, float *db1,
                        float *dw2, float *db2, void *stream)
{
    const int K0 = 3 + 2 * n_levels;
    RSDF_CHECK_ARG(n_levels >= 1 && n_levels <= 16, "sdfmlp_fd7_bwd: n_levels must be in [1,16]");
    RSDF_CHECK_ARG(RSDF_P(rsdf_sdfmlp_fd7_supported)(K0, H, N2), "sdfmlp_fd7_bwd: unsupported layer sizes");
    if (n_samples <= 0) return 0;
    if (n_active_levels < 0 || n_active_levels > n_levels) n_active_levels = n_levels;
    hipStream_t st = (hipStream_t)stream;
    if (!(H <= 64 && env_is("RSDF_MLP_BWD", "legacy"))) {
        if (d_feature != nullptr) {
            // d(h2) of the centre taps through the feature rows of the last layer: one per-layer product
            // [n, N2] x [N2, H]; the cooperative kernel adds the SDF row and continues the chain
            RSDF_CHECK_ARG(dh2c_scratch != nullptr, "sdfmlp_fd7_bwd: d_feature needs the [n, H] dh2c scratch");
            const int rc = RSDF_P(rsdf_linear_bwd_input)(d_feature, nullptr, N2, w2, n_samples, H, N2, RSDF_ACT_NONE, 0, H,
                                                 nullptr, dh2c_scratch, H, stream);
            if (rc) return rc;
        }
        if (H == 64 && !env_is("RSDF_MLP_BWD", "coop"))
            return RSDF_P(rsdf_quad_bwd)(H, x7t, planes, n_levels, n_active_levels, xyz_scale, xyz_offset, w0, b0, w1, b1, w2,
                                 n_samples, d_sdf7t, d_feature != nullptr ? dh2c_scratch : nullptr, d_planes, dw0, db0,
                                 dw1, db1, dw2, db2, st, nullptr);
        return RSDF_P(rsdf_coop_bwd)(H / 32, x7t, planes, n_levels, n_active_levels, xyz_scale, xyz_offset, w0, b0, w1, b1, w2,
                             n_samples, d_sdf7t, d_feature != nullptr ? dh2c_scratch : nullptr, d_planes, dw0, db0, dw1,
                             db1, dw2, db2, st, nullptr);
    }
    const unsigned grid = persistent_grid(n_samples, BWD_WAVES);
    const TileSrc src{x7t, planes, n_samples, n_levels, n_active_levels, xyz_scale, xyz_offset};
    // k-steps (of 16 output channels) of the feature-gradient product staged in LDS
    const int steps2 = d_feature != nullptr ? (N2 + 15) / 16 : 0;
    int rc;
    if (H == 64) {
        if ((rc = set_lds(sdfmlp_bwd_kernel<64>, bwd_lds<64>(steps2)))) return rc;
        sdfmlp_bwd_kernel<64><<<grid, BWD_THREADS, bwd_lds<64>(steps2), st>>>(src, w0, b0, w1, b1, w2, b2, N2, steps2,
                                                                               d_sdf7t, d_feature, d_planes, dw0,
                                                                               db0, dw1, db1, dw2, db2);
    } else {
        if ((rc = set_lds(sdfmlp_bwd_kernel<32>, bwd_lds<32>(steps2)))) return rc;
        sdfmlp_bwd_kernel<32><<<grid, BWD_THREADS, bwd_lds<32>(steps2), st>>>(src, w0, b0, w1, b1, w2, b2, N2, steps2,
                                                                               d_sdf7t, d_feature, d_planes, dw0,
                                                                               db0, dw1, db1, dw2, db2);
    }
    RSDF_RETURN_LAUNCH();
}

}  // extern "C"
