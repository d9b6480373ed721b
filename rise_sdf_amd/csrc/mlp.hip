// H3 / T3: VanillaMLP layers (models/network_utils.py:109-157), one kernel per layer and direction.
//
// The reference MLPs are plain fp32 nn.Linear (+ weight_norm) with Softplus(beta=100) or ReLU, run
// as cuBLAS GEMMs plus separate activation kernels.  Here:
//   linear_fwd        y = act(x W^T + b)                 split-bf16 matrix products (fp32-equivalent, split_bf16.h)
//   linear_bwd_input  dz = dy act'(y), dx = dz W[:, win]  split-bf16
//   linear_bwd_weight dW += dz^T x, db += colsum(dz)     fp32 MFMA (sum over rows)
// K and N are at most 128, so the whole weight matrix lives in LDS: forward and input-gradient kernels are
// PERSISTENT (<= 512 workgroups loop over 32-row tiles), stage the weights once per workgroup as split,
// fragment-ordered bf16 and compute TRANSPOSED, D[n][row] = sum_k W[n][k] x[row][k]: the x fragment of a lane
// is 8 consecutive floats of its own row (read straight from global memory, no LDS tile) and an accumulator
// register quad is 4 consecutive output features of that row (one 16-byte store).
// Operand maps: split_bf16.h / cdna_hip_programming.md section 3.
#include "common.h"
#include "split_bf16.h"
#include "act.h"

namespace {


// ------------------------------------------------------------------------------------------------
// split-bf16 per-layer kernels
// ------------------------------------------------------------------------------------------------
constexpr int S_WAVES = 8;                 // two waves per SIMD
constexpr int S_THREADS = S_WAVES * 64;
// Column steps whose loads are issued before the first is consumed (round 4; tools/bench_linear.py, 8 M rows, kernels alone):
// the row loops are bound by load latency at two to four waves per SIMD, not by the matrix pipe (28 % busy at 128 x 128).
//   forward     1 -> 4 steps:  35x64 1.11 -> 0.99 ms, 64x64 1.17 -> 0.99, 84x128 2.53 -> 1.97, 128x128 2.49 -> 2.35, 128x6 0.96 -> 0.87
//   input grad  1 -> 4 steps:  35x64 2.38 -> 2.18, 64x64 2.11 -> 1.92, 64x49 (unaligned rows) 4.80 -> 3.16, 84x128 4.06 -> 3.64
// (8 steps in the forward: no further gain; the wider variants run at one workgroup per CU and are still faster.)
#ifndef RSDF_LIN_KSU
#define RSDF_LIN_KSU 4
#endif
#ifndef RSDF_LIN_KSU_BI
#define RSDF_LIN_KSU_BI 4
#endif
// Weight gradient of a layer with K <= 64 input columns: capped at 128 registers (24 bytes of scratch) so that two
// workgroups share a CU: 1.14 -> 0.83 ms (35x64), 1.19 -> 0.91 (64x64).  The wide instantiations spill badly under the same
// cap (10-15 ms) and keep one workgroup per CU.
#ifndef RSDF_BWDW_OCC
#define RSDF_BWDW_OCC 4
#endif

// LDS weight image (16-byte units): [part 3][out tile OT][k-step KS][hf 2][c 32], then OT*32 floats of bias
__device__ __forceinline__ f32x16 mma6s(const u32x4 *__restrict__ wa, int part_stride, const Frag3 &b, f32x16 c)
{
    if (!RSDF_SPLIT3) return mma_bf16(wa[0], b.h, c);
    const u32x4 ah = wa[0], am = wa[part_stride], al = wa[2 * part_stride];
    c = mma_bf16(al, b.h, c);
    c = mma_bf16(ah, b.l, c);
    c = mma_bf16(am, b.m, c);
    c = mma_bf16(am, b.h, c);
    c = mma_bf16(ah, b.m, c);
    c = mma_bf16(ah, b.h, c);
    return c;
}

// 8 consecutive floats p[k .. k+8) of a row, zero beyond `limit`; vector loads when the row allows it
__device__ __forceinline__ void load8(const float *__restrict__ p, int k, int limit, bool vec, float (&v)[8])
{
    if (vec && k + 8 <= limit) {
        const float4 a = *reinterpret_cast<const float4 *>(p + k), b = *reinterpret_cast<const float4 *>(p + k + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (k + j < limit) ? p[k + j] : 0.0f;
    }
}

// y = act(x W^T + b)
template <int NT>
__global__ void __launch_bounds__(S_THREADS)
linear_fwd_kernel(const float *__restrict__ x, int ldx, const float *__restrict__ w,
                  const float *__restrict__ b, int64_t n, int K, int N, int act,
                  float *__restrict__ y, int ldy)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int KS = (K + 15) >> 4;
    const int part = NT * KS * 2 * 32;
    unsigned short *e16 = reinterpret_cast<unsigned short *>(smem_b);
    float *bias = reinterpret_cast<float *>(smem_b + (size_t)3 * part * 16);
    // A[i = n][k natural] : element e = (((nt*KS + ks)*2 + hf)*32 + c)*8 + j
    for (int e = threadIdx.x; e < part * 8; e += S_THREADS) {
        const int j = e & 7, c = (e >> 3) & 31, hf = (e >> 8) & 1, ks = (e >> 9) % KS, nt = (e >> 9) / KS;
        const int nn = 32 * nt + c, k = 16 * ks + 8 * hf + j;
        store3(e16, (size_t)part * 8, e, (nn < N && k < K) ? w[nn * K + k] : 0.0f);
    }
    for (int e = threadIdx.x; e < NT * 32; e += S_THREADS) bias[e] = (b && e < N) ? b[e] : 0.0f;
    __syncthreads();
    const u32x4 *wl = reinterpret_cast<const u32x4 *>(smem_b);

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, hf = lane >> 5;
    const bool vx = (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const bool vy = (ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
    const int64_t n_tiles = (n + 31) >> 5;
    for (int64_t tile = (int64_t)blockIdx.x * S_WAVES + wave; tile < n_tiles; tile += (int64_t)gridDim.x * S_WAVES) {
        const int64_t row = tile * 32 + c;
        const bool ok = row < n;
        const float *xr = x + (ok ? row : 0) * (int64_t)ldx;
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = bias[32 * t + (r & 3) + 8 * (r >> 2) + 4 * hf];
        // RSDF_LIN_KSU k-steps of the row are requested before the first of them is consumed: the wave holds KSU x 32 bytes
        // per lane in flight instead of 32 (the loop is bound by load latency at two workgroups per CU; a k-step beyond K
        // issues no load)
        for (int ks0 = 0; ks0 < KS; ks0 += RSDF_LIN_KSU) {
            float v[RSDF_LIN_KSU][8];
#pragma unroll
            for (int u = 0; u < RSDF_LIN_KSU; ++u) load8(xr, 16 * (ks0 + u) + 8 * hf, ok ? K : 0, vx, v[u]);
#pragma unroll
            for (int u = 0; u < RSDF_LIN_KSU; ++u) {
                const int ks = ks0 + u;
                if (ks < KS) {   // (uniform)
                    const Frag3 xb = split_frag(v[u][0], v[u][1], v[u][2], v[u][3], v[u][4], v[u][5], v[u][6], v[u][7]);
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[t] = mma6s(wl + ((t * KS + ks) * 2 + hf) * 32 + c, part, xb, acc[t]);
                }
            }
        }
        if (!ok) continue;
        float *yr = y + row * (int64_t)ldy;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = 32 * t + 8 * g + 4 * hf;   // registers 4g .. 4g+3 = features col .. col+3
                float o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = act_fwd(acc[t][4 * g + q], act);
                if (vy && col + 4 <= N) {
                    *reinterpret_cast<float4 *>(yr + col) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (col + q < N) yr[col + q] = o[q];
                }
            }
    }
}

// dz = dy * act'(y);  dx[:, 0:Kout] = dz @ W[:, k0:k0+Kout]
template <int JT>
__global__ void __launch_bounds__(S_THREADS)
linear_bwd_input_kernel(const float *dy, const float *__restrict__ y, int lddy,
                        const float *__restrict__ w, int64_t n, int K, int N, int act, int k0,
                        int Kout, float *dz, float *__restrict__ dx, int lddx)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int NS = (N + 15) >> 4;
    const int part = JT * NS * 2 * 32;
    if (dx) {
        // A[i = window column][k = n natural] = W[n][k0 + column]
        unsigned short *e16 = reinterpret_cast<unsigned short *>(smem_b);
        for (int e = threadIdx.x; e < part * 8; e += S_THREADS) {
            const int j = e & 7, c = (e >> 3) & 31, hf = (e >> 8) & 1, ns = (e >> 9) % NS, jt = (e >> 9) / NS;
            const int col = 32 * jt + c, nn = 16 * ns + 8 * hf + j;
            store3(e16, (size_t)part * 8, e, (col < Kout && nn < N) ? w[nn * K + k0 + col] : 0.0f);
        }
        __syncthreads();
    }
    const u32x4 *wl = reinterpret_cast<const u32x4 *>(smem_b);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, hf = lane >> 5;
    const bool vz = (lddy & 3) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 &&
                    (y == nullptr || (reinterpret_cast<uintptr_t>(y) & 15) == 0) &&
                    (dz == nullptr || (reinterpret_cast<uintptr_t>(dz) & 15) == 0);
    const bool vx = dx != nullptr && (lddx & 3) == 0 && (reinterpret_cast<uintptr_t>(dx) & 15) == 0;
    const int64_t n_tiles = (n + 31) >> 5;
    for (int64_t tile = (int64_t)blockIdx.x * S_WAVES + wave; tile < n_tiles; tile += (int64_t)gridDim.x * S_WAVES) {
        const int64_t row = tile * 32 + c;
        const bool ok = row < n;
        const int64_t rbase = (ok ? row : 0) * (int64_t)lddy;
        f32x16 acc[JT];
#pragma unroll
        for (int t = 0; t < JT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        // RSDF_LIN_KSU_BI column steps are requested before the first is consumed (dz may alias dy, so the compiler cannot
        // move a step's loads above the previous step's stores itself; a step only stores the columns it has loaded)
        for (int ns0 = 0; ns0 < NS; ns0 += RSDF_LIN_KSU_BI) {
            float vv[RSDF_LIN_KSU_BI][8], yy[RSDF_LIN_KSU_BI][8];
#pragma unroll
            for (int u = 0; u < RSDF_LIN_KSU_BI; ++u) {
                const int kk = 16 * (ns0 + u) + 8 * hf;
                load8(dy + rbase, kk, ok ? N : 0, vz, vv[u]);
                if (act != RSDF_ACT_NONE) load8(y + rbase, kk, ok ? N : 0, vz, yy[u]);
            }
#pragma unroll
            for (int u = 0; u < RSDF_LIN_KSU_BI; ++u) {
                const int ns = ns0 + u;
                if (ns >= NS) break;   // (uniform)
                const int kk = 16 * ns + 8 * hf;
                float (&v)[8] = vv[u];
                if (act != RSDF_ACT_NONE) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] *= act_bwd_from_y(yy[u][j], act);
                }
                if (dz != nullptr && ok) {
                    if (vz && kk + 8 <= N) {
                        *reinterpret_cast<float4 *>(dz + rbase + kk) = make_float4(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<float4 *>(dz + rbase + kk + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (kk + j < N) dz[rbase + kk + j] = v[j];
                    }
                }
                if (dx) {
                    const Frag3 zb = split_frag(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
#pragma unroll
                    for (int t = 0; t < JT; ++t)
                        acc[t] = mma6s(wl + ((t * NS + ns) * 2 + hf) * 32 + c, part, zb, acc[t]);
                }
            }
        }
        if (!dx || !ok) continue;
        float *xr = dx + row * (int64_t)lddx;
#pragma unroll
        for (int t = 0; t < JT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = 32 * t + 8 * g + 4 * hf;
                if (vx && col + 4 <= Kout) {
                    *reinterpret_cast<float4 *>(xr + col) =
                        make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (col + q < Kout) xr[col + q] = acc[t][4 * g + q];
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// dw[N,K] += dz^T x ; db[N] += colsum(dz)          (split-bf16, sum over rows = the MFMA k dimension)
// A[i = n][k = row] and B[k = row][j = input column]: a lane's fragment is 8 consecutive ROWS of its own
// column, so for each of the 8 rows the 32 lanes of a half read 128 contiguous bytes -- operands come straight
// from global memory, no transpose.  Wave w of a workgroup owns output n-tile w % NT (all KT k-tiles, in
// registers for the whole kernel) and row slab w / NT; workgroups are persistent and flush once with atomics.
// ------------------------------------------------------------------------------------------------
template <int KT>
__global__ void __launch_bounds__(S_THREADS, KT <= 2 ? RSDF_BWDW_OCC : 1)
linear_bwd_weight_kernel(const float *__restrict__ dz, int lddz, const float *__restrict__ x,
                         int ldx, int64_t n, int K, int N, float *__restrict__ dw,
                         float *__restrict__ db)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, hf = lane >> 5;
    const int NT = (N + 31) >> 5;
    const int slabs = S_WAVES / NT;              // row slabs per workgroup (waves beyond NT * slabs idle)
    const int nt = wave % NT, slab = wave / NT;
    if (slab >= slabs) return;
    f32x16 acc[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float bsum = 0.0f;
    const int ncol = nt * 32 + c;
    const bool nok = ncol < N;
    const int64_t n_steps = (n + 15) >> 4;       // 16 rows per MFMA k-step
    const int64_t stride = (int64_t)gridDim.x * slabs;
    for (int64_t step = (int64_t)blockIdx.x * slabs + slab; step < n_steps; step += stride) {
        const int64_t r0 = step * 16 + 8 * hf;
        float av[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) av[j] = (nok && r0 + j < n) ? dz[(r0 + j) * lddz + ncol] : 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum += av[j];
        const Frag3 af = split_frag(av[0], av[1], av[2], av[3], av[4], av[5], av[6], av[7]);
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const int kcol = t * 32 + c;
            float bv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) bv[j] = (kcol < K && r0 + j < n) ? x[(r0 + j) * ldx + kcol] : 0.0f;
            acc[t] = mma6r(af, split_frag(bv[0], bv[1], bv[2], bv[3], bv[4], bv[5], bv[6], bv[7]), acc[t]);
        }
    }
    // The row slabs of the workgroup meet in LDS, slab by slab, and slab 0 flushes: one add per workgroup and address.  A
    // narrow output layer (N <= 4: eight slabs, 512 workgroups) used to send 4096 float atomics to each of its few dw / db
    // addresses per launch; same-address atomics execute one after the other at the memory side, which was most of this
    // kernel's duration at a training step's 250 k rows (see also mlp_layer_bwd.hip).
    extern __shared__ float s_red[];                       // [NT][KT][16][64] sums, then [NT][32] bias sums
    float *mine = s_red + (size_t)nt * KT * 1024 + lane;
    float *bias = s_red + (size_t)NT * KT * 1024 + nt * 32;
    bsum += __shfl_xor(bsum, 32, 64);
    for (int sl = 0; sl < slabs; ++sl) {
        if (slab == sl) {
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float *p = mine + (t * 16 + r) * 64;
                    *p = sl == 0 ? acc[t][r] : *p + acc[t][r];
                }
            if (hf == 0) bias[c] = sl == 0 ? bsum : bias[c] + bsum;
        }
        __syncthreads();
    }
    if (slab != 0) return;
    // D[i = n within tile][j = k within tile]
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const int kcol = t * 32 + c;
        if (kcol < K) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nrow = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
                const float v = mine[(t * 16 + r) * 64];
                if (nrow < N && v != 0.0f) atomicAdd(&dw[nrow * K + kcol], v);
            }
        }
    }
    if (db && hf == 0 && nok && bias[c] != 0.0f) atomicAdd(&db[ncol], bias[c]);
}

// ------------------------------------------------------------------------------------------------
// dW / db of a NARROW layer (N <= 8 output columns: the 1-6 column output layers of the radiance networks,
// models/texture.py:237-327) on the vector ALU (round 4).  A 32-wide MFMA tile is 4-30x too wide for these layers, and
// the kernel above spends its time in 40 strided scalar loads per 16 rows; here a thread owns one input column k and
// walks the rows with coalesced x reads (a wave reads 256 contiguous bytes of a row; the <= 8 dz values of the row are
// the same address for every lane), N fp32 FMAs per row.  250 k rows x 128 columns: 0.25 -> ~0.05 ms per layer, six such
// layers per training step.  bf16 build: both operands rounded once to bf16, as the matrix form does.
// ------------------------------------------------------------------------------------------------
constexpr int NARROW_N = 8;
__device__ __forceinline__ float round_operand(float v)
{
#ifdef RSDF_BF16
    return bf16_lo(pack_bf16(v, 0.0f));
#else
    return v;
#endif
}
template <int NN>
__global__ void __launch_bounds__(256)
linear_bwd_weight_narrow_kernel(const float *__restrict__ dz, int lddz, const float *__restrict__ x, int ldx, int64_t n, int K,
                                int N, int KP, float *__restrict__ dw, float *__restrict__ db)
{
    __shared__ float s_acc[256 * NN];
    const int k = threadIdx.x & (KP - 1), rl = threadIdx.x / KP, RL = 256 / KP;       // KP: power of two >= K, <= 256
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < n ? r0 + per : n;
    float acc[NN], bs[NN];
#pragma unroll
    for (int j = 0; j < NN; ++j) acc[j] = bs[j] = 0.0f;
    const bool kok = k < K;
    int64_t r = r0 + rl;
    for (; r + 3 * RL < r1; r += 4 * RL) {          // four rows per trip: their loads are in flight together
        float xv[4], d[4][NN];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xv[u] = kok ? x[(r + u * RL) * ldx + k] : 0.0f;
#pragma unroll
            for (int j = 0; j < NN; ++j) d[u][j] = j < N ? dz[(r + u * RL) * lddz + j] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < NN; ++j) {
                acc[j] = fmaf(round_operand(d[u][j]), round_operand(xv[u]), acc[j]);
                bs[j] += d[u][j];
            }
    }
    for (; r < r1; r += RL) {
        const float xv = kok ? round_operand(x[r * ldx + k]) : 0.0f;
#pragma unroll
        for (int j = 0; j < NN; ++j)
            if (j < N) {
                const float d = dz[r * lddz + j];
                acc[j] = fmaf(round_operand(d), xv, acc[j]);
                bs[j] += d;
            }
    }
    // the RL row lanes of a column meet in LDS; one atomic per workgroup and address
#pragma unroll
    for (int j = 0; j < NN; ++j) s_acc[j * 256 + threadIdx.x] = acc[j];
    __syncthreads();
    if (rl == 0 && kok) {
#pragma unroll
        for (int j = 0; j < NN; ++j)
            if (j < N) {
                float v = 0.0f;
                for (int q = 0; q < RL; ++q) v += s_acc[j * 256 + q * KP + k];
                if (v != 0.0f) atomicAdd(&dw[j * K + k], v);
            }
    }
    if (db != nullptr) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NN; ++j) s_acc[j * 256 + threadIdx.x] = (k == 0) ? bs[j] : 0.0f;
        __syncthreads();
        if (threadIdx.x < N) {
            float v = 0.0f;
            for (int q = 0; q < RL; ++q) v += s_acc[threadIdx.x * 256 + q * KP];
            if (v != 0.0f) atomicAdd(&db[threadIdx.x], v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// weight_norm (dim=0): w = v * (g / ||v||_row)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
weight_norm_fwd_kernel(const float *__restrict__ g, const float *__restrict__ v, int K,
                       float *__restrict__ w)
{
    const int r = blockIdx.x, lane = threadIdx.x;
    float ss = 0.0f;
    for (int k = lane; k < K; k += 64) { const float t = v[r * K + k]; ss = fmaf(t, t, ss); }
    const float s = g[r] / sqrtf(wave_sum(ss));
    for (int k = lane; k < K; k += 64) w[r * K + k] = v[r * K + k] * s;
}

// dg = <dw, v>/||v|| ;  dv = (g/||v||) dw - (g <dw,v>/||v||^3) v
__global__ void __launch_bounds__(64)
weight_norm_bwd_kernel(const float *__restrict__ g, const float *__restrict__ v,
                       const float *__restrict__ dw, int K, float *__restrict__ dg,
                       float *__restrict__ dv)
{
    const int r = blockIdx.x, lane = threadIdx.x;
    float ss = 0.0f, dot = 0.0f;
    for (int k = lane; k < K; k += 64) {
        const float t = v[r * K + k];
        ss = fmaf(t, t, ss);
        dot = fmaf(dw[r * K + k], t, dot);
    }
    ss = wave_sum(ss);
    dot = wave_sum(dot);
    const float nrm = sqrtf(ss);
    const float gr = g[r];
    if (lane == 0) dg[r] = dot / nrm;
    const float c1 = gr / nrm, c2 = gr * dot / (nrm * ss);
    for (int k = lane; k < K; k += 64) dv[r * K + k] = c1 * dw[r * K + k] - c2 * v[r * K + k];
}

size_t fwd_lds_bytes(int K, int NT)
{
    const int KS = (K + 15) >> 4;
    return (size_t)3 * NT * KS * 2 * 32 * 16 + (size_t)NT * 32 * sizeof(float);
}
size_t bwd_lds_bytes(int N, int JT)
{
    const int NS = (N + 15) >> 4;
    return (size_t)3 * JT * NS * 2 * 32 * 16;
}
unsigned persistent_blocks(int64_t n)
{
    const int64_t want = ((n + 31) / 32 + S_WAVES - 1) / S_WAVES;
    return (unsigned)(want < 1 ? 1 : (want > 512 ? 512 : want));
}

template <typename Kern>
int allow_lds(Kern kern, size_t bytes)
{
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) { rsdf_set_error(hipGetErrorString(e)); return (int)e; }
    }
    return 0;
}

}  // namespace

extern "C" {

int RSDF_P(rsdf_linear_fwd)(const float *x, int ldx, const float *w, const float *b, int64_t n, int K, int N,
                    int act, float *y, int ldy, void *stream)
{
    RSDF_CHECK_ARG(K >= 1 && K <= 128 && N >= 1 && N <= 128, "linear_fwd: K and N must be in [1,128]");
    RSDF_CHECK_ARG(ldx >= K && ldy >= N, "linear_fwd: row stride smaller than the row");
    if (n <= 0) return 0;
    const int NT = (N + 31) / 32;
    const size_t lds = fwd_lds_bytes(K, NT);
    const unsigned grid = persistent_blocks(n);
    hipStream_t st = (hipStream_t)stream;
    int rc = 0;
#define LAUNCH_FWD(NT_)                                                                        \
    rc = allow_lds(linear_fwd_kernel<NT_>, lds);                                               \
    if (rc) return rc;                                                                         \
    linear_fwd_kernel<NT_><<<grid, S_THREADS, lds, st>>>(x, ldx, w, b, n, K, N, act, y, ldy)
    switch (NT) {
    case 1: LAUNCH_FWD(1); break;
    case 2: LAUNCH_FWD(2); break;
    case 3: LAUNCH_FWD(3); break;
    default: LAUNCH_FWD(4); break;
    }
#undef LAUNCH_FWD
    RSDF_RETURN_LAUNCH();
}

int RSDF_P(rsdf_linear_bwd_input)(const float *dy, const float *y, int lddy, const float *w, int64_t n, int K,
                          int N, int act, int k0, int Kout, float *dz, float *dx, int lddx,
                          void *stream)
{
    RSDF_CHECK_ARG(K >= 1 && K <= 128 && N >= 1 && N <= 128, "linear_bwd_input: K and N must be in [1,128]");
    RSDF_CHECK_ARG(lddy >= N, "linear_bwd_input: lddy < N");
    if (n <= 0) return 0;
    RSDF_CHECK_ARG(act == RSDF_ACT_NONE || y != nullptr, "linear_bwd_input: activation needs y");
    if (dx) {
        RSDF_CHECK_ARG(k0 >= 0 && Kout >= 1 && k0 + Kout <= K, "linear_bwd_input: bad column window");
        RSDF_CHECK_ARG(lddx >= Kout, "linear_bwd_input: lddx < Kout");
    }
    if (n <= 0) return 0;
    const int JT = dx ? (Kout + 31) / 32 : 1;
    const size_t lds = bwd_lds_bytes(N, JT);
    const unsigned grid = persistent_blocks(n);
    hipStream_t st = (hipStream_t)stream;
    int rc = 0;
#define LAUNCH_BI(JT_)                                                                         \
    rc = allow_lds(linear_bwd_input_kernel<JT_>, lds);                                         \
    if (rc) return rc;                                                                         \
    linear_bwd_input_kernel<JT_><<<grid, S_THREADS, lds, st>>>(dy, y, lddy, w, n, K, N, act, k0, \
                                                             Kout, dz, dx, lddx)
    switch (JT) {
    case 1: LAUNCH_BI(1); break;
    case 2: LAUNCH_BI(2); break;
    case 3: LAUNCH_BI(3); break;
    default: LAUNCH_BI(4); break;
    }
#undef LAUNCH_BI
    RSDF_RETURN_LAUNCH();
}

int RSDF_P(rsdf_linear_bwd_weight)(const float *dz, int lddz, const float *x, int ldx, int64_t n, int K, int N,
                           float *dw, float *db, void *stream)
{
    RSDF_CHECK_ARG(K >= 1 && K <= 128 && N >= 1 && N <= 128, "linear_bwd_weight: K and N must be in [1,128]");
    RSDF_CHECK_ARG(lddz >= N && ldx >= K, "linear_bwd_weight: row stride smaller than the row");
    if (n <= 0) return 0;
    // (up to 2^20 rows: a training step's 250 k.  On a whole render chunk -- millions of rows -- the matrix form's persistent
    //  workgroups stream better: 441 against 276 ms per c2 step when this form took those launches too)
    if (N <= NARROW_N && n <= (1 << 20) && !rsdf_env_is("RSDF_BWD_WEIGHT", "mfma")) {
        int KP = 32;
        while (KP < K) KP <<= 1;
        const int64_t rows_per_wg = 512;             // >= 2 rows per thread even at KP = 256
        int64_t g = (n + rows_per_wg - 1) / rows_per_wg;
        g = g < 1 ? 1 : (g > 1024 ? 1024 : g);
        linear_bwd_weight_narrow_kernel<NARROW_N><<<(unsigned)g, 256, 0, (hipStream_t)stream>>>(dz, lddz, x, ldx, n, K, N, KP, dw, db);
        RSDF_RETURN_LAUNCH();
    }
    const int KT = (K + 31) / 32;
    const int slabs = S_WAVES / ((N + 31) / 32);
    int64_t want = ((n + 15) / 16 + slabs - 1) / slabs;
    const unsigned grid = (unsigned)(want < 1 ? 1 : (want > 512 ? 512 : want));
    hipStream_t st = (hipStream_t)stream;
    const int NT = (N + 31) / 32;
    const size_t lds = ((size_t)NT * KT * 1024 + (size_t)NT * 32) * sizeof(float);    // <= 64.5 KB
#define RSDF_BWDW(KTV)                                                                                              \
    {                                                                                                               \
        if (lds > 48 * 1024)                                                                                        \
            if (int rc = rsdf_func_lds(reinterpret_cast<const void *>(linear_bwd_weight_kernel<KTV>), lds)) return rc; \
        linear_bwd_weight_kernel<KTV><<<grid, S_THREADS, lds, st>>>(dz, lddz, x, ldx, n, K, N, dw, db);              \
    }
    switch (KT) {
    case 1: RSDF_BWDW(1) break;
    case 2: RSDF_BWDW(2) break;
    case 3: RSDF_BWDW(3) break;
    default: RSDF_BWDW(4) break;
    }
#undef RSDF_BWDW
    RSDF_RETURN_LAUNCH();
}

#ifndef RSDF_BF16   // precision-independent: only in the default build
int rsdf_weight_norm_fwd(const float *g, const float *v, int N, int K, float *w, void *stream)
{
    if (N <= 0) return 0;
    weight_norm_fwd_kernel<<<N, 64, 0, (hipStream_t)stream>>>(g, v, K, w);
    RSDF_RETURN_LAUNCH();
}

int rsdf_weight_norm_bwd(const float *g, const float *v, const float *dw, int N, int K, float *dg,
                         float *dv, void *stream)
{
    if (N <= 0) return 0;
    weight_norm_bwd_kernel<<<N, 64, 0, (hipStream_t)stream>>>(g, v, dw, K, dg, dv);
    RSDF_RETURN_LAUNCH();
}
#endif

}  // extern "C"
