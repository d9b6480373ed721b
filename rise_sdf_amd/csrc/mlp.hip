// H3: VanillaMLP layers (models/network_utils.py:109-157) on the fp32 matrix cores.
//
// The reference MLPs are plain fp32 nn.Linear (+ weight_norm) with Softplus(beta=100) or ReLU, run
// as cuBLAS GEMMs plus separate activation kernels.  Here each layer is one kernel built on
// v_mfma_f32_32x32x2_f32: exact fp32 (a k-ordered fmaf chain seeded with the bias), bias and
// activation fused into the epilogue, the backward's activation derivative fused into the operand
// staging.  K and N are at most 128, so the whole weight matrix lives in LDS for the block.
//
// Operand maps of v_mfma_f32_32x32x2_f32 (cdna_hip_programming.md section 3), lane l:
//   A[i = l&31][k = l>>5],  B[k = l>>5][j = l&31],
//   C/D register r: row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31.
// LDS tiles are row-major with an odd row stride (KP+1), so the 32 lanes of a ds_read_b32 group
// (32 consecutive rows, same column) fall on 32 distinct banks.
#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int THREADS = 256;
constexpr int WAVES = THREADS / 64;
constexpr int ROWS_PER_WAVE = 32;
constexpr int ROWS_PER_BLOCK = WAVES * ROWS_PER_WAVE;  // 128

__device__ __forceinline__ float act_fwd(float z, int act)
{
    switch (act) {
    case RSDF_ACT_RELU: return fmaxf(z, 0.0f);
    case RSDF_ACT_SOFTPLUS100: {
        const float t = z * 100.0f;
        return t > 20.0f ? z : log1pf(expf(t)) / 100.0f;
    }
    case RSDF_ACT_SIGMOID: return 1.0f / (1.0f + expf(-z));
    default: return z;
    }
}

// derivative expressed through the OUTPUT y = act(z), so the forward only has to keep y
__device__ __forceinline__ float act_bwd_from_y(float y, int act)
{
    switch (act) {
    case RSDF_ACT_RELU: return y > 0.0f ? 1.0f : 0.0f;
    case RSDF_ACT_SOFTPLUS100:
        // y = log(1+e^{100 z})/100  =>  sigmoid(100 z) = 1 - e^{-100 y}
        return -expm1f(-100.0f * y);
    case RSDF_ACT_SIGMOID: return y * (1.0f - y);
    default: return 1.0f;
    }
}

// W[N,K] (row-major, global) -> LDS [NP][ldw], zero padded
__device__ __forceinline__ void stage_weights(const float *__restrict__ w, int N, int K, int NP,
                                              int KP, int ldw, float *Ws)
{
    for (int e = threadIdx.x; e < NP * KP; e += THREADS) {
        const int r = e / KP, c = e - r * KP;
        Ws[r * ldw + c] = (r < N && c < K) ? w[r * K + c] : 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// y = act(x W^T + b)
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(THREADS)
linear_fwd_kernel(const float *__restrict__ x, int ldx, const float *__restrict__ w,
                  const float *__restrict__ b, int64_t n, int K, int N, int act,
                  float *__restrict__ y, int ldy)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int KP = (K + 1) & ~1, ld = KP + 1;
    float *Ws = smem;                                   // [NT*32][ld]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *Xs = smem + NT * 32 * ld + wave * ROWS_PER_WAVE * ld;  // [32][ld], private to the wave

    stage_weights(w, N, K, NT * 32, KP, ld, Ws);

    const int64_t row0 = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wave * ROWS_PER_WAVE;
    for (int e = lane; e < ROWS_PER_WAVE * KP; e += 64) {
        const int r = e / KP, c = e - r * KP;
        const int64_t gr = row0 + r;
        Xs[r * ld + c] = (gr < n && c < K) ? x[gr * ldx + c] : 0.0f;
    }
    __syncthreads();

    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = t * 32 + li;
        const float bv = (b && col < N) ? b[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = bv;
    }
    for (int k0 = 0; k0 < KP; k0 += 2) {
        const float a = Xs[li * ld + k0 + lh];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float bb = Ws[(t * 32 + li) * ld + k0 + lh];
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = t * 32 + li;
        if (col < N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t gr = row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (gr < n) y[gr * ldy + col] = act_fwd(acc[t][r], act);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dz = dy * act'(y);  dx[:, 0:Kout] = dz @ W[:, k0:k0+Kout]
// ------------------------------------------------------------------------------------------------
template <int JT>
__global__ void __launch_bounds__(THREADS)
linear_bwd_input_kernel(const float *dy, const float *__restrict__ y, int lddy,
                        const float *__restrict__ w, int64_t n, int K, int N, int act, int k0,
                        int Kout, float *dz, float *__restrict__ dx, int lddx)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int KP = (K + 1) & ~1, ldw = KP + 1;
    const int NP = (N + 1) & ~1, ldz = NP + 1;
    const int NR = (N + 31) & ~31;
    float *Ws = smem;  // [NR][ldw]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *Zs = smem + NR * ldw + wave * ROWS_PER_WAVE * ldz;  // [32][ldz]

    if (dx) stage_weights(w, N, K, NR, KP, ldw, Ws);

    const int64_t row0 = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wave * ROWS_PER_WAVE;
    for (int e = lane; e < ROWS_PER_WAVE * NP; e += 64) {
        const int r = e / NP, c = e - r * NP;
        const int64_t gr = row0 + r;
        float v = 0.0f;
        if (gr < n && c < N) {
            v = dy[gr * lddy + c];
            if (act != RSDF_ACT_NONE) v *= act_bwd_from_y(y[gr * lddy + c], act);
            if (dz) dz[gr * lddy + c] = v;
        }
        Zs[r * ldz + c] = v;
    }
    __syncthreads();
    if (!dx) return;

    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[JT];
#pragma unroll
    for (int t = 0; t < JT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    for (int n0 = 0; n0 < NP; n0 += 2) {
        const float a = Zs[li * ldz + n0 + lh];
#pragma unroll
        for (int t = 0; t < JT; ++t) {
            const int col = k0 + t * 32 + li;  // < KP guaranteed by the launcher's padding rule
            const float bb = (col < KP) ? Ws[(n0 + lh) * ldw + col] : 0.0f;
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < JT; ++t) {
        const int col = t * 32 + li;
        if (col < Kout) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t gr = row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (gr < n) dx[gr * lddx + col] = acc[t][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dw[N,K] += dz^T x ; db[N] += colsum(dz)
// Each block owns SLAB rows; wave w owns rows [w*SLAB/4, (w+1)*SLAB/4) and walks the n-tiles one at
// a time, holding the KT k-tiles of that n-tile in accumulators.  Both MFMA operands are read
// straight from global memory: lane l reads element (row s0 + (l>>5), column tile*32 + (l&31)),
// i.e. two coalesced 128-byte row segments per instruction.
// ------------------------------------------------------------------------------------------------
constexpr int SLAB = 4096;

template <int KT>
__global__ void __launch_bounds__(THREADS)
linear_bwd_weight_kernel(const float *__restrict__ dz, int lddz, const float *__restrict__ x,
                         int ldx, int64_t n, int K, int N, float *__restrict__ dw,
                         float *__restrict__ db)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int64_t r_begin = (int64_t)blockIdx.x * SLAB + wave * (SLAB / WAVES);
    int64_t r_end = r_begin + SLAB / WAVES;
    if (r_end > n) r_end = n;
    if (r_begin >= n) return;
    const int NT = (N + 31) >> 5;
    for (int nt = 0; nt < NT; ++nt) {
        f32x16 acc[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        float bsum = 0.0f;
        const int ncol = nt * 32 + li;
        const bool nok = ncol < N;
        for (int64_t s0 = r_begin; s0 < r_end; s0 += 2) {
            const int64_t s = s0 + lh;
            const bool rok = s < r_end;
            const float a = (rok && nok) ? dz[s * lddz + ncol] : 0.0f;
            bsum += a;
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                const int kcol = t * 32 + li;
                const float bb = (rok && kcol < K) ? x[s * ldx + kcol] : 0.0f;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[t], 0, 0, 0);
            }
        }
        // D[i = n within tile][j = k within tile]
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const int kcol = t * 32 + li;
            if (kcol < K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nrow = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (nrow < N) atomicAdd(&dw[nrow * K + kcol], acc[t][r]);
                }
            }
        }
        if (db) {
            bsum += __shfl_xor(bsum, 32, 64);
            if (lh == 0 && nok) atomicAdd(&db[ncol], bsum);
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
    const int KP = (K + 1) & ~1, ld = KP + 1;
    return (size_t)(NT * 32 + ROWS_PER_BLOCK) * ld * sizeof(float);
}
size_t bwd_lds_bytes(int K, int N)
{
    const int KP = (K + 1) & ~1, ldw = KP + 1;
    const int NP = (N + 1) & ~1, ldz = NP + 1;
    const int NR = (N + 31) & ~31;
    return ((size_t)NR * ldw + (size_t)ROWS_PER_BLOCK * ldz) * sizeof(float);
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

int rsdf_linear_fwd(const float *x, int ldx, const float *w, const float *b, int64_t n, int K, int N,
                    int act, float *y, int ldy, void *stream)
{
    RSDF_CHECK_ARG(K >= 1 && K <= 128 && N >= 1 && N <= 128, "linear_fwd: K and N must be in [1,128]");
    RSDF_CHECK_ARG(ldx >= K && ldy >= N, "linear_fwd: row stride smaller than the row");
    if (n <= 0) return 0;
    const int NT = (N + 31) / 32;
    const size_t lds = fwd_lds_bytes(K, NT);
    const unsigned grid = rsdf_blocks(n, ROWS_PER_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    int rc = 0;
#define LAUNCH_FWD(NT_)                                                                        \
    rc = allow_lds(linear_fwd_kernel<NT_>, lds);                                               \
    if (rc) return rc;                                                                         \
    linear_fwd_kernel<NT_><<<grid, THREADS, lds, st>>>(x, ldx, w, b, n, K, N, act, y, ldy)
    switch (NT) {
    case 1: LAUNCH_FWD(1); break;
    case 2: LAUNCH_FWD(2); break;
    case 3: LAUNCH_FWD(3); break;
    default: LAUNCH_FWD(4); break;
    }
#undef LAUNCH_FWD
    RSDF_RETURN_LAUNCH();
}

int rsdf_linear_bwd_input(const float *dy, const float *y, int lddy, const float *w, int64_t n, int K,
                          int N, int act, int k0, int Kout, float *dz, float *dx, int lddx,
                          void *stream)
{
    RSDF_CHECK_ARG(K >= 1 && K <= 128 && N >= 1 && N <= 128, "linear_bwd_input: K and N must be in [1,128]");
    RSDF_CHECK_ARG(lddy >= N, "linear_bwd_input: lddy < N");
    RSDF_CHECK_ARG(act == RSDF_ACT_NONE || y != nullptr, "linear_bwd_input: activation needs y");
    if (dx) {
        RSDF_CHECK_ARG(k0 >= 0 && Kout >= 1 && k0 + Kout <= K, "linear_bwd_input: bad column window");
        RSDF_CHECK_ARG(lddx >= Kout, "linear_bwd_input: lddx < Kout");
    }
    if (n <= 0) return 0;
    const int JT = dx ? (Kout + 31) / 32 : 1;
    const size_t lds = bwd_lds_bytes(K, N);
    const unsigned grid = rsdf_blocks(n, ROWS_PER_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    int rc = 0;
#define LAUNCH_BI(JT_)                                                                         \
    rc = allow_lds(linear_bwd_input_kernel<JT_>, lds);                                         \
    if (rc) return rc;                                                                         \
    linear_bwd_input_kernel<JT_><<<grid, THREADS, lds, st>>>(dy, y, lddy, w, n, K, N, act, k0, \
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

int rsdf_linear_bwd_weight(const float *dz, int lddz, const float *x, int ldx, int64_t n, int K, int N,
                           float *dw, float *db, void *stream)
{
    RSDF_CHECK_ARG(K >= 1 && K <= 128 && N >= 1 && N <= 128, "linear_bwd_weight: K and N must be in [1,128]");
    RSDF_CHECK_ARG(lddz >= N && ldx >= K, "linear_bwd_weight: row stride smaller than the row");
    if (n <= 0) return 0;
    const int KT = (K + 31) / 32;
    const unsigned grid = rsdf_blocks(n, SLAB);
    hipStream_t st = (hipStream_t)stream;
    switch (KT) {
    case 1: linear_bwd_weight_kernel<1><<<grid, THREADS, 0, st>>>(dz, lddz, x, ldx, n, K, N, dw, db); break;
    case 2: linear_bwd_weight_kernel<2><<<grid, THREADS, 0, st>>>(dz, lddz, x, ldx, n, K, N, dw, db); break;
    case 3: linear_bwd_weight_kernel<3><<<grid, THREADS, 0, st>>>(dz, lddz, x, ldx, n, K, N, dw, db); break;
    default: linear_bwd_weight_kernel<4><<<grid, THREADS, 0, st>>>(dz, lddz, x, ldx, n, K, N, dw, db); break;
    }
    RSDF_RETURN_LAUNCH();
}

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

}  // extern "C"
