// The steps either side of the field: ray generation + pixel gather (N2) and the occupancy-grid update (M2).
//
//   N2  systems/split_occ.py:58-131 (train branch): directions[y, x] (or [index, y, x]) -> get_rays
//       (models/ray_utils.py:32-56: rays_d = sum(directions * c2w[:3,:3], -1), rays_o = c2w[:3,3]),
//       F.normalize(rays_d), rgb / fg_mask gather from the resident image stack and the masked background
//       blend  rgb * m + rgb_to_srgb(bg * (1 - m))  (:113-116).  One kernel instead of ~12 indexing kernels.
//   M2  lib/nerfacc/grid.py:196-239 (live call models/split_mixed_occ.py:126-131):
//       cell points  x = (coords + jitter) / res * (roi_max - roi_min) + roi_min,
//       occs[idx] = max(occs[idx] * decay, occ)  (duplicate indices resolve to the max of their candidates:
//       the reference's indexed assignment leaves that order undefined), binary = occs > min(mean(occs), thre).
#include "common.h"

namespace {

constexpr int THREADS = 256;

__device__ __forceinline__ float srgb_oetf(float f)
{
    return f <= 0.0031308f ? f * 12.92f : powf(fmaxf(f, 0.0031308f), 1.0f / 2.4f) * 1.055f - 0.055f;
}

__global__ void __launch_bounds__(THREADS)
gen_rays_kernel(const int64_t *__restrict__ index, int64_t n_index, const int64_t *__restrict__ ys,
                const int64_t *__restrict__ xs, const float *__restrict__ directions, int dirs_per_view,
                const float *__restrict__ c2w, const float *__restrict__ images, int channels,
                const float *__restrict__ masks, const float *__restrict__ bg, int apply_mask, int H, int W,
                int64_t n, float *__restrict__ rays, float *__restrict__ rgb, float *__restrict__ fg_mask)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const int64_t v = index[n_index == 1 ? 0 : i], y = ys[i], x = xs[i];
    const int64_t pix = y * W + x;
    const float *d = directions + ((dirs_per_view ? v * (int64_t)H * W : 0) + pix) * 3;
    const float *m = c2w + v * 12;  // [3][4] row major
    const float d0 = d[0], d1 = d[1], d2 = d[2];
    // (directions[:, None, :] * c2w[:, :3, :3]).sum(-1): three products, left-to-right sum
    float r[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) r[k] = (d0 * m[4 * k] + d1 * m[4 * k + 1]) + d2 * m[4 * k + 2];
    // F.normalize(p=2, dim=-1, eps=1e-12): v / max(||v||, eps)
    const float nrm = fmaxf(sqrtf((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]), 1e-12f);
    float *o = rays + i * 6;
    o[0] = m[3]; o[1] = m[7]; o[2] = m[11];
    o[3] = r[0] / nrm; o[4] = r[1] / nrm; o[5] = r[2] / nrm;
    if (rgb != nullptr) {
        const float mk = masks != nullptr ? masks[v * (int64_t)H * W + pix] : 1.0f;
        if (fg_mask != nullptr) fg_mask[i] = mk;
        const float *px = images + (v * (int64_t)H * W + pix) * channels;
        for (int c = 0; c < channels; ++c) {
            float val = px[c];
            if (apply_mask) val = val * mk + srgb_oetf(bg[c < 3 ? c : 2] * (1.0f - mk));
            rgb[i * channels + c] = val;
        }
    }
}

__global__ void __launch_bounds__(THREADS)
occ_points_kernel(const int64_t *__restrict__ indices, const float *__restrict__ jitter,
                  const float *__restrict__ roi, int rx, int ry, int rz, int64_t n, float *__restrict__ x)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const int64_t idx = indices != nullptr ? indices[i] : i;
    const int iz = (int)(idx % rz), iy = (int)((idx / rz) % ry), ix = (int)(idx / ((int64_t)ry * rz));
    const float c[3] = {(float)ix, (float)iy, (float)iz};
    const float res[3] = {(float)rx, (float)ry, (float)rz};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float u = (c[k] + jitter[3 * i + k]) / res[k];
        x[3 * i + k] = u * (roi[3 + k] - roi[k]) + roi[k];
    }
}

// pass 1: cand[idx] = max over duplicates of occ_i  (occ >= 0: float order == int order of the bit patterns)
__global__ void __launch_bounds__(THREADS)
occ_cand_kernel(const int64_t *__restrict__ indices, const float *__restrict__ occ, int64_t n,
                int *__restrict__ cand)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const int64_t idx = indices != nullptr ? indices[i] : i;
    atomicMax(cand + idx, __float_as_int(fmaxf(occ[i], 0.0f)));
}
// pass 2 (one thread per update; duplicates write the same value): occs = max(occs * decay, cand), idempotent
// because a cell's first writer flips the candidate's sign bit as a "done" mark
__global__ void __launch_bounds__(THREADS)
occ_apply_kernel(const int64_t *__restrict__ indices, int64_t n, float decay, int *__restrict__ cand,
                 float *__restrict__ occs)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const int64_t idx = indices != nullptr ? indices[i] : i;
    const int c = atomicOr(cand + idx, (int)0x80000000u);
    if (c < 0) return;  // another duplicate already applied this cell
    occs[idx] = fmaxf(occs[idx] * decay, __int_as_float(c));
}

__global__ void __launch_bounds__(THREADS)
occ_sum_kernel(const float *__restrict__ occs, int64_t n, double *__restrict__ sum)
{
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS)
        acc += (double)occs[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ double part[THREADS / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < THREADS / 64; ++w) t += part[w];
        atomicAdd(sum, t);
    }
}
__global__ void __launch_bounds__(THREADS)
occ_binary_kernel(const float *__restrict__ occs, int64_t n, const double *__restrict__ sum, float occ_thre,
                  uint8_t *__restrict__ binary)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const float thre = fminf((float)(*sum / (double)n), occ_thre);  // torch.clamp(occs.mean(), max=occ_thre)
    binary[i] = occs[i] > thre ? 1 : 0;
}

}  // namespace

extern "C" {

int rsdf_gen_rays(const int64_t *index, int64_t n_index, const int64_t *y, const int64_t *x,
                  const float *directions, int dirs_per_view, const float *c2w, const float *images, int channels,
                  const float *fg_masks, const float *background_color, int apply_mask, int H, int W, int64_t n,
                  float *rays, float *rgb, float *fg_mask, void *stream)
{
    RSDF_CHECK_ARG(n_index == 1 || n_index == n, "gen_rays: index must hold 1 or n view indices");
    RSDF_CHECK_ARG(rgb == nullptr || images != nullptr, "gen_rays: rgb output needs the image stack");
    RSDF_CHECK_ARG(!apply_mask || (fg_masks != nullptr && background_color != nullptr),
                   "gen_rays: apply_mask needs fg_masks and background_color");
    if (n <= 0) return 0;
    gen_rays_kernel<<<rsdf_blocks(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        index, n_index, y, x, directions, dirs_per_view, c2w, images, channels, fg_masks, background_color,
        apply_mask, H, W, n, rays, rgb, fg_mask);
    RSDF_RETURN_LAUNCH();
}

int rsdf_occ_cell_points(const int64_t *indices, const float *jitter, const float *roi, int res_x, int res_y,
                         int res_z, int64_t n, float *x, void *stream)
{
    if (n <= 0) return 0;
    occ_points_kernel<<<rsdf_blocks(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(indices, jitter, roi, res_x,
                                                                                   res_y, res_z, n, x);
    RSDF_RETURN_LAUNCH();
}

int64_t rsdf_occ_update_scratch_bytes(int64_t n_cells) { return n_cells * (int64_t)sizeof(int) + 8; }

int rsdf_occ_update(const int64_t *indices, const float *occ, int64_t n, float ema_decay, float occ_thre,
                    int64_t n_cells, float *occs, uint8_t *binary, void *scratch, void *stream)
{
    RSDF_CHECK_ARG(scratch != nullptr && occs != nullptr && binary != nullptr, "occ_update: null buffer");
    hipStream_t st = (hipStream_t)stream;
    double *sum = reinterpret_cast<double *>(scratch);
    int *cand = reinterpret_cast<int *>(reinterpret_cast<char *>(scratch) + 8);
    hipError_t e = hipMemsetAsync(scratch, 0, (size_t)rsdf_occ_update_scratch_bytes(n_cells), st);
    if (e != hipSuccess) { rsdf_set_error(hipGetErrorString(e)); return (int)e; }
    if (n > 0) {
        occ_cand_kernel<<<rsdf_blocks(n, THREADS), THREADS, 0, st>>>(indices, occ, n, cand);
        occ_apply_kernel<<<rsdf_blocks(n, THREADS), THREADS, 0, st>>>(indices, n, ema_decay, cand, occs);
    }
    const unsigned sblocks = (unsigned)(n_cells / (THREADS * 8) > 1024 ? 1024 : (n_cells + THREADS * 8 - 1) / (THREADS * 8));
    occ_sum_kernel<<<sblocks ? sblocks : 1, THREADS, 0, st>>>(occs, n_cells, sum);
    occ_binary_kernel<<<rsdf_blocks(n_cells, THREADS), THREADS, 0, st>>>(occs, n_cells, sum, occ_thre, binary);
    RSDF_RETURN_LAUNCH();
}

}  // extern "C"
