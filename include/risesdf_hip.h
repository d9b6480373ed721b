/*
 * risesdf_hip.h -- C ABI of librisesdf_hip.so, the MI355X (gfx950) implementation of
 * RISE-SDF's ray-marched SDF volume-rendering hot path.
 *
 * Conventions (mirror what the reference's in-tree native ops guarantee at the Python
 * surface, lib/nerfacc/cuda/csrc/include/helpers_cuda.h:20-32):
 *   - every pointer is a DEVICE pointer to contiguous memory unless marked "host";
 *   - the library never allocates, frees, or synchronises: the caller owns outputs and
 *     scratch (the Python host allocates them through torch's caching allocator);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); kernels are
 *     enqueued on it and the call returns immediately;
 *   - return value 0 = success, otherwise a hipError_t (kernel-launch error) or
 *     RSDF_EINVAL for a rejected argument; rsdf_last_error() gives a message;
 *   - fp32 everywhere, ray indices int64 and packed_info int32 as in nerfacc.
 *
 * Each entry point names the reference interface it replaces (file:line relative to the
 * upstream dehezhang2/RISE-SDF tree).  INTEGRATION.md shows the reference-side binding.
 */
#ifndef RISESDF_HIP_H
#define RISESDF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSDF_ABI_VERSION 3
#define RSDF_EINVAL 10001
#define RSDF_MAX_LEVELS 32
#define RSDF_TAP_MAJOR (-1)

/* activation ids for rsdf_linear_* (models/network_utils.py:152-157, models/utils.py:71-99) */
#define RSDF_ACT_NONE 0
#define RSDF_ACT_RELU 1
#define RSDF_ACT_SOFTPLUS100 2 /* nn.Softplus(beta=100, threshold=20) */
#define RSDF_ACT_SIGMOID 3

int rsdf_abi_version(void);
const char *rsdf_last_error(void);

/* ---- sticky device-side status words ------------------------------------------------------------
 * Kernels never synchronise, so conditions only the device can see are COUNTED into a caller-owned int32
 * [RSDF_STATUS_WORDS] device array (zeroed by the caller, nullable everywhere) that the host reads whenever it reads
 * anything else (rise_sdf_amd/_lib.py::poll_status: next to the marcher's sample count). */
#define RSDF_STATUS_WORDS 8
#define RSDF_STATUS_X2_FWD_NONFINITE 0 /* waves of rsdf_sdfmlp_fd7_fwd_x2 that produced an inf / nan output: an operand
                                          left the x2 format's fp16 class range (|input| < 255, |weight| < 1023,
                                          |hidden activation| < 454), where the reference's fp32 MLP stays finite */
#define RSDF_STATUS_X2_BWD_REROUTED 1  /* rsdf_sdfmlp_fd7_bwd_x2 launches that ran on the range-free kernels */
#define RSDF_STATUS_X2_BWD_GUARDED 2   /* rsdf_sdfmlp_fd7_bwd_x2 launches that the range guard examined */
#define RSDF_STATUS_PAIR_PACK_NONFINITE 3 /* of word 0's count: waves of rsdf_pair_pack / _pack2 (a network INPUT out of range) */
#define RSDF_STATUS_PAIR_FWD_NONFINITE 4  /* of word 0's count: workgroups of rsdf_pair_fwd (a weight or hidden activation) */

/* ---- M1: ray/AABB slab test ----------------------------------------------------------------
 * replaces _C.ray_aabb_intersect  (lib/nerfacc/cuda/csrc/intersection.cu:93-133, pybind.cu) and
 * nerfacc.ray_aabb_intersect.  aabb = {xmin,ymin,zmin,xmax,ymax,zmax}; miss => both 1e10;
 * t_min clamped to >= 0. */
int rsdf_ray_aabb_intersect(const float *rays_o, const float *rays_d, const float *aabb,
                            int64_t n_rays, float *t_min, float *t_max, void *stream);

/* ---- M3/M4: occupancy-grid marcher ----------------------------------------------------------
 * replaces _C.ray_marching (lib/nerfacc/cuda/csrc/ray_marching.cu:194-289) as an explicit
 * count -> scan -> write sequence with no host synchronisation inside the library.
 * `binary` is the bool/uint8 grid [res_x,res_y,res_z] (C order), `roi` 6 floats.
 * AABB contraction only (the only type the hot path uses, models/split_mixed_occ.py:66). */
int rsdf_march_count(const float *rays_o, const float *rays_d, const float *t_min,
                     const float *t_max, const float *roi, const uint8_t *binary, int res_x,
                     int res_y, int res_z, float step_size, float cone_angle, int64_t n_rays,
                     int32_t *num_steps, void *stream);
/* packed_info[r] = {exclusive prefix of counts, counts[r]}; *total (device int32) = sum.
 * scratch: >= rsdf_scan_scratch_bytes(n) bytes. */
int64_t rsdf_scan_scratch_bytes(int64_t n);
int rsdf_pack_from_counts(const int32_t *counts, int64_t n, int32_t *packed_info, int32_t *total,
                          void *scratch, void *stream);
int rsdf_march_write(const float *rays_o, const float *rays_d, const float *t_min,
                     const float *t_max, const float *roi, const uint8_t *binary, int res_x,
                     int res_y, int res_z, float step_size, float cone_angle, int64_t n_rays,
                     const int32_t *packed_info, int64_t *ray_indices, float *t_starts,
                     float *t_ends, void *stream);
/* STAGED form of the same two passes (ray_marching.cu:257-289 marches every ray twice; a pass is bound by the serial
 * recurrence, not by its stores): the count pass also parks each ray's (t0, t1) pairs in a slot of ``stride`` samples
 * (stage_t0 / stage_t1: float [n_rays][stride]); the write pass copies the rays whose count fits their slot and marches the
 * others again.  Same samples, same values, same order as rsdf_march_count + rsdf_march_write for every stride >= 1. */
int rsdf_march_count_staged(const float *rays_o, const float *rays_d, const float *t_min, const float *t_max,
                            const float *roi, const uint8_t *binary, int res_x, int res_y, int res_z, float step_size,
                            float cone_angle, int64_t n_rays, int32_t *num_steps, int64_t stride, float *stage_t0,
                            float *stage_t1, void *stream);
int rsdf_march_write_staged(const float *rays_o, const float *rays_d, const float *t_min, const float *t_max,
                            const float *roi, const uint8_t *binary, int res_x, int res_y, int res_z, float step_size,
                            float cone_angle, int64_t n_rays, const int32_t *packed_info, const int32_t *num_steps,
                            int64_t stride, const float *stage_t0, const float *stage_t1, int64_t *ray_indices,
                            float *t_starts, float *t_ends, void *stream);
/* replaces _C.query_occ (ray_marching.cu:295-358); cell (nullable) gets the linear cell index,
 * -1 outside the box. */
int rsdf_query_occ(const float *samples, const float *roi, const uint8_t *binary, int res_x,
                   int res_y, int res_z, int64_t n, uint8_t *occ, int32_t *cell, void *stream);

/* ---- M5/M6: pack / unpack / compaction ------------------------------------------------------
 * replaces pack_info (lib/nerfacc/pack.py:47-78), _C.unpack_info (pack.cu:7-28) and the
 * boolean-mask compaction of lib/nerfacc/ray_marching.py:213-218.
 * ray_indices must be sorted (as the marcher emits them). counts: int32 [n_rays] scratch. */
int rsdf_counts_from_ray_indices(const int64_t *ray_indices, int64_t n_samples, int64_t n_rays,
                                 int32_t *counts, void *stream);
int rsdf_unpack_info(const int32_t *packed_info, int64_t n_rays, int64_t *ray_indices,
                     void *stream);
/* keep[i] != 0 -> sample survives.  offsets: int32 [n] scratch (exclusive scan of keep, filled
 * by the call); *n_kept device int32. Outputs sized by the caller (>= n).  extra / extra_out (nullable, together): one more
 * per-sample float array compacted alongside (the alphas the visibility test was made from: a no-grad renderer of the kept
 * samples -- models/volrend.py:18-127, the secondary rays -- need not evaluate the field again for the same values). */
int rsdf_compact_samples(const uint8_t *keep, const int64_t *ray_indices, const float *t_starts,
                         const float *t_ends, int64_t n, int32_t *offsets, int32_t *n_kept,
                         void *scan_scratch, int64_t *ray_indices_out, float *t_starts_out,
                         float *t_ends_out, const float *extra, float *extra_out, void *stream);

/* ---- C1: transmittance / weights from alpha ---------------------------------------------------
 * replaces nerfacc.render_weight_from_alpha / render_transmittance_from_alpha
 * (call site models/volrend.py:851-855; spec lib/nerfacc/cuda/csrc/render_weight.cu:86-153,
 * render_transmittance.cu:85-145, render_transmittance_cub.cu:111-166).  One wavefront per ray,
 * wave64 multiplicative scan.  T_i = prod_{j<i}(1-a_j), w_i = a_i T_i. */
int rsdf_weight_from_alpha_fwd(const int32_t *packed_info, const float *alphas, int64_t n_rays,
                               float *weights, float *trans, void *stream);
int rsdf_weight_from_alpha_bwd(const int32_t *packed_info, const float *alphas,
                               const float *weights, const float *trans,
                               const float *grad_weights, int64_t n_rays, float *grad_alphas,
                               void *stream);
/* The backward in the reference's sequential order (render_weight.cu:114-153), one lane per ray.  fmad != 0: with the
 * multiply-add contraction nvcc's default --fmad=true applies to that source (the reference binary; what
 * rsdf_weight_from_alpha_bwd runs); fmad == 0: every source operation rounded once.  Additive entry point (round 6). */
int rsdf_weight_from_alpha_bwd_seq(const int32_t *packed_info, const float *alphas, const float *weights,
                                   const float *grad_weights, int64_t n_rays, int fmad, float *grad_alphas,
                                   void *stream);
int rsdf_transmittance_from_alpha_bwd(const int32_t *packed_info, const float *alphas,
                                      const float *trans, const float *grad_trans,
                                      int64_t n_rays, float *grad_alphas, void *stream);
/* visibility mask (lib/nerfacc/vol_rendering.py:503-520): keep = T >= eps && (thre<=0 || a >= thre) */
int rsdf_visibility_from_alpha(const int32_t *packed_info, const float *alphas, int64_t n_rays,
                               float early_stop_eps, float alpha_thre, uint8_t *keep,
                               void *stream);

/* ---- C2: accumulate along rays ----------------------------------------------------------------
 * replaces nerfacc.accumulate_along_rays (call sites models/volrend.py:871-885; spec
 * lib/nerfacc/vol_rendering.py:174-198).  values nullable (=> D must be 1, out = sum of weights).
 * Every ray's row of `out` is written (zeros for empty rays): no pre-zeroing needed. */
int rsdf_accumulate_fwd(const int32_t *packed_info, const float *weights, const float *values,
                        int64_t n_rays, int D, float *out, void *stream);
int rsdf_accumulate_bwd(const int32_t *packed_info, const float *weights, const float *values,
                        const float *grad_out, int64_t n_rays, int D, float *grad_weights,
                        float *grad_values, void *stream);
/* Opacity [n_rays] and depth [n_rays] of models/volrend.py:878-885 in one pass each way: accumulate_along_rays(weights, None)
 * and accumulate_along_rays(weights, (t_starts + t_ends)[..., None] / 2.0) with the midpoint formed in the kernel;
 * bit-identical to the two rsdf_accumulate_* calls.  Backward: grad_weights [S] = grad_opacity[ray] + grad_depth[ray] * mid
 * (either gradient nullable = zero; samples outside packed_info are not written). */
int rsdf_opacity_depth_fwd(const int32_t *packed_info, const float *weights, const float *t_starts, const float *t_ends,
                           int64_t n_rays, float *opacity, float *depth,
                           float *midpoints /* nullable: [S] = (t_starts + t_ends) / 2 of the samples packed_info owns */,
                           void *stream);
int rsdf_opacity_depth_bwd(const int32_t *packed_info, const float *t_starts, const float *t_ends,
                           const float *grad_opacity /* nullable */, const float *grad_depth /* nullable */,
                           int64_t n_rays, float *grad_weights, void *stream);
/* The same pass with the ray's normal map folded in (models/volrend.py:875-877: accumulate_along_rays(weights,
 * normals [n_samples,3])): normal_map [n_rays,3]; bit-identical to the separate accumulate call.  bwd: any of the three
 * incoming gradients may be NULL (zero); grad_weights [n_samples] and / or grad_normals [n_samples,3] (either nullable;
 * grad_normals needs grad_normal_map) are written for every sample packed_info covers. */
int rsdf_opacity_depth_normal_fwd(const int32_t *packed_info, const float *weights, const float *t_starts,
                                  const float *t_ends, const float *normals, int64_t n_rays, float *opacity, float *depth,
                                  float *normal_map, float *midpoints /*nullable*/, void *stream);
int rsdf_opacity_depth_normal_bwd(const int32_t *packed_info, const float *weights, const float *t_starts,
                                  const float *t_ends, const float *normals, const float *grad_opacity,
                                  const float *grad_depth, const float *grad_normal_map, int64_t n_rays,
                                  float *grad_weights, float *grad_normals, void *stream);

/* ---- H1: multiresolution hash-grid encoding ----------------------------------------------------
 * replaces tcnn.Encoding(3, {otype: HashGrid, ...}) forward/backward (constructed
 * models/network_utils.py:47-50, called :59; tiny-cuda-nn is not vendored).  The level table is
 * passed by value from the host. Table layout [level][entry][feature] fp32. */
typedef struct rsdf_grid_meta {
    uint32_t n_levels;
    uint32_t n_features; /* 1, 2 or 4 */
    float scale[RSDF_MAX_LEVELS];
    uint32_t res[RSDF_MAX_LEVELS];
    uint32_t offset[RSDF_MAX_LEVELS]; /* entries */
    uint32_t size[RSDF_MAX_LEVELS];   /* entries */
} rsdf_grid_meta;

/* host helper: fills meta exactly as the oracle does; returns the parameter count */
int64_t rsdf_grid_meta_init(rsdf_grid_meta *meta /*host*/, int n_levels, int n_features,
                            int log2_hashmap_size, int base_resolution, double per_level_scale);

/* x [n,3] in [0,1].  out row stride ld_out floats, features written at column col_off.
 * Levels >= n_active_levels are written as zeros without being fetched (H2: the progressive mask
 * of models/network_utils.py:58-68).  If write_xyz != 0 columns [0,3) get x*xyz_scale+xyz_offset
 * (CompositeEncoding include_xyz, models/network_utils.py:78-79; then col_off must be 3). */
int rsdf_hashgrid_fwd(const float *x, const float *table, const rsdf_grid_meta *meta /*host*/,
                      int64_t n, int n_active_levels, float *out, int ld_out, int col_off,
                      int write_xyz, float xyz_scale, float xyz_offset, void *stream);
/* The same encoding (bit-identical rows) for large batches: a level-major pass writes [n_active][n][F] planes into scratch
 * (every workgroup in flight reads the same L2-resident level), a second pass lays them out as rows through LDS.
 * scratch >= rsdf_hashgrid_fwd_staged_scratch_bytes(...) bytes; write_xyz needs col_off == 3. */
int64_t rsdf_hashgrid_fwd_staged_scratch_bytes(const rsdf_grid_meta *meta /*host*/, int64_t n, int n_active_levels);
int rsdf_hashgrid_fwd_staged(const float *x, const float *table, const rsdf_grid_meta *meta /*host*/,
                             int64_t n, int n_active_levels, float *out, int ld_out, int col_off,
                             int write_xyz, float xyz_scale, float xyz_offset, void *scratch,
                             int64_t scratch_bytes, void *stream);
/* dtable += scatter(w * dout); dout row stride ld_dout, features at column col_off.
 * dtable must be zeroed (or hold a running sum) by the caller. */
int rsdf_hashgrid_bwd(const float *x, const float *dout, const rsdf_grid_meta *meta /*host*/,
                      int64_t n, int n_active_levels, int ld_dout, int col_off, float *dtable,
                      void *stream);

/* Run-time choice of the hash backward's queue record format for this process (round 6; both forms are compiled in):
 * 0 = 16-byte elements of two block-float contributions (the larger feature gradient keeps 20 significant bits: the default),
 * 1 = 20-byte elements of fp32 values (what the reference's fp32 atomics accumulate, models/network_utils.py:47-59).  Scratch
 * sizes (rsdf_hashgrid_bwd_fd7_scratch_bytes, ..._scatter_binned_scratch_bytes) follow the current setting: set it once, before
 * the first backward (rise_sdf_amd reads RSDF_REC=fp32 at load). */
int rsdf_set_record_format(int fp32_values);
int rsdf_get_record_format(void);
/* ---- H1/H1b for the finite-difference stencil (models/geometry.py:229-244) -------------------------
 * Tap-major structure-of-arrays layouts (n = number of samples):
 *   x7t    [7][n][3]     tap t of sample s, unit cube; tap 0 = centre, then +x,-x,+y,-y,+z,-z
 *                        (rsdf_fd_points with tap_major = 1)
 *   planes [L][7][n][2]  encoding (or its gradient) of level l, tap t, sample s  (n_features = 2)
 * fwd: gathers each sample's centre cell once plus 4 corners per displaced tap and evaluates the 7
 *      interpolations from registers; bit-identical to rsdf_hashgrid_fwd on the same points.
 * bwd: merges the taps' contributions in registers, bins them by table slice through LDS and reduces
 *      in LDS: no per-corner global atomics.  eps_unit = eps / (2*radius) only sizes the queues.
 *      scratch >= rsdf_hashgrid_bwd_fd7_scratch_bytes(...) bytes; dtable is accumulated into.
 *      The queues carry a table entry's two feature gradients as an 8-byte block-float record (20 significant bits for
 *      the larger value, fp64 accumulation in LDS, one fp32 flush): per entry within 2^-20 of the sum of its records'
 *      larger components of the exact sum (csrc/hashgrid_fd7.hip PairRec; a library built with -DRSDF_REC_FP32 keeps fp32
 *      record values at 10 bytes per record). */
int rsdf_hashgrid_fwd_fd7(const float *x7t, const float *table, const rsdf_grid_meta *meta /*host*/,
                          int64_t n_samples, int n_active_levels, float *planes, void *stream);
int64_t rsdf_hashgrid_bwd_fd7_scratch_bytes(const rsdf_grid_meta *meta /*host*/, int64_t n_samples,
                                            int n_active_levels, float eps_unit);
int rsdf_hashgrid_bwd_fd7(const float *x7t, const float *dplanes, const rsdf_grid_meta *meta /*host*/,
                          int64_t n_samples, int n_active_levels, float eps_unit, float *dtable,
                          void *scratch, int64_t scratch_bytes, void *stream);
/* Layout conversions between the reference-shaped tensors of the stencil path and the tap-major buffers above, for callers
 * that hold the [n, 7] stencil interleaved, as tcnn.Encoding.forward receives it from VolumeSDF.forward
 * (models/geometry.py:229-244 -> models/network_utils.py:47-59):
 *   points_tap_major: x7 [n][7][3] -> x7t [7][n][3]
 *   planes_to_rows:   planes [L][7][n][2] (+ x7 for the xyz columns) -> rows out[(7 s + t)][col_off + 2 l + f], and, with
 *                     write_xyz, out[..][col_off - 3 + d] = x7 * xyz_scale + xyz_offset; levels >= n_active_levels are zeros
 *   rows_to_planes:   g[(7 s + t)][col_off + 2 l + f] -> dplanes [L][7][n][2]   (the backward's re-layout)
 * Pure data movement (bit-identical to the permuted copies they replace). */
int rsdf_stencil_points_tap_major(const float *x7, int64_t n_samples, float *x7t, void *stream);
int rsdf_stencil_planes_to_rows(const float *planes, const float *x7, int64_t n_samples, int n_levels, int n_active_levels,
                                float *out, int ld_out, int col_off, int write_xyz, float xyz_scale, float xyz_offset,
                                void *stream);
int rsdf_stencil_rows_to_planes(const float *g, int ld, int col_off, int64_t n_samples, int n_levels, float *dplanes,
                                void *stream);
/* The same two kernels with the stencil DERIVED in-kernel from each sample's world-space centre points [n][3]
 * (what rsdf_fd_points returns as `positions`): x +- eps e_k, clamp(-radius, radius), AABB contraction, in exactly
 * rsdf_fd_points' arithmetic (models/geometry.py:229-244), so cells and weights are bit-identical to the x7t form.
 * Both kernels walk the samples once per level; 16 x 84 B of tap positions per sample become 16 x 12 B.
 * (The contraction's division by 2 r is taken as five multiply-adds that round like the division for 2^-100 < |p + r| <
 * 2^100: RSDF_EINVAL for a radius outside (2^-101, 2^99); a centre at exactly -r gives +0 where the division gives -0, a
 * non-finite one NaN where it gives inf -- same cells, same weights, same garbage.)
 * bwd: eps_unit (= eps / (2*radius)) only sizes the queues, as above; pass the value the scratch size was asked for. */
int rsdf_hashgrid_fwd_fd7_pts(const float *points, float radius, float eps, const float *table,
                              const rsdf_grid_meta *meta /*host*/, int64_t n_samples, int n_active_levels,
                              float *planes, void *stream);
int rsdf_hashgrid_bwd_fd7_pts(const float *points, float radius, float eps, const float *dplanes,
                              const rsdf_grid_meta *meta /*host*/, int64_t n_samples, int n_active_levels,
                              float eps_unit, float *dtable, void *scratch, int64_t scratch_bytes, void *stream);

/* The table scatter of rsdf_hashgrid_bwd (mode 0) or of rsdf_hashgrid_dx_bwd (mode 1; g_dx [n,3] as there; call that
 * entry with dtable = NULL for its other outputs) for n_features = 2, through the bin-and-reduce queues of the stencil
 * backward instead of per-corner float atomics: same sums (fp64 accumulation per bin, then one add per entry).  Worth it
 * from a few 1e4 points (the training step's curvature term sends 2.6e5 per step).  dtable is accumulated into;
 * scratch >= rsdf_hashgrid_scatter_binned_scratch_bytes(meta, n, n_active_levels). */
int64_t rsdf_hashgrid_scatter_binned_scratch_bytes(const rsdf_grid_meta *meta /*host*/, int64_t n, int n_active_levels);
int rsdf_hashgrid_scatter_binned(int mode, const float *x, const float *dy, int ld_dy, int col_off, const float *g_dx,
                                 const rsdf_grid_meta *meta /*host*/, int64_t n, int n_active_levels, float *dtable,
                                 void *scratch, int64_t scratch_bytes, void *stream);

/* H1 input gradient (what tcnn's autograd supplies to analytic normals, models/geometry.py:224-228, and to the
 * curvature term, geometry.py:262-270) and its backward (tcnn double backward).  x in [0,1]; dx in the same
 * unit-cube coordinates; levels >= n_active_levels contribute nothing.
 *   rsdf_hashgrid_dx:      dx[n,3] = sum_l J_l(x)^T dy[:, col_off + l*F ...]
 *   rsdf_hashgrid_dx_bwd:  given g_dx = dL/d(dx): d_dy (nullable; ld_ddy/col_off_ddy; masked levels zeroed),
 *                          dtable (nullable, ACCUMULATES with float atomics), g_x[n,3] (nullable; the mixed
 *                          second derivatives of the trilinear weights). */
int rsdf_hashgrid_dx(const float *x, const float *table, const rsdf_grid_meta *meta /*host*/, int64_t n,
                     int n_active_levels, const float *dy, int ld_dy, int col_off, float *dx, void *stream);
int rsdf_hashgrid_dx_bwd(const float *x, const float *table, const rsdf_grid_meta *meta /*host*/, int64_t n,
                         int n_active_levels, const float *dy, int ld_dy, int col_off, const float *g_dx,
                         float *d_dy, int ld_ddy, int col_off_ddy, float *dtable, float *g_x, void *stream);

/* ---- H3: VanillaMLP layers on the fp32 matrix cores --------------------------------------------
 * replaces nn.Linear (+ activation) inside VanillaMLP (models/network_utils.py:109-157).
 * y[n,N] = act(x[n,K] @ w[N,K]^T + b[N]); row strides ldx / ldy.  K,N <= 128. */
int rsdf_linear_fwd(const float *x, int ldx, const float *w, const float *b, int64_t n, int K,
                    int N, int act, float *y, int ldy, void *stream);
/* dz = dy * act'(y) (written to dz, may alias dy; row stride lddy for both);
 * dx[n, k0:k0+Kout] = dz @ w[:, k0:k0+Kout]  (dx nullable; dx has row stride lddx and receives
 * Kout columns starting at its column 0). */
int rsdf_linear_bwd_input(const float *dy, const float *y, int lddy, const float *w, int64_t n,
                          int K, int N, int act, int k0, int Kout, float *dz, float *dx, int lddx,
                          void *stream);
/* dw[N,K] += dz^T @ x ; db[N] += colsum(dz).  Accumulates with fp32 atomics: zero first. */
int rsdf_linear_bwd_weight(const float *dz, int lddz, const float *x, int ldx, int64_t n, int K,
                           int N, float *dw, float *db, void *stream);
/* Both of the above in one pass over the rows, for the 128-wide layers (N == 128, K <= 128:
 * rsdf_linear_bwd_fused_supported): dz = dy * act'(y) stays on the CU, dx (nullable) and dw / db as above.
 * Replaces the dz round trip through HBM of the two-kernel form (models/network_utils.py:109-157 backward;
 * the texture networks of models/texture.py:237-327).  dw (required) and db (nullable) accumulate: zero first.
 * prev_act = RSDF_ACT_RELU (needs dx, k0 == 0): x is the ReLU output of the previous layer of a chain and dx is written as
 * THAT layer's dz = dx * (x > 0); its backward is then called with act = RSDF_ACT_NONE, y = NULL (nn.Sequential of
 * Linear + ReLU(inplace), models/network_utils.py:121-126).  Otherwise RSDF_ACT_NONE. */
int rsdf_linear_bwd_fused_supported(int K, int N);
int rsdf_linear_bwd_fused(const float *dy, const float *y, int lddy, const float *x, int ldx,
                          const float *w, int64_t n, int K, int N, int act, int k0, int Kout, float *dx,
                          int lddx, int prev_act, float *dw, float *db, void *stream);
/* The same for the hidden layer right below a network's narrow output layer (N2 <= 4 columns, weights w2 [N2][128], act =
 * ReLU): its input gradient dy = dz_out @ w2 is formed on the fly from the output layer's dz_out [n][N2] (computed by the
 * caller, e.g. rsdf_linear_bwd_input with dx = NULL), so that layer needs no input-gradient kernel at all. */
int rsdf_linear_bwd_fused_tail(const float *dz_out, int N2, const float *w2, const float *y, int lddy,
                               const float *x, int ldx, const float *w, int64_t n, int K, int N, int act,
                               int k0, int Kout, float *dx, int lddx, int prev_act, float *dw, float *db,
                               void *stream);
/* The same two calls with a WORKSPACE for the weight-gradient accumulators (round 3): the workgroups store their dW tiles
 * as plain per-workgroup partials and a second kernel adds their sum to dw, instead of up to 256 workgroups flushing
 * N K float atomics each onto the same N K addresses (~0.16 ms per launch whatever the row count: two thirds of the call
 * at a training step's 250 k rows).  workspace >= rsdf_linear_bwd_fused_workspace_bytes(n, K, N) bytes, private to the
 * stream until the call has run; NULL (or too small, or fewer than 64 row tiles) = the atomic flush.  Same results up to
 * the order of the fp32 additions. */
int64_t rsdf_linear_bwd_fused_workspace_bytes(int64_t n, int K, int N);
int rsdf_linear_bwd_fused_ws(const float *dy, const float *y, int lddy, const float *x, int ldx,
                             const float *w, int64_t n, int K, int N, int act, int k0, int Kout, float *dx,
                             int lddx, int prev_act, float *dw, float *db, void *workspace,
                             int64_t workspace_bytes, void *stream);
int rsdf_linear_bwd_fused_tail_ws(const float *dz_out, int N2, const float *w2, const float *y, int lddy,
                                  const float *x, int ldx, const float *w, int64_t n, int K, int N, int act,
                                  int k0, int Kout, float *dx, int lddx, int prev_act, float *dw, float *db,
                                  void *workspace, int64_t workspace_bytes, void *stream);
/* Fused SDF network for the finite-difference stencil: [x*xyz_scale+xyz_offset | planes] ->
 * Linear(K0,H) -> Softplus(100) -> Linear(H,H) -> Softplus(100) -> Linear(H,N2), K0 = 3 + 2*n_levels
 * (CompositeEncoding include_xyz + VanillaMLP n_hidden_layers=2, models/network_utils.py:71-157) on
 * the tap-major buffers above.  Activations stay on the CU; levels >= n_active_levels read as zero.
 * Supported: n_levels <= 16, H in {32, 64, 128}, N2 <= 64 (rsdf_sdfmlp_fd7_supported(K0, H, N2)); H = 128 is
 * the width of configs/split-mixed-occ-tensoir.yaml:73-84.
 * fwd: sdf7t [7][n] = output column 0 of every tap; feature [n, N2] (nullable) = full output of
 *      the centre taps; h2c [n, H] (nullable, needs feature) = their second hidden layer, kept for
 *      the weight gradient of the feature rows.
 * bwd: from d_sdf7t [7][n] and d_feature [n, N2] (nullable; then dh2c_scratch [n, H] must be given: it receives
 *      d_feature @ W2, the centre taps' d(h2) through the feature rows): d_planes (nullable) receives
 *      d/d(hash features); dw0 [H,K0], db0 [H], dw1 [H,H], db1 [H] are complete; dw2 [N2,H] / db2 [N2]
 *      receive the SDF-column part (row 0 / element 0) -- the caller adds the feature part with
 *      rsdf_linear_bwd_weight(d_feature, h2c).  All accumulated atomically: zero first. */
int rsdf_sdfmlp_fd7_supported(int K0, int H, int N2);
int rsdf_sdfmlp_fd7_fwd(const float *x7t, const float *planes, int n_levels, int n_active_levels,
                        float xyz_scale, float xyz_offset, int H, int N2, const float *w0,
                        const float *b0, const float *w1, const float *b1, const float *w2,
                        const float *b2, int64_t n_samples, float *sdf7t, float *feature, float *h2c,
                        void *stream);
int rsdf_sdfmlp_fd7_bwd(const float *x7t, const float *planes, int n_levels, int n_active_levels,
                        float xyz_scale, float xyz_offset, int H, int N2, const float *w0,
                        const float *b0, const float *w1, const float *b1, const float *w2,
                        const float *b2, int64_t n_samples, const float *d_sdf7t,
                        const float *d_feature, float *dh2c_scratch /*nullable*/, float *d_planes, float *dw0,
                        float *db0, float *dw1, float *db1, float *dw2, float *db2, void *stream);
/* ---- The same fused node in the "x2" form (round 4; csrc/mlp_x2.hip): every fp32 matrix operand is carried as TWO fp16
 * parts (v S = hi + lo with a power-of-two class scale S: v to 2^-24 relative, i.e. half an fp32 ulp) and every product is
 * three v_mfma_f32_*_f16 instructions with fp32 accumulation (the form above: three bf16 parts, six products).  A layer's
 * error is that of an fp32 GEMM (~1e-7 of the largest output).  The input arrives PRE-SPLIT: the stencil gather writes, in
 * place of the fp32 planes and at the same 2 x 16 bits per value,
 *     x2 [tile = row / 32][tap 7][part 2][column 36][32 rows] fp16,   rsdf_x2_bytes(n) bytes, rows = rsdf_x2_rows(n)
 *     column 2 l + f = feature f of level l (levels >= n_active_levels: zeros), 32..34 = the tap's x, y, z in the unit
 *     cube * xyz_scale + xyz_offset (CompositeEncoding's include_xyz pass-through, models/network_utils.py:71-88), 35 = 1
 *     (the bias column); all times 2^8; rows >= n of the last tile are zeros; the two 16-row halves of columns with bit 3
 *     of their index set are swapped (an LDS bank swizzle the MLP kernels expect).
 * rsdf_hashgrid_fwd_fd7_x2 takes the stencil either as x7t [7][n][3] (points = NULL) or derived from the world-space
 * centres points [n][3] with radius / eps (x7t = NULL), exactly as rsdf_hashgrid_fwd_fd7 / _pts do; hi + lo == the value
 * those write, times 2^8, to 2^-24.  The MLP entry points replace (x7t, planes, xyz_scale, xyz_offset) by x2: forward
 * H <= 64, backward H = 64 (rsdf_sdfmlp_fd7_x2_supported); same outputs and gradient contract as rsdf_sdfmlp_fd7_fwd / _bwd.
 * bwd additionally needs guard_scratch (32 bytes of device memory): it scans d_sdf7t (and d_feature's d(h2)) for their
 * largest magnitude first, from which the kernel derives the power-of-two scale of its fp16 gradient images.
 * Preconditions of the FORWARD (fp16 range): |input| < 255, |weight| < 1023, |hidden activation| < 454 (activations are
 * carried times 100 log2(e) = 144.27: the Softplus(beta = 100) then needs no multiply after its logarithm).  A violation
 * overflows to inf / nan, never to a wrong finite number, and is counted in status[RSDF_STATUS_X2_FWD_NONFINITE]
 * (the reference's fp32 network, models/network_utils.py:109-157, stays finite there: the Python side raises, naming
 * RSDF_X2=0, the range-free kernels above).
 * Dynamic range of the BACKWARD: the gradient images share one power-of-two scale per launch, so a row keeps 22 bits down to
 * 2^-15 of the launch's largest row, 11 bits to 2^-28, nothing below 2^-38 of it.  With reroute = 1 (needs parts = 2,
 * d_planes and x7t_scratch [7][n][3] floats) every launch is guarded on the device: it counts the non-zero rows of d_sdf7t and
 * those within 2^-20 of the launch bound, and when fewer than 1 / 1024 of them are (the bound was set by outliers, e.g.
 * saturated samples behind lib/nerfacc/cuda/csrc/render_weight.cu:139-151's 1 / max(1 - alpha, 1e-10)) THAT launch runs on
 * the range-free kernels of rsdf_sdfmlp_fd7_bwd instead, from planes rebuilt out of the image in place of d_planes -- no host
 * read either way (status[RSDF_STATUS_X2_BWD_REROUTED] counts them).
 * parts = 2 is the fp32-equivalent form above.  parts = 1 is the 16-bit mode of BASELINE.json configs[4] ("bf16 MLP on
 * MFMA") for this node: the lo parts are dropped everywhere -- operands rounded ONCE to fp16 (11 significant bits against
 * bf16's 8: the finite-difference normal divides a difference of these values by eps), one matrix instruction per
 * product, the image [tile][tap][1][36][32] (504 B per sample); fp32 accumulation, master weights and gradients. */
int64_t rsdf_x2_rows(int64_t n_samples);
int64_t rsdf_x2_bytes(int64_t n_samples, int parts);
int rsdf_hashgrid_fwd_fd7_x2(const float *x7t /*or NULL*/, const float *points /*or NULL*/, float radius, float eps,
                             const float *table, const rsdf_grid_meta *meta /*host*/, int64_t n_samples,
                             int n_active_levels, float xyz_scale, float xyz_offset, int parts, void *x2, void *stream);
int rsdf_sdfmlp_fd7_x2_supported(int K0, int H, int N2);
int rsdf_sdfmlp_fd7_fwd_x2(const void *x2, int parts, int n_levels, int H, int N2, const float *w0, const float *b0,
                           const float *w1, const float *b1, const float *w2, const float *b2, int64_t n_samples,
                           float *sdf7t, float *feature, float *h2c, int *status /*nullable*/, void *stream);
int rsdf_sdfmlp_fd7_bwd_x2(const void *x2, int parts, int n_levels, int n_active_levels, int H, int N2, const float *w0,
                           const float *b0, const float *w1, const float *b1, const float *w2, const float *b2,
                           int64_t n_samples, const float *d_sdf7t, const float *d_feature,
                           float *dh2c_scratch /*nullable*/, void *guard_scratch /*32 bytes*/,
                           float *x7t_scratch /*[7][n][3]; nullable when reroute = 0*/,
                           int reroute /*0 never, 1 guarded (decided per launch on the device), 2 always*/, float *d_planes,
                           float *dw0, float *db0, float *dw1, float *db1, float *dw2, float *db2, int *status /*nullable*/,
                           void *stream);
/* ---- T3, layer PAIRS of the 128-wide radiance networks (models/texture.py:237-327: Linear(K,128) ReLU Linear(128,128) ReLU
 * ... ; csrc/mlp_pair.hip): two layers per kernel in the x2 number format above.  The odd activation never leaves the CU
 * (forward) or is recomputed from the pair's input (backward); a pair's input and output cross HBM once each way as the
 * "pair image": [tile = row / 32][part 2][chunk = column / 8 (16)][row ^ 12 (chunk & 1)][8 columns] fp16, values x 2^6, 16 KB
 * per 32-row tile (rsdf_pair_image_bytes), the same 4 bytes per value as fp32 rows.
 *   rsdf_pair_pack    fp32 rows x [n][ldx] (columns [0, K), K <= 128) -> image (columns >= K: zeros)
 *   rsdf_pair_unpack  image -> fp32 rows [n][128]
 *   rsdf_pair_fwd     hb = relu(Wb relu(Wa x + ba) + bb); Wa [128][K], Wb [128][128] row-major (nn.Linear.weight); out_image
 *                     (for the next pair) and / or out_rows [n][128] (for the per-layer kernels of the narrow output layer)
 *   rsdf_pair_bwd     g = d hb [n][128] fp32 rows (g_masked != 0: already multiplied by hb > 0, as a pair above writes it;
 *                     hb_rows != NULL: the forward's own hb as fp32 rows supplies that mask; neither: hb is recomputed);
 *                     ``bound``: device word(s) holding the bits of a float >= max |g| (rsdf_pair_bound_from_rows scans g;
 *                     rsdf_pair_bound_from_out_layer derives it from the narrow output layer's dz [n][N2] and weights, 8 bytes
 *                     of scratch; a pair's own dx_absmax output serves the pair below); dx [n][lddx] columns [0, kout)
 *                     (nullable), multiplied by (x > 0) when x_relu != 0 (x is the ReLU output of the pair below, so that dx
 *                     IS that pair's masked gradient); dWa [128][K], dba, dWb, dbb are ACCUMULATED into (zero them first).
 * Ranges: |x|, |activation|, |weight| < 1023 (fp16 class scales 2^6); a violation gives inf / nan outputs and is counted in
 * status[RSDF_STATUS_X2_FWD_NONFINITE].  The gradient images share one power-of-two scale per launch (from ``bound``). */
int rsdf_pair_supported(int K, int Na, int Nb);
int64_t rsdf_pair_image_bytes(int64_t n_rows);
int rsdf_pair_pack(const float *x, int ldx, int K, int64_t n, void *image, int *status /*nullable*/, void *stream);
/* columns [0, K1) from x1, [K1, K1 + K2) from x2: the reference's torch.cat([feature, encoding], -1) (models/texture.py:299-313)
 * without materialising it; image chunks past the last 32-column group that holds a column are neither written nor read */
int rsdf_pair_pack2(const float *x1, int ld1, int K1, const float *x2 /*nullable with K2 = 0*/, int ld2, int K2, int64_t n,
                    void *image, int *status /*nullable*/, void *stream);
int rsdf_pair_unpack(const void *image, int64_t n, float *rows, void *stream);
int rsdf_pair_fwd(const void *x_image, int K, const float *wa, const float *ba, const float *wb, const float *bb, int64_t n,
                  void *out_image /*nullable*/, float *out_rows /*nullable*/, const float *w_out /*[N2][128], nullable*/,
                  const float *b_out, int N2 /*<= 8*/, int out_act /*RSDF_ACT_NONE | RSDF_ACT_SIGMOID*/,
                  float *y_out /*[n][N2], nullable: the network's narrow output layer folded in*/, int *status /*nullable*/,
                  void *stream);
int rsdf_pair_bound_from_rows(const float *g, int64_t count, void *bound /*4 bytes*/, void *stream);
/* db_out (nullable, N2 <= 8): the output layer's bias gradient, the column sums of dz_out, ACCUMULATED into db_out [N2] by the
 * same pass over dz_out that finds its maximum (one torch column reduction per network and chunk less) */
int rsdf_pair_bound_from_out_layer(const float *dz_out, int64_t n, int N2, const float *w_out /*[N2][128]*/,
                                   void *bound /*8 bytes*/, float *db_out /*nullable*/, void *stream);
int rsdf_pair_bwd(const void *x_image, int K, const float *wa, const float *ba, const float *wb, const float *bb, int64_t n,
                  const float *g /*nullable with dz_out*/, int g_masked, const float *hb_rows /*nullable*/,
                  const float *dz_out /*[n][N2], nullable: g = dz_out @ w_out is formed in the kernel*/,
                  const float *w_out /*[N2][128]*/, int N2 /*<= 8*/,
                  float *dw_out /*nullable: dW_out [N2][128] += dz_out^T hb_rows, accumulated*/, const void *bound,
                  float *dx /*nullable*/, int lddx, int kout,
                  float *dx2 /*nullable: columns [k1, kout) of the input gradient go to dx2 [n][ld2] instead (the two sources of
                               rsdf_pair_pack2 get their gradients as two contiguous tensors); k1 a multiple of 4*/,
                  int ld2, int k1,
                  int x_relu, void *dx_absmax /*nullable, 4 bytes, zeroed by the caller*/, float *dwa,
                  float *dba, float *dwb, float *dbb, void *stream);
/* The 16-bit mode of the layer-pair kernels (round 6; BASELINE.json configs[4] "bf16 MLP on MFMA", the networks of
 * configs/split-mixed-occ-tensoir.yaml:93-120 with ``precision: fp16 / bf16``): same arguments, images and fp32 tensors as
 * rsdf_pair_fwd / rsdf_pair_bwd, but every matrix operand is rounded ONCE to fp16 at its class scale (11 significant bits) and
 * each product is ONE v_mfma_f32_16x16x32_f16 with fp32 accumulation; only the hi part of an image is read or written. */
int rsdf_pair_fwd16(const void *x_image, int K, const float *wa, const float *ba, const float *wb, const float *bb, int64_t n,
                    void *out_image, float *out_rows, const float *w_out, const float *b_out, int N2, int out_act, float *y_out,
                    int *status, void *stream);
int rsdf_pair_bwd16(const void *x_image, int K, const float *wa, const float *ba, const float *wb, const float *bb, int64_t n,
                    const float *g, int g_masked, const float *hb_rows, const float *dz_out, const float *w_out, int N2,
                    float *dw_out, const void *bound, float *dx, int lddx, int kout, float *dx2, int ld2, int k1, int x_relu,
                    void *dx_absmax, float *dwa, float *dba, float *dwb, float *dbb, void *stream);
/* ---- config[4]'s "bf16 MLP on MFMA" (BASELINE.json configs[4]; models/network_utils.py:109-157 at reduced matrix
 * precision): the same entry points with the suffix _bf16.  Same arguments, layouts and fp32 tensors; every matrix
 * operand (weights, activations, gradients) is rounded ONCE to bf16 (round to nearest even) and each k-step is ONE
 * v_mfma_f32_*_bf16 product with fp32 accumulation; master weights, biases, activations in memory and weight-gradient
 * accumulators stay fp32.  Opt-in per network (``precision: bf16`` in a network's config node on the Python side); the
 * default entry points above compute fp32-equivalent products. */
int rsdf_linear_fwd_bf16(const float *x, int ldx, const float *w, const float *b, int64_t n, int K,
                         int N, int act, float *y, int ldy, void *stream);
int rsdf_linear_bwd_input_bf16(const float *dy, const float *y, int lddy, const float *w, int64_t n,
                               int K, int N, int act, int k0, int Kout, float *dz, float *dx, int lddx,
                               void *stream);
int rsdf_linear_bwd_weight_bf16(const float *dz, int lddz, const float *x, int ldx, int64_t n, int K,
                                int N, float *dw, float *db, void *stream);
int rsdf_linear_bwd_fused_supported_bf16(int K, int N);
int rsdf_linear_bwd_fused_bf16(const float *dy, const float *y, int lddy, const float *x, int ldx,
                               const float *w, int64_t n, int K, int N, int act, int k0, int Kout, float *dx,
                               int lddx, int prev_act, float *dw, float *db, void *stream);
int rsdf_linear_bwd_fused_tail_bf16(const float *dz_out, int N2, const float *w2, const float *y, int lddy,
                                    const float *x, int ldx, const float *w, int64_t n, int K, int N, int act,
                                    int k0, int Kout, float *dx, int lddx, int prev_act, float *dw, float *db,
                                    void *stream);
int rsdf_linear_bwd_fused_ws_bf16(const float *dy, const float *y, int lddy, const float *x, int ldx,
                                  const float *w, int64_t n, int K, int N, int act, int k0, int Kout, float *dx,
                                  int lddx, int prev_act, float *dw, float *db, void *workspace,
                                  int64_t workspace_bytes, void *stream);
int rsdf_linear_bwd_fused_tail_ws_bf16(const float *dz_out, int N2, const float *w2, const float *y, int lddy,
                                       const float *x, int ldx, const float *w, int64_t n, int K, int N, int act,
                                       int k0, int Kout, float *dx, int lddx, int prev_act, float *dw, float *db,
                                       void *workspace, int64_t workspace_bytes, void *stream);
int rsdf_sdfmlp_fd7_supported_bf16(int K0, int H, int N2);
int rsdf_sdfmlp_fd7_fwd_bf16(const float *x7t, const float *planes, int n_levels, int n_active_levels,
                             float xyz_scale, float xyz_offset, int H, int N2, const float *w0,
                             const float *b0, const float *w1, const float *b1, const float *w2,
                             const float *b2, int64_t n_samples, float *sdf7t, float *feature, float *h2c,
                             void *stream);
int rsdf_sdfmlp_fd7_bwd_bf16(const float *x7t, const float *planes, int n_levels, int n_active_levels,
                             float xyz_scale, float xyz_offset, int H, int N2, const float *w0,
                             const float *b0, const float *w1, const float *b1, const float *w2,
                             const float *b2, int64_t n_samples, const float *d_sdf7t,
                             const float *d_feature, float *dh2c_scratch /*nullable*/, float *d_planes, float *dw0,
                             float *db0, float *dw1, float *db1, float *dw2, float *db2, void *stream);
/* weight_norm (torch.nn.utils.weight_norm dim=0): w = g * v / ||v||_row */
int rsdf_weight_norm_fwd(const float *g, const float *v, int N, int K, float *w, void *stream);
int rsdf_weight_norm_bwd(const float *g, const float *v, const float *dw, int N, int K, float *dg,
                         float *dv, void *stream);

/* ---- P1/H4/A1: sample positions, finite-difference normals, NeuS alpha ---------------------------
 * rsdf_fd_points: positions = o[ri] + d[ri]*(t0+t1)/2 (models/split_mixed_occ.py:229-231), the six
 * taps x +- eps*e_k clamped to +-radius (models/geometry.py:229-241) and the AABB contraction
 * (x+r)/(2r) (geometry.py:17-19, models/utils.py:109-114).  x_unit [n,7,3] (tap_major = 0) or
 * [7,n,3] (tap_major = 1): tap 0 = centre, then +x,-x,+y,-y,+z,-z.  positions (nullable) [n,3]; x_unit nullable when
 * positions are asked for (the *_pts / *_x2 stencil kernels derive the taps from the positions: 12 instead of 96 B per sample). */
int rsdf_fd_points(const float *rays_o, const float *rays_d, const int64_t *ray_indices,
                   const float *t_starts, const float *t_ends, int64_t n, float radius, float eps,
                   float *x_unit, float *positions, int tap_major, void *stream);
/* Same taps from explicit world-space points (VolumeSDF.forward(points), models/geometry.py:206). */
int rsdf_fd_taps(const float *points, int64_t n, float radius, float eps, float *x_unit,
                 void *stream);
/* grad = 0.5*(f+ - f-)/eps from the 7 tap values (models/geometry.py:243); sdf nullable.
 * Backward writes column 0 of d_sdf7 rows (the other columns are left untouched). */
int rsdf_fd_gradient_fwd(const float *sdf7, int ld, float eps, int64_t n, float *sdf, float *grad,
                         void *stream);
int rsdf_fd_gradient_bwd(const float *d_sdf, const float *d_grad, float eps, int64_t n,
                         float *d_sdf7, int ld, void *stream);
/* sdf7: column 0 of the MLP output for the 7 taps: rows 7i+t of a [7n, ld] matrix (ld > 0), or the
 * tap-major [7][n] array when ld = -1 (RSDF_TAP_MAJOR) ->
 * sdf, grad = 0.5*(f+ - f-)/eps (geometry.py:243), normal = grad/max(|grad|,1e-6)
 * (split_mixed_occ.py:237), alpha (split_mixed_occ.py:151-177).  variance: device scalar,
 * inv_s = clip(exp(10 v),1e-6,1e6).  dirs are gathered from rays_d by ray index; dists = t1-t0. */
int rsdf_neus_alpha_fd_fwd(const float *sdf7, int ld, const float *rays_d,
                           const int64_t *ray_indices, const float *t_starts,
                           const float *t_ends, const float *variance, float cos_anneal_ratio,
                           float eps, int64_t n, float *sdf, float *grad, float *normal,
                           float *alpha, void *stream);
/* backward: given d_alpha, d_normal, d_sdf, d_grad (each nullable) -> d_sdf7 [n,7] (row stride ld,
 * fully written) and d_variance += (device scalar, atomically accumulated: zero first). */
int rsdf_neus_alpha_fd_bwd(const float *sdf7, int ld, const float *rays_d,
                           const int64_t *ray_indices, const float *t_starts,
                           const float *t_ends, const float *variance, float cos_anneal_ratio,
                           float eps, int64_t n, const float *d_alpha, const float *d_normal,
                           const float *d_sdf, const float *d_grad, float *d_sdf7, int ld_out,
                           float *d_variance, void *stream);
/* get_alpha with explicit normals/dirs/dists (the reference signature, split_mixed_occ.py:151) */
int rsdf_neus_alpha_fwd(const float *sdf, const float *normal, const float *dirs,
                        const float *dists, const float *variance, float cos_anneal_ratio,
                        int64_t n, float *alpha, void *stream);
/* A2: occ_eval_fn (models/split_mixed_occ.py:108-119): alpha of the occupancy-grid update = the formula above with
 * cos == -1 and dists == render_step_size, from the SDF at the cell points (no gradient). */
int rsdf_neus_occ_alpha(const float *sdf, const float *variance, float render_step_size, int64_t n, float *alpha,
                        void *stream);
int rsdf_neus_alpha_bwd(const float *sdf, const float *normal, const float *dirs,
                        const float *dists, const float *variance, float cos_anneal_ratio,
                        int64_t n, const float *d_alpha, float *d_sdf, float *d_normal,
                        float *d_variance, void *stream);

/* ---- T1/T2/S1/O1: per-sample kernels of the radiance branch -----------------------------------------
 * rsdf_freq_encode: VanillaFrequency (models/network_utils.py:14-40): out[:, col_off + 6k + 3f + c] =
 *   {sin,cos}_f(2^k (x_c*x_scale + x_offset)) * mask[k]   (mask nullable device [n_frequencies]; n_frequencies <= 24: a
 *   workgroup's rows are staged in LDS and stored along the rows).
 * rsdf_sh_encode_*: tcnn.Encoding(otype SphericalHarmonics, degree <= 5) (models/network_utils.py:98-99;
 *   call sites models/texture.py:312,348): real SH of 2*d01-1, degree^2 outputs; backward w.r.t. d01.
 * rsdf_reflect_*: wi = -dirs, wo = 2(wi.n)n - wi, written as wo01 = (wo+1)/2, nov = n.wi
 *   (models/texture.py:295-297,312); backward w.r.t. the normals.
 * rsdf_split_color0_*: stage-0 output of VolumeMixedMipSplitOcc.forward (models/texture.py:303-327):
 *   colors7 = [(1-blend) sigmoid(albedo6[:3]), blend sigmoid(spec3), blend], blend = sigmoid(metallic2[0]).
 * rsdf_rgb_to_srgb_*: lib/pbr/utils/nvdiffrecmc_util.py:95-103 (elementwise, n = number of floats). */
int rsdf_freq_encode(const float *x, int64_t n, int n_frequencies, float x_scale, float x_offset,
                     const float *mask, float *out, int ld_out, int col_off, void *stream);
int rsdf_sh_encode_fwd(const float *d01, int64_t n, int degree, float *out, int ld_out, int col_off,
                       void *stream);
int rsdf_sh_encode_bwd(const float *d01, const float *dout, int64_t n, int degree, int ld_dout,
                       int col_off, float *d_d01, void *stream);
int rsdf_reflect_fwd(const float *dirs, const float *normals, int64_t n, float *wo01, float *nov,
                     void *stream);
int rsdf_reflect_bwd(const float *dirs, const float *normals, int64_t n, const float *d_wo01,
                     const float *d_nov, float *d_normals, void *stream);
int rsdf_split_color0_fwd(const float *albedo6, const float *metallic2, const float *spec3, int64_t n,
                          float *colors7, void *stream);
int rsdf_split_color0_bwd(const float *albedo6, const float *metallic2, const float *spec3,
                          const float *d_colors7, int64_t n, float *d_albedo6, float *d_metallic2,
                          float *d_spec3, void *stream);
int rsdf_rgb_to_srgb_fwd(const float *x, int64_t n, float *y, void *stream);
int rsdf_rgb_to_srgb_bwd(const float *x, const float *dy, int64_t n, float *dx, void *stream);
/* H4, the analytic-gradient sweep (models/geometry.py:224-228 obtains it through autograd of the Softplus(beta=100) network,
 * models/network_utils.py:128-134): h = softplus(z, beta = 100, threshold = 20) and slope = sigmoid(100 z) = dh/dz of n floats in
 * one pass; backward dz = dh slope + dslope 100 slope (1 - slope) (dh / dslope nullable, not both). */
int rsdf_softplus100_slope_fwd(const float *z, int64_t n, float *h, float *slope, void *stream);
int rsdf_softplus100_slope_bwd(const float *slope, const float *dh /*nullable*/, const float *dslope /*nullable*/, int64_t n,
                               float *dz, void *stream);
/* O1, models/split_mixed_occ.py:405-436: y [n,3] = clamp(rgb_to_srgb(comp [n,3] + bg [3] * (1 - opacity [n])), 0, 1) in one
 * pass; backward writes d_comp [n,3] and d_opacity [n] (nullable).  Same values as the unfused chain. */
int rsdf_compose_srgb_fwd(const float *comp, const float *bg, const float *opacity, int64_t n, float *y, void *stream);
int rsdf_compose_srgb_bwd(const float *comp, const float *bg, const float *opacity, const float *dy, int64_t n,
                          float *d_comp, float *d_opacity /* nullable */, void *stream);

/* stage-1 split-sum shading (models/texture.py:329-345) on ACTIVATED material values (sigmoid already
 * applied: albedo6 = [diff_rgb | albedo], metallic2 = [blend | metallic]); colors24 in the channel order of
 * texture.py:345 (consumed by the slices at models/split_mixed_occ.py:293-304). */
int rsdf_split_shade1_fwd(const float *albedo6, const float *roughness, const float *metallic2,
                          const float *spec3, const float *diffuse_light, const float *specular_light,
                          const float *fg, int64_t n, float *colors24, void *stream);
int rsdf_split_shade1_bwd(const float *albedo6, const float *metallic2, const float *spec3,
                          const float *diffuse_light, const float *specular_light, const float *fg,
                          const float *d_colors24, int64_t n, float *d_albedo6, float *d_roughness,
                          float *d_metallic2, float *d_spec3, float *d_diffuse_light,
                          float *d_specular_light, float *d_fg, void *stream);

/* ---- S4/E1: environment light ----------------------------------------------------------------------
 * Prefilters replace the renderutils plugin (lib/renderutils/c_src/torch_bindings.cpp:740-890, kernels
 * cubemap.cu:110-350; Python wrappers lib/renderutils/ops.py:391-458).  Cube maps are NHWC [6,R,R,3] fp32.
 *   diffuse:  out = sum_L clamp(N.L,0,.999) area(L)/3.141592 c(L); bwd is the adjoint as a GATHER.
 *   bounds:   [6,R,R,24] float (per face xmin,xmax,ymin,ymax of texels with L.V >= cos_cutoff).
 *   specular: out4 = [sum w c (3), sum w]; bwd (gather over the same window) takes grad_out with
 *             grad_channels floats per texel (first 3 used) and returns grad_cubemap [6,R,R,3].
 *   avgpool:  2x2 average (cubemap_mip forward, lib/pbr/utils/light_utils.py:94-98).
 * Cube lookups replace dr.texture(boundary_mode='cube') (lib/pbr/light.py:194-206; nvdiffrast absent:
 * definition in oracle/envlight.py).  mips: HOST array of n_mips device pointers, level l is
 * [6, R0>>l, R0>>l, C]; level (nullable => level 0) is the mip_level_bias per sample; linear between the
 * two nearest levels.  bwd: grad_mips (host array, entries nullable) are accumulated atomically. */
int rsdf_diffuse_cubemap_fwd(const float *cubemap, int R, float *out, void *stream);
int rsdf_diffuse_cubemap_bwd(const float *grad_out, int R, float *grad_cubemap, void *stream);
int rsdf_specular_bounds(int R, float cos_cutoff, float *bounds, void *stream);
int rsdf_cubemap_texel_table(int R, float *table /* [6,R,R,4]: unit direction, solid angle / 4 */, void *stream);
int rsdf_specular_cubemap_fwd(const float *cubemap, const float *bounds, const float *texel_table /*nullable*/,
                              int R, float roughness, float cos_cutoff, float *out4, void *stream);
/* The forward with lib/renderutils/ops.py:458's normalisation inside (round 6): out3 [6,R,R,3] = sum w c / sum w, contiguous,
 * and wsum [6,R,R] = sum w (what the backward divides the incoming gradient by).  Additive entry point. */
int rsdf_specular_cubemap_fwd_norm(const float *cubemap, const float *bounds, const float *texel_table /*nullable*/, int R,
                                   float roughness, float cos_cutoff, float *out3, float *wsum, void *stream);
int rsdf_specular_cubemap_bwd(const float *grad_out, int grad_channels, const float *bounds,
                              const float *texel_table /*nullable*/, int R, float roughness, float cos_cutoff,
                              float *grad_cubemap, void *stream);
int rsdf_cubemap_avgpool(const float *cubemap, int R, int C, float *out, void *stream);
int rsdf_cube_sample_fwd(const float *const *mips /*host array*/, int n_mips, int R0, int C,
                         const float *dirs, const float *level, int64_t n, float *out, void *stream);
int rsdf_cube_sample_bwd(const float *const *mips /*host array*/, float *const *grad_mips /*host array*/,
                         int n_mips, int R0, int C, const float *dirs, const float *level, int64_t n,
                         const float *grad_out, float *grad_dirs, float *grad_level, void *stream);

/* ---- S2/S3: bilinear 2-D grid_sample with first- and second-order gradients ------------------------
 * replaces aten grid_sample / grid_sampler_2d_backward as used by utils/cuda_gridsample.py:25-73 and
 * grad2_2d (lib/grid_sample_grad2/gridsample_cuda.cpp:26-37, kernel gridsample_cuda.cu:27-210).
 * input [N,C,H,W], grid [N,Ho,Wo,2] (x,y in [-1,1]), output [N,C,Ho,Wo]; padding_border: 0 zeros, 1
 * border; aten coordinate conventions.  grad_input / g_input are ATOMICALLY accumulated (zero first);
 * nullable outputs are skipped.  bwd2 returns d/d(grad_output), d/d(input), d/d(grid) of
 * <g2_input, grad_input> + <g2_grid, grad_grid>. */
int rsdf_grid_sample2d_fwd(const float *input, const float *grid, int N, int C, int H, int W, int Ho,
                           int Wo, int padding_border, int align_corners, float *output, void *stream);
int rsdf_grid_sample2d_bwd(const float *grad_output, const float *input, const float *grid, int N, int C,
                           int H, int W, int Ho, int Wo, int padding_border, int align_corners,
                           float *grad_input, float *grad_grid, void *stream);
int rsdf_grid_sample2d_bwd2(const float *g2_input, const float *g2_grid, const float *grad_output,
                            const float *input, const float *grid, int N, int C, int H, int W, int Ho,
                            int Wo, int padding_border, int align_corners, float *gg_out, float *g_input,
                            float *g_grid, void *stream);

/* ---- N2: ray generation + pixel gather (systems/split_occ.py:58-131 train branch, models/ray_utils.py:32-56) ----
 * index [n_index] (n_index = 1: one view for the batch, else n), y/x [n] int64 pixel coordinates;
 * directions [H,W,3] (dirs_per_view = 0) or [V,H,W,3]; c2w [V,3,4]; images [V,H,W,channels]; fg_masks [V,H,W].
 * rays [n,6] = (c2w[:,3], normalize(R d)); rgb [n,channels] (nullable) = image pixel, with apply_mask
 * rgb*m + rgb_to_srgb(bg*(1-m)) (:113-116); fg_mask [n] (nullable). */
int rsdf_gen_rays(const int64_t *index, int64_t n_index, const int64_t *y, const int64_t *x,
                  const float *directions, int dirs_per_view, const float *c2w, const float *images, int channels,
                  const float *fg_masks, const float *background_color, int apply_mask, int H, int W, int64_t n,
                  float *rays, float *rgb, float *fg_mask, void *stream);

/* ---- M2: occupancy-grid update (lib/nerfacc/grid.py:196-239; live call models/split_mixed_occ.py:126-131) ----
 * rsdf_occ_cell_points: x[n,3] = (cell_coords(indices or 0..n-1) + jitter) / res * (roi_max - roi_min) + roi_min.
 * rsdf_occ_update: occs[idx] = max(occs[idx]*ema_decay, occ_i) (duplicates: max of their candidates), then
 * binary[c] = occs[c] > min(mean(occs), occ_thre).  occ >= 0.  scratch: rsdf_occ_update_scratch_bytes(n_cells). */
int rsdf_occ_cell_points(const int64_t *indices /*nullable: all cells*/, const float *jitter, const float *roi,
                         int res_x, int res_y, int res_z, int64_t n, float *x, void *stream);
int64_t rsdf_occ_update_scratch_bytes(int64_t n_cells);
int rsdf_occ_update(const int64_t *indices /*nullable*/, const float *occ, int64_t n, float ema_decay,
                    float occ_thre, int64_t n_cells, float *occs, uint8_t *binary, void *scratch, void *stream);

/* ---- N1: loss tail on the path's outputs (systems/split_occ.py:163-215; criterions.py:155-159) -------------
 * rays_fwd: raw fp64 sums [7] (ACCUMULATES: zero first) = { sum d^2, sum |d| over valid rays x 3 channels of
 *   comp_rgb - target; the same two for comp_rgb_phys (nullable); number of valid rays; sum of mask-BCE terms
 *   (fg_mask nullable); sum of "opaque" BCE terms } with opacity clamped to [1e-3, 1-1e-3].
 * rays_bwd: coef6 (device) = { mse, l1, phys mse, phys l1 weights each / (3 valid), mask / N, opaque / N }.
 * samples_fwd: sums [3] = { sum (||grad|| - 1)^2, sum exp(-scale |sdf|), sum |laplace| (nullable) };
 * samples_bwd: coef3 (device) = weights / S. */
int rsdf_loss_rays_fwd(const float *comp_rgb, const float *comp_rgb_phys, const float *target,
                       const uint8_t *rays_valid, const float *opacity, const float *fg_mask, int64_t n,
                       double *sums7, void *stream);
int rsdf_loss_rays_bwd(const float *comp_rgb, const float *comp_rgb_phys, const float *target,
                       const uint8_t *rays_valid, const float *opacity, const float *fg_mask, const float *coef6,
                       int64_t n, float *d_comp_rgb, float *d_comp_rgb_phys, float *d_opacity, void *stream);
int rsdf_loss_samples_fwd(const float *sdf, const float *sdf_grad, const float *laplace, float sparsity_scale,
                          int64_t n, double *sums3, void *stream);
int rsdf_loss_samples_bwd(const float *sdf, const float *sdf_grad, const float *laplace, float sparsity_scale,
                          const float *coef3, int64_t n, float *d_sdf, float *d_sdf_grad, float *d_laplace,
                          void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RISESDF_HIP_H */
