/*
 * occnerf_hip.h -- C ABI of the MI355X (gfx950) implementation of OccNeRF's per-ray
 * volumetric rendering hot path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name starts with h_ (host);
 *   - caller owns all buffers, outputs are caller-allocated (the reference's FFI
 *     convention, gridencoder/src/gridencoder.cu:448-503);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); launches are
 *     asynchronous, nothing here synchronises;
 *   - return value: 0 on success, non-zero on error (invalid argument or a HIP error);
 *     occnerf_last_error() returns a thread-local description.  The Python host layer
 *     turns a non-zero return into RuntimeError, matching the reference's TORCH_CHECK
 *     -> c10::Error -> RuntimeError behaviour (gridencoder.cu:15-18).
 *
 * Section 1 is the reference's own native operator interface for this path (the
 * `_gridencoder` pybind module).  Section 2 are the fused stages that sit behind the
 * reference's *module* seam (core/nets/occnerf/network.py `Network`,
 * canonical_mlps/occnerf_mlp.py `CanonicalMLP`): the reference evaluates them as chains
 * of torch ops and has no FFI for them; each entry cites the Python it replaces.
 */
#ifndef OCCNERF_HIP_H
#define OCCNERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an existing export's signature or a table layout changes (2: composite/msknn_clustered/
 * sample_features grew arguments in round 2, the Adam table row carries per-tensor bias corrections; 3: msknn_clustered
 * takes the cluster groups; 4: occnerf_agg_backward takes a scratch buffer; 5: round 6 -- the experiment knobs cohab_lds,
 * features_small, features_rowcache, linear_resident and split_refill and the kernels behind them left the library; the two
 * f16x3 calls take a domain flag). */
#define OCCNERF_ABI_VERSION 5

int occnerf_abi_version(void);
const char *occnerf_last_error(void);

/* Tuning knobs.  Two remain, both choosing between shipped forms with identical results: "agg_slices" (OCCNERF_AGG_SLICES,
 * 0 = automatic, else the sample slices of occnerf_agg_backward) and "grid_xcd" (OCCNERF_GRID_XCD: the operator-level D4C2
 * forward with the level pairs dealt to the XCDs -- 0: from 32 768 samples up (the default), 1: always, 2: never;
 * profiles/r05_xcd_levels.md).  Each is read from its environment variable once, at first use, and clamped to its valid range;
 * this call reads (value < 0) or sets it afterwards.  Returns the previous value, -1 for an unknown name.  No launch in this
 * library alters its outputs for diagnosis.  No counterpart in the reference. */
int occnerf_experiment_knob(const char *name, int value);

/* ------------------------------------------------------------------------------------
 * 1. Grid encoder -- replaces core/nets/occnerf/gridencoder/src/bindings.cpp:5-9
 *    (prototypes gridencoder.h:12-15, kernels gridencoder.cu:87-369,506-645).
 *    Argument order and meaning are the reference's; at::Tensor -> pointer, and the
 *    trailing stream is new (the reference launches on the legacy default stream).
 *    D in {2,3,4,5}, C in {1,2,4,8}.  The reference dispatches on the tensors' dtype
 *    (AT_DISPATCH_FLOATING_TYPES_AND_HALF, gridencoder.cu:467,500); a C ABI has no tensor to
 *    ask, so the dtype is in the entry point's name: no suffix = float32 (what the rendering
 *    path uses), _f16 = at::Half (what grid.py:44-45 feeds under autocast).  float64 is not
 *    built (nothing on the path produces double embeddings).
 * ---------------------------------------------------------------------------------- */

/* inputs[B,D] in [0,1] (rows outside -> zeros); embeddings[sO,C]; offsets[L+1] int32;
 * outputs[L,B,C] written in place; dy_dx[B,L*D*C] or NULL.
 * S = log2(per_level_scale), H = base resolution, gridtype 0 hash / 1 tiled,
 * interp 0 linear / 1 smoothstep.                         gridencoder.cu:448-471 */
int occnerf_grid_encode_forward(const float *inputs, const float *embeddings,
                                const int32_t *offsets, float *outputs, uint32_t B, uint32_t D,
                                uint32_t C, uint32_t L, float S, uint32_t H, float *dy_dx,
                                uint32_t gridtype, int align_corners, uint32_t interp,
                                void *stream);

/* grad[L,B,C]; grad_embeddings[sO,C] must be zero-filled by the caller and is
 * accumulated with fp32 atomics; dy_dx / grad_inputs[B,D] optional (both or neither).
 *                                                          gridencoder.cu:473-503 */
int occnerf_grid_encode_backward(const float *grad, const float *inputs, const float *embeddings,
                                 const int32_t *offsets, float *grad_embeddings, uint32_t B,
                                 uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                 const float *dy_dx, float *grad_inputs, uint32_t gridtype,
                                 int align_corners, uint32_t interp, void *stream);

/* occnerf_grid_encode_forward with the level offsets also given as a HOST array h_offsets[L+1]: the D = 4, C = 2 hash
 * encoder without dy_dx (what the canonical MLP evaluates on every sample) then runs as 8 lanes per sample x 2 levels
 * per lane with the per-level index mode (dense / power-of-two mask / generic) decided on the host; identical
 * results.  Everything else falls through to the reference-shaped kernel. */
int occnerf_grid_encode_forward_h(const float *inputs, const float *embeddings, const int32_t *offsets,
                                  const int32_t *h_offsets, float *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                  float S, uint32_t H, float *dy_dx, uint32_t gridtype, int align_corners,
                                  uint32_t interp, void *stream);

/* Host-side helper of the module's backward (grid.py:69-90 permutes autograd's [B, L*C] gradient to [L,B,C] before the
 * operator call): the same transposition with the rows of RUNS of consecutive samples whose inputs are bitwise identical
 * summed into the run's first sample, zeros in the others (the tiled backward skips zero rows).  Same gradient (another
 * summation order); only when no input gradient is wanted.  No counterpart in the reference. */
int occnerf_grid_grad_runs(const float *grad_rows, const float *inputs, int64_t B, uint32_t D, uint32_t L, uint32_t C,
                           float *grad, void *stream);

/* occnerf_grid_encode_backward with the level offsets also given as a HOST array h_offsets[L+1].  The
 * reference's signature above cannot know the level sizes without reading device memory, so it always runs
 * the atomic scatter (gridencoder.cu:248-340 as written); with the host copy, large D = 4, C = 2 hash batches
 * take the atomics-free path (workgroup-owned LDS tiles of the table, fp64 accumulators).  scratch (optional, device,
 * scratch_bytes >= L * B * 8): room for the per-(level, sample) tile sets of the pre-pass that lets a tile-job hash only the
 * samples that touch it; without it every tile-job re-hashes every sample (same results). */
int occnerf_grid_encode_backward_h(const float *grad, const float *inputs, const float *embeddings,
                                   const int32_t *offsets, const int32_t *h_offsets, float *grad_embeddings,
                                   uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                   const float *dy_dx, float *grad_inputs, uint32_t gridtype, int align_corners,
                                   uint32_t interp, void *scratch, int64_t scratch_bytes, void *stream);

/* The at::Half dispatch case of the two operators above (gridencoder.cu:467,500).  embeddings, outputs, dy_dx, grad,
 * grad_embeddings, grad_inputs are IEEE binary16 arrays (torch.half); inputs stay float32, as in the reference
 * (`const float *inputs`, gridencoder.cu:373).  Arithmetic follows c10::Half exactly: cell position and corner weights in
 * float, every `scalar_t +=` a half-rounded term added in float and rounded to half (occnerf_amd/csrc/grid_encode_f16.hip
 * spells it out); the backward adds channel pairs with packed-half atomics (gridencoder.cu:323-331), so C must be even
 * there -- the reference's own C = 1 half path is an empty stub (:22-26) and grid.py:44 never selects it. */
int occnerf_grid_encode_forward_f16(const float *inputs, const void *embeddings, const int32_t *offsets, void *outputs,
                                    uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void *dy_dx,
                                    uint32_t gridtype, int align_corners, uint32_t interp, void *stream);
int occnerf_grid_encode_backward_f16(const void *grad, const float *inputs, const void *embeddings, const int32_t *offsets,
                                     void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                     uint32_t H, const void *dy_dx, void *grad_inputs, uint32_t gridtype, int align_corners,
                                     uint32_t interp, void *stream);

/* The double dispatch case (gridencoder.cu:467,500 with scalar_t = double): embeddings, outputs, dy_dx, grad, grad_embeddings,
 * grad_inputs are float64; inputs stay float32 and so do the cell position, the corner weights and pos_deriv, exactly as the
 * reference's templates leave them; products with the double tensors are formed in double and `r += a * b` is one fma
 * (csrc/grid_encode_f64.hip).  The backward scatters with global_atomic_add_f64.  With the two calls above this completes
 * AT_DISPATCH_FLOATING_TYPES_AND_HALF: no dtype the reference's `_gridencoder` accepts is refused. */
int occnerf_grid_encode_forward_f64(const float *inputs, const double *embeddings, const int32_t *offsets, double *outputs,
                                    uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, double *dy_dx,
                                    uint32_t gridtype, int align_corners, uint32_t interp, void *stream);
int occnerf_grid_encode_backward_f64(const double *grad, const float *inputs, const double *embeddings, const int32_t *offsets,
                                     double *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                     uint32_t H, const double *dy_dx, double *grad_inputs, uint32_t gridtype, int align_corners,
                                     uint32_t interp, void *stream);

/* Total-variation gradient, gridencoder.cu:506-645.  Never called by the reference's
 * trainer; exported for interface completeness and
 * returns an error ("not implemented") without touching its arguments. */
int occnerf_grad_total_variation(const float *inputs, const float *embeddings, float *grad,
                                 const int32_t *offsets, float weight, uint32_t B, uint32_t D,
                                 uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                                 int align_corners, void *stream);

/* ------------------------------------------------------------------------------------
 * 2. Sample pipeline (behind Network.forward / CanonicalMLP.forward)
 * ---------------------------------------------------------------------------------- */

/* Ray sampling + backward warp to canonical space.
 * Replaces network.py:405-432 (_unpack_ray_batch, _get_samples_along_ray,
 * _stratified_sampling), :456 (pts) and :351-402 (_sample_motion_fields).
 * rays[n,8] = (o, d, near, far); t_vals[S] = linspace(0,1,S); t_rand[n,S] or NULL;
 * Rs[nb,3,3], Ts[nb,3]: motion bases; vol[>=nb,G,G,G]: motion-weight volume (a trailing
 * background channel is ignored); h_bbox_min[3], h_bbox_scale[3] on the HOST.
 * Outputs z_vals[n,S], x_skel[n*S,3], mask[n*S]; pts[n*S,3] optional (NULL to skip). */
int occnerf_sample_warp(const float *rays, int64_t n, int32_t S, const float *t_vals,
                        const float *t_rand, const float *Rs, const float *Ts, const float *vol,
                        int32_t nb, int32_t G, const float *h_bbox_min, const float *h_bbox_scale,
                        float *z_vals, float *pts, float *x_skel, float *mask, void *stream);
/* The renderer's form (no jitter, no pts): same outputs, bit for bit, with the bones whose motion-weight channel cannot
 * reach a wave's 64 samples skipped.  boxes[nb,6] int32 = {x0,x1,y0,y1,z0,z1} of every channel's non-zero voxels, from
 * occnerf_bone_boxes on the same vol (x0 > x1: all zero).  The culling applies when S is a multiple of 64 (a wave's samples
 * then lie on one ray); otherwise the call behaves like occnerf_sample_warp. */
int occnerf_bone_boxes(const float *vol, int32_t nb, int32_t G, int32_t *boxes, void *stream);
int occnerf_sample_warp_culled(const float *rays, int64_t n, int32_t S, const float *t_vals,
                               const float *Rs, const float *Ts, const float *vol, int32_t nb, int32_t G,
                               const int32_t *boxes, const float *h_bbox_min, const float *h_bbox_scale,
                               float *z_vals, float *x_skel, float *mask, void *stream);

/* Render order of a frame's rays: the permutation that walks them along a 2-D Morton curve of their directions (projected on
 * the plane normal to the mean direction), so that 64 consecutive rays are a compact ~8x8 pixel patch (kNN tiles, hash-grid
 * gathers) and 256 a ~16x16 one (the block a rank is dealt).  No counterpart in the reference, whose ray order is the
 * dataset's (freeview.py:190-208 row-major pixels); rays are independent, results are returned in the caller's order.
 * dirs: R directions `stride` floats apart (3: a [R,3] array; 8: columns 3..5 of rays8); order[R] int64 out; temp:
 * occnerf_ray_order_temp_bytes(R) bytes.  Deterministic (fixed-order reductions + a stable radix sort): every rank of a
 * sharded render computes the same walk from the same frame. */
int64_t occnerf_ray_order_temp_bytes(int64_t R);
int occnerf_ray_order(const float *dirs, int64_t R, int64_t stride, int64_t *order, void *temp, int64_t temp_bytes,
                      void *stream);

/* Per-frame ray generation (one thread per pixel): camera_util.py:133-160 get_rays_from_KRT +
 * :163-212 rays_intersect_3d_bbox, as the reference's datasets call them (tpose.py:155-172,
 * freeview.py:190-208).  HOST inputs: h_Kinv[9] = K^-1, h_R[9], h_T[3] (E[:3,:3], E[:3,3]), row major,
 * h_bbox_min/max[3] = the observation-space skeleton bbox (the 0.01 growth is applied inside);
 * f32_camera != 0: K and E were float32 arrays, so numpy ran pixel -> ray in float32 (the reference's
 * synthetic tpose / freeview cameras); 0: float64 (calibrated dataset cameras).
 * Outputs for every pixel p = row * W + col: rays8[p] = (origin, direction with |d|<1e-5 clamped as the
 * reference clamps it, near, far) and mask[p] = 1 when the ray crosses the box (exactly two face hits);
 * near/far are 0 where mask is 0.  The caller compacts by mask (occnerf_amd/rays.py). */
int occnerf_gen_rays(const double *h_Kinv, const double *h_R, const double *h_T, int32_t f32_camera,
                     int32_t H, int32_t W, const double *h_bbox_min, const double *h_bbox_max, float *rays8,
                     uint8_t *mask, void *stream);

/* Non-rigid offset MLP (105 -> 128 x6 (skip @4) -> 3) with the Hann-windowed Fourier
 * embedding.  Replaces embedders/hannw_fourier.py:9-63 + mlp_offset.py:45-62 as called at
 * network.py:225-232.
 * occnerf_nonrigid_pack: h_W/h_b = HOST arrays of the 7 device weight/bias pointers
 * (block_mlps.{0,2,...,12}, torch layout) -> packed[occnerf_nonrigid_packed_floats()], once
 * per checkpoint.  occnerf_nonrigid: cond[69] is the frame's condition code (folded into the
 * layer-0 bias inside `packed`, which is therefore written per call), W0/b0 the layer-0
 * weight/bias, h_hann[6] the HOST window weights.  xyz_out may alias xyz_in. */
int64_t occnerf_nonrigid_packed_floats(void);
int occnerf_nonrigid_pack(const float *const *h_W, const float *const *h_b, float *packed,
                          void *stream);
int occnerf_nonrigid(const float *xyz_in, int64_t N, const float *cond, const float *h_hann,
                     const float *W0, const float *b0, float *packed, float *xyz_out,
                     void *stream);
/* occnerf_nonrigid, in place, on the samples rows[0 .. *n_dev) of xyz[N_max,3] only (the frame's live samples;
 * list and count in device memory, occnerf_live_rows). */
int occnerf_nonrigid_rows(float *xyz, int64_t N_max, const int32_t *rows, const int32_t *n_dev,
                          const float *cond, const float *h_hann, const float *W0, const float *b0,
                          float *packed, void *stream);
/* The first fp32 version (32-sample waves, weights straight from L2); same arguments and packed buffer,
 * same results to fp32 rounding.  Cross-check and A/B timing; occnerf_nonrigid is the LDS-staged
 * 16-sample-tile kernel. */
int occnerf_nonrigid_direct(const float *xyz_in, int64_t N, const float *cond, const float *h_hann,
                            const float *W0, const float *b0, float *packed, float *xyz_out,
                            void *stream);

/* bf16x3 variant of the non-rigid MLP (split-bf16 operands, fp32 accumulation; see
 * occnerf_canonical_mlp_bf16x3).  packed = the fp32 blob (biases, folded layer-0 bias, output
 * rows); packed_bf16 = occnerf_nonrigid_packed_bf16_bytes() zero-initialised bytes filled by
 * occnerf_nonrigid_pack_bf16 from the first 6 weight pointers. */
int64_t occnerf_nonrigid_packed_bf16_bytes(void);
int occnerf_nonrigid_pack_bf16(const float *const *h_W, void *packed_bf16, void *stream);
int occnerf_nonrigid_bf16x3(const float *xyz_in, int64_t N, const float *cond, const float *h_hann,
                            const float *W0, const float *b0, float *packed, const void *packed_bf16,
                            float *xyz_out, void *stream);
/* In place on the listed samples only (as occnerf_nonrigid_rows for the fp32 kernel): xyz[rows[i]] += offset for
 * i < *n_dev (device-side count, <= N_max). */
int occnerf_nonrigid_bf16x3_rows(float *xyz, int64_t N_max, const int32_t *rows, const int32_t *n_dev,
                                 const float *cond, const float *h_hann, const float *W0, const float *b0,
                                 float *packed, const void *packed_bf16, void *stream);

/* Multi-scale exact kNN (k = 10, up to 4 point sets).  Replaces the pykeops
 * Kmin_argKmin reduction of knn.py:77-85 and the index bookkeeping of network.py:235-255.
 * points[M,3]: the scales concatenated (scale 0 = all base points first);
 * h_scale_begin[nscale+1] host prefix offsets into points; index_map[M]: base-point index
 * of every row (identity for scale 0, fps_index for the others);
 * h_seed_from_coarser[nscale]: non-zero when scale s is a superset of scale s+1, which lets
 * the search bound its radius by the coarser result (results are identical either way).
 * knn_idxs[N,nscale,10] int32, ascending distance, ties -> lower row first. */
int occnerf_msknn(const float *xyz, int64_t N, const float *points, const int32_t *index_map,
                  const int32_t *h_scale_begin, const int32_t *h_seed_from_coarser,
                  int32_t nscale, int32_t *knn_idxs, void *stream);

/* The same search with cluster culling (same results, ~5x fewer distance evaluations on the
 * SMPL body).  Queries are xyz[n_rays * samples_per_ray, 3] in ray-major order; a wavefront
 * works on 64 consecutive rays x 4 consecutive samples (callers that order rays in compact pixel
 * patches get the most out of it; any order is correct) and skips every cluster of support
 * points that the triangle inequality puts outside all of its queries' search radii.
 * Layout (built by the host once per model, occnerf_amd/geometry.py::build_knn_clusters):
 * points[M,4] = every scale but the coarsest stored cluster by cluster (cluster = nearest
 * coarsest-scale point), then the coarsest scale in original order at rows
 * h_coarse_rows[0..1); segments padded to multiples of 4 rows with +inf points; .w (int bits) =
 * original row within the scale << 16 | base-point index: the row is the tie-break key among equal
 * distances, the base index (< 65536) is what knn_idxs reports; centers[ncl,4];
 * cluster_ranges[nscale-1,ncl,2] row ranges into points; cluster_radius[nscale-1,ncl] >=
 * max |p - center| per cluster.  group_centers[ngrp,4], group_ranges[ngrp,2], group_radius[nscale-1,ngrp] (nullable together,
 * ngrp = 0): the clusters are listed group by group -- group g = clusters group_ranges[g][0..1) -- and group_radius bounds
 * everything those clusters hold at a scale around group_centers[g] (< 0: nothing there), so a search tests the group spheres
 * first and only the clusters of the groups in reach; same results.  mask (nullable) [n_rays * samples_per_ray]: samples whose mask is
 * exactly 0 (motion-weight sum, network.py:330: their alpha is multiplied by it) are skipped and their
 * knn_idxs rows left unwritten.  query_rows / n_query_dev / ray_start (nullable, together, instead of mask): the ascending
 * list of the samples to query (occnerf_live_rows or the heads of occnerf_repeat_heads) with its length in device memory and
 * an int32[n_rays + 1] scratch; a lane then takes the next four LISTED samples of its ray instead of four fixed sample slots,
 * so tiles are full wherever their rays still have listed samples. */
int occnerf_msknn_clustered(const float *xyz, const float *mask, int64_t n_rays, int32_t samples_per_ray,
                            const float *points, const float *centers,
                            const int32_t *cluster_ranges, const float *cluster_radius, int32_t ncl,
                            const float *group_centers, const int32_t *group_ranges, const float *group_radius, int32_t ngrp,
                            const int32_t *h_coarse_rows, const int32_t *h_seed_from_coarser,
                            int32_t nscale, const int32_t *query_rows, const int32_t *n_query_dev, int32_t *ray_start,
                            int32_t *knn_idxs, void *stream);
/* Centre cache.  occnerf_knn_center searches ONE point c[3] (device memory) against the brute-force layout of occnerf_msknn
 * (points[M,4], index_map[M], h_scale_begin[nscale+1]) and writes center_out[4] = (c, r^2) and idx_out[nscale,10]: c's neighbours
 * and the squared radius inside which every query provably has the same neighbours in the same order (0: none, e.g. a tie among
 * c's 11 nearest).  occnerf_msknn_clustered_centered is occnerf_msknn_clustered with that pair (nullable together): queries with
 * |q - c|^2 < r^2 receive idx_out without a search, tiles that hold no other query are skipped.  Same knn_idxs, bit for bit.
 * The renderer passes c = the non-rigid offset of the origin, onto which the samples with a vanishing motion-weight sum
 * collapse (two thirds of the live samples of a frame). */
int occnerf_knn_center(const float *c, const float *points, const int32_t *index_map, const int32_t *h_scale_begin,
                       int32_t nscale, float *center_out, int32_t *idx_out, void *stream);
int occnerf_msknn_clustered_centered(const float *xyz, const float *mask, int64_t n_rays, int32_t samples_per_ray,
                                     const float *points, const float *centers,
                                     const int32_t *cluster_ranges, const float *cluster_radius, int32_t ncl,
                                     const float *group_centers, const int32_t *group_ranges, const float *group_radius, int32_t ngrp,
                                     const int32_t *h_coarse_rows, const int32_t *h_seed_from_coarser,
                                     int32_t nscale, const int32_t *query_rows, const int32_t *n_query_dev, int32_t *ray_start,
                                     const float *center, const int32_t *center_idx, int32_t *knn_idxs, void *stream);

/* Plain exact kNN for small problems (k <= 16): idx[nq,k] rows of s, ascending.
 * Used for the per-point k=3 search of network.py:265-269 and the k=10 visibility update
 * of network.py:508-512. */
int occnerf_knn_small(const float *q, int32_t nq, const float *s, int32_t ns, int32_t k,
                      int32_t *idx, void *stream);

/* unit[P,3] = normals / max(|normals|, 1e-8) in float64: the normal-side half of
 * F.cosine_similarity (network.py:275, occnerf_mlp.py:165).  A per-model constant, computed
 * once after generate_neural_points and passed to the two functions below. */
int occnerf_unit_normals(const double *normals, int32_t P, double *unit, void *stream);

/* Per-point signed-distance block, network.py:263-284 (chunk-invariant; once per frame).
 * point_cloud[P,3] = point_base + point_dist; normals[P,3] float64; kidx[P,3] from
 * occnerf_knn_small.  Outputs knn_base[P,3] float64, dist[P] fp32. */
int occnerf_point_sdf(const float *point_cloud, const float *point_base, const double *normals,
                      const double *unit_normals, const int32_t *kidx, int32_t P,
                      double *knn_base, float *dist, void *stream);

/* Per-point feature table, occnerf_mlp.py:171-175:
 * table[P,T], T = occnerf_point_table_stride() floats (64: 256-byte rows, so that the 128-byte encoding
 * part of a gathered row is exactly one cache line); columns
 *   [encode((knn_base+bound)/(2 bound), clamp((sdf+0.2)/0.8,0,1)) (32), learnable xyz (3), 0, unused...].
 * h_offsets: optional HOST copy of offsets[L+1]; with it the kernel knows per level whether the
 * table is dense or a power-of-two hash and skips the generic 32-bit modulo (same indices). */
/* Training step: gradient of the learnable point offsets point_dist[P] (network.py:109,119: point_cloud = point_base +
 * point_dist, one offset per point added to all three coordinates) through occnerf_point_sdf -- network.py:263-284 under
 * autograd -- from d_knn_base[P,3] (fp64) and d_dist[P]; kidx: the 3-NN ids of the forward (constants of the graph). */
int occnerf_point_sdf_backward(const float *point_cloud, const float *point_base, const double *normals,
                               const double *unit_normals, const int32_t *kidx, int32_t P, const double *d_knn_base,
                               const float *d_dist, float *d_point_dist, void *stream);
int32_t occnerf_point_table_stride(void);
int occnerf_point_table(const double *knn_base, const float *point_sdf, const float *learnable,
                        int32_t P, float bound, float two_bound, const float *embeddings,
                        const int32_t *offsets, const int32_t *h_offsets, uint32_t L, float S,
                        uint32_t H, float *table, void *stream);

/* Per-sample features: neighbour geometry + hash encoding + visibility-softmax
 * aggregation.  Replaces occnerf_mlp.py:144-181 (+ simple_agg :86-126).
 * knn_idxs[N,nscale,10]: base-point rows; counter[P] visibility counts; table from
 * occnerf_point_table.  Outputs mlp_in[N,68] = [agg(35), var(1), enc(32)] and raw[N,5]
 * column 4 (signed distance); enc_in[N,4] optional (NULL to skip).
 * Two optional inputs serve callers that hold the reference's *gathered* CanonicalMLP
 * arguments instead of per-point arrays: geo_idxs[N,10] (rows of point_base/normals to use
 * for the geometry prelude instead of knn_idxs[:,0,:]) and att_in[N,nscale*10] (visibility
 * counts already gathered, used instead of counter[knn_idxs]); NULL for the normal path.
 * rows (nullable, renderer's path only): a compact list of N sample indices into xyz / knn_idxs -- the
 * samples that can contribute to their pixel (motion-weight sum != 0); output row m then belongs to
 * sample rows[m].  n_dev (nullable, needs rows): the length of the list in device memory; N is then the
 * capacity of the outputs.
 * point_geo[P,16] / point_tail[P,4] (nullable, together; from occnerf_point_pack): the per-point inputs repacked so that
 * one gather brings everything the prelude / the softmax needs of a point; with them (and 4 scales, no gathered
 * inputs) the 8-lanes-per-sample kernel runs, without them the thread-per-sample one (same results bit for bit).
 * P: rows of point_geo / point_tail (ignored without them). */
int occnerf_point_pack(const float *point_base, const double *normals, const double *unit_normals,
                       const float *counter, const float *table, int32_t P, float *point_geo,
                       float *point_tail, void *stream);
int occnerf_sample_features(const float *xyz, int64_t N, const int32_t *knn_idxs, int32_t nscale,
                            const float *point_base, const double *normals,
                            const double *unit_normals, const float *counter,
                            const float *table, float bound, float two_bound,
                            const float *embeddings, const int32_t *offsets,
                            const int32_t *h_offsets, uint32_t L, float S, uint32_t H,
                            const int32_t *geo_idxs, const float *att_in, const int32_t *rows,
                            const int32_t *n_dev, const float *point_geo, const float *point_tail, int32_t P,
                            float *mlp_in, float *raw, float *enc_in, void *stream);
/* The renderer's form with the kNN centre cache (occnerf_knn_center): center[4] = (c, r^2) and center_agg[72] = of a sample that
 * has c's neighbour lists (this function on a sample at c) columns 0..35 of mlp_in, the encoder input x[4] (enc_in), columns
 * 36..67 of mlp_in.  A group of 8 consecutive listed samples that all lie inside the radius -- they all carry c's 40 ids --
 * copies columns 0..35 instead of gathering the 40 rows, and columns 36..67 as well when every one of its encoder inputs equals
 * the centre's bit for bit.  Same outputs, bit for bit.  Both pointers nullable together. */
int occnerf_sample_features_centered(const float *xyz, int64_t N, const int32_t *knn_idxs, int32_t nscale,
                            const float *point_base, const double *normals,
                            const double *unit_normals, const float *counter,
                            const float *table, float bound, float two_bound,
                            const float *embeddings, const int32_t *offsets,
                            const int32_t *h_offsets, uint32_t L, float S, uint32_t H,
                            const int32_t *geo_idxs, const float *att_in, const int32_t *rows,
                            const int32_t *n_dev, const float *point_geo, const float *point_tail, int32_t P,
                            const float *center, const float *center_agg,
                                     float *mlp_in, float *raw, float *enc_in, void *stream);

/* Differentiable neighbour aggregation of the training path (occnerf_mlp.py:86-126 simple_agg with the
 * gather of :176-178): agg[n,:] = sum_j atts[n,j] * feats[knn[n,j],:] for feats[P,F] (F <= 64), knn[N,K],
 * atts[N,K] (detached in the reference).  Backward: partial[W,P,F] with W = occnerf_agg_backward_slices(N);
 * every element is written, grad_feats = partial.sum(0).  Workgroups own (sample slice, point tile) pairs
 * and accumulate in LDS -- no global atomics; a first pass sums the gradient rows of runs of consecutive samples with
 * bitwise identical neighbour lists AND weights (K <= 64; arbitrary atts are exact: round 6 compares them too) into `scratch`
 * (occnerf_agg_backward_scratch_bytes(N, F) bytes) so that a run is scattered once.  Limits, refused by name: F <= 64, K <= 64 in
 * the backward, and at most 32 point tiles (P <= 32 x min(1024, 18432 / F) rows: 16 832 at F = 35).  Replaces torch's feats[knn]
 * materialisation and its index_put backward. */
int occnerf_agg_forward(const float *feats, int32_t F, const int32_t *knn, const float *atts, int64_t N, int32_t K,
                        float *agg, void *stream);
int32_t occnerf_agg_backward_slices(int64_t N);
int64_t occnerf_agg_backward_scratch_bytes(int64_t N, int32_t F);
int occnerf_agg_backward(const float *grad_agg, int32_t F, const int32_t *knn, const float *atts, int64_t N,
                         int32_t K, int32_t P, float *partial, void *scratch, int64_t scratch_bytes, void *stream);

/* Live samples of a frame (mask = per-sample motion-weight sum, network.py:330: alpha is multiplied by it).
 * rows[0 .. *count) = ascending indices of the samples with mask != 0, both in device memory; temp = device
 * scratch of occnerf_live_rows_temp_bytes(N) bytes.  occnerf_scatter_raw copies the compact raw_c[m, 0..4] rows
 * to raw_full[rows[m], 0..4] for m < *n_dev (raw_full zero-initialised by the caller). */
int64_t occnerf_live_rows_temp_bytes(int64_t N);
int occnerf_live_rows(const float *mask, int64_t N, int32_t *rows, int32_t *count, void *temp,
                      int64_t temp_bytes, void *stream);
int occnerf_scatter_raw(const float *raw_c, const int32_t *rows, const int32_t *n_dev, int64_t N_max,
                        float *raw_full, void *stream);

/* Repeated samples.  Every stage after the warp is a pure per-sample function (occnerf_mlp.py:144-199, mlp_offset.py:45-62),
 * and consecutive entries of a frame's sample list often carry bitwise identical inputs (where the motion-weight sum is far
 * below the clamp of network.py:324 the warped position collapses onto the origin).  occnerf_repeat_heads compares the key
 * of entry m -- key_dwords 32-bit words at keys + r * stride_dwords, r = rows ? rows[m] : m -- with entry m-1's as bit
 * patterns for m < *n_dev and writes scan[m] = number of entries 0..m that differ from their predecessor ("heads"; entry 0
 * is one), heads[scan[m]-1] = r for every head, *head_count, and, when head_mask is given (zero-filled by the caller),
 * head_mask[r] = 1.  scan has N_max entries (those beyond *n_dev repeat the total); temp = device scratch of
 * occnerf_repeat_heads_temp_bytes(N_max) bytes.  occnerf_scatter_raw_heads: raw_full[rows[m], 0..3] = raw_h[h, 0..3] and
 * raw_full[rows[m], 4] = raw_c[a, 4] with a = scanA ? scanA[m]-1 : m and h = scanB ? scanB[a]-1 : a.
 * occnerf_canonical_mlp_rows: occnerf_canonical_mlp_counted reading input row in_rows[n] for output row n. */
int64_t occnerf_repeat_heads_temp_bytes(int64_t N_max);
int occnerf_repeat_heads(const void *keys, int64_t stride_dwords, int32_t key_dwords, const int32_t *rows,
                         const int32_t *n_dev, int64_t N_max, int32_t *scan, int32_t *heads, int32_t *head_count,
                         float *head_mask, void *temp, int64_t temp_bytes, void *stream);
/* Distinct entries of a whole list (what run-length elimination leaves: equal keys that are not neighbours).  Entry j < *n_dev
 * is row r = heads ? heads[j] : j with the key of occnerf_repeat_heads; entries are grouped by an open-addressing table
 * (atomicCAS on a slot, full-key compare against the claimant), heads_out / *count_out list one representative row per
 * distinct key in ascending entry order, and scan (nullable, with its device-side length and capacity) -- a map whose
 * values are 1-based entry numbers, e.g. occnerf_repeat_heads' scan -- is rewritten to 1-based positions in heads_out.  Which
 * of several equal entries represents them is not deterministic; since their keys are equal bit for bit, results are.
 * temp = device scratch of occnerf_unique_heads_temp_bytes(cap) bytes; cap = capacity of the list. */
int64_t occnerf_unique_heads_temp_bytes(int64_t cap);
int occnerf_unique_heads(const void *keys, int64_t stride_dwords, int32_t key_dwords, const int32_t *heads,
                         const int32_t *n_dev, int64_t cap, int32_t *heads_out, int32_t *count_out, int32_t *scan,
                         const int32_t *n_scan_dev, int64_t scan_cap, void *temp, int64_t temp_bytes, void *stream);
int occnerf_scatter_raw_heads(const float *raw_h, const float *raw_c, const int32_t *rows, const int32_t *n_dev,
                              const int32_t *scanA, const int32_t *scanB, int64_t N_max, float *raw_full, void *stream);
int occnerf_canonical_mlp_rows(const float *mlp_in, const int32_t *in_rows, int64_t N_max, const int32_t *n_dev,
                               const float *packed, float *raw, void *stream);

/* Canonical MLP weights -> MFMA operand order.  h_W/h_b: HOST arrays of the 10 device
 * weight/bias pointers in module order: pts_linears.{0,2,4,6}, geo_linear.0,
 * rgb_linears.{0,2,4,6}, output_linear.0 (torch layout [out,in]).  packed: device buffer of
 * occnerf_canonical_mlp_packed_floats() floats.  Only mlp_depth = 4, mlp_width = 256
 * (configs/occnerf/zju_mocap/387/occnerf.yaml:16-21) is built. */
int64_t occnerf_canonical_mlp_packed_floats(void);
int occnerf_canonical_mlp_pack(const float *const *h_W, const float *const *h_b, float *packed,
                               void *stream);

/* Geometry + colour trunks on fp32 MFMA.  Replaces occnerf_mlp.py:183-199.
 * mlp_in[N,68] -> raw[N,5] columns 0..3 (rgb logits, sigma); column 4 is left alone.
 * occnerf_canonical_mlp: 16-sample waves (v_mfma_f32_16x16x4_f32), weights staged through LDS -- the
 * default.  occnerf_canonical_mlp_direct: 32-sample waves (32x32x2), weights straight from L2; same
 * packed buffer, same results to fp32 rounding (the dot products associate differently), kept as the
 * cross-check and for A/B timing. */
int occnerf_canonical_mlp(const float *mlp_in, int64_t N, const float *packed, float *raw,
                          void *stream);
int occnerf_canonical_mlp_direct(const float *mlp_in, int64_t N, const float *packed, float *raw,
                                 void *stream);
/* occnerf_canonical_mlp with the row count in DEVICE memory (n_dev, written by occnerf_live_rows on the same
 * stream): the launch covers N_max rows and the workgroups beyond *n_dev leave at once. */
int occnerf_canonical_mlp_counted(const float *mlp_in, int64_t N_max, const int32_t *n_dev,
                                  const float *packed, float *raw, void *stream);

/* The same two trunks on the bf16 matrix pipe with split operands ("bf16x3"): every fp32
 * weight and activation is carried as hi + lo bf16 (16 significand bits) and each product is
 * Wh*xh + Wh*xl + Wl*xh accumulated in fp32 -- 5.3x the fp32-MFMA rate.  Outputs differ from
 * the fp32 kernel by <= ~1e-5 on the raw logits (DESIGN.md section 3.1 has the measured
 * pixel-level effect, two orders inside the 1e-4 gate).  packed = the fp32 blob above (biases,
 * sigma and colour-head rows stay fp32); packed_bf16 = occnerf_canonical_mlp_packed_bf16_bytes()
 * bytes (zero-initialised by the caller: the tail is read-ahead padding) written by
 * occnerf_canonical_mlp_pack_bf16 from the same 10 weight pointers.
 * variant 0: weights staged through LDS by LDS-DMA (the fast one); 1: every wave loads its
 * own operands from L2 (kept for A/B measurements). */
int64_t occnerf_canonical_mlp_packed_bf16_bytes(void);
int occnerf_canonical_mlp_pack_bf16(const float *const *h_W, void *packed_bf16, void *stream);
int occnerf_canonical_mlp_bf16x3(const float *mlp_in, int64_t N, const float *packed,
                                 const void *packed_bf16, float *raw, int32_t variant,
                                 void *stream);
/* The renderer's call (as occnerf_canonical_mlp_rows / _counted for the fp32 kernel): the entry count is read from
 * device memory (*n_dev <= N_max; the launch is sized for N_max), entry n's input row is mlp_in[in_rows[n]]
 * (in_rows nullable: row n), its result goes to raw[n]. */
int occnerf_canonical_mlp_bf16x3_rows(const float *mlp_in, const int32_t *in_rows, int64_t N_max,
                                      const int32_t *n_dev, const float *packed, const void *packed_bf16,
                                      float *raw, int32_t variant, void *stream);

/* Opt-in fp32-GRADE split (cfg.mlp_precision = 'f16x3'): occnerf_mlp.py:183-199 / mlp_offset.py:45-62 on
 * v_mfma_f32_32x32x16_f16 with every operand cut into two fp16 pieces kept in the normal range -- 22 significand bits per
 * operand (fp32: 24), three products, fp32 accumulation (csrc/split.h F16x3: activations travel scaled by 16, the weights'
 * low piece scaled by 2^11 against xh 2^-11; subnormal inputs are preserved by the instruction, measured with
 * tools/mfma_f16_probe.hip).  Domain: hidden activations below 65504 / 16 = 4 094 (larger ones saturate there) -- and leaving
 * it is REPORTED: domain_flag (nullable; a zero-initialised device word the caller owns) has bit 0 set by every wave in which a
 * value reached the clamp, so the caller can re-evaluate with the fp32 kernels (Network.forward does).
 * packed_f16: the same byte count and layout as the bf16 stream (occnerf_canonical_mlp_packed_bf16_bytes /
 * occnerf_nonrigid_packed_bf16_bytes, zero-initialised), written by the _pack_f16 call; packed: the fp32 blob (biases, head
 * rows).  in_rows / rows and n_dev nullable (every row, N_max entries); with a list the count is read from device memory.
 * The non-rigid call may run in place (xyz_out == xyz_in); with rows the offsets go to the listed samples. */
int occnerf_canonical_mlp_pack_f16(const float *const *h_W, void *packed_f16, void *stream);
int occnerf_canonical_mlp_f16x3(const float *mlp_in, const int32_t *in_rows, int64_t N_max, const int32_t *n_dev,
                                const float *packed, const void *packed_f16, float *raw, uint32_t *domain_flag,
                                void *stream);
int occnerf_nonrigid_pack_f16(const float *const *h_W, void *packed_f16, void *stream);
int occnerf_nonrigid_f16x3(const float *xyz_in, int64_t N_max, const int32_t *rows, const int32_t *n_dev,
                           const float *cond, const float *h_hann, const float *W0, const float *b0, float *packed,
                           const void *packed_f16, float *xyz_out, uint32_t *domain_flag, void *stream);

/* Alpha compositing, network.py:320-348.  raw[n,S,5], mask[n*S], z_vals[n,S],
 * rays[n,8] (direction at floats 3..5), h_bgcolor[3] host, 0..255.
 * Outputs rgb[n,3], acc[n], depth[n]; weights[n,S] and term[n] (argmax alpha) optional.
 * out_rows[n] (optional): ray r's rgb / acc / depth / term go to row out_rows[r] (the renderer walks the rays in
 * Morton order and hands the results back in the caller's order without a separate gather). */
int occnerf_composite(const float *raw, const float *mask, const float *z_vals, const float *rays,
                      const float *h_bgcolor, int64_t n, int32_t S, float *rgb, float *acc,
                      float *depth, float *weights, int32_t *term, const int64_t *out_rows, void *stream);

/* ------------------------------------------------------------------------------------
 * 3. Training step (BASELINE configs[4]).  The reference
 *    trains through torch autograd over nn.Linear / F.grid_sample / cumprod (trainer.py:239-249,
 *    occnerf_mlp.py:183-199, network.py:320-402); these are the hand-written forward-with-saved-
 *    activations and backward kernels the autograd Functions of occnerf_amd/train_ops.py call.
 * ---------------------------------------------------------------------------------- */

/* Linear layers on the matrix pipe, element type bf16 (bf16 != 0: v_mfma_f32_32x32x16_bf16, fp32
 * accumulate) or fp32 (v_mfma_f32_32x32x2_f32, exact).  All matrices are row-major with widths padded to
 * multiples of 32 elements (zero padding is exact: padded weights and activations are zeros).
 *
 * occnerf_linear_pack: W[out_dim,in_dim] fp32 (torch layout), b[out_dim] or NULL ->
 *   Wp[n_pad,k_pad] with Wp[n][k] = W[row_map[n]][col_map[k]] (0 where a map entry is -1), Wt[k_pad,n_pad] its
 *   transpose (either may be NULL), bias_p[n_pad] fp32 (may be NULL).  row_map/col_map: int32 device arrays.
 * occnerf_linear_forward: y[m,n] = epi(sum_k x[m,k] W[n,k]), x = [x0 | x1] two column segments (k1 = 0: one),
 *   ld* row pitches in ELEMENTS; epi = (+bias[n]) (ReLU) (zero where mask[m,n] <= 0), each optional;
 *   stored as the element type, or fp32 when out_f32; only columns < n_store are stored;
 *   aux (optional): column aux_col is also written, in fp32, to aux[m * aux_stride].
 *   Forward of a layer: W = Wp, bias, relu.  Input gradient of a layer: x = dZ, W = Wt, mask = the layer's
 *   saved input (its ReLU mask).
 * occnerf_linear_wgrad: part[G,n_pad,k_pad] / dbpart[G,n_pad] = per-workgroup partial sums over row slices of
 *   dZ^T X and of the column sums of dZ, G = occnerf_linear_wgrad_slices(M); occnerf_linear_wgrad_reduce adds
 *   the G slices into dW[out_dim,in_dim] (and db) through the same row/col maps (accumulate != 0: += ). */
int occnerf_linear_pack(const float *W, const float *b, int32_t out_dim, int32_t in_dim, const int32_t *row_map,
                        int32_t n_pad, const int32_t *col_map, int32_t k_pad, int32_t bf16, void *Wp, void *Wt,
                        float *bias_p, void *stream);
int occnerf_linear_forward(const void *x0, int64_t ld0, int32_t k0, const void *x1, int64_t ld1, int32_t k1,
                           const void *W, const float *bias, int32_t relu, const void *mask, int64_t ldm, void *y,
                           int64_t ldy, int32_t out_f32, int32_t n_store, float *aux, int32_t aux_col,
                           int64_t aux_stride, int64_t M, int32_t n_pad, int32_t bf16, void *stream);
int32_t occnerf_linear_wgrad_slices(int64_t M);
int occnerf_linear_wgrad(const void *dz, int64_t lddz, int32_t n_pad, const void *x, int64_t ldx, int32_t k_pad,
                         int64_t M, int32_t bf16, float *part, float *dbpart, void *stream);
int occnerf_linear_wgrad_reduce(const float *part, const float *dbpart, int32_t G, int32_t n_pad, int32_t k_pad,
                                const int32_t *row_map, const int32_t *col_map, float *dW, int32_t in_dim, float *db,
                                int32_t accumulate, void *stream);

/* The training step's forward of BOTH trunks in one launch (csrc/trunks.hip; occnerf_mlp.py:183-199, bf16 arithmetic with fp32
 * accumulation -- BASELINE configs[4]): a sample's activations stay in registers through the ten layers (the renderer's
 * transposed-MFMA scheme) and the tensors the backward reads are written on the way, row-major, exactly as the layer-by-layer
 * forward (occnerf_linear_forward) produced them: X0[M,96] = [agg 35 | var | enc 32 | 0], A1..A4[M,256], GEO[M,96] (geometry
 * features in columns 0..63, sigma in column 64), B1..B4[M,256], all bf16, and raw4[M,4] fp32 = (colour logits, sigma).
 * h_W (host array of device pointers): the nine MFMA layers pts_linears.{0,2,4,6}, geo_linear.0, rgb_linears.{0,2,4,6} in torch
 * layout; packed: occnerf_trunks_packed_bytes() bytes, zero-initialised once by the caller.  packed_f32: the blob of
 * occnerf_canonical_mlp_pack (biases, sigma row, colour rows).  h_A / h_B: host arrays of the four activation buffers. */
int64_t occnerf_trunks_packed_bytes(void);
int occnerf_trunks_pack_bf16(const float *const *h_W, void *packed, void *stream);
int occnerf_trunks_forward_bf16(const float *agg, const float *var, const float *enc, int64_t M, const float *packed_f32,
                                const void *packed_bf16, void *X0, void *const *h_A, void *GEO, void *const *h_B, float *raw4,
                                void *stream);

/* Backward of occnerf_composite (network.py:320-348 under autograd): g_rgb[n,3], g_acc[n], g_depth[n] (each may
 * be NULL = zero) -> d_raw[n*S,5] (column 4 = 0) and d_mask[n*S] (optional).  S <= 256. */
int occnerf_composite_backward(const float *raw, const float *mask, const float *z_vals, const float *rays,
                               const float *h_bgcolor, int64_t n, int32_t S, const float *g_rgb, const float *g_acc,
                               const float *g_depth, float *d_raw, float *d_mask, void *stream);

/* Backward of occnerf_sample_warp's mask output (network.py:351-402 under autograd; x_skel carries no gradient in
 * the reference's graph: it only enters CanonicalMLP through no_grad quantities).  g_mask[n*S] ->
 * d_vol_part[W,nb,G,G,G] and d_rt_part[W,nb,12] (rows: dRs[3,3] then dTs[3]) with
 * W = occnerf_warp_backward_slices(n*S); the caller sums over W.  G must be 32. */
int32_t occnerf_warp_backward_slices(int64_t n_samples);
int occnerf_warp_backward(const float *rays, int64_t n, int32_t S, const float *z_vals, const float *g_mask,
                          const float *Rs, const float *Ts, const float *vol, int32_t nb, int32_t G,
                          const float *h_bbox_min, const float *h_bbox_scale, float *d_vol_part, float *d_rt_part,
                          void *stream);

/* Attention weights of simple_agg (occnerf_mlp.py:110-125): counter[P], knn[N,K] (K <= 40) -> atts[N,K]
 * (softmax) and var[N] (unbiased variance of the normalised counts). */
int occnerf_agg_weights(const float *counter, const int32_t *knn, int64_t N, int32_t K, float *atts, float *var,
                        void *stream);

/* Training step: gradients of the pose refiner's five Linear layers through the chain occnerf_pose_motion_bases evaluates
 * (mlp_delta_body_pose.py:35-41, network_util.py:98-124 Rodrigues, network.py:535-539, network_util.py:166-200 forward
 * kinematics + inverse), given the cotangents dRs[24,3,3], dTs[24,3] of its outputs -- what torch autograd derives in ~460
 * tiny launches per step (trainer.py:239-249).  One workgroup; the forward is recomputed inside.  h_dW[l] / h_db[l]: device
 * buffers shaped like layer l's weight / bias, overwritten.  Only meaningful with refinement on. */
int occnerf_pose_motion_bases_backward(const float *const *h_W, const float *const *h_b, const float *posevec,
                                       const float *dst_Rs, const float *dst_Ts, const float *cnl_gtfms,
                                       const float *dRs, const float *dTs, float *const *h_dW, float *const *h_db,
                                       void *stream);

/* Per-frame preamble.
 * occnerf_pose_motion_bases: pose refiner + motion bases in one launch.  h_W/h_b: HOST arrays of the 5 device weight /
 *   bias pointers of BodyPoseRefiner.block_mlps.{0,2,4,6,8} (69 -> 256 x4 -> 69; mlp_delta_body_pose.py:35-41); refine = 0
 *   skips the refiner (iter_val < pose_decoder.kick_in_iter, network.py:558).  posevec[69], dst_Rs[24,3,3], dst_Ts[24,3],
 *   cnl_gtfms[24,4,4] -> Rs[24,3,3], Ts[24,3] (network_util.py:98-124,166-200; network.py:535-539).
 * occnerf_prior_softmax: vol[C,V] = softmax over C of decoded[C,V] + log(prior[C,V]) (deconv_vol_decoder.py:31-33).
 * occnerf_pack_rays: rays[2,R,3], near[R], far[R] -> rays8[R,8] = (o, d, near, far) of ray order[r] (order NULL: identity). */
int occnerf_pose_motion_bases(const float *const *h_W, const float *const *h_b, const float *posevec, int32_t refine,
                              const float *dst_Rs, const float *dst_Ts, const float *cnl_gtfms, float *Rs, float *Ts,
                              void *stream);
int occnerf_prior_softmax(const float *decoded, const float *prior, int32_t C, int64_t V, float *vol, void *stream);
int occnerf_pack_rays(const float *rays, const float *near, const float *far, const int64_t *order, int64_t R, float *rays8,
                      void *stream);

/* Image assembly after the renderer, run.py:46-63 (unpack_to_image) + image_util.py:19-20 (to_8b_image) in one kernel:
 * out_rgb[H*W*3] = uint8(255 * clip(x, 0, 1)) of the ray's colour where a ray exists, of h_bgcolor01[3] (HOST,
 * cfg.bgcolor / 255 as float32) elsewhere; out_alpha[H*W*3] (optional) the same for alpha, replicated to 3 channels,
 * 0 where no ray.  ray_index[R]: ascending flat pixel index of every ray (nonzero(ray_mask)), int64. */
int occnerf_assemble_image(const float *rgb, const float *alpha, const int64_t *ray_index, int64_t R, int32_t height,
                           int32_t width, const float *h_bgcolor01, uint8_t *out_rgb, uint8_t *out_alpha, void *stream);

/* ConvTranspose3d(kernel 4, stride 2, padding 1) of the motion-weight volume decoder (network_util.py:12-50) as GEMM +
 * gather.  cols[Cout*64, D*H*W] = W.view(Cin, Cout*64)^T x[Cin, D*H*W] (the caller's GEMM) ->
 * out[Cout, 2D, 2H, 2W] = bias[co] + the <= 8 taps that reach each output voxel (occnerf_convt3d_col2im);
 * its adjoint dcols[Cout*64, D*H*W] from gy[Cout, 2D, 2H, 2W] (occnerf_convt3d_im2col), after which
 * dx = W.view(Cin, Cout*64) dcols and dW = x dcols^T are plain GEMMs again.  No atomics, fixed summation order. */
int occnerf_convt3d_col2im(const float *cols, const float *bias, int32_t Cout, int32_t D, int32_t H, int32_t W,
                           float *out, void *stream);
int occnerf_convt3d_im2col(const float *gy, int32_t Cout, int32_t D, int32_t H, int32_t W, float *dcols, void *stream);

/* Optimiser step on the device: the reference's clip_grad_norm_(parameters, max_norm) + torch.optim.Adam.step()
 * (trainer.py:248-249, optimizer.py:12-43) as one multi-tensor pass.  table: n_tensors rows of
 * occnerf_adam_table_row_bytes() = 56 bytes in DEVICE memory, each {float *param; const float *grad; float *exp_avg;
 * float *exp_avg_sq; int64 numel; float lr; float bias_corr1 = 1 - beta1^t; float bias_corr2_sqrt = sqrt(1 - beta2^t);
 * float pad} with t the step count of THAT tensor (torch.optim.Adam keeps one per parameter); chunks[n_chunks,2] int32 =
 * (tensor index, chunk index), a chunk being chunk_elems consecutive elements of its tensor; max_grad_norm <= 0: no clipping.  scratch: n_chunks + 1 floats; scratch[0] receives the squared
 * global gradient norm.  Nothing visits the host. */
int32_t occnerf_adam_table_row_bytes(void);
int occnerf_adam_step(const void *table, int32_t n_tensors, const int32_t *chunks, int32_t n_chunks,
                      int32_t chunk_elems, double beta1, double beta2, double eps, double max_grad_norm, float *scratch,
                      void *stream);

#ifdef __cplusplus
}
#endif
#endif /* OCCNERF_HIP_H */
