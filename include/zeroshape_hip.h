/* zeroshape_hip.h - C ABI of libzeroshape_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for ZeroShape's dense SDF-query / Chamfer hot path.  Plain
 * pointers and sizes only: every pointer is a DEVICE pointer unless marked
 * [host]; `stream` is a hipStream_t passed as void* (NULL = default stream).
 * All entry points are asynchronous w.r.t. the host (they enqueue on `stream`
 * and return) and return 1 on success, 0 on failure - the convention of the
 * reference's native launchers (external/chamfer3D/chamfer3D.cu:145-151).
 * After a 0, zs_last_error() gives a thread-local message.  Nothing here
 * allocates device memory; callers own every buffer.
 *
 * Reference interfaces replaced (paths relative to the upstream ZeroShape tree):
 *   zs_chamfer_forward   <- chamfer_cuda_forward,  external/chamfer3D/chamfer3D.cu:136-154
 *                           (bound as chamfer_3D.forward, chamfer_cuda.cpp:19-21,31)
 *   zs_chamfer_backward  <- chamfer_cuda_backward, external/chamfer3D/chamfer3D.cu:176-195
 *                           (bound as chamfer_3D.backward, chamfer_cuda.cpp:24-28,32)
 *   zs_sdf_*             <- Implicit.forward, model/shape/implicit.py:251-288, as it is
 *                           driven by compute_level_grid, utils/eval_3D.py:22-45, and
 *                           get_dense_3D_grid, utils/eval_3D.py:11-20
 *   zs_bf_lower_bounds   <- pruning for brute_force_search, utils/eval_3D.py:140-170
 *   zs_pose_search_batch <- one batch of brute_force_search, utils/eval_3D.py:149-168
 *   zs_normalize_pc      <- normalize_pc, utils/eval_3D.py:93-102
 *   zs_standardize_pc, zs_icp_step <- standardize_pc, ICP, utils/eval_3D.py:83-91, 271-284
 *   zs_fscore            <- compute_fscore, utils/eval_3D.py:215-231
 *   zs_mc_*, zs_mesh_*   <- convert_to_explicit, utils/eval_3D.py:233-263 (PyMCubes
 *                           marching_cubes + trimesh.sample on the host)
 *   zs_seen_surface, zs_unproj_depth, zs_valid_norm_fac, zs_masked_resample,
 *   zs_intr_param2mtx    <- the seen-surface geometry of Graph.forward,
 *                           model/compute_graph/graph_shape.py:89-113,131-144, with
 *                           utils/camera.py:52-108 and utils/util.py:323-345
 *   zs_depth_metrics     <- DepthMetric.compute_metrics, utils/eval_depth.py:46-116
 *   zs_conv2d_nhwc, zs_group_norm_nhwc, zs_layer_norm, zs_attention, zs_max_pool_nhwc,
 *   zs_global_mean_nhwc, zs_upsample2x_nhwc, zs_assemble_tokens, zs_readout_concat,
 *   zs_nchw_to_nhwc, zs_nhwc_to_nchw
 *                        <- the layers of the image encoders: DPTDepthModel
 *                           (model/depth/dpt_depth.py:68-122, blocks.py, vit.py), CoordEncRes
 *                           (model/shape/seen_coord_enc.py:141-194), the intrinsics head
 *                           (model/compute_graph/graph_shape.py:18-28,125-129)
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 */
#ifndef ZEROSHAPE_HIP_H
#define ZEROSHAPE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZS_ABI_VERSION 38

/* ABI version of the loaded library (== ZS_ABI_VERSION it was built with). */
int zs_abi_version(void);

/* Thread-local description of the last failure ("" if none). [host] */
const char *zs_last_error(void);

/* ------------------------------------------------------------------------- *
 * Chamfer-3D nearest neighbour (external/chamfer3D/chamfer3D.cu)
 * ------------------------------------------------------------------------- */

/* For every point of xyz1[b][n][3] the SQUARED distance to, and index of, its
 * nearest point in xyz2[b][m][3], and vice versa.  fp32, contiguous.
 *   dist1[b][n], idx1[b][n]  : nearest in xyz2 for each xyz1 point
 *   dist2[b][m], idx2[b][m]  : nearest in xyz1 for each xyz2 point
 * d = fma(dz,dz, fma(dy,dy, dx*dx)) with (dx,dy,dz) = other - self (bit-exact with
 * the reference kernel under nvcc's default FMA contraction); ties resolve to the
 * LOWEST index (chamfer3D.cu:36,126).  If the other cloud is empty the outputs of
 * that direction are left untouched (the reference kernel writes nothing either;
 * its caller pre-zeroes them, dist_chamfer_3D.py:29-38). */
int zs_chamfer_forward(const float *xyz1, const float *xyz2, int b, int n, int m,
                       float *dist1, float *dist2, int *idx1, int *idx2, void *stream);

/* Same contract and bit-identical results, but spatially accelerated: the candidate clouds
 * are binned into uniform grids in `workspace` (zs_chamfer_workspace_bytes(b, n, m) bytes,
 * 16-byte aligned, contents scratch) and each query only evaluates the candidates whose
 * cells can still beat its running minimum.  Worth it from a few thousand points per cloud. */
size_t zs_chamfer_workspace_bytes(int b, int n, int m);
int zs_chamfer_forward_ws(const float *xyz1, const float *xyz2, int b, int n, int m,
                          float *dist1, float *dist2, int *idx1, int *idx2,
                          void *workspace, size_t workspace_bytes, void *stream);

/* Gradient scatter (chamfer3D.cu:155-195): gradxyz1[b][n][3] / gradxyz2[b][m][3] are
 * ACCUMULATED into with fp32 atomics (caller zeroes them, dist_chamfer_3D.py:51-55):
 *   g = 2*graddist1[i][j];  gradxyz1[i][j] += g*(p1 - p2[idx1]);  gradxyz2[i][idx1] -= same
 * and symmetrically for direction 2. */
int zs_chamfer_backward(const float *xyz1, const float *xyz2, int b, int n, int m,
                        float *gradxyz1, float *gradxyz2,
                        const float *graddist1, const float *graddist2,
                        const int *idx1, const int *idx2, void *stream);

/* ------------------------------------------------------------------------- *
 * Implicit occupancy decoder (model/shape/implicit.py), default architecture of
 * options/shape.yaml:19-44: C=256, 8 heads x 32, 2 attention blocks (mlp 4x),
 * 197 latent tokens, 8-layer softplus(beta=100) MLP with skips at 2,4,6.
 *
 * Data flow:
 *   zs_sdf_program_bytes()                  size of one packed "decoder program"
 *   [host] pack the state_dict             zeroshape_amd/program.py -> float32 words
 *   zs_sdf_prologue(program, latent, ...)  per image: hoisted latent path
 *                                          (latent_proj + pos_embed, block-0 latent
 *                                          self-attention + MLP, K/V of both blocks)
 *                                          written into the program's K/V records
 *   zs_sdf_query_points / _grid            per query point: the fused decoder; `workspace`
 *                                          (zs_sdf_workspace_bytes(), 16-byte aligned) may be
 *                                          shared by launches on the same stream
 * ------------------------------------------------------------------------- */

/* Bytes of one per-image decoder program (weights in MFMA operand order + K/V). */
size_t zs_sdf_program_bytes(void);
/* Bytes of the per-image scratch the prologue needs. */
size_t zs_sdf_prologue_scratch_bytes(void);
/* Bytes of the workspace the query kernels need (independent of batch and point count:
 * one 96 KiB slab per resident wave, 96 MiB in all; contents are scratch, the caller need not
 * initialise it: since ABI 33 the library zeroes the split-fp16 kernels' tile counter in its last
 * 4 KiB (dynamic tile order) on the stream in front of every launch - a 4-byte memset
 * node under stream capture).  One workspace per stream that launches concurrently.
 * ZS_SPLIT_STATIC_TILES=1 in the environment (read per launch) selects the static tile deal. */
size_t zs_sdf_workspace_bytes(void);
/* Extra workspace bytes zs_sdf_query_points needs BEHIND the fixed part when `attn` is
 * requested (raw probability tiles: ~14.5 KiB per point). */
size_t zs_sdf_attn_scratch_bytes(int batch, int m);

/* Fill the per-image K/V records of `programs[i]` (i < batch; programs are
 * program_stride_bytes apart, each initialised by copying the packed weights)
 * from latent_depth[batch][197][256] fp32.  `lat_params` is the packed latent-path
 * parameter block (zeroshape_amd/program.py: pack_latent_params). */
int zs_sdf_prologue(void *programs, size_t program_stride_bytes, const float *lat_params,
                    const float *latent_depth, int batch, void *scratch, void *stream);
/* The same with options (ABI 33).  ZS_SDF_POS_PERLAYER: Implicit(pos_perlayer=True) - the reference class's own default,
 * model/shape/implicit.py:269-272 - adds pos_embed to the latent rows in front of EVERY attention block, not only the first:
 * the K/V records of block 1 come from LN1'(x2 + pos_embed).  zs_sdf_prologue == flags 0 (options/shape.yaml:44). */
#define ZS_SDF_POS_PERLAYER 1
int zs_sdf_prologue_ex(void *programs, size_t program_stride_bytes, const float *lat_params,
                       const float *latent_depth, int batch, void *scratch, int flags, void *stream);

/* The comparison behind the choice of arithmetic (zeroshape_amd/model/shape/implicit.py: Implicit.prepare): got / want
 * [batch][m] logits of the same probe points from the split-fp16 and the exact-fp32 kernel -> stats[batch][5] = max |got - want|,
 * mean |got - want|, max |want|, max |sigmoid(got) - sigmoid(want)|, number of sign disagreements at |want| >= flip_band (any
 * non-finite input: the first four NaN); flags (optional) [batch][2] int32 = 1 when the image FAILS the raw-logit rule
 * (max |d| <= tol) / the occupancy rule (max |d occ| <= tol_occ and no such disagreement).  One launch, fixed reduction order. */
int zs_sdf_verdict_stats(const float *got, const float *want, int batch, int m, float flip_band, float tol, float tol_occ,
                         float *stats, int *flags, void *stream);

/* logits[batch][m] = Implicit(latent, None, points[batch][m][3]) (pre-sigmoid).
 * attn (optional, may be NULL): [batch][m][197] = mean over heads and blocks of the
 * point->latent attention probabilities (implicit.py:63,79,277; the self column is
 * excluded after the softmax over 198, so rows sum to < 1).  When given, `workspace` must
 * hold zs_sdf_workspace_bytes() + zs_sdf_attn_scratch_bytes(batch, m) bytes.
 * tile_mask (optional, NULL = every tile; here and in the two grid entry points below):
 * int32[batch * ceil(m / 128)], one entry per 128-point tile of each image in point order; a zero
 * entry leaves that tile's outputs untouched.  It is how the exact-fp32 kernel re-evaluates the
 * tiles the split-fp16 kernel flagged (see below); not combinable with attn. */
int zs_sdf_query_points(const void *programs, size_t program_stride_bytes, int batch,
                        const float *points, int m, float *logits, float *attn,
                        const int *tile_mask, void *workspace, void *stream);

/* Dense-grid query without materialising the points tensor.  Evaluates x-slices
 * [slice_begin, slice_end) of the G^3 grid (G = vox_res+1 samples per axis; coordinate
 * i -> axis[i], `axis` being torch.linspace(range_min, range_max, G) supplied by the
 * caller so values are bit-identical to utils/eval_3D.py:16).  Layout of out:
 * [batch][slice_end-slice_begin][G][G], x slowest / z fastest, like occ in
 * utils/eval_3D.py:45-46.  apply_sigmoid!=0 stores sigmoid(logit) (compute_level_grid's
 * return), else the logit. */
int zs_sdf_query_grid(const void *programs, size_t program_stride_bytes, int batch,
                      const float *axis, int G, int slice_begin, int slice_end,
                      int apply_sigmoid, float *out, const int *tile_mask, void *workspace,
                      void *stream);

/* Split-fp16 ("f16x3") decoder: the same network with every contraction on the 16-bit matrix
 * pipe, both operands split into two fp16 halves (A B ~= Ah Bh + Ah Bl + Al Bh, fp32
 * accumulation; halves rounded to nearest even: <= 2^-22 relative operand error without a sign
 * preference for |x| < 65520, inf / nan beyond): max |logit difference| to the exact-fp32 kernels
 * ~3e-6 on the seeded network, 1.5e-5 on a trained one with logits up to 17 - as far from the fp32
 * CPU oracle as the fp32 kernels are (contract 1e-4); 2.8x their throughput.
 *   zs_sdf_split_programs     fp32 programs (after zs_sdf_prologue) -> split programs of the
 *                             same size and stride rules (split_programs must not alias programs)
 *   zs_sdf_query_points_split / zs_sdf_query_grid_split
 *                             as zs_sdf_query_points (without the attention map) / _grid, on
 *                             split programs; same workspace
 * Envelope guard.  The operand error acts on the attention logits in absolute terms,
 * |dS| <= 2.5 * 2^-21 * d^-1/2 |q| |k|, and operands beyond +-65504 saturate.  tile_flags (optional,
 * NULL = no guard): int32[batch * ceil(m / 128)] ZEROED by the caller; the kernel sets the entry
 * of every 128-point tile in which some point and head reach d^-1/2 |q| max_l |k_l| > 64, or whose
 * program holds an operand that is not finite or outside the fp16 range (those tiles' outputs
 * are NaN).  Passing the same array as tile_mask to the exact-fp32 entry point on the same
 * stream re-evaluates exactly those tiles - no host round trip (Implicit.query_* does this). */
int zs_sdf_split_programs(const void *programs, size_t program_stride_bytes, void *split_programs,
                          size_t split_stride_bytes, int batch, void *stream);
int zs_sdf_query_points_split(const void *split_programs, size_t program_stride_bytes, int batch,
                              const float *points, int m, float *logits, int *tile_flags,
                              void *workspace, void *stream);
int zs_sdf_query_grid_split(const void *split_programs, size_t program_stride_bytes, int batch,
                            const float *axis, int G, int slice_begin, int slice_end,
                            int apply_sigmoid, float *out, int *tile_flags, void *workspace,
                            void *stream);

/* As zs_sdf_query_grid / zs_sdf_query_grid_split, for the points [point_begin, point_end) of
 * the grid in its memory order (x slowest, z fastest): out[batch][point_end - point_begin].
 * This is the unit of the multi-GPU sharding (zeroshape_amd/parallel.py): equal point ranges
 * instead of whole x-slices (129 slices over 8 ranks would leave 17-17-...-10). */
int zs_sdf_query_grid_range(const void *programs, size_t program_stride_bytes, int batch,
                            const float *axis, int G, long long point_begin, long long point_end,
                            int apply_sigmoid, float *out, const int *tile_mask, void *workspace,
                            void *stream);
int zs_sdf_query_grid_range_split(const void *split_programs, size_t program_stride_bytes, int batch,
                                  const float *axis, int G, long long point_begin,
                                  long long point_end, int apply_sigmoid, float *out,
                                  int *tile_flags, void *workspace, void *stream);

/* ------------------------------------------------------------------------- *
 * Brute-force pose search support (brute_force_search, utils/eval_3D.py:140-170).
 * lower_bounds[i] <= Chamfer-L1 of rotation i (normalize_pc(R_i pred) vs gt_normalized),
 * up to rounding: rigorous cell-distance bounds on 32^3 occupancy grids of the two clouds.
 * The host evaluates rotations exactly in order of increasing bound and stops when the
 * smallest remaining bound (with a margin) exceeds the best exact distance - same winner
 * as the exhaustive scan.  pred [n][3] raw, gt_normalized [m][3], rotations [k][3][3];
 * grid_gt / grid_pred: zs_bf_grid_bytes() each, scratch: zs_bf_scratch_bytes().
 * ------------------------------------------------------------------------- */
size_t zs_bf_grid_bytes(void);
size_t zs_bf_scratch_bytes(void);
int zs_bf_lower_bounds(const float *pred, int n, const float *gt_normalized, int m,
                       const float *rotations, int k, float *grid_gt, float *grid_pred,
                       void *scratch, float *lower_bounds, void *stream);

/* ------------------------------------------------------------------------- *
 * Fused pose search (brute_force_search, utils/eval_3D.py:140-170) and its helpers
 * normalize_pc (:93-102) and compute_fscore (:215-231).
 *
 * zs_pose_search_batch evaluates `count` (<= zs_pose_max_batch()) rotations exactly - rotate
 * pred[n][3], normalize_pc, nearest neighbours both ways against gt_normalized[m][3] with the
 * arithmetic of zs_chamfer_forward, sqrt, means, the six F-score thresholds - and merges the
 * lexicographic (Chamfer-L1, rotation index) minimum into the running record `best`
 * (zs_pose_best_bytes(), device; zs_pose_best_init first): float cd, int32 index, float acc,
 * comp, f[6], int32 rotations evaluated, int32 rotations scanned in full, float external bound (slot 12: +inf,
 * or the best distance other ranks have found - it tightens every pruning decision of later batches and never
 * enters the record; zeroshape_amd/utils/eval_3D.py shares it once per search).  That minimum over all batches is the reference's first
 * strict minimum (:161-168).  Rotation b of the batch is rotations[order ? order[b] : b]
 * ([..][3][3]); its reported index is that number + index_offset.  lower_bound (optional,
 * device): one float, the smallest zs_bf_lower_bounds value in this batch - when it proves the
 * batch cannot beat `best`, the kernels return at once, so all batches can be enqueued without
 * a host synchronisation.  No rotated cloud, distance or index array is materialised; results
 * are bit-reproducible (fixed reduction order) whatever batch a rotation is evaluated in.
 * scratch: zs_pose_scratch_bytes(n, m, count).
 * zs_pose_apply: out[n][3] = normalize_pc(rotations[index[0]] pred) (index: device int32;
 * scratch: 64 bytes).  zs_normalize_pc: out[b][n][3] = normalize_pc(pc[b][n][3]) (scratch: 64 b
 * bytes).  zs_fscore: out[b][n_thresholds] from UN-squared distances dist1[b][n], dist2[b][m].
 * ------------------------------------------------------------------------- */
int zs_pose_max_batch(void);
size_t zs_pose_scratch_bytes(int n, int m, int count);
size_t zs_pose_best_bytes(void);
int zs_pose_best_init(float *best, void *stream);
int zs_pose_search_batch(const float *pred, int n, const float *gt_normalized, int m,
                         const float *rotations, const int *order, int count, int index_offset,
                         const float *lower_bound, const float *thresholds6, float *best,
                         void *scratch, void *stream);
/* The same batch through uniform grids (csrc/zs_point_grid.h): the ground truth binned once per search
 * (zs_pose_gt_grid into the first part of `grids`, zs_pose_grid_bytes(n, m, max batch) bytes, 16-byte aligned), the
 * rotated + normalised prediction once per rotation; ~10^2 instead of 10^4 distance evaluations per query and
 * bit-identical records (a skipped candidate provably has a strictly larger distance; sums formed in the same order). */
size_t zs_pose_grid_bytes(int n, int m, int count);
int zs_pose_gt_grid(const float *gt_normalized, int m, void *grids, void *stream);
/* pred_stats (may be NULL = pred): the prediction in the order its mean is summed in (see zs_pose_search_batch_sorted). */
int zs_pose_search_batch_grid(const float *pred, int n, const float *gt_normalized, int m, const float *rotations,
                              const int *order, int count, int index_offset, const float *lower_bound,
                              const float *thresholds6, float *best, void *scratch, void *grids, const float *pred_stats,
                              void *stream);
/* Box-culled scan (round 3, the default of brute_force_search).  zs_morton_sort orders a cloud along the Z-order curve
 * of its bounding box (30-bit keys, stable radix sort: the permutation is a pure function of the coordinates); perm
 * (may be NULL) receives sorted[i] = points[perm[i]], tile_boxes (may be NULL) the boxes lo[3], hi[3] of every `tile`
 * consecutive sorted points.  scratch: zs_morton_scratch_bytes(n).
 * zs_pose_pack writes a sorted cloud as a PACK (zs_pose_pack_bytes(points)): [sub-tile of 64][x | y | z][64]
 * coordinates, the exact box of every sub-tile and of every tile of 16 - the form in which a wave reads candidates
 * through the scalar cache.  zs_pose_search_batch_sorted is zs_pose_search_batch on such clouds: statistics from `pred`
 * in the caller's order - the order zs_pose_apply sees - nearest neighbours from the sorted ones.  mode 0: the all-pairs
 * kernel of zs_pose_search_batch (gt_sorted; gt_pack unused); mode 1: all pairs on packs; mode 2: every (query,
 * 64 candidates) block whose box is farther than the query's current nearest neighbour is skipped, tiles nearest first.
 * All three give the same minima and the same fixed-order sums - the same record, bit for bit; mode 2 from a fraction of
 * the distance evaluations (utils/eval_3D.py:140-170 evaluates them all).
 * scratch: zs_pose_sorted_scratch_bytes(n, m, count) (one pack of the prediction per rotation of the batch). */
size_t zs_morton_scratch_bytes(int n);
int zs_morton_sort(const float *points, int n, float *sorted, int *perm, float *tile_boxes, int tile, void *scratch,
                   void *stream);
/* Sort-tile-recursive order (x slabs, y strips inside a slab, z inside a strip; cuts at whole leaves of 64 points, three
 * stable radix sorts): leaves with near-cubic boxes - the default order of the pose search, which then evaluates about
 * half the pairs it needs with the Z-order.  scratch: zs_str_scratch_bytes(n); perm may be NULL. */
size_t zs_str_scratch_bytes(int n);
int zs_str_sort(const float *points, int n, float *sorted, int *perm, void *scratch, void *stream);
size_t zs_pose_pack_bytes(int points);
int zs_pose_pack(const float *sorted_points, int points, float *pack, void *stream);
size_t zs_pose_sorted_scratch_bytes(int n, int m, int count);
int zs_pose_search_batch_sorted(const float *pred, const float *pred_sorted, int n, const float *gt_sorted,
                                const float *gt_pack, int m, const float *rotations, const int *order, int count,
                                int index_offset, const float *lower_bound, const float *thresholds6, float *best,
                                void *scratch, int mode, void *stream);
int zs_pose_apply(const float *pred, int n, const float *rotations, const int *index, float *out,
                  void *scratch, void *stream);
int zs_normalize_pc(const float *pc, int b, int n, float *out, void *scratch, void *stream);
/* standardize_pc (utils/eval_3D.py:83-91): zero mean, RMS distance from the origin 1/2; pc, out [b][n][3].
 * zs_icp_step: one iteration of the reference's ICP (utils/eval_3D.py:276-283) after its Chamfer call: idx1 [b][n] =
 * nearest neighbour of x1's points in x2 (zs_chamfer_forward's idx1) -> centroids, 3x3 cross-covariance (double sums),
 * R = V U^T from its SVD (Jacobi on the device) with the reference's sign rule, x1_out = (x1 - t1) R^T + t2.  x1_out
 * must not alias x1.  scratch: zs_icp_scratch_bytes(b) (R, t1, t2 per batch element stay there). */
int zs_standardize_pc(const float *pc, int b, int n, float *out, void *stream);
size_t zs_icp_scratch_bytes(int b);
int zs_icp_step(const float *x1, int n, const float *x2, int m, const int *idx1, int b, float *x1_out, void *scratch,
                void *stream);
int zs_fscore(const float *dist1, int n, const float *dist2, int m, int b, const float *thresholds,
              int n_thresholds, float *out, void *stream);

/* ------------------------------------------------------------------------- *
 * Iso-surface extraction + surface sampling (replaces convert_to_explicit,
 * utils/eval_3D.py:233-263: PyMCubes marching_cubes + trimesh sample on the host).
 * Case tables come from the caller (zeroshape_amd/mc_tables.py): tri_table int8
 * [256][table_stride] (edge ids, -1 padded), tri_count uint8 [256].
 *
 *   zs_mc_count  : triangle counts per unit of 256 cubes -> exclusive offsets and the list of non-empty
 *                  units in `scratch` (zs_mc_scratch_bytes(G), 8-byte aligned) and the grand total in
 *                  *total (device int); one read of the volume with 16-byte row loads
 *   zs_mc_emit   : tris[n_tris][3][3] fp32 triangle soup in world space
 *                  (index * scale + offset), cube order x-slowest, deterministic; reads the volume
 *                  only in the non-empty units (one wave each, a lane per triangle)
 *   zs_mesh_sample: n_samples area-weighted points (counter-based RNG, `seed`);
 *                  cum_area = scratch of zs_mesh_sample_scratch_doubles(n_tris) doubles (the cumulative
 *                  areas and the tile totals of their scan); an empty mesh yields zeros
 * vol is [G][G][G] fp32 (x slowest), a cube corner is "inside" when value < iso.
 * ------------------------------------------------------------------------- */
size_t zs_mc_scratch_bytes(int G);
size_t zs_mesh_sample_scratch_doubles(int n_tris);
int zs_mc_count(const float *vol, int G, float iso, const uint8_t *tri_count, void *scratch,
                int *total, void *stream);
int zs_mc_emit(const float *vol, int G, float iso, const int8_t *tri_table, int table_stride,
               const uint8_t *tri_count, const void *scratch, float scale, float offset,
               float *tris, int n_tris, void *stream);
int zs_mesh_sample(const float *tris, int n_tris, int n_samples, uint64_t seed, double *cum_area,
                   float *points, void *stream);

/* ------------------------------------------------------------------------- *
 * Seen-surface geometry front-end (model/compute_graph/graph_shape.py:131-144).
 * All maps are fp32, row-major, batch outermost.
 *
 *   zs_intr_param2mtx : params [B][3] = (scale_f, delta_cx, delta_cy) -> intr [B][3][3]
 *                       (graph_shape.py:89-113: f = 1.3875 * W * 4^tanh(p0), ...)
 *   zs_unproj_depth   : depth [B][H][W], intr [B][3][3] -> points [B][H*W][3] =
 *                       K^-1 [x,y,1]^T * depth (utils/camera.py:88-108)
 *   zs_valid_norm_fac : points [B][n][3], mask [B][n] bytes (torch.bool) -> mean [B][3] of
 *                       the selected points and max_dist [B] = max |p - mean|
 *                       (utils/camera.py:52-78).  An empty selection gives NaN (the
 *                       reference raises there).
 *   zs_masked_resample: interpolate_coordmap / interpolate_depth (utils/util.py:323-345):
 *                       map [B][C][H][W], mask [B][1][H][W] (valid where > 0.5) ->
 *                       out [B][C][Ho][Wo] = bilinear(map*mask)/(bilinear(mask)+1e-6) where
 *                       bilinear(mask) > 0.5 else `bg`; mask_out [B][1][Ho][Wo] in {0,1}.
 *                       Bilinear = torch interpolate(mode='bilinear', align_corners=False).
 *   zs_seen_surface   : the whole chain in one launch: unproject, masked mean / max radius
 *                       (-> mean [B][3], scale [B]), seen_points [B][H*W][3] =
 *                       (p - mean) / scale with invalid pixels zeroed, and, when coord_dsp
 *                       is not NULL, coord_dsp [B][3][Ho][Wo] + mask_dsp [B][1][Ho][Wo] =
 *                       interpolate_coordmap of the channel-major seen map.
 * ------------------------------------------------------------------------- */
int zs_intr_param2mtx(const float *params, int batch, int H, int W, float *intr, void *stream);
int zs_unproj_depth(const float *depth, const float *intr, int batch, int H, int W, float *points,
                    void *stream);
int zs_valid_norm_fac(const float *points, const uint8_t *mask, int batch, int n, float *mean,
                      float *max_dist, void *stream);
int zs_masked_resample(const float *map, const float *mask, int batch, int channels, int H, int W,
                       int Ho, int Wo, float bg, float *out, float *mask_out, void *stream);
int zs_seen_surface(const float *depth, const float *intr, const float *mask, int batch, int H, int W,
                    int Ho, int Wo, float *seen_points, float *mean, float *scale, float *coord_dsp,
                    float *mask_dsp, void *stream);
/* The same chain cut into 64 pixel chunks per image (three launches: chunk sums -> mean + chunk max radius -> scale +
 * normalise / resample): 80 -> ~12 us at batch 1, where one workgroup per image is a chain of dependent passes through one
 * CU.  The chunk sums are added in chunk order (reproducible; a different association than the one-workgroup form, ~1e-7
 * relative).  workspace: zs_seen_surface_workspace_bytes(batch) bytes; NULL = zs_seen_surface. */
size_t zs_seen_surface_workspace_bytes(int batch);
int zs_seen_surface_ws(const float *depth, const float *intr, const float *mask, int batch, int H, int W,
                       int Ho, int Wo, float *seen_points, float *mean, float *scale, float *coord_dsp,
                       float *mask_dsp, void *workspace, void *stream);

/* ------------------------------------------------------------------------- *
 * Depth metrics with least-squares scale/shift alignment in disparity space
 * (DepthMetric.compute_metrics, utils/eval_depth.py:46-116).  prediction, target,
 * mask: [B][n] fp32 (valid where mask > 0.5); thresholds: [host] n_thresholds <= 8
 * floats; depth_cap <= 0 means "no cap".  metrics [B][n_thresholds + 3] =
 * (d>thr_0 .. d>thr_k, rmse, l1_err, abs_rel); prediction_depth [B][n] (may be NULL) =
 * the aligned depth of every pixel; scale_shift [B][2] (may be NULL).
 * flags: ZS_DEPTH_PRED_IS_DISPARITY = prediction_type 'disparity' (:67-68);
 * ZS_DEPTH_SOLVE_ONLY = compute_scale_and_shift alone (:11-34): prediction and target are
 * used as given and only scale_shift is written.
 * ------------------------------------------------------------------------- */
#define ZS_DEPTH_PRED_IS_DISPARITY 1
#define ZS_DEPTH_SOLVE_ONLY 2
int zs_depth_metrics(const float *prediction, const float *target, const float *mask, int batch, int n,
                     int flags, float depth_cap, const float *thresholds,
                     int n_thresholds, float *metrics, float *prediction_depth, float *scale_shift,
                     void *stream);

/* ------------------------------------------------------------------------- *
 * Encoder layers (inference).  Activations are fp32 channels-last: [B][H][W][C]; a token
 * matrix [n][C] is the B=1, H=1, W=n case.
 *
 * zs_conv2d_nhwc: convolution / linear layer as an implicit GEMM on the fp32 MFMA pipe.
 *   packed_w: zs_conv2d_packed_floats(Cin,Cout,kh,kw) floats laid out [K16/4][CoutPad][4],
 *             value W[cout][tap][cin] at k = tap*Cin + cin (tap = ky*kw + kx), K16 = K rounded
 *             up to 16, CoutPad = Cout rounded up to 128, zero padded.  Cin % 4 == 0.
 *   out[p][c] = act( (sum_k A[p][k] W[k][c]) * scale[c] + shift[c] + res1[p][c] + res2[p][c] )
 *   A = input taps at (oy*stride - pad_t + ky, ox*stride - pad_l + kx); out-of-range taps are 0;
 *   in-range taps are first ReLU'd (flags & ZS_CONV_IN_RELU) and mapped a*in_scale + in_shift.
 *   scale / shift / res1 / res2 may be NULL (1 / 0 / none).  Explicit top/left padding with
 *   bounds-checked bottom/right covers torch's symmetric padding and timm's 'same' padding.
 *   Two tilings, chosen by problem size: 128x128 (LDS double-buffered) when that yields >= 192
 *   workgroups, else 32x64 with K split over the four waves and a fixed-order LDS reduction.
 * zs_group_norm_nhwc: y = GN_groups(x) * gamma + beta (+ residual) (ReLU if relu), per sample.
 * zs_layer_norm: rows of length C.
 * zs_attention: qkv [B][L][3*heads*head_dim] (q | k | v, head-major inside each) ->
 *   out [B][L][heads*head_dim] = softmax(q k^T / sqrt(head_dim)) v.  head_dim 32 or 64.
 * zs_max_pool_nhwc: k x k window, -inf padding.   zs_global_mean_nhwc: [B][HW][C] -> [B][C].
 * zs_upsample2x_nhwc: bilinear, align_corners=True, [B][H][W][C] -> [B][2H][2W][C].
 * zs_nchw_to_nhwc / zs_nhwc_to_nchw: boundary layout conversion (Cpad >= C zero-fills; mask [B][HW],
 *   may be NULL, multiplies every channel of a pixel: CoordEncRes' coord*mask, seen_coord_enc.py:184).
 * zs_assemble_tokens: tokens [B][n+1][C] = [cls | feat [B][n][C]] + pos [n+1][C].
 * zs_readout_concat: out [B][n][2C] = [tokens[b][1+i] | tokens[b][0]]  (vit.py:31-43).
 * zs_window_tokens: CoordEmb's token preparation (seen_coord_enc.py:50-71): emb [B][H][W][C],
 *   mask [B][H][W] bytes (invalid pixels take invalid_token [C]), cls [C], pos [win*win+1][C] ->
 *   out [B*(H/win)*(W/win)][win*win+1][C].
 * ------------------------------------------------------------------------- */
#define ZS_ACT_NONE 0
#define ZS_ACT_RELU 1
#define ZS_ACT_GELU 2
#define ZS_ACT_RELU_CLAMP1 3
#define ZS_ACT_SOFTPLUS 4 /* torch Softplus(beta, threshold=20); zs_act_forward / zs_act_backward only */
#define ZS_CONV_IN_RELU 1
#define ZS_CONV_FORCE_LARGE 2 /* tiling override (tests / tuning): 128x128 tiles */
#define ZS_CONV_FORCE_SMALL 4 /* 32x64 tiles with the K range split over the 4 waves */
#define ZS_CONV_F16X3 16      /* split-fp16 arithmetic on the 16-bit matrix pipe (csrc/zs_split16.h): operands
                               carried as two fp16 halves (~2^-21 relative for 2e-4 <~ |x| <= 65504), three
                               K = 16 MFMAs per eight fp32 ones; inference */
#define ZS_CONV_SPLIT_SMALL 32 /* with a workspace: small-tile layers that would leave most CUs idle split K across
                               * workgroups (partial sums in the workspace, fixed-order reduction kernel) */
#define ZS_CONV_STREAM_K 64   /* with a workspace: 128x128-tile layers run as a fixed number of workgroups that share the
                               * (tile, k-step) iteration space evenly; shared tiles are summed in k order by the
                               * workgroup that arrives last (deterministic) */
#define ZS_CONV_STREAM_K_ALWAYS 256 /* tests / tuning: with ZS_CONV_STREAM_K, stream-K wherever the kernels support it (by
                                     * default only where it was measured to pay: see csrc/nn_conv.hip) */
#define ZS_CONV_W_PRESPLIT 128 /* with ZS_CONV_F16X3: packed_w is the output of zs_conv2d_presplit_weight (the weights'
                               * fp16 halves, same size and indexing as the fp32 packing) - no operand split of the
                               * weights at run time */
#define ZS_CONV_IN_DILATE2 8  /* read the input as if zero-stuffed x2 ([B][2H-1][2W-1][Cin] virtual): the
                                 data gradient of a stride-2 convolution as a stride-1 convolution */
#define ZS_CONV_FORCE_TILE256 1024 /* tests / tuning: the 256 x 256 ping-pong kernel (csrc/nn_conv_pp256.h) for every layer it can
                                    * run - pointwise, ZS_CONV_F16X3 | ZS_CONV_W_PRESPLIT, Cin % 32 == 0, Cout % 4 == 0 - whatever
                                    * its size (by default: K >= 768 and at least half a round of tiles, or K >= 2048) */
#define ZS_CONV_OUT_K16 2048  /* pointwise split-fp16 layers of few rows only (the streaming GEMM kernel, csrc/nn_gemm_stream.hip): `out`
                              * is written K16-MAJOR, [Cout / 16][M][16] floats instead of [M][Cout] (Cout % 16 == 0, no residuals, no
                              * statistics) ... */
#define ZS_CONV_IN_K16 4096   /* ... and `in` is read that way ([Cin / 16][M][16]): the 32 rows of a K = 16 step are 2 KiB of consecutive
                              * bytes for the consumer (the ViT MLP's hidden tensor at batch 1: fc2 20 -> 14 us).  A layer the kernel
                              * does not take is refused (returns 0): ask zs_conv2d_k16_ok first. */
#define ZS_CONV_IN_UPSAMPLE2 512 /* zs_conv3x3_tail_nhwc only: `in` is [B][H/2][W/2][Cin] and the layer runs on its x2 bilinear
                                    up-sampling (align_corners, zs_upsample2x_nhwc's formula), which is never written */
size_t zs_conv2d_packed_floats(int Cin, int Cout, int kh, int kw);
/* packed_w (zs_pack_conv_weight / nn/pack.py layout, zs_conv2d_packed_floats() floats) -> split_w, same size: the
 * split-fp16 halves of every weight in the order the ZS_CONV_F16X3 kernels consume them (ZS_CONV_W_PRESPLIT). */
int zs_conv2d_presplit_weight(const float *packed_w, float *split_w, int Cin, int Cout, int kh, int kw, void *stream);
/* The same for n operands in ONE launch (what optim.amp does after every optimiser step, right behind
 * zs_pack_conv_weight_multi): device arrays packed[n], split[n], cout_pad[n] (the operands' padded column counts) and
 * pair_prefix[n + 1] = running sum of pairs_i = K16_i / 16 * 2 * cout_pad_i (pair_prefix[n] == total_pairs). */
int zs_conv2d_presplit_weight_multi(const void *const *packed, void *const *split, const int *cout_pad,
                                    const unsigned long long *pair_prefix, int n, unsigned long long total_pairs,
                                    void *stream);
int zs_conv2d_nhwc(const float *in, const float *packed_w, const float *scale, const float *shift,
                   const float *res1, const float *res2, float *out, int batch, int Hin, int Win, int Cin,
                   int Hout, int Wout, int Cout, int kh, int kw, int stride, int pad_t, int pad_l, int flags,
                   float in_scale, float in_shift, int act, void *stream);
/* As zs_conv2d_nhwc, with a workspace of zs_conv2d_splitk_workspace_bytes() bytes (16-byte aligned, ZEROED ONCE by
 * the caller when allocated - its first 1 MiB are arrival counters the kernels leave at zero - and used by one
 * stream at a time; NULL = zs_conv2d_nhwc).  What uses it is chosen by the flags: ZS_CONV_SPLIT_SMALL (problems
 * that would launch fewer than 256 small-tile workgroups - 14x14 feature maps, 197-token matrices at small
 * batch - split K across workgroups; a second kernel sums the partial tiles in split order and applies the
 * fused epilogue) and ZS_CONV_STREAM_K (128x128-tile problems; see above).  Independently of those flags, with
 * ZS_CONV_F16X3 | ZS_CONV_W_PRESPLIT a workspace lets two more paths split the contraction across workgroups the same
 * way (partial tiles, summed in order): 3x3 stride-1 layers of 64..191 input-patch workgroups (nn_conv_patch.h) and
 * layers of 40..191 tiles with >= 96 k-steps on the LDS-DMA kernel.  All of it is deterministic. */
size_t zs_conv2d_splitk_workspace_bytes(void);
int zs_conv2d_nhwc_ws(const float *in, const float *packed_w, const float *scale, const float *shift,
                   const float *res1, const float *res2, float *out, int batch, int Hin, int Win, int Cin,
                   int Hout, int Wout, int Cout, int kh, int kw, int stride, int pad_t, int pad_l, int flags,
                   float in_scale, float in_shift, int act, void *workspace, void *stream);

/* Fused normalisations around a small-tile launch of zs_conv2d_nhwc (batch-1 encoder: one launch per convolution instead
 * of convolution + GroupNorm / LayerNorm; reference: timm ResNetV2's GroupNormAct behind every StdConv2dSame, restated in
 * oracle/standins.py, and timm Block's norm1 / norm2, /root/reference/model/depth/vit.py:118-154).
 *   out_mode 1: besides `out`, the launch writes out_stats [ceil(M / 32)][out_groups][2] = (sum, sum of squares) of the
 *               stored values per 32-row tile and group of Cout / out_groups channels.
 *   out_mode 2: out_stats [M][ceil(Cout / tile columns)][2] = (sum, M2 about the tile-row mean) per row and column tile
 *               (zs_conv2d_fused_cols() columns per tile for this problem).
 *   in_mode 1:  A = relu(GroupNorm_{in_groups}(in) * in_gamma + in_beta), statistics from in_stats [in_tiles][in_groups][2]
 *               (a producer's out_mode 1 over the same tensor); zero padding stays zero.
 *   in_mode 2:  A = (in - mean_row) * rstd_row from in_stats [M][in_tiles <= 32][2] (a producer's out_mode 2); gamma / beta are
 *               expected inside packed_w / shift (W' = gamma * W, shift' = shift + beta W).  Pointwise layers only.
 * Statistics tiles must not straddle samples (batch 1, or Hin * Win a multiple of 32).  Requires ZS_CONV_F16X3 |
 * ZS_CONV_W_PRESPLIT, Cin % 8 == 0, Cin = 32 * 2^k <= 1024 for in_mode 1, Cout % 4 == 0; forces the small-tile kernel. */
typedef struct zs_conv_fuse {
    int in_mode, in_tiles, in_groups;
    int in_gshift;            /* set by the library (log2 of the channels per group) */
    const float *in_stats, *in_gamma, *in_beta;
    float in_eps;
    int out_mode, out_groups;
    float *out_stats;
} zs_conv_fuse;
int zs_conv2d_nhwc_fused(const float *in, const float *packed_w, const float *scale, const float *shift,
                         const float *res1, const float *res2, float *out, int batch, int Hin, int Win,
                         int Cin, int Hout, int Wout, int Cout, int kh, int kw, int stride, int pad_t,
                         int pad_l, int flags, float in_scale, float in_shift, int act, const zs_conv_fuse *fuse,
                         void *workspace, void *stream);
/* columns per tile (32 or 64) the fused small-tile launch uses for a problem of M rows and Cout columns: the consumer of
 * out_mode 2 statistics needs ceil(Cout / this) as its in_tiles */
int zs_conv2d_fused_cols(int M, int Cout);
/* 1 when a pointwise layer of M rows, Cin -> Cout, with `flags` (ZS_CONV_IN_K16 and / or ZS_CONV_OUT_K16) would be accepted:
 * ln_in_tiles > 0 = fuse.in_mode 2 with that many statistics tiles, row_stats_out = fuse.out_mode 2, has_res = a residual. */
int zs_conv2d_k16_ok(int M, int Cin, int Cout, int flags, int ln_in_tiles, int row_stats_out, int has_res);
/* y = [relu]( GroupNorm_32(x) * gamma + beta + r ) in ONE pass over x, from the (sum, sum of squares) tiles a fused launch wrote
 * (out_mode 1; `tiles` per sample).  r = residual [B][HW][C] (may be NULL), or GroupNorm_32(residual) * res_gamma + res_beta when
 * res_stats is given (the projection shortcut of a bottleneck's first block). */
/* y = max_pool_{k, stride, -inf padding}( relu(GroupNorm_32(x) * gamma + beta) ) from a fused launch's group statistics (timm
 * ResNetV2 stem: StdConv -> GroupNormAct -> MaxPool2dSame): one small launch turns the tile sums into `table` [B][C][2]
 * (scale, shift; caller-provided scratch), the pooling launch applies them on load - the normalised map is never written. */
int zs_gn_relu_max_pool_nhwc(const float *x, const float *stats, int tiles, const float *gamma, const float *beta,
                             float *table, float *y, int batch, int Hin, int Win, int C, int Hout, int Wout, int k,
                             int stride, int pad_t, int pad_l, float eps, void *stream);
int zs_group_norm_apply_stats(const float *x, const float *stats, int tiles, const float *gamma, const float *beta,
                              const float *residual, const float *res_stats, int res_tiles, const float *res_gamma,
                              const float *res_beta, float *y, int batch, int HW, int C, float eps, int relu, void *stream);
/* A 3x3 stride-1 pad-1 layer of at most 32 output channels with a fused pointwise TAIL to ONE channel - DPT's depth head,
 * model/depth/dpt_depth.py (reference: DPT output_conv[2..5]: Conv 128 -> 32 3x3, ReLU, Conv 32 -> 1, ReLU):
 *   out[b][y][x] = tail_act( tail_b[0] + sum_c tail_w[c] * act( conv3x3(in)[b][y][x][c] * scale[c] + shift[c] ) )
 * in [B][H][W][Cin] (Cin % 16 == 0), out [B][H][W] floats; packed_w = zs_conv2d_presplit_weight output of the 3x3 layer
 * (split-fp16 arithmetic only: flags must hold ZS_CONV_F16X3 | ZS_CONV_W_PRESPLIT; ZS_CONV_IN_RELU and ZS_CONV_IN_UPSAMPLE2 - H, W
 * even, the OUTPUT size - optional); tail_w [Cout],
 * tail_b [1] (NULL = 0).  The 32-channel intermediate ([B][H][W][32], 180 MB at 224^2 x 28) is never written.  Returns 0 with
 * zs_last_error set for geometries it does not take (the caller then issues the two layers separately). */
int zs_conv3x3_tail_nhwc(const float *in, const float *packed_w, const float *scale, const float *shift, float *out,
                         int batch, int H, int W, int Cin, int Cout, int flags, int act, const float *tail_w,
                         const float *tail_b, int tail_act, void *stream);
int zs_group_norm_nhwc(const float *x, const float *gamma, const float *beta, const float *residual, float *y,
                       int batch, int HW, int C, int groups, float eps, int relu, void *stream);
/* The same with a workspace (zs_group_norm_workspace_bytes; NULL = zs_group_norm_nhwc): tensors of >= 8 MiB take two
 * coalesced launches - per (sample, pixel chunk) group sums, then the normalisation - instead of one workgroup per
 * (sample, group) slice (whose loads use a quarter of every cache line in NHWC).  Deterministic (fixed-order sums). */
size_t zs_group_norm_workspace_bytes(int batch, int HW, int C, int groups);
int zs_group_norm_nhwc_ws(const float *x, const float *gamma, const float *beta, const float *residual, float *y,
                          int batch, int HW, int C, int groups, float eps, int relu, void *workspace, void *stream);
int zs_layer_norm(const float *x, const float *gamma, const float *beta, float *y, int rows, int C, float eps,
                  void *stream);
int zs_attention(const float *qkv, float *out, int batch, int L, int heads, int head_dim, void *stream);
/* zs_attention in split-fp16 arithmetic (three 16-bit MFMAs per product, ~2^-21 relative; csrc/zs_split16.h): the
 * inference encoders' default, like ZS_CONV_F16X3 for the convolutions. */
int zs_attention_split(const float *qkv, float *out, int batch, int L, int heads, int head_dim, void *stream);
int zs_max_pool_nhwc(const float *x, float *y, int batch, int Hin, int Win, int C, int Hout, int Wout, int k,
                     int stride, int pad_t, int pad_l, void *stream);
int zs_global_mean_nhwc(const float *x, float *y, int batch, int HW, int C, void *stream);
int zs_upsample2x_nhwc(const float *x, float *y, int batch, int Hin, int Win, int C, void *stream);
int zs_nchw_to_nhwc(const float *x, const float *mask, float *y, int batch, int C, int HW, int Cpad,
                    void *stream);
int zs_nhwc_to_nchw(const float *x, float *y, int batch, int C, int HW, void *stream);
int zs_assemble_tokens(const float *feat, const float *cls, const float *pos, float *tokens, int batch, int n,
                       int C, void *stream);
int zs_readout_concat(const float *tokens, float *out, int batch, int n, int C, void *stream);
int zs_window_tokens(const float *emb, const uint8_t *mask, const float *invalid_token, const float *cls,
                     const float *pos, float *out, int batch, int H, int W, int C, int win, void *stream);

/* ------------------------------------------------------------------------- *
 * Training (fp32; autograd over these entry points lives in zeroshape_amd/nn/autograd.py).
 * Replaces what the reference gets from torch.autograd + cuDNN/cuBLAS when train.py runs
 * Runner.train_iteration (model/shape_engine.py:248-297): Graph.forward(training=True)
 * (model/compute_graph/graph_shape.py:115-204), Loss.shape_loss (utils/loss.py:18-28),
 * loss.backward(), torch.optim.AdamW.step (model/shape_engine.py:132).
 * Every reduction runs in a fixed order (two-stage partial sums, no atomics).
 *
 * Convolution / linear layer gradients:
 *   zs_pack_conv_weight : torch-layout weight w[Cout][CinTot][kh][kw] (channel sub-range
 *       [cin0, cin0+Cin)) -> the packed operand of zs_conv2d_nhwc.  dgrad = 0: forward operand
 *       (Cin zero-padded to a multiple of 4; zs_conv2d_packed_floats(CinP, Cout, kh, kw) floats).
 *       dgrad = 1: operand of the DATA gradient, a convolution of dY (Cout zero-padded to a
 *       multiple of 4 = its input channels) with flipped taps and Cin outputs
 *       (zs_conv2d_packed_floats(CoutP, Cin, kh, kw) floats): dX = zs_conv2d_nhwc(dY, packed,
 *       stride 1, pad = k-1-pad, ZS_CONV_IN_DILATE2 when the forward stride was 2).
 *   zs_conv2d_wgrad : dw[cout][cin0+c][ky][kx] (=|+=) sum over output pixels of
 *       dy[pixel][cout] * A[pixel][(ky,kx,c)], A = the forward's input taps incl. its input
 *       transform (flags & ZS_CONV_IN_RELU, in_scale, in_shift).  in is [B][Hin][Win][CinP]
 *       (CinP % 4 == 0, channels >= Cin are padding), dy is [B][Hout][Wout][CoutP] with
 *       CoutP = Cout rounded up to 4.  db (may be NULL) receives the bias gradient [Cout] = the column
 *       sums of dy, gathered from the tiles the kernel stages anyway.
 *       workspace: zs_conv2d_wgrad_workspace_bytes(...).
 *   zs_standardize_weight(_bwd) : timm StdConv2d: rows of n = Cin*kh*kw weights,
 *       (w - mean) / sqrt(biased var + eps), and the adjoint for a gradient w.r.t. the result.
 * Pointwise / normalisation:
 *   zs_act_forward / zs_act_backward : GELU(erf), Softplus(beta), ReLU, ReLU+clamp1; backward
 *       takes `ref` = the pre-activation input (GELU, softplus) or the OUTPUT (ReLU variants).
 *   zs_add_scaled_rows : y[b][..] = x[b][..] + scale[b] * branch[b][..] (x may be NULL):
 *       residual with per-sample stochastic depth (timm DropPath, implicit.py:8,83-109).
 *   zs_column_sum : out[c] = scale * sum_rows x[row][c]   (bias gradients, reductions).
 *   zs_layer_norm_bwd : dx, dgamma, dbeta of zs_layer_norm (statistics recomputed from x).
 *   zs_attention_bwd : dqkv [B][L][3*heads*head_dim] of zs_attention; L <= 512.
 * Decoder (model/shape/implicit.py:25-79, ImplFuncAttention):
 *   zs_point_attention : out[b][i] = softmax over (Ll latent keys + the point itself) of
 *       q_i k^T / sqrt(d), times the values; qkv_points [B][M][3*heads*32], qkv_latent
 *       [B][Ll][3*heads*32] (q | k | v, head-major).  Ll <= 256, head_dim 32.
 *   zs_point_attention_bwd : dqkv_points (all of q, k, v), dqkv_latent: k and v columns
 *       written (accumulate_latent = 0, q columns zeroed) or added to (accumulate_latent = 1).
 * Loss / optimiser:
 *   zs_bce_logits(_bwd) : Loss.shape_loss: mean over n of w * BCEWithLogits(logit, sdf < 0),
 *       w = impt_weight where |sdf| < impt_thres else 1; backward scales by *grad_loss (device).
 *   zs_adamw_multi : torch.optim.AdamW (decoupled decay, bias correction, eps outside the
 *       sqrt) over a DEVICE table of tensors in one launch; chunk c = elements
 *       [chunk_start[c], +zs_multi_tensor_chunk_elems()) of tensor chunk_tensor[c].
 *       grad_scale (device scalar, may be NULL) multiplies every gradient (clipping, loss-scale removal);
 *       a grad_scale of 0 or nan skips the whole update (an overflowed gradient under loss scaling).
 *       `step` counts the calls; *skipped_steps (device int, may be NULL) the calls that were skipped that way:
 *       the bias corrections use step - *skipped_steps, as torch's GradScaler never steps on an overflow.
 *   zs_copy_multi  : entry.param[i] = entry.grad[i] * scale (gradient bucketing for all-reduce).
 *   zs_sumsq_multi : *sumsq = sum of entry.grad[i]^2 over the table (partial: n_chunks floats).
 * ------------------------------------------------------------------------- */
typedef struct zs_tensor_entry {
    float *param;
    const float *grad;
    float *exp_avg, *exp_avg_sq;
    unsigned long long n;
    float lr, weight_decay;
} zs_tensor_entry;

int zs_pack_conv_weight(const float *w, float *packed, int Cout, int Cin, int cin0, int CinTot, int kh, int kw,
                        int dgrad, void *stream);
/* zs_pack_conv_weight over a DEVICE table of layers in one launch (every operand of a model after
 * an optimiser step): ld = CinTot*kh*kw, taps = kh*kw, K16 / NPad = the padded operand dimensions
 * (K rounded up to 16, N rounded up to 128); chunk c = the chunk_start[c]-th of the
 * zs_pack_entry_chunks(...) chunks of entry chunk_entry[c] (an LDS tile of 64 operand columns x 16 or 64
 * channels x all taps for kernels up to 3x3, zs_pack_chunk_elems() consecutive elements otherwise), every chunk
 * of an entry exactly once.  The operand must have been packed once by zs_pack_conv_weight (its K padding rows
 * are not rewritten). */
typedef struct zs_pack_entry {
    const float *src;
    float *dst;
    int Cout, Cin, cin0, ld, taps, dgrad, K16, NPad;
} zs_pack_entry;
int zs_pack_chunk_elems(void);
int zs_pack_entry_chunks(int Cout, int Cin, int taps, int dgrad, int K16, int NPad);
int zs_pack_conv_weight_multi(const zs_pack_entry *table, const int *chunk_entry, const unsigned long long *chunk_start,
                              int n_chunks, void *stream);
/* The same launch also writing the fp16 halves of the operands (the layout of zs_conv2d_presplit_weight) from the tiles it
 * holds: split_dst[e] = the split buffer of entry e (K16 * NPad floats) or NULL; non-NULL only for entries for which
 * zs_pack_entry_inline_split(Cout, Cin, taps, dgrad) returns 1 (kernels up to 3x3 whose K-side channel count, rounded up
 * to 4, is a multiple of 16).  split_only != 0: the fp32 operand (entry.dst) of those entries is NOT rewritten - for
 * callers whose every consumer reads the halves (optim.amp: one pass over the weights instead of re-pack + split). */
int zs_pack_entry_inline_split(int Cout, int Cin, int taps, int dgrad);
int zs_pack_conv_weight_multi_split(const zs_pack_entry *table, const int *chunk_entry,
                                    const unsigned long long *chunk_start, int n_chunks, float *const *split_dst,
                                    int split_only, void *stream);
size_t zs_conv2d_wgrad_workspace_bytes(int batch, int Hout, int Wout, int Cin, int Cout, int kh, int kw);
int zs_conv2d_wgrad(const float *in, const float *dy, float *dw, float *db, void *workspace, int batch, int Hin, int Win,
                    int CinP, int Hout, int Wout, int Cout, int kh, int kw, int stride, int pad_t, int pad_l,
                    int flags, float in_scale, float in_shift, int Cin, int cin0, int CinTot, int accumulate,
                    void *stream);
/* Data gradient of a convolution with Cin <= 4 input channels (the network stems: as a GEMM it would
 * fill 3 of 128 tile columns), direct gather form: dx [B][H][W][CinP] (padding channels zeroed) =
 * scale * sum over the output pixels that read the input pixel of dy [B][Hout][Wout][Cout] . w
 * (torch layout, channel sub-range [cin0, cin0+Cin) of CinTot).  Cout % 4 == 0. */
int zs_conv2d_dgrad_small_cin(const float *dy, const float *w, float *dx, int batch, int H, int W, int CinP, int Hout,
                              int Wout, int Cout, int kh, int kw, int stride, int pad_t, int pad_l, int Cin, int cin0,
                              int CinTot, float scale, void *stream);
int zs_standardize_weight(const float *w, float *out, int Cout, int n, float eps, void *stream);
/* zs_standardize_weight over a device table of weights in one launch: entry e = rows x n weights (rows = Cout, n = the
 * fan-in) from w to out with its eps; row_prefix[e] = the number of rows of the entries before e (n_entries ints,
 * row_prefix[0] = 0), total_rows = their sum.  Same arithmetic as the single call. */
typedef struct zs_std_entry {
    const float *w;
    float *out;
    int rows, n;
    float eps;
    int pad;
} zs_std_entry;
int zs_standardize_weight_multi(const zs_std_entry *table, const int *row_prefix, int n_entries, int total_rows, void *stream);
int zs_standardize_weight_bwd(const float *w, const float *grad_out, float *dw, int Cout, int n, float eps,
                              void *stream);
int zs_act_forward(const float *x, float *y, size_t n, int act, float beta, void *stream);
int zs_act_backward(const float *dy, const float *ref, float *dx, size_t n, int act, float beta, void *stream);
/* NeRF positional encoding of 3D points - the reference's get_embedder(L, 3) (utils/layers.py:8-53; used by MLPBlocks when
 * posenc_3D = L > 0, model/shape/implicit.py:139-166): out[i][0:3+6L] = [x | sin(x 2^0) | cos(x 2^0) | ... | sin(x 2^(L-1)) |
 * cos(x 2^(L-1))], each group 3 wide (x, y, z), zero padded to `stride` floats per point (stride >= 3 + 6 L). */
int zs_posenc3d(const float *points, size_t n, int L, float *out, int stride, void *stream);
int zs_add_scaled_rows(const float *x, const float *branch, const float *scale, float *y, int batch,
                       size_t per_sample, void *stream);
size_t zs_column_sum_workspace_bytes(int rows, int C);
int zs_column_sum(const float *x, float *out, int rows, int C, float scale, void *workspace, void *stream);
size_t zs_layer_norm_bwd_workspace_bytes(int rows, int C);
int zs_layer_norm_bwd(const float *dy, const float *x, const float *gamma, float *dx, float *dgamma, float *dbeta,
                      int rows, int C, float eps, void *workspace, void *stream);
/* The same with dx = (LayerNorm's input gradient) + add [rows][C] (may be NULL): the gradient arriving at x from its other
 * consumer (a residual connection), summed in the same pass instead of by a separate elementwise launch. */
int zs_layer_norm_bwd_add(const float *dy, const float *x, const float *gamma, const float *add, float *dx, float *dgamma,
                          float *dbeta, int rows, int C, float eps, void *workspace, void *stream);
size_t zs_attention_bwd_workspace_bytes(int batch, int L, int heads);
int zs_attention_bwd(const float *qkv, const float *dout, float *dqkv, void *workspace, int batch, int L, int heads,
                     int head_dim, void *stream);
int zs_point_attention(const float *qkv_points, const float *qkv_latent, float *out, int batch, int M, int Ll,
                       int heads, int head_dim, void *stream);
/* The attention map a decoder call returns (implicit.py:60-66,277): attn[b][i][j] (=|+=, `accumulate`) weight * mean over the
 * heads of the softmax probability of point i for latent token j (softmax over the Ll latent keys AND the point itself; the
 * point's own column is dropped afterwards).  One call per attention block with weight = 1 / n_blocks gives the reference's
 * average over blocks.  Same size limits as zs_point_attention. */
int zs_point_attention_probs(const float *qkv_points, const float *qkv_latent, float *attn, int batch, int M, int Ll,
                             int heads, int head_dim, float weight, int accumulate, void *stream);
size_t zs_point_attention_bwd_workspace_bytes(int batch, int M, int Ll, int heads);
int zs_point_attention_bwd(const float *qkv_points, const float *qkv_latent, const float *dout, float *dqkv_points,
                           float *dqkv_latent, int accumulate_latent, void *workspace, int batch, int M, int Ll,
                           int heads, int head_dim, void *stream);
size_t zs_bce_logits_workspace_bytes(size_t n);
int zs_bce_logits(const float *logits, const float *sdf, size_t n, float impt_thres, float impt_weight, float *loss,
                  void *workspace, void *stream);
int zs_bce_logits_bwd(const float *logits, const float *sdf, size_t n, float impt_thres, float impt_weight,
                      const float *grad_loss, float *dlogits, void *stream);
/* The depth task's losses (options/depth.yaml).
 *   zs_midas_loss : MidasLoss.forward (model/depth/midas_loss.py:166-185, shrink_mask False) on
 *       prediction / target / mask [B][1][H][W] (valid where mask > 0.5): scale-and-shift-invariant MAE
 *       (median / mean-absolute-deviation alignment per image) + alpha * gradient matching over
 *       `scales` <= 4 power-of-two strides of the least-squares aligned maps (inverse_depth != 0:
 *       1 / (d + 1e-6) first), image-based reduction.  *loss is a device scalar; workspace
 *       (zs_midas_loss_workspace_bytes(B)) keeps the per-image statistics for the backward.
 *   zs_midas_loss_bwd : d loss / d prediction [B][1][H][W], scaled by *grad_loss (device), through
 *       the median element, the deviations and the 2x2 least-squares solve, like torch.autograd.
 *   zs_intr_loss(_bwd) : Loss.intr_loss (utils/loss.py:36-43): seen_* [n][3], mask [n] over the
 *       whole batch; out2 = (loss, sum of the mask) on the device. */
/* MidasLoss.erode_mask (model/depth/midas_loss.py:153-162; training.depth_loss.mask_shrink):
 * out[b][y][x] = 1 iff every value of the pool x pool block of mask[b] that holds (y, x) equals 1
 * (max_pool2d with stride = kernel, then F.interpolate(mode='nearest') back), else 0.  [B][H][W] fp32. */
int zs_erode_mask(const float *mask, int batch, int H, int W, int pool, float *out, void *stream);
size_t zs_midas_loss_workspace_bytes(int batch);
int zs_midas_loss(const float *prediction, const float *target, const float *mask, int batch, int H, int W, float alpha,
                  int scales, int inverse_depth, float *loss, void *workspace, void *stream);
int zs_midas_loss_bwd(const float *prediction, const float *target, const float *mask, int batch, int H, int W,
                      float alpha, int scales, int inverse_depth, const void *workspace, const float *grad_loss,
                      float *dprediction, void *stream);
int zs_intr_loss(const float *seen_pred, const float *seen_gt, const float *mask, size_t n, float *out2, void *stream);
int zs_intr_loss_bwd(const float *seen_pred, const float *seen_gt, const float *mask, size_t n, const float *out2,
                     const float *grad_loss, float *dseen_pred, void *stream);
int zs_multi_tensor_chunk_elems(void);
int zs_adamw_multi(const zs_tensor_entry *table, const int *chunk_tensor, const unsigned long long *chunk_start,
                   int n_chunks, float beta1, float beta2, float eps, int step, const float *grad_scale,
                   const int *skipped_steps, void *stream);
int zs_copy_multi(const zs_tensor_entry *table, const int *chunk_tensor, const unsigned long long *chunk_start,
                  int n_chunks, float scale, void *stream);
int zs_sumsq_multi(const zs_tensor_entry *table, const int *chunk_tensor, const unsigned long long *chunk_start,
                   int n_chunks, float *partial, float *sumsq, void *stream);

/* Encoder training (BatchNorm / GroupNorm / pooling / resampling / geometry backward).
 *   zs_batch_norm_train : nn.BatchNorm2d in training mode over [rows][C] (rows = B*H*W): batch
 *       mean / biased variance -> y = (x-mean)*rstd*gamma+beta (+residual) (ReLU if relu);
 *       running_mean / running_var (may both be NULL) updated with `momentum` and the unbiased
 *       variance like torch; save_mean / save_rstd [C] feed the backward.
 *   zs_batch_norm_bwd   : y_relu = the forward output when ReLU was fused (masks dy), else NULL;
 *       dresidual (may be NULL) = the masked dy.
 *   zs_group_norm_bwd   : backward of zs_group_norm_nhwc (statistics recomputed from x).
 *   zs_max_pool_bwd_nhwc / zs_global_mean_bwd_nhwc / zs_upsample2x_bwd_nhwc: adjoints of the
 *       forward layers in gather form (no atomics); max-pool routes to the first maximum of a
 *       window in row-major order, as torch does.
 *   zs_nhwc_to_nchw_masked : y[b][c][p] = x[b][p][c] * mask[b][p], c < C <= Cpad.
 *   zs_seen_surface_bwd : backward of zs_seen_surface for Ho = H, Wo = W (the ResNet coordinate
 *       encoder): d_seen_points [B][HW][3] and/or d_coord_dsp [B][3][H][W] (either may be NULL)
 *       -> d_depth [B][HW], d_intr [B][3][3], through the masked mean and the max-radius scale
 *       (utils/camera.py:52-108 as torch.autograd differentiates them).
 *   zs_intr_param2mtx_bwd : d_intr -> d_params [B][3] (graph_shape.py:89-113). */
size_t zs_batch_norm_workspace_bytes(int rows, int C);
int zs_batch_norm_train(const float *x, const float *gamma, const float *beta, const float *residual, float *y,
                        float *running_mean, float *running_var, float *save_mean, float *save_rstd, int rows, int C,
                        float eps, float momentum, int relu, void *workspace, void *stream);
int zs_batch_norm_bwd(const float *x, const float *dy, const float *y_relu, const float *gamma, const float *save_mean,
                      const float *save_rstd, float *dx, float *dresidual, float *dgamma, float *dbeta, int rows, int C,
                      void *workspace, void *stream);
size_t zs_group_norm_bwd_workspace_bytes(int batch, int C);
int zs_group_norm_bwd(const float *x, const float *dy, const float *y_relu, const float *gamma, float *dx,
                      float *dresidual, float *dgamma, float *dbeta, int batch, int HW, int C, int groups, float eps,
                      void *workspace, void *stream);
int zs_max_pool_bwd_nhwc(const float *x, const float *dy, float *dx, int batch, int Hin, int Win, int C, int Hout,
                         int Wout, int k, int stride, int pad_t, int pad_l, void *stream);
int zs_global_mean_bwd_nhwc(const float *dy, float *dx, int batch, int HW, int C, void *stream);
int zs_upsample2x_bwd_nhwc(const float *dy, float *dx, int batch, int Hin, int Win, int C, void *stream);
int zs_nhwc_to_nchw_masked(const float *x, const float *mask, float *y, int batch, int C, int HW, int Cpad,
                           void *stream);
int zs_seen_surface_bwd(const float *depth, const float *intr, const float *mask, const float *mean, const float *scale,
                        const float *d_seen_points, const float *d_coord_dsp, int batch, int H, int W, float *d_depth,
                        float *d_intr, void *stream);
int zs_intr_param2mtx_bwd(const float *params, const float *d_intr, int batch, int H, int W, float *d_params,
                          void *stream);
/* Training of the transformer coordinate encoder (CoordEmb / CoordEncAtt, seen_coord_enc.py:13-139):
 *   zs_window_tokens_bwd : adjoint of zs_window_tokens in gather form - d_out [B*nwy*nwx][win*win+1][C] ->
 *       d_emb [B][H][W][C] (valid pixels), d_inv_rows [B][H][W][C] (the rows that fed invalid_coord_token) and
 *       d_cls_rows [B*nwy*nwx][C]; their column sums (zs_column_sum) are the two token gradients.
 *   zs_coord_dsp2_bwd : adjoint of the 2x down-sampling of the seen-surface coordinate map (interpolate_coordmap,
 *       utils/util.py:336-345, dsp = 2) as the same-size gradient zs_seen_surface_bwd takes: d_dsp [B][3][H/2][W/2],
 *       mask [B][H][W], mask_dsp [B][H/2][W/2] -> d_full [B][3][H][W]. */
int zs_window_tokens_bwd(const float *d_out, const uint8_t *mask, float *d_emb, float *d_inv_rows, float *d_cls_rows,
                         int batch, int H, int W, int C, int win, void *stream);
int zs_coord_dsp2_bwd(const float *d_dsp, const float *mask, const float *mask_dsp, float *d_full, int batch, int H,
                      int W, void *stream);
/* Bilinear resize (align_corners=False) of a channels-last grid [Hi][Wi][C] -> [Ho][Wo][C]
 * (vit.py:103-120 _resize_pos_embed); backward != 0: x is the gradient of the [Ho][Wo][C] output
 * and y receives the gradient of the [Hi][Wi][C] input.  zs_readout_concat_bwd: adjoint of
 * zs_readout_concat (dout [B][n][2C] -> dtokens [B][n+1][C]). */
/* graph_shape.py:163-173: out[b][i] = ((R_b p + t_b) - mean_b) / scale_b with pose [B][3][4] = [R|t]:
 * the ground-truth query points in the normalised frame of the (GT) seen surface. */
int zs_transform_points(const float *points, const float *pose, const float *mean, const float *scale, float *out,
                        int batch, int n, void *stream);
int zs_resize_bilinear_nhwc(const float *x, float *y, int Hi, int Wi, int Ho, int Wo, int C, int backward,
                            void *stream);
int zs_readout_concat_bwd(const float *dout, float *dtokens, int batch, int n, int C, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ZEROSHAPE_HIP_H */
