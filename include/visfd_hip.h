/* visfd_hip.h -- C ABI of libvisfd_hip.so: the MI355X (gfx950) implementation of VISFD's dense
 * 3-D filtering hot path (separable Gaussian -> DoG/LoG blob detection -> Hessian/eigen ridge
 * saliency -> dense stick tensor voting).
 *
 * This is the drop-in boundary.  The reference has no FFI layer: its interface
 * for this path is the header-only template API in namespace visfd plus the filter_mrc handlers.
 * Each entry point below names the reference function it replaces (file:line under the
 * reference root).  include/visfd_hip.hpp re-creates the visfd:: templates on top of this ABI
 * (float*** arguments), INTEGRATION.md shows the binding a maintainer adds.
 *
 * Conventions
 *  - plain C, no torch/STL types; every function returns 0 on success or a VISFD_HIP_E* code;
 *    visfd_hip_last_error() returns the message of the last failure on the calling thread.
 *  - volumes are contiguous float32, row-major [iz][iy][ix], x fastest (lib/visfd/alloc3d.hpp:16-23);
 *    sizes are 64-bit (the reference's int arithmetic overflows at 2^31 voxels, alloc3d.hpp:33-35).
 *  - "mask" pointers may be NULL (= no mask); a voxel is masked out when mask == 0.
 *  - two faces per operation:
 *        visfd_hip_<op>      host pointers  (drop-in: copies in, runs, copies out, synchronous)
 *        visfd_hip_<op>_dev  device pointers (asynchronous on the context's HIP stream; the caller
 *                            owns the buffers; multi-channel fields are CHANNEL-PLANAR on the device)
 *  - multi-channel fields on the HOST face use the reference's layouts: direction/gradient as
 *    3 interleaved floats per voxel (array<float,3>***, handlers.cpp:1547-1556); Hessian / vote
 *    tensor as 6 interleaved floats per voxel in the order xx,yy,zz,xy,yz,xz
 *    (lib/visfd/lin3_utils.hpp:400-406).  On the DEVICE face they are channel-planar:
 *    field[c*nvox + voxel].
 *  - callee owns its temporaries (as the reference does: filter3d.hpp:731,1362; feature.hpp:1243),
 *    kept in the context's workspace between calls.
 */
#ifndef VISFD_HIP_H
#define VISFD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VISFD_HIP_OK 0
#define VISFD_HIP_EINVAL 1   /* bad argument (the reference would assert or throw VisfdErr) */
#define VISFD_HIP_EDEVICE 2  /* HIP runtime failure / no gfx950 device */
#define VISFD_HIP_ENOMEM 3   /* device allocation failed */
#define VISFD_HIP_ECAPACITY 4 /* an output list was too small; counts are still returned */

/* selfadjoint_eigen3::EigenOrderType (lib/visfd/eigen3_simple.hpp:36-43); only the two orders
 * the hot path uses (bin/filter_mrc/handlers.cpp:1524-1535). */
#define VISFD_HIP_INCREASING_EIVALS 0
#define VISFD_HIP_DECREASING_EIVALS 1

typedef struct visfd_hip_ctx visfd_hip_ctx;

/* ---- lifecycle ------------------------------------------------------------------------------- */
/* device: HIP ordinal.  stream: a hipStream_t to run on (e.g. the caller's), or NULL to create one. */
int visfd_hip_create(int device, void* stream, visfd_hip_ctx** out);
int visfd_hip_destroy(visfd_hip_ctx* ctx);
int visfd_hip_synchronize(visfd_hip_ctx* ctx);
/* the hipStream_t every _dev entry point of this context runs on (the one given to visfd_hip_create, or the context's own) */
void* visfd_hip_get_stream(visfd_hip_ctx* ctx);
/* release the cached workspace (it otherwise persists between calls) */
int visfd_hip_trim(visfd_hip_ctx* ctx);
const char* visfd_hip_last_error(void);
int visfd_hip_abi_version(void);   /* 9: entry points only get added between versions */
/* Tuning and test switches of a context (integers; unknown names are VISFD_HIP_EINVAL).  A new context starts from the
 * environment (VISFD_HIP_<NAME>, read once in visfd_hip_create); nothing reads the environment afterwards.
 *   gauss_3pass      1: the separable filter always takes its three single-axis passes
 *   gauss_wg_per_cu  workgroups per CU the single-sweep filter cuts the volume into (default 2)
 *   tv_dense         1: tensor voting by the baseline kernel
 *   tv_fma           1: TOLERANCE MODE of tensor voting (surfaces, angular exponent 2 or 4): the vote chain with fused
 *                    multiply-adds.  Tensors are then within 1e-5 of the field's scale of the reference's instead of
 *                    bit-identical to them (BASELINE north_star: 1e-5 relative for float voxel values); default 0 = exact
 *   gauss_fma        1: TOLERANCE MODE of the plain separable Gaussian (ApplyGauss / ApplySeparable, symmetric taps,
 *                    no mask): fused multiply-adds and a reciprocal normaliser, same 1e-5 bar.  DoG / LoG / BlobDog feed
 *                    index comparisons and ignore it (always the reference's bits); default 0 = exact
 *   eig_f32          1: TOLERANCE MODE of the device eigen solver (ridge scores and directions, post-vote score, the
 *                    diagonalise batch): its one angle -- atan2, sin, cos -- in single precision; eigenvalues move by about
 *                    one float ulp.  default 0 = the reference's double-precision angle (eigen3_simple.hpp:74-81)
 *   tv_exact_tiled   1: exact tensor voting always runs the general kernel (csrc/tv_tiled.hip), never the faster exact form of
 *                    csrc/tv_box.hip (surfaces, exponent 2 or 4, source mask absent or of zeros and ones); results are bit-identical either way
 *   tv_no_fold       tests: tolerance-mode voting keeps the saliency as a factor of every vote even when all saliencies are
 *                    positive (by default they are then folded into the listed normals: 18 instead of 19 instructions)
 *   tv_poison        tests: NaN bit patterns in LDS, list memory and the output before the kernels of csrc/tv_box.hip run (both forms)
 *   tv_reserve_wg    workgroup slots a voting kernel leaves free of its chip-filling grid (slab runs set it while a halo is in
 *                    flight, so that the transport's kernels find room; default 0)
 *   tv_zrun          receiver planes per unit of work (default 8 in csrc/tv_box.hip, 32 in csrc/tv_tiled.hip)
 *   tv_no_replay     csrc/tv_tiled.hip only: every sender plane is listed again for every receiver plane (no reuse within a run)
 *   tv_max_wg        cap on the number of persistent workgroups (tests: forces many units of work per workgroup)
 *   blob_test_cap    tests: capacity the pipelined blob scan pretends to have (exercises its overflow path)
 *   gauss_cfg, debug development aids */
int visfd_hip_set_option(visfd_hip_ctx* ctx, const char* name, int64_t value);
int visfd_hip_get_option(visfd_hip_ctx* ctx, const char* name, int64_t* value_out);
/* bytes of device workspace currently held by the context */
int64_t visfd_hip_workspace_bytes(visfd_hip_ctx* ctx);

/* ---- a1: filter taps (host arithmetic, long double) ------------------------------------------ */
/* GenFilterGauss1D<float>, lib/visfd/filter1d.hpp:409-460.  taps_out has 2*halfwidth+1 entries. */
int visfd_hip_gauss_taps(float sigma, int halfwidth, float* taps_out);
/* ratio = sqrt(-2 ln threshold) in float, bin/filter_mrc/filter3d_variants.hpp:513-518 */
float visfd_hip_ratio_from_threshold(float truncate_threshold);

/* ---- a4: ApplySeparable, lib/visfd/filter3d.hpp:686-1050 -------------------------------------- */
/* taps_d has 2*h_d+1 entries, centre at index h_d.  A_out (nullable) = product of the centre taps. */
int visfd_hip_separable3d(visfd_hip_ctx*, const float* src, float* dst, const float* mask,
                          int64_t nx, int64_t ny, int64_t nz,
                          const float* taps_x, int hx, const float* taps_y, int hy,
                          const float* taps_z, int hz, int normalize, float* A_out);
int visfd_hip_separable3d_dev(visfd_hip_ctx*, const float* src, float* dst, const float* mask,
                              int64_t nx, int64_t ny, int64_t nz,
                              const float* taps_x, int hx, const float* taps_y, int hy,
                              const float* taps_z, int hz, int normalize, float* A_out);

/* ---- a5: ApplyGauss(sigma[3], halfwidth[3]), lib/visfd/filter3d.hpp:1086-1124 ------------------ */
int visfd_hip_apply_gauss(visfd_hip_ctx*, const float* src, float* dst, const float* mask,
                          int64_t nx, int64_t ny, int64_t nz, const float sigma[3],
                          const int halfwidth[3], int normalize, float* A_out);
int visfd_hip_apply_gauss_dev(visfd_hip_ctx*, const float* src, float* dst, const float* mask,
                              int64_t nx, int64_t ny, int64_t nz, const float sigma[3],
                              const int halfwidth[3], int normalize, float* A_out);
/* halfwidth[d] = max(1, floor(sigma[d]*ratio)), lib/visfd/filter3d.hpp:1240-1247 */
int visfd_hip_gauss_halfwidths(const float sigma[3], float truncate_ratio, int halfwidth_out[3]);

/* ---- f4: LocalFluctuations (lib/visfd/filter3d.hpp:1698-1853) ---------------------------------- */
/* dst = sqrt(max(A * G((src - G(src))^2), 0)), G = ApplyGauss(sigma[3], truncate_ratio) with the same mask and
 * normalize flag, A = the central value of GenFilterGenGauss3D(sigma, exponent, truncate_ratio)
 * (filter3d.hpp:546-640, :1725, :1836).  Only exponent == 2 (the separable case) is provided; src != dst. */
int visfd_hip_local_fluctuations(visfd_hip_ctx*, const float* src, float* dst, const float* mask,
                                 int64_t nx, int64_t ny, int64_t nz, const float sigma[3], float exponent,
                                 float truncate_ratio, int normalize);
int visfd_hip_local_fluctuations_dev(visfd_hip_ctx*, const float* src, float* dst, const float* mask,
                                     int64_t nx, int64_t ny, int64_t nz, const float sigma[3], float exponent,
                                     float truncate_ratio, int normalize);
/* LocalFluctuationsByRadius (filter3d.hpp:1897-1926; bin/filter_mrc/filter3d_variants.hpp:651-681), host arithmetic:
 * sigma = radius / (9 pi / 2)^(1/6); a negative truncate_ratio is replaced by (-log threshold)^(1/exponent). */
int visfd_hip_fluctuation_sigmas(const float radius[3], float exponent, float truncate_ratio,
                                 float truncate_threshold, float sigma_out[3], float* ratio_out);

/* ---- a6: ApplyDog, lib/visfd/filter3d.hpp:1338-1402 -------------------------------------------- */
int visfd_hip_apply_dog(visfd_hip_ctx*, const float* src, float* dst, const float* mask,
                        int64_t nx, int64_t ny, int64_t nz, const float sigma_a[3],
                        const float sigma_b[3], const int halfwidth[3], float* A_out, float* B_out);
int visfd_hip_apply_dog_dev(visfd_hip_ctx*, const float* src, float* dst, const float* mask,
                            int64_t nx, int64_t ny, int64_t nz, const float sigma_a[3],
                            const float sigma_b[3], const int halfwidth[3], float* A_out,
                            float* B_out);

/* ---- a7: ApplyLog(sigma[3], delta, ratio), lib/visfd/filter3d.hpp:1428-1507 -------------------- */
int visfd_hip_apply_log(visfd_hip_ctx*, const float* src, float* dst, const float* mask,
                        int64_t nx, int64_t ny, int64_t nz, const float sigma[3],
                        float delta_sigma_over_sigma, float truncate_ratio, float* A_out,
                        float* B_out);
int visfd_hip_apply_log_dev(visfd_hip_ctx*, const float* src, float* dst, const float* mask,
                            int64_t nx, int64_t ny, int64_t nz, const float sigma[3],
                            float delta_sigma_over_sigma, float truncate_ratio, float* A_out,
                            float* B_out);

/* ---- a8: BlobDog, lib/visfd/feature.hpp:53-427 -------------------------------------------------- */
/* One record per detected blob.  ix,iy,iz are voxel indices (the reference stores them as floats,
 * feature.hpp:275-279); scale = index into blob_sigma[] of the blob's scale (ir-1 in the reference);
 * score = LoG value.  Lists are returned sorted by (scale, iz, iy, ix): the reference's in-memory
 * order depends on OpenMP scheduling (feature.hpp:310-345) and its callers sort anyway
 * (bin/filter_mrc/handlers.cpp:876-909). */
typedef struct visfd_hip_blob {
  int32_t ix, iy, iz;
  int32_t scale;
  float sigma; /* blob_sigma[scale] */
  float score;
} visfd_hip_blob;

/* aspect_ratio: NULL = {1,1,1}.  minima/maxima thresholds and use_threshold_ratios as
 * feature.hpp:72-74 (pass +INFINITY / -INFINITY to disable).  With ratios the thresholds are multiplied by the best
 * scores unconditionally, as the reference does (feature.hpp:369-372), so an infinite ratio is NOT "disabled":
 *   minima_threshold = +inf with a finite maxima ratio: no minimum is kept (inf * negative best = -inf);
 *   maxima_threshold = -inf: no maximum is ever recorded (feature.hpp:286-289: -inf * (-1) = +inf in every thread).
 * ONE FENCE: with BOTH sides infinite in ratio mode the reference keeps, of the minima, the first one each OpenMP
 * thread meets (its result depends on the thread count); this library returns all minima there (and no maxima).
 * NaN / Inf voxels propagate as in the reference (every comparison with a NaN is false in its scan, feature.hpp:245-304):
 * same lists (tests/test_gpu_parity.py::test_blob_detection_with_non_finite_voxels).
 * src and mask are HOST pointers in the first form, DEVICE pointers in the _dev form; the blob lists are always host
 * arrays of the given capacities.  Planes (nx * ny) of 2^29 voxels and more are VISFD_HIP_EINVAL (the scan addresses a
 * plane through one 2 GiB buffer descriptor). */
int visfd_hip_blob_dog(visfd_hip_ctx*, const float* src, const float* mask,
                       int64_t nx, int64_t ny, int64_t nz, const float* blob_sigma, int n_sigma,
                       const float* aspect_ratio, float delta_sigma_over_sigma,
                       float truncate_ratio, float minima_threshold, float maxima_threshold,
                       int use_threshold_ratios,
                       visfd_hip_blob* minima, int64_t minima_capacity, int64_t* n_minima,
                       visfd_hip_blob* maxima, int64_t maxima_capacity, int64_t* n_maxima);
int visfd_hip_blob_dog_dev(visfd_hip_ctx*, const float* src, const float* mask,
                           int64_t nx, int64_t ny, int64_t nz, const float* blob_sigma,
                           int n_sigma, const float* aspect_ratio, float delta_sigma_over_sigma,
                           float truncate_ratio, float minima_threshold, float maxima_threshold,
                           int use_threshold_ratios,
                           visfd_hip_blob* minima, int64_t minima_capacity, int64_t* n_minima,
                           visfd_hip_blob* maxima, int64_t maxima_capacity, int64_t* n_maxima);
/* The same detector in two halves, for hosts that have more work for the context's stream (ABI >= 9; no reference counterpart:
 * BlobDog is one call there).  `begin` queues every filter and scan and fetches the lists of all scales but the last few; `end`
 * fetches those, repeats scales whose buffers overflowed, merges, and hands the lists over exactly as visfd_hip_blob_dog_dev does.
 * Between the two the caller may queue other calls of the SAME context (a membrane stage): the device then goes from the last
 * scan straight into that work instead of idling through the host's list handling.  src and mask must stay unchanged until
 * `end` returns.  VISFD_HIP_ECAPACITY from `end` leaves the job alive and returns the counts: call `end` again with room for
 * them; every other return value of `end` -- and visfd_hip_blob_dog_abort -- frees the job. */
typedef struct visfd_hip_blob_job visfd_hip_blob_job;
int visfd_hip_blob_dog_begin_dev(visfd_hip_ctx*, const float* src, const float* mask,
                                 int64_t nx, int64_t ny, int64_t nz, const float* blob_sigma,
                                 int n_sigma, const float* aspect_ratio, float delta_sigma_over_sigma,
                                 float truncate_ratio, float minima_threshold, float maxima_threshold,
                                 int use_threshold_ratios, visfd_hip_blob_job** job_out);
int visfd_hip_blob_dog_end(visfd_hip_blob_job* job,
                           visfd_hip_blob* minima, int64_t minima_capacity, int64_t* n_minima,
                           visfd_hip_blob* maxima, int64_t maxima_capacity, int64_t* n_maxima);
void visfd_hip_blob_dog_abort(visfd_hip_blob_job* job);
/* BlobDogD's conversions, lib/visfd/feature.hpp:475 and :504 */
int visfd_hip_blob_diameters_to_sigmas(const float* diameters, int n, float* sigmas);
int visfd_hip_blob_sigmas_to_diameters(const float* sigmas, int n, float* diameters);

/* ---- f3: blob list post-processing (host-side; these take no context and touch no device) --------
 * Lists are three parallel arrays: crds[n][3] (x,y,z in voxels), diameters[n], scores[n]. */
enum {   /* SortCriteria, lib/visfd/visfd_utils.hpp:49-55 */
  VISFD_HIP_DO_NOT_SORT = 0,
  VISFD_HIP_SORT_DECREASING = 1,
  VISFD_HIP_SORT_INCREASING = 2,
  VISFD_HIP_SORT_DECREASING_MAGNITUDE = 3,
  VISFD_HIP_SORT_INCREASING_MAGNITUDE = 4
};
/* CalcSphereOverlap, lib/visfd/visfd_utils.hpp:95-118: volume shared by two spheres */
float visfd_hip_sphere_overlap(float rij, float ri, float rj);
/* SortBlobs, lib/visfd/feature.hpp:519-616: reorders the three arrays; `permutation` (nullable, n
 * entries) receives the original index of every new position.  Ties keep the reference's order:
 * (key, index) ascending, or exactly the reverse of that. */
int visfd_hip_sort_blobs(float* crds, float* diameters, float* scores, int64_t n, int sort_criteria,
                         int ascending_order, uint64_t* permutation);
/* DiscardMaskedBlobs, lib/visfd/feature.hpp:924-969: drops blobs whose centre voxel
 * floor(x+0.5) has mask == 0; *n is updated.  mask == NULL keeps everything.  A centre outside the
 * mask image is VISFD_HIP_EINVAL (the reference reads out of bounds). */
int visfd_hip_discard_masked_blobs(float* crds, float* diameters, float* scores, int64_t* n,
                                   const float* mask, int64_t nx, int64_t ny, int64_t nz);
/* DiscardOverlappingBlobs, lib/visfd/feature.hpp:720-913: sorts by `sort_criteria` (best first), then
 * keeps a blob unless an already-kept blob it meets in the coarse occupancy grid (cell = `scale`
 * voxels, reference default 6) is closer than (ri+rk)*min_radial_separation_ratio or overlaps more
 * than the given volume fractions (infinity disables a criterion).  *n is updated. */
int visfd_hip_discard_overlapping_blobs(float* crds, float* diameters, float* scores, int64_t* n,
                                        float min_radial_separation_ratio,
                                        float max_volume_overlap_large, float max_volume_overlap_small,
                                        int sort_criteria, int scale);

/* ---- f1: LabelConnected, lib/visfd/connect.hpp:168-1427 (host-side: a sequential priority flood) ------
 * Clusters the voxels whose saliency passes `threshold_saliency` into connected "islands", growing from the
 * local saliency maxima (minima if start_from_saliency_maxima == 0) in order of decreasing saliency and merging
 * islands that touch.  labels[nz][ny][nx] receives 1..n_clusters (1 = largest when sort_by_size != 0, otherwise
 * ordered by the height of the seeding maximum) and `label_undefined` for unclustered voxels; voxels with
 * mask == 0 receive the value n_seeds + 1 as in the reference (connect.hpp:1401-1403 skips them).
 * direction (nullable, [nz][ny][nx][3]) and tensor (nullable, [nz][ny][nx][6] = xx,yy,zz,xy,yz,xz) switch on the
 * compatibility tests of connect.hpp:455-553 and :625-672 with the four cosine thresholds (values below -1
 * disable a test; with consider_dot_product_sign == 0 negative vector thresholds become 0).  When
 * standardize_directions != 0 (and signs are ignored) `direction` is rewritten in place with consistent,
 * outward-pointing signs (the reference's aaaafVectorStandardized aliasing its aaaafVector, as in
 * handlers.cpp:1985-2013).  Per-cluster outputs (each nullable; cluster_capacity entries): seed position
 * x,y,z in final cluster order; sizes and seed saliencies in the provisional (seed-height) order -- the
 * reference only permutes the positions (connect.hpp:1294-1348).  Every dimension must be >= 3.
 * The _ex form adds the reference's remaining optional arguments:
 *   voxel_weights [nz][ny][nx] (nullable): a cluster's "size" is the sum of its voxels' weights (connect.hpp:1154-1183);
 *   must-link constraints (connect.hpp:829-1045): n groups of locations (x,y,z, voxels; must_link_crds[total][3],
 *   must_link_group_sizes[n]); the clusters of the clustered voxels nearest to consecutive locations of a group are
 *   merged.  must_link_directions (nullable, one per location): 0 = the two surfaces face the same way, 1 = opposite,
 *   2 = decide from the angles their normals make with the joining line (DirectionPairType). */
int visfd_hip_label_connected(const float* saliency, int64_t* labels, const float* mask, int64_t nx, int64_t ny,
                              int64_t nz, float threshold_saliency, float* direction,
                              float threshold_vector_saliency, float threshold_vector_neighbor,
                              int consider_dot_product_sign, const float* tensor,
                              float threshold_tensor_saliency, float threshold_tensor_neighbor,
                              int tensor_is_positive_definite_near_target, int connectivity,
                              int64_t label_undefined, int sort_by_size, int standardize_directions,
                              int start_from_saliency_maxima, int64_t* n_clusters, float* cluster_maxima,
                              float* cluster_sizes, float* cluster_saliencies, int64_t cluster_capacity);
int visfd_hip_label_connected_ex(const float* saliency, int64_t* labels, const float* mask, int64_t nx, int64_t ny,
                              int64_t nz, float threshold_saliency, float* direction,
                              float threshold_vector_saliency, float threshold_vector_neighbor,
                              int consider_dot_product_sign, const float* tensor,
                              float threshold_tensor_saliency, float threshold_tensor_neighbor,
                              int tensor_is_positive_definite_near_target, int connectivity,
                              int64_t label_undefined, int sort_by_size, int standardize_directions,
                              int start_from_saliency_maxima, int64_t* n_clusters, float* cluster_maxima,
                              float* cluster_sizes, float* cluster_saliencies, int64_t cluster_capacity,
                                 const float* voxel_weights, const float* must_link_crds,
                                 const int64_t* must_link_group_sizes, int64_t must_link_ngroups,
                                 const int* must_link_directions);

/* Principal eigenvector (eigenvector row 0 of ConvertFlatSym2Evects3 in the given order) of nvox flat tensors
 * [nvox][6] -> direction [nvox][3], computed on the HOST in the reference's arithmetic (the loop of
 * bin/filter_mrc/handlers.cpp:1935-1952); voxels with mask == 0 (mask nullable) are left untouched. */
int visfd_hip_principal_directions_host(const float* tensor, const float* mask, int64_t nvox, int order,
                                        float* direction);
/* Post-vote score lambda0 - lambda1 (bin/filter_mrc/handlers.cpp:1868-1888) of nvox flat tensors on the HOST in
 * the reference's arithmetic (the device kernel visfd_hip_tensor_saliency agrees to ~1e-7 relative only);
 * used before visfd_hip_label_connected, whose flood order and thresholds act on this number. */
int visfd_hip_tensor_saliency_host(const float* tensor, const float* mask, int64_t nvox, int order,
                                   float* saliency);

/* DiagonalizeFlatSym3 (lib/visfd/eigen3_simple.hpp:271-342) of n interleaved flat matrices [n][6] ->
 * [n][lambda0,lambda1,lambda2, shoemake0..2] on the HOST, bit-identical to the reference (the form a caller uses
 * per voxel inside their own loops; visfd_hip_diagonalize_flat_sym3 is the device batch). order 0/1. */
int visfd_hip_diagonalize_flat_sym3_host(const float* m6, float* out6, int64_t n, int eival_order);
/* ConvertFlatSym2Evects3<float> (lib/visfd/eigen3_simple.hpp:392-405; decode lib/visfd/lin3_utils.hpp:566-584):
 * one flat symmetric matrix -> eigenvalues and eigenvectors as rows (row-major 3x3), host. */
int visfd_hip_convert_flat_sym2_evects3_host(const float* m6, int eival_order, float* eivals3, float* eivects9);
/* DiagonalizeSym3<float> (lib/visfd/eigen3_simple.hpp:137-266) on the host: m9 = symmetric 3x3, row-major;
 * eivects9 = eigenvectors as rows; order 0..3 = INCREASING, DECREASING, INCREASING_ABS, DECREASING_ABS_EIVALS. */
int visfd_hip_diagonalize_sym3_f32_host(const float* m9, int order, float* eivals3, float* eivects9);
/* The oriented point cloud of a clustered surface (the -normals-file tail of HandleTV,
 * bin/filter_mrc/handlers.cpp:2039-2309), host-side.  voxel2cluster = the label volume as floats (what HandleTV
 * writes to its output image; NULL: export every unmasked voxel unchanged), direction = [nz][ny][nx][3]
 * (standardized), select_cluster = the label to export, curve_ds / find_ridge / max_distance_to_feature = the
 * reference's settings (defaults 0.2, 1, 1.3 voxels; settings.cpp:147-149).  crds/norms receive n_points x 3
 * floats each (capacity points; pass NULL pointers to only count). */
int visfd_hip_surface_points(const float* saliency, const float* voxel2cluster, const float* direction,
                             const float* mask, int64_t nx, int64_t ny, int64_t nz, int select_cluster,
                             const float voxel_width[3], float curve_ds, int find_ridge,
                             float max_distance_to_feature, float* crds, float* norms, int64_t capacity,
                             int64_t* n_points);

/* ---- f4: BinArray3D / UnbinArray3D, lib/visfd/resample.hpp:53-166 -------------------------------------
 * Sizes are {nx, ny, nz}.  bin[d] = floor(size_big[d] / size_small[d]); `offset` (nullable) shifts the
 * binning window and must satisfy 0 <= offset[d] < bin[d] (VISFD_HIP_EINVAL otherwise, where the
 * reference throws; also when the shifted window would leave the source, which the reference only
 * asserts).  Bin: dst = float sum over the bin in z,y,x order / bin volume; source voxels
 * beyond size_dst*bin are dropped.  Unbin: dst[I] = src[clamp((I - offset) / bin)]. */
int visfd_hip_bin_array3d(visfd_hip_ctx*, const float* src, const int64_t size_src[3], float* dst,
                          const int64_t size_dst[3], const int* offset);
int visfd_hip_bin_array3d_dev(visfd_hip_ctx*, const float* src, const int64_t size_src[3], float* dst,
                              const int64_t size_dst[3], const int* offset);
int visfd_hip_unbin_array3d(visfd_hip_ctx*, const float* src, const int64_t size_src[3], float* dst,
                            const int64_t size_dst[3], const int* offset);
int visfd_hip_unbin_array3d_dev(visfd_hip_ctx*, const float* src, const int64_t size_src[3], float* dst,
                                const int64_t size_dst[3], const int* offset);

/* ---- a9: CalcHessian, lib/visfd/feature.hpp:1203-1348 ------------------------------------------- */
/* gradient (nullable): 3 channels; hessian: 6 channels (xx,yy,zz,xy,yz,xz).  Voxels with mask==0
 * are left untouched.  Returns VISFD_HIP_EINVAL if any dimension < 3 (feature.hpp:1260-1264). */
int visfd_hip_calc_hessian(visfd_hip_ctx*, const float* src, float* gradient, float* hessian,
                           const float* mask, int64_t nx, int64_t ny, int64_t nz, float sigma,
                           float truncate_ratio);
int visfd_hip_calc_hessian_dev(visfd_hip_ctx*, const float* src, float* gradient, float* hessian,
                               const float* mask, int64_t nx, int64_t ny, int64_t nz, float sigma,
                               float truncate_ratio);

/* ---- a10: DiagonalizeFlatSym3 (batched), lib/visfd/eigen3_simple.hpp:271-342 -------------------- */
/* n matrices of 6 floats -> n x [lambda0,lambda1,lambda2, shoemake0..2]. Host: interleaved;
 * device: channel-planar with stride n. */
int visfd_hip_diagonalize_flat_sym3(visfd_hip_ctx*, const float* m6, float* out6, int64_t n,
                                    int eival_order);
int visfd_hip_diagonalize_flat_sym3_dev(visfd_hip_ctx*, const float* m6, float* out6, int64_t n,
                                        int eival_order);

/* ---- a12 (first half): ridge saliency + principal direction ------------------------------------- */
/* The per-voxel loop of HandleTV, bin/filter_mrc/handlers.cpp:1640-1746 (SURFACE_RIDGE, no
 * background subtraction): saliency = (l0^2-l1^2)^2 (feature.hpp:1557-1560), direction = first
 * eigenvector after the float Shoemake round trip (eigen3_simple.hpp:392-405).  saliency is zero
 * where mask==0; direction is written only where mask!=0. */
int visfd_hip_hessian_saliency(visfd_hip_ctx*, const float* hessian, const float* mask,
                               int64_t nvox, int eival_order, float* saliency, float* direction);
int visfd_hip_hessian_saliency_dev(visfd_hip_ctx*, const float* hessian, const float* mask,
                                   int64_t nvox, int eival_order, float* saliency,
                                   float* direction);
/* Fused form (no materialised Hessian): Gaussian smoothing + 19-point stencil + eigen + score.
 * Equivalent to calc_hessian followed by hessian_saliency.  Device face only. */
int visfd_hip_ridge_saliency_dev(visfd_hip_ctx*, const float* src, const float* mask,
                                 int64_t nx, int64_t ny, int64_t nz, float sigma,
                                 float truncate_ratio, int eival_order, float* saliency,
                                 float* direction);
/* The same in two steps, for callers that threshold the saliency before they need directions (HandleTV,
 * handlers.cpp:1751-1797): visfd_hip_ridge_scores_dev smooths and scores every voxel and hands back the smoothed
 * volume (`smoothed`: caller's buffer of nvox floats, distinct from src and saliency);
 * visfd_hip_ridge_directions_dev writes the principal direction (3 planes) of the voxels whose saliency is non-zero
 * and leaves the others untouched.  Scores and directions are bit-identical to visfd_hip_ridge_saliency_dev's. */
int visfd_hip_ridge_scores_dev(visfd_hip_ctx*, const float* src, const float* mask, int64_t nx, int64_t ny,
                               int64_t nz, float sigma, float truncate_ratio, int eival_order, float* saliency,
                               float* smoothed);
int visfd_hip_ridge_directions_dev(visfd_hip_ctx*, const float* smoothed, int64_t nx, int64_t ny, int64_t nz,
                                   float sigma, int eival_order, const float* saliency, float* direction);
/* The optional PEAK-HEIGHT factor of the two score loops (`-membrane-background SIGMA_B`, alias `-detection-background`;
 * bin/filter_mrc/settings.cpp:2802-2825): background = ApplyGauss(image, SIGMA_B, floor(SIGMA_B * ratio), mask, normalize)
 * (handlers.cpp:1577-1592), and every score is multiplied by (image - background) in float (handlers.cpp:1698-1702).
 * visfd_hip_peak_background_dev computes the background volume; visfd_hip_ridge_scores_bg_dev is
 * visfd_hip_ridge_scores_dev with the factor applied to every score (background == NULL: no factor). */
int visfd_hip_peak_background_dev(visfd_hip_ctx*, const float* image, const float* mask, int64_t nx, int64_t ny, int64_t nz,
                                  float sigma_background, float truncate_ratio, int normalize, float* background);
int visfd_hip_ridge_scores_bg_dev(visfd_hip_ctx*, const float* src, const float* mask, int64_t nx, int64_t ny,
                                  int64_t nz, float sigma, float truncate_ratio, int eival_order, const float* background,
                                  float* saliency, float* smoothed);

/* ---- a12 (second half): global top-fraction threshold, handlers.cpp:1751-1797 -------------------- */
/* threshold = (floor(n_unmasked*fraction))-th largest unmasked saliency; every voxel with
 * saliency < threshold is zeroed in place.  threshold_out nullable. */
int visfd_hip_threshold_fraction(visfd_hip_ctx*, float* saliency, const float* mask, int64_t nvox,
                                 float fraction, float* threshold_out);
int visfd_hip_threshold_fraction_dev(visfd_hip_ctx*, float* saliency, const float* mask,
                                     int64_t nvox, float fraction, float* threshold_out);
/* building blocks for the multi-GPU (Z-slab) form of the same select: the order-preserving 32-bit
 * key of the float is split into three digits (round 0: bits 31..21, round 1: bits 20..10, round 2:
 * bits 9..0); each call returns this rank's 2048-bin histogram of digit `round` over unmasked
 * voxels whose higher digits equal `prefix` (right-aligned; 0 for round 0).  The caller sums the
 * histograms across ranks, walks them from the top to find the digit of the k-th largest key, and
 * finally applies the threshold with visfd_hip_apply_threshold_dev.
 * visfd_amd/slab.py holds the host logic. */
int visfd_hip_select_histogram_dev(visfd_hip_ctx*, const float* saliency, const float* mask,
                                   int64_t nvox, int round, uint32_t prefix,
                                   uint64_t* hist_host /* 2048 */, uint64_t* n_unmasked_host);
/* the same histogram written to DEVICE memory (2048 counters), asynchronous on the context's stream: multi-GPU callers
 * all-reduce it there (RCCL) and fetch the sum once per round */
int visfd_hip_select_histogram_todev(visfd_hip_ctx*, const float* saliency, const float* mask, int64_t nvox, int round,
                                     uint32_t prefix, uint64_t* hist_dev /* 2048, device */);
int visfd_hip_apply_threshold_dev(visfd_hip_ctx*, float* saliency, int64_t nvox, float threshold);

/* ---- a13+a14: TV3D::TVDenseStick, lib/visfd/feature.hpp:1645-1675,1711-2037,2217-2384 ------------ */
/* As instantiated by HandleTV (handlers.cpp:1821-1836): normalize=false, diagonalize=false.
 * tensor (6 channels) is zeroed where mask_dst!=0 (or everywhere if NULL) and left untouched
 * elsewhere.  detect_curves selects the curve vote field (feature.hpp:2317-2347). */
int visfd_hip_tv_dense_stick(visfd_hip_ctx*, const float* saliency, const float* direction,
                             float* tensor, const float* mask_src, const float* mask_dst,
                             int64_t nx, int64_t ny, int64_t nz, float sigma_tv, int exponent,
                             float cutoff_ratio, int detect_curves);
int visfd_hip_tv_dense_stick_dev(visfd_hip_ctx*, const float* saliency, const float* direction,
                                 float* tensor, const float* mask_src, const float* mask_dst,
                                 int64_t nx, int64_t ny, int64_t nz, float sigma_tv, int exponent,
                                 float cutoff_ratio, int detect_curves);
/* Z-slab form for multi-GPU runs: the arrays cover planes [z_lo, z_lo+nz_local) of a volume whose
 * true height is nz_global; receivers are computed for planes [z_out0, z_out1) (global indices) and
 * senders outside the supplied planes are treated as absent (so the caller must supply
 * halfwidth ghost planes on interior faces). */
int visfd_hip_tv_dense_stick_slab_dev(visfd_hip_ctx*, const float* saliency, const float* direction,
                                      float* tensor, const float* mask_src, const float* mask_dst,
                                      int64_t nx, int64_t ny, int64_t nz_local, int64_t z_out0,
                                      int64_t z_out1, float sigma_tv, int exponent,
                                      float cutoff_ratio, int detect_curves);
/* The denominators of TVDenseStick(normalize=true) with a source mask (feature.hpp:1761-1822, 2376-2382):
 * den[voxel] = sum of w(j) * mask_src(sender) over the votes the voxel receives, added in vote order; voxels with
 * mask_dst == 0 keep the caller's value.  Host pointers.  include/visfd_hip.hpp divides the tensors by it exactly as
 * the reference does. */
int visfd_hip_tv_weight_sum(visfd_hip_ctx*, const float* saliency, float* den, const float* mask_src, const float* mask_dst,
                            int64_t nx, int64_t ny, int64_t nz, float sigma_tv, float cutoff_ratio);
/* halfwidth of the vote window, floor(sigma_tv*cutoff) (feature.hpp:1671); tables (nullable):
 * w[(2h+1)^3], rhat[(2h+1)^3][3] as built by filter3d.hpp:546-601 and feature.hpp:2468-2482. */
int visfd_hip_tv_tables(float sigma_tv, float cutoff_ratio, int* halfwidth_out, float* w,
                        float* rhat);

/* ---- a15: post-voting score, bin/filter_mrc/handlers.cpp:1870-1892 ------------------------------ */
/* saliency[v] = lambda0 - lambda1 of the diagonalised vote tensor where mask != 0; untouched elsewhere */
int visfd_hip_tensor_saliency(visfd_hip_ctx*, const float* tensor, const float* mask, int64_t nvox,
                              int eival_order, float* saliency_inout);
int visfd_hip_tensor_saliency_dev(visfd_hip_ctx*, const float* tensor, const float* mask,
                                  int64_t nvox, int eival_order, float* saliency_inout);
/* with the peak-height factor (handlers.cpp:1883-1887): score *= image - background (both NULL: no factor) */
int visfd_hip_tensor_saliency_bg_dev(visfd_hip_ctx*, const float* tensor, const float* mask, int64_t nvox, int eival_order,
                                     const float* image, const float* background, float* saliency_inout);

/* ---- a9+a10+a11+a12+a14+a15 in one call: the compute section of HandleTV ------------------------- */
/* bin/filter_mrc/handlers.cpp:1618-1892 for SURFACE_RIDGE (the _bg forms add the optional peak-height factor of
 * handlers.cpp:1577-1605,1698-1702,1883-1887: sigma_background > 0 multiplies both scores by image - background):
 * CalcHessian (feature.hpp:1203) -> per-voxel eigen/score/direction loop (handlers.cpp:1645-1746) ->
 * saliency threshold (handlers.cpp:1751-1797: top `best_fraction` of unmasked voxels when
 * best_fraction >= 0, else the absolute value `threshold_abs`) -> TV3D::TVDenseStick when sigma_tv > 0
 * (handlers.cpp:1821-1836, normalize=false) -> post-voting score (handlers.cpp:1870-1892).
 * saliency_out: 1 channel (zero where mask == 0 before voting; voxels with mask == 0 keep that zero).
 * tensor_out (nullable): 6 channels, interleaved on the host face / planar on the device face; this is
 * what `-save-progress` writes as <base>_tensor_{0..5}.rec (handlers.cpp:1897-1922).
 * direction_out (nullable): principal direction, 3 channels. threshold_out (nullable). */
int visfd_hip_membrane_detect(visfd_hip_ctx*, const float* src, const float* mask,
                              int64_t nx, int64_t ny, int64_t nz, float sigma, float truncate_ratio,
                              int eival_order, float best_fraction, float threshold_abs, float sigma_tv,
                              int tv_exponent, float tv_cutoff_ratio, float* saliency_out,
                              float* tensor_out, float* direction_out, float* threshold_out);
int visfd_hip_membrane_detect_dev(visfd_hip_ctx*, const float* src, const float* mask,
                                  int64_t nx, int64_t ny, int64_t nz, float sigma, float truncate_ratio,
                                  int eival_order, float best_fraction, float threshold_abs,
                                  float sigma_tv, int tv_exponent, float tv_cutoff_ratio,
                                  float* saliency_out, float* tensor_out, float* direction_out,
                                  float* threshold_out);
int visfd_hip_membrane_detect_bg(visfd_hip_ctx*, const float* src, const float* mask,
                                 int64_t nx, int64_t ny, int64_t nz, float sigma, float truncate_ratio,
                                 int eival_order, float best_fraction, float threshold_abs, float sigma_tv,
                                 int tv_exponent, float tv_cutoff_ratio, float sigma_background, int normalize_background,
                                 float* saliency_out, float* tensor_out, float* direction_out, float* threshold_out);
int visfd_hip_membrane_detect_bg_dev(visfd_hip_ctx*, const float* src, const float* mask,
                                     int64_t nx, int64_t ny, int64_t nz, float sigma, float truncate_ratio,
                                     int eival_order, float best_fraction, float threshold_abs,
                                     float sigma_tv, int tv_exponent, float tv_cutoff_ratio, float sigma_background,
                                     int normalize_background, float* saliency_out, float* tensor_out, float* direction_out,
                                     float* threshold_out);

/* ---- Z-slab helpers for the separable filter (multi-GPU) -------------------------- */
/* Same as apply_gauss_dev on a slab: arrays hold planes [z_lo, z_lo+nz_local) of a volume of height
 * nz_global; planes outside the slab are treated as outside the image ONLY at the true faces
 * (z_lo==0 / z_lo+nz_local==nz_global); the unmasked normaliser uses global coordinates
 * (filter3d.hpp:1004-1021).  Output planes within halfwidth[2] of an interior slab face are
 * invalid and must be discarded by the caller (they are ghost planes). */
int visfd_hip_apply_gauss_slab_dev(visfd_hip_ctx*, const float* src, float* dst,
                                   int64_t nx, int64_t ny, int64_t nz_local, int64_t z_lo,
                                   int64_t nz_global, const float sigma[3], const int halfwidth[3],
                                   int normalize, float* A_out);

/* ---- e: Z-slab runs across the GPUs of one node ------------------------------------------------------
 * The reference is single-process (OpenMP only): there is no reference interface to cite here.  These entry points are
 * what a host started once per GPU calls so that a volume larger than one GPU's HBM -- or simply more throughput -- runs on
 * the GPUs of a node: planes [z0, z1) of the volume per rank plus `ghost` planes on each INTERIOR face, halos exchanged
 * point-to-point with the two Z-neighbours only (RCCL send/recv over one xGMI link each, on a transfer stream of the
 * slab's own, overlapping the votes of the interior planes), three all-reduces of 2048 counters for the exact global
 * top-fraction threshold.  Outputs of owned planes are bit-identical to the single-volume run (exact kernels).
 * Volumes are DEVICE pointers of the local shape [nz_local][ny][nx]; everything is queued on the context's stream. */
typedef struct visfd_hip_slab visfd_hip_slab;
/* A transport other than RCCL (tests; MPI hosts).  Pointers are DEVICE pointers; work must be ordered on `stream`
 * (a hipStream_t) or complete on return.  Every callback returns 0 on success. */
typedef struct visfd_hip_transport {
  /* send `bytes` from sendbuf to rank `peer` and receive as many from it into recvbuf (a paired exchange) */
  int (*sendrecv)(void* user, int peer, const void* sendbuf, void* recvbuf, size_t bytes, void* stream);
  /* sum `count` uint64 counters over all ranks, in place */
  int (*allreduce_sum_u64)(void* user, uint64_t* buf, size_t count, void* stream);
  int (*group_start)(void* user);   /* optional (may be NULL): brackets the sendrecv calls of one halo exchange */
  int (*group_end)(void* user);
  void* user;
} visfd_hip_transport;
/* RCCL: rank 0 obtains a 128-byte id (ncclGetUniqueId), the host hands it to every rank by its own means (a file, MPI,
 * torch.distributed ...), every rank creates its slab with it (ncclCommInitRank).  librccl.so is loaded at run time. */
int visfd_hip_slab_unique_id(void* id_out_128_bytes);
/* 1 if librccl.so can be loaded in this process (no error is set otherwise): lets every rank agree BEFORE any of them
 * enters ncclCommInitRank, which would block for ever if one rank could not follow */
int visfd_hip_slab_rccl_available(void);
int visfd_hip_slab_create_rccl(visfd_hip_ctx*, const void* unique_id_128_bytes /* may be NULL when world == 1 */, int rank,
                               int world, int64_t nz_global, int ghost, visfd_hip_slab** out);
int visfd_hip_slab_create_custom(visfd_hip_ctx*, const visfd_hip_transport*, int rank, int world, int64_t nz_global,
                                 int ghost, visfd_hip_slab** out);
int visfd_hip_slab_destroy(visfd_hip_slab*);
/* out = {z0, z1, lo, hi, own0, own1, nz_local}: owned planes [z0, z1) and stored planes [lo, hi) of the volume; the owned
 * planes are [own0, own1) of the local array */
int visfd_hip_slab_layout(visfd_hip_slab*, int64_t out[7]);
/* The transport's smoke test: a grouped send/receive of `count` floats from this rank to itself on the transfer stream and
 * an all-reduce of 2048 counters, both verified.  A slab created with world == 1 AND an id owns a one-rank RCCL communicator:
 * that is what a one-GPU box can run of the RCCL path (run-time loading, call signatures, stream ordering). */
int visfd_hip_slab_selftest(visfd_hip_slab*, int64_t count);
/* workgroup slots the voting grid leaves free for the transport's kernels while a halo is in flight (default 64) */
int visfd_hip_slab_set_reserve(visfd_hip_slab*, int reserve_workgroups);
/* fill the ghost planes of `nvol` volumes within `depth` planes of the owned range (one group of sends/receives) */
int visfd_hip_slab_exchange_dev(visfd_hip_slab*, float* const* volumes, int nvol, int64_t nx, int64_t ny, int depth);
/* HandleTV (bin/filter_mrc/handlers.cpp:1501-1892) on one slab: ridge scores, GLOBAL top-fraction threshold, directions,
 * halo of (saliency, direction), votes (interior planes beside the transfer), post-vote score.  dirs: 3 planar channels,
 * tensor: 6, scratch: one volume.  Valid results: the owned planes of sal and tensor.  src_halo_ready != 0: the caller has
 * already exchanged src's ghost planes at least floor(sigma * ratio) + 1 deep in this step. */
int visfd_hip_membrane_detect_slab_dev(visfd_hip_slab*, float* src, float* sal, float* dirs, float* tensor, float* scratch,
                                       int64_t nx, int64_t ny, float sigma, float truncate_ratio, int eival_order,
                                       float best_fraction, float sigma_tv, int exponent, float tv_truncate_ratio,
                                       int src_halo_ready, float* threshold_out);
/* The same for a host whose volume lives in HOST memory (filter_mrc started once per GPU, `-slab`): src_owned / sal_owned
 * are the rank's OWNED planes [z0, z1) only ([z1-z0][ny][nx]); tensor_owned (may be NULL) receives six interleaved floats per
 * owned voxel, as visfd_hip_membrane_detect returns them.  Device arrays (12 slab volumes) live for the call only. */
int visfd_hip_membrane_detect_slab(visfd_hip_slab*, const float* src_owned, int64_t nx, int64_t ny, float sigma,
                                   float truncate_ratio, int eival_order, float best_fraction, float sigma_tv, int exponent,
                                   float tv_truncate_ratio, float* sal_owned, float* tensor_owned, float* threshold_out);
/* The slab forms with the peak-height factor (sigma_background > 0; `background`: one more slab volume on the device face).
 * The ghost depth must cover floor(sigma_background * truncate_ratio) as well. */
int visfd_hip_membrane_detect_slab_bg_dev(visfd_hip_slab*, float* src, float* sal, float* dirs, float* tensor, float* scratch,
                                          float* background, int64_t nx, int64_t ny, float sigma, float truncate_ratio,
                                          int eival_order, float best_fraction, float sigma_tv, int exponent,
                                          float tv_truncate_ratio, float sigma_background, int normalize_background,
                                          int src_halo_ready, float* threshold_out);
int visfd_hip_membrane_detect_slab_bg(visfd_hip_slab*, const float* src_owned, int64_t nx, int64_t ny, float sigma,
                                      float truncate_ratio, int eival_order, float best_fraction, float sigma_tv, int exponent,
                                      float tv_truncate_ratio, float sigma_background, int normalize_background,
                                      float* sal_owned, float* tensor_owned, float* threshold_out);
/* BlobDog (lib/visfd/feature.hpp:53-427) on one slab with absolute thresholds: blobs of the OWNED planes only, iz as
 * GLOBAL plane index.  The host merges the ranks' lists (and applies ratio thresholds, which need the global best). */
int visfd_hip_blob_dog_slab_dev(visfd_hip_slab*, float* src, int64_t nx, int64_t ny, const float* blob_sigma, int n_sigma,
                                float delta_sigma_over_sigma, float truncate_ratio, float minima_threshold,
                                float maxima_threshold, int src_halo_ready,
                                visfd_hip_blob* minima, int64_t minima_capacity, int64_t* n_minima,
                                visfd_hip_blob* maxima, int64_t maxima_capacity, int64_t* n_maxima);
/* Host-memory faces for a host that runs one process per GPU (`filter_mrc -gauss|-blob ... -slab`): the rank's OWNED planes
 * [z1-z0][ny][nx] in, the owned planes of the filtered volume / the blobs of the owned planes (iz GLOBAL) out.  A blob list that
 * does not fit returns VISFD_HIP_ECAPACITY with the needed counts in n_minima / n_maxima; the retry is local. */
int visfd_hip_apply_gauss_slab(visfd_hip_slab*, const float* src_owned, int64_t nx, int64_t ny, const float sigma[3],
                               const int halfwidth[3], int normalize, float* dst_owned, float* A_out);
int visfd_hip_blob_dog_slab(visfd_hip_slab*, const float* src_owned, int64_t nx, int64_t ny, const float* blob_sigma, int n_sigma,
                            float delta_sigma_over_sigma, float truncate_ratio, float minima_threshold, float maxima_threshold,
                            visfd_hip_blob* minima, int64_t minima_capacity, int64_t* n_minima,
                            visfd_hip_blob* maxima, int64_t maxima_capacity, int64_t* n_maxima);

#ifdef __cplusplus
}
#endif
#endif /* VISFD_HIP_H */
