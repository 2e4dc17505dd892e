/*
 * lslam.h — C ABI of the MI355X (gfx950) ICP scan-matching / occupancy-grid engine.
 *
 * The reference (fadli0029/LiDAR-SLAM) is pure Python and has no FFI layer; its
 * boundary for this path is the function surface of modules/icp.py and
 * modules/ogm.py.  Every entry point below names the reference interface it
 * replaces (paths relative to the reference root).  The binding a maintainer of
 * the reference would add is a ctypes.CDLL stub — see INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross this boundary;
 *   - every function returns LSLAM_OK (0) or a negative LSLAM_ERR_* code and
 *     records a message retrievable with lslam_last_error() (thread-local);
 *   - point clouds are float32, row-major (n, dim) with dim = 2 (planar scans,
 *     the reference pads z = 0: modules/icp.py:153-156) or 3;
 *   - transforms are float64 4x4 row-major, exactly numpy's C layout of the
 *     reference's matrices;
 *   - "_dev" functions take DEVICE pointers (hipMalloc / torch.Tensor.data_ptr())
 *     plus a hipStream_t passed as void*; they enqueue work and return without
 *     synchronising.  The others take HOST pointers, use library-owned device
 *     scratch and return when the results are in the caller's buffers;
 *   - the caller owns every input and output buffer; inputs are never written.
 *
 * There is no CPU fallback: on a machine without a HIP device every compute
 * entry point fails with LSLAM_ERR_HIP.
 */
#ifndef LSLAM_H
#define LSLAM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSLAM_OK                 0
#define LSLAM_ERR_BAD_ARG      (-1)  /* NULL pointer, bad dim, negative size, non-monotone offsets ...      */
#define LSLAM_ERR_EMPTY_TARGET (-2)  /* reference: IndexError from pc2[correspondences] (icp.py:118)        */
#define LSLAM_ERR_NONFINITE    (-3)  /* reference: ValueError "'x' must be finite" from KDTree (icp.py:161) */
#define LSLAM_ERR_HIP          (-4)  /* any HIP runtime failure, including "no device"; also: the pruned batch
                                       * kernel's waves gave up waiting for work (a can't-happen guard: no result
                                       * is returned rather than an incomplete one)                              */

#define LSLAM_ABI_VERSION 1

/* ---- housekeeping ------------------------------------------------------ */
int         lslam_abi_version(void);
const char *lslam_last_error(void);
/* number of visible HIP devices, 0 if none, never fails */
int         lslam_device_count(void);
/* bind the calling host thread to a device (one host thread / process per GPU) */
int         lslam_set_device(int device);

/* Pinned host memory owned by the library, one buffer per (calling thread, slot 0|1), grown on demand and valid until
 * the same thread asks for the same slot again.  Packing a scan list into it before lslam_icp_batch* / lslam_ogm_update
 * makes the upload one DMA at PCIe speed (no reference counterpart: the reference never leaves the host). */
int         lslam_host_staging(int slot, size_t bytes, void **out_ptr);

/* Device and pinned memory the library keeps per calling thread between calls (scratch arena, the staging buffers above,
 * the spatially ordered copies of the pruned batch kernels, the workspaces of the single-pair paths, the thread's stream):
 * they only grow, each growth behind a hipDeviceSynchronize().  A long-lived service returns them with this call (it waits
 * for the device, frees, and the thread's next call allocates afresh).  No reference counterpart. */
int         lslam_release_thread_resources(void);

/* Measurement aid (no reference counterpart): the float32 VALU issue rate this device sustains
 * under an all-VALU load, in lane-operations per second (≈ 10 ms of v_fma_f32 on every SIMD). */
int         lslam_valu_probe(int device, double *out_lane_ops_per_s);

/* ---- ICP: modules/icp.py:123-189 run_icp, batched ---------------------- */
/*
 * S independent point-to-point ICP problems in one launch (one workgroup per
 * problem, on-chip until converged).  Problem s aligns the source points
 * src[src_off[s] .. src_off[s+1]) to the target points tgt[tgt_off[s] .. tgt_off[s+1]).
 * (Offsets count POINTS, not floats.)  src and tgt may alias the same buffer:
 * for a scan sequence pass the scans once and src_off = off + 1, tgt_off = off
 * (source = scan i, target = scan i-1: modules/localization.py:120-123).
 *
 * Per problem this is run_icp(pc1, pc2, init_transform, epsilon, max_iters,
 * stopping_thresh, normalize_error=...) with the reference's stop rules in the
 * reference's order (icp.py:172-181) and its "error of the pre-update residual"
 * convention (icp.py:116-120).
 *
 *   init_T     S x 16 float64, row-major 4x4 per problem (icp.py:125)
 *   out_T      S x 16 float64
 *   out_err    S float64    error measured in the last executed iteration (icp.py:120)
 *   out_iters  S int32      number of icp_iteration calls executed
 *   out_corr   NULL, or int32 per source point laid out like src_off: index (relative to
 *              the problem's target) matched in the LAST iteration (icp.py:117, 187-189)
 *   flags      LSLAM_ICP_* bits
 */
#define LSLAM_ICP_NORMALIZE_ERROR 1u   /* icp.py:91-97 */
/* How the nearest neighbours are searched.  Both strategies return the same correspondences, iteration
 * counts and (to float64 summation order) transforms:
 *   all pairs   every source against every target out of LDS (the big batch build sweeps the expanded form
 *               |t|^2 - 2 a.t first and re-evaluates, with the exact recipe, only the chunks its rounding bound
 *               cannot exclude);
 *   pruning     both clouds are put into a spatial order once per problem (Hilbert curve of a grid; a cloud that arrives in a
 *               compact order, e.g. LiDAR beam order, is kept as it is) and every group of target points whose
 *               bounding box cannot hold a nearer (or equally near) point for any source of a wave is skipped,
 *               with a float32-safe bound.  Batched entry points: planar problems of at most 4096 points per
 *               cloud, 3-D problems of at most 4096 target / 8191 source points; lslam_icp_large*: any size.
 * With neither bit set the library picks: pruning where it is supported and measured faster (batched planar
 * problems from 256 points per cloud, batched 3-D problems from 256, one large pair from 2.5*10^8 (planar) / 10^9 (3-D) pair evaluations per
 * iteration; measured with tools/prune_threshold.py), all pairs elsewhere.
 * The environment variable LSLAM_ICP_PRUNE (0 / 1) changes that default; an explicit bit wins over it. */
#define LSLAM_ICP_PRUNE 2u       /* force pruning (ignored where unsupported) */
#define LSLAM_ICP_ALL_PAIRS 4u   /* force the all-pairs sweep */
/* host entry points only: the caller has already verified that every coordinate is finite (e.g. while packing the
 * scans), so the library skips its own pass over the points.  Non-finite points under this flag are undefined. */
#define LSLAM_ICP_INPUT_CHECKED 8u

int lslam_icp_batch(const float *src, const int64_t *src_off,
                    const float *tgt, const int64_t *tgt_off,
                    int dim, const double *init_T, int32_t S,
                    double epsilon, double stopping_thresh, int32_t max_iters,
                    uint32_t flags,
                    double *out_T, double *out_err, int32_t *out_iters,
                    int32_t *out_corr, int device);

/* Same, all pointers on the device; max_src_n / max_tgt_n bound the per-problem
 * sizes (they fix the launch geometry; the offsets are only read on the device).
 * Inputs are trusted to be finite and every target non-empty (the host entry
 * point validates; this one cannot without a synchronisation).
 * Enqueues on `stream` and returns — with two exceptions a caller overlapping streams must know: (1) the pruned
 * kernels keep spatially ordered copies in a workspace of the calling thread; the first call at a larger size (more
 * problems x points) re-allocates it behind a hipDeviceSynchronize(), and a call on a different stream than the
 * thread's previous workspace-using call waits (on the device) for that call; (2) the all-pairs sweep
 * (LSLAM_ICP_ALL_PAIRS, or the automatic choice for small clouds) uses no workspace and is never ordered or
 * synchronised by the library.  Not capturable into a hipGraph while a workspace may still grow. */
int lslam_icp_batch_dev(const float *d_src, const int64_t *d_src_off,
                        const float *d_tgt, const int64_t *d_tgt_off,
                        int dim, const double *d_init_T, int32_t S,
                        int64_t max_src_n, int64_t max_tgt_n,
                        double epsilon, double stopping_thresh, int32_t max_iters,
                        uint32_t flags,
                        double *d_out_T, double *d_out_err, int32_t *d_out_iters,
                        int32_t *d_out_corr, void *stream);

/* ---- float64-exact mode ---------------------------------------------------
 * The reference searches its nearest neighbours in float64 (scipy KDTree on float64 points,
 * modules/icp.py:29-41,161).  The entry points above take float32 points: their correspondences are those
 * of an exact float32 search, which differs from the reference's wherever rounding the points to float32
 * flips a near tie.  The entry points below take the float64 clouds as well.  The float32 search then only
 * proposes candidates: every target whose float32 distance lies within a proven bound of the winner's
 *  is re-evaluated as |T p - q|^2 in float64 on the float64 points, lowest index on exact
 * float64 ties, and the moments / error are accumulated from the float64 points.  Correspondences are then
 * those of modules/icp.py:29-41 (up to ties the KD-tree breaks arbitrarily) and T, error agree with the
 * reference to float64 summation order.  Same arguments otherwise; same search strategies.
 * lslam_icp_batch_f64: host float64 clouds (uploaded once, rounded to float32 on the device).
 * lslam_icp_batch_dev64: device pointers to BOTH copies, laid out alike (d_src64 / d_tgt64 may both be NULL:
 * then it is lslam_icp_batch_dev).  The float32 copy must be the float64 one rounded to nearest, coordinate by
 * coordinate — the near-tie bound is the bound of that rounding, and the kernels search with whichever of the
 * two is at hand. */
int lslam_icp_batch_f64(const double *src, const int64_t *src_off,
                        const double *tgt, const int64_t *tgt_off,
                        int dim, const double *init_T, int32_t S,
                        double epsilon, double stopping_thresh, int32_t max_iters,
                        uint32_t flags,
                        double *out_T, double *out_err, int32_t *out_iters,
                        int32_t *out_corr, int device);
int lslam_icp_batch_dev64(const float *d_src, const double *d_src64, const int64_t *d_src_off,
                          const float *d_tgt, const double *d_tgt64, const int64_t *d_tgt_off,
                          int dim, const double *d_init_T, int32_t S,
                          int64_t max_src_n, int64_t max_tgt_n,
                          double epsilon, double stopping_thresh, int32_t max_iters,
                          uint32_t flags,
                          double *d_out_T, double *d_out_err, int32_t *d_out_iters,
                          int32_t *d_out_corr, void *stream);

/* ---- ICP on ONE large cloud pair (whole chip on one problem) ------------ */
/*
 * run_icp for clouds too large for one workgroup (config "100k x 100k"): per
 * iteration a grid-wide LDS-tiled nearest-neighbour pass, a moment reduction
 * and a device-side solve; convergence is decided on the device, the host only
 * polls a flag every `poll_every` iterations.  d_src/d_tgt: device pointers.
 * init_T / out_T / out_err / out_iters: HOST pointers.  d_out_corr: device, or NULL.
 */
int lslam_icp_large_dev(const float *d_src, int64_t n, const float *d_tgt, int64_t m,
                        int dim, const double *init_T,
                        double epsilon, double stopping_thresh, int32_t max_iters,
                        uint32_t flags, int32_t poll_every,
                        double *out_T, double *out_err, int32_t *out_iters,
                        int32_t *d_out_corr, void *stream);

/* LiDAR-sized single pairs (below 8192 source points, target within one LDS tile) run the WHOLE iteration loop inside one
 * launch of co-resident workgroups that hand their partial moments to each other through memory.  That needs every
 * workgroup on a CU of its own at the same time: the library takes this path only when the grid is at most half of the
 * device's CUs (LSLAM_ICP_COOP_MAX_CUS in the environment caps the CU count assumed; LSLAM_ICP_COOP=0 disables the path), and
 * when a workgroup still waits longer than 50 ms for the others (CUs held by other streams or processes) the launch drains,
 * and the SAME call runs the problem again from init_T on the launch-per-iteration path — same result, never an error.
 * lslam_icp_coop_fallbacks(): how many calls of this thread took that detour. */
int lslam_icp_coop_fallbacks(void);

/* float64-exact mode of the two entry points above (see lslam_icp_batch_f64): the float64 clouds decide near
 * ties and feed the moments.  Both search strategies are available (the pruned search adds one refine launch per
 * iteration for the sources with a near tie). */
int lslam_icp_large_f64(const double *src, int64_t n, const double *tgt, int64_t m, int dim,
                        const double *init_T, double epsilon, double stopping_thresh,
                        int32_t max_iters, uint32_t flags,
                        double *out_T, double *out_err, int32_t *out_iters,
                        int32_t *out_corr, int device);
int lslam_icp_large_dev64(const float *d_src, const double *d_src64, int64_t n,
                          const float *d_tgt, const double *d_tgt64, int64_t m, int dim,
                          const double *init_T, double epsilon, double stopping_thresh,
                          int32_t max_iters, uint32_t flags, int32_t poll_every,
                          double *out_T, double *out_err, int32_t *out_iters,
                          int32_t *d_out_corr, void *stream);

/* Same with HOST clouds (float32, validated: finite, non-empty target) and HOST out_corr
 * (n int32 or NULL); copies through library-owned device scratch. */
int lslam_icp_large(const float *src, int64_t n, const float *tgt, int64_t m,
                    int dim, const double *init_T,
                    double epsilon, double stopping_thresh, int32_t max_iters,
                    uint32_t flags,
                    double *out_T, double *out_err, int32_t *out_iters,
                    int32_t *out_corr, int device);

/* One pair, S start transforms: the reference's multi-start registration loop (code/icp_warm_up/test_icp.py:70-88: 24 yaw
 * starts, one run_icp each) for clouds beyond the batched kernels.  Host clouds are validated and uploaded ONCE; the S runs
 * (init_T: S x 16, outputs S x 16 / S / S, all host) follow each other on the device-resident pair. */
int lslam_icp_large_starts(const float *src, int64_t n, const float *tgt, int64_t m, int dim,
                           const double *init_T, int32_t S,
                           double epsilon, double stopping_thresh, int32_t max_iters, uint32_t flags,
                           double *out_T, double *out_err, int32_t *out_iters, int device);
int lslam_icp_large_starts_f64(const double *src, int64_t n, const double *tgt, int64_t m, int dim,
                               const double *init_T, int32_t S,
                               double epsilon, double stopping_thresh, int32_t max_iters, uint32_t flags,
                               double *out_T, double *out_err, int32_t *out_iters, int device);

/* ---- single operators of modules/icp.py -------------------------------- */
/* get_correspondences (icp.py:29-41) after the homogeneous transform of icp.py:116:
 * for every source point the index of the nearest target point (exact float32
 * direct-difference distance, lowest index on ties).  T: HOST 4x4 float64 or NULL
 * (identity).  d_out_idx: n int32; d_out_d2: n float32 squared distances or NULL. */
int lslam_nn_dev(const float *d_src, int64_t n, const float *d_tgt, int64_t m, int dim,
                 const double *T, int32_t *d_out_idx, float *d_out_d2, void *stream);

/* Same with HOST pointers (validated like lslam_icp_large). */
int lslam_nn(const float *src, int64_t n, const float *tgt, int64_t m, int dim,
             const double *T, int32_t *out_idx, float *out_d2, int device);

/* float64-exact mode of lslam_nn (see lslam_icp_batch_f64): the index a float64 search returns — modules/icp.py:29-41 on the
 * reference's own float64 points — lowest index on exact float64 ties. */
int lslam_nn_f64(const double *src, int64_t n, const double *tgt, int64_t m, int dim,
                 const double *T, int32_t *out_idx, int device);

/* get_transform (icp.py:43-74): Kabsch rigid transform taking rows of p onto rows
 * of q.  p, q: HOST float64, n rows, row strides in doubles (>= 3; the reference
 * passes (N,4) homogeneous arrays and uses [:, 0:3]).  out_T: 16 float64. */
int lslam_kabsch(const double *p, int64_t p_stride, const double *q, int64_t q_stride,
                 int64_t n, double *out_T, int device);

/* get_error (icp.py:76-98): sum of squared distances of paired rows; with normalize
 * != 0 divided by (joint bounding-box diagonal)^2 * n.  p, q: HOST float64, n x cols
 * contiguous (cols = 3 in the reference). */
int lslam_sq_error(const double *p, const double *q, int64_t n, int32_t cols,
                   int32_t normalize, double *out_err, int device);

/* ---- voxel_downsample: modules/icp.py:4-27 --------------------------------------------- */
/* Mean of the points in every occupied voxel of size `voxel_size`, voxels in lexicographic order of
 * their integer index floor((p - min) / voxel_size).  points: HOST float64 n x dim (dim = 2 or 3);
 * out: capacity n x dim; *out_count receives the number of voxels.  At most 2^21 voxels per axis. */
int lslam_voxel_downsample(const double *points, int64_t n, int32_t dim, double voxel_size,
                           double *out, int64_t *out_count, int device);

/* ---- pose chain: modules/localization.py:111-130 ------------------------------------ */
/* Sequential float64 product of S relative 3x3 transforms starting at the identity, reported
 * as (x, y, atan2(T10, T00)) after every step: out_poses is (S+1) x 3 with row 0 = (0,0,0).
 * Pure host arithmetic (the chain is inherently sequential and O(S)); no GPU involved. */
int lslam_chain_se2(const double *rel_T3, int64_t S, double *out_poses);

/* ---- LiDAR pre-processing: modules/localization.py:201-307 (main.py:84-87) ---------- */
/*
 * The chain in front of the ICP path, kept on the device between its stages:
 *   stage LSLAM_PRE_POLAR   get_lidar_data (localization.py:252-307): ranges (K x B float64) ->
 *                           robot-frame points; beams with range outside [range_min, range_max] are
 *                           dropped; x = r*cos_a[b] + off_x, y = r*sin_a[b] + off_y (the caller passes
 *                           the beam-angle cos/sin table, B float64 each, and the lidar offset).
 *   stage LSLAM_PRE_DBSCAN  DBSCAN_outliers_removal (localization.py:201-221): drops the points
 *                           sklearn's DBSCAN(eps, min_samples) labels -1.
 *   stage LSLAM_PRE_RADIAL  statistical_outliers_removal (localization.py:223-250): drops points whose
 *                           range is >= mean + k_std*std of the ranges of ALL (remaining) points.
 * Input: `ranges` when LSLAM_PRE_POLAR is set, else the CSR point set (in_xy, in_off).  Output: CSR
 * (out_xy capacity = number of input points, out_off K+1), point and scan order preserved.
 * out_stats (3 float64: mean, std, threshold of the radial stage) may be NULL.  HOST pointers.
 */
#define LSLAM_PRE_POLAR  1u
#define LSLAM_PRE_DBSCAN 2u
#define LSLAM_PRE_RADIAL 4u
int lslam_lidar_preprocess(uint32_t stages,
                           const double *ranges, int32_t K, int32_t B,
                           const double *cos_a, const double *sin_a, double off_x, double off_y,
                           double range_min, double range_max,
                           const double *in_xy, const int64_t *in_off,
                           double eps, int32_t min_samples, double k_std,
                           double *out_xy, int64_t *out_off, double *out_stats, int device);

/* ---- device-resident scan sets ---------------------------------------------------------- */
/*
 * The reference hands the scans from stage to stage as a Python list (main.py:84-87 -> :120 -> :205).
 * A scan set keeps them in HBM instead: created from raw ranges or from points (optionally through the
 * pre-processing stages above), then consumed by scan matching and by the occupancy grid without
 * returning to the host.  It holds the points as float64 (ray casting) and float32 (ICP) plus offsets.
 */
int lslam_scans_create(uint32_t stages,
                       const double *ranges, int32_t K, int32_t B,
                       const double *cos_a, const double *sin_a, double off_x, double off_y,
                       double range_min, double range_max,
                       const double *in_xy, const int64_t *in_off,
                       double eps, int32_t min_samples, double k_std,
                       double *out_stats, int device, void **out_handle);
int lslam_scans_destroy(void *scans);
/* K scans, total points, largest scan */
int lslam_scans_info(void *scans, int64_t *K, int64_t *total_points, int64_t *max_points);
/* copy to the host: out_xy (total x 2 float64, may be NULL), out_off (K+1) */
int lslam_scans_download(void *scans, double *out_xy, int64_t *out_off);
/* poses_from_scan_matching's ICP batch (localization.py:115-124) on the resident scans: problem s aligns
 * scan s+1 (source) to scan s (target); init_T / out_*: HOST, S = K-1 problems. */
int lslam_scans_match_sequence(void *scans, const double *init_T,
                               double epsilon, double stopping_thresh, int32_t max_iters, uint32_t flags,
                               double *out_T, double *out_err, int32_t *out_iters);

/* ---- occupancy grid: modules/ogm.py ------------------------------------- */
/* OccupancyGridMap state (ogm.py:37-43) resident in HBM: the log-odds grid
 * (width x height, indexed [x_cell][y_cell], ogm.py:41) as a float32 base plus an
 * int32 count of log-odds steps on it — a scan adds its (hits - misses) to the
 * count, the clip of ogm.py:188 (or a write) moves the base, and a read returns
 * f32(base + count * step): the recurrence in exact arithmetic, one rounding —
 * plus per-scan hit/miss scratch planes (allocated on the first update). */
int lslam_ogm_create(double res, double world_min_x, double world_min_y,
                     int32_t width, int32_t height, double logodds_step, double logodds_clip,
                     int device, void **out_handle);
int lslam_ogm_destroy(void *handle);
/* the reference keeps the step in a public attribute (ogm.py:43 logodds_ratio); both must be > 0 */
int lslam_ogm_set_params(void *handle, double logodds_step, double logodds_clip);

/* update_map (ogm.py:149-188) for K scans applied IN ORDER (the per-scan clip makes
 * the order matter).  poses: K x 3 float64 (x, y, yaw); scan k is the (n_k, 2)
 * float64 robot-frame points scans_xy[scan_off[k] .. scan_off[k+1]).  HOST pointers. */
int lslam_ogm_update(void *handle, const double *poses, const double *scans_xy,
                     const int64_t *scan_off, int32_t K);

/* Same with the scans of a device-resident scan set (poses: K x 3, HOST). */
int lslam_ogm_update_scans(void *handle, void *scans, const double *poses);

/* The integer (hits - misses) grid ONE scan would add, without touching the
 * log-odds: the bit-exact contract for cell indices (ogm.py:169-185).  out: width*height int32. */
int lslam_ogm_scan_counts(void *handle, const double *pose, const double *scan_xy,
                          int32_t n, int32_t *out_counts);

int lslam_ogm_read_logodds(void *handle, float *out);         /* width*height float32 */
int lslam_ogm_write_logodds(void *handle, const float *in);
/* build_map's binarisation (ogm.py:61-64): 1 where 1/(1+exp(l)) > 0.5 else 0 */
int lslam_ogm_binarize(void *handle, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif /* LSLAM_H */
