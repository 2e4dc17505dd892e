// lslam_kernels.h — host-visible launch interface of the HIP kernels (internal to the library).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lslam_solve.h"

namespace lslam {

constexpr int kIcpBatchMaxThreads = 1024;
constexpr int kIcpBatchMaxWaves = kIcpBatchMaxThreads / 64;
constexpr int kIcpTileMaxPlanar = 8192;   // points per LDS tile (64 KiB as float2)
constexpr int kIcpTileMax3D = 4096;       // 64 KiB as float4
constexpr uint32_t kFlagNormalize = 1u;   // == LSLAM_ICP_NORMALIZE_ERROR

// targets examined per (min, chunk-id) record; the fix-up pass re-evaluates one chunk per source
template <int DIM> struct IcpChunk;
template <> struct IcpChunk<2> { static constexpr int value = 16; };   // 32 (two loads per record step) measured slower: the 32-point fix-up spills
template <> struct IcpChunk<3> { static constexpr int value = 8; };
// the 64-VGPR build for LiDAR-sized problems keeps fewer target points in registers
template <int DIM> struct IcpChunkSmall;
template <> struct IcpChunkSmall<2> { static constexpr int value = 8; };
template <> struct IcpChunkSmall<3> { static constexpr int value = 4; };
constexpr int kIcpSmallMaxThreads = 1024;   // what the 64-VGPR build is compiled for
constexpr int kIcpSmallAutoThreads = 640;   // automatic choice: problems of up to ~1300 source points
constexpr int kIcpSmallMaxTargets = 4096;

struct IcpBatchArgs {
  const float* src;          // device, DIM floats per point
  const int64_t* src_off;    // device, S+1 point offsets
  const float* tgt;
  const int64_t* tgt_off;
  const double* init_T;      // device, S x 16
  double* out_T;             // device, S x 16
  double* out_err;           // device, S
  int32_t* out_iters;        // device, S
  int32_t* out_corr;         // device or nullptr, laid out like src_off
  double epsilon;
  double stopping_thresh;
  int32_t max_iters;
  uint32_t flags;
  int32_t tile_points;       // filled by the launcher
  uint32_t tile_bytes;       // filled by the launcher
  uint32_t src_bytes;        // filled by the launcher: LDS reserved for the source cloud (0 = read it from memory)
  uint32_t w_bytes;          // filled by the launcher: LDS reserved for |t|² per target (planar filter build, else 0)
  uint32_t res_bytes;        // filled by the launcher: float64-exact mode, LDS for the int32 residual counts of target and source (0 = read the float64 clouds from memory)
  unsigned* debug;           // LSLAM_DEBUG: [0] filter passes attempted, [1] passes that fell back to the exact sweep (or nullptr)
  // float64-exact mode (both set, or both nullptr): the same clouds in float64, laid out like src / tgt.  The float32
  // search then only proposes candidates; near ties are decided, and the moments accumulated, on these points — the
  // correspondences are those of a float64 search (modules/icp.py:29-41), lowest index on exact float64 ties.
  const double* src64;
  const double* tgt64;
};

struct IcpBatchTuning {
  int points_per_lane = 0;   // 0 = heuristic; 1, 2 or 4
  int threads = 0;           // 0 = from max_src_n
  int tile_points = 0;       // 0 = default cap
  int variant = 0;           // 0 = automatic, 1 = force the 128-VGPR build, 2 = force the 64-VGPR build
  int src_in_lds = 0;        // 0 = automatic, 1 = always, 2 = never
  int filter = 0;            // expanded-form filter sweep of the big build: 0 = automatic (on), 2 = off (exact sweep only)
};

// ---- exact chunk pruning (lslam_icp_pruned.hip) ----------------------------------------------------
constexpr uint32_t kFlagPrune = 2u;        // == LSLAM_ICP_PRUNE
constexpr uint32_t kFlagAllPairs = 4u;     // == LSLAM_ICP_ALL_PAIRS
constexpr int64_t kAutoPruneBatchPoints = 256;          // automatic choice: batched planar problems from this many points per cloud (profiles/r03/prune_threshold.txt: 0.142 vs 0.170 ms at 256, 0.109 vs 0.078 at 128)
constexpr double kAutoPruneLargePairEvals3D = 1e9;      // one large pair: from this many pair evaluations per iteration (tools/prune_threshold.py)
constexpr double kAutoPruneLargePairEvals2D = 2.5e8;
constexpr int kPruneMaxPoints = 4096;      // per cloud: both sorted clouds + boxes + indices must fit LDS
struct PruneScratch {                      // Morton-sorted copies: problem k owns slot [k·stride, (k+1)·stride)
  uint16_t* src_rank;                      // float64-exact running sums: position of source o in the sorted order (the inverse of src_index), same slots
  float2* src_sorted; uint16_t* src_index;
  float2* tgt_sorted; uint16_t* tgt_index;
  int64_t src_stride, tgt_stride;          // = max_src_n, max_tgt_n of the batch
  // float64-exact mode: the float64 clouds in the same sorted order (written by icp_sort_kernel), so that the moments read
  // them sequentially / from neighbouring lines instead of gathering 16 bytes per point from the caller's order
  double2* src64_sorted; double2* tgt64_sorted;
  uint16_t* margins; size_t margin_cap;    // float64-exact mode, task-queue kernel: a row of margins (halves) per LDS slot of every workgroup; capacity in entries
  float* smax;                             // per problem: the largest |p|² of its float32 source (icp_sort_kernel has the cloud in LDS anyway)
  int32_t* order;                          // problems by class, written by icp_sort_kernel: those below 2048 points per cloud from the front, the others from the back (queue[48], queue[49] count them)
  unsigned* queue;                         // 256 bytes: [0] next problem of the task-queue kernel, [1] waves that gave up waiting, [2] tie passes, [4…] profile, [48] [49] problems per class, [50] waves that gave up since the host last looked (never reset by a launch)
};
// lslam_icp_queue.hip: byte offsets inside one LDS slot (the target tile is at 0), slots per workgroup, slabs per task
struct QueueLayout {
  uint32_t slot_bytes, ctl_bytes;
  uint32_t tdelta_off, src_off, prev_off, margin_off, rows_off;      // the chunk records (16 targets + box each) start at 0; src_off = 0xffffffff: sources read from memory
  int32_t fuse;                                                    // fused search + moment tasks from the second iteration on (certificates only)
  int32_t certify;                                                 // keep matches on a proof instead of searching every source in every iteration
  int32_t contested;                                               // sources without a margin are kept on "same float32 position" (lslam_icp_queue.hip: contested_of)
  float efloor_frac;                                               // … and of every later iteration's at least (fraction of the cap)
  float e0_frac;                                                   // reach extension of the first iteration's searches as a fraction of that cap
  float ecap_frac;                                                 // cap of the reach extension as a fraction of the targets' mean spacing
  int32_t Q, G, rows_cap;                                          // G = 0: task size chosen per problem and iteration
  unsigned* queue;
  int32_t debug;
  long long timeout_ticks;                                         // idle waves leave after this many 100 MHz ticks without any sign of life in their workgroup (4 s)
  int32_t draw;                                                    // word of `queue` the problems are drawn from
  int32_t cls;                                                     // −1: problems 0 … S−1; 0 / 1: the problems of that class (PruneScratch::order)
  const int32_t* order;
  uint16_t* margins; uint32_t margin_stride;                       // float64-exact mode: slot q of workgroup b keeps its margins at margins[(b·Q + q)·margin_stride …]
};
constexpr int64_t kQueueCertifyPoints = 2048;     // problems with a cloud this large keep matches on a proof and run on sums of changes (lslam_icp_queue.hip)
hipError_t launch_icp_queue(IcpBatchArgs a, const PruneScratch& sc, int S, int64_t max_src_n, int64_t max_tgt_n, hipStream_t stream, int cls = -1, bool counters_zeroed = false);
struct PruneLayout { uint32_t aabb_off, src_off, prev_off, red_off; unsigned* debug_counter; uint32_t slab_off; };   // slab_off: planar kernel only
bool icp_pruned_supported(int dim, int64_t max_src_n, int64_t max_tgt_n);
hipError_t launch_icp_pruned(IcpBatchArgs a, const PruneScratch& sc, int S, int64_t max_src_n, int64_t max_tgt_n,
                             hipStream_t stream);

// 3-D problems (lslam_icp_pruned3d.hip): float4 sorted copies, up to 4096 targets (one 64 KB LDS tile) and 8191 sources
constexpr int kPrune3MaxTgt = 4096;
constexpr int kPrune3MaxSrc = 8191;
constexpr int64_t kAutoPruneBatchPoints3D = 256;         // automatic choice for 3-D batches (tools/batch3d_probe.py, round 3: 0.94 vs 1.07 ms at 250 points, 1.22 vs 1.43 at 1000, 1.19 vs 2.31 at 2000, 1.59 vs 3.55 at 4000; round 2: from 2000)
struct PruneScratch3 {
  float4* src_sorted; uint16_t* src_index;
  float4* tgt_sorted; uint16_t* tgt_index;
  int64_t src_stride, tgt_stride;
  double* src64_sorted; double* tgt64_sorted;      // float64-exact mode: 3 doubles per point, sorted order
};
bool icp_pruned3_supported(int64_t max_src_n, int64_t max_tgt_n);
hipError_t launch_icp_pruned3(IcpBatchArgs a, const PruneScratch3& sc, int S, int64_t max_src_n, int64_t max_tgt_n, hipStream_t stream);

hipError_t launch_icp_batch(IcpBatchArgs a, int dim, int S, int64_t max_src_n, int64_t max_tgt_n,
                            const IcpBatchTuning& tune, hipStream_t stream);

// ---- one large pair (lslam_icp_large.hip) ---------------------------------------------------------
// chunk of the whole-chip search kernel (its planar sweep keeps a whole chunk in registers)
template <int DIM> struct NnChunk;
template <> struct NnChunk<2> { static constexpr int value = 16; };
template <> struct NnChunk<3> { static constexpr int value = 8; };
constexpr int kNnThreads = 256;
constexpr int kNnTilePlanar = 4096;   // 32 KiB of LDS per workgroup → 4 workgroups per CU
constexpr int kNnTile3D = 2048;       // 32 KiB
constexpr int kNnTileMin = 128;       // smallest tile worth a workgroup (small clouds: more splits)
constexpr int kNnTargetWorkgroups = 2048;
constexpr int kNnMaxSplits = 64;
constexpr int kNnMaxSplitsFilter = 1024;   // filter search: one tile per split (8 bytes of key per source and split)
constexpr int kMomThreads = 256;
constexpr int kMomMaxBlocks = 1024;

struct LargeState {          // lives in device memory for the whole run of one problem
  double T[16];
  double err;
  StopState stop;
  int32_t done;
  int32_t ticket;                // arrival counter of the fused small-problem iteration (0 between launches)
};

struct NnArgs {
  const float* src; int64_t n;
  const float* tgt; int64_t m;
  unsigned long long* keys;      // splits x n packed (d² bits << 32 | index); row s written by split s
  const LargeState* state;       // transform + done flag, or nullptr → use T below
  double T[16];
  int32_t tile_points;           // filled by the launcher
  int64_t chunks_per_split;      // filled by the launcher
  // fused tail (small problems): the last workgroup to finish also reduces the keys, accumulates the
  // moments, solves and applies the stop rules — one launch per iteration instead of three
  LargeState* fuse_state;        // nullptr = no fused tail
  int32_t* out_corr;
  int32_t splits;
  uint32_t flags;
  double epsilon, stopping_thresh;
  int32_t max_iters;
  // float64-exact mode (three-launch path only): second smallest distance per (split, source), laid out like keys, and
  // {max |t|², max |p|²} of the float32 clouds (launch_cloud_bounds)
  float* sec;
  const float* bounds;
};

struct NnGeometry {
  int P;                         // source points per lane
  unsigned grid_x;               // source blocks
  int splits;                    // target splits = grid.y = rows of `keys`
  int tile_points;
  int64_t chunks_per_split;
  size_t smem;
  bool filter;                   // 3-D: nn_tiled_filter_kernel (expanded-form sweep, one tile per split)
};

struct MomArgs {
  const float* src; int64_t n;
  const float* tgt;
  const unsigned long long* keys;
  int32_t splits;
  const LargeState* state;
  double* partials;              // rows x AccLayout<DIM>::kTotal
  int32_t* out_corr;             // or nullptr
  uint32_t flags;
  // float64-exact mode: the float64 clouds, the target size, the splits' second smallest distances, the clouds' bounds
  const double* src64; const double* tgt64;
  int64_t m;
  const float* sec;
  const float* bounds;
};

struct SolveArgs {
  LargeState* state;
  const double* partials;
  int32_t rows;
  uint32_t flags;
  double epsilon, stopping_thresh;
  int32_t max_iters;
};

NnGeometry nn_geometry(int dim, int64_t n, int64_t m);
hipError_t launch_nn(NnArgs a, int dim, const NnGeometry& g, hipStream_t stream);
hipError_t launch_cloud_bounds(const float* src, int64_t n, const float* tgt, int64_t m, int dim, float* bounds, hipStream_t stream);
hipError_t launch_nn_finalize(const unsigned long long* keys, int64_t n, int splits, int32_t* out_idx, float* out_d2, hipStream_t stream);
int moments_rows(int64_t n);
hipError_t launch_moments(MomArgs a, int dim, int rows, hipStream_t stream);
hipError_t launch_solve(SolveArgs a, int dim, hipStream_t stream);

// ---- one LiDAR-sized pair iterated inside ONE launch by co-resident workgroups (lslam_icp_coop.hip) ----------
struct CoopArgs {
  const float* src; int32_t n;
  const float* tgt; int32_t m;
  const double* src64;           // float64-exact mode: both set, or both nullptr
  const double* tgt64;
  LargeState* state;             // in: T = the start transform; out: T, err, stop, done = 1
  unsigned long long* gran;      // icp_coop_gran_bytes(): the workgroups' moment rows, zeroed by the launcher
  unsigned* failed;              // set to 1 if a workgroup gave up waiting for the others (never on a healthy device)
  int32_t* out_corr;             // or nullptr
  uint32_t flags;
  double epsilon, stopping_thresh;
  int32_t max_iters;
  uint32_t tile_bytes, tile64_bytes, box_bytes, gran_bytes;   // filled by the launcher
  unsigned long long timeout_ticks;                           // filled by the launcher: bound of one hand-off wait (100 MHz ticks)
};
bool icp_coop_supported(int dim, int64_t n, int64_t m, bool f64);
size_t icp_coop_gran_bytes(int dim, int64_t n);
hipError_t launch_icp_coop(CoopArgs a, int dim, hipStream_t stream);

// ---- whole-chip path with exact box pruning (lslam_icp_large_pruned.hip) ------------------------------
struct LpT { float m[3][4]; };
struct LpWs {                      // device workspace of one problem
  void* src_sorted; int32_t* src_index;        // float2 / float4 per point, original index
  void* tgt_sorted; int32_t* tgt_index;
  float4 *lo1, *hi1, *lo2, *hi2, *lo3, *hi3;   // box hierarchy over the sorted target
  int32_t* match;                              // per sorted source: sorted target position of its last match
  unsigned *keys_in, *keys_out, *vals_in, *vals_out;
  float* bbox_partial;                         // 1024 x 6
  float* grid;                                 // lo[3], scale[3]
  int32_t* flag_count; int32_t* flag_src; float* flag_lim;     // float64-exact mode: near-tie list (n entries)
  void* temp; size_t temp_bytes;
};
struct LpTree { int64_t n1, n2, n3; };
struct LpNnArgs {
  const void* src_sorted; int64_t n;
  const void* tgt_sorted; const int32_t* tgt_index; int64_t m;
  const float4 *lo1, *hi1, *lo2, *hi2, *lo3, *hi3;
  int64_t n1, n2, n3;
  int32_t* match;
  const LargeState* state;
  // float64-exact mode: {max |t|², max |p|²} of the clouds, and the list of sources with a near tie (sorted source index,
  // window) that lp_launch_refine then decides on the float64 points
  const float* bounds;
  int32_t* flag_count; int32_t* flag_src; float* flag_lim;
};
struct LpRefineArgs {
  const void* src_sorted; const int32_t* src_index;
  const void* tgt_sorted; const int32_t* tgt_index; int64_t m;
  const double* src64; const double* tgt64;
  const LargeState* state;
  const int32_t* flag_count; const int32_t* flag_src; const float* flag_lim;
  int32_t* match;
  const float4 *lo1, *hi1; int64_t n1;       // boxes of the target's chunks (level 1 of the hierarchy)
};
struct LpMomArgs {
  const void* src_sorted; const int32_t* src_index; int64_t n;
  const void* tgt_sorted; const int32_t* tgt_index;
  const int32_t* match;
  const LargeState* state;
  double* partials;
  int32_t* out_corr;
  uint32_t flags;
  const double* src64; const double* tgt64;    // float64-exact mode (original order), else nullptr
};
size_t lp_sort_temp_bytes(int64_t n);
hipError_t lp_prepare(int dim, const float* d_src, int64_t n, const float* d_tgt, int64_t m, const double* init_T,
                      const LpWs& ws, LpTree* tree, hipStream_t st);
hipError_t lp_launch_nn(int dim, const LpNnArgs& a, hipStream_t st);
hipError_t lp_launch_refine(int dim, const LpRefineArgs& a, hipStream_t st);
hipError_t lp_launch_moments(int dim, const LpMomArgs& a, int rows, hipStream_t st);

// ---- float64 pair operators (lslam_ops.hip) -------------------------------------------------------
struct PairOpArgs {
  const double* p; int64_t p_stride;
  const double* q; int64_t q_stride;
  int64_t n;
  int32_t cols;            // coordinates used per row (≤ 3)
  int32_t normalize;
  double* out_T;           // device, 16 doubles (or nullptr)
  double* out_err;         // device, 1 double (or nullptr)
};
hipError_t launch_pair_op(const PairOpArgs& a, hipStream_t stream);
hipError_t run_valu_probe(float* d_scratch, hipStream_t stream, double* lane_ops_per_s);

// ---- occupancy grid (lslam_ogm.hip) -----------------------------------------------------------------
constexpr int kOgmPlanes = 128;  // scans per tile/apply launch pair, each with its own int32 count plane

struct OgmGrid {
  float* logodds;          // width*height, [x][y]: the BASE of the log-odds, value = base + steps·step (float64 sum, rounded once on read)
  int32_t* steps;          // width*height: net hits − misses accumulated on the base since it was last set (by a clip or a host write)
  int32_t* counts;         // kOgmPlanes planes of width*height hit-miss counts (int16, or int32 for scans above 32767 rays); a plane is valid inside its scan's box only
  int32_t width, height;
  double res, min_x, min_y;
  double step, clip;
};
struct OgmScan {
  const double* scan_xy;   // device, n x 2
  double x, y, c, s;       // pose, cos/sin of yaw
  int64_t ray_base;        // first ray record of this scan
  int64_t batch_base;      // first 64-ray batch box of this scan
  int32_t n;
  int32_t pad_;
};
// d_rays: one int4 {end_x, end_y, k_first, k_last} per ray; d_boxes: {min_x, min_y, −max_x, −max_y} per scan;
// d_batch_boxes: the same per 64 consecutive rays of a scan.  `wide`: int32 count planes (a scan above 32767 rays), else int16.
hipError_t launch_ogm_rays(const OgmGrid& g, const OgmScan* d_scans, int nscans, int max_rays, int4* d_rays, int* d_boxes, int4* d_batch_boxes,
                           hipStream_t stream);
// tiles_x / tiles_y: the largest number of map tiles a box of the group spans per axis (ogm_tile_span on the host copy of the boxes)
hipError_t launch_ogm_tiles(const OgmGrid& g, const OgmScan* d_scans, int nscans, const int4* d_rays, const int* d_boxes,
                            const int4* d_batch_boxes, bool wide, int tiles_x, int tiles_y, hipStream_t stream);
// (x0, y0, x1, y1): union of the group's boxes
hipError_t launch_ogm_apply(const OgmGrid& g, const int* d_boxes, int nscans, bool wide, int x0, int y0, int x1, int y1, hipStream_t stream);
void ogm_tile_span(const int* box, int* tiles_x, int* tiles_y);
constexpr int kOgmBoxEmpty = 0x7f7f7f7f;
hipError_t launch_ogm_binarize(const OgmGrid& g, uint8_t* out, hipStream_t stream);
hipError_t launch_ogm_values(const OgmGrid& g, float* out, hipStream_t stream);   // out = f32(base + steps·step); out == g.logodds folds in place (steps ← 0)
hipError_t launch_ogm_init(const OgmGrid& g, hipStream_t stream);

// ---- LiDAR pre-processing (lslam_preprocess.hip) ----------------------------------------------------
constexpr int kRadialParts = 1024;
hipError_t launch_polar_mask(const double* ranges, int32_t K, int32_t B, const double* cos_a, const double* sin_a,
                             double off_x, double off_y, double rmin, double rmax, double* xy_full, uint8_t* keep,
                             int32_t* counts, hipStream_t st);
hipError_t launch_dbscan_mask(const double* xy, const int64_t* off, int32_t K, int64_t max_n, double eps, int32_t min_samples,
                              uint8_t* core, uint8_t* keep, int32_t* counts, hipStream_t st);
hipError_t launch_radial_mask(const double* xy, const int64_t* off, int32_t K, int64_t total, double k_std, double* partial,
                              double* stats, uint8_t* keep, int32_t* counts, hipStream_t st);
hipError_t launch_f64_to_f32(const double* in, float* out, int64_t n, hipStream_t st);
hipError_t launch_compact(const double* xy, const int64_t* in_off, int32_t B, const uint8_t* keep, const int32_t* counts,
                          int32_t K, int64_t* out_off, double* out_xy, hipStream_t st);

// ---- voxel_downsample (lslam_voxel.hip) -----------------------------------------------------------------
struct VoxelWs {
  double* partial;              // 1024 x 3
  double* lo;                   // 3
  int* bad;                     // 1
  unsigned long long* keys_in;  // n
  unsigned long long* keys_out; // n
  uint32_t* ids_in;             // n
  uint32_t* ids_out;            // n
  int64_t* starts;              // n + 1
  void* temp; size_t temp_bytes;
};
size_t voxel_temp_bytes(int64_t n);
hipError_t run_voxel_downsample(const double* pts, int64_t n, int dim, double voxel, VoxelWs ws, double* out,
                                int64_t* h_nvox, int* h_bad, hipStream_t st);

}  // namespace lslam
