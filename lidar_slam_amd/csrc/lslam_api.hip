// lslam_api.hip — the C ABI declared in include/lslam.h.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "lslam_host.h"
#include "lslam_kernels.h"

namespace lslam {

// ---- error string ---------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
void clear_error() { g_err[0] = 0; }

static thread_local int g_coop_fallbacks = 0;     // calls of this thread whose co-resident launch timed out and was re-run launch by launch

// ---- scratch arena --------------------------------------------------------------------------------
int Arena::reserve(int device, size_t bytes) {
  used_ = 0;
  if (device == device_ && bytes <= cap_) return LSLAM_OK;
  if (base_) { (void)hipFree(base_); base_ = nullptr; cap_ = 0; }
  size_t want = bytes + (bytes >> 2) + (1u << 20);          // head-room: avoid re-growing every call
  LSLAM_HIP_TRY(hipMalloc(&base_, want));
  cap_ = want;
  device_ = device;
  return LSLAM_OK;
}
void Arena::release() {
  if (base_) (void)hipFree(base_);
  base_ = nullptr; cap_ = 0; used_ = 0; device_ = -1;
}
void* Arena::carve(size_t bytes) {
  void* p = static_cast<char*>(base_) + used_;
  used_ += align(bytes);
  return used_ <= cap_ ? p : nullptr;
}
Arena& thread_arena() {
  static thread_local Arena a;
  return a;
}
static thread_local hipStream_t g_thread_stream = nullptr;
static thread_local int g_thread_stream_dev = -1;
int thread_stream(hipStream_t* out) {
  hipStream_t& s = g_thread_stream;
  int& dev = g_thread_stream_dev;
  int cur = 0;
  LSLAM_HIP_TRY(hipGetDevice(&cur));
  if (!s || dev != cur) {
    if (s) {                                                 // the thread moved to another device: the old stream is drained and destroyed
      int back = cur;
      if (hipSetDevice(dev) == hipSuccess) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
      (void)hipSetDevice(back);
      s = nullptr;
    }
    LSLAM_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    dev = cur;
  }
  *out = s;
  return LSLAM_OK;
}

static int use_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return fail(LSLAM_ERR_HIP, "no HIP device available (%s)", hipGetErrorString(e));
  if (device < 0 || device >= n) return fail(LSLAM_ERR_BAD_ARG, "device %d out of range [0,%d)", device, n);
  LSLAM_HIP_TRY(hipSetDevice(device));
  return LSLAM_OK;
}

static bool all_finite(const float* v, int64_t count) {
  // a float is non-finite iff its exponent bits are all ones
  const uint32_t* u = reinterpret_cast<const uint32_t*>(v);
  uint32_t bad = 0;
  for (int64_t i = 0; i < count; ++i) bad |= ((u[i] & 0x7f800000u) == 0x7f800000u);
  return bad == 0;
}

static int check_offsets(const int64_t* off, int32_t S, const char* what, int64_t* max_len, int64_t* total) {
  int64_t mx = 0;
  if (off[0] < 0) return fail(LSLAM_ERR_BAD_ARG, "%s[0] is negative", what);
  for (int32_t s = 0; s < S; ++s) {
    const int64_t len = off[s + 1] - off[s];
    if (len < 0) return fail(LSLAM_ERR_BAD_ARG, "%s is not monotone at %d", what, s);
    if (len > 0x7fffffff) return fail(LSLAM_ERR_BAD_ARG, "%s: problem %d has more than 2^31 points", what, s);
    if (len > mx) mx = len;
  }
  *max_len = mx;
  *total = off[S];
  return LSLAM_OK;
}

// search strategy of a call: explicit flag bits win, then LSLAM_ICP_PRUNE in the environment, then `automatic`
static bool want_pruning(uint32_t flags, bool automatic) {
  if (flags & kFlagAllPairs) return false;
  if (flags & kFlagPrune) return true;
  if (const char* e = getenv("LSLAM_ICP_PRUNE")) return atoi(e) != 0;
  return automatic;
}

static IcpBatchTuning tuning_from_env() {
  IcpBatchTuning t;
  if (const char* e = getenv("LSLAM_ICP_POINTS_PER_LANE")) t.points_per_lane = atoi(e);
  if (const char* e = getenv("LSLAM_ICP_THREADS")) t.threads = atoi(e);
  if (const char* e = getenv("LSLAM_ICP_TILE_POINTS")) t.tile_points = atoi(e);
  if (const char* e = getenv("LSLAM_ICP_VARIANT")) t.variant = atoi(e);
  if (const char* e = getenv("LSLAM_ICP_SRC_IN_LDS")) t.src_in_lds = atoi(e);
  if (const char* e = getenv("LSLAM_ICP_FILTER")) t.filter = atoi(e) != 0 ? 0 : 2;
  return t;
}

// A thread's device workspaces (Morton-sorted copies, key rows, iteration state, read-back slots) are reused by its
// next call.  Calls on ONE stream are ordered by the stream itself; a call on a different stream first waits for
// the event recorded when the previous call's last work was enqueued (the whole-chip driver returns while a
// queued-ahead group may still be pending).
struct WsFence {
  hipEvent_t ev = nullptr;
  hipStream_t last = nullptr;
  int device = -1;
  bool armed = false;
  int enter(hipStream_t st) {
    int dev = 0;
    LSLAM_HIP_TRY(hipGetDevice(&dev));
    if (dev != device) {                                     // workspaces of another device are rebuilt (hipFree synchronises)
      if (ev) (void)hipEventDestroy(ev);
      ev = nullptr; armed = false; device = dev;
      LSLAM_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    if (armed && last != st) LSLAM_HIP_TRY(hipStreamWaitEvent(st, ev, 0));
    return LSLAM_OK;
  }
  int leave(hipStream_t st) {
    LSLAM_HIP_TRY(hipEventRecord(ev, st));
    last = st; armed = true;
    return LSLAM_OK;
  }
};
static WsFence& ws_fence() { static thread_local WsFence f; return f; }

// Spatially ordered copies of a pruned batch (pair-local slots of S x max points: the caller's offsets live on the device and may
// start anywhere), kept by the calling thread and reused by its next call; grown behind a hipDeviceSynchronize().
struct PruneHolder {
  PruneScratch sc{};
  size_t cap_src = 0, cap_tgt = 0, cap_src64 = 0, cap_tgt64 = 0, cap_order = 0;
  int device = -1;
  void release() {
    if (sc.src_sorted) (void)hipFree(sc.src_sorted);
    if (sc.src_index) (void)hipFree(sc.src_index);
    if (sc.src_rank) (void)hipFree(sc.src_rank);
    if (sc.tgt_sorted) (void)hipFree(sc.tgt_sorted);
    if (sc.tgt_index) (void)hipFree(sc.tgt_index);
    if (sc.src64_sorted) (void)hipFree(sc.src64_sorted);
    if (sc.tgt64_sorted) (void)hipFree(sc.tgt64_sorted);
    if (sc.queue) (void)hipFree(sc.queue);
    if (sc.margins) (void)hipFree(sc.margins);
    if (sc.order) (void)hipFree(sc.order);
    if (sc.smax) (void)hipFree(sc.smax);
    cap_order = 0;
    sc = PruneScratch{}; cap_src = cap_tgt = cap_src64 = cap_tgt64 = 0; device = -1;
  }
  int ensure(int32_t S, int64_t max_src_n, int64_t max_tgt_n, bool f64) {
    int dev = 0;
    LSLAM_HIP_TRY(hipGetDevice(&dev));
    const size_t need_src = (size_t)S * (size_t)max_src_n, need_tgt = (size_t)S * (size_t)max_tgt_n;   // one slot per problem
    if (dev != device || need_src > cap_src || need_tgt > cap_tgt) {
      LSLAM_HIP_TRY(hipDeviceSynchronize());                 // earlier launches may still read the old buffers
      release();
      device = dev;
      LSLAM_HIP_TRY(hipMalloc(&sc.queue, 256));
      LSLAM_HIP_TRY(hipMemset(sc.queue, 0, 256));
      LSLAM_HIP_TRY(hipMalloc(&sc.src_sorted, sizeof(float2) * need_src));
      LSLAM_HIP_TRY(hipMalloc(&sc.src_index, sizeof(uint16_t) * need_src));
      LSLAM_HIP_TRY(hipMalloc(&sc.src_rank, sizeof(uint16_t) * need_src));
      LSLAM_HIP_TRY(hipMalloc(&sc.tgt_sorted, sizeof(float2) * need_tgt));
      LSLAM_HIP_TRY(hipMalloc(&sc.tgt_index, sizeof(uint16_t) * need_tgt));
      cap_src = need_src; cap_tgt = need_tgt;
    }
    // float64-exact mode: the float64 clouds in the sorted order as well (allocated on first use, sized like the others)
    if (f64 && (need_src > cap_src64 || need_tgt > cap_tgt64 || !sc.src64_sorted)) {
      LSLAM_HIP_TRY(hipDeviceSynchronize());
      if (sc.src64_sorted) (void)hipFree(sc.src64_sorted);
      if (sc.tgt64_sorted) (void)hipFree(sc.tgt64_sorted);
      sc.src64_sorted = nullptr; sc.tgt64_sorted = nullptr; cap_src64 = cap_tgt64 = 0;
      LSLAM_HIP_TRY(hipMalloc(&sc.src64_sorted, sizeof(double2) * need_src));
      LSLAM_HIP_TRY(hipMalloc(&sc.tgt64_sorted, sizeof(double2) * need_tgt));
      cap_src64 = need_src; cap_tgt64 = need_tgt;
    }
    if ((size_t)S > cap_order) {                               // the problems by class (icp_sort_kernel)
      LSLAM_HIP_TRY(hipDeviceSynchronize());
      if (sc.order) (void)hipFree(sc.order);
      sc.order = nullptr; cap_order = 0;
      LSLAM_HIP_TRY(hipMalloc(&sc.order, sizeof(int32_t) * (size_t)S));
      if (sc.smax) (void)hipFree(sc.smax);
      sc.smax = nullptr;
      LSLAM_HIP_TRY(hipMalloc(&sc.smax, sizeof(float) * (size_t)S));
      cap_order = (size_t)S;
    }
    // the task-queue kernel's margins (float64-exact mode, running sums): at most 16 slots of kPruneMaxPoints sources per workgroup, one workgroup per CU
    if (!sc.margins && (max_src_n >= kQueueCertifyPoints || max_tgt_n >= kQueueCertifyPoints || getenv("LSLAM_ICP_QUEUE_CERTIFY"))) {
      int cus = 0;
      LSLAM_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      if (cus <= 0) cus = 256;
      sc.margin_cap = (size_t)cus * 16 * (size_t)kPruneMaxPoints * 3;      // in halves; the running-sums build keeps a float and a match per source
      LSLAM_HIP_TRY(hipMalloc(&sc.margins, sizeof(uint16_t) * sc.margin_cap));
    }
    sc.src_stride = max_src_n; sc.tgt_stride = max_tgt_n;
    return LSLAM_OK;
  }
};
static PruneHolder& prune_holder() { static thread_local PruneHolder h; return h; }

struct Prune3Holder {
  PruneScratch3 sc{};
  size_t cap_src = 0, cap_tgt = 0, cap_src64 = 0, cap_tgt64 = 0;
  int device = -1;
  void release() {
    if (sc.src_sorted) (void)hipFree(sc.src_sorted);
    if (sc.src_index) (void)hipFree(sc.src_index);
    if (sc.tgt_sorted) (void)hipFree(sc.tgt_sorted);
    if (sc.tgt_index) (void)hipFree(sc.tgt_index);
    if (sc.src64_sorted) (void)hipFree(sc.src64_sorted);
    if (sc.tgt64_sorted) (void)hipFree(sc.tgt64_sorted);
    sc = PruneScratch3{}; cap_src = cap_tgt = cap_src64 = cap_tgt64 = 0; device = -1;
  }
  int ensure(int32_t S, int64_t max_src_n, int64_t max_tgt_n, bool f64) {
    int dev = 0;
    LSLAM_HIP_TRY(hipGetDevice(&dev));
    const size_t need_src = (size_t)S * (size_t)max_src_n, need_tgt = (size_t)S * (size_t)max_tgt_n;
    if (dev != device || need_src > cap_src || need_tgt > cap_tgt) {
      LSLAM_HIP_TRY(hipDeviceSynchronize());
      release();
      device = dev;
      LSLAM_HIP_TRY(hipMalloc(&sc.src_sorted, sizeof(float4) * need_src));
      LSLAM_HIP_TRY(hipMalloc(&sc.src_index, sizeof(uint16_t) * need_src));
      LSLAM_HIP_TRY(hipMalloc(&sc.tgt_sorted, sizeof(float4) * need_tgt));
      LSLAM_HIP_TRY(hipMalloc(&sc.tgt_index, sizeof(uint16_t) * need_tgt));
      cap_src = need_src; cap_tgt = need_tgt;
    }
    if (f64 && (need_src > cap_src64 || need_tgt > cap_tgt64 || !sc.src64_sorted)) {
      LSLAM_HIP_TRY(hipDeviceSynchronize());
      if (sc.src64_sorted) (void)hipFree(sc.src64_sorted);
      if (sc.tgt64_sorted) (void)hipFree(sc.tgt64_sorted);
      sc.src64_sorted = nullptr; sc.tgt64_sorted = nullptr; cap_src64 = cap_tgt64 = 0;
      LSLAM_HIP_TRY(hipMalloc(&sc.src64_sorted, sizeof(double) * 3 * need_src));
      LSLAM_HIP_TRY(hipMalloc(&sc.tgt64_sorted, sizeof(double) * 3 * need_tgt));
      cap_src64 = need_src; cap_tgt64 = need_tgt;
    }
    sc.src_stride = max_src_n; sc.tgt_stride = max_tgt_n;
    return LSLAM_OK;
  }
};
static Prune3Holder& prune3_holder() { static thread_local Prune3Holder h; return h; }

// pinned host staging buffers of the calling thread (lslam_host_staging)
struct StagingHolder {
  void* buf[2] = {nullptr, nullptr};
  size_t cap[2] = {0, 0};
  void release() {
    for (int k = 0; k < 2; ++k) { if (buf[k]) (void)hipHostFree(buf[k]); buf[k] = nullptr; cap[k] = 0; }
  }
};
static StagingHolder& staging_holder() { static thread_local StagingHolder h; return h; }

}  // namespace lslam

using namespace lslam;

extern "C" {

int lslam_abi_version(void) { return LSLAM_ABI_VERSION; }
const char* lslam_last_error(void) { return g_err; }

int lslam_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int lslam_set_device(int device) {
  clear_error();
  return use_device(device);
}

// Pinned host staging buffers of the calling thread (two slots): packing a scan list straight into one makes the upload
// that follows a single DMA at PCIe speed instead of a staged copy from pageable memory.
int lslam_host_staging(int slot, size_t bytes, void** out_ptr) {
  clear_error();
  if (!out_ptr || slot < 0 || slot > 1) return fail(LSLAM_ERR_BAD_ARG, "slot must be 0 or 1 and out_ptr non-null");
  StagingHolder& sh = staging_holder();
  void** buf = sh.buf;
  size_t* cap = sh.cap;
  if (bytes > cap[slot]) {
    if (buf[slot]) { LSLAM_HIP_TRY(hipDeviceSynchronize()); (void)hipHostFree(buf[slot]); }
    buf[slot] = nullptr; cap[slot] = 0;
    const size_t want = bytes + (bytes >> 2) + (1u << 20);
    LSLAM_HIP_TRY(hipHostMalloc(&buf[slot], want, hipHostMallocDefault));
    cap[slot] = want;
  }
  *out_ptr = buf[slot];
  return LSLAM_OK;
}

int lslam_valu_probe(int device, double* out_lane_ops_per_s) {
  clear_error();
  if (!out_lane_ops_per_s) return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  int rc;
  if ((rc = use_device(device)) != LSLAM_OK) return rc;
  Arena& ar = thread_arena();
  if ((rc = ar.reserve(device, sizeof(float) * 2048 * 256)) != LSLAM_OK) return rc;
  float* d = static_cast<float*>(ar.carve(sizeof(float) * 2048 * 256));
  hipStream_t st;
  if ((rc = thread_stream(&st)) != LSLAM_OK) return rc;
  LSLAM_HIP_TRY(run_valu_probe(d, st, out_lane_ops_per_s));
  return LSLAM_OK;
}

// ---------------------------------------------------------------------------------------------------
// which pruned kernel a batched call takes (0: the all-pairs sweep, which touches no workspace of the thread).
// automatic: planar batches from kAutoPruneBatchPoints points per cloud, 3-D batches from kAutoPruneBatchPoints3D
static int batch_prune_choice(int dim, uint32_t flags, int64_t max_src_n, int64_t max_tgt_n) {
  const bool prune = dim == 2 ? want_pruning(flags, max_src_n >= kAutoPruneBatchPoints && max_tgt_n >= kAutoPruneBatchPoints)
                              : want_pruning(flags, max_src_n >= kAutoPruneBatchPoints3D && max_tgt_n >= kAutoPruneBatchPoints3D);
  if (prune && dim == 3 && icp_pruned3_supported(max_src_n, max_tgt_n)) return 3;
  if (prune && dim == 2 && icp_pruned_supported(dim, max_src_n, max_tgt_n)) return 2;
  return 0;
}

static int icp_batch_dev_impl(const float* d_src, const double* d_src64, const int64_t* d_src_off, const float* d_tgt,
                              const double* d_tgt64, const int64_t* d_tgt_off, int dim, const double* d_init_T, int32_t S,
                              int64_t max_src_n, int64_t max_tgt_n, double epsilon, double stopping_thresh,
                              int32_t max_iters, uint32_t flags, double* d_out_T, double* d_out_err,
                              int32_t* d_out_iters, int32_t* d_out_corr, void* stream) {
  if (dim != 2 && dim != 3) return fail(LSLAM_ERR_BAD_ARG, "dim must be 2 or 3, got %d", dim);
  if (S < 0) return fail(LSLAM_ERR_BAD_ARG, "negative batch size");
  if (S == 0) return LSLAM_OK;
  if (!d_src || !d_src_off || !d_tgt || !d_tgt_off || !d_init_T || !d_out_T || !d_out_err || !d_out_iters)
    return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  if (max_src_n < 0 || max_tgt_n <= 0) return fail(LSLAM_ERR_BAD_ARG, "max_src_n/max_tgt_n must be positive");
  if ((d_src64 == nullptr) != (d_tgt64 == nullptr)) return fail(LSLAM_ERR_BAD_ARG, "float64 clouds: pass both or neither");
  IcpBatchArgs a{};
  a.src = d_src; a.src_off = d_src_off; a.tgt = d_tgt; a.tgt_off = d_tgt_off;
  a.src64 = d_src64; a.tgt64 = d_tgt64;
  a.init_T = d_init_T; a.out_T = d_out_T; a.out_err = d_out_err; a.out_iters = d_out_iters;
  a.out_corr = d_out_corr;
  a.epsilon = epsilon; a.stopping_thresh = stopping_thresh; a.max_iters = max_iters; a.flags = flags;
  const int pruned = batch_prune_choice(dim, flags, max_src_n, max_tgt_n);
  const bool prune = pruned != 0;
  if (prune && dim == 3 && icp_pruned3_supported(max_src_n, max_tgt_n)) {
    Prune3Holder& h = prune3_holder();
    int rc = h.ensure(S, max_src_n, max_tgt_n, d_src64 != nullptr);
    if (rc != LSLAM_OK) return rc;
    LSLAM_HIP_TRY(launch_icp_pruned3(a, h.sc, S, max_src_n, max_tgt_n, static_cast<hipStream_t>(stream)));
    return LSLAM_OK;
  }
  if (prune && dim == 2 && icp_pruned_supported(dim, max_src_n, max_tgt_n)) {
    PruneHolder& h = prune_holder();
    int rc = h.ensure(S, max_src_n, max_tgt_n, d_src64 != nullptr);
    if (rc != LSLAM_OK) return rc;
    LSLAM_HIP_TRY(launch_icp_pruned(a, h.sc, S, max_src_n, max_tgt_n, static_cast<hipStream_t>(stream)));
    return LSLAM_OK;
  }
  LSLAM_HIP_TRY(launch_icp_batch(a, dim, S, max_src_n, max_tgt_n, tuning_from_env(), static_cast<hipStream_t>(stream)));
  return LSLAM_OK;
}

int lslam_icp_batch_dev64(const float* d_src, const double* d_src64, const int64_t* d_src_off, const float* d_tgt,
                          const double* d_tgt64, const int64_t* d_tgt_off, int dim, const double* d_init_T, int32_t S,
                          int64_t max_src_n, int64_t max_tgt_n, double epsilon, double stopping_thresh,
                          int32_t max_iters, uint32_t flags, double* d_out_T, double* d_out_err,
                          int32_t* d_out_iters, int32_t* d_out_corr, void* stream) {
  clear_error();
  // only the pruned kernels use a workspace of the thread (the Morton-sorted copies): the all-pairs sweep needs no ordering
  // against the thread's earlier calls on other streams
  const bool uses_ws = (dim == 2 || dim == 3) && batch_prune_choice(dim, flags, max_src_n, max_tgt_n) != 0;
  int rc = uses_ws ? ws_fence().enter(static_cast<hipStream_t>(stream)) : LSLAM_OK;
  if (rc != LSLAM_OK) return rc;
  rc = icp_batch_dev_impl(d_src, d_src64, d_src_off, d_tgt, d_tgt64, d_tgt_off, dim, d_init_T, S, max_src_n, max_tgt_n, epsilon,
                          stopping_thresh, max_iters, flags, d_out_T, d_out_err, d_out_iters, d_out_corr, stream);
  const int rc2 = uses_ws ? ws_fence().leave(static_cast<hipStream_t>(stream)) : LSLAM_OK;
  return rc != LSLAM_OK ? rc : rc2;
}

int lslam_icp_batch_dev(const float* d_src, const int64_t* d_src_off, const float* d_tgt,
                        const int64_t* d_tgt_off, int dim, const double* d_init_T, int32_t S,
                        int64_t max_src_n, int64_t max_tgt_n, double epsilon, double stopping_thresh,
                        int32_t max_iters, uint32_t flags, double* d_out_T, double* d_out_err,
                        int32_t* d_out_iters, int32_t* d_out_corr, void* stream) {
  return lslam_icp_batch_dev64(d_src, nullptr, d_src_off, d_tgt, nullptr, d_tgt_off, dim, d_init_T, S, max_src_n, max_tgt_n,
                               epsilon, stopping_thresh, max_iters, flags, d_out_T, d_out_err, d_out_iters, d_out_corr, stream);
}

extern "C++" {
static bool all_finite64(const double* v, int64_t count) {
  const uint64_t* u = reinterpret_cast<const uint64_t*>(v);
  uint64_t bad = 0;
  for (int64_t i = 0; i < count; ++i) bad |= ((u[i] & 0x7ff0000000000000ull) == 0x7ff0000000000000ull);
  return bad == 0;
}

// host arrays in / out; PT = float (points as they are) or double (float64-exact mode: the float64 clouds are uploaded,
// rounded to float32 on the device for the search, and decide near ties / feed the moments)
template <typename PT>
static int icp_batch_host(const PT* src, const int64_t* src_off, const PT* tgt, const int64_t* tgt_off,
                          int dim, const double* init_T, int32_t S, double epsilon, double stopping_thresh,
                          int32_t max_iters, uint32_t flags, double* out_T, double* out_err,
                          int32_t* out_iters, int32_t* out_corr, int device) {
  constexpr bool kF64 = sizeof(PT) == sizeof(double);
  clear_error();
  if (dim != 2 && dim != 3) return fail(LSLAM_ERR_BAD_ARG, "dim must be 2 or 3, got %d", dim);
  if (S < 0) return fail(LSLAM_ERR_BAD_ARG, "negative batch size");
  if (S == 0) return LSLAM_OK;
  if (!src || !src_off || !tgt || !tgt_off || !init_T || !out_T || !out_err || !out_iters)
    return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  int64_t max_n = 0, max_m = 0, tot_n = 0, tot_m = 0;
  int rc;
  if ((rc = check_offsets(src_off, S, "src_off", &max_n, &tot_n)) != LSLAM_OK) return rc;
  if ((rc = check_offsets(tgt_off, S, "tgt_off", &max_m, &tot_m)) != LSLAM_OK) return rc;
  for (int32_t s = 0; s < S; ++s)
    if (tgt_off[s + 1] == tgt_off[s]) return fail(LSLAM_ERR_EMPTY_TARGET, "problem %d has an empty target cloud", s);
  bool finite = true;
  if (!(flags & LSLAM_ICP_INPUT_CHECKED)) {                // the caller has not vouched for the points
    if constexpr (kF64) finite = all_finite64(src + dim * src_off[0], dim * (tot_n - src_off[0])) && all_finite64(tgt + dim * tgt_off[0], dim * (tot_m - tgt_off[0]));
    else finite = all_finite(src + dim * src_off[0], dim * (tot_n - src_off[0])) && all_finite(tgt + dim * tgt_off[0], dim * (tot_m - tgt_off[0]));
  }
  if (!finite) return fail(LSLAM_ERR_NONFINITE, "point clouds must be finite (NaN or Inf found)");
  if (dim == 2)                                           // the planar kernels never materialise z
    for (int32_t s = 0; s < S; ++s) {
      const double* T = init_T + 16 * (size_t)s;
      if (T[8] != 0.0 || T[9] != 0.0 || T[11] != 0.0)
        return fail(LSLAM_ERR_BAD_ARG, "problem %d: init_T moves planar points out of the z = 0 plane; pass the clouds as (n,3)", s);
    }
  if ((rc = use_device(device)) != LSLAM_OK) return rc;

  const bool shared = (src == tgt);                       // scan sequence: upload the scans once
  const int64_t up_src = shared ? (tot_n > tot_m ? tot_n : tot_m) : tot_n;
  const size_t n_src = (size_t)dim * (size_t)up_src, n_tgt = shared ? 0 : (size_t)dim * (size_t)tot_m;
  const size_t b_src = sizeof(float) * n_src, b_tgt = sizeof(float) * n_tgt;
  const size_t b_src64 = kF64 ? sizeof(double) * n_src : 0, b_tgt64 = kF64 ? sizeof(double) * n_tgt : 0;
  const size_t b_off = sizeof(int64_t) * ((size_t)S + 1);
  const size_t b_T = sizeof(double) * 16 * (size_t)S;
  const size_t b_corr = out_corr ? sizeof(int32_t) * (size_t)tot_n : 0;
  Arena& ar = thread_arena();
  const size_t total = Arena::align(b_src) + Arena::align(b_tgt) + Arena::align(b_src64) + Arena::align(b_tgt64) + 2 * Arena::align(b_off) +
                       2 * Arena::align(b_T) + Arena::align(sizeof(double) * S) + Arena::align(sizeof(int32_t) * S) + Arena::align(b_corr);
  if ((rc = ar.reserve(device, total)) != LSLAM_OK) return rc;
  float* d_src = static_cast<float*>(ar.carve(b_src));
  float* d_tgt = shared ? d_src : static_cast<float*>(ar.carve(b_tgt));
  double* d_src64 = kF64 ? static_cast<double*>(ar.carve(b_src64)) : nullptr;
  double* d_tgt64 = kF64 ? (shared ? d_src64 : static_cast<double*>(ar.carve(b_tgt64))) : nullptr;
  int64_t* d_so = static_cast<int64_t*>(ar.carve(b_off));
  int64_t* d_to = static_cast<int64_t*>(ar.carve(b_off));
  double* d_iT = static_cast<double*>(ar.carve(b_T));
  double* d_oT = static_cast<double*>(ar.carve(b_T));
  double* d_err = static_cast<double*>(ar.carve(sizeof(double) * S));
  int32_t* d_it = static_cast<int32_t*>(ar.carve(sizeof(int32_t) * S));
  int32_t* d_corr = out_corr ? static_cast<int32_t*>(ar.carve(b_corr)) : nullptr;

  hipStream_t st;
  if ((rc = thread_stream(&st)) != LSLAM_OK) return rc;
  LSLAM_HIP_TRY(hipMemcpyAsync(d_so, src_off, b_off, hipMemcpyHostToDevice, st));
  LSLAM_HIP_TRY(hipMemcpyAsync(d_to, tgt_off, b_off, hipMemcpyHostToDevice, st));
  LSLAM_HIP_TRY(hipMemcpyAsync(d_iT, init_T, b_T, hipMemcpyHostToDevice, st));
  // upload [lo, hi) points of one cloud array (and round it to float32 on the device in float64-exact mode)
  auto upload = [&](const PT* host, float* d32, double* d64, int64_t lo, int64_t hi, hipStream_t cs) -> int {
    if (hi <= lo) return LSLAM_OK;
    const size_t at = (size_t)dim * (size_t)lo, cnt = (size_t)dim * (size_t)(hi - lo);
    if constexpr (kF64) {
      LSLAM_HIP_TRY(hipMemcpyAsync(d64 + at, reinterpret_cast<const double*>(host) + at, sizeof(double) * cnt, hipMemcpyHostToDevice, cs));
      LSLAM_HIP_TRY(launch_f64_to_f32(d64 + at, d32 + at, (int64_t)cnt, cs));
    } else {
      LSLAM_HIP_TRY(hipMemcpyAsync(d32 + at, reinterpret_cast<const float*>(host) + at, sizeof(float) * cnt, hipMemcpyHostToDevice, cs));
    }
    return LSLAM_OK;
  };
  // Big batches are pipelined: the problems go in kChunks launches, and while one runs the points of the next are on
  // their way over PCIe (a second stream + one event per chunk).  Problems are independent (modules/localization.py:116-123),
  // so the results do not depend on the split.  (Offsets grow with the problem index: check_offsets above.)
  constexpr int kChunks = 4;
  // only batches that fill the chip several times over: every chunk is a launch of its own, and a launch of fewer problems
  // than the chip has room for (256 CUs x 1-3 workgroups) runs as long as its slowest problem (C1: 96 problems, 2.4x slower in 4 chunks)
  const bool pipelined = S >= 512 * kChunks && (b_src + b_tgt + b_src64 + b_tgt64) >= (size_t)(8u << 20) && !getenv("LSLAM_NO_UPLOAD_PIPELINE");
  if (!pipelined) {
    if ((rc = upload(src, d_src, d_src64, 0, up_src, st)) != LSLAM_OK) return rc;
    if (!shared && (rc = upload(tgt, d_tgt, d_tgt64, 0, tot_m, st)) != LSLAM_OK) return rc;
    rc = lslam_icp_batch_dev64(d_src, d_src64, d_so, d_tgt, d_tgt64, d_to, dim, d_iT, S, max_n, max_m, epsilon, stopping_thresh,
                               max_iters, flags, d_oT, d_err, d_it, d_corr, st);
    if (rc != LSLAM_OK) return rc;
  } else {
    static thread_local hipStream_t copy_stream = nullptr;
    static thread_local hipEvent_t landed[kChunks] = {};
    static thread_local int cs_dev = -1;
    if (!copy_stream || cs_dev != device) {
      LSLAM_HIP_TRY(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
      for (int c = 0; c < kChunks; ++c) LSLAM_HIP_TRY(hipEventCreateWithFlags(&landed[c], hipEventDisableTiming));
      cs_dev = device;
    }
    // the copy stream starts after everything this thread queued before (the arena may still be read by an earlier call)
    LSLAM_HIP_TRY(hipEventRecord(landed[0], st));
    LSLAM_HIP_TRY(hipStreamWaitEvent(copy_stream, landed[0], 0));
    int64_t up_s = shared ? std::min(src_off[0], tgt_off[0]) : src_off[0], up_t = tgt_off[0];      // uploaded so far (points)
    int32_t s0 = 0;
    for (int c = 0; c < kChunks; ++c) {
      const int32_t s1 = (c + 1 == kChunks) ? S : (int32_t)((int64_t)S * (c + 1) / kChunks);
      const int64_t need_s = shared ? std::max(src_off[s1], tgt_off[s1]) : src_off[s1], need_t = tgt_off[s1];
      if ((rc = upload(src, d_src, d_src64, up_s, need_s, copy_stream)) != LSLAM_OK) return rc;
      up_s = std::max(up_s, need_s);
      if (!shared) { if ((rc = upload(tgt, d_tgt, d_tgt64, up_t, need_t, copy_stream)) != LSLAM_OK) return rc; up_t = std::max(up_t, need_t); }
      LSLAM_HIP_TRY(hipEventRecord(landed[c], copy_stream));
      LSLAM_HIP_TRY(hipStreamWaitEvent(st, landed[c], 0));
      rc = lslam_icp_batch_dev64(d_src, d_src64, d_so + s0, d_tgt, d_tgt64, d_to + s0, dim, d_iT + 16 * (size_t)s0, s1 - s0, max_n, max_m, epsilon,
                                 stopping_thresh, max_iters, flags, d_oT + 16 * (size_t)s0, d_err + s0, d_it + s0, d_corr, st);
      if (rc != LSLAM_OK) return rc;
      s0 = s1;
    }
  }
  LSLAM_HIP_TRY(hipMemcpyAsync(out_T, d_oT, b_T, hipMemcpyDeviceToHost, st));
  LSLAM_HIP_TRY(hipMemcpyAsync(out_err, d_err, sizeof(double) * S, hipMemcpyDeviceToHost, st));
  LSLAM_HIP_TRY(hipMemcpyAsync(out_iters, d_it, sizeof(int32_t) * S, hipMemcpyDeviceToHost, st));
  if (out_corr) LSLAM_HIP_TRY(hipMemcpyAsync(out_corr, d_corr, b_corr, hipMemcpyDeviceToHost, st));
  // the task-queue kernel's can't-happen guard (waves that found no task for 4 s and left): results would be missing — say so
  unsigned gave_up = 0;
  unsigned* const qflags = prune_holder().sc.queue;
  if (qflags) LSLAM_HIP_TRY(hipMemcpyAsync(&gave_up, qflags + 50, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  LSLAM_HIP_TRY(hipStreamSynchronize(st));
  if (gave_up) {
    (void)hipMemset(qflags + 50, 0, sizeof(unsigned));
    return fail(LSLAM_ERR_HIP, "the pruned batch kernel gave up waiting for work (%u waves): results are incomplete", gave_up);
  }
  return LSLAM_OK;
}
}  // extern "C++"

int lslam_icp_batch(const float* src, const int64_t* src_off, const float* tgt, const int64_t* tgt_off,
                    int dim, const double* init_T, int32_t S, double epsilon, double stopping_thresh,
                    int32_t max_iters, uint32_t flags, double* out_T, double* out_err,
                    int32_t* out_iters, int32_t* out_corr, int device) {
  return icp_batch_host<float>(src, src_off, tgt, tgt_off, dim, init_T, S, epsilon, stopping_thresh, max_iters, flags,
                               out_T, out_err, out_iters, out_corr, device);
}

int lslam_icp_batch_f64(const double* src, const int64_t* src_off, const double* tgt, const int64_t* tgt_off,
                        int dim, const double* init_T, int32_t S, double epsilon, double stopping_thresh,
                        int32_t max_iters, uint32_t flags, double* out_T, double* out_err,
                        int32_t* out_iters, int32_t* out_corr, int device) {
  return icp_batch_host<double>(src, src_off, tgt, tgt_off, dim, init_T, S, epsilon, stopping_thresh, max_iters, flags,
                                out_T, out_err, out_iters, out_corr, device);
}

// ---------------------------------------------------------------------------------------------------
// Per-(thread, device) workspace of the large path: keys, partial rows, state, pinned flag mirror.
extern "C++" {
namespace {
struct LargeWs {
  int device = -1;
  unsigned long long* keys = nullptr; int64_t keys_cap = 0;
  double* partials = nullptr;
  LargeState* state = nullptr;
  LargeState* h_state = nullptr;       // pinned
  LargeState* h_back[2] = {nullptr, nullptr};   // pinned read-back slots of alternating polls
  hipEvent_t polled[2] = {nullptr, nullptr};
  float* sec = nullptr; int64_t sec_cap = 0;                  // float64-exact mode: second smallest distance per (split, source)
  float* bounds = nullptr;                                    // {max |t|², max |p|²}
  int ensure_sec(int64_t count) {
    if (!bounds) LSLAM_HIP_TRY(hipMalloc(&bounds, 16));
    if (count > sec_cap) {
      if (sec) { LSLAM_HIP_TRY(hipDeviceSynchronize()); (void)hipFree(sec); }
      sec = nullptr; sec_cap = 0;
      LSLAM_HIP_TRY(hipMalloc(&sec, sizeof(float) * (size_t)count));
      sec_cap = count;
    }
    return LSLAM_OK;
  }
  unsigned long long* gran = nullptr; size_t gran_cap = 0;   // moment rows of the co-resident workgroups (lslam_icp_coop.hip)
  unsigned* coop_failed = nullptr;                            // device word
  unsigned* h_failed = nullptr;                               // pinned
  int ensure_coop(size_t bytes) {
    if (!coop_failed) {
      LSLAM_HIP_TRY(hipMalloc(&coop_failed, 128));
      LSLAM_HIP_TRY(hipHostMalloc(&h_failed, 128));
    }
    if (bytes > gran_cap) {
      if (gran) { LSLAM_HIP_TRY(hipDeviceSynchronize()); (void)hipFree(gran); }
      gran = nullptr; gran_cap = 0;
      LSLAM_HIP_TRY(hipMalloc(&gran, bytes));
      gran_cap = bytes;
    }
    return LSLAM_OK;
  }
  int ensure(int dev, int64_t n) {
    if (dev != device) { release(); device = dev; }
    if (!partials) {
      LSLAM_HIP_TRY(hipMalloc(&partials, sizeof(double) * (size_t)kMomMaxBlocks * 32));
      LSLAM_HIP_TRY(hipMalloc(&state, sizeof(LargeState)));
      LSLAM_HIP_TRY(hipHostMalloc(&h_state, sizeof(LargeState)));
      for (int i = 0; i < 2; ++i) {
        LSLAM_HIP_TRY(hipHostMalloc(&h_back[i], sizeof(LargeState)));
        LSLAM_HIP_TRY(hipEventCreateWithFlags(&polled[i], hipEventDisableTiming));
      }
    }
    if (n > keys_cap) {
      if (keys) (void)hipFree(keys);
      keys = nullptr; keys_cap = 0;
      LSLAM_HIP_TRY(hipMalloc(&keys, sizeof(unsigned long long) * (size_t)n));
      keys_cap = n;
    }
    return LSLAM_OK;
  }
  void release() {
    if (keys) (void)hipFree(keys);
    if (partials) (void)hipFree(partials);
    if (state) (void)hipFree(state);
    if (h_state) (void)hipHostFree(h_state);
    if (gran) (void)hipFree(gran);
    if (sec) (void)hipFree(sec);
    if (bounds) (void)hipFree(bounds);
    sec = nullptr; sec_cap = 0; bounds = nullptr;
    if (coop_failed) (void)hipFree(coop_failed);
    if (h_failed) (void)hipHostFree(h_failed);
    gran = nullptr; gran_cap = 0; coop_failed = nullptr; h_failed = nullptr;
    for (int i = 0; i < 2; ++i) {
      if (h_back[i]) (void)hipHostFree(h_back[i]);
      if (polled[i]) (void)hipEventDestroy(polled[i]);
      h_back[i] = nullptr; polled[i] = nullptr;
    }
    keys = nullptr; partials = nullptr; state = nullptr; h_state = nullptr; keys_cap = 0;
  }
};
LargeWs& large_ws() { static thread_local LargeWs w; return w; }

// workspace of the pruned whole-chip path: one arena block carved into the LpWs arrays
struct LpHolder {
  int device = -1;
  void* block = nullptr; size_t cap = 0;
  int ensure(int dev, int dim, int64_t n, int64_t m, LpWs* ws) {
    const int64_t mx = n > m ? n : m;
    const size_t pt = dim == 2 ? sizeof(float2) : sizeof(float4);
    const int CH = dim == 2 ? NnChunk<2>::value : NnChunk<3>::value;
    const int64_t n1 = (m + CH - 1) / CH, n2 = (n1 + 63) / 64, n3 = (n2 + 63) / 64;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    ws->temp_bytes = lp_sort_temp_bytes(mx);
    const size_t need = al(pt * n) + al(4 * n) + al(pt * m) + al(4 * m) + 2 * (al(16 * n1) + al(16 * n2) + al(16 * n3)) + al(4 * n) +
                        4 * al(4 * mx) + al(4 * 6 * 1024) + al(64) + al(ws->temp_bytes) + al(64) + 2 * al(4 * n);
    if (dev != device || need > cap) {
      if (block) { (void)hipDeviceSynchronize(); (void)hipFree(block); }
      block = nullptr; cap = 0; device = dev;
      LSLAM_HIP_TRY(hipMalloc(&block, need + (need >> 3)));
      cap = need + (need >> 3);
    }
    char* p = static_cast<char*>(block);
    auto take = [&](size_t b) { void* r = p; p += al(b); return r; };
    ws->src_sorted = take(pt * n); ws->src_index = static_cast<int32_t*>(take(4 * n));
    ws->tgt_sorted = take(pt * m); ws->tgt_index = static_cast<int32_t*>(take(4 * m));
    ws->lo1 = static_cast<float4*>(take(16 * n1)); ws->hi1 = static_cast<float4*>(take(16 * n1));
    ws->lo2 = static_cast<float4*>(take(16 * n2)); ws->hi2 = static_cast<float4*>(take(16 * n2));
    ws->lo3 = static_cast<float4*>(take(16 * n3)); ws->hi3 = static_cast<float4*>(take(16 * n3));
    ws->match = static_cast<int32_t*>(take(4 * n));
    ws->keys_in = static_cast<unsigned*>(take(4 * mx)); ws->keys_out = static_cast<unsigned*>(take(4 * mx));
    ws->vals_in = static_cast<unsigned*>(take(4 * mx)); ws->vals_out = static_cast<unsigned*>(take(4 * mx));
    ws->bbox_partial = static_cast<float*>(take(4 * 6 * 1024));
    ws->grid = static_cast<float*>(take(64));
    ws->temp = take(ws->temp_bytes);
    ws->flag_count = static_cast<int32_t*>(take(64));
    ws->flag_src = static_cast<int32_t*>(take(4 * n));
    ws->flag_lim = static_cast<float*>(take(4 * n));
    return LSLAM_OK;
  }
};
LpHolder& lp_holder() { static thread_local LpHolder h; return h; }
void lp_release() { LpHolder& h = lp_holder(); if (h.block) (void)hipFree(h.block); h.block = nullptr; h.cap = 0; h.device = -1; }
}  // namespace
}  // extern "C++"

static int icp_large_dev_impl(const float* d_src, const double* d_src64, int64_t n, const float* d_tgt, const double* d_tgt64, int64_t m, int dim,
                              const double* init_T, double epsilon, double stopping_thresh, int32_t max_iters,
                              uint32_t flags, int32_t poll_every, double* out_T, double* out_err,
                              int32_t* out_iters, int32_t* d_out_corr, void* stream) {
  if (dim != 2 && dim != 3) return fail(LSLAM_ERR_BAD_ARG, "dim must be 2 or 3, got %d", dim);
  if (!d_src || !d_tgt || !init_T || !out_T || !out_err || !out_iters) return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  if (n <= 0) return fail(LSLAM_ERR_BAD_ARG, "empty source cloud");
  if (m <= 0) return fail(LSLAM_ERR_EMPTY_TARGET, "empty target cloud");
  if (n > 0x7fffffff || m > 0x7fffffff) return fail(LSLAM_ERR_BAD_ARG, "clouds above 2^31 points are not supported");
  if ((d_src64 == nullptr) != (d_tgt64 == nullptr)) return fail(LSLAM_ERR_BAD_ARG, "float64 clouds: pass both or neither");
  const bool f64 = d_src64 != nullptr;
  if (poll_every <= 0) poll_every = 8;
  int dev = 0;
  LSLAM_HIP_TRY(hipGetDevice(&dev));
  LargeWs& ws = large_ws();
  const bool prune = want_pruning(flags, (double)n * (double)m >= (dim == 2 ? kAutoPruneLargePairEvals2D : kAutoPruneLargePairEvals3D));
  const NnGeometry geo = nn_geometry(dim, n, m);
  int rc = ws.ensure(dev, prune ? 1 : n * geo.splits);
  if (rc != LSLAM_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);

  // ---- LiDAR-sized pair: the whole run inside ONE launch of co-resident workgroups (lslam_icp_coop.hip) -------------
  if (!prune && icp_coop_supported(dim, n, m, f64)) {
    if ((rc = ws.ensure_coop(icp_coop_gran_bytes(dim, n))) != LSLAM_OK) return rc;
    LargeState* hs0 = ws.h_state;
    memcpy(hs0->T, init_T, sizeof(double) * 16);
    hs0->err = 0.0;
    stop_init(hs0->stop);
    hs0->done = 0;
    hs0->ticket = 0;
    LSLAM_HIP_TRY(hipMemcpyAsync(ws.state, hs0, sizeof(LargeState), hipMemcpyHostToDevice, st));
    LSLAM_HIP_TRY(hipMemsetAsync(ws.coop_failed, 0, 16, st));
    CoopArgs ca{};
    ca.src = d_src; ca.n = (int32_t)n; ca.tgt = d_tgt; ca.m = (int32_t)m; ca.src64 = d_src64; ca.tgt64 = d_tgt64;
    ca.state = ws.state; ca.gran = ws.gran; ca.failed = ws.coop_failed; ca.out_corr = d_out_corr; ca.flags = flags;
    ca.epsilon = epsilon; ca.stopping_thresh = stopping_thresh; ca.max_iters = max_iters;
    LSLAM_HIP_TRY(launch_icp_coop(ca, dim, st));
    LSLAM_HIP_TRY(hipMemcpyAsync(ws.h_back[0], ws.state, sizeof(LargeState), hipMemcpyDeviceToHost, st));
    LSLAM_HIP_TRY(hipMemcpyAsync(ws.h_failed, ws.coop_failed, 128, hipMemcpyDeviceToHost, st));
    LSLAM_HIP_TRY(hipStreamSynchronize(st));
#ifdef LSLAM_COOP_PROFILE
    {
      const unsigned long long* pr = reinterpret_cast<const unsigned long long*>(ws.h_failed + 8);
      fprintf(stderr, "[lslam] coop phases (us, workgroup 0 thread 0, %d iterations): sweep %.2f  barrierA %.2f  slices+refine %.2f  moments %.2f  publish %.2f  gather %.2f  solve %.2f  barrierD %.2f\n",
              ws.h_back[0]->stop.executed, pr[0] * 0.01, pr[1] * 0.01, pr[2] * 0.01, pr[3] * 0.01, pr[4] * 0.01, pr[5] * 0.01, pr[6] * 0.01, pr[7] * 0.01);
    }
#endif
    if (!ws.h_failed[0]) {
      memcpy(out_T, ws.h_back[0]->T, sizeof(double) * 16);
      *out_err = ws.h_back[0]->err;
      *out_iters = ws.h_back[0]->stop.executed;
      return LSLAM_OK;
    }
    // A workgroup gave up waiting for the others (they never all became resident: the CUs were held by other work).  The
    // launch has drained; the problem is run again from init_T on the launch-per-iteration path below, which needs no
    // co-residency — same result, more launches.
    g_coop_fallbacks += 1;
    if (getenv("LSLAM_DEBUG")) fprintf(stderr, "[lslam] co-resident ICP launch timed out waiting for its workgroups: taking the launch-per-iteration path\n");
  }
  if (f64) {                                                // the near ties are decided by the moments kernel (all pairs) / a refine launch (pruned)
    if ((rc = ws.ensure_sec(prune ? 1 : n * geo.splits)) != LSLAM_OK) return rc;
    LSLAM_HIP_TRY(launch_cloud_bounds(d_src, n, d_tgt, m, dim, ws.bounds, st));
  }
  LpWs lp{};
  LpTree tree{};
  if (prune) {
    if ((rc = lp_holder().ensure(dev, dim, n, m, &lp)) != LSLAM_OK) return rc;
    LSLAM_HIP_TRY(lp_prepare(dim, d_src, n, d_tgt, m, init_T, lp, &tree, st));
  }

  LargeState* hs = ws.h_state;
  memcpy(hs->T, init_T, sizeof(double) * 16);
  hs->err = 0.0;
  stop_init(hs->stop);
  hs->done = 0;
  hs->ticket = 0;
  LSLAM_HIP_TRY(hipMemcpyAsync(ws.state, hs, sizeof(LargeState), hipMemcpyHostToDevice, st));

  NnArgs na{};
  na.src = d_src; na.n = n; na.tgt = d_tgt; na.m = m; na.keys = ws.keys; na.state = ws.state;
  bool fuse = !prune && !f64 && geo.P == 1;                 // small problem: one launch per iteration
  if (f64) { na.sec = ws.sec; na.bounds = ws.bounds; }
  if (const char* e = getenv("LSLAM_ICP_FUSE_TAIL")) fuse = fuse && atoi(e) != 0;
  if (fuse) {
    na.fuse_state = ws.state; na.out_corr = d_out_corr; na.splits = geo.splits; na.flags = flags;
    na.epsilon = epsilon; na.stopping_thresh = stopping_thresh; na.max_iters = max_iters;
    LSLAM_HIP_TRY(hipMemsetAsync(ws.keys, 0xff, sizeof(unsigned long long) * n, st));   // neutral keys for the atomic mins
  }
  MomArgs ma{};
  ma.src = d_src; ma.n = n; ma.tgt = d_tgt; ma.keys = ws.keys; ma.splits = geo.splits; ma.state = ws.state; ma.partials = ws.partials;
  ma.out_corr = d_out_corr; ma.flags = flags;
  if (f64) { ma.src64 = d_src64; ma.tgt64 = d_tgt64; ma.m = m; ma.sec = ws.sec; ma.bounds = ws.bounds; }
  const int rows = moments_rows(n);
  SolveArgs sa{};
  sa.state = ws.state; sa.partials = ws.partials; sa.rows = rows; sa.flags = flags;
  sa.epsilon = epsilon; sa.stopping_thresh = stopping_thresh; sa.max_iters = max_iters;

  // At most max_iters + 1 iterations run (icp.py:174).  Iterations are enqueued in groups of poll_every and the
  // state is read back after each group; from the second group on the NEXT group is already in the queue when the
  // host waits for a read-back, so the GPU never idles on a poll.  Launches past convergence return at once on the
  // `done` flag (≈ 3 µs each: a group of 8 queued in vain costs about what one idle poll would).
  const int64_t cap = (int64_t)max_iters + 1;
  int64_t launched = 0;
  auto enqueue_group = [&](int count, int slot) -> int {
    for (int k = 0; k < count && launched < cap; ++k, ++launched) {
      if (prune) {
        LpNnArgs pa{};
        pa.src_sorted = lp.src_sorted; pa.n = n; pa.tgt_sorted = lp.tgt_sorted; pa.tgt_index = lp.tgt_index; pa.m = m;
        pa.lo1 = lp.lo1; pa.hi1 = lp.hi1; pa.lo2 = lp.lo2; pa.hi2 = lp.hi2; pa.lo3 = lp.lo3; pa.hi3 = lp.hi3;
        pa.n1 = tree.n1; pa.n2 = tree.n2; pa.n3 = tree.n3; pa.match = lp.match; pa.state = ws.state;
        LpMomArgs pm{};
        pm.src_sorted = lp.src_sorted; pm.src_index = lp.src_index; pm.n = n; pm.tgt_sorted = lp.tgt_sorted; pm.tgt_index = lp.tgt_index;
        pm.match = lp.match; pm.state = ws.state; pm.partials = ws.partials; pm.out_corr = d_out_corr; pm.flags = flags;
        if (f64) {
          pa.bounds = ws.bounds; pa.flag_count = lp.flag_count; pa.flag_src = lp.flag_src; pa.flag_lim = lp.flag_lim;
          pm.src64 = d_src64; pm.tgt64 = d_tgt64;
        }
        LSLAM_HIP_TRY(lp_launch_nn(dim, pa, st));
        if (f64) {
          LpRefineArgs ra{};
          ra.src_sorted = lp.src_sorted; ra.src_index = lp.src_index; ra.tgt_sorted = lp.tgt_sorted; ra.tgt_index = lp.tgt_index; ra.m = m;
          ra.src64 = d_src64; ra.tgt64 = d_tgt64; ra.state = ws.state;
          ra.flag_count = lp.flag_count; ra.flag_src = lp.flag_src; ra.flag_lim = lp.flag_lim; ra.match = lp.match;
          ra.lo1 = lp.lo1; ra.hi1 = lp.hi1; ra.n1 = tree.n1;
          LSLAM_HIP_TRY(lp_launch_refine(dim, ra, st));
        }
        LSLAM_HIP_TRY(lp_launch_moments(dim, pm, rows, st));
      } else {
        LSLAM_HIP_TRY(launch_nn(na, dim, geo, st));
        if (fuse) continue;                                 // the search kernel's last workgroup did the rest
        LSLAM_HIP_TRY(launch_moments(ma, dim, rows, st));
      }
      LSLAM_HIP_TRY(launch_solve(sa, dim, st));
    }
    LSLAM_HIP_TRY(hipMemcpyAsync(ws.h_back[slot], ws.state, sizeof(LargeState), hipMemcpyDeviceToHost, st));
    LSLAM_HIP_TRY(hipEventRecord(ws.polled[slot], st));
    return LSLAM_OK;
  };
  // The first group is waited for on its own (most LiDAR-sized problems converge inside it, and launches queued
  // past convergence would delay the caller's next call); from the second group on the queue is kept one group ahead.
  int group = poll_every, slot = 0;
  if ((rc = enqueue_group(group, slot)) != LSLAM_OK) return rc;
  bool ahead = false;                                       // is a group already queued behind the one being waited for?
  for (;;) {
    LSLAM_HIP_TRY(hipEventSynchronize(ws.polled[slot]));
    hs = ws.h_back[slot];
    if (hs->done || (!ahead && launched >= cap)) break;
    if (!ahead) {                                           // first poll said "not yet": queue the next group now …
      if ((rc = enqueue_group(group, slot ^ 1)) != LSLAM_OK) return rc;
    }
    slot ^= 1;                                              // … that group is what we wait for next,
    ahead = launched < cap;
    if (ahead) {                                            // and one more goes behind it
      if ((rc = enqueue_group(group, slot ^ 1)) != LSLAM_OK) return rc;
    }
  }
  memcpy(out_T, hs->T, sizeof(double) * 16);
  *out_err = hs->err;
  *out_iters = hs->stop.executed;
  return LSLAM_OK;
}

int lslam_icp_large_dev64(const float* d_src, const double* d_src64, int64_t n, const float* d_tgt, const double* d_tgt64, int64_t m, int dim,
                          const double* init_T, double epsilon, double stopping_thresh, int32_t max_iters,
                          uint32_t flags, int32_t poll_every, double* out_T, double* out_err,
                          int32_t* out_iters, int32_t* d_out_corr, void* stream) {
  clear_error();
  int rc = ws_fence().enter(static_cast<hipStream_t>(stream));
  if (rc != LSLAM_OK) return rc;
  rc = icp_large_dev_impl(d_src, d_src64, n, d_tgt, d_tgt64, m, dim, init_T, epsilon, stopping_thresh, max_iters, flags, poll_every,
                          out_T, out_err, out_iters, d_out_corr, stream);
  const int rc2 = ws_fence().leave(static_cast<hipStream_t>(stream));
  return rc != LSLAM_OK ? rc : rc2;
}

int lslam_icp_large_dev(const float* d_src, int64_t n, const float* d_tgt, int64_t m, int dim,
                        const double* init_T, double epsilon, double stopping_thresh, int32_t max_iters,
                        uint32_t flags, int32_t poll_every, double* out_T, double* out_err,
                        int32_t* out_iters, int32_t* d_out_corr, void* stream) {
  return lslam_icp_large_dev64(d_src, nullptr, n, d_tgt, nullptr, m, dim, init_T, epsilon, stopping_thresh, max_iters, flags,
                               poll_every, out_T, out_err, out_iters, d_out_corr, stream);
}

static int nn_dev_impl(const float* d_src, int64_t n, const float* d_tgt, int64_t m, int dim, const double* T,
                       int32_t* d_out_idx, float* d_out_d2, void* stream) {
  if (dim != 2 && dim != 3) return fail(LSLAM_ERR_BAD_ARG, "dim must be 2 or 3, got %d", dim);
  if (n < 0) return fail(LSLAM_ERR_BAD_ARG, "negative source size");
  if (n == 0) return LSLAM_OK;
  if (!d_src || !d_tgt || !d_out_idx) return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  if (m <= 0) return fail(LSLAM_ERR_EMPTY_TARGET, "empty target cloud");
  if (n > 0x7fffffff || m > 0x7fffffff) return fail(LSLAM_ERR_BAD_ARG, "clouds above 2^31 points are not supported");
  int dev = 0;
  LSLAM_HIP_TRY(hipGetDevice(&dev));
  LargeWs& ws = large_ws();
  const NnGeometry geo = nn_geometry(dim, n, m);
  int rc = ws.ensure(dev, n * geo.splits);
  if (rc != LSLAM_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  NnArgs na{};
  na.src = d_src; na.n = n; na.tgt = d_tgt; na.m = m; na.keys = ws.keys; na.state = nullptr;
  static const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  memcpy(na.T, T ? T : I4, sizeof(double) * 16);
  LSLAM_HIP_TRY(launch_nn(na, dim, geo, st));
  LSLAM_HIP_TRY(launch_nn_finalize(ws.keys, n, geo.splits, d_out_idx, d_out_d2, st));
  return LSLAM_OK;
}

int lslam_nn_dev(const float* d_src, int64_t n, const float* d_tgt, int64_t m, int dim, const double* T,
                 int32_t* d_out_idx, float* d_out_d2, void* stream) {
  clear_error();
  int rc = ws_fence().enter(static_cast<hipStream_t>(stream));
  if (rc != LSLAM_OK) return rc;
  rc = nn_dev_impl(d_src, n, d_tgt, m, dim, T, d_out_idx, d_out_d2, stream);
  const int rc2 = ws_fence().leave(static_cast<hipStream_t>(stream));
  return rc != LSLAM_OK ? rc : rc2;
}

// host-pointer wrappers of the two entry points above
static int upload_pair(const float* src, int64_t n, const float* tgt, int64_t m, int dim, int device,
                       size_t extra, float** d_src, float** d_tgt, void** d_extra, hipStream_t* st) {
  if (!all_finite(src, dim * n) || !all_finite(tgt, dim * m))
    return fail(LSLAM_ERR_NONFINITE, "point clouds must be finite (NaN or Inf found)");
  int rc;
  if ((rc = use_device(device)) != LSLAM_OK) return rc;
  Arena& ar = thread_arena();
  const size_t bs = sizeof(float) * dim * (size_t)n, bt = sizeof(float) * dim * (size_t)m;
  if ((rc = ar.reserve(device, Arena::align(bs) + Arena::align(bt) + Arena::align(extra))) != LSLAM_OK) return rc;
  *d_src = static_cast<float*>(ar.carve(bs));
  *d_tgt = static_cast<float*>(ar.carve(bt));
  *d_extra = extra ? ar.carve(extra) : nullptr;
  if ((rc = thread_stream(st)) != LSLAM_OK) return rc;
  LSLAM_HIP_TRY(hipMemcpyAsync(*d_src, src, bs, hipMemcpyHostToDevice, *st));
  LSLAM_HIP_TRY(hipMemcpyAsync(*d_tgt, tgt, bt, hipMemcpyHostToDevice, *st));
  return LSLAM_OK;
}

int lslam_icp_large(const float* src, int64_t n, const float* tgt, int64_t m, int dim, const double* init_T,
                    double epsilon, double stopping_thresh, int32_t max_iters, uint32_t flags,
                    double* out_T, double* out_err, int32_t* out_iters, int32_t* out_corr, int device) {
  clear_error();
  if (dim != 2 && dim != 3) return fail(LSLAM_ERR_BAD_ARG, "dim must be 2 or 3, got %d", dim);
  if (!src || !tgt) return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  if (n <= 0) return fail(LSLAM_ERR_BAD_ARG, "empty source cloud");
  if (m <= 0) return fail(LSLAM_ERR_EMPTY_TARGET, "empty target cloud");
  float *d_src, *d_tgt; void* d_corr; hipStream_t st;
  int rc = upload_pair(src, n, tgt, m, dim, device, out_corr ? sizeof(int32_t) * (size_t)n : 0, &d_src, &d_tgt, &d_corr, &st);
  if (rc != LSLAM_OK) return rc;
  rc = lslam_icp_large_dev(d_src, n, d_tgt, m, dim, init_T, epsilon, stopping_thresh, max_iters, flags, 8,
                           out_T, out_err, out_iters, static_cast<int32_t*>(d_corr), st);
  if (rc != LSLAM_OK) return rc;
  if (out_corr) {
    LSLAM_HIP_TRY(hipMemcpyAsync(out_corr, d_corr, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
    LSLAM_HIP_TRY(hipStreamSynchronize(st));
  }
  return LSLAM_OK;
}

int lslam_icp_large_f64(const double* src, int64_t n, const double* tgt, int64_t m, int dim, const double* init_T,
                        double epsilon, double stopping_thresh, int32_t max_iters, uint32_t flags,
                        double* out_T, double* out_err, int32_t* out_iters, int32_t* out_corr, int device) {
  clear_error();
  if (dim != 2 && dim != 3) return fail(LSLAM_ERR_BAD_ARG, "dim must be 2 or 3, got %d", dim);
  if (!src || !tgt) return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  if (n <= 0) return fail(LSLAM_ERR_BAD_ARG, "empty source cloud");
  if (m <= 0) return fail(LSLAM_ERR_EMPTY_TARGET, "empty target cloud");
  if (!all_finite64(src, dim * n) || !all_finite64(tgt, dim * m))
    return fail(LSLAM_ERR_NONFINITE, "point clouds must be finite (NaN or Inf found)");
  int rc;
  if ((rc = use_device(device)) != LSLAM_OK) return rc;
  Arena& ar = thread_arena();
  const size_t cs = (size_t)dim * (size_t)n, ct = (size_t)dim * (size_t)m;
  const size_t b_corr = out_corr ? sizeof(int32_t) * (size_t)n : 0;
  if ((rc = ar.reserve(device, Arena::align(4 * cs) + Arena::align(4 * ct) + Arena::align(8 * cs) + Arena::align(8 * ct) + Arena::align(b_corr))) != LSLAM_OK) return rc;
  float* d_src = static_cast<float*>(ar.carve(4 * cs));
  float* d_tgt = static_cast<float*>(ar.carve(4 * ct));
  double* d_src64 = static_cast<double*>(ar.carve(8 * cs));
  double* d_tgt64 = static_cast<double*>(ar.carve(8 * ct));
  int32_t* d_corr = out_corr ? static_cast<int32_t*>(ar.carve(b_corr)) : nullptr;
  hipStream_t st;
  if ((rc = thread_stream(&st)) != LSLAM_OK) return rc;
  LSLAM_HIP_TRY(hipMemcpyAsync(d_src64, src, 8 * cs, hipMemcpyHostToDevice, st));
  LSLAM_HIP_TRY(hipMemcpyAsync(d_tgt64, tgt, 8 * ct, hipMemcpyHostToDevice, st));
  LSLAM_HIP_TRY(launch_f64_to_f32(d_src64, d_src, (int64_t)cs, st));
  LSLAM_HIP_TRY(launch_f64_to_f32(d_tgt64, d_tgt, (int64_t)ct, st));
  rc = lslam_icp_large_dev64(d_src, d_src64, n, d_tgt, d_tgt64, m, dim, init_T, epsilon, stopping_thresh, max_iters, flags, 8,
                             out_T, out_err, out_iters, d_corr, st);
  if (rc != LSLAM_OK) return rc;
  if (out_corr) {
    LSLAM_HIP_TRY(hipMemcpyAsync(out_corr, d_corr, b_corr, hipMemcpyDeviceToHost, st));
    LSLAM_HIP_TRY(hipStreamSynchronize(st));
  }
  return LSLAM_OK;
}

// One pair, S start transforms (the reference's 24-yaw multi-start loop, code/icp_warm_up/test_icp.py:70-88, on clouds too large for
// the batched kernels): both clouds are uploaded ONCE and the S runs are enqueued back to back on the device-resident pair.
extern "C++" {
template <typename PT>
static int icp_large_starts_host(const PT* src, int64_t n, const PT* tgt, int64_t m, int dim, const double* init_T, int32_t S,
                                 double epsilon, double stopping_thresh, int32_t max_iters, uint32_t flags,
                                 double* out_T, double* out_err, int32_t* out_iters, int device) {
  constexpr bool kF64 = sizeof(PT) == sizeof(double);
  clear_error();
  if (dim != 2 && dim != 3) return fail(LSLAM_ERR_BAD_ARG, "dim must be 2 or 3, got %d", dim);
  if (S < 0) return fail(LSLAM_ERR_BAD_ARG, "negative number of starts");
  if (S == 0) return LSLAM_OK;
  if (!src || !tgt || !init_T || !out_T || !out_err || !out_iters) return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  if (n <= 0) return fail(LSLAM_ERR_BAD_ARG, "empty source cloud");
  if (m <= 0) return fail(LSLAM_ERR_EMPTY_TARGET, "empty target cloud");
  bool finite;
  if constexpr (kF64) finite = all_finite64(src, dim * n) && all_finite64(tgt, dim * m);
  else finite = all_finite(src, dim * n) && all_finite(tgt, dim * m);
  if (!finite) return fail(LSLAM_ERR_NONFINITE, "point clouds must be finite (NaN or Inf found)");
  int rc;
  if ((rc = use_device(device)) != LSLAM_OK) return rc;
  Arena& ar = thread_arena();
  const size_t cs = (size_t)dim * (size_t)n, ct = (size_t)dim * (size_t)m;
  if ((rc = ar.reserve(device, Arena::align(4 * cs) + Arena::align(4 * ct) + (kF64 ? Arena::align(8 * cs) + Arena::align(8 * ct) : 0))) != LSLAM_OK) return rc;
  float* d_src = static_cast<float*>(ar.carve(4 * cs));
  float* d_tgt = static_cast<float*>(ar.carve(4 * ct));
  double* d_src64 = kF64 ? static_cast<double*>(ar.carve(8 * cs)) : nullptr;
  double* d_tgt64 = kF64 ? static_cast<double*>(ar.carve(8 * ct)) : nullptr;
  hipStream_t st;
  if ((rc = thread_stream(&st)) != LSLAM_OK) return rc;
  if constexpr (kF64) {
    LSLAM_HIP_TRY(hipMemcpyAsync(d_src64, src, 8 * cs, hipMemcpyHostToDevice, st));
    LSLAM_HIP_TRY(hipMemcpyAsync(d_tgt64, tgt, 8 * ct, hipMemcpyHostToDevice, st));
    LSLAM_HIP_TRY(launch_f64_to_f32(d_src64, d_src, (int64_t)cs, st));
    LSLAM_HIP_TRY(launch_f64_to_f32(d_tgt64, d_tgt, (int64_t)ct, st));
  } else {
    LSLAM_HIP_TRY(hipMemcpyAsync(d_src, src, 4 * cs, hipMemcpyHostToDevice, st));
    LSLAM_HIP_TRY(hipMemcpyAsync(d_tgt, tgt, 4 * ct, hipMemcpyHostToDevice, st));
  }
  for (int32_t k = 0; k < S; ++k) {
    rc = lslam_icp_large_dev64(d_src, d_src64, n, d_tgt, d_tgt64, m, dim, init_T + 16 * (size_t)k, epsilon, stopping_thresh, max_iters, flags, 8,
                               out_T + 16 * (size_t)k, out_err + k, out_iters + k, nullptr, st);
    if (rc != LSLAM_OK) return rc;
  }
  return LSLAM_OK;
}
}  // extern "C++"

int lslam_icp_large_starts(const float* src, int64_t n, const float* tgt, int64_t m, int dim, const double* init_T, int32_t S,
                           double epsilon, double stopping_thresh, int32_t max_iters, uint32_t flags,
                           double* out_T, double* out_err, int32_t* out_iters, int device) {
  return icp_large_starts_host<float>(src, n, tgt, m, dim, init_T, S, epsilon, stopping_thresh, max_iters, flags, out_T, out_err, out_iters, device);
}
int lslam_icp_large_starts_f64(const double* src, int64_t n, const double* tgt, int64_t m, int dim, const double* init_T, int32_t S,
                               double epsilon, double stopping_thresh, int32_t max_iters, uint32_t flags,
                               double* out_T, double* out_err, int32_t* out_iters, int device) {
  return icp_large_starts_host<double>(src, n, tgt, m, dim, init_T, S, epsilon, stopping_thresh, max_iters, flags, out_T, out_err, out_iters, device);
}

int lslam_nn(const float* src, int64_t n, const float* tgt, int64_t m, int dim, const double* T,
             int32_t* out_idx, float* out_d2, int device) {
  clear_error();
  if (dim != 2 && dim != 3) return fail(LSLAM_ERR_BAD_ARG, "dim must be 2 or 3, got %d", dim);
  if (n < 0) return fail(LSLAM_ERR_BAD_ARG, "negative source size");
  if (n == 0) return LSLAM_OK;
  if (!src || !tgt || !out_idx) return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  if (m <= 0) return fail(LSLAM_ERR_EMPTY_TARGET, "empty target cloud");
  float *d_src, *d_tgt; void* d_out; hipStream_t st;
  const size_t bi = Arena::align(sizeof(int32_t) * (size_t)n);
  int rc = upload_pair(src, n, tgt, m, dim, device, bi + sizeof(float) * (size_t)n, &d_src, &d_tgt, &d_out, &st);
  if (rc != LSLAM_OK) return rc;
  int32_t* d_idx = static_cast<int32_t*>(d_out);
  float* d_d2 = reinterpret_cast<float*>(static_cast<char*>(d_out) + bi);
  rc = lslam_nn_dev(d_src, n, d_tgt, m, dim, T, d_idx, out_d2 ? d_d2 : nullptr, st);
  if (rc != LSLAM_OK) return rc;
  LSLAM_HIP_TRY(hipMemcpyAsync(out_idx, d_idx, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
  if (out_d2) LSLAM_HIP_TRY(hipMemcpyAsync(out_d2, d_d2, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, st));
  LSLAM_HIP_TRY(hipStreamSynchronize(st));
  return LSLAM_OK;
}

// get_correspondences in float64-exact mode: the whole-chip search with the splits' second smallest distances, then the
// float64-exact moments kernel used for its correspondences only (it decides the near ties on the float64 points)
int lslam_nn_f64(const double* src, int64_t n, const double* tgt, int64_t m, int dim, const double* T,
                 int32_t* out_idx, int device) {
  clear_error();
  if (dim != 2 && dim != 3) return fail(LSLAM_ERR_BAD_ARG, "dim must be 2 or 3, got %d", dim);
  if (n < 0) return fail(LSLAM_ERR_BAD_ARG, "negative source size");
  if (n == 0) return LSLAM_OK;
  if (!src || !tgt || !out_idx) return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  if (m <= 0) return fail(LSLAM_ERR_EMPTY_TARGET, "empty target cloud");
  if (n > 0x7fffffff || m > 0x7fffffff) return fail(LSLAM_ERR_BAD_ARG, "clouds above 2^31 points are not supported");
  if (!all_finite64(src, dim * n) || !all_finite64(tgt, dim * m))
    return fail(LSLAM_ERR_NONFINITE, "point clouds must be finite (NaN or Inf found)");
  int rc;
  if ((rc = use_device(device)) != LSLAM_OK) return rc;
  Arena& ar = thread_arena();
  const size_t cs = (size_t)dim * (size_t)n, ct = (size_t)dim * (size_t)m;
  if ((rc = ar.reserve(device, Arena::align(4 * cs) + Arena::align(4 * ct) + Arena::align(8 * cs) + Arena::align(8 * ct) + Arena::align(4 * (size_t)n))) != LSLAM_OK) return rc;
  float* d_src = static_cast<float*>(ar.carve(4 * cs));
  float* d_tgt = static_cast<float*>(ar.carve(4 * ct));
  double* d_src64 = static_cast<double*>(ar.carve(8 * cs));
  double* d_tgt64 = static_cast<double*>(ar.carve(8 * ct));
  int32_t* d_idx = static_cast<int32_t*>(ar.carve(4 * (size_t)n));
  hipStream_t st;
  if ((rc = thread_stream(&st)) != LSLAM_OK) return rc;
  if ((rc = ws_fence().enter(st)) != LSLAM_OK) return rc;
  LargeWs& ws = large_ws();
  const NnGeometry geo = nn_geometry(dim, n, m);
  if ((rc = ws.ensure(device, n * geo.splits)) != LSLAM_OK) return rc;
  if ((rc = ws.ensure_sec(n * geo.splits)) != LSLAM_OK) return rc;
  LSLAM_HIP_TRY(hipMemcpyAsync(d_src64, src, 8 * cs, hipMemcpyHostToDevice, st));
  LSLAM_HIP_TRY(hipMemcpyAsync(d_tgt64, tgt, 8 * ct, hipMemcpyHostToDevice, st));
  LSLAM_HIP_TRY(launch_f64_to_f32(d_src64, d_src, (int64_t)cs, st));
  LSLAM_HIP_TRY(launch_f64_to_f32(d_tgt64, d_tgt, (int64_t)ct, st));
  LSLAM_HIP_TRY(launch_cloud_bounds(d_src, n, d_tgt, m, dim, ws.bounds, st));
  static const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  LargeState* hs = ws.h_state;
  memcpy(hs->T, T ? T : I4, sizeof(double) * 16);
  hs->err = 0.0; stop_init(hs->stop); hs->done = 0; hs->ticket = 0;
  LSLAM_HIP_TRY(hipMemcpyAsync(ws.state, hs, sizeof(LargeState), hipMemcpyHostToDevice, st));
  NnArgs na{};
  na.src = d_src; na.n = n; na.tgt = d_tgt; na.m = m; na.keys = ws.keys; na.state = ws.state; na.sec = ws.sec; na.bounds = ws.bounds;
  LSLAM_HIP_TRY(launch_nn(na, dim, geo, st));
  MomArgs ma{};
  ma.src = d_src; ma.n = n; ma.tgt = d_tgt; ma.keys = ws.keys; ma.splits = geo.splits; ma.state = ws.state; ma.partials = ws.partials;
  ma.out_corr = d_idx; ma.flags = 0; ma.src64 = d_src64; ma.tgt64 = d_tgt64; ma.m = m; ma.sec = ws.sec; ma.bounds = ws.bounds;
  LSLAM_HIP_TRY(launch_moments(ma, dim, moments_rows(n), st));
  LSLAM_HIP_TRY(hipMemcpyAsync(out_idx, d_idx, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
  rc = ws_fence().leave(st);
  LSLAM_HIP_TRY(hipStreamSynchronize(st));
  return rc;
}

// ---------------------------------------------------------------------------------------------------
static int pair_op(const double* p, int64_t ps, const double* q, int64_t qs, int64_t n, int32_t cols,
                   int32_t normalize, double* out_T, double* out_err, int device) {
  int rc;
  if ((rc = use_device(device)) != LSLAM_OK) return rc;
  Arena& ar = thread_arena();
  const size_t bp = sizeof(double) * (size_t)(n > 0 ? (n - 1) * ps + cols : 0);
  const size_t bq = sizeof(double) * (size_t)(n > 0 ? (n - 1) * qs + cols : 0);
  if ((rc = ar.reserve(device, Arena::align(bp) + Arena::align(bq) + 512)) != LSLAM_OK) return rc;
  double* dp = static_cast<double*>(ar.carve(bp));
  double* dq = static_cast<double*>(ar.carve(bq));
  double* dT = static_cast<double*>(ar.carve(sizeof(double) * 16));
  double* de = static_cast<double*>(ar.carve(sizeof(double)));
  hipStream_t st;
  if ((rc = thread_stream(&st)) != LSLAM_OK) return rc;
  if (bp) LSLAM_HIP_TRY(hipMemcpyAsync(dp, p, bp, hipMemcpyHostToDevice, st));
  if (bq) LSLAM_HIP_TRY(hipMemcpyAsync(dq, q, bq, hipMemcpyHostToDevice, st));
  PairOpArgs a{};
  a.p = dp; a.p_stride = ps; a.q = dq; a.q_stride = qs; a.n = n; a.cols = cols; a.normalize = normalize;
  a.out_T = out_T ? dT : nullptr; a.out_err = out_err ? de : nullptr;
  LSLAM_HIP_TRY(launch_pair_op(a, st));
  if (out_T) LSLAM_HIP_TRY(hipMemcpyAsync(out_T, dT, sizeof(double) * 16, hipMemcpyDeviceToHost, st));
  if (out_err) LSLAM_HIP_TRY(hipMemcpyAsync(out_err, de, sizeof(double), hipMemcpyDeviceToHost, st));
  LSLAM_HIP_TRY(hipStreamSynchronize(st));
  return LSLAM_OK;
}

int lslam_kabsch(const double* p, int64_t p_stride, const double* q, int64_t q_stride, int64_t n,
                 double* out_T, int device) {
  clear_error();
  if (!p || !q || !out_T) return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  if (n < 0 || p_stride < 3 || q_stride < 3) return fail(LSLAM_ERR_BAD_ARG, "need n >= 0 and row strides >= 3");
  return pair_op(p, p_stride, q, q_stride, n, 3, 0, out_T, nullptr, device);
}

int lslam_sq_error(const double* p, const double* q, int64_t n, int32_t cols, int32_t normalize,
                   double* out_err, int device) {
  clear_error();
  if (!p || !q || !out_err) return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  if (n < 0 || cols < 1 || cols > 3) return fail(LSLAM_ERR_BAD_ARG, "need n >= 0 and 1 <= cols <= 3");
  return pair_op(p, cols, q, cols, n, cols, normalize, nullptr, out_err, device);
}

// ---------------------------------------------------------------------------------------------------
int lslam_voxel_downsample(const double* points, int64_t n, int32_t dim, double voxel_size, double* out,
                           int64_t* out_count, int device) {
  clear_error();
  if (!out_count) return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  *out_count = 0;
  if (n < 0 || (dim != 2 && dim != 3) || !(voxel_size > 0.0)) return fail(LSLAM_ERR_BAD_ARG, "need n >= 0, dim in {2,3}, voxel_size > 0");
  if (n == 0) return LSLAM_OK;
  if (!points || !out) return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  if (n > 0xfffffff0ll) return fail(LSLAM_ERR_BAD_ARG, "more than 2^32 points are not supported");
  int rc;
  if ((rc = use_device(device)) != LSLAM_OK) return rc;
  hipStream_t st;
  if ((rc = thread_stream(&st)) != LSLAM_OK) return rc;
  const size_t b_pts = sizeof(double) * (size_t)n * dim;
  VoxelWs ws{};
  ws.temp_bytes = voxel_temp_bytes(n);
  Arena& ar = thread_arena();
  const size_t need = 2 * Arena::align(b_pts) + Arena::align(sizeof(double) * 3 * 1024) + 2 * Arena::align(64) +
                      2 * Arena::align(sizeof(unsigned long long) * (size_t)n) + 2 * Arena::align(sizeof(uint32_t) * (size_t)n) +
                      Arena::align(sizeof(int64_t) * ((size_t)n + 1)) + Arena::align(ws.temp_bytes);
  if ((rc = ar.reserve(device, need)) != LSLAM_OK) return rc;
  double* d_pts = static_cast<double*>(ar.carve(b_pts));
  double* d_out = static_cast<double*>(ar.carve(b_pts));
  ws.partial = static_cast<double*>(ar.carve(sizeof(double) * 3 * 1024));
  ws.lo = static_cast<double*>(ar.carve(64));
  ws.bad = static_cast<int*>(ar.carve(64));
  ws.keys_in = static_cast<unsigned long long*>(ar.carve(sizeof(unsigned long long) * (size_t)n));
  ws.keys_out = static_cast<unsigned long long*>(ar.carve(sizeof(unsigned long long) * (size_t)n));
  ws.ids_in = static_cast<uint32_t*>(ar.carve(sizeof(uint32_t) * (size_t)n));
  ws.ids_out = static_cast<uint32_t*>(ar.carve(sizeof(uint32_t) * (size_t)n));
  ws.starts = static_cast<int64_t*>(ar.carve(sizeof(int64_t) * ((size_t)n + 1)));
  ws.temp = ar.carve(ws.temp_bytes);
  LSLAM_HIP_TRY(hipMemcpyAsync(d_pts, points, b_pts, hipMemcpyHostToDevice, st));
  int64_t nvox = 0;
  int bad = 0;
  LSLAM_HIP_TRY(run_voxel_downsample(d_pts, n, dim, voxel_size, ws, d_out, &nvox, &bad, st));
  if (bad) return fail(LSLAM_ERR_BAD_ARG, "voxel grid exceeds 2^21 cells along an axis (or non-finite points)");
  LSLAM_HIP_TRY(hipMemcpyAsync(out, d_out, sizeof(double) * (size_t)nvox * dim, hipMemcpyDeviceToHost, st));
  LSLAM_HIP_TRY(hipStreamSynchronize(st));
  *out_count = nvox;
  return LSLAM_OK;
}

// ---------------------------------------------------------------------------------------------------
int lslam_chain_se2(const double* rel, int64_t S, double* out) {
  clear_error();
  if (S < 0 || !out || (S > 0 && !rel)) return fail(LSLAM_ERR_BAD_ARG, "bad arguments");
  double T[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  out[0] = out[1] = out[2] = 0.0;
  for (int64_t k = 0; k < S; ++k) {
    const double* R = rel + 9 * k;
    double N[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) N[3 * i + j] = T[3 * i] * R[j] + T[3 * i + 1] * R[3 + j] + T[3 * i + 2] * R[6 + j];
    for (int i = 0; i < 9; ++i) T[i] = N[i];
    out[3 * (k + 1)] = T[2];
    out[3 * (k + 1) + 1] = T[5];
    out[3 * (k + 1) + 2] = atan2(T[3], T[0]);
  }
  return LSLAM_OK;
}

// ---------------------------------------------------------------------------------------------------
// Shared by lslam_lidar_preprocess (results to the host) and lslam_scans_create (results stay on the
// device): runs the selected stages; out_off (host, K+1) is always filled; the final points are copied
// to out_xy if it is non-null; *d_final / *st_out expose the device copy (valid until the thread's
// arena is reused) and the stream it was produced on.
static int preprocess_core(uint32_t stages, const double* ranges, int32_t K, int32_t B, const double* cos_a,
                           const double* sin_a, double off_x, double off_y, double range_min, double range_max,
                           const double* in_xy, const int64_t* in_off, double eps, int32_t min_samples, double k_std,
                           double* out_xy, int64_t* out_off, double* out_stats, int device, bool need_host_xy,
                           const double** d_final, hipStream_t* st_out) {
  if (K < 0) return fail(LSLAM_ERR_BAD_ARG, "negative scan count");
  if (!out_off) return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  const bool polar = (stages & LSLAM_PRE_POLAR) != 0;
  int64_t total = 0, max_n = 0;
  if (polar) {
    if (B <= 0 || !ranges || !cos_a || !sin_a) return fail(LSLAM_ERR_BAD_ARG, "polar stage needs ranges, cos_a, sin_a and B > 0");
    total = (int64_t)K * B;
    max_n = B;
  } else {
    if (!in_off) return fail(LSLAM_ERR_BAD_ARG, "null input offsets");
    if (in_off[0] != 0) return fail(LSLAM_ERR_BAD_ARG, "in_off must start at 0");
    for (int32_t k = 0; k < K; ++k) {
      const int64_t n = in_off[k + 1] - in_off[k];
      if (n < 0 || n > 0x7fffffff) return fail(LSLAM_ERR_BAD_ARG, "in_off is not monotone at %d", k);
      if (n > max_n) max_n = n;
    }
    total = in_off[K];
    if (total > 0 && !in_xy) return fail(LSLAM_ERR_BAD_ARG, "null input points");
  }
  if (d_final) *d_final = nullptr;
  if (K == 0) { out_off[0] = 0; return LSLAM_OK; }
  if (need_host_xy && total > 0 && !out_xy) return fail(LSLAM_ERR_BAD_ARG, "null output buffer");
  int rc;
  if ((rc = use_device(device)) != LSLAM_OK) return rc;
  hipStream_t st;
  if ((rc = thread_stream(&st)) != LSLAM_OK) return rc;
  if (st_out) *st_out = st;

  // device scratch: two point buffers (ping-pong), two offset arrays, masks, counts, statistics
  const size_t nb = (size_t)(B > 0 ? B : 1);
  const size_t b_pts = sizeof(double) * 2 * (size_t)(total > 0 ? total : 1);
  const size_t b_off = sizeof(int64_t) * ((size_t)K + 1);
  Arena& ar = thread_arena();
  const size_t need = Arena::align(sizeof(double) * (size_t)(total > 0 ? total : 1)) + 2 * Arena::align(sizeof(double) * nb) +
                      2 * Arena::align(b_pts) + 2 * Arena::align(b_off) + 2 * Arena::align((size_t)total + 16) +
                      Arena::align(sizeof(int32_t) * (size_t)K) + Arena::align(sizeof(double) * (kRadialParts + 8));
  if ((rc = ar.reserve(device, need)) != LSLAM_OK) return rc;
  double* d_ranges = static_cast<double*>(ar.carve(sizeof(double) * (size_t)(total > 0 ? total : 1)));
  double* d_cos = static_cast<double*>(ar.carve(sizeof(double) * nb));
  double* d_sin = static_cast<double*>(ar.carve(sizeof(double) * nb));
  double* d_pts[2] = {static_cast<double*>(ar.carve(b_pts)), static_cast<double*>(ar.carve(b_pts))};
  int64_t* d_off[2] = {static_cast<int64_t*>(ar.carve(b_off)), static_cast<int64_t*>(ar.carve(b_off))};
  uint8_t* d_keep = static_cast<uint8_t*>(ar.carve((size_t)total + 16));
  uint8_t* d_core = static_cast<uint8_t*>(ar.carve((size_t)total + 16));
  int32_t* d_cnt = static_cast<int32_t*>(ar.carve(sizeof(int32_t) * (size_t)K));
  double* d_partial = static_cast<double*>(ar.carve(sizeof(double) * (kRadialParts + 8)));
  double* d_stats = d_partial + kRadialParts;

  int cur = 0;            // d_pts[cur] / d_off[cur] hold the current CSR
  if (polar) {
    LSLAM_HIP_TRY(hipMemcpyAsync(d_ranges, ranges, sizeof(double) * (size_t)total, hipMemcpyHostToDevice, st));
    LSLAM_HIP_TRY(hipMemcpyAsync(d_cos, cos_a, sizeof(double) * B, hipMemcpyHostToDevice, st));
    LSLAM_HIP_TRY(hipMemcpyAsync(d_sin, sin_a, sizeof(double) * B, hipMemcpyHostToDevice, st));
    LSLAM_HIP_TRY(launch_polar_mask(d_ranges, K, B, d_cos, d_sin, off_x, off_y, range_min, range_max, d_pts[1], d_keep, d_cnt, st));
    LSLAM_HIP_TRY(launch_compact(d_pts[1], nullptr, B, d_keep, d_cnt, K, d_off[0], d_pts[0], st));
  } else {
    if (total > 0) LSLAM_HIP_TRY(hipMemcpyAsync(d_pts[0], in_xy, sizeof(double) * 2 * (size_t)total, hipMemcpyHostToDevice, st));
    LSLAM_HIP_TRY(hipMemcpyAsync(d_off[0], in_off, b_off, hipMemcpyHostToDevice, st));
  }
  if (stages & LSLAM_PRE_DBSCAN) {
    LSLAM_HIP_TRY(launch_dbscan_mask(d_pts[cur], d_off[cur], K, max_n, eps, min_samples, d_core, d_keep, d_cnt, st));
    LSLAM_HIP_TRY(launch_compact(d_pts[cur], d_off[cur], 0, d_keep, d_cnt, K, d_off[cur ^ 1], d_pts[cur ^ 1], st));
    cur ^= 1;
  }
  if (stages & LSLAM_PRE_RADIAL) {
    // the statistics run over the points that are left: their number is on the device
    int64_t left = 0;
    LSLAM_HIP_TRY(hipMemcpyAsync(&left, d_off[cur] + K, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    LSLAM_HIP_TRY(hipStreamSynchronize(st));
    LSLAM_HIP_TRY(launch_radial_mask(d_pts[cur], d_off[cur], K, left, k_std, d_partial, d_stats, d_keep, d_cnt, st));
    LSLAM_HIP_TRY(launch_compact(d_pts[cur], d_off[cur], 0, d_keep, d_cnt, K, d_off[cur ^ 1], d_pts[cur ^ 1], st));
    cur ^= 1;
    if (out_stats) LSLAM_HIP_TRY(hipMemcpyAsync(out_stats, d_stats, sizeof(double) * 3, hipMemcpyDeviceToHost, st));
  }
  LSLAM_HIP_TRY(hipMemcpyAsync(out_off, d_off[cur], b_off, hipMemcpyDeviceToHost, st));
  LSLAM_HIP_TRY(hipStreamSynchronize(st));
  if (need_host_xy && out_off[K] > 0)
    LSLAM_HIP_TRY(hipMemcpy(out_xy, d_pts[cur], sizeof(double) * 2 * (size_t)out_off[K], hipMemcpyDeviceToHost));
  if (d_final) *d_final = d_pts[cur];
  return LSLAM_OK;
}

int lslam_lidar_preprocess(uint32_t stages, const double* ranges, int32_t K, int32_t B, const double* cos_a,
                           const double* sin_a, double off_x, double off_y, double range_min, double range_max,
                           const double* in_xy, const int64_t* in_off, double eps, int32_t min_samples, double k_std,
                           double* out_xy, int64_t* out_off, double* out_stats, int device) {
  clear_error();
  if (!(stages & (LSLAM_PRE_POLAR | LSLAM_PRE_DBSCAN | LSLAM_PRE_RADIAL))) return fail(LSLAM_ERR_BAD_ARG, "no stage selected");
  return preprocess_core(stages, ranges, K, B, cos_a, sin_a, off_x, off_y, range_min, range_max, in_xy, in_off, eps,
                         min_samples, k_std, out_xy, out_off, out_stats, device, true, nullptr, nullptr);
}

// ---------------------------------------------------------------------------------------------------
extern "C++" {
namespace {
struct ScanSet {
  int device = 0;
  int64_t K = 0, total = 0, max_n = 0;
  double* d_xy64 = nullptr;
  float* d_xy32 = nullptr;
  int64_t* d_off = nullptr;
  std::vector<int64_t> h_off;
};
struct Ogm {
  OgmGrid g{};
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;                  // uploads of a long trajectory, one chunk ahead of the kernels
  hipEvent_t landed = nullptr;
  double* d_scan = nullptr; int64_t scan_cap = 0;     // points
  int4* d_rays = nullptr; int64_t rays_cap = 0;       // one record per ray of the current call
  int4* d_batch = nullptr;                            // one box per 64 rays of a scan (rays_cap / 64 + desc_cap of them)
  int planes_cap = 0;                                 // count planes allocated so far (grows to kOgmPlanes on the first long call)
  int* d_boxes = nullptr; int64_t boxes_cap = 0;      // one touched box per scan of the current call
  uint8_t* d_bin = nullptr;
  float* d_values = nullptr;                          // read-out staging: f32(base + steps·step)
  OgmScan* d_desc = nullptr; int64_t desc_cap = 0;    // one descriptor per scan of the current call
  std::vector<OgmScan> h_desc;
  std::vector<int> h_boxes;                           // host copy of the touched boxes: sizes the launches
};
int ogm_scan_buffer(Ogm* o, int64_t points) {
  if (points <= o->scan_cap) return LSLAM_OK;
  if (o->d_scan) (void)hipFree(o->d_scan);
  o->d_scan = nullptr; o->scan_cap = 0;
  const int64_t want = points + points / 4 + 1024;
  LSLAM_HIP_TRY(hipMalloc(&o->d_scan, sizeof(double) * 2 * (size_t)want));
  o->scan_cap = want;
  return LSLAM_OK;
}
}  // namespace
}  // extern "C++"

int lslam_scans_destroy(void* scans) {
  if (!scans) return LSLAM_OK;
  ScanSet* z = static_cast<ScanSet*>(scans);
  (void)hipSetDevice(z->device);
  if (z->d_xy64) (void)hipFree(z->d_xy64);
  if (z->d_xy32) (void)hipFree(z->d_xy32);
  if (z->d_off) (void)hipFree(z->d_off);
  delete z;
  return LSLAM_OK;
}

int lslam_scans_create(uint32_t stages, const double* ranges, int32_t K, int32_t B, const double* cos_a,
                       const double* sin_a, double off_x, double off_y, double range_min, double range_max,
                       const double* in_xy, const int64_t* in_off, double eps, int32_t min_samples, double k_std,
                       double* out_stats, int device, void** out_handle) {
  clear_error();
  if (!out_handle) return fail(LSLAM_ERR_BAD_ARG, "null handle pointer");
  *out_handle = nullptr;
  if (K < 0) return fail(LSLAM_ERR_BAD_ARG, "negative scan count");
  ScanSet* z = new (std::nothrow) ScanSet();
  if (!z) return fail(LSLAM_ERR_BAD_ARG, "out of host memory");
  z->device = device;
  z->K = K;
  z->h_off.assign((size_t)K + 1, 0);
  const double* d_final = nullptr;
  hipStream_t st = nullptr;
  int rc = preprocess_core(stages, ranges, K, B, cos_a, sin_a, off_x, off_y, range_min, range_max, in_xy, in_off, eps,
                           min_samples, k_std, nullptr, z->h_off.data(), out_stats, device, false, &d_final, &st);
  if (rc != LSLAM_OK) { delete z; return rc; }
  z->total = z->h_off[(size_t)K];
  for (int64_t k = 0; k < K; ++k) z->max_n = std::max(z->max_n, z->h_off[(size_t)k + 1] - z->h_off[(size_t)k]);
  hipError_t e = hipSuccess;
  const size_t pts = (size_t)(z->total > 0 ? z->total : 1);
  if ((e = hipMalloc(&z->d_xy64, sizeof(double) * 2 * pts)) != hipSuccess ||
      (e = hipMalloc(&z->d_xy32, sizeof(float) * 2 * pts)) != hipSuccess ||
      (e = hipMalloc(&z->d_off, sizeof(int64_t) * ((size_t)K + 1))) != hipSuccess) {
    lslam_scans_destroy(z);
    return fail(LSLAM_ERR_HIP, "scan set allocation failed: %s", hipGetErrorString(e));
  }
  if (K > 0) {
    if (z->total > 0) {
      e = hipMemcpyAsync(z->d_xy64, d_final, sizeof(double) * 2 * (size_t)z->total, hipMemcpyDeviceToDevice, st);
      if (e == hipSuccess) e = launch_f64_to_f32(z->d_xy64, z->d_xy32, 2 * z->total, st);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(z->d_off, z->h_off.data(), sizeof(int64_t) * ((size_t)K + 1), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { lslam_scans_destroy(z); return fail(LSLAM_ERR_HIP, "scan set upload failed: %s", hipGetErrorString(e)); }
  }
  *out_handle = z;
  return LSLAM_OK;
}

int lslam_scans_info(void* scans, int64_t* K, int64_t* total_points, int64_t* max_points) {
  clear_error();
  if (!scans) return fail(LSLAM_ERR_BAD_ARG, "null handle");
  ScanSet* z = static_cast<ScanSet*>(scans);
  if (K) *K = z->K;
  if (total_points) *total_points = z->total;
  if (max_points) *max_points = z->max_n;
  return LSLAM_OK;
}

int lslam_scans_download(void* scans, double* out_xy, int64_t* out_off) {
  clear_error();
  if (!scans || !out_off) return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  ScanSet* z = static_cast<ScanSet*>(scans);
  LSLAM_HIP_TRY(hipSetDevice(z->device));
  memcpy(out_off, z->h_off.data(), sizeof(int64_t) * ((size_t)z->K + 1));
  if (out_xy && z->total > 0) LSLAM_HIP_TRY(hipMemcpy(out_xy, z->d_xy64, sizeof(double) * 2 * (size_t)z->total, hipMemcpyDeviceToHost));
  return LSLAM_OK;
}

int lslam_scans_match_sequence(void* scans, const double* init_T, double epsilon, double stopping_thresh,
                               int32_t max_iters, uint32_t flags, double* out_T, double* out_err, int32_t* out_iters) {
  clear_error();
  if (!scans) return fail(LSLAM_ERR_BAD_ARG, "null handle");
  ScanSet* z = static_cast<ScanSet*>(scans);
  const int64_t S = z->K - 1;
  if (S <= 0) return LSLAM_OK;
  if (S > 0x7fffffff) return fail(LSLAM_ERR_BAD_ARG, "too many scans");
  if (!init_T || !out_T || !out_err || !out_iters) return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  for (int64_t s = 0; s < S; ++s) {
    if (z->h_off[(size_t)s + 1] == z->h_off[(size_t)s]) return fail(LSLAM_ERR_EMPTY_TARGET, "problem %lld has an empty target cloud", (long long)s);
    const double* T = init_T + 16 * (size_t)s;
    if (T[8] != 0.0 || T[9] != 0.0 || T[11] != 0.0) return fail(LSLAM_ERR_BAD_ARG, "problem %lld: init_T leaves the z = 0 plane", (long long)s);
  }
  int rc;
  if ((rc = use_device(z->device)) != LSLAM_OK) return rc;
  const size_t b_T = sizeof(double) * 16 * (size_t)S;
  Arena& ar = thread_arena();
  if ((rc = ar.reserve(z->device, 2 * Arena::align(b_T) + Arena::align(sizeof(double) * S) + Arena::align(sizeof(int32_t) * S))) != LSLAM_OK) return rc;
  double* d_iT = static_cast<double*>(ar.carve(b_T));
  double* d_oT = static_cast<double*>(ar.carve(b_T));
  double* d_err = static_cast<double*>(ar.carve(sizeof(double) * S));
  int32_t* d_it = static_cast<int32_t*>(ar.carve(sizeof(int32_t) * S));
  hipStream_t st;
  if ((rc = thread_stream(&st)) != LSLAM_OK) return rc;
  LSLAM_HIP_TRY(hipMemcpyAsync(d_iT, init_T, b_T, hipMemcpyHostToDevice, st));
  // source = scan s+1, target = scan s (localization.py:120-121): the same buffer with shifted offsets
  // both copies of the points are resident: float64-exact mode
  rc = lslam_icp_batch_dev64(z->d_xy32, z->d_xy64, z->d_off + 1, z->d_xy32, z->d_xy64, z->d_off, 2, d_iT, (int32_t)S, z->max_n, z->max_n,
                             epsilon, stopping_thresh, max_iters, flags, d_oT, d_err, d_it, nullptr, st);
  if (rc != LSLAM_OK) return rc;
  LSLAM_HIP_TRY(hipMemcpyAsync(out_T, d_oT, b_T, hipMemcpyDeviceToHost, st));
  LSLAM_HIP_TRY(hipMemcpyAsync(out_err, d_err, sizeof(double) * S, hipMemcpyDeviceToHost, st));
  LSLAM_HIP_TRY(hipMemcpyAsync(out_iters, d_it, sizeof(int32_t) * S, hipMemcpyDeviceToHost, st));
  LSLAM_HIP_TRY(hipStreamSynchronize(st));
  return LSLAM_OK;
}

int lslam_ogm_create(double res, double world_min_x, double world_min_y, int32_t width, int32_t height,
                     double logodds_step, double logodds_clip, int device, void** out_handle) {
  clear_error();
  if (!out_handle) return fail(LSLAM_ERR_BAD_ARG, "null handle pointer");
  *out_handle = nullptr;
  if (!(res > 0.0) || width <= 0 || height <= 0) return fail(LSLAM_ERR_BAD_ARG, "need res > 0 and positive grid dimensions");
  int rc;
  if ((rc = use_device(device)) != LSLAM_OK) return rc;
  Ogm* o = new (std::nothrow) Ogm();
  if (!o) return fail(LSLAM_ERR_BAD_ARG, "out of host memory");
  o->device = device;
  o->g.width = width; o->g.height = height; o->g.res = res; o->g.min_x = world_min_x; o->g.min_y = world_min_y;
  o->g.step = logodds_step; o->g.clip = logodds_clip;
  const size_t cells = (size_t)width * height;
  hipError_t e;
  if ((e = hipStreamCreateWithFlags(&o->stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipMalloc(&o->g.logodds, sizeof(float) * cells)) != hipSuccess ||
      (e = hipMalloc(&o->g.steps, sizeof(int32_t) * cells)) != hipSuccess ||
      (e = hipMalloc(&o->d_values, sizeof(float) * cells)) != hipSuccess ||
      (e = hipMalloc(&o->d_bin, cells)) != hipSuccess ||
      (e = launch_ogm_init(o->g, o->stream)) != hipSuccess ||
      (e = hipStreamSynchronize(o->stream)) != hipSuccess) {
    lslam_ogm_destroy(o);
    return fail(LSLAM_ERR_HIP, "occupancy grid allocation failed: %s", hipGetErrorString(e));
  }
  *out_handle = o;
  return LSLAM_OK;
}

int lslam_ogm_destroy(void* handle) {
  if (!handle) return LSLAM_OK;
  Ogm* o = static_cast<Ogm*>(handle);
  (void)hipSetDevice(o->device);
  if (o->stream) (void)hipStreamSynchronize(o->stream);
  if (o->copy_stream) { (void)hipStreamSynchronize(o->copy_stream); (void)hipStreamDestroy(o->copy_stream); }
  if (o->landed) (void)hipEventDestroy(o->landed);
  if (o->g.logodds) (void)hipFree(o->g.logodds);
  if (o->g.steps) (void)hipFree(o->g.steps);
  if (o->d_values) (void)hipFree(o->d_values);
  if (o->g.counts) (void)hipFree(o->g.counts);
  if (o->d_rays) (void)hipFree(o->d_rays);
  if (o->d_batch) (void)hipFree(o->d_batch);
  if (o->d_boxes) (void)hipFree(o->d_boxes);
  if (o->d_scan) (void)hipFree(o->d_scan);
  if (o->d_bin) (void)hipFree(o->d_bin);
  if (o->d_desc) (void)hipFree(o->d_desc);
  if (o->stream) (void)hipStreamDestroy(o->stream);
  delete o;
  return LSLAM_OK;
}

int lslam_ogm_set_params(void* handle, double logodds_step, double logodds_clip) {
  clear_error();
  if (!handle) return fail(LSLAM_ERR_BAD_ARG, "null handle");
  Ogm* o = static_cast<Ogm*>(handle);
  if (!(logodds_step > 0.0) || !(logodds_clip > 0.0)) return fail(LSLAM_ERR_BAD_ARG, "log-odds step and clip must be positive");
  if (logodds_step != o->g.step) {
    // cells hold (base, n) with value base + n·step: settle them at the old step size before it changes
    LSLAM_HIP_TRY(hipSetDevice(o->device));
    LSLAM_HIP_TRY(launch_ogm_values(o->g, o->g.logodds, o->stream));
    LSLAM_HIP_TRY(hipStreamSynchronize(o->stream));
  }
  o->g.step = logodds_step;
  o->g.clip = logodds_clip;
  return LSLAM_OK;
}

static int ogm_planes(Ogm* o, int planes) {
  if (planes <= o->planes_cap) return LSLAM_OK;
  LSLAM_HIP_TRY(hipStreamSynchronize(o->stream));
  if (o->g.counts) (void)hipFree(o->g.counts);
  o->g.counts = nullptr; o->planes_cap = 0;
  LSLAM_HIP_TRY(hipMalloc(&o->g.counts, sizeof(int32_t) * (size_t)o->g.width * o->g.height * (size_t)planes));
  o->planes_cap = planes;
  return LSLAM_OK;
}

// Upload one descriptor per scan; one launch turns every ray into a record and every scan into a touched
// box; then per group of up to kOgmPlanes scans: tile launch → apply (or nothing) launch.  No host
// synchronisation inside.
// descriptors / boxes for K scans, ray records (and 64-ray batch boxes) for `total` rays
static int ogm_reserve(Ogm* o, int64_t K, int64_t total) {
  if (K > o->desc_cap || total > o->rays_cap) {
    LSLAM_HIP_TRY(hipStreamSynchronize(o->stream));          // earlier work may still use the old buffers
    if (o->d_desc) (void)hipFree(o->d_desc);
    if (o->d_boxes) (void)hipFree(o->d_boxes);
    if (o->d_rays) (void)hipFree(o->d_rays);
    if (o->d_batch) (void)hipFree(o->d_batch);
    o->d_desc = nullptr; o->d_boxes = nullptr; o->d_rays = nullptr; o->d_batch = nullptr;
    const int64_t k_now = K > o->desc_cap ? K : o->desc_cap, r_now = total > o->rays_cap ? total : o->rays_cap;
    o->desc_cap = 0; o->rays_cap = 0;
    const int64_t want_k = k_now + k_now / 4 + 64, want_r = r_now + r_now / 4 + 1024;
    LSLAM_HIP_TRY(hipMalloc(&o->d_desc, sizeof(OgmScan) * (size_t)want_k));
    LSLAM_HIP_TRY(hipMalloc(&o->d_boxes, sizeof(int) * 4 * (size_t)want_k));
    LSLAM_HIP_TRY(hipMalloc(&o->d_rays, sizeof(int4) * (size_t)want_r));
    LSLAM_HIP_TRY(hipMalloc(&o->d_batch, sizeof(int4) * (size_t)(want_r / 64 + want_k + 1)));
    o->desc_cap = want_k; o->rays_cap = want_r;
  }
  return LSLAM_OK;
}

static int ogm_enqueue(Ogm* o, const double* poses, const double* d_points, const int64_t* scan_off,
                       int64_t first_point, int32_t K, bool apply) {
  if (K <= 0) return LSLAM_OK;
  const int64_t total = scan_off[K] - first_point;
  int rc_planes = ogm_planes(o, K < kOgmPlanes ? K : kOgmPlanes);     // update_map needs one plane, build_map a full group
  if (rc_planes != LSLAM_OK) return rc_planes;
  { const int rc_res = ogm_reserve(o, K, total); if (rc_res != LSLAM_OK) return rc_res; }
  o->h_desc.resize((size_t)K);
  int max_rays = 0;
  for (int32_t k = 0; k < K; ++k) {
    const double* pose = poses + 3 * (size_t)k;
    OgmScan& h = o->h_desc[(size_t)k];
    h.ray_base = scan_off[k] - first_point;
    h.batch_base = h.ray_base / 64 + k;                      // distinct slots: scan k+1 starts past the last batch of scan k
    h.scan_xy = d_points + 2 * h.ray_base;
    h.n = (int32_t)(scan_off[k + 1] - scan_off[k]);
    h.x = pose[0]; h.y = pose[1]; h.c = cos(pose[2]); h.s = sin(pose[2]);
    h.pad_ = 0;
    if (h.n > max_rays) max_rays = h.n;
  }
  LSLAM_HIP_TRY(hipMemcpyAsync(o->d_desc, o->h_desc.data(), sizeof(OgmScan) * (size_t)K, hipMemcpyHostToDevice, o->stream));
  const bool wide = max_rays > 32767;
  LSLAM_HIP_TRY(launch_ogm_rays(o->g, o->d_desc, K, max_rays, o->d_rays, o->d_boxes, o->d_batch, o->stream));
  // the boxes come back (16 bytes per scan) so that the launches below cover exactly the touched tiles and cells
  o->h_boxes.resize(4 * (size_t)K);
  LSLAM_HIP_TRY(hipMemcpyAsync(o->h_boxes.data(), o->d_boxes, sizeof(int) * 4 * (size_t)K, hipMemcpyDeviceToHost, o->stream));
  LSLAM_HIP_TRY(hipStreamSynchronize(o->stream));
  for (int32_t k0 = 0; k0 < K; k0 += kOgmPlanes) {
    const int32_t cnt = (K - k0 < kOgmPlanes) ? (K - k0) : kOgmPlanes;
    int tiles_x = 0, tiles_y = 0, x0 = 0x7fffffff, y0 = 0x7fffffff, x1 = -1, y1 = -1;
    for (int32_t b = 0; b < cnt; ++b) {
      const int* box = o->h_boxes.data() + 4 * (size_t)(k0 + b);
      if (box[0] == kOgmBoxEmpty) continue;
      int tx, ty;
      ogm_tile_span(box, &tx, &ty);
      tiles_x = tx > tiles_x ? tx : tiles_x; tiles_y = ty > tiles_y ? ty : tiles_y;
      x0 = box[0] < x0 ? box[0] : x0; y0 = box[1] < y0 ? box[1] : y0;
      x1 = -box[2] > x1 ? -box[2] : x1; y1 = -box[3] > y1 ? -box[3] : y1;
    }
    LSLAM_HIP_TRY(launch_ogm_tiles(o->g, o->d_desc + k0, cnt, o->d_rays, o->d_boxes + 4 * (size_t)k0, o->d_batch, wide, tiles_x, tiles_y, o->stream));
    if (apply) LSLAM_HIP_TRY(launch_ogm_apply(o->g, o->d_boxes + 4 * (size_t)k0, cnt, wide, x0, y0, x1, y1, o->stream));
  }
  return LSLAM_OK;
}

int lslam_ogm_update(void* handle, const double* poses, const double* scans_xy, const int64_t* scan_off, int32_t K) {
  clear_error();
  if (!handle || !poses || !scan_off) return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  if (K < 0) return fail(LSLAM_ERR_BAD_ARG, "negative scan count");
  Ogm* o = static_cast<Ogm*>(handle);
  LSLAM_HIP_TRY(hipSetDevice(o->device));
  for (int32_t k = 0; k < K; ++k)
    if (scan_off[k + 1] < scan_off[k] || scan_off[k + 1] - scan_off[k] > 0x7fffffff)
      return fail(LSLAM_ERR_BAD_ARG, "scan_off is not monotone at %d", k);
  const int64_t first = scan_off[0], total = scan_off[K] - first;
  int rc;
  if ((rc = ogm_scan_buffer(o, total)) != LSLAM_OK) return rc;
  if (total > 0 && !scans_xy) return fail(LSLAM_ERR_BAD_ARG, "null scan pointer");
  // A long trajectory goes in chunks of scans: while one chunk is traced and applied (scan order is kept: chunks and
  // the groups inside them run in order on the grid's stream), the points of the next are on their way over PCIe on a
  // second stream.  Same cells, same order, same result as one piece.
  constexpr int32_t kChunk = 8 * kOgmPlanes;                 // 1024 scans
  if (K <= kChunk || total * 16 < (8 << 20) || getenv("LSLAM_NO_UPLOAD_PIPELINE")) {
    if (total > 0) LSLAM_HIP_TRY(hipMemcpyAsync(o->d_scan, scans_xy + 2 * first, sizeof(double) * 2 * (size_t)total, hipMemcpyHostToDevice, o->stream));
    if ((rc = ogm_enqueue(o, poses, o->d_scan, scan_off, first, K, true)) != LSLAM_OK) return rc;
    LSLAM_HIP_TRY(hipStreamSynchronize(o->stream));
    return LSLAM_OK;
  }
  if (!o->copy_stream) {
    LSLAM_HIP_TRY(hipStreamCreateWithFlags(&o->copy_stream, hipStreamNonBlocking));
    LSLAM_HIP_TRY(hipEventCreateWithFlags(&o->landed, hipEventDisableTiming));
  }
  if ((rc = ogm_reserve(o, K, total)) != LSLAM_OK) return rc;           // once, for the whole trajectory (chunks index into it)
  LSLAM_HIP_TRY(hipEventRecord(o->landed, o->stream));       // the copies start after whatever still reads d_scan
  LSLAM_HIP_TRY(hipStreamWaitEvent(o->copy_stream, o->landed, 0));
  // every chunk's copy is queued up front (they run back to back on the copy stream), each followed by its own event
  const int32_t nchunks = (K + kChunk - 1) / kChunk;
  std::vector<hipEvent_t> ev((size_t)nchunks);
  for (int32_t c = 0; c < nchunks; ++c) {
    const int32_t k0 = c * kChunk, k1 = std::min(K, k0 + kChunk);
    const int64_t p0 = scan_off[k0] - first, p1 = scan_off[k1] - first;
    if (p1 > p0) LSLAM_HIP_TRY(hipMemcpyAsync(o->d_scan + 2 * p0, scans_xy + 2 * (first + p0), sizeof(double) * 2 * (size_t)(p1 - p0), hipMemcpyHostToDevice, o->copy_stream));
    LSLAM_HIP_TRY(hipEventCreateWithFlags(&ev[(size_t)c], hipEventDisableTiming));
    LSLAM_HIP_TRY(hipEventRecord(ev[(size_t)c], o->copy_stream));
  }
  rc = LSLAM_OK;
  for (int32_t c = 0; c < nchunks && rc == LSLAM_OK; ++c) {
    const int32_t k0 = c * kChunk, k1 = std::min(K, k0 + kChunk);
    if (hipStreamWaitEvent(o->stream, ev[(size_t)c], 0) != hipSuccess) { rc = fail(LSLAM_ERR_HIP, "hipStreamWaitEvent failed"); break; }
    rc = ogm_enqueue(o, poses + 3 * (size_t)k0, o->d_scan, scan_off + k0, first, k1 - k0, true);
  }
  const hipError_t e = hipStreamSynchronize(o->stream);
  (void)hipStreamSynchronize(o->copy_stream);
  for (auto& x : ev) (void)hipEventDestroy(x);
  if (rc != LSLAM_OK) return rc;
  LSLAM_HIP_TRY(e);
  return LSLAM_OK;
}

int lslam_ogm_update_scans(void* handle, void* scans, const double* poses) {
  clear_error();
  if (!handle || !scans || !poses) return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  Ogm* o = static_cast<Ogm*>(handle);
  ScanSet* z = static_cast<ScanSet*>(scans);
  if (o->device != z->device) return fail(LSLAM_ERR_BAD_ARG, "grid and scan set live on different devices");
  if (z->K > 0x7fffffff) return fail(LSLAM_ERR_BAD_ARG, "too many scans");
  LSLAM_HIP_TRY(hipSetDevice(o->device));
  int rc;
  if ((rc = ogm_enqueue(o, poses, z->d_xy64, z->h_off.data(), 0, (int32_t)z->K, true)) != LSLAM_OK) return rc;
  LSLAM_HIP_TRY(hipStreamSynchronize(o->stream));
  return LSLAM_OK;
}

int lslam_ogm_scan_counts(void* handle, const double* pose, const double* scan_xy, int32_t n, int32_t* out_counts) {
  clear_error();
  if (!handle || !pose || !out_counts || (n > 0 && !scan_xy)) return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  if (n < 0) return fail(LSLAM_ERR_BAD_ARG, "negative ray count");
  Ogm* o = static_cast<Ogm*>(handle);
  LSLAM_HIP_TRY(hipSetDevice(o->device));
  int rc;
  if ((rc = ogm_scan_buffer(o, n)) != LSLAM_OK) return rc;
  if (n > 0) LSLAM_HIP_TRY(hipMemcpyAsync(o->d_scan, scan_xy, sizeof(double) * 2 * (size_t)n, hipMemcpyHostToDevice, o->stream));
  const int64_t off[2] = {0, n};
  const size_t cells = (size_t)o->g.width * o->g.height;
  const bool wide = n > 32767;                              // as ogm_enqueue decides
  // plane 0 is written inside the scan's box only; the caller gets the whole grid
  if ((rc = ogm_planes(o, 1)) != LSLAM_OK) return rc;
  LSLAM_HIP_TRY(hipMemsetAsync(o->g.counts, 0, sizeof(int32_t) * cells, o->stream));
  if ((rc = ogm_enqueue(o, pose, o->d_scan, off, 0, 1, false)) != LSLAM_OK) return rc;
  if (wide) {
    LSLAM_HIP_TRY(hipMemcpyAsync(out_counts, o->g.counts, sizeof(int32_t) * cells, hipMemcpyDeviceToHost, o->stream));
    LSLAM_HIP_TRY(hipStreamSynchronize(o->stream));
  } else {
    std::vector<int16_t> narrow(cells);
    LSLAM_HIP_TRY(hipMemcpyAsync(narrow.data(), o->g.counts, sizeof(int16_t) * cells, hipMemcpyDeviceToHost, o->stream));
    LSLAM_HIP_TRY(hipStreamSynchronize(o->stream));
    for (size_t i = 0; i < cells; ++i) out_counts[i] = narrow[i];
  }
  return LSLAM_OK;
}

int lslam_ogm_read_logodds(void* handle, float* out) {
  clear_error();
  if (!handle || !out) return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  Ogm* o = static_cast<Ogm*>(handle);
  LSLAM_HIP_TRY(hipSetDevice(o->device));
  LSLAM_HIP_TRY(launch_ogm_values(o->g, o->d_values, o->stream));
  LSLAM_HIP_TRY(hipMemcpyAsync(out, o->d_values, sizeof(float) * (size_t)o->g.width * o->g.height, hipMemcpyDeviceToHost, o->stream));
  LSLAM_HIP_TRY(hipStreamSynchronize(o->stream));
  return LSLAM_OK;
}

int lslam_ogm_write_logodds(void* handle, const float* in) {
  clear_error();
  if (!handle || !in) return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  Ogm* o = static_cast<Ogm*>(handle);
  LSLAM_HIP_TRY(hipSetDevice(o->device));
  LSLAM_HIP_TRY(hipMemcpyAsync(o->g.logodds, in, sizeof(float) * (size_t)o->g.width * o->g.height, hipMemcpyHostToDevice, o->stream));
  LSLAM_HIP_TRY(hipMemsetAsync(o->g.steps, 0, sizeof(int32_t) * (size_t)o->g.width * o->g.height, o->stream));
  LSLAM_HIP_TRY(hipStreamSynchronize(o->stream));
  return LSLAM_OK;
}

int lslam_ogm_binarize(void* handle, uint8_t* out) {
  clear_error();
  if (!handle || !out) return fail(LSLAM_ERR_BAD_ARG, "null pointer argument");
  Ogm* o = static_cast<Ogm*>(handle);
  LSLAM_HIP_TRY(hipSetDevice(o->device));
  LSLAM_HIP_TRY(launch_ogm_binarize(o->g, o->d_bin, o->stream));
  LSLAM_HIP_TRY(hipMemcpyAsync(out, o->d_bin, (size_t)o->g.width * o->g.height, hipMemcpyDeviceToHost, o->stream));
  LSLAM_HIP_TRY(hipStreamSynchronize(o->stream));
  return LSLAM_OK;
}


/* how many calls of this thread had their co-resident one-launch ICP (LiDAR-sized single pairs) time out waiting for its
 * workgroups and were re-run on the launch-per-iteration path — 0 on an otherwise idle device */
int lslam_icp_coop_fallbacks(void) { return g_coop_fallbacks; }

/* Returns what the calling thread's earlier calls left allocated on the current device side: the scratch arena, the pinned
 * staging buffers, the spatially ordered copies of the pruned batch kernels, the workspaces of the single-pair paths and the
 * thread's stream.  Waits for the device first; the next call simply allocates again. */
int lslam_release_thread_resources(void) {
  clear_error();
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return LSLAM_OK;      // nothing was ever allocated
  LSLAM_HIP_TRY(hipDeviceSynchronize());
  thread_arena().release();
  staging_holder().release();
  prune_holder().release();
  prune3_holder().release();
  large_ws().release();
  large_ws().device = -1;
  lp_release();
  if (g_thread_stream) { (void)hipStreamDestroy(g_thread_stream); g_thread_stream = nullptr; g_thread_stream_dev = -1; }
  WsFence& f = ws_fence();
  if (f.ev) { (void)hipEventDestroy(f.ev); f.ev = nullptr; }
  f.armed = false; f.device = -1; f.last = nullptr;
  return LSLAM_OK;
}

}  // extern "C"
