// lslam_icp_queue.hip — the batched planar ICP with exact chunk pruning as a TASK QUEUE (the default from 256 points per cloud).
//
// Same operator, same search, same results as icp_pruned_kernel (lslam_icp_pruned.hip; modules/icp.py:123-189, the per-pair
// loop of modules/localization.py:115-124) — what changes is who waits for whom.  There, a problem is a workgroup: every
// iteration its waves meet at two barriers around a one-wave solve.  Here a workgroup is a pool of waves that stays on its CU
// for the whole launch and keeps Q problems ("slots") in LDS.  The unit of work is a TASK = G consecutive slabs of 64 sorted
// sources of one problem in its current iteration:
//   * a wave takes the next task of any slot that has one (an LDS counter per slot), searches its sources, accumulates their
//     float64 moments and leaves one row of partial sums per task;
//   * the wave that completes a slot's LAST task of the iteration sums the rows in task order (bitwise reproducible: which
//     wave ran which task does not matter), solves, applies the stop rules and re-opens the slot's counter — or, when the
//     problem has converged, writes its outputs, draws the next problem from a global counter and stages it into the slot;
//   * meanwhile every other wave works on the other slots.  There is no __syncthreads() after the first instruction and no
//     wave ever waits while any resident problem has an open task.  G is chosen per problem and iteration from the number of
//     problems the workgroup still holds: few, large tasks while every slot is busy (one reduction per task), single slabs when
//     the queue runs dry, so that the last problems of a launch are iterated by all waves.
// The hand-off is LDS only: a slot's ticket word (tasks of the round | tasks handed out) is parked at kQClosed while the slot is
// solved or re-staged (a late atomicAdd then returns an invalid ticket whatever else changes), everything a task reads is written
// before the ticket is re-opened (workgroup-scope release / acquire), and a wave reads n, m, T only after it has drawn its task.
// A wave looks at the tickets of all slots with one load (lane q: slot q) and at a slot's control block with another.
// Every wave leaves when all slots are empty: a slot empties when its problem has converged and the global counter is
// exhausted, which every problem reaches through max_iters at the latest; idle waves poll with s_sleep and give up after
// kQTimeoutTicks of wall clock without any task (a can't-happen guard: the grid must always drain).
//
// Iterations after the first do not search every source again.  A search leaves, per source, a MARGIN: half the gap between
// its nearest target and a proven lower bound on the distance to every other target (the second smallest distance evaluated, and
// the reach beyond which chunks were skipped — extended by E for that purpose), minus the rounding of the positions.  Every later
// iteration subtracts a bound on how far the new transform moved the source (‖ΔR‖₂·|p|max + |Δt|, |p|max over the cloud); while the margin
// stays positive the nearest target cannot have changed — by the triangle inequality, with the float32 recipe's rounding and in
// float64-exact mode the near-tie window inside the margin — and the source keeps its match.  The sources of a task whose margin
// ran out are gathered, 64 at a time, and searched as before.  Exactness is untouched: a source is skipped only on a proof.
//
// RUNNING SUMS (template parameter INCR: wherever certificates are on and the normalised error's box is not asked for).  With most matches
// kept, what an iteration still did for every pair was the moment pass — but the transform enters Σp, Σpqᵀ and Σ‖p − q‖² only through products
// with sums over the ORIGINAL sources and the matched targets, and those change only where a match changes (formulas at `finish`).  The first
// iteration takes them over all pairs (base_task), later searches add new − old for the lanes whose match changed, the finisher evaluates the
// formulas.  Σ‖p − q‖² cancels five to eight digits, so the sums are double-doubles (DD) and its 25 products are formed exactly.  Certificates
// become EXPIRIES (a float per source in a scratch buffer: the value of the slot's accumulated step bound at which the margin runs out), the
// matches live beside them, per block of Gb slabs the earliest expiry is kept in LDS — a block none of whose certificates runs out is not
// visited, and an iteration in which none runs out anywhere is taken by the finishing wave alone (new T from the same sums, error, stop
// rules).  A slot is then the chunk records + 2 KB: four 4096-point problems per CU, in float64-exact mode as well (the float64 points are
// read from the sorted copies in memory where a match changes).  Which build iterates a problem depends on the problem's own size
// (launch_icp_pruned files the problems of a mixed batch by class), so every bit of its result is independent of the batch.
//
// LDS record of a chunk: its 16 targets (128 B) followed by its box (16 B) — 144 B.  A lane that compares its point with a
// chunk reads eight 16-byte pairs at IMMEDIATE offsets from one address register (no per-target address arithmetic); lanes on
// different chunks fall on the same banks only when their chunk ids agree modulo 16, and the 64 boxes of a ballot group are a
// conflict-free stride.  Tail slots of the last chunk hold far, pairwise distinct points (finite distances ≈ 10³⁶: a real target
// always wins; sources farther than 10¹⁵ from the origin take the exhaustive pass).
//
// float64-exact mode (args.src64 / tgt64 set): a slot then holds the float64 SOURCES themselves (the float32
// point the search uses is their rounding, as in every other kernel) and, beside the float32 targets, the RESIDUAL of every
// target coordinate t64 − f64(t32) as an int32 count of 2^(e − 54), e = frexp exponent of t32: the residual is a multiple of
// that unit below 2²⁹ of them, so f64(t32) + ldexp(count, e − 54) IS the float64 target, bit for bit (8 bytes per target
// instead of 16).  The moments of every iteration come from LDS; round 3's kernels re-read the float64 clouds from HBM in every
// iteration (33 GB per C4 launch against 0.8 GB compulsory).  Near ties are rare and decided on the float64 targets read from the
// sorted copy in memory.  (That is the build WITHOUT running sums — LiDAR-sized problems; with them no float64 point is kept on chip.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stddef.h>
#include <stdlib.h>

#include "lslam_device.h"
#include "lslam_kernels.h"
#include "lslam_prune_common.h"

namespace lslam {

constexpr int kQCtlBytes = 256;
constexpr long long kQTimeoutTicks = 400000000ll;      // 4 s of the 100 MHz wall clock
constexpr int kQRec = kPruneChunk * 8 + 16;            // bytes per chunk record: 16 targets + the chunk's box
#ifndef QEXP_STAGE
#define QEXP_STAGE 8
#endif
#ifndef QEXP_F64ROWS
#define QEXP_F64ROWS 8
#endif
// rows of float64 partial sums per slot and slabs per row above 20 slabs (moment_slabs): float32 mode 4 rows of 16 (32 above 64 slabs) — a
// 4096-point problem then leaves room for a third slot per CU —, float64-exact mode 8 rows of 8 (16): its one 4096-point problem per CU needs the finer tasks
constexpr int queue_rows(bool f64) { return f64 ? QEXP_F64ROWS : 4; }
constexpr int queue_row_slabs(bool f64) { return f64 ? 64 / QEXP_F64ROWS : 16; }
constexpr float kQExtend = 4.f;                        // E = kQExtend × the last step's largest displacement (capped): what a search pays for later certificates
constexpr float kQPad = 1.0e18f;                       // x of the padding targets (times 1 + slot/16), y = 0
constexpr float kQFarD2 = 1.0e35f;                     // a best distance beyond this is not trusted (padding, overflow): exhaustive pass
static_assert(kPruneChunk == 16, "the chunk record and the pair loop are written for 16 targets per chunk");

struct QueueCtl {                                      // one per slot, in LDS; a task reads the whole block with ONE 64-lane load (ctl_*)
  double T[16];                                        // dwords 0..31
  double err;
  StopState stop;
  int32_t problem;
  int32_t n, m, nchunks, nslabs, G, phase;             // phase 0: search tasks of G slabs, 1: moment tasks of moment_slabs() slabs
  uint32_t done_tasks;                                 // tasks completed in this round
  float Rt, Rs;                                        // largest |t|, |p| of the problem
  float stepR, stepT;                                  // ‖R_k − R_{k−1}‖_F and |t_k − t_{k−1}| of the transform the current iteration searches with (rounded up)
  float extend;                                        // E of the current iteration's searches
  float ecap;                                          // cap of E for this problem
  uint32_t dirty;                                      // running-sums build: bit t = task t of this round left a row of changes
  uint32_t rerun;                                      // … bit b = block b (Gb slabs) holds a source whose certificate may have run out: its task must look
  float moved;                                         // … Σ of the steps' displacement bounds so far (rounded up): a source searched at `moved` = a with margin g is certified while a + g > moved
  int32_t Gb;                                          // … slabs per block = per task after the first iteration
  uint32_t rounds;                                     // (LSLAM_DEBUG) rounds this problem has opened after the first
};
static_assert(sizeof(QueueCtl) <= kQCtlBytes, "control block");
#define QCTL_DWORD(field) (int)(offsetof(QueueCtl, field) / 4)
// Ticket word of a slot: tasks of the current round in the high half, tasks handed out in the low half; kQClosed while the slot
// is being solved or staged (a late atomicAdd then returns an invalid ticket whatever else changes).  State word: 0 = empty for
// good, 1 = holds a problem, 2 = being staged for the first time.  Both live in arrays of their own, one word per slot, so that a
// wave looks at all slots with one load.
constexpr unsigned kQClosed = 0x00008000u;

__device__ __forceinline__ unsigned lds_load_u32(const uint32_t* p) {
  return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ int lds_load_i32(const int32_t* p) {
  return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}

// a slot's control block through ONE load: lane k holds dword k, fields come out with v_readlane (wave-uniform, in scalar registers)
__device__ __forceinline__ int ctl_load(const QueueCtl* c, int lane) {
  return __hip_atomic_load(reinterpret_cast<const int*>(c) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#define CTL_INT(cw, field) __builtin_amdgcn_readlane(cw, QCTL_DWORD(field))
#define CTL_FLOAT(cw, field) __int_as_float(__builtin_amdgcn_readlane(cw, QCTL_DWORD(field)))
#define CTL_T(cw, k) __hiloint2double(__builtin_amdgcn_readlane(cw, 2 * (k) + 1), __builtin_amdgcn_readlane(cw, 2 * (k)))

// ---- double-double sums (the running-sums build, below): value = hi + lo, |lo| tiny; every product enters exactly (two_prod), every
// addition keeps its rounding error (two_sum) — the pair is good to ~1e-32 of the largest magnitude met, and is not kept normalised
struct DD { double hi, lo; };
__device__ __forceinline__ void two_sum(double a, double b, double& s, double& e) { s = a + b; const double bb = s - a; e = (a - (s - bb)) + (b - bb); }
__device__ __forceinline__ void two_prod(double a, double b, double& p, double& e) { p = a * b; e = __builtin_fma(a, b, -p); }
__device__ __forceinline__ void dd_acc(DD& a, double x) { double s, e; two_sum(a.hi, x, s, e); a.hi = s; a.lo += e; }
__device__ __forceinline__ void dd_acc_prod(DD& a, double x, double y) { double p, pe, s, e; two_prod(x, y, p, pe); two_sum(a.hi, p, s, e); a.hi = s; a.lo += e + pe; }
__device__ __forceinline__ void dd_add(DD& a, const DD& b) { double s, e; two_sum(a.hi, b.hi, s, e); a.hi = s; a.lo += e + b.lo; }
__device__ __forceinline__ DD dd_prod_diff(double a, double bn, double bo) {      // a·bn − a·bo
  double pn, en, po, eo; two_prod(a, bn, pn, en); two_prod(a, bo, po, eo);
  DD r; two_sum(pn, -po, r.hi, r.lo); r.lo += en - eo; return r;
}
__device__ __forceinline__ DD dd_sq_diff(double a, double b) {      // a² − b²
  double pn, en, po, eo; two_prod(a, a, pn, en); two_prod(b, b, po, eo);
  DD r; two_sum(pn, -po, r.hi, r.lo); r.lo += en - eo; return r;
}
template <int CTRL, int ROWS>
__device__ __forceinline__ void dd_dpp_step(DD& v) { const DD o{dpp_d<CTRL, ROWS>(v.hi), dpp_d<CTRL, ROWS>(v.lo)}; dd_add(v, o); }
__device__ __forceinline__ DD wave_sum_dd_last(DD v) {          // the sum over the wave, valid in lane 63 only (the order of wave_sum_dpp_last)
  dd_dpp_step<kDppXor1, 0xf>(v); dd_dpp_step<kDppXor2, 0xf>(v); dd_dpp_step<kDppHalfMirror, 0xf>(v);
  dd_dpp_step<kDppMirror, 0xf>(v); dd_dpp_step<kDppBcast15, 0xa>(v); dd_dpp_step<kDppBcast31, 0xc>(v);
  return v;
}
constexpr int kQStage = QEXP_STAGE;                     // loads a lane keeps in flight while a problem is staged
constexpr int kQBase = 12;                             // running sums (DD each): Σx Σy Σxx Σxy Σyy of the sources | Σqx Σqy | Σx·qx Σx·qy Σy·qx Σy·qy | Σ|q|² of the matches
constexpr int kQDelta = 7;                             // … of which the last seven depend on the matches
constexpr int kQDeltaRows = 16;                        // tasks of a round that may leave a row of changes
constexpr int kQBaseRows = 8;                          // tasks of the first iteration's round of sums
constexpr int kQIncrRows = 2 * kQBase + (2 * kQDelta * kQDeltaRows > 2 * kQBase * kQBaseRows ? 2 * kQDelta * kQDeltaRows : 2 * kQBase * kQBaseRows);      // doubles: the sums, then the round's rows
constexpr int kQContested = 16;                        // sources per problem that may be kept on "same float32 position" instead of a margin
constexpr int kQIncrDoubles = kQIncrRows + kQDeltaRows / 2 + (4 + 2 * kQContested + 8 * kQContested + 7) / 8;      // … then one float per block (the earliest expiry among its sources), then the contested list: count, indices, positions

// A float32 source from its LDS copy or, where a slot has no room for one, from the sorted copy in memory — as two explicit
// paths: written as one conditional expression the compiler selects between the two POINTERS and emits a flat_load (the slow
// path for an LDS address, and a wait on both memory counters).  The empty asm keeps the paths apart.
__device__ __forceinline__ float2 load_source(bool in_lds, const float2* lds, const float2* mem, int i) {
  float2 v;
  if (in_lds) {
    v = lds[i];
  } else {
    v = mem[i];
    asm volatile("" : "+v"(v.x), "+v"(v.y));
  }
  return v;
}

__device__ __forceinline__ double2 load_source64(bool in_lds, const double2* lds, const double2* mem, int i) {
  double2 v;
  if (in_lds) {
    v = lds[i];
  } else {
    v = mem[i];
    asm volatile("" : "+v"(v.x), "+v"(v.y));
  }
  return v;
}

// byte offset of sorted target j inside a slot's records
__device__ __forceinline__ unsigned rec_off(int j) { return (unsigned)(j >> 4) * kQRec + (unsigned)(j & 15) * 8u; }

// One chunk against ONE point of this lane (a gather: every lane its own chunk), all eight pair reads in flight before the
// first distance.  bd / pos: the smallest distance so far and the sorted position of its target.
// TRACK: sec is the second smallest distance over everything evaluated for the point (v_med3) — inside the near-tie window ⇔ float64
//   must decide (float64-exact mode), equal to bd ⇔ two targets share the minimum, and every later certificate rests on it.
// !TRACK (float32 mode without certificates): a pair's nearer target enters one compare against the best, and sec only records the
//   VALUE at the last event "a distance equals the best so far" or "both distances of a pair are equal and take the lead".  The
//   best only decreases, so after the search sec == bd exactly when two targets share the final minimum.
template <bool TRACK>
__device__ __forceinline__ void queue_eval_chunk(unsigned tile0, int chunk, float x, float y, float& bd, float& sec, int& pos) {
  typedef float __attribute__((ext_vector_type(4))) Vec4;
  typedef const __attribute__((address_space(3))) Vec4* LdsPair;
  const unsigned rec = tile0 + (unsigned)chunk * kQRec;
  Vec4 v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = *(LdsPair)(size_t)(rec + (unsigned)c * 16u);
  __builtin_amdgcn_sched_barrier(0);
  int kb = -1;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float d0 = dist2(x, y, v[c].x, v[c].y);
    const float d1 = dist2(x, y, v[c].z, v[c].w);
    if constexpr (TRACK) {
      const bool lt0 = d0 < bd;
      sec = __builtin_amdgcn_fmed3f(bd, sec, d0);
      kb = lt0 ? 2 * c : kb;
      bd = lt0 ? d0 : bd;
      const bool lt1 = d1 < bd;
      sec = __builtin_amdgcn_fmed3f(bd, sec, d1);
      kb = lt1 ? 2 * c + 1 : kb;
      bd = lt1 ? d1 : bd;
    } else {
      const bool c01 = d1 < d0;
      const float mn = c01 ? d1 : d0;
      const int kk = c01 ? 2 * c + 1 : 2 * c;
      const bool lt = mn < bd;
      const bool ev = (mn == bd) | ((d0 == d1) & lt);
      sec = ev ? mn : sec;
      kb = lt ? kk : kb;
      bd = lt ? mn : bd;
    }
  }
  pos = kb >= 0 ? chunk * kPruneChunk + kb : pos;
}

#ifdef LSLAM_QUEUE_PROFILE
#define QUEUE_MARK(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); prof[k] += now_ - pt; pt = now_; } while (0)
#else
#define QUEUE_MARK(k) do {} while (0)
#endif

// CERT: sources keep their match on a proof (margins; see the header) instead of being searched in every iteration
// INCR (needs CERT, no normalised error): running sums instead of a pass over all pairs per iteration — see `finish`
template <bool F64, bool CERT, bool INCR>
__global__ void __launch_bounds__(kIcpBatchMaxThreads, 4)      // ≤ 128 VGPRs: one workgroup of up to 16 waves per CU (the VALU is saturated by four waves per SIMD: tools/queue_stats.sh)
icp_queue_kernel(IcpBatchArgs args, PruneScratch sc, QueueLayout lay, int S) {
  constexpr int CHUNK = kPruneChunk;
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int Q = lay.Q;
  const bool box = (args.flags & kFlagNormalize) != 0;
  const int RS = box ? 13 : 9;                                 // doubles per row: Σp(2) Σq(2) Σpqᵀ(4) Σ‖p−q‖² [+ lo(2) hi(2)]
  int* sh_fail = reinterpret_cast<int*>(smem + (size_t)Q * kQCtlBytes);
  int* sh_beat = sh_fail + 1;                                  // bumped by a wave that takes iterations on its own: the idle ones see that the workgroup is alive
  uint32_t* tickets = reinterpret_cast<uint32_t*>(smem + (size_t)Q * kQCtlBytes + 64);       // [16]
  int32_t* states = reinterpret_cast<int32_t*>(smem + (size_t)Q * kQCtlBytes + 128);         // [16]
  auto open_slot = [&](int q, int ntasks) { __hip_atomic_store(&tickets[q], (unsigned)ntasks << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  auto close_slot = [&](int q) { __hip_atomic_store(&tickets[q], kQClosed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  auto ctl_of = [&](int q) { return reinterpret_cast<QueueCtl*>(smem + (size_t)q * kQCtlBytes); };
  auto slot_of = [&](int q) { return smem + lay.ctl_bytes + (size_t)q * lay.slot_bytes; };
#ifdef LSLAM_QUEUE_PROFILE
  unsigned long long prof[6] = {0, 0, 0, 0, 0, 0}, pt = __builtin_amdgcn_s_memtime();
  unsigned long long cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // [6] cycles inside search_points, [7] tasks run; searches of ≤ 64 sources, surviving boxes, list rounds, (point, chunk) list entries, sources visited, sources searched
#define QUEUE_COUNT(k, v) do { cnt[k] += (v); } while (0)
#else
#define QUEUE_COUNT(k, v) do {} while (0)
#endif

  if (tid < 16) {
    tickets[tid] = kQClosed;
    states[tid] = tid < Q ? 2 : 0;
    if (tid < Q) { QueueCtl* c = ctl_of(tid); c->done_tasks = 0; c->problem = -1; }
  }
  if (tid == 0) { *sh_fail = 0; *sh_beat = 0; }
  __syncthreads();                                             // the only barrier of the kernel

  // An iteration of a problem is two rounds of tasks.  SEARCH tasks (phase 0) cover Ga consecutive slabs each and leave the
  // matches in LDS: small, so that a problem's waves finish together and a closing problem leaves few waves without work (the
  // search is exact: its grouping cannot change a result).  MOMENT tasks (phase 1) cover moment_slabs() slabs each and leave one
  // row of float64 partial sums: large, one cross-lane reduction per 512 pairs or more; a LiDAR-sized problem (up to 1280 sources)
  // has one, run by the wave that completed the last search task.  Row boundaries depend on the problem's size alone — not on the batch, the slot, or
  // timing — so every bit of a pair's result is the same wherever and whenever the pair is solved (tests/test_gpu_shard.py).
  // (first iteration: every source is searched — one or two slabs per task; later most keep their match and a task gathers the rest
  // of four or eight slabs into full waves)
  auto search_group = [&](int nslabs, int it) -> int {
    if (lay.G > 0) return lay.G;
    if (it == 0 || !CERT) return nslabs > 32 ? 2 : 1;
    const int fill = nslabs * Q / nwaves;                      // every wave a task when the slots open together
    const int most = nslabs > 32 ? 8 : 4;
    return fill < 2 ? 2 : (fill > most ? most : fill);
  };
#ifndef QEXP_SLEEP
#define QEXP_SLEEP 8
#endif
#ifndef QEXP_MOM
#define QEXP_MOM 20
#endif
#ifndef QEXP_PAIR
#define QEXP_PAIR (F64 ? 1 : 2)
#endif
  // Above QEXP_MOM slabs a row is the sum of QEXP_PAIR (1 or 2) tasks: each adds its partial sums into the zeroed row with LDS float64 atomics.
  // 0 + a is a, and a + b is b + a to the bit, so which of the two arrives first cannot change the row; twice the tasks at the same LDS.
  auto moment_slabs = [&](int nslabs) -> int { return nslabs <= QEXP_MOM ? (nslabs > 0 ? nslabs : 1) : (nslabs <= 64 ? queue_row_slabs(F64) : 2 * queue_row_slabs(F64)) / QEXP_PAIR; };
  auto moment_tasks = [&](int nslabs) -> int { return nslabs > 0 ? (nslabs + moment_slabs(nslabs) - 1) / moment_slabs(nslabs) : 1; };
  auto moment_rows = [&](int nslabs) -> int { return nslabs <= QEXP_MOM ? 1 : (moment_tasks(nslabs) + QEXP_PAIR - 1) / QEXP_PAIR; };
  // rows that tasks ADD into start from the neutral element (one wave; the caller publishes the stores with its release fence)
  auto clear_rows = [&](unsigned char* base, int nslabs) {
    if (INCR || QEXP_PAIR == 1 || nslabs <= QEXP_MOM) return;
    double* rows = reinterpret_cast<double*>(base + lay.rows_off);
    for (int k = lane; k < moment_rows(nslabs) * RS; k += kWave) { const int cl = k % RS; rows[k] = cl < 9 ? 0.0 : (cl < 11 ? (double)INFINITY : -(double)INFINITY); }
  };

  // running-sums build: the first iteration's round of sums — tasks of 8 slabs (16 above 64 slabs), one row of kQBase DD sums each
  auto base_slabs = [&](int nslabs) -> int { return nslabs <= 8 * kQBaseRows ? 8 : 16; };
  auto base_tasks = [&](int nslabs) -> int { return nslabs > 0 ? (nslabs + base_slabs(nslabs) - 1) / base_slabs(nslabs) : 1; };

  // margins: in LDS; in float64-exact mode (whose slot also holds the targets' float64 residuals) in this workgroup's rows of a scratch
  // buffer in memory, read and written a slab at a time — the workgroup's waves share the CU's L1, workgroup-scope fences order them
  auto margins_of = [&](int q, unsigned char* base) -> uint16_t* {
    if constexpr (F64 || INCR) return lay.margins + ((size_t)blockIdx.x * (size_t)Q + (size_t)q) * lay.margin_stride;
    else return reinterpret_cast<uint16_t*>(base + lay.margin_off);
  };

  // running-sums build: a source's certificate as an EXPIRY — the value of the slot's `moved` at which it runs out — one float per source
  // in the same scratch rows; checking a source is then a load and a compare, and nothing is written back
  // (a slot's row of the scratch buffer in that build: margin_stride floats of expiries, then margin_stride uint16 of matches — the matches
  // left LDS as well, so that a fourth 4096-point problem fits a CU)
  auto expiry_of = [&](int q) -> float* { return reinterpret_cast<float*>(lay.margins + ((size_t)blockIdx.x * (size_t)Q + (size_t)q) * (size_t)lay.margin_stride * 3); };
  auto prevs_of = [&](int q, unsigned char* base) -> uint16_t* {
    if constexpr (INCR) return lay.margins + ((size_t)blockIdx.x * (size_t)Q + (size_t)q) * (size_t)lay.margin_stride * 3 + (size_t)lay.margin_stride * 2;
    else return reinterpret_cast<uint16_t*>(base + lay.prev_off);
  };
  // A source within rounding of equidistant between two targets gets no margin, however small the steps become: its block would be searched
  // in every iteration and its problem would never have an iteration without a round (C4: 75 of 4096 problems, a millisecond each instead
  // of 0.4).  The search is a function of the float32 position it is given: such a source is listed (up to kQContested per problem) with
  // that position, its expiry set to +inf, and the finisher recomputes the position with every new transform — bit-identical: the search
  // would return what it returned; changed: the source's expiry is zeroed and its block visited.  Not in float64-exact mode when the
  // runner-up is within 1e-4 of the near-tie window (the window moves in its last bits with T).
  auto contested_of = [&](unsigned char* base, unsigned*& count, uint16_t*& idx, float2*& pos) {
    unsigned char* c0 = reinterpret_cast<unsigned char*>(reinterpret_cast<double*>(base + lay.rows_off) + kQIncrRows + kQDeltaRows / 2);
    count = reinterpret_cast<unsigned*>(c0);
    pos = reinterpret_cast<float2*>(c0 + 8);
    idx = reinterpret_cast<uint16_t*>(c0 + 8 + 8 * kQContested);
  };
  auto later_group = [&](int nslabs) -> int {                // slabs per task after the first iteration: a function of the problem's size alone
    int g = lay.G > 0 ? lay.G : (nslabs > 32 ? 8 : 4);
    if (g * kQDeltaRows < nslabs) g = (nslabs + kQDeltaRows - 1) / kQDeltaRows;
    return g;
  };

  // ---- one wave: draw the next problem and stage it into slot q (or mark the slot empty for good) ----
  auto refill = [&](int q) {
    QueueCtl* c = ctl_of(q);
    int p = 0;
    if (lane == 0) p = (int)atomicAdd(&lay.queue[lay.draw], 1u);
    p = __builtin_amdgcn_readfirstlane(p);
    // (a launch over one class of problems: their number was counted by icp_sort_kernel, their indices are in `order`)
    const int avail = lay.cls < 0 ? S : (int)__builtin_amdgcn_readfirstlane(__hip_atomic_load(&lay.queue[48 + lay.cls], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (p < avail && lay.cls >= 0) p = __builtin_amdgcn_readfirstlane(lay.order[lay.cls ? S - 1 - p : p]);
    else if (p >= avail) p = S;
    if (p >= S) {
      if (lane == 0) __hip_atomic_store(&states[q], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // the ticket stays closed
      return;
    }
    unsigned char* base = slot_of(q);
    int2* tdelta = reinterpret_cast<int2*>(base + lay.tdelta_off);
    float2* sh_src = reinterpret_cast<float2*>(base + lay.src_off);
    double2* sh_src64 = reinterpret_cast<double2*>(base + lay.src_off);
    uint16_t* sh_prev = prevs_of(q, base);
    uint16_t* sh_margin = margins_of(q, base);
    const bool src_lds = lay.src_off != 0xffffffffu;
    const int64_t s0 = args.src_off[p], t0 = args.tgt_off[p];
    const int n = (int)(args.src_off[p + 1] - s0), m = (int)(args.tgt_off[p + 1] - t0);
    const int64_t ss = (int64_t)p * sc.src_stride, ts = (int64_t)p * sc.tgt_stride;
    const int nchunks = (m + CHUNK - 1) / CHUNK, nslabs = (n + kWave - 1) / kWave;
    float tmax = 0.f, smax = 0.f;
    float bx0 = INFINITY, by0 = INFINITY, bx1 = -INFINITY, by1 = -INFINITY;      // the target's box
    for (int j0 = 0; j0 < nchunks * CHUNK; j0 += kQStage * kWave) {      // eight loads in flight per lane: one at a time, a 4096-point cloud took 64 round trips to memory
      float2 tv[kQStage];
#pragma unroll
      for (int u = 0; u < kQStage; ++u) { const int j = j0 + u * kWave + lane; tv[u] = (j < m) ? sc.tgt_sorted[ts + j] : make_float2(kQPad * (1.f + 0.0625f * (float)(j & 15)), 0.f); }
#pragma unroll
      for (int u = 0; u < kQStage; ++u) {
        const int j = j0 + u * kWave + lane;
        const float2 v = tv[u];
        if (j < nchunks * CHUNK) *reinterpret_cast<float2*>(base + rec_off(j)) = v;
        if ((F64 || CERT) && j < m) {
          tmax = fmaxf(tmax, __builtin_fmaf(v.y, v.y, v.x * v.x));
          if constexpr (CERT) { bx0 = fminf(bx0, v.x); by0 = fminf(by0, v.y); bx1 = fmaxf(bx1, v.x); by1 = fmaxf(by1, v.y); }
        }
      }
      if constexpr (F64 && !INCR) {
#pragma unroll
        for (int h = 0; h < kQStage; h += 4) {                       // (four float64 points in flight: eight spill)
          double2 td[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { const int j = j0 + (h + u) * kWave + lane; td[u] = (j < m) ? sc.tgt64_sorted[ts + j] : make_double2(0.0, 0.0); }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int j = j0 + (h + u) * kWave + lane;
            if (j < nchunks * CHUNK) tdelta[j] = (j < m) ? make_int2(residual_count(td[u].x, tv[h + u].x), residual_count(td[u].y, tv[h + u].y)) : make_int2(0, 0);
          }
        }
      }
    }
    // sources (all loads in flight before the first use), no matches and no margins yet
    for (int i0 = 0; src_lds && i0 < n; i0 += 8 * kWave) {
      if constexpr (F64) {
        double2 pd[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * kWave + lane; pd[u] = i < n ? sc.src64_sorted[ss + i] : make_double2(0.0, 0.0); }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * kWave + lane; if (i < n && src_lds) sh_src64[i] = pd[u]; }
      } else if (src_lds) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * kWave + lane; v[u] = i < n ? sc.src_sorted[ss + i] : make_float2(0.f, 0.f); }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * kWave + lane; if (i < n) sh_src[i] = v[u]; }
      }
    }
    for (int i = lane; i < n; i += kWave) {
      if constexpr (!INCR) sh_prev[i] = 0xffff;                 // (running sums: the first iteration does not look at the matches)
      if constexpr (CERT && !INCR) sh_margin[i] = 0;
    }
    if constexpr (INCR) {
      float* tmin = reinterpret_cast<float*>(reinterpret_cast<double*>(base + lay.rows_off) + kQIncrRows);
      if (lane < kQDeltaRows) tmin[lane] = INFINITY;
      unsigned* ccount; uint16_t* cidx; float2* cpos;
      contested_of(base, ccount, cidx, cpos);
      if (lane == 0) *ccount = 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    // the largest |p| of the cloud: what a change of the transform can move a source by (margins), the near-tie window (float64-exact mode)
    if constexpr (INCR) smax = sc.smax[p];                    // (running sums: the sources are not on chip — icp_sort_kernel left the figure)
    else if constexpr (F64 || CERT) {
      for (int i0 = 0; i0 < n; i0 += 8 * kWave) {              // eight loads in flight per lane
        float2 sv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + u * kWave + lane;
          sv[u] = make_float2(0.f, 0.f);
          if (i < n) {
            if constexpr (F64 && !INCR) { const double2 pd = load_source64(src_lds, sh_src64, sc.src64_sorted + ss, i); sv[u] = make_float2((float)pd.x, (float)pd.y); }
            else sv[u] = load_source(src_lds, sh_src, sc.src_sorted + ss, i);      // (running sums: the float32 clouds ARE the roundings of the float64 ones — lslam.h)
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) smax = fmaxf(smax, __builtin_fmaf(sv[u].y, sv[u].y, sv[u].x * sv[u].x));
      }
      smax = wave_max_dpp(smax);
    }
    for (int cidx = lane; cidx < nchunks; cidx += kWave) {     // boxes over the real points only
      float x0 = INFINITY, y0 = INFINITY, x1 = -INFINITY, y1 = -INFINITY;
      const float2* rec = reinterpret_cast<const float2*>(base + (size_t)cidx * kQRec);
      for (int k = 0; k < CHUNK && cidx * CHUNK + k < m; ++k) {
        const float2 qv = rec[k];
        x0 = fminf(x0, qv.x); y0 = fminf(y0, qv.y); x1 = fmaxf(x1, qv.x); y1 = fmaxf(y1, qv.y);
      }
      *reinterpret_cast<float4*>(base + (size_t)cidx * kQRec + CHUNK * 8) = make_float4(x0, y0, x1, y1);
    }
    if constexpr (F64 || CERT) tmax = wave_max_dpp(tmax);
    if constexpr (CERT) { bx0 = wave_min_dpp(bx0); by0 = wave_min_dpp(by0); bx1 = wave_max_dpp(bx1); by1 = wave_max_dpp(by1); }
    if (lane < 16) c->T[lane] = args.init_T[(int64_t)16 * p + lane];
    clear_rows(base, nslabs);
    if (lane == 0) __hip_atomic_store(&states[q], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int G = search_group(nslabs, 0);
    if (lane == 0) {
      StopState st0; stop_init(st0);
      c->stop = st0; c->err = 0.0;
      c->problem = p; c->n = n; c->m = m; c->nchunks = nchunks; c->nslabs = nslabs;
      c->G = G; c->phase = 0;
      c->done_tasks = 0; c->dirty = 0u; c->rerun = 0xffffffffu; c->moved = 0.f; c->Gb = INCR ? later_group(nslabs) : 0; c->rounds = 0u;
      c->Rt = sqrtf(tmax) * 1.0001f;
      c->Rs = sqrtf(smax) * 1.0001f;
      c->stepR = INFINITY; c->stepT = INFINITY;               // no earlier transform: nothing is certified in the first iteration
      // E is capped at a fraction of the targets' mean spacing (what the extended reach costs grows with E / spacing)
      const float area = CERT ? fmaxf(bx1 - bx0, 0.f) * fmaxf(by1 - by0, 0.f) : 0.f;
      c->ecap = CERT ? lay.ecap_frac * sqrtf(area / (float)(m > 0 ? m : 1)) : 0.f;
      // the first searches extend their reach by a share of the cap: their margins then outlast a first step of that size (without it a
      // source whose seed is already its nearest target leaves the first iteration with no margin at all)
      c->extend = CERT ? lay.e0_frac * c->ecap : 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) open_slot(q, nslabs > 0 ? (nslabs + G - 1) / G : 1);
  };

  // ---- one wave: the slot's last task of this iteration is done — reduce, solve, stop rules; outputs and refill on convergence ----
  auto finish = [&](int q) {
    QueueCtl* c = ctl_of(q);
    unsigned char* base = slot_of(q);
    double* rows = reinterpret_cast<double*>(base + lay.rows_off);       // [queue_rows][RS]
    const int n = lds_load_i32(&c->n), nslabs = lds_load_i32(&c->nslabs);
    // running-sums build: the sums after this round — the first iteration's rows, or the rows of changes the searches left (the sums do not
    // change in the iterations this wave takes by itself below: taken once, ahead of the loop)
    double B[kQBase];
    for (int kk = 0; kk < kQBase; ++kk) B[kk] = 0.0;
    double* sums = rows;                                       // [kQBase] DD
    if constexpr (INCR) {
    double* drow = rows + 2 * kQBase;                        // the rows the round's tasks left
    const bool full = !(__int_as_float(lds_load_i32(reinterpret_cast<const int32_t*>(&c->stepR))) < INFINITY);
    const int k = lane < kQBase ? lane : 0;
    DD v;
    if (full) {
      const int nt = base_tasks(nslabs);
      v = DD{drow[2 * k], drow[2 * k + 1]};
      for (int r = 1; r < nt; ++r) dd_add(v, DD{drow[(size_t)r * 2 * kQBase + 2 * k], drow[(size_t)r * 2 * kQBase + 2 * k + 1]});
    } else {
      v = DD{sums[2 * k], sums[2 * k + 1]};
      unsigned dirty = lds_load_u32(&c->dirty);
      const int kd = k - (kQBase - kQDelta);
      while (dirty) {                                        // task order: which wave ran which task does not matter
        const int t = __builtin_ctz(dirty);
        dirty &= dirty - 1u;
        if (kd >= 0) dd_add(v, DD{drow[(size_t)t * 2 * kQDelta + 2 * kd], drow[(size_t)t * 2 * kQDelta + 2 * kd + 1]});
      }
    }
    { double s_, e_; two_sum(v.hi, v.lo, s_, e_); v.hi = s_; v.lo = e_; }      // hi = the sum rounded to a double
    if (lane < kQBase) { sums[2 * lane] = v.hi; sums[2 * lane + 1] = v.lo; }
#pragma unroll
    for (int kk = 0; kk < kQBase; ++kk)
      B[kk] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v.hi), kk), __builtin_amdgcn_readlane(__double2loint(v.hi), kk));
    }
    double tn_out = 0.0, err_out = 0.0;
    StopState st_out;
    for (;;) {                                                 // (running-sums build: iterations in which no certificate runs out are taken here, one after the other)
    Moments mo;
    mo.n = (double)n;                                          // every source has a match: the count is the cloud's size
    for (int i = 0; i < 3; ++i) { mo.sp[i] = 0.0; mo.sq[i] = 0.0; mo.lo[i] = 0.0; mo.hi[i] = 0.0; }
    for (int i = 0; i < 9; ++i) mo.spq[i] = 0.0;
    double* scratch = rows;                                    // 16 doubles the solve may use once the rows are consumed
    if constexpr (INCR) {
      // ---- running sums.  The transform enters the moments of the matched pairs only through products with sums that do not
      // depend on it: with u_i = (T_i0, T_i1, T_i3), P = Σ (x, y, 1)(x, y, 1)ᵀ over the ORIGINAL sources and c_i = (Σ x·q_i, Σ y·q_i, Σ q_i),
      //   Σ (Tp)_i = u_i·P_2,   Σ (Tp)_i q_j = u_i·c_j,   Σ ‖Tp − q‖² = Σ|q|² + Σ_i (u_iᵀ P u_i − 2 u_i·c_i).
      // P is fixed, c and Σ|q|² change only where a match changes: the first iteration sums them over all pairs (base_task), later
      // ones add the changes their searches found (search_points).  The last formula cancels (terms ~ n·r², result ~ n·residual²), so
      // the sums are kept as double-doubles and the 25 products are formed exactly: the error is that of the exact sum for this T —
      // closer to it than a per-pair pass in float64 — and the per-iteration pass over all pairs is gone.
      scratch = rows + 2 * kQBase;
      const double T00 = c->T[0], T01 = c->T[1], T03 = c->T[3], T10 = c->T[4], T11 = c->T[5], T13 = c->T[7];
      mo.sp[0] = T00 * B[0] + T01 * B[1] + T03 * mo.n; mo.sp[1] = T10 * B[0] + T11 * B[1] + T13 * mo.n;
      mo.sq[0] = B[5]; mo.sq[1] = B[6];
      mo.spq[0] = T00 * B[7] + T01 * B[9] + T03 * B[5]; mo.spq[1] = T00 * B[8] + T01 * B[10] + T03 * B[6];
      mo.spq[3] = T10 * B[7] + T11 * B[9] + T13 * B[5]; mo.spq[4] = T10 * B[8] + T11 * B[10] + T13 * B[6];
      // Σ‖Tp − q‖²: lane L < 24 forms one product of row i = L / 12 (nine u_a·u_b·P_ab, three −2·u_a·c_a), lane 24 brings Σ|q|²
      DD term{0.0, 0.0};
#ifdef QX_NOTERM
      if (false) {
#else
      if (lane < 25) {
#endif
        const int i = lane >= 12 ? 1 : 0, r = lane - 12 * i;
        const int a = r < 9 ? r / 3 : r - 9, b = r < 9 ? r - 3 * (r / 3) : 0;
        constexpr unsigned long long kPIdx = 2ull | 3ull << 4 | 0ull << 8 | 3ull << 12 | 4ull << 16 | 1ull << 20 | 0ull << 24 | 1ull << 28 | 15ull << 32;
        int idx = r < 9 ? (int)((kPIdx >> (4 * (3 * a + b))) & 15ull) : (a == 0 ? 7 + i : (a == 1 ? 9 + i : 5 + i));
        double f1 = c->T[4 * i + (a < 2 ? a : 3)];
        double f2 = r < 9 ? c->T[4 * i + (b < 2 ? b : 3)] : -2.0;
        if (lane == 24) { idx = 11; f1 = 1.0; f2 = 1.0; }
        const DD D = idx == 15 ? DD{mo.n, 0.0} : DD{sums[2 * idx], sums[2 * idx + 1]};
        double ph, pl, h, l;
        two_prod(f1, f2, ph, pl);
        two_prod(ph, D.hi, h, l);
        l += ph * D.lo + pl * D.hi;
        term = DD{h, l};
      }
      term = wave_sum_dd_last(term);
      const double sse_l = term.hi + term.lo;
      const double sse = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(sse_l), 63), __builtin_amdgcn_readlane(__double2loint(sse_l), 63));
      mo.sse = sse < 0.0 ? 0.0 : sse;
      if (n == 1) {                                            // one pair: the centred cross-covariance is zero by definition, not to rounding (S = spq − sp·sqᵀ/n)
        mo.spq[0] = mo.sp[0] * mo.sq[0]; mo.spq[1] = mo.sp[0] * mo.sq[1]; mo.spq[3] = mo.sp[1] * mo.sq[0]; mo.spq[4] = mo.sp[1] * mo.sq[1];
      }
    } else {
    const int ntasks = moment_rows(nslabs);                    // rows of partial sums
    double col;
    {
      const int cl = lane < RS ? lane : 0;
      const bool is_sum = lane < 9, is_min = lane < 11;
      col = rows[cl];
      for (int w0 = 1; w0 < ntasks; w0 += 4) {                  // four rows per trip, their loads issued together; fixed order
        double x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = rows[(size_t)(w0 + u < ntasks ? w0 + u : 0) * RS + cl];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (!box) { col = (w0 + u < ntasks) ? col + x[u] : col; continue; }
          const double s_ = col + x[u], lo_ = fmin(col, x[u]), hi_ = fmax(col, x[u]);
          col = (w0 + u < ntasks) ? (is_sum ? s_ : (is_min ? lo_ : hi_)) : col;
        }
      }
    }
    double tot[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) {
      const int lo = __builtin_amdgcn_readlane(__double2loint(col), k);
      const int hi = __builtin_amdgcn_readlane(__double2hiint(col), k);
      tot[k] = __hiloint2double(hi, lo);
    }
    mo.sp[0] = tot[0]; mo.sp[1] = tot[1]; mo.sq[0] = tot[2]; mo.sq[1] = tot[3];
    mo.spq[0] = tot[4]; mo.spq[1] = tot[5]; mo.spq[3] = tot[6]; mo.spq[4] = tot[7];
    mo.sse = tot[8];
    if (box) { mo.lo[0] = tot[9]; mo.lo[1] = tot[10]; mo.hi[0] = tot[11]; mo.hi[1] = tot[12]; }
    }
    double R[9], pb[3], qb[3];
    step_parts(mo, true, R, pb, qb);
    const double told = c->T[lane & 15];
    const double tn = step_times_prev_lane(R, pb, qb, c->T, scratch, lane);   // element `lane` of T_step·T_prev (the rows are consumed: their first 16 doubles serve as its scratch)
    // how far the new transform can move a source p: ‖ΔR‖₂·|p| + |Δt|, rounded up
    float stepR = INFINITY, stepT = INFINITY;
    if constexpr (CERT) {
      // the differences in float64 (the entries agree to many digits), the norms in float32 — scaled by the largest entry so that nothing
      // underflows, and rounded up by far more (1e-5) than float32 arithmetic on numbers of size one can lose
      const float dl = fabsf((float)(tn - told));
      auto at = [&](int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dl), k)); };
      const float sR = fmaxf(fmaxf(at(0), at(1)), fmaxf(at(4), at(5))), sT = fmaxf(at(3), at(7));
      const float sd = (float)(tn - told);
      auto sg = [&](int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sd), k)); };
      const float iR = sR > 0.f ? 1.f / sR : 0.f, iT = sT > 0.f ? 1.f / sT : 0.f;
      const float d00 = sg(0) * iR, d01 = sg(1) * iR, d10 = sg(4) * iR, d11 = sg(5) * iR, d03 = sg(3) * iT, d13 = sg(7) * iT;
      // the largest singular value of the 2 x 2 block ΔR (for a pure rotation step: 2 sin(Δθ/2), a factor √2 below the Frobenius norm)
      const float f2 = d00 * d00 + d01 * d01 + d10 * d10 + d11 * d11, g1 = d00 * d00 + d01 * d01 - d10 * d10 - d11 * d11, g2 = d00 * d10 + d01 * d11;
      stepR = sqrtf(0.5f * (f2 + sqrtf(g1 * g1 + 4.f * g2 * g2))) * sR * 1.00001f + 1e-30f;
      stepT = sqrtf(d03 * d03 + d13 * d13) * sT * 1.00001f + 1e-30f;
    }
    const double err = error_from_moments(mo, box);
    StopState st = c->stop;
    const bool stop = stop_after_iteration(st, err, args.epsilon, args.stopping_thresh, args.max_iters);
    if (lane < 16) c->T[lane] = tn;
    QUEUE_MARK(2);
    if (!stop) {
      // from the second iteration on, with certificates: one round of fused tasks (a moment group's slabs searched, then its row)
      const bool fuse = CERT && !INCR && lay.fuse && moment_tasks(nslabs) > 1;
      int G = fuse ? moment_slabs(nslabs) : search_group(nslabs, st.executed);
      if constexpr (INCR) G = later_group(nslabs);              // a task may leave a row of changes: at most kQDeltaRows tasks, and a grouping that depends on the problem's size alone
      clear_rows(base, nslabs);
      // running-sums build: the blocks in which some certificate runs out before the next iteration's transform — none: nothing can
      // change a match, the sums stand, and the next iteration is this wave's next trip through the loop
      unsigned rerun = 0xffffffffu;
      float moved_next = 0.f;
      if constexpr (INCR) {
        const float step = __builtin_fmaf(stepR, c->Rs, stepT) * 1.00002f;      // every source of the cloud moves by less
        moved_next = next_up(c->moved + step);
        float* tmin = reinterpret_cast<float*>(rows + kQIncrRows);
        const int nblocks = nslabs > 0 ? (nslabs + G - 1) / G : 1;
        const bool mine = lane < nblocks && !(tmin[lane < kQDeltaRows ? lane : 0] > moved_next);
        rerun = stepR < INFINITY ? (unsigned)__ballot(mine) : 0xffffffffu;
        if (stepR < INFINITY && mine) tmin[lane] = INFINITY;   // the block's task will leave the new minimum
        // the contested sources: does the new transform still put each of them on the float32 position it was searched at?
        {
          unsigned* ccount; uint16_t* cidx; float2* cpos;
          contested_of(base, ccount, cidx, cpos);
          const unsigned craw = lds_load_u32(ccount);
          const unsigned cc = craw < (unsigned)kQContested ? craw : (unsigned)kQContested;
          if (!(stepR < INFINITY)) {                           // a round over everything: whoever is still contested is listed again
            if (lane == 0) *ccount = 0u;
          } else if (cc) {
            auto tl = [&](int k) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tn), k), __builtin_amdgcn_readlane(__double2loint(tn), k)); };
            const double n00 = tl(0), n01 = tl(1), n03 = tl(3), n10 = tl(4), n11 = tl(5), n13 = tl(7);
            const float2* __restrict__ gsrc = sc.src_sorted + (int64_t)lds_load_i32(&c->problem) * sc.src_stride;
            int ci = 0;
            bool gone = false;
            float2 lp = make_float2(0.f, 0.f);
            if (lane < (int)cc) {
              ci = cidx[lane]; lp = cpos[lane];
              const float2 v = gsrc[ci];
              const float nax = (float)(n00 * (double)v.x + n01 * (double)v.y + n03);      // the expression of search_points, operation for operation
              const float nay = (float)(n10 * (double)v.x + n11 * (double)v.y + n13);
              gone = !(nax == lp.x && nay == lp.y);
            }
            const unsigned long long gm = __ballot(gone);
            if (gm) {
              float* expiry = expiry_of(q);
              if (gone) expiry[ci] = 0.f;
              unsigned extra = 0u;
              for (unsigned long long m = gm; m; m &= m - 1ull) extra |= 1u << ((__builtin_amdgcn_readlane(ci, __builtin_ctzll(m)) / kWave) / G);
              rerun |= extra;
              if (lane < nblocks && ((extra >> lane) & 1u)) tmin[lane] = INFINITY;
              const bool stays = lane < (int)cc && !gone;
              const unsigned long long km = __ballot(stays);
              const int to = __builtin_popcountll(km & ((1ull << lane) - 1ull));
              if (stays) { cidx[to] = (uint16_t)ci; cpos[to] = lp; }
              if (lane == 0) *ccount = (unsigned)__builtin_popcountll(km);
            } else if (lane == 0 && craw != cc) *ccount = cc;
          }
        }
      }
      if (lane == 0) {
        c->stop = st; c->err = err; c->done_tasks = 0; c->dirty = 0u;
        c->G = G; c->phase = fuse ? 2 : 0;
        c->stepR = stepR; c->stepT = stepT;
        c->rerun = rerun; c->moved = moved_next;
        // the next searches extend their reach by E ≈ kQExtend × this step (the steps of a converging run shrink geometrically:
        // margins of E / 2 then outlast what is still to come); nothing while the steps exceed what E may cost
        const float moved = stepR * c->Rs + stepT, cap = c->ecap;
        // … and never less than a share of the cap: a source searched late in the run, when the steps are tiny, would otherwise get a margin
        // of E/2 below the rounding slack — none — and be searched again in every iteration left
        c->extend = CERT ? (moved <= cap ? fminf(fmaxf(kQExtend * moved, lay.efloor_frac * cap), cap) : 0.f) : 0.f;
      }
      if (INCR && lay.debug && lane == 0 && stepR < INFINITY) {
        if (rerun) { atomicAdd(&lay.queue[3], 1u); atomicAdd(&lay.queue[32], (unsigned)__builtin_popcount(rerun)); c->rounds += 1u; } else atomicAdd(&lay.queue[33], 1u);
      }
      if (INCR && rerun == 0u) {
        if (lane == 0) __hip_atomic_fetch_add(sh_beat, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        continue;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) open_slot(q, nslabs > 0 ? (nslabs + G - 1) / G : 1);                                    // next iteration
      return;
    }
    tn_out = tn; err_out = err; st_out = st;
    break;
    }
    const double tn = tn_out, err = err_out;
    const StopState st = st_out;
    const int p = lds_load_i32(&c->problem);
    if (INCR && lay.debug && lane == 0) { const unsigned r_ = c->rounds; atomicAdd(&lay.queue[40 + (r_ < 8u ? 0 : (r_ < 16u ? 1 : (r_ < 32u ? 2 : 3)))], 1u); }
    if (lane < 16) args.out_T[(int64_t)16 * p + lane] = tn;
    if (lane == 0) { args.out_err[p] = err; args.out_iters[p] = st.executed; }
    if (args.out_corr) {                                       // correspondences of the LAST iteration (icp.py:117, 187-189), caller's numbering
      const int64_t s0 = args.src_off[p];
      const int64_t ss = (int64_t)p * sc.src_stride, ts = (int64_t)p * sc.tgt_stride;
      const uint16_t* sh_prev = prevs_of(q, base);
      for (int i = lane; i < n; i += kWave) args.out_corr[s0 + sc.src_index[ss + i]] = sc.tgt_index[ts + sh_prev[i]];
    }
    refill(q);
    QUEUE_MARK(3);
  };

  // ---- one wave: search task t of slot q — the nearest neighbours of G slabs of 64 sources, left in LDS ----
  auto search_task = [&](int q, int t) {
    QueueCtl* c = ctl_of(q);
    unsigned char* base = slot_of(q);
    const float2* sh_src = reinterpret_cast<const float2*>(base + lay.src_off);
    const double2* sh_src64 = reinterpret_cast<const double2*>(base + lay.src_off);
    uint16_t* sh_prev = prevs_of(q, base);
    uint16_t* sh_margin = margins_of(q, base);
    uint16_t* gathered = reinterpret_cast<uint16_t*>(smem + (size_t)Q * kQCtlBytes + 192) + wave * kWave;      // this wave's list of sources to search
    const bool src_lds = lay.src_off != 0xffffffffu;
    const int cw = ctl_load(c, lane);
    const int n = CTL_INT(cw, n), m = CTL_INT(cw, m), nchunks = CTL_INT(cw, nchunks), nslabs = CTL_INT(cw, nslabs);
    const int G = CTL_INT(cw, G);
    const int p = CTL_INT(cw, problem);
    const float stepR = CERT ? CTL_FLOAT(cw, stepR) : INFINITY, stepT = CERT ? CTL_FLOAT(cw, stepT) : INFINITY, extend = CERT ? CTL_FLOAT(cw, extend) : 0.f;
    // running-sums build: the task's block, its sources' expiries, and how far the transform has come
    const int blk = INCR ? (t * G) / CTL_INT(cw, Gb) : 0;
    const float moved = INCR ? CTL_FLOAT(cw, moved) : 0.f;
    float* __restrict__ expiry = INCR ? expiry_of(q) : nullptr;
    float* tmin = reinterpret_cast<float*>(reinterpret_cast<double*>(base + lay.rows_off) + kQIncrRows);
    float emin = INFINITY;                                     // the earliest expiry among this lane's sources of the task
    if constexpr (INCR) {
      if (stepR < INFINITY && ((CTL_INT(cw, rerun) >> blk) & 1) == 0) { QUEUE_MARK(0); return; }      // every certificate of the block outlasts this iteration
    }
    const float2* __restrict__ gsrc = sc.src_sorted + (int64_t)p * sc.src_stride;
    const double2* __restrict__ gsrc64 = F64 ? sc.src64_sorted + (int64_t)p * sc.src_stride : nullptr;
    // running sums in float64-exact mode: no sorted float64 copies exist (icp_sort_kernel skips them) — the few float64 points that are needed
    // (near ties, changed matches, the first iteration's sums) are read from the caller's arrays through the sorted order's index maps
    const double2* __restrict__ src64o = F64 ? reinterpret_cast<const double2*>(args.src64) + args.src_off[p] : nullptr;
    const double2* __restrict__ tgt64o = F64 ? reinterpret_cast<const double2*>(args.tgt64) + args.tgt_off[p] : nullptr;
    const uint16_t* __restrict__ sidx = sc.src_index + (int64_t)p * sc.src_stride;
    double T[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int k = 0; k < 4; ++k) T[r][k] = CTL_T(cw, 4 * r + k);
    NearTie nt = near_tie_setup(false, 0.f, 0.f, 0.f);
    if constexpr (F64) {
      const float Rt = CTL_FLOAT(cw, Rt);
      const float Rs = CTL_FLOAT(cw, Rs);
      nt = near_tie_setup(true, Rt, Rs, sqrtf((float)(T[0][0] * T[0][0] + T[0][1] * T[0][1] + T[1][0] * T[1][0] + T[1][1] * T[1][1])) * 1.0001f);
    }
    // the bound a chunk must beat: the lane's best itself, or in float64-exact mode everything within its near-tie window;
    // divided by the safety factor once instead of multiplying every box distance by it (1.00002 > 1 / 0.99999)
    auto reach = [&](float b) { return (F64 ? __builtin_fmaf(b, nt.K1, nt.C) : b) * 1.00002f; };
    typedef const __attribute__((address_space(3))) unsigned char* LdsBytes;
    const unsigned tile0 = (unsigned)(size_t)(LdsBytes)(const void*)base;
    const int64_t ts = (int64_t)p * sc.tgt_stride;
    const uint16_t* __restrict__ tidx = sc.tgt_index + ts;
    const double2* __restrict__ gtgt64 = F64 ? sc.tgt64_sorted + ts : nullptr;
    auto target_at = [&](int j) { return *reinterpret_cast<const float2*>(base + rec_off(j)); };
    auto box_at = [&](int cidx) { return *reinterpret_cast<const float4*>(base + (size_t)cidx * kQRec + CHUNK * 8); };

    // running-sums build: after the first iteration (whose sums base_task takes over all pairs) a search adds what its changed matches
    // change to row t of the round; task_dirty: the row has been written
    const bool changes_wanted = INCR && stepR < INFINITY;
    bool task_dirty = false;
    double* drow = reinterpret_cast<double*>(base + lay.rows_off) + 2 * kQBase + (size_t)t * 2 * kQDelta;

    // ---- the nearest neighbours of up to 64 sources, one per lane (i: the lane's source, owns: it has one): exact search, then
    // the source's margin for the iterations to come
    auto search_points = [&](int i, bool owns) {
      QUEUE_COUNT(0, 1);
      float ax = 0.f, ay = 0.f;
      if (owns) {
        float2 v;
        if constexpr (F64 && !INCR) { const double2 pd = load_source64(src_lds, sh_src64, gsrc64, i); v = make_float2((float)pd.x, (float)pd.y); }
        else v = load_source(src_lds, sh_src, gsrc, i);
        ax = (float)(T[0][0] * (double)v.x + T[0][1] * (double)v.y + T[0][3]);      // icp.py:116
        ay = (float)(T[1][0] * (double)v.x + T[1][1] * (double)v.y + T[1][3]);
      }
      // the wave's box: the box of its transformed points
      const float wx0 = wave_min_dpp(owns ? ax : INFINITY), wy0 = wave_min_dpp(owns ? ay : INFINITY);
      const float wx1 = wave_max_dpp(owns ? ax : -INFINITY), wy1 = wave_max_dpp(owns ? ay : -INFINITY);
      // ---- seed: the chunk of last iteration's match (first iteration: of the target at the same relative rank), compared
      // target by target by the lane itself: an exact bound before any box test
      float bd = owns ? INFINITY : -INFINITY;                  // lanes without a point never ask for a chunk
      float sec = INFINITY;
      int pos = 0, own = -1, pv = 0xffff;
      bool tie = false;
      if (owns) {
        pv = (INCR && !(stepR < INFINITY)) ? 0xffff : (int)sh_prev[i];
        own = (pv != 0xffff ? pv : (int)(((long long)i * m) / n)) / CHUNK;
        queue_eval_chunk<F64 || CERT>(tile0, own, ax, ay, bd, sec, pos);
      }
      // Every chunk whose box lies within the seed's distance + E of the point is compared: the nearest target is among them (E = 0
      // would do for that), and what is skipped is farther than that distance + E — the second lower bound of the margin below.
      const float dseed = CERT ? (owns ? sqrtf(bd) * 1.000001f + extend : -INFINITY) : 0.f;
      const float rb = CERT ? reach(dseed * dseed * 1.000001f) : reach(bd);      // (−inf)² = +inf: lanes without a point are excluded by `owns` below
      const float wb = wave_max_dpp(owns ? rb : -INFINITY);    // no lane can ask for a chunk farther than this from the wave's box
      // ---- sweep: 64 chunk boxes tested at once, one per lane, against the wave's box; the survivors against every lane's own
      // point and current bound → a 64-bit list per lane, which the lane works through
      for (int c0 = 0; c0 < nchunks; c0 += 64) {
        bool cand = false;
        if (c0 + lane < nchunks) {
          const float4 b = box_at(c0 + lane);
          const float dx = fmaxf(fmaxf(b.x - wx1, wx0 - b.z), 0.f);
          const float dy = fmaxf(fmaxf(b.y - wy1, wy0 - b.w), 0.f);
          cand = wb >= __builtin_fmaf(dy, dy, dx * dx);
        }
        unsigned long long todo = __ballot(cand);
        if (!todo) continue;
        QUEUE_COUNT(1, __builtin_popcountll(todo));
        unsigned list_lo = 0u, list_hi = 0u;
        while (todo) {
          // four surviving boxes per trip, their (wave-uniform) reads issued together: one LDS round trip per box, each waited for
          // in turn, was the longest dependent chain of the search
          int tb[4];
          bool ok[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            ok[u] = todo != 0ull;
            tb[u] = ok[u] ? __builtin_ctzll(todo) : 0;
            todo &= todo - 1;                                  // 0 stays 0
          }
          float4 b[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) b[u] = box_at(c0 + tb[u]);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float ox = fmaxf(fmaxf(b[u].x - ax, ax - b[u].z), 0.f);
            const float oy = fmaxf(fmaxf(b[u].y - ay, ay - b[u].w), 0.f);
            const bool need = ok[u] && owns && rb >= __builtin_fmaf(oy, oy, ox * ox);
            // the bit's half is wave-uniform: one select and one or
            const unsigned bit = need ? (1u << (tb[u] & 31)) : 0u;
            if (tb[u] < 32) list_lo |= bit; else list_hi |= bit;
          }
        }
        // the seed chunk was compared already: a second visit would read as a tie of the point with itself
        {
          const int rel = own - c0;
          const unsigned clr = 1u << (rel & 31);
          if (rel >= 0 && rel < 32) list_lo &= ~clr;
          if (rel >= 32 && rel < 64) list_hi &= ~clr;
        }
        unsigned long long list = ((unsigned long long)list_hi << 32) | list_lo;
#ifdef LSLAM_QUEUE_PROFILE
        { unsigned long long items = (unsigned long long)__builtin_popcountll(list);
          for (int o = 32; o > 0; o >>= 1) items += __shfl_xor(items, o, 64);
          QUEUE_COUNT(3, items); }
#endif
        while (__any(list != 0ull)) {
          QUEUE_COUNT(2, 1);
          if (list != 0ull) {
            const int tbit = __builtin_ctzll(list);
            list &= list - 1;
            queue_eval_chunk<F64 || CERT>(tile0, c0 + tbit, ax, ay, bd, sec, pos);
          }
        }
      }
      bool near = false;
      float mg_tie = 0.f;                                      // float64-exact mode: the margin of a source whose near tie was decided on the float64 points
      if constexpr (F64) { near = owns && bd < kQFarD2 && sec <= near_lim(bd, nt); tie = near; }
      else tie = owns && (sec == bd || !(bd < kQFarD2));
      if constexpr (F64) {
        // a best distance that cannot be trusted (magnitudes near the padding's): every target is a candidate
        const bool far = owns && !(bd < kQFarD2);
        // ---- near ties: every target whose float32 distance is within lim of the winner's is compared in float64 on the float64
        // points (lowest original index on float64 ties); chunks walked with the flagged lanes' own points and windows
        if (__any(tie || far)) {
          if (lay.debug && lane == 0) atomicAdd(&lay.queue[2], 1u);
          // With certificates the window of a near tie is widened to the seed's reach: every target closer than that is then compared in
          // float64 (the winner cannot change: it lies inside the narrow window), and the runner-up's exact distance gives the source a
          // margin like any other — without one it would be searched, and walk all chunks here, in every iteration to come.
          const float lim = far ? INFINITY : (near ? (CERT ? fmaxf(near_lim(bd, nt), dseed * dseed * 1.000002f) : near_lim(bd, nt)) : -INFINITY);
          double adx = 0.0, ady = 0.0;
          if (near || far) {
            const double2 pd = INCR ? src64o[sidx[i]] : load_source64(src_lds, sh_src64, gsrc64, i);
            adx = T[0][0] * pd.x + T[0][1] * pd.y + T[0][3];
            ady = T[1][0] * pd.x + T[1][1] * pd.y + T[1][3];
          }
          double bv = INFINITY, bv2 = INFINITY;               // the smallest float64 distance² and the runner-up's (equal on an exact tie)
          int bo = 0x7fffffff, bp = pos;
          // 64 chunk boxes at a time, one per lane, against the box of the flagged lanes' points and their widest window; the survivors
          // one by one against every flagged lane's own point (a pass used to walk all chunks: 256 dependent LDS reads for one lane's tie)
          const bool flagged = near || far;
          const float fx0 = wave_min_dpp(flagged ? ax : INFINITY), fy0 = wave_min_dpp(flagged ? ay : INFINITY);
          const float fx1 = wave_max_dpp(flagged ? ax : -INFINITY), fy1 = wave_max_dpp(flagged ? ay : -INFINITY);
          const float flim = wave_max_dpp(lim) * 1.00002f;
          for (int c0 = 0; c0 < nchunks; c0 += 64) {
            bool cand = false;
            if (c0 + lane < nchunks) {
              const float4 b = box_at(c0 + lane);
              const float dx = fmaxf(fmaxf(b.x - fx1, fx0 - b.z), 0.f);
              const float dy = fmaxf(fmaxf(b.y - fy1, fy0 - b.w), 0.f);
              cand = flim >= __builtin_fmaf(dy, dy, dx * dx);
            }
            unsigned long long todo = __ballot(cand);
          while (todo) {
            const int cc = c0 + __builtin_ctzll(todo);
            todo &= todo - 1;
            const float4 b = box_at(cc);
            const float ox = fmaxf(fmaxf(b.x - ax, ax - b.z), 0.f);
            const float oy = fmaxf(fmaxf(b.y - ay, ay - b.w), 0.f);
            if (!__any(lim * 1.00002f >= __builtin_fmaf(oy, oy, ox * ox))) continue;
            for (int k = 0; k < CHUNK; ++k) {
              const int j = cc * CHUNK + k;
              if (j >= m) break;
              const float2 qv = target_at(j);
              if (dist2(ax, ay, qv.x, qv.y) <= lim) {          // rare: a candidate of this lane
                const int o = tidx[j];
                const double2 td = INCR ? tgt64o[o] : gtgt64[j];
                const double dx = adx - td.x, dy = ady - td.y;
                const double e = dx * dx + dy * dy;
                if (e < bv || (e == bv && o < bo)) { bv2 = bv; bv = e; bo = o; bp = j; }
                else if (e < bv2) bv2 = e;
              }
            }
          }
          }
          if (near || far) pos = bp;
          if (CERT && near && !far && bv < (double)kQFarD2) {
            // every target within √lim of the point was compared exactly; the others are at least dseed away as in the general case
            const double d1 = sqrt(bv);
            const float an = sqrtf(__builtin_fmaf(ay, ay, ax * ax));
            const float in64 = (float)(0.5 * (sqrt(bv2) - d1) - 1e-13 * ((double)an + d1 + 1.0)) * 0.999999f;
            const float out32 = 0.5f * (dseed * 0.99999f - (float)d1 * 1.0000003f) - (2.5e-7f * an + 1e-30f + nt.eps0 * 1.01f);
            mg_tie = fminf(in64, out32);
          }
        }
      } else if (__any(tie)) {
        // ---- equal minima (duplicate or equidistant points): among the targets at exactly the smallest distance the lowest
        // ORIGINAL index wins.  The search is exact, so bd is the smallest distance; a lane whose bd cannot be trusted takes the
        // minimum over all real targets first.
        if (lay.debug && lane == 0) atomicAdd(&lay.queue[2], 1u);
        if (__any(tie && !(bd < kQFarD2))) {
          float b2 = INFINITY;
          for (int j = 0; j < m; ++j) { const float2 qv = target_at(j); b2 = fminf(b2, dist2(ax, ay, qv.x, qv.y)); }
          if (tie && !(bd < kQFarD2)) bd = b2;
        }
        int oidx = 0x7fffffff;
        for (int j0 = 0; j0 < m; j0 += 8) {
          float2 qv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) qv[u] = target_at(j0 + u < m ? j0 + u : m - 1);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (tie && j0 + u < m && dist2(ax, ay, qv[u].x, qv[u].y) == bd) {      // rare
              const int o = tidx[j0 + u];
              if (o < oidx) { oidx = o; pos = j0 + u; }
            }
          }
        }
      }
      if constexpr (INCR) {
        const bool changed = changes_wanted && owns && pos != pv && pv != 0xffff;
#ifdef QX_NODELTA
        if (false) {
#else
        if (__any(changed)) {
#endif
          double x = 0.0, y = 0.0, qnx = 0.0, qny = 0.0, qox = 0.0, qoy = 0.0;
          if (changed) {
            if constexpr (F64) {
              const double2 sp = src64o[sidx[i]], tn_ = tgt64o[tidx[pos]], to_ = tgt64o[tidx[pv]];
              x = sp.x; y = sp.y; qnx = tn_.x; qny = tn_.y; qox = to_.x; qoy = to_.y;
            } else {
              const float2 v = gsrc[i], tn_ = target_at(pos), to_ = target_at(pv);
              x = (double)v.x; y = (double)v.y; qnx = (double)tn_.x; qny = (double)tn_.y; qox = (double)to_.x; qoy = (double)to_.y;
            }
          }
          auto leave = [&](int k, DD d) {                      // the wave's change of sum k, added to the task's row by lane 63
            d = wave_sum_dd_last(d);
            if (lane == 63) {
              if (task_dirty) dd_add(d, DD{drow[2 * k], drow[2 * k + 1]});
              drow[2 * k] = d.hi; drow[2 * k + 1] = d.lo;
            }
            __builtin_amdgcn_sched_barrier(0);                 // one sum at a time: seven in flight spill
          };
          { DD d; two_sum(qnx, -qox, d.hi, d.lo); leave(0, d); }
          { DD d; two_sum(qny, -qoy, d.hi, d.lo); leave(1, d); }
          leave(2, dd_prod_diff(x, qnx, qox)); leave(3, dd_prod_diff(x, qny, qoy));
          leave(4, dd_prod_diff(y, qnx, qox)); leave(5, dd_prod_diff(y, qny, qoy));
          { DD d = dd_sq_diff(qnx, qox), e = dd_sq_diff(qny, qoy); dd_add(d, e); leave(6, d); }
          task_dirty = true;
        }
      }
      if (owns) {
        sh_prev[i] = (uint16_t)pos;
        // The margin.  Real distances: nearest d1 ≤ √bd·(1+3u); every other target is at least L = min(√sec, √bd_seed + E)·(1−3u)
        // away (evaluated ones: sec; skipped chunks: their boxes were farther than the reach).  After the source has moved by δ
        // its nearest target is still the only one that the float32 recipe (relative error 4u) — and in float64-exact mode the
        // near-tie window, 2ε = 2(ε0 + 6e-8·d) wide — can rank first as long as  L − δ > (d1 + δ)(1 + 1e-5) + 2ε0:  margin =
        // (L − d1(1 + 2e-5))/2 − ε0 against δ(1 + 2e-5), less the rounding of the two float32 positions (2u|a|).  No margin after
        // a tie pass (the minimum is shared or was undecided).
        float mg = 0.f;
        if (CERT && !tie && bd < kQFarD2) {
          const float d1 = sqrtf(bd) * 1.00002f;
          const float L = fminf(sqrtf(sec), dseed) * 0.99999f;
          const float an = sqrtf(__builtin_fmaf(ay, ay, ax * ax));
          float slack = 2.5e-7f * an + 1e-30f;
          if constexpr (F64) slack += nt.eps0 * 1.01f;          // (the distance-proportional part of ε is inside the factors on d1 and on the step)
          mg = 0.5f * (L - d1) - slack;
        }
        if (F64 && CERT && tie) mg = mg_tie;
        if constexpr (INCR) {                                    // as an expiry, rounded down
          float ex = mg > 0.f ? __uint_as_float(__float_as_uint(moved + mg) - 1u) : 0.f;
          bool listable = lay.contested && !(mg > 0.f) && !tie && bd < kQFarD2;
          if constexpr (F64) listable = listable && sec > near_lim(bd, nt) * 1.0001f;
          if (listable) {
            unsigned* ccount; uint16_t* cidx; float2* cpos;
            contested_of(base, ccount, cidx, cpos);
            const unsigned k = atomicAdd(ccount, 1u);
            if (k < (unsigned)kQContested) { cidx[k] = (uint16_t)i; cpos[k] = make_float2(ax, ay); ex = INFINITY; }
          }
          expiry[i] = ex;
          emin = fminf(emin, ex);
        } else if constexpr (CERT) {                             // kept as a half, rounded toward zero (never up)
          const unsigned hb = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(mg > 0.f ? mg : 0.f, 0.f));
          sh_margin[i] = (uint16_t)(hb & 0xffffu);
        }
      }
    };

    // ---- the task's slabs: sources whose margin outlasts this iteration's step keep their match; the others are gathered and
    // searched 64 at a time (the first iteration, and any iteration without margins, searches slab by slab)
    const int slab_lo = t * G, slab_hi = min(slab_lo + G, nslabs);
    if constexpr (!CERT) {
      for (int slab = slab_lo; slab < slab_hi; ++slab) search_points(slab * kWave + lane, slab * kWave + lane < n);
      QUEUE_MARK(0);
      return;
    }
    int pending = 0;                                           // entries of `gathered`
    unsigned needbits = 0u;                                    // bit k: this lane's source of slab (group start + k) must be searched
    const float step = CERT ? (__builtin_fmaf(stepR, CTL_FLOAT(cw, Rs), stepT)) * 1.00002f : 0.f;      // every source of the cloud moved by less
    for (int slab = slab_lo; slab <= slab_hi; ++slab) {       // slab_hi: the trip that searches what is left in the list
      const int i = slab * kWave + lane;
      const bool owns = slab < slab_hi && i < n;
      const int u = (slab - slab_lo) & 7;
      if (u == 0 && slab < slab_hi) {
        // the margins of eight slabs at a time: their loads in flight together, the updated ones written back
        needbits = 0u;
        if constexpr (INCR) {
          float ev[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int ii = (slab + k) * kWave + lane;
            ev[k] = (stepR < INFINITY && slab + k < slab_hi && ii < n) ? expiry[ii] : 0.f;
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int ii = (slab + k) * kWave + lane;
            const bool ok = slab + k < slab_hi && ii < n;
            const bool keep = ok && stepR < INFINITY && ev[k] > moved;
            emin = keep ? fminf(emin, ev[k]) : emin;
            needbits |= (ok && !keep) ? (1u << k) : 0u;
          }
        } else {
        uint16_t mv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int ii = (slab + k) * kWave + lane;
          mv[k] = (stepR < INFINITY && slab + k < slab_hi && ii < n) ? sh_margin[ii] : (uint16_t)0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int ii = (slab + k) * kWave + lane;
          const bool ok = slab + k < slab_hi && ii < n;
          const float left = (float)__builtin_bit_cast(_Float16, mv[k]) - step;
          const bool keep = ok && stepR < INFINITY && left > 0.f;
          if (keep) {
            const unsigned hb = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(left, 0.f));
            sh_margin[ii] = (uint16_t)(hb & 0xffffu);
          }
          needbits |= (ok && !keep) ? (1u << k) : 0u;
        }
        }
      }
      const bool need = owns && ((needbits >> u) & 1u) != 0u;
      const unsigned long long mask = __ballot(need);
      const int nneed = __builtin_popcountll(mask);
      QUEUE_COUNT(4, __builtin_popcountll(__ballot(owns)));
      QUEUE_COUNT(5, nneed);
      if (pending > 0 && (slab == slab_hi || pending + nneed > kWave)) {      // the list is full, or this was the last slab
        const int gi = lane < pending ? gathered[lane] : 0;
#ifdef LSLAM_QUEUE_PROFILE
        const unsigned long long sp0_ = __builtin_amdgcn_s_memtime();
#endif
        search_points(gi, lane < pending);
#ifdef LSLAM_QUEUE_PROFILE
        cnt[6] += __builtin_amdgcn_s_memtime() - sp0_;
#endif
        pending = 0;
      }
      if (need) gathered[pending + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = (uint16_t)i;
      pending += nneed;
    }
    if constexpr (INCR) {
      if (task_dirty && lane == 0) atomicOr(&c->dirty, 1u << t);
      emin = wave_min_dpp(emin);                               // (expiries are ≥ 0: their bit patterns order as they do)
      if (lane == 0) atomicMin(reinterpret_cast<unsigned*>(&tmin[blk]), __float_as_uint(emin));
    }
    QUEUE_MARK(0);
  };

  // ---- one wave (running-sums build, first iteration): row r of the sums over all pairs, as double-doubles ----------------------
  auto base_task = [&](int q, int r) {
#ifdef QX_NOBASE
    if constexpr (false) {
#else
    if constexpr (INCR) {
#endif
      QueueCtl* c = ctl_of(q);
      unsigned char* base = slot_of(q);
      const uint16_t* sh_prev = prevs_of(q, base);
      double* drow = reinterpret_cast<double*>(base + lay.rows_off) + 2 * kQBase + (size_t)r * 2 * kQBase;
      const int cw = ctl_load(c, lane);
      const int n = CTL_INT(cw, n), nslabs = CTL_INT(cw, nslabs), p = CTL_INT(cw, problem);
      const float2* __restrict__ gsrc = sc.src_sorted + (int64_t)p * sc.src_stride;
      const double2* __restrict__ src64o = F64 ? reinterpret_cast<const double2*>(args.src64) + args.src_off[p] : nullptr;
      const double2* __restrict__ tgt64o = F64 ? reinterpret_cast<const double2*>(args.tgt64) + args.tgt_off[p] : nullptr;
      const uint16_t* __restrict__ tidx = sc.tgt_index + (int64_t)p * sc.tgt_stride;
      const int slab_lo = r * base_slabs(nslabs), slab_hi = min(slab_lo + base_slabs(nslabs), nslabs);
      // two passes (twelve double-double accumulators at once do not fit the 128 registers of a 16-wave workgroup), four slabs at a time:
      // the index maps, then the points they lead to, are in flight together — one slab at a time, a task waited for 80 dependent loads
      constexpr int BS = 2;
      auto leave = [&](int k0, DD v) {                       // sum k0 of the row, from lane 63
        v = wave_sum_dd_last(v);
        if (lane == 63) { drow[2 * k0] = v.hi; drow[2 * k0 + 1] = v.lo; }
        __builtin_amdgcn_sched_barrier(0);
      };
      // float64-exact mode: the slabs are slabs of the CALLER's order — the float64 sources are then read as they lie in memory (and in a
      // scan-matching batch, where source o matches target o or a neighbour, so are the targets); through the sorted order's index map every
      // lane touched a line of its own, three times the bytes.  A sum does not care about the order; it is a fixed one either way.
      const uint16_t* __restrict__ srank = sc.src_rank + (int64_t)p * sc.src_stride;
      auto sources_of = [&](int s0, bool (&ok)[BS], double (&x)[BS], double (&y)[BS]) {
        if constexpr (F64) {
#pragma unroll
          for (int u = 0; u < BS; ++u) {
            const int o = (s0 + u) * kWave + lane; ok[u] = s0 + u < slab_hi && o < n;
            const double2 sp = ok[u] ? src64o[o] : make_double2(0.0, 0.0); x[u] = sp.x; y[u] = sp.y;
          }
        } else {
#pragma unroll
          for (int u = 0; u < BS; ++u) {
            const int i = (s0 + u) * kWave + lane; ok[u] = s0 + u < slab_hi && i < n;
            const float2 v = ok[u] ? gsrc[i] : make_float2(0.f, 0.f); x[u] = (double)v.x; y[u] = (double)v.y;
          }
        }
      };
      {
        DD a0{0.0, 0.0}, a1{0.0, 0.0}, a2{0.0, 0.0}, a3{0.0, 0.0}, a4{0.0, 0.0};
        for (int s0 = slab_lo; s0 < slab_hi; s0 += BS) {
          bool ok[BS]; double x[BS], y[BS];
          sources_of(s0, ok, x, y);
#pragma unroll
          for (int u = 0; u < BS; ++u)
            if (ok[u]) { dd_acc(a0, x[u]); dd_acc(a1, y[u]); dd_acc_prod(a2, x[u], x[u]); dd_acc_prod(a3, x[u], y[u]); dd_acc_prod(a4, y[u], y[u]); }
        }
        leave(0, a0); leave(1, a1); leave(2, a2); leave(3, a3); leave(4, a4);
      }
      {
        DD a5{0.0, 0.0}, a6{0.0, 0.0}, a7{0.0, 0.0}, a8{0.0, 0.0}, a9{0.0, 0.0}, a10{0.0, 0.0}, a11{0.0, 0.0};
        for (int s0 = slab_lo; s0 < slab_hi; s0 += BS) {
          bool ok[BS]; double x[BS], y[BS], qx[BS], qy[BS];
          int ps[BS];
#pragma unroll
          for (int u = 0; u < BS; ++u) {
            const int i = (s0 + u) * kWave + lane;
            if constexpr (F64) ps[u] = (s0 + u < slab_hi && i < n) ? (int)srank[i] : 0;      // source o = i of the caller's order sits at srank[o] …
            else ps[u] = (s0 + u < slab_hi && i < n) ? (int)sh_prev[i] : 0;
          }
          if constexpr (F64) {
#pragma unroll
            for (int u = 0; u < BS; ++u) { const int i = (s0 + u) * kWave + lane; ps[u] = (s0 + u < slab_hi && i < n) ? (int)sh_prev[ps[u]] : 0; }      // … and is matched to sorted target ps
          }
          sources_of(s0, ok, x, y);
          if constexpr (F64) {
            int ti[BS];
#pragma unroll
            for (int u = 0; u < BS; ++u) ti[u] = ok[u] ? (int)tidx[ps[u]] : 0;
#pragma unroll
            for (int u = 0; u < BS; ++u) { const double2 tq = ok[u] ? tgt64o[ti[u]] : make_double2(0.0, 0.0); qx[u] = tq.x; qy[u] = tq.y; }
          } else {
#pragma unroll
            for (int u = 0; u < BS; ++u) { const float2 qv = *reinterpret_cast<const float2*>(base + rec_off(ps[u])); qx[u] = (double)qv.x; qy[u] = (double)qv.y; }
          }
#pragma unroll
          for (int u = 0; u < BS; ++u)
            if (ok[u]) {
              dd_acc(a5, qx[u]); dd_acc(a6, qy[u]);
              dd_acc_prod(a7, x[u], qx[u]); dd_acc_prod(a8, x[u], qy[u]); dd_acc_prod(a9, y[u], qx[u]); dd_acc_prod(a10, y[u], qy[u]);
              dd_acc_prod(a11, qx[u], qx[u]); dd_acc_prod(a11, qy[u], qy[u]);
            }
        }
        leave(5, a5); leave(6, a6); leave(7, a7); leave(8, a8); leave(9, a9); leave(10, a10); leave(11, a11);
      }
      QUEUE_MARK(1);
    }
  };

  // ---- one wave: row r of the slot's float64 partial sums — the matched pairs of slabs [r·moment_slabs(), …) --------------------
  auto moment_task = [&](int q, int r) {
    if constexpr (INCR) return;
    QueueCtl* c = ctl_of(q);
    unsigned char* base = slot_of(q);
    const int2* tdelta = reinterpret_cast<const int2*>(base + lay.tdelta_off);
    const float2* sh_src = reinterpret_cast<const float2*>(base + lay.src_off);
    const double2* sh_src64 = reinterpret_cast<const double2*>(base + lay.src_off);
    const uint16_t* sh_prev = prevs_of(q, base);
    double* rows = reinterpret_cast<double*>(base + lay.rows_off);
    const int cw = ctl_load(c, lane);
    const int n = CTL_INT(cw, n), nslabs = CTL_INT(cw, nslabs);
    double T[2][4];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int k = 0; k < 4; ++k) T[rr][k] = CTL_T(cw, 4 * rr + k);
    double a[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) a[k] = 0.0;
    double lo[2] = {INFINITY, INFINITY}, hi[2] = {-INFINITY, -INFINITY};
    const int slab_lo = r * moment_slabs(nslabs), slab_hi = min(slab_lo + moment_slabs(nslabs), nslabs);
    for (int slab = slab_lo; slab < slab_hi; ++slab) {
      const int i = slab * kWave + lane;
      if (i < n) {
        const int ps = sh_prev[i];
        double x, y, qx, qy;
        const float2 qv = *reinterpret_cast<const float2*>(base + rec_off(ps));
        if constexpr (F64) {
          const double2 sp = load_source64(lay.src_off != 0xffffffffu, sh_src64, sc.src64_sorted + (int64_t)CTL_INT(cw, problem) * sc.src_stride, i);
          const int2 dv = tdelta[ps];
          x = sp.x; y = sp.y;
          qx = rebuild64(qv.x, dv.x); qy = rebuild64(qv.y, dv.y);
        } else {
          const float2 v = load_source(lay.src_off != 0xffffffffu, sh_src, sc.src_sorted + (int64_t)CTL_INT(cw, problem) * sc.src_stride, i);
          x = (double)v.x; y = (double)v.y;
          qx = (double)qv.x; qy = (double)qv.y;
        }
        const double px = T[0][0] * x + T[0][1] * y + T[0][3];
        const double py = T[1][0] * x + T[1][1] * y + T[1][3];
        a[0] += px; a[1] += py; a[2] += qx; a[3] += qy;
        a[4] += px * qx; a[5] += px * qy; a[6] += py * qx; a[7] += py * qy;
        const double dx = px - qx, dy = py - qy;
        double e = 0.0;
        e += dx * dx; e += dy * dy;
        a[8] += e;
        if (box) {
          lo[0] = fmin(lo[0], fmin(px, qx)); lo[1] = fmin(lo[1], fmin(py, qy));
          hi[0] = fmax(hi[0], fmax(px, qx)); hi[1] = fmax(hi[1], fmax(py, qy));
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) a[k] = wave_sum_dpp_last(a[k]);
    if (box) {
#pragma unroll
      for (int k = 0; k < 2; ++k) { lo[k] = wave_min_dpp_last(lo[k]); hi[k] = wave_max_dpp_last(hi[k]); }
    }
    if (lane == 63) {
      if (QEXP_PAIR == 1 || nslabs <= QEXP_MOM) {
        double* row = rows + (size_t)r * RS;
#pragma unroll
        for (int k = 0; k < 9; ++k) row[k] = a[k];
        if (box) { row[9] = lo[0]; row[10] = lo[1]; row[11] = hi[0]; row[12] = hi[1]; }
      } else {
        typedef __attribute__((address_space(3))) double* LdsDouble;
        LdsDouble row = (LdsDouble)(rows + (size_t)(r / QEXP_PAIR) * RS);
#pragma unroll
        for (int k = 0; k < 9; ++k) (void)__hip_atomic_fetch_add(row + k, a[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (box) {
          (void)__hip_atomic_fetch_min(row + 9, lo[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          (void)__hip_atomic_fetch_min(row + 10, lo[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          (void)__hip_atomic_fetch_max(row + 11, hi[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          (void)__hip_atomic_fetch_max(row + 12, hi[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
    QUEUE_MARK(1);
  };

  // ---- one wave: task t of the slot's current round; the wave that completes the round's last task moves the problem on ---------
  auto run_task = [&](int q, int t, int ntasks) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    QueueCtl* c = ctl_of(q);
    int phase = lds_load_i32(&c->phase);
    // phase 2: FUSED tasks — the search of a moment group's slabs and then that group's row, in one task: once most sources keep
    // their match the search round is a handful of cheap tasks, and a second round only adds its hand-offs to every iteration
    const bool fused = phase == 2;
    bool solo = false;                                         // the one moment task of a small problem, run by the wave that closed its search round
    for (;;) {                                                 // (a loop so that each routine has one call site)
      if (phase != 1) search_task(q, t);
      if (phase != 0) { if constexpr (INCR) base_task(q, t); else moment_task(q, t); }
      if (!solo) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        unsigned d = 0;
        if (lane == 0) d = atomicAdd(&c->done_tasks, 1u) + 1u;
        d = __builtin_amdgcn_readfirstlane(d);
        if (d != (unsigned)ntasks) return;
        if (lane == 0) close_slot(q);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      if (phase != 0) break;
      if constexpr (INCR) {                                    // after the first iteration the searches have left their changes: no second round
        if (__int_as_float(lds_load_i32(reinterpret_cast<const int32_t*>(&c->stepR))) < INFINITY) break;
      }
      const int nrows = INCR ? base_tasks(lds_load_i32(&c->nslabs)) : moment_tasks(lds_load_i32(&c->nslabs));      // tasks of the moment round
      if (nrows > 1) {                                         // a large problem: its moments are a round of tasks as well
        if (lane == 0) { c->phase = 1; c->done_tasks = 0; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) open_slot(q, nrows);
        return;
      }
      phase = 1; t = 0; solo = true;
    }
    (void)fused;
    finish(q);
  };

#ifdef LSLAM_QUEUE_PROFILE
  const long long wg_t0 = lay.debug ? (long long)wall_clock64() : 0ll;
#endif
  // ---- first problems, then the queue ----
  for (int q = wave; q < Q; q += nwaves) refill(q);
  QUEUE_MARK(3);
  int pref = wave % Q;
  long long idle_since = -1;
  int last_beat = 0;
  for (;;) {
    // every slot's ticket with one load (lane q: slot q); the open slots as a ballot; the first one from `pref` on
    const unsigned w = __hip_atomic_load(&tickets[lane & 15], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned long long open = __ballot(lane < Q && (w & 0xffffu) < (w >> 16));
    if (open) {
      const unsigned long long from_pref = open >> pref;
      const int q = from_pref ? pref + __builtin_ctzll(from_pref) : __builtin_ctzll(open);
      unsigned old = 0;
      if (lane == 0) old = atomicAdd(&tickets[q], 1u);
      old = __builtin_amdgcn_readfirstlane(old);
      const unsigned t = old & 0xffffu, ntasks = old >> 16;
      if (t < ntasks) {
        QUEUE_MARK(4);
        QUEUE_COUNT(7, 1);
        run_task(q, (int)t, (int)ntasks);
        pref = q;
        idle_since = -1;
      }
      continue;
    }
    const int st_ = __hip_atomic_load(&states[lane & 15], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (!__any(lane < Q && st_ != 0)) break;
    if (lds_load_i32(sh_fail)) break;
    const long long now = (long long)wall_clock64();
    const int beat = lds_load_i32(sh_beat);
    if (idle_since < 0 || beat != last_beat) { idle_since = now; last_beat = beat; }
    else if (now - idle_since > lay.timeout_ticks) {             // can't happen on a healthy device: leave so that the grid drains
      if (lane == 0) { __hip_atomic_store(sh_fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); atomicAdd(&lay.queue[1], 1u); atomicAdd(&lay.queue[50], 1u); }
      break;
    }
    __builtin_amdgcn_s_sleep(QEXP_SLEEP);
#ifdef LSLAM_QUEUE_PROFILE
    if (__any(lane < Q && st_ == 0)) QUEUE_MARK(5); else QUEUE_MARK(4);       // [5]: idle while the workgroup's slots are draining (the queue is exhausted)
#endif
  }
  QUEUE_MARK(4);
#ifdef LSLAM_QUEUE_PROFILE
  if (lay.debug && tid == 0) {                                 // how evenly the workgroups finish (100 MHz ticks)
    const unsigned life = (unsigned)((long long)wall_clock64() - wg_t0);
    atomicMax(&lay.queue[36], life); atomicAdd(&lay.queue[38], life >> 8);
  }
  // cycles per phase and search statistics summed over the waves of the launch (printed by the launcher under LSLAM_DEBUG)
  if (lay.debug && lane == 0) {
    for (int k = 0; k < 6; ++k) atomicAdd(reinterpret_cast<unsigned long long*>(lay.queue + 4) + k, prof[k]);
    for (int k = 0; k < 8; ++k) atomicAdd(reinterpret_cast<unsigned long long*>(lay.queue + 4) + 6 + k, cnt[k]);
  }
#endif
}

// ---- launch ------------------------------------------------------------------------------------------
static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

hipError_t launch_icp_queue(IcpBatchArgs a, const PruneScratch& sc, int S, int64_t max_src_n, int64_t max_tgt_n, hipStream_t stream, int cls, bool counters_zeroed) {
  if ((a.src64 == nullptr) != (a.tgt64 == nullptr)) return hipErrorInvalidValue;
  const bool f64 = a.src64 != nullptr;
  const bool box = (a.flags & kFlagNormalize) != 0;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    if (cus <= 0) cus = 256;
  }
  constexpr size_t kLdsPerCu = 160 * 1024;
  const int RS = box ? 13 : 9;
  const int nchunks = (int)((max_tgt_n + kPruneChunk - 1) / kPruneChunk);
  const int padded = nchunks * kPruneChunk;
  const int nslabs = (int)((max_src_n + 63) / 64);
  auto up = [](size_t b) { return (b + 15) & ~(size_t)15; };
  QueueLayout lay{};
  // Certificates pay once a run has iterations in which little moves (C4, 50 iterations: 4 % of the source visits still search, 16.6
  // -> 6.9 ms, and 2.15 ms with the running sums they make possible); a LiDAR-sized pair stops after ~6 iterations of large steps, the sources it
  // could skip in the last of them are scattered over all slabs (C3: 0.72 -> 0.85 ms, 1.2 ms with running sums): from 2048 points per cloud by default.
  bool certify = max_src_n >= kQueueCertifyPoints || max_tgt_n >= kQueueCertifyPoints;
  if (const char* e = getenv("LSLAM_ICP_QUEUE_CERTIFY")) certify = atoi(e) != 0;
  if (cls >= 0 && !sc.order) return hipErrorInvalidValue;
  // running sums instead of a pass over all pairs per iteration (see finish): with certificates, unless the normalised error's box is asked for
  const bool incr = certify && !box && env_int("LSLAM_ICP_QUEUE_INCR", 1) != 0;
  const int rows_cap = queue_rows(f64);
  lay.rows_cap = rows_cap;
  // a slot's layout; float32 sources live in LDS when two slots still fit a CU, else they are read from the sorted copy in memory
  // (once per search and once per moment task, coalesced: 4096-point problems then keep two slots per CU)
  auto layout = [&](bool src_in_lds) -> size_t {
    size_t off = (size_t)nchunks * kQRec;
    lay.tdelta_off = (uint32_t)off; if (f64 && !incr) off += (size_t)padded * sizeof(float2);
    lay.src_off = 0xffffffffu;
    if (src_in_lds && !incr) { lay.src_off = (uint32_t)off; off += up((size_t)max_src_n * (f64 ? sizeof(double2) : sizeof(float2))); }
    lay.prev_off = (uint32_t)off; if (!incr) off += up((size_t)max_src_n * sizeof(uint16_t));      // running sums: in sc.margins
    lay.margin_off = (uint32_t)off; if (certify && !f64 && !incr) off += up((size_t)max_src_n * sizeof(uint16_t));      // float64-exact mode, running sums: sc.margins
    lay.rows_off = (uint32_t)off; off += sizeof(double) * (size_t)(incr ? kQIncrDoubles : (rows_cap * RS > 16 ? rows_cap * RS : 16));
    return up(off);
  };
  auto slots_for = [&](size_t slot_bytes) -> int {
    int q = 0;
    while (q < 16 && (size_t)(q + 1) * kQCtlBytes + 192 + 2048 + 16 + (size_t)(q + 1) * slot_bytes <= kLdsPerCu) ++q;
    return q;
  };
  size_t slot = layout(true);
  int Q = slots_for(slot);
  const int lean_below = env_int("LSLAM_ICP_QUEUE_LEAN_BELOW", f64 ? 6 : 2);      // sources out of LDS when fewer slots than this fit with them
  if (Q < lean_below && env_int("LSLAM_ICP_QUEUE_SRC_LDS", 0) == 0) {
    const size_t lean = layout(false);
    if (slots_for(lean) > Q) { slot = lean; Q = slots_for(lean); } else slot = layout(true);
  }
  if (Q < 1) return hipErrorInvalidValue;
  // geometry: one workgroup of W waves per CU holding Q slots
  int W = env_int("LSLAM_ICP_QUEUE_WAVES", 16);
  if (W < 1 || W > kIcpBatchMaxWaves) W = 16;
  {
    // fewer problems than the chip has slots: spread them over the CUs
    const int64_t slots_needed = ((int64_t)S + cus - 1) / cus;
    if (Q > slots_needed) Q = (int)(slots_needed > 0 ? slots_needed : 1);
    const int q_forced = env_int("LSLAM_ICP_QUEUE_Q", 0);
    if (q_forced > 0 && q_forced < Q) Q = q_forced;
  }
  if (Q * (nslabs > 0 ? nslabs : 1) < W) W = Q * (nslabs > 0 ? nslabs : 1);      // no more waves than tasks can ever be open
  // (builds without running sums) one slot per CU (a 4096-point problem in float64-exact mode): its rounds of 8–16 tasks do not feed 16 waves, and the idle ones poll
  // the LDS the working ones read (C4 float64-exact: 11.9 against 12.6 ms)
  if (Q == 1 && certify && W > 8 && !getenv("LSLAM_ICP_QUEUE_WAVES")) W = 8;
  // (builds without running sums) fused tasks from the second iteration on: a second round of tasks per iteration only adds hand-offs (C4, three slots of eight tasks:
  // 5.60 -> 5.23 ms; float64-exact, two slots: 9.5 -> 9.0 ms)
  lay.fuse = env_int("LSLAM_ICP_QUEUE_FUSE", 1);
  lay.slot_bytes = (uint32_t)slot;
  lay.ctl_bytes = (uint32_t)up((size_t)Q * kQCtlBytes + 64 + 128 + (size_t)kIcpBatchMaxWaves * kWave * sizeof(uint16_t));
  lay.Q = Q;
  lay.G = env_int("LSLAM_ICP_QUEUE_G", 0);                     // slabs per search task; 0: one, two for problems above 32 slabs
  if (nslabs > queue_rows(f64) * 2 * queue_row_slabs(f64) || nslabs > 16 * kQBaseRows) return hipErrorInvalidValue;
  lay.certify = certify ? 1 : 0;
  lay.ecap_frac = 0.25f;
  if (const char* e = getenv("LSLAM_ICP_QUEUE_ECAP")) lay.ecap_frac = (float)atof(e);
  lay.contested = env_int("LSLAM_ICP_QUEUE_CONTESTED", 1);
  lay.efloor_frac = 0.25f;
  if (const char* e = getenv("LSLAM_ICP_QUEUE_EFLOOR")) lay.efloor_frac = (float)atof(e);
  lay.e0_frac = 1.f;                                         // C4: 2.81 -> 2.58 ms (float64-exact 3.25 -> 3.08); flat from 1 to 3
  if (const char* e = getenv("LSLAM_ICP_QUEUE_E0")) lay.e0_frac = (float)atof(e);
  lay.queue = sc.queue;
  lay.timeout_ticks = kQTimeoutTicks;
  if (const char* e = getenv("LSLAM_ICP_QUEUE_TIMEOUT_US")) lay.timeout_ticks = (long long)(atof(e) * 100.0);      // (tests of the give-up path)
  lay.cls = cls; lay.order = sc.order;
  lay.draw = cls == 0 ? 44 : 0;                               // the two launches of a split batch draw from counters of their own (zeroed together, before the sort)
  lay.margins = sc.margins;
  lay.margin_stride = (uint32_t)((max_src_n + 63) & ~(int64_t)63);
  if ((f64 || incr) && certify && (!sc.margins || (size_t)(S < cus ? S : cus) * 16 * lay.margin_stride * (incr ? 3 : 1) > sc.margin_cap)) return hipErrorInvalidValue;
  lay.debug = getenv("LSLAM_DEBUG") ? 1 : 0;
  const size_t smem = lay.ctl_bytes + (size_t)Q * slot;
  if (smem > kLdsPerCu) return hipErrorInvalidValue;
  int grid = (S + Q - 1) / Q;
  if (grid > cus) grid = cus;
  const int threads = W * 64;
  if (lay.debug) {
    unsigned long long h[18];
    unsigned hu_tail[2] = {0, 0};
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(h, sc.queue, 144, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hu_tail, sc.queue + 32, 8, hipMemcpyDeviceToHost);
#ifdef LSLAM_QUEUE_PROFILE
    { unsigned life[3] = {0, 0, 0}; (void)hipMemcpy(life, sc.queue + 36, 12, hipMemcpyDeviceToHost);
      fprintf(stderr, "[lslam] queue kernel, previous launch: workgroup lifetimes max %.1f us, mean %.1f us\n", life[0] / 100.0, life[2] * 256.0 / 100.0 / (double)cus); }
#endif
    unsigned hu[34];
    for (int k = 0; k < 32; ++k) hu[k] = reinterpret_cast<const unsigned*>(h)[k];
    hu[32] = hu_tail[0]; hu[33] = hu_tail[1];
    { unsigned hist[4] = {0, 0, 0, 0}; (void)hipMemcpy(hist, sc.queue + 40, 16, hipMemcpyDeviceToHost);
      if (hist[0] + hist[1] + hist[2] + hist[3]) fprintf(stderr, "[lslam] queue kernel, previous launch: problems by rounds after the first: <8: %u, <16: %u, <32: %u, more: %u\n", hist[0], hist[1], hist[2], hist[3]); }
    fprintf(stderr, "[lslam] queue kernel, previous launch: problems drawn %u, timeouts %u, tie passes %u; running sums: %u rounds of %.2f blocks after the first, %u iterations without a round\n",
            hu[0], hu[1], hu[2], hu[3], hu[3] ? (double)hu[32] / hu[3] : 0.0, hu[33]);
#ifdef LSLAM_QUEUE_PROFILE
    double tot = 0; for (int k = 0; k < 6; ++k) tot += (double)h[2 + k];
    if (tot > 0) fprintf(stderr, "[lslam] queue kernel, share of wave cycles: search %.3f  moments %.3f  reduce+solve %.3f  outputs+staging %.3f  idle %.3f  idle while draining %.3f\n",
                         h[2] / tot, h[3] / tot, h[4] / tot, h[5] / tot, h[6] / tot, h[7] / tot);
    if (h[8] > 0) fprintf(stderr, "[lslam] queue kernel, gathered searches: %.3f of the search cycles (%.0f ticks each); %.0f ticks of search+moment+solve per task, %llu tasks\n",
                          (double)h[14] / (double)(h[2] ? h[2] : 1), (double)h[14] / h[8], (double)(h[2] + h[3] + h[4]) / (double)(h[15] ? h[15] : 1), h[15]);
    if (h[8] > 0) fprintf(stderr, "[lslam] queue kernel, %llu searches, %llu sources visited, %llu searched; wave cycles: search %.3g moments %.3g solve %.3g staging %.3g idle %.3g draining %.3g\n",
                          h[8], h[12], h[13], (double)h[2], (double)h[3], (double)h[4], (double)h[5], (double)h[6], (double)h[7]);
    if (h[8] > 0) fprintf(stderr, "[lslam] queue kernel, per search of <= 64 sources: %.2f surviving boxes, %.2f list rounds, %.2f (point, chunk) entries; %.3f of the sources visited were searched (%.1f per search)\n",
                          (double)h[9] / h[8], (double)h[10] / h[8], (double)h[11] / h[8], (double)h[13] / (double)(h[12] ? h[12] : 1), (double)h[13] / h[8]);
#endif
    fprintf(stderr, "[lslam] icp_queue_kernel: %d workgroups x %d waves, %d slots of %zu B (%zu B LDS), %d rows per slot, %s\n",
            grid, W, Q, slot, smem, rows_cap, f64 ? (incr ? "float64-exact, running sums" : "float64-exact") : (incr ? "float32, running sums" : "float32"));
  }
  hipError_t e = counters_zeroed ? hipSuccess : hipMemsetAsync(sc.queue, 0, 180, stream);      // (draw counter, debug and profile counters: words 0 … 43; the class counts [48] [49] and the sticky give-up count [50] stay)
  if (e != hipSuccess) return e;
  auto launch = [&](auto kernel) -> hipError_t {
    hipError_t ee = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (ee != hipSuccess) return ee;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), smem, stream, a, sc, lay, S);
    return hipGetLastError();
  };
  if (incr) return f64 ? launch(icp_queue_kernel<true, true, true>) : launch(icp_queue_kernel<false, true, true>);
  if (certify) return f64 ? launch(icp_queue_kernel<true, true, false>) : launch(icp_queue_kernel<false, true, false>);
  return f64 ? launch(icp_queue_kernel<true, false, false>) : launch(icp_queue_kernel<false, false, false>);
}

}  // namespace lslam
