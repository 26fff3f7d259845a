// fh_capi.hip — host side of the C ABI (include/fasterhip.h): context, buffers, kernel dispatch.
// Built with hipcc for gfx950 only into faster_amd/libfasterhip.so.  No CPU fallback exists: every entry
// point needs a HIP device and reports FH_ERR_DEVICE otherwise.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <chrono>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/fasterhip.h"
#include "fh_sample.hip.hpp"
#include "fh_solve.hip.hpp"
#include "fh_decomp.hip.hpp"
#include "fh_safe.hip.hpp"  // (after fh_solve: it switches FP contraction off for what follows, like fh_decomp)

struct fh_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::vector<hipEvent_t> ev;  // pairs (start, stop), one pair per solve-kernel launch
  size_t ev_used = 0;          // events in use since the last fh_timing_reset
  fh_params par;
  fh_sched sched;
  std::string err;
  // staging buffers of the host-pointer entry points (grown on demand, reused)
  // slot 5: snapshot workspace, 6: work-sharing control block + ring sequence numbers, 7: decomposition workspace,
  // 8: task slots of the ring, 9: share records
  // 10: corridor segments, 11: per-segment polytope rows, 12: per-segment row counts (fh_corridor_batch_device)
  // 13: launch order of a batch (order_kernel), 14: bounding boxes of the cloud's blocks (decomposition)
  // 15: reduced-space basis tables (fh_basis.hip.hpp), uploaded once by fh_create
  void* d_buf[20] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                     nullptr, nullptr, nullptr, nullptr};
  size_t d_cap[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int n_cu = 0;
  size_t lds_attr[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // largest dynamic-LDS size already set per kernel instantiation
  unsigned int* h_abort = nullptr;          // mapped host word polled by the kernels (fh_request_stop)
  unsigned int* d_abort = nullptr;          // its device address
  unsigned int* h_report = nullptr;         // pinned: the control block's report words of the last launch (copied with the results)
  double pair_margin = -1.0;                // fh_set_pair_margin
  fh_pair_rule pair_rule = {0, 0, 0.0, 0.0, 1.0, 0.5};  // fh_set_pair_rule
  fh::UnknownGrid unknown = {nullptr, 0.0, 0.0, 0.0, 1.0, 0, 0, 0, 0};  // fh_set_unknown_grid_device (rule mode 2); the flags belong to the caller
  bool ctl_ready = false;                   // the device-side control block is in its initial state (left so by the previous launch)
  bool launched = false;                    // a solve launch has been issued since the control block was last checked
  int last_grid = 0;
  fh_launch_info last_launch = {0, 0, 0, 0, 0, 0, 0, 0};  // fh_last_launch
  bool order_ready = false;                 // the launch-order counters are zero (left so by the previous scatter kernel)
  std::atomic<hipEvent_t> last_end{nullptr};  // the event behind the most recent solve launch (read by OTHER contexts: is a launch of this one in flight?)
};

// Every live context of the process (fh_create / fh_destroy): a solve launch asks the others whether they have a launch in flight on its
// device (fh_sched.look_every = 0).  The events of a context live until fh_destroy, which leaves the registry first, under the same lock.
// (Never destroyed: a caller's static SolverHip may outlive the statics of this library at process exit.)
static std::mutex& live_mu() {
  static std::mutex* m = new std::mutex;
  return *m;
}
static std::vector<fh_ctx*>& live_list() {
  static std::vector<fh_ctx*>* v = new std::vector<fh_ctx*>;
  return *v;
}
static int other_launches_in_flight(const fh_ctx* me) {
  int others = 0;
  std::lock_guard<std::mutex> lock(live_mu());
  for (const fh_ctx* c : live_list()) {
    if (c == me || c->device != me->device) continue;
    const hipEvent_t e = c->last_end.load(std::memory_order_acquire);
    if (e && hipEventQuery(e) == hipErrorNotReady) others++;
  }
  (void)hipGetLastError();  // (hipErrorNotReady is an answer, not a failure: it must not be what the next hipGetLastError() reports)
  return others;
}

#define FH_HIP(call)                                                                            \
  do {                                                                                          \
    hipError_t e__ = (call);                                                                    \
    if (e__ != hipSuccess) {                                                                    \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e__);                            \
      return FH_ERR_DEVICE;                                                                     \
    }                                                                                           \
  } while (0)

// Makes the context's device current for the duration of an entry point (a process may drive several GPUs from one thread)
// and restores the caller's device on exit.
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  explicit DeviceScope(const fh_ctx* ctx) {
    if (ctx && ctx->device >= 0 && hipGetDevice(&prev) == hipSuccess && prev != ctx->device)
      switched = hipSetDevice(ctx->device) == hipSuccess;
  }
  ~DeviceScope() {
    if (switched) (void)hipSetDevice(prev);
  }
};

static int ensure(fh_ctx* ctx, int slot, size_t bytes) {
  if (bytes <= ctx->d_cap[slot]) return FH_OK;
  if (ctx->d_buf[slot]) {
    // a running launch of this context may still use the old buffer (workspace, ring): wait for it before freeing
    FH_HIP(hipStreamSynchronize(ctx->stream));
    FH_HIP(hipFree(ctx->d_buf[slot]));
  }
  ctx->d_buf[slot] = nullptr;
  ctx->d_cap[slot] = 0;
  size_t want = std::max(bytes, (size_t)4096);
  FH_HIP(hipMalloc(&ctx->d_buf[slot], want));
  ctx->d_cap[slot] = want;
  return FH_OK;
}

// Initialises the work-sharing state (ticket, counters, hand-off counters, ring sequence numbers) before the FIRST solve launch of
// a context and after a launch that could not be issued; every launch leaves the block initialised for the next one (its last
// workgroup resets it: solve_kernel), so nothing a launch did can poison the next.
__global__ void share_init_kernel(fh::ShareCtl* ctl, unsigned long long* seqs) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (unsigned)FH_QCAP) seqs[i] = (unsigned long long)i;
  if (i < sizeof(fh::ShareCtl) / 4) reinterpret_cast<unsigned int*>(ctl)[i] = 0u;
  if (i < (unsigned)FH_MAX_GRID) reinterpret_cast<unsigned int*>(seqs + FH_QCAP)[i] = 0u;  // claims[] (ShareArgs)
}

#ifndef FH_ORDER_WINDOW
#define FH_ORDER_WINDOW 2  // the launch order interleaves ranks inside windows of FH_TICKET_CHUNK x FH_ORDER_WINDOW x (resident workgroups) tickets
                           // (measured on C4, one launch alone: 1: 3.5-3.6 ms, 2: 3.1-3.2 ms, the whole batch as one window: 3.3 ms — and 13.1 instead of 12.2 ms on C5)
#endif
// One solve launch: NSEG selects the kernel instantiation, PAIRS the whole -> hand-off -> safe unit.
template <int NSEG, bool PAIRS>
static int launch_solve(fh_ctx* ctx, const fh_problem* d_problems, const fh_face* d_faces, fh_result* d_results, fh::SolveArgs ka) {
  using SV = fh::Solver<NSEG>;
  const int n = ka.n;
  const size_t lds = SV::lds_bytes(ka.max_faces);
  // persistent grid: what is resident at once (LDS-limited, <= 8 workgroups per CU)
  // LDS is handed out in granules of 1280 B (residency census on MI355X: 14 080 B admit 11 workgroups per CU, 14 336 B only 10)
  const size_t lds_alloc = (lds + 1279) / 1280 * 1280;
  // The instantiation compiled for two wavefronts per SIMD (all its registers, no scratch) whenever no more than 8 solves are
  // resident per CU anyway: because LDS admits no more (N = 15: 5 per CU — the C5 workload: +3.4 %), because the batch is no larger
  // (one vehicle's replan, SolverHip::genNewTraj: a single problem), or because the caller says so (fh_sched.workgroups_per_cu <= 8:
  // a batch alone on the device is done 13 % sooner).
  // (fh_sched.workgroups_per_cu > 8 asks for the three-wavefront build whatever the batch: the tests run both on the same inputs)
  const int wpc = ctx->sched.workgroups_per_cu;
  // (rule mode 2 — the hand-off asks the caller's unknown voxels — has its instantiations in the two-wavefront build only)
  const bool unk = PAIRS && ka.rule.mode == 2;
  const bool two_waves = FH_WAVES_PER_SIMD > 2 && (unk || (wpc > 0 ? wpc <= 8 : ((160 * 1024) / lds_alloc <= 8 || n <= 8 * ctx->n_cu)));
  auto kern = unk ? fh::solve_kernel<NSEG, PAIRS, 2, PAIRS> : (two_waves ? fh::solve_kernel<NSEG, PAIRS, 2> : fh::solve_kernel<NSEG, PAIRS>);
  int per_cu = (int)std::min<size_t>((two_waves ? 2 : FH_WAVES_PER_SIMD) * 4, (160 * 1024) / lds_alloc);
  if (per_cu < 1) per_cu = 1;
  // fh_sched.workgroups_per_cu: fewer resident solves per CU than would fit.  The launch then asks for so much LDS that the hardware
  // cannot place more either, whatever else is in flight.
  size_t lds_launch = lds;
  if (ctx->sched.workgroups_per_cu > 0 && ctx->sched.workgroups_per_cu < per_cu) {
    per_cu = ctx->sched.workgroups_per_cu;
    lds_launch = std::max(lds, (size_t)(160 * 1024) / (size_t)per_cu / 1280 * 1280 - 16);
  }
  const int resident = ctx->n_cu * per_cu;
  ctx->last_launch = {NSEG, PAIRS ? 1 : 0, (two_waves || unk) ? 2 : FH_WAVES_PER_SIMD, 0, per_cu, (int32_t)lds_launch, unk ? 1 : 0, 0};  // (the UNK instantiation is always the two-wavefront one)
  const bool share = ctx->par.share != 0 && ctx->par.max_work == 0 && ctx->par.mip_gap == 0.0;
  // small batches get helper workgroups (one per CU) that take over subtrees of hard problems
  const int grid = share ? std::min(resident, std::max(n, ctx->n_cu)) : std::min(resident, n);
  ctx->last_launch.grid = grid;
  int rc;
  if ((rc = ensure(ctx, 5, sizeof(double) * (size_t)grid * ((size_t)NSEG * SV::SNAP_PADDED + 64))) != FH_OK) return rc;
  const size_t slot_stride = sizeof(fh::TaskHdr) + sizeof(double) * (size_t)SV::SNAP_PADDED;
  if (!ctx->d_buf[6]) ctx->ctl_ready = false;
  if ((rc = ensure(ctx, 6, 4096 + sizeof(unsigned long long) * FH_QCAP + sizeof(unsigned int) * FH_MAX_GRID)) != FH_OK) return rc;
  if ((rc = ensure(ctx, 8, slot_stride * FH_QCAP)) != FH_OK) return rc;
  if ((rc = ensure(ctx, 9, sizeof(fh::ShareRec) * FH_NRECS)) != FH_OK) return rc;
  fh::ShareArgs& sa = ka.sa;
  sa.ctl = reinterpret_cast<fh::ShareCtl*>(ctx->d_buf[6]);
  sa.seqs = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(ctx->d_buf[6]) + 4096);
  sa.slots = reinterpret_cast<unsigned char*>(ctx->d_buf[8]);
  sa.slot_stride = slot_stride;
  sa.recs = reinterpret_cast<fh::ShareRec*>(ctx->d_buf[9]);
  sa.host_abort = ctx->d_abort;
  sa.deadline_ticks = ctx->par.deadline_ms > 0 ? (unsigned long long)(ctx->par.deadline_ms * 1e5) : 0ull;  // 100 MHz clock
  sa.enabled = share && grid > 1 ? 1 : 0;
  sa.total_units = n;
  // Waiting workgroups.  Measured on C4 (32768 pairs per launch): 16 of them shorten a launch as much as 64 or 512 do (the tail is
  // bound by its critical path — the sequential factor trials of the hardest problem — not by hands), and every waiter holds
  // LDS that another launch of the same device could use (12 launches in flight: 7.4 M pairs/s with 16, 7.0 M with 64, 5.9 M with 256).
  // [r6] CUs / 64 (4): since held tickets are given away ahead of the takers (give_tickets) one launch alone ends as soon with 2-4 waiting
  // workgroups as with 16 (2.64-2.67 against 2.71 ms), and with launches in flight every waiter holds a slot their bulk could use
  // (23.45-23.6 against 23.3 M pairs/s)
  // (a batch of up to 8 problems per CU — one vehicle's replan: one problem and a helper workgroup per CU — keeps CUs / 16: its helpers ARE the waiters)
  sa.max_hungry = ctx->sched.waiting_workgroups > 0 ? ctx->sched.waiting_workgroups : (n <= 8 * ctx->n_cu ? std::max(8, ctx->n_cu / 16) : std::max(2, ctx->n_cu / 64));
  sa.min_nodes = ctx->sched.min_nodes;
  // Frames published AHEAD of the takers (served by workgroups between two problems).  What a launch ends on are problems with
  // 1200-4300 active-set iterations (C4: ~140 of 32768 pairs, mostly safe problems that are infeasible for all ten factors) that
  // started early and ran on ONE wavefront until the fresh problems were exhausted.  A problem that has used 4x the mean number of
  // iterations of the units finished so far may therefore publish up to 32 frames / trial ranges ahead: one launch alone 6.7 -> 6.1 ms,
  // 8 in flight -1 %.
  // (Publishing ahead from every problem with 64 nodes cost 13-25 % at any number of launches in flight: hop overhead in the bulk.)
  sa.backlog = ctx->sched.backlog;
  sa.giant_nodes = 1 << 30;
  sa.giant_factor = ctx->sched.publish_factor;
  sa.child_bound = ctx->sched.no_child_bound ? 0 : 1;
  sa.compact_results = ctx->sched.compact_results ? 1 : 0;
  sa.pair_outputs = ctx->sched.pair_outputs ? 1 : 0;
  // How often a tree looks around (fh_sched.look_every).  Measured on C4: with twelve launches in flight a period of 16 or 32 instead of 8
  // is +2.5 % (23.3 -> 23.8-24.05 M pairs/s: half as many frames change hands, 1100-1400 instead of 2300-2700 per launch, and a launch's
  // tail is hidden behind the other launches anyway); one launch ALONE ends 2-5 % later with 16 and 10 % later with 32-64 (2.64-2.71 ->
  // 2.71-2.81 -> 2.91-2.98 ms: its long trees find help later).  So a launch that is issued while another context has a solve launch in
  // flight on this device looks every 16th node, a launch that has the device to itself every 8th.
  const bool busy = other_launches_in_flight(ctx) > 0;  // another context of the process has a solve launch in flight on this device right now
  {
    const int period = ctx->sched.look_every > 0 ? ctx->sched.look_every : (busy ? FH_LOOK_EVERY_BUSY : FH_LOOK_EVERY);
    sa.look_mask = period - 1;
    ctx->last_launch.look_every = period;
  }
#ifdef FH_NO_DEAL  // (A/B builds: every ticket is drawn)
  sa.claims = nullptr;
#else
  sa.claims = grid <= FH_MAX_GRID ? reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned char*>(ctx->d_buf[6]) + 4096 + sizeof(unsigned long long) * FH_QCAP) : nullptr;
#endif
  sa.whole = PAIRS ? d_problems : nullptr;
  sa.wfaces = d_faces;
  sa.safe = PAIRS ? ka.safe : nullptr;
  sa.sfaces = PAIRS ? ka.sfaces : nullptr;
  sa.shrink = ka.shrink;
  sa.r_margin = ka.r_margin;
  ka.par = ctx->par;
  ka.workspace = (double*)ctx->d_buf[5];
  ka.basis = (const double*)ctx->d_buf[15];
  {  // raise the dynamic-LDS limit of this instantiation only when a launch needs more than any before it
    size_t& have = ctx->lds_attr[(NSEG <= 6 ? 0 : (NSEG <= 10 ? 1 : (NSEG <= 15 ? 2 : 3))) + (unk ? 16 : (PAIRS ? 4 : 0) + (two_waves ? 8 : 0))];
    if (lds_launch > have) {
      FH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_launch));
      have = lds_launch;
    }
  }
  if (ctx->ev_used + 2 > ctx->ev.size()) {
    if (ctx->ev.size() >= 8192) ctx->ev_used = 0;  // ring: keep the most recent launches only
    else
      for (int k = 0; k < 2; k++) {
        hipEvent_t e;
        FH_HIP(hipEventCreate(&e));
        ctx->ev.push_back(e);
      }
  }
  if (!ctx->ctl_ready) {
    hipLaunchKernelGGL(share_init_kernel, dim3((std::max(FH_QCAP, FH_MAX_GRID) + 255) / 256), dim3(256), 0, ctx->stream, sa.ctl, sa.seqs);
    FH_HIP(hipGetLastError());
  }
  ctx->ctl_ready = false;  // (true again once the launch below has been issued: it resets the block when it ends)
  // big batches are started hardest corridors first (order_kernel); results do not depend on the order.  [r6] ... when the launch has the
  // device to itself (fh_sched.launch_order = 1, the default; 2: always): the order keeps a launch from ENDING on a long tree, and with
  // other launches in flight that end is hidden behind them, while the two small launches that sort the batch queue behind the resident
  // grids (C4, fourteen in flight: 24.0 -> 24.25 M pairs/s without them)
  ka.order = nullptr;
  if (n >= 2048 && (ctx->sched.launch_order >= 2 || (ctx->sched.launch_order == 1 && !busy))) {
    const bool fresh = ctx->d_cap[13] < sizeof(int) * ((size_t)n + 128) || !ctx->order_ready;
    ctx->order_ready = false;  // (true again once all three launches below have been issued: a failed launch must not leave dirty counters behind)
    if ((rc = ensure(ctx, 13, sizeof(int) * ((size_t)n + 128))) != FH_OK) return rc;
    int* counters = (int*)ctx->d_buf[13];
    int* order = counters + 128;  // (2 * FH_ORDER_CLASSES + 1 = 73 counters)
    if (fresh) FH_HIP(hipMemsetAsync(counters, 0, sizeof(int) * 128, ctx->stream));  // afterwards the scatter kernel leaves them zeroed
    const unsigned blocks = (unsigned)((n + FH_ORDER_BLOCK - 1) / FH_ORDER_BLOCK);
    hipLaunchKernelGGL(fh::order_hist_kernel, dim3(blocks), dim3(FH_ORDER_BLOCK), 0, ctx->stream, d_problems, n, counters);
    hipLaunchKernelGGL(fh::order_scatter_kernel, dim3(blocks), dim3(FH_ORDER_BLOCK), 0, ctx->stream, d_problems, n, FH_TICKET_CHUNK * FH_ORDER_WINDOW * std::max(grid, 1), counters, order);
    FH_HIP(hipGetLastError());
    ka.order = order;
  }
  hipEvent_t e0 = ctx->ev[ctx->ev_used], e1 = ctx->ev[ctx->ev_used + 1];
  FH_HIP(hipEventRecord(e0, ctx->stream));
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds_launch, ctx->stream, d_problems, d_faces, d_results, ka);
  FH_HIP(hipGetLastError());
  FH_HIP(hipEventRecord(e1, ctx->stream));
  ctx->last_end.store(e1, std::memory_order_release);
  ctx->ev_used += 2;
  ctx->launched = true;
  ctx->ctl_ready = true;
  ctx->order_ready = ka.order != nullptr ? true : ctx->order_ready;
  ctx->last_grid = grid;
  return FH_OK;
}

// after the stream has been synchronised: did the last solve launch report a protocol failure?
static int check_share_error(fh_ctx* ctx) {
  if (!ctx->launched || !ctx->d_buf[6]) return FH_OK;
  ctx->launched = false;
  fh::ShareCtl h;
  FH_HIP(hipMemcpy(&h, ctx->d_buf[6], sizeof(h), hipMemcpyDeviceToHost));
  if (h.report[5]) {
    ctx->err = "solve kernel: work-sharing protocol failure (code " + std::to_string(h.report[5]) + "); results of the launch are incomplete";
    return FH_ERR_DEVICE;
  }
  return FH_OK;
}

// packed results: the words of an fh_result without the coefficient rows beyond n_seg (one thread per 8-byte word of the output)
__global__ void __launch_bounds__(256) pack_results_kernel(const double* __restrict__ in, double* __restrict__ out, long long total_words,
                                                           int words_out, int coeff_words) {
  const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
  if (w >= total_words) return;
  const long long r = w / words_out;
  const int k = (int)(w - r * words_out);
  constexpr int WORDS_IN = (int)(sizeof(fh_result) / 8), HEAD = 6, TAIL0 = WORDS_IN - 2;
  const int src = k < HEAD + coeff_words ? k : TAIL0 + (k - HEAD - coeff_words);
  out[w] = in[r * WORDS_IN + src];
}

extern "C" {

size_t fh_packed_result_size(int n_seg) { return (n_seg < 1 || n_seg > FH_MAX_SEG) ? 0 : (size_t)(64 + 96 * n_seg); }

int fh_pack_results_device(fh_ctx* ctx, const fh_result* d_results, int n, int n_seg, void* d_packed) {
  if (!ctx || n < 0 || n_seg < 1 || n_seg > FH_MAX_SEG) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  if (n == 0) return FH_OK;
  if (!d_results || !d_packed) return FH_ERR_ARG;
  const int words_out = (int)(fh_packed_result_size(n_seg) / 8);
  const long long total = (long long)n * words_out;
  hipLaunchKernelGGL(pack_results_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                     reinterpret_cast<const double*>(d_results), reinterpret_cast<double*>(d_packed), total, words_out, 12 * n_seg);
  FH_HIP(hipGetLastError());
  return FH_OK;
}

int fh_pack_results(const fh_result* results, int n, int n_seg, void* packed) {
  if (n < 0 || n_seg < 1 || n_seg > FH_MAX_SEG || (n > 0 && (!packed || !results))) return FH_ERR_ARG;
  const size_t rec = fh_packed_result_size(n_seg);
  unsigned char* p = static_cast<unsigned char*>(packed);
  for (int i = 0; i < n; i++, p += rec) {
    const fh_result& r = results[i];
    std::memcpy(p, &r, 48);
    std::memcpy(p + 48, &r.coeff[0][0], (size_t)96 * n_seg);
    std::memcpy(p + 48 + (size_t)96 * n_seg, &r.assign[0], 16);
  }
  return FH_OK;
}

int fh_unpack_results(const void* packed, int n, int n_seg, fh_result* results) {
  if (n < 0 || n_seg < 1 || n_seg > FH_MAX_SEG || (n > 0 && (!packed || !results))) return FH_ERR_ARG;
  const size_t rec = fh_packed_result_size(n_seg);
  const unsigned char* p = static_cast<const unsigned char*>(packed);
  for (int i = 0; i < n; i++, p += rec) {
    fh_result& r = results[i];
    std::memset(&r, 0, sizeof(r));
    std::memcpy(&r, p, 48);
    std::memcpy(&r.coeff[0][0], p + 48, (size_t)96 * n_seg);
    std::memcpy(&r.assign[0], p + 48 + (size_t)96 * n_seg, 16);
  }
  return FH_OK;
}

int fh_control_points(const fh_result* results, int n, int n_seg, double* cp) {
  if (n < 0 || n_seg < 1 || n_seg > FH_MAX_SEG || (n > 0 && (!results || !cp))) return FH_ERR_ARG;
  for (int i = 0; i < n; i++) {
    const fh_result& r = results[i];
    double* o = cp + (size_t)i * n_seg * 12;
    if (!r.solved) {
      std::memset(o, 0, sizeof(double) * (size_t)n_seg * 12);
      continue;
    }
    const double dt = r.dt;
    for (int t = 0; t < n_seg; t++, o += 12)
      for (int ax = 0; ax < 3; ax++) {
        const double a = r.coeff[t][0 + ax], b = r.coeff[t][3 + ax], c = r.coeff[t][6 + ax], d = r.coeff[t][9 + ax];
        const double Bn = b * dt * dt, Cn = c * dt, Dn = d;             // getBn / getCn / getDn
        o[0 + ax] = a * 0.0 * 0.0 * 0.0 + b * 0.0 * 0.0 + c * 0.0 + d;  // getCP0 = getPos(t, 0)
        o[3 + ax] = (Cn + 3 * Dn) / 3;                                  // getCP1
        o[6 + ax] = (Bn + 2 * Cn + 3 * Dn) / 3;                         // getCP2
        o[9 + ax] = a * dt * dt * dt + b * dt * dt + c * dt + d;        // getCP3 = getPos(t, dt)
      }
  }
  return FH_OK;
}

const char* fh_version(void) { return "fasterhip 0.4 gfx950"; }
int fh_abi_version(void) { return FH_ABI_VERSION; }

void fh_default_sched(fh_sched* s) {
  if (!s) return;
  std::memset(s, 0, sizeof(*s));
  s->launch_order = 1;
  s->publish_factor = 4;
  s->backlog = 32;
  s->waiting_workgroups = 0;
  s->min_nodes = 2;
  s->no_child_bound = 0;
  s->compact_results = 0;
  s->pair_outputs = 0;
  s->cloud_blocks = 1;
  s->look_every = 0;
  s->struct_size = (int32_t)sizeof(fh_sched);
}

int fh_set_sched(fh_ctx* ctx, const fh_sched* s) {
  if (!ctx || !s) return FH_ERR_ARG;
  if (s->struct_size != 0 && s->struct_size != (int32_t)sizeof(fh_sched)) {
    ctx->err = "fh_set_sched: fh_sched.struct_size " + std::to_string(s->struct_size) + " is not this library's " + std::to_string(sizeof(fh_sched)) +
               " (built against another round's fasterhip.h?)";
    return FH_ERR_ARG;
  }
  if (s->publish_factor < 0 || s->backlog < 0 || s->backlog > 512 || s->waiting_workgroups < 0 || s->min_nodes < 0 || s->workgroups_per_cu < 0) return FH_ERR_ARG;
  if (s->launch_order < 0 || s->launch_order > 2) return FH_ERR_ARG;
  if (s->look_every != 0 && (s->look_every < 2 || s->look_every > 1024 || (s->look_every & (s->look_every - 1)) != 0)) return FH_ERR_ARG;
  ctx->sched = *s;
  return FH_OK;
}

void fh_default_params(fh_params* p) {
  if (!p) return;
  p->feas_tol = 1e-9;
  p->dep_tol = 1e-10;
  p->max_nodes = 100000;
  p->max_iters = 2000;
  p->max_work = 0;
  p->share = 1;
  p->mip_gap = 0.0;
  p->deadline_ms = 0.0;
}

int fh_create(fh_ctx** out, int device) {
  if (!out) return FH_ERR_ARG;
  *out = nullptr;
  fh_ctx* ctx = new (std::nothrow) fh_ctx();
  if (!ctx) return FH_ERR_NOMEM;
  fh_default_params(&ctx->par);
  fh_default_sched(&ctx->sched);
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    // keep the context so that the caller can read the message; every other call will fail too
    ctx->err = std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    *out = ctx;
    ctx->device = -1;
    return FH_ERR_DEVICE;
  }
  if (device >= 0) {
    e = hipSetDevice(device);
    if (e != hipSuccess) {
      ctx->err = std::string("hipSetDevice: ") + hipGetErrorString(e);
      *out = ctx;
      ctx->device = -1;
      return FH_ERR_DEVICE;
    }
  }
  *out = ctx;
  FH_HIP(hipGetDevice(&ctx->device));
  {
    std::lock_guard<std::mutex> lock(live_mu());
    live_list().push_back(ctx);
  }
  hipDeviceProp_t prop;
  FH_HIP(hipGetDeviceProperties(&prop, ctx->device));
  ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  FH_HIP(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
  ctx->stream = ctx->own_stream;
  // the stop word lives in mapped, coherent host memory: StopExecution() is a plain store from any host thread
  FH_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_abort), 64, hipHostMallocMapped | hipHostMallocCoherent));
  *ctx->h_abort = 0u;
  FH_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->d_abort), ctx->h_abort, 0));
  FH_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_report), 64, hipHostMallocDefault));
  {  // the reduced-space basis tables: constants of N = 1..FH_MAX_SEG, computed here once and only read by the kernels
    const std::vector<double> tab = fh::build_basis_tables();
    int rc = ensure(ctx, 15, sizeof(double) * tab.size());
    if (rc != FH_OK) return rc;
    FH_HIP(hipMemcpy(ctx->d_buf[15], tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice));
  }
  return FH_OK;
}

void fh_destroy(fh_ctx* ctx) {
  if (!ctx) return;
  {
    std::lock_guard<std::mutex> lock(live_mu());
    live_list().erase(std::remove(live_list().begin(), live_list().end(), ctx), live_list().end());
  }
  if (ctx->device >= 0) {
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < 20; i++)
      if (ctx->d_buf[i]) (void)hipFree(ctx->d_buf[i]);
    if (ctx->h_abort) (void)hipHostFree(ctx->h_abort);
    if (ctx->h_report) (void)hipHostFree(ctx->h_report);
    for (hipEvent_t e : ctx->ev) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  }
  delete ctx;
}

const char* fh_last_error(const fh_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int fh_set_params(fh_ctx* ctx, const fh_params* p) {
  if (!ctx || !p) return FH_ERR_ARG;
  if (!(p->feas_tol > 0) || !(p->dep_tol > 0) || p->max_nodes < 1 || p->max_iters < 1 || p->max_work < 0) return FH_ERR_ARG;
  if (!(p->mip_gap >= 0) || !(p->mip_gap < 1) || !(p->deadline_ms >= 0) || !std::isfinite(p->deadline_ms)) return FH_ERR_ARG;
  ctx->par = *p;
  return FH_OK;
}

int fh_set_stream(fh_ctx* ctx, void* hip_stream) {
  if (!ctx) return FH_ERR_ARG;
  hipStream_t next = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  if (next != ctx->stream && ctx->device >= 0 && ctx->stream) {
    // one work queue / workspace per context: launches on the old stream must be over before the new stream may use them
    DeviceScope device_scope(ctx);
    FH_HIP(hipStreamSynchronize(ctx->stream));
  }
  ctx->stream = next;
  return FH_OK;
}

int fh_set_pair_margin(fh_ctx* ctx, double r_margin) {
  if (!ctx || !(r_margin == r_margin)) return FH_ERR_ARG;
  ctx->pair_margin = r_margin < 0 ? -1.0 : r_margin;
  return FH_OK;
}

int fh_set_pair_rule(fh_ctx* ctx, const fh_pair_rule* rule) {
  if (!ctx || !rule) return FH_ERR_ARG;
  if (rule->mode != 0 && rule->mode != 1 && rule->mode != 2) return FH_ERR_ARG;
  if (rule->mode == 1 && (!(rule->r_known > 0) || !(rule->drone_radius >= 0) || !(rule->delta_h > 0) || !(rule->delta_a > 0))) return FH_ERR_ARG;
  if (rule->mode == 2 && (!(rule->drone_radius > 0) || !(rule->delta_h > 0) || !(rule->delta_a > 0))) return FH_ERR_ARG;
  ctx->pair_rule = *rule;
  return FH_OK;
}

int fh_set_unknown_grid_device(fh_ctx* ctx, const fh_voxel_grid* grid, const unsigned char* d_flags) {
  if (!ctx) return FH_ERR_ARG;
  if (!d_flags) {  // none: rule mode 2 is refused until a grid is set again
    ctx->unknown.flags = nullptr;
    return FH_OK;
  }
  if (!grid || !(grid->res > 0) || grid->dims[0] < 1 || grid->dims[1] < 1 || grid->dims[2] < 1) return FH_ERR_ARG;
  if ((long long)grid->dims[0] * grid->dims[1] * grid->dims[2] > (1ll << 30)) return FH_ERR_ARG;
  ctx->unknown.flags = d_flags;
  ctx->unknown.ox = grid->origin[0]; ctx->unknown.oy = grid->origin[1]; ctx->unknown.oz = grid->origin[2]; ctx->unknown.res = grid->res;
  ctx->unknown.nx = grid->dims[0]; ctx->unknown.ny = grid->dims[1]; ctx->unknown.nz = grid->dims[2];
  return FH_OK;
}

int fh_request_stop(fh_ctx* ctx) {
  if (!ctx) return FH_ERR_ARG;
  if (ctx->device < 0 || !ctx->h_abort) return FH_ERR_DEVICE;
  __atomic_store_n(ctx->h_abort, 1u, __ATOMIC_RELEASE);
  return FH_OK;
}

int fh_clear_stop(fh_ctx* ctx) {
  if (!ctx) return FH_ERR_ARG;
  if (ctx->device < 0 || !ctx->h_abort) return FH_ERR_DEVICE;
  __atomic_store_n(ctx->h_abort, 0u, __ATOMIC_RELEASE);
  return FH_OK;
}

// diagnostic builds (-DFH_SHARE_PROFILE): the 16 profile words of the last launch (ticks of the 100 MHz clock / counts)
int fh_share_profile_read(fh_ctx* ctx, unsigned long long* out16) {
  if (!ctx || !out16) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  if (!ctx->d_buf[6]) return FH_ERR_ARG;
  FH_HIP(hipStreamSynchronize(ctx->stream));
  fh::ShareCtl h;
  FH_HIP(hipMemcpy(&h, ctx->d_buf[6], sizeof(h), hipMemcpyDeviceToHost));
  for (int i = 0; i < 8; i++) { out16[i] = h.prof[i]; out16[8 + i] = h.prof2[i]; }
  return FH_OK;
}

int fh_share_stats_read(fh_ctx* ctx, fh_share_stats* out) {
  if (!ctx || !out) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  std::memset(out, 0, sizeof(*out));
  if (!ctx->d_buf[6]) return FH_OK;
  FH_HIP(hipStreamSynchronize(ctx->stream));
  fh::ShareCtl h;
  FH_HIP(hipMemcpy(&h, ctx->d_buf[6], sizeof(h), hipMemcpyDeviceToHost));
  out->donated = h.report[0]; out->stolen = h.report[1]; out->queue_full = h.report[2]; out->records_full = h.report[3];
  out->records_used = std::min<unsigned>(h.report[4], FH_NRECS); out->error = h.report[5]; out->interrupted = h.report[6];
  out->workgroups = (uint32_t)ctx->last_grid;
  ctx->launched = false;
  if (h.report[5]) {
    ctx->err = "solve kernel: work-sharing protocol failure (code " + std::to_string(h.report[5]) + ")";
    return FH_ERR_DEVICE;
  }
  return FH_OK;
}

int fh_fp64_peak(fh_ctx* ctx, double* tflops) {
  if (!ctx || !tflops) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  int rc;
  const int blocks = ctx->n_cu * 8, threads = 256, iters = 20000;
  if ((rc = ensure(ctx, 3, sizeof(double) * (size_t)blocks * threads)) != FH_OK) return rc;
  hipEvent_t a, b;
  FH_HIP(hipEventCreate(&a));
  FH_HIP(hipEventCreate(&b));
  double best = 0;
  for (int rep = 0; rep < 4; rep++) {  // first repetition warms up clocks and code
    FH_HIP(hipEventRecord(a, ctx->stream));
    hipLaunchKernelGGL(fh::fp64_peak_kernel, dim3((unsigned)blocks), dim3(threads), 0, ctx->stream, (double*)ctx->d_buf[3], iters, 1.0);
    FH_HIP(hipGetLastError());
    FH_HIP(hipEventRecord(b, ctx->stream));
    FH_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    FH_HIP(hipEventElapsedTime(&ms, a, b));
    const double fl = 2.0 * 8.0 * (double)iters * (double)blocks * threads;
    if (rep > 0 && ms > 0) best = std::max(best, fl / (ms * 1e-3) / 1e12);
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  *tflops = best;
  return FH_OK;
}

int fh_sync(fh_ctx* ctx) {
  if (!ctx) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  FH_HIP(hipStreamSynchronize(ctx->stream));
  return check_share_error(ctx);
}

int fh_solve_batch_device(fh_ctx* ctx, const fh_problem* d_problems, const fh_face* d_faces, int n, int max_seg,
                          int max_faces, fh_result* d_results) {
  if (!ctx || n < 0) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  if (n == 0) return FH_OK;
  if (!d_problems || !d_results) return FH_ERR_ARG;
  if (max_seg <= 0 || max_seg > FH_MAX_SEG) max_seg = FH_MAX_SEG;
  if (max_faces <= 0 || max_faces > FH_MAX_FACES) max_faces = FH_MAX_FACES;
  max_faces = (max_faces + 7) & ~7;
  fh::SolveArgs ka;
  std::memset(&ka, 0, sizeof(ka));
  ka.n = n; ka.max_faces = max_faces;
  if (max_seg <= 6) return launch_solve<6, false>(ctx, d_problems, d_faces, d_results, ka);
  if (max_seg <= 10) return launch_solve<10, false>(ctx, d_problems, d_faces, d_results, ka);
  if (max_seg <= 15) return launch_solve<15, false>(ctx, d_problems, d_faces, d_results, ka);
  return launch_solve<FH_MAX_SEG, false>(ctx, d_problems, d_faces, d_results, ka);
}

int fh_solve_batch(fh_ctx* ctx, const fh_problem* problems, const fh_face* faces, int64_t n_faces, int n,
                   fh_result* results) {
  if (!ctx || n < 0 || n_faces < 0) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  if (n == 0) return FH_OK;
  if (!problems || !results || (n_faces > 0 && !faces)) return FH_ERR_ARG;
  int max_seg = 1, max_faces = 8;
  for (int i = 0; i < n; i++) {
    const fh_problem& p = problems[i];
    if (p.n_seg >= 1 && p.n_seg <= FH_MAX_SEG) max_seg = std::max(max_seg, (int)p.n_seg);
    if (p.n_poly >= 1 && p.n_poly <= FH_MAX_POLY) {
      const int nf = p.face_off[p.n_poly];
      if (nf >= 0 && nf <= FH_MAX_FACES) {
        // the kernel cannot see n_faces: reject corridors that point outside the face array here
        if (p.face_begin < 0 || (int64_t)p.face_begin + nf > n_faces) {
          ctx->err = "fh_solve_batch: problem " + std::to_string(i) + " addresses faces outside [0, n_faces)";
          return FH_ERR_ARG;
        }
        max_faces = std::max(max_faces, nf);
      }
    }
  }
  int rc;
  if ((rc = ensure(ctx, 0, sizeof(fh_problem) * (size_t)n)) != FH_OK) return rc;
  if ((rc = ensure(ctx, 1, sizeof(fh_face) * (size_t)std::max<int64_t>(n_faces, 1))) != FH_OK) return rc;
  if ((rc = ensure(ctx, 2, sizeof(fh_result) * (size_t)n)) != FH_OK) return rc;
  FH_HIP(hipMemcpyAsync(ctx->d_buf[0], problems, sizeof(fh_problem) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  if (n_faces > 0)
    FH_HIP(hipMemcpyAsync(ctx->d_buf[1], faces, sizeof(fh_face) * (size_t)n_faces, hipMemcpyHostToDevice, ctx->stream));
  rc = fh_solve_batch_device(ctx, (const fh_problem*)ctx->d_buf[0], (const fh_face*)ctx->d_buf[1], n, max_seg, max_faces,
                             (fh_result*)ctx->d_buf[2]);
  if (rc != FH_OK) return rc;
  FH_HIP(hipMemcpyAsync(results, ctx->d_buf[2], sizeof(fh_result) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  // the launch's report words ride along with the results (one synchronisation for a single genNewTraj())
  FH_HIP(hipMemcpyAsync(ctx->h_report, reinterpret_cast<unsigned char*>(ctx->d_buf[6]) + offsetof(fh::ShareCtl, report), 64,
                        hipMemcpyDeviceToHost, ctx->stream));
  FH_HIP(hipStreamSynchronize(ctx->stream));
  ctx->launched = false;
  if (ctx->h_report[5]) {
    ctx->err = "solve kernel: work-sharing protocol failure (code " + std::to_string(ctx->h_report[5]) + "); results of the launch are incomplete";
    return FH_ERR_DEVICE;
  }
  return FH_OK;
}

int fh_solve_batch_speculative(fh_ctx* ctx, const fh_problem* problems, const fh_face* faces, int64_t n_faces, int n,
                               int width, fh_result* results) {
  if (!ctx || n < 0 || n_faces < 0) return FH_ERR_ARG;
  if (width <= 1 || ctx->par.max_work > 0) return fh_solve_batch(ctx, problems, faces, n_faces, n, results);
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  if (n == 0) return FH_OK;
  if (!problems || !results || (n_faces > 0 && !faces)) return FH_ERR_ARG;
  // the factors of every problem, accumulated exactly as the reference loop does (repeated += in double)
  struct Search {
    std::vector<double> factors;
    size_t next = 0;
    bool done = false, passthrough = false;
    long long nodes = 0, iters = 0, kflops = 0;
  };
  std::vector<Search> search((size_t)n);
  for (int i = 0; i < n; i++) {
    const fh_problem& p = problems[i];
    Search& s = search[(size_t)i];
    const bool window_ok = p.f_inc > 0 && std::isfinite(p.f_init) && std::isfinite(p.f_final) &&
                           (p.f_final - p.f_init) / p.f_inc <= (double)FH_MAX_TRIALS;
    if (!window_ok) { s.passthrough = true; continue; }  // the kernel reports FH_ST_BAD_INPUT
    for (double f = p.f_init; f <= p.f_final; f = f + p.f_inc) s.factors.push_back(f);
    if (s.factors.empty()) s.passthrough = true;          // empty window: zero trials, as the sequential search
  }
  std::vector<fh_problem> sub;
  std::vector<int> owner;
  std::vector<fh_result> sub_res;
  // ONE wall-clock budget for the whole search (fh_params.deadline_ms), not one per window of factors
  const double deadline_ms = ctx->par.deadline_ms;
  const auto t_start = std::chrono::steady_clock::now();
  auto end_search = [&](int i, const fh_result* last, int status) {  // the search of problem i stops here, not solved
    Search& s = search[(size_t)i];
    fh_result& out = results[i];
    if (last) out = *last;
    else std::memset(&out, 0, sizeof(out));
    out.solved = 0;
    out.status = status;
    out.trials = (int32_t)s.next;
    out.nodes = (int32_t)s.nodes;
    out.qp_iters = (int32_t)s.iters;
    out.kflops = (int32_t)std::min<long long>(s.kflops, 0x7fffffffLL);
    out.factor = 0.0;
    out.cost = 0.0;
    std::memset(out.coeff, 0, sizeof(out.coeff));
    for (int t = 0; t < FH_MAX_SEG; t++) out.assign[t] = -1;
    s.done = true;
  };
  for (;;) {
    if (deadline_ms > 0) {  // what is left of the budget goes to the next window; nothing left: the open searches end interrupted
      const double used = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
      if (used >= deadline_ms) {
        for (int i = 0; i < n; i++)
          if (!search[(size_t)i].done) end_search(i, nullptr, FH_ST_INTERRUPTED);
        break;
      }
      ctx->par.deadline_ms = deadline_ms - used;
    }
    sub.clear();
    owner.clear();
    for (int i = 0; i < n; i++) {
      Search& s = search[(size_t)i];
      if (s.done) continue;
      if (s.passthrough) { sub.push_back(problems[i]); owner.push_back(i); continue; }
      for (int k = 0; k < width && s.next + (size_t)k < s.factors.size(); k++) {
        fh_problem q = problems[i];
        q.f_init = q.f_final = s.factors[s.next + (size_t)k];
        q.f_inc = 1.0;
        sub.push_back(q);
        owner.push_back(i);
      }
    }
    if (sub.empty()) break;
    sub_res.resize(sub.size());
    const int rc = fh_solve_batch(ctx, sub.data(), faces, n_faces, (int)sub.size(), sub_res.data());
    ctx->par.deadline_ms = deadline_ms;
    if (rc != FH_OK) return rc;
    for (size_t a = 0; a < sub.size();) {
      const int i = owner[a];
      Search& s = search[(size_t)i];
      size_t b = a;
      while (b < sub.size() && owner[b] == i) b++;
      if (s.passthrough) {
        results[i] = sub_res[a];
        s.done = true;
      } else {
        for (size_t k = a; k < b && !s.done; k++) {
          const fh_result& r = sub_res[k];
          s.nodes += r.nodes;
          s.iters += r.qp_iters;
          s.kflops += r.kflops;
          s.next++;
          const bool last = s.next == s.factors.size();
          if (r.status == FH_ST_INTERRUPTED) {  // stop request or deadline: terminal, as GRB_INTERRUPTED ends genNewTraj's loop
            end_search(i, &r, FH_ST_INTERRUPTED);   // (a later factor of the window must not be returned as "the first feasible one")
          } else if (r.solved || r.status == FH_ST_BAD_INPUT || last) {
            results[i] = r;
            results[i].trials = r.status == FH_ST_BAD_INPUT ? 0 : (int32_t)s.next;
            results[i].nodes = (int32_t)s.nodes;
            results[i].qp_iters = (int32_t)s.iters;
            results[i].kflops = (int32_t)std::min<long long>(s.kflops, 0x7fffffffLL);
            s.done = true;
          }
        }
      }
      a = b;
    }
  }
  return FH_OK;
}

int fh_sample_batch_device(fh_ctx* ctx, const fh_problem* d_problems, const fh_result* d_results, int n, int max_samples,
                           fh_state* d_states, int32_t* d_counts) {
  if (!ctx || n < 0 || max_samples < 0) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  if (n == 0) return FH_OK;
  if (!d_problems || !d_results || !d_counts || (max_samples > 0 && !d_states)) return FH_ERR_ARG;
  hipLaunchKernelGGL(fh::sample_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, d_problems, d_results, n, max_samples,
                     d_states, d_counts);
  FH_HIP(hipGetLastError());
  return FH_OK;
}

int fh_sample_batch(fh_ctx* ctx, const fh_problem* problems, const fh_result* results, int n, int max_samples,
                    fh_state* states, int32_t* counts) {
  if (!ctx || n < 0 || max_samples < 0) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  if (n == 0) return FH_OK;
  if (!problems || !results || !counts || (max_samples > 0 && !states)) return FH_ERR_ARG;
  int rc;
  const size_t sbytes = sizeof(fh_state) * (size_t)n * (size_t)max_samples;
  if ((rc = ensure(ctx, 0, sizeof(fh_problem) * (size_t)n)) != FH_OK) return rc;
  if ((rc = ensure(ctx, 2, sizeof(fh_result) * (size_t)n)) != FH_OK) return rc;
  if ((rc = ensure(ctx, 3, std::max(sbytes, (size_t)16))) != FH_OK) return rc;
  if ((rc = ensure(ctx, 4, sizeof(int32_t) * (size_t)n)) != FH_OK) return rc;
  FH_HIP(hipMemcpyAsync(ctx->d_buf[0], problems, sizeof(fh_problem) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  FH_HIP(hipMemcpyAsync(ctx->d_buf[2], results, sizeof(fh_result) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  FH_HIP(hipMemsetAsync(ctx->d_buf[3], 0, std::max(sbytes, (size_t)16), ctx->stream));
  rc = fh_sample_batch_device(ctx, (const fh_problem*)ctx->d_buf[0], (const fh_result*)ctx->d_buf[2], n, max_samples,
                              (fh_state*)ctx->d_buf[3], (int32_t*)ctx->d_buf[4]);
  if (rc != FH_OK) return rc;
  if (sbytes) FH_HIP(hipMemcpyAsync(states, ctx->d_buf[3], sbytes, hipMemcpyDeviceToHost, ctx->stream));
  FH_HIP(hipMemcpyAsync(counts, ctx->d_buf[4], sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  FH_HIP(hipStreamSynchronize(ctx->stream));
  return FH_OK;
}

// getDTInitial for a batch: one wavefront per problem, the device function the solve kernels call (fh_solve.hip.hpp: dt_initial)
__global__ void __launch_bounds__(64) dt_initial_kernel(const fh_problem* __restrict__ problems, int n, double* __restrict__ dt) {
  const int lane = threadIdx.x;
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const double v = fh::dt_initial(problems[i], fh::X0Rec<fh_problem>{problems[i]}, lane);
    if (lane == 0) dt[i] = v;
  }
}

int fh_dt_initial_batch_device(fh_ctx* ctx, const fh_problem* d_problems, int n, double* d_dt) {
  if (!ctx || n < 0) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  if (n == 0) return FH_OK;
  if (!d_problems || !d_dt) return FH_ERR_ARG;
  hipLaunchKernelGGL(dt_initial_kernel, dim3((unsigned)std::min(n, 32 * ctx->n_cu)), dim3(64), 0, ctx->stream, d_problems, n, d_dt);
  FH_HIP(hipGetLastError());
  return FH_OK;
}

int fh_dt_initial_batch(fh_ctx* ctx, const fh_problem* problems, int n, double* dt) {
  if (!ctx || n < 0) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  if (n == 0) return FH_OK;
  if (!problems || !dt) return FH_ERR_ARG;
  int rc;
  if ((rc = ensure(ctx, 0, sizeof(fh_problem) * (size_t)n)) != FH_OK) return rc;
  if ((rc = ensure(ctx, 4, sizeof(double) * (size_t)n)) != FH_OK) return rc;
  FH_HIP(hipMemcpyAsync(ctx->d_buf[0], problems, sizeof(fh_problem) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = fh_dt_initial_batch_device(ctx, (const fh_problem*)ctx->d_buf[0], n, (double*)ctx->d_buf[4])) != FH_OK) return rc;
  FH_HIP(hipMemcpyAsync(dt, ctx->d_buf[4], sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  FH_HIP(hipStreamSynchronize(ctx->stream));
  return FH_OK;
}

int fh_pair_glue_device(fh_ctx* ctx, const fh_problem* d_whole, const fh_result* d_whole_results, const fh_face* d_faces,
                        int n, double r_frac, double shrink, int max_safe_poly, fh_problem* d_safe, fh_face* d_safe_faces) {
  if (!ctx || n < 0) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  if (n == 0) return FH_OK;
  if (!d_whole || !d_whole_results || !d_safe) return FH_ERR_ARG;
  if (max_safe_poly < 0 || max_safe_poly > FH_MAX_POLY || !(r_frac >= 0) || !(r_frac <= 1) || !(shrink >= 0)) return FH_ERR_ARG;
  if (ctx->pair_rule.mode == 2 && !ctx->unknown.flags) {
    ctx->err = "fh_pair_rule mode 2 needs the unknown voxels: fh_set_unknown_grid_device";
    return FH_ERR_ARG;
  }
  hipLaunchKernelGGL(fh::pair_glue_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, d_whole,
                     d_whole_results, d_faces, n, r_frac, shrink, max_safe_poly, ctx->pair_margin, ctx->pair_rule, d_safe, d_safe_faces, ctx->unknown);
  FH_HIP(hipGetLastError());
  return FH_OK;
}

int fh_append_plans_device(fh_ctx* ctx, const fh_problem* d_whole, const fh_result* d_whole_results, const fh_problem* d_safe,
                           const fh_result* d_safe_results, int n, double r_frac, int max_states, fh_state* d_plans, int32_t* d_counts,
                           int32_t* d_k_safe) {
  if (!ctx || n < 0 || max_states < 0 || !(r_frac >= 0) || !(r_frac <= 1)) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  if (n == 0) return FH_OK;
  if (!d_whole || !d_whole_results || !d_safe || !d_safe_results || !d_counts || (max_states > 0 && !d_plans)) return FH_ERR_ARG;
  if (ctx->pair_rule.mode == 2 && !ctx->unknown.flags) {
    ctx->err = "fh_pair_rule mode 2 needs the unknown voxels: fh_set_unknown_grid_device";
    return FH_ERR_ARG;
  }
  hipLaunchKernelGGL(fh::plan_append_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, d_whole, d_whole_results, d_safe, d_safe_results, n,
                     r_frac, ctx->pair_rule, max_states, d_plans, d_counts, d_k_safe, ctx->unknown);
  FH_HIP(hipGetLastError());
  return FH_OK;
}

int fh_next_goals_device(fh_ctx* ctx, const fh_state* d_plans, const int32_t* d_counts, int32_t* d_cursor, int n, int max_states, int ticks,
                         fh_state* d_goals, int32_t* d_ok) {
  if (!ctx || n < 0 || max_states < 1 || ticks < 1) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  if (n == 0) return FH_OK;
  if (!d_plans || !d_counts || !d_cursor || !d_goals) return FH_ERR_ARG;
  hipLaunchKernelGGL(fh::next_goal_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_plans, d_counts, d_cursor, n, max_states,
                     ticks, d_goals, d_ok);
  FH_HIP(hipGetLastError());
  return FH_OK;
}

int fh_solve_pairs_device(fh_ctx* ctx, const fh_problem* d_whole, const fh_face* d_faces, int n, int max_seg, int max_faces,
                          double r_frac, double shrink, int max_safe_poly, fh_result* d_whole_results, fh_problem* d_safe,
                          fh_face* d_safe_faces, fh_result* d_safe_results) {
  if (!ctx || n < 0) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  if (n == 0) return FH_OK;
  if (!d_whole || !d_whole_results || !d_safe || !d_safe_results) return FH_ERR_ARG;
  if (max_safe_poly < 0 || max_safe_poly > FH_MAX_POLY || !(r_frac >= 0) || !(r_frac <= 1) || !(shrink >= 0)) return FH_ERR_ARG;
  if (ctx->pair_rule.mode == 2 && !ctx->unknown.flags) {
    ctx->err = "fh_pair_rule mode 2 needs the unknown voxels: fh_set_unknown_grid_device";
    return FH_ERR_ARG;
  }
  if (max_seg <= 0 || max_seg > FH_MAX_SEG) max_seg = FH_MAX_SEG;
  if (max_faces <= 0 || max_faces > FH_MAX_FACES) max_faces = FH_MAX_FACES;
  max_faces = (max_faces + 7) & ~7;
  fh::SolveArgs ka;
  std::memset(&ka, 0, sizeof(ka));
  ka.n = n; ka.max_faces = max_faces;
  ka.safe = d_safe; ka.sfaces = d_safe_faces; ka.sres = d_safe_results;
  ka.r_frac = r_frac; ka.shrink = shrink; ka.max_safe_poly = max_safe_poly; ka.r_margin = ctx->pair_margin; ka.rule = ctx->pair_rule;
  ka.unknown = ctx->unknown;
  if (max_seg <= 6) return launch_solve<6, true>(ctx, d_whole, d_faces, d_whole_results, ka);
  if (max_seg <= 10) return launch_solve<10, true>(ctx, d_whole, d_faces, d_whole_results, ka);
  if (max_seg <= 15) return launch_solve<15, true>(ctx, d_whole, d_faces, d_whole_results, ka);
  return launch_solve<FH_MAX_SEG, true>(ctx, d_whole, d_faces, d_whole_results, ka);
}

int fh_last_launch(const fh_ctx* ctx, fh_launch_info* out) {
  if (!ctx || !out || ctx->last_launch.n_seg == 0) return FH_ERR_ARG;
  *out = ctx->last_launch;
  return FH_OK;
}

int fh_timing_reset(fh_ctx* ctx) {
  if (!ctx) return FH_ERR_ARG;
  ctx->ev_used = 0;
  return FH_OK;
}

int fh_timing_read(fh_ctx* ctx, double* ms, int cap) {
  if (!ctx || cap < 0 || (cap > 0 && !ms)) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  const int count = (int)(ctx->ev_used / 2);
  if (count == 0) return 0;
  FH_HIP(hipEventSynchronize(ctx->ev[ctx->ev_used - 1]));
  for (int i = 0; i < count && i < cap; i++) {
    float t = 0.f;
    FH_HIP(hipEventElapsedTime(&t, ctx->ev[2 * i], ctx->ev[2 * i + 1]));
    ms[i] = (double)t;
  }
  return count;
}

static int decompose_device(fh_ctx* ctx, const double* d_cloud_xyz, int n_cloud, const double* d_segments, int n_segments,
                            const double local_bbox[3], double drone_radius, double z_ground, int max_faces, fh_face* d_faces, int32_t* d_counts,
                            const fh::UnknownLattice& lat, const double* d_seg_spheres);

int fh_decompose_batch_device(fh_ctx* ctx, const double* d_cloud_xyz, int n_cloud, const double* d_segments, int n_segments,
                              const double local_bbox[3], double drone_radius, double z_ground, int max_faces, fh_face* d_faces,
                              int32_t* d_counts) {
  fh::UnknownLattice lat;
  std::memset(&lat, 0, sizeof(lat));
  return decompose_device(ctx, d_cloud_xyz, n_cloud, d_segments, n_segments, local_bbox, drone_radius, z_ground, max_faces, d_faces, d_counts, lat,
                          nullptr);
}

// lat.on + d_seg_spheres: the unknown voxels of a grid (cells farther than sphere[3] from sphere[0..2], per segment) are points of the
// decomposition as well, listed before the cloud (fh_safe.hip.hpp)
static int decompose_device(fh_ctx* ctx, const double* d_cloud_xyz, int n_cloud, const double* d_segments, int n_segments,
                            const double local_bbox[3], double drone_radius, double z_ground, int max_faces, fh_face* d_faces, int32_t* d_counts,
                            const fh::UnknownLattice& lat, const double* d_seg_spheres) {
  if (!ctx || n_cloud < 0 || n_segments < 0 || max_faces < 8 || !local_bbox) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  if (n_segments == 0) return FH_OK;
  if (!d_segments || !d_faces || !d_counts || (n_cloud > 0 && !d_cloud_xyz)) return FH_ERR_ARG;
  if (!(local_bbox[0] > 0) || !(local_bbox[1] > 0) || !(local_bbox[2] > 0) || !(drone_radius >= 0)) return FH_ERR_ARG;
  const int grid = std::min(n_segments, ctx->n_cu * 12);  // LDS: 10.5 KB per workgroup
  int rc;
  if ((rc = ensure(ctx, 7, sizeof(double) * (size_t)grid * (size_t)FH_DECOMP_WS_DOUBLES)) != FH_OK) return rc;
  // bounding boxes of the blocks of 64 cloud points: most blocks cannot touch a segment's local box and are skipped (same results)
  double* d_blocks = nullptr;
  const int n_blocks = (n_cloud + 63) / 64;
  if (n_blocks >= 8 && ctx->sched.cloud_blocks) {
    if ((rc = ensure(ctx, 14, sizeof(double) * 6 * (size_t)n_blocks)) != FH_OK) return rc;
    d_blocks = (double*)ctx->d_buf[14];
    hipLaunchKernelGGL(fh::cloud_blocks_kernel, dim3((unsigned)n_blocks), dim3(64), 0, ctx->stream, d_cloud_xyz, n_cloud, d_blocks);
    FH_HIP(hipGetLastError());
  }
  if ((rc = ensure(ctx, 18, 64)) != FH_OK) return rc;
  FH_HIP(hipMemsetAsync(ctx->d_buf[18], 0, sizeof(int), ctx->stream));  // the segment counter of this launch
#ifdef FHD_EXPERIMENT
  {
    const char* e = std::getenv("FHD_STOP_AFTER");
    const int v = e ? std::atoi(e) : 0;
    FH_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(fh::fhd_stop_after), &v, sizeof(int), 0, hipMemcpyHostToDevice, ctx->stream));
  }
#endif
  hipLaunchKernelGGL(fh::decomp_kernel, dim3((unsigned)grid), dim3(64), 0, ctx->stream, d_cloud_xyz, n_cloud, d_segments, n_segments,
                     local_bbox[0], local_bbox[1], local_bbox[2], drone_radius, z_ground, max_faces, (double*)ctx->d_buf[7], d_faces,
                     d_counts, d_blocks, lat, lat.on ? d_seg_spheres : nullptr, (int*)ctx->d_buf[18]);
  FH_HIP(hipGetLastError());
#ifdef FHD_EXPERIMENT
  if (std::getenv("FHD_HIST")) {  // (diagnostic: how long the lists of this launch were)
    unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0}, z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    FH_HIP(hipStreamSynchronize(ctx->stream));
    FH_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(fh::fhd_hist), sizeof(h)));
    FH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(fh::fhd_hist), z, sizeof(z)));
    std::fprintf(stderr, "FHD_HIST segments %d lattice %d | lists <=256: %llu <=1536: %llu <=16384: %llu more: %llu | points %llu cells swept %llu\n", n_segments, lat.on,
                 h[0], h[1], h[2], h[3], h[4], h[5]);
  }
#endif
  return FH_OK;
}

int fh_decompose_batch(fh_ctx* ctx, const double* cloud_xyz, int n_cloud, const double* segments, int n_segments,
                       const double local_bbox[3], double drone_radius, double z_ground, int max_faces, fh_face* faces, int32_t* counts) {
  if (!ctx || n_cloud < 0 || n_segments < 0 || max_faces < 8) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  if (n_segments == 0) return FH_OK;
  if (!segments || !faces || !counts || (n_cloud > 0 && !cloud_xyz)) return FH_ERR_ARG;
  int rc;
  const size_t cb = sizeof(double) * 3 * (size_t)std::max(n_cloud, 1), sb = sizeof(double) * 6 * (size_t)n_segments;
  const size_t fb = sizeof(fh_face) * (size_t)n_segments * (size_t)max_faces, nb = sizeof(int32_t) * (size_t)n_segments;
  if ((rc = ensure(ctx, 0, cb)) != FH_OK) return rc;
  if ((rc = ensure(ctx, 1, sb)) != FH_OK) return rc;
  if ((rc = ensure(ctx, 3, fb)) != FH_OK) return rc;
  if ((rc = ensure(ctx, 4, nb)) != FH_OK) return rc;
  if (n_cloud > 0) FH_HIP(hipMemcpyAsync(ctx->d_buf[0], cloud_xyz, sizeof(double) * 3 * (size_t)n_cloud, hipMemcpyHostToDevice, ctx->stream));
  FH_HIP(hipMemcpyAsync(ctx->d_buf[1], segments, sb, hipMemcpyHostToDevice, ctx->stream));
  FH_HIP(hipMemsetAsync(ctx->d_buf[3], 0, fb, ctx->stream));
  rc = fh_decompose_batch_device(ctx, (const double*)ctx->d_buf[0], n_cloud, (const double*)ctx->d_buf[1], n_segments, local_bbox,
                                 drone_radius, z_ground, max_faces, (fh_face*)ctx->d_buf[3], (int32_t*)ctx->d_buf[4]);
  if (rc != FH_OK) return rc;
  FH_HIP(hipMemcpyAsync(faces, ctx->d_buf[3], fb, hipMemcpyDeviceToHost, ctx->stream));
  FH_HIP(hipMemcpyAsync(counts, ctx->d_buf[4], nb, hipMemcpyDeviceToHost, ctx->stream));
  FH_HIP(hipStreamSynchronize(ctx->stream));
  return FH_OK;
}

// ---- corridors of a batch of paths: segments -> decomposition -> polytope rows in the layout fh_problem points at ---------------
namespace {
// segment j of pair i = (vertex j, vertex j+1) of its path, NaN where the path has no such leg; a path of more than max_poly legs is
// cut to its first max_poly legs (deleteVertexes, utils.cpp:1117-1124); goal = the last vertex kept
__global__ void corridor_segments_kernel(const double* __restrict__ paths, const int32_t* __restrict__ n_points, int n, int max_points,
                                         int max_poly, double* __restrict__ segments, double* __restrict__ goal) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * max_poly) return;
  const int i = (int)(t / max_poly), j = (int)(t % max_poly);
  const int np = n_points[i];
  double* sg = segments + 6 * t;
  if (np >= 2 && j + 1 < np) {
    const double* v = paths + 3 * ((size_t)i * max_points + j);
    for (int k = 0; k < 6; k++) sg[k] = v[k];
  } else {
    for (int k = 0; k < 6; k++) sg[k] = __builtin_nan("");
  }
  if (j == 0 && goal) {
    const int last = np - 1 < max_poly ? np - 1 : max_poly;
    for (int k = 0; k < 3; k++) goal[3 * (size_t)i + k] = np >= 2 ? paths[3 * ((size_t)i * max_points + last) + k] : __builtin_nan("");
  }
}

// one wavefront per pair: the rows of its polytopes back to back at faces[i * faces_per_problem ...], offsets as fh_problem.face_off
__global__ void __launch_bounds__(64) corridor_assemble_kernel(const int32_t* __restrict__ n_points, int n, int max_poly, int seg_cap,
                                                               const fh_face* __restrict__ seg_faces, const int32_t* __restrict__ seg_counts,
                                                               int faces_per_problem, fh_face* __restrict__ faces,
                                                               int32_t* __restrict__ face_off, int32_t* __restrict__ n_poly) {
  const int i = (int)blockIdx.x, lane = (int)threadIdx.x;
  if (i >= n) return;
  const int np = n_points[i];
  const int legs = np >= 2 ? (np - 1 < max_poly ? np - 1 : max_poly) : 0;  // (at most max_poly legs are kept: deleteVertexes)
  int total = 0;
  bool fits = legs > 0 && legs <= FH_MAX_POLY;
  for (int p = 0; p < legs && fits; p++) {
    const int c = seg_counts[(size_t)i * max_poly + p];
    if (c <= 0 || total + c > faces_per_problem) { fits = false; break; }
    const fh_face* src = seg_faces + ((size_t)i * max_poly + p) * seg_cap;
    for (int r = lane; r < c; r += 64) faces[(size_t)i * faces_per_problem + total + r] = src[r];
    total += c;
    if (lane == 0) face_off[9 * (size_t)i + p + 1] = total;
  }
  if (lane == 0) {
    face_off[9 * (size_t)i] = 0;
    if (fits) {
      for (int p = legs; p < FH_MAX_POLY; p++) face_off[9 * (size_t)i + p + 1] = total;
      n_poly[i] = legs;
    } else {
      for (int p = 0; p <= FH_MAX_POLY; p++) face_off[9 * (size_t)i + p] = 0;
      n_poly[i] = 0;
    }
  }
}
}  // namespace

int fh_corridor_batch_device(fh_ctx* ctx, const double* d_cloud_xyz, int n_cloud, const double* d_paths, const int32_t* d_n_points, int n,
                             int max_points, int max_poly, const double local_bbox[3], double drone_radius, double z_ground,
                             int faces_per_problem, fh_face* d_faces, int32_t* d_face_off, int32_t* d_n_poly, double* d_goal) {
  if (!ctx || n < 0 || n_cloud < 0 || max_points < 2 || max_poly < 1 || max_poly > FH_MAX_POLY || faces_per_problem < 8 || !local_bbox)
    return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  if (n == 0) return FH_OK;
  if (!d_paths || !d_n_points || !d_faces || !d_face_off || !d_n_poly || (n_cloud > 0 && !d_cloud_xyz)) return FH_ERR_ARG;
  const size_t nseg = (size_t)n * max_poly;
  if (nseg > (size_t)0x7fffffff) return FH_ERR_ARG;
  const int seg_cap = FH_MAX_FACES_POLY;
  int rc;
  if ((rc = ensure(ctx, 10, sizeof(double) * 6 * nseg)) != FH_OK) return rc;
  if ((rc = ensure(ctx, 11, sizeof(fh_face) * nseg * seg_cap)) != FH_OK) return rc;
  if ((rc = ensure(ctx, 12, sizeof(int32_t) * nseg)) != FH_OK) return rc;
  hipLaunchKernelGGL(corridor_segments_kernel, dim3((unsigned)((nseg + 255) / 256)), dim3(256), 0, ctx->stream, d_paths, d_n_points, n, max_points,
                     max_poly, (double*)ctx->d_buf[10], d_goal);
  FH_HIP(hipGetLastError());
  if ((rc = fh_decompose_batch_device(ctx, d_cloud_xyz, n_cloud, (const double*)ctx->d_buf[10], (int)nseg, local_bbox, drone_radius, z_ground,
                                      seg_cap, (fh_face*)ctx->d_buf[11], (int32_t*)ctx->d_buf[12])) != FH_OK)
    return rc;
  hipLaunchKernelGGL(corridor_assemble_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, d_n_points, n, max_poly, seg_cap,
                     (const fh_face*)ctx->d_buf[11], (const int32_t*)ctx->d_buf[12], faces_per_problem, d_faces, d_face_off, d_n_poly);
  FH_HIP(hipGetLastError());
  return FH_OK;
}

// Problem records from corridors (fh_corridor_batch_device's outputs): the polytope table and xf of record i — see include/fasterhip.h.
int fh_corridor_problems_device(fh_ctx* ctx, const int32_t* d_n_points, const double* d_last_vertex, const double* d_goals, const fh_face* d_faces,
                                const int32_t* d_face_off, const int32_t* d_n_poly, int n, int faces_per_problem, int n_seg,
                                fh_problem* d_problems) {
  if (!ctx || n < 0 || faces_per_problem < 8 || n_seg < 1 || n_seg > FH_MAX_SEG) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  if (n == 0) return FH_OK;
  if (!d_n_points || !d_last_vertex || !d_goals || !d_faces || !d_face_off || !d_n_poly || !d_problems) return FH_ERR_ARG;
  if ((size_t)n * (size_t)faces_per_problem > (size_t)0x7fffffff) return FH_ERR_ARG;
  hipLaunchKernelGGL(fh::safe_finalize_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, d_n_points, d_last_vertex, d_goals, d_faces, d_face_off,
                     d_n_poly, n, faces_per_problem, n_seg, d_problems);
  FH_HIP(hipGetLastError());
  return FH_OK;
}

// The safe corridor of Faster::replan (faster.cpp:446-524) for a batch of pairs: see fh_safe.hip.hpp and include/fasterhip.h.
int fh_safe_corridor_batch_device(fh_ctx* ctx, const fh_problem* d_whole, const fh_result* d_whole_results, const double* d_paths,
                                  const int32_t* d_n_points, int max_points, const double* d_goals, const double* d_cloud_xyz, int n_cloud,
                                  const fh_voxel_grid* grid, int n, double r_frac, int max_poly_safe, const double local_bbox[3],
                                  double drone_radius, double z_ground, int faces_per_problem, int n_seg_safe, fh_problem* d_safe,
                                  fh_face* d_safe_faces, double* d_safe_paths, int32_t* d_safe_n_points) {
  if (!ctx || n < 0 || n_cloud < 0 || max_points < 2 || max_points > fh::SAFE_PATH_CAP || max_poly_safe < 1 || max_poly_safe > FH_MAX_POLY ||
      faces_per_problem < 8 || !local_bbox || !grid || n_seg_safe < 1 || n_seg_safe > FH_MAX_SEG)
    return FH_ERR_ARG;
  if (!(grid->res > 0) || grid->dims[0] < 1 || grid->dims[1] < 1 || grid->dims[2] < 1 || !(r_frac >= 0) || !(r_frac <= 1)) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  if (n == 0) return FH_OK;
  if (!d_whole || !d_whole_results || !d_paths || !d_n_points || !d_goals || !d_safe || !d_safe_faces || (n_cloud > 0 && !d_cloud_xyz)) return FH_ERR_ARG;
  if (ctx->pair_rule.mode == 2 && !ctx->unknown.flags) {
    ctx->err = "fh_pair_rule mode 2 needs the unknown voxels: fh_set_unknown_grid_device";
    return FH_ERR_ARG;
  }
  if ((size_t)n * (size_t)faces_per_problem > (size_t)0x7fffffff) return FH_ERR_ARG;
  const int mp = max_poly_safe + 1;
  const size_t nseg = (size_t)n * max_poly_safe;
  // one buffer: safe paths [n][mp][3] | spheres [n][4] | goal M [n][3] | face_off [n][9] | n_poly [n] | np [n]
  const size_t o_paths = 0, o_sph = o_paths + sizeof(double) * 3 * mp * (size_t)n, o_goal = o_sph + sizeof(double) * 4 * (size_t)n,
               o_off = o_goal + sizeof(double) * 3 * (size_t)n, o_np = o_off + sizeof(int32_t) * 9 * (size_t)n,
               o_cnt = o_np + sizeof(int32_t) * (size_t)n, total = o_cnt + sizeof(int32_t) * (size_t)n;
  int rc;
  if ((rc = ensure(ctx, 16, total)) != FH_OK) return rc;
  if ((rc = ensure(ctx, 17, sizeof(double) * 4 * nseg)) != FH_OK) return rc;
  unsigned char* base = (unsigned char*)ctx->d_buf[16];
  double* w_paths = d_safe_paths ? d_safe_paths : (double*)(base + o_paths);
  double* w_sph = (double*)(base + o_sph);
  double* w_goal = (double*)(base + o_goal);
  int32_t* w_off = (int32_t*)(base + o_off);
  int32_t* w_npoly = (int32_t*)(base + o_np);
  int32_t* w_np = d_safe_n_points ? d_safe_n_points : (int32_t*)(base + o_cnt);
  hipLaunchKernelGGL(fh::safe_path_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, d_whole, d_whole_results, d_paths, d_n_points, n, max_points,
                     r_frac, ctx->pair_rule, max_poly_safe, d_safe, w_paths, w_np, w_sph, ctx->unknown);
  hipLaunchKernelGGL(fh::safe_spheres_kernel, dim3((unsigned)((nseg + 255) / 256)), dim3(256), 0, ctx->stream, w_sph, n, max_poly_safe,
                     (double*)ctx->d_buf[17]);
  FH_HIP(hipGetLastError());
  const int seg_cap = FH_MAX_FACES_POLY;
  if ((rc = ensure(ctx, 10, sizeof(double) * 6 * nseg)) != FH_OK) return rc;
  if ((rc = ensure(ctx, 11, sizeof(fh_face) * nseg * seg_cap)) != FH_OK) return rc;
  if ((rc = ensure(ctx, 12, sizeof(int32_t) * nseg)) != FH_OK) return rc;
  hipLaunchKernelGGL(corridor_segments_kernel, dim3((unsigned)((nseg + 255) / 256)), dim3(256), 0, ctx->stream, w_paths, w_np, n, mp, max_poly_safe,
                     (double*)ctx->d_buf[10], w_goal);
  FH_HIP(hipGetLastError());
  fh::UnknownLattice lat;
  lat.ox = grid->origin[0]; lat.oy = grid->origin[1]; lat.oz = grid->origin[2]; lat.res = grid->res;
  lat.nx = grid->dims[0]; lat.ny = grid->dims[1]; lat.nz = grid->dims[2]; lat.on = 1;
  lat.flags = nullptr;
  if (ctx->pair_rule.mode == 2) {  // the caller's unknown voxels (the lattice of THAT grid), not the sphere model
    lat.ox = ctx->unknown.ox; lat.oy = ctx->unknown.oy; lat.oz = ctx->unknown.oz; lat.res = ctx->unknown.res;
    lat.nx = ctx->unknown.nx; lat.ny = ctx->unknown.ny; lat.nz = ctx->unknown.nz;
    lat.flags = ctx->unknown.flags;
  }
  if ((rc = decompose_device(ctx, d_cloud_xyz, n_cloud, (const double*)ctx->d_buf[10], (int)nseg, local_bbox, drone_radius, z_ground, seg_cap,
                             (fh_face*)ctx->d_buf[11], (int32_t*)ctx->d_buf[12], lat, (const double*)ctx->d_buf[17])) != FH_OK)
    return rc;
  hipLaunchKernelGGL(corridor_assemble_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, w_np, n, max_poly_safe, seg_cap,
                     (const fh_face*)ctx->d_buf[11], (const int32_t*)ctx->d_buf[12], faces_per_problem, d_safe_faces, w_off, w_npoly);
  hipLaunchKernelGGL(fh::safe_finalize_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, w_np, w_goal, d_goals, d_safe_faces, w_off, w_npoly, n,
                     faces_per_problem, n_seg_safe, d_safe);
  FH_HIP(hipGetLastError());
  return FH_OK;
}

double fh_last_kernel_ms(fh_ctx* ctx) {
  if (!ctx || ctx->device < 0 || ctx->ev_used < 2) return -1.0;
  if (hipEventSynchronize(ctx->ev[ctx->ev_used - 1]) != hipSuccess) return -1.0;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, ctx->ev[ctx->ev_used - 2], ctx->ev[ctx->ev_used - 1]) != hipSuccess) return -1.0;
  return (double)ms;
}

}  // extern "C"
