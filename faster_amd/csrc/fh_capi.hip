// fh_capi.hip — host side of the C ABI (include/fasterhip.h): context, buffers, kernel dispatch.
// Built with hipcc for gfx950 only into faster_amd/libfasterhip.so.  No CPU fallback exists: every entry
// point needs a HIP device and reports FH_ERR_DEVICE otherwise.
#include <hip/hip_runtime.h>

#include <algorithm>
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

struct fh_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::vector<hipEvent_t> ev;  // pairs (start, stop), one pair per solve-kernel launch
  size_t ev_used = 0;          // events in use since the last fh_timing_reset
  fh_params par;
  std::string err;
  // staging buffers of the host-pointer entry points (grown on demand, reused)
  void* d_buf[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t d_cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int n_cu = 0;
  unsigned long long ticket_base = 0;  // tickets handed out by all previous solve launches of this context
  size_t lds_attr[3] = {0, 0, 0};      // largest dynamic-LDS size already set per kernel instantiation
};

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
  if (ctx->d_buf[slot]) FH_HIP(hipFree(ctx->d_buf[slot]));
  ctx->d_buf[slot] = nullptr;
  ctx->d_cap[slot] = 0;
  size_t want = std::max(bytes, (size_t)4096);
  FH_HIP(hipMalloc(&ctx->d_buf[slot], want));
  ctx->d_cap[slot] = want;
  return FH_OK;
}

template <int NSEG>
static int launch_solve(fh_ctx* ctx, const fh_problem* d_problems, const fh_face* d_faces, int n, int max_faces,
                        fh_result* d_results) {
  size_t lds = fh::Solver<NSEG>::lds_bytes(max_faces);
  if (const char* pad = getenv("FH_DEBUG_LDS_PAD")) lds += (size_t)atoi(pad);  // occupancy experiments only
  auto kern = fh::solve_kernel<NSEG>;
  // persistent grid: what is resident at once (LDS-limited, <= 8 workgroups per CU), never more than the batch
  int per_cu = (int)std::min<size_t>(8, (160 * 1024) / lds);
  if (per_cu < 1) per_cu = 1;
  const int grid = std::min(n, ctx->n_cu * per_cu);
  // slot 5: snapshot workspace (one slot per tree level per workgroup), slot 6: the work counter
  int rc;
  if ((rc = ensure(ctx, 5, sizeof(double) * (size_t)grid * NSEG * fh::Solver<NSEG>::SNAP_PADDED)) != FH_OK) return rc;
  if (!ctx->d_buf[6]) {  // the ticket counter: zeroed once, never reset (see solve_kernel)
    if ((rc = ensure(ctx, 6, 256)) != FH_OK) return rc;
    FH_HIP(hipMemsetAsync(ctx->d_buf[6], 0, 256, ctx->stream));
    ctx->ticket_base = 0;
  }
  {  // raise the dynamic-LDS limit of this instantiation only when a launch needs more than any before it
    size_t& have = ctx->lds_attr[NSEG <= 6 ? 0 : (NSEG <= 10 ? 1 : 2)];
    if (lds > have) {
      FH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      have = lds;
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
  hipEvent_t e0 = ctx->ev[ctx->ev_used], e1 = ctx->ev[ctx->ev_used + 1];
  FH_HIP(hipEventRecord(e0, ctx->stream));
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds, ctx->stream, d_problems, d_faces, n, max_faces, ctx->par,
                     (double*)ctx->d_buf[5], (unsigned long long*)ctx->d_buf[6], ctx->ticket_base, d_results);
  ctx->ticket_base += (unsigned long long)n + (unsigned long long)grid;
  FH_HIP(hipGetLastError());
  FH_HIP(hipEventRecord(e1, ctx->stream));
  ctx->ev_used += 2;
  return FH_OK;
}

template <int NSEG>
static int launch_pairs(fh_ctx* ctx, const fh_problem* d_whole, const fh_face* d_faces, int n, int max_faces, double r_frac,
                        double shrink, int max_safe_poly, fh_result* d_wres, fh_problem* d_safe, fh_face* d_sfaces, fh_result* d_sres) {
  const size_t lds = fh::Solver<NSEG>::lds_bytes(max_faces);
  auto kern = fh::solve_pairs_kernel<NSEG>;
  int per_cu = (int)std::min<size_t>(8, (160 * 1024) / lds);
  if (per_cu < 1) per_cu = 1;
  const int grid = std::min(n, ctx->n_cu * per_cu);
  int rc;
  if ((rc = ensure(ctx, 5, sizeof(double) * (size_t)grid * NSEG * fh::Solver<NSEG>::SNAP_PADDED)) != FH_OK) return rc;
  if (!ctx->d_buf[6]) {
    if ((rc = ensure(ctx, 6, 256)) != FH_OK) return rc;
    FH_HIP(hipMemsetAsync(ctx->d_buf[6], 0, 256, ctx->stream));
    ctx->ticket_base = 0;
  }
  FH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (ctx->ev_used + 2 > ctx->ev.size()) {
    if (ctx->ev.size() >= 8192) ctx->ev_used = 0;
    else
      for (int k = 0; k < 2; k++) {
        hipEvent_t e;
        FH_HIP(hipEventCreate(&e));
        ctx->ev.push_back(e);
      }
  }
  hipEvent_t e0 = ctx->ev[ctx->ev_used], e1 = ctx->ev[ctx->ev_used + 1];
  FH_HIP(hipEventRecord(e0, ctx->stream));
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds, ctx->stream, d_whole, d_faces, n, max_faces, ctx->par,
                     (double*)ctx->d_buf[5], (unsigned long long*)ctx->d_buf[6], ctx->ticket_base, r_frac, shrink, max_safe_poly,
                     d_wres, d_safe, d_sfaces, d_sres);
  ctx->ticket_base += (unsigned long long)n + (unsigned long long)grid;
  FH_HIP(hipGetLastError());
  FH_HIP(hipEventRecord(e1, ctx->stream));
  ctx->ev_used += 2;
  return FH_OK;
}

extern "C" {

const char* fh_version(void) { return "fasterhip 0.1 gfx950"; }

void fh_default_params(fh_params* p) {
  if (!p) return;
  p->feas_tol = 1e-9;
  p->dep_tol = 1e-10;
  p->max_nodes = 100000;
  p->max_iters = 2000;
  p->max_work = 0;
  p->reserved = 0;
}

int fh_create(fh_ctx** out, int device) {
  if (!out) return FH_ERR_ARG;
  *out = nullptr;
  fh_ctx* ctx = new (std::nothrow) fh_ctx();
  if (!ctx) return FH_ERR_NOMEM;
  fh_default_params(&ctx->par);
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
  hipDeviceProp_t prop;
  FH_HIP(hipGetDeviceProperties(&prop, ctx->device));
  ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  FH_HIP(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
  ctx->stream = ctx->own_stream;
  return FH_OK;
}

void fh_destroy(fh_ctx* ctx) {
  if (!ctx) return;
  if (ctx->device >= 0) {
    for (int i = 0; i < 8; i++)
      if (ctx->d_buf[i]) (void)hipFree(ctx->d_buf[i]);
    for (hipEvent_t e : ctx->ev) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  }
  delete ctx;
}

const char* fh_last_error(const fh_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int fh_set_params(fh_ctx* ctx, const fh_params* p) {
  if (!ctx || !p) return FH_ERR_ARG;
  if (!(p->feas_tol > 0) || !(p->dep_tol > 0) || p->max_nodes < 1 || p->max_iters < 1 || p->max_work < 0) return FH_ERR_ARG;
  ctx->par = *p;
  return FH_OK;
}

int fh_set_stream(fh_ctx* ctx, void* hip_stream) {
  if (!ctx) return FH_ERR_ARG;
  ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  return FH_OK;
}

int fh_sync(fh_ctx* ctx) {
  if (!ctx) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  FH_HIP(hipStreamSynchronize(ctx->stream));
  return FH_OK;
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
  if (max_seg <= 6) return launch_solve<6>(ctx, d_problems, d_faces, n, max_faces, d_results);
  if (max_seg <= 10) return launch_solve<10>(ctx, d_problems, d_faces, n, max_faces, d_results);
  return launch_solve<FH_MAX_SEG>(ctx, d_problems, d_faces, n, max_faces, d_results);
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
  FH_HIP(hipStreamSynchronize(ctx->stream));
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
    long long nodes = 0, iters = 0;
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
  for (;;) {
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
          s.next++;
          const bool last = s.next == s.factors.size();
          if (r.solved || r.status == FH_ST_BAD_INPUT || last) {
            results[i] = r;
            results[i].trials = r.status == FH_ST_BAD_INPUT ? 0 : (int32_t)s.next;
            results[i].nodes = (int32_t)s.nodes;
            results[i].qp_iters = (int32_t)s.iters;
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

int fh_pair_glue_device(fh_ctx* ctx, const fh_problem* d_whole, const fh_result* d_whole_results, const fh_face* d_faces,
                        int n, double r_frac, double shrink, int max_safe_poly, fh_problem* d_safe, fh_face* d_safe_faces) {
  if (!ctx || n < 0) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  if (n == 0) return FH_OK;
  if (!d_whole || !d_whole_results || !d_safe) return FH_ERR_ARG;
  if (max_safe_poly < 0 || max_safe_poly > FH_MAX_POLY || !(r_frac >= 0) || !(r_frac <= 1) || !(shrink >= 0)) return FH_ERR_ARG;
  hipLaunchKernelGGL(fh::pair_glue_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, d_whole,
                     d_whole_results, d_faces, n, r_frac, shrink, max_safe_poly, d_safe, d_safe_faces);
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
  if (max_seg <= 0 || max_seg > FH_MAX_SEG) max_seg = FH_MAX_SEG;
  if (max_faces <= 0 || max_faces > FH_MAX_FACES) max_faces = FH_MAX_FACES;
  max_faces = (max_faces + 7) & ~7;
  if (max_seg <= 6)
    return launch_pairs<6>(ctx, d_whole, d_faces, n, max_faces, r_frac, shrink, max_safe_poly, d_whole_results, d_safe, d_safe_faces,
                           d_safe_results);
  if (max_seg <= 10)
    return launch_pairs<10>(ctx, d_whole, d_faces, n, max_faces, r_frac, shrink, max_safe_poly, d_whole_results, d_safe, d_safe_faces,
                            d_safe_results);
  return launch_pairs<FH_MAX_SEG>(ctx, d_whole, d_faces, n, max_faces, r_frac, shrink, max_safe_poly, d_whole_results, d_safe,
                                  d_safe_faces, d_safe_results);
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

int fh_decompose_batch_device(fh_ctx* ctx, const double* d_cloud_xyz, int n_cloud, const double* d_segments, int n_segments,
                              const double local_bbox[3], double drone_radius, double z_ground, int max_faces, fh_face* d_faces,
                              int32_t* d_counts) {
  if (!ctx || n_cloud < 0 || n_segments < 0 || max_faces < 8 || !local_bbox) return FH_ERR_ARG;
  if (ctx->device < 0) return FH_ERR_DEVICE;
  DeviceScope device_scope(ctx);
  if (n_segments == 0) return FH_OK;
  if (!d_segments || !d_faces || !d_counts || (n_cloud > 0 && !d_cloud_xyz)) return FH_ERR_ARG;
  if (!(local_bbox[0] > 0) || !(local_bbox[1] > 0) || !(local_bbox[2] > 0) || !(drone_radius >= 0)) return FH_ERR_ARG;
  const int grid = std::min(n_segments, ctx->n_cu * 4);  // LDS: 25 KB per workgroup
  int rc;
  if ((rc = ensure(ctx, 7, sizeof(double) * (size_t)grid * (3 * FH_DECOMP_CAP_GLOBAL + FH_DECOMP_CAP_GLOBAL / 8))) != FH_OK) return rc;
  hipLaunchKernelGGL(fh::decomp_kernel, dim3((unsigned)grid), dim3(64), 0, ctx->stream, d_cloud_xyz, n_cloud, d_segments, n_segments,
                     local_bbox[0], local_bbox[1], local_bbox[2], drone_radius, z_ground, max_faces, (double*)ctx->d_buf[7], d_faces,
                     d_counts);
  FH_HIP(hipGetLastError());
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

double fh_last_kernel_ms(fh_ctx* ctx) {
  if (!ctx || ctx->device < 0 || ctx->ev_used < 2) return -1.0;
  if (hipEventSynchronize(ctx->ev[ctx->ev_used - 1]) != hipSuccess) return -1.0;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, ctx->ev[ctx->ev_used - 2], ctx->ev[ctx->ev_used - 1]) != hipSuccess) return -1.0;
  return (double)ms;
}

}  // extern "C"
