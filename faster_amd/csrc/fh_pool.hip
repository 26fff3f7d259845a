// fh_pool.hip — ONE batch over several GPUs of a node, behind the C ABI (include/fasterhip.h, fh_pool_*).
//
// every genNewTraj() is independent and a whole+safe pair never leaves its GPU, so a batch is cut into
// contiguous blocks (device g solves problems [g*ceil(n/G), (g+1)*ceil(n/G)) — the same rule as faster_amd/shard.py), every
// device gets its block straight from the caller's host arrays (asynchronous H2D on the device's own stream: a scatter needs
// no collective), solves it with the single-device entry points, and the complete fh_result blocks come back either to the
// caller's host array or, for a device-resident consumer, into ONE device's memory with peer copies over xGMI
// (hipMemcpyPeerAsync; the all-to-one "gather" of the batch).  One host thread per device drives its stream, so the copies
// and solves of the devices overlap even from pageable host memory.  There is no CPU fallback: no device, no pool.
//
// The one-process-per-GPU form of the same thing (torch.distributed, RCCL all_gather of the result blocks) is bench.py
// --scaling strong; both use the block partition, so results are identical to the 1-way run by construction
// (tests/test_distributed_gloo.py, tests/test_gpu_parity.py::test_pool_shards_equal_one_way).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fasterhip.h"

namespace {

struct PoolDev {
  int device = -1;
  fh_ctx* ctx = nullptr;
  hipStream_t stream = nullptr;
  void* buf[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // problems, faces, results, safe problems, safe faces, safe results,
  size_t cap[7] = {0, 0, 0, 0, 0, 0, 0};                                            // the unknown voxel flags (fh_pool_set_unknown_grid)
  std::string err;
  int rc = FH_OK;
};

}  // namespace

struct fh_pool {
  std::vector<PoolDev> dev;
  std::string err;
};

namespace {

#define POOL_HIP(call)                                                            \
  do {                                                                            \
    hipError_t e__ = (call);                                                      \
    if (e__ != hipSuccess) {                                                      \
      d.err = std::string(#call) + ": " + hipGetErrorString(e__);                 \
      d.rc = FH_ERR_DEVICE;                                                       \
      return;                                                                     \
    }                                                                             \
  } while (0)

bool grow(PoolDev& d, int slot, size_t bytes) {
  if (bytes <= d.cap[slot]) return true;
  if (d.buf[slot]) (void)hipFree(d.buf[slot]);
  d.buf[slot] = nullptr;
  d.cap[slot] = 0;
  const size_t want = std::max(bytes, (size_t)4096);
  if (hipMalloc(&d.buf[slot], want) != hipSuccess) return false;
  d.cap[slot] = want;
  return true;
}

// contiguous block partition (pairs are never split): the rule of faster_amd/shard.py::shard_range
void shard_range(int n, int g, int G, int& lo, int& hi) {
  const int per = (n + G - 1) / G;
  lo = std::min(g * per, n);
  hi = std::min(lo + per, n);
}

struct Job {
  const fh_problem* problems;
  const fh_face* faces;
  int64_t n_faces;
  int n;
  fh_result* results;        // host array of n records, or null
  fh_result* d_results_root; // device array of n records on dev[root], or null
  int root;
  // pairs
  bool pairs;
  const fh_problem* safe_templates;
  double r_frac, shrink;
  int max_safe_poly;
  fh_result* safe_results;
  fh_result* d_safe_results_root;
};

void run_shard(fh_pool* pool, int g, const Job& job) {
  PoolDev& d = pool->dev[(size_t)g];
  d.rc = FH_OK;
  d.err.clear();
  int lo, hi;
  shard_range(job.n, g, (int)pool->dev.size(), lo, hi);
  const int m = hi - lo;
  if (m <= 0) return;
  POOL_HIP(hipSetDevice(d.device));
  // this shard's rows of the face array, and its problems rebased onto them
  std::vector<fh_problem> pr(job.problems + lo, job.problems + hi);
  int64_t f_lo = job.n_faces, f_hi = 0;
  int max_seg = 1, max_faces = 8;
  for (const fh_problem& p : pr) {
    if (p.n_seg >= 1 && p.n_seg <= FH_MAX_SEG) max_seg = std::max(max_seg, (int)p.n_seg);
    if (p.n_poly < 1 || p.n_poly > FH_MAX_POLY) continue;
    const int nf = p.face_off[p.n_poly];
    if (nf < 0 || nf > FH_MAX_FACES) continue;  // the kernel reports FH_ST_BAD_INPUT
    if (p.face_begin < 0 || (int64_t)p.face_begin + nf > job.n_faces) {
      d.err = "fh_pool: a problem addresses faces outside [0, n_faces)";
      d.rc = FH_ERR_ARG;
      return;
    }
    f_lo = std::min<int64_t>(f_lo, p.face_begin);
    f_hi = std::max<int64_t>(f_hi, (int64_t)p.face_begin + nf);
    max_faces = std::max(max_faces, nf);
  }
  if (f_hi < f_lo) f_lo = f_hi = 0;
  for (fh_problem& p : pr)
    if (p.n_poly >= 1 && p.n_poly <= FH_MAX_POLY && p.face_begin >= f_lo) p.face_begin -= (int32_t)f_lo;
  const size_t pb = sizeof(fh_problem) * (size_t)m, fb = sizeof(fh_face) * (size_t)std::max<int64_t>(f_hi - f_lo, 1), rb = sizeof(fh_result) * (size_t)m;
  if (!grow(d, 0, pb) || !grow(d, 1, fb) || !grow(d, 2, rb) || (job.pairs && (!grow(d, 3, pb) || !grow(d, 4, fb) || !grow(d, 5, rb)))) {
    d.err = "fh_pool: out of device memory";
    d.rc = FH_ERR_NOMEM;
    return;
  }
  POOL_HIP(hipMemcpyAsync(d.buf[0], pr.data(), pb, hipMemcpyHostToDevice, d.stream));
  if (f_hi > f_lo) POOL_HIP(hipMemcpyAsync(d.buf[1], job.faces + f_lo, sizeof(fh_face) * (size_t)(f_hi - f_lo), hipMemcpyHostToDevice, d.stream));
  int rc;
  if (job.pairs) {
    POOL_HIP(hipMemcpyAsync(d.buf[3], job.safe_templates + lo, pb, hipMemcpyHostToDevice, d.stream));
    rc = fh_solve_pairs_device(d.ctx, (const fh_problem*)d.buf[0], (const fh_face*)d.buf[1], m, max_seg, max_faces, job.r_frac, job.shrink,
                               job.max_safe_poly, (fh_result*)d.buf[2], (fh_problem*)d.buf[3], (fh_face*)d.buf[4], (fh_result*)d.buf[5]);
  } else {
    rc = fh_solve_batch_device(d.ctx, (const fh_problem*)d.buf[0], (const fh_face*)d.buf[1], m, max_seg, max_faces, (fh_result*)d.buf[2]);
  }
  if (rc != FH_OK) {
    d.err = fh_last_error(d.ctx);
    d.rc = rc;
    return;
  }
  // the gather: complete fh_result blocks to the host array, or into the root device's memory over xGMI
  if (job.results) POOL_HIP(hipMemcpyAsync(job.results + lo, d.buf[2], rb, hipMemcpyDeviceToHost, d.stream));
  if (job.pairs && job.safe_results) POOL_HIP(hipMemcpyAsync(job.safe_results + lo, d.buf[5], rb, hipMemcpyDeviceToHost, d.stream));
  if (job.d_results_root) {
    const int root_dev = pool->dev[(size_t)job.root].device;
    POOL_HIP(hipMemcpyPeerAsync(job.d_results_root + lo, root_dev, d.buf[2], d.device, rb, d.stream));
    if (job.pairs && job.d_safe_results_root)
      POOL_HIP(hipMemcpyPeerAsync(job.d_safe_results_root + lo, root_dev, d.buf[5], d.device, rb, d.stream));
  }
  rc = fh_sync(d.ctx);
  if (rc != FH_OK) {
    d.err = fh_last_error(d.ctx);
    d.rc = rc;
  }
}

int run_job(fh_pool* pool, const Job& job) {
  if (job.n == 0) return FH_OK;
  int prev = -1;
  (void)hipGetDevice(&prev);
  std::vector<std::thread> threads;
  const int G = (int)pool->dev.size();
  for (int g = 1; g < G; g++) threads.emplace_back(run_shard, pool, g, std::cref(job));
  run_shard(pool, 0, job);
  for (std::thread& t : threads) t.join();
  if (prev >= 0) (void)hipSetDevice(prev);
  for (const PoolDev& d : pool->dev)
    if (d.rc != FH_OK) {
      pool->err = "device " + std::to_string(d.device) + ": " + d.err;
      return d.rc;
    }
  return FH_OK;
}

}  // namespace

extern "C" {

int fh_pool_create(fh_pool** out, const int* devices, int n_devices) {
  if (!out) return FH_ERR_ARG;
  *out = nullptr;
  fh_pool* pool = new (std::nothrow) fh_pool();
  if (!pool) return FH_ERR_NOMEM;
  *out = pool;
  int count = 0;
  const hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    pool->err = std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    return FH_ERR_DEVICE;
  }
  if (n_devices <= 0) n_devices = count;
  int prev = -1;
  (void)hipGetDevice(&prev);
  for (int g = 0; g < n_devices; g++) {
    PoolDev d;
    d.device = devices ? devices[g] : g;
    if (d.device < 0 || d.device >= count) {
      pool->err = "fh_pool_create: device index out of range";
      return FH_ERR_ARG;
    }
    if (hipSetDevice(d.device) != hipSuccess || hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking) != hipSuccess) {
      pool->err = "fh_pool_create: cannot create a stream on device " + std::to_string(d.device);
      return FH_ERR_DEVICE;
    }
    const int rc = fh_create(&d.ctx, d.device);
    if (rc != FH_OK) {
      pool->err = d.ctx ? fh_last_error(d.ctx) : "fh_create failed";
      if (d.ctx) fh_destroy(d.ctx);
      return rc;
    }
    (void)fh_set_stream(d.ctx, d.stream);
    pool->dev.push_back(d);
  }
  // peer access for the device-resident gather (ignored where the pair of devices does not support it: the copy then stages)
  for (const PoolDev& a : pool->dev)
    for (const PoolDev& b : pool->dev)
      if (a.device != b.device) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, a.device, b.device) == hipSuccess && can && hipSetDevice(a.device) == hipSuccess)
          (void)hipDeviceEnablePeerAccess(b.device, 0);
      }
  (void)hipGetLastError();
  if (prev >= 0) (void)hipSetDevice(prev);
  return FH_OK;
}

void fh_pool_destroy(fh_pool* pool) {
  if (!pool) return;
  int prev = -1;
  (void)hipGetDevice(&prev);
  for (PoolDev& d : pool->dev) {
    (void)hipSetDevice(d.device);
    if (d.ctx) fh_destroy(d.ctx);
    for (int i = 0; i < 7; i++)
      if (d.buf[i]) (void)hipFree(d.buf[i]);
    if (d.stream) (void)hipStreamDestroy(d.stream);
  }
  if (prev >= 0) (void)hipSetDevice(prev);
  delete pool;
}

int fh_pool_size(const fh_pool* pool) { return pool ? (int)pool->dev.size() : 0; }
const char* fh_pool_last_error(const fh_pool* pool) { return pool ? pool->err.c_str() : "null pool"; }

int fh_pool_set_params(fh_pool* pool, const fh_params* p) {
  if (!pool || !p) return FH_ERR_ARG;
  for (PoolDev& d : pool->dev) {
    const int rc = fh_set_params(d.ctx, p);
    if (rc != FH_OK) return rc;
  }
  return FH_OK;
}

int fh_pool_set_pair_margin(fh_pool* pool, double r_margin) {
  if (!pool) return FH_ERR_ARG;
  for (PoolDev& d : pool->dev) {
    const int rc = fh_set_pair_margin(d.ctx, r_margin);
    if (rc != FH_OK) return rc;
  }
  return FH_OK;
}

int fh_pool_set_pair_rule(fh_pool* pool, const fh_pair_rule* rule) {
  if (!pool || !rule) return FH_ERR_ARG;
  for (PoolDev& d : pool->dev) {
    const int rc = fh_set_pair_rule(d.ctx, rule);
    if (rc != FH_OK) return rc;
  }
  return FH_OK;
}

// Unknown space as an input for the pool (rule mode 2): the flags are host memory here — every device of the pool gets its own copy
// (synchronous) and its context is told (fh_set_unknown_grid_device).  flags = NULL: none.
int fh_pool_set_unknown_grid(fh_pool* pool, const fh_voxel_grid* grid, const unsigned char* flags) {
  if (!pool) return FH_ERR_ARG;
  if (flags && (!grid || grid->dims[0] < 1 || grid->dims[1] < 1 || grid->dims[2] < 1)) return FH_ERR_ARG;
  for (PoolDev& d : pool->dev) {
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(d.device);
    int rc = FH_OK;
    if (!flags) {
      rc = fh_set_unknown_grid_device(d.ctx, nullptr, nullptr);
    } else {
      const size_t bytes = (size_t)grid->dims[0] * (size_t)grid->dims[1] * (size_t)grid->dims[2];
      // the context must not keep pointing at a buffer that grow() may free: no launch of it is running (fh_sync waits for the CONTEXT's
      // stream, whichever it is), it forgets the old flags first, and learns the new ones only once they are in place
      (void)fh_sync(d.ctx);
      (void)hipStreamSynchronize(d.stream);
      (void)fh_set_unknown_grid_device(d.ctx, nullptr, nullptr);
      if (!grow(d, 6, bytes) || hipMemcpy(d.buf[6], flags, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = FH_ERR_DEVICE;
      else rc = fh_set_unknown_grid_device(d.ctx, grid, (const unsigned char*)d.buf[6]);
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    if (rc != FH_OK) {
      pool->err = "fh_pool_set_unknown_grid: device " + std::to_string(d.device);
      return rc;
    }
  }
  return FH_OK;
}

int fh_pool_solve_batch(fh_pool* pool, const fh_problem* problems, const fh_face* faces, int64_t n_faces, int n, fh_result* results,
                        int root, fh_result* d_results_root) {
  if (!pool || n < 0 || n_faces < 0) return FH_ERR_ARG;
  if (pool->dev.empty()) return FH_ERR_DEVICE;
  if (n == 0) return FH_OK;
  if (!problems || (!results && !d_results_root) || (n_faces > 0 && !faces)) return FH_ERR_ARG;
  if (d_results_root && (root < 0 || root >= (int)pool->dev.size())) return FH_ERR_ARG;
  Job job;
  std::memset(&job, 0, sizeof(job));
  job.problems = problems; job.faces = faces; job.n_faces = n_faces; job.n = n; job.results = results;
  job.d_results_root = d_results_root; job.root = root;
  return run_job(pool, job);
}

int fh_pool_solve_pairs(fh_pool* pool, const fh_problem* whole, const fh_face* faces, int64_t n_faces, int n,
                        const fh_problem* safe_templates, double r_frac, double shrink, int max_safe_poly, fh_result* whole_results,
                        fh_result* safe_results, int root, fh_result* d_whole_results_root, fh_result* d_safe_results_root) {
  if (!pool || n < 0 || n_faces < 0) return FH_ERR_ARG;
  if (pool->dev.empty()) return FH_ERR_DEVICE;
  if (n == 0) return FH_OK;
  if (!whole || !safe_templates || (n_faces > 0 && !faces)) return FH_ERR_ARG;
  if ((!whole_results || !safe_results) && (!d_whole_results_root || !d_safe_results_root)) return FH_ERR_ARG;
  if (d_whole_results_root && (root < 0 || root >= (int)pool->dev.size())) return FH_ERR_ARG;
  if (d_safe_results_root && !d_whole_results_root) return FH_ERR_ARG;  // (a device gather names both arrays or neither)
  Job job;
  std::memset(&job, 0, sizeof(job));
  job.problems = whole; job.faces = faces; job.n_faces = n_faces; job.n = n; job.results = whole_results;
  job.d_results_root = d_whole_results_root; job.root = root;
  job.pairs = true; job.safe_templates = safe_templates; job.r_frac = r_frac; job.shrink = shrink; job.max_safe_poly = max_safe_poly;
  job.safe_results = safe_results; job.d_safe_results_root = d_safe_results_root;
  return run_job(pool, job);
}

}  // extern "C"
