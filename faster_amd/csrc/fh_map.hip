// fh_map.hip — host side of the voxel-map entry points of the C ABI (include/fasterhip.h, fh_map_*): occupancy grid from a point
// cloud and batched path search on the device (kernels: fh_path.hip.hpp).  Replaces JPS_Manager::updateJPSMap / solveJPS3D
// (/root/reference/faster/src/jps_manager.cpp:129-200) for batches of start/goal queries over one map.  No CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/fasterhip.h"
#include "../host/jps_tables.hpp"
#include "fh_path.hip.hpp"

struct fh_map {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
  int n_cu = 0;
  // the grid
  bool have_map = false;
  int nx = 0, ny = 0, nz = 0;
  double res = 0.1, origin[3] = {0, 0, 0}, inflation = 0.0;
  unsigned* d_bits = nullptr;
  size_t bits_cap = 0;
  // search workspace
  int waves = 0;
  size_t ws_total = 0;  // cells per wavefront the workspace was sized for
  fhp::CellState* d_cells = nullptr;
  unsigned long long* d_hkeys = nullptr;  // hashed cell records (fh_map_set_records): the keys of d_cells' slots
  int record_slots = -1;                  // fh_map_set_records: -1 by the size of the map, 0 one record per cell and wavefront, else hashed records, that many per wavefront
  int ws_slots = -1;                      // what the workspace was sized for
  size_t ws_bytes = 0;
  size_t chunk_words = 0;                 // words of d_chunks per wavefront
  unsigned* d_chunks = nullptr;
  unsigned* d_serials = nullptr;
  // [r6] A second search workspace, parked: the dense (one record per cell) workspace of the host-pointer entry point's retry of queries
  // that overflowed the hashed records.  The retry swaps it with the fields above for the duration of its launch (swap_workspaces) instead
  // of freeing and rebuilding the main workspace twice per call (up to 48 GB of memset each time: unreachable goals are common in replanning).
  struct ParkedWorkspace {
    int waves = 0;
    size_t ws_total = 0;
    fhp::CellState* d_cells = nullptr;
    unsigned long long* d_hkeys = nullptr;
    int ws_slots = -1;
    size_t ws_bytes = 0, chunk_words = 0;
    unsigned* d_chunks = nullptr;
    unsigned* d_serials = nullptr;
  } parked;
  int wave_cap = 0;                       // > 0: ensure_workspace sizes for at most this many wavefronts (the retry: a handful of queries)
  int* d_ticket = nullptr;
  int sched_waves_per_cu = 0, sched_launch_order = 1;  // fh_map_set_sched (0: 12 wavefronts per CU for A*, 20 for the jump point search)
  int* d_order = nullptr;  // 128 counters + launch order
  size_t order_cap = 0;
  double sphere_ra = 0.0;                 // fh_map_set_sphere
  int search_mode = 0;                    // fh_map_set_search: 0 A* with a total order, 1 jump point search in jps3d's order
  unsigned char* d_jps_tables = nullptr;  // neighbour tables of the jump point search (uploaded by the first fh_map_set_search(1))
  short* d_jps_entries = nullptr;         // jump tables of the current grid [cells][32] (fhp::jps_table_kernel), built by the first search in mode 1
  size_t entries_cap = 0;
  bool entries_valid = false;
  // staging of the host-pointer entry points
  void* d_stage[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t stage_cap[5] = {0, 0, 0, 0, 0};
};

#define FM_HIP(call)                                                   \
  do {                                                                 \
    hipError_t e__ = (call);                                           \
    if (e__ != hipSuccess) {                                           \
      m->err = std::string(#call) + ": " + hipGetErrorString(e__);     \
      return FH_ERR_DEVICE;                                            \
    }                                                                  \
  } while (0)

namespace {
struct MapDeviceScope {
  int prev = -1;
  bool switched = false;
  explicit MapDeviceScope(const fh_map* m) {
    if (m && hipGetDevice(&prev) == hipSuccess && prev != m->device) switched = hipSetDevice(m->device) == hipSuccess;
  }
  ~MapDeviceScope() {
    if (switched) (void)hipSetDevice(prev);
  }
};

int stage(fh_map* m, int slot, size_t bytes) {
  if (bytes <= m->stage_cap[slot]) return FH_OK;
  if (m->d_stage[slot]) {
    FM_HIP(hipStreamSynchronize(m->stream));
    FM_HIP(hipFree(m->d_stage[slot]));
  }
  m->d_stage[slot] = nullptr;
  m->stage_cap[slot] = 0;
  const size_t want = std::max(bytes, (size_t)4096);
  FM_HIP(hipMalloc(&m->d_stage[slot], want));
  m->stage_cap[slot] = want;
  return FH_OK;
}

// per-wavefront search state: sized by the grid (or, with hashed records, by the number of slots); at most 12 wavefronts per CU
// (3 per SIMD) and 48 GB
int ensure_workspace(fh_map* m) {
  const size_t total = (size_t)m->nx * m->ny * m->nz;
  // LDS: 12.5 KB per wavefront (A*: 12 per CU), 7.5 KB and 96 VGPRs (jump point search: 20 per CU)
  int waves = m->n_cu * (m->sched_waves_per_cu > 0 ? m->sched_waves_per_cu : (m->search_mode == 1 ? 20 : 12));
  int slots = m->search_mode == 1 ? m->record_slots : 0;
  const size_t budget = (size_t)48 << 30;
  if (slots < 0) {
    // by the size of the map: per-cell records while the budget still holds them for at least half of the wavefronts — fewer
    // wavefronts cost less than the table (measured: 816 k cells, 13 of 20 wavefronts per CU, are faster per cell; 1.45 M cells, 8 of
    // 20, are 1.6x faster hashed) and per-cell records have no limit on the cells a query may reach.  Else 131072 hashed slots
    // (5 MB per wavefront; a query may reach 98 304 cells).
    const size_t dense_per_wave = total * sizeof(fhp::CellState) + (size_t)fhp::NCHUNK * fhp::CHUNK_WORDS * 4;
    slots = budget / dense_per_wave >= (size_t)waves / 2 ? 0 : 131072;
  }
  const size_t records = slots > 0 ? (size_t)slots : total;
  // the chunk pool: open-list chunks (A*), the heap levels below LDS (jump point search: 78000 entries of 20 B — what the pool holds; with hashed records a
  // cell has one heap entry at most, so 3/4 of the slots), and the clean-up lists of the finished path (3 x MAXRAW ints)
  const size_t chunk_words = slots > 0 ? std::max<size_t>((size_t)(slots / 4 * 3) * 5, (size_t)3 * fhp::MAXRAW) + 16 : (size_t)fhp::NCHUNK * fhp::CHUNK_WORDS;
  const size_t per_wave = records * sizeof(fhp::CellState) + (slots > 0 ? (size_t)slots * 8 : 0) + chunk_words * 4;
  if (m->wave_cap > 0) waves = std::min(waves, m->wave_cap);
  if ((size_t)waves * per_wave > budget) waves = (int)std::max<size_t>(1, budget / per_wave);
  if (m->d_cells && m->ws_total == total && m->ws_slots == slots && m->waves >= 1) return FH_OK;  // (another grid size: other strides, stale stamps)
  FM_HIP(hipStreamSynchronize(m->stream));
  if (m->d_cells) FM_HIP(hipFree(m->d_cells));
  if (m->d_hkeys) FM_HIP(hipFree(m->d_hkeys));
  m->d_hkeys = nullptr;
  if (m->d_chunks) FM_HIP(hipFree(m->d_chunks));
  if (m->d_serials) FM_HIP(hipFree(m->d_serials));
  m->d_cells = nullptr; m->d_chunks = nullptr; m->d_serials = nullptr;
  m->waves = 0;
  FM_HIP(hipMalloc(&m->d_cells, (size_t)waves * records * sizeof(fhp::CellState)));
  if (slots > 0) {
    FM_HIP(hipMalloc(&m->d_hkeys, (size_t)waves * slots * 8));
    FM_HIP(hipMemsetAsync(m->d_hkeys, 0, (size_t)waves * slots * 8, m->stream));  // (serial numbers start at 1: every slot free)
  }
  FM_HIP(hipMalloc(&m->d_chunks, (size_t)waves * chunk_words * 4));
  m->chunk_words = chunk_words;
  FM_HIP(hipMalloc(&m->d_serials, (size_t)waves * 4));
  // stamps start at "never visited"; the serial numbers continue across calls, so this is the only clear
  FM_HIP(hipMemsetAsync(m->d_cells, 0, (size_t)waves * records * sizeof(fhp::CellState), m->stream));
  FM_HIP(hipMemsetAsync(m->d_serials, 0, (size_t)waves * 4, m->stream));
  if (!m->d_ticket) FM_HIP(hipMalloc(&m->d_ticket, 64));
  m->waves = waves;
  m->ws_total = total;
  m->ws_slots = slots;
  m->ws_bytes = (size_t)waves * per_wave;
  return FH_OK;
}
// exchanges the active search workspace with the parked one
void swap_workspaces(fh_map* m) {
  std::swap(m->waves, m->parked.waves);
  std::swap(m->ws_total, m->parked.ws_total);
  std::swap(m->d_cells, m->parked.d_cells);
  std::swap(m->d_hkeys, m->parked.d_hkeys);
  std::swap(m->ws_slots, m->parked.ws_slots);
  std::swap(m->ws_bytes, m->parked.ws_bytes);
  std::swap(m->chunk_words, m->parked.chunk_words);
  std::swap(m->d_chunks, m->parked.d_chunks);
  std::swap(m->d_serials, m->parked.d_serials);
}
// the retry of fh_map_plan_batch runs with per-cell records on the parked workspace; whatever way it ends, the map is back as it was
struct DenseRetryScope {
  fh_map* m;
  int saved_slots;
  explicit DenseRetryScope(fh_map* m_, int cap) : m(m_), saved_slots(m_->record_slots) {
    swap_workspaces(m);
    m->record_slots = 0;
    m->wave_cap = cap;
  }
  ~DenseRetryScope() {
    swap_workspaces(m);
    m->record_slots = saved_slots;
    m->wave_cap = 0;
  }
};
}  // namespace

extern "C" {

int fh_map_create(fh_map** out, int device) {
  if (!out) return FH_ERR_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return FH_ERR_DEVICE;
  fh_map* m = new (std::nothrow) fh_map();
  if (!m) return FH_ERR_NOMEM;
  m->device = device;
  MapDeviceScope scope(m);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess || hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete m;
    return FH_ERR_DEVICE;
  }
  m->n_cu = prop.multiProcessorCount;
  m->stream = m->own_stream;
  *out = m;
  return FH_OK;
}

void fh_map_destroy(fh_map* m) {
  if (!m) return;
  MapDeviceScope scope(m);
  (void)hipStreamSynchronize(m->stream);
  for (void* p : {(void*)m->d_bits, (void*)m->d_cells, (void*)m->d_hkeys, (void*)m->d_chunks, (void*)m->d_serials, (void*)m->d_ticket, (void*)m->d_order, (void*)m->d_jps_tables, (void*)m->d_jps_entries,
                  (void*)m->parked.d_cells, (void*)m->parked.d_hkeys, (void*)m->parked.d_chunks, (void*)m->parked.d_serials})
    if (p) (void)hipFree(p);
  for (void* p : m->d_stage)
    if (p) (void)hipFree(p);
  if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
  delete m;
}

const char* fh_map_last_error(const fh_map* m) { return m ? m->err.c_str() : "null map"; }

int fh_map_set_stream(fh_map* m, void* stream) {
  if (!m) return FH_ERR_ARG;
  MapDeviceScope scope(m);
  FM_HIP(hipStreamSynchronize(m->stream));
  m->stream = stream ? (hipStream_t)stream : m->own_stream;
  return FH_OK;
}

int fh_map_set_sched(fh_map* m, int waves_per_cu, int launch_order) {
  if (!m || waves_per_cu < 0 || waves_per_cu > 20) return FH_ERR_ARG;
  if (waves_per_cu != m->sched_waves_per_cu) m->ws_total = m->parked.ws_total = 0;  // the search workspace is sized by the number of wavefronts: reallocated by the next search
  m->sched_waves_per_cu = waves_per_cu;
  m->sched_launch_order = launch_order ? 1 : 0;
  return FH_OK;
}

// Which search fh_map_plan_batch* runs.  0 (default): A* with a total order of its own (an optimal path; equals the host restatement
// plan_path bit for bit).  1: jump point search with jps3d's pruning rules, successor order, comparator and heap (the optimal path
// FASTER itself gets from planner_ptr_->plan(start, goal, 1, true), jps_manager.cpp:166; equals plan_path_jps bit for bit).
int fh_map_set_search(fh_map* m, int mode) {
  if (!m || (mode != 0 && mode != 1)) return FH_ERR_ARG;
  MapDeviceScope scope(m);
  if (mode == 1 && !m->d_jps_tables) {
    // [27][28] natural neighbours, [27][12] cells to test, [27][12] directions to add; one byte per vector: (dx+1) | (dy+1) << 2 | (dz+1) << 4
    const fhfront::JpsTables& T = fhfront::jps_tables();
    std::vector<unsigned char> tab(27 * 28 + 2 * 27 * 12, 0);
    auto pk = [](const int v[3]) { return (unsigned char)((v[0] + 1) | ((v[1] + 1) << 2) | ((v[2] + 1) << 4)); };
    for (int id = 0; id < 27; id++) {
      for (int k = 0; k < 26; k++) tab[id * 28 + k] = pk(T.ns[id][k]);
      for (int k = 0; k < 12; k++) {
        tab[27 * 28 + id * 12 + k] = pk(T.f1[id][k]);
        tab[27 * 28 + 27 * 12 + id * 12 + k] = pk(T.f2[id][k]);
      }
    }
    FM_HIP(hipMalloc(&m->d_jps_tables, tab.size()));
    FM_HIP(hipMemcpy(m->d_jps_tables, tab.data(), tab.size(), hipMemcpyHostToDevice));
  }
  if (mode != m->search_mode) m->ws_total = m->parked.ws_total = 0;  // the two searches stamp the cell states differently: the workspace starts over
  m->search_mode = mode;
  return FH_OK;
}

// How the jump point search keeps its per-cell records (g, parent, direction, closed).  slots = 0: one record per cell of the
// map and wavefront (16 B x cells x wavefronts).  slots = a power of two in [1024, 2^22]: a hashed table of that many records per
// wavefront (24 B per slot + 15 B per slot of heap levels, whatever the size of the map) holding the cells the running query has
// reached; a query that reaches more than 3/4 of `slots` cells returns -2.  Same paths either way.  -1 (default): per-cell records
// while the 48 GB budget holds them for at least half of the wavefronts, else 131072 hashed slots.  Measured (65536 forest queries, profiles/
// r04_jps_records.json): 181 500 cells — per cell 54 ms / 23.8 GB, 8192 slots 66 ms / 1.6 GB; 1 452 000 cells — per cell 476 ms /
// 51.5 GB (2064 wavefronts fit the 48 GB budget), 32768 slots 298 ms / 6.5 GB.  The A* search (mode 0) always uses per-cell records.
int fh_map_set_records(fh_map* m, int slots) {
  if (!m || slots < -1 || (slots > 0 && (slots < 1024 || slots > (1 << 22) || (slots & (slots - 1)) != 0))) return FH_ERR_ARG;
  m->record_slots = slots;
  return FH_OK;
}

// Bytes of the search workspace as it stands (sized by the first search after a change of map, mode, records or scheduling).
long long fh_map_workspace_bytes(const fh_map* m) { return m ? (long long)m->ws_bytes : -1; }

// JPS_in of Faster::replan (faster.cpp:370-382): with ra > 0 every path is cut at its first crossing of the sphere of radius
// min(|goal - start| - 0.001, ra) around its start, the crossing point appended, BEFORE createMoreVertexes / deleteVertexes.  0 (default): off.
int fh_map_set_sphere(fh_map* m, double ra) {
  if (!m || !(ra >= 0.0)) return FH_ERR_ARG;
  m->sphere_ra = ra;
  return FH_OK;
}

int fh_map_sync(fh_map* m) {
  if (!m) return FH_ERR_ARG;
  MapDeviceScope scope(m);
  FM_HIP(hipStreamSynchronize(m->stream));
  return FH_OK;
}

// MapUtil::readMap (/root/reference/thirdparty/jps3d/include/jps_collision/map_util.h:30-185 — the function
// JPS_Manager::updateJPSMap calls, jps_manager.cpp:135): grid of cells[] cells (x, y widened by 5*inflation/res) centred on
// `center`, clipped to [z_ground, z_max]; the dimension arithmetic below is that function's (integer truncations included; the
// "+1" variant of faster/include/read_map.hpp's MapReader is not on FASTER's path).  Pinned to the reference's own compiled
// readMap — dimensions, origin, every cell, a z_ground-clipped map included — in tests/test_ref_frontend.py and
// tests/test_gpu_round3.py.
int fh_map_read_device(fh_map* m, const double* d_cloud_xyz, int n_cloud, const int32_t cells[3], double res, const double center[3],
                       double z_ground, double z_max, double inflation) {
  if (!m || !cells || !center || n_cloud < 0 || (n_cloud > 0 && !d_cloud_xyz) || !(res > 0.0) || !(inflation >= 0.0)) return FH_ERR_ARG;
  if (cells[0] <= 0 || cells[1] <= 0 || cells[2] <= 0) return FH_ERR_ARG;
  MapDeviceScope scope(m);
  int dx = cells[0] + (int)(5 * inflation / res), dy = cells[1] + (int)(5 * inflation / res), dz = cells[2];
  int down = (int)(dz / 2.0), up = (int)(dz / 2.0);
  if (center[2] - res * dz / 2.0 < z_ground) down = std::max((int)((center[2] - z_ground) / res), 0);
  if (center[2] + res * dz / 2.0 > z_max) {
    up = (int)((z_max - center[2]) / res);
    up = up > 0 ? up : 1;
  }
  dz = down + up;
  if (dz <= 0 || (long long)dx * dy * dz > (1ll << 27)) return FH_ERR_ARG;
  m->nx = dx; m->ny = dy; m->nz = dz;
  m->res = res;
  m->inflation = inflation;
  m->origin[0] = center[0] - res * dx / 2.0;
  m->origin[1] = center[1] - res * dy / 2.0;
  m->origin[2] = center[2] - res * down;
  const size_t total = (size_t)dx * dy * dz, words = (total + 31) / 32;
  if (words * 4 > m->bits_cap) {
    FM_HIP(hipStreamSynchronize(m->stream));
    if (m->d_bits) FM_HIP(hipFree(m->d_bits));
    m->d_bits = nullptr; m->bits_cap = 0;
    FM_HIP(hipMalloc(&m->d_bits, words * 4));
    m->bits_cap = words * 4;
  }
  FM_HIP(hipMemsetAsync(m->d_bits, 0, words * 4, m->stream));
  if (n_cloud > 0) {
    const int mcube = (int)std::floor(inflation / res);
    hipLaunchKernelGGL(fhp::mark_kernel, dim3((unsigned)((n_cloud + 255) / 256)), dim3(256), 0, m->stream, d_cloud_xyz, n_cloud, dx, dy, dz, res,
                       m->origin[0], m->origin[1], m->origin[2], mcube, m->d_bits);
    FM_HIP(hipGetLastError());
  }
  m->have_map = true;
  m->entries_valid = false;
  return FH_OK;
}

int fh_map_read(fh_map* m, const double* cloud_xyz, int n_cloud, const int32_t cells[3], double res, const double center[3], double z_ground,
                double z_max, double inflation) {
  if (!m || n_cloud < 0 || (n_cloud > 0 && !cloud_xyz)) return FH_ERR_ARG;
  MapDeviceScope scope(m);
  int rc;
  if (n_cloud > 0) {
    if ((rc = stage(m, 0, sizeof(double) * 3 * (size_t)n_cloud)) != FH_OK) return rc;
    FM_HIP(hipMemcpyAsync(m->d_stage[0], cloud_xyz, sizeof(double) * 3 * (size_t)n_cloud, hipMemcpyHostToDevice, m->stream));
  }
  if ((rc = fh_map_read_device(m, (const double*)m->d_stage[0], n_cloud, cells, res, center, z_ground, z_max, inflation)) != FH_OK) return rc;
  FM_HIP(hipStreamSynchronize(m->stream));
  return FH_OK;
}

int fh_map_dims(const fh_map* m, int32_t dims[3], double origin[3]) {
  if (!m || !m->have_map) return FH_ERR_ARG;
  if (dims) { dims[0] = m->nx; dims[1] = m->ny; dims[2] = m->nz; }
  if (origin) { origin[0] = m->origin[0]; origin[1] = m->origin[1]; origin[2] = m->origin[2]; }
  return FH_OK;
}

// occupancy as the reference stores it: one int8 per cell, 0 free / 100 occupied (x fastest)
int fh_map_occupancy(fh_map* m, int8_t* occ) {
  if (!m || !m->have_map || !occ) return FH_ERR_ARG;
  MapDeviceScope scope(m);
  const size_t total = (size_t)m->nx * m->ny * m->nz, words = (total + 31) / 32;
  std::vector<unsigned> bits(words);
  FM_HIP(hipMemcpyAsync(bits.data(), m->d_bits, words * 4, hipMemcpyDeviceToHost, m->stream));
  FM_HIP(hipStreamSynchronize(m->stream));
  for (size_t i = 0; i < total; i++) occ[i] = (bits[i >> 5] >> (i & 31)) & 1u ? 100 : 0;
  return FH_OK;
}

int fh_map_plan_batch_device(fh_map* m, const double* d_starts, const double* d_goals, int n, int max_points, double max_vertex_dist,
                             int max_poly, double* d_paths, int32_t* d_n_points, int64_t* d_expansions) {
  if (!m || !m->have_map || n < 0 || max_points < 2 || (n > 0 && (!d_starts || !d_goals || !d_paths || !d_n_points))) return FH_ERR_ARG;
  if (n == 0) return FH_OK;
  MapDeviceScope scope(m);
  int rc;
  if ((rc = ensure_workspace(m)) != FH_OK) return rc;
  fhp::MapView mv;
  mv.nx = m->nx; mv.ny = m->ny; mv.nz = m->nz;
  mv.total = m->nx * m->ny * m->nz;
  mv.inv_nxy = fhu::inverse(m->nx * m->ny); mv.inv_nx = fhu::inverse(m->nx);
  mv.m_free = (int)std::round((double)(float)m->inflation / m->res + 0.5);  // setFreeVoxelAndSurroundings(center, const float d), map_util.h:248-263
  mv.res = m->res; mv.ox = m->origin[0]; mv.oy = m->origin[1]; mv.oz = m->origin[2];
  mv.bits = m->d_bits;
  fhp::PlanArgs pa;
  pa.starts = d_starts; pa.goals = d_goals; pa.n = n; pa.max_points = max_points;
  pa.paths = d_paths; pa.n_points = d_n_points; pa.expansions = (long long*)d_expansions;
  pa.cells = m->d_cells; pa.hkeys = m->d_hkeys; pa.hslots = m->ws_slots > 0 ? m->ws_slots : 0; pa.chunk_words = (long long)m->chunk_words; pa.chunks = m->d_chunks; pa.serials = m->d_serials; pa.ticket = m->d_ticket;
  pa.max_vertex_dist = max_vertex_dist; pa.max_poly = max_poly;
  pa.jps_tables = m->d_jps_tables;
  pa.sphere_ra = m->sphere_ra;
  pa.profile_slot = -1;
#ifdef FHP_PROFILE
  if (const char* e = std::getenv("FHP_PROFILE_SLOT")) pa.profile_slot = std::atoi(e);
#endif
  FM_HIP(hipMemsetAsync(m->d_ticket, 0, 4, m->stream));
  pa.jps_entries = nullptr;
  if (m->search_mode == 1) {
    if (!m->entries_valid) {  // the jump tables of this grid: three small launches, once per map
      const size_t need = (size_t)32 * mv.total * sizeof(short);
      if (need > m->entries_cap) {
        FM_HIP(hipStreamSynchronize(m->stream));
        if (m->d_jps_entries) FM_HIP(hipFree(m->d_jps_entries));
        m->d_jps_entries = nullptr; m->entries_cap = 0;
        FM_HIP(hipMalloc(&m->d_jps_entries, need));
        m->entries_cap = need;
      }
      for (int level = 1; level <= 3; level++) {
        const long long threads = (long long)mv.total * (level == 1 ? 6 : (level == 2 ? 12 : 8));
        hipLaunchKernelGGL(fhp::jps_table_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, m->stream, mv, m->d_jps_tables,
                           m->d_jps_entries, level);
      }
      FM_HIP(hipGetLastError());
      m->entries_valid = true;
    }
    pa.jps_entries = m->d_jps_entries;
  }
  pa.order = nullptr;
  if (n > m->waves && m->sched_launch_order) {  // more queries than wavefronts: far-apart pairs first
    const size_t need = sizeof(int) * ((size_t)n + 128);
    if (need > m->order_cap) {
      FM_HIP(hipStreamSynchronize(m->stream));
      if (m->d_order) FM_HIP(hipFree(m->d_order));
      m->d_order = nullptr; m->order_cap = 0;
      FM_HIP(hipMalloc(&m->d_order, need));
      m->order_cap = need;
    }
    FM_HIP(hipMemsetAsync(m->d_order, 0, sizeof(int) * 128, m->stream));
    const double diag = m->res * std::sqrt((double)m->nx * m->nx + (double)m->ny * m->ny + (double)m->nz * m->nz);
    const double scale = 64.0 / diag;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(fhp::plan_order_hist_kernel, dim3(blocks), dim3(256), 0, m->stream, d_starts, d_goals, n, scale, m->d_order);
    hipLaunchKernelGGL(fhp::plan_order_scatter_kernel, dim3(blocks), dim3(256), 0, m->stream, d_starts, d_goals, n, scale, m->d_order, m->d_order + 128);
    FM_HIP(hipGetLastError());
    pa.order = m->d_order + 128;
  }
  const int grid = std::min(m->waves, n);
  if (m->search_mode == 1 && pa.hslots > 0) hipLaunchKernelGGL((fhp::plan_kernel<true, true>), dim3((unsigned)grid), dim3(64), 0, m->stream, mv, pa);
  else if (m->search_mode == 1) hipLaunchKernelGGL(fhp::plan_kernel<true>, dim3((unsigned)grid), dim3(64), 0, m->stream, mv, pa);
  else hipLaunchKernelGGL(fhp::plan_kernel<false>, dim3((unsigned)grid), dim3(64), 0, m->stream, mv, pa);
  FM_HIP(hipGetLastError());
  return FH_OK;
}

int fh_map_plan_batch(fh_map* m, const double* starts, const double* goals, int n, int max_points, double max_vertex_dist, int max_poly,
                      double* paths, int32_t* n_points, int64_t* expansions) {
  if (!m || !m->have_map || n < 0 || max_points < 2 || (n > 0 && (!starts || !goals || !paths || !n_points))) return FH_ERR_ARG;
  if (n == 0) return FH_OK;
  MapDeviceScope scope(m);
  int rc;
  const size_t bq = sizeof(double) * 3 * (size_t)n, bp = sizeof(double) * 3 * (size_t)n * max_points;
  if ((rc = stage(m, 1, 2 * bq)) != FH_OK) return rc;
  if ((rc = stage(m, 2, bp)) != FH_OK) return rc;
  if ((rc = stage(m, 3, 4 * (size_t)n)) != FH_OK) return rc;
  if ((rc = stage(m, 4, 8 * (size_t)n)) != FH_OK) return rc;
  double* d_q = (double*)m->d_stage[1];
  FM_HIP(hipMemcpyAsync(d_q, starts, bq, hipMemcpyHostToDevice, m->stream));
  FM_HIP(hipMemcpyAsync(d_q + 3 * (size_t)n, goals, bq, hipMemcpyHostToDevice, m->stream));
  if ((rc = fh_map_plan_batch_device(m, d_q, d_q + 3 * (size_t)n, n, max_points, max_vertex_dist, max_poly, (double*)m->d_stage[2],
                                     (int32_t*)m->d_stage[3], (int64_t*)m->d_stage[4])) != FH_OK)
    return rc;
  FM_HIP(hipMemcpyAsync(paths, m->d_stage[2], bp, hipMemcpyDeviceToHost, m->stream));
  FM_HIP(hipMemcpyAsync(n_points, m->d_stage[3], 4 * (size_t)n, hipMemcpyDeviceToHost, m->stream));
  if (expansions) FM_HIP(hipMemcpyAsync(expansions, m->d_stage[4], 8 * (size_t)n, hipMemcpyDeviceToHost, m->stream));
  FM_HIP(hipStreamSynchronize(m->stream));
  // Records chosen by the size of the map (fh_map_set_records -1) may be a hashed table, and a query that reaches more cells than the
  // table holds — an unreachable goal on a big map — ends at that limit (-2) where per-cell records would have answered "no path" or
  // found one.  This entry point is synchronous anyway: such queries are run again with one record per cell (fewer wavefronts; the
  // workspace is rebuilt, which costs far more than the queries — it is the rare case).  The device-pointer entry point reports -2.
  if (m->record_slots < 0 && m->search_mode == 1 && m->ws_slots > 0) {
    std::vector<int> again;
    for (int i = 0; i < n; i++)
      if (n_points[i] == -2) again.push_back(i);
    if (!again.empty()) {
      const int k = (int)again.size();
      std::vector<double> q((size_t)6 * k);
      for (int j = 0; j < k; j++)
        for (int c = 0; c < 3; c++) {
          q[(size_t)3 * j + c] = starts[(size_t)3 * again[j] + c];
          q[(size_t)3 * (k + j) + c] = goals[(size_t)3 * again[j] + c];
        }
      FM_HIP(hipStreamSynchronize(m->stream));  // (the main workspace is parked while nothing of this map is running)
      {
        // per-cell records on the parked workspace (kept from call to call: a handful of wavefronts, sized once), every field restored on
        // every way out — an early return must not leave record_slots at 0 (the next call would rebuild the main workspace twice)
        DenseRetryScope retry(m, std::max(64, std::min(k, 4 * m->n_cu)));
        FM_HIP(hipMemcpyAsync(d_q, q.data(), sizeof(double) * 6 * (size_t)k, hipMemcpyHostToDevice, m->stream));
        rc = fh_map_plan_batch_device(m, d_q, d_q + 3 * (size_t)k, k, max_points, max_vertex_dist, max_poly, (double*)m->d_stage[2],
                                      (int32_t*)m->d_stage[3], (int64_t*)m->d_stage[4]);
        if (rc == FH_OK) FM_HIP(hipStreamSynchronize(m->stream));
      }
      if (rc != FH_OK) return rc;
      std::vector<double> pp((size_t)3 * k * max_points);
      std::vector<int32_t> np((size_t)k);
      std::vector<int64_t> ex((size_t)k);
      FM_HIP(hipMemcpyAsync(pp.data(), m->d_stage[2], sizeof(double) * pp.size(), hipMemcpyDeviceToHost, m->stream));
      FM_HIP(hipMemcpyAsync(np.data(), m->d_stage[3], 4 * (size_t)k, hipMemcpyDeviceToHost, m->stream));
      FM_HIP(hipMemcpyAsync(ex.data(), m->d_stage[4], 8 * (size_t)k, hipMemcpyDeviceToHost, m->stream));
      FM_HIP(hipStreamSynchronize(m->stream));
      for (int j = 0; j < k; j++) {
        const int i = again[j];
        std::memcpy(paths + (size_t)3 * i * max_points, pp.data() + (size_t)3 * j * max_points, sizeof(double) * 3 * (size_t)max_points);
        n_points[i] = np[j];
        if (expansions) expansions[i] = ex[j];
      }
    }
  }
  return FH_OK;
}

}  // extern "C"
