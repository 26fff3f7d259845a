// fh_path.hip.hpp — voxel-grid path search on the device (gfx950): the first half of the corridor front-end.
//
// Reference: JPS_Manager::updateJPSMap -> MapUtil::readMap (/root/reference/faster/src/jps_manager.cpp:129-139,
// thirdparty/jps3d/include/jps_collision/map_util.h:30-185) builds an occupancy grid from a point cloud; JPS_Manager::solveJPS3D (jps_manager.cpp:141-200)
// frees the cells around start and goal, searches the 26-connected grid (jps3d graph_search.cpp: Euclidean step costs; heuristic: the
// exact empty-grid distance, which dominates jps3d's Euclidean one and stays consistent)
// and cleans the cell path up (jps_planner.cpp:36-105, :286-291: removeLinePts, removeCornerPts forwards and backwards).
// The CPU restatement this kernel is checked against cell for cell is faster_amd/host/corridor_frontend.{hpp,cpp} (plan_path).
//
// One query per 64-lane wavefront, persistent wavefronts pulling queries from a counter.  The search is A*; what a GPU needs is a
// priority queue without a serial heap:
//   * the open list is 256 unsorted sub-lists (a hash of the cell index picks the sub-list) of 64-entry chunks in HBM; the minimum
//     of every sub-list, its head chunk and fill count are a 16-byte record in LDS, so the next cell to expand is a lexicographic reduction of (key = f quantised to 2^-20
//     cells, squared distance to the goal, cell index) over the lanes on the DPP network — a strict total order, so the expansion
//     order does not depend on the container (the host restatement uses std::priority_queue with the same order and produces the
//     same paths);
//   * only the sub-list that held the minimum is scanned (one coalesced load per 64 entries) to remove the entry — the last entry
//     of the head chunk takes its place — and to find the next minimum of that sub-list;
//   * the 26 neighbours are relaxed one per lane; their state loads are issued before the scan (they depend on the popped index
//     only); relaxed neighbours are appended to their sub-lists by wave-uniform code;
//   * per-cell state (g, parent, open/closed) lives in a per-wavefront array in HBM that is never cleared: entries carry the serial
//     number of the query that wrote them.
// Chunk links, the free stack and the sub-list records are in LDS (12.5 KB).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)   // cell indices must equal the host's: no fused multiply-adds in the coordinate arithmetic

#include "fh_udiv.hpp"

namespace fhp {

constexpr int NCHUNK = 2048;    // chunks of 64 open-list entries per wavefront (131072 entries)
constexpr int CHUNK_WORDS = 192;  // 64 keys, 64 tie-breakers, 64 cells
constexpr int NS_LOG2 = 8, NS = 1 << NS_LOG2;  // sub-lists of the open list
constexpr int MAXRAW = 4096;    // longest raw cell path (the clean-up lists live in the chunk pool, which is dead by then)
constexpr double KEY_SCALE = 1048576.0;

struct MapView {
  int nx, ny, nz, total, m_free;
  unsigned inv_nxy, inv_nx;  // floor(2^32 / (nx ny)), floor(2^32 / nx) (0xffffffff for a divisor of 1): see fhu::div
  double res, ox, oy, oz;
  const unsigned* bits;  // occupancy, one bit per cell
};

struct CellState {  // 16 B
  double g;
  int parent;
  unsigned stamp;  // 2 * serial + closed
};

struct PlanArgs {
  const double* starts;   // [n][3]
  const double* goals;    // [n][3]
  int n;
  int max_points;
  double* paths;          // [n][max_points][3]
  int* n_points;          // [n]: vertices, 0 no path, -1 more than max_points, -2 a search limit was hit
  long long* expansions;  // [n] or null
  CellState* cells;       // [waves][total]; hashed records (hslots > 0): [waves][hslots]
  unsigned long long* hkeys;  // hashed records only: [waves][hslots] serial << 32 | cell (a key of another serial = a free slot)
  int hslots;             // 0: one record per cell of the map; a power of two: that many hashed records per wavefront
  long long chunk_words;  // words of `chunks` per wavefront (NCHUNK * CHUNK_WORDS; hashed records: what 3/4 hslots heap entries need)
  unsigned* chunks;       // [waves][NCHUNK][192]
  unsigned* serials;      // [waves]
  int* ticket;
  const int* order;       // ticket t works on query order[t] (null: t): far-apart pairs first, see plan_order_*_kernel
  // corridor post-processing (Faster::createMoreVertexes, faster.cpp:80-97; deleteVertexes, utils.cpp:1117-1124); 0 = off
  double max_vertex_dist;
  int max_poly;
  double sphere_ra;       // > 0: the path is clipped to JPS_in first (Faster::replan, faster.cpp:370-382), see clip in plan_kernel
  const unsigned char* jps_tables;  // jump point search only: the neighbour tables, see Planner::init_jps
  const short* jps_entries;         // ... and the jump tables of this map [total][32], see jps_table_kernel
  int profile_slot;                 // FHP_PROFILE builds only (scripts/jps_phase_profile.py): which phase's cycles `expansions` receives
};

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
// (cell indices are split with fhu::div, fh_udiv.hpp: on a wave-uniform index three scalar instructions instead of 45 vector ones)
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ int rank_in(unsigned long long m) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_min_step(int v) {
  const int o = __builtin_amdgcn_update_dpp(0x7fffffff, v, CTRL, ROW_MASK, 0xf, false);
  return o < v ? o : v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
  v = dpp_min_step<0x111, 0xf>(v);
  v = dpp_min_step<0x112, 0xf>(v);
  v = dpp_min_step<0x114, 0xf>(v);
  v = dpp_min_step<0x118, 0xf>(v);
  v = dpp_min_step<0x142, 0xa>(v);
  v = dpp_min_step<0x143, 0xc>(v);
  return __builtin_amdgcn_readlane(v, 63);
}

// getIntersectionWithSphere (faster/src/utils.cpp:713-776) with its arithmetic: single precision, except that pow(float, 2) is a
// double (the squares are summed in double, rounded once) and `- r * r` is a double subtraction.  The same expressions as
// fhfront::sphere_crossing (host/corridor_frontend.hpp).
__device__ inline void sphere_crossing(const double a_in[3], const double b_in[3], double r, const double c[3], double out[3]) {
  auto solve = [&](const double* A, const double* B, float& disc) {
    const float x1 = (float)A[0], y1 = (float)A[1], z1 = (float)A[2], x2 = (float)B[0], y2 = (float)B[1], z2 = (float)B[2];
    const float x3 = (float)c[0], y3 = (float)c[1], z3 = (float)c[2];
    const float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
    const float a = (float)((double)dx * (double)dx + (double)dy * (double)dy + (double)dz * (double)dz);
    const float b = 2.0f * (dx * (x1 - x3) + dy * (y1 - y3) + dz * (z1 - z3));
    const float cf = x3 * x3 + y3 * y3 + z3 * z3 + x1 * x1 + y1 * y1 + z1 * z1 - 2.0f * (x3 * x1 + y3 * y1 + z3 * z1);
    const float cc = (float)((double)cf - r * r);
    disc = b * b - 4.0f * a * cc;
    const float t = (-b + sqrtf(disc)) / (2.0f * a);
    out[0] = (double)(x1 + dx * t); out[1] = (double)(y1 + dy * t); out[2] = (double)(z1 + dz * t);
  };
  float disc;
  solve(a_in, b_in, disc);
  if (disc <= 0) solve(c, a_in, disc);
}

// FHP_PROFILE builds: the cycles a wavefront spends in one phase of the jump point search (PlanArgs::profile_slot) are written to
// `expansions` instead of the popped nodes.  0 pop + candidate addresses, 1 heap sift-down, 2 candidates settled from the tables,
// 3 cell-by-cell jumps, 4 move costs + wait for the cell records, 5 relaxation, 6 heap pushes, 7 the whole query; 8 / 9 the
// sift-down while the heap fits LDS / is deeper, 10 popped nodes with a deeper heap, 11 sum of the heap sizes.
// (diagnostic builds tell the limits apart: n_points = -20 - k)
#ifdef FHP_DEBUG_CODES
#define FHP_LIMIT(k) (-20 - (k))
#else
#define FHP_LIMIT(k) (-2)
#endif
#ifdef FHP_PROFILE
#define FHP_T(k) do { const long long t__ = (long long)__builtin_readcyclecounter(); if (prof_slot == (k)) prof_acc += t__ - prof_t; prof_t = t__; } while (0)
#else
#define FHP_T(k) do {} while (0)
#endif

struct Planner {
  int prof_slot = -1;
  long long prof_acc = 0, prof_t = 0;
  const MapView& mv;
  int lane;
  int s[3], t[3];  // start / goal cells (uniform)
  bool boxes_matter = true;
  // LDS
  short* cnext;  // [NCHUNK] next chunk of a bucket
  short* fstack; // [NCHUNK] free chunks (stack)
  int* r_key;    // [NS] per sub-list: the minimum (key, tie-breaker, cell) ...
  int* r_h2;
  int* r_id;
  int* r_hc;     // [NS] ... head chunk << 8 | entries in the head chunk (1..64); 0: empty
  short* claim;  // [NS] which lane inserts into a sub-list in this round
  int ftop;
  // HBM (the wavefront's chunk pool, once the search is over)
  int* raw;      // [MAXRAW]
  int* va;       // [MAXRAW]
  int* vb;       // [MAXRAW]

  __device__ Planner(const MapView& m, char* lds) : mv(m) {
    lane = lane_id();
    cnext = (short*)lds;
    fstack = cnext + NCHUNK;
    r_key = (int*)(fstack + NCHUNK);
    r_h2 = r_key + NS;
    r_id = r_h2 + NS;
    r_hc = r_id + NS;
    claim = (short*)(r_hc + NS);
    raw = va = vb = nullptr;
  }
  // stores of some lanes are read back by others: the clean-up lists are tiny, so simply wait for the stores
  __device__ __forceinline__ static void settle() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

  __device__ __forceinline__ bool outside(int x, int y, int z) const {
    return x < 0 || y < 0 || z < 0 || x >= mv.nx || y >= mv.ny || z >= mv.nz;
  }
  __device__ __forceinline__ int index(int x, int y, int z) const { return x + mv.nx * y + mv.nx * mv.ny * z; }
  __device__ __forceinline__ bool freed(int x, int y, int z) const {  // setFreeVoxelAndSurroundings around start and goal
    if (!boxes_matter) return false;  // (jump point search: no occupied cell inside the two cubes, so freeing them changes nothing)
    const int m = mv.m_free;
    const bool a = x >= s[0] - m && x <= s[0] + m && y >= s[1] - m && y <= s[1] + m && z >= s[2] - m && z <= s[2] + m;
    const bool b = x >= t[0] - m && x <= t[0] + m && y >= t[1] - m && y <= t[1] + m && z >= t[2] - m && z <= t[2] + m;
    return a || b;
  }
  __device__ __forceinline__ bool occupied_in(int x, int y, int z) const {  // inside the grid is the caller's business
    if (freed(x, y, z)) return false;
    const int id = index(x, y, z);
    return (mv.bits[id >> 5] >> (id & 31)) & 1u;
  }
  __device__ __forceinline__ bool is_free(int x, int y, int z) const { return !outside(x, y, z) && !occupied_in(x, y, z); }
  __device__ __forceinline__ void to_cell(double px, double py, double pz, int c[3]) const {  // MapUtil::floatToInt
    c[0] = (int)round((px - mv.ox) / mv.res - 0.5);
    c[1] = (int)round((py - mv.oy) / mv.res - 0.5);
    c[2] = (int)round((pz - mv.oz) / mv.res - 0.5);
  }
  __device__ __forceinline__ void decode(int id, int& x, int& y, int& z) const {
    const int nxy = mv.nx * mv.ny;
    z = fhu::div(id, nxy, mv.inv_nxy);
    const int rem = id - z * nxy;
    y = fhu::div(rem, mv.nx, mv.inv_nx);
    x = rem - y * mv.nx;
  }
  __device__ __forceinline__ void center(int id, double c[3]) const {  // MapUtil::intToFloat
    int x, y, z;
    decode(id, x, y, z);
    c[0] = (x + 0.5) * mv.res + mv.ox;
    c[1] = (y + 0.5) * mv.res + mv.oy;
    c[2] = (z + 0.5) * mv.res + mv.oz;
  }

  __device__ __forceinline__ static int sub_of(int id) { return (int)(((unsigned)id * 2654435761u) >> (32 - NS_LOG2)); }
  __device__ __forceinline__ int dist2(int x, int y, int z) const {  // tie-breaker: squared straight-line distance to the goal
    return (x - t[0]) * (x - t[0]) + (y - t[1]) * (y - t[1]) + (z - t[2]) * (z - t[2]);
  }
  // the exact length of a shortest 26-connected path to the goal in an empty grid (consistent; the host restatement's heuristic,
  // corridor_frontend.cpp: the same operations in the same order)
  __device__ __forceinline__ double heur(int x, int y, int z) const {
    int a = abs(x - t[0]), b = abs(y - t[1]), c = abs(z - t[2]);
    int tmp;
    if (a < b) { tmp = a; a = b; b = tmp; }
    if (b < c) { tmp = b; b = c; c = tmp; }
    if (a < b) { tmp = a; a = b; b = tmp; }
    return (double)c * sqrt(3.0) + (double)(b - c) * sqrt(2.0) + (double)(a - b);
  }

  // ray test of removeCornerPts (MapUtil::isBlocked-style sampling every 0.8 cell): uniform result
  __device__ __forceinline__ bool blocked(const double a[3], const double b[3]) const {
    const double dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
    const double mx = fmax(fabs(dx), fmax(fabs(dy), fabs(dz))) / mv.res;
    const int steps = (int)(mx / 0.8);
    if (steps <= 0) return false;
    const double sc = 1.0 / steps;
    for (int n0 = 1; n0 < steps; n0 += 64) {
      const int n = n0 + lane;
      bool out = false, occ = false;
      if (n < steps) {
        int c[3];  // rayTrace: pt = pt1 + (diff * s) * n, in this order (the rounding decides the cell when a sample sits on a corner)
        to_cell(a[0] + (dx * sc) * (double)n, a[1] + (dy * sc) * (double)n, a[2] + (dz * sc) * (double)n, c);
        out = outside(c[0], c[1], c[2]);
        if (!out) occ = occupied_in(c[0], c[1], c[2]);
      }
      const unsigned long long mo = __ballot(out);
      unsigned long long mc = __ballot(occ);
      if (mo) {
        mc &= (mo & (~mo + 1ull)) - 1ull;  // the walk stops at the first sample outside the grid
        return mc != 0ull;
      }
      if (mc) return true;
    }
    return false;
  }
  __device__ __forceinline__ static double dist(const double a[3], const double b[3]) {
    const double x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2];
    return sqrt(x * x + y * y + z * z);
  }

  // jps_planner.cpp:36-81 on a list of cells (every vertex of the clean-up is a cell centre); in -> out, returns the count
  __device__ __forceinline__ int remove_corner_points(const int* in, int n, int* out) const {
    if (n < 2) {
      if (n == 1 && lane == 0) out[0] = in[0];
      return n;
    }
    double prev[3], a[3], b[3];
    center(rfl(in[0]), prev);
    center(rfl(in[1]), b);
    if (lane == 0) out[0] = in[0];
    int no = 1;
    double c1 = blocked(prev, b) ? INFINITY : dist(prev, b);
    for (int i = 1; i + 1 < n; i++) {
      const int ia = rfl(in[i]);
      center(ia, a);
      center(rfl(in[i + 1]), b);
      const double dab = dist(a, b);
      const double c2 = blocked(a, b) ? INFINITY : dab;
      const double c3 = blocked(prev, b) ? INFINITY : dist(prev, b);
      if (c3 < c1 + c2) c1 = c3;
      else {
        if (lane == 0) out[no] = ia;
        no++;
        c1 = dab;
        prev[0] = a[0]; prev[1] = a[1]; prev[2] = a[2];
      }
    }
    if (lane == 0) out[no] = in[n - 1];
    no++;
    return no;
  }

  // One query.  Returns the number of cells of the cleaned path in va[] (start -> goal), 0 = no path, -2 = limit.
  //
  // Open list: 256 unsorted SUB-LISTS (a hash of the cell index picks the sub-list); LDS holds the minimum (key, tie-breaker,
  // cell) of each, its head chunk and fill count.  A pop is
  //   1. a lexicographic reduction of the minima (4 per lane from LDS, then the DPP network; no HBM access) -> the cell to expand; the loads of the expansion (states of
  //      the cell and of its 26 neighbours, occupancy words) depend on that index only and are issued at once;
  //   2. a scan of THAT sub-list (1/256 of the open list; one coalesced load per 64 entries) to remove the entry and find the
  //      sub-list's next minimum, while the loads of step 1 are in flight;
  //   3. the relaxed neighbours are inserted by all lanes at once (entry appended to the head chunk of its sub-list, the record
  //      updated; two neighbours that hash to one sub-list take turns).
  // The expansion order is that of the strict total order (key, tie-breaker, cell) whatever the container: the same cells in the
  // same order as the host's std::priority_queue.
  __device__ __forceinline__ int search(const PlanArgs& pa, CellState* cells, unsigned* chunks, unsigned serial, long long& expansions) {
    const unsigned st_open = serial * 2u, st_closed = serial * 2u + 1u;
    for (int i = lane; i < NCHUNK; i += 64) fstack[i] = (short)(NCHUNK - 1 - i);  // chunk 0 on top
    ftop = NCHUNK;
    const int sid = index(s[0], s[1], s[2]), tid = index(t[0], t[1], t[2]);
    int limit = 0;
    // the sub-list records: all empty
    for (int i = lane; i < NS; i += 64) { r_key[i] = 0x7fffffff; r_h2[i] = 0x7fffffff; r_id[i] = 0x7fffffff; r_hc[i] = 0; }
    // the start cell
    {
      const double f = 0.0 + heur(s[0], s[1], s[2]);
      if (f >= 2040.0) return -2;
      const int key = (int)(f * KEY_SCALE), h2 = dist2(s[0], s[1], s[2]);
      const int c = rfl((int)fstack[--ftop]);
      if (lane == 0) {
        unsigned* e = chunks + (size_t)c * CHUNK_WORDS;
        e[0] = (unsigned)key; e[64] = (unsigned)h2; e[128] = (unsigned)sid;
        CellState cs; cs.g = 0.0; cs.parent = -1; cs.stamp = st_open;
        cells[sid] = cs;
      }
      const int ssub = sub_of(sid);
      r_key[ssub] = key; r_h2[ssub] = h2; r_id[ssub] = sid; r_hc[ssub] = (c << 8) | 1;
      cnext[c] = -1;
    }
    int n_open = 1;
    bool found = false;
    // neighbour of this lane
    const int k = lane + (lane >= 13 ? 1 : 0);
    const int dx = k / 9 - 1, dy = (k / 3) % 3 - 1, dz = k % 3 - 1;
    const double step = sqrt((double)(dx * dx + dy * dy + dz * dz));
    const int nxy = mv.nx * mv.ny;

    while (n_open > 0) {
      // ---- 1. the minimum of the sub-list minima: first over this lane's NS / 64 records, then over the lanes
      int lk = r_key[lane], lh = r_h2[lane], lid = r_id[lane], ls = lane;
#pragma unroll
      for (int q = 1; q < NS / 64; q++) {
        const int sq = lane + 64 * q;
        const int qk = r_key[sq], qh = r_h2[sq], qi = r_id[sq];
        if (qk < lk || (qk == lk && (qh < lh || (qh == lh && qi < lid)))) { lk = qk; lh = qh; lid = qi; ls = sq; }
      }
      const int mf = wave_min_i32(lk);
      bool cand = lk == mf;
      const int mh = wave_min_i32(cand ? lh : 0x7fffffff);
      cand = cand && lh == mh;
      const int id = wave_min_i32(cand ? lid : 0x7fffffff);
      cand = cand && lid == id;
      const int wsub = __builtin_amdgcn_readlane(ls, (int)__builtin_ctzll(__ballot(cand)));  // the sub-list that holds it
      const int whc = rfl(r_hc[wsub]);
      const int hc = whc >> 8, cnt = whc & 255;
      // the loads of the expansion depend on `id` only: issue them now
      int cx, cy, cz;
      {
        cz = id / nxy;
        const int rem = id - cz * nxy;
        cy = rem / mv.nx;
        cx = rem - cy * mv.nx;
      }
      const int x = cx + dx, y = cy + dy, z = cz + dz;
      const bool inside = lane < 26 && !outside(x, y, z);
      const int nid = inside ? index(x, y, z) : id;
      // Entries and cell states written by one lane in the previous iteration are read by other lanes now: the stores must have
      // completed (loads and stores of a wavefront are not ordered with respect to each other across lanes; without this wait
      // one query in a thousand read an entry before it had landed)
      settle();
      const CellState cs = cells[id];
      const CellState ns = cells[nid];
      const unsigned occw = mv.bits[nid >> 5];
      // ---- 2. the sub-list: find the entry, the minimum of the others
      int bf = 0x7fffffff, bh = 0x7fffffff, bi = 0x7fffffff;  // this lane's best among the entries that stay
      int hf = 0, hh = 0, hid = 0;                            // the head chunk's entries
      int tslot = -1, matches = 0;
      {
        int cc = hc, ccnt = cnt;
        bool first = true;
        while (cc >= 0) {
          const unsigned* e = chunks + (size_t)cc * CHUNK_WORDS;
          int f = 0x7fffffff, h = 0x7fffffff, eid = 0x7fffffff;
          if (lane < ccnt) { f = (int)e[lane]; h = (int)e[64 + lane]; eid = (int)e[128 + lane]; }
          if (first) { hf = f; hh = h; hid = eid; first = false; }
          if (f == mf && h == mh && eid == id) {  // the entry (a duplicate of it stays: counted)
            if (tslot < 0) tslot = cc * 64 + lane;
            matches++;
          } else if (f < bf || (f == bf && (h < bh || (h == bh && eid < bi)))) { bf = f; bh = h; bi = eid; }
          cc = rfl((int)cnext[cc]);
          ccnt = 64;
        }
      }
      const unsigned long long tm = __ballot(tslot >= 0);
      if (!tm) return -2;  // (cannot happen: the minimum of a sub-list is one of its entries)
      const int wslot = __builtin_amdgcn_readlane(tslot, (int)__builtin_ctzll(tm));
      const bool copy_stays = __popcll(tm) > 1 || __ballot(matches > 1) != 0ull;
      int nk, nh, ni;  // the sub-list's minimum once the entry is gone
      if (copy_stays) { nk = mf; nh = mh; ni = id; }
      else {
        nk = wave_min_i32(bf);
        bool c2 = bf == nk;
        nh = wave_min_i32(c2 ? bh : 0x7fffffff);
        c2 = c2 && bh == nh;
        ni = wave_min_i32(c2 ? bi : 0x7fffffff);
      }
      // remove: the last entry of the head chunk takes the place
      const int last = cnt - 1;
      if (wslot != hc * 64 + last) {
        const int lf = __builtin_amdgcn_readlane(hf, last), lh2 = __builtin_amdgcn_readlane(hh, last), li = __builtin_amdgcn_readlane(hid, last);
        if (lane == 0) {
          unsigned* e = chunks + (size_t)(wslot >> 6) * CHUNK_WORDS + (wslot & 63);
          e[0] = (unsigned)lf; e[64] = (unsigned)lh2; e[128] = (unsigned)li;
        }
      }
      {
        int nhc = (hc << 8) | last;
        if (last == 0) {
          const int nhead = rfl((int)cnext[hc]);
          nhc = nhead >= 0 ? ((nhead << 8) | 64) : 0;
          fstack[ftop++] = (short)hc;
        }
        r_key[wsub] = nk; r_h2[wsub] = nh; r_id[wsub] = ni; r_hc[wsub] = nhc;
      }
      n_open--;
      // ---- the cell
      if (cs.stamp == st_closed) continue;  // a stale duplicate: the cell was expanded from a better entry
      if (lane == 0) cells[id].stamp = st_closed;
      if (id == tid) { found = true; break; }
      expansions++;
      bool ok = inside && (freed(x, y, z) || !((occw >> (nid & 31)) & 1u));
      int key = 0, h2 = 0;
      if (ok) {
        const bool visited = (ns.stamp >> 1) == serial;
        const double ng = cs.g + step;
        if (visited && ((ns.stamp & 1u) || !(ng < ns.g))) ok = false;
        else {
          CellState w; w.g = ng; w.parent = id; w.stamp = st_open;
          cells[nid] = w;
          h2 = dist2(x, y, z);
          const double f = ng + heur(x, y, z);
          if (f >= 2040.0) { limit = 1; ok = false; }
          else key = (int)(f * KEY_SCALE);
        }
      }
      // ---- 3. insert the relaxed neighbours, all lanes at once: a lane claims its sub-list (two neighbours of one expansion rarely
      //         share one: the loser goes in the next round), appends its entry to the head chunk and updates the record
      const int csub = sub_of(nid);
      bool pending = ok;
      while (__ballot(pending)) {
        if (pending) claim[csub] = (short)lane;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the read below must come from LDS: another lane may have overwritten the claim)
        const bool win = pending && claim[csub] == (short)lane;
        const int rhc = win ? r_hc[csub] : 0;
        int c0 = rhc ? (rhc >> 8) : -1, cn = rhc & 255;
        const bool need = win && (c0 < 0 || cn == 64);
        const unsigned long long nm = __ballot(need);
        const int total_new = (int)__popcll(nm);
        if (total_new > ftop) { limit = 1; break; }
        if (need) {
          const int nc = (int)fstack[ftop - 1 - rank_in(nm)];
          cnext[nc] = (short)c0;
          c0 = nc; cn = 0;
        }
        ftop -= total_new;
        if (win) {
          unsigned* e = chunks + (size_t)c0 * CHUNK_WORDS + cn;
          e[0] = (unsigned)key; e[64] = (unsigned)h2; e[128] = (unsigned)nid;
          const int rk = r_key[csub], rh = r_h2[csub], ri = r_id[csub];
          if (key < rk || (key == rk && (h2 < rh || (h2 == rh && nid < ri)))) { r_key[csub] = key; r_h2[csub] = h2; r_id[csub] = nid; }
          r_hc[csub] = (c0 << 8) | (cn + 1);
        }
        n_open += (int)__popcll(__ballot(win));
        pending = pending && !win;
      }
      if (__ballot(limit != 0)) return -2;
    }
    if (!found) return 0;
    return finish_path<false>(cells, chunks, sid, tid);
  }

  // ---- hashed cell records (jump point search with PlanArgs::hslots > 0): the records of the cells a query has reached, instead of
  // one record per cell of the map and wavefront.  Open addressing with linear probing over hmask + 1 slots of this wavefront;
  // a slot belongs to the running query when the upper half of its key is the query's serial number, so nothing is cleared
  // between queries.  Nothing is ever removed during a query: a cell's record is the first slot of its chain that carries its
  // key, and the chain ends at the first slot that is not the query's.
  unsigned long long* hk = nullptr;
  CellState* hr = nullptr;
  unsigned hmask = 0u, hshift = 0u;
  int cap_g = CAP_G;               // heap entries in HBM (hashed records: 3/4 of the slots — a cell has one heap entry at most)
  unsigned long long hser = 0ull;  // serial << 32
  int hcount = 0;                  // records of the running query
  __device__ __forceinline__ unsigned home(int id) const { return ((unsigned)id * 2654435761u) >> hshift; }
  __device__ __forceinline__ static bool other_query(unsigned long long k, unsigned long long ser) { return (k & 0xffffffff00000000ull) != ser; }
  // per lane: the record of cell id (stamp 0 when it has none), and where it is / where it would go
  __device__ __forceinline__ CellState hfind(bool want, int id, unsigned& slot, bool& present) const {
    CellState r; r.g = 0.0; r.parent = -1; r.stamp = 0u;
    const unsigned long long key = hser | (unsigned long long)(unsigned)id;
    slot = home(id);
    present = false;
    bool pend = want;
    while (__ballot(pend)) {
      if (pend) {
        const unsigned long long k = hk[slot];
        const CellState c = hr[slot];
        if (k == key) { r = c; present = true; pend = false; }
        else if (other_query(k, hser)) pend = false;
        else slot = (slot + 1u) & hmask;
      }
    }
    return r;
  }
  // the same for one cell, all lanes (uniform result)
  __device__ __forceinline__ CellState hfind_uniform(int id, unsigned& slot, bool& present) const {
    CellState r; r.g = 0.0; r.parent = -1; r.stamp = 0u;
    const unsigned long long key = hser | (unsigned long long)(unsigned)id;
    slot = home(id);
    present = false;
    for (;;) {
      const unsigned long long k = hk[slot];
      const CellState c = hr[slot];
      const unsigned klo = (unsigned)rfl((int)(unsigned)k), khi = (unsigned)rfl((int)(unsigned)(k >> 32));
      const unsigned long long ku = ((unsigned long long)khi << 32) | klo;
      if (ku == key) {
        r.g = __hiloint2double(rfl(__double2hiint(c.g)), rfl(__double2loint(c.g)));
        r.parent = rfl(c.parent);
        r.stamp = (unsigned)rfl((int)c.stamp);
        present = true;
        return r;
      }
      if (other_query(ku, hser)) return r;
      slot = (slot + 1u) & hmask;
    }
  }
  // per lane: write record w of cell id (distinct cells in distinct lanes) where hfind said.  New cells whose chains end in the same
  // free slot: the lowest lane takes it, the others walk on (they see what was written: the stores are waited for).
  // Returns false when the table is three quarters full.
  __device__ __forceinline__ bool hwrite(bool want, int id, unsigned slot, bool present, const CellState& w) {
    const unsigned long long key = hser | (unsigned long long)(unsigned)id;
    if (want && present) hr[slot] = w;
    bool need = want && !present;
    unsigned long long nm = __ballot(need);
    hcount += (int)__popcll(nm);
    if (hcount > (int)(((hmask + 1u) >> 2) * 3u)) return false;
    while (nm) {
      bool lose = false;
      if (__popcll(nm) > 1)
        for (unsigned long long m2 = nm; m2; m2 &= m2 - 1ull) {
          const int i = (int)__builtin_ctzll(m2);
          if (need && i < lane && (unsigned)__builtin_amdgcn_readlane((int)slot, i) == slot) lose = true;
        }
      if (need && !lose) {
        hk[slot] = key;
        hr[slot] = w;
        need = false;
      }
      nm = __ballot(need);
      if (nm) {
        settle();
        bool pend = need;
        if (need) slot = (slot + 1u) & hmask;
        while (__ballot(pend)) {
          if (pend) {
            if (other_query(hk[slot], hser)) pend = false;
            else slot = (slot + 1u) & hmask;
          }
        }
      }
    }
    return true;
  }
  // one cell, written by lane 0 (present / slot from hfind_uniform)
  __device__ __forceinline__ bool hwrite_uniform(int id, unsigned slot, bool present, const CellState& w) {
    if (!present) {
      hcount++;
      if (hcount > (int)(((hmask + 1u) >> 2) * 3u)) return false;
    }
    if (lane == 0) {
      if (!present) hk[slot] = hser | (unsigned long long)(unsigned)id;
      hr[slot] = w;
    }
    return true;
  }

  // The parent chain goal -> start, then the clean-up of jps_planner.cpp:283-291.  Returns the number of cells in va[] (start -> goal).
  template <bool HASHED>
  __device__ __forceinline__ int finish_path(const CellState* cells, unsigned* chunks, int sid, int tid) {
    // ---- raw cell path, goal -> start
    raw = (int*)chunks;
    va = raw + MAXRAW;
    vb = va + MAXRAW;
    int len = 0;
    for (int id = tid;;) {
      if (len >= MAXRAW) return FHP_LIMIT(4);
      if (lane == 0) raw[len] = id;
      len++;
      if (id == sid) break;
      if (HASHED) {
        unsigned slot;
        bool present;
        id = hfind_uniform(id, slot, present).parent;
      } else {
        id = rfl(cells[id].parent);
      }
      if (id < 0) break;
    }
    settle();
    // ---- removeLinePts (jps_planner.cpp:83-105) on the start -> goal order: raw[len-1-i]
    int na = 0;
    if (len < 3) {
      for (int i = lane; i < len; i += 64) va[i] = raw[len - 1 - i];
      na = len;
    } else {
      for (int i0 = 0; i0 < len; i0 += 64) {
        const int i = i0 + lane;
        bool keep = false;
        int idv = 0;
        if (i < len) {
          idv = raw[len - 1 - i];
          if (i == 0 || i == len - 1) keep = true;
          else {
            double p0[3], p1[3], p2[3];
            center(raw[len - i], p0);      // path[i-1]
            center(idv, p1);
            center(raw[len - 2 - i], p2);  // path[i+1]
            const double qx = (p2[0] - p1[0]) - (p1[0] - p0[0]), qy = (p2[1] - p1[1]) - (p1[1] - p0[1]), qz = (p2[2] - p1[2]) - (p1[2] - p0[2]);
            keep = fabs(qx) + fabs(qy) + fabs(qz) > 1e-2;
          }
        }
        const unsigned long long m = __ballot(keep);
        if (keep) va[na + rank_in(m)] = idv;
        na += (int)__popcll(m);
      }
    }
    settle();
    // ---- removeCornerPts forwards, then on the reversed path, then back (jps_planner.cpp:286-291)
    int nb = remove_corner_points(va, na, vb);
    settle();
    for (int i = lane; i < nb; i += 64) va[i] = vb[nb - 1 - i];
    settle();
    const int nc = remove_corner_points(va, nb, vb);
    settle();
    for (int i = lane; i < nc; i += 64) va[i] = vb[nc - 1 - i];
    settle();
    return nc;
  }

  // =================================================================================================================
  // Jump point search in jps3d's own order (thirdparty/jps3d/src/jps_planner/graph_search.cpp:123-470 — what
  // JPS_Manager::solveJPS3D runs, jps_manager.cpp:166): the pruning rules, the order in which successors are generated, the
  // tolerance comparator (graph_search.h:19-29) and the sift discipline of the binary heap select ONE of the optimal paths, and
  // FASTER's corridor is built around that one.  The CPU restatement is plan_path_jps (host/corridor_frontend.cpp), itself pinned
  // to the reference's compiled sources vertex for vertex; this search is checked against it bit for bit.
  //
  // The control flow of jps3d is serial (one heap, successors relaxed one after the other); what the 64 lanes share out is the
  // inside of a jump:
  //   * a straight jump tests 64 consecutive cells at once (free? goal? forced neighbour?): ballots give the first blocked cell
  //     and the first cell that ends the jump;
  //   * a plane-diagonal jump tests 32 diagonal cells at once, then the two straight jumps that leave every diagonal cell before
  //     the first stop run one per lane in lock step (a lane whose diagonal cell lies beyond one that has already succeeded
  //     retires);
  //   * a space-diagonal jump does the same for 8 diagonal cells x (3 straight + 3 plane-diagonal jumps), one jump per lane.
  // A jump only returns "some cell of this ray ends it" and the diagonal cell it happened at, so the order in which the lanes
  // find that out does not matter: the successor list and its order are jps3d's.
  // The heap is jps3d's binary heap, top 311 entries in LDS (the rest in the wavefront's chunk pool), moved by one scalar
  // program that all lanes execute; the position of an entry whose key decreases is found by a lane-parallel scan.
  // Cell state: g, parent, stamp = serial << 6 | direction id << 1 | closed.
  static constexpr int CAP_L = 311, CAP_G = 78000;  // (78000 x 20 B: what the wavefront's chunk pool, NCHUNK x CHUNK_WORDS words, holds; 311, an odd number: the children of a node are both in LDS or both in the chunk pool; 312: 7 644 B of LDS with the tables = 6 granules -> 20 workgroups per CU at 96 VGPRs;
                                                   //  432 and 16 per CU: 67 ms instead of 63 for 65536 forest queries; 248 and 24 per CU at 80 VGPRs, spilling: 65 ms)
  // (typed by address space: a select between an LDS and a chunk pool pointer cannot be formed, so an access is a ds_ or a global_
  // instruction, never a flat_ one through a pointer picked at run time)
  typedef __attribute__((address_space(3))) double lds_f64;
  typedef __attribute__((address_space(3))) int lds_i32;
  typedef __attribute__((address_space(1))) double pool_f64;
  typedef __attribute__((address_space(1))) int pool_i32;
  lds_f64* hf;           // LDS [CAP_L]
  lds_f64* hg;
  lds_i32* hid;
  const unsigned* jns;   // LDS: natural neighbours [27][28] bytes (26 used; 7 words per direction)
  const unsigned* jf1;   // cells to test [27][12] bytes (3 words per direction)
  const unsigned* jf2;   // directions to add [27][12] bytes
  pool_f64* gf;          // HBM overflow of the heap
  pool_f64* gg;
  pool_i32* gi;

  struct HE { int id; double f, g; };

  __device__ __forceinline__ void init_jps(char* lds, const unsigned char* tab, const short* entries) {
    hf = (lds_f64*)lds;
    hg = hf + CAP_L;
    hid = (lds_i32*)(hg + CAP_L);
    unsigned* w = (unsigned*)(lds + CAP_L * 20);
    const unsigned* tw = (const unsigned*)tab;
    for (int i = lane; i < JTAB_WORDS; i += 64) w[i] = tw[i];
    jt = entries;
    jns = w;
    jf1 = w + 27 * 7;
    jf2 = jf1 + 27 * 3;
  }
  static constexpr int JTAB_WORDS = 27 * 7 + 27 * 3 + 27 * 3;

  __device__ __forceinline__ static int ux(unsigned pk) { return (int)(pk & 3u) - 1; }
  __device__ __forceinline__ static int uy(unsigned pk) { return (int)((pk >> 2) & 3u) - 1; }
  __device__ __forceinline__ static int uz(unsigned pk) { return (int)((pk >> 4) & 3u) - 1; }
  __device__ __forceinline__ static int code_of(unsigned pk) { return (int)((pk & 3u) + 3u * ((pk >> 2) & 3u) + 9u * ((pk >> 4) & 3u)); }
  __device__ __forceinline__ static unsigned byte_of(const unsigned* w, int i) { return (w[i >> 2] >> (8 * (i & 3))) & 255u; }
  __device__ __forceinline__ unsigned nat(int code, int k) const { return byte_of(jns + code * 7, k); }

  // occupied = inside the grid, not freed, bit set (graph_search.cpp:63-66); the load is issued whatever the cell (clamped)
  __device__ __forceinline__ bool occ_any(int x, int y, int z) const {
    const bool in = !outside(x, y, z);
    const int id = in ? index(x, y, z) : 0;
    const unsigned w = mv.bits[id >> 5];
    return in && !freed(x, y, z) && ((w >> (id & 31)) & 1u);
  }
  __device__ __forceinline__ bool free_any(int x, int y, int z) const {
    const bool in = !outside(x, y, z);
    const int id = in ? index(x, y, z) : 0;
    const unsigned w = mv.bits[id >> 5];
    return in && (freed(x, y, z) || !((w >> (id & 31)) & 1u));
  }
  // does a jump in direction `code` end at this (free) cell: the goal, or a forced neighbour (graph_search.cpp:382-386, :417-470)
  __device__ __forceinline__ bool ends_at(int x, int y, int z, int code, int norm1) const {
    const unsigned w0 = jf1[code * 3], w1 = jf1[code * 3 + 1];
    bool any = x == t[0] && y == t[1] && z == t[2];
#pragma unroll
    for (int fn = 0; fn < 8; fn++) {
      const unsigned pk = ((fn < 4 ? w0 : w1) >> (8 * (fn & 3))) & 255u;
      const bool o = occ_any(x + ux(pk), y + uy(pk), z + uz(pk));
      any = any || (o && (fn < 6 || norm1 != 3));
    }
    return any;
  }
  __device__ __forceinline__ static int norm1_of(unsigned pk) { return abs(ux(pk)) + abs(uy(pk)) + abs(uz(pk)); }

  // ---- the heap: entry i in LDS below CAP_L, in HBM above
  // (an entry read by all lanes at once is the same in all of them: said to the compiler, what is computed from it — heap sizes,
  // positions, loop conditions — stays on the scalar unit)
  __device__ __forceinline__ static double ufl(double v) { return __hiloint2double(rfl(__double2hiint(v)), rfl(__double2loint(v))); }
  __device__ __forceinline__ HE hget(int i) const {
    HE e;
    if (i < CAP_L) { e.id = hid[i]; e.f = hf[i]; e.g = hg[i]; }
    else { const int j = i - CAP_L; e.id = gi[j]; e.f = gf[j]; e.g = gg[j]; }
    e.id = rfl(e.id); e.f = ufl(e.f); e.g = ufl(e.g);
    return e;
  }
  __device__ __forceinline__ void hset(int i, const HE& e) {
    if (i < CAP_L) {
      if (lane == 0) { hid[i] = e.id; hf[i] = e.f; hg[i] = e.g; }
    } else {
      const int j = i - CAP_L;
      if (lane == 0) { gi[j] = e.id; gf[j] = e.f; gg[j] = e.g; }  // (the caller waits for the stores when it is done)
    }
  }
  // compare_state (graph_search.h:19-29): a has LOWER priority than b.  Without branches: both comparisons are made, one is taken.
  __device__ __forceinline__ static bool lower_fg(double af, double ag, double bf, double bg) {
    const bool tie = (af >= bf - 0.000001) & (af <= bf + 0.000001);
    return (tie & (ag < bg)) | (!tie & (af > bf));
  }
  __device__ __forceinline__ static bool lower(const HE& a, const HE& b) { return lower_fg(a.f, a.g, b.f, b.g); }
  // m rises from position i.  The ancestors of a position are arithmetic: lane k holds the k-th ((i + 1 >> k) - 1), all of them are
  // fetched at once (LDS, or LDS and the chunk pool), compared with m at once — an ancestor of lower priority moves down, the first
  // one that is not ends the walk — and moved down by one lane each.  The comparisons of the serial walk, so the same heap.
  __device__ __forceinline__ void sift_up(int i, const HE& m) {
    const int d = 31 - __builtin_clz((unsigned)(i + 1));  // ancestors
    const bool on = lane >= 1 && lane <= d;
    const int pos = on ? ((i + 1) >> lane) - 1 : 0;
    if (i < CAP_L) {
      const double ef = hf[pos], eg = hg[pos];
      const int eid = hid[pos];
      const unsigned long long stopm = __ballot(on && !lower_fg(ef, eg, m.f, m.g));
      const int s = stopm ? (int)__builtin_ctzll(stopm) : d + 1;  // ancestors 1 .. s-1 move down, m lands where ancestor s-1 was
      if (on && lane < s) {
        const int dst = ((i + 1) >> (lane - 1)) - 1;
        hf[dst] = ef; hg[dst] = eg; hid[dst] = eid;
      }
      if (lane == 0) {
        const int dst = ((i + 1) >> (s - 1)) - 1;
        hf[dst] = m.f; hg[dst] = m.g; hid[dst] = m.id;
      }
      return;
    }
    // (both homes are read, each with a valid index, and the value is selected: one address space per access)
    const bool low = pos < CAP_L;
    const int pl = low ? pos : 0, pg = low ? 0 : pos - CAP_L;
    const double lf = hf[pl], lg = hg[pl], qf = gf[pg], qg = gg[pg];
    const int lid = hid[pl], qid = gi[pg];
    const double ef = low ? lf : qf, eg = low ? lg : qg;
    const int eid = low ? lid : qid;
    const unsigned long long stopm = __ballot(on && !lower_fg(ef, eg, m.f, m.g));
    const int s = stopm ? (int)__builtin_ctzll(stopm) : d + 1;
    const bool wr = (on && lane < s) || lane == 0;
    const int dst = ((i + 1) >> (lane == 0 ? s - 1 : lane - 1)) - 1;
    const double wf = lane == 0 ? m.f : ef, wg = lane == 0 ? m.g : eg;
    const int wid = lane == 0 ? m.id : eid;
    if (wr && dst < CAP_L) { hf[dst] = wf; hg[dst] = wg; hid[dst] = wid; }
    asm volatile("" ::: "memory");  // (keeps the two stores apart: merged, they become flat stores through a pointer table in scratch)
    if (wr && dst >= CAP_L) { const int j = dst - CAP_L; gf[j] = wf; gg[j] = wg; gi[j] = wid; }
    settle();
  }
  // m sinks from the root.  Which child a node prefers does not depend on m: while both children of a node sit in LDS (nodes below
  // LDS_INNER) every node compares its two children at once (one ballot per 64 nodes), the path of preferred children is walked on
  // the scalar unit, and the entries on it are compared with m and moved up by one lane each.  If m sinks past the end of that
  // path (a heap deeper than LDS holds) the serial walk takes over there.  The comparisons are the ones the serial walk makes
  // (child against child, then the preferred child against m), so the heap ends up in the same state.
  static_assert((size_t)CAP_G * 20 <= (size_t)NCHUNK * CHUNK_WORDS * 4, "the heap levels below LDS live in the chunk pool");
  static constexpr int LDS_INNER = (CAP_L - 1) / 2;  // nodes below have both children in LDS, the others both in the chunk pool
  static_assert(CAP_L % 2 == 1, "a node's children must not straddle LDS and the chunk pool");
  __device__ __forceinline__ void sift_down(const HE& m, int n) {
    auto prefer = [&](int base) -> unsigned long long {
      const int node = base + lane, first = 2 * node + 1;
      bool right = false;
      if (node < LDS_INNER && first + 1 < n) right = lower_fg(hf[first], hg[first], hf[first + 1], hg[first + 1]);
      return __ballot(right);
    };
    const int inner = min(n >> 1, LDS_INNER);
    const unsigned long long p0 = prefer(0);
    const unsigned long long p1 = inner > 64 ? prefer(64) : 0ull;
    const unsigned long long p2 = inner > 128 ? prefer(128) : 0ull;
    int idx = 0, depth = 0, pathv = 0;  // lane k of pathv: the k-th node of the path (lane 0: the root)
    while (idx < LDS_INNER) {
      const int first = 2 * idx + 1;
      if (first >= n) break;
      const unsigned long long w = idx < 64 ? p0 : (idx < 128 ? p1 : p2);
      idx = first + (int)((w >> (idx & 63)) & 1ull);
      depth++;
      pathv = lane == depth ? idx : pathv;
    }
    const bool on = lane >= 1 && lane <= depth;
    const int pos = on ? pathv : 0;
    const double ef = hf[pos], eg = hg[pos];
    const int eid = hid[pos];
    const unsigned long long sm = __ballot(on && lower_fg(ef, eg, m.f, m.g));
    const int s = sm ? (int)__builtin_ctzll(sm) : depth + 1;  // m lands on node s - 1 of the path, or sinks past its end
    if (on && lane < s) {
      const int up = (pos - 1) >> 1;
      hf[up] = ef; hg[up] = eg; hid[up] = eid;
    }
    int i = __builtin_amdgcn_readlane(pathv, s - 1);
    if (sm == 0ull && 2 * i + 1 < n) {  // the end of the LDS path has children: they are in the chunk pool
      for (;;) {
        const int first = 2 * i + 1;
        if (first >= n) break;
        const int j = first - CAP_L, j1 = first + 1 < n ? j + 1 : j;
        const double f0 = ufl(gf[j]), g0 = ufl(gg[j]), f1 = ufl(gf[j1]), g1 = ufl(gg[j1]);
        const int i0 = rfl(gi[j]), i1 = rfl(gi[j1]);
        const bool right = first + 1 < n && lower_fg(f0, g0, f1, g1);
        HE b;
        b.f = right ? f1 : f0; b.g = right ? g1 : g0; b.id = right ? i1 : i0;
        if (lower(b, m)) break;
        hset(i, b);
        i = first + (right ? 1 : 0);
      }
      hset(i, m);
      settle();
      return;
    }
    if (lane == 0) { hf[i] = m.f; hg[i] = m.g; hid[i] = m.id; }
  }
  // the position of cell `id` in the heap (an entry whose key decreases); the chunk pool part four loads at a time
  __device__ __forceinline__ int heap_find(int id, int n) const {
    const int nl = min(n, CAP_L);
    for (int b = 0; b < nl; b += 64) {
      const int i = b + lane;
      const int v = i < nl ? (hid[i] & IDMASK) : -1;
      const unsigned long long m = __ballot(v == id);
      if (m) return b + (int)__builtin_ctzll(m);
    }
    for (int b = CAP_L; b < n; b += 256) {
      int v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = gi[min(b + 64 * u + lane, n - 1) - CAP_L];  // (clamped, not predicated: the four loads are in flight together)
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = b + 64 * u + lane < n ? (v[u] & IDMASK) : -1;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const unsigned long long m = __ballot(v[u] == id);
        if (m) return b + 64 * u + (int)__builtin_ctzll(m);
      }
    }
    return -1;
  }

  // ---- jump tables (jps_table_kernel): jt[cell * 32 + code] = the outcome of the jump that leaves `cell` in direction `code` on
  // the map as read — no goal, no cells freed around start and goal: +k the jump ends (forced neighbour, or a lower jump that ends)
  // k cells away, -k the k-th cell is blocked, 0 not known.  A query may use an entry when nothing it depends on has changed:
  //   * the cells that differ from the map as read are the occupied cells inside the two freed cubes: their bounding boxes are
  //     dlo/dhi (empty box: lo > hi);
  //   * a straight jump looks at its ray and the 8 cells around every ray cell: the first ray cell whose surroundings touch a box is
  //     computed exactly (tube_contact); up to there the entry holds, from there on the cells are examined one by one;
  //   * a diagonal jump looks at cells of the cone it opens (one cell of margin): if a box meets the cone, or the goal lies where a
  //     ray of the jump could pass, the entry is not used and the jump is evaluated cell by cell with the entries one level down;
  //   * the goal on a straight ray is arithmetic.
  const short* jt;
  int dlo[2][3], dhi[2][3];
  static constexpr int BIGK = 1 << 28;
  static constexpr int IDMASK = (1 << 27) - 1;  // heap entries: cell | direction id << 27 (a map has at most 2^27 cells)

  __device__ __forceinline__ int tube_contact(int bx, int by, int bz, int ax, int ay, int az, int kend) const {
    int best = BIGK;
    if (!boxes_matter) return best;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      int lo, hi, ba, sg;
      bool perp;
      if (ax) {
        sg = ax; ba = bx; lo = dlo[q][0]; hi = dhi[q][0];
        perp = by >= dlo[q][1] - 1 && by <= dhi[q][1] + 1 && bz >= dlo[q][2] - 1 && bz <= dhi[q][2] + 1;
      } else if (ay) {
        sg = ay; ba = by; lo = dlo[q][1]; hi = dhi[q][1];
        perp = bx >= dlo[q][0] - 1 && bx <= dhi[q][0] + 1 && bz >= dlo[q][2] - 1 && bz <= dhi[q][2] + 1;
      } else {
        sg = az; ba = bz; lo = dlo[q][2]; hi = dhi[q][2];
        perp = bx >= dlo[q][0] - 1 && bx <= dhi[q][0] + 1 && by >= dlo[q][1] - 1 && by <= dhi[q][1] + 1;
      }
      int kf = sg > 0 ? lo - ba : ba - hi, kl = sg > 0 ? hi - ba : ba - lo;
      kf = kf > 1 ? kf : 1;
      kl = kl < kend ? kl : kend;
      if (perp && kf <= kl && kf < best) best = kf;
    }
    return best;
  }
  __device__ __forceinline__ int goal_on_ray(int bx, int by, int bz, int ax, int ay, int az) const {
    const int ex = t[0] - bx, ey = t[1] - by, ez = t[2] - bz;
    int k, off;
    if (ax) { k = ex * ax; off = abs(ey) + abs(ez); }
    else if (ay) { k = ey * ay; off = abs(ex) + abs(ez); }
    else { k = ez * az; off = abs(ex) + abs(ey); }
    return (off == 0 && k >= 1) ? k : BIGK;
  }
  // can a box of changed cells matter to the diagonal jump from (x, y, z) whose entry is J: the box must meet the cone the jump opens
  // (one cell of margin), AND come within reach — every cell the jump examines sits on a ray or a plane sweep that leaves one of
  // the first |J| diagonal cells, so along at least one axis of the jump it is at most |J| + 1 cells away.
  __device__ __forceinline__ bool cone_dirty(int x, int y, int z, int ax, int ay, int az, int J) const {
    bool dirty = false;
    if (boxes_matter) {
      const int reach = abs(J) + 1;
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const bool mx = ax > 0 ? dhi[q][0] >= x : (ax < 0 ? dlo[q][0] <= x : (dlo[q][0] - 1 <= x && x <= dhi[q][0] + 1));
        const bool my = ay > 0 ? dhi[q][1] >= y : (ay < 0 ? dlo[q][1] <= y : (dlo[q][1] - 1 <= y && y <= dhi[q][1] + 1));
        const bool mz = az > 0 ? dhi[q][2] >= z : (az < 0 ? dlo[q][2] <= z : (dlo[q][2] - 1 <= z && z <= dhi[q][2] + 1));
        const int ox = ax > 0 ? dlo[q][0] - x : (ax < 0 ? x - dhi[q][0] : BIGK);
        const int oy = ay > 0 ? dlo[q][1] - y : (ay < 0 ? y - dhi[q][1] : BIGK);
        const int oz = az > 0 ? dlo[q][2] - z : (az < 0 ? z - dhi[q][2] : BIGK);
        dirty = dirty || (mx && my && mz && min(ox, min(oy, oz)) <= reach);
      }
    }
    return dirty;
  }
  __device__ __forceinline__ int entry(int code, int x, int y, int z) const { return (int)jt[(size_t)index(x, y, z) * 32 + code]; }  // the 27 entries of a cell share one 64-byte line
  // A diagonal jump from (x, y, z) along (ax, ay, az) from the entries alone, the goal included (the cone must be clean of changed
  // cells).  -> 0 no jump point, 1 jump point k cells away, 2 not known.
  // The goal can only end the jump at the diagonal cell P_kg where the smallest of its offsets along the jump's axes runs out
  // (before that a ray would need a diagonal move, after that it would have to go back): there it is the cell itself, or sits on a
  // straight ray that leaves the cell, or (space diagonal) in reach of the plane-diagonal jump that leaves it.
  __device__ __forceinline__ int diag_jump(int x, int y, int z, int ax, int ay, int az, int& k) const {
    return diag_jump(x, y, z, ax, ay, az, k, entry((ax + 1) + 3 * (ay + 1) + 9 * (az + 1), x, y, z));
  }
  __device__ __forceinline__ int diag_jump(int x, int y, int z, int ax, int ay, int az, int& k, int J) const {  // J: the entry of the jump itself
    if (J == 0) return 2;
    k = J;
    const int plain = J > 0 ? 1 : 0;
    const int o0 = (t[0] - x) * ax, o1 = (t[1] - y) * ay, o2 = (t[2] - z) * az;
    if ((ax ? o0 < 1 : t[0] != x) || (ay ? o1 < 1 : t[1] != y) || (az ? o2 < 1 : t[2] != z)) return plain;  // the goal is not in the cone
    const int kg = min(ax ? o0 : BIGK, min(ay ? o1 : BIGK, az ? o2 : BIGK));
    if (kg > (J > 0 ? J : -J - 1)) return plain;  // the jump ends before P_kg
    if (kg == J) return 1;
    const int px = x + kg * ax, py = y + kg * ay, pz = z + kg * az;
    const int r0 = ax ? o0 - kg : 0, r1 = ay ? o1 - kg : 0, r2 = az ? o2 - kg : 0;
    const int npos = (r0 > 0) + (r1 > 0) + (r2 > 0);
    bool hit;
    if (npos == 0) hit = true;
    else if (npos == 1) {
      const int code = r0 > 0 ? 13 + ax : (r1 > 0 ? 13 + 3 * ay : 13 + 9 * az);
      const int Js = entry(code, px, py, pz);
      if (Js == 0) return 2;
      hit = r0 + r1 + r2 <= (Js > 0 ? Js : -Js - 1);
    } else {  // the plane-diagonal jump out of P_kg, in the plane of the two offsets that are left
      const int bx = r0 > 0 ? ax : 0, by = r1 > 0 ? ay : 0, bz = r2 > 0 ? az : 0;
      const int J2 = entry((bx + 1) + 3 * (by + 1) + 9 * (bz + 1), px, py, pz);
      if (J2 == 0) return 2;
      const int ra = r0 > 0 ? r0 : r1, rb = r2 > 0 ? r2 : r1;  // the two offsets
      const int k2 = min(ra, rb);
      hit = false;
      if (k2 <= (J2 > 0 ? J2 : -J2 - 1)) {
        if (ra == rb || k2 == J2) hit = true;
        else {
          const int qx = px + k2 * bx, qy = py + k2 * by, qz = pz + k2 * bz;
          const int s0 = r0 > 0 ? r0 - k2 : 0, s1 = r1 > 0 ? r1 - k2 : 0, s2 = r2 > 0 ? r2 - k2 : 0;
          const int code = s0 > 0 ? 13 + ax : (s1 > 0 ? 13 + 3 * ay : 13 + 9 * az);
          const int Js = entry(code, qx, qy, qz);
          if (Js == 0) return 2;
          hit = s0 + s1 + s2 <= (Js > 0 ? Js : -Js - 1);
        }
      }
    }
    if (hit) { k = kg; return 1; }
    return plain;
  }

  // ---- jumps one per lane, in lock step.  phase 3: a straight jump along `a` from P; phase 0: a plane-diagonal jump along d2 from P
  // (0: next diagonal cell; 1 / 2: the straight jumps along a / b that leave it).  `grp` orders the lanes: when a lane has
  // succeeded, the lanes of later groups retire.  One round = one table entry or one examined cell per lane.
  __device__ __forceinline__ bool run_jumps(bool active, int phase, int px, int py, int pz, unsigned d2, unsigned a, unsigned b, int grp) const {
    bool res = false;
    int qx = px, qy = py, qz = pz;
    while (__ballot(active)) {
      if (active) {
        const unsigned pk = phase == 0 ? d2 : (phase == 2 ? b : a);
        const int ax = ux(pk), ay = uy(pk), az = uz(pk), dcode = code_of(pk);
        const int bx = phase == 0 ? px : qx, by = phase == 0 ? py : qy, bz = phase == 0 ? pz : qz;
        int outcome = 0;  // 1 the jump along pk ends (true), 2 it is blocked, 3 examine the next cell, 0 go on (Q moved)
        int J = 0;
        if (phase == 0) {
          int kk;
          const int Jd = entry(dcode, bx, by, bz);
          const int st = cone_dirty(bx, by, bz, ax, ay, az, Jd) ? 2 : diag_jump(bx, by, bz, ax, ay, az, kk, Jd);
          outcome = st == 2 ? 3 : (st == 1 ? 1 : 2);
        } else J = entry(dcode, bx, by, bz);
        if (phase == 0) {}
        else if (J == 0) outcome = 3;
        else {
          const int kend = abs(J);
          const int k0 = tube_contact(bx, by, bz, ax, ay, az, kend), kt = goal_on_ray(bx, by, bz, ax, ay, az);
          if (k0 > kend) outcome = (J > 0 || kt < kend || (kt == kend && J > 0)) ? 1 : 2;
          else if (k0 > 1) {
            if (kt < k0) outcome = 1;
            else { qx = bx + (k0 - 1) * ax; qy = by + (k0 - 1) * ay; qz = bz + (k0 - 1) * az; }
          } else outcome = 3;
        }
        int x = 0, y = 0, z = 0;
        if (outcome == 3) {
          x = bx + ax; y = by + ay; z = bz + az;
          const bool fr = free_any(x, y, z);
          const bool ev = ends_at(x, y, z, dcode, phase == 0 ? 2 : 1);
          outcome = (fr && ev) ? 1 : (!fr ? 2 : 4);  // 4: a free cell that ends nothing
          if (phase == 0) { px = x; py = y; pz = z; }
        }
        if (outcome == 1) { res = true; active = false; }
        else if (outcome == 2) {
          if (phase == 3 || phase == 0) active = false;
          else if (phase == 1) { phase = 2; qx = px; qy = py; qz = pz; }
          else phase = 0;
        } else if (outcome == 4) {
          if (phase == 0) phase = 1;
          qx = x; qy = y; qz = z;
        }
      }
      const unsigned long long tm = __ballot(res);
      if (tm) {
        const int g0 = __builtin_amdgcn_readlane(grp, (int)__builtin_ctzll(tm));
        if (grp > g0) active = false;
      }
    }
    return res;
  }

  // graph_search.cpp:374-400 for one successor direction; (ox, oy, oz): the jump point
  __device__ __forceinline__ bool jump(int cx, int cy, int cz, unsigned pk, int& ox, int& oy, int& oz) const {
    const int dx = ux(pk), dy = uy(pk), dz = uz(pk), code = code_of(pk), n1 = abs(dx) + abs(dy) + abs(dz);
    int bx = cx, by = cy, bz = cz;
    if (n1 == 1) {
      for (;;) {
        const int J = rfl(entry(code, bx, by, bz));
        if (J != 0) {
          const int kend = abs(J);
          const int k0 = tube_contact(bx, by, bz, dx, dy, dz, kend), kt = goal_on_ray(bx, by, bz, dx, dy, dz);
          int k = -1;
          if (k0 > kend) {
            if (kt < kend || (kt == kend && J > 0)) k = kt;
            else if (J > 0) k = J;
            else return false;
          } else if (k0 > 1) {
            if (kt < k0) k = kt;
            else { bx += (k0 - 1) * dx; by += (k0 - 1) * dy; bz += (k0 - 1) * dz; }
          }
          if (k > 0) { ox = bx + k * dx; oy = by + k * dy; oz = bz + k * dz; return true; }
        }
        // the cells ahead may differ from the map as read: 64 of them at once
        const int k = lane + 1;
        const int x = bx + k * dx, y = by + k * dy, z = bz + k * dz;
        const bool fr = free_any(x, y, z);
        const bool ev = ends_at(x, y, z, code, 1);
        const unsigned long long bm = __ballot(!fr), em = __ballot(fr && ev);
        const int fb = bm ? (int)__builtin_ctzll(bm) : 64, fe = em ? (int)__builtin_ctzll(em) : 64;
        if (fe < fb) { ox = bx + (fe + 1) * dx; oy = by + (fe + 1) * dy; oz = bz + (fe + 1) * dz; return true; }
        if (fb < 64) return false;
        bx += 64 * dx; by += 64 * dy; bz += 64 * dz;
      }
    }
    const int per = n1 == 2 ? 2 : 8, shift = n1 == 2 ? 1 : 3, steps = 64 / per;
    const int sub = lane & (per - 1);
    // this lane's jump out of a diagonal cell
    int phase = 3;
    unsigned d2 = 0, a = 0, b = 0;
    bool takes = true;
    if (n1 == 2) a = nat(code, sub);
    else if (sub < 3) a = nat(code, sub);
    else if (sub < 6) {
      phase = 0;
      d2 = nat(code, sub);
      const int c2 = code_of(d2);
      a = nat(c2, 0);
      b = nat(c2, 1);
    } else takes = false;
    for (;;) {
      const int Jd = rfl(entry(code, bx, by, bz));
      if (!cone_dirty(bx, by, bz, dx, dy, dz, Jd)) {
        int kk = 0;
        const int st = rfl(diag_jump(bx, by, bz, dx, dy, dz, kk, Jd));
        kk = rfl(kk);
        if (st == 0) return false;
        if (st == 1) { ox = bx + kk * dx; oy = by + kk * dy; oz = bz + kk * dz; return true; }
      }
      const int k = (lane >> shift) + 1;
      const int x = bx + k * dx, y = by + k * dy, z = bz + k * dz;
      const bool fr = free_any(x, y, z);
      const bool ev = ends_at(x, y, z, code, n1);
      const unsigned long long sm = __ballot(!fr || ev);
      const int kstop = sm ? ((int)__builtin_ctzll(sm) >> shift) + 1 : steps + 1;
      const bool res = run_jumps(takes && k < kstop, phase, x, y, z, d2, a, b, k);
      const unsigned long long tm = __ballot(res);
      if (tm) {
        const int kt = ((int)__builtin_ctzll(tm) >> shift) + 1;
        ox = bx + kt * dx; oy = by + kt * dy; oz = bz + kt * dz;
        return true;
      }
      if (kstop <= steps) {
        ox = bx + kstop * dx; oy = by + kstop * dy; oz = bz + kstop * dz;
        return ((__ballot(fr) >> ((kstop - 1) << shift)) & 1ull) != 0ull;
      }
      bx += steps * dx; by += steps * dy; bz += steps * dz;
    }
  }

  // the occupied cells inside the cube freed around centre c (they are free for this query): their bounding box
  __device__ __forceinline__ void dirty_box(const int c[3], int q) {
    const int m = mv.m_free, w = 2 * m + 1, count = w * w * w;
    int lo0 = BIGK, lo1 = BIGK, lo2 = BIGK, hi0 = -BIGK, hi1 = -BIGK, hi2 = -BIGK;
    for (int i = lane; i < count; i += 64) {
      const int x = c[0] - m + i % w, y = c[1] - m + (i / w) % w, z = c[2] - m + i / (w * w);
      if (!outside(x, y, z)) {
        const int id = index(x, y, z);
        if ((mv.bits[id >> 5] >> (id & 31)) & 1u) {
          lo0 = min(lo0, x); lo1 = min(lo1, y); lo2 = min(lo2, z);
          hi0 = max(hi0, x); hi1 = max(hi1, y); hi2 = max(hi2, z);
        }
      }
    }
    dlo[q][0] = wave_min_i32(lo0); dlo[q][1] = wave_min_i32(lo1); dlo[q][2] = wave_min_i32(lo2);
    dhi[q][0] = -wave_min_i32(-hi0); dhi[q][1] = -wave_min_i32(-hi1); dhi[q][2] = -wave_min_i32(-hi2);
  }

  __device__ __forceinline__ double heur_jps(int x, int y, int z) const {  // graph_search.cpp:72-74, eps = 1
    return sqrt((double)dist2(x, y, z));
  }

  // HASHED: the cell records are the wavefront's hashed table (hk / hr above) instead of cells[cell]; what is read and written, and
  // in which order, is the same, so the two return the same paths.
  template <bool HASHED>
  __device__ __forceinline__ int search_jps(CellState* cells, unsigned* chunks, unsigned serial, long long& expansions) {
    gf = (pool_f64*)chunks;
    gg = gf + cap_g;
    gi = (pool_i32*)(gg + cap_g);
    const int sid = index(s[0], s[1], s[2]), tid = index(t[0], t[1], t[2]);
    const int nxy = mv.nx * mv.ny;
    boxes_matter = true;
    dirty_box(s, 0);
    dirty_box(t, 1);
    boxes_matter = dlo[0][0] <= dhi[0][0] || dlo[1][0] <= dhi[1][0];
    {
      HE e;
      e.id = sid | (13 << 27); e.g = 0.0; e.f = 0.0 + heur_jps(s[0], s[1], s[2]);
      hset(0, e);
      CellState cs; cs.g = 0.0; cs.parent = -1; cs.stamp = (serial << 6) | (13u << 1);
      if (HASHED) {
        hser = (unsigned long long)serial << 32;
        hcount = 0;
        unsigned slot;
        bool present;
        (void)hfind_uniform(sid, slot, present);
        if (!hwrite_uniform(sid, slot, present, cs)) return -2;
      } else if (lane == 0) {
        cells[sid] = cs;
      }
    }
    int n = 1;
    long long pops = 0;
    for (;;) {  // graph_search.cpp:123-217
      expansions++;
      if (++pops > (long long)mv.total) return FHP_LIMIT(1);  // (a cell is opened once: cannot happen)
      FHP_T(-2);
      const HE top = hget(0);
      const int cur = rfl(top.id) & IDMASK, code = (rfl(top.id) >> 27) & 31;  // (the heap entry carries the direction the node was reached in)
      unsigned long long ckey = 0ull;  // (hashed records: the key in the home slot of the popped cell, on its way while the heap is put in order)
      if (HASHED) ckey = hk[home(cur)];
      else if (lane == 0) cells[cur].stamp = (serial << 6) | ((unsigned)code << 1) | 1u;  // closed
      const int cz = fhu::div(cur, nxy, mv.inv_nxy), rem = cur - cz * nxy, cy = fhu::div(rem, mv.nx, mv.inv_nx), cx = rem - cy * mv.nx;
      const int n1 = abs(code % 3 - 1) + abs((code / 3) % 3 - 1) + abs(code / 9 - 1);
      const int num_neib = n1 == 0 ? 26 : (n1 == 1 ? 1 : (n1 == 2 ? 3 : 7)), num_fneib = n1 == 0 ? 0 : (n1 == 1 ? 8 : 12);
      // ---- the successors (getJpsSucc, :318-368), one candidate per lane in jps3d's order: lanes < num_neib the natural neighbours,
      // then the forced-neighbour entries.  A lane settles its jump from the jump tables when it can; the others are evaluated one
      // after the other by the whole wavefront.  The loads a candidate needs (its forced-neighbour cell, its table entry) depend on
      // the popped node only: they are issued before the heap is put in order again and arrive while it is.
      const bool cand = lane < num_neib + num_fneib;
      const int fk = (cand && lane >= num_neib) ? lane - num_neib : 0;
      const unsigned fpk = byte_of(jf1 + code * 3, fk);
      const unsigned pk = !cand ? 0x15u : (lane < num_neib ? nat(code, lane) : byte_of(jf2 + code * 3, fk));
      const int ax = ux(pk), ay = uy(pk), az = uz(pk), pcode = code_of(pk);
      const bool straight = abs(ax) + abs(ay) + abs(az) == 1;
      const int fx = cx + ux(fpk), fy = cy + uy(fpk), fz = cz + uz(fpk);
      const bool fin = !outside(fx, fy, fz);
      const int fid = fin ? index(fx, fy, fz) : 0;
      const unsigned fword = mv.bits[fid >> 5];
      const int J = jt[(size_t)cur * 32 + pcode];
      n--;
      FHP_T(0);
#ifdef FHP_PROFILE
      const long long prof_t0 = (long long)__builtin_readcyclecounter();
#endif
      if (n > 0) {
        const HE last = hget(n);
        sift_down(last, n);
      }
      FHP_T(1);
#ifdef FHP_PROFILE
      if (prof_slot == 8 && n <= CAP_L) prof_acc += (long long)__builtin_readcyclecounter() - prof_t0;
      if (prof_slot == 9 && n > CAP_L) prof_acc += (long long)__builtin_readcyclecounter() - prof_t0;
      if (prof_slot == 10 && n > CAP_L) prof_acc++;
      if (prof_slot == 11) prof_acc += n;
#endif
      if (cur == tid) break;
      if (HASHED) {  // closed
        unsigned cslot = home(cur);
        const unsigned long long ku = ((unsigned long long)(unsigned)rfl((int)(unsigned)(ckey >> 32)) << 32) | (unsigned)rfl((int)(unsigned)ckey);
        if (ku != (hser | (unsigned long long)(unsigned)cur)) {
          bool present;
          (void)hfind_uniform(cur, cslot, present);
          if (!present) return -2;  // (cannot happen: a popped cell has a record)
        }
        if (lane == 0) hr[cslot].stamp = (serial << 6) | ((unsigned)code << 1) | 1u;
      }
      const bool applies = lane < num_neib || (fin && !freed(fx, fy, fz) && ((fword >> (fid & 31)) & 1u));
      int status = 2, jk = 0;  // 0 no successor, 1 a successor jk cells away, 2 not settled
      if (!cand) {}
      else if (!straight) {
        if (!cone_dirty(cx, cy, cz, ax, ay, az, J)) status = diag_jump(cx, cy, cz, ax, ay, az, jk, J);
      } else if (J != 0) {
        {
          const int kend = abs(J);
          const int k0 = tube_contact(cx, cy, cz, ax, ay, az, kend), kt = goal_on_ray(cx, cy, cz, ax, ay, az);
          if (k0 > kend) {
            if (kt < kend || (kt == kend && J > 0)) { status = 1; jk = kt; }
            else if (J > 0) { status = 1; jk = J; }
            else status = 0;
          } else if (kt < k0) { status = 1; jk = kt; }
        }
      }
      if (!cand || !applies) status = 0;
      int jx = cx + jk * ax, jy = cy + jk * ay, jz = cz + jk * az;
      FHP_T(2);
      for (unsigned long long um = __ballot(status == 2); um; um &= um - 1ull) {
        const int j = (int)__builtin_ctzll(um);
        int ox, oy, oz;
        const bool found = jump(cx, cy, cz, (unsigned)__builtin_amdgcn_readlane((int)pk, j), ox, oy, oz);
        if (lane == j) {
          status = found ? 1 : 0;
          jx = ox; jy = oy; jz = oz;
        }
      }
      FHP_T(3);
      const bool ok = status == 1;
      const int nid = ok ? index(jx, jy, jz) : cur;
      // per successor, all at once: the cost of the move, h of the jump point, the direction id of the sign of the move
      const int ex = jx - cx, ey = jy - cy, ez = jz - cz;
      const double lcost = sqrt((double)(ex * ex + ey * ey + ez * ez)), lheur = heur_jps(jx, jy, jz);
      const int lsign = (ex > 0 ? 2 : (ex < 0 ? 0 : 1)) + 3 * (ey > 0 ? 2 : (ey < 0 ? 0 : 1)) + 9 * (ez > 0 ? 2 : (ez < 0 ? 0 : 1));
      settle();
      unsigned nslot = 0u;
      bool npresent = false;
      CellState ns;
      if (HASHED) ns = hfind(ok, nid, nslot, npresent);
      else ns = cells[nid];
      const unsigned long long okm = __ballot(ok);
      // a cell reached by two successors of this node: the second sees what the first wrote
      bool dup = false;
      if (__popcll(okm) > 1)
        for (unsigned long long m2 = okm; m2; m2 &= m2 - 1ull) {
          const int i = (int)__builtin_ctzll(m2);
          if (ok && i < lane && __builtin_amdgcn_readlane(nid, i) == nid) dup = true;
        }
      // ---- relaxation (:150-191).  What does not depend on the order is done by all successors at once: the comparison with the
      // stored g, the new cell record (a cell reached twice waits for its turn below).  The heap operations follow one after the
      // other in getJpsSucc's order.
      FHP_T(4);
      const bool visited = (ns.stamp >> 6) == serial;
      const bool closed = visited && (ns.stamp & 1u);
      const double tentative = top.g + lcost;
      const bool improves = ok && !dup && (!visited || tentative < ns.g);
      const int lcode = !visited ? pcode : (closed ? (int)((ns.stamp >> 1) & 31u) : lsign);  // an open node takes the sign of the move (:176-181)
      const double lf = tentative + lheur;
      {
        CellState w; w.g = tentative; w.parent = cur; w.stamp = (serial << 6) | ((unsigned)lcode << 1) | (closed ? 1u : 0u);
        if (HASHED) {
          if (!hwrite(improves, nid, nslot, npresent, w)) return -2;
        } else if (improves) {
          cells[nid] = w;  // (closed: jps3d updates g and the parent and goes on)
        }
      }
      const unsigned long long dupm = __ballot(dup);
      FHP_T(5);
      for (unsigned long long m2 = __ballot((improves && !closed) || dup); m2; m2 &= m2 - 1ull) {
        const int j = (int)__builtin_ctzll(m2);
        const int nj = __builtin_amdgcn_readlane(nid, j);
        HE me;
        bool push = !(bool)__builtin_amdgcn_readlane((int)visited, j);
        if (!((dupm >> j) & 1ull)) {
          me.g = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tentative), j), __builtin_amdgcn_readlane(__double2loint(tentative), j));
          me.f = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(lf), j), __builtin_amdgcn_readlane(__double2loint(lf), j));
          me.id = nj | (__builtin_amdgcn_readlane(lcode, j) << 27);
        } else {  // the cell was reached by an earlier successor of this node: its record is read again
          settle();
          unsigned jslot = 0u;
          bool jpresent = false;
          CellState again;
          if (HASHED) again = hfind_uniform(nj, jslot, jpresent);
          else again = cells[nj];
          const unsigned nstamp = (unsigned)rfl((int)again.stamp);
          const double ng = __hiloint2double(rfl(__double2hiint(again.g)), rfl(__double2loint(again.g)));
          const bool v2 = (nstamp >> 6) == serial, c2 = v2 && (nstamp & 1u);
          me.g = top.g + __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(lcost), j), __builtin_amdgcn_readlane(__double2loint(lcost), j));
          if (v2 && !(me.g < ng)) continue;
          me.f = me.g + __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(lheur), j), __builtin_amdgcn_readlane(__double2loint(lheur), j));
          const unsigned ncode = !v2 ? (unsigned)__builtin_amdgcn_readlane(pcode, j) : (c2 ? ((nstamp >> 1) & 31u) : (unsigned)__builtin_amdgcn_readlane(lsign, j));
          me.id = nj | (int)(ncode << 27);
          {
            CellState w; w.g = me.g; w.parent = cur; w.stamp = (serial << 6) | (ncode << 1) | (c2 ? 1u : 0u);
            if (HASHED) {
              if (!hwrite_uniform(nj, jslot, jpresent, w)) return -2;
            } else if (lane == 0) {
              cells[nj] = w;
            }
          }
          if (c2) continue;
          push = !v2;
        }
        if (push) {
          if (n >= CAP_L + cap_g) return FHP_LIMIT(2);
          sift_up(n, me);
          n++;
        } else {  // pq_.increase
          const int pos = heap_find(nj, n);
          if (pos < 0) return FHP_LIMIT(3);  // (cannot happen)
          sift_up(pos, me);
        }
      }
      FHP_T(6);
      if (n == 0) return 0;
    }
    settle();
    return finish_path<HASHED>(cells, chunks, sid, tid);
  }
};

constexpr int PLAN_LDS_BYTES = NCHUNK * 2 * 2 + NS * 4 * 4 + NS * 2;
constexpr int JPS_LDS_BYTES = Planner::CAP_L * 20 + Planner::JTAB_WORDS * 4;
constexpr unsigned JPS_SERIAL_LIMIT = (1u << 26) - 1u;

// One wavefront per workgroup.  Output vertices: the cleaned path with its ends forced onto start and goal
// (jps_manager.cpp:175-186), then optionally createMoreVertexes / deleteVertexes.
template <bool JPS, bool HASHED = false>
__global__ void __launch_bounds__(64, JPS ? 5 : 3) plan_kernel(MapView mv, PlanArgs pa) {
  __shared__ __attribute__((aligned(16))) char lds[JPS ? JPS_LDS_BYTES : PLAN_LDS_BYTES];
  Planner pl(mv, lds);
  if (JPS) pl.init_jps(lds, pa.jps_tables, pa.jps_entries);
  const int lane = pl.lane;
  const int wave = (int)blockIdx.x;
  CellState* cells = pa.cells + (size_t)wave * (HASHED ? pa.hslots : mv.total);
  if (HASHED) {
    pl.hk = pa.hkeys + (size_t)wave * pa.hslots;
    pl.hr = cells;
    pl.hmask = (unsigned)pa.hslots - 1u;
    pl.hshift = 32u - (unsigned)(31 - __builtin_clz((unsigned)pa.hslots));
    pl.cap_g = (pa.hslots >> 2) * 3;
  }
  unsigned* chunks = pa.chunks + (size_t)wave * (size_t)pa.chunk_words;
  unsigned serial = pa.serials[wave];
  for (;;) {
    int q = 0;
    if (lane == 0) q = atomicAdd(pa.ticket, 1);
    q = rfl(q);
    if (q >= pa.n) break;
    if (pa.order) q = rfl(pa.order[q]);
    double st[3], gl[3];
    for (int k = 0; k < 3; k++) { st[k] = pa.starts[3 * q + k]; gl[k] = pa.goals[3 * q + k]; }
    st[2] = fmax(st[2], 0.0);  // jps_manager.cpp:143-144
    gl[2] = fmax(gl[2], 0.0);
    pl.to_cell(st[0], st[1], st[2], pl.s);
    pl.to_cell(gl[0], gl[1], gl[2], pl.t);
    for (int k = 0; k < 3; k++) { pl.s[k] = rfl(pl.s[k]); pl.t[k] = rfl(pl.t[k]); }
    long long expansions = 0;
    int nv = 0;
#ifdef FHP_PROFILE
    pl.prof_slot = pa.profile_slot;
    pl.prof_acc = 0;
    const long long q_t0 = (long long)__builtin_readcyclecounter();
#endif
    if (!pl.outside(pl.s[0], pl.s[1], pl.s[2]) && !pl.outside(pl.t[0], pl.t[1], pl.t[2])) {
      serial++;
      if (serial >= (JPS ? JPS_SERIAL_LIMIT : 0x7fffffffu)) {  // 2^31 (2^26) queries of this wavefront: its stamps start over, so its cell states are cleared first
        if (HASHED) for (int c = lane; c < pa.hslots; c += 64) pl.hk[c] = 0ull;
        else for (int c = lane; c < mv.total; c += 64) cells[c].stamp = 0u;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        serial = 1;
      }
      nv = JPS ? pl.template search_jps<HASHED>(cells, chunks, serial, expansions) : pl.search(pa, cells, chunks, serial, expansions);
    }
    double* out = pa.paths + (size_t)q * pa.max_points * 3;
    int np = nv;
    if (nv > 0) {
      // vertices in LDS order va[0..nv): ends forced
      const int count = nv > 1 ? nv : 2;
      // createMoreVertexes / zero-length legs / deleteVertexes are sequential and short: lane 0 walks the legs
      if (lane == 0) {
        int w = 0;
        double last[3] = {st[0], st[1], st[2]};  // the last vertex kept
        auto put = [&](const double p[3]) {
          if (w < pa.max_points) { out[3 * w] = p[0]; out[3 * w + 1] = p[1]; out[3 * w + 2] = p[2]; }
          w++;
          last[0] = p[0]; last[1] = p[1]; last[2] = p[2];
        };
        // a vertex closer than 1e-9 to the last kept one is dropped (createMoreVertexes repeats the end of a leg that is an exact
        // multiple of the spacing; the host restatement erases the later of the two)
        auto emit = [&](const double p[3]) {
          if (pa.max_vertex_dist > 0.0 && Planner::dist(p, last) < 1e-9) return;
          put(p);
        };
        put(last);
        // JPS_in (faster.cpp:370-382): the path up to its first crossing of the sphere of radius min(|goal - start| - 0.001, Ra) around
        // the start, the crossing point E appended (getFirstIntersectionWithSphere, utils.cpp:782-870)
        auto vertex = [&](int i, double b[3]) {
          if (i == 0) { b[0] = st[0]; b[1] = st[1]; b[2] = st[2]; }
          else if (i == count - 1) { b[0] = gl[0]; b[1] = gl[1]; b[2] = gl[2]; }
          else pl.center(pl.va[i], b);
        };
        int n_in = count;
        bool has_e = false;
        double E[3] = {0, 0, 0};
        if (pa.sphere_ra > 0.0) {
          const double ra = fmin(Planner::dist(gl, st) - 0.001, pa.sphere_ra);
          for (int i = 1; i < count; i++) {
            double b[3];
            vertex(i, b);
            if (Planner::dist(b, st) > ra) {
              double p[3];
              vertex(i - 1, p);
              sphere_crossing(p, b, ra, st, E);
              n_in = i + 1;  // vertices 0 .. i-1, then E
              has_e = true;
              break;
            }
          }
        }
        double a[3] = {st[0], st[1], st[2]};  // start of the current leg: always the ORIGINAL vertex, as in createMoreVertexes
        for (int i = 1; i < n_in; i++) {
          double b[3];
          if (has_e && i == n_in - 1) { b[0] = E[0]; b[1] = E[1]; b[2] = E[2]; }
          else vertex(i, b);
          if (pa.max_vertex_dist > 0.0) {
            const double d = Planner::dist(b, a);
            if (d > pa.max_vertex_dist) {
              const int add = (int)floor(d / pa.max_vertex_dist);
              const double vx = (b[0] - a[0]) / d, vy = (b[1] - a[1]) / d, vz = (b[2] - a[2]) / d;
              double q[3] = {a[0], a[1], a[2]};
              for (int k = 0; k < add; k++) {
                q[0] = q[0] + vx * pa.max_vertex_dist; q[1] = q[1] + vy * pa.max_vertex_dist; q[2] = q[2] + vz * pa.max_vertex_dist;
                emit(q);
              }
            }
          }
          emit(b);
          a[0] = b[0]; a[1] = b[1]; a[2] = b[2];
        }
        if (pa.max_poly > 0 && w > pa.max_poly + 1) w = pa.max_poly + 1;
        np = w > pa.max_points ? -1 : w;
      }
      np = rfl(np);
    }
    if (lane == 0) {
      pa.n_points[q] = np;
#ifdef FHP_PROFILE
      if (pa.profile_slot == 7) expansions = (long long)__builtin_readcyclecounter() - q_t0;
      else if (pa.profile_slot >= 0) expansions = pl.prof_acc;
#endif
      if (pa.expansions) pa.expansions[q] = expansions;
    }
  }
  if (lane == 0) pa.serials[wave] = serial;
}

// Launch order of a batch of queries: by the distance between start and goal, farthest first (counting sort over 64 classes, two small
// launches; `counters`: 128 zeroed ints).  The number of expanded cells grows with the distance (correlation 0.47 in the forest
// maps, and the maximum is 20x the mean), and a long search that starts last is what the launch ends on: simulated makespan of 65536
// forest queries on 4096 wavefronts 29.4 k expansions in the given order, 18.8 k farthest first (17.6 k: the longest single query).
__device__ __forceinline__ int plan_class(const double* s, const double* g, double scale) {
  const double dx = g[0] - s[0], dy = g[1] - s[1], dz = g[2] - s[2];
  const double v = sqrt(dx * dx + dy * dy + dz * dz) * scale;
  return v >= 63.0 ? 63 : (v > 0.0 ? (int)v : 0);   // (NaN: class 0)
}
__global__ void __launch_bounds__(256) plan_order_hist_kernel(const double* starts, const double* goals, int n, double scale, int* counters) {
  __shared__ int cnt[64];
  if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i < n) atomicAdd(&cnt[plan_class(starts + 3 * (size_t)i, goals + 3 * (size_t)i, scale)], 1);
  __syncthreads();
  if (threadIdx.x < 64 && cnt[threadIdx.x]) atomicAdd(&counters[threadIdx.x], cnt[threadIdx.x]);
}
__global__ void __launch_bounds__(256) plan_order_scatter_kernel(const double* starts, const double* goals, int n, double scale, int* counters,
                                                                 int* order) {
  __shared__ int cnt[64], base[64];
  if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  int k = 0, mine = 0;
  if (i < n) {
    k = plan_class(starts + 3 * (size_t)i, goals + 3 * (size_t)i, scale);
    mine = atomicAdd(&cnt[k], 1);
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    int before = 0;
    for (int c = 63; c > (int)threadIdx.x; c--) before += counters[c];
    base[threadIdx.x] = before + (cnt[threadIdx.x] ? atomicAdd(&counters[64 + threadIdx.x], cnt[threadIdx.x]) : 0);
  }
  __syncthreads();
  if (i < n) order[base[k] + mine] = i;
}

// Jump tables of a map (see Planner::jt): one launch per level (1 straight, 2 plane-diagonal, 3 space-diagonal; a level reads the
// entries of the levels below).  A thread owns one line of cells in one direction — it starts from the cell whose successor lies
// outside the grid and walks backwards, so every entry costs one step: entry(c) = -1 if c + d is blocked, +1 if the jump ends at
// c + d (forced neighbour there, or a lower jump from there that ends), else entry(c + d) one further away.
__global__ void __launch_bounds__(256) jps_table_kernel(MapView mv, const unsigned char* tb, short* jt, int level) {
  const int ndirs = level == 1 ? 6 : (level == 2 ? 12 : 8);
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)mv.total * ndirs) return;
  const int cell = (int)(gid % mv.total), which = (int)(gid / mv.total);
  int code = 0;
  for (int c = 0, seen = 0; c < 27; c++) {
    const int n1 = abs(c % 3 - 1) + abs((c / 3) % 3 - 1) + abs(c / 9 - 1);
    if (n1 == level) {
      if (seen == which) { code = c; break; }
      seen++;
    }
  }
  const int dx = code % 3 - 1, dy = (code / 3) % 3 - 1, dz = code / 9 - 1;
  const int nxy = mv.nx * mv.ny;
  int z = cell / nxy, y = (cell - z * nxy) / mv.nx, x = cell - z * nxy - y * mv.nx;
  auto outside = [&](int px, int py, int pz) { return px < 0 || py < 0 || pz < 0 || px >= mv.nx || py >= mv.ny || pz >= mv.nz; };
  auto occupied = [&](int px, int py, int pz) {
    if (outside(px, py, pz)) return false;
    const int id = px + mv.nx * py + nxy * pz;
    return ((mv.bits[id >> 5] >> (id & 31)) & 1u) != 0u;
  };
  if (!outside(x + dx, y + dy, z + dz)) return;  // not the last cell of its line
  const int nsub = level == 1 ? 0 : (level == 2 ? 2 : 6), nforced = level == 3 ? 6 : 8;
  short* mine = jt + code;  // [cell][32]: the entries of a cell side by side
  while (!outside(x, y, z)) {
    const int X = x + dx, Y = y + dy, Z = z + dz;
    short val;
    if (outside(X, Y, Z) || occupied(X, Y, Z)) val = -1;
    else {
      const int xid = X + mv.nx * Y + nxy * Z;
      bool ends = false, unknown = false;
      for (int fn = 0; fn < nforced; fn++) {
        const unsigned pk = tb[27 * 28 + code * 12 + fn];
        ends = ends || occupied(X + (int)(pk & 3u) - 1, Y + (int)((pk >> 2) & 3u) - 1, Z + (int)((pk >> 4) & 3u) - 1);
      }
      for (int k = 0; k < nsub && !ends; k++) {
        const unsigned pk = tb[code * 28 + k];
        const int c2 = (int)((pk & 3u) + 3u * ((pk >> 2) & 3u) + 9u * ((pk >> 4) & 3u));
        const short v = jt[(size_t)xid * 32 + c2];
        ends = v > 0;
        unknown = unknown || v == 0;
      }
      if (ends) val = 1;
      else if (unknown) val = 0;
      else {
        const short nxt = mine[(size_t)xid * 32];
        val = (nxt == 0 || nxt >= 32766 || nxt <= -32766) ? (short)0 : (short)(nxt > 0 ? nxt + 1 : nxt - 1);
      }
    }
    mine[(size_t)(x + mv.nx * y + nxy * z) * 32] = val;
    x -= dx; y -= dy; z -= dz;
  }
}

// MapUtil::readMap (read_map.hpp:100-185): every point marks its cell and the cube of +-m cells around it; the flat index test is
// the reference's (no per-axis clipping).
__global__ void mark_kernel(const double* cloud, int n, int nx, int ny, int nz, double res, double ox, double oy, double oz, int m,
                            unsigned* bits) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  int c[3];
  c[0] = (int)round((cloud[3 * i] - ox) / res - 0.5);
  c[1] = (int)round((cloud[3 * i + 1] - oy) / res - 0.5);
  c[2] = (int)round((cloud[3 * i + 2] - oz) / res - 0.5);
  for (int k = 0; k < 3; k++) c[k] = c[k] > 0 ? c[k] : 0;
  const long long total = (long long)nx * ny * nz;
  for (int ix = c[0] - m; ix <= c[0] + m; ix++)
    for (int iy = c[1] - m; iy <= c[1] + m; iy++)
      for (int iz = c[2] - m; iz <= c[2] + m; iz++) {
        const long long id = ix + (long long)nx * iy + (long long)nx * ny * iz;
        if (id >= 0 && id < total) atomicOr(&bits[id >> 5], 1u << (id & 31));
      }
}

}  // namespace fhp
