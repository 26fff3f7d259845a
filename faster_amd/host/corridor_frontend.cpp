// corridor_frontend.cpp — path search + C wrapper of the corridor front-end (see corridor_frontend.hpp).
#include "corridor_frontend.hpp"
#include "jps_tables.hpp"

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <limits>
#include <queue>

namespace fhfront {

namespace {

// jps3d's clean-up of the raw cell path (jps_planner.cpp:83-105, :36-81)
std::vector<V3> remove_line_points(const std::vector<V3>& path) {
  if (path.size() < 3) return path;
  std::vector<V3> out;
  out.push_back(path.front());
  for (size_t i = 1; i + 1 < path.size(); i++) {
    const V3 p = (path[i + 1] - path[i]) - (path[i] - path[i - 1]);
    if (std::fabs(p.x) + std::fabs(p.y) + std::fabs(p.z) > 1e-2) out.push_back(path[i]);
  }
  out.push_back(path.back());
  return out;
}

std::vector<V3> remove_corner_points(const VoxelGrid& g, const std::vector<V3>& path) {
  if (path.size() < 2) return path;
  const double inf = std::numeric_limits<double>::infinity();
  std::vector<V3> out;
  V3 prev = path[0];
  out.push_back(prev);
  double c1 = g.blocked(path[0], path[1]) ? inf : (path[0] - path[1]).norm();
  for (size_t i = 1; i + 1 < path.size(); i++) {
    const V3 a = path[i], b = path[i + 1];
    const double c2 = g.blocked(a, b) ? inf : (a - b).norm();
    const double c3 = g.blocked(prev, b) ? inf : (prev - b).norm();
    if (c3 < c1 + c2) c1 = c3;
    else {
      out.push_back(a);
      c1 = (a - b).norm();
      prev = a;
    }
  }
  out.push_back(path.back());
  return out;
}

// Open-list order.  jps3d compares f with a 1e-6 tolerance and prefers the larger g on ties (graph_search.h:19-29) — not a strict
// weak order, so the expansion order depends on the heap implementation.  Here the order is a strict TOTAL one: f quantised to
// 2^-20 cells, then the squared straight-line distance to the goal (with the exact empty-grid heuristic whole plateaus of cells
// share one f: the tie-breaker is what makes the search dive for the goal — 419 instead of 850 expanded cells per forest query),
// then the cell index.  Any priority queue then expands the same cells in the same order — the device search
// (csrc/fh_path.hip.hpp, a bucket queue) is checked against this function vertex for vertex.
struct Node {
  int key, h2, id;
};
struct NodeOrder {
  bool operator()(const Node& a, const Node& b) const {
    if (a.key != b.key) return a.key > b.key;
    if (a.h2 != b.h2) return a.h2 > b.h2;
    return a.id > b.id;
  }
};
constexpr double kKeyScale = 1048576.0;

}  // namespace

bool plan_path(VoxelGrid& grid, const V3& start_in, const V3& goal_in, double inflation, std::vector<V3>& path, long long* expansions) {
  path.clear();
  const V3 start(start_in.x, start_in.y, std::max(start_in.z, 0.0)), goal(goal_in.x, goal_in.y, std::max(goal_in.z, 0.0));
  int s[3], t[3];
  grid.to_cell(start, s);
  grid.to_cell(goal, t);
  grid.set_free_around(s, inflation);  // jps_manager.cpp:158-159
  grid.set_free_around(t, inflation);
  if (!grid.is_free(s[0], s[1], s[2]) || !grid.is_free(t[0], t[1], t[2])) return false;

  const int total = grid.nx * grid.ny * grid.nz;
  std::vector<double> gval((size_t)total, std::numeric_limits<double>::infinity());
  std::vector<int> parent((size_t)total, -1);
  std::vector<char> closed((size_t)total, 0);
  std::priority_queue<Node, std::vector<Node>, NodeOrder> open;
  const int sid = grid.index(s[0], s[1], s[2]), tid = grid.index(t[0], t[1], t[2]);
  // Heuristic: the exact length of a shortest 26-connected path in an EMPTY grid — with a >= b >= c the sorted absolute cell
  // offsets, c diagonal steps through space, b - c diagonal steps in a plane, a - b straight steps.  It is consistent (a metric of the
  // grid graph) and dominates jps3d's Euclidean distance (graph_search.cpp:73-75), so the search stays optimal and expands 6.5x
  // fewer cells in the forest maps (857 instead of 5542 per query); like jump-point pruning it changes which of the equal-cost
  // paths is found, not their cost.  Only +, * and sqrt: the device search (csrc/fh_path.hip.hpp) computes the same bits.
  const double kSqrt2 = std::sqrt(2.0), kSqrt3 = std::sqrt(3.0);
  auto dist2 = [&](int x, int y, int z) {  // squared straight-line distance to the goal in cells: the tie-breaker
    return (x - t[0]) * (x - t[0]) + (y - t[1]) * (y - t[1]) + (z - t[2]) * (z - t[2]);
  };
  auto heur = [&](int x, int y, int z) {
    int a = std::abs(x - t[0]), b = std::abs(y - t[1]), c = std::abs(z - t[2]);
    if (a < b) std::swap(a, b);
    if (b < c) std::swap(b, c);
    if (a < b) std::swap(a, b);
    return (double)c * kSqrt3 + (double)(b - c) * kSqrt2 + (double)(a - b);
  };
  gval[sid] = 0;
  {
    const double f = 0.0 + heur(s[0], s[1], s[2]);
    if (f >= 2040.0) return false;  // the key range of the device search (fh_map_plan_batch reports -2)
    open.push({(int)(f * kKeyScale), dist2(s[0], s[1], s[2]), sid});
  }
  bool found = false;
  while (!open.empty()) {
    const Node cur = open.top();
    open.pop();
    if (closed[cur.id]) continue;  // a stale duplicate: the cell was expanded from a better entry
    closed[cur.id] = 1;
    if (cur.id == tid) { found = true; break; }
    if (expansions) ++*expansions;
    const double g = gval[cur.id];
    const int cz = cur.id / (grid.nx * grid.ny), rem = cur.id - cz * grid.nx * grid.ny, cy = rem / grid.nx, cx = rem - cy * grid.nx;
    for (int dx = -1; dx <= 1; dx++)
      for (int dy = -1; dy <= 1; dy++)
        for (int dz = -1; dz <= 1; dz++) {
          if (!dx && !dy && !dz) continue;
          const int x = cx + dx, y = cy + dy, z = cz + dz;
          if (!grid.is_free(x, y, z)) continue;
          const int id = grid.index(x, y, z);
          if (closed[id]) continue;
          const double ng = g + std::sqrt((double)(dx * dx + dy * dy + dz * dz));
          if (ng < gval[id]) {
            gval[id] = ng;
            parent[id] = cur.id;
            const double f = ng + heur(x, y, z);
            if (f >= 2040.0) return false;
            open.push({(int)(f * kKeyScale), dist2(x, y, z), id});
          }
        }
  }
  if (!found) return false;
  std::vector<V3> raw;
  for (int id = tid; id >= 0; id = parent[id]) {
    const int cz = id / (grid.nx * grid.ny), rem = id - cz * grid.nx * grid.ny, cy = rem / grid.nx, cx = rem - cy * grid.nx;
    raw.push_back(grid.cell_center(cx, cy, cz));
    if (id == sid) break;
  }
  std::reverse(raw.begin(), raw.end());
  std::vector<V3> p = remove_corner_points(grid, remove_line_points(raw));
  std::reverse(p.begin(), p.end());
  p = remove_corner_points(grid, p);
  std::reverse(p.begin(), p.end());
  if (p.size() > 1) {  // jps_manager.cpp:175-186: ends forced onto the requested points
    p.front() = start;
    p.back() = goal;
  } else {
    p.clear();
    p.push_back(start);
    p.push_back(goal);
  }
  path = p;
  return true;
}

// =====================================================================================================================
// Jump point search with jps3d's own expansion order (what FASTER runs: planner_ptr_->plan(start, goal, 1, true),
// faster/src/jps_manager.cpp:166 -> thirdparty/jps3d/src/jps_planner/graph_search.cpp:123-470).
//
// plan_path() above finds AN optimal path with a total order of its own; jps3d finds the optimal path that its pruning rules, the
// order in which it generates successors, its tolerance comparator (graph_search.h:19-29: f within 1e-6 => smaller g first) and
// the sift discipline of its binary heap select — FASTER's corridors are built around THAT path.  plan_path_jps() reproduces all
// four, so that the cleaned vertex list is jps3d's (checked vertex for vertex against the reference's own compiled sources in
// tests/test_ref_frontend.py).  The neighbour tables are generated from the geometric rules below, not tabulated; the test compares
// them entry by entry with the reference's JPS3DNeib.
namespace {

struct JpsSearch {
  const VoxelGrid& grid;
  const JpsTables& T;
  int gx, gy, gz;  // goal cell
  std::vector<double> g, h;
  std::vector<int> parent, heap_pos;
  std::vector<signed char> dir;  // [3 * id]
  std::vector<char> seen, opened, closed;
  std::vector<int> heap;

  JpsSearch(const VoxelGrid& gr) : grid(gr), T(jps_tables()) {
    const size_t n = (size_t)gr.nx * gr.ny * gr.nz;
    g.assign(n, std::numeric_limits<double>::infinity());
    h.assign(n, 0.0);
    parent.assign(n, -1);
    heap_pos.assign(n, -1);
    dir.assign(3 * n, 0);
    seen.assign(n, 0); opened.assign(n, 0); closed.assign(n, 0);
  }
  bool is_free(int x, int y, int z) const { return grid.is_free(x, y, z); }
  bool is_occupied(int x, int y, int z) const {  // inside the map AND not free (graph_search.cpp:63-66): outside is neither
    return !grid.outside(x, y, z) && grid.occ[grid.index(x, y, z)] > 0;
  }
  double heur(int x, int y, int z) const {  // graph_search.cpp:72-74 (eps = 1)
    return std::sqrt((double)((x - gx) * (x - gx) + (y - gy) * (y - gy) + (z - gz) * (z - gz)));
  }
  // ---- the binary heap of boost::heap::d_ary_heap<arity 2, mutable> with jps3d's comparator ----
  bool lower(int a, int b) const {  // compare_state: true = a has LOWER priority than b
    const double fa = g[a] + h[a], fb = g[b] + h[b];
    if (fa >= fb - 0.000001 && fa <= fb + 0.000001) return g[a] < g[b];
    return fa > fb;
  }
  void swap_nodes(int i, int j) {
    std::swap(heap[i], heap[j]);
    heap_pos[heap[i]] = i;
    heap_pos[heap[j]] = j;
  }
  void sift_up(int i) {
    while (i != 0) {
      const int p = (i - 1) / 2;
      if (!lower(heap[p], heap[i])) return;
      swap_nodes(p, i);
      i = p;
    }
  }
  void sift_down(int i) {
    for (;;) {
      const int first = 2 * i + 1;
      if (first >= (int)heap.size()) return;
      int best = first;
      if (first + 1 < (int)heap.size() && lower(heap[best], heap[first + 1])) best = first + 1;
      if (lower(heap[best], heap[i])) return;
      swap_nodes(best, i);
      i = best;
    }
  }
  void push(int id) {
    heap.push_back(id);
    heap_pos[id] = (int)heap.size() - 1;
    sift_up((int)heap.size() - 1);
  }
  int pop() {
    const int top = heap.front();
    swap_nodes(0, (int)heap.size() - 1);
    heap.pop_back();
    heap_pos[top] = -1;
    if (!heap.empty()) sift_down(0);
    return top;
  }
  bool has_forced(int x, int y, int z, int dx, int dy, int dz) const {  // graph_search.cpp:417-470: 8 / 8 / 6 cells by |d|_1
    const int norm1 = std::abs(dx) + std::abs(dy) + std::abs(dz), id = (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1);
    const int count = norm1 == 3 ? 6 : (norm1 >= 1 ? 8 : 0);
    for (int fn = 0; fn < count; fn++)
      if (is_occupied(x + T.f1[id][fn][0], y + T.f1[id][fn][1], z + T.f1[id][fn][2])) return true;
    return false;
  }
  // graph_search.cpp:374-400.  (Iterative along the move itself — the reference's tail call — recursive into the lower-dimensional moves.)
  bool jump(int x, int y, int z, int dx, int dy, int dz, int& nx, int& ny, int& nz) const {
    const int id = (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1), norm1 = std::abs(dx) + std::abs(dy) + std::abs(dz);
    const int num = kJpsCount[norm1][0];
    nx = x; ny = y; nz = z;
    for (;;) {
      nx += dx; ny += dy; nz += dz;
      if (!is_free(nx, ny, nz)) return false;
      if (nx == gx && ny == gy && nz == gz) return true;
      if (has_forced(nx, ny, nz, dx, dy, dz)) return true;
      for (int k = 0; k < num - 1; k++) {
        int ax, ay, az;
        if (jump(nx, ny, nz, T.ns[id][k][0], T.ns[id][k][1], T.ns[id][k][2], ax, ay, az)) return true;
      }
    }
  }
};

}  // namespace

bool plan_path_jps(VoxelGrid& grid, const V3& start_in, const V3& goal_in, double inflation, std::vector<V3>& path, long long* expansions,
                   double* raw_cost) {
  path.clear();
  const V3 start(start_in.x, start_in.y, std::max(start_in.z, 0.0)), goal(goal_in.x, goal_in.y, std::max(goal_in.z, 0.0));
  int s[3], t[3];
  grid.to_cell(start, s);
  grid.to_cell(goal, t);
  grid.set_free_around(s, inflation);  // jps_manager.cpp:161-162
  grid.set_free_around(t, inflation);
  if (!grid.is_free(s[0], s[1], s[2]) || !grid.is_free(t[0], t[1], t[2])) return false;  // jps_planner.cpp:214-239
  JpsSearch js(grid);
  js.gx = t[0]; js.gy = t[1]; js.gz = t[2];
  const JpsTables& T = js.T;
  const int sid = grid.index(s[0], s[1], s[2]), tid = grid.index(t[0], t[1], t[2]);
  const int nxy = grid.nx * grid.ny;
  js.g[sid] = 0.0;
  js.h[sid] = js.heur(s[0], s[1], s[2]);
  js.seen[sid] = 1;
  js.push(sid);
  js.opened[sid] = 1;
  int cur;
  for (;;) {  // graph_search.cpp:123-217
    if (expansions) ++*expansions;
    cur = js.pop();
    js.closed[cur] = 1;
    if (cur == tid) break;
    const int cz = cur / nxy, rem = cur - cz * nxy, cy = rem / grid.nx, cx = rem - cy * grid.nx;
    const int dx0 = js.dir[3 * cur], dy0 = js.dir[3 * cur + 1], dz0 = js.dir[3 * cur + 2];
    const int norm1 = std::abs(dx0) + std::abs(dy0) + std::abs(dz0), id = (dx0 + 1) + 3 * (dy0 + 1) + 9 * (dz0 + 1);
    const int num_neib = kJpsCount[norm1][0], num_fneib = kJpsCount[norm1][1];
    for (int dev = 0; dev < num_neib + num_fneib; dev++) {  // getJpsSucc, :318-368, successor by successor
      int nx, ny, nz, dx, dy, dz;
      if (dev < num_neib) {
        dx = T.ns[id][dev][0]; dy = T.ns[id][dev][1]; dz = T.ns[id][dev][2];
        if (!js.jump(cx, cy, cz, dx, dy, dz, nx, ny, nz)) continue;
      } else {
        const int k = dev - num_neib;
        if (!js.is_occupied(cx + T.f1[id][k][0], cy + T.f1[id][k][1], cz + T.f1[id][k][2])) continue;
        dx = T.f2[id][k][0]; dy = T.f2[id][k][1]; dz = T.f2[id][k][2];
        if (!js.jump(cx, cy, cz, dx, dy, dz, nx, ny, nz)) continue;
      }
      const int nid = grid.index(nx, ny, nz);
      if (!js.seen[nid]) {
        js.seen[nid] = 1;
        js.dir[3 * nid] = (signed char)dx; js.dir[3 * nid + 1] = (signed char)dy; js.dir[3 * nid + 2] = (signed char)dz;
        js.h[nid] = js.heur(nx, ny, nz);
      }
      const double cost = std::sqrt((double)((nx - cx) * (nx - cx) + (ny - cy) * (ny - cy) + (nz - cz) * (nz - cz)));
      const double tentative = js.g[cur] + cost;  // :150-191
      if (tentative < js.g[nid]) {
        js.parent[nid] = cur;
        js.g[nid] = tentative;
        if (js.opened[nid] && !js.closed[nid]) {
          js.sift_up(js.heap_pos[nid]);  // pq_.increase
          auto sgn = [](int v) { return v > 0 ? 1 : (v < 0 ? -1 : 0); };
          js.dir[3 * nid] = (signed char)sgn(nx - cx); js.dir[3 * nid + 1] = (signed char)sgn(ny - cy); js.dir[3 * nid + 2] = (signed char)sgn(nz - cz);
        } else if (!js.opened[nid]) {
          js.push(nid);
          js.opened[nid] = 1;
        }  // (opened and closed: the reference prints "ASTAR ERROR!" and goes on)
      }
    }
    if (js.heap.empty()) return false;
  }
  std::vector<V3> raw;
  double cost = 0.0;
  for (int id = tid;; id = js.parent[id]) {
    const int cz = id / nxy, rem = id - cz * nxy, cy = rem / grid.nx, cx = rem - cy * grid.nx;
    raw.push_back(grid.cell_center(cx, cy, cz));
    if (id == sid || js.parent[id] < 0) break;
  }
  std::reverse(raw.begin(), raw.end());
  for (size_t i = 1; i < raw.size(); i++) cost += (raw[i] - raw[i - 1]).norm();
  if (raw_cost) *raw_cost = cost;
  std::vector<V3> p = remove_corner_points(grid, remove_line_points(raw));  // jps_planner.cpp:283-291
  std::reverse(p.begin(), p.end());
  p = remove_corner_points(grid, p);
  std::reverse(p.begin(), p.end());
  if (p.size() > 1) {  // jps_manager.cpp:175-186: ends forced onto the requested points
    p.front() = start;
    p.back() = goal;
  } else {
    p.clear();
    p.push_back(start);
    p.push_back(goal);
  }
  path = p;
  return true;
}

void jps_neighbour_tables(int* ns, int* f1, int* f2) {  // [27][3][26], [27][3][12], [27][3][12]: the layout of jps3d's JPS3DNeib
  const JpsTables& T = jps_tables();
  for (int id = 0; id < 27; id++)
    for (int a = 0; a < 3; a++) {
      for (int k = 0; k < 26; k++) ns[(id * 3 + a) * 26 + k] = T.ns[id][k][a];
      for (int k = 0; k < 12; k++) { f1[(id * 3 + a) * 12 + k] = T.f1[id][k][a]; f2[(id * 3 + a) * 12 + k] = T.f2[id][k][a]; }
    }
}

}  // namespace fhfront

// ---- C wrapper (plain pointers) for tests and the Python workload generator -------------------------------------------
// Which search the batch entry points below run: 0 (default) plan_path — A* with a total order, what the device search
// (csrc/fh_path.hip.hpp) reproduces bit for bit; 1 plan_path_jps — jump point search in jps3d's own order, FASTER's exact path.
static int g_search_mode = 0;
static double g_sphere_ra = 0.0;  // ff_set_sphere: clip every path to JPS_in (Faster::replan, faster.cpp:370-382) before the vertex refinement
static bool run_search(fhfront::VoxelGrid& g, const fhfront::V3& s, const fhfront::V3& t, double inflation, std::vector<fhfront::V3>& path,
                       long long* expansions = nullptr) {
  return g_search_mode == 1 ? fhfront::plan_path_jps(g, s, t, inflation, path, expansions) : fhfront::plan_path(g, s, t, inflation, path, expansions);
}

extern "C" {

int ff_set_sphere(double ra) {
  g_sphere_ra = ra > 0.0 ? ra : 0.0;
  return 0;
}

int ff_set_search_mode(int mode) {
  if (mode != 0 && mode != 1) return -1;
  g_search_mode = mode;
  return 0;
}

// Decomposition of one path: writes the rows [a_x a_y a_z b] of polytope i to faces[face_off[i] .. face_off[i+1]).
// Returns the total number of faces, or -1 if max_faces is too small.  ellipsoids (optional): per segment 15 doubles
// (R row major 9, axes 3, centre 3).
int ff_decompose(const double* path_xyz, int n_points, const double* cloud_xyz, int n_cloud, const double* local_bbox,
                 double drone_radius, double z_ground, double* faces, int max_faces, int* face_off, double* ellipsoids) {
  using namespace fhfront;
  std::vector<V3> path, cloud;
  for (int i = 0; i < n_points; i++) path.push_back(V3(path_xyz[3 * i], path_xyz[3 * i + 1], path_xyz[3 * i + 2]));
  for (int i = 0; i < n_cloud; i++) cloud.push_back(V3(cloud_xyz[3 * i], cloud_xyz[3 * i + 1], cloud_xyz[3 * i + 2]));
  std::vector<Ellipsoid> ell;
  const std::vector<LinearConstraint> cs =
      decompose_path(path, cloud, drone_radius, z_ground, V3(local_bbox[0], local_bbox[1], local_bbox[2]), &ell);
  int total = 0;
  face_off[0] = 0;
  for (size_t i = 0; i < cs.size(); i++) {
    for (size_t f = 0; f < cs[i].faces(); f++) {
      if (total >= max_faces) return -1;
      faces[4 * total + 0] = cs[i].A[3 * f + 0];
      faces[4 * total + 1] = cs[i].A[3 * f + 1];
      faces[4 * total + 2] = cs[i].A[3 * f + 2];
      faces[4 * total + 3] = cs[i].b[f];
      total++;
    }
    face_off[i + 1] = total;
    if (ellipsoids) {
      double* e = ellipsoids + 15 * i;
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) e[3 * r + c] = ell[i].R.m[r][c];
      e[9] = ell[i].axes.x; e[10] = ell[i].axes.y; e[11] = ell[i].axes.z;
      e[12] = ell[i].d.x; e[13] = ell[i].d.y; e[14] = ell[i].d.z;
    }
  }
  return total;
}

// Voxel grid from a point cloud (FASTER's readMap) + path search.  Returns the number of path vertices (0 = no path),
// or -1 if max_points is too small.
int ff_plan(const double* cloud_xyz, int n_cloud, int cells_x, int cells_y, int cells_z, double res, const double* center,
            double z_ground, double z_max, double inflation, const double* start, const double* goal, double* path_xyz,
            int max_points) {
  using namespace fhfront;
  std::vector<V3> cloud;
  for (int i = 0; i < n_cloud; i++) cloud.push_back(V3(cloud_xyz[3 * i], cloud_xyz[3 * i + 1], cloud_xyz[3 * i + 2]));
  VoxelGrid g;
  g.build(cloud, cells_x, cells_y, cells_z, res, V3(center[0], center[1], center[2]), z_ground, z_max, inflation);
  std::vector<V3> path;
  if (!plan_path(g, V3(start[0], start[1], start[2]), V3(goal[0], goal[1], goal[2]), inflation, path)) return 0;
  if ((int)path.size() > max_points) return -1;
  for (size_t i = 0; i < path.size(); i++) {
    path_xyz[3 * i] = path[i].x; path_xyz[3 * i + 1] = path[i].y; path_xyz[3 * i + 2] = path[i].z;
  }
  return (int)path.size();
}


// jump point search with jps3d's expansion order (plan_path_jps): same arguments as ff_plan; raw_cost (optional): length of the raw
// cell path in metres
int ff_plan_jps(const double* cloud_xyz, int n_cloud, int cells_x, int cells_y, int cells_z, double res, const double* center,
                double z_ground, double z_max, double inflation, const double* start, const double* goal, double* path_xyz,
                int max_points, double* raw_cost, long long* expansions) {
  using namespace fhfront;
  std::vector<V3> cloud;
  for (int i = 0; i < n_cloud; i++) cloud.push_back(V3(cloud_xyz[3 * i], cloud_xyz[3 * i + 1], cloud_xyz[3 * i + 2]));
  VoxelGrid g;
  g.build(cloud, cells_x, cells_y, cells_z, res, V3(center[0], center[1], center[2]), z_ground, z_max, inflation);
  std::vector<V3> path;
  if (expansions) *expansions = 0;
  if (!plan_path_jps(g, V3(start[0], start[1], start[2]), V3(goal[0], goal[1], goal[2]), inflation, path, expansions, raw_cost)) return 0;
  if ((int)path.size() > max_points) return -1;
  for (size_t i = 0; i < path.size(); i++) {
    path_xyz[3 * i] = path[i].x; path_xyz[3 * i + 1] = path[i].y; path_xyz[3 * i + 2] = path[i].z;
  }
  return (int)path.size();
}
void ff_jps_tables(int* ns, int* f1, int* f2) { fhfront::jps_neighbour_tables(ns, f1, f2); }

// Faster::createMoreVertexes (faster/src/faster.cpp:80-97: legs longer than `max_vertex_dist` are cut, each new vertex one
// spacing beyond the previous one) and deleteVertexes (faster/src/utils.cpp:1117-1124: at most max_poly legs kept).
// createMoreVertexes repeats the end point when a leg is an exact multiple of the spacing; the reference would then decompose a
// zero-length segment — dropped here.
static void refine_vertices(std::vector<fhfront::V3>& path, double max_vertex_dist, int max_poly) {
  using fhfront::V3;
  if (max_vertex_dist > 0.0) {
    for (size_t j = 0; j + 1 < path.size(); j++) {
      const double dist = (path[j + 1] - path[j]).norm();
      const int add = (int)std::floor(dist / max_vertex_dist);
      if (dist > max_vertex_dist) {
        const V3 v = (path[j + 1] - path[j]).normalized();
        for (int k = 0; k < add; k++) {
          path.insert(path.begin() + j + 1, path[j] + v * max_vertex_dist);
          j++;
        }
      }
    }
    for (size_t j = 0; j + 1 < path.size();) {
      if ((path[j + 1] - path[j]).norm() < 1e-9) path.erase(path.begin() + j + 1);
      else j++;
    }
  }
  if (max_poly > 0 && (int)path.size() > max_poly + 1) path.resize(max_poly + 1);
}

// n start/goal queries over one map (OpenMP over queries): the CPU counterpart of fh_map_plan_batch (include/fasterhip.h), same
// arguments and outputs; occ (may be NULL) receives the grid, dims/origin its geometry.
int ff_plan_batch(const double* cloud_xyz, int n_cloud, int cells_x, int cells_y, int cells_z, double res, const double* center,
                  double z_ground, double z_max, double inflation, const double* starts, const double* goals, int n, int max_points,
                  double max_vertex_dist, int max_poly, double* paths, int* n_points, long long* expansions, signed char* occ, int* dims,
                  double* origin) {
  using namespace fhfront;
  std::vector<V3> cloud;
  for (int i = 0; i < n_cloud; i++) cloud.push_back(V3(cloud_xyz[3 * i], cloud_xyz[3 * i + 1], cloud_xyz[3 * i + 2]));
  VoxelGrid base;
  base.build(cloud, cells_x, cells_y, cells_z, res, V3(center[0], center[1], center[2]), z_ground, z_max, inflation);
  if (dims) { dims[0] = base.nx; dims[1] = base.ny; dims[2] = base.nz; }
  if (origin) { origin[0] = base.origin[0]; origin[1] = base.origin[1]; origin[2] = base.origin[2]; }
  if (occ) std::memcpy(occ, base.occ.data(), base.occ.size());
#pragma omp parallel for schedule(dynamic, 8)
  for (int i = 0; i < n; i++) {
    VoxelGrid g = base;  // the search frees the cells around start and goal
    std::vector<V3> path;
    long long ex = 0;
    const bool ok = run_search(g, V3(starts[3 * i], starts[3 * i + 1], starts[3 * i + 2]), V3(goals[3 * i], goals[3 * i + 1], goals[3 * i + 2]),
                               inflation, path, &ex);
    if (expansions) expansions[i] = ex;
    n_points[i] = 0;
    if (!ok) continue;
    // the clean-up walks past max_points like the device: deleteVertexes may bring an over-long list back under the limit
    fhfront::clip_to_sphere(path, g_sphere_ra);
    refine_vertices(path, max_vertex_dist, max_poly);
    if ((int)path.size() > max_points) { n_points[i] = -1; continue; }
    n_points[i] = (int)path.size();
    for (size_t j = 0; j < path.size(); j++) {
      double* o = paths + 3 * ((size_t)i * max_points + j);
      o[0] = path[j].x; o[1] = path[j].y; o[2] = path[j].z;
    }
  }
  return 0;
}

// Batch corridor generation for Monte-Carlo workloads (BASELINE config 5): one shared cloud, n start/goal pairs.
// Per pair: voxel path search -> Faster::createMoreVertexes (faster/src/faster.cpp:80-97, segments longer than
// `max_vertex_dist` are cut) -> deleteVertexes (faster/src/utils.cpp:1117-1124, at most max_poly segments kept) ->
// decomposition.  Outputs per pair i: n_poly[i] (0 = no path), face_off[i][0..8], faces at faces[i*faces_per_problem ...],
// goal[i] = last kept vertex (the solver's E, faster.cpp:393-394).  OpenMP over pairs.
int ff_corridor_batch(const double* cloud_xyz, int n_cloud, int cells_x, int cells_y, int cells_z, double res, const double* center,
                      double z_ground, double z_max, double inflation, double drone_radius, const double* starts,
                      const double* goals, int n, int max_poly, double max_vertex_dist, int faces_per_problem, double* faces,
                      int* face_off, int* n_poly, double* goal_out) {
  using namespace fhfront;
  std::vector<V3> cloud;
  for (int i = 0; i < n_cloud; i++) cloud.push_back(V3(cloud_xyz[3 * i], cloud_xyz[3 * i + 1], cloud_xyz[3 * i + 2]));
  VoxelGrid base;
  base.build(cloud, cells_x, cells_y, cells_z, res, V3(center[0], center[1], center[2]), z_ground, z_max, inflation);
  int overflow = 0;
#pragma omp parallel for schedule(dynamic, 8)
  for (int i = 0; i < n; i++) {
    VoxelGrid g = base;  // the search frees the cells around start and goal
    n_poly[i] = 0;
    for (int k = 0; k < 9; k++) face_off[9 * i + k] = 0;
    std::vector<V3> path;
    const V3 s(starts[3 * i], starts[3 * i + 1], starts[3 * i + 2]), t(goals[3 * i], goals[3 * i + 1], goals[3 * i + 2]);
    goal_out[3 * i] = t.x; goal_out[3 * i + 1] = t.y; goal_out[3 * i + 2] = t.z;
    if (!run_search(g, s, t, inflation, path)) continue;
    fhfront::clip_to_sphere(path, g_sphere_ra);
    refine_vertices(path, max_vertex_dist, max_poly);
    const std::vector<LinearConstraint> cs = decompose_path(path, cloud, drone_radius, z_ground);
    int total = 0;
    bool fits = true;
    for (size_t p = 0; p < cs.size() && fits; p++) {
      for (size_t f = 0; f < cs[p].faces(); f++) {
        if (total >= faces_per_problem) { fits = false; break; }
        double* row = faces + 4 * ((size_t)i * faces_per_problem + total);
        row[0] = cs[p].A[3 * f]; row[1] = cs[p].A[3 * f + 1]; row[2] = cs[p].A[3 * f + 2]; row[3] = cs[p].b[f];
        total++;
      }
      face_off[9 * i + p + 1] = total;
    }
    if (!fits) {
#pragma omp atomic
      overflow++;
      continue;
    }
    for (size_t p = cs.size(); p < 8; p++) face_off[9 * i + p + 1] = total;
    n_poly[i] = (int)cs.size();
    goal_out[3 * i] = path.back().x; goal_out[3 * i + 1] = path.back().y; goal_out[3 * i + 2] = path.back().z;
  }
  return overflow;
}

}  // extern "C"
