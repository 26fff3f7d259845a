// corridor_frontend.hpp — the producer of the solver's corridors, host C++14, no dependencies.
//
// Restates, without Eigen / PCL / ROS, the two steps FASTER runs right before the trajectory solver:
//   * path search on a voxel grid  ........ JPS_Manager::solveJPS3D (faster/src/jps_manager.cpp:141-196) on top of
//       jps3d's MapUtil (thirdparty/jps3d/include/jps_collision/map_util.h:30-185 readMap, :334-382 cell conventions and
//       ray test) and JPSPlanner::plan (src/jps_planner/jps_planner.cpp:195-295: search, removeLinePts :83-105,
//       removeCornerPts :36-81).  The graph search here is A* with jps3d's costs and heuristic
//       (graph_search.cpp:73-75; jps3d itself offers A* via plan(..., use_jps = false)): same optimal cost as jump point
//       search, possibly another path among equal-cost ones.
//   * convex decomposition around the path . JPS_Manager::cvxEllipsoidDecomp (jps_manager.cpp:80-127) =
//       EllipsoidDecomp3D::dilate (thirdparty/DecompROS/DecompUtil/include/decomp_util/ellipsoid_decomp.h:95-120) over
//       LineSegment3D::dilate (line_segment.h:34-39): obstacle inflation + ellipsoid fit (:156-252, FASTER's inflation
//       :178-190), separating planes (decomp_base.h:83-115, ellipsoid.h:48-73), local bounding box (:57-98), conversion to
//       A x <= b around the segment midpoint (decomp_geometry/polyhedron.h:131-152) and the ground plane (:113-124).
// Everything is double precision on the host; a batched GPU version of the decomposition is a later row.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace fhfront {

struct V3 {
  double x, y, z;
  V3() : x(0), y(0), z(0) {}
  V3(double a, double b, double c) : x(a), y(b), z(c) {}
  V3 operator+(const V3& o) const { return V3(x + o.x, y + o.y, z + o.z); }
  V3 operator-(const V3& o) const { return V3(x - o.x, y - o.y, z - o.z); }
  V3 operator*(double s) const { return V3(x * s, y * s, z * s); }
  V3 operator-() const { return V3(-x, -y, -z); }
  double dot(const V3& o) const { return x * o.x + y * o.y + z * o.z; }
  V3 cross(const V3& o) const { return V3(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x); }
  double norm() const { return std::sqrt(dot(*this)); }
  V3 normalized() const { const double n = norm(); return V3(x / n, y / n, z / n); }
};

struct M3 {  // row major
  double m[3][3];
  static M3 diag(double a, double b, double c) { M3 r{}; r.m[0][0] = a; r.m[1][1] = b; r.m[2][2] = c; return r; }
  M3 operator*(const M3& o) const {
    M3 r{};
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        for (int k = 0; k < 3; k++) r.m[i][j] += m[i][k] * o.m[k][j];
    return r;
  }
  V3 operator*(const V3& v) const {
    return V3(m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z, m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
              m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z);
  }
  M3 transposed() const {
    M3 r{};
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) r.m[i][j] = m[j][i];
    return r;
  }
};

struct Plane {  // { x : n.(x - p) <= 0 } is the kept side
  V3 p, n;
  double signed_dist(const V3& q) const { return n.dot(q - p); }
};

struct Ellipsoid {  // { d + C u : |u| <= 1 },  C = R diag(axes) R^T
  M3 R;
  V3 axes, d;
  // |C^{-1}(q - d)|
  double dist(const V3& q) const {
    const V3 l = R.transposed() * (q - d);
    return V3(l.x / axes.x, l.y / axes.y, l.z / axes.z).norm();
  }
  // C^{-1} C^{-T} (q - d): outward normal (not normalised) of the level set through q
  V3 gradient(const V3& q) const {
    const V3 l = R.transposed() * (q - d);
    return R * V3(l.x / (axes.x * axes.x), l.y / (axes.y * axes.y), l.z / (axes.z * axes.z));
  }
};

struct Polytope {
  std::vector<Plane> planes;
};

struct LinearConstraint {  // rows a.x <= b
  std::vector<double> A;   // [F][3]
  std::vector<double> b;   // [F]
  size_t faces() const { return b.size(); }
};

constexpr double kEps = 1e-10;  // DecompUtil's epsilon_ (decomp_basis/data_type.h:129)

// rotation taking the x axis onto v with zero roll: Rz(yaw) * Ry(pitch)   (geometric_utils.h:27-35).  The reference goes through
// atan2 / cos / sin; here the cosines and sines are taken straight from the components (cos(atan2(y, x)) = x / hypot): the same
// rotation, and only +, *, /, sqrt are involved — all correctly rounded on the host and on the device, so the device kernel
// (csrc/fh_decomp.hip.hpp) reproduces this decomposition bit for bit instead of to a libm's last digit (which decides on which side
// FASTER's inflation pushes a point that lies in the plane of the segment).
inline M3 rotation_onto(const V3& v) {
  const double hxy = std::sqrt(v.x * v.x + v.y * v.y), n3 = std::sqrt(hxy * hxy + v.z * v.z);
  const double cp = n3 > 0 ? hxy / n3 : 1.0, sp = n3 > 0 ? -v.z / n3 : 0.0;
  const double cy = hxy > 0 ? v.x / hxy : 1.0, sy = hxy > 0 ? v.y / hxy : 0.0;
  M3 r{};
  r.m[0][0] = cy * cp; r.m[0][1] = -sy; r.m[0][2] = cy * sp;
  r.m[1][0] = sy * cp; r.m[1][1] = cy;  r.m[1][2] = sy * sp;
  r.m[2][0] = -sp;     r.m[2][1] = 0;   r.m[2][2] = cp;
  return r;
}
// Rx(atan2(z, y))
inline M3 roll_about_x(double z, double y) {
  const double h = std::sqrt(y * y + z * z);
  const double c = h > 0 ? y / h : 1.0, s = h > 0 ? z / h : 0.0;
  M3 r{};
  r.m[0][0] = 1; r.m[1][1] = c; r.m[1][2] = -s; r.m[2][1] = s; r.m[2][2] = c;
  return r;
}

// ---- convex decomposition around one path segment -------------------------------------------------------------------
class SegmentDecomposer {
public:
  SegmentDecomposer(const V3& p1, const V3& p2, const V3& local_bbox, double inflate) : p1_(p1), p2_(p2), bbox_(local_bbox), inflate_(inflate) {}

  // obstacle points of interest: those inside the local bounding box (decomp_base.h:39-46)
  void set_obstacles(const std::vector<V3>& cloud) {
    Polytope box;
    add_local_bbox(box);
    obs_.clear();
    for (const V3& q : cloud) {
      bool in = true;
      for (const Plane& pl : box.planes)
        if (pl.signed_dist(q) > kEps) { in = false; break; }
      if (in) obs_.push_back(q);
    }
  }

  void run(Ellipsoid& ell_out, Polytope& poly_out) {
    fit_ellipsoid();
    separate();
    add_local_bbox(poly_);
    ell_out = ell_;
    poly_out = poly_;
  }

private:
  static int sgn(double v) { return (0.0 < v) - (v < 0.0); }

  void add_local_bbox(Polytope& P) const {  // line_segment.h:57-98 (plane order kept)
    if (bbox_.norm() == 0) return;
    const V3 dir = (p2_ - p1_).normalized();
    V3 dir_h(dir.y, -dir.x, 0.0);
    if (dir_h.norm() == 0) dir_h = V3(-1, 0, 0);
    dir_h = dir_h.normalized();
    P.planes.push_back({p1_ + dir_h * bbox_.y, dir_h});
    P.planes.push_back({p1_ - dir_h * bbox_.y, -dir_h});
    P.planes.push_back({p2_ + dir * bbox_.x, dir});
    P.planes.push_back({p1_ - dir * bbox_.x, -dir});
    const V3 dir_v = dir.cross(dir_h);
    P.planes.push_back({p1_ + dir_v * bbox_.z, dir_v});
    P.planes.push_back({p1_ - dir_v * bbox_.z, -dir_v});
  }

  const V3* closest(const std::vector<V3>& pts) const {
    const V3* best = pts.empty() ? nullptr : &pts[0];  // (never null for a non-empty set, even if every distance is NaN)
    double bd = 1e300;
    for (const V3& q : pts) {
      const double dd = ell_.dist(q);
      if (dd < bd) { bd = dd; best = &q; }
    }
    return best;
  }

  // line_segment.h:156-252: sphere on the segment, obstacles pushed inflate_ towards the centre along the ellipsoid axes
  // (FASTER's addition), then the two short axes are shrunk until no obstacle point is strictly inside.
  void fit_ellipsoid() {
    const double f = (p1_ - p2_).norm() / 2;
    V3 axes(f, f, f);
    const M3 Ri = rotation_onto(p2_ - p1_);
    ell_.R = Ri;
    ell_.axes = axes;
    ell_.d = (p1_ + p2_) * 0.5;
    for (V3& q : obs_) {
      const V3 l = Ri.transposed() * (q - ell_.d);
      q = Ri * V3(l.x - sgn(l.x) * inflate_, l.y - sgn(l.y) * inflate_, l.z - sgn(l.z) * inflate_) + ell_.d;
    }
    std::vector<V3> first;
    for (const V3& q : obs_)
      if (ell_.dist(q) <= 1) first.push_back(q);
    std::vector<V3> inside = first;
    M3 Rf = Ri;
    while (!inside.empty()) {  // second axis (and the roll that puts the closest point in the x-y plane)
      const V3 pw = *closest(inside);
      V3 l = Ri.transposed() * (pw - ell_.d);
      Rf = Ri * roll_about_x(l.z, l.y);
      l = Rf.transposed() * (pw - ell_.d);
      if (l.x < axes.x) axes.y = std::fabs(l.y) / std::sqrt(1 - (l.x / axes.x) * (l.x / axes.x));
      ell_.R = Rf;
      ell_.axes = V3(axes.x, axes.y, axes.y);
      std::vector<V3> keep;
      for (const V3& q : inside)
        if (1 - ell_.dist(q) > kEps) keep.push_back(q);
      inside.swap(keep);
    }
    ell_.R = Rf;
    ell_.axes = axes;  // third axis back to its initial length
    inside.clear();
    for (const V3& q : first)
      if (ell_.dist(q) <= 1) inside.push_back(q);
    while (!inside.empty()) {
      const V3 pw = *closest(inside);
      const V3 l = Rf.transposed() * (pw - ell_.d);
      const double dd = 1 - (l.x / axes.x) * (l.x / axes.x) - (l.y / axes.y) * (l.y / axes.y);
      if (dd > kEps) axes.z = std::fabs(l.z) / std::sqrt(dd);
      ell_.axes = axes;
      std::vector<V3> keep;
      for (const V3& q : inside)
        if (1 - ell_.dist(q) > kEps) keep.push_back(q);
      inside.swap(keep);
    }
  }

  // decomp_base.h:83-115: tangent plane at the closest remaining obstacle point, drop what it cuts off, repeat
  void separate() {
    poly_.planes.clear();
    std::vector<V3> remain = obs_;
    while (!remain.empty()) {
      const V3 cp = *closest(remain);
      const V3 g = ell_.gradient(cp);
      const double gn = g.norm();
      if (!(gn > 0) || !std::isfinite(gn)) break;  // degenerate ellipsoid (zero-length segment): no separating planes
      Plane pl{cp, V3(g.x / gn, g.y / gn, g.z / gn)};
      if (gn > 0) poly_.planes.push_back(pl);  // (the reference skips a zero normal, FASTER's guard :94-99)
      std::vector<V3> keep;
      for (const V3& q : remain)
        if (gn > 0 && pl.signed_dist(q) < 0) keep.push_back(q);
      remain.swap(keep);
    }
  }

  V3 p1_, p2_, bbox_;
  double inflate_;
  std::vector<V3> obs_;
  Ellipsoid ell_;
  Polytope poly_;
};

// A x <= b from planes, oriented so that `inside` satisfies it (polyhedron.h:131-152)
inline LinearConstraint to_constraint(const V3& inside, const Polytope& P) {
  LinearConstraint c;
  for (const Plane& pl : P.planes) {
    V3 n = pl.n;
    double off = pl.p.dot(n);
    if (n.dot(inside) - off > 0) { n = -n; off = -off; }
    c.A.push_back(n.x); c.A.push_back(n.y); c.A.push_back(n.z);
    c.b.push_back(off);
  }
  return c;
}

// JPS_Manager::cvxEllipsoidDecomp (jps_manager.cpp:80-127): one polytope per path segment, local bbox (2,2,1), obstacles
// inflated by the drone radius, ground plane -z <= -z_ground appended.
inline std::vector<LinearConstraint> decompose_path(const std::vector<V3>& path, const std::vector<V3>& cloud, double drone_radius,
                                                     double z_ground, const V3& local_bbox = V3(2, 2, 1),
                                                     std::vector<Ellipsoid>* ellipsoids = nullptr) {
  std::vector<LinearConstraint> out;
  for (size_t i = 0; i + 1 < path.size(); i++) {
    SegmentDecomposer sd(path[i], path[i + 1], local_bbox, drone_radius);
    sd.set_obstacles(cloud);
    Ellipsoid e;
    Polytope P;
    sd.run(e, P);
    LinearConstraint c = to_constraint((path[i] + path[i + 1]) * 0.5, P);
    c.A.push_back(0); c.A.push_back(0); c.A.push_back(-1);
    c.b.push_back(-z_ground);
    out.push_back(c);
    if (ellipsoids) ellipsoids->push_back(e);
  }
  return out;
}

// ---- voxel grid + path search ------------------------------------------------------------------------------------------
struct VoxelGrid {
  int nx = 0, ny = 0, nz = 0;
  double res = 0.1;
  double origin[3] = {0, 0, 0};
  std::vector<int8_t> occ;  // 0 free, 100 occupied

  int index(int x, int y, int z) const { return x + nx * y + nx * ny * z; }
  bool outside(int x, int y, int z) const { return x < 0 || y < 0 || z < 0 || x >= nx || y >= ny || z >= nz; }
  bool is_free(int x, int y, int z) const { return !outside(x, y, z) && occ[index(x, y, z)] == 0; }
  void to_cell(const V3& p, int c[3]) const {  // map_util.h:334-340
    c[0] = (int)std::round((p.x - origin[0]) / res - 0.5);
    c[1] = (int)std::round((p.y - origin[1]) / res - 0.5);
    c[2] = (int)std::round((p.z - origin[2]) / res - 0.5);
  }
  V3 cell_center(int x, int y, int z) const {  // :342-346
    return V3((x + 0.5) * res + origin[0], (y + 0.5) * res + origin[1], (z + 0.5) * res + origin[2]);
  }

  // FASTER's readMap (map_util.h:30-185): grid of cells_* cells (x,y widened by 5*inflation/res) centred on `center`, clipped
  // to [z_ground, z_max]; every point marks its cell and the cube of +-floor(inflation/res) cells around it.
  void build(const std::vector<V3>& cloud, int cells_x, int cells_y, int cells_z, double res_, const V3& center, double z_ground,
             double z_max, double inflation) {
    res = res_;
    int dx = cells_x + (int)(5 * inflation / res), dy = cells_y + (int)(5 * inflation / res), dz = cells_z;
    int down = (int)(dz / 2.0), up = (int)(dz / 2.0);
    if (center.z - res * dz / 2.0 < z_ground) down = std::max((int)((center.z - z_ground) / res), 0);
    if (center.z + res * dz / 2.0 > z_max) {
      up = (int)((z_max - center.z) / res);
      up = up > 0 ? up : 1;
    }
    dz = down + up;
    nx = dx; ny = dy; nz = dz;
    origin[0] = center.x - res * dx / 2.0;
    origin[1] = center.y - res * dy / 2.0;
    origin[2] = center.z - res * down;
    occ.assign((size_t)nx * ny * nz, 0);
    const int total = nx * ny * nz;
    const int m = (int)std::floor(inflation / res);
    for (const V3& p : cloud) {
      int c[3];
      to_cell(p, c);
      for (int k = 0; k < 3; k++) c[k] = c[k] > 0 ? c[k] : 0;
      for (int ix = c[0] - m; ix <= c[0] + m; ix++)
        for (int iy = c[1] - m; iy <= c[1] + m; iy++)
          for (int iz = c[2] - m; iz <= c[2] + m; iz++) {
            const int id = ix + nx * iy + nx * ny * iz;  // flat index test only, as in the reference
            if (id >= 0 && id < total) occ[id] = 100;
          }
    }
  }

  // setFreeVoxelAndSurroundings(center, const float d) (jps3d map_util.h:248-263): n_voxels = round(d / res + 0.5) with d a FLOAT
  // (inflation 0.3, res 0.2: 2 cells — not the floor(inflation / res) = 1 of readMap's inflation)
  void set_free_around(const int c[3], double inflation) {
    const int m = (int)std::round((double)(float)inflation / res + 0.5);
    for (int ix = c[0] - m; ix <= c[0] + m; ix++)
      for (int iy = c[1] - m; iy <= c[1] + m; iy++)
        for (int iz = c[2] - m; iz <= c[2] + m; iz++)
          if (!outside(ix, iy, iz)) occ[index(ix, iy, iz)] = 0;
  }

  // ray test of removeCornerPts: cells sampled every 0.8 cell along the ray (map_util.h:348-382)
  bool blocked(const V3& a, const V3& b) const {
    const V3 diff = b - a;
    const double mx = std::max(std::fabs(diff.x), std::max(std::fabs(diff.y), std::fabs(diff.z))) / res;
    const int steps = (int)(mx / 0.8);
    if (steps <= 0) return false;
    const double s = 1.0 / steps;
    int prev[3] = {-1, -1, -1};
    const V3 step = diff * s;  // rayTrace: step = diff * s, pt = pt1 + step * n — in THIS order: samples of a diagonal ray sit on cell
                               // corners, and the rounding of the product decides the cell
    for (int n = 1; n < steps; n++) {
      const V3 p = a + step * (double)n;
      int c[3];
      to_cell(p, c);
      if (outside(c[0], c[1], c[2])) break;
      if (c[0] != prev[0] || c[1] != prev[1] || c[2] != prev[2]) {
        if (occ[index(c[0], c[1], c[2])] >= 100) return true;
      }
      prev[0] = c[0]; prev[1] = c[1]; prev[2] = c[2];
    }
    return false;
  }
};

// Point where the segment a -> b leaves the sphere (centre c, radius r): getIntersectionWithSphere (faster/src/utils.cpp:713-776) with
// its arithmetic — single precision, except where the language promotes: pow(float, 2) is a double (so the squares are summed in
// double and rounded once) and `- r * r` is a double subtraction.  Tangent / no crossing: the ray centre -> a.
inline V3 sphere_crossing(const V3& a_in, const V3& b_in, double r, const V3& c) {
  auto solve = [&](const V3& A, const V3& B, float& disc) {
    const float x1 = (float)A.x, y1 = (float)A.y, z1 = (float)A.z, x2 = (float)B.x, y2 = (float)B.y, z2 = (float)B.z;
    const float x3 = (float)c.x, y3 = (float)c.y, z3 = (float)c.z;
    const float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
    const float a = (float)((double)dx * (double)dx + (double)dy * (double)dy + (double)dz * (double)dz);
    const float b = 2.0f * (dx * (x1 - x3) + dy * (y1 - y3) + dz * (z1 - z3));
    const float cf = x3 * x3 + y3 * y3 + z3 * z3 + x1 * x1 + y1 * y1 + z1 * z1 - 2.0f * (x3 * x1 + y3 * y1 + z3 * z1);
    const float cc = (float)((double)cf - r * r);
    disc = b * b - 4.0f * a * cc;
    const float t = (-b + std::sqrt(disc)) / (2.0f * a);
    return V3((double)(x1 + dx * t), (double)(y1 + dy * t), (double)(z1 + dz * t));
  };
  float disc;
  const V3 first = solve(a_in, b_in, disc);
  if (disc <= 0) return solve(c, a_in, disc);
  return first;
}
// JPS_in of Faster::replan (faster/src/faster.cpp:370-382): the path up to its first crossing of the sphere of radius
// ra = min(|goal - start| - 0.001, Ra) around its first vertex (getFirstIntersectionWithSphere, utils.cpp:782-870), the crossing
// point E appended; start and goal are the ends of the path (jps_manager.cpp:175-186 forces them there).  Ra <= 0: the path as it is.
inline void clip_to_sphere(std::vector<V3>& path, double Ra) {
  if (!(Ra > 0.0) || path.size() < 2) return;
  const V3 center = path[0];
  const double ra = std::min((path.back() - center).norm() - 0.001, Ra);
  int index = -1;
  for (size_t i = 0; i < path.size(); i++)
    if ((path[i] - center).norm() > ra) { index = (int)i; break; }
  if (index <= 0) return;  // nothing outside (or, impossibly, the first vertex): JPS_in is the whole path
  const V3 E = sphere_crossing(path[index - 1], path[index], ra, center);
  path.resize(index);
  path.push_back(E);
}

// 26-connected A* with Euclidean step costs and heuristic (graph_search.cpp:73-75), then jps3d's path clean-up
// (jps_planner.cpp:286-291).  Returns false if start/goal are not free or no path exists.
bool plan_path(VoxelGrid& grid, const V3& start, const V3& goal, double inflation, std::vector<V3>& path,
               long long* expansions = nullptr);

// Jump point search with jps3d's own expansion order (successor order, tolerance comparator, binary-heap discipline): the path
// FASTER itself would get from planner_ptr_->plan(start, goal, 1, true) (jps_manager.cpp:166).  corridor_frontend.cpp.
bool plan_path_jps(VoxelGrid& grid, const V3& start, const V3& goal, double inflation, std::vector<V3>& path, long long* expansions = nullptr,
                   double* raw_cost = nullptr);
void jps_neighbour_tables(int* ns, int* f1, int* f2);

}  // namespace fhfront
