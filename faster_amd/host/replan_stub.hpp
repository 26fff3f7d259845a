// replan_stub.hpp — a ROS/PCL/Eigen-free stand-in for the CALLER of the hot path.
//
// FASTER's planner loop is Faster::replan (faster/src/faster.cpp:296-595): pick the start state A on the committed plan, search a
// path to the projected goal, clip it to the sphere of radius Ra, build the corridor of the whole trajectory in known-occupied space
// and solve it; find where that trajectory enters unknown space (H), back off to a state R from which the vehicle can still brake,
// build the safe corridor in unknown+occupied space from R and solve the safe trajectory; splice A->R->safe into the plan and adapt
// the two factor windows.  This header restates that control flow so that the solver behind the SolverGurobi surface (SolverHip)
// is exercised exactly the way the reference drives it — same setter order, same public members, same use of X_temp_ and
// factor_that_worked_ — with the ROS node, the mutexes, the visual outputs and the yaw logic left out.
//
// Reference pieces restated here (file:line relative to faster/src/):
//   replan ............................. faster.cpp:296-595        findIndexH / findIndexR ...... faster.cpp:218-251 / :173-216
//   appendToPlan ....................... faster.cpp:606-648        getNextGoal (without yaw) .... faster.cpp:699-723
//   createMoreVertexes ................. faster.cpp:80-97          getFirstCollisionJPS ......... faster.cpp:767-926
//   deleteVertexes / reduceJPSbyDistance utils.cpp:1117-1124 / :690-710
//   getFirstIntersectionWithSphere ..... utils.cpp:782-870 (+ getIntersectionWithSphere :713-776, float arithmetic kept)
//   projectPointToBox .................. utils.cpp:1065-1115
// The kd-tree queries of the reference (pcl::KdTreeFLANN, faster.hpp:139-141) are brute-force nearest-neighbour scans here.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <deque>
#include <limits>
#include <vector>

#include "corridor_frontend.hpp"
#include "faster_stub.hpp"

namespace fhreplan {

using fhfront::V3;

struct Params {  // the planner-relevant subset of `parameters` (faster_types.hpp:17-77), defaults from param/faster.yaml
  double dc = 0.01, goal_radius = 0.3, drone_radius = 0.42;
  double Ra = 4.0, dist_max_vertexes = 1.5;
  int N_whole = 6, N_safe = 6;
  double v_max = 5.0, a_max = 5.0, j_max = 8.0;
  double gamma_whole = 20, gammap_whole = 20, increment_whole = 1.0;
  double gamma_safe = 20, gammap_safe = 20, increment_safe = 1.0;
  int max_poly_whole = 3, max_poly_safe = 3;
  double delta_a = 0.5, delta_H = 1.0;
  int deltaT = 10;  // states between "now" and the start state A (faster.hpp:131)
  double wdx = 20, wdy = 20, wdz = 4, res = 0.15;
  double z_ground = 0.0, z_max = 3.0, inflation_jps = 0.47, factor_jps = 1.0;
  // ---- what a test may pin down (defaults: the behaviour described above) ----
  bool jps = false;               // true: the path search is jump point search in jps3d's own order (plan_path_jps: what FASTER runs,
                                  //       jps_manager.cpp:166); false: the A* with a total order of its own (plan_path: the same cost)
  double decomp_radius = -1.0;    // >= 0: the inflation handed to the decomposition (cvxEllipsoidDecomp's drone radius) when it is to differ
                                  //       from drone_radius of the unknown-space tests; < 0: drone_radius
  bool map_fixed = false;         // true: the occupancy grid is map_cells cells of size res centred at map_center for every replan (a
  double map_center[3] = {0, 0, 0};  //    batch of independent pairs shares one map), not wdx x wdy x wdz around the vehicle
  int map_cells[3] = {0, 0, 0};
};

enum class Status { TRAVELING, GOAL_SEEN, GOAL_REACHED };

inline V3 pos_of(const state& s) { return V3(s.pos.x(), s.pos.y(), s.pos.z()); }

// ---- geometry helpers -------------------------------------------------------------------------------------------------
inline V3 project_to_box(const V3& c, const V3& p, double wx, double wy, double wz) {
  const double lo[3] = {c.x - wx / 2, c.y - wy / 2, c.z - wz / 2}, hi[3] = {c.x + wx / 2, c.y + wy / 2, c.z + wz / 2};
  const double q[3] = {p.x, p.y, p.z}, o[3] = {c.x, c.y, c.z};
  if (q[0] < hi[0] && q[0] > lo[0] && q[1] < hi[1] && q[1] > lo[1] && q[2] < hi[2] && q[2] > lo[2]) return p;
  double best = std::numeric_limits<double>::infinity();
  V3 out = p;
  for (int ax = 0; ax < 3; ax++)
    for (int side = 0; side < 2; side++) {  // intersection of the segment c->p with each face plane; the nearest one wins
      const double plane = side ? lo[ax] : hi[ax], den = q[ax] - o[ax];
      if (den == 0) continue;
      const double t = (plane - o[ax]) / den;
      if (t < 0 || t > 1) continue;
      const V3 x = c + (p - c) * t;
      const double dist = (x - c).norm();
      if (dist < best) { best = dist; out = x; }
    }
  return out;
}

using fhfront::sphere_crossing;  // getIntersectionWithSphere (utils.cpp:713-776): corridor_frontend.hpp

// first point of `path` on the sphere around `center` (utils.cpp:782-870).  last_inside: index of the last vertex inside.
inline V3 sphere_exit(const std::vector<V3>& path, double r, const V3& center, int* last_inside, bool* none_outside) {
  if (none_outside) *none_outside = false;
  int index = -1;
  for (size_t i = 0; i < path.size(); i++)
    if ((path[i] - center).norm() > r) { index = (int)i; break; }
  if (index == -1) {
    if (last_inside) *last_inside = (int)path.size() - 1;
    if (none_outside) *none_outside = true;
    return sphere_crossing(center, path.back(), r, center);
  }
  if (index == 0) {
    if (last_inside) *last_inside = 1;
    return path[0];
  }
  if (last_inside) *last_inside = index - 1;
  return sphere_crossing(path[index - 1], path[index], r, center);
}

inline void subdivide(std::vector<V3>& path, double d) {  // createMoreVertexes
  for (size_t j = 0; j + 1 < path.size(); j++) {
    const double dist = (path[j + 1] - path[j]).norm();
    const int add = (int)std::floor(dist / d);
    if (dist > d) {
      const V3 v = (path[j + 1] - path[j]).normalized();
      for (int k = 0; k < add; k++) {
        path.insert(path.begin() + j + 1, path[j] + v * d);
        j++;
      }
    }
  }
}
inline void keep_first(std::vector<V3>& path, int max_segments) {  // deleteVertexes
  if ((int)path.size() > max_segments + 1) path.resize(max_segments + 1);
}
inline void shorten_by(std::vector<V3>& path, double d) {  // reduceJPSbyDistance
  double acc = 0;
  for (int i = (int)path.size() - 1; i > 0; i--) {
    const V3 v = path[i] - path[i - 1];
    acc += v.norm();
    if (acc > d) {
      const double keep = acc - d;
      path.erase(path.begin() + i, path.end());
      path.push_back(path.back() + v.normalized() * keep);
      break;
    }
  }
}

struct Cloud {
  std::vector<V3> pts;
  bool nearest(const V3& q, double& dist) const {  // brute force in place of the kd-tree
    if (pts.empty()) return false;
    double best = std::numeric_limits<double>::infinity();
    for (const V3& p : pts) {
      const double dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z, d2 = dx * dx + dy * dy + dz * dz;
      if (d2 < best) best = d2;
    }
    dist = std::sqrt(best);
    return true;
  }
};

// getFirstCollisionJPS(path, UNKNOWN_MAP, RETURN_INTERSECTION) (faster.cpp:767-926): march along the path in spheres that are
// known to be clear; if the path comes within drone_radius of `cloud`, cut it there (backed off by drone_radius).
inline V3 march_to_cloud(std::vector<V3>& path, const Cloud& cloud, double drone_radius, bool* hit) {
  std::vector<V3> original = path;
  *hit = false;
  int iteration = 0;
  while (!path.empty()) {
    double r;
    if (!cloud.nearest(path[0], r)) { path = original; return original.back(); }
    if (r < drone_radius) {
      *hit = true;
      if (iteration == 0) {  // already inside at the first vertex: the reference returns a 1 cm stub
        const V3 tmp(original[0].x + 0.01, original[0].y, original[0].z);
        path.clear();
        path.push_back(original[0]);
        path.push_back(tmp);
        return tmp;
      }
      const int eliminated = (int)original.size() - (int)path.size() + 1;
      original.erase(original.begin() + eliminated, original.end());
      original.push_back(path[0]);
      shorten_by(original, drone_radius);
      path = original;
      return original.back();
    }
    bool none_outside = false;
    int last_id = -1;
    const V3 inters = sphere_exit(path, r, path[0], &last_id, &none_outside);
    if (none_outside) { path = original; return original.back(); }
    path.erase(path.begin(), path.begin() + last_id + 1);
    path.insert(path.begin(), inters);
    iteration++;
  }
  path = original;
  return original.back();
}

inline std::vector<LinearConstraint3D> to_solver_constraints(const std::vector<fhfront::LinearConstraint>& cs) {
  std::vector<LinearConstraint3D> out;
  for (const auto& c : cs) {
    fhstub::MatX3 A(c.faces());
    fhstub::VecX b(c.faces());
    for (size_t f = 0; f < c.faces(); f++) {
      A(f, 0) = c.A[3 * f]; A(f, 1) = c.A[3 * f + 1]; A(f, 2) = c.A[3 * f + 2];
      b(f) = c.b[f];
    }
    out.push_back(LinearConstraint3D(A, b));
  }
  return out;
}

struct ReplanLog {
  int stage = 0;  // 0 not started, 1 no path, 2 whole failed, 3 safe failed, 4 append failed, 5 committed
  bool needed_safe = false;
  int k_end_whole = 0, k_safe = 0, index_H = 0;
  double whole_factor = 0, safe_factor = 0;
  size_t n_whole = 0, n_safe = 0;
  std::vector<V3> safe_path;     // JPS_safe: R first (faster.cpp:478-490)
  V3 safe_goal = V3(0, 0, 0);    // what sg_safe_.setXf received: M, or G when G lies in the last polytope (:498-499)
  std::vector<int> safe_rows;    // rows per polytope of the safe corridor
};

// the decomposition policy of Planner when none is given: DecompUtil's algorithm on the host (corridor_frontend.hpp);
// DecompHip (decomp_hip.hpp) is the same step on the device
struct HostDecomposition {
  std::vector<fhfront::LinearConstraint> operator()(const std::vector<V3>& path, const std::vector<V3>& cloud, double drone_radius,
                                                    double z_ground) const {
    return fhfront::decompose_path(path, cloud, drone_radius, z_ground);
  }
};

// Solver: any type with the SolverGurobi surface (SolverHip in the product; an oracle-backed adapter in the CPU tests).
// Decomposition: callable (path, cloud, drone_radius, z_ground) -> polytopes, the role of JPS_Manager::cvxEllipsoidDecomp.
template <class Solver, class Decomposition = HostDecomposition>
class Planner {
public:
  explicit Planner(const Params& p) : par_(p) {
    double mv[3] = {p.v_max, p.a_max, p.j_max};
    // faster.cpp:52-71
    sg_whole_.setN(p.N_whole); sg_whole_.createVars(); sg_whole_.setDC(p.dc); sg_whole_.setBounds(mv);
    sg_whole_.setForceFinalConstraint(true); sg_whole_.setFactorInitialAndFinalAndIncrement(1, 10, p.increment_whole);
    sg_whole_.setVerbose(0); sg_whole_.setThreads(0); sg_whole_.setWMax(1.0);
    sg_safe_.setN(p.N_safe); sg_safe_.createVars(); sg_safe_.setDC(p.dc); sg_safe_.setBounds(mv);
    sg_safe_.setForceFinalConstraint(false); sg_safe_.setFactorInitialAndFinalAndIncrement(1, 10, p.increment_safe);
    sg_safe_.setVerbose(0); sg_safe_.setThreads(0); sg_safe_.setWMax(1.0);
  }

  void setTerminalGoal(const state& g) { G_term_ = g; goal_set_ = true; status_ = Status::TRAVELING; }
  void updateState(const state& s) {  // faster.cpp:139-155: the first state seeds the plan
    state_ = s;
    if (!state_set_) { plan_.clear(); plan_.push_back(s); }
    state_set_ = true;
  }
  void updateMap(const std::vector<V3>& occupied, const std::vector<V3>& unknown) {  // faster.cpp:99-137
    occupied_.pts = occupied;
    unknown_.pts = unknown;
    unknown_and_occupied_ = unknown;
    unknown_and_occupied_.insert(unknown_and_occupied_.end(), occupied.begin(), occupied.end());
    map_set_ = true;
  }
  bool getNextGoal(state& next) {  // faster.cpp:699-723 without yaw
    if (!state_set_ || !goal_set_ || plan_.empty()) return false;
    next = plan_.front();
    if (plan_.size() > 1) plan_.pop_front();
    return true;
  }
  Status status() const { return status_; }
  const std::deque<state>& plan() const { return plan_; }
  // A fresh vehicle for the same planner object (a batch of independent start/goal pairs through one pair of solver contexts): no
  // state, no plan, the factor windows back to what the constructor set (faster.cpp:57, :68)
  void reset() {
    state_set_ = false;
    goal_set_ = false;
    status_ = Status::TRAVELING;
    plan_.clear();
    sg_whole_.setFactorInitialAndFinalAndIncrement(1, 10, par_.increment_whole);
    sg_safe_.setFactorInitialAndFinalAndIncrement(1, 10, par_.increment_safe);
  }

  bool replan(ReplanLog* log = nullptr) {
    ReplanLog local;
    ReplanLog& L = log ? *log : local;
    L = ReplanLog();
    if (!(state_set_ && goal_set_ && map_set_)) return false;
    sg_whole_.ResetToNormalState();
    sg_safe_.ResetToNormalState();
    const V3 here = pos_of(state_), gterm = pos_of(G_term_);
    const V3 G = project_to_box(here, gterm, par_.wdx, par_.wdy, par_.wdz);
    const double dist_to_goal = (gterm - here).norm();
    if (dist_to_goal < par_.goal_radius) status_ = Status::GOAL_REACHED;
    if (status_ == Status::GOAL_REACHED) return false;

    // ---- start state A: deltaT states before the end of the committed plan (:351-352)
    const int k_end_whole = std::max((int)plan_.size() - par_.deltaT, 0);
    state A = plan_[plan_.size() - 1 - k_end_whole];
    L.k_end_whole = k_end_whole;

    // ---- path search in the known map (:361), clipped to the sphere S of radius ra around A (:373-384)
    fhfront::VoxelGrid grid;
    if (par_.map_fixed)
      grid.build(occupied_.pts, par_.map_cells[0], par_.map_cells[1], par_.map_cells[2], par_.factor_jps * par_.res,
                 V3(par_.map_center[0], par_.map_center[1], par_.map_center[2]), par_.z_ground, par_.z_max, par_.inflation_jps);
    else
      grid.build(occupied_.pts, (int)(par_.wdx / par_.res), (int)(par_.wdy / par_.res), (int)(par_.wdz / par_.res), par_.factor_jps * par_.res, here,
                 par_.z_ground, par_.z_max, par_.inflation_jps);
    std::vector<V3> JPSk;
    const bool found = par_.jps ? fhfront::plan_path_jps(grid, pos_of(A), G, par_.inflation_jps, JPSk)
                                : fhfront::plan_path(grid, pos_of(A), G, par_.inflation_jps, JPSk);
    if (!found) { L.stage = 1; return false; }
    const double ra = std::min(dist_to_goal - 0.001, par_.Ra);
    int li1 = 0;
    bool none_outside = false;
    V3 Epos = sphere_exit(JPSk, ra, JPSk[0], &li1, &none_outside);
    std::vector<V3> JPS_in(JPSk.begin(), JPSk.begin() + li1 + 1);
    if (!none_outside) JPS_in.push_back(Epos);
    subdivide(JPS_in, par_.dist_max_vertexes);

    // ---- whole trajectory: corridor in occupied space, final position forced (:390-431)
    std::vector<V3> JPS_whole = JPS_in;
    keep_first(JPS_whole, par_.max_poly_whole);
    Epos = JPS_whole.back();
    const double decomp_r = par_.decomp_radius >= 0 ? par_.decomp_radius : par_.drone_radius;
    l_constraints_whole_ = to_solver_constraints(decompose_(JPS_whole, occupied_.pts, decomp_r, par_.z_ground));
    if (l_constraints_whole_.empty()) { L.stage = 2; return false; }  // (a device decomposition that failed reports an empty corridor)
    if (l_constraints_whole_.back().inside(fhstub::Vec3(G.x, G.y, G.z))) Epos = G;
    state E;
    E.setPos(Epos.x, Epos.y, Epos.z);
    sg_whole_.setX0(A);
    sg_whole_.setXf(E);
    sg_whole_.setPolytopes(l_constraints_whole_);
    if (!sg_whole_.genNewTraj()) { L.stage = 2; return false; }
    sg_whole_.fillX();
    L.whole_factor = sg_whole_.factor_that_worked_;
    L.n_whole = sg_whole_.X_temp_.size();

    // ---- safe trajectory (:446-541)
    std::vector<V3> tmp = JPS_in;
    bool hit = false;
    V3 Mpos = march_to_cloud(tmp, unknown_, par_.drone_radius, &hit);
    bool need_safe = false;
    const int indexH = find_index_H(need_safe);
    L.index_H = indexH;
    L.needed_safe = need_safe;
    int k_safe;
    if (!need_safe) {
      k_safe = indexH;
      sg_safe_.X_temp_ = std::vector<state>();
    } else {
      k_safe = find_index_R(indexH);
      L.k_safe = k_safe;
      state R = sg_whole_.X_temp_[k_safe];
      tmp[0] = pos_of(R);
      std::vector<V3> JPS_safe = tmp;
      keep_first(JPS_safe, par_.max_poly_safe);
      Mpos = JPS_safe.back();
      l_constraints_safe_ = to_solver_constraints(decompose_(JPS_safe, unknown_and_occupied_, decomp_r, par_.z_ground));
      if (l_constraints_safe_.empty()) { L.stage = 3; return false; }  // failed (device) decomposition: no safe corridor, as for the whole one
      if (l_constraints_safe_.back().inside(fhstub::Vec3(G.x, G.y, G.z))) Mpos = G;
      L.safe_path = JPS_safe;
      L.safe_goal = Mpos;
      for (const auto& c : l_constraints_safe_) L.safe_rows.push_back((int)c.b().rows());
      state M;
      M.setPos(Mpos.x, Mpos.y, Mpos.z);
      sg_safe_.setX0(R);
      sg_safe_.setXf(M);  // only used to compute dt (:522)
      sg_safe_.setPolytopes(l_constraints_safe_);
      sg_safe_.setForceFinalConstraint(false);
      if (!sg_safe_.genNewTraj()) { L.stage = 3; return false; }
      sg_safe_.fillX();
      L.safe_factor = sg_safe_.factor_that_worked_;
      L.n_safe = sg_safe_.X_temp_.size();
    }
    L.k_safe = k_safe;

    // ---- commit (:548) and adapt the factor windows (:582-588)
    if (!append_to_plan(k_end_whole, sg_whole_.X_temp_, k_safe, sg_safe_.X_temp_)) { L.stage = 4; return false; }
    if ((gterm - pos_of(plan_.back())).norm() < par_.goal_radius) status_ = Status::GOAL_SEEN;
    sg_whole_.setFactorInitialAndFinalAndIncrement(std::max(sg_whole_.factor_that_worked_ - par_.gamma_whole, 1.0),
                                                   sg_whole_.factor_that_worked_ + par_.gammap_whole, par_.increment_whole);
    sg_safe_.setFactorInitialAndFinalAndIncrement(std::max(sg_safe_.factor_that_worked_ - par_.gamma_safe, 1.0),
                                                  sg_safe_.factor_that_worked_ + par_.gammap_safe, par_.increment_safe);
    L.stage = 5;
    return true;
  }

  Solver sg_whole_, sg_safe_;  // same member names as faster.hpp:74-75

private:
  int find_index_H(bool& need_safe) const {  // first sampled state of the whole trajectory that touches unknown space
    need_safe = false;
    int indexH = (int)sg_whole_.X_temp_.size() - 1;
    for (size_t i = 0; i < sg_whole_.X_temp_.size(); i += 10) {
      double d;
      if (unknown_.nearest(pos_of(sg_whole_.X_temp_[i]), d) && d < par_.drone_radius) {
        need_safe = true;
        indexH = (int)(par_.delta_H * (double)i);
        break;
      }
    }
    return indexH;
  }
  int find_index_R(int indexH) const {  // earliest state that could no longer brake before H (x, y only)
    const state& H = sg_whole_.X_temp_[indexH];
    for (int i = 0; i <= indexH; i++) {
      const state& s = sg_whole_.X_temp_[i];
      for (int ax = 0; ax < 2; ax++) {
        const double v = s.vel(ax), gap = H.pos(ax) - s.pos(ax);
        const double sg = (v * gap > 0) - (v * gap < 0);
        if (sg * v * v / (2 * par_.delta_a * par_.a_max) > std::fabs(gap)) return i;
      }
    }
    return indexH;
  }
  bool append_to_plan(int k_end_whole, const std::vector<state>& whole, int k_safe, const std::vector<state>& safe) {
    if ((int)plan_.size() - 1 - k_end_whole < 0) return false;
    plan_.erase(plan_.end() - k_end_whole - 1, plan_.end());
    for (int i = 0; i <= k_safe && i < (int)whole.size(); i++) plan_.push_back(whole[i]);
    for (const state& s : safe) plan_.push_back(s);
    return true;
  }

  Params par_;
  state state_, G_term_;
  bool state_set_ = false, goal_set_ = false, map_set_ = false;
  Status status_ = Status::TRAVELING;
  std::deque<state> plan_;
  Decomposition decompose_;
  Cloud occupied_, unknown_;
  std::vector<V3> unknown_and_occupied_;
  std::vector<LinearConstraint3D> l_constraints_whole_, l_constraints_safe_;
};

}  // namespace fhreplan
