// jps_hip.cpp — see jps_hip.hpp
#include "jps_hip.hpp"

#include <cstdio>

JpsHip::~JpsHip() {
  if (map_) fh_map_destroy(map_);
}

bool JpsHip::ensureMap() {
  if (map_) return true;
  if (create_failed_) return false;  // stay failed, loudly, instead of retrying on every replan
  rc_ = fh_map_create(&map_, 0);
  if (rc_ != FH_OK) {
    create_failed_ = true;
    err_ = "fh_map_create failed (no HIP device? there is no CPU fallback)";
    std::fprintf(stderr, "JpsHip: %s\n", err_.c_str());
    return false;
  }
  rc_ = fh_map_set_search(map_, jump_point_search_ ? 1 : 0);
  return rc_ == FH_OK;
}

bool JpsHip::setJumpPointSearch(bool on) {
  jump_point_search_ = on;
  if (!map_) return true;
  rc_ = fh_map_set_search(map_, on ? 1 : 0);
  return rc_ == FH_OK;
}

bool JpsHip::updateJPSMap(const std::vector<fhfront::V3>& cloud, const fhfront::V3& center) {
  if (!ensureMap()) return false;
  std::vector<double> xyz(3 * cloud.size());
  for (size_t i = 0; i < cloud.size(); i++) { xyz[3 * i] = cloud[i].x; xyz[3 * i + 1] = cloud[i].y; xyz[3 * i + 2] = cloud[i].z; }
  const double c[3] = {center.x, center.y, center.z};
  rc_ = fh_map_read(map_, xyz.empty() ? nullptr : xyz.data(), (int)cloud.size(), cells_, factor_jps_ * res_, c, z_ground_, z_max_, inflation_jps_);
  if (rc_ != FH_OK) {
    err_ = fh_map_last_error(map_);
    std::fprintf(stderr, "JpsHip: updateJPSMap: rc=%d %s\n", rc_, err_.c_str());
    have_cloud_ = false;
    return false;
  }
  cloud_ = cloud;
  center_ = center;
  have_cloud_ = true;
  grid_stale_ = true;
  return true;
}

std::vector<std::vector<fhfront::V3>> JpsHip::solveJPS3DBatch(const std::vector<fhfront::V3>& starts, const std::vector<fhfront::V3>& goals,
                                                              std::vector<char>* solved, double max_vertex_dist, int max_poly) {
  const size_t n = starts.size();
  std::vector<std::vector<fhfront::V3>> out(n);
  if (solved) solved->assign(n, 0);
  if (n == 0 || goals.size() != n || !ensureMap()) return out;
  std::vector<double> s(3 * n), g(3 * n);
  for (size_t i = 0; i < n; i++) {
    s[3 * i] = starts[i].x; s[3 * i + 1] = starts[i].y; s[3 * i + 2] = starts[i].z;
    g[3 * i] = goals[i].x; g[3 * i + 1] = goals[i].y; g[3 * i + 2] = goals[i].z;
  }
  int max_points = max_poly > 0 ? max_poly + 1 : 64;
  std::vector<double> paths;
  std::vector<int32_t> np(n);
  for (;;) {  // a path with more vertices than expected is reported (-1), never truncated: retry that call with more room
    paths.assign(3 * n * (size_t)max_points, 0.0);
    rc_ = fh_map_plan_batch(map_, s.data(), g.data(), (int)n, max_points, max_vertex_dist, max_poly, paths.data(), np.data(), nullptr);
    if (rc_ != FH_OK) {
      err_ = fh_map_last_error(map_);
      std::fprintf(stderr, "JpsHip: solveJPS3D: rc=%d %s\n", rc_, err_.c_str());
      return out;
    }
    bool over = false;
    for (size_t i = 0; i < n; i++) over = over || np[i] == -1;
    if (!over || max_points >= 4096) break;
    max_points *= 4;
  }
  for (size_t i = 0; i < n; i++) {
    if (np[i] <= 0) continue;  // 0: no path; -2: a search limit (include/fasterhip.h) — both "JPS didn't find a solution"
    out[i].reserve((size_t)np[i]);
    for (int k = 0; k < np[i]; k++) {
      const double* p = &paths[3 * (i * (size_t)max_points + (size_t)k)];
      out[i].push_back(fhfront::V3(p[0], p[1], p[2]));
    }
    if (solved) (*solved)[i] = 1;
  }
  return out;
}

std::vector<fhfront::V3> JpsHip::solveJPS3D(const fhfront::V3& start, const fhfront::V3& goal, bool* solved) {
  if (single_on_host_ && have_cloud_) {  // one query: the host restatement of the same search (see jps_hip.hpp)
    if (grid_stale_) {
      grid_.build(cloud_, cells_[0], cells_[1], cells_[2], factor_jps_ * res_, center_, z_ground_, z_max_, inflation_jps_);
      grid_stale_ = false;
    }
    std::vector<fhfront::V3> path;
    const bool found = jump_point_search_ ? fhfront::plan_path_jps(grid_, start, goal, inflation_jps_, path)
                                          : fhfront::plan_path(grid_, start, goal, inflation_jps_, path);
    if (solved) *solved = found;
    if (!found) {
      std::fprintf(stderr, "JPS didn't find a solution from (%g %g %g) to (%g %g %g)\n", start.x, start.y, start.z, goal.x, goal.y, goal.z);
      path.clear();
    }
    return path;
  }
  std::vector<char> ok;
  std::vector<std::vector<fhfront::V3>> r = solveJPS3DBatch({start}, {goal}, &ok);
  if (solved) *solved = !ok.empty() && ok[0];
  if (!ok.empty() && !ok[0]) std::fprintf(stderr, "JPS didn't find a solution from (%g %g %g) to (%g %g %g)\n", start.x, start.y, start.z, goal.x, goal.y, goal.z);
  return r.empty() ? std::vector<fhfront::V3>() : r[0];
}
