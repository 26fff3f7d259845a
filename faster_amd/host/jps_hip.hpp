// jps_hip.hpp — the path-search half of JPS_Manager on the device, behind the calls the reference makes.
//
// Reference: JPS_Manager (/root/reference/faster/include/jps_manager.hpp:40-59, src/jps_manager.cpp): setNumCells / setFactorJPS /
// setResolution / setInflationJPS / setZGroundAndZMax configure the map (:42-68), updateJPSMap(cloud, center) reads it (:129-139),
// solveJPS3D(start, goal, &solved, i) searches it (:141-200).  JpsHip keeps those names and meanings over the C ABI's fh_map_*
// (include/fasterhip.h) and searches as JPS_Manager does: jump point search in jps3d's own order (planner_ptr_->plan(start, goal, 1,
// true), jps_manager.cpp:166; fh_map_set_search mode 1), so the vertex lists are the reference's; setJumpPointSearch(false) selects
// the A* of mode 0 (another optimal path, no per-map jump tables).  The device searches one query per wavefront, so the call that pays is
// solveJPS3DBatch (Monte-Carlo goals, many agents sharing a map).  [r6] ONE query — solveJPS3D, what a single vehicle's replan asks
// (jps_manager.cpp:141-196) — is searched on the HOST by this library's own restatement of jps3d (fhfront::plan_path_jps,
// corridor_frontend.cpp: the same vertex list, checked against the device and against the compiled reference): a lone wavefront needs
// 3.3 ms for a forest query (769 pops at 4.2 us) where the host search needs ~0.1 ms, and a one-vehicle caller must not get a 33x slower
// front end than the reference's.  The host grid is MapUtil::readMap of the same cloud (built on the first single query after an
// updateJPSMap); like the reference's map it keeps the cells freed around a start and a goal until the next update (the device search
// frees them per query).  setSingleQueryOnHost(false) sends single queries to the device as a batch of one.  This is a routing
// decision inside the front end, not a fallback: updateJPSMap needs the device, and without one every search
// reports solved = false and lastError() says why.
#pragma once
#include <string>
#include <vector>

#include "../../include/fasterhip.h"
#include "corridor_frontend.hpp"

class JpsHip {
public:
  JpsHip() = default;
  ~JpsHip();
  JpsHip(const JpsHip&) = delete;
  JpsHip& operator=(const JpsHip&) = delete;

  void setNumCells(int cells_x, int cells_y, int cells_z) { cells_[0] = cells_x; cells_[1] = cells_y; cells_[2] = cells_z; }
  void setFactorJPS(double factor_jps) { factor_jps_ = factor_jps; }
  void setResolution(double res) { res_ = res; }
  void setInflationJPS(double inflation_jps) { inflation_jps_ = inflation_jps; }
  void setZGroundAndZMax(double z_ground, double z_max) { z_ground_ = z_ground; z_max_ = z_max; }
  bool setJumpPointSearch(bool on);  // default on: JPS_Manager plans with use_jps = true
  void setSingleQueryOnHost(bool on) { single_on_host_ = on; }  // default on (see above)

  // MapUtil::readMap with cell size factor_jps * res (jps_manager.cpp:135-136); returns false on a device error
  bool updateJPSMap(const std::vector<fhfront::V3>& cloud, const fhfront::V3& center);
  // one query (jps_manager.cpp:141-200): empty path and *solved = false when there is none
  std::vector<fhfront::V3> solveJPS3D(const fhfront::V3& start, const fhfront::V3& goal, bool* solved);
  // n queries over the current map in one launch; solved[i] as above.  max_vertex_dist > 0 adds Faster::createMoreVertexes
  // (faster.cpp:80-97) and, with max_poly > 0, deleteVertexes (utils.cpp:1117-1124): the vertices cvxEllipsoidDecomp gets.
  std::vector<std::vector<fhfront::V3>> solveJPS3DBatch(const std::vector<fhfront::V3>& starts, const std::vector<fhfront::V3>& goals,
                                                        std::vector<char>* solved, double max_vertex_dist = 0.0, int max_poly = 0);

  const std::string& lastError() const { return err_; }
  int deviceStatus() const { return rc_; }

private:
  bool ensureMap();
  fh_map* map_ = nullptr;
  bool create_failed_ = false, jump_point_search_ = true, single_on_host_ = true, have_cloud_ = false, grid_stale_ = true;
  std::vector<fhfront::V3> cloud_;  // of the last updateJPSMap: the host grid of single queries is built from it on demand
  fhfront::V3 center_;
  fhfront::VoxelGrid grid_;
  int rc_ = FH_OK;
  std::string err_;
  int32_t cells_[3] = {200, 200, 20};
  double factor_jps_ = 1.0, res_ = 0.1, inflation_jps_ = 0.0, z_ground_ = 0.0, z_max_ = 3.0;
};
