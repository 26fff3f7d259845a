// ref_frontend.cpp — TEST INFRASTRUCTURE ONLY.  A C ABI around the reference's OWN corridor front-end sources, compiled untouched
// from where they lie under /root/reference (build.sh; outputs only into oracle/_ref/):
//   thirdparty/DecompROS/DecompUtil/include/decomp_util/{ellipsoid_decomp.h, line_segment.h, decomp_base.h},
//   decomp_geometry/{ellipsoid.h, polyhedron.h, geometric_utils.h}      (header only)
//   thirdparty/jps3d/src/jps_planner/{graph_search.cpp, jps_planner.cpp}, include/jps_collision/map_util.h
// Eigen, Boost.Heap, ROS and PCL are absent from this image: oracle/ref_frontend/shim/ supplies the elementary matrix
// arithmetic, the d-ary heap and two empty/POD headers those sources name.  The algorithms are the reference's.
//
// This file restates only the few ROS-bound lines of FASTER that CALL those libraries, citing them:
//   JPS_Manager::cvxEllipsoidDecomp  /root/reference/faster/src/jps_manager.cpp:80-127
//   JPS_Manager::updateJPSMap        jps_manager.cpp:129-139
//   JPS_Manager::solveJPS3D          jps_manager.cpp:141-200
// Nothing under faster_amd/ links or loads this library; tests/ use it to pin the device front-end and its host restatement.
#include <cstring>
#include <memory>

#include <decomp_util/ellipsoid_decomp.h>
#include <jps_planner/jps_planner/jps_planner.h>

struct RefMap {
  std::shared_ptr<JPS::MapUtil<3>> map;
  double inflation;
};

extern "C" {

// cvxEllipsoidDecomp: polytopes A x <= b around every leg of `path` from the obstacle points `cloud`; rows [n_seg][max_rows][4] =
// (a_x, a_y, a_z, b), the ground row last.  Returns 0, or -1 if a polytope has more than max_rows rows.
int ref_decompose(const double* path, int n_pts, const double* cloud, int n_cloud, double drone_radius, double z_ground, double* out_rows,
                  int* out_counts, int max_rows) {
  vec_Vecf<3> p, obs;
  for (int i = 0; i < n_pts; i++) p.push_back(Vec3f(path[3 * i], path[3 * i + 1], path[3 * i + 2]));
  for (int i = 0; i < n_cloud; i++) obs.push_back(Vec3f(cloud[3 * i], cloud[3 * i + 1], cloud[3 * i + 2]));
  EllipsoidDecomp3D ellip_decomp_util_;
  ellip_decomp_util_.set_obs(obs);                          // jps_manager.cpp:91-98
  ellip_decomp_util_.set_local_bbox(Vec3f(2, 2, 1));        // :100
  ellip_decomp_util_.set_inflate_distance(drone_radius);    // :102
  ellip_decomp_util_.dilate(p);                             // :103
  auto polys = ellip_decomp_util_.get_polyhedrons();        // :109
  int rc = 0;
  for (size_t i = 0; i + 1 < p.size(); i++) {               // :113-125
    const Vec3f pt_inside = (p[i] + p[i + 1]) / 2;
    LinearConstraint3D cs(pt_inside, polys[i].hyperplanes());
    const int rows = (int)cs.A_.rows();
    out_counts[i] = rows + 1;
    if (rows + 1 > max_rows) { rc = -1; continue; }
    double* o = out_rows + (size_t)i * max_rows * 4;
    for (int r = 0; r < rows; r++) {
      o[4 * r] = cs.A_(r, 0); o[4 * r + 1] = cs.A_(r, 1); o[4 * r + 2] = cs.A_(r, 2); o[4 * r + 3] = cs.b_(r);
    }
    o[4 * rows] = 0; o[4 * rows + 1] = 0; o[4 * rows + 2] = -1; o[4 * rows + 3] = -z_ground;  // "above the ground" :118-122
  }
  return rc;
}

// updateJPSMap: MapUtil::readMap on a cloud of pcl::PointXYZ (FLOAT coordinates, as in the reference)
RefMap* ref_map_create(const float* cloud, int n_cloud, int cells_x, int cells_y, int cells_z, double res, const double* center, double z_ground,
                       double z_max, double inflation) {
  pcl::PointCloud<pcl::PointXYZ>::Ptr pclptr(new pcl::PointCloud<pcl::PointXYZ>());
  pclptr->points.resize((size_t)n_cloud);
  for (int i = 0; i < n_cloud; i++) pclptr->points[i] = pcl::PointXYZ{cloud[3 * i], cloud[3 * i + 1], cloud[3 * i + 2]};
  RefMap* m = new RefMap();
  m->map = std::make_shared<JPS::MapUtil<3>>();
  m->inflation = inflation;
  Vec3f center_map(center[0], center[1], center[2]);
  m->map->readMap(pclptr, cells_x, cells_y, cells_z, res, center_map, z_ground, z_max, inflation);  // jps_manager.cpp:135-136
  return m;
}
void ref_map_destroy(RefMap* m) { delete m; }
void ref_map_dims(RefMap* m, int* dims, double* origin) {
  const Veci<3> d = m->map->getDim();
  const Vecf<3> o = m->map->getOrigin();
  for (int i = 0; i < 3; i++) { dims[i] = d(i); origin[i] = o(i); }
}
void ref_map_occupancy(RefMap* m, signed char* out) {
  const JPS::Tmap t = m->map->getMap();
  std::memcpy(out, t.data(), t.size());
}

// solveJPS3D on a fresh copy of the map (the reference rebuilds its map between replans; freeing the voxels around start and goal
// is per call).  use_jps = 1: jump point search (what FASTER runs, jps_manager.cpp:164), 0: jps3d's A*.
// Returns the number of path points written (0: no path), -1 if max_pts is too small.  raw_cost: length of the raw path in metres.
int ref_map_plan(RefMap* m, const double* start_sent, const double* goal_sent, int use_jps, double* out_path, int max_pts, double* raw_cost,
                 int* raw_points) {
  auto map_util_ = std::make_shared<JPS::MapUtil<3>>(*m->map);
  Vec3f start(start_sent[0], start_sent[1], std::max(start_sent[2], 0.0));  // :143-144
  Vec3f goal(goal_sent[0], goal_sent[1], std::max(goal_sent[2], 0.0));
  const Veci<3> start_int = map_util_->floatToInt(start);                   // :158-159
  const Veci<3> goal_int = map_util_->floatToInt(goal);
  map_util_->setFreeVoxelAndSurroundings(start_int, m->inflation);          // :161-162
  map_util_->setFreeVoxelAndSurroundings(goal_int, m->inflation);
  JPSPlanner3D planner(false);
  planner.setMapUtil(map_util_);                                            // :164
  const bool valid_jps = planner.plan(start, goal, 1, use_jps != 0);        // :166
  if (raw_cost) *raw_cost = 0;
  if (raw_points) *raw_points = 0;
  if (!valid_jps) return 0;
  vec_Vecf<3> path = planner.getPath();                                     // :174-189
  if (path.size() > 1) {
    path[0] = start;
    path[path.size() - 1] = goal;
  } else {
    vec_Vecf<3> tmp;
    tmp.push_back(start);
    tmp.push_back(goal);
    path = tmp;
  }
  const vec_Vecf<3> raw = planner.getRawPath();
  if (raw_cost) *raw_cost = total_distance3f(raw);
  if (raw_points) *raw_points = (int)raw.size();
  if ((int)path.size() > max_pts) return -1;
  for (size_t i = 0; i < path.size(); i++)
    for (int k = 0; k < 3; k++) out_path[3 * i + k] = path[i](k);
  return (int)path.size();
}

// The raw path of the same call (jump points, start -> goal), for diagnosing a differing vertex list.  Returns the number of points.
int ref_map_raw_path(RefMap* m, const double* start_sent, const double* goal_sent, int use_jps, double* out_raw, int max_pts) {
  auto map_util_ = std::make_shared<JPS::MapUtil<3>>(*m->map);
  Vec3f start(start_sent[0], start_sent[1], std::max(start_sent[2], 0.0));
  Vec3f goal(goal_sent[0], goal_sent[1], std::max(goal_sent[2], 0.0));
  map_util_->setFreeVoxelAndSurroundings(map_util_->floatToInt(start), m->inflation);
  map_util_->setFreeVoxelAndSurroundings(map_util_->floatToInt(goal), m->inflation);
  JPSPlanner3D planner(false);
  planner.setMapUtil(map_util_);
  if (!planner.plan(start, goal, 1, use_jps != 0)) return 0;
  const vec_Vecf<3> raw = planner.getRawPath();
  if ((int)raw.size() > max_pts) return -1;
  for (size_t i = 0; i < raw.size(); i++)
    for (int k = 0; k < 3; k++) out_raw[3 * i + k] = raw[i](k);
  return (int)raw.size();
}

// jps3d's neighbour tables as its constructor builds them (graph_search.cpp:573-937): ns [27][3][26], f1 / f2 [27][3][12]
void ref_jps3d_tables(int* ns, int* f1, int* f2) {
  JPS::JPS3DNeib jn;
  std::memcpy(ns, jn.ns, sizeof(jn.ns));
  std::memcpy(f1, jn.f1, sizeof(jn.f1));
  std::memcpy(f2, jn.f2, sizeof(jn.f2));
}

const char* ref_frontend_sources(void) {
  return "DecompUtil decomp_util/{ellipsoid_decomp,line_segment,decomp_base}.h decomp_geometry/{ellipsoid,polyhedron,geometric_utils}.h; "
         "jps3d src/jps_planner/{graph_search,jps_planner}.cpp include/jps_collision/map_util.h — untouched, from /root/reference/thirdparty";
}
}
