// JpsHip (device) against fhfront::plan_path_jps and, with setJumpPointSearch(false), fhfront::plan_path (host) through the JPS_Manager-shaped calls: a forest scene read from a file written by
// the Python test.  usage: test_jps_hip scene.txt   -> prints JPS_OK <queries> <solved> on success
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "corridor_frontend.hpp"
#include "jps_hip.hpp"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::ifstream in(argv[1]);
  int n_cloud, n_q, cells[3];
  double res, infl, zg, zmax, center[3];
  in >> cells[0] >> cells[1] >> cells[2] >> res >> infl >> zg >> zmax >> center[0] >> center[1] >> center[2] >> n_cloud >> n_q;
  std::vector<fhfront::V3> cloud(n_cloud), starts(n_q), goals(n_q);
  for (auto& p : cloud) in >> p.x >> p.y >> p.z;
  for (int i = 0; i < n_q; i++) in >> starts[i].x >> starts[i].y >> starts[i].z >> goals[i].x >> goals[i].y >> goals[i].z;
  if (!in) { std::printf("bad scene file\n"); return 2; }

  JpsHip jps;
  jps.setNumCells(cells[0], cells[1], cells[2]);
  jps.setFactorJPS(1.0);
  jps.setResolution(res);
  jps.setInflationJPS(infl);
  jps.setZGroundAndZMax(zg, zmax);
  if (!jps.updateJPSMap(cloud, fhfront::V3(center[0], center[1], center[2]))) { std::printf("updateJPSMap failed: %s\n", jps.lastError().c_str()); return 1; }
  fhfront::VoxelGrid base;
  base.build(cloud, cells[0], cells[1], cells[2], res, fhfront::V3(center[0], center[1], center[2]), zg, zmax, infl);
  int n_solved = 0;
  std::vector<char> solved;
  std::vector<std::vector<fhfront::V3>> paths;
  for (int mode = 1; mode >= 0; mode--) {  // 1: jump point search in jps3d's order (the default, as JPS_Manager), 0: the A* of mode 0
    if (!jps.setJumpPointSearch(mode == 1)) { std::printf("setJumpPointSearch failed\n"); return 1; }
    paths = jps.solveJPS3DBatch(starts, goals, &solved);
    n_solved = 0;
    for (int i = 0; i < n_q; i++) {
      fhfront::VoxelGrid g = base;
      std::vector<fhfront::V3> ref;
      const bool ok = mode == 1 ? fhfront::plan_path_jps(g, starts[i], goals[i], infl, ref) : fhfront::plan_path(g, starts[i], goals[i], infl, ref);
      if (ok != (bool)solved[i]) { std::printf("mode %d query %d: solved %d vs host %d\n", mode, i, (int)solved[i], (int)ok); return 1; }
      if (!ok) continue;
      n_solved++;
      if (ref.size() != paths[i].size()) { std::printf("mode %d query %d: %zu vertices vs host %zu\n", mode, i, paths[i].size(), ref.size()); return 1; }
      for (size_t k = 0; k < ref.size(); k++)
        if (ref[k].x != paths[i][k].x || ref[k].y != paths[i][k].y || ref[k].z != paths[i][k].z) { std::printf("mode %d query %d vertex %zu differs\n", mode, i, k); return 1; }
    }
  }
  // the single-query call of the reference
  bool one = false;
  const auto p0 = jps.solveJPS3D(starts[0], goals[0], &one);
  if (one != (bool)solved[0] || p0.size() != paths[0].size()) { std::printf("solveJPS3D differs from the batch\n"); return 1; }
  // ... which JpsHip searches on the host by default (one query: 0.1 ms there, 3 ms on a lone wavefront); the same call sent to the device
  for (int i = 0; i < (n_q < 8 ? n_q : 8); i++) {
    bool on_host = false, on_device = false;
    jps.setSingleQueryOnHost(true);
    const auto ph = jps.solveJPS3D(starts[i], goals[i], &on_host);
    jps.setSingleQueryOnHost(false);
    const auto pd = jps.solveJPS3D(starts[i], goals[i], &on_device);
    if (on_host != on_device || ph.size() != pd.size()) { std::printf("single query %d: host route and device route differ\n", i); return 1; }
    for (size_t k = 0; k < ph.size(); k++)
      if (ph[k].x != pd[k].x || ph[k].y != pd[k].y || ph[k].z != pd[k].z) { std::printf("single query %d vertex %zu: host route and device route differ\n", i, k); return 1; }
  }
  std::printf("JPS_OK %d %d\n", n_q, n_solved);
  return 0;
}
