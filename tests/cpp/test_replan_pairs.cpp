// A batch of independent start/goal pairs through the CALLER of the hot path, one pair at a time: replan_stub.hpp — the restatement of
// Faster::replan (/root/reference/faster/src/faster.cpp:296-595) with its helpers — drives SolverHip exactly as the reference drives
// SolverGurobi, with explicit occupied and unknown point clouds (updateMap, faster.cpp:99-137).  tests/test_gpu_round4.py compares, pair
// by pair, what the DEVICE pipeline of the same replan commits for the same inputs (map, jump point search, whole corridor, whole
// solve, findIndexH / findIndexR and the safe corridor against the unknown voxels given as an input, safe solve, appendToPlan).
//   usage: test_replan_pairs <scenario.bin> <out.bin> [oracle liboracle.so]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "replan_stub.hpp"
#ifdef WITH_ORACLE
#include "oracle_solver.hpp"
#endif
#include "solver_hip.hpp"

using fhfront::V3;

template <class Solver>
static int run(const int32_t* hi, const double* hd, const std::vector<V3>& occ, const std::vector<V3>& unk, const std::vector<double>& pairs,
               const char* out_path) {
  fhreplan::Params par;
  par.N_whole = hi[0]; par.N_safe = hi[1]; par.max_poly_whole = hi[2]; par.max_poly_safe = hi[3];
  par.dc = hd[0]; par.v_max = hd[1]; par.a_max = hd[2]; par.j_max = hd[3]; par.Ra = hd[4]; par.drone_radius = hd[5]; par.decomp_radius = hd[6];
  par.dist_max_vertexes = hd[7]; par.delta_a = hd[8]; par.delta_H = hd[9]; par.res = hd[10]; par.inflation_jps = hd[11]; par.z_ground = hd[12];
  par.z_max = hd[13];
  par.map_fixed = true;
  for (int k = 0; k < 3; k++) { par.map_center[k] = hd[14 + k]; par.map_cells[k] = hi[4 + k]; }
  par.jps = true;
  par.wdx = par.wdy = par.wdz = 1e6;  // (the goal is never projected: the map is the fixed one)
  par.goal_radius = 0.0;
  fhreplan::Planner<Solver> planner(par);
  FILE* out = std::fopen(out_path, "wb");
  if (!out) return 4;
  const int n = hi[7];
  for (int i = 0; i < n; i++) {
    const double* p = &pairs[9 * (size_t)i];
    state A, G;
    A.setPos(p[0], p[1], p[2]);
    A.setVel(p[3], p[4], p[5]);
    G.setPos(p[6], p[7], p[8]);
    planner.reset();
    planner.setTerminalGoal(G);
    planner.updateState(A);
    planner.updateMap(occ, unk);
    fhreplan::ReplanLog L;
    const bool ok = planner.replan(&L);
    if (std::getenv("FH_DEBUG_PAIR") && std::atoi(std::getenv("FH_DEBUG_PAIR")) == i) {
      std::fprintf(stderr, "pair %d: stage %d safe path", i, L.stage);
      for (const V3& v : L.safe_path) std::fprintf(stderr, " (%.17g, %.17g, %.17g)", v.x, v.y, v.z);
      std::fprintf(stderr, " | xf (%.17g, %.17g, %.17g) | rows", L.safe_goal.x, L.safe_goal.y, L.safe_goal.z);
      for (int r : L.safe_rows) std::fprintf(stderr, " %d", r);
      std::fprintf(stderr, " | safe factor %g dt %.17g\n", L.safe_factor, planner.sg_safe_.dt_);
    }
    const auto& plan = planner.plan();
    const int32_t rec[8] = {ok ? 1 : 0, L.stage, L.needed_safe ? 1 : 0, L.index_H, L.k_safe, (int32_t)L.n_whole, (int32_t)L.n_safe,
                            ok ? (int32_t)plan.size() : 0};
    const double fac[2] = {L.whole_factor, L.safe_factor};
    std::fwrite(rec, sizeof(rec), 1, out);
    std::fwrite(fac, sizeof(fac), 1, out);
    if (ok)
      for (const state& s : plan) {
        const double v[12] = {s.pos.x(), s.pos.y(), s.pos.z(), s.vel.x(), s.vel.y(), s.vel.z(), s.accel.x(), s.accel.y(), s.accel.z(),
                              s.jerk.x(), s.jerk.y(), s.jerk.z()};
        std::fwrite(v, sizeof(v), 1, out);
      }
  }
  std::fclose(out);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 3;
  int32_t hi[16];
  double hd[32];
  if (std::fread(hi, sizeof(hi), 1, f) != 1 || std::fread(hd, sizeof(hd), 1, f) != 1) return 3;
  const int n_occ = hi[8], n_unk = hi[9], n = hi[7];
  std::vector<double> raw((size_t)3 * (n_occ + n_unk) + (size_t)9 * n);
  if (std::fread(raw.data(), sizeof(double), raw.size(), f) != raw.size()) return 3;
  std::fclose(f);
  std::vector<V3> occ, unk;
  for (int i = 0; i < n_occ; i++) occ.push_back(V3(raw[3 * i], raw[3 * i + 1], raw[3 * i + 2]));
  for (int i = 0; i < n_unk; i++) unk.push_back(V3(raw[3 * (size_t)(n_occ + i)], raw[3 * (size_t)(n_occ + i) + 1], raw[3 * (size_t)(n_occ + i) + 2]));
  std::vector<double> pairs(raw.begin() + 3 * (size_t)(n_occ + n_unk), raw.end());
#ifdef WITH_ORACLE
  if (argc > 3) {
    OracleSolver::lib_path() = argv[3];
    return run<OracleSolver>(hi, hd, occ, unk, pairs, argv[2]);
  }
#endif
  return run<SolverHip>(hi, hd, occ, unk, pairs, argv[2]);
}
