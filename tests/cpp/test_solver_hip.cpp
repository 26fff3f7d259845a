// Drives SolverHip through the exact call sequence Faster uses on SolverGurobi:
//   construction/configuration ... faster/src/faster.cpp:52-71
//   per-replan whole solve ........ faster.cpp:306-307, :406-408, :418, :427
//   whole -> safe hand-off ........ faster.cpp:475 (R = X_whole[k]), :521-527, :536
//   factor window adaptation ...... faster.cpp:582-588
// Input: a plain-number scenario file (written by tests/test_host_class.py). Output: one JSON object on stdout.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <vector>

#include "solver_hip.hpp"

static std::vector<LinearConstraint3D> read_polys(std::istream& in) {
  int P;
  in >> P;
  std::vector<LinearConstraint3D> out;
  for (int p = 0; p < P; p++) {
    int F;
    in >> F;
    fhstub::MatX3 A((size_t)F);
    fhstub::VecX b((size_t)F);
    for (int f = 0; f < F; f++) in >> A(f, 0) >> A(f, 1) >> A(f, 2) >> b(f);
    out.push_back(LinearConstraint3D(A, b));
  }
  return out;
}

static void dump(const char* name, SolverHip& s, bool solved, bool last) {
  std::printf("\"%s\": {\"solved\": %d, \"trials\": %d, \"factor\": %.17g, \"dt\": %.17g, \"cost\": %.17g, \"n\": %zu, \"runtime_ms\": %.3f",
              name, solved ? 1 : 0, s.trials_, s.factor_that_worked_, s.dt_, s.cost(), s.X_temp_.size(), s.runtime_ms_);
  if (solved && !s.X_temp_.empty()) {
    const state& a = s.X_temp_.front();
    const state& z = s.X_temp_.back();
    std::printf(", \"first\": [%.17g, %.17g, %.17g, %.17g, %.17g, %.17g], \"last\": [%.17g, %.17g, %.17g, %.17g, %.17g, %.17g]",
                a.pos.x(), a.pos.y(), a.pos.z(), a.vel.x(), a.vel.y(), a.vel.z(), z.pos.x(), z.pos.y(), z.pos.z(), z.vel.x(),
                z.vel.y(), z.jerk.z());
  }
  std::printf("}%s\n", last ? "" : ",");
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::ifstream in(argv[1]);
  int N_whole, N_safe;
  double dc, vmax, amax, jmax, inc_whole, inc_safe, gamma, gammap;
  in >> N_whole >> N_safe >> dc >> vmax >> amax >> jmax >> inc_whole >> inc_safe >> gamma >> gammap;
  state A, E, M;
  double v[9];
  for (double& x : v) in >> x;
  A.setPos(v[0], v[1], v[2]); A.setVel(v[3], v[4], v[5]); A.setAccel(v[6], v[7], v[8]);
  in >> v[0] >> v[1] >> v[2];
  E.setPos(v[0], v[1], v[2]);
  std::vector<LinearConstraint3D> l_constraints_whole_ = read_polys(in);
  in >> v[0] >> v[1] >> v[2];
  M.setPos(v[0], v[1], v[2]);
  std::vector<LinearConstraint3D> l_constraints_safe_ = read_polys(in);
  if (!in) return 3;

  SolverHip sg_whole_, sg_safe_;
  double max_values[3] = {vmax, amax, jmax};
  sg_whole_.setN(N_whole); sg_whole_.createVars(); sg_whole_.setDC(dc); sg_whole_.setBounds(max_values);
  sg_whole_.setForceFinalConstraint(true); sg_whole_.setFactorInitialAndFinalAndIncrement(1, 10, inc_whole);
  sg_whole_.setVerbose(0); sg_whole_.setThreads(0); sg_whole_.setWMax(1.0);
  sg_safe_.setN(N_safe); sg_safe_.createVars(); sg_safe_.setDC(dc); sg_safe_.setBounds(max_values);
  sg_safe_.setForceFinalConstraint(false); sg_safe_.setFactorInitialAndFinalAndIncrement(1, 10, inc_safe);
  sg_safe_.setVerbose(0); sg_safe_.setThreads(0); sg_safe_.setWMax(1.0);

  std::printf("{\n");
  // a cancelled solve: StopExecution() before genNewTraj => no trial, false, flag cleared (solverGurobi.cpp:445,474)
  sg_whole_.setX0(A); sg_whole_.setXf(E); sg_whole_.setPolytopes(l_constraints_whole_);
  sg_whole_.StopExecution();
  bool cancelled = sg_whole_.genNewTraj();
  std::printf("\"cancelled\": {\"ret\": %d, \"trials\": %d, \"flag_after\": %d},\n", cancelled ? 1 : 0, sg_whole_.trials_,
              sg_whole_.cb_.should_terminate_ ? 1 : 0);

  sg_whole_.ResetToNormalState();
  sg_safe_.ResetToNormalState();
  sg_whole_.setX0(A);
  sg_whole_.setXf(E);
  sg_whole_.setPolytopes(l_constraints_whole_);
  bool solved_whole = sg_whole_.genNewTraj();
  if (solved_whole) sg_whole_.fillX();
  dump("whole", sg_whole_, solved_whole, false);
  bool solved_safe = false;
  if (solved_whole) {
    state R = sg_whole_.X_temp_[sg_whole_.X_temp_.size() / 2];
    std::printf("\"R\": [%.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g],\n", R.pos.x(), R.pos.y(), R.pos.z(),
                R.vel.x(), R.vel.y(), R.vel.z(), R.accel.x(), R.accel.y(), R.accel.z());
    sg_safe_.setX0(R);
    sg_safe_.setXf(M);
    sg_safe_.setPolytopes(l_constraints_safe_);
    sg_safe_.setForceFinalConstraint(false);
    solved_safe = sg_safe_.genNewTraj();
    if (solved_safe) sg_safe_.fillX();
  }
  dump("safe", sg_safe_, solved_safe, false);
  // faster.cpp:582-588
  double new_init_whole = std::max(sg_whole_.factor_that_worked_ - gamma, 1.0);
  double new_final_whole = sg_whole_.factor_that_worked_ + gammap;
  sg_whole_.setFactorInitialAndFinalAndIncrement(new_init_whole, new_final_whole, inc_whole);
  bool again = sg_whole_.genNewTraj();
  std::printf("\"window\": [%.17g, %.17g],\n", new_init_whole, new_final_whole);
  dump("whole_again", sg_whole_, again, false);
  // extension: the same line search, three factors at a time (fh_solve_batch_speculative) — identical outputs expected
  sg_whole_.setFactorInitialAndFinalAndIncrement(1, 10, inc_whole);
  sg_whole_.setConcurrentFactors(3);
  bool conc = sg_whole_.genNewTraj();
  if (conc) sg_whole_.fillX();
  dump("whole_concurrent", sg_whole_, conc, true);
  std::printf("}\n");
  return 0;
}
