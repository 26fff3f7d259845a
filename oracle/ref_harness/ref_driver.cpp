// ref_driver.cpp — drives the UNTOUCHED reference class SolverGurobi (/root/reference/faster/src/solverGurobi.cpp, compiled from where
// it lies by build.sh) over a list of problems and prints what genNewTraj() leaves behind, one JSON object per problem.
// TEST INFRASTRUCTURE (oracle/): the tie-breaker of SURVEY.md 8(c)(5), usable only where a Gurobi installation + licence and Eigen3
// exist (GUROBI_HOME, EIGEN3_INCLUDE_DIR).  Call sequence = Faster::Faster + Faster::replan (faster/src/faster.cpp:52-71, :406-418).
//
// Input (plain numbers, written by diff_ref.py): n_problems, then per problem
//   N force_final dc v_max a_max j_max f_init f_final f_inc  x0[9]  xf[9]  P  { F  { a_x a_y a_z b } x F } x P
#include <cstdio>
#include <fstream>
#include <vector>

#include "solverGurobi.hpp"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::ifstream in(argv[1]);
  int n;
  in >> n;
  std::printf("[\n");
  for (int i = 0; i < n; i++) {
    int N, force, P;
    double dc, vaj[3], f0, f1, fi, x0[9], xf[9];
    in >> N >> force >> dc >> vaj[0] >> vaj[1] >> vaj[2] >> f0 >> f1 >> fi;
    for (double& v : x0) in >> v;
    for (double& v : xf) in >> v;
    in >> P;
    std::vector<LinearConstraint3D> polys;
    for (int p = 0; p < P; p++) {
      int F;
      in >> F;
      Eigen::Matrix<double, Eigen::Dynamic, 3> A(F, 3);
      Eigen::VectorXd b(F);
      for (int f = 0; f < F; f++) in >> A(f, 0) >> A(f, 1) >> A(f, 2) >> b(f);
      polys.push_back(LinearConstraint3D(A, b));  // polyhedron.h:123
    }
    if (!in) return 3;
    SolverGurobi sg;  // a fresh object per problem: Gurobi may warm-start a modified model from its previous solution
    sg.setN(N);
    sg.createVars();
    sg.setDC(dc);
    sg.setBounds(vaj);
    sg.setForceFinalConstraint(force != 0);
    sg.setFactorInitialAndFinalAndIncrement(f0, f1, fi);
    sg.setVerbose(0);
    sg.setThreads(1);
    state A0, E;
    A0.setPos(x0[0], x0[1], x0[2]); A0.setVel(x0[3], x0[4], x0[5]); A0.setAccel(x0[6], x0[7], x0[8]);
    E.setPos(xf[0], xf[1], xf[2]); E.setVel(xf[3], xf[4], xf[5]); E.setAccel(xf[6], xf[7], xf[8]);
    sg.setX0(A0);
    sg.setXf(E);
    sg.setPolytopes(polys);
    const bool ok = sg.genNewTraj();
    if (ok) sg.fillX();
    std::printf(" {\"solved\": %d, \"trials\": %d, \"factor\": %.17g, \"dt\": %.17g, \"runtime_ms\": %.3f, \"n_samples\": %zu", ok ? 1 : 0,
                sg.trials_, sg.factor_that_worked_, sg.dt_, sg.runtime_ms_, sg.X_temp_.size());
    if (ok) {
      // every sample of fillX (position + jerk): diffed against fh_sample_batch / the oracle; the objective is sum |jerk|^2 over segments
      std::printf(", \"samples\": [");
      for (size_t k = 0; k < sg.X_temp_.size(); k++) {
        const state& s = sg.X_temp_[k];
        std::printf("%s[%.17g, %.17g, %.17g, %.17g, %.17g, %.17g]", k ? ", " : "", s.pos.x(), s.pos.y(), s.pos.z(), s.jerk.x(), s.jerk.y(),
                    s.jerk.z());
      }
      std::printf("]");
    }
    std::printf("}%s\n", i + 1 < n ? "," : "");
  }
  std::printf("]\n");
  return 0;
}
