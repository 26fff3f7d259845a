// INTEGRATION.md §1 as a test: SolverHip compiled with -DFASTER_HIP_USE_REFERENCE_TYPES against the reference's OWN types —
// `state` (/root/reference/faster/include/faster_types.hpp:79-165) and DecompUtil's LinearConstraint3D
// (thirdparty/DecompROS/DecompUtil/include/decomp_geometry/polyhedron.h:115-185), both included where they lie, Eigen through the
// test-only shim of oracle/ref_frontend/shim — and driven through one replan's calls (faster/src/faster.cpp:406-427, :430):
// setX0 / setXf / setPolytopes / genNewTraj / fillX.  The two C-ABI calls are answered by the CPU oracle (tests/cpp/oracle_solver.hpp:
// test infrastructure), so this runs without a GPU; the known answer is SURVEY.md App. B KA-1.
#define FASTER_HIP_USE_REFERENCE_TYPES
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <vector>

#include "oracle_solver.hpp"  // -> solver_hip.hpp -> faster_stub.hpp, which includes the reference's headers under the define

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  OracleSolver::lib_path() = argv[1];
  std::ifstream in(argv[2]);
  double x0[3], xf[3];
  in >> x0[0] >> x0[1] >> x0[2] >> xf[0] >> xf[1] >> xf[2];
  int P;
  in >> P;
  std::vector<LinearConstraint3D> polys;
  for (int p = 0; p < P; p++) {
    int F;
    in >> F;
    Eigen::Matrix<double, Eigen::Dynamic, 3> A(F, 3);
    Eigen::Matrix<double, Eigen::Dynamic, 1> b(F);
    for (int f = 0; f < F; f++) in >> A(f, 0) >> A(f, 1) >> A(f, 2) >> b(f);
    polys.push_back(LinearConstraint3D(A, b));
  }
  if (!in) return 3;
  state A0, E;
  A0.setPos(x0[0], x0[1], x0[2]);
  E.setPos(xf[0], xf[1], xf[2]);
  // the reference's own inside() on its own type: x0 in the first polytope only, xf in the last only
  const bool geometry = polys.front().inside(A0.pos) && !polys.back().inside(A0.pos) && polys.back().inside(E.pos) && !polys.front().inside(E.pos);

  OracleSolver sg_whole_;
  double max_values[3] = {5, 3, 5};  // constructor defaults of the reference (solverGurobi.cpp:45-47)
  sg_whole_.setN(10); sg_whole_.createVars(); sg_whole_.setDC(0.01); sg_whole_.setBounds(max_values);
  sg_whole_.setForceFinalConstraint(true); sg_whole_.setFactorInitialAndFinalAndIncrement(1, 10, 1);
  sg_whole_.setX0(A0); sg_whole_.setXf(E); sg_whole_.setPolytopes(polys);
  const bool solved = sg_whole_.genNewTraj();
  sg_whole_.fillX();
  const state& last = sg_whole_.X_temp_.back();
  std::printf("{\"geometry\": %d, \"solved\": %d, \"trials\": %d, \"factor\": %.17g, \"dt\": %.17g, \"cost\": %.17g, \"n\": %zu, "
              "\"first\": [%.17g, %.17g, %.17g], \"last\": [%.17g, %.17g, %.17g], \"last_vel_norm\": %.17g}\n",
              geometry ? 1 : 0, solved ? 1 : 0, sg_whole_.trials_, sg_whole_.factor_that_worked_, sg_whole_.dt_, sg_whole_.cost(),
              sg_whole_.X_temp_.size(), sg_whole_.X_temp_.front().pos.x(), sg_whole_.X_temp_.front().pos.y(),
              sg_whole_.X_temp_.front().pos.z(), last.pos.x(), last.pos.y(), last.pos.z(), last.vel.norm());
  return 0;
}
