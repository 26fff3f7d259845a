// a12: SolverHip::StopExecution() from another thread while genNewTraj() is inside the device launch, as
// SolverGurobi::StopExecution() is meant to be used (/root/reference/faster/src/solverGurobi.cpp:15-39: the flag is polled in a
// Gurobi callback while m.optimize() runs).  Input: a scenario with ONE hard whole problem (written by tests/test_gpu_round2.py:
// N, dc, v/a/j max, factor increment, x0 (9), xf (3), polytopes).  Prints STOP_OK and the latency of the stop.
#include <chrono>
#include <cstdio>
#include <fstream>
#include <thread>
#include <vector>

#include "solver_hip.hpp"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::ifstream in(argv[1]);
  int N, P;
  double dc, vmax, amax, jmax, inc, v[9];
  in >> N >> dc >> vmax >> amax >> jmax >> inc;
  state A, E;
  for (double& x : v) in >> x;
  A.setPos(v[0], v[1], v[2]); A.setVel(v[3], v[4], v[5]); A.setAccel(v[6], v[7], v[8]);
  in >> v[0] >> v[1] >> v[2];
  E.setPos(v[0], v[1], v[2]);
  in >> P;
  std::vector<LinearConstraint3D> polys;
  for (int p = 0; p < P; p++) {
    int F;
    in >> F;
    fhstub::MatX3 Am((size_t)F);
    fhstub::VecX b((size_t)F);
    for (int f = 0; f < F; f++) in >> Am(f, 0) >> Am(f, 1) >> Am(f, 2) >> b(f);
    polys.push_back(LinearConstraint3D(Am, b));
  }
  if (!in) return 3;
  SolverHip sg;
  sg.setN(N);
  sg.createVars();
  sg.setDC(dc);
  double bounds[3] = {vmax, amax, jmax};
  sg.setBounds(bounds);
  sg.setForceFinalConstraint(true);
  sg.setFactorInitialAndFinalAndIncrement(1, 10, inc);
  sg.setX0(A);
  sg.setXf(E);
  sg.setPolytopes(polys);
  // warm-up on an easy window (context creation, allocations), then the hard search
  sg.setFactorInitialAndFinalAndIncrement(9, 10, 1);
  (void)sg.genNewTraj();
  sg.setFactorInitialAndFinalAndIncrement(1, 10, inc);
  using clk = std::chrono::steady_clock;
  clk::time_point t_stop;
  std::thread stopper([&] {
    std::this_thread::sleep_for(std::chrono::milliseconds(150));
    t_stop = clk::now();
    sg.StopExecution();
  });
  const clk::time_point t0 = clk::now();
  const bool ok = sg.genNewTraj();
  const clk::time_point t1 = clk::now();
  stopper.join();
  const double total_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
  const double lat_ms = std::chrono::duration<double, std::milli>(t1 - t_stop).count();
  std::printf("solved %d status %d trials %d total %.2f ms, returned %.3f ms after StopExecution()\n", ok ? 1 : 0, sg.result().status, sg.trials_,
              total_ms, lat_ms);
  if (ok || sg.result().status != FH_ST_INTERRUPTED || total_ms < 140.0 || lat_ms > 20.0) {
    std::printf("STOP_FAILED\n");
    return 1;
  }
  if (sg.cb_.should_terminate_) { std::printf("flag not reset (solverGurobi.cpp:474)\n"); return 1; }
  // the flag was cleared at the end of genNewTraj, as in the reference (:474): the next call solves normally
  sg.setFactorInitialAndFinalAndIncrement(9, 10, 1);
  (void)sg.genNewTraj();
  if (sg.result().status == FH_ST_INTERRUPTED) { std::printf("still interrupted after reset\n"); return 1; }
  std::printf("STOP_OK latency_ms %.3f\n", lat_ms);
  return 0;
}
