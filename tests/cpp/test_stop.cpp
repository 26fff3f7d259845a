// a12: SolverHip::StopExecution() from another thread while genNewTraj() is inside the device launch, as
// SolverGurobi::StopExecution() is meant to be used (/root/reference/faster/src/solverGurobi.cpp:15-39: the flag is polled in a
// Gurobi callback while m.optimize() runs; :474 resets it at the end of genNewTraj).
// Input: a scenario with ONE whole problem that no factor solves (written by tests/test_zz_gpu_timing.py: N, dc, v/a/j max, x0 (9),
// xf (3), polytopes), the delay after which the stopper fires and the time an un-stopped search must at least take (ms).
// Nothing here assumes that the problem is slow: the program makes the factor increment smaller (more refuted trials) until a
// genNewTraj() is still running when a watchdog stops it after `min_run_ms`, and only then runs the test proper with the short
// delay.  Exit code 77: no increment of the ladder was long enough (the caller skips).  Prints STOP_OK and the latency of the stop.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <thread>
#include <vector>

#include "solver_hip.hpp"

using clk = std::chrono::steady_clock;

struct Run { bool ok; double total_ms, lat_ms; bool during; };  // during: the stop request arrived while genNewTraj() was running

// one genNewTraj() with a thread that calls StopExecution() after delay_ms
static Run run_with_stop(SolverHip& sg, double delay_ms) {
  clk::time_point t_stop;
  const clk::time_point t0 = clk::now();
  std::thread stopper([&] {
    std::this_thread::sleep_for(std::chrono::duration<double, std::milli>(delay_ms));
    t_stop = clk::now();
    sg.StopExecution();
  });
  Run r;
  r.ok = sg.genNewTraj();
  const clk::time_point t1 = clk::now();
  stopper.join();
  r.total_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
  r.lat_ms = std::chrono::duration<double, std::milli>(t1 - t_stop).count();
  r.during = t_stop < t1;
  // a request that arrives after the search has ended (after genNewTraj() returned, or in its last microseconds) leaves the flag raised —
  // in the reference as well (:30-34 sets it, only the end of a genNewTraj :474 or ResetToNormalState :36-39 clears it): the caller
  // clears it, as Faster::replan does before every solve (:306-307).  Only a search that WAS interrupted owes the reset of :474.
  r.during = r.during && sg.result().status == FH_ST_INTERRUPTED;
  if (!r.during) sg.ResetToNormalState();
  return r;
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const double delay_ms = std::atof(argv[2]), min_run_ms = std::atof(argv[3]);
  std::ifstream in(argv[1]);
  int N, P;
  double dc, vmax, amax, jmax, v[9];
  in >> N >> dc >> vmax >> amax >> jmax;
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
  sg.setX0(A);
  sg.setXf(E);
  sg.setPolytopes(polys);
  // warm-up on an easy window (context creation, allocations)
  sg.setFactorInitialAndFinalAndIncrement(9, 10, 1);
  (void)sg.genNewTraj();
  // the ladder: which increment makes the search run for min_run_ms?  Measured with a watchdog, never assumed.
  const double ladder[] = {0.5, 0.05, 0.005, 0.0025};
  double inc = 0;
  for (double cand : ladder) {
    sg.setFactorInitialAndFinalAndIncrement(1, 10, cand);
    const Run w = run_with_stop(sg, min_run_ms);
    std::printf("increment %g: %s after %.1f ms (status %d, trials %d)\n", cand,
                sg.result().status == FH_ST_INTERRUPTED ? "still running, stopped by the watchdog" : "finished", w.total_ms, sg.result().status, sg.trials_);
    if (w.during && sg.cb_.should_terminate_) { std::printf("flag not reset (solverGurobi.cpp:474)\n"); return 1; }
    if (w.during && !w.ok && sg.result().status == FH_ST_INTERRUPTED) { inc = cand; break; }
  }
  if (inc == 0) { std::printf("STOP_SKIPPED: every search of the ladder ends within %.0f ms\n", min_run_ms); return 77; }
  sg.setFactorInitialAndFinalAndIncrement(1, 10, inc);
  const Run r = run_with_stop(sg, delay_ms);
  std::printf("solved %d status %d trials %d total %.2f ms, returned %.3f ms after StopExecution()\n", r.ok ? 1 : 0, sg.result().status, sg.trials_,
              r.total_ms, r.lat_ms);
  if (!r.during || r.ok || sg.result().status != FH_ST_INTERRUPTED || r.lat_ms < 0.0 || r.lat_ms > 20.0) {
    std::printf("STOP_FAILED\n");
    return 1;
  }
  if (sg.cb_.should_terminate_) { std::printf("flag not reset (solverGurobi.cpp:474)\n"); return 1; }
  // the flag was cleared at the end of genNewTraj, as in the reference (:474): the next call solves normally
  sg.setFactorInitialAndFinalAndIncrement(9, 10, 1);
  (void)sg.genNewTraj();
  if (sg.result().status == FH_ST_INTERRUPTED) { std::printf("still interrupted after reset\n"); return 1; }
  std::printf("STOP_OK latency_ms %.3f\n", r.lat_ms);
  return 0;
}
