// solver_hip.hpp — `SolverHip`: the SolverGurobi class surface on top of the C ABI (include/fasterhip.h).
//
// Drop-in for the two members `SolverGurobi sg_whole_, sg_safe_` of class Faster
// (/root/reference/faster/include/faster.hpp:74-75).  Same method names, argument meaning, public data and
// error behaviour as /root/reference/faster/include/solverGurobi.hpp:61-137:
//   * configuration calls of Faster::Faster (faster/src/faster.cpp:52-71): setN, createVars, setDC, setBounds,
//     setForceFinalConstraint, setFactorInitialAndFinalAndIncrement, setVerbose, setThreads, setWMax;
//   * per-replan calls (faster.cpp:306-307, :406-408, :418, :427, :521-527, :536): ResetToNormalState, setX0,
//     setXf, setPolytopes, genNewTraj, fillX;
//   * public data the caller reads AND writes: X_temp_ (faster.cpp:430, :438, :468, :475, :537, :548),
//     factor_that_worked_ (:582, :586); also dt_, trials_, temporal_, runtime_ms_, N_, cb_.
// genNewTraj() returns false (never throws) when no factor in the window yields an optimal solution or when the
// device reports an error (message on stderr), matching "solved <=> GRB_OPTIMAL" (solverGurobi.cpp:580-581).
//
// Where the reference rebuilds a Gurobi model per trial, this class only records inputs; one genNewTraj() is one
// fh_solve_batch() of batch size 1.  For throughput use the batch entry points directly (genNewTrajBatch below).
#pragma once
#include <string>
#include <vector>

#include "faster_stub.hpp"
#include "fasterhip.h"

// stand-in for `mycallback` (solverGurobi.hpp:50-59): only the flag is observable from outside
struct solverhip_callback {
  bool should_terminate_ = false;
};

class SolverHip {
public:
  SolverHip();
  virtual ~SolverHip();
  SolverHip(const SolverHip&) = delete;
  SolverHip& operator=(const SolverHip&) = delete;

  void setN(int N);
  void setX0(state& data);
  void setXf(state& data);
  void resetX();
  void setBounds(double max_values[3]);
  bool genNewTraj();
  double getDTInitial();  // the value used by the LAST genNewTraj (dt_ / factor of its last trial), 0 before
  void setDC(double dc);
  void setPolytopes(std::vector<LinearConstraint3D> polytopes);
  void fillX();
  void setForceFinalConstraint(bool forceFinalConstraint);
  void setFactorInitialAndFinalAndIncrement(double factor_initial, double factor_final, double factor_increment);

  // configuration calls that only made sense for Gurobi: accepted, no effect
  void createVars() {}
  void setMaxConstraints() {}
  void setThreads(int) {}
  // Extension (no reference counterpart): run the factor line search of genNewTraj `width` factors at a time on separate
  // wavefronts (fh_solve_batch_speculative): same result as the sequential search, lower latency for a single replan.
  void setConcurrentFactors(int width) { concurrent_factors_ = width < 1 ? 1 : width; }
  void setVerbose(int verbose) { verbose_ = verbose; }
  void setWMax(double w_max) { w_max_ = w_max; }  // isWmaxSatisfied is commented out in the reference (:459-462)
  void setMode(int mode) { mode_ = mode; }

  void StopExecution();       // solverGurobi.cpp:30-34
  void ResetToNormalState();  // solverGurobi.cpp:36-39

  // ---- extras (not in the reference class) ----
  double cost() const { return last_.cost; }                // ObjVal of the accepted trial
  const fh_result& result() const { return last_; }         // coefficients, assignment, diagnostics
  int deviceStatus() const { return device_rc_; }           // FH_OK or the last FH_ERR_* code
  const std::string& deviceError() const { return device_err_; }
  // Many independent solver objects in ONE device launch (e.g. sg_whole_ of many agents): returns per-solver flags.
  static std::vector<bool> genNewTrajBatch(const std::vector<SolverHip*>& solvers);

  // ---- public data, as in the reference ----
  std::vector<state> X_temp_;
  double dt_ = 0;
  int trials_ = 0;
  int temporal_ = 0;
  double runtime_ms_ = 0;
  double factor_that_worked_ = 0;
  int N_ = 10;
  solverhip_callback cb_;

protected:
  // The two places where the class crosses the C ABI.  Virtual so that CPU-only test harnesses can substitute another
  // implementation of the same two calls (tests/cpp/oracle_solver.hpp); the product never overrides them.
  virtual int solveProblems(const fh_problem* problems, const fh_face* faces, int64_t n_faces, int n, fh_result* results);
  virtual int sampleProblems(const fh_problem* problems, const fh_result* results, int n, int max_samples, fh_state* states,
                             int32_t* counts);
  void fillProblem(fh_problem& pr, std::vector<fh_face>& faces, int face_begin) const;
  void absorb(const fh_result& r);
  bool ensureContext();

  double xf_[9];
  double x0_[9];
  double v_max_ = 5, a_max_ = 3, j_max_ = 5;  // constructor defaults of the reference (solverGurobi.cpp:45-47)
  double DC = 0.01;
  std::vector<LinearConstraint3D> polytopes_;
  bool forceFinalConstraint_ = true;
  double factor_initial_ = 2, factor_final_ = 2, factor_increment_ = 2;  // solverGurobi.hpp:178-180
  int concurrent_factors_ = 1;
  double w_max_ = 1;
  int mode_ = 0;
  int verbose_ = 0;

  fh_ctx* ctx_ = nullptr;
  bool create_failed_ = false;  // fh_create failed once: every later call reports it without retrying
  fh_result last_;
  int device_rc_ = FH_OK;
  std::string device_err_;
};
