// solver_hip.cpp — see solver_hip.hpp.  Host C++ only; the arithmetic is behind the C ABI.
#include "solver_hip.hpp"

#include <chrono>
#include <cstdio>
#include <cstring>

SolverHip::SolverHip() {
  std::memset(xf_, 0, sizeof(xf_));
  std::memset(x0_, 0, sizeof(x0_));
  std::memset(&last_, 0, sizeof(last_));
}

SolverHip::~SolverHip() {
  if (ctx_) fh_destroy(ctx_);
}

bool SolverHip::ensureContext() {
  // creation status and per-call status are separate: a transient error of one solve (device_rc_) must not disable the
  // solver object for good, and a failed creation is not retried on every replan
  if (create_failed_) return false;
  if (ctx_) return true;
  if (fh_abi_version() != FH_ABI_VERSION) {  // this object was compiled against another generation of fasterhip.h than the library that is loaded
    create_failed_ = true;
    device_rc_ = FH_ERR_ARG;
    device_err_ = "libfasterhip.so has struct layout generation " + std::to_string(fh_abi_version()) + ", SolverHip was built for " + std::to_string(FH_ABI_VERSION);
    std::fprintf(stderr, "SolverHip: %s\n", device_err_.c_str());
    return false;
  }
  const int rc = fh_create(&ctx_, -1);
  if (rc != FH_OK) {
    create_failed_ = true;
    device_rc_ = rc;
    device_err_ = ctx_ ? fh_last_error(ctx_) : "fh_create failed";
    std::fprintf(stderr, "SolverHip: %s\n", device_err_.c_str());
    return false;
  }
  if (cb_.should_terminate_) fh_request_stop(ctx_);  // StopExecution() came before the first solve
  return true;
}

void SolverHip::setN(int N) { N_ = N; }
void SolverHip::setDC(double dc) { DC = dc; }

void SolverHip::setBounds(double max_values[3]) {  // idempotent (the reference would duplicate rows, :409-416)
  v_max_ = max_values[0];
  a_max_ = max_values[1];
  j_max_ = max_values[2];
}

void SolverHip::setForceFinalConstraint(bool forceFinalConstraint) { forceFinalConstraint_ = forceFinalConstraint; }

void SolverHip::setFactorInitialAndFinalAndIncrement(double factor_initial, double factor_final, double factor_increment) {
  factor_initial_ = factor_initial;
  factor_final_ = factor_final;
  factor_increment_ = factor_increment;
}

void SolverHip::setX0(state& data) {  // order of solverGurobi.cpp:304-312
  x0_[0] = data.pos.x(); x0_[1] = data.pos.y(); x0_[2] = data.pos.z();
  x0_[3] = data.vel.x(); x0_[4] = data.vel.y(); x0_[5] = data.vel.z();
  x0_[6] = data.accel.x(); x0_[7] = data.accel.y(); x0_[8] = data.accel.z();
}

void SolverHip::setXf(state& data) {
  xf_[0] = data.pos.x(); xf_[1] = data.pos.y(); xf_[2] = data.pos.z();
  xf_[3] = data.vel.x(); xf_[4] = data.vel.y(); xf_[5] = data.vel.z();
  xf_[6] = data.accel.x(); xf_[7] = data.accel.y(); xf_[8] = data.accel.z();
}

void SolverHip::setPolytopes(std::vector<LinearConstraint3D> polytopes) { polytopes_ = polytopes; }

void SolverHip::StopExecution() {  // may be called from another thread while genNewTraj() is inside the device launch
  cb_.should_terminate_ = true;
  if (ctx_ && !create_failed_) fh_request_stop(ctx_);  // the kernels poll this word between branch-and-bound nodes
  std::printf("Activated flag to stop execution\n");
}

void SolverHip::ResetToNormalState() {
  cb_.should_terminate_ = false;
  if (ctx_ && !create_failed_) fh_clear_stop(ctx_);
}

void SolverHip::resetX() {  // solverGurobi.cpp:382-388
  int size = (int)((int)(N_)*dt_ / DC);
  size = (size < 2) ? 2 : size;
  X_temp_.assign((size_t)size, state());
}

double SolverHip::getDTInitial() {
  if (last_.trials <= 0) return 0;
  const double f_last = factor_initial_ + (last_.trials - 1) * factor_increment_;
  return f_last != 0 ? dt_ / f_last : 0;
}

void SolverHip::fillProblem(fh_problem& pr, std::vector<fh_face>& faces, int face_begin) const {
  std::memset(&pr, 0, sizeof(pr));
  pr.n_seg = N_;
  pr.n_poly = (int32_t)polytopes_.size();
  pr.force_final_pos = forceFinalConstraint_ ? 1 : 0;
  pr.face_begin = face_begin;
  pr.dc = DC;
  pr.v_max = v_max_;
  pr.a_max = a_max_;
  pr.j_max = j_max_;
  pr.f_init = factor_initial_;
  pr.f_final = factor_final_;
  pr.f_inc = factor_increment_;
  std::memcpy(pr.x0, x0_, sizeof(x0_));
  std::memcpy(pr.xf, xf_, sizeof(xf_));
  int off = 0;
  const size_t np = polytopes_.size() <= (size_t)FH_MAX_POLY ? polytopes_.size() : (size_t)FH_MAX_POLY;
  for (size_t p = 0; p < np; p++) {
    const auto A = polytopes_[p].A();
    const auto b = polytopes_[p].b();
    for (size_t i = 0; i < (size_t)b.rows(); i++) {
      fh_face f;
      f.a[0] = A(i, 0); f.a[1] = A(i, 1); f.a[2] = A(i, 2);
      f.b = b(i);
      faces.push_back(f);
      off++;
    }
    pr.face_off[p + 1] = off;
  }
  for (size_t p = np; p < (size_t)FH_MAX_POLY; p++) pr.face_off[p + 1] = off;
}

void SolverHip::absorb(const fh_result& r) {
  last_ = r;
  trials_ = r.trials;
  dt_ = r.dt;
  if (r.solved) factor_that_worked_ = r.factor;  // left untouched on failure, as in the reference (:463-467)
  temporal_ += r.trials;                         // callOptimizer increments it once per trial (:560)
  resetX();                                      // the last trial's resetX (:456)
}

int SolverHip::solveProblems(const fh_problem* problems, const fh_face* faces, int64_t n_faces, int n, fh_result* results) {
  if (!ensureContext()) return FH_ERR_DEVICE;
  // concurrent_factors_ <= 1: the sequential line search inside one wavefront; otherwise the same search, `width` factors at a time
  const int rc = fh_solve_batch_speculative(ctx_, problems, faces, n_faces, n, concurrent_factors_, results);
  if (rc != FH_OK) device_err_ = fh_last_error(ctx_);
  return rc;
}

int SolverHip::sampleProblems(const fh_problem* problems, const fh_result* results, int n, int max_samples, fh_state* states,
                              int32_t* counts) {
  if (!ensureContext()) return FH_ERR_DEVICE;
  const int rc = fh_sample_batch(ctx_, problems, results, n, max_samples, states, counts);
  if (rc != FH_OK) device_err_ = fh_last_error(ctx_);
  return rc;
}

bool SolverHip::genNewTraj() {  // solverGurobi.cpp:426-477 for one solver object
  const auto t0 = std::chrono::steady_clock::now();
  trials_ = 0;
  runtime_ms_ = 0;
  std::memset(&last_, 0, sizeof(last_));
  if (factor_initial_ < 1) std::printf("factor_initial_ is less than one, it doesn't make sense\n");  // :438-441
  if (cb_.should_terminate_) {  // the factor loop is not entered (:445); flag cleared at the end (:474)
    ResetToNormalState();
    return false;
  }
  if (polytopes_.size() > (size_t)FH_MAX_POLY) {
    std::fprintf(stderr, "SolverHip: %zu polytopes exceed FH_MAX_POLY=%d\n", polytopes_.size(), FH_MAX_POLY);
    return false;
  }
  fh_problem pr;
  std::vector<fh_face> faces;
  fillProblem(pr, faces, 0);
  fh_result r;
  device_rc_ = solveProblems(&pr, faces.empty() ? nullptr : faces.data(), (int64_t)faces.size(), 1, &r);
  runtime_ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (cb_.should_terminate_) ResetToNormalState();  // a StopExecution() during the solve: `cb_.should_terminate_ = false` (:474)
  if (device_rc_ != FH_OK) {
    std::fprintf(stderr, "SolverHip::genNewTraj: device error %d: %s\n", device_rc_, device_err_.c_str());
    return false;
  }
  absorb(r);
  return r.solved != 0;
}

std::vector<bool> SolverHip::genNewTrajBatch(const std::vector<SolverHip*>& solvers) {
  std::vector<bool> ok(solvers.size(), false);
  if (solvers.empty()) return ok;
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<fh_problem> problems;
  std::vector<fh_face> faces;
  std::vector<size_t> who;
  for (size_t i = 0; i < solvers.size(); i++) {
    SolverHip* s = solvers[i];
    s->trials_ = 0;
    s->runtime_ms_ = 0;
    std::memset(&s->last_, 0, sizeof(s->last_));
    if (s->factor_initial_ < 1) std::printf("factor_initial_ is less than one, it doesn't make sense\n");  // :438-441
    if (s->cb_.should_terminate_) {  // the factor loop is not entered (:445); flag cleared at the end (:474)
      s->ResetToNormalState();
      continue;
    }
    if (s->polytopes_.size() > (size_t)FH_MAX_POLY) {
      std::fprintf(stderr, "SolverHip: %zu polytopes exceed FH_MAX_POLY=%d\n", s->polytopes_.size(), FH_MAX_POLY);
      continue;
    }
    fh_problem pr;
    s->fillProblem(pr, faces, (int)faces.size());
    problems.push_back(pr);
    who.push_back(i);
  }
  if (problems.empty()) return ok;
  SolverHip* lead = solvers[who[0]];
  std::vector<fh_result> results(problems.size());
  int rc = FH_ERR_DEVICE;
  if (lead->ensureContext())
    rc = fh_solve_batch(lead->ctx_, problems.data(), faces.empty() ? nullptr : faces.data(), (int64_t)faces.size(),
                        (int)problems.size(), results.data());
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  for (size_t k = 0; k < who.size(); k++) {
    SolverHip* s = solvers[who[k]];
    s->device_rc_ = rc;  // per-call status only: the followers' own contexts are untouched
    s->runtime_ms_ = ms;
    if (rc != FH_OK) {
      s->device_err_ = lead->ctx_ ? fh_last_error(lead->ctx_) : "no context";
      if (k == 0) std::fprintf(stderr, "SolverHip::genNewTraj: device error %d: %s\n", rc, s->device_err_.c_str());
      continue;
    }
    s->absorb(results[k]);
    ok[who[k]] = results[k].solved != 0;
  }
  return ok;
}

void SolverHip::fillX() {  // solverGurobi.cpp:122-168
  if (!last_.solved || X_temp_.empty()) return;  // the reference would read an unsolved model here (throws)
  fh_problem pr;
  std::vector<fh_face> unused;
  fillProblem(pr, unused, 0);
  const int cap = (int)X_temp_.size();  // honours a caller that resized X_temp_
  std::vector<fh_state> st((size_t)cap);
  int32_t count = 0;
  device_rc_ = sampleProblems(&pr, &last_, 1, cap, st.data(), &count);
  if (device_rc_ != FH_OK) {
    std::fprintf(stderr, "SolverHip::fillX: device error %d: %s\n", device_rc_, device_err_.c_str());
    return;
  }
  const int n = count < cap ? count : cap;
  for (int i = 0; i < n; i++) {
    state s;
    s.setPos(st[i].pos[0], st[i].pos[1], st[i].pos[2]);
    s.setVel(st[i].vel[0], st[i].vel[1], st[i].vel[2]);
    s.setAccel(st[i].accel[0], st[i].accel[1], st[i].accel[2]);
    s.setJerk(st[i].jerk[0], st[i].jerk[1], st[i].jerk[2]);
    X_temp_[(size_t)i] = s;
  }
}
