// TEST INFRASTRUCTURE: a SolverHip whose two C-ABI calls are answered by the CPU oracle (oracle/liboracle.so), so that the
// replan stub can be exercised without a GPU.  Never part of the product.
#pragma once
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <string>

#include "solver_hip.hpp"

class OracleSolver : public SolverHip {
public:
  static std::string& lib_path() { static std::string p; return p; }

protected:
  typedef void (*solve_fn)(const fh_problem*, const fh_face*, const fh_params*, fh_result*);
  typedef int (*sample_fn)(const fh_problem*, const fh_result*, int, fh_state*);
  static void* handle() {
    static void* h = dlopen(lib_path().c_str(), RTLD_NOW);
    if (!h) { std::fprintf(stderr, "cannot load %s: %s\n", lib_path().c_str(), dlerror()); std::exit(3); }
    return h;
  }
  int solveProblems(const fh_problem* problems, const fh_face* faces, int64_t, int n, fh_result* results) override {
    static solve_fn f = (solve_fn)dlsym(handle(), "orc_solve");
    fh_params par;
    fh_default_params(&par);
    for (int i = 0; i < n; i++) f(&problems[i], faces, &par, &results[i]);
    return FH_OK;
  }
  int sampleProblems(const fh_problem* problems, const fh_result* results, int n, int max_samples, fh_state* states,
                     int32_t* counts) override {
    static sample_fn f = (sample_fn)dlsym(handle(), "orc_sample");
    for (int i = 0; i < n; i++) counts[i] = f(&problems[i], &results[i], max_samples, states + (size_t)i * max_samples);
    return FH_OK;
  }
};
