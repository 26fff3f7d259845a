// fh_solve.hip.hpp — the genNewTraj() kernel for gfx950 (CDNA4, wave64).  Hand-written HIP, FP64.
//
// Replaces, for a whole batch at once, SolverGurobi::genNewTraj() and every callOptimizer() it issues
// (/root/reference/faster/src/solverGurobi.cpp:426-477, :549-657): time allocation (getDTInitial :659-759,
// findDT :494-497), the model the reference rebuilds per trial (:180-291, :332-407, :499-524, :86-120) and the
// MIQP solve that Gurobi performs inside m.optimize() (:566).
//
// Mapping to the machine
//   * one problem (= one genNewTraj call) per 64-lane wavefront, one wavefront per workgroup; the grid is what is
//     resident at once (12 workgroups per CU at N <= 6, 11 at N = 10, 5 at N = 15: LDS limited — fh_capi.hip computes it from the carve —, 3 wavefronts per SIMD by VGPRs) and every workgroup pulls problems from a
//     device-scope ticket counter, so problems of different difficulty balance across the 256 CUs;
//   * control flow is wave-uniform: the whole search (factor loop -> branch and bound -> dual active set) is a
//     scalar program; the 64 lanes are the data-parallel axis inside every step (rows of the constraint scan,
//     rows/columns of the QR factors, (segment, polytope) pairs of the leaf test);
//   * everything a solve touches lives in LDS (thin QR of the active normals, states, control points, the
//     polytope faces staged once per problem with coalesced 32-B face loads); HBM traffic is the compulsory
//     problem read and result write plus the branch-and-bound node snapshots (live prefix only, L2 resident);
//   * a workgroup being one wavefront, LDS traffic is ordered by program order: no barriers, only compiler fences;
//   * the constraint matrix is never formed: rows are (face normal) x (Toeplitz weight of the triple integrator)
//     and are evaluated from the control points (3 FMAs per row).
//
// Algorithm (the model the CPU oracle restates, oracle/faster_oracle.c): jerk-space QP min |x|^2, solved in the REDUCED space
// of the final-state equalities (setConstraintsXf, solverGurobi.cpp:332-357): x = xp + (Z (x) I3) y with Z the orthonormal
// complement of the 2 (safe) or 3 (whole) end-state functionals per axis — it depends on N only, fh_basis.hip.hpp — and xp the
// minimum-norm solution of the equalities, so a QP has 3(N-2) or 3(N-3) unknowns, cost |xp|^2 + |y|^2, and no equality rows in
// its factorisation (N = 10: 24 / 21 unknowns instead of 30 + 6 / 9 rows).  Dual active-set (Goldfarb-Idnani with identity
// Hessian, thin QR by re-orthogonalised Gram-Schmidt, Givens on removal), exact branch and bound over "segment t in polytope p"
// with lazily branched segments; the first child of a node is warm-started from its parent's factorisation (the dual method
// stays dual feasible when rows are added).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fasterhip.h"
#include "fh_basis.hip.hpp"
#include "fh_share.hip.hpp"
#include "fh_sample.hip.hpp"  // the hand-off of a pair: ProblemView / ResultView, glue_* (shared with the staged pipeline)

namespace fh {

// every FH_LOOK_EVERY-th node of a tree the worker reads the control block (stop request, hungry workers): power of two.  The default of a
// launch that has the device to itself; the period of a launch is fh_sched.look_every (ShareArgs.look_mask), which the library sets to
// FH_LOOK_EVERY_BUSY when other solve launches of the process are in flight on the device (fh_capi.hip: launch_solve)
#ifndef FH_LOOK_EVERY
#define FH_LOOK_EVERY 8
#endif
#ifndef FH_LOOK_EVERY_BUSY
#define FH_LOOK_EVERY_BUSY 16
#endif

// A workgroup is ONE wavefront (launch bounds 64): its LDS operations are issued and performed in program order, so what a
// multi-wave kernel would need a barrier for only needs the compiler not to reorder the accesses.
#ifdef FH_SYNC_BARRIER
#define FH_SYNC() __syncthreads()
#elif defined(FH_SYNC_ASM)
#define FH_SYNC() asm volatile("" ::: "memory")
#else
#define FH_SYNC() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")
#endif
// -DFH_PROFILE: per-phase cycle counters (s_memtime) accumulated per problem and written into the unused last
// coefficient row of the result (diagnostic builds only; scripts/phase_profile.py).
// a cycle stamp that the compiler does not move memory operations across (the kernel-loop probes of the diagnostic builds)
__device__ __forceinline__ unsigned long long pinned_clock() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  return t;
}
#ifdef FH_PROFILE
#define FH_T0() const unsigned long long t0__ = __builtin_readcyclecounter()
#define FH_T1(slot) do { prof[slot] += __builtin_readcyclecounter() - t0__; cnt[slot] += 1; } while (0)
// finer slots of round 6 (cycles only; row FH_MAX_SEG - 6 of the result): what search() and run_problem spend OUTSIDE the slots above
#define FH_U0() const unsigned long long u0__ = __builtin_readcyclecounter()
#define FH_U1(slot) do { prof2[slot] += __builtin_readcyclecounter() - u0__; } while (0)
#else
#define FH_T0()
#define FH_T1(slot)
#define FH_U0()
#define FH_U1(slot)
#endif
#define FH_MAX_TRIALS 4096
// The fused pair kernel stores the whole problem's result AFTER the hand-off has issued its loads (a load waits for every store before it:
// s_waitcnt vmcnt counts both).  The diagnostic build that stamps times into the result rows as it goes stores at once.
#ifdef FH_SHARE_PROFILE
#define FH_DEFER_RESULT false
#else
#define FH_DEFER_RESULT true
#endif
// Wavefronts per SIMD the solve kernels are compiled for (launch bounds: 512 / this many vector registers per lane, granule 8)
#ifndef FH_WAVES_PER_SIMD
#define FH_WAVES_PER_SIMD 3
#endif
// Gram-Schmidt is repeated when the first pass cancels more than this fraction of |g|^2 (Daniel-Gragg-Kaufman-Stewart use
// 1/2; a smaller value trades a bounded loss of orthogonality, O(eps / sqrt(threshold)) per inserted row, for half the sweeps)
#ifndef FH_REORTH_THRESHOLD
#define FH_REORTH_THRESHOLD 0.05
#endif
// Tree levels (of a worker's own stack) whose snapshot tail — the parent's optimum, its multipliers and active row ids — and child
// bounds stay in LDS instead of the HBM workspace: going back to a parent whose factor columns are still intact in LDS then costs two
// LDS copies and no memory round trip.  N <= 6 only, where LDS does not limit the resident workgroups: at N = 10 two levels cost a
// workgroup per CU (11 -> 10) and, measured, 3 % of the throughput — with the child bound a tree goes back to a parent 0.6 times per
// problem and a restore is 4 k cycles; the N = 15 / 16 carves have no LDS to spare.
#ifndef FH_TAIL_LEVELS
#define FH_TAIL_LEVELS 0
#endif
// Tickets are drawn FH_TICKET_CHUNK at a time while every workgroup still has that many ahead of it (then 2, then 1), and finished
// units are reported FH_DONE_BATCH at a time: the ticket counter and the `done` counter are single words that every workgroup of the
// launch hits with a device-scope atomic — once per unit each, that is 65 000 same-address atomics per 3.5 ms launch, one every
// 50 ns, which is about what one address sustains: the wavefronts queued behind each other for ~15 us per unit (measured as a
// 35 k-cycle "ticket" phase with nothing in it but one atomic and one load).
#ifndef FH_TICKET_CHUNK
#define FH_TICKET_CHUNK 4
#endif
#ifndef FH_DONE_BATCH
#define FH_DONE_BATCH 4
#endif
#ifndef FH_TICKET_BACKLOG
#define FH_TICKET_BACKLOG 512  // ticket frames (give_tickets) that may wait ahead of the takers (the ring has FH_QCAP = 1024 slots)
#endif
#ifndef FH_TAIL_LEVELS_BIG
#define FH_TAIL_LEVELS_BIG 0
#endif

enum { K_EQ = 0, K_JBOX = 1, K_VBOX = 2, K_ABOX = 3, K_POLY = 4 };
// weight kinds of a row: which linear functional of the state at the start of segment tt
enum { W_P = 0, W_CP1 = 1, W_CP2 = 2, W_V = 3, W_A = 4, W_KINDS = 5 };

__device__ __forceinline__ int mk_id(int kind, int t, int k, int f) { return (kind << 24) | (t << 16) | (k << 8) | f; }

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double uniform_f64(double v) {
  int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Scheduling fences.  in_flight(a): every element of `a` (just loaded) must be in its register here, so the loads are issued
// back to back and their LDS latencies overlap (the scheduler of this register-tight kernel otherwise serialises
// load -> wait -> use chains to save registers).  opaque(v): hides a lane constant from loop-invariant code motion, so that
// values derived from it are recomputed where they are used instead of living in registers across the whole solve.
__device__ __forceinline__ void in_flight(double (&a)[3]) { asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2])); }
__device__ __forceinline__ void in_flight(double (&a)[4]) { asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])); }
__device__ __forceinline__ void in_flight(double (&a)[6]) {
  asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]));
}
__device__ __forceinline__ void in_flight(double (&a)[9]) {
  asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]));
}
__device__ __forceinline__ void in_flight(double (&a)[10]) {
  asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
               "+v"(a[9]));
}
__device__ __forceinline__ void in_flight(double (&a)[12]) {
  asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
               "+v"(a[9]), "+v"(a[10]), "+v"(a[11]));
}
__device__ __forceinline__ void in_flight(double (&a)[15]) {
  asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
               "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]));
}
__device__ __forceinline__ void in_flight(double (&a)[16]) {
  asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
               "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
}
__device__ __forceinline__ double opaque(double v) {
  asm volatile("" : "+v"(v));
  return v;
}
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// Wave64 reductions on the DPP network (row_shr 1/2/4/8 inside each 16-lane row, then row_bcast:15 / row_bcast:31
// across rows; the total lands in lane 63) instead of ds_bpermute round trips through the LDS crossbar.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double identity, double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(__double2loint(identity), lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(__double2hiint(identity), hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
// Zero-filling variant (bound_ctrl, every row written): no `old` operand to initialise.  Rows 0 and 2 pick up partial sums
// they do not need in the row_bcast steps; only lane 63 is read, and it sees the values of before each step.
template <int CTRL>
__device__ __forceinline__ double dpp0_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// v_max_f64 / v_min_f64 without the canonicalising self-max the fmax/fmin lowering adds (no NaNs reach the reductions)
__device__ __forceinline__ double vmax_f64(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double vmin_f64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// lanes l and l+32 exchange a double and add: both halves end up with the same sum (v_permlane32_swap, gfx950)
__device__ __forceinline__ double halves_sum(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp0_f64<0x111>(v);
  v += dpp0_f64<0x112>(v);
  v += dpp0_f64<0x114>(v);
  v += dpp0_f64<0x118>(v);
  v += dpp0_f64<0x142>(v);
  v += dpp0_f64<0x143>(v);
  return readlane_f64(v, 63);
}
__device__ __forceinline__ unsigned wave_or(unsigned v) {
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xf, 0xf, true);
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xf, 0xf, true);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// max(0, max over the lanes): every caller only asks whether the maximum is positive and where it sits
__device__ __forceinline__ double wave_max_nonneg(double v) {
  v = vmax_f64(v, dpp0_f64<0x111>(v));
  v = vmax_f64(v, dpp0_f64<0x112>(v));
  v = vmax_f64(v, dpp0_f64<0x114>(v));
  v = vmax_f64(v, dpp0_f64<0x118>(v));
  v = vmax_f64(v, dpp0_f64<0x142>(v));
  v = vmax_f64(v, dpp0_f64<0x143>(v));
  return readlane_f64(v, 63);
}
__device__ __forceinline__ double wave_min(double v) {
  v = vmin_f64(v, dpp_f64<0x111, 0xf>(INFINITY, v));
  v = vmin_f64(v, dpp_f64<0x112, 0xf>(INFINITY, v));
  v = vmin_f64(v, dpp_f64<0x114, 0xf>(INFINITY, v));
  v = vmin_f64(v, dpp_f64<0x118, 0xf>(INFINITY, v));
  v = vmin_f64(v, dpp_f64<0x142, 0xa>(INFINITY, v));
  v = vmin_f64(v, dpp_f64<0x143, 0xc>(INFINITY, v));
  return readlane_f64(v, 63);
}
__device__ __forceinline__ int first_lane(bool pred) {  // lowest lane with pred, -1 if none (uniform)
  unsigned long long m = __ballot(pred);
  return m ? (int)__builtin_ctzll(m) : -1;
}
__device__ __forceinline__ bool wave_any(bool pred) { return __ballot(pred) != 0ull; }

// coefficient of jerk j_s in functional `kind` of the state at the start of segment tt, m = tt-1-s >= 0
__device__ __forceinline__ double wcoef(int kind, int m, double h) {
  const double dm = (double)m;
  const double cP = h * h * h * (1.0 / 6.0 + 0.5 * dm + 0.5 * dm * dm);
  const double cV = h * h * (0.5 + dm);
  switch (kind) {
    case W_P: return cP;
    case W_CP1: return cP + cV * (h / 3.0);
    case W_CP2: return cP + cV * (2.0 * h / 3.0) + h * (h * h / 6.0);
    case W_V: return cV;
    default: return h;
  }
}

// ---- time allocation: getDTInitial (solverGurobi.cpp:659-759); same closed forms as the oracle ------------------
__device__ inline double polish3(double c3, double c2, double c1, double c0, double t) {
#pragma clang fp contract(off)  // (the exact path rounds as the oracle does: oracle/Makefile builds with -ffp-contract=off)
  for (int it = 0; it < 3; it++) {
    double f = ((c3 * t + c2) * t + c1) * t + c0;
    double df = (3 * c3 * t + 2 * c2) * t + c1;
    if (df == 0 || !isfinite(df)) break;
    double tn = t - f / df;
    if (!isfinite(tn)) break;
    t = tn;
  }
  return t;
}
// One candidate root per lane: lane = 3 * axis + j evaluates root j of its axis' cubic / quadratic (same arithmetic per root as
// the CPU oracle's sequential loops; the minimum over the positive roots and the maximum over the axes do not depend on the
// order).  The double-precision cbrt / acos / cos behind a cubic are hundreds of instructions each: done once per wavefront
// instead of once per axis and root, they are ~4 % instead of ~25 % of a typical problem.  0: no positive root.
__device__ inline double quad_root_of_lane(double c2, double c1, double c0, int j) {
#pragma clang fp contract(off)
  const double disc = c1 * c1 - 4.0 * c2 * c0;
  if (disc < 0) return (j == 0 && sqrt(-disc) / fabs(2.0 * c2) < 1e-12) ? -c1 / (2.0 * c2) : 0.0;
  const double s = sqrt(disc);
  const double qq = -0.5 * (c1 + (c1 >= 0 ? s : -s));
  if (qq == 0) return 0.0;
  return j == 0 ? qq / c2 : (j == 1 ? c0 / qq : 0.0);
}
__device__ inline double cubic_root_of_lane(double c3, double c2, double c1, double c0, int j) {
  // [r6] c0 == 0 exactly (start and goal coincide on this axis): the cubic is t (c3 t^2 + c2 t + c1) — one root EXACTLY zero (lane j = 0;
  // MinPositiveElement drops it) and the quadratic's roots by the quadratic's closed form (lanes 1, 2).  A chosen convention, the same as
  // the oracle's (oracle/faster_oracle.c: real_roots_cubic): the general closed form returns the zero root as +-1e-17 by the last bit of cbrt / acos.
#pragma clang fp contract(off)
  if (c0 == 0.0) return j == 0 ? 0.0 : quad_root_of_lane(c3, c2, c1, j - 1);
  const double B = c2 / c3, C = c1 / c3, D = c0 / c3;
  const double p = C - B * B / 3.0;
  const double q = 2.0 * B * B * B / 27.0 - B * C / 3.0 + D;
  const double disc = q * q / 4.0 + p * p * p / 27.0;
  double cand = 0;
  if (disc > 0) {
    const double sq = sqrt(disc);
    const double u = cbrt(-q / 2.0 + sq), v = cbrt(-q / 2.0 - sq);
    if (j == 0) cand = polish3(c3, c2, c1, c0, u + v - B / 3.0);
    else if (j == 1 && 0.5 * sqrt(3.0) * fabs(u - v) < 1e-12) cand = -(u + v) / 2.0 - B / 3.0;
  } else if (p == 0) {
    if (j == 0) cand = -B / 3.0;
  } else {
    const double m = 2.0 * sqrt(-p / 3.0);
    double arg = 3.0 * q / (p * m);
    arg = fmin(1.0, fmax(-1.0, arg));
    const double phi = acos(arg) / 3.0;
    cand = polish3(c3, c2, c1, c0, m * cos(phi - 2.0 * 3.14159265358979323846 * (double)j / 3.0) - B / 3.0);
  }
  return cand;
}
// getDTInitial as the oracle evaluates it, one candidate root per lane (lane = 3 * axis + j; the exact path: closed forms with the
// double-precision cbrt / acos / cos of the device library, ~1100 vector instructions on nine useful lanes).
// Where a solve reads x0 from: the record (X0Rec: the stand-alone time allocation), or the wavefront's LDS (X0Lds: the solve kernels keep x0
// in the slots of the state arrays that belong to the start of segment 0 — the safe problem of a fused pair has no record to read it from).
template <class PR>
struct X0Rec {
  const PR& pr;
  __device__ __forceinline__ double pos(int i) const { return i == 0 ? pr.x0[0] : (i == 1 ? pr.x0[1] : pr.x0[2]); }
  __device__ __forceinline__ double vel(int i) const { return i == 0 ? pr.x0[3] : (i == 1 ? pr.x0[4] : pr.x0[5]); }
  __device__ __forceinline__ double acc(int i) const { return i == 0 ? pr.x0[6] : (i == 1 ? pr.x0[7] : pr.x0[8]); }
};
struct X0Lds {
  const double* P;  // Pc of the Solver; Vc and Ac follow it at `stride` doubles each (entries 0..2 of the three arrays are x0)
  int stride;
  __device__ __forceinline__ double pos(int i) const { return P[i]; }
  __device__ __forceinline__ double vel(int i) const { return P[stride + i]; }
  __device__ __forceinline__ double acc(int i) const { return P[2 * stride + i]; }
};
// (x0p, x0v, x0a: x0 on the axis of this lane, lane = 3 * axis + j, lanes beyond 8 repeat axis 2)
template <class PR>
__device__ __attribute__((noinline)) double dt_initial_exact(const PR& pr, double x0p, double x0v, double x0a, int lane) {  // (out of line: the rare path must not cost the kernels registers)
  const int ax = lane / 3, j = lane - 3 * ax;
  const int i = ax < 3 ? ax : 2;  // (lanes beyond 8 repeat axis 2: harmless duplicates)
  const double xfp = i == 0 ? pr.xf[0] : (i == 1 ? pr.xf[1] : pr.xf[2]);
  const double dx = xfp - x0p;
  const float tv = (float)(fabs(dx) / pr.v_max);                     // :672-674
  const float jerk = (float)(copysign(1.0, dx) * pr.j_max);          // :679-681
  const float a0 = (float)x0a, v0 = (float)x0v;                      // :682-687
  const double rj = cubic_root_of_lane((double)jerk / 6.0, (double)a0 / 2.0, (double)v0, -dx, j);  // :691-713
  const float acc = (float)(copysign(1.0, dx) * pr.a_max);           // :718-720
  const double ra = quad_root_of_lane(0.5 * (double)acc, (double)v0, -dx, j);                       // :724-746
  // MinPositiveElement (solverGurobi_utils.hpp:19-32) over the roots of an axis, cast to float as the reference stores them; then
  // the maximum over the nine values
  float mx = 0.f;
#pragma unroll
  for (int aa = 0; aa < 3; aa++) {
    const double bj = wave_min((i == aa && rj > 0) ? rj : INFINITY);
    const double ba = wave_min((i == aa && ra > 0) ? ra : INFINITY);
    const float tj = (float)(bj < INFINITY ? bj : 0.0), ta = (float)(ba < INFINITY ? ba : 0.0);
    const float tva = (float)readlane_f64((double)tv, 3 * aa);
    mx = fmaxf(mx, fmaxf(tva, fmaxf(ta, tj)));
  }
  double dt0 = (double)(mx / (float)pr.n_seg);  // :751  (float / int)
  if (dt0 > 10000) dt0 = 0;                     // :752-756
  return dt0;
}

// ---- the same value with a tenth of the transcendental work ------------------------------------------------------
// What the closed forms are FOR is a starting point of polish3: its three Newton steps square the error of the start three times, so
// a start that is good to 1e-5 ends on the same rounding-level neighbourhood of the root as one that is good to 1e-16 — and the
// result is stored as a float (:662-670).  The starts below are therefore computed with the hardware's single-precision
// transcendentals (v_log / v_exp for the cube roots, a 7th-degree polynomial for acos, v_cos) on quantities brought into their range
// first, the double-precision square root by v_sqrt_f64 without its correction steps.  The depressed cubic (B, C, D, p, q, the discriminant)
// only CHOOSES the case and the starting points here: its divisions are multiplications by a corrected reciprocal (v_rcp_f64 + one Newton
// step, an ulp or two) and its products are re-associated (B * B/3, (p/3)^3, (q/2)^2) — not the oracle's roundings.  That is safe where
// the case does not hang on those last bits, and the guards below send everything else to the exact path: |disc| below 1e-6 of its terms
// (two roots closer than a thousandth of their size), terms so small that the 1e-12 test for an imaginary part decides, a triple root
// with B != 0, a start or goal within 1e-30 on an axis.  What DECIDES a value is the oracle's arithmetic: MinPositiveElement, the float
// casts, and the quadratic's roots, which are used as they come (correctly rounded sqrt and divisions, no polish).  The polished cubic
// root can still differ from the exact path's by an ulp of the double, i.e. the float the reference stores can land on the other side
// of a rounding boundary about once in 1e8 roots: fh_dt_initial_batch against orc_dt_initial_batch on 2.1 M problems (tests/test_gpu_round5.py)
// and the A/B test against the FH_DT_EXACT_ONLY build (tests/test_gpu_round6.py) are statistical evidence, not a proof of bit equality —
// include/fasterhip.h says so.  dx = 0 exactly takes the exact path, where the convention of cubic_root_of_lane applies.
// Lane = 4 * axis + j (an axis' candidates share a quad: their minimum is two quad permutes), nine useful lanes.
__device__ __forceinline__ double cbrt_start(double a) {  // relative error ~1e-6
  const int e = __builtin_amdgcn_frexp_exp(a);              // a = m 2^e, |m| in [0.5, 1)
  const float m = (float)fabs(__builtin_amdgcn_frexp_mant(a));
  const int e3 = (e + 3072) / 3 - 1024, r = e - 3 * e3;     // e = 3 e3 + r, r in {0, 1, 2}
  const float y = __builtin_amdgcn_exp2f((__builtin_amdgcn_logf(m) + (float)r) * (1.0f / 3.0f));
  return copysign(__builtin_ldexp((double)y, e3), a);        // (a = 0: log -> -inf, exp -> 0)
}
__device__ __forceinline__ float acos_start(float x) {  // |error| < 1e-6 on [-1, 1] (Abramowitz & Stegun 4.4.46 in single precision)
  const float ax = fabsf(x);
  float pl = -0.0012624911f;
  pl = pl * ax + 0.0066700901f;
  pl = pl * ax - 0.0170881256f;
  pl = pl * ax + 0.0308918810f;
  pl = pl * ax - 0.0501743046f;
  pl = pl * ax + 0.0889789874f;
  pl = pl * ax - 0.2145988016f;
  pl = pl * ax + 1.5707963050f;
  const float r = __builtin_amdgcn_sqrtf(1.0f - ax) * pl;
  return x < 0.f ? 3.14159265f - r : r;
}
template <class PR, class X0>
__device__ __forceinline__ double dt_initial_exact_lanes(const PR& pr, const X0& x0, int lane) {  // the exact path with its own lane -> axis map
  const int ax = lane / 3;
  const int i = ax < 3 ? ax : 2;
  return dt_initial_exact(pr, x0.pos(i), x0.vel(i), x0.acc(i), lane);
}
template <class PR, class X0>
__device__ __forceinline__ double dt_initial(const PR& pr, const X0& x0, int lane) {
#ifdef FH_DT_EXACT_ONLY
  return dt_initial_exact_lanes(pr, x0, lane);
#else
  const int ax = lane >> 2, j = lane & 3;
  const int i = ax < 3 ? ax : 2;  // (lanes beyond 11 repeat axis 2, lane j = 3 of a quad holds no candidate)
  const double x0p = x0.pos(i), x0v = x0.vel(i), x0a = x0.acc(i);
  const double xfp = i == 0 ? pr.xf[0] : (i == 1 ? pr.xf[1] : pr.xf[2]);
  const double dx = xfp - x0p;
  const float tv = (float)(fabs(dx) / pr.v_max);                     // :672-674
  const float jerk = (float)(copysign(1.0, dx) * pr.j_max);          // :679-681
  const float a0 = (float)x0a, v0 = (float)x0v;                      // :682-687
  // the cubic (jerk / 6) t^3 + (a0 / 2) t^2 + v0 t - dx (:691-713).  Its depressed form only chooses the case and the starting points
  // here, so its divisions are multiplications by a reciprocal that is good to an ulp or two (v_rcp_f64 and one correction step)
  const double c3 = (double)jerk / 6.0, c2 = (double)a0 / 2.0, c1 = (double)v0, c0 = -dx;
  double ic3 = __builtin_amdgcn_rcp(c3);
  ic3 = fma(fma(-c3, ic3, 1.0), ic3, ic3);
  const double B = c2 * ic3, C = c1 * ic3, D = c0 * ic3, B3 = B * (1.0 / 3.0);
  const double p = C - B * B3;
  const double q = 2.0 * B3 * B3 * B3 - B3 * C + D;
  const double q2 = 0.5 * q, p3 = p * (1.0 / 3.0);
  const double p27 = p3 * p3 * p3;
  const double disc = q2 * q2 + p27;
  const double scale = q2 * q2 + fabs(p27);
  // the quadratic (acc / 2) t^2 + v0 t - dx (:724-746): its roots are used as they come, so the square root and the two divisions
  // are the correctly rounded ones.  acc has the sign of dx: its discriminant v0^2 + 2 |acc| |dx| is never negative
  const float acc = (float)(copysign(1.0, dx) * pr.a_max);           // :718-720
  const double qc2 = 0.5 * (double)acc;
  const double qdisc = c1 * c1 - 4.0 * qc2 * c0;
  const bool triple = !(disc > 0) && p == 0;  // (B = C = D = 0 — at rest on this axis with nowhere to go — is the case that occurs)
  // (dx = 0 — same height at start and goal, say — puts a root of the cubic AT zero: the exact path factors it out, cubic_root_of_lane;
  // a |dx| below 1e-30 that is not zero goes there too and takes the general closed form)
  const bool ill = !(fabs(disc) >= 1e-6 * scale) || (scale > 0.0 && scale < 1e-40) || !(scale < 1e280) || !(qdisc >= 0.0) || (triple && B != 0.0) ||
                   !(fabs(dx) >= 1e-30);
  if (wave_any(ill && lane < 12)) return dt_initial_exact_lanes(pr, x0, opaque(lane));
  double rj = 0.0;
  if (disc > 0) {  // one real root (the pair's imaginary part is far above 1e-12 here: j = 1 has no candidate)
    const double sq = __builtin_amdgcn_sqrt(disc);
    const double u = cbrt_start(sq - q2), v = cbrt_start(-q2 - sq);
    rj = u + v - B3;
  } else if (triple) {
    rj = 0.0;  // -B / 3 with B = 0, used as it is
  } else {  // three real roots
    const double m = 2.0 * __builtin_amdgcn_sqrt(-p3);
    const float arg = fminf(1.0f, fmaxf(-1.0f, (float)(3.0 * q * __builtin_amdgcn_rcp(p * m))));
    const float rev = acos_start(arg) * (1.0f / (3.0f * 6.28318531f)) - (float)j * (1.0f / 3.0f);  // (phi - 2 pi j / 3) in revolutions
    rj = m * (double)__builtin_amdgcn_cosf(rev) - B3;
  }
  if (!triple) {  // polish3 with the same kind of reciprocal: a Newton step's quotient is a correction, an ulp of it is far below one of t
#pragma unroll
    for (int it = 0; it < 3; it++) {
      const double f = ((c3 * rj + c2) * rj + c1) * rj + c0;
      const double df = (3 * c3 * rj + 2 * c2) * rj + c1;
      double r = __builtin_amdgcn_rcp(df);
      r = fma(fma(-df, r, 1.0), r, r);
      const double tn = rj - f * r;
      rj = (df == 0 || !isfinite(df) || !isfinite(tn)) ? rj : tn;
    }
  }
  if (((disc > 0 || triple) && j != 0) || j == 3) rj = 0.0;  // one real root: only j = 0 holds a candidate
  double ra = 0.0;
  {
    const double sr = sqrt(qdisc);
    const double qq = -0.5 * (c1 + (c1 >= 0 ? sr : -sr));
    if (qq != 0) ra = j == 0 ? qq / qc2 : (j == 1 ? c0 / qq : 0.0);
  }
  // MinPositiveElement (solverGurobi_utils.hpp:19-32) over the candidates of an axis (a quad), cast to float as the reference stores
  // them; then the maximum over the nine values
  double mj = rj > 0 ? rj : INFINITY, ma = ra > 0 ? ra : INFINITY;
  mj = vmin_f64(mj, dpp_f64<0xB1, 0xf>(INFINITY, mj));  // quad_perm [1, 0, 3, 2]
  ma = vmin_f64(ma, dpp_f64<0xB1, 0xf>(INFINITY, ma));
  mj = vmin_f64(mj, dpp_f64<0x4E, 0xf>(INFINITY, mj));  // quad_perm [2, 3, 0, 1]
  ma = vmin_f64(ma, dpp_f64<0x4E, 0xf>(INFINITY, ma));
  const float tj = (float)(mj < INFINITY ? mj : 0.0), ta = (float)(ma < INFINITY ? ma : 0.0);
  const float ml = fmaxf(tv, fmaxf(ta, tj));
  const int mli = __float_as_int(ml);
  const float mx = fmaxf(fmaxf(0.f, __int_as_float(__builtin_amdgcn_readlane(mli, 0))),
                         fmaxf(__int_as_float(__builtin_amdgcn_readlane(mli, 4)), __int_as_float(__builtin_amdgcn_readlane(mli, 8))));
  double dt0 = (double)(mx / (float)pr.n_seg);  // :751  (float / int)
  if (dt0 > 10000) dt0 = 0;                     // :752-756
  return dt0;
#endif
}

// The safe problem of a fused pair as a record and rows in memory (what the staged hand-off writes: glue_write_safe), from what the
// wavefront that solves it holds: x0 in its LDS slots, the polytope of the whole corridor its corridor starts at, the polytope count.
// Out of line on purpose (its callers are the hand-off with fh_sched.pair_outputs and the first donation of a safe problem: rare, and
// they must not cost the solve registers); explicit arguments, no `this`.  Write-through stores; the caller drains them.
typedef const __attribute__((address_space(3))) double lds_cdouble;
__device__ __attribute__((noinline)) void write_safe_problem(const fh_problem* whole, const fh_face* wfaces, fh_problem* ps, fh_face* sfaces, lds_cdouble* x0P,
                                                             int x0_stride, int start, int cnt, double shrink, double r_margin, int lane) {
  lds_cdouble* x0V = x0P + x0_stride;
  lds_cdouble* x0A = x0V + x0_stride;
  const double x0[9] = {x0P[0], x0P[1], x0P[2], x0V[0], x0V[1], x0V[2], x0A[0], x0A[1], x0A[2]};
  const int fb = __builtin_amdgcn_readfirstlane(whole->face_begin);
  glue_write_safe<true>(wfaces, fb, whole->face_off, start, cnt, x0, shrink, r_margin, *ps, sfaces, lane, nullptr);
}

// -----------------------------------------------------------------------------------------------------------------
// NORMS_TABLE: the per-lane inverse row norms are re-read from the basis table by every scan (the build for three wavefronts per SIMD:
// eight registers that are not spilled) instead of being kept in registers from setup_trial on (the build for two: nothing spills there,
// and the extra loads cost the N = 15 kernel 2.4 %).  The same doubles either way.
template <int NSEG, bool NORMS_TABLE = false>
struct Solver {
  static constexpr int NX = 3 * NSEG;        // jerks of the trajectory (x space)
  static constexpr int NXP = (NX + 7) & ~7;
  static constexpr int KMAX = NSEG - 2;      // free dimensions per axis once the final-state equalities are eliminated (safe: N - 2)
  static constexpr int NV = 3 * KMAX;        // unknowns y of a QP (reduced space)
  static constexpr int NVP = (NV + 7) & ~7;  // vectors / factor rows padded with zeros to a multiple of 8
  static constexpr int S = NVP + 1;          // odd row stride: conflict-free column AND row sweeps with ds_read_b64
  static constexpr int NT = NSEG + 1;
  static constexpr int ZS = NSEG | 1;        // row stride of the basis Zm[s][l] (odd: lanes of different segments hit different banks)
  static_assert(NSEG >= 3, "reduced space needs N - 2 > 0 columns");
  static constexpr int RPSZ = NVP * (NVP + 1) / 2;  // R is packed upper triangular, column major: R(r, c) = R[c(c+1)/2 + r]
  static __device__ __forceinline__ int rp(int r_, int c_) { return (c_ * (c_ + 1)) / 2 + r_; }

  // ---- LDS carve (doubles first) ----
  double *Q, *R;                                      // Q1 column major [NVP cols][S] (column c = active slot); R packed upper triangular [RPSZ]
  double *x, *z, *g, *d, *r, *u, *rinv;               // [NVP] x = the reduced unknowns y (r aliases d: only live inside the re-orthogonalisation pass)
  double *tcache, *tbnd;                              // [TC][SNAP_TAIL] snapshot tails and [TC][FH_MAX_POLY] child bounds of the first TC tree levels
  static constexpr int TC = NSEG <= 6 ? 2 : (NSEG <= 10 ? FH_TAIL_LEVELS : FH_TAIL_LEVELS_BIG);
  double* xs;                                         // [NXP] x-space scratch: Z y (compute_states), a row normal in x space (build_g)
  double* Zm;                                         // [NSEG][ZS] orthogonal basis of this N (fh_basis.hip.hpp), kept across problems
  double *Pc, *Vc, *Ac;                               // [NT*3] current states at segment starts
  double* viol;                                       // [NSEG][FH_MAX_POLY]
  double* xfl;                                        // [9] goal state (+3 pad)
  float* tolf;                                        // [n_faces] feas_tol / |a_f| (added to the scaled violation of the row that is selected:
                                                      //           ~1e-9, so a float's relative 6e-8 is 1e-16 absolute)
  fh_face* faces;                                     // [n_faces] NORMALISED rows: a/|a| and bt = -(b + feas_tol)/|a|, so that
                                                      //           a.cp + bt > 0  <=>  the original row is violated by more than feas_tol
  int *act, *assign, *bestassign, *fullassign, *stk_seg, *stk_next, *stk_cnt, *stk_q, *stk_mask, *stk_keep, *face_off;
  int* tb;                                            // [TB_WORDS] wave-uniform words that would otherwise sit in SGPRs for the whole solve
  enum { TB_B = 0, TB_PHASE = 1, TB_F = 2, TB_TRIALS = 4, TB_BASE = 5, TB_H = 7, TB_REC = 9, TB_DEPTH0 = 10, TB_KEY = 11, TB_QE = 13,
         TB_T0 = 14, TB_WORK = 16, TB_ZN = 17, TB_NEXT = 18, TB_NEXT_EI = 20, TB_NEXT_WT = 22, TB_DONE_N = 24, TB_DONE_IT = 25, TB_SAFE = 26, TB_POOL2 = 27, TB_FRESH = 29, TB_WORDS = 30 };
                                                                 // TB_FRESH: the states and jerks in LDS / registers are those of the incumbent leaf (no node was solved since it was found)
                                                                 // TB_POOL2 / + 1: a second range of tickets [next, end): tickets another workgroup gave away (give_tickets) while this one
                                                                 // still had some of its own; it becomes the pool when the pool is empty
                                                                 // TB_SAFE: the safe problem of the fused pair in hand — bits 0..7 the polytope of the whole corridor its corridor starts
                                                                 // at, bit 8: its record and rows are in memory (written back: always, or when the problem was first shared)
                                                                 // TB_NEXT / TB_NEXT + 1: this workgroup's own pool of tickets [next, end), drawn a chunk at a time (and ahead,
                                                                 // during a hand-off, with the control words TB_NEXT_EI / TB_NEXT_WT read at the same time); TB_DONE_LOCAL lives in tb[TB_ZN + ...]  // TB_WORK: active-set iterations of the unit in hand (reported with `done`);
                                                                 // TB_ZN: the N whose basis is in Zm (0: none yet)
  signed char* stk_order;                             // [NSEG][FH_MAX_POLY] child order per tree level

  static __host__ __device__ constexpr size_t lds_bytes(int max_faces) {
    // (the hardware hands out LDS in granules of 1280 B — measured with a residency census: 14 080 B admit 11 workgroups per CU,
    // 14 336 B only 10 — so every few hundred bytes of this carve decide a wavefront per CU)
    return sizeof(double) * (NVP * S + RPSZ + 3 * NVP + NVP / 2 + 2 * NVP + NXP + NSEG * ZS + 3 * NT * 3 + 12 + TC * (2 * NVP + NVP / 2 + FH_MAX_POLY)) +
           sizeof(int) * (9 * NSEG + FH_MAX_POLY + 1 + TB_WORDS) + ((NSEG * FH_MAX_POLY + 15) & ~15) +
           (sizeof(fh_face) + sizeof(float)) * max_faces + 16;
  }

  // ---- per-lane state kept in registers (LDS is what limits the number of resident solves) ----
  double p0r, v0r, a0r;   // lane = (tt, i) < 3 NT: state at the start of segment tt for y = 0: zero-jerk propagation of x0 plus the
                          //   contribution of xp (per trial)
  double xpr;             // lane = (s, i) < 3 N: jerk xp of the minimum-norm solution of the final-state equalities (per trial)
  double xj;              // lane = (s, i) < 3 N: current jerk xp + (Z y) (compute_states -> scan)
  // [r5] NORMS_TABLE: the inverse row norms of a lane's rows are re-read from the basis table (L1 / L2 resident) by every scan instead of
  // living in 8 registers across the whole search: same values, same throughput in the three-wavefront build (A/B: 19.96 / 20.00 M
  // pairs/s), but 112 instead of 160 B of scratch and 0.58 instead of 0.80 GB of HBM traffic per launch — the registers freed are
  // registers not spilled at the scope of a problem.
  const double* btab;     // the basis table of this N (uniform)
  double ih_trial;        // 1 / h of the trial (uniform)
  // the inverse row norms of this lane's rows (reduced space) from the table, as setup_trial computes them
  __device__ __forceinline__ void row_norms(double& wbj_, double& wbv_, double& wba_, double& wcp_) const {
#pragma clang fp contract(off)
    const int lane = opaque(this->lane);
    const double* C = btab + BT_C + (force_final ? BT_C_WORDS : 0);
    const double* CJ = btab + BT_CJ + (force_final ? BT_CJ_WORDS : 0);
    const int t = lane < nx ? lane / 3 : 0;
    const int tc = lane < 4 * N ? (lane >> 2) : 0, k = lane & 3;
    const double cj_ = CJ[t], cv_ = C[W_V * BT_C_TT + t], ca_ = C[W_A * BT_C_TT + t], cc_ = C[(k == 3 ? W_P : k) * BT_C_TT + tc + (k == 3 ? 1 : 0)];
    const double ih = ih_trial;
    const bool box = lane < nx;
    wbj_ = box ? cj_ : 0.0;
    wbv_ = box ? cv_ * (ih * ih) : 0.0;
    wba_ = box ? ca_ * ih : 0.0;
    wcp_ = (lane < 4 * N) ? cc_ * (ih * ih * ih) : 0.0;
  }
  double wbj, wbv, wba, wcp;  // inverse row norms IN THE REDUCED SPACE of this lane's box rows (lane = (t, i)) and corridor rows
                          //   (lane = (t, k)); 0: the row does not depend on y (per trial)
  double* bestx_g;        // [NVP] incumbent y (lane < n) in the workgroup's workspace: written when a better leaf is found, read when the problem ends or is first shared
  double cp_r[3];         // lane = (t, k): control point k of segment t at the current x (compute_states -> scan)
  int scan_f0, scan_F;    // lane = (t, k): first row and row count of the polytope segment t is assigned to (0 rows: free) (per node)
  // ---- wave-uniform scalars ----
  int lane, N, n, q, P;   // n = 3 K: unknowns of the reduced QP
  int nx;                 // 3 N: jerks of the trajectory
  int K, zc0;             // Z = columns zc0 .. zc0 + K - 1 of Zm (zc0 = 2 safe, 3 whole; K = max(N - zc0, 0))
  int qe;                 // equality rows in the factorisation: always 0 (they are eliminated; kept in the frame header layout)
  bool eq_ok;             // the final-state equalities of this trial are consistent (always, unless N < 3)
  double c0;              // |xp|^2: cost of the trial at y = 0
  double box_ub;          // 3N j_max^2 (1 + 1e-9): no feasible trajectory costs more
  double gx2;             // |normal in jerk space|^2 of the row build_g built last
  int maxF;  // max faces of one polytope of this problem (wave-uniform trip count of the face sweeps)
  int scan_trip;  // max faces of a polytope that a segment is ASSIGNED to in this active-set run (bind_assignment): the scan's trip count
  unsigned poly_ok;  // polytopes without a violated zero-normal face (such a polytope can never hold a segment)
#ifdef FH_PROFILE
  unsigned long long prof[24];   // 0-15: see scripts/phase_profile.py; 16 look-around + donations, 17 result write, 18 hand-off of the pair (charged to
  unsigned int cnt[24];          // its safe problem), 19 ticket + launch order fetch (charged to the problem drawn), 20 child order + bounds, 21 leaf bookkeeping
  unsigned long long take_cycles; unsigned int take_calls;  // take_task of a workgroup that still has tickets (is a frame pending?)
  unsigned long long glue_parts[4];  // of the hand-off: before the clock, the clock loop, R + the polytope test (first memory wait), the face copy
  unsigned long long pre_cycles, glue_cycles, drain_cycles;  // measured in the kernel loop, charged to the next problem
  unsigned long long prof2[12];  // [r6] 0 search prologue, 1 backtrack blocks, 2 limits / look / incumbent poll before a node, 3 qp_run, 4 record + LDS init of
                                 // the staging, 5 basis reload, 6 trial loop outside set-up and search, 7 between the trial loop and the result write,
                                 // 8-10 of the ticket phase: up to take_task, the draw, the order fetch; 11 unit_done
  unsigned long long pre_parts[3];
#endif
  unsigned allowed_first, allowed_last;  // polytopes not excluded for segment 0 / N-1 by jerk-independent rows
  double h, tol, dep2;
  double vmax, amax, jmax;
  int force_final;

  // The dual active-set state that a branch-and-bound node inherits is one contiguous LDS block, so that it can be
  // snapshotted to / restored from the per-workgroup HBM workspace with linear 16-B-per-lane copies.

  // Fixed-size arrays first (compile-time LDS offsets that fold into the ds_read/ds_write immediates and cost no
  // SGPRs), the two arrays sized by the batch's face bound last.
  __device__ void carve(unsigned char* base, int max_faces) {
    double* p = reinterpret_cast<double*>(base);
    Q = p; p += NVP * S;
    R = p; p += RPSZ;
    x = p; p += NVP;  u = p; p += NVP;
    act = reinterpret_cast<int*>(p); p += NVP / 2;
    // ---- end of the snapshot block (1 / diag(R) is not part of it: recomputed from R when a snapshot is restored) ----
    rinv = p; p += NVP;
    g = p; p += NVP;  d = p; p += NVP;  r = d;
    xs = p; p += NXP;
    z = xs;  // the remainder kept for the re-orthogonalisation pass lives where the x-space scratch does: xs is dead between the
             // reduction of a row normal (build_g) and the next compute_states
    static_assert(NXP >= NVP, "z aliases xs");
    Zm = p; p += NSEG * ZS;
    Pc = p; p += NT * 3;  Vc = p; p += NT * 3;  Ac = p; p += NT * 3;
    viol = g;  // [NSEG][FH_MAX_POLY] aliases g,d,xs: only live between two active-set runs (analyze)
    static_assert(NSEG * FH_MAX_POLY <= 2 * NVP + NXP && 9 <= NVP, "scratch must fit in g,d,xs");
    xfl = p; p += 12;
    tcache = p; p += TC * SNAP_TAIL;
    tbnd = p; p += TC * FH_MAX_POLY;
    int* ip = reinterpret_cast<int*>(p);
    assign = ip; ip += NSEG;  bestassign = ip; ip += NSEG;  fullassign = ip; ip += NSEG;
    stk_seg = ip; ip += NSEG;  stk_next = ip; ip += NSEG;  stk_cnt = ip; ip += NSEG;  stk_q = ip; ip += NSEG;  stk_mask = ip; ip += NSEG;  stk_keep = ip; ip += NSEG;
    face_off = ip; ip += FH_MAX_POLY + 1;
    tb = ip; ip += TB_WORDS;
    stk_order = reinterpret_cast<signed char*>(ip);
    const size_t off = (size_t)(reinterpret_cast<unsigned char*>(ip) - base) + NSEG * FH_MAX_POLY;
    faces = reinterpret_cast<fh_face*>(base + ((off + 15) & ~(size_t)15));  // [max_faces] 32-B rows, read 16 B at a time
    tolf = reinterpret_cast<float*>(faces + max_faces);
  }

  // ---- node state snapshots (HBM workspace, one slot per tree level; L2-resident in practice) ----
  // A snapshot holds only what is live: the first q columns of Q1 (contiguous: column major), the first q columns of the
  // packed R, and the fixed tail (x, u, 1/diag, active masks).  Workspace slot: [tail | Q | R], each padded for the
  // 16-B-per-lane copy granularity.
  static constexpr int SNAP_TAIL = 2 * NVP + NVP / 2;  // x, u, active row ids
  static constexpr int SNAP_QOFF = (SNAP_TAIL + 127) & ~127;
  // the lower bounds of a frame's children (child bound, search()), by rank, live in the padding behind the tail of the frame's
  // workspace slot: they travel with a frame that is given away
  static constexpr int SNAP_BOUNDS = SNAP_TAIL, SNAP_TAIL_COPY = SNAP_TAIL + FH_MAX_POLY;
  static_assert(SNAP_TAIL + FH_MAX_POLY <= ((SNAP_TAIL + 127) & ~127), "the child bounds live in the padding behind the tail");
  static constexpr int SNAP_ROFF = SNAP_QOFF + ((NVP * S + 127) & ~127) + 128;
  static constexpr int SNAP_PADDED = SNAP_ROFF + ((RPSZ + 127) & ~127) + 128;  // doubles per workspace slot
  __device__ __forceinline__ void copy_out(double* __restrict__ dst, const double* src, int count) const {
    const int n2 = (count + 1) >> 1;
    const double2* s2 = reinterpret_cast<const double2*>(src);
    double2* d2 = reinterpret_cast<double2*>(dst);
    for (int i = lane; i < n2; i += 64) d2[i] = s2[i];
  }
  __device__ __forceinline__ void copy_in(double* dst, const double* __restrict__ src, int count) const {
    const int n2 = (count + 1) >> 1;
    double2* d2 = reinterpret_cast<double2*>(dst);
    const double2* s2 = reinterpret_cast<const double2*>(src);
    for (int i0 = 0; i0 < n2; i0 += 256) {  // four 1-KiB loads in flight before the first LDS store
      double2 t[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int i = i0 + j * 64 + lane;
        t[j] = s2[i < n2 ? i : 0];
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int i = i0 + j * 64 + lane;
        if (i < n2) d2[i] = t[j];  // same lane <-> same words as the save
      }
    }
  }
  // stk_keep[level]: the columns of Q and R below this index have not changed in LDS since the snapshot of the level was taken (rows are
  // appended on the right; only dropping row k rewrites the columns from k on, drop_row), so a restore fetches the others only
  __device__ void snapshot_save(double* __restrict__ ws_level, int level) {
    const int lane = opaque(this->lane);  // (lane-derived constants are recomputed here, not kept in registers across the whole solve)
    FH_SYNC();
    if (level < TC) {  // the tail stays in LDS
      double* tc = tcache + level * SNAP_TAIL;
      for (int i = lane; i < SNAP_TAIL; i += 64) tc[i] = x[i];
    } else {
      copy_out(ws_level, x, SNAP_TAIL);
    }
    copy_out(ws_level + SNAP_QOFF, Q, q * S);
    copy_out(ws_level + SNAP_ROFF, R, (q * (q + 1)) / 2);
    if (lane == 0) stk_keep[level] = q;
  }
  // q_saved: number of active rows in the snapshot; the current q may be larger (columns to clear) or smaller
  __device__ void snapshot_restore(const double* __restrict__ ws_level, int q_saved, int level) {
    const int lane = opaque(this->lane);  // (lane-derived constants are recomputed here, not kept in registers across the whole solve)
    const int keep0 = uniform_i32(stk_keep[level]);
    const bool intact = keep0 >= q_saved;  // no row below q_saved was dropped since the snapshot: its factor columns are still in LDS
    const bool cached = level < TC;
    int keep = keep0 < 0 ? 0 : (keep0 > q_saved ? q_saved : keep0);
    keep &= ~1;  // (the copies move 16 bytes per lane: an even number of doubles from an even offset; S is odd)
    // the workspace was written by this same wavefront (snapshot_save of an ancestor node): drain its outstanding stores
    // before reading them back — only when something IS read back (a drain is a full memory round trip)
    if (!cached || !intact) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if (lane < NVP)
      for (int c = q_saved; c < q; c++) Q[c * S + lane] = 0.0;  // keep the zero padding beyond the active columns
    FH_SYNC();
    constexpr int TAIL2 = SNAP_TAIL / 2, TAIL_TRIPS = (TAIL2 + 63) / 64;
    static_assert(SNAP_TAIL % 2 == 0, "the tail is copied 16 B at a time");
    if (!intact) {
      // the tail and the first 2 KiB of R are requested before the Q copy starts, so that the three round trips overlap
      const int r0 = ((keep * (keep + 1)) / 2) & ~1;  // first double of R that is fetched (columns >= keep)
      const int nr = (q_saved * (q_saved + 1)) / 2 - r0, nr2 = (nr + 1) >> 1;
      const double2* st2 = reinterpret_cast<const double2*>(ws_level);
      const double2* sr2 = reinterpret_cast<const double2*>(ws_level + SNAP_ROFF + r0);
      double2 tail[TAIL_TRIPS];
      if (!cached) {
#pragma unroll
        for (int j = 0; j < TAIL_TRIPS; j++) tail[j] = st2[(j * 64 + lane) < TAIL2 ? j * 64 + lane : 0];
      }
      double2 rr[2];
#pragma unroll
      for (int j = 0; j < 2; j++) rr[j] = sr2[(j * 64 + lane) < nr2 ? j * 64 + lane : 0];
      copy_in(Q + keep * S, ws_level + SNAP_QOFF + keep * S, (q_saved - keep) * S);
      if (!cached) {
#pragma unroll
        for (int j = 0; j < TAIL_TRIPS; j++)
          if (j * 64 + lane < TAIL2) reinterpret_cast<double2*>(x)[j * 64 + lane] = tail[j];
      }
#pragma unroll
      for (int j = 0; j < 2; j++)
        if (j * 64 + lane < nr2) reinterpret_cast<double2*>(R + r0)[j * 64 + lane] = rr[j];
      if (nr2 > 128) copy_in(R + r0 + 256, ws_level + SNAP_ROFF + r0 + 256, nr - 256);  // more than 22 active rows to fetch
    } else if (!cached) {
      copy_in(x, ws_level, SNAP_TAIL);
    }
    if (cached) {
      const double* tc = tcache + level * SNAP_TAIL;
      for (int i = lane; i < SNAP_TAIL; i += 64) x[i] = tc[i];
    }
    q = q_saved;
    if (lane == 0) stk_keep[level] = q_saved;
    FH_SYNC();
    if (lane < q_saved) rinv[lane] = 1.0 / R[rp(lane, lane)];  // (the same division that produced the value when the row was added)
    FH_SYNC();
  }

  // Invariant kept by reset_qp/add_row/drop_row: Q[i][c] == 0 for i >= n or c >= q, and g, d, z, x are zero
  // beyond n (resp. q), so that the factor sweeps below can run in unpredicated blocks of 8.
  __device__ void init_problem() {
    const int lane = opaque(this->lane);  // (lane-derived constants are recomputed here, not kept in registers across the whole solve)
    for (int i = lane; i < NVP * S; i += 64) Q[i] = 0.0;
    for (int i = lane; i < RPSZ; i += 64) R[i] = 0.0;
    if (lane < NVP) { x[lane] = 0; g[lane] = 0; d[lane] = 0; u[lane] = 0; rinv[lane] = 0; act[lane] = 0; }
    if (lane < NXP) xs[lane] = 0;
    q = 0;
    FH_SYNC();
  }

  // ---- x-space jerks in xs[] -> their contribution to the state at the start of segment tt = lane / 3 (lane < 3 NT) ----
  // sum_{s<tt} c(m) x_s with m = tt-1-s and c polynomial in m: three moments S_k = sum m^k x_s carry all three states
  // (cP = h^3 (1/6 + m/2 + m^2/2), cV = h^2 (1/2 + m), cA = h); nothing per-(lane, s) to keep in registers.
  __device__ __forceinline__ void moments(double& dp, double& dv, double& da) const {
#pragma clang fp contract(off)  // (inlined at several sites: the same roundings at each of them)
    const int lane = opaque(this->lane);  // (lane-derived constants are recomputed here, not kept in registers across the whole solve)
    const int tt = lane / 3, i = lane - 3 * tt;
    const int ii = lane < 3 * NT ? i : 0;
    double xr[NSEG];
#pragma unroll
    for (int s = 0; s < NSEG; s++) xr[s] = xs[3 * s + ii];  // unpredicated (xs is zero beyond 3 N)
    in_flight(xr);
    double s0 = 0, s1 = 0, s2 = 0;
    const double dm0 = opaque((double)(tt - 1));
#pragma unroll
    for (int s = 0; s < NSEG; s++) {
      const double dm = dm0 - (double)s;
      const double xv = xr[s] * fmin(fmax(dm + 1.0, 0.0), 1.0);  // segments s >= tt contribute nothing
      const double t1 = dm * xv;
      s0 += xv;
      s1 += t1;
      s2 = fma(dm, t1, s2);
    }
    const double h2 = h * h, h3 = h2 * h;
    dp = h3 * (s0 * (1.0 / 6.0) + 0.5 * s1 + 0.5 * s2);
    dv = h2 * (0.5 * s0 + s1);
    da = h * s0;
  }

  // ---- per trial (step h): states for y = 0 and the inverse row norms.  bt: the basis table of this N (fh_basis.hip.hpp) ----
  // Returns true when the trial is refuted at y = 0 by the jerk box (only if may_end_early): nothing else was set up then.
  template <class PR>
  __device__ __forceinline__ bool setup_trial(const PR& pr, const double* __restrict__ bt, bool may_end_early = false) {
    // Inlined twice — the trial loop, and the worker that writes the result of a shared problem — and results must not depend on
    // which copy ran: no contraction into fused multiply-adds left to the optimiser's choice per site.
#pragma clang fp contract(off)
    const int lane = opaque(this->lane);  // (lane-derived constants are recomputed here, not kept in registers across the whole solve)
    // the inverse row norms of this lane's rows at h = 1 (reduced space): requested first, used last
    double cj_, cv_, ca_, cc_;
    {
      const double* C = bt + BT_C + (force_final ? BT_C_WORDS : 0);
      const double* CJ = bt + BT_CJ + (force_final ? BT_CJ_WORDS : 0);
      const int t = lane < nx ? lane / 3 : 0;
      const int tc = lane < 4 * N ? (lane >> 2) : 0, k = lane & 3;
      cj_ = CJ[t];
      cv_ = C[W_V * BT_C_TT + t];
      ca_ = C[W_A * BT_C_TT + t];
      cc_ = C[(k == 3 ? W_P : k) * BT_C_TT + tc + (k == 3 ? 1 : 0)];
    }
    double p, v, a;
    {  // zero-jerk propagation of x0 to the start of segment tt = lane / 3
      const int tt = lane / 3, i = lane - 3 * tt;
      const bool on = lane < 3 * NT;
      // x0 lives in LDS: entries 0..2 of Pc / Vc / Ac (the state at the start of segment 0 IS x0; compute_states leaves them alone)
      const double p0 = on ? Pc[i] : 0.0;
      const double v0 = on ? Vc[i] : 0.0;
      a = on ? Ac[i] : 0.0;
      const double T = (double)tt * h;
      p = p0 + T * (v0 + 0.5 * T * a);
      v = v0 + T * a;
    }
    // final-state equalities (setConstraintsXf :332-357): xp = Q[:, 0:3] (M rho), rho_j = (target_j - zero-jerk end state_j) / h^p_j
    // for the rows in the reference's order ([pos], vel, accel).  One lane per axis; M, G and the mask of dependent rows (N < 3:
    // a dependent row is skipped if consistent, else the trial is infeasible) come from the table.
    typedef const __attribute__((address_space(4))) double cdouble;
    const cdouble* E = (const cdouble*)(unsigned long long)(bt + BT_EQ + (force_final ? BT_EQ_WORDS : 0));
    bool bad = false;
    double l1 = 0.0;  // |xp|_1 (the early exit's second certificate)
    {
      double endP = 0, endV = 0, endA = 0;
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const double ep = readlane_f64(p, 3 * N + i), ev = readlane_f64(v, 3 * N + i), ea = readlane_f64(a, 3 * N + i);
        if (lane == i) { endP = ep; endV = ev; endA = ea; }
      }
      const int nrow = force_final ? 3 : 2, koff = 3 - nrow;
      const double ih = 1.0 / h;
      const double ihp[3] = {ih * ih * ih, ih * ih, ih};  // by kind: pos, vel, accel
      const double hp[3] = {h * h * h, h * h, h};
      const int axis = lane < 3 ? lane : 0;
      double rho[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int kind = (j + koff) < 3 ? j + koff : 2;
        const double base = kind == 0 ? endP : (kind == 1 ? endV : endA);
        rho[j] = (j < nrow) ? (xfl[kind * 3 + axis] - base) * (kind == 0 ? ihp[0] : (kind == 1 ? ihp[1] : ihp[2])) : 0.0;
      }
      const int mask = (int)E[18];
#pragma unroll
      for (int l = 0; l < 3; l++) {
        const double cl = E[l * 3 + 0] * rho[0] + E[l * 3 + 1] * rho[1] + E[l * 3 + 2] * rho[2];
        if (lane < 3) d[l * 3 + lane] = cl;  // (scratch: d is rewritten before it is read by the solver)
        if ((mask >> l) & 1) {
          const int kind = (l + koff) < 3 ? l + koff : 2;
          const double res = (kind == 0 ? hp[0] : (kind == 1 ? hp[1] : hp[2])) * (E[9 + l * 3 + 0] * rho[0] + E[9 + l * 3 + 1] * rho[1] + E[9 + l * 3 + 2] * rho[2]);
          if (lane < 3 && fabs(res) > tol) bad = true;
        }
      }
    }
    eq_ok = !wave_any(bad);
    FH_SYNC();
    {
      const int s = lane / 3, i = lane - 3 * s;
      double xp = 0.0;
      if (lane < nx) {
        const int nl = N < 3 ? N : 3;
        for (int l = 0; l < nl; l++) xp += Zm[s * ZS + l] * d[l * 3 + i];
      }
      xpr = xp;
      if (lane < NXP) xs[lane] = xp;
      c0 = wave_sum(xp * xp);
#ifndef FH_NO_L1_BOUND
      l1 = may_end_early ? wave_sum(fabs(xp)) : 0.0;
#endif
    }
    FH_SYNC();
#ifndef FH_NO_EARLY_OUT
    // The cheapest trajectory that meets the final-state equalities already costs more than any trajectory inside the jerk box may
    // (|x|^2 <= 3N j_max^2, setMaxConstraints :403-405): the root of this trial is infeasible before its first active-set iteration —
    // what qp_loop finds at y = 0 with the same comparison.  Most failed trials of a whole problem (94 % on C4) and a third of a safe
    // problem's end here: the states, the row norms, the screening and the root's set-up are skipped (run_problem counts the node).
    // [r5] ... and a sharper certificate from the same vector: xp is the minimum-norm solution, xp = W^T lambda for the equality rows W x = r,
    // so every x that meets the equalities has xp . x = lambda . r = |xp|^2; inside the jerk box (|x_i| <= j_max + tol) that product is at
    // most |xp|_1 (j_max + tol).  |xp|^2 > |xp|_1 (j_max + tol) therefore refutes the trial too (by Cauchy-Schwarz never later than
    // the 2-norm test): on C4 it ends 41 % instead of 34 % of a safe problem's failed trials at y = 0 (whole: 97 % instead of 94 %), and
    // fired on none of the 10 655 feasible trials of the sample.
#ifndef FH_EARLY_IN_SEARCH
#ifndef FH_NO_L1_BOUND
    if (may_end_early && eq_ok && (!(c0 <= box_ub) || c0 > l1 * (jmax + tol) * (1.0 + 1e-9))) return true;
#else
    if (may_end_early && eq_ok && !(c0 <= box_ub)) return true;
#endif
#endif
#endif
    {
      double dp, dv, da;
      moments(dp, dv, da);
      p0r = p + dp; v0r = v + dv; a0r = a + da;
    }
    {  // each lane keeps the inverse norms (reduced space) of the rows it scans; they scale with h^-3 (positions), h^-2, h^-1
      const double ih = 1.0 / h;
      if constexpr (NORMS_TABLE) {
        btab = bt;
        ih_trial = ih;
      } else {
        const bool box = lane < nx;
        wbj = box ? cj_ : 0.0;
        wbv = box ? cv_ * (ih * ih) : 0.0;
        wba = box ? ca_ * ih : 0.0;
        wcp = (lane < 4 * N) ? cc_ * (ih * ih * ih) : 0.0;
      }
    }
    FH_SYNC();  // (xs and z are rewritten in full before they are read again)
    return false;
  }

  // Bezier control point k of a segment from the state (P, V, A) at its start and the position Pn at its end (getCP0..3, :833-862)
  __device__ __forceinline__ double cp_of(int k, double P_, double V_, double A_, double Pn) const {
    const double wv = k == 1 ? h / 3.0 : (k == 2 ? 2.0 * h / 3.0 : 0.0);
    const double wa = k == 2 ? h * h / 6.0 : 0.0;
    return k == 3 ? Pn : P_ + wv * V_ + wa * A_;
  }

  // ---- jerks x = xp + Z y, states at segment starts and Bezier control points of the current y ----
  __device__ void compute_states() {
    const int lane = opaque(this->lane);  // (lane-derived constants are recomputed here, not kept in registers across the whole solve)
    {  // lane = (s, i): (Z y)_(s,i) = sum_k Zm[s][zc0 + k] y[3 k + i]
      const int s = lane / 3, i = lane - 3 * s;
      const bool on = lane < nx;
      const double* zr = Zm + (on ? s : 0) * ZS + zc0;
      const double* yr = x + (on ? i : 0);
      double a0 = 0, a1 = 0;
      const int kl = K > 0 ? K - 1 : 0;
      for (int k0 = 0; k0 < K; k0 += 4) {
        double zv[4], yv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int k = min(k0 + j, kl);
          zv[j] = zr[k];
          yv[j] = yr[3 * k];
        }
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          a0 += (k0 + j < K) ? zv[j] * yv[j] : 0.0;
          a1 += (k0 + j + 1 < K) ? zv[j + 1] * yv[j + 1] : 0.0;
        }
      }
      const double xz = on ? a0 + a1 : 0.0;
      xj = xpr + xz;
      if (lane < NXP) xs[lane] = xz;
    }
    FH_SYNC();
    if (lane < (N + 1) * 3) {
      double dp, dv, da;
      moments(dp, dv, da);
      // (lanes 0..2 — the start of segment 0 — would store x0 + 0: the slots hold x0 itself, staged once per problem, and everything
      // that needs x0 reads it there: setup_trial, dt_initial, the screening, the hand-off of a fused pair)
      if (lane >= 3) { Pc[lane] = p0r + dp; Vc[lane] = v0r + dv; Ac[lane] = a0r + da; }
    }
    FH_SYNC();
    if (lane < 4 * N) {  // lane = (segment, control point): three axes each, no integer divisions
      const int t = lane >> 2, k = lane & 3;
      const int o = t * 3;
      double st[12];
#pragma unroll
      for (int i = 0; i < 3; i++) { st[i] = Pc[o + i]; st[3 + i] = Vc[o + i]; st[6 + i] = Ac[o + i]; st[9 + i] = Pc[o + 3 + i]; }
      in_flight(st);
#pragma unroll
      for (int i = 0; i < 3; i++) cp_r[i] = cp_of(k, st[i], st[3 + i], st[6 + i], st[9 + i]);
    }
  }

  // assign[] only changes between two active-set runs: the scan lanes look their rows up once per run
  __device__ void bind_assignment() {
    const int lane = opaque(this->lane);  // (lane-derived constants are recomputed here, not kept in registers across the whole solve)
    const bool live = lane < 4 * N;
    const int p = live ? assign[lane >> 2] : -1;
    scan_f0 = p >= 0 ? face_off[p] : 0;
    scan_F = p >= 0 ? face_off[p + 1] - scan_f0 : 0;
    {  // flop accounting: rows x control points scanned per iteration of this run (integer sum over the lanes)
      int v = scan_F;
      v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
      v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
      v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
      v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
      v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xf, 0xf, true);
      v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xf, 0xf, true);
      rows4 = __builtin_amdgcn_readlane(v, 63);
    }
#ifdef FH_SCAN_TRIP  // the trip count of this run's face sweeps: the longest row list any lane scans (0 at a root with no segment assigned).
                     // Measured (A/B, C4): 19.4 M pairs/s with and without it — the sweeps are not what a node waits for; off by default
    {
      int m = scan_F;
      m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x111, 0xf, 0xf, true));
      m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x112, 0xf, 0xf, true));
      m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x114, 0xf, 0xf, true));
      m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x118, 0xf, 0xf, true));
      m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x142, 0xf, 0xf, true));
      m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x143, 0xf, 0xf, true));
      scan_trip = __builtin_amdgcn_readlane(m, 63);
    }
#else
    scan_trip = maxF;
#endif
  }

  // ---- most violated inactive inequality row; violation relative to the row norm. id<0: none. ----
  // Box rows: lane = variable.  Corridor rows: lane = (segment, control point), sweeping the faces of the
  // segment's polytope (4 lanes share each face read).  Sets const_bad if a jerk-independent row (segment 0,
  // control points 0..2) is violated.
  __device__ void scan(int& id_out, double& v_out, bool& const_bad) {
    const int lane = opaque(this->lane);  // (lane-derived constants are recomputed here, not kept in registers across the whole solve)
    double bs = 0, bv = 0;
    int bid = -1;
    bool bad = false;
    bool badb = false;  // a violated box row that does not depend on y (N < 4 only; the t = 0 rows are checked before the search)
    double wbj = this->wbj, wbv = this->wbv, wba = this->wba, wcp = this->wcp;
    if constexpr (NORMS_TABLE) row_norms(wbj, wbv, wba, wcp);
    if (lane < nx) {
      const int t = lane / 3, i = lane - 3 * t;
      const double xv = xj;
      const double V = Vc[lane], A = Ac[lane];
      const double val[3] = {xv, V, A};
      const double lim[3] = {jmax, vmax, amax};
      const double inv[3] = {wbj, wbv, wba};  // (zero for rows that do not depend on y: t = 0)
#pragma unroll
      for (int c = 0; c < 3; c++) {  // |value| - limit: at most one of the two box rows of a quantity can be violated
        const double cv = fabs(val[c]) - lim[c];
        const double sc = cv * inv[c];
        badb |= cv > tol && inv[c] == 0.0 && (c == 0 || t > 0);
        const bool take = cv > tol && sc > bs;
        bs = take ? sc : bs;
        bv = take ? cv : bv;
        bid = take ? mk_id(c == 0 ? K_JBOX : (c == 1 ? K_VBOX : K_ABOX), t, i, val[c] < 0.0 ? 1 : 0) : bid;
      }
    }
    {  // corridor rows: wave-uniform trip count (max faces per polytope) so that the face loads of 4 rows are in flight.
       // All rows of a lane share the weight factor, so the lane maximises the normalised violation and scales once.
      const bool live = lane < 4 * N;
      const int t = live ? (lane >> 2) : 0, k = lane & 3;
      const int f0 = scan_f0, F = scan_F;  // rows of the polytope this lane's segment is assigned to (bind_assignment)
      const double c0 = cp_r[0], c1 = cp_r[1], c2 = cp_r[2];  // this lane's control point (compute_states)
      const double wi = wcp;
      int bf = -1;
      double bvt = (F > 0) ? 0.0 : INFINITY;  // dead lanes never take
      const int fl = F > 0 ? F - 1 : 0;        // rows beyond the lane's polytope re-read its last row: never a strict improvement
      const fh_face* fp = faces + f0;
      for (int fb = 0; fb < scan_trip; fb += 4) {
        fh_face fc[4];
#pragma unroll
        for (int j = 0; j < 4; j++) fc[j] = fp[min(fb + j, fl)];  // the four loads in flight before the first use
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const double vt = fma(fc[j].a[0], c0, fma(fc[j].a[1], c1, fma(fc[j].a[2], c2, fc[j].b)));
          const bool take = vt > bvt;  // (an active row has v ~ 0, i.e. vt ~ -tol/|a| < 0: never re-selected)
          bvt = take ? vt : bvt;
          bf = take ? fb + j : bf;
        }
      }
      if (bf >= 0) {
        if (wi == 0.0) bad = true;  // a row that does not depend on y (segment 0: control points 0..2; whole trajectory: 1..3 of the last segment) violated
        else {
          const double sc = bvt * wi;
          if (sc > bs) { bs = sc; bv = bvt + (double)tolf[f0 + bf]; bid = mk_id(K_POLY, t, k, bf); }
        }
      }
    }
    const_bad = wave_any(bad || badb);
    if (const_bad) conflict = wave_or(bad ? (1u << ((lane >> 2) & 31)) : 0u);  // (the y-independent rows of a segment's polytope)
    const double mx = wave_max_nonneg(bs);
    id_out = -1;
    v_out = 0;
    if (mx > 0) {
      const int L = first_lane(bs == mx);
      id_out = __builtin_amdgcn_readlane(bid, L);
      v_out = readlane_f64(bv, L);
    }
  }

  // ---- normal of row `id` in the reduced space: g = (Z (x) I3)^T (normal in jerk space), lane = (k, i).  returns |g|^2 ----
  __device__ double build_g(int id) {
    const int lane = opaque(this->lane);  // (lane-derived constants are recomputed here, not kept in registers across the whole solve)
    const int kind = id >> 24, t = (id >> 16) & 255, k = (id >> 8) & 255, f = id & 255;
    double gv = 0;
    const int kk = lane / 3, ia = lane - 3 * kk;
    if (kind == K_JBOX) {  // x-space normal +-e_(t,k): row t of Z
      if (lane < n) gv = (ia == k) ? (f ? -Zm[t * ZS + zc0 + kk] : Zm[t * ZS + zc0 + kk]) : 0.0;
      gx2 = 1.0;
    } else {
      int wk, tt;
      {  // x-space normal: (3-vector) x (Toeplitz weight of the triple integrator), lane = (s, i)
        double gi;
        if (kind == K_POLY) {
          wk = (k == 3) ? W_P : k;
          tt = t + (k == 3 ? 1 : 0);
          gi = faces[face_off[assign[t]] + f].a[ia];
        } else {
          wk = (kind == K_VBOX) ? W_V : W_A;
          tt = t;
          gi = (ia == k) ? (f ? -1.0 : 1.0) : 0.0;
        }
        const int m = tt - 1 - kk;  // (here kk is the segment s of the x-space lane)
        const double gxv = (lane < nx && m >= 0) ? gi * wcoef(wk, m, h) : 0.0;
        if (lane < NXP) xs[lane] = gxv;
        gx2 = wave_sum(gxv * gxv);
      }
      FH_SYNC();
      const bool on = lane < n;
      const double* zr = Zm + zc0 + (on ? kk : 0);
      const double* gr = xs + (on ? ia : 0);
      const int smax = min(tt, N);  // (the normal is zero from segment tt on)
      const int sl = smax > 0 ? smax - 1 : 0;
      double a0 = 0, a1 = 0;
      for (int s0 = 0; s0 < smax; s0 += 4) {
        double zv[4], xv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int s_ = min(s0 + j, sl);
          zv[j] = zr[s_ * ZS];
          xv[j] = gr[3 * s_];
        }
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          a0 += (s0 + j < smax) ? zv[j] * xv[j] : 0.0;
          a1 += (s0 + j + 1 < smax) ? zv[j + 1] * xv[j + 1] : 0.0;
        }
      }
      gv = on ? a0 + a1 : 0.0;
    }
    if (lane < NVP) g[lane] = gv;
    FH_SYNC();
    return wave_sum(gv * gv);
  }

  // Q1 is stored COLUMN major, Q(i, c) = Q[c * S + i] (the live columns are a contiguous prefix: cheap node snapshots).
  // strided sweep: sum_k M[k * S + col] * v[k] over the padded length n8 (two accumulators, 8 loads in flight)
  __device__ __forceinline__ double col_dot(const double* __restrict__ M, int col, const double* __restrict__ v, int n8) const {
    double a0 = 0, a1 = 0;
    for (int i0 = 0; i0 < n8; i0 += 8) {
      double m[8], w[8];
#pragma unroll
      for (int j = 0; j < 8; j++) { m[j] = M[(i0 + j) * S + col]; w[j] = v[i0 + j]; }  // every load issued before the first use
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        a0 += m[j] * w[j];
        a1 += m[j + 1] * w[j + 1];
      }
    }
    return a0 + a1;
  }
  // contiguous sweep: sum_k M[row * S + k] * v[k] over the padded length q8
  __device__ __forceinline__ double row_dot(const double* __restrict__ M, int row, const double* __restrict__ v, int q8) const {
    double a0 = 0, a1 = 0;
    const double* Mr = M + row * S;
    for (int c0 = 0; c0 < q8; c0 += 8) {
      double m[8], w[8];
#pragma unroll
      for (int j = 0; j < 8; j++) { m[j] = Mr[c0 + j]; w[j] = v[c0 + j]; }
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        a0 += m[j] * w[j];
        a1 += m[j + 1] * w[j + 1];
      }
    }
    return a0 + a1;
  }

  // Two-way split of the same sweeps for NVP <= 32: lanes l and l+32 serve the same row/column, take alternate blocks of
  // four terms and exchange their partial sums, so that the upper half of the wavefront is not idle.
  // (For NVP = 32 a trip covers 16 terms, 8 per half, with every load issued before the first use; the padded length is
  // rounded up to 16, which stays inside the zero padding.)
  static constexpr int SPLIT_U = (NVP % 16 == 0) ? 8 : 4;
  __device__ __forceinline__ double col_dot2(const double* __restrict__ M, int col, const double* __restrict__ v, int n8) const {
    const int lane = opaque(this->lane);  // (lane-derived constants are recomputed here, not kept in registers across the whole solve)
    const int hh = (lane >> 5) * 4;
    const double* Mc = M + hh * S + col;
    const double* vv = v + hh;
    const int nn = (n8 + 2 * SPLIT_U - 1) & ~(2 * SPLIT_U - 1);
    double a0 = 0, a1 = 0;
    for (int i0 = 0; i0 < nn; i0 += 2 * SPLIT_U) {
      double m[SPLIT_U], w[SPLIT_U];
#pragma unroll
      for (int j = 0; j < SPLIT_U; j++) {
        const int k = i0 + (j >> 2) * 8 + (j & 3);
        m[j] = Mc[k * S];
        w[j] = vv[k];
      }
#pragma unroll
      for (int j = 0; j < SPLIT_U; j += 2) {
        a0 += m[j] * w[j];
        a1 += m[j + 1] * w[j + 1];
      }
    }
    return halves_sum(a0 + a1);
  }
  __device__ __forceinline__ double row_dot2(const double* __restrict__ M, int row, const double* __restrict__ v, int q8) const {
    const int lane = opaque(this->lane);  // (lane-derived constants are recomputed here, not kept in registers across the whole solve)
    const int hh = (lane >> 5) * 4;
    const double* Mr = M + row * S + hh;
    const double* vv = v + hh;
    const int nn = (q8 + 2 * SPLIT_U - 1) & ~(2 * SPLIT_U - 1);
    double a0 = 0, a1 = 0;
    for (int c0 = 0; c0 < nn; c0 += 2 * SPLIT_U) {
      double m[SPLIT_U], w[SPLIT_U];
#pragma unroll
      for (int j = 0; j < SPLIT_U; j++) {
        const int k = c0 + (j >> 2) * 8 + (j & 3);
        m[j] = Mr[k];
        w[j] = vv[k];
      }
#pragma unroll
      for (int j = 0; j < SPLIT_U; j += 2) {
        a0 += m[j] * w[j];
        a1 += m[j + 1] * w[j + 1];
      }
    }
    return halves_sum(a0 + a1);
  }

  // ---- z = (I - Q1 Q1^T) g, d = Q1^T g (lane c holds d_c, lane i holds z_i). Re-orthogonalises when the first
  // pass cancels more than half of |g|^2 (Daniel-Gragg-Kaufman-Stewart).  returns |z|^2 ----
  __device__ double project(double gg, double& dc, double& zi) {
    const int lane = opaque(this->lane);  // (lane-derived constants are recomputed here, not kept in registers across the whole solve)
    const int n8 = (n + 7) & ~7, q8 = (q + 7) & ~7;
    if constexpr (NVP <= 32) {
      const int l5 = lane & 31;
      const int ll = l5 < NVP ? l5 : NVP - 1;  // lanes beyond the padded size compute a harmless duplicate
      dc = row_dot2(Q, ll, g, n8);
      if (lane < NVP) d[lane] = dc;
      FH_SYNC();
      zi = g[ll] - col_dot2(Q, ll, d, q8);
      if (lane >= NVP) zi = 0.0;
      double zz = wave_sum(zi * zi);
      if (zz < FH_REORTH_THRESHOLD * gg) {
        if (lane < NVP) z[lane] = zi;
        FH_SYNC();
        const double ec = row_dot2(Q, ll, z, n8);
        dc += ec;
        if (lane < NVP) r[lane] = ec;
        FH_SYNC();
        zi -= col_dot2(Q, ll, r, q8);
        if (lane >= NVP) zi = 0.0;
        zz = wave_sum(zi * zi);
      }
      return zz;
    }
    const int ll = lane < NVP ? lane : NVP - 1;  // lanes beyond the padded size compute a harmless duplicate
    dc = row_dot(Q, ll, g, n8);
    if (lane < NVP) d[lane] = dc;
    FH_SYNC();
    zi = g[ll] - col_dot(Q, ll, d, q8);
    if (lane >= NVP) zi = 0.0;
    double zz = wave_sum(zi * zi);
    if (zz < FH_REORTH_THRESHOLD * gg) {
      if (lane < NVP) z[lane] = zi;
      FH_SYNC();
      const double ec = row_dot(Q, ll, z, n8);
      dc += ec;
      if (lane < NVP) r[lane] = ec;
      FH_SYNC();
      zi -= col_dot(Q, ll, r, q8);
      if (lane >= NVP) zi = 0.0;
      zz = wave_sum(zi * zi);
    }
    return zz;
  }

  // ---- r = R^{-1} d, column-oriented; lane c returns r_c ----
  // Row-scaled recurrence: lane l carries (d_l - sum_{j>l} R_lj r_j) / R_ll, which is final (= r_l) once column l has been
  // consumed, so that the serial chain per column is one broadcast and one FMA; the scaled, masked coefficients
  // R_lc / R_ll do not depend on the recurrence and are prepared four columns ahead.
  // (qe = 0: the final-state equalities are eliminated, every column is an inequality row.)
  __device__ double backsolve(double dc) {
    const int lane = opaque(this->lane);  // (lane-derived constants are recomputed here, not kept in registers across the whole solve)
    const int ll = lane < NVP ? lane : NVP - 1;
    const double ri = (lane < q && lane >= qe) ? rinv[lane] : 0.0;
    double dh = dc * ri;
    int c = q - 1;
    for (; c >= qe + 4; c -= 4) {
      double m[4];
#pragma unroll
      for (int j = 0; j < 4; j++) m[j] = R[rp(min(ll, c - j), c - j)];
#pragma unroll
      for (int j = 0; j < 4; j++) m[j] = (lane < c - j) ? m[j] * ri : 0.0;
#pragma unroll
      for (int j = 0; j < 4; j++) dh = fma(-m[j], readlane_f64(dh, c - j), dh);
    }
    for (; c >= qe + 1; c--) {
      const double m0 = (lane < c) ? R[rp(min(ll, c), c)] * ri : 0.0;
      dh = fma(-m0, readlane_f64(dh, c), dh);
    }
    return dh;
  }

  __device__ void add_row(int id, double zi, double zz, double dc, double up) {
    const int lane = opaque(this->lane);  // (lane-derived constants are recomputed here, not kept in registers across the whole solve)
    const double rho = sqrt(zz);
    const double inv = 1.0 / rho;
    if (lane < n) Q[q * S + lane] = zi * inv;
    if (lane < q) R[rp(lane, q)] = dc;
    if (lane == q) {
      R[rp(q, q)] = rho;
      rinv[q] = inv;
      act[q] = id;
      u[q] = up;
    }
    q++;
    FH_SYNC();
  }

  __device__ void drop_row(int kpos) {
    const int lane = opaque(this->lane);  // (lane-derived constants are recomputed here, not kept in registers across the whole solve)
    // shift the bookkeeping and the columns of R left
    int a_next = 0;
    double u_next = 0;
    if (lane >= kpos && lane < q - 1) { a_next = act[lane + 1]; u_next = u[lane + 1]; }
    FH_SYNC();
    if (lane >= kpos && lane < q - 1) { act[lane] = a_next; u[lane] = u_next; }
    FH_SYNC();
    // Givens rotations applied in place to the OLD columns k+1..q-1 (old column c+1 becomes new column c), then the
    // columns are compacted one slot to the left.
    for (int j = kpos; j < q - 1; j++) {  // zero element (j+1) of old column j+1 against element j
      const double a = R[rp(j, j + 1)], b = R[rp(j + 1, j + 1)];
      const double rr = sqrt(a * a + b * b);
      if (rr != 0.0) {
        const double irr = 1.0 / rr;
        const double cs = a * irr, sn = b * irr;
        if (lane >= j && lane < q - 1) {
          const double t1 = R[rp(j, lane + 1)], t2 = R[rp(j + 1, lane + 1)];
          R[rp(j, lane + 1)] = cs * t1 + sn * t2;
          R[rp(j + 1, lane + 1)] = -sn * t1 + cs * t2;
        }
        if (lane < n) {
          const double t1 = Q[j * S + lane], t2 = Q[(j + 1) * S + lane];
          Q[j * S + lane] = cs * t1 + sn * t2;
          Q[(j + 1) * S + lane] = -sn * t1 + cs * t2;
        }
      }
      FH_SYNC();
    }
    for (int c = kpos; c < q - 1; c++) {  // new column c <- old column c+1 (rows 0..c)
      double v = 0.0;
      if (lane <= c) v = R[rp(lane, c + 1)];
      FH_SYNC();
      if (lane <= c) R[rp(lane, c)] = v;
      FH_SYNC();
    }
    if (lane >= kpos && lane < q - 1) rinv[lane] = 1.0 / R[rp(lane, lane)];
    if (lane < NVP) Q[(q - 1) * S + lane] = 0.0;  // keep the zero padding beyond the active columns
    if (lane < NSEG && stk_keep[lane] > kpos) stk_keep[lane] = kpos;  // the snapshots of all levels: columns from kpos on differ now
    q--;
    FH_SYNC();
  }

  __device__ void reset_qp() {
    const int lane = opaque(this->lane);  // (lane-derived constants are recomputed here, not kept in registers across the whole solve)
    if (lane < NVP) {
      for (int c = 0; c < q; c++) Q[c * S + lane] = 0.0;  // columns >= q are zero already
      x[lane] = 0;
    }
    q = 0;
    FH_SYNC();
  }

  // ---- dual active set from the current (dual feasible) state.
  // returns 0 optimal, 1 infeasible, 2 bounded out by `ub`, 3 iteration limit ----
  // ub: prune when the dual objective (a lower bound at every iteration) exceeds it, or reaches it if `tie` (the incumbent
  // comes earlier in depth-first order than this node: the sequential search keeps the FIRST leaf of minimal cost)
  __device__ int qp_run(double ub, bool tie, int max_iters, int& iters, double& cost) {
    int it = 0;
    if (lane == 0) tb[TB_FRESH] = 0;  // (the states of the incumbent leaf, if they were still there, are about to be overwritten)
    bind_assignment();
    const int q0 = q;
    const int st = qp_loop(ub, tie, max_iters, it, cost);
    iters += it;
    // FP64 flop estimate of this active-set run (useful lanes only; reported as fh_result.kflops), once per node so that the
    // iteration loop carries no bookkeeping (counting per iteration cost 3 % of the throughput).  Per outer iteration (at most
    // it + 1 of them): the jerks Z y (2 x 3N x K), states / control points (3(N+1) lanes x (5N + 9), 4N lanes x 12), the row scan
    // (3N x 6 box rows, 7 per corridor row and control point) and the row normal (4 x 3N in jerk space, 2 n N to reduce it).  Per
    // inner iteration, with the mean number of active rows qa = (q_before + q_after) / 2 and n = 3K reduced unknowns: the two
    // Gram-Schmidt sweeps (4 n qa; a re-orthogonalisation is not counted), the back-substitution (qa^2), the step (3n + 4 qa) and
    // the update of the factors (n + qa).
    const int qa = (q0 + q) >> 1;
    flops += (unsigned long long)(unsigned)(it + 1) * (unsigned)(2 * nx * K + 3 * (N + 1) * (5 * N + 9) + 48 * N + 18 * N + 7 * rows4 + 4 * nx + 2 * n * N) +
             (unsigned long long)(unsigned)it * (unsigned)(4 * n * qa + qa * qa + 4 * n + 5 * qa);
    return st;
  }
  __device__ int qp_loop(double ub, bool tie, int max_iters, int& it, double& cost) {
    for (;;) {
      { FH_T0(); compute_states(); FH_T1(2); }
      int id;
      double vp;
      {
        FH_T0();
        {  // the dual objective is a lower bound: prune against the incumbent, and against the jerk box — every feasible trajectory
           // has |x|^2 <= 3N j_max^2 (setMaxConstraints :403-405), so a node whose lower bound exceeds it is infeasible.  The second
           // test also ends the divergence of an infeasible node before a numerically dependent row (|z| ~ 1e-10 |g|) can be
           // mistaken for an independent one and answered with a step of 1e24.  (NaN-safe: !(cost <= ub).)
          const double xl = (lane < n) ? x[lane] : 0.0;
          cost = c0 + wave_sum(xl * xl);
#ifndef FH_NO_L1_NODE_BOUND
          // [r5] The same certificate in the 1-norm (setup_trial has it for y = 0): the current point x* = xp + Z y* is the cheapest one that
          // satisfies the ACTIVE rows, x* = xp - sum mu_k a_k with mu >= 0 (the invariant of the dual method), so every feasible x has
          // x* . x >= |x*|^2 = cost; inside the jerk box x* . x <= |x*|_1 (j_max + tol).  cost > |x*|_1 (j_max + tol) is therefore
          // infeasible with the same conflict as the 2-norm bound, and never later (Cauchy-Schwarz): a refuted node ends iterations sooner.
          const double l1x = wave_sum(fabs(xj));
          if (!(cost <= box_ub) || cost > l1x * (jmax + tol) * (1.0 + 1e-9)) {
#else
          if (!(cost <= box_ub)) {
#endif
            // Infeasible, with a certificate: the current point is the cheapest one that satisfies the ACTIVE rows (the invariant of
            // the dual method), so the active rows and the jerk box exclude each other — the conflict is the segments whose corridor
            // rows are active (box rows do not depend on any decision).
            const int al = lane < NVP ? act[lane] : 0;
            conflict = wave_or((lane < q && (al >> 24) == K_POLY) ? (1u << ((al >> 16) & 31)) : 0u);
#ifdef FH_TRACE
            trace_src = 3; trace_id = 0; trace_v = cost;
#endif
            return 1;
          }
          if (cost > ub || (tie && cost == ub)) return 2;
        }
        bool cbad;
        scan(id, vp, cbad);
        FH_T1(3);
#ifdef FH_TRACE
        if (cbad) { trace_src = 1; trace_id = id; trace_v = vp; }
#endif
        if (cbad) return 1;
        if (id < 0) return 0;  // optimal (cost: above)
      }
      double gg;
      { FH_T0(); gg = build_g(id); FH_T1(4); }
      double up = 0;
      for (;;) {  // until row `id` is active
        if (++it > max_iters) return 3;
        double dc, zi, zz, rc;
        { FH_T0(); zz = project(gg, dc, zi); FH_T1(5); }
        FH_T0();
        rc = backsolve(dc);
        const bool dependent = zz <= dep2 * gx2;  // relative to the row's norm in JERK space: the rounding noise of z is eps |g_x|
        double ratio = INFINITY;
        if (lane < q && rc > 0) ratio = u[lane] / rc;
        const double t1 = wave_any(ratio < INFINITY) ? wave_min(ratio) : INFINITY;  // (no blocking row in most iterations)
        const int kb = (t1 < INFINITY) ? first_lane(ratio == t1) : -1;
        if (kb < 0 && dependent) {  // infeasible: row `id` is violated and a non-negative combination of active rows
          // The certificate: a_id = sum_c rc a_c over the active rows, and the remaining violation vp > 0.  Mathematically the
          // coefficient of a row that has nothing to do with the dependence is zero; on a warm-started factorisation it is
          // rounding noise, and counting such rows would put their segments into the conflict.  A corridor row may be left out of
          // the certificate if the proof survives without it: for every y whose jerks lie within the jerk box (rows of every node;
          // |y| <= |x| <= sqrt(3N) j_max because x = xp + Z y with xp orthogonal to the range of Z),
          // |sum_dropped rc (a_c.y - b_c)| <= sum_dropped |rc| |a_c| (|y_now| + sqrt(3N) j_max)   (a_c.y_now = b_c on active rows),
          // so rows with |rc| |a_c| below thr = 1e-9 vp / (q (|y_now| + sqrt(3N) j_max)) together cost at most 1e-9 vp, and the
          // remaining rows still prove a violation of vp (1 - 1e-9) > feas_tol.
          const int al = lane < NVP ? act[lane] : 0;
          bool in = lane < q && rc != 0.0;
          if (vp * (1.0 - 1e-9) > tol) {
            double cn = 0.0;  // |a_c|^2 = |R(:, c)|^2
            const int cl = lane < q ? lane : 0;
            for (int r_ = 0; r_ < q; r_++) {
              const double v = R[rp(min(r_, cl), cl)];
              cn += (r_ <= cl) ? v * v : 0.0;
            }
            const double xl = (lane < n) ? x[lane] : 0.0;
            const double xn = sqrt(wave_sum(xl * xl));
            const double thr = 1e-9 * vp / ((double)q * (xn + sqrt((double)nx) * jmax));
            in = in && fabs(rc) * sqrt(cn) > thr;
          }
          conflict = wave_or((in && (al >> 24) == K_POLY) ? (1u << ((al >> 16) & 31)) : 0u) |
                     ((id >> 24) == K_POLY ? (1u << ((id >> 16) & 31)) : 0u);
#ifdef FH_TRACE
          trace_src = 2; trace_id = id; trace_v = vp;
#endif
          return 1;
        }
        const double t2 = dependent ? INFINITY : vp / zz;
        const double t = fmin(t1, t2);
#if defined(FH_TRACE) && FH_TRACE == 2
        if (lane == 0 && trace) {
          double* tr = trace + 6 * (trace_it % 32);
          tr[0] = (double)trace_it; tr[1] = (double)(unsigned)id; tr[2] = vp; tr[3] = zz; tr[4] = gg; tr[5] = t1 < INFINITY ? -t1 : t2;
        }
        trace_it++;
#endif
        if (lane < q) u[lane] -= t * rc;
        up += t;
        if (!dependent) {
          if (lane < n) x[lane] -= t * zi;
          vp -= t * zz;
        }
        FH_T1(6);
        if (t2 <= t1) {
          FH_T0();
          add_row(id, zi, zz, dc, up);
#ifdef FH_QMAX_STAT
          qmax = max(qmax, q);
#endif
          FH_T1(7);
          break;
        }
        FH_SYNC();
        { FH_T0(); drop_row(kb); FH_T1(8); }
      }
    }
  }

  __device__ __forceinline__ unsigned allowed_mask(int t) const {
    unsigned m = P ? (((1u << P) - 1u) & poly_ok) : 0u;
    if (t == 0) m &= allowed_first;
    if (t == N - 1) m &= allowed_last;
    return m;
  }

  // ---- exact screening on jerk-independent indicator rows (same rule as the oracle's screen_constant_rows):
  // control points 0..2 of segment 0 depend on x0 and h only; with the final position forced, control points
  // 1..3 of the last segment depend on xf and h only.  One lane per polytope. ----
  template <class PR>
  __device__ void screen_constant_rows(const PR& pr) {
    const int lane = opaque(this->lane);  // (lane-derived constants are recomputed here, not kept in registers across the whole solve)
    allowed_first = allowed_last = 0xffffffffu;
    if (P == 0) return;
    // one lane per face row (all polytopes side by side), six wave-uniform points per row; a violated row marks its polytope
    double c0[9], cN[9];
    {
      const double h3 = h / 3.0, h23 = 2.0 * h / 3.0, h26 = h * h / 6.0;
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const double p0 = Pc[i], v0 = Vc[i], a0 = Ac[i];  // x0 (LDS, see compute_states)
        const double pf = xfl[i], vf = xfl[3 + i], af = xfl[6 + i];
        c0[i] = p0; c0[3 + i] = p0 + v0 * h3; c0[6 + i] = p0 + v0 * h23 + a0 * h26;
        cN[i] = pf; cN[3 + i] = pf - vf * h3; cN[6 + i] = pf - vf * h23 + af * h26;
      }
    }
    const int nf = face_off[P];
    unsigned bad = 0u;  // bits 0..7: polytope excluded for segment 0, bits 8..15: for the last segment
    for (int fb = 0; fb < nf; fb += 64) {
      const int f = fb + lane;
      const bool valid = f < nf;
      const fh_face fc = faces[valid ? f : 0];
      double w0 = -INFINITY, wN = -INFINITY;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        w0 = vmax_f64(w0, fma(fc.a[0], c0[3 * k], fma(fc.a[1], c0[3 * k + 1], fma(fc.a[2], c0[3 * k + 2], fc.b))));
        wN = vmax_f64(wN, fma(fc.a[0], cN[3 * k], fma(fc.a[1], cN[3 * k + 1], fma(fc.a[2], cN[3 * k + 2], fc.b))));
      }
      int pf = 0;
      for (int j = 1; j < P; j++) pf += (f >= face_off[j]) ? 1 : 0;
      if (valid) bad |= ((w0 > 0.0) ? (1u << pf) : 0u) | ((wN > 0.0) ? (256u << pf) : 0u);
    }
    bad = wave_or(bad);
    allowed_first = ~(bad & 255u);
    if (force_final) allowed_last = ~((bad >> 8) & 255u);
  }

  // ---- leaf test / branching choice for the node just solved. returns branch segment or -1 (leaf) ----
  // The segment to branch on: the one least inside any polytope (the quickest way to a good leaf) — unless the ROOT relaxation of
  // the trial ends outside the corridor (its last segment is inside no polytope: a trajectory that cannot stop in time, the typical
  // infeasible safe problem).  Below the root of such a trial the EARLIEST violated segment is taken: the trajectory is causal
  // (segment t depends on the jerks 0..t only), so deciding the early segments first makes the children's QPs tight and an
  // infeasible trial is refuted in a fraction of the nodes; when the root overshoots the corridor by more than 1.2 braking distances
  // from v_max (v_max^2 / 2 a_max), the root itself branches that way too.  The rule is a function of the trial's root alone (bit 16 of
  // tb[TB_QE], handed on with every frame that is given away), not of the order in which the tree is explored or of who explores
  // it; any rule is exact (it only orders the search), and the oracle uses the same one.
  template <class PR>
  __device__ int analyze(const PR& pr, bool root) {
    const int lane = opaque(this->lane);  // (lane-derived constants are recomputed here, not kept in registers across the whole solve)
    if (P == 0) {
      if (lane < N) fullassign[lane] = -1;
      FH_SYNC();
      return -1;
    }
    for (int pair0 = 0; pair0 < N * P; pair0 += 64) {
      const int pair = pair0 + lane;
      const bool live = pair < N * P;
      const int t = live ? pair / P : 0, p = live ? pair - t * P : 0;
      const bool need = live && assign[t] < 0 && ((allowed_mask(t) >> p) & 1u);
      const int f0 = face_off[p], F = need ? face_off[p + 1] - f0 : 0;
      // every face row is read once for the four control points of the segment; rows beyond the polytope re-read its last
      // row (harmless for a maximum), so that the trip count is wave-uniform and four rows are in flight
      double c[12];
      {
        double st[12];
#pragma unroll
        for (int i = 0; i < 3; i++) { st[i] = Pc[3 * t + i]; st[3 + i] = Vc[3 * t + i]; st[6 + i] = Ac[3 * t + i]; st[9 + i] = Pc[3 * t + 3 + i]; }
        in_flight(st);
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
          for (int i = 0; i < 3; i++) c[3 * k + i] = cp_of(k, st[i], st[3 + i], st[6 + i], st[9 + i]);
      }
      const int fl = F > 0 ? F - 1 : 0;
      const fh_face* fp = faces + f0;
      double worst = -INFINITY;
      for (int fb = 0; fb < maxF; fb += 4) {
        fh_face fc[4];
#pragma unroll
        for (int j = 0; j < 4; j++) fc[j] = fp[min(fb + j, fl)];
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
          for (int k = 0; k < 4; k++)
            worst = vmax_f64(worst, fma(fc[j].a[0], c[3 * k], fma(fc[j].a[1], c[3 * k + 1], fma(fc[j].a[2], c[3 * k + 2], fc[j].b))));
        }
      }
      if (!need) worst = -INFINITY;
      if (live) viol[t * FH_MAX_POLY + p] = (assign[t] < 0 && !need) ? INFINITY : worst;
    }
    FH_SYNC();
    double score = -INFINITY;
    if (lane < N) {
      int full = assign[lane];
      if (full < 0) {
        double mn = INFINITY;
        for (int p = 0; p < P; p++) {
          const double v = viol[lane * FH_MAX_POLY + p];
          if (v < mn) { mn = v; full = p; }
        }
        score = mn;
      }
      fullassign[lane] = full;
    }
    const double bw = wave_max_nonneg(score);
    FH_SYNC();
    if (!(bw > 0.0)) return -1;  // (normalised rows carry the tolerance)
    if (root) {
      const bool ends_outside = __ballot(lane == N - 1 && score > 0.0) != 0ull;
      const bool far_outside = __ballot(lane == N - 1 && score > 1.2 * (vmax * vmax) / (2.0 * amax)) != 0ull;
      if (lane == 0) tb[TB_QE] = ends_outside ? (1 << 16) : 0;
      return first_lane(far_outside ? score > 0.0 : score == bw);
    }
    const bool earliest = (uniform_i32(tb[TB_QE]) >> 16) != 0;
    return first_lane(earliest ? score > 0.0 : score == bw);
  }

  // =================================================================================================================
  // Branch and bound (one tree per factor trial) with work sharing between wavefronts (fh_share.hip.hpp)
  // =================================================================================================================
  // ---- wave-uniform state of the search ----
  int rec;                             // share record of the current problem (-1: nothing has been given away)
  int depth0;                          // tree level of this worker's stack frame 0 (0 for the owner of the root)
  unsigned long long cur_key;          // DFS key of the current node: 3 bits per tree level (child rank), level 0 in bits 45..47
  unsigned long long best_key;         // DFS key of the leaf that gave best_cost (~0: unknown / none)
  unsigned long long flops;            // FP64 flop estimate of the work done by this worker on the current problem
  int rows4;                           // 4 x faces of the polytopes of the assigned segments (flop accounting of the row scan)
  // Conflict-directed backjumping.  A node that the active-set method proves infeasible comes with a Farkas certificate: the
  // violated row and the active rows with a non-zero coefficient in its representation.  `conflict` = the segments whose corridor
  // rows are in it (box and final-state rows do not depend on any decision).  When every child of a branching is infeasible and
  // the union of their conflicts, minus the branched segment, does not contain the segment decided one level up, the siblings at
  // that level are infeasible for the same reason and are skipped (every complete assignment below them contains one of the
  // certificates).  Infeasible subtrees hold no leaf, so the result is unchanged; mostly-infeasible trials — the refutations that
  // dominate the hardest problems — shrink 2-6x (measured on the CPU restatement first).
#ifdef FH_TRACE
  double* trace;
  int trace_src, trace_id, trace_it;
  double trace_v;
#endif
  int fin_solved;                      // of the problem run_problem finished last (the fused pair kernel's hand-off reads them and the
  double fin_dt;                       //   coefficient table left in the LDS of Q instead of the result record in memory)
  unsigned conflict;                   // of the node just found infeasible
  unsigned allinf;                     // bit d: every child of stack frame d tried so far was infeasible (and none was given away)

  // what the worker is working on lives in LDS (tb[]): unit index TB_B, TB_PHASE 0 = whole / only problem, 1 = safe problem of a
  // pair, TB_F factor of the current trial, TB_BASE max(dt_initial, 2 DC), TB_TRIALS trials_ so far (including the current one)
  __device__ __forceinline__ void tb_put64(int at, unsigned long long v) { tb[at] = (int)(unsigned)v; tb[at + 1] = (int)(unsigned)(v >> 32); }
  __device__ __forceinline__ unsigned long long tb_get64(int at) const {
    return ((unsigned long long)(unsigned)uniform_i32(tb[at + 1]) << 32) | (unsigned)uniform_i32(tb[at]);
  }

  static __device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
  }
  static __device__ __forceinline__ unsigned long long f64_bits(double v) { return (unsigned long long)__double_as_longlong(v); }
  static __device__ __forceinline__ double bits_f64(unsigned long long v) { return __longlong_as_double((long long)v); }

  // private workspace -> shared task slot: plain loads, 8-byte write-through stores (four in flight per lane).
  // snapshot_save / snapshot_restore move 16 bytes per lane, i.e. an EVEN number of doubles: the same rounding here, or the
  // odd element (a zero of the padding) would be restored from whatever the taker's workspace held before.
  __device__ __forceinline__ void copy_out_shared(double* dst, const double* __restrict__ src, int count_) const {
    const int count = (count_ + 1) & ~1;
    for (int i0 = 0; i0 < count; i0 += 256) {
      double t[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int i = i0 + j * 64 + lane;
        t[j] = src[i < count ? i : 0];
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int i = i0 + j * 64 + lane;
        if (i < count) wt_store(dst + i, t[j]);
      }
    }
  }
  // shared task slot -> private workspace: L1-bypassing loads, plain stores
  __device__ __forceinline__ void copy_in_shared(double* __restrict__ dst, const double* src, int count_) const {
    const int count = (count_ + 1) & ~1;
    for (int i0 = 0; i0 < count; i0 += 256) {
      double t[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int i = i0 + j * 64 + lane;
        t[j] = cc_load(src + (i < count ? i : 0));
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int i = i0 + j * 64 + lane;
        if (i < count) dst[i] = t[j];
      }
    }
  }
  // the incumbent's assignment as two words, one byte per segment (lane t < N holds segment t)
  __device__ __forceinline__ void pack_bytes(int v, int count, unsigned long long& lo, unsigned long long& hi) const {
    const unsigned byte = (lane < 16) ? ((unsigned)((lane < count ? v : -1) & 0xff) << (8 * (lane & 3))) : 0u;
    unsigned w[4];
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = wave_or((lane >> 2) == k ? byte : 0u);
    lo = ((unsigned long long)w[1] << 32) | w[0];
    hi = ((unsigned long long)w[3] << 32) | w[2];
  }
  static __device__ __forceinline__ int unpack_byte(unsigned long long lo, unsigned long long hi, int t) {
    return (int)(signed char)(unsigned char)((t < 8 ? lo : hi) >> (8 * (t & 7)));
  }

  // Every FH_LOOK_EVERY-th node of a tree the worker looks around: has the host (StopExecution, another thread) or the deadline
  // asked to stop (a12, solverGurobi.cpp:15-39: the reference polls its flag in a Gurobi callback)?  Is somebody out of work?
  // returns bit 0: stop, bit 1: a frame may be published, bit 2: a workgroup without work is waiting right now, bits 8..13: how many
  __device__ int look_around(const ShareArgs& sa, int local_nodes, int iters_so_far = 0) {
    FH_SP_T0();
    int flags = 0;
    if (lane == 0) {
      const unsigned long long t_start = ((unsigned long long)(unsigned)tb[TB_T0 + 1] << 32) | (unsigned)tb[TB_T0];
      // two 8-byte loads: {error, interrupted} and {wait_ticket, q_tail}
      const unsigned long long ei = ald(reinterpret_cast<unsigned long long*>(&sa.ctl->error));
      const unsigned long long wt = ald(reinterpret_cast<unsigned long long*>(&sa.ctl->wait_ticket));
      const unsigned int waiters = (unsigned int)wt, tail = (unsigned int)(wt >> 32);
      unsigned int stop = (unsigned int)ei | (unsigned int)(ei >> 32);
      if (!stop) {
        if (sa.deadline_ticks && wall_ticks() - t_start > sa.deadline_ticks) stop = 2u;
        else if (sa.host_abort && (local_nodes & 15) == 0 &&  // the host's word costs a PCIe round trip: every 16th node of a tree
                 __hip_atomic_load(sa.host_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) stop = 1u;  // (everybody reads ctl->interrupted)
        if (stop) ast(&sa.ctl->interrupted, stop);
      }
      if (stop) flags = 1;
      else if (sa.enabled && tail < waiters + (unsigned int)sa.backlog) {
        flags = 2 | ((int)(waiters - tail) > 0 ? 4 : 0) | (min((int)(waiters - tail), 63) << 8);  // a taker is waiting (4: idle right now; count), or the backlog has room
        if (!(flags & 4) && sa.giant_factor > 0 && iters_so_far > 0) {
          // nobody idle, room in the backlog: has this problem used giant_factor times the mean of the units finished so far?
          const unsigned long long dw = ald(reinterpret_cast<unsigned long long*>(&sa.ctl->done));
          const unsigned long long done = dw & 0xffffffffull, sum = dw >> 32;
          // ... and only in a launch whose workgroups are all resident: with other launches in flight the tail of this one is hidden
          // behind their bulk and a hop is pure overhead
          if (done >= 256ull && (unsigned long long)iters_so_far * done >= (unsigned long long)sa.giant_factor * sum &&
              ald(&sa.ctl->started) == gridDim.x)
            flags |= 8;
        }
      }
    }
    FH_SP_ADD(prof, 0, 1);
    return uniform_i32(flags);
  }

  int trial;        // 0-based index of the factor trial this worker is exploring
  int trial_end;    // this worker goes on with the factors trial + 1 .. trial_end - 1 of the problem when the current trial is done
                    // (the owner of a fresh problem: the whole window; the taker of a trial frame: its range; shrinks when trials are
                    // given away; a taker of a tree frame has trial_end = trial + 1: nothing to go on with)

  // First half of every donation.  A frame number first (ONE attempt): from then on a taker is committed to that number, so
  // something MUST be published.  Then the problem's share record: created on its first donation (the incumbent and the
  // bookkeeping move there), one more outstanding part otherwise.  ~0ull: nothing to publish (no taker, or no record left — an
  // empty frame has been published for the taker).
  __device__ __forceinline__ unsigned long long begin_donation(const ShareArgs& sa, double best_cost) {
    unsigned long long pos = ~0ull;
    if (lane == 0) pos = q_reserve(sa);
    pos = uniform_u64(pos);
    if (pos == ~0ull) return pos;
    if (rec < 0) {
      int r = -1;
      if (lane == 0) {
        const unsigned int got = aadd(&sa.ctl->rec_next, 1u);
        if (got < (unsigned)FH_NRECS) r = (int)got;
        else aadd(&sa.ctl->rec_full, 1u);
      }
      r = uniform_i32(r);
      if (r < 0) {  // no record left: publish an empty frame (its taker draws a new ticket) and keep the work
        if (lane == 0) {
          wt_store(&slot_hdr(sa, pos)->w[TH_REC_B], 0xffffffffull);  // rec = -1
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          q_publish(sa, pos);
        }
        return ~0ull;
      }
      ShareRec* R_ = sa.recs + r;
      unsigned long long alo, ahi;
      pack_bytes(lane < NSEG ? bestassign[lane] : -1, N, alo, ahi);
      if (lane < n) wt_store(&R_->x[lane], bestx_g[lane]);
      if (lane == 0) {
        ast(&R_->lock, 0u);
        ast(&R_->pending, 2);  // this worker's part + the frame about to be published
        ast(&R_->inc_rank, best_cost < INFINITY ? rank_pack(trial, best_cost) : FH_RANK_NONE);
        ast(&R_->inc_cost, f64_bits(best_cost));
        ast(&R_->inc_key, best_key);
        ast(reinterpret_cast<unsigned long long*>(&R_->inc_f), tb_lane0_64(TB_F));
        ast(reinterpret_cast<unsigned long long*>(&R_->inc_h), f64_bits(h));
        ast(&R_->nodes, 0); ast(&R_->iters, 0); ast(&R_->last_status, 0u); ast(&R_->limit_kind, 0u); ast(&R_->flops, 0ull); ast(&R_->limited, 0ull);
        ast(&R_->assign_lo, alo); ast(&R_->assign_hi, ahi);
      }
      rec = r;
      // the safe problem of a fused pair exists only in this wavefront's LDS: whoever takes a frame of it stages it from memory
      if (!sa.pair_outputs && sa.safe && uniform_i32(tb[TB_PHASE]) == 1 && !(uniform_i32(tb[TB_SAFE]) & 0x100)) {
        const int unit_ = uniform_i32(tb[TB_B]);
        FH_SYNC();
        write_safe_problem(sa.whole + unit_, sa.wfaces, sa.safe + unit_, sa.sfaces, (lds_cdouble*)Pc, 3 * NT, uniform_i32(tb[TB_SAFE]) & 0xff, P, sa.shrink, sa.r_margin, lane);
        if (lane == 0) tb[TB_SAFE] |= 0x100;
        FH_SYNC();
      }
    } else if (lane == 0) {
      aadd(&(sa.recs + rec)->pending, 1);
    }
    return pos;
  }
  // (lane 0 only) a 64-bit word of tb[]
  __device__ __forceinline__ unsigned long long tb_lane0_64(int at) const { return ((unsigned long long)(unsigned)tb[at + 1] << 32) | (unsigned)tb[at]; }
  static __device__ __forceinline__ unsigned long long pack2(int lo_, int hi_) { return (unsigned long long)(unsigned)lo_ | ((unsigned long long)(unsigned)hi_ << 32); }

  // Give the shallowest stack frame that still has untried children to the queue: the children restart from the parent's
  // factorisation (the snapshot of that level) exactly as they would have here.
  __device__ void donate(const ShareArgs& sa, double* __restrict__ ws, int depth, double best_cost) {
    FH_SP_T0();
    const bool has = lane < depth && stk_next[lane < NSEG ? lane : 0] < stk_cnt[lane < NSEG ? lane : 0];
    const int d = first_lane(has);
    if (d < 0) return;
    const unsigned long long pos = begin_donation(sa, best_cost);
    if (pos == ~0ull) return;
    // partial assignment at the parent of frame d: the decisions of the frames d.. are undone; child order of the frame
    int a = -1;
    if (lane < N) {
      a = assign[lane];
      for (int dd = d; dd < depth; dd++)
        if (stk_seg[dd] == lane) a = -1;
    }
    unsigned long long alo, ahi, olo, ohi;
    pack_bytes(a, N, alo, ahi);
    pack_bytes(lane < FH_MAX_POLY ? (int)stk_order[d * FH_MAX_POLY + (lane & 7)] : -1, FH_MAX_POLY, olo, ohi);
    const int sh = 3 * (15 - (depth0 + d));
    const unsigned long long prefix = (cur_key >> (sh + 3)) << (sh + 3);
    const int qs = stk_q[d];
    TaskHdr* th = slot_hdr(sa, pos);
    if (lane == 0) {
      wt_store(&th->w[TH_REC_B], pack2(rec, tb[TB_B]));
      wt_store(&th->w[TH_PHASE_DEPTH], pack2(tb[TB_PHASE], depth0 + d));
      wt_store(&th->w[TH_KEY], prefix);
      wt_store(&th->w[TH_H], f64_bits(h));
      wt_store(&th->w[TH_F], tb_lane0_64(TB_F));
      wt_store(&th->w[TH_BASE], tb_lane0_64(TB_BASE));
      wt_store(&th->w[TH_TRIALS_SEG], pack2(trial + 1, stk_seg[d]));
      wt_store(&th->w[TH_CNT_NEXT], pack2(stk_cnt[d], stk_next[d]));
      wt_store(&th->w[TH_Q_QE], pack2(qs, tb[TB_QE]));  // (qe = 0 | the trial's branching rule << 16)
      wt_store(&th->w[TH_ORDER], olo);
      wt_store(&th->w[TH_ASSIGN_LO], alo);
      wt_store(&th->w[TH_ASSIGN_HI], ahi);
      wt_store(&th->w[TH_KIND], 0ull);
    }
    // the parent's dual active-set state as snapshot_save left it (written by this wavefront: drain its stores first)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const double* src = ws + (size_t)d * SNAP_PADDED;
    double* snap = slot_snap(sa, pos);
    if (d < TC) {  // (tail and child bounds of the first levels live in LDS)
      copy_out_shared(snap, tcache + d * SNAP_TAIL, SNAP_TAIL);
      copy_out_shared(snap + SNAP_BOUNDS, tbnd + d * FH_MAX_POLY, FH_MAX_POLY);
    } else {
      copy_out_shared(snap, src, SNAP_TAIL_COPY);
    }
    copy_out_shared(snap + SNAP_QOFF, src + SNAP_QOFF, qs * S);
    copy_out_shared(snap + SNAP_ROFF, src + SNAP_ROFF, (qs * (qs + 1)) / 2);
    drain_stores();
    if (lane == 0) {
      q_publish(sa, pos);
      aadd(&sa.ctl->donated, 1u);
      stk_cnt[d] = stk_next[d];  // the frame has no untried children left here
    }
    allinf &= ~(1u << d);        // (its other children are explored elsewhere: nothing can be concluded about the parent here)
    FH_SYNC();
    FH_SP_ADD(prof, 2, 1);
  }

  // Give the remaining factor trials of the problem away (genNewTraj's loop `for (f = f_init; f <= f_final && !solved; f += f_inc)`,
  // solverGurobi.cpp:445-446, is the outermost level of the problem's tree): the taker starts the next trial from its root while
  // this one is still being explored, and passes the trials after that on in the same way.  Speculative — a later factor only
  // matters if every earlier one turns out infeasible — so only problems that have proved hard do it, only while somebody is
  // idle, and never once the current trial has a feasible leaf.  The reference rebuilds its model from scratch for every trial
  // (:447-466): trials are independent.
  template <class PR>
  __device__ void donate_trials(const ShareArgs& sa, const PR& pr, double best_cost, int takers) {
    if (best_cost < INFINITY || trial >= 4000) return;
    // the trials this worker still owns after the current one: [trial + 1, end), end limited by the factor window
    double fk = bits_f64(tb_get64(TB_F));
    int end = trial + 1;
    for (double f2 = fk + pr.f_inc; f2 <= pr.f_final && end < trial_end; f2 = f2 + pr.f_inc) end++;  // accumulated exactly as the loop does
    int remaining = end - (trial + 1);
    if (remaining <= 0) return;
    // split them over the idle takers: consecutive ranges, published back to front so that this worker keeps what cannot be placed
    if (takers < 1) takers = 1;
    const int per = (remaining + takers - 1) / takers;
    while (remaining > 0) {
      const int k0 = max(trial + 1, end - per);  // this frame: trials [k0, end)
      double f0 = fk;
      for (int k = trial; k < k0; k++) f0 = f0 + pr.f_inc;
      const unsigned long long pos = begin_donation(sa, best_cost);
      if (pos == ~0ull) break;
      TaskHdr* th = slot_hdr(sa, pos);
      if (lane == 0) {
        wt_store(&th->w[TH_REC_B], pack2(rec, tb[TB_B]));
        wt_store(&th->w[TH_PHASE_DEPTH], pack2(tb[TB_PHASE], 0));
        wt_store(&th->w[TH_F], f64_bits(f0));
        wt_store(&th->w[TH_BASE], tb_lane0_64(TB_BASE));
        wt_store(&th->w[TH_TRIALS_SEG], pack2(k0 + 1, end));
        wt_store(&th->w[TH_KIND], 1ull);
      }
      drain_stores();
      if (lane == 0) {
        q_publish(sa, pos);
        aadd(&sa.ctl->donated, 1u);
        aadd(&sa.ctl->max_fill, 1u);  // (statistics: trial frames)
      }
      remaining -= end - k0;
      end = k0;
      trial_end = k0;
    }
  }

  // How many tickets the next draw takes: FH_TICKET_CHUNK while every workgroup of the launch still has that many units ahead of it,
  // then 2, then 1 (the units a workgroup holds but has not started cannot be given away: the tail of a launch stays one unit deep)
  // the tickets that are dealt to the workgroups at the start of a launch instead of being drawn: c0 each (as many as a chunk, fewer when the
  // batch is small), static_tickets in total; the ticket counter of the launch counts the tickets behind them
  static __device__ __forceinline__ int static_chunk(int n, int grid) { return n >= grid ? min(FH_TICKET_CHUNK, n / max(grid, 1)) : 1; }
  static __device__ __forceinline__ int static_tickets(int n, int grid) { return min(n, static_chunk(n, grid) * grid); }
  // A dealt chunk that nobody has claimed (its workgroup has not started yet): -1 if the one this call may look at is taken.  Called by a
  // workgroup for which the ticket counter has run dry.  The k-th such call of a launch (ctl->steal_tries) looks at chunk grid - 1 - k —
  // the workgroups that start last first — and at no other: every workgroup runs dry at least once, so every chunk is looked at once
  // unless its workgroup has claimed it before; no loop, no shared scan, three memory round trips at the end of a workgroup's life.
  static __device__ __forceinline__ int steal_chunk(ShareCtl* ctl, unsigned int* claims, int grid, int lane) {
    int got = -1;
    if (lane == 0) {
      const unsigned int k = aadd(&ctl->steal_tries, 1u);
      if (k < (unsigned int)grid) {
        const int idx = grid - 1 - (int)k;
        unsigned int expect = 0u;
        if (__hip_atomic_compare_exchange_strong(&claims[idx], &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, FH_AGENT)) got = idx;
      }
    }
    return uniform_i32(got);
  }
  static __device__ __forceinline__ int ticket_chunk(int n, int grid, int last_ticket) {
    const long long left = (long long)n - (long long)last_ticket;
    if (FH_TICKET_CHUNK >= 4 && left >= 4ll * FH_TICKET_CHUNK * (long long)grid / 4) return FH_TICKET_CHUNK;
    if (FH_TICKET_CHUNK >= 2 && left >= 2ll * (long long)grid) return 2;
    return 1;
  }
  // [r6] The tickets this workgroup holds but has not started go to a workgroup that waits for work: a frame of kind 2 (header only:
  // the range).  A chunk of tickets is drawn — or dealt — four at a time, and a workgroup whose problem in hand runs for 2 ms kept the
  // other three hostage: what one launch alone ended on were 20-us pairs that BEGAN at 2.7 ms (scripts/r4/tail_shape.py), behind
  // the 380 hardest problems, which the launch order puts first.  Called where a problem looks around (every FH_LOOK_EVERY-th node of a
  // tree: a problem that gets there is not a short one); the frames wait ahead of the takers, up to FH_TICKET_BACKLOG of them.
  __device__ __forceinline__ void give_tickets(const ShareArgs& sa) {
    const int a = uniform_i32(tb[TB_NEXT]), b = uniform_i32(tb[TB_NEXT + 1]);
    if (a >= b) return;
    unsigned long long pos = ~0ull;
    if (lane == 0) pos = q_reserve(sa, FH_TICKET_BACKLOG);  // (ahead of the takers: a workgroup that runs dry finds them at once; header-only frames)
    pos = uniform_u64(pos);
    if (pos == ~0ull) return;
    if (lane == 0) {
      TaskHdr* th = slot_hdr(sa, pos);
      wt_store(&th->w[TH_REC_B], pack2(0, 0));
      wt_store(&th->w[TH_TRIALS_SEG], pack2(a, b));
      wt_store(&th->w[TH_KIND], 2ull);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      q_publish(sa, pos);
      aadd(&sa.ctl->donated, 1u);
      tb[TB_NEXT] = b;  // (the pool is empty now: the next hand-off draws ahead as usual)
    }
    FH_SYNC();
  }
  // Units finished here are reported FH_DONE_BATCH at a time (one 8-byte add: count | iterations << 32); flush_done before waiting / leaving
  __device__ __forceinline__ void unit_done(const ShareArgs& sa) {
    if (lane == 0) {
      const int nd = tb[TB_DONE_N] + 1;
      const unsigned int it = (unsigned int)tb[TB_DONE_IT] + (unsigned int)min(tb[TB_WORK], 0xfffff);
      if (nd >= FH_DONE_BATCH) {
        aadd(reinterpret_cast<unsigned long long*>(&sa.ctl->done), (unsigned long long)(unsigned)nd | ((unsigned long long)min(it, 0xfffffffu) << 32));
        tb[TB_DONE_N] = 0; tb[TB_DONE_IT] = 0;
      } else {
        tb[TB_DONE_N] = nd; tb[TB_DONE_IT] = (int)it;
      }
    }
  }
  __device__ __forceinline__ void flush_done(const ShareArgs& sa) {
    if (lane == 0 && tb[TB_DONE_N] > 0) {
      aadd(reinterpret_cast<unsigned long long*>(&sa.ctl->done), (unsigned long long)(unsigned)tb[TB_DONE_N] | ((unsigned long long)(unsigned)tb[TB_DONE_IT] << 32));
      tb[TB_DONE_N] = 0; tb[TB_DONE_IT] = 0;
    }
  }

  // Out of fresh problems: draw a wait ticket and poll the slot the frame with that number will arrive in.  false: every unit
  // is done, enough others are waiting already, or the launch failed — leave.  On success the frame's snapshot is in workspace
  // level 0, the frame itself is stack level 0 in LDS (assignment at the parent, child order, untried children) and its header
  // words are in tb[]; nothing of it occupies registers while the problem is staged.
  // returns 0: leave / nothing; 1: a stack frame of a tree; 2: the remaining trials of a problem (tb[]: TB_TRIALS - 1 = first
  // trial, TB_F its factor).  idle: this workgroup has no fresh problem to draw (it waits for a frame); otherwise it only takes a
  // frame that is pending right now (one compare-and-swap, only when there is one) and returns 0 at once if there is none.
  __device__ __forceinline__ int take_task(const ShareArgs& sa, double* __restrict__ ws, bool idle) {
   for (;;) {  // (an empty frame — its donor found no share record — sends the taker back for a new ticket)
    unsigned long long pos = ~0ull;
    int state = 0;  // 1: got a frame, 2: leave
    if (lane == 0) {
      const unsigned long long wt = ald(reinterpret_cast<unsigned long long*>(&sa.ctl->wait_ticket));
      const unsigned int waiters = (unsigned int)wt, tail = (unsigned int)(wt >> 32);
      if (idle) {
        if ((int)(waiters - tail) >= sa.max_hungry) state = 2;           // enough idle hands already (no ticket drawn: free to go)
        else pos = (unsigned long long)aadd(&sa.ctl->wait_ticket, 1u);  // committed to frame number `pos` from here on
      } else {
        unsigned int expect = waiters;
        if ((int)(tail - waiters) > 0 &&
            __hip_atomic_compare_exchange_strong(&sa.ctl->wait_ticket, &expect, waiters + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, FH_AGENT))
          pos = (unsigned long long)waiters;  // frame `waiters` is published or about to be (its donor is copying it)
        else state = 2;
      }
    }
    state = uniform_i32(state);
    if (state == 2) return 0;
    unsigned long long t0 = wall_ticks();  // (lane 0) the watchdog measures LACK OF PROGRESS: restarted whenever a unit finishes or a
    unsigned int seen = 0u;                //  frame is given away anywhere in the launch — a long tail that is still moving is not a failure
    FH_SP_T0();
    for (unsigned round = 0;; round++) {
      if (lane == 0) {
        if (q_arrived(sa, pos)) state = 1;
        else if ((round & 7u) == 7u) {
          const unsigned int done = ald(&sa.ctl->done), err = ald(&sa.ctl->error);
          const unsigned int progress = done + ald(&sa.ctl->donated);
          if (progress != seen) { seen = progress; t0 = wall_ticks(); }
          if (!err && !ald(&sa.ctl->interrupted) && sa.host_abort &&
              __hip_atomic_load(sa.host_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM))
            ast(&sa.ctl->interrupted, 1u);  // (an idle workgroup relays the host's stop request to the busy ones)
          if (err || done >= (unsigned)sa.total_units) state = 2;  // no frame will be published any more
          else if (wall_ticks() - t0 > FH_WATCHDOG_TICKS) { ast(&sa.ctl->error, 4u); state = 2; }
        }
      }
      state = uniform_i32(state);
      if (state) break;
      __builtin_amdgcn_s_sleep(32);  // ~1 us between polls of this workgroup's own word, ~5 us once nothing has come for a while
      if (round > 64) __builtin_amdgcn_s_sleep(127);
    }
    if (state == 2) return 0;
    FH_SP_ADD(prof, 4, 1);
    pos = uniform_u64(pos);
#ifdef FH_SHARE_PROFILE
    const unsigned long long sp_t1__ = wall_ticks();
#endif
    acquire_agent();  // (plain loads of the problem record / faces below: a pair's safe problem was written inside this launch)
    const TaskHdr* hp = slot_hdr(sa, pos);
    FH_SYNC();
    const unsigned long long w_rec_b = cc_load(&hp->w[TH_REC_B]), w_phase_depth = cc_load(&hp->w[TH_PHASE_DEPTH]);
    if ((int)(unsigned)uniform_u64(w_rec_b) < 0) {
      if (lane == 0) q_release(sa, pos);
      if (!idle) return 0;
      continue;
    }
    const unsigned long long w_kind = uniform_u64(cc_load(&hp->w[TH_KIND]));
    if (w_kind == 2ull) {  // a range of fresh tickets another workgroup held and had not started (give_tickets): header only
      const unsigned long long w_ts = cc_load(&hp->w[TH_TRIALS_SEG]);
      if (lane == 0) {
        const int at = tb[TB_NEXT] < tb[TB_NEXT + 1] ? TB_POOL2 : TB_NEXT;  // (behind this workgroup's own tickets, if it has any)
        tb[at] = (int)(unsigned)w_ts; tb[at + 1] = (int)(w_ts >> 32);
        q_release(sa, pos);
        aadd(&sa.ctl->stolen, 1u);
      }
      FH_SYNC();
      return 3;
    }
    if (w_kind != 0ull) {  // the remaining trials of a problem: header only
      const unsigned long long w_ts = cc_load(&hp->w[TH_TRIALS_SEG]);
      if (lane == 0) {
        tb[TB_REC] = (int)(unsigned)w_rec_b; tb[TB_B] = (int)(w_rec_b >> 32); tb[TB_WORK] = 0; tb[TB_WORK] = 0;
        tb[TB_PHASE] = (int)(unsigned)w_phase_depth;
        tb[TB_TRIALS] = (int)(unsigned)w_ts;
        tb[TB_DEPTH0] = (int)(w_ts >> 32);  // (trial frames: the end of the range)
        tb_put64(TB_F, cc_load(&hp->w[TH_F])); tb_put64(TB_BASE, cc_load(&hp->w[TH_BASE]));
      }
      drain_stores();
      FH_SYNC();
      if (lane == 0) {
        q_release(sa, pos);
        aadd(&sa.ctl->stolen, 1u);
      }
      return 2;
    }
    const unsigned long long w_trials_seg = cc_load(&hp->w[TH_TRIALS_SEG]), w_cnt_next = cc_load(&hp->w[TH_CNT_NEXT]);
    const unsigned long long w_q_qe = cc_load(&hp->w[TH_Q_QE]), w_order = cc_load(&hp->w[TH_ORDER]);
    const unsigned long long w_alo = cc_load(&hp->w[TH_ASSIGN_LO]), w_ahi = cc_load(&hp->w[TH_ASSIGN_HI]);
    if (lane < NSEG) assign[lane] = unpack_byte(w_alo, w_ahi, lane);
    if (lane < FH_MAX_POLY) stk_order[lane] = (signed char)unpack_byte(w_order, 0ull, lane);
    if (lane == 0) {
      stk_seg[0] = (int)(w_trials_seg >> 32); stk_cnt[0] = (int)(unsigned)w_cnt_next; stk_next[0] = (int)(w_cnt_next >> 32);
      stk_q[0] = (int)(unsigned)w_q_qe;
      tb[TB_REC] = (int)(unsigned)w_rec_b; tb[TB_B] = (int)(w_rec_b >> 32); tb[TB_WORK] = 0;
      tb[TB_PHASE] = (int)(unsigned)w_phase_depth; tb[TB_DEPTH0] = (int)(w_phase_depth >> 32);
      tb[TB_TRIALS] = (int)(unsigned)w_trials_seg;
      tb[TB_QE] = (int)(w_q_qe >> 32);
      tb_put64(TB_F, cc_load(&hp->w[TH_F])); tb_put64(TB_BASE, cc_load(&hp->w[TH_BASE])); tb_put64(TB_H, cc_load(&hp->w[TH_H]));
      tb_put64(TB_KEY, cc_load(&hp->w[TH_KEY]));
    }
    const int qs = uniform_i32((int)(unsigned)w_q_qe);
    const double* snap = slot_snap(sa, pos);
    if (TC > 0) {  // the frame becomes stack level 0 of this worker: its tail and child bounds go where level 0 keeps them
      for (int i = lane; i < SNAP_TAIL; i += 64) tcache[i] = cc_load(snap + i);
      if (lane < FH_MAX_POLY) tbnd[lane] = cc_load(snap + SNAP_BOUNDS + lane);
    } else {
      copy_in_shared(ws, snap, SNAP_TAIL_COPY);
    }
    copy_in_shared(ws + SNAP_QOFF, snap + SNAP_QOFF, qs * S);
    copy_in_shared(ws + SNAP_ROFF, snap + SNAP_ROFF, (qs * (qs + 1)) / 2);
    drain_stores();  // (the loads have returned: their values were stored)
    FH_SYNC();
    if (lane == 0) {
      q_release(sa, pos);
      aadd(&sa.ctl->stolen, 1u);
#ifdef FH_SHARE_PROFILE
      aadd(&sa.ctl->prof[6], wall_ticks() - sp_t1__);
      aadd(&sa.ctl->prof[7], 1ull);
#endif
    }
    return 1;
   }
  }

  // The frame taken from the queue (stack level 0, tb[]) becomes current: problem staged, trial set up for its step.
  template <class PR>
  __device__ void install_frame(const PR& pr, const ShareArgs& sa, double& best_cost) {
    screen_constant_rows(pr);  // allowed_first / allowed_last of this trial
    qe = uniform_i32(tb[TB_QE]) & 0xffff;
    q = 0;  // (LDS factors are all zero after init_problem: nothing to clear when the snapshot is restored)
    if (lane < NSEG) stk_keep[lane] = 0;  // ... and nothing of the snapshot is in LDS yet
    depth0 = uniform_i32(tb[TB_DEPTH0]);
    cur_key = tb_get64(TB_KEY);
    // the incumbent's (rounded-up) cost prunes; its key is only compared under the record's lock (publish_incumbent)
    const unsigned long long r = uniform_u64(ald(&(sa.recs + rec)->inc_rank));
    best_cost = rank_trial(r) == trial ? rank_cost_up(r) : INFINITY;  // (an incumbent of an earlier trial ends the search at its first node)
    best_key = ~0ull;
  }

  // a better leaf was found by a worker of a shared problem: it enters the record if it beats the record lexicographically
  // (trial, cost, DFS key)
  __device__ void publish_incumbent(const ShareArgs& sa, double cost) {
    ShareRec* R_ = sa.recs + rec;
    unsigned long long alo, ahi;
    pack_bytes(lane < NSEG ? bestassign[lane] : -1, N, alo, ahi);
    int take = 0;
    if (lane == 0) {
      if (rec_lock(sa, R_)) {
        const int gt = rank_trial(ald(&R_->inc_rank));
        const double gc = bits_f64(ald(&R_->inc_cost));
        const unsigned long long gk = ald(&R_->inc_key);
        take = (trial < gt || (trial == gt && (cost < gc || (cost == gc && best_key < gk)))) ? 1 : 2;
      }
    }
    take = uniform_i32(take);
    if (take == 1) {
      if (lane < n) wt_store(&R_->x[lane], bestx_g[lane]);
      if (lane == 0) {
        ast(&R_->assign_lo, alo); ast(&R_->assign_hi, ahi);
        ast(&R_->inc_key, best_key); ast(&R_->inc_cost, f64_bits(cost));
        ast(reinterpret_cast<unsigned long long*>(&R_->inc_f), tb_lane0_64(TB_F));
        ast(reinterpret_cast<unsigned long long*>(&R_->inc_h), f64_bits(h));
      }
      drain_stores();
      if (lane == 0) ast(&R_->inc_rank, rank_pack(trial, cost));  // last: what the lock-free readers prune with
      drain_stores();
    }
    if (take && lane == 0) rec_unlock(R_);
  }

  // This worker has nothing more to do for the problem.  false: other parts of its tree are still being explored — whoever
  // finishes last writes the result.  true: this was the last part; nodes / iters / flops hold the totals of the problem.
  // limit: FH_ST_* limit status this worker ran into in its last search (0: none); last_trial: that search was the last factor.
  __device__ bool finish_part(const ShareArgs& sa, int& nodes, int& iters, unsigned limit, bool last_trial) {
    FH_SP_T0();
    ShareRec* R_ = sa.recs + rec;
    int last = 0;
    drain_stores();  // an incumbent this worker published is in place before its part counts as finished
    if (lane == 0) {
      aadd(&R_->nodes, nodes);
      aadd(&R_->iters, iters);
      aadd(&R_->flops, flops);
      if (limit == FH_ST_INTERRUPTED || (limit && last_trial)) __hip_atomic_fetch_max(&R_->last_status, limit, __ATOMIC_RELAXED, FH_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      last = __hip_atomic_fetch_sub(&R_->pending, 1, __ATOMIC_RELAXED, FH_AGENT) == 1 ? 1 : 0;
    }
    nodes = 0; iters = 0; flops = 0ull;
    last = uniform_i32(last);
    FH_SP_ADD(prof2, 4, 1);
    if (!last) return false;
    nodes = uniform_i32(ald(&R_->nodes));
    iters = uniform_i32(ald(&R_->iters));
    flops = uniform_u64(ald(&R_->flops));
    return true;
  }
  // a node / iteration limit in the current trial of a shared problem: the trial's tree is incomplete, its leaves cannot win
  __device__ void note_limited(const ShareArgs& sa, unsigned limit) {
    if (lane == 0) {
      __hip_atomic_fetch_or(&(sa.recs + rec)->limited, 1ull << (trial < 63 ? trial : 63), __ATOMIC_RELAXED, FH_AGENT);
      __hip_atomic_fetch_max(&(sa.recs + rec)->limit_kind, limit, __ATOMIC_RELAXED, FH_AGENT);  // which limit (ITER_LIMIT over NODE_LIMIT)
    }
  }


  // =================================================================================================================
  // The fused pair kernel: results leave through LDS, the safe problem never leaves the chip
  // =================================================================================================================
  // One face row of a problem into LDS as the solve reads it: a / |a| and -(b + feas_tol) / |a| (see `faces`).  ONE copy, used by the
  // staging of a problem record (run_problem) and by the on-chip hand-off of a fused pair (handoff_onchip): the same roundings.
  __device__ __forceinline__ void stage_face(fh_face fc, double feas_tol, int f, bool& degenerate_violated) {
#pragma clang fp contract(off)  // (two inline sites, one result: no contraction left to the optimiser's choice per site)
    const double nr = sqrt(fc.a[0] * fc.a[0] + fc.a[1] * fc.a[1] + fc.a[2] * fc.a[2]);
    if (nr > 0.0) {
      const double inv = 1.0 / nr;
      fc.a[0] *= inv; fc.a[1] *= inv; fc.a[2] *= inv;
      fc.b = -(fc.b + feas_tol) * inv;
      tolf[f] = (float)(feas_tol * inv);
    } else {  // 0 <= b: never binding if b >= -tol, else no point satisfies it
      degenerate_violated = -fc.b > feas_tol;
      fc.b = -1.0;
      tolf[f] = 0.0f;
    }
    faces[f] = fc;
  }
  // x0[j] in LDS (see compute_states).  Pc, Vc, Ac are consecutive arrays of 3 NT doubles (carve): ONE base pointer and an index — a
  // per-lane choice between three pointers would turn every array of the carve into a 64-bit generic pointer held in vector registers
  // The per-lane state of a problem (the registers below `per-lane state kept in registers`) is dead between two problems — said
  // explicitly, at the start of a problem and before the hand-off of a pair: the compiler cannot see it (the members are written by one
  // iteration of the kernel loop and, as far as it knows, read by the next), carried 24 registers across the ticket, the staging and
  // the hand-off, and spilled them there (measured: 29 -> 13 spilled registers in the plain N = 10 kernel).
  __device__ __forceinline__ void forget_lane_state() {
    p0r = v0r = a0r = xpr = xj = 0.0;
    cp_r[0] = cp_r[1] = cp_r[2] = 0.0;
    wbj = wbv = wba = wcp = 0.0;
    scan_f0 = scan_F = 0;
  }
  __device__ __forceinline__ double* x0_slot(int j) const { return Pc + (j < 3 ? j : (j < 6 ? 3 * NT + (j - 3) : 6 * NT + (j - 6))); }

  // ---- the result record, assembled in LDS (where Q was: the solve is over) and stored with ONE 16-byte-per-lane instruction per 1 KiB ----
  // Layout of the table T = Q: doubles 0..5 the head of fh_result (solved | trials, status | nodes, qp_iters | kflops, factor, dt, cost),
  // 6 .. 6 + 12 rows the coefficient rows (fh_result.coeff is contiguous behind the head), then the 16 bytes of fh_result.assign.  Chunk c
  // (16 bytes) goes to byte 16 c of the record, the last chunk to offsetof(assign).  rows: coefficient rows that are written — FH_MAX_SEG
  // (every word of the record, no memset needed) or, with fh_sched.compact_results, the segment count the kernel is built for (rows
  // beyond a problem's n_seg carry no information: 576 of 1600 bytes at N = 10).  The hand-off of a fused pair reads the coefficient
  // table from T + 6 (fin_solved, fin_dt) — not from memory.
  static constexpr int RES_HEAD = 6;
  __device__ __forceinline__ void emit_result(bool solved, int trials, int status, int nodes, int iters, double factor, double dt, double cost, int rows) {
    const int lane = opaque(this->lane);
    double* T = Q;
    FH_SYNC();
    for (int idx = lane; idx < 12 * rows; idx += 64) {
      const int t = idx / 12, rem = idx - 12 * t, kind = rem / 3, i = rem - 3 * kind;
      double v = 0.0;
      if (solved && t < N) {  // polynomial coefficients in the reference variable order (createVars :70-84)
        const int o = 3 * t + i;
        v = kind == 0 ? xs[o] / 6.0 : (kind == 1 ? Ac[o] / 2.0 : (kind == 2 ? Vc[o] : Pc[o]));
      }
      T[RES_HEAD + idx] = v;
    }
    if (lane < FH_MAX_SEG)
      reinterpret_cast<signed char*>(T + RES_HEAD + 12 * rows)[lane] = (solved && lane < N && P > 0) ? (signed char)bestassign[lane] : (signed char)-1;
    if (lane == 0) {
      const unsigned long long kf = flops / 1000ull;
      const int kfl = kf > 0x7fffffffull ? 0x7fffffff : (int)kf;
      int* Ti = reinterpret_cast<int*>(T);
      Ti[0] = solved ? 1 : 0; Ti[1] = trials; Ti[2] = status; Ti[3] = nodes; Ti[4] = iters; Ti[5] = kfl;
#ifdef FH_QMAX_STAT
      Ti[5] = qmax;
#endif
      T[3] = solved ? factor : 0.0;
      T[4] = dt;
      T[5] = solved ? cost : 0.0;
      tb[TB_WORK] += iters;
    }
    fin_solved = solved ? 1 : 0;
    fin_dt = dt;
    fin_rows = rows;
    FH_SYNC();
  }
  __device__ __forceinline__ void flush_result(fh_result& res) {
    const int lane = opaque(this->lane);
    const int rows = fin_rows;
    const int body = (RES_HEAD + 12 * rows) >> 1;  // 16-byte chunks of head + coefficient rows; one more for assign
    unsigned char* out = reinterpret_cast<unsigned char*>(&res);
    const double2* T2 = reinterpret_cast<const double2*>(Q);
    for (int c = lane; c <= body; c += 64) {
#ifdef FH_SHARE_PROFILE
      if (c == (RES_HEAD + (FH_MAX_SEG - 1) * 12 + 8) / 2) continue;  // (diagnostic: the owner's start time, written when the problem was begun)
#endif
#ifdef FH_TRACE
      if (!fin_solved && c >= RES_HEAD / 2 && c < body) continue;  // (diagnostic: the trace of an unsolved problem lives in its coefficient rows)
#endif
      const double2 v = T2[c];
      *reinterpret_cast<double2*>(out + (c < body ? (size_t)16 * c : offsetof(fh_result, assign))) = v;
    }
  }
  int fin_rows;  // coefficient rows in the table emit_result left behind
#ifdef FH_QMAX_STAT
  int qmax;      // (diagnostic build: the largest number of active rows of the problem in hand, reported in fh_result.kflops)
#endif

  // ---- the hand-off of a fused pair without a trip through memory -----------------------------------------------------
  // What pair_glue_one (fh_sample.hip.hpp) does for the staged pipeline — R from the whole trajectory (Faster::replan, faster.cpp:475), the
  // run of polytopes of the whole corridor from the one that holds R, shrunk — with the whole problem still staged in this wavefront's LDS
  // and its result table in T: the rows of the safe corridor are fetched ONCE (the whole problem's rows, L2-resident: this wavefront
  // staged them), tested against R, shifted, and normalised straight into `faces` — exactly what staging the written record would have
  // put there (stage_face, glue_face_b: one copy of each formula) —, x0 = R goes into the LDS slots the solve reads x0 from, and the
  // template's xf into xfl.  Nothing is written to memory: no write-through stores, no drain, no scalar-cache invalidate, no read-back
  // (round 5: 20 k cycles of hand-off + 14 k of staging per pair, 59 MB written and read per 32768-pair launch).  The record and the rows
  // of the safe problem exist in memory only if somebody needs them: always with fh_sched.pair_outputs (the tests compare them with the
  // staged pipeline), otherwise when the safe problem is first shared with another workgroup (begin_donation -> write_safe_problem).
  // returns 1: the safe problem is staged (faces, face_off, P, maxF, poly_ok, x0, xfl); 2: the pair has no safe problem.
  template <class PW, class PS>
  __device__ __forceinline__ int handoff_onchip(const PW& pw, const PS& ps, const fh_face* __restrict__ wfaces, double feas_tol, double r_frac, double shrink,
                                                int max_safe_poly, double r_margin, const fh_pair_rule& rule, const UnknownGrid* ug,
                                                unsigned long long* probe) {
    const int lane = opaque(this->lane);
    forget_lane_state();
    ProblemView pv;
    pv.n_seg = N; pv.n_poly = P; pv.face_begin = pw.face_begin; pv.dc = pw.dc; pv.a_max = pw.a_max;
    pv.x0[0] = uniform_f64(Pc[0]); pv.x0[1] = uniform_f64(Pc[1]); pv.x0[2] = uniform_f64(Pc[2]);
    pv.face_off = face_off;
    ResultView rv;
    rv.solved = fin_solved; rv.dt = fin_dt;
    rv.coeff = reinterpret_cast<const double (*)[12]>(Q + RES_HEAD);
    // the template's goal state: requested now, stored with the rows (lane = word; the record is read-only during the launch)
    typedef const __attribute__((address_space(4))) double cdouble;
    const double xfv = (reinterpret_cast<cdouble*>(&ps.xf[0]))[lane < 9 ? lane : 0];
    fh_state R;
    // ONE way through: a pair without a safe problem (no whole trajectory, or none is needed) runs on — R is the start of the trajectory
    // then, what is staged is never used — and says so through LDS at the end: an early return, or a corridor emptied by a flag in a
    // register, took the N = 10 pair kernel from 39 to 103 spilled registers (the allocation of this kernel is chaotic: DESIGN.md 4d).
    {
      const bool found = glue_find_r(pv, rv, r_frac, rule, lane, ug, R, probe);
      if (lane == 0) tb[TB_SAFE] = found ? 0 : 0x200;
    }
    // x0 of the safe problem = R, at once (nothing reads the whole problem's x0 any more): only R's position stays in registers
    if (lane < 9) {  // (selected, not indexed: an array indexed by the lane would live in scratch memory)
      const double rv9[9] = {R.pos[0], R.pos[1], R.pos[2], R.vel[0], R.vel[1], R.vel[2], R.accel[0], R.accel[1], R.accel[2]};
      double v = rv9[0];
#pragma unroll
      for (int j = 1; j < 9; j++) v = lane == j ? rv9[j] : v;
      *x0_slot(lane) = v;
      xfl[lane] = xfv;
    }
    const double Rp[3] = {R.pos[0], R.pos[1], R.pos[2]};
    const bool keep_r = r_margin >= 0.0;
    const double test_shrink = keep_r ? 0.0 : shrink;
    const double slack_ok = keep_r ? 1e-7 : 0.0;
    const int Pw = P, fb = pv.face_begin;
    const int nf = Pw ? uniform_i32(face_off[Pw]) : 0;
    double worst = -INFINITY;  // lane p: how far R is outside polytope p (its worst row)
    fh_face fc0 = {{0.0, 0.0, 0.0}, 0.0};  // rows 0..63 of the whole corridor stay in registers for the second pass
    for (int f0 = 0; f0 < nf; f0 += 64) {
      const int f = f0 + lane;
      double v = -INFINITY;
      int pf = -1;
      if (f < nf) {
        const fh_face fc = wfaces[fb + f];
        if (f0 == 0) fc0 = fc;
        v = glue_face_test(fc, glue_face_norm(fc), Rp, test_shrink);
        pf = 0;
        for (int p = 1; p < Pw; p++) pf += (f >= face_off[p]) ? 1 : 0;
      }
      for (int p = 0; p < Pw; p++) {
        const double m = glue_wave_max(pf == p ? v : -INFINITY);
        worst = lane == p ? fmax(worst, m) : worst;
      }
    }
    if (probe) probe[2] = __builtin_readcyclecounter();
    // glue_pick_start across the lanes: the first polytope that lets R in (worst <= slack), else the first of the least violated ones
    int start = first_lane(lane < Pw && worst <= slack_ok);
    if (start < 0) {
      const double least = wave_min(lane < Pw ? worst : INFINITY);
      start = first_lane(lane < Pw && worst == least);
      if (start < 0) start = 0;  // (no polytope at all, or NaN rows)
    }
    int cnt = Pw - start;
    if (cnt > max_safe_poly) cnt = max_safe_poly;
    if (Pw == 0) cnt = 0;
    const int src0 = cnt ? uniform_i32(face_off[start]) : 0;
    const int total = cnt ? uniform_i32(face_off[start + cnt]) - src0 : 0;
    // the safe problem's own face offsets: lane p holds entry p (entries beyond cnt repeat the total, as the staged hand-off writes them)
    const int noff = (cnt && lane <= FH_MAX_POLY) ? face_off[start + (lane < cnt ? lane : cnt)] - src0 : 0;
    const int first_end = __builtin_amdgcn_readlane(noff, 1);  // rows of the polytope that holds R
    FH_SYNC();
    unsigned badpoly = 0u;
    for (int f0 = 0; f0 < src0 + total; f0 += 64) {  // f: row of the whole corridor, j = f - src0: row of the safe corridor
      const int f = f0 + lane, j = f - src0;
      const bool on = j >= 0 && j < total;
      bool degenerate_violated = false;
      fh_face fc = fc0;
      if (f0 > 0 && on) fc = wfaces[fb + f];
      if (on) {
        fc.b = glue_face_b(fc, glue_face_norm(fc), Rp, shrink, keep_r, j < first_end, r_margin);
        stage_face(fc, feas_tol, j, degenerate_violated);
      }
      if (wave_any(degenerate_violated)) {
        int pf = 0;
        for (int p = 1; p < cnt; p++) pf += (j >= __builtin_amdgcn_readlane(noff, p)) ? 1 : 0;
        for (int p = 0; p < cnt; p++)
          if (wave_any(degenerate_violated && pf == p)) badpoly |= 1u << p;
      }
    }
    if (probe) probe[3] = __builtin_readcyclecounter();
    if (lane <= FH_MAX_POLY) face_off[lane] = noff;
    {
      int mf = 0;
      for (int p = 0; p < cnt; p++) mf = max(mf, __builtin_amdgcn_readlane(noff, p + 1) - __builtin_amdgcn_readlane(noff, p));
      maxF = mf;
    }
    P = cnt;
    poly_ok = ~badpoly;
    if (lane == 0) tb[TB_SAFE] |= start;
    FH_SYNC();
    return (uniform_i32(tb[TB_SAFE]) & 0x200) ? 2 : 1;
  }

  // ---- MIQP for one dt: depth-first branch and bound.  entry 0: from the root; entry 1: from the frame installed as
  // stack level 0 (install_frame).  returns FH_ST_* (OPTIMAL / INFEASIBLE refer to what THIS worker saw) ----
  // jerk-independent rows of the box: |v0| <= v_max, |a0| <= a_max (setMaxConstraints t = 0, :397-401)
  template <class PR>
  __device__ __forceinline__ bool x0_outside_box(const PR& pr) const {
    bool x0bad = false;
    for (int i = 0; i < 3; i++) x0bad |= (fabs(uniform_f64(Vc[i])) - vmax > tol) || (fabs(uniform_f64(Ac[i])) - amax > tol);  // x0 (LDS, see compute_states)
    return x0bad;
  }

  template <class PR>
  __device__ int search(const PR& pr, const fh_params& par, const ShareArgs& sa, double* __restrict__ ws, int entry,
                        double& best_cost, int& nodes, int& iters) {
    int depth = 0;
    int status_limit = 0;
    bool backtrack = false;
    bool carry_inf = false;   // the node the search is coming back from was infeasible,
    unsigned carry = 0u;      // with this conflict
    allinf = 0u;              // (a frame taken from the queue had children tried elsewhere: bit 0 stays clear)
    if (entry == 0) {
      FH_U0();
      best_cost = INFINITY;
      best_key = ~0ull;
      cur_key = 0ull;
      depth0 = 0;
      if (lane < NSEG) assign[lane] = -1;
      // jerk-independent rows of the box: |v0| <= v_max, |a0| <= a_max (setMaxConstraints t = 0, :397-401)
      if (x0_outside_box(pr)) return FH_ST_INFEASIBLE;
#ifdef FH_EARLY_IN_SEARCH  // (A/B: the same refutation, after the complete set-up of the trial)
      if (eq_ok && !(c0 <= box_ub) && par.max_nodes > 0 && !(par.max_work > 0 && iters >= par.max_work)) {
        nodes += 1;
        return FH_ST_INFEASIBLE;
      }
#endif
      {
        FH_T0();
        screen_constant_rows(pr);
        FH_T1(13);
      }
      if (P > 0) {
        if (allowed_mask(0) == 0u || allowed_mask(N - 1) == 0u) return FH_ST_INFEASIBLE;
        // fixed binaries (fh_problem.pin): pinned to an excluded polytope => no assignment is feasible
        const unsigned long long pins = (unsigned long long)pr.pin[0] | ((unsigned long long)pr.pin[1] << 32);
        bool pin_bad = false;
        for (int t = 0; t < N; t++) {
          const int v = (int)((pins >> (4 * t)) & 15ull);
          if (v && !((allowed_mask(t) >> (v - 1)) & 1u)) pin_bad = true;
        }
        if (pin_bad) return FH_ST_INFEASIBLE;
        FH_SYNC();
        if (lane < N) {
          const int v = (int)((pins >> (4 * lane)) & 15ull);
          if (v) assign[lane] = v - 1;
          else if (lane == 0 || lane == N - 1) {  // a segment with exactly one candidate polytope is not a decision
            const unsigned m = allowed_mask(lane);
            if ((m & (m - 1u)) == 0u) assign[lane] = __builtin_ctz(m);
          }
        }
        FH_SYNC();
      }
      reset_qp();  // y = 0: the minimum-norm point of the final-state equalities (setup_trial)
      qe = 0;
      if (lane == 0) tb[TB_QE] = 0;
      FH_U1(0);
      if (!eq_ok) return FH_ST_INFEASIBLE;
    } else {
      depth = 1;
      backtrack = true;
    }
    int local_nodes = 0;
    for (;;) {
      if (backtrack) {  // next untried sibling, deepest level first
        FH_U0();
        bool have_node = false;
        while (depth > 0) {
          const int d_ = depth - 1;
          const int seg = stk_seg[d_];
          if (carry_inf) {
            if (!((carry >> seg) & 1u)) {  // infeasible for reasons that do not involve this level's decision: so are the siblings,
              FH_SYNC();                   // and so is the parent, with the same conflict
              if (lane == 0) assign[seg] = -1;
              depth--;
              FH_SYNC();
              continue;
            }
            if (lane == 0) stk_mask[d_] |= (int)carry;
          } else {
            allinf &= ~(1u << d_);
          }
          carry_inf = false;
          const int nx = stk_next[d_];
          if (nx < stk_cnt[d_]) {
            if (sa.child_bound && best_cost < INFINITY) {  // the child's lower bound (written when the frame was made) against the incumbent: qp_loop's test, before the visit
              double lb;
              if (d_ < TC) {
                FH_SYNC();
                lb = uniform_f64(tbnd[d_ * FH_MAX_POLY + uniform_i32(nx)]);
              } else {
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                lb = uniform_f64(ws[(size_t)d_ * SNAP_PADDED + SNAP_BOUNDS + uniform_i32(nx)]);
              }
              const int shp = 3 * (15 - (depth0 + d_));
              const unsigned long long ckey = ((cur_key >> (shp + 3)) << (shp + 3)) | ((unsigned long long)(unsigned)uniform_i32(nx) << shp);
              const double ubp = best_cost * (1.0 - par.mip_gap);
              if (lb > ubp || (ckey > best_key && lb == ubp)) {
                FH_SYNC();
                if (lane == 0) stk_next[d_] = nx + 1;
                FH_SYNC();
                continue;  // (not infeasible: the next pass clears this level's all-infeasible bit)
              }
            }
            FH_SYNC();
            if (lane == 0) { stk_next[d_] = nx + 1; assign[seg] = stk_order[d_ * FH_MAX_POLY + nx]; }
            { FH_T0(); snapshot_restore(ws + (size_t)d_ * SNAP_PADDED, stk_q[d_], d_); FH_T1(11); }  // restart from the parent's optimum, not from scratch
            const int sh = 3 * (15 - (depth0 + d_));
            cur_key = ((cur_key >> (sh + 3)) << (sh + 3)) | ((unsigned long long)nx << sh);
            have_node = true;
            break;
          }
          FH_SYNC();
          if ((allinf >> d_) & 1u) {  // every child was infeasible: so is the parent, for the union of their reasons minus this decision
            carry_inf = true;
            carry = (unsigned)uniform_i32(stk_mask[d_]) & ~(1u << seg);
          }
          if (lane == 0) assign[seg] = -1;
          depth--;
          FH_SYNC();
        }
        FH_U1(1);
        if (!have_node) break;
      }
      backtrack = true;
#ifdef FH_PROFILE
      const unsigned long long u2__ = __builtin_readcyclecounter();
#endif
      if (local_nodes >= par.max_nodes) { status_limit = FH_ST_NODE_LIMIT; break; }
      if (par.max_work > 0 && iters >= par.max_work) { status_limit = FH_ST_ITER_LIMIT; break; }
      local_nodes++;
      // a problem that is already shared looks around twice as often; a taker looks before its first node (it hands the other
      // children of its frame, or the following trials, on at once if more takers are waiting)
      if ((local_nodes & (rec >= 0 ? (sa.look_mask >> 1) : sa.look_mask)) == 0 || (rec >= 0 && local_nodes == 1 && (entry == 1 || trial + 1 < trial_end))) {
        FH_T0();
        int fl = look_around(sa, local_nodes, iters);
        if (fl & 1) { status_limit = FH_ST_INTERRUPTED; break; }
#ifndef FH_NO_GIVE_TICKETS
        if (sa.enabled) give_tickets(sa);  // the tickets this workgroup holds behind the problem in hand: not hostages of a long problem
#endif
        // somebody is out of work: a problem that has proved hard (it already has a share record, or sa.min_nodes nodes so far)
        // gives its shallowest open frame away, and — once it is shared — the factor trials after this one, or a second frame.
        // (sa.enabled is 0 with a work cap or a MIP gap.)
        // (idle takers: nothing to lose by sharing early; no idle taker but room in the backlog: only the giants publish ahead)
        if ((fl & 2) && ((fl & 4) ? (rec >= 0 || nodes + local_nodes >= sa.min_nodes) : (nodes + local_nodes >= sa.giant_nodes || (fl & 8)))) {
          // the shallowest open frame first; then the factor trials after this one (a narrow tree may never have an open frame to
          // give, but its trials are independent), or, when it has none left to give, a second frame.  (ONE call site of donate():
          // a second one makes the compiler outline it, and a call puts the whole solver state into scratch memory.)
          for (int pass = 0; pass < 2; pass++) {
            if (pass == 1) {
              const int fl2 = look_around(sa, 1);
              if (!(fl2 & 2)) break;
              if (trial + 1 < trial_end) {
                donate_trials(sa, pr, best_cost, (fl2 >> 8) & 63);
                break;
              }
              if (rec < 0) break;
            }
            if (depth > 0) donate(sa, ws, depth, best_cost);
          }
        }
        FH_T1(16);
      }
      if (rec >= 0) {  // shared problem: other workers' leaves prune here too
        const unsigned long long r = uniform_u64(ald(&(sa.recs + rec)->inc_rank));
        const int rt = rank_trial(r);
        if (rt < trial) break;  // an EARLIER factor already has a feasible trajectory: nothing in this trial can win
        if (rt == trial) {
          const double gc = rank_cost_up(r);
          if (gc < best_cost) { best_cost = gc; best_key = ~0ull; }
        }
      }
      double cost = 0;
#ifdef FH_PROFILE
      const unsigned long long u3__ = __builtin_readcyclecounter();
      prof2[2] += u3__ - u2__;
#endif
      const int st = qp_run(best_cost * (1.0 - par.mip_gap), cur_key > best_key, par.max_iters, iters, cost);  // mip_gap 0 (default): exact
#ifdef FH_PROFILE
      prof2[3] += __builtin_readcyclecounter() - u3__;
#endif
      if (st == 3) { status_limit = FH_ST_ITER_LIMIT; break; }
      carry_inf = st == 1;
      carry = conflict;
#if defined(FH_TRACE) && FH_TRACE == 1
      if (lane == 0 && trace && local_nodes <= 32) {  // diagnostic builds: one record per node of an UNSOLVED single-trial problem
        double* tr = trace + 6 * (local_nodes - 1);
        tr[0] = (double)st + 10.0 * (double)depth; tr[1] = (double)(st == 1 ? conflict : 0u); tr[2] = (double)q + 100.0 * (double)iters;
        tr[3] = cost; tr[4] = (double)trace_src * 1e10 + (double)(unsigned)trace_id; tr[5] = trace_v;
      }
#endif
      if (st == 0) {
        int bseg;
        { FH_T0(); bseg = analyze(pr, entry == 0 && local_nodes == 1); FH_T1(9); }
        if (bseg < 0) {  // leaf: feasible for the MIQP.  The optimum is the lexicographic minimum of (cost, DFS key)
          FH_T0();
          if (cost < best_cost || (cost == best_cost && cur_key < best_key)) {
            best_cost = cost;
            best_key = cur_key;
            if (lane < n) bestx_g[lane] = x[lane];
            if (lane < N) bestassign[lane] = fullassign[lane];
            if (lane == 0) tb[TB_FRESH] = 1;  // Pc / Vc / Ac and the jerks xj are this leaf's until the next node is solved
            FH_SYNC();
            if (rec >= 0) publish_incumbent(sa, cost);
          }
          FH_T1(21);
        } else {  // branch on bseg, most promising polytope first (stable insertion sort)
          double lb_first = 0.0;  // lower bound of the first child (child bound)
          { FH_T0(); snapshot_save(ws + (size_t)depth * SNAP_PADDED, depth); FH_T1(10); }  // the children inherit this node's factorisation
          FH_T0();
          {  // child order: candidates sorted by how far the segment is outside each (ascending, ties by polytope index — the order
             // a stable insertion sort gives), by counting: lane p ranks polytope p among the candidates, no serial loop over LDS
            const unsigned am = allowed_mask(bseg) & (P ? ((1u << P) - 1u) : 0u);
            const bool cand = lane < FH_MAX_POLY && ((am >> lane) & 1u);
            const double key = viol[bseg * FH_MAX_POLY + (lane < FH_MAX_POLY ? lane : 0)];
            int rank = 0;
            for (int p2 = 0; p2 < P; p2++) {
              const double k2 = readlane_f64(key, p2);
              rank += (((am >> p2) & 1u) && (k2 < key || (k2 == key && p2 < lane))) ? 1 : 0;
            }
            if (cand) {
              stk_order[depth * FH_MAX_POLY + rank] = (signed char)lane;
              if (rank == 0) assign[bseg] = lane;
            }
            if (sa.child_bound) {
               // A lower bound of every child without visiting it.  The multipliers of this node's optimum y* together with ONE
               // multiplier on a row n of the child (violation v > 0 at y*) are dual feasible for the child's QP, and the best such
               // multiplier gives cost* + v^2 / |n|^2 (objective |y|^2): the largest of these over the rows of polytope p and the four
               // control points of the segment bounds child p from below.  |n| in the reduced space is 1 / wcp of lane (segment,
               // control point) for a normalised row — the scaled violation of the row scan, squared.
              const int sb = uniform_i32(bseg);
              const int lp = (lane >> 2) & 7, lk = lane & 3;
              const bool on = lane < 4 * FH_MAX_POLY && lp < P && ((am >> lp) & 1u);
              const int f0b = face_off[lp], Fb = on ? face_off[lp + 1] - f0b : 0;
              double cb[3];
#pragma unroll
              for (int i = 0; i < 3; i++) cb[i] = cp_of(lk, Pc[3 * sb + i], Vc[3 * sb + i], Ac[3 * sb + i], Pc[3 * sb + 3 + i]);
              double m = 0.0;
              const int flb = Fb > 0 ? Fb - 1 : 0;
              for (int f = 0; f < maxF; f++) {
                const fh_face fc = faces[f0b + min(f, flb)];
                const double vt = fma(fc.a[0], cb[0], fma(fc.a[1], cb[1], fma(fc.a[2], cb[2], fc.b)));
                m = (f < Fb && vt > m) ? vt : m;
              }
              double w = 0.0;
              double wcp = this->wcp;
              if constexpr (NORMS_TABLE) { double a_, b_, c_; row_norms(a_, b_, c_, wcp); }
#pragma unroll
              for (int k2 = 0; k2 < 4; k2++) {
                const double wk = readlane_f64(wcp, 4 * sb + k2);
                w = (lk == k2) ? wk : w;
              }
              double bnd = m * w;
              bnd = bnd * bnd;
              double bmax = 0.0;  // lane p < P: the bound of polytope p
              for (int p2 = 0; p2 < P; p2++) {
                double b4 = 0.0;
#pragma unroll
                for (int k2 = 0; k2 < 4; k2++) b4 = fmax(b4, readlane_f64(bnd, 4 * p2 + k2));
                bmax = (lane == p2) ? b4 : bmax;
              }
              // slack: 1e-9 of the increment (the rounding of the scaled violation) and 1e-12 of the bound itself — the leaf costs it is
              // compared with carry rounding noise relative to THEIR size, which the first term does not cover when bmax << cost
              const double lbv = (cost + bmax * (1.0 - 1e-9)) * (1.0 - 1e-12);
              if (cand) {
                if (depth < TC) tbnd[depth * FH_MAX_POLY + rank] = lbv;
                else ws[(size_t)depth * SNAP_PADDED + SNAP_BOUNDS + rank] = lbv;
              }
              const int l0 = first_lane(cand && rank == 0);
              lb_first = readlane_f64(lbv, l0 < 0 ? 0 : l0);
            }
            if (lane == 0) {
              stk_cnt[depth] = __builtin_popcount(am);
              stk_seg[depth] = bseg;
              stk_q[depth] = q;
              stk_next[depth] = 1;
              stk_mask[depth] = 0;
            }
          }
          allinf |= 1u << depth;
          depth++;
          FH_SYNC();
          FH_T1(20);
          backtrack = false;  // first child (rank 0: the key does not change): continue from the parent's factorisation
          if (sa.child_bound && best_cost < INFINITY) {  // ... unless its lower bound already loses against the incumbent
            const double ubp = best_cost * (1.0 - par.mip_gap);
            backtrack = lb_first > ubp || (cur_key > best_key && lb_first == ubp);
          }
        }
      }
    }
    nodes += local_nodes;
    if (status_limit) return status_limit;
    return (best_cost < INFINITY) ? FH_ST_OPTIMAL : FH_ST_INFEASIBLE;
  }
};

// What makes a record unusable, in two parts: the scalars (everything a solve reads from the record itself, also of the safe problem of a
// fused pair, whose corridor and x0 never exist as a record) and the corridor layout.  x0 and xf are tested where they are staged.
template <class PR>
__device__ inline bool bad_scalars(const PR& pr, int nseg_cap, int n_poly) {
  if (pr.n_seg < 1 || pr.n_seg > nseg_cap || n_poly < 0 || n_poly > FH_MAX_POLY) return true;
  if (!(pr.f_inc > 0) || !isfinite(pr.f_init) || !isfinite(pr.f_final)) return true;
  if ((pr.f_final - pr.f_init) / pr.f_inc > (double)FH_MAX_TRIALS) return true;
  if (!(pr.dc > 0) || !(pr.v_max > 0) || !(pr.a_max > 0) || !(pr.j_max > 0)) return true;
  const unsigned long long pins = (unsigned long long)pr.pin[0] | ((unsigned long long)pr.pin[1] << 32);
  for (int t = 0; t < FH_MAX_SEG; t++) {
    const int v = (int)((pins >> (4 * t)) & 15ull);
    if (v && (t >= pr.n_seg || v > n_poly)) return true;
  }
  return false;
}
template <class PR>
__device__ inline bool bad_corridor(const PR& pr, int face_cap) {
  if (pr.n_poly < 0 || pr.n_poly > FH_MAX_POLY) return true;
  if (pr.face_off[0] != 0 || pr.face_begin < 0) return true;
  for (int p = 0; p < pr.n_poly; p++) {
    const int c = pr.face_off[p + 1] - pr.face_off[p];
    if (c < 0 || c > FH_MAX_FACES_POLY) return true;
  }
  if (pr.n_poly && pr.face_off[pr.n_poly] > face_cap) return true;
  return false;
}

// coefficient rows of a result record that a kernel built for NSEG segments writes (Solver::emit_result; fh_sched.compact_results)
template <int NSEG>
__device__ __forceinline__ int result_rows(const ShareArgs& sa) {
#if defined(FH_PROFILE) || defined(FH_SHARE_PROFILE) || defined(FH_TRACE)
  return FH_MAX_SEG;  // (the diagnostic builds keep their numbers in the rows a problem does not use)
#else
  return sa.compact_results ? NSEG : FH_MAX_SEG;
#endif
}

// One problem (= one genNewTraj call), from the root of its first trial (entry 0) or from a frame of one of its trees taken
// from the queue (entry 1), until its final result is written (returns true) or until the tree this worker contributed to is
// still being explored elsewhere (returns false: the worker that finishes the last part continues the problem).
// staged: 0 the problem is staged from its record (x0, xf, the face rows); 1 the hand-off of a fused pair has staged it in LDS (the safe
// problem: handoff_onchip) — `pr` is then the caller's TEMPLATE of the safe problem, of which only the scalars are read; 2 the pair has no
// safe problem (its result says FH_ST_BAD_INPUT, as that of a record the staged hand-off marks with n_seg = 0).
// defer_store: the result is left in the LDS table (Solver::emit_result) and the caller stores it (Solver::flush_result) — the fused pair
// kernel runs the hand-off first, so that its loads are not queued behind the result's stores.
template <int NSEG, class SV, class PR>
__device__ __forceinline__ bool run_problem(SV& sv, const PR& pr, const fh_face* __restrict__ gfaces, int max_faces,
                            const fh_params& par, const ShareArgs& sa, const double* __restrict__ basis, double* __restrict__ ws, int entry,
                            bool interrupted, int staged, bool defer_store, fh_result& res) {
  const int lane = sv.lane;
  sv.fin_solved = 0;
  const int rows = result_rows<NSEG>(sa);  // coefficient rows of the record that are written (Solver::emit_result)
#ifdef FH_SHARE_PROFILE
  const unsigned long long sp_tp__ = wall_ticks();
#endif
#ifdef FH_PROFILE
  const unsigned long long tstart__ = pinned_clock();
  for (int i = 0; i < 24; i++) { sv.prof[i] = 0; sv.cnt[i] = 0; }
  for (int i = 0; i < 8; i++) sv.prof2[i] = 0;
  sv.prof2[8] = sv.pre_parts[0]; sv.prof2[9] = sv.pre_parts[1]; sv.prof2[10] = sv.pre_parts[2];
  sv.pre_parts[0] = sv.pre_parts[1] = sv.pre_parts[2] = 0;
  sv.prof[18] = sv.glue_cycles; sv.cnt[18] = sv.glue_cycles ? 1 : 0; sv.glue_cycles = 0;
  sv.prof[19] = sv.pre_cycles; sv.cnt[19] = sv.pre_cycles ? 1 : 0; sv.pre_cycles = 0;
  sv.prof[23] = sv.drain_cycles; sv.cnt[23] = sv.drain_cycles ? 1 : 0; sv.drain_cycles = 0;
  sv.prof[22] = sv.take_cycles; sv.cnt[22] = sv.take_calls; sv.take_cycles = 0; sv.take_calls = 0;
  const unsigned long long gp0__ = sv.glue_parts[0], gp1__ = sv.glue_parts[1], gp2__ = sv.glue_parts[2], gp3__ = sv.glue_parts[3];
  sv.glue_parts[0] = sv.glue_parts[1] = sv.glue_parts[2] = sv.glue_parts[3] = 0;
#endif
  sv.forget_lane_state();
#ifdef FH_QMAX_STAT
  sv.qmax = 0;
#endif
  FH_SYNC();  // the previous problem of this workgroup is completely done with LDS
  // x0 and xf: 18 consecutive doubles of the record, one load (lane = word), into the LDS slots the solve reads them from
  double x0xf = 0.0;
  if (staged == 0) {
    typedef const __attribute__((address_space(4))) double cdouble;
    static_assert(offsetof(fh_problem, xf) == offsetof(fh_problem, x0) + 9 * sizeof(double), "x0 and xf are adjacent");
    x0xf = (reinterpret_cast<cdouble*>(&pr.x0[0]))[lane < 18 ? lane : 0];
  }
  bool bad = interrupted;
  if (entry == 0 && !bad) bad = bad_scalars(pr, NSEG, staged ? sv.P : (int)pr.n_poly) || (staged == 0 && bad_corridor(pr, max_faces));
#ifndef FH_NO_EARLY_FACES
  // [r6] the first 64 rows of the corridor are requested NOW — the layout has just been checked — so that they travel together with x0
  // and xf: one memory round trip for the staging of a problem instead of two (both are first touches of their lines)
  fh_face fc_first = {{0.0, 0.0, 0.0}, 0.0};
  if (staged == 0 && !bad) {
    const int nf_ = pr.n_poly ? pr.face_off[pr.n_poly] : 0;
    if (lane < nf_) fc_first = gfaces[pr.face_begin + lane];
  }
#endif
  if (staged == 0) {
    if (lane < 9) *sv.x0_slot(lane) = x0xf;
    else if (lane < 18) sv.xfl[lane - 9] = x0xf;
    FH_SYNC();
  }
  if (entry == 0 && !bad) {
    const double v = lane < 9 ? *sv.x0_slot(lane) : sv.xfl[lane < 18 ? lane - 9 : 0];
    bad = wave_any(!isfinite(v));
  }
  if (entry == 0 && bad) {
    // (nothing of this problem is staged: an empty corridor, so that the hand-off of a pair — which runs on unconditionally and decides
    // at its end — walks no rows; LDS holds the previous problem's, or nothing at all for a workgroup's first unit)
    sv.N = 0; sv.flops = 0ull; sv.P = 0; sv.maxF = 0;
    if (lane <= FH_MAX_POLY) sv.face_off[lane] = 0;
    sv.emit_result(false, 0, interrupted ? FH_ST_INTERRUPTED : FH_ST_BAD_INPUT, 0, 0, 0.0, 0.0, 0.0, rows);
    if (!defer_store) sv.flush_result(res);
    return true;
  }

#ifdef FH_SHARE_PROFILE
  if (entry == 0 && lane == 0) {
    const unsigned long long t00 = ((unsigned long long)(unsigned)sv.tb[sv.TB_T0 + 1] << 32) | (unsigned)sv.tb[sv.TB_T0];
    res.coeff[FH_MAX_SEG - 1][8] = (double)(wall_ticks() - t00) / 100.0;
  }
#endif
#ifdef FH_TRACE
  sv.trace = &res.coeff[0][0];
  sv.trace_it = 0;
#endif
  sv.N = pr.n_seg;
  sv.nx = 3 * pr.n_seg;
  sv.zc0 = pr.force_final_pos ? 3 : 2;
  sv.K = max(pr.n_seg - sv.zc0, 0);
  sv.n = 3 * sv.K;
  if (staged == 0) sv.P = pr.n_poly;
  sv.tol = par.feas_tol;
  sv.dep2 = par.dep_tol * par.dep_tol;
  sv.vmax = pr.v_max; sv.amax = pr.a_max; sv.jmax = pr.j_max;
  sv.box_ub = (double)(3 * pr.n_seg) * pr.j_max * pr.j_max * (1.0 + 1e-9);
  sv.force_final = pr.force_final_pos;
  sv.init_problem();
#ifdef FH_PROFILE
  sv.prof2[4] = pinned_clock() - tstart__;
#endif
  // the orthogonal basis of this N (fh_basis.hip.hpp) stays in LDS from problem to problem: reloaded only when N changes
  const double* bt = reinterpret_cast<const double*>(sv.uniform_u64((unsigned long long)(basis + (size_t)(pr.n_seg - 1) * BT_STRIDE)));
  if (uniform_i32(sv.tb[sv.TB_ZN]) != pr.n_seg) {
    const int nn = pr.n_seg * pr.n_seg;
    for (int idx = lane; idx < nn; idx += 64) {
      const int s_ = idx / pr.n_seg;
      sv.Zm[s_ * sv.ZS + (idx - s_ * pr.n_seg)] = bt[BT_Z + idx];
    }
    if (lane == 0) sv.tb[sv.TB_ZN] = pr.n_seg;
  }
#ifdef FH_PROFILE
  sv.prof2[5] = pinned_clock() - tstart__ - sv.prof2[4];
#endif

  if (staged == 0) {  // stage the corridor once: coalesced 32-B face rows HBM -> LDS, normalised (stage_face)
    const int nf = pr.n_poly ? pr.face_off[pr.n_poly] : 0;
    if (lane <= FH_MAX_POLY) sv.face_off[lane] = pr.face_off[lane];
    {
      int mf = 0;
      for (int p = 0; p < pr.n_poly; p++) mf = max(mf, pr.face_off[p + 1] - pr.face_off[p]);
      sv.maxF = mf;
    }
    unsigned badpoly = 0u;
    for (int f0_ = 0; f0_ < nf; f0_ += 64) {
      const int f = f0_ + lane;
      bool degenerate_violated = false;
#ifndef FH_NO_EARLY_FACES
      if (f < nf) sv.stage_face(f0_ == 0 ? fc_first : gfaces[pr.face_begin + f], par.feas_tol, f, degenerate_violated);
#else
      if (f < nf) sv.stage_face(gfaces[pr.face_begin + f], par.feas_tol, f, degenerate_violated);
#endif
      if (wave_any(degenerate_violated)) {
        int pf = 0;
        for (int p = 0; p < pr.n_poly; p++) pf = (f >= pr.face_off[p]) ? p : pf;
        for (int p = 0; p < pr.n_poly; p++)
          if (wave_any(degenerate_violated && pf == p)) badpoly |= 1u << p;
      }
    }
    sv.poly_ok = ~badpoly;
  }
  FH_SYNC();

#ifdef FH_PROFILE
  sv.prof[0] = pinned_clock() - tstart__;
#endif
  const X0Lds x0l = {sv.Pc, 3 * sv.NT};
  const double dt0 = dt_initial(pr, x0l, lane);
  const double base = fmax(dt0, 2 * pr.dc);  // findDT :494-497
#ifdef FH_PROFILE
  sv.prof[14] = pinned_clock() - tstart__ - sv.prof[0]; sv.cnt[14] = 1;
#ifdef FH_PROFILE_ICACHE  // the same code again, now warm in the instruction cache: how much of the first call was instruction fetch?
  {
    const unsigned long long t2__ = __builtin_readcyclecounter();
    const double again = dt_initial(pr, x0l, opaque(lane));
    if (again != dt0) sv.cnt[22] += 1000;
    sv.prof[22] = __builtin_readcyclecounter() - t2__; sv.cnt[22] += 1;
  }
#endif
#endif
  // entry 0: a fresh problem, from its first factor.  entry 1: a stack frame of the tree of trial tb[TB_TRIALS] - 1 taken from the
  // queue.  entry 2: the trials of the problem from number tb[TB_TRIALS] - 1 (factor tb[TB_F]) on, taken from the queue.
  int trials = entry ? uniform_i32(sv.tb[sv.TB_TRIALS]) - 1 : 0, nodes = 0, iters = 0, status = FH_ST_INFEASIBLE;  // trials run before the current one
  bool solved = false;
  double f = entry ? sv.bits_f64(sv.tb_get64(sv.TB_F)) : pr.f_init, dt = 0, factor = 0, cost = 0;
  sv.rec = entry ? uniform_i32(sv.tb[sv.TB_REC]) : -1;
  sv.trial_end = entry == 0 ? 0x7fffffff : (entry == 2 ? uniform_i32(sv.tb[sv.TB_DEPTH0]) : 0);  // (entry 1: nothing beyond its trial)
  sv.flops = 0ull;
  if (lane == 0) sv.tb_put64(sv.TB_BASE, sv.f64_bits(base));
  unsigned limit = 0u;
  bool last_trial = false;
#ifdef FH_PROFILE
  const unsigned long long tloop__ = pinned_clock();
#endif
  for (;;) {  // genNewTraj :445-446: for (f = f_init; f <= f_final && !solved; f = f + f_inc)
    if (entry != 1) {
      if (!(f <= pr.f_final) || trials >= sv.trial_end) break;
      if (sv.rec >= 0 && rank_trial(sv.uniform_u64(ald(&(sa.recs + sv.rec)->inc_rank))) <= trials) break;  // an earlier factor is feasible already
      dt = f * base;
    } else {
      dt = sv.bits_f64(sv.tb_get64(sv.TB_H));
    }
    sv.trial = trials;
    trials++;
    last_trial = !(f + pr.f_inc <= pr.f_final);
    sv.h = dt;
    if (lane == 0) sv.tb_put64(sv.TB_F, sv.f64_bits(f));  // (what a frame given away by this trial has to say about it)
    bool early_inf;
    { FH_T0(); early_inf = sv.setup_trial(pr, bt, entry != 1);
#ifdef FH_PROFILE
      sv.prof[1] += __builtin_readcyclecounter() - t0__; sv.cnt[1] += 1;
#endif
    }
    double best = INFINITY;
    if (entry == 1) sv.install_frame(pr, sa, best);
#ifdef FH_SHARE_PROFILE
    const bool sp_taken__ = entry != 0;
    const unsigned long long sp_ts__ = wall_ticks();
    const int sp_n0__ = nodes;
    if (sp_taken__ && lane == 0) { aadd(&sa.ctl->prof2[0], sp_ts__ - sp_tp__); aadd(&sa.ctl->prof2[1], 1ull); }
#endif
#ifdef FH_PROFILE
    const unsigned long long ts0__ = __builtin_readcyclecounter();
#endif
    int st;
#ifndef FH_NO_EARLY_OUT
    // (the limits that search() tests before it opens a node keep their say: a node cap of zero, a work cap already used up)
    if (early_inf && !sv.x0_outside_box(pr) && par.max_nodes > 0 && !(par.max_work > 0 && iters >= par.max_work)) {  // the root node, refuted without an iteration
      st = FH_ST_INFEASIBLE;
      nodes += 1;  // (search() would count this root too — unless its screening leaves segment 0 or N - 1 without a polytope, or a pin is
                   //  excluded: then it answers INFEASIBLE before it opens a node, and `nodes` here is one larger than a search that screens
                   //  first would report.  The screening is what this exit saves; fh_result.nodes counts work, it is not a parity field there.)
      // a long window of refuted trials (up to FH_MAX_TRIALS) never enters search(), where the stop request and the deadline are polled:
      // every 8th trial looks at the launch's own words (the host's word is relayed there by whoever draws a unit or walks a tree)
      if ((trials & 7) == 0) {
        unsigned int stop = 0u;
        if (lane == 0) {
          const unsigned long long ei = ald(reinterpret_cast<unsigned long long*>(&sa.ctl->error));
          stop = (unsigned int)ei | (unsigned int)(ei >> 32);
          if (!stop && sa.deadline_ticks) {
            const unsigned long long t_start = ((unsigned long long)(unsigned)sv.tb[sv.TB_T0 + 1] << 32) | (unsigned)sv.tb[sv.TB_T0];
            if (wall_ticks() - t_start > sa.deadline_ticks) { stop = 2u; ast(&sa.ctl->interrupted, 2u); }
          }
        }
        if (uniform_i32((int)stop) != 0) st = FH_ST_INTERRUPTED;
      }
    } else  // (x0 outside the v / a box: search() says so before it touches what setup_trial skipped, and counts no node)
#endif
    st = sv.search(pr, par, sa, ws, entry == 1 ? 1 : 0, best, nodes, iters);
#ifdef FH_PROFILE
    sv.prof[15] += __builtin_readcyclecounter() - ts0__; sv.cnt[15] += 1;
#endif
#ifdef FH_SHARE_PROFILE
    if (sp_taken__ && lane == 0) { aadd(&sa.ctl->prof2[2], wall_ticks() - sp_ts__); aadd(&sa.ctl->prof2[3], (unsigned long long)(nodes - sp_n0__)); }
#endif
    entry = 0;
    limit = (st == FH_ST_NODE_LIMIT || st == FH_ST_ITER_LIMIT || st == FH_ST_INTERRUPTED) ? (unsigned)st : 0u;
    if (sv.rec < 0) {  // nothing of this problem was given away: the sequential rule, in registers
      status = limit ? (int)limit : (best < INFINITY ? FH_ST_OPTIMAL : FH_ST_INFEASIBLE);
      if (status == FH_ST_OPTIMAL) {
        solved = true;
        factor = f;
        cost = best;
        break;
      }
      if (status == FH_ST_INTERRUPTED || (par.max_work > 0 && status == FH_ST_ITER_LIMIT)) break;
      f = f + pr.f_inc;
      continue;
    }
    // shared problem: its leaves are in the record.  This worker goes on with the next factor if the following trials are still
    // its own and no factor up to this one has a feasible leaf (other parts of this trial may still be running elsewhere: the
    // next trial then starts speculatively, like the ones that were given away).
    if (limit && limit != FH_ST_INTERRUPTED) sv.note_limited(sa, limit);
    if (trials >= sv.trial_end || limit == FH_ST_INTERRUPTED || best < INFINITY) break;
    f = f + pr.f_inc;
  }
#ifdef FH_PROFILE
  sv.prof2[6] = pinned_clock() - tloop__;  // (the trial loop in total: set-up and search are slots 1 and 15)
  const unsigned long long tpost__ = pinned_clock();
#endif
  if (sv.rec >= 0) {
    if (!sv.finish_part(sa, nodes, iters, limit, last_trial)) return false;  // the problem's tree is still being explored elsewhere
    // the last part: the answer is the record's incumbent (first feasible factor, cheapest leaf, first in depth-first order)
    const ShareRec* R_ = sa.recs + sv.rec;
    const unsigned long long r = sv.uniform_u64(ald(&R_->inc_rank));
    const unsigned long long limited = sv.uniform_u64(ald(&R_->limited));
    const unsigned int last_status = (unsigned int)uniform_i32((int)ald(&R_->last_status));
    const int wt = rank_trial(r);
    solved = r != FH_RANK_NONE && last_status != FH_ST_INTERRUPTED && !((limited >> (wt < 63 ? wt : 63)) & 1ull);
    if (solved) {
      status = FH_ST_OPTIMAL;
      trials = wt + 1;
      cost = uniform_f64(sv.bits_f64(ald(&R_->inc_cost)));
      factor = uniform_f64(sv.bits_f64(ald(reinterpret_cast<const unsigned long long*>(&R_->inc_f))));
      dt = uniform_f64(sv.bits_f64(ald(reinterpret_cast<const unsigned long long*>(&R_->inc_h))));
      const unsigned long long alo = sv.uniform_u64(ald(&R_->assign_lo)), ahi = sv.uniform_u64(ald(&R_->assign_hi));
      if (lane < sv.n) sv.bestx_g[lane] = cc_load(&R_->x[lane]);
      if (lane < NSEG) sv.bestassign[lane] = sv.unpack_byte(alo, ahi, lane);
      sv.h = dt;
      (void)sv.setup_trial(pr, bt);  // the y = 0 states of the winning step (this worker may have been exploring another trial)
    } else {  // no factor of the window is feasible (or the search was cut short): trials_ and dt_ of the last trial of the window
      // (a limit in the winning trial disqualifies its leaves — the sequential search would have gone on to the next factor, which
      // cannot be reconstructed here: reported as not solved.  Otherwise the status of the last trial, as the sequential loop leaves it.)
      const unsigned int limit_kind = (unsigned int)uniform_i32((int)ald(&R_->limit_kind));
      status = (r != FH_RANK_NONE && last_status != FH_ST_INTERRUPTED) ? (int)(limit_kind ? limit_kind : (unsigned)FH_ST_NODE_LIMIT)
                                                                       : (last_status ? (int)last_status : FH_ST_INFEASIBLE);
      trials = 0;
      for (double fk = pr.f_init; fk <= pr.f_final; fk = fk + pr.f_inc) {
        trials++;
        dt = fk * base;
      }
    }
  }

#ifdef FH_PROFILE
  const unsigned long long tres__ = pinned_clock();
  sv.prof2[7] = tres__ - tpost__;
#endif
  if (solved) {  // the jerks and the states of the optimum: what the coefficient rows are made of (emit_result)
    FH_SYNC();
    // [r6] ... which are still in LDS and registers when no node was solved after the incumbent leaf (the siblings behind it were
    // pruned by their bounds: half of the solved problems) — the same numbers compute_states would produce again from the same y
    // (Off by default: with the branch around compute_states the allocator of <10, true, 3> spills 49 vector registers instead of 19 — 112
    // instead of 80 B of scratch per lane, 0.35 instead of 0.29 GB of HBM traffic per launch, rocprofv3 counters — for +0.5 % of throughput.)
#ifdef FH_FRESH_STATES
    const bool fresh = sv.rec < 0 && uniform_i32(sv.tb[sv.TB_FRESH]) != 0;
#else
    const bool fresh = false;
#endif
    if (!fresh) {
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (the incumbent was stored by this wavefront: its stores are drained before they are read back)
      if (lane < sv.NVP) sv.x[lane] = (lane < sv.n) ? sv.bestx_g[lane] : 0.0;
      FH_SYNC();
      sv.compute_states();
    }
    FH_SYNC();
    if (lane < sv.NXP) sv.xs[lane] = (lane < sv.nx) ? sv.xj : 0.0;  // the jerks xp + Z y
    FH_SYNC();
  }
  // every word of the result (of its first `rows` coefficient rows with fh_sched.compact_results) is written by the kernel: no memset of
  // the result buffer is needed
  sv.emit_result(solved, trials, status, nodes, iters, factor, dt, cost, rows);
#ifdef FH_PROFILE
  sv.prof[17] = pinned_clock() - tres__; sv.cnt[17] = 1;
  sv.prof[12] = pinned_clock() - tstart__;
  if (lane < 12 && sv.N <= FH_MAX_SEG - 4) {  // rows 15 / 14: cycles / calls of slots 0..11, rows 13 / 12: of slots 12..23
    unsigned long long pv = 0, pw = 0;
    unsigned int cv = 0, cw = 0;
    for (int i = 0; i < 12; i++) { pv = (i == lane) ? sv.prof[i] : pv; cv = (i == lane) ? sv.cnt[i] : cv; }
    for (int i = 0; i < 12; i++) { pw = (i == lane) ? sv.prof[12 + i] : pw; cw = (i == lane) ? sv.cnt[12 + i] : cw; }
    double* Tc = sv.Q + sv.RES_HEAD;  // (the coefficient rows of the table: [FH_MAX_SEG][12])
    Tc[(FH_MAX_SEG - 1) * 12 + lane] = (double)pv;
    Tc[(FH_MAX_SEG - 2) * 12 + lane] = (double)cv;
    Tc[(FH_MAX_SEG - 3) * 12 + lane] = (double)pw;
    Tc[(FH_MAX_SEG - 4) * 12 + lane] = (double)cw;
    if (lane < 4) Tc[(FH_MAX_SEG - 5) * 12 + lane] = (double)(lane == 0 ? gp0__ : (lane == 1 ? gp1__ : (lane == 2 ? gp2__ : gp3__)));
    if (sv.N <= FH_MAX_SEG - 6) {
      unsigned long long pu = 0;
      for (int i = 0; i < 12; i++) pu = (i == lane) ? sv.prof2[i] : pu;
      Tc[(FH_MAX_SEG - 6) * 12 + lane] = (double)pu;
    }
  }
  FH_SYNC();
#endif
  if (!defer_store) sv.flush_result(res);
#ifdef FH_SHARE_PROFILE
  if (lane == 0 && sv.N < FH_MAX_SEG) {  // diagnostic: when (us since this workgroup started) the problem began here / ended, shared?
    const unsigned long long t00 = ((unsigned long long)(unsigned)sv.tb[sv.TB_T0 + 1] << 32) | (unsigned)sv.tb[sv.TB_T0];
    res.coeff[FH_MAX_SEG - 1][11] = (double)(wall_ticks() - t00) / 100.0;
    res.coeff[FH_MAX_SEG - 1][10] = (double)(sp_tp__ - t00) / 100.0;
    res.coeff[FH_MAX_SEG - 1][9] = sv.rec >= 0 ? 1.0 : 0.0;
  }
#endif
  return true;
}

struct SolveArgs {  // (the problem / face / result arrays are separate `__restrict__` kernel parameters: scalar loads)
  int n, max_faces;
  fh_params par;
  double* workspace;
  const double* basis;  // [FH_MAX_SEG][BT_STRIDE] reduced-space basis tables (fh_basis.hip.hpp), read-only
  ShareArgs sa;
  // pair launches only (solve -> hand-off -> safe solve)
  fh_problem* safe;
  fh_face* sfaces;
  fh_result* sres;
  double r_frac, shrink, r_margin;
  int max_safe_poly, pad;
  fh_pair_rule rule;  // which sample of the whole trajectory becomes R (fh_set_pair_rule)
  UnknownGrid unknown; // rule mode 2: the caller's unknown voxels (fh_set_unknown_grid_device), read by the UNK instantiations only
  // launch order: ticket t works on unit order[t] (null: t).  Results do not depend on it; the hardest corridors go first so that
  // their trees are not what the launch ends on (order_kernel)
  const int* order;
};

// Persistent workgroups (one wavefront each).  Every workgroup pulls fresh units from a device-scope ticket counter
// until the batch is exhausted, so problems of very different difficulty balance across the 256 CUs and the snapshot
// workspace is sized by the resident grid, not by the batch; then it takes over frames of the trees that are still
// being explored (fh_share.hip.hpp) until every unit is done.
//
// PAIRS: a unit is a whole+safe pair, whole solve -> hand-off -> safe solve back to back: the data dependency of
// Faster::replan (faster.cpp:427 whole genNewTraj, :475 R = X_whole[k], :521-536 safe genNewTraj) is per pair, so nothing waits
// for the stragglers of a batch-wide whole launch before the safe solves start.  The safe problem record and its face rows are
// written and read back through L2 inside the launch (possibly by another workgroup): agent-scope fences order the two.
// WPS: wavefronts per SIMD the kernel is compiled for.  3 (168 registers, some spilled: 11 resident solves per CU at N = 10) is the
// throughput build; 2 (193 registers at N = 10, nothing spilled, no scratch: 8 per CU) solves a batch that is alone on the device
// 13 % sooner and many batches in flight 14 % slower — fh_sched.workgroups_per_cu <= 8 selects it.  The same arithmetic: the same bits.
// UNK: the hand-off of a pair asks the caller's unknown voxels (fh_pair_rule mode 2: findIndexH as the reference decides it,
// faster.cpp:218-251 with kdtree_unk_.nearestKSearch :236).  An instantiation of its own, built for two wavefronts per SIMD, so that
// the lookup's registers are not the throughput build's problem; modes 0 and 1 run the same kernels as before.
template <int NSEG, bool PAIRS, int WPS = FH_WAVES_PER_SIMD, bool UNK = false>
__global__ void __launch_bounds__(64, WPS) solve_kernel(const fh_problem* __restrict__ problems, const fh_face* __restrict__ faces,
                                                   fh_result* __restrict__ results, SolveArgs ka) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [r6] The row norms stay in registers in every build (NORMS_TABLE = false).  Round 5 re-read them from the basis table in the
  // three-wavefront build to save eight registers; with x0 in LDS and the safe problem built on chip the allocation is a different one:
  // the table variant has 41 spilled registers in <10, true, 3>, this one 19, and it is 1 % faster (A/B, same box: 23.15 against 22.9 M
  // pairs/s, one launch alone 2.77 against 2.9 ms) — the loads the table variant issues in every scan wait behind every store in flight.
#ifndef FH_NORMS_FROM_TABLE
#define FH_NORMS_FROM_TABLE false
#endif
  typedef Solver<NSEG, FH_NORMS_FROM_TABLE> SolverT;
  SolverT sv;
  sv.carve(smem, ka.max_faces);
  sv.lane = threadIdx.x;
  sv.q = 0;
  sv.qe = 0;
  if (threadIdx.x == 0) { sv.tb[sv.TB_ZN] = 0; sv.tb_put64(sv.TB_NEXT, 0ull); sv.tb_put64(sv.TB_POOL2, 0ull); sv.tb_put64(sv.TB_NEXT_EI, 0ull); sv.tb_put64(sv.TB_NEXT_WT, 0ull); sv.tb[sv.TB_DONE_N] = 0; sv.tb[sv.TB_DONE_IT] = 0; }
  const ShareArgs& sa = ka.sa;
  double* ws = ka.workspace + (size_t)blockIdx.x * ((size_t)NSEG * (size_t)Solver<NSEG>::SNAP_PADDED + 64);
  sv.bestx_g = ws + (size_t)NSEG * (size_t)Solver<NSEG>::SNAP_PADDED;
  if (threadIdx.x == 0) {
    sv.tb_put64(sv.TB_T0, wall_ticks());
    aadd(&sa.ctl->started, 1u);
  }
  bool tickets_left = true;
#ifdef FH_SHARE_PROFILE
  unsigned long long sp_dry__ = 0;
#endif
#ifdef FH_PROFILE
  sv.pre_cycles = 0; sv.glue_cycles = 0; sv.drain_cycles = 0; sv.take_cycles = 0; sv.take_calls = 0;
  sv.glue_parts[0] = sv.glue_parts[1] = sv.glue_parts[2] = sv.glue_parts[3] = 0;
  sv.pre_parts[0] = sv.pre_parts[1] = sv.pre_parts[2] = 0;
  for (int i = 0; i < 12; i++) sv.prof2[i] = 0;
#endif
  for (;;) {
    int entry = 0, unit = 0, phase = 0;
    bool interrupted = false;
    sv.forget_lane_state();  // (nothing of the unit before is carried through the ticket phase)
#ifdef FH_PROFILE
    const unsigned long long tpre__ = pinned_clock();
#endif
    // Tickets come from the workgroup's own pool [TB_NEXT, TB_NEXT + 1): chunks of FH_TICKET_CHUNK drawn with ONE atomic, and — in a pair
    // launch — drawn AHEAD, during the hand-off of the pair in hand, together with the control words of the block, so that those
    // memory round trips overlap the hand-off's own instead of standing between two pairs.  The control words are up to a chunk
    // old when they are used: a pending frame that is gone fails its compare-and-swap in take_task, a stop request is seen a few
    // units later (the trees poll it themselves).
    int pool_next = uniform_i32(sv.tb[sv.TB_NEXT]), pool_end = uniform_i32(sv.tb[sv.TB_NEXT + 1]);
    bool frame_pending = true, fresh_words = false;
    if (pool_next >= pool_end && pool_end > 0) {  // the pool is empty: tickets another workgroup gave away?
      const int p2a = uniform_i32(sv.tb[sv.TB_POOL2]), p2b = uniform_i32(sv.tb[sv.TB_POOL2 + 1]);
      if (p2a < p2b) {
        pool_next = p2a; pool_end = p2b;
        FH_SYNC();
        if (threadIdx.x == 0) { sv.tb[sv.TB_NEXT] = p2a; sv.tb[sv.TB_NEXT + 1] = p2b; sv.tb_put64(sv.TB_POOL2, 0ull); }
        FH_SYNC();
      }
    }
    if (tickets_left && pool_next < pool_end) {
      const unsigned long long wt = sv.tb_get64(sv.TB_NEXT_WT);
      frame_pending = (int)((unsigned int)(wt >> 32) - (unsigned int)wt) > 0 &&
                      uniform_i32(sv.tb[sv.TB_POOL2]) >= uniform_i32(sv.tb[sv.TB_POOL2 + 1]);  // (one range of given tickets at a time)
    }
    // a frame of a hard problem comes before a fresh problem; without fresh problems the workgroup waits for frames
    if (sa.enabled && (sa.backlog > 0 || !tickets_left) && (frame_pending || !tickets_left)) {
      if (!tickets_left) sv.flush_done(sa);  // (units this workgroup has finished but not yet reported: the waiting ends when all are)
#ifdef FH_PROFILE
      const unsigned long long ttake__ = pinned_clock();
#endif
      entry = sv.take_task(sa, ws, !tickets_left);
      // (the control words are a chunk of tickets old: ONE attempt per reading — the units that follow in the chunk do not ask again
      // for a frame that was pending back then; 98 % of such attempts found it gone and cost a memory round trip each)
      if (tickets_left && threadIdx.x == 0) sv.tb_put64(sv.TB_NEXT_WT, 0ull);
      if (entry == 3) {  // a range of fresh tickets (give_tickets): they are in the pool, or behind it
        tickets_left = true;
        continue;
      }
#ifdef FH_PROFILE
      if (tickets_left) { sv.take_cycles += pinned_clock() - ttake__; sv.take_calls += 1; }
#endif
    }
#ifdef FH_PROFILE
    const unsigned long long tpre1__ = pinned_clock();
    unsigned long long tpre2__ = tpre1__;
#endif
    if (entry) {
      unit = uniform_i32(sv.tb[sv.TB_B]);
      phase = uniform_i32(sv.tb[sv.TB_PHASE]);
#ifdef FH_DEBUG_BOUNDS
      if ((unsigned)unit >= (unsigned)ka.n || (unsigned)phase > 1u || entry > 2) { if (threadIdx.x == 0) ast(&sa.ctl->error, 103u + (unsigned)entry); break; }
#endif
    } else if (!tickets_left) {
      break;  // every unit is done (or enough others are waiting, or sharing is off)
    } else {
      unsigned int intr = 0;
      if (pool_next >= pool_end) {  // the pool is empty: draw a chunk now
        unsigned long long b0 = 0ull, ei = 0ull, wt = 0ull;
        // The FIRST chunk of every workgroup is dealt, not drawn: workgroup w starts with the tickets [w c0, (w + 1) c0) and the counter
        // hands out what follows them (static_tickets).  All workgroups of a launch that has the device to itself start together, and
        // their first draws were 2816 read-modify-writes of ONE word: the phase profile showed a first unit waiting 200 k cycles (85 us)
        // for its ticket where every later one waits 600 — 8 % of the units of a C4 launch, 18 % of the time of one launch alone.
        // A dealt chunk is CLAIMED (claims[w]: one compare-and-swap on a word of its own) by its workgroup when it starts — or, once
        // the counter has run dry, by any workgroup that finds it unclaimed (steal_chunk): with other launches in flight the
        // workgroups of a launch start one by one as wavefront slots become free, and tickets dealt to a workgroup that is not
        // resident yet must not wait for it (measured without the claim: 12 launches in flight 417 ms per step instead of 1.4).
        const bool deal = sa.claims != nullptr;
        const int c0 = sv.static_chunk(ka.n, (int)gridDim.x), dealt_total = deal ? sv.static_tickets(ka.n, (int)gridDim.x) : 0;
        int claimed = -1;
        if (deal && pool_end == 0) {  // (a pool that was ever filled ends at a positive ticket: this is the workgroup's first draw)
          unsigned int expect = 0u;
          if (threadIdx.x == 0 && __hip_atomic_compare_exchange_strong(&sa.claims[blockIdx.x], &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, FH_AGENT))
            claimed = (int)blockIdx.x;
          claimed = uniform_i32(claimed);
        }
        int ch = sv.ticket_chunk(ka.n, (int)gridDim.x, pool_end);
        if (claimed < 0) {
          if (threadIdx.x == 0) b0 = aadd(&sa.ctl->ticket, (unsigned long long)ch) + (unsigned long long)dealt_total;
          b0 = sv.uniform_u64(b0);
          if (deal && b0 >= (unsigned long long)ka.n) claimed = SolverT::steal_chunk(sa.ctl, sa.claims, (int)gridDim.x, (int)threadIdx.x);  // the counter is dry: a chunk whose workgroup has not started?
        }
        if (claimed >= 0) {
          b0 = (unsigned long long)min(claimed * c0, dealt_total);
          ch = min((claimed + 1) * c0, dealt_total) - (int)b0;
        }
        if (threadIdx.x == 0) {
          ei = ald(reinterpret_cast<unsigned long long*>(&sa.ctl->error));
          wt = ald(reinterpret_cast<unsigned long long*>(&sa.ctl->wait_ticket));
          sv.tb_put64(sv.TB_NEXT_EI, ei);
          sv.tb_put64(sv.TB_NEXT_WT, wt);
        }
        b0 = sv.uniform_u64(b0);
        pool_next = (int)min(b0, (unsigned long long)ka.n);
        pool_end = (int)min(b0 + (unsigned long long)ch, (unsigned long long)ka.n);
        fresh_words = true;
      }
      unsigned int b = (unsigned int)pool_next;
      if (threadIdx.x == 0) {
        sv.tb[sv.TB_NEXT] = pool_next + (pool_next < pool_end ? 1 : 0);
        sv.tb[sv.TB_NEXT + 1] = pool_end;
        const unsigned long long ei = ((unsigned long long)(unsigned)sv.tb[sv.TB_NEXT_EI + 1] << 32) | (unsigned)sv.tb[sv.TB_NEXT_EI];
        intr = (unsigned int)ei | (unsigned int)(ei >> 32);
        if (!intr && pool_next < pool_end && (fresh_words || (b & (FH_TICKET_CHUNK - 1)) == 0u)) {  // the host's stop word costs a PCIe read: one workgroup in 32 polls it, once per chunk of tickets
          const unsigned long long t_start = ((unsigned long long)(unsigned)sv.tb[sv.TB_T0 + 1] << 32) | (unsigned)sv.tb[sv.TB_T0];
          if (sa.deadline_ticks && wall_ticks() - t_start > sa.deadline_ticks) intr = 2u;
          else if (sa.host_abort && (blockIdx.x & 31u) == 0u && __hip_atomic_load(sa.host_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) intr = 1u;
          if (intr) ast(&sa.ctl->interrupted, intr);
        }
      }
      interrupted = __builtin_amdgcn_readfirstlane((int)intr) != 0;
#ifdef FH_PROFILE
      tpre2__ = pinned_clock();
#endif
      if (pool_next >= pool_end) {  // (the chunk lay beyond the batch)
        tickets_left = false;
#ifdef FH_SHARE_PROFILE
        sp_dry__ = wall_ticks();
#endif
        continue;
      } else {
        // (the launch order was written by order_scatter_kernel before this launch began: a scalar load through the constant address space)
        typedef const __attribute__((address_space(4))) int cint_t;
#ifdef FH_DEBUG_BOUNDS
        if (b >= (unsigned)ka.n) { if (threadIdx.x == 0) ast(&sa.ctl->error, 101u); tickets_left = false; continue; }
#endif
        unit = ka.order ? *(cint_t*)sv.uniform_u64((unsigned long long)(ka.order + b)) : (int)b;
#ifdef FH_DEBUG_BOUNDS
        if ((unsigned)unit >= (unsigned)ka.n) { if (threadIdx.x == 0) ast(&sa.ctl->error, 102u); tickets_left = false; continue; }
#endif
        if (threadIdx.x == 0) { sv.tb[sv.TB_B] = unit; sv.tb[sv.TB_PHASE] = 0; sv.tb[sv.TB_WORK] = 0; }
      }
    }
    // (wave-uniform by construction; said explicitly because the divergence analysis loses it across this loop nest and would keep
    // these — and every address derived from them — in vector registers)
    entry = uniform_i32(entry);
    unit = uniform_i32(unit);
    phase = uniform_i32(phase);
    interrupted = uniform_i32(interrupted ? 1 : 0) != 0;
#ifdef FH_PROFILE
    if (!entry) {
      asm volatile("" :: "s"(unit) : "memory");  // (the order word has arrived)
      const unsigned long long tpre3__ = pinned_clock();
      sv.pre_cycles = tpre3__ - tpre__;
      sv.pre_parts[0] = tpre1__ - tpre__; sv.pre_parts[1] = tpre2__ - tpre1__; sv.pre_parts[2] = tpre3__ - tpre2__;
    }
#endif
    int staged = 0;  // (a pair: 1 the safe problem was staged in LDS by the hand-off, 2 the pair has none; see run_problem)
    for (;;) {  // the problems of the unit (a pair has two)
      bool finished;
      if constexpr (PAIRS) {
        // The problem record of a pair launch may have been written inside this launch (the safe problem of a pair that is shared between
        // workgroups: write_safe_problem, write-through) — only a workgroup that TAKES a frame of it reads what was written: it invalidates
        // the scalar data cache, after which the record is read like the read-only records of a plain launch — uniform scalar loads through
        // the constant address space (a generic pointer would keep the address and every field in vector registers: 130 spilled VGPRs).
        // A record does not change while it is being solved.
        if (entry && phase) __builtin_amdgcn_s_dcache_inv();
        typedef const __attribute__((address_space(4))) fh_problem const_problem;
        const unsigned long long pr_addr = sv.uniform_u64((unsigned long long)(phase ? &ka.safe[unit] : &problems[unit]));
        const fh_face* fcs = reinterpret_cast<const fh_face*>(sv.uniform_u64((unsigned long long)(phase ? ka.sfaces : faces)));
        fh_result* out = reinterpret_cast<fh_result*>(sv.uniform_u64((unsigned long long)(phase ? &ka.sres[unit] : &results[unit])));
        // (the whole problem's result stays in LDS until the hand-off has issued its loads: FH_DEFER_RESULT)
        finished = run_problem<NSEG, SolverT, const_problem>(sv, *(const_problem*)pr_addr, fcs, ka.max_faces, ka.par, sa, ka.basis, ws, entry, interrupted,
                                                             entry ? 0 : staged, FH_DEFER_RESULT && phase == 0, *out);
      }
      else {
        // Nothing writes the problem records during a plain solve launch: reading them through the constant address space keeps
        // the uniform loads on the scalar unit (s_load) although the kernel contains fences and atomics, after which the
        // compiler no longer treats `const __restrict__` global memory as unclobbered (132 vector loads instead of 28 scalar ones).
        typedef const __attribute__((address_space(4))) fh_problem const_problem;
        const unsigned long long pr_addr = sv.uniform_u64((unsigned long long)(problems + unit));  // (provably wave-uniform)
        finished = run_problem<NSEG, SolverT, const_problem>(sv, *(const_problem*)pr_addr, faces, ka.max_faces, ka.par, sa, ka.basis, ws, entry, interrupted,
                                                    0, false, results[unit]);
      }
      if (!finished) break;  // the unit continues in another workgroup
      if constexpr (PAIRS) {
        if (phase == 0) {
          // The hand-off reads what this wavefront still holds — the whole problem staged in LDS, the result table emit_result left where Q
          // was, the scalars of the problem record through the scalar cache — and leaves the safe problem staged in the same LDS
          // (Solver::handoff_onchip): nothing of it goes through memory.
#ifdef FH_PROFILE
          const unsigned long long tglue__ = pinned_clock();
#endif
          unsigned long long ahead_b = 0ull, ahead_ei = 0ull, ahead_wt = 0ull;
          const int pool_end_now = uniform_i32(sv.tb[sv.TB_NEXT + 1]);
#ifdef FH_NO_AHEAD  // (A/B builds)
          const bool draw_ahead = false;
#else
          const bool draw_ahead = tickets_left && uniform_i32(sv.tb[sv.TB_NEXT]) >= pool_end_now;
#endif
          const int ahead_ch = sv.ticket_chunk(ka.n, (int)gridDim.x, pool_end_now);
          if (draw_ahead && threadIdx.x == 0) {  // issued here, stored after the hand-off
            ahead_b = aadd(&sa.ctl->ticket, (unsigned long long)ahead_ch) + (unsigned long long)(sa.claims ? sv.static_tickets(ka.n, (int)gridDim.x) : 0);
            ahead_ei = ald(reinterpret_cast<unsigned long long*>(&sa.ctl->error));
            ahead_wt = ald(reinterpret_cast<unsigned long long*>(&sa.ctl->wait_ticket));
          }
          {
            typedef const __attribute__((address_space(4))) fh_problem const_problem;
            const const_problem& pw = *(const_problem*)sv.uniform_u64((unsigned long long)&problems[unit]);
            const const_problem& ps = *(const_problem*)sv.uniform_u64((unsigned long long)&ka.safe[unit]);
#ifdef FH_PROFILE
            unsigned long long probe[4] = {0ull, 0ull, 0ull, 0ull};
            staged = sv.handoff_onchip(pw, ps, faces, ka.par.feas_tol, ka.r_frac, ka.shrink, ka.max_safe_poly, ka.r_margin, ka.rule, UNK ? &ka.unknown : nullptr, probe);
            if (probe[0]) {
              sv.glue_parts[0] = probe[0] - tglue__; sv.glue_parts[1] = probe[1] - probe[0]; sv.glue_parts[2] = probe[2] - probe[1];
              sv.glue_parts[3] = probe[3] - probe[2];
            }
#else
            staged = sv.handoff_onchip(pw, ps, faces, ka.par.feas_tol, ka.r_frac, ka.shrink, ka.max_safe_poly, ka.r_margin, ka.rule, UNK ? &ka.unknown : nullptr, nullptr);
#endif
            staged = uniform_i32(staged);
          }
#ifdef FH_PROFILE
          const unsigned long long tdrain__ = __builtin_readcyclecounter();
#endif
          if (draw_ahead) {
            drain_stores();  // the ticket has arrived (with the hand-off's rows as a rule: issued before them, the counter is in order)
            if (threadIdx.x == 0) {
              const unsigned long long nn = (unsigned long long)ka.n;
              if (ahead_b < nn) {
                sv.tb[sv.TB_NEXT] = (int)ahead_b;
                sv.tb[sv.TB_NEXT + 1] = (int)min(ahead_b + (unsigned long long)ahead_ch, nn);
              } else {  // beyond the batch: an empty pool at the end of the batch (the next draw finds the same and ends the tickets)
                sv.tb[sv.TB_NEXT] = ka.n;
                sv.tb[sv.TB_NEXT + 1] = ka.n;
              }
              sv.tb_put64(sv.TB_NEXT_EI, ahead_ei);
              sv.tb_put64(sv.TB_NEXT_WT, ahead_wt);
            }
          }
#ifdef FH_PROFILE
          sv.drain_cycles = __builtin_readcyclecounter() - tdrain__;
#endif
          // the whole problem's result: its stores come after every load of the hand-off, and nothing waits for them
          if (FH_DEFER_RESULT) sv.flush_result(results[unit]);
          if (sa.pair_outputs) {  // the safe problem as a record and rows in memory, as the staged hand-off leaves them (the tests compare)
            if (staged == 1) {
              write_safe_problem(&problems[unit], faces, &ka.safe[unit], ka.sfaces, (lds_cdouble*)sv.Pc, 3 * sv.NT, uniform_i32(sv.tb[sv.TB_SAFE]) & 0xff, sv.P, ka.shrink,
                                 ka.r_margin, opaque((int)threadIdx.x));
              if (threadIdx.x == 0) sv.tb[sv.TB_SAFE] |= 0x100;
            } else if (threadIdx.x == 0) {
              __hip_atomic_store(&ka.safe[unit].n_seg, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
#ifdef FH_PROFILE
          sv.glue_cycles = pinned_clock() - tglue__;
#endif
          if (staged == 2) {  // the pair has no safe problem: its safe result says FH_ST_BAD_INPUT, as that of a record the staged hand-off marks with n_seg = 0
            sv.N = 0; sv.flops = 0ull;
            sv.emit_result(false, 0, FH_ST_BAD_INPUT, 0, 0, 0.0, 0.0, 0.0, result_rows<NSEG>(sa));
            sv.flush_result(ka.sres[unit]);
            sv.unit_done(sa);
            break;
          }
          phase = 1;
          entry = 0;
          if (threadIdx.x == 0) sv.tb[sv.TB_PHASE] = 1;
          continue;
        }
      }
      sv.unit_done(sa);  // one more unit done, and what it cost
      break;
    }
  }
#ifdef FH_SHARE_PROFILE
  if (threadIdx.x == 0 && sp_dry__) { aadd(&sa.ctl->prof2[6], wall_ticks() - sp_dry__); aadd(&sa.ctl->prof2[7], 1ull); }
#endif
  // The last workgroup to leave hands the control block to the next launch of this context in its initial state (counters zero,
  // ring sequence numbers i), after copying what the host wants to read.  No separate initialisation kernel per launch: on a
  // chip whose vector registers are all held by persistent workgroups of other launches, such a tiny kernel waited milliseconds
  // for a slot (rocprofv3: 12 ms average with 8 launches in flight), and a single genNewTraj() paid a second launch.
  sv.flush_done(sa);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int last = 0;
  if (threadIdx.x == 0) last = aadd(&sa.ctl->exited, 1u) == gridDim.x - 1u ? 1 : 0;
  if (uniform_i32(last)) {
    ShareCtl* c = sa.ctl;
    if (threadIdx.x == 0) {
      unsigned int rp[8] = {ald(&c->donated), ald(&c->stolen), ald(&c->q_full), ald(&c->rec_full), ald(&c->rec_next), ald(&c->error),
                            ald(&c->interrupted), ald(&c->max_fill)};
      for (int i = 0; i < 8; i++) ast(&c->report[i], rp[i]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned int* w = reinterpret_cast<unsigned int*>(c);
    if (threadIdx.x < 48) ast(&w[threadIdx.x], 0u);  // lines 0-2: done/error/interrupted, ticket/statistics/exited, wait_ticket/q_tail
    for (int i = threadIdx.x; i < FH_QCAP; i += 64) ast(&sa.seqs[i], (unsigned long long)i);
    if (sa.claims)
      for (int i = threadIdx.x; i < (int)gridDim.x; i += 64) ast(&sa.claims[i], 0u);
  }
}

// Launch order of a batch: units sorted by the number of polytopes of their (whole) corridor, most first — a counting sort in two
// small launches (histogram, scatter; `counters`: 2 * (FH_MAX_POLY + 1) zeroed ints).  The work of a problem grows steeply with the
// number of polytopes (C4: 52 active-set iterations per pair with 2, 320 with 6; 34 of the 41 hardest pairs in 4096 have 6), and a
// hard tree that is started late is what a launch ends on.  The order inside a class is whatever the atomics give: no result
// depends on it.
// [r5] Within a class of equal polytope count the corridors that are SHORT for their number of polytopes go first: on the C4 batch the
// 1 % of the problems with most active-set iterations are 6-polytope corridors whose start and goal are 9.6 m apart on average
// (all 6-polytope corridors: 13.9 m) — a corridor that curls needs the late segments decided against the early ones — and with
// metres-per-polytope as the second key 19 of the 20 longest problems of 32768 are among the first 380 tickets (by polytope count alone
// they are spread over the first 6200: two rounds of the resident grid later).  A scheduling hint in four buckets; no result depends on it.
#define FH_ORDER_CLASSES (4 * (FH_MAX_POLY + 1))
__device__ __forceinline__ int order_class(const fh_problem& p) {
  const int k = min(max(p.n_poly, 0), FH_MAX_POLY);
  float d2 = 0.f;  // one axis at a time (the fences keep the six loads from being issued together: these kernels live on 8 registers)
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float d = (float)(p.xf[a] - p.x0[a]);
    d2 = fmaf(d, d, d2);
    asm volatile("" : "+v"(d2)::"memory");
  }
  const float kk = (float)max(k * k, 1);  // (metres per polytope)^2 against 1.5^2, 2.2^2, 3^2 — without a division
  return 4 * k + (d2 < 2.25f * kk ? 3 : (d2 < 4.84f * kk ? 2 : (d2 < 9.0f * kk ? 1 : 0)));
}
#define FH_ORDER_BLOCK 64  // one wavefront per workgroup: it is placed on whichever SIMD has room (see below)
__global__ void __launch_bounds__(FH_ORDER_BLOCK) order_hist_kernel(const fh_problem* __restrict__ problems, int n, int* __restrict__ counters) {
  __shared__ int cnt[FH_ORDER_CLASSES];
  if (threadIdx.x < FH_ORDER_CLASSES) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int i = (int)(blockIdx.x * FH_ORDER_BLOCK + threadIdx.x);
  if (i < n) atomicAdd(&cnt[order_class(problems[i])], 1);
  __syncthreads();
  if (threadIdx.x < FH_ORDER_CLASSES && cnt[threadIdx.x]) atomicAdd(&counters[threadIdx.x], cnt[threadIdx.x]);
}
// (Round 4's order kernels fit into 8 vector registers: with three 168-register solve wavefronts on a SIMD that is exactly what is left,
// so they slipped in beside the persistent solve kernels of the other streams; a version that also computed dt_initial of every problem
// — 87 registers, 256 threads — had to wait for a solve wavefront to leave: the timed region of bench.py lost 12 %.  With the second key
// they need 11 / 14 registers.  They are single-wavefront workgroups now: a CU holds 11 solves, i.e. one of its four SIMDs has a free
// wavefront slot and 176 free registers, and a one-wavefront workgroup is placed there, where a 256-thread one wanted a slot on every SIMD.)
__global__ void __launch_bounds__(FH_ORDER_BLOCK)
order_scatter_kernel(const fh_problem* __restrict__ problems, int n, int window, int* __restrict__ counters, int* __restrict__ order) {
  __shared__ int cnt[FH_ORDER_CLASSES], base[FH_ORDER_CLASSES];
  __shared__ int last_block;
  if (threadIdx.x < FH_ORDER_CLASSES) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int i = (int)(blockIdx.x * FH_ORDER_BLOCK + threadIdx.x);
  int k = 0, mine = 0;
  if (i < n) {
    k = order_class(problems[i]);
    mine = atomicAdd(&cnt[k], 1);
  }
  __syncthreads();
  if (threadIdx.x < FH_ORDER_CLASSES) {  // where this block's members of class threadIdx.x go: classes in descending order, blocks as they come
    int before = 0;
    for (int c = FH_ORDER_CLASSES - 1; c > (int)threadIdx.x; c--) before += counters[c];
    base[threadIdx.x] = before + (cnt[threadIdx.x] ? atomicAdd(&counters[FH_ORDER_CLASSES + threadIdx.x], cnt[threadIdx.x]) : 0);
  }
  __syncthreads();
  if (i < n) {
    // Rank r in the hardness order -> ticket.  Tickets are drawn FH_TICKET_CHUNK at a time by one workgroup: neighbours in rank must not
    // be neighbours in ticket, or the first workgroups would each hold four of the hardest problems one behind the other (measured: one
    // launch alone 3.6 -> 4.3 ms).  Inside every window of `window` ranks (a multiple of the resident grid, fh_capi.hip) ticket 4 w + j is
    // rank j Q + w, Q = a quarter of the window: a chunk holds one problem of each quarter of its window, the Q hardest are the first
    // tickets of Q different chunks, and the windows keep the order hardest first (ONE window over a batch with a large hard class — C5:
    // a quarter of 65536 problems — would start the last of that class at 80 % of the launch: 12.2 -> 13.1 ms).  The last window is as
    // long as what is left.
    const int r = base[k] + mine, w0 = r - r % window, wsize = min(window, n - w0) & ~(FH_TICKET_CHUNK - 1), rr = r - w0;
    const int Q = wsize / FH_TICKET_CHUNK;
    order[rr < wsize ? w0 + FH_TICKET_CHUNK * (rr % Q) + rr / Q : r] = i;
  }
  // the last block to finish leaves the counters zeroed for the next launch (no memset in the stream: on a chip whose registers are
  // all held by persistent workgroups every extra stream operation waits milliseconds for a slot)
  __syncthreads();
  if (threadIdx.x == 0) last_block = atomicAdd(&counters[2 * FH_ORDER_CLASSES], 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (last_block)
    for (int c = (int)threadIdx.x; c <= 2 * FH_ORDER_CLASSES; c += FH_ORDER_BLOCK) counters[c] = 0;
}

// FP64 vector peak of the device as this code can reach it: independent v_fma_f64 chains, 8 per lane, no memory traffic
// ("measure a peak-FMA microbenchmark and use the measured number").  2 flops per FMA.
__global__ void __launch_bounds__(256) fp64_peak_kernel(double* out, int iters, double seed) {
  double a[8];
#pragma unroll
  for (int j = 0; j < 8; j++) a[j] = seed + (double)(threadIdx.x + j);
  const double m = 1.0 - 1e-9, c = 1e-9;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = fma(a[j], m, c);
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) s += a[j];
  if (s == 12345.678) out[blockIdx.x * blockDim.x + threadIdx.x] = s;  // (never true: keeps the chains alive)
}

}  // namespace fh
