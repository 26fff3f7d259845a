// fh_sample.hip.hpp — resetX()+fillX() and the whole->safe hand-off on the device (gfx950, wave64).
//
// sample_kernel  : SolverGurobi::resetX (:382-388) + fillX (:122-168) of /root/reference/faster/src/solverGurobi.cpp.
//                  One wavefront per trajectory.  The sample clock is the reference's repeated `t = t + DC`
//                  (NOT (i+1)*DC: the rounding of the running sum decides which segment a sample on a knot
//                  belongs to), so every lane runs the scalar clock and keeps the tick that is its own; the 64
//                  states of a tile are transposed through LDS and written as one contiguous 6 KiB burst
//                  (16 B per lane per store) instead of 96-B-strided scalars.
// pair_glue_kernel: the data dependency whole -> safe of Faster::replan (faster/src/faster.cpp:456-475, :506-524)
//                  for synthetic pairs: R = sample (int)(r_frac*count) of the whole
//                  trajectory becomes x0 of the safe problem; the safe corridor is the run of up to
//                  `max_safe_poly` consecutive polytopes starting at the first one that contains R, each shrunk
//                  by `shrink` metres (emulating the unknown-space inflation).  One wavefront per pair.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/fasterhip.h"
#include "fh_clock.hpp"

namespace fh {

// (PW / RW: anything with the members of fh_problem / fh_result that are used — the records in memory, or the views the fused pair
// kernel builds from what its wavefront still holds in registers and LDS: ProblemView, ResultView)
template <class PW, class RW>
__device__ __forceinline__ int sample_count(const PW& pr, const RW& rs) {
  int size = (int)((double)((int)pr.n_seg) * rs.dt / pr.dc);  // :384
  return size < 2 ? 2 : size;                                  // :385
}

__device__ __forceinline__ void eval_state(const double* c, double tau, bool last, fh_state& s) {
#pragma unroll
  for (int a = 0; a < 3; a++) {
    s.pos[a] = c[0 + a] * tau * tau * tau + c[3 + a] * tau * tau + c[6 + a] * tau + c[9 + a];  // getPos :761-767
    s.vel[a] = 3 * c[0 + a] * tau * tau + 2 * c[3 + a] * tau + c[6 + a];                      // getVel :769-774
    s.accel[a] = 6 * c[0 + a] * tau + 2 * c[3 + a];                                           // getAccel :776-781
    s.jerk[a] = 6 * c[0 + a];                                                                 // getJerk :783-788
    if (last) { s.vel[a] = 0; s.accel[a] = 0; s.jerk[a] = 0; }                                // :165-167
  }
}

// The first `nwrite` samples of one trajectory (of `size`) to out[0 .. nwrite): the body of sample_kernel.  tile: [64 * 12], coef: [FH_MAX_SEG * 12] (LDS).
__device__ inline void sample_into(const fh_problem& pr, const fh_result& rs, int size, int nwrite, fh_state* __restrict__ out, double* tile,
                                   double* coef, int lane) {
  const int N = __builtin_amdgcn_readfirstlane(pr.n_seg);
  const double dt = rs.dt, DC = pr.dc;
  __syncthreads();
  for (int i = lane; i < N * 12; i += 64) coef[i] = rs.coeff[i / 12][i % 12];
  __syncthreads();
  double t = 0;
  int interval = 0;
  double knot = dt * (double)(interval + 1);  // dt_ * (interval + 1) of :133, recomputed only when the interval changes
  for (int base = 0; base < nwrite; base += 64) {
    double my_t = 0;
    int my_int = 0;
    const int lim = __builtin_amdgcn_readfirstlane((nwrite - base) < 64 ? (nwrite - base) : 64);
    for (int j = 0; j < lim; j++) {  // the reference's scalar clock, :131-135: every lane runs it and keeps its own tick
      t = t + DC;
      if (__ballot(t > knot) != 0ull) {  // wave-uniform; a real (rarely taken) branch, not two selects per tick
        asm volatile("");
        interval = (interval + 1 < N - 1) ? interval + 1 : N - 1;
        knot = dt * (double)(interval + 1);
      }
      const bool mine = j == lane;
      my_t = mine ? t : my_t;
      my_int = mine ? interval : my_int;
    }
    if (lane < lim) {
      fh_state s;
      eval_state(&coef[my_int * 12], my_t - my_int * dt, (base + lane) == size - 1, s);
      double* tl = &tile[lane * 12];
#pragma unroll
      for (int a = 0; a < 3; a++) { tl[a] = s.pos[a]; tl[3 + a] = s.vel[a]; tl[6 + a] = s.accel[a]; tl[9 + a] = s.jerk[a]; }
    }
    __syncthreads();
    // contiguous burst: lim*12 doubles, two per lane per store
    typedef double vec2 __attribute__((ext_vector_type(2)));
    vec2* dst = reinterpret_cast<vec2*>(out + base);
    const vec2* src = reinterpret_cast<const vec2*>(tile);
    for (int i = lane; i < lim * 6; i += 64) __builtin_nontemporal_store(src[i], &dst[i]);  // streamed once, never re-read here
    __syncthreads();
  }
}

__global__ void __launch_bounds__(64) sample_kernel(const fh_problem* __restrict__ problems, const fh_result* __restrict__ results,
                                                    int n, int max_samples, fh_state* __restrict__ states,
                                                    int32_t* __restrict__ counts) {
  __shared__ __attribute__((aligned(16))) double tile[64 * 12];
  __shared__ double coef[FH_MAX_SEG * 12];
  const int b = blockIdx.x;
  if (b >= n) return;
  const int lane = threadIdx.x;
  const fh_problem& pr = problems[b];
  const fh_result& rs = results[b];
  if (!rs.solved || pr.n_seg < 1 || pr.n_seg > FH_MAX_SEG) {
    if (lane == 0) counts[b] = 0;
    return;
  }
  const int size = __builtin_amdgcn_readfirstlane(sample_count(pr, rs));  // wave-uniform: scalar loop counters below
  if (lane == 0) counts[b] = size;
  sample_into(pr, rs, size, size < max_samples ? size : max_samples, states + (size_t)b * (size_t)max_samples, tile, coef, lane);
}

// One wavefront per pair: the scalar sample clock runs once (wave-uniform), the polytope tests and the face copy are
// lane-parallel over faces with coalesced 32-B rows.
// r_margin < 0: SURVEY.md 8(d) to the letter — every polytope of the safe corridor is the whole one shrunk by `shrink`, starting at
// the first shrunk polytope that contains R (else the least violated one: R may then lie outside its own corridor and the safe
// problem is infeasible for every factor).  r_margin >= 0: FASTER decomposes the safe corridor around R (faster.cpp:475-499: R is the
// first vertex of JPS_safe), so R is strictly inside its first polytope: the corridor starts at the first polytope that contains R
// and no face of that polytope is pulled closer to R than r_margin (faces R already touches stay where they are).
// WT: the safe problem record and its face rows are written with write-through (agent-scope, `sc1`) stores: inside the fused pair
// kernel they are read back in the same launch, possibly by a workgroup on another XCD, and a release FENCE there would write
// back the XCD's whole dirty L2 once per pair (measured: 0.7 GB of HBM writes per 32768-pair launch).
template <bool WT>
__device__ __forceinline__ void glue_store(double* p, double v) {
  if (WT) __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool WT>
__device__ __forceinline__ void glue_store(int32_t* p, int32_t v) {
  if (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
// sample i of fillX (solverGurobi.cpp:122-168) without running the clock: t = (i + 1) DC, segment = boundaries crossed (the
// reference's `if (t > dt (interval + 1)) interval++` once per sample, DC <= dt / 2).  Used by the search for R only — where two
// pieces meet they agree to rounding; the state that becomes x0 of the safe problem is evaluated with the reference's own clock.
template <class RW>
__device__ __forceinline__ void state_at(const RW& rw, int N, double DC, int i, int size, fh_state& s) {
  const double t = (double)(i + 1) * DC;
  int interval = (int)ceil(t / rw.dt) - 1;
  interval = interval < 0 ? 0 : (interval > N - 1 ? N - 1 : interval);
  eval_state(rw.coeff[interval], t - interval * rw.dt, i == size - 1, s);
}
// Unknown space as an INPUT (fh_set_unknown_grid_device; fh_pair_rule mode 2): the mapper's unknown voxels on a lattice — cell (ix, iy,
// iz) with flags[(iz ny + iy) nx + ix] != 0 is unknown and stands for its centre ((i + 0.5) res + origin), the point a cloud of the
// mapper's unknown voxels (pclptr_unk_ / vec_uo_ of FASTER: faster.cpp:99-137, jps_manager.cpp:91-98) would hold.  flags == null: none.
struct UnknownGrid {
  const unsigned char* flags;
  double ox, oy, oz, res;
  int nx, ny, nz, pad;
};
// Is an unknown voxel centre closer than `radius` to p?  What `kdtree_unk_.nearestKSearch(p, 1, ...)` followed by `sqrt(d2) < radius`
// decides in findIndexH (faster.cpp:236-240) — evaluated in DOUBLE precision against double voxel centres.  The reference searches a
// pcl::PointXYZ cloud: its query point, its voxel centres and d2 are single precision (faster.cpp:233-238); for a sample within float
// rounding (~1e-7 relative) of `radius` from a voxel centre the two can decide differently.  oracle/pair_glue.py shares the double model.
// One lane, the cells around p's own.
__device__ inline bool unknown_within(const UnknownGrid& ug, double px, double py, double pz, double radius) {
#pragma clang fp contract(off)
  if (!ug.flags || !(radius > 0)) return false;
  const int k = (int)floor(radius / ug.res + 0.5) + 1;
  const int cx = (int)floor((px - ug.ox) / ug.res), cy = (int)floor((py - ug.oy) / ug.res), cz = (int)floor((pz - ug.oz) / ug.res);
  const int x0 = max(cx - k, 0), x1 = min(cx + k, ug.nx - 1), y0 = max(cy - k, 0), y1 = min(cy + k, ug.ny - 1);
  const int z0 = max(cz - k, 0), z1 = min(cz + k, ug.nz - 1);
  bool hit = false;
  for (int iz = z0; iz <= z1; iz++)
    for (int iy = y0; iy <= y1; iy++) {
      const unsigned char* row = ug.flags + ((size_t)iz * ug.ny + iy) * ug.nx;
      const double qy = ((double)iy + 0.5) * ug.res + ug.oy, qz = ((double)iz + 0.5) * ug.res + ug.oz;
      const double dy = qy - py, dz = qz - pz;
      for (int ix = x0; ix <= x1; ix++) {
        if (!row[ix]) continue;
        const double qx = ((double)ix + 0.5) * ug.res + ug.ox, dx = qx - px;
        hit = hit || sqrt(dx * dx + dy * dy + dz * dz) < radius;
      }
    }
  return hit;
}

// What the hand-off reads of a problem / a result, for a caller that holds them in registers and LDS (the fused pair kernel: the whole
// problem has just been solved by this wavefront; reading its record and its result back from memory costs two dependent round trips)
struct ProblemView {
  int n_seg, n_poly, face_begin;
  double dc, a_max;
  double x0[3];
  const int* face_off;  // [n_poly + 1]
};
struct ResultView {
  int solved;
  double dt;
  const double (*coeff)[12];  // [n_seg][12], createVars order
};
// max over the wavefront (DPP row_shr 1/2/4/8, row_bcast 15/31; total in lane 63), any sign
__device__ __forceinline__ double glue_wave_max(double v) {
  auto step = [](double x, int ctrl_sel) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    const int nlo = __double2loint(-INFINITY), nhi = __double2hiint(-INFINITY);
    switch (ctrl_sel) {
      case 0: lo = __builtin_amdgcn_update_dpp(nlo, lo, 0x111, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(nhi, hi, 0x111, 0xf, 0xf, false); break;
      case 1: lo = __builtin_amdgcn_update_dpp(nlo, lo, 0x112, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(nhi, hi, 0x112, 0xf, 0xf, false); break;
      case 2: lo = __builtin_amdgcn_update_dpp(nlo, lo, 0x114, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(nhi, hi, 0x114, 0xf, 0xf, false); break;
      case 3: lo = __builtin_amdgcn_update_dpp(nlo, lo, 0x118, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(nhi, hi, 0x118, 0xf, 0xf, false); break;
      case 4: lo = __builtin_amdgcn_update_dpp(nlo, lo, 0x142, 0xa, 0xf, false); hi = __builtin_amdgcn_update_dpp(nhi, hi, 0x142, 0xa, 0xf, false); break;
      default: lo = __builtin_amdgcn_update_dpp(nlo, lo, 0x143, 0xc, 0xf, false); hi = __builtin_amdgcn_update_dpp(nhi, hi, 0x143, 0xc, 0xf, false); break;
    }
    return __hiloint2double(hi, lo);
  };
#pragma unroll
  for (int k = 0; k < 6; k++) v = fmax(v, step(v, k));
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}

// Which sample of the whole trajectory is R (k_safe of Faster::replan): mode 0 sample (int)(r_frac * size); mode 1 FASTER's own rule —
// findIndexH (faster.cpp:218-251) against the modelled unknown space, then findIndexR (:173-216).  Returns false when no safe
// trajectory is needed (needToComputeSafePath == false, :462-466): k is then indexH = the last sample (:231).
template <class PW, class RW>
__device__ inline bool choose_r_index(const PW& pw, const RW& rw, double r_frac, const fh_pair_rule& rule, int lane, int& k,
                                      const UnknownGrid* ug = nullptr) {
  const int N = pw.n_seg;
  const double DC = pw.dc;
  const int size = sample_count(pw, rw);
  k = (int)(r_frac * (double)size);
  if (rule.mode == 1 || rule.mode == 2) {
    // findIndexH (faster.cpp:218-251): samples 0, 10, 20, ... against unknown space — mode 1: modelled (farther than r_known from x0);
    // mode 2: the unknown voxels of the caller's grid (nearest unknown voxel centre closer than drone_radius)
    const double lim = rule.r_known - rule.drone_radius;
    int iH = 0x7fffffff;
    for (int base = 0; 10 * base < size && iH == 0x7fffffff; base += 64) {
      const int i = 10 * (base + lane);
      int mine = 0x7fffffff;
      if (i < size) {
        fh_state s;
        state_at(rw, N, DC, i, size, s);
        if (rule.mode == 2) {
          if (ug && unknown_within(*ug, s.pos[0], s.pos[1], s.pos[2], rule.drone_radius)) mine = i;
        } else {
          const double dx = s.pos[0] - pw.x0[0], dy = s.pos[1] - pw.x0[1], dz = s.pos[2] - pw.x0[2];
          if (sqrt(dx * dx + dy * dy + dz * dz) > lim) mine = i;
        }
      }
      iH = wave_min_i32(mine);
    }
    if (iH == 0x7fffffff) {  // needToComputeSafePath == false (:462-466): the pair ends with its whole trajectory
      k = size - 1;
      return false;
    }
    int indexH = (int)(rule.delta_h * (double)iH);
    indexH = indexH > size - 1 ? size - 1 : (indexH < 0 ? 0 : indexH);
    fh_state sH;
    state_at(rw, N, DC, indexH, size, sH);
    // findIndexR (faster.cpp:173-216): first sample from which braking before H is no longer possible (x and y only)
    const double den = 2.0 * rule.delta_a * pw.a_max;
    int iR = indexH;
    for (int base = 0; base <= indexH && iR == indexH; base += 64) {
      const int i = base + lane;
      int mine = indexH;
      if (i <= indexH) {
        fh_state s;
        state_at(rw, N, DC, i, size, s);
        bool collision = false;
#pragma unroll
        for (int a = 0; a < 2; a++) {
          const double diff = sH.pos[a] - s.pos[a], w = s.vel[a] * diff;
          const double sg = w > 0.0 ? 1.0 : (w < 0.0 ? -1.0 : 0.0);
          collision |= sg * s.vel[a] * s.vel[a] / den > fabs(diff);
        }
        if (collision) mine = i;
      }
      iR = wave_min_i32(mine);
    }
    k = iR;
  }
  if (k > size - 1) k = size - 1;
  if (k < 0) k = 0;
  return true;
}

// ---- pieces of the hand-off that its three users share — the staged kernel (pair_glue_kernel), the fused pair kernel's on-chip
// hand-off (Solver::handoff_onchip, fh_solve.hip.hpp) and the write-back of a safe problem (write_safe_problem) —: ONE copy of every
// formula, so that all of them round alike ----
// (no contraction into fused multiply-adds left to the optimiser's choice per inline site: the three users must produce the same bits)
__device__ __forceinline__ double glue_face_norm(const fh_face& fc) {
#pragma clang fp contract(off)
  return sqrt(fc.a[0] * fc.a[0] + fc.a[1] * fc.a[1] + fc.a[2] * fc.a[2]);
}
// how far R is outside row fc of the (shrunk) whole corridor: > slack means outside
__device__ __forceinline__ double glue_face_test(const fh_face& fc, double nr, const double (&Rp)[3], double test_shrink) {
#pragma clang fp contract(off)
  return fc.a[0] * Rp[0] + fc.a[1] * Rp[1] + fc.a[2] * Rp[2] - (fc.b - test_shrink * nr);
}
// right-hand side of row fc in the safe corridor: shrunk by `shrink` metres; a row of the polytope that holds R (first) is not pulled
// closer to R than r_margin, and never past R (keep_r: r_margin >= 0)
__device__ __forceinline__ double glue_face_b(const fh_face& fc, double nr, const double (&Rp)[3], double shrink, bool keep_r, bool first, double r_margin) {
#pragma clang fp contract(off)
  double b = fc.b - shrink * nr;
  if (keep_r && first) {
    const double ar = fc.a[0] * Rp[0] + fc.a[1] * Rp[1] + fc.a[2] * Rp[2];
    b = fmax(b, fmax(fmin(fc.b, ar + r_margin * nr), ar + 1e-6 * nr));
  }
  return b;
}
// R = the state of the whole trajectory at the sample the rule picks (k).  false: the pair has no safe problem (no whole trajectory:
// the reference returns from replan; or needToComputeSafePath == false).
template <class PW, class RW>
__device__ __forceinline__ bool glue_find_r(const PW& pw, const RW& rw, double r_frac, const fh_pair_rule& rule, int lane, const UnknownGrid* ug,
                                            fh_state& R, unsigned long long* probe) {
  // (R is evaluated on every path — at the start of the trajectory when there is nothing to hand off — so that a caller may run on
  // unconditionally and decide at its end; rows of an unsolved result are zeros)
  bool ok = rw.solved && pw.n_seg >= 1 && pw.n_seg <= FH_MAX_SEG;
  const int N = pw.n_seg;
  const double dt = rw.dt, DC = pw.dc;
  int k = 0, size = 2;
  if (ok) {
    size = sample_count(pw, rw);
    ok = choose_r_index(pw, rw, r_frac, rule, lane, k, ug);
  }
  if (probe) probe[0] = __builtin_readcyclecounter();
  double t = 0;
  int interval = 0;
  if (ok) clock_at(k, DC, dt, N, t, interval);  // the reference's clock at sample k (solverGurobi.cpp:131-135) — its own double, not (k + 1) DC: fh_clock.hpp
  if (probe) probe[1] = __builtin_readcyclecounter();
  eval_state(rw.coeff[interval], t - interval * dt, ok && k == size - 1, R);
  return ok;
}
// first polytope whose worst row lets R in (worst <= slack), else the least violated one
__device__ __forceinline__ int glue_pick_start(const double (&worst_p)[FH_MAX_POLY], int P, double slack_ok) {
  int start = 0;
  double best = INFINITY;
  bool found = false;
#pragma unroll
  for (int p = 0; p < FH_MAX_POLY; p++) {
    if (p < P && !found) {
      if (worst_p[p] <= slack_ok) { start = p; found = true; }
      else if (worst_p[p] < best) { best = worst_p[p]; start = p; }
    }
  }
  return start;
}
// The safe problem of a pair as a record + face rows in memory: x0 = R, the corridor = `cnt` polytopes of the whole corridor from
// polytope `start` on (rows [src0, src0 + total) of the whole problem, the first first_end of them the polytope that holds R), written at
// the whole problem's face_begin.  woff: the whole problem's face offsets.  WT: write-through stores (see glue_store).
template <bool WT, class OFF>
__device__ __forceinline__ void glue_write_safe(const fh_face* wfaces, int fb, const OFF& woff, int start, int cnt, const double (&x0)[9], double shrink,
                                                double r_margin, fh_problem& ps, fh_face* sfaces, int lane, unsigned long long* probe) {
  const bool keep_r = r_margin >= 0.0;
  const double Rp[3] = {x0[0], x0[1], x0[2]};
  if (lane < 9) {  // (selected, not indexed: an array indexed by the lane would live in scratch memory)
    double v = x0[0];
#pragma unroll
    for (int j = 1; j < 9; j++) v = lane == j ? x0[j] : v;
    glue_store<WT>(&ps.x0[lane], v);
  }
  const int src0 = cnt ? woff[start] : 0;
  const int total = cnt ? woff[start + cnt] - src0 : 0;
  const int first_end = cnt ? woff[start + 1] - src0 : 0;  // rows of the polytope that holds R
  if (WT) {
    // write-through stores are one fabric write per separate piece: lane = one DOUBLE of the output (4 per face row), so that an
    // instruction writes 512 contiguous bytes (a lane per face row, 8 bytes at a stride of 32, cost 8x the HBM write traffic)
    double* out = reinterpret_cast<double*>(sfaces + fb);
    for (int j = lane; j < 4 * total; j += 64) {
      const int f = j >> 2, c = j & 3;
      const fh_face fc = wfaces[fb + src0 + f];
      const double b = glue_face_b(fc, glue_face_norm(fc), Rp, shrink, keep_r, f < first_end, r_margin);
      glue_store<true>(out + j, c == 0 ? fc.a[0] : (c == 1 ? fc.a[1] : (c == 2 ? fc.a[2] : b)));
    }
  } else {
    for (int f = lane; f < total; f += 64) {
      fh_face fc = wfaces[fb + src0 + f];
      fc.b = glue_face_b(fc, glue_face_norm(fc), Rp, shrink, keep_r, f < first_end, r_margin);
      sfaces[fb + f] = fc;
    }
  }
  if (probe) probe[3] = __builtin_readcyclecounter();
  if (lane <= FH_MAX_POLY) {
    const int p = lane < cnt ? lane : cnt;
    glue_store<WT>(&ps.face_off[lane], cnt ? woff[start + p] - src0 : 0);
  }
  if (lane == 0) {
    glue_store<WT>(&ps.n_poly, (int32_t)cnt);
    glue_store<WT>(&ps.face_begin, (int32_t)fb);
  }
}

template <bool WT = false, class PW = fh_problem, class RW = fh_result>
__device__ inline void pair_glue_one(const PW& pw, const RW& rw, const fh_face* wfaces, double r_frac, double shrink,
                                     int max_safe_poly, double r_margin, const fh_pair_rule& rule, fh_problem& ps, fh_face* sfaces, int lane,
                                     unsigned long long* probe = nullptr,  // probe: cycle stamps of a diagnostic build (null otherwise)
                                     const UnknownGrid* ug = nullptr) {    // the unknown voxels of rule mode 2
  fh_state R;
  if (!glue_find_r(pw, rw, r_frac, rule, lane, ug, R, probe)) {  // no whole trajectory, or the pair ends with its whole trajectory
    if (lane == 0) glue_store<WT>(&ps.n_seg, 0);
    return;
  }
  const bool keep_r = r_margin >= 0.0;
  const double test_shrink = keep_r ? 0.0 : shrink;  // which polytope holds R: the original one / the shrunk one
  const double slack_ok = keep_r ? 1e-7 : 0.0;       // R is a point of the whole trajectory: inside its polytope up to the solver tolerance
  // first polytope that contains R, else the least violated one.  Every face of the whole corridor is tested at once, lane = face
  // (one memory round trip; a polytope after the other cost one each), then one maximum per polytope.
  const int P = pw.n_poly;
  const int fb = pw.face_begin;
  const int nf = P ? pw.face_off[P] : 0;
  double worst_p[FH_MAX_POLY];
#pragma unroll
  for (int p = 0; p < FH_MAX_POLY; p++) worst_p[p] = -INFINITY;
  for (int f0 = 0; f0 < nf; f0 += 64) {
    const int f = f0 + lane;
    double v = -INFINITY;
    int pf = -1;
    if (f < nf) {
      const fh_face fc = wfaces[fb + f];
      v = glue_face_test(fc, glue_face_norm(fc), R.pos, test_shrink);
      pf = 0;
      for (int p = 1; p < P; p++) pf += (f >= pw.face_off[p]) ? 1 : 0;
    }
#pragma unroll
    for (int p = 0; p < FH_MAX_POLY; p++)
      if (p < P) worst_p[p] = fmax(worst_p[p], glue_wave_max(pf == p ? v : -INFINITY));
  }
  if (probe) probe[2] = __builtin_readcyclecounter();
  const int start = glue_pick_start(worst_p, P, slack_ok);
  int cnt = P - start;
  if (cnt > max_safe_poly) cnt = max_safe_poly;
  if (P == 0) cnt = 0;
  const double x0[9] = {R.pos[0], R.pos[1], R.pos[2], R.vel[0], R.vel[1], R.vel[2], R.accel[0], R.accel[1], R.accel[2]};
  glue_write_safe<WT>(wfaces, fb, pw.face_off, start, cnt, x0, shrink, r_margin, ps, sfaces, lane, probe);
}

__global__ void __launch_bounds__(64) pair_glue_kernel(const fh_problem* __restrict__ whole, const fh_result* __restrict__ wres,
                                                       const fh_face* __restrict__ wfaces, int n, double r_frac, double shrink,
                                                       int max_safe_poly, double r_margin, fh_pair_rule rule, fh_problem* __restrict__ safe,
                                                       fh_face* __restrict__ sfaces, UnknownGrid ug) {
  const int b = blockIdx.x;
  if (b >= n) return;
  pair_glue_one<false>(whole[b], wres[b], wfaces, r_frac, shrink, max_safe_poly, r_margin, rule, safe[b], sfaces, threadIdx.x, nullptr, &ug);
}

// Faster::appendToPlan (faster/src/faster.cpp:606-648) for a batch of independent pairs whose plan holds only the start A
// (k_end_whole = 0): plan = the samples 0 .. k_safe of the whole trajectory, then every sample of the safe trajectory (:627-640).
// k_safe is the index the hand-off chose (choose_r_index, the same function).  A pair commits nothing (count 0) when the whole solve
// failed (:427-431) or a safe trajectory was needed and not found (:529-533); a pair that needs no safe trajectory commits its whole
// trajectory (k_safe = indexH = last sample, safe part empty, :462-466).  One wavefront per pair.
__global__ void __launch_bounds__(64) plan_append_kernel(const fh_problem* __restrict__ whole, const fh_result* __restrict__ wres,
                                                         const fh_problem* __restrict__ safe, const fh_result* __restrict__ sres, int n,
                                                         double r_frac, fh_pair_rule rule, int max_states, fh_state* __restrict__ plans,
                                                         int32_t* __restrict__ counts, int32_t* __restrict__ k_safe_out, UnknownGrid ug) {
  __shared__ __attribute__((aligned(16))) double tile[64 * 12];
  __shared__ double coef[FH_MAX_SEG * 12];
  const int b = blockIdx.x;
  if (b >= n) return;
  const int lane = threadIdx.x;
  const fh_problem& pw = whole[b];
  const fh_result& rw = wres[b];
  int count = 0, k = -1;
  if (rw.solved && pw.n_seg >= 1 && pw.n_seg <= FH_MAX_SEG) {
    const int size_w = __builtin_amdgcn_readfirstlane(sample_count(pw, rw));
    const bool need_safe = choose_r_index(pw, rw, r_frac, rule, lane, k, &ug);
    k = __builtin_amdgcn_readfirstlane(k);
    const fh_problem& ps = safe[b];
    const fh_result& rs = sres[b];
    const bool have_safe = need_safe && rs.solved && ps.n_seg >= 1 && ps.n_seg <= FH_MAX_SEG;
    if (!need_safe || have_safe) {
      const int size_s = have_safe ? __builtin_amdgcn_readfirstlane(sample_count(ps, rs)) : 0;
      count = k + 1 + size_s;
      fh_state* out = plans + (size_t)b * (size_t)max_states;
      const int nw = (k + 1) < max_states ? (k + 1) : max_states;
      sample_into(pw, rw, size_w, nw, out, tile, coef, lane);
      if (have_safe && k + 1 < max_states) {
        const int ns = (max_states - (k + 1)) < size_s ? (max_states - (k + 1)) : size_s;
        sample_into(ps, rs, size_s, ns, out + (k + 1), tile, coef, lane);
      }
    }
  }
  if (lane == 0) {
    counts[b] = count;
    if (k_safe_out) k_safe_out[b] = count ? k : -1;
  }
}

// Faster::getNextGoal (faster/src/faster.cpp:699-723, without the yaw of getDesiredYaw: yaw planning is out of scope) for a batch of
// committed plans: plan i is the states plans[i][cursor[i] .. counts[i]) — `next_goal = plan_.front(); if (plan_.size() > 1)
// plan_.pop_front()`: the front state goes out, and the plan keeps at least its last state (a vehicle that has used up its plan
// hovers on the last state).  An empty plan (counts[i] == 0: nothing was committed) gives a zero state (next_goal.setZero()) and
// ok[i] = 0.  `ticks` calls in a row: the state after ticks - 1 pops is returned and the cursor advanced by up to `ticks`.
__global__ void __launch_bounds__(256) next_goal_kernel(const fh_state* __restrict__ plans, const int32_t* __restrict__ counts,
                                                        int32_t* __restrict__ cursor, int n, int max_states, int ticks,
                                                        fh_state* __restrict__ goals, int32_t* __restrict__ ok) {
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i >= n) return;
  const int cnt = counts[i] < max_states ? counts[i] : max_states;  // (states actually stored)
  fh_state g;
  for (int k = 0; k < 3; k++) { g.pos[k] = 0; g.vel[k] = 0; g.accel[k] = 0; g.jerk[k] = 0; }
  int c = cursor[i];
  if (cnt > 0) {
    c = c < 0 ? 0 : (c > cnt - 1 ? cnt - 1 : c);
    const int last_read = c + (ticks - 1) < cnt - 1 ? c + (ticks - 1) : cnt - 1;  // front() of the last of the `ticks` calls
    g = plans[(size_t)i * (size_t)max_states + last_read];
    c = c + ticks < cnt - 1 ? c + ticks : cnt - 1;                                // pop_front() while more than one state is left
    cursor[i] = c;
  }
  goals[i] = g;
  if (ok) ok[i] = cnt > 0 ? 1 : 0;
}

}  // namespace fh
