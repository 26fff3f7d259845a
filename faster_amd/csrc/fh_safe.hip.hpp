// fh_safe.hip.hpp — the safe corridor of Faster::replan decomposed around R on the device (gfx950, wave64).
//
// Reference: /root/reference/faster/src/faster.cpp:446-524.  After the whole trajectory is solved, FASTER
//   * cuts the path inside the sphere, JPS_in, where it first comes within drone_radius of UNKNOWN space (getFirstCollisionJPS(…,
//     UNKNOWN_MAP, RETURN_INTERSECTION), :451-452 → :767-926: marching in spheres known to be clear), backed off by drone_radius;
//   * finds H and R on the whole trajectory (findIndexH / findIndexR, :456-475);
//   * replaces the first vertex by R, keeps max_poly_safe legs (deleteVertexes, :488-490), M = the last vertex (:491);
//   * decomposes that path against the UNKNOWN + OCCUPIED points (cvxEllipsoidDecomp, :494) and starts the safe solver from R
//     towards M — or G, when G lies in the last polytope (:498-499).
// A batch of independent problems has no mapper: unknown space is MODELLED as in fh_set_pair_rule — everything farther than r_known
// from the start A of the whole problem — for the distance queries (r_known - |p - A|), and as the voxels of the map's grid whose
// centres lie out there for the decomposition (fh_decomp.hip.hpp: UnknownLattice).
// The CPU restatement these kernels are checked against is oracle/pair_glue.py (safe_path, unknown_voxels) + the host decomposition.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/fasterhip.h"
#include "fh_sample.hip.hpp"

#pragma clang fp contract(off)  // (single- and double-precision expressions restated in numpy: the same roundings)

namespace fh {

struct P3 { double x, y, z; };
__device__ __forceinline__ P3 p3(double x, double y, double z) { P3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ double dist3(P3 a, P3 b) {
  const double dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
  return sqrt(dx * dx + dy * dy + dz * dz);
}

// point where the segment a -> b leaves the sphere (centre c, radius r): the reference's arithmetic (utils.cpp:713-776; the same
// expressions as fhfront::sphere_crossing, host/corridor_frontend.hpp)
__device__ inline P3 sphere_crossing(P3 a_in, P3 b_in, double r, P3 c) {
  auto solve = [&](P3 A, P3 B, float& disc) {
    const float x1 = (float)A.x, y1 = (float)A.y, z1 = (float)A.z, x2 = (float)B.x, y2 = (float)B.y, z2 = (float)B.z;
    const float x3 = (float)c.x, y3 = (float)c.y, z3 = (float)c.z;
    const float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
    const float a = (float)((double)dx * (double)dx + (double)dy * (double)dy + (double)dz * (double)dz);  // pow(float, 2) is a double
    const float b = 2.0f * (dx * (x1 - x3) + dy * (y1 - y3) + dz * (z1 - z3));
    const float cf = x3 * x3 + y3 * y3 + z3 * z3 + x1 * x1 + y1 * y1 + z1 * z1 - 2.0f * (x3 * x1 + y3 * y1 + z3 * z1);
    const float cc = (float)((double)cf - r * r);                                                            // `- r * r`: a double subtraction
    disc = b * b - 4.0f * a * cc;
    const float t = (-b + sqrtf(disc)) / (2.0f * a);
    return p3((double)(x1 + dx * t), (double)(y1 + dy * t), (double)(z1 + dz * t));
  };
  float disc;
  const P3 first = solve(a_in, b_in, disc);
  if (disc <= 0) return solve(c, a_in, disc);  // tangent / no crossing: the ray centre -> a
  return first;
}

constexpr int SAFE_PATH_CAP = 40;

// Distance from p to the nearest unknown voxel centre of the caller's grid (rule mode 2) — what `kdtree_unk_.nearestKSearch(p, 1, ...)`
// returns in getFirstCollisionJPS (faster.cpp:806-812): the minimum of dx^2 + dy^2 + dz^2 over the unknown voxels, then the square
// root — exact in DOUBLE precision; the reference's kd-tree holds pcl::PointXYZ, i.e. single-precision points and a float d2 (:797-812).  The whole wavefront searches cubes of cells around p's cell, lane = cell; a cube of half-width w has seen every voxel
// closer than (w + 1/2) res, so the search ends when the best distance found is inside that bound (or the cube covers the grid).
// INFINITY: the grid has no unknown voxel (the reference's kd-tree is empty: the path is returned as it was).
// cap: the caller only asks whether the distance is below `cap` and, if so, what it is (the march of getFirstCollisionJPS: a clear
// sphere that holds every remaining vertex of the path ends it whatever its radius).  Once a cube proves that every unknown voxel is
// farther than cap, a value above cap is returned instead of the exact distance — a sparse or empty unknown grid no longer makes every
// vertex of every pair scan the whole lattice.
__device__ inline double nearest_unknown(const UnknownGrid& ug, P3 p, int lane, double cap = INFINITY) {
  if (!ug.flags) return INFINITY;
  const int cx = (int)floor((p.x - ug.ox) / ug.res), cy = (int)floor((p.y - ug.oy) / ug.res), cz = (int)floor((p.z - ug.oz) / ug.res);
  const int wmax = max(max(max(cx, ug.nx - 1 - cx), max(cy, ug.ny - 1 - cy)), max(max(cz, ug.nz - 1 - cz), 0));  // covers the grid
  int w = 1;
  for (;;) {
    const int x0 = max(cx - w, 0), x1 = min(cx + w, ug.nx - 1), y0 = max(cy - w, 0), y1 = min(cy + w, ug.ny - 1);
    const int z0 = max(cz - w, 0), z1 = min(cz + w, ug.nz - 1);
    double best = INFINITY;
    if (x1 >= x0 && y1 >= y0 && z1 >= z0) {
      const int sx = x1 - x0 + 1, sy = y1 - y0 + 1, total = sx * sy * (z1 - z0 + 1);
      for (int idx = lane; idx < total; idx += 64) {
        const int iz = idx / (sx * sy), rem = idx - iz * (sx * sy), iy = rem / sx, ix = rem - iy * sx;
        if (ug.flags[((size_t)(z0 + iz) * ug.ny + (y0 + iy)) * ug.nx + (x0 + ix)]) {
          const double dx = ((double)(x0 + ix) + 0.5) * ug.res + ug.ox - p.x, dy = ((double)(y0 + iy) + 0.5) * ug.res + ug.oy - p.y,
                       dz = ((double)(z0 + iz) + 0.5) * ug.res + ug.oz - p.z;
          const double d2 = dx * dx + dy * dy + dz * dz;
          best = d2 < best ? d2 : best;
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = fmin(best, __shfl_xor(best, o));
    const double d = sqrt(best);
    if (w >= wmax || d <= ((double)w + 0.5) * ug.res) return d;
    if (((double)w + 0.5) * ug.res > cap) return fmin(d, 1e300);  // every unknown voxel is farther than cap: all the caller asks
    // next cube: wide enough to prove the candidate (or twice as wide when there is none yet)
    const int need = best < INFINITY ? (int)ceil(d / ug.res) : 2 * w + 1;
    w = min(max(need, w + 1), wmax);
  }
}

// first point of path[lo .. n) on the sphere around `center` (utils.cpp:782-870); last_inside is relative to lo
__device__ inline P3 sphere_exit(const P3* path, int n, double r, P3 center, int& last_inside, bool& none_outside) {
  none_outside = false;
  int index = -1;
  for (int i = 0; i < n; i++)
    if (dist3(path[i], center) > r) { index = i; break; }
  if (index == -1) {
    last_inside = n - 1;
    none_outside = true;
    return sphere_crossing(center, path[n - 1], r, center);
  }
  if (index == 0) {
    last_inside = 1;
    return path[0];
  }
  last_inside = index - 1;
  return sphere_crossing(path[index - 1], path[index], r, center);
}

// reduceJPSbyDistance: the path ends d before its end
__device__ inline void shorten_by(P3* path, int& n, double d) {
  double acc = 0;
  for (int i = n - 1; i > 0; i--) {
    const P3 v = p3(path[i].x - path[i - 1].x, path[i].y - path[i - 1].y, path[i].z - path[i - 1].z);
    const double len = sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    acc += len;
    if (acc > d) {
      const double keep = acc - d;
      n = i;
      const P3 b = path[n - 1];
      path[n] = p3(b.x + v.x / len * keep, b.y + v.y / len * keep, b.z + v.z / len * keep);
      n++;
      break;
    }
  }
}

// One wavefront per pair.  Writes the safe path (<= max_poly_safe + 1 vertices, R first) and its length (0: no safe trajectory is
// needed, or there is no whole trajectory), x0 = R of the safe problem, the sphere of known space of the pair.
__global__ void __launch_bounds__(64) safe_path_kernel(const fh_problem* __restrict__ whole, const fh_result* __restrict__ wres,
                                                       const double* __restrict__ paths, const int32_t* __restrict__ n_points, int n,
                                                       int max_points, double r_frac, fh_pair_rule rule, int max_poly_safe,
                                                       fh_problem* __restrict__ safe, double* __restrict__ safe_paths,
                                                       int32_t* __restrict__ safe_np, double* __restrict__ spheres, UnknownGrid ug) {
  __shared__ P3 s_orig[SAFE_PATH_CAP + 2], s_cur[SAFE_PATH_CAP + 2];  // (LDS, not per-lane arrays: those would be 2 KB of scratch per lane)
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= n) return;
  const fh_problem& pw = whole[b];
  const fh_result& rw = wres[b];
  const int mp = max_poly_safe + 1;
  int np_out = 0;
  const int np_in = n_points[b];
  if (rw.solved && pw.n_seg >= 1 && pw.n_seg <= FH_MAX_SEG && np_in >= 2 && np_in <= SAFE_PATH_CAP) {
    int k;
    if (choose_r_index(pw, rw, r_frac, rule, lane, k, &ug)) {
      const int N = pw.n_seg;
      const double dt = rw.dt, DC = pw.dc;
      const int size = sample_count(pw, rw);
      double t = 0;
      int interval = 0;
      clock_at(k, DC, dt, N, t, interval);  // the reference's clock at sample k (solverGurobi.cpp:131-135): fh_clock.hpp
      fh_state R;
      eval_state(rw.coeff[interval], t - interval * dt, k == size - 1, R);
      if (lane < 3) {
        safe[b].x0[lane] = lane == 0 ? R.pos[0] : (lane == 1 ? R.pos[1] : R.pos[2]);
        safe[b].x0[3 + lane] = lane == 0 ? R.vel[0] : (lane == 1 ? R.vel[1] : R.vel[2]);
        safe[b].x0[6 + lane] = lane == 0 ? R.accel[0] : (lane == 1 ? R.accel[1] : R.accel[2]);
      }
      {  // a handful of vertices: every lane walks them (wave-uniform), lane 0 writes; the distance query of mode 2 needs the wavefront
        const P3 A = p3(pw.x0[0], pw.x0[1], pw.x0[2]);
        P3* orig = s_orig;
        P3* cur = s_cur;
        int no = np_in, nc = np_in;
        __syncthreads();
        if (lane == 0)
          for (int i = 0; i < np_in; i++) {
            const double* v = paths + 3 * ((size_t)b * max_points + i);
            orig[i] = cur[i] = p3(v[0], v[1], v[2]);
          }
        __syncthreads();
        // getFirstCollisionJPS against unknown space.  Distance to it — mode 2: to the nearest unknown voxel of the caller's grid (no
        // unknown voxel at all: the reference returns the path as it was, :806-812); otherwise modelled: r_known - |p - A|, not below 0
        int iteration = 0;
        while (nc > 0) {
          double r;
          if (rule.mode == 2) {
            double cap = rule.drone_radius;  // below drone_radius the distance decides the cut; beyond the farthest remaining vertex a
                                             // clear sphere holds them all and ends the march (sphere_exit: none_outside)
            for (int i = 1; i < nc; i++) cap = fmax(cap, dist3(cur[i], cur[0]));
            r = nearest_unknown(ug, cur[0], lane, cap);
            if (!(r < INFINITY)) break;
          } else {
            r = rule.r_known - dist3(cur[0], A);
            r = r > 0 ? r : 0;
          }
          __syncthreads();
          if (r < rule.drone_radius) {
            if (lane == 0) {
              if (iteration == 0) {  // already there at the first vertex: the reference returns a 1 cm stub
                orig[1] = p3(orig[0].x + 0.01, orig[0].y, orig[0].z);
                no = 2;
              } else {
                const int eliminated = no - nc + 1;
                no = eliminated;
                orig[no++] = cur[0];
                shorten_by(orig, no, rule.drone_radius);
              }
            }
            no = __shfl(no, 0);
            break;
          }
          bool none_outside;
          int last_id;
          const P3 inters = sphere_exit(cur, nc, r, cur[0], last_id, none_outside);
          if (none_outside) break;  // the rest of the path is known to be clear: the path as it was
          const int drop = last_id + 1;  // erase [0, last_id], insert the intersection in front
          __syncthreads();
          if (lane == 0) {
            for (int i = drop; i < nc; i++) cur[i - drop + 1] = cur[i];
            cur[0] = inters;
          }
          nc = nc - drop + 1;
          __syncthreads();
          iteration++;
        }
        __syncthreads();
        // JPS_safe: R first, at most max_poly_safe legs (:478-490)
        np_out = no < mp ? no : mp;
        if (lane == 0) {
          orig[0] = p3(R.pos[0], R.pos[1], R.pos[2]);
          for (int i = 0; i < np_out; i++) {
            double* o = safe_paths + 3 * ((size_t)b * mp + i);
            o[0] = orig[i].x; o[1] = orig[i].y; o[2] = orig[i].z;
          }
        }
      }
    }
  }
  if (lane == 0) {
    safe_np[b] = np_out;
    spheres[4 * (size_t)b] = pw.x0[0]; spheres[4 * (size_t)b + 1] = pw.x0[1]; spheres[4 * (size_t)b + 2] = pw.x0[2];
    spheres[4 * (size_t)b + 3] = rule.r_known;
  }
}

// per segment of the fixed-stride segment table: the sphere of its pair
__global__ void safe_spheres_kernel(const double* __restrict__ pair_spheres, int n, int max_poly, double* __restrict__ seg_spheres) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * max_poly) return;
  for (int k = 0; k < 4; k++) seg_spheres[4 * t + k] = pair_spheres[4 * (t / max_poly) + k];
}

// The safe problem record of a pair from its corridor: polytope table, xf = M (the last vertex of the safe path) or G when G lies in
// the last polytope (:498-499).  n_seg = 0 marks a pair without a safe problem (none needed / no whole trajectory / no corridor).
__global__ void __launch_bounds__(64) safe_finalize_kernel(const int32_t* __restrict__ safe_np, const double* __restrict__ goal_m,
                                                           const double* __restrict__ goals_g, const fh_face* __restrict__ faces,
                                                           const int32_t* __restrict__ face_off, const int32_t* __restrict__ n_poly, int n,
                                                           int faces_per_problem, int n_seg_safe, fh_problem* __restrict__ safe) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= n) return;
  const int P = n_poly[b];
  if (safe_np[b] < 2 || P < 1) {
    if (lane == 0) safe[b].n_seg = 0;
    return;
  }
  // G inside the last polytope?
  const int f0 = face_off[9 * (size_t)b + P - 1], f1 = face_off[9 * (size_t)b + P];
  const double gx = goals_g[3 * (size_t)b], gy = goals_g[3 * (size_t)b + 1], gz = goals_g[3 * (size_t)b + 2];
  bool out = false;
  for (int f = f0 + lane; f < f1; f += 64) {
    const fh_face fc = faces[(size_t)b * faces_per_problem + f];
    out = out || (fc.a[0] * gx + fc.a[1] * gy + fc.a[2] * gz - fc.b > 0);  // LinearConstraint::inside: A x - b <= 0 for every row
  }
  const bool inside = __ballot(out) == 0ull;
  if (lane < 3) safe[b].xf[lane] = inside ? goals_g[3 * (size_t)b + lane] : goal_m[3 * (size_t)b + lane];
  if (lane <= FH_MAX_POLY) safe[b].face_off[lane] = face_off[9 * (size_t)b + lane];
  if (lane == 0) {
    safe[b].n_seg = n_seg_safe;
    safe[b].n_poly = P;
    safe[b].face_begin = b * faces_per_problem;
  }
}

}  // namespace fh
