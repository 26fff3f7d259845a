// fh_decomp.hip.hpp — batched convex decomposition around path segments on the device (gfx950, wave64, FP64).
//
// GPU version of next-row N1: what JPS_Manager::cvxEllipsoidDecomp (faster/src/jps_manager.cpp:80-127) computes per
// path segment through DecompUtil's LineSegment3D::dilate — obstacle inflation + ellipsoid fit (line_segment.h:156-252), separating
// planes (decomp_base.h:83-115, ellipsoid.h:48-73), local bounding box (line_segment.h:57-98), conversion to A x <= b around the segment
// midpoint (polyhedron.h:131-152) and the ground plane (jps_manager.cpp:113-124).  Same arithmetic as the host version
// (faster_amd/host/corridor_frontend.hpp), one wavefront per segment:
//   * one coalesced sweep over the obstacle cloud keeps the points inside the segment's local box (ballot + mbcnt compaction into an
//     LDS list, already inflated towards the ellipsoid centre);
//   * every "closest point" of the reference is a lane-parallel arg-min over that list on the DPP network; every "drop the points the
//     new plane/ellipsoid excludes" is a lane-parallel flag update.  The loops themselves (shrink the axes, add planes) are the
//     reference's and stay sequential per segment; thousands of segments run side by side.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/fasterhip.h"
#include "fh_solve.hip.hpp"  // wave reductions
#include "fh_udiv.hpp"

// every decision of the decomposition (which point is closest, on which side of a plane a point lies) must fall as on the host:
// the same products and sums, no fused multiply-adds
#pragma clang fp contract(off)

namespace fh {

#define FH_DECOMP_CAP 256  // points inside the local box whose (inflated) COORDINATES fit the LDS list of a segment
                            // (256: 10.5 KB of LDS, 12 workgroups per CU at 165 VGPRs — with 1024 and 4 per CU the same launches took 1.6x as long)
#define FH_DECOMP_CAP_IDS 1728  // ... longer lists are kept in the SAME LDS as 4-byte ids (a cloud index, or the packed cell of an unknown
                            // voxel) + a flag byte, and a point is rebuilt — loaded or enumerated, inflated — whenever it is looked at: the
                            // lists of the safe corridor (2-3 k unknown voxels per local box) used to live in the workgroup's HBM workspace
                            // and were rewritten there after every separating plane (3-5 GB of writes per dispatch).  (1728: with the
                            // candidate-block list the workgroup's LDS is 12 736 B, ten granules of 1280 B: 12 workgroups per CU.)
                            // [r6] A list longer than that keeps its first FH_DECOMP_CAP_IDS entries in LDS and the REST in the workspace
                            // (HybridList): with the caller's unknown voxels 55 % of the replan workload's segments hold 1537-4000 points, and
                            // their whole lists used to live in the workspace, filled by a second sweep.
#define FH_DECOMP_EPS 1e-10  // DecompUtil's epsilon_

struct D3 {
  double x, y, z;
};
__device__ __forceinline__ D3 d3(double x, double y, double z) { D3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return d3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ D3 operator*(D3 a, double s) { return d3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double norm(D3 a) { return sqrt(dot(a, a)); }

struct Rot {  // row major
  double m[3][3];
};
__device__ __forceinline__ D3 mul(const Rot& R, D3 v) {
  return d3(R.m[0][0] * v.x + R.m[0][1] * v.y + R.m[0][2] * v.z, R.m[1][0] * v.x + R.m[1][1] * v.y + R.m[1][2] * v.z,
            R.m[2][0] * v.x + R.m[2][1] * v.y + R.m[2][2] * v.z);
}
__device__ __forceinline__ D3 mulT(const Rot& R, D3 v) {
  return d3(R.m[0][0] * v.x + R.m[1][0] * v.y + R.m[2][0] * v.z, R.m[0][1] * v.x + R.m[1][1] * v.y + R.m[2][1] * v.z,
            R.m[0][2] * v.x + R.m[1][2] * v.y + R.m[2][2] * v.z);
}
// Rz(yaw) Ry(pitch), zero roll (geometric_utils.h:27-35); cosines and sines straight from the components, exactly as the host
// restatement (corridor_frontend.hpp: rotation_onto): only correctly rounded operations, so both sides agree bit for bit
__device__ __forceinline__ Rot rot_onto(D3 v) {
  const double hxy = sqrt(v.x * v.x + v.y * v.y), n3 = sqrt(hxy * hxy + v.z * v.z);
  const double cp = n3 > 0 ? hxy / n3 : 1.0, sp = n3 > 0 ? -v.z / n3 : 0.0;
  const double cy = hxy > 0 ? v.x / hxy : 1.0, sy = hxy > 0 ? v.y / hxy : 0.0;
  Rot r;
  r.m[0][0] = cy * cp; r.m[0][1] = -sy; r.m[0][2] = cy * sp;
  r.m[1][0] = sy * cp; r.m[1][1] = cy;  r.m[1][2] = sy * sp;
  r.m[2][0] = -sp;     r.m[2][1] = 0;   r.m[2][2] = cp;
  return r;
}
__device__ __forceinline__ Rot rot_roll(const Rot& Ri, double z, double y) {  // Ri * Rx(atan2(z, y))
  const double h = sqrt(y * y + z * z);
  const double c = h > 0 ? y / h : 1.0, s = h > 0 ? z / h : 0.0;
  Rot r;
  for (int i = 0; i < 3; i++) {
    r.m[i][0] = Ri.m[i][0];
    r.m[i][1] = Ri.m[i][1] * c + Ri.m[i][2] * s;
    r.m[i][2] = -Ri.m[i][1] * s + Ri.m[i][2] * c;
  }
  return r;
}
__device__ __forceinline__ double ell_dist(const Rot& R, D3 ax, D3 c, D3 q) {
  const D3 l = mulT(R, q - c);
  return norm(d3(l.x / ax.x, l.y / ax.y, l.z / ax.z));
}
__device__ __forceinline__ int sgn_i(double v) { return (0.0 < v) - (v < 0.0); }

// arg-min of the ellipsoid distance over the list entries whose flag has `bit`; returns the list index (-1 if none)
#ifdef FHD_EXPERIMENT
__constant__ int fhd_stop_after;
__device__ unsigned long long fhd_hist[8];  // segments whose list has <= 256, <= 1536, <= 16384, more points; sum of the counts; cells swept
#endif
template <class L>
__device__ __forceinline__ int closest_in(const L& list, int cnt, unsigned char bit, const Rot& R, D3 ax, D3 c, int lane) {
  double best = INFINITY;
  int bi = -1;
  for (int i = lane; i < cnt; i += 64)
    if (list.flag(i) & bit) {
      const double dd = ell_dist(R, ax, c, list.get(i));
      if (dd < best) { best = dd; bi = i; }
    }
  const double mn = wave_min(best);
  if (!(mn < INFINITY)) {  // empty set, or every distance is NaN: any member (the host version does the same)
    int anyi = -1;
    for (int i = lane; i < cnt && anyi < 0; i += 64)
      if (list.flag(i) & bit) anyi = i;
    const int l2 = first_lane(anyi >= 0);
    return l2 >= 0 ? __builtin_amdgcn_readlane(anyi, l2) : -1;
  }
  // exact ties (mirror-symmetric points of a regular cloud): the lowest list index wins, as in the host's sequential scan
  int v = best == mn ? bi : 0x7fffffff;
#define FH_DPP_MIN_I32(ctrl, rmask)                                                   \
  {                                                                                   \
    const int o = __builtin_amdgcn_update_dpp(0x7fffffff, v, ctrl, rmask, 0xf, false); \
    v = o < v ? o : v;                                                                \
  }
  FH_DPP_MIN_I32(0x111, 0xf) FH_DPP_MIN_I32(0x112, 0xf) FH_DPP_MIN_I32(0x114, 0xf) FH_DPP_MIN_I32(0x118, 0xf)
  FH_DPP_MIN_I32(0x142, 0xa) FH_DPP_MIN_I32(0x143, 0xc)
#undef FH_DPP_MIN_I32
  return __builtin_amdgcn_readlane(v, 63);
}

// Unknown space as the mapper would report it — one point per voxel it has never seen — MODELLED for a batch of independent
// problems: the voxels of a grid (cell centres) that lie farther than a radius from a point (the vehicle: it has seen what its
// sensor reaches and nothing else).  Nothing is stored: a segment enumerates the cells of the grid inside the bounding box of its
// local box, z-major, x fastest — the order in which a cloud holding ALL such voxels would list them.
// With `flags` (fh_set_unknown_grid_device, rule mode 2) the unknown voxels are the caller's: cell (ix, iy, iz) is unknown iff
// flags[(iz ny + iy) nx + ix] != 0 — the mapper's unknown cloud as FASTER feeds it to cvxEllipsoidDecomp together with the occupied
// points (jps_manager.cpp:91-98); the sphere is not used then.
struct UnknownLattice {
  double ox, oy, oz, res;
  int nx, ny, nz, on;
  const unsigned char* flags;
};
struct LatticeRange {
  int x0, cx, y0, cy, z0, cz, total;  // sub-block of the grid: first cell and count per axis
  unsigned inv_cxy, inv_cx;           // floor(2^32 / (cx cy)), floor(2^32 / cx): a cell number is split with two multiply-highs (fhu::div, fh_udiv.hpp)
  double ax, ay, az, r2;              // the sphere of known space
};
__device__ __forceinline__ LatticeRange lattice_range(const UnknownLattice& lat, const double lo[3], const double hi[3], const double* sphere) {
  LatticeRange g;
  g.total = 0; g.x0 = g.y0 = g.z0 = g.cx = g.cy = g.cz = 0;
  g.inv_cxy = g.inv_cx = 0xffffffffu;
  g.ax = g.ay = g.az = g.r2 = 0;
  if (!lat.on || (!sphere && !lat.flags)) return g;
  auto first = [](double v, double o, double res, int n) { int i = (int)floor((v - o) / res) - 1; return i < 0 ? 0 : (i > n ? n : i); };
  auto last = [](double v, double o, double res, int n) { int i = (int)floor((v - o) / res) + 1; return i > n - 1 ? n - 1 : i; };
  g.x0 = first(lo[0], lat.ox, lat.res, lat.nx); g.cx = last(hi[0], lat.ox, lat.res, lat.nx) - g.x0 + 1;
  g.y0 = first(lo[1], lat.oy, lat.res, lat.ny); g.cy = last(hi[1], lat.oy, lat.res, lat.ny) - g.y0 + 1;
  g.z0 = first(lo[2], lat.oz, lat.res, lat.nz); g.cz = last(hi[2], lat.oz, lat.res, lat.nz) - g.z0 + 1;
  if (g.cx <= 0 || g.cy <= 0 || g.cz <= 0) return g;
  const long long cells = (long long)g.cx * g.cy * g.cz;
  g.total = (cells > (1ll << 28) || g.cx > 1024 || g.cy > 1024 || g.cz > 1024) ? -1 : (int)cells;  // -1: a grid far too fine for the local box — the segment reports failure (count -1)
  if (sphere) { g.ax = sphere[0]; g.ay = sphere[1]; g.az = sphere[2]; g.r2 = sphere[3] * sphere[3]; }
  g.inv_cxy = fhu::inverse_fp(g.cx * g.cy);
  g.inv_cx = fhu::inverse_fp(g.cx);
  return g;
}
// centre of the cell (ix, iy, iz) of the sub-block; `packed` = iz << 20 | iy << 10 | ix (a sub-block has at most 1024 cells per axis)
__device__ __forceinline__ D3 lattice_centre(const UnknownLattice& lat, const LatticeRange& g, int packed) {
  const int ix = packed & 1023, iy = (packed >> 10) & 1023, iz = (packed >> 20) & 1023;
  return d3(((double)(g.x0 + ix) + 0.5) * lat.res + lat.ox, ((double)(g.y0 + iy) + 0.5) * lat.res + lat.oy, ((double)(g.z0 + iz) + 0.5) * lat.res + lat.oz);
}
// cell number idx of the sub-block: its centre, its packed coordinates, and whether it is an unknown voxel
__device__ __forceinline__ bool lattice_point(const UnknownLattice& lat, const LatticeRange& g, int idx, D3& q, int* packed = nullptr) {
  if (idx >= g.total) return false;
  const int iz = fhu::div(idx, g.cx * g.cy, g.inv_cxy), rem = idx - iz * (g.cx * g.cy), iy = fhu::div(rem, g.cx, g.inv_cx), ix = rem - iy * g.cx;
  if (packed) *packed = (iz << 20) | (iy << 10) | ix;
  q = d3(((double)(g.x0 + ix) + 0.5) * lat.res + lat.ox, ((double)(g.y0 + iy) + 0.5) * lat.res + lat.oy, ((double)(g.z0 + iz) + 0.5) * lat.res + lat.oz);
  if (lat.flags) return lat.flags[((size_t)(g.z0 + iz) * lat.ny + (g.y0 + iy)) * lat.nx + (g.x0 + ix)] != 0;
  const double dx = q.x - g.ax, dy = q.y - g.ay, dz = q.z - g.az;
  return dx * dx + dy * dy + dz * dz > g.r2;
}

#define FH_DECOMP_BLIST 1024  // candidate blocks per segment (more: full sweep)
#define FH_DECOMP_CAP_GLOBAL 16384  // list capacity with the tail of the (id) list in the HBM workspace (dense clouds); more => count = -1
#define FH_DECOMP_WS_DOUBLES (FH_DECOMP_CAP_GLOBAL / 2 + FH_DECOMP_CAP_GLOBAL / 8)  // workspace of a workgroup: 4-byte ids, then flag bytes

// ---- what a segment's list of box points is made of ----
// CoordList: the inflated coordinates themselves (short lists: a look is three LDS loads).
struct CoordList {
  double *px, *py, *pz;
  unsigned char* fl;
  typedef D3 Entry;
  __device__ __forceinline__ D3 get(int i) const { return d3(px[i], py[i], pz[i]); }
  __device__ __forceinline__ Entry entry(int i) const { return d3(px[i], py[i], pz[i]); }
  __device__ __forceinline__ D3 point(const Entry& e) const { return e; }
  __device__ __forceinline__ unsigned char flag(int i) const { return fl[i]; }
  __device__ __forceinline__ void set_flag(int i, unsigned char f) const { fl[i] = f; }
  __device__ __forceinline__ void store(int pos, const Entry& e, unsigned char f) const { px[pos] = e.x; py[pos] = e.y; pz[pos] = e.z; fl[pos] = f; }
};
// IdList: which point it is — cloud index (>= 0) or 0x80000000 | packed cell of the unknown lattice — rebuilt at every look with the
// arithmetic that CoordList applied once (the same doubles).
struct IdList {
  int* id;
  unsigned char* fl;
  const double* cloud;
  const UnknownLattice* lat;
  const LatticeRange* lr;
  Rot Ri;
  D3 c;
  double inflate;
  typedef int Entry;
  __device__ __forceinline__ D3 point(const Entry& e) const {
    const D3 q = e < 0 ? lattice_centre(*lat, *lr, e & 0x7fffffff) : d3(cloud[3 * (size_t)e], cloud[3 * (size_t)e + 1], cloud[3 * (size_t)e + 2]);
    const D3 l = mulT(Ri, q - c);
    return mul(Ri, d3(l.x - sgn_i(l.x) * inflate, l.y - sgn_i(l.y) * inflate, l.z - sgn_i(l.z) * inflate)) + c;
  }
  __device__ __forceinline__ D3 get(int i) const { return point(id[i]); }
  __device__ __forceinline__ Entry entry(int i) const { return id[i]; }
  __device__ __forceinline__ unsigned char flag(int i) const { return fl[i]; }
  __device__ __forceinline__ void set_flag(int i, unsigned char f) const { fl[i] = f; }
  __device__ __forceinline__ void store(int pos, const Entry& e, unsigned char f) const { id[pos] = e; fl[pos] = f; }
};

// HybridList: an id list whose entries [0, FH_DECOMP_CAP_IDS) are in LDS and the others in the workgroup's HBM workspace.  Every loop over
// a list walks it in blocks of 64 that start at a multiple of 64, so a block lies on one side (the choice is wave-uniform); the
// compaction after a separating plane moves entries towards the front, i.e. into LDS.
struct HybridList : IdList {
  int* gid;
  unsigned char* gfl;
  __device__ __forceinline__ Entry entry(int i) const {
    int v;
    if (i < FH_DECOMP_CAP_IDS) v = id[i]; else v = gid[i - FH_DECOMP_CAP_IDS];
    return v;
  }
  __device__ __forceinline__ D3 get(int i) const { return point(entry(i)); }
  __device__ __forceinline__ unsigned char flag(int i) const {
    unsigned char v;
    if (i < FH_DECOMP_CAP_IDS) v = fl[i]; else v = gfl[i - FH_DECOMP_CAP_IDS];
    return v;
  }
  __device__ __forceinline__ void set_flag(int i, unsigned char f) const {
    if (i < FH_DECOMP_CAP_IDS) fl[i] = f; else gfl[i - FH_DECOMP_CAP_IDS] = f;
  }
  __device__ __forceinline__ void store(int pos, const Entry& e, unsigned char f) const {
    if (pos < FH_DECOMP_CAP_IDS) { id[pos] = e; fl[pos] = f; }
    else { gid[pos - FH_DECOMP_CAP_IDS] = e; gfl[pos - FH_DECOMP_CAP_IDS] = f; }
  }
};

// The decomposition of one segment with its list of box points in `list` (CoordList / IdList in LDS, HybridList with its tail in the
// per-workgroup HBM workspace for longer lists).  prebuilt: the caller's sweep has put that many points of the box into the
// list.
template <class L>
__device__ void decomp_segment(const L& list, const double* __restrict__ cloud, int n_cloud, D3 p1, D3 p2, const D3* bp,
                               const D3* bn, double inflate, double z_ground, int max_faces, fh_face* __restrict__ out,
                               int32_t* __restrict__ count_out, int lane, const int* blist, int nb, const UnknownLattice& lat,
                               const LatticeRange& lrange, int prebuilt) {
  const D3 dvec = p2 - p1;
  const double f = norm(dvec) / 2;
  const Rot Ri = rot_onto(dvec);
  const D3 c = (p1 + p2) * 0.5;
  int cnt = prebuilt;  // (the caller's sweep has listed the points of the box: ids, or inflated coordinates)
  __syncthreads();
#ifdef FHD_EXPERIMENT
  if (lat.on && fhd_stop_after == 2) {
    if (lane == 0) *count_out = cnt ? 0 : 0;
    return;
  }
#endif

  // ---- ellipsoid fit (line_segment.h:156-252)
  D3 axes = d3(f, f, f);
  for (int i = lane; i < cnt; i += 64) {
    const bool fi = ell_dist(Ri, axes, c, list.get(i)) <= 1.0;
    list.set_flag(i, (unsigned char)((fi ? 3 : 0) | 4));  // first, inside; remain
  }
  __syncthreads();
  Rot Rf = Ri;
  D3 cur = axes;
  for (int guard = 0; guard <= cnt; guard++) {  // second axis
    const int ic = closest_in(list, cnt, 2, Rf, cur, c, lane);
    if (ic < 0) break;
    const D3 pw = list.get(ic);
    D3 l = mulT(Ri, pw - c);
    Rf = rot_roll(Ri, l.z, l.y);
    l = mulT(Rf, pw - c);
    if (l.x < axes.x) axes.y = fabs(l.y) / sqrt(1 - (l.x / axes.x) * (l.x / axes.x));
    cur = d3(axes.x, axes.y, axes.y);
    for (int i = lane; i < cnt; i += 64) {
      const unsigned char fl = list.flag(i);
      if ((fl & 2) && !(1 - ell_dist(Rf, cur, c, list.get(i)) > FH_DECOMP_EPS)) list.set_flag(i, (unsigned char)(fl & ~2));
    }
    __syncthreads();
  }
  for (int i = lane; i < cnt; i += 64) {  // third axis back to its initial length; restart from the first set
    const unsigned char fl = list.flag(i);
    const bool in2 = (fl & 1) && ell_dist(Rf, axes, c, list.get(i)) <= 1.0;
    list.set_flag(i, (unsigned char)((fl & ~2) | (in2 ? 2 : 0)));
  }
  __syncthreads();
  for (int guard = 0; guard <= cnt; guard++) {
    const int ic = closest_in(list, cnt, 2, Rf, axes, c, lane);
    if (ic < 0) break;
    const D3 l = mulT(Rf, list.get(ic) - c);
    const double dd = 1 - (l.x / axes.x) * (l.x / axes.x) - (l.y / axes.y) * (l.y / axes.y);
    if (dd > FH_DECOMP_EPS) axes.z = fabs(l.z) / sqrt(dd);
    for (int i = lane; i < cnt; i += 64) {
      const unsigned char fl = list.flag(i);
      if ((fl & 2) && !(1 - ell_dist(Rf, axes, c, list.get(i)) > FH_DECOMP_EPS)) list.set_flag(i, (unsigned char)(fl & ~2));
    }
    __syncthreads();
  }

#ifdef FHD_EXPERIMENT
  if (lat.on && fhd_stop_after == 3) {
    if (lane == 0) *count_out = (axes.z > 0) ? 0 : 0;
    return;
  }
#endif
  // ---- separating planes (decomp_base.h:83-115), written straight as rows oriented around the midpoint (polyhedron.h:131-152)
  int rows = 0;
  bool too_many = false;
  auto emit = [&](D3 p, D3 n) {
    double off = dot(p, n);
    if (dot(n, c) - off > 0) { n = n * -1.0; off = -off; }
    if (rows < max_faces) {
      if (lane == 0) { out[rows].a[0] = n.x; out[rows].a[1] = n.y; out[rows].a[2] = n.z; out[rows].b = off; }
    } else too_many = true;
    rows++;
  };
  const int cnt_planes = cnt;  // (every plane puts at least the point it passes through away)
  for (int guard = 0; guard <= cnt_planes; guard++) {
    const int ic = closest_in(list, cnt, 4, Rf, axes, c, lane);
    if (ic < 0) break;
    const D3 cp = list.get(ic);
    const D3 l = mulT(Rf, cp - c);
    const D3 g = mul(Rf, d3(l.x / (axes.x * axes.x), l.y / (axes.y * axes.y), l.z / (axes.z * axes.z)));
    const double gn = norm(g);
    if (!(gn > 0) || !isfinite(gn)) break;  // degenerate ellipsoid: no separating planes (as the host version)
    const D3 n = d3(g.x / gn, g.y / gn, g.z / gn);
    emit(cp, n);
    // the points the plane puts away are never looked at again: the list is compacted in place (order kept — the tie rule of
    // closest_in is the lowest index), so that the scans of the following planes get shorter
    int alive = 0;
    for (int i0 = 0; i0 < cnt; i0 += 64) {
      const int i = i0 + lane;
      bool stay = false;
      typename L::Entry e = typename L::Entry();
      unsigned char fl = 0;
      if (i < cnt) {
        fl = list.flag(i);
        e = list.entry(i);
        stay = (fl & 4) && (dot(n, list.point(e) - cp) < 0);
      }
      const unsigned long long m = __ballot(stay);
      __syncthreads();  // (every entry of this block of 64 has been read before any of them is overwritten)
      if (stay) list.store(alive + __popcll(m & ((1ull << lane) - 1ull)), e, fl);
      alive += __popcll(m);
      __syncthreads();
    }
    cnt = alive;
  }
#pragma unroll
  for (int k = 0; k < 6; k++) emit(bp[k], bn[k]);
  if (rows < max_faces) {  // ground plane -z <= -z_ground (jps_manager.cpp:113-124)
    if (lane == 0) { out[rows].a[0] = 0; out[rows].a[1] = 0; out[rows].a[2] = -1; out[rows].b = -z_ground; }
  } else too_many = true;
  rows++;
  if (lane == 0) *count_out = too_many ? -1 : rows;
}

// Bounding boxes of the blocks of 64 consecutive cloud points: blocks[b] = (min x, y, z, max x, y, z).  One wavefront per block.
__global__ void __launch_bounds__(64) cloud_blocks_kernel(const double* __restrict__ cloud, int n_cloud, double* __restrict__ blocks) {
  const int b = (int)blockIdx.x, lane = (int)threadIdx.x, i = b * 64 + lane;
  double v[3] = {INFINITY, INFINITY, INFINITY}, w[3] = {INFINITY, INFINITY, INFINITY};  // w: negated, so that one min reduction does both
  if (i < n_cloud)
    for (int k = 0; k < 3; k++) { v[k] = cloud[3 * i + k]; w[k] = -v[k]; }
  for (int k = 0; k < 3; k++) {
    const double mn = wave_min(v[k]), mx = -wave_min(w[k]);
    if (lane == 0) { blocks[6 * (size_t)b + k] = mn; blocks[6 * (size_t)b + 3 + k] = mx; }
  }
}

// Persistent workgroups of one wavefront; segments are drawn from a counter (`ticket`, zeroed by the caller) — a segment's cost goes
// with the number of points in its box (a few to 1500 and more with unknown voxels), so a fixed stride leaves the launch waiting
// for the unluckiest workgroup.  segments: [n][6] = p1, p2.  faces: [n][max_faces] rows
// (a, b); counts[n] = rows written, -1 on overflow.  workspace: per workgroup FH_DECOMP_WS_DOUBLES doubles (the tail of a long list: ids, flags).
__global__ void __launch_bounds__(64, 3) decomp_kernel(const double* __restrict__ cloud, int n_cloud, const double* __restrict__ segments,
                                                    int n_segments, double bx, double by, double bz, double inflate, double z_ground,
                                                    int max_faces, double* __restrict__ workspace, fh_face* __restrict__ faces,
                                                    int32_t* __restrict__ counts, const double* __restrict__ blocks, UnknownLattice lat,
                                                    const double* __restrict__ spheres, int* __restrict__ ticket) {
  // the segment's list of box points: 256 inflated points (3 x 256 doubles + 256 flag bytes) or, in the same bytes, 1536 ids + flag bytes.
  // flags: bit0 first (inside the initial sphere), bit1 inside (current loop), bit2 remain
  static_assert(FH_DECOMP_CAP_IDS % 64 == 0 && FH_DECOMP_CAP <= FH_DECOMP_CAP_IDS, "the id list aliases the coordinate list; blocks of 64 do not straddle its end");
  constexpr int LIST_DOUBLES = (FH_DECOMP_CAP_IDS / 2 > 3 * FH_DECOMP_CAP) ? FH_DECOMP_CAP_IDS / 2 : 3 * FH_DECOMP_CAP;  // ids or coordinates
  __shared__ double lraw[LIST_DOUBLES + FH_DECOMP_CAP_IDS / 8];
  __shared__ int lblist[FH_DECOMP_BLIST];         // blocks of the cloud that can touch the local box, ascending
  const int lane = threadIdx.x;
  double* gws = workspace + (size_t)blockIdx.x * (size_t)FH_DECOMP_WS_DOUBLES;  // (ids + flags behind the LDS part of a long list)
  for (;;) {
    __syncthreads();
    int seg = 0;
    if (lane == 0) seg = atomicAdd(ticket, 1);
    seg = __builtin_amdgcn_readfirstlane(seg);
    if (seg >= n_segments) break;
    const D3 p1 = d3(segments[6 * seg + 0], segments[6 * seg + 1], segments[6 * seg + 2]);
    const D3 p2 = d3(segments[6 * seg + 3], segments[6 * seg + 4], segments[6 * seg + 5]);
    fh_face* out = faces + (size_t)seg * (size_t)max_faces;
    if (p1.x != p1.x) {  // NaN: an unused slot of a fixed-stride segment table (fh_corridor_batch_device)
      if (lane == 0) counts[seg] = 0;
      continue;
    }
    // local bounding box (line_segment.h:57-98), plane order kept: +h, -h, +dir, -dir, +v, -v
    const D3 dvec = p2 - p1;
    const double dn = norm(dvec);
    const D3 dir = d3(dvec.x / dn, dvec.y / dn, dvec.z / dn);
    D3 dh = d3(dir.y, -dir.x, 0.0);
    if (norm(dh) == 0) dh = d3(-1, 0, 0);
    const double hn = norm(dh);
    dh = d3(dh.x / hn, dh.y / hn, dh.z / hn);
    const D3 dv = d3(dir.y * dh.z - dir.z * dh.y, dir.z * dh.x - dir.x * dh.z, dir.x * dh.y - dir.y * dh.x);
    D3 bp[6], bn[6];
    bp[0] = p1 + dh * by; bn[0] = dh;
    bp[1] = p1 - dh * by; bn[1] = dh * -1.0;
    bp[2] = p2 + dir * bx; bn[2] = dir;
    bp[3] = p1 - dir * bx; bn[3] = dir * -1.0;
    bp[4] = p1 + dv * bz; bn[4] = dv;
    bp[5] = p1 - dv * bz; bn[5] = dv * -1.0;
    // the candidates of the sweep.  Blocks of 64 consecutive cloud points whose bounding box (blocks: cloud_blocks_kernel) misses the bounding box of the local box
    // hold no point of interest: a mapper's cloud is spatially coherent, so most blocks are skipped.  The candidates are visited in
    // cloud order, so the list of points — and with it every tie rule — is the one of the full sweep.
    int nb = -1;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
#pragma unroll
    for (int corner = 0; corner < 8; corner++) {
      const D3 e = (corner & 1 ? p2 + dir * bx : p1 - dir * bx) + dh * (corner & 2 ? by : -by) + dv * (corner & 4 ? bz : -bz);
      lo[0] = fmin(lo[0], e.x); lo[1] = fmin(lo[1], e.y); lo[2] = fmin(lo[2], e.z);
      hi[0] = fmax(hi[0], e.x); hi[1] = fmax(hi[1], e.y); hi[2] = fmax(hi[2], e.z);
    }
    const LatticeRange lrange = lattice_range(lat, lo, hi, spheres ? spheres + 4 * (size_t)seg : nullptr);
    if (lrange.total < 0) {
      if (lane == 0) counts[seg] = -1;
      continue;
    }
    if (blocks) {
      const int n_blocks = (n_cloud + 63) / 64;
      nb = 0;
      for (int b0 = 0; b0 < n_blocks && nb >= 0; b0 += 64) {
        const int b = b0 + lane;
        bool hit = false;
        if (b < n_blocks) {
          const double* bb = blocks + 6 * (size_t)b;
          hit = bb[0] <= hi[0] + 1e-6 && bb[3] >= lo[0] - 1e-6 && bb[1] <= hi[1] + 1e-6 && bb[4] >= lo[1] - 1e-6 && bb[2] <= hi[2] + 1e-6 &&
                bb[5] >= lo[2] - 1e-6;
        }
        const unsigned long long hm = __ballot(hit);
        const int k = (int)__popcll(hm);
        if (nb + k > FH_DECOMP_BLIST) nb = -1;  // too many candidates: the full sweep
        else {
          if (hit) lblist[nb + (int)__popcll(hm & ((1ull << lane) - 1ull))] = b;
          nb += k;
        }
      }
      __syncthreads();
    }
    // ONE sweep over the candidates: the points of the box are listed as ids (a cloud index, or the packed cell of an unknown voxel)
    // in LDS, up to FH_DECOMP_CAP_IDS of them, behind that in the workspace (up to FH_DECOMP_CAP_GLOBAL in all), and counted beyond that.
    // (Round 3 swept twice — count, then store: the count decided where the list lives; rounds 4-5 swept a list that did not fit LDS
    // again into the workspace.)  A short list is then turned into coordinates in place.
    int* lids = reinterpret_cast<int*>(lraw);
    unsigned char* lflags = reinterpret_cast<unsigned char*>(lraw + LIST_DOUBLES);
    int* gids = reinterpret_cast<int*>(gws);  // the tail of a list longer than FH_DECOMP_CAP_IDS: ids, and flags behind them
    unsigned char* gflags = reinterpret_cast<unsigned char*>(gws + FH_DECOMP_CAP_GLOBAL / 2);
    int cnt = 0;
    // [r6] The six plane tests of the local box (:57-98 with epsilon_) decide as the host's do — but only a candidate within `band` of a
    // plane needs them: the box is |u| <= by, |w| <= bz, -bx <= d <= L + bx in the frame (dh, dir, dv) at p1, a candidate farther than
    // `band` outside one of the pairs fails the exact test of that pair, one farther than `band` inside all of them passes all six
    // (`band` is orders of magnitude above the rounding of either form and above epsilon_).  The sweep of the unknown lattice tests
    // ~11 000 cells per segment: three dot products instead of six, and the exact tests for the rare trip that has a candidate in the band.
    const double band = 1e-6 * (1.0 + fabs(p1.x) + fabs(p1.y) + fabs(p1.z) + fabs(p2.x) + fabs(p2.y) + fabs(p2.z) + bx + by + bz);
    auto note = [&](bool in, D3 q, int ident) {
      {
        const D3 r = q - p1;
        const double u = fabs(dot(dh, r)), w = fabs(dot(dv, r)), d = dot(dir, r);
        const bool sure_out = u > by + band || w > bz + band || d > dn + bx + band || d < -bx - band;
        const bool sure_in = u < by - band && w < bz - band && d < dn + bx - band && d > -bx + band;
        in = in && !sure_out;
        if (__ballot(in && !sure_in)) {  // (a NaN coordinate lands here too and is decided by the exact tests)
#pragma unroll
          for (int k = 0; k < 6; k++) in = in && !(dot(bn[k], q - bp[k]) > FH_DECOMP_EPS);
        }
      }
      const unsigned long long m = __ballot(in);
      if (m) {
        const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
        if (in) {
          if (pos < FH_DECOMP_CAP_IDS) lids[pos] = ident;
          else if (pos < FH_DECOMP_CAP_GLOBAL) gids[pos - FH_DECOMP_CAP_IDS] = ident;
        }
        cnt += __popcll(m);
      }
    };
    if (lat.flags) {
      // [r6] the caller's unknown voxels: a cell's flag is a byte in memory, and a trip of 64 cells that waits for its own load is a memory
      // round trip per trip.  Four trips' flags are requested together; the cells are then noted in the same order as before (the list —
      // and with it every tie rule — is unchanged).  (Measured: 8.9 -> 8.5 ms for the safe corridors of the replan workload.)
      const int cxy = lrange.cx * lrange.cy;
      for (int base = 0; base < lrange.total; base += 256) {
        int packed[4];
        unsigned char fl[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int idx = base + 64 * u + lane;
          const bool live = idx < lrange.total;
          const int ic = live ? idx : 0;
          const int iz = fhu::div(ic, cxy, lrange.inv_cxy), rem = ic - iz * cxy, iy = fhu::div(rem, lrange.cx, lrange.inv_cx), ix = rem - iy * lrange.cx;
          packed[u] = live ? ((iz << 20) | (iy << 10) | ix) : -1;
          fl[u] = lat.flags[((size_t)(lrange.z0 + iz) * lat.ny + (lrange.y0 + iy)) * lat.nx + (lrange.x0 + ix)];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          if (base + 64 * u >= lrange.total) break;
          const bool in = packed[u] >= 0 && fl[u] != 0;
          note(in, lattice_centre(lat, lrange, packed[u] & 0x3fffffff), (int)(0x80000000u | (unsigned)(packed[u] & 0x3fffffff)));
        }
      }
    } else {
      for (int base = 0; base < lrange.total; base += 64) {
        D3 q = d3(0, 0, 0);
        int packed = 0;
        const bool in = lattice_point(lat, lrange, base + lane, q, &packed);
        note(in, q, (int)(0x80000000u | (unsigned)packed));
      }
    }
    const int n_sweep = nb >= 0 ? nb : (n_cloud + 63) / 64;
    for (int j = 0; j < n_sweep; j++) {
      const int base = (nb >= 0 ? lblist[j] : j) * 64;
      const int i = base + lane;
      bool in = false;
      D3 q = d3(0, 0, 0);
      if (i < n_cloud) {
        q = d3(cloud[3 * i], cloud[3 * i + 1], cloud[3 * i + 2]);
        in = true;
      }
      note(in, q, i);
    }
    __syncthreads();
#ifdef FHD_EXPERIMENT  // (timing experiments only: what a launch with unknown voxels costs up to here; FHD_STOP_AFTER in the environment)
    if (lane == 0) {
      atomicAdd(&fhd_hist[cnt <= FH_DECOMP_CAP ? 0 : (cnt <= FH_DECOMP_CAP_IDS ? 1 : (cnt <= FH_DECOMP_CAP_GLOBAL ? 2 : 3))], 1ull);
      atomicAdd(&fhd_hist[4], (unsigned long long)cnt);
      atomicAdd(&fhd_hist[5], (unsigned long long)lrange.total);
    }
    if (lat.on && fhd_stop_after == 1) {
      if (lane == 0) counts[seg] = cnt ? 0 : 0;
      continue;
    }
#endif
    const Rot Ri0 = rot_onto(p2 - p1);
    const D3 c0 = (p1 + p2) * 0.5;
    IdList I;
    I.id = lids; I.fl = lflags;
    I.cloud = cloud; I.lat = &lat; I.lr = &lrange; I.Ri = Ri0; I.c = c0; I.inflate = inflate;
    if (cnt <= FH_DECOMP_CAP) {
      // ids -> inflated coordinates, in the same bytes: every lane takes its ids out first (IdList::point is the arithmetic the
      // coordinate list used to apply while sweeping: the same doubles)
      CoordList L;
      L.px = lraw; L.py = lraw + FH_DECOMP_CAP; L.pz = lraw + 2 * FH_DECOMP_CAP;
      L.fl = lflags;
      int mine[FH_DECOMP_CAP / 64];
#pragma unroll
      for (int k = 0; k < FH_DECOMP_CAP / 64; k++) mine[k] = (64 * k + lane < cnt) ? lids[64 * k + lane] : 0;
      __syncthreads();
#pragma unroll
      for (int k = 0; k < FH_DECOMP_CAP / 64; k++)
        if (64 * k + lane < cnt) {
          const D3 pt = I.point(mine[k]);
          L.px[64 * k + lane] = pt.x; L.py[64 * k + lane] = pt.y; L.pz[64 * k + lane] = pt.z;
        }
      __syncthreads();
      decomp_segment(L, cloud, n_cloud, p1, p2, bp, bn, inflate, z_ground, max_faces, out, &counts[seg], lane, lblist, nb, lat, lrange, cnt);
    } else if (cnt <= FH_DECOMP_CAP_IDS) {
      decomp_segment(I, cloud, n_cloud, p1, p2, bp, bn, inflate, z_ground, max_faces, out, &counts[seg], lane, lblist, nb, lat, lrange, cnt);
    } else if (cnt <= FH_DECOMP_CAP_GLOBAL) {
      HybridList H;
      H.id = lids; H.fl = lflags; H.gid = gids; H.gfl = gflags;
      H.cloud = cloud; H.lat = &lat; H.lr = &lrange; H.Ri = Ri0; H.c = c0; H.inflate = inflate;
      decomp_segment(H, cloud, n_cloud, p1, p2, bp, bn, inflate, z_ground, max_faces, out, &counts[seg], lane, lblist, nb, lat, lrange, cnt);
    }
    else if (lane == 0)
      counts[seg] = -1;
  }
}

}  // namespace fh
