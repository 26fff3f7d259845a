// jps_tables.hpp — the neighbour tables of 3-D jump point search, GENERATED from geometric rules (header only; used by the host
// restatement corridor_frontend.cpp and, uploaded once per map, by the device search csrc/fh_path.hip.hpp).
//
// jps3d writes the same tables down case by case in JPS3DNeib::Neib / FNeib (thirdparty/jps3d/src/jps_planner/graph_search.cpp:
// 573-937); their ORDER decides in which sequence successors reach the open list and therefore which of several equal-cost paths is
// found.  tests/test_ref_frontend.py compares what is generated here with the reference's tables entry by entry.
#pragma once
#include <cstdlib>
#include <cstring>

namespace fhfront {

struct JpsTables {
  // per direction id = (dx+1) + 3 (dy+1) + 9 (dz+1): natural neighbours ns, cells to test f1, directions to add when forced f2
  int ns[27][26][3], f1[27][12][3], f2[27][12][3];
};
static const int kJpsCount[4][2] = {{26, 0}, {1, 8}, {3, 12}, {7, 12}};  // by |d|_1: natural neighbours, forced-neighbour entries

inline JpsTables make_jps_tables() {
  JpsTables T;
  std::memset(&T, 0, sizeof(T));
  const int seq[3] = {0, 1, -1};
  for (int dz = -1; dz <= 1; dz++)
    for (int dy = -1; dy <= 1; dy++)
      for (int dx = -1; dx <= 1; dx++) {
        const int id = (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1);
        const int d[3] = {dx, dy, dz};
        const int norm1 = std::abs(dx) + std::abs(dy) + std::abs(dz);
        auto put = [](int (*arr)[3], int k, int x, int y, int z) { arr[k][0] = x; arr[k][1] = y; arr[k][2] = z; };
        if (norm1 == 0) {  // the start node: all 26 neighbours, z = 0 plane first; inside a plane y then x in the order 0, +1, -1
          int k = 0;
          for (int zi = 0; zi < 3; zi++)
            for (int yi = 0; yi < 3; yi++)
              for (int xi = 0; xi < 3; xi++)
                if (seq[xi] || seq[yi] || seq[zi]) put(T.ns[id], k++, seq[xi], seq[yi], seq[zi]);
        } else if (norm1 == 1) {  // straight: the move itself; forced: the 8 cells around the axis, (u, v) in a fixed order
          put(T.ns[id], 0, dx, dy, dz);
          const int uv[8][2] = {{0, 1}, {0, -1}, {1, 0}, {1, 1}, {1, -1}, {-1, 0}, {-1, 1}, {-1, -1}};
          for (int k = 0; k < 8; k++) {
            int f[3];
            if (dz) { f[0] = uv[k][0]; f[1] = uv[k][1]; f[2] = 0; }        // move along z: (u, v) = (x, y)
            else if (dx) { f[0] = 0; f[1] = uv[k][1]; f[2] = uv[k][0]; }   // along x: u -> z, v -> y
            else { f[0] = uv[k][0]; f[1] = 0; f[2] = uv[k][1]; }           // along y: u -> x, v -> z
            put(T.f1[id], k, f[0], f[1], f[2]);
            put(T.f2[id], k, f[0] + dx, f[1] + dy, f[2] + dz);
          }
        } else if (norm1 == 2) {  // diagonal in a plane: in-plane axes p < q (x before y before z), c the axis across the plane
          const int c = dx == 0 ? 0 : (dy == 0 ? 1 : 2);
          const int p_ = c == 0 ? 1 : 0, q_ = c == 2 ? 1 : 2;
          auto vec = [&](int ap, int aq, int ac, int out[3]) { out[0] = out[1] = out[2] = 0; out[p_] = ap; out[q_] = aq; out[c] = ac; };
          int v[3];
          vec(0, d[q_], 0, v); put(T.ns[id], 0, v[0], v[1], v[2]);
          vec(d[p_], 0, 0, v); put(T.ns[id], 1, v[0], v[1], v[2]);
          put(T.ns[id], 2, dx, dy, dz);
          // entries 0-1: in the plane; 2-3: across; 4-7: across and behind one component; 8-11: across, one component ahead
          const int F[12][3] = {{0, -1, 0}, {-1, 0, 0}, {0, 0, 1}, {0, 0, -1}, {0, -1, 1}, {-1, 0, 1}, {0, -1, -1}, {-1, 0, -1},
                                {0, 0, 1}, {0, 0, 1}, {0, 0, -1}, {0, 0, -1}};   // (p, q, c) in units of (d_p, d_q, 1)
          const int G[12][3] = {{1, -1, 0}, {-1, 1, 0}, {1, 1, 1}, {1, 1, -1}, {1, -1, 1}, {-1, 1, 1}, {1, -1, -1}, {-1, 1, -1},
                                {1, 0, 1}, {0, 1, 1}, {1, 0, -1}, {0, 1, -1}};
          for (int k = 0; k < 12; k++) {
            vec(F[k][0] * d[p_], F[k][1] * d[q_], F[k][2], v); put(T.f1[id], k, v[0], v[1], v[2]);
            vec(G[k][0] * d[p_], G[k][1] * d[q_], G[k][2], v); put(T.f2[id], k, v[0], v[1], v[2]);
          }
        } else {  // diagonal in space: the three axis moves, the three plane diagonals, the move itself
          const int M[7][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};
          for (int k = 0; k < 7; k++) put(T.ns[id], k, M[k][0] * dx, M[k][1] * dy, M[k][2] * dz);
          // forced: cells behind one component (0-2), behind two (3-5), and the "extras" (6-11); in units of (dx, dy, dz)
          const int F[12][3] = {{-1, 0, 0}, {0, -1, 0}, {0, 0, -1}, {0, -1, -1}, {-1, 0, -1}, {-1, -1, 0},
                                {-1, 0, 0}, {-1, 0, 0}, {0, -1, 0}, {0, -1, 0}, {0, 0, -1}, {0, 0, -1}};
          const int G[12][3] = {{-1, 1, 1}, {1, -1, 1}, {1, 1, -1}, {1, -1, -1}, {-1, 1, -1}, {-1, -1, 1},
                                {-1, 0, 1}, {-1, 1, 0}, {0, -1, 1}, {1, -1, 0}, {0, 1, -1}, {1, 0, -1}};
          for (int k = 0; k < 12; k++) {
            put(T.f1[id], k, F[k][0] * dx, F[k][1] * dy, F[k][2] * dz);
            put(T.f2[id], k, G[k][0] * dx, G[k][1] * dy, G[k][2] * dz);
          }
        }
      }
  return T;
}
inline const JpsTables& jps_tables() {
  static const JpsTables T = make_jps_tables();
  return T;
}


}  // namespace fhfront
