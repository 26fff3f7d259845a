"""Pure-Python restatement of FASTER's voxel map and path search.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: jps3d needs Eigen/Boost (absent here) and its tests assert nothing, so there is nothing of the
reference to diff against.  This module is an independent implementation — Python lists, heapq, IEEE doubles — of what
faster_amd/host/corridor_frontend.{hpp,cpp} (the C++ restatement the device kernels are compared with) computes:
  * MapUtil::readMap (faster/include/read_map.hpp:30-185): grid geometry with the reference's integer truncations, every point marks
    its cell and the cube of +-floor(inflation/res) cells around it (flat index test only);
  * JPS_Manager::solveJPS3D (faster/src/jps_manager.cpp:141-200): z clamped to >= 0, cells around start and goal freed, optimal
    26-connected path (A* with Euclidean step costs, the exact empty-grid heuristic and the strict total order
    (f quantised to 2^-20 cells, squared distance to the goal, cell index) on the open list), jps3d's clean-up
    (jps_planner.cpp:36-105, :286-291), ends forced onto the requested points.
With a total order on the open list the expansion sequence does not depend on the container, and all arithmetic is +, *, /, sqrt on
doubles: the C++ restatement must return the same vertices and the same number of expanded cells, bit for bit.
"""
import heapq
import math

KEY_SCALE = 1048576.0


def _round_half_away(v):  # std::round
    return int(math.copysign(math.floor(abs(v) + 0.5), v))


class Grid:
    def __init__(self, cloud, cells, res, center, z_ground, z_max, inflation):
        self.res = res
        dx = cells[0] + int(5 * inflation / res)
        dy = cells[1] + int(5 * inflation / res)
        dz = cells[2]
        down = int(dz / 2.0)
        up = int(dz / 2.0)
        if center[2] - res * dz / 2.0 < z_ground:
            down = max(int((center[2] - z_ground) / res), 0)
        if center[2] + res * dz / 2.0 > z_max:
            up = int((z_max - center[2]) / res)
            up = up if up > 0 else 1
        dz = down + up
        self.nx, self.ny, self.nz = dx, dy, dz
        self.origin = (center[0] - res * dx / 2.0, center[1] - res * dy / 2.0, center[2] - res * down)
        total = dx * dy * dz
        self.occ = bytearray(total)
        m = int(math.floor(inflation / res))
        self.m = m
        # setFreeVoxelAndSurroundings(center, const float d) (jps3d map_util.h:248-263): round(d / res + 0.5) cells, d a FLOAT
        import struct
        d32 = struct.unpack("f", struct.pack("f", inflation))[0]
        self.m_free = _round_half_away(d32 / res + 0.5)
        for p in cloud:
            c = [max(v, 0) for v in self.to_cell(p)]
            for ix in range(c[0] - m, c[0] + m + 1):
                for iy in range(c[1] - m, c[1] + m + 1):
                    for iz in range(c[2] - m, c[2] + m + 1):
                        i = ix + dx * iy + dx * dy * iz
                        if 0 <= i < total:
                            self.occ[i] = 100

    def to_cell(self, p):
        return [_round_half_away((p[k] - self.origin[k]) / self.res - 0.5) for k in range(3)]

    def center(self, c):
        return tuple((c[k] + 0.5) * self.res + self.origin[k] for k in range(3))

    def outside(self, x, y, z):
        return x < 0 or y < 0 or z < 0 or x >= self.nx or y >= self.ny or z >= self.nz

    def index(self, x, y, z):
        return x + self.nx * y + self.nx * self.ny * z

    def decode(self, i):
        z, rem = divmod(i, self.nx * self.ny)
        y, x = divmod(rem, self.nx)
        return x, y, z


def plan(grid, start, goal):
    """-> (vertices or None, number of expanded cells)"""
    start = (start[0], start[1], max(start[2], 0.0))
    goal = (goal[0], goal[1], max(goal[2], 0.0))
    s, t = grid.to_cell(start), grid.to_cell(goal)
    freed = set()
    for c in (s, t):
        for ix in range(c[0] - grid.m_free, c[0] + grid.m_free + 1):
            for iy in range(c[1] - grid.m_free, c[1] + grid.m_free + 1):
                for iz in range(c[2] - grid.m_free, c[2] + grid.m_free + 1):
                    if not grid.outside(ix, iy, iz):
                        freed.add(grid.index(ix, iy, iz))

    def occupied(i):
        return grid.occ[i] != 0 and i not in freed

    if grid.outside(*s) or grid.outside(*t):
        return None, 0
    sid, tid = grid.index(*s), grid.index(*t)
    s2, s3 = math.sqrt(2.0), math.sqrt(3.0)

    def heur(x, y, z):
        a, b, c = sorted((abs(x - t[0]), abs(y - t[1]), abs(z - t[2])), reverse=True)
        return float(c) * s3 + float(b - c) * s2 + float(a - b)

    def dist2(x, y, z):
        return (x - t[0]) ** 2 + (y - t[1]) ** 2 + (z - t[2]) ** 2

    g = {sid: 0.0}
    parent = {sid: -1}
    closed = set()
    f0 = 0.0 + heur(*s)
    if f0 >= 2040.0:
        return None, 0
    heap = [(int(f0 * KEY_SCALE), dist2(*s), sid)]
    expansions = 0
    found = False
    while heap:
        _, _, cur = heapq.heappop(heap)
        if cur in closed:
            continue
        closed.add(cur)
        if cur == tid:
            found = True
            break
        expansions += 1
        gc = g[cur]
        cx, cy, cz = grid.decode(cur)
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    if dx == 0 and dy == 0 and dz == 0:
                        continue
                    x, y, z = cx + dx, cy + dy, cz + dz
                    if grid.outside(x, y, z):
                        continue
                    i = grid.index(x, y, z)
                    if occupied(i) or i in closed:
                        continue
                    ng = gc + math.sqrt(float(dx * dx + dy * dy + dz * dz))
                    if ng < g.get(i, math.inf):
                        g[i] = ng
                        parent[i] = cur
                        f = ng + heur(x, y, z)
                        if f >= 2040.0:
                            return None, expansions
                        heapq.heappush(heap, (int(f * KEY_SCALE), dist2(x, y, z), i))
    if not found:
        return None, expansions
    raw = []
    i = tid
    while i >= 0:
        raw.append(grid.center(grid.decode(i)))
        if i == sid:
            break
        i = parent[i]
    raw.reverse()

    def blocked(a, b):
        d = [b[k] - a[k] for k in range(3)]
        mx = max(abs(d[0]), abs(d[1]), abs(d[2])) / grid.res
        steps = int(mx / 0.8)
        if steps <= 0:
            return False
        sc = 1.0 / steps
        for n in range(1, steps):
            c = grid.to_cell([a[k] + (d[k] * sc) * n for k in range(3)])   # rayTrace: pt1 + (diff * s) * n, in this order
            if grid.outside(*c):
                break
            if occupied(grid.index(*c)):
                return True
        return False

    def norm(a, b):
        x, y, z = a[0] - b[0], a[1] - b[1], a[2] - b[2]
        return math.sqrt(x * x + y * y + z * z)

    def remove_line_points(path):
        if len(path) < 3:
            return list(path)
        out = [path[0]]
        for i in range(1, len(path) - 1):
            q = [(path[i + 1][k] - path[i][k]) - (path[i][k] - path[i - 1][k]) for k in range(3)]
            if abs(q[0]) + abs(q[1]) + abs(q[2]) > 1e-2:
                out.append(path[i])
        out.append(path[-1])
        return out

    def remove_corner_points(path):
        if len(path) < 2:
            return list(path)
        prev = path[0]
        out = [prev]
        c1 = math.inf if blocked(path[0], path[1]) else norm(path[0], path[1])
        for i in range(1, len(path) - 1):
            a, b = path[i], path[i + 1]
            c2 = math.inf if blocked(a, b) else norm(a, b)
            c3 = math.inf if blocked(prev, b) else norm(prev, b)
            if c3 < c1 + c2:
                c1 = c3
            else:
                out.append(a)
                c1 = norm(a, b)
                prev = a
        out.append(path[-1])
        return out

    p = remove_corner_points(remove_line_points(raw))
    p.reverse()
    p = remove_corner_points(p)
    p.reverse()
    if len(p) > 1:
        p[0], p[-1] = start, goal
    else:
        p = [start, goal]
    return p, expansions
