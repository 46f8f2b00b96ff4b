"""Oracle: numpy restatement of the iso-surface extraction + surface sampling step
(csrc/marching_cubes.hip).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference step being replaced: utils/eval_3D.py:233-263 - ``mcubes.marching_cubes(level, 0.5)``
(pymcubes==0.1.4), vertex rescale ``v / S * (max - min) + min`` with S = G (:252-255),
``trimesh.Trimesh(...).sample(num_points)`` (trimesh==4.0.8, area-weighted, numpy global RNG).
Both packages are un-vendored dependencies that are NOT installable here -> **parity
unpinned**: this file restates the published algorithm (classic marching cubes on the
Bourke corner/edge numbering with a case bit set when value < iso; linear interpolation on
edges; area-weighted triangle choice + uniform barycentric sample with reflection) with the device code's exact
operation order so triangles can be compared bit for bit.  What it cannot pin: PyMCubes'
choice inside ambiguous cubes and trimesh's random stream (the reference's sampled cloud is
not reproducible run to run either).

The 256 case tables are the ORACLE'S OWN (round 6; until round 5 this file imported the product's
zeroshape_amd/mc_tables.py, so a wrong table entry was invisible to every comparison):
  * ``build_case_tables()`` derives them GEOMETRICALLY - faces are found from corner coordinates, their corners ordered
    by angle, the contour segments of a face come from the marching-squares rule on its four corner states (two
    diagonal inside corners are each cut off by their own segment), a segment's direction from a cross product
    (inside on the left, seen from outside the cube), the loops from linking segments over shared cube edges; each loop
    is fan-triangulated from its lowest edge id - none of the product generator's face cycles or run walking;
  * ``verify_case_tables(table, count)`` checks ANY table exhaustively without generating one: every triangle corner
    is a sign-change edge, no directed half-edge twice, every unmatched half-edge is exactly one of the face's directed
    contour segments and every such segment is there (the mesh of a cube is a manifold whose boundary is the face
    contours, consistently oriented), and for all 256 x 3 x 16 (cube, axis, neighbour) pairs the two cubes put the
    same segments, reversed, on their shared face: crack-free for every sign pattern, not on a sphere.
tests/test_oracle_mc.py holds the product's tables to both, and shows that changing any single entry of any row fails.
"""
import numpy as np

# the published numbering (Bourke / PyMCubes): corner i of a cube, the two corners of edge e
CORNERS = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0],
                    [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], np.int32)
_EA = np.array([0, 1, 3, 0, 4, 5, 7, 4, 0, 1, 2, 3])   # low-coordinate endpoint of each edge
_EB = np.array([1, 2, 2, 3, 5, 6, 6, 7, 4, 5, 6, 7])
_AXIS = np.array([0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2])
M64 = (1 << 64) - 1
MAX_TRIS = 5


# ----------------------------------------------------------------------------- #
# case tables: the oracle's own derivation and its table-independent verifier
# ----------------------------------------------------------------------------- #
def _edge_of(a, b):
    for e in range(12):
        if {int(_EA[e]), int(_EB[e])} == {int(a), int(b)}:
            return e
    raise KeyError((a, b))


def _faces():
    """[(axis, side, outward normal, corners in cyclic order)] from the corner coordinates alone."""
    out = []
    for axis in range(3):
        for side in (0, 1):
            n = np.zeros(3)
            n[axis] = 1.0 if side else -1.0
            cs = [c for c in range(8) if CORNERS[c][axis] == side]
            ctr = CORNERS[cs].mean(0)
            u = (CORNERS[cs[0]] - ctr).astype(np.float64)
            v = np.cross(n, u)                                    # counter-clockwise about the outward normal
            cs.sort(key=lambda c: np.arctan2(np.dot(CORNERS[c] - ctr, v), np.dot(CORNERS[c] - ctr, u)))
            out.append((axis, side, n, cs))
    return out


_FACES = _faces()


def _mid(e):
    return 0.5 * (CORNERS[_EA[e]] + CORNERS[_EB[e]])


def face_segments(case, face):
    """Directed contour segments [(edge_from, edge_to)] of one face: marching squares on its four corner states, diagonal
    inside corners cut off one by one; inside (value < iso, bit set) on the LEFT of from -> to, seen from outside."""
    axis, side, n, cs = face
    ins = [(case >> c) & 1 for c in cs]
    cross = [i for i in range(4) if ins[i] != ins[(i + 1) % 4]]           # boundary edge i joins cs[i], cs[i+1]
    if not cross:
        return []
    pairs = []
    if len(cross) == 2:
        q = next(cs[i] for i in range(4) if ins[i])
        pairs.append((cross[0], cross[1], q))
    else:                                                               # alternating states: each inside corner alone
        for i in range(4):
            if ins[i]:
                pairs.append(((i - 1) % 4, i, cs[i]))                   # the two boundary edges that meet at cs[i]
    segs = []
    for i, j, q in pairs:
        a, b = _edge_of(cs[i], cs[(i + 1) % 4]), _edge_of(cs[j], cs[(j + 1) % 4])
        pa, pb = _mid(a), _mid(b)
        if np.dot(np.cross(pb - pa, CORNERS[q] - pa), n) < 0:           # q must be on the left
            a, b = b, a
        segs.append((a, b))
    return segs


def case_segments(case):
    return [s for f in _FACES for s in face_segments(case, f)]


def build_case_tables():
    """-> (tri_table int8 [256, 15], tri_count int32 [256]) - see the module docstring."""
    table = -np.ones((256, 3 * MAX_TRIS), np.int8)
    count = np.zeros(256, np.int32)
    for case in range(256):
        nxt = {}
        for a, b in case_segments(case):
            assert a not in nxt, "two contour segments leave edge %d in case %d" % (a, case)
            nxt[a] = b
        assert sorted(nxt) == sorted(nxt.values())
        tris, seen = [], set()
        for start in sorted(nxt):
            if start in seen:
                continue
            loop, e = [], start
            while e not in seen:
                seen.add(e)
                loop.append(e)
                e = nxt[e]
            assert e == start and len(loop) >= 3
            tris += [(loop[0], loop[k], loop[k + 1]) for k in range(1, len(loop) - 1)]
        assert len(tris) <= MAX_TRIS
        count[case] = len(tris)
        table[case, :3 * len(tris)] = np.array(tris, np.int8).reshape(-1)
    return table, count


TRI_TABLE, TRI_COUNT = build_case_tables()


def _boundary_half_edges(row, n_tris):
    """(errors, unmatched directed half-edges) of the triangles of one table row."""
    errs, half = [], {}
    for t in range(n_tris):
        tri = [int(x) for x in row[3 * t:3 * t + 3]]
        if len(set(tri)) != 3:
            errs.append("triangle %d is degenerate %s" % (t, tri))
        for k in range(3):
            h = (tri[k], tri[(k + 1) % 3])
            if h in half:
                errs.append("directed half-edge %s twice" % (h,))
            half[h] = t
    return errs, {h for h in half if (h[1], h[0]) not in half}


def verify_case(case, row, n_tris):
    """Errors of ONE row (empty list = the row is a consistently oriented manifold mesh whose boundary is exactly the
    contour of the six faces)."""
    row = np.asarray(row).astype(np.int64)
    errs = []
    if not (0 <= n_tris <= MAX_TRIS):
        return ["triangle count %d" % n_tris]
    if np.any(row[3 * n_tris:] != -1):
        errs.append("entries beyond the count are not -1")
    used = row[:3 * n_tris]
    crossing = {e for e in range(12) if ((case >> int(_EA[e])) & 1) != ((case >> int(_EB[e])) & 1)}
    if np.any(used < 0) or np.any(used > 11):
        return errs + ["edge id out of range"]
    if set(used.tolist()) != crossing:
        errs.append("vertices %s are not the sign-change edges %s" % (sorted(set(used.tolist())), sorted(crossing)))
    e2, boundary = _boundary_half_edges(row, n_tris)
    errs += e2
    want = set(case_segments(case))
    if boundary != want:
        errs.append("mesh boundary %s is not the face contour %s" % (sorted(boundary), sorted(want)))
    return errs


def _shift_edge(e, axis):
    """The edge of the -axis face of the NEXT cube that coincides with edge e of this cube's +axis face."""
    a, b = CORNERS[_EA[e]].copy(), CORNERS[_EB[e]].copy()
    assert a[axis] == 1 and b[axis] == 1
    a[axis] = b[axis] = 0
    ca = next(c for c in range(8) if np.array_equal(CORNERS[c], a))
    cb = next(c for c in range(8) if np.array_equal(CORNERS[c], b))
    return _edge_of(ca, cb)


def verify_case_tables(table, count):
    """Exhaustive check of a whole table (module docstring).  -> list of error strings, empty when the table is sound."""
    table, count = np.asarray(table), np.asarray(count)
    errs = []
    if table.shape != (256, 3 * MAX_TRIS) or count.shape != (256,):
        return ["table shapes %s %s" % (table.shape, count.shape)]
    bnd = []
    for case in range(256):
        errs += ["case %d: %s" % (case, m) for m in verify_case(case, table[case], int(count[case]))]
        bnd.append(_boundary_half_edges(table[case].astype(np.int64), int(count[case]))[1])
    # neighbours: cube `c` and the next cube along `axis` share c's +axis face; whatever the other four corners of either cube
    for axis in range(3):
        plus = [c for c in range(8) if CORNERS[c][axis] == 1]
        on_plus = {e for e in range(12) if CORNERS[_EA[e]][axis] == 1 and CORNERS[_EB[e]][axis] == 1}
        on_minus = {e for e in range(12) if CORNERS[_EA[e]][axis] == 0 and CORNERS[_EB[e]][axis] == 0}
        twin = {c: next(k for k in range(8) if np.array_equal(CORNERS[k] + np.eye(3, dtype=np.int32)[axis], CORNERS[c]))
                for c in plus}                                   # corner of the next cube that IS corner c of this one
        shift = {e: _shift_edge(e, axis) for e in on_plus}
        for c in range(256):
            mine = {(shift[b], shift[a]) for a, b in bnd[c] if a in on_plus and b in on_plus}      # reversed: seen from the other side
            fixed = sum(((c >> k) & 1) << twin[k] for k in plus)
            free = [k for k in range(8) if CORNERS[k][axis] == 1]
            for other in range(16):
                c2 = fixed | sum(((other >> i) & 1) << free[i] for i in range(4))
                theirs = {(a, b) for a, b in bnd[c2] if a in on_minus and b in on_minus}
                if mine != theirs:
                    errs.append("cases %d | %d disagree on their shared face along axis %d: %s vs %s"
                                % (c, c2, axis, sorted(mine), sorted(theirs)))
    return errs


def _fma32(a, b, c):
    return (np.float64(a) * np.float64(b) + np.float64(c)).astype(np.float32)


def marching_cubes(vol, iso, scale, offset):
    """vol [G,G,G] float32 (x slowest) -> triangle soup [n,3,3] float32, world space, in
    cube order (x slowest, z fastest) and table order inside a cube.  All cubes at once in numpy (the same fp32
    operations in the same order as marching_cubes_loop, which tests/test_oracle_mc.py holds it to bit for bit):
    the vox-64 / vox-128 grids of BASELINE configs 2 / 3 take seconds instead of hours."""
    vol = np.asarray(vol, np.float32)
    G = vol.shape[0]
    C = G - 1
    iso = np.float32(iso)
    scale, offset = np.float32(scale), np.float32(offset)
    corners = np.asarray(CORNERS)
    f = np.stack([vol[c[0]:c[0] + C, c[1]:c[1] + C, c[2]:c[2] + C] for c in corners], -1)       # [C,C,C,8]
    case = np.zeros((C, C, C), np.int64)
    for b in range(8):
        case |= (f[..., b] < iso).astype(np.int64) << b
    count = np.asarray(TRI_COUNT)[case]
    ii, jj, kk = np.nonzero(count)                       # C order: x slowest, z fastest
    if len(ii) == 0:
        return np.zeros((0, 3, 3), np.float32)
    fc = f[ii, jj, kk]                                   # [n,8]
    cs = case[ii, jj, kk]
    e = np.asarray(TRI_TABLE)[cs][:, :15].astype(np.int64)          # [n,15] edge ids (-1 beyond the count)
    live = np.arange(15)[None, :] < 3 * count[ii, jj, kk][:, None]
    e = np.where(live, e, 0)
    a, b, ax = _EA[e], _EB[e], _AXIS[e]
    rows = np.arange(len(ii))[:, None]
    fa, fb = fc[rows, a], fc[rows, b]
    with np.errstate(divide="ignore", invalid="ignore"):
        tt = (iso - fa).astype(np.float32) / (fb - fa).astype(np.float32)
    base = np.stack([ii, jj, kk], -1).astype(np.float32)[:, None, :] + corners[a].astype(np.float32)     # [n,15,3]
    add = np.zeros_like(base)
    add[rows, np.arange(15)[None, :], ax] = tt
    p = (base + add).astype(np.float32)                  # only the edge's axis moves; + 0 is exact elsewhere
    world = _fma32(p, scale, offset)
    return world[live].reshape(-1, 3, 3)


def marching_cubes_loop(vol, iso, scale, offset):
    """The same extraction cube by cube (the first form of this oracle; kept as the check of the vectorised one)."""
    vol = np.asarray(vol, np.float32)
    G = vol.shape[0]
    C = G - 1
    iso = np.float32(iso)
    scale, offset = np.float32(scale), np.float32(offset)
    tris = []
    for i in range(C):
        for j in range(C):
            for k in range(C):
                f = np.array([vol[i + c[0], j + c[1], k + c[2]] for c in CORNERS], np.float32)
                case = int(sum(1 << b for b in range(8) if f[b] < iso))
                for t in range(TRI_COUNT[case]):
                    tri = np.zeros((3, 3), np.float32)
                    for v in range(3):
                        e = int(TRI_TABLE[case, 3 * t + v])
                        a, b, ax = _EA[e], _EB[e], _AXIS[e]
                        tt = np.float32(iso - f[a]) / np.float32(f[b] - f[a])
                        p = np.array([i, j, k], np.float32) + CORNERS[a].astype(np.float32)
                        p[ax] = np.float32(p[ax] + tt)
                        tri[v] = _fma32(p, scale, offset)
                    tris.append(tri)
    return np.stack(tris) if tris else np.zeros((0, 3, 3), np.float32)


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)


def u01(seed, ctr):
    return np.float32((_splitmix64(seed ^ _splitmix64(ctr)) >> 40) * (1.0 / 16777216.0))


def triangle_areas(tris):
    p = tris.astype(np.float32)
    u, v = p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]
    nx = u[:, 1] * v[:, 2] - u[:, 2] * v[:, 1]
    ny = u[:, 2] * v[:, 0] - u[:, 0] * v[:, 2]
    nz = u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]
    return 0.5 * np.sqrt(nx.astype(np.float64) ** 2 + ny.astype(np.float64) ** 2 + nz.astype(np.float64) ** 2)


def _u01_many(seed, ctr):
    """u01 for an array of counters (uint64 arithmetic wraps like the & M64 above)."""
    def mix(x):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))
    with np.errstate(over="ignore"):
        h = mix(np.uint64(seed) ^ mix(np.asarray(ctr, np.uint64)))
    return ((h >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)).astype(np.float32)


def sample_surface(tris, n_samples, seed):
    """area-weighted samples [n_samples,3] float32 + chosen triangle ids (all samples at once; sample_surface_loop is the
    sample-by-sample form it is checked against)."""
    if len(tris) == 0:
        return np.zeros((n_samples, 3), np.float32), np.zeros(n_samples, np.int64)
    cum = np.cumsum(triangle_areas(tris))
    s = np.arange(n_samples, dtype=np.uint64)
    target = _u01_many(seed, 3 * s).astype(np.float64) * cum[-1]
    ids = np.minimum(np.searchsorted(cum, target, side="right"), len(tris) - 1).astype(np.int64)
    r1, r2 = _u01_many(seed, 3 * s + np.uint64(1)), _u01_many(seed, 3 * s + np.uint64(2))
    flip = (r1 + r2).astype(np.float32) > np.float32(1.0)
    r1 = np.where(flip, np.float32(1.0) - r1, r1).astype(np.float32)
    r2 = np.where(flip, np.float32(1.0) - r2, r2).astype(np.float32)
    p = tris[ids].astype(np.float32)
    pts = _fma32(r2[:, None], p[:, 2] - p[:, 0], _fma32(r1[:, None], p[:, 1] - p[:, 0], p[:, 0]))
    return pts, ids


def sample_surface_loop(tris, n_samples, seed):
    """area-weighted samples [n_samples,3] float32 + chosen triangle ids."""
    if len(tris) == 0:
        return np.zeros((n_samples, 3), np.float32), np.zeros(n_samples, np.int64)
    cum = np.cumsum(triangle_areas(tris))
    pts = np.zeros((n_samples, 3), np.float32)
    ids = np.zeros(n_samples, np.int64)
    for s in range(n_samples):
        target = np.float64(u01(seed, 3 * s)) * cum[-1]
        t = min(int(np.searchsorted(cum, target, side="right")), len(tris) - 1)
        r1, r2 = u01(seed, 3 * s + 1), u01(seed, 3 * s + 2)
        if np.float32(r1 + r2) > np.float32(1.0):
            r1, r2 = np.float32(1.0) - r1, np.float32(1.0) - r2
        p = tris[t]
        pts[s] = _fma32(r2, p[2] - p[0], _fma32(r1, p[1] - p[0], p[0]))
        ids[s] = t
    return pts, ids
