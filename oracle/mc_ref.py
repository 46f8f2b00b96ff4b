"""Oracle: numpy restatement of the iso-surface extraction + surface sampling step
(csrc/marching_cubes.hip).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference step being replaced: utils/eval_3D.py:233-263 - ``mcubes.marching_cubes(level, 0.5)``
(pymcubes==0.1.4), vertex rescale ``v / S * (max - min) + min`` with S = G (:252-255),
``trimesh.Trimesh(...).sample(num_points)`` (trimesh==4.0.8, area-weighted, numpy global RNG).
Both packages are un-vendored dependencies that are NOT installable here -> **parity
unpinned**: this file restates the published algorithm (classic marching cubes on the
Bourke corner/edge numbering with a case bit set when value < iso; linear interpolation on
edges; area-weighted triangle choice + uniform barycentric sample with reflection) on the
build-owned case tables of zeroshape_amd/mc_tables.py, with the device code's exact
operation order so triangles can be compared bit for bit.  What it cannot pin: PyMCubes'
choice inside ambiguous cubes and trimesh's random stream (the reference's sampled cloud is
not reproducible run to run either).
"""
import numpy as np

from zeroshape_amd import mc_tables as T

_EA = np.array([0, 1, 3, 0, 4, 5, 7, 4, 0, 1, 2, 3])   # low-coordinate endpoint of each edge
_EB = np.array([1, 2, 2, 3, 5, 6, 6, 7, 4, 5, 6, 7])
_AXIS = np.array([0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2])
M64 = (1 << 64) - 1


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
    corners = np.asarray(T.CORNERS)
    f = np.stack([vol[c[0]:c[0] + C, c[1]:c[1] + C, c[2]:c[2] + C] for c in corners], -1)       # [C,C,C,8]
    case = np.zeros((C, C, C), np.int64)
    for b in range(8):
        case |= (f[..., b] < iso).astype(np.int64) << b
    count = np.asarray(T.TRI_COUNT)[case]
    ii, jj, kk = np.nonzero(count)                       # C order: x slowest, z fastest
    if len(ii) == 0:
        return np.zeros((0, 3, 3), np.float32)
    fc = f[ii, jj, kk]                                   # [n,8]
    cs = case[ii, jj, kk]
    e = np.asarray(T.TRI_TABLE)[cs][:, :15].astype(np.int64)          # [n,15] edge ids (-1 beyond the count)
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
                f = np.array([vol[i + c[0], j + c[1], k + c[2]] for c in T.CORNERS], np.float32)
                case = int(sum(1 << b for b in range(8) if f[b] < iso))
                for t in range(T.TRI_COUNT[case]):
                    tri = np.zeros((3, 3), np.float32)
                    for v in range(3):
                        e = int(T.TRI_TABLE[case, 3 * t + v])
                        a, b, ax = _EA[e], _EB[e], _AXIS[e]
                        tt = np.float32(iso - f[a]) / np.float32(f[b] - f[a])
                        p = np.array([i, j, k], np.float32) + T.CORNERS[a].astype(np.float32)
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
