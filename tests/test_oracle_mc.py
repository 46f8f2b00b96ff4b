"""Host checks of the marching-cubes tables and the numpy oracle (no GPU)."""
import numpy as np

from oracle import mc_ref as M
from zeroshape_amd import mc_tables as T


def test_tables_are_consistent():
    assert T.MAX_TRIS == 5 and T.TRI_COUNT[0] == 0 and T.TRI_COUNT[255] == 0
    for c in range(256):
        ins = [(c >> i) & 1 for i in range(8)]
        cross = {e for e, (a, b) in enumerate(T.EDGES) if ins[a] != ins[b]}
        used = set(T.TRI_TABLE[c][T.TRI_TABLE[c] >= 0].tolist())
        assert cross == used                       # every sign change is meshed, nothing else
        assert (T.TRI_TABLE[c] >= 0).sum() == 3 * T.TRI_COUNT[c]


def test_oracle_derives_its_own_tables_and_they_are_the_products():
    """oracle/mc_ref.py no longer imports the product's tables: it derives the 256 cases itself
    (geometric faces, marching squares per face, cross-product orientation) - and arrives at the same tables, entry for
    entry, so every bit-for-bit triangle comparison of the GPU kernel now runs against independently derived cases."""
    src = open(M.__file__.replace(".pyc", ".py")).read()
    assert "import mc_tables" not in src and "from zeroshape_amd" not in src
    np.testing.assert_array_equal(M.TRI_TABLE, T.TRI_TABLE)
    np.testing.assert_array_equal(M.TRI_COUNT, T.TRI_COUNT)
    np.testing.assert_array_equal(M.CORNERS, T.CORNERS)
    assert [tuple(sorted((int(a), int(b)))) for a, b in zip(M._EA, M._EB)] == [tuple(sorted(e)) for e in T.EDGES.tolist()]


def test_tables_are_crack_free_for_every_sign_pattern():
    """The table-independent verifier on the PRODUCT's tables: per case a consistently oriented manifold whose boundary is
    exactly the six faces' contour segments; for all 256 x 3 x 16 neighbour pairs the shared face carries the same
    segments reversed - crack-freeness and winding for every sign pattern, exhaustively (not on a sphere)."""
    assert M.verify_case_tables(T.TRI_TABLE, T.TRI_COUNT) == []
    # the face rule itself (case 1 = only corner 0 has its bit set, i.e. a value BELOW the iso level): the three segments
    # circle corner 0 and the triangle's normal points towards it - towards decreasing values, which for an occupancy grid
    # (occ > 0.5 inside the object) is out of the object
    segs = M.case_segments(1)
    assert sorted(e for s in segs for e in s) == [0, 0, 3, 3, 8, 8]
    tri = T.TRI_TABLE[1][:3]
    p = np.array([M._mid(int(e)) for e in tri])
    normal = np.cross(p[1] - p[0], p[2] - p[0])
    assert np.dot(normal, M.CORNERS[0] - p.mean(0)) > 0


def test_any_single_table_entry_perturbed_is_caught():
    """Every single-entry change of every row (each of the 11 other edge ids, and -1), every swap of two entries of a
    triangle (flipped winding), a dropped or duplicated triangle: verify_case reports it.  (What a row may change without
    being wrong - which diagonal a loop's fan uses, the order of its triangles - is pinned by the table equality above.)"""
    tried = 0
    for case in range(256):
        n = int(T.TRI_COUNT[case])
        base = T.TRI_TABLE[case].astype(np.int64)
        assert M.verify_case(case, base, n) == []
        for pos in range(3 * n):
            for val in list(range(12)) + [-1]:
                if val == base[pos]:
                    continue
                row = base.copy()
                row[pos] = val
                assert M.verify_case(case, row, n), (case, pos, val)
                tried += 1
        for t in range(n):
            row = base.copy()
            row[3 * t], row[3 * t + 1] = base[3 * t + 1], base[3 * t]
            assert M.verify_case(case, row, n), (case, t, "winding")
            row = base.copy()                                            # triangle t removed, the rest moved up
            row[3 * t:3 * n - 3] = base[3 * t + 3:3 * n]
            row[3 * n - 3:3 * n] = -1
            assert M.verify_case(case, row, n - 1), (case, t, "dropped")
            if n < M.MAX_TRIS:
                row = base.copy()
                row[3 * n:3 * n + 3] = base[3 * t:3 * t + 3]
                assert M.verify_case(case, row, n + 1), (case, t, "duplicated")
        if n:
            assert M.verify_case(case, base, n - 1) and M.verify_case(case ^ 1, base, n)
    assert tried > 25000
    # ... and a whole-table change that keeps every row valid by itself but breaks a shared face cannot exist under this
    # check: a row's boundary is pinned to the face rule, and the rule reads the four shared corners only (the neighbour
    # sweep of verify_case_tables confirms it on the table itself)
    bad = T.TRI_TABLE.copy()
    bad[1, :3] = bad[1, [1, 0, 2]]
    errs = M.verify_case_tables(bad, T.TRI_COUNT)
    assert any(e.startswith("case 1:") for e in errs) and any("disagree on their shared face" in e for e in errs)


def _sphere(G, r, c=None):
    ax = np.linspace(-1.5, 1.5, G, dtype=np.float32)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    c = c or (0.1, -0.05, 0.2)
    d = np.sqrt((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2)
    return (1.0 / (1.0 + np.exp((d - r) * 20.0))).astype(np.float32), ax     # >0.5 inside, like occ


def test_sphere_mesh_is_closed_and_accurate():
    G = 17
    vol, ax = _sphere(G, 0.8)
    # unit-scale world transform (index -> linspace coordinate) to check geometry
    step = float(ax[1] - ax[0])
    tris = M.marching_cubes(vol, 0.5, step, float(ax[0]))
    assert len(tris) > 100
    r = np.linalg.norm(tris.reshape(-1, 3) - np.array([0.1, -0.05, 0.2]), axis=1)
    assert np.abs(r - 0.8).max() < 0.03            # vertices sit on the iso-sphere (linear interp)
    # watertight: every undirected edge is shared by exactly two triangles, with opposite
    # directions (consistent winding) - possible to check exactly because shared vertices are
    # bit-identical
    edges = {}
    for t in tris:
        for a in range(3):
            p, q = tuple(t[a]), tuple(t[(a + 1) % 3])
            edges[(p, q)] = edges.get((p, q), 0) + 1
    assert all(v == 1 for v in edges.values())
    assert all((q, p) in edges for (p, q) in edges)
    # area ~ 4 pi r^2
    assert abs(M.triangle_areas(tris).sum() - 4 * np.pi * 0.64) / (4 * np.pi * 0.64) < 0.03


def test_reference_vertex_scaling_gotcha():
    """vertices are index / S * (max - min) + min with S = G, not G - 1 (utils/eval_3D.py:252-255)."""
    G = 9
    vol, _ = _sphere(G, 0.8, c=(0, 0, 0))
    tris = M.marching_cubes(vol, 0.5, np.float32(3.0 / G), -1.5)
    v = tris.reshape(-1, 3)
    # the sphere is centred in index space at (G-1)/2 -> world centre = (G-1)/2 * 3/G - 1.5 = -1.5/G
    np.testing.assert_allclose(v.mean(0), [-1.5 / G] * 3, atol=0.03)


def test_sampling_is_area_weighted_and_on_surface():
    G = 13
    vol, ax = _sphere(G, 0.9, c=(0, 0, 0))
    step = float(ax[1] - ax[0])
    tris = M.marching_cubes(vol, 0.5, step, float(ax[0]))
    pts, ids = M.sample_surface(tris, 4000, seed=3)
    # on the triangles it claims
    for s in range(0, 4000, 97):
        a, b, c = tris[ids[s]].astype(np.float64)
        n = np.cross(b - a, c - a)
        assert abs(np.dot(pts[s] - a, n)) / (np.linalg.norm(n) + 1e-30) < 1e-5
    r = np.linalg.norm(pts, axis=1)
    assert np.abs(r - 0.9).max() < 0.06
    # octant occupancy ~ uniform
    occ = np.bincount((pts[:, 0] > 0) * 4 + (pts[:, 1] > 0) * 2 + (pts[:, 2] > 0), minlength=8) / 4000.0
    assert np.abs(occ - 0.125).max() < 0.03
    # empty mesh -> zeros
    z, _ = M.sample_surface(np.zeros((0, 3, 3), np.float32), 5, 0)
    assert z.shape == (5, 3) and not z.any()


def test_vectorised_extraction_and_sampling_equal_the_loops_bit_for_bit():
    """oracle/mc_ref.py extracts all cubes (and draws all samples) at once since round 4 so that the pipeline tests run at
    vox 64 / 128; the cube-by-cube and sample-by-sample forms it replaced stay as its check."""
    import numpy as np
    from oracle import mc_ref as M
    rs = np.random.RandomState(0)
    for G in (5, 8, 11):
        ax = np.linspace(-1.5, 1.5, G)
        x, y, z = np.meshgrid(ax, ax, ax, indexing="ij")
        vol = (1.0 / (1.0 + np.exp(20 * (np.sqrt(x * x + 1.3 * y * y + 0.8 * z * z) - 0.9))) + 0.05 * rs.randn(G, G, G)).astype(np.float32)
        a = M.marching_cubes(vol, 0.5, np.float32(3.0 / G), -1.5)
        b = M.marching_cubes_loop(vol, 0.5, np.float32(3.0 / G), -1.5)
        assert a.shape == b.shape and len(a) > 0 and np.array_equal(a.view(np.uint32), b.view(np.uint32))
        p1, i1 = M.sample_surface(a, 300, seed=G)
        p2, i2 = M.sample_surface_loop(b, 300, seed=G)
        assert np.array_equal(i1, i2) and np.array_equal(p1.view(np.uint32), p2.view(np.uint32))
    assert len(M.marching_cubes(np.zeros((4, 4, 4), np.float32), 0.5, np.float32(1.0), 0.0)) == 0


def test_simple_mesh_is_indexed_like_mcubes():
    """utils/eval_3D.py:250-256: mcubes.marching_cubes returns welded (vertices, triangles); SimpleMesh welds the kernel's soup
    lazily (unique rows: shared vertices are bit-identical) and keeps the triangle set."""
    import numpy as np
    from oracle import mc_ref as M
    from zeroshape_amd.utils.eval_3D import SimpleMesh
    G = 20
    vol = _sphere(G, 0.9, c=(0, 0, 0))[0]
    tris = M.marching_cubes(vol, 0.5, np.float32(3.0 / G), -1.5)
    m = SimpleMesh(tris)
    assert len(m.faces) == len(tris) and len(m.vertices) < 3 * len(tris)
    assert len(np.unique(m.vertices.view(np.uint32), axis=0)) == len(m.vertices)
    assert np.array_equal(m.vertices[m.faces].view(np.uint32), (tris + np.float32(0)).view(np.uint32))
    assert len(m.vertices) - len(m.faces) // 2 == 2          # closed genus-0 surface: V - E + F = 2 with E = 3F/2
    e = SimpleMesh(np.zeros((0, 3, 3), np.float32))
    assert e.vertices.shape == (0, 3) and e.faces.shape == (0, 3)
