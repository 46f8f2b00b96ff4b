"""Marching-cubes case tables, GENERATED (not transcribed): for each of the 256 corner
sign patterns the iso-contour is traced face by face and closed into oriented loops, which
are fan-triangulated.

Conventions (those of the classic algorithm PyMCubes implements, utils/eval_3D.py:250):
  corner i of a cube at (x,y,z): v0 (0,0,0) v1 (1,0,0) v2 (1,1,0) v3 (0,1,0)
                                  v4 (0,0,1) v5 (1,0,1) v6 (1,1,1) v7 (0,1,1)
  edges: e0 v0v1, e1 v1v2, e2 v2v3, e3 v3v0, e4 v4v5, e5 v5v6, e6 v6v7, e7 v7v4,
         e8 v0v4, e9 v1v5, e10 v2v6, e11 v3v7
  case index bit i is set when value(corner i) < isovalue ("inside").

Ambiguous faces (two diagonal inside corners) are resolved by ONE rule that depends only on
the four corner states of the face - every maximal run of inside corners along the face
boundary is cut off by its own segment - so the two cubes sharing a face always agree and
the surface is crack-free by construction.  Triangle winding is consistent over the whole
mesh (inside on the left of every contour segment seen from outside the cube).

PyMCubes itself is not installable here ("parity unpinned"): geometry can
differ from its table only inside ambiguous cubes.
"""
import numpy as np

CORNERS = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0],
                    [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], np.int32)
EDGES = np.array([[0, 1], [1, 2], [2, 3], [3, 0], [4, 5], [5, 6], [6, 7], [7, 4],
                  [0, 4], [1, 5], [2, 6], [3, 7]], np.int32)
# faces as corner cycles, counter-clockwise when seen from outside the cube
FACES = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (3, 7, 6, 2), (0, 4, 7, 3), (1, 2, 6, 5)]
_EDGE_ID = {}
for _e, (_a, _b) in enumerate(EDGES):
    _EDGE_ID[(_a, _b)] = _e
    _EDGE_ID[(_b, _a)] = _e


def _case_triangles(case):
    inside = [(case >> i) & 1 for i in range(8)]
    nxt = {}                                   # directed contour: crossing edge -> crossing edge
    for face in FACES:
        s = [inside[c] for c in face]
        if sum(s) in (0, 4):
            continue
        for i in range(4):
            # in -> out crossing on boundary edge (face[i], face[i+1]): end of an inside run
            if s[i] == 1 and s[(i + 1) % 4] == 0:
                a = _EDGE_ID[(face[i], face[(i + 1) % 4])]
                j = i                              # walk back to the start of the run
                while s[(j - 1) % 4] == 1:
                    j = (j - 1) % 4
                b = _EDGE_ID[(face[(j - 1) % 4], face[j])]
                assert a not in nxt
                nxt[a] = b
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
        for k in range(1, len(loop) - 1):
            tris.append((loop[0], loop[k], loop[k + 1]))
    return tris


def build_tables():
    """-> (tri_table int8 [256, 3*MAX_TRIS], tri_count int32 [256], MAX_TRIS)."""
    all_tris = [_case_triangles(c) for c in range(256)]
    max_tris = max(len(t) for t in all_tris)
    table = -np.ones((256, 3 * max_tris), np.int8)
    count = np.zeros(256, np.int32)
    for c, tris in enumerate(all_tris):
        count[c] = len(tris)
        for t, tri in enumerate(tris):
            table[c, 3 * t:3 * t + 3] = tri
    return table, count, max_tris


TRI_TABLE, TRI_COUNT, MAX_TRIS = build_tables()
