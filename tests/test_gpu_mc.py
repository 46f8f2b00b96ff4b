"""GPU parity: marching cubes + surface sampling kernels vs oracle/mc_ref.py (same tables,
same operation order): triangles and sampled points bit-exact."""
import numpy as np
import pytest
import torch

from oracle import mc_ref as M
from tests.test_oracle_mc import _sphere

pytestmark = pytest.mark.gpu


def _noise_vol(G, seed):
    rs = np.random.RandomState(seed)
    return rs.uniform(0, 1, (G, G, G)).astype(np.float32)       # every case, incl. ambiguous ones


@pytest.mark.parametrize("G,kind", [(2, "noise"), (3, "noise"), (9, "noise"), (17, "sphere"), (12, "noise"), (23, "noise"), (33, "sphere"),
                                    (41, "noise")])
def test_triangles_bit_exact(G, kind):
    from zeroshape_amd.utils import eval_3D as E
    vol = _noise_vol(G, G) if kind == "noise" else _sphere(G, 0.8)[0]
    scale = np.float32(3.0 / G)
    want = M.marching_cubes(vol, 0.5, scale, -1.5)
    tris, _ = E.extract_surface(torch.from_numpy(vol).cuda(), 0.5, -1.5, 1.5)
    got = tris.cpu().numpy()
    assert got.shape == want.shape
    np.testing.assert_array_equal(got, want)


def test_empty_and_full_volumes():
    from zeroshape_amd.utils import eval_3D as E
    for val in (0.0, 1.0):
        tris, pts = E.extract_surface(torch.full((8, 8, 8), val).cuda(), 0.5, -1.5, 1.5, num_points=16)
        assert tris.shape == (0, 3, 3)
        assert pts.shape == (16, 3) and not bool(pts.any())       # utils/eval_3D.py:262


def test_sampling_matches_oracle():
    from zeroshape_amd.utils import eval_3D as E
    vol = _sphere(21, 0.9)[0]
    tris, pts = E.extract_surface(torch.from_numpy(vol).cuda(), 0.5, -1.5, 1.5, num_points=3000, seed=11)
    want, _ = M.sample_surface(tris.cpu().numpy(), 3000, seed=11)
    got = pts.cpu().numpy()
    same = np.all(got == want, axis=1)
    assert same.mean() > 0.999          # a target on a cumulative-area boundary may pick the neighbour
    np.testing.assert_allclose(got[same], want[same], atol=0, rtol=0)


def test_convert_to_explicit_contract_and_full_size():
    """reference signature: (meshes, pointclouds [B,10000,3] float64); 129^3 grid from the decoder
    path: cloud lies on the level set of the trilinear field within a voxel."""
    from zeroshape_amd.utils import eval_3D as E
    from zeroshape_amd.utils.options import EasyDict as edict
    opt = edict(dict(device="cuda", eval=dict(range=[-1.5, 1.5], num_points=10000)))
    G = 129
    vol = torch.from_numpy(_sphere(G, 1.0, c=(0, 0, 0))[0]).cuda()
    meshes, clouds = E.convert_to_explicit(opt, [vol, vol.cpu().numpy()], isoval=0.5, to_pointcloud=True)
    assert len(meshes) == 2 and clouds.shape == (2, 10000, 3) and clouds.dtype == np.float64
    # the indexed form of mcubes.marching_cubes (utils/eval_3D.py:250-256): welded vertices, the triangle set unchanged
    m = meshes[0]
    assert m.faces.shape[1] == 3 and len(m.faces) == len(m.triangles)
    assert len(np.unique(m.vertices.view(np.uint32), axis=0)) == len(m.vertices) < 3 * len(m.faces)
    assert np.array_equal(m.vertices[m.faces].view(np.uint32), (m.triangles + np.float32(0)).view(np.uint32))
    assert len(m.vertices) - 3 * len(m.faces) // 2 + len(m.faces) in (2, 0, 4)      # Euler characteristic of a closed surface (V - E + F, E = 3F/2)
    # world coordinates use the reference's S = G scaling: radius 1.0 * (G-1)/G, centre -1.5/G
    r = np.linalg.norm(clouds[0] - (-1.5 / G), axis=1)
    assert abs(r.mean() - (G - 1) / G) < 2e-3 and r.std() < 2e-3
    assert not np.array_equal(clouds[0], clouds[1])      # different seeds per grid index
