"""Oracle geometry helpers vs golden outputs of the real reference."""
import numpy as np
import torch

from oracle import geometry_ref as G
from zeroshape_amd import synthetic as syn


def test_rotation_sphere(geometry_golden):
    R = G.rotation_sphere(24, 24, 12)
    assert R.shape == (6912, 3, 3) and R.dtype == torch.float32
    np.testing.assert_array_equal(R.numpy(), geometry_golden["rot_all_f32"])
    np.testing.assert_array_equal(R[[0, 1, 12, 287, 288, 1234, 6911]].numpy(), geometry_golden["rot_rows"])


def test_normalize_pc(geometry_golden):
    pc = torch.from_numpy(syn.seeded_cloud(3, 2, 64)) * torch.tensor([1.0, 2.0, 3.0])
    np.testing.assert_array_equal(G.normalize_pc(pc).numpy(), geometry_golden["normalize_pc_out"])


def test_fscore_including_nan_branch(geometry_golden):
    d1 = torch.from_numpy(np.random.RandomState(5).uniform(0, 0.25, size=(3, 50)).astype(np.float32))
    d2 = torch.from_numpy(np.random.RandomState(6).uniform(0, 0.25, size=(3, 70)).astype(np.float32))
    d1[2] = 1.0
    d2[2] = 1.0
    f = G.compute_fscore(d1, d2)
    np.testing.assert_array_equal(f.numpy(), geometry_golden["fscore_out"])
    assert np.all(f[2].numpy() == 0)  # 0/0 -> NaN -> 0 (utils/eval_3D.py:228)


def test_brute_force_search_recovers_known_rotation():
    """pred = gt rotated by R[k]^-1 => best index must be k for a cloud with no symmetry,
    and the scan keeps the FIRST strict minimum (utils/eval_3D.py:161-168)."""
    R = G.rotation_sphere(24, 24, 12)
    rs = np.random.RandomState(0)
    gt = torch.from_numpy((rs.randn(400, 3) * np.array([0.5, 0.3, 0.2]) + rs.rand(400, 1) * 0.3).astype(np.float32))
    k = 1234
    pred = (R[k].T @ gt.T).T.contiguous()  # R[k] @ pred == gt
    sub = R[k - 30:k + 30]
    acc, comp, f, best_pc, gt_n, idx = G.brute_force_search(pred, gt, rotations=sub)
    assert idx == 30
    assert float(acc) < 1e-5 and float(comp) < 1e-5
    assert f.shape == (6,) and torch.all(f == 1)
    # duplicated rotation later in the list must NOT replace the first minimum
    sub2 = torch.cat([sub, sub[30:31]], 0)
    assert G.brute_force_search(pred, gt, rotations=sub2)[5] == 30


def test_standardize_pc_and_icp_vs_reference_golden():
    """oracle/geometry_ref.standardize_pc / icp against the outputs of the real reference's utils/eval_3D.py:83-91,
    :271-284 (tests/golden/make_icp_golden.py; the reference's Chamfer plugin replaced there by a float64 cdist)."""
    import os
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "icp_golden.npz")))
    a, b = syn.icp_clouds()
    A, B = torch.from_numpy(a), torch.from_numpy(b)
    got = G.standardize_pc(A * torch.tensor([2.0, 1.0, 3.0]) + 0.3)
    np.testing.assert_allclose(got.numpy(), g["standardize_pc_out"], rtol=0, atol=2e-7)
    for it in (1, 3, 50):
        np.testing.assert_allclose(G.icp(A.clone(), B.clone(), it).numpy(), g["icp_%d" % it], rtol=0, atol=2e-6)
    # it converges onto the rotated copy: residual at the noise level
    d1, _, _, _ = G.chamfer_distance(torch.from_numpy(g["icp_50"]), B)
    assert float(d1.mean()) < 0.02
