"""Known-answer + fp64 cross-checks pinning oracle/chamfer_ref.c (the reference's CUDA
kernel cannot run here and upstream has no vectors for it: 'parity unpinned upstream')."""
import numpy as np
import pytest

from oracle import chamfer_ref as C


def _brute_f64(a, b):
    d = ((a[:, :, None, :].astype(np.float64) - b[:, None, :, :].astype(np.float64)) ** 2).sum(-1)
    return d.min(2), d.argmin(2), d.min(1), d.argmin(1)


def test_hand_computed():
    a = np.array([[[0, 0, 0], [1, 0, 0], [0, 2, 0]]], np.float32)
    b = np.array([[[0, 0, 1], [1, 1, 0], [5, 5, 5], [0, 2, 0.5]]], np.float32)
    d1, d2, i1, i2 = C.chamfer_forward(a, b)
    np.testing.assert_array_equal(d1, [[1.0, 1.0, 0.25]])
    np.testing.assert_array_equal(i1, [[0, 1, 3]])
    np.testing.assert_array_equal(d2, [[1.0, 1.0, 59.0, 0.25]])
    np.testing.assert_array_equal(i2, [[0, 1, 2, 2]])
    assert d1.dtype == np.float32 and i1.dtype == np.int32


def test_tie_rule_lowest_index_wins():
    # every reference point is the same distance from the query; duplicates across the
    # 512-tile boundary of the CUDA kernel (chamfer3D.cu:13,126) must still give index 0
    m = 1100
    b = np.tile(np.array([[1, 0, 0]], np.float32), (m, 1))[None]
    a = np.zeros((1, 3, 3), np.float32)
    d1, d2, i1, i2 = C.chamfer_forward(a, b)
    assert np.all(i1 == 0) and np.all(d1 == 1.0)
    assert np.all(i2 == 0)
    b[0, 700] = [0.5, 0, 0]     # unique better point in the second tile
    b[0, 900] = [0.5, 0, 0]     # equal duplicate later: must not win
    assert np.all(C.chamfer_forward(a, b)[2] == 700)


@pytest.mark.parametrize("n,m", [(1, 1), (3, 4), (5, 3), (7, 511), (9, 512), (11, 513), (300, 1025), (2000, 1500)])
def test_ragged_sizes_vs_fp64(n, m):
    rs = np.random.RandomState(n * 7919 + m)
    a = rs.uniform(-0.5, 0.5, (2, n, 3)).astype(np.float32)
    b = rs.uniform(-0.5, 0.5, (2, m, 3)).astype(np.float32)
    d1, d2, i1, i2 = C.chamfer_forward(a, b)
    r1, j1, r2, j2 = _brute_f64(a, b)
    np.testing.assert_allclose(d1, r1, rtol=3e-6, atol=1e-9)
    np.testing.assert_allclose(d2, r2, rtol=3e-6, atol=1e-9)
    # indices agree wherever fp64 has a clear winner
    for d, ii, jj, x, y in ((d1, i1, j1, a, b), (d2, i2, j2, b, a)):
        bad = ii != jj
        if bad.any():
            bi, pi = np.nonzero(bad)
            da = ((x[bi, pi] - y[bi, ii[bi, pi]]).astype(np.float64) ** 2).sum(-1)
            db = ((x[bi, pi] - y[bi, jj[bi, pi]]).astype(np.float64) ** 2).sum(-1)
            assert np.all(np.abs(da - db) <= 1e-6 * np.maximum(da, 1e-12))


def test_fma_contraction_is_spelled_out():
    # d = fma(z,z, fma(y,y, x*x)): pick values where plain (x*x + y*y) + z*z rounds differently
    x, y, z = np.float32(0.1), np.float32(0.3), np.float32(0.7)
    a = np.zeros((1, 1, 3), np.float32)
    b = np.array([[[x, y, z]]], np.float32)
    d1 = C.chamfer_forward(a, b)[0][0, 0]
    xx = np.float32(x * x)
    t = np.float32(np.float64(y) * np.float64(y) + np.float64(xx))          # one rounding
    want = np.float32(np.float64(z) * np.float64(z) + np.float64(t))        # one rounding
    assert d1 == want


def test_empty_reference_cloud_leaves_zeros():
    a = np.ones((1, 4, 3), np.float32)
    b = np.zeros((1, 0, 3), np.float32)
    d1, d2, i1, i2 = C.chamfer_forward(a, b)
    assert d1.shape == (1, 4) and np.all(d1 == 0) and np.all(i1 == 0) and d2.shape == (1, 0)


def test_backward_matches_analytic():
    rs = np.random.RandomState(1)
    a = rs.randn(2, 50, 3).astype(np.float32)
    b = rs.randn(2, 60, 3).astype(np.float32)
    d1, d2, i1, i2 = C.chamfer_forward(a, b)
    g1 = rs.randn(2, 50).astype(np.float32)
    g2 = rs.randn(2, 60).astype(np.float32)
    ga, gb = C.chamfer_backward(a, b, g1, g2, i1, i2)
    ea = np.zeros_like(a, dtype=np.float64)
    eb = np.zeros_like(b, dtype=np.float64)
    for bi in range(2):
        for j in range(50):
            v = 2 * g1[bi, j] * (a[bi, j].astype(np.float64) - b[bi, i1[bi, j]])
            ea[bi, j] += v
            eb[bi, i1[bi, j]] -= v
        for j in range(60):
            v = 2 * g2[bi, j] * (b[bi, j].astype(np.float64) - a[bi, i2[bi, j]])
            eb[bi, j] += v
            ea[bi, i2[bi, j]] -= v
    np.testing.assert_allclose(ga, ea, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(gb, eb, rtol=1e-5, atol=1e-5)


def test_evaluation_size_clouds_vs_scipy_kdtree():
    """An independent implementation at the evaluation's size ([.,10000] x [.,10000] points, utils/eval_3D.py:136):
    scipy's cKDTree (exact nearest neighbour, float64).  Squared distances agree to fp32 rounding; indices agree
    wherever the runner-up is not within rounding of the winner."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(11)
    a = rng.uniform(-0.5, 0.5, (2, 10000, 3)).astype(np.float32)
    b = (a[:, rng.permutation(10000)] + rng.normal(0, 0.01, (2, 10000, 3))).astype(np.float32)
    d1, d2, i1, i2 = C.chamfer_forward(a, b)
    for k in range(2):
        for x, y, d, i in ((a[k], b[k], d1[k], i1[k]), (b[k], a[k], d2[k], i2[k])):
            dist, idx = cKDTree(y.astype(np.float64)).query(x.astype(np.float64), k=2)
            np.testing.assert_allclose(d, dist[:, 0] ** 2, rtol=2e-5, atol=1e-12)
            clear = dist[:, 1] ** 2 - dist[:, 0] ** 2 > 1e-6 * (1.0 + dist[:, 1] ** 2)
            assert clear.mean() > 0.99 and np.array_equal(i[clear], idx[clear, 0])
