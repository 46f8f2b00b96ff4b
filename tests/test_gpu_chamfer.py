"""GPU parity: HIP Chamfer kernels (through the C ABI / plugin mirror) vs the oracle.
Bar: squared distances and int32 indices BIT-EXACT (same fmaf chain, same tie rule)."""
import numpy as np
import pytest
import torch

from oracle import chamfer_ref as C
from oracle import geometry_ref as G
from zeroshape_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _run(a, b):
    from zeroshape_amd.external.chamfer3D.dist_chamfer_3D import chamfer_3DDist
    d1, d2, i1, i2 = chamfer_3DDist()(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    return d1.cpu().numpy(), d2.cpu().numpy(), i1.cpu().numpy(), i2.cpu().numpy()


@pytest.mark.parametrize("b,n,m", [(1, 1, 1), (2, 3, 4), (1, 5, 3), (3, 7, 511), (2, 9, 512), (2, 11, 513),
                                   (1, 1023, 1025), (2, 300, 2049), (33, 257, 100), (4, 5000, 3000),
                                   (24, 10000, 10000)])
def test_forward_bit_exact(b, n, m):
    rs = np.random.RandomState(b * 31 + n * 7 + m)
    a = rs.uniform(-0.5, 0.5, (b, n, 3)).astype(np.float32)
    c = rs.uniform(-0.5, 0.5, (b, m, 3)).astype(np.float32)
    got = _run(a, c)
    want = C.chamfer_forward(a, c)
    for g, w, name in zip(got, want, ("dist1", "dist2", "idx1", "idx2")):
        assert g.dtype == w.dtype and g.shape == w.shape, name
        np.testing.assert_array_equal(g, w, err_msg=name)


@pytest.mark.parametrize("lds", [False, True])
def test_brute_force_scans_bit_exact_on_chunk_edges(lds, monkeypatch):
    """zs_chamfer_forward's two scans - nn_both_sgpr_kernel (default since round 6: eight candidates per chunk through the
    scalar cache, two chunks in ping-pong, the last m % 8 one at a time) and the LDS-staged nn_both_kernel (ZS_CHAMFER_LDS=1) -
    against the oracle on every chunk-count parity and tail length, with duplicates that straddle chunk boundaries (the lowest
    index must win) and an infinite point."""
    from zeroshape_amd import chamfer_3D
    if lds:
        monkeypatch.setenv("ZS_CHAMFER_LDS", "1")
    else:
        monkeypatch.delenv("ZS_CHAMFER_LDS", raising=False)
    rs = np.random.RandomState(11)
    for n, m in [(1, 1), (5, 7), (64, 8), (65, 9), (300, 15), (257, 16), (513, 17), (100, 23), (100, 24), (100, 25),
                 (700, 1000), (1025, 1029), (3, 2047)]:
        b = 3
        a = rs.uniform(-0.5, 0.5, (b, n, 3)).astype(np.float32)
        c = rs.uniform(-0.5, 0.5, (b, m, 3)).astype(np.float32)
        if m >= 9:
            c[:, 8] = c[:, 7]                   # equal candidates on both sides of a chunk boundary
            c[1, m - 1] = c[1, 0]               # ... and in the tail / the last chunk
        if m >= 17:
            c[0, 16] = np.inf
        A, Cc = torch.from_numpy(a).cuda(), torch.from_numpy(c).cuda()
        d1, d2 = torch.zeros(b, n, device="cuda"), torch.zeros(b, m, device="cuda")
        i1, i2 = torch.zeros(b, n, dtype=torch.int32, device="cuda"), torch.zeros(b, m, dtype=torch.int32, device="cuda")
        assert chamfer_3D.forward(A, Cc, d1, d2, i1, i2, method="brute") == 1
        want = C.chamfer_forward(a, c)
        for g, w, name in zip((d1, d2, i1, i2), want, ("dist1", "dist2", "idx1", "idx2")):
            np.testing.assert_array_equal(g.cpu().numpy(), w, err_msg="%s n=%d m=%d lds=%s" % (name, n, m, lds))


@pytest.mark.parametrize("kind", ["uniform", "surface", "clusters", "flat", "outliers", "line", "scaled"])
def test_grid_accelerated_path_equals_brute_force(kind, monkeypatch):
    """zs_chamfer_forward_ws prunes candidates with a conservative bound: results must stay
    bit-identical to the brute-force kernel (and the oracle) on adversarial geometry."""
    rs = np.random.RandomState(hash(kind) % 1000)
    n, m, b = 3000, 4100, 3
    if kind == "uniform":
        a, c = rs.uniform(-0.5, 0.5, (b, n, 3)), rs.uniform(-0.5, 0.5, (b, m, 3))
    elif kind == "surface":
        a = rs.randn(b, n, 3); a /= np.linalg.norm(a, axis=-1, keepdims=True)
        c = rs.randn(b, m, 3); c /= np.linalg.norm(c, axis=-1, keepdims=True); c *= 0.98
    elif kind == "clusters":
        a = rs.randn(b, n, 3) * 0.01 + rs.randint(0, 3, (b, n, 1)) * 0.4
        c = rs.randn(b, m, 3) * 0.01 + rs.randint(0, 3, (b, m, 1)) * 0.4
    elif kind == "flat":
        a, c = rs.uniform(-1, 1, (b, n, 3)), rs.uniform(-1, 1, (b, m, 3))
        c[..., 2] = 0.25                      # candidate cloud degenerate along z
        a[0, :, 0] = -0.5                     # and a query cloud degenerate along x
    elif kind == "outliers":
        a, c = rs.uniform(-0.5, 0.5, (b, n, 3)), rs.uniform(-0.5, 0.5, (b, m, 3))
        a[:, :50] += 40.0                     # queries far outside the candidates' bounding box
        c[:, :3] -= 25.0                      # and a few far candidates stretching the grid
    elif kind == "line":
        t = rs.uniform(0, 1, (b, n, 1)); a = np.concatenate([t, 2 * t, -t], -1)
        t = rs.uniform(0, 1, (b, m, 1)); c = np.concatenate([t, 2 * t, -t], -1) + 1e-3
    else:
        a, c = rs.uniform(-500, 500, (b, n, 3)), rs.uniform(-500, 500, (b, m, 3))
    a, c = a.astype(np.float32), c.astype(np.float32)
    c[:, 100] = c[:, 7]                       # exact duplicates: lowest index must win
    a[:, 11] = c[:, 100]
    got = _run(a, c)
    want = C.chamfer_forward(a, c)
    for g, w_, name in zip(got, want, ("dist1", "dist2", "idx1", "idx2")):
        np.testing.assert_array_equal(g, w_, err_msg="%s (%s)" % (name, kind))
    monkeypatch.setenv("ZS_CHAMFER_BRUTE", "1")
    brute = _run(a, c)
    for g, w_ in zip(got, brute):
        np.testing.assert_array_equal(g, w_)


def test_ties_and_duplicates():
    m = 2100   # spans three 1024-candidate LDS tiles
    c = np.tile(np.array([[1, 0, 0]], np.float32), (m, 1))[None].copy()
    a = np.zeros((1, 70, 3), np.float32)
    d1, d2, i1, i2 = _run(a, c)
    assert np.all(i1 == 0) and np.all(d1 == 1.0) and np.all(i2 == 0)
    c[0, 1500] = [0.5, 0, 0]
    c[0, 2050] = [0.5, 0, 0]
    d1, d2, i1, i2 = _run(a, c)
    w = C.chamfer_forward(a, c)
    assert np.all(i1 == 1500)
    np.testing.assert_array_equal(i1, w[2])
    np.testing.assert_array_equal(d2, w[1])
    # clouds made only of duplicates of a few points (collisions everywhere)
    rs = np.random.RandomState(1)
    base = rs.randn(5, 3).astype(np.float32)
    a = base[rs.randint(0, 5, size=(2, 777))]
    c = base[rs.randint(0, 5, size=(2, 1300))]
    got, want = _run(a, c), C.chamfer_forward(a, c)
    for g, w_ in zip(got, want):
        np.testing.assert_array_equal(g, w_)


def test_empty_cloud_leaves_zeros():
    from zeroshape_amd import chamfer_3D
    xyz1 = torch.rand(2, 5, 3).cuda()
    xyz2 = torch.zeros(2, 0, 3).cuda()
    d1 = torch.zeros(2, 5).cuda(); d2 = torch.zeros(2, 0).cuda()
    i1 = torch.zeros(2, 5, dtype=torch.int32).cuda(); i2 = torch.zeros(2, 0, dtype=torch.int32).cuda()
    assert chamfer_3D.forward(xyz1, xyz2, d1, d2, i1, i2) == 1
    assert torch.all(d1 == 0) and torch.all(i1 == 0)


def test_plugin_rejects_bad_tensors():
    from zeroshape_amd import chamfer_3D
    x = torch.rand(1, 4, 3).cuda()
    d = torch.zeros(1, 4).cuda()
    i = torch.zeros(1, 4, dtype=torch.int32).cuda()
    with pytest.raises(TypeError):
        chamfer_3D.forward(x, x, d, d, i.long(), i)
    with pytest.raises(ValueError):
        chamfer_3D.forward(x.cpu(), x, d, d, i, i)
    with pytest.raises(ValueError):
        chamfer_3D.forward(x, x, d[:, :3], d, i, i)


def test_backward_matches_oracle():
    rs = np.random.RandomState(3)
    a = rs.randn(2, 400, 3).astype(np.float32)
    c = rs.randn(2, 300, 3).astype(np.float32)
    from zeroshape_amd.external.chamfer3D.dist_chamfer_3D import chamfer_3DDist
    ta = torch.from_numpy(a).cuda().requires_grad_(True)
    tc = torch.from_numpy(c).cuda().requires_grad_(True)
    d1, d2, i1, i2 = chamfer_3DDist()(ta, tc)
    g1 = torch.from_numpy(rs.randn(2, 400).astype(np.float32)).cuda()
    g2 = torch.from_numpy(rs.randn(2, 300).astype(np.float32)).cuda()
    (d1 * g1).sum().add((d2 * g2).sum()).backward()
    wa, wc = C.chamfer_backward(a, c, g1.cpu().numpy(), g2.cpu().numpy(), i1.cpu().numpy(), i2.cpu().numpy())
    # atomics add in a different order than the sequential oracle: tolerance, not bit-exact
    np.testing.assert_allclose(ta.grad.cpu().numpy(), wa, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(tc.grad.cpu().numpy(), wc, rtol=1e-5, atol=1e-5)


def test_chamfer_distance_and_fscore_helpers():
    from zeroshape_amd.utils import eval_3D as E
    a = torch.from_numpy(syn.seeded_cloud(1, 3, 2000))
    c = torch.from_numpy(syn.seeded_cloud(2, 3, 1500))
    d1, d2, i1, i2 = E.chamfer_distance(None, a.cuda(), c.cuda())
    w1, w2, j1, j2 = G.chamfer_distance(a, c)
    np.testing.assert_array_equal(i1.cpu().numpy(), j1.numpy())
    np.testing.assert_allclose(d1.cpu().numpy(), w1.numpy(), rtol=2e-7, atol=0)   # sqrt rounding only
    f = E.compute_fscore(d1, d2).cpu()
    np.testing.assert_allclose(f.numpy(), G.compute_fscore(w1, w2).numpy(), atol=1e-6)
    n = E.normalize_pc(a.cuda()).cpu()
    np.testing.assert_allclose(n.numpy(), G.normalize_pc(a).numpy(), atol=1e-6)


@pytest.mark.parametrize("shape", ["asym", "ellipsoid", "sphere", "noise"])
def test_pruned_search_equals_exhaustive_search(shape):
    """lower-bound pruning must not change anything: same rotation index, same bits."""
    from zeroshape_amd.utils import eval_3D as E
    from zeroshape_amd.utils.camera import get_rotation_sphere
    R = get_rotation_sphere(24, 24, 12, device="cuda")
    rs = np.random.RandomState(3)
    n = 1500
    if shape == "asym":
        p = rs.randn(n, 3) * np.array([0.5, 0.25, 0.1]) + rs.rand(n, 1) * np.array([0.4, 0.0, 0.2])
    elif shape == "ellipsoid":
        p = syn.ellipsoid_cloud(1, n)
    elif shape == "sphere":
        p = rs.randn(n, 3); p /= np.linalg.norm(p, axis=1, keepdims=True)   # every rotation ties
    else:
        p = rs.uniform(-1, 1, (n, 3))
    pred = torch.from_numpy(p.astype(np.float32))
    gt = (R[2345].cpu() @ pred.T).T.contiguous() + 2e-3 * torch.from_numpy(rs.randn(n, 3).astype(np.float32))
    sl = (0, 6912) if shape != "noise" else (1000, 1700)
    full = E.brute_force_search(pred, gt, device="cuda", rotations=R, rot_slice=sl, return_index=True, prune=False)
    n_full = E.brute_force_search.last_evaluated
    fast = E.brute_force_search(pred, gt, device="cuda", rotations=R, rot_slice=sl, return_index=True, prune=True)
    n_fast = E.brute_force_search.last_evaluated
    assert fast[5] == full[5] and fast[6] == full[6], (fast[5], full[5])
    for a, b in zip(fast[:5], full[:5]):
        assert torch.equal(a, b)
    assert n_fast <= n_full
    if shape in ("asym", "ellipsoid"):
        assert n_fast < n_full // 4, "pruning should remove most rotations (%d of %d evaluated)" % (n_fast, n_full)


def test_brute_force_search_matches_oracle_scan():
    from zeroshape_amd.utils import eval_3D as E
    from zeroshape_amd.utils.camera import get_rotation_sphere
    R = get_rotation_sphere(24, 24, 12, device="cpu")
    rs = np.random.RandomState(0)
    gt = torch.from_numpy((rs.randn(1500, 3) * np.array([0.5, 0.3, 0.2]) + rs.rand(1500, 1) * 0.3).astype(np.float32))
    k = 1234
    pred = (R[k].T @ gt.T).T.contiguous() + 1e-3 * torch.from_numpy(rs.randn(1500, 3).astype(np.float32))
    sub = R[k - 36:k + 36]
    acc, comp, f, best, gt_n, idx, cd = E.brute_force_search(pred, gt, device="cuda", rotations=sub.cuda(),
                                                             return_index=True)
    oacc, ocomp, of, obest, ogt, oidx = G.brute_force_search(pred, gt, rotations=sub)
    assert idx == oidx == 36
    np.testing.assert_allclose(float(acc), float(oacc), rtol=1e-5)
    np.testing.assert_allclose(float(comp), float(ocomp), rtol=1e-5)
    np.testing.assert_allclose(f.cpu().numpy(), of.numpy(), atol=1e-6)


def test_normalize_pc_and_fscore_kernels_vs_reference_goldens(geometry_golden):
    """zs_normalize_pc / zs_fscore against the outputs of the real reference's normalize_pc /
    compute_fscore (tests/golden/make_golden.py): F-score bit for bit (integer counts, same
    division order, 0/0 -> 0), normalize_pc to 1 ulp (the mean is a double sum rounded once)."""
    from zeroshape_amd.utils import eval_3D as E
    pc = torch.from_numpy(syn.seeded_cloud(3, 2, 64)) * torch.tensor([1.0, 2.0, 3.0])
    got = E.normalize_pc(pc.cuda()).cpu().numpy()
    np.testing.assert_allclose(got, geometry_golden["normalize_pc_out"], rtol=0, atol=1.2e-7)
    d1 = torch.from_numpy(np.random.RandomState(5).uniform(0, 0.25, size=(3, 50)).astype(np.float32))
    d2 = torch.from_numpy(np.random.RandomState(6).uniform(0, 0.25, size=(3, 70)).astype(np.float32))
    d1[2] = 1.0
    d2[2] = 1.0
    f = E.compute_fscore(d1.cuda(), d2.cuda()).cpu().numpy()
    np.testing.assert_array_equal(f, geometry_golden["fscore_out"])
    assert np.all(f[2] == 0)
    with pytest.raises(ValueError):
        E.normalize_pc(pc)                      # CPU tensors are rejected, not emulated
    # larger, ragged, batched: against the oracle
    a = torch.from_numpy(syn.seeded_cloud(8, 5, 3001)) * torch.tensor([0.3, 1.7, 5.0]) + 0.4
    np.testing.assert_allclose(E.normalize_pc(a.cuda()).cpu().numpy(), G.normalize_pc(a).numpy(), rtol=4e-7, atol=2e-7)  # the oracle's own fp32 mean is ~1e-7 off


def test_fused_pose_search_equals_the_unfused_kernels():
    """The fused batch kernels (csrc/pose_search.hip) against the same steps as separate launches:
    zs_pose_apply (rotate + normalize_pc) -> chamfer kernel -> sqrt / mean -> zs_fscore, per
    rotation.  Same winner, F-score bit for bit, distances to the last bits of the mean's
    summation order; and the search leaves the running record on the device (no sync needed)."""
    from zeroshape_amd.utils import eval_3D as E
    from zeroshape_amd.utils.camera import get_rotation_sphere
    from zeroshape_amd import _lib
    lib = _lib.load()
    R = get_rotation_sphere(24, 24, 12, device="cuda").float().contiguous()
    rs = np.random.RandomState(12)
    pred = torch.from_numpy((rs.randn(1777, 3) * np.array([0.5, 0.3, 0.2]) + rs.rand(1777, 1) * 0.3).astype(np.float32)).cuda()
    gt = (R[700] @ pred.T).T.contiguous()[:1500] + 3e-3 * torch.randn(1500, 3, device="cuda")
    sl = (660, 760)
    acc, comp, f, best_pred, gt_n, idx, cd = E.brute_force_search(pred, gt, device="cuda", rotations=R, rot_slice=sl,
                                                                  return_index=True, prune=False, batch_size=37)
    assert E.brute_force_search.last_evaluated == 100
    rows = []
    scratch = torch.empty(64, device="cuda")
    for k in range(*sl):
        rot = torch.empty(1777, 3, device="cuda")
        ki = torch.tensor([k], dtype=torch.int32, device="cuda")
        _lib.check(lib.zs_pose_apply(_lib.ptr(pred), 1777, _lib.ptr(R), _lib.ptr(ki), _lib.ptr(rot), _lib.ptr(scratch),
                                     _lib.current_stream_ptr(pred.device)), "zs_pose_apply")
        d1, d2, _, _ = E.chamfer_distance(None, rot[None], gt_n, method="brute")
        rows.append((float((d1.double().mean() + d2.double().mean()) / 2), k, d1, d2, rot))
    want = min(rows, key=lambda r: (r[0], r[1]))
    assert idx == want[1]
    assert abs(cd - want[0]) < 2e-7 * max(1.0, want[0]) and abs(float(acc) - float(want[2].mean())) < 1e-6
    assert torch.equal(best_pred, want[4])
    assert torch.equal(f, E.compute_fscore(want[2], want[3])[0])
    # the un-indexed call returns device tensors only
    out = E.brute_force_search(pred, gt, device="cuda", rotations=R, rot_slice=sl)
    assert len(out) == 5 and all(t.is_cuda for t in out) and torch.equal(out[0], acc) and torch.equal(out[3], best_pred)
    # fewer thresholds than six
    f2 = E.brute_force_search(pred, gt, [0.01, 0.1], device="cuda", rotations=R, rot_slice=sl)[2]
    assert f2.shape == (2,) and torch.equal(f2, f[[1, 4]])


@pytest.mark.parametrize("shape,n,m", [("asym", 1500, 1500), ("ellipsoid", 4000, 3100), ("sphere", 900, 1200),
                                       ("flat", 1000, 1000), ("dup", 700, 650), ("tiny", 5, 3)])
def test_grid_pose_search_equals_the_pairwise_scan(shape, n, m):
    """brute_force_search(nn="grid"): the exact evaluations walk uniform grids (ground truth binned once, the
    rotated + normalised prediction once per rotation) instead of scanning all pairs - same record bit for bit
    (Chamfer-L1, index, accuracy, completeness, six F-scores), exhaustive and pruned, also on degenerate clouds (a
    plane, duplicated points, fewer points than grid cells, every rotation tying on a sphere)."""
    from zeroshape_amd.utils import eval_3D as E
    from zeroshape_amd.utils.camera import get_rotation_sphere
    R = get_rotation_sphere(24, 24, 12, device="cuda")
    rs = np.random.RandomState(n + m)
    if shape == "asym":
        p = rs.randn(n, 3) * np.array([0.5, 0.25, 0.1]) + rs.rand(n, 1) * np.array([0.4, 0.0, 0.2])
    elif shape == "ellipsoid":
        p = syn.ellipsoid_cloud(2, n)
    elif shape == "sphere":
        p = rs.randn(n, 3); p /= np.linalg.norm(p, axis=1, keepdims=True)
    elif shape == "flat":
        p = rs.uniform(-1, 1, (n, 3)) * np.array([1.0, 0.6, 0.0])
    elif shape == "dup":
        p = np.repeat(rs.uniform(-1, 1, (n // 7, 3)), 7, axis=0)[:n]
    else:
        p = rs.uniform(-1, 1, (n, 3))
    pred = torch.from_numpy(p.astype(np.float32))
    gt = ((R[4321].cpu() @ pred.T).T.contiguous() + 2e-3 * torch.from_numpy(rs.randn(n, 3).astype(np.float32)))[
        torch.from_numpy(rs.permutation(n)[:m] if m <= n else rs.randint(0, n, m))]
    sl = (4000, 4700)
    for prune in (False, True):
        for bs in (192, 50):
            b = E.brute_force_search(pred, gt, device="cuda", rotations=R, rot_slice=sl, return_index=True, prune=prune,
                                     batch_size=bs, nn="brute")
            eb = E.brute_force_search.last_evaluated
            for nn in ("grid", "cull", "pairs"):   # the box-culled scan (the default) against the same all-pairs scan
                a = E.brute_force_search(pred, gt, device="cuda", rotations=R, rot_slice=sl, return_index=True, prune=prune,
                                         batch_size=bs, nn=nn)
                assert eb == E.brute_force_search.last_evaluated
                assert a[5] == b[5] and a[6] == b[6], (shape, nn, prune, bs, a[5], b[5], a[6], b[6])
                for x, y in zip(a[:5], b[:5]):
                    assert torch.equal(x, y), (shape, nn, prune, bs)


def test_morton_sort_is_a_stable_permutation_with_tile_boxes():
    """zs_morton_sort: a permutation of the cloud ordered by the 30-bit Z-order key of its bounding box (stable: equal
    keys keep their order, so the result is a pure function of the coordinates), non-finite points last, and the exact
    box of every tile of consecutive sorted points; consecutive points are close (that is what the culled scan needs)."""
    from zeroshape_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(4)
    for n, tile in ((10000, 1024), (1000, 64), (1, 1024), (1500, 1024)):
        p = rs.uniform(-1, 1, (n, 3)).astype(np.float32) * np.array([1.0, 0.5, 0.25], np.float32)
        if n >= 1000:
            p[7] = p[3]                                   # duplicates: equal keys
            p[11, 1] = np.nan
            p[13, 0] = np.inf
        pts = torch.from_numpy(p).cuda()
        out, perm = torch.empty_like(pts), torch.empty(n, dtype=torch.int32, device="cuda")
        nt = (n + tile - 1) // tile
        boxes = torch.empty(nt, 6, device="cuda")
        scratch = torch.empty(lib.zs_morton_scratch_bytes(n) // 4 + 1, device="cuda")
        _lib.check(lib.zs_morton_sort(_lib.ptr(pts), n, _lib.ptr(out), _lib.ptr(perm), _lib.ptr(boxes), tile,
                                      _lib.ptr(scratch), _lib.current_stream_ptr(pts.device)), "zs_morton_sort")
        perm_h, out_h = perm.cpu().numpy(), out.cpu().numpy()
        assert sorted(perm_h.tolist()) == list(range(n))
        assert np.array_equal(out_h, p[perm_h], equal_nan=True)
        fin = np.isfinite(p).all(1)
        lo, hi = p[fin].min(0), p[fin].max(0)
        cell = np.clip(((p - lo) / np.where(hi > lo, hi - lo, 1) * 1024).astype(np.int64), 0, 1023)
        key = np.zeros(n, np.int64)
        for b in range(10):
            for a in range(3):
                key |= ((cell[:, a] >> b) & 1) << (3 * b + a)
        key[~fin] = 0x7fffffff
        want = np.argsort(key, kind="stable")
        # (the float division may land a point that sits exactly on a cell boundary in the neighbouring cell)
        assert (perm_h == want).mean() > 0.99 or n < 4
        assert fin[perm_h][:fin.sum()].all() and not fin[perm_h][fin.sum():].any()
        for t in range(nt):
            blk = out_h[t * tile:(t + 1) * tile]
            blk = np.where(np.isfinite(blk), blk, np.nan)
            if np.isfinite(blk).any():
                np.testing.assert_array_equal(boxes[t, :3].cpu().numpy(), np.nanmin(blk, 0))
                np.testing.assert_array_equal(boxes[t, 3:].cpu().numpy(), np.nanmax(blk, 0))
        if n == 10000:
            step = np.linalg.norm(np.diff(out_h[:fin.sum() - 2], axis=0), axis=1)
            assert np.median(step) < 0.25 * np.median(np.linalg.norm(np.diff(p[fin], axis=0), axis=1))
        # deterministic: a second call gives the same permutation
        perm2 = torch.empty_like(perm)
        _lib.check(lib.zs_morton_sort(_lib.ptr(pts), n, _lib.ptr(out), _lib.ptr(perm2), _lib.ptr(boxes), tile,
                                      _lib.ptr(scratch), _lib.current_stream_ptr(pts.device)), "zs_morton_sort")
        assert torch.equal(perm, perm2)


def test_lower_bound_field_by_sweeps_equals_the_all_cells_form(monkeypatch):
    """csrc/bf_prune.hip: the 32^3 distance field of the pose search's lower bounds by three separable sweeps (round 6)
    against the form that visits every occupied cell (ZS_BF_FIELD_BRUTE=1): the same integer cell distances, so the fields
    and the 6,912 bounds are bit-identical - on a surface cloud, a volume cloud, one point, and a flat cloud - and a bound never
    exceeds the exact Chamfer-L1 of its rotation."""
    from zeroshape_amd.utils import eval_3D as E
    R = E._rotation_sphere(torch.device("cuda"))
    clouds = [(syn.ellipsoid_cloud(0, 3000), syn.ellipsoid_cloud(1, 2500)), (syn.seeded_cloud(4, 1, 2000)[0], syn.seeded_cloud(5, 1, 1500)[0]),
              (np.zeros((1, 3), np.float32) + 0.25, syn.ellipsoid_cloud(2, 300)),
              (syn.seeded_cloud(6, 1, 500)[0] * np.array([1, 1, 0], np.float32), syn.seeded_cloud(7, 1, 400)[0])]
    for a, b in clouds:
        pred = torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
        gt = E.normalize_pc(torch.from_numpy(np.ascontiguousarray(b, np.float32)).cuda()[None])[0]
        monkeypatch.setenv("ZS_BF_FIELD_BRUTE", "1")
        want = E._bf_lower_bounds(pred, gt, R).cpu()
        monkeypatch.setenv("ZS_BF_FIELD_BRUTE", "0")
        got = E._bf_lower_bounds(pred, gt, R).cpu()
        assert torch.equal(got, want)
        assert bool(torch.isfinite(got).all()) and float(got.min()) >= 0.0
    # rigorous: the bound of a rotation <= its exact Chamfer-L1 (checked on a few rotations of the first pair)
    from oracle import geometry_ref as G
    pred_c, gt_c = torch.from_numpy(clouds[0][0]), torch.from_numpy(clouds[0][1])
    lb = E._bf_lower_bounds(pred_c.cuda(), E.normalize_pc(gt_c.cuda()[None])[0], R).cpu()
    for k in (0, 100, 1234, 6911):
        rot = G.normalize_pc((R[k].cpu() @ pred_c.T).T[None])
        d1, d2, _, _ = G.chamfer_distance(rot, G.normalize_pc(gt_c[None]))
        assert float(lb[k]) <= 0.5 * (float(d1.mean()) + float(d2.mean())) * (1 + 1e-4) + 1e-6


def test_str_sort_is_a_permutation_into_compact_leaves():
    """zs_str_sort (sort-tile-recursive: x slabs, y strips, z inside; three stable sorts): a permutation, reproducible,
    non-finite points at the end, and leaves of 64 consecutive points with far smaller boxes than the input order - and
    smaller than the Z-order's, which is why it is the pose search's default."""
    from zeroshape_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(5)

    def run(fn_sort, p):
        pts = torch.from_numpy(p).cuda()
        n = len(p)
        out, perm = torch.empty_like(pts), torch.empty(n, dtype=torch.int32, device="cuda")
        if fn_sort == "str":
            scratch = torch.empty(lib.zs_str_scratch_bytes(n) // 4 + 1, device="cuda")
            rc = lib.zs_str_sort(_lib.ptr(pts), n, _lib.ptr(out), _lib.ptr(perm), _lib.ptr(scratch), _lib.current_stream_ptr(pts.device))
        else:
            scratch = torch.empty(lib.zs_morton_scratch_bytes(n) // 4 + 1, device="cuda")
            rc = lib.zs_morton_sort(_lib.ptr(pts), n, _lib.ptr(out), _lib.ptr(perm), None, 0, _lib.ptr(scratch),
                                    _lib.current_stream_ptr(pts.device))
        _lib.check(rc, "sort")
        return out.cpu().numpy(), perm.cpu().numpy()

    def leaf_diag(q):
        q = q[: len(q) // 64 * 64].reshape(-1, 64, 3)
        return float(np.linalg.norm(q.max(1) - q.min(1), axis=1).mean())

    for n in (10000, 4097, 63, 1):
        p = syn.ellipsoid_cloud(3, n) if n != 4097 else rs.uniform(-1, 1, (n, 3)).astype(np.float32)
        if n >= 4097:
            p[5, 2] = np.nan
            p[9] = p[2]
        out, perm = run("str", p)
        assert sorted(perm.tolist()) == list(range(n)) and np.array_equal(out, p[perm], equal_nan=True)
        out2, perm2 = run("str", p)
        assert np.array_equal(perm, perm2)
        if n >= 4097:
            assert perm[-1] == 5                      # the non-finite point closes the last leaf
            fin = np.delete(p, 5, axis=0)
            mo, _ = run("morton", p)
            assert leaf_diag(out[:-1]) < 0.35 * leaf_diag(fin) and leaf_diag(out[:-1]) < 0.9 * leaf_diag(mo[:-1])


@pytest.mark.parametrize("case", ["far", "aligned", "interior", "ragged"])
def test_culled_scan_equals_all_pairs_on_10k_points(case):
    """The benchmark geometries at full size (10k x 10k, a few dozen rotations): the box-culled scan returns the
    all-pairs record bit for bit where nothing can be pruned ("far": an ellipsoid shell against a uniform cube),
    where almost everything can ("aligned"), for queries deep inside the other cloud's hull ("interior") and for
    ragged sizes that leave partial tiles and sub-tiles."""
    from zeroshape_amd.utils import eval_3D as E
    R = E._rotation_sphere("cuda")
    n, m = (10000, 10000) if case != "ragged" else (4097, 1031)
    pred = torch.from_numpy(syn.ellipsoid_cloud(0, n)).cuda()
    if case == "far":
        gt = torch.from_numpy(syn.seeded_cloud(9, 1, m)[0]).cuda()
    elif case == "interior":
        gt = torch.from_numpy(syn.seeded_cloud(9, 1, m)[0] * 0.2).cuda()
    else:
        g = torch.Generator().manual_seed(0)
        gt = ((R[1234] @ pred.T).T.contiguous().cpu() + 1e-3 * torch.randn(n, 3, generator=g))[:m].cuda()
    for sl in ((1200, 1260), (0, 48)):
        b = E.brute_force_search(pred, gt, device="cuda", rot_slice=sl, return_index=True, prune=False, nn="brute")
        a = E.brute_force_search(pred, gt, device="cuda", rot_slice=sl, return_index=True, prune=False, nn="cull")
        assert a[5] == b[5] and a[6] == b[6], (case, sl, a[5], b[5], a[6], b[6])
        for x, y in zip(a[:5], b[:5]):
            assert torch.equal(x, y), (case, sl)
    # every rotation's own record, not only the winner's
    for k in range(3000, 3012):
        b = E.brute_force_search(pred, gt, device="cuda", rot_slice=(k, k + 1), return_index=True, prune=False, nn="brute")
        a = E.brute_force_search(pred, gt, device="cuda", rot_slice=(k, k + 1), return_index=True, prune=False, nn="cull")
        assert a[6] == b[6] and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), (case, k)


def test_standardize_pc_and_icp_vs_oracle_and_reference_golden():
    """zs_standardize_pc / zs_icp_step (the reference's utils/eval_3D.py:83-91, :271-284 without ATen or LAPACK in the
    loop) against the outputs of the REAL reference (tests/golden/icp_golden.npz) and the oracle: the rotation comes
    from a device-side Jacobi SVD in double, the reference's from torch.svd on an fp32 matrix - 1e-5 after 50 iterations."""
    import os
    from zeroshape_amd.utils import eval_3D as E
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "icp_golden.npz")))
    a, b = syn.icp_clouds()
    A, B = torch.from_numpy(a), torch.from_numpy(b)
    x = A * torch.tensor([2.0, 1.0, 3.0]) + 0.3
    got = E.standardize_pc(x.cuda()).cpu()
    np.testing.assert_allclose(got.numpy(), g["standardize_pc_out"], rtol=0, atol=3e-7)
    np.testing.assert_allclose(got.numpy(), G.standardize_pc(x).numpy(), rtol=0, atol=3e-7)
    for it in (1, 3, 50):
        out = E.ICP(None, A.cuda(), B.cuda(), num_iter=it).cpu().numpy()
        np.testing.assert_allclose(out, g["icp_%d" % it], rtol=0, atol=1e-5)
        np.testing.assert_allclose(out, G.icp(A.clone(), B.clone(), it).numpy(), rtol=0, atol=1e-5)
    # reflections and degenerate clouds: the sign rule is the reference's (row 2 negated), a planar cloud does not blow up
    flat = A.clone()
    flat[..., 2] = 0
    out = E.ICP(None, flat.cuda(), (flat * torch.tensor([1.0, -1.0, 1.0])).cuda(), num_iter=3).cpu()
    assert bool(torch.isfinite(out).all())
    np.testing.assert_allclose(out.numpy(), G.icp(flat.clone(), flat * torch.tensor([1.0, -1.0, 1.0]), 3).numpy(), rtol=0, atol=2e-4)
    with pytest.raises(ValueError):
        E.standardize_pc(x)                      # CPU tensors are rejected, not emulated
