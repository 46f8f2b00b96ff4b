"""CPU tests added in round 6: the product's host functions against the reference's goldens, the oracle against the
full-grid goldens of the reference, GradReducer.reduce_in_place over consecutive steps, the distributed prepare()."""
import os

import numpy as np
import pytest
import torch


def test_product_pos_embed_and_rotation_sphere_vs_the_reference(decoder_golden, geometry_golden):
    """The PRODUCT's host functions (not the oracle's copies) against the arrays the reference itself produced
    (utils/pos_embed.py:20-38, utils/camera.py:208-230)."""
    from zeroshape_amd.utils.camera import get_rotation_sphere
    from zeroshape_amd.utils.pos_embed import get_2d_sincos_pos_embed
    pe = get_2d_sincos_pos_embed(256, 14, cls_token=True)
    assert pe.shape == (197, 256)
    np.testing.assert_array_equal(pe.astype(np.float32), decoder_golden["pos_embed_f32"])
    np.testing.assert_array_equal(pe[[0, 1, 2, 14, 15, 100, 195, 196]], decoder_golden["pos_embed_f64_rows"])
    np.testing.assert_array_equal(np.array([pe.sum(), np.abs(pe).sum()]), decoder_golden["pos_embed_f64_sum"])
    R = get_rotation_sphere(24, 24, 12, device="cpu")
    assert tuple(R.shape) == (6912, 3, 3) and R.dtype == torch.float32
    np.testing.assert_array_equal(R.numpy(), geometry_golden["rot_all_f32"])
    np.testing.assert_array_equal(R[[0, 1, 12, 287, 288, 1234, 6911]].numpy(), geometry_golden["rot_rows"])


@pytest.mark.parametrize("N", [64, 128, 256])
def test_oracle_vs_full_grid_golden_of_the_reference(seeded_sd, grid_golden, grid256_golden, N):
    """oracle/decoder_ref.py against the reference's own full-grid run at the BASELINE sizes: every near-surface point
    (|logit| < 1e-3: the candidates for an index flip) and the strided sample, logits within 2e-5, no sign disagreement
    outside |logit| < 1e-5."""
    from oracle import decoder_ref as R
    from zeroshape_amd import synthetic as syn
    G = N + 1
    if N == 256:
        grid_golden = grid256_golden
    latent = torch.from_numpy(syn.seeded_latent(seed=0, batch=2))[:1]
    axis = torch.linspace(-1.5, 1.5, G)
    idx, want = grid_golden["near%d_idx" % N].astype(np.int64), grid_golden["near%d_logit" % N]
    ix, iy, iz = idx // (G * G), idx // G % G, idx % G
    pts = torch.stack([axis[ix], axis[iy], axis[iz]], -1)[None]
    got = R.implicit_forward(seeded_sd, latent, pts)[0][0].numpy()
    np.testing.assert_allclose(got, want, atol=2e-5, rtol=0)
    flips = (got > 0) != (want > 0)
    assert np.all(np.abs(want[flips]) < 1e-5)
    s = {64: 4, 128: 8, 256: 16}[N]
    a = axis[::s]
    gx, gy, gz = torch.meshgrid(a, a, a, indexing="ij")
    pts = torch.stack([gx, gy, gz], -1).view(1, -1, 3)
    got = R.implicit_forward(seeded_sd, latent, pts)[0][0].numpy()
    np.testing.assert_allclose(got, grid_golden["logit%d_s%d" % (N, s)].reshape(-1), atol=2e-5, rtol=0)


def _posenc_sd(posenc_golden, decoder_golden):
    from zeroshape_amd import synthetic as syn
    L = int(posenc_golden["posenc_3D"][0])
    sd = syn.seeded_state_dict(seed=0, pos_embed=decoder_golden["pos_embed_f32"], posenc_3D=L)
    return L, {k: torch.from_numpy(v) for k, v in sd.items()}


def test_oracle_posenc_3d_vs_reference_golden(posenc_golden, decoder_golden):
    """oracle/decoder_ref.py with posenc_3D = 4 (implicit.py:139-166) against the reference's own outputs: the embedding,
    the logits and attention rows of a training-shape call, the 9^3 level grid, and gradients through the differentiable form."""
    from oracle import decoder_ref as R
    from zeroshape_amd import synthetic as syn
    L, sd = _posenc_sd(posenc_golden, decoder_golden)
    assert sd["impl_mlp.layers.0.weight"].shape == (256, 3 + 6 * L + 256) and sd["impl_mlp.layers.2.weight"].shape == (256, 512 + 3 + 6 * L)
    rs = np.random.RandomState(321)
    pts = torch.from_numpy(rs.uniform(-1.5, 1.5, size=(2, 1024, 3)).astype(np.float32))
    np.testing.assert_allclose(R.posenc_3d(pts[0, :64], L).numpy(), posenc_golden["embed_rows"], atol=1e-6, rtol=0)
    latent = torch.from_numpy(syn.seeded_latent(seed=0, batch=2))
    lg, at = R.implicit_forward(sd, latent, pts)
    np.testing.assert_allclose(lg.numpy(), posenc_golden["logit"], atol=5e-6, rtol=0)
    np.testing.assert_allclose(at[:, ::128].numpy(), posenc_golden["attn_rows"], atol=2e-7, rtol=0)
    occ = R.level_grid(sd, latent[:1], R.dense_grid(-1.5, 1.5, 8))
    np.testing.assert_allclose(occ[0].numpy(), posenc_golden["occ8"], atol=2e-6, rtol=0)
    # gradients: torch autograd through the oracle's differentiable restatement
    leaf = {k: v.clone().requires_grad_(k != "pos_embed") for k, v in sd.items()}
    lat = latent.clone().requires_grad_(True)
    out = R.implicit_forward_train(leaf, lat, pts)
    (out * torch.from_numpy(posenc_golden["loss_weights"])).sum().backward()
    for k in [k[5:] for k in posenc_golden if k.startswith("grad.") and k != "grad.latent"]:
        w = torch.from_numpy(posenc_golden["grad." + k]).double()
        stride = 16 if "qkv" in k else (4 if k.startswith("impl_mlp") and leaf[k].shape[0] > 1 else 1)
        assert float((leaf[k].grad[::stride].double() - w).norm()) <= 2e-5 * float(w.norm()), k
    w = torch.from_numpy(posenc_golden["grad.latent"]).double()
    assert float((lat.grad[:, ::8].double() - w).norm()) <= 2e-5 * float(w.norm())


def _variant(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_variants_golden", os.path.join(os.path.dirname(__file__), "golden",
                                                                                       "make_variants_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                   # (only its tables and the seeded inputs: nothing of the reference is imported)
    return mod.VARIANTS[name], mod.inputs(name)


@pytest.mark.parametrize("name", ["defaults", "head", "skips"])
def test_oracle_other_constructor_configurations_vs_reference_golden(variants_golden, name):
    """oracle/decoder_ref.py on the reference class's own defaults, on a prediction head + semantic codes + three blocks, and on
    other skips + posenc_3D 2 - logits, attention rows and row sums, gradients - against the reference's outputs."""
    from oracle import decoder_ref as R
    from zeroshape_amd import synthetic as syn
    from zeroshape_amd.utils.pos_embed import get_2d_sincos_pos_embed
    v, (lat, sem, pts, w) = _variant(name)
    g = int(round(v["syn"]["num_patches"] ** 0.5))
    pe = get_2d_sincos_pos_embed(v["syn"]["n_channels"], g, cls_token=True).astype(np.float32)
    np.testing.assert_allclose([pe.astype(np.float64).sum(), np.abs(pe.astype(np.float64)).sum()], variants_golden[name + ".pos_embed_sum"],
                               rtol=1e-12)
    sd = {k: torch.from_numpy(a) for k, a in syn.seeded_state_dict(seed=3, pos_embed=pe, **v["syn"]).items()}
    per_layer = v["ctor"].get("pos_perlayer", True)
    sem_t = torch.from_numpy(sem) if sem is not None else None
    lg, at = R.implicit_forward(sd, torch.from_numpy(lat), torch.from_numpy(pts), num_heads=v["heads"], pos_perlayer=per_layer,
                                latent_semantic=sem_t)
    np.testing.assert_allclose(lg.numpy(), variants_golden[name + ".logit"], atol=5e-6, rtol=0)
    np.testing.assert_allclose(at[:, ::64].numpy(), variants_golden[name + ".attn_rows"], atol=2e-7, rtol=0)
    np.testing.assert_allclose(at.sum(-1).numpy(), variants_golden[name + ".attn_rowsum"], atol=2e-6, rtol=0)
    leaf = {k: t.clone().requires_grad_(k != "pos_embed") for k, t in sd.items()}
    lat_t = torch.from_numpy(lat).requires_grad_(True)
    full = torch.cat([lat_t, sem_t], -1) if sem_t is not None else lat_t
    out = R.implicit_forward_train(leaf, full, torch.from_numpy(pts), num_heads=v["heads"], pos_perlayer=per_layer)
    (out * torch.from_numpy(w)).sum().backward()
    for k in [k for k in variants_golden if k.startswith(name + ".grad.") and not k.endswith(".latent")]:
        pk = k[len(name) + 6:]
        stride = 16 if "qkv" in pk else (4 if pk.startswith("impl_mlp") else 1)
        want = torch.from_numpy(variants_golden[k]).double()
        got = leaf[pk].grad[::stride].double() if stride > 1 else leaf[pk].grad.double()
        assert float((got - want).norm()) <= 2e-5 * float(want.norm()), k
    want = torch.from_numpy(variants_golden[name + ".grad.latent"]).double()
    assert float((lat_t.grad[:, ::8].double() - want).norm()) <= 2e-5 * float(want.norm())


def test_oracle_evaluation_glue_vs_reference_golden(eval_golden):
    """oracle/geometry_ref.py (normalize_pc, chamfer_distance on the C kernel restatement, icp, compute_fscore,
    brute_force_search) composed the way utils/eval_3D.py:104-213 composes them, against what the reference's own
    eval_metrics_default / eval_metrics_BF / brute_force_search returned for the same clouds."""
    from oracle import geometry_ref as G
    g = eval_golden
    thresholds = (0.005, 0.01, 0.02, 0.05, 0.1, 0.2)
    pred, gt, pose = torch.from_numpy(g["pred"]), torch.from_numpy(g["gt"]), torch.from_numpy(g["pose"])

    def view_frame(flip):
        p = (pose[..., :3] @ gt.permute(0, 2, 1)).permute(0, 2, 1).contiguous()          # :120-121
        if flip:
            p[:, :, :2] *= -1                                                            # :122-123 (pix3d)
        return p
    for tag, flip, icp in (("default_synthetic", False, False), ("default_pix3d_icp", True, True)):
        a, b = G.normalize_pc(pred), G.normalize_pc(view_frame(flip))
        if icp:
            a = G.icp(a, b)
        d1, d2, _, _ = G.chamfer_distance(a, b)
        np.testing.assert_allclose(d1.mean(1).numpy(), g[tag + ".cd_acc"], atol=1e-6, rtol=0)
        np.testing.assert_allclose(d2.mean(1).numpy(), g[tag + ".cd_comp"], atol=1e-6, rtol=0)
        np.testing.assert_allclose(G.compute_fscore(d1, d2, thresholds).numpy(), g[tag + ".f_score"], atol=1e-6, rtol=0)
        np.testing.assert_allclose(a.numpy(), g[tag + ".dpc_pred"], atol=2e-6, rtol=0)
        np.testing.assert_allclose(b.numpy(), g[tag + ".dpc_gt"], atol=1e-6, rtol=0)
    out = G.brute_force_search(pred[0], gt[0], thresholds)
    acc, comp, fs, best_pred = out[0], out[1], out[2], out[3]
    assert abs(float(acc) - float(g["search.acc"])) < 1e-6 and abs(float(comp) - float(g["search.comp"])) < 1e-6
    np.testing.assert_allclose(np.asarray(fs, np.float32).reshape(-1), g["search.f_score"].reshape(-1), atol=1e-6, rtol=0)
    np.testing.assert_allclose(np.asarray(best_pred, np.float32).reshape(-1, 3), g["search.best_pred"], atol=2e-6, rtol=0)
