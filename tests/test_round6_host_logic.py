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


@pytest.mark.parametrize("N", [64, 128])
def test_oracle_vs_full_grid_golden_of_the_reference(seeded_sd, grid_golden, N):
    """oracle/decoder_ref.py against the reference's own full-grid run at the BASELINE sizes: every near-surface point
    (|logit| < 1e-3: the candidates for an index flip) and the strided sample, logits within 2e-5, no sign disagreement
    outside |logit| < 1e-5."""
    from oracle import decoder_ref as R
    from zeroshape_amd import synthetic as syn
    G = N + 1
    latent = torch.from_numpy(syn.seeded_latent(seed=0, batch=2))[:1]
    axis = torch.linspace(-1.5, 1.5, G)
    idx, want = grid_golden["near%d_idx" % N].astype(np.int64), grid_golden["near%d_logit" % N]
    ix, iy, iz = idx // (G * G), idx // G % G, idx % G
    pts = torch.stack([axis[ix], axis[iy], axis[iz]], -1)[None]
    got = R.implicit_forward(seeded_sd, latent, pts)[0][0].numpy()
    np.testing.assert_allclose(got, want, atol=2e-5, rtol=0)
    flips = (got > 0) != (want > 0)
    assert np.all(np.abs(want[flips]) < 1e-5)
    s = {64: 4, 128: 8}[N]
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
        assert float((leaf[k].grad.double() - w).norm()) <= 2e-5 * float(w.norm()), k
    w = torch.from_numpy(posenc_golden["grad.latent"]).double()
    assert float((lat.grad.double() - w).norm()) <= 2e-5 * float(w.norm())
