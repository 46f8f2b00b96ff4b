"""Oracle encoders (oracle/encoder_ref.py) vs golden outputs of the reference's own modules run
over the stand-in backbones (tests/golden/make_encoder_golden.py)."""
import numpy as np
import torch

from oracle import encoder_ref as E
from zeroshape_amd import synthetic as syn

TOL = dict(rtol=2e-5, atol=2e-5)


def close(got, want, tol=1e-5, msg=""):
    """max |got - want| <= tol * max|want| (torch picks conv algorithms by thread count and
    shape, so the last bits of a deep stack are not reproducible even on the CPU)."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, msg
    assert np.abs(got - want).max() <= tol * max(np.abs(want).max(), 1e-6), \
        "%s: err %.3g scale %.3g" % (msg, np.abs(got - want).max(), np.abs(want).max())


def sample(x, step):
    return x.detach().numpy().reshape(-1)[::step]


def test_dpt_depth_with_intermediates(encoder_golden, encoder_sd):
    rgb, _ = [torch.from_numpy(a) for a in syn.seeded_rgb_scene(seed=0, batch=2)]
    taps = {}
    depth, feat = E.dpt_depth(E._sub(encoder_sd, "dpt_depth."), rgb, taps)
    for name, t in taps.items():
        assert list(t.shape) == list(encoder_golden["dpt_%s_shape" % name]), name
        close(sample(t, 997), encoder_golden["dpt_%s_s997" % name], msg=name)
    np.testing.assert_allclose(sample(depth, 211), encoder_golden["depth_s211"], **TOL)
    np.testing.assert_allclose(sample(feat, 53), encoder_golden["intr_feat_s53"], **TOL)
    lo, hi = encoder_golden["depth_minmax"]
    assert 0.5 < lo < hi <= 1.0 and depth.shape == (2, 1, 224, 224) and feat.shape == (2, 768, 7, 7)


def test_graph_forward(encoder_golden, encoder_sd):
    rgb, mask = [torch.from_numpy(a) for a in syn.seeded_rgb_scene(seed=0, batch=2)]
    out = E.graph_forward(encoder_sd, rgb, mask)
    np.testing.assert_allclose(sample(out["depth_pred"], 211), encoder_golden["g_depth_pred_s211"], **TOL)
    np.testing.assert_allclose(out["intr_params"].numpy(), encoder_golden["g_intr_params"], **TOL)
    np.testing.assert_allclose(out["intr_pred"].numpy(), encoder_golden["g_intr_pred"], rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(sample(out["seen_points"], 101), encoder_golden["g_seen_points_s101"], rtol=1e-4,
                               atol=1e-4)
    assert list(out["latent_depth"].shape) == list(encoder_golden["g_latent_depth_shape"]) == [2, 197, 256]
    np.testing.assert_allclose(sample(out["latent_depth"], 37), encoder_golden["g_latent_depth_s37"], rtol=1e-4,
                               atol=1e-4)
    np.testing.assert_allclose(out["latent_depth"][:, 0].numpy(), encoder_golden["g_latent_global"], rtol=1e-4,
                               atol=1e-4)


def test_coord_enc_res(encoder_golden, encoder_sd):
    _, mask, _ = [torch.from_numpy(a) for a in syn.seeded_depth_scene(seed=1, batch=2)]
    coord = torch.from_numpy(np.random.RandomState(11).uniform(-1, 1, size=(2, 3, 224, 224)).astype(np.float32))
    lat = E.coord_enc_res(E._sub(encoder_sd, "coord_encoder."), coord, mask)
    np.testing.assert_allclose(sample(lat, 37), encoder_golden["res_latent_s37"], **TOL)


def test_coord_enc_att(encoder_golden, att_sd):
    _, mask, _ = [torch.from_numpy(a) for a in syn.seeded_depth_scene(seed=1, batch=2)]
    coord = torch.from_numpy(np.random.RandomState(12).uniform(-1, 1, size=(2, 112, 112, 3)).astype(np.float32))
    m = torch.nn.functional.interpolate(mask, (112, 112)) > 0.5
    lat = E.coord_enc_att(att_sd, coord, m[:, 0])
    assert list(lat.shape) == list(encoder_golden["att_latent_shape"]) == [2, 197, 256]
    np.testing.assert_allclose(sample(lat, 37), encoder_golden["att_latent_s37"], **TOL)


def test_state_dict_contract_lists_every_module(encoder_golden):
    keys = [str(k) for k in encoder_golden["graph_keys"]]
    for prefix in ("dpt_depth.pretrained.model.patch_embed.backbone.stem.conv.weight",
                   "dpt_depth.pretrained.model.blocks.11.mlp.fc2.bias", "dpt_depth.pretrained.act_postprocess4.4.bias",
                   "dpt_depth.scratch.refinenet1.out_conv.weight", "dpt_depth.scratch.output_conv.4.bias",
                   "intr_head.1.bn2.running_var", "intr_proj.weight", "coord_encoder.encoder.layer4.2.bn3.weight",
                   "coord_encoder.encoder.fc.2.bias", "coord_encoder.depth_feat_proj.2.weight",
                   "impl_network.point_embedding.weight" if False else "impl_network.pos_embed"):
        assert prefix in keys, prefix
