"""End-to-end parity of BASELINE config 2 (VERDICT r02 "next" 1-i): seeded RGB + mask ->
Graph.forward -> compute_level_grid at vox_res 64, against the CPU oracle of the SAME chain
(oracle/encoder_ref.graph_forward + oracle/decoder_ref.level_grid, i.e. the reference's
graph_shape.py:115-150 -> utils/eval_3D.py:22-45), for every combination of the encoder and
decoder arithmetics {f16x3, f32} x {f16x3, f32}.

Bars (north_star): max |occupancy difference| <= 1e-4 on full x-slices; the voxel index set
(occ > 0.5) equal except where the oracle's own value lies within the measured error of the level
(a flip "inside the band": two fp32 evaluation orders cannot agree on those either)."""
import numpy as np
import pytest
import torch

from oracle import decoder_ref as R
from oracle import encoder_ref as E
from tests.test_encoder_contract import make_opt
from zeroshape_amd import synthetic as syn
from zeroshape_amd.nn import ops
from zeroshape_amd.utils import eval_3D
from zeroshape_amd.utils.options import EasyDict as edict

pytestmark = pytest.mark.gpu

VOX = 64
SLICES = (0, 7, 16, 24, 32, 33, 41, 48, 57, 64)     # >= 8 full x-slices of the 65^3 grid, centre included
TOL = 1e-4
LOGIT_GAIN = 50.0


@pytest.fixture(scope="module")
def scene():
    rgb, mask = [torch.from_numpy(a) for a in syn.seeded_rgb_scene(seed=3, batch=1)]
    return rgb, mask


@pytest.fixture(scope="module")
def oracle_latent(scene, encoder_sd):
    rgb, mask = scene
    return E.graph_forward(encoder_sd, rgb, mask)


@pytest.fixture(scope="module")
def decoder_sd(seeded_sd, oracle_latent):
    """The seeded decoder with its last layer calibrated like a trained network's: the seeded weights put every
    logit of this scene in [-0.40, -0.24] (nothing occupied, no level set to compare).  The 256 -> 1 layer is linear,
    so weight * s and bias = -s * (median raw logit of the centre slice - bias) turn the logits into
    s * (raw - median): the level set passes through the middle of the grid and the logits span about +-4
    (occupancies 0.02 .. 0.98), which also multiplies the sensitivity to the latent by s."""
    sd = {k: v.clone() for k, v in seeded_sd.items()}
    grid = R.dense_grid(-1.5, 1.5, VOX)
    raw, _ = R.implicit_forward(sd, oracle_latent["latent_depth"], grid[:, VOX // 2].reshape(1, -1, 3))
    w, b = "impl_mlp.layers.8.weight", "impl_mlp.layers.8.bias"
    sd[b] = -LOGIT_GAIN * (raw.median() - sd[b])
    sd[w] = sd[w] * LOGIT_GAIN
    return sd


@pytest.fixture(scope="module")
def graph(encoder_sd, decoder_sd):
    from zeroshape_amd.model.compute_graph.graph_shape import Graph
    g = Graph(make_opt())
    full = dict(encoder_sd)
    full.update({"impl_network." + k: v for k, v in decoder_sd.items()})
    g.load_state_dict(full, strict=True)
    return g.cuda().eval()


@pytest.fixture(scope="module")
def oracle_chain(oracle_latent, decoder_sd):
    """The reference chain on the CPU: latent from the oracle graph, occupancy on SLICES."""
    want, seeded_sd = oracle_latent, decoder_sd
    grid = R.dense_grid(-1.5, 1.5, VOX)                   # [1,G,G,G,3]
    pts = grid[:, list(SLICES)]
    logits = []
    for i in range(len(SLICES)):                          # utils/eval_3D.py:39-45: one impl_network call per slice
        lg, _ = R.implicit_forward(seeded_sd, want["latent_depth"], pts[:, i].reshape(1, -1, 3))
        logits.append(lg)
    logits = torch.stack(logits, 1).view(1, len(SLICES), VOX + 1, VOX + 1)
    return want, logits, torch.sigmoid(logits)


def _run(graph, scene, enc_prec, dec_prec):
    rgb, mask = scene
    opt = make_opt()
    opt.arch.depth.dsp = 1
    opt.device = "cuda"
    opt.eval = edict(dict(vox_res=VOX, range=[-1.5, 1.5]))
    prev_enc, prev_dec = ops.CONV_PRECISION, graph.impl_network.precision
    ops.set_conv_precision(enc_prec)
    graph.impl_network.precision = dec_prec
    try:
        var = edict(dict(idx=[0], rgb_input_map=rgb.cuda(), mask_input_map=mask.cuda()))
        var = graph.forward(opt, var, training=False, get_loss=False)
        points = eval_3D.get_dense_3D_grid(opt, var)
        occ, _ = eval_3D.compute_level_grid(opt, graph.impl_network, var.latent_depth, var.latent_semantic, points,
                                            var.rgb_input_map)
    finally:
        ops.set_conv_precision(prev_enc)
        graph.impl_network.precision = prev_dec
    return var, occ


@pytest.mark.parametrize("enc_prec,dec_prec", [("f32", "f32"), ("f32", "f16x3"), ("f16x3", "f32"), ("f16x3", "f16x3")])
def test_image_to_occupancy_vs_oracle(graph, scene, oracle_chain, enc_prec, dec_prec):
    want, want_logits, want_occ = oracle_chain
    var, occ = _run(graph, scene, enc_prec, dec_prec)
    assert occ.shape == (1, VOX + 1, VOX + 1, VOX + 1)
    got = occ[:, list(SLICES)].cpu()
    lat_err = float((var.latent_depth.cpu() - want["latent_depth"]).abs().max())
    lat_scale = float(want["latent_depth"].abs().max())
    err = (got - want_occ).abs()
    flips = (got > 0.5) != (want_occ > 0.5)
    band = float((want_occ[flips] - 0.5).abs().max()) if bool(flips.any()) else 0.0
    print("image->occupancy enc=%s dec=%s: latent max|err| %.2e (scale %.2f), occ max|err| %.2e mean %.2e, "
          "flips %d of %d (furthest from the level %.2e), occupied %.3f, logit range [%.1f, %.1f]" %
          (enc_prec, dec_prec, lat_err, lat_scale, float(err.max()), float(err.mean()), int(flips.sum()),
           flips.numel(), band, float((want_occ > 0.5).float().mean()), float(want_logits.min()),
           float(want_logits.max())))
    assert float(err.max()) <= TOL, "occupancy differs from the oracle chain by %.3g" % float(err.max())
    # a voxel may change side only where the oracle's own value is within the error of the level
    assert band <= TOL
    assert 0.0 < float((want_occ > 0.5).float().mean()) < 1.0, "degenerate scene: the level set must cross the grid"


def test_default_arithmetics_are_the_tested_ones(graph):
    """The defaults the engine runs with are two of the four combinations above."""
    assert ops.CONV_PRECISION in ("f16x3", "f32") and graph.impl_network.precision in ("f16x3", "f32")
