"""Oracle of the seen-surface front-end / depth metrics vs golden outputs of the real reference
(tests/golden/make_frontend_golden.py).  Same torch-CPU ops as the reference, so equality is
exact."""
import numpy as np
import torch

from oracle import frontend_ref as F
from zeroshape_amd import synthetic as syn


def sample(x, step):
    return x.numpy().reshape(-1)[::step]


def scene():
    return [torch.from_numpy(a) for a in syn.seeded_depth_scene(seed=0, batch=3)]


def test_intr_param2mtx(frontend_golden):
    _, _, params = scene()
    np.testing.assert_array_equal(F.intr_param2mtx(224, 224, params).numpy(), frontend_golden["intr"])


def test_unproj_and_norm_fac(frontend_golden):
    depth, mask, _ = scene()
    intr = torch.from_numpy(frontend_golden["intr"])
    pts = F.unproj_depth(depth, intr)
    np.testing.assert_array_equal(sample(pts, 101), frontend_golden["unproj_s101"])
    assert pts.double().sum().item() == frontend_golden["unproj_sum"][0]
    mean, scale = F.valid_norm_fac(pts, mask > 0.5)
    np.testing.assert_array_equal(mean.numpy(), frontend_golden["mean"])
    np.testing.assert_array_equal(scale.numpy(), frontend_golden["scale"])


def test_seen_surface_chain(frontend_golden):
    depth, mask, _ = scene()
    intr = torch.from_numpy(frontend_golden["intr"])
    for dsp in (1, 2):
        seen, coord, mdsp, mean, scale = F.seen_surface(depth, intr, mask, dsp)
        np.testing.assert_array_equal(sample(seen, 101), frontend_golden["seen_s101"])
        np.testing.assert_array_equal(sample(coord, 53), frontend_golden["coord_dsp%d_s53" % dsp])
        np.testing.assert_array_equal(np.packbits(mdsp.numpy().reshape(-1) > 0.5),
                                      frontend_golden["mask_dsp%d_bits" % dsp])
        assert coord.double().abs().sum().item() == frontend_golden["coord_dsp%d_sum" % dsp][1]
    # invalid pixels are exactly zero, valid ones lie in the unit ball and one touches the sphere
    r = seen.norm(dim=-1)
    assert torch.all(r[(mask <= 0.5).view(3, -1)] == 0)
    assert abs(r.max().item() - 1) < 1e-6


def test_masked_resample_other_sizes(frontend_golden):
    depth, mask, _ = scene()
    intr = torch.from_numpy(frontend_golden["intr"])
    seen = F.seen_surface(depth, intr, mask, 1)[0]
    seen_map = seen.view(3, 224, 224, 3).permute(0, 3, 1, 2).contiguous()
    c, m = F.masked_resample(seen_map, mask, (96, 96))
    np.testing.assert_array_equal(sample(c, 53), frontend_golden["coord_96_s53"])
    np.testing.assert_array_equal(np.packbits(m.numpy().reshape(-1) > 0.5), frontend_golden["mask_96_bits"])
    d, m = F.masked_resample(depth, mask, (112, 112), bg=20)
    np.testing.assert_array_equal(sample(d, 53), frontend_golden["depth_112_s53"])


def test_depth_metrics(frontend_golden):
    pred, target, mask = [torch.from_numpy(a) for a in syn.seeded_depth_pair(seed=0, batch=3)]
    cases = (("plain", {}), ("cap", dict(depth_cap=1.5)), ("disp", dict(prediction_type="disparity")),
             ("thr", dict(thresholds=[1.02, 1.05, 1.1, 1.4])))
    for name, kw in cases:
        p = 1.0 / pred if name == "disp" else pred
        metrics, aligned = F.depth_metrics(p, target, mask, **kw)
        keys = list(frontend_golden["dm_%s_keys" % name])
        assert list(metrics.keys()) == keys
        vals = np.stack([metrics[k].numpy() for k in keys], 1)
        np.testing.assert_array_equal(vals, frontend_golden["dm_%s_vals" % name])
        np.testing.assert_array_equal(sample(aligned, 53), frontend_golden["dm_%s_depth_s53" % name])
    v = mask[:, 0] > 0.5
    pd = torch.where(v, 1.0 / (pred[:, 0] + 1e-6), torch.zeros(()))
    td = torch.where(v, 1.0 / target[:, 0], torch.zeros(()))
    s, t = F.scale_and_shift(pd, td, v.long())
    np.testing.assert_array_equal(torch.stack([s, t], 1).numpy(), frontend_golden["dm_scale_shift"])


def test_get_child_state_dict(frontend_golden):
    from zeroshape_amd.utils.util import get_child_state_dict
    sd = {"module.graph.a.w": 1, "graph.a.b.c": 2, "graphx.a": 3, "other.graph.a": 4, "graph.z": 5}
    child = get_child_state_dict(sd, "graph")
    assert sorted(child.keys()) == list(frontend_golden["child_keys"])
    assert child == {"a.w": 1, "a.b.c": 2, "z": 5}
