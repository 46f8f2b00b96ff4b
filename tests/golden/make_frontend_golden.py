#!/usr/bin/env python3
"""Generate tests/golden/frontend_golden.npz from the REAL reference (build container only).

Run:  python tests/golden/make_frontend_golden.py          (needs /root/reference)

Imports the reference's own functions for the seen-surface front-end and the depth metrics
  utils/camera.py::{unproj_depth, valid_norm_fac}, utils/util.py::{interpolate_coordmap,
  interpolate_depth, get_child_state_dict}, utils/eval_depth.py::DepthMetric,
  model/compute_graph/graph_shape.py::Graph.intr_param2mtx (called unbound: it does not use self)
feeds them the build-owned seeded inputs of zeroshape_amd/synthetic.py and stores expected
outputs only (strided samples + float64 checksums of the large maps).

graph_shape.py imports timm / torchvision model code at module scope; those packages are absent
here and are not used by intr_param2mtx, so they are replaced in this process by the empty
stand-ins of make_golden.py plus attribute-lenient module objects.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference"


class _Lenient(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (), {})


def _stubs():
    import make_golden
    make_golden._install_stubs()
    for n in ["timm", "timm.models", "timm.models.vision_transformer", "timm.models.layers", "torchvision",
              "torchvision.models"]:
        old = sys.modules.get(n)
        new = _Lenient(n)
        if old is not None:
            new.__dict__.update({k: v for k, v in old.__dict__.items() if not k.startswith("__")})
        sys.modules[n] = new


def sample(x, step):
    return np.ascontiguousarray(x.reshape(-1)[::step])


def main():
    assert os.path.isdir(REF), "reference tree not present: run in the build container"
    _stubs()
    sys.path.insert(0, REF)
    from model.compute_graph.graph_shape import Graph                     # noqa: E402 (reference)
    from utils.camera import unproj_depth, valid_norm_fac                 # noqa: E402 (reference)
    from utils.util import interpolate_coordmap, interpolate_depth, get_child_state_dict  # noqa: E402
    from utils.util import EasyDict as edict                              # noqa: E402 (reference)
    from utils.eval_depth import DepthMetric                              # noqa: E402 (reference)
    from zeroshape_amd import synthetic as syn

    torch.set_num_threads(8)
    out = {}
    opt = edict(dict(device="cpu", H=224, W=224))
    depth, mask, params = [torch.from_numpy(a) for a in syn.seeded_depth_scene(seed=0, batch=3)]

    intr = Graph.intr_param2mtx(None, opt, params)
    out["intr"] = intr.numpy()
    pts = unproj_depth(opt, depth, intr)
    out["unproj_s101"] = sample(pts.numpy(), 101)
    out["unproj_sum"] = np.array([pts.double().sum().item(), pts.double().abs().sum().item()])
    mean, scale = valid_norm_fac(pts, mask > 0.5)
    out["mean"], out["scale"] = mean.numpy(), scale.numpy()
    # graph_shape.py:139-144 around the reference's functions
    B = depth.shape[0]
    seen = (pts - mean.unsqueeze(1)) / scale.unsqueeze(-1).unsqueeze(-1)
    seen[(mask <= 0.5).view(B, -1)] = 0
    out["seen_s101"] = sample(seen.numpy(), 101)
    out["seen_sum"] = np.array([seen.double().sum().item(), seen.double().abs().sum().item()])
    seen_map = seen.view(B, 224, 224, 3).permute(0, 3, 1, 2).contiguous()
    for dsp in (1, 2):
        c, m = interpolate_coordmap(seen_map, mask, (224 // dsp, 224 // dsp))
        out["coord_dsp%d_s53" % dsp] = sample(c.numpy(), 53)
        out["coord_dsp%d_sum" % dsp] = np.array([c.double().sum().item(), c.double().abs().sum().item()])
        out["mask_dsp%d_bits" % dsp] = np.packbits(m.numpy().reshape(-1) > 0.5)
    # a non-integer ratio (224 -> 96) and interpolate_depth's background fill
    c, m = interpolate_coordmap(seen_map, mask, (96, 96))
    out["coord_96_s53"] = sample(c.numpy(), 53)
    out["mask_96_bits"] = np.packbits(m.numpy().reshape(-1) > 0.5)
    d, m = interpolate_depth(depth, mask, (112, 112))
    out["depth_112_s53"] = sample(d.numpy(), 53)
    out["depth_112_sum"] = np.array([d.double().sum().item()])

    # depth metrics
    pred, target, dmask = [torch.from_numpy(a) for a in syn.seeded_depth_pair(seed=0, batch=3)]
    for name, kw in (("plain", {}), ("cap", dict(depth_cap=1.5)), ("disp", dict(prediction_type="disparity")),
                     ("thr", dict(thresholds=[1.02, 1.05, 1.1, 1.4]))):
        dm = DepthMetric(**kw)
        p = 1.0 / pred if name == "disp" else pred
        metrics, aligned = dm.compute_metrics(p, target, dmask)
        out["dm_%s_keys" % name] = np.array(dm.metric_keys)
        out["dm_%s_vals" % name] = np.stack([metrics[k].numpy() for k in dm.metric_keys], 1)
        out["dm_%s_depth_s53" % name] = sample(aligned.numpy(), 53)
    v = (dmask[:, 0] > 0.5)
    pd = torch.where(v, 1.0 / (pred[:, 0] + 1e-6), torch.zeros(()))
    td = torch.where(v, 1.0 / target[:, 0], torch.zeros(()))
    s, t = DepthMetric().compute_scale_and_shift(pd, td, v.long())
    out["dm_scale_shift"] = torch.stack([s, t], 1).numpy()

    # checkpoint key helper
    sd = {"module.graph.a.w": 1, "graph.a.b.c": 2, "graphx.a": 3, "other.graph.a": 4, "graph.z": 5}
    out["child_keys"] = np.array(sorted(get_child_state_dict(sd, "graph").keys()))

    np.savez_compressed(os.path.join(HERE, "frontend_golden.npz"), **out)
    print("frontend_golden.npz: %d arrays, %d bytes" % (len(out), os.path.getsize(
        os.path.join(HERE, "frontend_golden.npz"))))
    print("scale", out["scale"], "metrics", out["dm_plain_vals"][0])


if __name__ == "__main__":
    main()
