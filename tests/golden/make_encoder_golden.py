#!/usr/bin/env python3
"""Generate tests/golden/encoder_golden.npz from the REAL reference encoders (build container).

Run:  python tests/golden/make_encoder_golden.py          (needs /root/reference)

Builds the reference's own model/compute_graph/graph_shape.py::Graph (DPTDepthModel,
intrinsics head, CoordEncRes, Implicit) and model/shape/seen_coord_enc.py::CoordEncAtt from
/root/reference, on top of oracle/standins.py for the two un-vendored packages (timm,
torchvision: published architectures restated, see that file), loads build-owned seeded
parameters (zeroshape_amd/synthetic.py::seeded_encoder_state_dict), runs them in eval mode on
seeded 224x224 inputs and stores
  * the state-dict key/shape contract of every module (names a real checkpoint must match),
  * strided samples + checksums of depth_pred, the intrinsics feature, intr_pred, seen_points,
    latent_depth and a few DPT intermediates captured with forward hooks.
The last head convolution of the seeded DPT is rescaled so the depth map is not saturated by
the clamp to [0,1]; the two calibration scalars are stored and re-applied by the tests.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference"


def sample(x, step):
    return np.ascontiguousarray(x.detach().numpy().reshape(-1)[::step])


def checksum(x):
    x = x.detach().double()
    return np.array([x.sum().item(), x.abs().sum().item()])


def shapes_of(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def main():
    assert os.path.isdir(REF), "reference tree not present: run in the build container"
    import make_golden
    make_golden._install_stubs()                      # render / mcubes / chamfer names only
    from oracle import standins
    standins.install()
    sys.path.insert(0, REF)
    from model.compute_graph.graph_shape import Graph                 # noqa: E402 (reference)
    from model.shape.seen_coord_enc import CoordEncAtt                # noqa: E402 (reference)
    from utils.util import EasyDict as edict                          # noqa: E402 (reference)
    import yaml
    from zeroshape_amd import synthetic as syn

    torch.manual_seed(0)
    torch.set_num_threads(8)
    opt = edict(yaml.safe_load(open(os.path.join(REF, "options/shape.yaml"))))
    opt.H, opt.W = opt.image_size
    opt.device = "cpu"
    opt.pretrain.depth = None
    opt.arch.depth.pretrained = None
    graph = Graph(opt).eval()
    out = {}

    shapes = shapes_of(graph)
    out["graph_keys"] = np.array(list(shapes.keys()))
    out["graph_shapes"] = np.array([",".join(map(str, s)) for s in shapes.values()])
    enc_shapes = {k: v for k, v in shapes.items() if not k.startswith("impl_network.")}
    sd = syn.seeded_encoder_state_dict(enc_shapes, seed=0)
    full = {k: torch.from_numpy(v) for k, v in sd.items()}
    # the decoder keeps the parameters of the decoder goldens
    dec = syn.seeded_state_dict(seed=0, pos_embed=np.load(os.path.join(HERE, "decoder_golden.npz"))["pos_embed_f32"])
    full.update({"impl_network." + k: torch.from_numpy(v) for k, v in dec.items()})
    graph.load_state_dict(full, strict=True)

    rgb, mask = [torch.from_numpy(a) for a in syn.seeded_rgb_scene(seed=0, batch=2)]

    # ---- calibrate the last head conv so depth lands in (0.6, 1.0) before the clamp ----
    head = graph.dpt_depth.scratch.output_conv
    pre = {}
    h = head[3].register_forward_hook(lambda m, i, o: pre.__setitem__("x", o.detach().clone()))
    with torch.no_grad():
        graph.dpt_depth(rgb)
    h.remove()
    raw = torch.nn.functional.conv2d(pre["x"], head[4].weight, None)
    lo, hi = raw.min().item(), raw.max().item()
    gain = 0.4 / max(hi - lo, 1e-6)
    offset = 0.6 - lo * gain
    with torch.no_grad():
        head[4].weight.mul_(gain)
        head[4].bias.fill_(offset)
    out["head_calibration"] = np.array([gain, offset], np.float64)

    # ---- DPT with intermediates ----
    taps = {}
    hooks = []
    def tap(name, module):
        hooks.append(module.register_forward_hook(lambda m, i, o: taps.__setitem__(name, o.detach().clone())))
    sc = graph.dpt_depth.scratch
    pm = graph.dpt_depth.pretrained.model
    tap("stem", pm.patch_embed.backbone.stem)
    tap("stage0", pm.patch_embed.backbone.stages[0])
    tap("stage1", pm.patch_embed.backbone.stages[1])
    tap("stage2", pm.patch_embed.backbone.stages[2])
    tap("block0", pm.blocks[0])
    tap("block8", pm.blocks[8])
    tap("block11", pm.blocks[11])
    tap("layer3_rn", sc.layer3_rn)
    tap("layer4_rn", sc.layer4_rn)
    tap("path4", sc.refinenet4)
    tap("path3", sc.refinenet3)
    tap("path2", sc.refinenet2)
    tap("path1", sc.refinenet1)
    with torch.no_grad():
        depth, feat = graph.dpt_depth(rgb, get_feat=True)
    for hk in hooks:
        hk.remove()
    for name, t in taps.items():
        out["dpt_%s_s997" % name] = sample(t, 997)
        out["dpt_%s_sum" % name] = checksum(t)
        out["dpt_%s_shape" % name] = np.array(t.shape)
    out["depth_s211"] = sample(depth, 211)
    out["depth_sum"] = checksum(depth)
    out["depth_minmax"] = np.array([depth.min().item(), depth.max().item()])
    out["intr_feat_s53"] = sample(feat, 53)
    out["intr_feat_sum"] = checksum(feat)

    # ---- whole Graph.forward (graph_shape.py:115-192), inference branch ----
    var = edict(dict(idx=torch.arange(2), rgb_input_map=rgb, mask_input_map=mask, pose_gt=torch.zeros(2, 3, 4)))
    with torch.no_grad():
        var = graph.forward(opt, var, training=False, get_loss=False)
    out["g_depth_pred_s211"] = sample(var.depth_pred, 211)
    out["g_intr_pred"] = var.intr_pred.numpy()
    out["g_seen_points_s101"] = sample(var.seen_points, 101)
    out["g_seen_points_sum"] = checksum(var.seen_points)
    out["g_latent_depth_s37"] = sample(var.latent_depth, 37)
    out["g_latent_depth_sum"] = checksum(var.latent_depth)
    out["g_latent_depth_shape"] = np.array(var.latent_depth.shape)
    out["g_latent_global"] = var.latent_depth[:, 0].numpy()
    # intr_proj is zero-initialised by the reference (graph_shape.py:26-28) but seeded here, so
    # intr_pred exercises the head; record the raw 3 parameters too
    with torch.no_grad():
        f = graph.intr_pool(graph.intr_head(feat)).squeeze(-1).squeeze(-1)
        out["g_intr_params"] = graph.intr_proj(f).numpy()

    # ---- CoordEncRes alone on a seeded coordinate map ----
    depth_s, mask_s, params = [torch.from_numpy(a) for a in syn.seeded_depth_scene(seed=1, batch=2)]
    coord = torch.from_numpy(np.random.RandomState(11).uniform(-1, 1, size=(2, 3, 224, 224)).astype(np.float32))
    with torch.no_grad():
        lat = graph.coord_encoder(coord, mask_s)
    out["res_latent_s37"] = sample(lat, 37)
    out["res_latent_sum"] = checksum(lat)

    # ---- CoordEncAtt (alternative encoder, options: arch.depth.encoder != resnet, dsp 2) ----
    att = CoordEncAtt(embed_dim=256, n_blocks=12, num_heads=8, win_size=16 // 2).eval()
    a_shapes = shapes_of(att)
    out["att_keys"] = np.array(list(a_shapes.keys()))
    out["att_shapes"] = np.array([",".join(map(str, s)) for s in a_shapes.values()])
    a_sd = syn.seeded_encoder_state_dict(a_shapes, seed=1)
    att.load_state_dict({k: torch.from_numpy(v) for k, v in a_sd.items()}, strict=True)
    coord112 = torch.from_numpy(np.random.RandomState(12).uniform(-1, 1, size=(2, 112, 112, 3)).astype(np.float32))
    mask112 = torch.nn.functional.interpolate(mask_s, (112, 112)) > 0.5
    with torch.no_grad():
        lat = att(coord112, mask112[:, 0])
    out["att_latent_s37"] = sample(lat, 37)
    out["att_latent_sum"] = checksum(lat)
    out["att_latent_shape"] = np.array(lat.shape)

    path = os.path.join(HERE, "encoder_golden.npz")
    np.savez_compressed(path, **out)
    print("encoder_golden.npz: %d arrays, %d bytes" % (len(out), os.path.getsize(path)))
    print("depth range", out["depth_minmax"], "intr", out["g_intr_pred"][0].tolist())
    print("latent |mean|", var.latent_depth.abs().mean().item(), "stage2 |mean|", taps["stage2"].abs().mean().item(),
          "block11 |mean|", taps["block11"].abs().mean().item(), "path1 |mean|", taps["path1"].abs().mean().item())


if __name__ == "__main__":
    main()
