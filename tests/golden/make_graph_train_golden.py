#!/usr/bin/env python3
"""Golden fixture for one TRAINING step of the shape graph, from the REAL reference (build
container only; needs /root/reference).

Builds the reference's model/compute_graph/graph_shape.py::Graph exactly like
make_encoder_golden.py (seeded parameters, stand-ins of oracle/standins.py for the un-vendored
timm / torchvision backbones), puts it in .train() mode and runs what Runner.train_iteration
(model/shape_engine.py:248-297) runs: Graph.forward(opt, var, training=True, get_loss=True) with
GT samples -> loss.shape -> backward().  Stored: the loss, pred_sample_occ, gt_points_cam samples,
the DropPath factors drawn by the decoder, gradient norms + strided samples of all 469 trainable
tensors, and the BatchNorm running statistics after the step for a few layers.
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

BN_WATCH = ["coord_encoder.encoder.bn1", "coord_encoder.encoder.layer3.2.bn3", "coord_encoder.depth_feat_proj.0.bn1",
            "coord_encoder.encoder.fc.1.bn2", "intr_head.0.bn1"]


def main():
    assert os.path.isdir(REF)
    import make_golden
    import make_train_golden as MT
    make_golden._install_stubs()
    from oracle import standins
    standins.install()
    sys.modules["timm.models.vision_transformer"].DropPath = MT.RecordingDropPath
    sys.path.insert(0, REF)
    from model.compute_graph.graph_shape import Graph                 # reference
    from utils.util import EasyDict as edict                          # reference
    import yaml
    from zeroshape_amd import synthetic as syn

    torch.manual_seed(0)
    torch.set_num_threads(8)
    opt = edict(yaml.safe_load(open(os.path.join(REF, "options/shape.yaml"))))
    opt.H, opt.W = opt.image_size
    opt.device = "cpu"
    opt.pretrain.depth = None
    opt.arch.depth.pretrained = None
    graph = Graph(opt)
    enc_gold = np.load(os.path.join(HERE, "encoder_golden.npz"))
    shapes = {k: tuple(v.shape) for k, v in graph.state_dict().items()}
    enc_shapes = {k: v for k, v in shapes.items() if not k.startswith("impl_network.")}
    sd = {k: torch.from_numpy(v) for k, v in syn.seeded_encoder_state_dict(enc_shapes, seed=0).items()}
    gain, offset = enc_gold["head_calibration"]
    sd["dpt_depth.scratch.output_conv.4.weight"] = sd["dpt_depth.scratch.output_conv.4.weight"] * float(gain)
    sd["dpt_depth.scratch.output_conv.4.bias"] = torch.full_like(sd["dpt_depth.scratch.output_conv.4.bias"], float(offset))
    dec = syn.seeded_state_dict(seed=0, pos_embed=np.load(os.path.join(HERE, "decoder_golden.npz"))["pos_embed_f32"])
    sd.update({"impl_network." + k: torch.from_numpy(v) for k, v in dec.items()})
    graph.load_state_dict(sd, strict=True)
    graph.train()

    # batch 4 (two seeded scenes of two): with 2 samples BatchNorm over the 1x1 global token is a sign
    # function of (x1 - x2) and a 1e-6 input difference flips channels - not a usable fixture
    B, N = 4, 256
    scenes = [syn.seeded_rgb_scene(seed=s, batch=2) for s in (0, 1)]
    rgb, mask = [torch.from_numpy(np.concatenate([sc[i] for sc in scenes])) for i in (0, 1)]
    gts = [syn.seeded_depth_scene(seed=s, batch=2) for s in (3, 4)]
    depth_gt, intr_params = [np.concatenate([g_[i] for g_ in gts]) for i in (0, 2)]
    rs = np.random.RandomState(21)
    pts = rs.uniform(-0.6, 0.6, (B, N, 3)).astype(np.float32)
    sdf = (np.linalg.norm(pts, axis=-1) - 0.45).astype(np.float32)
    pose = np.tile(np.concatenate([np.eye(3), np.array([[0.0], [0.0], [1.2]])], 1)[None], (B, 1, 1)).astype(np.float32)
    with torch.no_grad():
        intr_gt = graph.intr_param2mtx(opt, torch.from_numpy(intr_params))
    var = edict(dict(idx=torch.arange(B), rgb_input_map=rgb, mask_input_map=mask, pose_gt=torch.from_numpy(pose),
                     depth_input_map=torch.from_numpy(depth_gt), intr=intr_gt, gt_sample_points=torch.from_numpy(pts),
                     gt_sample_sdf=torch.from_numpy(sdf)))
    torch.manual_seed(5)
    MT.DRAWN.clear()
    var, loss = graph.forward(opt, var, training=True, get_loss=True)
    assert set(loss.keys()) == {"shape"}
    loss.shape.backward()

    out = dict(batch=np.int64(B), gt_sample_points=pts, gt_sample_sdf=sdf, pose_gt=pose, depth_seed=np.int64(3),
               intr_gt=intr_gt.numpy(), drop_scales=torch.stack(MT.DRAWN).numpy(), loss=np.float32(loss.shape.item()),
               pred_sample_occ=var.pred_sample_occ.detach().numpy(), gt_points_cam_s7=var.gt_points_cam.numpy().reshape(-1)[::7],
               latent_s211=var.latent_depth.detach().numpy().reshape(-1)[::211],
               depth_pred_s211=var.depth_pred.detach().numpy().reshape(-1)[::211], intr_pred=var.intr_pred.detach().numpy())
    n = 0
    for name, p in graph.named_parameters():
        if p.grad is None:
            continue
        g = p.grad
        out["gnorm/" + name] = np.float64(g.double().norm().item())
        out["gs/" + name] = g.reshape(-1)[::max(1, g.numel() // 64)][:64].numpy().copy()
        n += 1
    sdn = graph.state_dict()
    for p in BN_WATCH:
        out["bn/" + p + ".running_mean"] = sdn[p + ".running_mean"].numpy().copy()
        out["bn/" + p + ".running_var"] = sdn[p + ".running_var"].numpy().copy()
        out["bn/" + p + ".num_batches_tracked"] = sdn[p + ".num_batches_tracked"].numpy().copy()
    path = os.path.join(HERE, "graph_train_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; loss", loss.shape.item(), "tensors with grad", n,
          "drop", torch.stack(MT.DRAWN).tolist())


if __name__ == "__main__":
    main()
