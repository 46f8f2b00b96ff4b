#!/usr/bin/env python3
"""Time the encoder half of Graph.forward (DPT depth + intrinsics head + seen-surface geometry
+ coordinate encoder) on the HIP layers, per stage.

    python tools/bench_encoder.py [--batch 1 8] [--iters 10] [--encoder resnet|transformer]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshape_amd import synthetic as syn                                  # noqa: E402
from zeroshape_amd.model.compute_graph.graph_shape import Graph              # noqa: E402
from zeroshape_amd.utils import camera                                       # noqa: E402
from zeroshape_amd.utils.options import EasyDict as edict                    # noqa: E402


def make_opt(encoder):
    return edict(dict(H=224, W=224, device="cuda", pretrain=dict(depth=None),
                      arch=dict(num_heads=8, latent_dim=256, win_size=16,
                                depth=dict(encoder=encoder, n_blocks=12, dsp=2, pretrained=None),
                                rgb=dict(encoder=None, n_blocks=12),
                                impl=dict(n_channels=256, att_blocks=2, mlp_ratio=4., posenc_perlayer=False,
                                          mlp_layers=8, posenc_3D=0, skip_in=[2, 4, 6]))))


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


# multiply-accumulates per image at 224x224
GFLOP_DPT, GFLOP_RES = 2 * 41.3, 2 * 5.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[1, 8])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--encoder", default="resnet")
    a = ap.parse_args()
    opt = make_opt(a.encoder)
    torch.manual_seed(0)
    g = Graph(opt)
    with torch.no_grad():                      # any finite weights do: timing only
        torch.nn.init.normal_(g.intr_proj.weight, std=0.01)
    g = g.cuda().eval()
    for B in a.batch:
        rgb, mask = [torch.from_numpy(x).cuda() for x in syn.seeded_rgb_scene(0, B)]
        var = edict(dict(idx=list(range(B)), rgb_input_map=rgb, mask_input_map=mask))
        t_all = timed(lambda: g.forward(opt, var, training=False, get_loss=False), a.iters)
        g.enable_hip_graph(True)
        t_graph = timed(lambda: g.forward(opt, var, training=False, get_loss=False), a.iters)
        g.enable_hip_graph(False)
        t_dpt = timed(lambda: g.dpt_depth(rgb, get_feat=True), a.iters)
        depth, feat = g.dpt_depth(rgb, get_feat=True)
        t_intr = timed(lambda: g._intr.run(feat), a.iters)
        intr = g.intr_param2mtx(opt, g._intr.run(feat))
        t_geo = timed(lambda: camera.seen_surface(opt, depth, intr, mask, dsp=opt.arch.depth.dsp), a.iters)
        _, coord, mdsp, _, _ = camera.seen_surface(opt, depth, intr, mask, dsp=opt.arch.depth.dsp)
        if a.encoder == "resnet":
            t_enc = timed(lambda: g.coord_encoder(coord, mdsp), a.iters)
        else:
            c2, m2 = coord.permute(0, 2, 3, 1).contiguous(), mdsp.squeeze(1) > 0.5
            t_enc = timed(lambda: g.coord_encoder(c2, m2), a.iters)
        print("B=%d  Graph.forward %.2f ms eager, %.2f ms as one hipGraph (%.2f ms/image) | DPT %.2f ms (%.1f TFLOP/s) | intr head %.2f | "
              "geometry %.3f | coord encoder %.2f ms" % (B, t_all, t_graph, t_graph / B, t_dpt, GFLOP_DPT * B / t_dpt, t_intr,
                                                          t_geo, t_enc))


if __name__ == "__main__":
    main()
