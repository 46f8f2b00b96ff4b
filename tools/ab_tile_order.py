#!/usr/bin/env python3
"""static vs dynamic tile order of the split-fp16 decoder in ONE process on ONE box.

ZS_SPLIT_STATIC_TILES is read per launch, so the arms alternate block by block (blocks of `--block` launches, event-timed
on the launch stream); socket power and engine clock are sampled with rocm-smi per arm over a separate back-to-back run of
each.  Vox 128 (129^3 points, 16,771 tiles on 256 workgroups) and vox 64 (2,146 tiles: 98 of 256 workgroups get a ninth
tile in the static deal), with and without the per-image check on its side streams.

    python tools/ab_tile_order.py [--launches 24] [--block 4] > profiles/r05_tile_order_ab.txt
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B                                             # noqa: E402  (power_under_load)
from zeroshape_amd import synthetic as syn                     # noqa: E402
from zeroshape_amd.model.shape.implicit import Implicit        # noqa: E402
from zeroshape_amd.utils.pos_embed import get_2d_sincos_pos_embed   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=24)
    ap.add_argument("--block", type=int, default=4)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    pe = get_2d_sincos_pos_embed(256, 14, cls_token=True).astype(np.float32)
    sd = {k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(0, pos_embed=pe).items()}
    net = Implicit(syn.NUM_PATCHES, latent_dim=256, n_channels=256, n_blocks_attn=2, n_layers_mlp=8, num_heads=8,
                   skip_in=[2, 4, 6], pos_perlayer=False)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    latent = torch.from_numpy(syn.seeded_latent(0, 1)).to(dev)
    stream = torch.cuda.current_stream(dev)
    print("# static (ZS_SPLIT_STATIC_TILES=1) vs dynamic tile order, one process, alternating blocks of %d launches, %d launches per arm"
          % (a.block, a.launches))
    print("# launch = zs_sdf_query_grid_range_split of the whole grid between two HIP events; step = prepare() + launch (host clock)")
    print("%-7s %-22s %-8s %9s %9s %9s | %8s %8s" % ("vox", "leg", "arm", "mean ms", "median", "min", "socket W", "sclk MHz"))
    for N in (128, 64):
        G = N + 1
        axis = torch.linspace(-1.5, 1.5, G, device=dev)
        for leg, check in (("launch only", False), ("step with image check", True)):
            net.image_check = check
            st = net.prepare(latent)
            times = {"static": [], "dynamic": []}

            def one(arm):
                if arm == "static":
                    os.environ["ZS_SPLIT_STATIC_TILES"] = "1"
                else:
                    os.environ.pop("ZS_SPLIT_STATIC_TILES", None)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                s = net.prepare(latent) if check else st
                net.query_grid_range(latent, axis, 0, G ** 3, apply_sigmoid=True, state=s)
                e1.record(stream)
                return e0, e1
            for arm in ("static", "dynamic"):
                for _ in range(3):
                    one(arm)
            torch.cuda.synchronize()
            evs = []
            for blk in range(a.launches // a.block):
                for arm in (("static", "dynamic") if blk % 2 == 0 else ("dynamic", "static")):
                    for _ in range(a.block):
                        evs.append((arm,) + one(arm))
            torch.cuda.synchronize()
            for arm, e0, e1 in evs:
                times[arm].append(e0.elapsed_time(e1))
            for arm in ("static", "dynamic"):
                if arm == "static":
                    os.environ["ZS_SPLIT_STATIC_TILES"] = "1"
                else:
                    os.environ.pop("ZS_SPLIT_STATIC_TILES", None)
                pw = B.power_under_load(lambda: net.query_grid_range(latent, axis, 0, G ** 3, apply_sigmoid=True,
                                                                     state=(net.prepare(latent) if check else st)), seconds=1.2) or {}
                t = sorted(times[arm])
                print("%-7d %-22s %-8s %9.3f %9.3f %9.3f | %8s %8s" % (N, leg, arm, sum(t) / len(t), t[len(t) // 2], t[0],
                                                                       pw.get("socket_w", "-"), pw.get("sclk_mhz", "-")), flush=True)
            os.environ.pop("ZS_SPLIT_STATIC_TILES", None)
            s_, d_ = sorted(times["static"]), sorted(times["dynamic"])
            print("#   dynamic / static (median): %.4f" % (d_[len(d_) // 2] / s_[len(s_) // 2]), flush=True)


if __name__ == "__main__":
    main()
