"""Time the dense-grid query (129^3 points by default) in both decoder arithmetics and report
their agreement.  python tools/bench_precision.py [--vox 128] [--steps 5]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshape_amd import synthetic as syn                      # noqa: E402
from zeroshape_amd.model.shape.implicit import Implicit         # noqa: E402
from zeroshape_amd.utils.pos_embed import get_2d_sincos_pos_embed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vox", type=int, default=128)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--modes", default="f32,f16x3")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    pe = get_2d_sincos_pos_embed(256, 14, cls_token=True).astype(np.float32)
    sd = {k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(0, pos_embed=pe).items()}
    net = Implicit(syn.NUM_PATCHES, latent_dim=256, n_channels=256, n_blocks_attn=2, n_layers_mlp=8,
                   num_heads=8, skip_in=[2, 4, 6], pos_perlayer=False)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    latent = torch.from_numpy(syn.seeded_latent(0, 1)).to(dev)
    axis = torch.linspace(-1.5, 1.5, a.vox + 1, device=dev)
    outs = {}
    for mode in a.modes.split(","):
        st = net.prepare(latent, mode)
        out = net.query_grid(latent, axis, apply_sigmoid=False, state=st)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
        ev[0].record()
        for i in range(a.steps):
            out = net.query_grid(latent, axis, apply_sigmoid=False, state=st)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps)]
        outs[mode] = out
        pts = out.numel()
        print(json.dumps({"mode": mode, "vox": a.vox, "points": pts, "ms_min": min(ms), "ms_mean": sum(ms) / len(ms),
                          "points_per_s": pts / (min(ms) * 1e-3), "finite": bool(torch.isfinite(out).all())}), flush=True)
    if "f32" in outs and "f16x3" in outs:
        d = (outs["f32"] - outs["f16x3"]).abs()
        flips = (outs["f32"] > 0) != (outs["f16x3"] > 0)
        print(json.dumps({"max_abs_diff": float(d.max()), "mean_abs_diff": float(d.mean()), "flips": int(flips.sum()),
                          "max_abs_logit_at_flip": float(outs["f32"][flips].abs().max()) if int(flips.sum()) else 0.0}))


if __name__ == "__main__":
    main()
