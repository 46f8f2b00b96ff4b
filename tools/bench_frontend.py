#!/usr/bin/env python3
"""Time the seen-surface front-end and the depth metrics (one fused launch each) next to the
same chain written as the torch ops the reference issues (oracle/frontend_ref.py moved to the
GPU is NOT used: the comparison below re-states the op sequence with device tensors).

    python tools/bench_frontend.py [--batch 28] [--iters 50]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshape_amd import synthetic as syn                      # noqa: E402
from zeroshape_amd.utils import camera as C                      # noqa: E402
from zeroshape_amd.utils.eval_depth import DepthMetric           # noqa: E402
from zeroshape_amd.utils.options import EasyDict as edict        # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def torch_chain(depth, intr, mask):
    """The reference's op sequence (graph_shape.py:131-144) with device tensors."""
    import torch.nn.functional as F
    B, _, H, W = depth.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=depth.device),
                            torch.arange(W, dtype=torch.float32, device=depth.device), indexing="ij")
    pix = torch.stack([xs, ys, torch.ones_like(xs)], -1).view(1, -1, 3).repeat(B, 1, 1)
    pts = (torch.linalg.inv(intr) @ pix.permute(0, 2, 1)).permute(0, 2, 1) * depth.view(B, H * W, 1)
    sel = (mask > 0.5).view(B, -1)
    means, radii = [], []
    for b in range(B):
        p = pts[b][sel[b]]
        mu = p.mean(0)
        means.append(mu)
        radii.append((p - mu).norm(dim=1).max())
    mean, scale = torch.stack(means), torch.stack(radii)
    seen = (pts - mean[:, None]) / scale[:, None, None]
    seen[~sel] = 0
    m = (mask > 0.5).float()
    smap = seen.view(B, H, W, 3).permute(0, 3, 1, 2).contiguous()
    num = F.interpolate(smap * m, (H, W), mode="bilinear", align_corners=False)
    den = F.interpolate(m, (H, W), mode="bilinear", align_corners=False)
    keep = (den > 0.5).float()
    return seen, num / (den + 1e-6) * keep, keep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[1, 28])
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    opt = edict(dict(device="cuda", H=224, W=224, arch=dict(depth=dict(dsp=1))))
    for B in a.batch:
        depth, mask, params = [torch.from_numpy(x).cuda() for x in syn.seeded_depth_scene(0, B)]
        intr = C.intr_param2mtx(opt, params)
        t_fused = timed(lambda: C.seen_surface(opt, depth, intr, mask, dsp=1), a.iters)
        t_torch = timed(lambda: torch_chain(depth, intr, mask), max(a.iters // 5, 3))
        bytes_ = B * 224 * 224 * (8 + 12 + 16)
        pred, target, dmask = [torch.from_numpy(x).cuda() for x in syn.seeded_depth_pair(0, B)]
        dm = DepthMetric()
        t_dm = timed(lambda: dm.compute_metrics(pred, target, dmask), a.iters)
        print("B=%d  seen_surface fused %.3f ms (%.1f GB/s algorithmic)  torch-op chain %.3f ms  |  "
              "depth_metrics %.3f ms" % (B, t_fused, bytes_ / t_fused / 1e6, t_torch, t_dm))


if __name__ == "__main__":
    main()
