#!/usr/bin/env python3
"""Batch-1 (or --batch N) encoder half of Graph.forward as one replayed hipGraph: ms per forward (HIP events), C-ABI launches
per forward, and the outputs' max difference from the single-stream eager run.
    python tools/enc_b1.py [--batch 1] [--reps 30] [--encoder resnet|att]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_legs import _events, _graph                                # noqa: E402
from zeroshape_amd import _lib, synthetic as syn                            # noqa: E402
from zeroshape_amd.nn import branch                                          # noqa: E402
from zeroshape_amd.utils.options import EasyDict as edict                    # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--encoder", default="resnet")
args = ap.parse_args()
dev = torch.device("cuda:0")
opt, g = _graph(dev, encoder=args.encoder) if "encoder" in _graph.__code__.co_varnames else _graph(dev)
B = args.batch
rgb, mask = [torch.from_numpy(x).to(dev) for x in syn.seeded_rgb_scene(0, B)]
var = edict(dict(idx=list(range(B)), rgb_input_map=rgb, mask_input_map=mask))


def fwd():
    v = g.forward(opt, var, training=False, get_loss=False)
    return v[0] if isinstance(v, tuple) else v


# reference: eager, one stream
on = branch.ENABLED
branch.ENABLED = False
g.enable_hip_graph(False)
ref = fwd()
ref = {k: ref[k].clone() for k in ("depth_pred", "intr_pred", "seen_points", "latent_depth")}
n0 = _lib.launch_count() if hasattr(_lib, "launch_count") else None
branch.ENABLED = on
out = {}
for label, br in (("one stream", False), ("branches", True)):
    if br and not on:
        continue
    branch.ENABLED = br
    if br and branch.KINDS == 0:
        branch.KINDS = 15         # every branch kind (ZS_BRANCH_KINDS defaults to 0: without this both legs time the same graph)
    g.enable_hip_graph(True)
    ms, mn = _events(fwd, args.reps)
    v = fwd()
    err = max(float((v[k] - ref[k]).abs().max()) for k in ref)
    print("%-12s batch %d: %.3f ms per forward (min %.3f), max |diff| vs eager single stream %.2e" % (label, B, ms, mn, err))
    g.enable_hip_graph(False)
