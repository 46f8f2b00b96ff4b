#!/usr/bin/env python3
"""Pose-search timing and work counters on one GPU: the bench leg's three searches (tools/bench_legs.py:
pose_search_leg) for each nearest-neighbour kernel, plus the all-pairs scan on unsorted clouds' equivalent work.

    python tools/bench_pose.py [--nn cull,brute] [--reps 3]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nn", default="cull,pairs,brute")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    from zeroshape_amd import synthetic as syn
    from zeroshape_amd.utils import eval_3D as E
    dev = torch.device("cuda:0")
    n = 10000
    pred = torch.from_numpy(syn.ellipsoid_cloud(0, n)).to(dev)
    R = E._rotation_sphere(dev)
    g = torch.Generator(device="cpu").manual_seed(0)
    gt = ((R[1234] @ pred.T).T.contiguous().cpu() + 1e-3 * torch.randn(n, 3, generator=g)).to(dev)
    far = torch.from_numpy(syn.seeded_cloud(9, 1, n)[0]).to(dev)
    out = {}
    for nn in a.nn.split(","):
        res = {}
        for name, gt_, prune in (("exhaustive", gt, False), ("pruned", gt, True), ("unalignable_pruned", far, True),
                                 ("unalignable_exhaustive", far, False)):
            best = None
            for _ in range(a.reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                o = E.brute_force_search(pred, gt_, device=dev, prune=prune, return_index=True, nn=nn)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 1e3
                best = ms if best is None else min(best, ms)
            res[name] = {"ms": round(best, 2), "index": o[5], "cd": o[6], "evaluated": E.brute_force_search.last_evaluated,
                         "scanned_in_full": E.brute_force_search.last_scanned}
        out[nn] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
