#!/usr/bin/env python3
"""Pruned pose search (alignable ground truth) against the batch size of the exact evaluations."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshape_amd import synthetic as syn            # noqa: E402
from zeroshape_amd.utils import eval_3D as E          # noqa: E402

dev = torch.device("cuda:0")
n = 10000
pred = torch.from_numpy(syn.ellipsoid_cloud(0, n)).to(dev)
R = E._rotation_sphere(dev)
g = torch.Generator(device="cpu").manual_seed(0)
gt = ((R[1234] @ pred.T).T.contiguous().cpu() + 1e-3 * torch.randn(n, 3, generator=g)).to(dev)
far = torch.from_numpy(syn.seeded_cloud(9, 1, n)[0]).to(dev)
for name, target in (("alignable", gt), ("unrelated", far)):
    for bs in (192, 96, 48, 24, 12):
        E.brute_force_search(pred, target, device=dev, batch_size=bs)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            o = E.brute_force_search(pred, target, device=dev, batch_size=bs, return_index=True)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("%s batch %3d: %.2f ms, %d rotations evaluated, best %d cd %.6f" % (name, bs, min(ts), E.brute_force_search.last_evaluated,
                                                                            o[5], o[6]))
# the lower-bound pass alone
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    E._bf_lower_bounds(pred, E.normalize_pc(gt.unsqueeze(0))[0], R)
torch.cuda.synchronize()
print("lower bounds alone: %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
