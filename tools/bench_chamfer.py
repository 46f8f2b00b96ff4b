#!/usr/bin/env python3
"""Secondary benchmark: Chamfer-3D NN kernel and the 6912-rotation brute-force search
(BASELINE config 3).  Prints one JSON line; not the driver's bench (that is bench.py)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from zeroshape_amd import synthetic as syn
from zeroshape_amd.external.chamfer3D.dist_chamfer_3D import chamfer_3DDist
from zeroshape_amd.utils import eval_3D as E

PEAK_PAIRS = 157.3e12 / 8          # 8 flop per point pair on the fp32 vector roofline


def main():
    dev = torch.device("cuda:0")
    B, n, m = 24, 10000, 10000
    a = torch.from_numpy(syn.seeded_cloud(1, B, n)).to(dev)
    b = torch.from_numpy(syn.seeded_cloud(2, B, m)).to(dev)
    ch = chamfer_3DDist()
    for _ in range(3):
        ch(a, b)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    ev[0].record()
    for i in range(20):
        ch(a, b)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(20))
    call_ms = sum(ms) / len(ms)
    pairs = 2.0 * B * n * m
    # brute-force search, one sample
    pred = torch.from_numpy(syn.ellipsoid_cloud(0, n)).to(dev)
    R = E.get_rotation_sphere(24, 24, 12, device="cpu")
    gt = (R[1234] @ pred.cpu().T).T.contiguous() + 1e-3 * torch.randn(n, 3)
    E.brute_force_search(pred, gt, device=dev, return_index=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = E.brute_force_search(pred, gt, device=dev, return_index=True)
    torch.cuda.synchronize()
    bf_s = time.perf_counter() - t0
    print(json.dumps({
        "chamfer_call_ms": round(call_ms, 4), "chamfer_call_ms_min": round(ms[0], 4),
        "shape": [B, n, m], "pairs_per_s": round(pairs / (call_ms * 1e-3), 1),
        "frac_of_fp32_valu_roofline": round(pairs / (call_ms * 1e-3) / PEAK_PAIRS, 4),
        "brute_force_6912_s": round(bf_s, 4), "bf_best_index": int(out[5]), "bf_cd": float(out[6]),
        "bf_pairs_per_s": round(6912 * 2.0 * n * n / bf_s, 1)}))


if __name__ == "__main__":
    main()
