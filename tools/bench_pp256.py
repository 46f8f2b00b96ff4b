#!/usr/bin/env python3
"""A/B of the 256 x 256 ping-pong GEMM kernel (csrc/nn_conv_pp256.h) against the 128 x 128 LDS-DMA kernel on the large
pointwise shapes of the batch-28 encoder, interleaved rounds in ONE process (tiling="large" forces the old kernel).

    python tools/bench_pp256.py [--rounds 5] [--iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshape_amd.nn import ops, pack          # noqa: E402

SHAPES = [  # (name, M, K, N)
    ("fc1", 5516, 768, 3072), ("fc2", 5516, 3072, 768), ("qkv", 5516, 768, 2304), ("proj", 5516, 768, 768),
    ("big3r", 16384, 768, 3072), ("full1r", 8192, 768, 2048), ("r1024", 5488, 1024, 1024), ("r256", 5488, 1024, 256),
    ("c56a", 87808, 64, 256), ("c56b", 87808, 256, 64), ("c28", 21952, 128, 512), ("c28b", 21952, 512, 128),
    ("c14", 5488, 256, 1024),
    # window stage of the transformer coordinate encoder at batch 28 (5,488 windows x 65 tokens) and the 56 x 56 bottleneck layers
    ("w_qkv", 356720, 256, 768), ("w_proj", 356720, 256, 256), ("w_fc1", 356720, 256, 1024), ("w_fc2", 356720, 1024, 256),
    ("c56c", 87808, 256, 256), ("c56d", 87808, 512, 128), ("c28c", 21952, 1024, 256), ("c28d", 21952, 256, 1024),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    ops.set_conv_precision("f16x3")
    ev = lambda: torch.cuda.Event(enable_timing=True)      # noqa: E731
    print("%-8s %6s %5s %5s | %9s %9s | %7s %7s | tiles" % ("shape", "M", "K", "N", "old us", "new us", "old TF", "new TF"))
    for name, M, K, N in SHAPES:
        if a.only and name not in a.only.split(","):
            continue
        x = torch.randn(1, M, 1, K, device="cuda")
        w = torch.randn(N, K, 1, 1) / K ** 0.5
        pc = pack.pack_conv(w, torch.randn(N)).to("cuda")
        t = {"large": [], "tile256": []}
        for _ in range(a.rounds):
            for tiling in ("large", "tile256"):
                ops.conv2d(x, pc, tiling=tiling)
                torch.cuda.synchronize()
                e0, e1 = ev(), ev()
                e0.record()
                for _ in range(a.iters):
                    ops.conv2d(x, pc, tiling=tiling)
                e1.record()
                torch.cuda.synchronize()
                t[tiling].append(e0.elapsed_time(e1) / a.iters * 1e3)
        old, new = sorted(t["large"])[len(t["large"]) // 2], sorted(t["tile256"])[len(t["tile256"]) // 2]
        fl = 2.0 * M * K * N / 1e6
        tiles = -(-M // 256) * -(-N // 256)
        print("%-8s %6d %5d %5d | %9.1f %9.1f | %7.1f %7.1f | %d" % (name, M, K, N, old, new, fl / old, fl / new, tiles), flush=True)


if __name__ == "__main__":
    main()
