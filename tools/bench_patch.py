#!/usr/bin/env python3
"""Input-patch 3x3 kernels (csrc/nn_conv_patch.h) against the implicit-GEMM kernels on the encoder's 3x3 layer shapes at
batch 28: time per call of ops.conv2d with the default dispatch and with the GEMM kernel forced (tiling='large').

    python tools/bench_patch.py [--batch 28] [--iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshape_amd.nn import ops, pack          # noqa: E402

SHAPES = [(128, 32, 224), (256, 256, 56), (256, 128, 112), (256, 256, 28), (64, 64, 56), (128, 128, 28), (256, 256, 14)]   # Cin, Cout, H


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=28)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--shape", action="append", default=[], help="Cin,Cout,H (repeatable; default: the encoder's 3x3 layers)")
    a = ap.parse_args()
    ops.set_conv_precision("f16x3")
    for cin, cout, h in ([tuple(int(v) for v in t.split(",")) for t in a.shape] or SHAPES):
        x = torch.randn(a.batch, h, h, cin, device="cuda")
        w = torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5
        pc = pack.pack_conv(w, None, stride=1, padding=1).to("cuda")
        d = timed(lambda: ops.conv2d(x, pc), a.iters)
        g = timed(lambda: ops.conv2d(x, pc, tiling="large"), a.iters)
        err = (ops.conv2d(x, pc) - ops.conv2d(x, pc, tiling="large")).abs().max().item()
        fl = 2.0 * a.batch * h * h * cout * 9 * cin
        print("Cin %4d Cout %4d %3dx%-3d  default %8.1f us (%6.1f TFLOP/s)   gemm %8.1f us (%6.1f TFLOP/s)   max diff %.2e"
              % (cin, cout, h, h, d, fl / d / 1e6, g, fl / g / 1e6, err), flush=True)


if __name__ == "__main__":
    main()
