#!/usr/bin/env python3
"""Time zs_conv2d_nhwc (and its data / weight gradients) on one layer shape.

    python tools/bench_conv.py --B 4 --H 14 --Cin 256 --Cout 256 --k 3 [--iters 50] [--mode fwd|dgrad|wgrad]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshape_amd.nn import autograd as A          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    for n, d in (("B", 4), ("H", 14), ("Cin", 256), ("Cout", 256), ("k", 3), ("stride", 1), ("iters", 50), ("W", 0)):
        ap.add_argument("--" + n, type=int, default=d)
    ap.add_argument("--mode", default="all")
    ap.add_argument("--engine", default="autograd", help="autograd (training path, nn/autograd.py) | ops (inference path, nn/ops.py)")
    ap.add_argument("--tiling", default=None, help="ops engine: large | small (default: the size-based choice)")
    a = ap.parse_args()
    if a.engine == "ops":
        from zeroshape_amd.nn import ops, pack
        x = torch.randn(a.B, a.H, a.W or a.H, a.Cin, device="cuda")
        w = torch.randn(a.Cout, a.Cin, a.k, a.k) / (a.Cin * a.k * a.k) ** 0.5
        pc = pack.pack_conv(w, None, stride=a.stride, padding=a.k // 2).to("cuda")
        y = ops.conv2d(x, pc, tiling=a.tiling)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ops.conv2d(x, pc, tiling=a.tiling)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        print("M=%d N=%d K=%d: ops.conv2d %.1f us (%.1f TFLOP/s)" % (y.numel() // a.Cout, a.Cout, a.Cin * a.k * a.k, us,
                                                                     2.0 * y.numel() * a.Cin * a.k * a.k / us / 1e6))
        return
    x = torch.randn(a.B, a.H, a.H, a.Cin, device="cuda", requires_grad=True)
    w = (torch.randn(a.Cout, a.Cin, a.k, a.k, device="cuda") / (a.Cin * a.k * a.k) ** 0.5).requires_grad_(True)
    y = A.conv2d(x, w, None, stride=a.stride, padding=a.k // 2)
    gy = torch.randn_like(y)
    flops = 2.0 * y.numel() * a.Cin * a.k * a.k
    ev = lambda: torch.cuda.Event(enable_timing=True)      # noqa: E731

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters * 1e3

    out = {}
    if a.mode in ("all", "fwd"):
        with torch.no_grad():
            out["fwd"] = timed(lambda: A.conv2d(x, w, None, stride=a.stride, padding=a.k // 2))
    if a.mode in ("all", "dgrad"):
        out["dgrad"] = timed(lambda: torch.autograd.grad(y, x, gy, retain_graph=True))
    if a.mode in ("all", "wgrad"):
        out["wgrad"] = timed(lambda: torch.autograd.grad(y, w, gy, retain_graph=True))
    print("M=%d N=%d K=%d:" % (y.numel() // a.Cout, a.Cout, a.Cin * a.k * a.k),
          "  ".join("%s %.1f us (%.1f TFLOP/s)" % (k, v, flops / v / 1e6) for k, v in out.items()))


if __name__ == "__main__":
    main()
