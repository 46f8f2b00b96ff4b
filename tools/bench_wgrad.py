#!/usr/bin/env python3
"""Weight-gradient GEMM per layer shape of the per-GPU-batch-4 training step: fp32-MFMA kernel vs the split-fp16 one.
    python tools/bench_wgrad.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshape_amd.nn import autograd as A      # noqa: E402

SHAPES = [  # B, H, W, Cin, Cout, k, stride, pad
    (4, 1, 197, 768, 3072, 1, 1, 0), (4, 1, 197, 3072, 768, 1, 1, 0), (4, 1, 197, 768, 2304, 1, 1, 0), (4, 1, 197, 768, 768, 1, 1, 0),
    (4, 56, 56, 256, 256, 3, 1, 1), (4, 56, 56, 64, 64, 3, 1, 1), (4, 56, 56, 256, 64, 1, 1, 0), (4, 28, 28, 128, 128, 3, 1, 1),
    (4, 14, 14, 256, 256, 3, 1, 1), (4, 14, 14, 1024, 256, 1, 1, 0), (4, 14, 14, 256, 1024, 1, 1, 0), (4, 112, 112, 256, 128, 3, 1, 1),
    (4, 224, 224, 128, 32, 3, 1, 1), (4, 7, 7, 768, 768, 3, 1, 1), (4, 1, 4096, 256, 1024, 1, 1, 0), (4, 1, 4096, 1024, 256, 1, 1, 0),
]


def main():
    print("%-34s %10s %10s %7s" % ("shape", "f32 us", "f16x3 us", "ratio"))
    tot = {"f32": 0.0, "f16x3": 0.0}
    for B, H, W, Cin, Cout, k, stride, pad in SHAPES:
        x = torch.randn(B, H, W, Cin, device="cuda")
        w = torch.randn(Cout, Cin, k, k, device="cuda", requires_grad=True)
        res = {}
        for prec in ("f32", "f16x3"):
            A.BWD_WGRAD_PRECISION = prec
            y = A.conv2d(x, w, None, stride=stride, padding=pad)
            gy = torch.randn_like(y)

            def run():
                w.grad = None
                y.backward(gy, retain_graph=True)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(10):
                run()
            ev[1].record()
            torch.cuda.synchronize()
            res[prec] = ev[0].elapsed_time(ev[1]) / 10 * 1e3
            tot[prec] += res[prec]
        print("%-34s %10.1f %10.1f %7.2f" % (str((B, H, W, Cin, Cout, k)), res["f32"], res["f16x3"], res["f32"] / res["f16x3"]))
    print("total %.1f vs %.1f us" % (tot["f32"], tot["f16x3"]))
    A.BWD_WGRAD_PRECISION = None


if __name__ == "__main__":
    main()
