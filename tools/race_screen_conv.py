#!/usr/bin/env python3
"""Race screen of the convolution engine's LDS pipelines at the encoder's batch-28 layer shapes: every layer is run
`--repeats` times on the same input; all results must be BIT-identical to the first, and the first must agree with the
exact-fp32 engine to the split-fp16 tolerance.  (A fragment read hoisted above a barrier once passed every unit test
and failed only at this scale.)

    python tools/race_screen_conv.py [--batch 28] [--repeats 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshape_amd.nn import ops, pack          # noqa: E402

# Cin, Cout, H, W, k, stride, in_relu, residual
SHAPES = [(128, 32, 224, 224, 3, 1, False, False), (256, 256, 56, 56, 3, 1, True, True), (256, 128, 112, 112, 3, 1, False, False),
          (256, 256, 28, 28, 3, 1, True, False), (64, 64, 56, 56, 3, 1, False, True), (256, 256, 14, 14, 3, 1, False, True),
          (768, 768, 7, 7, 3, 1, False, False), (768, 3072, 1, 197, 1, 1, False, False), (3072, 768, 1, 197, 1, 1, False, True),
          (768, 2304, 1, 197, 1, 1, False, False), (64, 256, 56, 56, 1, 1, False, True), (1024, 256, 14, 14, 1, 1, False, False),
          (256, 1024, 14, 14, 1, 1, False, True), (512, 128, 28, 28, 1, 1, False, False),
          # batch-1 shapes of the streaming kernel (nn_conv_stream.h): 7 x 7 maps, the one-pixel fc head, stride 2, readout
          (768, 768, 14, 14, 3, 2, False, False), (512, 512, 7, 7, 3, 1, False, False), (2048, 512, 7, 7, 1, 1, False, False),
          (512, 2048, 7, 7, 1, 1, False, True), (2048, 2048, 1, 1, 1, 1, False, False), (1536, 768, 1, 196, 1, 1, False, False),
          (1024, 768, 14, 14, 1, 1, False, False), (768, 256, 14, 14, 3, 1, False, False), (1024, 1024, 14, 14, 1, 1, False, True),
          # the 256 x 256 ping-pong kernel (nn_conv_pp256.h) beyond the ViT shapes above: three rounds of tiles + a K-range tail,
          # a long contraction, a ragged last row tile
          (768, 3072, 1, 586, 1, 1, False, True), (3072, 1024, 1, 293, 1, 1, False, False), (2048, 772, 1, 200, 1, 1, False, True)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=28)
    ap.add_argument("--repeats", type=int, default=20)
    a = ap.parse_args()
    bad = 0
    for cin, cout, h, w, k, stride, in_relu, residual in SHAPES:
        g = torch.Generator().manual_seed(cin + cout + h)
        x = torch.randn(a.batch, h, w, cin, generator=g).cuda()
        wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        pc = pack.pack_conv(wt, torch.randn(cout, generator=g), stride=stride, padding=k // 2).to("cuda")
        res = torch.randn(a.batch, h, w, cout, generator=g).cuda() if residual else None
        kw = dict(res1=res, act=ops.ACT_RELU, in_relu=in_relu)
        ops.set_conv_precision("f32")
        exact = ops.conv2d(x, pc, **kw)
        ops.set_conv_precision("f16x3")
        first = ops.conv2d(x, pc, **kw)
        err = ((first - exact).abs().max() / exact.abs().max()).item()
        diff = 0
        for _ in range(a.repeats):
            diff += int((ops.conv2d(x, pc, **kw) != first).sum().item())
        ok = diff == 0 and err < 2e-5
        bad += 0 if ok else 1
        print("Cin %4d Cout %4d %3dx%-3d k%d  vs exact fp32 %.2e of scale   mismatching values over %d repeats: %d   %s"
              % (cin, cout, h, w, k, err, a.repeats, diff, "ok" if ok else "FAIL"), flush=True)
    print("race screen:", "clean" if bad == 0 else "%d shapes FAILED" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
