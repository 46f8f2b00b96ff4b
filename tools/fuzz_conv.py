#!/usr/bin/env python3
"""Random-shape sweep of the inference convolution engine: split-fp16 (whatever kernel the dispatcher picks: small
tiles, register-staged, LDS-DMA pipelined; stream-K forced on every other case) against the exact-fp32 kernels of
the same engine and against torch CPU on a subsample.   python tools/fuzz_conv.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshape_amd.nn import ops, pack          # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for case in range(cases):
    k = int(rng.choice([1, 1, 3, 3, 5]))
    stride = int(rng.choice([1, 1, 2]))
    cin = int(rng.choice([4, 16, 20, 32, 48, 64, 96, 128, 256, 320]))
    cout = int(rng.choice([1, 3, 32, 40, 64, 100, 128, 200, 256, 384]))
    B = int(rng.choice([1, 2, 5, 16]))
    H, W = int(rng.randint(3, 70)), int(rng.randint(3, 70))
    if k > min(H, W):
        k = 1
    in_relu = bool(rng.rand() < 0.3)
    act = int(rng.choice([ops.ACT_NONE, ops.ACT_RELU, ops.ACT_GELU]))
    tiling = [None, "large", "small"][int(rng.randint(3))]
    g = torch.Generator().manual_seed(case)
    x = torch.randn(B, H, W, cin, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g)
    pc = pack.pack_conv(w, b, stride=stride, padding=k // 2).to("cuda")
    xg = x.cuda()
    ops.STREAM_K = "always" if case % 2 else False
    outs = {}
    res = None
    for prec in ("f32", "f16x3"):
        ops.set_conv_precision(prec)
        if res is None:
            res = torch.randn(ops.conv2d(xg, pc, tiling=tiling).shape, generator=g).cuda()
        outs[prec] = ops.conv2d(xg, pc, res1=res, act=act, in_relu=in_relu, tiling=tiling).cpu()
    scale = float(outs["f32"].abs().max()) + 1e-6
    err = float((outs["f32"] - outs["f16x3"]).abs().max()) / scale
    worst = max(worst, err)
    ok = err < 2e-5 and bool(torch.isfinite(outs["f16x3"]).all())
    if case % 10 == 0:                       # torch CPU on every tenth case
        xr = F.relu(x) if in_relu else x
        ref = F.conv2d(xr.permute(0, 3, 1, 2), w, b, stride=stride, padding=k // 2).permute(0, 2, 3, 1) + res.cpu()
        ref = {ops.ACT_NONE: ref, ops.ACT_RELU: F.relu(ref), ops.ACT_GELU: F.gelu(ref)}[act]
        e2 = float((ref - outs["f32"]).abs().max()) / scale
        ok = ok and e2 < 2e-5
    if not ok:
        print("FAIL case %d: B=%d %dx%d cin=%d cout=%d k=%d s=%d in_relu=%s act=%d tiling=%s stream_k=%s err=%.3e"
              % (case, B, H, W, cin, cout, k, stride, in_relu, act, tiling, ops.STREAM_K, err))
        sys.exit(1)
print("%d cases ok, worst |f16x3 - f32| / max = %.2e" % (cases, worst))
