#!/usr/bin/env python3
"""Shader-clock stamps of the 256 x 256 ping-pong GEMM kernel (build with -DZS_EXP_PP_STAMPS:
python tools/build_variant_lib.py pp_stamps nn_conv.hip zeroshape_amd/csrc/nn_conv.hip -DZS_EXP_PP_STAMPS, then
ZS_LIB_PATH=tools/_timing/pp_stamps.so python tools/pp256_stamps.py): workgroup 0, stages 8-11, every wave."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshape_amd.nn import ops, pack          # noqa: E402

M, K, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (8192, 768, 2048)))
ops.set_conv_precision("f16x3")
x = torch.randn(1, M, 1, K, device="cuda")
pc = pack.pack_conv(torch.randn(N, K, 1, 1) / K ** 0.5, torch.randn(N)).to("cuda")
for _ in range(3):
    ops.conv2d(x, pc, tiling="tile256")
torch.cuda.synchronize()
ws = ops.splitk_workspace(x.device)
off = (1 << 18) + (64 << 20) // 4
st = ws.view(torch.int32)[off:off + 8 * 4 * 8].cpu().view(8, 4, 8).long() & 0xffffffff
t0 = int(st[:, 0, 0].min())
names = {0: ["B", "frags0", "mfma0+dma", "barA", "frags1", "mfma1", "waitvm", "-"],
         1: ["top", "frags0", "barA", "mfma0", "frags1", "waitvm", "barB", "mfma1+dma"]}
for w in range(8):
    print("wave %d (group %d)" % (w, w >> 2))
    for t in range(4):
        row = st[w, t] - t0
        n = 6 if w < 4 else 7
        d = [int(row[k + 1] - row[k]) for k in range(n)]
        print("  stage %2d: start %6d | " % (8 + t, int(row[0])) + "  ".join("%s %d" % (names[w >> 2][k + 1], d[k]) for k in range(n)))
ph = ws.view(torch.int32)[off + 512:off + 512 + 64].cpu().view(2, 8, 4).long() & 0xffffffff
for b, name in ((0, "workgroup 0"), (1, "last workgroup")):
    for w in (0, 4):
        r = ph[b, w]
        print("%s wave %d: entry -> loop end %d cycles (%d stages), epilogue %d" % (name, w, int(r[1] - r[0]), K // 32, int(r[2] - r[1])))
