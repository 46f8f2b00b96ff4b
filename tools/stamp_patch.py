#!/usr/bin/env python3
"""s_memtime stamps of one workgroup of conv3x3_patch32_kernel (library built with -DZS_EXP_P32_STAMPS=<block>):
per slab and wave the cycles spent in barrier / issue / taps / land.

    python tools/build_variant_lib.py p32_stamps nn_conv.hip zeroshape_amd/csrc/nn_conv.hip -DZS_EXP_P32_STAMPS=2000
    ZS_LIB_PATH=tools/_timing/p32_stamps.so ZS_CONV_SPLIT_K=1 python tools/stamp_patch.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshape_amd.nn import ops, pack          # noqa: E402


def main():
    ops.set_conv_precision("f16x3")
    assert ops.SPLIT_K, "run with ZS_CONV_SPLIT_K=1 (the stamps go to the split-K workspace)"
    x = torch.randn(28, 224, 224, 128, device="cuda")
    w = torch.randn(32, 128, 3, 3) / (9 * 128) ** 0.5
    pc = pack.pack_conv(w, None, stride=1, padding=1).to("cuda")
    for _ in range(3):
        ops.conv2d(x, pc)
    torch.cuda.synchronize()
    ws = ops.splitk_workspace(x.device)
    raw = ws[(1 << 18):(1 << 18) + 8 * 64 * 2].cpu().numpy().view(np.uint64).reshape(8, 64)
    names = ["barrier", "issue", "taps", "land", "loop"]
    for wv in range(8):
        if raw[wv, 63] == 0:
            continue
        t = raw[wv, :40].astype(np.int64).reshape(8, 5)
        print("wave %d: loop %d ticks, prologue %d, epilogue %d" % (wv, t[-1, -1] - t[0, 0], t[0, 0] - int(raw[wv, 62]), int(raw[wv, 63]) - t[-1, -1]))
        for s in range(8):
            nxt = t[s + 1, 0] if s < 7 else t[s, 4]
            d = [t[s, 1] - t[s, 0], t[s, 2] - t[s, 1], t[s, 3] - t[s, 2], t[s, 4] - t[s, 3], nxt - t[s, 4]]
            print("  slab %d: " % s + "  ".join("%s %6d" % (n, v) for n, v in zip(names, d)))


if __name__ == "__main__":
    main()
