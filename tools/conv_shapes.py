#!/usr/bin/env python3
"""Per-layer-shape time of the convolution engine inside one encoder forward (events around every ops.conv2d call).
    python tools/conv_shapes.py [batch]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_encoder import make_opt                                     # noqa: E402
from zeroshape_amd import synthetic as syn                                   # noqa: E402
from zeroshape_amd.model.compute_graph.graph_shape import Graph              # noqa: E402
from zeroshape_amd.nn import ops                                             # noqa: E402
from zeroshape_amd.utils.options import EasyDict as edict                    # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 28
opt = make_opt("resnet")
torch.manual_seed(0)
g = Graph(opt).cuda().eval()
rgb, mask = [torch.from_numpy(x).cuda() for x in syn.seeded_rgb_scene(0, B)]
var = edict(dict(idx=list(range(B)), rgb_input_map=rgb, mask_input_map=mask))
g.forward(opt, var, training=False, get_loss=False)
rec = collections.OrderedDict()
orig = ops.conv2d


def timed(x, pc, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = orig(x, pc, *a, **k)
    e1.record()
    torch.cuda.synchronize()
    M = (y[0] if isinstance(y, tuple) else y).numel() // pc.cout
    fz = ("gn" if k.get("gn_in") is not None else "ln" if k.get("ln_in") is not None else "-") + ">" + \
        ({"group": "gn", "row": "ln"}.get(k.get("stats_out"), "-"))
    key = (M, pc.cout, pc.cin * pc.kh * pc.kw, pc.kh, pc.stride, fz)
    r = rec.setdefault(key, [0, 0.0])
    r[0] += 1
    r[1] += e0.elapsed_time(e1)
    return y


ops.conv2d = timed
for mod in list(sys.modules.values()):
    if mod is not None and getattr(mod, "__name__", "").startswith("zeroshape_amd") and getattr(mod, "conv2d", None) is orig:
        mod.conv2d = timed
N = 3
for _ in range(N):
    g.forward(opt, var, training=False, get_loss=False)
tot = sum(v[1] for v in rec.values()) / N
print("B=%d: %d conv calls per forward, %.2f ms" % (B, sum(v[0] for v in rec.values()) // N, tot))
print("(fused: normalisation applied on load > statistics written by the epilogue; times include ~3-5 us of event markers)")
print("%8s %6s %6s %2s %2s %6s %5s %9s %8s %7s %9s" % ("M", "N", "K", "k", "s", "fused", "calls", "us/call", "ms", "TFLOP/s", "weight MB"))
for (M, Nn, K, kh, st, fz), (c, ms) in sorted(rec.items(), key=lambda kv: -kv[1][1]):
    us = ms / c * 1e3
    print("%8d %6d %6d %2d %2d %6s %5d %9.1f %8.3f %7.1f %9.2f" % (M, Nn, K, kh, st, fz, c // N, us, ms / N, 2.0 * M * Nn * K / us / 1e6,
                                                               Nn * K * 4 / 1e6))
