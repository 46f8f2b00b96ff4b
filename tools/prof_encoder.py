#!/usr/bin/env python3
"""Encoder half of Graph.forward at one batch size, eager launches (so rocprofv3 sees every kernel by name).
    rocprofv3 --kernel-trace --stats -d /tmp/p -- python3 tools/prof_encoder.py 28 5 [resnet|att]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_encoder import make_opt                                     # noqa: E402
from zeroshape_amd import synthetic as syn                                   # noqa: E402
from zeroshape_amd.model.compute_graph.graph_shape import Graph              # noqa: E402
from zeroshape_amd.utils.options import EasyDict as edict                    # noqa: E402

B, iters = int(sys.argv[1]), int(sys.argv[2])
opt = make_opt(sys.argv[3] if len(sys.argv) > 3 else "resnet")
torch.manual_seed(0)
g = Graph(opt).cuda().eval()
rgb, mask = [torch.from_numpy(x).cuda() for x in syn.seeded_rgb_scene(0, B)]
var = edict(dict(idx=list(range(B)), rgb_input_map=rgb, mask_input_map=mask))
for _ in range(iters + 1):
    g.forward(opt, var, training=False, get_loss=False)
torch.cuda.synchronize()
