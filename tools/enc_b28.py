"""Batch-28 encoder as one hipGraph replay (ms per forward); the batch-28 twin of tools/enc_b1.py for threshold sweeps."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_legs import _graph, _events
from zeroshape_amd import synthetic as syn
from zeroshape_amd.utils.options import EasyDict as edict
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 28
opt, g = _graph(dev)
rgb, mask = [torch.from_numpy(x).to(dev) for x in syn.seeded_rgb_scene(0, B)]
var = edict(dict(idx=list(range(B)), rgb_input_map=rgb, mask_input_map=mask))
g.forward(opt, var, training=False, get_loss=False)
g.enable_hip_graph(True)
ms, mn = _events(lambda: g.forward(opt, var, training=False, get_loss=False), 8)
print("batch %d: %.3f ms per forward (min %.3f)" % (B, ms, mn), flush=True)
