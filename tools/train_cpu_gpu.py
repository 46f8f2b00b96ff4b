#!/usr/bin/env python3
"""Is Runner.train_iteration bound by the host (Python + ctypes enqueue) or by the GPU?  Host time to enqueue one
step (no synchronisation) vs the step's wall time."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_legs import ROOT                      # noqa: E402
from zeroshape_amd.data.synthetic import Dataset       # noqa: E402
from zeroshape_amd.utils import options, util          # noqa: E402
from zeroshape_amd.utils.options import EasyDict as edict   # noqa: E402

cmd = options.parse_arguments(["--yaml=%s/options/shape.yaml" % ROOT, "--output_root=/tmp/zs_bench_train", "--batch_size=4",
                               "--pretrain.depth=", "--arch.depth.pretrained=", "--training.n_sdf_points=4096",
                               "--optim.lr=1.e-7", "--optim.lr_ft=1.e-7"] +     # (random weights: keep them where they are)
                              (["--optim.amp"] if os.environ.get("ZS_TRAIN_AMP") else []))
opt = options.set(cmd)
opt.world_size = 1
opt.output_path = None
from zeroshape_amd.model.shape_engine import Runner    # noqa: E402
r = Runner(opt)
r.load_train_dataset(opt, dataset=Dataset(opt, split="train", n_items=4, n_points=100, seed=0))
r.build_networks(opt)
r.setup_optimizer(opt)
r.graph.train()
var0 = util.move_to_device(edict(next(iter(r.train_loader))), opt.device)


def step():
    r.train_iteration(opt, edict({k: (v.clone() if torch.is_tensor(v) else v) for k, v in var0.items()}))


for _ in range(3):
    step()
torch.cuda.synchronize()
host, wall = [], []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0)
    wall.append(t2 - t0)
print("host enqueue %.1f ms (min %.1f), wall %.1f ms (min %.1f) per step" % (1e3 * sum(host) / len(host), 1e3 * min(host),
                                                                             1e3 * sum(wall) / len(wall), 1e3 * min(wall)))
