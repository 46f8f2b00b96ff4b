#!/usr/bin/env python3
"""Is the training step bit-reproducible?  Two runners from the same seed, the same batches, N steps each (eager, or
as the captured hipGraph with ZS_TRAIN_HIP_GRAPH=1); compares every parameter and the loss history bit for bit.
Every reduction of the training kernels has a fixed order (no atomics), so the answer should be yes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_legs import ROOT                      # noqa: E402
from zeroshape_amd.data.synthetic import Dataset       # noqa: E402
from zeroshape_amd.utils import options, util          # noqa: E402
from zeroshape_amd.utils.options import EasyDict as edict   # noqa: E402
from zeroshape_amd.model.shape_engine import Runner    # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def run():
    torch.manual_seed(0)
    cmd = options.parse_arguments(["--yaml=%s/options/shape.yaml" % ROOT, "--output_root=/tmp/zs_repro", "--batch_size=4",
                                   "--pretrain.depth=", "--arch.depth.pretrained=", "--training.n_sdf_points=4096"] +
                                  (["--optim.amp"] if os.environ.get("ZS_TRAIN_AMP") else []))
    opt = options.set(cmd)
    opt.world_size = 1
    opt.output_path = None
    r = Runner(opt)
    r.load_train_dataset(opt, dataset=Dataset(opt, split="train", n_items=8, n_points=100, seed=0))
    r.build_networks(opt)
    r.setup_optimizer(opt)
    r.graph.train()
    batches = list(torch.utils.data.DataLoader(r.train_data, batch_size=4, shuffle=False))
    torch.manual_seed(1)                                # DropPath draws
    losses = []
    for it in range(steps):
        var = util.move_to_device(edict(batches[it % 2]), opt.device)
        losses.append(r.train_iteration(opt, var).all.detach().clone())
    torch.cuda.synchronize()
    return torch.stack(losses).cpu(), {k: v.detach().cpu().clone() for k, v in r.graph.state_dict().items()}


l0, s0 = run()
l1, s1 = run()
same_loss = torch.equal(l0, l1)
diff = [k for k in s0 if not torch.equal(s0[k], s1[k])]
print("steps %d  losses bit-equal: %s  parameters/buffers differing: %d of %d  (first loss %.6f, last %.6f)"
      % (steps, same_loss, len(diff), len(s0), float(l0[0]), float(l0[-1])))
if diff:
    print("e.g.", diff[:5])
