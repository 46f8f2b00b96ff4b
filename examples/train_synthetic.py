#!/usr/bin/env python3
"""End-to-end training demonstration on the analytic data set (posed ellipsoids rendered in closed
form, zeroshape_amd/data/synthetic.py): the reference's recipe (options/shape.yaml: shape loss, AdamW
with the four parameter groups, DropPath, BatchNorm on batch statistics) through the Runner, every
forward / backward / optimiser op on the HIP library.  Prints the loss curve, the training throughput
and the Chamfer distance of the reconstructions before and after.

    python examples/train_synthetic.py [--items 32] [--epochs 12] [--batch 4] [--lr 3e-4]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshape_amd.data.synthetic import Dataset                     # noqa: E402
from zeroshape_amd.utils import options, util                        # noqa: E402
from zeroshape_amd.utils.options import EasyDict as edict            # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=32)
    ap.add_argument("--epochs", type=int, default=12)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--vox", type=int, default=32)
    ap.add_argument("--out", default="/tmp/zs_train_demo")
    ap.add_argument("--amp", action="store_true", help="optim.amp: split-fp16 forward / data-gradient GEMMs under the loss scaler")
    ap.add_argument("--hip-graph", action="store_true", help="optim.hip_graph: the step as one captured hipGraph")
    a = ap.parse_args()
    cmd = options.parse_arguments(["--yaml=%s/options/shape.yaml" % ROOT, "--output_root=%s" % a.out,
                                   "--batch_size=%d" % a.batch, "--max_epoch=%d" % a.epochs, "--pretrain.depth=",
                                   "--arch.depth.pretrained=", "--eval.vox_res=%d" % a.vox, "--eval.num_points=2000",
                                   "--eval.batch_size=4", "--training.n_sdf_points=2048", "--optim.lr=%g" % a.lr,
                                   "--optim.lr_ft=%g" % (a.lr / 3), "--freq.eval=1000"] +
                                  (["--optim.amp"] if a.amp else []) + (["--optim.hip_graph"] if a.hip_graph else []))
    opt = options.set(cmd)
    opt.world_size = 1
    from zeroshape_amd.model.shape_engine import Runner
    torch.manual_seed(0)
    r = Runner(opt)
    train = Dataset(opt, split="train", n_items=a.items, n_points=4000)
    r.load_dataset(opt, dataset=Dataset(opt, split="train", n_items=min(8, a.items), n_points=4000), train_dataset=train)
    r.build_networks(opt)
    r.setup_optimizer(opt)
    r.restore_checkpoint(opt)
    before = r.evaluate(opt, training=True)
    r.graph.train()
    curve = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for ep in range(a.epochs):
        losses = []
        for batch in r.train_loader:
            var = util.move_to_device(edict(batch), opt.device)
            losses.append(r.train_iteration(opt, var).all.detach().clone())     # (a captured step reuses its loss tensor)
        curve.append(float(torch.stack(losses).mean()))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    after = r.evaluate(opt, training=True)
    print(json.dumps(dict(items=a.items, epochs=a.epochs, iterations=r.it, batch=a.batch, seconds=round(dt, 2),
                          images_per_sec=round(r.it * a.batch / dt, 1), loss_first_epoch=round(curve[0], 4),
                          loss_last_epoch=round(curve[-1], 4), loss_curve=[round(c, 4) for c in curve],
                          chamfer_before=round(before["cd"], 4), chamfer_after=round(after["cd"], 4),
                          fscore_005_before=round(before["f_scores"][3], 4), fscore_005_after=round(after["f_scores"][3], 4),
                          non_finite_parameters=sum(int(not torch.isfinite(p).all()) for p in r.graph.parameters()),
                          loss_scale=(float(r.scaler.scale) if hasattr(r, "scaler") else None),
                          captured=getattr(r, "_captured", None) is not None)))


if __name__ == "__main__":
    main()
