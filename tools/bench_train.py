#!/usr/bin/env python3
"""Time one training iteration of the shape graph (Runner.train_iteration: Graph.forward(training=True)
-> Loss.shape_loss -> backward -> [gradient all-reduce] -> fused AdamW) on the HIP autograd path.

    python tools/bench_train.py [--batch 4] [--steps 10] [--fix-dpt] [--decoder-only-points 4096]
    python -m torch.distributed.run --nproc-per-node N tools/bench_train.py --batch 4     (N ranks, RCCL)

Prints one JSON line: images/s (whole job), ms per step and its forward / backward / optimiser split.
Synthetic batch of BASELINE.json configs[3]: 224x224 images, 4096 SDF samples per image, 4 images per GPU.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshape_amd.data.synthetic import Dataset                     # noqa: E402
from zeroshape_amd.utils import options, util                        # noqa: E402
from zeroshape_amd.utils.options import EasyDict as edict            # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4, help="images per GPU")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--fix-dpt", action="store_true")
    ap.add_argument("--points", type=int, default=4096)
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl")
    cmd = options.parse_arguments(["--yaml=%s/options/shape.yaml" % ROOT, "--output_root=/tmp/zs_bench_train",
                                   "--batch_size=%d" % (a.batch * world), "--pretrain.depth=", "--arch.depth.pretrained=",
                                   "--training.n_sdf_points=%d" % a.points, "--gpu=%d" % local] +
                                  (["--optim.fix_dpt"] if a.fix_dpt else []))
    opt = options.set(cmd)
    opt.world_size = world
    from zeroshape_amd.model.shape_engine import Runner
    r = Runner(opt)
    r.load_train_dataset(opt, dataset=Dataset(opt, split="train", n_items=a.batch * world, n_points=100, seed=0))
    r.build_networks(opt)
    r.setup_optimizer(opt)
    r.graph.train()
    batch = next(iter(r.train_loader))
    var0 = util.move_to_device(edict(batch), opt.device)

    def clone():
        return edict({k: (v.clone() if torch.is_tensor(v) else v) for k, v in var0.items()})
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    split = [0.0, 0.0, 0.0]

    def step(record):
        var = clone()
        if record:
            ev[0].record()
        var, loss = r.graph.forward(opt, var, training=True, get_loss=True)
        loss = r.summarize_loss(opt, var, loss)
        if record:
            ev[1].record()
        loss.all.backward()
        if r.reducer is not None:
            r.reducer.finish()
        if record:
            ev[2].record()
        r.optim.step()
        r.optim.zero_grad()
        if record:
            ev[3].record()
            torch.cuda.synchronize()
            for i in range(3):
                split[i] += ev[i].elapsed_time(ev[i + 1])
    for _ in range(a.warmup):
        step(False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    if rank == 0:
        n_params = sum(p.numel() for p in r.graph.parameters() if p.requires_grad)
        print(json.dumps(dict(metric="train_images_per_sec", value=a.batch * world * a.steps / dt, n_gpus=world,
                              ms_per_step=dt / a.steps * 1e3, forward_ms=split[0] / a.steps, backward_ms=split[1] / a.steps,
                              optimizer_ms=split[2] / a.steps, batch_per_gpu=a.batch, sdf_points=a.points,
                              fix_dpt=a.fix_dpt, trainable_params=n_params, dtype="f32")))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
