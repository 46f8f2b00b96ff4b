#!/usr/bin/env python3
"""The captured training step with a GradReducer on ONE GPU (RCCL group of one rank, GradReducer(always=True)): ms per step as
one hipGraph with every bucket behind it (optim.hip_graph_segments=false) and as one graph per backward segment with the
buckets issued between the replays (the default) - what the split itself costs where there is nothing to overlap with.
Under `tools/prof_train_timeline.sh` the kernel trace of the segmented form shows where the bucket launches
(`copy_multi_kernel` = pack, RCCL kernels when the group has more than one rank) fall inside the backward window.

    python3 tools/train_segments.py [segments|single|eager|both] [steps]"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_legs import ROOT                      # noqa: E402
from zeroshape_amd import parallel                     # noqa: E402
from zeroshape_amd.data.synthetic import Dataset       # noqa: E402
from zeroshape_amd.utils import options, util          # noqa: E402
from zeroshape_amd.utils.options import EasyDict as edict   # noqa: E402


def run(segments, steps, eager=False):
    """eager=True: the DEFAULT data-parallel step (options/shape.yaml: hip_graph false) - eager launches, the buckets packed and
    all-reduced by the gradient hooks under the backward pass (GradReducer.finish)."""
    from zeroshape_amd.model.shape_engine import Runner
    cmd = options.parse_arguments(["--yaml=%s/options/shape.yaml" % ROOT, "--output_root=/tmp/zs_bench_train", "--batch_size=4",
                                   "--pretrain.depth=", "--arch.depth.pretrained=", "--training.n_sdf_points=4096",
                                   "--optim.lr=1.e-7", "--optim.lr_ft=1.e-7"] +
                                  ([] if eager else ["--optim.hip_graph", "--optim.hip_graph_segments=%s" % ("true" if segments else "false")]) +
                                  (["--optim.amp"] if os.environ.get("ZS_TRAIN_AMP") else []))
    opt = options.set(cmd)
    opt.world_size = 1
    opt.output_path = None
    r = Runner(opt)
    r.load_train_dataset(opt, dataset=Dataset(opt, split="train", n_items=4, n_points=100, seed=0))
    r.build_networks(opt)
    r.setup_optimizer(opt)
    r.reducer = parallel.GradReducer(r.graph.parameters(), module=r.graph, always=True)
    r.graph.train()
    var0 = util.move_to_device(edict(next(iter(r.train_loader))), opt.device)
    issued = []
    inner = r.reducer.launch_done
    r.reducer.launch_done = lambda params: issued.append(inner(params)) or issued[-1]

    def step():
        r.train_iteration(opt, edict({k: (v.clone() if torch.is_tensor(v) else v) for k, v in var0.items()}))
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    cap = getattr(r, "_captured", None) or {"graphs": [], "seg_params": None}
    out = {"segments": bool(segments), "eager": bool(eager), "ms_per_step": round(ms, 3), "graphs": len(cap["graphs"]),
           "buckets": len(r.reducer.buckets), "bucket_mb": [round(f.numel() * 4 / 2 ** 20, 1) for f in r.reducer.flat]}
    if cap["seg_params"] is not None:
        out["parameters_per_segment"] = [len(g) for g in cap["seg_params"]]
        out["gradient_mb_per_segment"] = [round(sum(p.numel() for p in g) * 4 / 2 ** 20, 1) for g in cap["seg_params"]]
        out["buckets_issued_after_each_replay"] = issued[-len(cap["graphs"]):]
    r.reducer.close()
    return out


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "both"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        res = [run(m == "segments", steps, eager=m == "eager")
               for m in (("single", "segments", "eager") if mode == "both" else (mode,))]
    finally:
        dist.destroy_process_group()
    print(json.dumps(res), flush=True)
    if os.environ.get("ZS_TRAIN_SEGMENTS_OUT"):
        with open(os.environ["ZS_TRAIN_SEGMENTS_OUT"], "w") as f:
            json.dump(res, f)
