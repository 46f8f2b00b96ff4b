#!/usr/bin/env python3
"""Evaluation launcher with the reference's command line (evaluate.py of ZeroShape):

    python evaluate.py --yaml=options/shape.yaml --eval.vox_res=128 --eval.brute_force --eval.batch_size=1 \\
        --data.dataset_test=synthetic [--load=path/to/shape.ckpt]

Unlike the reference (whose evaluate() asserts a single process, model/shape_engine.py:350) every
visible GPU takes part: a shard of the test set each, per-sample metrics gathered before the result
files are written - or, with --eval.shard_image (automatic when there are more GPUs than test images),
a share of EVERY image: point ranges of its occupancy grid, one RCCL all-gather, and a share of the pose
search's rotations (BASELINE config 5: "SDF grid sharded 8-way with RCCL all-gather").

ZS_VIRTUAL_RANKS=N (with ZS_DEVICE_OVERRIDE=0 ZS_DIST_BACKEND=gloo) rehearses the N-rank path on one GPU.
"""
import importlib
import os
import sys

import torch
import torch.multiprocessing as mp

import zeroshape_amd.compat as compat

compat.install()
import utils.options as options          # noqa: E402
from utils.util import is_port_in_use    # noqa: E402


def main_worker(rank, world_size, port, opt):
    opt.device, opt.world_size, opt.port = rank, world_size, port
    torch.cuda.set_device(int(os.environ.get("ZS_DEVICE_OVERRIDE", rank)))
    engine = importlib.import_module('model.{}_engine'.format(os.path.basename(opt.yaml).split('.')[0]))
    evaluator = engine.Runner(opt)
    evaluator.load_dataset(opt)
    evaluator.test_data.id_filename_mapping(opt, os.path.join(opt.output_path, 'data_list.txt'))
    evaluator.build_networks(opt)
    evaluator.restore_checkpoint(opt, best=True, evaluate=True)
    evaluator.setup_visualizer(opt, test=True)
    evaluator.evaluate(opt, ep=0)


def main():
    print("[{}] (evaluating)".format(sys.argv[0]))
    opt = options.set(opt_cmd=options.parse_arguments(sys.argv[1:]))
    opt.eval.n_vis = 1
    port = (os.getpid() % 32000) + 32768
    while is_port_in_use(port):
        port += 1
    world_size = int(os.environ.get("ZS_VIRTUAL_RANKS", torch.cuda.device_count()))
    if world_size == 1:
        main_worker(0, world_size, port, opt)
    else:
        mp.spawn(main_worker, nprocs=world_size, args=(world_size, port, opt))


if __name__ == "__main__":
    main()
