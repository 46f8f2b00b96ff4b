#!/usr/bin/env python3
"""Training launcher with the reference's command line (train.py of ZeroShape):

    python train.py --yaml=options/shape.yaml [--batch_size=32] [--optim.fix_dpt] [--a.b=v ...]

One process per visible GPU (torch.multiprocessing.spawn, RCCL rendezvous on 127.0.0.1); the engine
is chosen by the yaml's basename (model.shape_engine / model.depth_engine).  The reference's own
train.py runs unchanged too once zeroshape_amd.compat.install() has aliased its package names.
"""
import importlib
import os
import sys

import torch
import torch.multiprocessing as mp

import zeroshape_amd.compat as compat

compat.install()
import utils.options as options          # noqa: E402  (zeroshape_amd.utils.options)
from utils.util import is_port_in_use    # noqa: E402


def main_worker(rank, world_size, port, opt):
    opt.device, opt.world_size, opt.port = rank, world_size, port
    torch.cuda.set_device(rank)
    engine = importlib.import_module('model.{}_engine'.format(os.path.basename(opt.yaml).split('.')[0]))
    trainer = engine.Runner(opt)
    trainer.load_dataset(opt)
    trainer.build_networks(opt)
    trainer.setup_optimizer(opt)
    trainer.restore_checkpoint(opt)
    trainer.setup_visualizer(opt)
    trainer.train(opt)


def main():
    print("[{}] (training)".format(sys.argv[0]))
    opt = options.set(opt_cmd=options.parse_arguments(sys.argv[1:]))
    options.save_options_file(opt)
    port = (os.getpid() % 32000) + 32768
    while is_port_in_use(port):
        port += 1
    world_size = torch.cuda.device_count()
    if world_size == 1:
        main_worker(0, world_size, port, opt)
    else:
        mp.spawn(main_worker, nprocs=world_size, args=(world_size, port, opt))


if __name__ == "__main__":
    main()
