#!/usr/bin/env python3
"""Decoder launches of ONE rank of an N-GPU step, for rocprofv3 (tools/profile_r06.sh):

    python3 tools/prof_vranks.py <world> [rank] [launches]

batch = world images, `parallel.point_bounds(129^3, world, rank)` of every image in one launch of the split-fp16 kernel
(world = 1: the single-GPU step's launch).  What the FETCH_SIZE pass answers: do `world` different 10 MB programs in one
launch (tiles image-major) cost more HBM-side traffic than one program?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from zeroshape_amd import parallel, synthetic as syn
    from zeroshape_amd.model.shape.implicit import Implicit
    from zeroshape_amd.utils.pos_embed import get_2d_sincos_pos_embed
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    launches = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    dev = torch.device("cuda", 0)
    pe = get_2d_sincos_pos_embed(256, 14, cls_token=True).astype(np.float32)
    sd = {k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(0, pos_embed=pe).items()}
    net = Implicit(syn.NUM_PATCHES, latent_dim=256, n_channels=256, n_blocks_attn=2, n_layers_mlp=8,
                   num_heads=8, skip_in=[2, 4, 6], pos_perlayer=False)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    net.image_check = False                      # the launch alone (the check's probe launches would share the kernel's name)
    G = 129
    latent = torch.from_numpy(syn.seeded_latent(0, world)).to(dev)
    axis = torch.linspace(-1.5, 1.5, G, device=dev)
    st = net.prepare(latent)
    torch.cuda.synchronize()
    b, e, _ = parallel.point_bounds(G ** 3, world, rank)
    for _ in range(launches):
        net.query_grid_range(latent, axis, b, e, apply_sigmoid=True, state=st)
    torch.cuda.synchronize()
    print("world %d rank %d: %d launches of %d x %d points" % (world, rank, launches, world, e - b))


if __name__ == "__main__":
    main()
