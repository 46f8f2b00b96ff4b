#!/usr/bin/env python3
"""Driver for the rocprofv3 passes over the evaluation kernels other than the decoder (profiles/r02_eval_*):
the Chamfer scan nn_both_kernel<2> on [24,10k]x[24,10k], the grid-accelerated exact kernel, one exhaustive
6912-rotation pose search (pose_stats / pose_pack / pose_nn_soa / pose_kill / pose_finish), and the iso-surface
kernels (mc_count, mc_emit, mesh sampling) on a 129^3 level grid of the decoder and on a 257^3 analytic one."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshape_amd import synthetic as syn                                              # noqa: E402
from zeroshape_amd.external.chamfer3D.dist_chamfer_3D import chamfer_3DDist             # noqa: E402
from zeroshape_amd.model.shape.implicit import Implicit                                 # noqa: E402
from zeroshape_amd.utils import eval_3D as E                                            # noqa: E402
from zeroshape_amd.utils.options import EasyDict as edict                               # noqa: E402
from zeroshape_amd.utils.pos_embed import get_2d_sincos_pos_embed                       # noqa: E402

dev = torch.device("cuda:0")
a = torch.from_numpy(syn.seeded_cloud(1, 24, 10000)).to(dev)
b = torch.from_numpy(syn.seeded_cloud(2, 24, 10000)).to(dev)
ch = chamfer_3DDist()
for _ in range(5):
    ch(a, b, "brute")
for _ in range(5):
    ch(a, b, "grid")
pred = torch.from_numpy(syn.ellipsoid_cloud(0, 10000)).to(dev)
gt = torch.from_numpy(syn.seeded_cloud(9, 1, 10000)[0]).to(dev)      # unrelated: no rotation can be dropped early, full scans
E.brute_force_search(pred, gt, device=dev, prune=False, first_batch=192)
pe = get_2d_sincos_pos_embed(256, 14, cls_token=True).astype(np.float32)
sd = {k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(0, pos_embed=pe).items()}
net = Implicit(196, latent_dim=256, n_channels=256, n_blocks_attn=2, n_layers_mlp=8, num_heads=8, skip_in=[2, 4, 6],
               pos_perlayer=False)
net.load_state_dict(sd)
net = net.to(dev).eval()
opt = edict(dict(device="cuda", H=224, W=224, arch=dict(win_size=16), data=dict(dataset_test="synthetic"),
                 eval=dict(vox_res=128, range=[-1.5, 1.5], num_points=10000, icp=False, brute_force=False,
                           f_thresholds=[0.005, 0.01, 0.02, 0.05, 0.1, 0.2])))
lat = torch.from_numpy(syn.seeded_latent(0, 1)).to(dev)
v = edict(dict(idx=[0]))
lv, _ = E.compute_level_grid(opt, net, lat, None, E.get_dense_3D_grid(opt, v), None)
for i in range(3):
    E._surface_clouds(opt, lv, seed=i)
# the iso-surface kernels at BASELINE config 5's size (257^3, an analytic ellipsoid level set: the bench leg's volume)
ax = torch.linspace(-1.5, 1.5, 257, device=dev)
x, y, z = torch.meshgrid(ax, ax, ax, indexing="ij")
vol = torch.sigmoid(-20.0 * (torch.sqrt(x * x + 1.3 * y * y + 0.8 * z * z) - 0.9)).contiguous()
for i in range(3):
    E.extract_surface(vol, 0.5, -1.5, 1.5, num_points=10000, seed=i)
torch.cuda.synchronize()
print("done")
