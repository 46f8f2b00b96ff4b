"""One warm + one profiled unalignable pose search (6912 rotations, 10k x 10k points) for kernel traces."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshape_amd import synthetic as syn
from zeroshape_amd.utils import eval_3D as E
dev = torch.device("cuda:0")
pred = torch.from_numpy(syn.ellipsoid_cloud(0, 10000)).to(dev)
far = torch.from_numpy(syn.seeded_cloud(9, 1, 10000)[0]).to(dev)
for _ in range(2):
    E.brute_force_search(pred, far, device=dev, prune=True)
torch.cuda.synchronize()
