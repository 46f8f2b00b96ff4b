"""Unalignable / alignable pose search (6912 rotations, 10k x 10k points) at several rotation batch sizes; run under different
ZS_POSE_STAGES to see what the staged drop costs when nothing can be dropped:  python tools/pose_stages.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshape_amd import synthetic as syn
from zeroshape_amd.utils import eval_3D as E
dev = torch.device("cuda:0")
n = 10000
pred = torch.from_numpy(syn.ellipsoid_cloud(0, n)).to(dev)
R = E._rotation_sphere(dev)
g = torch.Generator(device="cpu").manual_seed(0)
gt = ((R[1234] @ pred.T).T.contiguous().cpu() + 1e-3 * torch.randn(n, 3, generator=g)).to(dev)
far = torch.from_numpy(syn.seeded_cloud(9, 1, n)[0]).to(dev)
for bs in (96, 192, 256):
    out = []
    for name, g_ in (("unalignable", far), ("alignable", gt)):
        best = 1e9
        for _ in range(3):
            E.brute_force_search(pred, g_, device=dev, prune=True, batch_size=bs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            o = E.brute_force_search(pred, g_, device=dev, prune=True, return_index=True, batch_size=bs)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        out.append("%s %.2f ms (index %d)" % (name, best, int(o[5])))
    print("stages %-8s batch %3d: %s" % (os.environ.get("ZS_POSE_STAGES", "1,3,7"), bs, "; ".join(out)), flush=True)
