"""BASELINE config 2 once warm + N timed: image -> latent (captured encoder) -> 65^3 grid; prints wall ms per image and is the
target of tools/prof_by_grid.sh for the kernel-side breakdown:  python tools/inference_once.py [image_check 0|1] [vox]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_legs import _graph
from zeroshape_amd import synthetic as syn
from zeroshape_amd.utils import eval_3D as E
from zeroshape_amd.utils.options import EasyDict as edict
dev = torch.device("cuda:0")
check = (sys.argv[1] if len(sys.argv) > 1 else "1") != "0"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
opt, g = _graph(dev)
g.enable_hip_graph(True)
g.impl_network.image_check = check
rgb, mask = [torch.from_numpy(x).to(dev) for x in syn.seeded_rgb_scene(0, 1)]
var = edict(dict(idx=[0], rgb_input_map=rgb, mask_input_map=mask))
o = edict(dict(opt, eval=dict(vox_res=N, range=[-1.5, 1.5])))
o.device = str(dev)


def run():
    v = g.forward(opt, var, training=False, get_loss=False)
    v = v[0] if isinstance(v, tuple) else v
    pts = E.get_dense_3D_grid(o, v, N)
    return E.compute_level_grid(o, g.impl_network, v.latent_depth, None, pts, None)[0]


for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    run()
torch.cuda.synchronize()
print("image_check %d vox %d: %.3f ms per image (wall, 10 back-to-back)" % (check, N, (time.perf_counter() - t0) * 100))
