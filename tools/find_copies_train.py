"""Which Python call sites issue device copies / fills during one training iteration (torch profiler, grouped by stack).
ZS_TRAIN_AMP=1: optim.amp."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_legs import ROOT                      # noqa: E402
from zeroshape_amd.data.synthetic import Dataset       # noqa: E402
from zeroshape_amd.utils import options, util          # noqa: E402
from zeroshape_amd.utils.options import EasyDict as edict   # noqa: E402
from torch.profiler import profile, ProfilerActivity   # noqa: E402

cmd = options.parse_arguments(["--yaml=%s/options/shape.yaml" % ROOT, "--output_root=/tmp/zs_bench_train", "--batch_size=4",
                               "--pretrain.depth=", "--arch.depth.pretrained=", "--training.n_sdf_points=4096",
                               "--optim.lr=1.e-7", "--optim.lr_ft=1.e-7"] +
                              (["--optim.amp"] if os.environ.get("ZS_TRAIN_AMP") else []))
opt = options.set(cmd)
opt.world_size = 1
opt.output_path = None
from zeroshape_amd.model.shape_engine import Runner    # noqa: E402
r = Runner(opt)
r.load_train_dataset(opt, dataset=Dataset(opt, split="train", n_items=4, n_points=100, seed=0))
r.build_networks(opt)
r.setup_optimizer(opt)
r.graph.train()
var0 = util.move_to_device(edict(next(iter(r.train_loader))), opt.device)


def step():
    r.train_iteration(opt, edict({k: (v.clone() if torch.is_tensor(v) else v) for k, v in var0.items()}))


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
names = ("aten::copy_", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::cat",
         "aten::zeros", "aten::add_", "aten::mul", "aten::add", "aten::sum", "aten::mul_", "aten::div", "aten::stack")
for ev in prof.events():
    if ev.name in names:
        stack = [s for s in (ev.stack or []) if "zeroshape_amd" in s or "tools/" in s]
        par = ev.cpu_parent
        where = stack[0][-80:] if stack else "(autograd engine) under " + (par.name if par is not None else "-")
        cnt[(ev.name, str(ev.input_shapes)[:70], where)] += 1
for (name, shp, where), c in cnt.most_common(70):
    print("%4d  %-16s %-72s %s" % (c, name, shp, where))
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25)[:6000])
