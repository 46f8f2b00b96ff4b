"""Which Python call sites issue device copies during one batch-1 encoder forward (torch profiler, grouped by stack)."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_legs import _graph
from zeroshape_amd import synthetic as syn
from zeroshape_amd.utils.options import EasyDict as edict
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
opt, g = _graph(dev)
rgb, mask = [torch.from_numpy(x).to(dev) for x in syn.seeded_rgb_scene(0, 1)]
var = edict(dict(idx=[0], rgb_input_map=rgb, mask_input_map=mask))
for _ in range(2):
    g.forward(opt, var, training=False, get_loss=False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    g.forward(opt, var, training=False, get_loss=False)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::cat", "aten::zeros", "aten::empty_strided"):
        stack = [s for s in (ev.stack or []) if "zeroshape_amd" in s or "tools/" in s]
        cnt[(ev.name, str(ev.input_shapes)[:60], stack[0] if stack else "?")] += 1
for (name, shp, where), c in cnt.most_common(40):
    print("%4d  %-20s %-62s %s" % (c, name, shp, where[-90:]))
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12)[:3000])
