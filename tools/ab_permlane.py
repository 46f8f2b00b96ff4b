"""A/B of one decoder library against another (ZS_LIB_PATH): full 2 x 129^3 grid of the split-fp16 kernel - maximum
difference to the fp32 kernel, a checksum of the raw bits (equal checksums = bit-identical grids) and the launch time.
    python tools/build_variant_lib.py perm sdf_decoder_split.hip zeroshape_amd/csrc/sdf_decoder_split.hip -DZS_SPLIT_PERMLANE
    python tools/ab_permlane.py; ZS_LIB_PATH=tools/_timing/perm.so python tools/ab_permlane.py"""
import sys, os, json, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from zeroshape_amd import synthetic as syn
from zeroshape_amd.model.shape.implicit import Implicit
from zeroshape_amd.utils.pos_embed import get_2d_sincos_pos_embed
dev = torch.device("cuda:0")
pe = get_2d_sincos_pos_embed(256, 14, cls_token=True).astype(np.float32)
sd = {k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(0, pos_embed=pe).items()}
net = Implicit(196, latent_dim=256, n_channels=256, n_blocks_attn=2, n_layers_mlp=8, num_heads=8, skip_in=[2,4,6], pos_perlayer=False)
net.load_state_dict(sd); net = net.to(dev).eval()
lat = torch.from_numpy(syn.seeded_latent(0, 2)).to(dev)
axis = torch.linspace(-1.5, 1.5, 129, device=dev)
st = net.prepare(lat, "f16x3", calibrate=False)
ref32 = net.query_grid(lat, axis, apply_sigmoid=False, state=net.prepare(lat, "f32"))
out = net.query_grid(lat, axis, apply_sigmoid=False, state=st)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
ev[0].record()
for i in range(8):
    o2 = net.query_grid(lat[:1], axis, apply_sigmoid=False, state=net.prepare(lat[:1], "f16x3", calibrate=False)); ev[i + 1].record()
torch.cuda.synchronize()
ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(8))
h = int(torch.sum(out.view(torch.int32).long() * 31 % 1000003))
print(json.dumps({"lib": os.environ.get("ZS_LIB_PATH", "default"), "max_vs_f32": float((out - ref32).abs().max()), "checksum": h, "ms_median": ms[4], "ms_min": ms[0]}))
