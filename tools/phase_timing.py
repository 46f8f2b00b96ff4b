#!/usr/bin/env python3
"""Per-phase cycle stamps of the fused decoder (one wave, first tile).

Build the instrumented library and run on the GPU box:

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -DZS_EXP_TIMING \
          -c zeroshape_amd/csrc/sdf_decoder.hip -o /tmp/sdf_T.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o exp/lib_TIMING.so /tmp/sdf_T.o \
          zeroshape_amd/csrc/_obj/{chamfer,common,sdf_prologue}.o
    ZS_LIB_PATH=$PWD/exp/lib_TIMING.so python tools/phase_timing.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from zeroshape_amd import synthetic as syn
from zeroshape_amd.model.shape.implicit import Implicit
from zeroshape_amd.utils.pos_embed import get_2d_sincos_pos_embed
dev = torch.device("cuda:0")
pe = get_2d_sincos_pos_embed(256, 14, cls_token=True).astype(np.float32)
sd = {k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(0, pos_embed=pe).items()}
net = Implicit(196, latent_dim=256, n_channels=256, n_blocks_attn=2, n_layers_mlp=8, num_heads=8, skip_in=[2,4,6], pos_perlayer=False)
net.load_state_dict(sd); net = net.to(dev).eval()
lat = torch.from_numpy(syn.seeded_latent(0, 1)).to(dev)
axis = torch.linspace(-1.5, 1.5, 129, device=dev)
st = net.prepare(lat)
for _ in range(2):
    net.query_grid(lat, axis, state=st)
torch.cuda.synchronize()
ws = net.workspace(dev)
tail = ws[-1024:].view(torch.int64).cpu().numpy()[:16]
names = ["start","b0 LN1 in","b0 LN1 out","b0 heads done","b0 LN2 out","b1 LN1 in","b1 LN1 out","b1 heads done","b1 LN2 out","b1 MLP done","paramsB","L0 done","Z done","L1 done","loop done","end"]
ideal = {3: 8*736*64, 5: 32*256*64, 7: 8*736*64, 9: 32*256*64, 11: 1024*64, 12: 3072*64, 13: 1024*64, 14: 6144*64}
prev = tail[0]
for i in range(1, 16):
    d = int(tail[i] - tail[i-1])
    print("%-14s %9d ticks  ideal MFMA cycles %s" % (names[i], d, ideal.get(i, "")))
print("total", int(tail[15]-tail[0]))
