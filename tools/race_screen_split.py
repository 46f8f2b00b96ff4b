#!/usr/bin/env python3
"""Race screen of the split-fp16 decoder: its weight stream crosses waves through LDS-DMA + raw barriers
(csrc/sdf_decoder_split.hip, AStream), so a misplaced wait would show as rare wrong tiles.  Repeats the
full 2 x 129^3 launch and compares every value bit for bit with the first one.
    python tools/race_screen_split.py        (one MI355X; 40 repeats: 0 mismatches)"""
import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
st = net.prepare(lat, "f16x3")
ref = net.query_grid(lat, axis, apply_sigmoid=False, state=st)
bad = 0
for i in range(40):
    out = net.query_grid(lat, axis, apply_sigmoid=False, state=st)
    bad += int((out != ref).sum())
print("40 repeats of 2 x 129^3 points, mismatching values:", bad)
