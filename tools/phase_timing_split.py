#!/usr/bin/env python3
"""Per-phase cycle stamps of the split-fp16 decoder (block 0, wave 0, first tile of a full
129^3 launch).  Build csrc/sdf_decoder_split.hip with -DZS_EXP_TIMING into a side library and
run with ZS_LIB_PATH pointing at it."""
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
st = net.prepare(lat, "f16x3")
for _ in range(2):
    net.query_grid(lat, axis, state=st)
torch.cuda.synchronize()
ws = net.workspace(dev)
tail = ws[-1024:].view(torch.int64).cpu().numpy()[:16]
names = ["init", "point_proj", "b0 LN1", "b0 heads", "b0 LN2", "b0 MLP", "b1 LN1", "b1 heads", "b1 LN2", "b1 MLP",
         "final LN", "impl L0", "impl Z", "impl L1", "impl pairs", "drain"]
kb = {3: 8 * 92, 5: 1024, 7: 8 * 92, 9: 1024, 11: 128, 12: 384, 13: 128, 14: 768}   # K-blocks (x 96 cycles)
for i in range(1, 16):
    d = int(tail[i] - tail[i - 1])
    print("%-12s %9d ticks   MFMA cycles %s" % (names[i], d, kb[i] * 96 if i in kb else ""))
print("total", int(tail[15] - tail[0]), " MFMA", 4928 * 96)
