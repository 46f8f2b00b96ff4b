"""Host-side enqueue cost of Implicit.prepare with and without the per-image check (no synchronisation inside the timed part)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshape_amd import synthetic as syn
from zeroshape_amd.model.shape.implicit import Implicit
from zeroshape_amd.utils.pos_embed import get_2d_sincos_pos_embed
dev = torch.device("cuda:0")
pe = get_2d_sincos_pos_embed(256, 14, cls_token=True).astype(np.float32)
sd = {k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(0, pos_embed=pe).items()}
net = Implicit(syn.NUM_PATCHES, latent_dim=256, n_channels=256, n_blocks_attn=2, n_layers_mlp=8, num_heads=8, skip_in=[2, 4, 6], pos_perlayer=False)
net.load_state_dict(sd); net = net.to(dev).eval()
lat = torch.from_numpy(syn.seeded_latent(0, 1)).to(dev)
axis = torch.linspace(-1.5, 1.5, 65, device=dev)
for check in (True, False, True):
    net.image_check = check
    for _ in range(3):
        st = net.prepare(lat); net.query_grid(lat, axis, state=st)
    torch.cuda.synchronize()
    tp = tq = 0.0
    for _ in range(20):
        t0 = time.perf_counter(); st = net.prepare(lat); t1 = time.perf_counter(); net.query_grid(lat, axis, state=st); t2 = time.perf_counter()
        tp += t1 - t0; tq += t2 - t1
        torch.cuda.synchronize()
    print("image_check %d: host prepare %.3f ms, host query_grid %.3f ms" % (check, tp / 20 * 1e3, tq / 20 * 1e3), flush=True)
