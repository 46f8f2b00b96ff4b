#!/usr/bin/env python3
"""Full-grid goldens of the REAL reference at the BASELINE grid sizes (build container only).

Run:  python tests/golden/make_grid_golden.py          (needs /root/reference; ~2 minutes on 8 threads)
      python tests/golden/make_grid_golden.py 256      (BASELINE config 5's 257^3 grid -> grid256_golden.npz; ~17 minutes)

The reference's own `get_dense_3D_grid` + `compute_level_grid` (utils/eval_3D.py:11-45) are run over the WHOLE
(N+1)^3 grid at vox_res N = 64 and N = 128 (BASELINE.json configs 2 / 3) with the reference's `Implicit`
(model/shape/implicit.py:186-288) on the build-owned seeded weights and latent of zeroshape_amd/synthetic.py - the same
network and image as decoder_golden.npz.  `impl_network` is handed over wrapped in a recorder that forwards the call
untouched and keeps the raw logits of every slice, so one pass yields both what compute_level_grid returns (sigmoid
occupancies) and the logits it was built from.

Stored per N (arrays only, no reference text):
  occ{N}_bits          np.packbits(occ > 0.5) of the returned grid in memory order (x slowest, z fastest): the "voxel
                       indices" BASELINE.json's north_star wants bit-exact - 34 KB / 268 KB
  near{N}_idx / _logit flat index (int32) and raw logit (float32) of EVERY point with |logit| < 1e-3: the only points whose
                       index may legitimately flip under a different (but 1e-4-accurate) arithmetic
  logit{N}_s{S}        raw logits at stride S in every axis (values for a tolerance check away from the surface)
  occ{N}_s{S}          the returned occupancies at the same points
  logit{N}_slice_sum   float64 sum of the logits of every x-slice (a checksum over all points)
  logit{N}_absmax      max |logit| of the grid

The timm `Mlp` / `DropPath` stand-ins and the empty-module stubs are make_golden.py's (same process-local rules).
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

NEAR = 1e-3
STRIDE = {64: 4, 128: 8, 256: 16}


def main():
    import make_golden as mg
    assert os.path.isdir(mg.REF), "reference tree not present: run in the build container"
    mg._install_stubs()
    sys.path.insert(0, mg.REF)
    from model.shape.implicit import Implicit            # noqa: E402  (reference)
    from utils import eval_3D as ref_eval                 # noqa: E402  (reference)
    from utils.util import EasyDict as edict              # noqa: E402  (reference)
    from zeroshape_amd import synthetic as syn            # build-owned inputs

    torch.manual_seed(0)
    torch.set_num_threads(8)
    net = Implicit(syn.NUM_PATCHES, latent_dim=syn.LATENT_DIM, semantic=False,
                   n_channels=syn.N_CHANNELS, n_blocks_attn=syn.ATT_BLOCKS,
                   n_layers_mlp=syn.MLP_LAYERS, num_heads=syn.NUM_HEADS, posenc_3D=0,
                   mlp_ratio=syn.MLP_RATIO, skip_in=list(syn.SKIP_IN), pos_perlayer=False).eval()
    pos_ref = net.state_dict()["pos_embed"].numpy().copy()
    sd_np = syn.seeded_state_dict(seed=0, pos_embed=pos_ref)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    latent = torch.from_numpy(syn.seeded_latent(seed=0, batch=2))[:1]
    opt = edict(dict(device="cpu", H=224, W=224, eval=dict(vox_res=32, range=[-1.5, 1.5]), arch=dict(win_size=16)))
    var = edict(dict(idx=[0]))

    class Recorder(object):
        """impl_network as compute_level_grid calls it (utils/eval_3D.py:41); forwards the call and its two results
        untouched (the attention maps compute_level_grid keeps are 1.7 GB at vox 128) and keeps each slice's logits."""

        def __init__(self):
            self.logits = []

        def __call__(self, latent_depth, latent_semantic, points):
            lg, at = net(latent_depth, latent_semantic, points)
            self.logits.append(lg[0].clone())
            return lg, at

    # `python tests/golden/make_grid_golden.py 256`: BASELINE config 5's grid (257^3 = 17 M points, ~17 minutes, 13 GB of attention
    # maps the reference keeps alive) into its own file, grid256_golden.npz
    # `python tests/golden/make_grid_golden.py 128 gain=60 latent=1`: the same network with its last three MLP layers scaled to a
    # converged checkpoint's logit scale (synthetic.confident_state_dict: |logit| up to ~30) on ANOTHER image -> grid128_gain60_golden.npz
    sizes = tuple(int(a) for a in sys.argv[1:] if a.isdigit()) or (64, 128)
    kw = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
    gain, latent_seed = float(kw.get("gain", 1)), int(kw.get("latent", 0))
    target = "grid_golden.npz" if sizes == (64, 128) and not kw else \
        "grid%s%s_golden.npz" % ("_".join(str(n) for n in sizes), "_gain%g" % gain if gain != 1 else "")
    if gain != 1:
        sd_np = syn.confident_state_dict(sd_np, gain)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    latent = torch.from_numpy(syn.seeded_latent(seed=latent_seed, batch=2))[:1]
    out = {}
    for N in sizes:
        G = N + 1
        rec = Recorder()
        t0 = time.time()
        grid = ref_eval.get_dense_3D_grid(opt, var, N=N)
        with torch.no_grad():
            occ, _ = ref_eval.compute_level_grid(opt, rec, latent, None, grid, None, vis_attn=False)
        occ = occ[0].numpy()
        logit = torch.stack(rec.logits).view(G, G, G).numpy()
        assert occ.shape == (G, G, G) and len(rec.logits) == G
        assert np.array_equal(occ, torch.sigmoid(torch.from_numpy(logit)).numpy()), "recorder and return value disagree"
        flat = logit.reshape(-1)
        near = np.nonzero(np.abs(flat) < NEAR)[0]
        s = STRIDE[N]
        out["occ%d_bits" % N] = np.packbits((occ > 0.5).reshape(-1))
        out["near%d_idx" % N] = near.astype(np.int32)
        out["near%d_logit" % N] = flat[near].copy()
        out["logit%d_s%d" % (N, s)] = logit[::s, ::s, ::s].copy()
        out["occ%d_s%d" % (N, s)] = occ[::s, ::s, ::s].copy()
        out["logit%d_slice_sum" % N] = logit.reshape(G, -1).astype(np.float64).sum(1)
        out["logit%d_absmax" % N] = np.array([np.abs(flat).max()], np.float32)
        print("vox %d: %d points in %.0f s; occ > 0.5: %d; |logit| < %g: %d (min |logit| %.3g); max |logit| %.3f"
              % (N, flat.size, time.time() - t0, int((occ > 0.5).sum()), NEAR, near.size, np.abs(flat).min(),
                 np.abs(flat).max()), flush=True)
    out["near_band"] = np.array([NEAR], np.float64)
    out["gain_and_latent_seed"] = np.array([gain, latent_seed], np.float64)
    np.savez_compressed(os.path.join(HERE, target), **out)
    print("%s: %d arrays, %d bytes" % (target, len(out), os.path.getsize(os.path.join(HERE, target))))


if __name__ == "__main__":
    main()
