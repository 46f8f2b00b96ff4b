#!/usr/bin/env python3
"""Golden of the REAL reference with posenc_3D > 0 (build container only; needs /root/reference).

    python tests/golden/make_posenc_golden.py

The reference's `Implicit(posenc_3D=4)` (model/shape/implicit.py:139-166,186-288; get_embedder: utils/layers.py:8-53) on the
build-owned seeded weights (zeroshape_amd/synthetic.py, posenc_3D=4: layers 0 / 2 / 4 / 6 of impl_mlp are 24 columns wider) and
latents: logits + attention rows of a training-shape call, a 9^3 grid through the reference's own compute_level_grid, and
the gradients of a weighted logit sum with respect to the widened layers, the first attention block and the latent (eval
mode - the DropPath stand-in is the identity).  Arrays only; stubs as in make_golden.py."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
L = 4


def main():
    import make_golden as mg
    assert os.path.isdir(mg.REF)
    mg._install_stubs()
    sys.path.insert(0, mg.REF)
    from model.shape.implicit import Implicit            # noqa: E402  (reference)
    from utils import eval_3D as ref_eval                 # noqa: E402  (reference)
    from utils.layers import get_embedder                 # noqa: E402  (reference)
    from utils.util import EasyDict as edict              # noqa: E402  (reference)
    from zeroshape_amd import synthetic as syn

    torch.manual_seed(0)
    torch.set_num_threads(8)
    net = Implicit(syn.NUM_PATCHES, latent_dim=syn.LATENT_DIM, semantic=False, n_channels=syn.N_CHANNELS,
                   n_blocks_attn=syn.ATT_BLOCKS, n_layers_mlp=syn.MLP_LAYERS, num_heads=syn.NUM_HEADS, posenc_3D=L,
                   mlp_ratio=syn.MLP_RATIO, skip_in=list(syn.SKIP_IN), pos_perlayer=False).eval()
    ref_sd = net.state_dict()
    shapes = syn.impl_network_shapes(posenc_3D=L)
    assert list(ref_sd.keys()) == list(shapes.keys())
    for k in shapes:
        assert tuple(ref_sd[k].shape) == tuple(shapes[k]), (k, tuple(ref_sd[k].shape), shapes[k])
    sd_np = syn.seeded_state_dict(seed=0, pos_embed=ref_sd["pos_embed"].numpy().copy(), posenc_3D=L)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    out = {"posenc_3D": np.array([L])}
    # the embedding itself
    embed, dim = get_embedder(L, 3)
    rs = np.random.RandomState(321)
    pts = torch.from_numpy(rs.uniform(-1.5, 1.5, size=(2, 1024, 3)).astype(np.float32))
    assert dim == 3 + 6 * L
    out["embed_rows"] = embed(pts[0, :64]).numpy()
    latent = torch.from_numpy(syn.seeded_latent(seed=0, batch=2))
    with torch.no_grad():
        lg, at = net(latent, None, pts)
    out["logit"] = lg.numpy()
    out["attn_rows"] = at[:, ::128].numpy()
    # a 9^3 grid through the reference's own loop
    opt = edict(dict(device="cpu", H=224, W=224, eval=dict(vox_res=8, range=[-1.5, 1.5]), arch=dict(win_size=16)))
    grid = ref_eval.get_dense_3D_grid(opt, edict(dict(idx=[0])), N=8)
    with torch.no_grad():
        occ, _ = ref_eval.compute_level_grid(opt, net, latent[:1], None, grid, None, vis_attn=False)
    out["occ8"] = occ[0].numpy()
    # gradients (eval mode): loss = sum(w * logits)
    w = torch.from_numpy(np.random.RandomState(5).randn(2, 1024).astype(np.float32))
    lat = latent.clone().requires_grad_(True)
    for p in net.parameters():
        p.grad = None
    lg, _ = net(lat, None, pts)
    (lg * w).sum().backward()
    out["loss_weights"] = w.numpy()
    params = dict(net.named_parameters())
    for k in ["impl_mlp.layers.0.weight", "impl_mlp.layers.2.weight", "impl_mlp.layers.6.weight", "impl_mlp.layers.8.weight",
              "blocks_attn.0.attn.qkv.weight", "point_proj.proj.weight", "latent_proj.bias"]:
        out["grad." + k] = params[k].grad.numpy()[::(16 if "qkv" in k else 4 if k.startswith("impl_mlp") and k.endswith("weight") and params[k].shape[0] > 1 else 1)].copy()
    out["grad.latent"] = lat.grad.numpy()[:, ::8].copy()      # (rows: every 16th of the qkv weight, 4th of the MLP layers, 8th latent token)
    np.savez_compressed(os.path.join(HERE, "posenc_golden.npz"), **out)
    print("posenc_golden.npz: %d arrays, logit range [%.3f, %.3f], %d bytes"
          % (len(out), lg.min().item(), lg.max().item(), os.path.getsize(os.path.join(HERE, "posenc_golden.npz"))))


if __name__ == "__main__":
    main()
