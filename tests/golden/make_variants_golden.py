#!/usr/bin/env python3
"""Goldens of the REAL reference's `Implicit` in constructor configurations other than options/shape.yaml's (build container
only; needs /root/reference):

    python tests/golden/make_variants_golden.py

  defaults   the class's own defaults (model/shape/implicit.py:190-194): latent_dim 768, 512 channels, 16 heads, 2 blocks,
             6 MLP layers without skips, pos_perlayer=True, drop_path 0.1 (eval: identity)
  head       a prediction head instead of the per-point MLP (n_layers_mlp=0, :226-229), 3 attention blocks, 128 channels /
             4 heads, mlp_ratio 2, 49 patches, semantic=True (latent_depth 96 + latent_semantic 32 channels, :253)
  skips      256 channels / 8 heads, 5 MLP layers with skips at 1 and 3, posenc_3D 2, mlp_ratio 4, latent_dim 192
Per variant: logits + attention rows of a [2, 512, 3] call, and gradients of a weighted logit sum w.r.t. the first MLP layer
(or the head), the last block's qkv weight and the latent codes (every 4th / 16th / 8th row).  Seeded weights: zeroshape_amd/synthetic.py."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

VARIANTS = {
    "defaults": dict(ctor=dict(num_patches=196), syn=dict(n_channels=512, latent_dim=768, att_blocks=2, mlp_ratio=4.0, mlp_layers=6,
                                                           skip_in=(), num_patches=196), heads=16, sem=0),
    "head": dict(ctor=dict(num_patches=49, latent_dim=128, semantic=True, n_channels=128, n_blocks_attn=3, n_layers_mlp=0, num_heads=4,
                           mlp_ratio=2.0, pos_perlayer=False),
                 syn=dict(n_channels=128, latent_dim=128, att_blocks=3, mlp_ratio=2.0, mlp_layers=0, skip_in=(), num_patches=49),
                 heads=4, sem=32),
    "skips": dict(ctor=dict(num_patches=196, latent_dim=192, n_channels=256, n_blocks_attn=2, n_layers_mlp=5, num_heads=8, posenc_3D=2,
                            skip_in=[1, 3], pos_perlayer=True),
                  syn=dict(n_channels=256, latent_dim=192, att_blocks=2, mlp_ratio=4.0, mlp_layers=5, skip_in=(1, 3), num_patches=196,
                           posenc_3D=2), heads=8, sem=0),
}


def inputs(name):
    v = VARIANTS[name]
    rs = np.random.RandomState({"defaults": 11, "head": 12, "skips": 13}[name])
    L = v["syn"]["num_patches"] + 1
    lat = rs.randn(2, L, v["syn"]["latent_dim"] - v["sem"]).astype(np.float32)
    sem = rs.randn(2, L, v["sem"]).astype(np.float32) if v["sem"] else None
    pts = rs.uniform(-1.5, 1.5, size=(2, 512, 3)).astype(np.float32)
    w = rs.randn(2, 512).astype(np.float32)
    return lat, sem, pts, w


def main():
    import make_golden as mg
    assert os.path.isdir(mg.REF)
    mg._install_stubs()
    sys.path.insert(0, mg.REF)
    from model.shape.implicit import Implicit            # noqa: E402  (reference)
    from zeroshape_amd import synthetic as syn
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = {}
    for name, v in VARIANTS.items():
        net = Implicit(**v["ctor"]).eval()
        ref_sd = net.state_dict()
        shapes = syn.impl_network_shapes(**v["syn"])
        assert list(ref_sd.keys()) == list(shapes.keys()), (name, list(ref_sd.keys())[-4:], list(shapes.keys())[-4:])
        for k in shapes:
            assert tuple(ref_sd[k].shape) == tuple(shapes[k]), (name, k)
        sd_np = syn.seeded_state_dict(seed=3, pos_embed=ref_sd["pos_embed"].numpy().copy(), **v["syn"])
        net.load_state_dict({k: torch.from_numpy(a) for k, a in sd_np.items()}, strict=True)
        lat, sem, pts, w = inputs(name)
        lat_t = torch.from_numpy(lat).requires_grad_(True)
        sem_t = torch.from_numpy(sem) if sem is not None else None
        for p in net.parameters():
            p.grad = None
        lg, at = net(lat_t, sem_t, torch.from_numpy(pts))
        (lg * torch.from_numpy(w)).sum().backward()
        out[name + ".pos_embed_sum"] = np.array([ref_sd["pos_embed"].double().sum().item(), ref_sd["pos_embed"].double().abs().sum().item()])
        out[name + ".logit"] = lg.detach().numpy()
        out[name + ".attn_rows"] = at.detach()[:, ::64].numpy()
        out[name + ".attn_rowsum"] = at.detach().sum(-1).numpy()
        params = dict(net.named_parameters())
        first = "pred_head.weight" if v["syn"]["mlp_layers"] == 0 else "impl_mlp.layers.0.weight"
        last_qkv = "blocks_attn.%d.attn.qkv.weight" % (v["syn"]["att_blocks"] - 1)
        out[name + ".grad." + first] = params[first].grad.numpy()[::4].copy()
        out[name + ".grad." + last_qkv] = params[last_qkv].grad.numpy()[::16].copy()
        out[name + ".grad.latent"] = lat_t.grad.numpy()[:, ::8].copy()
        print("%-9s logits [%.3f, %.3f]  attention row sums [%.4f, %.4f]" % (name, lg.min().item(), lg.max().item(),
                                                                            at.sum(-1).min().item(), at.sum(-1).max().item()))
    np.savez_compressed(os.path.join(HERE, "variants_golden.npz"), **out)
    print("variants_golden.npz: %d arrays, %d bytes" % (len(out), os.path.getsize(os.path.join(HERE, "variants_golden.npz"))))


if __name__ == "__main__":
    main()
