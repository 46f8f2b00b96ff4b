"""Every constructor configuration of the reference's `Implicit` with a head dimension of 32 runs on the HIP path (round 6):
the fused kernels serve options/shape.yaml's geometry; the class's own defaults (512 channels, 16 heads, 6 MLP layers,
latent_dim 768), a prediction head + semantic codes + three blocks, other skips + posenc_3D run layer by layer on the training
path's kernels - against goldens of the reference itself (tests/golden/make_variants_golden.py): logits 2e-5 (contract 1e-4),
attention map 3e-7, gradients 1e-4 relative."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from zeroshape_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _variant(name):
    spec = importlib.util.spec_from_file_location("make_variants_golden", os.path.join(os.path.dirname(__file__), "golden",
                                                                                       "make_variants_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.VARIANTS[name], mod.inputs(name)


def _net(v):
    from zeroshape_amd.model.shape.implicit import Implicit
    from zeroshape_amd.utils.pos_embed import get_2d_sincos_pos_embed
    m = Implicit(**v["ctor"])
    shapes = syn.impl_network_shapes(**v["syn"])
    assert list(m.state_dict().keys()) == list(shapes.keys())                      # the reference's keys, in its order
    assert all(tuple(m.state_dict()[k].shape) == tuple(s) for k, s in shapes.items())
    g = int(round(v["syn"]["num_patches"] ** 0.5))
    pe = get_2d_sincos_pos_embed(v["syn"]["n_channels"], g, cls_token=True).astype(np.float32)
    m.load_state_dict({k: torch.from_numpy(a) for k, a in syn.seeded_state_dict(seed=3, pos_embed=pe, **v["syn"]).items()}, strict=True)
    return m.cuda().eval()


@pytest.mark.parametrize("name", ["defaults", "head", "skips"])
def test_other_constructor_configurations_vs_reference_golden(variants_golden, name):
    v, (lat, sem, pts, w) = _variant(name)
    net = _net(v)
    assert not net.fused
    lat_g, pts_g = torch.from_numpy(lat).cuda(), torch.from_numpy(pts).cuda()
    sem_g = torch.from_numpy(sem).cuda() if sem is not None else None
    with torch.no_grad():
        lg, at = net(lat_g, sem_g, pts_g)
        lg2, none = net(lat_g, sem_g, pts_g, need_attn=False)
    assert none is None and torch.equal(lg, lg2)
    np.testing.assert_allclose(lg.cpu().numpy(), variants_golden[name + ".logit"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(at[:, ::64].cpu().numpy(), variants_golden[name + ".attn_rows"], atol=3e-7, rtol=0)
    np.testing.assert_allclose(at.sum(-1).cpu().numpy(), variants_golden[name + ".attn_rowsum"], atol=3e-6, rtol=0)
    with pytest.raises(NotImplementedError):
        net.prepare(lat_g if sem_g is None else torch.cat([lat_g, sem_g], -1))
    # gradients (eval mode, like the golden): weights and latent codes
    lat_r = lat_g.clone().requires_grad_(True)
    for p in net.parameters():
        p.grad = None
    out, _ = net(lat_r, sem_g, pts_g, need_attn=False)
    (out * torch.from_numpy(w).cuda()).sum().backward()
    params = dict(net.named_parameters())
    for k in [k for k in variants_golden if k.startswith(name + ".grad.") and not k.endswith(".latent")]:
        pk = k[len(name) + 6:]
        stride = 16 if "qkv" in pk else (4 if pk.startswith("impl_mlp") else 1)
        want = torch.from_numpy(variants_golden[k]).double()
        got = params[pk].grad.cpu()[::stride].double()
        assert float((got - want).norm()) <= 1e-4 * float(want.norm()), (k, float((got - want).norm()) / float(want.norm()))
    want = torch.from_numpy(variants_golden[name + ".grad.latent"]).double()
    assert float((lat_r.grad.cpu()[:, ::8].double() - want).norm()) <= 1e-4 * float(want.norm())


def test_the_yaml_geometry_stays_on_the_fused_kernels_and_odd_head_sizes_raise(decoder_golden):
    from zeroshape_amd.model.shape.implicit import Implicit
    m = Implicit(syn.NUM_PATCHES, latent_dim=syn.LATENT_DIM, semantic=False, n_channels=syn.N_CHANNELS, n_blocks_attn=syn.ATT_BLOCKS,
                 n_layers_mlp=syn.MLP_LAYERS, num_heads=syn.NUM_HEADS, posenc_3D=0, mlp_ratio=syn.MLP_RATIO,
                 skip_in=list(syn.SKIP_IN), pos_perlayer=False)
    assert m.fused
    odd = Implicit(196, latent_dim=64, n_channels=256, num_heads=4, n_layers_mlp=2).cuda().eval()      # head dimension 64
    assert not odd.fused
    with pytest.raises(NotImplementedError):
        odd(torch.zeros(1, 197, 64).cuda(), None, torch.zeros(1, 8, 3).cuda())
    with pytest.raises(ValueError):
        m.cuda()(torch.zeros(1, 197, 256).cuda(), torch.zeros(1, 197, 8).cuda(), torch.zeros(1, 8, 3).cuda())


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_random_constructor_configurations_vs_oracle(seed):
    """Random constructor configurations (head dimension 32) against the CPU oracle: channel / head counts, 1-3 blocks,
    0-8 MLP layers with random skips, posenc_3D 0-3, mlp_ratio, patch counts, latent widths that are NOT multiples of 4 (the
    padded-operand path), semantic codes, pos_perlayer - logits, attention map and the gradient of the latent codes."""
    from oracle import decoder_ref as R
    from zeroshape_amd.model.shape.implicit import Implicit
    from zeroshape_amd.utils.pos_embed import get_2d_sincos_pos_embed
    rs = np.random.RandomState(100 + seed)
    heads = int(rs.choice([2, 4, 8, 16]))
    C = 32 * heads
    blocks = int(rs.randint(1, 4))
    mlp_layers = int(rs.choice([0, 1, 3, 5, 8]))
    skips = tuple(sorted(int(x) for x in rs.choice(np.arange(1, max(mlp_layers, 2)), size=min(2, max(mlp_layers - 1, 0)), replace=False))) \
        if mlp_layers > 1 else ()
    posenc = int(rs.randint(0, 4)) if mlp_layers else 0
    g = int(rs.choice([5, 7, 14]))
    sem = int(rs.choice([0, 0, 10]))
    latent_dim = int(rs.choice([64, 130, 257])) + sem
    ratio = float(rs.choice([1.0, 2.0, 4.0]))
    per_layer = bool(rs.randint(0, 2))
    cfg = dict(n_channels=C, latent_dim=latent_dim, att_blocks=blocks, mlp_ratio=ratio, mlp_layers=mlp_layers, skip_in=skips,
               num_patches=g * g, posenc_3D=posenc)
    pe = get_2d_sincos_pos_embed(C, g, cls_token=True).astype(np.float32)
    sd = {k: torch.from_numpy(a) for k, a in syn.seeded_state_dict(seed=20 + seed, pos_embed=pe, **cfg).items()}
    net = Implicit(g * g, latent_dim=latent_dim, semantic=sem > 0, n_channels=C, n_blocks_attn=blocks, n_layers_mlp=mlp_layers,
                   num_heads=heads, posenc_3D=posenc, mlp_ratio=ratio, skip_in=list(skips), pos_perlayer=per_layer)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    assert not net.fused
    M = int(rs.choice([1, 33, 200]))
    lat = torch.from_numpy(rs.randn(2, g * g + 1, latent_dim - sem).astype(np.float32))
    semt = torch.from_numpy(rs.randn(2, g * g + 1, sem).astype(np.float32)) if sem else None
    pts = torch.from_numpy(rs.uniform(-1.5, 1.5, size=(2, M, 3)).astype(np.float32))
    want, want_at = R.implicit_forward(sd, lat, pts, num_heads=heads, pos_perlayer=per_layer, latent_semantic=semt)
    with torch.no_grad():
        got, at = net(lat.cuda(), semt.cuda() if sem else None, pts.cuda())
    scale = max(1.0, float(want.abs().max()))
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=3e-5 * scale, rtol=0, err_msg=str(cfg))
    np.testing.assert_allclose(at.cpu().numpy(), want_at.numpy(), atol=5e-7, rtol=0)
    # gradient of the latent codes through the whole network (autograd over HIP kernels vs torch CPU autograd of the oracle)
    w = torch.from_numpy(rs.randn(2, M).astype(np.float32))
    lat_r = lat.clone().cuda().requires_grad_(True)
    out, _ = net(lat_r, semt.cuda() if sem else None, pts.cuda(), need_attn=False)
    (out * w.cuda()).sum().backward()
    leaf = {k: t.clone() for k, t in sd.items()}
    lat_c = lat.clone().requires_grad_(True)
    full = torch.cat([lat_c, semt], -1) if sem else lat_c
    (R.implicit_forward_train(leaf, full, pts, num_heads=heads, pos_perlayer=per_layer) * w).sum().backward()
    wg = lat_c.grad.double()
    assert float((lat_r.grad.cpu().double() - wg).norm()) <= 2e-4 * float(wg.norm()) + 1e-9, cfg
