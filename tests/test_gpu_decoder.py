"""GPU parity: fused HIP decoder (prologue + per-point kernel, through the C ABI) vs the
oracle and vs the golden outputs of the real reference.

Bars (BASELINE.json north_star): logits / occupancy within 1e-4 absolute (we assert a much
tighter 2e-5, the kernel is exact-fp32 MFMA); occupancy indices (occ > 0.5) bit-exact
outside an ambiguity band |logit| < 1e-5 around the level set, flips inside it counted
and bounded."""
import numpy as np
import pytest
import torch

from oracle import decoder_ref as R
from zeroshape_amd import program as P
from zeroshape_amd import synthetic as syn

pytestmark = pytest.mark.gpu

ATOL = 2e-5     # fp32 roundoff over ~25 dependent stages; the contract is 1e-4
BAND = 1e-5     # |logit| below which an occupancy flip is a rounding tie, not a bug


@pytest.fixture(scope="module")
def net(seeded_sd):
    from zeroshape_amd.model.shape.implicit import Implicit
    m = Implicit(syn.NUM_PATCHES, latent_dim=syn.LATENT_DIM, semantic=False, n_channels=syn.N_CHANNELS,
                 n_blocks_attn=syn.ATT_BLOCKS, n_layers_mlp=syn.MLP_LAYERS, num_heads=syn.NUM_HEADS,
                 posenc_3D=0, mlp_ratio=syn.MLP_RATIO, skip_in=list(syn.SKIP_IN), pos_perlayer=False)
    m.load_state_dict(seeded_sd, strict=True)
    m = m.cuda().eval()
    m.precision = "f32"          # this file pins the exact-fp32 kernels; test_gpu_decoder_split.py the default
    return m


def test_state_dict_contract(net, seeded_sd):
    assert list(net.state_dict().keys()) == list(syn.impl_network_shapes().keys())


def test_prologue_kv_records_match_oracle(net, seeded_sd):
    latent = torch.from_numpy(syn.seeded_latent(seed=0, batch=2))
    st = net.prepare(latent.cuda())
    got = st.programs.cpu().numpy()
    lp = R.latent_path(seeded_sd, latent)
    sd_np = {k: v.numpy() for k, v in seeded_sd.items()}
    for b in range(2):
        kv = {(blk, h): (lp["k%d" % blk][b, h].numpy(), lp["v%d" % blk][b, h].numpy())
              for blk in range(2) for h in range(8)}
        want = P.pack_program(sd_np, kv)
        # K/V records are computed on the device (fp32, different summation order than the
        # oracle's GEMMs); everything else is the packed weights, copied verbatim
        np.testing.assert_allclose(got[b], want, atol=3e-6, rtol=0)
        is_kv = np.zeros(want.size, bool)
        for blk in range(2):
            for h in range(8):
                o = P.kv_group_offset(blk, h) * P.GROUP_FLOATS
                is_kv[o:o + P.G_KV_HEAD * P.GROUP_FLOATS] = True
        np.testing.assert_array_equal(got[b][~is_kv], want[~is_kv])
        assert np.abs(want[is_kv]).max() > 0.1      # the K/V section is really populated


def test_training_shape_points_vs_golden(net, decoder_golden):
    latent = torch.from_numpy(syn.seeded_latent(seed=0, batch=2)).cuda()
    rs = np.random.RandomState(123)
    pts = torch.from_numpy(rs.uniform(-1, 1, size=(2, 4096, 3)).astype(np.float32)).cuda()
    lg, attn = net(latent, None, pts, need_attn=False)
    assert attn is None and lg.shape == (2, 4096) and lg.dtype == torch.float32
    np.testing.assert_allclose(lg.cpu().numpy(), decoder_golden["pts4096_logit"], atol=ATOL, rtol=0)
    # default call = the reference's contract: (logits, attn [B,M,197])
    lg2, attn = net(latent, None, pts)
    assert torch.equal(lg2, lg), "the attention variant must not change the logits"
    assert attn.shape == (2, 4096, 197)
    np.testing.assert_allclose(attn[:, ::512].cpu().numpy(), decoder_golden["pts4096_attn_rows"], atol=2e-7, rtol=0)
    np.testing.assert_allclose(attn.sum(-1).cpu().numpy(), decoder_golden["pts4096_attn_rowsum"], atol=2e-6, rtol=0)


def test_attention_map_vs_oracle_ragged(net, seeded_sd):
    latent = torch.from_numpy(syn.seeded_latent(seed=5, batch=2))
    pts = torch.from_numpy(syn.seeded_cloud(77, 2, 333, -1.5, 1.5))
    want_l, want_a = R.implicit_forward(seeded_sd, latent, pts)
    got_l, got_a = net(latent.cuda(), None, pts.cuda())
    np.testing.assert_allclose(got_l.cpu().numpy(), want_l.numpy(), atol=ATOL, rtol=0)
    np.testing.assert_allclose(got_a.cpu().numpy(), want_a.numpy(), atol=2e-7, rtol=0)
    assert bool(torch.all(got_a.sum(-1) < 1.0))     # self column excluded after the softmax


@pytest.mark.parametrize("m", [1, 31, 32, 33, 127, 128, 129, 1000])
def test_ragged_point_counts_vs_oracle(net, seeded_sd, m):
    latent = torch.from_numpy(syn.seeded_latent(seed=3, batch=1))
    pts = torch.from_numpy(syn.seeded_cloud(m, 1, m, -1.5, 1.5))
    want, _ = R.implicit_forward(seeded_sd, latent, pts)
    got, _ = net(latent.cuda(), None, pts.cuda(), need_attn=False)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=ATOL, rtol=0)


def test_empty_points(net):
    latent = torch.from_numpy(syn.seeded_latent(seed=0, batch=1)).cuda()
    lg, at = net(latent, None, torch.zeros(1, 0, 3).cuda())
    assert lg.shape == (1, 0) and at.shape == (1, 0, 197)


def test_grid32_full_vs_golden(net, decoder_golden):
    """vox_res = 32 through compute_level_grid's fused path: occupancy bits + values."""
    from zeroshape_amd.utils import eval_3D as E
    from zeroshape_amd.utils.options import EasyDict as edict
    latent = torch.from_numpy(syn.seeded_latent(seed=0, batch=2))[:1].cuda()
    opt = edict(dict(device="cuda", H=224, W=224, eval=dict(vox_res=32, range=[-1.5, 1.5]),
                     arch=dict(win_size=16)))
    var = edict(dict(idx=[0]))
    grid = E.get_dense_3D_grid(opt, var)
    assert grid.shape == (1, 33, 33, 33, 3)
    occ, vis = E.compute_level_grid(opt, net, latent, None, grid, None, vis_attn=False)
    assert vis is None and occ.shape == (1, 33, 33, 33)
    occ = occ[0].cpu().numpy()
    np.testing.assert_allclose(occ[::5, ::5, ::5], decoder_golden["occ32_stride5"], atol=ATOL, rtol=0)
    bits = np.unpackbits(decoder_golden["occ32_bits"])[: occ.size].astype(bool)
    mism = (occ > 0.5).reshape(-1) != bits
    # logits of the golden for three slices pin the band check
    logit = np.log(occ / (1 - occ)).reshape(-1)
    assert np.all(np.abs(logit[mism]) < BAND), "occupancy flip outside the rounding band"
    assert mism.sum() <= 2
    # raw logits for the stored slices
    lg = net.query_grid(latent, grid._zs_grid.axis, apply_sigmoid=False)[0].cpu().numpy()
    for i in (0, 16, 32):
        np.testing.assert_allclose(lg[i].reshape(-1), decoder_golden["logit32_slice%d" % i], atol=ATOL, rtol=0)


def test_generic_slice_loop_equals_fused_grid(net):
    """compute_level_grid on a tensor NOT made by get_dense_3D_grid takes the reference's
    slice loop through impl_network(...); must agree with the fused grid launch bit for bit
    (same kernel arithmetic, only the coordinate source differs)."""
    from zeroshape_amd.utils import eval_3D as E
    from zeroshape_amd.utils.options import EasyDict as edict
    latent = torch.from_numpy(syn.seeded_latent(seed=1, batch=2)).cuda()
    opt = edict(dict(device="cuda", H=224, W=224, eval=dict(vox_res=8, range=[-1.5, 1.5]),
                     arch=dict(win_size=16)))
    var = edict(dict(idx=[0, 1]))
    grid = E.get_dense_3D_grid(opt, var)
    fused, _ = E.compute_level_grid(opt, net, latent, None, grid, None)
    plain = grid.clone()            # loses the _zs_grid tag
    loop, _ = E.compute_level_grid(opt, net, latent, None, plain, None)
    assert torch.equal(fused, loop)
    # a tagged tensor that was modified in place is read, not regenerated from its axis
    moved = E.get_dense_3D_grid(opt, var)
    moved[..., 0] += 0.25
    got, _ = E.compute_level_grid(opt, net, latent, None, moved, None)
    want, _ = E.compute_level_grid(opt, net, latent, None, moved.clone(), None)
    assert torch.equal(got, want) and not torch.equal(got, fused)


@pytest.mark.parametrize("N", [64, 128])
def test_grid_slices_vs_golden(net, decoder_golden, N):
    latent = torch.from_numpy(syn.seeded_latent(seed=0, batch=2))[:1].cuda()
    axis = torch.linspace(-1.5, 1.5, N + 1, device="cuda")
    np.testing.assert_array_equal(axis.cpu().numpy(), decoder_golden["linspace_%d" % N])
    st = net.prepare(latent)
    for i in (0, N // 2, N):
        lg = net.query_grid(latent, axis, apply_sigmoid=False, slice_begin=i, slice_end=i + 1, state=st)
        got = lg[0, 0].reshape(-1)[::16].cpu().numpy()
        np.testing.assert_allclose(got, decoder_golden["logit%d_slice%d_s16" % (N, i)], atol=ATOL, rtol=0)


def test_full_size_grid128_properties(net, seeded_sd):
    """BASELINE config: 129^3 points.  Size-independent properties: (1) slab decomposition is
    exact (any x-slab of the full launch equals a separate launch of that slab) - the basis
    of the multi-GPU sharding; (2) spot-check 2048 random grid points against the oracle;
    (3) occupancy fraction is sane and identical between sigmoid>0.5 and logit>0."""
    N = 128
    latent_c = torch.from_numpy(syn.seeded_latent(seed=0, batch=1))
    latent = latent_c.cuda()
    axis = torch.linspace(-1.5, 1.5, N + 1, device="cuda")
    st = net.prepare(latent)
    full = net.query_grid(latent, axis, apply_sigmoid=False, state=st)
    assert full.shape == (1, N + 1, N + 1, N + 1)
    slab = net.query_grid(latent, axis, apply_sigmoid=False, slice_begin=40, slice_end=57, state=st)
    assert torch.equal(full[:, 40:57], slab)
    occ = net.query_grid(latent, axis, apply_sigmoid=True, state=st)
    # sigmoid(x) > 0.5 <=> x > 0 except where fp32 rounds 1 + exp(-x) to 2 (|x| < 6e-8)
    mism = (occ > 0.5) != (full > 0)
    assert int(mism.sum()) <= 4 and bool(torch.all(full[mism].abs() < 2e-7))
    frac = float((full > 0).float().mean())
    assert 0.05 < frac < 0.95
    rs = np.random.RandomState(7)
    idx = rs.randint(0, N + 1, size=(2048, 3))
    ax = axis.cpu()
    pts = torch.stack([ax[idx[:, 0]], ax[idx[:, 1]], ax[idx[:, 2]]], -1)[None]
    want, _ = R.implicit_forward(seeded_sd, latent_c, pts)
    got = full[0, idx[:, 0], idx[:, 1], idx[:, 2]].cpu().numpy()
    np.testing.assert_allclose(got, want[0].numpy(), atol=ATOL, rtol=0)
    flips = (got > 0) != (want[0].numpy() > 0)
    assert np.all(np.abs(want[0].numpy()[flips]) < BAND)


def test_vox256_slab_and_batched_grid(net, seeded_sd):
    """BASELINE config 5 geometry (257^3, sharded): one rank's slab of a batch of 2 images,
    checked against the oracle on random points; slab launches of a batch equal per-image launches."""
    N = 256
    latent_c = torch.from_numpy(syn.seeded_latent(seed=2, batch=2))
    latent = latent_c.cuda()
    axis = torch.linspace(-1.5, 1.5, N + 1, device="cuda")
    st = net.prepare(latent)
    slab = net.query_grid(latent, axis, apply_sigmoid=False, slice_begin=224, slice_end=226, state=st)
    assert slab.shape == (2, 2, N + 1, N + 1)
    one = net.query_grid(latent[1:], axis, apply_sigmoid=False, slice_begin=224, slice_end=226)
    assert torch.equal(slab[1:], one)
    rs = np.random.RandomState(5)
    jj, kk = rs.randint(0, N + 1, 500), rs.randint(0, N + 1, 500)
    ax = axis.cpu()
    pts = torch.stack([ax[225].expand(500), ax[jj], ax[kk]], -1)[None].repeat(2, 1, 1)
    want, _ = R.implicit_forward(seeded_sd, latent_c, pts)
    got = slab[:, 1, jj, kk].cpu().numpy()
    np.testing.assert_allclose(got, want.numpy(), atol=ATOL, rtol=0)


def test_weights_update_repacks(net, seeded_sd):
    latent = torch.from_numpy(syn.seeded_latent(seed=0, batch=1)).cuda()
    pts = torch.from_numpy(syn.seeded_cloud(5, 1, 64, -1, 1)).cuda()
    a, _ = net(latent, None, pts, need_attn=False)
    with torch.no_grad():
        net.impl_mlp.layers[8].bias.add_(0.25)
    b, _ = net(latent, None, pts, need_attn=False)
    np.testing.assert_allclose((b - a).cpu().numpy(), 0.25, atol=1e-6)
    with torch.no_grad():
        net.impl_mlp.layers[8].bias.sub_(0.25)


def test_unsupported_configs_raise():
    from zeroshape_amd.model.shape.implicit import Implicit
    m = Implicit(196, latent_dim=256, n_channels=256, n_blocks_attn=2, n_layers_mlp=8, num_heads=16,
                 skip_in=[2, 4, 6], pos_perlayer=False).cuda().eval()
    with pytest.raises(NotImplementedError):
        m(torch.zeros(1, 197, 256).cuda(), None, torch.zeros(1, 4, 3).cuda())


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_pos_perlayer_variant_vs_reference_golden_and_oracle(seeded_sd, decoder_golden, prec):
    """Implicit(pos_perlayer=True), the reference class's own default (model/shape/implicit.py:197,269-272): a prologue option
    since round 5 (zs_sdf_prologue_ex, ZS_SDF_POS_PERLAYER) - both arithmetics against the golden of the reference itself, the
    grid path against the oracle, and the training path (layer-by-layer autograd) against torch autograd on the oracle."""
    from oracle import decoder_ref as R
    from zeroshape_amd.model.shape.implicit import Implicit
    from zeroshape_amd.nn import autograd as A
    m = Implicit(syn.NUM_PATCHES, latent_dim=syn.LATENT_DIM, semantic=False, n_channels=syn.N_CHANNELS,
                 n_blocks_attn=syn.ATT_BLOCKS, n_layers_mlp=syn.MLP_LAYERS, num_heads=syn.NUM_HEADS,
                 posenc_3D=0, mlp_ratio=syn.MLP_RATIO, skip_in=list(syn.SKIP_IN), pos_perlayer=True)
    m.load_state_dict(seeded_sd, strict=True)
    m = m.cuda().eval()
    m.precision = prec
    latent = torch.from_numpy(syn.seeded_latent(seed=0, batch=2))
    rs = np.random.RandomState(123)
    pts = torch.from_numpy(rs.uniform(-1, 1, size=(2, 4096, 3)).astype(np.float32))[:, :1024]
    got, attn = m(latent.cuda(), None, pts.cuda(), need_attn=True)
    np.testing.assert_allclose(got.cpu().numpy(), decoder_golden["pp_pts1024_logit"], atol=1.5e-5, rtol=0)
    np.testing.assert_allclose(attn[:, ::128].cpu().numpy(), decoder_golden["pp_pts1024_attn_rows"], atol=2e-7, rtol=0)
    # the grid path
    N = 8
    want = R.level_grid(seeded_sd, latent[:1], R.dense_grid(-1.5, 1.5, N))
    axis = torch.linspace(-1.5, 1.5, N + 1).cuda()
    occ = m.query_grid(latent[:1].cuda(), axis, apply_sigmoid=True)
    assert float((occ.cpu() - want).abs().max()) > 1e-3                        # (the oracle default is pos_perlayer=False)
    lg, _ = R.implicit_forward(seeded_sd, latent[:1], R.dense_grid(-1.5, 1.5, N).reshape(1, -1, 3), pos_perlayer=True)
    np.testing.assert_allclose(occ.cpu().numpy().reshape(-1), torch.sigmoid(lg).numpy().reshape(-1), atol=1e-5, rtol=0)
    if prec == "f16x3":
        return
    # training path: logits and gradients against torch autograd on the oracle
    m.train()
    m.drop_path = 0.0
    sd = {k: v.clone().requires_grad_(k != "pos_embed") for k, v in seeded_sd.items()}
    lat_c = latent.clone().requires_grad_(True)
    sdf = torch.from_numpy(rs.normal(0, 0.2, (2, 1024)).astype(np.float32))
    want_t = R.implicit_forward_train(sd, lat_c, pts, None, pos_perlayer=True)
    R.shape_loss(want_t, sdf).backward()
    lat_g = latent.clone().cuda().requires_grad_(True)
    got_t, _ = m(lat_g, None, pts.cuda(), need_attn=False)
    np.testing.assert_allclose(got_t.detach().cpu().numpy(), want_t.detach().numpy(), atol=5e-5, rtol=0)
    A.bce_logits(got_t, sdf.cuda()).backward()
    for k, p in m.named_parameters():
        if k == "pos_embed":
            continue
        w = sd[k].grad.double()
        assert float((p.grad.cpu().double() - w).norm()) <= 5e-5 * float(w.norm()) + 1e-9, k
    w = lat_c.grad.double()
    assert float((lat_g.grad.cpu().double() - w).norm()) <= 5e-5 * float(w.norm())


def test_level_grid_with_attention_frames_is_self_contained(net, seeded_sd):
    """vis_attn=True (utils/eval_3D.py:47-80, the demo GIF): the slice loop with the attention map requested, frames drawn by
    this package's own jet heat map (no cv2, no reference module on sys.path).  The occupancies are those of the fused
    path; the frame count and the maps behind the frames follow the reference's loop, checked against the oracle's attention."""
    from zeroshape_amd.utils import eval_3D as E
    from zeroshape_amd.utils.options import EasyDict as edict
    import sys
    vis = sys.modules.get("utils.util_vis")       # (compat.install() may alias this package's own module under that name)
    assert vis is None or "reference" not in (getattr(vis, "__file__", None) or "")
    assert not any(p.rstrip("/") == "/root/reference" for p in sys.path)
    latent = torch.from_numpy(syn.seeded_latent(seed=0, batch=1)).cuda()
    opt = edict(dict(device="cuda", H=224, W=224, eval=dict(vox_res=16, range=[-1.5, 1.5]), arch=dict(win_size=16)))
    grid = E.get_dense_3D_grid(opt, edict(dict(idx=[0])))
    images = torch.rand(1, 3, 224, 224, generator=torch.Generator().manual_seed(1)).cuda()
    occ, frames = E.compute_level_grid(opt, net, latent, None, grid, images, vis_attn=True)
    occ_fused, none = E.compute_level_grid(opt, net, latent, None, grid, None, vis_attn=False)
    assert none is None and float((occ - occ_fused).abs().max()) < 2e-5
    N = 17
    cols = len(range(0, N // 8 * 8 + 1, 8))
    assert len(frames) == 1 and len(frames[0]) == len(range(0, N, 8)) * cols
    f = frames[0][0]
    assert f.shape == (224, 224, 3) and f.dtype == np.float32 and 0.0 <= f.min() and abs(f.max() - 1.0) < 1e-6
    # the map behind frame 0 (col 0, row 0): mean over z of the oracle's attention at x = 0, y = 0
    _, attn = R.implicit_forward(seeded_sd, latent.cpu(), grid.cpu().view(1, N, N * N, 3)[:, 0])
    a = attn.view(1, N, N, 197).mean(2)[0, 0]
    a = (a[:1].sum() + a[1:].view(14, 14))
    a = torch.nn.functional.interpolate(a[None, None], size=(224, 224), mode="bilinear", align_corners=False)[0, 0].numpy()
    want = E.show_att_on_image(images[0].permute(1, 2, 0).cpu().numpy(), a / a.max())
    assert float(np.abs(want - f).max()) < 2e-2          # (uint8 quantisation of the heat map: one level of 255 can move)
    lut = E._jet_lut()
    assert lut.shape == (256, 3) and tuple(lut[0]) == (0, 0, 128) and tuple(lut[255]) == (128, 0, 0) and tuple(lut[128])[1] == 255
