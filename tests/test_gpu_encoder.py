"""GPU parity of the encoder mirrors (DPTDepthModel, intrinsics head, CoordEncRes, CoordEncAtt,
Graph.forward) through the HIP layers vs oracle/encoder_ref.py and the golden outputs of the
reference's own modules.  The layers run in the engine's default arithmetic (zeroshape_amd/nn/ops.py:
CONV_PRECISION, "f16x3" = split-fp16 with round-to-nearest halves unless ZS_ENCODER_PRECISION=f32); a
~100-layer stack accumulates rounding, so activations are compared relative to their scale:
max|err| <= 1e-4 * max|want| (north_star's 1e-4), the depth map absolutely (values in [0,1]).  The chain
image -> latent -> occupancy grid is held to 1e-4 absolute for both arithmetics in
tests/test_gpu_image_to_occupancy.py."""
import numpy as np
import pytest
import torch

from oracle import encoder_ref as E
from tests.test_encoder_contract import make_opt
from zeroshape_amd import synthetic as syn
from zeroshape_amd.utils.options import EasyDict as edict

pytestmark = pytest.mark.gpu


def close(got, want, tol=1e-4, msg=""):
    got = np.asarray(got.detach().cpu() if torch.is_tensor(got) else got, np.float64)
    want = np.asarray(want.detach().cpu() if torch.is_tensor(want) else want, np.float64)
    assert got.shape == want.shape, "%s: %s vs %s" % (msg, got.shape, want.shape)
    err, scale = np.abs(got - want).max(), max(np.abs(want).max(), 1e-6)
    assert err <= tol * scale, "%s: err %.3g scale %.3g (rel %.3g)" % (msg, err, scale, err / scale)


def sample(x, step):
    return x.detach().cpu().numpy().reshape(-1)[::step]


@pytest.fixture(scope="module")
def graph(encoder_sd, seeded_sd):
    from zeroshape_amd.model.compute_graph.graph_shape import Graph
    g = Graph(make_opt())
    full = dict(encoder_sd)
    full.update({"impl_network." + k: v for k, v in seeded_sd.items()})
    g.load_state_dict(full, strict=True)
    return g.cuda().eval()


def test_dpt_depth_vs_oracle_and_golden(graph, encoder_sd, encoder_golden):
    rgb, _ = [torch.from_numpy(a) for a in syn.seeded_rgb_scene(seed=0, batch=2)]
    taps, otaps = {}, {}
    depth, feat = graph.dpt_depth(rgb.cuda(), get_feat=True, taps=taps)
    odepth, ofeat = E.dpt_depth(E._sub(encoder_sd, "dpt_depth."), rgb, otaps)
    for name in ("stage0", "stage1", "stage2", "block0", "block8", "block11", "layer3_rn", "layer4_rn", "path4",
                 "path3", "path2", "path1"):
        t = taps[name]
        t = t.permute(0, 3, 1, 2) if t.dim() == 4 else t           # channels-last -> NCHW
        close(t, otaps[name], msg=name)
        close(sample(t.contiguous(), 997), encoder_golden["dpt_%s_s997" % name], msg="golden " + name)
    assert depth.shape == (2, 1, 224, 224) and feat.shape == (2, 768, 7, 7)
    close(feat, ofeat, msg="layer_4")
    np.testing.assert_allclose(depth.cpu().numpy(), odepth.numpy(), atol=1e-4, rtol=0)
    np.testing.assert_allclose(sample(depth, 211), encoder_golden["depth_s211"], atol=1e-4, rtol=0)
    close(sample(feat, 53), encoder_golden["intr_feat_s53"], msg="golden layer_4")
    assert float(depth.min()) >= 0 and float(depth.max()) <= 1
    # get_feat=False returns the depth alone (dpt_depth.py:121-122)
    assert torch.equal(graph.dpt_depth(rgb.cuda()), depth)


def test_graph_forward_vs_oracle_and_golden(graph, encoder_sd, encoder_golden):
    rgb, mask = [torch.from_numpy(a) for a in syn.seeded_rgb_scene(seed=0, batch=2)]
    opt = make_opt()
    opt.arch.depth.dsp = 1
    var = edict(dict(idx=torch.arange(2), rgb_input_map=rgb.cuda(), mask_input_map=mask.cuda(),
                     pose_gt=torch.zeros(2, 3, 4).cuda()))
    var = graph.forward(opt, var, training=False, get_loss=False)
    want = E.graph_forward(encoder_sd, rgb, mask)
    np.testing.assert_allclose(var.depth_pred.cpu().numpy(), want["depth_pred"].numpy(), atol=1e-4, rtol=0)
    close(var.intr_pred, want["intr_pred"], msg="intr_pred")
    np.testing.assert_allclose(var.intr_pred.cpu().numpy(), encoder_golden["g_intr_pred"], rtol=2e-4, atol=2e-3)
    # seen points: the unit-ball normalisation divides by the max radius -> absolute 2e-4
    np.testing.assert_allclose(var.seen_points.cpu().numpy(), want["seen_points"].numpy(), atol=2e-4, rtol=0)
    np.testing.assert_allclose(sample(var.seen_points, 101), encoder_golden["g_seen_points_s101"], atol=2e-4, rtol=0)
    assert var.latent_depth.shape == (2, 197, 256)
    close(var.latent_depth, want["latent_depth"], msg="latent_depth")
    close(sample(var.latent_depth, 37), encoder_golden["g_latent_depth_s37"], msg="golden latent")
    assert torch.equal(var.validity_mask.cpu(), (mask > 0.5).float().view(2, -1))
    assert var.latent_semantic is None
    # the decoder takes the latent as is (graph_shape.py:185 call shape)
    pts = torch.from_numpy(np.random.RandomState(0).uniform(-1, 1, (2, 512, 3)).astype(np.float32)).cuda()
    logits, attn = graph.impl_network(var.latent_depth, None, pts)
    assert logits.shape == (2, 512) and attn.shape == (2, 512, 197) and bool(torch.isfinite(logits).all())


def test_coord_enc_res(graph, encoder_sd, encoder_golden):
    _, mask, _ = [torch.from_numpy(a) for a in syn.seeded_depth_scene(seed=1, batch=2)]
    coord = torch.from_numpy(np.random.RandomState(11).uniform(-1, 1, size=(2, 3, 224, 224)).astype(np.float32))
    lat = graph.coord_encoder(coord.cuda(), mask.cuda())
    want = E.coord_enc_res(E._sub(encoder_sd, "coord_encoder."), coord, mask)
    assert lat.shape == (2, 197, 256)
    close(lat, want, msg="CoordEncRes")
    close(sample(lat, 37), encoder_golden["res_latent_s37"], msg="golden CoordEncRes")


def test_coord_enc_att(att_sd, encoder_golden):
    from zeroshape_amd.model.shape.seen_coord_enc import CoordEncAtt
    enc = CoordEncAtt(embed_dim=256, n_blocks=12, num_heads=8, win_size=8)
    enc.load_state_dict(att_sd, strict=True)
    enc = enc.cuda().eval()
    _, mask, _ = [torch.from_numpy(a) for a in syn.seeded_depth_scene(seed=1, batch=2)]
    coord = torch.from_numpy(np.random.RandomState(12).uniform(-1, 1, size=(2, 112, 112, 3)).astype(np.float32))
    m = torch.nn.functional.interpolate(mask, (112, 112)) > 0.5
    lat = enc(coord.cuda(), m[:, 0].cuda())
    want = E.coord_enc_att(att_sd, coord, m[:, 0])
    assert lat.shape == (2, 197, 256)
    close(lat, want, msg="CoordEncAtt")
    close(sample(lat, 37), encoder_golden["att_latent_s37"], msg="golden CoordEncAtt")


def test_repack_after_weight_update(graph):
    """In-place parameter changes (optimizer step, load_state_dict) invalidate the packed copy."""
    rgb, _ = [torch.from_numpy(a) for a in syn.seeded_rgb_scene(seed=0, batch=1)]
    d0 = graph.dpt_depth(rgb.cuda()).clone()
    b = graph.dpt_depth.scratch.output_conv[4].bias
    with torch.no_grad():
        b.add_(-0.25)
    d1 = graph.dpt_depth(rgb.cuda()).clone()
    with torch.no_grad():
        b.add_(0.25)
    d2 = graph.dpt_depth(rgb.cuda())
    assert float((d0 - d1).abs().max()) > 0.1 and torch.equal(d0, d2)


def test_batch_of_one_and_other_resolution(graph, encoder_sd):
    """demo.py feeds B=1; DPT's flexible position embedding (vit.py:103-120) admits other sizes."""
    rgb = torch.from_numpy(syn.seeded_rgb_scene(seed=2, batch=1, size=160)[0])
    depth, feat = graph.dpt_depth(rgb.cuda(), get_feat=True)
    odepth, ofeat = E.dpt_depth(E._sub(encoder_sd, "dpt_depth."), rgb)
    assert depth.shape == (1, 1, 160, 160) and feat.shape == (1, 768, 5, 5)
    np.testing.assert_allclose(depth.cpu().numpy(), odepth.numpy(), atol=1e-4, rtol=0)
    close(feat, ofeat, msg="layer_4 @160")


def test_hip_graph_replay_equals_eager(graph):
    """The captured hipGraph replays the same launches: bit-identical outputs, also for new inputs
    of the captured shape."""
    opt = make_opt()
    opt.arch.depth.dsp = 1
    outs = {}
    for mode in (False, True, True):
        graph.enable_hip_graph(mode) if mode != graph._use_hip_graph else None
        for seed in (0, 4):
            rgb, mask = [torch.from_numpy(a).cuda() for a in syn.seeded_rgb_scene(seed=seed, batch=2)]
            var = edict(dict(idx=[0, 1], rgb_input_map=rgb, mask_input_map=mask))
            var = graph.forward(opt, var, training=False, get_loss=False)
            got = [var.depth_pred, var.intr_pred, var.seen_points, var.latent_depth]
            if seed in outs:
                for a, b in zip(outs[seed], got):
                    assert torch.equal(a, b)
            else:
                outs[seed] = [t.clone() for t in got]
    assert len(graph._captured) == 1
    graph.enable_hip_graph(False)


@pytest.mark.parametrize("B,size", [(2, 160), (2, 96), (2, 128), (3, 64)])
def test_batches_of_small_images_through_the_fused_trunk_gate(graph, encoder_sd, B, size, monkeypatch):
    """with B > 1 the fused GroupNorm path needs every map of the ResNetV2 trunk to hold a multiple of 32 pixels
    (statistics tiles must not straddle samples); the gate looked at the first two maps only and 2 x 160 x 160 died in
    stage 1 (20 x 20 pixels).  Batches of small images now run - fused where every map allows it (128: 64^2, 32^2, 16^2,
    8^2), unfused otherwise - and agree with the unfused engine and the oracle."""
    from zeroshape_amd.nn import blocks
    rgb = torch.from_numpy(syn.seeded_rgb_scene(seed=4, batch=B, size=size)[0])
    depth, feat = graph.dpt_depth(rgb.cuda(), get_feat=True)
    monkeypatch.setattr(blocks, "FUSED_GN_MAX_ROWS", 0)                     # the unfused engine
    depth_u, feat_u = graph.dpt_depth(rgb.cuda(), get_feat=True)
    close(feat, feat_u, tol=2e-5, msg="fused vs unfused trunk")
    np.testing.assert_allclose(depth.cpu().numpy(), depth_u.cpu().numpy(), atol=2e-5, rtol=0)
    odepth, ofeat = E.dpt_depth(E._sub(encoder_sd, "dpt_depth."), rgb)
    np.testing.assert_allclose(depth.cpu().numpy(), odepth.numpy(), atol=1e-4, rtol=0)
    close(feat, ofeat, msg="layer_4 @%d x %d" % (B, size))
    assert blocks._fused_maps_ok(B, size, size, 3) == (size == 128)


def test_batch_28_takes_the_large_tile_kernels_and_agrees_with_small_batches_and_the_oracle(graph, encoder_sd):
    """options/shape.yaml's batch of 28: the ViT qkv / fc1 / fc2 layers run the 256 x 256 ping-pong GEMM kernel (csrc/nn_conv_pp256.h,
    incl. its K-range tails: 264 = 256 + 8 tiles on fc1, 66 x 3 ranges on fc2) and attention with K / V staged in LDS - kernels the
    batch-2 tests above never select.  The same images in batches of 2 (128 x 128 / streaming kernels, register-only attention)
    give the same depth, tap-4 feature and latent to summation order, and the first two images agree with the CPU oracle."""
    B = 28
    rgb, mask = [torch.from_numpy(a) for a in syn.seeded_rgb_scene(seed=3, batch=B)]
    depth, feat = graph.dpt_depth(rgb.cuda(), get_feat=True)
    assert depth.shape == (B, 1, 224, 224) and feat.shape == (B, 768, 7, 7)
    d2, f2 = [], []
    for i in range(0, 8, 2):
        a, b = graph.dpt_depth(rgb[i:i + 2].cuda(), get_feat=True)
        d2.append(a)
        f2.append(b)
    close(feat[:8], torch.cat(f2).cpu(), tol=2e-5, msg="tap-4 feature, batch 28 vs batches of 2")
    np.testing.assert_allclose(depth[:8].cpu().numpy(), torch.cat(d2).cpu().numpy(), atol=2e-5, rtol=0)
    odepth, ofeat = E.dpt_depth(E._sub(encoder_sd, "dpt_depth."), rgb[:2])
    np.testing.assert_allclose(depth[:2].cpu().numpy(), odepth.numpy(), atol=1e-4, rtol=0)
    close(feat[:2], ofeat, msg="layer_4 at batch 28 vs oracle")
    # the whole graph: latent codes
    opt = make_opt()
    opt.arch.depth.dsp = 1
    var = edict(dict(idx=torch.arange(B), rgb_input_map=rgb.cuda(), mask_input_map=mask.cuda(), pose_gt=torch.zeros(B, 3, 4).cuda()))
    var = graph.forward(opt, var, training=False, get_loss=False)
    v2 = edict(dict(idx=torch.arange(2), rgb_input_map=rgb[:2].cuda(), mask_input_map=mask[:2].cuda(), pose_gt=torch.zeros(2, 3, 4).cuda()))
    v2 = graph.forward(opt, v2, training=False, get_loss=False)
    close(var.latent_depth[:2], v2.latent_depth.cpu(), tol=5e-5, msg="latent, batch 28 vs batch 2")


def test_k16_major_hidden_tensor_changes_no_bit(graph, monkeypatch):
    """Batch 1: the ViT MLP's hidden tensor travels between fc1 and fc2 in K16-major layout ([hidden / 16][rows][16],
    ZS_CONV_OUT_K16 / ZS_CONV_IN_K16: the streaming GEMM kernel's operand loads become whole lines).  Same fragments, same
    products, same order: the forward is bit-identical to the row-major one; a layer the kernel does not take is refused."""
    from zeroshape_amd import _lib
    from zeroshape_amd.nn import ops, pack
    lib = _lib.load()
    assert ops.k16_ok(197, 768, 3072, out_k16=True, ln_tiles=12) and ops.k16_ok(197, 3072, 768, in_k16=True, has_res=True)
    assert not ops.k16_ok(197, 3072, 768, out_k16=True, has_res=True)           # no residual into a K16-major output
    assert not ops.k16_ok(28 * 197, 768, 3072, out_k16=True)                   # batch 28: other kernels serve the layer
    rgb = torch.from_numpy(syn.seeded_rgb_scene(seed=3, batch=1)[0]).cuda()
    outs = []
    for on in (True, False, True):
        monkeypatch.setattr(ops, "K16_HIDDEN", on)
        depth, feat = graph.dpt_depth(rgb, get_feat=True)
        outs.append((depth.clone(), feat.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])
    # the pair on its own: gelu(x W1 + b1) W2 + b2 + r with the hidden tensor in either layout, and against torch
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 197, 768, generator=g).cuda()
    w1, b1 = torch.randn(3072, 768, generator=g) * 0.03, torch.randn(3072, generator=g) * 0.1
    w2, b2 = torch.randn(768, 3072, generator=g) * 0.02, torch.randn(768, generator=g) * 0.1
    p1, p2 = pack.pack_conv(w1, b1).to("cuda"), pack.pack_conv(w2, b2).to("cuda")
    monkeypatch.setattr(ops, "K16_HIDDEN", True)
    h_rows = ops.linear(x, p1, act=ops.ACT_GELU)
    ya = ops.linear(h_rows, p2, res1=x)
    hb = ops.linear(x, p1, act=ops.ACT_GELU, out_k16=True)
    yb = ops.linear(hb, p2, res1=x, in_k16=True)
    assert torch.equal(hb.view(3072 // 16, 197, 16).permute(1, 0, 2).reshape(1, 197, 3072), h_rows)
    # (outside the fused block the row-major fc2 of this shape may take another kernel - the K-split plan the K16 form excludes -
    # and sum in another order: equal to rounding here, bit-equal inside the forward above where both forms run the same plan)
    assert float((ya - yb).abs().max()) < 2e-5
    want = torch.nn.functional.gelu(x.cpu().double() @ w1.double().T + b1.double()) @ w2.double().T + b2.double() + x.cpu().double()
    assert float((ya.cpu().double() - want).abs().max()) < 2e-4
    with pytest.raises(_lib.ZeroShapeHipError):                                # a shape the streaming kernel does not take
        ops.linear(torch.randn(1, 8, 768, device="cuda"), p1, out_k16=True)
