"""GPU parity of the decoder's TRAINING path (Implicit in .train() mode through
zeroshape_amd/nn/autograd.py + Loss.shape_loss) against
  * the golden outputs of the REAL reference in train mode (tests/golden/decoder_train_golden.npz:
    logits, loss, every parameter gradient, the latent gradient, with the recorded DropPath draw);
  * torch autograd on the oracle (oracle/decoder_ref.py::implicit_forward_train) for other shapes.
Contract: 1e-4 (BASELINE.json); asserted at 5e-5 of each tensor's norm."""
import numpy as np
import pytest
import torch

from oracle import decoder_ref as R
from tests.test_oracle_decoder_train import check_grads
from zeroshape_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def make_net(seeded_sd):
    from zeroshape_amd.model.shape.implicit import Implicit
    m = Implicit(syn.NUM_PATCHES, latent_dim=syn.LATENT_DIM, semantic=False, n_channels=syn.N_CHANNELS,
                 n_blocks_attn=syn.ATT_BLOCKS, n_layers_mlp=syn.MLP_LAYERS, num_heads=syn.NUM_HEADS,
                 posenc_3D=0, mlp_ratio=syn.MLP_RATIO, skip_in=list(syn.SKIP_IN), pos_perlayer=False)
    m.load_state_dict(seeded_sd, strict=True)
    return m.cuda()


def test_train_step_matches_reference_golden(seeded_sd, decoder_train_golden):
    from zeroshape_amd.utils.loss import Loss
    from zeroshape_amd.utils.util import EasyDict as edict
    g = decoder_train_golden
    net = make_net(seeded_sd).train()
    net.drop_scales = [torch.from_numpy(s).cuda() for s in g["drop_scales"]]
    latent = torch.from_numpy(syn.seeded_latent(int(g["latent_seed"]), g["points"].shape[0])).cuda().requires_grad_(True)
    logits, attn = net(latent, None, torch.from_numpy(g["points"]).cuda(), need_attn=False)
    assert attn is None
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g["logits"], atol=5e-5, rtol=0)
    opt = edict(training=edict(shape_loss=edict(impt_weight=float(g["impt_weight"]), impt_thres=float(g["impt_thres"])),
                               depth_loss=edict(grad_reg=0.1, depth_inv=True, mask_shrink=False)))
    loss = Loss(opt).shape_loss(logits, torch.from_numpy(g["sdf"]).cuda())
    assert abs(float(loss) - float(g["loss"])) < 2e-6
    loss.backward()
    gn = float(g["grad_latent_norm"])
    assert abs(float(latent.grad.double().norm()) - gn) < 5e-5 * gn
    np.testing.assert_allclose(latent.grad.cpu().numpy()[:, ::13, ::17], g["grad_latent_sample"], atol=5e-5 * gn, rtol=0)
    grads = {k: p.grad.cpu() for k, p in net.named_parameters() if p.grad is not None}
    assert "pos_embed" not in grads
    check_grads(grads, g, rtol=5e-5)


@pytest.mark.parametrize("B,M,drop", [(1, 4096, False), (2, 77, True)])
def test_train_step_matches_oracle_autograd(seeded_sd, B, M, drop):
    net = make_net(seeded_sd).train()
    rs = np.random.RandomState(M)
    pts = torch.from_numpy(rs.uniform(-1, 1, (B, M, 3)).astype(np.float32))
    sdf = torch.from_numpy(rs.normal(0, 0.2, (B, M)).astype(np.float32))
    scales = [torch.tensor([0.0, 1 / 0.9][:B] if i % 2 else [1 / 0.9] * B) for i in range(4)] if drop else None
    if not drop:
        net.drop_path = 0.0
    sd = {k: v.clone().requires_grad_(k != "pos_embed") for k, v in seeded_sd.items()}
    lat_c = torch.from_numpy(syn.seeded_latent(3, B)).requires_grad_(True)
    want = R.implicit_forward_train(sd, lat_c, pts, scales)
    R.shape_loss(want, sdf).backward()
    net.drop_scales = None if scales is None else [s.cuda() for s in scales]
    lat_g = lat_c.detach().cuda().requires_grad_(True)
    from zeroshape_amd.nn import autograd as A
    got, _ = net(lat_g, None, pts.cuda(), need_attn=False)
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), atol=5e-5, rtol=0)
    A.bce_logits(got, sdf.cuda()).backward()
    for k, p in net.named_parameters():
        if k == "pos_embed":
            continue
        w = sd[k].grad.double()
        err = float((p.grad.cpu().double() - w).norm())
        assert err <= 5e-5 * float(w.norm()) + 1e-9, (k, err, float(w.norm()))
    w = lat_c.grad.double()
    assert float((lat_g.grad.cpu().double() - w).norm()) <= 5e-5 * float(w.norm())


def test_train_mode_draws_drop_path_and_eval_mode_does_not(seeded_sd):
    net = make_net(seeded_sd).train()
    lat = torch.from_numpy(syn.seeded_latent(0, 4)).cuda()
    pts = torch.rand(4, 64, 3, device="cuda") * 2 - 1
    torch.manual_seed(0)
    a = [net(lat, None, pts, need_attn=False)[0].detach() for _ in range(6)]
    assert any(not torch.equal(a[0], x) for x in a[1:])                 # stochastic depth is live
    net.eval()
    with torch.no_grad():
        fused, _ = net(lat, None, pts, need_attn=False)                   # fused inference kernel
    lat.requires_grad_(True)
    layered, _ = net(lat, None, pts, need_attn=False)                     # eval + autograd: layer path, no drop
    np.testing.assert_allclose(layered.detach().cpu().numpy(), fused.cpu().numpy(), atol=3e-5, rtol=0)


def test_few_adamw_steps_reduce_the_loss_and_refresh_the_fused_kernel(seeded_sd):
    """Optimiser writes go through raw pointers: the packed inference program must notice."""
    from zeroshape_amd.nn import autograd as A
    from zeroshape_amd.optim import FusedAdamW
    net = make_net(seeded_sd).train()
    net.drop_path = 0.0
    opt = FusedAdamW([dict(params=[p for p in net.parameters() if p.requires_grad], lr=1e-3, weight_decay=0.0)],
                     betas=(0.9, 0.95))
    lat = torch.from_numpy(syn.seeded_latent(0, 2)).cuda()
    rs = np.random.RandomState(0)
    pts = torch.from_numpy(rs.uniform(-1, 1, (2, 512, 3)).astype(np.float32)).cuda()
    sdf = (pts.norm(dim=-1) - 0.6).contiguous()
    with torch.no_grad():
        before = net.eval()(lat, None, pts, need_attn=False)[0].clone()
    net.train()
    losses = []
    for _ in range(8):
        loss = A.bce_logits(net(lat, None, pts, need_attn=False)[0], sdf)
        loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(float(loss))
    assert losses[-1] < 0.7 * losses[0], losses
    with torch.no_grad():
        after = net.eval()(lat, None, pts, need_attn=False)[0]
    assert float((after - before).abs().max()) > 1e-2                    # repacked, not the stale program
    net.train()
    lat.requires_grad_(False)
    layered = net(lat, None, pts, need_attn=False)[0]
    np.testing.assert_allclose(layered.detach().cpu().numpy(), after.cpu().numpy(), atol=5e-5, rtol=0)
