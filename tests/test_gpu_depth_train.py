"""GPU parity of the depth task's losses (zs_midas_loss / zs_intr_loss, forward and backward) against
the golden values of the REAL reference and against torch autograd on the oracle (oracle/loss_ref.py);
and training steps of the depth engine (graph_depth.Graph.forward(training=True) -> depth + intr
losses -> backward -> fused AdamW)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import loss_ref as R
from zeroshape_amd.data.synthetic import Dataset
from zeroshape_amd.utils import options

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_midas_and_intr_loss_match_reference_golden_and_oracle():
    from make_loss_golden import inputs
    from zeroshape_amd.nn import autograd as A
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "loss_golden.npz")))
    pred, target, mask, seen_pred, seen_gt, smask = inputs()
    for name, m in (("depth_loss", mask), ("depth_loss_empty", None)):
        if m is None:
            m = mask.clone()
            m[1] = 0
        p = pred.cuda().requires_grad_(True)
        loss = A.midas_loss(p, target.cuda(), m.cuda(), alpha=0.1, scales=4, inverse_depth=True) * 1.5
        assert abs(float(loss) / 1.5 - float(g[name])) < 2e-5 * float(g[name])
        loss.backward()
        po = pred.clone().requires_grad_(True)
        (R.midas_loss(po, target, m) * 1.5).backward()
        assert rel(p.grad, po.grad) < 2e-4, rel(p.grad, po.grad)
        key = "depth_grad_s97" if name == "depth_loss" else "depth_grad_empty_s97"
        np.testing.assert_allclose(p.grad.cpu().reshape(-1)[::97].numpy() / 1.5, g[key], atol=3e-4 * float(np.abs(g[key]).max()))
    # without the regulariser, and on plain (not inverse) depth
    for alpha, inv in ((0.0, True), (0.3, False)):
        p, po = pred.cuda().requires_grad_(True), pred.clone().requires_grad_(True)
        lg = A.midas_loss(p, target.cuda(), mask.cuda(), alpha=alpha, scales=3, inverse_depth=inv)
        lo = R.midas_loss(po, target, mask, alpha=alpha, scales=3, inverse_depth=inv)
        assert abs(float(lg) - float(lo)) < 2e-5 * abs(float(lo))
        lg.backward()
        lo.backward()
        assert rel(p.grad, po.grad) < 2e-4
    sp = seen_pred.cuda().requires_grad_(True)
    l2 = A.intr_loss(sp, seen_gt.cuda(), smask.cuda())
    assert abs(float(l2) - float(g["intr_loss"])) < 1e-6
    l2.backward()
    np.testing.assert_allclose(sp.grad.cpu().reshape(-1)[::7].numpy(), g["intr_grad_s7"], atol=1e-8)


def test_median_select_ties_and_even_counts():
    """torch.nanmedian returns the LOWER median; duplicates and tiny / even valid counts."""
    from zeroshape_amd.nn import autograd as A
    g = torch.Generator().manual_seed(3)
    pred = torch.rand(4, 1, 16, 16, generator=g) + 0.2
    pred[0, 0, :4] = 0.5                                   # many duplicates around the median
    target = torch.rand(4, 1, 16, 16, generator=g) + 0.2
    mask = (torch.rand(4, 1, 16, 16, generator=g) > 0.3).float()
    mask[2] = 0
    mask[2, 0, 0, :2] = 1                                  # two valid pixels
    mask[3] = 0
    mask[3, 0, 5, 5] = 1                                   # one valid pixel
    lg = A.midas_loss(pred.cuda(), target.cuda(), mask.cuda(), alpha=0.0)
    lo = R.midas_loss(pred, target, mask, alpha=0.0)
    assert abs(float(lg) - float(lo)) < 1e-5 * abs(float(lo))


def depth_opt(tmp_path, *extra):
    cmd = options.parse_arguments(["--yaml=%s/options/depth.yaml" % ROOT, "--output_root=%s" % tmp_path, "--batch_size=4",
                                   "--max_epoch=1", "--arch.depth.pretrained=", "--eval.batch_size=2", "--optim.lr=2.e-6"]
                                  + list(extra))
    opt = options.set(cmd)
    opt.world_size = 1
    return opt


@pytest.mark.parametrize("extra", [(), ("--optim.hip_graph",), ("--optim.hip_graph", "--optim.amp")],
                         ids=["eager", "captured", "captured-amp"])
def test_depth_engine_trains(tmp_path, encoder_sd, extra):
    """eager launches; the step as a captured hipGraph (optim.hip_graph: the MiDaS loss with its device-side medians is
    part of the capture); the same with split-fp16 forward / data gradients under the loss scaler."""
    from zeroshape_amd.nn import autograd as A
    try:
        _depth_engine_trains(tmp_path, encoder_sd, extra)
    finally:
        A.set_forward_precision("f32")
        A.set_backward_precision("f32")


def _depth_engine_trains(tmp_path, encoder_sd, extra):
    from zeroshape_amd.model.depth_engine import Runner
    from zeroshape_amd.utils import util
    from zeroshape_amd.utils.options import EasyDict as edict
    opt = depth_opt(tmp_path, *extra)
    r = Runner(opt)
    r.load_dataset(opt, dataset=Dataset(opt, n_items=2, load_3D=False),
                   train_dataset=Dataset(opt, split="train", n_items=4, load_3D=False, seed=1))
    r.build_networks(opt)
    sd = {k: v for k, v in encoder_sd.items() if k.startswith(("dpt_depth.", "intr_head.", "intr_proj."))}
    r.graph.load_state_dict(sd, strict=True)
    r.setup_optimizer(opt)
    r.restore_checkpoint(opt)
    assert [g["weight_decay"] for g in r.optim.param_groups] == [0.0, 0.05] and {g["lr"] for g in r.optim.param_groups} == {2e-6}
    before = {k: v.detach().clone() for k, v in r.graph.state_dict().items()}
    r.graph.train()
    batch = next(iter(r.train_loader))
    losses = []
    for it in range(5):
        var = util.move_to_device(edict({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}), opt.device)
        loss = r.train_iteration(opt, var)
        assert set(loss.keys()) == {"depth", "intr", "all"}
        losses.append((float(loss.depth), float(loss.intr), float(loss.all)))
        assert abs(losses[-1][2] - (losses[-1][0] + 10 * losses[-1][1])) < 1e-4 * abs(losses[-1][2])    # loss_weight 1 / 10
    assert np.isfinite(losses).all() and losses[-1][2] < losses[0][2], losses
    assert (getattr(r, "_captured", None) is not None) == ("--optim.hip_graph" in extra)
    if "--optim.amp" in extra:
        # GradScaler's rules: a scaled gradient beyond fp16's range is an overflow (round 3: the split halves round to
        # nearest and overflow to inf, where rounds 1-2 saturated silently) - the step is skipped and the scale halved.
        # At 2^16 the untrained depth head may take up to two of those; then the steps are clean.
        halvings = int(round(np.log2(65536.0 / float(r.scaler.scale))))
        assert float(r.scaler.scale) == 65536.0 / 2 ** halvings and 0 <= halvings <= 2
        assert float(r.scaler.found_inf) == 0.0 and 1 <= int(r.scaler.tracker) <= 5 - halvings
    after = r.graph.state_dict()
    moved = [k for k in before if before[k].is_floating_point() and not torch.equal(before[k], after[k])]
    assert any(k.startswith("intr_proj") for k in moved) and any(k.startswith("dpt_depth.scratch.output_conv") for k in moved)
    assert any(k.startswith("dpt_depth.pretrained.model.patch_embed.backbone.stem") for k in moved)
    val = r.evaluate(opt, ep=0)                                       # evaluation sees the trained weights
    assert np.isfinite(val)
    r.save_checkpoint(opt, ep=0, it=r.it, latest=True)
    ck = torch.load(os.path.join(opt.output_path, "latest.ckpt"), map_location="cpu")
    assert "optim" in ck and set(k.split(".")[0] for k in ck["graph"]) == {"dpt_depth", "intr_head", "intr_proj"}
    assert ("scaler" in ck) == ("--optim.amp" in extra)
    if "scaler" in ck:
        assert ck["scaler"]["scale"] == float(r.scaler.scale) and ck["scaler"]["_growth_tracker"] == int(r.scaler.tracker)


def test_mask_shrink_matches_reference_golden_and_oracle():
    """training.depth_loss.mask_shrink: zs_erode_mask bit for bit (golden of MidasLoss.erode_mask and the
    oracle on ragged sizes / fractional masks), and Loss.depth_loss on the eroded mask vs the golden."""
    from make_loss_golden import inputs
    from zeroshape_amd.nn import autograd as A
    from zeroshape_amd.utils.loss import Loss
    from zeroshape_amd.utils.options import EasyDict as edict
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "loss_golden.npz")))
    pred, target, mask, _, _, _ = inputs()
    er = A.erode_mask(mask.cuda())
    assert er.shape == mask.shape and int(er.sum()) == int(g["eroded_count"])
    np.testing.assert_array_equal(np.packbits(er.cpu().numpy().reshape(-1) > 0.5), g["eroded_bits"])
    gen = torch.Generator().manual_seed(3)
    for shape in ((2, 1, 9, 13), (1, 1, 4, 4), (3, 1, 31, 18), (1, 1, 225, 227)):
        m = (torch.rand(shape, generator=gen) > 0.1).float()
        m[0, 0, 0, 0] = 0.75                                   # fractional values are not "valid"
        want = R.erode_mask(m).float()
        assert torch.equal(A.erode_mask(m.cuda()).cpu(), want), shape
    opt = edict(training=edict(shape_loss=edict(impt_weight=1, impt_thres=0.01),
                               depth_loss=edict(grad_reg=0.1, depth_inv=True, mask_shrink=True)))
    p = pred.cuda().requires_grad_(True)
    loss = Loss(opt).depth_loss(p, target.cuda(), mask.cuda())
    assert abs(float(loss) - float(g["depth_loss_shrink"])) < 2e-5 * float(g["depth_loss_shrink"])
    loss.backward()
    np.testing.assert_allclose(p.grad.reshape(-1)[::97].cpu().numpy(), g["depth_grad_shrink_s97"],
                               atol=5e-5 * float(np.abs(g["depth_grad_shrink_s97"]).max()))
    from zeroshape_amd import _lib
    with pytest.raises(_lib.ZeroShapeHipError):
        A.erode_mask(torch.ones(1, 1, 3, 8).cuda())            # smaller than one pool window
