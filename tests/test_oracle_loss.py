"""Oracle of the depth task's losses (oracle/loss_ref.py) vs golden values and gradients of the REAL
reference's Loss.depth_loss (MiDaS) and Loss.intr_loss (tests/golden/loss_golden.npz)."""
import os
import sys

import numpy as np
import torch

from oracle import loss_ref as R

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def golden():
    return dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss_golden.npz")))


def test_midas_and_intr_loss_match_reference():
    from make_loss_golden import inputs
    g = golden()
    pred, target, mask, seen_pred, seen_gt, smask = inputs()
    p = pred.clone().requires_grad_(True)
    l = R.midas_loss(p, target, mask)
    l.backward()
    assert abs(float(l) - float(g["depth_loss"])) < 1e-6 * abs(float(g["depth_loss"])) + 1e-7
    assert abs(float(p.grad.double().norm()) - float(g["depth_grad_norm"])) < 1e-4 * float(g["depth_grad_norm"])
    np.testing.assert_allclose(p.grad.reshape(-1)[::97].numpy(), g["depth_grad_s97"], atol=2e-5 * float(np.abs(g["depth_grad_s97"]).max()))
    sp = seen_pred.clone().requires_grad_(True)
    l2 = R.intr_loss(sp, seen_gt, smask)
    l2.backward()
    assert abs(float(l2) - float(g["intr_loss"])) < 1e-6
    np.testing.assert_allclose(sp.grad.reshape(-1)[::7].numpy(), g["intr_grad_s7"], atol=1e-8)
    mask2 = mask.clone()
    mask2[1] = 0
    p = pred.clone().requires_grad_(True)
    l3 = R.midas_loss(p, target, mask2)
    l3.backward()
    assert abs(float(l3) - float(g["depth_loss_empty"])) < 1e-6
    np.testing.assert_allclose(p.grad.reshape(-1)[::97].numpy(), g["depth_grad_empty_s97"],
                               atol=2e-5 * float(np.abs(g["depth_grad_empty_s97"]).max()))
    assert float(p.grad[1].abs().max()) == 0


def test_mask_shrink_matches_reference():
    """training.depth_loss.mask_shrink: MidasLoss.erode_mask + the loss on the eroded mask."""
    from make_loss_golden import inputs
    g = golden()
    pred, target, mask, _, _, _ = inputs()
    er = R.erode_mask(mask)
    assert int(er.sum()) == int(g["eroded_count"])
    np.testing.assert_array_equal(np.packbits(er.numpy().reshape(-1)), g["eroded_bits"])
    p = pred.clone().requires_grad_(True)
    l = R.midas_loss(p, target, mask, shrink_mask=True)
    l.backward()
    assert abs(float(l) - float(g["depth_loss_shrink"])) < 1e-6
    np.testing.assert_allclose(p.grad.reshape(-1)[::97].numpy(), g["depth_grad_shrink_s97"],
                               atol=2e-5 * float(np.abs(g["depth_grad_shrink_s97"]).max()))
