"""Oracle: the depth task's losses on the CPU (torch fp32, differentiable: torch autograd is the
gradient oracle).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates
  model/depth/midas_loss.py:6-9      masked_l1_loss
  model/depth/midas_loss.py:11-30    compute_scale_and_shift (least squares per image)
  model/depth/midas_loss.py:33-62    masked_shift_and_scale (median / mean-absolute-deviation alignment)
  model/depth/midas_loss.py:76-108   reduction_image_based, gradient_loss
  model/depth/midas_loss.py:112-185  SSIMAE, GradientMatchingTerm (4 scales), MidasLoss.forward
                                      (shrink_mask False, as options/*.yaml set it)
  utils/loss.py:30-43                Loss.depth_loss, Loss.intr_loss
Pinned by tests/golden/loss_golden.npz (tests/golden/make_loss_golden.py runs the reference's own
Loss on seeded inputs, values and gradients).
"""
import torch


def _median_valid(x, valid):
    """nanmedian over the valid pixels of each image (lower median), 0 where none: [B,1,1,1]."""
    xn = x.clone()
    xn[~valid] = float("nan")
    t = xn.view(x.shape[0], x.shape[1], -1).nanmedian(-1, keepdim=True)[0].unsqueeze(-1)
    return torch.where(torch.isnan(t), torch.zeros_like(t), t)


def _align(x, valid):
    """midas_loss.py:33-62 for one of the two maps: (x - median) / (mean |x - median| + 1e-6)."""
    count = valid.view(valid.shape[0], valid.shape[1], -1).sum(-1, keepdim=True) + 1
    t = _median_valid(x, valid)
    dev = torch.abs(x - t)
    dev = torch.where(valid, dev, torch.zeros_like(dev))
    s = (dev.view(x.shape[0], x.shape[1], -1).sum(-1, keepdim=True) / count).unsqueeze(-1)
    return (x - t) / (s + 1e-6)


def ssi_mae(pred, gt, valid):
    """SSIMAE (:112-119): masked L1 between the aligned maps, normalised by the batch's valid count."""
    err = torch.abs(_align(pred, valid) - _align(gt, valid))
    err = torch.where(valid, err, torch.zeros_like(err))
    return err.sum() / (valid.sum() + 1.e-6)


def scale_and_shift(prediction, target, mask):
    """:11-30, prediction / target / mask [B,H,W]."""
    m = mask.float()
    a_00, a_01, a_11 = (m * prediction * prediction).sum((1, 2)), (m * prediction).sum((1, 2)), m.sum((1, 2))
    b_0, b_1 = (m * prediction * target).sum((1, 2)), (m * target).sum((1, 2))
    det = a_00 * a_11 - a_01 * a_01
    ok = det != 0
    x_0 = torch.where(ok, (a_11 * b_0 - a_01 * b_1) / (det + 1e-6), torch.zeros_like(det))
    x_1 = torch.where(ok, (-a_01 * b_0 + a_00 * b_1) / (det + 1e-6), torch.zeros_like(det))
    return x_0, x_1


def gradient_loss_image_based(prediction, target, mask):
    """:90-108 with reduction_image_based (:76-86)."""
    m = mask.float()
    M = m.sum((1, 2))
    diff = m * (prediction - target)
    grad_x = torch.abs(diff[:, :, 1:] - diff[:, :, :-1]) * (m[:, :, 1:] * m[:, :, :-1])
    grad_y = torch.abs(diff[:, 1:, :] - diff[:, :-1, :]) * (m[:, 1:, :] * m[:, :-1, :])
    image_loss = grad_x.sum((1, 2)) + grad_y.sum((1, 2))
    image_loss = torch.where(M != 0, image_loss / torch.where(M != 0, M, torch.ones_like(M)), image_loss)
    return image_loss.mean()


def erode_mask(mask, max_pool_size=4):
    """MidasLoss.erode_mask (:153-162): valid iff the whole 4x4 block is valid."""
    h, w = mask.shape[2], mask.shape[3]
    m = 1 - mask.float()
    m = torch.nn.functional.max_pool2d(m, kernel_size=max_pool_size)
    m = torch.nn.functional.interpolate(m, (h, w), mode="nearest")
    return m == 0


def midas_loss(prediction_raw, target_raw, mask_raw, alpha=0.1, scales=4, inverse_depth=True, shrink_mask=False):
    """MidasLoss.forward (:164-185); inputs [B,1,H,W]."""
    valid = erode_mask(mask_raw) if shrink_mask else mask_raw > 0.5
    total = ssi_mae(prediction_raw, target_raw, valid)
    if alpha <= 0:
        return total
    if inverse_depth:
        prediction, target = 1 / (prediction_raw.squeeze(1) + 1e-6), 1 / (target_raw.squeeze(1) + 1e-6)
    else:
        prediction, target = prediction_raw.squeeze(1), target_raw.squeeze(1)
    m = valid.squeeze(1)
    scale, shift = scale_and_shift(prediction, target, m)
    ssi = scale.view(-1, 1, 1) * prediction + shift.view(-1, 1, 1)
    reg = 0
    for k in range(scales):
        st = 2 ** k
        reg = reg + gradient_loss_image_based(ssi[:, ::st, ::st], target[:, ::st, ::st], m[:, ::st, ::st])
    return total + alpha * reg


def intr_loss(seen_pred, seen_gt, mask):
    """utils/loss.py:36-43: seen_* [B,HW,3], mask [B,HW]."""
    distance = torch.sum((seen_pred - seen_gt) ** 2, dim=-1)
    return (distance * mask).sum() / (mask.sum() + 1.e-8)
