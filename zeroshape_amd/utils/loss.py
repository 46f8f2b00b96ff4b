"""Mirror of the reference's utils/loss.py::Loss on the HIP training kernels.

  shape_loss(pred_occ_raw [B,N], gt_sdf [B,N])            utils/loss.py:18-28   zs_bce_logits(+_bwd)
  depth_loss(pred_depth, gt_depth, mask [B,1,H,W])        utils/loss.py:30-34   zs_midas_loss(+_bwd)
      = MidasLoss(alpha=grad_reg, inverse_depth=depth_inv, shrink_mask=mask_shrink) of
        model/depth/midas_loss.py (4 scales, image-based reduction)
  intr_loss(seen_pred, seen_gt [B,HW,3], mask [B,HW])     utils/loss.py:36-43   zs_intr_loss(+_bwd)

The shape recipe (options/shape.yaml:84-87) trains with the shape loss only; options/depth.yaml
with depth + 10 x intr.  training.depth_loss.mask_shrink (min-pooled masks, false in both yaml files):
zs_erode_mask in front of the loss."""
from copy import deepcopy

import torch.nn as nn

from ..nn import autograd as A


class Loss(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = deepcopy(opt)

    def shape_loss(self, pred_occ_raw, gt_sdf):
        assert len(pred_occ_raw.shape) == 2
        assert len(gt_sdf.shape) == 2
        sl = self.opt.training.shape_loss
        return A.bce_logits(pred_occ_raw, gt_sdf, float(sl.impt_thres), float(sl.impt_weight))

    def depth_loss(self, pred_depth, gt_depth, mask):
        assert len(pred_depth.shape) == len(gt_depth.shape) == len(mask.shape) == 4
        assert pred_depth.shape[1] == gt_depth.shape[1] == mask.shape[1] == 1
        dl = self.opt.training.depth_loss
        if dl.mask_shrink:                      # midas_loss.py:153-166: min-pooled (4 x 4) validity
            mask = A.erode_mask(mask)
        return A.midas_loss(pred_depth, gt_depth, mask, alpha=float(dl.grad_reg), scales=4, inverse_depth=bool(dl.depth_inv))

    def intr_loss(self, seen_pred, seen_gt, mask):
        assert len(seen_pred.shape) == len(seen_gt.shape) == 3
        assert len(mask.shape) == 2
        return A.intr_loss(seen_pred, seen_gt, mask)
