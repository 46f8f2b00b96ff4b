"""Mirror of the reference's utils/loss.py::Loss on the HIP training kernels.

  shape_loss(pred_occ_raw [B,N], gt_sdf [B,N])   utils/loss.py:18-28   zs_bce_logits(+_bwd)

The default recipe (options/shape.yaml:84-87) trains with the shape loss only; depth_loss
(MiDaS scale-and-shift invariant loss, model/depth/midas_loss.py) and intr_loss
(utils/loss.py:36-43) are used by options/depth.yaml and raise here until their kernels exist -
never silently approximated."""
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
        raise NotImplementedError("depth_loss (MiDaS loss, loss_weight.depth) is not on the HIP path yet")

    def intr_loss(self, seen_pred, seen_gt, mask):
        raise NotImplementedError("intr_loss (loss_weight.intr) is not on the HIP path yet")
