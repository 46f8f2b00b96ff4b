"""Mirror of the reference's model/compute_graph/graph_depth.py::Graph (depth + intrinsics task,
options/depth.yaml): DPTDepthModel, the optional intrinsics head, the normalised seen surfaces of
the prediction and of the ground truth (:62-97).  Inference runs on the HIP encoder layers and the
fused seen-surface kernel; the training losses of this task (MiDaS scale-and-shift invariant depth
loss, intrinsics loss) are not on the HIP path and raise (zeroshape_amd/utils/loss.py)."""
import torch
import torch.nn as nn

from ...nn import autograd as A
from ...nn import ops, train_blocks
from ...nn.module import HipModule
from ...utils import camera
from ...utils.layers import Bottleneck_Conv
from ...utils.loss import Loss
from ...utils.options import EasyDict as edict
from ..depth.dpt_depth import DPTDepthModel
from .graph_shape import _IntrHead


class Graph(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.dpt_depth = DPTDepthModel(backbone='vitb_rn50_384')
        import os
        pre = opt.arch.depth.pretrained
        if pre is not None and os.path.exists(pre):                           # :16-19 (no network: skip if absent)
            self.dpt_depth.load_state_dict(torch.load(pre, map_location="cpu")['model_state_dict'])
        self.with_intr = opt.loss_weight.intr is not None
        if self.with_intr:                                                    # :21-31
            self.intr_feat_channels = 768
            self.intr_head = nn.Sequential(Bottleneck_Conv(self.intr_feat_channels, kernel_size=3),
                                           Bottleneck_Conv(self.intr_feat_channels, kernel_size=3))
            self.intr_pool = nn.AdaptiveAvgPool2d((1, 1))
            self.intr_proj = nn.Linear(self.intr_feat_channels, 3)
            nn.init.zeros_(self.intr_proj.weight)
            nn.init.zeros_(self.intr_proj.bias)
            self._intr = _IntrHead(self)
        self.loss_fns = Loss(opt)
        self.eval()

    def __setattr__(self, name, value):
        if name == "_intr":
            object.__setattr__(self, name, value)
        else:
            super().__setattr__(name, value)

    def intr_param2mtx(self, opt, intr_params):
        return camera.intr_param2mtx(opt, intr_params)

    def forward(self, opt, var, training=False, get_loss=True):
        """:62-97."""
        HipModule._need_gpu(var.rgb_input_map, "var.rgb_input_map")
        batch_size = len(var.idx)
        rgb = var.rgb_input_map.detach().float().contiguous()
        mask = var.mask_input_map.detach().float().contiguous()
        autograd = torch.is_grad_enabled() and self.dpt_depth.training
        if not autograd:
            with torch.no_grad():
                if not self.with_intr:
                    var.depth_pred = self.dpt_depth(rgb)
                else:
                    var.depth_pred, intr_feat = self.dpt_depth(rgb, get_feat=True)
                    var.intr_pred = self.intr_param2mtx(opt, self._intr.run(intr_feat))
                    var.seen_points_pred = camera.seen_surface(opt, var.depth_pred, var.intr_pred, mask, dsp=1)[0]
        else:
            var.depth_pred, layer_4 = self.dpt_depth.forward_train(rgb)
            if self.with_intr:
                x = train_blocks.bottleneck_conv(train_blocks.bottleneck_conv(layer_4, self.intr_head[0]), self.intr_head[1])
                intr_params = A.linear(A.global_mean(x), self.intr_proj.weight, self.intr_proj.bias)
                var.intr_pred = A.intr_param2mtx(intr_params, opt.H, opt.W)
                var.seen_points_pred, _, _ = A.seen_surface(var.depth_pred, var.intr_pred, mask)
        if self.with_intr and ('depth_input_map' in var or training):
            with torch.no_grad():                                                  # :84-93
                var.seen_points_gt = camera.seen_surface(opt, var.depth_input_map, var.intr, mask, dsp=1)[0]
                var.validity_mask = (mask > 0.5).float().view(batch_size, -1)
        if get_loss:
            return var, self.compute_loss(opt, var, training)
        return var

    def compute_loss(self, opt, var, training=False):
        loss = edict()
        if opt.loss_weight.depth is not None:
            loss.depth = self.loss_fns.depth_loss(var.depth_pred, var.depth_input_map, var.mask_input_map)
        if opt.loss_weight.intr is not None:
            loss.intr = self.loss_fns.intr_loss(var.seen_points_pred, var.seen_points_gt, var.validity_mask)
        return loss
