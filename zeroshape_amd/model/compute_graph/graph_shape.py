"""Mirror of the reference's model/compute_graph/graph_shape.py::Graph, inference branch
(:115-150): depth + intrinsics prediction, seen-surface geometry, coordinate encoder, and the
implicit decoder as ``impl_network`` - every stage on the HIP library.

    graph = Graph(opt).cuda().eval(); graph.load_state_dict(ckpt["graph"])
    var = graph.forward(opt, var, training=False, get_loss=False)
    # adds var.depth_pred, var.intr_pred, var.validity_mask, var.seen_points, var.latent_depth

State-dict names equal the reference's (dpt_depth.*, intr_head.*, intr_proj.*,
coord_encoder.*, impl_network.*).  Training (losses, the GT branch :152-185) is not built:
asking for it raises.
"""
import torch
import torch.nn as nn

from ...nn import blocks, ops, pack
from ...nn.module import HipModule
from ...utils import camera
from ...utils.layers import Bottleneck_Conv
from ...utils.util import get_child_state_dict
from ..depth.dpt_depth import DPTDepthModel
from ..shape.implicit import Implicit
from ..shape.seen_coord_enc import CoordEncAtt, CoordEncRes


class _IntrHead(HipModule):
    """intr_head + intr_pool + intr_proj (graph_shape.py:18-28,125-127) as one packed unit that
    reads its parameters from the owning Graph."""

    def __init__(self, graph):
        super().__init__()
        object.__setattr__(self, "_graph", graph)          # not a submodule: no extra state-dict names

    def _tensors_key(self):
        g = self._graph
        ts = list(g.intr_head.parameters()) + list(g.intr_head.buffers()) + list(g.intr_proj.parameters())
        return tuple((t.data_ptr(), t._version) for t in ts)

    def state_dict(self, *a, **k):
        g = self._graph
        sd = {"intr_head." + n: v for n, v in g.intr_head.state_dict().items()}
        sd.update({"intr_proj." + n: v for n, v in g.intr_proj.state_dict().items()})
        return sd

    def _pack(self, sd, device):
        return dict(b0=blocks.pack_bottleneck_conv(sd, "intr_head.0", 3, device),
                    b1=blocks.pack_bottleneck_conv(sd, "intr_head.1", 3, device),
                    proj=pack.pack_conv(sd["intr_proj.weight"], sd["intr_proj.bias"]).to(device))

    def run(self, feat_nchw):
        self.training = self._graph.intr_head.training        # not a submodule: follow the owner's mode
        pk = self.packed(feat_nchw.device)
        x = ops.to_nhwc(feat_nchw)
        x = blocks.run_bottleneck_conv(blocks.run_bottleneck_conv(x, pk["b0"]), pk["b1"])
        pooled = ops.global_mean(x)                                            # [B,768]
        return ops.linear(pooled, pk["proj"])                                  # [B,3]


class Graph(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.intr_feat_channels = 768
        self.intr_head = nn.Sequential(Bottleneck_Conv(self.intr_feat_channels, kernel_size=3),
                                       Bottleneck_Conv(self.intr_feat_channels, kernel_size=3))
        self.intr_pool = nn.AdaptiveAvgPool2d((1, 1))
        self.intr_proj = nn.Linear(self.intr_feat_channels, 3)
        nn.init.zeros_(self.intr_proj.weight)                                  # :26-28
        nn.init.zeros_(self.intr_proj.bias)
        self.dpt_depth = DPTDepthModel(backbone='vitb_rn50_384')
        self.load_pretrained_depth(opt)
        if opt.arch.depth.encoder == 'resnet':
            opt.arch.depth.dsp = 1                                             # :41-43
            self.coord_encoder = CoordEncRes(opt)
        else:
            self.coord_encoder = CoordEncAtt(embed_dim=opt.arch.latent_dim, n_blocks=opt.arch.depth.n_blocks,
                                             num_heads=opt.arch.num_heads,
                                             win_size=opt.arch.win_size // opt.arch.depth.dsp)
        if opt.arch.rgb.encoder:
            raise NotImplementedError("the RGB branch (graph_shape.py:49-55) is not used by the released model "
                                      "and is not built")
        self.rgb_encoder = None
        feat_res = opt.H // opt.arch.win_size
        self.impl_network = Implicit(feat_res ** 2, latent_dim=opt.arch.latent_dim, semantic=False,
                                     n_channels=opt.arch.impl.n_channels, n_blocks_attn=opt.arch.impl.att_blocks,
                                     n_layers_mlp=opt.arch.impl.mlp_layers, num_heads=opt.arch.num_heads,
                                     posenc_3D=opt.arch.impl.posenc_3D, mlp_ratio=opt.arch.impl.mlp_ratio,
                                     skip_in=opt.arch.impl.skip_in, pos_perlayer=opt.arch.impl.posenc_perlayer)
        self._intr = _IntrHead(self)
        self._captured, self._use_hip_graph = {}, False
        self.eval()

    def __setattr__(self, name, value):
        if name in ("_intr", "_captured", "_use_hip_graph"):   # helpers stay out of the module / state-dict tree
            object.__setattr__(self, name, value)
        else:
            super().__setattr__(name, value)

    def load_pretrained_depth(self, opt):
        """graph_shape.py:69-87: depth + intrinsics weights from our depth checkpoint, or the
        omnidata DPT weights.  Skipped when the option is empty or the file is absent (no network)."""
        import os
        if getattr(opt.pretrain, "depth", None) and os.path.exists(opt.pretrain.depth):
            checkpoint = torch.load(opt.pretrain.depth, map_location="cpu")
            self.dpt_depth.load_state_dict(get_child_state_dict(checkpoint["graph"], "dpt_depth"))
            self.intr_head.load_state_dict(get_child_state_dict(checkpoint["graph"], "intr_head"))
            self.intr_proj.load_state_dict(get_child_state_dict(checkpoint["graph"], "intr_proj"))
        elif getattr(opt.arch.depth, "pretrained", None) and os.path.exists(opt.arch.depth.pretrained):
            checkpoint = torch.load(opt.arch.depth.pretrained, map_location="cpu")
            self.dpt_depth.load_state_dict(checkpoint['model_state_dict'])

    def intr_param2mtx(self, opt, intr_params):
        """:89-113."""
        return camera.intr_param2mtx(opt, intr_params)

    def enable_hip_graph(self, on=True):
        """Replay the encoder as one captured hipGraph per input shape instead of ~350 launches
        (weights must not change while enabled; disabling drops the captures)."""
        self._use_hip_graph = bool(on)
        self._captured = {}
        return self

    @torch.no_grad()
    def encode(self, opt, rgb_input_map, mask_input_map):
        """The encoder half of forward (:117-150) on tensors:
        -> (depth_pred, intr_pred, seen_points, latent_depth)."""
        dsp = opt.arch.depth.dsp
        resnet = opt.arch.depth.encoder == 'resnet'

        def run(rgb, mask):
            depth_pred, intr_feat = self.dpt_depth(rgb, get_feat=True)
            intr_pred = self.intr_param2mtx(opt, self._intr.run(intr_feat))
            # :131-144 in one launch
            seen_points, seen_3D_dsp, mask_dsp, _, _ = camera.seen_surface(opt, depth_pred, intr_pred, mask, dsp=dsp)
            if resnet:
                latent = self.coord_encoder(seen_3D_dsp, mask_dsp)
            else:
                latent = self.coord_encoder(seen_3D_dsp.permute(0, 2, 3, 1).contiguous(), mask_dsp.squeeze(1) > 0.5)
            return depth_pred, intr_pred, seen_points, latent
        rgb = rgb_input_map.detach().float().contiguous()
        mask = mask_input_map.detach().float().contiguous()
        if not self._use_hip_graph:
            return run(rgb, mask)
        from ...nn.capture import CapturedCall
        key = (tuple(rgb.shape), str(rgb.device), dsp, resnet)
        if key not in self._captured:
            self._captured[key] = CapturedCall(run, [rgb, mask])
        return self._captured[key](rgb, mask)

    @torch.no_grad()
    def forward(self, opt, var, training=False, get_loss=True):
        if training or get_loss or ('gt_sample_points' in var and 'gt_sample_sdf' in var and training):
            raise NotImplementedError("Graph.forward: only the inference branch (training=False, get_loss=False) "
                                      "runs on the HIP path")
        batch_size = len(var.idx)
        var.latent_semantic = None
        HipModule._need_gpu(var.rgb_input_map, "var.rgb_input_map")
        var.depth_pred, var.intr_pred, var.seen_points, var.latent_depth = self.encode(
            opt, var.rgb_input_map, var.mask_input_map)
        var.validity_mask = (var.mask_input_map > 0.5).float().view(batch_size, -1)
        var.pose = var.pose_gt if 'pose_gt' in var else None
        return var
