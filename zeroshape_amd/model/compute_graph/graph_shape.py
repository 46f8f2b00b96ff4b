"""Mirror of the reference's model/compute_graph/graph_shape.py::Graph, inference branch
(:115-150): depth + intrinsics prediction, seen-surface geometry, coordinate encoder, and the
implicit decoder as ``impl_network`` - every stage on the HIP library.

    graph = Graph(opt).cuda().eval(); graph.load_state_dict(ckpt["graph"])
    var = graph.forward(opt, var, training=False, get_loss=False)
    # adds var.depth_pred, var.intr_pred, var.validity_mask, var.seen_points, var.latent_depth

State-dict names equal the reference's (dpt_depth.*, intr_head.*, intr_proj.*,
coord_encoder.*, impl_network.*).

forward(opt, var, training=True, get_loss=True) is the training branch (:115-204): the same
network through the autograd bindings of zeroshape_amd/nn/autograd.py (every forward and backward
op a HIP kernel), the ground-truth branch (:152-181) and Loss.shape_loss.  Gradients reach every
parameter the reference trains: decoder, coordinate encoder and - unless optim.fix_dpt - the depth
model and intrinsics head, through the seen-surface geometry (zs_seen_surface_bwd).
"""
import torch
import torch.nn as nn

from ...nn import autograd as A
from ...nn import blocks, ops, pack, train_blocks
from ...nn import branch
from ...nn.branch import Branch
from ...nn.module import HipModule
from ...utils import camera
from ...utils.loss import Loss
from ...utils.options import EasyDict as edict
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
        return (A.GENERATION[0],) + tuple((t.data_ptr(), t._version) for t in ts)

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
        return self.run_nhwc(ops.to_nhwc(feat_nchw))

    def run_nhwc(self, x):
        """x [B,h,w,768] channels-last (DPT's tap-4 feature as the layers hold it)."""
        self.training = self._graph.intr_head.training        # not a submodule: follow the owner's mode
        pk = self.packed(x.device)
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
        if opt.get("optim") and opt.optim.get("fix_dpt"):                          # :33-36
            for m in (self.dpt_depth, self.intr_head, self.intr_proj):
                for p in m.parameters():
                    p.requires_grad_(False)
        self.loss_fns = Loss(opt)
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
        """Replay the encoder as one captured hipGraph per input shape instead of ~350 launches.
        A capture bakes in the packed weights of its moment: the key carries the weights' version
        (in-place updates, re-assignment, the fused optimiser's generation counter), so a
        load_state_dict / optimiser step / resume re-captures instead of replaying stale weights."""
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
            # the intrinsics head needs the tap-4 feature only: it runs beside DPT's fusion blocks (nn/branch.py)
            intr = {}

            def intr_branch(layer_4):
                intr["br"] = Branch(layer_4, kind=branch.INTR)
                with intr["br"]:
                    intr["pred"] = self.intr_param2mtx(opt, self._intr.run_nhwc(layer_4))
            depth_pred = self.dpt_depth(rgb, on_feat=intr_branch)
            intr_pred = intr["br"].join(intr["pred"])
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
        from ...nn import autograd as A
        wkey = (A.GENERATION[0],) + tuple((t.data_ptr(), t._version) for m in (self.dpt_depth, self.intr_head,
                                                                                self.intr_proj, self.coord_encoder)
                                          for t in list(m.parameters()) + list(m.buffers()))
        key = (tuple(rgb.shape), str(rgb.device), dsp, resnet)
        hit = self._captured.get(key)
        if hit is None or hit[0] != wkey:
            self._captured[key] = hit = (wkey, CapturedCall(run, [rgb, mask]))
        return hit[1](rgb, mask)

    def forward(self, opt, var, training=False, get_loss=True):
        """graph_shape.py:115-192.  training / get_loss / GT samples select the autograd branch."""
        batch_size = len(var.idx)
        var.latent_semantic = None
        HipModule._need_gpu(var.rgb_input_map, "var.rgb_input_map")
        with_samples = 'gt_sample_points' in var and 'gt_sample_sdf' in var
        if not (training or get_loss or with_samples) or not torch.is_grad_enabled():
            with torch.no_grad():
                var.depth_pred, var.intr_pred, var.seen_points, var.latent_depth = self.encode(
                    opt, var.rgb_input_map, var.mask_input_map)
                var.validity_mask = (var.mask_input_map > 0.5).float().view(batch_size, -1)
                var.pose = var.pose_gt if 'pose_gt' in var else None
                if with_samples:
                    self._gt_branch(opt, var)
                    var.pred_sample_occ, _ = self.impl_network(var.latent_depth, None, var.gt_points_cam,
                                                               need_attn=False)
                if get_loss:
                    return var, self.compute_loss(opt, var, training)
            return var
        # ---- autograd branch (:117-192 under graph.train()) ----
        with A.deferred_bn_counters():          # the BatchNorm step counters: one launch for all 66 layers
            H, W = opt.H, opt.W
            rgb = var.rgb_input_map.detach().float().contiguous()
            mask = var.mask_input_map.detach().float().contiguous()
            if self.dpt_depth.training:
                var.depth_pred, layer_4 = self.dpt_depth.forward_train(rgb)
            else:                                             # a frozen, eval-mode depth model
                with torch.no_grad():
                    var.depth_pred, feat = self.dpt_depth(rgb, get_feat=True)
                    layer_4 = ops.to_nhwc(feat)
            if self.intr_head.training:
                x = train_blocks.bottleneck_conv(train_blocks.bottleneck_conv(layer_4, self.intr_head[0]), self.intr_head[1])
                intr_params = A.linear(A.global_mean(x), self.intr_proj.weight, self.intr_proj.bias)      # :125-127
            else:
                with torch.no_grad():
                    intr_params = self._intr.run(ops.to_nchw(layer_4))
            var.intr_pred = A.intr_param2mtx(intr_params, H, W)                                            # :129
            var.validity_mask = (mask > 0.5).float().view(batch_size, -1)
            if opt.arch.depth.encoder == 'resnet':
                assert opt.arch.depth.dsp == 1
                var.seen_points, seen_3D_dsp, mask_dsp = A.seen_surface(var.depth_pred, var.intr_pred, mask)   # :131-144
                # (segmented backward, nn/autograd.py: depth + intrinsics model | coordinate encoder | decoder + losses)
                var.depth_pred, var.seen_points, seen_3D_dsp = A.cut(var.depth_pred, var.seen_points, seen_3D_dsp)
                A.segment_break()
                var.latent_depth = self.coord_encoder(seen_3D_dsp, mask_dsp)                               # :147-150
            else:                                              # transformer coordinate encoder
                assert opt.arch.depth.dsp == 2, "the transformer coordinate encoder trains with arch.depth.dsp = 2 (options/shape.yaml:28)"
                var.seen_points, seen_3D_dsp, mask_dsp = A.seen_surface_dsp2(var.depth_pred, var.intr_pred, mask)
                var.depth_pred, var.seen_points, seen_3D_dsp = A.cut(var.depth_pred, var.seen_points, seen_3D_dsp)
                A.segment_break()
                var.latent_depth = self.coord_encoder(seen_3D_dsp.permute(0, 2, 3, 1).contiguous(),
                                                      mask_dsp.squeeze(1) > 0.5)
            var.latent_depth = A.cut(var.latent_depth)
            A.segment_break()
            var.pose = var.pose_gt if 'pose_gt' in var else None
            if with_samples:
                with torch.no_grad():
                    self._gt_branch(opt, var)
                var.pred_sample_occ, _ = self.impl_network(var.latent_depth, None, var.gt_points_cam,     # :185
                                                           need_attn=False)
            if get_loss:
                return var, self.compute_loss(opt, var, training)
            return var

    def _gt_branch(self, opt, var):
        """:152-181 (no gradient): normalisation factors of the GT seen surface, GT query points
        moved to that frame, near-surface points for the visualiser."""
        batch_size = len(var.idx)
        seen_gt, _, _, mean_gt, scale_gt = camera.seen_surface(opt, var.depth_input_map, var.intr,
                                                               var.mask_input_map, dsp=1)
        var.seen_points_gt = seen_gt
        var.gt_points_cam = camera.transform_points(var.gt_sample_points, var.pose_gt, mean_gt, scale_gt)
        close_surf_idx = torch.topk(var.gt_sample_sdf.abs(), k=min(100, var.gt_sample_sdf.shape[1]), dim=1,
                                    largest=False)[1].unsqueeze(-1).repeat(1, 1, 3)
        var.gt_surf_points = torch.gather(var.gt_points_cam, dim=1, index=close_surf_idx)
        assert var.gt_points_cam.shape[0] == batch_size

    def compute_loss(self, opt, var, training=False):
        """:194-204."""
        loss = edict()
        if opt.loss_weight.depth is not None:
            loss.depth = self.loss_fns.depth_loss(var.depth_pred, var.depth_input_map, var.mask_input_map)
        if opt.loss_weight.intr is not None and training:
            loss.intr = self.loss_fns.intr_loss(var.seen_points, var.seen_points_gt, var.validity_mask)
        if opt.loss_weight.shape is not None and training:
            loss.shape = self.loss_fns.shape_loss(var.pred_sample_occ, var.gt_sample_sdf)
        return loss
