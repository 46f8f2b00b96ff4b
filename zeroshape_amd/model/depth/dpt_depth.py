"""Mirror of the reference's model/depth/dpt_depth.py::DPTDepthModel (backbone
"vitb_rn50_384", readout "project", the only configuration graph_shape.py:31 builds) on the HIP
encoder layers.

Same state-dict names as the reference module tree (``pretrained.model.*`` = timm's
vit_base_resnet50_384, ``pretrained.act_postprocess{3,4}.*``, ``scratch.*``), so
``load_state_dict(get_child_state_dict(ckpt["graph"], "dpt_depth"))`` (graph_shape.py:75) and the
omnidata checkpoint (:87) load by name.  Forward (dpt_depth.py:68-94,115-122, vit.py:57-154):

  x = 2*img - 1 -> ResNetV2 stem/stages (taps: stage0 256@H/4, stage1 512@H/8) -> 1x1 proj ->
  [cls | tokens] + resized pos_embed -> 12 ViT blocks (taps after blocks 8 and 11) ->
  readout-project + 1x1 (+ 3x3/s2 for tap 4) -> scratch.layerK_rn 3x3 -> refinenet4..1 ->
  head (3x3, x2 bilinear, 3x3 + ReLU, 1x1 + ReLU) -> clamp [0,1]

In .train() mode under autograd the same network runs through nn/train_blocks.py (gradients to
every parameter, incl. the StdConv weight standardisation and the re-sampled position embedding).
There is no PyTorch fallback.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ...nn import autograd as A
from ...nn import blocks, ops, pack, train_blocks
from ...nn import branch
from ...nn.branch import Branch
from ...nn.module import HipModule


# ---- parameter containers (names as timm 0.6.12 / the reference register them) ----
def _gn(c):
    return nn.GroupNorm(32, c)


class _ConvNorm(nn.Module):
    def __init__(self, cin, cout, k, stride=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, bias=False)
        self.norm = _gn(cout)


class _BottleneckV2(nn.Module):
    def __init__(self, cin, cout, stride, proj):
        super().__init__()
        mid = cout // 4
        if proj:
            self.downsample = _ConvNorm(cin, cout, 1, stride)
        self.conv1 = nn.Conv2d(cin, mid, 1, bias=False)
        self.norm1 = _gn(mid)
        self.conv2 = nn.Conv2d(mid, mid, 3, stride=stride, bias=False)
        self.norm2 = _gn(mid)
        self.conv3 = nn.Conv2d(mid, cout, 1, bias=False)
        self.norm3 = _gn(cout)


class _Stage(nn.Module):
    def __init__(self, cin, cout, stride, depth):
        super().__init__()
        self.blocks = nn.Sequential(*[_BottleneckV2(cin if i == 0 else cout, cout, stride if i == 0 else 1, i == 0)
                                      for i in range(depth)])


class _ResNetV2(nn.Module):
    def __init__(self):
        super().__init__()
        self.stem = _ConvNorm(3, 64, 7, 2)
        self.stages = nn.Sequential(_Stage(64, 256, 1, 3), _Stage(256, 512, 2, 4), _Stage(512, 1024, 2, 9))


class _HybridEmbed(nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone = _ResNetV2()
        self.proj = nn.Conv2d(1024, 768, 1)


class _Attention(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class ViTBlock(nn.Module):
    """timm Block parameter names (norm1, attn.qkv, attn.proj, norm2, mlp.fc1, mlp.fc2)."""

    def __init__(self, dim, mlp_ratio=4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))


class _HybridViT(nn.Module):
    def __init__(self, img_size=384, dim=768, depth=12, num_classes=1000):
        super().__init__()
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.randn(1, (img_size // 16) ** 2 + 1, dim) * 0.02)
        self.patch_embed = _HybridEmbed()
        self.blocks = nn.Sequential(*[ViTBlock(dim) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.head = nn.Linear(dim, num_classes)          # unused by DPT, present in the checkpoints


class _ProjectReadout(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.project = nn.Sequential(nn.Linear(2 * dim, dim), nn.GELU())


class _RCU(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv1 = nn.Conv2d(c, c, 3, padding=1)
        self.conv2 = nn.Conv2d(c, c, 3, padding=1)


class _Fusion(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.out_conv = nn.Conv2d(c, c, 1)
        self.resConfUnit1 = _RCU(c)
        self.resConfUnit2 = _RCU(c)


class DPTDepthModel(HipModule):
    def __init__(self, path=None, non_negative=True, num_channels=1, backbone="vitb_rn50_384", features=256,
                 readout="project", **kwargs):
        super().__init__()
        if backbone != "vitb_rn50_384" or readout != "project" or not non_negative or num_channels != 1 or kwargs:
            raise NotImplementedError("the HIP DPT is specialised for DPTDepthModel(backbone='vitb_rn50_384') "
                                      "(model/compute_graph/graph_shape.py:31)")
        self.features = features
        self.pretrained = nn.Module()
        self.pretrained.model = _HybridViT()
        ident = lambda: nn.Sequential(nn.Identity(), nn.Identity(), nn.Identity())      # noqa: E731
        self.pretrained.act_postprocess1, self.pretrained.act_postprocess2 = ident(), ident()
        self.pretrained.act_postprocess3 = nn.Sequential(_ProjectReadout(768), nn.Identity(), nn.Identity(),
                                                         nn.Conv2d(768, 768, 1))
        self.pretrained.act_postprocess4 = nn.Sequential(_ProjectReadout(768), nn.Identity(), nn.Identity(),
                                                         nn.Conv2d(768, 768, 1),
                                                         nn.Conv2d(768, 768, 3, stride=2, padding=1))
        self.scratch = nn.Module()
        for i, c in enumerate((256, 512, 768, 768), 1):
            setattr(self.scratch, "layer%d_rn" % i, nn.Conv2d(c, features, 3, padding=1, bias=False))
        for i in (1, 2, 3, 4):
            setattr(self.scratch, "refinenet%d" % i, _Fusion(features))
        self.scratch.output_conv = nn.Sequential(
            nn.Conv2d(features, features // 2, 3, padding=1), nn.Identity(),
            nn.Conv2d(features // 2, 32, 3, padding=1), nn.ReLU(True),
            nn.Conv2d(32, 1, 1), nn.ReLU(True), nn.Identity())
        nn.init.constant_(self.scratch.output_conv[-3].bias, 0.05)        # dpt_depth.py:108
        if path is not None:
            self.load(path)
        self.eval()

    def load(self, path):
        """model/depth/base_model.py:6-17."""
        parameters = torch.load(path, map_location=torch.device("cpu"))
        if "optimizer" in parameters:
            parameters = parameters["model"]
        self.load_state_dict(parameters)

    # ---- packing ----
    def _pack(self, sd, device):
        lin = lambda p: pack.pack_conv(sd[p + ".weight"], sd[p + ".bias"]).to(device)      # noqa: E731
        pk = dict(backbone=blocks.pack_resnetv2(sd, "pretrained.model.patch_embed.backbone.", device),
                  proj=lin("pretrained.model.patch_embed.proj"),
                  cls=sd["pretrained.model.cls_token"].reshape(-1).float().contiguous().to(device),
                  pos_native=sd["pretrained.model.pos_embed"].float(), pos={},
                  blocks=[blocks.pack_vit_block(sd, "pretrained.model.blocks.%d" % i, device) for i in range(12)],
                  ro3=lin("pretrained.act_postprocess3.0.project.0"), pp3=lin("pretrained.act_postprocess3.3"),
                  ro4=lin("pretrained.act_postprocess4.0.project.0"), pp4=lin("pretrained.act_postprocess4.3"),
                  pp4s=pack.pack_conv(sd["pretrained.act_postprocess4.4.weight"], sd["pretrained.act_postprocess4.4.bias"],
                                      stride=2, padding=1).to(device),
                  rn=[pack.pack_conv(sd["scratch.layer%d_rn.weight" % i], None, padding=1).to(device)
                      for i in (1, 2, 3, 4)],
                  fusion=[blocks.pack_fusion(sd, "scratch.refinenet%d" % i, device) for i in (1, 2, 3, 4)],
                  head0=pack.pack_conv(sd["scratch.output_conv.0.weight"], sd["scratch.output_conv.0.bias"],
                                       padding=1).to(device),
                  head2=pack.pack_conv(sd["scratch.output_conv.2.weight"], sd["scratch.output_conv.2.bias"],
                                       padding=1).to(device),
                  head4=lin("scratch.output_conv.4"), device=device)
        return pk

    @staticmethod
    def _pos_embed(pk, gh, gw):
        """vit.py:103-120 (_resize_pos_embed): bilinear, align_corners=False, from the native grid.
        The reference recomputes it on every forward; here once per (checkpoint, grid) on the host."""
        if (gh, gw) not in pk["pos"]:
            pos = pk["pos_native"]
            g0 = int(math.sqrt(pos.shape[1] - 1))
            grid = pos[0, 1:].reshape(1, g0, g0, -1).permute(0, 3, 1, 2)
            grid = F.interpolate(grid, size=(gh, gw), mode="bilinear", align_corners=False)
            pos = torch.cat([pos[0, :1], grid.permute(0, 2, 3, 1).reshape(gh * gw, -1)], 0)
            pk["pos"][(gh, gw)] = pos.contiguous().to(pk["device"])
        return pk["pos"][(gh, gw)]

    # ---- forward ----
    def forward(self, image, get_feat=False, taps=None, on_feat=None):
        """image [B,3,H,W] in [0,1] -> depth [B,1,H,W] in [0,1] (and the tap-4 feature
        [B,768,H/32,W/32] with get_feat), dpt_depth.py:115-122.  `taps` (a dict, tests only)
        receives channels-last intermediates; `on_feat` (inference only) is called with the channels-last
        tap-4 feature as soon as it is queued."""
        self._need_gpu(image, "image")
        B, C, H, W = image.shape
        if C != 3 or H % 32 or W % 32:
            raise ValueError("image must be [B,3,H,W] with H, W multiples of 32, got %s" % (tuple(image.shape),))
        if self.training and torch.is_grad_enabled():
            depth, layer_4 = self.forward_train(image, taps)
            return (depth, A.to_nchw(layer_4)) if get_feat else depth
        with torch.no_grad():
            return self._forward_eval(image, get_feat, taps, on_feat)

    def forward_train(self, image, taps=None):
        """Autograd path: -> (depth [B,1,H,W], layer_4 channels-last [B,H/32,W/32,768])."""
        B, _, H, W = image.shape
        gh, gw = H // 16, W // 16
        vit, pre, sc = self.pretrained.model, self.pretrained, self.scratch
        record = (lambda **kw: taps.update(kw)) if taps is not None else (lambda **kw: None)
        x = A.to_nhwc(image.float(), cpad=4)
        s0, s1, s2 = train_blocks.resnetv2(x, vit.patch_embed.backbone, in_scale=2.0, in_shift=-1.0)
        record(stage0=s0, stage1=s1, stage2=s2)
        s0c, s1c = A.cut(s0, s1)                     # (segmented backward, nn/autograd.py: read by the decoder, two cuts on)
        feat = A.conv2d(s2, vit.patch_embed.proj.weight, vit.patch_embed.proj.bias).view(B, gh * gw, 768)
        # vit.py:103-120: the native position grid re-sampled to (gh, gw) on every forward
        pos = vit.pos_embed[0]
        g0 = int(math.sqrt(pos.shape[0] - 1))
        grid = A.resize_grid(pos[1:].reshape(g0, g0, -1), gh, gw).reshape(gh * gw, -1)
        tok = A.assemble_tokens(feat, vit.cls_token.reshape(-1), torch.cat([pos[:1], grid], 0))
        hooked = {}
        for i, blk in enumerate(vit.blocks):
            tok = train_blocks.vit_block(tok, blk, 12)
            if i in (0, 8, 11):
                hooked[i] = tok
            if i in (2, 5):                          # cuts: stem + blocks 0-2 | blocks 3-5 | blocks 6-11
                tok = A.cut(tok)
                A.segment_break()
        record(block0=hooked[0], block8=hooked[8], block11=hooked[11])
        t8, t11 = A.cut(hooked[8], hooked[11])       # cut: ViT | reassemble + fusion decoder + head
        A.segment_break()

        def reassemble(t, post):
            lin = post[0].project[0]
            r = A.gelu(A.linear(A.readout_concat(t), lin.weight, lin.bias))
            return A.conv2d(r.view(B, gh, gw, 768), post[3].weight, post[3].bias)
        layer_3 = reassemble(t8, pre.act_postprocess3)
        p4 = pre.act_postprocess4
        layer_4 = A.conv2d(reassemble(t11, p4), p4[4].weight, p4[4].bias, stride=2, padding=1)
        rn = [A.conv2d(l, getattr(sc, "layer%d_rn" % i).weight, None, padding=1)
              for i, l in enumerate((s0c, s1c, layer_3, layer_4), 1)]
        record(layer3_rn=rn[2], layer4_rn=rn[3])
        path4 = train_blocks.fusion(rn[3], sc.refinenet4)
        path3 = train_blocks.fusion(path4, sc.refinenet3, rn[2])
        path2 = train_blocks.fusion(path3, sc.refinenet2, rn[1])
        path = train_blocks.fusion(path2, sc.refinenet1, rn[0])
        record(path4=path4, path3=path3, path2=path2, path1=path)
        oc = sc.output_conv
        o = A.upsample2x(A.conv2d(path, oc[0].weight, oc[0].bias, padding=1))
        o = A.conv2d(o, oc[2].weight, oc[2].bias, padding=1, act=A.ACT_RELU)
        o = A.conv2d(o, oc[4].weight, oc[4].bias, act=A.ACT_RELU_CLAMP1)        # [B,H,W,1]
        return o.view(B, 1, H, W), layer_4

    def _forward_eval(self, image, get_feat=False, taps=None, on_feat=None):
        B, C, H, W = image.shape
        pk = self.packed(image.device)
        gh, gw = H // 16, W // 16
        x = ops.to_nhwc(image, cpad=4)
        record = (lambda **kw: taps.update(kw)) if taps is not None else (lambda **kw: None)
        rn, heads, brs = [None] * 4, [None] * 4, [None] * 4

        def skip_branch(i, src, fn):
            """layerK_rn (+ the first convolution of refinenetK's skip unit) of tap i on a side branch: they depend on the
            tap only and are needed much later (nn/branch.py)."""
            brs[i] = Branch(src, kind=branch.SKIP)
            with brs[i]:
                rn[i] = ops.conv2d(fn(src), pk["rn"][i])
                if i < 3:                                        # refinenet4 has no skip unit
                    heads[i] = blocks.run_rcu_head(rn[i], pk["fusion"][i]["r1"])

        def on_stage(si, feat):
            if si < 2:
                skip_branch(si, feat, lambda t: t)
        s0, s1, s2 = blocks.run_resnetv2(x, pk["backbone"], in_scale=2.0, in_shift=-1.0, on_stage=on_stage)
        record(stage0=s0, stage1=s1, stage2=s2)
        feat = ops.conv2d(s2, pk["proj"]).view(B, gh * gw, 768)
        tok = ops.assemble_tokens(feat, pk["cls"], self._pos_embed(pk, gh, gw))

        def reassemble(t, ro, pp):
            r = ops.linear(ops.readout_concat(t), ro, act=ops.ACT_GELU)          # [B, gh*gw, 768]
            return ops.conv2d(r.view(B, gh, gw, 768), pp)
        hooked, st = {}, None
        for i, blk in enumerate(pk["blocks"]):
            tok, st = blocks.run_vit_block(tok, blk, 12, stats=st, want_stats=True)
            if i in (0, 8, 11):
                hooked[i] = tok
            if i == 8:                                           # tap 3's reassemble branch beside ViT blocks 9-11
                skip_branch(2, tok, lambda t: reassemble(t, pk["ro3"], pk["pp3"]))
        record(block0=hooked[0], block8=hooked[8], block11=hooked[11])
        layer_4 = ops.conv2d(reassemble(hooked[11], pk["ro4"], pk["pp4"]), pk["pp4s"])
        if on_feat is not None:            # eval only: the caller forks work on the channels-last tap-4 feature (graph_shape.encode)
            on_feat(layer_4)
        rn[3] = ops.conv2d(layer_4, pk["rn"][3])
        path4 = blocks.run_fusion(rn[3], pk["fusion"][3])
        rn[2], heads[2] = brs[2].join(rn[2], heads[2])
        record(layer3_rn=rn[2], layer4_rn=rn[3])
        path3 = blocks.run_fusion(path4, pk["fusion"][2], rn[2], heads[2])
        rn[1], heads[1] = brs[1].join(rn[1], heads[1])
        path2 = blocks.run_fusion(path3, pk["fusion"][1], rn[1], heads[1])
        rn[0], heads[0] = brs[0].join(rn[0], heads[0])
        path = blocks.run_fusion(path2, pk["fusion"][0], rn[0], heads[0])
        record(path4=path4, path3=path3, path2=path2, path1=path)
        o = ops.conv2d_tail(ops.conv2d(path, pk["head0"]), pk["head2"], pk["head4"], act=ops.ACT_RELU,
                            tail_act=ops.ACT_RELU_CLAMP1, upsample=True)                                   # [B,H,W,1]
        depth = o.view(B, 1, H, W)                                               # C == 1: same memory order
        if get_feat:
            return depth, ops.to_nchw(layer_4)
        return depth
