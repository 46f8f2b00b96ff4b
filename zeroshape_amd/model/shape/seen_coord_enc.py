"""Mirror of the reference's model/shape/seen_coord_enc.py (CoordEncRes :141-194, the default
seen-surface encoder; CoordEmb + CoordEncAtt :13-139, the transformer alternative) on the HIP
encoder layers.  Same constructor arguments and state-dict names; inference through the packed
layers, training (both encoders) through nn/autograd.py."""
from functools import partial

import torch
import torch.nn as nn

from ...nn import autograd as A
from ...nn import blocks, ops, pack, train_blocks
from ...nn import branch
from ...nn.branch import Branch
from ...nn.module import HipModule
from ...utils.layers import Bottleneck_Conv
from ...utils.pos_embed import get_2d_sincos_pos_embed
from ..depth.dpt_depth import ViTBlock


# ---- torchvision resnet50 parameter names ----
class _BottleneckV1(nn.Module):
    def __init__(self, inplanes, planes, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        if down:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))


def _layer(inplanes, planes, blocks_, stride):
    return nn.Sequential(*[_BottleneckV1(inplanes if i == 0 else planes * 4, planes, stride if i == 0 else 1, i == 0)
                           for i in range(blocks_)])


class _ResNet50(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = _layer(64, 64, 3, 1)
        self.layer2 = _layer(256, 128, 4, 2)
        self.layer3 = _layer(512, 256, 6, 2)
        self.layer4 = _layer(1024, 512, 3, 2)
        self.fc = nn.Linear(2048, 1000)


class CoordEncRes(HipModule):
    """Seen surface encoder based on resnet (seen_coord_enc.py:141-194)."""

    def __init__(self, opt):
        super().__init__()
        self.encoder = _ResNet50()
        self.encoder.fc = nn.Sequential(Bottleneck_Conv(2048), Bottleneck_Conv(2048),
                                        nn.Linear(2048, opt.arch.latent_dim))
        assert opt.arch.depth.dsp == 1
        if opt.arch.win_size == 16:
            self.tap, c = 2, 1024                  # layer3
        elif opt.arch.win_size == 32:
            self.tap, c = 3, 2048                  # layer4
        else:
            print('Make sure win_size is 16 or 32 when using resnet backbone!')
            raise NotImplementedError
        self.depth_feat_proj = nn.Sequential(Bottleneck_Conv(c), Bottleneck_Conv(c),
                                             nn.Conv2d(c, opt.arch.latent_dim, 1))
        self.eval()

    def _pack(self, sd, device):
        return dict(trunk=blocks.pack_resnet50(sd, "encoder.", device),
                    fc0=blocks.pack_bottleneck_conv(sd, "encoder.fc.0", 1, device),
                    fc1=blocks.pack_bottleneck_conv(sd, "encoder.fc.1", 1, device),
                    fc2=pack.pack_conv(sd["encoder.fc.2.weight"], sd["encoder.fc.2.bias"]).to(device),
                    p0=blocks.pack_bottleneck_conv(sd, "depth_feat_proj.0", 1, device),
                    p1=blocks.pack_bottleneck_conv(sd, "depth_feat_proj.1", 1, device),
                    p2=pack.pack_conv(sd["depth_feat_proj.2.weight"], sd["depth_feat_proj.2.bias"]).to(device))

    def forward(self, coord_obj, mask_obj):
        """coord_obj [B,3,H,W], mask_obj [B,1,H,W] -> [B, 1 + (H/ws)*(W/ws), latent_dim], global
        token first (:180-194).  In .train() mode under autograd: BatchNorm on batch statistics and
        gradients to every parameter and to coord_obj (nn/train_blocks.py)."""
        self._need_gpu(coord_obj, "coord_obj")
        assert len(coord_obj.shape) == len(mask_obj.shape) == 4
        if self.training and torch.is_grad_enabled():
            with A.deferred_bn_counters():
                return self._forward_train(coord_obj, mask_obj)
        with torch.no_grad():
            return self._forward_eval(coord_obj, mask_obj)

    def _forward_train(self, coord_obj, mask_obj):
        B = coord_obj.shape[0]
        enc = self.encoder
        x = A.to_nhwc(coord_obj.float(), cpad=4, mask=mask_obj)                 # coord * mask (:184)
        feats = train_blocks.resnet50(x, enc)
        g = A.global_mean(feats[3]).view(B, 1, 1, -1)
        g = train_blocks.bottleneck_conv(train_blocks.bottleneck_conv(g, enc.fc[0]), enc.fc[1])
        g = A.linear(g.view(B, 1, -1), enc.fc[2].weight, enc.fc[2].bias)
        proj = self.depth_feat_proj
        loc = train_blocks.bottleneck_conv(train_blocks.bottleneck_conv(feats[self.tap], proj[0]), proj[1])
        loc = A.conv2d(loc, proj[2].weight, proj[2].bias)
        return torch.cat([g, loc.view(B, -1, loc.shape[-1])], dim=1)

    def _forward_eval(self, coord_obj, mask_obj):
        B = coord_obj.shape[0]
        pk = self.packed(coord_obj.device)
        x = ops.to_nhwc(coord_obj, cpad=4, mask=mask_obj)                       # coord * mask (:184)
        side = {}

        def on_layer(li, feat):            # depth_feat_proj needs only its tap: beside the remaining layers and the fc head
            if li == self.tap and li < 3:
                side["br"] = Branch(feat, kind=branch.PROJ)
                with side["br"]:
                    loc = blocks.run_bottleneck_conv(blocks.run_bottleneck_conv(feat, pk["p0"]), pk["p1"])
                    side["loc"] = ops.conv2d(loc, pk["p2"])
        feats = blocks.run_resnet50(x, pk["trunk"], on_layer=on_layer)
        g = ops.global_mean(feats[3]).view(B, 1, 1, -1)                        # avgpool + flatten
        g = blocks.run_bottleneck_conv(blocks.run_bottleneck_conv(g, pk["fc0"]), pk["fc1"])
        g = ops.conv2d(g, pk["fc2"]).view(B, 1, -1)
        if "br" in side:
            loc = side["br"].join(side["loc"])
        else:
            loc = blocks.run_bottleneck_conv(blocks.run_bottleneck_conv(feats[self.tap], pk["p0"]), pk["p1"])
            loc = ops.conv2d(loc, pk["p2"])
        return torch.cat([g, loc.view(B, -1, loc.shape[-1])], dim=1)


class CoordEmb(nn.Module):
    """Parameter container of seen_coord_enc.py:13-47 (run by CoordEncAtt.forward)."""

    def __init__(self, embed_dim, win_size=8, num_heads=8):
        super().__init__()
        self.embed_dim, self.win_size, self.num_heads = embed_dim, win_size, num_heads
        self.two_d_pos_embed = nn.Parameter(torch.zeros(1, win_size * win_size + 1, embed_dim), requires_grad=False)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Linear(3, embed_dim)
        self.blocks = nn.ModuleList([ViTBlock(embed_dim, mlp_ratio=2.0)])
        self.invalid_coord_token = nn.Parameter(torch.zeros(embed_dim,))
        torch.nn.init.normal_(self.cls_token, std=.02)
        pe = get_2d_sincos_pos_embed(embed_dim, win_size, cls_token=True)
        self.two_d_pos_embed.data.copy_(torch.from_numpy(pe).float().unsqueeze(0))
        torch.nn.init.normal_(self.invalid_coord_token, std=.02)


class CoordEncAtt(HipModule):
    """Seen surface encoder based on transformer (seen_coord_enc.py:81-139)."""

    def __init__(self, embed_dim=768, n_blocks=12, num_heads=12, win_size=8, mlp_ratio=4.,
                 norm_layer=partial(nn.LayerNorm, eps=1e-6), drop_path=0.1):
        super().__init__()
        if embed_dim // num_heads not in (32, 64):
            raise NotImplementedError("zs_attention supports head_dim 32 or 64")
        self.num_heads, self.win_size = num_heads, win_size
        self.drop_path = float(drop_path)
        self.drop_scales = None    # tests: explicit list of 2 * n_blocks per-sample scale tensors [B]
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.coord_embed = CoordEmb(embed_dim, win_size, num_heads)
        self.blocks = nn.ModuleList([ViTBlock(embed_dim, mlp_ratio) for _ in range(n_blocks)])
        self.norm = norm_layer(embed_dim)
        torch.nn.init.normal_(self.cls_token, std=.02)
        for m in self.modules():                                   # :107-121
            if isinstance(m, nn.Linear):
                torch.nn.init.xavier_uniform_(m.weight)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        self.eval()

    def _pack(self, sd, device):
        d = lambda t: t.detach().float().contiguous().to(device)       # noqa: E731
        C = sd["cls_token"].shape[-1]
        return dict(embed=pack.pack_conv(sd["coord_embed.pos_embed.weight"], sd["coord_embed.pos_embed.bias"],
                                         cin_pad=4).to(device),
                    invalid=d(sd["coord_embed.invalid_coord_token"]), wcls=d(sd["coord_embed.cls_token"].reshape(-1)),
                    wpos=d(sd["coord_embed.two_d_pos_embed"][0]),
                    wblock=blocks.pack_vit_block(sd, "coord_embed.blocks.0", device),
                    cls=d(sd["cls_token"].reshape(-1)), zero_pos=None, C=C,
                    blocks=[blocks.pack_vit_block(sd, "blocks.%d" % i, device) for i in range(len(self.blocks))],
                    nw=d(sd["norm.weight"]), nb=d(sd["norm.bias"]))

    def forward(self, coord_obj, mask_obj):
        """coord_obj [B,H,W,3], mask_obj [B,H,W] bool -> [B, 1 + (H/ws)*(W/ws), C].  In .train() mode under
        autograd: every layer through nn/autograd.py (gradients to all parameters and to coord_obj), with
        timm's per-sample DropPath on the 12 global blocks (seen_coord_enc.py:93-97, drop_path 0.1)."""
        self._need_gpu(coord_obj, "coord_obj")
        if self.training and torch.is_grad_enabled():
            with A.deferred_bn_counters():
                return self._forward_train(coord_obj, mask_obj)
        with torch.no_grad():
            return self._forward_eval(coord_obj, mask_obj)

    def _drop_scale(self, B, device):
        if self.drop_path <= 0.0:
            return None
        keep = 1.0 - self.drop_path
        return torch.empty(B, dtype=torch.float32, device=device).bernoulli_(keep).div_(keep)

    def _forward_train(self, coord_obj, mask_obj):
        ce = self.coord_embed
        B, H, W, _ = coord_obj.shape
        win, C = self.win_size, self.cls_token.shape[-1]
        x4 = A.pad_channels(coord_obj.float().contiguous(), 4)
        emb = A.linear(x4, ce.pos_embed.weight, ce.pos_embed.bias, cin=3)                        # :51
        tok = A.window_tokens(emb, mask_obj, ce.invalid_coord_token, ce.cls_token.view(-1),      # :52-66
                              ce.two_d_pos_embed.detach()[0], win)
        tok = train_blocks.vit_block(tok, ce.blocks[0], self.num_heads)                          # :68-69
        n = (H // win) * (W // win)
        feat = tok[:, 0].reshape(B, n, C)                                                        # :71
        zero_pos = torch.zeros(n + 1, C, device=emb.device)
        x = A.assemble_tokens(feat, self.cls_token.view(-1), zero_pos)                           # :127-131
        nb = len(self.blocks)
        scales = list(self.drop_scales) if self.drop_scales is not None else \
            [self._drop_scale(B, emb.device) for _ in range(2 * nb)]
        for i, blk in enumerate(self.blocks):
            x = train_blocks.vit_block(x, blk, self.num_heads, (scales[2 * i], scales[2 * i + 1]))
        return A.layer_norm(x, self.norm.weight, self.norm.bias, 1e-6)

    def _forward_eval(self, coord_obj, mask_obj):
        pk = self.packed(coord_obj.device)
        B, H, W, _ = coord_obj.shape
        win, C = self.win_size, pk["C"]
        emb = ops.conv2d(ops.pad_channels(coord_obj.float().contiguous(), 4), pk["embed"])      # Linear(3, C)
        tok = ops.window_tokens(emb, mask_obj, pk["invalid"], pk["wcls"], pk["wpos"], win)
        tok = blocks.run_vit_block(tok, pk["wblock"], self.num_heads)
        n = (H // win) * (W // win)
        feat = tok[:, 0].reshape(B, n, C).contiguous()
        if pk["zero_pos"] is None or pk["zero_pos"].shape[0] != n + 1:
            pk["zero_pos"] = torch.zeros(n + 1, C, device=emb.device)
        x = ops.assemble_tokens(feat, pk["cls"], pk["zero_pos"])
        st = None                          # row statistics of x from the previous block's fc2 (few rows: fused LayerNorms)
        for blk in pk["blocks"]:
            x, st = blocks.run_vit_block(x, blk, self.num_heads, stats=st, want_stats=True)
        return ops.layer_norm(x, pk["nw"], pk["nb"], 1e-6)
