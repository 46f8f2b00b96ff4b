"""Oracle: CPU restatement of the image encoders as pure functions of a state dict with the
reference's parameter names.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates (torch-CPU fp32 functional ops, eval mode)
  model/depth/dpt_depth.py:68-122   DPT.forward / DPTDepthModel.forward
  model/depth/vit.py:31-43,57-154   ProjectReadout, forward_vit, _resize_pos_embed, forward_flex
  model/depth/blocks.py:222-343     ResidualConvUnit_custom, FeatureFusionBlock_custom
  utils/layers.py:76-100            Bottleneck_Conv
  model/shape/seen_coord_enc.py:13-78,81-139,141-194   CoordEmb, CoordEncAtt, CoordEncRes
  model/compute_graph/graph_shape.py:115-150           Graph.forward (inference branch)
plus the two un-vendored backbones the reference instantiates (published architectures, see
oracle/standins.py: timm vit_base_resnet50_384, torchvision resnet50).

Pinned by tests/golden/encoder_golden.npz (the reference's own modules run over the stand-ins,
tests/golden/make_encoder_golden.py).  The third-party backbones themselves are "parity
unpinned": neither package is available to compare against.
"""
import math

import torch
import torch.nn.functional as F

from . import frontend_ref


def _sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, 1e-5)


# ---- utils/layers.py:76-100 ----
def bottleneck_conv(sd, p, x, k=1):
    two_d = x.dim() == 2
    if two_d:
        x = x[:, :, None, None]
    out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".linear1.weight"], None, padding=k // 2)))
    out = _bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".linear2.weight"], None, padding=k // 2))
    out = F.relu(out + x)
    return out[:, :, 0, 0] if two_d else out


# ---- timm Block (0.6.12) ----
def vit_block(sd, p, x, heads):
    B, N, C = x.shape
    h = F.layer_norm(x, (C,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-6)
    qkv = F.linear(h, sd[p + ".attn.qkv.weight"], sd[p + ".attn.qkv.bias"])
    q, k, v = qkv.reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4).unbind(0)
    attn = ((q @ k.transpose(-2, -1)) * (C // heads) ** -0.5).softmax(dim=-1)
    a = (attn @ v).transpose(1, 2).reshape(B, N, C)
    x = x + F.linear(a, sd[p + ".attn.proj.weight"], sd[p + ".attn.proj.bias"])
    h = F.layer_norm(x, (C,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-6)
    h = F.gelu(F.linear(h, sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"]))
    return x + F.linear(h, sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])


# ---- timm ResNetV2 (3,4,9), non pre-activation, StdConv2dSame(eps=1e-8), GroupNorm(32) ----
def _pad_same(x, k, s, value=0.0):
    ih, iw = x.shape[-2:]
    ph = max((math.ceil(ih / s) - 1) * s + k - ih, 0)
    pw = max((math.ceil(iw / s) - 1) * s + k - iw, 0)
    return F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2], value=value) if (ph or pw) else x


def _std_conv(w, x, stride=1):
    ws = F.batch_norm(w.reshape(1, w.shape[0], -1), None, None, training=True, momentum=0.0, eps=1e-8).reshape_as(w)
    return F.conv2d(_pad_same(x, w.shape[-1], stride), ws, None, stride)


def _gn(sd, p, x, act=True):
    x = F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], 1e-5)
    return F.relu(x) if act else x


def resnetv2_stages(sd, x):
    """-> (stem, [stage0, stage1, stage2]) feature maps; sd keys relative to `...patch_embed.backbone.`"""
    x = _gn(sd, "stem.norm", _std_conv(sd["stem.conv.weight"], x, 2))
    x = stem = F.max_pool2d(_pad_same(x, 3, 2, value=-float("inf")), 3, 2)
    feats = []
    for s, depth in enumerate((3, 4, 9)):
        for b in range(depth):
            p = "stages.%d.blocks.%d" % (s, b)
            stride = 2 if (b == 0 and s > 0) else 1
            shortcut = x
            if b == 0:
                shortcut = _gn(sd, p + ".downsample.norm", _std_conv(sd[p + ".downsample.conv.weight"], x, stride),
                               act=False)
            y = _gn(sd, p + ".norm1", _std_conv(sd[p + ".conv1.weight"], x))
            y = _gn(sd, p + ".norm2", _std_conv(sd[p + ".conv2.weight"], y, stride))
            y = _gn(sd, p + ".norm3", _std_conv(sd[p + ".conv3.weight"], y), act=False)
            x = F.relu(y + shortcut)
        feats.append(x)
    return stem, feats


# ---- model/depth: DPT-hybrid ----
def _readout_project(sd, p, tokens):
    """vit.py:31-43: Linear(2C -> C) + GELU on [patch token | cls token]."""
    feats = torch.cat((tokens[:, 1:], tokens[:, :1].expand_as(tokens[:, 1:])), -1)
    return F.gelu(F.linear(feats, sd[p + ".project.0.weight"], sd[p + ".project.0.bias"]))


def _rcu(sd, p, x):
    """blocks.py:222-281 (bn=False, activation ReLU, not in place)."""
    out = F.conv2d(F.relu(x), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    out = F.conv2d(F.relu(out), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    return out + x


def _fusion(sd, p, x, skip=None):
    """blocks.py:284-343: (+ RCU1(skip)) -> RCU2 -> x2 bilinear (align_corners) -> 1x1."""
    if skip is not None:
        x = x + _rcu(sd, p + ".resConfUnit1", skip)
    x = _rcu(sd, p + ".resConfUnit2", x)
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    return F.conv2d(x, sd[p + ".out_conv.weight"], sd[p + ".out_conv.bias"])


@torch.no_grad()
def dpt_depth(sd, image, taps=None):
    """DPTDepthModel.forward(image, get_feat=True) -> (depth [B,1,H,W], layer_4 [B,768,H/32,W/32]).
    `sd` keys are relative to `dpt_depth.`; `taps` (dict) receives intermediates."""
    taps = {} if taps is None else taps
    x = image * 2 - 1
    B, _, H, W = x.shape
    vit = _sub(sd, "pretrained.model.")
    gh, gw = H // 16, W // 16
    # vit.py:103-120: position embedding resized from its native grid on every call
    pos = vit["pos_embed"]
    g0 = int(math.sqrt(pos.shape[1] - 1))
    grid = pos[0, 1:].reshape(1, g0, g0, -1).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=(gh, gw), mode="bilinear", align_corners=False)
    pos = torch.cat([pos[:, :1], grid.permute(0, 2, 3, 1).reshape(1, gh * gw, -1)], 1)
    stem, feats = resnetv2_stages(_sub(vit, "patch_embed.backbone."), x)
    taps.update(stem=stem, stage0=feats[0], stage1=feats[1], stage2=feats[2])
    tok = F.conv2d(feats[2], vit["patch_embed.proj.weight"], vit["patch_embed.proj.bias"]).flatten(2).transpose(1, 2)
    tok = torch.cat((vit["cls_token"].expand(B, -1, -1), tok), 1) + pos
    hooked = {}
    for i in range(12):
        tok = vit_block(vit, "blocks.%d" % i, tok, 12)
        hooked[i] = tok
    taps.update(block0=hooked[0], block8=hooked[8], block11=hooked[11])

    def reassemble(idx, p):
        t = _readout_project(sd, "pretrained.%s.0" % p, hooked[idx])
        t = t.transpose(1, 2).reshape(B, -1, gh, gw)
        return F.conv2d(t, sd["pretrained.%s.3.weight" % p], sd["pretrained.%s.3.bias" % p])
    layer_1, layer_2 = feats[0], feats[1]                      # Identity post-processing (vit.py:409-414)
    layer_3 = reassemble(8, "act_postprocess3")
    layer_4 = reassemble(11, "act_postprocess4")
    layer_4 = F.conv2d(layer_4, sd["pretrained.act_postprocess4.4.weight"], sd["pretrained.act_postprocess4.4.bias"],
                       stride=2, padding=1)
    rn = [F.conv2d(l, sd["scratch.layer%d_rn.weight" % (i + 1)], None, padding=1)
          for i, l in enumerate((layer_1, layer_2, layer_3, layer_4))]
    taps.update(layer3_rn=rn[2], layer4_rn=rn[3])
    path4 = _fusion(sd, "scratch.refinenet4", rn[3])
    path3 = _fusion(sd, "scratch.refinenet3", path4, rn[2])
    path2 = _fusion(sd, "scratch.refinenet2", path3, rn[1])
    path1 = _fusion(sd, "scratch.refinenet1", path2, rn[0])
    taps.update(path4=path4, path3=path3, path2=path2, path1=path1)
    o = F.conv2d(path1, sd["scratch.output_conv.0.weight"], sd["scratch.output_conv.0.bias"], padding=1)
    o = F.interpolate(o, scale_factor=2, mode="bilinear", align_corners=True)
    o = F.relu(F.conv2d(o, sd["scratch.output_conv.2.weight"], sd["scratch.output_conv.2.bias"], padding=1))
    o = F.relu(F.conv2d(o, sd["scratch.output_conv.4.weight"], sd["scratch.output_conv.4.bias"]))
    return o.clamp(min=0, max=1), layer_4


# ---- torchvision resnet50 + CoordEncRes (seen_coord_enc.py:141-194) ----
def _bottleneck_v1(sd, p, x, stride, down):
    identity = x
    if down:
        identity = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride))
    out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"])))
    out = F.relu(_bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], None, stride, 1)))
    out = _bn(sd, p + ".bn3", F.conv2d(out, sd[p + ".conv3.weight"]))
    return F.relu(out + identity)


@torch.no_grad()
def coord_enc_res(sd, coord, mask):
    """coord [B,3,H,W], mask [B,1,H,W] -> [B, 1 + (H/16)*(W/16), latent]; sd relative to `coord_encoder.`"""
    B = coord.shape[0]
    x = coord * mask.float()
    e = _sub(sd, "encoder.")
    x = F.max_pool2d(F.relu(_bn(e, "bn1", F.conv2d(x, e["conv1.weight"], None, 2, 3))), 3, 2, 1)
    layer3 = None
    for li, (blocks, stride) in enumerate(((3, 1), (4, 2), (6, 2), (3, 2)), 1):
        for b in range(blocks):
            x = _bottleneck_v1(e, "layer%d.%d" % (li, b), x, stride if b == 0 else 1, b == 0)
        if li == 3:
            layer3 = x
    g = x.mean((2, 3))
    g = bottleneck_conv(e, "fc.1", bottleneck_conv(e, "fc.0", g))
    g = F.linear(g, e["fc.2.weight"], e["fc.2.bias"]).unsqueeze(1)
    loc = bottleneck_conv(sd, "depth_feat_proj.1", bottleneck_conv(sd, "depth_feat_proj.0", layer3))
    loc = F.conv2d(loc, sd["depth_feat_proj.2.weight"], sd["depth_feat_proj.2.bias"])
    loc = loc.view(B, g.shape[-1], -1).permute(0, 2, 1)
    return torch.cat([g, loc], 1)


# ---- CoordEmb + CoordEncAtt (seen_coord_enc.py:13-139) ----
@torch.no_grad()
def coord_enc_att(sd, coord, mask, heads=8, win=8, n_blocks=12):
    """coord [B,H,W,3], mask [B,H,W] bool -> [B, 1 + (H/win)*(W/win), C]."""
    emb = F.linear(coord, sd["coord_embed.pos_embed.weight"], sd["coord_embed.pos_embed.bias"])
    emb = torch.where(mask[..., None], emb, sd["coord_embed.invalid_coord_token"].expand_as(emb))
    B, H, W, C = emb.shape
    emb = emb.view(B, H // win, win, W // win, win, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, win * win, C)
    pe = sd["coord_embed.two_d_pos_embed"]
    emb = emb + pe[:, 1:]
    cls = (sd["coord_embed.cls_token"] + pe[:, :1]).expand(emb.shape[0], -1, -1)
    emb = vit_block(sd, "coord_embed.blocks.0", torch.cat((cls, emb), 1), heads)
    tok = emb[:, 0].view(B, (H // win) * (W // win), C)
    tok = torch.cat((sd["cls_token"].expand(B, -1, -1), tok), 1)
    for i in range(n_blocks):
        tok = vit_block(sd, "blocks.%d" % i, tok, heads)
    return F.layer_norm(tok, (C,), sd["norm.weight"], sd["norm.bias"], 1e-6)


# ---- Graph.forward, inference branch (graph_shape.py:115-150) ----
@torch.no_grad()
def graph_forward(sd, rgb, mask, H=224, W=224):
    """-> dict(depth_pred, intr_feat, intr_params, intr_pred, seen_points, latent_depth) for the
    default configuration (resnet coordinate encoder, dsp forced to 1, graph_shape.py:41-43)."""
    depth, feat = dpt_depth(_sub(sd, "dpt_depth."), rgb)
    f = bottleneck_conv(sd, "intr_head.1", bottleneck_conv(sd, "intr_head.0", feat, 3), 3).mean((2, 3))
    params = F.linear(f, sd["intr_proj.weight"], sd["intr_proj.bias"])
    intr = frontend_ref.intr_param2mtx(H, W, params)
    seen, coord, mask_dsp, _, _ = frontend_ref.seen_surface(depth, intr, mask, 1)
    latent = coord_enc_res(_sub(sd, "coord_encoder."), coord, mask_dsp)
    return dict(depth_pred=depth, intr_feat=feat, intr_params=params, intr_pred=intr, seen_points=seen,
                latent_depth=latent)
