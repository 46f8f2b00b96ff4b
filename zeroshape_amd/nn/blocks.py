"""Composite blocks of the encoders: pack (host, once per checkpoint) + run (launch sequence).
Every `pack_*` takes a {name: CPU tensor} state dict with the reference's parameter names and a
prefix; every `run_*` takes channels-last GPU activations.  The arithmetic is entirely in
csrc/nn_conv.hip / csrc/nn_ops.hip."""
import torch

from . import ops, pack


CONV_BEFORE_UPSAMPLE = __import__("os").environ.get("ZS_FUSION_CONV_FIRST", "1") != "0"

def _dev(t, device):
    return t.detach().float().contiguous().to(device)


def _bn(sd, p):
    return dict(weight=sd[p + ".weight"], bias=sd[p + ".bias"], running_mean=sd[p + ".running_mean"],
                running_var=sd[p + ".running_var"], eps=1e-5)


# ---- utils/layers.py:76-100 Bottleneck_Conv: conv-BN-ReLU-conv-BN (+x) ReLU ----
def pack_bottleneck_conv(sd, p, k, device):
    return dict(c1=pack.pack_conv(sd[p + ".linear1.weight"], None, bn=_bn(sd, p + ".bn1"), padding=k // 2).to(device),
                c2=pack.pack_conv(sd[p + ".linear2.weight"], None, bn=_bn(sd, p + ".bn2"), padding=k // 2).to(device))


def run_bottleneck_conv(x, pk):
    h = ops.conv2d(x, pk["c1"], act=ops.ACT_RELU)
    return ops.conv2d(h, pk["c2"], res1=x, act=ops.ACT_RELU)


# ---- timm Block: x + proj(attn(LN(x))); x + fc2(gelu(fc1(LN(x)))) ----
def pack_vit_block(sd, p, device):
    lin = lambda n: pack.pack_conv(sd[p + n + ".weight"], sd[p + n + ".bias"]).to(device)   # noqa: E731
    return dict(n1w=_dev(sd[p + ".norm1.weight"], device), n1b=_dev(sd[p + ".norm1.bias"], device),
                n2w=_dev(sd[p + ".norm2.weight"], device), n2b=_dev(sd[p + ".norm2.bias"], device),
                qkv=lin(".attn.qkv"), proj=lin(".attn.proj"), fc1=lin(".mlp.fc1"), fc2=lin(".mlp.fc2"))


def run_vit_block(x, pk, heads):
    h = ops.layer_norm(x, pk["n1w"], pk["n1b"], 1e-6)
    a = ops.attention(ops.linear(h, pk["qkv"]), heads)
    x = ops.linear(a, pk["proj"], res1=x)
    h = ops.layer_norm(x, pk["n2w"], pk["n2b"], 1e-6)
    h = ops.linear(h, pk["fc1"], act=ops.ACT_GELU)
    return ops.linear(h, pk["fc2"], res1=x)


# ---- timm ResNetV2 stem + stages (StdConv2dSame eps 1e-8, GroupNorm 32) ----
def _std(sd, name, stride, device, cin_pad=None):
    return pack.pack_conv(pack.standardize_weight(sd[name], 1e-8), None, stride=stride, padding="same",
                          cin_pad=cin_pad).to(device)


def _gn(sd, p, device):
    return _dev(sd[p + ".weight"], device), _dev(sd[p + ".bias"], device)


def pack_resnetv2(sd, p, device, layers=(3, 4, 9)):
    pk = dict(stem=_std(sd, p + "stem.conv.weight", 2, device, cin_pad=4), stem_gn=_gn(sd, p + "stem.norm", device),
              stages=[])
    for s, depth in enumerate(layers):
        blocks = []
        for b in range(depth):
            q = "%sstages.%d.blocks.%d" % (p, s, b)
            stride = 2 if (b == 0 and s > 0) else 1
            blk = dict(c1=_std(sd, q + ".conv1.weight", 1, device), g1=_gn(sd, q + ".norm1", device),
                       c2=_std(sd, q + ".conv2.weight", stride, device), g2=_gn(sd, q + ".norm2", device),
                       c3=_std(sd, q + ".conv3.weight", 1, device), g3=_gn(sd, q + ".norm3", device))
            if b == 0:
                blk["cd"] = _std(sd, q + ".downsample.conv.weight", stride, device)
                blk["gd"] = _gn(sd, q + ".downsample.norm", device)
            blocks.append(blk)
        pk["stages"].append(blocks)
    return pk


def run_resnetv2(x, pk, in_scale=1.0, in_shift=0.0):
    """x [B,H,W,4] (RGB + zero channel) -> list of stage outputs."""
    x = ops.conv2d(x, pk["stem"], in_scale=in_scale, in_shift=in_shift)
    x = ops.group_norm(x, *pk["stem_gn"], relu=True)
    x = ops.max_pool(x, 3, 2, "same")
    feats = []
    for blocks in pk["stages"]:
        for blk in blocks:
            shortcut = x
            if "cd" in blk:
                shortcut = ops.group_norm(ops.conv2d(x, blk["cd"]), *blk["gd"], relu=False)
            y = ops.group_norm(ops.conv2d(x, blk["c1"]), *blk["g1"], relu=True)
            y = ops.group_norm(ops.conv2d(y, blk["c2"]), *blk["g2"], relu=True)
            x = ops.group_norm(ops.conv2d(y, blk["c3"]), *blk["g3"], relu=True, residual=shortcut)
        feats.append(x)
    return feats


# ---- torchvision ResNet-50 trunk (eval BatchNorm folded into the conv epilogues) ----
def pack_resnet50(sd, p, device):
    pk = dict(stem=pack.pack_conv(sd[p + "conv1.weight"], None, bn=_bn(sd, p + "bn1"), stride=2, padding=3,
                                  cin_pad=4).to(device), layers=[])
    for li, (blocks, stride) in enumerate(((3, 1), (4, 2), (6, 2), (3, 2)), 1):
        layer = []
        for b in range(blocks):
            q = "%slayer%d.%d" % (p, li, b)
            s = stride if b == 0 else 1
            blk = dict(c1=pack.pack_conv(sd[q + ".conv1.weight"], None, bn=_bn(sd, q + ".bn1")).to(device),
                       c2=pack.pack_conv(sd[q + ".conv2.weight"], None, bn=_bn(sd, q + ".bn2"), stride=s,
                                         padding=1).to(device),
                       c3=pack.pack_conv(sd[q + ".conv3.weight"], None, bn=_bn(sd, q + ".bn3")).to(device))
            if b == 0:
                blk["cd"] = pack.pack_conv(sd[q + ".downsample.0.weight"], None, bn=_bn(sd, q + ".downsample.1"),
                                           stride=s).to(device)
            layer.append(blk)
        pk["layers"].append(layer)
    return pk


def run_resnet50(x, pk):
    """x [B,H,W,4] -> [layer1, layer2, layer3, layer4] outputs."""
    x = ops.max_pool(ops.conv2d(x, pk["stem"], act=ops.ACT_RELU), 3, 2, 1)
    feats = []
    for layer in pk["layers"]:
        for blk in layer:
            identity = ops.conv2d(x, blk["cd"]) if "cd" in blk else x
            y = ops.conv2d(x, blk["c1"], act=ops.ACT_RELU)
            y = ops.conv2d(y, blk["c2"], act=ops.ACT_RELU)
            x = ops.conv2d(y, blk["c3"], res1=identity, act=ops.ACT_RELU)
        feats.append(x)
    return feats


# ---- DPT fusion (model/depth/blocks.py:222-343) ----
def pack_rcu(sd, p, device):
    return dict(c1=pack.pack_conv(sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1).to(device),
                c2=pack.pack_conv(sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1).to(device))


def run_rcu(x, pk, plus=None):
    """conv2(relu(conv1(relu(x)))) + x (+ plus): ResidualConvUnit_custom with the fusion add folded
    into the second convolution's epilogue."""
    h = ops.conv2d(x, pk["c1"], in_relu=True, act=ops.ACT_RELU)
    return ops.conv2d(h, pk["c2"], res1=x, res2=plus)


def pack_fusion(sd, p, device):
    return dict(r1=pack_rcu(sd, p + ".resConfUnit1", device), r2=pack_rcu(sd, p + ".resConfUnit2", device),
                out=pack.pack_conv(sd[p + ".out_conv.weight"], sd[p + ".out_conv.bias"]).to(device))


def run_fusion(x, pk, skip=None):
    if skip is not None:
        x = run_rcu(skip, pk["r1"], plus=x)
    x = run_rcu(x, pk["r2"])
    # The reference interpolates, then applies out_conv (1x1 + bias, no activation; blocks.py:232-342).  Both are linear and the
    # bilinear weights of a pixel sum to 1, so conv1x1(upsample(x)) == upsample(conv1x1(x)) in real arithmetic (floating point:
    # a reassociation, ~1e-7 relative); the convolution at the LOW resolution is a quarter of the work (refinenet1: 87,808
    # instead of 351,232 pixels at batch 28).  ZS_FUSION_CONV_FIRST=0 restores the reference's order.
    if CONV_BEFORE_UPSAMPLE:
        return ops.upsample2x(ops.conv2d(x, pk["out"]))
    return ops.conv2d(ops.upsample2x(x), pk["out"])
