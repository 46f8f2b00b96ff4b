"""Composite blocks of the encoders: pack (host, once per checkpoint) + run (launch sequence).
Every `pack_*` takes a {name: CPU tensor} state dict with the reference's parameter names and a
prefix; every `run_*` takes channels-last GPU activations.  The arithmetic is entirely in
csrc/nn_conv.hip / csrc/nn_ops.hip."""
import torch

from . import ops, pack
from . import branch
from .branch import Branch


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

    def lin_ln(n, norm):
        """Linear(LayerNorm(x)) with the norm's affine folded in: ((x - mu) rstd * g + b) W^T + c = ((x - mu) rstd) (g . W)^T +
        (W b + c) - the fused launch (ops.conv2d ln_in) then only needs the row statistics."""
        w, c = sd[p + n + ".weight"].double(), sd[p + n + ".bias"].double()
        g, b = sd[p + norm + ".weight"].double(), sd[p + norm + ".bias"].double()
        return pack.pack_conv((w * g[None, :]).float(), (c + w @ b).float()).to(device)
    return dict(n1w=_dev(sd[p + ".norm1.weight"], device), n1b=_dev(sd[p + ".norm1.bias"], device),
                n2w=_dev(sd[p + ".norm2.weight"], device), n2b=_dev(sd[p + ".norm2.bias"], device),
                qkv=lin(".attn.qkv"), proj=lin(".attn.proj"), fc1=lin(".mlp.fc1"), fc2=lin(".mlp.fc2"),
                qkv_ln=lin_ln(".attn.qkv", ".norm1"), fc1_ln=lin_ln(".mlp.fc1", ".norm2"))


FUSED_VIT_MAX_ROWS = 1024       # beyond that the large-tile GEMM kernels serve the linear layers: separate LayerNorm launches


def run_vit_block(x, pk, heads, stats=None, want_stats=False):
    """timm Block.  With few rows (batch 1) the two LayerNorms ride on the neighbouring GEMMs: proj / fc2 write the row
    statistics of their (residual-added) outputs, qkv / fc1 normalise their input rows on load (ops.conv2d ln_in /
    stats_out).  `stats` = the row statistics of x from the previous block's fc2 (None: norm1 runs as its own launch);
    want_stats: return (x, Stats of x) for the next block."""
    rows = x.numel() // x.shape[-1]
    if not (ops.fused_ok(x.reshape(1, 1, rows, x.shape[-1])) and rows <= FUSED_VIT_MAX_ROWS):
        h = ops.layer_norm(x, pk["n1w"], pk["n1b"], 1e-6)
        a = ops.attention(ops.linear(h, pk["qkv"]), heads)
        x = ops.linear(a, pk["proj"], res1=x)
        h = ops.layer_norm(x, pk["n2w"], pk["n2b"], 1e-6)
        h = ops.linear(h, pk["fc1"], act=ops.ACT_GELU)
        y = ops.linear(h, pk["fc2"], res1=x)
        return (y, None) if want_stats else y
    if stats is None:
        qkv = ops.linear(ops.layer_norm(x, pk["n1w"], pk["n1b"], 1e-6), pk["qkv"])
    else:
        qkv = ops.linear(x, pk["qkv_ln"], ln_in=(stats, 1e-6))
    a = ops.attention(qkv, heads)
    x, st = ops.linear(a, pk["proj"], res1=x, stats_out="row")
    # the hidden tensor in K16-major layout ([hidden / 16][rows][16]) where the streaming kernel serves both layers: fc2's operand
    # loads become whole lines (it is never read by anything else)
    dim, hidden = pk["fc1_ln"].cin, pk["fc1_ln"].cout
    k16 = (ops.k16_ok(rows, dim, hidden, out_k16=True, ln_tiles=st.tiles) and
           ops.k16_ok(rows, hidden, dim, in_k16=True, row_stats=want_stats, has_res=True))
    h = ops.linear(x, pk["fc1_ln"], act=ops.ACT_GELU, ln_in=(st, 1e-6), out_k16=k16)
    if want_stats:
        return ops.linear(h, pk["fc2"], res1=x, stats_out="row", in_k16=k16)
    return ops.linear(h, pk["fc2"], res1=x, in_k16=k16)


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


def _fused_maps_ok(B, H, W, stages):
    """The fused launches write their GroupNorm statistics per 32-row tile of the [B * H * W, C] matrix, and a tile must not
    straddle two samples (zs_conv2d_nhwc_fused rejects it): with B > 1 EVERY map of the trunk - the stem's output, the pooled
    map and each stride-2 stage's ('same' padding: ceil halves) - needs a multiple of 32 pixels.  (only the first
    two were checked; 2 x 160 x 160 passed the gate and died in stage 1 with 20 x 20 = 400 pixels.)"""
    if B == 1:
        return True
    h, w = H, W
    for _ in range(2 + stages - 1):          # stem conv (stride 2), max pool (stride 2), stages 1.. (stride 2 each)
        h, w = -(-h // 2), -(-w // 2)
        if (h * w) % 32:
            return False
    return True


def run_resnetv2(x, pk, in_scale=1.0, in_shift=0.0, on_stage=None):
    """x [B,H,W,4] (RGB + zero channel) -> list of stage outputs.  on_stage(index, output) is called as soon as a stage's
    output is queued (the caller forks work that needs only that tap)."""
    B, H, W = x.shape[0], x.shape[1], x.shape[2]
    if ops.fused_ok(x) and B * (H // 4) * (W // 4) <= FUSED_GN_MAX_ROWS and _fused_maps_ok(B, H, W, len(pk["stages"])):
        # the stem's GroupNorm + ReLU ride on the max pool (statistics from the convolution's epilogue), then the fused stages
        x, st = ops.conv2d(x, pk["stem"], in_scale=in_scale, in_shift=in_shift, stats_out="group")
        x = ops.gn_relu_max_pool(x, st, pk["stem_gn"][0], pk["stem_gn"][1], 3, 2, "same", GN_EPS)
        return _run_resnetv2_fused(x, pk, on_stage)
    x = ops.conv2d(x, pk["stem"], in_scale=in_scale, in_shift=in_shift)
    x = ops.group_norm(x, *pk["stem_gn"], relu=True)
    x = ops.max_pool(x, 3, 2, "same")
    feats = []
    for blocks in pk["stages"]:
        for blk in blocks:
            shortcut, br = x, None
            if "cd" in blk:                                   # the projection shortcut beside the residual branch
                br = Branch(x, kind=branch.SHORTCUT)
                with br:
                    shortcut = ops.group_norm(ops.conv2d(x, blk["cd"]), *blk["gd"], relu=False)
            y = ops.group_norm(ops.conv2d(x, blk["c1"]), *blk["g1"], relu=True)
            y = ops.group_norm(ops.conv2d(y, blk["c2"]), *blk["g2"], relu=True)
            y = ops.conv2d(y, blk["c3"])
            if br is not None:
                shortcut = br.join(shortcut)
            x = ops.group_norm(y, *blk["g3"], relu=True, residual=shortcut)
        feats.append(x)
        if on_stage is not None:
            on_stage(len(feats) - 1, x)
    return feats


FUSED_GN_MAX_ROWS = 4096        # pixels (batch x map) up to which the small-tile fused launches serve the bottlenecks
GN_EPS = 1e-5


def _run_resnetv2_fused(x, pk, on_stage=None):
    """The bottleneck stages with the GroupNorms riding on the neighbouring convolutions (ops.conv2d gn_in / stats_out): a
    convolution's epilogue writes per-tile group sums of its output; conv2 / conv3 turn them into per-channel scale / shift
    and normalise their operand on load; the block's output relu(GN3(z3) + shortcut) is one pass over z3 from the same sums
    (ops.group_norm_apply, also normalising a projection shortcut).  4 launches per bottleneck instead of 6 (5 instead of 8
    with a projection shortcut), no statistics pass over any tensor."""
    feats = []
    for stage in pk["stages"]:
        for blk in stage:
            shortcut, res_gn = x, None
            if "cd" in blk:
                shortcut, std = ops.conv2d(x, blk["cd"], stats_out="group")
                res_gn = (std, blk["gd"][0], blk["gd"][1])
            z1, st1 = ops.conv2d(x, blk["c1"], stats_out="group")
            z2, st2 = ops.conv2d(z1, blk["c2"], gn_in=(st1, blk["g1"][0], blk["g1"][1], GN_EPS), stats_out="group")
            z3, st3 = ops.conv2d(z2, blk["c3"], gn_in=(st2, blk["g2"][0], blk["g2"][1], GN_EPS), stats_out="group")
            x = ops.group_norm_apply(z3, st3, blk["g3"][0], blk["g3"][1], GN_EPS, relu=True, residual=shortcut, res_gn=res_gn)
        feats.append(x)
        if on_stage is not None:
            on_stage(len(feats) - 1, x)
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


def run_resnet50(x, pk, on_layer=None):
    """x [B,H,W,4] -> [layer1, layer2, layer3, layer4] outputs.  on_layer(index, output): see run_resnetv2."""
    x = ops.max_pool(ops.conv2d(x, pk["stem"], act=ops.ACT_RELU), 3, 2, 1)
    feats = []
    for layer in pk["layers"]:
        for blk in layer:
            identity, br = x, None
            if "cd" in blk:                                   # the projection shortcut beside the residual branch
                br = Branch(x, kind=branch.SHORTCUT)
                with br:
                    identity = ops.conv2d(x, blk["cd"])
            y = ops.conv2d(x, blk["c1"], act=ops.ACT_RELU)
            y = ops.conv2d(y, blk["c2"], act=ops.ACT_RELU)
            if br is not None:
                identity = br.join(identity)
            x = ops.conv2d(y, blk["c3"], res1=identity, act=ops.ACT_RELU)
        feats.append(x)
        if on_layer is not None:
            on_layer(len(feats) - 1, x)
    return feats


# ---- DPT fusion (model/depth/blocks.py:222-343) ----
def pack_rcu(sd, p, device):
    return dict(c1=pack.pack_conv(sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1).to(device),
                c2=pack.pack_conv(sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1).to(device))


def run_rcu_head(x, pk):
    """The first half of run_rcu: relu(conv1(relu(x))) - for a skip input it depends on the backbone tap only."""
    return ops.conv2d(x, pk["c1"], in_relu=True, act=ops.ACT_RELU)


def run_rcu(x, pk, plus=None, head=None):
    """conv2(relu(conv1(relu(x)))) + x (+ plus): ResidualConvUnit_custom with the fusion add folded
    into the second convolution's epilogue.  `head` = run_rcu_head(x, pk) computed earlier (on a side branch)."""
    h = run_rcu_head(x, pk) if head is None else head
    return ops.conv2d(h, pk["c2"], res1=x, res2=plus)


def pack_fusion(sd, p, device):
    return dict(r1=pack_rcu(sd, p + ".resConfUnit1", device), r2=pack_rcu(sd, p + ".resConfUnit2", device),
                out=pack.pack_conv(sd[p + ".out_conv.weight"], sd[p + ".out_conv.bias"]).to(device))


def run_fusion(x, pk, skip=None, skip_head=None):
    if skip is not None:
        x = run_rcu(skip, pk["r1"], plus=x, head=skip_head)
    x = run_rcu(x, pk["r2"])
    # The reference interpolates, then applies out_conv (1x1 + bias, no activation; blocks.py:232-342).  Both are linear and the
    # bilinear weights of a pixel sum to 1, so conv1x1(upsample(x)) == upsample(conv1x1(x)) in real arithmetic (floating point:
    # a reassociation, ~1e-7 relative); the convolution at the LOW resolution is a quarter of the work (refinenet1: 87,808
    # instead of 351,232 pixels at batch 28).  ZS_FUSION_CONV_FIRST=0 restores the reference's order.
    if CONV_BEFORE_UPSAMPLE:
        return ops.upsample2x(ops.conv2d(x, pk["out"]))
    return ops.conv2d(ops.upsample2x(x), pk["out"])
