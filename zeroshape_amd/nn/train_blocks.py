"""Training-mode forward of the encoder blocks: the same launch sequences as nn/blocks.py, but
through the autograd bindings (nn/autograd.py) on the modules' own parameters (torch layout, no
host-side packing), with BatchNorm on batch statistics.  Each function mirrors a reference module's
forward under graph.train() (model/shape_engine.py:248-297)."""
import torch.nn as nn

from . import autograd as A


def _bn(x, bn, relu=False, residual=None):
    if not isinstance(bn, nn.BatchNorm2d) or not bn.training:
        raise NotImplementedError("the autograd path runs BatchNorm on batch statistics (module.train()); "
                                  "eval-mode BatchNorm inside a differentiated graph is not built")
    return A.batch_norm_train(x, bn, relu=relu, residual=residual)


def bottleneck_conv(x, m):
    """utils/layers.py:76-100 Bottleneck_Conv."""
    k = m.linear1.kernel_size[0]
    h, x = A.conv2d(x, m.linear1.weight, padding=k // 2, fork=True)         # (see resnet50)
    h = _bn(h, m.bn1, relu=True)
    return _bn(A.conv2d(h, m.linear2.weight, padding=k // 2), m.bn2, relu=True, residual=x)


def resnet50(x, enc):
    """torchvision resnet50 trunk on x [B,H,W,4] (3 channels + zero pad) -> [layer1..layer4]."""
    x = _bn(A.conv2d(x, enc.conv1.weight, stride=2, padding=3), enc.bn1, relu=True)
    x = A.max_pool(x, 3, 2, 1)
    feats = []
    for layer in (enc.layer1, enc.layer2, enc.layer3, enc.layer4):
        for blk in layer:
            stride = blk.conv2.stride[0]
            # x has two consumers: conv1 hands it through to the other one (A.conv2d fork=True: its data-gradient GEMM adds
            # the second gradient in its epilogue, the autograd engine does not launch an add)
            y, identity = A.conv2d(x, blk.conv1.weight, fork=True)
            if hasattr(blk, "downsample"):
                identity = _bn(A.conv2d(identity, blk.downsample[0].weight, stride=stride), blk.downsample[1])
            y = _bn(y, blk.bn1, relu=True)
            y = _bn(A.conv2d(y, blk.conv2.weight, stride=stride, padding=1), blk.bn2, relu=True)
            x = _bn(A.conv2d(y, blk.conv3.weight), blk.bn3, relu=True, residual=identity)
        feats.append(x)
    return feats


def vit_block(x, blk, heads, scales=None):
    """timm Block: x + drop_path(proj(attn(LN x))); x + drop_path(fc2(gelu(fc1(LN x)))).  scales = (s_attn, s_mlp):
    per-sample DropPath factors [B] (bernoulli(keep) / keep, timm layers/drop.py) or None for drop_path 0 / eval."""
    s_attn, s_mlp = scales if scales is not None else (None, None)
    # (fork=True: the normalisation hands x through to the residual connection and its backward kernel adds that
    # connection's gradient - A.layer_norm)
    h, x = A.layer_norm(x, blk.norm1.weight, blk.norm1.bias, 1e-6, fork=True)
    a = A.attention(A.linear(h, blk.attn.qkv.weight, blk.attn.qkv.bias), heads)
    if s_attn is None:
        x = A.linear(a, blk.attn.proj.weight, blk.attn.proj.bias, res1=x)
    else:
        x = A.add_scaled_rows(x, A.linear(a, blk.attn.proj.weight, blk.attn.proj.bias), s_attn)
    h, x = A.layer_norm(x, blk.norm2.weight, blk.norm2.bias, 1e-6, fork=True)
    h = A.gelu(A.linear(h, blk.mlp.fc1.weight, blk.mlp.fc1.bias))
    if s_mlp is None:
        return A.linear(h, blk.mlp.fc2.weight, blk.mlp.fc2.bias, res1=x)
    return A.add_scaled_rows(x, A.linear(h, blk.mlp.fc2.weight, blk.mlp.fc2.bias), s_mlp)


STD_EPS = 1e-8      # timm StdConv2dSame


def _gn(x, norm, relu, residual=None):
    return A.group_norm(x, norm.weight, norm.bias, norm.num_groups, norm.eps, relu=relu, residual=residual)


def resnetv2(x, bb, in_scale=1.0, in_shift=0.0):
    """timm ResNetV2 (layers (3,4,9), StdConv2dSame, GroupNorm) on x [B,H,W,4] -> stage outputs."""
    x = A.conv2d(x, bb.stem.conv.weight, stride=2, padding="same", std_eps=STD_EPS, in_scale=in_scale, in_shift=in_shift)
    x = _gn(x, bb.stem.norm, True)
    x = A.max_pool(x, 3, 2, "same")
    feats = []
    for stage in bb.stages:
        for blk in stage.blocks:
            stride = blk.conv2.stride[0]
            y, shortcut = A.conv2d(x, blk.conv1.weight, padding="same", std_eps=STD_EPS, fork=True)     # (see resnet50)
            if hasattr(blk, "downsample"):
                shortcut = _gn(A.conv2d(shortcut, blk.downsample.conv.weight, stride=stride, padding="same", std_eps=STD_EPS),
                               blk.downsample.norm, False)
            y = _gn(y, blk.norm1, True)
            y = _gn(A.conv2d(y, blk.conv2.weight, stride=stride, padding="same", std_eps=STD_EPS), blk.norm2, True)
            x = _gn(A.conv2d(y, blk.conv3.weight, padding="same", std_eps=STD_EPS), blk.norm3, True, residual=shortcut)
        feats.append(x)
    return feats


def rcu(x, m, plus=None):
    """ResidualConvUnit_custom (model/depth/blocks.py:222-287): conv2(relu(conv1(relu(x)))) + x (+ plus)."""
    h = A.conv2d(x, m.conv1.weight, m.conv1.bias, padding=1, in_relu=True, act=A.ACT_RELU)
    return A.conv2d(h, m.conv2.weight, m.conv2.bias, padding=1, res1=x, res2=plus)


def fusion(x, m, skip=None):
    """FeatureFusionBlock_custom (blocks.py:290-343)."""
    if skip is not None:
        x = rcu(skip, m.resConfUnit1, plus=x)
    x = rcu(x, m.resConfUnit2)
    # (the inference blocks run out_conv BEFORE the interpolation - blocks.run_fusion; under training that reorder - the same
    # function, rounded differently - was enough to send the 300-iteration run of tests/test_gpu_trained_weights.py down another
    # trajectory, where the from-scratch depth head dies (all-zero depth) and the degenerate seen surface turns the intrinsics
    # head's gradients into NaN; the training path keeps the reference's order, it gained 0.2 ms of 36)
    return A.conv2d(A.upsample2x(x), m.out_conv.weight, m.out_conv.bias)
