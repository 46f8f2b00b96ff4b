"""Stand-ins for the un-vendored model code the reference imports.  TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py).

The reference builds its encoders out of two third-party packages that are absent from
/root/reference and from this image:
  * timm (unpinned `pip install timm`, README.md:20; timm==0.6.12 semantics restated here):
      timm.create_model("vit_base_resnet50_384")   model/depth/vit.py:478
      timm.models.vision_transformer.Block         model/shape/seen_coord_enc.py:8,32,99
  * torchvision (0.12, README.md:18):
      torchvision.models.resnet50                  model/shape/seen_coord_enc.py:149
These classes restate the PUBLISHED architectures (torch-CPU nn.Modules, parameter names as
the packages register them, so state-dict keys match real checkpoints).  They are "parity
unpinned": no copy of either package exists here to check them against.  What IS pinned with
their help: every line of the reference's OWN encoder code (DPT reassemble / fusion / head,
pos-embed resize, readout projection, CoordEncRes, CoordEncAtt, Bottleneck_Conv, the intrinsics
head, Graph.forward), which tests/golden/make_encoder_golden.py runs on top of these stand-ins.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# timm.models.vision_transformer (0.6.12): Mlp, Attention, Block, VisionTransformer (hybrid)
# ----------------------------------------------------------------------------------------------
class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, bias=True, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.drop1(self.act(self.fc1(x)))))


class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        assert not self.training, "stand-in is eval-only"
        return x


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        assert dim % num_heads == 0
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        attn = (q @ k.transpose(-2, -1)) * self.scale
        attn = self.attn_drop(attn.softmax(dim=-1))
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj_drop(self.proj(x))


class Block(nn.Module):
    """Pre-LN transformer block; LayerScale is the identity when init_values is None."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, drop=0.0, attn_drop=0.0, init_values=None,
                 drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        assert init_values is None
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
        self.ls1 = nn.Identity()
        self.drop_path1 = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.ls2 = nn.Identity()
        self.drop_path2 = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()

    def forward(self, x):
        x = x + self.drop_path1(self.ls1(self.attn(self.norm1(x))))
        x = x + self.drop_path2(self.ls2(self.mlp(self.norm2(x))))
        return x


def pad_same(x, k, s, value=0.0):
    """timm layers/padding.py: TF 'SAME': total = max((ceil(n/s)-1)*s + k - n, 0), the odd pixel
    goes to the bottom / right."""
    ih, iw = x.shape[-2:]
    ph = max((math.ceil(ih / s) - 1) * s + k - ih, 0)
    pw = max((math.ceil(iw / s) - 1) * s + k - iw, 0)
    if ph > 0 or pw > 0:
        x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2], value=value)
    return x


class StdConv2dSame(nn.Conv2d):
    """Weight-standardised conv with TF 'SAME' padding (layers/std_conv.py), bias-free."""

    def __init__(self, in_channel, out_channels, kernel_size, stride=1, eps=1e-8):
        super().__init__(in_channel, out_channels, kernel_size, stride=stride, padding=0, bias=False)
        self.eps = eps

    def forward(self, x):
        x = pad_same(x, self.kernel_size[0], self.stride[0])
        weight = F.batch_norm(self.weight.reshape(1, self.out_channels, -1), None, None, training=True,
                              momentum=0.0, eps=self.eps).reshape_as(self.weight)
        return F.conv2d(x, weight, None, self.stride, (0, 0), self.dilation, self.groups)


class GroupNormAct(nn.GroupNorm):
    def __init__(self, num_channels, num_groups=32, eps=1e-5, apply_act=True):
        super().__init__(num_groups, num_channels, eps=eps, affine=True)
        self.apply_act = apply_act

    def forward(self, x):
        x = F.group_norm(x, self.num_groups, self.weight, self.bias, self.eps)
        return F.relu(x) if self.apply_act else x


class MaxPool2dSame(nn.Module):
    def forward(self, x):
        return F.max_pool2d(pad_same(x, 3, 2, value=-float("inf")), 3, 2, (0, 0))


class DownsampleConv(nn.Module):
    def __init__(self, in_chs, out_chs, stride):
        super().__init__()
        self.conv = StdConv2dSame(in_chs, out_chs, 1, stride=stride)
        self.norm = GroupNormAct(out_chs, apply_act=False)

    def forward(self, x):
        return self.norm(self.conv(x))


class BottleneckV2(nn.Module):
    """timm resnetv2.Bottleneck (the non pre-activation variant the hybrid ViTs use)."""

    def __init__(self, in_chs, out_chs, stride, proj):
        super().__init__()
        mid = out_chs // 4
        self.downsample = DownsampleConv(in_chs, out_chs, stride) if proj else None
        self.conv1 = StdConv2dSame(in_chs, mid, 1)
        self.norm1 = GroupNormAct(mid)
        self.conv2 = StdConv2dSame(mid, mid, 3, stride=stride)
        self.norm2 = GroupNormAct(mid)
        self.conv3 = StdConv2dSame(mid, out_chs, 1)
        self.norm3 = GroupNormAct(out_chs, apply_act=False)

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        x = self.norm1(self.conv1(x))
        x = self.norm2(self.conv2(x))
        x = self.norm3(self.conv3(x))
        return F.relu(x + shortcut)


class ResNetStage(nn.Module):
    def __init__(self, in_chs, out_chs, stride, depth):
        super().__init__()
        self.blocks = nn.Sequential(*[BottleneckV2(in_chs if i == 0 else out_chs, out_chs, stride if i == 0 else 1,
                                                   proj=(i == 0)) for i in range(depth)])

    def forward(self, x):
        return self.blocks(x)


class ResNetV2(nn.Module):
    """ResNetV2(layers=(3,4,9), num_classes=0, global_pool='', preact=False, stem_type='same',
    conv_layer=partial(StdConv2dSame, eps=1e-8)) as vision_transformer_hybrid._resnetv2 builds it."""

    def __init__(self, layers=(3, 4, 9), in_chans=3):
        super().__init__()
        self.stem = nn.Sequential()
        self.stem.add_module("conv", StdConv2dSame(in_chans, 64, 7, stride=2))
        self.stem.add_module("norm", GroupNormAct(64))
        self.stem.add_module("pool", MaxPool2dSame())
        chans, prev, stages = (256, 512, 1024, 2048), 64, []
        for i, depth in enumerate(layers):
            stages.append(ResNetStage(prev, chans[i], 1 if i == 0 else 2, depth))
            prev = chans[i]
        self.stages = nn.Sequential(*stages)
        self.norm = nn.Identity()
        self.head = nn.Identity()
        self.num_features = prev

    def forward(self, x):
        return self.head(self.norm(self.stages(self.stem(x))))


class HybridEmbed(nn.Module):
    def __init__(self, backbone, feature_dim=1024, embed_dim=768):
        super().__init__()
        self.backbone = backbone
        self.proj = nn.Conv2d(feature_dim, embed_dim, kernel_size=1, stride=1)

    def forward(self, x):
        x = self.backbone(x)
        return self.proj(x).flatten(2).transpose(1, 2)


class VisionTransformerHybrid(nn.Module):
    """vit_base_resnet50_384: R50 (3,4,9) + ViT-B/16-equivalent grid, 384^2 -> 24^2 + cls tokens."""

    def __init__(self, img_size=384, embed_dim=768, depth=12, num_heads=12, num_classes=1000):
        super().__init__()
        self.patch_embed = HybridEmbed(ResNetV2(), 1024, embed_dim)
        n = (img_size // 16) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.randn(1, n + 1, embed_dim) * 0.02)
        self.pos_drop = nn.Dropout(0.0)
        norm = lambda d: nn.LayerNorm(d, eps=1e-6)   # noqa: E731
        self.blocks = nn.Sequential(*[Block(embed_dim, num_heads, 4.0, qkv_bias=True, norm_layer=norm)
                                      for _ in range(depth)])
        self.norm = norm(embed_dim)
        self.fc_norm = nn.Identity()
        self.head = nn.Linear(embed_dim, num_classes)


def create_model(name, pretrained=False, **kwargs):
    assert name == "vit_base_resnet50_384", name
    return VisionTransformerHybrid()


# ----------------------------------------------------------------------------------------------
# torchvision.models.resnet50 (0.12; ResNet v1.5: the stride sits on the 3x3 conv)
# ----------------------------------------------------------------------------------------------
class BottleneckV1(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + identity)


class ResNet50(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, 3, 1)
        self.layer2 = self._make_layer(128, 4, 2)
        self.layer3 = self._make_layer(256, 6, 2)
        self.layer4 = self._make_layer(512, 3, 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, num_classes)

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
        layers = [BottleneckV1(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        layers += [BottleneckV1(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = torch.flatten(self.avgpool(x), 1)
        return self.fc(x)


def resnet50(pretrained=False, **kwargs):
    return ResNet50()


def install():
    """Register the stand-ins under the names the reference imports (this process only)."""
    import sys
    import types
    timm = types.ModuleType("timm")
    models = types.ModuleType("timm.models")
    vt = types.ModuleType("timm.models.vision_transformer")
    vt.Mlp, vt.DropPath, vt.Block, vt.Attention = Mlp, DropPath, Block, Attention
    vt.PatchEmbed = type("PatchEmbed", (nn.Module,), {})        # rgb_enc.py imports the name only
    timm.create_model = create_model
    timm.models, models.vision_transformer = models, vt
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvm.resnet50 = resnet50
    tv.models = tvm
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.vision_transformer": vt,
                        "torchvision": tv, "torchvision.models": tvm})
