"""GPU parity of the encoder layers (csrc/nn_conv.hip, csrc/nn_ops.hip) through the C ABI vs the
same op in PyTorch fp32 on the CPU (torch.nn.functional).  fp32 MFMA accumulates in a different
order than the CPU kernels, so comparisons are relative to the output scale: <= 2e-5 of max|y| -
for the exact-fp32 and for the split-fp16 (ZS_CONV_F16X3) arithmetic of the convolution engine alike."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def close(got, want, tol=2e-5):
    got, want = got.detach().cpu().double(), want.detach().double()
    assert got.shape == want.shape, (got.shape, want.shape)
    scale = max(want.abs().max().item(), 1e-6)
    err = (got - want).abs().max().item()
    assert err <= tol * scale, "max err %.3g vs scale %.3g" % (err, scale)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def same_pad(x, k, s, value=0.0):
    """timm pad_same: total = max((ceil(n/s)-1)*s + k - n, 0), extra on the bottom/right."""
    H, W = x.shape[-2:]
    ph = max((-(-H // s) - 1) * s + k - H, 0)
    pw = max((-(-W // s) - 1) * s + k - W, 0)
    return F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), value=value)


CONV_CASES = [
    (2, 1024, 14, 14, 256, 1, 1, 0),          # long K split four ways
    (1, 768, 7, 7, 768, 3, 1, 1),             # intrinsics head: 49 pixels, K = 6912
    (1, 256, 56, 56, 256, 3, 1, 1),           # DPT fusion at B=1
    # B, Cin, H, W, Cout, k, stride, padding
    (2, 64, 14, 14, 256, 1, 1, 0),
    (1, 256, 9, 7, 64, 3, 1, 1),
    (2, 128, 15, 15, 128, 3, 2, 1),
    (1, 64, 28, 28, 96, 3, 2, "same"),
    (1, 4, 37, 41, 64, 7, 2, 3),
    (1, 4, 32, 32, 64, 7, 2, "same"),
    (3, 32, 12, 12, 1, 1, 1, 0),
    (1, 768, 7, 7, 3, 1, 1, 0),
    (1, 512, 5, 5, 200, 1, 2, 0),
]


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,stride,padding", CONV_CASES)
def test_conv2d_matches_torch(B, Cin, H, W, Cout, k, stride, padding):
    from zeroshape_amd.nn import ops, pack
    g = torch.Generator().manual_seed(Cin * 31 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    if padding == "same":
        want = F.conv2d(same_pad(x, k, stride), w, b, stride=stride)
    else:
        want = F.conv2d(x, w, b, stride=stride, padding=padding)
    pc = pack.pack_conv(w, b, stride=stride, padding=padding).to("cuda")
    prev = ops.CONV_PRECISION
    try:
        for prec in ("f32", "f16x3"):          # exact fp32 MFMA / split-fp16 (ZS_CONV_F16X3), every tiling
            ops.set_conv_precision(prec)
            for tiling in (None, "large", "small"):
                got = ops.conv2d(nhwc(x).cuda(), pc, tiling=tiling)
                close(got, nhwc(want))
    finally:
        ops.set_conv_precision(prev)


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_conv2d_fused_epilogue_and_input_transforms(prec, monkeypatch):
    from zeroshape_amd.nn import ops, pack
    monkeypatch.setattr(ops, "CONV_PRECISION", prec)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 64, 10, 10, generator=g)
    w = torch.randn(160, 64, 3, 3, generator=g) / 24
    bn = dict(weight=torch.rand(160, generator=g) + 0.5, bias=torch.randn(160, generator=g),
              running_mean=torch.randn(160, generator=g), running_var=torch.rand(160, generator=g) + 0.5, eps=1e-5)
    r1, r2 = torch.randn(2, 160, 10, 10, generator=g), torch.randn(2, 160, 10, 10, generator=g)
    conv = F.conv2d(F.relu(x), w, None, padding=1)
    want = F.relu(F.batch_norm(conv, bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"], False, 0.0,
                               bn["eps"]) + r1 + r2)
    pc = pack.pack_conv(w, None, bn=bn, padding=1).to("cuda")
    for tiling in ("large", "small"):
        got = ops.conv2d(nhwc(x).cuda(), pc, res1=nhwc(r1).cuda(), res2=nhwc(r2).cuda(), act=ops.ACT_RELU,
                         in_relu=True, tiling=tiling)
        close(got, nhwc(want))
    # affine input transform applies to in-bounds taps only (zero padding stays zero): conv(2x-1)
    want = F.gelu(F.conv2d(2 * x - 1, w, None, padding=1))
    for tiling in ("large", "small"):
        got = ops.conv2d(nhwc(x).cuda(), pack.pack_conv(w, None, padding=1).to("cuda"), act=ops.ACT_GELU,
                         in_scale=2.0, in_shift=-1.0, tiling=tiling)
        close(got, nhwc(want))
    want = F.conv2d(x, w, None, padding=1).clamp(0, 1)
    got = ops.conv2d(nhwc(x).cuda(), pack.pack_conv(w, None, padding=1).to("cuda"), act=ops.ACT_RELU_CLAMP1)
    close(got, nhwc(want))


def test_linear_on_tokens():
    from zeroshape_amd.nn import ops, pack
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 197, 768, generator=g)
    w = torch.randn(2304, 768, generator=g) / 28
    b = torch.randn(2304, generator=g)
    got = ops.linear(x.cuda(), pack.pack_conv(w, b).to("cuda"))
    close(got, F.linear(x, w, b))
    r = torch.randn(2, 197, 2304, generator=g)
    got = ops.linear(x.cuda(), pack.pack_conv(w, b).to("cuda"), res1=r.cuda(), act=ops.ACT_GELU)
    close(got, F.gelu(F.linear(x, w, b) + r))


def test_cin_padding_for_rgb_input():
    from zeroshape_amd.nn import ops, pack
    g = torch.Generator().manual_seed(4)
    x = torch.rand(2, 3, 40, 40, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) / 12
    want = F.conv2d(x, w, None, stride=2, padding=3)
    pc = pack.pack_conv(w, None, stride=2, padding=3, cin_pad=4).to("cuda")
    got = ops.conv2d(ops.to_nhwc(x.cuda(), cpad=4), pc)
    close(got, nhwc(want))
    close(ops.to_nchw(got), want)


def test_std_conv_weight_standardisation():
    from zeroshape_amd.nn import pack
    w = torch.randn(32, 16, 3, 3)
    want = F.batch_norm(w.reshape(1, 32, -1), None, None, training=True, momentum=0.0, eps=1e-8).reshape_as(w)
    close(pack.standardize_weight(w, 1e-8), want, tol=1e-6)


@pytest.mark.parametrize("C,HW,relu,res", [(64, (12, 12), True, False), (256, (7, 9), False, True),
                                           (1024, (4, 4), True, True)])
def test_group_norm(C, HW, relu, res):
    from zeroshape_amd.nn import ops
    g = torch.Generator().manual_seed(C)
    x = torch.randn(2, C, *HW, generator=g) * 3 + 1
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    r = torch.randn(2, C, *HW, generator=g) if res else None
    want = F.group_norm(x, 32, gamma, beta, 1e-5)
    if res:
        want = want + r
    if relu:
        want = F.relu(want)
    got = ops.group_norm(nhwc(x).cuda(), gamma.cuda(), beta.cuda(), 32, 1e-5, relu,
                         nhwc(r).cuda() if res else None)
    close(got, nhwc(want))


@pytest.mark.parametrize("B,C,HW,relu,res", [(9, 256, (56, 56), True, False), (5, 64, (112, 112), False, True),
                                             (6, 128, (56, 60), True, True), (14, 64, (57, 43), False, False),
                                             (3, 1024, (28, 28), True, True)])
def test_group_norm_large_tensors_take_the_coalesced_two_launch_form(B, C, HW, relu, res, monkeypatch):
    """Tensors of >= 8 MiB (the batch-28 encoder's ResNetV2 stages): zs_group_norm_nhwc_ws - per (sample, pixel chunk)
    group sums, then the normalisation, both with full-line loads - against torch, against the one-launch kernel (same
    double-precision statistics: a few ulp), reproducible run to run, ragged last chunks included."""
    from zeroshape_amd.nn import ops
    g = torch.Generator().manual_seed(C + B)
    x = torch.randn(B, C, *HW, generator=g) * 3 + 1
    assert x.numel() * 4 >= 8 << 20
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    r = torch.randn(B, C, *HW, generator=g) if res else None
    want = F.group_norm(x, 32, gamma, beta, 1e-5)
    if res:
        want = want + r
    if relu:
        want = F.relu(want)
    args = (nhwc(x).cuda(), gamma.cuda(), beta.cuda(), 32, 1e-5, relu, nhwc(r).cuda() if res else None)
    got = ops.group_norm(*args)
    close(got, nhwc(want))
    assert torch.equal(got, ops.group_norm(*args))
    monkeypatch.setattr(ops, "_group_norm_workspace", lambda device, nbytes: None)     # NULL workspace: one launch
    one = ops.group_norm(*args)
    assert float((one - got).abs().max()) <= 4e-6 * float(want.abs().max())


def test_layer_norm():
    from zeroshape_amd.nn import ops
    g = torch.Generator().manual_seed(1)
    for C in (768, 256, 100):
        x = torch.randn(3, 50, C, generator=g) * 2 + 0.5
        gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
        close(ops.layer_norm(x.cuda(), gamma.cuda(), beta.cuda(), 1e-6), F.layer_norm(x, (C,), gamma, beta, 1e-6))


@pytest.mark.parametrize("L,heads,d", [(197, 12, 64), (65, 8, 32), (300, 2, 64), (5, 1, 32)])
def test_attention(L, heads, d):
    from zeroshape_amd.nn import ops
    g = torch.Generator().manual_seed(L)
    C = heads * d
    qkv = torch.randn(2, L, 3 * C, generator=g)
    q, k, v = qkv.reshape(2, L, 3, heads, d).permute(2, 0, 3, 1, 4)
    attn = ((q * d ** -0.5) @ k.transpose(-2, -1)).softmax(-1)
    want = (attn @ v).transpose(1, 2).reshape(2, L, C)
    prev = ops.CONV_PRECISION
    try:
        for prec in ("f32", "f16x3"):              # zs_attention (fp32 MFMA) and zs_attention_split (three fp16 MFMAs)
            ops.set_conv_precision(prec)
            close(ops.attention(qkv.cuda(), heads), want)
            close(ops.attention((qkv * 4.0).cuda(), heads),
                  (((q * 4 * d ** -0.5) @ (k * 4).transpose(-2, -1)).softmax(-1) @ (v * 4)).transpose(1, 2).reshape(2, L, C))
    finally:
        ops.set_conv_precision(prev)


def test_pooling_and_resampling():
    from zeroshape_amd.nn import ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 64, 17, 20, generator=g)
    close(ops.max_pool(nhwc(x).cuda(), 3, 2, 1), nhwc(F.max_pool2d(x, 3, 2, 1)), tol=0)
    want = F.max_pool2d(same_pad(x, 3, 2, value=float("-inf")), 3, 2)
    close(ops.max_pool(nhwc(x).cuda(), 3, 2, "same"), nhwc(want), tol=0)
    close(ops.global_mean(nhwc(x).cuda()), x.mean((2, 3)), tol=1e-6)
    want = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    close(ops.upsample2x(nhwc(x).cuda()), nhwc(want), tol=2e-6)
    one = torch.randn(1, 8, 1, 1, generator=g)
    close(ops.upsample2x(nhwc(one).cuda()), nhwc(F.interpolate(one, scale_factor=2, mode="bilinear",
                                                               align_corners=True)), tol=1e-6)


def test_token_helpers():
    from zeroshape_amd.nn import ops
    g = torch.Generator().manual_seed(5)
    feat, cls, pos = torch.randn(2, 196, 768, generator=g), torch.randn(768, generator=g), \
        torch.randn(197, 768, generator=g)
    tok = ops.assemble_tokens(feat.cuda(), cls.cuda(), pos.cuda())
    want = torch.cat([cls.expand(2, 1, 768), feat], 1) + pos
    close(tok, want, tol=0)
    cat = ops.readout_concat(tok)
    want2 = torch.cat([want[:, 1:], want[:, :1].expand(-1, 196, -1)], -1)
    close(cat, want2, tol=0)


def test_bad_arguments_fail_loudly():
    from zeroshape_amd import _lib
    from zeroshape_amd.nn import ops, pack
    pc = pack.pack_conv(torch.randn(8, 6, 1, 1).repeat(1, 1, 1, 1)[:, :4], None).to("cuda")
    with pytest.raises(AssertionError):
        ops.conv2d(torch.zeros(1, 4, 4, 8, device="cuda"), pc)
    with pytest.raises(ValueError):
        ops.conv2d(torch.zeros(1, 4, 4, 4), pc)
    lib = _lib.load()
    assert lib.zs_conv2d_nhwc(None, None, None, None, None, None, None, 1, 4, 4, 6, 4, 4, 8, 1, 1, 1, 0, 0, 0, 1.0,
                              0.0, 0, None) == 0
    assert b"multiple of 4" in lib.zs_last_error()
    assert lib.zs_attention(None, None, 1, 10, 2, 48, None) == 0


def test_conv2d_split_fp16_range():
    """ZS_CONV_F16X3 carries 2e-4 <~ |x| < 65520 at ~2^-22 (two fp16 halves, round to nearest even):
    large activations keep the 2e-5 parity; beyond, hi is +-inf and every output that touches such a value is
    inf / nan - loud, never a finite wrong number (rounds 1-2 truncated and saturated silently); a tensor that is
    small as a whole (|x| ~ 1e-3) has its lo halves in fp16's subnormals and degrades to ~3e-5 relative - the
    exact kernels (ZS_ENCODER_PRECISION=f32) are the documented alternative for such data."""
    from zeroshape_amd.nn import ops, pack
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 64, 12, 12, generator=g)
    w = torch.randn(96, 64, 3, 3, generator=g) / 24
    pc = pack.pack_conv(w, None, padding=1).to("cuda")
    prev = ops.CONV_PRECISION
    try:
        ops.set_conv_precision("f16x3")
        for s in (1e-1, 1.0, 1e3, 1.2e4):                      # |x| up to ~5e4
            got = ops.conv2d(nhwc(x * s).cuda(), pc)
            close(got, nhwc(F.conv2d(x * s, w, None, padding=1)))
        close(ops.conv2d(nhwc(x * 1e-3).cuda(), pc), nhwc(F.conv2d(x * 1e-3, w, None, padding=1)), tol=1e-4)
        big = ops.conv2d(nhwc(x * 1e7).cuda(), pc)
        assert not bool(torch.isfinite(big).any())
        ops.set_conv_precision("f32")
        close(ops.conv2d(nhwc(x * 1e7).cuda(), pc), nhwc(F.conv2d(x * 1e7, w, None, padding=1)))
    finally:
        ops.set_conv_precision(prev)


@pytest.mark.parametrize("B,Cin,H,W,Cout,k", [(1, 1024, 14, 14, 256, 1), (1, 768, 7, 7, 768, 3), (1, 256, 14, 14, 2304, 1),
                                             (1, 64, 5, 5, 40, 3), (4, 768, 14, 14, 768, 1)])
def test_conv2d_split_k_across_workgroups(B, Cin, H, W, Cout, k, monkeypatch):
    """zs_conv2d_nhwc_ws: K split across workgroups for launches of fewer than 256 workgroups (partials
    through the workspace, summed in split order by a second kernel) - same result as the single-kernel
    path to the last bits of the summation order, bit-reproducible, fused epilogue included."""
    from zeroshape_amd.nn import ops, pack
    g = torch.Generator().manual_seed(Cin + Cout)
    x = nhwc(torch.randn(B, Cin, H, W, generator=g)).cuda()
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    pc = pack.pack_conv(w, b, stride=1, padding=k // 2).to("cuda")
    res = torch.randn(B, H, W, Cout, generator=g).cuda()
    want = nhwc(F.relu(F.conv2d(x.permute(0, 3, 1, 2).cpu(), w, b, padding=k // 2) + res.permute(0, 3, 1, 2).cpu()))
    prev = ops.CONV_PRECISION
    try:
        for prec in ("f32", "f16x3"):
            ops.set_conv_precision(prec)
            monkeypatch.setattr(ops, "SPLIT_K", True)
            a1 = ops.conv2d(x, pc, res1=res, act=ops.ACT_RELU)
            a2 = ops.conv2d(x, pc, res1=res, act=ops.ACT_RELU)
            monkeypatch.setattr(ops, "SPLIT_K", False)
            single = ops.conv2d(x, pc, res1=res, act=ops.ACT_RELU)
            assert torch.equal(a1, a2)
            close(a1, want)
            assert float((a1 - single).abs().max()) < 2e-5 * max(1.0, float(single.abs().max()))
    finally:
        ops.set_conv_precision(prev)


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,stride,in_relu", [
    (28, 768, 14, 14, 768, 1, 1, False), (6, 3072, 14, 14, 768, 1, 1, False), (3, 64, 56, 56, 256, 3, 1, False),
    (2, 32, 57, 41, 200, 3, 2, False), (5, 20, 31, 33, 130, 5, 1, False), (1, 256, 64, 64, 128, 1, 1, False),
    (8, 256, 28, 28, 320, 3, 1, True), (3, 128, 40, 36, 256, 3, 2, False), (28, 768, 14, 14, 2304, 1, 1, False)])
def test_conv2d_stream_k(B, Cin, H, W, Cout, k, stride, in_relu, monkeypatch):
    """ZS_CONV_STREAM_K: the large-tile kernels as a fixed number of workgroups sharing the (tile, k-step) space
    (128 x 128 tiles over 768 workgroups; 256 x 256 tiles over 256 for layers with >= 192 output channels); tiles
    cut across workgroups are summed in k order by whichever arrives last.  Same result as one workgroup per tile
    up to the summation order, bit-reproducible over repeated launches (counters return to zero), every operand
    path (pointwise, tap-major with / without input ReLU, generic taps; ragged M and Cout), fused epilogue."""
    from zeroshape_amd.nn import ops, pack
    g = torch.Generator().manual_seed(Cin + Cout + k)
    xc = torch.randn(B, Cin, H, W, generator=g)
    x = nhwc(xc).cuda()
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    pc = pack.pack_conv(w, b, stride=stride, padding=k // 2).to("cuda")
    ref = F.conv2d(F.relu(xc) if in_relu else xc, w, b, stride=stride, padding=k // 2)
    res = torch.randn(ref.shape, generator=g)
    want = nhwc(F.relu(ref + res))
    res = nhwc(res).cuda()
    prev = ops.CONV_PRECISION
    try:
        for prec in ("f32", "f16x3"):
            ops.set_conv_precision(prec)
            monkeypatch.setattr(ops, "STREAM_K", "always")
            runs = [ops.conv2d(x, pc, res1=res, act=ops.ACT_RELU, in_relu=in_relu, tiling="large") for _ in range(4)]
            monkeypatch.setattr(ops, "STREAM_K", False)
            single = ops.conv2d(x, pc, res1=res, act=ops.ACT_RELU, in_relu=in_relu, tiling="large")
            for r in runs[1:]:
                assert torch.equal(runs[0], r)
            close(runs[0], want)
            assert float((runs[0] - single).abs().max()) < 2e-5 * max(1.0, float(single.abs().max()))
        assert int(ops.splitk_workspace(x.device)[:1 << 18].view(torch.int32).abs().max()) == 0    # counters at rest
    finally:
        ops.set_conv_precision(prev)


def test_conv2d_random_shape_sweep():
    """tools/fuzz_conv.py: 80 random layer shapes (kernel 1 / 3 / 5, stride 1 / 2, ragged M and Cout, 4..320 input
    channels, input ReLU, three activations, forced tilings, stream-K on every other case) - split-fp16 against the
    exact-fp32 kernels at 2e-5 of the output scale, the fp32 kernels against torch CPU on every tenth case."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_conv.py"), "80", "11"], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "80 cases ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


PATCH_CASES = [
    # B, Cin, H, W, Cout, in_relu, residual, act     (3 x 3, stride 1, pad 1: the input-patch kernels of nn_conv_patch.h)
    (3, 128, 40, 40, 32, False, False, "relu"),       # DPT head layer geometry: tiles overhang the image (40 = 2.5 x 16)
    (2, 16, 9, 23, 7, True, True, "none"),            # one slab, ragged Cout, image smaller than a tile in one direction
    (1, 64, 16, 16, 32, False, True, "none"),         # exactly one tile: every halo pixel is padding
    (2, 48, 33, 17, 20, True, False, "relu"),
    # more than 32 output channels, >= 192 workgroups: the 128-column kernel (8 x 16-pixel tiles, weights streamed per tap)
    (8, 32, 48, 48, 256, False, False, "relu"),
    (6, 48, 41, 37, 160, True, True, "none"),         # ragged tiles in both directions, second column tile 32 of 128 wide
    (50, 16, 16, 24, 64, False, True, "relu"),        # one slab: prologue and the last-slab path only
    (3, 256, 56, 56, 256, True, False, "none"),       # the DPT fusion layer: 16 slabs
    # fewer than 192 workgroups: the slabs are split across blockIdx.z, partial tiles summed by the reduce kernel
    (28, 256, 14, 14, 256, True, True, "relu"),       # 112 workgroups x 4 ranges of 4 slabs
    (1, 256, 56, 56, 256, False, False, "none"),      # 56 workgroups x 7 uneven ranges
    (4, 64, 14, 14, 200, False, True, "relu"),        # 16 workgroups x 2 ranges, ragged column tile
]


@pytest.mark.parametrize("B,Cin,H,W,Cout,in_relu,residual,act", PATCH_CASES)
def test_conv3x3_input_patch_kernels(B, Cin, H, W, Cout, in_relu, residual, act, monkeypatch):
    """3 x 3 stride-1 layers in split-fp16 take the input-patch kernels (the tile's input patch staged once per
    16-channel slab, split in place, read at nine shifted offsets): same 2e-5 parity against torch as the GEMM kernels,
    agreement with the GEMM kernel (ZS_CONV_FORCE_LARGE bypasses the patch path) to summation order, bit-reproducible."""
    from zeroshape_amd.nn import ops, pack
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g) if residual else None
    want = F.conv2d(F.relu(x) if in_relu else x, w, b, padding=1)
    if residual:
        want = want + res
    if act == "relu":
        want = F.relu(want)
    pc = pack.pack_conv(w, b, stride=1, padding=1).to("cuda")
    kw = dict(res1=None if res is None else nhwc(res).cuda(), act=ops.ACT_RELU if act == "relu" else ops.ACT_NONE, in_relu=in_relu)
    prev = ops.CONV_PRECISION
    try:
        ops.set_conv_precision("f16x3")
        xg = nhwc(x).cuda()
        got = ops.conv2d(xg, pc, **kw)
        close(got, nhwc(want))
        assert torch.equal(got, ops.conv2d(xg, pc, **kw))
        gemm = ops.conv2d(xg, pc, tiling="large", **kw)
        close(got, gemm.cpu(), tol=5e-6)
    finally:
        ops.set_conv_precision(prev)


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,stride,in_relu", [
    (28, 768, 7, 7, 768, 3, 1, False),        # intrinsics head at batch 28: 66 tiles x 432 k-steps -> 6 ranges
    (6, 2048, 14, 14, 512, 1, 1, False),      # pointwise, 40 tiles x 128 k-steps -> 5 ranges
    (13, 256, 28, 28, 256, 3, 2, True),       # stride 2, input ReLU, 40 tiles x 144 k-steps -> 6 ranges
    (3, 512, 9, 11, 130, 3, 1, False),        # ragged rows and columns, 6 tiles: below the 40-tile floor, the small-tile kernel
])
def test_conv2d_long_k_layers_split_the_contraction(B, Cin, H, W, Cout, k, stride, in_relu):
    """Layers of few 128 x 128 tiles and >= 96 k-steps run the LDS-DMA kernel with the k-steps split across blockIdx.z
    (partial tiles through the workspace, summed in order by the reduce kernel with the fused epilogue): parity with
    torch, agreement with the small-tile kernel, bit-reproducible."""
    from zeroshape_amd.nn import ops, pack
    g = torch.Generator().manual_seed(Cin + Cout + B)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    want = F.conv2d(F.relu(x) if in_relu else x, w, b, stride=stride, padding=k // 2)
    res = torch.randn(want.shape, generator=g)
    want = F.relu(want + res)
    pc = pack.pack_conv(w, b, stride=stride, padding=k // 2).to("cuda")
    prev = ops.CONV_PRECISION
    try:
        ops.set_conv_precision("f16x3")
        xg, rg = nhwc(x).cuda(), nhwc(res).cuda()
        got = ops.conv2d(xg, pc, res1=rg, act=ops.ACT_RELU, in_relu=in_relu)
        close(got, nhwc(want))
        assert torch.equal(got, ops.conv2d(xg, pc, res1=rg, act=ops.ACT_RELU, in_relu=in_relu))
        close(got, ops.conv2d(xg, pc, res1=rg, act=ops.ACT_RELU, in_relu=in_relu, tiling="small").cpu(), tol=5e-6)
    finally:
        ops.set_conv_precision(prev)


@pytest.mark.parametrize("B,Cin,H,W,Cout,k", [
    (50, 16, 16, 24, 50, 3),          # input-patch kernel, 128-column tiles, Cout % 4 = 2
    (8, 64, 56, 56, 130, 1),          # LDS-DMA GEMM kernel (392 tiles), Cout % 4 = 2
    (2, 32, 40, 40, 31, 3),           # input-patch kernel for <= 32 channels, odd Cout
    (2, 64, 20, 20, 70, 3),           # few tiles: small-tile kernel, Cout % 4 = 2
])
def test_conv2d_channel_counts_that_are_no_multiple_of_four(B, Cin, H, W, Cout, k):
    """The kernels form their products transposed and move four consecutive output channels per lane; Cout % 4 != 0
    takes their scalar epilogue: same parity, in both arithmetics and both tilings."""
    from zeroshape_amd.nn import ops, pack
    g = torch.Generator().manual_seed(Cout * 7 + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g)
    want = nhwc(F.relu(F.conv2d(x, w, b, padding=k // 2) + res))
    pc = pack.pack_conv(w, b, stride=1, padding=k // 2).to("cuda")
    prev = ops.CONV_PRECISION
    try:
        for prec in ("f32", "f16x3"):
            ops.set_conv_precision(prec)
            for tiling in (None, "large"):
                close(ops.conv2d(nhwc(x).cuda(), pc, res1=nhwc(res).cuda(), act=ops.ACT_RELU, tiling=tiling), want)
    finally:
        ops.set_conv_precision(prev)


@pytest.mark.parametrize("B,Cin,H,W,Cmid,in_relu", [(3, 128, 40, 40, 32, False), (2, 32, 19, 33, 20, True)])
def test_conv3x3_with_fused_pointwise_tail(B, Cin, H, W, Cmid, in_relu, monkeypatch):
    """zs_conv3x3_tail_nhwc (DPT's depth head: Conv 3x3 -> ReLU -> Conv 1x1 to one channel -> ReLU, clamp): one launch of
    the input-patch kernel, the intermediate map never written; parity with torch and with the two separate launches."""
    from zeroshape_amd.nn import ops, pack
    g = torch.Generator().manual_seed(Cin + Cmid)
    x = torch.randn(B, Cin, H, W, generator=g)
    w1 = torch.randn(Cmid, Cin, 3, 3, generator=g) / np.sqrt(9 * Cin)
    b1 = torch.randn(Cmid, generator=g) * 0.1
    w2 = torch.randn(1, Cmid, 1, 1, generator=g) / np.sqrt(Cmid)
    b2 = torch.randn(1, generator=g) * 0.1
    mid = F.relu(F.conv2d(F.relu(x) if in_relu else x, w1, b1, padding=1))
    want = nhwc(torch.clamp(F.relu(F.conv2d(mid, w2, b2)), max=1.0))
    pc1 = pack.pack_conv(w1, b1, stride=1, padding=1).to("cuda")
    pc2 = pack.pack_conv(w2, b2, stride=1, padding=0).to("cuda")
    prev = ops.CONV_PRECISION
    try:
        ops.set_conv_precision("f16x3")
        xg = nhwc(x).cuda()
        fused = ops.conv2d_tail(xg, pc1, pc2, act=ops.ACT_RELU, tail_act=ops.ACT_RELU_CLAMP1, in_relu=in_relu)
        assert fused.shape == (B, H, W, 1)
        close(fused, want)
        assert torch.equal(fused, ops.conv2d_tail(xg, pc1, pc2, act=ops.ACT_RELU, tail_act=ops.ACT_RELU_CLAMP1, in_relu=in_relu))
        monkeypatch.setattr(ops, "FUSE_TAIL", False)
        close(fused, ops.conv2d_tail(xg, pc1, pc2, act=ops.ACT_RELU, tail_act=ops.ACT_RELU_CLAMP1, in_relu=in_relu).cpu(), tol=5e-6)
        ops.set_conv_precision("f32")                   # the exact engine has no fused form: two launches
        close(ops.conv2d_tail(xg, pc1, pc2, act=ops.ACT_RELU, tail_act=ops.ACT_RELU_CLAMP1, in_relu=in_relu), want)
    finally:
        ops.set_conv_precision(prev)
    lib = __import__("zeroshape_amd._lib", fromlist=["load"]).load()
    assert lib.zs_conv3x3_tail_nhwc(None, None, None, None, None, 1, 16, 16, 20, 8, 16 | 128, 0, None, None, 0, None) == 0
    assert b"Cin % 16" in lib.zs_last_error()


def test_presplit_of_many_operands_in_one_launch():
    """zs_conv2d_presplit_weight_multi == zs_conv2d_presplit_weight per operand, bit for bit (three layers of different
    shapes, pair ranges that straddle the 4096-pair chunks of a workgroup)."""
    from zeroshape_amd import _lib
    from zeroshape_amd.nn import pack
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    shapes = [(40, 32, 3), (200, 64, 1), (7, 16, 3)]                      # Cout, Cin, k
    packed, single, multi, cps, prefix = [], [], [], [], [0]
    for cout, cin, k in shapes:
        pc = pack.pack_conv(torch.randn(cout, cin, k, k, generator=g), None, stride=1, padding=k // 2).to("cuda")
        sp = torch.empty_like(pc.w)
        _lib.check(lib.zs_conv2d_presplit_weight(_lib.ptr(pc.w), _lib.ptr(sp), cin, cout, k, k, None), "presplit")
        packed.append(pc.w); single.append(sp); multi.append(torch.zeros_like(pc.w))
        coutp = -(-cout // 128) * 128
        cps.append(coutp)
        prefix.append(prefix[-1] + pc.w.numel() // coutp // 16 * 2 * coutp)
    dev = packed[0].device
    src = torch.tensor([t.data_ptr() for t in packed], dtype=torch.int64).to(dev)
    dst = torch.tensor([t.data_ptr() for t in multi], dtype=torch.int64).to(dev)
    cp = torch.tensor(cps, dtype=torch.int32).to(dev)
    pre = torch.tensor(prefix, dtype=torch.int64).to(dev)
    _lib.check(lib.zs_conv2d_presplit_weight_multi(_lib.ptr(src), _lib.ptr(dst), _lib.ptr(cp), _lib.ptr(pre), len(shapes),
                                                   prefix[-1], None), "presplit multi")
    torch.cuda.synchronize()
    for a, b in zip(single, multi):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))


@pytest.mark.parametrize("B,Cin,H,W,Cmid", [(3, 128, 20, 24, 32), (2, 32, 9, 17, 20), (1, 64, 56, 56, 32)])
def test_conv3x3_tail_on_the_upsampled_input(B, Cin, H, W, Cmid):
    """ZS_CONV_IN_UPSAMPLE2: the 3x3 layer of the fused head runs on the x2 bilinear (align_corners) up-sampling of its input,
    interpolated inside the patch loader: parity with torch's interpolate -> conv -> ReLU -> 1x1 -> clamp and with the
    unfused launches (upsample2x, conv, conv)."""
    from zeroshape_amd.nn import ops, pack
    g = torch.Generator().manual_seed(Cin + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w1 = torch.randn(Cmid, Cin, 3, 3, generator=g) / np.sqrt(9 * Cin)
    b1 = torch.randn(Cmid, generator=g) * 0.1
    w2 = torch.randn(1, Cmid, 1, 1, generator=g) / np.sqrt(Cmid)
    b2 = torch.randn(1, generator=g) * 0.1
    up = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    want = nhwc(torch.clamp(F.relu(F.conv2d(F.relu(F.conv2d(up, w1, b1, padding=1)), w2, b2)), max=1.0))
    pc1 = pack.pack_conv(w1, b1, stride=1, padding=1).to("cuda")
    pc2 = pack.pack_conv(w2, b2, stride=1, padding=0).to("cuda")
    prev, prev_up = ops.CONV_PRECISION, ops.FUSE_UPSAMPLE
    try:
        ops.set_conv_precision("f16x3")
        xg = nhwc(x).cuda()
        kw = dict(act=ops.ACT_RELU, tail_act=ops.ACT_RELU_CLAMP1, upsample=True)
        fused = ops.conv2d_tail(xg, pc1, pc2, **kw)
        assert fused.shape == (B, 2 * H, 2 * W, 1)
        close(fused, want)
        assert torch.equal(fused, ops.conv2d_tail(xg, pc1, pc2, **kw))
        ops.FUSE_UPSAMPLE = False
        close(fused, ops.conv2d_tail(xg, pc1, pc2, **kw).cpu(), tol=5e-6)
    finally:
        ops.set_conv_precision(prev)
        ops.FUSE_UPSAMPLE = prev_up


@pytest.mark.parametrize("M,Cin,Cout,res,act", [
    (5516, 768, 3072, False, "gelu"),         # ViT fc1 at batch 28: 264 tiles = 256 whole + 8 tail tiles in K ranges
    (5516, 768, 2304, False, "none"),         # qkv: 198 whole tiles, no tail
    (5516, 3072, 768, True, "none"),          # fc2: 66 tiles, every one of them in K ranges
    (5488, 768, 768, True, "none"),           # proj (196-token form): ragged last row tile
    (16384, 256, 1028, True, "relu"),         # Cout % 256 = 4 (CoutPad 1152: the last column tile is half empty), 5 x 64 tiles
    (12544, 32, 512, False, "relu"),          # one stage of K = 32
])
def test_conv2d_pointwise_256_tiles(M, Cin, Cout, res, act):
    """The 256 x 256 ping-pong kernel of the large pointwise layers (csrc/nn_conv_pp256.h): parity with torch, whole tiles
    bit-identical to the 128 x 128 LDS-DMA kernel (same MFMAs in the same k order), tail tiles (K ranges summed by a second
    launch) within summation order of it, bit-reproducible."""
    from zeroshape_amd.nn import ops, pack
    g = torch.Generator().manual_seed(M + Cin + Cout)
    x = torch.randn(1, M, 1, Cin, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / np.sqrt(Cin)
    b = torch.randn(Cout, generator=g)
    r = torch.randn(1, M, 1, Cout, generator=g) if res else None
    want = F.linear(x, w[:, :, 0, 0], b)
    if res:
        want = want + r
    want = {"gelu": F.gelu, "relu": F.relu, "none": lambda t: t}[act](want)
    a = {"gelu": ops.ACT_GELU, "relu": ops.ACT_RELU, "none": ops.ACT_NONE}[act]
    pc = pack.pack_conv(w, b).to("cuda")
    prev = ops.CONV_PRECISION
    try:
        ops.set_conv_precision("f16x3")
        xg, rg = x.cuda(), (r.cuda() if res else None)
        got = ops.conv2d(xg, pc, res1=rg, act=a, tiling="tile256")
        close(got, want)
        assert torch.equal(got, ops.conv2d(xg, pc, res1=rg, act=a, tiling="tile256"))
        tiles = -(-M // 256) * -(-Cout // 256)
        if Cin >= 768 and (tiles >= 128 or Cin >= 2048):            # large layers take this kernel by themselves
            assert torch.equal(got, ops.conv2d(xg, pc, res1=rg, act=a))
        old = ops.conv2d(xg, pc, res1=rg, act=a, tiling="large")
        close(got, old.cpu(), tol=5e-6)
        tiles = -(-M // 256) * -(-Cout // 256)
        if tiles > 128 and tiles % 256 == 0 or 128 < tiles < 256:      # no tail: every tile whole
            assert torch.equal(got, old)
    finally:
        ops.set_conv_precision(prev)


@pytest.mark.parametrize("B,H,W", [(1, 14, 14), (1, 9, 7), (2, 8, 8), (3, 16, 8)])
def test_fused_group_norm_launches_directly(B, H, W):
    """zs_conv2d_nhwc_fused (statistics from a convolution's epilogue, normalisation on the consumer's operand load),
    zs_group_norm_apply_stats and zs_gn_relu_max_pool_nhwc had no unit tests of their own.  Each against the unfused ops
    (ops.group_norm on the materialised tensors) and torch, at batch 1 (any map) and batch > 1 (maps of a multiple of 32)."""
    from zeroshape_amd.nn import ops, pack
    g = torch.Generator().manual_seed(B * 100 + H)
    C0, C1, C2 = 64, 128, 256
    x = torch.randn(B, C0, H, W, generator=g)
    w1 = torch.randn(C1, C0, 1, 1, generator=g) / 8
    w2 = torch.randn(C2, C1, 3, 3, generator=g) / 34
    ga1, be1 = torch.randn(C1, generator=g) * 0.3 + 1, torch.randn(C1, generator=g) * 0.2
    ga2, be2 = torch.randn(C2, generator=g) * 0.3 + 1, torch.randn(C2, generator=g) * 0.2
    res = torch.randn(B, C2, H, W, generator=g)
    prev = ops.CONV_PRECISION
    try:
        ops.set_conv_precision("f16x3")
        xg = nhwc(x).cuda()
        assert ops.fused_ok(xg)
        p1, p2 = pack.pack_conv(w1, None).to("cuda"), pack.pack_conv(w2, None, padding=1).to("cuda")
        # producer statistics -> consumer normalises on load -> one-pass GN + residual + ReLU of the block output
        z1, st1 = ops.conv2d(xg, p1, stats_out="group")
        z2, st2 = ops.conv2d(z1, p2, gn_in=(st1, ga1.cuda(), be1.cuda(), 1e-5), stats_out="group")
        y = ops.group_norm_apply(z2, st2, ga2.cuda(), be2.cuda(), 1e-5, relu=True, residual=nhwc(res).cuda())
        # the unfused engine on the same packed layers
        u1 = ops.conv2d(xg, p1)
        close(z1, u1.cpu(), tol=1e-6)
        u2 = ops.conv2d(ops.group_norm(u1, ga1.cuda(), be1.cuda(), relu=True), p2)
        close(z2, u2.cpu(), tol=2e-5)
        close(y, ops.group_norm(u2, ga2.cuda(), be2.cuda(), relu=True, residual=nhwc(res).cuda()).cpu(), tol=2e-5)
        # torch
        t1 = F.conv2d(x, w1)
        t2 = F.conv2d(F.relu(F.group_norm(t1, 32, ga1, be1, 1e-5)), w2, padding=1)
        close(y, nhwc(F.relu(F.group_norm(t2, 32, ga2, be2, 1e-5) + res)))
        # GN + ReLU riding on the max pool
        pooled = ops.gn_relu_max_pool(z1, st1, ga1.cuda(), be1.cuda(), 3, 2, 1, 1e-5)
        close(pooled, nhwc(F.max_pool2d(F.relu(F.group_norm(t1, 32, ga1, be1, 1e-5)), 3, 2, 1)))
        close(pooled, ops.max_pool(ops.group_norm(u1, ga1.cuda(), be1.cuda(), relu=True), 3, 2, 1).cpu(), tol=2e-5)
    finally:
        ops.set_conv_precision(prev)


@pytest.mark.parametrize("B,L,heads,d", [(28, 197, 12, 64), (28, 197, 8, 32), (600, 65, 8, 32), (40, 33, 4, 64), (20, 224, 8, 64), (17, 96, 8, 32)])
def test_attention_with_keys_and_values_staged_in_lds(B, L, heads, d):
    """Many (sample, head) pairs (the ViT blocks at batch 28, the window stage's windows): attention_lds_kernel stages K and V once per
    pair in LDS, pre-split and in fragment order (csrc/nn_ops.hip).  Parity with torch, and - the same MFMAs on the same operand
    bits in the same order - bit-identical to the one-wave-per-query-tile kernel, which sub-batches of fewer than 128 pairs take."""
    from zeroshape_amd.nn import ops
    g = torch.Generator().manual_seed(B + L)
    C = heads * d
    qkv = torch.randn(B, L, 3 * C, generator=g)
    q, k, v = qkv.reshape(B, L, 3, heads, d).permute(2, 0, 3, 1, 4)
    want = (((q * d ** -0.5) @ k.transpose(-2, -1)).softmax(-1) @ v).transpose(1, 2).reshape(B, L, C)
    prev = ops.CONV_PRECISION
    try:
        ops.set_conv_precision("f16x3")
        assert B * heads >= 128
        x = qkv.cuda()
        got = ops.attention(x, heads)
        close(got, want)
        assert torch.equal(got, ops.attention(x, heads))
        step = max(1, 127 // heads)                     # sub-batches below the 128-pair threshold: the register-only kernels
        assert step * heads < 128
        parts = torch.cat([ops.attention(x[i:i + step].contiguous(), heads) for i in range(0, min(B, 4 * step) // step * step, step)], 0)
        if L > 64 and step * heads * ((L + 31) // 32) < 512:
            close(got[:parts.shape[0]], parts.cpu(), tol=2e-6)      # (few pairs and several key tiles: the key-split kernel, another association)
        else:
            assert torch.equal(got[:parts.shape[0]], parts)
    finally:
        ops.set_conv_precision(prev)


@pytest.mark.parametrize("B,H,W,C,win", [(2, 16, 24, 256, 8), (3, 8, 8, 64, 4), (1, 16, 16, 20, 8), (2, 16, 16, 768, 8)])
def test_window_tokens_quad_and_scalar_forms(B, H, W, C, win):
    """zs_window_tokens (CoordEmb's window gather, model/depth/... via nn/blocks.py): [cls | win^2 pixels of a window] + pos, masked
    pixels replaced by the invalid token - bit-equal to the torch construction, through the four-channels-per-lane kernel (C / 4
    divides 256) and the scalar one (C = 20: 5 quads do not; 768: 192 quads do not)."""
    from zeroshape_amd.nn import ops
    g = torch.Generator().manual_seed(C + win)
    emb = torch.randn(B, H, W, C, generator=g)
    mask = torch.rand(B, H, W, generator=g) > 0.3
    inv, cls, pos = torch.randn(C, generator=g), torch.randn(C, generator=g), torch.randn(win * win + 1, C, generator=g)
    got = ops.window_tokens(emb.cuda(), mask.cuda(), inv.cuda(), cls.cuda(), pos.cuda(), win)
    x = torch.where(mask[..., None], emb, inv.expand_as(emb))
    x = x.reshape(B, H // win, win, W // win, win, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, win * win, C)
    want = torch.cat([cls.expand(x.shape[0], 1, C), x], 1) + pos
    assert got.shape == want.shape and torch.equal(got.cpu(), want)
