"""GPU parity of the training kernels (include/zeroshape_hip.h "Training") through
zeroshape_amd/nn/autograd.py: every forward and backward against torch's CPU fp32 autograd of the
same op (the plain-PyTorch reference of a floating-point kernel).  Tolerances are relative to the
tensor's scale; the contract of BASELINE.json is 1e-4."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

RTOL = 2e-5


def close(got, want, rtol=RTOL, what=""):
    got, want = got.detach().cpu().double(), want.detach().double()
    scale = float(want.abs().max()) + 1e-30
    err = float((got - want).abs().max())
    assert got.shape == want.shape, what
    assert err <= rtol * scale, "%s: max err %.3e vs scale %.3e" % (what, err, scale)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("cfg", [
    # B, H, W, Cin, Cout, k, stride, pad, bias, act, in_relu, in_scale
    (2, 14, 14, 64, 96, 3, 1, 1, True, "relu", False, 1.0),
    (1, 17, 13, 32, 40, 1, 1, 0, True, None, False, 1.0),
    (2, 16, 16, 48, 64, 3, 2, 1, False, None, False, 1.0),
    (2, 15, 15, 32, 64, 1, 2, 0, False, None, False, 1.0),
    (1, 32, 32, 16, 32, 7, 2, 3, False, "relu", False, 1.0),
    (2, 12, 12, 64, 64, 3, 1, 1, True, None, True, 1.0),
    (1, 1, 300, 256, 256, 1, 1, 0, True, None, False, 0.70710678),
    (3, 1, 50, 128, 1, 1, 1, 0, True, None, False, 1.0),
    (1, 9, 9, 32, 3, 1, 1, 0, True, None, False, 1.0),
    (2, 56, 56, 32, 128, 3, 1, 1, True, "clamp1", False, 1.0),
    (2, 15, 14, 32, 48, 3, 2, "same", False, None, False, 1.0),
])
def test_conv_forward_dgrad_wgrad(cfg):
    from zeroshape_amd.nn import autograd as A
    B, H, W, Cin, Cout, k, stride, pad, use_bias, act, in_relu, in_scale = cfg
    g = torch.Generator().manual_seed(B * 1000 + H * 37 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g) if use_bias else None
    res = torch.randn(B, Cout, 1, 1, generator=g)
    # ---- torch CPU reference ----
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = None if b is None else b.clone().requires_grad_(True)
    xin = (F.relu(xr) if in_relu else xr) * in_scale
    if pad == "same":       # TF 'SAME' (timm StdConv2dSame)
        oh, ow = -(-H // stride), -(-W // stride)
        ph, pw = max((oh - 1) * stride + k - H, 0), max((ow - 1) * stride + k - W, 0)
        xin = F.pad(xin, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
        y = F.conv2d(xin, wr, br, stride=stride)
    else:
        y = F.conv2d(xin, wr, br, stride=stride, padding=pad)
    resr = res.expand_as(y).contiguous().requires_grad_(True)
    y = y + resr
    y = F.relu(y) if act == "relu" else (y.clamp(0, 1) if act == "clamp1" else y)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    # ---- HIP ----
    xg = nhwc(x).cuda().requires_grad_(True)
    wg = w.cuda().requires_grad_(True)
    bg = None if b is None else b.cuda().requires_grad_(True)
    rg = nhwc(resr.detach()).cuda().requires_grad_(True)
    code = {"relu": A.ACT_RELU, "clamp1": A.ACT_RELU_CLAMP1, None: A.ACT_NONE}[act]
    yg = A.conv2d(xg, wg, bg, stride=stride, padding=pad, act=code, in_relu=in_relu, in_scale=in_scale, res1=rg)
    close(yg.permute(0, 3, 1, 2), y, what="forward")
    yg.backward(nhwc(gy).cuda())
    close(xg.grad.permute(0, 3, 1, 2), xr.grad, what="dgrad")
    close(wg.grad, wr.grad, what="wgrad")
    close(rg.grad.permute(0, 3, 1, 2), resr.grad, what="dres")
    if b is not None:
        close(bg.grad, br.grad, what="dbias")


def test_conv_weight_column_ranges_and_padded_input():
    """The decoder's skip layers: one [256, 515] weight applied as three column-range products
    (x | xyz | feat) / sqrt(2); xyz has 3 channels padded to 4."""
    from zeroshape_amd.nn import autograd as A
    g = torch.Generator().manual_seed(5)
    n, C = 333, 256
    xs, pts, feat = torch.randn(1, n, C, generator=g), torch.randn(1, n, 3, generator=g), torch.randn(1, n, C, generator=g)
    w, b = torch.randn(C, 2 * C + 3, generator=g) / 16, torch.randn(C, generator=g)
    r2 = 0.7071067811865476
    xr, fr, wr, br = [t.clone().requires_grad_(True) for t in (xs, feat, w, b)]
    y = F.linear(torch.cat([xr, pts, fr], -1) / np.sqrt(2), wr, br)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xg, fg, wg, bg = [t.cuda().requires_grad_(True) for t in (xs, feat, w, b)]
    p4 = A._pad_channels(pts.cuda(), 4)
    yg = A.linear(p4, wg, None, in_scale=r2, cin0=C, cin=3)
    yg = A.linear(fg, wg, None, in_scale=r2, res1=yg, cin0=C + 3, cin=C)
    yg = A.linear(xg, wg, bg, in_scale=r2, res1=yg, cin0=0, cin=C)
    close(yg, y, what="forward")
    yg.backward(gy.cuda())
    close(xg.grad, xr.grad, what="dx")
    close(fg.grad, fr.grad, what="dfeat")
    close(wg.grad, wr.grad, what="dW")
    close(bg.grad, br.grad, what="db")


def test_std_conv_weight_standardisation_gradient():
    """timm StdConv2dSame (eps 1e-8): gradient flows through the per-channel standardisation."""
    from zeroshape_amd.nn import autograd as A
    g = torch.Generator().manual_seed(9)
    x, w = torch.randn(2, 32, 10, 10, generator=g), torch.randn(48, 32, 3, 3, generator=g) * 0.3 + 0.1
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    flat = wr.reshape(48, -1)
    ws = ((flat - flat.mean(1, keepdim=True)) / torch.sqrt(flat.var(1, unbiased=False, keepdim=True) + 1e-8)).reshape(w.shape)
    y = F.conv2d(xr, ws, None, padding=1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xg, wg = nhwc(x).cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    yg = A.conv2d(xg, wg, None, padding="same", std_eps=1e-8)
    close(yg.permute(0, 3, 1, 2), y, what="forward")
    yg.backward(nhwc(gy).cuda())
    close(xg.grad.permute(0, 3, 1, 2), xr.grad, what="dx")
    close(wg.grad, wr.grad, rtol=1e-4, what="dW")


@pytest.mark.parametrize("rows,C", [(37, 256), (1000, 768), (5, 64), (200, 1024)])
def test_layer_norm_backward(rows, C):
    from zeroshape_amd.nn import autograd as A
    g = torch.Generator().manual_seed(rows)
    x, ga, be = torch.randn(rows, C, generator=g) * 2 + 0.5, torch.randn(C, generator=g), torch.randn(C, generator=g)
    xr, gr, br = [t.clone().requires_grad_(True) for t in (x, ga, be)]
    y = F.layer_norm(xr, (C,), gr, br, 1e-6)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xg, gg, bg = [t.cuda().requires_grad_(True) for t in (x, ga, be)]
    yg = A.layer_norm(xg, gg, bg, 1e-6)
    close(yg, y, what="forward")
    yg.backward(gy.cuda())
    close(xg.grad, xr.grad, what="dx")
    close(gg.grad, gr.grad, what="dgamma")
    close(bg.grad, br.grad, what="dbeta")


@pytest.mark.parametrize("name", ["gelu", "softplus", "relu"])
def test_activations(name):
    from zeroshape_amd.nn import autograd as A
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4000, generator=g) * (0.3 if name == "softplus" else 3)
    x[:5] = torch.tensor([0.0, 0.19, 0.21, -0.5, 5.0])
    xr = x.clone().requires_grad_(True)
    y = {"gelu": F.gelu, "softplus": lambda t: F.softplus(t, beta=100), "relu": F.relu}[name](xr)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xg = x.cuda().requires_grad_(True)
    yg = {"gelu": A.gelu, "softplus": lambda t: A.softplus(t, 100.0), "relu": A.relu}[name](xg)
    close(yg, y, what="forward")
    yg.backward(gy.cuda())
    close(xg.grad, xr.grad, what="backward")


@pytest.mark.parametrize("B,L,heads,d", [(2, 197, 12, 64), (3, 65, 8, 32), (1, 197, 8, 32), (2, 101, 12, 64)])
def test_attention_backward(B, L, heads, d):
    from zeroshape_amd.nn import autograd as A
    g = torch.Generator().manual_seed(L)
    C = heads * d
    qkv = torch.randn(B, L, 3 * C, generator=g)
    qr = qkv.clone().requires_grad_(True)
    q, k, v = qr.reshape(B, L, 3, heads, d).permute(2, 0, 3, 1, 4).unbind(0)
    o = ((q @ k.transpose(-2, -1)) * d ** -0.5).softmax(-1) @ v
    o = o.transpose(1, 2).reshape(B, L, C)
    go = torch.randn(o.shape, generator=g)
    o.backward(go)
    qg = qkv.cuda().requires_grad_(True)
    og = A.attention(qg, heads)
    close(og, o, what="forward")
    og.backward(go.cuda())
    close(qg.grad, qr.grad, what="dqkv")


@pytest.mark.parametrize("B,M,Ll", [(2, 300, 197), (1, 1000, 197), (3, 7, 50), (1, 4099, 197)])
def test_point_attention(B, M, Ll):
    """ImplFuncAttention's point rows (implicit.py:44-66) incl. the gradient to the latent k / v."""
    from zeroshape_amd.nn import autograd as A
    heads, d = 8, 32
    C = heads * d
    g = torch.Generator().manual_seed(M)
    qp, ql = torch.randn(B, M, 3 * C, generator=g), torch.randn(B, Ll, 3 * C, generator=g)
    pr, lr = qp.clone().requires_grad_(True), ql.clone().requires_grad_(True)
    sp = lambda t, n: t.reshape(B, n, 3, heads, d).permute(2, 0, 3, 1, 4).unbind(0)   # noqa: E731
    q_p, k_p, v_p = sp(pr, M)
    _, k_l, v_l = sp(lr, Ll)
    cross = (q_p @ k_l.transpose(-2, -1)) * d ** -0.5
    self_ = (q_p * k_p).sum(-1, keepdim=True) * d ** -0.5
    joint = torch.cat([cross, self_], -1).softmax(-1)
    o = (joint[..., :Ll] @ v_l + joint[..., Ll:] * v_p).transpose(1, 2).reshape(B, M, C)
    go = torch.randn(o.shape, generator=g)
    o.backward(go)
    pg, lg = qp.cuda().requires_grad_(True), ql.cuda().requires_grad_(True)
    og = A.point_attention(pg, lg, heads)
    close(og, o, what="forward")
    og.backward(go.cuda())
    close(pg.grad, pr.grad, what="dqkv_points")
    close(lg.grad, lr.grad, what="dqkv_latent")
    assert float(lg.grad[..., :C].abs().max()) == 0          # latent queries are unused here


def test_drop_path_residual_and_bce_loss():
    from zeroshape_amd.nn import autograd as A
    from oracle import decoder_ref as R
    g = torch.Generator().manual_seed(1)
    x, br = torch.randn(3, 40, 16, generator=g), torch.randn(3, 40, 16, generator=g)
    sc = torch.tensor([0.0, 1.0 / 0.9, 1.0 / 0.9])
    xr, brr = x.clone().requires_grad_(True), br.clone().requires_grad_(True)
    y = xr + brr * sc.view(3, 1, 1)
    logits = y.sum(-1) * 0.5
    sdf = torch.randn(3, 40, generator=g) * 0.02
    loss = R.shape_loss(logits, sdf, 0.01, 2.5) * 3.0
    loss.backward()
    xg, bg = x.cuda().requires_grad_(True), br.cuda().requires_grad_(True)
    lg = A.add_scaled_rows(xg, bg, sc.cuda()).sum(-1) * 0.5
    lossg = A.bce_logits(lg, sdf.cuda(), 0.01, 2.5) * 3.0
    assert abs(float(lossg) - float(loss)) < 1e-6 * abs(float(loss))
    lossg.backward()
    close(xg.grad, xr.grad, what="dx")
    close(bg.grad, brr.grad, what="dbranch")


def test_adamw_multi_tensor_matches_torch():
    from zeroshape_amd.optim import FusedAdamW
    g = torch.Generator().manual_seed(2)
    shapes = [(300, 70), (70,), (5, 5, 3, 3), (40000,), (1,)]
    ref = [torch.randn(s, generator=g).requires_grad_(True) for s in shapes]
    mine = [p.detach().clone().cuda().requires_grad_(True) for p in ref]
    groups = lambda ps: [dict(params=ps[:2], lr=3e-3, weight_decay=0.0), dict(params=ps[2:], lr=1e-3, weight_decay=0.05)]  # noqa: E731
    o_ref = torch.optim.AdamW(groups(ref), betas=(0.9, 0.95))
    o_mine = FusedAdamW(groups(mine), betas=(0.9, 0.95))
    for step in range(4):
        for pr, pm in zip(ref, mine):
            gr = torch.randn(pr.shape, generator=g)
            pr.grad, pm.grad = gr.clone(), gr.cuda()
        o_ref.step()
        o_mine.step()
        o_ref.zero_grad()
        o_mine.zero_grad()
    for pr, pm in zip(ref, mine):
        close(pm, pr, rtol=1e-6, what="param")
    sd = o_mine.state_dict()
    assert sd["state"][0]["step"] == 4 and sd["state"][0]["exp_avg"].shape == (300, 70)


@pytest.mark.parametrize("cfg", [
    # B, H, W, Cin, Cout, k, stride, pad, bias, in_relu, in_scale      (tile 128 / 64, pointwise / fast / generic loaders)
    (4, 56, 56, 64, 256, 1, 1, 0, True, False, 1.0),
    (4, 14, 14, 256, 256, 3, 1, 1, False, False, 1.0),
    (2, 28, 28, 128, 128, 3, 2, 1, True, False, 1.0),
    (4, 1, 197, 768, 3072, 1, 1, 0, True, False, 1.0),
    (2, 12, 12, 64, 64, 3, 1, 1, True, True, 0.70710678),
    (3, 5, 7, 36, 20, 3, 1, 1, False, False, 1.0),
    (1, 9, 9, 32, 3, 1, 1, 0, True, False, 1.0),
])
def test_weight_gradient_in_split_fp16(cfg, monkeypatch):
    """zs_conv2d_wgrad with ZS_CONV_F16X3 (wgrad_split_kernel: the loader splits into fp16 halves and stores pixel pairs,
    three 16-bit MFMAs per 16 pixels) against torch autograd and against the fp32-MFMA kernel: the operand error of the
    split is 2^-22, so the two kernels agree to a few 1e-7 of the gradient's scale; ragged channel counts, partial pixel
    splits, strides, padding and the input transform included; the bias gradient is the same fp32 sum in both."""
    from zeroshape_amd.nn import autograd as A
    B, H, W, Cin, Cout, k, stride, pad, use_bias, in_relu, in_scale = cfg
    g = torch.Generator().manual_seed(B * 1000 + H * 37 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g) if use_bias else None
    xr, wr = x.clone(), w.clone().requires_grad_(True)
    br = None if b is None else b.clone().requires_grad_(True)
    y = F.conv2d((F.relu(xr) if in_relu else xr) * in_scale, wr, br, stride=stride, padding=pad)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    grads = {}
    for prec in ("f32", "f16x3"):
        monkeypatch.setattr(A, "BWD_WGRAD_PRECISION", prec)
        xg = nhwc(x).cuda()
        wg = w.cuda().requires_grad_(True)
        bg = None if b is None else b.cuda().requires_grad_(True)
        yg = A.conv2d(xg, wg, bg, stride=stride, padding=pad, in_relu=in_relu, in_scale=in_scale)
        yg.backward(nhwc(gy).cuda())
        grads[prec] = (wg.grad.clone(), None if bg is None else bg.grad.clone())
        close(wg.grad, wr.grad, what="wgrad " + prec)
        if b is not None:
            close(bg.grad, br.grad, what="dbias " + prec)
    scale = float(wr.grad.abs().max())
    assert float((grads["f32"][0] - grads["f16x3"][0]).abs().max()) <= 3e-6 * scale
    if b is not None:
        assert torch.equal(grads["f32"][1], grads["f16x3"][1])


def test_overflow_steps_do_not_count_as_adamw_steps():
    """GradScaler semantics: on an overflow torch never calls optimizer.step(), so neither the bias
    correction nor the checkpointed `step` advance.  Two of five LossScaler steps carry an inf gradient: parameters
    and state equal torch.optim.AdamW stepped three times on the clean gradients."""
    from zeroshape_amd.optim import FusedAdamW, LossScaler
    g = torch.Generator().manual_seed(4)
    ref = [torch.randn(s, generator=g).requires_grad_(True) for s in ((33, 17), (17,), (20000,))]
    mine = [p.detach().clone().cuda().requires_grad_(True) for p in ref]
    o_ref = torch.optim.AdamW([dict(params=ref, lr=2e-3, weight_decay=0.05)], betas=(0.9, 0.95))
    o_mine = FusedAdamW([dict(params=mine, lr=2e-3, weight_decay=0.05)], betas=(0.9, 0.95))
    scaler = LossScaler("cuda", init_scale=1024.0)
    scales = []
    for step in range(5):
        overflow = step in (0, 3)
        scale = float(scaler.scale)
        scales.append(scale)
        for pr, pm in zip(ref, mine):
            gr = torch.randn(pr.shape, generator=g)
            pm.grad = gr.cuda() * scale                       # what backward of the scaled loss leaves
            if overflow:
                pm.grad.view(-1)[0] = float("inf")
            else:
                pr.grad = gr.clone()
        scaler.step(o_mine)
        if not overflow:
            o_ref.step()
        o_ref.zero_grad()
        o_mine.zero_grad()
    assert scales == [1024.0, 512.0, 512.0, 512.0, 256.0]     # halved after each overflow
    for pr, pm in zip(ref, mine):
        close(pm, pr, rtol=1e-6, what="param")
    sd = o_mine.state_dict()
    assert sd["state"][0]["step"] == 3
    close(sd["state"][0]["exp_avg"], o_ref.state_dict()["state"][0]["exp_avg"], rtol=1e-5, what="exp_avg")
    # a reloaded state continues from the applied count
    o2 = FusedAdamW([dict(params=mine, lr=2e-3, weight_decay=0.05)], betas=(0.9, 0.95))
    o2.load_state_dict(sd)
    assert o2.state_dict()["state"][0]["step"] == 3


@pytest.mark.parametrize("k,stride,pad,H,Cout,std", [(7, 2, 3, 32, 64, False), (7, 2, "same", 30, 64, True), (3, 1, 1, 9, 32, False)])
def test_stem_data_gradient_small_cin(k, stride, pad, H, Cout, std):
    """Convolutions with 3 input channels (zero-padded to 4): the data gradient takes the direct
    gather kernel (zs_conv2d_dgrad_small_cin), incl. TF-'SAME' padding and standardised weights."""
    from zeroshape_amd.nn import autograd as A
    g = torch.Generator().manual_seed(k + H)
    x = torch.randn(2, 3, H, H + 2, generator=g)
    w = torch.randn(Cout, 3, k, k, generator=g) * 0.2 + 0.05
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ws = wr
    if std:
        flat = wr.reshape(Cout, -1)
        ws = ((flat - flat.mean(1, keepdim=True)) / torch.sqrt(flat.var(1, unbiased=False, keepdim=True) + 1e-8)).reshape(w.shape)
    xin = xr * 0.5
    if pad == "same":
        Hh, Ww = x.shape[2:]
        ph = max((-(-Hh // stride) - 1) * stride + k - Hh, 0)
        pw = max((-(-Ww // stride) - 1) * stride + k - Ww, 0)
        y = F.conv2d(F.pad(xin, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2)), ws, None, stride=stride)
    else:
        y = F.conv2d(xin, ws, None, stride=stride, padding=pad)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    x4 = torch.zeros(2, H, H + 2, 4)
    x4[..., :3] = nhwc(x)
    xg, wg = x4.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    yg = A.conv2d(xg, wg, None, stride=stride, padding=pad, in_scale=0.5, std_eps=1e-8 if std else None)
    close(yg.permute(0, 3, 1, 2), y, what="forward")
    yg.backward(nhwc(gy).cuda())
    close(xg.grad[..., :3].permute(0, 3, 1, 2), xr.grad, what="dx")
    assert float(xg.grad[..., 3].abs().max()) == 0
    close(wg.grad, wr.grad, rtol=1e-4, what="dW")


def test_repack_of_all_operands_in_one_launch_equals_the_single_packs():
    """zs_pack_conv_weight_multi (the re-pack after an optimiser step: LDS tiles for kernels up to 3x3, element chunks
    otherwise) writes exactly what zs_pack_conv_weight writes, forward and data-gradient operands, channel
    sub-ranges, ragged channel counts."""
    from zeroshape_amd import _lib
    from zeroshape_amd.nn import autograd as A
    g = torch.Generator().manual_seed(5)
    shapes = [(3072, 768, 1, 1), (768, 3072, 1, 1), (256, 256, 3, 3), (64, 3, 7, 7), (10, 6, 3, 3), (130, 70, 1, 1),
              (32, 128, 3, 3), (1, 32, 1, 1), (96, 67, 2, 2)]
    weights = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]
    keys = []
    for w in weights:
        for dgrad in (False, True):
            keys.append((w, 0, w.shape[1], dgrad))
    sub = weights[2]
    keys.append((sub, 64, 128, False))           # channels [64, 192) of 256, like the decoder's concatenated inputs
    keys.append((sub, 64, 128, True))
    first = [A._pack(w, c0, c, d).clone() for w, c0, c, d in keys]
    with torch.no_grad():
        for w in weights:
            w.copy_(torch.randn(w.shape, generator=g).cuda())
    A.bump_generation()
    again = [A._pack(w, c0, c, d) for w, c0, c, d in keys]          # the first call re-packs every stale operand
    lib = _lib.load()
    for (w, c0, c, d), old, new in zip(keys, first, again):
        want = torch.empty_like(new)
        kh, kw = w.shape[2], w.shape[3]
        with torch.cuda.device(w.device):
            _lib.check(lib.zs_pack_conv_weight(_lib.ptr(w.detach()), _lib.ptr(want), w.shape[0], c, c0, w.shape[1], kh, kw,
                                               1 if d else 0, _lib.current_stream_ptr(w.device)), "zs_pack_conv_weight")
        assert not torch.equal(old, new)
        assert torch.equal(new, want), (tuple(w.shape), c0, c, d)


@pytest.mark.gpu
def test_standardize_all_equals_the_single_launches():
    """zs_standardize_weight_multi (one launch for every stale StdConv weight after an optimiser step) writes exactly what
    zs_standardize_weight writes per weight; fresh weights are skipped, a repeated weight is standardised once."""
    from zeroshape_amd.nn import autograd as A
    A.clear_pack_cache()
    g = torch.Generator().manual_seed(9)
    shapes = [(64, 3, 7, 7), (256, 64, 1, 1), (64, 64, 3, 3), (1024, 512, 1, 1), (10, 6, 3, 3), (1, 32, 1, 1)]
    ws = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]
    want = [A.standardize(w, 1e-8).clone() for w in ws]                    # single launches
    with torch.no_grad():
        for w in ws:
            w.mul_(1.5).add_(0.25)
    A.bump_generation()
    want2 = []
    for w in ws:
        m = w.detach().reshape(w.shape[0], -1)
        want2.append(((m - m.mean(1, keepdim=True)) / torch.sqrt(m.var(1, unbiased=False, keepdim=True) + 1e-8)).reshape(w.shape))
    A.standardize_all([(w, 1e-8) for w in ws] + [(ws[1], 1e-8)])           # one launch (ws[1] listed twice)
    for w, old, ref in zip(ws, want, want2):
        got = A._STD[id(w)][2]
        assert A._STD[id(w)][3] == A._stamp(w)
        assert not torch.equal(got, old) or w.shape[0] == 1
        single = torch.empty_like(got)
        from zeroshape_amd import _lib
        with torch.cuda.device(w.device):
            _lib.check(_lib.load().zs_standardize_weight(_lib.ptr(w.detach()), _lib.ptr(single), w.shape[0], w[0].numel(), 1e-8,
                                                         _lib.current_stream_ptr(w.device)), "zs_standardize_weight")
        assert torch.equal(got, single), tuple(w.shape)
        close(got, ref.cpu(), rtol=1e-5, what="standardised weight")
    A.clear_pack_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("fwd,bwd", [("f16x3", "f16x3"), ("f16x3", "f32")])
def test_repack_writes_the_fp16_halves_itself(fwd, bwd, monkeypatch):
    """optim.amp: zs_pack_conv_weight_multi_split writes the halves of every operand whose tiles hold whole K = 16 groups in
    the launch that re-packs them - bit for bit what zs_conv2d_presplit_weight makes of the single pack - and, when forward
    AND data gradients read the halves, leaves the fp32 operand of those entries alone; the others (3-channel stem, ragged
    channel counts, 7x7) are split by the launch behind it as before."""
    from zeroshape_amd import _lib
    from zeroshape_amd.nn import autograd as A
    A.clear_pack_cache()
    A.set_forward_precision(fwd)
    A.set_backward_precision(bwd)
    try:
        assert A._inline_split_mode() == (2 if bwd == "f16x3" else 1)
        g = torch.Generator().manual_seed(6)
        shapes = [(3072, 768, 1, 1), (768, 3072, 1, 1), (256, 256, 3, 3), (64, 3, 7, 7), (10, 6, 3, 3), (130, 70, 1, 1),
                  (32, 128, 3, 3), (48, 32, 1, 1), (96, 64, 2, 2)]
        weights = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]
        keys = [(w, 0, w.shape[1], d) for w in weights for d in (False, True)]
        keys += [(weights[2], 64, 128, False), (weights[2], 64, 128, True)]
        first = [A._pack(w, c0, c, d).clone() for w, c0, c, d in keys]
        with torch.no_grad():
            for w in weights:
                w.copy_(torch.randn(w.shape, generator=g).cuda())
        A.bump_generation()
        A._pack(*keys[0])                       # re-packs (and splits) every stale operand
        lib = _lib.load()
        inline = 0
        for (w, c0, c, d), old in zip(keys, first):
            packed = A._pack(w, c0, c, d)
            rec = A._rec_of(packed)
            kh, kw = w.shape[2], w.shape[3]
            want, want_split = torch.empty_like(packed), torch.empty_like(packed)
            cout, cin = (c, w.shape[0]) if d else (w.shape[0], c)      # the operand's N and K-side channel counts
            with torch.cuda.device(w.device):
                st = _lib.current_stream_ptr(w.device)
                _lib.check(lib.zs_pack_conv_weight(_lib.ptr(w.detach()), _lib.ptr(want), w.shape[0], c, c0, w.shape[1], kh, kw,
                                                   1 if d else 0, st), "zs_pack_conv_weight")
                _lib.check(lib.zs_conv2d_presplit_weight(_lib.ptr(want), _lib.ptr(want_split), (cin + 3) // 4 * 4, cout, kh, kw,
                                                         st), "zs_conv2d_presplit_weight")
            assert A._split_of(packed) is not None, (tuple(w.shape), c0, c, d)
            assert torch.equal(A._split_of(packed).view(torch.int32), want_split.view(torch.int32)), (tuple(w.shape), c0, c, d)
            eligible = bool(lib.zs_pack_entry_inline_split(w.shape[0], c, kh * kw, 1 if d else 0))
            assert rec.inline_split == eligible
            inline += eligible
            if eligible and bwd == "f16x3":
                assert torch.equal(packed, old), "split_only rewrote the fp32 operand"
            else:
                assert torch.equal(packed, want), (tuple(w.shape), c0, c, d)
        assert inline >= 10
        # back to fp32: the generation moves, every fp32 operand is current again before anything reads it
        A.set_forward_precision("f32")
        A.set_backward_precision("f32")
        for (w, c0, c, d) in keys[:6]:
            packed = A._pack(w, c0, c, d)
            want = torch.empty_like(packed)
            with torch.cuda.device(w.device):
                _lib.check(lib.zs_pack_conv_weight(_lib.ptr(w.detach()), _lib.ptr(want), w.shape[0], c, c0, w.shape[1],
                                                   w.shape[2], w.shape[3], 1 if d else 0, _lib.current_stream_ptr(w.device)),
                           "zs_pack_conv_weight")
            assert torch.equal(packed, want)
    finally:
        A.set_forward_precision("f32")
        A.set_backward_precision("f32")
        A.clear_pack_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("presplit_all", [False, True])
def test_split_operand_follows_in_place_weight_updates(presplit_all, monkeypatch):
    """the fp16 halves of a packed operand (optim.amp forward) were cached by packed.data_ptr() + generation, so a
    weight changed by torch without bump_generation() - load_state_dict, w.mul_() under no_grad, a torch optimiser - kept computing
    with the old split.  The split now lives on the pack record and is stamped with the pack it was made from."""
    from zeroshape_amd.nn import autograd as A
    monkeypatch.setattr(A, "FWD_CONV_PRECISION", "f16x3")
    monkeypatch.setattr(A, "PRESPLIT_ALL", presplit_all)
    monkeypatch.setattr(A, "PRESPLIT_MIN_TILES", 1)            # every layer takes the pre-split path
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 16, 16, 32, generator=g).cuda()          # channels-last
    w = (torch.randn(64, 32, 3, 3, generator=g) / 17.0).cuda().requires_grad_(True)
    with torch.no_grad():
        y0 = A.conv2d(x, w, None, padding=1).clone()
        w.mul_(2.0)                                             # in place (bumps w._version), no bump_generation()
        y1 = A.conv2d(x, w, None, padding=1).clone()
        w.copy_(torch.zeros_like(w))                            # what load_state_dict does
        y2 = A.conv2d(x, w, None, padding=1).clone()
    scale = float(y0.abs().max())
    assert float((y1 - 2.0 * y0).abs().max()) <= 1e-5 * scale, "the split operand did not follow w.mul_()"
    assert float(y2.abs().max()) == 0.0, "the split operand did not follow w.copy_()"
