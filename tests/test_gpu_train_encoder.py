"""GPU parity of the encoder TRAINING kernels and modules (BatchNorm on batch statistics, GroupNorm,
pooling / resampling / layout adjoints, seen-surface geometry backward, DPT-hybrid and the ResNet-50
coordinate encoder under autograd, one whole Graph training step).

Op level and module level: against torch CPU autograd of the same op / of the oracle
(oracle/encoder_ref.py, oracle/frontend_ref.py under oracle/train_ref.differentiable()), on
well-conditioned inputs, at <= 2e-4 of each tensor's scale.  Whole graph: against the golden step of
the REAL reference (tests/golden/graph_train_golden.npz) with the conditioning-aware bands of
tests/test_oracle_graph_train.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import encoder_ref, frontend_ref, train_ref
from tests.test_gpu_train_ops import close, nhwc
from tests.test_oracle_graph_train import check_bn_stats, check_graph_grads, full_state_dict, graph_train_inputs

pytestmark = pytest.mark.gpu


def nchw(t):
    return t.permute(0, 3, 1, 2)


@pytest.mark.parametrize("B,H,C,relu,res", [(4, 14, 96, True, True), (3, 7, 64, False, False), (4, 1, 2048, True, True),
                                            (2, 56, 32, True, False), (5, 9, 40, False, True)])
def test_batch_norm_train(B, H, C, relu, res):
    from zeroshape_amd.nn import autograd as A
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, C, H, H, generator=g) * 1.5 + 0.3
    r = torch.randn(B, C, H, H, generator=g) if res else None
    bn_ref = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn_ref.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn_ref.bias.copy_(torch.randn(C, generator=g) * 0.2)
        bn_ref.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
        bn_ref.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    bn_gpu = torch.nn.BatchNorm2d(C)
    bn_gpu.load_state_dict(bn_ref.state_dict())
    bn_gpu = bn_gpu.cuda()
    xr = x.clone().requires_grad_(True)
    rr = None if r is None else r.clone().requires_grad_(True)
    y = bn_ref(xr)
    if rr is not None:
        y = y + rr
    y = F.relu(y) if relu else y
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xg = nhwc(x).cuda().requires_grad_(True)
    rg = None if r is None else nhwc(r).cuda().requires_grad_(True)
    yg = A.batch_norm_train(xg, bn_gpu, relu=relu, residual=rg)
    close(nchw(yg), y, what="forward")
    yg.backward(nhwc(gy).cuda())
    close(nchw(xg.grad), xr.grad, rtol=1e-4, what="dx")
    close(bn_gpu.weight.grad, bn_ref.weight.grad, rtol=1e-4, what="dgamma")
    close(bn_gpu.bias.grad, bn_ref.bias.grad, rtol=1e-4, what="dbeta")
    if rr is not None:
        close(nchw(rg.grad), rr.grad, what="dres")
    close(bn_gpu.running_mean, bn_ref.running_mean, what="running_mean")
    close(bn_gpu.running_var, bn_ref.running_var, what="running_var")
    assert int(bn_gpu.num_batches_tracked) == 1


@pytest.mark.parametrize("B,H,C,relu,res", [(2, 14, 256, True, True), (3, 28, 64, True, False), (1, 7, 1024, False, False)])
def test_group_norm_backward(B, H, C, relu, res):
    from zeroshape_amd.nn import autograd as A
    g = torch.Generator().manual_seed(C)
    x = torch.randn(B, C, H, H, generator=g) * 2 + 0.5
    ga, be = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    r = torch.randn(B, C, H, H, generator=g) if res else None
    xr, gr, br = [t.clone().requires_grad_(True) for t in (x, ga, be)]
    rr = None if r is None else r.clone().requires_grad_(True)
    y = F.group_norm(xr, 32, gr, br, 1e-5)
    if rr is not None:
        y = y + rr
    y = F.relu(y) if relu else y
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xg, gg, bg = nhwc(x).cuda().requires_grad_(True), ga.cuda().requires_grad_(True), be.cuda().requires_grad_(True)
    rg = None if r is None else nhwc(r).cuda().requires_grad_(True)
    yg = A.group_norm(xg, gg, bg, 32, 1e-5, relu=relu, residual=rg)
    close(nchw(yg), y, what="forward")
    yg.backward(nhwc(gy).cuda())
    close(nchw(xg.grad), xr.grad, rtol=1e-4, what="dx")
    close(gg.grad, gr.grad, rtol=1e-4, what="dgamma")
    close(bg.grad, br.grad, rtol=1e-4, what="dbeta")
    if rr is not None:
        close(nchw(rg.grad), rr.grad, what="dres")


@pytest.mark.parametrize("pad", [1, "same"])
def test_max_pool_backward_with_ties(pad):
    """Post-ReLU maps are full of exact zeros: ties must go to the first maximum like torch."""
    from zeroshape_amd.nn import autograd as A
    g = torch.Generator().manual_seed(4)
    x = F.relu(torch.randn(2, 16, 23, 22, generator=g))
    xr = x.clone().requires_grad_(True)
    if pad == "same":
        H, W = x.shape[2:]
        ph, pw = max((-(-H // 2) - 1) * 2 + 3 - H, 0), max((-(-W // 2) - 1) * 2 + 3 - W, 0)
        y = F.max_pool2d(F.pad(xr, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), value=float("-inf")), 3, 2)
    else:
        y = F.max_pool2d(xr, 3, 2, 1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xg = nhwc(x).cuda().requires_grad_(True)
    yg = A.max_pool(xg, 3, 2, pad)
    close(nchw(yg), y, what="forward")
    yg.backward(nhwc(gy).cuda())
    close(nchw(xg.grad), xr.grad, what="dx")


def test_upsample_global_mean_layout_and_token_adjoints():
    from zeroshape_amd.nn import autograd as A
    g = torch.Generator().manual_seed(6)
    # x2 bilinear, align_corners=True
    for shape in [(2, 8, 7, 9), (1, 4, 1, 5), (1, 3, 14, 14)]:
        x = torch.randn(shape, generator=g)
        xr = x.clone().requires_grad_(True)
        y = F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=True)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        xg = nhwc(x).cuda().requires_grad_(True)
        yg = A.upsample2x(xg)
        close(nchw(yg), y, what="upsample forward")
        yg.backward(nhwc(gy).cuda())
        close(nchw(xg.grad), xr.grad, what="upsample backward")
    # global mean
    x = torch.randn(3, 20, 7, 7, generator=g)
    xr = x.clone().requires_grad_(True)
    y = xr.mean((2, 3))
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xg = nhwc(x).cuda().requires_grad_(True)
    yg = A.global_mean(xg)
    close(yg, y, what="mean forward")
    yg.backward(gy.cuda())
    close(nchw(xg.grad), xr.grad, what="mean backward")
    # masked NCHW -> NHWC (+ channel padding)
    x, m = torch.randn(2, 3, 9, 8, generator=g), (torch.rand(2, 1, 9, 8, generator=g) > 0.4).float()
    xr = x.clone().requires_grad_(True)
    y = xr * m
    gy = torch.randn(2, 4, 9, 8, generator=g)
    y.backward(gy[:, :3])
    xg = x.cuda().requires_grad_(True)
    yg = A.to_nhwc(xg, cpad=4, mask=m.cuda())
    close(nchw(yg)[:, :3], y, what="to_nhwc forward")
    assert float(yg[..., 3].abs().max()) == 0
    yg.backward(nhwc(gy).cuda())
    close(xg.grad, xr.grad, what="to_nhwc backward")
    # position-grid resize (align_corners=False) 24x24 -> 14x14 and 10x10
    for go in (14, 10):
        p = torch.randn(24, 24, 32, generator=g)
        pr = p.clone().requires_grad_(True)
        y = F.interpolate(pr.permute(2, 0, 1)[None], size=(go, go), mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        pg = p.cuda().requires_grad_(True)
        yg = A.resize_grid(pg, go, go)
        close(yg, y, what="resize forward")
        yg.backward(gy.cuda())
        close(pg.grad, pr.grad, what="resize backward")
    # token assembly + readout concat
    feat, cls, pos = torch.randn(3, 12, 16, generator=g), torch.randn(16, generator=g), torch.randn(13, 16, generator=g)
    fr, cr, pr = [t.clone().requires_grad_(True) for t in (feat, cls, pos)]
    tok = torch.cat([cr.expand(3, 1, 16), fr], 1) + pr
    ro = torch.cat([tok[:, 1:], tok[:, :1].expand(-1, 12, -1)], -1)
    gy = torch.randn(ro.shape, generator=g)
    ro.backward(gy)
    fg, cg, pg = [t.cuda().requires_grad_(True) for t in (feat, cls, pos)]
    rog = A.readout_concat(A.assemble_tokens(fg, cg, pg))
    close(rog, ro, what="tokens forward")
    rog.backward(gy.cuda())
    close(fg.grad, fr.grad, what="dfeat")
    close(cg.grad, cr.grad, what="dcls")
    close(pg.grad, pr.grad, what="dpos")


def test_seen_surface_and_intrinsics_backward():
    """graph_shape.py:89-144 differentiated like torch.autograd does in the reference: through the
    matrix inverse, the masked mean and the max-radius normalisation."""
    from zeroshape_amd import synthetic as syn
    from zeroshape_amd.nn import autograd as A
    depth, mask, params = [torch.from_numpy(a) for a in syn.seeded_depth_scene(seed=2, batch=3)]
    H = W = depth.shape[-1]
    g = torch.Generator().manual_seed(8)
    pr, dr = params.clone().requires_grad_(True), depth.clone().requires_grad_(True)
    with train_ref.differentiable():
        K = frontend_ref.intr_param2mtx(H, W, pr)
        seen, coord, mask_dsp, _, _ = frontend_ref.seen_surface(dr, K, mask, 1)
        gs, gc = torch.randn(seen.shape, generator=g), torch.randn(coord.shape, generator=g)
        ((seen * gs).sum() + (coord * gc).sum()).backward()
    pg, dg = params.cuda().requires_grad_(True), depth.cuda().requires_grad_(True)
    Kg = A.intr_param2mtx(pg, H, W)
    seen_g, coord_g, mask_g = A.seen_surface(dg, Kg, mask.cuda())
    close(Kg, K, what="intr")
    close(seen_g, seen, rtol=5e-5, what="seen")
    close(coord_g, coord, rtol=5e-5, what="coord")
    assert torch.equal(mask_g.cpu(), mask_dsp)
    ((seen_g * gs.cuda()).sum() + (coord_g * gc.cuda()).sum()).backward()
    close(dg.grad, dr.grad, rtol=2e-4, what="d_depth")
    close(pg.grad, pr.grad, rtol=2e-4, what="d_params")


def _load(module, sd, prefix):
    module.load_state_dict({k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}, strict=True)
    return module.cuda()


def _opt():
    from zeroshape_amd.utils.options import EasyDict as edict
    return edict(H=224, W=224, image_size=[224, 224], device="cuda:0",
                 arch=edict(latent_dim=256, win_size=16, num_heads=8, depth=edict(encoder="resnet", dsp=1, n_blocks=12, pretrained=None),
                            rgb=edict(encoder=None, n_blocks=12),
                            impl=edict(n_channels=256, att_blocks=2, mlp_ratio=4.0, posenc_perlayer=False, mlp_layers=8,
                                       posenc_3D=0, skip_in=[2, 4, 6])),
                 pretrain=edict(depth=None),
                 training=edict(shape_loss=edict(impt_weight=1, impt_thres=0.01),
                                depth_loss=edict(grad_reg=0.1, depth_inv=True, mask_shrink=False)),
                 loss_weight=edict(shape=1, depth=None, intr=None),
                 optim=edict(lr=3e-5, lr_ft=1e-5, weight_decay=0.05, fix_dpt=False, fix_clip=True, clip_norm=None,
                             amp=False, accum=1, sched=False))


def _trainable(sd):
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")) and "head." not in k:
            v.requires_grad_(True)
    return sd


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


# ---- block level: every building block of the encoders, forward + all gradients, tight ----
def _block_case(name, x_nchw, ref_fn, gpu_fn, sd, params, tol=5e-6):
    g = torch.Generator().manual_seed(len(name))
    xr = x_nchw.clone().requires_grad_(True)
    with train_ref.differentiable():
        want = ref_fn(sd, xr)
    gy = torch.randn(want.shape, generator=g)
    want.backward(gy)
    tokens = x_nchw.dim() == 3
    xg = (x_nchw if tokens else nhwc(x_nchw)).cuda().requires_grad_(True)
    got = gpu_fn(xg)
    got.backward((gy if tokens else nhwc(gy)).cuda())
    errs = dict(forward=_rel(got if tokens else nchw(got), want), dx=_rel(xg.grad if tokens else nchw(xg.grad), xr.grad))
    for pn, p in params.items():
        errs[pn] = _rel(p.grad, sd[pn].grad)
    bad = {k: v for k, v in errs.items() if not v < tol}
    assert not bad, (name, bad)
    return max(errs.values())


def test_encoder_blocks_forward_and_gradients_tight(encoder_sd):
    """ResNet-50 bottlenecks (BatchNorm on batch statistics; plain, projection, stride 2), a
    ResNetV2 stage (StdConv 'SAME' + GroupNorm), a ViT block, a DPT fusion block and Bottleneck_Conv,
    each alone against torch autograd of the oracle: <= 5e-6 relative L2 on the output, the input
    gradient and every parameter gradient."""
    from zeroshape_amd.model.depth.dpt_depth import DPTDepthModel
    from zeroshape_amd.model.shape.seen_coord_enc import CoordEncRes
    from zeroshape_amd.nn import autograd as A, train_blocks as T
    g = torch.Generator().manual_seed(1)
    enc = _load(CoordEncRes(_opt()), encoder_sd, "coord_encoder.").train()
    dpt = _load(DPTDepthModel(), encoder_sd, "dpt_depth.").train()
    e = enc.encoder
    worst = 0.0

    def v1(blk):
        def f(x):
            stride = blk.conv2.stride[0]
            identity = x
            if hasattr(blk, "downsample"):
                identity = T._bn(A.conv2d(x, blk.downsample[0].weight, stride=stride), blk.downsample[1])
            y = T._bn(A.conv2d(x, blk.conv1.weight), blk.bn1, relu=True)
            y = T._bn(A.conv2d(y, blk.conv2.weight, stride=stride, padding=1), blk.bn2, relu=True)
            return T._bn(A.conv2d(y, blk.conv3.weight), blk.bn3, relu=True, residual=identity)
        return f
    for name, blk, cin, stride, down in (("layer1.0", e.layer1[0], 64, 1, True), ("layer1.1", e.layer1[1], 256, 1, False),
                                         ("layer2.0", e.layer2[0], 256, 2, True), ("layer4.0", e.layer4[0], 1024, 2, True)):
        sd = _trainable({k[len("coord_encoder.encoder."):]: v.clone() for k, v in encoder_sd.items()
                         if k.startswith("coord_encoder.encoder." + name + ".")})
        x = torch.randn(4, cin, 24 if cin < 1024 else 8, 24 if cin < 1024 else 8, generator=g)
        params = {name + "." + k: p for k, p in blk.named_parameters()}
        enc.zero_grad()
        worst = max(worst, _block_case(name, x, lambda sd_, x_, n=name, s_=stride, d=down:
                                       encoder_ref._bottleneck_v1(sd_, n, x_, s_, d), v1(blk), sd, params))
    # Bottleneck_Conv (utils/layers.py:76-100), k = 1 and 3
    for name, mod, k, c in (("depth_feat_proj.0", enc.depth_feat_proj[0], 1, 1024),):
        sd = _trainable({k_[len("coord_encoder."):]: v.clone() for k_, v in encoder_sd.items()
                         if k_.startswith("coord_encoder." + name + ".")})
        params = {name + "." + k_: p for k_, p in mod.named_parameters()}
        enc.zero_grad()
        worst = max(worst, _block_case(name, torch.randn(4, c, 14, 14, generator=g),
                                       lambda sd_, x_, n=name, kk=k: encoder_ref.bottleneck_conv(sd_, n, x_, kk),
                                       lambda x_, m=mod: T.bottleneck_conv(x_, m), sd, params))
    # one ViT block and one fusion block of DPT
    vit = dpt.pretrained.model
    pre = "dpt_depth.pretrained.model."
    sd = _trainable({k[len(pre):]: v.clone() for k, v in encoder_sd.items() if k.startswith(pre + "blocks.3.")})
    params = {"blocks.3." + k: p for k, p in vit.blocks[3].named_parameters()}
    dpt.zero_grad()
    worst = max(worst, _block_case("vit", torch.randn(2, 197, 768, generator=g),
                                   lambda sd_, x_: encoder_ref.vit_block(sd_, "blocks.3", x_, 12),
                                   lambda x_: T.vit_block(x_, vit.blocks[3], 12), sd, params))
    pre = "dpt_depth."
    sd = _trainable({k[len(pre):]: v.clone() for k, v in encoder_sd.items() if k.startswith(pre + "scratch.refinenet2.")})
    params = {"scratch.refinenet2." + k: p for k, p in dpt.scratch.refinenet2.named_parameters()}
    skip = torch.randn(2, 256, 28, 28, generator=g)
    dpt.zero_grad()
    worst = max(worst, _block_case("fusion", torch.randn(2, 256, 28, 28, generator=g),
                                   lambda sd_, x_: encoder_ref._fusion(sd_, "scratch.refinenet2", x_, skip),
                                   lambda x_: T.fusion(x_, dpt.scratch.refinenet2, nhwc(skip).cuda()), sd, params))
    # ResNetV2 stem + stages (StdConv 'SAME', GroupNorm, MaxPoolSame) on a small image: stage 0 output
    pre = "dpt_depth.pretrained.model.patch_embed.backbone."
    sd = _trainable({k[len(pre):]: v.clone() for k, v in encoder_sd.items() if k.startswith(pre)})
    bb = vit.patch_embed.backbone
    params = {k: p for k, p in bb.named_parameters() if k.startswith(("stem.", "stages.0."))}
    dpt.zero_grad()

    def gpu_v2(x4):
        return T.resnetv2(x4, bb)[0]
    xin = torch.randn(2, 3, 64, 64, generator=g)
    xr = xin.clone().requires_grad_(True)
    with train_ref.differentiable():
        want = encoder_ref.resnetv2_stages(sd, xr)[1][0]
    gy = torch.randn(want.shape, generator=g)
    want.backward(gy)
    xg = xin.cuda().requires_grad_(True)
    got = gpu_v2(A.to_nhwc(xg, cpad=4))
    got.backward(nhwc(gy).cuda())
    errs = dict(forward=_rel(nchw(got), want), dx=_rel(xg.grad, xr.grad))
    errs.update({k: _rel(p.grad, sd[k].grad) for k, p in params.items()})
    assert max(errs.values()) < 2e-5, {k: v for k, v in errs.items() if v > 2e-5}
    print("encoder blocks: worst relative L2 error %.2e (ResNetV2 stem+stage0: %.2e)" % (worst, max(errs.values())))


# ---- module level: tolerance calibrated by the network's own conditioning ----
def _module_grad_errors(module, sd_ref, sd_noise, prefix, skip=()):
    """Aggregate relative L2 error of all parameter gradients of `module` against the oracle's, and
    the same figure between two oracle runs whose inputs differ by 1e-6 (the noise floor: what
    fp32-level perturbations do to this network's gradients).  Per-parameter outliers bounded too."""
    num = den = nnum = 0.0
    worst = (0.0, None)
    for name, p in module.named_parameters():
        want = sd_ref[prefix + name].grad
        if name.startswith(skip) or want is None:
            continue
        w, n = want.double(), sd_noise[prefix + name].grad.double()
        e = float((p.grad.cpu().double() - w).norm())
        num, den, nnum = num + e * e, den + float(w.norm()) ** 2, nnum + float((n - w).norm()) ** 2
        worst = max(worst, (e / (float(w.norm()) + 1e-30), name))
    return (num / den) ** 0.5, (nnum / den) ** 0.5, worst


@pytest.mark.parametrize("B,S,global_grad", [(4, 224, False), (16, 96, True)])
def test_coord_encoder_train_step_vs_oracle_autograd(encoder_sd, B, S, global_grad):
    """CoordEncRes in .train() mode on DIVERSE samples, whole module.  The 16 blocks match one by
    one at 1e-6 (test above) but a 50-layer random-weight network on batch statistics of 4 samples
    grows every perturbation ~1.45x per block, forward and again backward: the tolerance is set by
    the oracle's own response to a 1e-6 input perturbation, measured here.  BatchNorm over the B
    values of the 1x1 global token is the worst amplifier (~100x per norm at B = 4), so the gradient
    THROUGH the global token is only compared at B = 16."""
    from zeroshape_amd.model.shape.seen_coord_enc import CoordEncRes
    g = torch.Generator().manual_seed(12)
    coord = torch.randn(B, 3, S, S, generator=g) * 0.3
    mask = (torch.rand(B, 1, S, S, generator=g) > 0.4).float()
    noise = torch.randn(coord.shape, generator=g) * 0.3e-6
    gy = None
    runs = []
    for c in (coord, coord + noise):
        sd = _trainable({k: v.clone() for k, v in encoder_sd.items() if k.startswith("coord_encoder.")})
        cr = c.clone().requires_grad_(True)
        with train_ref.differentiable():
            want = encoder_ref.coord_enc_res(encoder_ref._sub(sd, "coord_encoder."), cr, mask)
        if gy is None:
            gy = torch.randn(want.shape, generator=g)
            if not global_grad:
                gy[:, 0] = 0
        want.backward(gy)
        runs.append((sd, cr, want))
    (sd, cr, want), (sd_n, cr_n, want_n) = runs
    enc = _load(CoordEncRes(_opt()), encoder_sd, "coord_encoder.").train()
    cg = coord.cuda().requires_grad_(True)
    got = enc(cg, mask.cuda())
    fwd_noise = _rel(want_n[:, 1:], want[:, 1:])
    assert _rel(got[:, 1:], want[:, 1:]) < max(1e-4, 5 * fwd_noise)
    if global_grad:
        assert _rel(got[:, 0], want[:, 0]) < max(1e-4, 5 * _rel(want_n[:, 0], want[:, 0]))
    got.backward(gy.cuda())
    skip = () if global_grad else ("encoder.fc.", "encoder.layer4.")          # reached only through the global token
    err, floor, worst = _module_grad_errors(enc, sd, sd_n, "coord_encoder.", skip)
    d_err, d_floor = _rel(cg.grad, cr.grad), _rel(cr_n.grad, cr.grad)
    print("coord encoder B=%d: parameter gradients rel L2 %.2e (oracle noise floor %.2e, worst %.2e %s); d_coord %.2e "
          "(floor %.2e)" % (B, err, floor, worst[0], worst[1], d_err, d_floor))
    assert err < max(1e-3, 3 * floor) and d_err < max(1e-3, 3 * d_floor)
    assert worst[0] < max(5e-3, 10 * floor), worst
    for k in ("encoder.bn1.running_mean", "encoder.layer3.2.bn3.running_var", "depth_feat_proj.0.bn1.running_mean"):
        close(enc.state_dict()[k], sd["coord_encoder." + k].detach(), rtol=1e-3, what=k)


def test_dpt_train_step_vs_oracle_autograd(encoder_sd):
    """DPTDepthModel.forward_train + backward, whole module (tolerance: see the coordinate encoder)."""
    from zeroshape_amd import synthetic as syn
    from zeroshape_amd.model.depth.dpt_depth import DPTDepthModel
    g = torch.Generator().manual_seed(13)
    rgb = torch.from_numpy(syn.seeded_rgb_scene(seed=0, batch=2)[0])
    noise = torch.randn(rgb.shape, generator=g) * 1e-6
    gd = gf = None
    runs = []
    for im in (rgb, rgb + noise):
        sd = _trainable({k: v.clone() for k, v in encoder_sd.items() if k.startswith("dpt_depth.")})
        with train_ref.differentiable():
            depth, feat = encoder_ref.dpt_depth(encoder_ref._sub(sd, "dpt_depth."), im)
        if gd is None:
            gd, gf = torch.randn(depth.shape, generator=g), torch.randn(feat.shape, generator=g) * 0.05
        ((depth * gd).sum() + (feat * gf).sum()).backward()
        runs.append((sd, depth, feat))
    (sd, depth, feat), (sd_n, depth_n, feat_n) = runs
    dpt = _load(DPTDepthModel(), encoder_sd, "dpt_depth.").train()
    d_g, f_g = dpt(rgb.cuda(), get_feat=True)
    close(d_g, depth, rtol=1e-4, what="depth")
    close(f_g, feat, rtol=1e-4, what="layer_4")
    ((d_g * gd.cuda()).sum() + (f_g * gf.cuda()).sum()).backward()
    err, floor, worst = _module_grad_errors(dpt, sd, sd_n, "dpt_depth.")
    print("DPT: parameter gradients rel L2 %.2e (oracle noise floor %.2e, worst %.2e %s)" % (err, floor, worst[0], worst[1]))
    assert err < max(1e-3, 3 * floor)
    assert worst[0] < max(5e-3, 10 * floor), worst


def test_graph_train_step_matches_reference_golden(encoder_sd, seeded_sd, graph_train_golden):
    """Runner.train_iteration's forward/backward on the whole graph vs the real reference."""
    from zeroshape_amd.model.compute_graph.graph_shape import Graph
    from zeroshape_amd.utils.options import EasyDict as edict
    g = graph_train_golden
    opt = _opt()
    graph = Graph(opt)
    graph.load_state_dict(full_state_dict(encoder_sd, seeded_sd), strict=True)
    graph = graph.cuda().train()
    graph.impl_network.drop_scales = [torch.from_numpy(s).cuda() for s in g["drop_scales"]]
    var = edict({k: v.cuda() for k, v in graph_train_inputs(g).items()})
    var.idx = torch.arange(int(g["batch"]))
    var, loss = graph.forward(opt, var, training=True, get_loss=True)
    assert set(loss.keys()) == {"shape"}
    assert abs(float(loss.shape) - float(g["loss"])) < 3e-5, (float(loss.shape), float(g["loss"]))
    np.testing.assert_allclose(var.pred_sample_occ.detach().cpu().numpy(), g["pred_sample_occ"], atol=3e-3, rtol=0)
    np.testing.assert_allclose(var.gt_points_cam.cpu().numpy().reshape(-1)[::7], g["gt_points_cam_s7"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(var.depth_pred.detach().cpu().numpy().reshape(-1)[::211], g["depth_pred_s211"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(var.intr_pred.detach().cpu().numpy(), g["intr_pred"], rtol=1e-4, atol=1e-3)
    loss.shape.backward()
    grads = {k: p.grad.cpu() for k, p in graph.named_parameters() if p.grad is not None}
    assert check_graph_grads(grads, g) == 609
    check_bn_stats(graph.state_dict(), g)


def test_fix_dpt_freezes_depth_model_and_skips_its_backward(encoder_sd, seeded_sd, graph_train_golden):
    """optim.fix_dpt (graph_shape.py:33-36): no gradient reaches dpt_depth / intr_head / intr_proj."""
    from zeroshape_amd.model.compute_graph.graph_shape import Graph
    from zeroshape_amd.utils.options import EasyDict as edict
    g = graph_train_golden
    opt = _opt()
    opt.optim.fix_dpt = True
    graph = Graph(opt)
    graph.load_state_dict(full_state_dict(encoder_sd, seeded_sd), strict=True)
    graph = graph.cuda().train()
    var = edict({k: v.cuda()[:2] for k, v in graph_train_inputs(g).items()})
    var.idx = torch.arange(2)
    var, loss = graph.forward(opt, var, training=True, get_loss=True)
    loss.shape.backward()
    for k, p in graph.named_parameters():
        frozen = k.startswith(("dpt_depth.", "intr_head.", "intr_proj."))
        assert (p.grad is None) == (frozen or k == "impl_network.pos_embed"), k


# ---- the transformer coordinate encoder (arch.depth.encoder != 'resnet', dsp = 2) under autograd ----
def test_seen_surface_dsp2_backward():
    """graph_shape.py:131-144 with arch.depth.dsp = 2: the half-size masked resample
    (utils/util.py:336-345) and its adjoint, chained into the same-size geometry backward."""
    from zeroshape_amd import synthetic as syn
    from zeroshape_amd.nn import autograd as A
    depth, mask, params = [torch.from_numpy(a) for a in syn.seeded_depth_scene(seed=4, batch=3)]
    H = W = depth.shape[-1]
    g = torch.Generator().manual_seed(18)
    pr, dr = params.clone().requires_grad_(True), depth.clone().requires_grad_(True)
    with train_ref.differentiable():
        K = frontend_ref.intr_param2mtx(H, W, pr)
        seen, coord, mask_dsp, _, _ = frontend_ref.seen_surface(dr, K, mask, 2)
        gs, gc = torch.randn(seen.shape, generator=g), torch.randn(coord.shape, generator=g)
        ((seen * gs).sum() + (coord * gc).sum()).backward()
    pg, dg = params.cuda().requires_grad_(True), depth.cuda().requires_grad_(True)
    Kg = A.intr_param2mtx(pg, H, W)
    seen_g, coord_g, mask_g = A.seen_surface_dsp2(dg, Kg, mask.cuda())
    assert coord_g.shape == (3, 3, H // 2, W // 2) and mask_g.shape == (3, 1, H // 2, W // 2)
    close(seen_g, seen, rtol=5e-5, what="seen")
    close(coord_g, coord, rtol=5e-5, what="coord (dsp 2)")
    close(mask_g, mask_dsp, what="mask (dsp 2)")
    ((seen_g * gs.cuda()).sum() + (coord_g * gc.cuda()).sum()).backward()
    close(dg.grad, dr.grad, rtol=2e-4, what="d_depth")
    close(pg.grad, pr.grad, rtol=2e-4, what="d_params")


def test_window_tokens_backward():
    """CoordEmb's token preparation (seen_coord_enc.py:50-66): where(mask, emb, invalid) -> windows -> + pos,
    cls + pos[0] in front; gradients to the embedding, the invalid token and the class token."""
    from zeroshape_amd.nn import autograd as A
    g = torch.Generator().manual_seed(19)
    B, H, W, C, win = 2, 16, 24, 32, 8
    emb, inv, cls = torch.randn(B, H, W, C, generator=g), torch.randn(C, generator=g), torch.randn(C, generator=g)
    pos = torch.randn(win * win + 1, C, generator=g)
    mask = torch.rand(B, H, W, generator=g) > 0.4
    er, ir, cr = [t.clone().requires_grad_(True) for t in (emb, inv, cls)]
    x = torch.where(mask[..., None], er, ir.expand_as(er))
    x = x.view(B, H // win, win, W // win, win, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, win * win, C) + pos[1:]
    want = torch.cat(((cr + pos[0]).expand(x.shape[0], 1, C), x), 1)
    gy = torch.randn(want.shape, generator=g)
    want.backward(gy)
    eg, ig, cg = [t.cuda().requires_grad_(True) for t in (emb, inv, cls)]
    got = A.window_tokens(eg, mask.cuda(), ig, cg, pos.cuda(), win)
    close(got, want, what="window tokens")
    got.backward(gy.cuda())
    close(eg.grad, er.grad, what="d_emb")
    close(ig.grad, ir.grad, rtol=1e-5, what="d_invalid_coord_token")
    close(cg.grad, cr.grad, rtol=1e-5, what="d_cls_token")


def _att_reference(sd, coord, mask, scales, heads=8, win=8, n_blocks=12):
    """oracle/encoder_ref.coord_enc_att with timm's per-sample DropPath factors on the two residual branches of
    the global blocks (seen_coord_enc.py:93-97; timm layers/drop.py: x * bernoulli(keep) / keep), given explicitly."""
    E = encoder_ref
    emb = F.linear(coord, sd["coord_embed.pos_embed.weight"], sd["coord_embed.pos_embed.bias"])
    emb = torch.where(mask[..., None], emb, sd["coord_embed.invalid_coord_token"].expand_as(emb))
    B, H, W, C = emb.shape
    emb = emb.view(B, H // win, win, W // win, win, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, win * win, C)
    pe = sd["coord_embed.two_d_pos_embed"]
    cls = (sd["coord_embed.cls_token"] + pe[:, :1]).expand(emb.shape[0], -1, -1)
    emb = E.vit_block(sd, "coord_embed.blocks.0", torch.cat((cls, emb + pe[:, 1:]), 1), heads)
    x = torch.cat((sd["cls_token"].expand(B, -1, -1), emb[:, 0].view(B, -1, C)), 1)
    for i in range(n_blocks):
        p = "blocks.%d" % i
        h = F.layer_norm(x, (C,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-6)
        qkv = F.linear(h, sd[p + ".attn.qkv.weight"], sd[p + ".attn.qkv.bias"])
        q, k, v = qkv.reshape(B, -1, 3, heads, C // heads).permute(2, 0, 3, 1, 4).unbind(0)
        a = ((q @ k.transpose(-2, -1)) * (C // heads) ** -0.5).softmax(dim=-1) @ v
        a = F.linear(a.transpose(1, 2).reshape(B, -1, C), sd[p + ".attn.proj.weight"], sd[p + ".attn.proj.bias"])
        x = x + a * scales[2 * i].view(B, 1, 1)
        h = F.layer_norm(x, (C,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-6)
        h = F.linear(F.gelu(F.linear(h, sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"])),
                     sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])
        x = x + h * scales[2 * i + 1].view(B, 1, 1)
    return F.layer_norm(x, (C,), sd["norm.weight"], sd["norm.bias"], 1e-6)


@pytest.mark.parametrize("drop", [False, True])
def test_coord_enc_att_train_step(att_sd, drop):
    """CoordEncAtt in .train() mode: forward + every parameter gradient + d_coord against torch CPU autograd
    of the oracle (drop = False: oracle/encoder_ref.coord_enc_att itself; True: the same with DropPath factors,
    three of four samples dropped somewhere).  LayerNorm network: well conditioned, tight tolerance."""
    from zeroshape_amd.model.shape.seen_coord_enc import CoordEncAtt
    g = torch.Generator().manual_seed(21)
    B, S = 4, 112
    coord = torch.rand(B, S, S, 3, generator=g) * 2 - 1
    mask = torch.rand(B, S, S, generator=g) > 0.45
    mask[1, :16] = False                                  # whole windows of invalid pixels
    keep = 0.9
    scales = [(torch.rand(B, generator=g) < (0.7 if drop else 2.0)).float() / (keep if drop else 1.0) for _ in range(24)]
    sd = {k: v.clone() for k, v in att_sd.items()}
    for k, v in sd.items():
        if k != "coord_embed.two_d_pos_embed":
            v.requires_grad_(True)
    cr = coord.clone().requires_grad_(True)
    if drop:
        want = _att_reference(sd, cr, mask, scales)
    else:
        with train_ref.differentiable():
            want = encoder_ref.coord_enc_att(sd, cr, mask)
    gy = torch.randn(want.shape, generator=g)
    want.backward(gy)
    enc = CoordEncAtt(embed_dim=256, n_blocks=12, num_heads=8, win_size=8)
    enc.load_state_dict(att_sd, strict=True)
    enc = enc.cuda().train()
    enc.drop_scales = [s.cuda() for s in scales] if drop else [None] * 24
    cg = coord.cuda().requires_grad_(True)
    got = enc(cg, mask.cuda())
    assert got.shape == (B, 1 + (S // 8) ** 2, 256) and got.requires_grad
    assert _rel(got, want) < 2e-5, _rel(got, want)
    got.backward(gy.cuda())
    errs = {"d_coord": _rel(cg.grad, cr.grad)}
    for name, p in enc.named_parameters():
        if name == "coord_embed.two_d_pos_embed":
            assert p.grad is None
            continue
        assert p.grad is not None, name
        errs[name] = _rel(p.grad, sd[name].grad)
    worst = max(errs.items(), key=lambda kv: kv[1])
    print("CoordEncAtt train (drop_path %s): forward %.2e, worst gradient %.2e (%s)" % (drop, _rel(got, want), worst[1], worst[0]))
    assert worst[1] < 1e-4, {k: v for k, v in errs.items() if v > 1e-4}


def test_coord_enc_att_drop_path_sampling(att_sd):
    """Without explicit factors the module draws bernoulli(0.9)/0.9 per sample and branch (timm DropPath):
    seeded -> reproducible; eval mode takes the packed inference path (no DropPath)."""
    from zeroshape_amd.model.shape.seen_coord_enc import CoordEncAtt
    enc = CoordEncAtt(embed_dim=256, n_blocks=12, num_heads=8, win_size=8)
    enc.load_state_dict(att_sd, strict=True)
    enc = enc.cuda().train()
    g = torch.Generator().manual_seed(22)
    coord = (torch.rand(8, 32, 32, 3, generator=g) * 2 - 1).cuda()
    mask = (torch.rand(8, 32, 32, generator=g) > 0.3).cuda()
    s = enc._drop_scale(4096, coord.device)
    assert all(v == 0.0 or abs(v - 1.0 / 0.9) < 1e-6 for v in s.unique().tolist())
    assert 0.86 < float((s > 0).float().mean()) < 0.94
    torch.manual_seed(5)
    a = enc(coord, mask)
    torch.manual_seed(5)
    b = enc(coord, mask)
    torch.manual_seed(6)
    c = enc(coord, mask)
    assert torch.equal(a, b) and not torch.equal(a, c)
    with torch.no_grad(), pytest.raises(NotImplementedError):     # .train() without autograd: refused, like CoordEncRes
        enc(coord, mask)
    e = enc.eval()(coord, mask)
    enc.train().drop_scales = [None] * 24
    assert _rel(enc(coord, mask), e.cpu()) < 1e-5


def test_graph_trains_with_the_transformer_coordinate_encoder(encoder_sd, seeded_sd, graph_train_golden):
    """Graph.forward(training=True) with arch.depth.encoder = 'att', dsp = 2 (graph_shape.py:41-47, :131-150):
    the latent equals the oracle's encoder on the oracle's half-size seen-surface map of the predicted depth, and
    the shape loss reaches every trainable parameter of the coordinate encoder and the depth model."""
    from zeroshape_amd.model.compute_graph.graph_shape import Graph
    from zeroshape_amd.utils.options import EasyDict as edict
    g = graph_train_golden
    opt = _opt()
    opt.arch.depth.encoder, opt.arch.depth.dsp = "att", 2
    torch.manual_seed(3)
    graph = Graph(opt)
    sd = {k: v for k, v in full_state_dict(encoder_sd, seeded_sd).items() if not k.startswith("coord_encoder.")}
    missing = graph.load_state_dict(sd, strict=False)
    assert all(k.startswith("coord_encoder.") for k in missing.missing_keys) and not missing.unexpected_keys
    graph = graph.cuda().train()
    graph.coord_encoder.drop_scales = [None] * 24
    var = edict({k: v.cuda()[:2] for k, v in graph_train_inputs(g).items()})
    var.idx = torch.arange(2)
    var, loss = graph.forward(opt, var, training=True, get_loss=True)
    assert var.latent_depth.shape == (2, 197, 256) and bool(torch.isfinite(loss.shape))
    att = {k: v.detach().cpu() for k, v in graph.coord_encoder.state_dict().items()}
    _, coord, mask_dsp, _, _ = frontend_ref.seen_surface(var.depth_pred.detach().cpu(), var.intr_pred.detach().cpu(),
                                                         var.mask_input_map.cpu(), 2)
    want = encoder_ref.coord_enc_att(att, coord.permute(0, 2, 3, 1).contiguous(), mask_dsp.squeeze(1) > 0.5)
    assert _rel(var.latent_depth, want) < 1e-4, _rel(var.latent_depth, want)
    loss.shape.backward()
    unused_by_reference = {k for k in sd if k.startswith("dpt_depth.") and ("gnorm/" + k) not in g}
    for k, p in graph.named_parameters():
        if k in ("impl_network.pos_embed", "coord_encoder.coord_embed.two_d_pos_embed"):
            assert p.grad is None, k
        elif k.startswith("dpt_depth.") and p.grad is None:
            assert k in unused_by_reference, k
        else:
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k
    assert float(graph.coord_encoder.coord_embed.invalid_coord_token.grad.abs().max()) > 0
    assert float(graph.dpt_depth.scratch.output_conv[4].weight.grad.abs().max()) > 0
