"""Launch wrappers for the encoder layers (include/zeroshape_hip.h, "Encoder layers").
Tensors are fp32 channels-last GPU tensors [B,H,W,C]; token matrices are [B,L,C]."""
import ctypes
import os

import torch

from .. import _lib

ACT_NONE, ACT_RELU, ACT_GELU, ACT_RELU_CLAMP1 = 0, 1, 2, 3


def _chk(t, what):
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError("%s: contiguous fp32 GPU tensor required" % what)
    return t


def _stream(t):
    return _lib.current_stream_ptr(t.device)


_TILING = {None: 0, "large": 2, "small": 4, "tile256": 1024}

# Arithmetic of the inference convolutions / linear layers (zs_conv2d_nhwc): "f32" = exact fp32 MFMA,
# "f16x3" = split-fp16 on the 16-bit matrix pipe (ZS_CONV_F16X3; operands as two fp16 halves, ~2^-21
# relative for 2e-4 <~ |x| <= 65504, saturating beyond 131008).  The training path (nn/autograd.py) is always fp32.
CONV_PRECISION = os.environ.get("ZS_ENCODER_PRECISION", "f16x3")
# K split across workgroups for launches that would leave most CUs idle (zs_conv2d_nhwc_ws).  OFF by default:
# measured at batch 1 (profiles/r02_encoder_b1_splitk*.txt) the 3x3 small-tile launches drop 16.7 -> 11.3 us but
# the pointwise ones stay at 11.5 us (bound by their dependent first loads, not by workgroup count) and the
# reduction kernel costs 4.5 us per layer: 4.65 vs 4.37 ms of kernel time per forward.  ZS_CONV_SPLIT_K=1 enables it.
SPLIT_K = os.environ.get("ZS_CONV_SPLIT_K", "0") != "0"
# (The workspace itself is always passed: the 3x3 input-patch kernels and the LDS-DMA kernel split the contraction of layers
# with few tiles through it by their own rules - csrc/nn_conv.hip, round 3.)
# Stream-K for the large-tile launches (ZS_CONV_STREAM_K): a fixed number of workgroups share the (tile, k-step)
# space evenly instead of one tile each.  Built, tested (test_conv2d_stream_k) and OFF by default: measured inside the
# batch-28 encoder (tools/conv_shapes.py) the large-tile kernels are bound by the bytes they pull into LDS, a short
# last round simply runs faster, and the exchange of tile shares costs more than the balance gains (22.9 vs 20.6 ms
# of convolution time with it everywhere).  ZS_CONV_STREAM_K=1 enables the kernel's own shape rule,
# "always" (tests) forces it wherever supported.
STREAM_K = {"0": False, "1": True, "always": "always"}[os.environ.get("ZS_CONV_STREAM_K", "0")]
_CONV_SPLIT_SMALL, _CONV_STREAM_K, _CONV_W_PRESPLIT, _CONV_STREAM_K_ALWAYS = 32, 64, 128, 256
# f16x3: the weights' fp16 halves are computed once per layer (zs_conv2d_presplit_weight) instead of in every
# workgroup of every launch.  ZS_CONV_PRESPLIT=0: split at run time (A/B measurements).
PRESPLIT = os.environ.get("ZS_CONV_PRESPLIT", "1") != "0"
# the ViT MLP's hidden tensor between fc1 and fc2 in K16-major layout where the streaming kernel serves both (batch 1);
# ZS_K16_HIDDEN=0: row-major everywhere (A/B measurements)
K16_HIDDEN = os.environ.get("ZS_K16_HIDDEN", "1") != "0"


def set_conv_precision(p):
    global CONV_PRECISION
    if p not in ("f32", "f16x3"):
        raise ValueError("conv precision must be 'f32' or 'f16x3', got %r" % (p,))
    CONV_PRECISION = p


_SPLITK_WS = {}


def _stream_key(device):
    """(device, stream the next launch goes to): the scratch buffers below are per stream (two streams
    running convolutions on one device would race on shared partial tiles; nn/branch.py does run two)."""
    return (str(device), torch._C._cuda_getCurrentRawStream(device.index if device.index is not None
                                                             else torch.cuda.current_device()))


def splitk_workspace(device):
    """Per-(device, stream) scratch of zs_conv2d_nhwc_ws: arrival counters (zeroed here once, left at zero by every
    launch) + partial tiles.  Fixed size and address: a captured hipGraph bakes the pointer in (the warm-up runs of
    nn.capture allocate it before the capture starts).  Launches on one stream serialise, so sharing it is safe."""
    key = _stream_key(device)
    if key not in _SPLITK_WS:
        _SPLITK_WS[key] = torch.zeros(_lib.load().zs_conv2d_splitk_workspace_bytes() // 4, dtype=torch.float32,
                                      device=device)
    return _SPLITK_WS[key]


def conv_flags():
    return (16 if CONV_PRECISION == "f16x3" else 0) | (_CONV_SPLIT_SMALL if SPLIT_K else 0) | \
        (_CONV_STREAM_K if STREAM_K else 0) | (_CONV_STREAM_K_ALWAYS if STREAM_K == "always" else 0)


class Stats:
    """Statistics a fused launch wrote beside its output (zs_conv_fuse.out_mode): `group` = [row tiles][groups][2] (sum, sum of
    squares) per 32-row tile, `row` = [M][column tiles][2] (sum, M2) per row; consumed by the next launch's in_mode."""

    def __init__(self, kind, data, tiles, groups=0):
        self.kind, self.data, self.tiles, self.groups = kind, data, tiles, groups


def fused_ok(x, pc=None):
    """The fused-normalisation launches (zs_conv2d_nhwc_fused) serve the split-fp16 inference engine when statistics tiles
    do not straddle samples: batch 1, or maps of a multiple of 32 pixels.  ZS_CONV_FUSE_NORM=0 disables them (A/B)."""
    B, H, W = x.shape[0], x.shape[1], x.shape[2]
    return FUSE_NORM and CONV_PRECISION == "f16x3" and PRESPLIT and (B == 1 or (H * W) % 32 == 0)


FUSE_NORM = os.environ.get("ZS_CONV_FUSE_NORM", "1") != "0"


_CONV_OUT_K16, _CONV_IN_K16 = 2048, 4096       # include/zeroshape_hip.h


def k16_ok(rows, cin, cout, in_k16=False, out_k16=False, ln_tiles=0, row_stats=False, has_res=False):
    """Would the library take this pointwise layer with K16-major operands (zs_conv2d_k16_ok)?  Asked before a tensor's layout is
    chosen: only the streaming GEMM kernel of the batch-1 engine reads / writes [C / 16][rows][16]."""
    if not (CONV_PRECISION == "f16x3" and PRESPLIT and K16_HIDDEN):
        return False
    flags = (_CONV_IN_K16 if in_k16 else 0) | (_CONV_OUT_K16 if out_k16 else 0)
    return bool(_lib.load().zs_conv2d_k16_ok(int(rows), int(cin), int(cout), flags, int(ln_tiles), 1 if row_stats else 0,
                                             1 if has_res else 0))


def conv2d(x, pc, res1=None, res2=None, act=ACT_NONE, in_relu=False, in_scale=1.0, in_shift=0.0, tiling=None,
           gn_in=None, ln_in=None, stats_out=None, out_groups=32, in_k16=False, out_k16=False):
    """x [B,H,W,Cin] -> [B,Ho,Wo,Cout] with the fused epilogue of zs_conv2d_nhwc.  `tiling`
    ("large" / "small") overrides the size-based choice of kernel variant (tests, tuning).

    Fused normalisations (zs_conv2d_nhwc_fused; the caller checks fused_ok):
      gn_in = (Stats 'group' of x, gamma, beta, eps): the layer runs on relu(GroupNorm(x));
      ln_in = (Stats 'row' of x, eps): the layer runs on (x - mean) * rstd per row (gamma / beta folded into pc);
      stats_out = 'group' | 'row': returns (out, Stats of out)."""
    lib = _lib.load()
    _chk(x, "conv2d input")
    B, H, W, C = x.shape
    assert C == pc.cin, "conv2d: input has %d channels, layer expects %d" % (C, pc.cin)
    Ho, pt = pc.out_size(H, pc.kh)
    Wo, pl = pc.out_size(W, pc.kw)
    out = torch.empty(B, Ho, Wo, pc.cout, dtype=torch.float32, device=x.device)
    for r in (res1, res2):
        if r is not None:
            _chk(r, "conv2d residual")
            assert r.shape == out.shape
    flags = (1 if in_relu else 0) | _TILING[tiling] | conv_flags()
    # K16-major operands (pointwise layers of few rows; the caller has asked k16_ok): x / the result keep their logical shape,
    # the bytes are [C / 16][rows][16]
    flags |= (_CONV_IN_K16 if in_k16 else 0) | (_CONV_OUT_K16 if out_k16 else 0)
    w = pc.w
    if PRESPLIT and CONV_PRECISION == "f16x3":
        if pc.w16 is None:
            pc.w16 = torch.empty_like(pc.w)
            with _lib.on(x.device):
                _lib.check(lib.zs_conv2d_presplit_weight(_lib.ptr(pc.w), _lib.ptr(pc.w16), C, pc.cout, pc.kh, pc.kw,
                                                         _stream(x)), "zs_conv2d_presplit_weight")
        w, flags = pc.w16, flags | _CONV_W_PRESPLIT
    fused = gn_in is not None or ln_in is not None or stats_out is not None
    if not fused:
        with _lib.on(x.device):
            _lib.check(lib.zs_conv2d_nhwc_ws(_lib.ptr(x), _lib.ptr(w), _lib.ptr(pc.scale), _lib.ptr(pc.shift),
                                             _lib.ptr(res1), _lib.ptr(res2), _lib.ptr(out), B, H, W, C, Ho, Wo, pc.cout,
                                             pc.kh, pc.kw, pc.stride, pt, pl,
                                             flags, float(in_scale), float(in_shift), act,
                                             _lib.ptr(splitk_workspace(x.device)), _stream(x)),
                       "zs_conv2d_nhwc_ws")
        return out
    assert CONV_PRECISION == "f16x3" and PRESPLIT and not in_relu, "fused normalisations: split-fp16 inference engine only"
    fz = _lib.ConvFuse()
    st_out = None
    if gn_in is not None:
        st, gamma, beta, eps = gn_in
        assert st.kind == "group" and st.groups == 32
        fz.in_mode, fz.in_tiles, fz.in_groups, fz.in_eps = 1, st.tiles, 32, float(eps)
        fz.in_stats, fz.in_gamma, fz.in_beta = st.data.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    elif ln_in is not None:
        st, eps = ln_in
        assert st.kind == "row"
        fz.in_mode, fz.in_tiles, fz.in_eps, fz.in_stats = 2, st.tiles, float(eps), st.data.data_ptr()
    M = B * Ho * Wo
    if stats_out == "group":
        tiles = (M + 31) // 32
        data = torch.empty(tiles, out_groups, 2, dtype=torch.float32, device=x.device)
        fz.out_mode, fz.out_groups, fz.out_stats = 1, out_groups, data.data_ptr()
        st_out = Stats("group", data, tiles // B, out_groups)
    elif stats_out == "row":
        cols = lib.zs_conv2d_fused_cols(M, pc.cout)
        tiles = (pc.cout + cols - 1) // cols
        data = torch.empty(M, tiles, 2, dtype=torch.float32, device=x.device)
        fz.out_mode, fz.out_stats = 2, data.data_ptr()
        st_out = Stats("row", data, tiles)
    with _lib.on(x.device):
        _lib.check(lib.zs_conv2d_nhwc_fused(_lib.ptr(x), _lib.ptr(w), _lib.ptr(pc.scale), _lib.ptr(pc.shift),
                                            _lib.ptr(res1), _lib.ptr(res2), _lib.ptr(out), B, H, W, C, Ho, Wo, pc.cout,
                                            pc.kh, pc.kw, pc.stride, pt, pl, flags & ~_CONV_SPLIT_SMALL, float(in_scale),
                                            float(in_shift), act, ctypes.addressof(fz),
                                            _lib.ptr(splitk_workspace(x.device)), _stream(x)),
                   "zs_conv2d_nhwc_fused")
    return out if st_out is None else (out, st_out)


# DPT's depth head ends in Conv 3x3 (128 -> 32) + ReLU + Conv 1x1 (32 -> 1) + ReLU: zs_conv3x3_tail_nhwc runs both as one
# launch of the input-patch kernel (the 32-channel map - 180 MB at 224 x 224 x 28 - is never written).  ZS_CONV_FUSE_TAIL=0: two
# launches (A/B measurements).
FUSE_TAIL = os.environ.get("ZS_CONV_FUSE_TAIL", "1") != "0"


FUSE_UPSAMPLE = os.environ.get("ZS_CONV_FUSE_UPSAMPLE", "1") != "0"


def conv2d_tail(x, pc, pc_tail, act=ACT_NONE, tail_act=ACT_NONE, in_relu=False, upsample=False):
    """tail_act(conv1x1_to_one_channel(act(conv3x3(x)))) -> [B,H,W,1]; fused when the layer pair fits zs_conv3x3_tail_nhwc
    (split-fp16 arithmetic, 3x3 stride 1 pad 1, <= 32 channels in between), two conv2d calls otherwise.  upsample=True: the
    3x3 layer runs on upsample2x(x) (DPT's head) - inside the same launch when fused (ZS_CONV_IN_UPSAMPLE2)."""
    up_fused = upsample and FUSE_UPSAMPLE and FUSE_TAIL and CONV_PRECISION == "f16x3" and PRESPLIT
    if upsample and not up_fused:
        return conv2d_tail(upsample2x(x), pc, pc_tail, act=act, tail_act=tail_act, in_relu=in_relu)
    B, H, W, C = x.shape
    if up_fused:
        H, W = 2 * H, 2 * W
    fits = (FUSE_TAIL and CONV_PRECISION == "f16x3" and PRESPLIT and pc.kh == 3 and pc.kw == 3 and pc.stride == 1 and
            pc.padding == 1 and pc.cout <= 32 and C % 16 == 0 and H >= 8 and W >= 8 and pc_tail.kh == 1 and
            pc_tail.kw == 1 and pc_tail.stride == 1 and pc_tail.cout == 1 and pc_tail.cin == pc.cout and
            pc_tail.padding in (0, "same"))
    if not fits:
        if up_fused:
            x = upsample2x(x)
        return conv2d(conv2d(x, pc, act=act, in_relu=in_relu), pc_tail, act=tail_act)
    lib = _lib.load()
    _chk(x, "conv2d_tail input")
    assert C == pc.cin
    if pc.w16 is None:
        pc.w16 = torch.empty_like(pc.w)
        with _lib.on(x.device):
            _lib.check(lib.zs_conv2d_presplit_weight(_lib.ptr(pc.w), _lib.ptr(pc.w16), C, pc.cout, pc.kh, pc.kw, _stream(x)),
                       "zs_conv2d_presplit_weight")
    if getattr(pc_tail, "tail_vec", None) is None:          # column 0 of the packed [K16/4][CoutPad][4] operand, scale folded in
        wv = pc_tail.w.view(-1, pc_tail.w.numel() // (4 * ((pc_tail.cin + 15) // 16 * 4)), 4)[:, 0, :].reshape(-1)[:pc_tail.cin]
        if pc_tail.scale is not None:
            wv = wv * pc_tail.scale[0]
        pc_tail.tail_vec = wv.contiguous()
    out = torch.empty(B, H, W, 1, dtype=torch.float32, device=x.device)
    flags = (1 if in_relu else 0) | 16 | _CONV_W_PRESPLIT | (512 if up_fused else 0)
    with _lib.on(x.device):
        _lib.check(lib.zs_conv3x3_tail_nhwc(_lib.ptr(x), _lib.ptr(pc.w16), _lib.ptr(pc.scale), _lib.ptr(pc.shift), _lib.ptr(out),
                                            B, H, W, C, pc.cout, flags, act, _lib.ptr(pc_tail.tail_vec),
                                            _lib.ptr(pc_tail.shift), tail_act, _stream(x)), "zs_conv3x3_tail_nhwc")
    return out


def linear(x, pc, res1=None, act=ACT_NONE, ln_in=None, stats_out=None, in_k16=False, out_k16=False):
    """x [..., Cin] -> [..., Cout] through the same GEMM (1x1 geometry).  ln_in / stats_out: see conv2d (rows = tokens);
    with stats_out the result is (y, Stats).  in_k16 / out_k16: the operand's bytes are K16-major (conv2d)."""
    lead = x.shape[:-1]
    n = 1
    for d in lead:
        n *= d
    y = conv2d(x.reshape(1, 1, n, x.shape[-1]), pc,
               res1=None if res1 is None else res1.reshape(1, 1, n, pc.cout), act=act, ln_in=ln_in, stats_out=stats_out,
               in_k16=in_k16, out_k16=out_k16)
    if stats_out is not None:
        return y[0].view(*lead, pc.cout), y[1]
    return y.view(*lead, pc.cout)


_GN_WS = {}


def _group_norm_workspace(device, nbytes):
    """Per-device scratch of zs_group_norm_nhwc_ws (group sums per pixel chunk; grown on demand, address stable
    between growths: a captured hipGraph bakes the pointer in, and nn.capture's warm-up runs size it first)."""
    key = _stream_key(device)
    if key not in _GN_WS or _GN_WS[key].numel() * 8 < nbytes:
        _GN_WS[key] = torch.empty(max(nbytes // 8, 1 << 16), dtype=torch.float64, device=device)
    return _GN_WS[key]


def group_norm(x, gamma, beta, groups=32, eps=1e-5, relu=False, residual=None):
    lib = _lib.load()
    _chk(x, "group_norm input")
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    ws = None
    if x.numel() * 4 >= (8 << 20):      # large tensors: the coalesced two-launch form (csrc/nn_ops.hip: gn_partial / gn_apply)
        ws = _group_norm_workspace(x.device, lib.zs_group_norm_workspace_bytes(B, H * W, C, groups))
    with _lib.on(x.device):
        _lib.check(lib.zs_group_norm_nhwc_ws(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(residual),
                                             _lib.ptr(y), B, H * W, C, groups, float(eps), 1 if relu else 0,
                                             _lib.ptr(ws), _stream(x)), "zs_group_norm_nhwc")
    return y


def group_norm_apply(x, st, gamma, beta, eps=1e-5, relu=False, residual=None, res_gn=None):
    """GroupNorm(32) of x from the Stats 'group' a fused convolution wrote beside it, one pass: [relu](GN(x) + r) with
    r = residual, or GroupNorm(residual) when res_gn = (Stats, gamma, beta) (zs_group_norm_apply_stats)."""
    lib = _lib.load()
    _chk(x, "group_norm_apply input")
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    rst = rg = rb = None
    rt = 0
    if res_gn is not None:
        rst, rg, rb = res_gn[0].data, res_gn[1], res_gn[2]
        rt = res_gn[0].tiles
    with _lib.on(x.device):
        _lib.check(lib.zs_group_norm_apply_stats(_lib.ptr(x), _lib.ptr(st.data), st.tiles, _lib.ptr(gamma), _lib.ptr(beta),
                                                 _lib.ptr(residual), _lib.ptr(rst), rt, _lib.ptr(rg), _lib.ptr(rb), _lib.ptr(y),
                                                 B, H * W, C, float(eps), 1 if relu else 0, _stream(x)),
                   "zs_group_norm_apply_stats")
    return y


def layer_norm(x, gamma, beta, eps=1e-6):
    lib = _lib.load()
    _chk(x, "layer_norm input")
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty_like(x)
    with _lib.on(x.device):
        _lib.check(lib.zs_layer_norm(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(y), rows, C, float(eps),
                                     _stream(x)), "zs_layer_norm")
    return y


def attention(qkv, heads):
    """qkv [B,L,3*C] -> [B,L,C]."""
    lib = _lib.load()
    _chk(qkv, "attention input")
    B, L, C3 = qkv.shape
    C = C3 // 3
    out = torch.empty(B, L, C, dtype=torch.float32, device=qkv.device)
    with _lib.on(qkv.device):
        fn = lib.zs_attention_split if CONV_PRECISION == "f16x3" else lib.zs_attention
        _lib.check(fn(_lib.ptr(qkv), _lib.ptr(out), B, L, heads, C // heads, _stream(qkv)), "zs_attention")
    return out


def max_pool(x, k=3, stride=2, padding=1):
    """padding: int (torch, -inf) or "same" (timm MaxPool2dSame)."""
    lib = _lib.load()
    _chk(x, "max_pool input")
    B, H, W, C = x.shape

    def size(n):
        if padding == "same":
            out = -(-n // stride)
            return out, max((out - 1) * stride + k - n, 0) // 2
        return (n + 2 * padding - k) // stride + 1, padding
    Ho, pt = size(H)
    Wo, pl = size(W)
    y = torch.empty(B, Ho, Wo, C, dtype=torch.float32, device=x.device)
    with _lib.on(x.device):
        _lib.check(lib.zs_max_pool_nhwc(_lib.ptr(x), _lib.ptr(y), B, H, W, C, Ho, Wo, k, stride, pt, pl, _stream(x)),
                   "zs_max_pool_nhwc")
    return y


def gn_relu_max_pool(x, st, gamma, beta, k=3, stride=2, padding=1, eps=1e-5):
    """max_pool(relu(GroupNorm(32)(x))) from the Stats 'group' the producing convolution wrote (zs_gn_relu_max_pool_nhwc: a
    tiny table launch + the pooling launch; the normalised map is never written).  padding as max_pool."""
    lib = _lib.load()
    _chk(x, "gn_relu_max_pool input")
    B, H, W, C = x.shape

    def size(n):
        if padding == "same":
            out = -(-n // stride)
            return out, max((out - 1) * stride + k - n, 0) // 2
        return (n + 2 * padding - k) // stride + 1, padding
    Ho, pt = size(H)
    Wo, pl = size(W)
    y = torch.empty(B, Ho, Wo, C, dtype=torch.float32, device=x.device)
    table = torch.empty(B, C, 2, dtype=torch.float32, device=x.device)
    with _lib.on(x.device):
        _lib.check(lib.zs_gn_relu_max_pool_nhwc(_lib.ptr(x), _lib.ptr(st.data), st.tiles, _lib.ptr(gamma), _lib.ptr(beta),
                                                _lib.ptr(table), _lib.ptr(y), B, H, W, C, Ho, Wo, k, stride, pt, pl, float(eps),
                                                _stream(x)), "zs_gn_relu_max_pool_nhwc")
    return y


def global_mean(x):
    """[B,H,W,C] -> [B,C]."""
    lib = _lib.load()
    _chk(x, "global_mean input")
    B, H, W, C = x.shape
    y = torch.empty(B, C, dtype=torch.float32, device=x.device)
    with _lib.on(x.device):
        _lib.check(lib.zs_global_mean_nhwc(_lib.ptr(x), _lib.ptr(y), B, H * W, C, _stream(x)), "zs_global_mean_nhwc")
    return y


def upsample2x(x):
    lib = _lib.load()
    _chk(x, "upsample2x input")
    B, H, W, C = x.shape
    y = torch.empty(B, 2 * H, 2 * W, C, dtype=torch.float32, device=x.device)
    with _lib.on(x.device):
        _lib.check(lib.zs_upsample2x_nhwc(_lib.ptr(x), _lib.ptr(y), B, H, W, C, _stream(x)), "zs_upsample2x_nhwc")
    return y


def to_nhwc(x, cpad=None, mask=None):
    """NCHW [B,C,H,W] -> NHWC [B,H,W,cpad or C] (zero-filled extra channels), optionally times a
    per-pixel mask [B,1,H,W]."""
    lib = _lib.load()
    x = _chk(x.float().contiguous(), "to_nhwc input")
    B, C, H, W = x.shape
    cp = cpad or C
    if mask is not None:
        mask = _chk(mask.float().contiguous(), "to_nhwc mask")
        assert mask.numel() == B * H * W
    y = torch.empty(B, H, W, cp, dtype=torch.float32, device=x.device)
    with _lib.on(x.device):
        _lib.check(lib.zs_nchw_to_nhwc(_lib.ptr(x), _lib.ptr(mask), _lib.ptr(y), B, C, H * W, cp, _stream(x)),
                   "zs_nchw_to_nhwc")
    return y


def pad_channels(x, cpad):
    """[..., C] -> [..., cpad] zero-filled (each row is a one-pixel image for zs_nchw_to_nhwc)."""
    lib = _lib.load()
    _chk(x, "pad_channels input")
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty(*x.shape[:-1], cpad, dtype=torch.float32, device=x.device)
    with _lib.on(x.device):
        _lib.check(lib.zs_nchw_to_nhwc(_lib.ptr(x), None, _lib.ptr(y), rows, C, 1, cpad, _stream(x)),
                   "zs_nchw_to_nhwc")
    return y


def window_tokens(emb, mask, invalid_token, cls, pos, win):
    """emb [B,H,W,C], mask [B,H,W] bool -> [B*(H/win)*(W/win), win*win+1, C] (zs_window_tokens)."""
    lib = _lib.load()
    _chk(emb, "window_tokens input")
    B, H, W, C = emb.shape
    m = mask.to(torch.uint8).contiguous()
    out = torch.empty(B * (H // win) * (W // win), win * win + 1, C, dtype=torch.float32, device=emb.device)
    with _lib.on(emb.device):
        _lib.check(lib.zs_window_tokens(_lib.ptr(emb), _lib.ptr(m), _lib.ptr(invalid_token), _lib.ptr(cls),
                                        _lib.ptr(pos), _lib.ptr(out), B, H, W, C, win, _stream(emb)),
                   "zs_window_tokens")
    return out


def to_nchw(x):
    lib = _lib.load()
    _chk(x, "to_nchw input")
    B, H, W, C = x.shape
    y = torch.empty(B, C, H, W, dtype=torch.float32, device=x.device)
    with _lib.on(x.device):
        _lib.check(lib.zs_nhwc_to_nchw(_lib.ptr(x), _lib.ptr(y), B, C, H * W, _stream(x)), "zs_nhwc_to_nchw")
    return y


def assemble_tokens(feat, cls, pos):
    """feat [B,n,C], cls [C], pos [n+1,C] -> [B,n+1,C]."""
    lib = _lib.load()
    _chk(feat, "assemble_tokens input")
    B, n, C = feat.shape
    y = torch.empty(B, n + 1, C, dtype=torch.float32, device=feat.device)
    with _lib.on(feat.device):
        _lib.check(lib.zs_assemble_tokens(_lib.ptr(feat), _lib.ptr(cls), _lib.ptr(pos), _lib.ptr(y), B, n, C,
                                          _stream(feat)), "zs_assemble_tokens")
    return y


def readout_concat(tokens):
    """tokens [B,n+1,C] -> [B,n,2C] = [patch token | cls token]."""
    lib = _lib.load()
    _chk(tokens, "readout_concat input")
    B, n1, C = tokens.shape
    y = torch.empty(B, n1 - 1, 2 * C, dtype=torch.float32, device=tokens.device)
    with _lib.on(tokens.device):
        _lib.check(lib.zs_readout_concat(_lib.ptr(tokens), _lib.ptr(y), B, n1 - 1, C, _stream(tokens)),
                   "zs_readout_concat")
    return y
