"""Mirror of the reference's model/shape/implicit.py::Implicit (:186-288) on the
hand-written HIP decoder.

Same constructor arguments, same ``state_dict`` keys and shapes (SURVEY.md section 8-b4:
``pos_embed``, ``point_proj.proj``, ``latent_proj``, ``blocks_attn.{i}.{norm1,attn.qkv,
attn.proj,norm2,mlp.fc1,mlp.fc2}``, ``norm``, ``impl_mlp.layers.{l}``), same call:

    logits[B, M], attn[B, M, 197] = impl_network(latent_depth[B,197,C], None, points[B,M,3])

The arithmetic runs in csrc/sdf_prologue.hip (per image, the point-independent latent
half) + csrc/sdf_decoder.hip (per point, fused) through the C ABI of
include/zeroshape_hip.h.  There is no PyTorch fallback: without the library, or for a
configuration the kernels are not specialised for, this module raises.

Training (``.train()`` mode or any call under autograd that needs gradients) runs the same
network layer by layer through zeroshape_amd/nn/autograd.py - HIP kernels for every forward and
backward op, torch.autograd only as the tape - including timm's per-sample DropPath
(implicit.py:83-109, drop_path=0.1).

Every OTHER constructor configuration of the reference class with a head dimension of 32 - its own defaults (512 channels,
16 heads, 6 MLP layers, latent_dim 768), a prediction head instead of the MLP (n_layers_mlp=0), other skips / ratios / block
counts, ``posenc_3D > 0`` (implicit.py:139-166), ``semantic=True`` - runs layer by layer on the training path's HIP kernels in
inference too (`fused` is False: forward(), with the attention map from zs_point_attention_probs, and compute_level_grid's
slice loop; prepare / query_* raise).  Head dimensions other than 32 and more than 256 latent rows raise.
"""
import contextlib
import os
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from ... import _lib
from ... import program as P
from ...nn import autograd as A
from ...utils.pos_embed import get_2d_sincos_pos_embed


class _Mlp(nn.Module):
    """Parameter container with timm 0.6.12 ``Mlp`` key names (fc1 / fc2)."""

    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)


class _Attention(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class _Block(nn.Module):
    def __init__(self, dim, mlp_ratio, norm_layer):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = _Attention(dim)
        self.norm2 = norm_layer(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))


class _Proj3D(nn.Module):
    def __init__(self, embed_dim):
        super().__init__()
        self.proj = nn.Linear(3, embed_dim)


class _MLPBlocks(nn.Module):
    def __init__(self, num_hidden_layers, n_channels, latent_dim, skip_in, posenc_res=0):
        super().__init__()
        # implicit.py:139-150: get_embedder(posenc_res, 3) widens the point part of `inputs` from 3 to 3 + 6 posenc_res
        dims = [3 + 6 * int(posenc_res) + latent_dim] + [n_channels] * num_hidden_layers + [1]
        self.layers = nn.ModuleList([
            nn.Linear(dims[l] + (dims[0] if l in skip_in else 0), dims[l + 1])
            for l in range(len(dims) - 1)])


class DecoderState(object):
    """Per-batch device state produced by Implicit.prepare(): one decoder program per
    image (weights + that image's K/V records).  A split-fp16 state also keeps the fp32
    programs it was derived from: the exact kernel re-evaluates the tiles the split kernel
    flags as outside its envelope (program.py: S_GUARD)."""

    def __init__(self, programs, batch, precision="f32", exact=None):
        self.programs, self.batch, self.precision, self.exact = programs, batch, precision, exact
        # split states from a calibrating prepare(): int32 [batch] on the device, 1 = this image's probe points differed by
        # more than CALIBRATION_TOL between the two arithmetics -> every tile of the image is re-evaluated in fp32
        self.image_flags = None       # int32 [B] from prepare()'s per-image check (1: evaluate this image in fp32), raw-logit rule
        self.image_flags_occ = None   # the same under the occupancy rule (calls that return sigmoid(logit))
        self.check_event = None       # recorded behind that check on its side stream; consumers wait for it on THEIR stream
        # per-weights verdicts of a calibrating prepare() on a split state (Implicit._calibrate): may calls that return raw
        # logits / calls that return occupancies use the split arithmetic?  (an uncalibrated split state: both True)
        self.logit_ok = True
        self.occ_ok = True

    @property
    def stride_bytes(self):
        return self.programs.stride(0) * 4


class Implicit(nn.Module):
    """Implicit function conditioned on depth encodings (implicit.py:186-288)."""

    def __init__(self, num_patches, latent_dim=768, semantic=False, n_channels=512,
                 n_blocks_attn=2, n_layers_mlp=6, num_heads=16, posenc_3D=0,
                 mlp_ratio=4., norm_layer=partial(nn.LayerNorm, eps=1e-6), drop_path=0.1,
                 skip_in=[], pos_perlayer=True):
        super().__init__()
        self.num_patches = num_patches
        self.pos_perlayer = pos_perlayer
        self.semantic = semantic
        self.num_heads = num_heads
        self.skip_in = tuple(skip_in)
        self.cfg = dict(num_patches=num_patches, latent_dim=latent_dim, n_channels=n_channels,
                        n_blocks_attn=n_blocks_attn, n_layers_mlp=n_layers_mlp, num_heads=num_heads,
                        posenc_3D=posenc_3D, mlp_ratio=float(mlp_ratio), skip_in=tuple(skip_in),
                        pos_perlayer=bool(pos_perlayer), semantic=bool(semantic))
        self.point_proj = _Proj3D(n_channels)
        self.latent_proj = nn.Linear(latent_dim, n_channels, bias=True)
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, n_channels), requires_grad=False)
        self.blocks_attn = nn.ModuleList([_Block(n_channels, mlp_ratio, norm_layer)
                                          for _ in range(n_blocks_attn)])
        self.norm = norm_layer(n_channels)
        self.posenc_3D = int(posenc_3D)
        self.impl_mlp = _MLPBlocks(n_layers_mlp, n_channels, n_channels, self.skip_in, self.posenc_3D) \
            if n_layers_mlp > 0 else None
        if self.impl_mlp is None:
            self.pred_head = nn.Linear(n_channels, 1, bias=True)
        self.drop_path = float(drop_path)
        self.drop_scales = None   # tests: explicit list of 2*n_blocks per-sample scale tensors [B]
        self.initialize_weights()
        self._packed = None       # (key, template programs tensor, lat_params tensor)
        # arithmetic of the fused inference kernels (default "f16x3"; ZS_DECODER_PRECISION=f32 or
        # .precision = "f32" selects the exact one): "f32" = exact fp32 MFMA (bitwise an fmaf
        # chain); "f16x3" = split-fp16 on the 16-bit matrix pipe (csrc/sdf_decoder_split.hip:
        # |logit difference| ~3e-6 to the fp32 kernel, contract 1e-4, 2.6x faster) INSIDE its
        # tested envelope (program.py: W_MAX on the host, S_GUARD per 128-point tile on the
        # device); outside it the exact kernel runs - whole calls or single tiles.  Logits come
        # from the same arithmetic whether or not the attention map is requested; the map itself
        # and the training path always use fp32.
        self.precision = os.environ.get("ZS_DECODER_PRECISION", "f16x3")
        self._workspace = {}      # (device, stream) -> scratch tensor for the query kernels
        self._check_streams = {}  # device -> side stream of the per-image check
        self._probe_cache = {}    # device -> the probe points
        self._check_keepalive = []  # (event, split programs, exact programs) of checks that may still be running
        self.last_tile_flags = None   # int32 per 128-point tile of the last split-fp16 query (1 = re-evaluated in fp32)
        self.envelope_guard = True    # False: raw split-fp16 results everywhere (measurements of the arithmetic itself)
        # Output-error calibration of the default arithmetic (prepare()): once per weight version the raw split
        # kernel and the exact kernel evaluate the same CALIBRATION_POINTS probe points of the first image seen;
        # "f16x3" is kept per output space (CALIBRATION_TOL on raw logits, CALIBRATION_TOL_OCC + no index flip on
        # occupancies: see the constants); when both fail every later prepare() of these weights returns an fp32 state.  The envelope fences above look
        # at operands; this one looks at the result.  ZS_DECODER_CALIBRATE=0 / .calibrate = False turns it off.
        self.calibrate = os.environ.get("ZS_DECODER_CALIBRATE", "1") != "0"
        # per-image output check of prepare().  Its two probe launches cost the latency of one fp32 wave tile + one
        # split wave tile (1.2 + 0.45 ms at any batch up to 8 images, however few probe points); they run on a side stream beside
        # the first grid launch (_launch_image_check), which still delays that launch's last workgroups: ~0.8 ms per prepare()
        # instead of 2.6 ms in front of it.  ZS_DECODER_IMAGE_CHECK=0 (or impl.image_check = False) leaves the per-weights
        # verdict + the kernel's own envelope flags (S_GUARD) as the only guards, as in round 3.
        self.image_check = os.environ.get("ZS_DECODER_IMAGE_CHECK", "1") != "0"
        self.calibration_range = (-1.5, 1.5)      # probe cube (options/shape.yaml:52 eval.range)
        self._last_calibration = None  # dict(max_abs_diff, mean_abs_diff, max_abs_logit, points, tol, selected)
        self._last_check_event = None  # the newest per-image check (side stream): readers of last_calibration wait for it
        self._calibration = None      # (weights key, last_calibration)

    # ---- init (implicit.py:232-249) -------------------------------------------------
    def initialize_weights(self):
        pe = get_2d_sincos_pos_embed(self.pos_embed.shape[-1], int(self.num_patches ** .5), cls_token=True)
        self.pos_embed.data.copy_(torch.from_numpy(pe).float().unsqueeze(0))
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # ---- HIP path ---------------------------------------------------------------------
    @property
    def fused(self):
        """True when the fused inference kernels (prepare / query_*) serve this configuration: the geometry of
        options/shape.yaml:19-44 the kernels are specialised for (256 channels, 8 heads, 2 blocks, 8 MLP layers, skips 2/4/6, 196
        patches, posenc_3D 0, no semantic codes).  Every other constructor configuration of the reference class (its own
        defaults - 512 channels, 16 heads, 6 MLP layers, latent_dim 768 - a prediction head instead of the MLP, other skips /
        ratios / block counts, posenc_3D > 0, semantic codes) runs layer by layer on the same HIP kernels as the training
        path instead - forward(), and the slice loop of compute_level_grid."""
        c = self.cfg
        want = dict(num_patches=P.L - 1, n_channels=P.C, latent_dim=P.C, n_blocks_attn=P.BLOCKS,
                    n_layers_mlp=P.MLP_LAYERS - 1, num_heads=P.HEADS, posenc_3D=0, mlp_ratio=4.0,
                    skip_in=P.SKIP_IN, semantic=False)
        return all(c[k] == v for k, v in want.items())

    def _check_supported(self, fused=True):
        c = self.cfg
        if fused:
            if not self.fused:
                raise NotImplementedError(
                    "the fused HIP decoder kernels are specialised for options/shape.yaml:19-44; this configuration (%s) runs "
                    "layer by layer - call the module (forward) instead of prepare / query_*" % c)
            return
        # the layer-by-layer path: what its kernels need (zs_point_attention / zs_attention: head dimension 32, <= 256 latent rows)
        if c["n_channels"] % c["num_heads"] or c["n_channels"] // c["num_heads"] != 32:
            raise NotImplementedError("the HIP attention kernels need a head dimension of 32 (n_channels %d / num_heads %d)"
                                      % (c["n_channels"], c["num_heads"]))
        if c["num_patches"] + 1 > 256 or c["n_blocks_attn"] < 1:
            raise NotImplementedError("the HIP point-attention kernels take at most 256 latent rows (num_patches + 1 = %d) and "
                                      "at least one attention block" % (c["num_patches"] + 1))

    def _weights_key(self):
        return (A.GENERATION[0], bool(self.pos_perlayer)) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def packed(self, device):
        """(template program [PROGRAM_FLOATS], lat_params) on ``device``; repacked when any
        parameter changed (in-place update or re-assignment)."""
        key = (str(device),) + self._weights_key()
        if self._packed is None or self._packed[0] != key:
            self._check_supported()
            sd = {k: v.detach().float().cpu().numpy() for k, v in self.state_dict().items()}
            prog = torch.from_numpy(P.pack_program(sd)).to(device)
            lat = torch.from_numpy(P.pack_latent_params(sd)).to(device)
            self._packed = (key, prog, lat, P.split_envelope(sd)[0])
        return self._packed[1], self._packed[2]

    @property
    def last_calibration(self):
        """The newest calibration record.  Its per-image maxima are written by the side-stream check of prepare(): the
        CURRENT stream is made to wait for that check first (no host wait), so a later .cpu() / .tolist() reads finished data."""
        ev = self._last_check_event
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        return self._last_calibration

    @last_calibration.setter
    def last_calibration(self, value):
        self._last_calibration = value

    def split_allowed(self, device):
        """Host half of the envelope guard: weights finite and within program.W_MAX."""
        self.packed(device)
        return self._packed[3]

    def workspace(self, device, extra_bytes=0):
        """Scratch for the query kernels (zs_sdf_workspace_bytes() [+ the attention dump], one per (device, stream):
        launches on one stream serialise, so sharing it there is safe)."""
        key = (str(device), torch.cuda.current_stream(device).cuda_stream)      # (the per-image check runs on a side stream)
        need = (_lib.load().zs_sdf_workspace_bytes() + extra_bytes + 3) // 4
        if key not in self._workspace or self._workspace[key].numel() < need:
            # zeroed ONCE: the tail holds the split kernel's tile counter, which every launch leaves at zero again
            self._workspace[key] = torch.zeros(need, dtype=torch.float32, device=device)
        return self._workspace[key]

    CALIBRATION_POINTS = 4096
    # The verdict is taken in the space the call RETURNS.  Raw logits (query_points, apply_sigmoid=False):
    # max |logit difference| <= CALIBRATION_TOL, a quarter of the 1e-4 contract.  Occupancies (compute_level_grid's sigmoid,
    # utils/eval_3D.py:44-45 - what the 1e-4 contract and the occ > 0.5 index set are defined on): max |occupancy difference|
    # <= CALIBRATION_TOL_OCC and no occ > 0.5 flip outside |logit| < FLIP_BAND.  A confident checkpoint (|logit| 30-100) has raw
    # differences that grow with the logit scale while its occupancies agree to 1e-7: under the raw rule alone its grids would
    # silently run the 2.7x slower fp32 kernel.
    # Why a QUARTER of the contract: 4,096 probes sample the grid - the largest difference on a full 129^3 grid has measured up
    # to 2.1x the probes' maximum (trained weights: 1.14e-5 on the probes, 2.38e-5 on the grid; bench.py: trained_weights) -
    # so a state that passes here stays a factor 2 inside 1e-4 on the grid it serves.
    CALIBRATION_TOL = 2.5e-5
    CALIBRATION_TOL_OCC = 2.5e-5
    FLIP_BAND = 1e-5

    def _probe_points(self, device):
        """Deterministic probe cloud in the evaluation cube: a scrambled lattice (golden-ratio steps per axis), so
        every call and every box uses the same 4096 points."""
        cached = self._probe_cache.get(str(device))
        if cached is not None:
            return cached
        i = torch.arange(self.CALIBRATION_POINTS, dtype=torch.float64)
        lo, hi = self.calibration_range
        frac = torch.stack([(i * a + b) % 1.0 for a, b in ((0.7548776662466927, 0.5), (0.5698402909980532, 0.25),
                                                            (0.4301597090019468, 0.75))], -1)
        pts = self._probe_cache[str(device)] = (lo + (hi - lo) * frac).to(torch.float32)[None].to(device)
        return pts

    def _verdict_stats(self, got, want):
        """([max |dlogit|, mean |dlogit|, max |logit|, max |docc|, flips outside the band] per leading row of got / want [B, M],
        int32 [B, 2]: 1 where the row FAILS the raw-logit rule / the occupancy rule) - one launch of zs_sdf_verdict_stats on
        the current stream.  (Round 6: these were ~15 ATen launches; on the check's side stream, beside the grid launch's
        priority-raised waves, each of them took milliseconds.)"""
        lib = _lib.load()
        got, want = got.contiguous(), want.contiguous()
        B, M = got.shape
        stats = torch.empty(B, 5, dtype=torch.float32, device=got.device)
        flags = torch.empty(B, 2, dtype=torch.int32, device=got.device)
        with _lib.on(got.device):
            _lib.check(lib.zs_sdf_verdict_stats(_lib.ptr(got), _lib.ptr(want), B, M, float(self.FLIP_BAND), float(self.CALIBRATION_TOL),
                                                float(self.CALIBRATION_TOL_OCC), _lib.ptr(stats), _lib.ptr(flags),
                                                _lib.current_stream_ptr(got.device)), "zs_sdf_verdict_stats")
        return stats, flags

    @torch.no_grad()
    def _calibrate(self, split, exact):
        """Raw f16x3 vs fp32 logits over the probe points of image 0 -> last_calibration (one host read) -> (logit_ok, occ_ok).
        Cached per weight version: an optimizer step or load_state_dict() triggers the next measurement."""
        key = self._weights_key()
        if self._calibration is not None and self._calibration[0] == key:
            self.last_calibration = self._calibration[1]
            return self.last_calibration["selected"] == "f16x3", self.last_calibration["selected_occ"] == "f16x3"
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("Implicit.prepare: the f16x3 calibration of new weights needs one host read; run one "
                               "eager prepare() before capturing, or request precision='f32'")
        pts = self._probe_points(split.device)
        guard, flags = self.envelope_guard, self.last_tile_flags
        self.envelope_guard = False                 # the raw arithmetic is what is being measured
        try:
            raw = DecoderState(split[:1], 1, "f16x3", exact=exact[:1])
            got = self.query_points(raw, pts)
        finally:
            self.envelope_guard, self.last_tile_flags = guard, flags
        want = self.query_points(DecoderState(exact[:1], 1), pts)
        stats = self._verdict_stats(got, want)[0][0].cpu()
        if not bool(torch.isfinite(stats[2])):
            # the exact kernel itself is not finite on this image (NaN latent): no verdict on the weights - this
            # call gets the fp32 state (NaN like the reference), the next image calibrates
            self.last_calibration = None
            return False, False
        finite = bool(torch.isfinite(stats).all())
        ok = finite and float(stats[0]) <= self.CALIBRATION_TOL
        ok_occ = finite and float(stats[3]) <= self.CALIBRATION_TOL_OCC and float(stats[4]) == 0.0
        self.last_calibration = dict(max_abs_diff=float(stats[0]), mean_abs_diff=float(stats[1]),
                                     max_abs_logit=float(stats[2]), max_abs_occ_diff=float(stats[3]),
                                     flips_outside_band=int(stats[4]) if finite else -1,
                                     points=self.CALIBRATION_POINTS, tol=self.CALIBRATION_TOL, tol_occ=self.CALIBRATION_TOL_OCC,
                                     flip_band=self.FLIP_BAND,
                                     selected="f16x3" if ok else "f32", selected_occ="f16x3" if ok_occ else "f32")
        self._calibration = (key, self.last_calibration)
        return ok, ok_occ

    @torch.no_grad()
    def prepare(self, latent_depth, precision=None, calibrate=None, shard=None):
        """Per-image prologue: latent_depth [B,197,C] (any float dtype, GPU) -> DecoderState.
        ``shard``: None, or (rank, world, exchange) when the grid of every image is sharded over `world` ranks that all hold the
        whole batch (parallel.prepare_sharded): every rank runs every prologue (0.5 ms at any batch up to 8, bit-identical
        programs everywhere, no collective), but the per-image f16x3-vs-fp32 CHECK of image i - 4,096 probe points through both
        kernels, 6 % of a rank's step when all 8 images are checked by all 8 ranks - runs on rank i % world only; ``exchange``
        (own int32 [k, 2] -> [B, 2]; parallel.exchange_image_flags: one 8k-byte all_gather) completes the flags on the check's
        side stream.
        ``precision``: None = self.precision.  "f16x3" is a request: outside the host envelope (W_MAX), or when
        the calibration of these weights fails BOTH rules (raw logits beyond CALIBRATION_TOL and occupancies beyond
        CALIBRATION_TOL_OCC / an index flip), the state returned is an fp32 one (``state.precision`` says which).  A split
        state that passes only one rule serves the calls of that kind and hands the others to the exact kernels
        (``state.logit_ok`` / ``state.occ_ok``).  ``calibrate``: None = self.calibrate;
        False returns the split state unchecked (measurements of the arithmetic itself)."""
        precision = self.precision if precision is None else precision
        calibrate = self.calibrate if calibrate is None else calibrate
        self._check_supported()                 # (raises for configurations that run layer by layer)
        if precision not in ("f32", "f16x3"):
            raise ValueError("decoder precision must be 'f32' or 'f16x3', got %r" % (precision,))
        if not latent_depth.is_cuda:
            raise ValueError("latent_depth must be a GPU tensor; zeroshape_amd has no CPU path")
        lib = _lib.load()
        lat = latent_depth.detach().to(torch.float32).contiguous()
        B = lat.shape[0]
        if tuple(lat.shape[1:]) != (P.L, P.C):
            raise ValueError("latent_depth must be [B,%d,%d], got %s" % (P.L, P.C, tuple(lat.shape)))
        template, lat_params = self.packed(lat.device)
        assert lib.zs_sdf_program_bytes() == P.PROGRAM_BYTES, "layout mismatch between program.py and the library"
        programs = template.unsqueeze(0).repeat(B, 1)
        scratch = torch.empty(B * (lib.zs_sdf_prologue_scratch_bytes() // 4), dtype=torch.float32,
                              device=lat.device)
        with _lib.on(lat.device):
            rc = lib.zs_sdf_prologue_ex(_lib.ptr(programs), programs.stride(0) * 4, _lib.ptr(lat_params),
                                        _lib.ptr(lat), B, _lib.ptr(scratch), 1 if self.pos_perlayer else 0,     # ZS_SDF_POS_PERLAYER
                                        _lib.current_stream_ptr(lat.device))
        _lib.check(rc, "zs_sdf_prologue_ex")
        if precision == "f16x3" and self.split_allowed(lat.device):
            split = torch.empty_like(programs)
            with _lib.on(lat.device):
                rc = lib.zs_sdf_split_programs(_lib.ptr(programs), programs.stride(0) * 4, _lib.ptr(split),
                                               split.stride(0) * 4, B, _lib.current_stream_ptr(lat.device))
            _lib.check(rc, "zs_sdf_split_programs")
            if not calibrate:
                return DecoderState(split, B, "f16x3", exact=programs)
            ok, ok_occ = self._calibrate(split, programs)
            if ok or ok_occ:
                state = DecoderState(split, B, "f16x3", exact=programs)
                state.logit_ok, state.occ_ok = ok, ok_occ
                if self.image_check:
                    self._launch_image_check(state, split, programs, shard)
                return state
        return DecoderState(programs, B)

    def _sharded_image_check(self, split, exact, shard, streams=None):
        """_image_check() of this rank's images (rank, rank + world, ...) + the exchange that completes the flags: the same
        four tensors, over the whole batch (the maxima of the other ranks' images are not exchanged: -1)."""
        rank, world, exchange = shard
        B = split.shape[0]
        k = (B + world - 1) // world
        a = streams[0] if streams is not None else None
        ctx = (lambda st: torch.cuda.stream(st)) if streams is not None else (lambda st: contextlib.nullcontext())
        mine = len(range(rank, B, world))
        if mine:
            # (a strided view of the programs: the C ABI takes the program stride)
            f, f_occ, mx, mx_occ = self._image_check(split[rank::world], exact[rank::world], streams=streams)
        with ctx(a):
            own = torch.zeros(k, 2, dtype=torch.int32, device=split.device)
            maxima = torch.full((B,), -1.0, dtype=torch.float32, device=split.device)
            maxima_occ = maxima.clone()
            if mine:
                own[:mine, 0], own[:mine, 1] = f, f_occ
                maxima[rank::world], maxima_occ[rank::world] = mx, mx_occ
            flags = exchange(own, B)                    # [B, 2] on every rank
            return flags[:, 0].contiguous(), flags[:, 1].contiguous(), maxima, maxima_occ

    def _launch_image_check(self, state, split, exact, shard=None):
        """The per-image check on a SIDE stream, so that it runs beside the first grid launch instead of in front of it: the
        fp32 probe launch alone is the latency of one fp32 wave tile (1.2 ms however few points).  The state carries the
        flags and an event; the query paths wait for the event (on their stream, no host wait) between the split launch and
        the fp32 launch that re-evaluates flagged tiles, and OR the image flags into the tile flags there.  Inside a stream
        capture the check runs inline (a fork that the capture might not join is not worth the risk)."""
        dev = split.device
        main = torch.cuda.current_stream(dev)
        if torch.cuda.is_current_stream_capturing() or os.environ.get("ZS_DECODER_CHECK_INLINE", "0") != "0":
            state.image_flags, state.image_flags_occ, maxima, maxima_occ = \
                self._image_check(split, exact) if shard is None else self._sharded_image_check(split, exact, shard)
            self._last_check_event = None
        else:
            pair = self._check_streams.get(str(dev))
            if pair is None:
                # HIGH priority: when the prologue ends, the probe launches and the caller's grid launch become ready together;
                # a resident persistent grid launch would keep the probes waiting until its workgroups retire
                pair = self._check_streams[str(dev)] = (torch.cuda.Stream(device=dev, priority=-1),
                                                        torch.cuda.Stream(device=dev, priority=-1))
            side, side2 = pair
            side.wait_stream(main)                          # the programs are ready
            side2.wait_stream(main)
            # ... and the caller's stream waits for a marker at the HEAD of each side stream: when the prologue ends all three
            # queues become ready at once, and whichever launch is dispatched first takes the CUs.  Behind the markers the probe
            # launches are the next packets of their queues, while the caller's queue first has to resolve its barrier: the
            # probes start first (vox 128 without the markers: the grid launch won the race and the fp32 probe ran AFTER its 28 ms)
            for st_ in (side, side2):
                head = torch.cuda.Event()
                head.record(st_)
                main.wait_event(head)
            state.image_flags, state.image_flags_occ, maxima, maxima_occ = \
                self._image_check(split, exact, streams=(side, side2)) if shard is None else \
                self._sharded_image_check(split, exact, shard, streams=(side, side2))
            state.check_event = torch.cuda.Event()
            state.check_event.record(side)
            # the side streams read `split` / `exact` (10 MB per image each): they must outlive that work even if the caller drops
            # the state at once.  Not record_stream() - it parks the blocks behind events and makes the allocator grow its pool
            # with fresh 10 MB hipMallocs for several calls - but a reference held here until the check's event has completed
            if os.environ.get("ZS_DECODER_CHECK_RECORD_STREAM", "0") != "0":      # A/B: the allocator-side alternative
                for t in (split, exact):
                    t.record_stream(side)
                    t.record_stream(side2)
            else:
                while self._check_keepalive and self._check_keepalive[0][0].query():
                    self._check_keepalive.pop(0)
                self._check_keepalive.append((state.check_event, split, exact))
            for t in (state.image_flags, state.image_flags_occ, maxima, maxima_occ):
                t.record_stream(main)                       # written over there, read here
            self._last_check_event = state.check_event
        self._last_calibration = dict(self._last_calibration, per_image_max_abs_diff=maxima, per_image_max_abs_occ_diff=maxima_occ)

    @torch.no_grad()
    def _image_check(self, split, exact, streams=None):
        """The f16x3 error also depends on the image's K / V records, and the per-weights verdict above was measured on the
        first image seen.  So every prepare() runs the probe points of EVERY image through both kernels and flags - on the
        device, no host read - the images that fail the raw-logit rule / the occupancy rule (or are not finite): the fp32
        launch behind every split launch re-evaluates them entirely (_join_image_check).  `streams` = (a, b): the fp32 probes
        on a, the split probes on b (concurrently), the comparison on a; None: everything on the current stream.
        -> (int32 flags [B] under the raw-logit rule, int32 flags [B] under the occupancy rule, float32 max |dlogit| [B],
        float32 max |docc| [B]), all device tensors."""
        B = split.shape[0]
        a, b = streams if streams is not None else (None, None)
        ctx = (lambda st: torch.cuda.stream(st)) if streams is not None else (lambda st: contextlib.nullcontext())
        with ctx(a):
            pts = self._probe_points(split.device).expand(B, -1, -1).contiguous()
        if b is not None:
            b.wait_stream(a)                            # pts (before a's probe launch: b must not wait for THAT)
        with ctx(a):
            want = self.query_points(DecoderState(exact, B), pts)
        guard, flags = self.envelope_guard, self.last_tile_flags
        self.envelope_guard = False                 # the raw arithmetic is what is being measured
        try:
            with ctx(b):
                got = self.query_points(DecoderState(split, B, "f16x3", exact=exact), pts)
        finally:
            self.envelope_guard, self.last_tile_flags = guard, flags
        if b is not None:
            a.wait_stream(b)
            got.record_stream(a)
            pts.record_stream(b)
        with ctx(a):
            st, flags = self._verdict_stats(got, want)       # [B, 5], [B, 2] (NaN / inf anywhere: both flags set)
            return flags[:, 0], flags[:, 1], st[:, 0], st[:, 3]

    def _tile_flags(self, batch, m, device, state=None):
        """Tile flags of one split launch: zero - the kernel sets the tiles that leave its envelope, _join_image_check() adds
        the images the per-image check of prepare() flagged.  None: neither guard is on.  No host read anywhere."""
        if not self.envelope_guard and getattr(state, "image_flags", None) is None:
            return None
        return torch.zeros(batch * ((m + 127) // 128), dtype=torch.int32, device=device)

    @staticmethod
    def _for_space(state, occupancy):
        """The state a call returning raw logits (occupancy=False) / sigmoid occupancies (True) runs on: a split state whose
        per-weights verdict rejected that space hands the call to the exact kernels."""
        if state.precision == "f16x3" and not (state.occ_ok if occupancy else state.logit_ok):
            return DecoderState(state.exact, state.batch)
        return state

    @staticmethod
    def _join_image_check(state, flags, occupancy=False):
        """Between a split launch and the fp32 launch that re-evaluates flagged tiles: wait (this stream, not the host) for the
        state's per-image check and flag every tile of the images it flagged - under the rule of the space the call returns
        (occupancy: sigmoid outputs) - the fp32 launch then evaluates them whole."""
        image_flags = getattr(state, "image_flags_occ" if occupancy else "image_flags", None)
        if image_flags is None or flags is None:
            return
        if state.check_event is not None:
            torch.cuda.current_stream(flags.device).wait_event(state.check_event)
        flags.view(state.batch, -1).bitwise_or_(image_flags[:, None])

    @torch.no_grad()
    def query_points(self, state, points_3D, need_attn=False):
        """state from prepare(); points_3D [B,M,3] -> logits [B,M] fp32 (and, with need_attn,
        the attention map [B,M,197] of implicit.py:277)."""
        lib = _lib.load()
        state = self._for_space(state, False)
        pts = points_3D.detach().to(torch.float32).contiguous()
        if pts.dim() != 3 or pts.shape[2] != 3 or pts.shape[0] != state.batch:
            raise ValueError("points_3D must be [%d,M,3], got %s" % (state.batch, tuple(pts.shape)))
        if pts.device != state.programs.device:
            raise ValueError("points_3D and latent_depth live on different devices")
        M = pts.shape[1]
        out = torch.empty(state.batch, M, dtype=torch.float32, device=pts.device)
        attn, extra = None, 0
        if state.precision == "f16x3":
            if need_attn:
                raise ValueError("the attention map needs an fp32 DecoderState (prepare(..., precision='f32'))")
            flags = self._tile_flags(state.batch, M, pts.device, state)
            ws, st = _lib.ptr(self.workspace(pts.device)), _lib.current_stream_ptr(pts.device)
            with _lib.on(pts.device):
                rc = lib.zs_sdf_query_points_split(_lib.ptr(state.programs), state.stride_bytes, state.batch,
                                                   _lib.ptr(pts), M, _lib.ptr(out), _lib.ptr(flags), ws, st)
                _lib.check(rc, "zs_sdf_query_points_split")
                # flagged tiles (outside the split arithmetic's envelope) again, exactly
                self._join_image_check(state, flags)
                if flags is not None:
                    rc = lib.zs_sdf_query_points(_lib.ptr(state.exact), state.exact.stride(0) * 4, state.batch,
                                                 _lib.ptr(pts), M, _lib.ptr(out), None, _lib.ptr(flags), ws, st)
            _lib.check(rc, "zs_sdf_query_points")
            self.last_tile_flags = flags
            return out
        if need_attn:
            attn = torch.empty(state.batch, M, P.L, dtype=torch.float32, device=pts.device)
            extra = lib.zs_sdf_attn_scratch_bytes(state.batch, M)
        with _lib.on(pts.device):
            rc = lib.zs_sdf_query_points(_lib.ptr(state.programs), state.stride_bytes, state.batch,
                                         _lib.ptr(pts), M, _lib.ptr(out), _lib.ptr(attn), None,
                                         _lib.ptr(self.workspace(pts.device, extra)),
                                         _lib.current_stream_ptr(pts.device))
        _lib.check(rc, "zs_sdf_query_points")
        return (out, attn) if need_attn else out

    @torch.no_grad()
    def query_grid(self, latent_depth, axis, apply_sigmoid=True, slice_begin=0, slice_end=None,
                   state=None):
        """Dense-grid query (get_dense_3D_grid + compute_level_grid, utils/eval_3D.py:11-46)
        without a points tensor: ``axis`` = torch.linspace(range_min, range_max, G) on the GPU.
        Returns [B, slice_end - slice_begin, G, G] (x slowest, z fastest)."""
        lib = _lib.load()
        if state is None:
            state = self.prepare(latent_depth)
        state = self._for_space(state, bool(apply_sigmoid))
        axis = axis.detach().to(torch.float32).contiguous()
        if axis.device != state.programs.device:
            raise ValueError("axis and latent_depth live on different devices")
        G = axis.numel()
        slice_end = G if slice_end is None else slice_end
        out = torch.empty(state.batch, slice_end - slice_begin, G, G, dtype=torch.float32,
                          device=axis.device)
        if state.precision == "f16x3":
            flags = self._tile_flags(state.batch, (slice_end - slice_begin) * G * G, axis.device, state)
            ws, st = _lib.ptr(self.workspace(axis.device)), _lib.current_stream_ptr(axis.device)
            with _lib.on(axis.device):
                rc = lib.zs_sdf_query_grid_split(_lib.ptr(state.programs), state.stride_bytes, state.batch,
                                                 _lib.ptr(axis), G, slice_begin, slice_end,
                                                 1 if apply_sigmoid else 0, _lib.ptr(out), _lib.ptr(flags), ws, st)
                _lib.check(rc, "zs_sdf_query_grid_split")
                self._join_image_check(state, flags, occupancy=bool(apply_sigmoid))
                if flags is not None:
                    rc = lib.zs_sdf_query_grid(_lib.ptr(state.exact), state.exact.stride(0) * 4, state.batch,
                                               _lib.ptr(axis), G, slice_begin, slice_end,
                                               1 if apply_sigmoid else 0, _lib.ptr(out), _lib.ptr(flags), ws, st)
            _lib.check(rc, "zs_sdf_query_grid")
            self.last_tile_flags = flags
            return out
        with _lib.on(axis.device):
            rc = lib.zs_sdf_query_grid(_lib.ptr(state.programs), state.stride_bytes, state.batch,
                                       _lib.ptr(axis), G, slice_begin, slice_end,
                                       1 if apply_sigmoid else 0, _lib.ptr(out), None,
                                       _lib.ptr(self.workspace(axis.device)),
                                       _lib.current_stream_ptr(axis.device))
        _lib.check(rc, "zs_sdf_query_grid")
        return out

    @torch.no_grad()
    def query_grid_range(self, latent_depth, axis, point_begin, point_end, apply_sigmoid=True, state=None):
        """Points [point_begin, point_end) of the dense grid in memory order (x slowest, z fastest):
        [B, point_end - point_begin].  The unit of the multi-GPU sharding (parallel.py)."""
        lib = _lib.load()
        if state is None:
            state = self.prepare(latent_depth)
        state = self._for_space(state, bool(apply_sigmoid))
        axis = axis.detach().to(torch.float32).contiguous()
        if axis.device != state.programs.device:
            raise ValueError("axis and latent_depth live on different devices")
        G = axis.numel()
        out = torch.empty(state.batch, point_end - point_begin, dtype=torch.float32, device=axis.device)
        ws, st = _lib.ptr(self.workspace(axis.device)), _lib.current_stream_ptr(axis.device)
        sig = 1 if apply_sigmoid else 0
        with _lib.on(axis.device):
            if state.precision == "f16x3":
                flags = self._tile_flags(state.batch, point_end - point_begin, axis.device, state)
                rc = lib.zs_sdf_query_grid_range_split(_lib.ptr(state.programs), state.stride_bytes, state.batch,
                                                       _lib.ptr(axis), G, point_begin, point_end, sig, _lib.ptr(out),
                                                       _lib.ptr(flags), ws, st)
                _lib.check(rc, "zs_sdf_query_grid_range_split")
                self._join_image_check(state, flags, occupancy=bool(apply_sigmoid))
                if flags is not None:
                    rc = lib.zs_sdf_query_grid_range(_lib.ptr(state.exact), state.exact.stride(0) * 4, state.batch,
                                                     _lib.ptr(axis), G, point_begin, point_end, sig, _lib.ptr(out),
                                                     _lib.ptr(flags), ws, st)
                self.last_tile_flags = flags
            else:
                rc = lib.zs_sdf_query_grid_range(_lib.ptr(state.programs), state.stride_bytes, state.batch,
                                                 _lib.ptr(axis), G, point_begin, point_end, sig, _lib.ptr(out),
                                                 None, ws, st)
        _lib.check(rc, "zs_sdf_query_grid_range")
        return out

    def forward(self, latent_depth, latent_semantic, points_3D, need_attn=True):
        """implicit.py:251-288.  Returns (logits [B,M], attn [B,M,197]) like the reference;
        callers that drop the attention map (our compute_level_grid without vis, training-shape
        probes) pass need_attn=False and get (logits, None) from the faster kernel variant."""
        if self.semantic:                                                                            # :253
            if latent_semantic is None:
                raise ValueError("this Implicit was built with semantic=True: latent_semantic is required")
            latent_depth = torch.cat([latent_depth.to(torch.float32), latent_semantic.to(torch.float32)], dim=-1)
        elif latent_semantic is not None:
            raise ValueError("latent_semantic given to an Implicit built with semantic=False")
        if not self.fused:
            return self._forward_autograd(latent_depth, points_3D, want_attn=need_attn)
        if torch.is_grad_enabled() and (self.training or points_3D.requires_grad or latent_depth.requires_grad):
            logits = self._forward_autograd(latent_depth, points_3D)[0]
            if not need_attn:
                return logits, None
            with torch.no_grad():      # the attention map carries no gradient in the reference's losses
                return logits, self.query_points(self.prepare(latent_depth, "f32"), points_3D, need_attn=True)[1]
        state = self.prepare(latent_depth)
        if need_attn and state.precision == "f32":
            return self.query_points(state, points_3D, need_attn=True)
        logits = self.query_points(state, points_3D)
        if not need_attn:
            return logits, None
        # the map only exists in the exact kernel; the logits stay those of the configured arithmetic,
        # so a caller gets the same numbers with and without the map
        return logits, self.query_points(DecoderState(state.exact, state.batch), points_3D, need_attn=True)[1]

    # ---- training path (layer by layer, autograd over HIP kernels) -------------------------
    def _drop_scale(self, B, device):
        """timm drop_path (timm==0.6.12 layers/drop.py): bernoulli(keep) / keep per sample; None when
        the branch is kept as is (eval mode or drop_path == 0)."""
        if not self.training or self.drop_path <= 0.0:
            return None
        keep = 1.0 - self.drop_path
        return torch.empty(B, dtype=torch.float32, device=device).bernoulli_(keep).div_(keep)

    def _forward_autograd(self, latent_depth, points_3D, want_attn=False):
        """implicit.py:251-288 with the latent rows and the point rows kept as two row blocks
        (every op but the attention is row-wise, and the attention treats the blocks differently
        anyway, implicit.py:38-71), so no concatenated [B,197+M,C] tensor is ever built.  Any constructor configuration
        with a head dimension of 32 (_check_supported).  -> (logits [B,M], attention map [B,M,1+num_patches] | None)."""
        self._check_supported(fused=False)
        if not (latent_depth.is_cuda and points_3D.is_cuda):
            raise ValueError("latent_depth / points_3D must be GPU tensors; zeroshape_amd has no CPU path")
        lat = latent_depth.to(torch.float32)
        pts = points_3D.detach().to(torch.float32).contiguous()
        B, M = pts.shape[0], pts.shape[1]
        H = self.num_heads
        Ll = self.num_patches + 1
        if tuple(lat.shape[1:]) != (Ll, self.latent_proj.in_features):
            raise ValueError("latent codes must be [B,%d,%d], got %s" % (Ll, self.latent_proj.in_features, tuple(lat.shape)))
        attn = torch.empty(B, M, Ll, dtype=torch.float32, device=pts.device) if want_attn else None
        pts4 = A._pad_channels(pts, 4)
        xp = A.linear(pts4, self.point_proj.proj.weight, self.point_proj.proj.bias, cin=3)            # :253
        pos = self.pos_embed.detach().expand(B, -1, -1).contiguous()
        cl = lat.shape[-1]
        if cl % 4:
            lat = A.pad_channels(lat.contiguous(), (cl + 3) // 4 * 4)
        xl = A.linear(lat, self.latent_proj.weight, self.latent_proj.bias, res1=pos, cin=cl)         # :255,271-272
        nb = len(self.blocks_attn)
        scales = list(self.drop_scales) if self.drop_scales is not None else \
            [self._drop_scale(B, pts.device) for _ in range(2 * nb)]

        def residual(x, branch_in, w, b, scale):
            if scale is None:
                return A.linear(branch_in, w, b, res1=x)
            return A.add_scaled_rows(x, A.linear(branch_in, w, b), scale)

        ones = torch.ones(B, dtype=torch.float32, device=pts.device) if self.pos_perlayer and nb > 1 else None
        for i, blk in enumerate(self.blocks_attn):
            last = i == nb - 1
            s_attn, s_mlp = scales[2 * i], scales[2 * i + 1]
            if self.pos_perlayer and i > 0:
                xl = A.add_scaled_rows(xl, pos, ones)                                                # :269-272, every block
            # (fork=True: a normalisation hands its input through to the residual connection behind it and adds that
            # connection's gradient in its own backward kernel - nn/autograd.py "Forks")
            if last:
                hl, xl_res = A.layer_norm(xl, blk.norm1.weight, blk.norm1.bias), None
            else:
                hl, xl_res = A.layer_norm(xl, blk.norm1.weight, blk.norm1.bias, fork=True)
            qkv_l = A.linear(hl, blk.attn.qkv.weight, blk.attn.qkv.bias)
            hp, xp = A.layer_norm(xp, blk.norm1.weight, blk.norm1.bias, fork=True)
            qkv_p = A.linear(hp, blk.attn.qkv.weight, blk.attn.qkv.bias)
            if want_attn:                                                                            # :60-66,277
                A.point_attention_probs(qkv_p, qkv_l, H, attn, weight=1.0 / nb, accumulate=i > 0)
            op = A.point_attention(qkv_p, qkv_l, H)                                                  # :44-66
            xp = residual(xp, op, blk.attn.proj.weight, blk.attn.proj.bias, s_attn)
            hp, xp = A.layer_norm(xp, blk.norm2.weight, blk.norm2.bias, fork=True)
            hp = A.gelu(A.linear(hp, blk.mlp.fc1.weight, blk.mlp.fc1.bias))
            xp = residual(xp, hp, blk.mlp.fc2.weight, blk.mlp.fc2.bias, s_mlp)
            if not last:                                                                             # :67-76
                ol = A.attention(qkv_l, H)
                xl = residual(xl_res, ol, blk.attn.proj.weight, blk.attn.proj.bias, s_attn)
                hl, xl = A.layer_norm(xl, blk.norm2.weight, blk.norm2.bias, fork=True)
                hl = A.gelu(A.linear(hl, blk.mlp.fc1.weight, blk.mlp.fc1.bias))
                xl = residual(xl, hl, blk.mlp.fc2.weight, blk.mlp.fc2.bias, s_mlp)
        feat = A.layer_norm(xp, self.norm.weight, self.norm.bias)                                    # :279
        if self.impl_mlp is None:                                                                    # :284-286: prediction head
            return A.linear(feat, self.pred_head.weight, self.pred_head.bias).squeeze(-1), attn
        # MLPBlocks (:168-184): inputs = cat([xyz, feat]); skip layers see cat([x, inputs]) / sqrt(2).
        # The concatenations become sums of column-range products of the same weight matrix.
        C = feat.shape[-1]
        layers = self.impl_mlp.layers
        r2 = 0.7071067811865476
        x = None
        # the point part of `inputs`: xyz, or its NeRF encoding [xyz | sin, cos at 2^0 .. 2^(L-1)] (implicit.py:158-160)
        E = 3 + 6 * self.posenc_3D
        enc = pts4 if self.posenc_3D == 0 else A.posenc3d(pts, self.posenc_3D)
        for l, lin in enumerate(layers):
            if l == 0:
                y = A.linear(enc, lin.weight, None, cin0=0, cin=E)
                y = A.linear(feat, lin.weight, lin.bias, res1=y, cin0=E, cin=C)
            elif l in self.skip_in:
                y = A.linear(enc, lin.weight, None, in_scale=r2, cin0=C, cin=E)
                y = A.linear(feat, lin.weight, None, in_scale=r2, res1=y, cin0=C + E, cin=C)
                y = A.linear(x, lin.weight, lin.bias, in_scale=r2, res1=y, cin0=0, cin=C)
            else:
                y = A.linear(x, lin.weight, lin.bias)
            x = A.softplus(y, 100.0) if l < len(layers) - 1 else y
        return x.squeeze(-1), attn
