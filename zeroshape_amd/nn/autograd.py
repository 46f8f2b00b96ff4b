"""torch.autograd bindings of the HIP training kernels (include/zeroshape_hip.h, "Training").

Every op here is a torch.autograd.Function whose forward AND backward are launches of
libzeroshape_hip.so; torch supplies the tape, the tensor memory and the stream, nothing else.
Activations are fp32 channels-last ([B,H,W,C]; token matrices [B,L,C]), parameters stay in the
reference's torch layout ([Cout,Cin,kh,kw] / [Cout,Cin]) and are re-packed on the GPU when they
change (zs_pack_conv_weight), so optimiser steps need no host work.

Reference behaviour reproduced: what torch.autograd gives train.py for Graph.forward(training=True)
(model/compute_graph/graph_shape.py:115-204) and Loss.shape_loss (utils/loss.py:18-28).
"""
import contextlib
import os

import torch

from .. import _lib

ACT_NONE, ACT_RELU, ACT_GELU, ACT_RELU_CLAMP1, ACT_SOFTPLUS = 0, 1, 2, 3, 4
_CONV_IN_RELU, _CONV_IN_DILATE2 = 1, 8
_CONV_F16X3 = 16
# Bumped whenever parameters are updated through raw pointers (the fused optimiser): tensor
# ._version does not see those writes, so every pack cache also keys on this counter.
GENERATION = [0]
# Arithmetic of the FORWARD convolutions / linear layers of the training path: "f32" (default) or
# "f16x3" = split-fp16 (opt-in: 15.4 -> 14.4 ms forward per step at batch 4, but the 1e-6 forward
# differences are amplified by the batch-statistics BatchNorms to ~2e-3 relative in some gradients,
# beyond this repository's gradient parity bar).
FWD_CONV_PRECISION = os.environ.get("ZS_TRAIN_FWD_PRECISION", "f32")


def set_forward_precision(p):
    """"f32" | "f16x3": arithmetic of the forward convolutions / linear layers under autograd (optim.amp selects
    "f16x3": the 16-bit matrix pipe with split operands, fp32 accumulation)."""
    global FWD_CONV_PRECISION
    if p not in ("f32", "f16x3"):
        raise ValueError("forward precision must be 'f32' or 'f16x3', got %r" % (p,))
    if p != FWD_CONV_PRECISION:
        GENERATION[0] += 1      # operands re-packed under the other setting may hold only the form that one reads
    FWD_CONV_PRECISION = p

# Data gradients (dx = dy * W^T through the same convolution engine).  "f16x3" needs the incoming gradients inside
# fp16's range: the Runner multiplies the loss by a dynamic power of two (optim.LossScaler, torch's GradScaler
# rules) and the optimiser divides it out, an overflow (inf / nan in any gradient) skips the step and halves the
# scale.  Weight gradients stay exact fp32 (they sum over every pixel of the batch).
BWD_DATA_PRECISION = os.environ.get("ZS_TRAIN_BWD_PRECISION", "f32")


# ... and of the weight-gradient GEMMs (zs_conv2d_wgrad with ZS_CONV_F16X3: wgrad_split_kernel, round 3).  Follows the
# data-gradient setting (optim.amp) unless ZS_TRAIN_WGRAD_PRECISION pins it (A/B measurements).
BWD_WGRAD_PRECISION = os.environ.get("ZS_TRAIN_WGRAD_PRECISION")


def wgrad_precision():
    return BWD_WGRAD_PRECISION or BWD_DATA_PRECISION


def set_backward_precision(p):
    global BWD_DATA_PRECISION
    if p not in ("f32", "f16x3"):
        raise ValueError("data-gradient precision must be 'f32' or 'f16x3', got %r" % (p,))
    if p != BWD_DATA_PRECISION:
        GENERATION[0] += 1
    BWD_DATA_PRECISION = p


def bump_generation():
    GENERATION[0] += 1


def _stream(t):
    return _lib.current_stream_ptr(t.device)


def _f32c(t, what):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise ValueError("%s: fp32 GPU tensor required (zeroshape_amd has no CPU path)" % what)
    return t if t.is_contiguous() else t.contiguous()


# ---- scratch buffers (grow-only, one per device and purpose; launches on one stream serialise) ----
_SCRATCH = {}


SCRATCH_GENERATION = [0]     # bumped when a workspace is (re)allocated: a captured step holds the old address


_WS_BYTES = {}      # (query, args) -> bytes: the *_workspace_bytes queries are pure functions of their arguments


def _ws_bytes(query, *args):
    """A workspace-size query of the C ABI, asked once per argument tuple (a training step made 459 such calls - a quarter of
    its C-ABI calls - for sizes that never change)."""
    key = (query,) + args
    n = _WS_BYTES.get(key)
    if n is None:
        n = _WS_BYTES[key] = int(getattr(_lib.load(), query)(*args))
    return n


def scratch(device, name, nbytes):
    key = (str(device), name)
    n = (int(nbytes) + 3) // 4
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.empty(max(n, 1), dtype=torch.float32, device=device)
        _SCRATCH[key] = buf
        SCRATCH_GENERATION[0] += 1
    return buf


def _ceil4(n):
    return (n + 3) // 4 * 4


def _stamp(w):
    # ... and the switches that decide WHICH forms of a packed operand are kept current (in mode 2 the re-pack
    # writes only the fp16 halves; a tool or test that flips PRESPLIT_ALL / INLINE_SPLIT at run time must find every record
    # stale, not a fresh-looking stamp over an fp32 operand nobody has rewritten since)
    return (w.data_ptr(), w._version, GENERATION[0], _operand_forms())


def _operand_forms():
    g = globals()
    return (bool(g.get("PRESPLIT_ALL", True)), bool(g.get("INLINE_SPLIT", True)), FWD_CONV_PRECISION == "f16x3", BWD_DATA_PRECISION == "f16x3")


_STD = {}       # id(weight) -> [weakref, eps, buffer, stamp]: standardised weights in persistent buffers


def standardize(weight, eps):
    """timm StdConv2d weight standardisation of a Parameter, refreshed when the weight changes."""
    import weakref
    rec = _STD.get(id(weight))
    if rec is None or rec[0]() is not weight or rec[1] != float(eps):
        rec = [weakref.ref(weight, lambda _r, k=id(weight): _STD.pop(k, None)), float(eps),
               torch.empty_like(weight.detach()), None]
        _STD[id(weight)] = rec
    if rec[3] != _stamp(weight):
        lib = _lib.load()
        w = weight.detach()
        with _lib.on(w.device):
            _lib.check(lib.zs_standardize_weight(_lib.ptr(w), _lib.ptr(rec[2]), w.shape[0], w[0].numel(), float(eps),
                                                 _stream(w)), "zs_standardize_weight")
        rec[3] = _stamp(weight)
    return rec[2]


_STD_TABLE = {}     # device -> (signature, entry table, row prefix, entries, rows): the launch table of standardize_all


def standardize_all(pairs):
    """standardize() for a list of (weight, eps) in ONE launch (zs_standardize_weight_multi): the stale ones of the list go
    through a cached device table - after an optimiser step that is every StdConv weight of the model (52 launches before)."""
    import numpy as np
    import weakref
    todo, seen = [], set()
    for weight, eps in pairs:
        rec = _STD.get(id(weight))
        if rec is None or rec[0]() is not weight or rec[1] != float(eps):
            rec = [weakref.ref(weight, lambda _r, k=id(weight): _STD.pop(k, None)), float(eps),
                   torch.empty_like(weight.detach()), None]
            _STD[id(weight)] = rec
        if rec[3] != _stamp(weight) and id(weight) not in seen:
            seen.add(id(weight))
            todo.append((weight, rec))
    if not todo:
        return
    if len(todo) == 1:
        standardize(todo[0][0], todo[0][1][1])
        return
    lib = _lib.load()
    device = todo[0][0].device
    sig = tuple((w.data_ptr(), rec[2].data_ptr(), w.shape[0], w[0].numel(), rec[1]) for w, rec in todo)
    cached = _STD_TABLE.get(device)
    if cached is None or cached[0] != sig:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the set of standardised weights changed inside a stream capture; call "
                               "zeroshape_amd.nn.autograd.refresh_packs(device) before capturing")
        dt = np.dtype([("w", "<u8"), ("out", "<u8"), ("rows", "<i4"), ("n", "<i4"), ("eps", "<f4"), ("pad", "<i4")])
        assert dt.itemsize == 32
        tab = np.zeros(len(sig), dt)
        prefix, total = [], 0
        for i, e in enumerate(sig):
            tab[i] = e + (0,)
            prefix.append(total)
            total += e[2]
        cached = (sig, torch.from_numpy(tab.view(np.uint8).reshape(-1)).to(device),
                  torch.tensor(prefix, dtype=torch.int32).to(device), len(sig), total)
        _STD_TABLE[device] = cached
    _, tab_d, pre_d, n, total = cached
    with _lib.on(device):
        _lib.check(lib.zs_standardize_weight_multi(_lib.ptr(tab_d), _lib.ptr(pre_d), n, total, _lib.current_stream_ptr(device)),
                   "zs_standardize_weight_multi")
    for w, rec in todo:
        rec[3] = _stamp(w)


# ---- packed GEMM operands: persistent buffers, re-packed for ALL layers in one launch ----
class _PackRec(object):
    # split / split_stamp: the operand's fp16 halves (optim.amp) and the stamp of the pack they were made from - a re-pack
    # (in-place weight update, load_state_dict, optimiser step) moves `stamp` and so invalidates the split with it; the
    # buffer dies with the record (a cache keyed by packed.data_ptr() + generation served stale weights)
    # inline_split: the multi re-pack writes this operand's halves itself (zs_pack_conv_weight_multi_split)
    __slots__ = ("ref", "key", "packed", "stamp", "dims", "split", "split_stamp", "inline_split", "__weakref__")


_PACKS = {}     # (id(weight), cin0, cin, dgrad, std_eps) -> _PackRec
_PACK_TABLE = {}    # device -> (signature, table tensors)
_PACK_EPOCH = [0]   # bumped whenever _PACKS gains or loses a record
_PACK_FAST = {}     # device -> dict(epoch, generation, recs, std): the last complete re-pack, for its replay


def _drop_pack(key):
    if _PACKS.pop(key, None) is not None:
        _PACK_EPOCH[0] += 1


def clear_pack_cache():
    _PACKS.clear()
    _STD.clear()
    _STD_TABLE.clear()
    _PACK_TABLE.clear()
    _PACK_FAST.clear()
    _PACK_EPOCH[0] += 1


def _pack_dims(w, cin, dgrad):
    cout, cintot = w.shape[0], w.shape[1]
    kh, kw = (w.shape[2], w.shape[3]) if w.dim() == 4 else (1, 1)
    taps = kh * kw
    kc, n = (_ceil4(cout), cin) if dgrad else (_ceil4(cin), cout)
    return cout, cintot, kh, kw, taps, (taps * kc + 15) // 16 * 16, (n + 127) // 128 * 128


INLINE_SPLIT = os.environ.get("ZS_TRAIN_INLINE_SPLIT", "1") != "0"      # A/B switch


def _inline_split_mode():
    """0: re-pack only.  1: the re-pack also writes the fp16 halves.  2: ... and leaves the fp32 operands of those entries
    alone (forward AND data gradients read the halves: nothing reads the fp32 form until the precision changes, which
    bumps the generation and so re-packs everything)."""
    if not (INLINE_SPLIT and _amp_splits_operands()):
        return 0
    return 2 if FWD_CONV_PRECISION == "f16x3" and BWD_DATA_PRECISION == "f16x3" else 1


def _launch_multi_pack(lib, device, recs):
    _, tab_d, ce_d, cs_d, n, split_d, _splits = _PACK_TABLE[device]
    with _lib.on(device):
        if split_d is None:
            _lib.check(lib.zs_pack_conv_weight_multi(_lib.ptr(tab_d), _lib.ptr(ce_d), _lib.ptr(cs_d), n,
                                                     _lib.current_stream_ptr(device)), "zs_pack_conv_weight_multi")
        else:
            _lib.check(lib.zs_pack_conv_weight_multi_split(_lib.ptr(tab_d), _lib.ptr(ce_d), _lib.ptr(cs_d), n,
                                                           _lib.ptr(split_d), 1 if _PACK_TABLE[device][0][-1] == 2 else 0,
                                                           _lib.current_stream_ptr(device)),
                       "zs_pack_conv_weight_multi_split")


def _mark_inline_splits(device, recs):
    """After the launch and the new stamps: the halves written by the re-pack are those of the current operand."""
    if _PACK_TABLE[device][5] is not None:
        for rec in recs:
            if rec.inline_split and rec.split is not None:
                rec.split_stamp = rec.stamp


def _refresh_all_packs(device):
    """Re-pack every registered operand on `device` whose weight changed, in ONE launch."""
    import numpy as np
    lib = _lib.load()
    # The state after an optimiser step (every step of a training run): the same records as the last time, all of
    # them stale because the generation moved.  Replay that re-pack - standardise the StdConv weights into their
    # persistent buffers, launch over the cached table - without rebuilding the entry list.
    fast = _PACK_FAST.get(device)
    if fast is not None and fast["epoch"] == _PACK_EPOCH[0] and fast["generation"] != GENERATION[0]:
        live = [(rec, rec.ref()) for rec in fast["recs"]]
        if all(w is not None and w.data_ptr() == ptr for (rec, w), ptr in zip(live, fast["ptrs"])):
            standardize_all([(w, rec.key[4]) for rec, w in live if rec.key[4] is not None])
            if _PACK_TABLE[device][0][-1] == _inline_split_mode():
                _launch_multi_pack(lib, device, [rec for rec, _ in live])
                for rec, w in live:
                    rec.stamp = _stamp(w)
                _mark_inline_splits(device, [rec for rec, _ in live])
                fast["generation"] = GENERATION[0]
                return
    recs = []
    for key, rec in list(_PACKS.items()):
        w = rec.ref()
        if w is None:
            _drop_pack(key)
        elif w.device == device and rec.stamp != _stamp(w):
            recs.append((rec, w))
    if not recs:
        return
    complete = len(recs) == sum(1 for rec in _PACKS.values() if rec.ref() is not None and rec.ref().device == device)
    entries = []
    standardize_all([(w, rec.key[4]) for rec, w in recs if rec.key[4] is not None and w.device == device])
    for rec, w in recs:
        _, cin0, cin, dgrad, std_eps = rec.key
        src = standardize(w, std_eps) if std_eps is not None else w.detach()
        cout, cintot, kh, kw, taps, K16, NPad = rec.dims
        entries.append((src.data_ptr(), rec.packed.data_ptr(), cout, cin, cin0, cintot * taps, taps, 1 if dgrad else 0,
                        K16, NPad))
    mode = _inline_split_mode()
    sig = tuple(entries) + (mode,)
    cached = _PACK_TABLE.get(device)
    if cached is None or cached[0] != sig:
        if torch.cuda.is_current_stream_capturing():
            # the table would travel by a host-to-device copy from pageable memory: not capturable (the graph
            # would keep the host address).  refresh_packs() before the capture builds it.
            raise RuntimeError("the set of packed operands changed inside a stream capture; call "
                               "zeroshape_amd.nn.autograd.refresh_packs(device) before capturing")
        dt = np.dtype([("src", "<u8"), ("dst", "<u8")] + [(n, "<i4") for n in ("Cout", "Cin", "cin0", "ld", "taps", "dgrad",
                                                                             "K16", "NPad")])
        assert dt.itemsize == 48
        tab = np.zeros(len(entries), dt)
        ce, cs = [], []
        for i, e in enumerate(entries):
            tab[i] = e
            n_chunks = lib.zs_pack_entry_chunks(e[2], e[3], e[6], e[7], e[8], e[9])
            _lib.check(1 if n_chunks > 0 else 0, "zs_pack_entry_chunks")
            starts = np.arange(n_chunks, dtype=np.uint64)
            ce.append(np.full(len(starts), i, np.int32))
            cs.append(starts)
        ce, cs = np.concatenate(ce), np.concatenate(cs)
        # optim.amp: the halves of every operand whose tiles hold whole K = 16 groups leave the same launch (split_d: one
        # pointer per entry, 0 = the operand is split by _presplit_all's launch behind this one)
        split_d, splits = None, []
        if mode:
            ptrs = []
            for (rec, _), e in zip(recs, entries):
                rec.inline_split = bool(lib.zs_pack_entry_inline_split(e[2], e[3], e[6], e[7]))
                if rec.inline_split:
                    if rec.split is None or rec.split.numel() != rec.packed.numel() or rec.split.device != device:
                        rec.split = torch.empty_like(rec.packed)
                    ptrs.append(rec.split.data_ptr())
                    splits.append(rec.split)
                else:
                    ptrs.append(0)
            split_d = torch.tensor(ptrs, dtype=torch.int64).to(device)
        else:
            for rec, _ in recs:
                rec.inline_split = False
        cached = (sig, torch.from_numpy(tab.view(np.uint8).reshape(-1)).to(device), torch.from_numpy(ce).to(device),
                  torch.from_numpy(cs.view(np.int64)).to(device), len(ce), split_d, splits)
        _PACK_TABLE[device] = cached
    _launch_multi_pack(lib, device, [rec for rec, _ in recs])
    for rec, w in recs:
        rec.stamp = _stamp(w)
    _mark_inline_splits(device, [rec for rec, _ in recs])
    if complete:
        _PACK_FAST[device] = dict(epoch=_PACK_EPOCH[0], generation=GENERATION[0], recs=[rec for rec, _ in recs],
                                  ptrs=[w.data_ptr() for _, w in recs])
    else:
        _PACK_FAST.pop(device, None)


_refresh_all_packs_only = _refresh_all_packs


def _refresh_all_packs(device):      # noqa: F811 - the re-pack, then (optim.amp) the split of every operand
    _refresh_all_packs_only(device)
    if _amp_splits_operands():
        _presplit_all(device)


def refresh_packs(device):
    """Re-pack every registered operand on `device` now (and leave the launch table of that set cached): what a
    stream capture of a training step must do first, so that the re-pack inside the capture finds its table."""
    bump_generation()
    _refresh_all_packs(torch.device(device) if not isinstance(device, torch.device) else device)


def _pack(weight, cin0, cin, dgrad, std_eps=None):
    """Packed GEMM operand of `weight` (torch layout [Cout, CinTot(, kh, kw)], optionally
    standardised first) for the forward product (dgrad=False) or the data gradient.  Operands live
    in persistent buffers; the first use of a layer packs it alone, afterwards a stale operand
    triggers ONE launch that re-packs every registered operand whose weight changed (the state after
    an optimiser step), instead of ~450 small launches per training step."""
    import weakref
    key = (id(weight), cin0, cin, bool(dgrad), std_eps)
    rec = _PACKS.get(key)
    if rec is not None and rec.ref() is weight:
        if rec.stamp != _stamp(weight):
            _refresh_all_packs(weight.device)
        return rec.packed
    lib = _lib.load()
    w = standardize(weight, std_eps) if std_eps is not None else weight.detach()
    dims = _pack_dims(w, cin, dgrad)
    cout, cintot, kh, kw, taps, K16, NPad = dims
    rec = _PackRec()
    rec.ref = weakref.ref(weight, lambda _r, k=key: _drop_pack(k))
    rec.key, rec.dims = key, dims
    rec.packed = torch.empty(K16 * NPad, dtype=torch.float32, device=w.device)
    rec.split, rec.split_stamp, rec.inline_split = None, None, False
    rec.packed._zs_rec = weakref.ref(rec)              # _conv_launch finds the record (and its split) from the operand
    with _lib.on(w.device):
        _lib.check(lib.zs_pack_conv_weight(_lib.ptr(w), _lib.ptr(rec.packed), cout, cin, cin0, cintot, kh, kw,
                                           1 if dgrad else 0, _stream(w)), "zs_pack_conv_weight")
    rec.stamp = _stamp(weight)
    _PACKS[key] = rec
    _PACK_EPOCH[0] += 1
    return rec.packed


def _out_size(n, k, stride, padding):
    if padding == "same":
        out = -(-n // stride)
        return out, max((out - 1) * stride + k - n, 0) // 2
    return (n + 2 * padding - k) // stride + 1, padding


_CONV_W_PRESPLIT = 128
_PRESPLIT_TABLE = {}    # device -> (signature, device arrays of zs_conv2d_presplit_weight_multi, n, total pairs, splits)
# ONE launch right behind the re-pack splits every registered operand (zs_conv2d_presplit_weight_multi), so every split-fp16
# kernel reads ready halves and the 40 per-layer split launches go.  Round 3 measured this SLOWER (29.75 -> 30.1 ms: the
# small-tile kernels are latency-bound and the split of 192 M parameters moves 1.5 GB); since round 4 the few-row pointwise
# layers with a split operand take the streaming GEMM kernel (csrc/nn_gemm_stream.hip; batch 4 = 788 token rows) and it pays:
# optim.amp step 29.71 -> 28.87 ms on the same box.  ZS_TRAIN_PRESPLIT_ALL=0: only the >= 192-tile layers, per layer.
PRESPLIT_ALL = os.environ.get("ZS_TRAIN_PRESPLIT_ALL", "1") != "0"


def _amp_splits_operands():
    return PRESPLIT_ALL and (FWD_CONV_PRECISION == "f16x3" or BWD_DATA_PRECISION == "f16x3")


def _presplit_all(device):
    """Split every registered packed operand on `device` (call right after they were re-packed)."""
    lib = _lib.load()
    recs = [rec for rec in _PACKS.values() if rec.ref() is not None and rec.packed.device == device and
            not (rec.inline_split and rec.split is not None and rec.split_stamp == rec.stamp)]   # (split by the re-pack itself)
    if not recs:
        return
    sig = tuple((rec.packed.data_ptr(), rec.packed.numel()) for rec in recs)
    cached = _PRESPLIT_TABLE.get(device)
    if cached is None or cached[0] != sig:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the set of packed operands changed inside a stream capture; call "
                               "zeroshape_amd.nn.autograd.refresh_packs(device) before capturing")
        splits, src, dst, cps, prefix, tot = [], [], [], [], [0], 0
        for rec in recs:
            K16, NPad = rec.dims[5], rec.dims[6]
            sp = rec.split
            if sp is None or sp.numel() != rec.packed.numel() or sp.device != device:
                sp = torch.empty_like(rec.packed)
            splits.append(sp)
            src.append(rec.packed.data_ptr()); dst.append(sp.data_ptr()); cps.append(NPad)
            tot += K16 // 16 * 2 * NPad
            prefix.append(tot)
        cached = (sig, torch.tensor(src, dtype=torch.int64).to(device), torch.tensor(dst, dtype=torch.int64).to(device),
                  torch.tensor(cps, dtype=torch.int32).to(device), torch.tensor(prefix, dtype=torch.int64).to(device),
                  len(recs), tot, splits)
        _PRESPLIT_TABLE[device] = cached
    _, src_d, dst_d, cp_d, pre_d, n, tot, splits = cached
    with _lib.on(device):
        _lib.check(lib.zs_conv2d_presplit_weight_multi(_lib.ptr(src_d), _lib.ptr(dst_d), _lib.ptr(cp_d), _lib.ptr(pre_d), n, tot,
                                                       _lib.current_stream_ptr(device)), "zs_conv2d_presplit_weight_multi")
    for rec, sp in zip(recs, splits):
        rec.split, rec.split_stamp = sp, rec.stamp
# optim.amp: layers large enough for the LDS-DMA GEMM kernel or the 3x3 input-patch kernels (>= 192 tiles of 128 x 128,
# channels a multiple of 16, no input affine) get their packed operand split into fp16 halves once per optimiser step
# (zs_conv2d_presplit_weight), which is what those kernels consume; the other layers split on the fly as before.
PRESPLIT_MIN_TILES = int(os.environ.get("ZS_TRAIN_PRESPLIT_MIN_TILES", "192"))


def _rec_of(packed):
    ref = getattr(packed, "_zs_rec", None)
    return ref() if ref is not None else None


def _split_of(packed):
    """The operand's current fp16 halves, or None (never split, or split from an older pack)."""
    rec = _rec_of(packed)
    if rec is not None and rec.split is not None and rec.split_stamp == rec.stamp and rec.split.numel() == packed.numel():
        return rec.split
    return None


def _presplit(packed, C, Co, kh, kw, device):
    lib = _lib.load()
    split = _split_of(packed)
    if split is None:
        rec = _rec_of(packed)
        split = rec.split if rec is not None and rec.split is not None and rec.split.numel() == packed.numel() \
            else torch.empty_like(packed)
        with _lib.on(device):
            _lib.check(lib.zs_conv2d_presplit_weight(_lib.ptr(packed), _lib.ptr(split), C, Co, kh, kw,
                                                     _lib.current_stream_ptr(device)), "zs_conv2d_presplit_weight")
        if rec is not None:                    # an operand without a record (none today) is split on every use
            rec.split, rec.split_stamp = split, rec.stamp
    return split


def _conv_launch(x, packed, shift, res1, res2, out, kh, kw, stride, pt, pl, flags, in_scale, in_shift, act):
    lib = _lib.load()
    B, H, W, C = x.shape
    _, Ho, Wo, Co = out.shape
    if flags & _CONV_F16X3:
        split = _split_of(packed) if _amp_splits_operands() else None
        if split is not None:
            packed = split                         # split with every other operand right after the re-pack
            flags |= _CONV_W_PRESPLIT
        elif not (flags & _CONV_IN_DILATE2) and C % 16 == 0 and in_scale == 1.0 and in_shift == 0.0 and \
                -(-(B * Ho * Wo) // 128) * -(-Co // 128) >= PRESPLIT_MIN_TILES:
            packed = _presplit(packed, C, Co, kh, kw, x.device)
            flags |= _CONV_W_PRESPLIT
    with _lib.on(x.device):
        _lib.check(lib.zs_conv2d_nhwc(_lib.ptr(x), _lib.ptr(packed), None, _lib.ptr(shift), _lib.ptr(res1),
                                      _lib.ptr(res2), _lib.ptr(out), B, H, W, C, Ho, Wo, Co, kh, kw, stride, pt, pl,
                                      flags, float(in_scale), float(in_shift), act, _stream(x)), "zs_conv2d_nhwc")


def column_sum(x2d, scale=1.0):
    """[rows, C] -> [C]."""
    lib = _lib.load()
    rows, C = x2d.shape
    out = torch.empty(C, dtype=torch.float32, device=x2d.device)
    ws = scratch(x2d.device, "colsum", _ws_bytes("zs_column_sum_workspace_bytes", rows, C))
    with _lib.on(x2d.device):
        _lib.check(lib.zs_column_sum(_lib.ptr(x2d), _lib.ptr(out), rows, C, float(scale), _lib.ptr(ws), _stream(x2d)),
                   "zs_column_sum")
    return out


def _act_backward(dy, ref, act, beta=0.0):
    lib = _lib.load()
    dx = torch.empty_like(dy)
    with _lib.on(dy.device):
        _lib.check(lib.zs_act_backward(_lib.ptr(dy), _lib.ptr(ref), _lib.ptr(dx), dy.numel(), act, float(beta),
                                       _stream(dy)), "zs_act_backward")
    return dx


def _pad_channels(x, cpad):
    lib = _lib.load()
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty(*x.shape[:-1], cpad, dtype=torch.float32, device=x.device)
    with _lib.on(x.device):
        _lib.check(lib.zs_nchw_to_nhwc(_lib.ptr(x), None, _lib.ptr(y), rows, C, 1, cpad, _stream(x)), "zs_nchw_to_nhwc")
    return y


# Segmented backward (the captured training step on more than one GPU, model/shape_engine.py).  One hipGraph of forward +
# backward leaves the bucket all-reduce nothing to overlap with: every gradient appears "at once" when the replay ends.  So
# the forward marks a few places where the network can be CUT (A.cut / A.segment_break, no-ops unless a segmenting capture
# is active): a cut tensor is handed to the consumers of LATER segments as a detached leaf, so torch.autograd can run the
# backward pass one segment at a time - backward_segment(s) starts from the cut tensors segment s produced, with the
# gradients their leaves collected, and stops at the leaves of earlier segments - each segment captured into its own
# hipGraph.  Between two replays the engine packs and all-reduces the buckets whose gradients are final while the next
# segment's backward runs (parallel.GradReducer.launch_done).  The arithmetic is the unsegmented step's: a cut adds no
# operation, and a tensor with consumers on both sides of a cut receives the same two-term gradient sum.
SEGMENTS = None               # {"index": current segment, "cuts": [(producer segment, tensor, detached leaf)]} while segmenting


def begin_segments():
    global SEGMENTS
    SEGMENTS = {"index": 0, "cuts": []}


def end_segments():
    global SEGMENTS
    SEGMENTS = None


def segment_break():
    """The forward moves on to the next segment (everything computed from here on may only read earlier segments' tensors
    through cut())."""
    if SEGMENTS is not None:
        SEGMENTS["index"] += 1


def cut(*tensors):
    """The tensors as LATER segments must read them: detached leaves that collect their gradient (the same storage - no copy);
    the originals stay usable inside the segment that produced them.  Identity when no segmenting capture is active, and for
    tensors that carry no gradient."""
    out = []
    for x in tensors:
        if SEGMENTS is not None and torch.is_tensor(x) and x.requires_grad:
            leaf = x.detach().requires_grad_()
            SEGMENTS["cuts"].append((SEGMENTS["index"], x, leaf))
            x = leaf
        out.append(x)
    return out[0] if len(out) == 1 else tuple(out)


# Weight gradients on a side stream (see _Conv.backward).  Off unless the caller of backward() switches it on AND joins
# afterwards: a gradient tensor handed to autograd is written later, on another stream.
SIDE_WGRAD = [False]
_SIDE_STREAMS = {}
_SIDE_KEEP = []


def _side_wgrad_stream(device):
    st = _SIDE_STREAMS.get(str(device))
    if st is None:
        st = _SIDE_STREAMS[str(device)] = torch.cuda.Stream(device=device)
    return st


def join_side_wgrads():
    """The current stream of every device with weight gradients in flight waits for them; their operands may be freed."""
    for dev, st in _SIDE_STREAMS.items():
        torch.cuda.current_stream(torch.device(dev)).wait_stream(st)
    _SIDE_KEEP.clear()


class side_wgrads(object):
    """with side_wgrads(on): loss.backward()  - weight gradients overlap the data-gradient chain, joined on exit."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.prev = SIDE_WGRAD[0]
        SIDE_WGRAD[0] = self.on
        return self

    def __exit__(self, *exc):
        SIDE_WGRAD[0] = self.prev
        if self.on:
            join_side_wgrads()
        return False


def segment_has_work(s):
    """Does segment s have anything to differentiate (a cut tensor it produced that collected a gradient)?"""
    return any(seg == s and leaf.grad is not None for seg, _, leaf in SEGMENTS["cuts"])


def backward_segment(s, params, loss=None):
    """Backward pass of segment s: from `loss` (the last segment) and from the cut tensors segment s produced - seeded with
    what their leaves collected in the later segments - to the parameters (those of other segments are unreachable and
    skipped) and the leaves cut by earlier segments.  -> False when the segment has nothing to differentiate."""
    roots, grads = ([loss], [None]) if loss is not None else ([], [])
    for seg, x, leaf in SEGMENTS["cuts"]:
        if seg == s and leaf.grad is not None:
            roots.append(x)
            grads.append(leaf.grad)
    inputs = list(params) + [leaf for seg, _, leaf in SEGMENTS["cuts"] if seg < s]
    if not roots or not inputs:
        return False
    torch.autograd.backward(roots, grads, inputs=inputs)
    return True


def posenc3d(points, L):
    """points [..., 3] (no gradient: the decoder's query points are data) -> NeRF encoding [..., pad4(3 + 6 L)], the
    reference's get_embedder(L, 3) (utils/layers.py:8-53), channels beyond 3 + 6 L zero (the channel count the GEMM stages)."""
    lib = _lib.load()
    pts = _f32c(points.detach(), "posenc3d input")
    cpad = (3 + 6 * L + 3) // 4 * 4
    out = torch.empty(*pts.shape[:-1], cpad, dtype=torch.float32, device=pts.device)
    with _lib.on(pts.device):
        _lib.check(lib.zs_posenc3d(_lib.ptr(pts), pts.numel() // 3, int(L), _lib.ptr(out), cpad, _stream(pts)), "zs_posenc3d")
    return out


class _PadChannels(torch.autograd.Function):
    """[..., C] -> [..., cpad] with zero padding channels (the channel count the convolution engine stages)."""

    @staticmethod
    def forward(ctx, x, cpad):
        ctx.C = x.shape[-1]
        return _pad_channels(_f32c(x, "pad input"), cpad)

    @staticmethod
    def backward(ctx, dy):
        return dy[..., :ctx.C].contiguous(), None


def pad_channels(x, cpad):
    return x if x.shape[-1] == cpad else _PadChannels.apply(x, cpad)


class _Conv(torch.autograd.Function):
    """y = act(conv(T(x), W[:, cin0:cin0+cin]) + bias + res1 + res2), T = input ReLU / affine."""

    @staticmethod
    def forward(ctx, x, weight, bias, res1, res2, cfg):
        x = _f32c(x, "conv input")
        stride, padding, act = cfg["stride"], cfg["padding"], cfg["act"]
        assert act in (ACT_NONE, ACT_RELU, ACT_RELU_CLAMP1), "fuse only ReLU-type activations when training"
        B, H, W, Cx = x.shape
        cout = weight.shape[0]
        kh, kw = (weight.shape[2], weight.shape[3]) if weight.dim() == 4 else (1, 1)
        cin0 = cfg.get("cin0", 0)
        cin = cfg.get("cin") or weight.shape[1]
        assert Cx == _ceil4(cin), "input has %d channels, layer expects %d (padded to 4)" % (Cx, cin)
        std_eps = cfg.get("std_eps")
        Ho, pt = _out_size(H, kh, stride, padding)
        Wo, pl = _out_size(W, kw, stride, padding)
        out = torch.empty(B, Ho, Wo, cout, dtype=torch.float32, device=x.device)
        flags = (_CONV_IN_RELU if cfg.get("in_relu") else 0) | (_CONV_F16X3 if FWD_CONV_PRECISION == "f16x3" else 0)
        _conv_launch(x, _pack(weight, cin0, cin, False, std_eps), None if bias is None else bias.detach(),
                     None if res1 is None else _f32c(res1, "res1"), None if res2 is None else _f32c(res2, "res2"),
                     out, kh, kw, stride, pt, pl, flags, cfg.get("in_scale", 1.0), cfg.get("in_shift", 0.0), act)
        ctx.cfg = dict(cfg, pt=pt, pl=pl, kh=kh, kw=kw, cin0=cin0, cin=cin)
        ctx.has = (bias is not None, res1 is not None, res2 is not None)
        ctx.save_for_backward(x, weight, out if act != ACT_NONE else None)
        return (out, x.view_as(x)) if cfg.get("fork") else out

    @staticmethod
    def backward(ctx, dy, dpass=None):
        lib = _lib.load()
        x, weight, out = ctx.saved_tensors
        if dpass is not None:
            dpass = _f32c(dpass, "conv pass-through grad")
        cfg = ctx.cfg
        kh, kw, stride, pt, pl = cfg["kh"], cfg["kw"], cfg["stride"], cfg["pt"], cfg["pl"]
        cin0, cin, act = cfg["cin0"], cfg["cin"], cfg["act"]
        in_relu, in_scale, in_shift = bool(cfg.get("in_relu")), cfg.get("in_scale", 1.0), cfg.get("in_shift", 0.0)
        std_eps = cfg.get("std_eps")
        g = _f32c(dy, "conv grad")
        if act != ACT_NONE:
            g = _act_backward(g, out, act)
        B, Ho, Wo, cout = g.shape
        _, H, W, Cx = x.shape
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        want_b = ctx.has[0] and need_b
        gp = g if cout % 4 == 0 else _pad_channels(g, _ceil4(cout))
        dw = db = None
        if want_b and not need_w:
            db = column_sum(g.view(-1, cout))
        if need_w:
            sub = cin != weight.shape[1]
            dw = torch.zeros_like(weight) if sub else torch.empty_like(weight)
            if want_b:          # the bias gradient rides on the weight-gradient kernel (it stages dY anyway)
                db = torch.empty(cout, dtype=torch.float32, device=x.device)
            dws = torch.empty_like(weight) if std_eps is not None else None
            flags = (_CONV_IN_RELU if in_relu else 0) | (_CONV_F16X3 if wgrad_precision() == "f16x3" else 0)
            # SIDE_WGRAD (the engine's backward passes): nothing in the backward pass waits for a weight gradient, so it goes to a
            # side stream behind everything enqueued so far and the data-gradient chain continues at once - the weight-gradient
            # launches fill the tails of the chain's launches (join_side_wgrads() before anyone reads a gradient)
            side = _side_wgrad_stream(x.device) if SIDE_WGRAD[0] else None
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(x.device))
                _SIDE_KEEP.append((x, gp, weight, dw, db, dws))       # alive until the join: the side stream reads / writes them
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                ws = scratch(x.device, "wgrad_side" if side is not None else "wgrad",
                             _ws_bytes("zs_conv2d_wgrad_workspace_bytes", B, Ho, Wo, Cx, cout, kh, kw))
                with _lib.on(x.device):
                    _lib.check(lib.zs_conv2d_wgrad(_lib.ptr(x), _lib.ptr(gp), _lib.ptr(dw), _lib.ptr(db), _lib.ptr(ws), B, H, W, Cx, Ho,
                                                   Wo, cout, kh, kw, stride, pt, pl, flags, float(in_scale),
                                                   float(in_shift), cin, cin0, weight.shape[1], 0, _stream(x)),
                               "zs_conv2d_wgrad")
                if std_eps is not None:
                    with _lib.on(x.device):
                        _lib.check(lib.zs_standardize_weight_bwd(_lib.ptr(weight.detach()), _lib.ptr(dw), _lib.ptr(dws),
                                                                 weight.shape[0], weight[0].numel(), float(std_eps),
                                                                 _stream(x)), "zs_standardize_weight_bwd")
            if std_eps is not None:
                dw = dws
        dx = None
        if need_x:
            assert stride in (1, 2), "data gradient: stride 1 or 2"
            if cin <= 4 and cout % 4 == 0 and kh * kw * cout * 16 <= 160 * 1024:
                # a network stem: direct gather kernel instead of a GEMM with 3 useful columns
                w_used = standardize(weight, std_eps) if std_eps is not None else weight.detach()
                dx = torch.empty(B, H, W, Cx, dtype=torch.float32, device=x.device)
                with _lib.on(x.device):
                    _lib.check(lib.zs_conv2d_dgrad_small_cin(_lib.ptr(g), _lib.ptr(w_used), _lib.ptr(dx), B, H, W, Cx, Ho, Wo,
                                                             cout, kh, kw, stride, pt, pl, cin, cin0, weight.shape[1],
                                                             float(in_scale), _stream(x)), "zs_conv2d_dgrad_small_cin")
            else:
                dx = torch.empty(B, H, W, cin, dtype=torch.float32, device=x.device)
                flags = (_CONV_IN_DILATE2 if stride == 2 else 0) | (_CONV_F16X3 if BWD_DATA_PRECISION == "f16x3" else 0)
                fused = dpass is not None and cin == Cx and not in_relu     # the pass-through's gradient: the GEMM's residual
                _conv_launch(gp, _pack(weight, cin0, cin, True, std_eps), None, dpass if fused else None, None, dx, kh, kw, 1,
                             kh - 1 - pt, kw - 1 - pl, flags, in_scale, 0.0, ACT_NONE)
                if fused:
                    dpass = None
                if cin != Cx:                       # the input carried zero padding channels
                    dx = _pad_channels(dx, Cx)
            if in_relu:
                dx = _act_backward(dx, x, ACT_RELU)
        if dpass is not None:
            dx = dpass if dx is None else dx + dpass
        return dx, dw, db, (g if ctx.has[1] else None), (g if ctx.has[2] else None), None


def conv2d(x, weight, bias=None, stride=1, padding=0, act=ACT_NONE, in_relu=False, in_scale=1.0, in_shift=0.0,
           res1=None, res2=None, std_eps=None, cin0=0, cin=None, fork=False):
    """fork=True: returns (conv(x), x) - see "Forks" below; x's other consumer takes the returned x."""
    cfg = dict(stride=stride, padding=padding, act=act, in_relu=in_relu, in_scale=in_scale, in_shift=in_shift,
               std_eps=std_eps, cin0=cin0, cin=cin, fork=bool(fork) and FUSE_FORKS)
    y = _Conv.apply(x, weight, bias, res1, res2, cfg)
    return (y, x) if fork and not FUSE_FORKS else y


def linear(x, weight, bias=None, act=ACT_NONE, res1=None, in_scale=1.0, cin0=0, cin=None):
    """x [..., CinP] -> [..., Cout]."""
    lead = x.shape[:-1]
    n = 1
    for d in lead:
        n *= d
    y = conv2d(x.reshape(1, 1, n, x.shape[-1]), weight, bias, act=act, in_scale=in_scale, cin0=cin0, cin=cin,
               res1=None if res1 is None else res1.reshape(1, 1, n, weight.shape[0]))
    return y.view(*lead, weight.shape[0])


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, beta):
        lib = _lib.load()
        x = _f32c(x, "activation input")
        y = torch.empty_like(x)
        with _lib.on(x.device):
            _lib.check(lib.zs_act_forward(_lib.ptr(x), _lib.ptr(y), x.numel(), act, float(beta), _stream(x)),
                       "zs_act_forward")
        ctx.act, ctx.beta = act, beta
        ctx.save_for_backward(y if act in (ACT_RELU, ACT_RELU_CLAMP1) else x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (ref,) = ctx.saved_tensors
        return _act_backward(_f32c(dy, "activation grad"), ref, ctx.act, ctx.beta), None, None


def gelu(x):
    return _Act.apply(x, ACT_GELU, 0.0)


def softplus(x, beta=100.0):
    return _Act.apply(x, ACT_SOFTPLUS, beta)


def relu(x):
    return _Act.apply(x, ACT_RELU, 0.0)


# Forks.  A tensor with two consumers (x -> [norm / conv -> branch] and x -> residual add) receives two gradients, which
# the autograd engine sums with an elementwise launch of its own (~95 per training step of the shape graph, 4 us each).
# With fork=True the first consumer also hands x through (`y, x = op(x, fork=True)`, the second consumer takes THAT x):
# the pass-through's gradient arrives in the op's backward, which adds it inside the kernel that writes dx anyway
# (zs_layer_norm_bwd_add; the data-gradient GEMM's residual operand), and x's producer sees ONE gradient.
FUSE_FORKS = os.environ.get("ZS_TRAIN_FUSE_FORKS", "1") != "0"      # A/B switch: off = the pass-through is x itself


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, fork):
        lib = _lib.load()
        x = _f32c(x, "layer_norm input")
        C = x.shape[-1]
        y = torch.empty_like(x)
        with _lib.on(x.device):
            _lib.check(lib.zs_layer_norm(_lib.ptr(x), _lib.ptr(gamma.detach()), _lib.ptr(beta.detach()), _lib.ptr(y),
                                         x.numel() // C, C, float(eps), _stream(x)), "zs_layer_norm")
        ctx.eps = eps
        ctx.save_for_backward(x, gamma)
        return (y, x.view_as(x)) if fork else y

    @staticmethod
    def backward(ctx, dy, dpass=None):
        lib = _lib.load()
        x, gamma = ctx.saved_tensors
        dy = _f32c(dy, "layer_norm grad")
        add = None if dpass is None else _f32c(dpass, "layer_norm pass-through grad")
        C = x.shape[-1]
        rows = x.numel() // C
        dx = torch.empty_like(x)
        dg = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = scratch(x.device, "ln_bwd", _ws_bytes("zs_layer_norm_bwd_workspace_bytes", rows, C))
        with _lib.on(x.device):
            _lib.check(lib.zs_layer_norm_bwd_add(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(gamma.detach()), _lib.ptr(add),
                                                 _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db), rows, C, float(ctx.eps),
                                                 _lib.ptr(ws), _stream(x)), "zs_layer_norm_bwd_add")
        return dx, dg, db, None, None


def layer_norm(x, gamma, beta, eps=1e-6, fork=False):
    """fork=True: returns (LayerNorm(x), x) - see "Forks" above."""
    if not fork:
        return _LayerNorm.apply(x, gamma, beta, eps, False)
    if not FUSE_FORKS:
        return _LayerNorm.apply(x, gamma, beta, eps, False), x
    return _LayerNorm.apply(x, gamma, beta, eps, True)


ATT_FWD_SPLIT = os.environ.get("ZS_TRAIN_ATT_SPLIT", "1") != "0"      # A/B switch


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, heads):
        lib = _lib.load()
        qkv = _f32c(qkv, "attention input")
        B, L, C3 = qkv.shape
        C = C3 // 3
        out = torch.empty(B, L, C, dtype=torch.float32, device=qkv.device)
        # optim.amp: the split-fp16 forms, like the forward GEMMs around it (batch 4 = 48 (sample, head) pairs: the key-split
        # kernel, 25 -> 9 us per ViT block); the backward pass recomputes the probabilities in fp32 either way
        split = FWD_CONV_PRECISION == "f16x3" and ATT_FWD_SPLIT
        with _lib.on(qkv.device):
            _lib.check((lib.zs_attention_split if split else lib.zs_attention)(
                _lib.ptr(qkv), _lib.ptr(out), B, L, heads, C // heads, _stream(qkv)),
                "zs_attention_split" if split else "zs_attention")
        ctx.heads = heads
        ctx.save_for_backward(qkv)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        (qkv,) = ctx.saved_tensors
        dout = _f32c(dout, "attention grad")
        B, L, C3 = qkv.shape
        heads = ctx.heads
        dqkv = torch.empty_like(qkv)
        ws = scratch(qkv.device, "attn_bwd", _ws_bytes("zs_attention_bwd_workspace_bytes", B, L, heads))
        with _lib.on(qkv.device):
            _lib.check(lib.zs_attention_bwd(_lib.ptr(qkv), _lib.ptr(dout), _lib.ptr(dqkv), _lib.ptr(ws), B, L, heads,
                                            C3 // 3 // heads, _stream(qkv)), "zs_attention_bwd")
        return dqkv, None


def attention(qkv, heads):
    return _Attention.apply(qkv, heads)


class _PointAttention(torch.autograd.Function):
    """ImplFuncAttention's point rows (implicit.py:44-66): softmax over the latent keys + self."""

    @staticmethod
    def forward(ctx, qkv_p, qkv_l, heads):
        lib = _lib.load()
        qkv_p, qkv_l = _f32c(qkv_p, "point qkv"), _f32c(qkv_l, "latent qkv")
        B, M, C3 = qkv_p.shape
        Ll, C = qkv_l.shape[1], C3 // 3
        out = torch.empty(B, M, C, dtype=torch.float32, device=qkv_p.device)
        with _lib.on(qkv_p.device):
            _lib.check(lib.zs_point_attention(_lib.ptr(qkv_p), _lib.ptr(qkv_l), _lib.ptr(out), B, M, Ll, heads,
                                              C // heads, _stream(qkv_p)), "zs_point_attention")
        ctx.heads = heads
        ctx.save_for_backward(qkv_p, qkv_l)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        qkv_p, qkv_l = ctx.saved_tensors
        dout = _f32c(dout, "point attention grad")
        B, M, C3 = qkv_p.shape
        Ll, C, heads = qkv_l.shape[1], C3 // 3, ctx.heads
        dp, dl = torch.empty_like(qkv_p), torch.empty_like(qkv_l)
        ws = scratch(qkv_p.device, "pa_bwd", _ws_bytes("zs_point_attention_bwd_workspace_bytes", B, M, Ll, heads))
        with _lib.on(qkv_p.device):
            _lib.check(lib.zs_point_attention_bwd(_lib.ptr(qkv_p), _lib.ptr(qkv_l), _lib.ptr(dout), _lib.ptr(dp),
                                                  _lib.ptr(dl), 0, _lib.ptr(ws), B, M, Ll, heads, C // heads,
                                                  _stream(qkv_p)), "zs_point_attention_bwd")
        return dp, dl, None


def point_attention(qkv_p, qkv_l, heads):
    return _PointAttention.apply(qkv_p, qkv_l, heads)


def point_attention_probs(qkv_p, qkv_l, heads, attn, weight=1.0, accumulate=False):
    """attn [B,M,Ll] (=|+=) weight * mean over the heads of the points' softmax probabilities over the latent tokens (the
    attention map the reference returns, implicit.py:60-66,277).  No gradient (visualisation only)."""
    lib = _lib.load()
    qp, ql = _f32c(qkv_p.detach(), "qkv points"), _f32c(qkv_l.detach(), "qkv latent")
    B, M, Ll = qp.shape[0], qp.shape[1], ql.shape[1]
    assert attn.shape == (B, M, Ll) and attn.is_contiguous() and attn.dtype == torch.float32
    with _lib.on(qp.device):
        _lib.check(lib.zs_point_attention_probs(_lib.ptr(qp), _lib.ptr(ql), _lib.ptr(attn), B, M, Ll, heads, qp.shape[-1] // (3 * heads),
                                                float(weight), 1 if accumulate else 0, _stream(qp)), "zs_point_attention_probs")
    return attn


def _scaled_rows(x, branch, scale):
    lib = _lib.load()
    B = branch.shape[0]
    y = torch.empty_like(branch)
    with _lib.on(branch.device):
        _lib.check(lib.zs_add_scaled_rows(_lib.ptr(x), _lib.ptr(branch), _lib.ptr(scale), _lib.ptr(y), B,
                                          branch.numel() // B, _stream(branch)), "zs_add_scaled_rows")
    return y


class _AddScaledRows(torch.autograd.Function):
    """x + scale[b] * branch: residual connection under per-sample stochastic depth (timm DropPath)."""

    @staticmethod
    def forward(ctx, x, branch, scale):
        x, branch = _f32c(x, "residual"), _f32c(branch, "branch")
        ctx.save_for_backward(scale)
        return _scaled_rows(x, branch, scale)

    @staticmethod
    def backward(ctx, dy):
        (scale,) = ctx.saved_tensors
        dy = _f32c(dy, "residual grad")
        return dy, _scaled_rows(None, dy, scale), None


def add_scaled_rows(x, branch, scale):
    return _AddScaledRows.apply(x, branch, _f32c(scale, "drop-path scale"))


class _BCELogits(torch.autograd.Function):
    """Loss.shape_loss (utils/loss.py:18-28): mean of w * BCEWithLogits(pred, sdf < 0)."""

    @staticmethod
    def forward(ctx, logits, sdf, thres, weight):
        lib = _lib.load()
        logits, sdf = _f32c(logits, "logits"), _f32c(sdf, "sdf")
        n = logits.numel()
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        ws = scratch(logits.device, "bce", _ws_bytes("zs_bce_logits_workspace_bytes", n))
        with _lib.on(logits.device):
            _lib.check(lib.zs_bce_logits(_lib.ptr(logits), _lib.ptr(sdf), n, float(thres), float(weight),
                                         _lib.ptr(loss), _lib.ptr(ws), _stream(logits)), "zs_bce_logits")
        ctx.thres, ctx.weight = thres, weight
        ctx.save_for_backward(logits, sdf)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        lib = _lib.load()
        logits, sdf = ctx.saved_tensors
        dloss = _f32c(dloss, "loss grad")
        dx = torch.empty_like(logits)
        with _lib.on(logits.device):
            _lib.check(lib.zs_bce_logits_bwd(_lib.ptr(logits), _lib.ptr(sdf), logits.numel(), float(ctx.thres),
                                             float(ctx.weight), _lib.ptr(dloss), _lib.ptr(dx), _stream(logits)),
                       "zs_bce_logits_bwd")
        return dx, None, None, None


def bce_logits(logits, sdf, impt_thres=0.01, impt_weight=1.0):
    return _BCELogits.apply(logits, sdf, impt_thres, impt_weight)


# =============================================================================================
# Encoder layers: BatchNorm (training), GroupNorm, pooling, resampling, layout, geometry
# =============================================================================================
class _BatchNormTrain(torch.autograd.Function):
    """nn.BatchNorm2d in training mode over channels-last x (+ residual, ReLU fused): batch
    statistics, running-stat update (momentum, unbiased variance) as torch does."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, eps, momentum, relu):
        lib = _lib.load()
        x = _f32c(x, "batch_norm input")
        C = x.shape[-1]
        rows = x.numel() // C
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        rstd = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = scratch(x.device, "bn", _ws_bytes("zs_batch_norm_workspace_bytes", rows, C))
        res = None if residual is None else _f32c(residual, "batch_norm residual")
        with _lib.on(x.device):
            _lib.check(lib.zs_batch_norm_train(_lib.ptr(x), _lib.ptr(gamma.detach()), _lib.ptr(beta.detach()),
                                               _lib.ptr(res), _lib.ptr(y), _lib.ptr(running_mean),
                                               _lib.ptr(running_var), _lib.ptr(mean), _lib.ptr(rstd), rows, C,
                                               float(eps), float(momentum), 1 if relu else 0, _lib.ptr(ws),
                                               _stream(x)), "zs_batch_norm_train")
        ctx.relu, ctx.has_res = relu, residual is not None
        ctx.save_for_backward(x, gamma, mean, rstd, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, gamma, mean, rstd, y = ctx.saved_tensors
        dy = _f32c(dy, "batch_norm grad")
        C = x.shape[-1]
        rows = x.numel() // C
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if (ctx.has_res and ctx.relu) else None
        dg = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = scratch(x.device, "bn", _ws_bytes("zs_batch_norm_workspace_bytes", rows, C))
        with _lib.on(x.device):
            _lib.check(lib.zs_batch_norm_bwd(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(y), _lib.ptr(gamma.detach()),
                                             _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(dx), _lib.ptr(dres), _lib.ptr(dg),
                                             _lib.ptr(db), rows, C, _lib.ptr(ws), _stream(x)), "zs_batch_norm_bwd")
        if ctx.has_res and not ctx.relu:
            dres = dy
        return dx, dg, db, dres, None, None, None, None, None


_BN_COUNTERS = [None]      # inside deferred_bn_counters(): the num_batches_tracked buffers to bump on exit


class deferred_bn_counters(object):
    """Within the block batch_norm_train() only notes the layers' `num_batches_tracked` buffers; on exit they are all
    incremented by ONE multi-tensor launch (a training forward of the shape graph has 66 BatchNorm layers: 66 one-element
    kernels, 0.27 ms of a 27 ms step).  Re-entrant: an inner block leaves the flush to the outermost one."""

    def __enter__(self):
        self.outer = _BN_COUNTERS[0] is not None
        if not self.outer:
            _BN_COUNTERS[0] = []
        return self

    def __exit__(self, *exc):
        if not self.outer:
            pending, _BN_COUNTERS[0] = _BN_COUNTERS[0], None
            # flushed whether or not an exception leaves the block: the counters describe running-statistics updates that
            # already happened in place (a BatchNorm with momentum=None averages over exactly this count - ADVICE r05)
            if pending:
                torch._foreach_add_(pending, 1)
        return False


def batch_norm_train(x, bn, relu=False, residual=None):
    """bn: an nn.BatchNorm2d in training mode (its running statistics are updated in place)."""
    momentum = 0.1 if bn.momentum is None else bn.momentum
    y = _BatchNormTrain.apply(x, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var, bn.eps, momentum, relu)
    if bn.num_batches_tracked is not None:
        if _BN_COUNTERS[0] is not None:
            _BN_COUNTERS[0].append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked += 1
    return y


class _GroupNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, groups, eps, relu):
        lib = _lib.load()
        x = _f32c(x, "group_norm input")
        B, H, W, C = x.shape
        y = torch.empty_like(x)
        res = None if residual is None else _f32c(residual, "group_norm residual")
        with _lib.on(x.device):
            _lib.check(lib.zs_group_norm_nhwc(_lib.ptr(x), _lib.ptr(gamma.detach()), _lib.ptr(beta.detach()),
                                              _lib.ptr(res), _lib.ptr(y), B, H * W, C, groups, float(eps),
                                              1 if relu else 0, _stream(x)), "zs_group_norm_nhwc")
        ctx.cfg = (groups, eps, relu, residual is not None)
        ctx.save_for_backward(x, gamma, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, gamma, y = ctx.saved_tensors
        groups, eps, relu, has_res = ctx.cfg
        dy = _f32c(dy, "group_norm grad")
        B, H, W, C = x.shape
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if (has_res and relu) else None
        dg = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = scratch(x.device, "gn_bwd", _ws_bytes("zs_group_norm_bwd_workspace_bytes", B, C))
        with _lib.on(x.device):
            _lib.check(lib.zs_group_norm_bwd(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(y), _lib.ptr(gamma.detach()),
                                             _lib.ptr(dx), _lib.ptr(dres), _lib.ptr(dg), _lib.ptr(db), B, H * W, C,
                                             groups, float(eps), _lib.ptr(ws), _stream(x)), "zs_group_norm_bwd")
        if has_res and not relu:
            dres = dy
        return dx, dg, db, dres, None, None, None


def group_norm(x, gamma, beta, groups=32, eps=1e-5, relu=False, residual=None):
    return _GroupNorm.apply(x, gamma, beta, residual, groups, eps, relu)


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, stride, padding):
        lib = _lib.load()
        x = _f32c(x, "max_pool input")
        B, H, W, C = x.shape
        Ho, pt = _out_size(H, k, stride, padding)
        Wo, pl = _out_size(W, k, stride, padding)
        y = torch.empty(B, Ho, Wo, C, dtype=torch.float32, device=x.device)
        with _lib.on(x.device):
            _lib.check(lib.zs_max_pool_nhwc(_lib.ptr(x), _lib.ptr(y), B, H, W, C, Ho, Wo, k, stride, pt, pl,
                                            _stream(x)), "zs_max_pool_nhwc")
        ctx.cfg = (k, stride, pt, pl, Ho, Wo)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        k, stride, pt, pl, Ho, Wo = ctx.cfg
        dy = _f32c(dy, "max_pool grad")
        B, H, W, C = x.shape
        dx = torch.empty_like(x)
        with _lib.on(x.device):
            _lib.check(lib.zs_max_pool_bwd_nhwc(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dx), B, H, W, C, Ho, Wo, k, stride,
                                                pt, pl, _stream(x)), "zs_max_pool_bwd_nhwc")
        return dx, None, None, None


def max_pool(x, k=3, stride=2, padding=1):
    return _MaxPool.apply(x, k, stride, padding)


class _GlobalMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _f32c(x, "global_mean input")
        B, H, W, C = x.shape
        y = torch.empty(B, C, dtype=torch.float32, device=x.device)
        with _lib.on(x.device):
            _lib.check(lib.zs_global_mean_nhwc(_lib.ptr(x), _lib.ptr(y), B, H * W, C, _stream(x)),
                       "zs_global_mean_nhwc")
        ctx.shape = (B, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        B, H, W, C = ctx.shape
        dy = _f32c(dy, "global_mean grad")
        dx = torch.empty(B, H, W, C, dtype=torch.float32, device=dy.device)
        with _lib.on(dy.device):
            _lib.check(lib.zs_global_mean_bwd_nhwc(_lib.ptr(dy), _lib.ptr(dx), B, H * W, C, _stream(dy)),
                       "zs_global_mean_bwd_nhwc")
        return dx


def global_mean(x):
    return _GlobalMean.apply(x)


class _Upsample2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _f32c(x, "upsample2x input")
        B, H, W, C = x.shape
        y = torch.empty(B, 2 * H, 2 * W, C, dtype=torch.float32, device=x.device)
        with _lib.on(x.device):
            _lib.check(lib.zs_upsample2x_nhwc(_lib.ptr(x), _lib.ptr(y), B, H, W, C, _stream(x)), "zs_upsample2x_nhwc")
        ctx.shape = (B, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        B, H, W, C = ctx.shape
        dy = _f32c(dy, "upsample2x grad")
        dx = torch.empty(B, H, W, C, dtype=torch.float32, device=dy.device)
        with _lib.on(dy.device):
            _lib.check(lib.zs_upsample2x_bwd_nhwc(_lib.ptr(dy), _lib.ptr(dx), B, H, W, C, _stream(dy)),
                       "zs_upsample2x_bwd_nhwc")
        return dx


def upsample2x(x):
    return _Upsample2x.apply(x)


class _ToNHWC(torch.autograd.Function):
    """NCHW [B,C,H,W] (times an optional per-pixel mask [B,1,H,W]) -> channels-last, zero padded to cpad."""

    @staticmethod
    def forward(ctx, x, mask, cpad):
        lib = _lib.load()
        x = _f32c(x, "to_nhwc input")
        B, C, H, W = x.shape
        m = None if mask is None else _f32c(mask.float(), "to_nhwc mask")
        y = torch.empty(B, H, W, cpad, dtype=torch.float32, device=x.device)
        with _lib.on(x.device):
            _lib.check(lib.zs_nchw_to_nhwc(_lib.ptr(x), _lib.ptr(m), _lib.ptr(y), B, C, H * W, cpad, _stream(x)),
                       "zs_nchw_to_nhwc")
        ctx.shape = (B, C, H, W, cpad)
        ctx.save_for_backward(m)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (m,) = ctx.saved_tensors
        B, C, H, W, cpad = ctx.shape
        dy = _f32c(dy, "to_nhwc grad")
        dx = torch.empty(B, C, H, W, dtype=torch.float32, device=dy.device)
        with _lib.on(dy.device):
            _lib.check(lib.zs_nhwc_to_nchw_masked(_lib.ptr(dy), _lib.ptr(m), _lib.ptr(dx), B, C, H * W, cpad,
                                                  _stream(dy)), "zs_nhwc_to_nchw_masked")
        return dx, None, None


def to_nhwc(x, cpad=None, mask=None):
    return _ToNHWC.apply(x, mask, cpad or x.shape[1])


class _ToNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _f32c(x, "to_nchw input")
        B, H, W, C = x.shape
        y = torch.empty(B, C, H, W, dtype=torch.float32, device=x.device)
        with _lib.on(x.device):
            _lib.check(lib.zs_nhwc_to_nchw(_lib.ptr(x), _lib.ptr(y), B, C, H * W, _stream(x)), "zs_nhwc_to_nchw")
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        dy = _f32c(dy, "to_nchw grad")
        B, C, H, W = dy.shape
        dx = torch.empty(B, H, W, C, dtype=torch.float32, device=dy.device)
        with _lib.on(dy.device):
            _lib.check(lib.zs_nchw_to_nhwc(_lib.ptr(dy), None, _lib.ptr(dx), B, C, H * W, C, _stream(dy)),
                       "zs_nchw_to_nhwc")
        return dx


def to_nchw(x):
    return _ToNCHW.apply(x)


class _SeenSurface(torch.autograd.Function):
    """graph_shape.py:131-144 in one launch each way: (depth [B,1,H,W], intr [B,3,3], mask) ->
    (seen_points [B,HW,3], seen_3D_dsp [B,3,H,W], mask_dsp [B,1,H,W]); same-size resample only."""

    @staticmethod
    def forward(ctx, depth, intr, mask):
        lib = _lib.load()
        depth, intr = _f32c(depth, "depth"), _f32c(intr, "intr")
        m = _f32c(mask.float(), "mask")
        B, _, H, W = depth.shape
        dev = depth.device
        seen = torch.empty(B, H * W, 3, dtype=torch.float32, device=dev)
        mean = torch.empty(B, 3, dtype=torch.float32, device=dev)
        scale = torch.empty(B, dtype=torch.float32, device=dev)
        coord = torch.empty(B, 3, H, W, dtype=torch.float32, device=dev)
        mask_dsp = torch.empty(B, 1, H, W, dtype=torch.float32, device=dev)
        with _lib.on(dev):
            _lib.check(lib.zs_seen_surface(_lib.ptr(depth), _lib.ptr(intr), _lib.ptr(m), B, H, W, H, W, _lib.ptr(seen),
                                           _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(coord), _lib.ptr(mask_dsp),
                                           _stream(depth)), "zs_seen_surface")
        ctx.save_for_backward(depth, intr, m, mean, scale)
        ctx.mark_non_differentiable(mask_dsp)
        return seen, coord, mask_dsp

    @staticmethod
    def backward(ctx, d_seen, d_coord, _d_mask):
        lib = _lib.load()
        depth, intr, m, mean, scale = ctx.saved_tensors
        B, _, H, W = depth.shape
        d_seen = None if d_seen is None else _f32c(d_seen, "seen grad")
        d_coord = None if d_coord is None else _f32c(d_coord, "coord grad")
        dd = torch.empty_like(depth)
        dk = torch.empty_like(intr)
        with _lib.on(depth.device):
            _lib.check(lib.zs_seen_surface_bwd(_lib.ptr(depth), _lib.ptr(intr), _lib.ptr(m), _lib.ptr(mean),
                                               _lib.ptr(scale), _lib.ptr(d_seen), _lib.ptr(d_coord), B, H, W,
                                               _lib.ptr(dd), _lib.ptr(dk), _stream(depth)), "zs_seen_surface_bwd")
        return dd, dk, None


def seen_surface(depth, intr, mask):
    return _SeenSurface.apply(depth, intr, mask)


class _IntrParam2Mtx(torch.autograd.Function):
    @staticmethod
    def forward(ctx, params, H, W):
        lib = _lib.load()
        params = _f32c(params, "intr params")
        B = params.shape[0]
        intr = torch.empty(B, 3, 3, dtype=torch.float32, device=params.device)
        with _lib.on(params.device):
            _lib.check(lib.zs_intr_param2mtx(_lib.ptr(params), B, H, W, _lib.ptr(intr), _stream(params)),
                       "zs_intr_param2mtx")
        ctx.hw = (H, W)
        ctx.save_for_backward(params)
        return intr

    @staticmethod
    def backward(ctx, d_intr):
        lib = _lib.load()
        (params,) = ctx.saved_tensors
        d_intr = _f32c(d_intr, "intr grad")
        dp = torch.empty_like(params)
        with _lib.on(params.device):
            _lib.check(lib.zs_intr_param2mtx_bwd(_lib.ptr(params), _lib.ptr(d_intr), params.shape[0], ctx.hw[0],
                                                 ctx.hw[1], _lib.ptr(dp), _stream(params)), "zs_intr_param2mtx_bwd")
        return dp, None, None


def intr_param2mtx(params, H, W):
    return _IntrParam2Mtx.apply(params, H, W)


# ---- ViT token plumbing of DPT-hybrid (model/depth/vit.py:103-148, :31-43) ----
class _ResizeGrid(torch.autograd.Function):
    """Bilinear (align_corners=False) resize of a channels-last grid [Hi,Wi,C] -> [Ho,Wo,C]."""

    @staticmethod
    def forward(ctx, x, Ho, Wo):
        lib = _lib.load()
        x = _f32c(x, "resize input")
        Hi, Wi, C = x.shape
        y = torch.empty(Ho, Wo, C, dtype=torch.float32, device=x.device)
        with _lib.on(x.device):
            _lib.check(lib.zs_resize_bilinear_nhwc(_lib.ptr(x), _lib.ptr(y), Hi, Wi, Ho, Wo, C, 0, _stream(x)),
                       "zs_resize_bilinear_nhwc")
        ctx.shape = (Hi, Wi, Ho, Wo, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        Hi, Wi, Ho, Wo, C = ctx.shape
        dy = _f32c(dy, "resize grad")
        dx = torch.empty(Hi, Wi, C, dtype=torch.float32, device=dy.device)
        with _lib.on(dy.device):
            _lib.check(lib.zs_resize_bilinear_nhwc(_lib.ptr(dy), _lib.ptr(dx), Hi, Wi, Ho, Wo, C, 1, _stream(dy)),
                       "zs_resize_bilinear_nhwc(bwd)")
        return dx, None, None


def resize_grid(x, Ho, Wo):
    return _ResizeGrid.apply(x, Ho, Wo)


class _AssembleTokens(torch.autograd.Function):
    """tokens[b] = [cls | feat[b]] + pos   (feat [B,n,C], cls [C], pos [n+1,C])."""

    @staticmethod
    def forward(ctx, feat, cls, pos):
        lib = _lib.load()
        feat, cls, pos = _f32c(feat, "tokens"), _f32c(cls, "cls"), _f32c(pos, "pos")
        B, n, C = feat.shape
        y = torch.empty(B, n + 1, C, dtype=torch.float32, device=feat.device)
        with _lib.on(feat.device):
            _lib.check(lib.zs_assemble_tokens(_lib.ptr(feat), _lib.ptr(cls), _lib.ptr(pos), _lib.ptr(y), B, n, C,
                                              _stream(feat)), "zs_assemble_tokens")
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _f32c(dy, "tokens grad")
        B, n1, C = dy.shape
        dpos = column_sum(dy.view(B, n1 * C)).view(n1, C)
        return dy[:, 1:].contiguous(), dpos[0].contiguous(), dpos


def assemble_tokens(feat, cls, pos):
    return _AssembleTokens.apply(feat, cls, pos)


class _WindowTokens(torch.autograd.Function):
    """CoordEmb's token preparation (seen_coord_enc.py:50-71; zs_window_tokens): emb [B,H,W,C], mask [B,H,W] bool,
    invalid_token [C], cls [C], pos [win*win+1, C] (fixed) -> [B*(H/win)*(W/win), win*win+1, C]."""

    @staticmethod
    def forward(ctx, emb, mask, invalid_token, cls, pos, win):
        lib = _lib.load()
        emb, invalid_token = _f32c(emb, "window_tokens input"), _f32c(invalid_token, "invalid_coord_token")
        cls, pos = _f32c(cls, "cls_token"), _f32c(pos, "two_d_pos_embed")
        B, H, W, C = emb.shape
        m = mask.to(torch.uint8).contiguous()
        out = torch.empty(B * (H // win) * (W // win), win * win + 1, C, dtype=torch.float32, device=emb.device)
        with _lib.on(emb.device):
            _lib.check(lib.zs_window_tokens(_lib.ptr(emb), _lib.ptr(m), _lib.ptr(invalid_token), _lib.ptr(cls), _lib.ptr(pos),
                                            _lib.ptr(out), B, H, W, C, win, _stream(emb)), "zs_window_tokens")
        ctx.save_for_backward(m)
        ctx.geom = (B, H, W, C, win)
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (m,) = ctx.saved_tensors
        B, H, W, C, win = ctx.geom
        dy = _f32c(dy, "window_tokens grad")
        d_emb = torch.empty(B, H, W, C, dtype=torch.float32, device=dy.device)
        d_inv = torch.empty_like(d_emb)
        d_cls = torch.empty(B * (H // win) * (W // win), C, dtype=torch.float32, device=dy.device)
        with _lib.on(dy.device):
            _lib.check(lib.zs_window_tokens_bwd(_lib.ptr(dy), _lib.ptr(m), _lib.ptr(d_emb), _lib.ptr(d_inv), _lib.ptr(d_cls),
                                                B, H, W, C, win, _stream(dy)), "zs_window_tokens_bwd")
        return d_emb, None, column_sum(d_inv.view(-1, C)), column_sum(d_cls), None, None


def window_tokens(emb, mask, invalid_token, cls, pos, win):
    return _WindowTokens.apply(emb, mask, invalid_token, cls, pos, win)


class _SeenSurfaceDsp2(torch.autograd.Function):
    """graph_shape.py:131-144 with arch.depth.dsp = 2 (the transformer coordinate encoder): as _SeenSurface, the
    coordinate map and its mask resampled to half the size (interpolate_coordmap, utils/util.py:336-345)."""

    @staticmethod
    def forward(ctx, depth, intr, mask):
        lib = _lib.load()
        depth, intr = _f32c(depth, "depth"), _f32c(intr, "intr")
        m = _f32c(mask.float(), "mask")
        B, _, H, W = depth.shape
        Ho, Wo = H // 2, W // 2
        dev = depth.device
        seen = torch.empty(B, H * W, 3, dtype=torch.float32, device=dev)
        mean = torch.empty(B, 3, dtype=torch.float32, device=dev)
        scale = torch.empty(B, dtype=torch.float32, device=dev)
        coord = torch.empty(B, 3, Ho, Wo, dtype=torch.float32, device=dev)
        mask_dsp = torch.empty(B, 1, Ho, Wo, dtype=torch.float32, device=dev)
        with _lib.on(dev):
            _lib.check(lib.zs_seen_surface(_lib.ptr(depth), _lib.ptr(intr), _lib.ptr(m), B, H, W, Ho, Wo, _lib.ptr(seen),
                                           _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(coord), _lib.ptr(mask_dsp),
                                           _stream(depth)), "zs_seen_surface")
        ctx.save_for_backward(depth, intr, m, mean, scale, mask_dsp)
        ctx.mark_non_differentiable(mask_dsp)
        return seen, coord, mask_dsp

    @staticmethod
    def backward(ctx, d_seen, d_coord, _d_mask):
        lib = _lib.load()
        depth, intr, m, mean, scale, mask_dsp = ctx.saved_tensors
        B, _, H, W = depth.shape
        d_seen = None if d_seen is None else _f32c(d_seen, "seen grad")
        d_full = None
        with _lib.on(depth.device):
            if d_coord is not None:
                d_full = torch.empty(B, 3, H, W, dtype=torch.float32, device=depth.device)
                _lib.check(lib.zs_coord_dsp2_bwd(_lib.ptr(_f32c(d_coord, "coord grad")), _lib.ptr(m), _lib.ptr(mask_dsp),
                                                 _lib.ptr(d_full), B, H, W, _stream(depth)), "zs_coord_dsp2_bwd")
            dd = torch.empty_like(depth)
            dk = torch.empty_like(intr)
            _lib.check(lib.zs_seen_surface_bwd(_lib.ptr(depth), _lib.ptr(intr), _lib.ptr(m), _lib.ptr(mean),
                                               _lib.ptr(scale), _lib.ptr(d_seen), _lib.ptr(d_full), B, H, W,
                                               _lib.ptr(dd), _lib.ptr(dk), _stream(depth)), "zs_seen_surface_bwd")
        return dd, dk, None


def seen_surface_dsp2(depth, intr, mask):
    return _SeenSurfaceDsp2.apply(depth, intr, mask)


class _ReadoutConcat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tokens):
        lib = _lib.load()
        tokens = _f32c(tokens, "readout input")
        B, n1, C = tokens.shape
        y = torch.empty(B, n1 - 1, 2 * C, dtype=torch.float32, device=tokens.device)
        with _lib.on(tokens.device):
            _lib.check(lib.zs_readout_concat(_lib.ptr(tokens), _lib.ptr(y), B, n1 - 1, C, _stream(tokens)),
                       "zs_readout_concat")
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        dy = _f32c(dy, "readout grad")
        B, n, C2 = dy.shape
        dt = torch.empty(B, n + 1, C2 // 2, dtype=torch.float32, device=dy.device)
        with _lib.on(dy.device):
            _lib.check(lib.zs_readout_concat_bwd(_lib.ptr(dy), _lib.ptr(dt), B, n, C2 // 2, _stream(dy)),
                       "zs_readout_concat_bwd")
        return dt


def readout_concat(tokens):
    return _ReadoutConcat.apply(tokens)


# ---- the depth task's losses (options/depth.yaml) ----
class _MidasLoss(torch.autograd.Function):
    """MidasLoss.forward (model/depth/midas_loss.py:166-185): ssi MAE + alpha * gradient matching."""

    @staticmethod
    def forward(ctx, prediction, target, mask, alpha, scales, inverse_depth):
        lib = _lib.load()
        p, t = _f32c(prediction, "depth prediction"), _f32c(target.float(), "depth target")
        m = _f32c(mask.float(), "depth mask")
        B, _, H, W = p.shape
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        ws = torch.empty((_ws_bytes("zs_midas_loss_workspace_bytes", B) + 3) // 4, dtype=torch.float32, device=p.device)
        with _lib.on(p.device):
            _lib.check(lib.zs_midas_loss(_lib.ptr(p), _lib.ptr(t), _lib.ptr(m), B, H, W, float(alpha), int(scales),
                                         1 if inverse_depth else 0, _lib.ptr(loss), _lib.ptr(ws), _stream(p)),
                       "zs_midas_loss")
        ctx.cfg = (alpha, scales, inverse_depth)
        ctx.save_for_backward(p, t, m, ws)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        lib = _lib.load()
        p, t, m, ws = ctx.saved_tensors
        alpha, scales, inverse_depth = ctx.cfg
        B, _, H, W = p.shape
        dloss = _f32c(dloss, "loss grad")
        dp = torch.empty_like(p)
        with _lib.on(p.device):
            _lib.check(lib.zs_midas_loss_bwd(_lib.ptr(p), _lib.ptr(t), _lib.ptr(m), B, H, W, float(alpha), int(scales),
                                             1 if inverse_depth else 0, _lib.ptr(ws), _lib.ptr(dloss), _lib.ptr(dp),
                                             _stream(p)), "zs_midas_loss_bwd")
        return dp, None, None, None, None, None


@torch.no_grad()
def erode_mask(mask, pool=4):
    """MidasLoss.erode_mask (model/depth/midas_loss.py:153-162): [B,1,H,W] -> [B,1,H,W] in {0, 1}."""
    lib = _lib.load()
    m = _f32c(mask.float(), "mask")
    B, _, H, W = m.shape
    out = torch.empty_like(m)
    with _lib.on(m.device):
        _lib.check(lib.zs_erode_mask(_lib.ptr(m), B, H, W, pool, _lib.ptr(out), _stream(m)), "zs_erode_mask")
    return out


def midas_loss(prediction, target, mask, alpha=0.1, scales=4, inverse_depth=True):
    return _MidasLoss.apply(prediction, target, mask, alpha, scales, inverse_depth)


class _IntrLoss(torch.autograd.Function):
    """Loss.intr_loss (utils/loss.py:36-43)."""

    @staticmethod
    def forward(ctx, seen_pred, seen_gt, mask):
        lib = _lib.load()
        a, b, m = _f32c(seen_pred, "seen_pred"), _f32c(seen_gt.float(), "seen_gt"), _f32c(mask.float(), "mask")
        n = m.numel()
        assert a.numel() == 3 * n and b.numel() == 3 * n
        out = torch.empty(2, dtype=torch.float32, device=a.device)
        with _lib.on(a.device):
            _lib.check(lib.zs_intr_loss(_lib.ptr(a), _lib.ptr(b), _lib.ptr(m), n, _lib.ptr(out), _stream(a)), "zs_intr_loss")
        ctx.save_for_backward(a, b, m, out)
        return out[0].clone()

    @staticmethod
    def backward(ctx, dloss):
        lib = _lib.load()
        a, b, m, out = ctx.saved_tensors
        dloss = _f32c(dloss, "loss grad")
        da = torch.empty_like(a)
        with _lib.on(a.device):
            _lib.check(lib.zs_intr_loss_bwd(_lib.ptr(a), _lib.ptr(b), _lib.ptr(m), m.numel(), _lib.ptr(out), _lib.ptr(dloss),
                                            _lib.ptr(da), _stream(a)), "zs_intr_loss_bwd")
        return da, None, None


def intr_loss(seen_pred, seen_gt, mask):
    return _IntrLoss.apply(seen_pred, seen_gt, mask)
