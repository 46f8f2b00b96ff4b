"""CPU tests of round 5's host-side logic (no GPU, no library calls): the fused-GroupNorm gate over every trunk map, the heat map
of the attention frames, the logit-scale stand-in for a converged checkpoint, the per-space verdict arithmetic of
Implicit.prepare (torch CPU), and the 256 x 256 GEMM kernel's tile / tail plan restated from csrc/nn_conv.hip."""
import numpy as np
import torch

from oracle import decoder_ref as R
from zeroshape_amd import synthetic as syn


def test_fused_group_norm_gate_checks_every_trunk_map():
    from zeroshape_amd.nn.blocks import _fused_maps_ok
    assert _fused_maps_ok(1, 160, 160, 3) and _fused_maps_ok(1, 37, 41, 3)             # batch 1: any map
    assert not _fused_maps_ok(2, 160, 160, 3)        # 80^2, 40^2 fine; 20^2 = 400 pixels is no multiple of 32
    assert not _fused_maps_ok(2, 224, 224, 3)        # 28^2 = 784
    assert _fused_maps_ok(2, 128, 128, 3) and _fused_maps_ok(5, 256, 256, 3)
    assert not _fused_maps_ok(3, 64, 64, 3)          # the last map is 4 x 4


def test_jet_heat_map_and_overlay():
    from zeroshape_amd.utils.eval_3D import _jet_lut, show_att_on_image
    lut = _jet_lut()
    assert lut.shape == (256, 3) and lut.dtype == np.uint8
    assert tuple(lut[0]) == (0, 0, 128) and tuple(lut[255]) == (128, 0, 0) and lut[128, 1] == 255
    assert np.all(np.diff(lut[:96, 2].astype(int)) >= 0) and np.all(np.diff(lut[160:, 0].astype(int)) <= 0)
    img = np.random.RandomState(0).rand(8, 8, 3).astype(np.float32)
    mask = np.linspace(0, 1, 64, dtype=np.float32).reshape(8, 8)
    out = show_att_on_image(img, mask)
    assert out.shape == (8, 8, 3) and out.dtype == np.float32 and abs(out.max() - 1.0) < 1e-6 and out.min() >= 0


def test_confident_state_dict_scales_the_logits():
    pe = R.pos_embed_2d_sincos(256, 14).astype(np.float32) if hasattr(R, "pos_embed_2d_sincos") else None
    sd = {k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(0, pos_embed=pe).items()}
    latent = torch.from_numpy(syn.seeded_latent(0, 1))
    pts = torch.from_numpy(syn.seeded_cloud(4, 1, 64, -1.5, 1.5))
    base, _ = R.implicit_forward(sd, latent, pts)
    big = syn.confident_state_dict(sd, 64.0, layers=3)
    changed = [k for k in sd if not torch.equal(sd[k], big[k])]
    assert sorted(changed) == sorted("impl_mlp.layers.%d.%s" % (l, p) for l in (6, 7, 8) for p in ("weight", "bias"))
    assert max(float(big[k].abs().max()) for k in changed) < 16.0            # inside program.W_MAX
    out, _ = R.implicit_forward(big, latent, pts)
    ratio = float(out.abs().max() / base.abs().max())
    assert 20.0 < ratio < 200.0              # ~gain (softplus(beta=100) is homogeneous to ~1e-2, upstream biases are not scaled)


def _pp256_plan(M, cout_pad, K, cus=256, max_split=8, min_steps=4, parts_cap=512):
    """csrc/nn_conv.hip, the 256 x 256 kernel's launcher: (whole tiles, tail tiles, K ranges per tail tile)."""
    T = -(-M // 256) * -(-cout_pad // 256)
    steps = K // 32
    tail = T if T <= cus // 2 else (T % cus if T % cus < cus // 2 else 0)
    if cus // 2 < T < cus:
        tail = 0
    sp = 1
    if tail:
        sp = min(cus // tail, max_split, steps // min_steps)
        if tail * sp > parts_cap:
            sp = parts_cap // tail
    if sp < 2:
        tail, sp = 0, 1
    return T - tail, tail, sp


def test_pp256_tile_and_tail_plan_of_the_vit_layers():
    assert _pp256_plan(5516, 3072, 768) == (256, 8, 6)          # fc1: 264 tiles = one round + 8 tiles in 6 K ranges of 4 stages
    assert _pp256_plan(5516, 2304, 768) == (198, 0, 1)          # qkv: most of one round, whole tiles
    assert _pp256_plan(5516, 768, 3072) == (0, 66, 3)           # fc2: every tile in three ranges of 32 stages
    assert _pp256_plan(16384, 3072, 768) == (768, 0, 1)         # three whole rounds
    full, tail, sp = _pp256_plan(8204, 2048, 768)               # 33 x 8 = 264 again
    assert (full, tail) == (256, 8) and full + tail * sp <= 256 + 8 * 8


def test_deferred_bn_counters_flush_once_also_when_an_exception_leaves_the_block():
    """nn/autograd.deferred_bn_counters: batch_norm_train() only notes the `num_batches_tracked` buffers inside the block, the
    OUTERMOST block bumps them all with one multi-tensor add - also when an exception leaves the block: the running
    statistics of the layers that ran were updated in place, the counters must say so."""
    from zeroshape_amd.nn import autograd as A
    counters = [torch.zeros((), dtype=torch.int64) for _ in range(5)]
    assert A._BN_COUNTERS[0] is None
    with A.deferred_bn_counters():
        A._BN_COUNTERS[0].extend(counters[:2])              # what batch_norm_train() does inside a block
        with A.deferred_bn_counters():                      # e.g. the coordinate encoder inside Graph.forward
            A._BN_COUNTERS[0].extend(counters[2:])
        assert all(int(c) == 0 for c in counters)           # the inner block did not flush
    assert all(int(c) == 1 for c in counters) and A._BN_COUNTERS[0] is None
    try:
        with A.deferred_bn_counters():
            A._BN_COUNTERS[0].append(counters[0])
            raise RuntimeError("forward failed")
    except RuntimeError:
        pass
    assert int(counters[0]) == 2 and int(counters[1]) == 1 and A._BN_COUNTERS[0] is None


def test_inline_split_mode_follows_the_precisions_and_a_change_re_packs_everything():
    """The re-pack writes the operands' fp16 halves itself only under optim.amp's settings, skips the fp32 operands only when
    forward AND data gradients read the halves, and any precision change moves the generation (stale fp32 operands must not
    be served after switching back)."""
    from zeroshape_amd.nn import autograd as A
    try:
        A.set_forward_precision("f32")
        A.set_backward_precision("f32")
        assert A._inline_split_mode() == 0
        g0 = A.GENERATION[0]
        A.set_forward_precision("f16x3")
        assert A.GENERATION[0] == g0 + 1 and A._inline_split_mode() == 1
        A.set_forward_precision("f16x3")                    # no change: no re-pack
        assert A.GENERATION[0] == g0 + 1
        A.set_backward_precision("f16x3")
        assert A.GENERATION[0] == g0 + 2 and A._inline_split_mode() == 2
        A.set_forward_precision("f32")
        assert A._inline_split_mode() == 1                  # the data gradients still read halves, the forward reads fp32
    finally:
        A.set_forward_precision("f32")
        A.set_backward_precision("f32")


def test_operand_form_switches_are_part_of_the_stamp():
    """flipping PRESPLIT_ALL / INLINE_SPLIT at run time (tools, A/B tests) must make every packed operand stale -
    in inline-split mode 2 the fp32 form has not been rewritten since the mode was entered."""
    from zeroshape_amd.nn import autograd as A
    w = torch.zeros(4, 4)
    keep = (A.PRESPLIT_ALL, A.INLINE_SPLIT)
    try:
        s0 = A._stamp(w)
        A.PRESPLIT_ALL = not keep[0]
        s1 = A._stamp(w)
        A.INLINE_SPLIT = not keep[1]
        s2 = A._stamp(w)
        assert len({s0, s1, s2}) == 3
    finally:
        A.PRESPLIT_ALL, A.INLINE_SPLIT = keep
    assert A._stamp(w) == s0
