"""Host logic: the packed decoder program (zeroshape_amd/program.py), consumed in the
kernel's schedule by a numpy MFMA emulator (tests/mfma_emulator.py), reproduces the
oracle.  This pins record order, the accumulator-layout permutation and every params
offset on the CPU, before any GPU run."""
import numpy as np
import torch

from oracle import decoder_ref as R
from zeroshape_amd import program as P
from zeroshape_amd import synthetic as syn
from tests import mfma_emulator as E


def test_layout_constants():
    assert P.G_HEAD % P.RING == 0 and P.G_MLP_TILE % P.RING == 0 and P.G_KV_HEAD % P.RING == 0
    assert P.G_TOTAL == 9856 and P.G_TOTAL * 4 == 39424
    assert P.REC_FLOATS * 4 == (9856 + 8) * 1024
    assert P.PROGRAM_BYTES == P.REC_FLOATS * 4 + P.PARAM_FLOATS * 4
    assert P.PARAMS.total * 4 < 64 * 1024          # params are staged in LDS
    assert sorted(P.ROW_TABLE.reshape(-1).tolist()) == list(range(32))


def test_rowparam_roundtrip():
    v = np.arange(256, dtype=np.float32)
    rpv = P.rowparam(v).reshape(8, 2, 16)
    for t in range(8):
        for hi in range(2):
            for r in range(16):
                assert rpv[t, hi, r] == 32 * t + P.row(r, hi)


def test_mfma_emulator_is_a_matmul():
    rs = np.random.RandomState(0)
    W = rs.randn(32, 32)            # one output tile, one input tile
    X = rs.randn(32, 32)            # [feature][point]
    recs = P._interleave(P._records_linear(W.astype(np.float32), 0, 0)).astype(np.float64)
    xt = np.stack([X[E.ROWS[r], E.COL] for r in range(16)])      # activation tile in registers
    acc = E.gemm_tile(E.Stream(recs), [xt], np.zeros((16, 64)))
    Y = W.astype(np.float32).astype(np.float64) @ X              # asymmetric: catches transposes
    np.testing.assert_allclose(acc, np.stack([Y[E.ROWS[r], E.COL] for r in range(16)]), atol=1e-12)


def test_emulated_kernel_matches_oracle(seeded_sd):
    sd_np = {k: v.numpy() for k, v in seeded_sd.items()}
    latent = torch.from_numpy(syn.seeded_latent(seed=0, batch=1))
    lp = R.latent_path(seeded_sd, latent)
    kv = {}
    for blk in range(2):
        for h in range(8):
            kv[(blk, h)] = (lp["k%d" % blk][0, h].numpy(), lp["v%d" % blk][0, h].numpy())
    recs = P.pack_records(sd_np, kv)
    params = P.pack_params(sd_np)
    assert recs.size == P.REC_FLOATS and recs.dtype == np.float32
    assert P.pack_program(sd_np, kv).size == P.PROGRAM_FLOATS
    pts = syn.seeded_cloud(11, 1, 32, -1.5, 1.5)
    want, _ = R.implicit_forward(seeded_sd, latent, torch.from_numpy(pts))
    got = E.decode_wave(recs, params, pts[0])
    np.testing.assert_allclose(got, want[0].numpy().astype(np.float64), atol=2e-5, rtol=0)


def test_split_stream_order_and_layout():
    src = P.split_source_kblocks()
    assert sorted(src) == list(range(src.size))                    # a permutation
    kb_block, kb_att = P.G_BLOCK // 2, P.HEADS * P.G_HEAD // 2
    assert np.array_equal(src[:kb_att], np.arange(kb_att))         # attention sections untouched
    m = src[kb_att:kb_att + 48] - kb_att
    assert list(m[:16]) == list(range(16))                         # fc1(0)
    assert list(m[16:32]) == list(range(32, 48))                   # fc1(1)
    assert list(m[32:48]) == list(range(16, 32))                   # fc2(0)
    # impl_mlp: L0 | L1 | per pair [x part | feat part] per output tile, plain layer; then the zero tail
    b0, kl = 2 * kb_block, 128
    assert np.array_equal(src[b0:b0 + kl], b0 + np.arange(kl)) and np.array_equal(src[b0 + kl:b0 + 2 * kl], b0 + 4 * kl + np.arange(kl))
    assert list(src[b0 + 2 * kl:b0 + 2 * kl + 34] - b0) == list(range(5 * kl, 5 * kl + 16)) + list(range(kl, kl + 16)) + [5 * kl + 16, 5 * kl + 17]
    assert np.array_equal(src[b0 + 4 * kl:b0 + 5 * kl], b0 + 6 * kl + np.arange(kl))                     # layer 3
    assert np.array_equal(src[b0 + 11 * kl:], np.arange(b0 + 11 * kl, src.size))                        # tail
    # one 32x32 unit: K-block j holds records 8j..8j+7, hi + lo ~ the weight
    rs = np.random.RandomState(1)
    prog = np.zeros(P.PROGRAM_FLOATS, np.float32)
    W = rs.randn(32, 32).astype(np.float32)
    prog[:1024] = P._interleave(P._records_linear(W, 0, 0))
    words = P.split_program(prog)
    val = words[:1024].view(np.float16).reshape(2, 2, 64, 8).astype(np.float32)
    for j in range(2):
        rec = np.stack([P._records_linear(W, 0, 0)[8 * j + e] for e in range(8)], axis=1)   # [lane][e]
        np.testing.assert_allclose(val[j, 0] + val[j, 1], rec, rtol=2 ** -20, atol=1e-7)
        np.testing.assert_allclose(val[j, 0], rec, rtol=2 ** -10, atol=0)


def test_split_mfma_emulator_is_a_matmul():
    rs = np.random.RandomState(0)
    W = rs.randn(32, 32).astype(np.float32)
    X = rs.randn(32, 32)
    prog = np.zeros(P.PROGRAM_FLOATS, np.float32)
    prog[:1024] = P._interleave(P._records_linear(W, 0, 0))
    xt = np.stack([X[E.ROWS[r], E.COL] for r in range(16)])
    acc = E.gemm_tile_split(E.SplitStream(P.split_program(prog)), [xt], np.zeros((16, 64)))
    Y = W.astype(np.float64) @ X
    np.testing.assert_allclose(acc, np.stack([Y[E.ROWS[r], E.COL] for r in range(16)]), atol=5e-5)   # split arithmetic: ~2^-20 relative; a layout error would be O(1)


def test_emulated_split_kernel_matches_oracle(seeded_sd):
    """The split-fp16 schedule (permuted MLP stream, K-block operands, hi/lo products) on the CPU:
    same program the device derives, within the split arithmetic's error of the oracle."""
    sd_np = {k: v.numpy() for k, v in seeded_sd.items()}
    latent = torch.from_numpy(syn.seeded_latent(seed=0, batch=1))
    lp = R.latent_path(seeded_sd, latent)
    kv = {(blk, h): (lp["k%d" % blk][0, h].numpy(), lp["v%d" % blk][0, h].numpy())
          for blk in range(2) for h in range(8)}
    words = P.split_program(P.pack_program(sd_np, kv))
    assert words.size == P.PROGRAM_FLOATS
    pts = syn.seeded_cloud(11, 1, 32, -1.5, 1.5)
    want, _ = R.implicit_forward(seeded_sd, latent, torch.from_numpy(pts))
    got = E.decode_wave(words, P.pack_params(sd_np), pts[0], split=True)
    err = np.abs(got - want[0].numpy().astype(np.float64))
    assert err.max() < 1e-5, err.max()
