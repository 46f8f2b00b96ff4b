"""Index parity at the BASELINE grid sizes against the reference ITSELF.

tests/golden/grid_golden.npz holds what the reference's own `get_dense_3D_grid` + `compute_level_grid`
(utils/eval_3D.py:11-45) returned over the WHOLE 65^3 and 129^3 grids - and, in grid256_golden.npz, the 257^3 grid of
BASELINE config 5 - for the seeded network / image (made by tests/golden/make_grid_golden.py): the packed `occ > 0.5` bits of every point, the raw logit of every point with
|logit| < 1e-3, strided logits / occupancies and per-slice logit sums.

BASELINE.json north_star: "bit-exact on voxel indices and within 1e-4 on SDF floats".  Two fp32 evaluation orders of the
same network cannot agree on the sign of a logit that is smaller than their rounding difference, so the index clause is
asserted like this, for BOTH arithmetics of the HIP path (exact fp32 MFMA and split-fp16):
  * ZERO flips at points whose reference |logit| >= BAND (1e-5);
  * flips inside the band are counted and bounded by MAX_FLIPS[N] (and can never exceed the number of band points);
  * every logit of the near-surface set (|logit| < 1e-3: the only candidates) within NEAR_ATOL (2e-5) of the reference's;
  * strided logits / occupancies within ATOL, per-slice sums within G^2 * 2e-6 (a systematic offset would show there).
The grid goes through the product's own `utils.eval_3D.get_dense_3D_grid` + `compute_level_grid` (fused launch)."""
import numpy as np
import pytest
import torch

from zeroshape_amd import synthetic as syn

pytestmark = pytest.mark.gpu

BAND = 1e-5                       # |reference logit| below which a flip is a rounding tie
NEAR_ATOL = 2e-5                  # near-surface logits (the contract is 1e-4)
ATOL = {"f32": 2e-5, "f16x3": 2e-5}
MAX_FLIPS = {64: 3, 128: 12, 256: 60}     # stated bound on in-band flips (there are 11 / 75 / 647 band points; measured: 0-1 / 1-2 / 9-13)
STRIDE = {64: 4, 128: 8, 256: 16}


def _net(seeded_sd, precision):
    from zeroshape_amd.model.shape.implicit import Implicit
    m = Implicit(syn.NUM_PATCHES, latent_dim=syn.LATENT_DIM, semantic=False, n_channels=syn.N_CHANNELS,
                 n_blocks_attn=syn.ATT_BLOCKS, n_layers_mlp=syn.MLP_LAYERS, num_heads=syn.NUM_HEADS,
                 posenc_3D=0, mlp_ratio=syn.MLP_RATIO, skip_in=list(syn.SKIP_IN), pos_perlayer=False)
    m.load_state_dict(seeded_sd, strict=True)
    m = m.cuda().eval()
    m.precision = precision
    return m


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("N", [64, 128, 256])
def test_full_grid_indices_vs_the_reference(seeded_sd, grid_golden, grid256_golden, N, precision):
    from zeroshape_amd.utils import eval_3D as E
    if N == 256:                       # BASELINE config 5 (17 M points): its own fixture file
        grid_golden = grid256_golden
    from zeroshape_amd.utils.options import EasyDict as edict
    net = _net(seeded_sd, precision)
    G = N + 1
    latent = torch.from_numpy(syn.seeded_latent(seed=0, batch=2))[:1].cuda()
    opt = edict(dict(device="cuda", H=224, W=224, eval=dict(vox_res=N, range=[-1.5, 1.5]), arch=dict(win_size=16)))
    grid = E.get_dense_3D_grid(opt, edict(dict(idx=[0])))
    occ, _ = E.compute_level_grid(opt, net, latent, None, grid, None)
    assert occ.shape == (1, G, G, G)
    if precision == "f16x3":
        assert net.last_calibration["selected_occ"] == "f16x3"          # the split kernel really produced this grid
    bits = np.packbits((occ[0] > 0.5).reshape(-1).cpu().numpy())
    want_bits = grid_golden["occ%d_bits" % N]
    assert bits.shape == want_bits.shape
    flipped = np.nonzero(np.unpackbits(bits ^ want_bits)[:G ** 3])[0]
    near_idx, near_logit = grid_golden["near%d_idx" % N], grid_golden["near%d_logit" % N]
    band = near_idx[np.abs(near_logit) < BAND]
    outside = np.setdiff1d(flipped, band)
    assert outside.size == 0, "occupancy index flips OUTSIDE |logit| < %g at flat indices %s" % (BAND, outside[:8])
    assert flipped.size <= min(MAX_FLIPS[N], band.size), (flipped.size, band.size)
    # raw logits: the near-surface set, the strided sample, the per-slice checksums
    axis = torch.linspace(-1.5, 1.5, G, device="cuda")
    lg = net.query_grid(latent, axis, apply_sigmoid=False)[0]
    got_near = lg.reshape(-1)[torch.from_numpy(near_idx.astype(np.int64)).cuda()].cpu().numpy()
    np.testing.assert_allclose(got_near, near_logit, atol=NEAR_ATOL, rtol=0)
    s = STRIDE[N]
    np.testing.assert_allclose(lg[::s, ::s, ::s].cpu().numpy(), grid_golden["logit%d_s%d" % (N, s)], atol=ATOL[precision], rtol=0)
    np.testing.assert_allclose(occ[0, ::s, ::s, ::s].cpu().numpy(), grid_golden["occ%d_s%d" % (N, s)], atol=ATOL[precision], rtol=0)
    sums = lg.reshape(G, -1).double().sum(1).cpu().numpy()
    np.testing.assert_allclose(sums, grid_golden["logit%d_slice_sum" % N], atol=G * G * 2e-6, rtol=0)
    assert abs(float(lg.abs().max()) - float(grid_golden["logit%d_absmax" % N][0])) < ATOL[precision]
    print("vox %d %s: %d flips (all inside |logit| < %g; %d band points), near-surface max |dlogit| %.2e over %d points"
          % (N, precision, flipped.size, BAND, band.size, np.abs(got_near - near_logit).max(), near_idx.size))


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_full_grid_indices_of_a_confident_network(decoder_golden, grid_gain60_golden, precision):
    """The same check at a converged checkpoint's logit scale and on another image: the seeded network with its last three MLP
    layers scaled by 60^(1/3) each (|logit| up to 36; the reference's full 129^3 grid in grid128_gain60_golden.npz).  Two fp32
    evaluation orders of THIS network differ by ~1e-4 in the raw logit near the surface (the seeded network's 2e-6 times the
    gain), so the rounding band is 60 x wider in logit space - the contract is stated on occupancies, where it is met with a
    factor 3 to spare: |d occ| < 3.5e-5 (asserted 1e-4), no index flip outside |logit| < 6e-4, flips inside counted."""
    from zeroshape_amd.model.shape.implicit import Implicit
    from zeroshape_amd.utils import eval_3D as E
    from zeroshape_amd.utils.options import EasyDict as edict
    g = grid_gain60_golden
    gain, latent_seed = float(g["gain_and_latent_seed"][0]), int(g["gain_and_latent_seed"][1])
    assert gain == 60 and latent_seed == 1
    sd = syn.confident_state_dict(syn.seeded_state_dict(seed=0, pos_embed=decoder_golden["pos_embed_f32"]), gain)
    net = Implicit(syn.NUM_PATCHES, latent_dim=syn.LATENT_DIM, semantic=False, n_channels=syn.N_CHANNELS,
                   n_blocks_attn=syn.ATT_BLOCKS, n_layers_mlp=syn.MLP_LAYERS, num_heads=syn.NUM_HEADS,
                   posenc_3D=0, mlp_ratio=syn.MLP_RATIO, skip_in=list(syn.SKIP_IN), pos_perlayer=False)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net = net.cuda().eval()
    net.precision = precision
    N, G = 128, 129
    latent = torch.from_numpy(syn.seeded_latent(seed=latent_seed, batch=2))[:1].cuda()
    opt = edict(dict(device="cuda", H=224, W=224, eval=dict(vox_res=N, range=[-1.5, 1.5]), arch=dict(win_size=16)))
    occ, _ = E.compute_level_grid(opt, net, latent, None, E.get_dense_3D_grid(opt, edict(dict(idx=[0]))), None)
    if precision == "f16x3":
        cal = net.last_calibration
        assert cal["selected_occ"] == "f16x3", cal               # the occupancy rule keeps the split kernel at this scale
    band_logit = 1e-5 * gain                                     # 6e-4
    bits = np.packbits((occ[0] > 0.5).reshape(-1).cpu().numpy())
    flipped = np.nonzero(np.unpackbits(bits ^ g["occ128_bits"])[:G ** 3])[0]
    near_idx, near_logit = g["near128_idx"], g["near128_logit"]
    band = near_idx[np.abs(near_logit) < band_logit]
    assert np.setdiff1d(flipped, band).size == 0 and flipped.size <= band.size
    want_occ = 1.0 / (1.0 + np.exp(-near_logit.astype(np.float64)))
    got_occ = occ[0].reshape(-1)[torch.from_numpy(near_idx.astype(np.int64)).cuda()].cpu().numpy()
    np.testing.assert_allclose(got_occ, want_occ, atol=1e-4, rtol=0)
    np.testing.assert_allclose(occ[0, ::8, ::8, ::8].cpu().numpy(), g["occ128_s8"], atol=1e-4, rtol=0)
    assert abs(float(g["logit128_absmax"][0])) > 30
    print("gain 60, %s: %d flips of %d points inside |logit| < %g; near-surface max |d occ| %.2e; strided max |d occ| %.2e"
          % (precision, flipped.size, band.size, band_logit, np.abs(got_occ - want_occ).max(),
             np.abs(occ[0, ::8, ::8, ::8].cpu().numpy() - g["occ128_s8"]).max()))
