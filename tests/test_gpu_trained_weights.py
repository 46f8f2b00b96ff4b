"""The split-fp16 decoder on TRAINED weights: the reference's recipe
(options/shape.yaml, AdamW, DropPath, BatchNorm on batch statistics) is trained for >= 300
iterations on the analytic data set (examples/train_synthetic.py does the same), then the full
129^3 grid of a test image is evaluated by the f16x3 kernel, the exact-fp32 kernel and - on a
sample - the fp32 CPU oracle.  Bars: logits within 1e-4 (asserted tighter), occupancy flips only
inside the rounding band, Chamfer-L1 of the extracted surfaces within 1e-4."""
import os

import numpy as np
import pytest
import torch

from oracle import decoder_ref as R
from zeroshape_amd.data.synthetic import Dataset
from zeroshape_amd.utils import options, util
from zeroshape_amd.utils.options import EasyDict as edict

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ITERATIONS = 304


@pytest.fixture(scope="module")
def trained(tmp_path_factory):
    out = tmp_path_factory.mktemp("trained")
    cmd = options.parse_arguments(["--yaml=%s/options/shape.yaml" % ROOT, "--output_root=%s" % out, "--batch_size=4",
                                   "--max_epoch=38", "--pretrain.depth=", "--arch.depth.pretrained=",
                                   "--eval.vox_res=32", "--eval.num_points=2000", "--eval.batch_size=4",
                                   "--training.n_sdf_points=2048", "--optim.lr=3.e-4", "--optim.lr_ft=1.e-4",
                                   "--freq.eval=1000"])
    opt = options.set(cmd)
    opt.world_size = 1
    from zeroshape_amd.model.shape_engine import Runner
    torch.manual_seed(0)
    r = Runner(opt)
    r.load_dataset(opt, dataset=Dataset(opt, split="train", n_items=4, n_points=4000),
                   train_dataset=Dataset(opt, split="train", n_items=32, n_points=4000))
    r.build_networks(opt)
    r.setup_optimizer(opt)
    r.restore_checkpoint(opt)
    r.graph.train()
    first, last = [], []
    for ep in range(38):
        for batch in r.train_loader:
            var = util.move_to_device(edict(batch), opt.device)
            loss = r.train_iteration(opt, var).all.detach()
            (first if ep == 0 else last if ep == 37 else []).append(loss)
    assert r.it >= ITERATIONS >= 300
    l0, l1 = float(torch.stack(first).mean()), float(torch.stack(last).mean())
    assert l1 < 0.75 * l0, "training did not reduce the shape loss (%.3f -> %.3f)" % (l0, l1)
    r.graph.eval()
    batch = next(iter(r.test_loader))
    var = util.move_to_device(edict(batch), opt.device)
    with torch.no_grad():
        var = r.graph.forward(opt, var, training=False, get_loss=False)
    return opt, r.graph.impl_network, var.latent_depth[:1].detach().clone()


def test_trained_weights_full_grid_f16x3_vs_f32_vs_oracle(trained):
    opt, net, latent = trained
    assert net.precision == "f16x3"
    N = 128
    axis = torch.linspace(-1.5, 1.5, N + 1, device="cuda")
    st = net.prepare(latent, calibrate=False)
    assert st.precision == "f16x3", "trained weights left the host envelope (W_MAX)"
    split = net.query_grid(latent, axis, apply_sigmoid=False, state=st)
    flagged = int(net.last_tile_flags.sum())
    exact = net.query_grid(latent, axis, apply_sigmoid=False, state=net.prepare(latent, "f32"))
    err = (split - exact).abs()
    flips = (split > 0) != (exact > 0)
    print("trained weights: max |f16x3 - f32| = %.2e, mean %.2e, flips %d of %d, tiles re-evaluated %d of %d, "
          "occupied %.3f" % (float(err.max()), float(err.mean()), int(flips.sum()), split.numel(), flagged,
                             net.last_tile_flags.numel(), float((exact > 0).float().mean())))
    assert float(err.max()) < 5e-5                      # contract 1e-4
    assert bool(torch.all(exact[flips].abs() < 5e-5))   # a flip only where the level set passes within the error
    assert int(flips.sum()) <= 8
    assert 0.0 < float((exact > 0).float().mean()) < 0.5, "the trained network should predict a bounded shape"
    # the raw split arithmetic (guard off) is what the bound is about: measure it too
    net.envelope_guard = False
    try:
        raw = net.query_grid(latent, axis, apply_sigmoid=False, state=st)
    finally:
        net.envelope_guard = True
    assert float((raw - exact).abs().max()) < 1e-4
    # fp32 CPU oracle on 4096 random grid points
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    rs = np.random.RandomState(11)
    idx = rs.randint(0, N + 1, size=(4096, 3))
    ax = axis.cpu()
    pts = torch.stack([ax[idx[:, 0]], ax[idx[:, 1]], ax[idx[:, 2]]], -1)[None]
    want, _ = R.implicit_forward(sd, latent.cpu(), pts)
    scale = max(1.0, float(want.abs().max()))
    for name, grid in (("f16x3", split), ("f32", exact)):
        got = grid[0, idx[:, 0], idx[:, 1], idx[:, 2]].cpu()
        e = float((got - want[0]).abs().max())
        print("trained weights: max |%s kernel - oracle| = %.2e (logit scale %.1f)" % (name, e, scale))
        assert e < 1e-4 * scale


def test_calibration_selects_fp32_exactly_when_the_contract_would_break(trained):
    """Implicit.prepare's output-error calibration.  The last layer of the trained
    network is scaled until the logit scale is that of a converged checkpoint (30 / 60 / 100) and far beyond (1,000); the
    raw split error grows with it.  The verdict is taken in the space a call RETURNS:
      * raw logits (query_points, apply_sigmoid=False): the probe measurement must predict the full-grid error, the split
        arithmetic may only serve them while the full 129^3 grid is inside the 1e-4 contract, and what the default path
        returns is always inside it;
      * occupancies (compute_level_grid's sigmoid, utils/eval_3D.py:44-45): at 30 / 60 / 100 the state STAYS f16x3, the
        occupancy grid is within 1e-4 (asserted 2.5e-5) of the exact kernel's with the same occ > 0.5 index set outside the
        |logit| < 1e-5 band, and the per-image check hands no image to the fp32 kernel."""
    opt, net, latent = trained
    N = 128
    axis = torch.linspace(-1.5, 1.5, N + 1, device="cuda")
    w, b = net.impl_mlp.layers[8].weight, net.impl_mlp.layers[8].bias
    w0, b0 = w.detach().clone(), b.detach().clone()
    base = float(net.query_grid(latent, axis, apply_sigmoid=False, state=net.prepare(latent, "f32")).abs().max())
    seen = set()
    try:
        for target in (None, 30.0, 60.0, 100.0, 1000.0):
            gain = 1.0 if target is None else target / base
            with torch.no_grad():
                w.copy_(w0 * gain)
                b.copy_(b0 * gain)
            st = net.prepare(latent)                                     # calibrates: new weight version
            cal = dict(net.last_calibration)
            st32 = net.prepare(latent, "f32")
            exact = net.query_grid(latent, axis, apply_sigmoid=False, state=st32)
            raw_state = net.prepare(latent, calibrate=False)
            net.envelope_guard = False
            try:
                raw = net.query_grid(latent, axis, apply_sigmoid=False, state=raw_state)
            finally:
                net.envelope_guard = True
            full = float((raw - exact).abs().max())
            got = net.query_grid(latent, axis, apply_sigmoid=False, state=st)
            scale = float(exact.abs().max())
            raw_runs = "f16x3" if st.precision == "f16x3" and st.logit_ok else "f32"
            occ_runs = "f16x3" if st.precision == "f16x3" and st.occ_ok else "f32"
            print("calibration at logit scale %.1f: probe max |f16x3 - f32| = %.2e (mean %.2e), full grid %.2e, probe occupancy %.2e "
                  "-> raw logits %s, occupancies %s" % (scale, cal["max_abs_diff"], cal["mean_abs_diff"], full,
                                                         cal["max_abs_occ_diff"], cal["selected"], cal["selected_occ"]))
            assert target is None or scale >= 0.99 * target
            assert cal["selected"] == raw_runs == ("f16x3" if cal["max_abs_diff"] <= net.CALIBRATION_TOL else "f32")
            assert cal["selected_occ"] == occ_runs
            assert cal["points"] == 4096 and cal["tol"] == 2.5e-5 and cal["tol_occ"] == 2.5e-5
            # 4096 probes see the bulk of the error distribution: the full grid's maximum stays within 4x of it
            assert full <= 4.0 * max(cal["max_abs_diff"], 1e-7)
            if full > 1e-4:
                assert raw_runs == "f32", "the contract is violated on the grid and the split kernel served raw logits"
            if raw_runs == "f16x3":
                assert full <= 1e-4
            else:
                assert torch.equal(got, exact)
            assert float((got - exact).abs().max()) <= 1e-4           # what the default path returns
            # the occupancy grid: what compute_level_grid hands to marching cubes
            occ = net.query_grid(latent, axis, apply_sigmoid=True, state=st)
            sent = int(net.last_tile_flags.sum()) if occ_runs == "f16x3" else None
            occ32 = net.query_grid(latent, axis, apply_sigmoid=True, state=st32)
            d_occ = float((occ - occ32).abs().max())
            flips = ((occ > 0.5) != (occ32 > 0.5)) & (exact.abs() >= net.FLIP_BAND)
            print("    occupancy grid: max |d| %.2e, flips outside the band %d, tiles sent to the fp32 kernel %s" % (d_occ, int(flips.sum()), sent))
            assert d_occ <= 1e-4
            if occ_runs == "f16x3":
                assert d_occ <= 2.5e-5 * 4 and int(flips.sum()) == 0
            else:
                assert torch.equal(occ, occ32)
            if target in (None, 30.0, 60.0, 100.0):
                assert occ_runs == "f16x3", "a confident checkpoint's grids must keep the split arithmetic"
                assert sent <= net.last_tile_flags.numel() // 100          # (the kernel's own envelope guard may send a few tiles)
                assert d_occ <= 2.5e-5
            # cached per weight version: a second prepare() does not measure again
            marker = net._calibration
            net.prepare(latent)
            assert net._calibration is marker
            seen.add(raw_runs)
    finally:
        with torch.no_grad():
            w.copy_(w0)
            b.copy_(b0)
    assert seen == {"f16x3", "f32"}, "the sweep must cross the switch-over (saw %s)" % sorted(seen)


def test_trained_weights_chamfer_l1_delta(trained):
    """Surface extraction + sampling + Chamfer-L1 of the two arithmetics' 129^3 grids."""
    from zeroshape_amd.utils import eval_3D as E
    opt, net, latent = trained
    o = edict(dict(device="cuda", H=224, W=224, arch=dict(win_size=16), data=dict(dataset_test="synthetic"),
                   eval=dict(vox_res=128, range=[-1.5, 1.5], num_points=10000, icp=False, brute_force=False,
                             f_thresholds=[0.005, 0.01, 0.02, 0.05, 0.1, 0.2])))
    rs = np.random.RandomState(3)
    gt = torch.from_numpy(rs.uniform(-0.6, 0.6, size=(1, 10000, 3)).astype(np.float32))
    res = {}
    for prec in ("f32", "f16x3"):
        net.precision = prec
        var = edict(dict(idx=[0], latent_depth=latent, latent_semantic=None,
                         rgb_input_map=torch.zeros(1, 3, 224, 224).cuda(),
                         pose_gt=torch.eye(3, 4)[None].cuda(), dpc=dict(points=gt.clone().cuda())))
        E.eval_metrics(o, var, net)
        res[prec] = var
    net.precision = "f16x3"
    a, b = res["f32"], res["f16x3"]
    d_acc = float((a.cd_acc - b.cd_acc).abs().max())
    d_comp = float((a.cd_comp - b.cd_comp).abs().max())
    print("trained weights: Chamfer-L1 delta f16x3 vs f32: acc %.2e comp %.2e (cd %.4f)" %
          (d_acc, d_comp, float((a.cd_acc + a.cd_comp) / 2)))
    assert d_acc < 1e-4 and d_comp < 1e-4
    assert float((a.f_score - b.f_score).abs().max()) < 2e-3
