"""GPU end-to-end training through the Runner (model/shape_engine.py:85-297 mirror): option file ->
Runner.setup_optimizer (the reference's four AdamW groups) -> train() on the analytic dataset ->
checkpoint with optimiser state -> resume -> evaluate with the trained weights."""
import os

import numpy as np
import pytest
import torch

from zeroshape_amd.data.synthetic import Dataset
from zeroshape_amd.utils import options

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def train_opt(tmp_path, *extra):
    cmd = options.parse_arguments(["--yaml=%s/options/shape.yaml" % ROOT, "--output_root=%s" % tmp_path, "--batch_size=4",
                                   "--max_epoch=1", "--pretrain.depth=", "--arch.depth.pretrained=", "--eval.vox_res=16",
                                   "--eval.num_points=500", "--training.n_sdf_points=512", "--optim.lr=3.e-4",
                                   "--optim.lr_ft=1.e-5"] + list(extra))
    opt = options.set(cmd)
    opt.world_size = 1
    return opt


def make_runner(opt, encoder_sd, seeded_sd, n_train=8):
    from zeroshape_amd.model.shape_engine import Runner
    r = Runner(opt)
    r.load_train_dataset(opt, dataset=Dataset(opt, split="train", n_items=n_train, n_points=1000, seed=1))
    r.load_dataset(opt, dataset=Dataset(opt, n_items=2, n_points=1000))
    r.build_networks(opt)
    full = dict(encoder_sd)
    full.update({"impl_network." + k: v for k, v in seeded_sd.items()})
    r.graph.load_state_dict(full, strict=True)
    r.setup_optimizer(opt)
    r.restore_checkpoint(opt)
    return r


def test_optimizer_groups_follow_the_reference(tmp_path, encoder_sd, seeded_sd):
    r = make_runner(train_opt(tmp_path), encoder_sd, seeded_sd)
    g = r.optim.param_groups
    assert [x["lr"] for x in g] == [1e-5, 1e-5, 3e-4, 3e-4] and [x["weight_decay"] for x in g] == [0.0, 0.05, 0.0, 0.05]
    names = {id(p): n for n, p in r.graph.named_parameters()}
    for i, grp in enumerate(g):
        for p in grp["params"]:
            n = names[id(p)]
            assert (("dpt_depth" in n or "intr_" in n) == (i < 2)) and ((p.ndim <= 1 or n.endswith(".bias")) == (i % 2 == 0)), n
    assert sum(len(x["params"]) for x in g) == sum(1 for p in r.graph.parameters() if p.requires_grad)
    r2 = make_runner(train_opt(tmp_path, "--optim.fix_dpt"), encoder_sd, seeded_sd)
    assert len(r2.optim.param_groups) == 2 and all("dpt_depth" not in names_ and "intr_" not in names_ for names_ in
                                                   [n for n, p in r2.graph.named_parameters() if p.requires_grad])


def test_train_checkpoint_resume_evaluate(tmp_path, encoder_sd, seeded_sd):
    opt = train_opt(tmp_path)
    r = make_runner(opt, encoder_sd, seeded_sd)
    before = {k: v.detach().clone() for k, v in r.graph.state_dict().items()}
    r.graph.train()
    losses = []
    for ep in range(3):                                   # 3 passes over 8 items, batch 4 -> 6 iterations
        for batch in r.train_loader:
            from zeroshape_amd.utils import util
            from zeroshape_amd.utils.options import EasyDict as edict
            var = util.move_to_device(edict(batch), opt.device)
            losses.append(float(r.train_iteration(opt, var).all))
    assert r.it == 6 and np.isfinite(losses).all()
    assert np.mean(losses[-2:]) < np.mean(losses[:2]), losses                # it learns something
    after = r.graph.state_dict()
    moved = [k for k in before if before[k].is_floating_point() and not torch.equal(before[k], after[k])]
    assert any(k.startswith("impl_network.") for k in moved) and any(k.startswith("coord_encoder.") for k in moved)
    assert any(k.startswith("dpt_depth.scratch") for k in moved) and any("running_mean" in k for k in moved)
    assert torch.equal(before["impl_network.pos_embed"], after["impl_network.pos_embed"])
    assert int(after["coord_encoder.encoder.bn1.num_batches_tracked"]) == 6
    # checkpoint with optimiser state (utils/util.py:252-277 layout) and resume
    r.save_checkpoint(opt, ep=2, it=r.it, best_val=0.5, best_ep=1, latest=True)
    ck = torch.load(os.path.join(opt.output_path, "latest.ckpt"), map_location="cpu")
    assert set(ck) == {"epoch", "iter", "best_val", "best_ep", "graph", "optim"} and ck["iter"] == 6
    assert len(ck["optim"]["param_groups"]) == 4 and float(ck["optim"]["state"][0]["step"]) == 6
    opt2 = train_opt(tmp_path, "--resume")
    r2 = make_runner(opt2, encoder_sd, seeded_sd)
    assert (r2.epoch_start, r2.iter_start, r2.best_val) == (2, 6, 0.5)
    for k, v in r2.graph.state_dict().items():
        assert torch.equal(v.cpu(), after[k].cpu()), k
    st, st2 = r.optim.state_dict()["state"], r2.optim.state_dict()["state"]
    assert all(torch.equal(st[i]["exp_avg"].cpu(), st2[i]["exp_avg"].cpu()) for i in (0, 5, 100))
    # one more identical step on both runners gives identical weights (the optimiser state is live)
    batch = next(iter(r.train_loader))
    for rr, oo in ((r, opt), (r2, opt2)):
        rr.graph.train()
        torch.manual_seed(7)
        rr.train_iteration(oo, util.move_to_device(edict({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}),
                                                   oo.device))
    for (k, a), (_, b) in zip(r.graph.state_dict().items(), r2.graph.state_dict().items()):
        assert torch.allclose(a, b, atol=0, rtol=0), k
    # evaluation sees the trained weights (the packed inference programs are refreshed)
    out = r.evaluate(opt)
    assert np.isfinite(out["cd"]) and out["cd"] > 0


def test_grad_reducer_rehearsal_on_one_gpu(tmp_path, encoder_sd, seeded_sd):
    """The multi-GPU gradient path on one GPU: an RCCL group of one rank, GradReducer(always=True) -
    HIP bucket packing (zs_copy_multi), async all_reduce on RCCL's stream under the backward pass,
    p.grad re-pointed at the buckets, fused AdamW reading them - gives the same weights as the
    plain single-GPU step."""
    import torch.distributed as dist
    from zeroshape_amd import parallel
    from zeroshape_amd.utils import util
    from zeroshape_amd.utils.options import EasyDict as edict
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        results = []
        for use_reducer in (False, True):
            opt = train_opt(tmp_path, "--optim.fix_dpt")
            r = make_runner(opt, encoder_sd, seeded_sd, n_train=4)
            if use_reducer:
                r.reducer = parallel.GradReducer(r.graph.parameters(), bucket_mb=16.0, always=True)
            r.graph.train()
            batch = next(iter(torch.utils.data.DataLoader(r.train_data, batch_size=4, shuffle=False)))
            for it in range(3):                          # iteration 0 discovers the used parameters, 1-2 overlap
                torch.manual_seed(11 + it)
                var = util.move_to_device(edict({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}),
                                          opt.device)
                r.train_iteration(opt, var)
            if use_reducer:
                assert len(r.reducer.buckets) >= 5 and all(p.grad is None for p in r.graph.parameters())
                r.reducer.close()
            results.append({k: v.detach().clone() for k, v in r.graph.state_dict().items()})
        for k in results[0]:
            assert torch.equal(results[0][k], results[1][k]), k
    finally:
        dist.destroy_process_group()


def test_captured_step_with_a_grad_reducer_on_one_gpu(tmp_path, encoder_sd, seeded_sd):
    """the captured step is no longer switched off by a GradReducer.  RCCL group of one rank,
    GradReducer(always=True): the graph holds forward + backward, reduce_in_place() packs / all-reduces / copies the buckets back
    into the graph's static gradients behind every replay - same weights as the captured step without a reducer."""
    import torch.distributed as dist
    from zeroshape_amd import parallel
    from zeroshape_amd.utils import util
    from zeroshape_amd.utils.options import EasyDict as edict
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        results = []
        for use_reducer in (False, True):
            opt = train_opt(tmp_path, "--optim.fix_dpt", "--optim.hip_graph")
            r = make_runner(opt, encoder_sd, seeded_sd, n_train=4)
            if use_reducer:
                r.reducer = parallel.GradReducer(r.graph.parameters(), bucket_mb=16.0, always=True)
            assert r._step_capture_enabled(opt)
            r.graph.train()
            batch = next(iter(torch.utils.data.DataLoader(r.train_data, batch_size=4, shuffle=False)))
            for it in range(5):                          # two eager warm-up steps, the capture, two replays
                torch.manual_seed(11 + it)
                var = util.move_to_device(edict({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}),
                                          opt.device)
                r.train_iteration(opt, var)
            assert getattr(r, "_captured", None) is not None
            if use_reducer:
                assert len(r.reducer.buckets) >= 5 and not r.reducer.armed
                r.reducer.close()
            results.append({k: v.detach().clone() for k, v in r.graph.state_dict().items()})
        for k in results[0]:
            assert torch.equal(results[0][k], results[1][k]), k
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("amp", [False, True])
def test_segmented_captured_step_overlaps_buckets_and_changes_nothing(tmp_path, encoder_sd, seeded_sd, amp):
    """with a reducer that really exchanges buckets the step is captured as one hipGraph per backward
    segment (nn/autograd.py "Segmented backward": decoder + losses | coordinate encoder | DPT decoder + heads | ViT blocks
    6-11 | blocks 3-5 | stem + blocks 0-2) and the buckets go out between the replays.  RCCL group of one rank, the WHOLE
    network trainable: six graphs, buckets issued after the first, second, ... replay - not all behind the last - and the
    weights after two eager + three replayed steps equal those of the single-graph captured step
    (optim.hip_graph_segments=false) bit for bit."""
    import torch.distributed as dist
    from zeroshape_amd import parallel
    from zeroshape_amd.utils import util
    from zeroshape_amd.utils.options import EasyDict as edict
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29537" if amp else "29536"
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        results = []
        for segments in (False, True):
            opt = train_opt(tmp_path, "--optim.hip_graph", "--optim.hip_graph_segments=%s" % ("true" if segments else "false"),
                            *(["--optim.amp"] if amp else []))      # amp: split-fp16 GEMMs + the loss scaler (unscale / skip read .grad)
            r = make_runner(opt, encoder_sd, seeded_sd, n_train=4)
            r.reducer = parallel.GradReducer(r.graph.parameters(), bucket_mb=16.0, always=True)
            assert r._step_capture_enabled(opt) and r._segmented_capture(opt) == segments
            r.graph.train()
            issued = []
            launch_done = r.reducer.launch_done
            r.reducer.launch_done = lambda params: issued.append(launch_done(params)) or issued[-1]
            batch = next(iter(torch.utils.data.DataLoader(r.train_data, batch_size=4, shuffle=False)))
            for it in range(5):                          # two eager warm-up steps, the capture, two more replays
                torch.manual_seed(11 + it)
                var = util.move_to_device(edict({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}),
                                          opt.device)
                loss = r.train_iteration(opt, var)
            assert np.isfinite(float(loss.all))
            cap = r._captured
            if segments:
                assert len(cap["graphs"]) == 6 and len(cap["seg_params"]) == 6
                trainable = [p for p in r.graph.parameters() if p.requires_grad and p.grad is not None]
                assert sum(len(g) for g in cap["seg_params"]) == len(trainable)
                names = {id(p): n for n, p in r.graph.named_parameters()}
                first = {names[id(p)].split(".")[0] for p in cap["seg_params"][0]}
                assert first == {"impl_network"}, first                      # the decoder's gradients are final first ...
                assert all(names[id(p)].startswith("dpt_depth.pretrained.model.") for p in cap["seg_params"][5])      # ... the stem's last
                per_step = issued[-6:]
                assert per_step == sorted(per_step) and per_step[-1] == len(r.reducer.buckets)
                assert per_step[1] >= 1 and per_step[3] > per_step[1], per_step        # buckets leave while backward is still replaying
            else:
                assert len(cap["graphs"]) == 1 and cap["seg_params"] is None and not issued
            r.reducer.close()
            results.append({k: v.detach().clone() for k, v in r.graph.state_dict().items()})
        for k in results[0]:
            assert torch.equal(results[0][k], results[1][k]), k
    finally:
        dist.destroy_process_group()


def test_train_loop_checkpoints_and_resume_skip(tmp_path, encoder_sd, seeded_sd):
    """Runner.train like the reference's (model/shape_engine.py:164-246, 283-284): an evaluation before
    the first step of a fresh run, latest.ckpt every freq.ckpt_latest iterations, checkpoint/ep<N>.ckpt at
    the end, and a resumed run skips the slots of its first epoch that were trained before."""
    opt = train_opt(tmp_path, "--freq.ckpt_latest=3", "--optim.fix_dpt", "--freq.eval=1000")
    opt.max_epoch = 2
    r = make_runner(opt, encoder_sd, seeded_sd)
    r.load_train_dataset(opt, dataset=Dataset(opt, split="train", n_items=16, n_points=1000, seed=1))   # 4 iterations per epoch
    saves, evals = [], []
    r.save_checkpoint = lambda opt, **kw: saves.append(kw)
    ev = r.evaluate
    r.evaluate = lambda *a, **kw: (evals.append(kw), ev(*a, **kw))[1]
    r.train(opt)
    assert r.it == 8 and evals == [dict(ep=0, training=True)]
    assert [(s["it"], s.get("latest", False)) for s in saves] == [(3, True), (6, True), (8, False)]
    assert saves[-1]["ep"] == 1
    # resume from iteration 6 of 8: epoch 1 has two slots left
    r.epoch_start, r.iter_start = 1, 6
    opt.resume = True
    saves.clear(), evals.clear()
    r.best_val = 0.25
    r.train(opt)
    assert r.it == 8 and evals == [] and r.best_val == 0.25
    assert [(s["it"], s.get("latest", False)) for s in saves] == [(6, True), (8, False)]


def test_optim_amp_runs_the_forward_convolutions_in_split_fp16(tmp_path, encoder_sd, seeded_sd):
    """--optim.amp (the reference: fp16 autocast + GradScaler, model/shape_engine.py:135-136, :252-269): the encoders'
    forward GEMMs on the 16-bit matrix pipe with split operands, everything else fp32 - same loss to 1e-4 relative,
    the step updates the weights, and a runner built without the flag is back on fp32."""
    from zeroshape_amd.nn import autograd as A
    from zeroshape_amd.utils import util
    from zeroshape_amd.utils.options import EasyDict as edict
    losses = {}
    try:
        for amp in (False, True):
            opt = train_opt(tmp_path, *(["--optim.amp"] if amp else []))
            r = make_runner(opt, encoder_sd, seeded_sd, n_train=4)
            assert A.FWD_CONV_PRECISION == A.BWD_DATA_PRECISION == ("f16x3" if amp else "f32")
            assert hasattr(r, "scaler") == amp
            r.graph.train()
            batch = next(iter(torch.utils.data.DataLoader(r.train_data, batch_size=4, shuffle=False)))
            before = r.graph.coord_encoder.encoder.conv1.weight.detach().clone()
            torch.manual_seed(3)
            losses[amp] = float(r.train_iteration(opt, util.move_to_device(edict(batch), opt.device)).all)
            assert not torch.equal(before, r.graph.coord_encoder.encoder.conv1.weight)
        assert np.isfinite(losses[True]) and abs(losses[True] - losses[False]) < 1e-4 * max(1.0, abs(losses[False])), losses
    finally:
        A.set_forward_precision("f32")
        A.set_backward_precision("f32")


def test_optim_amp_gradients_and_the_loss_scaler(tmp_path, encoder_sd, seeded_sd):
    """optim.amp data gradients on the split-fp16 convolution engine under the dynamic loss scale: every parameter's
    gradient (unscaled) within 3e-4 of the one the exact-fp32 data-gradient kernels give for the same forward pass,
    relative to the tensor's largest entry (measured 1.1e-4 at the initial scale 2^16, 4.5e-3 at 2^10: the error is
    the fp16 subnormal floor under the scaled gradients, it shrinks as the scale grows).  A clean step grows the
    tracker, an overflowing scale skips the step - weights and moments untouched - and halves."""
    from zeroshape_amd.nn import autograd as A
    from zeroshape_amd.utils import util
    from zeroshape_amd.utils.options import EasyDict as edict
    grads = {}
    try:
        for split_bwd in (False, True):
            opt = train_opt(tmp_path, "--optim.amp")
            r = make_runner(opt, encoder_sd, seeded_sd, n_train=4)
            if not split_bwd:
                A.set_backward_precision("f32")
            for m in r.graph.modules():
                if hasattr(m, "drop_path") and isinstance(m.drop_path, float):
                    m.drop_path = 0.0
            r.graph.train()
            batch = next(iter(torch.utils.data.DataLoader(r.train_data, batch_size=4, shuffle=False)))
            var, loss = r.graph.forward(opt, util.move_to_device(edict(batch), opt.device), training=True, get_loss=True)
            loss = r.summarize_loss(opt, var, loss)
            r.scaler.scale_loss(loss.all).backward()
            inv = 1.0 / float(r.scaler.scale)
            grads[split_bwd] = {n: p.grad.detach().double().cpu() * inv for n, p in r.graph.named_parameters()
                                if p.grad is not None}
        assert grads[True].keys() == grads[False].keys() and len(grads[True]) > 300
        worst = max((float((grads[True][n] - g).abs().max()) / (float(g.abs().max()) + 1e-30), n) for n, g in grads[False].items())
        assert 0 < worst[0] < 3e-4, worst
        # r is the amp runner, its gradients are in place: a clean step, then an overflowing one
        w = r.graph.coord_encoder.encoder.conv1.weight
        before = w.detach().clone()
        r.scaler.step(r.optim, None)
        assert not torch.equal(before, w) and int(r.scaler.tracker) == 1 and float(r.scaler.scale) == 65536.0
        r.optim.zero_grad()
        r.scaler.scale.fill_(2.0 ** 100)
        before = w.detach().clone()
        m_before = r.optim.state[w]["exp_avg"].clone()
        r.train_iteration(opt, util.move_to_device(edict(batch), opt.device))
        assert torch.equal(before, w) and torch.equal(m_before, r.optim.state[w]["exp_avg"])
        assert float(r.scaler.scale) == 2.0 ** 99 and int(r.scaler.tracker) == 0 and float(r.scaler.found_inf) == 1.0
    finally:
        A.set_forward_precision("f32")
        A.set_backward_precision("f32")


def test_captured_step_matches_the_eager_step(tmp_path, encoder_sd, seeded_sd):
    """--optim.hip_graph: forward + loss + backward replayed as one captured hipGraph (two eager warm-up steps, then the
    capture), the optimiser behind it.  With DropPath off (its draws differ between the two modes only by the state
    of the generator) the same batches give the same losses and the same weights as the eager step."""
    from zeroshape_amd.utils import util
    from zeroshape_amd.utils.options import EasyDict as edict
    results = {}
    for captured in (False, True):
        opt = train_opt(tmp_path, *(["--optim.hip_graph"] if captured else []))
        r = make_runner(opt, encoder_sd, seeded_sd, n_train=8)
        for m in r.graph.modules():
            if hasattr(m, "drop_path") and isinstance(m.drop_path, float):
                m.drop_path = 0.0
        r.graph.train()
        batches = list(torch.utils.data.DataLoader(r.train_data, batch_size=4, shuffle=False))
        losses = []
        for it in range(6):
            var = util.move_to_device(edict(batches[it % 2]), opt.device)
            losses.append(float(r.train_iteration(opt, var).all))
        assert (getattr(r, "_captured", None) is not None) == captured and r.it == 6
        results[captured] = (losses, {k: v.detach().clone() for k, v in r.graph.state_dict().items()})
    (l0, sd0), (l1, sd1) = results[False], results[True]
    assert np.allclose(l0, l1, rtol=1e-6, atol=0), (l0, l1)
    assert l0[4] != l0[0]                                      # the weights move
    worst = max(float((sd0[k].float() - sd1[k].float()).abs().max()) for k in sd0)
    assert worst <= 1e-6, worst


def test_captured_step_is_recaptured_when_a_workspace_moves_or_the_batch_changes(tmp_path, encoder_sd, seeded_sd):
    """A capture holds the addresses of the workspaces and the shapes of its inputs: when a workspace is reallocated
    (an evaluation between epochs can grow one) or another batch shape arrives, the runner drops it, warms up eagerly
    twice and captures again - training goes on, gradients keep flowing into the optimiser."""
    from zeroshape_amd.nn import autograd as A
    from zeroshape_amd.utils import util
    from zeroshape_amd.utils.options import EasyDict as edict
    opt = train_opt(tmp_path, "--optim.hip_graph")
    r = make_runner(opt, encoder_sd, seeded_sd, n_train=8)
    r.graph.train()
    batch = next(iter(torch.utils.data.DataLoader(r.train_data, batch_size=4, shuffle=False)))
    half = {k: (v[:2] if torch.is_tensor(v) else v[:2] if isinstance(v, list) else v) for k, v in batch.items()}

    def steps(b, n):
        return [float(r.train_iteration(opt, util.move_to_device(edict(b), opt.device)).all) for _ in range(n)]
    first = steps(batch, 4)
    cap0 = r._captured
    assert cap0 is not None
    A.SCRATCH_GENERATION[0] += 1                     # what a reallocated workspace does
    second = steps(batch, 4)
    cap1 = r._captured
    assert cap1 is not None and cap1 is not cap0
    w = r.graph.impl_network.point_proj.proj.weight.detach().clone() if hasattr(r.graph.impl_network, "point_proj") else None
    third = steps(half, 4)                           # two images per step: another input signature
    cap2 = r._captured
    assert cap2 is not None and cap2 is not cap1 and cap2["static"]["rgb_input_map"].shape[0] == 2
    assert np.isfinite(first + second + third).all() and r.it == 12
    assert second[-1] < first[0]                     # eight steps on the same batch: the loss went down
    if w is not None:
        assert not torch.equal(w, r.graph.impl_network.point_proj.proj.weight)


def test_training_steps_are_bit_reproducible(tmp_path, encoder_sd, seeded_sd):
    """Every reduction of the training kernels has a fixed order (no atomics) and DropPath draws from the seeded device
    generator: two runs of five steps (two eager, the capture, two replays) from the same state give bit-identical
    losses, parameters and BatchNorm buffers."""
    from zeroshape_amd.utils import util
    from zeroshape_amd.utils.options import EasyDict as edict
    runs = []
    for _ in range(2):
        opt = train_opt(tmp_path, "--optim.hip_graph")
        r = make_runner(opt, encoder_sd, seeded_sd, n_train=8)
        r.graph.train()
        batches = list(torch.utils.data.DataLoader(r.train_data, batch_size=4, shuffle=False))
        torch.manual_seed(7)
        losses = [r.train_iteration(opt, util.move_to_device(edict(batches[it % 2]), opt.device)).all.detach().clone()
                  for it in range(5)]
        runs.append((torch.stack(losses).cpu(), {k: v.detach().cpu().clone() for k, v in r.graph.state_dict().items()}))
    assert torch.equal(runs[0][0], runs[1][0]), (runs[0][0], runs[1][0])
    assert [k for k in runs[0][1] if not torch.equal(runs[0][1][k], runs[1][1][k])] == []
