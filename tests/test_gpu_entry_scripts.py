"""GPU: the launchers with the reference's command line (train.py / evaluate.py at the repository
root) run end to end on the analytic data set, and the depth engine evaluates."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from zeroshape_amd.data.synthetic import Dataset
from zeroshape_amd.utils import options

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--pretrain.depth=", "--arch.depth.pretrained=", "--eval.vox_res=16", "--eval.num_points=400",
          "--training.n_sdf_points=256", "--max_epoch=1", "--batch_size=4"]


def run(script, *args, items=4, standin=None):
    # ZS_SYNTHETIC_STANDIN: the data.* modules of this repository are analytic stand-ins, handed out on an explicit
    # opt-in only (zeroshape_amd/data/__init__.py)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="0", ZS_SYNTHETIC_ITEMS=str(items), ZS_SYNTHETIC_STANDIN="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, script)] + list(args), cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    if r.returncode != 0:           # keep the whole output of a failed launcher where a later reader finds it (scratch directory)
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "launcher_failure_%s.log" % script.replace(".py", "")), "w") as f:
                f.write("argv: %s\nrc: %d\n--- stdout\n%s\n--- stderr\n%s\n" % (list(args), r.returncode, r.stdout, r.stderr))
        except OSError:
            pass
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    if script in ("train.py", "evaluate.py") if standin is None else standin:     # the launchers that load data by name
        assert "SYNTHETIC STAND-IN" in r.stderr
    return r.stdout


def lines(path):
    """Result file without the stand-in tag the engines put on top (asserted to be there)."""
    rows = open(path).read().split("\n")
    assert rows[0].startswith("# SYNTHETIC STAND-IN DATA"), rows[0]
    return rows[1:]


def test_stand_in_data_needs_an_opt_in(tmp_path):
    env = {k: v for k, v in os.environ.items() if k != "ZS_SYNTHETIC_STANDIN"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "evaluate.py"), "--yaml=options/shape.yaml",
                        "--data.dataset_test=pix3d", "--output_root=%s" % tmp_path, "--pretrain.depth=",
                        "--arch.depth.pretrained="], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "analytic stand-in" in r.stderr and "ZS_SYNTHETIC_STANDIN" in r.stderr


def test_train_script_with_the_captured_amp_step(tmp_path):
    """train.py --optim.hip_graph --optim.amp: an epoch of four steps (two eager, the capture, a replay), the
    validation pass behind it and the checkpoint with the loss scaler's state."""
    out = "--output_root=%s" % tmp_path
    run("train.py", "--yaml=options/shape.yaml", out, "--optim.hip_graph", "--optim.amp", *COMMON, items=16)
    ck = torch.load(os.path.join(str(tmp_path), "shape", "shape_recon", "latest.ckpt"), map_location="cpu")
    assert ck["iter"] == 4 and "optim" in ck and ck["scaler"]["scale"] > 0 and ck["scaler"]["_growth_tracker"] <= 4
    assert all(np.isfinite(v.float().numpy()).all() for v in ck["graph"].values())


def test_train_then_evaluate_scripts(tmp_path):
    out = "--output_root=%s" % tmp_path
    run("train.py", "--yaml=options/shape.yaml", out, *COMMON)
    run_dir = os.path.join(str(tmp_path), "shape", "shape_recon")
    ck = torch.load(os.path.join(run_dir, "latest.ckpt"), map_location="cpu")
    assert ck["iter"] == 1 and "optim" in ck and len(ck["graph"]) == 813
    assert os.path.exists(os.path.join(run_dir, "options.yaml"))
    run("evaluate.py", "--yaml=options/shape.yaml", out, "--load=%s/latest.ckpt" % run_dir, "--eval.batch_size=2", *COMMON)
    q = lines(os.path.join(run_dir, "quantitative_synthetic.txt"))
    assert q[0].startswith("CD     Acc    Comp") and np.isfinite(float(q[1].split()[0]))
    assert len(open(os.path.join(run_dir, "data_list.txt")).read().strip().split("\n")) == 4
    # BASELINE config 3, literally (README.md:108): Pix3D-shaped items, vox_res 128, brute-force alignment - incl.
    # the Pix3D-only xy flip of utils/eval_3D.py:122-123 - and config 5: OmniObject3D-shaped items at vox_res 256
    base = ["--pretrain.depth=", "--arch.depth.pretrained=", "--load=%s/latest.ckpt" % run_dir, out]
    run("evaluate.py", "--yaml=options/shape.yaml", "--data.dataset_test=pix3d", "--eval.vox_res=128", "--eval.brute_force",
        "--eval.batch_size=1", *base)
    rows = [r for r in lines(os.path.join(run_dir, "pix3d_full_results.txt")) if r.strip()]
    assert len(rows) == 1 + 4 and all(np.isfinite(float(v)) for r in rows[1:] for v in r.split("\t")[1:])
    cats = open(os.path.join(run_dir, "cd_cat.txt")).read()
    assert "bed" in cats and "desk" in cats
    run("evaluate.py", "--yaml=options/shape.yaml", "--data.dataset_test=omniobj3d", "--eval.vox_res=256", "--eval.batch_size=2",
        *base)
    q = lines(os.path.join(run_dir, "quantitative_omniobj3d.txt"))
    assert np.isfinite(float(q[1].split()[0]))


def test_sharded_grid_rehearsal_on_one_gpu():
    """zeroshape_amd/parallel.py's grid sharding with 4 virtual ranks on this GPU (tools/rehearse_sharded_grid.py;
    the 8-rank vox-256 run of BASELINE config 5 is recorded in profiles/)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import rehearse_sharded_grid as rh
    rec = rh.main(["--world", "4", "--vox-res", "64", "--points", "512"])
    assert rec["gathered_equals_single_launch"] and rec["max_abs_occupancy_error_vs_oracle"] < 1e-4
    assert sum(rec["points_per_rank"]) == 65 ** 3 and max(rec["points_per_rank"]) - min(rec["points_per_rank"]) < 128 * 4
    # the multi-GPU step's batch (more images than ranks, uneven): the per-image checks are sharded (image i on rank i % 2), the
    # verdicts really exchanged between the processes, the gathered grids equal one process's unsharded launch bit for bit
    rec = rh.main(["--world", "2", "--vox-res", "32", "--points", "256", "--batch", "3"])
    assert rec["gathered_equals_single_launch"] and rec["max_abs_occupancy_error_vs_oracle"] < 1e-4
    assert rec["image_flags_identical_on_every_rank"] is True


def test_depth_engine_evaluates(tmp_path, encoder_sd):
    from zeroshape_amd.model.depth_engine import Runner
    cmd = options.parse_arguments(["--yaml=%s/options/depth.yaml" % ROOT, "--output_root=%s" % tmp_path,
                                   "--arch.depth.pretrained=", "--eval.batch_size=2"])
    opt = options.set(cmd)
    opt.world_size = 1
    r = Runner(opt)
    r.load_dataset(opt, dataset=Dataset(opt, n_items=3, load_3D=False))
    r.build_networks(opt)
    sd = {k: v for k, v in encoder_sd.items() if k.startswith(("dpt_depth.", "intr_head.", "intr_proj."))}
    r.graph.load_state_dict(sd, strict=True)
    val = r.evaluate(opt, ep=0)
    assert np.isfinite(val) and val == r.last_metrics["l1_err"]
    lines = open(os.path.join(opt.output_path, "best_val.txt")).read().strip().split("\n")
    assert [l.split(":")[0] for l in lines] == ["d>1.02", "d>1.05", "d>1.1", "d>1.2", "rmse", "l1_err", "abs_rel"]
    var = r.evaluate_batch(opt, options.EasyDict(next(iter(r.test_loader))))
    assert var.depth_pred.shape == (2, 1, 224, 224) and var.intr_pred.shape == (2, 3, 3)
    assert var.seen_points_pred.shape == var.seen_points_gt.shape == (2, 224 * 224, 3)


def test_demo_script_shape_and_depth(tmp_path, encoder_sd, seeded_sd):
    """demo.py with the reference's command line on one synthetic RGBA example (configs[0] of
    BASELINE.json, run on the GPU path: there is no CPU path to run it on)."""
    from PIL import Image
    opt = options.set(options.parse_arguments(["--yaml=%s/options/shape.yaml" % ROOT, "--output_root=%s" % tmp_path]),
                      need_gpu=False)
    item = Dataset(opt, n_items=1)[0]
    os.makedirs(tmp_path / "data" / "images")
    os.makedirs(tmp_path / "data" / "masks")
    rgb = (item["rgb_input_map"].numpy().transpose(1, 2, 0) * 255).astype(np.uint8)
    Image.fromarray(rgb).save(tmp_path / "data" / "images" / "blob.png")
    Image.fromarray((item["mask_input_map"][0].numpy() * 255).astype(np.uint8)).save(tmp_path / "data" / "masks" / "blob.png")
    full = dict(encoder_sd)
    full.update({"impl_network." + k: v for k, v in seeded_sd.items()})
    torch.save(dict(epoch=0, iter=0, best_val=1.0, best_ep=0, graph=full), tmp_path / "shape.ckpt")
    run("demo.py", "--yaml=options/shape.yaml", "--task=shape", "--datadir=%s/data" % tmp_path, "--eval.vox_res=32",
        "--ckpt=%s/shape.ckpt" % tmp_path, "--output_root=%s" % tmp_path)
    preds = tmp_path / "data" / "preds"
    assert {p.name for p in preds.iterdir()} == {"blob_image_input.png", "blob_mask_input.png", "blob_depth_est.png",
                                                  "blob_mesh.obj"}
    assert Image.open(preds / "blob_image_input.png").size == (224, 224)
    obj = open(preds / "blob_mesh.obj").read().split("\n")
    assert obj[0].startswith("# zeroshape_amd mesh")            # random weights: the surface may be empty, the file is not
    depth_sd = {k: v for k, v in full.items() if k.startswith(("dpt_depth.", "intr_head.", "intr_proj."))}
    torch.save(dict(epoch=0, iter=0, best_val=1.0, best_ep=0, graph=depth_sd), tmp_path / "depth.ckpt")
    run("demo.py", "--yaml=options/depth.yaml", "--task=depth", "--datadir=%s/data" % tmp_path,
        "--ckpt=%s/depth.ckpt" % tmp_path, "--output_root=%s" % tmp_path)
    assert (preds / "blob_depth_est.png").exists() and not (preds / "blob_mesh.obj").exists()


def test_bench_line_contract():
    """`python bench.py` (short run, no CPU legs): ONE JSON line with the driver's keys, the roofline object consistent
    with itself (achieved = points x algorithmic flop / launch time, frac = achieved / peak) and with the timed
    region, measured traffic only from a profile of the current kernel source, and the evaluation legs."""
    import json
    # (the legs with tests of their own - the 304-iteration training run of test_gpu_trained_weights.py, the CPU-oracle pipeline
    # at vox 128 of test_gpu_eval_pipeline.py, the logit-scale sweeps - are left out: the suite must fit the driver's limit)
    out = run("bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
              "--skip-legs", "trained_weights,chamfer_l1_vox128,logit_scale_sweep,vox256,encoder_att")
    lines = [ln for ln in out.strip().split("\n") if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "f16x3" and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    pts = 129 ** 3
    assert d["config"]["points_per_step"] == pts and abs(d["value"] - pts / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["peak"] == 2500.0 and r["unit"] == "TFLOP/s"
    achieved = pts * r["algorithmic_flop_per_point"] / (r["launch_ms_mean"] * 1e-3) / 1e12
    assert abs(r["achieved"] - achieved) < 1e-2 * achieved and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["launch_ms_mean"] <= d["ms_per_step"] * 1.02           # the kernel is inside the step
    assert r["traffic"] is None or (r["traffic"] > pts * 4 and "stale" not in r["traffic_source"])
    assert d["exact_f32"]["occupancy_flips"] <= 2 and d["exact_f32"]["max_abs_logit_diff"] < 1e-4
    # the exact-fp32 arithmetic under the same timed contract: its launch is inside its step, its value is its step time
    x = d["exact_f32"]
    assert x["steps"] == 3 and x["roofline"]["peak"] == 157.3 and x["roofline"]["launch_ms_mean"] <= x["ms_per_step"] * 1.02
    assert abs(x["value"] - pts / (x["ms_per_step"] * 1e-3)) < 1e-3 * x["value"] and x["ms_per_step"] > d["ms_per_step"]
    for w in (8, 4, 2):
        v = d["virtual_ranks_%d" % w]
        assert v["world"] == w and len(v["step_ms_per_rank"]) == w and sum(v["points_per_rank"]) == w * pts and 0.5 < v["bound"] < 1.1
    for leg in ("chamfer", "pose_search", "chamfer_l1", "encoder", "inference", "iso_surface", "train_step"):
        assert leg in d, leg
    assert d["iso_surface"]["vox128"]["triangles"] > 1000 and 0 < d["iso_surface"]["vox128"]["ms"] < 5
    assert 0 < d["train_step"]["ms_amp"] < d["train_step"]["ms"] <= d["train_step"]["ms_eager"] * 1.05
    wr = d["train_step"]["with_reducer_1rank"]          # the data-parallel step's rank-local work: six segment graphs, buckets between
    assert wr["graphs"] == 6 and wr["buckets_issued_after_each_replay"][-1] == wr["buckets"] and 0.8 < wr["compute_side_bound"] < 1.05
    assert d["inference"]["vox64"]["points"] == 65 ** 3 and 0 < d["inference"]["vox64"]["ms"] < d["inference"]["vox128"]["ms"]
    assert d["pose_search"]["pruned_equals_exhaustive"] is True and d["chamfer_l1"]["chamfer_l1_vs_oracle_pipeline_vox16"] < 1e-4


def test_bench_two_ranks_rehearsed_on_one_gpu():
    """`bench.py --gpus 2` end to end with both ranks on this GPU and gloo in place of RCCL (ZS_DEVICE_OVERRIDE /
    ZS_DIST_BACKEND: the rehearsal switches of bench.py): the self-launch, the sharded prepare with its verdict exchange, the
    point-range launches, the all-gather, the max-over-ranks timing and rank 0's single line."""
    import json
    env = dict(os.environ, HIP_VISIBLE_DEVICES="0", ZS_DEVICE_OVERRIDE="0", ZS_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--vox-res", "64", "--no-extras", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    pts = 2 * 65 ** 3
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch_images"] == 2
    assert d["config"]["points_per_step"] == pts and abs(d["value"] - pts / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    assert d["roofline"]["points_per_launch"] >= 65 ** 3 and d["roofline"]["launch_ms_mean"] <= d["ms_per_step"] * 1.02
    assert len(d["calibration"]["per_image_max_abs_diff"]) == 2 and d["dtype"] == "f16x3"
