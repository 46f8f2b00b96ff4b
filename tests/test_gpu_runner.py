"""GPU end-to-end: RGB + mask -> Graph.forward (HIP encoders) -> eval_metrics (HIP decoder,
marching cubes, Chamfer) through the Runner, on the analytic dataset."""
import os

import numpy as np
import pytest
import torch

from tests.test_encoder_contract import make_opt
from zeroshape_amd.data.synthetic import Dataset
from zeroshape_amd.utils.options import EasyDict as edict

pytestmark = pytest.mark.gpu


def runner_opt(tmp_path, brute_force=False):
    opt = make_opt()
    opt.update(device="cuda:0", output_path=str(tmp_path), load=None, world_size=1,
               data=dict(dataset_test="synthetic", num_classes_test=1),
               training=dict(n_sdf_points=1024),
               eval=dict(batch_size=2, vox_res=32, range=[-1.5, 1.5], num_points=2000, icp=False,
                         brute_force=brute_force, f_thresholds=[0.005, 0.01, 0.02, 0.05, 0.1, 0.2]))
    return opt


def test_runner_evaluate_end_to_end(tmp_path, encoder_sd, seeded_sd):
    from zeroshape_amd.model.shape_engine import Runner
    opt = runner_opt(tmp_path)
    r = Runner(opt)
    r.load_dataset(opt, dataset=Dataset(opt, n_items=3, n_points=4000))     # 2 + 1: ragged last batch
    r.build_networks(opt)
    full = dict(encoder_sd)
    full.update({"impl_network." + k: v for k, v in seeded_sd.items()})
    r.graph.load_state_dict(full, strict=True)
    out = r.evaluate(opt)
    assert set(out) == {"cd", "dist_acc", "dist_cov", "f_scores"} and len(out["f_scores"]) == 6
    assert np.isfinite([out["cd"], out["dist_acc"], out["dist_cov"]]).all() and out["cd"] > 0
    lines = open(os.path.join(str(tmp_path), "synthetic_full_results.txt")).read().split("\n")
    assert lines[0].startswith("# SYNTHETIC STAND-IN DATA")      # stand-in data says so on top of every result file
    lines = lines[1:]
    assert lines[0].startswith("IND, CD, ACC, COMP, F-score@0.50") and len(lines) == 4
    assert [int(l.split("\t")[0]) for l in lines[1:]] == [0, 1, 2]
    per = np.array([[float(x) for x in l.split("\t")[1:4]] for l in lines[1:]])
    assert abs(per[:, 1].mean() - out["dist_acc"]) < 1e-3 and abs(per[:, 2].mean() - out["dist_cov"]) < 1e-3
    q = open(os.path.join(str(tmp_path), "quantitative_synthetic.txt")).read().split("\n")[1:]
    assert q[0].startswith("CD     Acc    Comp") and q[2].startswith("F-score @ 0.50:")
    assert "ellipsoid" in open(os.path.join(str(tmp_path), "cd_cat.txt")).read()
    # same numbers from the hand-written loop evaluate.py's users would write
    from zeroshape_amd.utils import eval_3D, util
    accs = []
    for batch in r.test_loader:
        var = r.evaluate_batch(opt, edict(batch))
        assert var.latent_depth.shape[1:] == (197, 256) and var.depth_pred.shape[1:] == (1, 224, 224)
        eval_3D.eval_metrics(opt, var, r.graph.impl_network)
        accs.append(var.cd_acc)
    assert abs(torch.cat(accs).mean().item() - out["dist_acc"]) < 1e-6


def test_runner_with_hip_graph_and_brute_force(tmp_path, encoder_sd, seeded_sd):
    from zeroshape_amd.model.shape_engine import Runner
    opt = runner_opt(tmp_path, brute_force=True)
    opt.eval.batch_size = 1
    r = Runner(opt)
    r.load_dataset(opt, dataset=Dataset(opt, n_items=2, n_points=2000))
    r.build_networks(opt)
    full = dict(encoder_sd)
    full.update({"impl_network." + k: v for k, v in seeded_sd.items()})
    r.graph.load_state_dict(full, strict=True)
    eager = r.evaluate(opt)
    r.graph.enable_hip_graph(True)
    replay = r.evaluate(opt)
    assert eager == replay                               # one captured hipGraph, identical metrics
    assert len(r.graph._captured) == 1
