"""CPU checks: the analytic dataset honours the reference's sample-dict contract and is
geometrically self-consistent; checkpoints round-trip through the reference's layout."""
import os

import numpy as np
import pytest
import torch

from tests.test_encoder_contract import make_opt
from zeroshape_amd.data.synthetic import Dataset
from zeroshape_amd.utils.options import EasyDict as edict


def test_sample_dict_contract():
    opt = edict(dict(H=224, W=224, training=dict(n_sdf_points=1024)))
    ds = Dataset(opt, n_items=4)
    assert len(ds) == 4
    s = ds[2]
    want = dict(pose_gt=(3, 4), intr=(3, 3), rgb_input_map=(3, 224, 224), mask_input_map=(1, 224, 224),
                depth_input_map=(1, 224, 224), gt_sample_points=(1024, 3), gt_sample_sdf=(1024,))
    for k, shape in want.items():
        assert tuple(s[k].shape) == shape and s[k].dtype == torch.float32, k
    assert s["idx"] == 2 and s["category_label"] == 0 and tuple(s["dpc"]["points"].shape) == (16384, 3)
    m = s["mask_input_map"]
    assert set(m.unique().tolist()) == {0.0, 1.0} and 0.02 < m.mean() < 0.6
    assert bool((s["depth_input_map"][m == 0] == 0).all()) and float(s["depth_input_map"][m > 0].min()) > 1.0
    assert bool((s["rgb_input_map"][:, m[0] == 0] == 1).all())             # white background
    assert torch.equal(ds[2]["rgb_input_map"], s["rgb_input_map"])          # deterministic per index
    assert not torch.equal(ds[1]["rgb_input_map"], s["rgb_input_map"])
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=2)))       # default collate works
    assert batch["dpc"]["points"].shape == (2, 16384, 3) and batch["idx"].tolist() == [0, 1]
    assert Dataset(opt, n_items=1, load_3D=False)[0].keys() == {"idx", "category_label", "pose_gt", "intr",
                                                                "rgb_input_map", "mask_input_map",
                                                                "depth_input_map"}


def test_depth_unprojects_onto_the_gt_surface():
    """K^-1 [u,v,1] * depth, moved to the object frame by pose_gt^-1, lies on the GT cloud; the
    SDF sign agrees with the implicit ellipsoid."""
    from scipy.spatial import cKDTree
    opt = edict(dict(H=224, W=224, training=dict(n_sdf_points=2048)))
    s = Dataset(opt, n_items=2)[1]
    K, pose = s["intr"].numpy(), s["pose_gt"].numpy()
    v, u = np.nonzero(s["mask_input_map"][0].numpy())
    z = s["depth_input_map"][0].numpy()[v, u]
    pc = np.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z], 1)
    pw = (pc - pose[:, 3]) @ pose[:, :3]
    tree = cKDTree(s["dpc"]["points"].numpy())
    assert tree.query(pw)[0].max() < 0.03
    q, sdf = s["gt_sample_points"].numpy(), s["gt_sample_sdf"].numpy()
    d = tree.query(q)[0]
    near = np.abs(sdf) < 0.02
    assert near.any() and d[near].max() < 0.05                               # small |sdf| => close to the surface
    assert 0.01 < (sdf < 0).mean() < 0.5


@pytest.fixture
def big_tmp_path(tmp_path):
    """~2 GB of checkpoints: in memory where /dev/shm exists (disk writes took minutes on slow hosts)."""
    import shutil
    import tempfile
    from pathlib import Path
    if os.path.isdir("/dev/shm"):
        d = tempfile.mkdtemp(dir="/dev/shm", prefix="zs_ckpt_")
        try:
            yield Path(d)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    else:
        yield tmp_path


def test_checkpoint_roundtrip_in_reference_layout(big_tmp_path):
    tmp_path = big_tmp_path
    from zeroshape_amd.model.compute_graph.graph_shape import Graph
    from zeroshape_amd.utils import util

    class Model:                                   # the Runner surface load/save use
        pass
    torch.manual_seed(0)
    a, b = Model(), Model()
    a.graph, b.graph = Graph(make_opt()), Graph(make_opt())
    with torch.no_grad():
        a.graph.intr_proj.weight.normal_()
    opt = edict(dict(output_path=str(tmp_path), device="cpu"))
    util.save_checkpoint(opt, a, ep=3, it=7, best_val=0.5, best_ep=2, latest=False)
    ckpt = torch.load(str(tmp_path / "latest.ckpt"), map_location="cpu")
    assert set(ckpt.keys()) == {"epoch", "iter", "best_val", "best_ep", "graph"} and ckpt["epoch"] == 3
    assert (tmp_path / "checkpoint" / "ep3.ckpt").exists()
    assert list(ckpt["graph"].keys()) == list(a.graph.state_dict().keys())
    assert not torch.equal(a.graph.intr_proj.weight, b.graph.intr_proj.weight)
    util.restore_checkpoint(opt, b, load_name=str(tmp_path / "latest.ckpt"))
    for (k, va), (_, vb) in zip(a.graph.state_dict().items(), b.graph.state_dict().items()):
        assert torch.equal(va, vb), k
    # a depth-only checkpoint (children=...) restores just those children (utils/util.py:228-239)
    util.save_checkpoint(opt, a, 0, 0, 0, 0, latest=True, children=("dpt_depth", "intr_head", "intr_proj"))
    part = torch.load(str(tmp_path / "latest.ckpt"), map_location="cpu")["graph"]
    assert all(k.split(".")[0] in ("dpt_depth", "intr_head", "intr_proj") for k in part)
    c = Model()
    c.graph = Graph(make_opt())
    before = c.graph.coord_encoder.encoder.conv1.weight.clone()
    util.load_checkpoint(opt, c, str(tmp_path / "latest.ckpt"))
    assert torch.equal(c.graph.intr_proj.weight, a.graph.intr_proj.weight)
    assert torch.equal(c.graph.coord_encoder.encoder.conv1.weight, before)


def test_pix3d_and_omniobj3d_shaped_items():
    """Sample-dict keys of the reference's data/pix3d.py:86-113 and data/omniobj3d.py:129-165, and the
    Pix3D camera convention: the GT cloud reaches the view frame only through the xy flip of
    utils/eval_3D.py:122-123 (oracle/geometry check on the CPU; the HIP path runs it in
    tests/test_gpu_entry_scripts.py)."""
    import importlib
    opt = edict(dict(H=224, W=224, data=dict(pix3d=dict(cat=None), bgcolor=1), training=dict(n_sdf_points=256)))
    pix = importlib.import_module("zeroshape_amd.data.pix3d").Dataset(opt, split="test", n_items=11, n_points=500)
    omn = importlib.import_module("zeroshape_amd.data.omniobj3d").Dataset(opt, split="test", n_items=4, n_points=500)
    p, o = pix[10], omn[3]
    assert set(p) == {"idx", "rgb_input_map", "mask_input_map", "category_label", "pose_gt", "intr", "dpc"}
    assert set(o) == {"idx", "category_label", "pose_gt", "intr", "rgb_input_map", "mask_input_map",
                      "depth_input_map", "dpc"}
    assert set(importlib.import_module("zeroshape_amd.data.omniobj3d").Dataset(opt, load_3D=False, n_items=1)[0]) == \
        set(o) - {"dpc"}
    assert len(pix.label2cat) == 9 and p["category_label"] == 10 % 9 and pix.cat2label["chair"] == 2
    assert tuple(p["pose_gt"].shape) == (3, 4) and tuple(p["dpc"]["points"].shape) == (500, 3)
    assert torch.equal(o["mask_input_map"], (o["depth_input_map"] != 0).float())
    opt2 = edict(dict(H=224, W=224, data=dict(pix3d=dict(cat="chair,sofa")), training=dict(n_sdf_points=256)))
    assert importlib.import_module("zeroshape_amd.data.pix3d").Dataset(opt2, n_items=2).label2cat == ["chair", "sofa"]
    # geometry: depth unprojected into the camera frame == flip(R_pix3d @ gt) + t, but not without the flip
    from scipy.spatial import cKDTree
    K, pose = p["intr"].numpy(), p["pose_gt"].numpy()
    v, u = np.nonzero(p["mask_input_map"][0].numpy())
    base = importlib.import_module("zeroshape_amd.data.synthetic").Dataset(opt, split="test", n_items=11, n_points=500, seed=7)[10]
    z = base["depth_input_map"][0].numpy()[v, u]
    cam = np.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z], 1)
    gt = p["dpc"]["points"].numpy().astype(np.float64)
    rot = gt @ pose[:, :3].T
    flipped = rot * np.array([-1.0, -1.0, 1.0]) + base["pose_gt"].numpy()[:, 3]
    d_flip = cKDTree(flipped).query(cam[::50])[0]
    assert np.median(d_flip) < 0.05                                    # the visible surface lies on the flipped cloud
    view = gt @ base["pose_gt"].numpy()[:, :3].T.astype(np.float64) + base["pose_gt"].numpy()[:, 3]
    np.testing.assert_allclose(flipped, view, atol=1e-5)               # = the render's own view-frame cloud
    assert np.abs(rot + base["pose_gt"].numpy()[:, 3] - view).max() > 0.1      # ... and only with the flip
    pix.id_filename_mapping(opt, "/tmp/zs_pix3d_map.txt")
    assert len(open("/tmp/zs_pix3d_map.txt").read().strip().split("\n")[0].split(" ")) == 4


def test_stand_in_datasets_need_an_opt_in(monkeypatch, capsys):
    """the data.* modules are analytic stand-ins under the reference's loader names - the engines' way in
    (data.load_by_name) refuses them without ZS_SYNTHETIC_STANDIN / opt.data.synthetic_standin, warns with it, and the
    Dataset carries the line that tags every result file."""
    import pytest
    from zeroshape_amd import data
    opt = edict(dict(H=224, W=224, data=dict(pix3d=dict(cat=None), bgcolor=1), training=dict(n_sdf_points=256)))
    monkeypatch.delenv("ZS_SYNTHETIC_STANDIN", raising=False)
    for name in ("synthetic", "pix3d", "omniobj3d"):
        with pytest.raises(RuntimeError, match="analytic stand-in"):
            data.load_by_name(opt, name, split="test")
    with pytest.raises(ModuleNotFoundError):
        data.load_by_name(opt, "ocrtoc", split="test")
    monkeypatch.setenv("ZS_SYNTHETIC_STANDIN", "1")
    d = data.load_by_name(opt, "pix3d", split="test")
    assert "SYNTHETIC STAND-IN" in capsys.readouterr().err
    assert d.synthetic_standin.startswith("# SYNTHETIC STAND-IN DATA") and "pix3d" in d.synthetic_standin
    monkeypatch.delenv("ZS_SYNTHETIC_STANDIN")
    opt.data.synthetic_standin = True
    assert "omniobj3d" in data.load_by_name(opt, "omniobj3d", split="test").synthetic_standin
