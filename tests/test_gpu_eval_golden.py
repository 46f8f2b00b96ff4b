"""The evaluation glue against the reference ITSELF: `eval_metrics_default`,
`eval_metrics_BF` and `brute_force_search` of utils/eval_3D.py:104-213 were run end to end on the CPU by
tests/golden/make_eval_golden.py - ground truth into the view frame with pose_gt, the pix3d sign flip, normalize_pc, ICP, the
6,912-rotation search in batches of 24 with the strict-< winner, compute_fscore - with the Chamfer plugin stood in for by the
fp32 C restatement of its kernel and `convert_to_explicit` by seeded clouds.  Here the product's `eval_metrics` runs the same
samples on the GPU (its surface sampler replaced by the same seeded clouds): metrics within 2e-6, clouds within 2e-6."""
import numpy as np
import pytest
import torch

from zeroshape_amd import synthetic as syn

pytestmark = pytest.mark.gpu
THRESHOLDS = [0.005, 0.01, 0.02, 0.05, 0.1, 0.2]


@pytest.fixture(scope="module")
def net(seeded_sd):
    from zeroshape_amd.model.shape.implicit import Implicit
    m = Implicit(syn.NUM_PATCHES, latent_dim=syn.LATENT_DIM, semantic=False, n_channels=syn.N_CHANNELS,
                 n_blocks_attn=syn.ATT_BLOCKS, n_layers_mlp=syn.MLP_LAYERS, num_heads=syn.NUM_HEADS,
                 posenc_3D=0, mlp_ratio=syn.MLP_RATIO, skip_in=list(syn.SKIP_IN), pos_perlayer=False)
    m.load_state_dict(seeded_sd, strict=True)
    return m.cuda().eval()


@pytest.mark.parametrize("tag,brute_force,dataset,icp", [("default_synthetic", False, "synthetic", False),
                                                         ("default_pix3d_icp", False, "pix3d", True),
                                                         ("bf_pix3d", True, "pix3d", False)])
def test_eval_metrics_vs_the_reference(net, eval_golden, monkeypatch, tag, brute_force, dataset, icp):
    from zeroshape_amd.utils import eval_3D as E
    from zeroshape_amd.utils.options import EasyDict as edict
    g = eval_golden
    pred = torch.from_numpy(g["pred"]).cuda()
    # the sampled surface cloud is the one thing that cannot be reproduced (trimesh's random stream): hand over the golden's
    monkeypatch.setattr(E, "_surface_clouds", lambda opt, level_vox, seed=0: ([None] * level_vox.shape[0], pred.clone()))
    opt = edict(dict(device="cuda", H=224, W=224, arch=dict(win_size=16), data=dict(dataset_test=dataset),
                     eval=dict(vox_res=16, range=[-1.5, 1.5], num_points=pred.shape[1], icp=icp, brute_force=brute_force,
                               f_thresholds=THRESHOLDS)))
    var = edict(dict(idx=[0, 1], latent_depth=torch.from_numpy(syn.seeded_latent(seed=0, batch=2)).cuda(), latent_semantic=None,
                     rgb_input_map=torch.zeros(2, 3, 224, 224).cuda(), pose_gt=torch.from_numpy(g["pose"]).cuda(),
                     dpc=edict(dict(points=torch.from_numpy(g["gt"]).cuda()))))
    ret = E.eval_metrics(opt, var, net)
    np.testing.assert_allclose([float(ret[0]), float(ret[1])], g[tag + ".ret"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(var.cd_acc.cpu().numpy(), g[tag + ".cd_acc"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(var.cd_comp.cpu().numpy(), g[tag + ".cd_comp"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(var.f_score.cpu().numpy(), g[tag + ".f_score"], atol=2.1e-3 if icp else 1e-6, rtol=0)
    tol = 2e-5 if icp else 2e-6                       # (50 ICP iterations: 1e-5)
    np.testing.assert_allclose(var.dpc_pred.cpu().numpy(), g[tag + ".dpc_pred"], atol=tol, rtol=0)
    np.testing.assert_allclose(var.dpc.points.cpu().numpy(), g[tag + ".dpc_gt"], atol=2e-6, rtol=0)
    np.testing.assert_array_equal(var.eval_vox[0, [0, 1, 17, 4912]].cpu().numpy(), g[tag + ".eval_vox_corner"])


def test_brute_force_search_vs_the_reference(eval_golden):
    from zeroshape_amd.utils import eval_3D as E
    g = eval_golden
    for kw in (dict(), dict(prune=False)):
        acc, comp, fs, best_pred, gt_n = E.brute_force_search(torch.from_numpy(g["pred"][0]).cuda(), torch.from_numpy(g["gt"][0]).cuda(),
                                                              THRESHOLDS, "cuda", **kw)
        assert abs(float(acc) - float(g["search.acc"])) < 2e-6 and abs(float(comp) - float(g["search.comp"])) < 2e-6
        np.testing.assert_allclose(fs.cpu().numpy(), g["search.f_score"], atol=1e-6, rtol=0)
        np.testing.assert_allclose(best_pred.cpu().numpy(), g["search.best_pred"], atol=2e-6, rtol=0)      # the SAME rotation won
        np.testing.assert_allclose(gt_n.cpu().numpy().reshape(g["search.gt_normalized"].shape), g["search.gt_normalized"], atol=1e-6, rtol=0)
