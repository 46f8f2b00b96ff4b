"""Per-image sharding inside the engine path (BASELINE config 5)
rehearsed with W virtual ranks on ONE GPU (processes, gloo rendezvous, all on device 0): eval_metrics with
opt.eval.shard_image - point ranges of the grid + one all_gather, interleaved shares of the bound-sorted rotation
sphere, the running best shared after the first batch, device-side record reduce - must return, on every rank, the
single-rank result bit for bit: occupancy-derived clouds, Chamfer accuracy / completeness, F-scores, the aligned
prediction.  What this does not exercise is RCCL itself (gloo carries the collectives here)."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _opt(vox, brute):
    from zeroshape_amd.utils.options import EasyDict as edict
    return edict(dict(device="cuda:0", H=224, W=224, arch=dict(win_size=16), data=dict(dataset_test="synthetic"),
                      eval=dict(vox_res=vox, range=[-1.5, 1.5], num_points=3000, icp=False, brute_force=brute,
                                shard_image=True, f_thresholds=[0.005, 0.01, 0.02, 0.05, 0.1, 0.2])))


def _worker(rank, world, initfile, vox, brute, out_dir):
    from zeroshape_amd import synthetic as syn
    from zeroshape_amd.model.shape.implicit import Implicit
    from zeroshape_amd.utils import eval_3D as E
    from zeroshape_amd.utils.options import EasyDict as edict
    from zeroshape_amd.utils.pos_embed import get_2d_sincos_pos_embed
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        pe = get_2d_sincos_pos_embed(256, 14, cls_token=True).astype(np.float32)
        sd = {k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(0, pos_embed=pe).items()}
        net = Implicit(syn.NUM_PATCHES, latent_dim=256, n_channels=256, n_blocks_attn=2, n_layers_mlp=8, num_heads=8,
                       skip_in=[2, 4, 6], pos_perlayer=False)
        net.load_state_dict(sd, strict=True)
        net = net.cuda().eval()
        latent = torch.from_numpy(syn.seeded_latent(0, 2)).cuda()
        # a level set that crosses the grid: the seeded network is negative everywhere, so the (linear) last layer is
        # re-centred on the median logit of a coarse grid and given a gain (same arithmetic on every rank)
        coarse = net.query_grid(latent[:1], torch.linspace(-1.5, 1.5, 17).cuda(), apply_sigmoid=False,
                                state=net.prepare(latent[:1], "f32"))
        with torch.no_grad():
            w, b = net.impl_mlp.layers[8].weight, net.impl_mlp.layers[8].bias
            b.copy_(-40.0 * (coarse.median() - b))
            w.mul_(40.0)
        gt = torch.from_numpy(np.stack([syn.ellipsoid_cloud(s, 3000) for s in (0, 1)])).cuda()

        def run(shard):
            opt = _opt(vox, brute)
            opt.eval.shard_image = shard
            var = edict(dict(idx=[0, 1], latent_depth=latent, latent_semantic=None,
                             rgb_input_map=torch.zeros(2, 3, 224, 224).cuda(),
                             pose_gt=torch.eye(3, 4)[None].repeat(2, 1, 1).cuda(), dpc=dict(points=gt.clone())))
            E.eval_metrics(opt, var, net)
            return var
        single = run(False)                     # every rank computes the unsharded result for itself
        assert E.image_sharding(_opt(vox, brute)) == (rank, world)
        sharded = run(True)
        assert float(single.cd_acc.min()) > 0 and int((single.dpc_pred.abs().sum(-1) > 0).sum()) > 1000, "degenerate surface"
        for name in ("cd_acc", "cd_comp", "f_score", "dpc_pred"):
            a, b = getattr(single, name), getattr(sharded, name)
            assert torch.equal(a, b), "rank %d: %s differs between the sharded and the single-rank evaluation" % (rank, name)
        assert torch.equal(single.dpc.points, sharded.dpc.points)
        np.save(os.path.join(out_dir, "rank%d.npy" % rank),
                torch.cat([sharded.cd_acc.view(-1), sharded.cd_comp.view(-1), sharded.f_score.view(-1)]).cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,vox,brute", [(2, 32, False), (3, 32, True), (8, 24, True)])
def test_sharded_eval_metrics_equal_single_rank(world, vox, brute):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, os.path.join(d, "init"), vox, brute, d), nprocs=world, join=True)
        rows = [np.load(os.path.join(d, "rank%d.npy" % r)) for r in range(world)]
        for r in rows[1:]:
            assert np.array_equal(rows[0], r), "ranks disagree on the gathered metrics"
