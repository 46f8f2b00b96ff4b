#!/usr/bin/env python3
"""End-to-end use of the MI355X hot path on synthetic inputs, the way evaluate.py drives the
reference (model/shape_engine.py:364 -> utils/eval_3D.eval_metrics), optionally sharded over
the GPUs of one node:

    python examples/eval_synthetic.py --vox-res 128 --brute-force
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \\
           examples/eval_synthetic.py --vox-res 256

    python examples/eval_synthetic.py --from-images --vox-res 128 --items 4

Single process: eval_metrics(opt, var, impl_network) exactly like the reference.
--from-images: the whole inference graph (model/shape_engine.py::Runner): analytic RGB + mask
renders -> DPT depth + intrinsics -> seen-surface geometry -> coordinate encoder -> decoder grid
-> marching cubes -> Chamfer, with seeded (untrained) weights.
Multi process: every rank runs the per-image prologue, evaluates its x-slab of the grid and
one RCCL all_gather rebuilds the level grid (zeroshape_amd/parallel.py); the pose search is
sharded by rotation range and reduced to the sequential scan's winner.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ZS_SYNTHETIC_STANDIN", "1")   # this example runs on the analytic stand-in data on purpose
import numpy as np
import torch
import torch.distributed as dist

from zeroshape_amd import parallel, synthetic as syn
from zeroshape_amd.model.shape.implicit import Implicit
from zeroshape_amd.utils import eval_3D as E
from zeroshape_amd.utils.options import EasyDict as edict
from zeroshape_amd.utils.pos_embed import get_2d_sincos_pos_embed


def from_images(args):
    """evaluate.py's flow on the analytic dataset, every stage on the GPU."""
    from zeroshape_amd.data.synthetic import Dataset
    from zeroshape_amd.model.shape_engine import Runner
    opt = edict(dict(H=224, W=224, device="cuda:0", output_path=args.output, load=None, world_size=1,
                     pretrain=dict(depth=None), data=dict(dataset_test="synthetic", num_classes_test=1),
                     training=dict(n_sdf_points=4096),
                     arch=dict(num_heads=8, latent_dim=256, win_size=16,
                               depth=dict(encoder="resnet", n_blocks=12, dsp=2, pretrained=None),
                               rgb=dict(encoder=None, n_blocks=12),
                               impl=dict(n_channels=256, att_blocks=2, mlp_ratio=4., posenc_perlayer=False,
                                         mlp_layers=8, posenc_3D=0, skip_in=[2, 4, 6])),
                     eval=dict(batch_size=1, vox_res=args.vox_res, range=[-1.5, 1.5], num_points=args.num_points,
                               icp=False, brute_force=args.brute_force,
                               f_thresholds=[0.005, 0.01, 0.02, 0.05, 0.1, 0.2])))
    torch.manual_seed(0)
    r = Runner(opt)
    r.load_dataset(opt, dataset=Dataset(opt, n_items=args.items))
    r.build_networks(opt)
    gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                        "encoder_golden.npz")
    if os.path.exists(gold):                                   # deterministic seeded weights
        g = np.load(gold)
        shapes = {str(k): tuple(int(x) for x in str(v).split(",") if x)
                  for k, v in zip(g["graph_keys"], g["graph_shapes"]) if not str(k).startswith("impl_network.")}
        sd = {k: torch.from_numpy(v) for k, v in syn.seeded_encoder_state_dict(shapes, 0).items()}
        r.graph.load_state_dict(sd, strict=False)
    r.graph.enable_hip_graph(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = r.evaluate(opt)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("image -> metrics, %d samples at vox_res %d: CD %.4f (acc %.4f, comp %.4f), F@0.05 %.4f; %.1f ms/sample "
          "incl. first-call setup%s" % (args.items, args.vox_res, out["cd"], out["dist_acc"], out["dist_cov"],
                                        out["f_scores"][3], dt * 1e3 / args.items,
                                        "; results in " + args.output if args.output else ""))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vox-res", type=int, default=64)
    ap.add_argument("--brute-force", action="store_true")
    ap.add_argument("--num-points", type=int, default=10000)
    ap.add_argument("--from-images", action="store_true")
    ap.add_argument("--items", type=int, default=4)
    ap.add_argument("--output", default=None, help="directory for the reference's result files")
    args = ap.parse_args()
    if args.from_images:
        return from_images(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local = int(os.environ.get("ZS_DEVICE_OVERRIDE", local))      # single-GPU rehearsal only
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        backend = os.environ.get("ZS_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    pe = get_2d_sincos_pos_embed(256, 14, cls_token=True).astype(np.float32)
    sd = {k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(0, pos_embed=pe).items()}
    net = Implicit(196, latent_dim=256, n_channels=256, n_blocks_attn=2, n_layers_mlp=8, num_heads=8,
                   skip_in=[2, 4, 6], pos_perlayer=False)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    opt = edict(dict(device=str(dev), H=224, W=224, arch=dict(win_size=16), data=dict(dataset_test="synthetic"),
                     eval=dict(vox_res=args.vox_res, range=[-1.5, 1.5], num_points=args.num_points, icp=False,
                               brute_force=args.brute_force, f_thresholds=[0.005, 0.01, 0.02, 0.05, 0.1, 0.2])))
    latent = torch.from_numpy(syn.seeded_latent(0, 1)).to(dev)
    gt = torch.from_numpy(syn.ellipsoid_cloud(0, args.num_points))[None].to(dev)
    var = edict(dict(idx=[0], latent_depth=latent, latent_semantic=None,
                     rgb_input_map=torch.zeros(1, 3, 224, 224, device=dev),
                     pose_gt=torch.eye(3, 4, device=dev)[None], dpc=dict(points=gt.clone())))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if world == 1:
        E.eval_metrics(opt, var, net)
    else:
        G = args.vox_res + 1
        axis = torch.linspace(-1.5, 1.5, G, device=dev)
        state = net.prepare(latent)
        occ = parallel.sharded_level_grid_points(
            lambda b, e: net.query_grid_range(latent, axis, b, e, state=state), G)
        _, cloud = E._surface_clouds(opt, occ)            # identical on every rank (same seed)
        if args.brute_force:
            sl = parallel.rotation_range(6912, world, rank)
            if sl[1] > sl[0]:
                acc, comp, f, _, _, idx, cd = E.brute_force_search(cloud[0], gt[0], opt.eval.f_thresholds, dev,
                                                                   rot_slice=sl, return_index=True)
                payload = torch.cat([acc.view(1), comp.view(1), f.view(-1)])
            else:
                idx, cd, payload = 6912, float("inf"), torch.zeros(8, device=dev)
            payload, cd, idx = parallel.reduce_best_rotation(cd, idx, payload)
            var.cd_acc, var.cd_comp, var.f_score = payload[0:1], payload[1:2], payload[None, 2:]
        else:
            pred = E.normalize_pc(cloud)
            d1, d2, _, _ = E.chamfer_distance(opt, pred, E.normalize_pc(gt))
            var.cd_acc, var.cd_comp = d1.mean(1), d2.mean(1)
            var.f_score = E.compute_fscore(d1, d2, opt.eval.f_thresholds)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        print("vox_res %d, %d GPU(s): cd_acc %.6f cd_comp %.6f f-score@0.05 %.4f  (%.1f ms incl. first-call setup)"
              % (args.vox_res, world, float(var.cd_acc[0]), float(var.cd_comp[0]), float(var.f_score[0, 3]), dt * 1e3))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
