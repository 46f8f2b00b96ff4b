#!/usr/bin/env python3
"""BASELINE config 5 rehearsed on ONE GPU: W virtual ranks (processes, gloo rendezvous, all on the
same device) shard the (vox_res+1)^3 grid exactly as the 8-GPU run does (zeroshape_amd/parallel.py:
sharded_level_grid_points -> point_bounds / gather_points), each rank's launch timed ALONE with HIP
events (the ranks take turns, so the times are per-rank kernel times, not contention), the gathered
grid compared with the single-launch grid bit for bit and with the CPU oracle on random points.

    python tools/rehearse_sharded_grid.py [--world 8] [--vox-res 256] [--points 4096] [--precision f16x3]

Prints one JSON line.  What this does NOT measure: the RCCL all_gather over xGMI (the gather here goes
through host memory) - that leg has not run on hardware."""
import argparse
import json
import os
import sys
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def worker(rank, world, initfile, a, out_path):
    from oracle import decoder_ref
    from zeroshape_amd import parallel, synthetic as syn
    from zeroshape_amd.model.shape.implicit import Implicit
    from zeroshape_amd.utils.pos_embed import get_2d_sincos_pos_embed
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", int(os.environ.get("ZS_DEVICE_OVERRIDE", "0")))
        torch.cuda.set_device(dev)
        G = a.vox_res + 1
        P = G ** 3
        pe = get_2d_sincos_pos_embed(256, 14, cls_token=True).astype(np.float32)
        sd = {k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(0, pos_embed=pe).items()}
        net = Implicit(syn.NUM_PATCHES, latent_dim=256, n_channels=256, n_blocks_attn=2, n_layers_mlp=8,
                       num_heads=8, skip_in=[2, 4, 6], pos_perlayer=False)
        net.load_state_dict(sd, strict=True)
        net = net.to(dev).eval()
        net.precision = a.precision
        latent_c = torch.from_numpy(syn.seeded_latent(2, a.batch))
        latent = latent_c.to(dev)
        axis = torch.linspace(-1.5, 1.5, G, device=dev)
        # the multi-GPU step's prepare: every prologue on every rank, image i's output check on rank i % world, the verdicts
        # all-gathered (here: gloo on CUDA tensors)
        st = parallel.prepare_sharded(net, latent)
        torch.cuda.synchronize()
        b, e, per = parallel.point_bounds(P, world, rank)
        net.query_grid_range(latent, axis, b, min(e, b + 4096), state=st)      # warm-up (workspace, code object)
        torch.cuda.synchronize()
        ms = torch.zeros(world, dtype=torch.float64)
        local = None
        for turn in range(world):                                              # one rank on the GPU at a time
            if turn == rank:
                best = None
                for _ in range(a.repeats):      # the first launch after a turn change also pays the clock ramp
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev0.record()
                    local = net.query_grid_range(latent, axis, b, e, apply_sigmoid=True, state=st)
                    ev1.record()
                    torch.cuda.synchronize()
                    t = ev0.elapsed_time(ev1)
                    best = t if best is None else min(best, t)
                ms[rank] = best
            dist.barrier()
        dist.all_reduce(ms)
        full = parallel.sharded_level_grid_points(lambda bb, ee: local.cpu(), G)     # the product's partition + gather
        assert full.shape == (a.batch, G, G, G)
        flags_everywhere = None
        if st.image_flags is not None:
            fl = torch.stack([st.image_flags, st.image_flags_occ], -1).cpu()
            every = [torch.zeros_like(fl) for _ in range(world)]
            dist.all_gather(every, fl)
            flags_everywhere = bool(all(torch.equal(every[0], t) for t in every)) and tuple(fl.shape) == (a.batch, 2)
        if rank == 0:
            single = net.query_grid_range(latent, axis, 0, P, apply_sigmoid=True, state=net.prepare(latent)).cpu().view(a.batch, G, G, G)
            rs = np.random.RandomState(3)
            idx = rs.randint(0, G, size=(a.points, 3))
            ax = axis.cpu()
            pts = torch.stack([ax[idx[:, 0]], ax[idx[:, 1]], ax[idx[:, 2]]], -1)[None]
            want, _ = decoder_ref.implicit_forward(sd, latent_c, pts.expand(a.batch, -1, -1))
            got = full[:, idx[:, 0], idx[:, 1], idx[:, 2]]
            err = float((got - torch.sigmoid(want)).abs().max())
            times = ms.tolist()
            mean = sum(times) / len(times)
            rec = {"world": world, "vox_res": a.vox_res, "points": P, "precision": a.precision, "batch": a.batch,
                   "image_flags_identical_on_every_rank": flags_everywhere,
                   "points_per_rank": [min(P, (r + 1) * per) - min(P, r * per) for r in range(world)],
                   "launch_ms_per_rank": [round(t, 3) for t in times], "launches_per_rank": a.repeats,
                   "imbalance_max_over_mean": round(max(times) / mean - 1.0, 5),
                   "slowest_rank_points_per_s": round(per / (max(times) * 1e-3), 1),
                   "aggregate_points_per_s_if_concurrent": round(P / (max(times) * 1e-3), 1),
                   "gathered_equals_single_launch": bool(torch.equal(full, single)),
                   "max_abs_occupancy_error_vs_oracle": err, "oracle_points": a.points,
                   "transport": "gloo through host memory (rehearsal); RCCL/xGMI leg unmeasured"}
            with open(out_path, "w") as f:
                json.dump(rec, f)
    finally:
        dist.destroy_process_group()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--vox-res", type=int, default=256)
    ap.add_argument("--points", type=int, default=4096)
    ap.add_argument("--precision", default="f16x3")
    ap.add_argument("--batch", type=int, default=1, help="images per step (bench.py --gpus N: N)")
    ap.add_argument("--repeats", type=int, default=4, help="launches per rank; the fastest is reported")
    a = ap.parse_args(argv)
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "out.json")
        mp.spawn(worker, args=(a.world, os.path.join(d, "init"), a, out), nprocs=a.world, join=True)
        rec = json.load(open(out))
    print(json.dumps(rec), flush=True)
    return rec


if __name__ == "__main__":
    main()
