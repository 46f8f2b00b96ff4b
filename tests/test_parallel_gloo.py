"""Multi-process (world_size 2 and 3, gloo, CPU tensors) tests of the sharding logic in
zeroshape_amd/parallel.py: uneven x-slab / balanced point-range partition + padded all_gather ==
unsharded grid;
rotation-range sharding + lexicographic reduce == sequential first-strict-minimum scan."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from zeroshape_amd import parallel


def _worker(rank, world, initfile, G, B):
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        full = torch.arange(B * G * G * G, dtype=torch.float32).reshape(B, G, G, G)

        def query_slab(b, e):          # stand-in for the HIP grid kernel on this rank's slab
            return full[:, b:e].clone()

        got = parallel.sharded_level_grid(query_slab, G)
        assert got.shape == (B, G, G, G)
        assert torch.equal(got, full), "rank %d: gathered grid differs" % rank

        def query_range(b, e):         # ... on this rank's point range (memory order)
            return full.reshape(B, -1)[:, b:e].clone()

        got = parallel.sharded_level_grid_points(query_range, G)
        assert got.shape == (B, G, G, G)
        assert torch.equal(got, full), "rank %d: gathered grid (point ranges) differs" % rank

        # rotation sharding: every rank scans its range, winner = first strict minimum overall
        n_rot = 6912
        rs = np.random.RandomState(1)
        cds = rs.rand(n_rot)
        cds[[100, 5000]] = -1.0        # tie across ranks: index 100 must win
        b, e = parallel.rotation_range(n_rot, world, rank)
        assert (e - b) % 24 == 0 or e == n_rot
        loc = cds[b:e]
        if len(loc):
            j = int(np.argmin(loc))
            local_cd, local_idx = float(loc[j]), b + j
        else:
            local_cd, local_idx = float("inf"), n_rot
        payload = torch.tensor([local_cd * 2, local_idx * 3.0], dtype=torch.float32)
        pl, cd, idx = parallel.reduce_best_rotation(local_cd, local_idx, payload)
        assert idx == 100 and cd == -1.0
        assert pl[0].item() == -2.0 and pl[1].item() == 300.0
        # the ranges tile [0, n_rot) exactly
        cover = torch.zeros(n_rot)
        cover[b:e] = 1
        dist.all_reduce(cover)
        assert torch.all(cover == 1)
        # sharded per-image check of Implicit.prepare (parallel.prepare_sharded): image i's verdict pair comes from rank i % W
        for batch in (1, world, world + 1, 2 * world + 1):
            k = (batch + world - 1) // world
            mine = list(range(rank, batch, world))
            own = torch.zeros(k, 2, dtype=torch.int32)
            for slot, i in enumerate(mine):
                own[slot, 0], own[slot, 1] = 10 * i + 1, 10 * i + 2          # (what only image i's owner knows)
            got_flags = parallel.exchange_image_flags(own, batch)
            want_flags = torch.tensor([[10 * i + 1, 10 * i + 2] for i in range(batch)], dtype=torch.int32)
            assert got_flags.dtype == torch.int32 and torch.equal(got_flags, want_flags), (rank, batch, got_flags)
            # the single-process stand-in of the collective places this rank's slots where the collective would
            solo = parallel.exchange_image_flags(own, batch, gather=parallel.solo_gather(rank, world))
            for i in range(batch):
                assert torch.equal(solo[i], want_flags[i] if i % world == rank else torch.zeros(2, dtype=torch.int32))
        # the sharded pose search refuses ranks that hold different clouds
        from zeroshape_amd.utils.eval_3D import _check_identical_inputs
        pred, gt, order = torch.randn(50, 3), torch.randn(1, 40, 3), torch.arange(24, dtype=torch.int32)
        _check_identical_inputs(pred, gt, order)                       # same seed on every rank: passes
        bad = pred.clone()
        if rank == world - 1:
            bad[7, 1] += 1e-3
        try:
            _check_identical_inputs(bad, gt, order)
            raised = False
        except RuntimeError:
            raised = True
        assert raised, "rank %d: differing clouds were accepted" % rank
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,G,B", [(2, 9, 2), (2, 33, 1), (3, 5, 1), (3, 2, 2)])
def test_sharded_grid_and_rotation_reduce(world, G, B):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, os.path.join(d, "init"), G, B), nprocs=world, join=True)


def test_slab_bounds_cover_and_match_reference_layout():
    for G in (1, 2, 33, 129, 257):
        for W in (1, 2, 4, 8):
            seen = []
            for r in range(W):
                b, e, per = parallel.slab_bounds(G, W, r)
                assert 0 <= b <= e <= G and e - b <= per
                seen += list(range(b, e))
            assert seen == list(range(G))


def test_point_bounds_cover_tile_aligned_and_balanced():
    for G in (1, 2, 33, 129, 257):
        P = G ** 3
        for W in (1, 2, 3, 4, 8):
            seen, sizes = 0, []
            for r in range(W):
                b, e, per = parallel.point_bounds(P, W, r)
                assert b == min(P, seen) and b <= e <= P and e - b <= per and per % 128 == 0
                seen = e
                sizes.append(e - b)
            assert seen == P
            assert max(sizes) - P / W < 128 + 1e-9       # within one kernel tile of the ideal share
    # the case of the headline benchmark: 129^3 over 8 ranks, against whole x-slices
    sizes = [parallel.point_bounds(129 ** 3, 8, r)[1] - parallel.point_bounds(129 ** 3, 8, r)[0] for r in range(8)]
    slabs = [(parallel.slab_bounds(129, 8, r)[1] - parallel.slab_bounds(129, 8, r)[0]) * 129 * 129 for r in range(8)]
    assert max(sizes) < 1.001 * 129 ** 3 / 8 and max(slabs) > 1.05 * 129 ** 3 / 8


def test_single_process_passthrough():
    x = torch.rand(1, 4, 4, 4)
    assert parallel.gather_slabs(x, 4) is x
    pl, cd, idx = parallel.reduce_best_rotation(0.5, 7, torch.tensor([1.0, 2.0]))
    assert cd == 0.5 and idx == 7


def _rows_worker(rank, world, initfile, n_items):
    """Dataset-sharded evaluation: DistributedSampler shards (with its padding duplicates) ->
    gather_sample_rows == the single-process table."""
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        table = torch.arange(n_items * 6, dtype=torch.float32).reshape(n_items, 6) * 0.5
        sampler = torch.utils.data.distributed.DistributedSampler(list(range(n_items)), num_replicas=world,
                                                                  rank=rank, shuffle=False, drop_last=False)
        mine = torch.tensor(list(iter(sampler)), dtype=torch.long)
        ids, (rows, col) = parallel.gather_sample_rows(mine, [table[mine], table[mine, 0]])
        assert ids.tolist() == list(range(n_items)), "rank %d: ids %s" % (rank, ids.tolist())
        assert torch.equal(rows, table) and torch.equal(col, table[:, 0])
        # ragged shards (no sampler padding): ranks hold different counts
        lo, hi = rank * n_items // world, (rank + 1) * n_items // world
        if rank == world - 1:
            hi = n_items
        own = torch.arange(lo, hi)
        ids, (rows,) = parallel.gather_sample_rows(own, [table[own]])
        assert ids.tolist() == list(range(n_items)) and torch.equal(rows, table)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_items", [(2, 5), (3, 7), (2, 4)])
def test_gather_sample_rows(world, n_items):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_rows_worker, args=(world, os.path.join(d, "init"), n_items), nprocs=world, join=True)


def test_gather_sample_rows_single_process():
    ids = torch.tensor([2, 0, 1, 1])
    vals = torch.tensor([[20.], [0.], [10.], [11.]])
    out_ids, (out,) = parallel.gather_sample_rows(ids, [vals])
    assert out_ids.tolist() == [0, 1, 2] and out[:, 0].tolist() == [0., 10., 20.]   # first occurrence kept


def _records_worker(rank, world, initfile):
    """Device-side record reduce of the sharded pose search (parallel.reduce_best_records): 16-float records, index
    as int32 bits, interleaved shares order[r::W] of a sorted sphere; cross-rank ties, NaN records and an empty
    share (more ranks than rotations) included."""
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        n_rot = 6912
        rs = np.random.RandomState(3)
        cds = rs.rand(n_rot).astype(np.float32)
        cds[[4000, 77, 6000]] = np.float32(-1.0)       # tie over (most likely) different ranks: index 77 must win
        order = np.argsort(rs.rand(n_rot), kind="stable")          # the bound-sorted order every rank derives
        for case, mine in (("interleaved", order[rank::world]), ("empty", order[:1] if rank == 0 else order[:0])):
            rec = torch.zeros(16)
            rec[0], rec[12] = float("inf"), float("inf")
            rec[1:2] = torch.tensor([0x7fffffff], dtype=torch.int32).view(torch.float32)
            if len(mine):
                loc = cds[mine]
                j = np.lexsort((mine, loc))[0]
                rec[0] = float(loc[j])
                rec[1:2] = torch.tensor([int(mine[j])], dtype=torch.int32).view(torch.float32)
                rec[2:10] = torch.arange(8) + float(mine[j])           # payload tagged with the winner's index
            rec[10:12] = torch.tensor([len(mine), rank + 1], dtype=torch.int32).view(torch.float32)
            out = parallel.reduce_best_records(rec)
            idx = int(out[1:2].view(torch.int32))
            want = 77 if case == "interleaved" else int(order[0])
            assert idx == want and float(out[0]) == float(cds[want]) and float(out[2]) == float(want), (case, rank, idx)
            counters = out[10:12].view(torch.int32).tolist()
            total = n_rot if case == "interleaved" else 1
            assert counters == [total, world * (world + 1) // 2]
        # a NaN record never wins (`cd < best_cd`, utils/eval_3D.py:162)
        rec = torch.zeros(16)
        rec[0] = float("nan") if rank == 0 else 0.5 + rank
        rec[1:2] = torch.tensor([rank], dtype=torch.int32).view(torch.float32)
        out = parallel.reduce_best_records(rec)
        assert int(out[1:2].view(torch.int32)) == 1 and float(out[0]) == 1.5
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_reduce_best_records(world):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_records_worker, args=(world, os.path.join(d, "init")), nprocs=world, join=True)


def test_reduce_best_records_single_process_passthrough():
    rec = torch.arange(16, dtype=torch.float32)
    assert parallel.reduce_best_records(rec) is rec
