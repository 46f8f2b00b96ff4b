"""Multi-GPU sharding of the hot path (one process per GPU, torch.distributed over RCCL).

The reference never shards a single image (its only parallelism is
dataset-level DDP); the path shards naturally because query points are independent
given the per-image latent and rotations are independent given the two clouds
(SURVEY.md section 8e):

  * dense grid:   rank r evaluates x-slices [r*ceil(G/W), min(G,(r+1)*ceil(G/W))) of every
                  image in the batch; x is the slowest axis of occ[B,G,G,G]
                  (utils/eval_3D.py:16-18), so slabs are contiguous; ONE padded
                  all_gather_into_tensor rebuilds the full grid on every rank.
  * brute-force:  rank r scans a contiguous range of the 6912 rotations; the ranks'
                  (cd, index) records are all-gathered and reduced lexicographically so the
                  winner is the FIRST strict minimum, like the sequential scan
                  (utils/eval_3D.py:161-168).

  * dataset-level evaluation (the reference's own parallelism, model/shape_engine.py:414-432):
                  per-sample metric rows of every rank are gathered, the DistributedSampler's
                  padding duplicates dropped, and rows ordered by sample index
                  (gather_sample_rows).

The partition / merge logic is device-agnostic (tested with gloo on CPU tensors); the
compute callbacks are the HIP kernels.
"""
import math

import torch
import torch.distributed as dist


def world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def slab_bounds(G, world_size, rank):
    """x-slice range [begin, end) of ``rank``; every rank gets ceil(G/W) slices except the
    tail ranks, which get what is left (possibly nothing)."""
    per = int(math.ceil(G / float(world_size)))
    b = min(G, rank * per)
    e = min(G, b + per)
    return b, e, per


def gather_slabs(local, G, group=None):
    """local: [B, n_local, G, G] slab of this rank (n_local may be < per on tail ranks).
    Returns the full [B, G, G, G] on every rank with ONE all_gather_into_tensor of
    equally padded slabs."""
    rank, W = world(group)
    if W == 1:
        return local
    b, e, per = slab_bounds(G, W, rank)
    B = local.shape[0]
    assert local.shape[1] == e - b
    padded = local.new_zeros((B, per, G, G))
    padded[:, : e - b] = local
    out = local.new_empty((W, B, per, G, G))
    dist.all_gather_into_tensor(out.view(-1), padded.view(-1), group=group)
    # [W, B, per, G, G] -> [B, W*per, G, G] -> trim the padding of the tail ranks
    out = out.permute(1, 0, 2, 3, 4).reshape(B, W * per, G, G)
    return out[:, :G].contiguous()


def sharded_level_grid(query_slab, G, group=None):
    """query_slab(begin, end) -> [B, end-begin, G, G] for this rank's slab; returns the
    full occupancy grid [B,G,G,G] on every rank."""
    rank, W = world(group)
    b, e, _ = slab_bounds(G, W, rank)
    local = query_slab(b, e)
    return gather_slabs(local, G, group)


def rotation_range(n_rot, world_size, rank, batch=24):
    """contiguous rotation range of ``rank``, aligned to the reference's batches of 24
    so every rank evaluates whole batches (utils/eval_3D.py:149-152)."""
    n_batches = int(math.ceil(n_rot / float(batch)))
    per = int(math.ceil(n_batches / float(world_size)))
    b = min(n_rot, rank * per * batch)
    e = min(n_rot, (rank + 1) * per * batch)
    return b, e


def reduce_best_rotation(local_cd, local_idx, payload, group=None):
    """Each rank holds its best (cd, global rotation index) and a payload vector (acc, comp,
    fscore[6], ...).  Returns the payload of the lexicographic minimum of (cd, idx) - the
    first strict minimum of the sequential scan - on every rank, plus (cd, idx)."""
    rank, W = world(group)
    rec = torch.cat([torch.tensor([float(local_cd), float(local_idx)], dtype=torch.float64,
                                  device=payload.device), payload.double().reshape(-1)])
    if W == 1:
        return payload, float(local_cd), int(local_idx)
    out = rec.new_empty((W, rec.numel()))
    dist.all_gather_into_tensor(out.view(-1), rec, group=group)
    cds, idxs = out[:, 0], out[:, 1]
    best = 0
    for r in range(1, W):
        if (cds[r] < cds[best]) or (cds[r] == cds[best] and idxs[r] < idxs[best]):
            best = r
    return out[best, 2:].to(payload.dtype).reshape(payload.shape), float(cds[best]), int(idxs[best])


def gather_sample_rows(ids, tensors, group=None):
    """Dataset-sharded evaluation (model/shape_engine.py:414-432): every rank holds the metric
    rows of its samples (ids [n_r] int64, each tensor [n_r, ...]).  Returns (ids, tensors) of ALL
    samples on every rank, sorted by id, with the duplicates a DistributedSampler appends to even
    out the shards removed.  Ragged shards are padded to the longest one for the all_gather."""
    rank, W = world(group)
    if W > 1:
        n = torch.tensor([ids.numel()], device=ids.device)
        counts = [torch.zeros_like(n) for _ in range(W)]
        dist.all_gather(counts, n, group=group)
        counts = [int(c.item()) for c in counts]
        nmax = max(counts)

        def gather(t):
            pad = t.new_zeros((nmax,) + tuple(t.shape[1:]))
            pad[:t.shape[0]] = t
            parts = [torch.zeros_like(pad) for _ in range(W)]
            dist.all_gather(parts, pad, group=group)
            return torch.cat([p[:c] for p, c in zip(parts, counts)])
        ids, tensors = gather(ids), [gather(t) for t in tensors]
    order = torch.argsort(ids, stable=True)
    ids, tensors = ids[order], [t[order] for t in tensors]
    keep = torch.ones_like(ids, dtype=torch.bool)
    keep[1:] = ids[1:] != ids[:-1]                      # first occurrence of every sample index
    return ids[keep], [t[keep] for t in tensors]
