"""Mirror of the reference's utils/eval_3D.py for the hot path: same function names,
argument meaning and return values, running on the hand-written HIP kernels.

  get_dense_3D_grid   utils/eval_3D.py:11-20
  compute_level_grid  utils/eval_3D.py:22-81   (vis_attn branch: host-side, see below)
  standardize_pc      utils/eval_3D.py:83-91
  normalize_pc        utils/eval_3D.py:93-102
  brute_force_search  utils/eval_3D.py:140-170
  compute_fscore      utils/eval_3D.py:215-231
  chamfer_distance    utils/eval_3D.py:265-269
  ICP                 utils/eval_3D.py:271-284
  eval_metrics(_default|_BF)  utils/eval_3D.py:104-138,172-213
  convert_to_explicit utils/eval_3D.py:233-263  (GPU marching cubes + sampling)

torch is used here for device memory, streams and orchestration; every numeric step -
the decoder over the dense grid, the nearest-neighbour searches, normalisation, F-score,
ICP, iso-surface extraction - is a HIP kernel behind the C ABI and raises if
libzeroshape_hip.so is missing (no fallback).

``convert_to_explicit`` (utils/eval_3D.py:233-263: PyMCubes + trimesh on the host in the
reference) runs on the GPU here: csrc/marching_cubes.hip.
"""

import numpy as np
import os

import torch

from ..external.chamfer3D.dist_chamfer_3D import chamfer_3DDist
from .camera import get_rotation_sphere


class _GridInfo(object):
    """Metadata attached to tensors made by get_dense_3D_grid so that
    compute_level_grid can use the fused grid kernel (coordinates generated in the
    kernel from the linspace axis) instead of re-reading the points tensor."""
    __slots__ = ("axis", "G", "version")

    def __init__(self, axis, G, version):
        self.axis, self.G, self.version = axis, G, version   # version: the tensor's in-place counter at tagging


@torch.no_grad()
def get_dense_3D_grid(opt, var, N=None):
    """utils/eval_3D.py:11-20: [B, N+1, N+1, N+1, 3] fp32, x slowest, z fastest."""
    batch_size = len(var.idx)
    N = N or opt.eval.vox_res
    range_min, range_max = opt.eval.range
    grid = torch.linspace(range_min, range_max, N + 1, device=opt.device)
    points_3D = torch.stack(torch.meshgrid(grid, grid, grid, indexing='ij'), dim=-1)
    points_3D = points_3D.repeat(batch_size, 1, 1, 1, 1)
    points_3D._zs_grid = _GridInfo(grid, N + 1, points_3D._version)
    return points_3D


def image_sharding(opt):
    """(rank, world) of the per-image sharding of SURVEY.md section 8e - the grid's point ranges and the pose
    search's rotations split over the ranks of ONE image - or None: on when ``opt.eval.shard_image`` is set (the
    engine resolves its "auto" default: more ranks than test images, model/shape_engine.py) and a process group of more
    than one rank exists."""
    from .. import parallel
    try:
        on = bool(opt.eval.shard_image)
    except (AttributeError, KeyError):
        on = False
    rank, world = parallel.world()
    return (rank, world) if on and world > 1 else None


@torch.no_grad()
def compute_level_grid(opt, impl_network, latent_depth, latent_semantic, points_3D, images,
                       vis_attn=False):
    """utils/eval_3D.py:22-81.  Returns (occ [B,G,G,G] in (0,1), images_vis | None).

    Fast path: ``impl_network`` is the HIP decoder and ``points_3D`` came from
    get_dense_3D_grid -> one fused launch per image batch over the whole grid
    (no per-slice loop, no points tensor read, no attention materialisation).
    Otherwise: the reference's slice loop through ``impl_network(...)``."""
    latent_depth = latent_depth.to(torch.float32) if latent_depth is not None else None
    latent_semantic = latent_semantic.to(torch.float32) if latent_semantic is not None else None
    batch_size = points_3D.shape[0]
    N = points_3D.shape[1]
    assert N == points_3D.shape[2] == points_3D.shape[3]
    assert points_3D.shape[4] == 3

    info = getattr(points_3D, "_zs_grid", None)
    if info is not None and (info.version != points_3D._version or info.G != N):
        info = None            # modified in place since get_dense_3D_grid made it: read the points
    if info is not None and hasattr(impl_network, "query_grid") and getattr(impl_network, "fused", True) and not vis_attn \
            and latent_semantic is None:
        if image_sharding(opt) is not None:
            # every rank holds the same images (and ran the same prologue); rank r evaluates its tile-aligned
            # point range of every image and ONE all_gather_into_tensor rebuilds the grids everywhere
            from .. import parallel
            state = parallel.prepare_sharded(impl_network, latent_depth)      # image i's output check on rank i % world
            occ = parallel.sharded_level_grid_points(
                lambda b, e: impl_network.query_grid_range(latent_depth, info.axis, b, e, apply_sigmoid=True,
                                                           state=state), N)
            return occ, None
        occ = impl_network.query_grid(latent_depth, info.axis, apply_sigmoid=True)
        return occ, None

    pts = points_3D.view(batch_size, N, N * N, 3)
    occ, attn = [], []
    hip_net = hasattr(impl_network, "query_grid")
    for i in range(N):
        if hip_net:   # our decoder: skip the attention dump unless it is going to be drawn
            occ_slice, attn_slice = impl_network(latent_depth, latent_semantic, pts[:, i], need_attn=vis_attn)
        else:
            occ_slice, attn_slice = impl_network(latent_depth, latent_semantic, pts[:, i])
        occ.append(occ_slice)
        if vis_attn:
            attn.append(attn_slice.detach())
    occ = torch.stack(occ, dim=1).view(batch_size, N, N, N)
    occ = torch.sigmoid(occ)
    images_vis = None
    if vis_attn:
        images_vis = _attention_frames(opt, attn, images, batch_size, N)
    return occ, images_vis


def _jet_lut():
    """The 256-entry 'jet' colour table (RGB, uint8) from its closed form - r, g, b = clamp(1.5 - |4x - 3|, |4x - 2|, |4x - 1|)
    at x = i / 255 - the map cv2.COLORMAP_JET tabulates (utils/util_vis.py:285; cv2 is not a dependency of this package,
    and its table may differ from the closed form by one grey level)."""
    x = np.arange(256, dtype=np.float64) / 255.0
    chan = lambda c: np.clip(1.5 - np.abs(4.0 * x - c), 0.0, 1.0)      # noqa: E731
    return np.uint8(np.round(255.0 * np.stack([chan(3.0), chan(2.0), chan(1.0)], -1)))


def show_att_on_image(img, mask):
    """utils/util_vis.py:267-293: the attention map [H, W] in [0, 1] as a jet heat map added onto the image [H, W, 3] in
    [0, 1], rescaled to a maximum of 1.  numpy only (round 5: the product path no longer reaches for the reference's module)."""
    assert np.max(img) <= 1 and np.max(mask) <= 1
    heatmap = np.float32(_jet_lut()[np.uint8(255 * mask)]) / 255
    merged = heatmap + np.float32(img)
    return merged / np.max(merged)


def _attention_frames(opt, attn, images, batch_size, N):
    """utils/eval_3D.py:47-80: host-side heat-map frames for the demo GIF (not part of the hot path)."""
    N_global = 1
    feat_res = opt.H // opt.arch.win_size
    attn = torch.stack(attn, dim=1).view(batch_size, N, N, N, N_global + feat_res ** 2)
    attn = torch.mean(attn, dim=3)
    attn_global = attn[:, :, :, :N_global].sum(dim=-1, keepdim=True)
    attn_local = attn[:, :, :, N_global:].view(batch_size, N, N, feat_res, feat_res)
    attn_vis = attn_global.unsqueeze(-1) + attn_local
    images_vis = []
    for b in range(batch_size):
        frames = []
        for row in range(0, N, 8):
            col_range = range(0, N // 8 * 8 + 1, 8) if row % 16 == 0 else range(N // 8 * 8, -1, -8)
            for col in col_range:
                a = attn_vis[b, col, row]
                a = torch.nn.functional.interpolate(a[None, None], size=(opt.H, opt.W), mode='bilinear',
                                                    align_corners=False)[0, 0].cpu().numpy()
                a /= a.max()
                frames.append(show_att_on_image(images[b].permute(1, 2, 0).cpu().numpy(), a))
        images_vis.append(frames)
    return images_vis


def _gpu_f32(t, what):
    if not t.is_cuda:
        raise ValueError("%s must be a GPU tensor; zeroshape_amd has no CPU path" % what)
    return t.detach().to(torch.float32).contiguous()


@torch.no_grad()
def standardize_pc(pc):
    """utils/eval_3D.py:83-91: zs_standardize_pc."""
    from .. import _lib
    assert len(pc.shape) == 3 and pc.shape[2] == 3
    lib = _lib.load()
    x = _gpu_f32(pc, "pc")
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = lib.zs_standardize_pc(_lib.ptr(x), x.shape[0], x.shape[1], _lib.ptr(out), _lib.current_stream_ptr(x.device))
    _lib.check(rc, "zs_standardize_pc")
    return out


@torch.no_grad()
def normalize_pc(pc):
    """utils/eval_3D.py:93-102 (z extent ignored, like the reference): zs_normalize_pc."""
    from .. import _lib
    assert len(pc.shape) == 3 and pc.shape[2] == 3
    lib = _lib.load()
    x = _gpu_f32(pc, "pc")
    B, n = x.shape[0], x.shape[1]
    out = torch.empty_like(x)
    scratch = torch.empty(16 * max(B, 1), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.zs_normalize_pc(_lib.ptr(x), B, n, _lib.ptr(out), _lib.ptr(scratch), _lib.current_stream_ptr(x.device))
    _lib.check(rc, "zs_normalize_pc")
    return out


_THRESHOLDS = {}


def _threshold_tensor(thresholds, device, pad_to=None):
    key = (tuple(float(t) for t in thresholds), str(device), pad_to)
    if key not in _THRESHOLDS:
        vals = list(key[0]) + [0.0] * max(0, (pad_to or 0) - len(key[0]))   # d < 0 never holds: padded entries score 0
        _THRESHOLDS[key] = torch.tensor(vals, dtype=torch.float32, device=device)
    return _THRESHOLDS[key]


@torch.no_grad()
def compute_fscore(dist1, dist2, thresholds=[0.005, 0.01, 0.02, 0.05, 0.1, 0.2]):
    """utils/eval_3D.py:215-231 on un-squared distances [B,n], [B,m] -> [B, len(thresholds)]: zs_fscore."""
    from .. import _lib
    lib = _lib.load()
    d1, d2 = _gpu_f32(dist1, "dist1"), _gpu_f32(dist2, "dist2")
    B = d1.shape[0]
    thr = _threshold_tensor(thresholds, d1.device)
    out = torch.empty(B, len(thresholds), dtype=torch.float32, device=d1.device)
    with torch.cuda.device(d1.device):
        rc = lib.zs_fscore(_lib.ptr(d1), d1.shape[1], _lib.ptr(d2), d2.shape[1], B, _lib.ptr(thr), len(thresholds),
                           _lib.ptr(out), _lib.current_stream_ptr(d1.device))
    _lib.check(rc, "zs_fscore")
    return out


_CHAMFER = chamfer_3DDist()  # stateless; the reference builds a new module per call (:267)


def chamfer_distance(opt, X1, X2, method="auto"):
    """utils/eval_3D.py:265-269: un-squared NN distances both ways + int32 indices.
    ``method`` picks the kernel ("auto" | "grid" | "brute"); outputs are bit-identical."""
    assert X1.shape[2] == 3
    dist_1, dist_2, idx_1, idx_2 = _CHAMFER(X1, X2, method)
    return dist_1.sqrt(), dist_2.sqrt(), idx_1, idx_2


def _bf_lower_bounds(pc_pred, pc_gt_n, rotations):
    """zs_bf_lower_bounds: per-rotation lower bounds of the Chamfer-L1 (csrc/bf_prune.hip)."""
    from .. import _lib
    lib = _lib.load()
    dev = pc_pred.device
    pred = pc_pred.reshape(-1, 3).contiguous()
    gt = pc_gt_n.reshape(-1, 3).contiguous()
    R = rotations.to(dev).float().contiguous()
    gw = lib.zs_bf_grid_bytes() // 4
    grids = torch.empty(2 * gw, dtype=torch.float32, device=dev)
    scratch = torch.empty(lib.zs_bf_scratch_bytes() // 4, dtype=torch.int32, device=dev)
    lb = torch.empty(R.shape[0], dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.zs_bf_lower_bounds(_lib.ptr(pred), pred.shape[0], _lib.ptr(gt), gt.shape[0], _lib.ptr(R),
                                    R.shape[0], _lib.ptr(grids), _lib.ptr(grids[gw:]), _lib.ptr(scratch),
                                    _lib.ptr(lb), _lib.current_stream_ptr(dev))
    _lib.check(rc, "zs_bf_lower_bounds")
    return lb


_POSE_MODES = {"brute": 0, "pairs": 1, "cull": 2}     # zs_pose_search_batch_sorted: mode


def _spatially_sorted(points, order=None):
    """points [n,3] (GPU, fp32) -> the same points in an order whose runs of 64 are compact boxes: "str" (default;
    sort-tile-recursive, zs_str_sort) or "morton" (Z-order curve, zs_morton_sort); ZS_POSE_ORDER overrides."""
    from .. import _lib
    lib = _lib.load()
    order = (order or os.environ.get("ZS_POSE_ORDER", "str")).lower()
    pts = points.contiguous()
    n = pts.shape[0]
    out = torch.empty_like(pts)
    st = _lib.current_stream_ptr(pts.device)
    with torch.cuda.device(pts.device):
        if order == "str":
            scratch = torch.empty(max(1, lib.zs_str_scratch_bytes(n) // 4), dtype=torch.float32, device=pts.device)
            rc = lib.zs_str_sort(_lib.ptr(pts), n, _lib.ptr(out), None, _lib.ptr(scratch), st)
        elif order == "morton":
            scratch = torch.empty(max(1, lib.zs_morton_scratch_bytes(n) // 4), dtype=torch.float32, device=pts.device)
            rc = lib.zs_morton_sort(_lib.ptr(pts), n, _lib.ptr(out), None, None, 0, _lib.ptr(scratch), st)
        else:
            raise ValueError("point order must be 'str' or 'morton', got %r" % (order,))
    _lib.check(rc, "zs_%s_sort" % order)
    return out


_ROTATIONS = {}


def _rotation_sphere(device):
    key = str(torch.device(device))
    if key not in _ROTATIONS:       # 6912 x 3 x 3, built once per device (the reference rebuilds it per call, :148)
        _ROTATIONS[key] = get_rotation_sphere(azim_sample=24, elev_sample=24, roll_sample=12, scales=[1.0],
                                              device=device).float().contiguous()
    return _ROTATIONS[key]


@torch.no_grad()
def brute_force_search(pc_pred, pc_gt, f_thresholds=[0.005, 0.01, 0.02, 0.05, 0.1, 0.2], device="cuda",
                       rotations=None, rot_slice=None, return_index=False, batch_size=256, prune=True, nn=None,
                       first_batch=None, rot_shard=None, group=None, peek=None):
    """utils/eval_3D.py:140-170: best rigid alignment over the 6912-rotation sphere by
    Chamfer-L1, with the reference's first-strict-minimum rule (:161-168).

    Same result as the exhaustive scan, less work: every rotation first gets a rigorous lower
    bound of its Chamfer-L1 (``prune``; csrc/bf_prune.hip); rotations are then evaluated exactly
    in order of increasing bound, ``batch_size`` at a time (the reference: 24 in index order,
    :149), by the fused kernels of csrc/pose_search.hip - rotate, normalize_pc, nearest
    neighbours both ways, sqrt / means / F-score, running (cd, rotation index) minimum, all on
    the device - and a batch whose smallest bound already exceeds the best exact distance returns
    at once.  No rotated clouds, no PyTorch or BLAS launch inside the search; the batches are enqueued without waiting
    for each other, except that after the 2nd and the 4th one (``peek``, default on; ZS_POSE_PEEK=0: never) the host reads the
    running best (one 64-byte copy) and STOPS enqueuing when the next batch's smallest bound already fails the kernels' own
    test - an alignable ground truth finds its winner in the first batch or two, and the 27 batches behind it were 240
    launches that returned at once: 1.1 ms of a 2.8 ms search.  Same record, bit for bit: the bounds ascend and the best only
    falls, so every batch the host skips is one the device would have skipped.  The winner is the lexicographic minimum of
    (cd, rotation index) over the evaluated rotations; pruned ones are strictly worse, so this IS
    the first strict minimum of the full scan.
    ``nn`` picks the nearest-neighbour kernels of the exact evaluations (ZS_POSE_NN overrides the default).  Both
    clouds are first put into a spatial order (sort-tile-recursive, zs_str_sort; once per search), so that 64
    consecutive points fill a compact box - also after the rotation; the statistics of normalize_pc and
    the returned clouds keep the caller's order.  "cull" (default, round 3) reads candidates 64 at a time
    through the scalar cache and skips every block whose bounding box is farther from a query than its current
    nearest neighbour, nearest tiles first; "pairs" is the same kernel without the skipping, "brute" round 2's
    LDS-staged scan of every pair of the same sorted clouds; "grid" walks uniform grids - the ground
    truth binned once per search, the rotated prediction once per rotation.  The records are bit-identical
    (tests/test_gpu_chamfer.py): every query ends with the same minimum and the sums keep their fixed order.  The
    grid walk wins 3.6x on clouds that already lie on each other (the Chamfer call of the final metrics), but most
    of the 6,912 rotations do NOT, their queries walk many rings (divergent lanes), and the whole search is
    slower than the plain scan (round 2: 1,139 vs 154 ms exhaustive) - the culled scan keeps the regular
    wave-wide inner loop and drops whole blocks instead.
    ``rot_slice=(start, stop)`` restricts the scan to a contiguous index range; ``return_index`` appends the
    winning global rotation index and its cd (one host read of the 64-byte record).
    ``rot_shard=(rank, world)`` is the multi-GPU form (every rank of ``group`` calls with the
    same clouds): every rank bounds and sorts the whole sphere (1 ms, identical on all of them), rank r takes
    positions r, r + W, ... of that order - equal shares of promising and hopeless rotations - and after its first
    batch the ranks exchange their best distance (ONE 4-byte all-reduce MIN on the stream, no host read) so every
    later batch is pruned against the GLOBAL best; at the end the 64-byte records are all-gathered and the
    lexicographic (cd, index) minimum taken on the device - the same record, bit for bit, as the single-rank scan
    (a rotation's record does not depend on the batch it was evaluated in)."""
    from .. import _lib
    lib = _lib.load()
    if len(f_thresholds) > 6:
        raise ValueError("the fused pose search carries six F-score thresholds (options/shape.yaml:54)")
    dev = torch.device(device)
    pred = pc_pred.to(dev).float().contiguous()
    pc_gt = normalize_pc(pc_gt.to(dev).unsqueeze(0).float().contiguous())
    rotations = _rotation_sphere(dev) if rotations is None else rotations.to(dev).float().contiguous()
    start, stop = (0, len(rotations)) if rot_slice is None else rot_slice
    if stop <= start:
        raise ValueError("empty rotation range")
    batch_size = min(int(batch_size), lib.zs_pose_max_batch())      # (256: the kernels address a batch by thread)
    first_batch = min(batch_size, 32) if first_batch is None else max(1, min(int(first_batch), batch_size))
    K = stop - start
    n, m = pred.shape[0], pc_gt.shape[1]
    order = lb_sorted = None
    if prune and (K > batch_size or rot_shard is not None):
        lb = _bf_lower_bounds(pred, pc_gt[0], rotations[start:stop])
        lb_sorted, order = torch.sort(lb, stable=True)
        order = order.to(torch.int32)
    if rot_shard is not None:
        srank, sworld = rot_shard
        if order is None:
            order = torch.arange(K, dtype=torch.int32, device=dev)
        # The shares only partition the sphere - and the shared running best only prunes validly - if every rank holds
        # bit-identical clouds (per-rank random weights, or a dataset that samples points per process, would
        # silently return a wrong winner).  One 16-byte MIN/MAX all-reduce of a checksum per call; a mismatch raises.
        _check_identical_inputs(pred, pc_gt, order, group)
        order = order[srank::sworld].contiguous()
        lb_sorted = lb_sorted[srank::sworld].contiguous() if lb_sorted is not None else None
        K = int(order.numel())
    thr = _threshold_tensor(f_thresholds, dev, pad_to=6)
    best = torch.empty(lib.zs_pose_best_bytes() // 4, dtype=torch.float32, device=dev)
    scratch = torch.empty(lib.zs_pose_scratch_bytes(n, m, max(1, min(batch_size, K))) // 4, dtype=torch.float32, device=dev)
    nn = (nn or os.environ.get("ZS_POSE_NN", "cull")).lower()
    if nn not in ("grid", "brute", "cull", "pairs"):
        raise ValueError("nn must be 'cull', 'pairs', 'grid' or 'brute', got %r" % (nn,))
    pred_s, gt_s = _spatially_sorted(pred), _spatially_sorted(pc_gt[0])
    gt_pack = None
    if nn != "grid":
        gt_pack = torch.empty(lib.zs_pose_pack_bytes(m) // 4, dtype=torch.float32, device=dev)
        scratch = torch.empty(lib.zs_pose_sorted_scratch_bytes(n, m, max(1, min(batch_size, K))) // 4, dtype=torch.float32,
                              device=dev)
    grids = None
    if nn == "grid":
        grids = torch.empty(lib.zs_pose_grid_bytes(n, m, max(1, min(batch_size, K))) // 4, dtype=torch.float32, device=dev)
    st = _lib.current_stream_ptr(dev)
    rot_base = rotations.data_ptr() + 36 * start

    def launch(rot_ptr, order_ptr, count, offset, lb_ptr):
        if grids is None:
            return lib.zs_pose_search_batch_sorted(_lib.ptr(pred), _lib.ptr(pred_s), n, _lib.ptr(gt_s), _lib.ptr(gt_pack), m,
                                                   rot_ptr, order_ptr, count, offset, lb_ptr, _lib.ptr(thr), _lib.ptr(best),
                                                   _lib.ptr(scratch), _POSE_MODES[nn], st)
        return lib.zs_pose_search_batch_grid(_lib.ptr(pred_s), n, _lib.ptr(gt_s), m, rot_ptr, order_ptr, count, offset, lb_ptr,
                                             _lib.ptr(thr), _lib.ptr(best), _lib.ptr(scratch), _lib.ptr(grids),
                                             _lib.ptr(pred), st)

    with torch.cuda.device(dev):
        _lib.check(lib.zs_pose_best_init(_lib.ptr(best), st), "zs_pose_best_init")
        if grids is not None:
            _lib.check(lib.zs_pose_gt_grid(_lib.ptr(gt_s), m, _lib.ptr(grids), st), "zs_pose_gt_grid")
        else:
            _lib.check(lib.zs_pose_pack(_lib.ptr(gt_s), m, _lib.ptr(gt_pack), st), "zs_pose_pack")
        # a small first batch: its winner lets the staged drop (csrc/pose_search.hip) thin out every later batch
        starts, pos = [], 0
        while pos < K:
            starts.append((pos, min(first_batch if pos == 0 else batch_size, K - pos)))
            pos += starts[-1][1]
        peek = (os.environ.get("ZS_POSE_PEEK", "1") != "0") if peek is None else bool(peek)
        lb_host = None
        for bi, (pos, count) in enumerate(starts):
            if peek and lb_sorted is not None and bi in (2, 4) and not torch.cuda.is_current_stream_capturing():
                # the kernels' batch_pruned() (csrc/pose_search.hip:75-77) on the host, in the same fp32 arithmetic
                if lb_host is None:
                    lb_host = lb_sorted.cpu().numpy()
                rec = best.cpu().numpy()
                bound = min(np.float32(rec[0]), np.float32(rec[12]))
                if np.float32(np.float32(lb_host[pos]) * np.float32(1.0 - 1e-3)) - np.float32(1e-6) > bound:
                    break
            if order is not None:      # rotation b of the batch = rotations[start + order[pos + b]]
                rc = launch(rot_base, order.data_ptr() + 4 * pos, count, start,
                            lb_sorted.data_ptr() + 4 * pos if lb_sorted is not None else None)
            else:                      # index order
                rc = launch(rot_base + 36 * pos, None, count, start + pos, None)
            _lib.check(rc, "zs_pose_search_batch")
            if rot_shard is not None and bi == 0:
                _share_running_best(best, group)
        if rot_shard is not None:
            if K == 0:                 # more ranks than rotations: this rank reports the neutral record
                _share_running_best(best, group)
            best = _reduce_best_records(best, group)
        ibest = best.view(torch.int32)
        best_pred = torch.empty(n, 3, dtype=torch.float32, device=dev)
        rc = lib.zs_pose_apply(_lib.ptr(pred), n, _lib.ptr(rotations), ibest.data_ptr() + 4, _lib.ptr(best_pred),
                               _lib.ptr(scratch), st)
        _lib.check(rc, "zs_pose_apply")
    out = (best[2].clone(), best[3].clone(), best[4:4 + len(f_thresholds)].clone(), best_pred, pc_gt)
    brute_force_search.last_evaluated = brute_force_search.last_scanned = None
    if return_index:
        rec = best.cpu()
        irec = rec.view(torch.int32)
        brute_force_search.last_evaluated = int(irec[10])        # diagnostics: rotations the lower bounds let through
        brute_force_search.last_scanned = int(irec[11])          # ... of which scanned in full (not killed by the probe)
        out = out + (int(irec[1]), float(rec[0]))
    return out


def _check_identical_inputs(pred, gt, order, group=None):
    """Raise unless every rank of `group` passed the same prediction / ground-truth clouds and derived the same rotation order
    (bit patterns summed as int64: order-independent per tensor, but any differing element changes it)."""
    import torch.distributed as dist
    sums = torch.stack([pred.contiguous().view(torch.int32).to(torch.int64).sum(),
                        gt.contiguous().view(torch.int32).to(torch.int64).sum(),
                        (order.to(torch.int64) * torch.arange(1, order.numel() + 1, device=order.device)).sum()])
    lo, hi = sums.clone(), sums.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    if not bool((lo == hi).all()):
        raise RuntimeError("brute_force_search(rot_shard=...): the ranks hold different prediction / ground-truth clouds "
                           "(checksums %s vs %s); sharding the rotation sphere needs identical inputs on every rank "
                           "(same weights, same sample)" % (lo.tolist(), hi.tolist()))


def _share_running_best(best, group=None):
    """best[12] <- min over the ranks of best[0] (csrc/pose_search.hip: prune_bound).  Stream-ordered, no host read."""
    import torch.distributed as dist
    bound = best[0:1].clone()
    dist.all_reduce(bound, op=dist.ReduceOp.MIN, group=group)
    best[12:13].copy_(bound)


def _reduce_best_records(best, group=None):
    """The ranks' 64-byte records all-gathered; returns the record of the lexicographic (cd, rotation index) minimum
    - the first strict minimum of the sequential scan (utils/eval_3D.py:161-168) - with the evaluated / scanned
    counters summed over the ranks.  Device-side throughout."""
    from .. import parallel
    return parallel.reduce_best_records(best, group)


def _surface_clouds(opt, level_vox, seed=0):
    """level grids [B,G,G,G] (GPU tensor) -> (meshes, dpc_pred [B,num_points,3] fp32 on the
    device): convert_to_explicit without the device->host->device round trip of the
    reference (utils/eval_3D.py:115-118,181-184)."""
    meshes, clouds = [], []
    range_min, range_max = opt.eval.range
    for i in range(level_vox.shape[0]):
        tris, pts = extract_surface(level_vox[i], 0.5, range_min, range_max, opt.eval.num_points, seed + i)
        meshes.append(SimpleMesh(tris.cpu().numpy()))
        clouds.append(pts)
    return meshes, torch.stack(clouds, dim=0)


def _gt_to_view_frame(opt, var):
    """utils/eval_3D.py:120-123 / :186-190: the ground truth rotated into the view frame (R_gt p, no translation; x and y
    negated for pix3d).  One launch of zs_transform_points (round 6: the reference's `R @ p^T` was a rocBLAS batched GEMM on
    this path) - the pix3d sign flip rides on the rotation's first two rows, which negates exactly."""
    from . import camera
    pose = var.pose_gt.detach().to(torch.float32)
    B = pose.shape[0]
    T = torch.zeros(B, 3, 4, dtype=torch.float32, device=pose.device)
    T[:, :, :3] = pose[..., :3]
    if opt.data.dataset_test == 'pix3d':
        T[:, :2] = -T[:, :2]
    zero, one = torch.zeros(B, 3, device=pose.device), torch.ones(B, device=pose.device)
    var.dpc.points = camera.transform_points(var.dpc.points, T, zero, one)


@torch.no_grad()
def eval_metrics_default(opt, var, impl_network, vis_only=False):
    """utils/eval_3D.py:104-138."""
    points_3D = get_dense_3D_grid(opt, var)
    batch_size = points_3D.shape[0]
    level_vox, attn_vis = compute_level_grid(opt, impl_network, var.latent_depth, var.latent_semantic,
                                             points_3D, var.rgb_input_map, vis_only)
    if attn_vis:
        var.attn_vis = attn_vis
    var.eval_vox = points_3D.view(batch_size, -1, 3)
    var.mesh_pred, var.dpc_pred = _surface_clouds(opt, level_vox)
    _gt_to_view_frame(opt, var)
    var.dpc_pred = normalize_pc(var.dpc_pred)
    var.dpc.points = normalize_pc(var.dpc.points)
    if vis_only:
        return
    if opt.eval.icp:
        var.dpc_pred = ICP(opt, var.dpc_pred, var.dpc.points)
    dist_acc, dist_comp, _, _ = chamfer_distance(opt, X1=var.dpc_pred, X2=var.dpc.points)
    var.f_score = compute_fscore(dist_acc, dist_comp, opt.eval.f_thresholds)
    assert dist_acc.shape[1] == opt.eval.num_points
    var.cd_acc = dist_acc.mean(dim=1)
    var.cd_comp = dist_comp.mean(dim=1)
    return dist_acc.mean(), dist_comp.mean()


def eval_metrics_BF(opt, var, impl_network, vis_only=False):
    """utils/eval_3D.py:172-207."""
    points_3D = get_dense_3D_grid(opt, var)
    batch_size = points_3D.shape[0]
    level_vox, attn_vis = compute_level_grid(opt, impl_network, var.latent_depth, var.latent_semantic,
                                             points_3D, var.rgb_input_map, vis_only)
    if attn_vis:
        var.attn_vis = attn_vis
    var.eval_vox = points_3D.view(batch_size, -1, 3)
    var.mesh_pred, var.dpc_pred = _surface_clouds(opt, level_vox)
    _gt_to_view_frame(opt, var)
    if vis_only:
        return
    cd_acc, cd_comp, f_score = [], [], []
    shard = image_sharding(opt)
    for i in range(batch_size):
        best_acc, best_comp, best_fscore, best_pred, best_gt = \
            brute_force_search(var.dpc_pred[i], var.dpc.points[i], opt.eval.f_thresholds, opt.device, rot_shard=shard)
        var.dpc_pred[i] = best_pred.clone()
        var.dpc.points[i] = best_gt.clone()
        cd_acc.append(best_acc)
        cd_comp.append(best_comp)
        f_score.append(best_fscore)
    var.cd_acc = torch.stack(cd_acc, dim=0)
    var.cd_comp = torch.stack(cd_comp, dim=0)
    var.f_score = torch.stack(f_score, dim=0)
    return var.cd_acc.mean(), var.cd_comp.mean()


def eval_metrics(opt, var, impl_network, vis_only=False):
    """utils/eval_3D.py:209-213."""
    if opt.eval.brute_force:
        return eval_metrics_BF(opt, var, impl_network, vis_only)
    return eval_metrics_default(opt, var, impl_network, vis_only)


@torch.no_grad()
def ICP(opt, X1, X2, num_iter=50):
    """utils/eval_3D.py:271-284: every iteration is the Chamfer call + zs_icp_step (centroids, 3x3 cross-covariance, its SVD
    by Jacobi rotations, the reference's sign rule and the rigid update on the device: no ATen / LAPACK call, no host
    read in the loop)."""
    from .. import _lib
    assert len(X1) == len(X2)
    lib = _lib.load()
    x1, x2 = _gpu_f32(X1, "X1"), _gpu_f32(X2, "X2")
    B, n, m = x1.shape[0], x1.shape[1], x2.shape[1]
    scratch = torch.empty(max(1, lib.zs_icp_scratch_bytes(B) // 4), dtype=torch.float32, device=x1.device)
    nxt = torch.empty_like(x1)
    if x1.data_ptr() == X1.data_ptr():
        x1 = x1.clone()                         # the caller's tensor is not updated in place (the reference rebinds X1)
    for it in range(num_iter):
        _, _, idx, _ = chamfer_distance(opt, x1, x2)
        with torch.cuda.device(x1.device):
            rc = lib.zs_icp_step(_lib.ptr(x1), n, _lib.ptr(x2), m, _lib.ptr(idx), B, _lib.ptr(nxt), _lib.ptr(scratch),
                                 _lib.current_stream_ptr(x1.device))
        _lib.check(rc, "zs_icp_step")
        x1, nxt = nxt, x1
    return x1


class SimpleMesh(object):
    """Stand-in for the ``trimesh.Trimesh`` objects the reference stores in ``var.mesh_pred`` (dumps and the demo's .obj):
    ``triangles`` [n,3,3] = the iso-surface kernel's triangle soup (what sampling reads); ``vertices`` [v,3] / ``faces`` [n,3] =
    the INDEXED form ``mcubes.marching_cubes`` returns (utils/eval_3D.py:250-256), welded lazily on first access: a vertex on
    a grid edge is computed from the same two corner values by every cube sharing the edge, so equal vertices are equal bit
    for bit and welding = unique rows of the soup (``vertices[faces]`` reproduces ``triangles`` exactly)."""

    def __init__(self, triangles):
        self.triangles = np.asarray(triangles, np.float32).reshape(-1, 3, 3)
        self._indexed = None

    def _weld(self):
        if self._indexed is None:
            soup = np.ascontiguousarray(self.triangles.reshape(-1, 3))
            if len(soup) == 0:
                self._indexed = (np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int64))
            else:
                keys = (soup + np.float32(0.0)).view(np.uint32)          # -0.0 and 0.0 are one vertex
                _, first, inverse = np.unique(keys, axis=0, return_index=True, return_inverse=True)
                self._indexed = (soup[first], inverse.reshape(-1, 3).astype(np.int64))
        return self._indexed

    @property
    def vertices(self):
        return self._weld()[0]

    @property
    def faces(self):
        return self._weld()[1]


_MC_TABLES = {}


def _mc_tables(device):
    key = str(device)
    if key not in _MC_TABLES:
        from .. import mc_tables as T
        stride = 16
        tab = -np.ones((256, stride), np.int8)
        tab[:, :T.TRI_TABLE.shape[1]] = T.TRI_TABLE
        _MC_TABLES[key] = (torch.from_numpy(tab).to(device), torch.from_numpy(T.TRI_COUNT.astype(np.uint8)).to(device),
                           stride)
    return _MC_TABLES[key]


@torch.no_grad()
def extract_surface(level_vox, isoval, range_min, range_max, num_points=0, seed=0):
    """GPU marching cubes (+ optional area-weighted sampling) of ONE level grid [G,G,G] that
    already lives on the device.  Returns (triangles [n,3,3] fp32 GPU tensor,
    points [num_points,3] fp32 GPU tensor | None).  Vertex scaling reproduces the reference's
    ``v / S * (max - min) + min`` with S = G (utils/eval_3D.py:252-255)."""
    from .. import _lib
    lib = _lib.load()
    vol = level_vox.detach().to(torch.float32).contiguous()
    assert vol.dim() == 3 and vol.shape[0] == vol.shape[1] == vol.shape[2] and vol.is_cuda
    G = vol.shape[0]
    dev = vol.device
    tab, cnt, stride = _mc_tables(dev)
    scratch = torch.empty(lib.zs_mc_scratch_bytes(G) // 4 + 1, dtype=torch.int32, device=dev)
    total = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = _lib.current_stream_ptr(dev)
    with torch.cuda.device(dev):
        _lib.check(lib.zs_mc_count(_lib.ptr(vol), G, float(isoval), _lib.ptr(cnt), _lib.ptr(scratch),
                                   _lib.ptr(total), stream), "zs_mc_count")
        n_tris = int(total.item())                       # 4-byte D2H: sizes the output
        tris = torch.empty(n_tris, 3, 3, dtype=torch.float32, device=dev)
        scale = np.float32((range_max - range_min) / G)
        _lib.check(lib.zs_mc_emit(_lib.ptr(vol), G, float(isoval), _lib.ptr(tab), stride, _lib.ptr(cnt),
                                  _lib.ptr(scratch), float(scale), float(range_min), _lib.ptr(tris), n_tris,
                                  stream), "zs_mc_emit")
        pts = None
        if num_points:
            pts = torch.empty(num_points, 3, dtype=torch.float32, device=dev)
            cum = torch.empty(max(lib.zs_mesh_sample_scratch_doubles(n_tris), 1), dtype=torch.float64, device=dev)
            _lib.check(lib.zs_mesh_sample(_lib.ptr(tris), n_tris, num_points, int(seed) & ((1 << 64) - 1),
                                          _lib.ptr(cum), _lib.ptr(pts), stream), "zs_mesh_sample")
    return tris, pts


def convert_to_explicit(opt, level_grids, isoval=0., to_pointcloud=False, seed=0):
    """utils/eval_3D.py:233-263 on the GPU: marching cubes at ``isoval`` + area-weighted surface
    sampling, one level grid at a time (no Python threads, no PyMCubes / trimesh).
    ``level_grids``: list of [G,G,G] arrays or GPU tensors (a numpy grid is uploaded - pass
    the tensor compute_level_grid returned to keep the grid in HBM).  Returns ``meshes``
    (SimpleMesh list) and, with ``to_pointcloud``, ``pointclouds`` float64 [B,num_points,3]
    like the reference (an empty mesh gives zeros, :262).  The random stream is a
    counter-based generator seeded by ``seed`` + the grid index, not numpy's global RNG."""
    device = getattr(opt, "device", "cuda")
    meshes, clouds = [], []
    for i, g in enumerate(level_grids):
        vol = g if isinstance(g, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(g, np.float32))
        vol = vol.to(device)
        assert vol.shape[0] == vol.shape[1] == vol.shape[2]
        range_min, range_max = opt.eval.range
        tris, pts = extract_surface(vol, isoval, range_min, range_max,
                                    opt.eval.num_points if to_pointcloud else 0, seed + i)
        meshes.append(SimpleMesh(tris.cpu().numpy()))
        if to_pointcloud:
            clouds.append(pts.cpu().numpy().astype(np.float64))
    if to_pointcloud:
        return meshes, np.stack(clouds, axis=0)
    return meshes
