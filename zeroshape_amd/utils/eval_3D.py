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

torch is used here for device memory, streams and the tiny elementwise glue the
reference also does in torch (means, extents, thresholds); the heavy steps - the
decoder over the dense grid and the nearest-neighbour search - are the HIP
kernels, and they raise if libzeroshape_hip.so is missing (no fallback).

Out of scope here: convert_to_explicit / eval_metrics*
need PyMCubes + trimesh on the host (utils/eval_3D.py:233-263), neither of which
is in this image; ``convert_to_explicit`` imports them lazily and raises a clear
error when absent.
"""
import threading

import numpy as np
import torch

from ..external.chamfer3D.dist_chamfer_3D import chamfer_3DDist
from .camera import get_rotation_sphere


class _GridInfo(object):
    """Metadata attached to tensors made by get_dense_3D_grid so that
    compute_level_grid can use the fused grid kernel (coordinates generated in the
    kernel from the linspace axis) instead of re-reading the points tensor."""
    __slots__ = ("axis", "G")

    def __init__(self, axis, G):
        self.axis, self.G = axis, G


@torch.no_grad()
def get_dense_3D_grid(opt, var, N=None):
    """utils/eval_3D.py:11-20: [B, N+1, N+1, N+1, 3] fp32, x slowest, z fastest."""
    batch_size = len(var.idx)
    N = N or opt.eval.vox_res
    range_min, range_max = opt.eval.range
    grid = torch.linspace(range_min, range_max, N + 1, device=opt.device)
    points_3D = torch.stack(torch.meshgrid(grid, grid, grid, indexing='ij'), dim=-1)
    points_3D = points_3D.repeat(batch_size, 1, 1, 1, 1)
    points_3D._zs_grid = _GridInfo(grid, N + 1)
    return points_3D


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
    if info is not None and hasattr(impl_network, "query_grid") and not vis_attn \
            and latent_semantic is None:
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


def _attention_frames(opt, attn, images, batch_size, N):
    """utils/eval_3D.py:47-80: host-side heat-map frames for the demo GIF.  Needs the
    reference's utils.util_vis.show_att_on_image (cv2) -> not part of the hot path."""
    try:
        from utils.util_vis import show_att_on_image  # the reference's own helper, if on path
    except Exception as e:  # pragma: no cover
        raise RuntimeError("vis_attn=True needs the reference's utils.util_vis (cv2) on sys.path") from e
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


@torch.no_grad()
def standardize_pc(pc):
    """utils/eval_3D.py:83-91."""
    assert len(pc.shape) == 3
    pc_mean = pc.mean(dim=1, keepdim=True)
    pc_zmean = pc - pc_mean
    origin_distance = (pc_zmean ** 2).sum(dim=2, keepdim=True).sqrt()
    scale = torch.sqrt(torch.sum(origin_distance ** 2, dim=1, keepdim=True) / pc.shape[1])
    return pc_zmean / (scale * 2)


@torch.no_grad()
def normalize_pc(pc):
    """utils/eval_3D.py:93-102 (z extent ignored, like the reference)."""
    assert len(pc.shape) == 3
    pc_mean = pc.mean(dim=1, keepdim=True)
    pc_zmean = pc - pc_mean
    length_x = pc_zmean[:, :, 0].max(dim=-1)[0] - pc_zmean[:, :, 0].min(dim=-1)[0]
    length_y = pc_zmean[:, :, 1].max(dim=-1)[0] - pc_zmean[:, :, 1].min(dim=-1)[0]
    length_max = torch.stack([length_x, length_y], dim=-1).max(dim=-1)[0].unsqueeze(-1).unsqueeze(-1)
    return pc_zmean / (length_max + 1.e-7)


def compute_fscore(dist1, dist2, thresholds=[0.005, 0.01, 0.02, 0.05, 0.1, 0.2]):
    """utils/eval_3D.py:215-231."""
    fscores = []
    for threshold in thresholds:
        precision = torch.mean((dist1 < threshold).float(), dim=1)
        recall = torch.mean((dist2 < threshold).float(), dim=1)
        fscore = 2 * precision * recall / (precision + recall)
        fscore[torch.isnan(fscore)] = 0
        fscores.append(fscore)
    return torch.stack(fscores, dim=1)


_CHAMFER = chamfer_3DDist()  # stateless; the reference builds a new module per call (:267)


def chamfer_distance(opt, X1, X2):
    """utils/eval_3D.py:265-269: un-squared NN distances both ways + int32 indices."""
    assert X1.shape[2] == 3
    dist_1, dist_2, idx_1, idx_2 = _CHAMFER(X1, X2)
    return dist_1.sqrt(), dist_2.sqrt(), idx_1, idx_2


def brute_force_search(pc_pred, pc_gt, f_thresholds=[0.005, 0.01, 0.02, 0.05, 0.1, 0.2], device="cuda",
                       rotations=None, rot_slice=None, return_index=False, batch_size=192):
    """utils/eval_3D.py:140-170: best rigid alignment over the 6912-rotation sphere by
    Chamfer-L1.  Same scan order and the same strict-first-minimum rule, but rotations are
    evaluated ``batch_size`` at a time (the reference: 24, :149 - every rotation is
    independent, so the batch size only changes the number of launches), the winner of each
    batch is picked on the device (one argmin + one sync per batch instead of one
    ``if cd[j] < best_cd`` sync per rotation, :161-168) and the rotation table is cached.  ``rot_slice=(start, stop)`` restricts the scan to a contiguous range of
    rotation indices (multi-GPU sharding, see zeroshape_amd/parallel.py);
    ``return_index`` appends the winning global rotation index and its cd."""
    pc_pred = pc_pred.to(device).unsqueeze(0).float()
    pc_gt = pc_gt.to(device).unsqueeze(0).float().contiguous()
    pc_gt = normalize_pc(pc_gt)
    if rotations is None:
        rotations = get_rotation_sphere(azim_sample=24, elev_sample=24, roll_sample=12, scales=[1.0],
                                        device=device)
    start, stop = (0, len(rotations)) if rot_slice is None else rot_slice
    best_cd = np.inf
    best = None
    for i in range(start, stop, batch_size):
        rotation_batch = rotations[i:min(i + batch_size, stop)].to(device)
        nb = rotation_batch.shape[0]
        pc_pred_rotated = (rotation_batch @ pc_pred.repeat(nb, 1, 1).permute(0, 2, 1)).permute(0, 2, 1)
        pc_pred_rotated = normalize_pc(pc_pred_rotated).contiguous()
        acc, comp, _, _ = chamfer_distance(None, pc_pred_rotated, pc_gt.repeat(nb, 1, 1).contiguous())
        f_score = compute_fscore(acc, comp, f_thresholds)
        acc, comp = acc.mean(dim=1), comp.mean(dim=1)
        cd = (acc + comp) / 2
        j = int(torch.argmin(cd))            # first minimum of the batch (ties -> lowest j)
        cd_j = float(cd[j])
        if cd_j < best_cd:                    # strict: an equal later batch does not win
            best_cd = cd_j
            best = (acc[j], comp[j], f_score[j], pc_pred_rotated[j].clone(), i + j)
    if best is None:
        raise ValueError("empty rotation range")
    out = (best[0], best[1], best[2], best[3], pc_gt)
    if return_index:
        out = out + (best[4], best_cd)
    return out


def ICP(opt, X1, X2, num_iter=50):
    """utils/eval_3D.py:271-284."""
    assert len(X1) == len(X2)
    for it in range(num_iter):
        d1, d2, idx, _ = chamfer_distance(opt, X1, X2)
        X2_corresp = torch.zeros_like(X1)
        for i in range(len(X1)):
            X2_corresp[i] = X2[i][idx[i].long()]
        t1 = X1.mean(dim=-2, keepdim=True)
        t2 = X2_corresp.mean(dim=-2, keepdim=True)
        U, S, V = ((X1 - t1).transpose(1, 2) @ (X2_corresp - t2)).svd(some=True)
        R = V @ U.transpose(1, 2)
        R[R.det() < 0, 2] *= -1
        X1 = (X1 - t1) @ R.transpose(1, 2) + t2
    return X1


def convert_to_explicit(opt, level_grids, isoval=0., to_pointcloud=False):
    """utils/eval_3D.py:233-263 - third-party boundary (PyMCubes + trimesh on the host).
    Kept call-compatible; raises when the packages are not installed."""
    try:
        import mcubes
        import trimesh
    except ImportError as e:
        raise RuntimeError("convert_to_explicit needs PyMCubes and trimesh on the host "
                           "(SURVEY.md section 8f rank 1: GPU marching cubes is a later row)") from e
    N = len(level_grids)
    meshes = [None] * N
    pointclouds = [None] * N if to_pointcloud else None

    def worker(i):
        vertices, faces = mcubes.marching_cubes(level_grids[i], isovalue=isoval)
        S = level_grids[i].shape[0]
        range_min, range_max = opt.eval.range
        vertices = vertices / S * (range_max - range_min) + range_min
        mesh = trimesh.Trimesh(vertices, faces)
        meshes[i] = mesh
        if pointclouds is not None:
            pointclouds[i] = mesh.sample(opt.eval.num_points) if len(mesh.triangles) != 0 \
                else np.zeros([opt.eval.num_points, 3])

    threads = [threading.Thread(target=worker, args=(i,), daemon=False) for i in range(N)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if to_pointcloud:
        return meshes, np.stack(pointclouds, axis=0)
    return meshes
