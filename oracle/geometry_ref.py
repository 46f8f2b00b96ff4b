"""Oracle: CPU restatement of the Chamfer-based 3-D metrics around the native
kernel.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates
  utils/eval_3D.py:93-102    normalize_pc
  utils/eval_3D.py:215-231   compute_fscore
  utils/eval_3D.py:265-269   chamfer_distance  (sqrt of the kernel's squared NN distance)
  utils/eval_3D.py:140-170   brute_force_search
  utils/camera.py:150-206    azim/elev/roll_to_rotation_matrix ('angle' representation)
  utils/camera.py:208-230    get_rotation_sphere
with torch-CPU fp32 ops (same ops as the reference, so elementwise results are
bit-identical to the reference run on CPU) and oracle/chamfer_ref.c for the
nearest-neighbour step.  Pinned by tests/golden/geometry_*.npz.
"""
import numpy as np
import torch

from . import chamfer_ref


@torch.no_grad()
def normalize_pc(pc):
    """utils/eval_3D.py:93-102: centre on the mean, divide by max(x-extent, y-extent)+1e-7
    (the z extent is ignored by the reference)."""
    assert pc.dim() == 3
    pc_mean = pc.mean(dim=1, keepdim=True)
    pc_zmean = pc - pc_mean
    length_x = pc_zmean[:, :, 0].max(dim=-1)[0] - pc_zmean[:, :, 0].min(dim=-1)[0]
    length_y = pc_zmean[:, :, 1].max(dim=-1)[0] - pc_zmean[:, :, 1].min(dim=-1)[0]
    length_max = torch.stack([length_x, length_y], dim=-1).max(dim=-1)[0].unsqueeze(-1).unsqueeze(-1)
    return pc_zmean / (length_max + 1.0e-7)


@torch.no_grad()
def compute_fscore(dist1, dist2, thresholds=(0.005, 0.01, 0.02, 0.05, 0.1, 0.2)):
    """utils/eval_3D.py:215-231."""
    out = []
    for th in thresholds:
        precision = torch.mean((dist1 < th).float(), dim=1)
        recall = torch.mean((dist2 < th).float(), dim=1)
        f = 2 * precision * recall / (precision + recall)
        f[torch.isnan(f)] = 0
        out.append(f)
    return torch.stack(out, dim=1)


@torch.no_grad()
def chamfer_distance(X1, X2):
    """utils/eval_3D.py:265-269: un-squared NN distances both ways + indices."""
    assert X1.shape[2] == 3
    d1, d2, i1, i2 = chamfer_ref.chamfer_forward(X1.contiguous().numpy(), X2.contiguous().numpy())
    return (torch.from_numpy(d1).sqrt(), torch.from_numpy(d2).sqrt(),
            torch.from_numpy(i1), torch.from_numpy(i2))


def standardize_pc(pc):
    """utils/eval_3D.py:83-91: zero mean, RMS distance from the origin 1/2."""
    assert len(pc.shape) == 3
    pc_zmean = pc - pc.mean(dim=1, keepdim=True)
    origin_distance = (pc_zmean ** 2).sum(dim=2, keepdim=True).sqrt()
    scale = torch.sqrt(torch.sum(origin_distance ** 2, dim=1, keepdim=True) / pc.shape[1])
    return pc_zmean / (scale * 2)


def icp(X1, X2, num_iter=50):
    """utils/eval_3D.py:271-284: point-to-point ICP of X1 [B,n,3] onto X2 [B,m,3] - nearest neighbours (the Chamfer
    kernel's indices), centroids, R = V U^T from the SVD of the 3x3 cross-covariance, the reference's own sign rule
    (row 2 of R negated when det R < 0), X1 <- (X1 - t1) R^T + t2."""
    assert len(X1) == len(X2)
    for _ in range(num_iter):
        _, _, idx, _ = chamfer_distance(X1, X2)
        X2c = torch.stack([X2[i][idx[i].long()] for i in range(len(X1))])
        t1, t2 = X1.mean(dim=-2, keepdim=True), X2c.mean(dim=-2, keepdim=True)
        U, S, V = ((X1 - t1).transpose(1, 2) @ (X2c - t2)).svd(some=True)
        R = V @ U.transpose(1, 2)
        R[R.det() < 0, 2] *= -1
        X1 = (X1 - t1) @ R.transpose(1, 2) + t2
    return X1


def _rot_y(azim_deg):
    # utils/camera.py:150-167 ('angle')
    a = torch.tensor([azim_deg]) * np.pi / 180
    c, s = torch.cos(a), torch.sin(a)
    R = torch.eye(3)[None].repeat(1, 1, 1)
    z = torch.zeros(1)
    R[:, 0, :] = torch.stack([c, z, s], dim=-1)
    R[:, 2, :] = torch.stack([-s, z, c], dim=-1)
    return R


def _rot_x(elev_deg):
    # utils/camera.py:169-185
    a = torch.tensor([elev_deg]) * np.pi / 180
    c, s = torch.cos(a), torch.sin(a)
    R = torch.eye(3)[None].repeat(1, 1, 1)
    R[:, 1, 1:] = torch.stack([c, -s], dim=-1)
    R[:, 2, 1:] = torch.stack([s, c], dim=-1)
    return R


def _rot_z(roll_deg):
    # utils/camera.py:187-206
    a = torch.tensor([roll_deg]) * np.pi / 180
    c, s = torch.cos(a), torch.sin(a)
    R = torch.eye(3)[None].repeat(1, 1, 1)
    R[:, 0, :2] = torch.stack([c, s], dim=-1)
    R[:, 1, :2] = torch.stack([-s, c], dim=-1)
    return R


def rotation_sphere(azim_sample=24, elev_sample=24, roll_sample=12, scales=(1.0,)):
    """utils/camera.py:208-230: R = scale * Rz @ Rx @ Ry @ R_permute, azim outermost,
    roll innermost.  float64 angles from np.linspace feed torch.tensor -> float64
    trig, then .float() (as the reference does)."""
    azims = np.linspace(0, 360, num=azim_sample, endpoint=False)
    elevs = np.linspace(0, 360, num=elev_sample, endpoint=False)
    rolls = np.linspace(0, 360, num=roll_sample, endpoint=False)
    P = torch.tensor([[-1, 0, 0], [0, 0, -1], [0, -1, 0]]).float().unsqueeze(0)
    out = []
    for scale in scales:
        for azim in azims:
            for elev in elevs:
                for roll in rolls:
                    Ry, Rx, Rz = _rot_y(azim), _rot_x(elev), _rot_z(roll)
                    out.append((scale * Rz @ Rx @ Ry @ P).float())
    return torch.cat(out, dim=0)


@torch.no_grad()
def brute_force_search(pc_pred, pc_gt, f_thresholds=(0.005, 0.01, 0.02, 0.05, 0.1, 0.2),
                       rotations=None, batch_size=24):
    """utils/eval_3D.py:140-170.  Returns (best_acc, best_comp, best_fscore, best_pc_pred,
    pc_gt_normalised, best_index).  ``best_index`` is extra (the reference keeps
    best_rotation instead); ``rotations`` lets tests pass a subset."""
    pc_pred = pc_pred.unsqueeze(0).float()
    pc_gt = normalize_pc(pc_gt.unsqueeze(0).float().contiguous())
    if rotations is None:
        rotations = rotation_sphere()
    best_cd = np.inf
    best = None
    for i in range(0, len(rotations), batch_size):
        rb = rotations[i:i + batch_size]
        rot = (rb @ pc_pred.repeat(rb.shape[0], 1, 1).permute(0, 2, 1)).permute(0, 2, 1)
        rot = normalize_pc(rot).contiguous()
        acc, comp, _, _ = chamfer_distance(rot, pc_gt.repeat(rb.shape[0], 1, 1).contiguous())
        f = compute_fscore(acc, comp, f_thresholds)
        acc, comp = acc.mean(dim=1), comp.mean(dim=1)
        cd = (acc + comp) / 2
        for j in range(len(cd)):
            if cd[j] < best_cd:  # strict: first minimum wins (utils/eval_3D.py:162)
                best_cd = cd[j]
                best = (acc[j], comp[j], f[j], rot[j].clone(), i + j)
    return best[0], best[1], best[2], best[3], pc_gt, best[4]
