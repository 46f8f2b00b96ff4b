"""Oracle: CPU restatement of the seen-surface geometry front-end and the depth metrics.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates
  model/compute_graph/graph_shape.py:89-113   Graph.intr_param2mtx
  utils/camera.py:80-108                      get_pixel_grid, unproj_depth
  utils/camera.py:52-78                       valid_norm_fac
  model/compute_graph/graph_shape.py:131-144  normalise + zero invalid + resample
  utils/util.py:323-345                       interpolate_depth / interpolate_coordmap
  utils/eval_depth.py:11-34,46-116            DepthMetric
with torch-CPU fp32 ops.  Pinned by tests/golden/frontend_golden.npz, generated from the
reference's own functions by tests/golden/make_frontend_golden.py.
"""
import torch
import torch.nn.functional as F


@torch.no_grad()
def intr_param2mtx(H, W, params):
    """graph_shape.py:89-113: focal = 1.3875 * size * 4^tanh(p0); principal point =
    size/2 * (1 + tanh(p1|p2))."""
    p = params.float()
    zoom = torch.pow(4.0, torch.tanh(p[:, 0]))
    K = torch.zeros(p.shape[0], 3, 3)
    K[:, 0, 0] = 1.3875 * W * zoom
    K[:, 1, 1] = 1.3875 * H * zoom
    K[:, 0, 2] = W / 2 + torch.tanh(p[:, 1]) * W / 2
    K[:, 1, 2] = H / 2 + torch.tanh(p[:, 2]) * H / 2
    K[:, 2, 2] = 1
    return K


@torch.no_grad()
def unproj_depth(depth, intr):
    """utils/camera.py:88-108: rays = K^-1 @ [x, y, 1]^T for the H*W pixels (x fastest), times
    the depth of the pixel."""
    B, _, H, W = depth.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32),
                            indexing="ij")
    pix = torch.stack([xs, ys, torch.ones_like(xs)], -1).reshape(1, -1, 3)          # [1,HW,3]
    rays = torch.linalg.inv(intr).float() @ pix.expand(B, -1, -1).transpose(1, 2)   # [B,3,HW]
    return rays.transpose(1, 2) * depth.reshape(B, H * W, 1)


@torch.no_grad()
def valid_norm_fac(points, mask):
    """utils/camera.py:52-78: per sample, mean of the selected points and the largest distance of
    a selected point from that mean."""
    B, n = points.shape[:2]
    sel = mask.reshape(B, n)
    means, radii = [], []
    for b in range(B):
        p = points[b][sel[b]]
        mu = p.mean(0)
        means.append(mu)
        radii.append((p - mu).norm(dim=1).max())
    return torch.stack(means), torch.stack(radii)


@torch.no_grad()
def masked_resample(map_, mask, size, bg=0.0):
    """utils/util.py:323-345 (interpolate_depth with bg=20, interpolate_coordmap with bg=0)."""
    m = (mask > 0.5).float()
    num = F.interpolate(map_ * m, size, mode="bilinear", align_corners=False)
    den = F.interpolate(m, size, mode="bilinear", align_corners=False)
    keep = (den > 0.5).float()
    return (num / (den + 1.0e-6)) * keep + bg * (1 - keep), keep


@torch.no_grad()
def seen_surface(depth, intr, mask, dsp):
    """graph_shape.py:131-144.  Returns (seen_points [B,HW,3], coord_dsp [B,3,H/dsp,W/dsp],
    mask_dsp [B,1,H/dsp,W/dsp], mean [B,3], scale [B])."""
    B, _, H, W = depth.shape
    pts = unproj_depth(depth, intr)
    mean, scale = valid_norm_fac(pts, mask > 0.5)
    seen = (pts - mean[:, None]) / scale[:, None, None]
    seen[(mask <= 0.5).reshape(B, -1)] = 0
    seen_map = seen.reshape(B, H, W, 3).permute(0, 3, 1, 2).contiguous()
    coord, mask_dsp = masked_resample(seen_map, mask, (H // dsp, W // dsp))
    return seen, coord, mask_dsp, mean, scale


@torch.no_grad()
def scale_and_shift(pred, target, mask):
    """utils/eval_depth.py:11-34: closed-form 2x2 least squares per image ([B,H,W] inputs)."""
    m = mask.float()
    a00, a01, a11 = (m * pred * pred).sum((1, 2)), (m * pred).sum((1, 2)), m.sum((1, 2))
    b0, b1 = (m * pred * target).sum((1, 2)), (m * target).sum((1, 2))
    det = a00 * a11 - a01 * a01
    ok = det > 0
    scale, shift = torch.zeros_like(b0), torch.zeros_like(b1)
    scale[ok] = (a11[ok] * b0[ok] - a01[ok] * b1[ok]) / det[ok]
    shift[ok] = (-a01[ok] * b0[ok] + a00[ok] * b1[ok]) / det[ok]
    return scale, shift


@torch.no_grad()
def depth_metrics(pred, target, mask, thresholds=(1.25, 1.25 ** 2, 1.25 ** 3), depth_cap=None,
                  prediction_type="depth"):
    """utils/eval_depth.py:46-116.  [B,1,H,W] inputs -> (dict name -> [B], aligned depth [B,1,H,W])."""
    p, t = pred.float()[:, 0], target.float()[:, 0]
    v = mask.float()[:, 0] > 0.5
    pd, td = torch.zeros_like(p), torch.zeros_like(t)
    pd[v] = 1.0 / (p[v] + 1.0e-6) if prediction_type == "depth" else p[v]
    td[v] = 1.0 / t[v]
    scale, shift = scale_and_shift(pd, td, v.long())
    aligned = scale.view(-1, 1, 1) * pd + shift.view(-1, 1, 1)
    if depth_cap is not None:
        aligned[aligned < 1.0 / depth_cap] = 1.0 / depth_cap
    d = 1.0 / aligned
    n = v.float().sum((1, 2))

    def masked_mean(values):
        full = torch.zeros_like(d)
        full[v] = values
        return full.sum((1, 2)) / n

    out = {}
    ratio = torch.max(d[v] / t[v], t[v] / d[v])
    for th in thresholds:
        out["d>{}".format(th)] = masked_mean((ratio > th).float())
    out["rmse"] = torch.sqrt(masked_mean((d[v] - t[v]) ** 2))
    out["l1_err"] = masked_mean((d[v] - t[v]).abs())
    out["abs_rel"] = masked_mean((d[v] - t[v]).abs() / t[v])
    return out, d.unsqueeze(1)
