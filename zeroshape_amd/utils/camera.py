"""Mirror of the reference's utils/camera.py:150-230 rotation helpers used by the
brute-force Chamfer search.  Host-side, fp64 trig then fp32 like the reference;
built once and cached (the reference rebuilds all 6912 matrices with 27k tiny
torch ops on every brute_force_search call, utils/eval_3D.py:148)."""
import functools

import numpy as np
import torch


def _angles(n):
    return np.linspace(0, 360, num=n, endpoint=False) * np.pi / 180


@functools.lru_cache(maxsize=8)
def _rotation_sphere_cpu(azim_sample, elev_sample, roll_sample, scales):
    """R = scale * Rz(roll) @ Rx(elev) @ Ry(azim) @ R_permute, azim outermost and roll
    innermost (utils/camera.py:208-230).  Each factor is built in fp64 and rounded to
    fp32 (torch.eye(3) is fp32 in the reference, :162,180,200), the products are fp32
    matmuls in the reference's association order."""
    def ry(a):
        c, s = np.cos(a), np.sin(a)
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]).astype(np.float32)

    def rx(a):
        c, s = np.cos(a), np.sin(a)
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]]).astype(np.float32)

    def rz(a):
        c, s = np.cos(a), np.sin(a)
        return np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]]).astype(np.float32)

    P = torch.tensor([[-1, 0, 0], [0, 0, -1], [0, -1, 0]]).float()
    Ry = torch.from_numpy(np.stack([ry(a) for a in _angles(azim_sample)]))
    Rx = torch.from_numpy(np.stack([rx(a) for a in _angles(elev_sample)]))
    Rz = torch.from_numpy(np.stack([rz(a) for a in _angles(roll_sample)]))
    # batched matmul with broadcasting == the reference's [1,3,3] @ [1,3,3] products
    # bit for bit (the 2-D mm path rounds differently); verified against the golden.
    out = []
    for scale in scales:
        R = (scale * Rz)[None, None] @ Rx[None, :, None] @ Ry[:, None, None] @ P
        out.append(R.reshape(-1, 3, 3).float())
    return torch.cat(out, dim=0)


def get_rotation_sphere(azim_sample=4, elev_sample=4, roll_sample=4, scales=[1.0], device='cuda'):
    """utils/camera.py:208-230.  Returns [len(scales)*A*E*R, 3, 3] fp32 on ``device``."""
    R = _rotation_sphere_cpu(int(azim_sample), int(elev_sample), int(roll_sample),
                             tuple(float(s) for s in scales))
    return R.to(device)


# --------------------------------------------------------------------------------------
# Seen-surface geometry (utils/camera.py:52-108 of the reference), on the HIP library.
# --------------------------------------------------------------------------------------
def _dev_f32(t, what):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise ValueError("%s must be a GPU tensor (there is no CPU path)" % what)
    return t.detach().to(torch.float32).contiguous()


def get_pixel_grid(opt, H, W):
    """utils/camera.py:80-86: [H*W, 3] rows (x, y, 1), x fastest."""
    y_range = torch.arange(H, dtype=torch.float32, device=opt.device)
    x_range = torch.arange(W, dtype=torch.float32, device=opt.device)
    Y, X = torch.meshgrid(y_range, x_range, indexing="ij")
    return torch.stack([X, Y, torch.ones_like(Y)], dim=-1).view(-1, 3)


def unproj_depth(opt, depth, intr):
    """utils/camera.py:88-108: depth [B,1,H,W], intr [B,3,3] -> seen points [B,H*W,3] in the
    camera frame (K^-1 [x,y,1]^T * depth).  One launch of zs_unproj_depth."""
    from .. import _lib
    lib = _lib.load()
    batch_size, _, H, W = depth.shape
    assert opt.H == H == W
    d, K = _dev_f32(depth, "depth"), _dev_f32(intr, "intr")
    assert K.shape == (batch_size, 3, 3)
    out = torch.empty(batch_size, H * W, 3, dtype=torch.float32, device=d.device)
    with torch.cuda.device(d.device):
        _lib.check(lib.zs_unproj_depth(_lib.ptr(d), _lib.ptr(K), batch_size, H, W, _lib.ptr(out),
                                       _lib.current_stream_ptr(d.device)), "zs_unproj_depth")
    return out


def valid_norm_fac(seen_points, mask):
    """utils/camera.py:52-78: seen_points [B,H*W,3], mask [B,1,H,W] boolean -> (mean [B,3],
    max_dist [B]) over the selected points of each sample, without the reference's Python loop
    and boolean gathers.  A sample with no valid pixel gives NaN (the reference raises)."""
    from .. import _lib
    lib = _lib.load()
    batch_size, n = seen_points.shape[0], seen_points.shape[1]
    p = _dev_f32(seen_points, "seen_points")
    m = mask.reshape(batch_size, n)
    m = (m if m.dtype == torch.bool else m > 0.5).to(torch.uint8).contiguous()
    mean = torch.empty(batch_size, 3, dtype=torch.float32, device=p.device)
    dist = torch.empty(batch_size, dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        _lib.check(lib.zs_valid_norm_fac(_lib.ptr(p), _lib.ptr(m), batch_size, n, _lib.ptr(mean), _lib.ptr(dist),
                                         _lib.current_stream_ptr(p.device)), "zs_valid_norm_fac")
    return mean, dist


def intr_param2mtx(opt, intr_params):
    """Graph.intr_param2mtx, model/compute_graph/graph_shape.py:89-113: [B,3] raw parameters
    (scale_f, delta_cx, delta_cy) -> [B,3,3] intrinsics."""
    from .. import _lib
    lib = _lib.load()
    p = _dev_f32(intr_params, "intr_params")
    assert p.dim() == 2 and p.shape[1] == 3
    out = torch.empty(p.shape[0], 3, 3, dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        _lib.check(lib.zs_intr_param2mtx(_lib.ptr(p), p.shape[0], int(opt.H), int(opt.W), _lib.ptr(out),
                                         _lib.current_stream_ptr(p.device)), "zs_intr_param2mtx")
    return out


def seen_surface(opt, depth_map, intr, mask_input_map, dsp=None):
    """The seen-surface block of Graph.forward (graph_shape.py:131-144) as ONE launch
    (zs_seen_surface): unproject, per-sample masked mean / max-radius, normalise, zero the
    invalid pixels, and resample to (H//dsp, W//dsp) with interpolate_coordmap.

    Returns (seen_points [B,H*W,3], seen_3D_dsp [B,3,H//dsp,W//dsp], mask_dsp [B,1,H//dsp,W//dsp],
    mean [B,3], scale [B])."""
    from .. import _lib
    lib = _lib.load()
    batch_size, _, H, W = depth_map.shape
    assert opt.H == H == W
    if dsp is None:
        dsp = opt.arch.depth.dsp
    Ho, Wo = H // dsp, W // dsp
    d, K, m = _dev_f32(depth_map, "depth_map"), _dev_f32(intr, "intr"), _dev_f32(mask_input_map, "mask_input_map")
    assert m.shape == d.shape and K.shape == (batch_size, 3, 3)
    dev = d.device
    seen = torch.empty(batch_size, H * W, 3, dtype=torch.float32, device=dev)
    mean = torch.empty(batch_size, 3, dtype=torch.float32, device=dev)
    scale = torch.empty(batch_size, dtype=torch.float32, device=dev)
    coord = torch.empty(batch_size, 3, Ho, Wo, dtype=torch.float32, device=dev)
    mask_dsp = torch.empty(batch_size, 1, Ho, Wo, dtype=torch.float32, device=dev)
    ws = torch.empty(lib.zs_seen_surface_workspace_bytes(batch_size) // 4, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.zs_seen_surface_ws(_lib.ptr(d), _lib.ptr(K), _lib.ptr(m), batch_size, H, W, Ho, Wo,
                                          _lib.ptr(seen), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(coord),
                                          _lib.ptr(mask_dsp), _lib.ptr(ws), _lib.current_stream_ptr(dev)), "zs_seen_surface")
    return seen, coord, mask_dsp, mean, scale


def transform_points(points, pose, mean, scale):
    """graph_shape.py:163-173: points [B,N,3] in the object frame -> camera frame by pose [B,3,4]
    -> normalised by the seen surface's (mean [B,3], scale [B]): ((R p + t) - mean) / scale."""
    from .. import _lib
    lib = _lib.load()
    p, T = _dev_f32(points, "points"), _dev_f32(pose, "pose")
    m, s = _dev_f32(mean, "mean"), _dev_f32(scale, "scale")
    B, N, _ = p.shape
    assert T.shape == (B, 3, 4)
    out = torch.empty_like(p)
    with torch.cuda.device(p.device):
        _lib.check(lib.zs_transform_points(_lib.ptr(p), _lib.ptr(T), _lib.ptr(m), _lib.ptr(s), _lib.ptr(out), B, N,
                                           _lib.current_stream_ptr(p.device)), "zs_transform_points")
    return out
