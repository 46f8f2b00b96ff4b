"""Drop-in for the reference's native plugin module ``chamfer_3D``
(external/chamfer3D/chamfer_cuda.cpp:30-33): the same two functions with the same
tensor signatures, ownership and return convention, backed by the hand-written HIP
kernels in csrc/chamfer.hip through the C ABI (include/zeroshape_hip.h).

    forward(xyz1, xyz2, dist1, dist2, idx1, idx2) -> int
    backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2) -> int

All tensors are allocated by the caller; outputs are written in place; 1 = success.
Unlike the reference (which reads ``.data<float>()`` with no checks,
chamfer3D.cu:142), wrong dtype / device / layout raise instead of corrupting memory,
and the kernels run on torch's CURRENT stream of the tensors' device rather than
the legacy default stream.
"""
import os

import torch

from . import _lib

# clouds at least this large take the grid-accelerated kernel (bit-identical results);
# ZS_CHAMFER_BRUTE=1 forces the brute-force scan (cross-checks, experiments)
GRID_MIN_POINTS = 1024
_WS = {}


def _workspace(device, nbytes):
    key = str(device)
    if key not in _WS or _WS[key].numel() * 4 < nbytes:
        _WS[key] = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
    return _WS[key]


def _chk(t, name, dtype, shape=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise ValueError("%s must live on the GPU (got %s); there is no CPU path" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s (got %s)" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))


def forward(xyz1, xyz2, dist1, dist2, idx1, idx2, method="auto"):
    """method: "auto" (grid-accelerated kernel from GRID_MIN_POINTS points, else brute force),
    "grid" or "brute" - all three give bit-identical outputs.  The grid kernel wins when the
    clouds overlap (nearest neighbours within a few cells); for far-apart clouds (most
    rotations of a brute-force pose search) the plain scan is faster."""
    _chk(xyz1, "xyz1", torch.float32)
    _chk(xyz2, "xyz2", torch.float32)
    if xyz1.dim() != 3 or xyz2.dim() != 3 or xyz1.shape[2] != 3 or xyz2.shape[2] != 3 \
            or xyz1.shape[0] != xyz2.shape[0]:
        raise ValueError("xyz1/xyz2 must be [B,n,3] / [B,m,3]")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    _chk(dist1, "dist1", torch.float32, (b, n))
    _chk(dist2, "dist2", torch.float32, (b, m))
    _chk(idx1, "idx1", torch.int32, (b, n))
    _chk(idx2, "idx2", torch.int32, (b, m))
    lib = _lib.load()
    if method not in ("auto", "grid", "brute"):
        raise ValueError("method must be auto, grid or brute")
    use_grid = min(n, m) > 0 and (method == "grid" or (
        method == "auto" and max(n, m) >= GRID_MIN_POINTS and not os.environ.get("ZS_CHAMFER_BRUTE")))
    with torch.cuda.device(xyz1.device):
        if use_grid:
            nbytes = lib.zs_chamfer_workspace_bytes(b, n, m)
            ws = _workspace(xyz1.device, nbytes)
            rc = lib.zs_chamfer_forward_ws(_lib.ptr(xyz1), _lib.ptr(xyz2), b, n, m, _lib.ptr(dist1),
                                           _lib.ptr(dist2), _lib.ptr(idx1), _lib.ptr(idx2), _lib.ptr(ws),
                                           nbytes, _lib.current_stream_ptr(xyz1.device))
        else:
            rc = lib.zs_chamfer_forward(_lib.ptr(xyz1), _lib.ptr(xyz2), b, n, m, _lib.ptr(dist1),
                                        _lib.ptr(dist2), _lib.ptr(idx1), _lib.ptr(idx2),
                                        _lib.current_stream_ptr(xyz1.device))
    _lib.check(rc, "zs_chamfer_forward")
    return rc


def backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
    _chk(xyz1, "xyz1", torch.float32)
    _chk(xyz2, "xyz2", torch.float32)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    _chk(gradxyz1, "gradxyz1", torch.float32, (b, n, 3))
    _chk(gradxyz2, "gradxyz2", torch.float32, (b, m, 3))
    _chk(graddist1, "graddist1", torch.float32, (b, n))
    _chk(graddist2, "graddist2", torch.float32, (b, m))
    _chk(idx1, "idx1", torch.int32, (b, n))
    _chk(idx2, "idx2", torch.int32, (b, m))
    lib = _lib.load()
    with torch.cuda.device(xyz1.device):
        rc = lib.zs_chamfer_backward(_lib.ptr(xyz1), _lib.ptr(xyz2), b, n, m, _lib.ptr(gradxyz1),
                                     _lib.ptr(gradxyz2), _lib.ptr(graddist1), _lib.ptr(graddist2),
                                     _lib.ptr(idx1), _lib.ptr(idx2),
                                     _lib.current_stream_ptr(xyz1.device))
    _lib.check(rc, "zs_chamfer_backward")
    return rc
