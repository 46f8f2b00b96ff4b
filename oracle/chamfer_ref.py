"""ctypes front-end of oracle/chamfer_ref.c.  TEST INFRASTRUCTURE ONLY.

Restates external/chamfer3D/dist_chamfer_3D.py:22-70 (allocation + call) on CPU
numpy arrays: returns (dist1, dist2, idx1, idx2) with SQUARED distances and
int32 indices, exactly what chamfer_3DFunction.forward returns.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libzs_oracle.so")
    src = os.path.join(_HERE, "chamfer_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libzs_oracle.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(build())
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int)
        lib.zs_oracle_chamfer_forward.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  fp, fp, ip, ip]
        lib.zs_oracle_chamfer_forward.restype = ctypes.c_int
        lib.zs_oracle_chamfer_backward.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                   fp, fp, fp, fp, ip, ip]
        lib.zs_oracle_chamfer_backward.restype = ctypes.c_int
        _LIB = lib
    return _LIB


def _f(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def chamfer_forward(xyz1, xyz2):
    """xyz1 [B,n,3], xyz2 [B,m,3] float32 -> dist1 [B,n], dist2 [B,m] (squared),
    idx1 [B,n], idx2 [B,m] int32.  Outputs start zeroed like
    dist_chamfer_3D.py:29-33."""
    xyz1 = np.ascontiguousarray(xyz1, dtype=np.float32)
    xyz2 = np.ascontiguousarray(xyz2, dtype=np.float32)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist1 = np.zeros((b, n), np.float32)
    dist2 = np.zeros((b, m), np.float32)
    idx1 = np.zeros((b, n), np.int32)
    idx2 = np.zeros((b, m), np.int32)
    rc = _lib().zs_oracle_chamfer_forward(_f(xyz1), _f(xyz2), b, n, m,
                                          _f(dist1), _f(dist2), _i(idx1), _i(idx2))
    assert rc == 1
    return dist1, dist2, idx1, idx2


def chamfer_backward(xyz1, xyz2, graddist1, graddist2, idx1, idx2):
    """dist_chamfer_3D.py:44-60: returns (gradxyz1, gradxyz2)."""
    xyz1 = np.ascontiguousarray(xyz1, dtype=np.float32)
    xyz2 = np.ascontiguousarray(xyz2, dtype=np.float32)
    graddist1 = np.ascontiguousarray(graddist1, dtype=np.float32)
    graddist2 = np.ascontiguousarray(graddist2, dtype=np.float32)
    idx1 = np.ascontiguousarray(idx1, dtype=np.int32)
    idx2 = np.ascontiguousarray(idx2, dtype=np.int32)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = np.zeros_like(xyz1)
    g2 = np.zeros_like(xyz2)
    rc = _lib().zs_oracle_chamfer_backward(_f(xyz1), _f(xyz2), b, n, m, _f(g1), _f(g2),
                                           _f(graddist1), _f(graddist2), _i(idx1), _i(idx2))
    assert rc == 1
    return g1, g2
