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
