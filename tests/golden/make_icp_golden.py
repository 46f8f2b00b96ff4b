#!/usr/bin/env python3
"""Golden fixtures for standardize_pc / ICP from the REAL reference (build container only).

Run:  python tests/golden/make_icp_golden.py          (needs /root/reference)

Imports the reference's utils/eval_3D.py (:83-91 standardize_pc, :271-284 ICP) with the module stubs of
tests/golden/make_golden.py; the Chamfer plugin it calls (a CUDA extension) is replaced IN THIS PROCESS ONLY by the
stand-in SURVEY.md section 8-c describes: torch.cdist in float64 + min / argmin, returning squared distances and int32
indices like chamfer_3DDist.  Inputs are the seeded clouds of zeroshape_amd/synthetic.py (well separated points: no
near ties between the float64 stand-in and an fp32 kernel).  Only arrays leave this script."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as M          # noqa: E402  (the stubs)

REF = "/root/reference"


class _Chamfer(torch.nn.Module):
    def forward(self, a, b):
        d = torch.cdist(a.double(), b.double()) ** 2
        d1, i1 = d.min(2)
        d2, i2 = d.min(1)
        return d1.float(), d2.float(), i1.int(), i2.int()


def clouds():
    from zeroshape_amd import synthetic as syn
    return syn.icp_clouds()


def main():
    M._install_stubs()
    sys.modules["external.chamfer3D.dist_chamfer_3D"].chamfer_3DDist = _Chamfer
    sys.path.insert(0, REF)
    import importlib
    E = importlib.import_module("utils.eval_3D")
    a, b = clouds()
    A, B = torch.from_numpy(a), torch.from_numpy(b)
    out = {"standardize_pc_out": E.standardize_pc(A * torch.tensor([2.0, 1.0, 3.0]) + 0.3).numpy()}
    for it in (1, 3, 50):
        out["icp_%d" % it] = E.ICP(None, A.clone(), B.clone(), num_iter=it).numpy()
    np.savez_compressed(os.path.join(HERE, "icp_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
