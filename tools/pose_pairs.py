#!/usr/bin/env python3
"""How many distance evaluations the culled pose search really performs ("pairs evaluated per
rotation").  Needs the counting build of csrc/pose_search.hip:

    python tools/build_variant_lib.py pose_count pose_search.hip zeroshape_amd/csrc/pose_search.hip -DZS_POSE_COUNT
    ZS_LIB_PATH=tools/_timing/pose_count.so python tools/pose_pairs.py

Counts (run of 64 queries) x (64 candidates) blocks scanned by pose_nn_soa_kernel over whole searches of the bench
leg's clouds (10k x 10k points, 6912 rotations), against the blocks of the same launches scanned in full."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from zeroshape_amd import _lib, synthetic as syn
    from zeroshape_amd.utils import eval_3D as E
    lib = _lib.load()
    fn = getattr(ctypes.CDLL(lib._name), "zs_pose_debug_counters", None) if hasattr(lib, "_name") else None
    if fn is None:
        raise SystemExit("load the counting build: ZS_LIB_PATH=tools/_timing/pose_count.so (see the docstring)")
    fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int
    dev = torch.device("cuda:0")
    n = 10000
    pred = torch.from_numpy(syn.ellipsoid_cloud(0, n)).to(dev)
    R = E._rotation_sphere(dev)
    g = torch.Generator(device="cpu").manual_seed(0)
    gt = ((R[1234] @ pred.T).T.contiguous().cpu() + 1e-3 * torch.randn(n, 3, generator=g)).to(dev)
    far = torch.from_numpy(syn.seeded_cloud(9, 1, n)[0]).to(dev)
    out = {}
    buf = (ctypes.c_ulonglong * 2)()
    for order in ("str", "morton"):
        os.environ["ZS_POSE_ORDER"] = order
        for name, gt_, prune in (("exhaustive", gt, False), ("pruned", gt, True), ("unalignable", far, True)):
            torch.cuda.synchronize()
            fn(buf, 1)
            o = E.brute_force_search(pred, gt_, device=dev, prune=prune, return_index=True, nn="cull")
            torch.cuda.synchronize()
            fn(buf, 1)
            done, total = int(buf[0]), int(buf[1])
            out["%s/%s" % (order, name)] = {
                "blocks_scanned": done, "blocks_of_the_launched_query_blocks": total,
                "fraction": round(done / max(total, 1), 4),
                "pairs_evaluated": done * 4096, "pairs_evaluated_per_rotation": round(done * 4096 / 6912.0),
                "all_pairs_per_rotation": 2 * n * n, "index": o[5], "rotations_scanned_in_full": E.brute_force_search.last_scanned}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
