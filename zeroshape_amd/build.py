"""Ahead-of-time build of libzeroshape_hip.so for gfx950 (hipcc cross-compiles
without a GPU).  The .so is written in-tree (zeroshape_amd/libzeroshape_hip.so):
git-ignored, but it travels to the GPU box with the gpurun snapshot.

    python -m zeroshape_amd.build [--force] [--verbose]
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libzeroshape_hip.so")
OBJDIR = os.path.join(HERE, "csrc", "_obj")

ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall",
          "-Wno-unused-function", "-I", INCLUDE]
# per-file extra flags
EXTRA = {
    # bit-exact distance arithmetic: only the fmaf calls written in the source may fuse
    # (SLP vectorisation packs the distance arithmetic into v_pk_*_f32, which is no faster
    # on gfx950 and costs extra v_mov: off)
    "chamfer.hip": ["-ffp-contract=off", "-fno-slp-vectorize"],
    "chamfer_grid.hip": ["-ffp-contract=off"],
    # the same distance arithmetic as chamfer.hip, and normalize_pc / F-score exactly as written
    "pose_search.hip": ["-ffp-contract=off"],
    # packed f32 VALU ops beside 16-bit MFMAs cost more than the scalar forms they replace
    # ... and MFMA accumulators in VGPRs: the activation code reads and writes them in place
    # (with AGPR accumulators every tile paid 32 v_accvgpr moves; 484 -> 352 registers)
    # ... and no NaN canonicalisation (v_max x, x in front of every fmaxf on an MFMA result;
    # infinities - the softmax mask - keep their meaning)
    "sdf_decoder_split.hip": ["-fno-slp-vectorize", "-fno-honor-nans", "-mllvm", "-amdgpu-mfma-vgpr-form"],
    # vertices / sampled points reproducible op for op by oracle/mc_ref.py
    "marching_cubes.hip": ["-ffp-contract=off"],
}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _includes(path, seen):
    """The quoted includes of a source file that live under csrc/ or include/, transitively."""
    import re
    with open(path, "rb") as fh:
        text = fh.read().decode("utf-8", "replace")
    for name in re.findall(r'^\s*#\s*include\s+["<]([^">]+)[">]', text, flags=re.M):
        # relative to the including file first, then through `-I include` (angle-bracket includes too: a header found there is
        # ours; system headers do not exist under either and are skipped)
        for base in (os.path.dirname(path), INCLUDE):
            f = os.path.normpath(os.path.join(base, name))
            if os.path.exists(f):
                if f not in seen:
                    seen.append(f)
                    _includes(f, seen)
                break
    return seen


def _stamp(src, flags):
    h = hashlib.sha1()
    # (paths relative to the package: the stamp must not change when the tree is copied - a gpurun box sees it elsewhere)
    h.update(" ".join(os.path.relpath(f, HERE) if os.path.isabs(f) else f for f in flags).encode())
    for f in [src] + sorted(_includes(src, [])):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJDIR, exist_ok=True)
    objs, rebuilt = [], False
    for name in sources():
        src = os.path.join(CSRC, name)
        obj = os.path.join(OBJDIR, name[:-4] + ".o")
        flags = COMMON + EXTRA.get(name, [])
        stamp_file = obj + ".stamp"
        stamp = _stamp(src, flags)
        old = open(stamp_file).read() if os.path.exists(stamp_file) else ""
        if force or not os.path.exists(obj) or old != stamp:
            cmd = [hipcc] + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            with open(stamp_file, "w") as fh:
                fh.write(stamp)
            rebuilt = True
        objs.append(obj)
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))
