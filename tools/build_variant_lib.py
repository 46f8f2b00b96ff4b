#!/usr/bin/env python3
"""Side build of the library with ANOTHER version of one source file (A/B measurements on the same GPU
box: `ZS_LIB_PATH=tools/_timing/<name>.so python bench.py ...`).

    python tools/build_variant_lib.py <name> <file.hip in csrc> <replacement source | git rev> [-Dflag ...]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zeroshape_amd import build as B   # noqa: E402


def main():
    name, fname, src = sys.argv[1:4]
    extra = sys.argv[4:]
    B.build()
    out_dir = os.path.join(ROOT, "tools", "_timing")
    os.makedirs(out_dir, exist_ok=True)
    if not os.path.exists(src):        # a git revision
        text = subprocess.check_output(["git", "show", "%s:zeroshape_amd/csrc/%s" % (src, fname)], cwd=ROOT)
        src = os.path.join(B.CSRC, "_variant_" + fname)
        open(src, "wb").write(text)
    obj = os.path.join(out_dir, name + ".o")
    try:
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + B.COMMON + B.EXTRA.get(fname, []) + extra +
                              ["-c", src, "-o", obj], stderr=subprocess.DEVNULL)
    finally:
        if os.path.basename(src).startswith("_variant_"):
            os.remove(src)
    objs = [obj if n == fname else os.path.join(B.OBJDIR, n[:-4] + ".o") for n in B.sources()]
    lib = os.path.join(out_dir, name + ".so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", lib] + objs)
    print(lib)


if __name__ == "__main__":
    main()
