#!/usr/bin/env python3
"""Side build of libzeroshape_hip.so with the split decoder's cycle stamps compiled in
(-DZS_EXP_TIMING): tools/_timing/libzs_timing.so, for tools/phase_timing_split.py
(ZS_LIB_PATH=tools/_timing/libzs_timing.so python tools/phase_timing_split.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zeroshape_amd import build as B   # noqa: E402

B.build()
out_dir = os.path.join(ROOT, "tools", "_timing")
os.makedirs(out_dir, exist_ok=True)
name = "sdf_decoder_split.hip"
obj = os.path.join(out_dir, "split_timing.o")
subprocess.check_call([B.os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + B.COMMON + B.EXTRA.get(name, []) +
                      ["-DZS_EXP_TIMING", "-c", os.path.join(B.CSRC, name), "-o", obj], stderr=subprocess.DEVNULL)
objs = [obj if n == name else os.path.join(B.OBJDIR, n[:-4] + ".o") for n in B.sources()]
lib = os.path.join(out_dir, "libzs_timing.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", lib] + objs)
print(lib)
