#!/usr/bin/env python3
"""Audit the ISA of the split-fp16 decoder (csrc/sdf_decoder_split.hip).

The kernel's LDS-DMA statements write M0 without saving it and count their own vmcnt, so:
  * M0 may only be touched inside ;;#ASMSTART/;;#ASMEND (hipcc must have no use of its own);
  * no scratch traffic / no private segment (a spill would also upset the counted waits);
  * every MFMA is a v_mfma_f32_32x32x16_f16 accumulating in place (the VGPR form where the
    registers allow, -mllvm -amdgpu-mfma-vgpr-form); the dynamic count (14,784 per wave tile) is
    in the instruction counters of profiles/;
  * each decode kernel carries LDS-DMAs and raw barriers, and allocates all 160 KiB of LDS.

    python tools/check_split_isa.py            (compiles with the flags of zeroshape_amd/build.py)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def compile_to_asm():
    from zeroshape_amd import build as B
    name = "sdf_decoder_split.hip"
    out = os.path.join(tempfile.mkdtemp(), "split.s")
    flags = [f for f in B.COMMON if f != "-fPIC"] + B.EXTRA.get(name, [])
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags +
                          ["-S", "--cuda-device-only", os.path.join(B.CSRC, name), "-o", out],
                          stderr=subprocess.DEVNULL)
    return out


def check(path):
    errors, stats = [], {}
    kern, in_asm = None, False
    for no, raw in enumerate(open(path).read().split("\n"), 1):
        l = raw.strip()
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", l)
        if m and "sdf_decode_split_kernel" in m.group(1):
            kern = m.group(1)
            stats[kern] = dict(mfma=0, dma=0, barrier=0, lds=None, private=None)
            continue
        m = re.match(r"^\.amdhsa_(group_segment_fixed_size|private_segment_fixed_size)\s+(\d+)", l)
        if m and stats:
            stats[list(stats)[-1]]["lds" if m.group(1).startswith("group") else "private"] = int(m.group(2))
        if kern is None:
            continue
        if l.startswith(".Lfunc_end"):
            kern = None
            continue
        if l.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if l.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not l or l.startswith(";") or l.startswith("."):
            continue
        st = stats[kern]
        if re.search(r"\bm0\b", l) and not in_asm:
            errors.append("%d: compiler instruction touches m0: %s" % (no, l))
        if l.startswith("scratch_") or "buffer_store" in l and "offen" in l:
            errors.append("%d: scratch traffic: %s" % (no, l))
        if l.startswith("v_mfma"):
            st["mfma"] += 1
            if not re.match(r"v_mfma_f32_32x32x16_f16 ([va])\[\d+:\d+\], [va]\[\d+:\d+\], [va]\[\d+:\d+\], (\1\[\d+:\d+\]|0)", l):
                errors.append("%d: unexpected MFMA form: %s" % (no, l))
        if l.startswith("global_load_lds_dwordx4"):
            st["dma"] += 1
            if not in_asm:
                errors.append("%d: LDS-DMA outside the asm statements" % no)
        if l.startswith("s_barrier"):
            st["barrier"] += 1
    if len(stats) != 2:
        errors.append("expected the <GRID> and the point-list kernels, found %s" % list(stats))
    for k, st in stats.items():
        if st["lds"] != 160 * 1024:
            errors.append("%s: LDS %s, expected 163840" % (k, st["lds"]))
        if st["private"] != 0:
            errors.append("%s: private segment %s" % (k, st["private"]))
        if not (st["mfma"] and st["dma"] and st["barrier"]):
            errors.append("%s: %s" % (k, st))
    return errors, stats


if __name__ == "__main__":
    errs, stats = check(sys.argv[1] if len(sys.argv) > 1 else compile_to_asm())
    for k, st in stats.items():
        print(k[:60], st)
    for e in errs:
        print("ERROR", e)
    sys.exit(1 if errs else 0)
