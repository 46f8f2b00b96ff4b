#!/usr/bin/env python3
"""Audit the decoder kernel's .s for the inline-asm prefetch ring (csrc/sdf_decoder.hip).

The ring lives in v[240:255] + a[240:255], hidden from the compiler (amdgpu_num_vgpr(240) + literal
register names + clobbers).  An in-flight load may land in those registers at any time, so
the invariant is simple: outside ;;#ASMSTART/;;#ASMEND no instruction of the kernel may
mention v240..v255 / a240..a255, the kernel must have no scratch (spill) traffic, and the descriptor
must allocate all 256 accumulator registers.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -S --cuda-device-only \
          zeroshape_amd/csrc/sdf_decoder.hip -o /tmp/sdf.s && python tools/check_asm_ring.py /tmp/sdf.s
"""
import re
import sys

REG = re.compile(r"\b[va](\d+)\b|\b[va]\[(\d+):(\d+)\]")
LO, HI = 240, 255


def regs_in(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check(path):
    lines = open(path).read().split("\n")
    errors, stats = [], {}
    kern, in_asm = None, False
    for no, raw in enumerate(lines, 1):
        l = raw.strip()
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", l)
        if m and "sdf_decode_kernel" in m.group(1):
            kern = m.group(1)
            stats[kern] = dict(loads=0, waits=0, scratch=0, mfma=0)
            continue
        m = re.match(r"^\.amdhsa_(accum_offset|next_free_vgpr)\s+(\d+)", l)
        if m and stats:
            last = list(stats)[-1]
            stats[last][m.group(1)] = int(m.group(2))
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
        code = l.split(";")[0]
        st = stats[kern]
        if "scratch_" in code:
            st["scratch"] += 1
        if "v_mfma" in code:
            st["mfma"] += 1
        if in_asm:
            if "global_load_dwordx4" in code:
                st["loads"] += 1
            if "s_waitcnt" in code:
                st["waits"] += 1
            continue
        bad = [r for r in regs_in(code) if LO <= r <= HI]
        if bad:
            errors.append("%s:%d: compiler instruction '%s' touches reserved ring register(s) %s"
                          % (path, no, code.strip(), sorted(bad)))
    return stats, errors


if __name__ == "__main__":
    stats, errs = check(sys.argv[1])
    ok = bool(stats) and not errs
    for name, st in stats.items():
        n_agpr = st.get("next_free_vgpr", 0) - st.get("accum_offset", 0)
        print("%s: %d MFMAs, %d asm loads, %d asm waits, %d scratch ops, accum_offset %s, agprs %d"
              % (name[:48], st["mfma"], st["loads"], st["waits"], st["scratch"], st.get("accum_offset"), n_agpr))
        ok = ok and n_agpr >= 256      # scratch ops are reported, not fatal (perf only)
    for e in errs[:40]:
        print("ERROR", e)
    print("OK" if ok else "FAILED (%d errors)" % len(errs))
    sys.exit(0 if ok else 1)
