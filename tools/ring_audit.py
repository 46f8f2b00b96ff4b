"""Static audit of an inline-asm register ring in gfx950 ISA text (hipcc -S --cuda-device-only).

The streaming kernels issue `global_load_dwordx4` from inline asm and wait with a counted `s_waitcnt vmcnt(n)`.  The compiler
believes the destination registers hold their value as soon as the asm statement has "executed", so it is free to COPY a ring
register (v_mov / v_accvgpr_write / v_pk_mov) or to reuse it for another value while the load is still in flight.  Either one is
a silent bug: the copy reads stale data, and the late-landing load overwrites whatever now lives in the register.

For every kernel this script walks the instruction stream in program order and keeps, per VGPR, the number of vector-memory
loads issued since the load that targets it.  A register is "in flight" until a s_waitcnt vmcnt(n) with n < (loads issued
after it) retires it.  It reports every instruction that reads or writes an in-flight register.  Only loads issued from inline
asm (between ;#ASMSTART / ;#ASMEND) are tracked - the compiler waits for its own loads.  Loops are handled by walking
each kernel's text twice (state carried over the back edge in layout order), which is exact for the single-loop kernels here.

usage: python tools/ring_audit.py file.s [kernel-name-substring]
"""
import re
import sys

REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs_of(tok):
    out = []
    for m in REG.finditer(tok):
        if m.group(1):
            out.append((m.group(1), int(m.group(2))))
        else:
            out += [(m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1)]
    return out


def audit(name, lines, quiet=False):
    inflight = {}          # reg -> serial number of its load
    issued = 0             # vector memory loads issued so far (vmcnt counts loads and stores returning data... loads only here)
    findings = []
    for rnd in range(2):
        in_asm = False
        for ln, text in lines:
            if ";#ASMSTART" in text:
                in_asm = True
            elif ";#ASMEND" in text:
                in_asm = False
            ins = text.split(";")[0].strip()
            if not ins or ins.endswith(":") or ins.startswith("."):
                continue
            op = ins.split()[0]
            args = ins[len(op):]
            if op.startswith("s_waitcnt"):
                m = re.search(r"vmcnt\((\d+)\)", ins)
                if m:
                    n = int(m.group(1))
                    for r in [r for r, s in inflight.items() if issued - s >= n]:
                        del inflight[r]
                continue
            parts = [p.strip() for p in args.split(",")]
            is_vload = op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load"))
            is_vstore = op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic"))
            touched = regs_of(args)
            dst = regs_of(parts[0]) if parts and not is_vstore else []
            if is_vload and "lds" not in ins:
                hit = [r for r in touched if r in inflight and r not in dst]     # address registers in flight
                rew = [r for r in dst if r in inflight]                          # reloaded before the wait: legal (in-order return)
                if hit and rnd == 1:
                    findings.append((ln, ins, hit))
                issued += 1
                for r in dst:
                    if in_asm:            # the compiler waits for its own loads correctly; only asm-issued ones are audited
                        inflight[r] = issued
                    else:
                        inflight.pop(r, None)
                continue
            if is_vstore or is_vload:
                issued += 1          # stores and LDS-DMA loads (no register destination) count in vmcnt too
            hit = [r for r in touched if r in inflight]
            if hit and rnd == 1:
                findings.append((ln, ins, hit))
    if not quiet:
        print("%-90s %s" % (name[:90], "CLEAN" if not findings else "%d finding(s)" % len(findings)))
        for ln, ins, hit in findings[:12]:
            print("    line %d: %s   <- in flight: %s" % (ln, ins, " ".join("%s%d" % r for r in hit[:6])))
    return len(findings)


def compile_to_asm(src, out, extra=()):
    """hipcc -S of one translation unit of zeroshape_amd/csrc for gfx950 (no GPU needed)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from zeroshape_amd import build as B          # the flags the shipped object is built with: the audited ISA is the shipped one
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + B.COMMON + B.EXTRA.get(src, []) + ["--cuda-device-only", "-S",
           os.path.join(root, "zeroshape_amd", "csrc", src), "-o", out] + list(extra)
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def audit_file(path, want="", quiet=False):
    """-> {kernel symbol: number of findings} for the kernels of one ISA text whose name contains `want`."""
    cur, body, res = None, [], {}
    for i, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            cur, body = m.group(1), []
            continue
        if cur and line.startswith(".Lfunc_end"):
            if want in cur:
                res[cur] = audit(cur, body, quiet)
            cur = None
            continue
        if cur:
            body.append((i, line))
    return res


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    sys.exit(1 if sum(audit_file(path, want).values()) else 0)


if __name__ == "__main__":
    main()
