"""Inline-asm load rings (csrc/nn_gemm_stream.hip, csrc/nn_conv_patch.h): hipcc believes a ring register holds its value as soon as
the asm statement that issued the load has run.  Round 4 lost hours to the consequence - epilogue addresses computed into ring
registers ahead of the final vmcnt wait, overwritten when the last refills landed ("layout dependent" memory faults).
tools/ring_audit.py walks the gfx950 ISA and reports every instruction that touches a register with an asm-issued load in
flight; this test compiles the two translation units (no GPU needed) and requires a clean audit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _audit(src, tmp_path, want):
    import shutil

    import pytest
    import ring_audit as R
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")) and shutil.which("hipcc") is None:
        pytest.skip("no hipcc on this host: the ISA audit needs the compiler")
    res = R.audit_file(R.compile_to_asm(src, str(tmp_path / (src + ".s"))), want, quiet=True)
    assert res, "no kernel matching %r in %s" % (want, src)
    bad = {k: v for k, v in res.items() if v}
    assert not bad, bad
    return res


def test_stream_gemm_ring_registers_stay_allocated_until_their_loads_land(tmp_path):
    res = _audit("nn_gemm_stream.hip", tmp_path, "stream_gemm_kernel")
    assert len(res) == 12                    # 2 row shapes x 3 column shapes x (plain, LayerNorm-on-load)


def test_auditor_sees_the_bug_it_was_written_for(tmp_path):
    """The auditor on a hand-written snippet with the round-4 bug: a ring register reused before the wait."""
    import ring_audit as R
    body = ["\t;;#ASMSTART", "\tglobal_load_dwordx4 v[4:7], v1, s[2:3] offset:0", "\t;;#ASMEND",
            "\tv_lshlrev_b32_e32 v5, 2, v0", "\t;;#ASMSTART", "\ts_waitcnt vmcnt(0)", "\t;;#ASMEND", "\tv_add_u32_e32 v9, v5, v4"]
    assert R.audit("snippet", list(enumerate(body, 1)), quiet=True) == 1
    body[3], body[5] = body[5], body[3]      # the wait first: clean
    assert R.audit("snippet", list(enumerate(body, 1)), quiet=True) == 0
