"""The split-fp16 decoder's inline asm relies on facts about hipcc's output (M0 untouched by the
compiler, no scratch, VGPR-form MFMAs, raw barriers + LDS-DMA present, all of LDS allocated):
tools/check_split_isa.py compiles the kernel for gfx950 (no GPU needed) and audits the ISA."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_split_decoder_isa_invariants():
    import check_split_isa as C
    errs, stats = C.check(C.compile_to_asm())
    assert not errs, errs
    assert len(stats) == 2
