"""CPU oracle for the ZeroShape dense SDF-query / Chamfer hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import, call, link or execute it, and there only as the
checker - never as the thing measured or shipped.  The product path
(``zeroshape_amd``) fails loudly when the HIP library is missing; it never
falls back to this code.

Each function cites the reference file:line (relative to the upstream
ZeroShape tree) whose arithmetic it restates.  Parity pinning: the reference
has no tests or golden vectors of its own, so the oracle
is pinned against outputs of the reference itself, generated in the build
container by ``tests/golden/make_golden.py`` (which imports the reference's
Python from /root/reference) and committed as small fixtures under
``tests/golden/``.  The Chamfer CUDA kernel cannot run anywhere in this
environment; its restatement is pinned by known-answer tests and an fp64
cross-check instead ("parity unpinned upstream").
"""
