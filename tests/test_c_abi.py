"""The C-ABI library loads on a GPU-less host and exports exactly what include/zeroshape_hip.h
declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "zeroshape_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(zs_\w+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    from zeroshape_amd import build, _lib
    build.build()
    return _lib.load()


def test_header_and_binding_agree(lib):
    from zeroshape_amd import _lib
    names = _declared()
    assert len(names) >= 9
    assert sorted(_lib.SIGNATURES) == names


def test_every_declared_symbol_is_exported(lib):
    from zeroshape_amd import _lib
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert getattr(raw, name) is not None


def test_versions_and_sizes(lib):
    from zeroshape_amd import program as P
    hdr = open(os.path.join(ROOT, "include", "zeroshape_hip.h")).read()
    assert lib.zs_abi_version() == int(re.search(r"#define ZS_ABI_VERSION (\d+)", hdr).group(1))
    assert lib.zs_sdf_program_bytes() == P.PROGRAM_BYTES
    assert lib.zs_sdf_prologue_scratch_bytes() == P.SCRATCH_FLOATS * 4
    assert lib.zs_sdf_workspace_bytes() == 256 * 4 * 3 * 32768 + 4096
    assert lib.zs_sdf_attn_scratch_bytes(2, 129) == 2 * 2 * 4 * (16 * 7 * 4 * 64 * 16 + 16 * 9 * 64 * 4)
    assert lib.zs_sdf_attn_scratch_bytes(0, 5) == 0
    assert lib.zs_last_error() in (b"", None) or isinstance(lib.zs_last_error(), bytes)


def test_argument_errors_are_reported_without_touching_the_gpu(lib):
    # negative sizes / null pointers are rejected before any HIP call: 0 + message
    assert lib.zs_chamfer_forward(None, None, 1, -1, 3, None, None, None, None, None) == 0
    assert b"negative" in lib.zs_last_error()
    assert lib.zs_chamfer_forward(None, None, 1, 4, 3, None, None, None, None, None) == 0
    assert b"null" in lib.zs_last_error()
    assert lib.zs_sdf_query_grid(None, 0, 1, None, 9, 3, 2, 1, None, None, None, None) == 0
    assert b"bad range" in lib.zs_last_error()
    # empty problems succeed trivially (nothing to launch)
    assert lib.zs_chamfer_forward(None, None, 0, 4, 3, None, None, None, None, None) == 1
    assert lib.zs_sdf_query_points(None, 0, 0, None, 5, None, None, None, None, None) == 1


def test_product_path_fails_loudly_without_library(monkeypatch):
    from zeroshape_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libzeroshape_hip.so")
    with pytest.raises(_lib.ZeroShapeHipError):
        _lib.load()


def test_cpu_tensors_are_rejected_not_emulated():
    import torch
    from zeroshape_amd import chamfer_3D
    x = torch.rand(1, 4, 3)
    d = torch.zeros(1, 4)
    i = torch.zeros(1, 4, dtype=torch.int32)
    with pytest.raises(ValueError):
        chamfer_3D.forward(x, x, d, d, i, i)
    from zeroshape_amd.model.shape.implicit import Implicit
    m = Implicit(196, latent_dim=256, n_channels=256, n_blocks_attn=2, n_layers_mlp=8, num_heads=8,
                 skip_in=[2, 4, 6], pos_perlayer=False).eval()
    with pytest.raises(ValueError):
        m(torch.zeros(1, 197, 256), None, torch.zeros(1, 4, 3))
