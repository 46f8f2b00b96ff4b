"""CPU checks of the training path's host logic (no GPU, no kernels): the multi-tensor tables of the
fused optimiser / bucket packer, the compat aliases for the reference's entry scripts, the loud
failure of the training ops without a GPU, and that every training symbol of include/zeroshape_hip.h
is exported and typed."""
import ctypes
import sys

import numpy as np
import pytest
import torch


def test_multi_tensor_table_layout():
    from zeroshape_amd import _lib
    from zeroshape_amd.optim import _ENTRY, build_table
    chunk = _lib.load().zs_multi_tensor_chunk_elems()
    entries = [(0x1000, 0x2000, 0x3000, 0x4000, 5, 1e-3, 0.0), (0x5000, 0x6000, 0x7000, 0x8000, 2 * chunk + 7, 3e-5, 0.05)]
    tab, ct, cs, n = build_table(entries, "cpu")
    assert n == 1 + 3 and ct.tolist() == [0, 1, 1, 1] and cs.tolist() == [0, 0, chunk, 2 * chunk]
    rec = np.frombuffer(tab.numpy().tobytes(), _ENTRY)
    assert rec["param"].tolist() == [0x1000, 0x5000] and rec["n"].tolist() == [5, 2 * chunk + 7]
    assert rec["lr"][1] == np.float32(3e-5) and rec["wd"][1] == np.float32(0.05)
    # the C struct the kernels read: 4 pointers, one u64, two floats = 48 bytes

    class Entry(ctypes.Structure):
        _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p),
                    ("n", ctypes.c_ulonglong), ("lr", ctypes.c_float), ("wd", ctypes.c_float)]
    assert ctypes.sizeof(Entry) == _ENTRY.itemsize == 48


def test_training_symbols_are_exported():
    from zeroshape_amd import _lib
    lib = _lib.load()
    for name in ("zs_pack_conv_weight", "zs_pack_conv_weight_multi", "zs_conv2d_wgrad", "zs_conv2d_dgrad_small_cin",
                 "zs_standardize_weight_bwd", "zs_act_backward", "zs_layer_norm_bwd", "zs_attention_bwd",
                 "zs_point_attention", "zs_point_attention_bwd", "zs_bce_logits", "zs_bce_logits_bwd", "zs_adamw_multi",
                 "zs_copy_multi", "zs_sumsq_multi", "zs_batch_norm_train", "zs_batch_norm_bwd", "zs_group_norm_bwd",
                 "zs_max_pool_bwd_nhwc", "zs_global_mean_bwd_nhwc", "zs_upsample2x_bwd_nhwc", "zs_nhwc_to_nchw_masked",
                 "zs_seen_surface_bwd", "zs_intr_param2mtx_bwd", "zs_transform_points", "zs_resize_bilinear_nhwc",
                 "zs_readout_concat_bwd", "zs_add_scaled_rows", "zs_column_sum"):
        assert name in _lib.SIGNATURES and getattr(lib, name) is not None
    assert lib.zs_conv2d_wgrad_workspace_bytes(4, 14, 14, 256, 256, 3, 3) > 0
    assert lib.zs_abi_version() == _lib.ABI_VERSION


def test_training_ops_refuse_cpu_tensors():
    from zeroshape_amd.nn import autograd as A
    from zeroshape_amd.optim import FusedAdamW
    with pytest.raises(ValueError):
        A.layer_norm(torch.zeros(2, 8, requires_grad=True), torch.ones(8), torch.zeros(8))
    with pytest.raises(ValueError):
        A.bce_logits(torch.zeros(2, 3), torch.zeros(2, 3))
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(ValueError):
        FusedAdamW([p]).step()


def test_compat_aliases_cover_what_the_reference_scripts_import():
    import zeroshape_amd.compat as compat
    saved = {k: sys.modules.get(k) for k in compat._ALIASES}
    try:
        names = compat.install()
        import utils.options as options                      # train.py:7
        from utils.util import is_port_in_use                # train.py:8
        import importlib
        engine = importlib.import_module("model.shape_engine")          # train.py:17
        assert callable(options.parse_arguments) and callable(is_port_in_use)
        for method in ("load_dataset", "build_networks", "setup_optimizer", "restore_checkpoint", "setup_visualizer",
                       "train", "evaluate"):                             # train.py:19-25, evaluate.py:15-22
            assert callable(getattr(engine.Runner, method)), method
        depth = importlib.import_module("model.depth_engine")
        assert callable(depth.Runner.evaluate)
        import data.synthetic
        assert callable(data.synthetic.Dataset.id_filename_mapping)       # evaluate.py:17
        assert "external.chamfer3D.dist_chamfer_3D" in names
        assert callable(importlib.import_module("data.pix3d").Dataset.id_filename_mapping)      # configs 3 and 5
        assert callable(importlib.import_module("data.omniobj3d").Dataset.id_filename_mapping)
        with pytest.raises(ModuleNotFoundError):
            importlib.import_module("data.ocrtoc")                         # not mirrored: fails loudly
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
