import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# the tests run the engines on this repository's analytic stand-in data on purpose (zeroshape_amd/data/__init__.py:
# the opt-in the engines ask for; the fence itself is tested with the variable removed)
os.environ.setdefault("ZS_SYNTHETIC_STANDIN", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle (torch fp32) stops scaling at ~32 intra-op threads; a GPU box's default is every one of its 256 cores, which
    # ran the vox-128 oracle grid at 11 k points/s instead of 49 k (187 s of the suite's 1,009 s in round 5)
    try:
        import torch
        torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    except Exception:
        pass


@pytest.fixture(scope="session")
def decoder_golden():
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "decoder_golden.npz")))


@pytest.fixture(scope="session")
def grid_golden():
    """Full 65^3 / 129^3 grids of the reference's own compute_level_grid (tests/golden/make_grid_golden.py)."""
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "grid_golden.npz")))


@pytest.fixture(scope="session")
def grid256_golden():
    """The full 257^3 grid (BASELINE config 5) of the reference's own compute_level_grid (make_grid_golden.py 256)."""
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "grid256_golden.npz")))


@pytest.fixture(scope="session")
def grid_gain60_golden():
    """The full 129^3 grid of the reference for the seeded network at a converged checkpoint's logit scale (last three MLP
    layers x 60^(1/3): |logit| up to 36) on another image (make_grid_golden.py 128 gain=60 latent=1)."""
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "grid128_gain60_golden.npz")))


@pytest.fixture(scope="session")
def eval_golden():
    """The reference's own eval_metrics_default / eval_metrics_BF / brute_force_search on seeded clouds (make_eval_golden.py)."""
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "eval_golden.npz")))


@pytest.fixture(scope="session")
def posenc_golden():
    """The reference's Implicit(posenc_3D=4) on the seeded weights (tests/golden/make_posenc_golden.py)."""
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "posenc_golden.npz")))


@pytest.fixture(scope="session")
def variants_golden():
    """The reference's Implicit in other constructor configurations (tests/golden/make_variants_golden.py)."""
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "variants_golden.npz")))


@pytest.fixture(scope="session")
def geometry_golden():
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "geometry_golden.npz")))


@pytest.fixture(scope="session")
def frontend_golden():
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "frontend_golden.npz")))


@pytest.fixture(scope="session")
def encoder_golden():
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "encoder_golden.npz")))


def _shapes(keys, shapes):
    return {str(k): tuple(int(x) for x in str(s).split(",") if x) for k, s in zip(keys, shapes)}


@pytest.fixture(scope="session")
def encoder_sd(encoder_golden):
    """Seeded encoder parameters under the reference Graph's state-dict names (torch CPU), with
    the head calibration of tests/golden/make_encoder_golden.py applied."""
    import torch
    from zeroshape_amd import synthetic as syn
    shapes = _shapes(encoder_golden["graph_keys"], encoder_golden["graph_shapes"])
    enc = {k: v for k, v in shapes.items() if not k.startswith("impl_network.")}
    sd = {k: torch.from_numpy(v) for k, v in syn.seeded_encoder_state_dict(enc, seed=0).items()}
    gain, offset = encoder_golden["head_calibration"]
    sd["dpt_depth.scratch.output_conv.4.weight"] = sd["dpt_depth.scratch.output_conv.4.weight"] * float(gain)
    sd["dpt_depth.scratch.output_conv.4.bias"] = torch.full_like(sd["dpt_depth.scratch.output_conv.4.bias"],
                                                                 float(offset))
    return sd


@pytest.fixture(scope="session")
def att_sd(encoder_golden):
    import torch
    from zeroshape_amd import synthetic as syn
    shapes = _shapes(encoder_golden["att_keys"], encoder_golden["att_shapes"])
    return {k: torch.from_numpy(v) for k, v in syn.seeded_encoder_state_dict(shapes, seed=1).items()}


@pytest.fixture(scope="session")
def seeded_sd(decoder_golden):
    """Seeded decoder weights as torch CPU tensors (pos_embed from the golden file,
    i.e. as the reference initialised it)."""
    import torch
    from zeroshape_amd import synthetic as syn
    sd = syn.seeded_state_dict(seed=0, pos_embed=decoder_golden["pos_embed_f32"])
    return {k: torch.from_numpy(v) for k, v in sd.items()}


@pytest.fixture(scope="session")
def decoder_train_golden():
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "decoder_train_golden.npz")))


@pytest.fixture(scope="session")
def graph_train_golden():
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "graph_train_golden.npz")))
