import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def decoder_golden():
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "decoder_golden.npz")))


@pytest.fixture(scope="session")
def geometry_golden():
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "geometry_golden.npz")))


@pytest.fixture(scope="session")
def frontend_golden():
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "frontend_golden.npz")))


@pytest.fixture(scope="session")
def seeded_sd(decoder_golden):
    """Seeded decoder weights as torch CPU tensors (pos_embed from the golden file,
    i.e. as the reference initialised it)."""
    import torch
    from zeroshape_amd import synthetic as syn
    sd = syn.seeded_state_dict(seed=0, pos_embed=decoder_golden["pos_embed_f32"])
    return {k: torch.from_numpy(v) for k, v in sd.items()}
