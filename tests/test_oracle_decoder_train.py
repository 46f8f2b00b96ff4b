"""Oracle training restatement (oracle/decoder_ref.py::implicit_forward_train, shape_loss) vs the
golden outputs of the REAL reference in train mode (tests/golden/decoder_train_golden.npz, made by
tests/golden/make_train_golden.py: Implicit.train() + Loss.shape_loss + backward)."""
import numpy as np
import torch

from oracle import decoder_ref as R
from zeroshape_amd import synthetic as syn


def run_oracle(seeded_sd, g):
    sd = {k: v.clone().requires_grad_(k != "pos_embed") for k, v in seeded_sd.items()}
    latent = torch.from_numpy(syn.seeded_latent(int(g["latent_seed"]), g["points"].shape[0])).requires_grad_(True)
    scales = [torch.from_numpy(s) for s in g["drop_scales"]]
    logits = R.implicit_forward_train(sd, latent, torch.from_numpy(g["points"]), scales)
    loss = R.shape_loss(logits, torch.from_numpy(g["sdf"]), float(g["impt_thres"]), float(g["impt_weight"]))
    loss.backward()
    return sd, latent, logits, loss


def check_grads(named_grads, g, rtol, atol_scale=1.0):
    """named_grads: {name: CPU tensor}; compares norms, full small tensors and strided samples."""
    n = 0
    for key in g:
        if not key.startswith("gnorm/"):
            continue
        name = key[6:]
        got = named_grads[name].double()
        want_norm = float(g[key])
        assert abs(float(got.norm()) - want_norm) <= rtol * want_norm + 1e-12, name
        tol = max(want_norm, 1e-12) * rtol * atol_scale
        if "g/" + name in g:
            np.testing.assert_allclose(got.float().numpy(), g["g/" + name], atol=tol, rtol=0, err_msg=name)
        else:
            np.testing.assert_allclose(got.float().reshape(-1)[::97].numpy(), g["gs/" + name], atol=tol, rtol=0,
                                       err_msg=name)
        n += 1
    assert n == 48       # every trainable tensor of impl_network (pos_embed is frozen)


def test_train_forward_backward_matches_reference(seeded_sd, decoder_train_golden):
    g = decoder_train_golden
    assert (g["drop_scales"] == 0).any() and (g["drop_scales"] > 1).any()      # DropPath really dropped a sample
    sd, latent, logits, loss = run_oracle(seeded_sd, g)
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], atol=5e-6, rtol=0)
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    assert abs(float(latent.grad.double().norm()) - float(g["grad_latent_norm"])) < 1e-5 * float(g["grad_latent_norm"])
    np.testing.assert_allclose(latent.grad.numpy()[:, ::13, ::17], g["grad_latent_sample"],
                               atol=1e-5 * float(g["grad_latent_norm"]), rtol=0)
    check_grads({k: v.grad for k, v in sd.items() if v.grad is not None}, g, rtol=2e-5)
