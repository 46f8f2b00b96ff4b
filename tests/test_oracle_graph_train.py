"""Oracle training step of the whole shape graph (oracle/train_ref.py) vs the golden outputs of
the REAL reference's Graph.forward(training=True) + backward (tests/golden/graph_train_golden.npz)."""
import numpy as np
import torch

from oracle import train_ref
from zeroshape_amd import synthetic as syn


# Whole-graph gradients at this fixture are only loosely comparable upstream of the decoder: the
# reference's OWN gradients of dpt_depth / intr_head / coord_encoder move by ~30 % (median, relative
# L2) when 1e-6 noise is added to the input image (measured on /root/reference with these weights;
# BatchNorm over the 4 nearly identical 1x1 global-token samples amplifies differences ~100x per
# layer), while the decoder's move by 3e-4.  So: decoder gradients are held to DECODER_RTOL, the
# rest to a factor-of-two sanity band here, and to tight tolerances module by module on
# well-conditioned inputs in tests/test_gpu_train_encoder.py.
DECODER_RTOL = 5e-3


def graph_train_inputs(g):
    assert int(g["batch"]) == 4 and int(g["depth_seed"]) == 3
    scenes = [syn.seeded_rgb_scene(seed=s, batch=2) for s in (0, 1)]
    rgb, mask = [torch.from_numpy(np.concatenate([sc[i] for sc in scenes])) for i in (0, 1)]
    depth_gt = np.concatenate([syn.seeded_depth_scene(seed=s, batch=2)[0] for s in (3, 4)])
    return dict(rgb_input_map=rgb, mask_input_map=mask, depth_input_map=torch.from_numpy(depth_gt),
                intr=torch.from_numpy(g["intr_gt"]), pose_gt=torch.from_numpy(g["pose_gt"]),
                gt_sample_points=torch.from_numpy(g["gt_sample_points"]),
                gt_sample_sdf=torch.from_numpy(g["gt_sample_sdf"]))


def full_state_dict(encoder_sd, seeded_sd):
    sd = {k: v.clone() for k, v in encoder_sd.items()}
    sd.update({"impl_network." + k: v.clone() for k, v in seeded_sd.items()})
    return sd


def check_graph_grads(named_grads, g, rtol=DECODER_RTOL):
    n = 0
    for key in g:
        if not key.startswith("gnorm/"):
            continue
        name = key[6:]
        want_norm = float(g[key])
        got = named_grads[name].double().reshape(-1)
        n += 1
        if not name.startswith("impl_network."):
            assert 0.5 * want_norm <= float(got.norm()) <= 2.0 * want_norm + 1e-12, (name, float(got.norm()), want_norm)
            continue
        assert abs(float(got.norm()) - want_norm) <= rtol * want_norm + 1e-10, (name, float(got.norm()), want_norm)
        step = max(1, got.numel() // 64)
        err = np.abs(got[::step][:64].numpy() - g["gs/" + name]).max()
        scale = want_norm / np.sqrt(got.numel()) + 1e-12          # the tensor's RMS gradient
        assert err <= 20 * rtol * scale + 1e-10, (name, err, scale)
    return n


def test_graph_train_step_matches_reference(encoder_sd, seeded_sd, graph_train_golden):
    g = graph_train_golden
    sd = full_state_dict(encoder_sd, seeded_sd)
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")) and k != "impl_network.pos_embed":
            v.requires_grad_(True)
    scales = [torch.from_numpy(s) for s in g["drop_scales"]]
    loss, out = train_ref.graph_train_forward(sd, graph_train_inputs(g), scales)
    assert abs(float(loss) - float(g["loss"])) < 2e-6
    np.testing.assert_allclose(out["pred_sample_occ"].detach().numpy(), g["pred_sample_occ"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(out["gt_points_cam"].numpy().reshape(-1)[::7], g["gt_points_cam_s7"], atol=1e-5, rtol=0)
    # BatchNorm on batch statistics over 4 samples amplifies the ~1e-6 rounding differences of the
    # depth map by ~500x through the 50-layer coordinate encoder (measured: 8e-4 between two CPU
    # runs of the SAME ops, module vs functional form); the loss is insensitive to it
    np.testing.assert_allclose(out["latent_depth"].detach().numpy().reshape(-1)[::211], g["latent_s211"], atol=3e-3, rtol=0)
    np.testing.assert_allclose(out["intr_pred"].detach().numpy(), g["intr_pred"], rtol=1e-5)
    loss.backward()
    grads = {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}
    n = check_graph_grads(grads, g)
    assert n == 609, n
    check_bn_stats(sd, g)


def check_bn_stats(sd, g):
    """BatchNorm running statistics after the step (momentum 0.1, unbiased variance)."""
    for key in g:
        if not key.startswith("bn/"):
            continue
        got = sd[key[3:]].detach().cpu().numpy()
        if key.endswith("num_batches_tracked"):
            assert int(got) == int(g[key]) == 1
        else:
            deep = "fc.1" in key or "layer3" in key or "depth_feat_proj" in key     # behind the amplifying layers
            np.testing.assert_allclose(got, g[key], atol=3e-2 if deep else 1e-4, rtol=0, err_msg=key)
