"""Oracle: one TRAINING step of the shape graph on the CPU (torch autograd = gradient oracle).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates what model/shape_engine.py:248-297 (Runner.train_iteration) runs on the reference's
graph in .train() mode: Graph.forward(opt, var, training=True, get_loss=True)
(model/compute_graph/graph_shape.py:115-204) = the encoders of oracle/encoder_ref.py with
BatchNorm on batch statistics (running statistics updated in place in the state dict), the
seen-surface geometry of oracle/frontend_ref.py, the ground-truth branch (:152-181), the decoder
of oracle/decoder_ref.py::implicit_forward_train (DropPath factors supplied by the caller) and
Loss.shape_loss (utils/loss.py:18-28).

The inference restatements are reused as they are: their torch.no_grad decorators are lifted for
the duration of a call (differentiable()), nothing is duplicated.
Pinned by tests/golden/graph_train_golden.npz (tests/golden/make_graph_train_golden.py).
"""
import contextlib

import torch
import torch.nn.functional as F

from . import decoder_ref, encoder_ref, frontend_ref


def _bn_train(sd, p, x):
    """nn.BatchNorm2d.forward in training mode (momentum 0.1; updates the running statistics)."""
    if p + ".num_batches_tracked" in sd:
        sd[p + ".num_batches_tracked"] += 1
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        True, 0.1, 1e-5)


@contextlib.contextmanager
def differentiable():
    """Lift the @torch.no_grad() decorators of the inference oracles and switch their BatchNorm to
    batch statistics while the block runs."""
    saved = []
    for mod in (encoder_ref, frontend_ref):
        for name, fn in list(vars(mod).items()):
            if callable(fn) and hasattr(fn, "__wrapped__"):
                saved.append((mod, name, fn))
                setattr(mod, name, fn.__wrapped__)
    saved.append((encoder_ref, "_bn", encoder_ref._bn))
    encoder_ref._bn = _bn_train
    try:
        with torch.enable_grad():
            yield
    finally:
        for mod, name, fn in saved:
            setattr(mod, name, fn)


def graph_train_forward(sd, var, drop_scales, H=224, W=224, impt_thres=0.01, impt_weight=1.0):
    """sd: full graph state dict (tensors; the trainable ones with requires_grad=True).  var: dict
    with rgb_input_map, mask_input_map, depth_input_map, intr, pose_gt, gt_sample_points,
    gt_sample_sdf.  Returns (loss, out dict)."""
    sub = encoder_ref._sub
    with differentiable():
        rgb, mask = var["rgb_input_map"], var["mask_input_map"]
        depth, feat = encoder_ref.dpt_depth(sub(sd, "dpt_depth."), rgb)
        f = encoder_ref.bottleneck_conv(sd, "intr_head.1", encoder_ref.bottleneck_conv(sd, "intr_head.0", feat, 3), 3)
        params = F.linear(f.mean((2, 3)), sd["intr_proj.weight"], sd["intr_proj.bias"])
        intr = frontend_ref.intr_param2mtx(H, W, params)
        seen, coord, mask_dsp, _, _ = frontend_ref.seen_surface(depth, intr, mask, 1)
        latent = encoder_ref.coord_enc_res(sub(sd, "coord_encoder."), coord, mask_dsp)
        with torch.no_grad():                                                    # graph_shape.py:152-181
            _, _, _, mean_gt, scale_gt = frontend_ref.seen_surface(var["depth_input_map"], var["intr"], mask, 1)
            pose = var["pose_gt"]
            cam = (pose[:, :, :3] @ var["gt_sample_points"].permute(0, 2, 1) + pose[:, :, 3:]).permute(0, 2, 1)
            gt_points_cam = (cam - mean_gt[:, None]) / scale_gt[:, None, None]
        pred = decoder_ref.implicit_forward_train(sub(sd, "impl_network."), latent, gt_points_cam, drop_scales)
        loss = decoder_ref.shape_loss(pred, var["gt_sample_sdf"], impt_thres, impt_weight)
    return loss, dict(depth_pred=depth, intr_pred=intr, seen_points=seen, latent_depth=latent,
                      gt_points_cam=gt_points_cam, pred_sample_occ=pred)
