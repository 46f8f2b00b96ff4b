"""Mirror of the hot-path helpers of the reference's utils/util.py: the masked resampling used
between the depth head and the coordinate encoder (:323-345) and the checkpoint key helper
(:201-210).  The resampling runs on the HIP library (zs_masked_resample); there is no CPU
path."""
import torch

from .options import EasyDict  # noqa: F401  (utils/util.py:378-412)


def _masked_resample(map_, mask_input, size, bg):
    from .. import _lib
    lib = _lib.load()
    assert len(map_.shape) == len(mask_input.shape) == 4
    if not (map_.is_cuda and mask_input.is_cuda):
        raise ValueError("interpolate_*: GPU tensors required (there is no CPU path)")
    x = map_.detach().to(torch.float32).contiguous()
    m = mask_input.detach().to(torch.float32).contiguous()
    B, C, H, W = x.shape
    assert m.shape == (B, 1, H, W)
    Ho, Wo = (size, size) if isinstance(size, int) else (int(size[0]), int(size[1]))
    out = torch.empty(B, C, Ho, Wo, dtype=torch.float32, device=x.device)
    mask_out = torch.empty(B, 1, Ho, Wo, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.zs_masked_resample(_lib.ptr(x), _lib.ptr(m), B, C, H, W, Ho, Wo, float(bg), _lib.ptr(out),
                                          _lib.ptr(mask_out), _lib.current_stream_ptr(x.device)),
                   "zs_masked_resample")
    return out, mask_out


def interpolate_depth(depth_input, mask_input, size, bg_depth=20):
    """utils/util.py:323-332."""
    return _masked_resample(depth_input, mask_input, size, bg_depth)


def interpolate_coordmap(coord_map, mask_input, size, bg_coord=0):
    """utils/util.py:336-345: bilinear(coord*mask) / (bilinear(mask) + 1e-6) where the resampled
    mask is > 0.5, `bg_coord` elsewhere; returns (coord_out, mask_binary)."""
    return _masked_resample(coord_map, mask_input, size, bg_coord)


def get_child_state_dict(state_dict, key):
    """utils/util.py:201-210: the entries of a (possibly DDP-wrapped, ``module.``-prefixed)
    state dict that live under ``key.``, with the first name component dropped (like the
    reference, exactly one component is dropped even for a dotted key)."""
    prefix = key + "."
    child = {}
    for name, value in state_dict.items():
        name = name[len("module."):] if name.startswith("module.") else name
        if name.startswith(prefix):
            child[name.split(".", 1)[1]] = value
    return child


def move_to_device(X, device):
    """utils/util.py:162-175: recursive .to(device) through dicts / lists / namedtuples."""
    if isinstance(X, dict):
        for k, v in X.items():
            X[k] = move_to_device(v, device)
    elif isinstance(X, list):
        for i, e in enumerate(X):
            X[i] = move_to_device(e, device)
    elif isinstance(X, tuple) and hasattr(X, "_fields"):
        return type(X)(**move_to_device(X._asdict(), device))
    elif isinstance(X, torch.Tensor):
        return X.to(device=device, non_blocking=True)
    return X


def _bare_graph(model):
    g = model.graph
    return g.module if hasattr(g, "module") else g


def load_checkpoint(opt, model, load_name):
    """utils/util.py:228-239: load the children of ``model.graph`` that the checkpoint's "graph"
    entry covers (strictly, per child), skip the others."""
    checkpoint = torch.load(load_name, map_location="cpu")
    for name, child in _bare_graph(model).named_children():
        child_state_dict = get_child_state_dict(checkpoint["graph"], name)
        if child_state_dict:
            child.load_state_dict(child_state_dict, strict=True)
    return None, None, None, None


def resume_checkpoint(opt, model, best):
    """utils/util.py:212-226: latest.ckpt / best.ckpt of opt.output_path -> graph (strict) plus every
    training-state attribute of the runner named optim* / sched* / scaler* that the file holds."""
    load_name = "{0}/best.ckpt".format(opt.output_path) if best else "{0}/latest.ckpt".format(opt.output_path)
    checkpoint = torch.load(load_name, map_location="cpu")
    _bare_graph(model).load_state_dict(checkpoint["graph"], strict=True)
    from ..nn import autograd
    autograd.bump_generation()
    for key in model.__dict__:
        if key.split("_")[0] in ["optim", "sched", "scaler"] and key in checkpoint:
            getattr(model, key).load_state_dict(checkpoint[key])
    ep, it = checkpoint["epoch"], checkpoint["iter"]
    best_val, best_ep = checkpoint["best_val"], checkpoint["best_ep"] if "best_ep" in checkpoint else 0
    return ep, it, best_val, best_ep


def restore_checkpoint(opt, model, load_name=None, resume=False, best=False, evaluate=False):
    """utils/util.py:241-250."""
    assert not (load_name is not None and resume)
    if resume:
        return resume_checkpoint(opt, model, best)
    return load_checkpoint(opt, model, load_name)


def save_checkpoint(opt, model, ep, it, best_val, best_ep, latest=False, best=False, children=None):
    """utils/util.py:252-277: {epoch, iter, best_val, best_ep, graph} + the state of every runner
    attribute named optim* / sched* / scaler*, written to <output_path>/latest.ckpt (and best.ckpt /
    checkpoint/ep<N>.ckpt)."""
    import os
    import shutil
    os.makedirs("{0}/checkpoint".format(opt.output_path), exist_ok=True)
    sd = _bare_graph(model).state_dict()
    if children is not None:
        sd = {k: v for k, v in sd.items() if k.startswith(children)}
    checkpoint = dict(epoch=ep, iter=it, best_val=best_val, best_ep=best_ep, graph=sd)
    for key in model.__dict__:
        if key.split("_")[0] in ["optim", "sched", "scaler"]:
            checkpoint.update({key: getattr(model, key).state_dict()})
    torch.save(checkpoint, "{0}/latest.ckpt".format(opt.output_path))
    if best:
        shutil.copy("{0}/latest.ckpt".format(opt.output_path), "{0}/best.ckpt".format(opt.output_path))
    if not latest:
        shutil.copy("{0}/latest.ckpt".format(opt.output_path),
                    "{0}/checkpoint/ep{1}.ckpt".format(opt.output_path, ep))


# ---- process plumbing used by train.py / evaluate.py and the engines (utils/util.py:304-356) ----
def toggle_grad(model, requires_grad):
    for p in model.parameters():
        p.requires_grad_(requires_grad)


def is_port_in_use(port):
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        return s.connect_ex(('localhost', port)) == 0


def setup(rank, world_size, port_no):
    """One process per GPU over RCCL ("nccl" is RCCL on ROCm).  Rendezvous on 127.0.0.1; under
    torchrun (RANK / MASTER_* already in the environment) those settings win."""
    import os
    import torch.distributed as dist
    if dist.is_initialized():
        return
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(port_no))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    # ZS_DEVICE_OVERRIDE / ZS_DIST_BACKEND exist only to rehearse the multi-rank code paths on a single-GPU box
    # (all ranks on one device, gloo instead of RCCL; evaluate.py: ZS_VIRTUAL_RANKS)
    torch.cuda.set_device(int(os.environ.get("ZS_DEVICE_OVERRIDE", rank)))
    dist.init_process_group(os.environ.get("ZS_DIST_BACKEND", "nccl"), rank=rank, world_size=world_size)


def cleanup():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def print_eval(opt, loss=None, chamfer=None, depth_metrics=None):
    """utils/util.py:140-151."""
    message = "[eval] "
    if loss is not None:
        message += "loss:{:.3e}".format(float(loss.all))
    if chamfer is not None:
        message += " chamfer:{:.4f}|{:.4f}|{:.4f}".format(chamfer[0], chamfer[1], (chamfer[0] + chamfer[1]) / 2)
    if depth_metrics is not None:
        message += ", ".join("{}:{:.4f}".format(k, v) for k, v in depth_metrics.items())
    print(message)
