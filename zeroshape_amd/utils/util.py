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
