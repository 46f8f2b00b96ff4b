"""Build-owned deterministic synthetic inputs (no checkpoints or datasets ship with
the reference: weights/.gitignore:1-3).  The same generator is used by the golden
fixture script (in the build container, beside the real reference), by the tests
and by bench.py (on the GPU box), so no weight files need to travel.

Shapes/keys follow the reference's ``impl_network`` state_dict
(model/shape/implicit.py:186-231) for the default
options/shape.yaml:19-44 configuration.
"""
import numpy as np

# options/shape.yaml:19-44
N_CHANNELS = 256
LATENT_DIM = 256
NUM_HEADS = 8
ATT_BLOCKS = 2
MLP_RATIO = 4
MLP_LAYERS = 8
SKIP_IN = (2, 4, 6)
NUM_PATCHES = 196  # (224 // 16) ** 2, graph_shape.py:58-64


def impl_network_shapes(n_channels=N_CHANNELS, latent_dim=LATENT_DIM, att_blocks=ATT_BLOCKS,
                        mlp_ratio=MLP_RATIO, mlp_layers=MLP_LAYERS, skip_in=SKIP_IN,
                        num_patches=NUM_PATCHES, posenc_3D=0):
    """Ordered {key: shape} of the reference decoder's state_dict (posenc_3D: implicit.py:139-150 widens the point part of
    the per-point MLP's inputs to 3 + 6 posenc_3D)."""
    C = n_channels
    s = {}
    s["pos_embed"] = (1, num_patches + 1, C)
    s["point_proj.proj.weight"] = (C, 3)
    s["point_proj.proj.bias"] = (C,)
    s["latent_proj.weight"] = (C, latent_dim)
    s["latent_proj.bias"] = (C,)
    for b in range(att_blocks):
        p = "blocks_attn.%d." % b
        s[p + "norm1.weight"] = (C,)
        s[p + "norm1.bias"] = (C,)
        s[p + "attn.qkv.weight"] = (3 * C, C)
        s[p + "attn.qkv.bias"] = (3 * C,)
        s[p + "attn.proj.weight"] = (C, C)
        s[p + "attn.proj.bias"] = (C,)
        s[p + "norm2.weight"] = (C,)
        s[p + "norm2.bias"] = (C,)
        s[p + "mlp.fc1.weight"] = (int(C * mlp_ratio), C)
        s[p + "mlp.fc1.bias"] = (int(C * mlp_ratio),)
        s[p + "mlp.fc2.weight"] = (C, int(C * mlp_ratio))
        s[p + "mlp.fc2.bias"] = (C,)
    s["norm.weight"] = (C,)
    s["norm.bias"] = (C,)
    if mlp_layers == 0:                                   # implicit.py:226-229: a prediction head instead of the MLP
        s["pred_head.weight"] = (1, C)
        s["pred_head.bias"] = (1,)
        return s
    dims = [3 + 6 * int(posenc_3D) + C] + [C] * mlp_layers + [1]
    for l in range(len(dims) - 1):
        in_dim = dims[l] + (dims[0] if l in skip_in else 0)
        s["impl_mlp.layers.%d.weight" % l] = (dims[l + 1], in_dim)
        s["impl_mlp.layers.%d.bias" % l] = (dims[l + 1],)
    return s


def seeded_state_dict(seed=0, pos_embed=None, **cfg):
    """Deterministic fp32 numpy state_dict.  Unlike the reference's init
    (xavier weights, zero biases, unit LayerNorm - implicit.py:238-249) every
    tensor is non-trivial so that bias / affine / skip paths are exercised:
      weights  ~ U(-a, a), a = sqrt(6/(fan_in+fan_out))   (xavier-uniform range)
      biases   ~ 0.1 * N(0,1)
      LN gamma ~ 1 + 0.1 * N(0,1);  LN beta ~ 0.1 * N(0,1)
    (logits then straddle 0: about a third of a [-1.5,1.5]^3 grid is 'inside').
    One RandomState per key (seed, crc of key) -> independent of key order.
    ``pos_embed`` must be supplied by the caller (fixed sin-cos table)."""
    import zlib
    shapes = impl_network_shapes(**cfg)
    sd = {}
    for k, shp in shapes.items():
        rs = np.random.RandomState((seed * 1000003 + zlib.crc32(k.encode())) % (2 ** 31))
        if k == "pos_embed":
            if pos_embed is None:
                raise ValueError("pos_embed table required")
            sd[k] = np.asarray(pos_embed, np.float32).reshape(shp)
        elif k.endswith("weight") and len(shp) == 2:
            a = np.sqrt(6.0 / (shp[0] + shp[1]))
            w = rs.uniform(-a, a, size=shp)
            sd[k] = w.astype(np.float32)
        elif "norm" in k and k.endswith("weight"):
            sd[k] = (1.0 + 0.1 * rs.randn(*shp)).astype(np.float32)
        else:
            sd[k] = (0.1 * rs.randn(*shp)).astype(np.float32)
    return sd


def confident_state_dict(sd, gain, layers=3):
    """A decoder state dict whose logits are ~gain x those of ``sd``: weight and bias of the last ``layers`` linear layers of
    impl_mlp each x gain^(1/layers) (softplus(beta=100) is positively homogeneous to ~1e-2, so the factors multiply; spread
    over several layers so that no single weight leaves the split arithmetic's host envelope, program.W_MAX).  A stand-in for
    the logit scale of a converged checkpoint (|logit| 30-100) - no trained weights exist in this container."""
    out = dict(sd)
    last = max(int(k.split(".")[2]) for k in sd if k.startswith("impl_mlp.layers."))
    f = float(gain) ** (1.0 / layers)
    for l in range(last - layers + 1, last + 1):
        for part in ("weight", "bias"):
            k = "impl_mlp.layers.%d.%s" % (l, part)
            out[k] = sd[k] * f
    return out


def seeded_latent(seed=0, batch=1, n_tokens=NUM_PATCHES + 1, dim=LATENT_DIM):
    """latent_depth stand-in: N(0,1) [B, 197, 256]."""
    rs = np.random.RandomState(seed + 7919)
    return rs.randn(batch, n_tokens, dim).astype(np.float32)


def seeded_cloud(seed, batch, n, lo=-0.5, hi=0.5):
    """uniform point clouds in [lo,hi]^3 (SURVEY.md section 8-d Chamfer inputs)."""
    rs = np.random.RandomState(seed + 104729)
    return rs.uniform(lo, hi, size=(batch, n, 3)).astype(np.float32)


def icp_clouds():
    """(a [2,400,3], b [2,350,3]): a flattened uniform cloud and a rotated (0.35 rad about z), shifted, slightly noisy
    subset of it - the ICP fixture (tests/golden/make_icp_golden.py; well separated points, no near ties)."""
    rs = np.random.RandomState(21)
    a = seeded_cloud(31, 2, 400, -0.5, 0.5) * np.array([1.0, 0.7, 0.4], np.float32)
    ang = 0.35
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
    b = a @ R.T + np.array([0.05, -0.03, 0.02], np.float32) + 2e-3 * rs.randn(*a.shape).astype(np.float32)
    b = b[:, rs.permutation(400)[:350]]
    return a.astype(np.float32), b.astype(np.float32)


def ellipsoid_cloud(seed, n, radii=(0.5, 0.35, 0.25)):
    """points on an axis-aligned ellipsoid surface (brute-force-search test shape)."""
    rs = np.random.RandomState(seed + 15485863)
    v = rs.randn(n, 3)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return (v * np.asarray(radii)).astype(np.float32)


def seeded_depth_scene(seed=0, batch=2, size=224):
    """Build-owned inputs for the seen-surface front-end (graph_shape.py:118-144): a smooth
    bumpy depth map in [0.7, 1.7], an object mask (disk with random holes; the last sample of a
    batch >= 3 keeps only a few pixels), and raw intrinsics parameters.  Returns float32 arrays
    depth [B,1,S,S], mask [B,1,S,S] in {0,1}, intr_params [B,3]."""
    rs = np.random.RandomState(1000 + seed)
    y, x = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    depth, mask = [], []
    for b in range(batch):
        fx, fy, ph = rs.uniform(20, 60), rs.uniform(20, 60), rs.uniform(0, 6.28)
        d = 1.2 + 0.3 * np.sin(x / fx + ph) * np.cos(y / fy) + 0.05 * rs.rand(size, size)
        cx, cy, rad = rs.uniform(0.35, 0.65) * size, rs.uniform(0.35, 0.65) * size, rs.uniform(0.25, 0.4) * size
        m = ((x - cx) ** 2 + (y - cy) ** 2 < rad ** 2) & (rs.rand(size, size) > 0.05)
        if batch >= 3 and b == batch - 1:
            m = np.zeros((size, size), bool)
            m[rs.randint(0, size, 5), rs.randint(0, size, 5)] = True
        depth.append(d)
        mask.append(m)
    depth = np.stack(depth)[:, None].astype(np.float32)
    mask = np.stack(mask)[:, None].astype(np.float32)
    params = rs.uniform(-0.5, 0.5, size=(batch, 3)).astype(np.float32)
    return depth, mask, params


def seeded_depth_pair(seed=0, batch=2, size=224):
    """Prediction / target / mask for the depth metrics (utils/eval_depth.py): the prediction
    is an affine-in-disparity distortion of the target plus noise.  float32 [B,1,S,S] each."""
    rs = np.random.RandomState(2000 + seed)
    target, mask, _ = seeded_depth_scene(seed + 7, batch, size)
    disp = 1.0 / target
    pred_disp = rs.uniform(0.5, 2.0, size=(batch, 1, 1, 1)) * disp + rs.uniform(0.0, 0.3, size=(batch, 1, 1, 1)) \
        + 0.15 * rs.randn(*target.shape)
    pred = (1.0 / np.maximum(pred_disp, 0.05)).astype(np.float32)
    return pred, target.astype(np.float32), mask


def seeded_encoder_state_dict(shapes, seed=0):
    """Build-owned deterministic parameters for the image encoders.  `shapes` is an ordered
    mapping name -> shape with the reference's state-dict names; the value of every entry is
    drawn from a RandomState seeded by (seed, position) with a scale picked from the name, so
    activations stay O(1) through ~100 layers:
      conv / linear weights  N(0, 1/fan_in);  biases, BN/GN/LN beta, running_mean  0.1 N(0,1)
      BN/GN/LN gamma  U(0.8, 1.2);  running_var  U(0.8, 1.2);  tokens / pos_embed  0.5 N(0,1)
      num_batches_tracked  0."""
    out = {}
    for i, (name, shape) in enumerate(shapes.items()):
        rs = np.random.RandomState((seed * 7919 + i * 104729 + 17) % (2 ** 31))
        shape = tuple(int(s) for s in shape)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            v = np.zeros(shape, np.int64)
        elif leaf == "running_var":
            v = rs.uniform(0.8, 1.2, shape)
        elif leaf == "running_mean":
            v = 0.1 * rs.randn(*shape)
        elif leaf in ("cls_token", "pos_embed", "two_d_pos_embed", "invalid_coord_token"):
            v = 0.5 * rs.randn(*shape)
        elif leaf == "bias":
            v = 0.1 * rs.randn(*shape)
        elif leaf == "weight" and len(shape) == 1:
            v = rs.uniform(0.8, 1.2, shape)
        elif leaf == "weight":
            fan_in = int(np.prod(shape[1:]))
            v = rs.randn(*shape) / np.sqrt(fan_in)
        else:
            raise KeyError("seeded_encoder_state_dict: no rule for %r" % name)
        out[name] = v.astype(np.int64 if leaf == "num_batches_tracked" else np.float32)
    return out


def seeded_rgb_scene(seed=0, batch=2, size=224):
    """RGB image in [0,1] and object mask in {0,1} ([B,3,S,S], [B,1,S,S] float32): a shaded blob
    over a white background, like the reference's preprocessed inputs (data/synthetic.py)."""
    depth, mask, _ = seeded_depth_scene(seed, batch, size)
    rs = np.random.RandomState(3000 + seed)
    tint = rs.uniform(0.2, 0.9, size=(batch, 3, 1, 1))
    shade = (1.9 - depth) / 1.2
    rgb = mask * (tint * shade + 0.05 * rs.rand(batch, 3, size, size)) + (1 - mask) * 1.0
    return np.clip(rgb, 0, 1).astype(np.float32), mask
