#!/usr/bin/env python3
"""Generate golden fixtures from the REAL reference (build container only).

Run:  python tests/golden/make_golden.py          (needs /root/reference)

What it does: imports the reference's own Python for the hot path
(model/shape/implicit.py::Implicit, utils/eval_3D.py::{get_dense_3D_grid,
compute_level_grid, normalize_pc, compute_fscore}, utils/camera.py::
get_rotation_sphere, utils/pos_embed.py) from /root/reference, feeds it the
build-owned seeded inputs of zeroshape_amd/synthetic.py, and stores INPUT-free
expected outputs (arrays only) as small .npz files next to this script.  The
reference itself never travels to the GPU box; only these arrays do.

Un-vendored dependencies of the reference that are absent from this image are
replaced IN THIS PROCESS ONLY by the minimal stand-ins SURVEY.md section 8-c describes:
  - timm.models.vision_transformer.{Mlp, DropPath, Block}  (timm==0.6.12): Mlp is
    Linear -> nn.GELU() -> Dropout -> Linear -> Dropout with attribute names
    fc1/act/drop1/fc2/drop2; DropPath is the identity in eval mode; Block is
    never constructed on this path.
  - mcubes, trimesh, torchvision, cv2, pyrender, imageio, external.chamfer3D...:
    imported at module scope by utils/eval_3D.py / utils/util_vis.py but not
    used by the functions called here -> empty module objects.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def _install_stubs():
    timm = types.ModuleType("timm")
    models = types.ModuleType("timm.models")
    vt = types.ModuleType("timm.models.vision_transformer")

    class Mlp(nn.Module):  # timm 0.6.12 layers/mlp.py semantics
        def __init__(self, in_features, hidden_features=None, out_features=None,
                     act_layer=nn.GELU, bias=True, drop=0.0):
            super().__init__()
            out_features = out_features or in_features
            hidden_features = hidden_features or in_features
            self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
            self.act = act_layer()
            self.drop1 = nn.Dropout(drop)
            self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
            self.drop2 = nn.Dropout(drop)

        def forward(self, x):
            return self.drop2(self.fc2(self.drop1(self.act(self.fc1(x)))))

    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            assert not self.training, "stand-in is eval-only"
            return x

    class Block(nn.Module):
        def __init__(self, *a, **k):
            raise RuntimeError("timm Block stand-in must not be constructed")

    vt.Mlp, vt.DropPath, vt.Block = Mlp, DropPath, Block
    timm.models, models.vision_transformer = models, vt
    sys.modules.update({"timm": timm, "timm.models": models,
                        "timm.models.vision_transformer": vt})
    for name in ["mcubes", "trimesh", "cv2", "pyrender", "imageio", "torchvision",
                 "torchvision.transforms", "torchvision.transforms.functional",
                 "matplotlib", "matplotlib.pyplot", "external", "external.chamfer3D",
                 "external.chamfer3D.dist_chamfer_3D", "PIL", "PIL.Image", "PIL.ImageDraw",
                 "PIL.ImageFont", "tensorboard", "torch.utils.tensorboard"]:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    sys.modules["external.chamfer3D.dist_chamfer_3D"].chamfer_3DDist = object
    # utils/util_vis.py pulls in many render-only deps; eval_3D only needs one name.
    uv = types.ModuleType("utils.util_vis")
    uv.show_att_on_image = lambda *a, **k: None
    sys.modules["utils.util_vis"] = uv


def main():
    assert os.path.isdir(REF), "reference tree not present: run in the build container"
    _install_stubs()
    sys.path.insert(0, REF)
    from model.shape.implicit import Implicit            # noqa: E402  (reference)
    from utils import eval_3D as ref_eval                 # noqa: E402  (reference)
    from utils.camera import get_rotation_sphere          # noqa: E402  (reference)
    from utils.pos_embed import get_2d_sincos_pos_embed   # noqa: E402  (reference)
    from utils.util import EasyDict as edict              # noqa: E402  (reference)
    from zeroshape_amd import synthetic as syn            # build-owned inputs

    torch.manual_seed(0)
    torch.set_num_threads(8)

    # ---- reference decoder, options/shape.yaml:19-44 via graph_shape.py:58-64 ----
    net = Implicit(syn.NUM_PATCHES, latent_dim=syn.LATENT_DIM, semantic=False,
                   n_channels=syn.N_CHANNELS, n_blocks_attn=syn.ATT_BLOCKS,
                   n_layers_mlp=syn.MLP_LAYERS, num_heads=syn.NUM_HEADS, posenc_3D=0,
                   mlp_ratio=syn.MLP_RATIO, skip_in=list(syn.SKIP_IN), pos_perlayer=False).eval()
    ref_sd = net.state_dict()
    shapes = syn.impl_network_shapes()
    assert list(ref_sd.keys()) == list(shapes.keys()), "state_dict key contract drifted"
    for k in shapes:
        assert tuple(ref_sd[k].shape) == tuple(shapes[k]), k
    pos_ref = ref_sd["pos_embed"].numpy().copy()          # as initialised by the reference
    pos64 = get_2d_sincos_pos_embed(syn.N_CHANNELS, 14, cls_token=True)
    sd_np = syn.seeded_state_dict(seed=0, pos_embed=pos_ref)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)

    out = {}
    out["pos_embed_f32"] = pos_ref[0]                      # [197,256] float32
    out["pos_embed_f64_rows"] = pos64[[0, 1, 2, 14, 15, 100, 195, 196]]
    out["pos_embed_f64_sum"] = np.array([pos64.sum(), np.abs(pos64).sum()])

    latent = torch.from_numpy(syn.seeded_latent(seed=0, batch=2))
    opt = edict(dict(device="cpu", H=224, W=224,
                     eval=dict(vox_res=32, range=[-1.5, 1.5]), arch=dict(win_size=16)))

    # ---- linspace / grid (utils/eval_3D.py:11-20) ----
    for N in (32, 64, 128, 256):
        out["linspace_%d" % N] = torch.linspace(-1.5, 1.5, N + 1).numpy()
    var = edict(dict(idx=[0]))
    grid32 = ref_eval.get_dense_3D_grid(opt, var, N=32)
    out["grid32_corner_pts"] = grid32[0, [0, 0, 5, 32], [0, 7, 6, 32], [0, 3, 9, 32]].numpy()

    # ---- vox_res=32 full level grid through the reference's own loop ----
    with torch.no_grad():
        occ32, _ = ref_eval.compute_level_grid(opt, net, latent[:1], None, grid32, None, vis_attn=False)
    occ32 = occ32[0].numpy()
    out["occ32_bits"] = np.packbits((occ32 > 0.5).reshape(-1))
    out["occ32_shape"] = np.array(occ32.shape)
    out["occ32_stride5"] = occ32[::5, ::5, ::5].copy()
    # logits + attention for a few slices via the decoder call itself
    pts32 = grid32.view(1, 33, 33 * 33, 3)
    with torch.no_grad():
        for i in (0, 16, 32):
            lg, at = net(latent[:1], None, pts32[:, i])
            out["logit32_slice%d" % i] = lg[0].numpy()
            if i == 16:
                out["attn32_slice16_rows"] = at[0, ::97].numpy()

    # ---- vox_res 64 / 128: three slices each at point stride 16 ----
    for N in (64, 128):
        g = ref_eval.get_dense_3D_grid(opt, var, N=N).view(1, N + 1, (N + 1) ** 2, 3)
        with torch.no_grad():
            for i in (0, N // 2, N):
                lg, _ = net(latent[:1], None, g[:, i])
                out["logit%d_slice%d_s16" % (N, i)] = lg[0, ::16].numpy()

    # ---- training-shape call: B=2, M=4096 un-gridded points ----
    rs = np.random.RandomState(123)
    pts = torch.from_numpy(rs.uniform(-1, 1, size=(2, 4096, 3)).astype(np.float32))
    with torch.no_grad():
        lg, at = net(latent, None, pts)
    out["pts4096_logit"] = lg.numpy()
    out["pts4096_attn_rows"] = at[:, ::512].numpy()
    out["pts4096_attn_rowsum"] = at.sum(-1).numpy()
    # ---- pos_perlayer=True (the reference class's own default, implicit.py:197,269-272): pos_embed added to the latent rows in
    # front of EVERY block; same seeded weights ----
    net_pp = Implicit(syn.NUM_PATCHES, latent_dim=syn.LATENT_DIM, semantic=False,
                      n_channels=syn.N_CHANNELS, n_blocks_attn=syn.ATT_BLOCKS,
                      n_layers_mlp=syn.MLP_LAYERS, num_heads=syn.NUM_HEADS, posenc_3D=0,
                      mlp_ratio=syn.MLP_RATIO, skip_in=list(syn.SKIP_IN), pos_perlayer=True).eval()
    net_pp.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    with torch.no_grad():
        lg_pp, at_pp = net_pp(latent, None, pts[:, :1024])
        lg_pp32, _ = net_pp(latent[:1], None, pts32[:, 16])
    out["pp_pts1024_logit"] = lg_pp.numpy()
    out["pp_pts1024_attn_rows"] = at_pp[:, ::128].numpy()
    out["pp_logit32_slice16"] = lg_pp32[0].numpy()
    assert float((lg_pp - lg[:, :1024]).abs().max()) > 1e-3, "pos_perlayer must change the logits"
    np.savez_compressed(os.path.join(HERE, "decoder_golden.npz"), **out)
    print("decoder_golden.npz: %d arrays; logit32 range [%.3f, %.3f], occ32 frac>0.5 = %.4f"
          % (len(out), min(out["logit32_slice16"].min(), lg.min()), lg.max(),
             (occ32 > 0.5).mean()))

    # ---- geometry helpers ----
    geo = {}
    R = get_rotation_sphere(24, 24, 12, device="cpu")
    assert R.shape == (6912, 3, 3)
    geo["rot_rows"] = R[[0, 1, 12, 287, 288, 1234, 6911]].numpy()
    geo["rot_sum"] = np.array([R.double().sum().item(), R.double().abs().sum().item()])
    geo["rot_all_f32"] = R.numpy()
    pc = torch.from_numpy(syn.seeded_cloud(3, 2, 64)) * torch.tensor([1.0, 2.0, 3.0])
    geo["normalize_pc_out"] = ref_eval.normalize_pc(pc).numpy()
    d1 = torch.from_numpy(np.random.RandomState(5).uniform(0, 0.25, size=(3, 50)).astype(np.float32))
    d2 = torch.from_numpy(np.random.RandomState(6).uniform(0, 0.25, size=(3, 70)).astype(np.float32))
    d2[2] = 1.0  # recall 0 everywhere and precision 0 at small thresholds -> NaN->0 branch
    d1[2] = 1.0
    geo["fscore_out"] = ref_eval.compute_fscore(d1, d2, [0.005, 0.01, 0.02, 0.05, 0.1, 0.2]).numpy()
    np.savez_compressed(os.path.join(HERE, "geometry_golden.npz"), **geo)
    print("geometry_golden.npz: %d arrays" % len(geo))


if __name__ == "__main__":
    main()
