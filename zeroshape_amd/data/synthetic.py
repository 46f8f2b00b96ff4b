"""Analytic stand-in for the reference's data/synthetic.py::Dataset (Objaverse/ShapeNet renders
that live on Dropbox and cannot be fetched here): every item is a posed ellipsoid rendered in
closed form, returned under the SAME sample-dict keys and tensor shapes (data/synthetic.py:126-176)
so Graph.forward / eval_metrics / the Runner run unchanged without any file on disk:

  idx, category_label, pose_gt [3,4], intr [3,3], rgb_input_map [3,H,W] in [0,1] (white
  background, data.bgcolor 1), mask_input_map [1,H,W] in {0,1}, depth_input_map [1,H,W] (0 on the
  background, :98-101), dpc.points [N,3] (object frame), gt_sample_points [n_sdf,3],
  gt_sample_sdf [n_sdf]

Host-side numpy: this is input synthesis, not part of the measured path."""
import numpy as np
import torch

SYNTHETIC_STANDIN = True     # data/__init__.py: load_by_name's fence


def _rotation(rs):
    q = rs.randn(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


class Dataset(torch.utils.data.Dataset):
    label2cat = ["ellipsoid"]
    cat2label = {"ellipsoid": 0}

    @property
    def synthetic_standin(self):
        """First line of every result file an engine writes from this data (model/shape_engine.py: evaluate)."""
        name = type(self).__module__.rsplit(".", 1)[-1]
        return ("# SYNTHETIC STAND-IN DATA (analytic ellipsoids under the %s loader's keys) - NOT %s benchmark numbers"
                % (name, name))

    def __init__(self, opt, split="test", n_items=None, load_3D=True, n_points=16384, seed=0):
        super().__init__()
        import os
        if n_items is None:            # launched through train.py / evaluate.py: size from the environment
            n_items = int(os.environ.get("ZS_SYNTHETIC_ITEMS", "8"))
        if split == "train":
            seed += 1
        self.opt, self.split, self.load_3D, self.n_points, self.seed = opt, split, load_3D, n_points, seed
        self.list = list(range(n_items))

    def __len__(self):
        return len(self.list)

    def id_filename_mapping(self, opt, outpath):
        """data/synthetic.py:178-186: index -> item name, one per line."""
        with open(outpath, "w") as f:
            for i in self.list:
                f.write("%d ellipsoid_%04d\n" % (i, i))

    def __getitem__(self, idx):
        opt = self.opt
        H, W = opt.H, opt.W
        rs = np.random.RandomState(self.seed * 100003 + idx)
        radii = rs.uniform(0.2, 0.5, 3)
        Q = _rotation(rs)                                  # object axes in the world frame
        A = Q @ np.diag(1.0 / radii ** 2) @ Q.T            # x^T A x = 1
        R = _rotation(rs)                                  # world -> camera
        t = np.array([rs.uniform(-0.1, 0.1), rs.uniform(-0.1, 0.1), rs.uniform(1.6, 2.2)])
        K = np.array([[1.3875 * W, 0, W / 2], [0, 1.3875 * H, H / 2], [0, 0, 1.0]])
        # rays x_cam = s d (d_z = 1 => s is the depth); world point R^T (s d - t)
        v, u = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        d = np.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], np.ones_like(u, float)], -1)
        dw, ow = d @ R, -(R.T @ t)                         # R^T d per pixel (row-vector form), origin
        a = np.einsum("hwi,ij,hwj->hw", dw, A, dw)
        b = 2 * np.einsum("hwi,ij,j->hw", dw, A, ow)
        c = ow @ A @ ow - 1
        disc = b * b - 4 * a * c
        hit = disc > 0
        s = np.where(hit, (-b - np.sqrt(np.where(hit, disc, 0))) / (2 * a), 0.0)
        depth = np.where(hit, s, 0.0)
        mask = hit.astype(np.float32)
        xw = ow + dw * s[..., None]
        normal = xw @ A
        normal /= np.maximum(np.linalg.norm(normal, axis=-1, keepdims=True), 1e-9)
        light = R.T @ np.array([0.3, -0.5, -0.8])
        shade = np.clip(normal @ (light / np.linalg.norm(light)), 0.05, 1.0)
        tint = rs.uniform(0.3, 0.9, 3)
        rgb = np.where(hit[None], tint[:, None, None] * shade[None], 1.0)
        pose = np.concatenate([R, t[:, None]], 1)
        sample = dict(idx=idx, category_label=0,
                      pose_gt=torch.from_numpy(pose).float(), intr=torch.from_numpy(K).float(),
                      rgb_input_map=torch.from_numpy(rgb).float(),
                      mask_input_map=torch.from_numpy(mask)[None].float(),
                      depth_input_map=torch.from_numpy(depth)[None].float())
        if not self.load_3D:
            return sample
        dirs = rs.randn(self.n_points, 3)
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        pts = (dirs * radii) @ Q.T
        sample.update(dpc=dict(points=torch.from_numpy(pts).float()))
        n_sdf = opt.training.n_sdf_points if "training" in opt and opt.training.n_sdf_points else 4096
        q = rs.uniform(-0.55, 0.55, (n_sdf, 3))
        ql = q @ Q                                          # object-frame coordinates
        k0 = np.linalg.norm(ql / radii, axis=1)
        k1 = np.linalg.norm(ql / radii ** 2, axis=1)
        sdf = k0 * (k0 - 1.0) / np.maximum(k1, 1e-9)       # first-order distance; exact sign
        sample.update(gt_sample_points=torch.from_numpy(q).float(), gt_sample_sdf=torch.from_numpy(sdf).float())
        return sample
