"""Pix3D-shaped evaluation items (the reference's data/pix3d.py:86-113) without the Pix3D files
(data/Pix3D lives on the authors' Dropbox and cannot be fetched here): the analytic ellipsoid
renders of data/synthetic.py returned under Pix3D's sample-dict keys and conventions, so
`evaluate.py --data.dataset_test=pix3d --eval.vox_res=128 --eval.brute_force` (BASELINE config 3,
README.md:108) runs end to end, including the Pix3D-only branch of the metrics
(utils/eval_3D.py:122-123, 189-190: `points[:,:,:2] *= -1` after the rotation into the view frame).

  idx, rgb_input_map [3,H,W], mask_input_map [1,H,W], category_label, pose_gt [3,4], intr [3,3],
  dpc.points [N,3]                                  (no depth map, no SDF samples: data/pix3d.py:86-113)

Pix3D's camera looks down -z with x to the left: its rotation is diag(-1,-1,1) times the rotation
of the camera convention the renders use, which is exactly what that flip undoes - the items
carry `pose_gt` in Pix3D's convention, so the GT cloud lands in the view frame only through the
flip branch.  Reading the real files is not built: there is nothing here to pin it against."""
import numpy as np
import torch

from . import synthetic

SYNTHETIC_STANDIN = True     # data/__init__.py: load_by_name's fence

FLIP = np.diag([-1.0, -1.0, 1.0])


class Dataset(synthetic.Dataset):
    cat_id_all = ["bed", "bookcase", "chair", "desk", "misc", "sofa", "table", "tool", "wardrobe"]   # data/pix3d.py:18-28

    def __init__(self, opt, split="train", n_items=None, n_points=16384, seed=0):
        super().__init__(opt, split=split, n_items=n_items, load_3D=True, n_points=n_points, seed=seed + 7)
        want = None
        try:
            want = opt.data.pix3d.cat
        except (AttributeError, KeyError):
            pass
        self.cat_id = list(self.cat_id_all) if not want else [c for c in self.cat_id_all if c in str(want).split(",")]
        self.cat2label = {c: i for i, c in enumerate(self.cat_id)}          # data/pix3d.py:31-41
        self.label2cat = list(self.cat_id)
        self.path = "data/Pix3D"

    def id_filename_mapping(self, opt, outpath):
        """data/pix3d.py:74-84: index, image, mask and point-cloud file per line."""
        with open(outpath, "w") as f:
            for i in self.list:
                c = self.cat_id[i % len(self.cat_id)]
                f.write("{0} {1}/img/{2}/{3:04d}.png {1}/mask/{2}/{3:04d}.png {1}/pointclouds/{2}/{3:04d}.npy\n".format(
                    i, self.path, c, i))

    def __getitem__(self, idx):
        s = super().__getitem__(idx)
        pose = s["pose_gt"].double().numpy()
        pose = torch.from_numpy(FLIP @ pose).float()                         # Pix3D's camera convention
        return dict(idx=idx, rgb_input_map=s["rgb_input_map"], mask_input_map=s["mask_input_map"],
                    category_label=idx % len(self.cat_id), pose_gt=pose, intr=s["intr"], dpc=s["dpc"])
