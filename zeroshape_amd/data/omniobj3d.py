"""OmniObject3D-shaped evaluation items (the reference's data/omniobj3d.py:129-165) without the
OmniObject3D files (Dropbox, not fetchable here): the analytic ellipsoid renders of
data/synthetic.py under that loader's sample-dict keys, so `evaluate.py
--data.dataset_test=omniobj3d --eval.vox_res=256` (BASELINE config 5) runs end to end.

  idx, category_label, pose_gt [3,4], intr [3,3] (f = 1.3875, data/omniobj3d.py:53-58),
  rgb_input_map [3,H,W] (background data.bgcolor), mask_input_map [1,H,W] = depth != 0,
  depth_input_map [1,H,W], dpc.points [N,3] when load_3D

It is a test-split-only data set in the reference too (`self.get_list(opt, "test")`, :29).
Reading the real files is not built: there is nothing here to pin it against."""
from . import synthetic

SYNTHETIC_STANDIN = True     # data/__init__.py: load_by_name's fence


class Dataset(synthetic.Dataset):
    cat_names = ["ellipsoid_flat", "ellipsoid_long", "ellipsoid_round"]

    def __init__(self, opt, split="train", load_3D=True, n_items=None, n_points=16384, seed=0):
        super().__init__(opt, split="test", n_items=n_items, load_3D=load_3D, n_points=n_points, seed=seed + 13)
        self.path = "data/OmniObject3D"
        self.cat2label = {c: i for i, c in enumerate(self.cat_names)}       # data/omniobj3d.py:20-26
        self.label2cat = list(self.cat_names)

    def id_filename_mapping(self, opt, outpath):
        """data/omniobj3d.py:40-48."""
        with open(outpath, "w") as f:
            for i in self.list:
                c = self.cat_names[i % len(self.cat_names)]
                f.write("{0} {1}/images_processed/{2}/{3:04d}.png {1}/masks_processed/{2}/{3:04d}.png "
                        "{1}/pointclouds/{2}/{3:04d}.npy\n".format(i, self.path, c, i))

    def __getitem__(self, idx):
        s = super().__getitem__(idx)
        out = dict(idx=idx, category_label=idx % len(self.cat_names), pose_gt=s["pose_gt"], intr=s["intr"],
                   rgb_input_map=s["rgb_input_map"], mask_input_map=(s["depth_input_map"] != 0).float(),
                   depth_input_map=s["depth_input_map"])
        if self.load_3D:
            out["dpc"] = s["dpc"]
        return out
