#!/usr/bin/env python3
"""Single-image demo with the reference's command line (demo.py of ZeroShape):

    python demo.py --yaml=options/shape.yaml --task=shape --datadir=examples --eval.vox_res=128 --ckpt=weights/shape.ckpt
    python demo.py --yaml=options/depth.yaml --task=depth --datadir=examples --ckpt=weights/depth.ckpt

<datadir>/images/*.png|jpg with <datadir>/masks/<same name>.png -> <datadir>/preds/: the cropped
input and mask, and for the shape task the reconstructed mesh (marching cubes at 0.5 of the
(vox_res+1)^3 occupancy grid, as OBJ), for the depth task the depth map.  Same preprocessing as the
reference (demo.py:27-70: mask binarised at 127, square crop 1.2x the mask's bounding box, resize to
image_size, white background, intrinsics f = 1.3875 * W).  Everything after image loading runs on
the GPU through the HIP library.
"""
import importlib
import os
import shutil
import sys

import numpy as np
import torch
from PIL import Image

import zeroshape_amd.compat as compat

compat.install()
import utils.options as options                                              # noqa: E402
from utils.eval_3D import compute_level_grid, convert_to_explicit, get_dense_3D_grid   # noqa: E402
from utils.util import EasyDict as edict                                     # noqa: E402
from zeroshape_amd.utils import util_vis                                     # noqa: E402


def bbox_from_mask(mask, thr):
    on = (mask > thr).astype(np.float32)
    assert on.sum() > 0, "Empty mask!"
    cols, rows = np.flatnonzero(on.sum(axis=-2)), np.flatnonzero(on.sum(axis=-1))
    return cols[0], rows[0], cols[-1], rows[-1]


def preprocess_image(opt, image, bbox):
    """demo.py:38-58: square crop (PIL pads outside the frame with transparent black, like
    torchvision's crop of a PIL image), resize, composite on the background colour."""
    x1, y1, x2, y2 = bbox
    size = max(y2 - y1, x2 - x1) * 1.2
    yc, xc = (y1 + y2) / 2, (x1 + x2) / 2
    top, left, side = int(yc - size / 2), int(xc - size / 2), int(size)
    image = image.crop((left, top, left + side, top + side))
    if image.size[0] != opt.W or image.size[1] != opt.H:
        image = image.resize((opt.W, opt.H))
    arr = torch.from_numpy(np.asarray(image, np.float32) / 255.0).permute(2, 0, 1)
    rgb, mask = arr[:3], arr[3:]
    if opt.data.bgcolor is not None:
        rgb = rgb * mask + opt.data.bgcolor * (1 - mask)
        mask = (mask > 0.5).float()
    return rgb, mask


def get_image(opt, image_name, mask_name):
    image = Image.open(os.path.join(opt.datadir, 'images', image_name)).convert("RGB")
    mask = Image.open(os.path.join(opt.datadir, 'masks', mask_name)).convert("L")
    mask_np = (np.array(mask) > 127).astype(np.float32)                   # demo.py:61-63
    image = Image.merge("RGBA", (*image.split(), mask))
    return preprocess_image(opt, image, bbox_from_mask(mask_np, 0.5))


def prepare_data(opt):
    names = sorted(n for n in os.listdir(os.path.join(opt.datadir, 'images')) if n.endswith(('.png', '.jpg')))
    f = 1.3875
    K = torch.tensor([[f * opt.W, 0, opt.W / 2], [0, f * opt.H, opt.H / 2], [0, 0, 1]]).float()
    data = []
    for i, name in enumerate(names):
        rgb, mask = get_image(opt, name, name[:-4] + '.png')
        data.append(edict(rgb_input_map=rgb.unsqueeze(0).to(opt.device), mask_input_map=mask.unsqueeze(0).to(opt.device),
                          intr=K.unsqueeze(0).to(opt.device), idx=torch.tensor([i + 1]).to(opt.device).long()))
    return data, [n[:-4] for n in names]


@torch.no_grad()
def marching_cubes(opt, var, impl_network):
    points_3D = get_dense_3D_grid(opt, var)                                # [B,G,G,G,3] (tagged: fused grid query)
    level_vox, _ = compute_level_grid(opt, impl_network, var.latent_depth, var.latent_semantic, points_3D,
                                      var.rgb_input_map, False)
    var.eval_vox = level_vox
    var.mesh_pred = convert_to_explicit(opt, list(level_vox), isoval=0.5, to_pointcloud=False)
    return var


def main():
    opt = options.set(opt_cmd=options.parse_arguments(sys.argv[1:]), safe_check=False)
    opt.device = "cuda:0"
    if os.path.basename(opt.yaml).split('.')[0] != opt.task:
        raise ValueError('Detected different tasks between specified and the yaml, please double check!')
    if opt.task == 'shape':
        opt.pretrain.depth = None
    opt.arch.depth.pretrained = None
    graph = importlib.import_module("model.compute_graph.graph_{}".format(opt.task)).Graph(opt).to(opt.device)
    checkpoint = torch.load(opt.ckpt, map_location="cpu")
    print("resuming from epoch {} (iteration {}, best_val {:.4f})".format(checkpoint["epoch"] + 1, checkpoint["iter"],
                                                                          checkpoint["best_val"]))
    graph.load_state_dict(checkpoint["graph"], strict=True)
    graph.eval()
    data_list, name_list = prepare_data(opt)
    save_folder = os.path.join(opt.datadir, 'preds')
    if os.path.isdir(save_folder):
        shutil.rmtree(save_folder)
    os.makedirs(save_folder)
    opt.output_path = opt.datadir
    for var, name in zip(data_list, name_list):
        with torch.no_grad():
            var = graph.forward(opt, var, training=False, get_loss=False)
            util_vis.dump_images(opt, [name], "image_input", var.rgb_input_map, from_range=(0, 1), folder='preds')
            util_vis.dump_images(opt, [name], "mask_input", var.mask_input_map, folder='preds')
            util_vis.dump_depths(opt, [name], "depth_est", var.depth_pred, var.mask_input_map, rescale=True, folder='preds')
            if opt.task == 'shape':
                var = marching_cubes(opt, var, graph.impl_network)
                util_vis.dump_meshes(opt, [name], "mesh", var.mesh_pred, folder='preds')
        print("{}: done".format(name))


if __name__ == "__main__":
    main()
