"""Minimal result dumps for demo.py (the reference's utils/util_vis.py renders with pyrender / cv2 /
trimesh, which this image does not have and the hot path does not need): input image and mask as
PNG, depth maps as 16-bit PNG, meshes as Wavefront OBJ with welded vertices, attention maps as .npy.
File names follow the reference's `<output_path>/<folder>/<idx>_<name>.<ext>` pattern."""
import os

import numpy as np


def _path(opt, folder, idx, name, ext):
    d = os.path.join(opt.output_path, folder)
    os.makedirs(d, exist_ok=True)
    return os.path.join(d, "{}_{}.{}".format(idx, name, ext))


def dump_images(opt, idx, name, images, masks=None, from_range=(0, 1), folder="dump", **_):
    """images [B,C,H,W] (C = 1 or 3) in `from_range` -> 8-bit PNG (with `masks` as alpha)."""
    from PIL import Image
    lo, hi = from_range
    x = ((images.detach().float().cpu().numpy() - lo) / (hi - lo)).clip(0, 1)
    for b, i in enumerate(idx):
        img = (x[b].transpose(1, 2, 0) * 255 + 0.5).astype(np.uint8)
        if img.shape[2] == 1:
            img = img[:, :, 0]
        im = Image.fromarray(img)
        if masks is not None:
            im.putalpha(Image.fromarray((masks[b, 0].detach().cpu().numpy() * 255).astype(np.uint8)))
        im.save(_path(opt, folder, i, name, "png"))


def dump_depths(opt, idx, name, depths, masks=None, rescale=False, folder="dump", **_):
    """depths [B,1,H,W] -> 16-bit PNG; with `rescale` the masked range is stretched to full scale."""
    from PIL import Image
    d = depths.detach().float().cpu().numpy()[:, 0]
    for b, i in enumerate(idx):
        x = d[b]
        if masks is not None:
            m = masks[b, 0].detach().cpu().numpy() > 0.5
            if rescale and m.any():
                lo, hi = x[m].min(), x[m].max()
                x = (x - lo) / max(hi - lo, 1e-8)
            x = np.where(m, x, 1.0)
        Image.fromarray((x.clip(0, 1) * 65535 + 0.5).astype(np.uint16)).save(_path(opt, folder, i, name, "png"))


def dump_meshes(opt, idx, name, meshes, folder="dump", **_):
    """meshes: eval_3D.SimpleMesh objects -> OBJ with shared vertices (the indexed form mcubes.marching_cubes returns in
    the reference; the marching-cubes kernel emits bit-identical coordinates for shared vertices)."""
    for i, mesh in zip(idx, meshes):
        verts, faces = mesh.vertices, mesh.faces + 1              # the indexed form (eval_3D.SimpleMesh welds the soup)
        with open(_path(opt, folder, i, name, "obj"), "w") as f:
            f.write("# zeroshape_amd mesh: %d vertices, %d faces\n" % (len(verts), len(faces)))
            for v in verts:
                f.write("v %.6f %.6f %.6f\n" % tuple(v))
            for t in faces:
                f.write("f %d %d %d\n" % tuple(t))


def dump_attentions(opt, idx, name, attn, folder="dump", **_):
    """The reference colour-maps attention over the image with cv2; here the raw maps are stored."""
    if attn is None:
        return
    for b, i in enumerate(idx):
        a = attn[b] if not isinstance(attn, (list, tuple)) else attn[b]
        np.save(_path(opt, folder, i, name, "npy"), np.asarray(a.detach().cpu() if hasattr(a, "detach") else a))
