"""Fixed 2-D sin-cos position table for the latent tokens (reference:
utils/pos_embed.py:21-68, used by model/shape/implicit.py:232-236).  Init-time only.

Layout facts that matter for checkpoint parity: the table is float64 until the caller
casts it; row 0 (cls token) is all zeros; the first half of the channels encodes the
column index ("w goes first" meshgrid), the second half the row index; within each
half, sines come before cosines; frequencies are 10000^(-i/(D/4)) computed in fp32.
"""
import numpy as np


def _encode_axis(dim, positions):
    """[M] positions -> [M, dim]: sin(p * w_i) | cos(p * w_i), w_i = 10000^(-i / (dim/2))."""
    freq = np.arange(dim // 2, dtype=np.float32)
    freq /= dim / 2.
    freq = 1. / 10000 ** freq
    angle = np.einsum('m,d->md', positions.reshape(-1), freq)
    return np.concatenate([np.sin(angle), np.cos(angle)], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    """-> [grid_size**2 (+1), embed_dim] float64."""
    assert embed_dim % 4 == 0
    coords = np.arange(grid_size, dtype=np.float32)
    col, row = np.meshgrid(coords, coords)          # col varies fastest
    table = np.concatenate([_encode_axis(embed_dim // 2, col), _encode_axis(embed_dim // 2, row)], axis=1)
    if cls_token:
        table = np.concatenate([np.zeros([1, embed_dim]), table], axis=0)
    return table
