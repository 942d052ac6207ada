"""Fixed 2-D sin-cos position table (host-side, init only).  Mirrors the reference interface
star/utils/pos_embed.py:20-67 -- same function name, arguments and output bits after the float32 cast."""
import numpy as np


def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    assert embed_dim % 2 == 0
    freq = np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.0)
    angle = np.asarray(pos, dtype=np.float64).reshape(-1)[:, None] * (1.0 / 10000 ** freq)[None, :]
    return np.concatenate([np.sin(angle), np.cos(angle)], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    """-> [grid*grid (+1), embed_dim] float64.  The first half of the channels encodes the column index
    (the reference builds its meshgrid with w first), the second half the row index."""
    assert embed_dim % 2 == 0
    idx = np.arange(grid_size, dtype=np.float32)
    col, row = np.meshgrid(idx, idx)
    table = np.concatenate([get_1d_sincos_pos_embed_from_grid(embed_dim // 2, col),
                            get_1d_sincos_pos_embed_from_grid(embed_dim // 2, row)], axis=1)
    if cls_token:
        table = np.concatenate([np.zeros([1, embed_dim]), table], axis=0)
    return table
