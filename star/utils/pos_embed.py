from star_amd.utils.pos_embed import get_2d_sincos_pos_embed, get_1d_sincos_pos_embed_from_grid  # noqa: F401
