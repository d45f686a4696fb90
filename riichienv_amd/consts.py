"""Tile-dimension constants under the reference's names (src/riichienv/consts.py)."""
N_TILE_TYPES_4P = 34   # 1-9m, 1-9p, 1-9s, four winds, three dragons
N_TILE_TYPES_3P = 27   # no 2m-8m
N_TILES_4P = 136
N_TILES_3P = 108
