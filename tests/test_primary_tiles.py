"""The tiled enumeration of the primary rays (primary_tile / primary_padded in csrc/sq_device.hip) is a bijection between the
non-padding positions q and the shard's pixels: mirrored here in Python (the device function cannot run on the CPU), so that
an edit of one without the other shows up in review; the GPU parity tests prove coverage (a missing pixel would stay black),
this test also rules out duplicates, which parity cannot see (a pixel traced twice still gets the right colour)."""
import numpy as np
import pytest


def tile_rows_for(n_shards, row_block):
    rb = 8 if n_shards <= 1 else row_block
    return 8 if rb >= 8 and rb % 8 == 0 else 4 if rb >= 4 and rb % 4 == 0 else 2 if rb >= 2 and rb % 2 == 0 else 1


def enumerate_pixels(local_rows, h, tile_rows):
    tw = 64 // tile_rows
    tiles_x = (h + tw - 1) // tw
    padded = ((local_rows + tile_rows - 1) // tile_rows) * tiles_x * 64
    q = np.arange(padded, dtype=np.int64)
    lane, tile = q & 63, q >> 6
    ty, tx = tile // tiles_x, tile % tiles_x
    jj, xx = lane // tw, lane % tw
    j, x = ty * tile_rows + jj, tx * tw + xx
    ok = (j < local_rows) & (x < h)
    return np.where(ok, j * h + x, -1), padded


@pytest.mark.parametrize("local_rows,h,n_shards,row_block", [
    (1, 1, 1, 1), (7, 5, 1, 1), (64, 64, 1, 1), (1080, 1920, 1, 1), (135, 1080, 8, 2), (240, 1080, 8, 2), (9, 33, 3, 3), (12, 100, 2, 6),
    (16, 31, 2, 8), (32, 65, 4, 16), (5, 640, 5, 1), (48, 48, 2, 4), (1, 4097, 7, 1)])
def test_tiles_enumerate_every_pixel_once(local_rows, h, n_shards, row_block):
    tr = tile_rows_for(n_shards, row_block)
    assert 64 % tr == 0 and (n_shards <= 1 or row_block % tr == 0)       # a tile never straddles two row blocks of a shard
    pix, padded = enumerate_pixels(local_rows, h, tr)
    assert padded % 64 == 0 and padded >= local_rows * h
    real = pix[pix >= 0]
    assert len(real) == local_rows * h and np.array_equal(np.sort(real), np.arange(local_rows * h))
    # every wave of 64 positions is one tile: its pixels span at most tile_rows rows and 64 / tile_rows columns
    for w0 in range(0, min(padded, 64 * 50), 64):
        p = pix[w0:w0 + 64]; p = p[p >= 0]
        if len(p):
            rows, cols = p // h, p % h
            assert rows.max() - rows.min() < tr and cols.max() - cols.min() < 64 // tr
