"""world_size-2 `gloo` test of the multi-GPU driver on CPU: shard -> render rows -> all_gather -> de-interleave.

The product renders only on a GPU, so each rank's rows are produced here by the oracle (as a stand-in
renderer inside the test only); what is under test is the sharding/gather logic of squigly-trace_amd/dist.py,
which must reassemble exactly the unsharded frame.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import DATA, ROOT


def _worker(rank, world, port, w, h, n, row_block, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import importlib
    import pyoracle as O
    d = importlib.import_module("squigly-trace_amd.dist")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ob = O.BIH(O.tris_from_obj(os.path.join(DATA, "scene.obj"), DATA))
        cam = O.load_camera(os.path.join(DATA, "camera"))
        rows = d.shard_rows(w, row_block, rank, world)
        local = np.zeros((len(rows), h, 3), np.float32)
        for j, y in enumerate(rows):
            local[j] = ob.render(cam, n, w, h, rows=(y, y + 1), want_rgb=False)[0][0]
        frame = d.gather_frame(torch.from_numpy(local), w, row_block)
        np.save(os.path.join(out_dir, f"frame_{rank}.npy"), frame.numpy())
        # the RGB8 frame of the bench (uint8, ragged shards padded to the largest) through the same cached plan, twice
        tag = (np.arange(len(rows) * h * 3, dtype=np.int64).reshape(len(rows), h, 3) * 7 + np.array(rows)[:, None, None]).astype(np.uint8)
        for _ in range(2):
            f8 = d.gather_frame(torch.from_numpy(tag), w, row_block)
        np.save(os.path.join(out_dir, f"tag_{rank}.npy"), f8.numpy())
        assert d.plan_for(w, row_block, world, torch.device("cpu")) is d.plan_for(w, row_block, world, torch.device("cpu"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h,row_block", [(2, 21, 16, 8), (2, 16, 12, 2), (3, 19, 8, 4)])
def test_gather_reassembles_the_frame(tmp_path, O, oracle_scene, world, w, h, row_block):
    n = 2
    port = 29500 + (os.getpid() % 2000) + w + 7 * world
    mp.spawn(_worker, args=(world, port, w, h, n, row_block, str(tmp_path)), nprocs=world, join=True)
    ob, cam, _ = oracle_scene
    full, _, _ = ob.render(cam, n, w, h, threads=2, want_rgb=False)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"frame_{r}.npy"))
        assert np.array_equal(got.view(np.uint32), full.view(np.uint32))
        # uint8 frame: rebuild what every rank sent and compare
        import importlib
        d = importlib.import_module("squigly-trace_amd.dist")
        want = np.zeros((w, h, 3), np.uint8)
        for q in range(world):
            rows = d.shard_rows(w, row_block, q, world)
            want[rows] = (np.arange(len(rows) * h * 3, dtype=np.int64).reshape(len(rows), h, 3) * 7 + np.array(rows)[:, None, None]).astype(np.uint8)
        assert np.array_equal(np.load(os.path.join(str(tmp_path), f"tag_{r}.npy")), want)
