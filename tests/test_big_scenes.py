"""BASELINE configs[2] and configs[4] stand-ins (procedural; tools/gen_scenes.py): loader and BIH build at
scale on the CPU, and GPU parity of the streaming trace kernel (scene too large for LDS, 32-bit stack words,
top-of-tree node prefix in LDS) against the oracle."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_scenes as G  # noqa: E402


def _same_tree(sqt, O, obj, sq):
    mesh = sqt.Mesh.from_text(obj, sq)
    bih = sqt.BIH(mesh)
    ob = O.BIH(O.tris_from_text(obj, sq))
    kind, lmax, rmin, cnt = ob.preorder()
    nd = bih.nodes
    br = kind != 3
    assert np.array_equal(nd["kind"] & 3, kind)
    assert np.array_equal(nd["lmax"][br], lmax[br]) and np.array_equal(nd["rmin"][br], rmin[br])
    assert np.array_equal((nd["kind"] >> 2)[~br], cnt[~br])
    assert np.array_equal(bih.tris["v0"], ob.flatten()["a"]) and np.array_equal(bih.tris["v2"], ob.flatten()["c"])
    assert (bih.height, bih.num_leaves, bih.longest_leaf) == (ob.height, ob.num_leaves, ob.longest_leaf)
    return bih, ob


def test_generated_scenes_load_and_build_like_the_oracle(sqt, O):
    for obj, sq, _ in (G.blob_scene(4), G.heightfield_scene(60)):
        bih, ob = _same_tree(sqt, O, obj, sq)
        assert bih.scene.n_tris == ob.n_tris > 5000


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["blob6", "heightfield708"])
def test_streaming_kernel_parity_on_large_scenes(sqt, O, which):
    import torch
    obj, sq, camt = G.blob_scene(6) if which == "blob6" else G.heightfield_scene(708)
    bih, ob = _same_tree(sqt, O, obj, sq)
    assert bih.scene.n_tris > 80000
    cam, ocam = sqt.camera_from_text(camt), O.camera_from_text(camt)
    ds = sqt.DeviceScene(bih, 0)
    w, h, n = 56, 40, 3
    o, o8, _ = ob.render(ocam, n, w, h, threads=min(os.cpu_count() or 1, 16))
    for variant in (2, 1):
        ds.set_option("variant", variant)
        a, r = ds.render_rows(cam, n, w, h)
        torch.cuda.synchronize()
        assert np.array_equal(a.cpu().numpy().view(np.uint32), o.view(np.uint32)), (which, variant)
        assert np.array_equal(r.cpu().numpy(), o8)
    # cast mode (shadow rays) too
    oc, _, _ = ob.render(ocam, 1, w, h, cast=True, threads=min(os.cpu_count() or 1, 16))
    a, _ = ds.render_rows(cam, 1, w, h, cast=True, want_rgb=False)
    torch.cuda.synchronize()
    assert np.array_equal(a.cpu().numpy().view(np.uint32), oc.view(np.uint32))
    ds.close()
