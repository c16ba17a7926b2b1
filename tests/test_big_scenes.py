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


# Full-size frames of BASELINE configs[2] and configs[4] (their procedural stand-ins).  The oracle needs ~10 s per
# full-spp row of these scenes on 16 cores (it tests every triangle of every leaf the reference visits: 1100-1600
# tests per bounce ray), so FOUR rows spread over the frame are compared bit for bit; everything else about the
# frame is checked GPU against GPU: culling on = culling off, a second render = the first, and eight interleaved
# row shards (what eight ranks render) reassemble to the one-GPU frame.
FULL = {"blob6": (lambda: G.blob_scene(6), 512, (97, 803, 1211, 1790)),
        "heightfield708": (lambda: G.heightfield_scene(708), 256, (160, 741, 1302, 1874))}


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["blob6", "heightfield708"])
def test_full_size_frame_of_the_large_scenes(sqt, O, which):
    import torch
    from importlib import import_module
    d = import_module("squigly-trace_amd.dist")
    make, spp, rows = FULL[which]
    obj, sq, camt = make()
    mesh = sqt.Mesh.from_text(obj, sq)
    bih = sqt.BIH(mesh, device=0)                      # the device build, as the bench and the CLI use for >= 50 000 triangles
    ob = O.BIH(O.tris_from_text(obj, sq))
    assert np.array_equal(bih.tris["v0"], ob.flatten()["a"]) and bih.height == ob.height
    cam, ocam = sqt.camera_from_text(camt), O.camera_from_text(camt)
    w, h = 1920, 1080
    ds = sqt.DeviceScene(bih, 0)
    avg, rgb = ds.render_rows(cam, spp, w, h)
    torch.cuda.synchronize()
    threads = min(os.cpu_count() or 1, 16)
    sel = torch.tensor(rows, device=avg.device)
    g, g8 = avg[sel].cpu().numpy(), rgb[sel].cpu().numpy()
    for i, y in enumerate(rows):                       # one oracle call per row: rows=(y, y + 1)
        o, o8, _ = ob.render(ocam, spp, w, h, threads=threads, rows=(y, y + 1))
        assert np.array_equal(g[i].view(np.uint32), o[0].view(np.uint32)), \
            f"{which} row {y}: {int((g[i].view(np.uint32) != o[0].view(np.uint32)).any(-1).sum())} of {h} pixels differ"
        assert np.array_equal(g8[i], o8[0])
    assert int((rgb.sum(-1) > 0).sum()) > w * h // 4   # a real picture, not a black frame that happens to match
    # culling boxes off: the reference's every visit is made; same bits
    ds.set_option("cull", 0)
    a0, r0 = ds.render_rows(cam, spp, w, h)
    torch.cuda.synchronize()
    assert torch.equal(a0.view(torch.int32), avg.view(torch.int32)) and torch.equal(r0, rgb)
    ds.set_option("cull", 1)
    # what eight ranks render: interleaved blocks of d.ROW_BLOCK rows, reassembled
    frame = torch.zeros_like(avg)
    for r in range(8):
        a, _ = ds.render_rows(cam, spp, w, h, shard=(d.ROW_BLOCK, r, 8), want_rgb=False)
        frame[torch.tensor(d.shard_rows(w, d.ROW_BLOCK, r, 8), device=a.device)] = a
    torch.cuda.synchronize()
    assert torch.equal(frame.view(torch.int32), avg.view(torch.int32))
    ds.close()
    sqt.release_cached_memory()
