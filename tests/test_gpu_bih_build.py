"""GPU BIH build (sq_bih_build_device) against the host build and the oracle's tree:
every array must be bit-identical (nodes in pre-order, triangles in leaf order, root box, height / leaves /
longest leaf), because tree shape and leaf order decide traversal tie-breaks (src/BIH.hs:62-99)."""
import os
import sys

import numpy as np
import pytest

from conftest import DATA, ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_scenes as G  # noqa: E402

pytestmark = pytest.mark.gpu


def _raw(a):
    return np.ascontiguousarray(a).view(np.uint8)


def _same(sqt, mesh):
    host = sqt.BIH(mesh)
    dev = sqt.BIH(mesh, device=0)
    assert (dev.height, dev.num_leaves, dev.longest_leaf) == (host.height, host.num_leaves, host.longest_leaf)
    assert np.array_equal(_raw(dev.bounds), _raw(host.bounds))
    assert dev.scene.n_nodes == host.scene.n_nodes and dev.scene.n_tris == host.scene.n_tris
    assert np.array_equal(_raw(dev.nodes), _raw(host.nodes))
    assert np.array_equal(_raw(dev.tris), _raw(host.tris))
    assert np.array_equal(_raw(dev.materials), _raw(host.materials))
    return host, dev


def _mesh(sqt, v, mat=None):
    v = np.asarray(v, np.float32).reshape(-1, 3, 3)
    tris = np.zeros(len(v), sqt._native.TRI_DTYPE)
    tris["v0"], tris["v1"], tris["v2"] = v[:, 0], v[:, 1], v[:, 2]
    if mat is not None:
        tris["mat"] = mat
    mats = np.zeros(2, sqt._native.MAT_DTYPE)
    mats["surf"] = [[0.5, 0.5, 0.5], [0.9, 0.1, 0.1]]
    return sqt.Mesh.from_arrays(tris, mats)


def test_scene_obj_tree_is_identical_and_matches_the_oracle(sqt, O, oracle_scene):
    mesh = sqt.Mesh.from_obj(os.path.join(DATA, "scene.obj"), DATA)
    host, dev = _same(sqt, mesh)
    ob = oracle_scene[0]
    kind, lmax, rmin, cnt = ob.preorder()
    nd = dev.nodes
    br = kind != 3
    assert np.array_equal(nd["kind"] & 3, kind)
    assert np.array_equal(nd["lmax"][br], lmax[br]) and np.array_equal(nd["rmin"][br], rmin[br])
    assert np.array_equal((nd["kind"] >> 2)[~br], cnt[~br])
    assert np.array_equal(dev.tris["v1"], ob.flatten()["b"])
    assert (dev.height, dev.num_leaves, dev.longest_leaf) == (13, 640, 14)


def test_rendering_from_the_device_built_tree_hits_the_golden_frame(sqt, product_scene):
    from conftest import GOLDEN
    mesh = product_scene[2]
    dev = sqt.BIH(mesh, device=0)
    want = np.load(os.path.join(GOLDEN, "scene_64x64_4spp_avg.npy"))
    got = sqt.render_f32(dev, product_scene[1], 4, (64, 64))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("n", [0, 1, 14, 15, 16, 63, 64, 65, 1000, 2049, 40000])
def test_random_soups_of_every_size_class(sqt, n):
    rng = np.random.default_rng(100 + n)
    v = rng.uniform(-2, 2, (n, 1, 3)) + rng.normal(0, 0.2, (n, 3, 3))
    _same(sqt, _mesh(sqt, v, rng.integers(0, 2, n)))


def test_terminal_branches_signed_zeros_and_ties(sqt):
    rng = np.random.default_rng(7)
    # identical triangles: every centroid equals the plane, nothing goes left (src/BIH.hs:70-72)
    one = rng.uniform(-1, 1, (1, 3, 3))
    host, _ = _same(sqt, _mesh(sqt, np.repeat(one, 40, 0)))
    assert host.longest_leaf == 40
    # two clusters far apart with identical members: proper split at the root, terminal branches below
    v = np.concatenate([np.repeat(one, 30, 0), np.repeat(one + 5.0, 30, 0)])
    host, _ = _same(sqt, _mesh(sqt, v))
    assert host.longest_leaf == 30
    # boxes whose extreme is a zero of either sign, in either order: min keeps the first, max the last
    for flip in (False, True):
        v = rng.uniform(-1, 0, (200, 3, 3)).astype(np.float32)
        v[::7, 0, :] = -0.0 if flip else 0.0
        v[3::7, 1, :] = 0.0 if flip else -0.0
        _same(sqt, _mesh(sqt, v))
        _same(sqt, _mesh(sqt, -v))
    # equal extents on two axes (longestAxis keeps the later one), grid-aligned centroids (ties at the plane)
    g = np.stack(np.meshgrid(np.arange(12.0), np.arange(12.0), np.arange(3.0), indexing="ij"), -1).reshape(-1, 1, 3)
    v = g + np.array([[0, 0, 0], [0.5, 0, 0], [0, 0.5, 0]])[None]
    _same(sqt, _mesh(sqt, v))
    # huge and tiny magnitudes (centroid sums that overflow to inf, denormals)
    v = rng.uniform(-1, 1, (300, 3, 3)).astype(np.float32)
    v[:100] *= np.float32(3.0e38)
    v[100:200] *= np.float32(1.0e-41)
    _same(sqt, _mesh(sqt, v))


def test_generated_scenes(sqt):
    for obj, sq, _ in (G.blob_scene(4), G.heightfield_scene(60)):
        _same(sqt, sqt.Mesh.from_text(obj, sq))


def test_million_triangle_heightfield(sqt):
    obj, sq, _ = G.heightfield_scene(708)
    host, dev = _same(sqt, sqt.Mesh.from_text(obj, sq))
    assert dev.scene.n_tris > 1000000


def test_refuses_what_it_cannot_build_exactly(sqt):
    v = np.random.default_rng(3).uniform(-1, 1, (50, 3, 3)).astype(np.float32)
    v[17, 1, 2] = np.nan
    with pytest.raises(sqt.SquiglyError, match="finite"):
        sqt.BIH(_mesh(sqt, v), device=0)
    v[17, 1, 2] = np.inf
    with pytest.raises(sqt.SquiglyError, match="finite"):
        sqt.BIH(_mesh(sqt, v), device=0)
    with pytest.raises(sqt.SquiglyError, match="device"):
        sqt.BIH(_mesh(sqt, v[:5]), device=99)
