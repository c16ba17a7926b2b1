"""CPU tests of the product's host side (loaders, BIH build/flatten, C-ABI surface) against the oracle."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import DATA, ROOT


def test_library_exports_every_declared_symbol(sqt):
    """Every function declared in include/*.h is exported by libsquigly_hip.so (no compute calls)."""
    declared = set()
    for hdr in ("squigly_hip.h", "squigly_host.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b(sq_[a-z0-9_]+)\s*\(", text))
    assert declared == set(sqt.EXPORTED_SYMBOLS)
    nm = subprocess.check_output(["nm", "-D", "--defined-only", sqt.LIB_PATH]).decode()
    exported = set(re.findall(r" T (sq_[a-z0-9_]+)", nm))
    assert declared <= exported, declared - exported
    L = sqt.lib()
    for name in declared:
        assert getattr(L, name) is not None
    assert L.sq_abi_version() == 1


def test_loader_matches_oracle(sqt, O, product_scene, oracle_scene):
    _, _, mesh = product_scene
    _, _, otris = oracle_scene
    t = mesh.tris
    assert len(t) == len(otris) == 6238
    assert np.array_equal(t["v0"], otris["a"]) and np.array_equal(t["v1"], otris["b"]) and np.array_equal(t["v2"], otris["c"])
    mats = mesh.materials
    assert np.array_equal(mats["reflective"][t["mat"]], otris["reflective"])
    assert np.array_equal(mats["surf"][t["mat"]], otris["surf"])
    assert np.array_equal(mats["emissive"][t["mat"]], otris["emissive"]) and np.array_equal(mats["emit"][t["mat"]], otris["emit"])


def test_bih_build_matches_oracle_tree(sqt, O, product_scene, oracle_scene):
    bih, cam, _ = product_scene
    ob, ocam, _ = oracle_scene
    assert (bih.height, bih.num_leaves, bih.longest_leaf) == (ob.height, ob.num_leaves, ob.longest_leaf) == (13, 640, 14)
    kind, lmax, rmin, cnt = ob.preorder()
    nd = bih.nodes
    assert np.array_equal(nd["kind"] & 3, kind)
    br = kind != 3
    assert np.array_equal(nd["lmax"][br], lmax[br]) and np.array_equal(nd["rmin"][br], rmin[br])
    assert np.array_equal((nd["kind"] >> 2)[~br], cnt[~br])
    flat = ob.flatten()
    pt = bih.tris
    assert np.array_equal(pt["v0"], flat["a"]) and np.array_equal(pt["v1"], flat["b"]) and np.array_equal(pt["v2"], flat["c"])
    assert np.array_equal(bih.bounds, ob.bounds())
    # leaves index the flattened triangles contiguously, in order
    firsts = nd["link"][~br]
    assert firsts[0] == 0 and np.array_equal(firsts[1:], np.cumsum(cnt[~br])[:-1])
    assert np.array_equal(np.array(list(cam.rot), np.float32), O.camera_arrays(ocam)[1])
    assert list(cam.pos) == [0.0, 7.0, 0.75]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_bih_build_random_soups_match_oracle(sqt, O, seed):
    """Random triangle soups incl. duplicates (equal centroids -> empty sides) and ragged sizes."""
    rng = np.random.default_rng(seed)
    n = [14, 15, 400][seed]
    v = rng.uniform(-3, 3, (n, 3, 3)).astype(np.float32)
    if seed == 2:
        v[50:90] = v[50]                       # 40 identical triangles
        v[:, :, 2] = np.round(v[:, :, 2])      # many exactly-equal coordinates: exercises longest-axis ties
    tris = np.zeros(n, sqt._native.TRI_DTYPE)
    tris["v0"], tris["v1"], tris["v2"] = v[:, 0], v[:, 1], v[:, 2]
    mats = np.zeros(1, sqt._native.MAT_DTYPE)
    bih = sqt.BIH(sqt.Mesh.from_arrays(tris, mats))
    ot = np.zeros(n, O.TRI_DTYPE)
    ot["a"], ot["b"], ot["c"] = v[:, 0], v[:, 1], v[:, 2]
    ob = O.BIH(ot)
    kind, lmax, rmin, cnt = ob.preorder()
    nd = bih.nodes
    assert len(nd) == len(kind) and np.array_equal(nd["kind"] & 3, kind)
    br = kind != 3
    assert np.array_equal(nd["lmax"][br], lmax[br]) and np.array_equal(nd["rmin"][br], rmin[br])
    assert np.array_equal((nd["kind"] >> 2)[~br], cnt[~br])
    assert np.array_equal(bih.tris["v0"], ob.flatten()["a"])
    assert (bih.height, bih.num_leaves, bih.longest_leaf) == (ob.height, ob.num_leaves, ob.longest_leaf)


def test_loader_edge_cases_match_oracle(sqt, O):
    sq = b"newmtl A\nreflective 0 1 1 1\nemissive 0 0 0 0\n\nnewmtl B\nreflective 1 0.5 0.5 0.5\nemissive 2 1 1 1\n"
    obj = (b"mtllib s.sq\r\no Cube.001_x\r\nv 0 0 0\r\nv 1 0 0\r\nv 0 1 0\r\nusemtl A\r\ns off\r\nf 1 2 3\r\n"
           b"o Second\nv 0 0 1\nusemtl B\ns on\nf 1 2 4\nf 4 2 1\n")
    cases = [(obj, sq), (obj.replace(b"usemtl B", b"usemtl C"), sq),
             (obj, sq + b"newmtl A\nreflective 0 0 0 0\nemissive 0 0 0 0\n"), (b"mtllib s.sq\n", sq),
             (obj, b" " + sq)]           # leading blank: `many loadMtl` stops at once -> no materials -> no triangles
    for o, s in cases:
        pt, ot = sqt.Mesh.from_text(o, s), O.tris_from_text(o, s)
        assert len(pt) == len(ot)
        if len(ot):
            t, m = pt.tris, pt.materials
            assert np.array_equal(t["v0"], ot["a"]) and np.array_equal(t["v2"], ot["c"])
            assert np.array_equal(m["reflective"][t["mat"]], ot["reflective"]) and np.array_equal(m["emit"][t["mat"]], ot["emit"])
    bad = [b"o X\n", b"mtllib s.sq\no X\nv 0 0 0\nf 1 1 1\n", b"mtllib s.sq\no X\nv 0 0 0\nusemtl A\nf 1/1 1 1\n",
           b"mtllib s.sq\no X\nv 0 0 0\nusemtl A\nf 1 1 2\n", b"mtllib s.sq\no X\nv 1e-3 0 0\nusemtl A\nf 1 1 1\n",
           b"mtllib s.sq\no X\nv .5 0 0\nusemtl A\nf 1 1 1\n", b"mtllib s.sq\no X\nv 0 0 0\nusemtl A\ns 1\nf 1 1 1\n",
           b"mtllib s.sq\no X\nv 0 0 0\nusemtl A\nf 0 1 1\n"]
    for o in bad:
        with pytest.raises(sqt.SquiglyError):
            sqt.Mesh.from_text(o, sq)
        with pytest.raises(O.OracleError):
            O.tris_from_text(o, sq)
    with pytest.raises(sqt.SquiglyError):
        sqt.Mesh.from_obj("/nonexistent.obj", DATA)
    with pytest.raises(sqt.SquiglyError):
        sqt.camera_from_text(b"0 7\n")
    for ang in [(0.3, -1.2, 2.5), (0, 0, 0), (1.5707963267948966, 0, -0.09817477042468103), (100.0, -37.5, 6.25)]:
        r = (C.c_float * 9)()
        O.lib().sqo_rot_matrix_rads(*ang, O.TRIG_CRD, r)
        assert np.array_equal(sqt.rot_matrix_rads(*ang).ravel(), np.array(list(r), np.float32))


def test_render_without_gpu_fails_loudly_not_silently(sqt, product_scene):
    """No CPU fallback: without a HIP device the render entry point returns an error."""
    if sqt.device_count() > 0:
        pytest.skip("a GPU is present")
    bih, cam, mesh = product_scene
    with pytest.raises(sqt.SquiglyError, match="no HIP device|no CPU fallback"):
        sqt.render_rgb8(bih, cam, 1, (4, 4))
    with pytest.raises(sqt.SquiglyError, match="no HIP device"):
        sqt.BIH(mesh, device=0)                      # the GPU build does not quietly become the host build


def test_shard_helpers(sqt):
    from importlib import import_module
    d = import_module("squigly-trace_amd.dist")
    L = sqt.lib()
    for w, rb, world in [(1920, 8, 8), (1080, 8, 4), (37, 8, 3), (5, 8, 2), (64, 1, 64), (100, 16, 1)]:
        seen = []
        for r in range(world):
            sh = sqt.Shard(rb, r, world)
            rows = d.shard_rows(w, rb, r, world)
            assert L.sq_shard_rows(w, sh) == len(rows)
            assert [L.sq_shard_global_row(j, sh) for j in range(len(rows))] == rows
            seen += rows
        assert sorted(seen) == list(range(w))
    assert L.sq_shard_rows(10, sqt.Shard(0, 0, 1)) == -1 and L.sq_shard_rows(10, sqt.Shard(4, 2, 2)) == -1


def test_product_does_not_touch_the_oracle():
    """The product path may not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "squigly-trace_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dp, f), errors="replace").read()
                assert "pyoracle" not in text and "sqo_" not in text and "sq_oracle" not in text, os.path.join(dp, f)
                assert "/root/reference" not in text
    ldd = subprocess.check_output(["ldd", os.path.join(pkg, "libsquigly_hip.so")]).decode()
    assert "oracle" not in ldd
    # the measurement aids under tools/ stay away from it too: checker scripts live under tests/
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith((".py", ".sh")):
            text = open(os.path.join(ROOT, "tools", f), errors="replace").read()
            assert "pyoracle" not in text and "sqo_" not in text, f


def test_hit_distance_is_monotone_in_t():
    """The trace kernel compares two hits of one ray by t first and evaluates dist = norm((o + t*d) - o) only when
    t does not decide (csrc/sq_scene.h: dist_gt).  That rests on dist being monotone non-decreasing in t under
    round-to-nearest fp32; check it on adjacent and random t pairs, including tiny, huge and mixed-sign inputs."""
    rng = np.random.default_rng(3)
    f = np.float32

    def dist(o, d, t):
        p = o + t[:, None] * d
        v = p - o
        return np.sqrt((v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2], dtype=np.float32)

    n = 400000
    scale = (10.0 ** rng.uniform(-6, 6, (n, 1))).astype(f)
    o = (rng.normal(size=(n, 3)) * scale).astype(f)
    d = (rng.normal(size=(n, 3)) * (10.0 ** rng.uniform(-3, 3, (n, 1)))).astype(f)
    d[rng.random(n) < 0.1, 0] = 0
    d[rng.random(n) < 0.1, 1] = -0.0
    t1 = (10.0 ** rng.uniform(-4, 4, n)).astype(f)
    for t2 in (np.nextafter(t1, f(np.inf)), (t1 * (1 + 10.0 ** rng.uniform(-7, 0, n))).astype(f), (t1 + f(1e-4)).astype(f)):
        with np.errstate(over="ignore", invalid="ignore"):
            d1, d2 = dist(o, d, t1), dist(o, d, t2.astype(f))
        assert np.all(t2 >= t1)
        assert np.all(d2 >= d1), int((d2 < d1).sum())


def test_loader_differential_fuzz_slice(sqt, O):
    """A slice of tests/fuzz_loader.py: random .obj / .sq / camera texts from the reference grammar plus byte
    mutations; the product loader and the oracle's accept and reject the same inputs and agree bit for bit."""
    import fuzz_loader
    failures = [(seed, msg) for seed in range(5000, 7000) if (msg := fuzz_loader.run_case(seed))]
    assert not failures, failures[:3]


def test_bih_build_matches_oracle_on_fuzz_scenes(sqt, O):
    """The host BIH build against the oracle's tree on the scene generators of tests/fuzz_gpu.py (ties, duplicates,
    slivers, overflowing magnitudes, NaN / infinite coordinates, long identical-triangle leaves): every node and the
    leaf order, bit for bit."""
    import fuzz_gpu as F
    for seed in range(3000, 3400):
        rng = np.random.default_rng(seed)
        v, mats, mat, _ = F.make_scene(rng)
        tris = np.zeros(len(v), sqt._native.TRI_DTYPE)
        tris["v0"], tris["v1"], tris["v2"], tris["mat"] = v[:, 0], v[:, 1], v[:, 2], mat
        bih = sqt.BIH(sqt.Mesh.from_arrays(tris, mats))
        ot = np.zeros(len(v), O.TRI_DTYPE)
        ot["a"], ot["b"], ot["c"] = v[:, 0], v[:, 1], v[:, 2]
        for f in ("reflective", "surf", "emissive", "emit"):
            ot[f] = mats[f][mat]
        ob = O.BIH(ot)
        kind, lmax, rmin, cnt = ob.preorder()
        nd = bih.nodes
        br = kind != 3
        u = lambda a: np.ascontiguousarray(a).view(np.uint32)
        assert len(nd) == len(kind) and np.array_equal(nd["kind"] & 3, kind), seed
        assert np.array_equal(u(nd["lmax"][br]), u(lmax[br])) and np.array_equal(u(nd["rmin"][br]), u(rmin[br])), seed
        assert np.array_equal((nd["kind"] >> 2)[~br], cnt[~br]), seed
        fl = ob.flatten()
        assert np.array_equal(u(bih.tris["v0"]), u(fl["a"])) and np.array_equal(u(bih.tris["v1"]), u(fl["b"])) and np.array_equal(u(bih.tris["v2"]), u(fl["c"])), seed
        assert np.array_equal(u(bih.bounds), u(np.array(ob.bounds(), np.float32).ravel())), seed
