"""CPU tests of the oracle itself: known-answer vectors, golden fixtures, internal cross-checks.

The reference has no tests (test/Spec.hs:1-2 is a stub) and cannot be run here, so the pins are:
public Threefish-256 KATs, the reference's own --debug statistics, the
reference's intended BIH-vs-naive cross-check (app/Main.hs:52-53), and render/example.png statistics.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import DATA, GOLDEN

STATS = json.load(open(os.path.join(GOLDEN, "scene_stats.json")))


def test_threefish_known_answers(O):
    for kat in STATS["threefish_kat"]:
        ct = O.threefish256(kat["key"], kat["tweak"], kat["pt"])
        assert ["%016X" % v for v in ct] == kat["ct_hex"]


def test_tfgen_words_golden_and_layout(O):
    for seed, words in STATS["tfgen_words"].items():
        assert O.tfgen_words(int(seed)) == words
    # block = E_{(seed,0,0,0)}(0): low half of each 64-bit word first
    ct = O.threefish256([7, 0, 0, 0], [0, 0], [0, 0, 0, 0])
    w = O.tfgen_words(7)
    assert w == [x for c in ct for x in (c & 0xFFFFFFFF, c >> 32)]
    # variant bit0 swaps halves (SURVEY App. B candidate)
    assert O.tfgen_words(7, 1) == [x for c in ct for x in (c >> 32, c & 0xFFFFFFFF)]


def test_bih_shape_matches_reference_debug_stats(oracle_scene):
    bih, _, tris = oracle_scene
    # app/Main.hs:72-74 prints these three; SURVEY.md App. C: 13 / 14 / 640
    assert (bih.height, bih.longest_leaf, bih.num_leaves) == (13, 14, 640)
    assert len(tris) == 6238 and bih.n_nodes == 1279
    kind, a, b, c = bih.preorder()
    flat = bih.flatten()
    h1 = hashlib.sha256(np.ascontiguousarray(np.stack([flat["a"], flat["b"], flat["c"]], 1)).tobytes()).hexdigest()
    h2 = hashlib.sha256(kind.tobytes() + a[kind != 3].tobytes() + b[kind != 3].tobytes() + c.tobytes()).hexdigest()
    assert h1 == STATS["sha256_flatten_vertices"] and h2 == STATS["sha256_preorder"]
    assert c[kind == 3].sum() == 6238 and (c[kind == 3] < 15).all()


def test_bih_agrees_with_naive_intersector(O, oracle_scene):
    """The reference's own intended cross-check (naiveIntersect, src/Geometry.hs:110-115)."""
    bih, cam, _ = oracle_scene
    rng = np.random.default_rng(1)
    n_hit = 0
    for i in range(600):
        if i % 2 == 0:      # camera rays
            o, d = O.make_ray(64, 64, int(rng.integers(64)), int(rng.integers(64)), cam)
        else:               # rays from inside the room
            o = rng.uniform(-1.5, 1.5, 3).astype(np.float32)
            d = rng.normal(size=3).astype(np.float32)
        hb, hn = bih.intersect(o, d), bih.intersect_naive(o, d)
        assert hb.hit == hn.hit
        if hb.hit:
            n_hit += 1
            # BIH pruning may pick a different triangle only at exactly equal distance
            assert hb.dist == hn.dist
    assert n_hit > 300


def test_render_matches_golden(oracle_scene):
    bih, cam, _ = oracle_scene
    avg, rgb, cnt = bih.render(cam, 4, 64, 64, threads=4)
    assert np.array_equal(avg.view(np.uint32), np.load(os.path.join(GOLDEN, "scene_64x64_4spp_avg.npy")).view(np.uint32))
    assert np.array_equal(rgb, np.load(os.path.join(GOLDEN, "scene_64x64_4spp_rgb8.npy")))
    assert cnt == STATS["counters_64x64_4spp"]
    avg2, _, _ = bih.render(cam, 3, 40, 72, threads=4)
    assert avg2.shape == (40, 72, 3)      # w ROWS x h COLUMNS (massiv quirk, src/Lib.hs:70-71)
    assert np.array_equal(avg2.view(np.uint32), np.load(os.path.join(GOLDEN, "scene_40x72_3spp_avg.npy")).view(np.uint32))


def test_thread_count_and_row_ranges_do_not_change_pixels(oracle_scene):
    bih, cam, _ = oracle_scene
    a1, _, _ = bih.render(cam, 2, 24, 24, threads=1)
    a4, _, _ = bih.render(cam, 2, 24, 24, threads=4)
    assert np.array_equal(a1, a4)
    part, _, _ = bih.render(cam, 2, 24, 24, threads=2, rows=(5, 17))
    assert np.array_equal(part, a1[5:17])
    strided, _, _ = bih.render(cam, 2, 24, 24, threads=2, rows=(1, 24), row_step=4)
    assert np.array_equal(strided, a1[1::4])


def test_pixel_is_ordered_mean_of_sample_radiances(oracle_scene):
    bih, cam, _ = oracle_scene
    gold = np.load(os.path.join(GOLDEN, "scene_64x64_4spp_samples.npy"))
    avg = np.load(os.path.join(GOLDEN, "scene_64x64_4spp_avg.npy"))
    for i, (y, x) in enumerate(STATS["sample_pixels"]):
        rad = np.array([bih.sample_radiance(cam, 4, 64, 64, y, x, k) for k in range(4)], np.float32)
        assert np.array_equal(rad, gold[i])
        s = np.zeros(3, np.float32)
        for k in range(4):
            s = s + rad[k]                                  # foldl (+) 0, src/Lib.hs:88
        assert np.array_equal(np.float32(1) / np.float32(4) * s, avg[y, x])


def test_cast_mode_golden(oracle_scene):
    bih, cam, _ = oracle_scene
    avg, rgb, _ = bih.render(cam, 2, 64, 64, cast=True, threads=4)
    assert np.array_equal(rgb, np.load(os.path.join(GOLDEN, "scene_64x64_cast_rgb8.npy")))
    assert np.array_equal(avg, np.load(os.path.join(GOLDEN, "scene_64x64_cast_avg.npy")))
    # cast is RNG-free: spp only changes rounding of the mean
    a1, _, _ = bih.render(cam, 1, 64, 64, cast=True, threads=4)
    assert np.allclose(a1, avg, rtol=1e-6, atol=0)


def test_tonemap_edge_cases(O):
    assert O.tonemap((0, 0, 0)) == (0, 0, 0)                 # 0/0 = NaN -> floor -> Word8 0
    assert O.tonemap((100, 100, 100)) == (253, 253, 253)     # emitter seen directly; example.png max is 253
    r = O.tonemap((1.0, 0.5, 0.25))
    assert r[0] > r[1] > r[2] > 0 and r[0] == int(np.floor(np.arctan(0.625) / (np.pi / 2) * 255))
    assert O.tonemap((float("inf"), 1, 1)) == (0, 0, 0) or True   # must not crash


def test_slab_and_mt_edge_cases(O):
    import ctypes as C
    b = O.Bounds(O.V3(-1, -1, -1), O.V3(1, 1, 1))
    L = O.lib()
    assert L.sqo_intersects_bb(C.byref(b), O.V3(0, 0, -5), O.V3(0, 0, 1)) == 1     # zero components: 1/0 = inf
    assert L.sqo_intersects_bb(C.byref(b), O.V3(0, 0, 5), O.V3(0, 0, 1)) == 0      # box behind the ray
    assert L.sqo_intersects_bb(C.byref(b), O.V3(2, 0, -5), O.V3(0, 0, 1)) == 0
    # origin exactly on a slab plane with zero direction: 0 * inf = NaN goes through Haskell's min/max rules
    assert L.sqo_intersects_bb(C.byref(b), O.V3(1, 0, -5), O.V3(0, 0, 1)) in (0, 1)
    nan = float("nan")
    assert L.sqo_intersects_bb(C.byref(b), O.V3(0, 0, -5), O.V3(nan, 0, 1)) in (0, 1)
    tri = O.Triangle(O.V3(0, 0, 0), O.V3(1, 0, 0), O.V3(0, 1, 0), O.Material())
    p, d = O.V3(), C.c_float()
    hit = L.sqo_moller_trumbore(O.V3(0.25, 0.25, 1), O.V3(0, 0, -1), C.byref(tri), C.byref(p), C.byref(d))
    assert hit == 1 and d.value == 1.0 and (p.x, p.y, p.z) == (0.25, 0.25, 0.0)
    # double-sided
    assert L.sqo_moller_trumbore(O.V3(0.25, 0.25, -1), O.V3(0, 0, 1), C.byref(tri), C.byref(p), C.byref(d)) == 1
    # edge hit u = 0 is inside; parallel ray and hits closer than eps are rejected
    assert L.sqo_moller_trumbore(O.V3(0, 0.5, 1), O.V3(0, 0, -1), C.byref(tri), C.byref(p), C.byref(d)) == 1
    assert L.sqo_moller_trumbore(O.V3(0.25, 0.25, 1), O.V3(1, 0, 0), C.byref(tri), C.byref(p), C.byref(d)) == 0
    assert L.sqo_moller_trumbore(O.V3(0.25, 0.25, 0.00005), O.V3(0, 0, -1), C.byref(tri), C.byref(p), C.byref(d)) == 0
    assert L.sqo_moller_trumbore(O.V3(0.25, 0.25, 1), O.V3(nan, 0, -1), C.byref(tri), C.byref(p), C.byref(d)) == 0


def test_crd_trig_against_host_libm(O):
    """The 'crd' spec vs the libm GHC would call: double kernels accurate to ~1e-16, float results
    equal to libm's except for a small, measured fraction of 1-ulp differences."""
    import math
    rng = np.random.default_rng(7)
    L = O.lib()
    for fn, ref, lo, hi in (("sin_d", math.sin, -7.0, 7.0), ("cos_d", math.cos, -7.0, 7.0),
                            ("acos_d", math.acos, -1.0, 1.0), ("atan_d", math.atan, -200.0, 200.0)):
        xs = rng.uniform(lo, hi, 20000)
        err = max(abs(getattr(L, "sqo_" + fn)(float(x)) - ref(float(x))) / max(abs(ref(float(x))), 1e-300) for x in xs)
        assert err < 1e-15, (fn, err)
    assert L.sqo_acos_d(1.0) == 0.0 and L.sqo_acos_d(-1.0) == math.pi and L.sqo_atan_d(float("inf")) == math.pi / 2
    for fn, lo, hi in (("sinf", 0.0, 6.2831855), ("cosf", 0.0, 6.2831855), ("acosf", -1.0, 1.0), ("atanf", 0.0, 120.0)):
        xs = rng.uniform(lo, hi, 50000).astype(np.float32)
        f = getattr(L, "sqo_" + fn)
        a = np.array([f(float(x), O.TRIG_CRD) for x in xs], np.float32)
        b = np.array([f(float(x), O.TRIG_LIBM) for x in xs], np.float32)
        ulp = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
        assert ulp.max() <= 1, (fn, ulp.max())
        # measured on glibc 2.35: sinf/cosf/atanf differ from correct rounding on ~1.3 % of inputs,
        # acosf (fdlibm float kernel) on ~7.7 %; always by one ulp
        assert (ulp != 0).mean() < (0.10 if fn == "acosf" else 0.02), (fn, (ulp != 0).mean())


def test_libm_mode_changes_few_pixels(oracle_scene, O):
    """Quantifies SURVEY A.10: crd vs host-libm transcendentals at config C1's shape (reduced)."""
    bih, cam, _ = oracle_scene
    a, _, _ = bih.render(cam, 4, 64, 64, threads=4, trig=O.TRIG_CRD)
    cam_l = O.load_camera(os.path.join(DATA, "camera"), O.TRIG_LIBM)
    b, _, _ = bih.render(cam_l, 4, 64, 64, threads=4, trig=O.TRIG_LIBM)
    differing = (np.abs(a - b).max(-1) > 1e-4).mean()
    assert differing < 0.02


def test_example_png_statistics(O):
    """Statistical pin against the reference's only rendered artefact (render/example.png).
    example.png predates the current data/scene.sq: it shows no mirror image of the light in the
    back wall, and its wall radiances match a render with the walls' `reflective 0.2` set to 0
    (DESIGN.md §oracle).  With that edit the oracle reproduces patch radiances within 12 %."""
    ex = json.load(open(os.path.join(GOLDEN, "example_png_patches.json")))
    obj = open(os.path.join(DATA, "scene.obj"), "rb").read()
    sq = open(os.path.join(DATA, "scene.sq"), "rb").read().replace(b"reflective 0.2", b"reflective 0")
    bih = O.BIH(O.tris_from_text(obj, sq))
    cam = O.load_camera(os.path.join(DATA, "camera"))
    y0, y1 = ex["rows"]
    avg, _, _ = bih.render(cam, 96, 540, 540, threads=os.cpu_count() or 1, rows=(y0, y1), want_rgb=False)
    for name, p in ex["patches"].items():
        c0, c1 = p["cols"]
        ours = avg[:, c0:c1].mean((0, 1))
        assert np.allclose(ours, p["radiance"], rtol=0.12, atol=0.01), (name, ours, p["radiance"])
    # geometry silhouette: every primary miss is an exactly-black pixel of example.png (a few hit
    # pixels are black too), checked globally and per 60x60 cell
    bih0 = O.BIH(O.tris_from_text(obj, open(os.path.join(DATA, "scene.sq"), "rb").read()))
    miss = np.zeros((9, 9)); tot = np.zeros((9, 9))
    for y in range(2, 540, 6):
        for x in range(2, 540, 6):
            o, d = O.make_ray(540, 540, y, x, cam)
            miss[y // 60, x // 60] += 0 if bih0.intersect(o, d).hit else 1
            tot[y // 60, x // 60] += 1
    frac = miss.sum() / tot.sum()
    assert frac <= ex["black_fraction"] + 0.005 and ex["black_fraction"] - frac < 0.03
    cells = np.array(ex["black_fraction_9x9_cells"])
    assert np.all((miss / tot)[cells == 1.0] == 1.0)
    outside_suzanne = np.ones((9, 9), bool)
    outside_suzanne[6:9, 1:5] = False        # the mirror monkey: hit pixels that still come out black
    assert np.abs(miss / tot - cells)[outside_suzanne].max() < 0.06
    assert np.all((miss / tot - cells)[~outside_suzanne] <= 0.0)


def test_loader_grammar_edge_cases(O):
    sq = b"newmtl A\nreflective 0 1 1 1\nemissive 0 0 0 0\n\nnewmtl B\nreflective 1 .5 0.5 0.5\nemissive 2 1 1 1\n"
    with pytest.raises(O.OracleError):     # ".5" is not readable by Haskell `read`
        O.tris_from_text(b"mtllib s.sq\no X\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl B\nf 1 2 3\n", sq)
    sq = sq.replace(b".5 0.5", b"0.5 0.5")
    obj = (b"mtllib s.sq\r\no Cube.001_x\r\nv 0 0 0\r\nv 1 0 0\r\nv 0 1 0\r\nusemtl A\r\ns off\r\nf 1 2 3\r\n"
           b"o Second\nv 0 0 1\nusemtl B\ns on\nf 1 2 4\nf 4 2 1\n")
    t = O.tris_from_text(obj, sq)
    assert len(t) == 3
    assert t["c"][1].tolist() == [0.0, 1.0, 0.0]           # global 1-based index 4 -> (0,0,1) with swapYZ
    assert t["reflective"].tolist() == [0.0, 1.0, 1.0] and t["emissive"].tolist() == [0.0, 2.0, 2.0]
    # unmatched material: object silently dropped (src/Obj.hs:75); duplicate names: duplicated triangles
    assert len(O.tris_from_text(obj.replace(b"usemtl B", b"usemtl C"), sq)) == 1
    assert len(O.tris_from_text(obj, sq + b"newmtl A\nreflective 0 0 0 0\nemissive 0 0 0 0\n")) == 4
    for bad in (b"o X\n", b"mtllib s.sq\no X\nv 0 0 0\nf 1 1 1\n", b"mtllib s.sq\no X\nv 0 0 0\nusemtl A\nf 1/1 1 1\n",
                b"mtllib s.sq\no X\nv 0 0 0\nusemtl A\nf 1 1 2\n", b"mtllib s.sq\no X\nv 1e-3 0 0\nusemtl A\nf 1 1 1\n"):
        with pytest.raises(O.OracleError):
            O.tris_from_text(bad, sq)
    # empty scene: no objects -> no triangles
    assert len(O.tris_from_text(b"mtllib s.sq\n", sq)) == 0
    cam = O.camera_from_text(b"0 7 0.75\n1.5707963267948966 0 -0.09817477042468103\n")
    assert O.camera_arrays(cam)[0].tolist() == [0.0, 7.0, 0.75]


def test_small_and_degenerate_trees(O):
    # fewer than 15 triangles: the root is a Leaf and no box test happens (src/BIH.hs:69,105)
    sq = b"newmtl A\nreflective 0 1 1 1\nemissive 1 1 1 1\n"
    obj = b"mtllib s.sq\no X\nv -1 0 -1\nv 1 0 -1\nv 0 0 1\nusemtl A\nf 1 2 3\n"
    b1 = O.BIH(O.tris_from_text(obj, sq))
    assert (b1.height, b1.num_leaves, b1.n_nodes) == (1, 1, 1)
    # swapYZ puts the triangle in the plane z = 0
    assert b1.intersect((0, 0, 1), (0, 0, 1)).hit == 0 and b1.intersect((0, 0, -3), (0, 0, 1)).hit == 1
    # 20 identical triangles: every centroid equals the mean -> left side empty -> terminal branch
    # with an empty leaf and an oversized leaf (src/BIH.hs:70-72)
    obj20 = b"mtllib s.sq\no X\nv -1 0 -1\nv 1 0 -1\nv 0 1 1\nusemtl A\n" + b"f 1 2 3\n" * 20
    b20 = O.BIH(O.tris_from_text(obj20, sq))
    kind, _, _, cnt = b20.preorder()
    assert kind.tolist()[1:] == [3, 3] and cnt.tolist() == [0, 0, 20] and b20.longest_leaf == 20
    h = b20.intersect((0, 0, -3), (0, 0, 1))
    assert h.hit == 1 and h.tri == 0                         # ties go to the earliest triangle
    # a FLAT root box (all triangles in one axis plane) never passes `tmin < tmax`: reference behaviour
    flat20 = b"mtllib s.sq\no X\nv -1 0 -1\nv 1 0 -1\nv 0 0 1\nusemtl A\n" + b"f 1 2 3\n" * 20
    assert O.BIH(O.tris_from_text(flat20, sq)).intersect((0, 0, -3), (0, 0, 1)).hit == 0
    # empty scene
    b0 = O.BIH(O.tris_from_text(b"mtllib s.sq\n", sq))
    assert b0.n_tris == 0 and b0.intersect((0, 0, -3), (0, 0, 1)).hit == 0


def test_atan_and_tonemap_accept_non_finite_input(O):
    """Regression for the checker's own crash in round 1 (gpurun_out/fuzz3.log): the table-driven atan
    indexed TAB[(int)NaN] (INT_MIN on x86) for a NaN radiance, a host out-of-bounds read.  Every entry point that can
    see a non-finite pixel must return the Haskell value: atan NaN = NaN, atan +-inf = +-pi/2, and
    `floor :: Float -> Word8` of NaN / inf = 0 (src/Lib.hs:93-104)."""
    import ctypes as C
    import math
    L = O.lib()
    L.sqo_atan_d.restype = C.c_double
    L.sqo_atan_d.argtypes = [C.c_double]
    L.sqo_atanf.restype = C.c_float
    L.sqo_atanf.argtypes = [C.c_float, C.c_int]
    for trig in (O.TRIG_CRD, O.TRIG_LIBM):
        assert math.isnan(L.sqo_atanf(float("nan"), trig))
        assert L.sqo_atanf(float("inf"), trig) == np.float32(math.pi / 2)
        assert L.sqo_atanf(float("-inf"), trig) == -np.float32(math.pi / 2)
        assert L.sqo_atanf(3.0e38, trig) == np.float32(math.pi / 2)
    assert math.isnan(L.sqo_atan_d(float("nan")))
    assert L.sqo_atan_d(float("inf")) == math.pi / 2 and L.sqo_atan_d(float("-inf")) == -math.pi / 2
    nan, inf = float("nan"), float("inf")
    for c in ((nan, nan, nan), (nan, 1.0, 0.5), (1.0, nan, 0.5), (inf, inf, inf), (inf, 1.0, 0.0), (-inf, 2.0, 1.0), (0.0, 0.0, 0.0)):
        for trig in (O.TRIG_CRD, O.TRIG_LIBM):
            out = O.tonemap(c, trig)                           # must not crash; NaN / inf channels floor to 0
            assert len(out) == 3
    assert tuple(O.tonemap((nan, nan, nan))) == (0, 0, 0)
    assert tuple(O.tonemap((0.0, 0.0, 0.0))) == (0, 0, 0)      # 0/0 = NaN scale: black
    assert tuple(O.tonemap((inf, 1.0, 0.0))) == (0, 0, 0)      # intensity 1 / inf = 0 scale, 0 * inf = NaN -> 0


def overflow_room_obj():
    """A closed cube room of absorbing walls (material Black) with a lamp quad (material Sun) inside, as .obj text."""
    r = 3
    corners = [(x, y, z) for x in (-r, r) for y in (-r, r) for z in (-r, r)]
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    out = ["mtllib scene.sq", "o Room"] + ["v %d %d %d" % c for c in corners] + ["usemtl Black", "s off"]
    for a, b, c, d in quads:
        out += ["f %d %d %d" % (a + 1, b + 1, c + 1), "f %d %d %d" % (a + 1, c + 1, d + 1)]
    out += ["o Lamp", "v -2 2.5 -2", "v 2 2.5 -2", "v 2 2.5 2", "v -2 2.5 2", "usemtl Sun", "s off", "f 9 10 11", "f 9 11 12"]
    return ("\n".join(out) + "\n").encode()


def test_absorbing_surface_in_front_of_overflowing_radiance_is_nan(O):
    """src/Lib.hs:135: `surfColor * raytrace ...` with surfColor = 0 is 0 * L.  While L is finite that is +0 (the
    shortcut the HIP path takes); when the emission `emissive *^ emitColor` overflows, L = inf and 0 * inf = NaN.
    The .sq grammar reaches that with long digit strings (no exponent syntax, src/Obj.hs:115-121)."""
    big = b"1" + b"0" * 30                                      # 1e30, 31 digits
    sq = (b"newmtl Black\nreflective 0 0 0 0\nemissive 0 0 0 0\n\n"
          b"newmtl Sun\nreflective 0 0 0 0\nemissive " + big + b" " + big + b" " + big + b" " + big + b"\n")
    obj = overflow_room_obj()
    tris = O.tris_from_text(obj, sq)
    with np.errstate(over="ignore"):
        assert np.isinf(np.float32(tris["emissive"][-1]) * np.float32(tris["emit"][-1][0]))
    ob = O.BIH(tris)
    cam = O.camera_from_text(b"0 0 0\n0 0 0\n")               # inside the closed black room
    avg, rgb, _ = ob.render(cam, 8, 16, 16, threads=2)
    assert np.isnan(avg).any(), "some scatter ray off the absorbing wall must reach the overflowing lamp"
    assert (rgb[np.isnan(avg).any(-1)] == 0).all()


def test_ghc_golden_pins_the_oracle(O, oracle_scene):
    """The slot that turns "parity unpinned" into "pinned": consumes the files integration/DumpGolden.hs writes when
    it is run against a GHC build of the reference (tests/golden/ghc_tfgen_words.bin, ghc_avg_64x64_4spp.bin) and,
    optionally, the PNG of the reference's own CLI (`-d 64,64 -s 4 -p ghc_64x64_4spp.png`).  Skipped until they exist:
    nobody can produce them in this repository's environments (no GHC)."""
    words_path = os.path.join(GOLDEN, "ghc_tfgen_words.bin")
    avg_path = os.path.join(GOLDEN, "ghc_avg_64x64_4spp.bin")
    png_path = os.path.join(GOLDEN, "ghc_64x64_4spp.png")
    if not (os.path.exists(words_path) or os.path.exists(avg_path) or os.path.exists(png_path)):
        pytest.skip("no tests/golden/ghc_* files: run integration/DumpGolden.hs against a GHC build of the reference")
    ob, cam, _ = oracle_scene
    variant = 0
    if os.path.exists(words_path):
        want = np.fromfile(words_path, "<u4").reshape(4, 8)
        seeds = [0, 1, 2, 2 ** 32 + 5]
        matching = [v for v in range(4) if all(O.tfgen_words(s, v) == [int(x) for x in want[i]] for i, s in enumerate(seeds))]
        assert matching, ("no TFGen layout candidate of the oracle (SURVEY App. B) reproduces tf-random's words; first seed: "
                          f"GHC {[hex(int(x)) for x in want[0]]} vs variant 0 {[hex(x) for x in O.tfgen_words(0, 0)]}")
        variant = matching[0]
        assert variant == 0, f"tf-random's word layout is oracle rng_variant {variant}, not the default 0: switch the default (and sq_math.h tfgen3)"
    cam_by_trig = {t: O.load_camera(os.path.join(DATA, "camera"), t) for t in (O.TRIG_CRD, O.TRIG_LIBM)}
    if os.path.exists(avg_path):
        want = np.fromfile(avg_path, "<f4").reshape(64, 64, 3)
        stats = {}
        for name, trig in (("crd", O.TRIG_CRD), ("libm", O.TRIG_LIBM)):
            got, _, _ = ob.render(cam_by_trig[trig], 4, 64, 64, threads=4, trig=trig, rng_variant=variant, want_rgb=False)
            close = np.abs(got - want).max(-1) <= 1e-4
            stats[name] = (float(close.mean()), float((got.view(np.uint32) == want.view(np.uint32)).all(-1).mean()))
        print("GHC avg vs oracle: fraction of pixels within 1e-4 / bit-identical:", stats)
        # host-libm mode should agree almost everywhere (libm versions differ in the last ulp of a few calls);
        # the crd mode the GPU uses is expected to move < 2 % of the pixels by more than 1e-4
        assert stats["libm"][0] >= 0.98, stats
        assert stats["crd"][0] >= 0.97, stats
    if os.path.exists(png_path):
        from PIL import Image
        want8 = np.asarray(Image.open(png_path).convert("RGB"))
        _, got8, _ = ob.render(cam_by_trig[O.TRIG_CRD], 4, 64, 64, threads=4, rng_variant=variant)
        assert want8.shape == got8.shape
        assert (np.abs(want8.astype(int) - got8.astype(int)).max(-1) <= 1).mean() >= 0.97
