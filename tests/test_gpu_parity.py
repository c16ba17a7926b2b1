"""GPU parity tests proper: the HIP path (through the C-ABI) against the CPU oracle, bit for bit.

The bar is bit-exact on the fp32 `avg` framebuffer (src/Lib.hs:88) and on the RGB8 output; the
north-star tolerance of 1e-4 per pixel is therefore met with margin.  Run with -m gpu on an MI355X.
"""
import json
import os

import numpy as np
import pytest

from conftest import DATA, GOLDEN, ROOT

pytestmark = pytest.mark.gpu
THREADS = min(os.cpu_count() or 1, 16)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def dev(sqt, product_scene):
    assert sqt.device_count() >= 1, "no HIP device: the product has no CPU fallback"
    bih, cam, _ = product_scene
    ds = sqt.DeviceScene(bih, 0)
    yield ds
    ds.close()


def test_native_library_is_the_path_that_runs(sqt):
    import ctypes
    assert os.path.exists(sqt.LIB_PATH)
    maps = open("/proc/self/maps").read()
    sqt.lib()
    assert "libsquigly_hip.so" in open("/proc/self/maps").read() or "libsquigly_hip.so" in maps


def test_golden_fixtures_bit_exact(sqt, product_scene):
    bih, cam, _ = product_scene
    avg = sqt.render_f32(bih, cam, 4, (64, 64))
    rgb = sqt.render_rgb8(bih, cam, 4, (64, 64))
    assert np.array_equal(bits(avg), bits(np.load(os.path.join(GOLDEN, "scene_64x64_4spp_avg.npy"))))
    assert np.array_equal(rgb, np.load(os.path.join(GOLDEN, "scene_64x64_4spp_rgb8.npy")))
    avg2 = sqt.render_f32(bih, cam, 3, (40, 72))          # w rows x h columns (src/Lib.hs:70-71)
    assert avg2.shape == (40, 72, 3)
    assert np.array_equal(bits(avg2), bits(np.load(os.path.join(GOLDEN, "scene_40x72_3spp_avg.npy"))))


def test_cast_mode_golden(sqt, product_scene):
    bih, cam, _ = product_scene
    rgb = sqt.render_rgb8(bih, cam, 2, (64, 64), cast=True)
    avg = sqt.render_f32(bih, cam, 2, (64, 64), cast=True)
    assert np.array_equal(rgb, np.load(os.path.join(GOLDEN, "scene_64x64_cast_rgb8.npy")))
    assert np.array_equal(bits(avg), bits(np.load(os.path.join(GOLDEN, "scene_64x64_cast_avg.npy"))))


@pytest.mark.parametrize("w,h,n", [(96, 96, 16), (33, 129, 5), (128, 17, 7), (1, 1, 3), (256, 256, 4)])
def test_against_oracle_same_inputs(sqt, product_scene, oracle_scene, w, h, n):
    bih, cam, _ = product_scene
    ob, ocam, _ = oracle_scene
    g = sqt.render_f32(bih, cam, n, (w, h))
    g8 = sqt.render_rgb8(bih, cam, n, (w, h))
    o, o8, _ = ob.render(ocam, n, w, h, threads=THREADS)
    assert np.array_equal(bits(g), bits(o)), f"{int((bits(g) != bits(o)).any(-1).sum())} pixels differ"
    assert np.array_equal(g8, o8)
    assert np.abs(g - o).max() <= 1e-4                     # the north-star tolerance, trivially


def test_seeds_beyond_32_bits(sqt, product_scene, oracle_scene):
    """rix = n*(x+y*w) exceeds 2^32 on config C4 (3840x2160 @ 1024): seeds are 64-bit Ints.
    A 70000-row x 2-column image at 40000 spp reaches the same range; compare a few rows."""
    bih, cam, _ = product_scene
    ob, ocam, _ = oracle_scene
    w, h, n = 70000, 2, 3
    # rix max = 3*(1 + 69999*70000) = 1.47e10 > 2^32
    ds = sqt.DeviceScene(bih, 0)
    import torch
    avg, _ = ds.render_rows(cam, n, w, h, want_rgb=False)
    torch.cuda.synchronize()
    g = avg.cpu().numpy()
    for y0 in (0, 35000, 69990):
        o, _, _ = ob.render(ocam, n, w, h, threads=THREADS, rows=(y0, y0 + 10), want_rgb=False)
        assert np.array_equal(bits(g[y0:y0 + 10]), bits(o))
    ds.close()


def test_small_degenerate_and_empty_scenes(sqt, O):
    sq = b"newmtl A\nreflective 0 1 1 1\nemissive 1 1 1 1\nnewmtl M\nreflective 1 0.9 0.9 0.9\nemissive 0 0 0 0\n"
    cam_txt = b"0 5 0.3\n1.5707963267948966 0 -0.05\n"
    scenes = {
        "single-leaf": b"mtllib s.sq\no X\nv -1 0 -1\nv 1 0 -1\nv 0 1 1\nusemtl A\nf 1 2 3\n",
        "twenty-identical": b"mtllib s.sq\no X\nv -1 0 -1\nv 1 0 -1\nv 0 1 1\nusemtl A\n" + b"f 1 2 3\n" * 20,
        # a terminal leaf longer than a packed leaf reference can describe (127): the leaf-table path of the streaming form
        "two-hundred-identical": (b"mtllib s.sq\no X\nv -1 0 -1\nv 1 0 -1\nv 0 1 1\nusemtl A\n" + b"f 1 2 3\n" * 200 +
                                  b"o M\nv -3 -3 -3\nv 3 -3 -3\nv 0 -3 3\nusemtl M\nf 4 5 6\n"),
        "flat-root-box": b"mtllib s.sq\no X\nv -1 0 -1\nv 1 0 -1\nv 0 0 1\nusemtl A\n" + b"f 1 2 3\n" * 20,
        "empty": b"mtllib s.sq\n",
        "mirror-and-light": (b"mtllib s.sq\no L\nv -1 -1 -1\nv 1 -1 -1\nv 0 -1 1\nusemtl A\nf 1 2 3\n"
                             b"o M\nv -3 -3 -3\nv 3 -3 -3\nv 0 -3 3\nusemtl M\nf 4 5 6\n"),
    }
    cam_p, cam_o = sqt.camera_from_text(cam_txt), O.camera_from_text(cam_txt)
    # a material file with NEGATIVE components switches off the exact `surfColor == 0` shortcuts (they are
    # only identities when every radiance is >= +0): both code paths must equal the oracle
    sq_neg = sq.replace(b"emissive 1 1 1 1", b"emissive 1 1 -0.5 1").replace(b"reflective 1 0.9 0.9 0.9", b"reflective 1 0.9 -0.9 0")
    for name, obj in list(scenes.items()) + [("negative-materials", scenes["mirror-and-light"])]:
        if name == "negative-materials":
            sq = sq_neg
        bih = sqt.BIH(sqt.Mesh.from_text(obj, sq))
        ob = O.BIH(O.tris_from_text(obj, sq))
        for cast in (False, True):
            g = sqt.render_f32(bih, cam_p, 3, (24, 20), cast=cast)
            o, _, _ = ob.render(cam_o, 3, 24, 20, cast=cast, threads=4)
            assert np.array_equal(bits(g), bits(o)), (name, cast)


def test_sharded_render_equals_unsharded(sqt, product_scene, dev):
    """§8e: any row partition reproduces the single-GPU image bit for bit (virtual shards on 1 GPU)."""
    import torch
    from importlib import import_module
    d = import_module("squigly-trace_amd.dist")
    bih, cam, _ = product_scene
    w, h, n = 75, 40, 3
    full, full8 = dev.render_rows(cam, n, w, h)
    torch.cuda.synchronize()
    for rb, world in [(8, 2), (8, 8), (1, 3), (16, 5)]:
        frame = torch.zeros_like(full)
        frame8 = torch.zeros_like(full8)
        for r in range(world):
            a, b8 = dev.render_rows(cam, n, w, h, shard=(rb, r, world))
            rows = d.shard_rows(w, rb, r, world)
            assert a.shape[0] == len(rows)
            if rows:
                idx = torch.tensor(rows, device=a.device)
                frame[idx] = a
                frame8[idx] = b8
        torch.cuda.synchronize()
        assert torch.equal(frame.view(torch.int32), full.view(torch.int32)) and torch.equal(frame8, full8)


@pytest.mark.parametrize("w,h,n,slots", [(120, 90, 9, 120 * 90 * 2), (64, 200, 5, 1 << 20)])
def test_kernel_variants_agree(sqt, product_scene, oracle_scene, dev, w, h, n, slots):
    """The one-lane-per-pixel kernel (variant 1) and the wavefront pipeline (variant 2, also with several
    sample batches) produce identical bits, and both equal the oracle."""
    import torch
    bih, cam, _ = product_scene
    ob, ocam, _ = oracle_scene
    outs = []
    # variant 1 = per-pixel kernel; variant 2 = wavefront pipeline with the resident (LDS) or the streaming trace
    # kernel, with and without the lane-occupancy counters (a separate template instantiation)
    # ... and with the overlapped two-track schedule (several batches in flight on two streams)
    # ... and with the primary rays through the pooled trace kernel (option primary_pooled), both scene forms and the two-pipeline schedule
    for variant, resident, profile, overlap, pooled_primary in ((1, 1, 0, 0, 0), (2, 1, 0, 0, 0), (2, 0, 0, 0, 0), (2, 1, 1, 0, 0), (2, 0, 1, 0, 0), (2, 1, 0, 1, 0), (2, 0, 0, 1, 0),
                                                                (2, 1, 0, 0, 1), (2, 0, 0, 0, 1), (2, 1, 0, 2, 1)):
        dev.set_option("variant", variant)
        dev.set_option("resident", resident)
        dev.set_option("profile", profile)
        dev.set_option("overlap", overlap)
        dev.set_option("primary_pooled", pooled_primary)
        dev.set_option("slots", slots)
        a, r = dev.render_rows(cam, n, w, h)
        torch.cuda.synchronize()
        outs.append((a.cpu().numpy(), r.cpu().numpy()))
    st = dev.stats(reset=True)
    assert st[0] > 0 and st[5] > 0                      # rays traced, triangle tests counted by the profile build
    dev.set_option("variant", 2)
    dev.set_option("resident", 1)
    dev.set_option("profile", 0)
    dev.set_option("overlap", 0)
    dev.set_option("primary_pooled", 0)
    dev.set_option("slots", 512 << 20)
    o, o8, _ = ob.render(ocam, n, w, h, threads=THREADS)
    for a, r in outs:
        assert np.array_equal(bits(a), bits(o)) and np.array_equal(r, o8)


def test_c_abi_error_behaviour(sqt, product_scene):
    bih, cam, _ = product_scene
    for dims, n in [((0, 4), 1), ((4, 0), 1), ((4, 4), 0), ((-1, 4), 1)]:
        with pytest.raises(sqt.SquiglyError):
            sqt.render_rgb8(bih, cam, n, dims)
    # malformed trees are rejected at upload, never launched
    import ctypes as C
    nodes = bih.nodes.copy()
    for mutate in ("link", "range", "kind"):
        bad = nodes.copy()
        if mutate == "link":
            bad["link"][0] = 5 if bad["link"][0] != 5 else 6
        elif mutate == "range":
            leaf = np.nonzero((bad["kind"] & 3) == 3)[0][-1]
            bad["link"][leaf] = 1 << 30
        else:
            bad["kind"][0] = 3 | (10 << 2)
        sc = sqt._native.Scene()
        C.memmove(C.byref(sc), C.byref(bih.scene), C.sizeof(sc))
        sc.nodes = bad.ctypes.data
        h = C.c_void_p()
        assert sqt.lib().sq_scene_upload(C.byref(sc), 0, C.byref(h)) != 0
        assert len(sqt.lib().sq_last_error()) > 0


def test_full_size_c2_frame(sqt, product_scene, oracle_scene, dev):
    """BASELINE configs[1], the headline workload: 1920x1080 @ 256 spp on the GPU.
    (a) every third row of the frame (640 rows = 691 200 pixels = 177 M samples) equals the oracle bit for bit,
        fp32 and RGB8;  (b) re-rendering gives the same bits;  (c) eight interleaved shards reassemble to the
        same frame;  (d) a pixel whose primary ray misses is exactly zero."""
    import torch
    from importlib import import_module
    d = import_module("squigly-trace_amd.dist")
    bih, cam, _ = product_scene
    ob, ocam, _ = oracle_scene
    w, h, n = 1920, 1080, 256
    avg, rgb = dev.render_rows(cam, n, w, h)
    torch.cuda.synchronize()
    g, g8 = avg.cpu().numpy(), rgb.cpu().numpy()
    o, o8, _ = ob.render(ocam, n, w, h, threads=THREADS, rows=(1, w), row_step=3)
    assert np.array_equal(bits(g[1::3]), bits(o)), f"{int((bits(g[1::3]) != bits(o)).any(-1).sum())} pixels differ"
    assert np.array_equal(g8[1::3], o8)
    avg2, rgb2 = dev.render_rows(cam, n, w, h)
    torch.cuda.synchronize()
    assert torch.equal(avg2.view(torch.int32), avg.view(torch.int32)) and torch.equal(rgb2, rgb)
    frame = torch.zeros_like(avg)
    for r in range(8):
        a, _ = dev.render_rows(cam, n, w, h, shard=(d.ROW_BLOCK, r, 8), want_rgb=False)
        frame[torch.tensor(d.shard_rows(w, d.ROW_BLOCK, r, 8), device=a.device)] = a
    torch.cuda.synchronize()
    assert torch.equal(frame.view(torch.int32), avg.view(torch.int32))
    import pyoracle
    for y in (7, 600, 1333):
        for x in range(0, h, 7):
            o_, d_ = pyoracle.make_ray(w, h, y, x, ocam)
            if not ob.intersect(o_, d_).hit:
                assert (g[y, x] == 0).all()


def test_device_primitives_bit_exact(sqt, O):
    """Every primitive of the numeric spec, evaluated on the GPU, equals the oracle bit for bit:
    IEEE sqrt and divide (no 1-ulp native forms), the crd transcendentals, randomR, TFGen, tonemap."""
    rng = np.random.default_rng(11)
    L = O.lib()
    n = 200000
    x = np.concatenate([rng.uniform(0, 50, n), 10.0 ** rng.uniform(-40, 38, n), [0.0, 1e-45, 3e-39, np.inf]]).astype(np.float32)
    assert np.array_equal(bits(sqt.debug_eval("sqrt", x)), bits(np.sqrt(x)))
    a = np.concatenate([rng.normal(0, 10, n), 10.0 ** rng.uniform(-30, 30, n), [1, 2, 0, -0.0, 1, np.inf]]).astype(np.float32)
    b = np.concatenate([rng.normal(0, 3, n), 10.0 ** rng.uniform(-30, 30, n), [0, 3, 0, 5, -0.0, np.inf]]).astype(np.float32)
    with np.errstate(all="ignore"):
        want = a / b
    got = sqt.debug_eval("div", a, b)
    ok = (bits(got) == bits(want)) | (np.isnan(got) & np.isnan(want))
    assert ok.all(), int((~ok).sum())
    for op, fn, lo, hi in (("sin", "sqo_sinf", -8, 8), ("cos", "sqo_cosf", -8, 8), ("acos", "sqo_acosf", -1, 1), ("atan", "sqo_atanf", -150, 150)):
        xs = np.concatenate([rng.uniform(lo, hi, 60000), [lo, hi, 0.0, 0.5, -0.5]]).astype(np.float32)
        f = getattr(L, fn)
        want = np.array([f(float(v), O.TRIG_CRD) for v in xs], np.float32)
        assert np.array_equal(bits(sqt.debug_eval(op, xs)), bits(want)), op
    u = np.concatenate([rng.integers(0, 2 ** 32, 100000, dtype=np.uint64), [0, 1, 2 ** 32 - 1, 2 ** 31, 2 ** 24 + 1]]).astype(np.uint32)
    want = (u.astype(np.float32) / np.float32(4294967296.0)).astype(np.float32)
    assert np.array_equal(bits(sqt.debug_eval("unit_float", u)), bits(want))
    seeds = np.concatenate([rng.integers(0, 2 ** 40, 3000), [0, 1, 2, 2 ** 32 + 5, 8493465599, -1, -2 ** 63]]).astype(np.int64)
    got = sqt.debug_eval("tfgen3", seeds)
    want = np.array([O.tfgen_words(int(s))[:3] for s in seeds], np.uint32)
    assert np.array_equal(got, want)
    c = np.concatenate([rng.uniform(0, 3, (20000, 3)), rng.uniform(0, 120, (2000, 3)), [[0, 0, 0], [100, 100, 100], [0, 0, 1], [np.inf, 1, 1]]]).astype(np.float32)
    got = sqt.debug_eval("tonemap", c)
    want = np.array([O.tonemap(tuple(float(v) for v in row)) for row in c], np.uint8)
    assert np.array_equal(got, want)


def test_short_reciprocal_is_the_ieee_quotient_on_its_whole_range(sqt):
    """The pooled triangle test replaces 1 / det by v_rcp_f32 + two FMA Newton steps where 2^-14 <= |det| <= 2^100
    (sq_scene.h rcp_midrange; smaller determinants are rejected by the test's first guard, larger ones take the division).
    Exhaustive: every float of that range, both signs (1.9e9 values), against the device's IEEE division, which
    test_device_primitives_bit_exact ties to numpy's."""
    lo, hi = 0x3880, 0x7180                      # upper halves of 2^-14 and 2^100
    blocks = np.arange(lo, hi + 1, dtype=np.uint32)
    blocks = np.concatenate([blocks, blocks | 0x8000]).astype(np.uint32)
    bad = sqt.debug_eval("rcp_sweep", blocks)
    assert int(bad.sum()) == 0, [(hex(int(b)), int(c)) for b, c in zip(blocks[bad > 0][:8], bad[bad > 0][:8])]
    # outside the range the short form is NOT exact (which is why the kernel checks): denormal results lose bits
    assert int(sqt.debug_eval("rcp_sweep", np.array([0x7e80], np.uint32)).sum()) > 0


def test_axis_aligned_rays_take_the_exact_slab_path(sqt, O):
    """A camera with zero Euler angles shoots dir = (1, xoffs, yoffs): the centre column has d.y == 0 and the
    centre row d.z == 0, so 1/d is infinite, `(bound - o) * df` can be 0 * inf = NaN, and the traversal must
    follow Haskell's min/max through NaN (slab(), not the v_min/v_max form).  Bit-equal to the oracle."""
    cam_txt = b"-7 0.25 0.5\n0 0 0\n"
    obj = open(os.path.join(DATA, "scene.obj"), "rb").read()
    sq = open(os.path.join(DATA, "scene.sq"), "rb").read()
    bih = sqt.BIH(sqt.Mesh.from_text(obj, sq))
    ob = O.BIH(O.tris_from_text(obj, sq))
    cam_p, cam_o = sqt.camera_from_text(cam_txt), O.camera_from_text(cam_txt)
    d = O.make_ray(64, 64, 32, 32, cam_o)[1]
    assert d[1] == 0.0 and d[2] == 0.0                      # the centre pixel looks exactly along +x
    for (w, h, n, cast) in [(64, 64, 3, False), (33, 48, 2, False), (64, 64, 1, True)]:
        g = sqt.render_f32(bih, cam_p, n, (w, h), cast=cast)
        o, _, _ = ob.render(cam_o, n, w, h, cast=cast, threads=THREADS)
        assert np.array_equal(bits(g), bits(o)), (w, h, n, cast)
    assert ob.intersect(*O.make_ray(64, 64, 32, 32, cam_o)).hit == 1     # the axis-aligned centre ray does hit the scene


def test_origin_on_box_planes_and_degenerate_directions(sqt, O):
    """Rays that start exactly on slab planes with zero direction components (0 * inf = NaN in the slab test),
    from a hand-made scene whose coordinates are exact in binary."""
    sq = b"newmtl A\nreflective 0 1 1 1\nemissive 1 1 1 1\nnewmtl W\nreflective 0 0.5 0.5 0.5\nemissive 0 0 0 0\n"
    rng = np.random.default_rng(5)
    v = []
    f = []
    for i in range(40):                                     # 40 small axis-aligned quads on a lattice -> a real tree
        cx, cy, cz = rng.integers(-4, 5, 3) * 0.5
        base = len(v)
        v += [(cx, cy, cz), (cx + 0.5, cy, cz), (cx + 0.5, cy + 0.5, cz), (cx, cy + 0.5, cz)]
        f += [(base + 1, base + 2, base + 3), (base + 1, base + 3, base + 4)]
    obj = b"mtllib s.sq\no Q\n" + b"".join(b"v %.1f %.1f %.1f\n" % (a, c, b) for a, b, c in v) + b"usemtl A\n" + \
        b"".join(b"f %d %d %d\n" % t for t in f)
    bih = sqt.BIH(sqt.Mesh.from_text(obj, sq))
    ob = O.BIH(O.tris_from_text(obj, sq))
    assert bih.height > 1
    for cam_txt in (b"-2 0 0.5\n0 0 0\n", b"0 -2 0\n1.5707963267948966 0 0\n", b"-2.5 -2 -2\n0 0 0\n"):
        cam_p, cam_o = sqt.camera_from_text(cam_txt), O.camera_from_text(cam_txt)
        for cast in (False, True):
            g = sqt.render_f32(bih, cam_p, 2, (32, 32), cast=cast)
            o, _, _ = ob.render(cam_o, 2, 32, 32, cast=cast, threads=4)
            assert np.array_equal(bits(g), bits(o)), (cam_txt, cast)


def test_cpp_cli_writes_the_golden_image(sqt, tmp_path):
    """The C++ `squigly-trace` executable (app/Main.hs flags over the C-ABI) renders data/scene.obj and writes a PNG
    whose pixels are the golden RGB8 fixture; the Python CLI mirror does the same."""
    import subprocess
    from PIL import Image
    from conftest import ROOT
    exe = os.path.join(ROOT, "squigly-trace_amd", "bin", "squigly-trace")
    assert os.path.exists(exe), "build() must produce the CLI"
    gold = np.load(os.path.join(GOLDEN, "scene_64x64_4spp_rgb8.npy"))
    out = str(tmp_path / "cli.png")
    r = subprocess.run([exe, "-s", "4", "--dimensions=64,64", "-p", out, "--debug"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "BIH height is 13" in r.stdout and "Number of leaves is 640" in r.stdout and "Took " in r.stdout
    assert np.array_equal(np.array(Image.open(out).convert("RGB")), gold)
    # cast mode through the same binary
    gold_cast = np.load(os.path.join(GOLDEN, "scene_64x64_cast_rgb8.npy"))
    r = subprocess.run([exe, "-s", "2", "-d", "64,64", "--savepath", out, "--cast"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(np.array(Image.open(out).convert("RGB")), gold_cast)
    # error behaviour: unknown file
    r = subprocess.run([exe, "--objpath", "/nonexistent.obj"], cwd=ROOT, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "cannot open" in r.stderr
    # Python mirror of the CLI
    import importlib
    cli = importlib.import_module("squigly-trace_amd.cli")
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        assert cli.main(["-s", "4", "-d", "64,64", "-p", out]) == 0
    finally:
        os.chdir(cwd)
    assert np.array_equal(np.array(Image.open(out).convert("RGB")), gold)


def test_config_c4_shard_of_eight(sqt, product_scene, oracle_scene, dev):
    """BASELINE configs[3]: 3840x2160 @ 1024 spp tiled over 8 GPUs.  One rank's shard (rank 3 of 8, interleaved
    blocks of d.ROW_BLOCK rows) is rendered on this GPU; two of its rows are compared with the oracle bit for bit.
    Seeds reach 1024*(2159 + 3839*3840) = 1.5e10 > 2^32 here."""
    import torch
    from importlib import import_module
    d = import_module("squigly-trace_amd.dist")
    bih, cam, _ = product_scene
    ob, ocam, _ = oracle_scene
    w, h, n, rank, world = 3840, 2160, 1024, 3, 8
    rows = d.shard_rows(w, d.ROW_BLOCK, rank, world)
    avg, rgb = dev.render_rows(cam, n, w, h, shard=(d.ROW_BLOCK, rank, world))
    torch.cuda.synchronize()
    assert avg.shape == (len(rows), h, 3) and len(rows) == w // world
    for j in (5, 300):
        y = rows[j]
        o, o8, _ = ob.render(ocam, n, w, h, threads=THREADS, rows=(y, y + 1))
        assert np.array_equal(bits(avg[j:j + 1].cpu().numpy()), bits(o)), y
        assert np.array_equal(rgb[j:j + 1].cpu().numpy(), o8)


@pytest.mark.parametrize("seed,n_emit", [(1, 3), (2, 0), (3, 70), (4, 12)])
def test_random_soups_with_mirrors_and_emitters(sqt, O, seed, n_emit):
    """Random triangle soups with diffuse, half-mirror, full-mirror and emissive triangles: exercises the
    once-per-pixel mirror ray, the last-bounce emitter test (few emitters), its off switch (70 > 64 emitters),
    a scene without emitters, occluded emitters, and COMBINE frames on overlapping geometry."""
    rng = np.random.default_rng(seed)
    n = 400
    c = rng.uniform(-1.5, 1.5, (n, 1, 3))
    v = (c + rng.normal(0, 0.35, (n, 3, 3))).astype(np.float32)
    mats = np.zeros(4, sqt._native.MAT_DTYPE)
    mats["reflective"] = [0.0, 0.5, 1.0, 0.0]
    mats["surf"] = [[0.7, 0.6, 0.5], [0.4, 0.8, 0.6], [0.9, 0.9, 0.9], [0.0, 0.0, 0.0]]
    mats["emissive"] = [0, 0, 0, 25]
    mats["emit"] = [[0, 0, 0], [0, 0, 0], [0, 0, 0], [1.0, 0.8, 0.6]]
    mat = rng.integers(0, 3, n)
    mat[rng.choice(n, n_emit, replace=False)] = 3
    tris = np.zeros(n, sqt._native.TRI_DTYPE)
    tris["v0"], tris["v1"], tris["v2"], tris["mat"] = v[:, 0], v[:, 1], v[:, 2], mat
    bih = sqt.BIH(sqt.Mesh.from_arrays(tris, mats))
    ot = np.zeros(n, O.TRI_DTYPE)
    ot["a"], ot["b"], ot["c"] = v[:, 0], v[:, 1], v[:, 2]
    for f in ("reflective", "surf", "emissive", "emit"):
        ot[f] = mats[f][mat]
    ob = O.BIH(ot)
    cam_txt = b"-6 0.1 0.2\n0 0 0\n"
    cam_p, cam_o = sqt.camera_from_text(cam_txt), O.camera_from_text(cam_txt)
    w, h, spp = 40, 40, 48
    g = sqt.render_f32(bih, cam_p, spp, (w, h))
    o, _, cnt = ob.render(cam_o, spp, w, h, threads=THREADS)
    assert np.array_equal(bits(g), bits(o)), int((bits(g) != bits(o)).any(-1).sum())
    assert cnt["b_rays"] > 20000
    if n_emit:
        assert (g > 0).any()


def test_randomised_campaign(sqt, O):
    """A slice of tests/fuzz_gpu.py (random scenes x cameras x frame shapes, both kernel forms, host- and
    device-built trees): every case bit-equal to the oracle.  The full campaign is run by hand on the GPU box."""
    import fuzz_gpu
    failures = [(seed, msg) for seed in range(1000, 1400) if (msg := fuzz_gpu.run_case(seed))]
    assert not failures, failures[:5]
    # Found by the campaign: coordinates around 1e19 make `f * dot e2 q` overflow, the hit has t = +inf, its point
    # has a NaN where the direction has a zero, its distance is NaN, and Haskell's `compare` answers GT for NaN
    # either way -- the shortcut "ta <= tb, so dist a <= dist b" must not be taken for such hits.
    assert fuzz_gpu.run_case(504773, kinds=9) is None


def test_one_shot_call_spreads_over_devices(sqt, product_scene, oracle_scene, monkeypatch):
    """sq_render_rgb8 / sq_render_f32 shard the rows over SQ_DEVICES (default: every visible GPU for large frames),
    one host thread per device, and de-interleave the result.  On a one-GPU box the same device is named several
    times: the threading, the 2-row interleave and the ragged last block are what is under test."""
    bih, cam, _ = product_scene
    ob, ocam, _ = oracle_scene
    want = np.load(os.path.join(GOLDEN, "scene_64x64_4spp_avg.npy"))
    for devs in ("0", "0,0", "0,0,0", "0,0,0,0,0,0,0,0,0,0,0"):          # 11 shards of a 64-row frame: three get no rows
        monkeypatch.setenv("SQ_DEVICES", devs)
        assert np.array_equal(bits(sqt.render_f32(bih, cam, 4, (64, 64))), bits(want)), devs
        assert np.array_equal(sqt.render_rgb8(bih, cam, 4, (64, 64)), np.load(os.path.join(GOLDEN, "scene_64x64_4spp_rgb8.npy"))), devs
    monkeypatch.setenv("SQ_DEVICES", "0,0,0")
    o, o8, _ = ob.render(ocam, 5, 37, 29, threads=THREADS)                # 37 rows: 18 blocks of 2 and a last one of 1
    assert np.array_equal(bits(sqt.render_f32(bih, cam, 5, (37, 29))), bits(o))
    assert np.array_equal(sqt.render_rgb8(bih, cam, 5, (37, 29)), o8)
    oc, _, _ = ob.render(ocam, 1, 37, 29, cast=True, threads=THREADS)
    assert np.array_equal(bits(sqt.render_f32(bih, cam, 1, (37, 29), cast=True)), bits(oc))
    for bad in ("0,x", "99", "", "0,,1", "-1"):
        monkeypatch.setenv("SQ_DEVICES", bad)
        out = np.full((8, 8, 3), 7, np.uint8)
        with pytest.raises(sqt.SquiglyError, match="SQ_DEVICES"):
            sqt.render_rgb8(bih, cam, 1, (8, 8))
    monkeypatch.delenv("SQ_DEVICES")
    assert np.array_equal(bits(sqt.render_f32(bih, cam, 4, (64, 64))), bits(want))
    # the frame workspace a freed scene leaves behind for the next one-shot call can be handed back, and comes back
    sqt.lib().sq_release_cached_memory()
    assert np.array_equal(bits(sqt.render_f32(bih, cam, 4, (64, 64))), bits(want))


def test_render_frame_over_rccl_equals_the_unsharded_render(sqt, product_scene, dev):
    """dist.render_frame under the `nccl` backend (= RCCL) with one rank: shard -> render -> all_gather_into_tensor ->
    de-interleave must give exactly the frame of the unsharded render; pool and classic trace kernels agree."""
    import socket
    import torch
    import torch.distributed as dist
    import importlib
    d = importlib.import_module("squigly-trace_amd.dist")
    bih, cam, _ = product_scene
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for w, h, n in ((64, 64, 4), (37, 29, 5)):
            want_avg, want_rgb = dev.render_rows(cam, n, w, h)
            got_rgb = d.render_frame(dev, cam, n, w, h, want="rgb")
            got_avg = d.render_frame(dev, cam, n, w, h, want="avg")
            torch.cuda.synchronize()
            assert got_rgb.shape == (w, h, 3) and torch.equal(got_rgb, want_rgb)
            assert torch.equal(got_avg.view(torch.int32), want_avg.view(torch.int32))
        assert np.array_equal(got_avg.cpu().numpy().view(np.uint32).shape, (37, 29, 3))
    finally:
        dist.destroy_process_group()
    assert np.array_equal(bits(d.render_frame(dev, cam, 4, 64, 64, want="avg").cpu().numpy()),
                          bits(np.load(os.path.join(GOLDEN, "scene_64x64_4spp_avg.npy"))))      # and without a process group


def test_bench_refuses_more_ranks_than_devices(sqt):
    """`python bench.py --gpus N` starts its own ranks; with fewer than N devices it must fail loudly, never report n_gpus: 1."""
    import subprocess
    import sys
    n = sqt.device_count() + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0
    assert "device" in p.stderr and "metric" not in p.stdout


def test_pooled_and_classic_trace_kernels_agree(sqt, product_scene, oracle_scene, dev):
    """Option `pool` (a wave tests its open leaves as pooled (ray, triangle) pairs) and its tunables change no bit."""
    import torch
    bih, cam, _ = product_scene
    ob, ocam, _ = oracle_scene
    w, h, n = 160, 120, 24
    o, o8, _ = ob.render(ocam, n, w, h, threads=THREADS)
    try:
        for opts in ({"pool": 0}, {"pool": 1, "refill_min": 1, "flush_min": 0}, {"pool": 1, "refill_min": 64, "flush_min": 64},
                     {"pool": 1, "refill_min": 12, "flush_min": 40, "resident": 0}, {"pool": 0, "resident": 0},
                     {"pool": 1, "resident": 0, "pixel_major": 1}, {"pool": 1, "resident": 1, "pixel_major": 1},
                     {"pool": 1, "resident": 1, "pixel_major": 0, "cull": 0}, {"pool": 0, "resident": 0, "cull": 0},
                     {"pool": 1, "resident": 1, "cull": 1, "descend_extra": 0}, {"pool": 1, "resident": 1, "descend_extra": 7, "descend_lanes": 1},
                     {"pool": 1, "resident": 0, "descend_extra": 3, "descend_lanes": 33},
                     {"pool": 1, "resident": 0, "lds_node_kb": 1}, {"pool": 0, "resident": 0, "lds_node_kb": 0, "cull": 1}):
            for k, v in opts.items():
                dev.set_option(k, v)
            avg, rgb = dev.render_rows(cam, n, w, h)
            torch.cuda.synchronize()
            assert np.array_equal(bits(avg.cpu().numpy()), bits(o)), opts
            assert np.array_equal(rgb.cpu().numpy(), o8), opts
    finally:
        for k, v in {"pool": 1, "refill_min": 12, "flush_min": 40, "resident": 1, "pixel_major": -1, "cull": 1, "descend_extra": 2, "descend_lanes": 16,
                     "lds_node_kb": 32}.items():
            dev.set_option(k, v)


def test_overflowing_emission_defeats_the_absorbing_shortcut(sqt, O):
    """`surfColor == 0` lets the HIP path skip nested rays only while the nested radiance is finite: with an emission
    product that overflows (31-digit literals in the .sq text; every component itself is finite) the reference computes
    0 * inf = NaN (src/Lib.hs:135) and so must the device.  Also large-but-finite products, where the bound
    max_surf * max_e + max_e decides."""
    from test_oracle import overflow_room_obj
    obj = overflow_room_obj()
    cam_txt = b"0 0 0\n0 0 0\n"
    for emissive, emit in ((b"1" + b"0" * 30, b"1" + b"0" * 30), (b"1" + b"0" * 20, b"1" + b"0" * 19), (b"3" + b"0" * 38, b"0.5"), (b"7", b"1")):
        sq = (b"newmtl Black\nreflective 0 0 0 0\nemissive 0 0 0 0\n\n"
              b"newmtl Sun\nreflective 0 0 0 0\nemissive " + emissive + b" " + emit + b" " + emit + b" " + emit + b"\n")
        ob = O.BIH(O.tris_from_text(obj, sq))
        bih = sqt.BIH(sqt.Mesh.from_text(obj, sq))
        for w, h, n in ((16, 16, 8), (24, 9, 33)):
            o, o8, _ = ob.render(O.camera_from_text(cam_txt), n, w, h, threads=THREADS)
            g = sqt.render_f32(bih, sqt.camera_from_text(cam_txt), n, (w, h))
            g8 = sqt.render_rgb8(bih, sqt.camera_from_text(cam_txt), n, (w, h))
            co, cg = bits(o).copy(), bits(g).copy()
            co[np.isnan(o)] = 0x7FC00000; cg[np.isnan(g)] = 0x7FC00000      # IEEE leaves NaN payloads open
            assert np.array_equal(co, cg), (emissive, int((co != cg).any(-1).sum()))
            assert np.array_equal(g8, o8)
    # surf * L overflows one level down: a grey wall with surfColor 3e38 in front of a finite but large emitter
    sq = (b"newmtl Black\nreflective 0 " + b"3" + b"0" * 38 + b" 0 0\nemissive 0 0 0 0\n\n"
          b"newmtl Sun\nreflective 0 0 0 0\nemissive 5 1 1 1\n")
    ob = O.BIH(O.tris_from_text(obj, sq)); bih = sqt.BIH(sqt.Mesh.from_text(obj, sq))
    o, o8, _ = ob.render(O.camera_from_text(cam_txt), 16, 16, 16, threads=THREADS)
    g = sqt.render_f32(bih, sqt.camera_from_text(cam_txt), 16, (16, 16))
    co, cg = bits(o).copy(), bits(g).copy()
    co[np.isnan(o)] = 0x7FC00000; cg[np.isnan(g)] = 0x7FC00000
    assert np.array_equal(co, cg)


def test_bench_launches_its_own_ranks(sqt):
    """`python bench.py --gpus 2` with no torchrun around it: the parent starts two rank processes, they shard the frame,
    gather it and rank 0's JSON line comes back through the parent.  On this one-GPU box the two ranks share the device
    and gather over gloo (test-only flags; RCCL refuses two ranks on one GPU); the line is marked as not a measurement."""
    import json
    import subprocess
    import sys
    common = ["--steps", "1", "--warmup", "1", "--no-cpu", "--no-other", "--width", "96", "--height", "64", "--spp", "8"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    j1 = json.loads(one.stdout.strip().splitlines()[-1])
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--oversubscribe"] + common, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    lines = [ln for ln in two.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, two.stdout                                  # exactly one line, rank 0's
    j2 = json.loads(lines[0])
    # the driver's contract for the line: every key it reads, and the two objects this tier adds at N = 1
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in j1 and key in j2, key
    assert j1["unit"] == "Msamples/s" and j1["higher_is_better"] is True and j1["vs_baseline"] is None and j1["dtype"] == "f32"
    assert "workload" in j1["config"] and "model" not in j1["config"] and j1["steps"] == 1 and j1["warmup"] == 1
    assert abs(j1["value"] - j1["config"]["samples_per_step"] / j1["ms_per_step"] / 1e3) <= 0.02 * j1["value"] + 0.01
    roof = j1["roofline"]
    assert roof["bound"] == "valu" and roof["kernel"] == "sq_trace_rays" and roof["launches"] >= 1 and roof["kernel_ms"] > 0
    for key in ("achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["frac"] is None or 0 < roof["frac"] <= 1                # None: no PMC pass of this (overridden) workload is committed
    assert "pmc_stale" in roof and j1["build_id"] == sqt.build_id()     # the counters are tied to the library that ran
    assert "roofline" not in j2 and "cpu_baseline" not in j2            # N = 1 figures
    rk = j2["ranks"]                                                    # what a scaling shortfall would be attributed with
    assert rk["world_size"] == 2 and rk["backend"] == "gloo"
    assert 0 < rk["per_rank_ms"][0] <= rk["per_rank_ms"][1] and 0 <= rk["gather_ms"][0] <= rk["gather_ms"][1]
    assert j1["n_gpus"] == 1 and j2["n_gpus"] == 2 and j2["scaling"] == "strong" and "not_a_measurement" in j2
    assert j2["config"]["samples_per_step"] == j1["config"]["samples_per_step"] == 96 * 64 * 8      # the SAME frame, shared
    assert j2["config"]["nonblack_pixels"] == j1["config"]["nonblack_pixels"] > 0                   # and the same image


def test_culling_slab_test_on_the_device_is_the_one_the_lemma_is_about(sqt):
    """SQ_OP_CULL_SLAB (the trace kernels' culling slab test: v_fma_mix_f32 on packed binary16 planes) against exact rational
    arithmetic rounded once per plane value: every binary16 value in both halves of a word, infinities and the excluded
    subnormals included, and random boxes and rays."""
    from fractions import Fraction
    from importlib import import_module
    N = import_module("squigly-trace_amd._native")
    rng = np.random.default_rng(77)
    f32 = np.float32
    # 1. conversion of every binary16 value from either half: box [x, x] x R x R against a ray along +x from x - 1 hits iff x is not NaN
    halves = np.arange(0x10000, dtype=np.uint32)
    hv = halves.astype(np.uint16).view(np.float16).astype(np.float64)
    for pos in (0, 1):
        a = np.zeros((len(halves), 9), np.uint32)
        a[:, 0] = (halves | (0x7C00 << 16)) if pos == 0 else ((halves << 16) | 0xFC00)
        a[:, 1] = a[:, 2] = 0xFC00 | (0x7C00 << 16)
        ray = np.zeros((len(halves), 6), f32); ray[:, 0] = 0.25; ray[:, 3:] = (1.0, 0.5, 0.5)
        a[:, 3:] = ray.view(np.uint32)
        got = N.debug_eval("cull_slab", a)
        # lo = x (pos 0): the slab is [x, inf): hit iff it reaches t > 0, always unless x = +inf or NaN; hi = x (pos 1): (-inf, x]: hit iff x > 0.25
        want = (hv < np.inf) if pos == 0 else (hv > 0.25)
        skip = np.isnan(hv) | ((np.abs(hv) < 2.0 ** -14) & (hv != 0))     # NaN and subnormals: never produced by sq_half_outward
        assert np.array_equal(got[~skip].astype(bool), want[~skip]), pos
    # 2. random boxes and rays against exact arithmetic
    n = 4000
    L = N.lib()
    lo = rng.standard_normal((n, 3)) * 2; hi = lo + np.abs(rng.standard_normal((n, 3))) * 10.0 ** rng.uniform(-3, 0.5, (n, 1))
    o = (rng.standard_normal((n, 3)) * 3).astype(f32)
    d = rng.standard_normal((n, 3)); d = (d / np.linalg.norm(d, axis=1)[:, None] * rng.uniform(0.6, 1.2, (n, 1))).astype(f32)
    # aim half of the rays at a box face so that hits and misses are both common and some are close calls
    tgt = lo + (hi - lo) * rng.random((n, 3)) + rng.standard_normal((n, 3)) * 10.0 ** rng.uniform(-6, -1, (n, 1))
    aim = rng.random(n) < 0.6
    dd = (tgt - o); dd = (dd / np.linalg.norm(dd, axis=1)[:, None]).astype(f32)
    d[aim] = dd[aim]
    a = np.zeros((n, 9), np.uint32)
    hb = np.zeros((n, 6))
    for i in range(n):
        for c in range(3):
            l16, h16 = L.sq_half_outward(float(lo[i, c]), 0), L.sq_half_outward(float(hi[i, c]), 1)
            a[i, c] = l16 | (h16 << 16)
            hb[i, c] = float(np.array([l16], np.uint16).view(np.float16)[0]); hb[i, 3 + c] = float(np.array([h16], np.uint16).view(np.float16)[0])
    a[:, 3:6] = o.view(np.uint32); a[:, 6:9] = d.view(np.uint32)
    got = N.debug_eval("cull_slab", a).astype(bool)
    with np.errstate(all="ignore"):
        df = (f32(1) / d); nodf = (-o * df)
    want = np.zeros(n, bool)
    for i in range(n):
        tl = [f32(float(Fraction(hb[i, c]) * Fraction(float(df[i, c])) + Fraction(float(nodf[i, c])))) for c in range(3)]
        th = [f32(float(Fraction(hb[i, 3 + c]) * Fraction(float(df[i, c])) + Fraction(float(nodf[i, c])))) for c in range(3)]
        tmin = max(min(x, y) for x, y in zip(tl, th)); tmax = min(max(x, y) for x, y in zip(tl, th))
        want[i] = tmax > 0 and tmin < tmax
    assert 0.2 < want.mean() < 0.8
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]


def test_leaf_culling_changes_no_bit_and_removes_most_triangle_tests(sqt, product_scene, oracle_scene, dev):
    """Option `cull` (a ray that misses a leaf's culling box skips the leaf: sq_cull_boxes) against the oracle, which tests every
    triangle of every leaf the reference visits; with the profile build's counters: the tests it saves."""
    import torch
    bih, cam, _ = product_scene
    ob, ocam, _ = oracle_scene
    w, h, n = 200, 150, 16
    o, o8, _ = ob.render(ocam, n, w, h, threads=THREADS)
    tested = {}
    try:
        for cull in (0, 1):
            for opts in ({"pool": 1, "profile": 1}, {"pool": 0, "profile": 0}, {"pool": 1, "profile": 0, "primary_resident": 0}, {"pool": 1, "profile": 0, "resident": 0}):
                for k, v in {**opts, "cull": cull}.items():
                    dev.set_option(k, v)
                dev.stats(reset=True)
                avg, rgb = dev.render_rows(cam, n, w, h)
                torch.cuda.synchronize()
                assert np.array_equal(bits(avg.cpu().numpy()), bits(o)), (cull, opts)
                assert np.array_equal(rgb.cpu().numpy(), o8), (cull, opts)
                if opts.get("profile"):
                    tested[cull] = dev.stats()[5]                     # pooled form, profile build: triangle tests run
    finally:
        for k, v in {"pool": 1, "profile": 0, "primary_resident": 1, "cull": 1, "resident": 1}.items():
            dev.set_option(k, v)
    assert tested[0] > 0 and tested[1] < 0.6 * tested[0], tested
