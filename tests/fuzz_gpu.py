"""Randomised parity campaign on the GPU box: random scenes x random cameras x random frame shapes, the HIP
path against the CPU oracle, bit for bit (fp32 `avg` and RGB8).  Prints one line per mismatch with the seed
that reproduces it, and a summary.  Checker use of the oracle only (like tests/).

    python tests/fuzz_gpu.py [seconds=240] [first_seed=0] [big]

`big`: frames up to 320 x 320 at up to 64 spp, scenes up to 20 000 triangles (the streaming kernel form), and a
small sample-slot budget so that a frame takes several batches.
"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as O  # noqa: E402

O.lib()
sqt = importlib.import_module("squigly-trace_amd")
sqt.lib()
import torch  # noqa: E402

THREADS = min(os.cpu_count() or 1, 16)
BIG = "big" in sys.argv


def make_scene(rng, kinds=12):
    kind = rng.integers(0, kinds)            # `kinds` < 10 replays seeds found before the later kinds were added
    scale = float(rng.choice([1e-3, 1.0, 1.0, 1.0, 50.0, 1e4]))
    if kind == 0:      # soup
        n = int(rng.choice([3000, 8000, 20000] if BIG else [1, 2, 14, 15, 16, 40, 200, 1000, 3000]))
        c = rng.uniform(-2, 2, (n, 1, 3))
        v = c + rng.normal(0, float(rng.choice([0.02, 0.3, 1.5])), (n, 3, 3))
    elif kind == 1:    # lattice: exact ties everywhere (centroids, planes, edges, coplanar faces)
        k = int(rng.integers(2, 7))
        g = np.stack(np.meshgrid(np.arange(k), np.arange(k), np.arange(k), indexing="ij"), -1).reshape(-1, 1, 3).astype(np.float64)
        offs = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 0, 0], [0, 1, 0], [0, 0, 1]], [[0, 0, 0], [0, 0, 1], [1, 0, 0]]], np.float64)
        v = (g[:, None] + offs[None]).reshape(-1, 3, 3) - k / 2
        v = v[rng.permutation(len(v))[: int(rng.integers(1, len(v) + 1))]]
    elif kind == 2:    # duplicated and overlapping triangles (dist ties, COMBINE frames)
        n = int(rng.integers(1, 60))
        base = rng.uniform(-1.5, 1.5, (n, 3, 3))
        v = np.concatenate([base] * int(rng.integers(2, 6)))
        v = v[rng.permutation(len(v))]
    elif kind == 3:    # closed room of big quads + clutter (rays never escape: deep paths)
        r = 3.0
        q = []
        for ax in range(3):
            for s in (-r, r):
                a, b = [i for i in range(3) if i != ax]
                p = np.zeros((4, 3)); p[:, ax] = s
                p[:, a] = [-r, r, r, -r]; p[:, b] = [-r, -r, r, r]
                q += [p[[0, 1, 2]], p[[0, 2, 3]]]
        n = int(rng.integers(0, 400))
        c = rng.uniform(-2, 2, (n, 1, 3))
        v = np.concatenate([np.array(q), c + rng.normal(0, 0.25, (n, 3, 3))])
    elif kind == 4:    # slivers and degenerate triangles (|a| near the 1e-4 cut-off, zero area)
        n = int(rng.integers(20, 600))
        a = rng.uniform(-2, 2, (n, 1, 3))
        d = rng.normal(0, 1, (n, 1, 3))
        t = rng.uniform(-1, 1, (n, 3, 1))
        v = a + d * t + rng.normal(0, float(rng.choice([0, 1e-6, 1e-3])), (n, 3, 3))
    elif kind == 6:    # height-field strip: long thin leaves, grazing rays, shared vertices
        k = int(rng.integers(4, 40))
        xs, ys = np.meshgrid(np.linspace(-2, 2, k), np.linspace(-2, 2, k), indexing="ij")
        zs = 0.3 * np.sin(3 * xs + rng.uniform(0, 6)) * np.cos(2 * ys) + rng.normal(0, 0.02, xs.shape)
        p = np.stack([xs, ys, zs], -1)
        q = [np.stack([p[:-1, :-1], p[1:, :-1], p[:-1, 1:]], 2).reshape(-1, 3, 3), np.stack([p[1:, :-1], p[1:, 1:], p[:-1, 1:]], 2).reshape(-1, 3, 3)]
        lamp = np.array([[[-1, -1, 1.5], [1, -1, 1.5], [0, 1, 1.5]]])
        v = np.concatenate(q + [lamp])
    elif kind == 7:    # mirror room: every path runs to the bounce limit, mirror rays of mirror rays
        r = 2.0
        q = []
        for ax in range(3):
            for sgn in (-r, r):
                a, b = [i for i in range(3) if i != ax]
                p = np.zeros((4, 3)); p[:, ax] = sgn
                p[:, a] = [-r, r, r, -r]; p[:, b] = [-r, -r, r, r]
                q += [p[[0, 1, 2]], p[[0, 2, 3]]]
        n = int(rng.integers(0, 60))
        v = np.concatenate([np.array(q), rng.uniform(-1.5, 1.5, (n, 1, 3)) + rng.normal(0, 0.3, (n, 3, 3))])
    elif kind == 8:    # magnitudes that overflow or underflow inside the ray-triangle test (inf, NaN, denormals)
        n = int(rng.integers(15, 200))
        v = rng.uniform(-1, 1, (n, 1, 3)) + rng.normal(0, 0.4, (n, 3, 3))
        scale = float(rng.choice([1e-30, 1e-20, 1e12, 1e19, 3e37]))
    elif kind == 9:    # a few NaN / infinite vertex coordinates (only the host BIH build accepts them)
        n = int(rng.integers(15, 300))
        v = rng.uniform(-2, 2, (n, 1, 3)) + rng.normal(0, 0.4, (n, 3, 3))
        for _ in range(int(rng.integers(1, 6))):
            v[rng.integers(0, n), rng.integers(0, 3), rng.integers(0, 3)] = rng.choice([np.nan, np.inf, -np.inf])
    elif kind == 11:   # closed room + clutter with LARGE FINITE material values (see below): radiance overflows to inf / NaN
        r = 2.5
        q = []
        for ax in range(3):
            for sgn in (-r, r):
                a, b = [i for i in range(3) if i != ax]
                p = np.zeros((4, 3)); p[:, ax] = sgn
                p[:, a] = [-r, r, r, -r]; p[:, b] = [-r, -r, r, r]
                q += [p[[0, 1, 2]], p[[0, 2, 3]]]
        n = int(rng.integers(4, 120))
        v = np.concatenate([np.array(q), rng.uniform(-1.8, 1.8, (n, 1, 3)) + rng.normal(0, 0.5, (n, 3, 3))])
        scale = 1.0
    elif kind == 10:   # many identical triangles: terminal leaves of 32..200 (beyond what a packed leaf reference holds)
        base = rng.uniform(-1.5, 1.5, (int(rng.integers(1, 5)), 3, 3))
        v = np.concatenate([np.repeat(base[i:i + 1], int(rng.integers(20, 200)), 0) for i in range(len(base))] +
                           [rng.uniform(-2, 2, (int(rng.integers(0, 40)), 3, 3))])
    else:              # axis-aligned thin plates through the origin (rays parallel to slab planes, zeros of both signs)
        n = int(rng.integers(16, 300))
        v = rng.uniform(-2, 2, (n, 3, 3))
        ax = rng.integers(0, 3, n)
        for i in range(n):
            v[i, :, ax[i]] = float(rng.choice([0.0, -0.0, 1.0, -1.0]))
    v = (v * scale).astype(np.float32)
    nm = int(rng.integers(1, 6))
    mats = np.zeros(nm, sqt._native.MAT_DTYPE)
    mats["reflective"] = rng.choice([0.0, 0.0, 0.3, 1.0, 1.0], nm)
    mats["surf"] = rng.uniform(0, 1, (nm, 3))
    mats["emissive"] = rng.choice([0.0, 0.0, 1.0, 20.0], nm)
    mats["emit"] = rng.uniform(0, 1, (nm, 3))
    if kind == 7:
        mats["reflective"] = 1.0                     # all mirrors ...
        mats["reflective"][nm - 1] = 0.0             # ... but the light
    if rng.random() < 0.8:
        mats["emissive"][nm - 1] = 15.0              # most scenes have a light
    if rng.random() < 0.3:
        mats["surf"][nm - 1] = 0.0                   # a black surface (like the reference's lamp): absorbs every nested ray
    if rng.random() < 0.1:
        mats["surf"][0] = [0.0, -0.0, 0.0]           # a zero with its sign bit set is "negative" to the shortcut test
    if rng.random() < 0.1:
        mats["reflective"][0] = float(rng.choice([2.0, -1.0, np.nan, 0.5]))
    if rng.random() < 0.1:
        mats["surf"][0, 0] = -0.5                    # negative component: the exact `== 0` shortcuts are off
    if rng.random() < 0.05:
        mats["emit"][0, 1] = np.inf                  # non-finite material: the emitter cull is off
    if kind == 11:
        # Every component is finite and <= 3e38, but the products are not: `emissive *^ emitColor` (src/Lib.hs:136) and
        # `surfColor * L` overflow, and an absorbing surface (surfColor 0) in front of such radiance gives 0 * inf = NaN
        # (src/Lib.hs:135), which the `surfColor == 0` shortcut must not replace by +0.
        nm = int(rng.integers(2, 6))
        mats = np.zeros(nm, sqt._native.MAT_DTYPE)
        big = lambda size: (10.0 ** rng.uniform(18, 38.47, size)).astype(np.float32)
        mats["reflective"] = rng.choice([0.0, 0.0, 0.3, 1.0], nm)
        mats["surf"] = np.where(rng.random((nm, 3)) < 0.5, big((nm, 3)), rng.uniform(0, 1, (nm, 3)))
        mats["emissive"] = rng.choice([0.0, 1.0, 1e10, 1e20, 1e30, 3e38], nm)
        mats["emit"] = np.where(rng.random((nm, 3)) < 0.5, big((nm, 3)), rng.uniform(0, 1, (nm, 3)))
        mats["surf"][0] = 0.0                        # the absorbing surface ...
        mats["emissive"][0] = float(rng.choice([0.0, 5.0]))
        mats["emissive"][nm - 1] = float(rng.choice([1e20, 1e30, 3e38, 7.0]))   # ... and the (mostly overflowing) emitter
    mat = rng.integers(0, nm, len(v))
    if rng.random() < 0.15 and len(v) > 70:          # around the 64-emitter limit of the last-bounce emitter test
        mat[:] = rng.integers(0, max(nm - 1, 1), len(v))
        mats["emissive"][: nm - 1] = 0.0
        mats["emissive"][nm - 1] = 9.0
        mat[rng.choice(len(v), int(rng.choice([1, 63, 64, 65])), replace=False)] = nm - 1
    return v, mats, mat, scale


def _fixed(x):
    """Decimal text without exponent (the reference grammar has none), exact for the float32 it denotes."""
    t = "%.40f" % float(np.float32(x))
    return t.rstrip("0") + "0"


def make_camera(rng, scale):
    """A few candidate poses; the one whose central ray points best at the scene centre wins, so that most
    cases see geometry (a quarter are left to chance: rays that miss everything are a case too)."""
    mode = rng.integers(0, 4)
    best = None
    for _ in range(1 if rng.random() < 0.25 else 8):
        if mode == 0:
            pos = rng.uniform(-4, 4, 3) * scale
            ang = rng.uniform(-3.2, 3.2, 3)
        elif mode == 1:    # exact axis-aligned view from a lattice point
            pos = rng.integers(-3, 4, 3).astype(np.float64) * scale
            ang = rng.integers(-2, 3, 3) * (np.pi / 2)
        elif mode == 2:    # from the origin / on box planes
            pos = np.zeros(3)
            ang = rng.uniform(-3.2, 3.2, 3)
        else:
            pos = rng.uniform(-0.5, 0.5, 3) * scale
            ang = np.array([0.0, 0.0, 0.0]) if rng.random() < 0.3 else rng.uniform(-3.2, 3.2, 3)
        text = (" ".join(_fixed(p) for p in pos) + "\n" + " ".join(_fixed(a) for a in ang) + "\n").encode()
        cam = O.camera_from_text(text)
        _, d = O.make_ray(8, 8, 4, 4, cam)
        to_centre = -np.asarray(pos, np.float64)
        score = float(np.dot(d, to_centre) / (np.linalg.norm(d) * (np.linalg.norm(to_centre) + 1e-30)))
        if best is None or score > best[0]:
            best = (score, text)
    return best[1]


def canon(a):
    """uint32 view with every NaN mapped to one pattern (IEEE leaves NaN payloads to the implementation)."""
    u = np.ascontiguousarray(a).view(np.uint32).copy()
    u[np.isnan(a)] = 0x7FC00000
    return u


def run_case(seed, kinds=12):
    rng = np.random.default_rng(seed)
    v, mats, mat, scale = make_scene(rng, kinds)
    camt = make_camera(rng, scale)
    w, h = (int(rng.integers(40, 321)), int(rng.integers(40, 321))) if BIG else (int(rng.integers(1, 49)), int(rng.integers(1, 49)))
    spp = int(rng.choice([5, 16, 33, 64] if BIG else [1, 2, 3, 7, 16, 40]))
    cast = bool(rng.random() < 0.15)
    tris = np.zeros(len(v), sqt._native.TRI_DTYPE)
    tris["v0"], tris["v1"], tris["v2"], tris["mat"] = v[:, 0], v[:, 1], v[:, 2], mat
    mesh = sqt.Mesh.from_arrays(tris, mats)
    on_device = bool(rng.random() < 0.5)
    if on_device and not np.isfinite(v).all():
        try:
            sqt.BIH(mesh, device=0)
            return "the device BIH build accepted non-finite vertices"
        except sqt.SquiglyError:
            on_device = False
    bih = sqt.BIH(mesh, device=0 if on_device else None)
    if on_device:                                       # the GPU build returns the host build's arrays, bit for bit
        host = sqt.BIH(mesh)
        raw = lambda a: np.ascontiguousarray(a).view(np.uint8)
        if not (np.array_equal(raw(bih.nodes), raw(host.nodes)) and np.array_equal(raw(bih.tris), raw(host.tris))
                and np.array_equal(raw(bih.bounds), raw(host.bounds))):
            return "the device BIH build differs from the host build"
    ot = np.zeros(len(v), O.TRI_DTYPE)
    ot["a"], ot["b"], ot["c"] = v[:, 0], v[:, 1], v[:, 2]
    for f in ("reflective", "surf", "emissive", "emit"):
        ot[f] = mats[f][mat]
    ob = O.BIH(ot)
    if (bih.height, bih.num_leaves, bih.longest_leaf) != (ob.height, ob.num_leaves, ob.longest_leaf):
        return "tree shape differs (device build)" if on_device else "tree shape differs"
    cam_p, cam_o = sqt.camera_from_text(camt), O.camera_from_text(camt)
    o, o8, _ = ob.render(cam_o, spp, w, h, cast=cast, threads=THREADS)
    ds = sqt.DeviceScene(bih, 0)
    # trace-kernel form and its tunables: pooled or one ray per lane, scene in LDS or streamed; no setting may change a bit
    knobs = {"pool": int(rng.integers(0, 2)), "resident": int(rng.integers(0, 2)), "refill_min": int(rng.choice([1, 8, 12, 33, 64])),
             "flush_min": int(rng.choice([0, 1, 40, 64])), "guided": int(rng.integers(0, 2)),
             "primary_resident": int(rng.integers(0, 2)), "pixel_major": int(rng.integers(0, 2)),
             "cull": int(rng.random() < 0.8), "descend_extra": int(rng.choice([0, 1, 2, 5])), "descend_lanes": int(rng.choice([1, 16, 40])),
             # from the seed, not from rng: the draws that follow stay what they were for every recorded seed.  Campaign seeds only
             # (>= 2e6): the slice in `pytest -m gpu` (seeds 1000..1399, 504773) keeps the knobs it has always had
             "primary_pooled": int(seed >= 2000000 and ((seed * 2654435761) >> 7) % 4 == 0)}
    if seed >= 30000000:   # round-3 campaign seeds: the streaming form's plain (1, 2 workgroups per CU) and six-wave (0 = auto, 3) builds
        knobs["trace_blocks_per_cu"] = ((seed * 40503) >> 3) % 4
        knobs["guided"] = ((seed * 2654435761) >> 11) % 4          # per bounce level since round 3
        if BIG:            # the overlapped schedules' scheduling options (BIG draws `overlap` below)
            knobs["aux_polite"] = ((seed * 2246822519) >> 5) % 3
            knobs["trace_prio"] = (((seed * 3266489917) >> 4) % 2) * 2
    if seed >= 40000000:   # second round-3 campaign (merged branch records in the streaming form, one-region culling tests, ballot builtin)
        knobs["lds_node_kb"] = (0, 1, 4, 32)[((seed * 40503) >> 5) % 4]
    for kv in os.environ.get("SQ_FUZZ_FORCE", "").split(","):          # e.g. SQ_FUZZ_FORCE=primary_pooled=1,cull=0
        if "=" in kv:
            knobs[kv.split("=")[0]] = int(kv.split("=")[1])
    if os.environ.get("SQ_FUZZ_VERBOSE"):
        print(f"seed {seed}: {len(v)} tris, {w}x{h} @ {spp}, cast {cast}, {knobs}", flush=True)
    try:
        for k, val in knobs.items():
            ds.set_option(k, val)
        for variant in (2, 1):
            ds.set_option("variant", variant)
            if BIG:
                ds.set_option("overlap", int(rng.integers(0, 3)))
                ds.set_option("slots", int(rng.choice([w * h, 3 * w * h + 17, 1 << 22])))
            a, r = ds.render_rows(cam_p, spp, w, h, cast=cast)
            torch.cuda.synchronize()
            a, r = a.cpu().numpy(), r.cpu().numpy()
            if not np.array_equal(canon(a), canon(o)):
                bad = int((canon(a) != canon(o)).any(-1).sum())
                return f"avg differs in {bad}/{w * h} pixels (variant {variant}, cast {cast}, tris {len(v)}, spp {spp}, {w}x{h}, {knobs})"
            if not np.array_equal(r, o8):
                return f"rgb8 differs (variant {variant})"
        ds.set_option("variant", 2)
        # a random shard of the same frame (interleaved row blocks) equals those rows of the whole frame
        if rng.random() < 0.3 and w > 1:
            rb, ns = int(rng.choice([1, 2, 3, 8, 16])), int(rng.integers(2, 6))
            si = int(rng.integers(0, ns))
            a, r = ds.render_rows(cam_p, spp, w, h, cast=cast, shard=(rb, si, ns))
            torch.cuda.synchronize()
            sh = sqt.Shard(rb, si, ns)
            rows = [sqt.lib().sq_shard_global_row(j, sh) for j in range(a.shape[0])]
            if not np.array_equal(canon(a.cpu().numpy()), canon(o[rows])) or not np.array_equal(r.cpu().numpy(), o8[rows]):
                return f"shard ({rb},{si},{ns}) differs from the rows {rows[:4]}... of the whole frame"
        # one row of a frame so tall that the sample seeds n*(x + y*w) pass 2^32 (and 2^40)
        if rng.random() < 0.2:
            wv = int(rng.choice([70000, 300000, 2000000]))
            y = int(rng.integers(wv // 2, wv))
            a, r = ds.render_rows(cam_p, spp, wv, h, cast=cast, shard=(1, y, wv))
            torch.cuda.synchronize()
            ov, ov8, _ = ob.render(cam_o, spp, wv, h, cast=cast, threads=1, rows=(y, y + 1))
            if a.shape[0] != 1 or not np.array_equal(canon(a.cpu().numpy()), canon(ov)) or not np.array_equal(r.cpu().numpy(), ov8):
                return f"row {y} of a {wv}-row frame differs (64-bit seeds)"
    finally:
        ds.close()
    return None


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    n = bad = 0
    last = t0
    while time.time() - t0 < budget:
        msg = run_case(seed)
        n += 1
        if msg:
            bad += 1
            print(f"MISMATCH seed={seed}: {msg}", flush=True)
        if time.time() - last > 30:
            last = time.time()
            print(f"... {n} cases, {bad} mismatches, next seed {seed + 1}", flush=True)
        seed += 1
    print(f"fuzz: {n} cases in {time.time() - t0:.0f} s, {bad} mismatches (seeds {seed - n}..{seed - 1})", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
