"""Differential fuzz of the text loaders on the CPU: the product's C++ loader (sq_mesh_from_text,
sq_camera_from_text) against the oracle's independent C restatement of src/Obj.hs, on random texts built from
the reference grammar plus random mutations.  Both must accept and reject the same byte strings, and produce the
same triangles, materials and camera matrices bit for bit.

    python tests/fuzz_loader.py [seconds=60] [first_seed=0]
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


def number(rng):
    if rng.random() < 0.01:
        return str(rng.choice(["1e5", ".5", "5.", "+1", "1,5", "0x10", "nan", "inf", "--1", "1.2.3"]))  # not in the grammar
    style = rng.integers(0, 6)
    if style == 0:
        return str(int(rng.integers(-5, 6)))
    if style == 1:
        return "%.6f" % rng.uniform(-3, 3)
    if style == 2:
        return "-0.000000"
    if style == 3:
        digits = lambda k: "".join(str(int(d)) for d in rng.integers(0, 10, k))
        return digits(int(rng.integers(1, 30))) + "." + digits(int(rng.integers(1, 30)))
    if style == 4:
        return "0." + "0" * int(rng.integers(30, 60)) + str(int(rng.integers(1, 999)))          # underflows to a denormal or zero
    if style == 5:
        return str(int(rng.integers(1, 9))) + "0" * int(rng.integers(30, 50)) + ".0"               # overflows to Infinity
    return "%.*f" % (int(rng.integers(0, 12)), rng.normal(0, 10))


def make_sq(rng):
    names = ["A", "B", "Mat.001", "x_y", "A"]          # a repeated name duplicates triangles (src/Obj.hs:75)
    out = []
    for _ in range(int(rng.integers(0, 6))):
        sep = rng.choice([" ", " ", "  ", "\t", "\n"])
        out.append("newmtl " + str(rng.choice(names)) + "\n" + "reflective " + number(rng) + sep + sep.join(number(rng) for _ in range(3)) + "\n" +
                   "emissive " + number(rng) + " " + " ".join(number(rng) for _ in range(3)) + str(rng.choice(["\n", "\n\n", "\r\n", ""])))
    return "".join(out)


def make_obj(rng):
    nl = str(rng.choice(["\n", "\n", "\r\n"]))
    out = ["mtllib " + str(rng.choice(["s.sq", "scene.sq", "scene.sq", "a b"])) + nl] if rng.random() < 0.98 else []
    nv = 0
    for _ in range(int(rng.integers(0, 4))):
        out.append("o " + (str(rng.choice(["", "bad-name"])) if rng.random() < 0.03 else str(rng.choice(["Cube", "Cube.001", "a_b", "X9"]))) + nl)
        k = int(rng.integers(0, 8))
        for _ in range(k):
            out.append("v " + " ".join(number(rng) for _ in range(3)) + nl)
        nv += k
        if rng.random() < 0.98:
            out.append("usemtl " + str(rng.choice(["A", "B", "Mat.001", "x_y", "A", "missing"])) + nl)
        if rng.random() < 0.5:
            out.append((str(rng.choice(["s 1", "s  off"])) if rng.random() < 0.04 else str(rng.choice(["s off", "s on"]))) + nl)
        for _ in range(int(rng.integers(0, 5))):
            idx = [str(int(rng.integers(0 if rng.random() < 0.01 else 1, max(nv, 1) + (2 if rng.random() < 0.01 else 1)))) for _ in range(3)]
            if rng.random() < 0.01:
                idx[0] += "/1"
            out.append("f " + " ".join(idx) + nl)
    return "".join(out)


def mutate(rng, text):
    b = bytearray(text.encode())
    for _ in range(int(rng.integers(0, 3))):
        if not b:
            break
        i = int(rng.integers(0, len(b)))
        op = rng.integers(0, 3)
        if op == 0:
            del b[i]
        elif op == 1:
            b.insert(i, int(rng.choice(list(b" \n\t-.0123456789vfosxe#"))))
        else:
            b[i] = int(rng.choice(list(b" \n\t-.0123456789vfosxe#")))
    return bytes(b)


def outcome(fn):
    try:
        return fn(), None
    except (sqt.SquiglyError, O.OracleError) as e:
        return None, str(e)


def run_case(seed):
    rng = np.random.default_rng(seed)
    obj, sq = make_obj(rng), make_sq(rng)
    if rng.random() < 0.25:
        obj = mutate(rng, obj)
    else:
        obj = obj.encode()
    sq = mutate(rng, sq) if rng.random() < 0.15 else sq.encode()
    p, pe = outcome(lambda: sqt.Mesh.from_text(obj, sq))
    o, oe = outcome(lambda: O.tris_from_text(obj, sq))
    if (p is None) != (o is None):
        return f"loader: product {'rejects: ' + pe if p is None else 'accepts'}, oracle {'rejects: ' + oe if o is None else 'accepts'}\nOBJ={obj!r}\nSQ={sq!r}"
    if p is not None:
        if len(p) != len(o):
            return f"loader: {len(p)} vs {len(o)} triangles\nOBJ={obj!r}\nSQ={sq!r}"
        if len(o):
            t, m = p.tris, p.materials
            same = all(np.array_equal(np.ascontiguousarray(x).view(np.uint32), np.ascontiguousarray(y).view(np.uint32)) for x, y in
                       ((t["v0"], o["a"]), (t["v1"], o["b"]), (t["v2"], o["c"]), (m["reflective"][t["mat"]], o["reflective"]),
                        (m["surf"][t["mat"]], o["surf"]), (m["emissive"][t["mat"]], o["emissive"]), (m["emit"][t["mat"]], o["emit"])))
            if not same:
                return f"loader: different triangle data\nOBJ={obj!r}\nSQ={sq!r}"
    cam = (" ".join(number(rng) for _ in range(3)) + str(rng.choice(["\n", " ", "\r\n"])) + " ".join(number(rng) for _ in range(int(rng.choice([3, 3, 3, 2]))))).encode()
    if rng.random() < 0.3:
        cam = mutate(rng, cam.decode())
    pc, pce = outcome(lambda: sqt.camera_from_text(cam))
    oc, oce = outcome(lambda: O.camera_from_text(cam))
    if (pc is None) != (oc is None):
        return f"camera: product {'rejects' if pc is None else 'accepts'}, oracle {'rejects' if oc is None else 'accepts'}: {cam!r}"
    if pc is not None:
        a = np.array(list(pc.pos) + list(pc.rot), np.float32).view(np.uint32)
        b = np.array([oc.pos.x, oc.pos.y, oc.pos.z] + list(oc.rot), np.float32).view(np.uint32)
        nan = np.isnan(a.view(np.float32)) & np.isnan(b.view(np.float32))
        if not np.array_equal(a[~nan], b[~nan]):
            return f"camera: different matrices for {cam!r}"
    return None


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    n = bad = 0
    while time.time() - t0 < budget:
        msg = run_case(seed)
        n += 1
        if msg:
            bad += 1
            if bad <= 10:
                print(f"MISMATCH seed={seed}: {msg}", flush=True)
        seed += 1
    print(f"loader fuzz: {n} cases in {time.time() - t0:.0f} s, {bad} mismatches", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
