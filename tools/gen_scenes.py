#!/usr/bin/env python3
"""Procedural stand-ins for BASELINE.json configs[2] and configs[4] (no Stanford Bunny or other mesh
exists offline).  They emit .obj/.sq TEXT in the reference's dialect (src/Obj.hs:96-161) so they go
through the same loaders as data/scene.obj.  Deterministic: fixed seeds.

    blob_scene(subdiv)   closed displaced icosphere (subdiv 6 = 81 920 triangles) inside a Cornell room
    heightfield_scene(n) n x n jittered height-field (2 n^2 triangles; n = 708 -> 1 002 528) + walls + emitter
"""
import numpy as np

SQ_TEXT = (b"newmtl Diffuse\nreflective 0 0.700000 0.700000 0.700000\nemissive 0 0 0 0\n\n"
           b"newmtl Glossy\nreflective 0.2 0.500000 0.300000 0.200000\nemissive 0 0 0 0\n\n"
           b"newmtl Mirror\nreflective 1 0.900000 0.900000 0.900000\nemissive 0 0 0 0\n\n"
           b"newmtl Light\nreflective 0 0 0 0\nemissive 60 1 1 1\n")
CAMERA_TEXT = b"0 7 0.75\n1.5707963267948966 0 -0.09817477042468103\n"


def _fmt_obj(objects):
    """objects: list of (name, verts[n,3] in OBJ axes, material, faces[m,3] 0-based LOCAL) -> bytes.
    Face indices in the file are 1-based and GLOBAL (src/Obj.hs:76,83-85)."""
    out = [b"mtllib scene.sq\n"]
    base = 0
    for name, v, mtl, f in objects:
        out.append(b"o " + name.encode() + b"\n")
        out.append("".join("v %.6f %.6f %.6f\n" % (a, b, c) for a, b, c in np.asarray(v, np.float64)).encode())
        out.append(b"usemtl " + mtl.encode() + b"\ns off\n")
        g = np.asarray(f, np.int64) + base + 1
        out.append("".join("f %d %d %d\n" % (a, b, c) for a, b, c in g).encode())
        base += len(v)
    return b"".join(out)


def _room():
    """Open-front box [-2,2]^3 in scene axes (x, y=depth, z=up); OBJ axes are (x, z, y) because the loader swaps."""
    def quad(p0, p1, p2, p3):
        return np.array([p0, p1, p2, p3], np.float64), np.array([[0, 1, 2], [0, 2, 3]])
    S = 2.0
    parts = []
    # scene coords (x,y,z) -> obj (x,z,y)
    def o(p):
        return (p[0], p[2], p[1])
    walls = {
        "Floor": ([-S, -S, -S], [S, -S, -S], [S, S, -S], [-S, S, -S], "Diffuse"),
        "Ceiling": ([-S, -S, S], [S, -S, S], [S, S, S], [-S, S, S], "Diffuse"),
        "Back": ([-S, -S, -S], [S, -S, -S], [S, -S, S], [-S, -S, S], "Glossy"),
        "Left": ([-S, -S, -S], [-S, S, -S], [-S, S, S], [-S, -S, S], "Glossy"),
        "Right": ([S, -S, -S], [S, S, -S], [S, S, S], [S, -S, S], "Diffuse"),
        "Lamp": ([-0.6, -0.6, 1.98], [0.6, -0.6, 1.98], [0.6, 0.6, 1.98], [-0.6, 0.6, 1.98], "Light"),
    }
    for name, (a, b, c, d, m) in walls.items():
        v, f = quad(o(a), o(b), o(c), o(d))
        parts.append((name, v, m, f))
    return parts


def _icosphere(subdiv):
    t = (1 + 5 ** 0.5) / 2
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], np.int64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    for _ in range(subdiv):
        e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
        ue, inv = np.unique(e, axis=0, return_inverse=True)
        mid = v[ue[:, 0]] + v[ue[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        m = inv.reshape(3, -1) + len(v)
        v = np.concatenate([v, mid])
        a, b, c = f[:, 0], f[:, 1], f[:, 2]
        f = np.concatenate([np.stack([a, m[0], m[2]], 1), np.stack([b, m[1], m[0]], 1),
                            np.stack([c, m[2], m[1]], 1), np.stack([m[0], m[1], m[2]], 1)])
    return v, f


def blob_scene(subdiv=6, seed=7):
    """Displaced icosphere (20 * 4^subdiv triangles), mirror material, standing in the room."""
    rng = np.random.default_rng(seed)
    v, f = _icosphere(subdiv)
    k = rng.normal(size=(6, 3)); ph = rng.uniform(0, 6.28, 6)
    disp = sum(0.06 * np.sin(3.0 * (v @ k[i]) + ph[i]) for i in range(6))
    v = v * (0.95 + disp)[:, None]
    v = v * 1.1 + np.array([0.2, -0.3, -0.9])           # scene coords
    objs = _room() + [("Blob", v[:, [0, 2, 1]], "Mirror", f)]
    return _fmt_obj(objs), SQ_TEXT, CAMERA_TEXT


def heightfield_scene(n=708, seed=11):
    """n x n cells, 2 n^2 small triangles with jittered vertices; 80 % diffuse, 10 % glossy, 10 % mirror strips."""
    rng = np.random.default_rng(seed)
    g = np.linspace(-1.95, 1.95, n + 1)
    X, Y = np.meshgrid(g, g, indexing="ij")
    cell = 3.9 / n
    X = X + rng.uniform(-0.3, 0.3, X.shape) * cell
    Y = Y + rng.uniform(-0.3, 0.3, Y.shape) * cell
    Z = -1.6 + 0.25 * np.sin(2.1 * X) * np.cos(1.7 * Y) + 0.05 * np.sin(9 * X + 4 * Y) + rng.uniform(-0.2, 0.2, X.shape) * cell
    v = np.stack([X.ravel(), Z.ravel(), Y.ravel()], 1)   # OBJ axes (x, z, y)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    a = (i * (n + 1) + j).ravel(); b = a + 1; c = a + (n + 1); d = c + 1
    faces = np.concatenate([np.stack([a, b, d], 1), np.stack([a, d, c], 1)])
    band = (faces[:, 0] // (n + 1)) * 10 // n             # 10 bands along x
    objs = _room()
    # one object per material band; vertices are shared globally, so the first object carries them all
    names = {0: "Glossy", 5: "Mirror"}
    first = True
    for bnd in range(10):
        fb = faces[band == bnd]
        objs.append(("Field%d" % bnd, v if first else np.zeros((0, 3)), names.get(bnd, "Diffuse"), fb))
        first = False
    # faces index the height-field's vertex block, which starts after the room's vertices: _fmt_obj adds a
    # per-object base, so shift later objects' local indices back to the shared block
    room_nv = sum(len(o[1]) for o in _room())
    fixed = []
    base = 0
    for name, vv, m, ff in objs:
        if name.startswith("Field"):
            ff = ff + room_nv - base
        fixed.append((name, vv, m, ff))
        base += len(vv)
    return _fmt_obj(fixed), SQ_TEXT, CAMERA_TEXT


if __name__ == "__main__":
    import sys
    obj, sq, cam = (blob_scene() if (len(sys.argv) < 2 or sys.argv[1] == "blob") else heightfield_scene(int(sys.argv[2]) if len(sys.argv) > 2 else 708))
    sys.stdout.write("obj %d bytes, %d faces\n" % (len(obj), obj.count(b"\nf ")))
