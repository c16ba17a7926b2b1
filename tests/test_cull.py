"""Leaf culling (sq_cull_boxes, include/squigly_host.h): the lemma behind it, searched for counter-examples on the CPU.

Claim: a ray inside the stated limits for which the reference's fp32 mollerTrumbore (src/Geometry.hs:117-142) ACCEPTS a
triangle passes the fp32 slab test of the culling box of that triangle's leaf -- in the binary16 encoding the resident
kernels keep it in, too.  mollerTrumbore is restated here in numpy float32 (one rounding per operation, no FMA: the
reference's expression tree), the slab test as the kernels compute it (l * (1/d) + (-o/d), one rounding per plane value).
Rays: uniform ones, and grazing ones aimed just outside the edges and corners of a triangle with a determinant barely
above the test's epsilon -- where rounding moves the accepted region furthest.
"""
import importlib
import os

import numpy as np
import pytest

sqt = importlib.import_module("squigly-trace_amd")
N = importlib.import_module("squigly-trace_amd._native")
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")
f32 = np.float32


def mt_accepts(o, d, v0, v1, v2):
    """mollerTrumbore in float32, vectorised over rows; the reference's operation order (src/V3.hs dot/cross)."""
    def dot(a, b): return (a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]) + a[:, 2] * b[:, 2]
    def cross(a, b): return np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2], a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], 1)
    eps = f32(0.0001)
    e1, e2 = v1 - v0, v2 - v0
    h = cross(d, e2)
    a = dot(e1, h)
    with np.errstate(all="ignore"):
        f = f32(1) / a
        s = o - v0
        u = f * dot(s, h)
        q = cross(s, e1)
        v = f * dot(d, q)
        t = f * dot(e2, q)
        return ~((a > -eps) & (a < eps)) & ~((u < 0) | (u > 1)) & ~((v < 0) | (u + v > 1)) & (t > eps), a


def fma32(a, b, c):
    """fl32(a * b + c) with ONE rounding, for float32 arrays -- exactly what v_fma_f32 / v_fma_mix_f32 return.
    a * b is exact in binary64 (24 + 24 bits); the sum is taken with TwoSum, and the binary64 sum is turned into the
    round-to-odd value of the exact result (if inexact, whichever neighbour has an odd last bit), which then rounds to
    binary32 as the exact value would (53 >= 24 + 2: no double rounding).  x86 long double (64-bit significand, the first
    version of this test) is NOT enough: its own rounding can land on a binary32 midpoint."""
    p = a.astype(np.float64) * b.astype(np.float64)
    c = np.broadcast_to(c.astype(np.float64), p.shape)
    with np.errstate(all="ignore"):
        s = p + c
        bb = s - p
        err = (p - (s - bb)) + (c - bb)                      # exact: s + err == p + c
    fin = np.isfinite(s) & np.isfinite(err) & (err != 0)
    bits = s.view(np.int64).copy() if s.flags.writeable else s.copy().view(np.int64)
    even = (bits & 1) == 0
    toward_larger_magnitude = (err > 0) == (s > 0)           # exact value lies on this side of s
    adj = fin & even
    # for s == 0 (cannot happen with err != 0: the sum of two finite doubles that rounds to 0 is exact) nothing to do
    step = np.where(toward_larger_magnitude, 1, -1)
    bits = np.where(adj, bits + step, bits)
    return bits.view(np.float64).astype(np.float32)


def slab_passes(box, o, d):
    """The kernels' culling slab test: plane value = fma(l, 1/d, -o * (1/d)) with one rounding, v_min/v_max, tmax > 0 && tmin < tmax."""
    with np.errstate(all="ignore"):
        df = f32(1) / d
        nodf = -o * df
        tl = fma32(box[:, 0:3], df, nodf)
        th = fma32(box[:, 3:6], df, nodf)
    tmin = np.minimum(tl, th).max(1)
    tmax = np.maximum(tl, th).min(1)
    return (tmax > 0) & (tmin < tmax)


def test_fma32_is_a_single_rounding():
    """Against exact rational arithmetic, on random operands and on operands built so that a * b + c sits within a few
    binary64 ulps of a binary32 rounding midpoint (where long double or plain binary64 evaluation double-rounds)."""
    from fractions import Fraction
    rng = np.random.default_rng(3)
    n = 4000
    a = (rng.standard_normal(n) * 10.0 ** rng.uniform(-3, 3, n)).astype(f32)
    b = (rng.standard_normal(n) * 10.0 ** rng.uniform(-3, 3, n)).astype(f32)
    c = (rng.standard_normal(n) * 10.0 ** rng.uniform(-3, 3, n)).astype(f32)
    # midpoints: pick a float32 r and its successor, m = their mean; c := the float32 nearest to m - a*b
    r = (rng.standard_normal(n) * 10.0 ** rng.uniform(-2, 2, n)).astype(f32)
    mid = (r.astype(np.float64) + np.nextafter(r, f32(np.inf)).astype(np.float64)) / 2
    a2 = (rng.standard_normal(n)).astype(f32); b2 = (rng.standard_normal(n) * 1e-4).astype(f32)
    c2 = (mid - a2.astype(np.float64) * b2.astype(np.float64)).astype(f32)
    A, B, Cc = np.concatenate([a, a2]), np.concatenate([b, b2]), np.concatenate([c, c2])
    got = fma32(A, B, Cc)

    def rn32(q):                                             # correctly rounded binary32 of a Fraction
        if q == 0:
            return f32(0)
        lo = f32(float(q))                                   # float(q) is correctly rounded to binary64: within half a binary32 ulp of the answer
        cands = sorted({float(lo), float(np.nextafter(lo, f32(-np.inf))), float(np.nextafter(lo, f32(np.inf)))})
        best = min(cands, key=lambda v: (abs(Fraction(v) - q), int(np.array([v], f32).view(np.uint32)[0]) & 1))
        return f32(best)
    for i in range(len(A)):
        want = rn32(Fraction(float(A[i])) * Fraction(float(B[i])) + Fraction(float(Cc[i])))
        assert got[i] == want, (i, A[i], B[i], Cc[i], got[i], want)


def half_box(box):
    """The binary16 encoding (sq_half_outward), back as float32."""
    L = N.lib()
    flat = box.reshape(-1, 6)
    out = np.empty_like(flat)
    cache = {}
    for i in range(flat.shape[0]):
        for c in range(6):
            key = (float(flat[i, c]), c >= 3)
            if key not in cache:
                cache[key] = np.array([L.sq_half_outward(float(flat[i, c]), int(c >= 3))], np.uint16).view(np.float16)[0]
            out[i, c] = f32(cache[key])
    return out


def test_half_outward_is_outward_tight_and_never_subnormal():
    L = N.lib()
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.standard_normal(4000).astype(f32) * f32(3), (rng.standard_normal(2000) * 1e-5).astype(f32), (rng.standard_normal(500) * 1e5).astype(f32),
                         np.array([0.0, -0.0, 65504.0, 65505.0, -65504.0, 7e4, -7e4, 6.1035156e-05, 6.0e-05, -6.0e-05, 1e-30, -1e-30, np.inf, -np.inf, 1.0, 1.0009765625, 2047.5], f32)])
    halves = np.arange(0x10000, dtype=np.uint16).view(np.float16)
    ok = np.isfinite(halves.astype(np.float64)) | np.isinf(halves.astype(np.float64))
    allowed = np.unique(halves[ok & ((np.abs(halves.astype(np.float64)) >= 2.0 ** -14) | (halves == 0))].astype(np.float64))
    for x in xs:
        for up in (0, 1):
            bits = L.sq_half_outward(float(x), up)
            v = float(np.array([bits], np.uint16).view(np.float16)[0])
            assert (bits & 0x7C00) != 0 or (bits & 0x3FF) == 0, (x, up, hex(bits))            # not subnormal
            want = allowed[allowed >= float(x)].min() if up else allowed[allowed <= float(x)].max()
            assert v == want, (float(x), up, v, want)
    assert L.sq_half_outward(float("nan"), 1) == 0x7C00 and L.sq_half_outward(float("nan"), 0) == 0xFC00


def _leaf_table(bih):
    nodes, tris = bih.nodes, bih.tris
    boxes, lim = bih.cull_boxes()
    leaf_of_tri = np.empty(len(tris), np.int64)
    for i, nd in enumerate(nodes):
        if nd["kind"] & 3 == 3:
            leaf_of_tri[nd["link"]: nd["link"] + (nd["kind"] >> 2)] = i
    return nodes, tris, boxes, lim, leaf_of_tri


def _grazing_rays(rng, tris, n, omax, eps_scale):
    """Rays nearly parallel to a triangle's plane, aimed at a point just outside (or inside) one of its edges or corners."""
    k = rng.integers(0, len(tris), n)
    v0, v1, v2 = (tris[f][k].astype(np.float64) for f in ("v0", "v1", "v2"))
    e1, e2 = v1 - v0, v2 - v0
    nrm = np.cross(e1, e2)
    area2 = np.linalg.norm(nrm, axis=1)
    good = area2 > 1e-12
    nrm = nrm / np.maximum(area2, 1e-300)[:, None]
    # target point: barycentric (u, v) on an edge or a corner, pushed outwards by a log-uniform amount
    kind = rng.integers(0, 6, n)
    lam = rng.random(n)
    bu = np.select([kind == 0, kind == 1, kind == 2, kind == 3, kind == 4, kind == 5], [lam, 0 * lam, lam, 0 * lam, 1 + 0 * lam, 0 * lam])
    bv = np.select([kind == 0, kind == 1, kind == 2, kind == 3, kind == 4, kind == 5], [0 * lam, lam, 1 - lam, 0 * lam, 0 * lam, 1 + 0 * lam])
    push = 10.0 ** rng.uniform(-9, -1.5, n) * rng.choice([-1.0, 1.0], n)
    cen = (v0 + v1 + v2) / 3
    tgt = v0 + bu[:, None] * e1 + bv[:, None] * e2
    out = tgt - cen
    out /= np.maximum(np.linalg.norm(out, axis=1), 1e-300)[:, None]
    tgt = tgt + push[:, None] * out
    # direction: in-plane unit vector tilted so that |a| = |d . (e1 x e2)| is a small multiple of the test's epsilon
    ang = rng.uniform(0, 2 * np.pi, n)
    t1 = e1 / np.maximum(np.linalg.norm(e1, axis=1), 1e-300)[:, None]
    t2 = np.cross(nrm, t1)
    inpl = np.cos(ang)[:, None] * t1 + np.sin(ang)[:, None] * t2
    want_a = 1e-4 * 10.0 ** rng.uniform(0, eps_scale, n) * rng.choice([-1.0, 1.0], n)
    tilt = np.clip(want_a / np.maximum(area2, 1e-300), -0.9, 0.9)
    d = inpl * np.sqrt(1 - tilt ** 2)[:, None] + tilt[:, None] * nrm
    d *= rng.uniform(0.8, 1.2, n)[:, None]                      # |d| in [0.8, 1.2]: inside the limits
    dist = rng.uniform(0.01, 1.0, n) * omax
    o = tgt - dist[:, None] * d / np.linalg.norm(d, axis=1)[:, None]
    return k[good], o[good].astype(f32), d[good].astype(f32)


@pytest.mark.parametrize("scene", ["scene.obj", "big"])
def test_accepted_triangles_pass_their_leafs_culling_box(scene):
    rng = np.random.default_rng(20261004)
    if scene == "scene.obj":
        bih = sqt.BIH(sqt.Mesh.from_obj(os.path.join(DATA, "scene.obj"), DATA))
    else:   # large, long and sliver triangles at coordinates up to 40: margins grow with size, distance and coordinate
        n = 3000
        base = rng.uniform(-40, 40, (n, 3))
        tr = np.zeros(n, N.TRI_DTYPE)
        tr["v0"] = base
        tr["v1"] = base + rng.standard_normal((n, 3)) * 10.0 ** rng.uniform(-2, 0.6, (n, 1))
        tr["v2"] = base + rng.standard_normal((n, 3)) * 10.0 ** rng.uniform(-2, 0.6, (n, 1))
        mats = np.zeros(1, N.MAT_DTYPE)
        bih = sqt.BIH(sqt.Mesh.from_arrays(tr, mats))
    nodes, tris, boxes, (o2max, d2min, d2max), leaf_of_tri = _leaf_table(bih)
    assert o2max > 0
    omax = np.sqrt(o2max)
    hboxes = half_box(boxes)
    assert np.all(hboxes[:, :3] <= boxes[:, :3]) and np.all(hboxes[:, 3:] >= boxes[:, 3:])
    finite = np.isfinite(boxes).all(1)
    is_leaf = (nodes["kind"] & 3) == 3
    assert finite[is_leaf].mean() > 0.5                         # the lemma must not be vacuous
    total_acc = 0
    worst = 0.0
    ancestors_checked = 0
    parent = np.full(len(nodes), -1, np.int64)               # pre-order: left child = i + 1, right child = link
    for i, nd in enumerate(nodes):
        if nd["kind"] & 3 != 3:
            parent[i + 1] = i; parent[nd["link"]] = i
    inner = ~is_leaf
    assert np.all(boxes[inner][:, :3] <= np.minimum(boxes[np.flatnonzero(inner) + 1][:, :3], boxes[nodes["link"][inner]][:, :3]))   # a branch's box contains its children's
    assert np.all(boxes[inner][:, 3:] >= np.maximum(boxes[np.flatnonzero(inner) + 1][:, 3:], boxes[nodes["link"][inner]][:, 3:]))
    for rounds in range(3):
        ks, os_, ds = [], [], []
        for eps_scale in (0.05, 0.5, 2.0):
            k, o, d = _grazing_rays(rng, tris, 150_000, 0.999 * omax, eps_scale)
            ks.append(k); os_.append(o); ds.append(d)
        # uniform rays towards random points of random triangles
        n = 150_000
        k = rng.integers(0, len(tris), n)
        w = rng.dirichlet([1, 1, 1], n)
        tgt = w[:, :1] * tris["v0"][k] + w[:, 1:2] * tris["v1"][k] + w[:, 2:] * tris["v2"][k]
        o = rng.standard_normal((n, 3)); o *= (rng.random(n) ** (1 / 3) * 0.999 * omax / np.linalg.norm(o, axis=1))[:, None]
        d = tgt - o; d *= (rng.uniform(0.6, 1.2, n) / np.maximum(np.linalg.norm(d, axis=1), 1e-30))[:, None]
        ks.append(k); os_.append(o.astype(f32)); ds.append(d.astype(f32))
        k = np.concatenate(ks); o = np.concatenate(os_); d = np.concatenate(ds)
        oo = (o.astype(np.float64) ** 2).sum(1); dd = (d.astype(np.float64) ** 2).sum(1)
        with np.errstate(all="ignore"):
            inside = (oo <= o2max * (1 - 1e-6)) & (dd >= d2min * 1.001) & (dd <= d2max * 0.999) & np.isfinite(f32(1) / d).all(1)
        k, o, d = k[inside], o[inside], d[inside]
        acc, a = mt_accepts(o, d, tris["v0"][k], tris["v1"][k], tris["v2"][k])
        k, o, d, a = k[acc], o[acc], d[acc], a[acc]
        total_acc += len(k)
        leaf = leaf_of_tri[k]
        for bx, name in ((boxes, "fp32"), (hboxes, "binary16")):
            ok = slab_passes(bx[leaf], o, d)
            assert ok.all(), (name, int((~ok).sum()), o[~ok][:3], d[~ok][:3], k[~ok][:3])
            # ... and the box of EVERY ancestor (the union of its subtree's boxes): trav_descend skips a whole subtree on a
            # miss of its box, before any frame is pushed, so an accepted ray must pass all of them on the way down
            cur = leaf.copy()
            while True:
                up = parent[cur]
                live = up >= 0
                if not live.any():
                    break
                cur = np.where(live, up, cur)
                ok = slab_passes(bx[cur[live]], o[live], d[live])
                assert ok.all(), (name, "ancestor", int((~ok).sum()))
                ancestors_checked += int(live.sum())
        # How much of a triangle's margin do accepted rays use?  Exact (binary64) Chebyshev distance from the ray (t >= 0) to
        # the triangle's own bounding box -- a convex piecewise-linear function of t, minimised by ternary search -- over the
        # margin the analysis grants that triangle: 32 (u/eps) P (omax + |v0| + E1 + E2) + 6u (E1 + E2).
        o64, d64 = o.astype(np.float64), d.astype(np.float64)
        v = np.stack([tris["v0"][k], tris["v1"][k], tris["v2"][k]], 1).astype(np.float64)
        tri_lo, tri_hi = v.min(1), v.max(1)
        def fdist(t):
            pts = o64 + t[:, None] * d64
            return np.maximum(np.maximum(tri_lo - pts, pts - tri_hi), 0).max(1)
        lo_t = np.zeros(len(k)); hi_t = np.full(len(k), 4 * omax / 0.5)
        for _ in range(200):
            m1 = lo_t + (hi_t - lo_t) / 3; m2 = hi_t - (hi_t - lo_t) / 3
            left = fdist(m1) <= fdist(m2)
            hi_t = np.where(left, m2, hi_t); lo_t = np.where(left, lo_t, m1)
        need = fdist((lo_t + hi_t) / 2)
        e1 = (tris["v1"][k] - tris["v0"][k]).astype(np.float64); e2 = (tris["v2"][k] - tris["v0"][k]).astype(np.float64)
        E1, E2, V0 = np.linalg.norm(e1, axis=1), np.linalg.norm(e2, axis=1), np.linalg.norm(v[:, 0], axis=1)
        uu = 2.0 ** -24
        grant = 32 * (uu / float(f32(0.0001))) * (E1 * E2 * 1.25) * (omax + V0 + E1 + E2) + 6 * uu * (E1 + E2)
        worst = max(worst, float((need / grant).max()))
    assert total_acc > 100_000, total_acc
    assert ancestors_checked > 5 * total_acc or len(nodes) < 64, ancestors_checked
    assert worst < 0.5, worst          # found cases use ~1 % of the margin; anything near the whole of it would question the analysis
    print(f"{scene}: {total_acc} accepted (ray, triangle) pairs, none culled; largest (distance from ray to triangle box) / (margin granted) = {worst:.4f}")
