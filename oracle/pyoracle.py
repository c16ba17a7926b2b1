"""ctypes binding of oracle/libsq_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Nothing under squigly-trace_amd/ imports it (tests/test_host.py::test_product_does_not_touch_the_oracle enforces that).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libsq_oracle.so")

TRIG_CRD, TRIG_LIBM = 0, 1


class V3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Material(C.Structure):
    _fields_ = [("reflective", C.c_float), ("surf", V3), ("emissive", C.c_float), ("emit", V3)]


class Triangle(C.Structure):
    _fields_ = [("a", V3), ("b", V3), ("c", V3), ("mat", Material)]


class Camera(C.Structure):
    _fields_ = [("pos", V3), ("rot", C.c_float * 9)]


class Bounds(C.Structure):
    _fields_ = [("lo", V3), ("hi", V3)]


class Hit(C.Structure):
    _fields_ = [("point", V3), ("dist", C.c_float), ("tri", C.c_int), ("hit", C.c_int)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("samples", "rays", "branch_visits", "slab_tests", "leaf_visits", "tri_tests", "hits",
                 "b_rays", "b_branch_visits", "b_tri_tests", "b_hits")]

    def asdict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


TRI_DTYPE = np.dtype([("a", "<f4", 3), ("b", "<f4", 3), ("c", "<f4", 3), ("reflective", "<f4"),
                      ("surf", "<f4", 3), ("emissive", "<f4"), ("emit", "<f4", 3)])
assert TRI_DTYPE.itemsize == C.sizeof(Triangle) == 68


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile) if the library is missing (or force=True).
    No mtime check: on the GPU box the library arrives prebuilt and spawning `make` from a process that has
    already initialised the GPU is not allowed there.  __graft_entry__.build() always runs make."""
    if force or not os.path.exists(_LIB):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB)
    L.sqo_last_error.restype = C.c_char_p
    L.sqo_tris_from_text.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                     C.POINTER(C.POINTER(Triangle)), C.POINTER(C.c_int)]
    L.sqo_tris_from_obj.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.POINTER(Triangle)), C.POINTER(C.c_int)]
    L.sqo_mtllib_of_text.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    L.sqo_camera_from_text.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(Camera)]
    L.sqo_load_camera.argtypes = [C.c_char_p, C.c_int, C.POINTER(Camera)]
    L.sqo_free.argtypes = [C.c_void_p]
    L.sqo_rot_matrix_rads.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int, C.POINTER(C.c_float)]
    L.sqo_rot_vert.argtypes = [V3, C.POINTER(C.c_float)]
    L.sqo_rot_vert.restype = V3
    L.sqo_intersects_bb.argtypes = [C.POINTER(Bounds), V3, V3]
    L.sqo_moller_trumbore.argtypes = [V3, V3, C.POINTER(Triangle), C.POINTER(V3), C.POINTER(C.c_float)]
    for f in ("sinf", "cosf", "acosf", "atanf"):
        fn = getattr(L, "sqo_" + f)
        fn.argtypes = [C.c_float, C.c_int]
        fn.restype = C.c_float
    for f in ("sin_d", "cos_d", "acos_d", "atan_d"):
        fn = getattr(L, "sqo_" + f)
        fn.argtypes = [C.c_double]
        fn.restype = C.c_double
    L.sqo_threefish256.argtypes = [C.POINTER(C.c_uint64)] * 3 + [C.c_int, C.POINTER(C.c_uint64)]
    L.sqo_tfgen_words.argtypes = [C.c_int64, C.c_int, C.POINTER(C.c_uint32)]
    L.sqo_tonemap.argtypes = [V3, C.c_int, C.POINTER(C.c_uint8)]
    L.sqo_make_ray.argtypes = [C.c_int] * 4 + [C.POINTER(Camera), C.POINTER(V3), C.POINTER(V3)]
    L.sqo_random_vector.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
    L.sqo_random_vector.restype = V3
    L.sqo_make_bih.argtypes = [C.POINTER(Triangle), C.c_int]
    L.sqo_make_bih.restype = C.c_void_p
    L.sqo_free_bih.argtypes = [C.c_void_p]
    for f in ("height", "num_leaves", "longest_leaf", "num_nodes", "num_tris"):
        getattr(L, "sqo_bih_" + f).argtypes = [C.c_void_p]
    L.sqo_bih_bounds.argtypes = [C.c_void_p, C.POINTER(Bounds)]
    L.sqo_bih_flatten.argtypes = [C.c_void_p, C.POINTER(Triangle)]
    L.sqo_bih_preorder.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float),
                                   C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    L.sqo_intersect_bih.argtypes = [C.c_void_p, V3, V3, C.POINTER(Hit), C.POINTER(Counters)]
    L.sqo_intersect_naive.argtypes = [C.c_void_p, V3, V3, C.POINTER(Hit)]
    L.sqo_render.argtypes = [C.c_void_p, C.POINTER(Camera)] + [C.c_int] * 7 + [
        C.c_void_p, C.c_void_p, C.POINTER(Counters)]
    L.sqo_render_rows.argtypes = [C.c_void_p, C.POINTER(Camera)] + [C.c_int] * 9 + [
        C.c_void_p, C.c_void_p, C.POINTER(Counters)]
    L.sqo_render_rows_strided.argtypes = [C.c_void_p, C.POINTER(Camera)] + [C.c_int] * 10 + [
        C.c_void_p, C.c_void_p, C.POINTER(Counters)]
    L.sqo_sample_radiance.argtypes = [C.c_void_p, C.POINTER(Camera)] + [C.c_int] * 8 + [C.POINTER(C.c_float)]
    _lib = L
    return L


class OracleError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise OracleError(lib().sqo_last_error().decode())


def v3(t):
    return V3(float(t[0]), float(t[1]), float(t[2]))


def tris_from_text(obj_text: bytes, sq_text: bytes) -> np.ndarray:
    """Obj.trisFromObj on in-memory texts -> structured array (TRI_DTYPE) in loader order."""
    p = C.POINTER(Triangle)()
    n = C.c_int()
    _check(lib().sqo_tris_from_text(obj_text, len(obj_text), sq_text, len(sq_text), C.byref(p), C.byref(n)))
    arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value * 68,)).copy() if n.value else \
        np.zeros(0, np.uint8)
    lib().sqo_free(p)
    return arr.view(TRI_DTYPE)


def tris_from_obj(obj_path: str, mtl_dir: str) -> np.ndarray:
    p = C.POINTER(Triangle)()
    n = C.c_int()
    _check(lib().sqo_tris_from_obj(obj_path.encode(), mtl_dir.encode(), C.byref(p), C.byref(n)))
    arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value * 68,)).copy() if n.value else \
        np.zeros(0, np.uint8)
    lib().sqo_free(p)
    return arr.view(TRI_DTYPE)


def load_camera(path: str, trig=TRIG_CRD) -> Camera:
    cam = Camera()
    _check(lib().sqo_load_camera(path.encode(), trig, C.byref(cam)))
    return cam


def camera_from_text(text: bytes, trig=TRIG_CRD) -> Camera:
    cam = Camera()
    _check(lib().sqo_camera_from_text(text, len(text), trig, C.byref(cam)))
    return cam


def camera_arrays(cam: Camera):
    return (np.array([cam.pos.x, cam.pos.y, cam.pos.z], np.float32), np.array(list(cam.rot), np.float32))


class BIH:
    """BIH.makeBIH result (recursive tree inside the C library)."""

    def __init__(self, tris: np.ndarray):
        tris = np.ascontiguousarray(tris)
        assert tris.dtype == TRI_DTYPE
        self._h = lib().sqo_make_bih(tris.ctypes.data_as(C.POINTER(Triangle)), len(tris))
        self.n_tris = lib().sqo_bih_num_tris(self._h)
        self.n_nodes = lib().sqo_bih_num_nodes(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().sqo_free_bih(self._h)
            self._h = None

    height = property(lambda s: lib().sqo_bih_height(s._h))
    num_leaves = property(lambda s: lib().sqo_bih_num_leaves(s._h))
    longest_leaf = property(lambda s: lib().sqo_bih_longest_leaf(s._h))

    def bounds(self):
        b = Bounds()
        lib().sqo_bih_bounds(self._h, C.byref(b))
        return np.array([b.lo.x, b.lo.y, b.lo.z, b.hi.x, b.hi.y, b.hi.z], np.float32)

    def flatten(self) -> np.ndarray:
        out = np.zeros(self.n_tris, TRI_DTYPE)
        lib().sqo_bih_flatten(self._h, out.ctypes.data_as(C.POINTER(Triangle)))
        return out

    def preorder(self):
        n = self.n_nodes
        kind = np.zeros(n, np.int32)
        a = np.zeros(n, np.float32)
        b = np.zeros(n, np.float32)
        cnt = np.zeros(n, np.int32)
        lib().sqo_bih_preorder(self._h, kind.ctypes.data_as(C.POINTER(C.c_int32)),
                               a.ctypes.data_as(C.POINTER(C.c_float)), b.ctypes.data_as(C.POINTER(C.c_float)),
                               cnt.ctypes.data_as(C.POINTER(C.c_int32)))
        return kind, a, b, cnt

    def intersect(self, o, d, counters=None):
        h = Hit()
        lib().sqo_intersect_bih(self._h, v3(o), v3(d), C.byref(h), C.byref(counters) if counters else None)
        return h

    def intersect_naive(self, o, d):
        h = Hit()
        lib().sqo_intersect_naive(self._h, v3(o), v3(d), C.byref(h))
        return h

    def render(self, cam: Camera, samples, w, h, cast=False, threads=1, trig=TRIG_CRD, rng_variant=0,
               rows=None, row_step=1, want_avg=True, want_rgb=True):
        """Lib.render minus the PNG write. Returns (avg[w,h,3] f32, rgb[w,h,3] u8, counters dict).
        rows=(y0,y1) with row_step renders rows y0, y0+row_step, ... < y1 into a compact array."""
        y0, y1 = rows if rows is not None else (0, w)
        nrows = len(range(y0, y1, row_step))
        avg = np.zeros((nrows, h, 3), np.float32) if want_avg else None
        rgb = np.zeros((nrows, h, 3), np.uint8) if want_rgb else None
        c = Counters()
        _check(lib().sqo_render_rows_strided(self._h, C.byref(cam), samples, w, h, int(cast), y0, y1, row_step, threads, trig,
                                     rng_variant, avg.ctypes.data if want_avg else None,
                                     rgb.ctypes.data if want_rgb else None, C.byref(c)))
        return avg, rgb, c.asdict()

    def sample_radiance(self, cam, samples, w, h, y, x, k, trig=TRIG_CRD, rng_variant=0):
        out = (C.c_float * 3)()
        lib().sqo_sample_radiance(self._h, C.byref(cam), samples, w, h, y, x, k, trig, rng_variant, out)
        return np.array(list(out), np.float32)


def threefish256(key, tweak, pt, variant=0):
    k = (C.c_uint64 * 4)(*key)
    t = (C.c_uint64 * 2)(*tweak)
    p = (C.c_uint64 * 4)(*pt)
    o = (C.c_uint64 * 4)()
    lib().sqo_threefish256(k, t, p, variant, o)
    return [int(v) for v in o]


def tfgen_words(seed, variant=0):
    o = (C.c_uint32 * 8)()
    lib().sqo_tfgen_words(seed, variant, o)
    return [int(v) for v in o]


def tonemap(c, trig=TRIG_CRD):
    o = (C.c_uint8 * 3)()
    lib().sqo_tonemap(v3(c), trig, o)
    return tuple(int(v) for v in o)


def make_ray(w, h, y, x, cam):
    o, d = V3(), V3()
    lib().sqo_make_ray(w, h, y, x, C.byref(cam), C.byref(o), C.byref(d))
    return (np.array([o.x, o.y, o.z], np.float32), np.array([d.x, d.y, d.z], np.float32))
