"""Host-side data model, mirroring the reference's Obj / Geometry / BIH modules.

    trisFromObj / loadCamera   src/Obj.hs:49-70      -> Mesh.from_obj, load_camera
    makeBIH / flatten          src/BIH.hs:50-52,62-99 -> BIH(mesh)
    height/numLeaves/longestLeaf src/BIH.hs:46-60    -> BIH.height / .num_leaves / .longest_leaf
The arithmetic lives in csrc/sq_host.cpp; these classes only own the C objects.
"""
import ctypes as C

import numpy as np

from . import _native as N


class Mesh:
    """[Triangle] in loader order (Obj.trisFromObj)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_obj(cls, obj_path, mtl_dir="./data"):
        h = C.c_void_p()
        N.check(N.lib().sq_mesh_from_obj(str(obj_path).encode(), str(mtl_dir).encode(), C.byref(h)))
        return cls(h)

    @classmethod
    def from_text(cls, obj_text: bytes, sq_text: bytes):
        h = C.c_void_p()
        N.check(N.lib().sq_mesh_from_text(obj_text, len(obj_text), sq_text, len(sq_text), C.byref(h)))
        return cls(h)

    @classmethod
    def from_arrays(cls, tris: np.ndarray, mats: np.ndarray):
        tris = np.ascontiguousarray(tris, dtype=N.TRI_DTYPE)
        mats = np.ascontiguousarray(mats, dtype=N.MAT_DTYPE)
        h = C.c_void_p()
        N.check(N.lib().sq_mesh_from_arrays(tris.ctypes.data, len(tris), mats.ctypes.data, len(mats), C.byref(h)))
        return cls(h)

    def __len__(self):
        return N.lib().sq_mesh_num_tris(self._h)

    def debug_show(self):
        """(`show (head objs)`, `show mats`): the two lines `--debug` prints while loading (src/Obj.hs:55-57)."""
        a, b = C.c_char_p(), C.c_char_p()
        N.lib().sq_mesh_debug_show(self._h, C.byref(a), C.byref(b))
        return (a.value or b"").decode("latin-1"), (b.value or b"").decode("latin-1")

    @property
    def tris(self) -> np.ndarray:
        n = len(self)
        if n == 0:
            return np.zeros(0, N.TRI_DTYPE)
        buf = (C.c_uint8 * (n * N.TRI_DTYPE.itemsize)).from_address(N.lib().sq_mesh_tris(self._h))
        return np.frombuffer(buf, dtype=N.TRI_DTYPE).copy()

    @property
    def materials(self) -> np.ndarray:
        n = N.lib().sq_mesh_num_materials(self._h)
        if n == 0:
            return np.zeros(0, N.MAT_DTYPE)
        buf = (C.c_uint8 * (n * N.MAT_DTYPE.itemsize)).from_address(N.lib().sq_mesh_materials(self._h))
        return np.frombuffer(buf, dtype=N.MAT_DTYPE).copy()

    def __del__(self):
        if getattr(self, "_h", None) and N is not None and N._lib is not None:
            N._lib.sq_mesh_free(self._h)
        self._h = None


class BIH:
    """BIH.makeBIH result, flattened to the pre-order arrays of include/squigly_hip.h."""

    def __init__(self, mesh: Mesh, device=None):
        """device=None: host build (sq_bih_build); device=k: the same tree built on GPU k (sq_bih_build_device)."""
        h = C.c_void_p()
        if device is None:
            N.check(N.lib().sq_bih_build(mesh._h, C.byref(h)))
        else:
            N.check(N.lib().sq_bih_build_device(mesh._h, int(device), C.byref(h)))
        self._h = h
        self.scene = N.Scene()
        N.lib().sq_bih_scene(self._h, C.byref(self.scene))

    height = property(lambda s: N.lib().sq_bih_height(s._h))
    num_leaves = property(lambda s: N.lib().sq_bih_num_leaves(s._h))
    longest_leaf = property(lambda s: N.lib().sq_bih_longest_leaf(s._h))

    def _view(self, ptr, n, dt):
        if n == 0:
            return np.zeros(0, dt)
        buf = (C.c_uint8 * (n * dt.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dt).copy()

    @property
    def nodes(self):
        return self._view(self.scene.nodes, self.scene.n_nodes, N.NODE_DTYPE)

    @property
    def tris(self):
        return self._view(self.scene.tris, self.scene.n_tris, N.TRI_DTYPE)

    @property
    def materials(self):
        return self._view(self.scene.mats, self.scene.n_mats, N.MAT_DTYPE)

    @property
    def bounds(self):
        return np.array(list(self.scene.root.lo) + list(self.scene.root.hi), np.float32)

    def cull_boxes(self):
        """(boxes [n_nodes, 6], (o2max, d2min, d2max)): the culling boxes the device kernels use (sq_cull_boxes)."""
        boxes = np.empty((self.scene.n_nodes, 6), np.float32)
        lim = np.empty(3, np.float32)
        N.check(N.lib().sq_cull_boxes(C.byref(self.scene), boxes.ctypes.data, lim.ctypes.data))
        return boxes, tuple(float(x) for x in lim)

    def __del__(self):
        if getattr(self, "_h", None) and N is not None and N._lib is not None:
            N._lib.sq_bih_free(self._h)
        self._h = None


def load_camera(path) -> N.Camera:
    """Obj.loadCamera (src/Obj.hs:60-70)."""
    cam = N.Camera()
    N.check(N.lib().sq_camera_from_file(str(path).encode(), C.byref(cam)))
    return cam


def camera_from_text(text: bytes) -> N.Camera:
    cam = N.Camera()
    N.check(N.lib().sq_camera_from_text(text, len(text), C.byref(cam)))
    return cam


def rot_matrix_rads(a, b, g) -> np.ndarray:
    """Geometry.rotMatrixRads (src/Geometry.hs:90-102), row-major 3x3."""
    out = (C.c_float * 9)()
    N.lib().sq_rot_matrix_rads(a, b, g, out)
    return np.array(list(out), np.float32).reshape(3, 3)
