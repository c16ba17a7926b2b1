"""ctypes binding of libsquigly_hip.so (include/squigly_hip.h, include/squigly_host.h).

There is no Python or CPU fallback for the render path: if the shared library is missing this
module raises, and if no HIP device is usable the render entry points return an error.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SQ_LIB_PATH: development aid, loads an experimental build of the same library (build.py --out=...) for A/B timing
LIB_PATH = os.environ.get("SQ_LIB_PATH") or os.path.join(_HERE, "libsquigly_hip.so")

# numpy mirrors of the C structs
NODE_DTYPE = np.dtype([("kind", "<i4"), ("lmax", "<f4"), ("rmin", "<f4"), ("link", "<i4")])
TRI_DTYPE = np.dtype([("v0", "<f4", 3), ("v1", "<f4", 3), ("v2", "<f4", 3), ("mat", "<i4")])
MAT_DTYPE = np.dtype([("reflective", "<f4"), ("surf", "<f4", 3), ("emissive", "<f4"), ("emit", "<f4", 3)])
assert NODE_DTYPE.itemsize == 16 and TRI_DTYPE.itemsize == 40 and MAT_DTYPE.itemsize == 32


class Bounds(C.Structure):
    _fields_ = [("lo", C.c_float * 3), ("hi", C.c_float * 3)]


class Camera(C.Structure):
    """Geometry.Camera: position + rotation matrix (row-major)."""
    _fields_ = [("pos", C.c_float * 3), ("rot", C.c_float * 9)]


class Scene(C.Structure):
    _fields_ = [("root", Bounds), ("nodes", C.c_void_p), ("n_nodes", C.c_int32), ("tris", C.c_void_p),
                ("n_tris", C.c_int32), ("mats", C.c_void_p), ("n_mats", C.c_int32), ("height", C.c_int32)]


class Shard(C.Structure):
    _fields_ = [("row_block", C.c_int32), ("shard", C.c_int32), ("n_shards", C.c_int32)]


class SquiglyError(RuntimeError):
    """Raised when a C-ABI call returns non-zero; carries sq_last_error()."""


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python squigly-trace_amd/build.py` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # PyTorch (the plumbing for device buffers and streams) bundles its own HIP runtime.  It has to be the
    # first HIP runtime loaded into the process: if libsquigly_hip.so pulls in the system one first, torch's
    # later initialisation reports "No HIP GPUs are available".
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, sz = C.c_void_p, C.c_int32, C.c_size_t
    L.sq_last_error.restype = C.c_char_p
    L.sq_device_count.restype = i32
    L.sq_abi_version.restype = i32
    L.sq_build_id.restype = C.c_char_p
    L.sq_render_rgb8.argtypes = [C.POINTER(Scene), C.POINTER(Camera), i32, i32, i32, i32, vp]
    L.sq_render_f32.argtypes = [C.POINTER(Scene), C.POINTER(Camera), i32, i32, i32, i32, vp]
    L.sq_scene_upload.argtypes = [C.POINTER(Scene), i32, C.POINTER(vp)]
    L.sq_scene_free.argtypes = [vp]
    L.sq_scene_free.restype = None
    L.sq_shard_rows.argtypes = [i32, Shard]
    L.sq_shard_rows.restype = i32
    L.sq_shard_global_row.argtypes = [i32, Shard]
    L.sq_shard_global_row.restype = i32
    L.sq_render_rows_device.argtypes = [vp, C.POINTER(Camera), i32, i32, i32, i32, Shard, vp, vp, vp]
    L.sq_kernel_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_char_p)]
    L.sq_kernel_timing_reset.argtypes = [vp]
    L.sq_kernel_timing_reset.restype = None
    L.sq_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.sq_get_stats.argtypes = [vp, C.POINTER(C.c_uint64), i32, i32]
    L.sq_debug_eval.argtypes = [i32, i32, vp, vp, C.c_int64, vp]
    # host side
    L.sq_mesh_from_obj.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(vp)]
    L.sq_mesh_from_text.argtypes = [C.c_char_p, sz, C.c_char_p, sz, C.POINTER(vp)]
    L.sq_mesh_from_arrays.argtypes = [vp, i32, vp, i32, C.POINTER(vp)]
    L.sq_mesh_num_tris.argtypes = [vp]
    L.sq_mesh_num_materials.argtypes = [vp]
    L.sq_mesh_tris.argtypes = [vp]
    L.sq_mesh_tris.restype = vp
    L.sq_mesh_materials.argtypes = [vp]
    L.sq_mesh_materials.restype = vp
    L.sq_mesh_free.argtypes = [vp]
    L.sq_mesh_free.restype = None
    L.sq_camera_from_file.argtypes = [C.c_char_p, C.POINTER(Camera)]
    L.sq_camera_from_text.argtypes = [C.c_char_p, sz, C.POINTER(Camera)]
    L.sq_rot_matrix_rads.argtypes = [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)]
    L.sq_rot_matrix_rads.restype = None
    L.sq_mesh_debug_show.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]
    L.sq_mesh_debug_show.restype = None
    L.sq_release_cached_memory.argtypes = []
    L.sq_release_cached_memory.restype = None
    L.sq_bih_build.argtypes = [vp, C.POINTER(vp)]
    L.sq_bih_build_device.argtypes = [vp, C.c_int32, C.POINTER(vp)]
    L.sq_cull_boxes.argtypes = [C.POINTER(Scene), vp, vp]
    L.sq_half_outward.argtypes = [C.c_float, i32]
    L.sq_half_outward.restype = C.c_uint32
    L.sq_bih_scene.argtypes = [vp, C.POINTER(Scene)]
    L.sq_bih_scene.restype = None
    for f in ("height", "num_leaves", "longest_leaf"):
        getattr(L, "sq_bih_" + f).argtypes = [vp]
        getattr(L, "sq_bih_" + f).restype = i32
    L.sq_bih_free.argtypes = [vp]
    L.sq_bih_free.restype = None
    _lib = L
    return L


OPS = {"sqrt": 0, "div": 1, "sin": 2, "cos": 3, "acos": 4, "atan": 5, "unit_float": 6, "tfgen3": 7, "tonemap": 8, "rcp_sweep": 9, "cull_slab": 10}


def debug_eval(op, a, b=None, device=0):
    """sq_debug_eval: one primitive of the numeric spec evaluated on the GPU (diagnostics)."""
    code = OPS[op]
    if op == "tfgen3":
        a = np.ascontiguousarray(a, np.int64); n = a.size; out = np.empty((n, 3), np.uint32)
    elif op == "tonemap":
        a = np.ascontiguousarray(a, np.float32).reshape(-1, 3); n = len(a); out = np.empty((n, 3), np.uint8)
    elif op == "unit_float":
        a = np.ascontiguousarray(a, np.uint32); n = a.size; out = np.empty(n, np.float32)
    elif op == "cull_slab":
        a = np.ascontiguousarray(a, np.uint32).reshape(-1, 9); n = len(a); out = np.empty(n, np.uint32)
    elif op == "rcp_sweep":
        a = np.ascontiguousarray(a, np.uint32); n = a.size; out = np.empty(n, np.uint32)
    else:
        a = np.ascontiguousarray(a, np.float32); n = a.size; out = np.empty(n, np.float32)
    if b is not None:
        b = np.ascontiguousarray(b, np.float32)
    check(lib().sq_debug_eval(device, code, a.ctypes.data, b.ctypes.data if b is not None else None, n, out.ctypes.data))
    return out


def build_id():
    """sq_build_id(): hash of the sources and flags the loaded library was built from (build.py: source_id)."""
    return lib().sq_build_id().decode()


def check(rc):
    if rc != 0:
        raise SquiglyError(lib().sq_last_error().decode(errors="replace"))


EXPORTED_SYMBOLS = [
    # include/squigly_hip.h
    "sq_render_rgb8", "sq_render_f32", "sq_scene_upload", "sq_scene_free", "sq_shard_rows",
    "sq_shard_global_row", "sq_render_rows_device", "sq_kernel_timing", "sq_kernel_timing_reset",
    "sq_set_option", "sq_get_stats", "sq_debug_eval", "sq_device_count", "sq_abi_version", "sq_build_id", "sq_last_error",
    # include/squigly_host.h
    "sq_mesh_from_obj", "sq_mesh_from_text", "sq_mesh_from_arrays", "sq_mesh_num_tris",
    "sq_mesh_num_materials", "sq_mesh_tris", "sq_mesh_materials", "sq_mesh_free", "sq_camera_from_file",
    "sq_camera_from_text", "sq_rot_matrix_rads", "sq_release_cached_memory", "sq_mesh_debug_show", "sq_bih_build", "sq_bih_build_device", "sq_cull_boxes", "sq_half_outward", "sq_bih_scene", "sq_bih_height",
    "sq_bih_num_leaves", "sq_bih_longest_leaf", "sq_bih_free",
]
