"""squigly-trace on MI355X: the per-pixel sampling loop of rrruko/squigly-trace as HIP kernels.

Import with importlib (the directory name contains a hyphen):
    sqt = importlib.import_module("squigly-trace_amd")
"""
from ._native import Camera, Shard, SquiglyError, lib, LIB_PATH, EXPORTED_SYMBOLS, debug_eval, build_id  # noqa: F401
from .scene import BIH, Mesh, load_camera, camera_from_text, rot_matrix_rads       # noqa: F401
from .render import Settings, render, render_rgb8, render_f32                       # noqa: F401
from .png import write_png                                                          # noqa: F401


def device_count() -> int:
    return lib().sq_device_count()


def release_cached_memory() -> None:
    """Hand the frame workspaces kept for the next scene back to the driver (sq_release_cached_memory)."""
    lib().sq_release_cached_memory()


def __getattr__(name):
    # torch is only needed for the resident-scene path
    if name in ("DeviceScene",):
        from .device import DeviceScene
        return DeviceScene
    if name in ("dist",):
        import importlib
        return importlib.import_module(__name__ + ".dist")
    raise AttributeError(name)
