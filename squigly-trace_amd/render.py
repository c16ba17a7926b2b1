"""Mirror of the reference's `Lib` module (src/Lib.hs): Settings and render.

    render :: Scene a -> Camera -> Settings -> IO ()          src/Lib.hs:68-75
Here `render(bih, cam, settings)` fills the same w-rows x h-columns RGB8 image through
sq_render_rgb8 — the foreign call that replaces src/Lib.hs:73-74 — and writes the PNG.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import Tuple

import numpy as np

from . import _native as N
from .png import write_png


@dataclass
class Settings:
    """src/Lib.hs:54-63, defaults from app/Main.hs:13-30."""
    samples: int = 10
    dimensions: Tuple[int, int] = (540, 540)
    savePath: str = "./render/result.png"
    objPath: str = "./data/scene.obj"
    cameraPath: str = "./data/camera"
    debug: bool = False
    debugPath: str = ""
    cast: bool = False


def render_rgb8(bih, cam, samples, dimensions, cast=False) -> np.ndarray:
    """The array `img` of src/Lib.hs:74: shape (w, h, 3) uint8 — w ROWS, h COLUMNS."""
    w, h = dimensions
    out = np.empty((max(w, 0), max(h, 0), 3), np.uint8)
    N.check(N.lib().sq_render_rgb8(C.byref(bih.scene), C.byref(cam), samples, w, h, int(bool(cast)), out.ctypes.data))
    return out


def render_f32(bih, cam, samples, dimensions, cast=False) -> np.ndarray:
    """The pre-tonemap `avg` of src/Lib.hs:88 for every pixel: shape (w, h, 3) float32."""
    w, h = dimensions
    out = np.empty((max(w, 0), max(h, 0), 3), np.float32)
    N.check(N.lib().sq_render_f32(C.byref(bih.scene), C.byref(cam), samples, w, h, int(bool(cast)), out.ctypes.data))
    return out


def render(bih, cam, settings: Settings):
    """Lib.render: compute the image and write it to settings.savePath."""
    img = render_rgb8(bih, cam, settings.samples, settings.dimensions, settings.cast)
    write_png(settings.savePath, img)
    return img
