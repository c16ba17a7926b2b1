"""Resident-scene rendering on one MI355X: the scene is uploaded once, frames stay in HBM.

PyTorch is used only for device memory and streams (plumbing); every pixel is produced by the HIP
kernels in csrc/sq_device.hip through the C-ABI of include/squigly_hip.h.
"""
import ctypes as C

import torch

from . import _native as N


class DeviceScene:
    """A BIH uploaded to one GPU (sq_scene_upload)."""

    def __init__(self, bih, device=0):
        self.bih = bih                      # keeps the host arrays alive
        self.device = int(device)
        h = C.c_void_p()
        N.check(N.lib().sq_scene_upload(C.byref(bih.scene), self.device, C.byref(h)))
        self._h = h

    def set_option(self, key, value):
        N.check(N.lib().sq_set_option(self._h, key.encode(), int(value)))

    def render_rows(self, cam, samples, w, h, cast=False, shard=(None, 0, 1), want_avg=True, want_rgb=True,
                    stream=None, out_avg=None, out_rgb=None):
        """Enqueue the render of this shard's rows; returns (avg, rgb) CUDA tensors [rows, h, 3].

        shard = (row_block, shard_index, n_shards); row_block None = all rows in one block.
        """
        rb, si, ns = shard
        sh = N.Shard(int(w if rb is None else rb), int(si), int(ns))
        rows = N.lib().sq_shard_rows(w, sh)
        if rows < 0:
            raise N.SquiglyError(f"bad shard {shard}")
        dev = torch.device("cuda", self.device)
        if want_avg and out_avg is None:
            out_avg = torch.empty((rows, h, 3), dtype=torch.float32, device=dev)
        if want_rgb and out_rgb is None:
            out_rgb = torch.empty((rows, h, 3), dtype=torch.uint8, device=dev)
        st = stream if stream is not None else torch.cuda.current_stream(dev)
        N.check(N.lib().sq_render_rows_device(
            self._h, C.byref(cam), samples, w, h, int(bool(cast)), sh,
            out_avg.data_ptr() if out_avg is not None else None,
            out_rgb.data_ptr() if out_rgb is not None else None,
            C.c_void_p(st.cuda_stream)))
        return out_avg, out_rgb

    def enable_timing(self, on=True):
        """Bracket every launch of the dominant kernel with hipEvents (read back by kernel_timing)."""
        self.set_option("timing", int(bool(on)))

    def kernel_timing(self):
        """(average ms per launch of the dominant kernel, launches, kernel name) since the last reset."""
        ms, n, name = C.c_double(), C.c_int64(), C.c_char_p()
        N.check(N.lib().sq_kernel_timing(self._h, C.byref(ms), C.byref(n), C.byref(name)))
        return ms.value, n.value, (name.value or b"").decode()

    def stats(self, reset=False):
        """Cumulative trace-kernel statistics: [rays traced, profile counters...] (synchronises)."""
        out = (C.c_uint64 * 32)()
        N.check(N.lib().sq_get_stats(self._h, out, 32, int(reset)))
        return [int(v) for v in out]

    def reset_timing(self):
        N.lib().sq_kernel_timing_reset(self._h)

    def close(self):
        if getattr(self, "_h", None) and N is not None and N._lib is not None:
            N._lib.sq_scene_free(self._h)
        self._h = None

    __del__ = close
