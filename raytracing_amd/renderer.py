"""ctypes binding of raytracing_amd/librtx.so (the C ABI of include/rtx.h).

There is no fallback: if the library is missing or a call fails, an exception is
raised.  Radiance is only ever computed by the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import _abi
from .scene import Scene

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTX_LIB selects another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("RTX_LIB") or os.path.join(_HERE, "librtx.so")

# every entry point include/rtx.h declares: (name, restype, argtypes)
_FP = C.POINTER(C.c_float)
SYMBOLS = {
    "rtx_abi_version": (C.c_uint32, []),
    "rtx_last_error": (C.c_char_p, []),
    "rtx_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(_abi.rtx_opts)]),
    "rtx_destroy": (None, [C.c_void_p]),
    "rtx_set_scene": (C.c_int, [C.c_void_p, C.POINTER(_abi.rtx_scene)]),
    "rtx_render": (C.c_int, [C.c_void_p, C.POINTER(_abi.rtx_scene), _FP]),
    "rtx_render_device": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "rtx_render_samples": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "rtx_render_samples_static": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "rtx_accumulate_samples": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "rtx_set_share": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "rtx_tonemap_bytes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_void_p, C.c_void_p]),
    "rtx_render_bytes": (C.c_int, [C.c_void_p, C.POINTER(_abi.rtx_scene), C.POINTER(C.c_uint8)]),
    "rtx_progress_begin": (C.c_int, [C.c_void_p, C.POINTER(_abi.rtx_scene), _FP, C.c_uint32]),
    "rtx_progress_step": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rtx_progress_read": (C.c_int, [C.c_void_p, _FP, _FP, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)]),
    "rtx_divide": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_void_p]),
    "rtx_primary": (C.c_int, [C.c_void_p, C.POINTER(_abi.rtx_scene), C.POINTER(C.c_int32), _FP, _FP,
                              C.POINTER(C.c_uint8)]),
    "rtx_get_stats": (C.c_int, [C.c_void_p, C.POINTER(_abi.rtx_stats)]),
    "rtx_last_kernel_ms": (C.c_int, [C.c_void_p, _FP]),
    "rtx_get_lights": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int32]),
    "rtx_device_count": (C.c_int, [C.c_void_p]),
    "rtx_alloc_pinned": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint64]),
    "rtx_free_pinned": (None, [C.c_void_p]),
}

_lib = None


class RtxError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"librtx error {code}: {message}")
        self.code = code


def load_library():
    """Load librtx.so and declare its prototypes.  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: build it with `make -C raytracing_amd/csrc` "
                "(there is no CPU fallback)")
        # librtx.so and PyTorch-ROCm both need libamdhip64.so.7.  A process must hold ONE
        # HIP runtime: if torch is going to be used (device buffers, torch.distributed) it
        # has to be loaded first so that librtx.so binds to the runtime torch brought in.
        if os.environ.get("RTX_NO_TORCH", "0") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        lib = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.rtx_abi_version() != _abi.RTX_ABI_VERSION:
            raise RuntimeError("librtx.so ABI version does not match the Python binding")
        _lib = lib
    return _lib


def _check(rc: int):
    if rc < 0:
        raise RtxError(rc, load_library().rtx_last_error().decode("utf-8", "replace"))
    return rc


class PinnedImage:
    """A float32 (height, width, 3) array in page-locked host memory (rtx_alloc_pinned): what
    Renderer.render(..., out=) writes into without the runtime staging the copy."""

    def __init__(self, height: int, width: int):
        self._lib = load_library()
        self._ptr = C.c_void_p()
        n = height * width * 3
        _check(self._lib.rtx_alloc_pinned(C.byref(self._ptr), n * 4))
        self.array = np.ctypeslib.as_array(C.cast(self._ptr, _FP), shape=(n,)).reshape(height, width, 3)

    def close(self):
        if getattr(self, "_ptr", None):
            self.array = None
            self._lib.rtx_free_pinned(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Renderer:
    """One rtx_ctx: drives one GPU -- or, with `devices`, several GPUs of the node -- from one
    host thread."""

    def __init__(self, seed: int = 1, mode: int = _abi.MODE_MEGAKERNEL, accel: int = _abi.ACCEL_AUTO,
                 rng_mode: int = _abi.RNG_PHILOX, const_words=(0, 0, 0, 0), device: int = -1,
                 collect_stats: bool = False, devices=None, reduce: int = _abi.REDUCE_P2P):
        self._lib = load_library()
        o = _abi.rtx_opts()
        o.seed, o.mode, o.accel, o.rng_mode = seed, mode, accel, rng_mode
        o.const_words[:] = const_words
        o.device, o.collect_stats = device, int(collect_stats)
        o.reduce = reduce
        if devices is not None:
            # HIP ordinals the frame is dealt out to (an ordinal may repeat)
            self._devices = (C.c_int32 * len(devices))(*devices)
            o.n_devices, o.devices = len(devices), self._devices
        self._ctx = C.c_void_p()
        _check(self._lib.rtx_create(C.byref(self._ctx), C.byref(o)))
        self._scene: Optional[Scene] = None

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.rtx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- scene -----------------------------------------------------------
    def set_scene(self, scene: Scene):
        s, keep = scene.to_ctypes()
        _check(self._lib.rtx_set_scene(self._ctx, C.byref(s)))
        self._scene = scene

    def device_count(self) -> int:
        return _check(self._lib.rtx_device_count(self._ctx))

    def lights(self):
        n = _check(self._lib.rtx_get_lights(self._ctx, None, 0))
        buf = (C.c_int32 * max(1, n))()
        _check(self._lib.rtx_get_lights(self._ctx, buf, n))
        return list(buf[:n])

    # -- drop-in for render(&mut scene) -----------------------------------
    def render(self, scene: Scene, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Mean linear radiance, shape (height, width, 3), row 0 = top.  out: a C-contiguous float32
        array of that shape to write into (e.g. PinnedImage.array)."""
        s, keep = scene.to_ctypes()
        if out is None:
            out = np.empty((scene.height, scene.width, 3), dtype=np.float32)
        elif out.dtype != np.float32 or out.shape != (scene.height, scene.width, 3) or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float32 array of shape (height, width, 3)")
        _check(self._lib.rtx_render(self._ctx, C.byref(s), out.ctypes.data_as(_FP)))
        self._scene = scene
        return out

    def render_device(self, n_samples: int, d_rgb_ptr: int):
        """render() of the scene given to set_scene with the mean image left in a device buffer on
        the context's first GPU (raw pointer); all of the context's GPUs take part."""
        _check(self._lib.rtx_render_device(self._ctx, n_samples, C.c_void_p(d_rgb_ptr)))

    # -- device-resident pieces (multi-GPU, benchmarks) -------------------
    def render_samples(self, sample_first: int, sample_count: int, d_sum_ptr: int, stream: int = 0):
        """Sum of samples [first, first+count) into a device buffer (raw pointer)."""
        _check(self._lib.rtx_render_samples(self._ctx, sample_first, sample_count,
                                            C.c_void_p(d_sum_ptr), C.c_void_p(stream)))

    def set_share(self, share_index: int, share_count: int):
        """One process per GPU: from now on render_samples covers share `share_index` of `share_count` of the
        frame's work units and writes zeros elsewhere; the shares' buffers added up are the one-GPU frame, bit
        for bit."""
        _check(self._lib.rtx_set_share(self._ctx, share_index, share_count))

    def render_samples_static(self, sample_first: int, sample_count: int, d_sum_ptr: int, stream: int = 0):
        """render_samples without any device allocation (static tile tables)."""
        _check(self._lib.rtx_render_samples_static(self._ctx, sample_first, sample_count,
                                                   C.c_void_p(d_sum_ptr), C.c_void_p(stream)))

    def accumulate_samples(self, sample_first: int, sample_count: int, d_sum_ptr: int, stream: int = 0):
        """As render_samples, continuing the sums already in the device buffer."""
        _check(self._lib.rtx_accumulate_samples(self._ctx, sample_first, sample_count,
                                                C.c_void_p(d_sum_ptr), C.c_void_p(stream)))

    def tonemap_bytes(self, d_sum_ptr: int, n_floats: int, divisor: float, d_bytes_ptr: int, stream: int = 0):
        """ACES + gamma + byte conversion of sum / divisor, device buffer to device buffer."""
        _check(self._lib.rtx_tonemap_bytes(self._ctx, C.c_void_p(d_sum_ptr), n_floats, divisor,
                                           C.c_void_p(d_bytes_ptr), C.c_void_p(stream)))

    def render_bytes(self, scene: Scene) -> np.ndarray:
        """render + color_correction + byte conversion on the device: uint8 (height, width, 3)."""
        s, keep = scene.to_ctypes()
        out = np.empty((scene.height, scene.width, 3), dtype=np.uint8)
        _check(self._lib.rtx_render_bytes(self._ctx, C.byref(s), out.ctypes.data_as(C.POINTER(C.c_uint8))))
        self._scene = scene
        return out

    # -- progressive rendering (stop / checkpoint / resume) ---------------
    def progress_begin(self, scene: Scene, sum: Optional[np.ndarray] = None, samples_done: int = 0):
        s, keep = scene.to_ctypes()
        ptr = None
        if sum is not None:
            sum = np.ascontiguousarray(sum, dtype=np.float32)
            if sum.size != scene.width * scene.height * 3:
                raise ValueError("sum does not have width*height*3 values")
            ptr = sum.ctypes.data_as(_FP)
        _check(self._lib.rtx_progress_begin(self._ctx, C.byref(s), ptr, samples_done))
        self._scene = scene

    def progress_step(self, sample_count: int):
        _check(self._lib.rtx_progress_step(self._ctx, sample_count))

    def progress_read(self, what: str = "mean"):
        """what = 'mean' | 'sum' (float32 (h, w, 3)) | 'bytes' (uint8); returns (array, samples_done)."""
        sc = self._scene
        done = C.c_uint32()
        shape = (sc.height, sc.width, 3)
        if what == "bytes":
            out = np.empty(shape, dtype=np.uint8)
            _check(self._lib.rtx_progress_read(self._ctx, None, None, out.ctypes.data_as(C.POINTER(C.c_uint8)),
                                               C.byref(done)))
        else:
            out = np.empty(shape, dtype=np.float32)
            args = (out.ctypes.data_as(_FP), None) if what == "mean" else (None, out.ctypes.data_as(_FP))
            _check(self._lib.rtx_progress_read(self._ctx, args[0], args[1], None, C.byref(done)))
        return out, done.value

    def progress_read_all(self):
        """mean, sum and bytes from ONE rtx_progress_read call: (mean, sum, bytes, samples_done)."""
        sc = self._scene
        done = C.c_uint32()
        shape = (sc.height, sc.width, 3)
        mean = np.empty(shape, dtype=np.float32)
        total = np.empty(shape, dtype=np.float32)
        data = np.empty(shape, dtype=np.uint8)
        _check(self._lib.rtx_progress_read(self._ctx, mean.ctypes.data_as(_FP), total.ctypes.data_as(_FP),
                                           data.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(done)))
        return mean, total, data, done.value

    def divide(self, d_ptr: int, n_floats: int, divisor: float, stream: int = 0):
        _check(self._lib.rtx_divide(self._ctx, C.c_void_p(d_ptr), n_floats, divisor, C.c_void_p(stream)))

    def primary(self, scene: Scene):
        """Deterministic primary-visibility pass: (hit_id, t, normal, inside) arrays."""
        s, keep = scene.to_ctypes()
        n = scene.width * scene.height
        ids = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float32)
        nrm = np.empty((n, 3), dtype=np.float32)
        ins = np.empty(n, dtype=np.uint8)
        _check(self._lib.rtx_primary(self._ctx, C.byref(s), ids.ctypes.data_as(C.POINTER(C.c_int32)),
                                     t.ctypes.data_as(_FP), nrm.ctypes.data_as(_FP),
                                     ins.ctypes.data_as(C.POINTER(C.c_uint8))))
        self._scene = scene
        shp = (scene.height, scene.width)
        return ids.reshape(shp), t.reshape(shp), nrm.reshape(shp + (3,)), ins.reshape(shp)

    def stats(self) -> dict:
        st = _abi.rtx_stats()
        _check(self._lib.rtx_get_stats(self._ctx, C.byref(st)))
        return st.as_dict()

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        _check(self._lib.rtx_last_kernel_ms(self._ctx, C.byref(ms)))
        return ms.value
