"""ctypes binding of raytracing_amd/librtx_host.so (include/rtx_host.h): the reference's
scene parser (src/parser.rs), tone map and PPM writer (src/image.rs) restated in C++.
Nothing here touches the GPU."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _abi
from .scene import Prim, Scene

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librtx_host.so")
_FP = C.POINTER(C.c_float)


class rtx_host_scene(C.Structure):
    _fields_ = [("scene", _abi.rtx_scene), ("prims", C.POINTER(_abi.rtx_prim)), ("fov_x", C.c_float)]


class rtx_host_checkpoint(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("samples_done", C.c_uint32),
                ("seed", C.c_uint64), ("fingerprint", C.c_uint64)]


SYMBOLS = {
    "rtx_host_parse_file": (C.c_int, [C.c_char_p, C.POINTER(rtx_host_scene)]),
    "rtx_host_parse_text": (C.c_int, [C.c_char_p, C.POINTER(rtx_host_scene)]),
    "rtx_host_free_scene": (None, [C.POINTER(rtx_host_scene)]),
    "rtx_host_color_correction": (None, [_FP, C.c_uint64]),
    "rtx_host_to_bytes": (None, [_FP, C.c_uint64, C.POINTER(C.c_uint8)]),
    "rtx_host_write_ppm": (C.c_int, [C.c_char_p, _FP, C.c_uint32, C.c_uint32]),
    "rtx_host_write_ppm_bytes": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32]),
    "rtx_host_scene_fingerprint": (C.c_uint64, [C.POINTER(_abi.rtx_scene)]),
    "rtx_host_save_checkpoint": (C.c_int, [C.c_char_p, C.POINTER(rtx_host_checkpoint), _FP]),
    "rtx_host_load_checkpoint": (C.c_int, [C.c_char_p, C.POINTER(rtx_host_checkpoint), _FP, C.c_uint64]),
    "rtx_host_last_error": (C.c_char_p, []),
}
_lib = None


class SceneParseError(ValueError):
    pass


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} is missing: build it with `make -C raytracing_amd/csrc`")
        lib = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
        _lib = lib
    return _lib


def _to_scene(hs: rtx_host_scene) -> Scene:
    s = hs.scene
    prims = []
    for i in range(s.n_prims):
        p = hs.prims[i]
        prims.append(Prim(kind=p.kind, param=tuple(p.param), position=tuple(p.position),
                          rotation=tuple(p.rotation), color=tuple(p.color), emission=tuple(p.emission),
                          material=p.material, ior=p.ior))
    ax = list(s.cam_axis)
    return Scene(width=s.width, height=s.height, ray_depth=s.ray_depth, n_samples=s.n_samples,
                 background=tuple(s.background), cam_pos=tuple(s.cam_pos), cam_right=tuple(ax[0:3]),
                 cam_up=tuple(ax[3:6]), cam_forward=tuple(ax[6:9]), fov_x=hs.fov_x, prims=prims)


def _parse(fn_name: str, arg: bytes):
    lib = load_library()
    hs = rtx_host_scene()
    rc = getattr(lib, fn_name)(arg, C.byref(hs))
    if rc != 0:
        raise SceneParseError(lib.rtx_host_last_error().decode("utf-8", "replace"))
    try:
        scene = _to_scene(hs)
        tangents = (hs.scene.tg_fov_x, hs.scene.tg_fov_y)
    finally:
        lib.rtx_host_free_scene(C.byref(hs))
    return scene, tangents


def parse_scene(path: str) -> Scene:
    """parse_scene(path) of the reference (src/parser.rs:104)."""
    return _parse("rtx_host_parse_file", os.fsencode(path))[0]


def parse_scene_text(text: str, with_tangents: bool = False):
    scene, tangents = _parse("rtx_host_parse_text", text.encode("utf-8"))
    return (scene, tangents) if with_tangents else scene


def color_correction(rgb: np.ndarray) -> np.ndarray:
    """Image::color_correction (src/image.rs:51-83) on a copy."""
    out = np.ascontiguousarray(rgb, dtype=np.float32).copy()
    load_library().rtx_host_color_correction(out.ctypes.data_as(_FP), out.size)
    return out


def to_bytes(rgb: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(rgb, dtype=np.float32)
    out = np.empty(a.shape, dtype=np.uint8)
    load_library().rtx_host_to_bytes(a.ctypes.data_as(_FP), a.size, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def write_ppm(path: str, rgb: np.ndarray):
    """Image::write (src/image.rs:31-49); rgb has shape (height, width, 3), already colour-corrected."""
    a = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w = a.shape[0], a.shape[1]
    rc = load_library().rtx_host_write_ppm(os.fsencode(path), a.ctypes.data_as(_FP), w, h)
    if rc != 0:
        raise OSError(f"cannot write {path}")


def write_ppm_bytes(path: str, data: np.ndarray):
    """Image::write for an image that is bytes already: uint8 (height, width, 3)."""
    a = np.ascontiguousarray(data, dtype=np.uint8)
    rc = load_library().rtx_host_write_ppm_bytes(os.fsencode(path), a.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                 a.shape[1], a.shape[0])
    if rc != 0:
        raise OSError(f"cannot write {path}")


def scene_fingerprint(scene: Scene) -> int:
    s, keep = scene.to_ctypes()
    return int(load_library().rtx_host_scene_fingerprint(C.byref(s)))


def save_checkpoint(path: str, scene: Scene, seed: int, samples_done: int, sum: np.ndarray):
    """State of a progressive render: the per-pixel sum of samples [0, samples_done)."""
    a = np.ascontiguousarray(sum, dtype=np.float32)
    if a.size != scene.width * scene.height * 3:
        raise ValueError("sum does not have width*height*3 values")
    info = rtx_host_checkpoint(scene.width, scene.height, samples_done, seed, scene_fingerprint(scene))
    if load_library().rtx_host_save_checkpoint(os.fsencode(path), C.byref(info), a.ctypes.data_as(_FP)) != 0:
        raise OSError(f"cannot write {path}")


def load_checkpoint(path: str):
    """Returns (info dict, sum float32 (height, width, 3)); raises OSError if it is not a checkpoint."""
    lib = load_library()
    info = rtx_host_checkpoint()
    if lib.rtx_host_load_checkpoint(os.fsencode(path), C.byref(info), None, 0) != 0:
        raise OSError(f"{path} is not a checkpoint")
    out = np.empty((info.height, info.width, 3), dtype=np.float32)
    if lib.rtx_host_load_checkpoint(os.fsencode(path), C.byref(info), out.ctypes.data_as(_FP), out.size) != 0:
        raise OSError(f"{path} is truncated or has trailing data")
    d = {k: getattr(info, k) for k, _ in rtx_host_checkpoint._fields_}
    return d, out
