"""Pure-Python restatement of the reference's scene parser (src/parser.rs:104-217 and
:50-101).  TEST INFRASTRUCTURE ONLY: the checker for raytracing_amd/csrc/host/scene_parser.cpp.
PARITY UNPINNED: the reference ships no parser tests; the only fixture is its sample scene,
whose values are listed in raytracing_amd.scenes.reference_scene().

Decimal -> f32 conversion uses libc strtof (correctly rounded, like Rust's str::parse::<f32>);
what is restated here is the grammar and the statement semantics.
"""
import ctypes
import ctypes.util
import re

import numpy as np

from raytracing_amd.scene import Prim, Scene

_libc = ctypes.CDLL(ctypes.util.find_library("c") or "libc.so.6")
_libc.strtof.restype = ctypes.c_float
_libc.strtof.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
_FLOAT_RE = re.compile(r"^[+-]?((\d+\.?\d*|\.\d+)([eE][+-]?\d+)?|inf|infinity|nan)$", re.IGNORECASE)
_UINT_RE = re.compile(r"^\+?\d+$")


class Panic(Exception):
    """Where the reference would panic (unwrap on a parse error, index out of range, ...)."""


def parse_f32(tok: str) -> float:
    if not _FLOAT_RE.match(tok):
        raise Panic(f"ParseFloatError: {tok!r}")
    return float(_libc.strtof(tok.encode(), None))


def parse_usize(tok: str) -> int:
    if not _UINT_RE.match(tok):
        raise Panic(f"ParseIntError: {tok!r}")
    v = int(tok)
    if v > 0xFFFFFFFF:
        raise Panic("integer beyond what the C ABI carries")
    return v


def _vec3(tokens):
    try:
        return tuple(parse_f32(tokens[i]) for i in range(3))      # parser.rs:201-207
    except IndexError:
        raise Panic("index out of bounds")


def parse_scene_text(text: str):
    """Returns (Scene, (tg_fov_x, tg_fov_y)) or raises Panic."""
    st = dict(width=None, height=None, bg=None, cam_pos=None, axis=[None, None, None], fov=None,
              depth=None, samples=None)
    prims = []
    f32 = np.float32

    def last():
        if not prims:
            raise Panic("attempt to subtract with overflow")       # parser.objects.len() - 1
        return prims[-1]

    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()                                                # BufRead::lines: no trailing empty line
    for line in lines:
        if line.endswith("\r"):
            line = line[:-1]
        t = line.split(" ")
        try:
            kw = t[0]
            if kw == "DIMENSIONS":
                st["width"], st["height"] = parse_usize(t[1]), parse_usize(t[2])
            elif kw == "RAY_DEPTH":
                st["depth"] = parse_usize(t[1])
            elif kw == "SAMPLES":
                st["samples"] = parse_usize(t[1])
            elif kw == "BG_COLOR":
                st["bg"] = _vec3(t[1:])
            elif kw == "CAMERA_POSITION":
                st["cam_pos"] = _vec3(t[1:])
            elif kw == "CAMERA_RIGHT":
                st["axis"][0] = _vec3(t[1:])
            elif kw == "CAMERA_UP":
                st["axis"][1] = _vec3(t[1:])
            elif kw == "CAMERA_FORWARD":
                st["axis"][2] = _vec3(t[1:])
            elif kw == "CAMERA_FOV_X":
                st["fov"] = parse_f32(t[1])
            elif kw == "NEW_PRIMITIVE":
                pass
            elif kw in ("PLANE", "ELLIPSOID", "BOX"):
                prims.append(Prim({"PLANE": 0, "ELLIPSOID": 1, "BOX": 2}[kw], _vec3(t[1:])))
            elif kw == "POSITION":
                v = _vec3(t[1:]); last().position = v
            elif kw == "ROTATION":
                q = [f32(parse_f32(t[1 + k])) for k in range(4)]   # x y z w, parser.rs:209-216
                p = last()
                n = np.sqrt((q[0] * q[0] + q[2] * q[2]) + (q[1] * q[1] + q[3] * q[3]), dtype=np.float32)
                p.rotation = tuple(float(c / n) for c in q)
            elif kw == "COLOR":
                v = _vec3(t[1:]); last().color = v
            elif kw == "EMISSION":
                v = _vec3(t[1:]); last().emission = v
            elif kw == "METALLIC":
                last().material = 1
            elif kw == "DIELECTRIC":
                p = last(); p.material = 2; p.ior = 1.0
            elif kw == "IOR":
                ior = parse_f32(t[1])
                p = last()
                if p.material == 2:
                    p.ior = ior
        except IndexError:
            raise Panic("index out of bounds")
    for key in ("width", "fov", "cam_pos", "depth", "samples", "bg"):
        if st[key] is None:
            raise Panic(f"called Option::unwrap() on a None value ({key})")
    if any(a is None for a in st["axis"]):
        raise Panic("called Option::unwrap() on a None value (camera axis)")
    scene = Scene(width=st["width"], height=st["height"], ray_depth=st["depth"], n_samples=st["samples"],
                  background=st["bg"], cam_pos=st["cam_pos"], cam_right=st["axis"][0], cam_up=st["axis"][1],
                  cam_forward=st["axis"][2], fov_x=st["fov"], prims=prims)
    return scene, scene.tangents()
