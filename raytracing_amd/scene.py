"""Host-side scene description: the Python view of the reference's `Scene`
(src/parser.rs:12-24) before it is flattened into the plain-data `rtx_scene` of
include/rtx.h.

Only data lives here; nothing in this module computes radiance.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import dataclasses
from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

import numpy as np

from . import _abi

_libm = C.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.tanf.restype = C.c_float
_libm.tanf.argtypes = [C.c_float]

Vec3 = Tuple[float, float, float]


@dataclass
class Prim:
    """One object: src/objects/object.rs:11-27 + src/objects/figures.rs:19-33 defaults."""
    kind: int
    param: Vec3
    position: Vec3 = (0.0, 0.0, 0.0)
    rotation: Tuple[float, float, float, float] = (0.0, 0.0, 0.0, 1.0)  # x, y, z, w
    color: Vec3 = (0.0, 0.0, 0.0)
    emission: Vec3 = (0.0, 0.0, 0.0)
    material: int = _abi.DIFFUSE
    ior: float = 1.0


def normalize_quaternion(x: float, y: float, z: float, w: float):
    """`UnitQuaternion::from_quaternion` as src/parser.rs:209-216 applies it: divide by
    the f32 norm.  nalgebra sums a 4-vector's squares as (x*x + z*z) + (y*y + w*w)."""
    f = np.float32
    x, y, z, w = f(x), f(y), f(z), f(w)
    n = np.sqrt((x * x + z * z) + (y * y + w * w), dtype=np.float32)
    return (float(x / n), float(y / n), float(z / n), float(w / n))


@dataclass
class Scene:
    """Render input.  Camera axes are the CAMERA_RIGHT / _UP / _FORWARD vectors as given
    (not re-orthogonalised, src/parser.rs:127-135)."""
    width: int
    height: int
    ray_depth: int
    n_samples: int
    background: Vec3
    cam_pos: Vec3
    cam_right: Vec3
    cam_up: Vec3
    cam_forward: Vec3
    fov_x: float
    prims: List[Prim] = field(default_factory=list)

    def replace(self, **kw) -> "Scene":
        return dataclasses.replace(self, **kw)

    # src/parser.rs:53-55
    def tangents(self) -> Tuple[float, float]:
        f = np.float32
        tgx = f(_libm.tanf(C.c_float(float(f(self.fov_x) / f(2.0)))))
        aspect = f(self.height) / f(self.width)
        return float(tgx), float(aspect * tgx)

    def light_indices(self) -> List[int]:
        """src/parser.rs:69-89: emissive ellipsoids and boxes, in object order."""
        out = []
        for i, p in enumerate(self.prims):
            e = np.asarray(p.emission, dtype=np.float32)
            if float((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]) == 0.0:
                continue
            if p.kind == _abi.PLANE:
                continue
            out.append(i)
        return out

    def to_ctypes(self, prim_type=_abi.rtx_prim, scene_type=_abi.rtx_scene):
        """Flatten into (scene struct, keep-alive prim array)."""
        arr = (prim_type * max(1, len(self.prims)))()
        for dst, p in zip(arr, self.prims):
            dst.kind = p.kind
            dst.param[:] = p.param
            dst.position[:] = p.position
            dst.rotation[:] = p.rotation
            dst.color[:] = p.color
            dst.emission[:] = p.emission
            dst.material = p.material
            dst.ior = p.ior
        s = scene_type()
        s.width, s.height = self.width, self.height
        s.ray_depth, s.n_samples = self.ray_depth, self.n_samples
        s.background[:] = self.background
        s.cam_pos[:] = self.cam_pos
        s.cam_axis[:] = list(self.cam_right) + list(self.cam_up) + list(self.cam_forward)
        s.tg_fov_x, s.tg_fov_y = self.tangents()
        s.n_prims = len(self.prims)
        s.prims = C.cast(arr, C.POINTER(prim_type))
        return s, arr


_KEYWORD = {_abi.PLANE: "PLANE", _abi.ELLIPSOID: "ELLIPSOID", _abi.BOX: "BOX"}


def scene_to_text(scene: Scene) -> str:
    """Serialise in the reference's line-oriented scene format (src/parser.rs:104-199).
    Floats are printed with enough digits to round-trip through f32 parsing."""
    def f(vals):
        return " ".join(np.format_float_positional(np.float32(v), unique=True, trim="-") for v in vals)

    lines = [
        f"DIMENSIONS {scene.width} {scene.height}",
        f"RAY_DEPTH {scene.ray_depth}",
        f"SAMPLES {scene.n_samples}",
        "",
        "BG_COLOR " + f(scene.background),
        "",
        "CAMERA_POSITION " + f(scene.cam_pos),
        "CAMERA_RIGHT " + f(scene.cam_right),
        "CAMERA_UP " + f(scene.cam_up),
        "CAMERA_FORWARD " + f(scene.cam_forward),
        "CAMERA_FOV_X " + f([scene.fov_x]),
    ]
    for p in scene.prims:
        lines += ["", "NEW_PRIMITIVE", f"{_KEYWORD[p.kind]} " + f(p.param)]
        if tuple(p.position) != (0.0, 0.0, 0.0):
            lines.append("POSITION " + f(p.position))
        if tuple(p.rotation) != (0.0, 0.0, 0.0, 1.0):
            lines.append("ROTATION " + f(p.rotation))
        if tuple(p.color) != (0.0, 0.0, 0.0):
            lines.append("COLOR " + f(p.color))
        if tuple(p.emission) != (0.0, 0.0, 0.0):
            lines.append("EMISSION " + f(p.emission))
        if p.material == _abi.METALLIC:
            lines.append("METALLIC")
        elif p.material == _abi.DIELECTRIC:
            lines.append("DIELECTRIC")
            lines.append("IOR " + f([p.ior]))
    return "\n".join(lines) + "\n"
