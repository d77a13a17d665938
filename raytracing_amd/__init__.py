"""raytracing_amd: MI355X (gfx950) implementation of the reference path tracer's
per-pixel trace loop behind the C ABI of include/rtx.h.

Python here is plumbing only (ctypes binding, scene construction, multi-GPU
orchestration with torch.distributed).  All radiance is computed by the HIP
kernels in csrc/; there is no CPU fallback -- loading the binding raises if
librtx.so is missing.
"""
from . import _abi
from .scene import Prim, Scene, scene_to_text, normalize_quaternion
from .scenes import reference_scene, dielectric_scene, synthetic_scene

__all__ = ["_abi", "Prim", "Scene", "scene_to_text", "normalize_quaternion",
           "reference_scene", "dielectric_scene", "synthetic_scene"]
