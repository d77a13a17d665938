"""The workloads named by the benchmark configurations, built programmatically.

* reference_scene(): the values of the reference's only shipped scene
  (assets/scene.txt:1-52): a Cornell-style room of five planes, one emissive box,
  one rotated box and one sphere, 512x512, depth 6, 10 samples.
* dielectric_scene(): the same room with the sphere turned to glass, the tall
  box to metal and a 5x5 grid of alternating glass / metal spheres; depth 16.
* synthetic_scene(): a seeded random scene of `n_prims` objects in a +-50 room.
"""
from __future__ import annotations

import numpy as np

from . import _abi
from .scene import Prim, Scene, normalize_quaternion


def reference_scene(width: int = 512, height: int = 512, n_samples: int = 10, ray_depth: int = 6) -> Scene:
    prims = [
        # assets/scene.txt:13-36 -- floor, back wall, ceiling, left (red), right (green)
        Prim(_abi.PLANE, (0, 1, 0), position=(0, -5, 0), color=(1, 1, 1)),
        Prim(_abi.PLANE, (0, 0, 1), position=(0, 0, -5), color=(1, 1, 1)),
        Prim(_abi.PLANE, (0, -1, 0), position=(0, 5, 0), color=(1, 1, 1)),
        Prim(_abi.PLANE, (1, 0, 0), position=(-5, 0, 0), color=(1, 0.25, 0.25)),
        Prim(_abi.PLANE, (-1, 0, 0), position=(5, 0, 0), color=(0.25, 1, 0.25)),
        # assets/scene.txt:38-41 -- the light: no COLOR line, so albedo stays 0
        Prim(_abi.BOX, (2, 0.1, 2), position=(0, 5, 0), emission=(2, 2, 2)),
        # assets/scene.txt:43-47 -- tall box, rotated 45 degrees about y
        Prim(_abi.BOX, (1.5, 3, 1.5), position=(-2, -2, -1),
             rotation=normalize_quaternion(0, 0.3826834, 0, 0.9238795), color=(0.6, 0.8, 1)),
        # assets/scene.txt:49-52
        Prim(_abi.ELLIPSOID, (2, 2, 2), position=(2, -3, 2), color=(1, 0.8, 0.6)),
    ]
    return Scene(width=width, height=height, ray_depth=ray_depth, n_samples=n_samples,
                 background=(0, 0, 0), cam_pos=(0, 0, 15), cam_right=(1, 0, 0), cam_up=(0, 1, 0),
                 cam_forward=(0, 0, -1), fov_x=0.927295218, prims=prims)


def dielectric_scene(width: int = 1920, height: int = 1080, n_samples: int = 256, ray_depth: int = 16) -> Scene:
    """Divergence stress: glass and metal everywhere, long paths."""
    base = reference_scene(width, height, n_samples, ray_depth)
    prims = list(base.prims)
    prims[6] = Prim(_abi.BOX, (1.5, 3, 1.5), position=(-2, -2, -1),
                    rotation=base.prims[6].rotation, color=(0.9, 0.9, 0.95), material=_abi.METALLIC)
    prims[7] = Prim(_abi.ELLIPSOID, (2, 2, 2), position=(2, -3, 2), color=(1, 0.95, 0.9),
                    material=_abi.DIELECTRIC, ior=1.5)
    for gy in range(5):
        for gx in range(5):
            glass = (gx + gy) % 2 == 0
            prims.append(Prim(
                _abi.ELLIPSOID, (0.45, 0.45, 0.45),
                position=(-3.6 + 1.8 * gx, -1.0 + 1.2 * gy, 3.5 - 0.3 * gy),
                color=(0.95, 0.97, 1.0) if glass else (0.95, 0.8, 0.6),
                material=_abi.DIELECTRIC if glass else _abi.METALLIC,
                ior=1.5 if glass else 1.0))
    return base.replace(prims=prims)


def random_unit_quaternion(rng: np.random.Generator):
    q = rng.normal(size=4)
    q = q / np.linalg.norm(q)
    return normalize_quaternion(*[float(np.float32(v)) for v in q])


def synthetic_scene(n_prims: int = 10000, seed: int = 12345, width: int = 1920, height: int = 1080,
                    n_samples: int = 64, ray_depth: int = 6, n_lights: int = 8) -> Scene:
    """Six bounding planes of a +-50 room plus random ellipsoids and boxes (50/50):
    centres U[-45,45]^3, radii / half-sizes U[0.3,1.5]^3, random rotations, materials
    70 % diffuse / 15 % metallic / 15 % dielectric (ior U[1.3,1.8]), albedo U[0.2,1]^3;
    the first `n_lights` non-plane objects are emissive (emission U[5,20], grey)."""
    rng = np.random.default_rng(seed)
    f = lambda a: tuple(float(np.float32(v)) for v in a)
    prims = [
        Prim(_abi.PLANE, (0, 1, 0), position=(0, -50, 0), color=(0.8, 0.8, 0.8)),
        Prim(_abi.PLANE, (0, -1, 0), position=(0, 50, 0), color=(0.8, 0.8, 0.8)),
        Prim(_abi.PLANE, (1, 0, 0), position=(-50, 0, 0), color=(0.8, 0.4, 0.4)),
        Prim(_abi.PLANE, (-1, 0, 0), position=(50, 0, 0), color=(0.4, 0.8, 0.4)),
        Prim(_abi.PLANE, (0, 0, 1), position=(0, 0, -50), color=(0.8, 0.8, 0.8)),
        Prim(_abi.PLANE, (0, 0, -1), position=(0, 0, 50), color=(0.8, 0.8, 0.8)),
    ]
    n_obj = max(0, n_prims - len(prims))
    for k in range(n_obj):
        kind = _abi.ELLIPSOID if rng.random() < 0.5 else _abi.BOX
        size = f(rng.uniform(0.3, 1.5, size=3))
        pos = f(rng.uniform(-45.0, 45.0, size=3))
        rot = random_unit_quaternion(rng)
        albedo = f(rng.uniform(0.2, 1.0, size=3))
        m = rng.random()
        ior = float(np.float32(rng.uniform(1.3, 1.8)))
        e = float(np.float32(rng.uniform(5.0, 20.0)))
        if k < n_lights:
            prims.append(Prim(kind, size, position=pos, rotation=rot, color=albedo, emission=(e, e, e)))
        elif m < 0.70:
            prims.append(Prim(kind, size, position=pos, rotation=rot, color=albedo))
        elif m < 0.85:
            prims.append(Prim(kind, size, position=pos, rotation=rot, color=albedo, material=_abi.METALLIC))
        else:
            prims.append(Prim(kind, size, position=pos, rotation=rot, color=albedo,
                              material=_abi.DIELECTRIC, ior=ior))
    return Scene(width=width, height=height, ray_depth=ray_depth, n_samples=n_samples,
                 background=(0, 0, 0), cam_pos=(0, 0, 48), cam_right=(1, 0, 0), cam_up=(0, 1, 0),
                 cam_forward=(0, 0, -1), fov_x=1.2, prims=prims)
