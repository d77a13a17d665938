"""Writes tests/golden/reference_scene_64x64.npz with the CPU oracle: the 64x64 primary
pass of the reference scene (ids, t, normals), a 4-spp seeded render and a zero-randomness
render, both in the reference's evaluation order.  Run from the repository root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402
from raytracing_amd import reference_scene  # noqa: E402

CONST_WORDS = (0x40000000, 0x9E3779B9, 0x7F4A7C15, 0xDEADBEEF)

sc = reference_scene(width=64, height=64, n_samples=4)
ids, t, n, ins = O.primary(sc)
img = O.render(sc, seed=1, order=O.ORDER_REFERENCE)
cx = O.render(sc, order=O.ORDER_REFERENCE, rng_mode=O.RNG_CONST, const_words=CONST_WORDS)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "reference_scene_64x64.npz"),
                    hit_id=ids, t=t, normal=n, inside=ins, radiance_seed1_4spp=img, radiance_const_xi=cx,
                    const_words=np.array(CONST_WORDS, dtype=np.uint32))
print("written")

# second file: materials, long paths, rotated objects, several lights -- in the summation order
# the device uses (sum front to back, divide once), so a device render must equal it bit for bit
from raytracing_amd import dielectric_scene, synthetic_scene  # noqa: E402

out = {}
for name, scene, seed in (("reference", reference_scene(width=64, height=64, n_samples=4), 1),
                          ("dielectric", dielectric_scene(width=48, height=32, n_samples=3, ray_depth=12), 5),
                          ("synthetic", synthetic_scene(n_prims=60, seed=9, width=48, height=32, n_samples=2, ray_depth=5,
                                                        n_lights=3), 7)):
    ids, t, n, ins = O.primary(scene)
    out[name + "_hit_id"], out[name + "_t"] = ids, t
    out[name + "_forward"] = O.render(scene, seed=seed, order=O.ORDER_FORWARD)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "forward_order_renders.npz"), **out)
print("written")
