"""Writes tests/golden/converged_renders.npz with the CPU oracle in the REFERENCE's evaluation
order (recursion, f32 running mean: src/trace.rs:11-71, src/main.rs:23-27): renders of 16 384
samples per pixel, the "converged high-spp reference render" Monte-Carlo renders of the device
are compared with (per-pixel RMSE).  Takes about half a minute on 8 threads.  Run from the
repository root:
    python tests/golden/make_converged.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402
from raytracing_amd import reference_scene, dielectric_scene  # noqa: E402

SPP = 16384
SEED = 1000


def converged_scenes():
    return (("reference", reference_scene(width=64, height=64, n_samples=SPP)),
            ("dielectric", dielectric_scene(width=48, height=32, n_samples=SPP, ray_depth=16)))


if __name__ == "__main__":
    threads = max(1, len(os.sched_getaffinity(0)))
    out = {"spp": np.array(SPP), "seed": np.array(SEED)}
    for name, sc in converged_scenes():
        t0 = time.time()
        out[name] = O.render(sc, seed=SEED, order=O.ORDER_REFERENCE, threads=threads)
        print(name, out[name].shape, "mean", float(out[name].mean()), "%.1f s" % (time.time() - t0))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "converged_renders.npz"), **out)
    print("written")
