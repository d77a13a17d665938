"""Bit-for-bit comparison of the device with the CPU oracle at the benchmark's full sizes
(GPU box; the oracle runs on all host threads, about a minute for C2):
    python tools/full_size_parity.py [--full-paths] [c2] [c5] [c3]
--full-paths: the oracle follows every path to the depth limit, as the reference does (the
device ends paths whose throughput is zero; the image must be the same).
c2: reference scene 1920x1080, 256 spp.  c5: glass / metal scene, depth 16, 1920x1080, 64 spp.
c3: 10 000 objects, 960x540, 4 spp (the oracle scans every object for every ray).
c5q: c5 at 128 spp, which the device renders in chunks drawn from its work queue."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import pyoracle as O
from raytracing_amd import reference_scene, dielectric_scene, synthetic_scene
from raytracing_amd.renderer import Renderer

threads = int(os.environ.get("OMP_NUM_THREADS", "16"))
work = {"c2": lambda: reference_scene(1920, 1080, 256, 6), "c5": lambda: dielectric_scene(1920, 1080, 64, 16),
        "c3": lambda: synthetic_scene(10000, 12345, 960, 540, 4, 6),
        # enough samples for the one-pixel kernels' work queue (two chunks of 60 after the 8-sample probe)
        "c5q": lambda: dielectric_scene(1920, 1080, 128, 16)}
ok = True
full_paths = "--full-paths" in sys.argv
names = [a for a in sys.argv[1:] if not a.startswith("--")]
for name in names or ["c2"]:
    sc = work[name]()
    t0 = time.time()
    with Renderer(seed=1) as r:
        got = r.render(sc)
    t1 = time.time()
    want = O.render(sc, seed=1, order=O.ORDER_FORWARD, threads=threads, full_paths=full_paths)
    t2 = time.time()
    diff = (want.view(np.uint32) != got.view(np.uint32)).any(axis=2)
    ok = ok and not diff.any()
    print(f"{name}: {sc.width}x{sc.height} x {sc.n_samples} spp, {len(sc.prims)} objects: {int(diff.sum())} of {diff.size} pixels differ "
          f"(device {t1 - t0:.1f} s incl. set-up, oracle {t2 - t1:.1f} s on {threads} threads), mean radiance {float(want.mean()):.6f}", flush=True)
sys.exit(0 if ok else 1)
