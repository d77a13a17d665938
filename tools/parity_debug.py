"""For seeds that tools/parity_sweep.py reported: where the device and the oracle differ
(pixel, sample, first ray depth at which the difference appears, both values)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import pyoracle as O
from raytracing_amd import _abi
from raytracing_amd.renderer import Renderer
from parity_sweep import random_scene, extreme_scene

accel = {"auto": _abi.ACCEL_AUTO, "scalar": _abi.ACCEL_BRUTE_SCALAR, "bvh": _abi.ACCEL_BVH}[sys.argv[1]]
for seed in [int(a) for a in sys.argv[2:]]:
    feats = os.environ.get("SWEEP_EXTREME")
    sc = random_scene(seed) if not feats else extreme_scene(seed, None if feats == "1" else [x for x in feats.split(",") if x])
    kinds = [p.kind for p in sc.prims]
    print(f"seed {seed}: {len(sc.prims)} objects kinds {kinds.count(0)}/{kinds.count(1)}/{kinds.count(2)}, lights {sc.light_indices()}, "
          f"{sc.width}x{sc.height}, depth {sc.ray_depth}, spp {sc.n_samples}")
    n = sc.width * sc.height * 3
    buf = torch.empty(n, dtype=torch.float32, device="cuda")
    for s in range(sc.n_samples):
        for depth in range(1, sc.ray_depth + 1):
            scd = sc.replace(ray_depth=depth)
            want = O.render(scd, seed=seed, order=O.ORDER_FORWARD, sample_first=s, sample_count=1, output_sum=True)
            with Renderer(seed=seed, accel=accel) as r:
                r.set_scene(scd)
                r.render_samples(s, 1, buf.data_ptr())
                torch.cuda.synchronize()
            got = buf.cpu().numpy().reshape(want.shape)
            diff = (want.view(np.uint32) != got.view(np.uint32)).any(axis=2)
            if diff.any():
                for (y, x) in np.argwhere(diff)[:4]:
                    print(f"  sample {s} depth {depth} pixel ({x},{y}) [index {y * sc.width + x}]: oracle {want[y, x]} device {got[y, x]}")
                break
