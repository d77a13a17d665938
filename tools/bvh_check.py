"""Quick A/B of a librtx build (RTX_LIB): BVH render equals the linear scan bit for bit on a
synthetic scene, then C3 / C5 rates."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracing_amd import _abi, synthetic_scene, dielectric_scene
from raytracing_amd.renderer import Renderer

sc = synthetic_scene(1500, 7, 160, 96, 3, 5)
with Renderer(seed=3, accel=_abi.ACCEL_BRUTE_SCALAR) as r: want = r.render(sc)
for mode in (_abi.MODE_MEGAKERNEL, _abi.MODE_WAVEFRONT):
    with Renderer(seed=3, accel=_abi.ACCEL_BVH, mode=mode) as r: got = r.render(sc)
    print("mode", mode, "bit-identical to the scan:", np.array_equal(want.view(np.uint32), got.view(np.uint32)), flush=True)
for name, sc in (("c3", synthetic_scene(10000, 12345, 1920, 1080, 16, 6)), ("c5", dielectric_scene(1920, 1080, 32, 16))):
    buf = torch.empty(sc.width * sc.height * 3, dtype=torch.float32, device="cuda")
    with Renderer(accel=_abi.ACCEL_BVH) as r:
        r.set_scene(sc)
        best = 1e9
        for _ in range(3):
            r.render_samples(0, sc.n_samples, buf.data_ptr()); best = min(best, r.last_kernel_ms())
    print(name, f"{sc.width * sc.height * sc.n_samples / best / 1e3:8.1f} Msamples/s", flush=True)
