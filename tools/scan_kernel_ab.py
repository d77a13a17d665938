"""One- against two-pixel scan megakernel (RTX_LIB selects the build) on synthetic scenes:
    python tools/scan_kernel_ab.py width height spp n_objects..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import _abi, synthetic_scene, reference_scene, dielectric_scene
from raytracing_amd.renderer import Renderer

w, h, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
scenes = [("n=%d" % n, synthetic_scene(int(n), 7, w, h, spp, 6, n_lights=2)) for n in map(int, sys.argv[4:])]
scenes += [("reference", reference_scene(w, h, spp, 6)), ("glass/metal", dielectric_scene(w, h, spp, 16))]
for name, sc in scenes:
    buf = torch.empty(sc.width * sc.height * 3, dtype=torch.float32, device="cuda")
    with Renderer(accel=_abi.ACCEL_BRUTE_LDS) as r:
        r.set_scene(sc)
        best = 1e9
        for _ in range(3):
            r.render_samples(0, sc.n_samples, buf.data_ptr())
            best = min(best, r.last_kernel_ms())
    print(f"{name:12s} {best:9.2f} ms  {sc.width * sc.height * sc.n_samples / best / 1e3:8.1f} Msamples/s  sum {float(buf.double().sum()):.4f}", flush=True)
