"""Brute-force scan against BVH traversal by object count: Msamples/s of both on synthetic
scenes (room of 6 planes + n random ellipsoids and boxes, 2 lights), 960x540, 16 spp."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import _abi, synthetic_scene, dielectric_scene
from raytracing_amd.renderer import Renderer


def rate(sc, accel, mode=_abi.MODE_MEGAKERNEL):
    n = sc.width * sc.height * 3
    buf = torch.empty(n, dtype=torch.float32, device="cuda")
    with Renderer(accel=accel, mode=mode) as r:
        r.set_scene(sc)
        r.render_samples(0, sc.n_samples, buf.data_ptr())
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            r.render_samples(0, sc.n_samples, buf.data_ptr())
            best = min(best, r.last_kernel_ms())
    return sc.width * sc.height * sc.n_samples / best / 1e3, float(buf.double().sum())


for n in [int(a) for a in sys.argv[1:]] or [12, 16, 24, 32, 48, 64, 128, 256, 512]:
    sc = synthetic_scene(n, 7, 960, 540, 16, 6, n_lights=2)
    b, cb = rate(sc, _abi.ACCEL_BRUTE_LDS)
    v, cv = rate(sc, _abi.ACCEL_BVH)
    print(f"n={n:4d}  brute {b:8.1f}  bvh {v:8.1f} Msamples/s  ratio {v / b:5.2f}  same image: {cb == cv}", flush=True)
sc = dielectric_scene(960, 540, 16, 16)
b, cb = rate(sc, _abi.ACCEL_BRUTE_LDS)
v, cv = rate(sc, _abi.ACCEL_BVH)
print(f"c5 (34 objects, depth 16): brute {b:8.1f}  bvh {v:8.1f}  ratio {v / b:5.2f}  same image: {cb == cv}")
