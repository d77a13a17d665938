"""What one GPU of an N-GPU job has to do, measured on one GPU: the time of share 0 of N of a workload
(rtx_set_share), times N against the whole frame.   python tools/share_timing.py c4 1,2,4,8"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_workload
from raytracing_amd.renderer import Renderer

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
worlds = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
scene, _ = build_workload(name)
n = scene.width * scene.height * 3
buf = torch.zeros(n, dtype=torch.float32, device="cuda")
base = None
for world in worlds:
    for share in sorted({0, world - 1}):
        with Renderer(seed=1) as r:
            r.set_share(share, world)
            r.set_scene(scene)
            r.render_samples(0, scene.n_samples, buf.data_ptr())
            torch.cuda.synchronize()
            ms = []
            for _ in range(2):
                r.render_samples(0, scene.n_samples, buf.data_ptr())
                torch.cuda.synchronize()
                ms.append(r.last_kernel_ms())
        t = min(ms)
        if base is None:
            base = t
        print(f"{name} share {share} of {world}: {t:.2f} ms, x{world} = {t * world:.1f} ms, efficiency {base / (t * world):.3f}", flush=True)
