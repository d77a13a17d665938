"""The reference's driver (src/main.rs:33-42) on one or several GPUs of a node:

    python -m raytracing_amd.render scene.txt out.ppm [--size W H] [--spp N] [--depth N] [--seed N]
    python -m torch.distributed.run --nproc-per-node 8 -m raytracing_amd.render scene.txt out.ppm ...

One process per GPU.  Each rank parses the scene (src/parser.rs through librtx_host), renders
its share of the samples (raytracing_amd.parallel), the sums are added with one all-reduce
(RCCL) and rank 0 tone-maps and writes the PPM (src/image.rs through librtx_host).
"""
from __future__ import annotations

import argparse
import os


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("scene", nargs="?", default="assets/scene.txt")
    ap.add_argument("output", nargs="?", default="/tmp/out.ppm")
    ap.add_argument("--size", type=int, nargs=2, default=None, metavar=("W", "H"))
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--mode", default="megakernel", choices=["megakernel", "wavefront"])
    args = ap.parse_args(argv)

    import torch
    import torch.distributed as dist
    from . import _abi, host
    from .parallel import render_distributed
    from .renderer import Renderer

    distributed = "RANK" in os.environ
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if distributed:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    rank = dist.get_rank() if distributed else 0

    scene = host.parse_scene(args.scene)
    if args.size:
        scene = scene.replace(width=args.size[0], height=args.size[1])
    if args.spp is not None:
        scene = scene.replace(n_samples=args.spp)
    if args.depth is not None:
        scene = scene.replace(ray_depth=args.depth)

    buf = torch.zeros(scene.width * scene.height * 3, dtype=torch.float32, device="cuda")
    mode = _abi.MODE_WAVEFRONT if args.mode == "wavefront" else _abi.MODE_MEGAKERNEL
    with Renderer(seed=args.seed, mode=mode, device=local_rank) as r:
        if scene.n_samples > 0:
            render_distributed(r, scene, buf)
        torch.cuda.synchronize()
    if rank == 0:
        image = buf.cpu().numpy().reshape(scene.height, scene.width, 3)
        host.write_ppm(args.output, host.color_correction(image))      # color_correction(); write(&output)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
