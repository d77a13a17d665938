"""Multi-GPU partition of the sample loop (src/main.rs:14 of the reference).

Samples are independent given the counter-based RNG keyed on the GLOBAL sample index,
so rank g of G renders sample indices [first, first+count) of every pixel into its own
radiance-sum buffer; the buffers are added with one all-reduce (RCCL over xGMI when the
backend is "nccl") and divided by the total sample count.  One process per GPU.
"""
from __future__ import annotations

from typing import Tuple


def sample_range(n_samples: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split: the first n_samples % world ranks get one extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} of {world}")
    base, extra = divmod(n_samples, world)
    first = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    return first, count


def render_distributed(renderer, scene, buf, process_group=None):
    """Render `scene` split over the ranks of `process_group` (default group when None).

    `buf` is a float32 torch tensor of width*height*3 elements on this rank's GPU; on
    return every rank holds the mean radiance (row 0 = top).  `renderer` is a
    raytracing_amd.renderer.Renderer bound to the same GPU.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(process_group) if dist.is_initialized() else 1
    rank = dist.get_rank(process_group) if dist.is_initialized() else 0
    first, count = sample_range(scene.n_samples, rank, world)
    stream = torch.cuda.current_stream(buf.device).cuda_stream if buf.is_cuda else 0
    renderer.set_scene(scene)
    if count > 0:
        renderer.render_samples(first, count, buf.data_ptr(), stream)
    else:
        buf.zero_()
    if world > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=process_group)
    # zero samples leave the zero image the reference starts from (src/main.rs:14, src/image.rs:17)
    if scene.n_samples > 0:
        renderer.divide(buf.data_ptr(), buf.numel(), float(scene.n_samples), stream)
    return buf
