"""One process per GPU: the frame of `render(&mut Scene)` (src/main.rs:13-31 of the reference) spread over ranks.

The frame is cut into the kernels' work units (blocks of 8 x 8 or 16 x 8 pixels), dealt out round-robin: rank r of
W renders every sample step of units r, r + W, ... into its own radiance-sum buffer, which is zero elsewhere
(`Renderer.set_share`), the buffers are added with one all-reduce (RCCL over xGMI when the backend is "nccl") and
divided by the sample count.  A pixel's samples are all traced and added by one rank, in order, and adding zeros
changes no bit: every rank ends with the image one GPU alone would have rendered.
"""
from __future__ import annotations

from typing import Tuple


def sample_range(n_samples: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of a sample range: the first n_samples % world parts get one extra.  (Not how a
    frame is spread over ranks -- see render_distributed -- but how a caller cuts the loop into steps.)"""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} of {world}")
    base, extra = divmod(n_samples, world)
    first = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    return first, count


def render_distributed(renderer, scene, buf, process_group=None):
    """Render `scene` spread over the ranks of `process_group` (default group when None).

    `buf` is a float32 torch tensor of width*height*3 elements on this rank's GPU; on
    return every rank holds the mean radiance (row 0 = top).  `renderer` is a
    raytracing_amd.renderer.Renderer bound to the same GPU.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(process_group) if dist.is_initialized() else 1
    rank = dist.get_rank(process_group) if dist.is_initialized() else 0
    stream = torch.cuda.current_stream(buf.device).cuda_stream if buf.is_cuda else 0
    renderer.set_share(rank, world)
    renderer.set_scene(scene)
    if scene.n_samples > 0:
        renderer.render_samples(0, scene.n_samples, buf.data_ptr(), stream)
    else:
        buf.zero_()
    if world > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=process_group)
    # zero samples leave the zero image the reference starts from (src/main.rs:14, src/image.rs:17)
    if scene.n_samples > 0:
        renderer.divide(buf.data_ptr(), buf.numel(), float(scene.n_samples), stream)
    return buf
