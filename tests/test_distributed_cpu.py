"""The process-per-GPU path (the frame's work units dealt out to the ranks, all-reduce of the radiance sums,
divide) exercised with world_size 2 and 3 over gloo on the CPU.  The renderer is replaced by a
test double that fills the rank's buffer with the ORACLE's sums of the rank's share of the frame (blocks of
16 x 8 pixels dealt out round-robin, zeros elsewhere -- what rtx_set_share makes the kernels write), so what is
tested here is the partition, the collective and the final division -- the same
`render_distributed` code the GPU ranks run -- and that the result has the bits of a one-rank render."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from raytracing_amd import reference_scene  # noqa: E402
from raytracing_amd.parallel import sample_range, render_distributed  # noqa: E402


def test_sample_range_partitions_exactly():
    for n in (0, 1, 7, 256, 1024, 1000):
        for world in (1, 2, 3, 4, 8):
            got = [sample_range(n, r, world) for r in range(world)]
            assert sum(c for _, c in got) == n
            pos = 0
            for first, count in got:
                assert first == pos
                pos += count
            assert max(c for _, c in got) - min(c for _, c in got) <= 1
    with pytest.raises(ValueError):
        sample_range(4, 2, 2)


class OracleDouble:
    """Stands in for raytracing_amd.renderer.Renderer on a machine without a GPU."""

    def __init__(self, seed, buf):
        self.seed, self.buf, self.scene, self.share = seed, buf, None, (0, 1)

    def set_share(self, index, count):
        assert 0 <= index < count
        self.share = (index, count)

    def set_scene(self, scene):
        self.scene = scene

    def render_samples(self, first, count, ptr, stream=0):
        from oracle import pyoracle
        assert ptr == self.buf.data_ptr()
        part = pyoracle.render(self.scene, seed=self.seed, order=pyoracle.ORDER_FORWARD,
                               sample_first=first, sample_count=count, output_sum=True)
        # the share's pixels: units of 16 x 8 pixels in row-major order, dealt out round-robin
        h, w = part.shape[:2]
        y, x = np.mgrid[0:h, 0:w]
        unit = (y // 8) * ((w + 15) // 16) + x // 16
        part = np.where((unit % self.share[1] == self.share[0])[..., None], part, np.float32(0)).astype(np.float32)
        self.buf.copy_(torch.from_numpy(part.reshape(-1)))

    def divide(self, ptr, n, divisor, stream=0):
        assert ptr == self.buf.data_ptr() and n == self.buf.numel()
        self.buf.div_(divisor)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, spp, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = reference_scene(width=24, height=16, n_samples=spp)
        buf = torch.zeros(sc.width * sc.height * 3, dtype=torch.float32)
        render_distributed(OracleDouble(5, buf), sc, buf)
        gathered = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(gathered, buf)
        if rank == 0:
            for g in gathered:  # every rank ends with the same image
                assert torch.equal(g, gathered[0])
            np.save(out_path, buf.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,spp", [(2, 8), (3, 7), (2, 1), (2, 0)])
def test_frame_shares_allreduce_matches_single_rank(tmp_path, world, spp, oracle):
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), spp, out), nprocs=world, join=True)
    got = np.load(out)
    sc = reference_scene(width=24, height=16, n_samples=spp)
    if spp == 0:
        # no sample step runs (src/main.rs:14): the zero image (src/image.rs:17), not 0 / 0
        assert not got.any() and not oracle.render(sc, seed=5, order=oracle.ORDER_REFERENCE).any()
        return
    want_fwd = oracle.render(sc, seed=5, order=oracle.ORDER_FORWARD).reshape(-1)
    want_ref = oracle.render(sc, seed=5, order=oracle.ORDER_REFERENCE).reshape(-1)
    # every pixel is one rank's sum plus the other ranks' zeros: the bits of a single rank
    assert np.array_equal(got.view(np.uint32), want_fwd.view(np.uint32))
    err = np.abs(got - want_ref) / np.maximum(np.maximum(np.abs(got), np.abs(want_ref)), 1e-6)
    assert err.max() <= 1e-5
