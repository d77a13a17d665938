"""Several GPUs behind one context (rtx_opts.n_devices) and one context per GPU with rtx_set_share: the
frame's work units dealt out round-robin, every sample step of src/main.rs:14 for a unit on the GPU that
owns it, the per-device buffers (zero outside their share) added on the devices.

One GPU is all a test box has, so the device list names it several times: every worker then has
its own stream, sum buffer and share of the frame on that one GPU, and the slice reduce reads the
other workers' buffers directly instead of over peer access -- the same code on every other
count.  What is compared: the oracle's image, bit for bit -- a pixel's samples are added on one GPU in
order, so the number of GPUs changes nothing.
"""
import numpy as np
import pytest

from raytracing_amd import _abi, reference_scene, dielectric_scene, synthetic_scene
from raytracing_amd.renderer import Renderer, RtxError

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_one_device_in_the_list_is_the_single_gpu_path(oracle):
    sc = reference_scene(width=96, height=64, n_samples=8)
    with Renderer(seed=7) as r:
        want = r.render(sc)
    with Renderer(seed=7, devices=[0]) as r:
        assert r.device_count() == 1
        got = r.render(sc)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
    assert np.array_equal(oracle.render(sc, seed=7, order=oracle.ORDER_FORWARD).view(np.uint32), got.view(np.uint32))


@pytest.mark.parametrize("parts", [2, 3, 5])
@pytest.mark.parametrize("mode", [_abi.MODE_MEGAKERNEL, _abi.MODE_WAVEFRONT])
def test_a_frame_dealt_out_to_several_workers_has_the_one_gpu_bits(oracle, parts, mode):
    sc = dielectric_scene(width=75, height=41, n_samples=11, ray_depth=8)      # 9225 floats: slices with a ragged tail
    want = oracle.render(sc, seed=5, order=oracle.ORDER_FORWARD)
    with Renderer(seed=5, devices=[0] * parts, mode=mode) as r:
        assert r.device_count() == parts
        got = r.render(sc)
        again = r.render(sc)
    assert np.array_equal(bits(want), bits(got))
    assert np.array_equal(bits(got), bits(again))
    with Renderer(seed=5, mode=mode) as r:
        assert np.array_equal(bits(r.render(sc)), bits(got))


@pytest.mark.parametrize("scene_name", ["reference", "coloured light", "bvh", "bvh many lights"])
def test_every_kernel_kind_dealt_out_to_three_workers(oracle, scene_name):
    """The two-pixel scan kernel (16 x 8 units), the one-pixel scan kernel, the BVH kernels: three workers, both
    kernel variants, against the oracle and with the counters of the three shares adding up."""
    import dataclasses
    if scene_name == "reference":
        sc = reference_scene(width=100, height=37, n_samples=6)
    elif scene_name == "coloured light":
        sc = reference_scene(width=100, height=37, n_samples=6)
        prims = list(sc.prims)
        prims[5] = dataclasses.replace(prims[5], color=(0.78, 0.78, 0.78))
        sc = sc.replace(prims=prims)
    elif scene_name == "bvh":
        sc = dielectric_scene(width=61, height=50, n_samples=5, ray_depth=10)
    else:
        sc = synthetic_scene(400, 21, 70, 45, 3, 6)
    want, want_stats = oracle.render(sc, seed=8, order=oracle.ORDER_FORWARD, want_stats=True)
    for mode in (_abi.MODE_MEGAKERNEL, _abi.MODE_WAVEFRONT):
        with Renderer(seed=8, devices=[0, 0, 0], mode=mode) as r:
            assert np.array_equal(bits(want), bits(r.render(sc))), mode
        with Renderer(seed=8, mode=mode, collect_stats=True) as r:
            r.render(sc)
            one = r.stats()
        with Renderer(seed=8, devices=[0, 0, 0], mode=mode, collect_stats=True) as r:
            assert np.array_equal(bits(want), bits(r.render(sc))), mode
            st = r.stats()
        assert st == one            # the shares' counters add up to the one-GPU numbers, node tests included
        if "bvh" not in scene_name:
            st.pop("bvh_node_tests")
            assert st == want_stats


def test_more_workers_than_units_zero_samples_and_depth_zero(oracle):
    sc = reference_scene(width=33, height=20, n_samples=2)      # 3 x 3 blocks of 16 x 8 pixels, 5 x 3 tiles
    want = oracle.render(sc, seed=3, order=oracle.ORDER_FORWARD)
    for mode in (_abi.MODE_MEGAKERNEL, _abi.MODE_WAVEFRONT):
        with Renderer(seed=3, devices=[0] * 16, mode=mode) as r:
            got = r.render(sc)                                   # some workers own nothing
            assert np.array_equal(bits(want), bits(got))
            assert not r.render(sc.replace(n_samples=0)).any()
            assert not r.render(sc.replace(ray_depth=0)).any()


def test_one_context_per_share_adds_up_to_the_frame(oracle):
    """rtx_set_share, the process-per-GPU form: each context writes its share's sums and zeros elsewhere; the
    buffers added are the oracle's sums bit for bit, each pixel non-zero in at most one buffer; continued sums
    (rtx_accumulate_samples) and the allocation-free entry point included."""
    import torch
    sc = reference_scene(width=90, height=52, n_samples=7)
    want = oracle.render(sc, seed=12, order=oracle.ORDER_FORWARD, output_sum=True).reshape(-1)
    n = sc.width * sc.height * 3
    for mode in (_abi.MODE_MEGAKERNEL, _abi.MODE_WAVEFRONT):
        for world in (1, 2, 4):
            total = np.zeros(n, dtype=np.float32)
            owners = np.zeros(n, dtype=np.int32)
            for rank in range(world):
                with Renderer(seed=12, mode=mode) as r:
                    r.set_share(rank, world)
                    r.set_scene(sc)
                    buf = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
                    torch.cuda.synchronize()
                    if rank % 2 == 0:
                        r.render_samples(0, 7, buf.data_ptr())
                    else:
                        # in two steps, the second continuing the first; then the same through the static entry point
                        r.render_samples(0, 3, buf.data_ptr())
                        r.accumulate_samples(3, 4, buf.data_ptr())
                        other = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
                        torch.cuda.synchronize()
                        r.render_samples_static(0, 7, other.data_ptr())
                        torch.cuda.synchronize()
                        assert torch.equal(buf.view(torch.int32), other.view(torch.int32))
                    torch.cuda.synchronize()
                    part = buf.cpu().numpy()
                assert np.isfinite(part).all()
                owners += (part != 0).astype(np.int32)
                total = (total + part).astype(np.float32)
            assert owners.max() <= 1
            assert np.array_equal(bits(want), bits(total)), (mode, world)
    with Renderer(seed=12) as r:
        for bad in ((2, 2), (0, 0)):
            with pytest.raises(RtxError) as e:
                r.set_share(*bad)
            assert e.value.code == _abi.RTX_ERR_INVALID_ARG
    with Renderer(seed=12, devices=[0, 0]) as r:
        with pytest.raises(RtxError) as e:
            r.set_share(0, 2)
        assert e.value.code == _abi.RTX_ERR_UNSUPPORTED


def test_bytes_progressive_and_statistics_across_devices(oracle):
    from raytracing_amd import host
    sc = reference_scene(width=64, height=40, n_samples=12)
    with Renderer(seed=9, devices=[0, 0, 0], collect_stats=True) as r:
        mean = r.render(sc)
        st = r.stats()
        data = r.render_bytes(sc)
        # progressive: every worker continues its own pixels, the state to save is the added-up sum
        r.progress_begin(sc)
        r.progress_step(5)
        r.progress_step(7)
        pm, done = r.progress_read("mean")
        state, _ = r.progress_read("sum")
        pb, _ = r.progress_read("bytes")
        assert done == 12
    want, want_stats = oracle.render(sc, seed=9, order=oracle.ORDER_FORWARD, want_stats=True)
    st.pop("bvh_node_tests")
    assert st == want_stats                                  # the workers' counters add up to the whole
    assert np.array_equal(bits(want), bits(mean))
    want_bytes = host.to_bytes(host.color_correction(mean))
    assert np.abs(data.astype(int) - want_bytes.astype(int)).max() <= 1
    # however the steps are cut: the bits of the one-shot render
    assert np.array_equal(bits(pm), bits(mean))
    assert np.array_equal(bits(state / np.float32(12)), bits(pm))
    assert np.abs(pb.astype(int) - want_bytes.astype(int)).max() <= 1
    want16 = oracle.render(sc.replace(n_samples=16), seed=9, order=oracle.ORDER_FORWARD)
    # resume the saved state on one GPU, on two workers and on five, in both kernel variants
    for opts in (dict(), dict(devices=[0, 0]), dict(devices=[0] * 5), dict(devices=[0, 0, 0], mode=_abi.MODE_WAVEFRONT)):
        with Renderer(seed=9, **opts) as r:
            r.progress_begin(sc.replace(n_samples=16), sum=state, samples_done=12)
            back, _ = r.progress_read("sum")
            assert np.array_equal(bits(back), bits(state))     # the shares put together again
            r.progress_step(4)
            resumed, done = r.progress_read("mean")
            assert done == 16
        assert np.array_equal(bits(want16), bits(resumed)), opts


def test_progress_read_of_everything_at_once_and_the_image_left_on_the_device(oracle):
    """One rtx_progress_read for mean, sum and bytes adds the GPUs' sums once and returns what three
    separate reads return; rtx_render_device leaves the bits of rtx_render in a device buffer."""
    import torch
    sc = dielectric_scene(width=75, height=41, n_samples=9, ray_depth=8)
    for opts in (dict(devices=[0, 0, 0]), dict(devices=[0], reduce=_abi.REDUCE_RCCL), dict()):
        with Renderer(seed=4, **opts) as r:
            want = r.render(sc)
            r.progress_begin(sc)
            r.progress_step(4)
            r.progress_step(5)
            m1, _ = r.progress_read("mean")
            s1, _ = r.progress_read("sum")
            b1, _ = r.progress_read("bytes")
            m, s, b, done = r.progress_read_all()
            assert done == 9
            assert np.array_equal(m.view(np.uint32), m1.view(np.uint32))
            assert np.array_equal(s.view(np.uint32), s1.view(np.uint32))
            assert np.array_equal(b, b1)
            assert np.array_equal((s / np.float32(9)).view(np.uint32), m.view(np.uint32))
            # the image left on the device
            buf = torch.full((sc.width * sc.height * 3,), float("nan"), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            r.set_scene(sc)
            r.render_device(sc.n_samples, buf.data_ptr())
            got = buf.cpu().numpy().reshape(want.shape)
            assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
            r.render_device(0, buf.data_ptr())
            assert not buf.cpu().numpy().any()
            if opts:
                assert r.last_kernel_ms() >= 0.0


def test_distinct_gpus_both_reduces(oracle):
    """Two different GPUs: peer access, the slice reduce over it, and RCCL with two ranks.  Skipped on a
    one-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    sc = dielectric_scene(width=75, height=41, n_samples=11, ray_depth=8)
    want = oracle.render(sc, seed=5, order=oracle.ORDER_FORWARD)
    for reduce in (_abi.REDUCE_P2P, _abi.REDUCE_RCCL):
        with Renderer(seed=5, devices=[0, 1], reduce=reduce) as r:
            got = r.render(sc)
            buf = torch.full((sc.width * sc.height * 3,), float("nan"), dtype=torch.float32, device="cuda:0")
            torch.cuda.synchronize()
            r.render_device(sc.n_samples, buf.data_ptr())
            dev = buf.cpu().numpy().reshape(got.shape)
            data = r.render_bytes(sc)
        assert np.array_equal(got.view(np.uint32), dev.view(np.uint32))
        assert data.shape == got.shape
        # every value is one GPU's sum plus the other's zero: both reduces give the oracle's bits
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32))


def test_render_into_pinned_memory_has_the_same_bits():
    from raytracing_amd.renderer import PinnedImage
    sc = reference_scene(width=97, height=43, n_samples=5)
    img = PinnedImage(sc.height, sc.width)
    try:
        for opts in (dict(), dict(devices=[0, 0, 0])):
            with Renderer(seed=6, **opts) as r:
                want = r.render(sc)
                img.array.fill(np.nan)
                got = r.render(sc, out=img.array)
                assert got is img.array
                assert np.array_equal(want.view(np.uint32), img.array.view(np.uint32))
    finally:
        img.close()


def test_full_hd_frame_split_in_two_and_in_eight(oracle):
    """The slice reduce and the launch plan's tables (cost order, work queue) at a real size: 1920 x 1080 at 64
    samples, two and eight workers, against the unsplit render -- the reference scene (two-pixel kernel, work queue),
    the glass scene (cost order and queue) -- bit for bit."""
    for sc in (reference_scene(width=1920, height=1080, n_samples=64), dielectric_scene(width=1920, height=1080, n_samples=64, ray_depth=8)):
        with Renderer(seed=1) as r:
            one = r.render(sc)
        for parts in (2, 8):
            with Renderer(seed=1, devices=[0] * parts) as r:
                got = r.render(sc)
                ms = r.last_kernel_ms()
            assert ms > 0
            assert np.array_equal(bits(one), bits(got)), parts


def test_rccl_all_reduce_with_one_rank(oracle):
    """RTX_REDUCE_RCCL on the one GPU there is: communicator set-up, the grouped ncclAllReduce and
    the finish on GPU 0 all run (with a single rank the sum is the rank's own buffer)."""
    sc = reference_scene(width=64, height=48, n_samples=6)
    with Renderer(seed=2) as r:
        want = r.render(sc)
        want_bytes = r.render_bytes(sc)
    with Renderer(seed=2, devices=[0], reduce=_abi.REDUCE_RCCL) as r:
        got = r.render(sc)
        got_bytes = r.render_bytes(sc)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
    assert np.array_equal(want_bytes, got_bytes)
    with pytest.raises(RtxError) as e:
        Renderer(devices=[0, 0], reduce=_abi.REDUCE_RCCL)
    assert e.value.code == _abi.RTX_ERR_INVALID_ARG


def test_bad_device_lists():
    with pytest.raises(RtxError) as e:
        Renderer(devices=[0, 99])
    assert e.value.code == _abi.RTX_ERR_INVALID_ARG
    with pytest.raises(RtxError) as e:
        Renderer(devices=[0] * 17)
    assert e.value.code == _abi.RTX_ERR_INVALID_ARG
