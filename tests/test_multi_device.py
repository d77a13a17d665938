"""Several GPUs behind one context (rtx_opts.n_devices): the sample loop of src/main.rs:14 cut
into contiguous ranges, one per device, and the per-device sums added on the devices.

One GPU is all a test box has, so the device list names it several times: every worker then has
its own stream, sum buffer and sample range on that one GPU, and the slice reduce reads the
other workers' buffers directly instead of over peer access -- the same code on every other
count.  What is compared: the oracle's partial sums of the same ranges, added in device order.
"""
import numpy as np
import pytest

from raytracing_amd import _abi, reference_scene, dielectric_scene
from raytracing_amd.parallel import sample_range
from raytracing_amd.renderer import Renderer, RtxError

pytestmark = pytest.mark.gpu


def oracle_split_mean(oracle, sc, seed, parts):
    total = None
    for g in range(parts):
        first, count = sample_range(sc.n_samples, g, parts)
        if count == 0:
            part = np.zeros((sc.height, sc.width, 3), dtype=np.float32)
        else:
            part = oracle.render(sc, seed=seed, order=oracle.ORDER_FORWARD, sample_first=first, sample_count=count,
                                 output_sum=True)
        total = part if total is None else (total + part).astype(np.float32)
    return (total / np.float32(sc.n_samples)).astype(np.float32)


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
def test_split_render_equals_the_oracles_partial_sums_added_in_device_order(oracle, parts, mode):
    sc = dielectric_scene(width=75, height=41, n_samples=11, ray_depth=8)      # 9225 floats: slices with a ragged tail
    want = oracle_split_mean(oracle, sc, 5, parts)
    with Renderer(seed=5, devices=[0] * parts, mode=mode) as r:
        assert r.device_count() == parts
        got = r.render(sc)
        again = r.render(sc)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
    assert np.array_equal(got.view(np.uint32), again.view(np.uint32))
    # against the unsplit image: only the order of `parts` additions differs
    with Renderer(seed=5) as r:
        one = r.render(sc)
    err = np.abs(one - got) / np.maximum(np.maximum(np.abs(one), np.abs(got)), 1e-6)
    assert err.max() <= 1e-6


def test_more_devices_than_samples_zero_samples_and_depth_zero(oracle):
    sc = reference_scene(width=33, height=20, n_samples=2)
    want = oracle_split_mean(oracle, sc, 3, 4)
    with Renderer(seed=3, devices=[0] * 4) as r:
        got = r.render(sc)                                   # two workers get no sample
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
        assert not r.render(sc.replace(n_samples=0)).any()
        assert not r.render(sc.replace(ray_depth=0)).any()


def test_bytes_progressive_and_statistics_across_devices(oracle):
    from raytracing_amd import host
    sc = reference_scene(width=64, height=40, n_samples=12)
    with Renderer(seed=9, devices=[0, 0, 0], collect_stats=True) as r:
        mean = r.render(sc)
        st = r.stats()
        data = r.render_bytes(sc)
        # progressive: every step is split over the devices, the state to save is the added-up sum
        r.progress_begin(sc)
        r.progress_step(5)
        r.progress_step(7)
        pm, done = r.progress_read("mean")
        state, _ = r.progress_read("sum")
        pb, _ = r.progress_read("bytes")
        assert done == 12
    _, want_stats = oracle.render(sc, seed=9, order=oracle.ORDER_FORWARD, want_stats=True)
    st.pop("bvh_node_tests")
    assert st == want_stats                                  # the workers' counters add up to the whole
    want_bytes = host.to_bytes(host.color_correction(mean))
    assert np.abs(data.astype(int) - want_bytes.astype(int)).max() <= 1
    # steps of 5 and 7 samples cut 2+2+1 and 3+2+2: another order of additions than 4+4+4
    err = np.abs(pm - mean) / np.maximum(np.maximum(np.abs(pm), np.abs(mean)), 1e-6)
    assert err.max() <= 1e-6
    assert np.array_equal((state / np.float32(12)).view(np.uint32), pm.view(np.uint32))
    assert np.abs(pb.astype(int) - want_bytes.astype(int)).max() <= 1
    # resume in a single-GPU context from the multi-GPU state
    with Renderer(seed=9) as r:
        r.progress_begin(sc.replace(n_samples=16), sum=state, samples_done=12)
        r.progress_step(4)
        resumed, done = r.progress_read("mean")
        assert done == 16
    with Renderer(seed=9) as r:
        whole = r.render(sc.replace(n_samples=16))
    err = np.abs(resumed - whole) / np.maximum(np.maximum(np.abs(resumed), np.abs(whole)), 1e-6)
    assert err.max() <= 1e-6


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
    want = oracle_split_mean(oracle, sc, 5, 2)
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
        # a sum of two terms is the same addition whoever adds: both reduces give the oracle's bits
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


def test_full_hd_frame_split_in_two(oracle):
    """The slice reduce at a real size: 1920 x 1080, two workers, against the unsplit render."""
    sc = reference_scene(width=1920, height=1080, n_samples=4)
    with Renderer(seed=1) as r:
        one = r.render(sc)
    with Renderer(seed=1, devices=[0, 0]) as r:
        two = r.render(sc)
        ms = r.last_kernel_ms()
    assert ms > 0
    err = np.abs(one - two) / np.maximum(np.maximum(np.abs(one), np.abs(two)), 1e-6)
    assert err.max() <= 1e-6 and np.isfinite(two).all()


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
