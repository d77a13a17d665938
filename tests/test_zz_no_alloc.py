"""rtx_render_samples_static -- the entry point that allocates nothing -- against
rtx_render_samples, through the C ABI: same bits, on two calls in a row, the second on a
smaller image.  The sizes are above the thresholds of the cost-ordered dispatch (64 samples,
8192 tiles), so the static tile tables are what both calls use; the second call finds the
first one's contents in them.

"Allocates nothing" is checked on the device's own free-memory figure (hipMemGetInfo, which sees
a hipMalloc from any library), on contexts whose FIRST render above the thresholds is the static
one, so that no earlier call can have grown a table the static call would then silently reuse.
"""
import numpy as np
import pytest

from raytracing_amd import _abi, reference_scene, dielectric_scene
from raytracing_amd.renderer import Renderer, RtxError

pytestmark = pytest.mark.gpu


def _static_then_plain(sc, torch, seed=17, **opts):
    """Renders `sc` with rtx_render_samples_static as the first render of a fresh context and checks the
    device's free memory around it; then with rtx_render_samples on another fresh context.
    Returns (want, got)."""
    n = sc.width * sc.height * 3
    stream = torch.cuda.current_stream().cuda_stream
    # poisoned outputs, allocated before the call under test: it writes, it does not add
    want = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
    got = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
    with Renderer(seed=seed, **opts) as r:
        # the same launches once on a throw-away context: code objects and whatever the runtime sets up
        # per kernel are in place afterwards, and this context's own tables cannot serve the next one
        r.set_scene(sc)
        r.render_samples_static(0, sc.n_samples, got.data_ptr(), stream)
        torch.cuda.synchronize()
        got.fill_(float("nan"))
    with Renderer(seed=seed, **opts) as r:
        r.set_scene(sc)                                            # the scene block is set_scene's, not the render's
        torch.cuda.synchronize()
        free_before, _ = torch.cuda.mem_get_info()
        r.render_samples_static(0, sc.n_samples, got.data_ptr(), stream)
        torch.cuda.synchronize()
        free_after, _ = torch.cuda.mem_get_info()
        assert free_after == free_before, f"device memory changed by {free_before - free_after} bytes"
    with Renderer(seed=seed, **opts) as r:
        r.set_scene(sc)
        r.render_samples(0, sc.n_samples, want.data_ptr(), stream)
        torch.cuda.synchronize()
    return want.cpu().numpy(), got.cpu().numpy()


def _both(r, sc, torch):
    n = sc.width * sc.height * 3
    r.set_scene(sc)
    stream = torch.cuda.current_stream().cuda_stream
    want = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
    got = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
    r.render_samples(0, sc.n_samples, want.data_ptr(), stream)
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info()
    r.render_samples_static(0, sc.n_samples, got.data_ptr(), stream)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before
    return want.cpu().numpy(), got.cpu().numpy()


@pytest.mark.parametrize("accel", [_abi.ACCEL_BRUTE_LDS, _abi.ACCEL_BVH])
def test_static_entry_point_has_the_bits_of_render_samples(accel):
    import torch
    with Renderer(seed=17, accel=accel) as r:
        big = reference_scene(width=1021, height=771, n_samples=64)      # 128 x 97 = 12 416 tiles
        want, got = _both(r, big, torch)
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
        assert np.isfinite(got).all() and got.mean() > 0.01
        small = dielectric_scene(width=904, height=728, n_samples=72, ray_depth=8)   # 113 x 91 = 10 283 tiles
        want, got = _both(r, small, torch)
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
        # below the thresholds no table is used at all; ragged, tiny and empty ranges
        tiny = reference_scene(width=37, height=21, n_samples=3)
        want, got = _both(r, tiny, torch)
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32))


@pytest.mark.parametrize("accel", [_abi.ACCEL_BRUTE_LDS, _abi.ACCEL_BVH])
def test_first_large_render_of_a_context_is_the_static_one_and_allocates_nothing(accel):
    """No rtx_render_samples ran on the context before, so there is no context-owned table the static
    call could fall back on: 12 416 tiles' costs and order can only be in the static array."""
    import torch
    big = reference_scene(width=1021, height=771, n_samples=64)
    want, got = _static_then_plain(big, torch, accel=accel)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
    assert np.isfinite(got).all() and got.mean() > 0.01
    small = dielectric_scene(width=904, height=728, n_samples=72, ray_depth=8)
    want, got = _static_then_plain(small, torch, accel=accel)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))


def test_static_entry_point_under_graph_capture():
    """What the header advertises: the call captured into a HIP graph (global capture mode: an
    allocation or a wait anywhere in the process would invalidate the capture), replayed twice."""
    import torch
    sc = reference_scene(width=1021, height=771, n_samples=64)
    n = sc.width * sc.height * 3
    want = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
    got = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
    with Renderer(seed=23) as r:
        r.set_scene(sc)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            r.render_samples_static(0, 1, got.data_ptr(), side.cuda_stream)    # code object loaded before the capture
        side.synchronize()
        got.fill_(float("nan"))
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side, capture_error_mode="global"):
            r.render_samples_static(0, sc.n_samples, got.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.isnan(got).any()              # capturing ran nothing
        same = []
        for _ in range(3):
            got.fill_(float("nan"))
            torch.cuda.synchronize()
            graph.replay()
            torch.cuda.synchronize()
            r.render_samples(0, sc.n_samples, want.data_ptr(), 0)
            torch.cuda.synchronize()
            same.append(np.array_equal(want.cpu().numpy().view(np.uint32), got.cpu().numpy().view(np.uint32)))
        del graph
        assert same == [True, True, True], same


def test_static_entry_point_limits():
    import torch
    buf = torch.zeros(16, dtype=torch.float32, device="cuda")
    # more tiles than the static tables hold: 8 x 8 pixel tiles, 262 144 of them at most
    wide = reference_scene(width=8 * 1024, height=8 * 257, n_samples=1)
    with Renderer() as r:
        r.set_scene(wide)
        with pytest.raises(RtxError) as e:
            r.render_samples_static(0, 1, buf.data_ptr(), 0)
        assert e.value.code == _abi.RTX_ERR_UNSUPPORTED and "tiles" in str(e.value)
    # reading the event counters back waits for the stream: not for callers under capture
    with Renderer(collect_stats=True) as r:
        r.set_scene(reference_scene(width=16, height=16, n_samples=1))
        with pytest.raises(RtxError) as e:
            r.render_samples_static(0, 1, buf.data_ptr(), 0)
        assert e.value.code == _abi.RTX_ERR_UNSUPPORTED
    # the wavefront variant keeps its path state in a device block when the BVH is deep (the
    # traversal stacks then fill the LDS)
    from raytracing_amd import synthetic_scene
    with Renderer(mode=_abi.MODE_WAVEFRONT, accel=_abi.ACCEL_BVH) as r:
        r.set_scene(synthetic_scene(n_prims=4000, seed=3, width=16, height=16, n_samples=1, ray_depth=3))
        with pytest.raises(RtxError) as e:
            r.render_samples_static(0, 1, buf.data_ptr(), 0)
        assert e.value.code == _abi.RTX_ERR_UNSUPPORTED
    # ... and with its state in LDS (linear scan; BVH with a shallow tree) it needs nothing
    sc = reference_scene(width=40, height=24, n_samples=5)
    for accel in (_abi.ACCEL_BRUTE_LDS, _abi.ACCEL_BVH):
        with Renderer(seed=3, mode=_abi.MODE_WAVEFRONT, accel=accel) as r:
            want, got = _both(r, sc, torch)
            assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
