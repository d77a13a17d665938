"""rtx_render_samples_static -- the entry point that allocates nothing -- against
rtx_render_samples, through the C ABI: same bits, on two calls in a row, the second on a
smaller image.  The sizes are above the thresholds of the cost-ordered dispatch (64 samples,
8192 tiles), so the static tile tables are what both calls use; the second call finds the
first one's contents in them."""
import numpy as np
import pytest

from raytracing_amd import _abi, reference_scene, dielectric_scene
from raytracing_amd.renderer import Renderer, RtxError

pytestmark = pytest.mark.gpu


def _both(r, sc, torch):
    n = sc.width * sc.height * 3
    r.set_scene(sc)
    stream = torch.cuda.current_stream().cuda_stream
    # poisoned outputs, allocated before the call under test: it writes, it does not add
    want = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
    got = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
    r.render_samples(0, sc.n_samples, want.data_ptr(), stream)
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    r.render_samples_static(0, sc.n_samples, got.data_ptr(), stream)
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
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
