"""GPU parity: the HIP path (through the C ABI of librtx.so) against the CPU oracle on
the same seeded inputs.

Bars: hit ids bit-exact; with -ffp-contract=off and correctly rounded div/sqrt on the
device, t, normals and the FORWARD-order radiance are also compared bit for bit; the
REFERENCE-order oracle (recursion + f32 running mean, as src/main.rs:13-31 and
src/trace.rs:11-71 evaluate them) is matched within 1e-5 relative (the tolerance the
task states for deterministic passes).
"""
import os

import numpy as np
import pytest

from raytracing_amd import _abi, reference_scene, dielectric_scene, synthetic_scene
from raytracing_amd.renderer import Renderer, RtxError

pytestmark = pytest.mark.gpu

ACCELS = [_abi.ACCEL_BRUTE_LDS, _abi.ACCEL_BRUTE_SCALAR]


def rel_err(a, b):
    return np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-6)


@pytest.mark.parametrize("accel", ACCELS)
def test_primary_pass_bit_exact(oracle, accel):
    sc = reference_scene(width=160, height=120)
    ids, t, n, ins = oracle.primary(sc)
    with Renderer(accel=accel) as r:
        gids, gt, gn, gins = r.primary(sc)
    assert np.array_equal(ids, gids)
    assert np.array_equal(t.view(np.uint32), gt.view(np.uint32))
    assert np.array_equal(n.view(np.uint32), gn.view(np.uint32))
    assert np.array_equal(ins, gins)
    assert set(np.unique(ids)) == {0, 1, 2, 3, 4, 5, 6, 7}


def test_primary_pass_reference_resolution(oracle):
    sc = reference_scene()  # 512 x 512, the reference's own configuration
    ids, t, n, ins = oracle.primary(sc)
    with Renderer() as r:
        gids, gt, gn, gins = r.primary(sc)
    assert np.array_equal(ids, gids)
    assert np.array_equal(t.view(np.uint32), gt.view(np.uint32))
    assert np.array_equal(n.view(np.uint32), gn.view(np.uint32))


@pytest.mark.parametrize("accel", ACCELS)
def test_render_matches_forward_oracle_bitwise(oracle, accel):
    sc = reference_scene(width=96, height=64, n_samples=8)
    want = oracle.render(sc, seed=7, order=oracle.ORDER_FORWARD)
    with Renderer(seed=7, accel=accel) as r:
        got = r.render(sc)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))


def test_render_matches_reference_order_oracle(oracle):
    sc = reference_scene(width=96, height=64, n_samples=8)
    want = oracle.render(sc, seed=3, order=oracle.ORDER_REFERENCE)
    with Renderer(seed=3) as r:
        got = r.render(sc)
    assert rel_err(want, got).max() <= 1e-5


def test_render_dielectric_metallic_depth16(oracle):
    sc = dielectric_scene(width=80, height=60, n_samples=4, ray_depth=16)
    want_f = oracle.render(sc, seed=11, order=oracle.ORDER_FORWARD)
    want_r = oracle.render(sc, seed=11, order=oracle.ORDER_REFERENCE)
    with Renderer(seed=11) as r:
        got = r.render(sc)
    assert np.array_equal(want_f.view(np.uint32), got.view(np.uint32))
    assert rel_err(want_r, got).max() <= 1e-5


def test_render_ellipsoid_light_and_background(oracle):
    sc = reference_scene(width=64, height=48, n_samples=6)
    prims = list(sc.prims)
    # swap the box light for an emissive, rotated ellipsoid and open the back wall
    from raytracing_amd import Prim, normalize_quaternion
    prims[5] = Prim(_abi.ELLIPSOID, (1.5, 0.4, 1.0), position=(0.5, 3.5, 0.5),
                    rotation=normalize_quaternion(0.2, 0.1, -0.3, 0.9), emission=(6, 5, 4), color=(0.5, 0.5, 0.5))
    del prims[1]
    sc = sc.replace(prims=prims, background=(0.1, 0.2, 0.4))
    want = oracle.render(sc, seed=5, order=oracle.ORDER_FORWARD)
    with Renderer(seed=5) as r:
        got = r.render(sc)
        assert r.lights() == oracle.lights(sc) == [4]
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))


@pytest.mark.parametrize("words", [(0, 0, 0, 0), (0x80000000, 0x40000000, 0xC0000000, 0x12345678),
                                   (0xFFFFFFFF,) * 4, (0x40000000, 0x9E3779B9, 0x7F4A7C15, 0xDEADBEEF)])
def test_zero_randomness_shading(oracle, words):
    """Every draw block returns the same four words: both MIS branches get covered."""
    sc = reference_scene(width=64, height=64, n_samples=2)
    want_f = oracle.render(sc, order=oracle.ORDER_FORWARD, rng_mode=oracle.RNG_CONST, const_words=words)
    want_r = oracle.render(sc, order=oracle.ORDER_REFERENCE, rng_mode=oracle.RNG_CONST, const_words=words)
    with Renderer(rng_mode=_abi.RNG_CONST, const_words=words) as r:
        got = r.render(sc)
    assert np.array_equal(want_f.view(np.uint32), got.view(np.uint32))
    assert rel_err(want_r, got).max() <= 1e-5


def test_event_counters_match_oracle(oracle):
    sc = dielectric_scene(width=64, height=48, n_samples=3, ray_depth=8)
    _, want = oracle.render(sc, seed=2, order=oracle.ORDER_FORWARD, want_stats=True)
    with Renderer(seed=2, collect_stats=True, accel=_abi.ACCEL_BRUTE_LDS) as r:
        r.render(sc)
        got = r.stats()
    assert got.pop("bvh_node_tests") == 0
    assert got == want


def test_light_pixel_is_exactly_emission(oracle):
    """A pixel whose primary hit is the light has radiance (2,2,2) in every sample: the
    light's albedo is 0 (assets/scene.txt:38-41), so the scattered term is exactly 0."""
    sc = reference_scene(width=128, height=128, n_samples=4)
    ids, *_ = oracle.primary(sc)
    with Renderer(seed=9) as r:
        img = r.render(sc)
    # interior of the light (all 4 jittered samples stay on it)
    from scipy.ndimage import binary_erosion
    core = binary_erosion(ids == 5, iterations=2)
    assert core.sum() > 10
    assert np.all(img[core] == 2.0)
    px = oracle.to_bytes(oracle.color_correction(img[core]))
    assert np.all(px == 245)


def test_sample_ranges_add_up(oracle):
    """Splitting the sample loop (as the multi-GPU path does) gives the same sums."""
    import ctypes
    import torch
    sc = reference_scene(width=64, height=48, n_samples=8)
    n = sc.width * sc.height * 3
    with Renderer(seed=4) as r:
        r.set_scene(sc)
        a = torch.empty(n, dtype=torch.float32, device="cuda")
        b = torch.empty(n, dtype=torch.float32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        r.render_samples(0, 3, a.data_ptr(), stream)
        r.render_samples(3, 5, b.data_ptr(), stream)
        torch.cuda.synchronize()
        sa, sb = a.cpu().numpy(), b.cpu().numpy()
    wa = oracle.render(sc, seed=4, order=oracle.ORDER_FORWARD, sample_first=0, sample_count=3, output_sum=True)
    wb = oracle.render(sc, seed=4, order=oracle.ORDER_FORWARD, sample_first=3, sample_count=5, output_sum=True)
    assert np.array_equal(wa.reshape(-1).view(np.uint32), sa.view(np.uint32))
    assert np.array_equal(wb.reshape(-1).view(np.uint32), sb.view(np.uint32))
    full = oracle.render(sc, seed=4, order=oracle.ORDER_REFERENCE).reshape(-1)
    assert rel_err(full, (sa + sb) / 8.0).max() <= 1e-5


def test_ragged_image_sizes(oracle):
    """Widths and heights that are not multiples of the 8x8 wave tile."""
    for w, h in [(1, 1), (7, 5), (9, 17), (65, 3)]:
        sc = reference_scene(width=w, height=h, n_samples=2)
        want = oracle.render(sc, seed=1, order=oracle.ORDER_FORWARD)
        with Renderer(seed=1) as r:
            got = r.render(sc)
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), (w, h)


def test_empty_scene_and_zero_samples(oracle):
    sc = reference_scene(width=16, height=16, n_samples=2).replace(prims=[], background=(0.25, 0.5, 1.0))
    with Renderer() as r:
        img = r.render(sc)
        assert np.all(img == np.array([0.25, 0.5, 1.0], dtype=np.float32))
        ids, *_ = r.primary(sc)
        assert np.all(ids == -1)
        img0 = r.render(reference_scene(width=8, height=8, n_samples=0))
        assert np.all(img0 == 0)


@pytest.mark.parametrize("mode", [_abi.MODE_MEGAKERNEL, _abi.MODE_WAVEFRONT])
def test_ray_depth_zero_renders_black(oracle, mode):
    """RAY_DEPTH 0: the camera ray itself is cut before it intersects anything (src/trace.rs:12-14),
    so neither the background nor a light shows.  Every accel, both kernels, and a continued sum."""
    import torch
    sc = reference_scene(width=40, height=24, n_samples=3).replace(ray_depth=0, background=(0.5, 0.25, 1.0))
    want = oracle.render(sc, seed=1, order=oracle.ORDER_FORWARD)
    want_ref = oracle.render(sc, seed=1, order=oracle.ORDER_REFERENCE)
    assert not want.any() and not want_ref.any()
    for accel in (_abi.ACCEL_BRUTE_LDS, _abi.ACCEL_BRUTE_SCALAR, _abi.ACCEL_BVH):
        with Renderer(seed=1, mode=mode, accel=accel, collect_stats=True) as r:
            got = r.render(sc)
            st = r.stats()
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), accel
        assert st["samples"] == 40 * 24 * 3 and st["rays"] == 0
    # a continued sum keeps its values (and -0 + 0 = +0, as the additions of zero radiance give)
    with Renderer(seed=1, mode=mode) as r:
        r.set_scene(sc)
        buf = torch.full((40 * 24 * 3,), 1.5, dtype=torch.float32, device="cuda")
        buf[0] = -0.0
        r.accumulate_samples(0, 2, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        out = buf.cpu().numpy()
        assert out[0] == 0 and not np.signbit(out[0]) and (out[1:] == 1.5).all()
        r.render_samples(0, 2, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert not buf.cpu().numpy().any()


def test_error_codes():
    from raytracing_amd import Prim
    with Renderer() as r:
        bad = reference_scene(width=0, height=4)
        with pytest.raises(RtxError) as e:
            r.render(bad)
        assert e.value.code == _abi.RTX_ERR_INVALID_SCENE
        sc = reference_scene(width=8, height=8)
        sc.prims[5] = Prim(_abi.BOX, (2, 0, 2), position=(0, 5, 0), emission=(2, 2, 2))
        with pytest.raises(RtxError) as e:
            r.render(sc)
        assert e.value.code == _abi.RTX_ERR_INVALID_SCENE
    with Renderer() as r:
        with pytest.raises(RtxError) as e:
            r.render_samples(0, 1, 1, 0)
        assert e.value.code == _abi.RTX_ERR_NO_SCENE


# Per-pixel RMSE (radiance clipped at 4, so a single firefly does not decide the figure) of an N-spp
# render against the committed 16 384-spp render of the oracle in the reference's evaluation order
# (tests/golden/converged_renders.npz, written by tests/golden/make_converged.py).  Both are
# independent Monte-Carlo estimates of the same integral, so the expected value is
# sigma * sqrt(1/N + 1/16384) with sigma the root mean per-pixel variance of one sample: measured
# 0.26 to 0.28 on the reference scene and 0.85 to 1.33 on the glass / metal scene at depth 16
# (heavy-tailed: caustic paths).  The bounds are those with a third to spare; the ratio between
# 64 and 1024 spp checks the 1 / sqrt(N) slope (ideal 3.9).
RMSE_SIGMA = {"reference": 0.36, "dielectric": 1.8}
RMSE_SLOPE = {"reference": (3.0, 5.0), "dielectric": (2.0, 5.5)}


@pytest.mark.parametrize("mode", [_abi.MODE_MEGAKERNEL, _abi.MODE_WAVEFRONT])
def test_monte_carlo_rmse_vs_converged_reference_render(mode):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_converged import converged_scenes, SPP
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "converged_renders.npz"))
    clip = lambda x: np.minimum(x, 4.0)
    for name, sc in converged_scenes():
        ref = g[name]
        rmse = {}
        with Renderer(seed=200, mode=mode) as r:              # the fixture's seed is 1000
            for n in (64, 256, 1024):
                img = r.render(sc.replace(n_samples=n))
                rmse[n] = float(np.sqrt(np.mean((clip(img) - clip(ref)) ** 2)))
                assert rmse[n] <= RMSE_SIGMA[name] * np.sqrt(1.0 / n + 1.0 / SPP), (name, n, rmse)
            assert abs(float(img.mean()) - float(ref.mean())) <= 0.02 * float(ref.mean()), name
        lo, hi = RMSE_SLOPE[name]
        assert lo < rmse[64] / rmse[1024] < hi, (name, rmse)


# ----------------------------------------------------------------- BVH (large scenes)
def test_bvh_matches_linear_scan_on_synthetic_scene(oracle):
    """Host-built BVH + traversal must return exactly what the reference's linear scan
    returns (src/trace.rs:102-121): same ids, same bits."""
    sc = synthetic_scene(n_prims=1500, seed=7, width=96, height=64, n_samples=2, ray_depth=4)
    ids, t, n, ins = oracle.primary(sc)
    with Renderer(accel=_abi.ACCEL_BVH) as r:
        gids, gt, gn, gins = r.primary(sc)
    assert np.array_equal(ids, gids)
    assert np.array_equal(t.view(np.uint32), gt.view(np.uint32))
    assert np.array_equal(n.view(np.uint32), gn.view(np.uint32))
    assert len(np.unique(ids)) > 50
    want = oracle.render(sc, seed=21, order=oracle.ORDER_FORWARD)
    with Renderer(seed=21, accel=_abi.ACCEL_BVH) as r:
        got = r.render(sc)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
    with Renderer(seed=21, accel=_abi.ACCEL_BRUTE_SCALAR) as r:
        got2 = r.render(sc)
    assert np.array_equal(want.view(np.uint32), got2.view(np.uint32))


def test_bvh_on_the_reference_scene_and_auto_selection(oracle):
    sc = reference_scene(width=64, height=64, n_samples=4)
    want = oracle.render(sc, seed=8, order=oracle.ORDER_FORWARD)
    with Renderer(seed=8, accel=_abi.ACCEL_BVH) as r:
        got = r.render(sc)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
    # AUTO: the scan for a handful of ellipsoids and boxes (the reference scene has 3), else BVH
    with Renderer(seed=8, collect_stats=True) as r:
        r.render(sc)
        assert r.stats()["bvh_node_tests"] == 0
        r.render(dielectric_scene(width=32, height=32, n_samples=1, ray_depth=4))
        assert r.stats()["bvh_node_tests"] > 0
    big = synthetic_scene(n_prims=600, seed=3, width=32, height=32, n_samples=1, ray_depth=3)
    want = oracle.render(big, seed=8, order=oracle.ORDER_FORWARD)
    with Renderer(seed=8, accel=_abi.ACCEL_AUTO, collect_stats=True) as r:
        got = r.render(big)
        st = r.stats()
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
    assert st["bvh_node_tests"] > 0 and sum(st["prim_tests"]) < st["rays"] * 600 / 4
    with Renderer(accel=_abi.ACCEL_BRUTE_LDS) as r:
        with pytest.raises(RtxError) as e:
            r.render(big)
        assert e.value.code == _abi.RTX_ERR_UNSUPPORTED


def test_two_pixels_per_lane_kernel_variants(oracle):
    """Small all-diffuse scenes whose lights are black run the scan kernel that keeps two pixels
    per lane.  Its variants -- few and many lights, LDS and scalar table, event counters, a
    continued sum, a ragged image one tile wide -- against the oracle, bits and counters."""
    import torch
    from raytracing_amd import Prim, normalize_quaternion
    base = reference_scene(width=77, height=43, n_samples=5)
    many = base.replace(prims=base.prims + [
        Prim(_abi.BOX, (0.5, 0.1, 0.5), position=(-3, 4.5, 1), emission=(3, 2, 1)),
        Prim(_abi.ELLIPSOID, (0.6, 0.3, 0.5), position=(3, 3.5, 1), rotation=normalize_quaternion(0.3, 0.1, 0.2, 0.9),
             emission=(1, 2, 3)),
        Prim(_abi.BOX, (0.4, 0.4, 0.4), position=(0, -4, 3), rotation=normalize_quaternion(0.1, 0.5, 0.0, 0.8),
             emission=(2, 2, 2))])
    for sc in (base, many, base.replace(width=5, height=19)):
        want, wstats = oracle.render(sc, seed=14, order=oracle.ORDER_FORWARD, want_stats=True)
        for accel in (_abi.ACCEL_BRUTE_LDS, _abi.ACCEL_BRUTE_SCALAR):
            with Renderer(seed=14, accel=accel) as r:
                got = r.render(sc)
            assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), (len(sc.prims), accel)
        with Renderer(seed=14, accel=_abi.ACCEL_BRUTE_LDS, collect_stats=True) as r:
            got = r.render(sc)
            st = r.stats()
        st.pop("bvh_node_tests")
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32)) and st == wstats
    # sums continued over two calls, the second starting in the middle of the sample range
    n = many.width * many.height * 3
    with Renderer(seed=14) as r:
        r.set_scene(many)
        buf = torch.empty(n, dtype=torch.float32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        r.render_samples(0, 2, buf.data_ptr(), stream)
        r.accumulate_samples(2, 3, buf.data_ptr(), stream)
        torch.cuda.synchronize()
        got = buf.cpu().numpy().reshape(many.height, many.width, 3)
    want = oracle.render(many, seed=14, order=oracle.ORDER_FORWARD, output_sum=True)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))


def test_linear_scan_with_small_and_large_object_tables(oracle):
    """The linear scan with object tables of 100 and 400 entries, staged in LDS or read through
    scalar loads (the 400-entry table fills the workgroup's LDS share: no hand-over pool), and
    with the cost-ordered dispatch on top (64 spp, 8192+ tiles), against the oracle and the BVH."""
    for n_prims, w, h, spp in ((100, 57, 33, 3), (400, 57, 33, 2), (100, 1021, 771, 64)):
        sc = synthetic_scene(n_prims=n_prims, seed=17, width=w, height=h, n_samples=spp, ray_depth=4, n_lights=2)
        if spp < 10:
            want = oracle.render(sc, seed=9, order=oracle.ORDER_FORWARD)
        for accel in (_abi.ACCEL_BRUTE_LDS, _abi.ACCEL_BRUTE_SCALAR):
            with Renderer(seed=9, accel=accel) as r:
                got = r.render(sc)
            if spp < 10:
                assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), (n_prims, accel)
            elif accel == _abi.ACCEL_BRUTE_LDS:
                first = got
            else:
                assert np.array_equal(first.view(np.uint32), got.view(np.uint32))
    # the large frame against the BVH kernels (one pixel per lane, another launch shape)
    with Renderer(seed=9, accel=_abi.ACCEL_BVH) as r:
        assert np.array_equal(first.view(np.uint32), r.render(sc).view(np.uint32))


def test_bvh_ties_go_to_the_lowest_index(oracle):
    """Coincident objects: the scan keeps the first minimum; the BVH must too."""
    from raytracing_amd import Prim
    sc = reference_scene(width=48, height=48, n_samples=1)
    dup = [Prim(_abi.ELLIPSOID, (1.5, 1.5, 1.5), position=(0, 0, 0), color=(0.1 * k, 0.5, 0.5)) for k in range(7)]
    boxes = [Prim(_abi.BOX, (1, 1, 1), position=(-3, -3, 1), color=(0.5, 0.1 * k, 0.5)) for k in range(6)]
    sc = sc.replace(prims=sc.prims[:5] + boxes[:3] + dup + boxes[3:])
    ids, t, n, ins = oracle.primary(sc)
    with Renderer(accel=_abi.ACCEL_BVH) as r:
        gids, gt, gn, gins = r.primary(sc)
    assert np.array_equal(ids, gids)
    assert 8 in ids and 5 in ids and not (set(range(9, 15)) & set(np.unique(ids)))


def test_scan_ties_across_object_categories_go_to_the_lowest_index(oracle):
    """The linear scan visits the objects by category, not in object order.  A unit sphere and a
    unit cube at the same place are hit at exactly the same distance by rays along the axes, in
    different categories; coincident planes, some on the axis-plane path and some not, tie as
    well.  The lower index must win whichever category is visited first."""
    from raytracing_amd import Prim, normalize_quaternion
    sphere = Prim(_abi.ELLIPSOID, (1, 1, 1), position=(0, 0, 0), color=(0.9, 0.2, 0.2))
    cube = Prim(_abi.BOX, (1, 1, 1), position=(0, 0, 0), color=(0.2, 0.9, 0.2))
    sphere3 = Prim(_abi.ELLIPSOID, (3, 3, 3), position=(0, 0, 0), color=(0.9, 0.2, 0.2))   # not a power of two
    cube3 = Prim(_abi.BOX, (3, 3, 3), position=(0, 0, 0), color=(0.2, 0.9, 0.2))
    floor_axis = Prim(_abi.PLANE, (0, 1, 0), position=(0, -4, 0), color=(0.5, 0.5, 0.5))
    floor_scaled = Prim(_abi.PLANE, (0, 2, 0), position=(0, -4, 0), color=(0.6, 0.6, 0.6))   # same plane, general path
    half_turn = normalize_quaternion(0.0, 1.0, 0.0, 0.0)                                    # about y: the plane maps onto itself
    floor_turned = Prim(_abi.PLANE, (0, 1, 0), position=(0, -4, 0), rotation=half_turn, color=(0.7, 0.7, 0.7))
    base = reference_scene(width=9, height=9, n_samples=2).replace(
        cam_pos=(0.0, 0.0, 8.0), cam_right=(1.0, 0.0, 0.0), cam_up=(0.0, 1.0, 0.0), cam_forward=(0.0, 0.0, -1.0))
    for prims in ([sphere, cube, floor_scaled, floor_axis, floor_turned], [cube, sphere, floor_turned, floor_axis, floor_scaled],
                  [floor_axis, sphere3, cube3, floor_scaled], [cube3, floor_turned, sphere3, floor_axis]):
        sc = base.replace(prims=list(prims))
        ids, t, n, ins = oracle.primary(sc)
        want = oracle.render(sc, seed=3, order=oracle.ORDER_FORWARD)
        for mode, accel in ((_abi.MODE_MEGAKERNEL, _abi.ACCEL_BRUTE_LDS), (_abi.MODE_MEGAKERNEL, _abi.ACCEL_BRUTE_SCALAR),
                            (_abi.MODE_WAVEFRONT, _abi.ACCEL_BRUTE_LDS), (_abi.MODE_MEGAKERNEL, _abi.ACCEL_BVH)):
            with Renderer(seed=3, mode=mode, accel=accel) as r:
                gids, gt, gn, gins = r.primary(sc)
                got = r.render(sc)
            assert np.array_equal(ids, gids), (mode, accel)
            assert np.array_equal(t.view(np.uint32), gt.view(np.uint32))
            assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
        # the centre pixel's ray runs along -z: the unit sphere and the unit cube tie there exactly,
        # and the first of them is kept (with radius 3 the quotients round, so no exact tie is claimed)
        if sphere in prims:
            assert ids[4, 4] == min(i for i, p in enumerate(prims) if p.kind != _abi.PLANE)


def test_bvh_event_counters(oracle):
    sc = synthetic_scene(n_prims=800, seed=5, width=48, height=32, n_samples=2, ray_depth=4)
    _, want = oracle.render(sc, seed=2, order=oracle.ORDER_FORWARD, want_stats=True)
    with Renderer(seed=2, accel=_abi.ACCEL_BVH, collect_stats=True) as r:
        r.render(sc)
        got = r.stats()
    for key in want:
        if key != "prim_tests":
            assert got[key] == want[key], key
    assert sum(got["prim_tests"]) < sum(want["prim_tests"]) / 5


# ------------------------------------------------------------------ wavefront variant
@pytest.mark.parametrize("accel", [_abi.ACCEL_BRUTE_LDS, _abi.ACCEL_BRUTE_SCALAR, _abi.ACCEL_BVH])
def test_wavefront_matches_oracle_bitwise(oracle, accel):
    """Staged kernels + wave64 queue compaction give the same bits as the megakernel:
    every pixel still accumulates its samples in order."""
    sc = reference_scene(width=96, height=64, n_samples=8)
    want = oracle.render(sc, seed=7, order=oracle.ORDER_FORWARD)
    with Renderer(seed=7, accel=accel, mode=_abi.MODE_WAVEFRONT) as r:
        got = r.render(sc)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))


def test_wavefront_dielectric_scene_and_counters(oracle):
    sc = dielectric_scene(width=72, height=40, n_samples=3, ray_depth=16)
    want, wstats = oracle.render(sc, seed=13, order=oracle.ORDER_FORWARD, want_stats=True)
    for accel in (_abi.ACCEL_BRUTE_LDS, _abi.ACCEL_AUTO):       # AUTO traverses the BVH here
        with Renderer(seed=13, mode=_abi.MODE_WAVEFRONT, collect_stats=True, accel=accel) as r:
            got = r.render(sc)
            gstats = r.stats()
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
        nodes = gstats.pop("bvh_node_tests")
        if accel == _abi.ACCEL_AUTO:
            # the traversal tests fewer objects than the scan; every other event is the same
            assert nodes > 0 and sum(gstats["prim_tests"]) < sum(wstats["prim_tests"])
            gstats["prim_tests"] = wstats["prim_tests"]
        assert gstats == wstats


def test_wavefront_ragged_sizes_ranges_and_empty(oracle):
    import torch
    for w, h, spp in [(1, 1, 3), (13, 7, 2), (70, 3, 1)]:
        sc = reference_scene(width=w, height=h, n_samples=spp)
        want = oracle.render(sc, seed=1, order=oracle.ORDER_FORWARD)
        with Renderer(seed=1, mode=_abi.MODE_WAVEFRONT) as r:
            got = r.render(sc)
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), (w, h)
    sc = reference_scene(width=40, height=24, n_samples=6)
    with Renderer(seed=4, mode=_abi.MODE_WAVEFRONT) as r:
        r.set_scene(sc)
        a = torch.empty(sc.width * sc.height * 3, dtype=torch.float32, device="cuda")
        r.render_samples(2, 3, a.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        part = a.cpu().numpy()
        r.render_samples(0, 0, a.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.all(a.cpu().numpy() == 0)
    want = oracle.render(sc, seed=4, order=oracle.ORDER_FORWARD, sample_first=2, sample_count=3, output_sum=True)
    assert np.array_equal(want.reshape(-1).view(np.uint32), part.view(np.uint32))
    empty = reference_scene(width=16, height=16, n_samples=2).replace(prims=[], background=(0.5, 0.25, 1.0))
    with Renderer(mode=_abi.MODE_WAVEFRONT) as r:
        assert np.all(r.render(empty) == np.array([0.5, 0.25, 1.0], dtype=np.float32))


def test_bvh_depth_is_bounded_on_adversarial_input(oracle):
    """Geometrically spaced objects make the surface-area heuristic peel off one object per
    level; the builder must fall back to balanced splits so the traversal stack never
    overflows, and the result must still equal the linear scan."""
    from raytracing_amd import Prim
    sc = reference_scene(width=64, height=48, n_samples=1)
    objs = [Prim(_abi.ELLIPSOID, (0.02 * 1.25 ** k,) * 3, position=(-4 + 0.03 * 1.27 ** k, -4.5 + 0.02 * 1.25 ** k, 0.0),
                 color=(0.3, 0.6, 0.9)) for k in range(28)]
    objs += [Prim(_abi.BOX, (0.01,) * 3, position=(3.0 + 1e-3 * k, 3.0, 1.0), color=(0.5, 0.5, 0.5)) for k in range(600)]
    sc = sc.replace(prims=sc.prims[:6] + objs)
    ids, t, n, ins = oracle.primary(sc)
    with Renderer(accel=_abi.ACCEL_BVH) as r:
        gids, gt, gn, gins = r.primary(sc)
    assert np.array_equal(ids, gids)
    assert np.array_equal(t.view(np.uint32), gt.view(np.uint32))
    want = oracle.render(sc, seed=2, order=oracle.ORDER_FORWARD)
    with Renderer(seed=2, accel=_abi.ACCEL_BVH) as r:
        got = r.render(sc)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))


# ----------------------------------------------------- guards of the exact shortcuts
def test_rays_with_negative_zero_or_axis_aligned_components(oracle):
    """The identity-rotation shortcut only applies to rays without a -0 component; axis
    aligned cameras, -0 camera coordinates and odd image sizes (a pixel-centre ray with an
    exactly zero component) must give the oracle's bits either way."""
    from raytracing_amd import Prim
    for cam in [(-0.0, 0.0, 15.0), (0.0, -0.0, 15.0), (0.0, 0.0, 15.0)]:
        for (w, h) in [(33, 17), (65, 65)]:
            sc = reference_scene(width=w, height=h, n_samples=3).replace(cam_pos=cam)
            ids, t, n, ins = oracle.primary(sc)
            with Renderer() as r:
                gids, gt, gn, gins = r.primary(sc)
                got = r.render(sc)
            assert np.array_equal(ids, gids)
            assert np.array_equal(t.view(np.uint32), gt.view(np.uint32))
            assert np.array_equal(n.view(np.uint32), gn.view(np.uint32))
            want = oracle.render(sc, seed=1, order=oracle.ORDER_FORWARD)
            assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
    # a camera looking straight down an axis at an axis-aligned box light and a huge scene offset
    sc = reference_scene(width=31, height=31, n_samples=2).replace(
        cam_pos=(0.0, 0.0, 0.0), cam_forward=(0.0, 1.0, 0.0), cam_up=(0.0, 0.0, 1.0))
    sc.prims.append(Prim(_abi.BOX, (0.5, 0.5, 0.5), position=(1e30, 0.0, 0.0), emission=(1, 1, 1)))
    want = oracle.render(sc, seed=4, order=oracle.ORDER_FORWARD)
    with Renderer(seed=4) as r:
        got = r.render(sc)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))


def test_axis_planes_through_the_ray_origin_or_parallel_to_the_ray(oracle):
    """Planes whose normal is a coordinate axis take one subtraction and one division instead of
    two dot products.  Origins exactly on such a plane (distance +-0), rays exactly parallel to it
    (distance inf or NaN) and normals of either sign must leave the oracle's bits, for random and
    for constant draw words (the latter put a whole pixel column on d.x = 0)."""
    from raytracing_amd import Prim
    # the two planes of an axis that face each other are tested with one division when the origin lies
    # strictly between them: origins on one of them, between them and outside both (the last two cameras)
    cams = [(0.0, -5.0, 15.0), (5.0, 0.0, 15.0), (-5.0, 5.0, 4.0), (0.0, 0.0, -5.0), (0.0, 0.0, 15.0), (0.0, 7.0, 15.0),
            (-8.0, -6.5, 3.0)]
    for cam in cams:
        for words in (None, (0, 0, 0, 0)):
            sc = reference_scene(width=34, height=18, n_samples=3).replace(cam_pos=cam)
            # a second plane through the origin with the opposite normal, and a non-unit axis plane
            sc.prims.append(Prim(_abi.PLANE, (0.0, 0.0, -1.0), position=(0.0, 0.0, cam[2]), color=(0.5, 0.6, 0.7)))
            sc.prims.append(Prim(_abi.PLANE, (0.0, -2.0, 0.0), position=(0.0, 4.5, 0.0), color=(0.7, 0.6, 0.5)))
            kw = dict(seed=9)
            if words is not None:
                kw.update(rng_mode=_abi.RNG_CONST, const_words=words)
            want = oracle.render(sc, order=oracle.ORDER_FORWARD, **kw)
            for mode, accel in ((_abi.MODE_MEGAKERNEL, _abi.ACCEL_BRUTE_LDS), (_abi.MODE_MEGAKERNEL, _abi.ACCEL_BVH),
                                (_abi.MODE_WAVEFRONT, _abi.ACCEL_BRUTE_SCALAR)):
                with Renderer(mode=mode, accel=accel, **kw) as r:
                    got = r.render(sc)
                assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), (cam, words, mode, accel)


def test_many_lights_of_both_kinds(oracle):
    """More than two lights switch the light pdf to the candidate-mask form; it must add the
    same terms in the same order (boxes and ellipsoids, rotated, overlapping, one far away)."""
    from raytracing_amd import Prim, normalize_quaternion
    sc = reference_scene(width=72, height=56, n_samples=4)
    lights = [
        Prim(_abi.BOX, (0.6, 0.1, 0.6), position=(-3, 4.5, 0), emission=(3, 2, 1), color=(0.2, 0.2, 0.2)),
        Prim(_abi.ELLIPSOID, (0.7, 0.3, 0.5), position=(3, 3.5, 1), rotation=normalize_quaternion(0.3, 0.1, 0.2, 0.9),
             emission=(1, 2, 3), color=(0.5, 0.5, 0.5)),
        Prim(_abi.BOX, (0.4, 0.4, 0.4), position=(0, -4, 3), rotation=normalize_quaternion(0.1, 0.5, 0.0, 0.8),
             emission=(2, 2, 2)),
        Prim(_abi.ELLIPSOID, (0.5, 0.5, 0.5), position=(0.2, -4, 3.1), emission=(0.5, 0.5, 4)),   # overlaps the box
        Prim(_abi.BOX, (1, 1, 1), position=(0, 0, 400), emission=(9, 9, 9)),                        # behind the camera
    ]
    sc = sc.replace(prims=sc.prims + lights)
    want, wstats = oracle.render(sc, seed=6, order=oracle.ORDER_FORWARD, want_stats=True)
    for mode in (_abi.MODE_MEGAKERNEL, _abi.MODE_WAVEFRONT):
        with Renderer(seed=6, mode=mode, collect_stats=True, accel=_abi.ACCEL_BRUTE_LDS) as r:
            got = r.render(sc)
            gstats = r.stats()
            assert r.lights() == [5, 8, 9, 10, 11, 12]
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
        gstats.pop("bvh_node_tests")
        assert gstats == wstats


# ------------------------------------------------- progressive rendering, device tone map
@pytest.mark.parametrize("mode", [_abi.MODE_MEGAKERNEL, _abi.MODE_WAVEFRONT])
def test_progressive_steps_equal_the_one_shot_render(mode):
    """Sums continue across calls: any cut of the sample range, with a save / restore of the
    state in the middle, leaves the bits of the single call."""
    sc = reference_scene(width=72, height=40, n_samples=13)
    with Renderer(seed=8, mode=mode) as r:
        want = r.render(sc)
        r.progress_begin(sc)
        assert r.progress_read("mean")[1] == 0 and not r.progress_read("mean")[0].any()
        for n in (1, 5, 0, 2):
            r.progress_step(n)
        state, done = r.progress_read("sum")
        assert done == 8
        part = r.render(sc.replace(n_samples=8))          # ends the progressive render
        with pytest.raises(RtxError):
            r.progress_step(1)
    assert np.array_equal((state / np.float32(8)).view(np.uint32), part.view(np.uint32))
    with Renderer(seed=8, mode=mode) as r:                 # a new context, as after a restart
        r.progress_begin(sc, sum=state, samples_done=8)
        r.progress_step(5)
        got, done = r.progress_read("mean")
        assert done == 13
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
        with pytest.raises(RtxError):
            r.progress_begin(sc, sum=None, samples_done=3)


def test_packed_division_has_the_bits_of_the_builtin_one():
    """rtx_math.hpp computes two quotients at a time with packed FMA steps; the checker compares
    3.2e9 random operand pairs and all pairs of a table of special values with `/` on the GPU."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracing_amd", "check_div2")
    assert os.path.exists(exe), "raytracing_amd/check_div2 is not built (make -C raytracing_amd/csrc)"
    res = subprocess.run([exe], capture_output=True, text=True, timeout=100)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.startswith("0 quotients differ"), res.stdout


@pytest.mark.parametrize("accel", [_abi.ACCEL_BRUTE_LDS, _abi.ACCEL_BRUTE_SCALAR, _abi.ACCEL_BVH])
def test_sample_hand_over_between_lanes_changes_no_bits(oracle, accel, monkeypatch):
    """Lanes of the megakernel that run out of samples take over the last samples of slower
    lanes through a pool in LDS; the owner adds them in index order.  With the pool switched
    off, nearly empty (it runs out at once) and at its default size the image has the bits of
    the oracle; lanes outside a ragged image take part too."""
    sc = dielectric_scene(width=67, height=45, n_samples=37, ray_depth=12)
    want = oracle.render(sc, seed=31, order=oracle.ORDER_FORWARD)
    for entries in ("0", "1", "7", None):
        if entries is None:
            monkeypatch.delenv("RTX_POOL_ENTRIES", raising=False)
        else:
            monkeypatch.setenv("RTX_POOL_ENTRIES", entries)
        with Renderer(seed=31, accel=accel) as r:
            got = r.render(sc)
            part = r.render(sc.replace(n_samples=2))       # fewer samples than lanes can share out
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), entries
        assert np.array_equal(oracle.render(sc.replace(n_samples=2), seed=31, order=oracle.ORDER_FORWARD).view(np.uint32),
                              part.view(np.uint32)), entries


@pytest.mark.parametrize("accel", [_abi.ACCEL_BRUTE_LDS, _abi.ACCEL_BVH])
def test_cost_ordered_dispatch_changes_no_bits(accel):
    """From 64 samples and 8192 tiles on the megakernel renders 8 samples first, sorts the tiles
    by what they cost and runs the rest longest tile first.  Two steps of 32 samples stay below
    the threshold and run in image order, a counting context always does: the same bits."""
    sc = reference_scene(width=1021, height=771, n_samples=64)      # 128 x 97 tiles, ragged edges
    with Renderer(seed=21, accel=accel) as r:
        ordered = r.render(sc)
        r.progress_begin(sc)
        r.progress_step(32)
        r.progress_step(32)
        plain, done = r.progress_read("mean")
        assert done == 64
        r.progress_begin(sc)                 # and continued on top of an existing sum
        r.progress_step(24)
        r.progress_step(64)
        state, done = r.progress_read("sum")
        assert done == 88
    assert np.array_equal(ordered.view(np.uint32), plain.view(np.uint32))
    with Renderer(seed=21, accel=accel, collect_stats=True) as r:
        counted = r.render(sc)
        assert r.stats()["samples"] == 1021 * 771 * 64
        want = r.render(sc.replace(n_samples=88))
    assert np.array_equal(ordered.view(np.uint32), counted.view(np.uint32))
    assert np.array_equal((state / np.float32(88)).view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("which", ["two_pixel_scan", "one_pixel_scan", "bvh", "bvh_many_lights"])
def test_work_queue_of_sample_chunks_changes_no_bits(which):
    """From 8192 tiles on, a render of enough samples is cut into chunks that waves draw from a queue in
    device memory (a unit's chunk k + 1 waits for its chunk k and continues its sums).  A counting context
    always renders in one plain launch, every sample of a unit in one wave: the same bits, for every kernel
    that takes the queue, twice in a row on one context (the queue's words are reused), and continued on
    top of an existing sum."""
    import dataclasses
    from raytracing_amd import synthetic_scene
    sc = reference_scene(width=1021, height=771, n_samples=200)      # 128 x 97 tiles, ragged edges
    accel = _abi.ACCEL_AUTO
    if which == "one_pixel_scan":       # a coloured light: paths do not end there, the one-pixel kernel is chosen
        prims = list(sc.prims)
        prims[5] = dataclasses.replace(prims[5], color=(0.7, 0.7, 0.7))
        sc = sc.replace(prims=prims)
    elif which == "bvh":
        sc = dielectric_scene(width=1021, height=771, n_samples=150, ray_depth=8)
    elif which == "bvh_many_lights":
        sc = synthetic_scene(n_prims=300, seed=5, width=1021, height=771, n_samples=136, ray_depth=4)
    with Renderer(seed=33, accel=accel) as r:
        first = r.render(sc)
        again = r.render(sc)
        r.progress_begin(sc)
        r.progress_step(24)
        r.progress_step(sc.n_samples - 24)     # the queue on top of an existing sum
        continued, done = r.progress_read("mean")
        assert done == sc.n_samples
    with Renderer(seed=33, accel=accel, collect_stats=True) as r:
        plain = r.render(sc)
        assert r.stats()["samples"] == sc.width * sc.height * sc.n_samples
    assert np.isfinite(plain).all() and plain.mean() > 0.005
    assert np.array_equal(first.view(np.uint32), plain.view(np.uint32))
    assert np.array_equal(again.view(np.uint32), plain.view(np.uint32))
    assert np.array_equal(continued.view(np.uint32), plain.view(np.uint32))


@pytest.mark.parametrize("which", ["one_pixel_scan", "bvh", "bvh_many_lights"])
def test_a_scene_is_measured_once_and_its_order_reused_without_changing_a_bit(which):
    """The first render of enough samples of a scene measures what each tile costs (a probe launch) and dispatches
    the rest longest first; later renders on the context skip the probe and go out in that order, until the scene
    is set again.  Three steps of a progressive render (the second measures, the third reuses, with and without the
    work queue), then two whole renders through rtx_render_samples (measure, reuse) and one after the share was
    changed and changed back (measured again): the bits of a counting context's single plain launch."""
    import dataclasses
    import torch
    from raytracing_amd import synthetic_scene
    sc = reference_scene(width=1021, height=771, n_samples=214)
    if which == "one_pixel_scan":
        prims = list(sc.prims)
        prims[5] = dataclasses.replace(prims[5], color=(0.7, 0.7, 0.7))
        sc = sc.replace(prims=prims)
    elif which == "bvh":
        sc = dielectric_scene(width=1021, height=771, n_samples=214, ray_depth=8)
    else:
        sc = synthetic_scene(n_prims=300, seed=5, width=1021, height=771, n_samples=214, ray_depth=4)
    n = sc.width * sc.height * 3
    with Renderer(seed=21) as r:
        r.progress_begin(sc)
        r.progress_step(10)
        r.progress_step(70)      # measures
        r.progress_step(70)      # reuses, one launch
        r.progress_step(64)      # reuses
        steps, done = r.progress_read("sum")
        assert done == 214
        buf = torch.empty(n, dtype=torch.float32, device="cuda")
        whole = []
        for k in range(3):
            if k == 2:
                r.set_share(0, 2)
                r.set_share(0, 1)
            r.render_samples(0, 214, buf.data_ptr())
            torch.cuda.synchronize()
            whole.append(buf.cpu().numpy().reshape(steps.shape))
    with Renderer(seed=21, collect_stats=True) as r:
        r.progress_begin(sc)
        r.progress_step(214)
        plain, _ = r.progress_read("sum")
    assert np.isfinite(plain).all() and plain.mean() > 0.5
    assert np.array_equal(steps.view(np.uint32), plain.view(np.uint32))
    for k in range(3):
        assert np.array_equal(whole[k].view(np.uint32), plain.view(np.uint32)), k


def test_device_tonemap_matches_the_host_output_stage():
    from raytracing_amd import host
    sc = reference_scene(width=160, height=120, n_samples=6)
    with Renderer(seed=2) as r:
        linear = r.render(sc)
        got = r.render_bytes(sc)
        r.progress_begin(sc)
        r.progress_step(6)
        again, _ = r.progress_read("bytes")
    want = host.to_bytes(host.color_correction(linear))
    assert got.shape == want.shape == (120, 160, 3) and np.array_equal(got, again)
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff != 0).mean() < 1e-4
    assert (got[want == 245] == 245).all() and (want == 245).any()      # the light's pixels
    # special values through the device buffers: NaN -> 0, inf -> NaN (inf / inf) -> 0, negatives
    import torch
    x = torch.tensor([0.0, 2.0, 4.0, float("inf"), float("nan"), -0.5, 1e-8, 1e6], dtype=torch.float32, device="cuda")
    out = torch.empty(x.numel(), dtype=torch.uint8, device="cuda")
    with Renderer() as r:
        r.tonemap_bytes(x.data_ptr(), x.numel(), 2.0, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    want = host.to_bytes(host.color_correction((x.cpu().numpy() / np.float32(2.0))))
    assert np.abs(out.cpu().numpy().astype(int) - want.astype(int)).max() <= 1
    assert out.cpu().numpy()[[0, 3, 4]].tolist() == want[[0, 3, 4]].tolist()


def test_bvh_node_grid_extremes(oracle):
    """Node boxes live on a 16-bit grid spanning all bounded objects.  A far-away outlier
    coarsens the grid, a camera far outside it takes the always-descend path, axis-parallel
    rays make the slab arithmetic produce NaNs: each must still give the scan's bits."""
    from raytracing_amd import Prim
    base = synthetic_scene(n_prims=200, seed=11, width=48, height=32, n_samples=2, ray_depth=4)
    cases = {
        "outlier": base.replace(prims=base.prims + [Prim(_abi.ELLIPSOID, (3, 2, 1), position=(4e6, -2e6, 1e7),
                                                         color=(0.5, 0.5, 0.5))]),
        "far camera": base.replace(cam_pos=(0.0, 0.0, 3.0e8), fov_x=4e-7),
        "axis parallel": base.replace(width=33, height=33, cam_pos=(0.0, 0.0, 40.0), cam_forward=(0.0, 0.0, -1.0)),
        "flat": base.replace(prims=base.prims[:6] + [Prim(_abi.BOX, (1, 1, 0.0), position=(x, y, 0.0), color=(1, 1, 1))
                                                     for x in range(-4, 5, 2) for y in range(-4, 5, 2)]),
    }
    for name, sc in cases.items():
        with Renderer(seed=5, accel=_abi.ACCEL_BRUTE_SCALAR) as r:
            ids, t, n, ins = r.primary(sc)
            want = r.render(sc)
        for mode in (_abi.MODE_MEGAKERNEL, _abi.MODE_WAVEFRONT):
            with Renderer(seed=5, accel=_abi.ACCEL_BVH, mode=mode) as r:
                gids, gt, gn, gins = r.primary(sc)
                got = r.render(sc)
            assert np.array_equal(ids, gids), name
            assert np.array_equal(t.view(np.uint32), gt.view(np.uint32)), name
            assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), name
        if name != "far camera":
            assert (ids >= 6).any(), name          # the bounded objects are seen
    oids, ot, *_ = oracle.primary(cases["far camera"])
    with Renderer(accel=_abi.ACCEL_BVH) as r:
        gids, gt, *_ = r.primary(cases["far camera"])
    assert np.array_equal(oids, gids) and np.array_equal(ot.view(np.uint32), gt.view(np.uint32))


def test_device_equals_the_committed_golden_fixtures():
    """The device against the fixtures directly (no oracle in the loop): hit ids, distances
    and seeded renders of three scenes bit for bit, for every kernel variant."""
    import os
    from test_oracle_kat import golden_scenes
    here = os.path.dirname(__file__)
    g = np.load(os.path.join(here, "golden", "forward_order_renders.npz"))
    for name, sc, seed in golden_scenes():
        for mode in (_abi.MODE_MEGAKERNEL, _abi.MODE_WAVEFRONT):
            for accel in (_abi.ACCEL_BRUTE_LDS, _abi.ACCEL_BRUTE_SCALAR, _abi.ACCEL_BVH):
                with Renderer(seed=seed, mode=mode, accel=accel) as r:
                    ids, t, _, _ = r.primary(sc)
                    img = r.render(sc)
                assert np.array_equal(ids, g[name + "_hit_id"]), (name, mode, accel)
                assert np.array_equal(t.view(np.uint32), g[name + "_t"].view(np.uint32)), (name, mode, accel)
                assert np.array_equal(img.view(np.uint32), g[name + "_forward"].view(np.uint32)), (name, mode, accel)
    ref = np.load(os.path.join(here, "golden", "reference_scene_64x64.npz"))
    sc = reference_scene(width=64, height=64, n_samples=4)
    with Renderer(seed=1) as r:
        img = r.render(sc)
    assert rel_err(ref["radiance_seed1_4spp"], img).max() <= 1e-5      # the reference's evaluation order
    with Renderer(rng_mode=_abi.RNG_CONST, const_words=tuple(int(x) for x in ref["const_words"])) as r:
        cx = r.render(sc)
    assert rel_err(ref["radiance_const_xi"], cx).max() <= 1e-5


def test_random_scene_sweep_and_its_past_failures(oracle):
    """tools/parity_sweep.py in small: random scenes through every kernel variant, bit for bit.
    The listed seeds once failed: distant ray origins (open scenes, paths coming back from
    planes thousands of units away), where the f32 ellipsoid test reports hits for rays that
    pass well outside the object -- the many-lights skip test and the BVH boxes have to bound
    that zone, not the geometric object."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from parity_sweep import random_scene, extreme_scene
    scenes = [(seed, random_scene(seed)) for seed in [1878, 3019, 3645, 4002, 4365, 4813, 4866] + list(range(40))]
    # inputs only the C ABI lets through: quaternions that are not unit (the world -> local map is
    # then no rotation, which the bounds have to follow), zero sizes, and the other oddities
    scenes += [(seed, extreme_scene(seed, ["quat"])) for seed in (1, 9, 36, 64, 981)]
    scenes += [(858, extreme_scene(858, ["huge"]))]       # an object 1e31 away stretches the node grid: overflow in the slab products
    scenes += [(seed, extreme_scene(seed)) for seed in range(30)]
    variants = [(m, a) for m in (_abi.MODE_MEGAKERNEL, _abi.MODE_WAVEFRONT)
                for a in (_abi.ACCEL_AUTO, _abi.ACCEL_BRUTE_SCALAR, _abi.ACCEL_BVH)]
    same = lambda a, b: bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())
    for seed, sc in scenes:
        ids, t, n, ins = oracle.primary(sc)
        want = oracle.render(sc, seed=seed, order=oracle.ORDER_FORWARD)
        for mode, accel in variants:
            with Renderer(seed=seed, mode=mode, accel=accel) as r:
                gids, gt, gn, gins = r.primary(sc)
                got = r.render(sc)
            assert np.array_equal(ids, gids), (seed, mode, accel)
            assert same(t, gt) and same(n, gn), (seed, mode, accel)          # any NaN equals any NaN
            assert same(want, got), (seed, mode, accel)


def test_full_resolution_render_equals_the_oracle(oracle):
    """The benchmark's resolution, 1920x1080 (32 400 waves, every tile shape of the launch),
    at 16 of its 256 samples per pixel: all 2 073 600 pixels bit for bit.  The oracle renders
    column bands on the host's threads.  (tools/full_size_parity.py does the same at the full
    256 spp, 531 M samples, in about a minute.)"""
    import os
    sc = reference_scene(width=1920, height=1080, n_samples=16)
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    want = oracle.render(sc, seed=1, order=oracle.ORDER_FORWARD, threads=threads)
    with Renderer(seed=1) as r:
        got = r.render(sc)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
    assert 0.05 < float(got.mean()) < 0.2


def test_full_resolution_glass_and_metal_scene_equals_the_oracle(oracle):
    """The divergence-stress configuration at its full frame (1920x1080, depth 16, 33 objects, every
    material) at 2 of its 256 samples per pixel, both kernels through the BVH the library picks for
    it: all 2 073 600 pixels bit for bit, and the event counters of the megakernel."""
    sc = dielectric_scene(width=1920, height=1080, n_samples=2, ray_depth=16)
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    want = oracle.render(sc, seed=5, order=oracle.ORDER_FORWARD, threads=threads)
    for mode in (_abi.MODE_MEGAKERNEL, _abi.MODE_WAVEFRONT):
        with Renderer(seed=5, mode=mode) as r:
            got = r.render(sc)
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), mode
    assert 0.05 < float(want.mean()) < 0.5


def test_ten_thousand_objects_equal_the_linear_scan_oracle(oracle):
    """The large-scene configuration at its full object count (10 000 objects, 8 lights, the BVH
    several levels deeper than any other test's): primary ids, distances, normals and a seeded
    render of both kernels against the oracle, which scans all objects for every ray."""
    sc = synthetic_scene(10000, 12345, 96, 54, 2)
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    ids, t, n, ins = oracle.primary(sc)
    want = oracle.render(sc, seed=12, order=oracle.ORDER_FORWARD, threads=threads)
    assert len(np.unique(ids)) > 300
    for mode in (_abi.MODE_MEGAKERNEL, _abi.MODE_WAVEFRONT):
        with Renderer(seed=12, mode=mode) as r:                    # AUTO: the BVH
            gids, gt, gn, gins = r.primary(sc)
            got = r.render(sc)
        assert np.array_equal(ids, gids), mode
        assert np.array_equal(t.view(np.uint32), gt.view(np.uint32)), mode
        assert np.array_equal(n.view(np.uint32), gn.view(np.uint32)), mode
        assert np.array_equal(ins, gins), mode
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), mode


def test_4k_frame_in_steps_and_in_shares(oracle):
    """3840 x 2160, the multi-GPU configuration's frame, at 2 samples per pixel: one sample step per call, each
    equal to the oracle's sums bit for bit; then the frame dealt out to two workers inside the library and to two
    contexts with their shares set (as two processes would hold them) against the steps' sum -- two samples added
    to zero are the same two additions in either order."""
    import torch
    sc = reference_scene(width=3840, height=2160, n_samples=2)
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    n = sc.width * sc.height * 3
    parts = []
    with Renderer(seed=1) as r:
        r.set_scene(sc)
        buf = torch.empty(n, dtype=torch.float32, device="cuda")
        for first in (0, 1):
            r.render_samples(first, 1, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            parts.append(buf.cpu().numpy().reshape(sc.height, sc.width, 3))
    for first in (0, 1):
        want = oracle.render(sc, seed=1, order=oracle.ORDER_FORWARD, sample_first=first, sample_count=1, output_sum=True,
                             threads=threads)
        assert np.array_equal(want.view(np.uint32), parts[first].view(np.uint32)), first
    with Renderer(seed=1, devices=[0, 0]) as r:
        got = r.render(sc)
    total = (parts[0] + parts[1]).astype(np.float32)
    want = (total / np.float32(2.0)).astype(np.float32)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
    added = np.zeros(n, dtype=np.float32)
    for share in (0, 1):
        with Renderer(seed=1) as r:
            r.set_share(share, 2)
            r.set_scene(sc)
            r.render_samples(0, 2, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            added = (added + buf.cpu().numpy()).astype(np.float32)
    assert np.array_equal(total.reshape(-1).view(np.uint32), added.view(np.uint32))


def test_zero_throughput_rule_matches_full_paths_and_is_off_for_untame_scenes(oracle):
    """The device ends paths with zero throughput in tame scenes (same image, fewer rays) and
    keeps full paths otherwise; the oracle's forward order does the same, so counters agree."""
    from raytracing_amd import Prim
    sc = reference_scene(width=64, height=48, n_samples=4)
    full, full_stats = oracle.render(sc, seed=3, order=oracle.ORDER_FORWARD, want_stats=True, full_paths=True)
    with Renderer(seed=3, collect_stats=True, accel=_abi.ACCEL_BRUTE_LDS) as r:
        got = r.render(sc)
        st = r.stats()
    assert np.array_equal(full.view(np.uint32), got.view(np.uint32))
    assert st["rays"] < 0.8 * full_stats["rays"] and st["samples"] == full_stats["samples"]
    # an object whose colour is inf makes 0 * inf = NaN possible: full paths, NaNs where the oracle has them
    wild = sc.replace(prims=sc.prims + [Prim(_abi.ELLIPSOID, (1, 1, 1), position=(0, 0, 2), color=(np.inf, 1, 1))])
    assert not oracle.scene_is_tame(wild)
    want, wstats = oracle.render(wild, seed=3, order=oracle.ORDER_FORWARD, want_stats=True)
    with Renderer(seed=3, collect_stats=True, accel=_abi.ACCEL_BRUTE_LDS) as r:
        got = r.render(wild)
        st = r.stats()
    assert np.array_equal(np.isnan(want), np.isnan(got)) and np.isnan(want).any()
    ok = ~np.isnan(want)
    assert np.array_equal(want[ok].view(np.uint32), got[ok].view(np.uint32))
    st.pop("bvh_node_tests")
    assert st == wstats
