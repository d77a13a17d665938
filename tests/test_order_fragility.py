"""How fragile are the parity claims to the operation orders RECALLED for the reference's
dependencies?  nalgebra 0.32 and rand 0.8 are not on disk (no Cargo.lock, no network), so the
oracle restates their small kernels from memory (oracle/rt_oracle.c header).  These tests swap
each recalled order for another plausible one and measure what moves (tools/order_fragility.py),
and check exhaustively that the one un-modelled branch of rand's gen_range never fires.

Reference call sites: src/objects/geometry.rs:25-31 (rotation * v, normalize, dot),
src/camera.rs:19 (Matrix3 * v), src/random.rs:34-35 and src/objects/sample.rs:30,46 (gen_range).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_alternative_orders_flip_no_primary_hit_id():
    from order_fragility import study
    res = study((1920, 1080))
    assert set(res) == {"DOT", "NORMALIZE", "QUAT", "MATVEC"}
    for name, r in res.items():
        # hit ids of the deterministic pass: none of 2 073 600 changes under any alternative
        assert r["pixels"] == 1920 * 1080 and r["ids_flipped"] == 0, (name, r)
        # hit distances: the last bits move at up to 13 % of the pixels, but beyond the 1e-5
        # relative tolerance at fewer than 1 in 10 000 (cancellation at grazing incidence)
        assert r["t_beyond_1e-5"] <= r["pixels"] // 10000, (name, r)
        assert r["t_max_rel"] < 1e-3 and r["n_max_abs"] < 1e-2, (name, r)
        # a seeded render: a path whose bounce lands on another object takes another course, so a
        # few values move a lot; under 1.5 % of them move beyond 1e-5 at all
        assert r["render_values_beyond_1e-5"] <= 0.015 * r["render_values"], (name, r)
    # the alternatives really are different arithmetic (except the camera's matrix product, whose
    # matrix in the reference scene holds only zeros and ones)
    assert res["DOT"]["t_bits_changed"] > 1000 and res["NORMALIZE"]["t_bits_changed"] > 1000
    assert res["QUAT"]["t_bits_changed"] > 1000


def test_gen_range_never_reaches_its_redraw_branch():
    """rand 0.8's gen_range(low..high) for f32 computes res = value0_1 * (high - low) + low from 23
    random bits and draws AGAIN when res >= high (rounding could reach the excluded end).  The
    oracle and the device do not model the redraw.  For every range the reference draws from --
    [0, 2 pi) and [0, 1) (src/random.rs:34-35), [0, area) and [-s, s) (src/objects/sample.rs:30,46)
    -- the largest of the 2^23 possible values stays below `high`, so the branch is dead: checked
    for the largest mantissa over a sweep of sizes, and for all 2^23 mantissas on the scene's own."""
    top = np.float32((2 ** 23 - 1) * 2.0 ** -23)
    sizes = np.concatenate([np.float32(2.0) ** np.arange(-20, 21), np.random.default_rng(1).uniform(1e-3, 1e3, 200000)]).astype(np.float32)
    for low, high in ((np.zeros_like(sizes), sizes), (-sizes, sizes)):
        res = top * (high - low) + low
        assert (res < high).all()
    allv = (np.arange(2 ** 23, dtype=np.uint32).astype(np.float32) * np.float32(2.0 ** -23))
    two_pi = np.float32(2.0) * np.float32(3.14159274101257324)
    # the reference scene's light: half-sizes (2, 0.1, 2) -> area sum ab + bc + ac
    a, b, c = np.float32(2.0), np.float32(0.1), np.float32(2.0)
    area = (a * b + b * c) + a * c
    for low, high in ((np.float32(0), two_pi), (np.float32(0), np.float32(1)), (np.float32(0), area), (-a, a), (-b, b)):
        res = allv * (high - low) + low
        assert (res < high).all() and (res >= low).all()
