"""Randomised parity sweep (GPU box): many small random scenes -- mixed kinds, poses (rotated,
identity), power-of-two and ordinary sizes, 0..7 lights of both kinds, all materials, dense
and sparse, tilted planes -- rendered by every kernel variant and compared bit for bit with
the CPU oracle (hit ids, distances, normals, seeded radiance).  Prints one line per mismatch
and a summary; exit status 1 if anything differed.
    python tools/parity_sweep.py [n_scenes] [first_seed]
Environment: SWEEP_EXTREME=1 (or a comma-separated list of FEATURES) adds inputs only the C ABI
lets through; SWEEP_BIG=k multiplies image edges and sample counts by k; SWEEP_OPTIONS=1 checks
the other inputs of a render instead (64-bit seeds, sample ranges near 2^32, sums continued over
two calls, constant draw words) and the event counters; SWEEP_ROOMS=1 draws small all-diffuse
rooms with black lights (the scenes of the two-pixels-per-lane kernel); SWEEP_REUSE=1 keeps one context per
kernel variant for the whole sweep.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import pyoracle as O
from raytracing_amd import _abi, Prim, Scene, normalize_quaternion
from raytracing_amd.renderer import Renderer


def random_scene(seed):
    rng = np.random.default_rng(seed)
    f = lambda a: tuple(float(np.float32(v)) for v in a)
    room = float(rng.choice([6.0, 12.0, 40.0]))
    prims = []
    n_planes = int(rng.integers(0, 7))
    axes = [(0, 1, 0), (0, -1, 0), (1, 0, 0), (-1, 0, 0), (0, 0, 1), (0, 0, -1)]
    for k in range(n_planes):
        n = np.array(axes[k], dtype=np.float64)
        if rng.random() < 0.3:
            n = n * rng.uniform(0.5, 3.0) + rng.normal(scale=0.2, size=3)      # tilted, not unit
        rot = (0, 0, 0, 1) if rng.random() < 0.7 else normalize_quaternion(*f(rng.normal(size=4) * [0.1, 0.1, 0.1, 1]))
        prims.append(Prim(_abi.PLANE, f(n), position=f(-np.array(axes[k]) * room), rotation=rot,
                          color=f(rng.uniform(0.2, 1, 3)),
                          material=_abi.METALLIC if rng.random() < 0.15 else _abi.DIFFUSE))
    n_obj = int(rng.choice([0, 1, 3, 6, 7, 12, 40, 150]))
    n_lights = int(rng.integers(0, 8))
    for k in range(n_obj):
        kind = _abi.ELLIPSOID if rng.random() < 0.5 else _abi.BOX
        mode = rng.random()
        if mode < 0.25:
            size = f(rng.choice([0.25, 0.5, 1.0, 2.0], size=3))                 # powers of two
        elif mode < 0.35:
            size = f([rng.choice([0.5, 1.0, 2.0])] * 3)
        else:
            size = f(rng.uniform(0.2, room / 4, size=3))
        pos = f(rng.uniform(-room * 0.9, room * 0.9, size=3))
        if rng.random() < 0.1 and prims:
            pos = prims[-1].position                                            # coincident centres: ties
        q = rng.normal(size=4)
        rot = (0, 0, 0, 1) if rng.random() < 0.4 else normalize_quaternion(*f(q / np.linalg.norm(q)))
        albedo = f(rng.uniform(0.0, 1.0, size=3))
        m = rng.random()
        if k < n_lights:
            e = f(rng.uniform(0.5, 20.0, size=3) * (rng.random(3) < 0.8))
            prims.append(Prim(kind, size, position=pos, rotation=rot, color=albedo, emission=e))
        elif m < 0.6:
            prims.append(Prim(kind, size, position=pos, rotation=rot, color=albedo))
        elif m < 0.8:
            prims.append(Prim(kind, size, position=pos, rotation=rot, color=albedo, material=_abi.METALLIC))
        else:
            prims.append(Prim(kind, size, position=pos, rotation=rot, color=albedo, material=_abi.DIELECTRIC,
                              ior=float(np.float32(rng.uniform(1.0, 2.4)))))
    rng.shuffle(prims)
    w, h = int(rng.integers(1, 49)), int(rng.integers(1, 41))
    cam = f(rng.uniform(-room * 0.5, room * 0.5, size=3))
    fwd = rng.normal(size=3)
    fwd /= np.linalg.norm(fwd)
    up = np.cross(fwd, rng.normal(size=3))
    up /= np.linalg.norm(up)
    right = np.cross(fwd, up)
    if rng.random() < 0.3:
        right, up, fwd = (1, 0, 0), (0, 1, 0), (0, 0, -1)
        cam = (0.0, 0.0, room * 0.9)
    return Scene(width=w, height=h, ray_depth=int(rng.integers(1, 10)), n_samples=int(rng.integers(1, 4)),
                 background=f(rng.uniform(0, 1, 3) * (rng.random() < 0.5)), cam_pos=cam, cam_right=f(right), cam_up=f(up),
                 cam_forward=f(fwd), fov_x=float(np.float32(rng.uniform(0.3, 2.0))), prims=list(prims))


def room_scene(seed):
    """Small all-diffuse rooms whose lights are black (no COLOR, as in the reference's scene): the
    scenes the two-pixels-per-lane scan kernel is chosen for.  1 to 16 objects, 1 to 5 lights of
    both kinds, images up to 70 pixels wide (several 16 x 8 work units, ragged edges), up to 40
    samples (the hand-over pool gets used)."""
    rng = np.random.default_rng(seed + 7_000_000)
    f = lambda a: tuple(float(np.float32(v)) for v in a)
    room = float(rng.choice([5.0, 8.0]))
    axes = [(0, 1, 0), (0, -1, 0), (1, 0, 0), (-1, 0, 0), (0, 0, 1), (0, 0, -1)]
    prims = [Prim(_abi.PLANE, f(a), position=f(-np.array(a) * room), color=f(rng.uniform(0.2, 1, 3)))
             for a in axes[:int(rng.integers(3, 7))]]
    n_lights = int(rng.integers(1, 6))
    n_obj = int(rng.integers(n_lights, 17 - len(prims)))
    for k in range(n_obj):
        kind = _abi.ELLIPSOID if rng.random() < 0.5 else _abi.BOX
        size = f(rng.choice([0.25, 0.5, 1.0, 2.0], size=3)) if rng.random() < 0.3 else f(rng.uniform(0.1, room / 3, size=3))
        q = rng.normal(size=4)
        rot = (0, 0, 0, 1) if rng.random() < 0.5 else normalize_quaternion(*f(q / np.linalg.norm(q)))
        pos = f(rng.uniform(-room * 0.8, room * 0.8, size=3))
        if k < n_lights:
            prims.append(Prim(kind, size, position=pos, rotation=rot, emission=f(rng.uniform(0.5, 10.0, size=3))))
        else:
            prims.append(Prim(kind, size, position=pos, rotation=rot, color=f(rng.uniform(0.0, 1.0, size=3))))
    rng.shuffle(prims)
    return Scene(width=int(rng.integers(1, 71)), height=int(rng.integers(1, 41)), ray_depth=int(rng.integers(1, 9)),
                 n_samples=int(rng.choice([1, 2, 3, 5, 17, 40])), background=(0.0, 0.0, 0.0),
                 cam_pos=(0.0, 0.0, room * 0.9), cam_right=(1, 0, 0), cam_up=(0, 1, 0), cam_forward=(0, 0, -1),
                 fov_x=float(np.float32(rng.uniform(0.5, 1.8))), prims=list(prims))


FEATURES = ("scale", "negsize", "zerosize", "zeroplane", "farpos", "quat", "negquat", "lights", "ior", "color",
            "negzero", "huge")


def extreme_scene(seed, features=None):
    """Inputs a careful author would not write but the C ABI accepts: quaternions that are not
    unit, negative / zero / tiny / huge sizes, zero plane normals, objects a hundred thousand
    units away, up to 80 lights, refractive indices below 1, albedo above 1.  `features`
    restricts the oddities to the named ones (FEATURES)."""
    on = set(FEATURES if features is None else features)
    rng = np.random.default_rng(seed + 1_000_000)
    f = lambda a: tuple(float(np.float32(v)) for v in a)
    base = random_scene(seed)
    prims = list(base.prims)
    n_extra = int(rng.integers(1, 90))
    many_lights = "lights" in on and rng.random() < 0.3
    for k in range(n_extra):
        kind = int(rng.integers(0, 3))
        size = rng.uniform(0.05, 3.0, size=3)
        if "scale" in on:
            size = size * rng.choice([1.0, 1.0, 1.0, 1e-3, 50.0])
        if "negsize" in on and rng.random() < 0.15:
            size = size * rng.choice([-1.0, 1.0], size=3)
        if "zerosize" in on and rng.random() < 0.08:
            size[int(rng.integers(0, 3))] = 0.0
        if "zeroplane" in on and kind == _abi.PLANE and rng.random() < 0.1:
            size = np.zeros(3)
        pos = rng.uniform(-10, 10, size=3)
        if "farpos" in on:
            pos = pos * rng.choice([1.0, 1.0, 1.0, 100.0, 1e5])
        q = rng.normal(size=4)
        q = q / np.linalg.norm(q)
        if "quat" in on:
            q = q * rng.choice([1.0, 1.0, 0.5, 1.5, 1.0001])
        if "negquat" in on and rng.random() < 0.05:
            q = np.array([0.0, 0.0, 0.0, -1.0])
        if "negzero" in on and rng.random() < 0.3:       # exact zeros of either sign, axis-aligned values
            pos = np.where(rng.random(3) < 0.5, rng.choice([-0.0, 0.0], size=3), np.round(pos))
            if kind == _abi.PLANE:
                size = np.where(rng.random(3) < 0.6, rng.choice([-0.0, 0.0], size=3), rng.choice([-1.0, 1.0, 2.0], size=3))
        if "huge" in on and rng.random() < 0.05:
            pos = pos * rng.choice([1e12, 1e31, 1e-30])
        emission = (0, 0, 0)
        if many_lights or rng.random() < 0.1:
            emission = f(rng.uniform(0, 30, size=3))
        if kind == _abi.BOX and any(emission) and not (size > 0).all():
            size = np.abs(size) + 0.01                   # emissive boxes need positive sizes
        material = int(rng.integers(0, 3))
        ior = 1.5
        if "ior" in on:
            ior = float(np.float32(rng.choice([0.3, 0.9, 1.0, 1.5, 3.0])))
        prims.append(Prim(kind, f(size), position=f(pos), rotation=f(q),
                          color=f(rng.uniform(0, 1.2 if "color" in on else 1.0, size=3)),
                          emission=emission, material=material, ior=ior if material == _abi.DIELECTRIC else 1.0))
    rng.shuffle(prims)
    out = base.replace(prims=list(prims))
    if "negzero" in on and rng.random() < 0.5:
        out = out.replace(cam_pos=f(np.where(rng.random(3) < 0.5, -0.0, np.round(np.array(base.cam_pos)))),
                          cam_right=(1.0, 0.0, -0.0), cam_up=(-0.0, 1.0, 0.0), cam_forward=(0.0, -0.0, -1.0),
                          width=base.width | 1, height=base.height | 1)
    return out


def check_options(sc, seed, variants, same):
    """The other inputs of a render: 64-bit seeds, sample ranges far from zero, sums continued
    over several calls, constant draw words, event counters.  Returns the number of mismatches."""
    import torch
    rng = np.random.default_rng(seed + 7_000_000)
    seed64 = int(rng.integers(0, 2**63)) * 2 + int(rng.integers(0, 2))
    first = int(rng.choice([0, 1, 255, 2**31 - 2, 2**32 - 40]))
    count = int(rng.integers(1, 5))
    cut = int(rng.integers(0, count + 1))
    const = rng.random() < 0.25
    words = tuple(int(x) for x in rng.integers(0, 2**32, size=4))
    if const and rng.random() < 0.5:
        words = tuple(int(x) for x in rng.choice([0, 0x80000000, 0x40000000, 0xFFFFFFFF, 0xC0000000, 0x7FFFFFFF], size=4))
    kw = dict(seed=seed64, rng_mode=O.RNG_CONST if const else O.RNG_PHILOX, const_words=words)
    want, wstats = O.render(sc, order=O.ORDER_FORWARD, sample_first=first, sample_count=count, output_sum=True,
                            want_stats=True, **kw)
    bad = 0
    buf = torch.empty(sc.width * sc.height * 3, dtype=torch.float32, device="cuda")
    for mode, accel in variants:
        with Renderer(seed=seed64, mode=mode, accel=accel, rng_mode=_abi.RNG_CONST if const else _abi.RNG_PHILOX,
                      const_words=words, collect_stats=True) as r:
            r.set_scene(sc)
            r.render_samples(first, count, buf.data_ptr())
            torch.cuda.synchronize()
            whole = buf.cpu().numpy().reshape(want.shape)
            gstats = r.stats()
            r.render_samples(first, cut, buf.data_ptr())
            r.accumulate_samples(first + cut, count - cut, buf.data_ptr())
            torch.cuda.synchronize()
            parts = buf.cpu().numpy().reshape(want.shape)
        what = []
        if not same(want, whole): what.append("sum")
        if not same(want, parts): what.append(f"sum continued at {cut}")
        nodes = gstats.pop("bvh_node_tests")
        if nodes:
            gstats["prim_tests"] = wstats["prim_tests"]
        if gstats != wstats:
            what.append("counters " + ", ".join(k for k in wstats if wstats[k] != gstats[k]))
        if what:
            bad += 1
            print(f"seed {seed} mode {mode} accel {accel} ({len(sc.prims)} objects, {sc.width}x{sc.height}, depth {sc.ray_depth}, "
                  f"samples {first}+{count}, const {const}): " + ", ".join(what), flush=True)
    return bad


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    variants = [(m, a) for m in (_abi.MODE_MEGAKERNEL, _abi.MODE_WAVEFRONT)
                for a in (_abi.ACCEL_AUTO, _abi.ACCEL_BRUTE_SCALAR, _abi.ACCEL_BVH)]
    bad = 0
    make = random_scene
    if os.environ.get("SWEEP_EXTREME"):     # "1" = every oddity, or a comma-separated list of FEATURES ("" entries allowed)
        feats = None if os.environ["SWEEP_EXTREME"] == "1" else [x for x in os.environ["SWEEP_EXTREME"].split(",") if x]
        make = lambda seed: extreme_scene(seed, feats)
    if os.environ.get("SWEEP_ROOMS") == "1":
        make = room_scene
    big = int(os.environ.get("SWEEP_BIG", "1"))   # image edge and sample count multiplier
    options = os.environ.get("SWEEP_OPTIONS") == "1"
    reuse = os.environ.get("SWEEP_REUSE") == "1"
    workers = int(os.environ.get("SWEEP_WORKERS", "1"))
    kept = {}
    same = lambda a, b: bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())
    for seed in range(first, first + n):
        sc = make(seed)
        if big > 1:
            sc = sc.replace(width=sc.width * big, height=sc.height * big, n_samples=sc.n_samples * big)
        if options:
            bad += check_options(sc, seed, variants, same)
            if (seed - first) % 25 == 24:
                print(f"... {seed - first + 1} scenes", flush=True)
            continue
        try:
            ids, t, nrm, ins = O.primary(sc)
            want = O.render(sc, seed=1 if reuse else seed, order=O.ORDER_FORWARD)
        except Exception as e:  # the oracle refuses what the reference would panic on
            print(f"seed {seed}: oracle: {e}")
            continue
        for mode, accel in variants:
            if reuse:      # one context per variant for the whole sweep (seed 1): buffers and scene state are reused
                if (mode, accel) not in kept:
                    kept[(mode, accel)] = Renderer(seed=1, mode=mode, accel=accel)
                r = kept[(mode, accel)]
            elif workers > 1:   # SWEEP_WORKERS=N: the frame dealt out to N workers on this GPU, the same image expected
                r = Renderer(seed=seed, mode=mode, accel=accel, devices=[0] * workers)
            else:
                r = Renderer(seed=seed, mode=mode, accel=accel)
            try:
                gids, gt, gn, gins = r.primary(sc)
                got = r.render(sc)
            except Exception as e:
                print(f"seed {seed} mode {mode} accel {accel}: {e}")
                bad += 1
                continue
            finally:
                if not reuse:
                    r.close()
            what = []
            if not np.array_equal(ids, gids): what.append(f"ids ({int((ids != gids).sum())})")
            if not same(t, gt): what.append("t")
            if not same(nrm, gn): what.append("normal")
            if not np.array_equal(ins, gins): what.append("inside")
            if not same(want, got): what.append(f"radiance ({int(((want.view(np.uint32) != got.view(np.uint32)) & ~(np.isnan(want) & np.isnan(got))).any(axis=2).sum())} px)")
            if what:
                bad += 1
                print(f"seed {seed} mode {mode} accel {accel} ({len(sc.prims)} objects, {sc.width}x{sc.height}, "
                      f"depth {sc.ray_depth}): " + ", ".join(what), flush=True)
        if (seed - first) % 25 == 24:
            print(f"... {seed - first + 1} scenes", flush=True)
    print(f"{n} scenes x {len(variants)} variants: {bad} mismatching renders")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
