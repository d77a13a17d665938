"""The BVH builder on the host (raytracing_amd/csrc/rtx_bvh.cpp), without a GPU: tests/native/bvh_check.cpp is
compiled against it twice -- as shipped, and with the exact sweep and the insertion-based optimisation switched
off -- and both trees are checked for what a traversal relies on: every bounded object in exactly one leaf with
the same box either way, planes on the always-tested list, every stored child box containing the boxes below it,
breadth-first node order, depth within the limit.  The shipped tree must not cost more node visits (summed
half-area of its inner nodes) than the plain binned one.  Which tree is built never changes a closest hit
(src/trace.rs:102-121 is a scan over all objects; the traversal returns the scan's result for any tree)."""
import json
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracing_amd", "csrc")
HIP_INCLUDE = "/opt/rocm/include"

pytestmark = pytest.mark.skipif(shutil.which("g++") is None or not os.path.isdir(HIP_INCLUDE),
                                reason="needs g++ and the HIP headers (vector types of rtx_bvh.h)")


@pytest.fixture(scope="module")
def checkers(tmp_path_factory):
    out = tmp_path_factory.mktemp("bvh_check")
    built = {}
    for name, flags in (("shipped", []), ("plain", ["-DRTX_BVH_SWEEP_MAX=0", "-DRTX_BVH_OPT_PASSES=0"])):
        exe = str(out / name)
        subprocess.run(["g++", "-O2", "-std=c++17", "-w", "-D__HIP_PLATFORM_AMD__", "-I" + HIP_INCLUDE, "-I" + CSRC, *flags,
                        "-o", exe, os.path.join(ROOT, "tests", "native", "bvh_check.cpp"), os.path.join(CSRC, "rtx_bvh.cpp")],
                       check=True, timeout=300)
        built[name] = exe
    return built, out


def _dump(path, prims):
    with open(path, "wb") as f:
        f.write(struct.pack("I", len(prims)))
        for p in prims:
            f.write(struct.pack("I3f3f4f3f3fIf", int(p.kind), *p.param, *p.position, *p.rotation, *p.color, *p.emission,
                                int(p.material), p.ior))


def _run(exe, path):
    return json.loads(subprocess.run([exe, path], check=True, capture_output=True, timeout=300).stdout)


def _check_tree(r, prims):
    from raytracing_amd import _abi
    assert r["bad_refs"] == 0 and r["not_contained"] == 0 and r["unvisited_nodes"] == 0 and r["nodes_reached_twice"] == 0
    assert r["depth"] <= 30 and r["builder_depth"] >= r["depth"] - 1
    always = set(r["always_tested_list"])
    for i, p in enumerate(prims):
        if int(p.kind) == _abi.PLANE:
            assert i in always
        assert r["leaf_count_by_object"][i] == (0 if i in always else 1), i
    bounded = len(prims) - len(always)
    assert r["nodes"] == max(bounded - 1, 1 if bounded else 0)


@pytest.mark.parametrize("n_objects, seed", [(2000, 1), (10000, 12345), (300, 7), (9, 3), (7, 5)])
def test_both_trees_hold_every_object_once_and_the_shipped_one_is_no_worse(checkers, n_objects, seed):
    from raytracing_amd import synthetic_scene
    exes, out = checkers
    prims = synthetic_scene(n_objects, seed, 64, 64, 1, 6).prims
    path = str(out / f"s{n_objects}_{seed}.bin")
    _dump(path, prims)
    shipped, plain = _run(exes["shipped"], path), _run(exes["plain"], path)
    _check_tree(shipped, prims)
    _check_tree(plain, prims)
    assert shipped["leaf_boxes"] == plain["leaf_boxes"]          # same objects, same boxes, same grid
    assert shipped["always_tested_list"] == plain["always_tested_list"] and shipped["far_slope"] == plain["far_slope"]
    assert shipped["inner_area"] <= plain["inner_area"] * (1.0 + 1e-9)
    if n_objects >= 2000:
        assert shipped["inner_area"] < 0.99 * plain["inner_area"]   # 10 000 objects: 72.9 -> 70.4


def test_clustered_and_degenerate_inputs(checkers):
    """Objects of very different sizes in clusters (where moving subtrees pays most), many objects at one centre
    (no split separates them: the builder falls back to splitting by index and the depth limit holds), one object,
    none."""
    from raytracing_amd import _abi
    from raytracing_amd.scene import Prim
    exes, out = checkers
    rng = np.random.default_rng(11)
    f = lambda a: tuple(float(np.float32(v)) for v in a)
    clustered = [Prim(_abi.PLANE, (0, 1, 0), position=(0, -60, 0))]
    for c in range(12):
        centre = rng.uniform(-40, 40, size=3)
        scale = 10.0 ** rng.uniform(-1.5, 0.8)
        for _ in range(60):
            kind = _abi.ELLIPSOID if rng.random() < 0.5 else _abi.BOX
            q = rng.normal(size=4)
            q /= np.linalg.norm(q)
            clustered.append(Prim(kind, f(rng.uniform(0.2, 1.0, size=3) * scale), position=f(centre + rng.normal(size=3) * 4 * scale),
                                  rotation=f(q)))
    same_centre = [Prim(_abi.ELLIPSOID, (1.0 + 0.001 * i, 1.0, 1.0), position=(1, 2, 3)) for i in range(3000)]
    for name, prims in (("clustered", clustered), ("same_centre", same_centre), ("one", same_centre[:1]), ("none", clustered[:1])):
        path = str(out / f"{name}.bin")
        _dump(path, prims)
        shipped, plain = _run(exes["shipped"], path), _run(exes["plain"], path)
        _check_tree(shipped, prims)
        _check_tree(plain, prims)
        assert shipped["leaf_boxes"] == plain["leaf_boxes"]
        assert shipped["inner_area"] <= plain["inner_area"] * (1.0 + 1e-9), name
