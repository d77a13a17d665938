"""The FLOP accounting behind bench.py's roofline object: algorithmic FLOPs (events x the constants taken from
the reference's source lines), the part of them an exact shortcut elides, and the executed FLOPs from the
committed counters."""
import glob
import json
import os

from raytracing_amd import reference_scene, synthetic_scene
from raytracing_amd.flops import (LIGHT_PDF, TO_LOCAL, algorithmic_flops, elided_flops, executed_flops)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stats(rays, per_kind, diffuse=0, light_tests=(0, 0, 0), light_pdf=(0, 0, 0)):
    return {"samples": 10, "rays": rays, "hits": [0, 0, 0], "cosine_samples": 0, "light_samples": [0, 0, 0],
            "pdf_evals": 0, "light_pdf": list(light_pdf), "diffuse_bounces": diffuse, "metallic_bounces": 0,
            "dielectric_bounces": 0, "prim_tests": [rays * k for k in per_kind], "light_tests": list(light_tests)}


def test_elided_work_of_the_reference_scene():
    sc = reference_scene()
    # 5 planes, 1 sphere, 2 boxes; every object but the tall box is in the identity pose
    st = _stats(rays=100, per_kind=(5, 1, 2), diffuse=7, light_tests=(0, 0, 30), light_pdf=(10, 4, 8))
    got = elided_flops(st, sc, bvh=False)
    want = 100 * 5 * 60 + 100 * 1 * 60 + 100 * 2 // 2 * 60 + 7 * 3 + 30 * 60 + (4 + 2 * 8) * 39
    assert got == want
    assert got < algorithmic_flops(st)
    assert TO_LOCAL == 63 and LIGHT_PDF == (94, 330, 440)          # the constants the elided shares are parts of


def test_elided_work_of_a_bvh_scene_counts_the_planes_only():
    sc = synthetic_scene(n_prims=50, seed=3, width=8, height=8, n_samples=1)
    st = _stats(rays=10, per_kind=(6, 3, 2), diffuse=4, light_tests=(0, 5, 5), light_pdf=(8, 1, 1))
    # rotated ellipsoid / box lights: nothing of the light pdf is elided; leaf tests are not counted
    assert elided_flops(st, sc, bvh=True) == 10 * 6 * 60 + 4 * 3


def test_executed_flops_from_the_committed_counters():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "c2_megakernel_pmc.json")))
    assert files
    pmc = json.load(open(files[-1]))
    ex = executed_flops(pmc)
    # one C2 step: 2 700 algorithmic FLOP per sample x 531 M samples = 1.43e12, about half of them elided
    assert 0.5e12 < ex < 1.43e12
    assert executed_flops({}) is None and executed_flops(None) is None
