"""Algorithmic FLOP constants of the trace loop, per path event.

Counting rule: +, -, *, /, sqrt, min, max, sin, cos = 1 FLOP each; negation, abs,
compare and select = 0.  Each constant is the operation count of the reference
source lines named beside it (reference paths are relative to its src/ directory).
A render's algorithmic FLOPs are sum(events * constants), with the events counted on
the device (rtx_get_stats); nothing the device does beyond these (RNG integer
arithmetic, address arithmetic, re-computation) is credited.
"""

PRIMARY_RAY = 34          # main.rs:19-20, camera.rs:18-19, ray.rs:14
RAY_LENGTH = 6            # trace.rs:107
TO_LOCAL = 63             # objects/geometry.rs:24-27: 3 sub + 2 quaternion rotations of 30
INNER_TEST = (11, 32, 22)  # plane geometry.rs:58 | ellipsoid :75-91 | box :114-124
RANGE_FILTER = 1          # trace.rs:114
HIT_FINALISE = 44 + 6     # geometry.rs:30-31 (rotate 30, normalize 9, flip dot 5) + trace.rs:21
HIT_NORMAL = (0, 9, 14)   # geometry.rs:104 | :136-143
BVH_NODE_TEST = 22        # slab test of one node box with a precomputed 1/d (this build's definition)
COSINE_SAMPLE = 65        # random.rs:34-52
LIGHT_SAMPLE = 60         # random.rs:76-80, objects/sample.rs:14-50
MIS_PDF_MIX = 11          # random.rs:56, 144-146
LIGHT_PDF = (94, 330, 440)  # random.rs:88-105, 118-122, objects/sample.rs:19-22, 53-57: 0 / 1 / 2 hits
DIFFUSE_BOUNCE = 35       # trace.rs:43-48, 70, ray.rs:18-23
METALLIC_BOUNCE = 35      # trace.rs:54-56, 123-126
DIELECTRIC_BOUNCE = 91    # trace.rs:84-99, 128-149


def algorithmic_flops(stats: dict, bvh_node_tests: int = 0) -> int:
    """FLOPs of one render from its event counters (an rtx_stats as a dict)."""
    f = stats["samples"] * PRIMARY_RAY + stats["rays"] * RAY_LENGTH
    for k in range(3):
        f += stats["prim_tests"][k] * (TO_LOCAL + INNER_TEST[k] + RANGE_FILTER)
        f += stats["hits"][k] * (HIT_FINALISE + HIT_NORMAL[k])
    f += stats["cosine_samples"] * COSINE_SAMPLE
    f += sum(stats["light_samples"]) * LIGHT_SAMPLE
    f += stats["pdf_evals"] * MIS_PDF_MIX
    for k in range(3):
        f += stats["light_pdf"][k] * LIGHT_PDF[k]
    f += stats["diffuse_bounces"] * DIFFUSE_BOUNCE
    f += stats["metallic_bounces"] * METALLIC_BOUNCE
    f += stats["dielectric_bounces"] * DIELECTRIC_BOUNCE
    f += bvh_node_tests * BVH_NODE_TEST
    return int(f)


# MI355X FP32 vector peak: 256 CUs x 4 SIMDs x 64 FLOP/clk (packed FMA) x 2.4 GHz
PEAK_FP32_VECTOR_TFLOPS = 157.3
# one non-packed FMA per lane per clock
PEAK_FP32_NONPACKED_TFLOPS = 78.6


def _identity_pose(p) -> bool:
    return tuple(float(x) for x in p.rotation) == (0.0, 0.0, 0.0, 1.0)


def elided_flops(stats: dict, scene, bvh: bool) -> int:
    """The part of algorithmic_flops() the kernels provably never execute, because an exact
    shortcut returns the same bits without the arithmetic the constants above credit:

    * the two quaternion rotations of the world -> local map (60 of TO_LOCAL's 63) for an object
      in the identity pose: every object test of a linear scan (all rays of a tame scene are
      "plain"), and the tests of the always-tested planes of a BVH scene.  BVH leaf tests skip
      them only when all 64 lanes hold identity-pose objects, which is not counted;
    * colour / pi (3 divisions per diffuse bounce), divided once at upload;
    * for a light that is a box in the identity pose, per light intersection: the two rotations
      (60) and, when it hits, the rotation and normalisation of its normal (39), replaced by
      |n . d| = |d_k|.
    Power-of-two radii turn divisions into as many multiplications, which is not an elision.
    """
    from . import _abi
    per_kind_identity = [0, 0, 0]
    per_kind = [0, 0, 0]
    for p in scene.prims:
        per_kind[p.kind] += 1
        if _identity_pose(p):
            per_kind_identity[p.kind] += 1
    f = 0
    for k in range(3):
        if per_kind[k] == 0 or (bvh and k != _abi.PLANE):
            continue
        # tests are spread evenly over the objects of a kind (every ray tests every scanned object)
        f += stats["prim_tests"][k] * per_kind_identity[k] // per_kind[k] * 60
    f += stats["diffuse_bounces"] * 3
    lights = scene.light_indices()
    if lights and all(scene.prims[i].kind == _abi.BOX and _identity_pose(scene.prims[i]) for i in lights):
        tests = sum(stats["light_tests"])
        hits = stats["light_pdf"][1] + 2 * stats["light_pdf"][2]
        f += tests * 60 + hits * 39
    return int(f)


def executed_flops(pmc: dict):
    """FP32 operations the lanes actually executed in one profiled step, from the rocprofv3
    counters of that step (wave-level instruction counts by class): (ADD + MUL + TRANS + 2 FMA)
    x 64 lanes x the thread-active fraction of the step's VALU instructions.  A packed instruction
    (v_pk_fma_f32, v_pk_mul_f32: two operations per lane) is counted once by its class counter,
    so the figure is a lower bound by the packed share (about 8 % of the static VALU count of the
    reference scene's kernel).  None when the counters are missing."""
    need = ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
            "thread_active_fraction")
    if not pmc or any(k not in pmc for k in need):
        return None
    ops = (pmc["SQ_INSTS_VALU_ADD_F32"] + pmc["SQ_INSTS_VALU_MUL_F32"] + pmc["SQ_INSTS_VALU_TRANS_F32"]
           + 2.0 * pmc["SQ_INSTS_VALU_FMA_F32"])
    return ops * 64.0 * pmc["thread_active_fraction"]
