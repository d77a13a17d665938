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
