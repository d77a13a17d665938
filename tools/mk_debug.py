"""Rounds and lanes served per step of the two-pixel megakernel (a -DRTX_MK_DEBUG build):
    tools/build_variant.sh dbg -DRTX_MK_DEBUG
    RTX_LIB=build_variants/dbg/librtx.so python tools/mk_debug.py c2 [spp]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_workload
from raytracing_amd.renderer import Renderer, load_library

names = ["rounds", "scan_rounds", "scan_lanes", "scatter_rounds", "scatter_lanes", "swap_rounds", "swap_lanes", "fresh_lanes",
         "node_rounds", "node_lanes", "leaf_rounds", "leaf_lanes", "start_rounds", "start_lanes"]
sc, _ = build_workload(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 32)
lib = load_library()
buf = torch.zeros(sc.width * sc.height * 3, dtype=torch.float32, device="cuda")
out = (C.c_ulonglong * 16)()
with Renderer(seed=1) as r:
    r.set_scene(sc)
    lib.rtx_mk_debug_read(out, 1)
    r.render_samples(0, sc.n_samples, buf.data_ptr(), 0)
    torch.cuda.synchronize()
    lib.rtx_mk_debug_read(out, 1)
v = dict(zip(names, list(out)))
print(v)
if v["node_rounds"]:
    print("BVH megakernel: outer rounds %d; traversal starts in %.2f of them with %.1f lanes; node steps %.1f per round with %.1f lanes; "
          "object tests %.2f per round with %.1f lanes; scatter in %.2f of the rounds with %.1f lanes" % (
              v["rounds"], v["start_rounds"] / v["rounds"], v["start_lanes"] / max(1, v["start_rounds"]), v["node_rounds"] / v["rounds"],
              v["node_lanes"] / v["node_rounds"], v["leaf_rounds"] / v["rounds"], v["leaf_lanes"] / max(1, v["leaf_rounds"]),
              v["scatter_rounds"] / v["rounds"], v["scatter_lanes"] / max(1, v["scatter_rounds"])))
    sys.exit(0)
print("rounds %d: scan in %.2f of them with %.1f lanes; scatter in %.2f with %.1f lanes; swaps in %.2f per round with %.1f lanes; fresh %.1f per round" % (
    v["rounds"], v["scan_rounds"] / v["rounds"], v["scan_lanes"] / max(1, v["scan_rounds"]), v["scatter_rounds"] / v["rounds"],
    v["scatter_lanes"] / max(1, v["scatter_rounds"]), v["swap_rounds"] / v["rounds"], v["swap_lanes"] / max(1, v["swap_rounds"]),
    v["fresh_lanes"] / v["rounds"]))
