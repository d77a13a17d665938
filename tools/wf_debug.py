"""Rounds and lanes served per stage of the wavefront kernel (a -DRTX_WF_DEBUG build):
    tools/build_variant.sh dbg -DRTX_WF_DEBUG
    RTX_LIB=build_variants/dbg/librtx.so python tools/wf_debug.py c3 [spp]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_workload
from raytracing_amd import _abi
from raytracing_amd.renderer import Renderer, load_library

names = ["node_rounds", "node_lanes", "leaf_rounds", "leaf_lanes", "serve_rounds", "serve_done", "serve_fetched",
         "gen_rounds", "gen_lanes", "shade_rounds", "shade_lanes", "iterations"]
wl = sys.argv[1]
sc, _ = build_workload(wl, int(sys.argv[2]) if len(sys.argv) > 2 else 8)
lib = load_library()
buf = torch.zeros(sc.width * sc.height * 3, dtype=torch.float32, device="cuda")
out = (C.c_ulonglong * 16)()
with Renderer(seed=1, mode=_abi.MODE_WAVEFRONT) as r:
    r.set_scene(sc)
    lib.rtx_wf_debug_read(out, 1)
    r.render_samples(0, sc.n_samples, buf.data_ptr(), 0)
    torch.cuda.synchronize()
    lib.rtx_wf_debug_read(out, 1)
v = dict(zip(names, list(out)))
print(v)
for k in ("node", "leaf", "gen", "shade"):
    if v[k + "_rounds"]:
        print(f"{k:6s} rounds {v[k + '_rounds']:>12d}  lanes/round {v[k + '_lanes'] / v[k + '_rounds']:6.2f}")
if v["serve_rounds"]:
    print(f"serve  rounds {v['serve_rounds']:>12d}  done/round {v['serve_done'] / v['serve_rounds']:6.2f}  fetched/round {v['serve_fetched'] / v['serve_rounds']:6.2f}")
