import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import synthetic_scene, _abi
from raytracing_amd.renderer import Renderer
sc = synthetic_scene(10000, 12345, 480, 270, 16, 6)
with Renderer(seed=1, accel=_abi.ACCEL_BVH, collect_stats=True) as r:
    t=time.time(); r.render(sc); dt=time.time()-t
    st = r.stats()
rays = st["rays"]
print(json.dumps(st))
print("rays/sample", rays/st["samples"], "node tests/ray", st["bvh_node_tests"]/rays, "prim tests/ray", sum(st["prim_tests"])/rays,
      "light tests per pdf eval", sum(st["light_tests"])/max(1,st["pdf_evals"]), "pdf evals/ray", st["pdf_evals"]/rays, "hits", st["hits"])
