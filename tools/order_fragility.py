"""How much of the oracle's output hangs on the operation orders recalled for the reference's
dependencies (nalgebra's dot / normalize / quaternion rotation / matrix-vector product: sources
not available, oracle/rt_oracle.c header)?  Rebuilds the oracle with each order swapped for
another plausible one (make -C oracle alternatives) and compares, on the reference scene:
  * the primary pass at 1920x1080 (src/main.rs:19-21 with du = dv = 0.5, src/trace.rs:102-121):
    hit ids that change, largest relative change of a hit distance and a normal component;
  * a seeded 64x64x4 render in the reference's order: pixels that move by more than 1e-5
    relative, and the largest relative change.
Each alternative runs in a process of its own (the library is chosen at import):
    python tools/order_fragility.py            prints a table (markdown)
    python tools/order_fragility.py --one X    prints one JSON record (used by the above)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALTS = ["DOT", "NORMALIZE", "QUAT", "MATVEC"]
WHAT = {"DOT": "dot = x + (y + z)", "NORMALIZE": "normalize = v * (1 / |v|)", "QUAT": "q * v = (v + t w) + u x t",
        "MATVEC": "M v = m0 x + (m1 y + m2 z)"}


def outputs(width=1920, height=1080):
    sys.path.insert(0, ROOT)
    import numpy as np
    from oracle import pyoracle as O
    from raytracing_amd import reference_scene
    ids, t, n, _ = O.primary(reference_scene(width=width, height=height))
    img = O.render(reference_scene(width=64, height=64, n_samples=4), seed=1, order=O.ORDER_REFERENCE)
    return ids, t, n, img


def compare(base, alt):
    import numpy as np
    ids0, t0, n0, img0 = base
    ids1, t1, n1, img1 = alt
    same = ids0 == ids1
    hit = same & (ids0 >= 0)
    rel = lambda a, b: np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-6)
    rimg = rel(img0, img1)
    return {"pixels": int(ids0.size), "ids_flipped": int((~same).sum()),
            "t_bits_changed": int((t0.view(np.uint32) != t1.view(np.uint32))[hit].sum()),
            "t_beyond_1e-5": int((rel(t0, t1)[hit] > 1e-5).sum()), "t_max_rel": float(rel(t0, t1)[hit].max()), "n_max_abs": float(np.abs(n0 - n1)[hit].max()),
            "render_values_beyond_1e-5": int((rimg > 1e-5).sum()), "render_values": int(rimg.size),
            "render_max_rel": float(rimg.max())}


def run_alternative(name, size):
    lib = os.path.join(ROOT, "oracle", "_alt", f"librt_oracle_{name}.so")
    env = dict(os.environ, RT_ORACLE_LIB=lib)
    out = subprocess.check_output([sys.executable, os.path.abspath(__file__), "--one", name, str(size[0]), str(size[1])], env=env)
    return json.loads(out.decode().strip().splitlines()[-1])


def study(size=(1920, 1080)):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "alternatives"])
    return {name: run_alternative(name, size) for name in ALTS}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        import numpy as np
        w, h = int(sys.argv[3]), int(sys.argv[4])
        alt = outputs(w, h)                      # this process loaded the alternative library
        # the default build, through a second copy of the binding
        import importlib
        os.environ.pop("RT_ORACLE_LIB")
        for m in [m for m in sys.modules if m.startswith("oracle")]:
            del sys.modules[m]
        base = outputs(w, h)
        print(json.dumps(compare(base, alt)))
    else:
        res = study()
        print("| order swapped | primary ids flipped (of 2 073 600) | t: values changed / beyond 1e-5 rel / max rel | normal max abs | 64x64x4 render: values > 1e-5 / max rel |")
        print("|---|---|---|---|---|")
        for name, r in res.items():
            print(f"| {WHAT[name]} | {r['ids_flipped']} | {r['t_bits_changed']} / {r['t_beyond_1e-5']} / {r['t_max_rel']:.2e} | {r['n_max_abs']:.2e} | "
                  f"{r['render_values_beyond_1e-5']} of {r['render_values']} / {r['render_max_rel']:.2e} |")
