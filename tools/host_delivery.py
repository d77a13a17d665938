"""What rtx_render costs beyond its kernels when the image goes to host memory: end-to-end time of the
call minus the render kernels' time (the slowest worker's HIP events), for a multi-GPU context whose
device list names one GPU several times.  Loads the library given by RTX_LIB (default: the in-tree one)
with a handful of prototypes of its own, so that it also runs against an older build of the ABI.

    python tools/host_delivery.py [workers] [width height spp]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401  (one HIP runtime per process: torch's)
from raytracing_amd import _abi, reference_scene

workers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w, h, spp = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (3840, 2160, 16)
path = os.environ.get("RTX_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "raytracing_amd", "librtx.so")
lib = C.CDLL(path)
FP = C.POINTER(C.c_float)
lib.rtx_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(_abi.rtx_opts)]
lib.rtx_set_scene.argtypes = [C.c_void_p, C.POINTER(_abi.rtx_scene)]
lib.rtx_render.argtypes = [C.c_void_p, C.POINTER(_abi.rtx_scene), FP]
lib.rtx_last_kernel_ms.argtypes = [C.c_void_p, FP]
lib.rtx_destroy.argtypes = [C.c_void_p]
lib.rtx_last_error.restype = C.c_char_p

pinned_ok = hasattr(lib, "rtx_alloc_pinned")
if pinned_ok:
    lib.rtx_alloc_pinned.argtypes = [C.POINTER(C.c_void_p), C.c_uint64]
    lib.rtx_free_pinned.argtypes = [C.c_void_p]

sc = reference_scene(w, h, spp, 6)
s, keep = sc.to_ctypes()
for n, pinned in ((1, False), (workers, False), (1, True), (workers, True)):
    if pinned and not pinned_ok:
        continue
    o = _abi.rtx_opts()
    o.seed = 1
    devs = (C.c_int32 * n)(*([0] * n))
    o.n_devices, o.devices = n, devs
    ctx = C.c_void_p()
    assert lib.rtx_create(C.byref(ctx), C.byref(o)) == 0, lib.rtx_last_error()
    if pinned:
        ptr = C.c_void_p()
        assert lib.rtx_alloc_pinned(C.byref(ptr), h * w * 12) == 0, lib.rtx_last_error()
        out = np.ctypeslib.as_array(C.cast(ptr, FP), shape=(h * w * 3,)).reshape(h, w, 3)
    else:
        out = np.empty((h, w, 3), dtype=np.float32)
    rows = []
    for k in range(6):
        t0 = time.perf_counter()
        assert lib.rtx_render(ctx, C.byref(s), out.ctypes.data_as(FP)) == 0, lib.rtx_last_error()
        dt = (time.perf_counter() - t0) * 1e3
        ms = C.c_float()
        lib.rtx_last_kernel_ms(ctx, C.byref(ms))
        rows.append((dt, ms.value))
    lib.rtx_destroy(ctx)
    rows = rows[2:]
    e2e = sum(r[0] for r in rows) / len(rows)
    ker = sum(r[1] for r in rows) / len(rows)
    print(f"{os.path.basename(os.path.dirname(path)) or 'in-tree'} workers={n} {'pinned' if pinned else 'pageable'} {w}x{h}x{spp}: rtx_render {e2e:.2f} ms, "
          f"slowest worker's kernels {ker:.2f} ms, beyond the kernels {e2e - ker:.2f} ms  checksum {float(out.astype(np.float64).sum()):.6e}")
