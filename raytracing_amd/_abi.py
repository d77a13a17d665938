"""ctypes mirror of include/rtx.h (plain-data structs and enums).

The struct layouts here must match include/rtx.h byte for byte;
tests/test_abi.py checks sizes and offsets against the built library.
"""
import ctypes as C

# rtx_kind / rtx_material (src/parser.rs:43-47, src/objects/object.rs:5-9 of the reference)
PLANE, ELLIPSOID, BOX = 0, 1, 2
DIFFUSE, METALLIC, DIELECTRIC = 0, 1, 2
# rtx_mode / rtx_accel / rtx_rng
MODE_MEGAKERNEL, MODE_WAVEFRONT = 0, 1
ACCEL_AUTO, ACCEL_BRUTE_LDS, ACCEL_BRUTE_SCALAR, ACCEL_BVH = 0, 1, 2, 3
RNG_PHILOX, RNG_CONST = 0, 1
REDUCE_P2P, REDUCE_RCCL = 0, 1

RTX_OK = 0
RTX_ERR_INVALID_ARG = -1
RTX_ERR_INVALID_SCENE = -2
RTX_ERR_DEVICE = -3
RTX_ERR_NO_SCENE = -4
RTX_ERR_UNSUPPORTED = -5
RTX_ERR_NO_MEMORY = -6

RTX_ABI_VERSION = 5
RTX_MAX_LDS_PRIMS = 512
RTX_MAX_DEVICES = 16
RTX_STATIC_MAX_TILES = 262144


class rtx_prim(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("param", C.c_float * 3),
        ("position", C.c_float * 3),
        ("rotation", C.c_float * 4),
        ("color", C.c_float * 3),
        ("emission", C.c_float * 3),
        ("material", C.c_uint32),
        ("ior", C.c_float),
    ]


class rtx_scene(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("ray_depth", C.c_uint32),
        ("n_samples", C.c_uint32),
        ("background", C.c_float * 3),
        ("cam_pos", C.c_float * 3),
        ("cam_axis", C.c_float * 9),
        ("tg_fov_x", C.c_float),
        ("tg_fov_y", C.c_float),
        ("n_prims", C.c_uint32),
        ("prims", C.POINTER(rtx_prim)),
    ]


class rtx_opts(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("mode", C.c_uint32),
        ("accel", C.c_uint32),
        ("rng_mode", C.c_uint32),
        ("const_words", C.c_uint32 * 4),
        ("device", C.c_int32),
        ("collect_stats", C.c_uint32),
        ("n_devices", C.c_int32),
        ("reduce", C.c_uint32),
        ("devices", C.POINTER(C.c_int32)),
    ]


class rtx_stats(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64),
        ("rays", C.c_uint64),
        ("hits", C.c_uint64 * 3),
        ("cosine_samples", C.c_uint64),
        ("light_samples", C.c_uint64 * 3),
        ("pdf_evals", C.c_uint64),
        ("light_pdf", C.c_uint64 * 3),
        ("diffuse_bounces", C.c_uint64),
        ("metallic_bounces", C.c_uint64),
        ("dielectric_bounces", C.c_uint64),
        ("prim_tests", C.c_uint64 * 3),
        ("light_tests", C.c_uint64 * 3),
        ("bvh_node_tests", C.c_uint64),
    ]

    def as_dict(self):
        out = {}
        for name, ctype in self._fields_:
            v = getattr(self, name)
            out[name] = list(v) if hasattr(v, "__len__") else int(v)
        return out
