"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/rtx.h declares; struct layouts agree between the header, the Python binding and
the oracle.  No compute call is made here."""
import ctypes as C
import os
import re

import pytest

from raytracing_amd import _abi
from raytracing_amd import renderer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "rtx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtx_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = renderer.load_library()
    names = header_functions()
    assert "rtx_render" in names and "rtx_primary" in names and len(names) >= 12
    for name in names:
        assert hasattr(lib, name), f"librtx.so does not export {name}"
    assert set(names) == set(renderer.SYMBOLS), "Python binding and header disagree"
    assert lib.rtx_abi_version() == _abi.RTX_ABI_VERSION


def test_struct_layouts():
    assert C.sizeof(_abi.rtx_prim) == 76
    assert _abi.rtx_prim.param.offset == 4 and _abi.rtx_prim.rotation.offset == 28
    assert _abi.rtx_prim.material.offset == 68 and _abi.rtx_prim.ior.offset == 72
    assert C.sizeof(_abi.rtx_scene) == 96 and _abi.rtx_scene.prims.offset == 88
    assert _abi.rtx_scene.cam_axis.offset == 40 and _abi.rtx_scene.tg_fov_x.offset == 76
    assert C.sizeof(_abi.rtx_opts) == 64 and _abi.rtx_opts.const_words.offset == 20
    assert _abi.rtx_opts.n_devices.offset == 44 and _abi.rtx_opts.devices.offset == 56
    assert C.sizeof(_abi.rtx_stats) == 23 * 8


def test_oracle_structs_share_the_layout(oracle):
    assert C.sizeof(oracle.orc_prim) == C.sizeof(_abi.rtx_prim)
    assert C.sizeof(oracle.orc_scene) == C.sizeof(_abi.rtx_scene)
    assert C.sizeof(oracle.orc_stats) + 8 == C.sizeof(_abi.rtx_stats)  # + bvh_node_tests
    for f, _ in _abi.rtx_prim._fields_:
        assert getattr(oracle.orc_prim, f).offset == getattr(_abi.rtx_prim, f).offset


def test_create_reports_errors_instead_of_aborting():
    lib = renderer.load_library()
    ctx = C.c_void_p()
    assert lib.rtx_create(C.byref(ctx), None) == _abi.RTX_ERR_INVALID_ARG
    assert b"null" in lib.rtx_last_error()
    o = _abi.rtx_opts()
    o.mode = 99
    assert lib.rtx_create(C.byref(ctx), C.byref(o)) == _abi.RTX_ERR_INVALID_ARG
    assert lib.rtx_render(None, None, None) == _abi.RTX_ERR_INVALID_ARG
    lib.rtx_destroy(None)


def test_product_package_does_not_import_the_oracle():
    """The oracle is a checker: nothing under raytracing_amd/ may import or link it."""
    pkg = os.path.join(ROOT, "raytracing_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h", ".hpp")) or fn == "Makefile":
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "pyoracle" not in text and "rt_oracle" not in text and "import oracle" not in text, fn


def test_rust_bindings_name_the_header_functions_and_fields():
    """integration/rtx-sys cannot be compiled here (no Rust toolchain); at least its extern
    block and struct fields must be the header's, in the header's order."""
    rs = open(os.path.join(ROOT, "integration", "rtx-sys", "src", "lib.rs")).read()
    extern = rs[rs.index('extern "C" {'):]
    extern = extern[:extern.index("\n}\n")]
    assert sorted(re.findall(r"pub fn (rtx_[a-z_0-9]+)\(", extern)) == header_functions()
    assert f"RTX_ABI_VERSION: u32 = {_abi.RTX_ABI_VERSION};" in rs
    for name, ctype in (("rtx_prim", _abi.rtx_prim), ("rtx_scene", _abi.rtx_scene), ("rtx_opts", _abi.rtx_opts),
                        ("rtx_stats", _abi.rtx_stats)):
        body = rs[rs.index(f"pub struct {name} {{"):]
        body = body[:body.index("\n}\n")]
        assert re.findall(r"pub ([a-z_0-9]+):", body) == [f for f, _ in ctype._fields_], name


def test_host_patch_uses_the_header_names():
    """integration/host_patch/ holds the Rust functions a maintainer adds to the reference; the
    struct literals in them must name exactly the fields of the header's structs, in order."""
    patch = os.path.join(ROOT, "integration", "host_patch")
    main = open(os.path.join(patch, "main.rs")).read()
    parser = open(os.path.join(patch, "parser.rs")).read()
    body = main[main.index("rtx_sys::rtx_scene {"):]
    body = body[:body.index("};")]
    assert re.findall(r"\n\s+([a-z_0-9]+)(?::|,)", body) == [f for f, _ in _abi.rtx_scene._fields_]
    body = parser[parser.index("rtx_sys::rtx_prim {"):]
    body = body[:body.index("}\n")]
    assert re.findall(r"\n\s+([a-z_0-9]+)(?::|,)", body) == [f for f, _ in _abi.rtx_prim._fields_]
    for name in re.findall(r"rtx_sys::(RTX_[A-Z_0-9]+)", parser):
        assert f"pub const {name}:" in open(os.path.join(ROOT, "integration", "rtx-sys", "src", "lib.rs")).read(), name
    assert "n_devices: gpu_count()" in main and "ctx.render(" in main
