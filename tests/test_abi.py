"""CPU tier: the C ABI itself.  The shared library loads, exports every function the two headers declare, the
ctypes structures have the C compiler's layout, and without a GPU the compute entry points fail loudly."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from raytracing_rust_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [os.path.join(ROOT, "include", h) for h in ("rtx.h", "rtx_host.h")]


def declared_functions():
    names = []
    for h in HEADERS:
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
        for m in re.finditer(r"^[A-Za-z_][\w \*]*?\b(rtxh?_\w+)\s*\(", text, flags=re.M):
            names.append(m.group(1))
    return sorted(set(names))


def test_headers_declare_what_the_binding_binds():
    declared = declared_functions()
    assert len(declared) >= 40
    assert declared == sorted(_capi.SYMBOLS), "include/*.h and the ctypes symbol table disagree"


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(_capi.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"librtx_amd.so does not export {missing}"


def test_library_has_a_gfx950_code_object():
    """The kernels are in the library (an offload bundle for gfx950), not JIT-compiled and not a CPU stand-in."""
    blob = open(_capi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"k_wf_traverse" in blob and b"k_wf_shade" in blob


STRUCTS = {
    "rtx_node": _capi.Node, "rtx_aabb": _capi.Aabb, "rtx_sphere": _capi.Sphere, "rtx_quad": _capi.Quad,
    "rtx_xform_op": _capi.XformOp, "rtx_xform": _capi.Xform, "rtx_medium": _capi.Medium,
    "rtx_material": _capi.Material, "rtx_texture": _capi.Texture, "rtx_image": _capi.Image,
    "rtx_perlin": _capi.Perlin, "rtx_light": _capi.Light, "rtx_scene_desc": _capi.SceneDesc,
    "rtx_camera_desc": _capi.CameraDesc, "rtx_render_opts": _capi.RenderOpts, "rtx_stats": _capi.Stats,
}


def test_struct_layouts_match_the_c_compiler(tmp_path):
    """sizeof and every field offset, as gcc sees include/rtx.h, against the ctypes declarations."""
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rtx.h"', 'int main(void) {']
    for cname, cls in STRUCTS.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in STRUCTS.items():
        assert int(out[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(out[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_headers_compile_as_c_and_cpp(tmp_path):
    for compiler, ext in (("gcc", "c"), ("g++", "cpp")):
        src = tmp_path / f"t.{ext}"
        src.write_text('#include "rtx.h"\n#include "rtx_host.h"\nint main(void) { return 0; }\n')
        subprocess.check_call([compiler, "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                               "-o", str(tmp_path / "t.o")])


def test_no_device_means_an_error_not_a_fallback():
    """In a process without a GPU the device layer refuses to create a context (run in a child process so that
    probing HIP does not touch this one)."""
    code = (
        "import ctypes as C, sys\n"
        "from raytracing_rust_amd import _capi\n"
        "h = C.c_void_p()\n"
        "st = _capi.lib.rtx_context_create(1, None, C.byref(h))\n"
        "msg = _capi.lib.rtx_last_error(None)\n"
        "print(st, (msg or b'').decode())\n"
    )
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1", PYTHONPATH=ROOT)
    out = subprocess.check_output([sys.executable, "-c", code], env=env, text=True).strip()
    st, msg = out.split(" ", 1)
    assert int(st) == _capi.RTX_ERR_NO_DEVICE
    assert "no HIP device" in msg and "no CPU path" in msg


def test_missing_library_raises_import_error():
    with pytest.raises(ImportError) as e:
        _capi.load_library(os.path.join(ROOT, "raytracing_rust_amd", "does_not_exist.so"))
    assert "no fallback" in str(e.value).lower()


def test_rust_sys_crate_declares_the_same_abi():
    """bindings/rust/rtx-sys/src/lib.rs cannot be compiled here (no cargo/rustc), so its text is checked instead: the
    extern block names exactly the functions the headers declare, and the #[repr(C)] structs list the fields of the C
    structs in the same order (layouts themselves are pinned against gcc above)."""
    src = open(os.path.join(ROOT, "bindings", "rust", "rtx-sys", "src", "lib.rs")).read()
    assert sorted(set(re.findall(r"pub fn (rtxh?_\w+)\s*\(", src))) == declared_functions()
    for rust_name, cls in (("rtx_camera_desc", _capi.CameraDesc), ("rtx_render_opts", _capi.RenderOpts),
                           ("rtx_stats", _capi.Stats), ("rtx_scene_desc", _capi.SceneDesc)):
        body = re.search(r"pub struct %s \{(.*?)\n\}" % rust_name, src, flags=re.S).group(1)
        fields = re.findall(r"pub (\w+)\s*:", body)
        assert fields == [f for f, _ in cls._fields_], rust_name
