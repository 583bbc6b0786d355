"""GPU tier: a plain-C program (tests/c_caller/caller.c: gcc, include/*.h, -lrtx_amd) records the Cornell scene object
by object and calls rtxh_camera_render, the sequence a binding for the reference performs behind
Camera::render(world, lights) (camera.rs:271-328).  Its image must be the one the ctypes path renders for the library's
own scene function with the same seed: same paths, so the 8-bit images differ at most where the order of the f64
additions moved a value across a quantisation step (tolerance stated: at most 1 step, in under 0.5 % of the values)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import raytracing_rust_amd as rt
from raytracing_rust_amd import _capi, scenes
from test_cli_gpu import read_png

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CALLER = os.path.join(ROOT, "tests", "c_caller", "_build", "caller")


@pytest.mark.gpu
def test_c_caller_renders_the_ctypes_image(gpu_ctx, tmp_path):
    assert os.path.exists(CALLER), "build it with `make -C tests/c_caller` (done by __graft_entry__.build())"
    width, spp, depth, seed = 96, 16, 12, 0x5EED0002
    raw, png = tmp_path / "c.rgb", tmp_path / "c.png"
    out = subprocess.run([CALLER, str(width), str(spp), str(depth), hex(seed), str(raw), str(png)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert f"{width * width * 16} paths" in out.stdout
    from_c = np.frombuffer(open(raw, "rb").read(), dtype=np.uint8).reshape(width, width, 3)
    assert (read_png(png) == from_c).all()

    g = rt.Graph()
    world, lights, cam = scenes.build("cornell", g, width=width, spp=spp, depth=depth)
    rgb = np.empty((width, width, 3), dtype=np.uint8)
    st = _capi.lib.rtxh_camera_render(g.session, gpu_ctx.handle, C.byref(cam), world, lights, seed, None,
                                      rgb.ctypes.data_as(C.POINTER(C.c_uint8)), None)
    assert st == _capi.RTX_OK
    diff = np.abs(rgb.astype(int) - from_c.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.005, (diff.max(), (diff > 0).mean())
    assert from_c.std() > 10
