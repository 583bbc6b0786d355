"""The reference's entry points (src/main.rs:12-40) and their scene functions.

`build(name, graph, ...)` calls the C++ scene builders of include/rtx_host.h
(scenes/cornell_box.rs, scenes/final_scene.rs restated) and returns
(world, lights, camera_desc) with the camera initialised.
"""
import ctypes as C
import os

import numpy as np

from . import _capi
from .render import initialize_camera

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
SCENE_SEED = 0x5EED0001   # random scene content (box heights, sphere centres, Perlin tables)
PATH_SEED = 0x5EED0002    # path RNG key

# name -> (scene function, image_width, samples_per_pixel, max_depth)   [main.rs:13-39]
ENTRY_POINTS = {
    "cornell": ("cornell", 600, 1000, 50),
    "final": ("final", 800, 5000, 75),
    "quick": ("final", 400, 100, 20),
    "smoke": ("cornell_smoke", 600, 1000, 50),
    # not a reference entry point: 1 M random spheres and quads in one BVH, the synthetic stress configuration
    "stress": ("stress", 3840, 4096, 50),
}
STRESS_SEED = 0x5EED0005
STRESS_PRIMS = 1_000_000


def load_ppm(path):
    with open(path, "rb") as f:
        data = f.read()
    parts = data.split(None, 4)
    assert parts[0] == b"P6" and parts[3] == b"255", "binary PPM (P6, maxval 255) expected"
    w, h = int(parts[1]), int(parts[2])
    return np.frombuffer(parts[4][: w * h * 3], dtype=np.uint8).reshape(h, w, 3)


def earth_pixels():
    """Decoded texels of the reference's textures/earthmap.jpg (see tools/make_earthmap_ppm.py)."""
    return load_ppm(os.path.join(ASSETS, "earthmap.ppm"))


def build(name, graph, width=None, spp=None, depth=None, scene_seed=SCENE_SEED, prefix="rtxh_", lib=None,
          n_prims=STRESS_PRIMS):
    """Build one of the named scenes into `graph`.  Returns (world, lights, cam)."""
    fn_name, w0, s0, d0 = ENTRY_POINTS[name]
    width, spp, depth = width or w0, spp or s0, depth or d0
    lib = lib or _capi.lib
    cam = _capi.CameraDesc()
    lights = C.c_int(-1)
    fn = getattr(lib, prefix + "scene_" + fn_name)
    if fn_name == "final":
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_int), C.POINTER(_capi.CameraDesc)]
        earth = graph.image(earth_pixels())
        world = fn(graph.session, width, spp, depth, scene_seed, earth, C.byref(lights), C.byref(cam))
    elif fn_name == "stress":
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(_capi.CameraDesc)]
        seed = STRESS_SEED if scene_seed == SCENE_SEED else scene_seed
        world = fn(graph.session, int(n_prims), seed, width, spp, depth, C.byref(lights), C.byref(cam))
    else:
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(_capi.CameraDesc)]
        world = fn(graph.session, width, spp, depth, C.byref(lights), C.byref(cam))
    if world < 0:
        raise RuntimeError(f"scene {name} failed to build")
    initialize_camera(cam)
    return world, lights.value, cam
