"""Shared test plumbing: loaders for the oracle (oracle/_build/liboracle.so) and the host build of the device
core (tests/hostsim/_build/libhostsim.so), both test infrastructure, plus small drivers around them."""
import ctypes as C
import os
import subprocess

import numpy as np

import raytracing_rust_amd as rt
from raytracing_rust_amd import _capi
from raytracing_rust_amd.graph import Graph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
HOSTSIM_SO = os.path.join(ROOT, "tests", "hostsim", "_build", "libhostsim.so")

PD = C.POINTER(C.c_double)
PU32 = C.POINTER(C.c_uint32)
PU8 = C.POINTER(C.c_uint8)


def _ensure(path, make_dir):
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", make_dir], stdout=subprocess.DEVNULL)
    return path


class OracleOpts(C.Structure):
    _fields_ = [("sample_begin", C.c_int32), ("sample_end", C.c_int32), ("x0", C.c_int32), ("y0", C.c_int32),
                ("x1", C.c_int32), ("y1", C.c_int32), ("seed", C.c_uint64), ("use_lights", C.c_int32),
                ("n_threads", C.c_int32)]


_oracle = None
_hostsim = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(_ensure(ORACLE_SO, os.path.join(ROOT, "oracle")))
        lib.orc_first_hit.restype = C.c_int
        lib.orc_first_hit.argtypes = [C.c_void_p, C.c_int, PD, C.c_size_t, C.c_int, C.c_uint64, PD, PD, PD, PD, PU32, PU32, PU8]
        lib.orc_render.restype = C.c_int
        lib.orc_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(_capi.CameraDesc), C.POINTER(OracleOpts), PD,
                                   C.POINTER(C.c_uint64)]
        lib.orc_camera_initialize.argtypes = [C.POINTER(_capi.CameraDesc)]
        lib.orc_pdf_value.argtypes = [C.c_void_p, C.c_int, PD, PD, C.c_size_t, PD]
        lib.orc_random_dirs.argtypes = [C.c_void_p, C.c_int, PD, C.c_uint64, C.c_size_t, PD]
        lib.orc_tex_value.argtypes = [C.c_void_p, C.c_int, PD, C.c_size_t, PD]
        lib.orc_pdf_sample.argtypes = [C.c_int, PD, C.c_uint64, C.c_size_t, PD, PD]
        lib.orc_mat_scatter.restype = C.c_int
        lib.orc_mat_scatter.argtypes = [C.c_void_p, C.c_int, PD, PD, PD, C.c_int, C.c_uint64, PD, PD]
        lib.orc_resolve_rgb8.argtypes = [PD, C.c_int, C.c_int, C.c_int, PU8]
        lib.orc_ray_color.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(_capi.CameraDesc), PD, C.c_size_t, C.c_int,
                                      C.c_uint64, PD]
        _oracle = lib
    return _oracle


def hostsim_lib():
    global _hostsim
    if _hostsim is None:
        lib = C.CDLL(_ensure(HOSTSIM_SO, os.path.join(ROOT, "tests", "hostsim")))
        lib.sim_first_hit.argtypes = [C.POINTER(_capi.SceneDesc), PD, C.c_size_t, PD, PD, PU32, PU32, PU8]
        lib.sim_render.argtypes = [C.POINTER(_capi.SceneDesc), C.POINTER(_capi.CameraDesc), C.POINTER(_capi.RenderOpts), PD,
                                   C.POINTER(C.c_uint64), C.c_int]
        lib.sim_resolve_rgb8.argtypes = [PD, C.c_int, C.c_int, C.c_int, PU8]
        lib.sim_rng_double.restype = C.c_double
        lib.sim_rng_double.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.sim_philox.argtypes = [PU32, PU32, PU32]
        lib.sim_texture_value.argtypes = [C.POINTER(_capi.SceneDesc), C.c_uint32, PD, C.c_size_t, PD]
        lib.sim_lights_pdf.argtypes = [C.POINTER(_capi.SceneDesc), PD, PD, C.c_size_t, PD]
        lib.sim_lights_random.argtypes = [C.POINTER(_capi.SceneDesc), PD, C.c_uint64, C.c_size_t, PD]
        lib.sim_camera_ray.argtypes = [C.POINTER(_capi.CameraDesc), C.c_int, C.c_int, C.c_int, C.c_uint64, PD]
        _hostsim = lib
    return _hostsim


def oracle_graph():
    """A Graph that records into the oracle's trait-object scene instead of the product's host layer."""
    return Graph(lib=oracle_lib(), prefix="orc_")


def dptr(a):
    return a.ctypes.data_as(PD)


def oracle_first_hit(g, world, rays, skip_media=True, seed=1):
    r = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 7)
    n = r.shape[0]
    out = {"t": np.empty(n), "normal": np.empty((n, 3)), "p": np.empty((n, 3)), "uv": np.empty((n, 2)),
           "prim_id": np.empty(n, dtype=np.uint32), "mat_id": np.empty(n, dtype=np.uint32),
           "front": np.empty(n, dtype=np.uint8)}
    st = oracle_lib().orc_first_hit(g.session, world, dptr(r), n, 1 if skip_media else 0, seed, dptr(out["t"]),
                                    dptr(out["normal"]), dptr(out["p"]), dptr(out["uv"]),
                                    out["prim_id"].ctypes.data_as(PU32), out["mat_id"].ctypes.data_as(PU32),
                                    out["front"].ctypes.data_as(PU8))
    assert st == 0
    return out


def hostsim_first_hit(desc, rays):
    r = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 7)
    n = r.shape[0]
    out = {"t": np.empty(n), "normal": np.empty((n, 3)), "prim_id": np.empty(n, dtype=np.uint32),
           "mat_id": np.empty(n, dtype=np.uint32), "front": np.empty(n, dtype=np.uint8)}
    hostsim_lib().sim_first_hit(C.byref(desc), dptr(r), n, dptr(out["t"]), dptr(out["normal"]),
                                out["prim_id"].ctypes.data_as(PU32), out["mat_id"].ctypes.data_as(PU32),
                                out["front"].ctypes.data_as(PU8))
    return out


def hostsim_first_hit_staged(desc, rays, near=0):
    """First hits through the stages of the wavefront pipeline (streamed head section, rest of the world program,
    streamed deferred section).  Returns (aov dict, number of rays that went through a stream walk).
    near: 1 = the deferred section nearer child first, 2 = the head section too (hostsim_lib().sim_near_walks()
    tells how many sections were walked that way)."""
    hostsim_lib().sim_set_near(near)
    r = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 7)
    n = r.shape[0]
    out = {"t": np.empty(n), "normal": np.empty((n, 3)), "prim_id": np.empty(n, dtype=np.uint32),
           "mat_id": np.empty(n, dtype=np.uint32), "front": np.empty(n, dtype=np.uint8)}
    k = hostsim_lib().sim_first_hit_staged(C.byref(desc), dptr(r), n, dptr(out["t"]), dptr(out["normal"]),
                                           out["prim_id"].ctypes.data_as(PU32), out["mat_id"].ctypes.data_as(PU32),
                                           out["front"].ctypes.data_as(PU8))
    hostsim_lib().sim_set_near(0)
    return out, k


def oracle_render(g, world, lights, cam, sample_begin=0, sample_end=None, rect=None, seed=7, use_lights=True, threads=8):
    o = OracleOpts()
    o.sample_begin = sample_begin
    o.sample_end = cam.sqrt_spp ** 2 if sample_end is None else sample_end
    o.x0, o.y0, o.x1, o.y1 = rect if rect is not None else (0, 0, cam.image_width, cam.image_height)
    o.seed, o.use_lights, o.n_threads = seed, 1 if use_lights else 0, threads
    accum = np.zeros((cam.image_height, cam.image_width, 3))
    counters = (C.c_uint64 * 6)()
    st = oracle_lib().orc_render(g.session, world, -1 if lights is None else lights, C.byref(cam), C.byref(o), dptr(accum), counters)
    assert st == 0
    names = ["paths", "segments", "pdf_probes", "box_tests", "sphere_tests", "quad_tests"]
    return accum, dict(zip(names, [int(c) for c in counters]))


def hostsim_render(desc, cam, opts, threads=8):
    accum = np.zeros((cam.image_height, cam.image_width, 3))
    counters = (C.c_uint64 * 3)()
    hostsim_lib().sim_render(C.byref(desc), C.byref(cam), C.byref(opts), dptr(accum), counters, threads)
    return accum, {"paths": int(counters[0]), "segments": int(counters[1]), "pdf_probes": int(counters[2])}


def assert_aov_equal(a, b, what=""):
    """Bit-exact comparison of first-hit AOVs (t, normal, ids, front flag)."""
    for k in ("prim_id", "mat_id", "front"):
        bad = np.nonzero(a[k] != b[k])[0]
        assert bad.size == 0, f"{what}: {k} differs at {bad[:5]} ({a[k][bad[:5]]} vs {b[k][bad[:5]]}), {bad.size} rays"
    for k in ("t", "normal"):
        x, y = a[k].view(np.uint64), b[k].view(np.uint64)
        bad = np.nonzero(x != y)
        assert bad[0].size == 0, f"{what}: {k} not bit-identical at {bad[0][:5]}: {a[k][bad][:5]} vs {b[k][bad][:5]}"


def build_both(build_fn):
    """Run the same recipe on the oracle graph and on the product graph."""
    go, gp = oracle_graph(), rt.Graph()
    ro, rp = build_fn(go), build_fn(gp)
    return go, ro, gp, rp


def camera_rays(cam, sample, seed):
    """The device's camera ray of every pixel for one stratified sample index (Camera::get_ray with the path RNG's
    addressing; only + - x / on doubles, so the host build and the GPU make the same bits).  (H*W, 7)."""
    lib = hostsim_lib()
    out = np.empty((cam.image_height * cam.image_width, 7))
    buf = (C.c_double * 7)()
    k = 0
    for j in range(cam.image_height):
        for i in range(cam.image_width):
            lib.sim_camera_ray(C.byref(cam), i, j, sample, seed, buf)
            out[k] = buf[:]
            k += 1
    return out


def oracle_first_bounce_emission(g, world, texs, cam, sample, seed):
    """What a path of depth 1 returns in the oracle for the device's camera rays of one sample index, when every
    primitive of `world` is a DiffuseLight whose texture is texs[prim_id]: the texture value at the oracle's own hit
    record (u, v, p), or the background for a miss (camera.rs:179-190).  (H, W, 3)."""
    rays = camera_rays(cam, sample, seed)
    hit = oracle_first_hit(g, world, rays)
    out = np.empty((len(rays), 3))
    out[:] = np.array(cam.background[:])
    for prim, tex in enumerate(texs):
        sel = np.nonzero(hit["prim_id"] == prim)[0]
        if sel.size == 0:
            continue
        uvp = np.ascontiguousarray(np.concatenate([hit["uv"][sel], hit["p"][sel]], axis=1))
        val = np.empty((sel.size, 3))
        assert oracle_lib().orc_tex_value(g.session, tex, dptr(uvp), sel.size, dptr(val)) == 0
        out[sel] = val
    return out.reshape(cam.image_height, cam.image_width, 3), hit
