"""Replay plumbing of the parity tests: one set of caller-supplied numbers takes the place of the random number
generator in the oracle (oracle.cpp, Tape) and in the host build of the device core (rt_rng.h, RngTape), so that
everything that consumes random numbers can be compared value by value instead of in distribution.

The oracle draws in the reference's order and tags every draw (what it was for, the recursion level, the medium).  The
device core makes the same draws in another arrangement: the roulette draw of camera.rs:228 comes first in a bounce
and is addressed by the bounce, a medium's free-path draw (constant_medium.rs:74) is addressed by bounce and medium, a
list of one light draws no index (hittable_list.rs:94 does: the draw has one possible outcome).  `device_inputs` puts
the oracle's numbers where the device core looks for them; a place the oracle never drew for holds a random filler,
which the device core may read but which must not change its result."""
import ctypes as C

import numpy as np

from raytracing_rust_amd import _capi
import helpers as H

TAG_OTHER, TAG_CAMERA, TAG_SHADE, TAG_ROULETTE, TAG_SINGLE_INDEX, TAG_MEDIUM = 0, 1, 3, 4, 5, 6
PSZ = C.POINTER(C.c_size_t)
MAX_RAYS = 128

_bound = False


def _bind():
    global _bound
    if _bound:
        return
    o, s = H.oracle_lib(), H.hostsim_lib()
    o.orc_tape_path.restype = C.c_long
    o.orc_tape_path.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(_capi.CameraDesc), C.c_int, C.c_int, C.c_int, H.PD,
                                H.PD, C.c_size_t, H.PU32, PSZ, H.PD, C.c_size_t, PSZ, H.PD]
    o.orc_tape_vec.restype = C.c_long
    o.orc_tape_vec.argtypes = [C.c_int, H.PD, C.c_size_t, H.PD, PSZ]
    o.orc_tape_random.restype = C.c_long
    o.orc_tape_random.argtypes = [C.c_void_p, C.c_int, H.PD, H.PD, C.c_size_t, H.PU32, H.PD, PSZ]
    s.sim_scene_new.restype = C.c_void_p
    s.sim_scene_new.argtypes = [C.POINTER(_capi.SceneDesc)]
    s.sim_scene_free.argtypes = [C.c_void_p]
    s.sim_tape_path.restype = C.c_long
    s.sim_tape_path.argtypes = [C.c_void_p, C.POINTER(_capi.CameraDesc), C.c_int, C.c_int, C.c_int, C.c_int, H.PD,
                                H.PD, C.c_size_t, H.PD, C.c_uint32, H.PD, C.c_uint32, PSZ, H.PD, C.c_size_t, PSZ, H.PD, H.PU32]
    s.sim_tape_vec.restype = C.c_long
    s.sim_tape_vec.argtypes = [C.c_int, H.PD, C.c_size_t, H.PD, PSZ]
    s.sim_tape_lights_random.restype = C.c_long
    s.sim_tape_lights_random.argtypes = [C.POINTER(_capi.SceneDesc), H.PD, H.PD, C.c_size_t, H.PD, PSZ]
    _bound = True


def oracle_path(g, world, lights, cam, tape, pixel=None, ray=None):
    """ray_color of the oracle on the camera ray of pixel = (i, j, sample index) or on `ray` (7 doubles), random numbers
    from `tape`.  Returns dict(color, rays[n,7], tags[consumed], consumed)."""
    _bind()
    tape = np.ascontiguousarray(tape, dtype=np.float64)
    tags = np.zeros(len(tape), dtype=np.uint32)
    rays = np.zeros((MAX_RAYS, 7))
    color = np.zeros(3)
    consumed, n_rays = C.c_size_t(0), C.c_size_t(0)
    i, j, s = pixel if pixel is not None else (0, 0, 0)
    r7 = None if ray is None else H.dptr(np.ascontiguousarray(ray, dtype=np.float64))
    over = H.oracle_lib().orc_tape_path(g.session, world, -1 if lights is None else lights, C.byref(cam), i, j, s, r7,
                                        H.dptr(tape), len(tape), tags.ctypes.data_as(H.PU32), C.byref(consumed),
                                        H.dptr(rays), MAX_RAYS, C.byref(n_rays), H.dptr(color))
    assert over == 0, "the oracle asked for more numbers than the tape holds"
    assert n_rays.value <= MAX_RAYS
    return {"color": color, "rays": rays[:n_rays.value].copy(), "tags": tags[:consumed.value].copy(), "consumed": consumed.value}


def device_inputs(tape, tags, n_bounces, n_media, rng):
    """The oracle's numbers arranged the way the device core draws: (sequence of camera and shading draws, roulette
    table [n_bounces], medium table [n_bounces, n_media]); the places of draws the oracle did not make hold fillers."""
    kind, level, med = tags & 0xFF, (tags >> 8) & 0xFFF, tags >> 20
    v = np.asarray(tape[:len(tags)])
    seq = v[(kind == TAG_CAMERA) | (kind == TAG_SHADE) | (kind == TAG_OTHER)]
    roulette = rng.uniform(0, 1, n_bounces)
    medium = rng.uniform(0, 1, (n_bounces, max(n_media, 1)))
    for k in np.nonzero(kind == TAG_ROULETTE)[0]:
        roulette[level[k]] = v[k]
    for k in np.nonzero(kind == TAG_MEDIUM)[0]:
        medium[level[k], med[k]] = v[k]
    return np.ascontiguousarray(seq), roulette, np.ascontiguousarray(medium)


class DeviceScene:
    """The compiled program of a flattened description in the host build of the device core."""
    def __init__(self, desc, keep=None):
        _bind()
        self.desc, self.keep = desc, keep          # keep: the graph whose session owns the description's tables
        self.h = C.c_void_p(H.hostsim_lib().sim_scene_new(C.byref(desc)))

    def close(self):
        if self.h:
            H.hostsim_lib().sim_scene_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def path(self, cam, use_lights, seq, roulette, medium, pixel=None, ray=None):
        rays = np.zeros((MAX_RAYS, 7))
        color = np.zeros(3)
        consumed, n_rays = C.c_size_t(0), C.c_size_t(0)
        counters = np.zeros(2, dtype=np.uint32)
        i, j, s = pixel if pixel is not None else (0, 0, 0)
        r7 = None if ray is None else H.dptr(np.ascontiguousarray(ray, dtype=np.float64))
        over = H.hostsim_lib().sim_tape_path(self.h, C.byref(cam), 1 if use_lights else 0, i, j, s, r7, H.dptr(seq), len(seq),
                                             H.dptr(roulette), len(roulette), H.dptr(medium), medium.shape[1],
                                             C.byref(consumed), H.dptr(rays), MAX_RAYS, C.byref(n_rays), H.dptr(color),
                                             counters.ctypes.data_as(H.PU32))
        return {"color": color, "rays": rays[:n_rays.value].copy(), "consumed": consumed.value, "overrun": over,
                "segments": int(counters[0]), "probes": int(counters[1])}


def ulp_distance(a, b):
    """Distance of two float64 arrays in units in the last place (same-sign finite values)."""
    x, y = np.ascontiguousarray(a, dtype=np.float64).view(np.int64), np.ascontiguousarray(b, dtype=np.float64).view(np.int64)
    return np.abs(x - y)


def compare_path(go, wo, lo, dev, cam, rng, n_media, pixel=None, ray=None, tape_len=2048):
    """One path through both, same numbers.  Returns (oracle result, device result) after asserting that both trace the
    same rays bit for bit, consume the same sequence draws and return the same radiance to 1e-12."""
    tape = rng.uniform(0, 1, tape_len)
    a = oracle_path(go, wo, lo, cam, tape, pixel=pixel, ray=ray)
    seq, roulette, medium = device_inputs(tape, a["tags"], max(cam.max_depth, 1), n_media, rng)
    b = dev.path(cam, lo is not None, seq, roulette, medium, pixel=pixel, ray=ray)
    what = f"pixel {pixel} ray {ray}"
    assert b["overrun"] == 0, f"{what}: the device core asked for {b['overrun']} draws the oracle never made"
    assert len(a["rays"]) == len(b["rays"]), f"{what}: {len(a['rays'])} segments in the oracle, {len(b['rays'])} in the device core"
    same = a["rays"].view(np.uint64) == b["rays"].view(np.uint64)
    assert same.all(), f"{what}: segment {np.nonzero(~same.all(axis=1))[0][0]} differs:\n{a['rays'][~same.all(axis=1)][:1]}\n{b['rays'][~same.all(axis=1)][:1]}"
    # a path the roulette ends: the oracle sampled a direction first (camera.rs:208-231), the device core did not
    kind, level = a["tags"] & 0xFF, (a["tags"] >> 8) & 0xFFF
    last = len(a["rays"]) - 1
    ended = (kind == TAG_ROULETTE) & (level == last) & (tape[:len(kind)] > 0.8)
    expect = len(seq) - (int(((kind == TAG_SHADE) & (level == last)).sum()) if ended.any() else 0)
    assert b["consumed"] == expect, f"{what}: the device core made {b['consumed']} sequence draws, the oracle {expect}"
    scale = max(np.abs(a["color"]).max(), 1e-300)
    assert np.abs(a["color"] - b["color"]).max() <= 1e-12 * scale, f"{what}: radiance {a['color']} vs {b['color']}"
    return a, b
