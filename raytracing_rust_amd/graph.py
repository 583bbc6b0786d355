"""Recording scene-graph builder over the host layer's C constructors.

`Graph` is a thin, explicit wrapper: one method per constructor of
include/rtx_host.h, integer handles in and out.  It is written against a
(library, prefix) pair so that the test-suite can drive a second library that
exports the same constructor set under another prefix with the same calls.
"""
import ctypes as C

import numpy as np

from . import _capi

_CTOR_NAMES = [
    "session_new", "session_free", "tex_solid", "tex_checker", "tex_noise", "tex_noise_tables", "tex_image",
    "tex_image_file", "mat_lambertian", "mat_metal", "mat_dielectric", "mat_diffuse_light", "mat_isotropic",
    "mat_none", "sphere", "sphere_moving", "quad", "box", "list_new", "list_add", "bvh", "translate", "rotate_y",
    "medium", "bounding_box",
]


def bind_constructors(lib, prefix):
    """Set ctypes prototypes for `<prefix><name>` using the signatures of the rtxh_ set."""
    fns = {}
    for name in _CTOR_NAMES:
        res, args = _capi.SYMBOLS["rtxh_" + name]
        fn = getattr(lib, prefix + name)
        fn.restype, fn.argtypes = res, args
        fns[name] = fn
    return fns


def _d3(v):
    return (C.c_double * 3)(float(v[0]), float(v[1]), float(v[2]))


class GraphError(RuntimeError):
    pass


class Graph:
    def __init__(self, lib=None, prefix="rtxh_"):
        self.lib = _capi.lib if lib is None else lib
        self.prefix = prefix
        self.f = bind_constructors(self.lib, prefix)
        self.session = C.c_void_p(self.f["session_new"]())
        self._keep = []

    def use_context(self, ctx):
        """Let BvhNode of large lists be built on ctx's GPU (rtxh_session_set_context); None detaches.  The tree is the
        one the host build makes."""
        if self.prefix != "rtxh_":
            raise GraphError("only the product's scene layer builds on a GPU")
        self.lib.rtxh_session_set_context(self.session, ctx.handle if ctx is not None else None)
        self._ctx = ctx

    def close(self):
        if self.session:
            self.f["session_free"](self.session)
            self.session = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, h, what):
        if h < 0:
            raise GraphError(f"{what} failed")
        return h

    # textures
    def solid(self, r, g, b): return self._chk(self.f["tex_solid"](self.session, r, g, b), "tex_solid")
    def checker(self, scale, even, odd): return self._chk(self.f["tex_checker"](self.session, scale, even, odd), "tex_checker")
    def noise(self, scale, seed): return self._chk(self.f["tex_noise"](self.session, scale, seed), "tex_noise")

    def noise_tables(self, scale, ranvec, px, py, pz):
        rv = np.ascontiguousarray(ranvec, dtype=np.float64).reshape(768)
        pp = [np.ascontiguousarray(p, dtype=np.int32).reshape(256) for p in (px, py, pz)]
        return self._chk(self.f["tex_noise_tables"](
            self.session, scale, rv.ctypes.data_as(_capi.PD), *[p.ctypes.data_as(C.POINTER(C.c_int32)) for p in pp]),
            "tex_noise_tables")

    def image(self, rgb):
        """rgb: uint8 array (H, W, 3), or None for the missing-file texture."""
        if rgb is None:
            return self._chk(self.f["tex_image"](self.session, 0, 0, None), "tex_image")
        a = np.ascontiguousarray(rgb, dtype=np.uint8)
        h, w, _ = a.shape
        return self._chk(self.f["tex_image"](self.session, w, h, a.ctypes.data_as(C.POINTER(C.c_uint8))), "tex_image")

    def image_file(self, name): return self._chk(self.f["tex_image_file"](self.session, name.encode()), "tex_image_file")

    # materials
    def lambertian(self, tex): return self._chk(self.f["mat_lambertian"](self.session, tex), "mat_lambertian")
    def lambertian_color(self, r, g, b): return self.lambertian(self.solid(r, g, b))
    def metal(self, r, g, b, fuzz): return self._chk(self.f["mat_metal"](self.session, r, g, b, fuzz), "mat_metal")
    def dielectric(self, ior): return self._chk(self.f["mat_dielectric"](self.session, ior), "mat_dielectric")
    def diffuse_light(self, tex): return self._chk(self.f["mat_diffuse_light"](self.session, tex), "mat_diffuse_light")
    def diffuse_light_color(self, r, g, b): return self.diffuse_light(self.solid(r, g, b))
    def isotropic(self, tex): return self._chk(self.f["mat_isotropic"](self.session, tex), "mat_isotropic")
    def none(self): return self._chk(self.f["mat_none"](self.session), "mat_none")

    # hittables
    def sphere(self, c, r, mat): return self._chk(self.f["sphere"](self.session, _d3(c), r, mat), "sphere")
    def sphere_moving(self, c0, c1, r, mat): return self._chk(self.f["sphere_moving"](self.session, _d3(c0), _d3(c1), r, mat), "sphere_moving")
    def quad(self, q, u, v, mat): return self._chk(self.f["quad"](self.session, _d3(q), _d3(u), _d3(v), mat), "quad")
    def box(self, a, b, mat): return self._chk(self.f["box"](self.session, _d3(a), _d3(b), mat), "box")
    def list(self, items=()):
        h = self._chk(self.f["list_new"](self.session), "list_new")
        for it in items:
            self.add(h, it)
        return h
    def add(self, lst, obj): return self._chk(self.f["list_add"](self.session, lst, obj), "list_add")
    def bvh(self, lst): return self._chk(self.f["bvh"](self.session, lst), "bvh")
    def translate(self, obj, off): return self._chk(self.f["translate"](self.session, obj, _d3(off)), "translate")
    def rotate_y(self, obj, deg): return self._chk(self.f["rotate_y"](self.session, obj, deg), "rotate_y")
    def medium(self, boundary, density, tex): return self._chk(self.f["medium"](self.session, boundary, density, tex), "medium")

    def bounding_box(self, obj):
        out = (C.c_double * 6)()
        r = self.f["bounding_box"](self.session, obj, out)
        if r < 0:
            raise GraphError("bounding_box failed")
        return None if r == 0 else np.array(out[:], dtype=np.float64)
