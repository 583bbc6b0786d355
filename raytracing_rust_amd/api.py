"""Python mirror of the reference's scene API for this path.

Same names, argument meaning and behaviour as the reference's constructors
(src/ray_tracing/{geometry,materials,acceleration,volumes}/...), so scene code
reads like the reference's scenes/*.rs.  Objects record themselves into a host
session (include/rtx_host.h); `Camera.render(world, lights)` flattens the graph
and runs the GPU path tracer, like camera.rs:271.

    world = HittableList()
    red = Lambertian.new(Color(0.65, 0.05, 0.05))
    world.add(Quad.new(Point3(555, 0, 0), Vec3(0, 555, 0), Vec3(0, 0, 555), red))
    cam = Camera(); cam.image_width = 600; ...
    image = cam.render(world, lights)
"""
import ctypes as C
import time

import numpy as np

from . import _capi, render as _r
from .graph import Graph

_graph = None
_context = None


def session():
    """The process-wide recording session (the reference keeps its graph in Arc pointers instead)."""
    global _graph
    if _graph is None:
        _graph = Graph()
    return _graph


def reset_session():
    global _graph
    if _graph is not None:
        _graph.close()
    _graph = None


def context():
    global _context
    if _context is None:
        _context = _r.Context(0)
    return _context


def Vec3(x, y, z):
    return np.array([x, y, z], dtype=np.float64)


Point3 = Vec3
Color = Vec3


class _Handle:
    def __init__(self, h):
        self.h = h


# ---- textures (materials/texture/*.rs)
class SolidColor(_Handle):
    @classmethod
    def new(cls, albedo): return cls(session().solid(*albedo))


class CheckerTexture(_Handle):
    @classmethod
    def new(cls, scale, even, odd): return cls(session().checker(scale, even.h, odd.h))

    @classmethod
    def new_colors(cls, scale, c1, c2): return cls.new(scale, SolidColor.new(c1), SolidColor.new(c2))

    @classmethod
    def new_rgb(cls, scale, r1, g1, b1, r2, g2, b2): return cls.new_colors(scale, Color(r1, g1, b1), Color(r2, g2, b2))


class NoiseTexture(_Handle):
    @classmethod
    def new(cls, scale, seed=0x5EED0001): return cls(session().noise(scale, seed))

    @classmethod
    def new_with_noise(cls, tables, scale):
        """tables = (ranvec[256,3], perm_x, perm_y, perm_z), the fields of procedural/noise.rs:6-11."""
        return cls(session().noise_tables(scale, *tables))


class ImageTexture(_Handle):
    @classmethod
    def new(cls, image_filename): return cls(session().image_file(image_filename))

    @classmethod
    def from_pixels(cls, rgb): return cls(session().image(rgb))


# ---- materials (materials/*.rs)
class Lambertian(_Handle):
    @classmethod
    def new(cls, albedo): return cls(session().lambertian_color(*albedo))

    @classmethod
    def new_texture(cls, tex): return cls(session().lambertian(tex.h))


class Metal(_Handle):
    @classmethod
    def new(cls, albedo, fuzz): return cls(session().metal(albedo[0], albedo[1], albedo[2], fuzz))


class Dielectric(_Handle):
    @classmethod
    def new(cls, refraction_index): return cls(session().dielectric(refraction_index))


class DiffuseLight(_Handle):
    @classmethod
    def new(cls, emit): return cls(session().diffuse_light(emit.h))

    @classmethod
    def new_color(cls, color): return cls(session().diffuse_light_color(*color))


class Isotropic(_Handle):
    @classmethod
    def new(cls, tex): return cls(session().isotropic(tex.h))

    @classmethod
    def new_color(cls, color): return cls(session().isotropic(session().solid(*color)))


class NoMaterial(_Handle):
    def __init__(self):
        super().__init__(session().none())


# ---- hittables (geometry/*.rs, acceleration/bvh.rs, volumes/constant_medium.rs)
class Hittable(_Handle):
    def bounding_box(self):
        return session().bounding_box(self.h)


class Sphere(Hittable):
    @classmethod
    def new(cls, center, radius, mat): return cls(session().sphere(center, radius, mat.h))

    @classmethod
    def new_moving(cls, center0, center1, radius, mat): return cls(session().sphere_moving(center0, center1, radius, mat.h))


class Quad(Hittable):
    @classmethod
    def new(cls, q, u, v, mat): return cls(session().quad(q, u, v, mat.h))


class HittableList(Hittable):
    def __init__(self, h=None):
        super().__init__(session().list() if h is None else h)
        self.objects = []

    @classmethod
    def new(cls): return cls()

    def add(self, obj):
        session().add(self.h, obj.h)
        self.objects.append(obj)

    def __len__(self): return len(self.objects)

    def is_empty(self): return not self.objects


def box_new(a, b, mat):
    return HittableList(session().box(a, b, mat.h))


class BvhNode(Hittable):
    @classmethod
    def new(cls, lst): return cls(session().bvh(lst.h))


class Translate(Hittable):
    @classmethod
    def new(cls, obj, offset): return cls(session().translate(obj.h, offset))


class RotateY(Hittable):
    @classmethod
    def new(cls, obj, angle): return cls(session().rotate_y(obj.h, angle))


class ConstantMedium(Hittable):
    @classmethod
    def new(cls, boundary, density, texture): return cls(session().medium(boundary.h, density, texture.h))

    @classmethod
    def new_color(cls, boundary, density, color): return cls(session().medium(boundary.h, density, session().solid(*color)))


def _failure_message():
    """rtxh_camera_render clears the messages of both layers when it starts, so whichever is set afterwards belongs to
    the step that failed: the device layer's (the context's, or the context-less one of the camera checks), else the
    session's (flattening)."""
    for msg in (_capi.lib.rtx_last_error(context().handle), _capi.lib.rtx_last_error(None),
                _capi.lib.rtxh_last_error(session().session)):
        if msg:
            return msg.decode()
    return ""


class Camera:
    """rendering/camera.rs:16-49; `render` is camera.rs:271-328 with the pixel loops on the GPU."""

    def __init__(self):
        self.aspect_ratio = 1.0
        self.image_width = 100
        self.samples_per_pixel = 10
        self.max_depth = 10
        self.background = Color(0.7, 0.8, 1.0)
        self.output_filename = "output.png"
        self.vfov = 90.0
        self.lookfrom = Point3(0, 0, 0)
        self.lookat = Point3(0, 0, -1)
        self.vup = Vec3(0, 1, 0)
        self.defocus_angle = 0.0
        self.focus_dist = 10.0
        self.seed = 0x5EED0002
        self.last_render_seconds = None

    @classmethod
    def new(cls): return cls()

    def desc(self):
        c = _r.camera_desc(aspect_ratio=self.aspect_ratio, image_width=self.image_width,
                           samples_per_pixel=self.samples_per_pixel, max_depth=self.max_depth,
                           background=self.background, vfov=self.vfov, lookfrom=self.lookfrom, lookat=self.lookat,
                           vup=self.vup, defocus_angle=self.defocus_angle, focus_dist=self.focus_dist)
        return _r.initialize_camera(c)

    def render(self, world, lights=None, save=True):
        cam = self.desc()
        rgb = np.empty((cam.image_height, cam.image_width, 3), dtype=np.uint8)
        secs = C.c_double(0.0)
        fn = self.output_filename.encode() if (save and self.output_filename) else None
        st = _capi.lib.rtxh_camera_render(session().session, context().handle, C.byref(cam), world.h,
                                          -1 if lights is None else lights.h, self.seed, fn,
                                          rgb.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(secs))
        if st != _capi.RTX_OK:
            raise _r.RtxError(st, _failure_message())
        self.last_render_seconds = secs.value
        return rgb

    def render_progressive(self, world, lights=None, rows_per_batch=1, checkpoint=None, progress=None, save=True):
        """The same render in batches of rows of the stratification, with a checkpoint of the accumulator after every
        batch (resumed if `checkpoint` already holds one of this render) and `progress(rows_done, rows_total, seconds)`
        after every batch; a true return value stops the render.  Returns (rgb, rows_done)."""
        cam = self.desc()
        rgb = np.empty((cam.image_height, cam.image_width, 3), dtype=np.uint8)
        done = C.c_int(0)
        cb = _capi.PROGRESS_FN((lambda r, n, s, u: 1 if progress(r, n, s) else 0) if progress else (lambda r, n, s, u: 0))
        fn = self.output_filename.encode() if (save and self.output_filename) else None
        st = _capi.lib.rtxh_camera_render_progressive(
            session().session, context().handle, C.byref(cam), world.h, -1 if lights is None else lights.h, self.seed,
            int(rows_per_batch), str(checkpoint).encode() if checkpoint else None, cb, None, fn,
            rgb.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(done))
        if st != _capi.RTX_OK:
            raise _r.RtxError(st, _failure_message())
        return rgb, done.value
