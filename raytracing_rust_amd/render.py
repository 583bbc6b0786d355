"""Device-layer wrappers over include/rtx.h: context, uploaded scene, render,
first-hit AOV, resolve, statistics.  numpy arrays cross the boundary as host
pointers; torch CUDA tensors cross as device pointers (rtx_render_device)."""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import lib


class RtxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"rtx status {status}: {message}")
        self.status = status


def _check(status, ctx=None):
    if status != _capi.RTX_OK:
        msg = lib.rtx_last_error(ctx)
        raise RtxError(status, msg.decode() if msg else "")


def camera_desc(**kw):
    """Camera::new defaults (camera.rs:54-86) with overrides; call initialize_camera afterwards."""
    c = _capi.CameraDesc()
    c.aspect_ratio, c.image_width, c.samples_per_pixel, c.max_depth = 1.0, 100, 10, 10
    c.background[:] = (0.7, 0.8, 1.0)
    c.vfov = 90.0
    c.lookfrom[:] = (0.0, 0.0, 0.0)
    c.lookat[:] = (0.0, 0.0, -1.0)
    c.vup[:] = (0.0, 1.0, 0.0)
    c.defocus_angle, c.focus_dist = 0.0, 10.0
    for k, v in kw.items():
        if k in ("background", "lookfrom", "lookat", "vup"):
            getattr(c, k)[:] = tuple(float(x) for x in v)
        else:
            setattr(c, k, v)
    return c


def initialize_camera(cam):
    _check(lib.rtx_camera_initialize(C.byref(cam)))
    return cam


def render_opts(cam, sample_begin=0, sample_end=None, rect=None, seed=0x5EED0002, use_lights=True,
                pipeline=_capi.PIPELINE_DEFAULT):
    o = _capi.RenderOpts()
    o.sample_begin = sample_begin
    o.sample_end = cam.sqrt_spp * cam.sqrt_spp if sample_end is None else sample_end
    o.x0, o.y0, o.x1, o.y1 = rect if rect is not None else (0, 0, cam.image_width, cam.image_height)
    o.seed, o.use_lights, o.pipeline = seed, 1 if use_lights else 0, pipeline
    return o


class Context:
    """One per process and GPU (rtx_context_create)."""

    def __init__(self, device=0):
        self.handle = C.c_void_p()
        ids = (C.c_int * 1)(device)
        _check(lib.rtx_context_create(1, ids, C.byref(self.handle)))

    def close(self):
        if self.handle:
            lib.rtx_context_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stats(self):
        s = _capi.Stats()
        _check(lib.rtx_get_stats(self.handle, C.byref(s)), self.handle)
        return {k: getattr(s, k) for k, _ in _capi.Stats._fields_}

    def reset_stats(self):
        _check(lib.rtx_reset_stats(self.handle), self.handle)

    def resolve_rgb8(self, accum, samples_per_pixel):
        a = np.ascontiguousarray(accum, dtype=np.float64)
        h, w, _ = a.shape
        out = np.empty((h, w, 3), dtype=np.uint8)
        _check(lib.rtx_resolve_rgb8(self.handle, a.ctypes.data_as(_capi.PD), w, h, samples_per_pixel,
                                    out.ctypes.data_as(C.POINTER(C.c_uint8))), self.handle)
        return out


def flatten(graph, world, lights=None):
    """rtxh_flatten: returns an rtx_scene_desc whose arrays live in the graph's session."""
    desc = _capi.SceneDesc()
    st = lib.rtxh_flatten(graph.session, world, -1 if lights is None else lights, C.byref(desc))
    if st != _capi.RTX_OK:
        raise RtxError(st, lib.rtxh_last_error(graph.session).decode())
    return desc


class Scene:
    """A scene uploaded to the GPU (rtx_scene_create)."""

    def __init__(self, ctx, desc):
        self.ctx = ctx
        self.handle = C.c_void_p()
        _check(lib.rtx_scene_create(ctx.handle, C.byref(desc), C.byref(self.handle)), ctx.handle)

    @classmethod
    def from_graph(cls, ctx, graph, world, lights=None):
        return cls(ctx, flatten(graph, world, lights))

    def close(self):
        if self.handle:
            lib.rtx_scene_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, cam, opts=None, accum=None):
        """Host accumulator (numpy, H x W x 3 float64); sums are ADDED into it."""
        if opts is None:
            opts = render_opts(cam)
        if accum is None:
            accum = np.zeros((cam.image_height, cam.image_width, 3), dtype=np.float64)
        assert accum.dtype == np.float64 and accum.flags.c_contiguous
        assert accum.shape == (cam.image_height, cam.image_width, 3)
        _check(lib.rtx_render(self.handle, C.byref(cam), C.byref(opts), accum.ctypes.data_as(_capi.PD)), self.ctx.handle)
        return accum

    @staticmethod
    def _device_pointer(cam, accum):
        """accum: a contiguous float64 CUDA tensor of H*W*3 elements, or a raw device address (int / c_void_p)."""
        if hasattr(accum, "data_ptr"):
            assert accum.is_cuda and accum.is_contiguous() and accum.element_size() == 8
            assert accum.numel() == cam.image_height * cam.image_width * 3
            return C.c_void_p(accum.data_ptr())
        return accum if isinstance(accum, C.c_void_p) else C.c_void_p(int(accum))

    def render_device(self, cam, opts, accum, stream=None):
        """Device accumulator (rtx_render_device).  The result is complete once `stream` has been synchronised; the
        call itself keeps the host busy for most of the render."""
        _check(lib.rtx_render_device(self.handle, C.byref(cam), C.byref(opts), self._device_pointer(cam, accum),
                                     C.c_void_p(stream) if stream else None), self.ctx.handle)

    def render_device_noalloc(self, cam, opts, accum, stream=None):
        """rtx_render_device_noalloc: at most NOALLOC_MAX_PATHS camera paths per call, nothing allocated, never blocks."""
        _check(lib.rtx_render_device_noalloc(self.handle, C.byref(cam), C.byref(opts), self._device_pointer(cam, accum),
                                             C.c_void_p(stream) if stream else None), self.ctx.handle)

    def first_hit(self, rays):
        """rays: (n, 7) float64 = origin, direction, time.  Returns dict of t, normal, prim_id, mat_id, front."""
        r = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 7)
        n = r.shape[0]
        out = {"t": np.empty(n), "normal": np.empty((n, 3)), "prim_id": np.empty(n, dtype=np.uint32),
               "mat_id": np.empty(n, dtype=np.uint32), "front": np.empty(n, dtype=np.uint8)}
        _check(lib.rtx_first_hit_aov(self.handle, r.ctypes.data_as(_capi.PD), n, out["t"].ctypes.data_as(_capi.PD),
                                     out["normal"].ctypes.data_as(_capi.PD),
                                     out["prim_id"].ctypes.data_as(C.POINTER(C.c_uint32)),
                                     out["mat_id"].ctypes.data_as(C.POINTER(C.c_uint32)),
                                     out["front"].ctypes.data_as(C.POINTER(C.c_uint8))), self.ctx.handle)
        return out


def pixel_center_rays(cam, time=0.0):
    """Deterministic camera rays through pixel centres (no jitter): (H*W, 7)."""
    jj, ii = np.meshgrid(np.arange(cam.image_height), np.arange(cam.image_width), indexing="ij")
    p00, du, dv = (np.array(x[:]) for x in (cam.pixel00_loc, cam.pixel_delta_u, cam.pixel_delta_v))
    center = np.array(cam.center[:])
    target = p00 + ii[..., None] * du + jj[..., None] * dv
    rays = np.empty((cam.image_height, cam.image_width, 7))
    rays[..., 0:3] = center
    rays[..., 3:6] = target - center
    rays[..., 6] = time
    return rays.reshape(-1, 7)


def write_png(path, rgb):
    a = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = a.shape
    _check(lib.rtxh_write_png(str(path).encode(), a.ctypes.data_as(C.POINTER(C.c_uint8)), w, h))
