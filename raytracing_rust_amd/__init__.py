"""MI355X-native path-tracing backend for the reference's `ray_color` hot path.

The compute lives in librtx_amd.so (HIP kernels for gfx950 behind the C ABI of
include/rtx.h and include/rtx_host.h).  This package is the host-side mirror of
the reference's scene API plus thin ctypes plumbing; importing it fails when
the library has not been built.
"""
from . import _capi  # noqa: F401  (raises ImportError when librtx_amd.so is missing)
from .graph import Graph, GraphError  # noqa: F401
from .render import (Context, RtxError, Scene, camera_desc, flatten, initialize_camera,  # noqa: F401
                     pixel_center_rays, render_opts, write_png)
from . import scenes  # noqa: F401
