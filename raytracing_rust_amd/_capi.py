"""ctypes view of the C ABI declared in include/rtx.h and include/rtx_host.h.

The shared library is the product: if librtx_amd.so is missing this module
raises at import time (there is no Python or CPU substitute for the kernels).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTX_AMD_LIB") or os.path.join(_HERE, "librtx_amd.so")  # RTX_AMD_LIB: a tuning build

u32, u64, i32, f64, u8 = C.c_uint32, C.c_uint64, C.c_int32, C.c_double, C.c_uint8
f64x3 = f64 * 3

RTX_OK, RTX_ERR_INVALID, RTX_ERR_NO_DEVICE, RTX_ERR_HIP, RTX_ERR_UNSUPPORTED, RTX_ERR_IO = range(6)
NODE_BOX, NODE_SPHERE, NODE_QUAD, NODE_ENTER, NODE_EXIT, NODE_MEDIUM, NODE_NONE = 0, 1, 2, 3, 4, 5, 15
PIPELINE_DEFAULT, PIPELINE_FUSED, PIPELINE_WAVEFRONT = 0, 1, 2
NOALLOC_MAX_PATHS = 65536


class Node(C.Structure):
    _fields_ = [("kind_idx", u32), ("skip", u32)]


class Aabb(C.Structure):
    _fields_ = [("lo", f64x3), ("hi", f64x3)]


class Sphere(C.Structure):
    _fields_ = [("c0", f64x3), ("dc", f64x3), ("radius", f64), ("mat", u32), ("prim_id", u32)]


class Quad(C.Structure):
    _fields_ = [("q", f64x3), ("u", f64x3), ("v", f64x3), ("normal", f64x3), ("w", f64x3), ("d", f64), ("area", f64),
                ("mat", u32), ("prim_id", u32)]


class XformOp(C.Structure):
    _fields_ = [("kind", u32), ("_pad", u32), ("a", f64), ("b", f64), ("c", f64)]


class Xform(C.Structure):
    _fields_ = [("first_op", u32), ("n_ops", u32)]


class Medium(C.Structure):
    _fields_ = [("b_begin", u32), ("b_end", u32), ("mat", u32), ("needs_uv", u32), ("neg_inv_density", f64)]


class Material(C.Structure):
    _fields_ = [("kind", u32), ("tex", u32), ("albedo", f64x3), ("param", f64), ("needs_uv", u32), ("_pad", u32)]


class Texture(C.Structure):
    _fields_ = [("kind", u32), ("a", u32), ("b", u32), ("_pad", u32), ("color", f64x3), ("scale", f64)]


class Image(C.Structure):
    _fields_ = [("width", u32), ("height", u32), ("offset", u64)]


class Perlin(C.Structure):
    _fields_ = [("ranvec", f64 * 768), ("perm_x", i32 * 256), ("perm_y", i32 * 256), ("perm_z", i32 * 256)]


class Light(C.Structure):
    _fields_ = [("kind", u32), ("idx", u32), ("ctx", u32), ("_pad", u32), ("weight", f64), ("cum", f64)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("n_nodes", u32), ("nodes", C.POINTER(Node)),
        ("world_begin", u32), ("world_end", u32),
        ("n_boxes", u32), ("boxes", C.POINTER(Aabb)),
        ("n_spheres", u32), ("spheres", C.POINTER(Sphere)),
        ("n_quads", u32), ("quads", C.POINTER(Quad)),
        ("n_xforms", u32), ("xforms", C.POINTER(Xform)),
        ("n_xform_ops", u32), ("xform_ops", C.POINTER(XformOp)),
        ("n_media", u32), ("media", C.POINTER(Medium)),
        ("n_materials", u32), ("materials", C.POINTER(Material)),
        ("n_textures", u32), ("textures", C.POINTER(Texture)),
        ("n_images", u32), ("images", C.POINTER(Image)),
        ("n_texel_bytes", u64), ("texels", C.POINTER(u8)),
        ("n_perlins", u32), ("perlins", C.POINTER(Perlin)),
        ("n_lights", u32), ("lights", C.POINTER(Light)),
        ("has_lights", u32), ("_pad", u32),
    ]


class CameraDesc(C.Structure):
    """camera.rs:18-48: public fields, then the fields Camera::initialize derives."""
    _fields_ = [
        ("aspect_ratio", f64), ("image_width", i32), ("samples_per_pixel", i32), ("max_depth", i32), ("_pad0", i32),
        ("background", f64x3), ("vfov", f64), ("lookfrom", f64x3), ("lookat", f64x3), ("vup", f64x3),
        ("defocus_angle", f64), ("focus_dist", f64),
        ("image_height", i32), ("sqrt_spp", i32), ("pixel_samples_scale", f64), ("recip_sqrt_spp", f64),
        ("center", f64x3), ("pixel00_loc", f64x3), ("pixel_delta_u", f64x3), ("pixel_delta_v", f64x3),
        ("u", f64x3), ("v", f64x3), ("w", f64x3), ("defocus_disk_u", f64x3), ("defocus_disk_v", f64x3),
    ]


class RenderOpts(C.Structure):
    _fields_ = [("sample_begin", i32), ("sample_end", i32), ("x0", i32), ("y0", i32), ("x1", i32), ("y1", i32),
                ("seed", u64), ("use_lights", i32), ("pipeline", i32)]


class Stats(C.Structure):
    _fields_ = [("paths", u64), ("segments", u64), ("pdf_probes", u64), ("render_ms", f64), ("launches", u64),
                ("last_kernel_ms", f64), ("traverse_ms", f64), ("traverse_launches", u64), ("tail_ms", f64),
                ("tail_launches", u64), ("head_ms", f64), ("head_launches", u64), ("shade_ms", f64), ("shade_launches", u64),
                ("roulette_drops", u64), ("order_redos", u64), ("iterations", u64)]


PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p)

# Every symbol the two headers declare: name -> (restype, argtypes).
P = C.c_void_p
PD = C.POINTER(f64)
SYMBOLS = {
    # include/rtx.h
    "rtx_context_create": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(P)]),
    "rtx_context_destroy": (None, [P]),
    "rtx_last_error": (C.c_char_p, [P]),
    "rtx_scene_create": (C.c_int, [P, C.POINTER(SceneDesc), C.POINTER(P)]),
    "rtx_scene_destroy": (None, [P]),
    "rtx_camera_initialize": (C.c_int, [C.POINTER(CameraDesc)]),
    "rtx_render": (C.c_int, [P, C.POINTER(CameraDesc), C.POINTER(RenderOpts), PD]),
    "rtx_render_device": (C.c_int, [P, C.POINTER(CameraDesc), C.POINTER(RenderOpts), P, P]),
    "rtx_render_device_noalloc": (C.c_int, [P, C.POINTER(CameraDesc), C.POINTER(RenderOpts), P, P]),
    "rtx_resolve_rgb8": (C.c_int, [P, PD, C.c_int, C.c_int, C.c_int, C.POINTER(u8)]),
    "rtx_resolve_rgb8_device": (C.c_int, [P, P, C.c_int, C.c_int, C.c_int, P, P]),
    "rtx_first_hit_aov": (C.c_int, [P, PD, C.c_size_t, PD, PD, C.POINTER(u32), C.POINTER(u32), C.POINTER(u8)]),
    "rtx_bvh_leaf_order": (C.c_int, [P, u32, PD, C.POINTER(u32)]),
    "rtx_get_stats": (C.c_int, [P, C.POINTER(Stats)]),
    "rtx_reset_stats": (C.c_int, [P]),
    # include/rtx_host.h
    "rtxh_session_new": (P, []),
    "rtxh_session_free": (None, [P]),
    "rtxh_session_set_context": (None, [P, P]),
    "rtxh_last_error": (C.c_char_p, [P]),
    "rtxh_tex_solid": (C.c_int, [P, f64, f64, f64]),
    "rtxh_tex_checker": (C.c_int, [P, f64, C.c_int, C.c_int]),
    "rtxh_tex_noise": (C.c_int, [P, f64, u64]),
    "rtxh_tex_noise_tables": (C.c_int, [P, f64, PD, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "rtxh_tex_image": (C.c_int, [P, u32, u32, C.POINTER(u8)]),
    "rtxh_tex_image_file": (C.c_int, [P, C.c_char_p]),
    "rtxh_mat_lambertian": (C.c_int, [P, C.c_int]),
    "rtxh_mat_metal": (C.c_int, [P, f64, f64, f64, f64]),
    "rtxh_mat_dielectric": (C.c_int, [P, f64]),
    "rtxh_mat_diffuse_light": (C.c_int, [P, C.c_int]),
    "rtxh_mat_isotropic": (C.c_int, [P, C.c_int]),
    "rtxh_mat_none": (C.c_int, [P]),
    "rtxh_sphere": (C.c_int, [P, PD, f64, C.c_int]),
    "rtxh_sphere_moving": (C.c_int, [P, PD, PD, f64, C.c_int]),
    "rtxh_quad": (C.c_int, [P, PD, PD, PD, C.c_int]),
    "rtxh_box": (C.c_int, [P, PD, PD, C.c_int]),
    "rtxh_list_new": (C.c_int, [P]),
    "rtxh_list_add": (C.c_int, [P, C.c_int, C.c_int]),
    "rtxh_bvh": (C.c_int, [P, C.c_int]),
    "rtxh_translate": (C.c_int, [P, C.c_int, PD]),
    "rtxh_rotate_y": (C.c_int, [P, C.c_int, f64]),
    "rtxh_medium": (C.c_int, [P, C.c_int, f64, C.c_int]),
    "rtxh_bounding_box": (C.c_int, [P, C.c_int, PD]),
    "rtxh_flatten": (C.c_int, [P, C.c_int, C.c_int, C.POINTER(SceneDesc)]),
    "rtxh_scene_cornell": (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(CameraDesc)]),
    "rtxh_scene_cornell_smoke": (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(CameraDesc)]),
    "rtxh_scene_final": (C.c_int, [P, C.c_int, C.c_int, C.c_int, u64, C.c_int, C.POINTER(C.c_int), C.POINTER(CameraDesc)]),
    "rtxh_scene_stress": (C.c_int, [P, C.c_int, u64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(CameraDesc)]),
    "rtxh_camera_render": (C.c_int, [P, P, C.POINTER(CameraDesc), C.c_int, C.c_int, u64, C.c_char_p, C.POINTER(u8), PD]),
    "rtxh_camera_render_progressive": (C.c_int, [P, P, C.POINTER(CameraDesc), C.c_int, C.c_int, u64, C.c_int, C.c_char_p,
                                                 PROGRESS_FN, P, C.c_char_p, C.POINTER(u8), C.POINTER(C.c_int)]),
    "rtxh_checkpoint_fingerprint": (u64, [C.POINTER(SceneDesc), C.POINTER(CameraDesc)]),
    "rtxh_write_png": (C.c_int, [C.c_char_p, C.POINTER(u8), C.c_int, C.c_int]),
}


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no fallback implementation.")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError here = the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load_library()
