//! Raw declarations of the backend's C ABI.  One item per declaration of include/rtx.h and include/rtx_host.h, same
//! names, same order of fields; tests/test_abi.py pins the C side of these layouts.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub type rtx_status = c_int;
pub const RTX_OK: rtx_status = 0;
pub const RTX_ERR_INVALID: rtx_status = 1;
pub const RTX_ERR_NO_DEVICE: rtx_status = 2;
pub const RTX_ERR_HIP: rtx_status = 3;
pub const RTX_ERR_UNSUPPORTED: rtx_status = 4;
pub const RTX_ERR_IO: rtx_status = 5;
pub const RTX_NOALLOC_MAX_PATHS: u32 = 65536;

#[repr(C)] pub struct rtx_context { _p: [u8; 0] }
#[repr(C)] pub struct rtx_scene { _p: [u8; 0] }
#[repr(C)] pub struct rtxh_session { _p: [u8; 0] }

/// camera.rs:18-33 (public fields) followed by camera.rs:36-48 (what Camera::initialize derives).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct rtx_camera_desc {
    pub aspect_ratio: f64,
    pub image_width: i32, pub samples_per_pixel: i32, pub max_depth: i32, pub _pad0: i32,
    pub background: [f64; 3],
    pub vfov: f64, pub lookfrom: [f64; 3], pub lookat: [f64; 3], pub vup: [f64; 3],
    pub defocus_angle: f64, pub focus_dist: f64,
    pub image_height: i32, pub sqrt_spp: i32,
    pub pixel_samples_scale: f64, pub recip_sqrt_spp: f64,
    pub center: [f64; 3], pub pixel00_loc: [f64; 3], pub pixel_delta_u: [f64; 3], pub pixel_delta_v: [f64; 3],
    pub u: [f64; 3], pub v: [f64; 3], pub w: [f64; 3], pub defocus_disk_u: [f64; 3], pub defocus_disk_v: [f64; 3],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rtx_render_opts {
    pub sample_begin: i32, pub sample_end: i32,
    pub x0: i32, pub y0: i32, pub x1: i32, pub y1: i32,
    pub seed: u64,
    pub use_lights: i32, pub pipeline: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rtx_stats {
    pub paths: u64, pub segments: u64, pub pdf_probes: u64,
    pub render_ms: f64, pub launches: u64, pub last_kernel_ms: f64,
    pub traverse_ms: f64, pub traverse_launches: u64,
    pub tail_ms: f64, pub tail_launches: u64,
    pub head_ms: f64, pub head_launches: u64,
    pub shade_ms: f64, pub shade_launches: u64,
    pub roulette_drops: u64, pub order_redos: u64, pub iterations: u64,
}

/// The flat scene description; produced by rtxh_flatten, consumed by rtx_scene_create.  A binding only passes it on,
/// so the tables are left as raw pointers (their element types are in include/rtx.h).
#[repr(C)]
pub struct rtx_scene_desc {
    pub n_nodes: u32, pub nodes: *const c_void,
    pub world_begin: u32, pub world_end: u32,
    pub n_boxes: u32, pub boxes: *const c_void,
    pub n_spheres: u32, pub spheres: *const c_void,
    pub n_quads: u32, pub quads: *const c_void,
    pub n_xforms: u32, pub xforms: *const c_void,
    pub n_xform_ops: u32, pub xform_ops: *const c_void,
    pub n_media: u32, pub media: *const c_void,
    pub n_materials: u32, pub materials: *const c_void,
    pub n_textures: u32, pub textures: *const c_void,
    pub n_images: u32, pub images: *const c_void,
    pub n_texel_bytes: u64, pub texels: *const u8,
    pub n_perlins: u32, pub perlins: *const c_void,
    pub n_lights: u32, pub lights: *const c_void,
    pub has_lights: u32, pub _pad: u32,
}

pub type rtxh_progress_fn = Option<unsafe extern "C" fn(rows_done: c_int, rows_total: c_int, seconds: f64, user: *mut c_void) -> c_int>;

extern "C" {
    // ---- include/rtx.h
    pub fn rtx_context_create(n_devices: c_int, device_ids: *const c_int, out: *mut *mut rtx_context) -> rtx_status;
    pub fn rtx_context_destroy(ctx: *mut rtx_context);
    pub fn rtx_last_error(ctx: *const rtx_context) -> *const c_char;
    pub fn rtx_scene_create(ctx: *mut rtx_context, desc: *const rtx_scene_desc, out: *mut *mut rtx_scene) -> rtx_status;
    pub fn rtx_scene_destroy(scene: *mut rtx_scene);
    pub fn rtx_camera_initialize(cam: *mut rtx_camera_desc) -> rtx_status;
    pub fn rtx_render(scene: *mut rtx_scene, cam: *const rtx_camera_desc, opts: *const rtx_render_opts, accum_host: *mut f64) -> rtx_status;
    pub fn rtx_render_device(scene: *mut rtx_scene, cam: *const rtx_camera_desc, opts: *const rtx_render_opts, accum_dev: *mut f64, hip_stream: *mut c_void) -> rtx_status;
    pub fn rtx_render_device_noalloc(scene: *mut rtx_scene, cam: *const rtx_camera_desc, opts: *const rtx_render_opts, accum_dev: *mut f64, hip_stream: *mut c_void) -> rtx_status;
    pub fn rtx_resolve_rgb8(ctx: *mut rtx_context, accum_host: *const f64, width: c_int, height: c_int, samples_per_pixel: c_int, rgb_host: *mut u8) -> rtx_status;
    pub fn rtx_resolve_rgb8_device(ctx: *mut rtx_context, accum_dev: *const f64, width: c_int, height: c_int, samples_per_pixel: c_int, rgb_dev: *mut u8, hip_stream: *mut c_void) -> rtx_status;
    pub fn rtx_first_hit_aov(scene: *mut rtx_scene, rays: *const f64, n: usize, t: *mut f64, normal: *mut f64, prim_id: *mut u32, mat_id: *mut u32, front: *mut u8) -> rtx_status;
    pub fn rtx_bvh_leaf_order(ctx: *mut rtx_context, n: u32, boxes: *const f64, order_out: *mut u32) -> rtx_status;
    pub fn rtx_get_stats(ctx: *mut rtx_context, out: *mut rtx_stats) -> rtx_status;
    pub fn rtx_reset_stats(ctx: *mut rtx_context) -> rtx_status;

    // ---- include/rtx_host.h
    pub fn rtxh_session_new() -> *mut rtxh_session;
    pub fn rtxh_session_free(s: *mut rtxh_session);
    pub fn rtxh_session_set_context(s: *mut rtxh_session, ctx: *mut rtx_context);
    pub fn rtxh_last_error(s: *const rtxh_session) -> *const c_char;
    pub fn rtxh_tex_solid(s: *mut rtxh_session, r: f64, g: f64, b: f64) -> c_int;
    pub fn rtxh_tex_checker(s: *mut rtxh_session, scale: f64, even: c_int, odd: c_int) -> c_int;
    pub fn rtxh_tex_noise(s: *mut rtxh_session, scale: f64, seed: u64) -> c_int;
    pub fn rtxh_tex_noise_tables(s: *mut rtxh_session, scale: f64, ranvec768: *const f64, perm_x256: *const i32, perm_y256: *const i32, perm_z256: *const i32) -> c_int;
    pub fn rtxh_tex_image(s: *mut rtxh_session, width: u32, height: u32, rgb: *const u8) -> c_int;
    pub fn rtxh_tex_image_file(s: *mut rtxh_session, filename: *const c_char) -> c_int;
    pub fn rtxh_mat_lambertian(s: *mut rtxh_session, tex: c_int) -> c_int;
    pub fn rtxh_mat_metal(s: *mut rtxh_session, r: f64, g: f64, b: f64, fuzz: f64) -> c_int;
    pub fn rtxh_mat_dielectric(s: *mut rtxh_session, refraction_index: f64) -> c_int;
    pub fn rtxh_mat_diffuse_light(s: *mut rtxh_session, tex: c_int) -> c_int;
    pub fn rtxh_mat_isotropic(s: *mut rtxh_session, tex: c_int) -> c_int;
    pub fn rtxh_mat_none(s: *mut rtxh_session) -> c_int;
    pub fn rtxh_sphere(s: *mut rtxh_session, center3: *const f64, radius: f64, mat: c_int) -> c_int;
    pub fn rtxh_sphere_moving(s: *mut rtxh_session, c0: *const f64, c1: *const f64, radius: f64, mat: c_int) -> c_int;
    pub fn rtxh_quad(s: *mut rtxh_session, q3: *const f64, u3: *const f64, v3: *const f64, mat: c_int) -> c_int;
    pub fn rtxh_box(s: *mut rtxh_session, a3: *const f64, b3: *const f64, mat: c_int) -> c_int;
    pub fn rtxh_list_new(s: *mut rtxh_session) -> c_int;
    pub fn rtxh_list_add(s: *mut rtxh_session, list: c_int, object: c_int) -> c_int;
    pub fn rtxh_bvh(s: *mut rtxh_session, list: c_int) -> c_int;
    pub fn rtxh_translate(s: *mut rtxh_session, object: c_int, offset3: *const f64) -> c_int;
    pub fn rtxh_rotate_y(s: *mut rtxh_session, object: c_int, degrees: f64) -> c_int;
    pub fn rtxh_medium(s: *mut rtxh_session, boundary: c_int, density: f64, tex: c_int) -> c_int;
    pub fn rtxh_bounding_box(s: *mut rtxh_session, object: c_int, out6: *mut f64) -> c_int;
    pub fn rtxh_flatten(s: *mut rtxh_session, world: c_int, lights: c_int, out: *mut rtx_scene_desc) -> rtx_status;
    pub fn rtxh_scene_cornell(s: *mut rtxh_session, width: c_int, spp: c_int, depth: c_int, lights_out: *mut c_int, cam: *mut rtx_camera_desc) -> c_int;
    pub fn rtxh_scene_cornell_smoke(s: *mut rtxh_session, width: c_int, spp: c_int, depth: c_int, lights_out: *mut c_int, cam: *mut rtx_camera_desc) -> c_int;
    pub fn rtxh_scene_final(s: *mut rtxh_session, width: c_int, spp: c_int, depth: c_int, seed: u64, earth_tex: c_int, lights_out: *mut c_int, cam: *mut rtx_camera_desc) -> c_int;
    pub fn rtxh_scene_stress(s: *mut rtxh_session, n_prims: c_int, seed: u64, width: c_int, spp: c_int, depth: c_int, lights_out: *mut c_int, cam: *mut rtx_camera_desc) -> c_int;
    pub fn rtxh_camera_render(s: *mut rtxh_session, ctx: *mut rtx_context, cam: *mut rtx_camera_desc, world: c_int, lights: c_int, seed: u64, output_filename: *const c_char, rgb_out: *mut u8, seconds_out: *mut f64) -> rtx_status;
    pub fn rtxh_camera_render_progressive(s: *mut rtxh_session, ctx: *mut rtx_context, cam: *mut rtx_camera_desc, world: c_int, lights: c_int, seed: u64, rows_per_batch: c_int, checkpoint_path: *const c_char, progress: rtxh_progress_fn, user: *mut c_void, output_filename: *const c_char, rgb_out: *mut u8, rows_done_out: *mut c_int) -> rtx_status;
    pub fn rtxh_checkpoint_fingerprint(desc: *const rtx_scene_desc, cam: *const rtx_camera_desc) -> u64;
    pub fn rtxh_write_png(path: *const c_char, rgb: *const u8, width: c_int, height: c_int) -> rtx_status;
}
