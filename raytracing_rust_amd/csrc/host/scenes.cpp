// The reference's scene functions restated on the recording constructors
// (scenes/cornell_box.rs, scenes/final_scene.rs), Camera::initialize, and the
// Camera::render driver with its output stage (resolve + PNG).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <chrono>
#include <string>
#include <vector>

#include "../../../include/rtx_host.h"
#include "host_internal.h"
#include "../device_internal.h"

using namespace rtxh;

namespace {

void camera_defaults(rtx_camera_desc* c) {  // Camera::new, camera.rs:54-86
  memset(c, 0, sizeof(*c));
  c->aspect_ratio = 1.0; c->image_width = 100; c->samples_per_pixel = 10; c->max_depth = 10;
  c->background[0] = 0.7; c->background[1] = 0.8; c->background[2] = 1.0;
  c->vfov = 90.0; c->lookat[2] = -1.0; c->vup[1] = 1.0; c->defocus_angle = 0.0; c->focus_dist = 10.0;
}
void set3(double* a, double x, double y, double z) { a[0] = x; a[1] = y; a[2] = z; }

int lambertian_color(rtxh_session* s, double r, double g, double b) {  // Lambertian::new, lambertian.rs:17-21
  return rtxh_mat_lambertian(s, rtxh_tex_solid(s, r, g, b));
}
int diffuse_light_color(rtxh_session* s, double r, double g, double b) {  // DiffuseLight::new_color, diffuse_light.rs:22-26
  return rtxh_mat_diffuse_light(s, rtxh_tex_solid(s, r, g, b));
}
int quad9(rtxh_session* s, double qx, double qy, double qz, double ux, double uy, double uz, double vx, double vy,
          double vz, int mat) {
  double q[3] = {qx, qy, qz}, u[3] = {ux, uy, uz}, v[3] = {vx, vy, vz};
  return rtxh_quad(s, q, u, v, mat);
}

// build_cornell_box_scene, cornell_box.rs:35-104, with the light rectangle and its radiance as parameters.
void cornell_shell(rtxh_session* s, int world, int lights, const double* lq, const double* lu, const double* lv, double emit) {
  int red = lambertian_color(s, 0.65, 0.05, 0.05);
  int white = lambertian_color(s, 0.73, 0.73, 0.73);
  int green = lambertian_color(s, 0.12, 0.45, 0.15);
  int light = diffuse_light_color(s, emit, emit, emit);
  rtxh_list_add(s, world, quad9(s, 555, 0, 0, 0, 555, 0, 0, 0, 555, green));
  rtxh_list_add(s, world, quad9(s, 0, 0, 0, 0, 555, 0, 0, 0, 555, red));
  rtxh_list_add(s, world, quad9(s, 0, 555, 0, 555, 0, 0, 0, 0, 555, white));
  rtxh_list_add(s, world, quad9(s, 0, 0, 0, 555, 0, 0, 0, 0, 555, white));
  rtxh_list_add(s, world, quad9(s, 0, 0, 555, 555, 0, 0, 0, 555, 0, white));
  rtxh_list_add(s, world, rtxh_quad(s, lq, lu, lv, light));
  rtxh_list_add(s, lights, rtxh_quad(s, lq, lu, lv, rtxh_mat_none(s)));
}
void cornell_camera(rtx_camera_desc* cam, int width, int spp, int depth) {  // cornell_box.rs:137-149
  camera_defaults(cam);
  cam->aspect_ratio = 1.0; cam->image_width = width; cam->samples_per_pixel = spp; cam->max_depth = depth;
  set3(cam->background, 0, 0, 0);
  cam->vfov = 40.0;
  set3(cam->lookfrom, 278, 278, -800); set3(cam->lookat, 278, 278, 0); set3(cam->vup, 0, 1, 0);
  cam->defocus_angle = 0.0;
}

}  // namespace

extern "C" {

int rtxh_scene_cornell(rtxh_session* s, int width, int spp, int depth, int* lights_out, rtx_camera_desc* cam) {
  int world = rtxh_list_new(s), lights = rtxh_list_new(s);
  double lq[3] = {213, 554, 227}, lu[3] = {130, 0, 0}, lv[3] = {0, 0, 105};
  cornell_shell(s, world, lights, lq, lu, lv, 15.0);
  int white = lambertian_color(s, 0.73, 0.73, 0.73);                       // cornell_box.rs:111-119
  double a[3] = {0, 0, 0}, b[3] = {165, 330, 165}, off[3] = {265, 0, 295};
  rtxh_list_add(s, world, rtxh_translate(s, rtxh_rotate_y(s, rtxh_box(s, a, b, white), 15.0), off));
  double c[3] = {190, 90, 190};                                            // cornell_box.rs:122-134
  rtxh_list_add(s, world, rtxh_sphere(s, c, 90.0, rtxh_mat_dielectric(s, 1.5)));
  rtxh_list_add(s, lights, rtxh_sphere(s, c, 90.0, rtxh_mat_none(s)));
  cornell_camera(cam, width, spp, depth);
  *lights_out = lights;
  return world;
}

int rtxh_scene_cornell_smoke(rtxh_session* s, int width, int spp, int depth, int* lights_out, rtx_camera_desc* cam) {
  int world = rtxh_list_new(s), lights = rtxh_list_new(s);
  double lq[3] = {113, 554, 127}, lu[3] = {330, 0, 0}, lv[3] = {0, 0, 305};
  cornell_shell(s, world, lights, lq, lu, lv, 7.0);
  int white = lambertian_color(s, 0.73, 0.73, 0.73);
  double a[3] = {0, 0, 0}, b1[3] = {165, 330, 165}, o1[3] = {265, 0, 295}, b2[3] = {165, 165, 165}, o2[3] = {130, 0, 65};
  int box1 = rtxh_translate(s, rtxh_rotate_y(s, rtxh_box(s, a, b1, white), 15.0), o1);
  int box2 = rtxh_translate(s, rtxh_rotate_y(s, rtxh_box(s, a, b2, white), -18.0), o2);
  rtxh_list_add(s, world, rtxh_medium(s, box1, 0.01, rtxh_tex_solid(s, 0, 0, 0)));
  rtxh_list_add(s, world, rtxh_medium(s, box2, 0.01, rtxh_tex_solid(s, 1, 1, 1)));
  cornell_camera(cam, width, spp, depth);
  *lights_out = lights;
  return world;
}

int rtxh_scene_final(rtxh_session* s, int width, int spp, int depth, uint64_t seed, int earth_tex, int* lights_out,
                     rtx_camera_desc* cam) {
  HostRng rng(seed);
  int world = rtxh_list_new(s);
  int ground = lambertian_color(s, 0.48, 0.83, 0.53);                      // final_scene.rs:45
  int boxes1 = rtxh_list_new(s);
  const int boxes_per_side = 20;
  for (int i = 0; i < boxes_per_side; ++i)                                 // final_scene.rs:51-67
    for (int j = 0; j < boxes_per_side; ++j) {
      double w = 100.0;
      double x0 = -1000.0 + (double)i * w, z0 = -1000.0 + (double)j * w, y0 = 0.0;
      double x1 = x0 + w, y1 = rng.range(1.0, 101.0), z1 = z0 + w;
      double a[3] = {x0, y0, z0}, b[3] = {x1, y1, z1};
      rtxh_list_add(s, boxes1, rtxh_box(s, a, b, ground));
    }
  rtxh_list_add(s, world, rtxh_bvh(s, boxes1));                            // 70
  int light = diffuse_light_color(s, 7, 7, 7);                             // 73-79
  rtxh_list_add(s, world, quad9(s, 123, 554, 147, 300, 0, 0, 0, 0, 265, light));
  double c1[3] = {400, 400, 200}, c2[3] = {400.0 + 30.0, 400.0 + 0.0, 200.0 + 0.0};  // 82-90
  rtxh_list_add(s, world, rtxh_sphere_moving(s, c1, c2, 50.0, lambertian_color(s, 0.7, 0.3, 0.1)));
  double cg[3] = {260, 150, 45};                                           // 93-97
  rtxh_list_add(s, world, rtxh_sphere(s, cg, 50.0, rtxh_mat_dielectric(s, 1.5)));
  double cm[3] = {0, 150, 145};                                            // 100-104
  rtxh_list_add(s, world, rtxh_sphere(s, cm, 50.0, rtxh_mat_metal(s, 0.8, 0.8, 0.9, 1.0)));
  double cb[3] = {360, 150, 145};                                          // 107-117
  int boundary = rtxh_sphere(s, cb, 70.0, rtxh_mat_dielectric(s, 1.5));
  rtxh_list_add(s, world, boundary);
  rtxh_list_add(s, world, rtxh_medium(s, boundary, 0.2, rtxh_tex_solid(s, 0.2, 0.4, 0.9)));
  double c0[3] = {0, 0, 0};                                                // 120-129
  int fog_boundary = rtxh_sphere(s, c0, 5000.0, rtxh_mat_dielectric(s, 1.5));
  rtxh_list_add(s, world, rtxh_medium(s, fog_boundary, 0.0001, rtxh_tex_solid(s, 1, 1, 1)));
  if (earth_tex < 0) earth_tex = rtxh_tex_image_file(s, "earthmap.ppm");   // 132-138
  double ce[3] = {400, 200, 400};
  rtxh_list_add(s, world, rtxh_sphere(s, ce, 100.0, rtxh_mat_lambertian(s, earth_tex)));
  rtx_perlin P;                                                            // 141-146
  perlin_from_rng(rng, P);
  int noise_tex = add_noise_texture(s, 0.2, P);
  double cn[3] = {220, 280, 300};
  rtxh_list_add(s, world, rtxh_sphere(s, cn, 80.0, rtxh_mat_lambertian(s, noise_tex)));
  int boxes2 = rtxh_list_new(s);                                           // 149-163
  int white = lambertian_color(s, 0.73, 0.73, 0.73);
  for (int k = 0; k < 1000; ++k) {
    double d[9];
    for (int q = 0; q < 9; ++q) d[q] = rng.range(0.0, 165.0);
    double c[3] = {d[0], d[4], d[8]};
    rtxh_list_add(s, boxes2, rtxh_sphere(s, c, 10.0, white));
  }
  double off[3] = {-100, 270, 395};                                        // 166-172
  rtxh_list_add(s, world, rtxh_translate(s, rtxh_rotate_y(s, rtxh_bvh(s, boxes2), 15.0), off));
  int lights = rtxh_list_new(s);                                           // 175-181
  rtxh_list_add(s, lights, quad9(s, 123, 554, 147, 300, 0, 0, 0, 0, 265, rtxh_mat_none(s)));
  camera_defaults(cam);                                                    // 184-196
  cam->aspect_ratio = 1.0; cam->image_width = width; cam->samples_per_pixel = spp; cam->max_depth = depth;
  set3(cam->background, 0, 0, 0);
  cam->vfov = 40.0;
  set3(cam->lookfrom, 478, 278, -600); set3(cam->lookat, 278, 278, 0); set3(cam->vup, 0, 1, 0);
  cam->defocus_angle = 0.0;
  *lights_out = lights;
  return world;
}

// Synthetic stress scene (not in the reference): n_prims primitives, even ones spheres (centre U[-1000,1000)^3, radius
// U[1,5)), odd ones quads (corner U[-1000,1000)^3, edges U[-5,5)^3), in ONE BvhNode.  Materials by one draw per
// primitive: 70 % Lambertian of a random colour, 15 % Metal (albedo U[0.5,1)^3, fuzz U[0,0.5)), 10 % Dielectric 1.5,
// 5 % emitters of radiance 10 when the primitive is a quad (grey Lambertian otherwise); the first 64 emitting quads
// form the lights list.  Camera at (0,0,-2500) looking at the origin, vfov 40, 16:9, black background.
int rtxh_scene_stress(rtxh_session* s, int n_prims, uint64_t seed, int width, int spp, int depth, int* lights_out,
                      rtx_camera_desc* cam) {
  if (n_prims < 1) return -1;
  HostRng rng(seed);
  int all = rtxh_list_new(s), lights = rtxh_list_new(s), n_lights = 0;
  for (int i = 0; i < n_prims; ++i) {
    const bool quad = (i & 1) != 0;
    const double pick = rng.next_double();
    int mat;
    bool emitter = false;
    if (pick < 0.70) { double r = rng.next_double(), g = rng.next_double(), b = rng.next_double(); mat = lambertian_color(s, r, g, b); }
    else if (pick < 0.85) { double r = rng.range(0.5, 1.0), g = rng.range(0.5, 1.0), b = rng.range(0.5, 1.0); mat = rtxh_mat_metal(s, r, g, b, rng.range(0.0, 0.5)); }
    else if (pick < 0.95) mat = rtxh_mat_dielectric(s, 1.5);
    else if (quad) { mat = diffuse_light_color(s, 10, 10, 10); emitter = true; }
    else mat = lambertian_color(s, 0.5, 0.5, 0.5);
    int obj;
    if (!quad) {
      double c[3] = {rng.range(-1000, 1000), rng.range(-1000, 1000), rng.range(-1000, 1000)};
      obj = rtxh_sphere(s, c, rng.range(1.0, 5.0), mat);
    } else {
      double q[3] = {rng.range(-1000, 1000), rng.range(-1000, 1000), rng.range(-1000, 1000)};
      double u[3] = {rng.range(-5, 5), rng.range(-5, 5), rng.range(-5, 5)};
      double v[3] = {rng.range(-5, 5), rng.range(-5, 5), rng.range(-5, 5)};
      obj = rtxh_quad(s, q, u, v, mat);
      if (emitter && n_lights < 64) { rtxh_list_add(s, lights, obj); ++n_lights; }
    }
    if (obj < 0) return -1;
    rtxh_list_add(s, all, obj);
  }
  int world = rtxh_list_new(s);
  rtxh_list_add(s, world, rtxh_bvh(s, all));
  camera_defaults(cam);
  cam->aspect_ratio = 16.0 / 9.0; cam->image_width = width; cam->samples_per_pixel = spp; cam->max_depth = depth;
  set3(cam->background, 0, 0, 0);
  cam->vfov = 40.0;
  set3(cam->lookfrom, 0, 0, -2500); set3(cam->lookat, 0, 0, 0); set3(cam->vup, 0, 1, 0);
  cam->defocus_angle = 0.0;
  *lights_out = n_lights ? lights : -1;
  return world;
}

// Camera::initialize, camera.rs:89-129.
rtx_status rtx_camera_initialize(rtx_camera_desc* c) {
  if (!c) return rtx_internal_fail(nullptr, RTX_ERR_INVALID, "rtx_camera_initialize: null camera");
  if (c->image_width < 1 || !(c->aspect_ratio > 0.0) || c->samples_per_pixel < 1)
    return rtx_internal_fail(nullptr, RTX_ERR_INVALID, "rtx_camera_initialize: image_width and samples_per_pixel must be at least 1 and aspect_ratio positive");
  c->image_height = (int32_t)((double)c->image_width / c->aspect_ratio);
  if (c->image_height < 1) c->image_height = 1;
  c->sqrt_spp = (int32_t)sqrt((double)c->samples_per_pixel);
  c->pixel_samples_scale = 1.0 / (double)(c->sqrt_spp * c->sqrt_spp);
  c->recip_sqrt_spp = 1.0 / (double)c->sqrt_spp;
  const double pi = 3.14159265358979323846;
  double theta = c->vfov * pi / 180.0;
  double h = tan(theta / 2.0);
  double viewport_height = 2.0 * h * c->focus_dist;
  double viewport_width = viewport_height * ((double)c->image_width / (double)c->image_height);
  double w[3] = {c->lookfrom[0] - c->lookat[0], c->lookfrom[1] - c->lookat[1], c->lookfrom[2] - c->lookat[2]};
  double wl = sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
  for (int k = 0; k < 3; ++k) w[k] = w[k] / wl;
  const double* up = c->vup;
  double u[3] = {up[1] * w[2] - up[2] * w[1], up[2] * w[0] - up[0] * w[2], up[0] * w[1] - up[1] * w[0]};
  double ul = sqrt((u[0] * u[0] + u[1] * u[1]) + u[2] * u[2]);
  for (int k = 0; k < 3; ++k) u[k] = u[k] / ul;
  double v[3] = {w[1] * u[2] - w[2] * u[1], w[2] * u[0] - w[0] * u[2], w[0] * u[1] - w[1] * u[0]};
  double defocus_radius = c->focus_dist * tan((c->defocus_angle / 2.0) * pi / 180.0);
  for (int k = 0; k < 3; ++k) {
    double viewport_u = viewport_width * u[k];
    double viewport_v = viewport_height * (-v[k]);
    c->pixel_delta_u[k] = viewport_u / (double)c->image_width;
    c->pixel_delta_v[k] = viewport_v / (double)c->image_height;
    double upper_left = c->lookfrom[k] - (c->focus_dist * w[k]) - viewport_u / 2.0 - viewport_v / 2.0;
    c->pixel00_loc[k] = upper_left + 0.5 * (c->pixel_delta_u[k] + c->pixel_delta_v[k]);
    c->center[k] = c->lookfrom[k];
    c->u[k] = u[k]; c->v[k] = v[k]; c->w[k] = w[k];
    c->defocus_disk_u[k] = u[k] * defocus_radius;
    c->defocus_disk_v[k] = v[k] * defocus_radius;
  }
  return RTX_OK;
}

// 8-bit RGB PNG: signature, IHDR, one zlib-compressed IDAT (filter 0 on every row), IEND.
rtx_status rtxh_write_png(const char* path, const uint8_t* rgb, int width, int height) {
  if (!path || !rgb || width < 1 || height < 1) return RTX_ERR_INVALID;
  std::vector<uint8_t> raw((size_t)height * ((size_t)width * 3 + 1));
  for (int y = 0; y < height; ++y) {
    raw[(size_t)y * ((size_t)width * 3 + 1)] = 0;
    memcpy(&raw[(size_t)y * ((size_t)width * 3 + 1) + 1], rgb + (size_t)y * width * 3, (size_t)width * 3);
  }
  uLongf clen = compressBound((uLong)raw.size());
  std::vector<uint8_t> comp(clen);
  if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return RTX_ERR_IO;
  FILE* f = fopen(path, "wb");
  if (!f) return RTX_ERR_IO;
  auto be32 = [](uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; };
  auto chunk = [&](const char* type, const uint8_t* data, uint32_t len) {
    uint8_t hdr[8];
    be32(hdr, len); memcpy(hdr + 4, type, 4);
    fwrite(hdr, 1, 8, f);
    if (len) fwrite(data, 1, len, f);
    uLong crc = crc32(0L, (const Bytef*)type, 4);
    if (len) crc = crc32(crc, data, len);
    uint8_t c[4];
    be32(c, (uint32_t)crc);
    fwrite(c, 1, 4, f);
  };
  const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
  fwrite(sig, 1, 8, f);
  uint8_t ihdr[13];
  be32(ihdr, (uint32_t)width); be32(ihdr + 4, (uint32_t)height);
  ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
  chunk("IHDR", ihdr, 13);
  chunk("IDAT", comp.data(), (uint32_t)clen);
  chunk("IEND", nullptr, 0);
  bool ok = ferror(f) == 0;
  ok = (fclose(f) == 0) && ok;
  return ok ? RTX_OK : RTX_ERR_IO;
}

// Camera::render, camera.rs:271-328, with the pixel loops on the GPU.
rtx_status rtxh_camera_render(rtxh_session* s, rtx_context* ctx, rtx_camera_desc* cam, int world, int lights,
                              uint64_t seed, const char* output_filename, uint8_t* rgb_out, double* seconds_out) {
  if (!s || !ctx || !cam) return RTX_ERR_INVALID;
  rtx_internal_fail(ctx, RTX_OK, "");   // a message left by an earlier call must not be taken for this call's
  rtx_status st = rtx_camera_initialize(cam);
  if (st != RTX_OK) return st;
  rtx_scene_desc desc;
  st = rtxh_flatten(s, world, lights, &desc);
  if (st != RTX_OK) return st;
  rtx_scene* scene = nullptr;
  st = rtx_scene_create(ctx, &desc, &scene);
  if (st != RTX_OK) return st;
  const size_t npix = (size_t)cam->image_width * (size_t)cam->image_height;
  std::vector<double> accum(npix * 3, 0.0);
  rtx_render_opts o;
  memset(&o, 0, sizeof(o));
  o.sample_begin = 0; o.sample_end = cam->sqrt_spp * cam->sqrt_spp;
  o.x0 = 0; o.y0 = 0; o.x1 = cam->image_width; o.y1 = cam->image_height;
  o.seed = seed; o.use_lights = 1; o.pipeline = RTX_PIPELINE_DEFAULT;
  auto t0 = std::chrono::steady_clock::now();
  st = rtx_render(scene, cam, &o, accum.data());
  auto t1 = std::chrono::steady_clock::now();
  if (seconds_out) *seconds_out = std::chrono::duration<double>(t1 - t0).count();
  std::vector<uint8_t> rgb(npix * 3);
  if (st == RTX_OK) st = rtx_resolve_rgb8(ctx, accum.data(), cam->image_width, cam->image_height, cam->samples_per_pixel, rgb.data());
  rtx_scene_destroy(scene);
  if (st != RTX_OK) return st;
  if (rgb_out) memcpy(rgb_out, rgb.data(), rgb.size());
  if (output_filename) {
    st = rtxh_write_png(output_filename, rgb.data(), cam->image_width, cam->image_height);
    if (st == RTX_OK) fprintf(stderr, "image saved as %s\n", output_filename);          // camera.rs:323
    else fprintf(stderr, "error while saving image %s\n", output_filename);            // camera.rs:324
  }
  return st;
}

// ---- progressive rendering with checkpoints -------------------------------------------------------------------
namespace {
struct CkptHeader {
  char magic[8];                       // "RTXCKPT2"
  uint32_t header_bytes, _pad;         // sizeof(CkptHeader) of the build that wrote the file
  int32_t width, height, sqrt_spp, samples_per_pixel, max_depth, next_row;
  uint64_t seed, fingerprint;
};
uint64_t fnv1a(uint64_t h, const void* p, size_t n) {
  const unsigned char* b = (const unsigned char*)p;
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}
// A cheap identity of scene + camera: a checkpoint of another render must not be resumed.  Every table of the
// description goes in, with its count: a changed fog density, transform, light list or texel must not match.
uint64_t fingerprint(const rtx_scene_desc& d, const rtx_camera_desc& c) {
  uint64_t h = 1469598103934665603ull;
  auto table = [&](const void* p, uint64_t count, size_t elem) {
    h = fnv1a(h, &count, sizeof(count));
    if (count) h = fnv1a(h, p, (size_t)count * elem);
  };
  table(d.nodes, d.n_nodes, sizeof(rtx_node));
  table(d.boxes, d.n_boxes, sizeof(rtx_aabb));
  table(d.spheres, d.n_spheres, sizeof(rtx_sphere));
  table(d.quads, d.n_quads, sizeof(rtx_quad));
  table(d.xforms, d.n_xforms, sizeof(rtx_xform));
  table(d.xform_ops, d.n_xform_ops, sizeof(rtx_xform_op));
  table(d.media, d.n_media, sizeof(rtx_medium));
  table(d.materials, d.n_materials, sizeof(rtx_material));
  table(d.textures, d.n_textures, sizeof(rtx_texture));
  table(d.images, d.n_images, sizeof(rtx_image));
  table(d.texels, d.n_texel_bytes, 1);
  table(d.perlins, d.n_perlins, sizeof(rtx_perlin));
  table(d.lights, d.n_lights, sizeof(rtx_light));
  const uint32_t tail[3] = {d.world_begin, d.world_end, d.has_lights};
  h = fnv1a(h, tail, sizeof(tail));
  h = fnv1a(h, &c, sizeof(c));
  return h;
}
bool save_checkpoint(const char* path, const CkptHeader& hd, const std::vector<double>& accum) {
  const std::string tmp = std::string(path) + ".tmp";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  bool ok = fwrite(&hd, sizeof(hd), 1, f) == 1 && fwrite(accum.data(), sizeof(double), accum.size(), f) == accum.size();
  ok = (fclose(f) == 0) && ok;
  if (ok) ok = rename(tmp.c_str(), path) == 0;   // the previous checkpoint stays valid until the new one is complete
  if (!ok) remove(tmp.c_str());
  return ok;
}
bool load_checkpoint(const char* path, const CkptHeader& want, CkptHeader& got, std::vector<double>& accum) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  bool ok = fread(&got, sizeof(got), 1, f) == 1 && memcmp(got.magic, want.magic, 8) == 0 && got.header_bytes == want.header_bytes &&
            got.width == want.width &&
            got.height == want.height && got.sqrt_spp == want.sqrt_spp && got.samples_per_pixel == want.samples_per_pixel &&
            got.max_depth == want.max_depth && got.seed == want.seed && got.fingerprint == want.fingerprint &&
            got.next_row >= 0 && got.next_row <= want.sqrt_spp;
  std::vector<double> tmp(accum.size());
  ok = ok && fread(tmp.data(), sizeof(double), tmp.size(), f) == tmp.size();
  fclose(f);
  if (ok) accum.swap(tmp);
  return ok;
}
}  // namespace

uint64_t rtxh_checkpoint_fingerprint(const rtx_scene_desc* desc, const rtx_camera_desc* cam) {
  if (!desc || !cam) return 0;
  return fingerprint(*desc, *cam);
}

rtx_status rtxh_camera_render_progressive(rtxh_session* s, rtx_context* ctx, rtx_camera_desc* cam, int world, int lights,
                                          uint64_t seed, int rows_per_batch, const char* checkpoint_path,
                                          rtxh_progress_fn progress, void* user, const char* output_filename,
                                          uint8_t* rgb_out, int* rows_done_out) {
  if (!s || !ctx || !cam || rows_per_batch < 1) return RTX_ERR_INVALID;
  rtx_internal_fail(ctx, RTX_OK, "");
  rtx_status st = rtx_camera_initialize(cam);
  if (st != RTX_OK) return st;
  rtx_scene_desc desc;
  st = rtxh_flatten(s, world, lights, &desc);
  if (st != RTX_OK) return st;
  rtx_scene* scene = nullptr;
  st = rtx_scene_create(ctx, &desc, &scene);
  if (st != RTX_OK) return st;
  const size_t npix = (size_t)cam->image_width * (size_t)cam->image_height;
  std::vector<double> accum(npix * 3, 0.0);
  CkptHeader hd;
  memset(&hd, 0, sizeof(hd));
  memcpy(hd.magic, "RTXCKPT2", 8);
  hd.header_bytes = (uint32_t)sizeof(CkptHeader);
  hd.width = cam->image_width; hd.height = cam->image_height; hd.sqrt_spp = cam->sqrt_spp;
  hd.samples_per_pixel = cam->samples_per_pixel; hd.max_depth = cam->max_depth; hd.seed = seed;
  hd.fingerprint = fingerprint(desc, *cam);
  int row = 0;
  if (checkpoint_path) {
    CkptHeader got;
    if (load_checkpoint(checkpoint_path, hd, got, accum)) row = got.next_row;   // resume where the last run stopped
  }
  const int rows = cam->sqrt_spp;
  auto t0 = std::chrono::steady_clock::now();
  bool stopped = false;
  while (row < rows && st == RTX_OK && !stopped) {
    const int end = row + rows_per_batch < rows ? row + rows_per_batch : rows;
    rtx_render_opts o;
    memset(&o, 0, sizeof(o));
    o.sample_begin = row * rows; o.sample_end = end * rows;
    o.x0 = 0; o.y0 = 0; o.x1 = cam->image_width; o.y1 = cam->image_height;
    o.seed = seed; o.use_lights = 1; o.pipeline = RTX_PIPELINE_DEFAULT;
    st = rtx_render(scene, cam, &o, accum.data());
    if (st != RTX_OK) break;
    row = end;
    if (checkpoint_path) {
      hd.next_row = row;
      if (!save_checkpoint(checkpoint_path, hd, accum)) { st = RTX_ERR_IO; break; }
    }
    if (progress) {
      const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      stopped = progress(row, rows, secs, user) != 0;
    }
  }
  if (rows_done_out) *rows_done_out = row;
  std::vector<uint8_t> rgb(npix * 3);
  // an unfinished image is resolved with the samples it has, scaled as the finished one would be (color.rs:16-52)
  const int spp_so_far = row == rows ? cam->samples_per_pixel
                                     : (int)((long long)cam->samples_per_pixel * row / rows > 0 ? (long long)cam->samples_per_pixel * row / rows : 1);
  if (st == RTX_OK) st = rtx_resolve_rgb8(ctx, accum.data(), cam->image_width, cam->image_height, spp_so_far, rgb.data());
  rtx_scene_destroy(scene);
  if (st != RTX_OK) return st;
  if (rgb_out) memcpy(rgb_out, rgb.data(), rgb.size());
  if (output_filename && row == rows) {
    st = rtxh_write_png(output_filename, rgb.data(), cam->image_width, cam->image_height);
    if (st == RTX_OK) fprintf(stderr, "image saved as %s\n", output_filename);
    else fprintf(stderr, "error while saving image %s\n", output_filename);
  }
  return st;
}

}  // extern "C"
