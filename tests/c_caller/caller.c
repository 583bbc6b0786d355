/* A native caller of the C ABI, written the way a binding for the reference would drive it: the call sequence of
 * cornell_box_with_glass_sphere (scenes/cornell_box.rs:107-159) followed by Camera::render (rendering/camera.rs:271-328)
 * - record the world and the lights object by object through the constructors, create the context, render, resolve,
 * save the PNG - in plain C, linked against librtx_amd.so.  No Python, no ctypes: it proves that the headers and the
 * library are usable from compiled code.  Test infrastructure (tests/test_c_caller_gpu.py compares its image with the
 * ctypes path's).
 *
 *   caller <width> <samples_per_pixel> <max_depth> <seed> <out.rgb> [out.png]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rtx_host.h"

static int fail(const char* what, rtxh_session* s, rtx_context* ctx) {
  fprintf(stderr, "caller: %s failed: device layer '%s', scene layer '%s'\n", what, rtx_last_error(ctx), s ? rtxh_last_error(s) : "");
  return 1;
}

/* What Material::record / Hittable::record of the binding do for each object of the scene, depth first. */
static int lambertian(rtxh_session* s, double r, double g, double b) { return rtxh_mat_lambertian(s, rtxh_tex_solid(s, r, g, b)); }
static int quad(rtxh_session* s, double qx, double qy, double qz, double ux, double uy, double uz, double vx, double vy, double vz, int mat) {
  const double q[3] = {qx, qy, qz}, u[3] = {ux, uy, uz}, v[3] = {vx, vy, vz};
  return rtxh_quad(s, q, u, v, mat);
}

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: caller <width> <spp> <depth> <seed> <out.rgb> [out.png]\n"); return 2; }
  const int width = atoi(argv[1]), spp = atoi(argv[2]), depth = atoi(argv[3]);
  const unsigned long long seed = strtoull(argv[4], NULL, 0);

  rtx_context* ctx = NULL;
  if (rtx_context_create(1, NULL, &ctx) != RTX_OK) return fail("rtx_context_create", NULL, NULL);   /* no GPU: an error, never a fallback */
  rtxh_session* s = rtxh_session_new();
  if (!s) return fail("rtxh_session_new", NULL, ctx);

  /* build_cornell_box_scene, cornell_box.rs:35-104 */
  const int world = rtxh_list_new(s), lights = rtxh_list_new(s);
  const int red = lambertian(s, 0.65, 0.05, 0.05), white = lambertian(s, 0.73, 0.73, 0.73), green = lambertian(s, 0.12, 0.45, 0.15);
  const int light = rtxh_mat_diffuse_light(s, rtxh_tex_solid(s, 15.0, 15.0, 15.0));
  rtxh_list_add(s, world, quad(s, 555, 0, 0, 0, 555, 0, 0, 0, 555, green));
  rtxh_list_add(s, world, quad(s, 0, 0, 0, 0, 555, 0, 0, 0, 555, red));
  rtxh_list_add(s, world, quad(s, 0, 555, 0, 555, 0, 0, 0, 0, 555, white));
  rtxh_list_add(s, world, quad(s, 0, 0, 0, 555, 0, 0, 0, 0, 555, white));
  rtxh_list_add(s, world, quad(s, 0, 0, 555, 555, 0, 0, 0, 555, 0, white));
  rtxh_list_add(s, world, quad(s, 213, 554, 227, 130, 0, 0, 0, 0, 105, light));
  rtxh_list_add(s, lights, quad(s, 213, 554, 227, 130, 0, 0, 0, 0, 105, rtxh_mat_none(s)));
  /* cornell_box.rs:110-134: the rotated, translated box, the glass sphere, the sphere as a light */
  const int white2 = lambertian(s, 0.73, 0.73, 0.73);
  const double a[3] = {0, 0, 0}, b[3] = {165, 330, 165}, off[3] = {265, 0, 295}, c[3] = {190, 90, 190};
  rtxh_list_add(s, world, rtxh_translate(s, rtxh_rotate_y(s, rtxh_box(s, a, b, white2), 15.0), off));
  rtxh_list_add(s, world, rtxh_sphere(s, c, 90.0, rtxh_mat_dielectric(s, 1.5)));
  rtxh_list_add(s, lights, rtxh_sphere(s, c, 90.0, rtxh_mat_none(s)));
  if (world < 0 || lights < 0) return fail("scene recording", s, ctx);

  /* the camera's public fields, cornell_box.rs:137-149 over Camera::new (camera.rs:54-86) */
  rtx_camera_desc cam;
  memset(&cam, 0, sizeof(cam));
  cam.aspect_ratio = 1.0; cam.image_width = width; cam.samples_per_pixel = spp; cam.max_depth = depth;
  cam.vfov = 40.0;
  cam.lookfrom[0] = 278; cam.lookfrom[1] = 278; cam.lookfrom[2] = -800;
  cam.lookat[0] = 278; cam.lookat[1] = 278; cam.lookat[2] = 0;
  cam.vup[1] = 1.0;
  cam.defocus_angle = 0.0; cam.focus_dist = 10.0;

  /* Camera::render, camera.rs:271-328 */
  unsigned char* rgb = (unsigned char*)malloc((size_t)width * (size_t)width * 3);
  double seconds = 0.0;
  if (rtxh_camera_render(s, ctx, &cam, world, lights, seed, argc > 6 ? argv[6] : NULL, rgb, &seconds) != RTX_OK)
    return fail("rtxh_camera_render", s, ctx);
  rtx_stats st;
  if (rtx_get_stats(ctx, &st) != RTX_OK) return fail("rtx_get_stats", s, ctx);
  printf("%d x %d, %llu paths, %llu segments, %.3f s\n", cam.image_width, cam.image_height, (unsigned long long)st.paths,
         (unsigned long long)st.segments, seconds);
  FILE* f = fopen(argv[5], "wb");
  if (!f || fwrite(rgb, 1, (size_t)cam.image_width * (size_t)cam.image_height * 3, f) != (size_t)cam.image_width * (size_t)cam.image_height * 3) return fail("writing the image", s, ctx);
  fclose(f);
  free(rgb);
  rtxh_session_free(s);
  rtx_context_destroy(ctx);
  return 0;
}
