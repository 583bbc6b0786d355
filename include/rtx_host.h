/*
 * rtx_host.h — C ABI of the host scene layer: recording constructors that
 * mirror the reference's scene API one to one, the flattening of the recorded
 * graph into an rtx_scene_desc, the reference's scene functions, and the
 * output stage.  Pure host code: nothing here touches the GPU except
 * rtxh_camera_render, which drives the device layer of rtx.h.
 *
 * The reference's traits have no introspection hook (hittable.rs:97-115), so
 * a trait-object graph cannot be flattened from outside; a binding therefore
 * calls these constructors from the reference's own `new` functions, one
 * provided method per trait that re-records the object.  Handles are small non-negative integers local to a
 * session; a negative return value is an error (rtxh_last_error).
 *
 * Primitive ids and material ids (reported by the first-hit AOV) are creation
 * ordinals: every Sphere/Quad constructed gets the next primitive id (a box
 * takes six, in the face order of quad.rs:136-171); every material the next
 * material id (a ConstantMedium creates its Isotropic phase function and takes
 * one, constant_medium.rs:22-38).
 */
#ifndef RTX_HOST_H
#define RTX_HOST_H

#include "rtx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtxh_session rtxh_session;

rtxh_session* rtxh_session_new(void);
void rtxh_session_free(rtxh_session* s);
/* Optional: lets rtxh_bvh sort lists of RTXH_GPU_BVH_MIN objects or more on the context's GPU (rtx_bvh_leaf_order);
 * the tree is the one the host build makes.  ctx must outlive the session's rtxh_bvh calls; NULL detaches. */
void rtxh_session_set_context(rtxh_session* s, rtx_context* ctx);
#define RTXH_GPU_BVH_MIN 4096
const char* rtxh_last_error(const rtxh_session* s);

/* Textures */
int rtxh_tex_solid(rtxh_session* s, double r, double g, double b);          /* SolidColor::new, solid_color.rs:13 */
int rtxh_tex_checker(rtxh_session* s, double scale, int even, int odd);     /* CheckerTexture::new, checker.rs:16 */
/* NoiseTexture::new (texture/noise.rs:15): the Perlin tables (procedural/noise.rs:16-31) come from a SplitMix64
 * stream seeded with `seed`: 256 x normalize(U[-1,1)^3), then three Fisher-Yates permutations with
 * target = floor(u*(i+1)). */
int rtxh_tex_noise(rtxh_session* s, double scale, uint64_t seed);
/* NoiseTexture::new_with_noise (texture/noise.rs:24) with explicit tables. */
int rtxh_tex_noise_tables(rtxh_session* s, double scale, const double* ranvec768, const int32_t* perm_x256,
                          const int32_t* perm_y256, const int32_t* perm_z256);
int rtxh_tex_image(rtxh_session* s, uint32_t width, uint32_t height, const uint8_t* rgb); /* decoded texels; height 0 = missing */
/* ImageTexture::new (image.rs:15-43): tries name, textures/name, ../textures/name, $RTW_IMAGES/name; reads binary
 * PPM (P6); a file that cannot be read yields the reference's cyan fallback. */
int rtxh_tex_image_file(rtxh_session* s, const char* filename);

/* Materials */
int rtxh_mat_lambertian(rtxh_session* s, int tex);                          /* Lambertian::new_texture, lambertian.rs:25 */
int rtxh_mat_metal(rtxh_session* s, double r, double g, double b, double fuzz); /* Metal::new, metal.rs:16 */
int rtxh_mat_dielectric(rtxh_session* s, double refraction_index);          /* Dielectric::new, dielectric.rs:17 */
int rtxh_mat_diffuse_light(rtxh_session* s, int tex);                       /* DiffuseLight::new, diffuse_light.rs:16 */
int rtxh_mat_isotropic(rtxh_session* s, int tex);                           /* Isotropic::new, isotropic.rs:17 */
int rtxh_mat_none(rtxh_session* s);                                         /* NoMaterial, material.rs:69 */

/* Hittables */
int rtxh_sphere(rtxh_session* s, const double* center3, double radius, int mat);           /* Sphere::new, sphere.rs:22 */
int rtxh_sphere_moving(rtxh_session* s, const double* c0, const double* c1, double radius, int mat); /* sphere.rs:36 */
int rtxh_quad(rtxh_session* s, const double* q3, const double* u3, const double* v3, int mat); /* Quad::new, quad.rs:27 */
int rtxh_box(rtxh_session* s, const double* a3, const double* b3, int mat);                /* box_new, quad.rs:125 */
int rtxh_list_new(rtxh_session* s);                                                        /* HittableList::new */
int rtxh_list_add(rtxh_session* s, int list, int object);                                  /* HittableList::add */
int rtxh_bvh(rtxh_session* s, int list);                                                   /* BvhNode::new, bvh.rs:21 */
int rtxh_translate(rtxh_session* s, int object, const double* offset3);                    /* Translate::new */
int rtxh_rotate_y(rtxh_session* s, int object, double degrees);                            /* RotateY::new */
int rtxh_medium(rtxh_session* s, int boundary, double density, int tex);                   /* ConstantMedium::new */

/* Hittable::bounding_box: writes xmin xmax ymin ymax zmin zmax; returns 1, 0 for None, negative on error. */
int rtxh_bounding_box(rtxh_session* s, int object, double* out6);

/* Flatten `world` (and `lights`, or -1 for None) into a scene description.  The arrays belong to the session and
 * stay valid until the next rtxh_flatten on it or rtxh_session_free. */
rtx_status rtxh_flatten(rtxh_session* s, int world, int lights, rtx_scene_desc* out);

/* The reference's scene functions.  Each returns the world handle, stores the lights handle and fills the camera's
 * public fields (call rtx_camera_initialize before rendering).
 *   rtxh_scene_cornell       cornell_box_with_glass_sphere, scenes/cornell_box.rs:107-149
 *   rtxh_scene_final         final_scene_next_week, scenes/final_scene.rs:41-196; random content drawn from one
 *                            SplitMix64(seed) stream in program order: 400 box heights, the Perlin tables, 1000
 *                            sphere centres (nine draws each, keeping the 1st, 5th and 9th, final_scene.rs:155-159);
 *                            earth_tex < 0 loads "earthmap.ppm" through rtxh_tex_image_file
 *   rtxh_scene_cornell_smoke not in the reference's sources (only images/output_cornell_smoke.png): Cornell walls,
 *                            light (113,554,127) 330x305 emitting 7, two box-bounded media of density 0.01 */
int rtxh_scene_cornell(rtxh_session* s, int width, int spp, int depth, int* lights_out, rtx_camera_desc* cam);
int rtxh_scene_cornell_smoke(rtxh_session* s, int width, int spp, int depth, int* lights_out, rtx_camera_desc* cam);
int rtxh_scene_final(rtxh_session* s, int width, int spp, int depth, uint64_t seed, int earth_tex, int* lights_out,
                     rtx_camera_desc* cam);

/* Synthetic stress scene (not in the reference): n_prims random spheres and quads of mixed materials in one BvhNode,
 * up to 64 emitting quads as lights, camera at (0,0,-2500), 16:9.  The benchmark configuration is 1 M primitives at
 * 3840x2160, 4096 spp, depth 50, seed 0x5EED0005.  lights_out receives -1 when no primitive emits. */
int rtxh_scene_stress(rtxh_session* s, int n_prims, uint64_t seed, int width, int spp, int depth, int* lights_out,
                      rtx_camera_desc* cam);

/* Camera::render (camera.rs:271-328): initialise the camera, flatten, render every sample of every pixel on the GPU,
 * resolve to RGB8 and write cam's image as PNG (output_filename may be NULL to skip the file).  rgb_out may be NULL.
 * seconds_out (may be NULL) receives the wall time of the render call alone. */
rtx_status rtxh_camera_render(rtxh_session* s, rtx_context* ctx, rtx_camera_desc* cam, int world, int lights,
                              uint64_t seed, const char* output_filename, uint8_t* rgb_out, double* seconds_out);

/* Camera::render in batches of rows of the sqrt_spp x sqrt_spp stratification, with progress reports and a
 * checkpoint of the accumulator after every batch.  No counterpart in the reference, which shows an indicatif
 * progress bar (camera.rs:277-285) and saves only at the end (camera.rs:322): a 5000-spp render cannot be resumed
 * there.  If checkpoint_path names a checkpoint of the same scene, camera and seed, the render continues after the
 * rows it holds; the file is rewritten atomically after every batch.  progress (may be NULL) is called after every
 * batch; a non-zero return stops the render (the checkpoint then holds the rows done).  rows_done_out (may be NULL)
 * receives the rows rendered so far; the PNG is written only when all rows are done; rgb_out (may be NULL) receives
 * the image of the rows done, scaled as the finished image would be. */
typedef int (*rtxh_progress_fn)(int rows_done, int rows_total, double seconds, void* user);
rtx_status rtxh_camera_render_progressive(rtxh_session* s, rtx_context* ctx, rtx_camera_desc* cam, int world, int lights,
                                          uint64_t seed, int rows_per_batch, const char* checkpoint_path,
                                          rtxh_progress_fn progress, void* user, const char* output_filename,
                                          uint8_t* rgb_out, int* rows_done_out);

/* The identity of (scene, camera) a checkpoint stores: a hash over every table of the description with its count
 * (geometry, transforms, media, materials, textures, texels, Perlin tables, lights) and the camera.  A checkpoint
 * whose value differs is not resumed.  No counterpart in the reference. */
uint64_t rtxh_checkpoint_fingerprint(const rtx_scene_desc* desc, const rtx_camera_desc* cam);

/* RgbImage::save (camera.rs:322): 8-bit RGB PNG through zlib. */
rtx_status rtxh_write_png(const char* path, const uint8_t* rgb, int width, int height);

#ifdef __cplusplus
}
#endif
#endif /* RTX_HOST_H */
