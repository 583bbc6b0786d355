/*
 * rtx.h — C ABI of the MI355X path-tracing backend (device layer).
 *
 * This is the boundary a binding for the reference renderer
 * (Rukkhadevata123/RayTracing_Rust) would target: everything inside
 * `Camera::render(&mut self, world, lights)` (src/ray_tracing/rendering/camera.rs:271-328)
 * runs behind these entry points on the GPU.  Plain pointers and sizes only;
 * every call returns an rtx_status and never throws or aborts.
 *
 * The scene crosses the boundary as a flat, caller-owned description
 * (rtx_scene_desc).  The library copies it to device memory in
 * rtx_scene_create, so the caller may free its arrays right after.  The
 * companion header rtx_host.h offers recording constructors that mirror the
 * reference's scene API and produce this description.
 *
 * All geometry and shading arithmetic is IEEE binary64, as in the reference
 * (src/ray_tracing/math/vec3.rs:4).
 *
 * Threading: a context is driven by one host thread at a time.  Distinct
 * contexts are independent.  There is no CPU fallback: every compute entry
 * point fails with RTX_ERR_NO_DEVICE when no HIP device is usable.
 */
#ifndef RTX_H
#define RTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int rtx_status;
enum {
  RTX_OK = 0,
  RTX_ERR_INVALID = 1,     /* bad argument / inconsistent scene description */
  RTX_ERR_NO_DEVICE = 2,   /* no usable HIP device (no CPU fallback exists) */
  RTX_ERR_HIP = 3,         /* a HIP runtime call failed; see rtx_last_error */
  RTX_ERR_UNSUPPORTED = 4, /* valid reference scene graph this backend cannot flatten */
  RTX_ERR_IO = 5
};

/* ------------------------------------------------------------------ scene */

/* The world is a "hit program": the reference's trait-object graph
 * (HittableList / BvhNode / Translate / RotateY / ConstantMedium over Sphere /
 * Quad leaves) linearised in the exact order the reference's recursive
 * `hit` visits it (hittable_list.rs:53-67, bvh.rs:87-105).  One closest-so-far
 * tmax shrinks along the program, which reproduces the reference's interval
 * passing including its tie-breaks. */
enum {
  RTX_NODE_BOX = 0,    /* Aabb::hit (aabb.rs:132-167); on failure continue at `skip` */
  RTX_NODE_SPHERE = 1, /* Sphere::hit (sphere.rs:83-118), strict interval */
  RTX_NODE_QUAD = 2,   /* Quad::hit (quad.rs:66-95), inclusive interval */
  RTX_NODE_ENTER = 3,  /* enter transform context idx (translate.rs:35-47, rotate_y.rs:111-127) */
  RTX_NODE_EXIT = 4,   /* return to transform context idx (0 = world) */
  RTX_NODE_MEDIUM = 5, /* ConstantMedium::hit (constant_medium.rs:42-90) */
  RTX_NODE_NONE = 15   /* miss marker in hit records */
};
#define RTX_NODE_KIND(n) ((n) >> 28)
#define RTX_NODE_IDX(n) ((n) & 0x0FFFFFFFu)
#define RTX_NODE_MAKE(kind, idx) (((uint32_t)(kind) << 28) | ((uint32_t)(idx) & 0x0FFFFFFFu))

typedef struct rtx_node {
  uint32_t kind_idx; /* RTX_NODE_MAKE(kind, index into the kind's table) */
  uint32_t skip;     /* BOX: next node when the slab test fails; otherwise unused */
} rtx_node;

typedef struct rtx_aabb { double lo[3], hi[3]; } rtx_aabb;

typedef struct rtx_sphere { /* sphere.rs:12-17; centre(t) = c0 + t*dc */
  double c0[3], dc[3], radius;
  uint32_t mat, prim_id;
} rtx_sphere;

typedef struct rtx_quad { /* quad.rs:12-22 with the constructor's derived fields (27-50) */
  double q[3], u[3], v[3], normal[3], w[3], d, area;
  uint32_t mat, prim_id;
} rtx_quad;

enum { RTX_XF_TRANSLATE = 0, RTX_XF_ROTATE_Y = 1 };
typedef struct rtx_xform_op { /* TRANSLATE: a,b,c = offset; ROTATE_Y: a = sin, b = cos */
  uint32_t kind, _pad;
  double a, b, c;
} rtx_xform_op;
typedef struct rtx_xform { /* context = ops[first_op .. first_op+n_ops), outermost first; context 0 = world */
  uint32_t first_op, n_ops;
} rtx_xform;

typedef struct rtx_medium { /* constant_medium.rs:13-17 */
  uint32_t b_begin, b_end; /* boundary hit program: nodes[b_begin .. b_end) */
  uint32_t mat;            /* the Isotropic phase function */
  uint32_t needs_uv;       /* phase texture reads (u,v): keep the stale record values (constant_medium.rs:81-87) */
  double neg_inv_density;
} rtx_medium;

enum {
  RTX_MAT_NONE = 0,          /* material.rs:69-82 */
  RTX_MAT_LAMBERTIAN = 1,    /* lambertian.rs */
  RTX_MAT_METAL = 2,         /* metal.rs */
  RTX_MAT_DIELECTRIC = 3,    /* dielectric.rs */
  RTX_MAT_DIFFUSE_LIGHT = 4, /* diffuse_light.rs */
  RTX_MAT_ISOTROPIC = 5      /* isotropic.rs */
};
typedef struct rtx_material {
  uint32_t kind;
  uint32_t tex;      /* LAMBERTIAN / DIFFUSE_LIGHT / ISOTROPIC */
  double albedo[3];  /* METAL */
  double param;      /* METAL: fuzz (already clamped to [0,1]); DIELECTRIC: refraction index */
  uint32_t needs_uv; /* texture tree contains an image texture */
  uint32_t _pad;
} rtx_material;

enum { RTX_TEX_SOLID = 0, RTX_TEX_CHECKER = 1, RTX_TEX_NOISE = 2, RTX_TEX_IMAGE = 3 };
typedef struct rtx_texture {
  uint32_t kind;
  uint32_t a, b;    /* CHECKER: even, odd texture; NOISE: perlin index; IMAGE: image index */
  uint32_t _pad;
  double color[3];  /* SOLID */
  double scale;     /* CHECKER: 1/scale (checker.rs:18); NOISE: scale */
} rtx_texture;
#define RTX_MAX_TEXTURE_DEPTH 8 /* checker-of-checker nesting accepted by rtx_scene_create */

typedef struct rtx_image { /* RGB8, row-major, top row first; height 0 = missing file (cyan, image.rs:61-63) */
  uint32_t width, height;
  uint64_t offset; /* into texels */
} rtx_image;

typedef struct rtx_perlin { /* procedural/noise.rs:6-11 */
  double ranvec[256 * 3];
  int32_t perm_x[256], perm_y[256], perm_z[256];
} rtx_perlin;

/* The `lights` hittable flattened to weighted leaves: pdf_value = sum w_i * pdf_i, random = pick i with
 * probability w_i (hittable_list.rs:78-97, bvh.rs:113-126, translate.rs:55-66, rotate_y.rs:135-149). */
typedef struct rtx_light {
  uint32_t kind;   /* RTX_NODE_SPHERE, RTX_NODE_QUAD, or RTX_NODE_NONE (default pdf 0 / direction (1,0,0)) */
  uint32_t idx, ctx, _pad;
  double weight, cum; /* cum = inclusive running sum of weights */
} rtx_light;

typedef struct rtx_scene_desc {
  uint32_t n_nodes;     const rtx_node* nodes;
  uint32_t world_begin, world_end; /* the world's hit program */
  uint32_t n_boxes;     const rtx_aabb* boxes;
  uint32_t n_spheres;   const rtx_sphere* spheres;
  uint32_t n_quads;     const rtx_quad* quads;
  uint32_t n_xforms;    const rtx_xform* xforms; /* xforms[0] must be {0,0} */
  uint32_t n_xform_ops; const rtx_xform_op* xform_ops;
  uint32_t n_media;     const rtx_medium* media;
  uint32_t n_materials; const rtx_material* materials;
  uint32_t n_textures;  const rtx_texture* textures;
  uint32_t n_images;    const rtx_image* images;
  uint64_t n_texel_bytes; const uint8_t* texels;
  uint32_t n_perlins;   const rtx_perlin* perlins;
  uint32_t n_lights;    const rtx_light* lights;
  uint32_t has_lights;  /* `lights` was Some(..) (camera.rs:204) */
  uint32_t _pad;
} rtx_scene_desc;

/* ----------------------------------------------------------------- camera */

/* Public fields of camera.rs:18-33 followed by the derived fields of camera.rs:36-48.
 * Fill the public part, then call rtx_camera_initialize (camera.rs:89-129). */
typedef struct rtx_camera_desc {
  double aspect_ratio;
  int32_t image_width, samples_per_pixel, max_depth, _pad0;
  double background[3];
  double vfov, lookfrom[3], lookat[3], vup[3], defocus_angle, focus_dist;
  /* derived */
  int32_t image_height, sqrt_spp;
  double pixel_samples_scale, recip_sqrt_spp;
  double center[3], pixel00_loc[3], pixel_delta_u[3], pixel_delta_v[3];
  double u[3], v[3], w[3], defocus_disk_u[3], defocus_disk_v[3];
} rtx_camera_desc;

enum { RTX_PIPELINE_DEFAULT = 0, RTX_PIPELINE_FUSED = 1, RTX_PIPELINE_WAVEFRONT = 2 };

typedef struct rtx_render_opts {
  int32_t sample_begin, sample_end; /* stratified sample indices, subset of [0, sqrt_spp^2) (camera.rs:257-263) */
  int32_t x0, y0, x1, y1;           /* pixel rectangle [x0,x1) x [y0,y1) */
  uint64_t seed;                    /* path RNG key; streams are keyed by (seed, pixel, sample index) */
  int32_t use_lights;               /* 0: ignore the scene's light list (camera.rs:211-216) */
  int32_t pipeline;                 /* RTX_PIPELINE_* */
} rtx_render_opts;

typedef struct rtx_stats { /* totals since context creation */
  uint64_t paths;        /* camera paths started */
  uint64_t segments;     /* world closest-hit queries made by the integrator (camera.rs:179) */
  uint64_t pdf_probes;   /* light-leaf pdf_value evaluations (sphere.rs:125, quad.rs:102) */
  double render_ms;      /* stream time of render calls (HIP events around each call) */
  uint64_t launches;     /* number of render calls */
  double last_kernel_ms; /* device time of the most recent render call */
  double traverse_ms;    /* device time of the world-program traversal launches (wavefront pipeline): wall clock of the
                          * first block in / last block out of every launch, summed */
  uint64_t traverse_launches;
  double tail_ms;        /* device time of the deferred-section traversal launches, when the program has one (HIP events
                          * around each launch; taken only by calls that run a single pipeline, where launches do not
                          * share the GPU; the launch counts are always kept) */
  uint64_t tail_launches;
  double head_ms;        /* device time of the head-section traversal launches, when the program has one */
  uint64_t head_launches;
  double shade_ms;       /* device time of the shading launches, measured like traverse_ms */
  uint64_t shade_launches;
  uint64_t roulette_drops; /* wavefront pipeline: hits whose path the roulette (camera.rs:227-231) ended before shading;
                            * the shading kernel reads only their hit kind and path coordinates */
  uint64_t order_redos;    /* rays a nearer-child-first walk could not vouch for (a hit on the boundary of its own bounding
                            * box, or a box culled by a hair) and that were walked again in the reference's order */
  uint64_t iterations;     /* wavefront pipeline: iterations enqueued = launches of EACH of its kernels, the ones that
                            * found an empty queue included (traverse_launches and shade_launches count only launches
                            * that had work; a profiler counts every launch) */
} rtx_stats;

typedef struct rtx_context rtx_context;
typedef struct rtx_scene rtx_scene;

/* ------------------------------------------------------------ entry points */

/* One context per process and GPU.  device_ids may be NULL (device 0).  n_devices must be 1: multi-GPU runs use
 * one process per GPU, each rendering its own range of sample indices, and sum the accumulators outside (the
 * library itself does no communication). */
rtx_status rtx_context_create(int n_devices, const int* device_ids, rtx_context** out);
void rtx_context_destroy(rtx_context* ctx);
/* Message of the last failing call on this context (or of the last failing context-less call when ctx is NULL). */
const char* rtx_last_error(const rtx_context* ctx);

/* Validates and uploads a scene description.  Replaces the trait-object world/lights pair handed to
 * Camera::render (camera.rs:271). */
rtx_status rtx_scene_create(rtx_context* ctx, const rtx_scene_desc* desc, rtx_scene** out);
void rtx_scene_destroy(rtx_scene* scene);

/* Camera::initialize (camera.rs:89-129), on the host. */
rtx_status rtx_camera_initialize(rtx_camera_desc* cam);

/* The hot path: Camera::render's pixel/sample loops and ray_color (camera.rs:133-268, 288-313) for the given sample
 * range and pixel rectangle.  ADDS per-pixel radiance sums into accum (image_width*image_height*3 doubles, row-major
 * RGB).  rtx_render takes a host accumulator and returns when it holds the result.  rtx_render_device takes a device
 * pointer and a hipStream_t (NULL = default stream): the accumulator is complete once work enqueued on that stream
 * after the call runs (or the stream is synchronised).  The call itself BLOCKS the host for most of the render: the
 * wavefront pipeline enqueues iterations in chunks and waits for the counters of the chunk before last to learn
 * whether paths are left; only the last chunk is still in flight when it returns.  Calls on one context are ordered
 * one after the other on the GPU whatever streams they are given (they share the context's workspace and
 * statistics); calls on different contexts that add into the same accumulator must be ordered by the caller. */
rtx_status rtx_render(rtx_scene* scene, const rtx_camera_desc* cam, const rtx_render_opts* opts, double* accum_host);
rtx_status rtx_render_device(rtx_scene* scene, const rtx_camera_desc* cam, const rtx_render_opts* opts,
                             double* accum_dev, void* hip_stream);

/* rtx_render_device for callers that cannot allocate device memory or wait on the GPU: code under HIP graph capture,
 * tight serving loops that render tile after tile, harnesses that forbid hipMalloc.  No counterpart in the reference.
 * Same arguments, same paths and same result as rtx_render_device (up to the order of the f64 additions), but the
 * call allocates nothing, creates no stream or event, reads nothing back and never blocks: it enqueues a fixed
 * sequence of launches on hip_stream.  Its workspace (path queues, hit records, counters) is an array of
 * RTX_NOALLOC_MAX_PATHS x 220 B + 4 KiB = 13.8 MiB that is part of the library's device code; every GPU holds one
 * copy from the first context created on it until the process ends.  Limit: the call's camera paths - pixels of the
 * rectangle x sample indices - must not exceed RTX_NOALLOC_MAX_PATHS (RTX_ERR_INVALID otherwise; split the rectangle
 * or the sample range).  All such calls on one GPU share that GPU's array: calls of one context are ordered by the
 * library; calls of different contexts (or processes' threads) on the same GPU that overlap in time race and must be
 * ordered by the caller.  Under stream capture (hipStreamBeginCapture on hip_stream) the call's launches become nodes
 * of the caller's graph and the library's ordering between calls is left out: launches of that graph are ordered
 * against other calls of the context by the caller; rtx_stats counts launches at capture time, the device-side
 * counters (paths, segments, probes) at every replay; the launch plan is the one of the time of capture.
 * opts->pipeline must be RTX_PIPELINE_DEFAULT or RTX_PIPELINE_WAVEFRONT (RTX_ERR_INVALID otherwise). */
#define RTX_NOALLOC_MAX_PATHS 65536
rtx_status rtx_render_device_noalloc(rtx_scene* scene, const rtx_camera_desc* cam, const rtx_render_opts* opts,
                                     double* accum_dev, void* hip_stream);

/* color_to_rgb_with_samples (rendering/color.rs:16-52) over a whole accumulator: NaN -> 0, divide by the REQUESTED
 * samples_per_pixel (camera.rs:317), gamma 2, clamp to [0,0.999], quantise. */
rtx_status rtx_resolve_rgb8(rtx_context* ctx, const double* accum_host, int width, int height, int samples_per_pixel,
                            uint8_t* rgb_host);
rtx_status rtx_resolve_rgb8_device(rtx_context* ctx, const double* accum_dev, int width, int height,
                                   int samples_per_pixel, uint8_t* rgb_dev, void* hip_stream);

/* Deterministic first-hit AOV: `world.hit(r, [0.001, inf))` (camera.rs:179) for n explicit rays (7 doubles each:
 * origin, direction, time).  Media are skipped (their hit is stochastic).  Any output pointer may be NULL.
 * Misses report t = +inf and ids 0xFFFFFFFF.  Host pointers. */
rtx_status rtx_first_hit_aov(rtx_scene* scene, const double* rays, size_t n, double* t, double* normal,
                             uint32_t* prim_id, uint32_t* mat_id, uint8_t* front);

/* Order in which BvhNode::from_slice (acceleration/bvh.rs:26-71) arranges n objects, computed on the GPU for all nodes
 * of a tree level at once: per node the longest axis of the merged box (aabb.rs:77-89), a stable sort by the boxes'
 * lower bound on that axis (bvh.rs:61-62,75-83), the split at span/2, and the two-object rule of bvh.rs:48-57.
 * boxes: n x {lo[3], hi[3]} (an object without a box: lo = +inf, hi = -inf); order_out[k] = index of the object at
 * leaf position k of the finished tree.  The shape of the tree follows from n alone, so this permutation IS the
 * build; rtxh_bvh uses it for large lists once rtxh_session_set_context has given the session a context. */
rtx_status rtx_bvh_leaf_order(rtx_context* ctx, uint32_t n, const double* boxes, uint32_t* order_out);

rtx_status rtx_get_stats(rtx_context* ctx, rtx_stats* out);
rtx_status rtx_reset_stats(rtx_context* ctx);

#ifdef __cplusplus
}
#endif
#endif /* RTX_H */
