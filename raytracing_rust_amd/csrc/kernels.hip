// Device layer of the MI355X path-tracing backend: HIP kernels for gfx950 and
// the C ABI of include/rtx.h.  There is no CPU path in this file: every
// compute entry point needs a HIP device and fails with RTX_ERR_NO_DEVICE
// otherwise.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/rtx.h"
#include "core/rt_shade.h"

using namespace rt;

// --------------------------------------------------------------- kernels ---

// Fused path tracer: one lane owns one pixel and walks its samples one after another.  Every trip of the loop is
// one segment for every lane of the wave (closest hit, then shading), so the wave stays in one phase; a lane whose
// path ended starts its next sample in place (path regeneration), which keeps lanes busy without any path state in
// memory.  A wave covers an 8x8 pixel tile, a 256-thread block a 16x16 tile (the reference's tile, camera.rs:288).
__global__ __launch_bounds__(256) void k_render_fused(Scene S, rtx_camera_desc cam, rtx_render_opts o, int use_lights,
                                                      double* __restrict__ accum, unsigned long long* __restrict__ counters) {
  const int tiles_x = (o.x1 - o.x0 + 15) >> 4;
  const int bx = (int)(blockIdx.x % (unsigned)tiles_x), by = (int)(blockIdx.x / (unsigned)tiles_x);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = o.x0 + bx * 16 + (wave & 1) * 8 + (lane & 7);
  const int j = o.y0 + by * 16 + (wave >> 1) * 8 + (lane >> 3);
  const bool inrect = i < o.x1 && j < o.y1;
  const V3 bg = v3(cam.background);
  const uint32_t pixel = (uint32_t)(j * cam.image_width + i);

  V3 sum = v3(0.0, 0.0, 0.0);
  uint32_t segs = 0, probes = 0, npaths = 0;
  int s = o.sample_begin;
  bool alive = false;
  PathState ps;
  Ray ray;
  ray.o = v3(0, 0, 0); ray.d = v3(0, 0, 1); ray.tm = 0;
  for (;;) {
    if (!alive) {
      if (!inrect || s >= o.sample_end) break;
      PathRng rng = rng_make(o.seed, pixel, (uint32_t)s);
      ray = camera_ray(cam, i, j, s, rng);
      path_begin(ps, cam.max_depth, rng);
      ++s; ++npaths;
      alive = ps.depth > 0;
      if (!alive) continue;
    }
    Hit hit;
    ++segs;
    bool any = closest_hit<true>(S, S.world_begin, S.world_end, ray, 0, 0.001, RT_INF, hit, &ps.rng, ps.bounce);
    Ray next;
    alive = shade_bounce(S, bg, use_lights != 0, ray, any, hit, ps, next, probes);
    ray = next;
    if (!alive) sum = sum + ps.L;
  }
  if (inrect) {
    double* a = accum + (size_t)pixel * 3;
    a[0] += sum.x; a[1] += sum.y; a[2] += sum.z;
  }
  // per-block totals -> three atomics per block
  __shared__ unsigned int sh[3];
  if (threadIdx.x < 3) sh[threadIdx.x] = 0;
  __syncthreads();
  atomicAdd(&sh[0], npaths); atomicAdd(&sh[1], segs); atomicAdd(&sh[2], probes);
  __syncthreads();
  if (threadIdx.x < 3) atomicAdd(&counters[threadIdx.x], (unsigned long long)sh[threadIdx.x]);
}

__global__ __launch_bounds__(256) void k_first_hit(Scene S, const double* __restrict__ rays, size_t n, double* t,
                                                   double* normal, uint32_t* prim_id, uint32_t* mat_id, uint8_t* front) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const double* r = rays + 7 * idx;
  Ray ray;
  ray.o = v3(r[0], r[1], r[2]); ray.d = v3(r[3], r[4], r[5]); ray.tm = r[6];
  Hit h;
  bool any = closest_hit<false>(S, S.world_begin, S.world_end, ray, 0, 0.001, RT_INF, h, nullptr, 0);
  Surface sf;
  sf.n = v3(0, 0, 0); sf.prim_id = 0xFFFFFFFFu; sf.mat = 0xFFFFFFFFu; sf.front = false;
  if (any) resolve_hit(S, ray, h, false, sf);
  if (t) t[idx] = any ? h.t : RT_INF;
  if (normal) { normal[3 * idx] = sf.n.x; normal[3 * idx + 1] = sf.n.y; normal[3 * idx + 2] = sf.n.z; }
  if (prim_id) prim_id[idx] = sf.prim_id;
  if (mat_id) mat_id[idx] = sf.mat;
  if (front) front[idx] = sf.front ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_resolve(const double* __restrict__ accum, size_t n, double scale, uint8_t* __restrict__ rgb) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) rgb[idx] = resolve_channel(accum[idx], scale);
}

__global__ __launch_bounds__(256) void k_add(double* __restrict__ dst, const double* __restrict__ src, size_t n) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) dst[idx] += src[idx];
}

// ------------------------------------------------------------------ host ---

static thread_local std::string g_error;

struct rtx_context {
  int device = 0;
  std::string error;
  unsigned long long* d_counters = nullptr;  // paths, segments, pdf_probes
  double render_ms = 0.0;
  uint64_t launches = 0;
  double last_kernel_ms = 0.0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;  // timed launches not yet read back
  std::vector<hipEvent_t> free_events;
};

struct rtx_scene {
  rtx_context* ctx = nullptr;
  Scene dev;  // device pointers
  std::vector<void*> allocs;
};

static rtx_status fail(rtx_context* ctx, rtx_status st, const std::string& msg) {
  if (ctx) ctx->error = msg;
  g_error = msg;
  return st;
}
#define HIP_TRY(ctx, call)                                                                                  \
  do {                                                                                                      \
    hipError_t e_ = (call);                                                                                 \
    if (e_ != hipSuccess) return fail(ctx, RTX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

static hipEvent_t get_event(rtx_context* ctx) {
  if (!ctx->free_events.empty()) { hipEvent_t e = ctx->free_events.back(); ctx->free_events.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
static void drain_events(rtx_context* ctx) {
  for (auto& p : ctx->pending) {
    float ms = 0.f;
    if (hipEventSynchronize(p.second) == hipSuccess && hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
      ctx->render_ms += ms; ctx->last_kernel_ms = ms; ctx->launches += 1;
    }
    ctx->free_events.push_back(p.first); ctx->free_events.push_back(p.second);
  }
  ctx->pending.clear();
}

extern "C" {

const char* rtx_last_error(const rtx_context* ctx) { return ctx ? ctx->error.c_str() : g_error.c_str(); }

rtx_status rtx_context_create(int n_devices, const int* device_ids, rtx_context** out) {
  if (!out) return fail(nullptr, RTX_ERR_INVALID, "rtx_context_create: out is null");
  *out = nullptr;
  if (n_devices != 1) return fail(nullptr, RTX_ERR_INVALID, "rtx_context_create: n_devices must be 1 (one context per process and GPU)");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count < 1) return fail(nullptr, RTX_ERR_NO_DEVICE, "rtx_context_create: no HIP device available (this backend has no CPU path)");
  int dev = device_ids ? device_ids[0] : 0;
  if (dev < 0 || dev >= count) return fail(nullptr, RTX_ERR_INVALID, "rtx_context_create: device id out of range");
  rtx_context* ctx = new rtx_context();
  ctx->device = dev;
  if (hipSetDevice(dev) != hipSuccess || hipMalloc(&ctx->d_counters, 3 * sizeof(unsigned long long)) != hipSuccess ||
      hipMemset(ctx->d_counters, 0, 3 * sizeof(unsigned long long)) != hipSuccess) {
    delete ctx;
    return fail(nullptr, RTX_ERR_HIP, "rtx_context_create: device initialisation failed");
  }
  *out = ctx;
  return RTX_OK;
}

void rtx_context_destroy(rtx_context* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  drain_events(ctx);
  for (hipEvent_t e : ctx->free_events) (void)hipEventDestroy(e);
  if (ctx->d_counters) (void)hipFree(ctx->d_counters);
  delete ctx;
}

}  // extern "C"

static bool check_texture_depth(const rtx_scene_desc* d, uint32_t tex, int depth) {
  if (depth > RTX_MAX_TEXTURE_DEPTH) return false;
  const rtx_texture& t = d->textures[tex];
  if (t.kind == RTX_TEX_CHECKER) return check_texture_depth(d, t.a, depth + 1) && check_texture_depth(d, t.b, depth + 1);
  return true;
}

// Everything the kernels index without bounds checks is validated here.
static const char* validate_scene(const rtx_scene_desc* d) {
  if (d->world_begin > d->world_end || d->world_end > d->n_nodes) return "world program range outside nodes";
  if (d->n_xforms < 1 || !d->xforms || d->xforms[0].n_ops != 0) return "xforms[0] must be the identity context";
  if ((d->n_nodes && !d->nodes) || (d->n_boxes && !d->boxes) || (d->n_spheres && !d->spheres) || (d->n_quads && !d->quads) ||
      (d->n_xform_ops && !d->xform_ops) || (d->n_media && !d->media) || (d->n_materials && !d->materials) ||
      (d->n_textures && !d->textures) || (d->n_images && !d->images) || (d->n_texel_bytes && !d->texels) ||
      (d->n_perlins && !d->perlins) || (d->n_lights && !d->lights))
    return "null table with a non-zero count";
  for (uint32_t i = 0; i < d->n_xforms; ++i)
    if ((uint64_t)d->xforms[i].first_op + d->xforms[i].n_ops > d->n_xform_ops) return "xform op range out of bounds";
  for (uint32_t i = 0; i < d->n_xform_ops; ++i)
    if (d->xform_ops[i].kind > RTX_XF_ROTATE_Y) return "unknown xform op kind";
  for (uint32_t i = 0; i < d->n_textures; ++i) {
    const rtx_texture& t = d->textures[i];
    if (t.kind > RTX_TEX_IMAGE) return "unknown texture kind";
    if (t.kind == RTX_TEX_CHECKER && (t.a >= d->n_textures || t.b >= d->n_textures)) return "checker child out of range";
    if (t.kind == RTX_TEX_NOISE && t.a >= d->n_perlins) return "noise texture perlin index out of range";
    if (t.kind == RTX_TEX_IMAGE && t.a >= d->n_images) return "image texture index out of range";
  }
  for (uint32_t i = 0; i < d->n_textures; ++i)
    if (!check_texture_depth(d, i, 0)) return "checker textures nested deeper than RTX_MAX_TEXTURE_DEPTH (or cyclic)";
  for (uint32_t i = 0; i < d->n_images; ++i) {
    const rtx_image& im = d->images[i];
    if (im.height && (im.width == 0 || im.offset + (uint64_t)im.width * im.height * 3 > d->n_texel_bytes)) return "image texels out of bounds";
  }
  for (uint32_t i = 0; i < d->n_perlins; ++i)
    for (int k = 0; k < 256; ++k)
      if ((uint32_t)d->perlins[i].perm_x[k] > 255u || (uint32_t)d->perlins[i].perm_y[k] > 255u || (uint32_t)d->perlins[i].perm_z[k] > 255u)
        return "perlin permutation entry outside 0..255";
  for (uint32_t i = 0; i < d->n_materials; ++i) {
    const rtx_material& m = d->materials[i];
    if (m.kind > RTX_MAT_ISOTROPIC) return "unknown material kind";
    if ((m.kind == RTX_MAT_LAMBERTIAN || m.kind == RTX_MAT_DIFFUSE_LIGHT || m.kind == RTX_MAT_ISOTROPIC) && m.tex >= d->n_textures)
      return "material texture out of range";
  }
  for (uint32_t i = 0; i < d->n_spheres; ++i) if (d->spheres[i].mat >= d->n_materials) return "sphere material out of range";
  for (uint32_t i = 0; i < d->n_quads; ++i) if (d->quads[i].mat >= d->n_materials) return "quad material out of range";
  for (uint32_t i = 0; i < d->n_media; ++i) {
    const rtx_medium& m = d->media[i];
    if (m.b_begin > m.b_end || m.b_end > d->n_nodes) return "medium boundary range outside nodes";
    if (m.mat >= d->n_materials) return "medium material out of range";
    for (uint32_t k = m.b_begin; k < m.b_end; ++k)
      if (RTX_NODE_KIND(d->nodes[k].kind_idx) == RTX_NODE_MEDIUM) return "medium inside a medium boundary";
  }
  for (uint32_t i = 0; i < d->n_nodes; ++i) {
    const uint32_t kind = RTX_NODE_KIND(d->nodes[i].kind_idx), idx = RTX_NODE_IDX(d->nodes[i].kind_idx);
    switch (kind) {
      case RTX_NODE_BOX:
        if (idx >= d->n_boxes) return "box index out of range";
        if (d->nodes[i].skip <= i || d->nodes[i].skip > d->n_nodes) return "box skip link must point forward inside nodes";
        break;
      case RTX_NODE_SPHERE: if (idx >= d->n_spheres) return "sphere index out of range"; break;
      case RTX_NODE_QUAD: if (idx >= d->n_quads) return "quad index out of range"; break;
      case RTX_NODE_ENTER: case RTX_NODE_EXIT: if (idx >= d->n_xforms) return "transform context out of range"; break;
      case RTX_NODE_MEDIUM: if (idx >= d->n_media) return "medium index out of range"; break;
      default: return "unknown node kind";
    }
  }
  for (uint32_t i = 0; i < d->n_lights; ++i) {
    const rtx_light& L = d->lights[i];
    if (L.ctx >= d->n_xforms) return "light context out of range";
    if (L.kind == RTX_NODE_SPHERE) { if (L.idx >= d->n_spheres) return "light sphere out of range"; }
    else if (L.kind == RTX_NODE_QUAD) { if (L.idx >= d->n_quads) return "light quad out of range"; }
    else if (L.kind != RTX_NODE_NONE) return "unknown light kind";
  }
  return nullptr;
}

template <class T>
static rtx_status upload(rtx_scene* sc, const T* src, size_t count, const T** dst) {
  *dst = nullptr;
  size_t bytes = count * sizeof(T);
  void* p = nullptr;
  HIP_TRY(sc->ctx, hipMalloc(&p, bytes ? bytes : 16));  // never hand the kernels a null table
  sc->allocs.push_back(p);
  if (bytes) HIP_TRY(sc->ctx, hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
  *dst = (const T*)p;
  return RTX_OK;
}

extern "C" {

rtx_status rtx_scene_create(rtx_context* ctx, const rtx_scene_desc* d, rtx_scene** out) {
  if (!ctx || !d || !out) return fail(ctx, RTX_ERR_INVALID, "rtx_scene_create: null argument");
  *out = nullptr;
  if (const char* why = validate_scene(d)) return fail(ctx, RTX_ERR_INVALID, std::string("rtx_scene_create: ") + why);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  rtx_scene* sc = new rtx_scene();
  sc->ctx = ctx;
  sc->dev = *d;
  rtx_status st = RTX_OK;
#define UP(field, count) if (st == RTX_OK) st = upload(sc, d->field, (size_t)(count), &sc->dev.field)
  UP(nodes, d->n_nodes); UP(boxes, d->n_boxes); UP(spheres, d->n_spheres); UP(quads, d->n_quads);
  UP(xforms, d->n_xforms); UP(xform_ops, d->n_xform_ops); UP(media, d->n_media); UP(materials, d->n_materials);
  UP(textures, d->n_textures); UP(images, d->n_images); UP(texels, d->n_texel_bytes); UP(perlins, d->n_perlins);
  UP(lights, d->n_lights);
#undef UP
  if (st != RTX_OK) { rtx_scene_destroy(sc); return st; }
  *out = sc;
  return RTX_OK;
}

void rtx_scene_destroy(rtx_scene* sc) {
  if (!sc) return;
  (void)hipSetDevice(sc->ctx->device);
  (void)hipDeviceSynchronize();
  for (void* p : sc->allocs) (void)hipFree(p);
  delete sc;
}

}  // extern "C"

static const char* validate_render(const rtx_camera_desc* cam, const rtx_render_opts* o) {
  if (cam->image_width < 1 || cam->image_height < 1 || cam->sqrt_spp < 1) return "camera not initialised (call rtx_camera_initialize)";
  if ((int64_t)cam->image_width * cam->image_height > 0x7FFFFFFF / 4) return "image too large";
  if (o->x0 < 0 || o->y0 < 0 || o->x1 > cam->image_width || o->y1 > cam->image_height || o->x0 > o->x1 || o->y0 > o->y1) return "pixel rectangle outside the image";
  if (o->sample_begin < 0 || o->sample_begin > o->sample_end || o->sample_end > cam->sqrt_spp * cam->sqrt_spp) return "sample range outside [0, sqrt_spp^2]";
  if (o->pipeline < RTX_PIPELINE_DEFAULT || o->pipeline > RTX_PIPELINE_WAVEFRONT) return "unknown pipeline";
  return nullptr;
}

extern "C" {

rtx_status rtx_render_device(rtx_scene* sc, const rtx_camera_desc* cam, const rtx_render_opts* o, double* accum_dev, void* hip_stream) {
  if (!sc || !cam || !o || !accum_dev) return fail(sc ? sc->ctx : nullptr, RTX_ERR_INVALID, "rtx_render_device: null argument");
  rtx_context* ctx = sc->ctx;
  if (const char* why = validate_render(cam, o)) return fail(ctx, RTX_ERR_INVALID, std::string("rtx_render_device: ") + why);
  if (o->x0 == o->x1 || o->y0 == o->y1 || o->sample_begin == o->sample_end) return RTX_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t stream = (hipStream_t)hip_stream;
  const int tiles_x = (o->x1 - o->x0 + 15) / 16, tiles_y = (o->y1 - o->y0 + 15) / 16;
  const int use_lights = (o->use_lights && sc->dev.has_lights) ? 1 : 0;
  hipEvent_t e0 = get_event(ctx), e1 = get_event(ctx);
  if (!e0 || !e1) return fail(ctx, RTX_ERR_HIP, "rtx_render_device: hipEventCreate failed");
  HIP_TRY(ctx, hipEventRecord(e0, stream));
  hipLaunchKernelGGL(k_render_fused, dim3((unsigned)(tiles_x * tiles_y)), dim3(256), 0, stream, sc->dev, *cam, *o, use_lights, accum_dev, ctx->d_counters);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipEventRecord(e1, stream));
  ctx->pending.emplace_back(e0, e1);
  if (ctx->pending.size() > 4096) drain_events(ctx);
  return RTX_OK;
}

rtx_status rtx_render(rtx_scene* sc, const rtx_camera_desc* cam, const rtx_render_opts* o, double* accum_host) {
  if (!sc || !cam || !o || !accum_host) return fail(sc ? sc->ctx : nullptr, RTX_ERR_INVALID, "rtx_render: null argument");
  rtx_context* ctx = sc->ctx;
  if (const char* why = validate_render(cam, o)) return fail(ctx, RTX_ERR_INVALID, std::string("rtx_render: ") + why);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t n = (size_t)cam->image_width * (size_t)cam->image_height * 3;
  double* d = nullptr;
  HIP_TRY(ctx, hipMalloc((void**)&d, n * sizeof(double)));
  rtx_status st = RTX_OK;
  std::vector<double> tmp;
  do {
    if (hipMemset(d, 0, n * sizeof(double)) != hipSuccess) { st = fail(ctx, RTX_ERR_HIP, "rtx_render: hipMemset failed"); break; }
    st = rtx_render_device(sc, cam, o, d, nullptr);
    if (st != RTX_OK) break;
    if (hipDeviceSynchronize() != hipSuccess) { st = fail(ctx, RTX_ERR_HIP, std::string("rtx_render: kernel failed: ") + hipGetErrorString(hipGetLastError())); break; }
    tmp.resize(n);
    if (hipMemcpy(tmp.data(), d, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { st = fail(ctx, RTX_ERR_HIP, "rtx_render: copy back failed"); break; }
    for (size_t k = 0; k < n; ++k) accum_host[k] += tmp[k];
  } while (0);
  (void)hipFree(d);
  return st;
}

rtx_status rtx_resolve_rgb8_device(rtx_context* ctx, const double* accum_dev, int width, int height, int spp, uint8_t* rgb_dev, void* hip_stream) {
  if (!ctx || !accum_dev || !rgb_dev || width < 1 || height < 1 || spp < 1) return fail(ctx, RTX_ERR_INVALID, "rtx_resolve_rgb8_device: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  size_t n = (size_t)width * (size_t)height * 3;
  hipLaunchKernelGGL(k_resolve, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, accum_dev, n, 1.0 / (double)spp, rgb_dev);
  HIP_TRY(ctx, hipGetLastError());
  return RTX_OK;
}

rtx_status rtx_resolve_rgb8(rtx_context* ctx, const double* accum_host, int width, int height, int spp, uint8_t* rgb_host) {
  if (!ctx || !accum_host || !rgb_host || width < 1 || height < 1 || spp < 1) return fail(ctx, RTX_ERR_INVALID, "rtx_resolve_rgb8: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  size_t n = (size_t)width * (size_t)height * 3;
  double* d = nullptr;
  uint8_t* r = nullptr;
  HIP_TRY(ctx, hipMalloc((void**)&d, n * sizeof(double)));
  rtx_status st = RTX_OK;
  do {
    if (hipMalloc((void**)&r, n) != hipSuccess) { st = fail(ctx, RTX_ERR_HIP, "rtx_resolve_rgb8: hipMalloc failed"); break; }
    if (hipMemcpy(d, accum_host, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { st = fail(ctx, RTX_ERR_HIP, "rtx_resolve_rgb8: upload failed"); break; }
    st = rtx_resolve_rgb8_device(ctx, d, width, height, spp, r, nullptr);
    if (st != RTX_OK) break;
    if (hipMemcpy(rgb_host, r, n, hipMemcpyDeviceToHost) != hipSuccess) { st = fail(ctx, RTX_ERR_HIP, "rtx_resolve_rgb8: copy back failed"); break; }
  } while (0);
  (void)hipFree(d);
  if (r) (void)hipFree(r);
  return st;
}

rtx_status rtx_first_hit_aov(rtx_scene* sc, const double* rays, size_t n, double* t, double* normal, uint32_t* prim_id, uint32_t* mat_id, uint8_t* front) {
  if (!sc || (n && !rays)) return fail(sc ? sc->ctx : nullptr, RTX_ERR_INVALID, "rtx_first_hit_aov: null argument");
  rtx_context* ctx = sc->ctx;
  if (n == 0) return RTX_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  double *d_rays = nullptr, *d_t = nullptr, *d_n = nullptr;
  uint32_t *d_p = nullptr, *d_m = nullptr;
  uint8_t* d_f = nullptr;
  rtx_status st = RTX_OK;
  auto H = [&](hipError_t e, const char* what) { if (e != hipSuccess && st == RTX_OK) st = fail(ctx, RTX_ERR_HIP, std::string("rtx_first_hit_aov: ") + what + ": " + hipGetErrorString(e)); return e == hipSuccess; };
  do {
    if (!H(hipMalloc((void**)&d_rays, n * 7 * sizeof(double)), "hipMalloc")) break;
    if (!H(hipMemcpy(d_rays, rays, n * 7 * sizeof(double), hipMemcpyHostToDevice), "upload")) break;
    if (t && !H(hipMalloc((void**)&d_t, n * sizeof(double)), "hipMalloc")) break;
    if (normal && !H(hipMalloc((void**)&d_n, n * 3 * sizeof(double)), "hipMalloc")) break;
    if (prim_id && !H(hipMalloc((void**)&d_p, n * sizeof(uint32_t)), "hipMalloc")) break;
    if (mat_id && !H(hipMalloc((void**)&d_m, n * sizeof(uint32_t)), "hipMalloc")) break;
    if (front && !H(hipMalloc((void**)&d_f, n), "hipMalloc")) break;
    hipLaunchKernelGGL(k_first_hit, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, sc->dev, d_rays, n, d_t, d_n, d_p, d_m, d_f);
    if (!H(hipGetLastError(), "launch")) break;
    if (!H(hipDeviceSynchronize(), "kernel")) break;
    if (t && !H(hipMemcpy(t, d_t, n * sizeof(double), hipMemcpyDeviceToHost), "copy back")) break;
    if (normal && !H(hipMemcpy(normal, d_n, n * 3 * sizeof(double), hipMemcpyDeviceToHost), "copy back")) break;
    if (prim_id && !H(hipMemcpy(prim_id, d_p, n * sizeof(uint32_t), hipMemcpyDeviceToHost), "copy back")) break;
    if (mat_id && !H(hipMemcpy(mat_id, d_m, n * sizeof(uint32_t), hipMemcpyDeviceToHost), "copy back")) break;
    if (front && !H(hipMemcpy(front, d_f, n, hipMemcpyDeviceToHost), "copy back")) break;
  } while (0);
  (void)hipFree(d_rays); (void)hipFree(d_t); (void)hipFree(d_n); (void)hipFree(d_p); (void)hipFree(d_m); (void)hipFree(d_f);
  return st;
}

rtx_status rtx_get_stats(rtx_context* ctx, rtx_stats* out) {
  if (!ctx || !out) return fail(ctx, RTX_ERR_INVALID, "rtx_get_stats: null argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  drain_events(ctx);
  unsigned long long c[3] = {0, 0, 0};
  HIP_TRY(ctx, hipMemcpy(c, ctx->d_counters, sizeof(c), hipMemcpyDeviceToHost));
  out->paths = c[0]; out->segments = c[1]; out->pdf_probes = c[2];
  out->render_ms = ctx->render_ms; out->launches = ctx->launches; out->last_kernel_ms = ctx->last_kernel_ms;
  return RTX_OK;
}

rtx_status rtx_reset_stats(rtx_context* ctx) {
  if (!ctx) return fail(ctx, RTX_ERR_INVALID, "rtx_reset_stats: null context");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  drain_events(ctx);
  HIP_TRY(ctx, hipMemset(ctx->d_counters, 0, 3 * sizeof(unsigned long long)));
  ctx->render_ms = 0.0; ctx->launches = 0; ctx->last_kernel_ms = 0.0;
  return RTX_OK;
}

}  // extern "C"
