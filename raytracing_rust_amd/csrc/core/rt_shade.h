// One bounce of the reference integrator (Camera::ray_color,
// src/ray_tracing/rendering/camera.rs:167-247) in iterative form, with the
// materials, textures and PDFs it calls.
//
// The recursion  L = Le + a*(spdf*Li)/pdf  becomes  L += T*Le;  T *= a*spdf/pdf
// carried in a PathState; a path ends exactly where the recursion returns
// without recursing.
#pragma once
#include "rt_traverse.h"

namespace rt {

// --------------------------------------------------------------- textures ---

// procedural/noise.rs:96-127
RT_HD double perlin_interp(const V3 c[2][2][2], double u, double v, double w) {
  double uu = u * u * (3.0 - 2.0 * u);
  double vv = v * v * (3.0 - 2.0 * v);
  double ww = w * w * (3.0 - 2.0 * w);
  double accum = 0.0;
RT_UNROLL
  for (int i = 0; i < 2; ++i)
RT_UNROLL
    for (int j = 0; j < 2; ++j)
RT_UNROLL
      for (int k = 0; k < 2; ++k) {
        double fi = i, fj = j, fk = k;
        V3 wv = v3(u - fi, v - fj, w - fk);
        double weight = (fi * uu + (1.0 - fi) * (1.0 - uu)) * (fj * vv + (1.0 - fj) * (1.0 - vv)) *
                        (fk * ww + (1.0 - fk) * (1.0 - ww));
        accum += weight * dot(c[i][j][k], wv);
      }
  return accum;
}
// procedural/noise.rs:35-60
RT_HD double perlin_noise(const rtx_perlin& P, const V3& p) {
  double fx = floor(p.x), fy = floor(p.y), fz = floor(p.z);
  double u = p.x - fx, v = p.y - fy, w = p.z - fz;
  int i = (int)fx, j = (int)fy, k = (int)fz;
  V3 c[2][2][2];
RT_UNROLL
  for (int di = 0; di < 2; ++di)
RT_UNROLL
    for (int dj = 0; dj < 2; ++dj)
RT_UNROLL
      for (int dk = 0; dk < 2; ++dk) {
        int idx = P.perm_x[(i + di) & 255] ^ P.perm_y[(j + dj) & 255] ^ P.perm_z[(k + dk) & 255];
        c[di][dj][dk] = v3(P.ranvec[3 * idx], P.ranvec[3 * idx + 1], P.ranvec[3 * idx + 2]);
      }
  return perlin_interp(c, u, v, w);
}
// procedural/noise.rs:64-76
RT_HD double perlin_turb(const rtx_perlin& P, const V3& p, int depth) {
  double accum = 0.0, weight = 1.0;
  V3 tp = p;
  for (int i = 0; i < depth; ++i) {
    accum += weight * perlin_noise(P, tp);
    weight *= 0.5;
    tp = tp * 2.0;
  }
  return fabs(accum);
}

// Texture::value for the four texture kinds (texture/solid_color.rs:18-22, checker.rs:41-56, noise.rs:29-35,
// image.rs:58-88).  Checker recursion is a bounded descent (depth validated when the scene is created).
RT_HD V3 texture_value(const Scene& S, uint32_t tex, double u, double v, const V3& p) {
  for (int depth = 0; depth <= RTX_MAX_TEXTURE_DEPTH; ++depth) {
    const rtx_texture t = S.textures[tex];
    if (t.kind == RTX_TEX_SOLID) return v3(t.color);
    if (t.kind == RTX_TEX_CHECKER) {
      int xi = (int)floor(t.scale * p.x);
      int yi = (int)floor(t.scale * p.y);
      int zi = (int)floor(t.scale * p.z);
      bool is_even = (xi + yi + zi) % 2 == 0;
      tex = is_even ? t.a : t.b;
      continue;
    }
    if (t.kind == RTX_TEX_NOISE) {
      double nv = 1.0 + sin(t.scale * p.z + 10.0 * perlin_turb(S.perlins[t.a], p, 7));
      return v3(0.5, 0.5, 0.5) * nv;
    }
    // IMAGE
    const rtx_image im = S.images[t.a];
    if (im.height == 0) return v3(0.0, 1.0, 1.0);
    double uc = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);
    double vcl = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
    double vc = 1.0 - vcl;
    uint32_t i = (uint32_t)(uc * (double)im.width);
    uint32_t j = (uint32_t)(vc * (double)im.height);
    if (i > im.width - 1) i = im.width - 1;
    if (j > im.height - 1) j = im.height - 1;
    const uint8_t* px = S.texels + im.offset + ((uint64_t)j * im.width + i) * 3;
    double s = 1.0 / 255.0;
    return v3(s * (double)px[0], s * (double)px[1], s * (double)px[2]);
  }
  return v3(0.0, 0.0, 0.0);
}

// ------------------------------------------------------------ light leaves ---

// Point into a light leaf's transform context (translate.rs:57,64; rotate_y.rs:137,145).
RT_HD V3 ctx_point_to_local(const Scene& S, uint32_t ctx, V3 p) {
  const rtx_xform x = S.xforms[ctx];
  for (uint32_t k = 0; k < x.n_ops; ++k) {
    const rtx_xform_op op = S.xform_ops[x.first_op + k];
    if (op.kind == RTX_XF_TRANSLATE) p = p - v3(op.a, op.b, op.c);
    else p = rot_w2l(p, op.a, op.b);
  }
  return p;
}
RT_HD V3 ctx_dir_to_local(const Scene& S, uint32_t ctx, V3 d) {
  const rtx_xform x = S.xforms[ctx];
  for (uint32_t k = 0; k < x.n_ops; ++k) {
    const rtx_xform_op op = S.xform_ops[x.first_op + k];
    if (op.kind == RTX_XF_ROTATE_Y) d = rot_w2l(d, op.a, op.b);
  }
  return d;
}

// Sphere::pdf_value, sphere.rs:125-141 (probe ray at time 0)
RT_HD double sphere_pdf_value(const rtx_sphere& s, const V3& origin, const V3& dir) {
  double t;
  if (!sphere_hit_t(s, origin, dir, 0.0, 0.001, RT_INF, t)) return 0.0;
  V3 cc = v3(s.c0) + v3(s.dc) * 0.0;
  double dist2 = norm2(cc - origin);
  double cos_theta_max = sqrt(1.0 - s.radius * s.radius / dist2);
  double solid_angle = 2.0 * RT_PI * (1.0 - cos_theta_max);
  return 1.0 / solid_angle;
}
// Sphere::random + random_to_sphere, sphere.rs:69-79,143-150
RT_HD V3 sphere_random(const rtx_sphere& s, const V3& origin, PathRng& rng) {
  V3 cc = v3(s.c0) + v3(s.dc) * 0.0;
  V3 direction = cc - origin;
  double dist2 = norm2(direction);
  Onb onb = onb_from(direction);
  double r1, r2;
  rng_next2(rng, r1, r2);
  double z = 1.0 + r2 * (sqrt(1.0 - s.radius * s.radius / dist2) - 1.0);
  double phi = 2.0 * RT_PI * r1;
  double sq = sqrt(1.0 - z * z);
  double sn, cs;
  rt_sincos(phi, sn, cs);
  double x = cs * sq;
  double y = sn * sq;
  return onb_local(onb, v3(x, y, z));
}
// Quad::pdf_value, quad.rs:102-116
RT_HD double quad_pdf_value(const rtx_quad& q, const V3& origin, const V3& dir) {
  double t;
  if (!quad_hit_t(q, origin, dir, 0.001, RT_INF, t)) return 0.0;
  double dist2 = t * t * norm2(dir);
  // rec.normal is +-normal; only its absolute cosine is used
  double cosine = fabs(dot(dir, v3(q.normal)) / norm(dir));
  return dist2 / (cosine * q.area);
}
// Quad::random, quad.rs:118-121
RT_HD V3 quad_random(const rtx_quad& q, const V3& origin, PathRng& rng) {
  double a, b;
  rng_next2(rng, a, b);
  V3 p = v3(q.q) + (a * v3(q.u)) + (b * v3(q.v));
  return p - origin;
}

// HittablePDF::value over the flattened light leaves (hittable_pdf.rs:25-27 -> hittable_list.rs:78-88 etc.)
RT_HD double light_pdf_term(const Scene& S, uint32_t i, const V3& origin, const V3& dir, uint32_t& probes) {
  const rtx_light L = S.lights[i];
  double pv = 0.0;
  if (L.kind != RTX_NODE_NONE) {
    V3 o = origin, d = dir;
    if (L.ctx != 0) { o = ctx_point_to_local(S, L.ctx, o); d = ctx_dir_to_local(S, L.ctx, d); }
    if (L.kind == RTX_NODE_QUAD) pv = quad_pdf_value(S.light_quads ? S.light_quads[i] : S.quads[L.idx], o, d);
    else pv = sphere_pdf_value(S.light_spheres ? S.light_spheres[i] : S.spheres[L.idx], o, d);
    ++probes;
  }
  return L.weight * pv;
}
RT_HD double lights_pdf_value(const Scene& S, const V3& origin, const V3& dir, uint32_t& probes) {
  double sum = 0.0;
  const bool table = S.n_light_nodes != 0 && fabs(origin.x) <= S.light_far && fabs(origin.y) <= S.light_far && fabs(origin.z) <= S.light_far;
  if (!table) {
    for (uint32_t i = 0; i < S.n_lights; ++i) sum += light_pdf_term(S, i, origin, dir, probes);
    return sum;
  }
  // With many lights: only those whose widened box the probe ray passes over [0.001, inf) are evaluated - every other
  // light's pdf_value is exactly 0 (its hit test fails), and adding w * 0 leaves the sum as it is - in the list's order
  // (hittable_list.rs:84-86), group of 64 by group of 64, so the sum is the reference's to the last bit.
  const LRay lr = lray_from_ops(nullptr, 0, origin, dir);
  for (uint32_t g = 0; g < S.n_light_groups; ++g) {
    unsigned long long mask = 0ull;
    uint32_t k = S.light_group_begin[g];
    const uint32_t end = S.light_group_begin[g + 1u];
    while (k < end) {
      const LightNode* nd = S.light_nodes + k;
      const double b[6] = {nd->lo[0], nd->lo[1], nd->lo[2], nd->hi[0], nd->hi[1], nd->hi[2]};
      const uint32_t skip = nd->skip, light = nd->light;
      if (slab_hit(b, lr, 0.001, RT_INF)) {
        if (light != LIGHT_NODE_INNER) mask |= 1ull << (light - 64u * g);
        k += 1u;
      } else {
        k = skip;
      }
    }
    while (mask) {
      const uint32_t bit = (uint32_t)__builtin_ctzll(mask);
      mask &= mask - 1ull;
      sum += light_pdf_term(S, 64u * g + bit, origin, dir, probes);
    }
  }
  return sum;
}
// HittablePDF::generate (hittable_pdf.rs:30-32 -> hittable_list.rs:90-97 etc.)
RT_HD V3 lights_random(const Scene& S, const V3& origin, PathRng& rng) {
  if (S.n_lights == 0) return v3(1.0, 0.0, 0.0);
  uint32_t pick = 0;
  if (S.n_lights > 1) {
    double xi = rng_next(rng);
    pick = S.n_lights - 1;
    for (uint32_t i = 0; i + 1 < S.n_lights; ++i)
      if (xi < S.lights[i].cum) { pick = i; break; }
  }
  const rtx_light L = S.lights[pick];
  V3 o = origin;
  if (L.ctx != 0) o = ctx_point_to_local(S, L.ctx, o);
  V3 d;
  if (L.kind == RTX_NODE_QUAD) d = quad_random(S.light_quads ? S.light_quads[pick] : S.quads[L.idx], o, rng);
  else if (L.kind == RTX_NODE_SPHERE) d = sphere_random(S.light_spheres ? S.light_spheres[pick] : S.spheres[L.idx], o, rng);
  else d = v3(1.0, 0.0, 0.0);
  if (L.ctx != 0) d = ctx_dir_to_world(S, L.ctx, d);
  return d;
}

// -------------------------------------------------------------- integrator ---

// Optional cycle accounting of the shading code (a tuning build defines RT_SPROF; the product does not).
#ifndef RT_SPROF
#define RT_SPROF_START() do { } while (0)
#define RT_SPROF(slot) do { } while (0)
#endif

struct PathState {
  V3 T;           // product of attenuation*spdf/pdf factors so far
  V3 L;           // radiance gathered so far
  int depth;      // remaining depth, counts down from max_depth (camera.rs:174,197,237)
  uint32_t bounce;
  PathRng rng;
};

RT_HD void path_begin(PathState& ps, int max_depth, const PathRng& rng) {
  ps.T = v3(1.0, 1.0, 1.0); ps.L = v3(0.0, 0.0, 0.0); ps.depth = max_depth; ps.bounce = 0; ps.rng = rng;
}

// Dielectric::reflectance, dielectric.rs:23-27 (powi(5) = b * (b^2)^2)
RT_HD double schlick(double cosine, double ratio) {
  double r0 = (1.0 - ratio) / (1.0 + ratio);
  double r0s = r0 * r0;
  double b = 1.0 - cosine;
  double b2 = b * b, b4 = b2 * b2, b5 = b4 * b;
  return r0s + (1.0 - r0s) * b5;
}

// The roulette draw (camera.rs:228) of a Lambertian or isotropic hit: the first block of the bounce's shading
// stream.  A function of the path's coordinates only, so the wavefront shading kernel can evaluate it ahead of
// shade_bounce and drop the paths it ends.
RT_HD double shade_roulette_draw(PathRng rng, uint32_t bounce) {
#if defined(RT_RNG_TAPE) && !defined(__HIPCC__)
  if (RT_RNG_TAPE_ACTIVE) {
    if (bounce < rt_rng_tape->n_bounces) return rt_rng_tape->roulette[bounce];
    rt_rng_tape->overrun++;
    return 0.5;
  }
#endif
  rng_stream(rng, bounce, RNG_DOMAIN_SHADE);
  return rng_next(rng);
}

// Everything ray_color does after world.hit returned (camera.rs:180-246).  Returns true when the path goes on
// with `next`; false when it ended (ps.L is final).  A hit of kind NONE is a miss (camera.rs:179-181).
RT_HD bool shade_bounce(const Scene& S, const V3& background, bool use_lights, const Ray& wray,
                        const Hit& hit, const double* stale_uv, PathState& ps, Ray& next, uint32_t& probes) {
  RT_SPROF_START();
  if (!hit_any(hit)) { ps.L = ps.L + cmul(ps.T, background); RT_SPROF(1); return false; }
  rng_stream(ps.rng, ps.bounce, RNG_DOMAIN_SHADE);
  const uint32_t hkind = RTX_NODE_KIND(hit.kind_idx);
  uint32_t mat_idx;
  if (hkind == RTX_NODE_SPHERE) mat_idx = S.spheres[RTX_NODE_IDX(hit.kind_idx)].mat;
  else if (hkind == RTX_NODE_QUAD) mat_idx = S.quads[RTX_NODE_IDX(hit.kind_idx)].mat;
  else mat_idx = S.media[RTX_NODE_IDX(hit.kind_idx)].mat;
  const rtx_material m = S.materials[mat_idx];
  Surface sf;
  RT_SPROF(2);
  resolve_hit(S, wray, hit, m.needs_uv != 0, stale_uv, sf);
  RT_SPROF(3);

  // emission (camera.rs:184; diffuse_light.rs:37-39: both faces emit)
  if (m.kind == RTX_MAT_DIFFUSE_LIGHT) {
    V3 e = texture_value(S, m.tex, sf.u, sf.v, sf.p);
    ps.L = ps.L + cmul(ps.T, e);
    RT_SPROF(4);
    return false;                                   // scatter() == false (camera.rs:188-190)
  }
  if (m.kind == RTX_MAT_NONE) return false;          // material.rs:74-76

  if (m.kind == RTX_MAT_METAL) {                     // metal.rs:25-37
    RT_SPROF(15);
    V3 reflected = reflect(normalize(wray.d), sf.n);
    V3 dir = reflected + m.param * random_in_unit_sphere(ps.rng);
    RT_SPROF(5);
    if (dot(dir, sf.n) <= 0.0) return false;
    ps.T = cmul(ps.T, v3(m.albedo));                 // camera.rs:193-200, no roulette on this branch
    next.o = sf.p; next.d = dir; next.tm = wray.tm;
    ps.depth -= 1; ps.bounce += 1;
    return ps.depth > 0;                             // camera.rs:174-176: depth 0 contributes nothing
  }
  if (m.kind == RTX_MAT_DIELECTRIC) {                // dielectric.rs:31-54
    RT_SPROF(15);
    double ri = sf.front ? 1.0 / m.param : m.param;
    V3 unit = normalize(wray.d);
    double cos_theta = fmin(dot(-unit, sf.n), 1.0);
    double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
    bool cannot_refract = ri * sin_theta > 1.0;
    V3 dir;
    if (cannot_refract || schlick(cos_theta, ri) > rng_next(ps.rng)) dir = reflect(unit, sf.n);
    else dir = refract(unit, sf.n, ri);
    next.o = sf.p; next.d = dir; next.tm = wray.tm;  // attenuation (1,1,1)
    RT_SPROF(6);
    ps.depth -= 1; ps.bounce += 1;
    return ps.depth > 0;
  }

  // LAMBERTIAN (lambertian.rs:31-46) or ISOTROPIC (isotropic.rs:31-43)
  const bool lambert = m.kind == RTX_MAT_LAMBERTIAN;
  // Russian roulette (camera.rs:227-231) is decided FIRST.  The reference draws it after the direction and the pdf
  // check, but a path it ends returns the hit's emission either way, which is zero for these two materials
  // (material.rs:59-61), and the draw is independent of everything else: the estimator is the same, and a path that
  // is going to be ended skips the texture, the sampling and the light probes.
  if (ps.depth > 3 && shade_roulette_draw(ps.rng, ps.bounce) > 0.8) return false;
  ps.rng.n = 1;
  RT_SPROF(15);
  V3 atten = texture_value(S, m.tex, sf.u, sf.v, sf.p);
  RT_SPROF(7);
  Onb onb;
  if (lambert) onb = onb_from(sf.n);                 // CosinePDF::new, cosine_pdf.rs:14-18

  // direction from the mixture (camera.rs:204-216; mixture_pdf.rs:45-52: pdf1 = lights, weight 0.5)
  V3 dir;
  bool from_light = false;
  if (use_lights) from_light = rng_next(ps.rng) < 0.5;
  if (from_light) dir = lights_random(S, sf.p, ps.rng);
  else if (lambert) dir = onb_local(onb, random_cosine_direction(ps.rng));   // cosine_pdf.rs:29-31
  else dir = random_unit_vector(ps.rng);                                     // sphere_pdf.rs:24-26

  RT_SPROF(8);
  // density of that direction
  double mat_pdf;
  if (lambert) {                                     // cosine_pdf.rs:23-26
    double c = dot(normalize(dir), onb.w);
    mat_pdf = fmax(0.0, c / RT_PI);
  } else {
    mat_pdf = 1.0 / (4.0 * RT_PI);                   // sphere_pdf.rs:18-21
  }
  double pdf_value;
  if (use_lights) {                                  // mixture_pdf.rs:36-42
    double lp = lights_pdf_value(S, sf.p, dir, probes);
    double value = 0.5 * lp + 0.5 * mat_pdf;
    pdf_value = value < 1e-8 ? 1e-8 : value;
  } else {
    pdf_value = mat_pdf;
  }
  RT_SPROF(9);
  if (pdf_value < 1e-6 || !isfinite(pdf_value)) return false;                // camera.rs:219-221

  double spdf;
  if (lambert) {                                     // lambertian.rs:39-46
    double c = dot(sf.n, normalize(dir));
    spdf = c < 0.0 ? 0.0 : c / RT_PI;
  } else {
    spdf = 1.0 / (4.0 * RT_PI);                      // isotropic.rs:40-43
  }

  V3 f = atten * spdf;
  if (ps.depth > 3) f = f * (1.0 / 0.8);             // camera.rs:232-240 (the roulette itself was decided above)
  ps.T = cmul(ps.T, f / pdf_value);
  next.o = sf.p; next.d = dir; next.tm = wray.tm;   // camera.rs:223
  ps.depth -= 1; ps.bounce += 1;
  RT_SPROF(10);
  return ps.depth > 0;
}

// ------------------------------------------------------------------ camera ---

// Camera::get_ray with stratified jitter, camera.rs:133-164.  (i, j) pixel, sample index -> (s_i, s_j) as in
// camera.rs:262-263.  Draw order: jitter x, jitter y, [lens], time.
RT_HD Ray camera_ray(const rtx_camera_desc& c, int i, int j, int sample_idx, PathRng& rng) {
  rng_stream(rng, 0, RNG_DOMAIN_CAMERA);
  int s_i = sample_idx / c.sqrt_spp, s_j = sample_idx % c.sqrt_spp;
  double jx, jy;
  rng_next2(rng, jx, jy);
  double ox = ((double)s_i + jx) * c.recip_sqrt_spp - 0.5;
  double oy = ((double)s_j + jy) * c.recip_sqrt_spp - 0.5;
  V3 pixel_sample = v3(c.pixel00_loc) + (((double)i + ox) * v3(c.pixel_delta_u)) + (((double)j + oy) * v3(c.pixel_delta_v));
  V3 origin;
  if (c.defocus_angle <= 0.0) origin = v3(c.center);
  else {
    V3 p = random_in_unit_disk(rng);
    origin = v3(c.center) + (p.x * v3(c.defocus_disk_u)) + (p.y * v3(c.defocus_disk_v));
  }
  Ray r;
  r.o = origin; r.d = pixel_sample - origin; r.tm = rng_range(rng, 0.0, 1.0);
  return r;
}

// Test-only hook: a build may define RT_TRACE_RAY(ray) to record the ray of every segment of a path.
#ifndef RT_TRACE_RAY
#define RT_TRACE_RAY(ray) ((void)0)
#endif

// ray_color on one ray (camera.rs:167-247) as a loop over bounces.
RT_HD V3 trace_ray(const Scene& S, const rtx_camera_desc& cam, bool use_lights, Ray ray, const PathRng& rng,
                   uint32_t& segments, uint32_t& probes) {
  PathState ps;
  path_begin(ps, cam.max_depth, rng);
  const V3 bg = v3(cam.background);
  bool alive = ps.depth > 0;
  while (alive) {
    Hit hit;
    double stale_uv[2] = {0.0, 0.0};
    ++segments;
    RT_TRACE_RAY(ray);
    RayRegs rs;
    rs.r = ray;
    closest_hit<true>(S, S.prog, S.world_begin, true, rs, ray.tm, 0.001, RT_INF, hit, true, ps.rng, ps.bounce, stale_uv);
    Ray next;
    alive = shade_bounce(S, bg, use_lights, ray, hit, stale_uv, ps, next, probes);
    ray = next;
  }
  return ps.L;
}

// One whole camera path (calculate_pixel_color's closure body, camera.rs:261-266).
RT_HD V3 trace_path(const Scene& S, const rtx_camera_desc& cam, bool use_lights, int i, int j, int sample_idx,
                    uint64_t seed, uint32_t& segments, uint32_t& probes) {
  PathRng rng = rng_make(seed, (uint32_t)(j * cam.image_width + i), (uint32_t)sample_idx);
  Ray ray = camera_ray(cam, i, j, sample_idx, rng);
  return trace_ray(S, cam, use_lights, ray, rng, segments, probes);
}

// color_to_rgb_with_samples, rendering/color.rs:7-52, one channel.
RT_HD uint8_t resolve_channel(double c, double scale) {
  if (c != c) c = 0.0;
  c *= scale;
  c = c > 0.0 ? sqrt(c) : 0.0;
  c = c < 0.0 ? 0.0 : (c > 0.999 ? 0.999 : c);
  return (uint8_t)(256.0 * c);
}

}  // namespace rt
