// Closest-hit query over a flattened hit program (see rtx.h: rtx_node).
//
// The reference walks a graph of trait objects recursively
// (hittable_list.rs:53-67, bvh.rs:87-105, translate.rs:35-47,
// rotate_y.rs:111-127, constant_medium.rs:42-90).  Its visiting order is fixed
// (list order, left child before right child) and the only state that flows
// between siblings is the shrinking upper bound of the ray interval.  The hit
// program keeps exactly that: nodes in visiting order, one running tmax, box
// nodes that jump over their subtree when the slab test fails.  Comparisons are
// written with the reference's polarity (strict for spheres, inclusive for
// quads, `<`/`<=` as in the source) so that NaNs and ties resolve identically.
#pragma once
#include "../../../include/rtx.h"
#include "rt_math.h"
#include "rt_rng.h"

namespace rt {

typedef rtx_scene_desc Scene;  // on the device the pointers are device pointers

struct LRay {     // ray in the current transform context + per-axis reciprocals (aabb.rs:146)
  V3 o, d, inv;
  uint32_t skip;  // bit k set: |d_k| < 1e-8, axis k is skipped by the slab test (aabb.rs:142-144)
};

struct Hit {
  double t;
  uint32_t kind_idx;  // RTX_NODE_MAKE(SPHERE|QUAD|MEDIUM, table index); kind NONE = miss
  uint32_t ctx;       // transform context the hit was found in
  double u, v;        // QUAD: alpha, beta.  MEDIUM with needs_uv: stale (u,v) of the record (constant_medium.rs:81-87)
};

// World -> context: apply the chain outermost first (translate.rs:37, rotate_y.rs:113-115).
RT_HD void ctx_to_local(const Scene& S, uint32_t ctx, V3& o, V3& d) {
  const rtx_xform x = S.xforms[ctx];
  for (uint32_t k = 0; k < x.n_ops; ++k) {
    const rtx_xform_op op = S.xform_ops[x.first_op + k];
    if (op.kind == RTX_XF_TRANSLATE) {
      o = o - v3(op.a, op.b, op.c);
    } else {
      o = rot_w2l(o, op.a, op.b);
      d = rot_w2l(d, op.a, op.b);
    }
  }
}
// Context -> world for a hit point and normal: innermost first (rotate_y.rs:123-124, translate.rs:45).
RT_HD void ctx_to_world(const Scene& S, uint32_t ctx, V3& p, V3& n) {
  const rtx_xform x = S.xforms[ctx];
  for (uint32_t k = x.n_ops; k > 0; --k) {
    const rtx_xform_op op = S.xform_ops[x.first_op + k - 1];
    if (op.kind == RTX_XF_TRANSLATE) {
      p = p + v3(op.a, op.b, op.c);
    } else {
      p = rot_l2w(p, op.a, op.b);
      n = rot_l2w(n, op.a, op.b);
    }
  }
}
// Context -> world for a direction (rotate_y.rs:143-149; Translate::random returns the child's direction unchanged).
RT_HD V3 ctx_dir_to_world(const Scene& S, uint32_t ctx, V3 d) {
  const rtx_xform x = S.xforms[ctx];
  for (uint32_t k = x.n_ops; k > 0; --k) {
    const rtx_xform_op op = S.xform_ops[x.first_op + k - 1];
    if (op.kind == RTX_XF_ROTATE_Y) d = rot_l2w(d, op.a, op.b);
  }
  return d;
}

RT_HD LRay lray_make(const Scene& S, uint32_t ctx, const V3& wo, const V3& wd) {
  LRay l;
  l.o = wo; l.d = wd;
  if (ctx != 0) ctx_to_local(S, ctx, l.o, l.d);
  l.inv = v3(1.0 / l.d.x, 1.0 / l.d.y, 1.0 / l.d.z);
  l.skip = (fabs(l.d.x) < 1e-8 ? 1u : 0u) | (fabs(l.d.y) < 1e-8 ? 2u : 0u) | (fabs(l.d.z) < 1e-8 ? 4u : 0u);
  return l;
}

// Aabb::hit, aabb.rs:132-167.  The per-axis early return is folded into one final comparison: the interval only
// narrows, so it is empty at the end exactly when it was empty after some axis.
RT_HD bool slab_hit(const rtx_aabb& b, const LRay& r, double tmin, double tmax) {
  if (r.skip == 0) {
    double t0 = (b.lo[0] - r.o.x) * r.inv.x, t1 = (b.hi[0] - r.o.x) * r.inv.x;
    bool pos = r.inv.x >= 0.0;
    tmin = fmax(tmin, pos ? t0 : t1); tmax = fmin(tmax, pos ? t1 : t0);
    t0 = (b.lo[1] - r.o.y) * r.inv.y; t1 = (b.hi[1] - r.o.y) * r.inv.y;
    pos = r.inv.y >= 0.0;
    tmin = fmax(tmin, pos ? t0 : t1); tmax = fmin(tmax, pos ? t1 : t0);
    t0 = (b.lo[2] - r.o.z) * r.inv.z; t1 = (b.hi[2] - r.o.z) * r.inv.z;
    pos = r.inv.z >= 0.0;
    tmin = fmax(tmin, pos ? t0 : t1); tmax = fmin(tmax, pos ? t1 : t0);
    return !(tmax <= tmin);
  }
  const double o[3] = {r.o.x, r.o.y, r.o.z}, inv[3] = {r.inv.x, r.inv.y, r.inv.z};
  for (int a = 0; a < 3; ++a) {
    if (r.skip & (1u << a)) continue;
    double t0 = (b.lo[a] - o[a]) * inv[a], t1 = (b.hi[a] - o[a]) * inv[a];
    bool pos = inv[a] >= 0.0;
    tmin = fmax(tmin, pos ? t0 : t1); tmax = fmin(tmax, pos ? t1 : t0);
    if (tmax <= tmin) return false;
  }
  return true;
}

// Sphere::hit up to the accepted root, sphere.rs:83-104.
RT_HD bool sphere_hit_t(const rtx_sphere& s, const V3& o, const V3& d, double time, double tmin, double tmax, double& t) {
  V3 center = v3(s.c0) + v3(s.dc) * time;
  V3 oc = o - center;
  double a = norm2(d);
  double half_b = dot(oc, d);
  double c = norm2(oc) - s.radius * s.radius;
  double disc = half_b * half_b - a * c;
  if (disc < 0.0) return false;
  double sqrtd = sqrt(disc);
  double root = (-half_b - sqrtd) / a;
  if (!(tmin < root && root < tmax)) {
    root = (-half_b + sqrtd) / a;
    if (!(tmin < root && root < tmax)) return false;
  }
  t = root;
  return true;
}

// Quad::hit, quad.rs:66-86 (plane test, interval, interior test).
RT_HD bool quad_hit_t(const rtx_quad& q, const V3& o, const V3& d, double tmin, double tmax, double& t, double& alpha,
                      double& beta) {
  V3 n = v3(q.normal);
  double denom = dot(n, d);
  if (fabs(denom) < 1e-8) return false;
  double tt = (q.d - dot(n, o)) / denom;
  if (!(tmin <= tt && tt <= tmax)) return false;
  V3 p = ray_at(o, d, tt);
  V3 planar = p - v3(q.q);
  V3 w = v3(q.w);
  double a = dot(w, cross(planar, v3(q.v)));
  double b = dot(w, cross(v3(q.u), planar));
  if (!(0.0 <= a && a <= 1.0) || !(0.0 <= b && b <= 1.0)) return false;
  t = tt; alpha = a; beta = b;
  return true;
}

// sphere.rs:57-66
RT_HD void sphere_uv(const V3& p, double& u, double& v) {
  double theta = acos(-p.y);
  double phi = atan2(-p.z, p.x) + RT_PI;
  u = phi / (2.0 * RT_PI);
  v = theta / RT_PI;
}

// (u,v) the reference's HitRecord would carry for a SPHERE/QUAD hit.
RT_HD void hit_uv(const Scene& S, const Ray& wray, const Hit& h, double& u, double& v) {
  uint32_t kind = RTX_NODE_KIND(h.kind_idx), idx = RTX_NODE_IDX(h.kind_idx);
  if (kind == RTX_NODE_QUAD) { u = h.u; v = h.v; return; }
  if (kind == RTX_NODE_SPHERE) {
    V3 o = wray.o, d = wray.d;
    if (h.ctx != 0) ctx_to_local(S, h.ctx, o, d);
    const rtx_sphere s = S.spheres[idx];
    V3 center = v3(s.c0) + v3(s.dc) * wray.tm;
    V3 outward = (ray_at(o, d, h.t) - center) / s.radius;
    sphere_uv(outward, u, v);
    return;
  }
  u = h.u; v = h.v;
}

// Closest hit of the program nodes[begin,end) for the world-space ray `wray`, starting in transform context ctx0.
// MEDIA = false skips ConstantMedium nodes (used for medium boundaries and for the deterministic first-hit AOV).
// `rng`/`bounce` address the medium free-path draws; the draw for a medium node uses that node's index as counter.
template <bool MEDIA>
RT_HD bool closest_hit(const Scene& S, uint32_t begin, uint32_t end, const Ray& wray, uint32_t ctx0, double tmin,
                       double tmax, Hit& best, const PathRng* rng, uint32_t bounce) {
  uint32_t cur = ctx0;
  LRay lr = lray_make(S, cur, wray.o, wray.d);
  bool found = false;
  uint32_t i = begin;
  while (i < end) {
    const rtx_node nd = S.nodes[i];
    const uint32_t kind = RTX_NODE_KIND(nd.kind_idx), idx = RTX_NODE_IDX(nd.kind_idx);
    if (kind == RTX_NODE_BOX) {
      i = slab_hit(S.boxes[idx], lr, tmin, tmax) ? i + 1 : nd.skip;
      continue;
    }
    if (kind == RTX_NODE_QUAD) {
      double t, a, b;
      if (quad_hit_t(S.quads[idx], lr.o, lr.d, tmin, tmax, t, a, b)) {
        found = true; tmax = t;
        best.t = t; best.kind_idx = nd.kind_idx; best.ctx = cur; best.u = a; best.v = b;
      }
    } else if (kind == RTX_NODE_SPHERE) {
      double t;
      if (sphere_hit_t(S.spheres[idx], lr.o, lr.d, wray.tm, tmin, tmax, t)) {
        found = true; tmax = t;
        best.t = t; best.kind_idx = nd.kind_idx; best.ctx = cur; best.u = 0.0; best.v = 0.0;
      }
    } else if (kind == RTX_NODE_ENTER || kind == RTX_NODE_EXIT) {
      cur = idx;
      lr = lray_make(S, cur, wray.o, wray.d);
    } else if (MEDIA && kind == RTX_NODE_MEDIUM) {
      if (rng != nullptr) {
        const rtx_medium m = S.media[idx];
        Hit h1, h2;
        // constant_medium.rs:47-57: entry over the whole line, exit after entry + 1e-4
        if (closest_hit<false>(S, m.b_begin, m.b_end, wray, cur, -RT_INF, RT_INF, h1, nullptr, 0) &&
            closest_hit<false>(S, m.b_begin, m.b_end, wray, cur, h1.t + 0.0001, RT_INF, h2, nullptr, 0)) {
          double t1 = fmax(h1.t, tmin);                       // 60
          double t2 = fmin(h2.t, tmax);                       // 61
          if (!(t1 >= t2)) {                                  // 63-65
            t1 = fmax(t1, 0.0);                               // 67
            double ray_length = norm(lr.d);                   // 70
            double inside = (t2 - t1) * ray_length;           // 71
            double xi = rng_at(*rng, bounce | ((uint32_t)RNG_DOMAIN_MEDIUM << 16), i);
            double hit_distance = m.neg_inv_density * log(xi);  // 74
            if (!(hit_distance > inside)) {                   // 76-78
              double t = t1 + hit_distance / ray_length;      // 81
              double su = 0.0, sv = 0.0;
              if (m.needs_uv && found) hit_uv(S, wray, best, su, sv);
              found = true; tmax = t;
              best.t = t; best.kind_idx = nd.kind_idx; best.ctx = cur; best.u = su; best.v = sv;
            }
          }
        }
      }
    }
    ++i;
  }
  return found;
}

// What the integrator needs from a hit: the fields of HitRecord (hittable.rs:9-17) in world space.
struct Surface {
  V3 p, n;
  double u, v;
  uint32_t mat, prim_id;
  bool front;
};

// Rebuild the HitRecord for `h`: same expressions, in the hit's own context, then carried out through the enclosing
// transforms (sphere.rs:106-115, quad.rs:88-92, constant_medium.rs:81-87, rotate_y.rs:123-124, translate.rs:45).
RT_HD void resolve_hit(const Scene& S, const Ray& wray, const Hit& h, bool want_uv, Surface& sf) {
  const uint32_t kind = RTX_NODE_KIND(h.kind_idx), idx = RTX_NODE_IDX(h.kind_idx);
  V3 o = wray.o, d = wray.d;
  if (h.ctx != 0) ctx_to_local(S, h.ctx, o, d);
  sf.p = ray_at(o, d, h.t);
  sf.u = h.u; sf.v = h.v;
  if (kind == RTX_NODE_SPHERE) {
    const rtx_sphere s = S.spheres[idx];
    V3 center = v3(s.c0) + v3(s.dc) * wray.tm;
    V3 outward = (sf.p - center) / s.radius;
    if (want_uv) sphere_uv(outward, sf.u, sf.v);
    sf.front = dot(d, outward) < 0.0;                 // hittable.rs:44-51
    sf.n = sf.front ? outward : -outward;
    sf.mat = s.mat; sf.prim_id = s.prim_id;
  } else if (kind == RTX_NODE_QUAD) {
    const rtx_quad q = S.quads[idx];
    V3 n = v3(q.normal);
    sf.front = dot(d, n) < 0.0;
    sf.n = sf.front ? n : -n;
    sf.mat = q.mat; sf.prim_id = q.prim_id;
  } else {  // MEDIUM
    sf.n = v3(1.0, 0.0, 0.0);
    sf.front = true;
    sf.mat = S.media[idx].mat; sf.prim_id = 0xFFFFFFFEu;
  }
  if (h.ctx != 0) ctx_to_world(S, h.ctx, sf.p, sf.n);
}

}  // namespace rt
