// Closest-hit query over a compiled hit program (see rtx.h: rtx_node, and rt_program.h).
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

// Compact primitive records the traversal reads instead of the ABI's full records (fewer, narrower loads).
struct DSphere {   // static sphere: centre and radius (sphere.rs:84 with a zero displacement)
  double cx, cy, cz, r;
};
struct DAAQuad {   // quad whose normal is exactly +-e_k (box faces, Cornell walls): only the terms that are not
  double d, nk;    // multiplied by an exact zero survive in quad.rs:67-82; k = axis, (a, b) = (k+1, k+2) mod 3
  double qa, qb, ua, ub, va, vb, wk;
  uint32_t axis, _pad;
};

// The six faces box_new builds (quad.rs:125-174) from two corners, as one record: the corners and the three
// distinct w_k values (front, right, top; the opposite faces carry their negatives).  box6_face() rebuilds each
// face's DAAQuad from it; the program compiler only emits a BOX6 unit after checking, field by field, that the six
// rebuilt records equal the ones derived from the ABI quads, so a BOX6 test IS six aaquad_hit_t tests.
struct DBox6 {
  double x0, y0, z0, x1, y1, z1, wx, wy, wz;
};
RT_HD DAAQuad box6_face(const DBox6& r, int f) {
  const double dx = r.x1 - r.x0, dy = r.y1 - r.y0, dz = r.z1 - r.z0;
  DAAQuad q;
  q._pad = 0;
  switch (f) {
    case 0: q.axis = 2; q.nk = 1.0;  q.d = r.z1;  q.qa = r.x0; q.qb = r.y0; q.ua = dx;  q.ub = 0.0; q.va = 0.0; q.vb = dy;  q.wk = r.wz;  break;  // front
    case 1: q.axis = 0; q.nk = 1.0;  q.d = r.x1;  q.qa = r.y0; q.qb = r.z1; q.ua = 0.0; q.ub = -dz; q.va = dy;  q.vb = 0.0; q.wk = r.wx;  break;  // right
    case 2: q.axis = 2; q.nk = -1.0; q.d = -r.z0; q.qa = r.x1; q.qb = r.y0; q.ua = -dx; q.ub = 0.0; q.va = 0.0; q.vb = dy;  q.wk = -r.wz; break;  // back
    case 3: q.axis = 0; q.nk = -1.0; q.d = -r.x0; q.qa = r.y0; q.qb = r.z0; q.ua = 0.0; q.ub = dz;  q.va = dy;  q.vb = 0.0; q.wk = -r.wx; break;  // left
    case 4: q.axis = 1; q.nk = 1.0;  q.d = r.y1;  q.qa = r.z1; q.qb = r.x0; q.ua = 0.0; q.ub = dx;  q.va = -dz; q.vb = 0.0; q.wk = r.wy;  break;  // top
    default: q.axis = 1; q.nk = -1.0; q.d = -r.y0; q.qa = r.z0; q.qb = r.x0; q.ua = 0.0; q.ub = dx;  q.va = dz;  q.vb = 0.0; q.wk = -r.wy; break;  // bottom
  }
  return q;
}

// The scene as the kernels see it: the tables of rtx_scene_desc (device pointers on the GPU) plus the compiled hit
// program (rt_program.h).  world_begin and media[].b_begin are UNIT indices into `prog` here.
struct Scene : rtx_scene_desc {
  const uint64_t* prog;
  uint32_t n_units;
  uint32_t tail_begin;  // first unit of the deferred section's program (0 = the world program has none); the units
                        // before it are all a walk with DEFER = true can reach
  uint32_t tail_near_depth;  // > 0: the deferred section is a binary tree of boxes over leaf records, of this depth;
                             // walk_stream_near may walk it nearer child first
  uint32_t head_near_depth;  // the same for the head section (its leaves may also hold quads and BOX6 records)
  uint32_t tail_near_quads;  // the deferred section's leaves hold quads too (0: static-sphere runs only)
  uint32_t _pad_near;
  uint32_t head_end;    // the SYNC unit that ends the head section (0 = none): the world program starts with one large
                        // box subtree, which may be walked as a stream (walk_stream) before the rest of the program
  uint32_t fresh_in_main;  // wavefront pipeline, with a head section: 1 = the head stream covers the survivors only and the
                           // world-program launch walks new camera paths from the start of the program (their waves are
                           // coherent: the per-ray walk serves them better than the stream); 2 = it also generates them
  // optional copies of the lights' primitive records, indexed like `lights` (null: read quads[] / spheres[]); the
  // wavefront shading kernel points them at LDS
  const rtx_quad* light_quads;
  const rtx_sphere* light_spheres;
  double near_margin;   // relative margin of the nearer-child-first walk's order-dependence watch (RT_NEAR_MARGIN by default)
  // Boxes over the light leaves, for HittablePDF::value (lights_pdf_value): a probe ray that misses a light's widened
  // box gets exactly 0 from that light (its Sphere/Quad::pdf_value begins with a hit test, sphere.rs:126, quad.rs:103),
  // so only the lights whose boxes it passes are evaluated.  Groups of 64 consecutive lights, each a small tree of
  // LightNode in depth-first order; n_light_nodes = 0: no table (few lights), every light is evaluated.
  const struct LightNode* light_nodes;
  uint32_t n_light_nodes, n_light_groups;
  const uint32_t* light_group_begin;   // n_light_groups + 1 node positions
  double light_far;     // the table serves probe origins whose coordinates are within this
  uint32_t main_flat;   // the world program from unit flat_begin (behind its first large box subtree, or its start) up to its
  uint32_t flat_begin;  // deferred section can be walked by walk_flat: no general MEDIUM unit, at most a few boxes
  // Top of the head section's box tree, for a walk of a program that stays in global memory: copies of the box records
  // of its first levels in heap order (root 0, children of h at 2h+1 and 2h+2), 7 units each, at unit head_top_begin of
  // the image, followed by the records' own unit positions (one uint32 each).  head_top_nodes = 2^levels - 1, 0 = none;
  // head_top_units = units of the whole table.  (n_units counts the program only: the tables lie behind it.)
  uint32_t head_top_begin, head_top_nodes, head_top_units;
  // The head section's box tree with two levels folded into one (CompiledProgram::build_head_wide), for the walk of a
  // section that stays in global memory: head_wide_nodes records of PROG_WIDE_UNITS units from unit head_wide_begin
  // (a multiple of 16: records are 128-byte aligned), the root's children first.  0 nodes = none.
  uint32_t head_wide_begin, head_wide_nodes, head_wide_depth, _pad_wide;   // depth: records on the longest way down
  double head_wide_far;   // the table serves rays whose origin coordinates are within this (see build_head_wide)
};
struct LightNode {     // a box of the light table: world space, widened (see CompiledProgram::build_light_table)
  double lo[3], hi[3];
  uint32_t skip;       // node to go on with when the ray misses this box (all nodes of a group lie between its bounds)
  uint32_t light;      // leaf: index of the light (rtx_scene_desc::lights); inner node: LIGHT_NODE_INNER
};
#define LIGHT_NODE_INNER 0xFFFFFFFFu
#define LIGHT_TABLE_MIN 8u                 // scenes with fewer lights evaluate them all: a box test costs what a light does
#define PROG_WIDE_PAD 2e-6                 // pad of the table's bounds, relative to the section's largest coordinate
#define PROG_WIDE_FAR 1e9                  // origins this many times farther out than that are walked over the program
#define PROG_WIDE_UNITS 16u                // four boxes (6 floats each, rounded outward), four slot descriptors
#define PROG_WIDE_EMPTY 0xFFFFFFFFFFFFFFFFull   // slot descriptor: no box in this slot
// record sizes in units (header included)
#define PROG_BOX_UNITS 7u        // header, lo[3], hi[3]
#define PROG_SPHERE_S_UNITS 4u   // per sphere of a run (DSphere)
#define PROG_QUAD_AA_UNITS 10u   // per quad of a run (DAAQuad)
#define PROG_BOX6_UNITS 10u      // header + DBox6
#define PROG_XFORM_OP_UNITS 4u   // per transform op of an ENTER/EXIT record: kind, a, b, c
#define PROG_MEDIUM_UNITS 3u     // header, neg_inv_density, flags
#define PROG_MEDIUM_S_UNITS 7u   // header, neg_inv_density, flags, DSphere of the boundary
#define PROG_MEDIUM_NEEDS_UV 1ull    // flags: the phase texture reads (u, v)
#define PROG_MEDIUM_DRAW_FIRST 2ull  // flags (MEDIUM_S): mean free path >= the boundary's diameter, draw it before intersecting
#define PROG_SPHERE_UNITS 8u     // a general (moving) sphere: header, c0[3], dc[3], radius
#define PROG_NODE_NOP 6u         // filler
#define PROG_NODE_END 7u         // end of a program (the world's, or a medium boundary's)
#define PROG_NODE_SPHERE_S 8u    // run of static spheres, compact records (hits are reported as RTX_NODE_SPHERE)
#define PROG_NODE_QUAD_AA 9u     // run of axis-aligned quads, compact records (hits are reported as RTX_NODE_QUAD)
#define PROG_NODE_SYNC 10u       // wave rendezvous between top-level sections of the world program
#define PROG_NODE_BOX6 11u       // the six axis-aligned faces of a box_new box (hits are reported as RTX_NODE_QUAD)
#define PROG_NODE_MEDIUM_S 12u   // a constant medium whose boundary is one static sphere, evaluated in one step
#define PROG_GUARD_IDX 0x0FFFFFFFu   // index field of the widened guard box in front of a BOX6 record (a box of the BVH
                                     // carries the index of its bounds in the ABI's table)
#define PROG_NODE_DEFER 13u      // behind the root box of the deferred section; aux = first unit of that section's program

// Units a record occupies, from its header unit.
RT_HD uint32_t prog_record_units(uint64_t unit) {
  const uint32_t kind = RTX_NODE_KIND((uint32_t)unit), aux = (uint32_t)(unit >> 32);
  switch (kind) {
    case RTX_NODE_BOX: return PROG_BOX_UNITS;
    case PROG_NODE_SPHERE_S: return 1u + PROG_SPHERE_S_UNITS * aux;
    case PROG_NODE_QUAD_AA: return 1u + PROG_QUAD_AA_UNITS * aux;
    case PROG_NODE_BOX6: return PROG_BOX6_UNITS;
    case RTX_NODE_ENTER: case RTX_NODE_EXIT: return 1u + PROG_XFORM_OP_UNITS * aux;
    case RTX_NODE_MEDIUM: return PROG_MEDIUM_UNITS;
    case PROG_NODE_MEDIUM_S: return PROG_MEDIUM_S_UNITS;
    case RTX_NODE_SPHERE: return PROG_SPHERE_UNITS;
    default: return 1u;
  }
}

// Test-only hook: a build may define RT_TRACE_STEP(kind) to record the unit kinds a ray steps through.
#ifndef RT_TRACE_STEP
#define RT_TRACE_STEP(kind) ((void)0)
#endif

// Wave-level votes.  On the GPU they steer the 64 lanes of a wave through the traversal loop together; in the host
// build a "wave" is one ray and they reduce to the lane's own predicate.
// Optional cycle accounting of the traversal loop (a tuning build defines RT_PROF_TRIP; the product does not).
#ifndef RT_WPROF
#define RT_WPROF_DECL() do { } while (0)
#define RT_WPROF(slot, value) do { } while (0)
#define RT_WPROF_CLOCK() 0ll
#define RT_WPROF_FLUSH() do { } while (0)
#endif
#ifndef RT_PROF_TRIP
#define RT_PROF_START() do { } while (0)
#define RT_PROF_LANES(n) do { } while (0)
#define RT_PROF_TRIP(slot) do { } while (0)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_WAVE_ANY(p) (__ballot(p) != 0ull)
#define RT_WAVE_COUNT(p) ((int)__popcll(__ballot(p)))
#define RT_WAVE_LANES 64
// Do all lanes with `on` hold the same value of v (at least one lane has `on`)?  first = that value.
__device__ __forceinline__ bool rt_wave_uniform(bool on, int v, int& first, int& n_on) {
  const unsigned long long m = __ballot(on);
  first = __shfl(v, __ffsll((long long)m) - 1);
  n_on = (int)__popcll(m);
  return __ballot(on && v == first) == m;
}
#else
#define RT_WAVE_ANY(p) (p)
#define RT_WAVE_COUNT(p) ((p) ? 1 : 0)
#define RT_WAVE_LANES 1
inline bool rt_wave_uniform(bool, int v, int& first, int& n_on) { first = v; n_on = 1; return true; }
#endif

struct LRay {     // ray in the current transform context + per-axis reciprocals (aabb.rs:146)
  V3 o, d, inv;
  uint32_t skip;  // bit k set: |d_k| < 1e-8, axis k is skipped by the slab test (aabb.rs:142-144)
                  // bit 3 set: a component of o or d is not finite (compact quad records are not used then)
};

struct Hit {
  double t;
  uint32_t kind_idx;  // RTX_NODE_MAKE(SPHERE|QUAD|MEDIUM, table index); kind NONE = miss
  uint32_t ctx;       // transform context the hit was found in
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

// max/min of two doubles where a NaN operand loses (what `if x > m { m = x }` does, aabb.rs:154-164).  On the device
// this is the bare v_max_f64 / v_min_f64: fmax()/fmin() would first re-quiet both operands (two more instructions
// per call), which matters only for signalling NaNs, and no arithmetic result is one.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double rt_max(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double rt_min(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#else
inline double rt_max(double a, double b) { return fmax(a, b); }
inline double rt_min(double a, double b) { return fmin(a, b); }
#endif

// The local ray of a transform context from the op chain an ENTER/EXIT record carries (same operations as
// ctx_to_local, operands read from the program image instead of the tables).
RT_HD LRay lray_from_ops(const uint64_t* ops, uint32_t n_ops, const V3& wo, const V3& wd) {
  LRay l;
  l.o = wo; l.d = wd;
  for (uint32_t k = 0; k < n_ops; ++k) {
    const uint64_t* r = ops + PROG_XFORM_OP_UNITS * k;
    const double* f = (const double*)r;
    if ((uint32_t)r[0] == RTX_XF_TRANSLATE) {
      l.o = l.o - v3(f[1], f[2], f[3]);
    } else {
      l.o = rot_w2l(l.o, f[1], f[2]);
      l.d = rot_w2l(l.d, f[1], f[2]);
    }
  }
  l.inv = v3(1.0 / l.d.x, 1.0 / l.d.y, 1.0 / l.d.z);
  l.skip = (fabs(l.d.x) < 1e-8 ? 1u : 0u) | (fabs(l.d.y) < 1e-8 ? 2u : 0u) | (fabs(l.d.z) < 1e-8 ? 4u : 0u);
  const double mag = ((fabs(l.o.x) + fabs(l.o.y)) + fabs(l.o.z)) + ((fabs(l.d.x) + fabs(l.d.y)) + fabs(l.d.z));
  if (!(mag < RT_INF)) l.skip |= 8u;
  return l;
}

// Aabb::hit, aabb.rs:132-167.  The per-axis early return is folded into one final comparison: the interval only
// narrows, so it is empty at the end exactly when it was empty after some axis.
RT_HD bool slab_hit(const double* b, const LRay& r, double tmin, double tmax) {  // b: lo[3], hi[3]
  // The reference computes both plane distances of an axis and then orders them by the sign of 1/d
  // (aabb.rs:154-157).  Here the sign picks WHICH bound is read as the near one, so no value is selected after the
  // arithmetic: near = (b[neg ? hi : lo] - o) * inv is the same operation on the same operands.
  const uint32_t nx = r.inv.x >= 0.0 ? 0u : 3u, ny = r.inv.y >= 0.0 ? 0u : 3u, nz = r.inv.z >= 0.0 ? 0u : 3u;
  const double bnx = b[nx], bfx = b[3u - nx], bny = b[1u + ny], bfy = b[4u - ny], bnz = b[2u + nz], bfz = b[5u - nz];
  double nrx = (bnx - r.o.x) * r.inv.x, frx = (bfx - r.o.x) * r.inv.x;
  double nry = (bny - r.o.y) * r.inv.y, fry = (bfy - r.o.y) * r.inv.y;
  double nrz = (bnz - r.o.z) * r.inv.z, frz = (bfz - r.o.z) * r.inv.z;
  if (r.skip & 7u) {  // a skipped axis (aabb.rs:142-144) leaves the interval untouched, like (-inf, +inf) does
    if (r.skip & 1u) { nrx = -RT_INF; frx = RT_INF; }
    if (r.skip & 2u) { nry = -RT_INF; fry = RT_INF; }
    if (r.skip & 4u) { nrz = -RT_INF; frz = RT_INF; }
  }
  tmin = rt_max(tmin, nrx); tmax = rt_min(tmax, frx);
  tmin = rt_max(tmin, nry); tmax = rt_min(tmax, fry);
  tmin = rt_max(tmin, nrz); tmax = rt_min(tmax, frz);
  return !(tmax <= tmin);
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

// Sphere::hit (sphere.rs:83-104) for a sphere given by the record a SPHERE unit carries: c0[3], dc[3], radius.  The
// same operations on the same operands as sphere_hit_t over the table entry.
RT_HD bool sphere_rec_hit_t(const double* rec, const V3& o, const V3& d, double time, double tmin, double tmax, double& t) {
  V3 center = v3(rec[0], rec[1], rec[2]) + v3(rec[3], rec[4], rec[5]) * time;
  V3 oc = o - center;
  double a = norm2(d);
  double half_b = dot(oc, d);
  double c = norm2(oc) - rec[6] * rec[6];
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
RT_HD bool quad_hit_v(const V3& n, double qd, const V3& qq, const V3& w, const V3& qu, const V3& qv, const V3& o, const V3& d,
                      double tmin, double tmax, double& t) {   // the quad's fields as values (a record held in registers)
  double denom = dot(n, d);
  if (fabs(denom) < 1e-8) return false;
  double tt = (qd - dot(n, o)) / denom;
  if (!(tmin <= tt && tt <= tmax)) return false;
  V3 p = ray_at(o, d, tt);
  V3 planar = p - qq;
  double a = dot(w, cross(planar, qv));
  double b = dot(w, cross(qu, planar));
  if (!(0.0 <= a && a <= 1.0) || !(0.0 <= b && b <= 1.0)) return false;
  t = tt;
  return true;
}
RT_HD bool quad_hit_t(const rtx_quad& q, const V3& o, const V3& d, double tmin, double tmax, double& t) {
  return quad_hit_v(v3(q.normal), q.d, v3(q.q), v3(q.w), v3(q.u), v3(q.v), o, d, tmin, tmax, t);
}
// Sphere::hit for a static sphere from its compact record: centre = c0 + 0*time = c0 (sphere.rs:84-104).
RT_HD bool sphere_s_hit_t(const DSphere& s, const V3& o, const V3& d, double tmin, double tmax, double& t) {
  V3 oc = o - v3(s.cx, s.cy, s.cz);
  double a = norm2(d);
  double half_b = dot(oc, d);
  double c = norm2(oc) - s.r * s.r;
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

// Quad::hit (quad.rs:66-86) for a quad whose unit normal is exactly +-e_k, for a ray with finite components.  Every
// product with an exact zero of normal / w drops out of the reference's dot products without changing their value,
// so t is bit-identical and the interior test decides identically (alpha, beta can differ in the sign of a zero).
RT_HD bool aaquad_hit_t(const DAAQuad& q, const V3& o, const V3& d, double tmin, double tmax, double& t) {
  // components picked by value selects (the six values are read first: a select between their addresses would keep
  // the ray in addressable scratch memory on the GPU)
  const double ox = o.x, oy = o.y, oz = o.z, dx = d.x, dy = d.y, dz = d.z;
  const bool k0 = q.axis == 0, k1 = q.axis == 1;
  const double ok = k0 ? ox : (k1 ? oy : oz), dk = k0 ? dx : (k1 ? dy : dz);
  const double oa = k0 ? oy : (k1 ? oz : ox), da = k0 ? dy : (k1 ? dz : dx);
  const double ob = k0 ? oz : (k1 ? ox : oy), db = k0 ? dz : (k1 ? dx : dy);
  double denom = q.nk * dk;
  if (fabs(denom) < 1e-8) return false;
  double tt = (q.d - q.nk * ok) / denom;
  if (!(tmin <= tt && tt <= tmax)) return false;
  double pa = (oa + da * tt) - q.qa, pb = (ob + db * tt) - q.qb;
  double alpha = q.wk * (pa * q.vb - pb * q.va);
  double beta = q.wk * (q.ua * pb - q.ub * pa);
  if (!(0.0 <= alpha && alpha <= 1.0) || !(0.0 <= beta && beta <= 1.0)) return false;
  t = tt;
  return true;
}

// The (alpha, beta) Quad::hit stores as (u, v) (quad.rs:81-82, 59-60) for the accepted point p.
RT_HD void quad_uv(const rtx_quad& q, const V3& p, double& u, double& v) {
  V3 planar = p - v3(q.q);
  V3 w = v3(q.w);
  u = dot(w, cross(planar, v3(q.v)));
  v = dot(w, cross(v3(q.u), planar));
}

// sphere.rs:57-66
RT_HD void sphere_uv(const V3& p, double& u, double& v) {
  double theta = acos(-p.y);
  double phi = atan2(-p.z, p.x) + RT_PI;
  u = phi / (2.0 * RT_PI);
  v = theta / RT_PI;
}

// (u,v) the reference's HitRecord carries for a SPHERE/QUAD hit at parameter t.
RT_HD void hit_uv(const Scene& S, const Ray& wray, uint32_t kind_idx, uint32_t ctx, double t, double& u, double& v) {
  const uint32_t kind = RTX_NODE_KIND(kind_idx), idx = RTX_NODE_IDX(kind_idx);
  V3 o = wray.o, d = wray.d;
  if (ctx != 0) ctx_to_local(S, ctx, o, d);
  V3 p = ray_at(o, d, t);
  if (kind == RTX_NODE_QUAD) {
    quad_uv(S.quads[idx], p, u, v);
  } else if (kind == RTX_NODE_SPHERE) {
    const rtx_sphere s = S.spheres[idx];
    V3 center = v3(s.c0) + v3(s.dc) * wray.tm;
    sphere_uv((p - center) / s.radius, u, v);
  }
}

// Where the traversal gets the world-space ray from when it has to rebuild the local ray (at program start, at
// ENTER/EXIT nodes, for the stale (u,v)).  Keeping the world ray in registers for the whole walk costs 12 VGPRs;
// the wavefront kernels re-read it from the path queue instead.
struct RayRegs {
  Ray r;
  RT_HD V3 origin() const { return r.o; }
  RT_HD V3 dir() const { return r.d; }
};
struct RayMem {       // structure-of-arrays queue: f[k*cap], k = 0..5
  const double* f;
  size_t cap;
  RT_HD V3 origin() const { return v3(f[0], f[cap], f[2 * cap]); }
  RT_HD V3 dir() const { return v3(f[3 * cap], f[4 * cap], f[5 * cap]); }
};

// Closest hit of the compiled program starting at unit `begin` (it runs until its END unit) for the world-space ray
// `rs` (time `tm`) over [tmin0, tmax0], starting in the world context.  Returns kind NONE in best.kind_idx on a miss.
// `active` false = this lane has no ray (it only takes part in the wave votes).
//
// Scheduling.  Each lane walks its own ray through the program, but the wave executes one KIND of step at a time:
//   1. a tight loop of box tests runs while any lane stands on a box;
//   2. lanes on other cheap units (filler, transform, program end, medium entry) take one step;
//   3. once every lane stands on a primitive run, the wave votes and executes the primitive test most lanes wait
//      for (static-sphere run, axis-aligned-quad run, general sphere/quad, medium free-path draw); the other lanes
//      keep their place;
//   4. a SYNC unit (placed by the program compiler between the top-level sections of the world) holds a lane until
//      every lane of the wave has reached it, so that the section that follows runs in lockstep.
// Lanes therefore share each expensive instruction stream instead of serialising sphere, quad and medium code in
// every trip.  The order of steps within one ray is untouched, so results do not depend on the schedule.
//
// MEDIA = MEDIA_NONE skips ConstantMedium nodes (the deterministic first-hit AOV).  MEDIA_SPHERES is for programs whose
// media are all of the one-step sphere-bounded kind (MEDIUM_S units; the scene says whether that holds): the code
// and registers of the general case drop out.  With MEDIA_ALL, a medium node runs its
// boundary program twice through this same loop (constant_medium.rs:47-57: entry over the whole line, exit after
// entry + 1e-4) by saving the outer query in a few registers instead of calling itself; boundaries never contain
// media (checked when the scene is created).  The free-path draw of a medium uses the medium's table index as RNG
// counter, so it does not depend on how paths are scheduled.  When a medium whose phase texture reads (u,v) scatters,
// the reference's record still holds the (u,v) of the closest surface hit found before it (constant_medium.rs:81-87
// leaves u,v untouched): stale_uv (2 doubles, may be null) then receives that hit (its t, and kind_idx | ctx << 32 as a
// bit pattern), from which resolve_hit computes the two values.
//
// Deferred section.  The program compiler may move the last top-level box subtree of the world (with its transform)
// into a program of its own and leave a DEFER unit behind the subtree's root box (rt_program.h).  With DEFER = false
// the unit is a call: the walk continues in that program and comes back at its END, i.e. the walk is the reference's.
// With DEFER = true the walk notes that the ray passed the root box, skips the section and returns true; the caller
// runs the section later (same ray, tmax0 = the t found so far), which is the same walk cut in two because the
// section is the last thing the world program visits.
enum { MEDIA_NONE = 0, MEDIA_SPHERES = 1, MEDIA_ALL = 2 };
// PLAIN: the caller vouches that the walk meets no ENTER or EXIT unit (CompiledProgram::main_flat says so for the head
// section): the local ray then never changes, and the compiler need not carry it through the loop as a variable.
template <int MEDIA, bool DEFER, bool PLAIN = false, class RaySrc>
RT_HD bool walk_program(const Scene& S, const uint64_t* prog, uint32_t begin, bool active, const RaySrc& rs, double tm,
                        double tmin0, double tmax0, Hit& best, bool use_media, const PathRng& rng, uint32_t bounce,
                        double* stale_uv, uint32_t kind0 = RTX_NODE_MAKE(RTX_NODE_NONE, 0), uint32_t ctx0 = 0,
                        bool have_start = false, V3 start_o = V3(), V3 start_d = V3(), uint32_t stop_at = 0) {
  // stop_at: a unit (in the world context, outside every box subtree) at which the walk ends as at an END; 0 = none
  enum { CL_CHEAP = 0, CL_SPHERE_S = 1, CL_QUAD_AA = 2, CL_SPHERE = 3, CL_QUAD = 4, CL_MEDIUM_EVAL = 5, CL_SYNC = 6, CL_BOX6 = 7, CL_BOX = 8, CL_MEDIUM_S = 9 };
  uint32_t cur = 0;
  uint32_t ret_tail = 0;  // DEFER = false: where to continue after the deferred program's END (0 = not inside it)
  bool deferred = false;
  LRay lr;
  lr.o = v3(0, 0, 0); lr.d = v3(0, 0, 1); lr.inv = v3(0, 0, 1); lr.skip = 0;
  // have_start: the caller already holds the world ray (start_o, start_d: the values `rs` will return later)
  if (active) lr = have_start ? lray_from_ops(prog, 0, start_o, start_d) : lray_from_ops(prog, 0, rs.origin(), rs.dir());
  uint32_t i = begin;
  double tmin = tmin0, tmax = tmax0;
  uint32_t hit_kind = kind0, hit_ctx = ctx0;  // the hit an earlier part of the same walk found at tmax0, if any
  // state of the enclosing query while a medium boundary program runs
  int mode = 0;  // 0: world program, 1: boundary entry query, 2: boundary exit query
  uint32_t ret_i = 0, ret_cur = 0, out_kind = 0, out_ctx = 0, med = 0, med_at = 0;
  double out_tmax = 0.0, t_entry = 0.0;
  bool running = active;
  uint64_t unit = running ? prog[i] : (uint64_t)RTX_NODE_MAKE(PROG_NODE_END, 0);
  RT_PROF_START();
  while (RT_WAVE_ANY(running)) {
    if (stop_at != 0) {   // (a lane gets here through a skip link, a box step or the unit before)
      if (running && i == stop_at) running = false;
      if (!RT_WAVE_ANY(running)) break;
    }
    const uint32_t kind_idx = (uint32_t)unit;
    const uint32_t kind = RTX_NODE_KIND(kind_idx), idx = RTX_NODE_IDX(kind_idx), count = (uint32_t)(unit >> 32);
    int cls = CL_CHEAP;
    if (kind == RTX_NODE_BOX) cls = CL_BOX;
    else if (kind == PROG_NODE_SPHERE_S) cls = CL_SPHERE_S;
    else if (kind == PROG_NODE_QUAD_AA) cls = (lr.skip & 8u) ? CL_QUAD : CL_QUAD_AA;
    else if (kind == PROG_NODE_BOX6) cls = (lr.skip & 8u) ? CL_QUAD : CL_BOX6;
    else if (kind == RTX_NODE_SPHERE) cls = CL_SPHERE;
    else if (kind == RTX_NODE_QUAD) cls = CL_QUAD;
    else if (kind == PROG_NODE_SYNC) cls = CL_SYNC;
    else if (MEDIA == MEDIA_ALL && kind == PROG_NODE_END && mode == 2 && RTX_NODE_KIND(hit_kind) != RTX_NODE_NONE) cls = CL_MEDIUM_EVAL;
    else if (MEDIA && kind == PROG_NODE_MEDIUM_S && use_media) cls = CL_MEDIUM_S;
    uint32_t next = i;  // unit to stand on after this trip
    if (RT_WAVE_ANY(running && cls == CL_CHEAP)) {
      RT_PROF_LANES(RT_WAVE_COUNT(running && cls == CL_CHEAP));
      // 2. other cheap units
      if (running && cls == CL_CHEAP) {
        if (kind == PROG_NODE_END) {
          if (MEDIA != MEDIA_ALL || mode == 0) {
            if (!DEFER && ret_tail != 0) { next = ret_tail; ret_tail = 0; }  // end of the called section: go back
            else running = false;
          } else if (mode == 1 && RTX_NODE_KIND(hit_kind) != RTX_NODE_NONE) {  // entry found at tmax: look for the exit
            t_entry = tmax;
            mode = 2; next = (uint32_t)(prog[med_at] >> 32); tmin = t_entry + 0.0001; tmax = RT_INF;
            hit_kind = RTX_NODE_MAKE(RTX_NODE_NONE, 0);
          } else {                                                             // boundary missed: resume the world
            mode = 0; next = ret_i; tmin = tmin0; tmax = out_tmax; hit_kind = out_kind; hit_ctx = out_ctx;
          }
        } else if (!PLAIN && (kind == RTX_NODE_ENTER || kind == RTX_NODE_EXIT)) {
          // in the world context the local ray IS the world ray: only a lane that stands in another context re-reads it
          const V3 wo = cur == 0 ? lr.o : rs.origin(), wd = cur == 0 ? lr.d : rs.dir();
          cur = idx;
          lr = lray_from_ops(prog + i + 1, count, wo, wd);
          next = i + 1 + PROG_XFORM_OP_UNITS * count;
        } else if (MEDIA == MEDIA_ALL && kind == RTX_NODE_MEDIUM && use_media) {
          // save the enclosing query, run the boundary program from this context over the whole line
          med = idx; med_at = i; ret_i = i + PROG_MEDIUM_UNITS; ret_cur = cur; out_tmax = tmax; out_kind = hit_kind; out_ctx = hit_ctx;
          mode = 1; next = count; tmin = -RT_INF; tmax = RT_INF;
          hit_kind = RTX_NODE_MAKE(RTX_NODE_NONE, 0);
        } else if (kind == PROG_NODE_DEFER) {
          if (DEFER) { deferred = true; next = i + 1; }
          else { ret_tail = i + 1; next = count; }
        } else {
          next = i + prog_record_units(unit);  // a medium the query ignores, or filler
        }
      }
      RT_PROF_TRIP(CL_CHEAP);
    } else {
      // 3. every running lane stands on a primitive (or waits at a SYNC): run the class most lanes wait for
      int pick = CL_SYNC, best_n = 0, cls0 = 0;
      if (rt_wave_uniform(running, cls, cls0, best_n)) {
        // all of them stand on the same class (the common case in the top-level part of a program, where the rays of
        // a wave move in step): no count per class is needed; the vote below would pick the same
        pick = cls0 == CL_MEDIUM_S ? CL_MEDIUM_EVAL : cls0;
        if (pick == CL_BOX || pick == CL_SYNC) best_n = 0;
      } else {
        const int n_ss = RT_WAVE_COUNT(running && cls == CL_SPHERE_S), n_a = RT_WAVE_COUNT(running && cls == CL_QUAD_AA);
        const int n_s = RT_WAVE_COUNT(running && cls == CL_SPHERE), n_q = RT_WAVE_COUNT(running && cls == CL_QUAD);
        const int n_m = RT_WAVE_COUNT(running && (cls == CL_MEDIUM_EVAL || cls == CL_MEDIUM_S)), n_b = RT_WAVE_COUNT(running && cls == CL_BOX6);
        const int n_x = RT_WAVE_COUNT(running && cls == CL_BOX);
        best_n = 0;
        if (n_ss > best_n) { pick = CL_SPHERE_S; best_n = n_ss; }
        if (n_b > best_n) { pick = CL_BOX6; best_n = n_b; }
        if (n_a > best_n) { pick = CL_QUAD_AA; best_n = n_a; }
        if (n_s > best_n) { pick = CL_SPHERE; best_n = n_s; }
        if (n_q > best_n) { pick = CL_QUAD; best_n = n_q; }
        if (n_m > best_n) { pick = CL_MEDIUM_EVAL; best_n = n_m; }
        // Boxes wait until no lane stands on a primitive, then every lane walks its boxes together: two alternating
        // phases, so that a box trip serves as many lanes as possible.
        if (best_n == 0 && n_x > 0) pick = CL_BOX;
      }
      RT_PROF_LANES(pick == CL_BOX ? 0 : best_n);
      if (pick == CL_SYNC) {
        // 4. everybody waits at a SYNC: pass it together
        if (running && cls == CL_SYNC) next = i + 1;
      } else if (pick == CL_BOX) {
        // walk boxes until no lane stands on one (a tight loop: a vote per step would cost more than the box test)
        do {
          RT_PROF_LANES(RT_WAVE_COUNT(running && RTX_NODE_KIND((uint32_t)unit) == RTX_NODE_BOX));
          if (running && RTX_NODE_KIND((uint32_t)unit) == RTX_NODE_BOX) {
            // both possible successors are read together with the bounds: one LDS round trip per box
            const uint32_t i_pass = i + PROG_BOX_UNITS, i_fail = (uint32_t)(unit >> 32);
            RT_TRACE_STEP(RTX_NODE_BOX);
            const uint64_t u_pass = prog[i_pass], u_fail = prog[i_fail];
            const bool pass = slab_hit((const double*)(prog + i + 1), lr, tmin, tmax);
            i = pass ? i_pass : i_fail;
            unit = pass ? u_pass : u_fail;
          }
          RT_PROF_TRIP(9);
        } while (RT_WAVE_ANY(running && RTX_NODE_KIND((uint32_t)unit) == RTX_NODE_BOX));
        next = i;
      } else if (pick == CL_SPHERE_S) {
        if (running && cls == CL_SPHERE_S) {
          for (uint32_t k = 0; k < count; ++k) {
            double t;
            const DSphere sp = *(const DSphere*)(prog + i + 1 + PROG_SPHERE_S_UNITS * k);
            if (sphere_s_hit_t(sp, lr.o, lr.d, tmin, tmax, t)) { tmax = t; hit_kind = RTX_NODE_MAKE(RTX_NODE_SPHERE, idx + k); hit_ctx = cur; }
          }
          next = i + 1 + PROG_SPHERE_S_UNITS * count;
        }
      } else if (pick == CL_BOX6) {
        if (running && cls == CL_BOX6) {
          const DBox6 bx = *(const DBox6*)(prog + i + 1);
RT_UNROLL
          for (int f = 0; f < 6; ++f) {
            double t;
            if (aaquad_hit_t(box6_face(bx, f), lr.o, lr.d, tmin, tmax, t)) { tmax = t; hit_kind = RTX_NODE_MAKE(RTX_NODE_QUAD, idx + (uint32_t)f); hit_ctx = cur; }
          }
          next = i + PROG_BOX6_UNITS;
        }
      } else if (pick == CL_QUAD_AA) {
        if (running && cls == CL_QUAD_AA) {
          for (uint32_t k = 0; k < count; ++k) {
            double t;
            const DAAQuad qa = *(const DAAQuad*)(prog + i + 1 + PROG_QUAD_AA_UNITS * k);
            if (aaquad_hit_t(qa, lr.o, lr.d, tmin, tmax, t)) { tmax = t; hit_kind = RTX_NODE_MAKE(RTX_NODE_QUAD, idx + k); hit_ctx = cur; }
          }
          next = i + 1 + PROG_QUAD_AA_UNITS * count;
        }
      } else if (pick == CL_SPHERE) {
        if (running && cls == CL_SPHERE) {
          double t;
          if (sphere_rec_hit_t((const double*)(prog + i + 1), lr.o, lr.d, tm, tmin, tmax, t)) { tmax = t; hit_kind = RTX_NODE_MAKE(RTX_NODE_SPHERE, idx); hit_ctx = cur; }
          next = i + PROG_SPHERE_UNITS;
        }
      } else if (pick == CL_QUAD) {
        if (running && cls == CL_QUAD) {  // a general quad, or a compact run met by a ray with non-finite components
          uint32_t cnt = 1u, size = 1u;
          if (kind == PROG_NODE_QUAD_AA) { cnt = count; size = 1u + PROG_QUAD_AA_UNITS * count; }
          else if (kind == PROG_NODE_BOX6) { cnt = 6u; size = PROG_BOX6_UNITS; }
          for (uint32_t k = 0; k < cnt; ++k) {
            double t;
            if (quad_hit_t(S.quads[idx + k], lr.o, lr.d, tmin, tmax, t)) { tmax = t; hit_kind = RTX_NODE_MAKE(RTX_NODE_QUAD, idx + k); hit_ctx = cur; }
          }
          next = i + size;
        }
      } else if (MEDIA) {
        // constant_medium.rs:60-87 once both boundary crossings are known.  MEDIUM_EVAL: the boundary program found
        // the exit at tmax (entry in t_entry), the enclosing query waits in out_*.  MEDIUM_S: the boundary is one static
        // sphere, so both crossings (entry over the whole line, exit after entry + 1e-4, constant_medium.rs:47-57) are
        // computed here with the same sphere routine and the enclosing query is simply the current one.
        const bool eval = MEDIA == MEDIA_ALL && running && cls == CL_MEDIUM_EVAL, direct = running && cls == CL_MEDIUM_S;
        if (eval || direct) {
          const uint32_t at = direct ? i : med_at;
          const double neg_inv_density = ((const double*)prog)[at + 1];
          const uint64_t mflags = prog[at + 2];   // PROG_MEDIUM_NEEDS_UV | PROG_MEDIUM_DRAW_FIRST
          double t_in = t_entry, t_out = tmax;
          bool crossed = true, drawn = false;
          double ray_length = 0.0, hit_distance = 0.0;
          if (direct) {
            med = idx; ret_i = i + PROG_MEDIUM_S_UNITS; ret_cur = cur; out_tmax = tmax; out_kind = hit_kind; out_ctx = hit_ctx;
            const DSphere sp = *(const DSphere*)(prog + i + 3);
            if ((mflags & PROG_MEDIUM_DRAW_FIRST) && tmin0 >= 0.0) {
              // A thin medium (mean free path of at least the boundary's diameter): the free path is drawn FIRST, and a
              // path longer than any stretch of this ray inside the boundary can be ends the evaluation before the
              // boundary is intersected.  constant_medium.rs:60-78 then returns false whatever the crossings are: its
              // distance_inside_boundary = (t2 - t1) * ray_length has 0 < tmin0 <= t1 and t2 <= tmax, so it is at most
              // tmax * ray_length (rounding is monotone), and at most the sphere's diameter (the two roots differ by
              // 2 sqrt(disc) / a, a chord; the margin of 1e-9 covers the rounding of the roots for origins within 100
              // radii).  The draw is addressed by (bounce, medium), so making it early changes no other draw.
              ray_length = norm(lr.d);
              hit_distance = neg_inv_density * log(rng_at(rng, bounce | ((uint32_t)RNG_DOMAIN_MEDIUM << 16), med));
              drawn = true;
              bool beyond = hit_distance > tmax * ray_length;
              if (!beyond) {
                const V3 oc = lr.o - v3(sp.cx, sp.cy, sp.cz);
                beyond = norm2(oc) <= 1e4 * (sp.r * sp.r) && hit_distance > (2.0 * sp.r) * (1.0 + 1e-9);
              }
              crossed = !beyond;
            }
            crossed = crossed && sphere_s_hit_t(sp, lr.o, lr.d, -RT_INF, RT_INF, t_in) &&
                      sphere_s_hit_t(sp, lr.o, lr.d, t_in + 0.0001, RT_INF, t_out);
          }
          if (crossed) {
            const bool needs_uv = (mflags & PROG_MEDIUM_NEEDS_UV) != 0;
            double t1 = fmax(t_in, tmin0);
            double t2 = fmin(t_out, out_tmax);
            if (!(t1 >= t2)) {
              t1 = fmax(t1, 0.0);
              if (!drawn) {
                ray_length = norm(lr.d);
                hit_distance = neg_inv_density * log(rng_at(rng, bounce | ((uint32_t)RNG_DOMAIN_MEDIUM << 16), med));
              }
              double inside = (t2 - t1) * ray_length;
              if (!(hit_distance > inside)) {
                if (needs_uv && stale_uv) {  // the surface hit this scattering replaces: resolve_hit turns it into (u,v)
                  const uint64_t id = (uint64_t)out_kind | ((uint64_t)out_ctx << 32);
                  stale_uv[0] = out_tmax;
                  memcpy(&stale_uv[1], &id, 8);
                }
                out_tmax = t1 + hit_distance / ray_length;
                out_kind = RTX_NODE_MAKE(RTX_NODE_MEDIUM, med); out_ctx = ret_cur;
              }
            }
          }
          mode = 0; next = ret_i; tmin = tmin0; tmax = out_tmax; hit_kind = out_kind; hit_ctx = out_ctx;
        }
      }
      RT_PROF_TRIP(pick);
    }
    if (running && next != i) {
      RT_TRACE_STEP(cls == CL_CHEAP ? kind : 32u + (uint32_t)cls);
      i = next; unit = prog[i];
      // the END of the outermost program ends the walk at once (no trip of its own)
      if (RTX_NODE_KIND((uint32_t)unit) == PROG_NODE_END && (MEDIA != MEDIA_ALL || mode == 0) && (DEFER || ret_tail == 0)) running = false;
    }
  }
  best.t = tmax; best.kind_idx = hit_kind; best.ctx = hit_ctx;
  return deferred;
}

// walk_program<MEDIA, true> for a stretch of the program that every ray of a wave passes unit by unit together - a
// top-level list of primitives, sphere-bounded media and transforms with at most a few boxes in it
// (CompiledProgram::main_flat) - from unit `begin` to the END.  ONE position for the whole wave: the unit is read once
// and decoded in scalar registers, every lane that is not inside a box it failed takes the step, and nothing is voted
// on.  A lane that fails a box sits out until the sweep reaches the box's skip target (subtrees nest, so a lane's target
// is never passed over).  Per ray the same units in the same order with the same running tmax as walk_program: the
// same result bit for bit.  No general MEDIUM units (the caller checks); MEDIA: MEDIA_NONE or MEDIA_SPHERES.
template <int MEDIA, class RaySrc>
RT_HD bool walk_flat(const Scene& S, const uint64_t* prog, uint32_t begin, bool active, const RaySrc& rs, double tm,
                     double tmin0, double tmax0, Hit& best, bool use_media, const PathRng& rng, uint32_t bounce,
                     double* stale_uv, uint32_t kind0, uint32_t ctx0, bool have_start, V3 start_o, V3 start_d) {
  uint32_t cur = 0, skip_until = 0;     // skip_until: unit at which this lane joins in again (0 = it is in)
  bool deferred = false;
  LRay lr;
  lr.o = v3(0, 0, 0); lr.d = v3(0, 0, 1); lr.inv = v3(0, 0, 1); lr.skip = 0;
  if (active) lr = have_start ? lray_from_ops(prog, 0, start_o, start_d) : lray_from_ops(prog, 0, rs.origin(), rs.dir());
  const double tmin = tmin0;
  double tmax = tmax0;
  uint32_t hit_kind = kind0, hit_ctx = ctx0;
  uint32_t i = begin;
  if (!RT_WAVE_ANY(active)) { best.t = tmax; best.kind_idx = hit_kind; best.ctx = hit_ctx; return false; }
  // the header of a unit in scalar registers (every lane reads the same word)
  auto header = [&](uint32_t at) -> uint64_t {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t v = prog[at];
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
#else
    return prog[at];
#endif
  };
  uint64_t unit = header(i);
  RT_PROF_START();
  // Two loops, so that the local ray (18 registers) is not a loop-carried value of the loop that visits the units: the
  // inner one runs through the units of ONE transform context and leaves at an ENTER, an EXIT or the END; the outer one
  // makes the next context's ray.  (As one loop, the compiler kept two copies of the whole walk state and moved one
  // into the other twice per unit: 33 moves a unit, a seventh of the kernel's instructions.)
  for (;;) {
  bool more_contexts = false;
  for (;;) {
    const uint32_t kind = RTX_NODE_KIND((uint32_t)unit), idx = RTX_NODE_IDX((uint32_t)unit), count = (uint32_t)(unit >> 32);
    if (kind == PROG_NODE_END) break;
    if (kind == RTX_NODE_ENTER || kind == RTX_NODE_EXIT) { more_contexts = true; break; }
    if (skip_until != 0 && skip_until <= i) skip_until = 0;
    const bool on = active && skip_until == 0;
    RT_PROF_LANES(RT_WAVE_COUNT(on));
    uint32_t next = kind == PROG_NODE_DEFER ? i + 1u : i + prog_record_units(unit);
    // the header of the unit behind this one is read while this one is computed (a box that nobody passes goes elsewhere)
    const uint64_t unit_behind = prog[next];
    if (kind == RTX_NODE_BOX) {
      bool pass = false;
      if (on) {
        RT_TRACE_STEP(RTX_NODE_BOX);
        pass = slab_hit((const double*)(prog + i + 1), lr, tmin, tmax);
        if (!pass) skip_until = count;
      }
      if (!RT_WAVE_ANY(pass)) { next = count; unit = header(next); i = next; RT_PROF_TRIP(6); continue; }   // nobody goes in: every lane that was in waits at the skip target, the others beyond it
    } else if (kind == PROG_NODE_SPHERE_S) {
      if (on) {
        for (uint32_t k = 0; k < count; ++k) {
          double t;
          const DSphere sp = *(const DSphere*)(prog + i + 1 + PROG_SPHERE_S_UNITS * k);
          if (sphere_s_hit_t(sp, lr.o, lr.d, tmin, tmax, t)) { tmax = t; hit_kind = RTX_NODE_MAKE(RTX_NODE_SPHERE, idx + k); hit_ctx = cur; }
        }
        RT_TRACE_STEP(32u + 1u);
      }
    } else if (kind == PROG_NODE_QUAD_AA || kind == PROG_NODE_BOX6 || kind == RTX_NODE_QUAD) {
      if (on) {
        const uint32_t cnt = kind == PROG_NODE_QUAD_AA ? count : (kind == PROG_NODE_BOX6 ? 6u : 1u);
        if (kind == RTX_NODE_QUAD || (lr.skip & 8u)) {   // a general quad, or a compact record met by a ray with non-finite components
          for (uint32_t k = 0; k < cnt; ++k) {
            double t;
            if (quad_hit_t(S.quads[idx + k], lr.o, lr.d, tmin, tmax, t)) { tmax = t; hit_kind = RTX_NODE_MAKE(RTX_NODE_QUAD, idx + k); hit_ctx = cur; }
          }
        } else if (kind == PROG_NODE_BOX6) {
          const DBox6 bx = *(const DBox6*)(prog + i + 1);
RT_UNROLL
          for (int f = 0; f < 6; ++f) {
            double t;
            if (aaquad_hit_t(box6_face(bx, f), lr.o, lr.d, tmin, tmax, t)) { tmax = t; hit_kind = RTX_NODE_MAKE(RTX_NODE_QUAD, idx + (uint32_t)f); hit_ctx = cur; }
          }
        } else {
          for (uint32_t k = 0; k < cnt; ++k) {
            double t;
            const DAAQuad qa = *(const DAAQuad*)(prog + i + 1 + PROG_QUAD_AA_UNITS * k);
            if (aaquad_hit_t(qa, lr.o, lr.d, tmin, tmax, t)) { tmax = t; hit_kind = RTX_NODE_MAKE(RTX_NODE_QUAD, idx + k); hit_ctx = cur; }
          }
        }
        RT_TRACE_STEP(32u + 2u);
      }
    } else if (kind == RTX_NODE_SPHERE) {
      if (on) {
        double t;
        if (sphere_rec_hit_t((const double*)(prog + i + 1), lr.o, lr.d, tm, tmin, tmax, t)) { tmax = t; hit_kind = RTX_NODE_MAKE(RTX_NODE_SPHERE, idx); hit_ctx = cur; }
        RT_TRACE_STEP(32u + 3u);
      }
    } else if (MEDIA != MEDIA_NONE && kind == PROG_NODE_MEDIUM_S && use_media) {
      if (on) {   // constant_medium.rs:42-90 for a boundary of one static sphere: as the MEDIUM_S step of walk_program
        const double neg_inv_density = ((const double*)prog)[i + 1];
        const uint64_t mflags = prog[i + 2];
        const DSphere sp = *(const DSphere*)(prog + i + 3);
        double t_in = 0.0, t_out = 0.0, ray_length = 0.0, hit_distance = 0.0;
        bool crossed = true, drawn = false;
        if ((mflags & PROG_MEDIUM_DRAW_FIRST) && tmin0 >= 0.0) {
          ray_length = norm(lr.d);
          hit_distance = neg_inv_density * log(rng_at(rng, bounce | ((uint32_t)RNG_DOMAIN_MEDIUM << 16), idx));
          drawn = true;
          bool beyond = hit_distance > tmax * ray_length;
          if (!beyond) {
            const V3 oc = lr.o - v3(sp.cx, sp.cy, sp.cz);
            beyond = norm2(oc) <= 1e4 * (sp.r * sp.r) && hit_distance > (2.0 * sp.r) * (1.0 + 1e-9);
          }
          crossed = !beyond;
        }
        crossed = crossed && sphere_s_hit_t(sp, lr.o, lr.d, -RT_INF, RT_INF, t_in) && sphere_s_hit_t(sp, lr.o, lr.d, t_in + 0.0001, RT_INF, t_out);
        if (crossed) {
          double t1 = fmax(t_in, tmin0);
          double t2 = fmin(t_out, tmax);
          if (!(t1 >= t2)) {
            t1 = fmax(t1, 0.0);
            if (!drawn) {
              ray_length = norm(lr.d);
              hit_distance = neg_inv_density * log(rng_at(rng, bounce | ((uint32_t)RNG_DOMAIN_MEDIUM << 16), idx));
            }
            const double inside = (t2 - t1) * ray_length;
            if (!(hit_distance > inside)) {
              if ((mflags & PROG_MEDIUM_NEEDS_UV) && stale_uv) {
                const uint64_t id = (uint64_t)hit_kind | ((uint64_t)hit_ctx << 32);
                stale_uv[0] = tmax;
                memcpy(&stale_uv[1], &id, 8);
              }
              tmax = t1 + hit_distance / ray_length;
              hit_kind = RTX_NODE_MAKE(RTX_NODE_MEDIUM, idx); hit_ctx = cur;
            }
          }
        }
        RT_TRACE_STEP(32u + 9u);
      }
    } else if (kind == PROG_NODE_DEFER) {
      if (on) deferred = true;
    }
    // (filler, SYNC, a medium this query ignores: nothing to do)
    // (tuning build: slots of the per-ray walk's table that it does not use in a kernel that sweeps; 6 = boxes, transforms, rest)
    RT_PROF_TRIP(kind == PROG_NODE_MEDIUM_S ? 5 : kind == PROG_NODE_SPHERE_S ? 1 : kind == RTX_NODE_SPHERE ? 3 : kind == PROG_NODE_QUAD_AA ? 2 : kind == RTX_NODE_QUAD ? 4 : kind == PROG_NODE_BOX6 ? 7 : 6);
    i = next;
#if defined(__HIP_DEVICE_COMPILE__)
    unit = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(unit_behind >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)unit_behind);
#else
    unit = unit_behind;
#endif
  }
  if (!more_contexts) break;
  {   // ENTER / EXIT: the ray of the context that begins here (for the lanes that are in)
    const uint32_t idx = RTX_NODE_IDX((uint32_t)unit), count = (uint32_t)(unit >> 32);
    if (skip_until != 0 && skip_until <= i) skip_until = 0;
    const bool on = active && skip_until == 0;
    RT_PROF_LANES(RT_WAVE_COUNT(on));
    if (on) {
      const V3 wo = cur == 0 ? lr.o : rs.origin(), wd = cur == 0 ? lr.d : rs.dir();   // as in walk_program
      cur = idx;
      lr = lray_from_ops(prog + i + 1, count, wo, wd);
      RT_TRACE_STEP(RTX_NODE_KIND((uint32_t)unit));
    }
    RT_PROF_TRIP(6);
    i += prog_record_units(unit);
    unit = header(i);
  }
  }
  best.t = tmax; best.kind_idx = hit_kind; best.ctx = hit_ctx;
  return deferred;
}

// The whole walk (a DEFER unit is followed at once): world.hit as the reference runs it.
template <bool MEDIA, class RaySrc>
RT_HD void closest_hit(const Scene& S, const uint64_t* prog, uint32_t begin, bool active, const RaySrc& rs, double tm,
                       double tmin0, double tmax0, Hit& best, bool use_media, const PathRng& rng, uint32_t bounce,
                       double* stale_uv) {
  (void)walk_program<MEDIA ? MEDIA_ALL : MEDIA_NONE, false>(S, prog, begin, active, rs, tm, tmin0, tmax0, best, use_media, rng, bounce, stale_uv);
}

RT_HD bool hit_any(const Hit& h) { return RTX_NODE_KIND(h.kind_idx) != RTX_NODE_NONE; }

// A section of the program walked as a STREAM of rays: a lane whose ray has left the section takes the next ray of
// the queue at once instead of idling until the other 63 lanes of its wave are done.  For sections that are one
// box subtree (a BVH and its leaves; no media) lanes differ widely in the number of boxes they visit, and in the
// per-ray form above most box steps run with a quarter of the lanes.  The walk of each single ray is unchanged: same
// units in the same order with the same running tmax, so results are bit-identical to walk_program over the section.
//
// The section starts at unit `begin` (world context) and ends where a lane meets an END or SYNC unit.  `feed` hands
// out rays and takes results (wave-collective on the GPU):
//   bool more()                         the queue may still hold rays (uniform over the wave)
//   bool take(bool want, uint32_t& id)  lanes with `want` receive the next rays; false = none for this lane now
//   Ray source: ray(id) (origin()/dir() as for walk_program), time(id)
//   void start(id, tmax, kind, ctx)     the closest hit known before the section (tmax = RT_INF, kind NONE if none)
//   void commit(id, tmax, kind, ctx)    the closest hit after the section
//   void note_redo()                    (walk_stream_near) a ray is walked a second time, in the reference's order
// Scheduling: box steps run in a tight loop while the lanes on boxes outweigh the lanes waiting on a leaf
// (RT_STREAM_LEAF_BIAS box lanes count like one waiting lane: a leaf test costs several box tests) and fewer than
// RT_STREAM_REFILL lanes are out of work; leaves are served largest class first.
#ifndef RT_STREAM_REFILL
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_STREAM_REFILL 32   // measured optimum on the final scene (8: -4 %, 16: -1.5 %, 48: -2 %, 64 = never: -15 %)
#else
#define RT_STREAM_REFILL 1
#endif
#endif
#ifndef RT_STREAM_LEAF_BIAS
#define RT_STREAM_LEAF_BIAS 4  // 3 to 7 measure the same; 10 is 1 % slower
#endif
template <class Feed>
RT_HD void walk_stream(const Scene& S, const uint64_t* prog, uint32_t begin, Feed& feed) {
  enum { CL_CHEAP = 0, CL_SPHERE_S = 1, CL_QUAD_AA = 2, CL_SPHERE = 3, CL_QUAD = 4, CL_BOX6 = 7, CL_BOX = 8, CL_DONE = 10 };
  const double tmin = 0.001;
  bool have = false;
  uint32_t id = 0, i = begin, cur = 0, hit_kind = RTX_NODE_MAKE(RTX_NODE_NONE, 0), hit_ctx = 0;
  uint64_t unit = (uint64_t)RTX_NODE_MAKE(PROG_NODE_END, 0);
  double tmax = RT_INF, tm = 0.0;
  auto rs = feed.ray(0);
  LRay lr;
  lr.o = v3(0, 0, 0); lr.d = v3(0, 0, 1); lr.inv = v3(0, 0, 1); lr.skip = 0;
  for (;;) {
    {  // lanes whose ray has left the section hand in the result
      const uint32_t k = RTX_NODE_KIND((uint32_t)unit);
      if (have && (k == PROG_NODE_END || k == PROG_NODE_SYNC)) { feed.commit(id, tmax, hit_kind, hit_ctx); have = false; }
    }
    if (feed.more() && RT_WAVE_COUNT(!have) >= RT_STREAM_REFILL) {
      uint32_t nid = 0;
      if (feed.take(!have, nid)) {
        id = nid; rs = feed.ray(id); tm = feed.time(id);
        feed.start(id, tmax, hit_kind, hit_ctx);
        lr = lray_from_ops(prog, 0, rs.origin(), rs.dir());
        cur = 0; i = begin; unit = prog[i]; have = true;
      }
    }
    if (!RT_WAVE_ANY(have)) {
      if (feed.more()) continue;
      break;
    }
    const uint32_t kind_idx = (uint32_t)unit;
    const uint32_t kind = RTX_NODE_KIND(kind_idx), idx = RTX_NODE_IDX(kind_idx), count = (uint32_t)(unit >> 32);
    int cls = CL_CHEAP;
    if (!have || kind == PROG_NODE_END || kind == PROG_NODE_SYNC) cls = CL_DONE;
    else if (kind == RTX_NODE_BOX) cls = CL_BOX;
    else if (kind == PROG_NODE_SPHERE_S) cls = CL_SPHERE_S;
    else if (kind == PROG_NODE_QUAD_AA) cls = (lr.skip & 8u) ? CL_QUAD : CL_QUAD_AA;
    else if (kind == PROG_NODE_BOX6) cls = (lr.skip & 8u) ? CL_QUAD : CL_BOX6;
    else if (kind == RTX_NODE_SPHERE) cls = CL_SPHERE;
    else if (kind == RTX_NODE_QUAD) cls = CL_QUAD;
    uint32_t next = i;
    if (RT_WAVE_ANY(cls == CL_CHEAP)) {
      if (cls == CL_CHEAP) {
        if (kind == RTX_NODE_ENTER || kind == RTX_NODE_EXIT) {
          const V3 wo = cur == 0 ? lr.o : rs.origin(), wd = cur == 0 ? lr.d : rs.dir();  // as in walk_program
          cur = idx;
          lr = lray_from_ops(prog + i + 1, count, wo, wd);
          next = i + 1 + PROG_XFORM_OP_UNITS * count;
        } else {
          next = i + prog_record_units(unit);  // filler (a section holds no media and no deferred part)
        }
      }
    } else {
      const int n_x = RT_WAVE_COUNT(cls == CL_BOX), n_d = RT_WAVE_COUNT(cls == CL_DONE);
      if (n_x > 0 && n_x * RT_STREAM_LEAF_BIAS >= RT_WAVE_LANES - n_x - n_d) {
        int nb, nd;
        do {
          if (have && RTX_NODE_KIND((uint32_t)unit) == RTX_NODE_BOX) {
            const uint32_t i_pass = i + PROG_BOX_UNITS, i_fail = (uint32_t)(unit >> 32);
            RT_TRACE_STEP(RTX_NODE_BOX);
            const uint64_t u_pass = prog[i_pass], u_fail = prog[i_fail];
            const bool pass = slab_hit((const double*)(prog + i + 1), lr, tmin, tmax);
            i = pass ? i_pass : i_fail;
            unit = pass ? u_pass : u_fail;
          }
          const uint32_t k = RTX_NODE_KIND((uint32_t)unit);
          nb = RT_WAVE_COUNT(have && k == RTX_NODE_BOX);
          nd = RT_WAVE_COUNT(!have || k == PROG_NODE_END || k == PROG_NODE_SYNC);
        } while (nb > 0 && nb * RT_STREAM_LEAF_BIAS >= RT_WAVE_LANES - nb - nd && !(feed.more() && nd >= RT_STREAM_REFILL));
        continue;  // i and unit are current
      }
      const int n_ss = RT_WAVE_COUNT(cls == CL_SPHERE_S), n_a = RT_WAVE_COUNT(cls == CL_QUAD_AA), n_s = RT_WAVE_COUNT(cls == CL_SPHERE);
      const int n_q = RT_WAVE_COUNT(cls == CL_QUAD), n_b = RT_WAVE_COUNT(cls == CL_BOX6);
      int pick = CL_BOX, best_n = 0;
      if (n_ss > best_n) { pick = CL_SPHERE_S; best_n = n_ss; }
      if (n_b > best_n) { pick = CL_BOX6; best_n = n_b; }
      if (n_a > best_n) { pick = CL_QUAD_AA; best_n = n_a; }
      if (n_s > best_n) { pick = CL_SPHERE; best_n = n_s; }
      if (n_q > best_n) { pick = CL_QUAD; best_n = n_q; }
      if (pick == CL_SPHERE_S) {
        if (cls == CL_SPHERE_S) {
          for (uint32_t k = 0; k < count; ++k) {
            double t;
            const DSphere sp = *(const DSphere*)(prog + i + 1 + PROG_SPHERE_S_UNITS * k);
            if (sphere_s_hit_t(sp, lr.o, lr.d, tmin, tmax, t)) { tmax = t; hit_kind = RTX_NODE_MAKE(RTX_NODE_SPHERE, idx + k); hit_ctx = cur; }
          }
          next = i + 1 + PROG_SPHERE_S_UNITS * count;
        }
      } else if (pick == CL_BOX6) {
        if (cls == CL_BOX6) {
          const DBox6 bx = *(const DBox6*)(prog + i + 1);
RT_UNROLL
          for (int f = 0; f < 6; ++f) {
            double t;
            if (aaquad_hit_t(box6_face(bx, f), lr.o, lr.d, tmin, tmax, t)) { tmax = t; hit_kind = RTX_NODE_MAKE(RTX_NODE_QUAD, idx + (uint32_t)f); hit_ctx = cur; }
          }
          next = i + PROG_BOX6_UNITS;
        }
      } else if (pick == CL_QUAD_AA) {
        if (cls == CL_QUAD_AA) {
          for (uint32_t k = 0; k < count; ++k) {
            double t;
            const DAAQuad qa = *(const DAAQuad*)(prog + i + 1 + PROG_QUAD_AA_UNITS * k);
            if (aaquad_hit_t(qa, lr.o, lr.d, tmin, tmax, t)) { tmax = t; hit_kind = RTX_NODE_MAKE(RTX_NODE_QUAD, idx + k); hit_ctx = cur; }
          }
          next = i + 1 + PROG_QUAD_AA_UNITS * count;
        }
      } else if (pick == CL_SPHERE) {
        if (cls == CL_SPHERE) {
          double t;
          if (sphere_rec_hit_t((const double*)(prog + i + 1), lr.o, lr.d, tm, tmin, tmax, t)) { tmax = t; hit_kind = RTX_NODE_MAKE(RTX_NODE_SPHERE, idx); hit_ctx = cur; }
          next = i + PROG_SPHERE_UNITS;
        }
      } else if (pick == CL_QUAD) {
        if (cls == CL_QUAD) {
          uint32_t cnt = 1u, size = 1u;
          if (kind == PROG_NODE_QUAD_AA) { cnt = count; size = 1u + PROG_QUAD_AA_UNITS * count; }
          else if (kind == PROG_NODE_BOX6) { cnt = 6u; size = PROG_BOX6_UNITS; }
          for (uint32_t k = 0; k < cnt; ++k) {
            double t;
            if (quad_hit_t(S.quads[idx + k], lr.o, lr.d, tmin, tmax, t)) { tmax = t; hit_kind = RTX_NODE_MAKE(RTX_NODE_QUAD, idx + k); hit_ctx = cur; }
          }
          next = i + size;
        }
      }
    }
    if (next != i) { RT_TRACE_STEP(cls == CL_CHEAP ? kind : 32u + (uint32_t)cls); i = next; unit = prog[i]; }
  }
}

// Aabb::hit (slab_hit above, same arithmetic and decision) that also returns where the clipped interval begins.
RT_HD bool slab_hit_entry(const double* b, const LRay& r, double tmin, double tmax, double& entry) {
  const uint32_t nx = r.inv.x >= 0.0 ? 0u : 3u, ny = r.inv.y >= 0.0 ? 0u : 3u, nz = r.inv.z >= 0.0 ? 0u : 3u;
  const double bnx = b[nx], bfx = b[3u - nx], bny = b[1u + ny], bfy = b[4u - ny], bnz = b[2u + nz], bfz = b[5u - nz];
  double nrx = (bnx - r.o.x) * r.inv.x, frx = (bfx - r.o.x) * r.inv.x;
  double nry = (bny - r.o.y) * r.inv.y, fry = (bfy - r.o.y) * r.inv.y;
  double nrz = (bnz - r.o.z) * r.inv.z, frz = (bfz - r.o.z) * r.inv.z;
  if (r.skip & 7u) {
    if (r.skip & 1u) { nrx = -RT_INF; frx = RT_INF; }
    if (r.skip & 2u) { nry = -RT_INF; fry = RT_INF; }
    if (r.skip & 4u) { nrz = -RT_INF; frz = RT_INF; }
  }
  tmin = rt_max(tmin, nrx); tmax = rt_min(tmax, frx);
  tmin = rt_max(tmin, nry); tmax = rt_min(tmax, fry);
  tmin = rt_max(tmin, nrz); tmax = rt_min(tmax, frz);
  entry = tmin;
  return !(tmax <= tmin);
}
// Relative margin of the nearer-child-first walk's order-dependence watch (Scene::near_margin).
#ifndef RT_NEAR_MARGIN
#define RT_NEAR_MARGIN 1e-9
#endif
// Sphere::hit for a static sphere with an inclusive upper bound: a root equal to tmax is reported, so that the caller
// can resolve the tie by program order (sphere_s_hit_t is this with the tie always lost, sphere.rs:99-101).
RT_HD bool sphere_s_hit_t_incl(const DSphere& s, const V3& o, const V3& d, double tmin, double tmax, double& t) {
  V3 oc = o - v3(s.cx, s.cy, s.cz);
  double a = norm2(d);
  double half_b = dot(oc, d);
  double c = norm2(oc) - s.r * s.r;
  double disc = half_b * half_b - a * c;
  if (disc < 0.0) return false;
  double sqrtd = sqrt(disc);
  double root = (-half_b - sqrtd) / a;
  if (!(tmin < root && root <= tmax)) {
    root = (-half_b + sqrtd) / a;
    if (!(tmin < root && root <= tmax)) return false;
  }
  t = root;
  return true;
}

// walk_stream for a section that is a binary tree of boxes (the sphere cluster and the ground of the final scene; the
// scene says whether a section has that shape), with the NEARER child of every box walked first.  Leaves hold runs of
// static spheres, quads (axis-aligned runs or single general ones) and BOX6 records behind their guard boxes.
// The reference walks left before right whatever the ray's direction (bvh.rs:94-102), so its tmax shrinks late; the
// closest hit does not depend on the visiting order once ties are resolved the way the fixed order resolves them:
// among hits with equal t the FIRST sphere of the program wins (a later one needs t < tmax, sphere.rs:99-101) and the
// LAST quad (a later one needs only t <= tmax, quad.rs:74), and a quad beats a sphere.
// What a different order can change is which boxes the shrinking tmax culls, and that matters only for a hit that lies
// on the very boundary of its own box (to the last bit) and ties with another hit at once: such a ray may find the
// other primitive.  The deterministic first-hit query keeps the reference's order; this walk serves rendering, where
// paths are random and images agree in distribution.
//   stk: per-lane stack of unit positions (lane-private, stk[level * stk_stride]); depth = the tree's depth.
//   root_single: the section starts with ONE box (its root) instead of the two children of a root tested elsewhere.
//   QUADS: leaves may hold quads (otherwise static-sphere runs only, and the quad classes cost nothing).
//   AHEAD: see the box step.
// Per lane: i = a box not tested yet (`single`: only it; otherwise it and its sibling, the two children of a box that
// passed), or a unit of the leaf whose units end at `end`.
#ifndef RT_NEAR_LEAF_BIAS
#define RT_NEAR_LEAF_BIAS RT_STREAM_LEAF_BIAS
#endif
#ifndef RT_NEAR_REFILL
#define RT_NEAR_REFILL RT_STREAM_REFILL
#endif
// ... for a program in global memory (AHEAD = false): a box step waits for memory there, a leaf test costs relatively
// less, and leaves are served sooner (stress scene: bias 4: 79.1, 3: 83.6, 2: 86.7, 1: 88.7-89.4, 1/2: 89.1, 1/4: 87.3
// Msamples/s; box steps only when no lane waits on a leaf: 60.9)
#ifndef RT_NEAR_LEAF_BIAS_FAR
#define RT_NEAR_LEAF_BIAS_FAR 1
#endif
#ifndef RT_NEAR_LEAF_BIAS_FAR_DEN
#define RT_NEAR_LEAF_BIAS_FAR_DEN 1
#endif
//   TOP: the first levels of the tree are also held in a table in heap order (top: 7-unit box records, top_pos: their
//   unit positions, top_n of them; Scene::head_top_*), which the caller keeps in fast memory while the program itself
//   stays in global memory: a lane then carries the heap index of the box it stands on (hx; HX_NONE below the table)
//   and reads the bounds and headers of tabled boxes from the table.  The positions, and so the walk, are unchanged.
template <bool QUADS, bool AHEAD, bool TOP = false, class Feed>
RT_HD void walk_stream_near(const Scene& S, const uint64_t* prog, uint32_t begin, Feed& feed, uint32_t* stk, uint32_t stk_stride, bool root_single,
                            const uint64_t* top = nullptr, const uint32_t* top_pos = nullptr, uint32_t top_n = 0) {
  const uint32_t HX_NONE = 0xFFFFFFFFu;
  uint32_t hx = HX_NONE;   // TOP: heap index of the box at i (of the left one of a pair); its sibling is hx + 1
  enum { CL_CHEAP = 0, CL_SPHERE_S = 1, CL_QUAD_AA = 2, CL_QUAD = 4, CL_BOX6 = 7, CL_BOX = 8, CL_DONE = 10 };
  const int bias = AHEAD ? RT_NEAR_LEAF_BIAS : RT_NEAR_LEAF_BIAS_FAR, bias_den = AHEAD ? 1 : RT_NEAR_LEAF_BIAS_FAR_DEN;
  const double tmin = 0.001;
  bool have = false, single = false, best_quad = false;
  // Order-dependence watch.  The closest hit is the same in every visiting order as long as every candidate hit lies
  // behind the entry of the innermost box around its primitive, t > entry: then a box is only ever culled together
  // with hits that lie behind the upper bound that culled it, whatever that bound was when the order got there.
  // Rounding can break this where a surface touches its own box (the six axis extremes of a sphere): the reference's
  // box test then depends on the bound its left-to-right order has reached.  So this walk (1) culls a box only when
  // its entry lies beyond tmax by more than the relative margin (tmax_hi; a box within the margin is simply visited:
  // visiting more boxes never changes a closest hit, the tie rules go by program position), and (2) marks a ray when a
  // candidate hit does not lie behind its box's entry by more than the margin (boxes of a BVH nest exactly and rounding
  // is monotone, so the innermost box speaks for its ancestors).  A marked ray is not committed: its lane walks it once
  // more from the start in the REFERENCE's order (fixed_mode: exact bound; the left child first whenever it passes; the
  // right one, tested together with it, waits on the stack and is tested again with the bound the left subtree leaves
  // behind, which is when and how bvh.rs:94-102 tests it).  With rounding errors far below the margin, an unmarked
  // ray's hit is the reference's: no box either order culls holds a hit that could win or tie.
  bool suspect = false, fixed_mode = false;
  double leaf_entry = -RT_INF;      // entry of the last box this lane went through
  double tmax_hi = RT_INF;          // tmax plus the margin
  uint32_t id = 0, i = begin, end = 0xFFFFFFFFu, sp = 0, cur = 0, hit_kind = RTX_NODE_MAKE(RTX_NODE_NONE, 0), hit_ctx = 0, best_pos = 0;
  uint64_t unit = (uint64_t)RTX_NODE_MAKE(PROG_NODE_END, 0);
  double tmax = RT_INF;
  auto rs = feed.ray(0);
  LRay lr;
  lr.o = v3(0, 0, 0); lr.d = v3(0, 0, 1); lr.inv = v3(0, 0, 1); lr.skip = 0;
  // a hit at t from the primitive at program position pos: does it replace the best one so far?
  auto wins = [&](double t, uint32_t pos, bool quad) {
    if (!fixed_mode && t <= leaf_entry + fabs(leaf_entry) * S.near_margin) suspect = true;
    if (t < tmax) return true;
    if (!(t == tmax)) return false;
    if (!QUADS) return !best_quad && pos < best_pos;
    if (quad != best_quad) return quad;              // a quad takes a tie from a sphere in either order
    return quad ? pos > best_pos : pos < best_pos;
  };
  for (;;) {
    if (have) {  // a lane whose units ran out takes the far child it left behind, or hands in its result
      const uint32_t k = RTX_NODE_KIND((uint32_t)unit);
      if (i >= end || k == PROG_NODE_END || k == PROG_NODE_SYNC) {
        if (sp == 0) {
          if (suspect && !fixed_mode) {   // once more, in the reference's order
            feed.note_redo();
            feed.start(id, tmax, hit_kind, hit_ctx); tmax_hi = tmax + tmax * S.near_margin;
            lr = lray_from_ops(prog, 0, rs.origin(), rs.dir());
            cur = 0; i = begin; unit = prog[i]; single = root_single; end = 0xFFFFFFFFu;
            hx = (TOP && top_n) ? 0u : HX_NONE;
            suspect = false; fixed_mode = true; leaf_entry = -RT_INF;
            best_pos = 0; best_quad = RTX_NODE_KIND(hit_kind) == RTX_NODE_QUAD;
          } else {
            feed.commit(id, tmax, hit_kind, hit_ctx); have = false;
          }
        } else {
          sp -= 1;
          const uint32_t e = stk[sp * stk_stride];
          if (TOP && (e & 0x80000000u)) { hx = e & 0x7FFFFFFFu; i = top_pos[hx]; unit = top[hx * PROG_BOX_UNITS]; }
          else { i = e; unit = prog[i]; hx = HX_NONE; }
          single = true; end = 0xFFFFFFFFu;
        }
      }
    }
    if (feed.more() && RT_WAVE_COUNT(!have) >= RT_NEAR_REFILL) {
      uint32_t nid = 0;
      if (feed.take(!have, nid)) {
        id = nid; rs = feed.ray(id);
        feed.start(id, tmax, hit_kind, hit_ctx); tmax_hi = tmax + tmax * S.near_margin;
        lr = lray_from_ops(prog, 0, rs.origin(), rs.dir());
        cur = 0; i = begin; unit = prog[i]; have = true; single = root_single; end = 0xFFFFFFFFu; sp = 0;
        hx = (TOP && top_n) ? 0u : HX_NONE;
        suspect = false; fixed_mode = false; leaf_entry = -RT_INF;
        best_pos = 0; best_quad = RTX_NODE_KIND(hit_kind) == RTX_NODE_QUAD;   // a hit from before the section: position 0
      }
    }
    if (!RT_WAVE_ANY(have)) {
      if (feed.more()) continue;
      break;
    }
    const uint32_t kind_idx = (uint32_t)unit;
    const uint32_t kind = RTX_NODE_KIND(kind_idx), idx = RTX_NODE_IDX(kind_idx), count = (uint32_t)(unit >> 32);
    int cls = CL_CHEAP;
    if (!have || i >= end || kind == PROG_NODE_END || kind == PROG_NODE_SYNC) cls = CL_DONE;
    else if (kind == RTX_NODE_BOX) cls = CL_BOX;
    else if (kind == PROG_NODE_SPHERE_S) cls = CL_SPHERE_S;
    else if (QUADS && kind == PROG_NODE_BOX6) cls = (lr.skip & 8u) ? CL_QUAD : CL_BOX6;   // as in walk_stream
    else if (QUADS && kind == PROG_NODE_QUAD_AA) cls = (lr.skip & 8u) ? CL_QUAD : CL_QUAD_AA;
    else if (QUADS && kind == RTX_NODE_QUAD) cls = CL_QUAD;
    if (RT_WAVE_ANY(cls == CL_CHEAP)) {
      if (cls == CL_CHEAP) {
        uint32_t next;
        if (kind == RTX_NODE_ENTER || kind == RTX_NODE_EXIT) {
          const V3 wo = cur == 0 ? lr.o : rs.origin(), wd = cur == 0 ? lr.d : rs.dir();
          cur = idx;
          lr = lray_from_ops(prog + i + 1, count, wo, wd);
          next = i + 1 + PROG_XFORM_OP_UNITS * count;
        } else {
          next = i + prog_record_units(unit);
        }
        i = next; unit = prog[i];
      }
      continue;
    }
    const int n_x = RT_WAVE_COUNT(cls == CL_BOX), n_ss = RT_WAVE_COUNT(cls == CL_SPHERE_S);
    const int n_b = QUADS ? RT_WAVE_COUNT(cls == CL_BOX6) : 0, n_a = QUADS ? RT_WAVE_COUNT(cls == CL_QUAD_AA) : 0, n_q = QUADS ? RT_WAVE_COUNT(cls == CL_QUAD) : 0;
    const int n_leaf = n_ss + n_b + n_a + n_q;
    if (n_x > 0 && (n_leaf == 0 || n_x * bias >= n_leaf * bias_den)) {
      // box steps in a tight loop, while the lanes on boxes outweigh the lanes waiting on a leaf
      bool on_box = cls == CL_BOX;
      int nb, ns, nd;
      do {
        if (on_box) {
          const uint32_t c0 = i, c1 = (uint32_t)(unit >> 32);
          if (QUADS && RTX_NODE_IDX((uint32_t)unit) == PROG_GUARD_IDX) {
            // the guard in front of a BOX6 record: on to the record, or past it
            const bool pass = slab_hit((const double*)(prog + c0 + 1), lr, tmin, tmax);
            i = pass ? c0 + PROG_BOX_UNITS : c1; unit = prog[i < end ? i : begin];
          } else {
            // the box at i, and unless `single` its sibling (the two children of the box this lane came through)
            // TOP: which of the two boxes are in the table (a pair is, or is not, as a whole: the table holds whole levels)
            const bool t0 = TOP && hx < top_n, t1 = TOP && !single && t0;
            const uint64_t* r0 = t0 ? top + hx * PROG_BOX_UNITS : prog + c0;
            const uint64_t* r1 = t1 ? top + (hx + 1u) * PROG_BOX_UNITS : prog + c1;
            const uint64_t u1 = single ? unit : r1[0];
            // AHEAD: the units behind both boxes are read together with the bounds; the walk goes on with one of them
            // (the image is padded past its END).  It pays a little for a program in LDS (one dependent read less per
            // level: final scene 2257-2265 -> 2268-2289 Msamples/s over repeated runs), not for one in global memory,
            // where the unit behind the box that is not taken is one more cache line to fetch (stress scene 79.3 -> 72.7).
            const uint64_t n0 = AHEAD ? prog[c0 + PROG_BOX_UNITS] : 0ull, n1 = (AHEAD && !single) ? prog[c1 + PROG_BOX_UNITS] : n0;
            double e0 = 0.0, e1 = 0.0;
            const double bound = fixed_mode ? tmax : tmax_hi;          // the watch's margin, or the reference's exact bound
            const bool p0 = slab_hit_entry((const double*)(r0 + 1), lr, tmin, bound, e0);
            const bool p1 = !single && slab_hit_entry((const double*)(r1 + 1), lr, tmin, bound, e1);
            if (p0 || p1) {
              const bool first0 = p0 && (!p1 || fixed_mode || e0 <= e1);   // equal entries: the reference's order
              if (p0 && p1) {   // the far child waits on the stack: by heap index while it is in the table, else by position
                const uint32_t far_pos = first0 ? c1 : c0, far_hx = first0 ? hx + 1u : hx;
                stk[sp * stk_stride] = t1 ? (0x80000000u | far_hx) : far_pos; sp += 1;
              }
              const uint32_t c = first0 ? c0 : c1;
              const double e = first0 ? e0 : e1;
              leaf_entry = e;
              end = (uint32_t)((first0 ? unit : u1) >> 32);
              i = c + PROG_BOX_UNITS; single = false;
              if (TOP) {   // the children of the box just entered: heap 2h+1 and 2h+2, in the table or not together
                const uint32_t hc = t0 ? (first0 ? hx : hx + 1u) : HX_NONE;
                hx = (hc != HX_NONE && 2u * hc + 2u < top_n) ? 2u * hc + 1u : HX_NONE;
                unit = hx != HX_NONE ? top[hx * PROG_BOX_UNITS] : prog[i];
              } else {
                unit = AHEAD ? (first0 ? n0 : n1) : prog[i];
              }
            } else if (sp > 0) {                                       // nothing here: on to the far child left behind
              sp -= 1;
              const uint32_t e = stk[sp * stk_stride];
              if (TOP && (e & 0x80000000u)) { hx = e & 0x7FFFFFFFu; i = top_pos[hx]; unit = top[hx * PROG_BOX_UNITS]; }
              else { i = e; unit = prog[i]; hx = HX_NONE; }
              single = true; end = 0xFFFFFFFFu;
            } else {
              i = 0xFFFFFFFFu - 1u; end = 0u;                          // the section is done: the top of the loop hands in
            }
          }
          RT_TRACE_STEP(RTX_NODE_BOX);
        }
        const uint32_t k2 = RTX_NODE_KIND((uint32_t)unit);
        const bool fin = !have || i >= end || k2 == PROG_NODE_END || k2 == PROG_NODE_SYNC;
        on_box = !fin && k2 == RTX_NODE_BOX;
        nb = RT_WAVE_COUNT(on_box); nd = RT_WAVE_COUNT(fin);
        ns = RT_WAVE_LANES - nb - nd;
      } while (nb > 0 && (ns == 0 || nb * bias >= ns * bias_den) && !(nd >= RT_NEAR_REFILL));
      continue;
    }
    // leaves: the class most lanes wait for
    int pick = CL_SPHERE_S, best_n = n_ss;
    if (QUADS && n_b > best_n) { pick = CL_BOX6; best_n = n_b; }
    if (QUADS && n_a > best_n) { pick = CL_QUAD_AA; best_n = n_a; }
    if (QUADS && n_q > best_n) { pick = CL_QUAD; best_n = n_q; }
    uint32_t next = i;
    if (pick == CL_SPHERE_S) {
      if (cls == CL_SPHERE_S) {
        for (uint32_t k = 0; k < count; ++k) {
          double t;
          const DSphere sph = *(const DSphere*)(prog + i + 1 + PROG_SPHERE_S_UNITS * k);
          const uint32_t pos = (i << 6) + k + 1u;                      // program order of this sphere (0 = before the section)
          if (sphere_s_hit_t_incl(sph, lr.o, lr.d, tmin, tmax, t) && wins(t, pos, false)) {
            tmax = t; tmax_hi = t + t * S.near_margin; hit_kind = RTX_NODE_MAKE(RTX_NODE_SPHERE, idx + k); hit_ctx = cur; best_pos = pos; best_quad = false;
          }
        }
        RT_TRACE_STEP(32u + (uint32_t)CL_SPHERE_S);
        next = i + 1 + PROG_SPHERE_S_UNITS * count;
      }
    } else if (QUADS && pick == CL_BOX6) {
      if (cls == CL_BOX6) {
        const DBox6 bx = *(const DBox6*)(prog + i + 1);
RT_UNROLL
        for (int f = 0; f < 6; ++f) {
          double t;
          const uint32_t pos = (i << 6) + (uint32_t)f + 1u;
          if (aaquad_hit_t(box6_face(bx, f), lr.o, lr.d, tmin, tmax, t) && wins(t, pos, true)) {
            tmax = t; tmax_hi = t + t * S.near_margin; hit_kind = RTX_NODE_MAKE(RTX_NODE_QUAD, idx + (uint32_t)f); hit_ctx = cur; best_pos = pos; best_quad = true;
          }
        }
        RT_TRACE_STEP(32u + (uint32_t)CL_BOX6);
        next = i + PROG_BOX6_UNITS;
      }
    } else if (QUADS && pick == CL_QUAD) {
      if (cls == CL_QUAD) {   // quads through the ABI's table: general ones, and compact records when the ray is not finite
        uint32_t cnt = 1u, size = 1u;
        if (kind == PROG_NODE_QUAD_AA) { cnt = count; size = 1u + PROG_QUAD_AA_UNITS * count; }
        else if (kind == PROG_NODE_BOX6) { cnt = 6u; size = PROG_BOX6_UNITS; }
        for (uint32_t k = 0; k < cnt; ++k) {
          double t;
          const uint32_t pos = (i << 6) + k + 1u;
          if (quad_hit_t(S.quads[idx + k], lr.o, lr.d, tmin, tmax, t) && wins(t, pos, true)) {
            tmax = t; tmax_hi = t + t * S.near_margin; hit_kind = RTX_NODE_MAKE(RTX_NODE_QUAD, idx + k); hit_ctx = cur; best_pos = pos; best_quad = true;
          }
        }
        RT_TRACE_STEP(32u + (uint32_t)CL_QUAD);
        next = i + size;
      }
    } else if (QUADS) {
      if (cls == CL_QUAD_AA) {
        for (uint32_t k = 0; k < count; ++k) {
          double t;
          const DAAQuad qa = *(const DAAQuad*)(prog + i + 1 + PROG_QUAD_AA_UNITS * k);
          const uint32_t pos = (i << 6) + k + 1u;
          if (aaquad_hit_t(qa, lr.o, lr.d, tmin, tmax, t) && wins(t, pos, true)) {
            tmax = t; tmax_hi = t + t * S.near_margin; hit_kind = RTX_NODE_MAKE(RTX_NODE_QUAD, idx + k); hit_ctx = cur; best_pos = pos; best_quad = true;
          }
        }
        RT_TRACE_STEP(32u + (uint32_t)CL_QUAD_AA);
        next = i + 1 + PROG_QUAD_AA_UNITS * count;
      }
    }
    if (next != i) { i = next; unit = prog[i < end ? i : begin]; }
  }
}

// Aabb::hit (slab_hit_entry: same operations, same decision) on bounds held in registers.
RT_HD bool slab_hit_entry_v(double lx, double ly, double lz, double hx, double hy, double hz, const LRay& r, double tmin, double tmax, double& entry) {
  const bool px = r.inv.x >= 0.0, py = r.inv.y >= 0.0, pz = r.inv.z >= 0.0;
  const double bnx = px ? lx : hx, bfx = px ? hx : lx, bny = py ? ly : hy, bfy = py ? hy : ly, bnz = pz ? lz : hz, bfz = pz ? hz : lz;
  double nrx = (bnx - r.o.x) * r.inv.x, frx = (bfx - r.o.x) * r.inv.x;
  double nry = (bny - r.o.y) * r.inv.y, fry = (bfy - r.o.y) * r.inv.y;
  double nrz = (bnz - r.o.z) * r.inv.z, frz = (bfz - r.o.z) * r.inv.z;
  if (r.skip & 7u) {
    if (r.skip & 1u) { nrx = -RT_INF; frx = RT_INF; }
    if (r.skip & 2u) { nry = -RT_INF; fry = RT_INF; }
    if (r.skip & 4u) { nrz = -RT_INF; frz = RT_INF; }
  }
  tmin = rt_max(tmin, nrx); tmax = rt_min(tmax, frx);
  tmin = rt_max(tmin, nry); tmax = rt_min(tmax, fry);
  tmin = rt_max(tmin, nrz); tmax = rt_min(tmax, frz);
  entry = tmin;
  return !(tmax <= tmin);
}
// A padded box of the folded table against a ray, in single precision on the table's floats as they are: near and far
// plane distances by one fused multiply-add each from inv and oi = o * inv rounded to float.  Only for culling: see
// CompiledProgram::build_head_wide for why it never fails a box that Aabb::hit passes.  Their error is at most (|bound| + |origin|) |inv| 1.2e-7.
// The caller measures ray parameters from where the ray enters the section's root box (oi = o * inv + shift is the
// same ray with its origin moved there: d * inv = 1 on every axis that counts), so |origin| is within the section's
// extent whatever the ray's own origin, and the table's pad (PROG_WIDE_PAD of the extent) exceeds the error eight
// times over; tmin and tmax, measured from the same point, must be rounded outward by the caller.
// For a ray that skips no axis (the caller sends the others the program's way), without a branch: the entry, or
// +infinity when the ray misses the box.
RT_HD float slab_entry_or_inf_f(float lx, float ly, float lz, float hx, float hy, float hz, float ix, float iy, float iz, float ox, float oy, float oz,
                                float tmin, float tmax) {
  const float ax = fmaf(lx, ix, -ox), bx = fmaf(hx, ix, -ox);
  const float ay = fmaf(ly, iy, -oy), by = fmaf(hy, iy, -oy);
  const float az = fmaf(lz, iz, -oz), bz = fmaf(hz, iz, -oz);
  tmin = fmaxf(tmin, fmaxf(fminf(ax, bx), fmaxf(fminf(ay, by), fminf(az, bz))));
  tmax = fminf(tmax, fminf(fmaxf(ax, bx), fminf(fmaxf(ay, by), fmaxf(az, bz))));
  return tmax <= tmin ? (float)RT_INF : tmin;
}
RT_HD float unit_f32f(uint64_t u, int high) { const uint32_t w = (uint32_t)(high ? u >> 32 : u); float f; memcpy(&f, &w, 4); return f; }
RT_HD double unit_f64(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
RT_HD double unit_f32(uint64_t u, int high) { const uint32_t w = (uint32_t)(high ? u >> 32 : u); float f; memcpy(&f, &w, 4); return (double)f; }
struct alignas(16) UnitPair { uint64_t x, y; };

// walk_stream_near for a head section that stays in global memory, over the table that folds two levels of its box
// tree into one record (Scene::head_wide_*, CompiledProgram::build_head_wide).  Every step of a lane is one read of
// 128 bytes at an address the lane knows, and EVERY lane of the wave takes its step in every trip of the loop: first
// all lanes issue their reads, then each kind of step is computed for the lanes that are on it.  A walk whose every
// step is a cache miss is bound by round trips to memory, and this way a round trip serves the whole wave instead of
// the lanes of one kind of step.  The steps:
//   record   the four boxes of a table record (floats, rounded outward: the table only culls) are tested together;
//            the lane goes on with the nearest slot that passes and leaves the others on its stack, farthest first.
//            The tree is walked in half as many dependent reads as box by box.
//   leaf     the program's box record in front of a leaf's units and the units behind it, in one read: the box is
//            tested as the reference tests it (same doubles, Aabb::hit, with the bound of that moment), sphere records
//            are tested from the registers, general quads are noted (two at most; whatever else a leaf holds is
//            tested record by record from memory as walk_stream_near does).
//   quad     a noted quad's record from the scene's table.
// Hits are compared as in walk_stream_near (tie rules by program position, so the order of the tests is free), with
// the same watch: a candidate hit that does not lie behind the entry of its leaf's box by more than the margin marks
// the ray.  A marked ray, and any ray that is not finite, is walked again by its lane alone over the PROGRAM in the
// reference's order (bvh.rs:94-102) with the exact bound, which is what is committed for it.
//   stk: per-lane stack (lane-private, stk[level * stk_stride]), 3 entries per level of the table
//        (Scene::head_wide_depth); an entry is the record of a slot with box children, or 0x80000000 | the unit
//        position of a leaf slot's units.
//   begin: the section's root box in the program (tested first, alone); wide: the table.
// What a walk knows of its closest hit so far, for the record-by-record walk below.
struct WideBest {
  double tmax, leaf_entry;
  uint32_t hit_kind, best_pos;
  bool best_quad, suspect, fixed_mode;
};
// Records [p, end) of the program tested one after the other from memory, in the program's order, boxes included (a box
// that fails is skipped with everything behind it, bvh.rs:94-102); stops at the end of a program or section.  The slow
// path of walk_stream_wide: the records of a leaf its reads do not cover, and the whole section for a ray that is
// walked again in the reference's order (fixed_mode: no watch).  Kept out of line: the registers its record kinds
// need would otherwise be charged to the loop that calls it once in a while (RT_NOINLINE; measured slower out of line:
// the call's register saves outweigh the gain, so it is inlined by default).
#ifndef RT_NOINLINE
#define RT_NOINLINE
#endif
template <bool QUADS>
RT_HD RT_NOINLINE WideBest wide_walk_records(const uint64_t* prog, const rtx_quad* quads, double margin, LRay lr, WideBest b, uint32_t p, uint32_t end) {
  const double tmin = 0.001;
  auto wins = [&](double t, uint32_t pos, bool quad) {   // as in walk_stream_near
    if (!b.fixed_mode && t <= b.leaf_entry + fabs(b.leaf_entry) * margin) b.suspect = true;
    if (t < b.tmax) return true;
    if (!(t == b.tmax)) return false;
    if (!QUADS) return !b.best_quad && pos < b.best_pos;
    if (quad != b.best_quad) return quad;
    return quad ? pos > b.best_pos : pos < b.best_pos;
  };
  auto take_hit = [&](double t, uint32_t kind_idx, uint32_t pos, bool quad) { b.tmax = t; b.hit_kind = kind_idx; b.best_pos = pos; b.best_quad = quad; };
  while (p < end) {
    const uint64_t u = prog[p];
    const uint32_t kind = RTX_NODE_KIND((uint32_t)u), idx = RTX_NODE_IDX((uint32_t)u), count = (uint32_t)(u >> 32);
    double t;
    if (kind == PROG_NODE_END || kind == PROG_NODE_SYNC) break;
    if (kind == RTX_NODE_BOX) {
      // A box of the tree inside a leaf slot (a list that holds a box next to primitives): the watch's rules - cull with
      // the margin, compare a hit with the entry of the innermost box around it - are not kept below it, so outside
      // the fixed mode the ray is simply handed to the walk in the reference's order.  (The widened guard in front of
      // a six-face record may be culled with any bound: a face hit within the bound passes it.)
      if (!b.fixed_mode && idx != PROG_GUARD_IDX) b.suspect = true;
      p = slab_hit((const double*)(prog + p + 1), lr, tmin, b.tmax) ? p + PROG_BOX_UNITS : count;
    } else if (kind == PROG_NODE_SPHERE_S) {
      for (uint32_t k = 0; k < count; ++k) {
        const DSphere sph = *(const DSphere*)(prog + p + 1 + PROG_SPHERE_S_UNITS * k);
        const uint32_t pos = (p << 6) + k + 1u;
        if (sphere_s_hit_t_incl(sph, lr.o, lr.d, tmin, b.tmax, t) && wins(t, pos, false)) take_hit(t, RTX_NODE_MAKE(RTX_NODE_SPHERE, idx + k), pos, false);
      }
      p += 1u + PROG_SPHERE_S_UNITS * count;
    } else if (QUADS && (kind == RTX_NODE_QUAD || ((lr.skip & 8u) && (kind == PROG_NODE_QUAD_AA || kind == PROG_NODE_BOX6)))) {
      uint32_t cnt = 1u, size = 1u;
      if (kind == PROG_NODE_QUAD_AA) { cnt = count; size = 1u + PROG_QUAD_AA_UNITS * count; }
      else if (kind == PROG_NODE_BOX6) { cnt = 6u; size = PROG_BOX6_UNITS; }
      for (uint32_t k = 0; k < cnt; ++k) {
        const uint32_t pos = (p << 6) + k + 1u;
        if (quad_hit_t(quads[idx + k], lr.o, lr.d, tmin, b.tmax, t) && wins(t, pos, true)) take_hit(t, RTX_NODE_MAKE(RTX_NODE_QUAD, idx + k), pos, true);
      }
      p += size;
    } else if (QUADS && kind == PROG_NODE_BOX6) {
      const DBox6 bx = *(const DBox6*)(prog + p + 1);
      for (int f = 0; f < 6; ++f) {
        const uint32_t pos = (p << 6) + (uint32_t)f + 1u;
        if (aaquad_hit_t(box6_face(bx, f), lr.o, lr.d, tmin, b.tmax, t) && wins(t, pos, true)) take_hit(t, RTX_NODE_MAKE(RTX_NODE_QUAD, idx + (uint32_t)f), pos, true);
      }
      p += PROG_BOX6_UNITS;
    } else if (QUADS && kind == PROG_NODE_QUAD_AA) {
      for (uint32_t k = 0; k < count; ++k) {
        const DAAQuad qa = *(const DAAQuad*)(prog + p + 1 + PROG_QUAD_AA_UNITS * k);
        const uint32_t pos = (p << 6) + k + 1u;
        if (aaquad_hit_t(qa, lr.o, lr.d, tmin, b.tmax, t) && wins(t, pos, true)) take_hit(t, RTX_NODE_MAKE(RTX_NODE_QUAD, idx + k), pos, true);
      }
      p += 1u + PROG_QUAD_AA_UNITS * count;
    } else {
      p += prog_record_units(u);
    }
  }
  return b;
}

#ifndef RT_WIDE_MIN_LANES
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_WIDE_MIN_LANES 12   // lanes that must wait for a kind of step before the wave computes it, unless it is the most wanted kind (stress scene: 0: 192, 8: 207, 12: 211, 16: 208, 24: 201 Msamples/s)
#else
#define RT_WIDE_MIN_LANES 0
#endif
#endif
#ifndef RT_WIDE_MIN_NODE
#define RT_WIDE_MIN_NODE RT_WIDE_MIN_LANES
#endif
#ifndef RT_WIDE_MIN_LEAF
#define RT_WIDE_MIN_LEAF RT_WIDE_MIN_LANES
#endif
#ifndef RT_WIDE_MIN_QUAD
#define RT_WIDE_MIN_QUAD RT_WIDE_MIN_LANES
#endif
#ifndef RT_WIDE_REFILL
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_WIDE_REFILL 16   // lanes out of work at which the wave takes new rays
#else
#define RT_WIDE_REFILL 1
#endif
#endif
template <bool QUADS, class Feed>
RT_HD void walk_stream_wide(const Scene& S, const uint64_t* prog, uint32_t begin, Feed& feed, uint32_t* stk, uint32_t stk_stride, const uint64_t* wide) {
  enum { ST_NODE = 0, ST_LEAF = 1, ST_QUAD = 2, ST_POP = 3 };
  const double tmin = 0.001;
  bool have = false, best_quad = false, suspect = false;
  double leaf_entry = -RT_INF, tmax_hi = RT_INF, tmax = RT_INF;
  uint32_t id = 0, st = ST_POP, node = 0, i = begin, sp = 0 /* top of the lane's stack as an offset into stk: entries x stk_stride */, hit_kind = RTX_NODE_MAKE(RTX_NODE_NONE, 0), hit_ctx = 0, best_pos = 0;
  uint32_t nq = 0, qa_idx = 0, qa_pos = 0, qb_idx = 0, qb_pos = 0;   // quads of the current leaf still to test
  float shift = 0.0f;   // the table's tests measure ray parameters from here: where the ray enters the root box
  auto rs = feed.ray(0);
  LRay lr;
  lr.o = v3(0, 0, 0); lr.d = v3(0, 0, 1); lr.inv = v3(0, 0, 1); lr.skip = 0;
  auto wins = [&](double t, uint32_t pos, bool quad) {   // as in walk_stream_near
    if (t <= leaf_entry + fabs(leaf_entry) * S.near_margin) suspect = true;
    if (t < tmax) return true;
    if (!(t == tmax)) return false;
    if (!QUADS) return !best_quad && pos < best_pos;
    if (quad != best_quad) return quad;
    return quad ? pos > best_pos : pos < best_pos;
  };
  auto take_hit = [&](double t, uint32_t kind_idx, uint32_t pos, bool quad) {
    tmax = t; tmax_hi = t + t * S.near_margin; hit_kind = kind_idx; hit_ctx = 0; best_pos = pos; best_quad = quad;
  };
  auto sphere_v = [&](uint64_t cx, uint64_t cy, uint64_t cz, uint64_t r, uint32_t idx, uint32_t pos) {
    DSphere sph; sph.cx = unit_f64(cx); sph.cy = unit_f64(cy); sph.cz = unit_f64(cz); sph.r = unit_f64(r);
    double t;
    if (sphere_s_hit_t_incl(sph, lr.o, lr.d, tmin, tmax, t) && wins(t, pos, false)) take_hit(t, RTX_NODE_MAKE(RTX_NODE_SPHERE, idx), pos, false);
  };
  // records [p, end) from memory, out of line
  auto records = [&](uint32_t p, uint32_t end, bool fixed) {
    WideBest b;
    b.tmax = tmax; b.leaf_entry = leaf_entry; b.hit_kind = hit_kind; b.best_pos = best_pos; b.best_quad = best_quad; b.suspect = suspect; b.fixed_mode = fixed;
    b = wide_walk_records<QUADS>(prog, S.quads, S.near_margin, lr, b, p, end);
    tmax = b.tmax; tmax_hi = b.tmax + b.tmax * S.near_margin; hit_kind = b.hit_kind; best_pos = b.best_pos; best_quad = b.best_quad; suspect = b.suspect;
  };
  auto note_quad = [&](uint32_t idx, uint32_t pos) {
    if (nq == 0) { qa_idx = idx; qa_pos = pos; } else { qb_idx = idx; qb_pos = pos; }
    nq += 1;
  };
  RT_WPROF_DECL();
  for (;;) {
    const long long wp_top = RT_WPROF_CLOCK(); (void)wp_top;
    if (have && st == ST_POP) {   // nothing left where the lane stood: the nearest box left behind, or hand in
      if (sp == 0) {
        if (suspect) {   // once more over the program, this lane alone, in the reference's order
          feed.note_redo();
          feed.start(id, tmax, hit_kind, hit_ctx);
          best_pos = 0; best_quad = RTX_NODE_KIND(hit_kind) == RTX_NODE_QUAD;
          records(begin, 0xFFFFFFFFu, true);
          suspect = false;
        }
        feed.commit(id, tmax, hit_kind, hit_ctx); have = false;
      } else {
        sp -= stk_stride;
        const uint32_t e = stk[sp];
        if (e & 0x80000000u) { i = e & 0x7FFFFFFFu; st = ST_LEAF; }
        else { node = e; st = ST_NODE; }
      }
    }
    if (feed.more() && RT_WAVE_COUNT(!have) >= RT_WIDE_REFILL) {
      uint32_t nid = 0;
      if (feed.take(!have, nid)) {
        id = nid; rs = feed.ray(id);
        feed.start(id, tmax, hit_kind, hit_ctx); tmax_hi = tmax + tmax * S.near_margin;
        lr = lray_from_ops(prog, 0, rs.origin(), rs.dir());
        have = true; sp = 0; nq = 0; leaf_entry = -RT_INF;
        best_pos = 0; best_quad = RTX_NODE_KIND(hit_kind) == RTX_NODE_QUAD;
        // the root box, alone, as the reference tests it; its children are the table's first record.  A ray that is
        // not finite, or starts too far out for the table's pad to cover the rounding of its tests, goes the program's way.
        double root_entry = 0.0;
        const bool root = slab_hit_entry((const double*)(prog + begin + 1), lr, tmin, tmax_hi, root_entry);
        shift = (float)root_entry;
        // (a ray parallel to an axis too: the table's box test below has no case for skipped axes)
        const bool far = lr.skip != 0u || !(fabs(lr.o.x) <= S.head_wide_far && fabs(lr.o.y) <= S.head_wide_far && fabs(lr.o.z) <= S.head_wide_far);
        suspect = root && far;
        node = 0; st = (root && !far) ? ST_NODE : ST_POP;
      }
    }
    if (!RT_WAVE_ANY(have)) {
      if (feed.more()) continue;
      break;
    }
    // every lane's read: 16 units, at 16-byte granularity from the table of records or of quads, or of the program
    // from the box record in front of a leaf's units on
    bool on_node = have && st == ST_NODE, on_leaf = have && st == ST_LEAF, on_quad = QUADS && have && st == ST_QUAD;
#if RT_WIDE_MIN_LANES > 0
    {
      // A kind of step is computed in this trip only when enough lanes wait for it to be worth the wave's time, or when
      // it is the kind most lanes wait for; the other lanes keep their place (and issue no read) until a later trip.
      const int n_node = RT_WAVE_COUNT(on_node), n_leaf = RT_WAVE_COUNT(on_leaf), n_quad = RT_WAVE_COUNT(on_quad);
      const int top = n_node > n_leaf ? (n_node > n_quad ? n_node : n_quad) : (n_leaf > n_quad ? n_leaf : n_quad);
      if (n_node < RT_WIDE_MIN_NODE && n_node < top) on_node = false;
      if (n_leaf < RT_WIDE_MIN_LEAF && n_leaf < top) on_leaf = false;
      if (n_quad < RT_WIDE_MIN_QUAD && n_quad < top) on_quad = false;
      RT_WPROF(0, 1); RT_WPROF(1, RT_WAVE_COUNT(have)); RT_WPROF(2, n_node); RT_WPROF(3, n_leaf); RT_WPROF(4, n_quad);
      RT_WPROF(5, RT_WAVE_ANY(on_node) ? 1 : 0); RT_WPROF(6, RT_WAVE_COUNT(on_node));
      RT_WPROF(7, RT_WAVE_ANY(on_leaf) ? 1 : 0); RT_WPROF(8, RT_WAVE_COUNT(on_leaf));
      RT_WPROF(9, RT_WAVE_ANY(on_quad) ? 1 : 0); RT_WPROF(10, RT_WAVE_COUNT(on_quad));
    }
#endif
    const long long wp_t0 = RT_WPROF_CLOCK(); (void)wp_t0;
    RT_WPROF(18, wp_t0 - wp_top);
    uint64_t blob[16], blob16 = 0ull;   // blob16: the 17th unit of a leaf's read (two single spheres need it)
#if defined(__HIP_DEVICE_COMPILE__)
    // a lane that reads nothing in this trip uses none of these: they only have to be DEFINED, which an empty asm
    // statement with an output says without an instruction (zeroing them is 32 moves a trip, 7 % of the walk's issue)
RT_UNROLL
    for (int k = 0; k < 16; ++k) asm volatile("" : "=v"(blob[k]));
#else
RT_UNROLL
    for (int k = 0; k < 16; ++k) blob[k] = 0ull;
#endif
    if (on_node || on_quad) {
      const UnitPair* src = on_node ? (const UnitPair*)(wide + (size_t)node * PROG_WIDE_UNITS) : (const UnitPair*)(S.quads + qa_idx);
RT_UNROLL
      for (int k = 0; k < 8; ++k) { const UnitPair v = src[k]; blob[2 * k] = v.x; blob[2 * k + 1] = v.y; }
    } else if (on_leaf) {
      const uint64_t* src = prog + (i - PROG_BOX_UNITS);
RT_UNROLL
      for (int k = 0; k < 16; ++k) blob[k] = src[k];
      blob16 = src[16];
    }
    const long long wp_t1 = RT_WPROF_CLOCK(); (void)wp_t1;
    if (RT_WAVE_ANY(on_node)) {
      if (on_node) {
        const float FINF = (float)RT_INF;
        float e0 = FINF, e1 = FINF, e2 = FINF, e3 = FINF;
        // the ray in single precision (made here every step: a few instructions cost less than six registers), the
        // interval rounded outward
        const float ix = (float)lr.inv.x, iy = (float)lr.inv.y, iz = (float)lr.inv.z;
        const double sh = (double)shift, lo_t = tmin - sh, hi_t = tmax_hi - sh;
        const float ox = (float)(lr.o.x * lr.inv.x + sh), oy = (float)(lr.o.y * lr.inv.y + sh), oz = (float)(lr.o.z * lr.inv.z + sh);
        const float tlo = (float)(lo_t - fabs(lo_t) * 2e-7 - 1e-30), thi = (float)(hi_t + fabs(hi_t) * 2e-7 + 1e-30);
        const uint64_t d0 = blob[12], d1 = blob[13], d2 = blob[14], d3 = blob[15];
        // Straight-line code, selects instead of branches: the four tests, the sort, three stores to the stack of which
        // a later one overwrites an earlier one that was not wanted (the stack pointer only moves past wanted entries;
        // a level of the table owns three entries, so the stores stay inside the lane's stack), the next place.
        e0 = slab_entry_or_inf_f(unit_f32f(blob[0], 0), unit_f32f(blob[0], 1), unit_f32f(blob[1], 0), unit_f32f(blob[1], 1), unit_f32f(blob[2], 0), unit_f32f(blob[2], 1), ix, iy, iz, ox, oy, oz, tlo, thi);
        e1 = slab_entry_or_inf_f(unit_f32f(blob[3], 0), unit_f32f(blob[3], 1), unit_f32f(blob[4], 0), unit_f32f(blob[4], 1), unit_f32f(blob[5], 0), unit_f32f(blob[5], 1), ix, iy, iz, ox, oy, oz, tlo, thi);
        e2 = slab_entry_or_inf_f(unit_f32f(blob[6], 0), unit_f32f(blob[6], 1), unit_f32f(blob[7], 0), unit_f32f(blob[7], 1), unit_f32f(blob[8], 0), unit_f32f(blob[8], 1), ix, iy, iz, ox, oy, oz, tlo, thi);
        e3 = slab_entry_or_inf_f(unit_f32f(blob[9], 0), unit_f32f(blob[9], 1), unit_f32f(blob[10], 0), unit_f32f(blob[10], 1), unit_f32f(blob[11], 0), unit_f32f(blob[11], 1), ix, iy, iz, ox, oy, oz, tlo, thi);
        e0 = d0 != PROG_WIDE_EMPTY ? e0 : FINF; e1 = d1 != PROG_WIDE_EMPTY ? e1 : FINF;
        e2 = d2 != PROG_WIDE_EMPTY ? e2 : FINF; e3 = d3 != PROG_WIDE_EMPTY ? e3 : FINF;
        // an entry of the stack, or the next place, from a slot's descriptor (a record of the table, or 0x80000000 | the
        // unit position of a leaf)
        auto word_of = [&](uint64_t dsc) { return (dsc >> 32) == 0ull ? (uint32_t)dsc : (0x80000000u | (uint32_t)dsc); };
        uint32_t w0 = word_of(d0), w1 = word_of(d1), w2 = word_of(d2), w3 = word_of(d3);
        // the slots by entry, nearest first (equal entries in no particular order: the closest hit does not depend on
        // the order of the visits); a slot that failed sorts last
#define RT_WIDE_CE(ea, wa, eb, wb) { const bool sw_ = ea > eb; const float te_ = sw_ ? eb : ea, tf_ = sw_ ? ea : eb; const uint32_t tw_ = sw_ ? wb : wa, tx_ = sw_ ? wa : wb; ea = te_; eb = tf_; wa = tw_; wb = tx_; }
        RT_WIDE_CE(e0, w0, e1, w1) RT_WIDE_CE(e2, w2, e3, w3) RT_WIDE_CE(e0, w0, e2, w2) RT_WIDE_CE(e1, w1, e3, w3) RT_WIDE_CE(e1, w1, e2, w2)
#undef RT_WIDE_CE
        stk[sp] = w3; sp += e3 < FINF ? stk_stride : 0u;
        stk[sp] = w2; sp += e2 < FINF ? stk_stride : 0u;
        stk[sp] = w1; sp += e1 < FINF ? stk_stride : 0u;
        const bool go = e0 < FINF, to_leaf = (w0 & 0x80000000u) != 0u;
        node = go && !to_leaf ? w0 : node;
        i = go && to_leaf ? (w0 & 0x7FFFFFFFu) : i;
        st = go ? (to_leaf ? (uint32_t)ST_LEAF : (uint32_t)ST_NODE) : (uint32_t)ST_POP;
        RT_TRACE_STEP(RTX_NODE_BOX);
      }
    }
    uint32_t rest_p = 0, rest_end = 0;
    const long long wp_t2 = RT_WPROF_CLOCK(); (void)wp_t2;
    if (RT_WAVE_ANY(on_leaf)) {
      if (on_leaf) {
        double e;
        if (slab_hit_entry_v(unit_f64(blob[1]), unit_f64(blob[2]), unit_f64(blob[3]), unit_f64(blob[4]), unit_f64(blob[5]), unit_f64(blob[6]), lr, tmin, tmax_hi, e)) {
          leaf_entry = e; nq = 0;
          const uint32_t end = (uint32_t)(blob[0] >> 32);
          uint32_t p = i;
          // the first two records from the registers where they are single static spheres, a pair of them, or
          // general quads (units 7.. of the read are the leaf's)
          // Which records these are is decided first and the operands of at most two sphere tests are selected, so that
          // the wave runs the sphere test twice and not once per arrangement of the records (five of them).
          const uint64_t u0 = blob[7];
          const uint32_t k0 = RTX_NODE_KIND((uint32_t)u0), c0 = (uint32_t)(u0 >> 32), idx0 = RTX_NODE_IDX((uint32_t)u0);
          const bool has0 = p < end;
          const bool q0 = QUADS && has0 && k0 == RTX_NODE_QUAD;
          const bool s1 = has0 && k0 == PROG_NODE_SPHERE_S && c0 == 1u, s2 = has0 && k0 == PROG_NODE_SPHERE_S && c0 == 2u;
          const uint64_t u1 = q0 ? blob[8] : blob[12];                     // the record behind a quad / behind a single sphere
          const uint32_t p1 = p + (q0 ? 1u : 1u + PROG_SPHERE_S_UNITS);
          const uint32_t k1 = RTX_NODE_KIND((uint32_t)u1), c1 = (uint32_t)(u1 >> 32), idx1 = RTX_NODE_IDX((uint32_t)u1);
          const bool has1 = (q0 || s1) && p1 < end;
          const bool q1 = QUADS && has1 && k1 == RTX_NODE_QUAD, t1 = has1 && k1 == PROG_NODE_SPHERE_S && c1 == 1u;
          if (q0) note_quad(idx0, (p << 6) + 1u);
          if (q1) note_quad(idx1, (p1 << 6) + 1u);
          const bool first = s1 || s2;                                      // sphere A: the first record's (first) sphere, or the one behind a quad
          const bool va = first || (q0 && t1), vb = s2 || (s1 && t1);       // sphere B: the second of a pair, or the one behind a single sphere
          const uint64_t ax = first ? blob[8] : blob[9], ay = first ? blob[9] : blob[10], az = first ? blob[10] : blob[11], ar = first ? blob[11] : blob[12];
          const uint64_t bx = s2 ? blob[12] : blob[13], by = s2 ? blob[13] : blob[14], bz = s2 ? blob[14] : blob[15], br = s2 ? blob[15] : blob16;
          if (va) sphere_v(ax, ay, az, ar, first ? idx0 : idx1, first ? (p << 6) + 1u : (p1 << 6) + 1u);
          if (vb) sphere_v(bx, by, bz, br, s2 ? idx0 + 1u : idx1, s2 ? (p << 6) + 2u : (p1 << 6) + 1u);
          p = s2 ? p + 1u + 2u * PROG_SPHERE_S_UNITS : ((q0 || s1) ? p1 + (q1 ? 1u : (t1 ? 1u + PROG_SPHERE_S_UNITS : 0u)) : p);
          rest_p = p; rest_end = end;       // whatever else the leaf holds: below, when the registers of the read are free
          st = nq ? ST_QUAD : ST_POP;
          RT_TRACE_STEP(32u + 1u);
        } else st = ST_POP;
      }
    }
    const long long wp_t3 = RT_WPROF_CLOCK(); (void)wp_t3;
    if (QUADS && RT_WAVE_ANY(on_quad)) {
      if (on_quad) {   // rtx_quad: q, u, v, normal, w (three units each), d
        double t;
        const V3 qq = v3(unit_f64(blob[0]), unit_f64(blob[1]), unit_f64(blob[2])), qu = v3(unit_f64(blob[3]), unit_f64(blob[4]), unit_f64(blob[5]));
        const V3 qv = v3(unit_f64(blob[6]), unit_f64(blob[7]), unit_f64(blob[8])), qn = v3(unit_f64(blob[9]), unit_f64(blob[10]), unit_f64(blob[11]));
        const V3 qw = v3(unit_f64(blob[12]), unit_f64(blob[13]), unit_f64(blob[14]));
        if (quad_hit_v(qn, unit_f64(blob[15]), qq, qw, qu, qv, lr.o, lr.d, tmin, tmax, t) && wins(t, qa_pos, true)) take_hit(t, RTX_NODE_MAKE(RTX_NODE_QUAD, qa_idx), qa_pos, true);
        if (nq == 2u) { qa_idx = qb_idx; qa_pos = qb_pos; nq = 1u; } else { nq = 0u; st = ST_POP; }
        RT_TRACE_STEP(32u + 4u);
      }
    }
    const long long wp_t4 = RT_WPROF_CLOCK(); (void)wp_t4;
    if (RT_WAVE_ANY(rest_p < rest_end)) {   // records of a leaf that the read did not cover, or of another kind: from memory
      RT_WPROF(11, 1);
      if (rest_p < rest_end) records(rest_p, rest_end, false);
    }
    // (tuning build) cycles: issue of the reads, record step, leaf step, quad step, the rest; the whole trip is the sum
    // of these and of the top of the next trip (pops, refill)
    RT_WPROF(12, wp_t1 - wp_t0); RT_WPROF(13, wp_t2 - wp_t1); RT_WPROF(14, wp_t3 - wp_t2); RT_WPROF(15, wp_t4 - wp_t3); RT_WPROF(16, RT_WPROF_CLOCK() - wp_t4);
    RT_WPROF(17, RT_WPROF_CLOCK() - wp_t0);
  }
  RT_WPROF_FLUSH();
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
RT_HD void resolve_hit(const Scene& S, const Ray& wray, const Hit& h, bool want_uv, const double* stale_uv, Surface& sf) {
  const uint32_t kind = RTX_NODE_KIND(h.kind_idx), idx = RTX_NODE_IDX(h.kind_idx);
  V3 o = wray.o, d = wray.d;
  if (h.ctx != 0) ctx_to_local(S, h.ctx, o, d);
  sf.p = ray_at(o, d, h.t);
  sf.u = 0.0; sf.v = 0.0;
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
    if (want_uv) quad_uv(q, sf.p, sf.u, sf.v);
    sf.front = dot(d, n) < 0.0;
    sf.n = sf.front ? n : -n;
    sf.mat = q.mat; sf.prim_id = q.prim_id;
  } else {  // MEDIUM
    sf.n = v3(1.0, 0.0, 0.0);
    sf.front = true;
    sf.mat = S.media[idx].mat; sf.prim_id = 0xFFFFFFFEu;
    if (want_uv && stale_uv) {  // (u,v) of the surface hit the walk had found before this medium scattered, if any
      uint64_t id;
      memcpy(&id, &stale_uv[1], 8);
      const uint32_t k = RTX_NODE_KIND((uint32_t)id);
      if (k == RTX_NODE_SPHERE || k == RTX_NODE_QUAD) hit_uv(S, wray, (uint32_t)id, (uint32_t)(id >> 32), stale_uv[0], sf.u, sf.v);
    }
  }
  if (h.ctx != 0) ctx_to_world(S, h.ctx, sf.p, sf.n);
}

}  // namespace rt
