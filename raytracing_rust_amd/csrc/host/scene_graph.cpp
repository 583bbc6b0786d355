// Host scene layer: recording constructors mirroring the reference's scene
// API, bounding boxes and the median-split BVH build as the reference computes
// them, and the flattening of the recorded graph into an rtx_scene_desc.
// No GPU calls in this file.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/rtx_host.h"
#include "host_internal.h"

namespace rtxh {

static const double kInf = std::numeric_limits<double>::infinity();
static const double kPi = 3.14159265358979323846;

// ---- intervals and boxes exactly as the reference builds them (math/interval.rs, math/aabb.rs) ----
struct Interval {
  double min, max;
  Interval() : min(kInf), max(-kInf) {}
  Interval(double a, double b) : min(a), max(b) {}
  double size() const { return max - min; }
  Interval expand(double delta) const { double p = delta / 2.0; return Interval(min - p, max + p); }  // interval.rs:65-71
  Interval merge(const Interval& o) const { return Interval(fmin(min, o.min), fmax(max, o.max)); }   // interval.rs:81-86
};
struct Box {
  Interval x, y, z;
  static Box from_points(const double* a, const double* b) {  // Aabb::new_point, aabb.rs:22-42
    Box r;
    r.x = a[0] <= b[0] ? Interval(a[0], b[0]) : Interval(b[0], a[0]);
    r.y = a[1] <= b[1] ? Interval(a[1], b[1]) : Interval(b[1], a[1]);
    r.z = a[2] <= b[2] ? Interval(a[2], b[2]) : Interval(b[2], a[2]);
    const double DELTA = 0.0001;                              // pad_to_minimums, aabb.rs:92-103
    if (r.x.size() < DELTA) r.x = r.x.expand(DELTA);
    if (r.y.size() < DELTA) r.y = r.y.expand(DELTA);
    if (r.z.size() < DELTA) r.z = r.z.expand(DELTA);
    return r;
  }
  Box merge(const Box& o) const { Box r; r.x = x.merge(o.x); r.y = y.merge(o.y); r.z = z.merge(o.z); return r; }  // aabb.rs:113-119
  const Interval& axis(int a) const { return a == 0 ? x : (a == 1 ? y : z); }
  int longest_axis() const {                                  // aabb.rs:77-89
    double xs = x.size(), ys = y.size(), zs = z.size();
    if (xs >= ys && xs >= zs) return 0;
    if (ys >= zs) return 1;
    return 2;
  }
};

enum Kind { K_SPHERE, K_QUAD, K_LIST, K_BVH, K_TRANSLATE, K_ROTATE_Y, K_MEDIUM };

struct BvhTree {  // BvhNode, bvh.rs:12-16: each side is a leaf object or a subtree
  Box bbox;
  int left_obj = -1, right_obj = -1;
  std::unique_ptr<BvhTree> left, right;
};

struct Obj {
  Kind kind;
  bool has_box = false;
  Box box;
  uint32_t prim = 0;              // SPHERE / QUAD: index into the session's table
  std::vector<int> children;      // LIST
  std::unique_ptr<BvhTree> tree;  // BVH
  int child = -1;                 // TRANSLATE / ROTATE_Y / MEDIUM(boundary)
  double a = 0, b = 0, c = 0;     // TRANSLATE offset; ROTATE_Y sin, cos; MEDIUM neg_inv_density in a
  uint32_t mat = 0;               // MEDIUM phase function
};

}  // namespace rtxh

using namespace rtxh;

struct rtxh_session {
  std::vector<rtx_texture> textures;
  std::vector<rtx_image> images;
  std::vector<uint8_t> texels;
  std::vector<rtx_perlin> perlins;
  std::vector<rtx_material> materials;
  std::vector<rtx_sphere> spheres;
  std::vector<rtx_quad> quads;
  std::vector<Obj> objs;
  uint32_t next_prim = 0;
  std::string error;
  rtx_context* ctx = nullptr;  // set: rtxh_bvh sorts large lists on this context's GPU
  // flatten outputs
  std::vector<rtx_node> nodes;
  std::vector<rtx_aabb> boxes;
  std::vector<rtx_xform> xforms;
  std::vector<rtx_xform_op> ops;
  std::vector<rtx_medium> media;
  std::vector<rtx_light> lights;

  bool ok_tex(int i) const { return i >= 0 && (size_t)i < textures.size(); }
  bool ok_mat(int i) const { return i >= 0 && (size_t)i < materials.size(); }
  bool ok_obj(int i) const { return i >= 0 && (size_t)i < objs.size(); }
  int fail(const char* msg) { error = msg; return -1; }
};

namespace rtxh {

static bool tex_needs_uv(const rtxh_session* s, uint32_t tex, int depth = 0) {
  if (depth > RTX_MAX_TEXTURE_DEPTH) return false;
  const rtx_texture& t = s->textures[tex];
  if (t.kind == RTX_TEX_IMAGE) return true;
  if (t.kind == RTX_TEX_CHECKER) return tex_needs_uv(s, t.a, depth + 1) || tex_needs_uv(s, t.b, depth + 1);
  return false;
}

static int add_material(rtxh_session* s, uint32_t kind, int tex, double r, double g, double b, double param) {
  rtx_material m;
  memset(&m, 0, sizeof(m));
  m.kind = kind; m.tex = tex >= 0 ? (uint32_t)tex : 0;
  m.albedo[0] = r; m.albedo[1] = g; m.albedo[2] = b; m.param = param;
  m.needs_uv = (tex >= 0 && tex_needs_uv(s, (uint32_t)tex)) ? 1 : 0;
  s->materials.push_back(m);
  return (int)s->materials.size() - 1;
}

static inline void set3(double* d, double x, double y, double z) { d[0] = x; d[1] = y; d[2] = z; }
static inline double dot3(const double* a, const double* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static inline void cross3(const double* a, const double* b, double* o) {
  set3(o, a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}

// Quad::new, quad.rs:27-50
static int add_quad(rtxh_session* s, const double* q, const double* u, const double* v, int mat) {
  rtx_quad Q;
  memset(&Q, 0, sizeof(Q));
  memcpy(Q.q, q, 24); memcpy(Q.u, u, 24); memcpy(Q.v, v, 24);
  double n[3];
  cross3(u, v, n);
  double nn = dot3(n, n), len = sqrt(nn);
  set3(Q.normal, n[0] / len, n[1] / len, n[2] / len);
  Q.d = dot3(Q.normal, q);
  set3(Q.w, n[0] / nn, n[1] / nn, n[2] / nn);
  Q.area = len;
  Q.mat = (uint32_t)mat; Q.prim_id = s->next_prim++;
  double quv[3] = {q[0] + u[0] + v[0], q[1] + u[1] + v[1], q[2] + u[2] + v[2]};
  double qu[3] = {q[0] + u[0], q[1] + u[1], q[2] + u[2]}, qv[3] = {q[0] + v[0], q[1] + v[1], q[2] + v[2]};
  Obj o;
  o.kind = K_QUAD; o.has_box = true;
  o.box = Box::from_points(q, quv).merge(Box::from_points(qu, qv));
  o.prim = (uint32_t)s->quads.size();
  s->quads.push_back(Q);
  s->objs.push_back(std::move(o));
  return (int)s->objs.size() - 1;
}

static bool obj_box(const rtxh_session* s, int h, Box& b) {  // Hittable::bounding_box of each implementor
  const Obj& o = s->objs[(size_t)h];
  if (o.kind == K_LIST && o.children.empty()) return false;   // hittable_list.rs:70-76
  if (o.kind == K_MEDIUM) return obj_box(s, o.child, b);      // constant_medium.rs:93-95
  b = o.box;  // default-constructed = Aabb::empty() when no child contributed a box (translate.rs:20-24)
  return true;
}

// BvhNode::from_slice, bvh.rs:26-71
static std::unique_ptr<BvhTree> build_bvh(const rtxh_session* s, const std::vector<int>& objects, size_t start, size_t end) {
  std::unique_ptr<BvhTree> n(new BvhTree());
  for (size_t i = start; i < end; ++i) { Box b; if (obj_box(s, objects[i], b)) n->bbox = n->bbox.merge(b); }
  size_t span = end - start;
  int axis = n->bbox.longest_axis();
  auto box_min = [s, axis](int h) { Box b; if (!obj_box(s, h, b)) b = Box(); return b.axis(axis).min; };  // box_compare, 75-83
  if (span == 1) {
    n->left_obj = n->right_obj = objects[start];
  } else if (span == 2) {
    if (box_min(objects[start]) < box_min(objects[start + 1])) { n->left_obj = objects[start]; n->right_obj = objects[start + 1]; }
    else { n->left_obj = objects[start + 1]; n->right_obj = objects[start]; }
  } else {
    std::vector<int> sorted(objects.begin() + (long)start, objects.begin() + (long)end);
    std::stable_sort(sorted.begin(), sorted.end(), [&](int a, int b) { return box_min(a) < box_min(b); });
    size_t mid = span / 2;
    n->left = build_bvh(s, sorted, 0, mid);
    n->right = build_bvh(s, sorted, mid, sorted.size());
  }
  return n;
}

// The tree BvhNode::from_slice builds, from the leaf order it ends with (rtx_bvh_leaf_order): the shape follows from
// the span alone (bvh.rs:39-66), the sorts and the two-object comparison have been applied to `order` already.
static std::unique_ptr<BvhTree> build_bvh_ordered(const rtxh_session* s, const std::vector<int>& order, size_t start, size_t end) {
  std::unique_ptr<BvhTree> n(new BvhTree());
  const size_t span = end - start;
  if (span <= 2) {
    for (size_t i = start; i < end; ++i) { Box b; if (obj_box(s, order[i], b)) n->bbox = n->bbox.merge(b); }
    n->left_obj = order[start];
    n->right_obj = order[end - 1];
  } else {
    const size_t mid = start + span / 2;
    n->left = build_bvh_ordered(s, order, start, mid);
    n->right = build_bvh_ordered(s, order, mid, end);
    n->bbox = n->left->bbox.merge(n->right->bbox);   // min/max are exact: the merge of all objects' boxes in any order
  }
  return n;
}

// ------------------------------------------------------------- flattening ---
struct Flattener {
  rtxh_session* s;
  rtx_status status = RTX_OK;
  std::vector<std::vector<rtx_node>> side;  // medium boundary programs, appended after the world program
  void fail(rtx_status st, const char* msg) { if (status == RTX_OK) { status = st; s->error = msg; } }

  uint32_t push_ctx(uint32_t parent, const rtx_xform_op& op) {
    rtx_xform px = s->xforms[parent];
    rtx_xform nx;
    nx.first_op = (uint32_t)s->ops.size(); nx.n_ops = px.n_ops + 1;
    for (uint32_t k = 0; k < px.n_ops; ++k) { rtx_xform_op c = s->ops[px.first_op + k]; s->ops.push_back(c); }
    s->ops.push_back(op);
    s->xforms.push_back(nx);
    return (uint32_t)s->xforms.size() - 1;
  }
  static rtx_xform_op op_of(const Obj& o) {
    rtx_xform_op op;
    memset(&op, 0, sizeof(op));
    op.kind = o.kind == K_TRANSLATE ? RTX_XF_TRANSLATE : RTX_XF_ROTATE_Y;
    op.a = o.a; op.b = o.b; op.c = o.c;
    return op;
  }
  bool has_medium(int h, int depth = 0) const {
    if (depth > 64) return false;
    const Obj& o = s->objs[(size_t)h];
    switch (o.kind) {
      case K_MEDIUM: return true;
      case K_LIST: for (int c : o.children) if (has_medium(c, depth + 1)) return true; return false;
      case K_BVH: return tree_has_medium(o.tree.get(), depth + 1);
      case K_TRANSLATE: case K_ROTATE_Y: return has_medium(o.child, depth + 1);
      default: return false;
    }
  }
  bool tree_has_medium(const BvhTree* t, int depth) const {
    if (!t) return false;
    if (t->left ? tree_has_medium(t->left.get(), depth) : has_medium(t->left_obj, depth)) return true;
    if (t->left_obj >= 0 && t->left_obj == t->right_obj && !t->left) return false;
    return t->right ? tree_has_medium(t->right.get(), depth) : has_medium(t->right_obj, depth);
  }

  void emit_tree(const BvhTree* t, std::vector<rtx_node>& out, uint32_t ctx, bool in_boundary, int depth) {
    rtx_aabb b;
    b.lo[0] = t->bbox.x.min; b.hi[0] = t->bbox.x.max; b.lo[1] = t->bbox.y.min; b.hi[1] = t->bbox.y.max;
    b.lo[2] = t->bbox.z.min; b.hi[2] = t->bbox.z.max;
    size_t at = out.size();
    rtx_node nd;
    nd.kind_idx = RTX_NODE_MAKE(RTX_NODE_BOX, s->boxes.size()); nd.skip = 0;
    s->boxes.push_back(b);
    out.push_back(nd);
    if (!t->left && t->left_obj == t->right_obj) {
      // span 1 (bvh.rs:39-46): the reference tests the same object as left and as right.  For Sphere/Quad leaves
      // the second test cannot change the record (same t, strict/inclusive bounds against tmax = that t), so it
      // is emitted once; a subtree holding a medium draws a fresh free path each time, so it is emitted twice.
      emit(t->left_obj, out, ctx, in_boundary, depth + 1);
      if (has_medium(t->left_obj)) emit(t->left_obj, out, ctx, in_boundary, depth + 1);
    } else {
      if (t->left) emit_tree(t->left.get(), out, ctx, in_boundary, depth + 1); else emit(t->left_obj, out, ctx, in_boundary, depth + 1);
      if (t->right) emit_tree(t->right.get(), out, ctx, in_boundary, depth + 1); else emit(t->right_obj, out, ctx, in_boundary, depth + 1);
    }
    out[at].skip = (uint32_t)out.size();
  }

  void emit(int h, std::vector<rtx_node>& out, uint32_t ctx, bool in_boundary, int depth) {
    if (status != RTX_OK) return;
    if (depth > 64) { fail(RTX_ERR_INVALID, "scene graph deeper than 64 levels (cycle?)"); return; }
    const Obj& o = s->objs[(size_t)h];
    rtx_node nd;
    nd.skip = 0;
    switch (o.kind) {
      case K_SPHERE: nd.kind_idx = RTX_NODE_MAKE(RTX_NODE_SPHERE, o.prim); out.push_back(nd); break;
      case K_QUAD: nd.kind_idx = RTX_NODE_MAKE(RTX_NODE_QUAD, o.prim); out.push_back(nd); break;
      case K_LIST: for (int c : o.children) emit(c, out, ctx, in_boundary, depth + 1); break;  // hittable_list.rs:58-64
      case K_BVH: emit_tree(o.tree.get(), out, ctx, in_boundary, depth + 1); break;
      case K_TRANSLATE:
      case K_ROTATE_Y: {
        uint32_t inner = push_ctx(ctx, op_of(o));
        nd.kind_idx = RTX_NODE_MAKE(RTX_NODE_ENTER, inner); out.push_back(nd);
        emit(o.child, out, inner, in_boundary, depth + 1);
        nd.kind_idx = RTX_NODE_MAKE(RTX_NODE_EXIT, ctx); out.push_back(nd);
        break;
      }
      case K_MEDIUM: {
        if (in_boundary) { fail(RTX_ERR_UNSUPPORTED, "a ConstantMedium inside another medium's boundary is not supported"); return; }
        if (has_medium(o.child)) { fail(RTX_ERR_UNSUPPORTED, "a ConstantMedium inside another medium's boundary is not supported"); return; }
        rtx_medium m;
        memset(&m, 0, sizeof(m));
        m.mat = o.mat; m.neg_inv_density = o.a;
        m.needs_uv = s->materials[o.mat].needs_uv;
        // the boundary program is evaluated from the medium's own context; its ENTER/EXIT nodes are relative to it
        side.emplace_back();
        size_t slot = side.size() - 1;
        std::vector<rtx_node> prog;
        emit(o.child, prog, ctx, true, depth + 1);
        side[slot] = std::move(prog);
        m.b_begin = (uint32_t)slot;  // patched to node indices in finish()
        m.b_end = 0;
        nd.kind_idx = RTX_NODE_MAKE(RTX_NODE_MEDIUM, s->media.size());
        s->media.push_back(m);
        out.push_back(nd);
        break;
      }
    }
  }

  void finish(std::vector<rtx_node>& world) {
    s->nodes = std::move(world);
    for (rtx_medium& m : s->media) {
      std::vector<rtx_node>& prog = side[m.b_begin];
      uint32_t base = (uint32_t)s->nodes.size();
      for (rtx_node nd : prog) {
        if (RTX_NODE_KIND(nd.kind_idx) == RTX_NODE_BOX) nd.skip += base;
        s->nodes.push_back(nd);
      }
      m.b_begin = base; m.b_end = (uint32_t)s->nodes.size();
    }
  }

  // pdf_value / random of the lights graph as weighted leaves
  void light_leaf(uint32_t kind, uint32_t idx, uint32_t ctx, double w) {
    rtx_light L;
    memset(&L, 0, sizeof(L));
    L.kind = kind; L.idx = idx; L.ctx = ctx; L.weight = w;
    s->lights.push_back(L);
  }
  void emit_lights(int h, double w, uint32_t ctx, int depth) {
    if (status != RTX_OK) return;
    if (depth > 64) { fail(RTX_ERR_INVALID, "lights graph deeper than 64 levels (cycle?)"); return; }
    const Obj& o = s->objs[(size_t)h];
    switch (o.kind) {
      case K_SPHERE: light_leaf(RTX_NODE_SPHERE, o.prim, ctx, w); break;
      case K_QUAD: light_leaf(RTX_NODE_QUAD, o.prim, ctx, w); break;
      case K_LIST:                                             // hittable_list.rs:78-97
        if (o.children.empty()) { light_leaf(RTX_NODE_NONE, 0, ctx, w); break; }
        for (int c : o.children) emit_lights(c, w * (1.0 / (double)o.children.size()), ctx, depth + 1);
        break;
      case K_BVH: emit_light_tree(o.tree.get(), w, ctx, depth + 1); break;
      case K_TRANSLATE: case K_ROTATE_Y: emit_lights(o.child, w, push_ctx(ctx, op_of(o)), depth + 1); break;
      case K_MEDIUM: light_leaf(RTX_NODE_NONE, 0, ctx, w); break;  // trait defaults, hittable.rs:107-114
    }
  }
  void emit_light_tree(const BvhTree* t, double w, uint32_t ctx, int depth) {  // bvh.rs:113-126
    if (t->left) emit_light_tree(t->left.get(), 0.5 * w, ctx, depth + 1); else emit_lights(t->left_obj, 0.5 * w, ctx, depth + 1);
    if (t->right) emit_light_tree(t->right.get(), 0.5 * w, ctx, depth + 1); else emit_lights(t->right_obj, 0.5 * w, ctx, depth + 1);
  }
};

bool load_ppm(const char* path, std::vector<uint8_t>& rgb, uint32_t& w, uint32_t& h) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  char magic[3] = {0, 0, 0};
  int maxv = 0;
  unsigned ww = 0, hh = 0;
  bool ok = fscanf(f, "%2s %u %u %d", magic, &ww, &hh, &maxv) == 4 && strcmp(magic, "P6") == 0 && maxv == 255 && ww > 0 && hh > 0;
  if (ok) {
    fgetc(f);
    rgb.resize((size_t)ww * hh * 3);
    ok = fread(rgb.data(), 1, rgb.size(), f) == rgb.size();
  }
  fclose(f);
  if (ok) { w = ww; h = hh; }
  return ok;
}

double HostRng::next_double() {
  uint64_t z = (state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
double HostRng::range(double mn, double mx) { return mn + (mx - mn) * next_double(); }
int HostRng::int_range(int mn, int mx) {
  int v = mn + (int)floor(next_double() * (double)(mx - mn + 1));
  return v > mx ? mx : v;
}

// Perlin::new, procedural/noise.rs:16-31,80-93, drawing from a seeded stream.
void perlin_from_rng(HostRng& rng, rtx_perlin& P) {
  for (int i = 0; i < 256; ++i) {
    double a = rng.range(-1.0, 1.0), b = rng.range(-1.0, 1.0), c = rng.range(-1.0, 1.0);
    double len = sqrt((a * a + b * b) + c * c);
    P.ranvec[3 * i] = a / len; P.ranvec[3 * i + 1] = b / len; P.ranvec[3 * i + 2] = c / len;
  }
  int32_t* perms[3] = {P.perm_x, P.perm_y, P.perm_z};
  for (int k = 0; k < 3; ++k) {
    int32_t* p = perms[k];
    for (int i = 0; i < 256; ++i) p[i] = i;
    for (int i = 255; i >= 1; --i) { int target = rng.int_range(0, i); std::swap(p[i], p[target]); }
  }
}

int add_noise_texture(rtxh_session* s, double scale, const rtx_perlin& P) {
  rtx_texture t;
  memset(&t, 0, sizeof(t));
  t.kind = RTX_TEX_NOISE; t.a = (uint32_t)s->perlins.size(); t.scale = scale;
  s->perlins.push_back(P);
  s->textures.push_back(t);
  return (int)s->textures.size() - 1;
}

}  // namespace rtxh

extern "C" {

rtxh_session* rtxh_session_new(void) { return new rtxh_session(); }
void rtxh_session_free(rtxh_session* s) { delete s; }
const char* rtxh_last_error(const rtxh_session* s) { return s ? s->error.c_str() : "null session"; }

int rtxh_tex_solid(rtxh_session* s, double r, double g, double b) {
  rtx_texture t;
  memset(&t, 0, sizeof(t));
  t.kind = RTX_TEX_SOLID; set3(t.color, r, g, b);
  s->textures.push_back(t);
  return (int)s->textures.size() - 1;
}
int rtxh_tex_checker(rtxh_session* s, double scale, int even, int odd) {
  if (!s->ok_tex(even) || !s->ok_tex(odd)) return s->fail("tex_checker: bad texture handle");
  rtx_texture t;
  memset(&t, 0, sizeof(t));
  t.kind = RTX_TEX_CHECKER; t.a = (uint32_t)even; t.b = (uint32_t)odd; t.scale = 1.0 / scale;  // checker.rs:18
  s->textures.push_back(t);
  return (int)s->textures.size() - 1;
}
int rtxh_tex_noise(rtxh_session* s, double scale, uint64_t seed) {
  rtx_perlin P;
  HostRng rng(seed);
  perlin_from_rng(rng, P);
  return add_noise_texture(s, scale, P);
}
int rtxh_tex_noise_tables(rtxh_session* s, double scale, const double* ranvec, const int32_t* px, const int32_t* py, const int32_t* pz) {
  if (!ranvec || !px || !py || !pz) return s->fail("tex_noise_tables: null table");
  rtx_perlin P;
  memcpy(P.ranvec, ranvec, sizeof(P.ranvec));
  memcpy(P.perm_x, px, sizeof(P.perm_x)); memcpy(P.perm_y, py, sizeof(P.perm_y)); memcpy(P.perm_z, pz, sizeof(P.perm_z));
  for (int i = 0; i < 256; ++i)
    if ((unsigned)P.perm_x[i] > 255u || (unsigned)P.perm_y[i] > 255u || (unsigned)P.perm_z[i] > 255u)
      return s->fail("tex_noise_tables: permutation entry outside 0..255");
  return add_noise_texture(s, scale, P);
}
int rtxh_tex_image(rtxh_session* s, uint32_t width, uint32_t height, const uint8_t* rgb) {
  rtx_image im;
  memset(&im, 0, sizeof(im));
  if (width > 0 && height > 0 && rgb) {
    im.width = width; im.height = height; im.offset = s->texels.size();
    s->texels.insert(s->texels.end(), rgb, rgb + (size_t)width * height * 3);
  }
  rtx_texture t;
  memset(&t, 0, sizeof(t));
  t.kind = RTX_TEX_IMAGE; t.a = (uint32_t)s->images.size();
  s->images.push_back(im);
  s->textures.push_back(t);
  return (int)s->textures.size() - 1;
}
int rtxh_tex_image_file(rtxh_session* s, const char* filename) {  // image.rs:15-43
  std::string fn(filename ? filename : "");
  std::vector<std::string> paths = {fn, "textures/" + fn, "../textures/" + fn};
  if (const char* env = getenv("RTW_IMAGES")) paths.push_back(std::string(env) + "/" + fn);
  std::vector<uint8_t> rgb;
  uint32_t w = 0, h = 0;
  for (const auto& p : paths)
    if (load_ppm(p.c_str(), rgb, w, h)) return rtxh_tex_image(s, w, h, rgb.data());
  fprintf(stderr, "ERROR: Could not load image file '%s'.\n", fn.c_str());  // image.rs:37
  return rtxh_tex_image(s, 0, 0, nullptr);
}

int rtxh_mat_lambertian(rtxh_session* s, int tex) { if (!s->ok_tex(tex)) return s->fail("mat_lambertian: bad texture handle"); return add_material(s, RTX_MAT_LAMBERTIAN, tex, 0, 0, 0, 0); }
int rtxh_mat_metal(rtxh_session* s, double r, double g, double b, double fuzz) {
  double f = fuzz < 0.0 ? 0.0 : (fuzz > 1.0 ? 1.0 : fuzz);  // metal.rs:19
  return add_material(s, RTX_MAT_METAL, -1, r, g, b, f);
}
int rtxh_mat_dielectric(rtxh_session* s, double ri) { return add_material(s, RTX_MAT_DIELECTRIC, -1, 0, 0, 0, ri); }
int rtxh_mat_diffuse_light(rtxh_session* s, int tex) { if (!s->ok_tex(tex)) return s->fail("mat_diffuse_light: bad texture handle"); return add_material(s, RTX_MAT_DIFFUSE_LIGHT, tex, 0, 0, 0, 0); }
int rtxh_mat_isotropic(rtxh_session* s, int tex) { if (!s->ok_tex(tex)) return s->fail("mat_isotropic: bad texture handle"); return add_material(s, RTX_MAT_ISOTROPIC, tex, 0, 0, 0, 0); }
int rtxh_mat_none(rtxh_session* s) { return add_material(s, RTX_MAT_NONE, -1, 0, 0, 0, 0); }

static int add_sphere(rtxh_session* s, const double* c0, const double* c1, double radius, int mat, bool moving) {
  if (!s->ok_mat(mat)) return s->fail("sphere: bad material handle");
  rtx_sphere S;
  memset(&S, 0, sizeof(S));
  memcpy(S.c0, c0, 24);
  if (moving) set3(S.dc, c1[0] - c0[0], c1[1] - c0[1], c1[2] - c0[2]);  // sphere.rs:48
  S.radius = radius; S.mat = (uint32_t)mat; S.prim_id = s->next_prim++;
  Obj o;
  o.kind = K_SPHERE; o.has_box = true;
  double lo[3] = {c0[0] - radius, c0[1] - radius, c0[2] - radius}, hi[3] = {c0[0] + radius, c0[1] + radius, c0[2] + radius};
  o.box = Box::from_points(lo, hi);                                     // sphere.rs:23-24
  if (moving) {                                                         // sphere.rs:42-45
    double lo1[3] = {c1[0] - radius, c1[1] - radius, c1[2] - radius}, hi1[3] = {c1[0] + radius, c1[1] + radius, c1[2] + radius};
    o.box = o.box.merge(Box::from_points(lo1, hi1));
  }
  o.prim = (uint32_t)s->spheres.size();
  s->spheres.push_back(S);
  s->objs.push_back(std::move(o));
  return (int)s->objs.size() - 1;
}
int rtxh_sphere(rtxh_session* s, const double* c, double radius, int mat) { return add_sphere(s, c, c, radius, mat, false); }
int rtxh_sphere_moving(rtxh_session* s, const double* c0, const double* c1, double radius, int mat) { return add_sphere(s, c0, c1, radius, mat, true); }
int rtxh_quad(rtxh_session* s, const double* q, const double* u, const double* v, int mat) {
  if (!s->ok_mat(mat)) return s->fail("quad: bad material handle");
  return add_quad(s, q, u, v, mat);
}
int rtxh_list_new(rtxh_session* s) {
  Obj o;
  o.kind = K_LIST;
  s->objs.push_back(std::move(o));
  return (int)s->objs.size() - 1;
}
int rtxh_list_add(rtxh_session* s, int list, int object) {  // hittable_list.rs:27-37
  if (!s->ok_obj(list) || !s->ok_obj(object) || s->objs[(size_t)list].kind != K_LIST) return s->fail("list_add: bad handle");
  if (list == object) return s->fail("list_add: a list cannot contain itself");
  Box b;
  if (obj_box(s, object, b)) {
    Obj& l = s->objs[(size_t)list];
    l.box = l.children.empty() ? b : l.box.merge(b);
    l.has_box = true;
  }
  s->objs[(size_t)list].children.push_back(object);
  return 0;
}
int rtxh_box(rtxh_session* s, const double* a, const double* b, int mat) {  // box_new, quad.rs:125-174
  if (!s->ok_mat(mat)) return s->fail("box: bad material handle");
  int list = rtxh_list_new(s);
  double mn[3] = {fmin(a[0], b[0]), fmin(a[1], b[1]), fmin(a[2], b[2])};
  double mx[3] = {fmax(a[0], b[0]), fmax(a[1], b[1]), fmax(a[2], b[2])};
  double dx[3] = {mx[0] - mn[0], 0, 0}, dy[3] = {0, mx[1] - mn[1], 0}, dz[3] = {0, 0, mx[2] - mn[2]};
  double ndx[3] = {-dx[0], -dx[1], -dx[2]}, ndz[3] = {-dz[0], -dz[1], -dz[2]};
  double q0[3] = {mn[0], mn[1], mx[2]}, q1[3] = {mx[0], mn[1], mx[2]}, q2[3] = {mx[0], mn[1], mn[2]};
  double q3[3] = {mn[0], mn[1], mn[2]}, q4[3] = {mn[0], mx[1], mx[2]}, q5[3] = {mn[0], mn[1], mn[2]};
  rtxh_list_add(s, list, add_quad(s, q0, dx, dy, mat));    // front
  rtxh_list_add(s, list, add_quad(s, q1, ndz, dy, mat));   // right
  rtxh_list_add(s, list, add_quad(s, q2, ndx, dy, mat));   // back
  rtxh_list_add(s, list, add_quad(s, q3, dz, dy, mat));    // left
  rtxh_list_add(s, list, add_quad(s, q4, dx, ndz, mat));   // top
  rtxh_list_add(s, list, add_quad(s, q5, dx, dz, mat));    // bottom
  return list;
}
void rtxh_session_set_context(rtxh_session* s, rtx_context* ctx) { if (s) s->ctx = ctx; }
int rtxh_bvh(rtxh_session* s, int list) {
  if (!s->ok_obj(list) || s->objs[(size_t)list].kind != K_LIST) return s->fail("bvh: not a list handle");
  if (s->objs[(size_t)list].children.empty()) return s->fail("bvh: empty list");  // the reference would index out of bounds
  Obj o;
  o.kind = K_BVH;
  const std::vector<int>& children = s->objs[(size_t)list].children;
  if (s->ctx && children.size() >= (size_t)RTXH_GPU_BVH_MIN) {
    std::vector<double> boxes(children.size() * 6);
    for (size_t i = 0; i < children.size(); ++i) {
      Box b;
      if (!obj_box(s, children[i], b)) b = Box();
      double* o6 = &boxes[6 * i];
      o6[0] = b.x.min; o6[1] = b.y.min; o6[2] = b.z.min; o6[3] = b.x.max; o6[4] = b.y.max; o6[5] = b.z.max;
    }
    std::vector<uint32_t> perm(children.size());
    if (rtx_bvh_leaf_order(s->ctx, (uint32_t)children.size(), boxes.data(), perm.data()) != RTX_OK) {
      s->error = std::string("bvh: GPU build failed: ") + rtx_last_error(s->ctx);
      return -1;
    }
    std::vector<int> order(children.size());
    for (size_t i = 0; i < children.size(); ++i) order[i] = children[perm[i]];
    o.tree = build_bvh_ordered(s, order, 0, order.size());
  } else {
    o.tree = build_bvh(s, children, 0, children.size());
  }
  o.has_box = true; o.box = o.tree->bbox;
  s->objs.push_back(std::move(o));
  return (int)s->objs.size() - 1;
}
int rtxh_translate(rtxh_session* s, int object, const double* off) {  // translate.rs:18-31
  if (!s->ok_obj(object)) return s->fail("translate: bad handle");
  Obj o;
  o.kind = K_TRANSLATE; o.child = object; o.a = off[0]; o.b = off[1]; o.c = off[2];
  Box b;
  o.has_box = true;
  if (obj_box(s, object, b)) {                                       // aabb.rs:177-188
    o.box.x = Interval(b.x.min + off[0], b.x.max + off[0]);
    o.box.y = Interval(b.y.min + off[1], b.y.max + off[1]);
    o.box.z = Interval(b.z.min + off[2], b.z.max + off[2]);
  }
  s->objs.push_back(std::move(o));
  return (int)s->objs.size() - 1;
}
int rtxh_rotate_y(rtxh_session* s, int object, double degrees) {    // rotate_y.rs:20-67
  if (!s->ok_obj(object)) return s->fail("rotate_y: bad handle");
  Obj o;
  o.kind = K_ROTATE_Y; o.child = object;
  double radians = degrees * kPi / 180.0;
  double sin_theta = sin(radians), cos_theta = cos(radians);
  o.a = sin_theta; o.b = cos_theta;
  Box ob;
  o.has_box = true;
  if (obj_box(s, object, ob)) {
    double mn[3] = {kInf, kInf, kInf}, mx[3] = {-kInf, -kInf, -kInf};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 2; ++k) {
      double x = (double)i * ob.x.max + (double)(1 - i) * ob.x.min;
      double y = (double)j * ob.y.max + (double)(1 - j) * ob.y.min;
      double z = (double)k * ob.z.max + (double)(1 - k) * ob.z.min;
      double newx = cos_theta * x + sin_theta * z;
      double newz = -sin_theta * x + cos_theta * z;
      mn[0] = fmin(mn[0], newx); mn[1] = fmin(mn[1], y); mn[2] = fmin(mn[2], newz);
      mx[0] = fmax(mx[0], newx); mx[1] = fmax(mx[1], y); mx[2] = fmax(mx[2], newz);
    }
    o.box = Box::from_points(mn, mx);
  }
  s->objs.push_back(std::move(o));
  return (int)s->objs.size() - 1;
}
int rtxh_medium(rtxh_session* s, int boundary, double density, int tex) {  // constant_medium.rs:22-38
  if (!s->ok_obj(boundary) || !s->ok_tex(tex)) return s->fail("medium: bad handle");
  Obj o;
  o.kind = K_MEDIUM; o.child = boundary; o.a = -1.0 / density;
  o.mat = (uint32_t)add_material(s, RTX_MAT_ISOTROPIC, tex, 0, 0, 0, 0);
  s->objs.push_back(std::move(o));
  return (int)s->objs.size() - 1;
}

int rtxh_bounding_box(rtxh_session* s, int object, double* out) {
  if (!s->ok_obj(object)) return s->fail("bounding_box: bad handle");
  Box b;
  if (!obj_box(s, object, b)) return 0;
  out[0] = b.x.min; out[1] = b.x.max; out[2] = b.y.min; out[3] = b.y.max; out[4] = b.z.min; out[5] = b.z.max;
  return 1;
}

rtx_status rtxh_flatten(rtxh_session* s, int world, int lights, rtx_scene_desc* out) {
  if (!s || !out) return RTX_ERR_INVALID;
  s->error.clear();   // the message, if any, is this call's
  if (!s->ok_obj(world) || (lights >= 0 && !s->ok_obj(lights))) { s->error = "flatten: bad handle"; return RTX_ERR_INVALID; }
  s->nodes.clear(); s->boxes.clear(); s->xforms.clear(); s->ops.clear(); s->media.clear(); s->lights.clear();
  rtx_xform root;
  root.first_op = 0; root.n_ops = 0;
  s->xforms.push_back(root);
  Flattener f;
  f.s = s;
  std::vector<rtx_node> prog;
  f.emit(world, prog, 0, false, 0);
  if (f.status != RTX_OK) return f.status;
  uint32_t world_end = (uint32_t)prog.size();
  f.finish(prog);
  if (lights >= 0) {
    f.emit_lights(lights, 1.0, 0, 0);
    if (f.status != RTX_OK) return f.status;
    // A lone empty list is Some(empty): pdf 0, direction (1,0,0) — the same as one NONE leaf of weight 1.
    double cum = 0.0;
    for (rtx_light& L : s->lights) { cum += L.weight; L.cum = cum; }
  }
  if (s->nodes.size() >= (1u << 28) || s->boxes.size() >= (1u << 28)) { s->error = "flatten: scene too large for 28-bit node indices"; return RTX_ERR_UNSUPPORTED; }
  memset(out, 0, sizeof(*out));
  out->n_nodes = (uint32_t)s->nodes.size(); out->nodes = s->nodes.data();
  out->world_begin = 0; out->world_end = world_end;
  out->n_boxes = (uint32_t)s->boxes.size(); out->boxes = s->boxes.data();
  out->n_spheres = (uint32_t)s->spheres.size(); out->spheres = s->spheres.data();
  out->n_quads = (uint32_t)s->quads.size(); out->quads = s->quads.data();
  out->n_xforms = (uint32_t)s->xforms.size(); out->xforms = s->xforms.data();
  out->n_xform_ops = (uint32_t)s->ops.size(); out->xform_ops = s->ops.data();
  out->n_media = (uint32_t)s->media.size(); out->media = s->media.data();
  out->n_materials = (uint32_t)s->materials.size(); out->materials = s->materials.data();
  out->n_textures = (uint32_t)s->textures.size(); out->textures = s->textures.data();
  out->n_images = (uint32_t)s->images.size(); out->images = s->images.data();
  out->n_texel_bytes = s->texels.size(); out->texels = s->texels.data();
  out->n_perlins = (uint32_t)s->perlins.size(); out->perlins = s->perlins.data();
  out->n_lights = (uint32_t)s->lights.size(); out->lights = s->lights.data();
  out->has_lights = lights >= 0 ? 1 : 0;
  return RTX_OK;
}

}  // extern "C"
