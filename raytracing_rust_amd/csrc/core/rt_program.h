// Compilation of the boundary's hit program (rtx_scene_desc.nodes) into the image the kernels walk.
//
// The ABI keeps nodes, boxes, primitives, transforms and media in separate tables.  For the device the program
// becomes ONE array of 8-byte units in which every record carries the operands its step needs, so that a walk reads a
// single address space (the whole image is staged in LDS when it fits) and never gathers from the tables:
//
//   unit = kind_idx (low 32 bits, as RTX_NODE_MAKE) | aux << 32, followed by the record's payload units
//   BOX       aux = unit index to continue at on a miss        payload lo[3], hi[3]                       (7 units)
//   SPHERE_S  idx = first sphere, aux = count                  payload count x {cx, cy, cz, r}
//   QUAD_AA   idx = first quad, aux = count                    payload count x DAAQuad (10 units)
//   BOX6      idx = first of the six quads                     payload DBox6 (9 doubles)
//   ENTER/EXIT idx = transform context, aux = op count         payload ops x {kind, a, b, c}
//   MEDIUM    idx = medium, aux = first unit of its boundary   payload {neg_inv_density, flags}
//   MEDIUM_S  idx = medium whose boundary is one static sphere payload {neg_inv_density, flags, DSphere}
//   SPHERE    idx = sphere (moving spheres)                    payload {c0[3], dc[3], radius}             (8 units)
//   QUAD      (general quads): no payload, the walker reads the ABI table
//   SYNC, NOP, END: no payload; an END closes the world program and every medium boundary program.
//
// Visiting order and skip links are those of the ABI program.  Static spheres and quads whose normal is an exact
// coordinate axis get the compact records above; six consecutive faces that box_new built become one BOX6.
#pragma once
#include <math.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <vector>

#include "rt_traverse.h"

namespace rt {

struct CompiledProgram {
  std::vector<uint64_t> units;
  std::vector<rtx_medium> media;   // copy of the scene's media with b_begin as a unit index (b_end is unused)
  std::vector<DAAQuad> daaquads;   // parallel to d.quads; valid where quad_aa[] is set
  std::vector<uint8_t> sphere_static, quad_aa;
  uint32_t world_begin = 0;
  uint32_t n_box6 = 0;
  uint32_t head_end = 0;           // SYNC unit behind the world's first top-level item when that is a large box subtree
  bool split_head = true;          // look for such a head section (tests switch it off to compare)
  uint32_t n_general_media = 0;    // MEDIUM units emitted (media whose boundary needs a program of its own)
  bool guard_box6 = true;          // a widened box test in front of every BOX6 (tests switch it off to compare)
  bool fuse_sphere_media = true;   // emit MEDIUM_S for sphere-bounded media (tests switch it off to compare)
  // MEDIUM_S of a thin medium draws its free path before it intersects the boundary and leaves early when the path is
  // longer than the ray can be inside (exact: see walk_program).  Off by default: every walker here steps the 64 rays of
  // a wave through a medium together, some lane always needs the boundary, and the wave then pays for both (final
  // scene: 2268 against 2269 Msamples/s, 2 % more VALU instructions in the world-program kernel).  Tests switch it on.
  bool draw_first_thin_media = false;
  bool defer_last_section = true;  // split the last top-level box subtree off into a program of its own (same)
  uint32_t tail_begin = 0;         // first unit of that program, 0 = none
  uint32_t tail_near_depth = 0;    // > 0: that program is a binary tree of boxes of this depth (walk_stream_near's shape)
  uint32_t head_near_depth = 0;    // > 0: so is the head section under its root box
  uint32_t tail_near_quads = 0;    // the deferred section's leaves hold quads (it needs the walk's quad classes)
  uint32_t defer_pos = 0;          // unit index of the DEFER unit
  uint32_t prog_units = 0;         // units of the program itself; tables for particular walks follow in `units`
  uint32_t head_top_begin = 0, head_top_nodes = 0, head_top_units = 0;   // Scene::head_top_*: the first levels of the head section's box tree
  uint32_t head_top_max_levels = 10;                 // 1023 records = 57 KB of LDS next to the walk's stacks
  double head_wide_far = 0.0;
  uint32_t head_wide_begin = 0, head_wide_nodes = 0, head_wide_depth = 0; // Scene::head_wide_*: the head section's box tree, two levels per record
  bool head_wide = true;                             // build that table (tests switch it off to compare)

  // The light table (Scene::light_nodes): per group of 64 consecutive light leaves a median-split tree over their
  // boxes.  A leaf's box bounds its primitive in WORLD space - a quad's four corners, a static sphere's centre +- radius
  // (Sphere::pdf_value probes the sphere of time 0, sphere.rs:135), carried out of the leaf's transform context corner
  // by corner - widened by 1e-6 of (1 + its largest coordinate): the rounding of the reference's hit tests and of the
  // slab test is many orders of magnitude below that for probe origins within light_far, so a ray that misses the
  // widened box cannot be accepted by Quad::hit / Sphere::hit.  Lights of kind NONE have no box (their pdf is 0).
  std::vector<LightNode> light_nodes;
  std::vector<uint32_t> light_group_begin;
  double light_far = 0.0;
  bool light_table = true;         // build it (tests switch it off to compare)
  void build_light_table(const rtx_scene_desc& d) {
    light_nodes.clear(); light_group_begin.clear(); light_far = 0.0;
    if (!light_table || d.n_lights < LIGHT_TABLE_MIN) return;
    struct Leaf { double lo[3], hi[3]; uint32_t light; };
    double extent = 0.0;
    auto leaf_of = [&](uint32_t i, Leaf& lf) -> bool {
      const rtx_light& L = d.lights[i];
      if (L.kind != RTX_NODE_QUAD && L.kind != RTX_NODE_SPHERE) return false;
      double c[8][3];
      int nc = 0;
      if (L.kind == RTX_NODE_QUAD) {
        const rtx_quad& q = d.quads[L.idx];
        for (int k = 0; k < 4; ++k, ++nc)
          for (int a = 0; a < 3; ++a) c[nc][a] = q.q[a] + ((k & 1) ? q.u[a] : 0.0) + ((k & 2) ? q.v[a] : 0.0);
      } else {
        const rtx_sphere& sp = d.spheres[L.idx];
        const double r = fabs(sp.radius);
        for (int k = 0; k < 8; ++k, ++nc)
          for (int a = 0; a < 3; ++a) c[nc][a] = sp.c0[a] + (((k >> a) & 1) ? r : -r);
      }
      for (int a = 0; a < 3; ++a) { lf.lo[a] = INFINITY; lf.hi[a] = -INFINITY; }
      for (int k = 0; k < nc; ++k) {
        double p[3] = {c[k][0], c[k][1], c[k][2]};
        const rtx_xform x = d.xforms[L.ctx];            // context -> world: innermost first (ctx_to_world)
        for (uint32_t s = x.n_ops; s > 0; --s) {
          const rtx_xform_op op = d.xform_ops[x.first_op + s - 1];
          if (op.kind == RTX_XF_TRANSLATE) { p[0] += op.a; p[1] += op.b; p[2] += op.c; }
          else { const double px = op.b * p[0] + op.a * p[2], pz = -op.a * p[0] + op.b * p[2]; p[0] = px; p[2] = pz; }
        }
        for (int a = 0; a < 3; ++a) { lf.lo[a] = std::min(lf.lo[a], p[a]); lf.hi[a] = std::max(lf.hi[a], p[a]); }
      }
      for (int a = 0; a < 3; ++a) {
        if (!(fabs(lf.lo[a]) < 1e300) || !(fabs(lf.hi[a]) < 1e300)) return false;
        const double pad = 1e-6 * (1.0 + std::max(fabs(lf.lo[a]), fabs(lf.hi[a])));
        lf.lo[a] -= pad; lf.hi[a] += pad;
        extent = std::max(extent, std::max(fabs(lf.lo[a]), fabs(lf.hi[a])));
      }
      lf.light = i;
      return true;
    };
    std::function<void(std::vector<Leaf>&, size_t, size_t)> emit = [&](std::vector<Leaf>& v, size_t b, size_t e) {
      LightNode nd;
      for (int a = 0; a < 3; ++a) { nd.lo[a] = INFINITY; nd.hi[a] = -INFINITY; }
      for (size_t k = b; k < e; ++k)
        for (int a = 0; a < 3; ++a) { nd.lo[a] = std::min(nd.lo[a], v[k].lo[a]); nd.hi[a] = std::max(nd.hi[a], v[k].hi[a]); }
      const size_t at = light_nodes.size();
      nd.skip = 0; nd.light = e - b == 1 ? v[b].light : LIGHT_NODE_INNER;
      light_nodes.push_back(nd);
      if (e - b > 1) {
        int axis = 0;
        for (int a = 1; a < 3; ++a) if (nd.hi[a] - nd.lo[a] > nd.hi[axis] - nd.lo[axis]) axis = a;
        std::sort(v.begin() + (long)b, v.begin() + (long)e, [axis](const Leaf& x, const Leaf& y) { return x.lo[axis] + x.hi[axis] < y.lo[axis] + y.hi[axis]; });
        const size_t mid = b + (e - b) / 2;
        emit(v, b, mid); emit(v, mid, e);
      }
      light_nodes[at].skip = (uint32_t)light_nodes.size();
    };
    bool whole = true;   // a light whose box cannot be made (a non-finite record) must still be evaluated: no table then
    for (uint32_t g0 = 0; g0 < d.n_lights && whole; g0 += 64u) {
      light_group_begin.push_back((uint32_t)light_nodes.size());
      std::vector<Leaf> leaves;
      for (uint32_t i = g0; i < std::min(d.n_lights, g0 + 64u); ++i) {
        Leaf lf;
        if (leaf_of(i, lf)) leaves.push_back(lf);
        else if (d.lights[i].kind != RTX_NODE_NONE) whole = false;
      }
      if (!leaves.empty()) emit(leaves, 0, leaves.size());
    }
    light_group_begin.push_back((uint32_t)light_nodes.size());
    if (!whole || light_nodes.empty()) { light_nodes.clear(); light_group_begin.clear(); return; }
    light_far = 1e6 * (1.0 + extent);
  }

  // `d` must already be validated (index ranges, forward skip links that stay inside their program, no medium inside
  // a medium boundary).
  void compile(const rtx_scene_desc& d) {
    build_light_table(d);
    units.clear();
    n_box6 = 0;
    build_compact(d);
    media.assign(d.media, d.media + d.n_media);
    for (uint32_t m = 0; m < d.n_media; ++m) {  // boundaries first: MEDIUM units of the world point at them
      media[m].b_begin = emit_range(d, d.media[m].b_begin, d.media[m].b_end, false);
      media[m].b_end = 0;
    }
    tail_begin = 0;
    uint32_t root = 0, ctx = 0;
    bool has_xf = false;
    if (defer_last_section && find_deferrable(d, root, has_xf, ctx)) {
      // world program with a DEFER node in place of the subtree under `root`
      const uint32_t skip = d.nodes[root].skip;
      // (appended behind a copy of all nodes, so that every other index of the description stays valid)
      std::vector<rtx_node> nodes(d.nodes, d.nodes + d.n_nodes);
      const uint32_t mb = d.n_nodes, shift = mb - d.world_begin;
      for (uint32_t k = d.world_begin; k <= root; ++k) {
        rtx_node nd = d.nodes[k];
        if (RTX_NODE_KIND(nd.kind_idx) == RTX_NODE_BOX) nd.skip += shift;
        nodes.push_back(nd);
      }
      nodes.back().skip = (uint32_t)nodes.size() + 1u;                    // the root box skips the DEFER node
      nodes.push_back(rtx_node{RTX_NODE_MAKE(PROG_NODE_DEFER, 0), 0});
      nodes.insert(nodes.end(), d.nodes + skip, d.nodes + d.world_end);   // only EXIT nodes follow: no links to fix
      rtx_scene_desc dm = d;
      dm.nodes = nodes.data(); dm.n_nodes = (uint32_t)nodes.size();
      world_begin = emit_range(dm, mb, (uint32_t)nodes.size(), true);
      if (units.size() & 1u) units.push_back(nop());  // the deferred program starts on a 16-byte boundary
      tail_begin = (uint32_t)units.size();
      if (has_xf) emit_enter(d, ctx);
      emit_range(d, root + 1, skip, false);
      units[defer_pos] |= (uint64_t)tail_begin << 32;  // the DEFER unit learns where the section's program starts
    } else {
      world_begin = emit_range(d, d.world_begin, d.world_end, true);
    }
    tail_near_quads = 0;
    tail_near_depth = tail_begin ? near_depth_of_tail(false) : 0u;
    if (tail_begin && tail_near_depth == 0) { tail_near_depth = near_depth_of_tail(true); tail_near_quads = tail_near_depth ? 1u : 0u; }
    if (split_head) find_head(d);
    head_near_depth = 0;
    if (head_end) {
      const int dpt = near_depth(world_begin + PROG_BOX_UNITS, head_end, 0, true);
      head_near_depth = dpt > 0 ? (uint32_t)dpt : 0u;
    }
    if (units.size() & 1u) units.push_back(nop());  // even unit count: the image is copied in 16-byte pieces
    units.insert(units.end(), 8, nop());            // tail pad: the successor prefetch of a box may read past END
    prog_units = (uint32_t)units.size();
    build_head_top();
    build_head_wide();
    // can the stretch of the world program that the wavefront pipeline walks per ray (behind the head section, up to the
    // deferred section) be swept by a wave in step (walk_flat)?
    main_flat = false;
    if (flat_walk) {
      uint32_t boxes = 0;
      bool ok = true;
      for (uint32_t i = head_end ? head_end + 1u : world_begin; i < prog_units && ok; i += prog_record_units(units[i])) {
        const uint32_t k = RTX_NODE_KIND((uint32_t)units[i]);
        if (k == PROG_NODE_END) break;
        if (k == RTX_NODE_MEDIUM) ok = false;
        if (k == RTX_NODE_BOX) ++boxes;
      }
      flat_begin = head_end ? head_end + 1u : world_begin;
      // the part in front of it (the head section) is walked per ray by the same kernels, with walk_program<.., PLAIN>:
      // it must hold no transform
      for (uint32_t i = world_begin; i < flat_begin && ok; i += prog_record_units(units[i])) {
        const uint32_t k = RTX_NODE_KIND((uint32_t)units[i]);
        if (k == RTX_NODE_ENTER || k == RTX_NODE_EXIT || k == RTX_NODE_MEDIUM || k == PROG_NODE_MEDIUM_S || k == PROG_NODE_DEFER) ok = false;
      }
      main_flat = ok && boxes <= 16u;
    }
  }
  bool main_flat = false;
  uint32_t flat_begin = 0;         // first unit of that stretch (head_end + 1 whether or not the head section is streamed)
  bool flat_walk = true;           // look for such a stretch (tests switch it off to compare)

  // The scene as the traversal code sees it, over host memory (tables of `d`, program of this object).
  Scene host_view(const rtx_scene_desc& d) const {
    Scene S;
    static_cast<rtx_scene_desc&>(S) = d;
    S.nodes = nullptr; S.n_nodes = 0;
    S.world_begin = world_begin; S.world_end = 0;
    S.media = media.data();
    S.prog = units.data(); S.n_units = prog_units; S.tail_begin = tail_begin;
    S.head_end = head_end; S.tail_near_depth = tail_near_depth; S.head_near_depth = head_near_depth; S.tail_near_quads = tail_near_quads; S._pad_near = 0; S.fresh_in_main = 0; S.light_quads = nullptr; S.light_spheres = nullptr; S.near_margin = RT_NEAR_MARGIN;
    S.head_top_begin = head_top_begin; S.head_top_nodes = head_top_nodes; S.head_top_units = head_top_units;
    S.head_wide_begin = head_wide_begin; S.head_wide_nodes = head_wide_nodes; S.head_wide_depth = head_wide_depth; S._pad_wide = 0; S.head_wide_far = head_wide_far;
    S.light_nodes = light_nodes.data(); S.n_light_nodes = (uint32_t)light_nodes.size();
    S.n_light_groups = light_group_begin.empty() ? 0u : (uint32_t)light_group_begin.size() - 1u;
    S.light_group_begin = light_group_begin.data(); S.light_far = light_far;
    S.main_flat = main_flat ? 1u : 0u; S.flat_begin = flat_begin;
    return S;
  }

 private:
  // Depth of the deferred section if it has the shape walk_stream_near needs: behind leading ENTER/NOP units, every
  // box is followed either by exactly two box subtrees that fill it, or by leaf records only: static-sphere runs, and
  // with `quads` also quads (general ones, axis-aligned runs) and BOX6 records behind their guards.  0 = another shape.
  uint32_t near_depth_of_tail(bool quads) const {
    uint32_t pos = tail_begin;
    const uint32_t n = (uint32_t)units.size();
    while (pos < n) {
      const uint32_t k = RTX_NODE_KIND((uint32_t)units[pos]);
      if (k == RTX_NODE_ENTER || k == PROG_NODE_NOP) pos += prog_record_units(units[pos]); else break;
    }
    uint32_t end = pos;
    while (end < n && RTX_NODE_KIND((uint32_t)units[end]) != PROG_NODE_END) end += prog_record_units(units[end]);
    if (end >= n) return 0;
    const int dpt = near_depth(pos, end, 0, quads);
    return dpt > 0 ? (uint32_t)dpt : 0u;
  }
  int near_depth(uint32_t pos, uint32_t end, int level, bool quads) const {   // -1: not the shape
    if (level > 48 || pos >= end) return -1;
    if (RTX_NODE_KIND((uint32_t)units[pos]) == RTX_NODE_BOX && RTX_NODE_IDX((uint32_t)units[pos]) != PROG_GUARD_IDX) {
      const uint32_t c0 = pos, c1 = (uint32_t)(units[c0] >> 32);
      if (c1 <= c0 || c1 >= end || RTX_NODE_KIND((uint32_t)units[c1]) != RTX_NODE_BOX || RTX_NODE_IDX((uint32_t)units[c1]) == PROG_GUARD_IDX ||
          (uint32_t)(units[c1] >> 32) != end) return -1;
      const int a = near_depth(c0 + PROG_BOX_UNITS, c1, level + 1, quads), b = near_depth(c1 + PROG_BOX_UNITS, end, level + 1, quads);
      return (a < 0 || b < 0) ? -1 : 1 + std::max(a, b);
    }
    for (uint32_t p = pos; p < end; p += prog_record_units(units[p])) {
      const uint32_t k = RTX_NODE_KIND((uint32_t)units[p]);
      if (k == RTX_NODE_BOX && quads) {   // only a guard, with its BOX6 record right behind it
        const uint32_t rec = p + PROG_BOX_UNITS;
        if (RTX_NODE_IDX((uint32_t)units[p]) != PROG_GUARD_IDX || rec >= end || RTX_NODE_KIND((uint32_t)units[rec]) != PROG_NODE_BOX6 ||
            (uint32_t)(units[p] >> 32) != rec + PROG_BOX6_UNITS) return -1;
        continue;
      }
      if (k != PROG_NODE_SPHERE_S && k != PROG_NODE_NOP && !(quads && (k == PROG_NODE_QUAD_AA || k == PROG_NODE_BOX6 || k == RTX_NODE_QUAD))) return -1;
    }
    return 0;
  }
  // The first levels of the head section's box tree in heap order, appended to the image: as many whole levels (at most
  // head_top_max_levels) as consist of boxes with two box children each.  Only for a head section that has the
  // nearer-child-first shape and is too large to be staged whole (a section that fits LDS is read from there anyway).
  void build_head_top() {
    head_top_begin = 0; head_top_nodes = 0; head_top_units = 0;
    if (!head_end || head_near_depth == 0 || (size_t)head_end * 8 <= 160u * 1024u) return;
    auto is_box = [&](uint32_t p) { return p < head_end && RTX_NODE_KIND((uint32_t)units[p]) == RTX_NODE_BOX && RTX_NODE_IDX((uint32_t)units[p]) != PROG_GUARD_IDX; };
    std::vector<uint32_t> pos{world_begin};
    if (!is_box(world_begin)) return;
    uint32_t levels = 1;
    for (size_t first = 0; levels < head_top_max_levels; ++levels) {   // level by level: every node needs two box children
      const size_t last = pos.size();
      std::vector<uint32_t> next;
      bool whole = true;
      for (size_t k = first; k < last && whole; ++k) {
        const uint32_t c0 = pos[k] + PROG_BOX_UNITS;
        if (!is_box(c0)) { whole = false; break; }
        const uint32_t c1 = (uint32_t)(units[c0] >> 32);
        if (!is_box(c1)) { whole = false; break; }
        next.push_back(c0); next.push_back(c1);
      }
      if (!whole) break;
      first = last;
      pos.insert(pos.end(), next.begin(), next.end());
    }
    if (levels < 3) return;
    head_top_nodes = (1u << levels) - 1u;
    pos.resize(head_top_nodes);
    head_top_begin = (uint32_t)units.size();
    std::vector<uint64_t> table;
    table.reserve((size_t)head_top_nodes * PROG_BOX_UNITS);
    for (uint32_t h = 0; h < head_top_nodes; ++h) table.insert(table.end(), units.begin() + pos[h], units.begin() + pos[h] + PROG_BOX_UNITS);
    units.insert(units.end(), table.begin(), table.end());
    if (units.size() & 1u) units.push_back(nop());
    for (uint32_t h = 0; h < head_top_nodes; h += 2) units.push_back((uint64_t)pos[h] | ((uint64_t)(h + 1 < head_top_nodes ? pos[h + 1] : 0u) << 32));
    if (units.size() & 1u) units.push_back(nop());
    head_top_units = (uint32_t)units.size() - head_top_begin;
  }
  // The box tree of such a head section with two levels folded into one, appended to the image for the walk from
  // global memory (walk_stream_wide): a record of PROG_WIDE_UNITS units (128 bytes, 128-byte aligned: one cache line)
  // per box that has two box children, holding the boxes of its (up to four) grandchildren, or of a child itself where
  // that child is a leaf, in the program's order: 24 floats (lo[3], hi[3] per slot, each bound rounded OUTWARD from the
  // program's double), then a descriptor per slot: the record of the slot's own box when that has box children (low
  // word, high word 0), or the unit position behind the box of a leaf slot (low word; high word: where its units end,
  // never 0), or PROG_WIDE_EMPTY.  Records lie in depth-first order like the program (a subtree is one stretch of
  // memory); record 0 belongs to the section's root box.
  // The table only CULLS: a ray that passes a box of the program passes the wider box here, and it passes every box
  // around it (a box contains its children's boxes exactly: aabb.rs:38-45 takes minima and maxima; every operation of
  // Aabb::hit, aabb.rs:132-167, is monotone in the bounds), so leaving the boxes in between untested loses nothing.
  // Whether a leaf's units are tested is decided by the program's own box record in front of them.
  // The bounds are padded by PROG_WIDE_PAD times the largest coordinate of the section's root box before they are
  // rounded outward: the walk tests them in single precision with one fused multiply-add per plane, measuring ray
  // parameters from where the ray enters the root box (slab_entry_or_inf_f), and its results differ from Aabb::hit's by
  // at most 2.4e-7 of that largest coordinate in space units, an eighth of the pad, for ray origins within
  // head_wide_far = PROG_WIDE_FAR times it (beyond that the shift of the origin itself loses digits).
  static float round_down(double v) { float f = (float)v; return (double)f >= v ? nextafterf(f, -INFINITY) : f; }
  static float round_up(double v) { float f = (float)v; return (double)f <= v ? nextafterf(f, INFINITY) : f; }
  void build_head_wide() {
    head_wide_begin = 0; head_wide_nodes = 0; head_wide_depth = 0;
    uint32_t depth = 0;
    if (!head_wide || !head_end || head_near_depth == 0 || (size_t)head_end * 8 <= 160u * 1024u) return;
    auto is_box = [&](uint32_t p) { return p < head_end && RTX_NODE_KIND((uint32_t)units[p]) == RTX_NODE_BOX && RTX_NODE_IDX((uint32_t)units[p]) != PROG_GUARD_IDX; };
    auto inner = [&](uint32_t p) { return is_box(p) && is_box(p + PROG_BOX_UNITS) && is_box((uint32_t)(units[p + PROG_BOX_UNITS] >> 32)); };
    if (!inner(world_begin)) return;
    double extent = 0.0;
    for (uint32_t j = 0; j < 6; ++j) { double v; memcpy(&v, &units[world_begin + 1 + j], 8); extent = std::max(extent, fabs(v)); }
    if (!(extent > 0.0) || !(extent < 1e300)) return;
    const double pad = extent * PROG_WIDE_PAD;
    std::vector<uint64_t> table;
    bool ok = true;
    // emits the record of inner box p (and, behind it, the records of its inner slots); returns its index
    std::function<uint32_t(uint32_t, int)> emit = [&](uint32_t p, int level) -> uint32_t {
      const uint32_t me = (uint32_t)(table.size() / PROG_WIDE_UNITS);
      if (level > 40 || me >= (1u << 30)) { ok = false; return 0u; }
      depth = std::max(depth, (uint32_t)level + 1u);
      table.insert(table.end(), PROG_WIDE_UNITS, 0ull);
      uint32_t slots[4], n = 0;
      const uint32_t c0 = p + PROG_BOX_UNITS, c1 = (uint32_t)(units[c0] >> 32);
      for (uint32_t c : {c0, c1}) {
        if (inner(c)) { slots[n++] = c + PROG_BOX_UNITS; slots[n++] = (uint32_t)(units[c + PROG_BOX_UNITS] >> 32); }
        else slots[n++] = c;
      }
      for (uint32_t k = 0; k < 4; ++k) {
        uint64_t desc = PROG_WIDE_EMPTY;
        if (k < n) {
          const uint32_t q = slots[k];
          float f[6];
          for (uint32_t j = 0; j < 6; ++j) {
            double v; memcpy(&v, &units[q + 1 + j], 8);
            f[j] = j < 3 ? round_down(v - pad) : round_up(v + pad);
          }
          memcpy((char*)(table.data() + (size_t)me * PROG_WIDE_UNITS) + 24u * k, f, 24);
          if (inner(q)) desc = (uint64_t)emit(q, level + 1);
          else desc = (uint64_t)(q + PROG_BOX_UNITS) | ((units[q] >> 32) << 32);
        }
        table[(size_t)me * PROG_WIDE_UNITS + 12 + k] = desc;
      }
      return me;
    };
    emit(world_begin, 0);
    while (units.size() & 15u) units.push_back(nop());
    if (!ok || units.size() + table.size() + 64u >= (1ull << 31)) return;   // unit positions of leaves go on the stack with a flag bit
    head_wide_begin = (uint32_t)units.size(); head_wide_nodes = (uint32_t)(table.size() / PROG_WIDE_UNITS); head_wide_depth = depth;
    head_wide_far = extent * PROG_WIDE_FAR;
    units.insert(units.end(), table.begin(), table.end());
  }
  static uint64_t nop() { return (uint64_t)RTX_NODE_MAKE(PROG_NODE_NOP, 0); }
  static uint64_t bits(double v) { uint64_t u; memcpy(&u, &v, 8); return u; }

  void build_compact(const rtx_scene_desc& d) {
    sphere_static.assign(d.n_spheres, 0);
    for (uint32_t i = 0; i < d.n_spheres; ++i) {
      const rtx_sphere& s = d.spheres[i];
      if (s.dc[0] == 0.0 && s.dc[1] == 0.0 && s.dc[2] == 0.0) sphere_static[i] = 1;
    }
    DAAQuad zero;
    memset(&zero, 0, sizeof(zero));
    daaquads.assign(d.n_quads, zero);
    quad_aa.assign(d.n_quads, 0);
    for (uint32_t i = 0; i < d.n_quads; ++i) {
      const rtx_quad& q = d.quads[i];
      int k = -1, zeros = 0;
      for (int c = 0; c < 3; ++c) {
        if (q.normal[c] == 0.0 && q.w[c] == 0.0) ++zeros;
        else k = c;
      }
      if (zeros != 2 || fabs(q.normal[k]) != 1.0 || !std::isfinite(q.w[k]) || !std::isfinite(q.d)) continue;
      const int a = (k + 1) % 3, b = (k + 2) % 3;
      const double vals[8] = {q.q[a], q.q[b], q.u[a], q.u[b], q.v[a], q.v[b], q.q[k], q.u[k] + q.v[k]};
      bool finite = true;
      for (double v : vals) finite = finite && std::isfinite(v);
      if (!finite) continue;
      quad_aa[i] = 1;
      DAAQuad& o = daaquads[i];
      o.d = q.d; o.nk = q.normal[k];
      o.qa = q.q[a]; o.qb = q.q[b]; o.ua = q.u[a]; o.ub = q.u[b]; o.va = q.v[a]; o.vb = q.v[b]; o.wk = q.w[k];
      o.axis = (uint32_t)k; o._pad = 0;
    }
  }

  static bool same_face(const DAAQuad& a, const DAAQuad& b) {
    return a.axis == b.axis && a.d == b.d && a.nk == b.nk && a.qa == b.qa && a.qb == b.qb && a.ua == b.ua &&
           a.ub == b.ub && a.va == b.va && a.vb == b.vb && a.wk == b.wk;
  }
  // Do quads[q0 .. q0+6) equal, field by field, the six faces box6_face() rebuilds from their corners?
  bool try_box6(const rtx_scene_desc& d, uint32_t q0, DBox6& out) const {
    if (q0 + 6 > d.n_quads) return false;
    for (uint32_t k = 0; k < 6; ++k) if (!quad_aa[q0 + k]) return false;
    const rtx_quad& front = d.quads[q0];       // q = (x0, y0, z1), u = (dx, 0, 0), v = (0, dy, 0)
    const rtx_quad& back = d.quads[q0 + 2];    // q = (x1, y0, z0)
    const rtx_quad& top = d.quads[q0 + 4];     // q = (x0, y1, z1)
    out.x0 = front.q[0]; out.y0 = front.q[1]; out.z1 = front.q[2];
    out.x1 = back.q[0]; out.z0 = back.q[2]; out.y1 = top.q[1];
    out.wz = daaquads[q0].wk; out.wx = daaquads[q0 + 1].wk; out.wy = daaquads[q0 + 4].wk;
    for (int f = 0; f < 6; ++f)
      if (!same_face(box6_face(out, f), daaquads[q0 + (uint32_t)f])) return false;
    return true;
  }

  struct Item { uint32_t kind, idx, count, node; DBox6 box6; };

  // Is the last top-level item of the world a box subtree, possibly inside one ENTER/EXIT pair, that is worth a
  // program of its own?  root = its root BOX node.  Media inside are excluded (their boundary programs live with the
  // world program).
  static bool find_deferrable(const rtx_scene_desc& d, uint32_t& root, bool& has_xf, uint32_t& ctx) {
    uint32_t i = d.world_begin, last = d.world_begin;
    while (i < d.world_end) {
      last = i;
      const uint32_t kind = RTX_NODE_KIND(d.nodes[i].kind_idx);
      if (kind == RTX_NODE_BOX) {
        i = d.nodes[i].skip;
      } else if (kind == RTX_NODE_ENTER) {
        int depth = 1;
        for (++i; i < d.world_end && depth > 0; ++i) {
          const uint32_t k = RTX_NODE_KIND(d.nodes[i].kind_idx);
          if (k == RTX_NODE_ENTER) ++depth;
          else if (k == RTX_NODE_EXIT) --depth;
        }
      } else {
        ++i;
      }
    }
    if (d.world_end <= d.world_begin) return false;
    if (last == d.world_begin) return false;   // a world of one subtree: deferring it would leave the first pass empty
    const uint32_t kind = RTX_NODE_KIND(d.nodes[last].kind_idx);
    uint32_t end = d.world_end;
    if (kind == RTX_NODE_BOX) {
      root = last; has_xf = false; ctx = 0;
    } else if (kind == RTX_NODE_ENTER) {
      // k nested transforms around the subtree: k ENTER nodes, the root box, ..., k EXIT nodes.  A context is the whole
      // chain from the world, so the innermost ENTER alone sets up the section's program.
      uint32_t k = 0;
      while (last + k < d.world_end && RTX_NODE_KIND(d.nodes[last + k].kind_idx) == RTX_NODE_ENTER) ++k;
      if (last + k >= d.world_end || d.world_end - last < 2 * k + 1) return false;
      if (RTX_NODE_KIND(d.nodes[last + k].kind_idx) != RTX_NODE_BOX) return false;
      for (uint32_t j = 0; j < k; ++j)
        if (RTX_NODE_KIND(d.nodes[d.world_end - 1 - j].kind_idx) != RTX_NODE_EXIT) return false;
      root = last + k; has_xf = true; ctx = RTX_NODE_IDX(d.nodes[last + k - 1].kind_idx); end = d.world_end - k;
    } else {
      return false;
    }
    if (d.nodes[root].skip != end) return false;
    uint32_t boxes = 0;
    for (uint32_t k = root + 1; k < end; ++k) {
      const uint32_t kk = RTX_NODE_KIND(d.nodes[k].kind_idx);
      if (kk == RTX_NODE_MEDIUM) return false;
      if (kk == RTX_NODE_BOX) ++boxes;
    }
    return boxes >= 32;
  }

  // Head section: the world program starts (in the world context) with a box subtree of at least 32 boxes that holds
  // no medium; it ends at the SYNC unit its root box skips to.
  void find_head(const rtx_scene_desc& d) {
    head_end = 0;
    if (d.world_end <= d.world_begin || RTX_NODE_KIND(d.nodes[d.world_begin].kind_idx) != RTX_NODE_BOX) return;
    const uint64_t root = units[world_begin];
    if (RTX_NODE_KIND((uint32_t)root) != RTX_NODE_BOX) return;
    const uint32_t end = (uint32_t)(root >> 32);
    if (end <= world_begin || end >= units.size() || RTX_NODE_KIND((uint32_t)units[end]) != PROG_NODE_SYNC) return;
    uint32_t boxes = 0;
    for (uint32_t i = world_begin; i < end; i += prog_record_units(units[i])) {
      const uint32_t k = RTX_NODE_KIND((uint32_t)units[i]);
      if (k == RTX_NODE_MEDIUM || k == PROG_NODE_MEDIUM_S || k == PROG_NODE_DEFER || k == PROG_NODE_END || k == PROG_NODE_SYNC) return;
      if (k == RTX_NODE_BOX) ++boxes;
    }
    if (boxes >= 32) head_end = end;
  }

  // ENTER record of transform context `ctx` (header + op chain).
  void emit_enter(const rtx_scene_desc& d, uint32_t ctx) {
    const rtx_xform x = d.xforms[ctx];
    units.push_back((uint64_t)RTX_NODE_MAKE(RTX_NODE_ENTER, ctx) | ((uint64_t)x.n_ops << 32));
    for (uint32_t s = 0; s < x.n_ops; ++s) {
      const rtx_xform_op op = d.xform_ops[x.first_op + s];
      units.push_back((uint64_t)op.kind); units.push_back(bits(op.a)); units.push_back(bits(op.b)); units.push_back(bits(op.c));
    }
  }

  // Is the boundary of medium m exactly one static sphere (in the medium's own transform context)?
  bool sphere_bounded(const rtx_scene_desc& d, uint32_t m) const {
    const rtx_medium& md = d.media[m];
    if (md.b_end - md.b_begin != 1u) return false;
    const uint32_t ki = d.nodes[md.b_begin].kind_idx;
    return RTX_NODE_KIND(ki) == RTX_NODE_SPHERE && sphere_static[RTX_NODE_IDX(ki)];
  }

  uint32_t item_units(const rtx_scene_desc& d, const Item& it) const {
    switch (it.kind) {
      case RTX_NODE_BOX: return PROG_BOX_UNITS;
      case PROG_NODE_SPHERE_S: return 1u + PROG_SPHERE_S_UNITS * it.count;
      case PROG_NODE_QUAD_AA: return 1u + PROG_QUAD_AA_UNITS * it.count;
      case PROG_NODE_BOX6: return (guard_box6 ? PROG_BOX_UNITS : 0u) + PROG_BOX6_UNITS;
      case RTX_NODE_ENTER: case RTX_NODE_EXIT: return 1u + PROG_XFORM_OP_UNITS * d.xforms[it.idx].n_ops;
      case RTX_NODE_MEDIUM: return PROG_MEDIUM_UNITS;
      case PROG_NODE_MEDIUM_S: return PROG_MEDIUM_S_UNITS;
      case RTX_NODE_SPHERE: return PROG_SPHERE_UNITS;
      default: return 1u;
    }
  }

  // Appends nodes[b,e) + END; returns the unit index of the first record.  Consecutive static spheres / axis-aligned
  // quads whose table indices are consecutive too (the six faces of a box) become one run unit; a run never spans a
  // node some box skips to.  With `sync`, a SYNC unit follows every top-level box subtree of the range (skip links
  // that end such a subtree land on the SYNC).
  uint32_t emit_range(const rtx_scene_desc& d, uint32_t b, uint32_t e, bool sync) {
    const uint32_t n = e - b;
    std::vector<uint8_t> target((size_t)n + 1, 0), sync_before((size_t)n + 1, 0);
    uint32_t enclosing_end = b;  // nodes before this index lie inside some box subtree
    for (uint32_t i = b; i < e; ++i) {
      if (RTX_NODE_KIND(d.nodes[i].kind_idx) != RTX_NODE_BOX) continue;
      const uint32_t skip = std::min(std::max(d.nodes[i].skip, b), e);  // validated to lie in (i, e]
      target[skip - b] = 1;
      if (sync && i >= enclosing_end) sync_before[skip - b] = 1;
      enclosing_end = std::max(enclosing_end, skip);
    }
    DBox6 no_box;
    memset(&no_box, 0, sizeof(no_box));
    std::vector<Item> items;
    // An ENTER/EXIT only sets the current context and rebuilds the local ray from the world ray, so it is redundant
    // when another ENTER/EXIT follows immediately, and so is one right before the END of the world program.
    std::vector<uint8_t> dropped((size_t)n + 1, 0);
    for (uint32_t i = b; i < e; ++i) {
      uint32_t kind = RTX_NODE_KIND(d.nodes[i].kind_idx);
      const uint32_t idx = RTX_NODE_IDX(d.nodes[i].kind_idx);
      if (kind == RTX_NODE_ENTER || kind == RTX_NODE_EXIT) {
        const uint32_t nk = i + 1 < e ? RTX_NODE_KIND(d.nodes[i + 1].kind_idx) : (uint32_t)RTX_NODE_NONE;
        if (nk == RTX_NODE_ENTER || nk == RTX_NODE_EXIT || (i + 1 == e && sync)) {
          dropped[i - b] = 1;
          if (sync_before[i - b]) { sync_before[i - b] = 0; sync_before[i - b + 1] = 1; }
          continue;
        }
      }
      if (kind == RTX_NODE_SPHERE && sphere_static[idx]) kind = PROG_NODE_SPHERE_S;
      else if (kind == RTX_NODE_QUAD && quad_aa[idx]) kind = PROG_NODE_QUAD_AA;
      else if (kind == RTX_NODE_MEDIUM && fuse_sphere_media && sphere_bounded(d, idx)) kind = PROG_NODE_MEDIUM_S;
      const bool run_kind = kind == PROG_NODE_SPHERE_S || kind == PROG_NODE_QUAD_AA;
      if (run_kind && !items.empty() && !target[i - b] && items.back().kind == kind &&
          items.back().idx + items.back().count == idx && items.back().node + items.back().count == i &&
          items.back().count < 64)
        items.back().count += 1;
      else
        items.push_back(Item{kind, idx, 1u, i, no_box});
    }
    // split axis-aligned-quad runs: every six consecutive faces that form a box_new box become one BOX6 item
    {
      std::vector<Item> split;
      for (const Item& it : items) {
        if (it.kind != PROG_NODE_QUAD_AA) { split.push_back(it); continue; }
        uint32_t k = 0;
        while (k < it.count) {
          DBox6 bx;
          if (it.count - k >= 6 && try_box6(d, it.idx + k, bx)) {
            split.push_back(Item{PROG_NODE_BOX6, it.idx + k, 6u, it.node + k, bx});
            n_box6 += 1;
            k += 6;
          } else {
            if (!split.empty() && split.back().kind == PROG_NODE_QUAD_AA && split.back().node + split.back().count == it.node + k &&
                split.back().idx + split.back().count == it.idx + k && k > 0)
              split.back().count += 1;
            else
              split.push_back(Item{PROG_NODE_QUAD_AA, it.idx + k, 1u, it.node + k, no_box});
            k += 1;
          }
        }
      }
      items.swap(split);
    }
    // unit positions; link[k] = where a skip to node b+k lands
    std::vector<uint32_t> link((size_t)n + 1, 0), item_pos(items.size());
    uint32_t at = (uint32_t)units.size();
    for (size_t k = 0; k < items.size(); ++k) {
      const uint32_t rel = items[k].node - b;
      if (sync_before[rel]) { link[rel] = at; at += 1; }
      item_pos[k] = at;
      if (!sync_before[rel]) link[rel] = at;
      at += item_units(d, items[k]);
    }
    link[n] = at;
    if (sync_before[n]) at += 1;
    for (uint32_t rel = n; rel-- > 0;)
      if (dropped[rel]) link[rel] = link[rel + 1];   // a skip to a dropped node lands on what follows it
    const uint32_t end_at = at;
    units.resize((size_t)end_at + 1, nop());
    for (size_t k = 0; k < items.size(); ++k) {
      const Item& it = items[k];
      const uint32_t p = item_pos[k], rel = it.node - b;
      if (sync_before[rel]) units[link[rel]] = (uint64_t)RTX_NODE_MAKE(PROG_NODE_SYNC, 0);
      uint64_t* u = &units[p];
      if (it.kind == RTX_NODE_BOX) {
        const rtx_node nd = d.nodes[it.node];
        const uint32_t skip = std::min(std::max(nd.skip, b), e);
        u[0] = (uint64_t)nd.kind_idx | ((uint64_t)link[skip - b] << 32);
        memcpy(u + 1, &d.boxes[it.idx], sizeof(rtx_aabb));
      } else if (it.kind == PROG_NODE_SPHERE_S) {
        u[0] = (uint64_t)RTX_NODE_MAKE(it.kind, it.idx) | ((uint64_t)it.count << 32);
        for (uint32_t s = 0; s < it.count; ++s) {
          const rtx_sphere& sp = d.spheres[it.idx + s];
          const DSphere rec = {sp.c0[0], sp.c0[1], sp.c0[2], sp.radius};
          memcpy(u + 1 + PROG_SPHERE_S_UNITS * s, &rec, sizeof(rec));
        }
      } else if (it.kind == RTX_NODE_SPHERE) {   // a general (moving) sphere carries its record inline
        const rtx_sphere& sp = d.spheres[it.idx];
        u[0] = (uint64_t)d.nodes[it.node].kind_idx;
        const double rec[7] = {sp.c0[0], sp.c0[1], sp.c0[2], sp.dc[0], sp.dc[1], sp.dc[2], sp.radius};
        memcpy(u + 1, rec, sizeof(rec));
      } else if (it.kind == PROG_NODE_QUAD_AA) {
        u[0] = (uint64_t)RTX_NODE_MAKE(it.kind, it.idx) | ((uint64_t)it.count << 32);
        for (uint32_t s = 0; s < it.count; ++s) memcpy(u + 1 + PROG_QUAD_AA_UNITS * s, &daaquads[it.idx + s], sizeof(DAAQuad));
      } else if (it.kind == PROG_NODE_BOX6) {
        if (guard_box6) {
          // Guard: a box test of the box's own bounds, widened, in front of its six faces.  A ray that misses the widened
          // bounds cannot hit a face (the faces lie on the unwidened surface and the margin exceeds the rounding of both
          // tests by many orders of magnitude), so the guard only saves the six quad tests and never changes a result.
          const DBox6& bx = it.box6;
          const double lo[3] = {std::min(bx.x0, bx.x1), std::min(bx.y0, bx.y1), std::min(bx.z0, bx.z1)};
          const double hi[3] = {std::max(bx.x0, bx.x1), std::max(bx.y0, bx.y1), std::max(bx.z0, bx.z1)};
          rtx_aabb g;
          for (int a = 0; a < 3; ++a) {
            const double pad = 1e-6 * (1.0 + std::max(fabs(lo[a]), fabs(hi[a])));
            g.lo[a] = lo[a] - pad; g.hi[a] = hi[a] + pad;
          }
          u[0] = (uint64_t)RTX_NODE_MAKE(RTX_NODE_BOX, PROG_GUARD_IDX) | ((uint64_t)(p + PROG_BOX_UNITS + PROG_BOX6_UNITS) << 32);
          memcpy(u + 1, &g, sizeof(g));
          u += PROG_BOX_UNITS;
        }
        u[0] = (uint64_t)RTX_NODE_MAKE(it.kind, it.idx) | ((uint64_t)6u << 32);
        memcpy(u + 1, &it.box6, sizeof(DBox6));
      } else if (it.kind == RTX_NODE_ENTER || it.kind == RTX_NODE_EXIT) {
        const rtx_xform x = d.xforms[it.idx];
        u[0] = (uint64_t)d.nodes[it.node].kind_idx | ((uint64_t)x.n_ops << 32);
        for (uint32_t s = 0; s < x.n_ops; ++s) {
          const rtx_xform_op op = d.xform_ops[x.first_op + s];
          uint64_t* r = u + 1 + PROG_XFORM_OP_UNITS * s;
          r[0] = (uint64_t)op.kind; r[1] = bits(op.a); r[2] = bits(op.b); r[3] = bits(op.c);
        }
      } else if (it.kind == PROG_NODE_MEDIUM_S) {
        const rtx_sphere& sp = d.spheres[RTX_NODE_IDX(d.nodes[d.media[it.idx].b_begin].kind_idx)];
        const DSphere rec = {sp.c0[0], sp.c0[1], sp.c0[2], sp.radius};
        u[0] = (uint64_t)RTX_NODE_MAKE(PROG_NODE_MEDIUM_S, it.idx);
        u[1] = bits(d.media[it.idx].neg_inv_density);
        u[2] = (d.media[it.idx].needs_uv ? PROG_MEDIUM_NEEDS_UV : 0ull) |
               ((draw_first_thin_media && -d.media[it.idx].neg_inv_density >= 2.0 * sp.radius) ? PROG_MEDIUM_DRAW_FIRST : 0ull);
        memcpy(u + 3, &rec, sizeof(rec));
      } else if (it.kind == RTX_NODE_MEDIUM) {
        ++n_general_media;
        u[0] = (uint64_t)d.nodes[it.node].kind_idx | ((uint64_t)media[it.idx].b_begin << 32);
        u[1] = bits(d.media[it.idx].neg_inv_density);
        u[2] = d.media[it.idx].needs_uv ? PROG_MEDIUM_NEEDS_UV : 0ull;
      } else {
        u[0] = (uint64_t)d.nodes[it.node].kind_idx;
        if (it.kind == PROG_NODE_DEFER) defer_pos = p;
      }
    }
    if (sync_before[n]) units[link[n]] = (uint64_t)RTX_NODE_MAKE(PROG_NODE_SYNC, 0);
    units[end_at] = (uint64_t)RTX_NODE_MAKE(PROG_NODE_END, 0);
    return n ? link[0] : end_at;
  }
};

}  // namespace rt
