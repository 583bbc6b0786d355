// TEST INFRASTRUCTURE ONLY — host-compiled build of the device core headers
// (raytracing_rust_amd/csrc/core/*.h).  It lets the CPU-only test tier step
// through the very code the HIP kernels run (hit-program traversal, shading,
// RNG addressing) against the oracle without a GPU.  The product library does
// not contain, link or load this file; it has no CPU path.
#include <stdint.h>
// step tracer for the scheduling study (tools/sched_sim.py): unit kinds a ray steps through, in order
static thread_local uint8_t* g_trace = nullptr;
static thread_local uint32_t g_trace_len = 0, g_trace_cap = 0;
#define RT_TRACE_STEP(kind) do { if (g_trace && g_trace_len < g_trace_cap) g_trace[g_trace_len++] = (uint8_t)(kind); } while (0)
// ray log of the replay probes (sim_tape_path): the ray of every segment of a path, in order
#include <vector>
static thread_local std::vector<double>* g_ray_log = nullptr;
#define RT_TRACE_RAY(ray) do { if (g_ray_log) { const double f_[7] = {(ray).o.x, (ray).o.y, (ray).o.z, (ray).d.x, (ray).d.y, (ray).d.z, (ray).tm}; g_ray_log->insert(g_ray_log->end(), f_, f_ + 7); } } while (0)
#define RT_RNG_TAPE 1   // the device core's draws can be replayed from a caller's numbers (rt_rng.h)
#include <atomic>
#include <thread>
#include <functional>
#include <vector>

#include "../../raytracing_rust_amd/csrc/core/rt_program.h"
#include "../../raytracing_rust_amd/csrc/core/rt_shade.h"

using namespace rt;

static bool g_fuse_sphere_media = true, g_defer_last_section = true, g_guard_box6 = true, g_draw_first = false, g_light_table = true;  // compile options under test
static uint64_t g_last_probes = 0;
static int g_near_walks = 0, g_top_walks = 0, g_wide_walks = 0;
static double g_near_margin = 0.0;
static int g_head_table = 2;  // head section in global memory: 2 = the folded table when the program has one, 1 = first levels in heap order, 0 = neither
static int g_near = 0;  // staged walk: 1 = the deferred section nearer child first, 2 = the head section too

extern "C" {

int sim_first_hit(const rtx_scene_desc* desc, const double* rays, size_t n, double* t, double* normal, uint32_t* prim_id,
                  uint32_t* mat_id, uint8_t* front) {
  CompiledProgram prog;
  prog.fuse_sphere_media = g_fuse_sphere_media; prog.draw_first_thin_media = g_draw_first; prog.light_table = g_light_table; prog.defer_last_section = g_defer_last_section; prog.guard_box6 = g_guard_box6; prog.compile(*desc);
  const Scene view = prog.host_view(*desc);
  const Scene* S = &view;
  for (size_t i = 0; i < n; ++i) {
    const double* r = rays + 7 * i;
    Ray ray;
    ray.o = v3(r[0], r[1], r[2]); ray.d = v3(r[3], r[4], r[5]); ray.tm = r[6];
    Hit h;
    RayRegs rs;
    rs.r = ray;
    closest_hit<false>(*S, S->prog, S->world_begin, true, rs, ray.tm, 0.001, RT_INF, h, false, PathRng(), 0, nullptr);
    const bool any = hit_any(h);
    Surface sf;
    if (any) resolve_hit(*S, ray, h, false, nullptr, sf);
    if (t) t[i] = any ? h.t : RT_INF;
    if (normal) { normal[3 * i] = any ? sf.n.x : 0; normal[3 * i + 1] = any ? sf.n.y : 0; normal[3 * i + 2] = any ? sf.n.z : 0; }
    if (prim_id) prim_id[i] = any ? sf.prim_id : 0xFFFFFFFFu;
    if (mat_id) mat_id[i] = any ? sf.mat : 0xFFFFFFFFu;
    if (front) front[i] = any ? (sf.front ? 1 : 0) : 0;
  }
  return 0;
}

// The wavefront pipeline's three-stage first hit, stage by stage as the GPU launches it: the head section as a stream
// (walk_stream), the rest of the world program from the hit found so far with deferred rays only noted, and the
// deferred section as a stream over the noted rays.  Same outputs as sim_first_hit; returns the number of rays
// that went through a stream walk, -1 if the program has neither a head nor a deferred section.
struct HostFeed {
  const double* rays; const uint32_t* list; size_t n, next;
  Hit* hits; bool from_hits;
  bool more() const { return next < n; }
  bool take(bool want, uint32_t& id) { if (!want || next >= n) return false; id = list ? list[next] : (uint32_t)next; ++next; return true; }
  RayRegs ray(uint32_t id) const {
    RayRegs rs;
    if (n == 0) { rs.r.o = v3(0, 0, 0); rs.r.d = v3(0, 0, 1); rs.r.tm = 0; return rs; }
    const double* r = rays + 7 * (size_t)id;
    rs.r.o = v3(r[0], r[1], r[2]); rs.r.d = v3(r[3], r[4], r[5]); rs.r.tm = r[6];
    return rs;
  }
  double time(uint32_t id) const { return rays[7 * (size_t)id + 6]; }
  void start(uint32_t id, double& tmax, uint32_t& kind, uint32_t& ctx) const {
    if (from_hits) { tmax = hits[id].t; kind = hits[id].kind_idx; ctx = hits[id].ctx; }
    else { tmax = RT_INF; kind = RTX_NODE_MAKE(RTX_NODE_NONE, 0); ctx = 0; }
  }
  void commit(uint32_t id, double tmax, uint32_t kind, uint32_t ctx) const { hits[id].t = tmax; hits[id].kind_idx = kind; hits[id].ctx = ctx; }
  int* redos;                     // rays the nearer-child-first walk walked a second time, in the fixed order
  void note_redo() const { if (redos) *redos += 1; }
};
static int g_redos = 0;

int sim_first_hit_staged(const rtx_scene_desc* desc, const double* rays, size_t n, double* t, double* normal, uint32_t* prim_id,
                         uint32_t* mat_id, uint8_t* front) {
  CompiledProgram prog;
  prog.fuse_sphere_media = g_fuse_sphere_media; prog.draw_first_thin_media = g_draw_first; prog.light_table = g_light_table; prog.defer_last_section = g_defer_last_section; prog.guard_box6 = g_guard_box6; prog.compile(*desc);
  Scene S = prog.host_view(*desc);
  if (g_near_margin > 0.0) S.near_margin = g_near_margin;
  if (!S.head_end && !S.tail_begin) return -1;
  std::vector<Hit> hits(n);
  std::vector<uint32_t> deferred;
  int streamed = 0, near_walks = 0, redos = 0;
  if (S.head_end) {
    HostFeed feed{rays, nullptr, n, 0, hits.data(), false, &redos};
    std::vector<uint32_t> stk(S.head_near_depth + 1u);
    if (g_near == 2 && S.head_near_depth > 0) {
      if (S.head_wide_nodes && g_head_table == 2) {   // as the GPU walks it by default: two levels of the tree per step
        std::vector<uint32_t> wstk(3u * S.head_wide_depth + 1u);
        walk_stream_wide<true>(S, S.prog, S.world_begin, feed, wstk.data(), 1u, S.prog + S.head_wide_begin);
        g_wide_walks += 1;
      } else if (S.head_top_nodes && g_head_table >= 1) {   // as the GPU walks a head section that stays in global memory: first levels from the table
        const uint64_t* top = S.prog + S.head_top_begin;
        const uint32_t box_units = (S.head_top_nodes * PROG_BOX_UNITS + 1u) & ~1u;
        walk_stream_near<true, false, true>(S, S.prog, S.world_begin, feed, stk.data(), 1u, true, top, (const uint32_t*)(top + box_units), S.head_top_nodes);
        g_top_walks += 1;
      } else {
        walk_stream_near<true, true>(S, S.prog, S.world_begin, feed, stk.data(), 1u, true);
      }
      near_walks += 1;
    }
    else walk_stream(S, S.prog, S.world_begin, feed);
    streamed += (int)n;
  }
  for (size_t i = 0; i < n; ++i) {
    HostFeed one{rays, nullptr, n, 0, hits.data(), false, nullptr};
    const RayRegs rs = one.ray((uint32_t)i);
    Hit h;
    const double t0 = S.head_end ? hits[i].t : RT_INF;
    const uint32_t k0 = S.head_end ? hits[i].kind_idx : (uint32_t)RTX_NODE_MAKE(RTX_NODE_NONE, 0), c0 = S.head_end ? hits[i].ctx : 0u;
    const bool d = walk_program<MEDIA_NONE, true>(S, S.prog, S.head_end ? S.head_end + 1u : S.world_begin, true, rs, rs.r.tm, 0.001, t0, h, false,
                                                  PathRng(), 0, nullptr, k0, c0);
    hits[i] = h;
    if (d) deferred.push_back((uint32_t)i);
  }
  if (S.tail_begin) {
    HostFeed feed{rays, deferred.data(), deferred.size(), 0, hits.data(), true, &redos};
    std::vector<uint32_t> stk(S.tail_near_depth + 1u);
    if (g_near >= 1 && S.tail_near_depth > 0) {
      if (S.tail_near_quads) walk_stream_near<true, false>(S, S.prog, S.tail_begin, feed, stk.data(), 1u, false);
      else walk_stream_near<false, true>(S, S.prog, S.tail_begin, feed, stk.data(), 1u, false);
      near_walks += 1;
    }
    else walk_stream(S, S.prog, S.tail_begin, feed);
    streamed += (int)deferred.size();
  }
  for (size_t i = 0; i < n; ++i) {
    HostFeed one{rays, nullptr, n, 0, hits.data(), false, nullptr};
    const Ray ray = one.ray((uint32_t)i).r;
    const Hit h = hits[i];
    const bool any = hit_any(h);
    Surface sf;
    if (any) resolve_hit(S, ray, h, false, nullptr, sf);
    if (t) t[i] = any ? h.t : RT_INF;
    if (normal) { normal[3 * i] = any ? sf.n.x : 0; normal[3 * i + 1] = any ? sf.n.y : 0; normal[3 * i + 2] = any ? sf.n.z : 0; }
    if (prim_id) prim_id[i] = any ? sf.prim_id : 0xFFFFFFFFu;
    if (mat_id) mat_id[i] = any ? sf.mat : 0xFFFFFFFFu;
    if (front) front[i] = any ? (sf.front ? 1 : 0) : 0;
  }
  g_near_walks = near_walks; g_redos = redos;
  return streamed;
}

// Invariants of the folded table of a head section in global memory (CompiledProgram::build_head_wide), checked
// against the program it was made from.  Returns the number of leaf slots, 0 if the program has no such table, and a
// negative number naming the first violated invariant.
long sim_check_wide_table(const rtx_scene_desc* desc, uint32_t* n_records, uint32_t* depth) {
  CompiledProgram prog;
  prog.compile(*desc);
  if (!prog.head_wide_nodes) return 0;
  const std::vector<uint64_t>& u = prog.units;
  if (prog.head_wide_begin % 16u) return -1;                                       // records are 128-byte aligned
  if ((size_t)prog.head_wide_begin + (size_t)prog.head_wide_nodes * PROG_WIDE_UNITS != u.size()) return -2;
  auto is_box = [&](uint32_t p) { return p < prog.head_end && RTX_NODE_KIND((uint32_t)u[p]) == RTX_NODE_BOX && RTX_NODE_IDX((uint32_t)u[p]) != PROG_GUARD_IDX; };
  // leaf boxes of the program's tree, in program order
  std::vector<uint32_t> leaves;
  for (uint32_t p = prog.world_begin; p < prog.head_end; p += prog_record_units(u[p]))
    if (is_box(p) && !is_box(p + PROG_BOX_UNITS)) leaves.push_back(p + PROG_BOX_UNITS);
  std::vector<uint32_t> seen;
  std::vector<char> visited(prog.head_wide_nodes, 0);
  uint32_t max_depth = 0;
  long bad = 0;
  // every record: its slots' boxes contain the program's boxes they stand for, with room; a slot's record lies behind it
  std::function<void(uint32_t, uint32_t, uint32_t)> walk = [&](uint32_t rec, uint32_t box_pos, uint32_t level) {
    if (bad) return;
    if (rec >= prog.head_wide_nodes || visited[rec]) { bad = -3; return; }
    visited[rec] = 1;
    max_depth = std::max(max_depth, level + 1u);
    // the program boxes this record's slots stand for: grandchildren of box_pos, or a child that is a leaf
    uint32_t slots[4], n = 0;
    const uint32_t c0 = box_pos + PROG_BOX_UNITS, c1 = (uint32_t)(u[c0] >> 32);
    for (uint32_t c : {c0, c1}) {
      if (is_box(c + PROG_BOX_UNITS)) { slots[n++] = c + PROG_BOX_UNITS; slots[n++] = (uint32_t)(u[c + PROG_BOX_UNITS] >> 32); }
      else slots[n++] = c;
    }
    const uint64_t* r = u.data() + prog.head_wide_begin + (size_t)rec * PROG_WIDE_UNITS;
    const float* fb = (const float*)r;
    for (uint32_t k = 0; k < 4; ++k) {
      const uint64_t d = r[12 + k];
      if (k >= n) { if (d != PROG_WIDE_EMPTY) bad = -4; continue; }
      const uint32_t q = slots[k];
      for (uint32_t j = 0; j < 6; ++j) {
        double v; memcpy(&v, &u[q + 1 + j], 8);
        const double f = (double)fb[6 * k + j];
        if (j < 3 ? !(f < v) : !(f > v)) { bad = -5; return; }                     // strictly outside the program's box
      }
      if (is_box(q + PROG_BOX_UNITS)) {
        if ((d >> 32) != 0ull || (uint32_t)d <= rec) { bad = -6; return; }         // depth first: a child record lies behind
        walk((uint32_t)d, q, level + 1u);
      } else {
        if ((uint32_t)d != q + PROG_BOX_UNITS || (uint32_t)(d >> 32) != (uint32_t)(u[q] >> 32) || (d >> 32) == 0ull) { bad = -7; return; }
        seen.push_back((uint32_t)d);
      }
    }
  };
  walk(0u, prog.world_begin, 0u);
  if (bad) return bad;
  for (char v : visited) if (!v) return -8;                                        // no record is unreachable
  if (seen != leaves) return -9;                                                   // every leaf once, in program order
  if (max_depth != prog.head_wide_depth) return -10;
  if (!(prog.head_wide_far > 0.0)) return -11;
  if (n_records) *n_records = prog.head_wide_nodes;
  if (depth) *depth = prog.head_wide_depth;
  return (long)seen.size();
}

void sim_set_near(int mode) { g_near = mode; }
int sim_near_walks(void) { return g_near_walks; }
void sim_set_head_table(int mode) { g_head_table = mode; }
void sim_set_near_margin(double m) { g_near_margin = m; }   // the order watch's relative margin (0 = the default)
int sim_wide_walks(void) { return g_wide_walks; }      // head sections walked through the folded table, so far
int sim_top_walks(void) { return g_top_walks; }        // head sections walked with their first levels from the heap-ordered table, so far
int sim_order_redos(void) { return g_redos; }          // rays of the last staged walk that went through the fixed order again   // sections the last staged walk took nearer child first
void sim_set_fuse_sphere_media(int on) { g_fuse_sphere_media = on != 0; }
void sim_set_guard_box6(int on) { g_guard_box6 = on != 0; }
void sim_set_light_table(int on) { g_light_table = on != 0; }   // boxes over the light leaves cull lights_pdf_value
void sim_set_draw_first(int on) { g_draw_first = on != 0; }   // thin sphere-bounded media draw their free path first
void sim_set_defer_last_section(int on) { g_defer_last_section = on != 0; }

// counters: paths, segments, pdf_probes
int sim_render(const rtx_scene_desc* desc, const rtx_camera_desc* cam, const rtx_render_opts* o, double* accum,
               uint64_t* counters, int nthreads) {
  CompiledProgram prog;
  prog.fuse_sphere_media = g_fuse_sphere_media; prog.draw_first_thin_media = g_draw_first; prog.light_table = g_light_table; prog.defer_last_section = g_defer_last_section; prog.guard_box6 = g_guard_box6; prog.compile(*desc);
  const Scene view = prog.host_view(*desc);
  const Scene* S = &view;
  if (nthreads < 1) nthreads = 1;
  const int W = cam->image_width;
  std::atomic<int> next_row(o->y0);
  std::vector<uint64_t> segs((size_t)nthreads, 0), probes((size_t)nthreads, 0), paths((size_t)nthreads, 0);
  auto worker = [&](int tid) {
    for (;;) {
      int j = next_row.fetch_add(1);
      if (j >= o->y1) break;
      for (int i = o->x0; i < o->x1; ++i) {
        V3 sum = v3(0, 0, 0);
        for (int sidx = o->sample_begin; sidx < o->sample_end; ++sidx) {
          uint32_t sg = 0, pb = 0;
          sum = sum + trace_path(*S, *cam, o->use_lights != 0 && S->has_lights != 0, i, j, sidx, o->seed, sg, pb);
          segs[(size_t)tid] += sg; probes[(size_t)tid] += pb; paths[(size_t)tid] += 1;
        }
        double* a = accum + ((size_t)j * (size_t)W + (size_t)i) * 3;
        a[0] += sum.x; a[1] += sum.y; a[2] += sum.z;
      }
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nthreads; ++t) th.emplace_back(worker, t);
  worker(0);
  for (auto& t : th) t.join();
  if (counters) {
    counters[0] = counters[1] = counters[2] = 0;
    for (int t = 0; t < nthreads; ++t) { counters[0] += paths[(size_t)t]; counters[1] += segs[(size_t)t]; counters[2] += probes[(size_t)t]; }
  }
  return 0;
}

// For each ray: the kinds of the units it steps through (media included, RNG keyed by ray index).  seq is n x cap
// bytes, len receives the per-ray lengths.
void sim_trace_steps(const rtx_scene_desc* desc, const double* rays, size_t n, uint32_t cap, uint8_t* seq, uint32_t* len) {
  CompiledProgram prog;
  prog.fuse_sphere_media = g_fuse_sphere_media; prog.draw_first_thin_media = g_draw_first; prog.light_table = g_light_table; prog.defer_last_section = g_defer_last_section; prog.guard_box6 = g_guard_box6; prog.compile(*desc);
  const Scene S = prog.host_view(*desc);
  for (size_t i = 0; i < n; ++i) {
    const double* r = rays + 7 * i;
    RayRegs rs;
    rs.r.o = v3(r[0], r[1], r[2]); rs.r.d = v3(r[3], r[4], r[5]); rs.r.tm = r[6];
    g_trace = seq + i * cap; g_trace_len = 0; g_trace_cap = cap;
    Hit h;
    PathRng rng = rng_make(12345, (uint32_t)i, 0);
    double uv[2];
    closest_hit<true>(S, S.prog, S.world_begin, true, rs, r[6], 0.001, RT_INF, h, true, rng, 0, uv);
    len[i] = g_trace_len;
    g_trace = nullptr;
  }
}

// Histogram of unit kinds of the compiled program (16 counters) + total units in counts[16].
void sim_program_stats(const rtx_scene_desc* desc, uint32_t* counts) {
  CompiledProgram prog;
  prog.fuse_sphere_media = g_fuse_sphere_media; prog.draw_first_thin_media = g_draw_first; prog.light_table = g_light_table; prog.defer_last_section = g_defer_last_section; prog.guard_box6 = g_guard_box6; prog.compile(*desc);
  for (int k = 0; k < 17; ++k) counts[k] = 0;
  for (size_t i = 0; i < prog.units.size();) {
    const uint32_t kind = RTX_NODE_KIND((uint32_t)prog.units[i]);
    counts[kind] += 1;
    i += prog_record_units(prog.units[i]);
  }
  counts[16] = (uint32_t)prog.units.size();
}

// walk_flat against walk_program: the closest hit of every ray through the stages the wavefront pipeline walks (head
// section by walk_program up to its end, the flat stretch by walk_flat, the deferred section as a call) against the
// undivided walk, media included (free paths addressed by the ray's number).  Returns the number of rays whose hit
// (t, kind, context, stale (u, v) record) differs; -1 when the program has no flat stretch.
long sim_check_flat_walk(const rtx_scene_desc* desc, const double* rays, size_t n, uint32_t bounce, size_t* n_hits, size_t* n_media_hits) {
  CompiledProgram prog;
  prog.fuse_sphere_media = g_fuse_sphere_media; prog.draw_first_thin_media = g_draw_first; prog.light_table = g_light_table; prog.defer_last_section = g_defer_last_section; prog.guard_box6 = g_guard_box6; prog.compile(*desc);
  const Scene S = prog.host_view(*desc);
  if (!S.main_flat || prog.n_general_media != 0) return -1;
  long bad = 0;
  size_t hits = 0, media = 0;
  const uint32_t none = (uint32_t)RTX_NODE_MAKE(RTX_NODE_NONE, 0);
  for (size_t k = 0; k < n; ++k) {
    const double* r = rays + 7 * k;
    RayRegs rs;
    rs.r.o = v3(r[0], r[1], r[2]); rs.r.d = v3(r[3], r[4], r[5]); rs.r.tm = r[6];
    const PathRng rng = rng_make(991, (uint32_t)k, 3);
    Hit a, b, hh;
    double uva[2] = {0.0, 0.0}, uvb[2] = {0.0, 0.0};
    closest_hit<true>(S, S.prog, S.world_begin, true, rs, r[6], 0.001, RT_INF, a, true, rng, bounce, uva);
    hh.t = RT_INF; hh.kind_idx = none; hh.ctx = 0;
    if (S.flat_begin != S.world_begin) walk_program<MEDIA_NONE, true, true>(S, S.prog, S.world_begin, true, rs, r[6], 0.001, RT_INF, hh, false, rng, bounce, nullptr, none, 0u, false, V3(), V3(), S.flat_begin - 1u);
    const bool deferred = walk_flat<MEDIA_SPHERES>(S, S.prog, S.flat_begin, true, rs, r[6], 0.001, hh.t, b, true, rng, bounce, uvb,
                                                   hh.kind_idx, hh.ctx, false, V3(), V3());
    if (deferred && S.tail_begin) {
      Hit h2;
      walk_program<MEDIA_NONE, false>(S, S.prog, S.tail_begin, true, rs, r[6], 0.001, b.t, h2, false, rng, bounce, uvb);
      if (hit_any(h2)) b = h2;
    }
    const bool same = memcmp(&a.t, &b.t, 8) == 0 && a.kind_idx == b.kind_idx && (!hit_any(a) || a.ctx == b.ctx) && memcmp(uva, uvb, 16) == 0;
    if (!same) ++bad;
    if (hit_any(a)) ++hits;
    if (RTX_NODE_KIND(a.kind_idx) == RTX_NODE_MEDIUM) ++media;
  }
  if (n_hits) *n_hits = hits;
  if (n_media_hits) *n_media_hits = media;
  return bad;
}

// Layout of the compiled program: units of the program, head_end, tail_begin, units with the tables behind it,
// general media, head/tail nearer-child-first depths, light table nodes.
void sim_program_layout(const rtx_scene_desc* desc, uint32_t* out8) {
  CompiledProgram prog;
  prog.fuse_sphere_media = g_fuse_sphere_media; prog.draw_first_thin_media = g_draw_first; prog.light_table = g_light_table; prog.defer_last_section = g_defer_last_section; prog.guard_box6 = g_guard_box6; prog.compile(*desc);
  out8[0] = prog.prog_units; out8[1] = prog.head_end; out8[2] = prog.tail_begin; out8[3] = (uint32_t)prog.units.size();
  out8[4] = prog.n_general_media; out8[5] = prog.head_near_depth; out8[6] = prog.tail_near_depth; out8[7] = (uint32_t)prog.light_nodes.size();
}

void sim_resolve_rgb8(const double* accum, int w, int h, int spp, uint8_t* rgb) {
  double scale = 1.0 / (double)spp;
  for (size_t i = 0; i < (size_t)w * (size_t)h * 3; ++i) rgb[i] = resolve_channel(accum[i], scale);
}

// Probes of single device functions for unit tests.
double sim_rng_double(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t n) {
  PathRng r = rng_make(seed, pixel, sample);
  r.stream = stream; r.n = n;
  return rng_next(r);
}
void sim_philox(const uint32_t* c, const uint32_t* k, uint32_t* out) { philox4x32(c[0], c[1], c[2], c[3], k[0], k[1], out); }
void sim_philox10(const uint32_t* c, const uint32_t* k, uint32_t* out) { philox4x32_r<10>(c[0], c[1], c[2], c[3], k[0], k[1], out); }
void sim_texture_value(const rtx_scene_desc* desc, uint32_t tex, const double* uvp, size_t n, double* out) {
  CompiledProgram prog;
  prog.fuse_sphere_media = g_fuse_sphere_media; prog.draw_first_thin_media = g_draw_first; prog.light_table = g_light_table; prog.defer_last_section = g_defer_last_section; prog.guard_box6 = g_guard_box6; prog.compile(*desc);
  const Scene view = prog.host_view(*desc);
  const Scene* S = &view;
  for (size_t i = 0; i < n; ++i) {
    V3 c = texture_value(*S, tex, uvp[5 * i], uvp[5 * i + 1], v3(uvp + 5 * i + 2));
    out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
  }
}
void sim_lights_pdf(const rtx_scene_desc* desc, const double* origin, const double* dirs, size_t n, double* out) {
  CompiledProgram prog;
  prog.fuse_sphere_media = g_fuse_sphere_media; prog.draw_first_thin_media = g_draw_first; prog.light_table = g_light_table; prog.defer_last_section = g_defer_last_section; prog.guard_box6 = g_guard_box6; prog.compile(*desc);
  const Scene view = prog.host_view(*desc);
  const Scene* S = &view;
  uint64_t total = 0;
  for (size_t i = 0; i < n; ++i) { uint32_t pb = 0; out[i] = lights_pdf_value(*S, v3(origin), v3(dirs + 3 * i), pb); total += pb; }
  g_last_probes = total;
}
uint64_t sim_last_probes(void) { return g_last_probes; }   // lights evaluated by the last sim_lights_pdf call
void sim_lights_random(const rtx_scene_desc* desc, const double* origin, uint64_t seed, size_t n, double* out) {
  CompiledProgram prog;
  prog.fuse_sphere_media = g_fuse_sphere_media; prog.draw_first_thin_media = g_draw_first; prog.light_table = g_light_table; prog.defer_last_section = g_defer_last_section; prog.guard_box6 = g_guard_box6; prog.compile(*desc);
  const Scene view = prog.host_view(*desc);
  const Scene* S = &view;
  for (size_t i = 0; i < n; ++i) {
    PathRng r = rng_make(seed, (uint32_t)i, 0);
    V3 d = lights_random(*S, v3(origin), r);
    out[3 * i] = d.x; out[3 * i + 1] = d.y; out[3 * i + 2] = d.z;
  }
}
// ---- replay probes: the caller's numbers in place of the path RNG (RngTape, rt_rng.h) ----
// One camera path of the device core: camera_ray for (i, j, sample_idx) when ray7 is null, else trace_ray on the given
// ray.  seq: the numbers of the camera and shading draws in the order the device core makes them; roulette: one
// number per bounce; medium: n_bounces x n_media numbers.  rays_out: the ray of every segment (7 doubles each, at
// most max_rays).  Returns the number of draws asked for beyond what was supplied.
struct SimScene { CompiledProgram prog; Scene view; };
void* sim_scene_new(const rtx_scene_desc* desc) {      // the compiled program of a description, kept for many probe calls
  SimScene* h = new SimScene();
  h->prog.fuse_sphere_media = g_fuse_sphere_media; h->prog.draw_first_thin_media = g_draw_first; h->prog.light_table = g_light_table; h->prog.defer_last_section = g_defer_last_section; h->prog.guard_box6 = g_guard_box6;
  h->prog.compile(*desc);
  h->view = h->prog.host_view(*desc);
  return h;
}
void sim_scene_free(void* h) { delete (SimScene*)h; }
long sim_tape_path(void* scene, const rtx_camera_desc* cam, int use_lights, int i, int j, int sample_idx,
                   const double* ray7, const double* seq, size_t n_seq, const double* roulette, uint32_t n_bounces,
                   const double* medium, uint32_t n_media, size_t* consumed, double* rays_out, size_t max_rays,
                   size_t* n_rays, double* color, uint32_t* counters) {
  const Scene& S = ((SimScene*)scene)->view;
  RngTape tape{seq, n_seq, 0, 0, roulette, n_bounces, medium, n_media};
  std::vector<double> log;
  rt_rng_tape = &tape; g_ray_log = &log;
  PathRng rng = rng_make(1, 0, 0);
  Ray ray;
  if (ray7) { ray.o = v3(ray7[0], ray7[1], ray7[2]); ray.d = v3(ray7[3], ray7[4], ray7[5]); ray.tm = ray7[6]; }
  else ray = camera_ray(*cam, i, j, sample_idx, rng);
  uint32_t sg = 0, pb = 0;
  const V3 c = trace_ray(S, *cam, use_lights != 0 && S.has_lights != 0, ray, rng, sg, pb);
  rt_rng_tape = nullptr; g_ray_log = nullptr;
  if (consumed) *consumed = tape.pos;
  const size_t nr = log.size() / 7;
  if (n_rays) *n_rays = nr;
  if (rays_out) memcpy(rays_out, log.data(), sizeof(double) * 7 * std::min(nr, max_rays));
  if (color) { color[0] = c.x; color[1] = c.y; color[2] = c.z; }
  if (counters) { counters[0] = sg; counters[1] = pb; }
  return (long)tape.overrun;
}
// The random vectors of rt_rng.h from a tape: kind 0 random_in_unit_sphere, 1 random_unit_vector,
// 2 random_cosine_direction, 3 random_in_unit_disk.
long sim_tape_vec(int kind, const double* seq, size_t n_seq, double* out3, size_t* consumed) {
  RngTape tape{seq, n_seq, 0, 0, nullptr, 0, nullptr, 0};
  rt_rng_tape = &tape;
  PathRng rng = rng_make(1, 0, 0);
  const V3 v = kind == 0 ? random_in_unit_sphere(rng) : kind == 1 ? random_unit_vector(rng) : kind == 2 ? random_cosine_direction(rng) : random_in_unit_disk(rng);
  rt_rng_tape = nullptr;
  out3[0] = v.x; out3[1] = v.y; out3[2] = v.z;
  if (consumed) *consumed = tape.pos;
  return (long)tape.overrun;
}
// lights_random(origin) from a tape.
long sim_tape_lights_random(const rtx_scene_desc* desc, const double* origin, const double* seq, size_t n_seq, double* out3, size_t* consumed) {
  CompiledProgram prog;
  prog.compile(*desc);
  const Scene S = prog.host_view(*desc);
  RngTape tape{seq, n_seq, 0, 0, nullptr, 0, nullptr, 0};
  rt_rng_tape = &tape;
  PathRng rng = rng_make(1, 0, 0);
  const V3 v = lights_random(S, v3(origin), rng);
  rt_rng_tape = nullptr;
  out3[0] = v.x; out3[1] = v.y; out3[2] = v.z;
  if (consumed) *consumed = tape.pos;
  return (long)tape.overrun;
}
void sim_camera_ray(const rtx_camera_desc* cam, int i, int j, int sample_idx, uint64_t seed, double* out7) {
  PathRng r = rng_make(seed, (uint32_t)(j * cam->image_width + i), (uint32_t)sample_idx);
  Ray ray = camera_ray(*cam, i, j, sample_idx, r);
  out7[0] = ray.o.x; out7[1] = ray.o.y; out7[2] = ray.o.z; out7[3] = ray.d.x; out7[4] = ray.d.y; out7[5] = ray.d.z; out7[6] = ray.tm;
}

}  // extern "C"
