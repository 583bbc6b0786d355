// Device layer of the MI355X path-tracing backend: HIP kernels for gfx950 and
// the C ABI of include/rtx.h.  There is no CPU path in this file: every
// compute entry point needs a HIP device and fails with RTX_ERR_NO_DEVICE
// otherwise.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/rtx.h"
#include "device_internal.h"

#if defined(RT_PROF)
// Tuning build only (make prof): cycles, trips and lanes per class of the traversal loop.  Wave leaders accumulate
// in LDS; the traversal kernel flushes the block's sums to g_prof once, when it ends.
__device__ unsigned long long g_prof[64];
__device__ unsigned long long g_sprof[32];
__device__ unsigned long long g_wprof[32];   // folded-table walk: trips, lanes waiting and served per kind of step, cycles
__shared__ unsigned long long g_prof_lds[16 * 30];
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_PROF_START() long long prof_t0 = clock64(); int prof_lanes = 0; (void)prof_lanes
#define RT_PROF_LANES(n) prof_lanes = (n)
#define RT_PROF_TRIP(slot)                                                       \
  do {                                                                           \
    const long long t1_ = clock64();                                             \
    if ((threadIdx.x & 63u) == 0) {                                              \
      unsigned long long* p_ = g_prof_lds + (threadIdx.x >> 6) * 30;             \
      p_[(slot)] += (unsigned long long)(t1_ - prof_t0);                         \
      p_[10 + (slot)] += 1ull;                                                   \
      p_[20 + (slot)] += (unsigned long long)prof_lanes;                         \
    }                                                                            \
    prof_t0 = clock64();                                                         \
  } while (0)
// folded-table walk (walk_stream_wide): sums per trip kept in registers, added up by lane 0 of every wave at the end
#define RT_WPROF_DECL() unsigned long long wp_acc[20] = {0}
#define RT_WPROF(slot, value) wp_acc[(slot)] += (unsigned long long)(value)
#define RT_WPROF_CLOCK() clock64()
#define RT_WPROF_FLUSH()                                                                              \
  do {                                                                                                \
    if ((threadIdx.x & 63u) == 0)                                                                     \
      for (int k_ = 0; k_ < 20; ++k_) atomicAdd(&g_wprof[k_], wp_acc[k_]);                            \
  } while (0)
// shading kernel: sections of shade_bounce, measured by the first active lane of the wave, one block in eight
#define RT_SPROF_START() long long sprof_t0 = clock64()
#define RT_SPROF(slot)                                                           \
  do {                                                                           \
    const long long t1_ = clock64();                                             \
    const unsigned long long m_ = __ballot(1);                                   \
    if ((blockIdx.x & 7u) == 0 && (int)(threadIdx.x & 63u) == __ffsll((long long)m_) - 1) { \
      atomicAdd(&g_sprof[(slot)], (unsigned long long)(t1_ - sprof_t0));         \
      atomicAdd(&g_sprof[16 + (slot)], 1ull);                                    \
    }                                                                            \
    sprof_t0 = clock64();                                                        \
  } while (0)
#endif
#endif
#if defined(RT_PROF) && defined(__HIP_DEVICE_COMPILE__)
// tuning build: cycles of a phase of the shading kernel, per wave, one block in eight
#define RT_KPROF_T(name) const long long name = clock64()
#define RT_KPROF_ADD(slot, t0)                                                                        \
  do {                                                                                                \
    if ((blockIdx.x & 7u) == 0 && (threadIdx.x & 63u) == 0) {                                         \
      atomicAdd(&g_sprof[(slot)], (unsigned long long)(clock64() - (t0)));                            \
      atomicAdd(&g_sprof[16 + (slot)], 1ull);                                                         \
    }                                                                                                 \
  } while (0)
#else
#define RT_KPROF_T(name) do { } while (0)
#define RT_KPROF_ADD(slot, t0) do { } while (0)
#endif
#include "core/rt_program.h"
#include "core/rt_shade.h"

using namespace rt;

// ================================================================ kernels ===

#ifndef RT_SHADE_ATTR
#define RT_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(2)))  // at least two waves per SIMD (up to 256 registers); it needs 160, so three run
#endif
#ifndef RT_TRAV_THREADS
#define RT_TRAV_THREADS 1024  // threads of a traversal workgroup (one workgroup per CU)
#endif

// ------------------------------------------------------- wavefront state ---
// Paths in flight live in structure-of-arrays queues so that every load and store of a wave is one contiguous run.
// f[k*cap + slot], k = 0..9: ray origin (3), ray direction (3), ray time, throughput (3).
// u[k*cap + slot], k = 0..2: pixel index, sample index, remaining depth << 16 | bounce.
struct WfQueue {
  double* f;
  uint32_t* u;
};
struct WfHits {        // output of the traversal kernel, input of the shading kernel
  double* t;           // hit parameter
  uint32_t* kind_idx;  // RTX_NODE_MAKE(kind, index); kind NONE = miss
  uint32_t* ctx;       // transform context of the hit
  double* uv;          // [2*cap]: the stale record (u,v) of a medium hit whose phase texture reads them; rarely written
};
// Every counter that a whole launch hammers with atomics sits in a 128-byte line of its own: atomics on one line are
// served one after the other by the L2, whatever word of the line they address, and at tens of millions per second
// that queue, not the arithmetic, sets the pace of a launch (measured: one more atomic per 256 shaded entries on the
// line of n_out tripled the duration of the shading launch).
struct WfCounters {
  uint32_t n_in;        // paths in the queue the traversal/shading kernels read
  uint32_t gen_count;   // camera paths to start in this iteration
  uint32_t gen_slot;    // queue slot of the first of them (they follow the survivors)
  uint32_t gen_sample;  // sample offset of the first new path
  uint32_t gen_rem;     // its pixel rank within that sample
  uint32_t finished;    // set by k_wf_finish: the survivors have been run to their end
  unsigned long long next_path, total_paths;
  alignas(128) uint32_t n_out;      // queue slots the shading kernel has given to (possible) survivors
  alignas(128) uint32_t n_defer;    // rays of this iteration that entered the deferred section
  alignas(128) uint32_t work_main;  // next queue entry to hand out in the world-program traversal launch
  alignas(128) uint32_t work_tail;  // ... in the deferred-section launch
  alignas(128) uint32_t work_shade; // ... in the shading launch
  alignas(128) uint32_t work_head;  // ... in the head-section launch
  alignas(128) uint32_t work_finish; // survivors the finishing kernel's workgroups have taken beyond their first ones
  // Device-side timing of the world-program traversal launches (wall_clock64 ticks): first block in, last block out
  // of the current launch; sum and number of the launches completed so far.  HIP events cannot time a kernel while
  // other pipelines share the GPU: they would include the wait for a free CU.
  alignas(128) unsigned long long k1_start, k1_end, k1_sum, k1_count;
  alignas(128) unsigned long long sh_start, sh_end, sh_sum, sh_count;  // the same for the shading launches
};
__device__ __forceinline__ void wf_flush_timing(WfCounters* c) {  // one thread, between launches
  if (c->k1_end > c->k1_start) { c->k1_sum += c->k1_end - c->k1_start; c->k1_count += 1; }
  c->k1_start = ~0ull; c->k1_end = 0ull;
  if (c->sh_end > c->sh_start) { c->sh_sum += c->sh_end - c->sh_start; c->sh_count += 1; }
  c->sh_start = ~0ull; c->sh_end = 0ull;
}
struct WfFrame {       // what a path index means: (sample offset, pixel rank); pixels of the rectangle in 8x8 tiles
  int x0, y0, w, h, sample_begin;
  uint32_t per_sample;  // w * h
};

__global__ void k_wf_begin(WfCounters* c, unsigned long long total) {
  c->n_in = 0; c->n_out = 0; c->gen_count = 0; c->gen_slot = 0; c->gen_sample = 0; c->gen_rem = 0;
  c->finished = 0; c->n_defer = 0; c->work_main = 0; c->work_tail = 0; c->work_shade = 0; c->work_head = 0; c->work_finish = 0;
  c->next_path = 0; c->total_paths = total;
  c->k1_start = ~0ull; c->k1_end = 0ull; c->k1_sum = 0ull; c->k1_count = 0ull;
  c->sh_start = ~0ull; c->sh_end = 0ull; c->sh_sum = 0ull; c->sh_count = 0ull;
}

// One thread: decide how many new camera paths top the queue up and flip the in/out roles.  New paths take the
// slots behind the survivors, so the ray generation kernel needs no atomics.
__global__ void k_wf_prepare(WfCounters* c, uint32_t cap, uint32_t per_sample, unsigned long long* __restrict__ stats, int count_segments) {
  const uint32_t survivors = c->finished ? 0u : c->n_out;
  c->finished = 0;
  wf_flush_timing(c);
  const unsigned long long remaining = c->total_paths - c->next_path;
  const unsigned long long room = cap - survivors;
  uint32_t g = (uint32_t)(remaining < room ? remaining : room);
  if ((unsigned long long)g < remaining) g &= ~63u;  // whole waves of new paths, except for the last ones
  c->gen_count = g; c->gen_slot = survivors;
  c->gen_sample = (uint32_t)(c->next_path / per_sample); c->gen_rem = (uint32_t)(c->next_path % per_sample);
  c->next_path += g;
  c->n_in = survivors + g;
  c->n_out = 0;
  c->n_defer = 0; c->work_main = 0; c->work_tail = 0; c->work_shade = 0; c->work_head = 0; c->work_finish = 0;
  if (g) atomicAdd(&stats[0], (unsigned long long)g);  // two pipelines may be preparing at the same time
  if (count_segments && survivors + g) atomicAdd(&stats[1], (unsigned long long)(survivors + g));  // no world-program launch will
}

// Ray generation (Camera::get_ray, camera.rs:133-164): path index -> (sample, pixel), pixels enumerated tile by tile
// (8 x 8, clipped at the rectangle's edges) so that the 64 lanes of a wave start the same stratified sample of
// neighbouring pixels.  Path k of the iteration goes to slot gen_slot + k.
// Path k of an iteration's new camera paths -> pixel (i, j) and sample index s.
__device__ __forceinline__ void wf_path_coords(const rtx_camera_desc& cam, const WfFrame& fr, uint32_t s0, uint32_t r0, uint32_t k,
                                               int& i, int& j, int& s) {
  const uint32_t W = (uint32_t)fr.w, H = (uint32_t)fr.h;
  const uint32_t r = r0 + k;                         // < per_sample + capacity, fits 32 bits (checked by the host)
  const uint32_t so = r / fr.per_sample, rem = r - so * fr.per_sample;
  const uint32_t row_px = W * 8u;
  const uint32_t ty = rem / row_px, r2 = rem - ty * row_px;
  const uint32_t hy = H - 8u * ty < 8u ? H - 8u * ty : 8u;
  const uint32_t tile_px = 8u * hy;
  uint32_t tx = r2 / tile_px;
  if (tx > (W - 1u) / 8u) tx = (W - 1u) / 8u;
  const uint32_t r3 = r2 - tx * tile_px;
  const uint32_t wx = W - 8u * tx < 8u ? W - 8u * tx : 8u;
  const uint32_t ly = r3 / wx, lx = r3 - ly * wx;
  i = fr.x0 + (int)(8u * tx + lx); j = fr.y0 + (int)(8u * ty + ly);
  s = fr.sample_begin + (int)(s0 + so);
  (void)cam;
}
// The queue record of a new camera path (throughput 1, full depth, bounce 0).
__device__ __forceinline__ void wf_store_new_path(const WfQueue& q, uint32_t cap, uint32_t slot, const Ray& ray, uint32_t pixel, int s, int max_depth) {
  double* f = q.f + slot;
  f[0 * (size_t)cap] = ray.o.x; f[1 * (size_t)cap] = ray.o.y; f[2 * (size_t)cap] = ray.o.z;
  f[3 * (size_t)cap] = ray.d.x; f[4 * (size_t)cap] = ray.d.y; f[5 * (size_t)cap] = ray.d.z;
  f[6 * (size_t)cap] = ray.tm;
  f[7 * (size_t)cap] = 1.0; f[8 * (size_t)cap] = 1.0; f[9 * (size_t)cap] = 1.0;
  uint32_t* u = q.u + slot;
  u[0] = pixel; u[(size_t)cap] = (uint32_t)s; u[2 * (size_t)cap] = (uint32_t)max_depth << 16;
}
__global__ __launch_bounds__(256) void k_wf_raygen(rtx_camera_desc cam, WfFrame fr, unsigned long long seed, uint32_t cap,
                                                   const WfCounters* c, WfQueue q) {
  const uint32_t count = c->gen_count, slot0 = c->gen_slot, s0 = c->gen_sample, r0 = c->gen_rem;
  for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < count; idx += gridDim.x * blockDim.x) {
    int i, j, s;
    wf_path_coords(cam, fr, s0, r0, idx, i, j, s);
    const uint32_t pixel = (uint32_t)(j * cam.image_width + i);
    PathRng rng = rng_make(seed, pixel, (uint32_t)s);
    Ray ray = camera_ray(cam, i, j, s, rng);
    wf_store_new_path(q, cap, slot0 + idx, ray, pixel, s, cam.max_depth);
  }
}

// Traversal kernels: one closest-hit query per queued path (world.hit, camera.rs:179).  Persistent 1024-thread
// workgroups, one per CU: each stages its part of the program image in LDS once, then walks its share of the queue;
// only the ray and hit streams (and the rare general primitive records) touch global memory.
//
// TAIL = false walks the world program.  When the program has a deferred section (Scene::tail_begin), a ray that
// passes the section's root box is only noted: its queue index goes to the defer list (collected per wave in LDS,
// flushed 64 at a time with one atomic).  TAIL = true then walks the section's own program for the rays of that
// list, densely packed, starting from the hit found so far.  Together the two are the undivided walk.
#ifndef RT_FEED_BATCH
#define RT_FEED_BATCH 512u   // the same for the waves of a stream walk
#endif
#ifndef RT_DEFER_FLUSH
#define RT_DEFER_FLUSH 64u   // entries of the defer list a wave appends per atomic (a multiple of 64)
#endif
#define RT_DEFER_SLOTS (RT_DEFER_FLUSH + 64u)  // per wave: fewer than RT_DEFER_FLUSH waiting entries + 64 new ones
#ifndef RT_WAVE_BATCH
#define RT_WAVE_BATCH 512u   // most queue entries a wave takes per atomic (a multiple of 64; measured on the final
                             // scene: 128: 1793, 256: 1900, 512: 1913, 1024: 1909 Msamples/s)
#endif
//
// MM (MEDIA_NONE / MEDIA_SPHERES / MEDIA_ALL) is what the world program needs for its media: a program without
// general MEDIUM units runs the variant without the boundary-program machinery, which saves it registers.
template <bool USE_LDS, bool TAIL, int MM, bool GEN = false, bool FLAT = false>   // FLAT: the variant for a program whose stretch behind the head section is swept in step (Scene::main_flat)
__global__ __launch_bounds__(RT_TRAV_THREADS) void k_wf_traverse(Scene S, uint32_t cap, unsigned long long seed, WfCounters* c,
                                                      WfQueue q, WfHits h, uint32_t* __restrict__ defer_list,
                                                      unsigned long long* __restrict__ stats, rtx_camera_desc cam, WfFrame fr) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lds_prog[];
  const uint32_t n = TAIL ? c->n_defer : c->n_in;
  if (blockIdx.x * blockDim.x >= n) return;
  if (!TAIL && threadIdx.x == 0) atomicMin(&c->k1_start, (unsigned long long)wall_clock64());
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  // With a head section (walked by k_wf_stream before this launch) the world program is entered behind it, with the
  // hit the head section found; media boundary programs sit in front of the world program, so a scene with general
  // media still stages the image from its start.
  const bool behind_head = !TAIL && S.head_end != 0;
  // ... except for the new camera paths of the iteration when S.fresh_in_main is set: they sit behind the survivors
  // (from gen_slot on), have not been through the head stream and are walked here from the start of the program
  const bool fresh_whole = behind_head && S.fresh_in_main != 0;
  // S.fresh_in_main == 2: no ray generation launch has run; this kernel makes the camera rays of the new paths itself
  // (Camera::get_ray), stores their queue records for the kernels that follow and walks them from registers
  const bool fresh_gen = GEN && !TAIL && S.fresh_in_main == 2u;   // GEN: the variant compiled with that code
  const uint32_t fresh_from = (fresh_whole || fresh_gen) ? c->gen_slot : 0xFFFFFFFFu;
  const uint32_t gen_s0 = fresh_gen ? c->gen_sample : 0u, gen_r0 = fresh_gen ? c->gen_rem : 0u;
  const uint32_t first_unit = TAIL ? S.tail_begin : ((behind_head && !fresh_whole && MM != MEDIA_ALL) ? ((S.head_end + 1u) & ~1u) : 0u);
  const uint32_t end_unit = (!TAIL && S.tail_begin) ? S.tail_begin : S.n_units;
  const uint64_t* prog = S.prog;
#if defined(RT_PROF)
  for (uint32_t k = threadIdx.x; k < 16 * 30; k += blockDim.x) g_prof_lds[k] = 0;
  __syncthreads();
#endif
  if (USE_LDS) {
    const uint4* src = (const uint4*)(S.prog + first_unit);   // both bounds are even: 16-byte pieces
    uint4* dst = (uint4*)lds_prog;
    for (uint32_t k = threadIdx.x; k < (end_unit - first_unit) / 2; k += blockDim.x) dst[k] = src[k];
    __syncthreads();
    prog = lds_prog - first_unit;                              // unit indices stay those of the whole image
  }
  // per-wave collection buffer of deferred queue indices, behind the staged program
  uint32_t* wave_defer = (uint32_t*)(lds_prog + (USE_LDS ? end_unit - first_unit : 0u)) + wave * RT_DEFER_SLOTS;
  uint32_t n_wait = 0;  // entries waiting in wave_defer (the same in every lane)
  const bool collect = !TAIL && S.tail_begin != 0;
  if (!TAIL && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[1], (unsigned long long)n);
  // Work is handed out dynamically, RT_WAVE_BATCH queue entries per wave at a time (one atomic each): rays differ a
  // lot in cost, and a static split leaves CUs idle at the end of the launch.  The trip count is the same for every
  // lane of a wave: lanes past the end of the queue only vote.
  uint32_t* work = TAIL ? &c->work_tail : &c->work_main;
  const uint32_t n_waves = gridDim.x * (blockDim.x / 64u);
  uint32_t seen = 0;  // the counter as this wave last saw it: batches shrink as the queue runs out (guided scheduling)
  for (;;) {
    uint32_t batch = ((n - seen) / (2u * n_waves)) & ~63u;
    batch = batch < 64u ? 64u : (batch > RT_WAVE_BATCH ? RT_WAVE_BATCH : batch);
    uint32_t wbase = 0;
    if (lane == 0) wbase = atomicAdd(work, batch);
    wbase = __shfl(wbase, 0);
    if (wbase >= n) break;
    seen = wbase;
  for (uint32_t pass = 0; pass < batch && wbase + pass < n; pass += 64u) {
    RT_KPROF_T(kp_pass);
    const uint32_t k = wbase + pass + lane;
    const bool active = k < n;
    const uint32_t idx = TAIL ? defer_list[active ? k : 0] : (active ? k : 0);
    RayMem rs;
    rs.f = q.f + idx; rs.cap = cap;
    const bool gen = GEN && !TAIL && fresh_gen && active && idx >= fresh_from;
    Ray nray;
    nray.o = v3(0, 0, 0); nray.d = v3(0, 0, 1); nray.tm = 0.0;
    PathRng rng = rng_make(seed, 0, 0);
    uint32_t bounce = 0;
    if (gen) {
      int pi, pj, ps;
      wf_path_coords(cam, fr, gen_s0, gen_r0, idx - fresh_from, pi, pj, ps);
      const uint32_t pixel = (uint32_t)(pj * cam.image_width + pi);
      rng = rng_make(seed, pixel, (uint32_t)ps);
      PathRng crng = rng;
      nray = camera_ray(cam, pi, pj, ps, crng);
      wf_store_new_path(q, cap, idx, nray, pixel, ps, cam.max_depth);
    }
    double tm = nray.tm;
    if (!gen) tm = rs.f[6 * (size_t)cap];
    if (!TAIL && MM != MEDIA_NONE && !gen) {
      const uint32_t* u = q.u + idx;
      rng = rng_make(seed, u[0], u[(size_t)cap]);
      bounce = u[2 * (size_t)cap] & 0xFFFFu;
    }
    Hit hit;
    if (TAIL) {
      // the section's program never holds media (the program compiler checks), so no RNG is needed here
      const double t_so_far = h.t[idx];
      walk_program<MEDIA_NONE, false>(S, prog, S.tail_begin, active, rs, tm, 0.001, t_so_far, hit, false, rng, bounce, h.uv + 2 * (size_t)idx);
      if (active && hit_any(hit)) { h.t[idx] = hit.t; h.kind_idx[idx] = hit.kind_idx; h.ctx[idx] = hit.ctx; }
    } else {
      const bool behind = behind_head && idx < fresh_from;
      const double t0 = behind ? h.t[idx] : RT_INF;
      RT_KPROF_ADD(14, kp_pass);
      const uint32_t kind0 = behind ? h.kind_idx[idx] : (uint32_t)RTX_NODE_MAKE(RTX_NODE_NONE, 0), ctx0 = behind ? h.ctx[idx] : 0u;
      bool deferred;
      if (FLAT && MM != MEDIA_ALL) {
        // the stretch behind the head section is a list every ray passes unit by unit: the wave sweeps it in step
        // (walk_flat); lanes that have not been through the head section (new camera paths) walk that first
        double t1 = t0;
        uint32_t kind1 = kind0, ctx1 = ctx0;
        if (S.flat_begin != S.world_begin) {   // (whether or not that section was streamed for the survivors: S.head_end)
          const bool fresh = active && !behind;
          if (RT_WAVE_ANY(fresh)) {
            Hit hh;
            walk_program<MEDIA_NONE, true, true>(S, prog, S.world_begin, fresh, rs, tm, 0.001, RT_INF, hh, false, rng, bounce, nullptr,
                                           (uint32_t)RTX_NODE_MAKE(RTX_NODE_NONE, 0), 0u, gen, nray.o, nray.d, S.flat_begin - 1u);
            if (fresh) { t1 = hh.t; kind1 = hh.kind_idx; ctx1 = hh.ctx; }
          }
        }
        deferred = walk_flat<MM>(S, prog, S.flat_begin, active, rs, tm, 0.001, t1, hit, MM != MEDIA_NONE,
                                 rng, bounce, h.uv + 2 * (size_t)idx, kind1, ctx1, gen, nray.o, nray.d);
      } else {
        deferred = walk_program<MM, true>(S, prog, behind ? S.head_end + 1u : S.world_begin, active, rs, tm, 0.001, t0, hit,
                                          MM != MEDIA_NONE, rng, bounce, h.uv + 2 * (size_t)idx, kind0, ctx0, gen, nray.o, nray.d);
      }
      if (active) { h.t[idx] = hit.t; h.kind_idx[idx] = hit.kind_idx; h.ctx[idx] = hit.ctx; }
      RT_KPROF_ADD(11, kp_pass);
      if (collect) {
        const unsigned long long m = __ballot(active && deferred);
        if (active && deferred) wave_defer[n_wait + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = idx;
        n_wait += (uint32_t)__popcll(m);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (n_wait >= RT_DEFER_FLUSH) {                        // flush RT_DEFER_FLUSH entries with one atomic
          uint32_t at = 0;
          if (lane == 0) at = atomicAdd(&c->n_defer, RT_DEFER_FLUSH);
          at = __shfl(at, 0);
          for (uint32_t k = 0; k < RT_DEFER_FLUSH; k += 64u) defer_list[at + k + lane] = wave_defer[k + lane];
          const uint32_t rest = lane < n_wait - RT_DEFER_FLUSH ? wave_defer[RT_DEFER_FLUSH + lane] : 0u;   // fewer than 64 are left
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          if (lane < n_wait - RT_DEFER_FLUSH) wave_defer[lane] = rest;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          n_wait -= RT_DEFER_FLUSH;
        }
      }
    }
  }
  }
  if (collect && n_wait) {
    uint32_t at = 0;
    if (lane == 0) at = atomicAdd(&c->n_defer, n_wait);
    at = __shfl(at, 0);
    if (lane < n_wait) defer_list[at + lane] = wave_defer[lane];
  }
  if (!TAIL) {
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&c->k1_end, (unsigned long long)wall_clock64());
  }
#if defined(RT_PROF)
  __syncthreads();
  if (threadIdx.x < 30) {
    unsigned long long sum = 0;
    for (uint32_t w = 0; w < blockDim.x / 64; ++w) sum += g_prof_lds[w * 30 + threadIdx.x];
    atomicAdd(&g_prof[(threadIdx.x / 10) * 16 + threadIdx.x % 10], sum);
  }
#endif
}

// A section of the program walked as a stream (walk_stream): the head section of the world program for every queued
// path (TAIL = false; it leaves its closest hit in the hit arrays, where the world-program launch picks it up), or the
// deferred section for the rays of the defer list (TAIL = true; it starts from the hit found so far).  Queue entries
// are taken in per-wave batches (one atomic each, shrinking towards the end of the queue) and handed to lanes as they
// run out of work.
struct QueueFeed {
  const double* qf; size_t cap;
  const uint32_t* list;            // defer list, or null: entry k is path k
  double* ht; uint32_t* hk; uint32_t* hc;
  uint32_t* work; uint32_t n, n_waves, lane;
  uint32_t wnext, wend, seen;      // this wave's batch [wnext, wend) and the queue counter as last seen
  bool more_, from_hits;
  __device__ bool more() const { return more_; }
  __device__ bool take(bool want, uint32_t& id) {
    if (wnext >= wend) {
      uint32_t batch = ((n - seen) / (2u * n_waves)) & ~63u;
      batch = batch < 64u ? 64u : (batch > RT_FEED_BATCH ? RT_FEED_BATCH : batch);
      uint32_t b = 0;
      if (lane == 0) b = atomicAdd(work, batch);
      b = __shfl(b, 0);
      seen = b < n ? b : n;
      if (b >= n) { more_ = false; return false; }
      wnext = b; wend = b + batch < n ? b + batch : n;
    }
    const unsigned long long m = __ballot(want);
    const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    const uint32_t avail = wend - wnext, asked = (uint32_t)__popcll(m);
    const uint32_t k = wnext + rank;
    wnext += asked < avail ? asked : avail;
    if (want && rank < avail) { id = list ? list[k] : k; return true; }
    return false;
  }
  __device__ RayMem ray(uint32_t id) const { RayMem r; r.f = qf + id; r.cap = cap; return r; }
  __device__ double time(uint32_t id) const { return qf[id + 6 * cap]; }
  __device__ void start(uint32_t id, double& tmax, uint32_t& kind, uint32_t& ctx) const {
    if (from_hits) { tmax = ht[id]; kind = hk[id]; ctx = hc[id]; }
    else { tmax = RT_INF; kind = RTX_NODE_MAKE(RTX_NODE_NONE, 0); ctx = 0; }
  }
  __device__ void commit(uint32_t id, double tmax, uint32_t kind, uint32_t ctx) const { ht[id] = tmax; hk[id] = kind; hc[id] = ctx; }
  // A ray the nearer-child-first walk could not vouch for and walked again in the reference's order (statistics only).
  uint32_t redos;
  __device__ void note_redo() { redos += 1u; }
};

template <bool USE_LDS, bool TAIL, int NEAR = 0>   // NEAR: 1 = nearer child first over sphere leaves, 2 = quads in the leaves too
__global__ __launch_bounds__(RT_TRAV_THREADS) void k_wf_stream(Scene S, uint32_t cap, WfCounters* c, WfQueue q, WfHits h,
                                                    const uint32_t* __restrict__ defer_list, unsigned long long* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lds_prog[];
  const uint32_t n = TAIL ? c->n_defer : (S.fresh_in_main ? c->gen_slot : c->n_in);   // the head stream may leave the new paths out
  if (blockIdx.x * blockDim.x >= n) return;
  const uint32_t first_unit = TAIL ? S.tail_begin : 0u;
  const uint32_t end_unit = TAIL ? S.n_units : ((S.head_end + 2u) & ~1u);   // the SYNC that ends the head section included
  const uint64_t* prog = S.prog;
  if (USE_LDS) {
    const uint4* src = (const uint4*)(S.prog + first_unit);
    uint4* dst = (uint4*)lds_prog;
    for (uint32_t k = threadIdx.x; k < (end_unit - first_unit) / 2; k += blockDim.x) dst[k] = src[k];
    __syncthreads();
    prog = lds_prog - first_unit;
  }
  QueueFeed feed;
  feed.qf = q.f; feed.cap = cap; feed.list = TAIL ? defer_list : nullptr;
  feed.ht = h.t; feed.hk = h.kind_idx; feed.hc = h.ctx; feed.redos = 0;
  feed.work = TAIL ? &c->work_tail : &c->work_head; feed.n = n; feed.n_waves = gridDim.x * (blockDim.x / 64u); feed.lane = threadIdx.x & 63u;
  feed.wnext = 0; feed.wend = 0; feed.seen = 0; feed.more_ = true; feed.from_hits = TAIL;
  if (NEAR) {  // the section's box tree nearer child first; the lanes' stacks sit behind the staged program
    uint32_t* stk = (uint32_t*)(lds_prog + (USE_LDS ? end_unit - first_unit : 0u)) + threadIdx.x;
    if (!USE_LDS && !TAIL && NEAR == 2 && S.head_wide_nodes) {
      // a head section walked from global memory through the table that folds two levels of its tree into one record
      walk_stream_wide<true>(S, prog, S.world_begin, feed, stk, blockDim.x, S.prog + S.head_wide_begin);
    } else if (!USE_LDS && !TAIL && NEAR == 2 && S.head_top_nodes) {
      // a head section walked from global memory: the first levels of its tree (heap order, with their unit positions)
      // are staged behind the stacks, so that the walk leaves LDS only below them
      uint64_t* top = lds_prog + (size_t)S.head_near_depth * (blockDim.x / 2u);
      const uint32_t box_units = (S.head_top_nodes * PROG_BOX_UNITS + 1u) & ~1u, all_units = S.head_top_units;
      for (uint32_t k = threadIdx.x; k < all_units; k += blockDim.x) top[k] = S.prog[S.head_top_begin + k];
      __syncthreads();
      walk_stream_near<true, false, 1>(S, prog, S.world_begin, feed, stk, blockDim.x, true, top, (const uint32_t*)(top + box_units), S.head_top_nodes);
    } else {
      walk_stream_near<NEAR == 2, USE_LDS>(S, prog, TAIL ? S.tail_begin : S.world_begin, feed, stk, blockDim.x, !TAIL);
    }
    uint32_t r = feed.redos;   // statistics: one atomic per wave that had any
    for (int off = 32; off > 0; off >>= 1) r += __shfl_down(r, off);
    if ((threadIdx.x & 63u) == 0 && r) atomicAdd(&stats[5], (unsigned long long)r);
  } else {
    walk_stream(S, prog, TAIL ? S.tail_begin : S.world_begin, feed);
  }
}

// Shading kernel: everything ray_color does after world.hit (camera.rs:180-246) for every queued path.  A path that
// ends adds throughput x radiance to its pixel; a path that goes on is written to the other queue.
//
// Sorting.  The branches of shade_bounce differ a lot in cost (a miss or a light ends the path at once, a marble hit
// evaluates seven octaves of noise), and a wave pays for every branch one of its lanes takes.  A workgroup therefore
// takes RT_SHADE_CHUNK (2048) queue entries at a time and first sorts them by a key made of the hit material's class
// (rtx_scene::mat_class) and the kind of primitive: a stable counting sort in LDS, ballots within each wave and a
// small prefix over (key, wave).  Waves then shade runs of equal keys.  Paths are addressed by their own
// coordinates (RNG included), so the order in which they are shaded changes no result.
//
// Queue slots.  The classes that always end a path (miss, light, NoMaterial) sort first; every other entry of the
// chunk may go on, and the chunk takes that many slots of the other queue with ONE atomic (an atomic per 256-entry
// pass on the counter's line is what used to bound this kernel).  The few entries that end after all (an absorbed
// metal reflection, a rejected pdf, the last bounce) leave their slot as a DEAD entry: remaining depth 0 and a
// ray that misses everything at once.  The traversal kernels walk a dead entry like any other; the next sort drops it.
#ifndef RT_SHADE_PER_THREAD
// entries per thread and trip.  Shading kernel per 10 rows of the final scene, three workgroups per CU (160 registers):
// 4: 63.4 ms, 6: 59.3, 8: 59.0, 12: 59.9, 16: 63.8 (52 KB of LDS per workgroup leave the CU no room beside three of them)
#define RT_SHADE_PER_THREAD 8u
#endif
#define RT_SHADE_CHUNK (256u * RT_SHADE_PER_THREAD)  // entries a workgroup sorts and shades per trip
#define RT_SHADE_SLOTS (4u * RT_SHADE_PER_THREAD)    // (wave, pass) pairs of a trip
#define RT_SHADE_KEYS 16u             // material class << 1 | sphere
#ifndef RT_SHADE_LDS_TABLES
#define RT_SHADE_LDS_TABLES 16384     // bytes of LDS for the small scene tables (0 = leave them in global memory)
#endif
#ifndef RT_SHADE_DROP
#define RT_SHADE_DROP 1
#endif
enum { MC_MISS = 0, MC_END = 1, MC_METAL = 2, MC_DIELECTRIC = 3, MC_LAMBERT = 4, MC_LAMBERT_IMAGE = 5, MC_LAMBERT_NOISE = 6, MC_ISOTROPIC = 7 };
__global__ __launch_bounds__(256) RT_SHADE_ATTR void k_wf_shade(Scene S, rtx_camera_desc cam, uint32_t cap, unsigned long long seed,
                                                  int use_lights, WfCounters* c, WfQueue qin, WfHits h, WfQueue qout,
                                                  double* __restrict__ accum, unsigned long long* __restrict__ stats,
                                                  const uint8_t* __restrict__ mat_class) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t n = c->n_in;
  // a queue of n entries has no more than n / RT_SHADE_CHUNK + 1 chunks to hand out: workgroups beyond that leave before
  // staging anything (a launch over a nearly empty queue took 50 us for its 4096 workgroups to stage their tables)
  // A short queue is handed out in smaller chunks, down to one entry per thread, so that it is spread over the whole
  // GPU: with full chunks 60 000 entries keep 15 workgroups busy for 16 trips each, and a trip is one whole chain
  // of dependent loads (the late iterations of a call are nothing but such latencies).
  uint32_t chunk = RT_SHADE_CHUNK;
  const uint32_t spread = gridDim.x >= 8u ? gridDim.x / 8u : 1u;   // two workgroups per CU (every workgroup stages the scene's small tables first)
  if ((unsigned long long)spread * RT_SHADE_CHUNK > n) {
    chunk = ((n + spread - 1u) / spread + 255u) & ~255u;
    chunk = chunk < 256u ? 256u : (chunk > RT_SHADE_CHUNK ? RT_SHADE_CHUNK : chunk);
  }
  if (blockIdx.x != 0 && (unsigned long long)blockIdx.x * chunk >= n) return;
  uint32_t probes = 0, drops = 0, dead = 0;
  __shared__ uint32_t s_first, s_out_base;
  __shared__ uint32_t s_cnt[RT_SHADE_SLOTS * RT_SHADE_KEYS], s_key_base[RT_SHADE_KEYS + 1], s_sorted[RT_SHADE_CHUNK];
  __shared__ uint32_t s_kind[RT_SHADE_CHUNK];  // hit kind of the sorted entries: the shading pass starts its chain of dependent loads from LDS
  if (n != 0 && threadIdx.x == 0) atomicMin(&c->sh_start, (unsigned long long)wall_clock64());
#if RT_SHADE_LDS_TABLES
  // Small scene tables are staged in LDS: shade_bounce reaches them through chains of dependent loads (hit ->
  // material -> texture, light -> its primitive, read twice per diffuse bounce by HittablePDF::generate and ::value),
  // and with two waves per SIMD every round trip to the L2 shows.  Each table that still fits the budget is copied, in
  // the order below; the table pointers of this kernel's copy of the scene then point into LDS.
  __shared__ __attribute__((aligned(16))) unsigned char s_tables[RT_SHADE_LDS_TABLES];
  if (n != 0) {
    const rtx_light* g_lights = S.lights;   // the global copy: the light primitives are found through it
    uint32_t at = 0;
    auto stage = [&](const void* src, uint32_t bytes) -> const void* {   // bytes: a multiple of 8
      if (bytes == 0 || at + bytes > RT_SHADE_LDS_TABLES) return nullptr;
      uint64_t* dst = (uint64_t*)(s_tables + at);
      for (uint32_t k = threadIdx.x; k < bytes / 8u; k += 256u) dst[k] = ((const uint64_t*)src)[k];
      at += bytes;
      return dst;
    };
    if (const void* p = stage(S.lights, S.n_lights * (uint32_t)sizeof(rtx_light))) S.lights = (const rtx_light*)p;
    if (S.n_lights && at + S.n_lights * (uint32_t)(sizeof(rtx_quad) + sizeof(rtx_sphere)) <= RT_SHADE_LDS_TABLES) {
      uint64_t* lq = (uint64_t*)(s_tables + at);
      uint64_t* ls = lq + S.n_lights * (sizeof(rtx_quad) / 8u);
      const uint32_t uq = (uint32_t)sizeof(rtx_quad) / 8u, us = (uint32_t)sizeof(rtx_sphere) / 8u;
      for (uint32_t k = threadIdx.x; k < S.n_lights * uq; k += 256u) {
        const rtx_light L = g_lights[k / uq];
        lq[k] = L.kind == RTX_NODE_QUAD ? ((const uint64_t*)&S.quads[L.idx])[k % uq] : 0ull;
      }
      for (uint32_t k = threadIdx.x; k < S.n_lights * us; k += 256u) {
        const rtx_light L = g_lights[k / us];
        ls[k] = L.kind == RTX_NODE_SPHERE ? ((const uint64_t*)&S.spheres[L.idx])[k % us] : 0ull;
      }
      at += S.n_lights * (uint32_t)(sizeof(rtx_quad) + sizeof(rtx_sphere));
      S.light_quads = (const rtx_quad*)lq;
      S.light_spheres = (const rtx_sphere*)ls;
    }
    if (const void* p = stage(S.xform_ops, S.n_xform_ops * (uint32_t)sizeof(rtx_xform_op))) {
      S.xform_ops = (const rtx_xform_op*)p;
      if (const void* x = stage(S.xforms, S.n_xforms * (uint32_t)sizeof(rtx_xform))) S.xforms = (const rtx_xform*)x;
    } else if (S.n_xform_ops == 0) {
      if (const void* x = stage(S.xforms, S.n_xforms * (uint32_t)sizeof(rtx_xform))) S.xforms = (const rtx_xform*)x;
    }
    if (const void* p = stage(S.materials, S.n_materials * (uint32_t)sizeof(rtx_material))) S.materials = (const rtx_material*)p;
    if (const void* p = stage(S.textures, S.n_textures * (uint32_t)sizeof(rtx_texture))) S.textures = (const rtx_texture*)p;
    __syncthreads();
  }
#endif
  // the queue is handed out dynamically, one chunk at a time: the trip count is the same for every thread of a
  // workgroup, and no workgroup is left with a long static share at the end of the launch
  for (;;) {
  if (threadIdx.x == 0) s_first = atomicAdd(&c->work_shade, chunk);
  for (uint32_t k = threadIdx.x; k < RT_SHADE_KEYS * RT_SHADE_SLOTS; k += 256u) s_cnt[k] = 0;
  __syncthreads();
  const uint32_t first = s_first;
  if (first >= n) break;
  const uint32_t lim = first + chunk < n ? first + chunk : n;   // this trip's entries: [first, lim)
  RT_KPROF_T(kp_t0);
  uint32_t my_key[RT_SHADE_PER_THREAD], my_rank[RT_SHADE_PER_THREAD];
  // Keys of this thread's entries.  Every step is written as a loop of its own over the entries, with unconditional
  // loads from addresses that are always valid, so that the loads of a step are in flight together: one round trip
  // per step instead of one per step and entry.
  uint32_t c_kind[RT_SHADE_PER_THREAD], c_mat[RT_SHADE_PER_THREAD], c_db[RT_SHADE_PER_THREAD], c_px[RT_SHADE_PER_THREAD], c_sm[RT_SHADE_PER_THREAD];
RT_UNROLL
  for (uint32_t e = 0; e < RT_SHADE_PER_THREAD; ++e) {
    const uint32_t k = first + e * 256u + threadIdx.x, ks = k < lim ? k : first;
    c_kind[e] = h.kind_idx[ks];
    const uint32_t* u = qin.u + ks;
    c_px[e] = u[0]; c_sm[e] = u[(size_t)cap]; c_db[e] = u[2 * (size_t)cap];
  }
RT_UNROLL
  for (uint32_t e = 0; e < RT_SHADE_PER_THREAD; ++e) {
    const uint32_t hkind = RTX_NODE_KIND(c_kind[e]), hidx = RTX_NODE_IDX(c_kind[e]);
    const uint32_t* mp = &S.materials[0].kind;   // a miss reads something harmless
    if (hkind == RTX_NODE_SPHERE) mp = &S.spheres[hidx].mat;
    else if (hkind == RTX_NODE_QUAD) mp = &S.quads[hidx].mat;
    else if (hkind == RTX_NODE_MEDIUM) mp = &S.media[hidx].mat;
    c_mat[e] = *mp;
  }
RT_UNROLL
  for (uint32_t e = 0; e < RT_SHADE_PER_THREAD; ++e) {
    const uint32_t hkind = RTX_NODE_KIND(c_kind[e]);
    const bool hit = hkind == RTX_NODE_SPHERE || hkind == RTX_NODE_QUAD || hkind == RTX_NODE_MEDIUM;
    const uint32_t cls = mat_class[hit ? c_mat[e] : 0u];
    c_mat[e] = hit ? ((cls << 1) | (hkind == RTX_NODE_SPHERE ? 1u : 0u)) : 0u;
  }
  // Entries that are not shaded: 1 = a dead entry (remaining depth 0, see above); 2 = a Lambertian or isotropic hit
  // whose path the roulette ends.  shade_bounce makes that draw first for these materials; a path it ends
  // contributes nothing (no emission) and is dropped here (drawn for every entry, without a branch: the draws overlap).
RT_UNROLL
  for (uint32_t e = 0; e < RT_SHADE_PER_THREAD; ++e) {
    const uint32_t depth = c_db[e] >> 16;
    uint32_t skip = depth == 0u ? 1u : 0u;
    if (RT_SHADE_DROP) {
      const double xi = shade_roulette_draw(rng_make(seed, c_px[e], c_sm[e]), c_db[e] & 0xFFFFu);
      if (skip == 0u && (c_mat[e] >> 1) >= MC_LAMBERT && depth > 3u && xi > 0.8) skip = 2u;
    }
    c_db[e] = skip;
  }
RT_UNROLL
  for (uint32_t e = 0; e < RT_SHADE_PER_THREAD; ++e) {
    const uint32_t k = first + e * 256u + threadIdx.x;
    const bool valid = k < lim && c_db[e] == 0u;
    if (k < lim) { dead += c_db[e] == 1u ? 1u : 0u; drops += c_db[e] == 2u ? 1u : 0u; }
    const uint32_t key = c_mat[e];
    // stable rank among the entries of this wave with the same key: one ballot per key present in the wave
    const uint32_t v = e * 4u + wave;
    unsigned long long todo = __ballot(valid);
    uint32_t rank = 0;
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const uint32_t lk = __shfl(key, leader);
      const unsigned long long m = __ballot(valid && key == lk);
      if (key == lk) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      if ((int)lane == leader) s_cnt[v * RT_SHADE_KEYS + lk] = (uint32_t)__popcll(m);
      todo &= ~m;
    }
    my_key[e] = valid ? key : 0xFFFFFFFFu; my_rank[e] = rank;
  }
  __syncthreads();
  if (wave == 0) {
    // lane k < RT_SHADE_KEYS: exclusive prefix over the (wave, pass) slots of key k (all counts read first, so the LDS
    // round trips overlap), then an exclusive prefix of the keys' totals across the lanes
    uint32_t x[RT_SHADE_SLOTS], run = 0;
    const uint32_t kk = lane < RT_SHADE_KEYS ? lane : 0u;
RT_UNROLL
    for (uint32_t v = 0; v < RT_SHADE_SLOTS; ++v) x[v] = s_cnt[v * RT_SHADE_KEYS + kk];
RT_UNROLL
    for (uint32_t v = 0; v < RT_SHADE_SLOTS; ++v) { const uint32_t c0 = x[v]; x[v] = run; run += c0; }
    if (lane < RT_SHADE_KEYS) {
RT_UNROLL
      for (uint32_t v = 0; v < RT_SHADE_SLOTS; ++v) s_cnt[v * RT_SHADE_KEYS + kk] = x[v];
    }
    uint32_t incl = lane < RT_SHADE_KEYS ? run : 0u;
RT_UNROLL
    for (uint32_t off = 1; off < RT_SHADE_KEYS; off <<= 1) { const uint32_t y = __shfl_up(incl, off); if (lane >= off) incl += y; }
    if (lane < RT_SHADE_KEYS) s_key_base[lane] = incl - run;
    // entries to shade, and the queue slots of those that may go on (keys from MC_METAL << 1 on): one atomic
    const uint32_t total = __shfl(incl, RT_SHADE_KEYS - 1u), n_end = __shfl(incl, (MC_METAL << 1) - 1u);
    if (lane == 0) {
      s_key_base[RT_SHADE_KEYS] = total;
      s_out_base = total > n_end ? atomicAdd(&c->n_out, total - n_end) : 0u;
    }
  }
  __syncthreads();
RT_UNROLL
  for (uint32_t e = 0; e < RT_SHADE_PER_THREAD; ++e)
    if (my_key[e] != 0xFFFFFFFFu) {
      const uint32_t pos = s_key_base[my_key[e]] + s_cnt[(e * 4u + wave) * RT_SHADE_KEYS + my_key[e]] + my_rank[e];
      s_sorted[pos] = first + e * 256u + threadIdx.x;
      s_kind[pos] = c_kind[e];
    }
  __syncthreads();
  const uint32_t chunk_n = s_key_base[RT_SHADE_KEYS], n_end = s_key_base[MC_METAL << 1], out_base = s_out_base;
  RT_KPROF_ADD(12, kp_t0);
  for (uint32_t e = 0; e < RT_SHADE_PER_THREAD && e * 256u < chunk_n; ++e) {
    RT_KPROF_T(kp_t1);
    const uint32_t pos = e * 256u + threadIdx.x;
    if (pos < chunk_n) {
      const uint32_t idx = s_sorted[pos];
      PathState ps;
      Ray next;
      const double* f = qin.f + idx;
      Ray ray;
      ray.o = v3(f[0 * (size_t)cap], f[1 * (size_t)cap], f[2 * (size_t)cap]);
      ray.d = v3(f[3 * (size_t)cap], f[4 * (size_t)cap], f[5 * (size_t)cap]);
      ray.tm = f[6 * (size_t)cap];
      ps.T = v3(f[7 * (size_t)cap], f[8 * (size_t)cap], f[9 * (size_t)cap]);
      ps.L = v3(0.0, 0.0, 0.0);
      const uint32_t* u = qin.u + idx;
      const uint32_t pixel = u[0], sample = u[(size_t)cap], db = u[2 * (size_t)cap];
      ps.depth = (int)(db >> 16); ps.bounce = db & 0xFFFFu;
      ps.rng = rng_make(seed, pixel, sample);
      Hit hit;
      hit.t = h.t[idx]; hit.kind_idx = s_kind[pos]; hit.ctx = h.ctx[idx];
      next.o = v3(0.0, 0.0, 0.0); next.d = v3(1.0, 1.0, 1.0); next.tm = 0.0;
      const bool alive = shade_bounce(S, v3(cam.background), use_lights != 0, ray, hit, h.uv + 2 * (size_t)idx, ps, next, probes);
      if (!alive && (ps.L.x != 0.0 || ps.L.y != 0.0 || ps.L.z != 0.0)) {
        double* a = accum + (size_t)pixel * 3;
        unsafeAtomicAdd(a + 0, ps.L.x); unsafeAtomicAdd(a + 1, ps.L.y); unsafeAtomicAdd(a + 2, ps.L.z);
      }
      if (pos >= n_end) {   // this entry owns a slot of the other queue: the path's next segment, or a dead entry
        const uint32_t slot = out_base + (pos - n_end);
        if (!alive) {
          // A dead entry's ray must be cheap to walk: it starts far outside any scene and points away from it, so the
          // first box it meets rejects it and every top-level primitive misses (all roots and plane distances are
          // about -1e30).  Not axis-parallel: Aabb::hit skips the axes a ray is parallel to (aabb.rs:142-144), and a
          // ray that skips two axes passes a large part of a BVH.
          next.o = v3(1e30, 1e30, 1e30); next.d = v3(1.0, 1.0, 1.0); next.tm = 0.0; ps.T = v3(0.0, 0.0, 0.0);
        }
        double* g = qout.f + slot;
        g[0 * (size_t)cap] = next.o.x; g[1 * (size_t)cap] = next.o.y; g[2 * (size_t)cap] = next.o.z;
        g[3 * (size_t)cap] = next.d.x; g[4 * (size_t)cap] = next.d.y; g[5 * (size_t)cap] = next.d.z;
        g[6 * (size_t)cap] = next.tm;
        g[7 * (size_t)cap] = ps.T.x; g[8 * (size_t)cap] = ps.T.y; g[9 * (size_t)cap] = ps.T.z;
        uint32_t* w = qout.u + slot;
        w[0] = pixel; w[(size_t)cap] = sample;
        w[2 * (size_t)cap] = alive ? (((uint32_t)ps.depth << 16) | (ps.bounce & 0xFFFFu)) : 0u;
      }
    }
    RT_KPROF_ADD(13, kp_t1);
  }
  }
  // statistics: one atomic per wave and counter
  uint32_t wp = probes, wd = drops, wx = dead;
  for (int off = 32; off > 0; off >>= 1) { wp += __shfl_down(wp, off); wd += __shfl_down(wd, off); wx += __shfl_down(wx, off); }
  if (lane == 0 && wp) atomicAdd(&stats[2], (unsigned long long)wp);
  if (lane == 0 && wd) atomicAdd(&stats[3], (unsigned long long)wd);
  if (lane == 0 && wx) atomicAdd(&stats[4], (unsigned long long)wx);
  if (n != 0 && threadIdx.x == 0) atomicMax(&c->sh_end, (unsigned long long)wall_clock64());
}

// Tail of a render call.  Once no new camera paths are left and only a few survivors remain, an iteration of three
// launches per bounce is all launch latency.  This kernel takes the survivors (at most `threshold`) and runs every one
// of them to its end in place: closest hit + shading in a loop, the program read from global memory.  It is launched
// after every chunk of iterations and does nothing until that moment; afterwards the queue is empty.
__global__ __launch_bounds__(256) void k_wf_finish(Scene S, rtx_camera_desc cam, uint32_t cap, unsigned long long seed,
                                                   int use_lights, uint32_t threshold, WfCounters* c, WfQueue qin,
                                                   double* __restrict__ accum, unsigned long long* __restrict__ stats) {
  const uint32_t n = c->n_out;  // survivors of the last shading launch
  if (blockIdx.x == 0 && threadIdx.x == 0) wf_flush_timing(c);
  if (n == 0 || n > threshold || c->next_path != c->total_paths) return;
  if (blockIdx.x * blockDim.x >= n) return;
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  const V3 bg = v3(cam.background);
  bool alive = idx < n;
  const uint32_t ld = alive ? idx : 0;
  const double* f = qin.f + ld;
  Ray ray;
  ray.o = v3(f[0 * (size_t)cap], f[1 * (size_t)cap], f[2 * (size_t)cap]);
  ray.d = v3(f[3 * (size_t)cap], f[4 * (size_t)cap], f[5 * (size_t)cap]);
  ray.tm = f[6 * (size_t)cap];
  PathState ps;
  ps.T = v3(f[7 * (size_t)cap], f[8 * (size_t)cap], f[9 * (size_t)cap]);
  ps.L = v3(0.0, 0.0, 0.0);
  const uint32_t* u = qin.u + ld;
  const uint32_t pixel = u[0], sample = u[(size_t)cap], db = u[2 * (size_t)cap];
  ps.depth = (int)(db >> 16); ps.bounce = db & 0xFFFFu;
  ps.rng = rng_make(seed, pixel, sample);
  alive = alive && ps.depth > 0;   // a dead entry of the shading kernel
  uint32_t segs = 0, probes = 0;
  while (__any(alive)) {  // the wave leaves together: the traversal votes across all 64 lanes
    Hit hit;
    double stale_uv[2] = {0.0, 0.0};
    if (alive) ++segs;
    RayRegs rs;
    rs.r = ray;
    closest_hit<true>(S, S.prog, S.world_begin, alive, rs, ray.tm, 0.001, RT_INF, hit, S.n_media != 0, ps.rng, ps.bounce, stale_uv);
    if (alive) {
      Ray next;
      ps.L = v3(0.0, 0.0, 0.0);
      alive = shade_bounce(S, bg, use_lights != 0, ray, hit, stale_uv, ps, next, probes);
      ray = next;
      if (!alive && (ps.L.x != 0.0 || ps.L.y != 0.0 || ps.L.z != 0.0)) {
        double* a = accum + (size_t)pixel * 3;
        unsafeAtomicAdd(a + 0, ps.L.x); unsafeAtomicAdd(a + 1, ps.L.y); unsafeAtomicAdd(a + 2, ps.L.z);
      }
    }
  }
  __shared__ unsigned int sh[2];
  if (threadIdx.x < 2) sh[threadIdx.x] = 0;
  __syncthreads();
  atomicAdd(&sh[0], segs); atomicAdd(&sh[1], probes);
  __syncthreads();
  if (threadIdx.x < 2 && sh[threadIdx.x]) atomicAdd(&stats[1 + threadIdx.x], (unsigned long long)sh[threadIdx.x]);
  if (blockIdx.x == 0 && threadIdx.x == 0) c->finished = 1u;  // the next k_wf_prepare drops the survivors
}

// The same tail of a call for a program whose world program fits LDS: a workgroup keeps its survivors (one per thread,
// refilled from the queue) to their end and walks every bounce of the world program from LDS, staged once; a ray that
// passes the deferred section's root box is noted, as in the wavefront traversal, and walks that section where it lies,
// in global memory - the walk of world.hit cut in two, bit for bit.  Nothing in the loop synchronises the waves of a
// workgroup: each takes its bounces at its own pace.  (Staging the two parts in turn, with the barriers that needs,
// made every bounce wait three times for the slowest of twelve waves: `quick` 1300 against 1340 Msamples/s; with the
// deferred section walked nearer child first by the stream walker the kernel grew and lost that again: 1325.)
#ifndef RT_FINISH_THREADS
#define RT_FINISH_THREADS 768   // (one workgroup per CU for its LDS: 512: quick 1095, 768: 1135, 1024: 1125 Msamples/s - the registers of shading spill, but more paths finish per round)
#endif
template <int MM, bool FLAT = false>   // FLAT: as in k_wf_traverse
__global__ __launch_bounds__(RT_FINISH_THREADS) void k_wf_finish_lds(Scene S, rtx_camera_desc cam, uint32_t cap, unsigned long long seed,
                                                                    int use_lights, uint32_t threshold, WfCounters* c, WfQueue qin,
                                                                    double* __restrict__ accum, unsigned long long* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lds_prog[];
  const uint32_t n = c->n_out;  // survivors of the last shading launch
  if (blockIdx.x == 0 && threadIdx.x == 0) wf_flush_timing(c);
  if (n == 0 || n > threshold || c->next_path != c->total_paths) return;
  if (blockIdx.x * blockDim.x >= n) return;
  const V3 bg = v3(cam.background);
  const uint32_t lane = threadIdx.x & 63u;
  // The workgroups are persistent: a thread starts with the survivor of its own number and, whenever its path has
  // ended, takes the next one nobody has taken yet (one atomic per wave and bounce for the lanes that are free), so a
  // workgroup is not held by its longest path while its other lanes idle.  Paths do not depend on each other and
  // their random numbers are addressed by their own coordinates: who traces which changes nothing.
  const uint32_t first_free = gridDim.x * blockDim.x;   // survivors below this number have an owner from the start
  bool more = first_free < n;                            // (uniform over a wave) survivors may be left to take
  Ray ray;
  ray.o = v3(0, 0, 0); ray.d = v3(0, 0, 1); ray.tm = 0.0;
  PathState ps;
  ps.T = v3(0, 0, 0); ps.L = v3(0, 0, 0); ps.depth = 0; ps.bounce = 0; ps.rng = rng_make(seed, 0, 0);
  uint32_t pixel = 0;
  bool alive = false;
  auto load_path = [&](uint32_t idx) {
    const double* f = qin.f + idx;
    ray.o = v3(f[0 * (size_t)cap], f[1 * (size_t)cap], f[2 * (size_t)cap]);
    ray.d = v3(f[3 * (size_t)cap], f[4 * (size_t)cap], f[5 * (size_t)cap]);
    ray.tm = f[6 * (size_t)cap];
    ps.T = v3(f[7 * (size_t)cap], f[8 * (size_t)cap], f[9 * (size_t)cap]);
    ps.L = v3(0.0, 0.0, 0.0);
    const uint32_t* u = qin.u + idx;
    const uint32_t sample = u[(size_t)cap], db = u[2 * (size_t)cap];
    pixel = u[0];
    ps.depth = (int)(db >> 16); ps.bounce = db & 0xFFFFu;
    ps.rng = rng_make(seed, pixel, sample);
    alive = ps.depth > 0;   // depth 0: a dead entry of the shading kernel
  };
  {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) load_path(idx);
  }
  uint32_t segs = 0, probes = 0;
  const uint32_t main_units = S.tail_begin ? S.tail_begin : S.n_units, tail_units = S.tail_begin ? S.n_units - S.tail_begin : 0u;   // both even
  // The world program is staged ONCE and stays; the deferred section is walked where it lies, in global memory (the
  // L2), by the rays that passed its root box.  No barrier follows in the loop: every wave takes its bounces at its own
  // pace instead of waiting three times per bounce for the slowest of the workgroup's waves.
  {
    const uint4* src = (const uint4*)S.prog;
    uint4* dst = (uint4*)lds_prog;
    for (uint32_t k = threadIdx.x; k < main_units / 2; k += blockDim.x) dst[k] = src[k];
  }
  __syncthreads();
  for (;;) {
    if (more) {   // lanes without a path take the next survivors
      const unsigned long long want = __ballot(!alive);
      if (want) {
        const uint32_t cnt = (uint32_t)__popcll(want);
        uint32_t base = 0;
        if (lane == (uint32_t)(__ffsll((long long)want) - 1)) base = atomicAdd(&c->work_finish, cnt);
        base = __shfl(base, __ffsll((long long)want) - 1) + first_free;
        const uint32_t mine = base + (uint32_t)__popcll(want & ((1ull << lane) - 1ull));
        if (!alive && mine < n) load_path(mine);
        if (base + cnt >= n) more = false;
      }
    }
    if (!__any(alive)) {
      if (more) continue;       // everything taken so far was a dead entry: take again
      break;
    }
    Hit hit;
    double stale_uv[2] = {0.0, 0.0};
    if (alive) ++segs;
    RayRegs rs;
    rs.r = ray;
    bool deferred;
    if (FLAT && MM != MEDIA_ALL) {
      Hit hh;
      hh.t = RT_INF; hh.kind_idx = (uint32_t)RTX_NODE_MAKE(RTX_NODE_NONE, 0); hh.ctx = 0u;
      if (S.flat_begin != S.world_begin)
        walk_program<MEDIA_NONE, true, true>(S, lds_prog, S.world_begin, alive, rs, ray.tm, 0.001, RT_INF, hh, false, ps.rng, ps.bounce, nullptr,
                                       (uint32_t)RTX_NODE_MAKE(RTX_NODE_NONE, 0), 0u, false, V3(), V3(), S.flat_begin - 1u);
      deferred = walk_flat<MM>(S, lds_prog, S.flat_begin, alive, rs, ray.tm, 0.001, hh.t, hit, MM != MEDIA_NONE,
                               ps.rng, ps.bounce, stale_uv, hh.kind_idx, hh.ctx, false, V3(), V3());
    } else {
      deferred = walk_program<MM, true>(S, lds_prog, S.world_begin, alive, rs, ray.tm, 0.001, RT_INF, hit, MM != MEDIA_NONE, ps.rng, ps.bounce, stale_uv);
    }
    if (tail_units) {
      const bool in_section = alive && deferred;
      if (__any(in_section)) {
        Hit h2;
        walk_program<MEDIA_NONE, false>(S, S.prog, S.tail_begin, in_section, rs, ray.tm, 0.001, hit.t, h2, false, ps.rng, ps.bounce, stale_uv);
        if (in_section && hit_any(h2)) hit = h2;
      }
    }
    if (alive) {
      Ray next;
      ps.L = v3(0.0, 0.0, 0.0);
      alive = shade_bounce(S, bg, use_lights != 0, ray, hit, stale_uv, ps, next, probes);
      ray = next;
      if (!alive && (ps.L.x != 0.0 || ps.L.y != 0.0 || ps.L.z != 0.0)) {
        double* a = accum + (size_t)pixel * 3;
        unsafeAtomicAdd(a + 0, ps.L.x); unsafeAtomicAdd(a + 1, ps.L.y); unsafeAtomicAdd(a + 2, ps.L.z);
      }
    }
  }
  __shared__ unsigned int sh[2];
  if (threadIdx.x < 2) sh[threadIdx.x] = 0;
  __syncthreads();
  atomicAdd(&sh[0], segs); atomicAdd(&sh[1], probes);
  __syncthreads();
  if (threadIdx.x < 2 && sh[threadIdx.x]) atomicAdd(&stats[1 + threadIdx.x], (unsigned long long)sh[threadIdx.x]);
  if (blockIdx.x == 0 && threadIdx.x == 0) c->finished = 1u;  // the next k_wf_prepare drops the survivors
}

// ------------------------------------------------------------ fused path ---
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
  path_begin(ps, 0, rng_make(o.seed, pixel, 0));
  for (;;) {
    if (!alive && inrect && s < o.sample_end) {
      PathRng rng = rng_make(o.seed, pixel, (uint32_t)s);
      ray = camera_ray(cam, i, j, s, rng);
      path_begin(ps, cam.max_depth, rng);
      ++s; ++npaths;
      alive = ps.depth > 0;
    }
    if (!__any(alive)) break;  // the wave leaves together: the traversal below votes across all 64 lanes
    Hit hit;
    double stale_uv[2] = {0.0, 0.0};
    if (alive) ++segs;
    RayRegs rs;
    rs.r = ray;
    closest_hit<true>(S, S.prog, S.world_begin, alive, rs, ray.tm, 0.001, RT_INF, hit, true, ps.rng, ps.bounce, stale_uv);
    if (alive) {
      Ray next;
      alive = shade_bounce(S, bg, use_lights != 0, ray, hit, stale_uv, ps, next, probes);
      ray = next;
      if (!alive) sum = sum + ps.L;
    }
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
  const bool active = idx < n;
  const double* r = rays + 7 * (active ? idx : 0);
  Ray ray;
  ray.o = v3(r[0], r[1], r[2]); ray.d = v3(r[3], r[4], r[5]); ray.tm = r[6];
  Hit h;
  RayRegs rs;
  rs.r = ray;
  closest_hit<false>(S, S.prog, S.world_begin, active, rs, ray.tm, 0.001, RT_INF, h, false, PathRng(), 0, nullptr);
  if (!active) return;
  const bool any = hit_any(h);
  Surface sf;
  sf.n = v3(0, 0, 0); sf.prim_id = 0xFFFFFFFFu; sf.mat = 0xFFFFFFFFu; sf.front = false;
  if (any) resolve_hit(S, ray, h, false, nullptr, sf);
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

// =================================================================== host ===

static thread_local std::string g_error;

struct TimedSpan { hipEvent_t e0, e1; int what; };  // what: 0 = whole render call, 1 = world-program traversal, 2 = deferred section, 3 = head section

#define RT_WF_MAX_LANES 8
struct WfBuffers {                 // device memory of one wavefront pipeline
  WfQueue q[2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  WfHits hits = {nullptr, nullptr, nullptr, nullptr};
  uint32_t* defer = nullptr;       // queue indices of the rays that entered the deferred section
  WfCounters* c = nullptr;
};
struct rtx_context {
  int device = 0;
  int n_cus = 256;
  double wall_clock_khz = 100000.0;  // rate of wall_clock64()
  std::string error;
  unsigned long long* d_counters = nullptr;  // paths, queue entries traversed, pdf_probes, roulette drops, dead entries among those traversed, rays walked again in the reference's order
  double render_ms = 0.0, last_kernel_ms = 0.0, traverse_ms = 0.0, tail_ms = 0.0, head_ms = 0.0, shade_ms = 0.0;
  uint64_t launches = 0, traverse_launches = 0, tail_launches = 0, head_launches = 0, shade_launches = 0, iterations = 0;
  std::vector<TimedSpan> pending;  // timed spans not yet read back
  std::vector<hipEvent_t> free_events;
  // wavefront workspace (allocated on first use, kept for the context's lifetime): up to RT_WF_MAX_LANES independent pipelines
  bool wf_ready = false;
  struct WfLane : WfBuffers {
    uint32_t cap = 0;              // paths in flight its buffers hold
    std::vector<void*> allocs;
    WfCounters* host = nullptr;    // pinned, two slots
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipStream_t stream = nullptr;  // the second pipeline runs on a stream of its own
    hipEvent_t joined = nullptr;
  } wf[RT_WF_MAX_LANES];
  hipEvent_t wf_fork = nullptr;
  // Render calls of one context share its workspace (queues, counters, statistics), whatever stream each is given:
  // every call first makes its stream wait for the end of the one before it.
  hipEvent_t call_done = nullptr;
  bool call_recorded = false;
  double* host_accum_dev = nullptr;  // rtx_render's device copy of the caller's host accumulator; grows, never shrinks
  size_t host_accum_len = 0;
};

struct rtx_scene {
  rtx_context* ctx = nullptr;
  Scene dev;  // device pointers
  uint32_t n_general_media = 0;  // media of the world program whose boundary is a program of its own
  bool nothing_behind_head = false;  // the world program holds nothing but its head section (no unit behind it tests anything)
  const uint8_t* mat_class = nullptr;  // per material: the class (MC_*) the shading kernel sorts hits by
  std::vector<void*> allocs;
};

static rtx_status fail(rtx_context* ctx, rtx_status st, const std::string& msg) {
  if (ctx) ctx->error = msg;
  g_error = msg;
  return st;
}
int rtx_internal_device(const rtx_context* ctx) { return ctx->device; }
rtx_status rtx_internal_fail(rtx_context* ctx, rtx_status st, const char* message) { return fail(ctx, st, message); }
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
    if (hipEventSynchronize(p.e1) == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
      if (p.what == 0) { ctx->render_ms += ms; ctx->last_kernel_ms = ms; ctx->launches += 1; }
      else if (p.what == 1) { ctx->traverse_ms += ms; ctx->traverse_launches += 1; }
      else if (p.what == 2) ctx->tail_ms += ms;   // the section launches themselves are counted where they are made
      else ctx->head_ms += ms;
    }
    ctx->free_events.push_back(p.e0); ctx->free_events.push_back(p.e1);
  }
  ctx->pending.clear();
}
// Begin/end a timed span on `stream`; returns false when no event could be created (timing is then skipped).
static bool span_begin(rtx_context* ctx, hipStream_t stream, TimedSpan& sp, int what) {
  sp.e0 = get_event(ctx); sp.e1 = get_event(ctx); sp.what = what;
  if (!sp.e0 || !sp.e1) return false;
  return hipEventRecord(sp.e0, stream) == hipSuccess;
}
static void span_end(rtx_context* ctx, hipStream_t stream, TimedSpan& sp) {
  if (hipEventRecord(sp.e1, stream) == hipSuccess) ctx->pending.push_back(sp);
}

static bool check_texture_depth(const rtx_scene_desc* d, uint32_t tex, int depth) {
  if (depth > RTX_MAX_TEXTURE_DEPTH) return false;
  const rtx_texture& t = d->textures[tex];
  if (t.kind == RTX_TEX_CHECKER) return check_texture_depth(d, t.a, depth + 1) && check_texture_depth(d, t.b, depth + 1);
  return true;
}

// Cost class of a texture tree: 0 = constant or checker of constants, 1 = reads an image, 2 = evaluates noise.
static int texture_cost(const rtx_scene_desc* d, uint32_t tex, int depth) {
  if (depth > RTX_MAX_TEXTURE_DEPTH) return 0;
  const rtx_texture& t = d->textures[tex];
  if (t.kind == RTX_TEX_CHECKER) return std::max(texture_cost(d, t.a, depth + 1), texture_cost(d, t.b, depth + 1));
  return t.kind == RTX_TEX_NOISE ? 2 : (t.kind == RTX_TEX_IMAGE ? 1 : 0);
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
  if (d->n_nodes >= (1u << 28)) return "more than 2^28 nodes";
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
  // a box may only skip forward, and never out of the program (world or medium boundary) it belongs to
  for (uint32_t r = 0; r <= d->n_media; ++r) {
    const uint32_t b = r == 0 ? d->world_begin : d->media[r - 1].b_begin, e = r == 0 ? d->world_end : d->media[r - 1].b_end;
    for (uint32_t i = b; i < e; ++i)
      if (RTX_NODE_KIND(d->nodes[i].kind_idx) == RTX_NODE_BOX && (d->nodes[i].skip <= i || d->nodes[i].skip > e))
        return "box skip link must point forward inside its program";
    if (r > 0)
      for (uint32_t i = b; i < e; ++i)
        if (RTX_NODE_KIND(d->nodes[i].kind_idx) == RTX_NODE_MEDIUM) return "medium inside a medium boundary";
  }
  for (uint32_t i = 0; i < d->n_nodes; ++i) {
    const uint32_t kind = RTX_NODE_KIND(d->nodes[i].kind_idx), idx = RTX_NODE_IDX(d->nodes[i].kind_idx);
    switch (kind) {
      case RTX_NODE_BOX:
        if (idx >= d->n_boxes) return "box index out of range";
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

static const char* validate_render(const rtx_camera_desc* cam, const rtx_render_opts* o) {
  if (cam->image_width < 1 || cam->image_height < 1 || cam->sqrt_spp < 1) return "camera not initialised (call rtx_camera_initialize)";
  if ((int64_t)cam->image_width * cam->image_height > 0x7FFFFFFF / 4) return "image too large";
  if (o->x0 < 0 || o->y0 < 0 || o->x1 > cam->image_width || o->y1 > cam->image_height || o->x0 > o->x1 || o->y0 > o->y1) return "pixel rectangle outside the image";
  if (o->sample_begin < 0 || o->sample_begin > o->sample_end || o->sample_end > cam->sqrt_spp * cam->sqrt_spp) return "sample range outside [0, sqrt_spp^2]";
  if (o->pipeline < RTX_PIPELINE_DEFAULT || o->pipeline > RTX_PIPELINE_WAVEFRONT) return "unknown pipeline";
  return nullptr;
}

#if defined(RT_PROF)
static const size_t kLdsBudget = 160 * 1024 - 4096;  // the tuning build keeps 4 KiB for its counters
#else
static const size_t kLdsBudget = 160 * 1024;  // gfx950: 160 KiB of LDS per CU, one persistent workgroup per CU
#endif

// Every traversal kernel instantiation that is launched with dynamic LDS (the staged program, the waves' defer
// buffers, the lanes' stacks of the nearer-child-first walk: up to the whole 160 KiB) is given that limit once, when
// the first context is created.
static const size_t kFinishLdsBudget = kLdsBudget - 1024;
static hipError_t wf_set_lds_limits() {
  const void* kernels[] = {
      (const void*)k_wf_traverse<true, false, MEDIA_NONE>, (const void*)k_wf_traverse<true, false, MEDIA_SPHERES>, (const void*)k_wf_traverse<true, false, MEDIA_ALL>,
      (const void*)k_wf_traverse<true, false, MEDIA_NONE, true>, (const void*)k_wf_traverse<true, false, MEDIA_SPHERES, true>, (const void*)k_wf_traverse<true, false, MEDIA_ALL, true>,
      (const void*)k_wf_traverse<true, false, MEDIA_NONE, true, true>, (const void*)k_wf_traverse<true, false, MEDIA_SPHERES, true, true>,
      (const void*)k_wf_traverse<true, false, MEDIA_NONE, false, true>, (const void*)k_wf_traverse<true, false, MEDIA_SPHERES, false, true>,
      (const void*)k_wf_traverse<false, false, MEDIA_NONE>, (const void*)k_wf_traverse<false, false, MEDIA_SPHERES>, (const void*)k_wf_traverse<false, false, MEDIA_ALL>,
      (const void*)k_wf_traverse<true, true, MEDIA_NONE>, (const void*)k_wf_traverse<false, true, MEDIA_NONE>,
      (const void*)k_wf_stream<true, false, 0>, (const void*)k_wf_stream<false, false, 0>, (const void*)k_wf_stream<true, false, 2>, (const void*)k_wf_stream<false, false, 2>,
      (const void*)k_wf_stream<true, true, 0>, (const void*)k_wf_stream<false, true, 0>, (const void*)k_wf_stream<true, true, 1>, (const void*)k_wf_stream<false, true, 1>,
      (const void*)k_wf_stream<true, true, 2>, (const void*)k_wf_stream<false, true, 2>};
  for (const void* k : kernels) {
    const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget);
    if (e != hipSuccess) return e;
  }
  const void* finishers[] = {(const void*)k_wf_finish_lds<MEDIA_NONE>, (const void*)k_wf_finish_lds<MEDIA_SPHERES>, (const void*)k_wf_finish_lds<MEDIA_ALL>,
                             (const void*)k_wf_finish_lds<MEDIA_NONE, true>, (const void*)k_wf_finish_lds<MEDIA_SPHERES, true>};
  for (const void* k : finishers) {   // (this kernel also has a few bytes of static LDS)
    const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFinishLdsBudget);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// Workspace of pipeline `lane` for at least `want` paths in flight (a multiple of 256).  Buffers only grow; they are
// kept for the context's lifetime.  *oom is set when device memory ran out (the lane is then left without buffers).
static rtx_status wf_ensure(rtx_context* ctx, int lane, uint32_t want, bool* oom) {
  *oom = false;
  if (!ctx->wf_ready) {
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->wf_fork, hipEventDisableTiming));
    ctx->wf_ready = true;
  }
  rtx_context::WfLane& L = ctx->wf[lane];
  if (!L.host) {
    HIP_TRY(ctx, hipMalloc((void**)&L.c, sizeof(WfCounters)));
    HIP_TRY(ctx, hipHostMalloc((void**)&L.host, 2 * sizeof(WfCounters), hipHostMallocDefault));
    for (int k = 0; k < 2; ++k) HIP_TRY(ctx, hipEventCreateWithFlags(&L.ev[k], hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&L.joined, hipEventDisableTiming));
    if (lane > 0) HIP_TRY(ctx, hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
  }
  if (L.cap >= want) return RTX_OK;
  HIP_TRY(ctx, hipDeviceSynchronize());  // nothing may still be reading the buffers that are replaced
  for (void* p : L.allocs) (void)hipFree(p);
  L.allocs.clear();
  L.cap = 0;
  hipError_t err = hipSuccess;
  auto A = [&](void** p, size_t bytes) { if (err == hipSuccess) { err = hipMalloc(p, bytes); if (err == hipSuccess) L.allocs.push_back(*p); } };
  for (int k = 0; k < 2; ++k) {
    A((void**)&L.q[k].f, (size_t)want * 10 * sizeof(double));
    A((void**)&L.q[k].u, (size_t)want * 3 * sizeof(uint32_t));
  }
  A((void**)&L.hits.t, (size_t)want * sizeof(double));
  A((void**)&L.hits.kind_idx, (size_t)want * sizeof(uint32_t));
  A((void**)&L.hits.ctx, (size_t)want * sizeof(uint32_t));
  A((void**)&L.hits.uv, (size_t)want * 2 * sizeof(double));
  A((void**)&L.defer, (size_t)want * sizeof(uint32_t));
  if (err != hipSuccess) {
    for (void* p : L.allocs) (void)hipFree(p);
    L.allocs.clear();
    if (err == hipErrorOutOfMemory) { (void)hipGetLastError(); *oom = true; }
    return fail(ctx, RTX_ERR_HIP, std::string("wavefront workspace of ") + std::to_string(want) + " paths: " + hipGetErrorString(err));
  }
  L.cap = want;
  return RTX_OK;
}

// Survivors at which a finishing kernel takes over (a multiple of 512); RTX_WF_FINISH overrides for tuning.  A program
// walked from global memory is finished late: its finishing kernel walks the program in the reference's order, a
// bounce of which costs more than an iteration over a short queue (stress scene: 65536: 171, 4096: 176 Msamples/s).
static unsigned finish_threshold(bool head_in_global) {
  unsigned v = head_in_global ? 4096u : 128u * 1024u;   // (from LDS, persistent workgroups: `quick` 65536: 1110, 131072 and 262144: 1165, 524288: 1040 Msamples/s)
  if (const char* e = getenv("RTX_WF_FINISH")) { long x = atol(e); if (x >= 256 && x <= (1l << 22)) v = ((unsigned)x + 511u) & ~511u; }
  return v;
}

// How one iteration of the wavefront loop is launched for a given scene and queue capacity: which sections are walked
// as streams, what each traversal kernel stages in LDS, which instantiation serves the world program.
struct WfPlan {
  Scene dev;                 // the scene as the traversal kernels see it (head_end, fresh_in_main set for this plan)
  uint32_t cap = 0;
  uint32_t head_end = 0, tail_begin = 0;
  size_t head_bytes = 0, tail_bytes = 0, main_bytes = 0, defer_bytes = 0, near_stack_bytes = 0, near_head_stack = 0, head_top_bytes = 0;
  bool head_lds = false, tail_lds = false, main_lds = false, stream_tail = false;
  bool near_lds = false, near_global = false, near_quads = false, near_head_lds = false, near_head_global = false;
  bool skip_main = false;
  void (*k_main)(Scene, uint32_t, unsigned long long, WfCounters*, WfQueue, WfHits, uint32_t*, unsigned long long*, rtx_camera_desc, WfFrame) = nullptr;
  unsigned trav_blocks = 0, gen_blocks = 0, shade_blocks = 0;
  // the finishing kernel that walks from LDS, when the world program and the deferred section each fit (else null)
  void (*k_finish_lds)(Scene, rtx_camera_desc, uint32_t, unsigned long long, int, uint32_t, WfCounters*, WfQueue, double*, unsigned long long*) = nullptr;
  size_t finish_lds_bytes = 0;
  unsigned finish_at = 65536;   // survivors at which the finishing kernel takes over
};

static WfPlan wf_make_plan(const rtx_scene* sc, uint32_t cap, unsigned long long share) {   // share: camera paths per pipeline of the call
  const rtx_context* ctx = sc->ctx;
  WfPlan P;
  P.cap = cap;
  // what each traversal kernel stages in LDS: the world program (up to the deferred section, if any) + the waves'
  // defer buffers; the deferred section's own program
  const uint32_t tail_begin = sc->dev.tail_begin, n_units = sc->dev.n_units;
  P.tail_begin = tail_begin;
  // Stream walks (k_wf_stream) of the head and of the deferred section: lanes take the next ray as soon as theirs leaves
  // the section.  Measured on the final scene: one pipeline 1827 Msamples/s streamed against 1523 with the sections
  // walked per ray inside the world-program kernel, two pipelines 1906 against 1597; a section walked from global
  // memory gains more (stress scene: +34 %).  So sections are streamed by default; RTX_WF_STREAM=0 streams none,
  // 2 only the head, 3 only the deferred section.
  const char* stream_env = getenv("RTX_WF_STREAM");
  const int stream_mode = stream_env ? atoi(stream_env) : -1;  // -1: by size
  P.head_bytes = sc->dev.head_end ? (size_t)((sc->dev.head_end + 2u) & ~1u) * 8 : 0;
  P.head_lds = P.head_bytes <= kLdsBudget;
  P.tail_bytes = tail_begin ? (size_t)(n_units - tail_begin) * 8 : 0;
  P.tail_lds = P.tail_bytes <= kLdsBudget;
  // ... by size: a stream is one more launch per iteration, and a call whose queues are short is all launch latency
  // (final scene, Msamples/s streamed / not streamed: 16 M paths per call 807 / 896, 64 M 1400 / 1374, 256 M 1905 / 1698,
  // 1 G 2167 / 1818; `quick`: 924 / 1029).  Sections that are in LDS anyway are therefore streamed only when a pipeline's
  // share of the call is at least RTX_WF_STREAM_MIN camera paths (16 M); a section in global memory always is.
  unsigned long long stream_min = 16ull << 20;
  if (const char* e = getenv("RTX_WF_STREAM_MIN")) { const long long v = atoll(e); if (v >= 0) stream_min = (unsigned long long)v; }
  const bool big = share >= stream_min;
  const bool stream_head = sc->dev.head_end != 0 && (stream_mode < 0 ? (big || !P.head_lds) : (stream_mode == 1 || stream_mode == 2));
  P.stream_tail = tail_begin != 0 && (stream_mode < 0 ? (big || !P.tail_lds) : (stream_mode == 1 || stream_mode == 3));
  // Sections that are binary trees of boxes over leaf records can be walked nearer child first (walk_stream_near);
  // each lane's stack (one unit position per tree level) lives in LDS.  By default (RTX_WF_NEAR=1) the deferred
  // section is, and the head section when it is too large for LDS: the stress scene's BVH of 1 M spheres and quads
  // (20 levels, walked from global memory) goes from 49.6 to 76.9 Msamples/s, while the 400 boxes of the final
  // scene's ground are slower that way (10 sample rows: 1957 and 1963 Msamples/s against 2004 and 2026 in the fixed
  // order; most rays meet the ground from above, the first box they enter holds their hit in either order, and the
  // walk pays for two box tests per step and the stack).  RTX_WF_NEAR=0 keeps the reference's left-to-right order
  // everywhere, 2 walks every section of the shape nearer child first.
  const char* near_env = getenv("RTX_WF_NEAR");
  const int near_mode = near_env ? atoi(near_env) : 1;
  const bool near_ok = P.stream_tail && sc->dev.tail_near_depth > 0 && near_mode != 0;
  P.near_stack_bytes = (size_t)sc->dev.tail_near_depth * RT_TRAV_THREADS * sizeof(uint32_t);
  P.near_lds = near_ok && P.tail_lds && P.tail_bytes + P.near_stack_bytes <= kLdsBudget;
  P.near_global = near_ok && !P.tail_lds && P.near_stack_bytes <= kLdsBudget;
  P.near_quads = sc->dev.tail_near_quads != 0;
  const bool near_head_ok = stream_head && sc->dev.head_near_depth > 0 && (near_mode == 2 || (near_mode == 1 && !P.head_lds));
  P.near_head_stack = (size_t)sc->dev.head_near_depth * RT_TRAV_THREADS * sizeof(uint32_t);
  P.near_head_lds = near_head_ok && P.head_lds && P.head_bytes + P.near_head_stack <= kLdsBudget;
  P.near_head_global = near_head_ok && !P.head_lds && P.near_head_stack <= kLdsBudget;
  P.head_top_bytes = (size_t)(sc->dev.head_top_nodes ? sc->dev.head_top_units : 0u) * 8;
  const char* top_env = getenv("RTX_WF_TOP");   // 0: walk the whole head section from global memory
  // ... or all of it through the table that folds two levels into one record (walk_stream_wide; RTX_WF_WIDE=0: not),
  // whose stack takes three entries per record on the way down
  const char* wide_env = getenv("RTX_WF_WIDE");
  const size_t wide_stack = (size_t)sc->dev.head_wide_depth * 3u * RT_TRAV_THREADS * sizeof(uint32_t);
  const bool use_wide = stream_head && sc->dev.head_wide_nodes != 0 && near_mode != 0 && !P.head_lds && wide_stack <= kLdsBudget && !(wide_env && atoi(wide_env) == 0);
  if (use_wide) { P.near_head_global = true; P.near_head_stack = wide_stack; }
  if (!P.near_head_global || use_wide || P.near_head_stack + P.head_top_bytes > kLdsBudget || (top_env && atoi(top_env) == 0)) P.head_top_bytes = 0;
  const uint32_t head_end = stream_head ? sc->dev.head_end : 0u;
  P.head_end = head_end;
  P.dev = sc->dev;
  if (P.head_top_bytes == 0) P.dev.head_top_nodes = 0;
  if (!use_wide) P.dev.head_wide_nodes = 0;
  // testing knob: a wide margin sends many rays through the redo path (results must not change)
  if (const char* e = getenv("RTX_WF_NEAR_MARGIN")) { const double v = atof(e); if (v >= 1e-15 && v <= 1.0) P.dev.near_margin = v; }
  P.dev.head_end = head_end;
  // New camera paths are coherent (a wave holds one stratified sample of an 8 x 8 pixel tile): the per-ray walk with its
  // votes serves them better than the stream (final scene, camera rays only: 54.9 ms per 448 M rays through head and
  // world program in one walk against 25.5 + 42.3 ms streamed and then walked).  When the head section fits LDS next to
  // the rest of the world program, the head stream therefore covers the survivors only.  RTX_WF_FRESH=0 turns it off.
  const char* fresh_env = getenv("RTX_WF_FRESH");
  const bool fresh_in_main = head_end != 0 && !(fresh_env && atoi(fresh_env) == 0) &&
                             (size_t)(tail_begin ? tail_begin : n_units) * 8 + (tail_begin ? (size_t)(RT_TRAV_THREADS / 64) * RT_DEFER_SLOTS * sizeof(uint32_t) : 0) <= kLdsBudget;
  // ... and makes their camera rays itself instead of reading them back from a ray generation launch (one pipeline:
  // 2057 against 1947 Msamples/s; RTX_WF_FRESH=1 keeps the separate launch)
  P.dev.fresh_in_main = fresh_in_main ? ((fresh_env && atoi(fresh_env) == 1) ? 1u : 2u) : 0u;
  const uint32_t main_first = (head_end && !fresh_in_main && sc->n_general_media == 0) ? ((head_end + 1u) & ~1u) : 0u;
  P.main_bytes = (size_t)((tail_begin ? tail_begin : n_units) - main_first) * 8;
  P.defer_bytes = tail_begin ? (size_t)(RT_TRAV_THREADS / 64) * RT_DEFER_SLOTS * sizeof(uint32_t) : 0;
  P.main_lds = P.main_bytes + P.defer_bytes <= kLdsBudget;
  // the world-program kernel this scene needs: program in LDS or not, and what its media ask for
  const int mm = sc->dev.n_media == 0 ? MEDIA_NONE : (sc->n_general_media == 0 ? MEDIA_SPHERES : MEDIA_ALL);
  // without a head section every path starts at the start of the program anyway: the kernel only has to make the rays
  if (head_end == 0 && P.main_lds && !(fresh_env && (atoi(fresh_env) == 0 || atoi(fresh_env) == 1))) P.dev.fresh_in_main = 2u;
  const bool gen_in_main = P.dev.fresh_in_main == 2u;   // then main_lds holds (the whole world program fits LDS)
  // A world that is one large BVH is all head section: the head stream leaves every path's closest hit in the hit
  // arrays, and a world-program launch would only read and rewrite them (stress scene: 23 % of the GPU time).
  P.skip_main = head_end != 0 && P.dev.fresh_in_main == 0u && sc->nothing_behind_head;
  const bool flat = sc->dev.main_flat != 0 && mm != MEDIA_ALL && P.main_lds;   // (a program walked from global memory keeps the per-ray walk)
  if (flat) P.k_main = gen_in_main ? (mm == MEDIA_NONE ? k_wf_traverse<true, false, MEDIA_NONE, true, true> : k_wf_traverse<true, false, MEDIA_SPHERES, true, true>)
                                   : (mm == MEDIA_NONE ? k_wf_traverse<true, false, MEDIA_NONE, false, true> : k_wf_traverse<true, false, MEDIA_SPHERES, false, true>);
  else
  P.k_main = gen_in_main ? (mm == MEDIA_NONE ? k_wf_traverse<true, false, MEDIA_NONE, true> : mm == MEDIA_SPHERES ? k_wf_traverse<true, false, MEDIA_SPHERES, true> : k_wf_traverse<true, false, MEDIA_ALL, true>)
           : P.main_lds ? (mm == MEDIA_NONE ? k_wf_traverse<true, false, MEDIA_NONE> : mm == MEDIA_SPHERES ? k_wf_traverse<true, false, MEDIA_SPHERES> : k_wf_traverse<true, false, MEDIA_ALL>)
                        : (mm == MEDIA_NONE ? k_wf_traverse<false, false, MEDIA_NONE> : mm == MEDIA_SPHERES ? k_wf_traverse<false, false, MEDIA_SPHERES> : k_wf_traverse<false, false, MEDIA_ALL>);
  {
    const size_t whole_main = (size_t)(tail_begin ? tail_begin : n_units) * 8;
    const char* fe = getenv("RTX_WF_FINISH_LDS");   // 0: the finishing kernel walks the program in global memory
    if (whole_main <= kFinishLdsBudget && !(fe && atoi(fe) == 0)) {   // (the deferred section is walked where it lies)
      P.k_finish_lds = mm == MEDIA_NONE ? k_wf_finish_lds<MEDIA_NONE> : mm == MEDIA_SPHERES ? k_wf_finish_lds<MEDIA_SPHERES> : k_wf_finish_lds<MEDIA_ALL>;
      if (sc->dev.main_flat != 0 && mm != MEDIA_ALL) P.k_finish_lds = mm == MEDIA_NONE ? k_wf_finish_lds<MEDIA_NONE, true> : k_wf_finish_lds<MEDIA_SPHERES, true>;
      P.finish_lds_bytes = whole_main;
    }
  }
  P.finish_at = finish_threshold(sc->dev.head_end != 0 && !P.head_lds);
  const unsigned blocks256 = cap / 256;
  P.trav_blocks = (unsigned)ctx->n_cus;
  P.gen_blocks = std::min(blocks256, (unsigned)ctx->n_cus * 8u);
  P.shade_blocks = std::min(blocks256, (unsigned)ctx->n_cus * 16u);
  return P;
}

// One iteration on stream s: top the queue up, traverse (head section, world program, deferred section), shade.
// B.q[cur] is read, B.q[cur ^ 1] receives the survivors.  timed: HIP event pairs around the section launches (they
// create events when the context's pool is empty).
static void wf_launch_iteration(rtx_scene* sc, const WfPlan& P, const rtx_camera_desc* cam, const WfFrame& fr, unsigned long long seed,
                                int use_lights, const WfBuffers& B, int cur, hipStream_t s, double* accum_dev, bool timed) {
  rtx_context* ctx = sc->ctx;
  const uint32_t cap = P.cap;
  const Scene& dev = P.dev;
  const dim3 tb(P.trav_blocks), tt(RT_TRAV_THREADS);
  WfQueue qin = B.q[cur], qout = B.q[cur ^ 1];
  ctx->iterations += 1;
  hipLaunchKernelGGL(k_wf_prepare, dim3(1), dim3(1), 0, s, B.c, cap, fr.per_sample, ctx->d_counters, P.skip_main ? 1 : 0);
  if (dev.fresh_in_main != 2u) hipLaunchKernelGGL(k_wf_raygen, dim3(P.gen_blocks), dim3(256), 0, s, *cam, fr, seed, cap, B.c, qin);
  if (P.head_end) {  // the head section of the world program, streamed
    ctx->head_launches += 1;
    TimedSpan sh;
    const bool timed_head = timed && span_begin(ctx, s, sh, 3);
    if (P.near_head_lds)
      hipLaunchKernelGGL((k_wf_stream<true, false, 2>), tb, tt, P.head_bytes + P.near_head_stack, s, dev, cap, B.c, qin, B.hits, B.defer, ctx->d_counters);
    else if (P.near_head_global)
      hipLaunchKernelGGL((k_wf_stream<false, false, 2>), tb, tt, P.near_head_stack + P.head_top_bytes, s, dev, cap, B.c, qin, B.hits, B.defer, ctx->d_counters);
    else if (P.head_lds)
      hipLaunchKernelGGL((k_wf_stream<true, false, 0>), tb, tt, P.head_bytes, s, dev, cap, B.c, qin, B.hits, B.defer, ctx->d_counters);
    else
      hipLaunchKernelGGL((k_wf_stream<false, false, 0>), tb, tt, 0, s, dev, cap, B.c, qin, B.hits, B.defer, ctx->d_counters);
    if (timed_head) span_end(ctx, s, sh);
  }
  if (!P.skip_main)
    hipLaunchKernelGGL(P.k_main, tb, tt, (P.main_lds ? P.main_bytes : 0) + P.defer_bytes, s, dev, cap, seed, B.c, qin, B.hits, B.defer, ctx->d_counters, *cam, fr);
  if (P.tail_begin) {  // the deferred section for the rays that passed its root box
    ctx->tail_launches += 1;
    TimedSpan st;
    const bool timed_tail = timed && span_begin(ctx, s, st, 2);
    if (P.stream_tail && P.near_lds && P.near_quads)
      hipLaunchKernelGGL((k_wf_stream<true, true, 2>), tb, tt, P.tail_bytes + P.near_stack_bytes, s, dev, cap, B.c, qin, B.hits, B.defer, ctx->d_counters);
    else if (P.stream_tail && P.near_lds)
      hipLaunchKernelGGL((k_wf_stream<true, true, 1>), tb, tt, P.tail_bytes + P.near_stack_bytes, s, dev, cap, B.c, qin, B.hits, B.defer, ctx->d_counters);
    else if (P.stream_tail && P.near_global && P.near_quads)
      hipLaunchKernelGGL((k_wf_stream<false, true, 2>), tb, tt, P.near_stack_bytes, s, dev, cap, B.c, qin, B.hits, B.defer, ctx->d_counters);
    else if (P.stream_tail && P.near_global)
      hipLaunchKernelGGL((k_wf_stream<false, true, 1>), tb, tt, P.near_stack_bytes, s, dev, cap, B.c, qin, B.hits, B.defer, ctx->d_counters);
    else if (P.stream_tail && P.tail_lds)
      hipLaunchKernelGGL((k_wf_stream<true, true, 0>), tb, tt, P.tail_bytes, s, dev, cap, B.c, qin, B.hits, B.defer, ctx->d_counters);
    else if (P.stream_tail)
      hipLaunchKernelGGL((k_wf_stream<false, true, 0>), tb, tt, 0, s, dev, cap, B.c, qin, B.hits, B.defer, ctx->d_counters);
    else if (P.tail_lds)
      hipLaunchKernelGGL((k_wf_traverse<true, true, MEDIA_NONE>), tb, tt, P.tail_bytes, s, dev, cap, seed, B.c, qin, B.hits, B.defer, ctx->d_counters, *cam, fr);
    else
      hipLaunchKernelGGL((k_wf_traverse<false, true, MEDIA_NONE>), tb, tt, 0, s, dev, cap, seed, B.c, qin, B.hits, B.defer, ctx->d_counters, *cam, fr);
    if (timed_tail) span_end(ctx, s, st);
  }
  hipLaunchKernelGGL(k_wf_shade, dim3(P.shade_blocks), dim3(256), 0, s, sc->dev, *cam, cap, seed, use_lights, B.c, qin, B.hits, qout, accum_dev, ctx->d_counters, sc->mat_class);
}

// The wavefront loop: [prepare, ray generation, traversal, shading+compaction] per iteration until every camera path
// of the call has ended.  Iterations are enqueued in chunks; the counters of chunk k are inspected while chunk k+1 is
// already enqueued, so the GPU never waits for the host.  Kernels of an iteration with nothing to do exit at once.
//
// A large call is split by sample index between up to four such pipelines, one on the caller's stream and the others
// on streams of the library (forked from and joined to the caller's stream with events): each kernel of an iteration
// depends on the one before it, so a single pipeline leaves CUs idle while every launch drains; independent pipelines
// fill those gaps with each other's kernels.  Measured on the final scene (Msamples/s): 1 pipeline 1928, 2: 1906,
// 3 x 32 M paths in flight 1960, 4 x 16 M 1731, 4 x 32 M 1933, 4 x 64 M 2058, 5: 1691, 6: 1675.  Default: one pipeline per
// 4 M camera paths of the call, at most four, each holding at most 64 M paths (14.7 GB, 59 GB for four; 128 M paths
// each, RTX_WF_CAPACITY=134217728, takes 118 GB of the 288 GB and gains 2 %: 4 x 64 M 2189, 4 x 128 M 2231, 3 x 96 M 2212)
// and no more than its share of the call needs.  RTX_WF_LANES and RTX_WF_CAPACITY override both numbers.
// When device memory does not hold that workspace the capacity is halved until it does (down to 64 K paths).
static rtx_status render_wavefront(rtx_scene* sc, const rtx_camera_desc* cam, const rtx_render_opts* o, double* accum_dev, hipStream_t stream) {
  rtx_context* ctx = sc->ctx;
  const int n_samples = o->sample_end - o->sample_begin;
  const unsigned long long per_sample = (unsigned long long)(o->x1 - o->x0) * (unsigned long long)(o->y1 - o->y0);
  const unsigned long long call_paths = per_sample * (unsigned long long)std::max(n_samples, 0);
  const char* lanes_env = getenv("RTX_WF_LANES");
  // (a head section walked from global memory keeps most SIMDs busy by itself: two pipelines at most; stress scene with
  // 1 / 2 / 3 / 4 pipelines: 241 / 245 / 234 / 235 Msamples/s)
  const bool head_in_global = sc->dev.head_end != 0 && (size_t)((sc->dev.head_end + 2u) & ~1u) * 8 > kLdsBudget;
  const int want_lanes = lanes_env ? atoi(lanes_env) : (int)std::min<unsigned long long>(head_in_global ? 2ull : 4ull, (call_paths + (1ull << 22) - 1ull) >> 22);
  const int n_lanes = std::max(1, std::min(std::min(want_lanes, n_samples), RT_WF_MAX_LANES));
  uint32_t cap = 1u << 26;
  if (const char* e = getenv("RTX_WF_CAPACITY")) { long v = atol(e); if (v >= 1024 && v <= (1l << 27)) cap = (uint32_t)v; }
  {  // no more than a pipeline's share of the call needs (whole sample indices; at least 64 K)
    const unsigned long long share = per_sample * (unsigned long long)((n_samples + n_lanes - 1) / n_lanes);
    if (share < cap) cap = (uint32_t)std::max<unsigned long long>(share, 65536ull);
  }
  cap = (cap + 255u) & ~255u;
  if (per_sample + cap >= (1ull << 32)) return fail(ctx, RTX_ERR_UNSUPPORTED, "rtx_render: more than 2^32 pixels in one call");
  for (int l = 0; l < n_lanes;) {
    bool oom = false;
    rtx_status st = wf_ensure(ctx, l, cap, &oom);
    if (st == RTX_OK) { ++l; continue; }
    if (!oom || cap <= 65536u) return st;
    cap = std::max(65536u, (cap / 2u + 255u) & ~255u);   // every pipeline of the call works with the same capacity
    l = 0;
  }
  const int use_lights = (o->use_lights && sc->dev.has_lights) ? 1 : 0;
  const WfPlan P = wf_make_plan(sc, cap, per_sample * (unsigned long long)((n_samples + n_lanes - 1) / n_lanes));
  const unsigned long long seed = (unsigned long long)o->seed;

  struct Run { rtx_context::WfLane* L; hipStream_t s; WfFrame fr; int cur, k, iters, limit; bool done, closed; };
  Run runs[RT_WF_MAX_LANES];
  if (n_lanes > 1) HIP_TRY(ctx, hipEventRecord(ctx->wf_fork, stream));
  for (int l = 0; l < n_lanes; ++l) {
    Run& r = runs[l];
    r.L = &ctx->wf[l]; r.s = l == 0 ? stream : r.L->stream; r.cur = 0; r.k = 0; r.done = false; r.iters = 0; r.closed = false;
    // sample indices of this pipeline: an equal share of the call's range
    const int sb = o->sample_begin + (int)((long long)n_samples * l / n_lanes), se = o->sample_begin + (int)((long long)n_samples * (l + 1) / n_lanes);
    r.fr.x0 = o->x0; r.fr.y0 = o->y0; r.fr.w = o->x1 - o->x0; r.fr.h = o->y1 - o->y0; r.fr.sample_begin = sb; r.fr.per_sample = (uint32_t)per_sample;
    // a path of depth 0 is black (camera.rs:169-171): nothing to trace
    const unsigned long long total = cam->max_depth > 0 ? per_sample * (unsigned long long)(se - sb) : 0ull;
    // When the queue holds the pipeline's whole share, every camera path starts in the first iteration and none lives
    // longer than max_depth bounces, plus one iteration that drops the last dead entries: no iteration beyond that is
    // enqueued (a chunk of launches over empty queues is a quarter of a millisecond at the end of a small call).
    r.limit = total <= (unsigned long long)cap ? cam->max_depth + 1 : 0x7FFFFFFF;
    if (l > 0) HIP_TRY(ctx, hipStreamWaitEvent(r.s, ctx->wf_fork, 0));
    hipLaunchKernelGGL(k_wf_begin, dim3(1), dim3(1), 0, r.s, r.L->c, total);
  }
  // Iterations are enqueued round robin, one per pipeline at a time, so that every pipeline has work on the GPU within
  // the first few launches of the call (chunk after chunk per pipeline, a short call's last pipeline started when the
  // first was nearly through: the host needs about a millisecond to enqueue a chunk).  HIP event pairs around the
  // section launches only with a single pipeline: side by side the spans would overlap, and four event records per
  // iteration are a fifth of the host's work here.
  int chunk = 4;   // (8: quick 1081, 4: 1097-1114 Msamples/s; the large render does not care)
  if (const char* e = getenv("RTX_WF_CHUNK")) { const int v = atoi(e); if (v >= 2 && v <= 64) chunk = v & ~1; }   // even: the queues swap roles every iteration
  const bool timed_sections = n_lanes == 1;
  // persistent workgroups of the finishing kernel (one per CU for its LDS): the pipelines of a call reach that stage at
  // about the same time and share the CUs
  unsigned finish_blocks = std::max(16u, (unsigned)ctx->n_cus / (unsigned)n_lanes);
  if (const char* e = getenv("RTX_WF_FINISH_BLOCKS")) { const int v = atoi(e); if (v >= 1 && v <= 4096) finish_blocks = (unsigned)v; }
  for (bool all_done = false; !all_done;) {
    all_done = true;
    for (int it = 0; it < chunk; ++it)
      for (int l = 0; l < n_lanes; ++l) {
        Run& r = runs[l];
        if (r.done || r.closed || r.iters >= r.limit) continue;
        wf_launch_iteration(sc, P, cam, r.fr, seed, use_lights, *r.L, r.cur, r.s, accum_dev, timed_sections);
        r.cur ^= 1; r.iters += 1;
      }
    for (int l = 0; l < n_lanes; ++l) {
      Run& r = runs[l];
      if (r.done || r.closed) continue;
      rtx_context::WfLane& L = *r.L;
      hipStream_t s = r.s;
      if (P.k_finish_lds)
        hipLaunchKernelGGL(P.k_finish_lds, dim3(std::min<unsigned>((P.finish_at + RT_FINISH_THREADS - 1) / RT_FINISH_THREADS, finish_blocks)), dim3(RT_FINISH_THREADS), P.finish_lds_bytes, s, sc->dev, *cam, cap, seed, use_lights, (uint32_t)P.finish_at, L.c, L.q[r.cur], accum_dev, ctx->d_counters);
      else
        hipLaunchKernelGGL(k_wf_finish, dim3(P.finish_at / 256), dim3(256), 0, s, sc->dev, *cam, cap, seed, use_lights, (uint32_t)P.finish_at, L.c, L.q[r.cur], accum_dev, ctx->d_counters);
      HIP_TRY(ctx, hipGetLastError());
      HIP_TRY(ctx, hipMemcpyAsync(&L.host[r.k & 1], L.c, sizeof(WfCounters), hipMemcpyDeviceToHost, s));
      HIP_TRY(ctx, hipEventRecord(L.ev[r.k & 1], s));
    }
    for (int l = 0; l < n_lanes; ++l) {
      Run& r = runs[l];
      if (r.done) continue;
      rtx_context::WfLane& L = *r.L;
      // counters of chunk `kk`: is the pipeline through?  (finished: a finishing kernel ran the survivors to their end)
      auto look = [&](int kk) -> hipError_t {
        const hipError_t e = hipEventSynchronize(L.ev[kk & 1]);
        if (e != hipSuccess) return e;
        const WfCounters& c = L.host[kk & 1];
        r.done = (c.n_out == 0 || c.finished != 0) && c.next_path == c.total_paths;
        if (r.done) {  // the traversal launches of this pipeline, timed on the device
          ctx->traverse_ms += (double)c.k1_sum / ctx->wall_clock_khz;
          ctx->traverse_launches += c.k1_count;
          ctx->shade_ms += (double)c.sh_sum / ctx->wall_clock_khz;
          ctx->shade_launches += c.sh_count;
        }
        return hipSuccess;
      };
      if (r.k > 0) {
        HIP_TRY(ctx, look(r.k - 1));
        // The host runs one chunk ahead of the counters it has seen.  When those say that no camera path is left to
        // start and the survivors are few enough for the chunk just enqueued to end with the finishing kernel (a queue
        // shrinks to a tenth over four bounces), the next chunk would be nothing but empty launches - 64 of them at the
        // end of a small call, most of a millisecond: wait for the counters of the chunk just enqueued instead.
        const WfCounters& c = L.host[(r.k - 1) & 1];
        if (!r.done && c.next_path == c.total_paths && (unsigned long long)c.n_out <= 8ull * P.finish_at) HIP_TRY(ctx, look(r.k));
      }
      if (r.closed && !r.done) { r.closed = false; r.limit = 0x7FFFFFFF; }   // (not expected: go on chunk by chunk)
      else if (r.iters >= r.limit) r.closed = true;                          // only the counters of this chunk are still to come
      r.k += 1;
      if (!r.done) all_done = false;
      if (ctx->pending.size() > 2048) drain_events(ctx);
    }
  }
  // the caller's stream continues when every pipeline is through (the last chunk enqueued on each included)
  for (int l = 1; l < n_lanes; ++l) {
    HIP_TRY(ctx, hipEventRecord(runs[l].L->joined, runs[l].s));
    HIP_TRY(ctx, hipStreamWaitEvent(stream, runs[l].L->joined, 0));
  }
  return RTX_OK;
}

// Ordering of render calls of one context: the workspace, the counters and the statistics are the context's, so a
// call given another stream than the one before it must not start before that one has ended.
static hipError_t call_begin(rtx_context* ctx, hipStream_t stream) {
  return ctx->call_recorded ? hipStreamWaitEvent(stream, ctx->call_done, 0) : hipSuccess;
}
static hipError_t call_end(rtx_context* ctx, hipStream_t stream) {
  const hipError_t e = hipEventRecord(ctx->call_done, stream);
  if (e == hipSuccess) ctx->call_recorded = true;
  return e;
}

// ------------------------------------------------- the entry point that allocates nothing ---
// rtx_render_device_noalloc runs one wavefront pipeline whose whole workspace (both queues, the hit arrays, the defer
// list and the counters) lives in g_noalloc_scratch, a device array that is part of the library's code object: every
// GPU the library runs on holds its own copy from the moment the code object is loaded until the process ends.  Host
// code never uses the array's name as an address; noalloc_setup asks the runtime for the array's device address once
// per device and everything after that goes through that pointer.
#define RT_NOALLOC_COUNTER_BYTES 4096u
#define RT_NOALLOC_PER_PATH (2u * (10u * 8u + 3u * 4u) + 8u + 4u + 4u + 16u + 4u)   // two queues, hit (t, kind, ctx, uv), defer list
#define RT_NOALLOC_BYTES (RT_NOALLOC_COUNTER_BYTES + (size_t)RTX_NOALLOC_MAX_PATHS * RT_NOALLOC_PER_PATH)
static_assert(sizeof(WfCounters) <= RT_NOALLOC_COUNTER_BYTES, "counters outgrew their part of the scratch array");
static_assert(RTX_NOALLOC_MAX_PATHS % 256 == 0, "queue capacity must be a multiple of 256");
static __device__ __align__(256) unsigned char g_noalloc_scratch[RT_NOALLOC_BYTES];
#define RT_MAX_DEVICES 64
static unsigned char* g_noalloc_ptr[RT_MAX_DEVICES];   // device address of each GPU's copy of the array

static hipError_t noalloc_setup(int device) {   // with `device` current
  if (device < 0 || device >= RT_MAX_DEVICES) return hipErrorInvalidDevice;
  if (g_noalloc_ptr[device]) return hipSuccess;
  void* p = nullptr;
  const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_noalloc_scratch));
  if (e != hipSuccess) return e;
  g_noalloc_ptr[device] = (unsigned char*)p;
  return hipSuccess;
}
static WfBuffers noalloc_buffers(unsigned char* base) {
  const size_t cap = RTX_NOALLOC_MAX_PATHS;
  WfBuffers B;
  unsigned char* p = base;
  B.c = (WfCounters*)p; p += RT_NOALLOC_COUNTER_BYTES;
  for (int k = 0; k < 2; ++k) { B.q[k].f = (double*)p; p += cap * 10 * sizeof(double); }
  B.hits.t = (double*)p; p += cap * sizeof(double);
  B.hits.uv = (double*)p; p += cap * 2 * sizeof(double);
  for (int k = 0; k < 2; ++k) { B.q[k].u = (uint32_t*)p; p += cap * 3 * sizeof(uint32_t); }
  B.hits.kind_idx = (uint32_t*)p; p += cap * sizeof(uint32_t);
  B.hits.ctx = (uint32_t*)p; p += cap * sizeof(uint32_t);
  B.defer = (uint32_t*)p; p += cap * sizeof(uint32_t);
  return B;
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
  hipDeviceProp_t prop;
  if (hipSetDevice(dev) != hipSuccess || hipMalloc(&ctx->d_counters, 6 * sizeof(unsigned long long)) != hipSuccess ||
      hipMemset(ctx->d_counters, 0, 6 * sizeof(unsigned long long)) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    delete ctx;
    return fail(nullptr, RTX_ERR_HIP, "rtx_context_create: device initialisation failed");
  }
  ctx->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  hipError_t setup = wf_set_lds_limits();
  if (setup == hipSuccess) setup = hipEventCreateWithFlags(&ctx->call_done, hipEventDisableTiming);
  if (setup == hipSuccess) setup = noalloc_setup(dev);
  if (setup != hipSuccess) {
    const std::string why = std::string("rtx_context_create: device set-up failed: ") + hipGetErrorString(setup);
    rtx_context_destroy(ctx);
    return fail(nullptr, RTX_ERR_HIP, why);
  }
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0) ctx->wall_clock_khz = (double)khz;
  *out = ctx;
  return RTX_OK;
}

void rtx_context_destroy(rtx_context* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  drain_events(ctx);
  for (hipEvent_t e : ctx->free_events) (void)hipEventDestroy(e);
  for (auto& L : ctx->wf) {
    for (void* p : L.allocs) (void)hipFree(p);
    if (L.c) (void)hipFree(L.c);
    if (L.host) (void)hipHostFree(L.host);
    for (int k = 0; k < 2; ++k) if (L.ev[k]) (void)hipEventDestroy(L.ev[k]);
    if (L.joined) (void)hipEventDestroy(L.joined);
    if (L.stream) (void)hipStreamDestroy(L.stream);
  }
  if (ctx->wf_fork) (void)hipEventDestroy(ctx->wf_fork);
  if (ctx->call_done) (void)hipEventDestroy(ctx->call_done);
  if (ctx->host_accum_dev) (void)hipFree(ctx->host_accum_dev);
  if (ctx->d_counters) (void)hipFree(ctx->d_counters);
  delete ctx;
}

rtx_status rtx_scene_create(rtx_context* ctx, const rtx_scene_desc* d, rtx_scene** out) {
  if (!ctx || !d || !out) return fail(ctx, RTX_ERR_INVALID, "rtx_scene_create: null argument");
  *out = nullptr;
  if (const char* why = validate_scene(d)) return fail(ctx, RTX_ERR_INVALID, std::string("rtx_scene_create: ") + why);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  CompiledProgram prog;
  if (const char* e = getenv("RTX_MEDIA_DRAW_FIRST")) prog.draw_first_thin_media = atoi(e) != 0;   // tuning knob: 0 = thin media intersect first too
  if (const char* e = getenv("RTX_LIGHT_TABLE")) prog.light_table = atoi(e) != 0;                  // tuning knob: 0 = every light is evaluated
  if (const char* e = getenv("RTX_FLAT_WALK")) prog.flat_walk = atoi(e) != 0;                      // tuning knob: 0 = no wave-uniform sweep of flat stretches
  prog.compile(*d);
  rtx_scene* sc = new rtx_scene();
  sc->ctx = ctx;
  static_cast<rtx_scene_desc&>(sc->dev) = *d;
  sc->dev.nodes = nullptr; sc->dev.n_nodes = 0;
  sc->dev.world_begin = prog.world_begin; sc->dev.world_end = 0;
  sc->dev.n_units = prog.prog_units; sc->dev.tail_begin = prog.tail_begin;
  sc->dev.head_end = prog.head_end; sc->dev.tail_near_depth = prog.tail_near_depth; sc->dev.head_near_depth = prog.head_near_depth; sc->dev.tail_near_quads = prog.tail_near_quads; sc->dev._pad_near = 0; sc->dev.fresh_in_main = 0; sc->dev.light_quads = nullptr; sc->dev.light_spheres = nullptr; sc->dev.near_margin = RT_NEAR_MARGIN; sc->dev.head_top_begin = prog.head_top_begin; sc->dev.head_top_nodes = prog.head_top_nodes; sc->dev.head_top_units = prog.head_top_units;
  sc->dev.head_wide_begin = prog.head_wide_begin; sc->dev.head_wide_nodes = prog.head_wide_nodes; sc->dev.head_wide_depth = prog.head_wide_depth; sc->dev._pad_wide = 0; sc->dev.head_wide_far = prog.head_wide_far;
  sc->n_general_media = prog.n_general_media;
  sc->nothing_behind_head = false;
  if (prog.head_end && !prog.tail_begin && d->n_media == 0) {
    bool nothing = true;
    for (uint32_t i = prog.head_end; i < (uint32_t)prog.units.size() && nothing; i += rt::prog_record_units(prog.units[i])) {
      const uint32_t k = RTX_NODE_KIND((uint32_t)prog.units[i]);
      if (k == PROG_NODE_END) break;
      nothing = k == PROG_NODE_SYNC || k == PROG_NODE_NOP || k == RTX_NODE_ENTER || k == RTX_NODE_EXIT;
    }
    sc->nothing_behind_head = nothing;
  }
  sc->dev.light_nodes = nullptr; sc->dev.light_group_begin = nullptr;
  sc->dev.n_light_nodes = (uint32_t)prog.light_nodes.size();
  sc->dev.n_light_groups = prog.light_group_begin.empty() ? 0u : (uint32_t)prog.light_group_begin.size() - 1u;
  sc->dev.light_far = prog.light_far;
  sc->dev.main_flat = prog.main_flat ? 1u : 0u; sc->dev.flat_begin = prog.flat_begin;
  rtx_status st = upload(sc, prog.units.data(), prog.units.size(), &sc->dev.prog);
  if (st == RTX_OK) st = upload(sc, prog.light_nodes.data(), prog.light_nodes.size(), &sc->dev.light_nodes);
  if (st == RTX_OK) st = upload(sc, prog.light_group_begin.data(), prog.light_group_begin.size(), &sc->dev.light_group_begin);
  if (st == RTX_OK) st = upload(sc, prog.media.data(), prog.media.size(), &sc->dev.media);
#define UP(field, count) if (st == RTX_OK) st = upload(sc, d->field, (size_t)(count), &sc->dev.field)
  UP(boxes, d->n_boxes); UP(spheres, d->n_spheres); UP(quads, d->n_quads);
  UP(xforms, d->n_xforms); UP(xform_ops, d->n_xform_ops); UP(materials, d->n_materials);
  UP(textures, d->n_textures); UP(images, d->n_images); UP(texels, d->n_texel_bytes); UP(perlins, d->n_perlins);
  UP(lights, d->n_lights);
#undef UP
  if (st == RTX_OK) {  // material classes for the shading kernel's sort
    std::vector<uint8_t> mc(d->n_materials ? d->n_materials : 1u, (uint8_t)MC_END);
    for (uint32_t i = 0; i < d->n_materials; ++i) {
      const rtx_material& m = d->materials[i];
      if (m.kind == RTX_MAT_METAL) mc[i] = MC_METAL;
      else if (m.kind == RTX_MAT_DIELECTRIC) mc[i] = MC_DIELECTRIC;
      else if (m.kind == RTX_MAT_ISOTROPIC) mc[i] = MC_ISOTROPIC;
      else if (m.kind == RTX_MAT_LAMBERTIAN) {
        const int cost = texture_cost(d, m.tex, 0);
        mc[i] = cost == 2 ? MC_LAMBERT_NOISE : (cost == 1 ? MC_LAMBERT_IMAGE : MC_LAMBERT);
      }
    }
    st = upload(sc, mc.data(), mc.size(), &sc->mat_class);
  }
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

rtx_status rtx_render_device(rtx_scene* sc, const rtx_camera_desc* cam, const rtx_render_opts* o, double* accum_dev, void* hip_stream) {
  if (!sc || !cam || !o || !accum_dev) return fail(sc ? sc->ctx : nullptr, RTX_ERR_INVALID, "rtx_render_device: null argument");
  rtx_context* ctx = sc->ctx;
  if (const char* why = validate_render(cam, o)) return fail(ctx, RTX_ERR_INVALID, std::string("rtx_render_device: ") + why);
  if (o->x0 == o->x1 || o->y0 == o->y1 || o->sample_begin == o->sample_end) return RTX_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t stream = (hipStream_t)hip_stream;
  int pipeline = o->pipeline;
  if (pipeline == RTX_PIPELINE_DEFAULT) {
    const char* e = getenv("RTX_PIPELINE");
    pipeline = (e && atoi(e) == RTX_PIPELINE_FUSED) ? RTX_PIPELINE_FUSED : RTX_PIPELINE_WAVEFRONT;
  }
  if (cam->max_depth > 0xFFFF) pipeline = RTX_PIPELINE_FUSED;  // the wavefront state packs depth and bounce into 16 bits each
  HIP_TRY(ctx, call_begin(ctx, stream));
  TimedSpan sp;
  const bool timed = span_begin(ctx, stream, sp, 0);
  rtx_status st = RTX_OK;
  if (pipeline == RTX_PIPELINE_FUSED) {
    const int tiles_x = (o->x1 - o->x0 + 15) / 16, tiles_y = (o->y1 - o->y0 + 15) / 16;
    const int use_lights = (o->use_lights && sc->dev.has_lights) ? 1 : 0;
    hipLaunchKernelGGL(k_render_fused, dim3((unsigned)(tiles_x * tiles_y)), dim3(256), 0, stream, sc->dev, *cam, *o, use_lights, accum_dev, ctx->d_counters);
    HIP_TRY(ctx, hipGetLastError());
  } else {
    st = render_wavefront(sc, cam, o, accum_dev, stream);
  }
  if (timed) span_end(ctx, stream, sp);
  HIP_TRY(ctx, call_end(ctx, stream));
  if (ctx->pending.size() > 4096) drain_events(ctx);
  return st;
}

rtx_status rtx_render_device_noalloc(rtx_scene* sc, const rtx_camera_desc* cam, const rtx_render_opts* o, double* accum_dev, void* hip_stream) {
  if (!sc || !cam || !o || !accum_dev) return fail(sc ? sc->ctx : nullptr, RTX_ERR_INVALID, "rtx_render_device_noalloc: null argument");
  rtx_context* ctx = sc->ctx;
  if (const char* why = validate_render(cam, o)) return fail(ctx, RTX_ERR_INVALID, std::string("rtx_render_device_noalloc: ") + why);
  const unsigned long long per_sample = (unsigned long long)(o->x1 - o->x0) * (unsigned long long)(o->y1 - o->y0);
  const unsigned long long total = per_sample * (unsigned long long)(o->sample_end - o->sample_begin);
  if (total > (unsigned long long)RTX_NOALLOC_MAX_PATHS)
    return fail(ctx, RTX_ERR_INVALID, "rtx_render_device_noalloc: " + std::to_string(total) + " camera paths in one call, the limit is RTX_NOALLOC_MAX_PATHS = " +
                                          std::to_string(RTX_NOALLOC_MAX_PATHS) + " (pixels of the rectangle x sample indices): split the call, or use rtx_render_device");
  if (cam->max_depth > 0xFFFF) return fail(ctx, RTX_ERR_INVALID, "rtx_render_device_noalloc: max_depth above 65535");
  if (total == 0 || cam->max_depth <= 0) return RTX_OK;   // a path of depth 0 is black (camera.rs:169-171)
  unsigned char* base = (ctx->device >= 0 && ctx->device < RT_MAX_DEVICES) ? g_noalloc_ptr[ctx->device] : nullptr;
  if (!base) return fail(ctx, RTX_ERR_HIP, "rtx_render_device_noalloc: the scratch array of this device was not set up");
  if (o->pipeline == RTX_PIPELINE_FUSED) return fail(ctx, RTX_ERR_INVALID, "rtx_render_device_noalloc: runs the wavefront pipeline only (opts->pipeline must not be RTX_PIPELINE_FUSED)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t stream = (hipStream_t)hip_stream;
  // Under stream capture the launches become nodes of the caller's graph: the context's ordering event stays out of it
  // (recorded inside a capture it would tie later eager calls to that capture), and the caller orders launches of the
  // graph against other calls of this context.
  hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(stream, &capture) == hipSuccess && capture == hipStreamCaptureStatusActive;
  if (!capturing) HIP_TRY(ctx, call_begin(ctx, stream));
  const uint32_t cap = RTX_NOALLOC_MAX_PATHS;
  const WfBuffers B = noalloc_buffers(base);
  const WfPlan P = wf_make_plan(sc, cap, total);
  WfFrame fr;
  fr.x0 = o->x0; fr.y0 = o->y0; fr.w = o->x1 - o->x0; fr.h = o->y1 - o->y0; fr.sample_begin = o->sample_begin; fr.per_sample = (uint32_t)per_sample;
  const int use_lights = (o->use_lights && sc->dev.has_lights) ? 1 : 0;
  const unsigned long long seed = (unsigned long long)o->seed;
  // All camera paths of the call fit the queue at once: one iteration traces their first segment and shades it, and
  // k_wf_finish (threshold = the whole queue) runs every survivor to its end.  Nothing is read back, so the host
  // never waits and the sequence of launches does not depend on the data.  k_wf_begin clears every counter the array
  // kept from the call before; queue entries are written before they are read.
  hipLaunchKernelGGL(k_wf_begin, dim3(1), dim3(1), 0, stream, B.c, total);
  wf_launch_iteration(sc, P, cam, fr, seed, use_lights, B, 0, stream, accum_dev, false);
  hipLaunchKernelGGL(k_wf_finish, dim3(cap / 256), dim3(256), 0, stream, sc->dev, *cam, cap, seed, use_lights, cap, B.c, B.q[1], accum_dev, ctx->d_counters);
  HIP_TRY(ctx, hipGetLastError());
  if (!capturing) HIP_TRY(ctx, call_end(ctx, stream));
  return RTX_OK;
}

rtx_status rtx_render(rtx_scene* sc, const rtx_camera_desc* cam, const rtx_render_opts* o, double* accum_host) {
  if (!sc || !cam || !o || !accum_host) return fail(sc ? sc->ctx : nullptr, RTX_ERR_INVALID, "rtx_render: null argument");
  rtx_context* ctx = sc->ctx;
  if (const char* why = validate_render(cam, o)) return fail(ctx, RTX_ERR_INVALID, std::string("rtx_render: ") + why);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t n = (size_t)cam->image_width * (size_t)cam->image_height * 3;
  if (ctx->host_accum_len < n) {   // the device copy is the context's and only grows: no allocation per call
    HIP_TRY(ctx, hipDeviceSynchronize());
    if (ctx->host_accum_dev) (void)hipFree(ctx->host_accum_dev);
    ctx->host_accum_dev = nullptr; ctx->host_accum_len = 0;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->host_accum_dev, n * sizeof(double)));
    ctx->host_accum_len = n;
  }
  double* d = ctx->host_accum_dev;
  HIP_TRY(ctx, hipMemset(d, 0, n * sizeof(double)));
  rtx_status st = rtx_render_device(sc, cam, o, d, nullptr);
  if (st != RTX_OK) return st;
  if (hipDeviceSynchronize() != hipSuccess) return fail(ctx, RTX_ERR_HIP, std::string("rtx_render: kernel failed: ") + hipGetErrorString(hipGetLastError()));
  std::vector<double> tmp(n);
  HIP_TRY(ctx, hipMemcpy(tmp.data(), d, n * sizeof(double), hipMemcpyDeviceToHost));
  for (size_t k = 0; k < n; ++k) accum_host[k] += tmp[k];
  return RTX_OK;
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
  unsigned long long c[6] = {0, 0, 0, 0, 0, 0};
  HIP_TRY(ctx, hipMemcpy(c, ctx->d_counters, sizeof(c), hipMemcpyDeviceToHost));
  out->paths = c[0]; out->segments = c[1] - c[4]; out->pdf_probes = c[2]; out->roulette_drops = c[3]; out->order_redos = c[5];  // c[4]: dead entries the traversal walked
  out->render_ms = ctx->render_ms; out->launches = ctx->launches; out->last_kernel_ms = ctx->last_kernel_ms;
  out->traverse_ms = ctx->traverse_ms; out->traverse_launches = ctx->traverse_launches;
  out->tail_ms = ctx->tail_ms; out->tail_launches = ctx->tail_launches;
  out->head_ms = ctx->head_ms; out->head_launches = ctx->head_launches;
  out->shade_ms = ctx->shade_ms; out->shade_launches = ctx->shade_launches; out->iterations = ctx->iterations;
  return RTX_OK;
}

rtx_status rtx_reset_stats(rtx_context* ctx) {
  if (!ctx) return fail(ctx, RTX_ERR_INVALID, "rtx_reset_stats: null context");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  drain_events(ctx);
  HIP_TRY(ctx, hipMemset(ctx->d_counters, 0, 6 * sizeof(unsigned long long)));
  ctx->render_ms = 0.0; ctx->launches = 0; ctx->last_kernel_ms = 0.0; ctx->traverse_ms = 0.0; ctx->traverse_launches = 0; ctx->tail_ms = 0.0; ctx->tail_launches = 0; ctx->head_ms = 0.0; ctx->head_launches = 0; ctx->shade_ms = 0.0; ctx->shade_launches = 0; ctx->iterations = 0;
  return RTX_OK;
}

}  // extern "C"

#if defined(RT_PROF)
extern "C" int rtx_debug_prof_read(unsigned long long* out64, int reset) {
  if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
  if (reset) {
    unsigned long long zero[64] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), zero, sizeof(zero)) != hipSuccess) return -1;
  }
  return 0;
}
extern "C" int rtx_debug_wprof_read(unsigned long long* out32, int reset) {
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_wprof), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
  if (reset) {
    unsigned long long zero[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wprof), zero, sizeof(zero)) != hipSuccess) return -1;
  }
  return 0;
}
extern "C" int rtx_debug_sprof_read(unsigned long long* out32, int reset) {
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_sprof), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
  if (reset) {
    unsigned long long zero[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_sprof), zero, sizeof(zero)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
