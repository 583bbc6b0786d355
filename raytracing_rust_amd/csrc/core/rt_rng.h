// Counter-based path RNG.
//
// The reference draws every random number from an unseeded thread-local
// generator (src/ray_tracing/utils/random.rs:9-25), so its sequences cannot be
// reproduced; only the distributions matter.  Here every draw is a pure
// function of (seed, pixel, sample index, bounce/domain, draw counter) through
// Philox4x32 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3",
// 7 rounds = the Crush-resistant variant).  Consequences: no RNG state lives in
// memory between kernels, results do not depend on how paths are scheduled
// onto lanes, launches or GPUs, and two pipelines that consume draws in the
// same order produce the same paths.
#pragma once
#include "rt_math.h"

namespace rt {

#define RT_PHILOX_ROUNDS 7

struct PathRng {
  uint32_t k0, k1;      // key: render seed
  uint32_t pixel;       // counter word 3: pixel index (j * width + i)
  uint32_t sample;      // counter word 2: stratified sample index
  uint32_t stream;      // counter word 1: bounce | domain << 16
  uint32_t n;           // counter word 0: draw counter inside the stream
};

template <int ROUNDS>
RT_HD void philox4x32_r(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
RT_UNROLL
  for (int r = 0; r < ROUNDS; ++r) {
    uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
RT_HD void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
  philox4x32_r<RT_PHILOX_ROUNDS>(c0, c1, c2, c3, k0, k1, out);
}

RT_HD double u53(uint32_t hi, uint32_t lo) {  // uniform in [0,1) with 53 random bits, like rand's f64 sampling
  uint64_t x = ((uint64_t)hi << 32) | (uint64_t)lo;
  return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

enum { RNG_DOMAIN_SHADE = 0, RNG_DOMAIN_MEDIUM = 1, RNG_DOMAIN_CAMERA = 2 };

RT_HD PathRng rng_make(uint64_t seed, uint32_t pixel, uint32_t sample) {
  PathRng r;
  r.k0 = (uint32_t)seed; r.k1 = (uint32_t)(seed >> 32);
  r.pixel = pixel; r.sample = sample; r.stream = 0; r.n = 0;
  return r;
}
RT_HD void rng_stream(PathRng& r, uint32_t bounce, uint32_t domain) { r.stream = bounce | (domain << 16); r.n = 0; }

// Test-only replay (a host build may define RT_RNG_TAPE; the product never does): while a tape is set, the draws of
// the camera and shading streams come from the caller's numbers one after the other, the roulette draw of a bounce and
// the free-path draw of a medium from tables indexed the way those draws are addressed (bounce; bounce and medium).
#if defined(RT_RNG_TAPE) && !defined(__HIPCC__)
struct RngTape {
  const double* seq; size_t n_seq, pos, overrun;
  const double* roulette; uint32_t n_bounces;      // [bounce]
  const double* medium; uint32_t n_media;          // [bounce * n_media + medium]
};
inline thread_local RngTape* rt_rng_tape = nullptr;
inline double rng_tape_next() {
  RngTape& t = *rt_rng_tape;
  if (t.pos >= t.n_seq) { t.overrun++; return 0.5; }
  return t.seq[t.pos++];
}
#define RT_RNG_TAPE_ACTIVE (rt_rng_tape != nullptr)
#else
#define RT_RNG_TAPE_ACTIVE false
#endif

RT_HD double rng_next(PathRng& r) {  // random_double(), random.rs:10-13
#if defined(RT_RNG_TAPE) && !defined(__HIPCC__)
  if (RT_RNG_TAPE_ACTIVE) return rng_tape_next();
#endif
  uint32_t o[4];
  philox4x32(r.n++, r.stream, r.sample, r.pixel, r.k0, r.k1, o);
  return u53(o[0], o[1]);
}
RT_HD void rng_next2(PathRng& r, double& a, double& b) {  // two draws from one block
#if defined(RT_RNG_TAPE) && !defined(__HIPCC__)
  if (RT_RNG_TAPE_ACTIVE) { a = rng_tape_next(); b = rng_tape_next(); return; }
#endif
  uint32_t o[4];
  philox4x32(r.n++, r.stream, r.sample, r.pixel, r.k0, r.k1, o);
  a = u53(o[0], o[1]);
  b = u53(o[2], o[3]);
}
// One draw addressed by an explicit counter (used for media met during traversal: counter = node index).
RT_HD double rng_at(const PathRng& r, uint32_t stream, uint32_t n) {
#if defined(RT_RNG_TAPE) && !defined(__HIPCC__)
  if (RT_RNG_TAPE_ACTIVE && (stream >> 16) == RNG_DOMAIN_MEDIUM) {
    const uint32_t bounce = stream & 0xFFFFu;
    if (bounce < rt_rng_tape->n_bounces && n < rt_rng_tape->n_media) return rt_rng_tape->medium[(size_t)bounce * rt_rng_tape->n_media + n];
    rt_rng_tape->overrun++;
    return 0.5;
  }
#endif
  uint32_t o[4];
  philox4x32(n, stream, r.sample, r.pixel, r.k0, r.k1, o);
  return u53(o[0], o[1]);
}

// random_double_range, random.rs:16-19
RT_HD double rng_range(PathRng& r, double mn, double mx) { return mn + (mx - mn) * rng_next(r); }

// vec3.rs:31-39
RT_HD V3 random_unit_vector(PathRng& r) {
  for (;;) {
    double a, b;
    rng_next2(r, a, b);
    double c = rng_next(r);
    V3 p = v3(-1.0 + 2.0 * a, -1.0 + 2.0 * b, -1.0 + 2.0 * c);
    double l2 = norm2(p);
    if (1e-160 < l2 && l2 <= 1.0) return p / sqrt(l2);
  }
}
// vec3.rs:41-48
RT_HD V3 random_in_unit_sphere(PathRng& r) {
  for (;;) {
    double a, b;
    rng_next2(r, a, b);
    double c = rng_next(r);
    V3 p = v3(-1.0 + 2.0 * a, -1.0 + 2.0 * b, -1.0 + 2.0 * c);
    if (norm2(p) < 1.0) return p;
  }
}
// vec3.rs:50-61
RT_HD V3 random_in_unit_disk(PathRng& r) {
  for (;;) {
    double a, b;
    rng_next2(r, a, b);
    V3 p = v3(-1.0 + 2.0 * a, -1.0 + 2.0 * b, 0.0);
    if (norm2(p) < 1.0) return p;
  }
}
// vec3.rs:63-73
RT_HD V3 random_cosine_direction(PathRng& r) {
  double r1, r2;
  rng_next2(r, r1, r2);
  double phi = 2.0 * RT_PI * r1;
  double s = sqrt(r2);
  double sn, cs;
  rt_sincos(phi, sn, cs);
  double x = cs * s;
  double y = sn * s;
  double z = sqrt(1.0 - r2);
  return v3(x, y, z);
}

}  // namespace rt
