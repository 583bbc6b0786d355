// CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into, loaded by, or
// called from the product library (librtx_amd.so); only tests/, the smoke
// check and the bench's cpu_baseline leg may load this file's shared object.
//
// A plain C++17 / f64 restatement of the reference renderer's hot path
// (Rukkhadevata123/RayTracing_Rust, recursive Camera::ray_color and everything
// below it).  It keeps the reference's shape on purpose: trait-like virtual
// classes (Hittable / Material / Texture / PDF), per-sample recursion, the
// left-then-right BVH walk, shrinking-tmax lists.  Each routine cites the
// reference file:line it follows (paths relative to the reference's src/).
//
// PARITY UNPINNED at the level of single values: the reference ships no tests,
// golden vectors or fixtures for this path and its toolchain (cargo/rustc) is
// absent, so no arithmetic result of this restatement can be checked against
// the reference's.  It is pinned by closed-form known-answer tests
// (tests/test_oracle_kat.py) and, statistically, by the five output images of
// this path's scenes the reference has checked in (tests/test_reference_images.py):
// images/cornell_box.png and images/output_cornell_smoke.png (deterministic
// geometry: renders of this oracle agree with their 30 x 30 block means to
// 2.4 / 2.6 of 255 on average and to 0.2 of 255 in the global mean),
// images/quick_test.png and images/output_final_scene_800x800_10000spp.png
// (random geometry per run: 10 x 10 block means within the spread of this
// build's own scene seeds), images/output_final_scene_800x800_5000spp.png
// (20 of 255 brighter than the other two of the same scene: layout only).
// Third-party arithmetic restated from its published semantics:
//   nalgebra 0.33.2 Vector3<f64>: dot = (x*x' + y*y') + z*z' (no FMA),
//   norm = sqrt(norm_squared), normalize = component / norm, textbook cross.
//   rand 0.9.0 thread RNG (unseeded ChaCha12) is replaced by seeded streams;
//   only distributions can match.
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math -shared -fPIC
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace orc {

static const double kPi = 3.14159265358979323846;  // std::f64::consts::PI
static const double kInf = std::numeric_limits<double>::infinity();

// ---------------------------------------------------------------- random ---
// utils/random.rs:9-25.  The reference draws from an unseeded thread RNG; the
// oracle uses an explicit xoshiro256** state reached through a thread-local
// pointer so that a render is reproducible per pixel.
struct Rng {
  uint64_t s[4];
  static uint64_t splitmix(uint64_t& x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  explicit Rng(uint64_t seed = 1) {
    uint64_t x = seed;
    for (int i = 0; i < 4; ++i) s[i] = splitmix(x);
  }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next_u64() {
    uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
  }
  double next_double() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
};

static thread_local Rng* tl_rng = nullptr;

struct Counters {
  uint64_t segments = 0;     // world.hit calls made by ray_color
  uint64_t pdf_probes = 0;   // Hittable::pdf_value calls on leaf lights
  uint64_t box_tests = 0;    // Aabb::hit calls
  uint64_t sphere_tests = 0; // Sphere::hit calls
  uint64_t quad_tests = 0;   // Quad::hit calls
  uint64_t paths = 0;
  void add(const Counters& o) {
    segments += o.segments; pdf_probes += o.pdf_probes; box_tests += o.box_tests;
    sphere_tests += o.sphere_tests; quad_tests += o.quad_tests; paths += o.paths;
  }
};
static thread_local Counters tl_cnt;
static thread_local bool tl_skip_media = false;  // first-hit AOV mode: media are stochastic, so skipped

// Replay mode, for tests that compare single draws with the device code: while a tape is set, random_double()
// returns the caller's numbers one after the other instead of the generator's, and notes beside each number what it
// was drawn for (kind | recursion level << 8 | medium number << 20), so that a test can hand the same numbers to code
// that draws them in another order.
enum { TAG_OTHER = 0, TAG_CAMERA = 1, TAG_SHADE = 3, TAG_ROULETTE = 4, TAG_SINGLE_INDEX = 5, TAG_MEDIUM = 6 };
struct Tape {
  const double* v; size_t n, pos; uint32_t* tags; size_t overrun;
};
static thread_local Tape* tl_tape = nullptr;
static thread_local uint32_t tl_tag = 0;          // kind | level << 8 of the draws being made
static thread_local std::vector<double>* tl_ray_log = nullptr;   // rays ray_color was called with (7 doubles each)
static inline uint32_t tape_tag(uint32_t kind) { return (tl_tag & 0xFFF00u) | kind; }
struct TagScope {                                 // sets the kind of the draws made while it lives
  uint32_t saved;
  explicit TagScope(uint32_t kind, uint32_t extra = 0) : saved(tl_tag) { tl_tag = tape_tag(kind) | extra; }
  ~TagScope() { tl_tag = saved; }
};

static inline double random_double() {                                                         // random.rs:10-13
  if (tl_tape) {
    Tape& t = *tl_tape;
    if (t.pos >= t.n) { t.overrun++; return 0.5; }
    if (t.tags) t.tags[t.pos] = tl_tag;
    return t.v[t.pos++];
  }
  return tl_rng->next_double();
}
static inline double random_double_range(double mn, double mx) { return mn + (mx - mn) * random_double(); }  // random.rs:16-19
static inline int random_int_range(int mn, int mx) {                                           // random.rs:22-25 (inclusive)
  int v = mn + (int)std::floor(random_double() * (double)(mx - mn + 1));
  return v > mx ? mx : v;
}
static inline double degrees_to_radians(double d) { return d * kPi / 180.0; }                   // random.rs:5-7

// Host-side seeded stream for random scene content (box heights, sphere
// positions, Perlin tables).  SplitMix64; the product's host layer documents
// and implements the same generator independently.
struct HostRng {
  uint64_t state;
  explicit HostRng(uint64_t seed) : state(seed) {}
  double next_double() { return (double)(Rng::splitmix(state) >> 11) * (1.0 / 9007199254740992.0); }
  double range(double mn, double mx) { return mn + (mx - mn) * next_double(); }
  int int_range(int mn, int mx) {
    int v = mn + (int)std::floor(next_double() * (double)(mx - mn + 1));
    return v > mx ? mx : v;
  }
};

// ------------------------------------------------------------------ vec3 ---
// math/vec3.rs:2-8 (nalgebra Vector3<f64> / Point3<f64>).
struct Vec3 {
  double x, y, z;
  Vec3() : x(0), y(0), z(0) {}
  Vec3(double a, double b, double c) : x(a), y(b), z(c) {}
  double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
typedef Vec3 Point3;
typedef Vec3 Color;
static inline Vec3 operator+(const Vec3& a, const Vec3& b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline Vec3 operator-(const Vec3& a, const Vec3& b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline Vec3 operator-(const Vec3& a) { return Vec3(-a.x, -a.y, -a.z); }
static inline Vec3 operator*(const Vec3& a, double s) { return Vec3(a.x * s, a.y * s, a.z * s); }
static inline Vec3 operator*(double s, const Vec3& a) { return Vec3(s * a.x, s * a.y, s * a.z); }
static inline Vec3 operator/(const Vec3& a, double s) { return Vec3(a.x / s, a.y / s, a.z / s); }
static inline double dot(const Vec3& a, const Vec3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline Vec3 cross(const Vec3& a, const Vec3& b) {
  return Vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline double norm_squared(const Vec3& a) { return dot(a, a); }
static inline double norm(const Vec3& a) { return std::sqrt(norm_squared(a)); }
static inline Vec3 normalize(const Vec3& a) { return a / norm(a); }
static inline Vec3 component_mul(const Vec3& a, const Vec3& b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }

static inline Vec3 random_range_vec(double mn, double mx) {                 // vec3.rs:23-29
  double a = random_double_range(mn, mx), b = random_double_range(mn, mx), c = random_double_range(mn, mx);
  return Vec3(a, b, c);
}
static inline Vec3 random_unit_vector() {                                   // vec3.rs:31-39
  for (;;) {
    Vec3 p = random_range_vec(-1.0, 1.0);
    double l2 = norm_squared(p);
    if (1e-160 < l2 && l2 <= 1.0) return p / std::sqrt(l2);
  }
}
static inline Vec3 random_in_unit_sphere() {                                // vec3.rs:41-48
  for (;;) {
    Vec3 p = random_range_vec(-1.0, 1.0);
    if (norm_squared(p) < 1.0) return p;
  }
}
static inline Vec3 random_in_unit_disk() {                                  // vec3.rs:50-61
  for (;;) {
    double a = random_double_range(-1.0, 1.0), b = random_double_range(-1.0, 1.0);
    Vec3 p(a, b, 0.0);
    if (norm_squared(p) < 1.0) return p;
  }
}
static inline Vec3 random_cosine_direction() {                              // vec3.rs:63-73
  double r1 = random_double(), r2 = random_double();
  double phi = 2.0 * kPi * r1;
  double x = std::cos(phi) * std::sqrt(r2);
  double y = std::sin(phi) * std::sqrt(r2);
  double z = std::sqrt(1.0 - r2);
  return Vec3(x, y, z);
}
static inline Vec3 reflect(const Vec3& v, const Vec3& n) { return v - n * 2.0 * dot(v, n); }  // vec3.rs:76-78
static inline Vec3 refract(const Vec3& uv, const Vec3& n, double eta) {     // vec3.rs:80-85
  double cos_theta = std::fmin(dot(-uv, n), 1.0);
  Vec3 r_out_perp = (uv + n * cos_theta) * eta;
  Vec3 r_out_parallel = n * -std::sqrt(std::fabs(1.0 - norm_squared(r_out_perp)));
  return r_out_perp + r_out_parallel;
}

// ------------------------------------------------------------------- ray ---
struct Ray {                                                                // math/ray.rs:4-19
  Point3 orig; Vec3 dir; double time;
  Ray() : time(0) {}
  Ray(const Point3& o, const Vec3& d, double t) : orig(o), dir(d), time(t) {}
  Point3 at(double t) const { return orig + dir * t; }
};

// -------------------------------------------------------------- interval ---
struct Interval {                                                           // math/interval.rs
  double min, max;
  Interval() : min(kInf), max(-kInf) {}                                    // empty(), 17-22 / Default 89-93
  Interval(double a, double b) : min(a), max(b) {}
  static Interval universe() { return Interval(-kInf, kInf); }             // 26-31
  double size() const { return max - min; }                                // 35-37
  bool contains(double x) const { return min <= x && x <= max; }           // 41-43
  bool surrounds(double x) const { return min < x && x < max; }            // 47-49
  double clamp(double x) const { return x < min ? min : (x > max ? max : x); }  // 53-61
  Interval expand(double delta) const { double p = delta / 2.0; return Interval(min - p, max + p); }  // 65-71
  Interval merge(const Interval& o) const { return Interval(std::fmin(min, o.min), std::fmax(max, o.max)); }  // 81-86
};

// ------------------------------------------------------------------ aabb ---
struct Aabb {                                                               // math/aabb.rs
  Interval x, y, z;
  Aabb() {}                                                                 // empty(), 46-52
  Aabb(const Interval& a, const Interval& b, const Interval& c) : x(a), y(b), z(c) {}
  static Aabb from_points(const Point3& a, const Point3& b) {               // new_point, 22-42
    Aabb r;
    r.x = a.x <= b.x ? Interval(a.x, b.x) : Interval(b.x, a.x);
    r.y = a.y <= b.y ? Interval(a.y, b.y) : Interval(b.y, a.y);
    r.z = a.z <= b.z ? Interval(a.z, b.z) : Interval(b.z, a.z);
    r.pad_to_minimums();
    return r;
  }
  const Interval& axis_interval(int a) const { return a == 0 ? x : (a == 1 ? y : z); }  // 66-73
  int longest_axis() const {                                                // 77-89
    double xs = x.size(), ys = y.size(), zs = z.size();
    if (xs >= ys && xs >= zs) return 0;
    if (ys >= zs) return 1;
    return 2;
  }
  void pad_to_minimums() {                                                  // 92-103
    const double DELTA = 0.0001;
    if (x.size() < DELTA) x = x.expand(DELTA);
    if (y.size() < DELTA) y = y.expand(DELTA);
    if (z.size() < DELTA) z = z.expand(DELTA);
  }
  Aabb merge(const Aabb& o) const { return Aabb(x.merge(o.x), y.merge(o.y), z.merge(o.z)); }  // 113-119
  Aabb offset(const Vec3& o) const {                                        // Add<Vec3>, 177-188
    return Aabb(Interval(x.min + o.x, x.max + o.x), Interval(y.min + o.y, y.max + o.y),
                Interval(z.min + o.z, z.max + o.z));
  }
  bool hit(const Ray& ray, Interval ray_t) const {                          // 132-167
    tl_cnt.box_tests++;
    for (int axis = 0; axis < 3; ++axis) {
      const Interval& ax = axis_interval(axis);
      double d = ray.dir[axis];
      if (std::fabs(d) < 1e-8) continue;
      double adinv = 1.0 / d;
      double o = ray.orig[axis];
      double t0 = (ax.min - o) * adinv;
      double t1 = (ax.max - o) * adinv;
      double t_min, t_max;
      if (adinv >= 0.0) { t_min = t0; t_max = t1; } else { t_min = t1; t_max = t0; }
      ray_t.min = std::fmax(ray_t.min, t_min);
      ray_t.max = std::fmin(ray_t.max, t_max);
      if (ray_t.max <= ray_t.min) return false;
    }
    return true;
  }
};

// ------------------------------------------------------------------- onb ---
struct ONB {                                                                // math/onb.rs:12-31
  Vec3 axis[3];
  explicit ONB(const Vec3& n) {
    axis[2] = normalize(n);
    Vec3 a = std::fabs(axis[2].x) > 0.9 ? Vec3(0, 1, 0) : Vec3(1, 0, 0);
    axis[1] = normalize(cross(axis[2], a));
    axis[0] = cross(axis[1], axis[2]);
  }
  const Vec3& w() const { return axis[2]; }
  Vec3 local_to_world(const Vec3& l) const { return axis[0] * l.x + axis[1] * l.y + axis[2] * l.z; }  // 53-55
};

// -------------------------------------------------------------- textures ---
struct Texture {                                                            // materials/texture/mod.rs:10-12
  virtual ~Texture() {}
  virtual Color value(double u, double v, const Point3& p) const = 0;
};
typedef std::shared_ptr<Texture> TexturePtr;

struct SolidColor : Texture {                                               // texture/solid_color.rs:18-22
  Color albedo;
  explicit SolidColor(const Color& c) : albedo(c) {}
  Color value(double, double, const Point3&) const override { return albedo; }
};

struct CheckerTexture : Texture {                                           // texture/checker.rs:16-22,41-56
  double inv_scale; TexturePtr even, odd;
  CheckerTexture(double scale, TexturePtr e, TexturePtr o) : inv_scale(1.0 / scale), even(e), odd(o) {}
  Color value(double u, double v, const Point3& p) const override {
    int xi = (int)std::floor(inv_scale * p.x);
    int yi = (int)std::floor(inv_scale * p.y);
    int zi = (int)std::floor(inv_scale * p.z);
    bool is_even = (xi + yi + zi) % 2 == 0;
    return is_even ? even->value(u, v, p) : odd->value(u, v, p);
  }
};

struct Perlin {                                                             // procedural/noise.rs
  Vec3 ranvec[256]; int perm_x[256], perm_y[256], perm_z[256];
  static void generate_perm(HostRng& rng, int* p) {                         // 80-93
    for (int i = 0; i < 256; ++i) p[i] = i;
    for (int i = 255; i >= 1; --i) { int target = rng.int_range(0, i); std::swap(p[i], p[target]); }
  }
  void init_from(HostRng& rng) {                                            // new(), 16-31 (seeded)
    for (int i = 0; i < 256; ++i) {
      double a = rng.range(-1.0, 1.0), b = rng.range(-1.0, 1.0), c = rng.range(-1.0, 1.0);
      ranvec[i] = normalize(Vec3(a, b, c));
    }
    generate_perm(rng, perm_x); generate_perm(rng, perm_y); generate_perm(rng, perm_z);
  }
  static double perlin_interp(const Vec3 c[2][2][2], double u, double v, double w) {  // 96-127
    double uu = u * u * (3.0 - 2.0 * u);
    double vv = v * v * (3.0 - 2.0 * v);
    double ww = w * w * (3.0 - 2.0 * w);
    double accum = 0.0;
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j)
        for (int k = 0; k < 2; ++k) {
          double i_f = i, j_f = j, k_f = k;
          Vec3 weight_v(u - i_f, v - j_f, w - k_f);
          double weight = (i_f * uu + (1.0 - i_f) * (1.0 - uu)) * (j_f * vv + (1.0 - j_f) * (1.0 - vv)) *
                          (k_f * ww + (1.0 - k_f) * (1.0 - ww));
          accum += weight * dot(c[i][j][k], weight_v);
        }
    return accum;
  }
  double noise(const Point3& p) const {                                     // 35-60
    double u = p.x - std::floor(p.x), v = p.y - std::floor(p.y), w = p.z - std::floor(p.z);
    int i = (int)std::floor(p.x), j = (int)std::floor(p.y), k = (int)std::floor(p.z);
    Vec3 c[2][2][2];
    for (int di = 0; di < 2; ++di)
      for (int dj = 0; dj < 2; ++dj)
        for (int dk = 0; dk < 2; ++dk) {
          int idx = perm_x[(i + di) & 255] ^ perm_y[(j + dj) & 255] ^ perm_z[(k + dk) & 255];
          c[di][dj][dk] = ranvec[idx];
        }
    return perlin_interp(c, u, v, w);
  }
  double turb(const Point3& p, int depth) const {                           // 64-76
    double accum = 0.0; Point3 temp_p = p; double weight = 1.0;
    for (int i = 0; i < depth; ++i) {
      accum += weight * noise(temp_p);
      weight *= 0.5;
      temp_p = temp_p * 2.0;
    }
    return std::fabs(accum);
  }
};

struct NoiseTexture : Texture {                                             // texture/noise.rs:29-35
  Perlin noise; double scale;
  Color value(double, double, const Point3& p) const override {
    double noise_value = 1.0 + std::sin(scale * p.z + 10.0 * noise.turb(p, 7));
    return Color(0.5, 0.5, 0.5) * noise_value;
  }
};

struct ImageTexture : Texture {                                             // texture/image.rs:58-88
  std::vector<uint8_t> rgb; uint32_t width = 0, height = 0;
  Color value(double u, double v, const Point3&) const override {
    if (height == 0) return Color(0.0, 1.0, 1.0);
    double uc = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);       // f64::clamp
    double vcl = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
    double vc = 1.0 - vcl;
    uint32_t i = (uint32_t)(uc * (double)width);
    uint32_t j = (uint32_t)(vc * (double)height);
    i = std::min(i, width - 1); j = std::min(j, height - 1);
    const uint8_t* px = &rgb[((size_t)j * width + i) * 3];
    double s = 1.0 / 255.0;
    return Color(s * (double)px[0], s * (double)px[1], s * (double)px[2]);
  }
};

// ------------------------------------------------------------- materials ---
struct Material;
struct HitRecord {                                                          // geometry/hittable.rs:9-17
  Point3 p; Vec3 normal; const Material* mat = nullptr; double t = 0, u = 0, v = 0; bool front_face = false;
  uint32_t prim_id = 0xFFFFFFFFu;   // oracle-only bookkeeping for the first-hit AOV
  void set_face_normal(const Ray& r, const Vec3& outward) {                 // 44-51
    front_face = dot(r.dir, outward) < 0.0;
    normal = front_face ? outward : -outward;
  }
};

struct PDF {                                                                // sampling/pdf/mod.rs:10-16
  virtual ~PDF() {}
  virtual double value(const Vec3& direction) const = 0;
  virtual Vec3 generate() const = 0;
};

struct ScatterRecord {                                                      // materials/material.rs:8-42
  Color attenuation; std::shared_ptr<PDF> pdf_ptr; bool skip_pdf = false; Ray skip_pdf_ray;
};

struct Material {                                                           // materials/material.rs:51-66
  uint32_t mat_id = 0;
  virtual ~Material() {}
  virtual bool scatter(const Ray& r_in, const HitRecord& rec, ScatterRecord& srec) const = 0;
  virtual Color emitted(double, double, const Point3&) const { return Color(0, 0, 0); }
  virtual double scattering_pdf(const Ray&, const HitRecord&, const Ray&) const { return 0.0; }
};
typedef std::shared_ptr<Material> MaterialPtr;

struct NoMaterial : Material {                                              // material.rs:69-82
  bool scatter(const Ray&, const HitRecord&, ScatterRecord&) const override { return false; }
  double scattering_pdf(const Ray&, const HitRecord&, const Ray&) const override { return 1.0; }
};

struct CosinePDF : PDF {                                                    // pdf/cosine_pdf.rs:21-31
  ONB uvw;
  explicit CosinePDF(const Vec3& n) : uvw(n) {}
  double value(const Vec3& d) const override {
    double c = dot(normalize(d), uvw.w());
    return std::fmax(0.0, c / kPi);
  }
  Vec3 generate() const override { return uvw.local_to_world(random_cosine_direction()); }
};

struct SpherePDF : PDF {                                                    // pdf/sphere_pdf.rs:16-27
  double value(const Vec3&) const override { return 1.0 / (4.0 * kPi); }
  Vec3 generate() const override { return random_unit_vector(); }
};

struct Lambertian : Material {                                              // materials/lambertian.rs:30-47
  TexturePtr albedo;
  explicit Lambertian(TexturePtr a) : albedo(a) {}
  bool scatter(const Ray&, const HitRecord& rec, ScatterRecord& srec) const override {
    srec.attenuation = albedo->value(rec.u, rec.v, rec.p);
    srec.pdf_ptr = std::make_shared<CosinePDF>(rec.normal);
    srec.skip_pdf = false;
    return true;
  }
  double scattering_pdf(const Ray&, const HitRecord& rec, const Ray& scattered) const override {
    double c = dot(rec.normal, normalize(scattered.dir));
    return c < 0.0 ? 0.0 : c / kPi;
  }
};

struct Metal : Material {                                                   // materials/metal.rs:16-38
  Color albedo; double fuzz;
  Metal(const Color& a, double f) : albedo(a), fuzz(f < 0.0 ? 0.0 : (f > 1.0 ? 1.0 : f)) {}
  bool scatter(const Ray& r_in, const HitRecord& rec, ScatterRecord& srec) const override {
    Vec3 reflected = reflect(normalize(r_in.dir), rec.normal);
    Vec3 scattered_dir = reflected + fuzz * random_in_unit_sphere();
    if (dot(scattered_dir, rec.normal) <= 0.0) return false;
    srec.attenuation = albedo; srec.skip_pdf = true; srec.pdf_ptr.reset();
    srec.skip_pdf_ray = Ray(rec.p, scattered_dir, r_in.time);
    return true;
  }
};

struct Dielectric : Material {                                              // materials/dielectric.rs:23-55
  double refraction_index;
  explicit Dielectric(double ri) : refraction_index(ri) {}
  static double reflectance(double cosine, double ratio) {
    double r0 = (1.0 - ratio) / (1.0 + ratio);
    double r0s = r0 * r0;
    double b = 1.0 - cosine;
    // f64::powi(5): the standard library lowers it to repeated multiplication
    // (square, square, times base).
    double b2 = b * b; double b4 = b2 * b2; double b5 = b4 * b;
    return r0s + (1.0 - r0s) * b5;
  }
  bool scatter(const Ray& r_in, const HitRecord& rec, ScatterRecord& srec) const override {
    double ri = rec.front_face ? 1.0 / refraction_index : refraction_index;
    Vec3 unit_direction = normalize(r_in.dir);
    double cos_theta = std::fmin(dot(-unit_direction, rec.normal), 1.0);
    double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
    bool cannot_refract = ri * sin_theta > 1.0;
    Vec3 direction;
    if (cannot_refract || reflectance(cos_theta, ri) > random_double()) direction = reflect(unit_direction, rec.normal);
    else direction = refract(unit_direction, rec.normal, ri);
    srec.attenuation = Color(1, 1, 1); srec.skip_pdf = true; srec.pdf_ptr.reset();
    srec.skip_pdf_ray = Ray(rec.p, direction, r_in.time);
    return true;
  }
};

struct DiffuseLight : Material {                                            // materials/diffuse_light.rs:29-40
  TexturePtr emit;
  explicit DiffuseLight(TexturePtr e) : emit(e) {}
  bool scatter(const Ray&, const HitRecord&, ScatterRecord&) const override { return false; }
  Color emitted(double u, double v, const Point3& p) const override { return emit->value(u, v, p); }
};

struct Isotropic : Material {                                               // materials/isotropic.rs:30-44
  TexturePtr albedo;
  explicit Isotropic(TexturePtr a) : albedo(a) {}
  bool scatter(const Ray&, const HitRecord& rec, ScatterRecord& srec) const override {
    srec.attenuation = albedo->value(rec.u, rec.v, rec.p);
    srec.pdf_ptr = std::make_shared<SpherePDF>();
    srec.skip_pdf = false;
    return true;
  }
  double scattering_pdf(const Ray&, const HitRecord&, const Ray&) const override { return 1.0 / (4.0 * kPi); }
};

// -------------------------------------------------------------- hittables ---
struct Hittable {                                                           // geometry/hittable.rs:97-115
  virtual ~Hittable() {}
  virtual bool hit(const Ray& r, Interval ray_t, HitRecord& rec) const = 0;
  virtual bool bounding_box(Aabb&) const { return false; }
  virtual double pdf_value(const Point3&, const Vec3&) const { return 0.0; }
  virtual Vec3 random(const Point3&) const { return Vec3(1.0, 0.0, 0.0); }
};
typedef std::shared_ptr<Hittable> HittablePtr;

struct Sphere : Hittable {                                                  // geometry/sphere.rs
  Ray center; double radius; MaterialPtr mat; Aabb bbox; uint32_t prim_id = 0;
  Sphere(const Point3& c, double r, MaterialPtr m) : center(c, Vec3(0, 0, 0), 0.0), radius(r), mat(m) {  // 22-32
    Vec3 rvec(r, r, r);
    bbox = Aabb::from_points(c - rvec, c + rvec);
  }
  Sphere(const Point3& c0, const Point3& c1, double r, MaterialPtr m) : center(c0, c1 - c0, 0.0), radius(r), mat(m) {  // 36-53
    Vec3 rvec(r, r, r);
    Aabb b0 = Aabb::from_points(c0 - rvec, c0 + rvec), b1 = Aabb::from_points(c1 - rvec, c1 + rvec);
    bbox = b0.merge(b1);
  }
  static void get_sphere_uv(const Vec3& p, double& u, double& v) {          // 57-66
    double theta = std::acos(-p.y);
    double phi = std::atan2(-p.z, p.x) + kPi;
    u = phi / (2.0 * kPi);
    v = theta / kPi;
  }
  Vec3 random_to_sphere(double distance_squared) const {                    // 69-79
    double r1 = random_double(), r2 = random_double();
    double z = 1.0 + r2 * (std::sqrt(1.0 - radius * radius / distance_squared) - 1.0);
    double phi = 2.0 * kPi * r1;
    double x = std::cos(phi) * std::sqrt(1.0 - z * z);
    double y = std::sin(phi) * std::sqrt(1.0 - z * z);
    return Vec3(x, y, z);
  }
  bool hit(const Ray& r, Interval ray_t, HitRecord& rec) const override {   // 83-118
    tl_cnt.sphere_tests++;
    Point3 current_center = center.at(r.time);
    Vec3 oc = r.orig - current_center;
    double a = norm_squared(r.dir);
    double half_b = dot(oc, r.dir);
    double c = norm_squared(oc) - radius * radius;
    double discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0) return false;
    double sqrtd = std::sqrt(discriminant);
    double root = (-half_b - sqrtd) / a;
    if (!ray_t.surrounds(root)) {
      root = (-half_b + sqrtd) / a;
      if (!ray_t.surrounds(root)) return false;
    }
    rec.t = root;
    rec.p = r.at(rec.t);
    Vec3 outward = (rec.p - current_center) / radius;
    get_sphere_uv(outward, rec.u, rec.v);
    rec.set_face_normal(r, outward);
    rec.mat = mat.get();
    rec.prim_id = prim_id;
    return true;
  }
  bool bounding_box(Aabb& b) const override { b = bbox; return true; }
  double pdf_value(const Point3& origin, const Vec3& direction) const override {  // 125-141
    tl_cnt.pdf_probes++;
    HitRecord rec;
    if (!hit(Ray(origin, direction, 0.0), Interval(0.001, kInf), rec)) return 0.0;
    Point3 cc = center.at(0.0);
    double dist_squared = norm_squared(cc - origin);
    double cos_theta_max = std::sqrt(1.0 - radius * radius / dist_squared);
    double solid_angle = 2.0 * kPi * (1.0 - cos_theta_max);
    return 1.0 / solid_angle;
  }
  Vec3 random(const Point3& origin) const override {                        // 143-150
    Point3 cc = center.at(0.0);
    Vec3 direction = cc - origin;
    double distance_squared = norm_squared(direction);
    ONB onb(direction);
    return onb.local_to_world(random_to_sphere(distance_squared));
  }
};

struct Quad : Hittable {                                                    // geometry/quad.rs
  Point3 q; Vec3 u, v; MaterialPtr mat; Aabb bbox; Vec3 normal; double d; Vec3 w; double area; uint32_t prim_id = 0;
  Quad(const Point3& q_, const Vec3& u_, const Vec3& v_, MaterialPtr m) : q(q_), u(u_), v(v_), mat(m) {  // 27-50
    Vec3 n = cross(u, v);
    normal = normalize(n);
    d = dot(normal, q);
    w = n / dot(n, n);
    area = norm(n);
    Aabb d1 = Aabb::from_points(q, q + u + v), d2 = Aabb::from_points(q + u, q + v);
    bbox = d1.merge(d2);
  }
  bool hit(const Ray& r, Interval ray_t, HitRecord& rec) const override {   // 66-95
    tl_cnt.quad_tests++;
    double denom = dot(normal, r.dir);
    if (std::fabs(denom) < 1e-8) return false;
    double t = (d - dot(normal, r.orig)) / denom;
    if (!ray_t.contains(t)) return false;
    Point3 intersection = r.at(t);
    Vec3 planar = intersection - q;
    double alpha = dot(w, cross(planar, v));
    double beta = dot(w, cross(u, planar));
    // is_interior, 54-62: (0.0..=1.0).contains(&a)
    if (!(0.0 <= alpha && alpha <= 1.0) || !(0.0 <= beta && beta <= 1.0)) return false;
    rec.u = alpha; rec.v = beta;
    rec.t = t; rec.p = intersection; rec.mat = mat.get();
    rec.set_face_normal(r, normal);
    rec.prim_id = prim_id;
    return true;
  }
  bool bounding_box(Aabb& b) const override { b = bbox; return true; }
  double pdf_value(const Point3& origin, const Vec3& direction) const override {  // 102-116
    tl_cnt.pdf_probes++;
    HitRecord rec;
    if (!hit(Ray(origin, direction, 0.0), Interval(0.001, kInf), rec)) return 0.0;
    double distance_squared = rec.t * rec.t * norm_squared(direction);
    double cosine = std::fabs(dot(direction, rec.normal) / norm(direction));
    return distance_squared / (cosine * area);
  }
  Vec3 random(const Point3& origin) const override {                        // 118-121
    double a = random_double();
    Vec3 t1 = a * u;
    double b = random_double();
    Point3 p = q + t1 + (b * v);
    return p - origin;
  }
};

struct HittableList : Hittable {                                            // geometry/hittable_list.rs
  std::vector<HittablePtr> objects; Aabb bbox;
  void add(HittablePtr o) {                                                 // 27-37
    Aabb b;
    if (o->bounding_box(b)) bbox = objects.empty() ? b : bbox.merge(b);
    objects.push_back(o);
  }
  bool hit(const Ray& r, Interval ray_t, HitRecord& rec) const override {   // 53-67
    HitRecord temp_rec; bool hit_anything = false; double closest = ray_t.max;
    for (const auto& o : objects) {
      if (o->hit(r, Interval(ray_t.min, closest), temp_rec)) {
        hit_anything = true; closest = temp_rec.t; rec = temp_rec;
      }
    }
    return hit_anything;
  }
  bool bounding_box(Aabb& b) const override { if (objects.empty()) return false; b = bbox; return true; }  // 70-76
  double pdf_value(const Point3& origin, const Vec3& direction) const override {  // 78-88
    if (objects.empty()) return 0.0;
    double weight = 1.0 / (double)objects.size();
    double sum = 0.0;
    for (const auto& o : objects) sum += weight * o->pdf_value(origin, direction);
    return sum;
  }
  Vec3 random(const Point3& origin) const override {                        // 90-97
    if (objects.empty()) return Vec3(1.0, 0.0, 0.0);
    int idx;
    {
      TagScope tag(objects.size() == 1 ? (uint32_t)TAG_SINGLE_INDEX : (tl_tag & 0xFFu));   // a draw with one possible outcome
      idx = random_int_range(0, (int)objects.size() - 1);
    }
    return objects[(size_t)idx]->random(origin);
  }
};

static std::shared_ptr<HittableList> box_new(const Point3& a, const Point3& b, MaterialPtr mat,
                                             std::vector<std::shared_ptr<Quad>>* made) {  // geometry/quad.rs:125-174
  auto sides = std::make_shared<HittableList>();
  Point3 mn(std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z));
  Point3 mx(std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z));
  Vec3 dx(mx.x - mn.x, 0, 0), dy(0, mx.y - mn.y, 0), dz(0, 0, mx.z - mn.z);
  auto add = [&](const Point3& q, const Vec3& u, const Vec3& v) {
    auto s = std::make_shared<Quad>(q, u, v, mat);
    if (made) made->push_back(s);
    sides->add(s);
  };
  add(Point3(mn.x, mn.y, mx.z), dx, dy);     // front
  add(Point3(mx.x, mn.y, mx.z), -dz, dy);    // right
  add(Point3(mx.x, mn.y, mn.z), -dx, dy);    // back
  add(Point3(mn.x, mn.y, mn.z), dz, dy);     // left
  add(Point3(mn.x, mx.y, mx.z), dx, -dz);    // top
  add(Point3(mn.x, mn.y, mn.z), dx, dz);     // bottom
  return sides;
}

struct Translate : Hittable {                                               // geometry/transforms/translate.rs
  HittablePtr object; Vec3 offset; Aabb bbox;
  Translate(HittablePtr o, const Vec3& off) : object(o), offset(off) {      // 18-31
    Aabb b; if (o->bounding_box(b)) bbox = b.offset(off);
  }
  bool hit(const Ray& r, Interval ray_t, HitRecord& rec) const override {   // 35-47
    Ray offset_r(r.orig - offset, r.dir, r.time);
    if (!object->hit(offset_r, ray_t, rec)) return false;
    rec.p = rec.p + offset;
    return true;
  }
  bool bounding_box(Aabb& b) const override { b = bbox; return true; }
  double pdf_value(const Point3& o, const Vec3& d) const override { return object->pdf_value(o - offset, d); }  // 55-59
  Vec3 random(const Point3& o) const override { return object->random(o - offset); }                           // 62-66
};

struct RotateY : Hittable {                                                 // geometry/transforms/rotate_y.rs
  HittablePtr object; double sin_theta, cos_theta; Aabb bbox;
  RotateY(HittablePtr o, double angle) : object(o) {                        // 20-67
    double radians = degrees_to_radians(angle);
    sin_theta = std::sin(radians); cos_theta = std::cos(radians);
    Aabb ob;
    if (o->bounding_box(ob)) {
      Point3 mn(kInf, kInf, kInf), mx(-kInf, -kInf, -kInf);
      for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 2; ++k) {
        double x = (double)i * ob.x.max + (double)(1 - i) * ob.x.min;
        double y = (double)j * ob.y.max + (double)(1 - j) * ob.y.min;
        double z = (double)k * ob.z.max + (double)(1 - k) * ob.z.min;
        double newx = cos_theta * x + sin_theta * z;
        double newz = -sin_theta * x + cos_theta * z;
        mn.x = std::fmin(mn.x, newx); mn.y = std::fmin(mn.y, y); mn.z = std::fmin(mn.z, newz);
        mx.x = std::fmax(mx.x, newx); mx.y = std::fmax(mx.y, y); mx.z = std::fmax(mx.z, newz);
      }
      bbox = Aabb::from_points(mn, mx);
    }
  }
  Vec3 w2l(const Vec3& p) const { return Vec3(cos_theta * p.x - sin_theta * p.z, p.y, sin_theta * p.x + cos_theta * p.z); }   // 71-77, 91-97
  Vec3 l2w(const Vec3& p) const { return Vec3(cos_theta * p.x + sin_theta * p.z, p.y, -sin_theta * p.x + cos_theta * p.z); }  // 81-87, 101-107
  bool hit(const Ray& r, Interval ray_t, HitRecord& rec) const override {   // 111-127
    Ray rotated(w2l(r.orig), w2l(r.dir), r.time);
    if (!object->hit(rotated, ray_t, rec)) return false;
    rec.p = l2w(rec.p);
    rec.normal = l2w(rec.normal);
    return true;
  }
  bool bounding_box(Aabb& b) const override { b = bbox; return true; }
  double pdf_value(const Point3& o, const Vec3& d) const override { return object->pdf_value(w2l(o), w2l(d)); }  // 135-140
  Vec3 random(const Point3& o) const override { return l2w(object->random(w2l(o))); }                            // 143-149
};

struct BvhNode : Hittable {                                                 // acceleration/bvh.rs
  HittablePtr left, right; Aabb bbox;
  static Aabb box_of(const HittablePtr& h) { Aabb b; h->bounding_box(b); return b; }  // unwrap_or_default → empty
  static bool box_less(const HittablePtr& a, const HittablePtr& b, int axis) {       // box_compare, 75-83
    return box_of(a).axis_interval(axis).min < box_of(b).axis_interval(axis).min;
  }
  BvhNode(const std::vector<HittablePtr>& objects, size_t start, size_t end) {        // from_slice, 26-71
    for (size_t i = start; i < end; ++i) { Aabb b; if (objects[i]->bounding_box(b)) bbox = bbox.merge(b); }
    size_t span = end - start;
    int axis = bbox.longest_axis();
    if (span == 1) {
      left = right = objects[start];
    } else if (span == 2) {
      if (box_less(objects[start], objects[start + 1], axis)) { left = objects[start]; right = objects[start + 1]; }
      else { left = objects[start + 1]; right = objects[start]; }
    } else {
      std::vector<HittablePtr> sorted(objects.begin() + (long)start, objects.begin() + (long)end);
      std::stable_sort(sorted.begin(), sorted.end(),
                       [axis](const HittablePtr& a, const HittablePtr& b) { return box_less(a, b, axis); });
      size_t mid = span / 2;
      left = std::make_shared<BvhNode>(sorted, 0, mid);
      right = std::make_shared<BvhNode>(sorted, mid, sorted.size());
    }
  }
  bool hit(const Ray& r, Interval ray_t, HitRecord& rec) const override {   // 87-105
    if (!bbox.hit(r, ray_t)) return false;
    bool hit_left = left->hit(r, ray_t, rec);
    Interval right_interval = hit_left ? Interval(ray_t.min, rec.t) : ray_t;
    bool hit_right = right->hit(r, right_interval, rec);
    return hit_left || hit_right;
  }
  bool bounding_box(Aabb& b) const override { b = bbox; return true; }
  double pdf_value(const Point3& o, const Vec3& d) const override {         // 113-116
    return 0.5 * (left->pdf_value(o, d) + right->pdf_value(o, d));
  }
  Vec3 random(const Point3& o) const override {                             // 119-126
    if (random_double() < 0.5) return left->random(o);
    return right->random(o);
  }
};

struct ConstantMedium : Hittable {                                          // volumes/constant_medium.rs
  HittablePtr boundary; MaterialPtr phase_function; double neg_inv_density;
  uint32_t medium_seq = 0;                                                  // number of this medium in its session (replay tags)
  ConstantMedium(HittablePtr b, double density, MaterialPtr iso) : boundary(b), phase_function(iso), neg_inv_density(-1.0 / density) {}
  bool hit(const Ray& r, Interval ray_t, HitRecord& rec) const override {   // 42-90
    if (tl_skip_media) return false;
    HitRecord rec1, rec2;
    if (!boundary->hit(r, Interval::universe(), rec1)) return false;
    if (!boundary->hit(r, Interval(rec1.t + 0.0001, kInf), rec2)) return false;
    rec1.t = std::fmax(rec1.t, ray_t.min);
    rec2.t = std::fmin(rec2.t, ray_t.max);
    if (rec1.t >= rec2.t) return false;
    rec1.t = std::fmax(rec1.t, 0.0);
    double ray_length = norm(r.dir);
    double distance_inside_boundary = (rec2.t - rec1.t) * ray_length;
    double xi;
    { TagScope tag(TAG_MEDIUM, medium_seq << 20); xi = random_double(); }
    double hit_distance = neg_inv_density * std::log(xi);
    if (hit_distance > distance_inside_boundary) return false;
    rec.t = rec1.t + hit_distance / ray_length;
    rec.p = r.at(rec.t);
    rec.normal = Vec3(1.0, 0.0, 0.0);
    rec.front_face = true;
    rec.mat = phase_function.get();
    rec.prim_id = 0xFFFFFFFEu;
    return true;
  }
  bool bounding_box(Aabb& b) const override { return boundary->bounding_box(b); }  // 93-95
};

// ------------------------------------------------------------------ pdfs ---
struct HittablePDF : PDF {                                                  // pdf/hittable_pdf.rs:23-33
  const Hittable* objects; Point3 origin;
  HittablePDF(const Hittable* o, const Point3& org) : objects(o), origin(org) {}
  double value(const Vec3& d) const override { return objects->pdf_value(origin, d); }
  Vec3 generate() const override { return objects->random(origin); }
};

struct MixturePDF : PDF {                                                   // pdf/mixture_pdf.rs:17-53
  const PDF* pdf1; const PDF* pdf2; double weight1, weight2;
  MixturePDF(const PDF* a, const PDF* b, double w1 = 0.5) : pdf1(a), pdf2(b) {
    weight1 = w1 < 0.0 ? 0.0 : (w1 > 1.0 ? 1.0 : w1);
    weight2 = 1.0 - weight1;
  }
  double value(const Vec3& d) const override {
    double value = weight1 * pdf1->value(d) + weight2 * pdf2->value(d);
    return value < 1e-8 ? 1e-8 : value;
  }
  Vec3 generate() const override {
    if (random_double() < weight1) return pdf1->generate();
    return pdf2->generate();
  }
};

// ---------------------------------------------------------------- camera ---
// Same field order as the product's rtx_camera_desc (declared independently).
struct CameraDesc {                                                         // rendering/camera.rs:16-49
  double aspect_ratio; int32_t image_width, samples_per_pixel, max_depth, _pad0;
  double background[3];
  double vfov, lookfrom[3], lookat[3], vup[3], defocus_angle, focus_dist;
  int32_t image_height, sqrt_spp;
  double pixel_samples_scale, recip_sqrt_spp;
  double center[3], pixel00_loc[3], pixel_delta_u[3], pixel_delta_v[3], u[3], v[3], w[3], defocus_disk_u[3], defocus_disk_v[3];
};
static inline Vec3 V(const double* a) { return Vec3(a[0], a[1], a[2]); }
static inline void S(double* a, const Vec3& v) { a[0] = v.x; a[1] = v.y; a[2] = v.z; }

static void camera_initialize(CameraDesc& c) {                              // camera.rs:89-129
  c.image_height = (int32_t)((double)c.image_width / c.aspect_ratio);
  c.image_height = std::max(c.image_height, 1);
  c.sqrt_spp = (int32_t)std::sqrt((double)c.samples_per_pixel);
  c.pixel_samples_scale = 1.0 / (double)(c.sqrt_spp * c.sqrt_spp);
  c.recip_sqrt_spp = 1.0 / (double)c.sqrt_spp;
  Vec3 lookfrom = V(c.lookfrom), lookat = V(c.lookat), vup = V(c.vup);
  Vec3 center = lookfrom;
  double theta = degrees_to_radians(c.vfov);
  double h = std::tan(theta / 2.0);
  double viewport_height = 2.0 * h * c.focus_dist;
  double viewport_width = viewport_height * ((double)c.image_width / (double)c.image_height);
  Vec3 w = normalize(lookfrom - lookat);
  Vec3 u = normalize(cross(vup, w));
  Vec3 v = cross(w, u);
  Vec3 viewport_u = viewport_width * u;
  Vec3 viewport_v = viewport_height * (-v);
  Vec3 pdu = viewport_u / (double)c.image_width;
  Vec3 pdv = viewport_v / (double)c.image_height;
  Vec3 upper_left = center - (c.focus_dist * w) - viewport_u / 2.0 - viewport_v / 2.0;
  Vec3 p00 = upper_left + 0.5 * (pdu + pdv);
  double defocus_radius = c.focus_dist * std::tan(degrees_to_radians(c.defocus_angle / 2.0));
  S(c.center, center); S(c.pixel00_loc, p00); S(c.pixel_delta_u, pdu); S(c.pixel_delta_v, pdv);
  S(c.u, u); S(c.v, v); S(c.w, w); S(c.defocus_disk_u, u * defocus_radius); S(c.defocus_disk_v, v * defocus_radius);
}

static Ray camera_get_ray(const CameraDesc& c, int i, int j, int s_i, int s_j) {  // camera.rs:133-164
  double ox = ((double)s_i + random_double()) * c.recip_sqrt_spp - 0.5;
  double oy = ((double)s_j + random_double()) * c.recip_sqrt_spp - 0.5;
  Vec3 pixel_sample = V(c.pixel00_loc) + (((double)i + ox) * V(c.pixel_delta_u)) + (((double)j + oy) * V(c.pixel_delta_v));
  Point3 origin;
  if (c.defocus_angle <= 0.0) origin = V(c.center);
  else {
    Vec3 p = random_in_unit_disk();
    origin = V(c.center) + (p.x * V(c.defocus_disk_u)) + (p.y * V(c.defocus_disk_v));
  }
  Vec3 direction = pixel_sample - origin;
  double time = random_double_range(0.0, 1.0);
  return Ray(origin, direction, time);
}

static Color ray_color(const CameraDesc& cam, const Ray& r, int depth, const Hittable* world, const Hittable* lights) {  // camera.rs:167-247
  if (depth <= 0) return Color(0, 0, 0);
  HitRecord rec;
  tl_cnt.segments++;
  if (tl_ray_log) { const double f[7] = {r.orig.x, r.orig.y, r.orig.z, r.dir.x, r.dir.y, r.dir.z, r.time}; tl_ray_log->insert(tl_ray_log->end(), f, f + 7); }
  tl_tag = (uint32_t)TAG_SHADE | ((uint32_t)(cam.max_depth - depth) << 8);   // replay tags only: no effect on the result
  if (!world->hit(r, Interval(0.001, kInf), rec)) return V(cam.background);
  Color emission = rec.mat->emitted(rec.u, rec.v, rec.p);
  ScatterRecord srec;
  if (!rec.mat->scatter(r, rec, srec)) return emission;
  if (srec.skip_pdf) return emission + component_mul(srec.attenuation, ray_color(cam, srec.skip_pdf_ray, depth - 1, world, lights));
  Vec3 scattered_direction; double pdf_value;
  if (lights) {
    HittablePDF light_pdf(lights, rec.p);
    MixturePDF mixture(&light_pdf, srec.pdf_ptr.get());
    scattered_direction = mixture.generate();
    pdf_value = mixture.value(scattered_direction);
  } else {
    scattered_direction = srec.pdf_ptr->generate();
    pdf_value = srec.pdf_ptr->value(scattered_direction);
  }
  if (pdf_value < 1e-6 || !std::isfinite(pdf_value)) return emission;
  Ray scattered(rec.p, scattered_direction, r.time);
  double scattering_pdf = rec.mat->scattering_pdf(r, rec, scattered);
  if (depth > 3) {
    double rr_prob = 0.8;
    double rr_draw;
    { TagScope tag(TAG_ROULETTE); rr_draw = random_double(); }
    if (rr_draw > rr_prob) return emission;
    double rr_scale = 1.0 / rr_prob;
    return emission + rr_scale * component_mul(srec.attenuation, scattering_pdf * ray_color(cam, scattered, depth - 1, world, lights)) / pdf_value;
  }
  return emission + component_mul(srec.attenuation, scattering_pdf * ray_color(cam, scattered, depth - 1, world, lights)) / pdf_value;
}

// ---------------------------------------------------------------- session ---
struct Session {
  std::vector<TexturePtr> textures;
  std::vector<MaterialPtr> materials;
  std::vector<HittablePtr> hittables;
  uint32_t next_prim = 0, next_medium = 0;
  std::string error;
  int add_tex(TexturePtr t) { textures.push_back(t); return (int)textures.size() - 1; }
  int add_mat(MaterialPtr m) { m->mat_id = (uint32_t)materials.size(); materials.push_back(m); return (int)materials.size() - 1; }
  int add_hit(HittablePtr h) { hittables.push_back(h); return (int)hittables.size() - 1; }
  bool ok_tex(int i) const { return i >= 0 && (size_t)i < textures.size(); }
  bool ok_mat(int i) const { return i >= 0 && (size_t)i < materials.size(); }
  bool ok_hit(int i) const { return i >= 0 && (size_t)i < hittables.size(); }
};

static bool load_ppm(const char* path, std::vector<uint8_t>& rgb, uint32_t& w, uint32_t& h) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  char magic[3] = {0, 0, 0}; int maxv = 0; unsigned ww = 0, hh = 0;
  bool ok = std::fscanf(f, "%2s %u %u %d", magic, &ww, &hh, &maxv) == 4 && std::strcmp(magic, "P6") == 0 && maxv == 255;
  if (ok) {
    std::fgetc(f);
    rgb.resize((size_t)ww * hh * 3);
    ok = std::fread(rgb.data(), 1, rgb.size(), f) == rgb.size();
  }
  std::fclose(f);
  if (ok) { w = ww; h = hh; }
  return ok;
}

}  // namespace orc

using namespace orc;

// ======================================================================= C API
extern "C" {

void* orc_session_new(void) { return new Session(); }
void orc_session_free(void* s) { delete (Session*)s; }

int orc_tex_solid(void* s, double r, double g, double b) { return ((Session*)s)->add_tex(std::make_shared<SolidColor>(Color(r, g, b))); }
int orc_tex_checker(void* s_, double scale, int even, int odd) {
  Session* s = (Session*)s_;
  if (!s->ok_tex(even) || !s->ok_tex(odd)) return -1;
  return s->add_tex(std::make_shared<CheckerTexture>(scale, s->textures[(size_t)even], s->textures[(size_t)odd]));
}
int orc_tex_noise(void* s, double scale, uint64_t seed) {
  auto t = std::make_shared<NoiseTexture>();
  HostRng rng(seed);
  t->noise.init_from(rng);
  t->scale = scale;
  return ((Session*)s)->add_tex(t);
}
int orc_tex_noise_tables(void* s, double scale, const double* ranvec, const int32_t* px, const int32_t* py, const int32_t* pz) {
  auto t = std::make_shared<NoiseTexture>();
  for (int i = 0; i < 256; ++i) {
    t->noise.ranvec[i] = Vec3(ranvec[3 * i], ranvec[3 * i + 1], ranvec[3 * i + 2]);
    t->noise.perm_x[i] = px[i]; t->noise.perm_y[i] = py[i]; t->noise.perm_z[i] = pz[i];
  }
  t->scale = scale;
  return ((Session*)s)->add_tex(t);
}
int orc_tex_image(void* s, uint32_t w, uint32_t h, const uint8_t* rgb) {
  auto t = std::make_shared<ImageTexture>();
  if (h > 0 && w > 0 && rgb) { t->width = w; t->height = h; t->rgb.assign(rgb, rgb + (size_t)w * h * 3); }
  return ((Session*)s)->add_tex(t);
}
int orc_tex_image_file(void* s, const char* filename) {                    // texture/image.rs:15-43 search order
  auto t = std::make_shared<ImageTexture>();
  std::string fn(filename);
  std::vector<std::string> paths = {fn, "textures/" + fn, "../textures/" + fn};
  if (const char* env = std::getenv("RTW_IMAGES")) paths.push_back(std::string(env) + "/" + fn);
  for (const auto& p : paths)
    if (load_ppm(p.c_str(), t->rgb, t->width, t->height)) break;
  return ((Session*)s)->add_tex(t);
}

int orc_mat_lambertian(void* s_, int tex) { Session* s = (Session*)s_; if (!s->ok_tex(tex)) return -1; return s->add_mat(std::make_shared<Lambertian>(s->textures[(size_t)tex])); }
int orc_mat_metal(void* s, double r, double g, double b, double fuzz) { return ((Session*)s)->add_mat(std::make_shared<Metal>(Color(r, g, b), fuzz)); }
int orc_mat_dielectric(void* s, double ior) { return ((Session*)s)->add_mat(std::make_shared<Dielectric>(ior)); }
int orc_mat_diffuse_light(void* s_, int tex) { Session* s = (Session*)s_; if (!s->ok_tex(tex)) return -1; return s->add_mat(std::make_shared<DiffuseLight>(s->textures[(size_t)tex])); }
int orc_mat_isotropic(void* s_, int tex) { Session* s = (Session*)s_; if (!s->ok_tex(tex)) return -1; return s->add_mat(std::make_shared<Isotropic>(s->textures[(size_t)tex])); }
int orc_mat_none(void* s) { return ((Session*)s)->add_mat(std::make_shared<NoMaterial>()); }

int orc_sphere(void* s_, const double* c, double r, int mat) {
  Session* s = (Session*)s_; if (!s->ok_mat(mat)) return -1;
  auto o = std::make_shared<Sphere>(V(c), r, s->materials[(size_t)mat]); o->prim_id = s->next_prim++;
  return s->add_hit(o);
}
int orc_sphere_moving(void* s_, const double* c0, const double* c1, double r, int mat) {
  Session* s = (Session*)s_; if (!s->ok_mat(mat)) return -1;
  auto o = std::make_shared<Sphere>(V(c0), V(c1), r, s->materials[(size_t)mat]); o->prim_id = s->next_prim++;
  return s->add_hit(o);
}
int orc_quad(void* s_, const double* q, const double* u, const double* v, int mat) {
  Session* s = (Session*)s_; if (!s->ok_mat(mat)) return -1;
  auto o = std::make_shared<Quad>(V(q), V(u), V(v), s->materials[(size_t)mat]); o->prim_id = s->next_prim++;
  return s->add_hit(o);
}
int orc_box(void* s_, const double* a, const double* b, int mat) {
  Session* s = (Session*)s_; if (!s->ok_mat(mat)) return -1;
  std::vector<std::shared_ptr<Quad>> made;
  auto l = box_new(V(a), V(b), s->materials[(size_t)mat], &made);
  for (auto& q : made) q->prim_id = s->next_prim++;
  return s->add_hit(l);
}
int orc_list_new(void* s) { return ((Session*)s)->add_hit(std::make_shared<HittableList>()); }
int orc_list_add(void* s_, int list, int obj) {
  Session* s = (Session*)s_; if (!s->ok_hit(list) || !s->ok_hit(obj)) return -1;
  auto l = std::dynamic_pointer_cast<HittableList>(s->hittables[(size_t)list]); if (!l) return -1;
  l->add(s->hittables[(size_t)obj]); return 0;
}
int orc_bvh(void* s_, int list) {
  Session* s = (Session*)s_; if (!s->ok_hit(list)) return -1;
  auto l = std::dynamic_pointer_cast<HittableList>(s->hittables[(size_t)list]); if (!l || l->objects.empty()) return -1;
  return s->add_hit(std::make_shared<BvhNode>(l->objects, 0, l->objects.size()));
}
int orc_translate(void* s_, int obj, const double* off) { Session* s = (Session*)s_; if (!s->ok_hit(obj)) return -1; return s->add_hit(std::make_shared<Translate>(s->hittables[(size_t)obj], V(off))); }
int orc_rotate_y(void* s_, int obj, double deg) { Session* s = (Session*)s_; if (!s->ok_hit(obj)) return -1; return s->add_hit(std::make_shared<RotateY>(s->hittables[(size_t)obj], deg)); }
int orc_medium(void* s_, int boundary, double density, int tex) {           // constant_medium.rs:22-38
  Session* s = (Session*)s_; if (!s->ok_hit(boundary) || !s->ok_tex(tex)) return -1;
  auto iso = std::make_shared<Isotropic>(s->textures[(size_t)tex]);
  s->add_mat(iso);
  auto medium = std::make_shared<ConstantMedium>(s->hittables[(size_t)boundary], density, iso);
  medium->medium_seq = s->next_medium++;
  return s->add_hit(medium);
}

// Bounding box of any hittable (6 doubles: xmin xmax ymin ymax zmin zmax); returns 0 when it has none.
int orc_bounding_box(void* s_, int obj, double* out) {
  Session* s = (Session*)s_; if (!s->ok_hit(obj)) return -1;
  Aabb b; if (!s->hittables[(size_t)obj]->bounding_box(b)) return 0;
  out[0] = b.x.min; out[1] = b.x.max; out[2] = b.y.min; out[3] = b.y.max; out[4] = b.z.min; out[5] = b.z.max;
  return 1;
}

void orc_camera_initialize(CameraDesc* c) { camera_initialize(*c); }

// Closest hit of `world` on n rays (7 doubles each: orig, dir, time) over [0.001, inf), as ray_color does.
// Outputs may be null.  Misses: t = +inf, prim_id = mat_id = 0xFFFFFFFF.
int orc_first_hit(void* s_, int world, const double* rays, size_t n, int skip_media, uint64_t seed,
                  double* t, double* normal, double* p, double* uv, uint32_t* prim_id, uint32_t* mat_id, uint8_t* front) {
  Session* s = (Session*)s_; if (!s->ok_hit(world)) return -1;
  const Hittable* w = s->hittables[(size_t)world].get();
  tl_skip_media = skip_media != 0;
  for (size_t i = 0; i < n; ++i) {
    Rng rng(seed * 0x9E3779B97F4A7C15ull + i); tl_rng = &rng;
    const double* r = rays + 7 * i;
    Ray ray(Point3(r[0], r[1], r[2]), Vec3(r[3], r[4], r[5]), r[6]);
    HitRecord rec;
    bool h = w->hit(ray, Interval(0.001, kInf), rec);
    if (t) t[i] = h ? rec.t : kInf;
    if (normal) { normal[3 * i] = h ? rec.normal.x : 0; normal[3 * i + 1] = h ? rec.normal.y : 0; normal[3 * i + 2] = h ? rec.normal.z : 0; }
    if (p) { p[3 * i] = h ? rec.p.x : 0; p[3 * i + 1] = h ? rec.p.y : 0; p[3 * i + 2] = h ? rec.p.z : 0; }
    if (uv) { uv[2 * i] = h ? rec.u : 0; uv[2 * i + 1] = h ? rec.v : 0; }
    if (prim_id) prim_id[i] = h ? rec.prim_id : 0xFFFFFFFFu;
    if (mat_id) mat_id[i] = h ? rec.mat->mat_id : 0xFFFFFFFFu;
    if (front) front[i] = h ? (rec.front_face ? 1 : 0) : 0;
  }
  tl_rng = nullptr; tl_skip_media = false;
  return 0;
}

// Hittable::pdf_value / Hittable::random probes for the light-sampling tests.
int orc_pdf_value(void* s_, int obj, const double* origin, const double* dirs, size_t n, double* out) {
  Session* s = (Session*)s_; if (!s->ok_hit(obj)) return -1;
  for (size_t i = 0; i < n; ++i) out[i] = s->hittables[(size_t)obj]->pdf_value(V(origin), V(dirs + 3 * i));
  return 0;
}
int orc_random_dirs(void* s_, int obj, const double* origin, uint64_t seed, size_t n, double* out) {
  Session* s = (Session*)s_; if (!s->ok_hit(obj)) return -1;
  Rng rng(seed); tl_rng = &rng;
  for (size_t i = 0; i < n; ++i) S(out + 3 * i, s->hittables[(size_t)obj]->random(V(origin)));
  tl_rng = nullptr; return 0;
}
int orc_tex_value(void* s_, int tex, const double* uvp, size_t n, double* out) {  // uvp: n x (u, v, px, py, pz)
  Session* s = (Session*)s_; if (!s->ok_tex(tex)) return -1;
  for (size_t i = 0; i < n; ++i) S(out + 3 * i, s->textures[(size_t)tex]->value(uvp[5 * i], uvp[5 * i + 1], V(uvp + 5 * i + 2)));
  return 0;
}
// CosinePDF / SpherePDF sampling and density (kind 0 = cosine about `normal`, 1 = uniform sphere).
int orc_pdf_sample(int kind, const double* normal, uint64_t seed, size_t n, double* dirs, double* values) {
  Rng rng(seed); tl_rng = &rng;
  std::unique_ptr<PDF> pdf;
  if (kind == 0) pdf.reset(new CosinePDF(V(normal))); else pdf.reset(new SpherePDF());
  for (size_t i = 0; i < n; ++i) { Vec3 d = pdf->generate(); S(dirs + 3 * i, d); values[i] = pdf->value(d); }
  tl_rng = nullptr; return 0;
}
// Material::scatter on a synthetic hit: returns 0 absorbed, 1 specular (out = ray dir), 2 diffuse.
int orc_mat_scatter(void* s_, int mat, const double* dir_in, const double* p, const double* normal, int front, uint64_t seed,
                    double* atten, double* out_dir) {
  Session* s = (Session*)s_; if (!s->ok_mat(mat)) return -1;
  Rng rng(seed); tl_rng = &rng;
  HitRecord rec; rec.p = V(p); rec.normal = V(normal); rec.front_face = front != 0; rec.t = 1.0;
  Ray rin(V(p) - V(dir_in), V(dir_in), 0.0);
  ScatterRecord srec;
  bool ok = s->materials[(size_t)mat]->scatter(rin, rec, srec);
  tl_rng = nullptr;
  if (!ok) return 0;
  S(atten, srec.attenuation);
  if (srec.skip_pdf) { S(out_dir, srec.skip_pdf_ray.dir); return 1; }
  return 2;
}

// ---- replay probes: the caller's numbers in place of the generator (see Tape above) ----
// One camera path: get_ray for (i, j, sample_idx) when ray7 is null, else ray_color on the given ray (orig, dir,
// time).  tags (n_tape entries, may be null) receives what each consumed number was drawn for; rays_out (7 doubles
// per call of ray_color that traced a segment, at most max_rays of them) the rays of the path in order.  Returns the
// number of draws asked for beyond the tape's end (0 = the tape sufficed), -1 on a bad handle.
long orc_tape_path(void* s_, int world, int lights, const CameraDesc* cam, int i, int j, int sample_idx, const double* ray7,
                   const double* tape, size_t n_tape, uint32_t* tags, size_t* consumed, double* rays_out, size_t max_rays,
                   size_t* n_rays, double* color) {
  Session* s = (Session*)s_; if (!s->ok_hit(world)) return -1;
  if (lights >= 0 && !s->ok_hit(lights)) return -1;
  const Hittable* w = s->hittables[(size_t)world].get();
  const Hittable* l = lights >= 0 ? s->hittables[(size_t)lights].get() : nullptr;
  Tape t{tape, n_tape, 0, tags, 0};
  std::vector<double> log;
  tl_tape = &t; tl_ray_log = &log; tl_tag = TAG_CAMERA;
  Ray ray;
  if (ray7) ray = Ray(Point3(ray7[0], ray7[1], ray7[2]), Vec3(ray7[3], ray7[4], ray7[5]), ray7[6]);
  else ray = camera_get_ray(*cam, i, j, sample_idx / cam->sqrt_spp, sample_idx % cam->sqrt_spp);   // camera.rs:262-264
  Color c = ray_color(*cam, ray, cam->max_depth, w, l);
  tl_tape = nullptr; tl_ray_log = nullptr; tl_tag = 0;
  if (consumed) *consumed = t.pos;
  const size_t nr = log.size() / 7;
  if (n_rays) *n_rays = nr;
  if (rays_out) std::memcpy(rays_out, log.data(), sizeof(double) * 7 * std::min(nr, max_rays));
  if (color) S(color, c);
  return (long)t.overrun;
}
// The random vectors of math/vec3.rs from a tape: kind 0 random_in_unit_sphere, 1 random_unit_vector,
// 2 random_cosine_direction, 3 random_in_unit_disk.
long orc_tape_vec(int kind, const double* tape, size_t n_tape, double* out3, size_t* consumed) {
  Tape t{tape, n_tape, 0, nullptr, 0};
  tl_tape = &t;
  Vec3 v = kind == 0 ? random_in_unit_sphere() : kind == 1 ? random_unit_vector() : kind == 2 ? random_cosine_direction() : random_in_unit_disk();
  tl_tape = nullptr;
  S(out3, v);
  if (consumed) *consumed = t.pos;
  return (long)t.overrun;
}
// Hittable::random(origin) from a tape (hittable_list.rs:90-97, quad.rs:118-121, sphere.rs:143-150).
long orc_tape_random(void* s_, int obj, const double* origin, const double* tape, size_t n_tape, uint32_t* tags, double* out3, size_t* consumed) {
  Session* s = (Session*)s_; if (!s->ok_hit(obj)) return -1;
  Tape t{tape, n_tape, 0, tags, 0};
  tl_tape = &t; tl_tag = TAG_SHADE;
  S(out3, s->hittables[(size_t)obj]->random(V(origin)));
  tl_tape = nullptr; tl_tag = 0;
  if (consumed) *consumed = t.pos;
  return (long)t.overrun;
}

struct RenderOpts {
  int32_t sample_begin, sample_end;   // stratified sample indices in [0, sqrt_spp^2)
  int32_t x0, y0, x1, y1;             // pixel rectangle [x0,x1) x [y0,y1)
  uint64_t seed;
  int32_t use_lights, n_threads;
};

// Camera::render (camera.rs:271-313) restricted to a sample range and a pixel rectangle.  Adds the per-pixel
// sums into accum (image_width*image_height*3 doubles, row-major, RGB).  counters_out (6 x u64) may be null:
// paths, segments, pdf_probes, box_tests, sphere_tests, quad_tests.
int orc_render(void* s_, int world, int lights, const CameraDesc* cam_in, const RenderOpts* o, double* accum, uint64_t* counters_out) {
  Session* s = (Session*)s_; if (!s->ok_hit(world)) return -1;
  if (lights >= 0 && !s->ok_hit(lights)) return -1;
  CameraDesc cam = *cam_in;
  const Hittable* w = s->hittables[(size_t)world].get();
  const Hittable* l = (lights >= 0 && o->use_lights) ? s->hittables[(size_t)lights].get() : nullptr;
  const int tile = 16;                                                      // camera.rs:288
  int ntx = (o->x1 - o->x0 + tile - 1) / tile, nty = (o->y1 - o->y0 + tile - 1) / tile;
  int total_tiles = ntx * nty;
  std::atomic<int> next(0);
  int nthreads = o->n_threads > 0 ? o->n_threads : (int)std::thread::hardware_concurrency();
  if (nthreads < 1) nthreads = 1;
  std::vector<Counters> cnts((size_t)nthreads);
  auto worker = [&](int tid) {
    tl_cnt = Counters();
    for (;;) {
      int ti = next.fetch_add(1);
      if (ti >= total_tiles) break;
      int tx = o->x0 + (ti % ntx) * tile, ty = o->y0 + (ti / ntx) * tile;
      for (int j = ty; j < std::min(ty + tile, o->y1); ++j)
        for (int i = tx; i < std::min(tx + tile, o->x1); ++i) {
          Rng rng(o->seed * 0x9E3779B97F4A7C15ull + (uint64_t)j * (uint64_t)cam.image_width + (uint64_t)i +
                  ((uint64_t)o->sample_begin << 40));
          tl_rng = &rng;
          Color sum(0, 0, 0);
          for (int idx = o->sample_begin; idx < o->sample_end; ++idx) {     // calculate_pixel_color, camera.rs:250-268
            int s_i = idx / cam.sqrt_spp, s_j = idx % cam.sqrt_spp;
            Ray ray = camera_get_ray(cam, i, j, s_i, s_j);
            tl_cnt.paths++;
            sum = sum + ray_color(cam, ray, cam.max_depth, w, l);
          }
          double* a = accum + ((size_t)j * (size_t)cam.image_width + (size_t)i) * 3;
          a[0] += sum.x; a[1] += sum.y; a[2] += sum.z;
        }
    }
    tl_rng = nullptr;
    cnts[(size_t)tid] = tl_cnt;
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nthreads; ++t) th.emplace_back(worker, t);
  worker(0);
  for (auto& t : th) t.join();
  if (counters_out) {
    Counters c; for (auto& x : cnts) c.add(x);
    counters_out[0] = c.paths; counters_out[1] = c.segments; counters_out[2] = c.pdf_probes;
    counters_out[3] = c.box_tests; counters_out[4] = c.sphere_tests; counters_out[5] = c.quad_tests;
  }
  return 0;
}

// ray_color on explicit rays (n x 7), `repeats` independent evaluations each, summed into out (n x 3).
int orc_ray_color(void* s_, int world, int lights, const CameraDesc* cam, const double* rays, size_t n, int repeats, uint64_t seed, double* out) {
  Session* s = (Session*)s_; if (!s->ok_hit(world)) return -1;
  const Hittable* w = s->hittables[(size_t)world].get();
  const Hittable* l = lights >= 0 ? s->hittables[(size_t)lights].get() : nullptr;
  for (size_t i = 0; i < n; ++i) {
    Rng rng(seed * 0x9E3779B97F4A7C15ull + i); tl_rng = &rng;
    const double* r = rays + 7 * i;
    Ray ray(Point3(r[0], r[1], r[2]), Vec3(r[3], r[4], r[5]), r[6]);
    Color sum(0, 0, 0);
    for (int k = 0; k < repeats; ++k) sum = sum + ray_color(*cam, ray, cam->max_depth, w, l);
    S(out + 3 * i, sum);
  }
  tl_rng = nullptr; return 0;
}

// color_to_rgb_with_samples, rendering/color.rs:7-52.
void orc_resolve_rgb8(const double* accum, int w, int h, int samples_per_pixel, uint8_t* rgb) {
  Interval intensity(0.000, 0.999);
  double scale = 1.0 / (double)samples_per_pixel;
  for (size_t i = 0; i < (size_t)w * (size_t)h * 3; ++i) {
    double c = accum[i];
    if (std::isnan(c)) c = 0.0;
    c *= scale;
    c = c > 0.0 ? std::sqrt(c) : 0.0;
    rgb[i] = (uint8_t)(256.0 * intensity.clamp(c));
  }
}

// ---------------------------------------------------------------- scenes ---
// Each builder returns the world handle and writes the lights handle and camera (public fields only;
// call orc_camera_initialize afterwards).  Random content comes from one HostRng(seed) in program order.
static void camera_defaults(CameraDesc& c) {                                // camera.rs:54-86
  std::memset(&c, 0, sizeof(c));
  c.aspect_ratio = 1.0; c.image_width = 100; c.samples_per_pixel = 10; c.max_depth = 10;
  c.background[0] = 0.7; c.background[1] = 0.8; c.background[2] = 1.0;
  c.vfov = 90.0; c.lookat[2] = -1.0; c.vup[1] = 1.0; c.defocus_angle = 0.0; c.focus_dist = 10.0;
}
static void d3(double* a, double x, double y, double z) { a[0] = x; a[1] = y; a[2] = z; }

static int lambert_color(void* s, double r, double g, double b) { return orc_mat_lambertian(s, orc_tex_solid(s, r, g, b)); }
static int light_color(void* s, double r, double g, double b) { return orc_mat_diffuse_light(s, orc_tex_solid(s, r, g, b)); }
static int quad3(void* s, double qx, double qy, double qz, double ux, double uy, double uz, double vx, double vy, double vz, int m) {
  double q[3] = {qx, qy, qz}, u[3] = {ux, uy, uz}, v[3] = {vx, vy, vz};
  return orc_quad(s, q, u, v, m);
}

// scenes/cornell_box.rs:35-104 (walls + light) with the light rectangle as parameters.
static void cornell_walls(void* s, int world, int lights, const double* lq, const double* lu, const double* lv, double emit) {
  int red = lambert_color(s, 0.65, 0.05, 0.05), white = lambert_color(s, 0.73, 0.73, 0.73), green = lambert_color(s, 0.12, 0.45, 0.15);
  int light = light_color(s, emit, emit, emit);
  orc_list_add(s, world, quad3(s, 555, 0, 0, 0, 555, 0, 0, 0, 555, green));
  orc_list_add(s, world, quad3(s, 0, 0, 0, 0, 555, 0, 0, 0, 555, red));
  orc_list_add(s, world, quad3(s, 0, 555, 0, 555, 0, 0, 0, 0, 555, white));
  orc_list_add(s, world, quad3(s, 0, 0, 0, 555, 0, 0, 0, 0, 555, white));
  orc_list_add(s, world, quad3(s, 0, 0, 555, 555, 0, 0, 0, 555, 0, white));
  orc_list_add(s, world, orc_quad(s, lq, lu, lv, light));
  orc_list_add(s, lights, orc_quad(s, lq, lu, lv, orc_mat_none(s)));
}
static void cornell_camera(CameraDesc* cam, int width, int spp, int depth) {  // cornell_box.rs:137-149
  camera_defaults(*cam);
  cam->aspect_ratio = 1.0; cam->image_width = width; cam->samples_per_pixel = spp; cam->max_depth = depth;
  d3(cam->background, 0, 0, 0);
  cam->vfov = 40.0; d3(cam->lookfrom, 278, 278, -800); d3(cam->lookat, 278, 278, 0); d3(cam->vup, 0, 1, 0);
  cam->defocus_angle = 0.0;
}

int orc_scene_cornell(void* s, int width, int spp, int depth, int* lights_out, CameraDesc* cam) {  // cornell_box.rs:107-149
  int world = orc_list_new(s), lights = orc_list_new(s);
  double lq[3] = {213, 554, 227}, lu[3] = {130, 0, 0}, lv[3] = {0, 0, 105};
  cornell_walls(s, world, lights, lq, lu, lv, 15.0);
  int white = lambert_color(s, 0.73, 0.73, 0.73);
  double a[3] = {0, 0, 0}, b[3] = {165, 330, 165}, off[3] = {265, 0, 295};
  orc_list_add(s, world, orc_translate(s, orc_rotate_y(s, orc_box(s, a, b, white), 15.0), off));
  double c[3] = {190, 90, 190};
  orc_list_add(s, world, orc_sphere(s, c, 90.0, orc_mat_dielectric(s, 1.5)));
  orc_list_add(s, lights, orc_sphere(s, c, 90.0, orc_mat_none(s)));
  cornell_camera(cam, width, spp, depth);
  *lights_out = lights;
  return world;
}

// Cornell smoke: no scene function exists in the reference (only images/output_cornell_smoke.png); authored from the
// reference's primitives: cornell walls, light quad (113,554,127) 330x305 emitting 7, two boxes as ConstantMedium
// boundaries (density 0.01): 165x330x165 @15deg -> (265,0,295) black, 165x165x165 @-18deg -> (130,0,65) white.
int orc_scene_cornell_smoke(void* s, int width, int spp, int depth, int* lights_out, CameraDesc* cam) {
  int world = orc_list_new(s), lights = orc_list_new(s);
  double lq[3] = {113, 554, 127}, lu[3] = {330, 0, 0}, lv[3] = {0, 0, 305};
  cornell_walls(s, world, lights, lq, lu, lv, 7.0);
  int white = lambert_color(s, 0.73, 0.73, 0.73);
  double a[3] = {0, 0, 0}, b1[3] = {165, 330, 165}, o1[3] = {265, 0, 295}, b2[3] = {165, 165, 165}, o2[3] = {130, 0, 65};
  int box1 = orc_translate(s, orc_rotate_y(s, orc_box(s, a, b1, white), 15.0), o1);
  int box2 = orc_translate(s, orc_rotate_y(s, orc_box(s, a, b2, white), -18.0), o2);
  orc_list_add(s, world, orc_medium(s, box1, 0.01, orc_tex_solid(s, 0, 0, 0)));
  orc_list_add(s, world, orc_medium(s, box2, 0.01, orc_tex_solid(s, 1, 1, 1)));
  cornell_camera(cam, width, spp, depth);
  *lights_out = lights;
  return world;
}

// scenes/final_scene.rs:41-196.  earth_tex: a texture handle made by the caller (image), or -1 for the missing-file cyan.
int orc_scene_final(void* s, int width, int spp, int depth, uint64_t seed, int earth_tex, int* lights_out, CameraDesc* cam) {
  HostRng rng(seed);
  int world = orc_list_new(s);
  int ground = lambert_color(s, 0.48, 0.83, 0.53);
  int boxes1 = orc_list_new(s);
  for (int i = 0; i < 20; ++i)
    for (int j = 0; j < 20; ++j) {
      double w = 100.0, x0 = -1000.0 + (double)i * w, z0 = -1000.0 + (double)j * w, y0 = 0.0;
      double x1 = x0 + w, y1 = rng.range(1.0, 101.0), z1 = z0 + w;
      double a[3] = {x0, y0, z0}, b[3] = {x1, y1, z1};
      orc_list_add(s, boxes1, orc_box(s, a, b, ground));
    }
  orc_list_add(s, world, orc_bvh(s, boxes1));
  int light = light_color(s, 7, 7, 7);
  orc_list_add(s, world, quad3(s, 123, 554, 147, 300, 0, 0, 0, 0, 265, light));
  double c1[3] = {400, 400, 200}, c2[3] = {430, 400, 200};
  orc_list_add(s, world, orc_sphere_moving(s, c1, c2, 50.0, lambert_color(s, 0.7, 0.3, 0.1)));
  double cg[3] = {260, 150, 45};
  orc_list_add(s, world, orc_sphere(s, cg, 50.0, orc_mat_dielectric(s, 1.5)));
  double cm[3] = {0, 150, 145};
  orc_list_add(s, world, orc_sphere(s, cm, 50.0, orc_mat_metal(s, 0.8, 0.8, 0.9, 1.0)));
  double cb[3] = {360, 150, 145};
  int boundary = orc_sphere(s, cb, 70.0, orc_mat_dielectric(s, 1.5));
  orc_list_add(s, world, boundary);
  orc_list_add(s, world, orc_medium(s, boundary, 0.2, orc_tex_solid(s, 0.2, 0.4, 0.9)));
  double c0[3] = {0, 0, 0};
  int fog_boundary = orc_sphere(s, c0, 5000.0, orc_mat_dielectric(s, 1.5));
  orc_list_add(s, world, orc_medium(s, fog_boundary, 0.0001, orc_tex_solid(s, 1, 1, 1)));
  if (earth_tex < 0) earth_tex = orc_tex_image(s, 0, 0, nullptr);
  double ce[3] = {400, 200, 400};
  orc_list_add(s, world, orc_sphere(s, ce, 100.0, orc_mat_lambertian(s, earth_tex)));
  // NoiseTexture::new(0.2): Perlin tables drawn from the scene stream (final_scene.rs:141).
  auto nt = std::make_shared<NoiseTexture>(); nt->noise.init_from(rng); nt->scale = 0.2;
  int noise_tex = ((Session*)s)->add_tex(nt);
  double cn[3] = {220, 280, 300};
  orc_list_add(s, world, orc_sphere(s, cn, 80.0, orc_mat_lambertian(s, noise_tex)));
  int boxes2 = orc_list_new(s);
  int white = lambert_color(s, 0.73, 0.73, 0.73);
  for (int k = 0; k < 1000; ++k) {
    // final_scene.rs:155-159 draws three random vectors and keeps x of the first, y of the second, z of the third.
    double d[9]; for (int q = 0; q < 9; ++q) d[q] = rng.range(0.0, 165.0);
    double c[3] = {d[0], d[4], d[8]};
    orc_list_add(s, boxes2, orc_sphere(s, c, 10.0, white));
  }
  double off[3] = {-100, 270, 395};
  orc_list_add(s, world, orc_translate(s, orc_rotate_y(s, orc_bvh(s, boxes2), 15.0), off));
  int lights = orc_list_new(s);
  orc_list_add(s, lights, quad3(s, 123, 554, 147, 300, 0, 0, 0, 0, 265, orc_mat_none(s)));
  camera_defaults(*cam);
  cam->aspect_ratio = 1.0; cam->image_width = width; cam->samples_per_pixel = spp; cam->max_depth = depth;
  d3(cam->background, 0, 0, 0);
  cam->vfov = 40.0; d3(cam->lookfrom, 478, 278, -600); d3(cam->lookat, 278, 278, 0); d3(cam->vup, 0, 1, 0);
  cam->defocus_angle = 0.0;
  *lights_out = lights;
  return world;
}

// Synthetic stress scene (not in the reference; the same recipe as the product's host layer, stated independently):
// primitive i is a sphere for even i (centre U[-1000,1000)^3, radius U[1,5)) and a quad for odd i (corner
// U[-1000,1000)^3, edges U[-5,5)^3); one draw picks the material: < 0.70 Lambertian of a random colour, < 0.85 Metal
// (albedo U[0.5,1)^3, fuzz U[0,0.5)), < 0.95 Dielectric 1.5, else an emitter of radiance 10 if it is a quad and grey
// Lambertian if not; the first 64 emitting quads are the lights.  Everything sits in one BvhNode.
int orc_scene_stress(void* s, int n_prims, uint64_t seed, int width, int spp, int depth, int* lights_out, CameraDesc* cam) {
  if (n_prims < 1) return -1;
  HostRng rng(seed);
  int all = orc_list_new(s), lights = orc_list_new(s), n_lights = 0;
  for (int i = 0; i < n_prims; ++i) {
    const bool quad = (i & 1) != 0;
    const double pick = rng.next_double();
    int mat;
    bool emitter = false;
    if (pick < 0.70) { double r = rng.next_double(), g = rng.next_double(), b = rng.next_double(); mat = lambert_color(s, r, g, b); }
    else if (pick < 0.85) { double r = rng.range(0.5, 1.0), g = rng.range(0.5, 1.0), b = rng.range(0.5, 1.0); mat = orc_mat_metal(s, r, g, b, rng.range(0.0, 0.5)); }
    else if (pick < 0.95) mat = orc_mat_dielectric(s, 1.5);
    else if (quad) { mat = orc_mat_diffuse_light(s, orc_tex_solid(s, 10, 10, 10)); emitter = true; }
    else mat = lambert_color(s, 0.5, 0.5, 0.5);
    int obj;
    if (!quad) {
      double c[3] = {rng.range(-1000, 1000), rng.range(-1000, 1000), rng.range(-1000, 1000)};
      obj = orc_sphere(s, c, rng.range(1.0, 5.0), mat);
    } else {
      double q[3] = {rng.range(-1000, 1000), rng.range(-1000, 1000), rng.range(-1000, 1000)};
      double u[3] = {rng.range(-5, 5), rng.range(-5, 5), rng.range(-5, 5)};
      double v[3] = {rng.range(-5, 5), rng.range(-5, 5), rng.range(-5, 5)};
      obj = orc_quad(s, q, u, v, mat);
      if (emitter && n_lights < 64) { orc_list_add(s, lights, obj); ++n_lights; }
    }
    if (obj < 0) return -1;
    orc_list_add(s, all, obj);
  }
  int world = orc_list_new(s);
  orc_list_add(s, world, orc_bvh(s, all));
  camera_defaults(*cam);
  cam->aspect_ratio = 16.0 / 9.0; cam->image_width = width; cam->samples_per_pixel = spp; cam->max_depth = depth;
  d3(cam->background, 0, 0, 0);
  cam->vfov = 40.0; d3(cam->lookfrom, 0, 0, -2500); d3(cam->lookat, 0, 0, 0); d3(cam->vup, 0, 1, 0);
  cam->defocus_angle = 0.0;
  *lights_out = n_lights ? lights : -1;
  return world;
}

}  // extern "C"
