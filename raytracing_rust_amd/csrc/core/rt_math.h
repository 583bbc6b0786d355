// f64 vector arithmetic for the hit program and the integrator.
//
// Operation order follows the reference's nalgebra Vector3<f64> usage
// (src/ray_tracing/math/vec3.rs:2-8): dot = (x*x' + y*y') + z*z', norm =
// sqrt(norm_squared), normalize = component / norm, textbook cross.  The file
// must be compiled with floating-point contraction OFF so that no multiply-add
// pair is fused: first-hit results are compared bit for bit with a CPU build.
//
// RT_HD functions compile for gfx950 under hipcc and for the host under g++
// (the host build exists only so that tests can step through the same code
// without a GPU; the product library never runs it).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#define RT_HD_NOINLINE __host__ __device__ __noinline__
#define RT_UNROLL _Pragma("unroll")
#else
#define RT_HD inline
#define RT_HD_NOINLINE inline
#define RT_UNROLL
#endif

namespace rt {

#define RT_PI 3.14159265358979323846
#define RT_INF (__builtin_huge_val())

// sin and cos of one angle (vec3.rs:66-68, sphere.rs:74-76 call them one after the other): one argument reduction
// on the GPU instead of two.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void rt_sincos(double x, double& s, double& c) { ::sincos(x, &s, &c); }
#else
inline void rt_sincos(double x, double& s, double& c) { s = sin(x); c = cos(x); }
#endif

struct V3 {
  double x, y, z;
};

RT_HD V3 v3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD V3 v3(const double* a) { return v3(a[0], a[1], a[2]); }
RT_HD V3 operator+(const V3& a, const V3& b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD V3 operator-(const V3& a, const V3& b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD V3 operator-(const V3& a) { return v3(-a.x, -a.y, -a.z); }
RT_HD V3 operator*(const V3& a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
RT_HD V3 operator*(double s, const V3& a) { return v3(s * a.x, s * a.y, s * a.z); }
RT_HD V3 operator/(const V3& a, double s) { return v3(a.x / s, a.y / s, a.z / s); }
RT_HD V3 cmul(const V3& a, const V3& b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD double dot(const V3& a, const V3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_HD V3 cross(const V3& a, const V3& b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RT_HD double norm2(const V3& a) { return dot(a, a); }
RT_HD double norm(const V3& a) { return sqrt(norm2(a)); }
RT_HD V3 normalize(const V3& a) { return a / norm(a); }

// vec3.rs:76-78
RT_HD V3 reflect(const V3& v, const V3& n) { return v - n * 2.0 * dot(v, n); }
// vec3.rs:80-85
RT_HD V3 refract(const V3& uv, const V3& n, double eta) {
  double cos_theta = fmin(dot(-uv, n), 1.0);
  V3 perp = (uv + n * cos_theta) * eta;
  V3 par = n * -sqrt(fabs(1.0 - norm2(perp)));
  return perp + par;
}

// Rotation about Y as written in rotate_y.rs:71-107 (world->local and local->world forms).
RT_HD V3 rot_w2l(const V3& p, double s, double c) { return v3(c * p.x - s * p.z, p.y, s * p.x + c * p.z); }
RT_HD V3 rot_l2w(const V3& p, double s, double c) { return v3(c * p.x + s * p.z, p.y, -s * p.x + c * p.z); }

// onb.rs:12-31 / 53-55
struct Onb {
  V3 u, v, w;
};
RT_HD Onb onb_from(const V3& n) {
  Onb o;
  o.w = normalize(n);
  V3 a = fabs(o.w.x) > 0.9 ? v3(0.0, 1.0, 0.0) : v3(1.0, 0.0, 0.0);
  o.v = normalize(cross(o.w, a));
  o.u = cross(o.v, o.w);
  return o;
}
RT_HD V3 onb_local(const Onb& o, const V3& l) { return o.u * l.x + o.v * l.y + o.w * l.z; }

struct Ray {  // ray.rs:4-8
  V3 o, d;
  double tm;
};
RT_HD V3 ray_at(const V3& o, const V3& d, double t) { return o + d * t; }  // ray.rs:17-19

}  // namespace rt
