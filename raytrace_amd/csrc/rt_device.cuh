// rt_device.cuh — device-side building blocks of the path tracer (gfx950).
//
// Arithmetic contract: this file is compiled with -ffp-contract=off and HIP's
// correctly rounded f32 divide / sqrt, and every function below performs the
// IEEE binary32 operations of the reference shader (shaders/testFragment.fs)
// in the shader's order, so results are bit-identical to a scalar CPU
// evaluation of the same formulas.  Where GLSL leaves a built-in's rounding
// open the choice is: dot = (x*x' + y*y') + z*z' (as vec3.h:132-137),
// vector / scalar = (1/s) * v (vec3.h:127-130), normalize = v / sqrt(dot(v,v)),
// reflect / refract by the GLSL 3.30 formulas, pow(x,5) = ((x*x)*(x*x))*x.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/raytrace_amd.h"

#define RT_DEV __device__ __forceinline__

struct f3 {
  float x, y, z;
};

RT_DEV f3 mk3(float x, float y, float z) {
  f3 r;
  r.x = x;
  r.y = y;
  r.z = z;
  return r;
}
RT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
RT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_DEV f3 operator*(float s, f3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_DEV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_DEV f3 div3(f3 a, float s) { return (1.0f / s) * a; }

// The correctly rounded square root, as sqrtf() gives it under -fhip-fp32-correctly-rounded-divide-sqrt, in fewer
// instructions when every lane of the wave that is here holds an ordinary value.  The compiler's expansion scales
// arguments below 2^-96 up before the hardware's 1-ulp v_sqrt_f32 and the result back down, settles the last bit with
// two fused residuals (s -+ 1 ulp against x), and puts zeros and infinities back at the end: sixteen instructions.  For
// 2^-96 <= x < infinity the scaling and the fix-up are identities, and what is left — the middle of that same sequence,
// so the same bits — is nine.  One vote per call picks the sequence for the whole wave.
#ifndef RT_SHORT_SQRT
#define RT_SHORT_SQRT 1
#endif
RT_DEV float sqrt_rn(float x) {
#if RT_SHORT_SQRT
  const bool ordinary = (__float_as_uint(x) - 0x0F800000u) < (0x7F800000u - 0x0F800000u);  // also false for negatives and NaN
  if (__builtin_amdgcn_ballot_w64(!ordinary) != 0ull) return sqrtf(x);
  float s = __builtin_amdgcn_sqrtf(x);
  const float down = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
  const float r_down = __builtin_fmaf(-down, s, x), r_up = __builtin_fmaf(-up, s, x);
  s = r_down <= 0.0f ? down : s;
  s = r_up > 0.0f ? up : s;
  return s;
#else
  return sqrtf(x);
#endif
}
RT_DEV f3 normalize3(f3 v) { return div3(v, sqrt_rn(dot3(v, v))); }
RT_DEV f3 reflect3(f3 I, f3 N) { return I - (2.0f * dot3(N, I)) * N; }
RT_DEV f3 refract3(f3 I, f3 N, float eta) {
  float ni = dot3(N, I);
  float k = 1.0f - eta * eta * (1.0f - ni * ni);
  if (k < 0.0f) return mk3(0.0f, 0.0f, 0.0f);
  return eta * I - (eta * ni + sqrt_rn(k)) * N;
}
RT_DEV float pow5(float x) {
  float x2 = x * x;
  float x4 = x2 * x2;
  return x4 * x;
}

// ---- random numbers (testFragment.fs:326-345, 394-398) ---------------------

RT_DEV uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

// xorshift32 13/17/5 on the running state (fs:326-332).  Both RNG modes draw
// from it; they differ in where the state comes from: RT_RNG_XORSHIFT_REF seeds
// it once per pixel, RT_RNG_COUNTER re-keys it per (seed, pixel, sample).
template <int RNG>
RT_DEV uint32_t rng_next(uint32_t& a) {
  uint32_t x = a;
  x ^= x << 13;
  x ^= x >> 17;
  x ^= x << 5;
  a = x;
  return x;
}

template <int RNG>
RT_DEV float rng_float(uint32_t& a) {  // 23 mantissa bits -> [1,2) - 1
  uint32_t x = rng_next<RNG>(a);
  x = (x & 0x007FFFFFu) | 0x3F800000u;
  return __uint_as_float(x) - 1.0f;
}

RT_DEV uint32_t cantor_pair(uint32_t k1, uint32_t k2) {
  uint32_t x = (k1 + k2) * (k1 + k2 + 1u);
  x = x >> 1;
  return x + k2;
}

RT_DEV uint32_t counter_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
  uint32_t h = mix32(seed ^ mix32(pixel));
  h = mix32(h + sample * 0x9E3779B9u);
  return h ? h : 0x9E3779B9u;  // zero is xorshift's fixed point
}

// Per-lane work counters, kept in registers and flushed once per lane.
struct lane_counters {
  uint32_t paths, segments, sphere_tests, roots, node_tests, hits, sky;
  uint32_t lambertian, metallic, dialectric, perlin, unit_tries, disk_tries, limit_hits;
  uint32_t trav_wave_steps, trav_lane_steps, path_wave_phases, path_lane_phases, leaf_wave_steps, leaf_lane_steps;
};
#define RT_COUNT(c, field, n) \
  do {                        \
    if (COUNTED) (c).field += (n); \
  } while (0)

// Rejection sampler of testFragment.fs:351-366.  The candidate lies in [0,1)^3
// (only the positive octant is sampled) and the state is bumped by the try
// index before every try.
template <int RNG, bool COUNTED>
RT_DEV f3 random_unit_vector(uint32_t& a, lane_counters& k) {
  // The accepted candidate is normalised after the loop, once for the whole wave: inside the
  // loop the square root and the divide would be issued again in every iteration in which some
  // lane accepts.
  uint32_t i = 0u;
  f3 p;
  float lensq;
  while (true) {
    a += i;
    float x = rng_float<RNG>(a);
    float y = rng_float<RNG>(a);
    float z = rng_float<RNG>(a);
    p = mk3(x, y, z);
    lensq = dot3(p, p);
    RT_COUNT(k, unit_tries, 1);
    if (0.0f < lensq && lensq <= 1.0f) break;
    i += 1u;
  }
  return div3(p, sqrt_rn(lensq));
}

// testFragment.fs:368-383: point of the unit quarter-disk, not normalised.
template <int RNG, bool COUNTED>
RT_DEV f3 random_unit_disk(uint32_t& a, lane_counters& k) {
  uint32_t i = 0u;
  while (true) {
    a += i;
    float x = rng_float<RNG>(a);
    float y = rng_float<RNG>(a);
    float lensq = (x * x + y * y) + 0.0f * 0.0f;
    RT_COUNT(k, disk_tries, 1);
    if (lensq <= 1.0f) return mk3(x, y, 0.0f);
    i += 1u;
  }
}

template <int RNG, bool COUNTED>
RT_DEV f3 random_on_hemisphere(uint32_t& a, f3 normal, lane_counters& k) {  // fs:385-392
  f3 u = random_unit_vector<RNG, COUNTED>(a, k);
  if (dot3(u, normal) >= 0.0f) return u;
  return -u;
}

RT_DEV bool near_zero3(f3 v) {  // fs:400-403
  const float s = 1e-8f;
  return fabsf(v.x) < s && fabsf(v.y) < s && fabsf(v.z) < s;
}

// ---- sphere intersection (testFragment.fs:143-163) --------------------------

// Roots of one sphere given the loop-invariant a = dot(d,d) and the sphere's
// precomputed r2 = r*r.  Returns the shader's hit_sphere value: the accepted
// root, or -1.  `disc` must be >= 0 (the caller tests it).
template <bool COUNTED>
RT_DEV float sphere_roots(float h, float disc, float a, lane_counters& k) {
  RT_COUNT(k, roots, 1);
  float sq = sqrt_rn(disc);
  float t = (h - sq) / a;
  if (t < 0.001f) {
    RT_COUNT(k, roots, 1);
    t = (h + sq) / a;
    if (t < 0.001f) t = -1.0f;
  }
  return t;
}

// h and discriminant of fs:145-150.
RT_DEV void sphere_disc(float cx, float cy, float cz, float r2, f3 o, f3 d, float a, float& h, float& disc) {
  float ocx = cx - o.x, ocy = cy - o.y, ocz = cz - o.z;
  h = (d.x * ocx + d.y * ocy) + d.z * ocz;
  float c = ((ocx * ocx + ocy * ocy) + ocz * ocz) - r2;
  disc = h * h - a * c;
}

// ---- quads: the second kind of object ----------------------------------------
// No counterpart in the reference, whose hittable base (hittable.h:9-14) anticipates other objects but has only the
// sphere: a parallelogram Q + a*u + b*v, 0 <= a, b <= 1.  From the corner and the two edge vectors the upload
// derives, once, in binary32 and in this order: n = cross(u, v), nn = dot(n, n), normal = (1 / sqrt(nn)) * n,
// D = dot(normal, Q), w = (1 / nn) * n.
RT_DEV f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }  // as vec3.h:139-144

struct quad_dev {
  f3 Q, u, v, normal, w;
  float D;
};

// The ray parameter at which o + t*d meets the quad, or -1: the plane's parameter must exceed the shader's 0.001
// (fs:153), and the planar coordinates of the point computed at it lie in [0,1]^2.
RT_DEV float hit_quad(const quad_dev& q, f3 o, f3 d) {
  const float denom = dot3(q.normal, d);
  if (fabsf(denom) < 1e-8f) return -1.0f;
  const float t = (q.D - dot3(q.normal, o)) / denom;
  if (!(t > 0.001f)) return -1.0f;
  const f3 hp = (o + d * t) - q.Q;
  const float alpha = dot3(q.w, cross3(hp, q.v));
  const float beta = dot3(q.w, cross3(q.u, hp));
  if (!(alpha >= 0.0f && alpha <= 1.0f && beta >= 0.0f && beta <= 1.0f)) return -1.0f;
  return t;
}

// ---- 2-D gradient noise (perlin.h:11-85) -----------------------------------

RT_DEV void perlin_gradient(int ix, int iy, float& gx, float& gy) {  // perlin.h:31-43
  unsigned a = (unsigned)ix, b = (unsigned)iy;
  a *= 3284157443u;
  b ^= (a << 16) | (a >> 16);
  b *= 1911520717u;
  a ^= (b << 16) | (b >> 16);
  a *= 2048419325u;
  // The angle is a double product rounded to float, and cos / sin are taken
  // in double of that float, then narrowed (perlin.h:39-41).
  float random = (float)((double)a * (3.14159265 / 2147483648.0));
  double s, c;
  sincos((double)random, &s, &c);
  gx = (float)c;
  gy = (float)s;
}

// Gradients of the lattice points [-half, half)^2, computed once per scene by perlin_gradient itself
// (so a table entry is bit for bit what the function returns): a gradient costs a double-precision
// sincos, four per noise value.  Points outside the window are computed on the spot.
struct perlin_table {
  const float2* grad;  // (iy + half) * 2 * half + (ix + half); null: no table
  int half;
};

RT_DEV float perlin_dot(float gx, float gy, int ix, int iy, float x, float y) {  // perlin.h:46-56
  float dx = x - (float)ix;
  float dy = y - (float)iy;
  return dx * gx + dy * gy;
}

RT_DEV float perlin_lerp(float a0, float a1, float w) { return (a1 - a0) * w + a0; }  // perlin.h:11-16

// (int)floor(v) of perlin.h:61-63.  Beyond the int range (and for NaN) C++ leaves the conversion undefined; the x86
// hosts the reference runs on give 0x80000000 there, gfx950's own conversion saturates: the host's value is kept.
RT_DEV int perlin_cell(float v) {
  const float f = floorf(v);
  return fabsf(f) < 2147483648.0f ? (int)f : (int)0x80000000;
}

RT_DEV float perlin2(float x, float y, perlin_table tab) {  // perlin.h:59-85
  int x0 = perlin_cell(x);
  int x1 = (int)((unsigned)x0 + 1u);
  int y0 = perlin_cell(y);
  int y1 = (int)((unsigned)y0 + 1u);
  float sx = x - (float)x0;
  float sy = y - (float)y0;
  float2 g00, g10, g01, g11;
  if (tab.grad != nullptr && x0 >= -tab.half && x1 < tab.half && y0 >= -tab.half && y1 < tab.half) {
    const float2* row0 = tab.grad + (size_t)(y0 + tab.half) * (size_t)(2 * tab.half) + (size_t)(x0 + tab.half);
    const float2* row1 = row0 + 2 * tab.half;
    g00 = row0[0];
    g10 = row0[1];
    g01 = row1[0];
    g11 = row1[1];
  } else {
    perlin_gradient(x0, y0, g00.x, g00.y);
    perlin_gradient(x1, y0, g10.x, g10.y);
    perlin_gradient(x0, y1, g01.x, g01.y);
    perlin_gradient(x1, y1, g11.x, g11.y);
  }
  float n0 = perlin_dot(g00.x, g00.y, x0, y0, x, y);
  float n1 = perlin_dot(g10.x, g10.y, x1, y0, x, y);
  float ix0 = perlin_lerp(n0, n1, sx);
  n0 = perlin_dot(g01.x, g01.y, x0, y1, x, y);
  n1 = perlin_dot(g11.x, g11.y, x1, y1, x, y);
  float ix1 = perlin_lerp(n0, n1, sx);
  float value = perlin_lerp(ix0, ix1, sy);
  return (float)(((double)value * 0.5) + 0.5);
}

// ---- shading (testFragment.fs:225-324) --------------------------------------

RT_DEV float schlick(float cosine, float ri) {  // fs:310-315
  float r0 = (1.0f - ri) / (1.0f + ri);
  r0 = r0 * r0;
  return r0 + (1.0f - r0) * pow5(1.0f - cosine);
}

// fs:318-324 together with its call site fs:217: the caller multiplies the
// throughput by (throughput * sky), so the radiance is albedo^2 * sky.
RT_DEV f3 sky_radiance(f3 dir, f3 albedo) {
  float a = 0.5f * (1.0f + normalize3(dir).y);
  f3 sky = (1.0f - a) * mk3(1.0f, 1.0f, 1.0f) + a * mk3(0.5f, 0.7f, 1.0f);
  return albedo * (albedo * sky);
}

// Hit record of hit_any (fs:32-39) reduced to what shading needs.
struct hit_rec {
  f3 point, normal;
  bool interior;
  int mat;
};

// material_shade, fs:225-289.  Returns whether the path continues
// (ray.bounce); updates dir and albedo.  The lambertian, perlin-lambertian and
// metallic branches all start with one random_unit_vector: it is drawn once,
// ahead of the branch, so that lanes with different materials share the
// rejection loop (each lane's own draw sequence is unchanged).
// NOISE: whether the scene has a noise-textured material.  The render kernels exist in both forms, because the mere
// presence of the noise function — the double-precision sincos of lattice points outside the gradient table, 600
// instructions and a dozen scalar registers spilled into lanes — costs a kernel that never calls it 1.7 % of its
// rate (1920x1080x256 on 487 spheres: 9 950 -> 10 120 Mpaths/s) through the registers it takes from everything else.
template <int RNG, bool COUNTED, bool NOISE>
RT_DEV bool shade_hit(const hit_rec& h, int type, float param1, f3 m_albedo, perlin_table noise, f3& dir, f3& albedo,
                      uint32_t& rng, lane_counters& k) {
  const bool diffuse = type == RT_MAT_LAMBERTIAN || type == RT_MAT_PERLIN_LAMBERTIAN;
  const bool metal = type == RT_MAT_METALLIC;
  f3 ruv = mk3(0.0f, 0.0f, 0.0f);
  if (diffuse || metal) ruv = random_unit_vector<RNG, COUNTED>(rng, k);
  if (diffuse) {  // fs:243-252, random_on_hemisphere fs:385-392
    if (!(dot3(ruv, h.normal) >= 0.0f)) ruv = -ruv;
    f3 nd = h.normal + ruv;
    if (near_zero3(nd)) nd = h.normal;
    dir = nd;
    if (type == RT_MAT_LAMBERTIAN) {
      RT_COUNT(k, lambertian, 1);
      albedo = albedo * m_albedo;
    } else {
      RT_COUNT(k, perlin, 1);
      // (NOISE false: kernels for scenes without this material, which never get here)
      const float n = NOISE ? perlin2(param1 * h.point.x, param1 * h.point.z, noise) : 1.0f;
      albedo = albedo * (m_albedo * n);
    }
    return true;
  }
  if (metal) {  // fs:254-265
    RT_COUNT(k, metallic, 1);
    f3 r = reflect3(dir, normalize3(h.normal));
    r = normalize3(r) + (param1 * ruv);
    dir = r;
    albedo = albedo * m_albedo;
    bool go = dot3(r, h.normal) > 0.0f;
    if (!go) albedo = mk3(0.0f, 0.0f, 0.0f);
    return go;
  }
  if (type == RT_MAT_DIALECTRIC) {  // fs:267-289
    RT_COUNT(k, dialectric, 1);
    float ri = param1;
    if (!h.interior) ri = 1.0f / ri;
    f3 unit_dir = normalize3(dir);
    f3 unit_normal = normalize3(h.normal);
    float cos_theta = dot3(-unit_dir, h.normal);
    cos_theta = (1.0f < cos_theta) ? 1.0f : cos_theta;  // min(x, 1.0)
    float sin_theta = sqrt_rn(1.0f - cos_theta * cos_theta);
    bool cannot_refract = ri * sin_theta > 1.0f;
    bool refl = cannot_refract;
    if (!refl) refl = schlick(cos_theta, ri) > rng_float<RNG>(rng);  // no draw on total internal reflection
    dir = refl ? reflect3(unit_dir, unit_normal) : refract3(unit_dir, unit_normal, ri);
    return true;
  }
  albedo = mk3(1.0f, 0.0f, 0.0f);  // fs:236: unknown type paints red and stops
  return false;
}
