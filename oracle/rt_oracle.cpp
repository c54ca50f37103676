// rt_oracle.cpp — CPU restatement of the reference fragment shader.
//
// TEST INFRASTRUCTURE ONLY.  This file is the checker that the HIP kernels are
// compared against; nothing in the shipped library links, loads or calls it.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
//
// What it restates: /root/reference/shaders/testFragment.fs (all of it), the
// 2-D noise of /root/reference/perlin.h, and the std140 record layouts written
// by /root/reference/sphere.h:22-35 and /root/reference/material.h:29-46.
// Each statement carries the .fs line it follows ("fs:NNN").
//
// Parity pinning: the reference has no tests and no golden vectors, and its
// arithmetic is GLSL that cannot be executed without a GL device, so image
// parity is UNPINNED by the reference.  What pins this file: the hand-derivable
// known answers of the shader's scalar functions (tests/test_oracle_kat.py),
// the perlin() value table and the camera vectors produced by compiling the
// reference's own perlin.h / vec3.h (oracle/ref_harness.cpp -> oracle/_ref/).
//
// Floating point: every operation is a single IEEE-754 binary32 operation in
// the order written; compile with -ffp-contract=off -fno-fast-math.  GLSL
// built-ins are restated as follows (GLSL 3.30 leaves their rounding open):
//   dot(a,b)      (a.x*b.x + a.y*b.y) + a.z*b.z            as vec3.h:132-137
//   v / s         (1/s) * v                                 as vec3.h:127-130
//   normalize(v)  v / sqrt(dot(v,v))                        as vec3.h:146-149
//   reflect(I,N)  I - (2*dot(N,I))*N                        GLSL 3.30 s8.5
//   refract       k = 1 - eta*eta*(1 - dot(N,I)*dot(N,I));  GLSL 3.30 s8.5
//                 k<0 ? 0 : eta*I - (eta*dot(N,I) + sqrt(k))*N
//   pow(x,5)      x2 = x*x; x4 = x2*x2; x4*x
//   min(a,b)      b < a ? b : a
//
// Build: see oracle/Makefile.

#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/raytrace_amd.h"

#if defined(__FAST_MATH__)
#error "the oracle must not be built with fast-math"
#endif

namespace {

struct V3 {
  float x, y, z;
};

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return a * s; }
inline V3 operator/(V3 a, float s) { return (1.0f / s) * a; }  // vec3.h:127-130
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }  // vec3.h:132-137
inline V3 normalize(V3 v) { return v / std::sqrt(dot(v, v)); }               // vec3.h:146-149
inline V3 reflect(V3 I, V3 N) { return I - (2.0f * dot(N, I)) * N; }
inline V3 refract(V3 I, V3 N, float eta) {
  float ni = dot(N, I);
  float k = 1.0f - eta * eta * (1.0f - ni * ni);
  if (k < 0.0f) return v3(0.0f, 0.0f, 0.0f);
  return eta * I - (eta * ni + std::sqrt(k)) * N;
}
inline float pow5(float x) {
  float x2 = x * x;
  float x4 = x2 * x2;
  return x4 * x;
}
inline float fmin2(float a, float b) { return b < a ? b : a; }

// ---- scene records --------------------------------------------------------

struct Sphere {  // fs:58-63, bytes per sphere.h:22-35
  int32_t mat;
  float radius;
  V3 origin;
};
struct Material {  // fs:50-56, bytes per material.h:29-46
  int32_t id;
  int32_t type;
  float param1;
  V3 albedo;
};

Sphere decode_sphere(const uint8_t* p) {
  Sphere s;
  std::memcpy(&s.mat, p + 0, 4);
  std::memcpy(&s.radius, p + 4, 4);
  std::memcpy(&s.origin, p + 16, 12);
  return s;
}
Material decode_material(const uint8_t* p) {
  Material m;
  std::memcpy(&m.id, p + 0, 4);
  std::memcpy(&m.type, p + 4, 4);
  std::memcpy(&m.param1, p + 8, 4);
  std::memcpy(&m.albedo, p + 16, 12);
  return m;
}

// A parallelogram (no counterpart in the reference, whose hittable has one subclass, hittable.h:6-15, sphere.h:11):
// the record rt_scene_upload_objects documents, and what is derived from it once per scene.
struct Quad {
  int32_t mat;
  V3 Q, u, v;
  V3 normal, w;  // unit normal; n / dot(n, n), n = cross(u, v)
  float D;       // dot(normal, Q)
};

struct Scene {
  std::vector<Sphere> spheres;  // one per object; a quad's entry holds its material only
  std::vector<int32_t> quad_of; // object -> index into quads, or -1 (empty: no quads)
  std::vector<Quad> quads;      // in upload order
  std::vector<int32_t> quad_object;
  std::vector<Material> materials;
};

// ---- random numbers -------------------------------------------------------

inline uint32_t mix32(uint32_t x) {  // counter-mode finaliser (not in the reference)
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

struct Rng {  // fs:28-30 xorshift32_state, plus the injected counter mode
  uint32_t a;
  int mode;
};

inline uint32_t next_u32(Rng& s) {  // fs:326-332; the counter mode draws from the same generator
  uint32_t x = s.a;
  x ^= x << 13;
  x ^= x >> 17;
  x ^= x << 5;
  return s.a = x;
}

inline float rand_float(Rng& s) {  // fs:334-345
  uint32_t x = next_u32(s);
  x &= 0x007FFFFFu;
  x |= 0x3F800000u;
  float f;
  std::memcpy(&f, &x, 4);
  return f - 1.0f;
}

inline uint32_t cantor(uint32_t k1, uint32_t k2) {  // fs:394-398
  uint32_t x = (k1 + k2) * (k1 + k2 + 1u);
  x = x >> 1;
  return x + k2;
}

// Counter mode: the xorshift32 state a camera sample starts from is a hash of
// (seed, pixel, sample) alone, so a sample's stream does not depend on which
// other samples were drawn before it.  Zero is xorshift's fixed point: avoided.
inline uint32_t counter_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
  uint32_t h = mix32(seed ^ mix32(pixel));
  h = mix32(h + sample * 0x9E3779B9u);
  return h ? h : 0x9E3779B9u;
}

struct Counters {
  rt_counters c;
  Counters() { std::memset(&c, 0, sizeof(c)); }
};

inline V3 rand_vec(Rng& s) {  // fs:347-349, constructor arguments left to right
  float x = rand_float(s);
  float y = rand_float(s);
  float z = rand_float(s);
  return v3(x, y, z);
}

V3 random_unit_vector(Rng& s, Counters& k) {  // fs:351-366
  uint32_t i = 0u;
  while (true) {
    s.a += i;                    // fs:357
    V3 p = rand_vec(s);          // fs:358
    float lensq = dot(p, p);     // fs:359
    k.c.unit_tries++;
    if (0.0f < lensq && lensq <= 1.0f) {  // fs:360 (1e-160 is 0 in binary32)
      return p / std::sqrt(lensq);        // fs:362
    }
    i += 1u;                     // fs:364
  }
}

V3 random_unit_disk(Rng& s, Counters& k) {  // fs:368-383
  uint32_t i = 0u;
  while (true) {
    s.a += i;                                 // fs:374
    float x = rand_float(s);
    float y = rand_float(s);
    V3 p = v3(x, y, 0.0f);                    // fs:375
    float lensq = dot(p, p);                  // fs:376
    k.c.disk_tries++;
    if (lensq <= 1.0f) return p;              // fs:377-380
    i += 1u;
  }
}

V3 random_on_hemisphere(Rng& s, V3 normal, Counters& k) {  // fs:385-392
  V3 u = random_unit_vector(s, k);
  if (dot(u, normal) >= 0.0f) return u;
  return -u;
}

inline bool near_zero(V3 v) {  // fs:400-403
  float s = 1e-8f;
  return std::fabs(v.x) < s && std::fabs(v.y) < s && std::fabs(v.z) < s;
}

// ---- intersection ---------------------------------------------------------

float hit_sphere(V3 origin, float radius, V3 ray_dir, V3 ray_orig, Counters& k) {  // fs:143-163
  k.c.sphere_tests++;
  V3 oc = origin - ray_orig;                 // fs:145
  float a = dot(ray_dir, ray_dir);           // fs:146
  float h = dot(ray_dir, oc);                // fs:147
  float c = dot(oc, oc) - radius * radius;   // fs:148
  float discriminant = h * h - a * c;        // fs:150
  if (discriminant >= 0.0f) {                // fs:151
    k.c.roots++;
    float t = (h - std::sqrt(discriminant)) / a;  // fs:152
    if (t < 0.001f) {                             // fs:153
      k.c.roots++;
      t = (h + std::sqrt(discriminant)) / a;      // fs:154
      if (t < 0.001f) t = -1.0f;                  // fs:155-157
    }
    return t;
  }
  return -1.0f;  // fs:161
}

inline V3 cross(V3 a, V3 b) {  // vec3.h:139-144
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

Quad decode_quad(const uint8_t* p) {
  Quad q;
  std::memcpy(&q.mat, p + 0, 4);
  std::memcpy(&q.Q, p + 16, 12);
  std::memcpy(&q.u, p + 32, 12);
  std::memcpy(&q.v, p + 48, 12);
  const V3 n = cross(q.u, q.v);
  const float nn = dot(n, n);
  q.normal = (1.0f / std::sqrt(nn)) * n;
  q.D = dot(q.normal, q.Q);
  q.w = (1.0f / nn) * n;
  return q;
}

// Ray parameter of the hit, or -1.  The plane is met at t = (D - normal.o) / (normal.d) unless the ray runs along
// it; t must pass the shader's 0.001 gate (fs:153); the point computed at t, relative to Q, has planar coordinates
// alpha = w.(p x v), beta = w.(u x p), and both must lie in [0, 1].
float hit_quad(const Quad& q, V3 ray_dir, V3 ray_orig) {
  const float denom = dot(q.normal, ray_dir);
  if (std::fabs(denom) < 1e-8f) return -1.0f;
  const float t = (q.D - dot(q.normal, ray_orig)) / denom;
  if (!(t > 0.001f)) return -1.0f;
  const V3 hp = (ray_orig + ray_dir * t) - q.Q;
  const float alpha = dot(q.w, cross(hp, q.v));
  const float beta = dot(q.w, cross(q.u, hp));
  if (!(alpha >= 0.0f && alpha <= 1.0f && beta >= 0.0f && beta <= 1.0f)) return -1.0f;
  return t;
}

struct Hit {  // fs:32-39 plus the index and t, which the batch interface reports
  V3 point, normal;
  bool hit, interior;
  int mat;
  int index;
  float t;
};

Hit hit_any(const Scene& sc, V3 ray_orig, V3 ray_dir, Counters& k) {  // fs:165-199
  Hit h;
  h.point = v3(0, 0, 0);
  h.normal = v3(0, 0, 0);
  h.hit = false;
  h.interior = false;
  h.mat = 0;
  h.index = -1;
  float t = -1.0f;
  int si = -1;
  k.c.segments++;
  const int n = (int)sc.spheres.size();
  const bool mixed = !sc.quad_of.empty();
  for (int i = 0; i < n; i++) {  // fs:176
    const Sphere& sp = sc.spheres[i];
    float new_t = mixed && sc.quad_of[i] >= 0 ? -1.0f : hit_sphere(sp.origin, sp.radius, ray_dir, ray_orig, k);  // fs:178
    if (t < 0.0f || (new_t < t && new_t > 0.001f)) {                       // fs:180
      t = new_t;
      si = i;
    }
  }
  // Quads come after the spheres: one replaces the result when it is nearer, or as near and uploaded earlier.
  for (size_t qi = 0; qi < sc.quads.size(); qi++) {
    const float new_t = hit_quad(sc.quads[qi], ray_dir, ray_orig);
    const int idx = sc.quad_object[qi];
    if (new_t > 0.001f && (!(t > 0.001f) || new_t < t || (new_t == t && idx < si))) {
      t = new_t;
      si = idx;
    }
  }
  h.t = -1.0f;
  if (t > 0.001f) {  // fs:186
    const Sphere& s = sc.spheres[si];
    h.point = ray_orig + ray_dir * t;                 // fs:188
    h.normal = (h.point - s.origin) / s.radius;       // fs:189
    if (mixed && sc.quad_of[si] >= 0) h.normal = sc.quads[sc.quad_of[si]].normal;
    if (dot(h.normal, ray_dir) >= 0.0f) {             // fs:190
      h.normal = -h.normal;
      h.interior = true;
    }
    h.mat = s.mat;   // fs:194
    h.hit = true;    // fs:195
    h.index = si;
    h.t = t;
    k.c.hits++;
  }
  return h;
}

// ---- perlin.h -------------------------------------------------------------

inline float p_interpolate(float a0, float a1, float w) { return (a1 - a0) * w + a0; }  // perlin.h:11-16

inline void p_gradient(int ix, int iy, float& gx, float& gy) {  // perlin.h:31-43
  const unsigned w = 32, s = 16;
  unsigned a = (unsigned)ix, b = (unsigned)iy;
  a *= 3284157443u;
  b ^= a << s | a >> (w - s);
  b *= 1911520717u;
  a ^= b << s | b >> (w - s);
  a *= 2048419325u;
  float random = (float)((double)a * (3.14159265 / 2147483648.0));  // perlin.h:39
  gx = (float)std::cos((double)random);                              // perlin.h:41
  gy = (float)std::sin((double)random);
}

inline float p_dot_grid(int ix, int iy, float x, float y) {  // perlin.h:46-56
  float gx, gy;
  p_gradient(ix, iy, gx, gy);
  float dx = x - (float)ix;
  float dy = y - (float)iy;
  return dx * gx + dy * gy;
}

float perlin2(float x, float y) {  // perlin.h:59-85
  int x0 = (int)std::floor(x);
  int x1 = x0 + 1;
  int y0 = (int)std::floor(y);
  int y1 = y0 + 1;
  float sx = x - (float)x0;
  float sy = y - (float)y0;
  float n0 = p_dot_grid(x0, y0, x, y);
  float n1 = p_dot_grid(x1, y0, x, y);
  float ix0 = p_interpolate(n0, n1, sx);
  n0 = p_dot_grid(x0, y1, x, y);
  n1 = p_dot_grid(x1, y1, x, y);
  float ix1 = p_interpolate(n0, n1, sx);
  float value = p_interpolate(ix0, ix1, sy);
  return (float)(((double)value * 0.5) + 0.5);  // perlin.h:84, double arithmetic
}

// ---- shading --------------------------------------------------------------

struct Ray {  // fs:41-48
  V3 origin, dir;
  bool bounce;
  uint32_t count;
  V3 albedo;
};

float shlick(float cosine, float ri) {  // fs:310-315
  float r0 = (1.0f - ri) / (1.0f + ri);
  r0 = r0 * r0;
  return r0 + (1.0f - r0) * pow5(1.0f - cosine);
}

V3 shade_sky(V3 dir, V3 albedo) {  // fs:318-324
  float a = 0.5f * (1.0f + normalize(dir).y);
  albedo = albedo * ((1.0f - a) * v3(1.0f, 1.0f, 1.0f) + a * v3(0.5f, 0.7f, 1.0f));
  return albedo;
}

void lambertian(const Material& m, Hit& h, Ray& r, Rng& s, Counters& k) {  // fs:243-252
  k.c.lambertian++;
  r.dir = h.normal + random_on_hemisphere(s, h.normal, k);
  if (near_zero(r.dir)) r.dir = h.normal;
  r.bounce = true;
  r.albedo = r.albedo * m.albedo;
}

// Extension (type RT_MAT_PERLIN_LAMBERTIAN): lambertian whose albedo is scaled
// by perlin(k*p.x, k*p.z) at the hit point, k = param1.
void perlin_lambertian(const Material& m, Hit& h, Ray& r, Rng& s, Counters& k) {
  k.c.perlin++;
  r.dir = h.normal + random_on_hemisphere(s, h.normal, k);
  if (near_zero(r.dir)) r.dir = h.normal;
  r.bounce = true;
  float n = perlin2(m.param1 * h.point.x, m.param1 * h.point.z);
  r.albedo = r.albedo * (m.albedo * n);
}

void metallic(const Material& m, Hit& h, Ray& r, Rng& s, Counters& k) {  // fs:254-265
  k.c.metallic++;
  r.dir = reflect(r.dir, normalize(h.normal));                          // fs:256
  r.dir = normalize(r.dir) + (m.param1 * random_unit_vector(s, k));     // fs:257
  r.albedo = r.albedo * m.albedo;                                       // fs:258
  r.bounce = dot(r.dir, h.normal) > 0.0f;                               // fs:260
  if (!r.bounce) r.albedo = v3(0, 0, 0);
}

void dialectric(const Material& m, Hit& h, Ray& r, Rng& s, Counters& k) {  // fs:267-289
  k.c.dialectric++;
  float ri = m.param1;
  if (!h.interior) ri = 1.0f / ri;  // fs:270-272
  V3 unit_dir = normalize(r.dir);
  V3 unit_normal = normalize(h.normal);
  float cos_theta = fmin2(dot(-unit_dir, h.normal), 1.0f);      // fs:277
  float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);    // fs:278
  bool cannot_refract = ri * sin_theta > 1.0f;                  // fs:279
  if (cannot_refract || shlick(cos_theta, ri) > rand_float(s)) {  // fs:281, || short-circuits
    r.dir = reflect(unit_dir, unit_normal);
  } else {
    r.dir = refract(unit_dir, unit_normal, ri);
  }
  r.bounce = true;
}

void material_shade(const Scene& sc, Hit& h, Ray& r, Rng& s, Counters& k) {  // fs:225-241
  const Material& m = sc.materials[h.mat];
  if (m.type == 1) {
    lambertian(m, h, r, s, k);
  } else if (m.type == 2) {
    metallic(m, h, r, s, k);
  } else if (m.type == 3) {
    dialectric(m, h, r, s, k);
  } else if (m.type == RT_MAT_PERLIN_LAMBERTIAN) {
    perlin_lambertian(m, h, r, s, k);
  } else {
    r.albedo = v3(1.0f, 0.0f, 0.0f);  // fs:236
    r.bounce = false;
  }
}

Ray bounce(const Scene& sc, Ray r, uint32_t bounce_limit, Rng& s, Counters& k) {  // fs:201-223
  if (r.count >= bounce_limit) {
    r.albedo = v3(0, 0, 0);
    r.bounce = false;
    k.c.limit_hits++;
  } else {
    Hit h = hit_any(sc, r.origin, r.dir, k);
    if (h.hit) {
      r.origin = h.point;
      r.count = r.count + 1u;
      material_shade(sc, h, r, s, k);
    } else {
      r.albedo = r.albedo * shade_sky(r.dir, r.albedo);  // fs:217: albedo squared times sky
      r.bounce = false;
      k.c.sky++;
    }
  }
  return r;
}

V3 raycast(const Scene& sc, V3 o, V3 d, uint32_t bounce_limit, Rng& s, Counters& k) {  // fs:292-308
  Ray r;
  r.count = 0u;
  r.origin = o;
  r.dir = d;
  r.albedo = v3(1, 1, 1);
  r.bounce = true;
  k.c.paths++;
  while (true) {
    r = bounce(sc, r, bounce_limit, s, k);
    if (!r.bounce) break;
  }
  return r.albedo;
}

inline V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }

// One pixel of main(), fs:103-133: returns the SUM over the sample range.
// `colour` enters holding the pixel's sum over the samples rendered before
// (zero for a fresh frame, fs:115) and leaves with this range added in sample
// order, so that a frame rendered in several sample ranges, in order, is the
// frame rendered in one.
V3 pixel_sum(const Scene& sc, const rt_params& p, int x, int y, V3 colour, Counters& k) {
  Rng s;
  s.mode = p.rng_mode;
  const uint32_t pixel = (uint32_t)y * (uint32_t)p.width + (uint32_t)x;
  if (p.rng_mode == RT_RNG_XORSHIFT_REF) {
    uint32_t seed = p.seed_tex ? p.seed_tex[pixel] : 0u;  // fs:105-106
    seed ^= cantor((uint32_t)x, (uint32_t)y);             // fs:107
    s.a = seed;
  } else {
    s.a = 0;
  }
  const V3 delta_u = ld3(p.delta_u), delta_v = ld3(p.delta_v);
  const V3 cam = ld3(p.camera_origin), vtl = ld3(p.viewport_top_left);
  const V3 ddu = ld3(p.defocus_disk_u), ddv = ld3(p.defocus_disk_v);
  for (int i = p.sample_begin; i < p.sample_end; i++) {  // fs:117
    if (p.rng_mode == RT_RNG_COUNTER) s.a = counter_key(p.seed, pixel, (uint32_t)i);
    float jx = rand_float(s) - 0.5f;  // fs:119
    float jy = rand_float(s) - 0.5f;
    V3 frag_loc = vtl + ((float)x + jx) * delta_u + ((float)y + jy) * delta_v;  // fs:120-121
    V3 ray_origin;
    if (p.defocus_angle <= 0.0f) {
      ray_origin = cam;
    } else {
      V3 rd = random_unit_disk(s, k);
      ray_origin = cam + rd.x * ddu + rd.y * ddv;  // fs:129
    }
    colour = colour + raycast(sc, ray_origin, frag_loc - ray_origin, p.bounce_limit, s, k);  // fs:132
  }
  return colour;
}

void add_counters(rt_counters& dst, const rt_counters& src) {
  const uint64_t* s = reinterpret_cast<const uint64_t*>(&src);
  uint64_t* d = reinterpret_cast<uint64_t*>(&dst);
  for (size_t i = 0; i < sizeof(rt_counters) / sizeof(uint64_t); i++) d[i] += s[i];
}

Scene make_scene(const void* spheres, int ns, const void* mats, int nm) {
  Scene sc;
  sc.spheres.resize(ns);
  sc.materials.resize(nm);
  for (int i = 0; i < ns; i++) sc.spheres[i] = decode_sphere((const uint8_t*)spheres + 32 * (size_t)i);
  for (int i = 0; i < nm; i++) sc.materials[i] = decode_material((const uint8_t*)mats + 32 * (size_t)i);
  return sc;
}

// Records with their kind at byte 8 (0: sphere, 32 bytes; 1: quad, 64 bytes), n_bytes in all.  False if malformed.
bool make_scene_objects(Scene& sc, const void* objects, size_t n_bytes, int n, const void* mats, int nm) {
  sc.spheres.resize(n);
  sc.quad_of.assign(n, -1);
  sc.materials.resize(nm);
  const uint8_t* p = (const uint8_t*)objects;
  size_t off = 0;
  for (int i = 0; i < n; i++) {
    if (off + 32 > n_bytes) return false;
    int32_t kind;
    std::memcpy(&kind, p + off + 8, 4);
    if (kind == 1) {
      if (off + 64 > n_bytes) return false;
      const Quad q = decode_quad(p + off);
      sc.spheres[i] = Sphere{q.mat, 0.0f, v3(0, 0, 0)};
      sc.quad_of[i] = (int32_t)sc.quads.size();
      sc.quads.push_back(q);
      sc.quad_object.push_back(i);
      off += 64;
    } else if (kind == 0) {
      sc.spheres[i] = decode_sphere(p + off);
      off += 32;
    } else {
      return false;
    }
  }
  for (int i = 0; i < nm; i++) sc.materials[i] = decode_material((const uint8_t*)mats + 32 * (size_t)i);
  return true;
}

int render_scene(const Scene& sc, int n_materials, const rt_params* p, float* accum, rt_counters* counters, int threads);
int trace_scene(const Scene& sc, const float* o_xyz, const float* d_xyz, size_t n, int32_t* id, float* t, float* nrm,
                rt_counters* counters, int threads);

}  // namespace

extern "C" {

// The two entry points below on a list of spheres and quads (the records of rt_scene_upload_objects).
int orc_render_objects(const void* objects, size_t n_bytes, int n_objects, const void* materials_std140, int n_materials,
                       const rt_params* p, float* accum, rt_counters* counters, int threads) {
  if (!p || !accum || n_objects < 0 || n_materials < 0) return -1;
  Scene sc;
  if (!make_scene_objects(sc, objects, n_bytes, n_objects, materials_std140, n_materials)) return -1;
  return render_scene(sc, n_materials, p, accum, counters, threads);
}

int orc_trace_batch_objects(const void* objects, size_t n_bytes, int n_objects, const float* o_xyz, const float* d_xyz,
                            size_t n, int32_t* id, float* t, float* nrm, rt_counters* counters, int threads) {
  static const uint8_t no_materials[32] = {0};
  Scene sc;
  if (!make_scene_objects(sc, objects, n_bytes, n_objects, no_materials, 0)) return -1;
  return trace_scene(sc, o_xyz, d_xyz, n, id, t, nrm, counters, threads);
}

// Sum of radiance over samples [sample_begin, sample_end) for the pixels of the
// tile, added sample by sample onto accum (width*height*3 floats, row 0 on top).  `threads` rows
// are rendered concurrently (row y goes to thread y % threads).
int orc_render(const void* spheres_std140, int n_spheres, const void* materials_std140,
               int n_materials, const rt_params* p, float* accum, rt_counters* counters,
               int threads) {
  if (!p || !accum || n_spheres < 0 || n_materials < 0) return -1;
  Scene sc = make_scene(spheres_std140, n_spheres, materials_std140, n_materials);
  return render_scene(sc, n_materials, p, accum, counters, threads);
}

}  // extern "C"

namespace {

int render_scene(const Scene& sc, int n_materials, const rt_params* p, float* accum, rt_counters* counters, int threads) {
  for (const Sphere& s : sc.spheres)
    if (s.mat < 0 || s.mat >= n_materials) return -1;
  if (threads < 1) threads = 1;
  std::vector<Counters> ks((size_t)threads);
  auto work = [&](int tid) {
    for (int y = p->tile_y0 + tid; y < p->tile_y1; y += threads) {
      // interleaved shares (rt_params.strip_count / strip_index): strips of eight rows from the top of the tile
      if (p->strip_count != 0 && ((y - p->tile_y0) / 8) % p->strip_count != p->strip_index) continue;
      for (int x = p->tile_x0; x < p->tile_x1; x++) {
        float* dst = accum + 3 * ((size_t)y * (size_t)p->width + (size_t)x);
        const V3 c = pixel_sum(sc, *p, x, y, v3(dst[0], dst[1], dst[2]), ks[(size_t)tid]);
        dst[0] = c.x;
        dst[1] = c.y;
        dst[2] = c.z;
      }
    }
  };
  if (threads == 1) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back(work, t);
    for (auto& t : pool) t.join();
  }
  if (counters) {
    std::memset(counters, 0, sizeof(*counters));
    for (auto& k : ks) add_counters(*counters, k.c);
  }
  return 0;
}

}  // namespace

extern "C" {

// End of main(), fs:135-140, and the RGB8 store (raytrace.cpp:574): mean,
// gamma 1/2.2, clamp to [0,1], round to nearest.
void orc_resolve(const float* accum, size_t n_pixels, int samples, float* rgb_linear, uint8_t* rgb8) {
  for (size_t i = 0; i < 3 * n_pixels; i++) {
    float c = accum[i] * (1.0f / (float)samples);  // fs:135 (vec / scalar)
    if (rgb_linear) rgb_linear[i] = c;
    if (rgb8) {
      float g = std::pow(c, 1.0f / 2.2f);  // fs:138-140
      if (!(g > 0.0f)) g = 0.0f;
      if (g > 1.0f) g = 1.0f;
      rgb8[i] = (uint8_t)std::floor(g * 255.0f + 0.5f);
    }
  }
}

// hit_any on a batch: id (-1 on a miss), t (-1 on a miss), normal (0 on a miss).
int orc_trace_batch(const void* spheres_std140, int n_spheres, const float* o_xyz,
                    const float* d_xyz, size_t n, int32_t* id, float* t, float* nrm,
                    rt_counters* counters, int threads) {
  static const uint8_t no_materials[32] = {0};
  Scene sc = make_scene(spheres_std140, n_spheres, no_materials, 0);
  return trace_scene(sc, o_xyz, d_xyz, n, id, t, nrm, counters, threads);
}

}  // extern "C"

namespace {

int trace_scene(const Scene& sc, const float* o_xyz, const float* d_xyz, size_t n, int32_t* id, float* t, float* nrm,
                rt_counters* counters, int threads) {
  if (threads < 1) threads = 1;
  std::vector<Counters> ks((size_t)threads);
  auto work = [&](int tid) {
    for (size_t i = (size_t)tid; i < n; i += (size_t)threads) {
      Hit h = hit_any(sc, ld3(o_xyz + 3 * i), ld3(d_xyz + 3 * i), ks[(size_t)tid]);
      id[i] = h.index;
      t[i] = h.t;
      nrm[3 * i + 0] = h.normal.x;
      nrm[3 * i + 1] = h.normal.y;
      nrm[3 * i + 2] = h.normal.z;
    }
  };
  if (threads == 1) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    for (int th = 0; th < threads; th++) pool.emplace_back(work, th);
    for (auto& th : pool) th.join();
  }
  if (counters) {
    std::memset(counters, 0, sizeof(*counters));
    for (auto& k : ks) add_counters(*counters, k.c);
  }
  return 0;
}

}  // namespace

extern "C" {

// ---- scalar entry points for the known-answer tests ------------------------

float orc_hit_quad(const void* record64, const float* d, const float* o) {
  return hit_quad(decode_quad((const uint8_t*)record64), v3(d[0], d[1], d[2]), v3(o[0], o[1], o[2]));
}

uint32_t orc_xorshift32(uint32_t* state) {
  Rng s{*state, RT_RNG_XORSHIFT_REF};
  uint32_t r = next_u32(s);
  *state = s.a;
  return r;
}
float orc_rand_float(uint32_t* state, int mode) {
  Rng s{*state, mode};
  float r = rand_float(s);
  *state = s.a;
  return r;
}
uint32_t orc_cantor(uint32_t k1, uint32_t k2) { return cantor(k1, k2); }
uint32_t orc_counter_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
  return counter_key(seed, pixel, sample);
}
float orc_hit_sphere(const float* c, float radius, const float* d, const float* o) {
  Counters k;
  return hit_sphere(ld3(c), radius, ld3(d), ld3(o), k);
}
float orc_shlick(float cosine, float ri) { return shlick(cosine, ri); }
void orc_shade_sky(const float* dir, const float* albedo, float* out) {
  V3 r = shade_sky(ld3(dir), ld3(albedo));
  out[0] = r.x;
  out[1] = r.y;
  out[2] = r.z;
}
void orc_random_unit_vector(uint32_t* state, int mode, float* out, int* tries) {
  Rng s{*state, mode};
  Counters k;
  V3 r = random_unit_vector(s, k);
  *state = s.a;
  out[0] = r.x;
  out[1] = r.y;
  out[2] = r.z;
  if (tries) *tries = (int)k.c.unit_tries;
}
void orc_random_unit_disk(uint32_t* state, int mode, float* out, int* tries) {
  Rng s{*state, mode};
  Counters k;
  V3 r = random_unit_disk(s, k);
  *state = s.a;
  out[0] = r.x;
  out[1] = r.y;
  out[2] = r.z;
  if (tries) *tries = (int)k.c.disk_tries;
}
void orc_reflect(const float* i, const float* n, float* out) {
  V3 r = reflect(ld3(i), ld3(n));
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_refract(const float* i, const float* n, float eta, float* out) {
  V3 r = refract(ld3(i), ld3(n), eta);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
// ---------------------------------------------------------------------------
// The seed image: shaders/perlin.fs drawn into an RGB8 texture
// (raytrace.cpp:309-328, 566-579) and read back by the ray shader (fs:105-106).
// GLSL leaves sin() to the driver; this restatement takes it as the double
// sine rounded to binary32.  fract(x) = x - floor(x).
namespace seedtex {
float interpolate(float a0, float a1, float w) {  // perlin.fs:44-46
  return (a1 - a0) * (3.0f - w * 2.0f) * w * w + a0;
}
float glsl_sin(float x) { return (float)std::sin((double)x); }
float fract(float x) { return x - std::floor(x); }
void random_gradient(int ix, int iy, float& gx, float& gy) {  // perlin.fs:48-53
  const float ux = (float)ix, uy = (float)iy;                 // :49
  const float a = ux * 127.1f + uy * 311.7f;                  // :50
  const float b = ux * 269.5f + uy * 183.3f;                  // :51
  gx = -1.0f + 2.0f * fract(glsl_sin(a) * 43758.5453123f);    // :52
  gy = -1.0f + 2.0f * fract(glsl_sin(b) * 43758.5453123f);
}
float dot_grid_gradient(int ix, int iy, float x, float y) {  // perlin.fs:55-62
  float gx, gy;
  random_gradient(ix, iy, gx, gy);
  const float dx = x - (float)ix;
  const float dy = y - (float)iy;
  return dx * gx + dy * gy;
}
float perlin(float x, float y) {  // perlin.fs:64-87
  const int x0 = (int)std::floor(x), x1 = x0 + 1;
  const int y0 = (int)std::floor(y), y1 = y0 + 1;
  const float sx = x - (float)x0, sy = y - (float)y0;
  float n0 = dot_grid_gradient(x0, y0, x, y);
  float n1 = dot_grid_gradient(x1, y0, x, y);
  const float ix0 = interpolate(n0, n1, sx);
  n0 = dot_grid_gradient(x0, y1, x, y);
  n1 = dot_grid_gradient(x1, y1, x, y);
  const float ix1 = interpolate(n0, n1, sx);
  return interpolate(ix0, ix1, sy);
}
// perlin.fs:15-42 for the fragment whose window coordinates are (fx, fy).
float fragment_value(float fx, float fy, float grid, float contrast, int layers, float lacunarity) {
  float value = 0.0f, freq = 1.0f, amp = 1.0f;
  for (int i = 0; i < layers; i++) {
    value += perlin(fx * freq / grid, fy * freq / grid) * amp;  // :22
    freq *= lacunarity;                                          // :24
    amp /= 2.0f;                                                 // :25
  }
  value = value * contrast;          // :28
  if (value > 1.0f) value = 1.0f;    // :30-32
  if (value < -1.0f) value = -1.0f;  // :33-35
  return value * 0.5f + 0.5f;        // :37
}
}  // namespace seedtex

// seed_tex[y * width + x] for the ray shader's pixel (x, y), origin upper left
// (fs:3): the bits of tex.r + tex.g + tex.b (fs:106) of texel (x, height-1-y) of
// the RGB8 image perlin.fs draws with the window origin lower left.  The stored
// channel is round(255 v), the sampled one c / 255.
void orc_seed_texture(int width, int height, float grid, float contrast, int layers, float lacunarity, uint32_t* out) {
  for (int y = 0; y < height; y++) {
    const int ty = height - 1 - y;
    for (int x = 0; x < width; x++) {
      const float v = seedtex::fragment_value((float)x + 0.5f, (float)ty + 0.5f, grid, contrast, layers, lacunarity);
      const float c = std::floor(v * 255.0f + 0.5f) / 255.0f;
      const float sum = (c + c) + c;
      uint32_t bits;
      std::memcpy(&bits, &sum, 4);
      out[(size_t)y * width + x] = bits;
    }
  }
}

float orc_perlin(float x, float y) { return perlin2(x, y); }
void orc_perlin_gradient(int ix, int iy, float* g) { p_gradient(ix, iy, g[0], g[1]); }

}  // extern "C"
