// rt_scene_api.hpp — host-side scene description API.
//
// Mirrors the scene-building interface of the reference project so that code
// written against its headers compiles against this one: the same class names
// (vec3, hittable, hittable_list, sphere, material, lambertian, metallic,
// dialectric, material_list), the same constructor signatures and the same
// add(buffer, offset) contract — each add() writes the object's fields into a
// 32-byte std140 slot of a uniform buffer through a bind / sub-data / unbind
// triple per field.
//   reference interface        this file
//   vec3.h:33-149              vec3, point3, colour, dot, cross, unit_vector
//   hittable.h:6-15            hittable
//   hittable_list.h:6-21       hittable_list
//   sphere.h:11-36             sphere
//   (none; extension)          quad
//   material.h:9-65            MATERIAL_TYPE, material, lambertian, metallic, dialectric
//   material_list.h:6-22       material_list
// The writes go to rt_gl_bind_buffer / rt_gl_buffer_sub_data (raytrace_amd.h),
// the same two functions a build of the reference's own headers is pointed at
// through glad_glBindBuffer / glad_glBufferSubData.
//
// Everything is inline, so the header may be included from several
// translation units.
#ifndef RT_SCENE_API_HPP
#define RT_SCENE_API_HPP

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <ostream>

#include "../../include/raytrace_amd.h"

constexpr uint32_t RT_GL_UNIFORM_BUFFER = 0x8A11;  // GL_UNIFORM_BUFFER

// ---- vec3 -----------------------------------------------------------------

const float pi = 3.1415926535897932385;

// Arithmetic types below follow vec3.h:14-26 exactly (float in, double
// intermediate where a double literal takes part, float out): the camera set-up
// is checked bit for bit against values computed with the reference's header.
inline float degrees_to_radians(float degrees) { return degrees * pi / 180.0; }
inline float random_float() { return std::rand() / (RAND_MAX + 1.0); }
inline float random_float(float lo, float hi) { return lo + (hi - lo) * random_float(); }

struct vec2 {
  float x, y;
};

class vec3 {
 public:
  float e[3];

  vec3() : e{0.0f, 0.0f, 0.0f} {}
  vec3(float a, float b, float c) : e{a, b, c} {}

  float x() const { return e[0]; }
  float y() const { return e[1]; }
  float z() const { return e[2]; }
  float operator[](int i) const { return e[i]; }
  float& operator[](int i) { return e[i]; }

  vec3 operator-() const { return vec3(-e[0], -e[1], -e[2]); }
  vec3& operator+=(const vec3& o) {
    for (int i = 0; i < 3; i++) e[i] += o.e[i];
    return *this;
  }
  vec3& operator*=(float t) {
    for (int i = 0; i < 3; i++) e[i] *= t;
    return *this;
  }
  vec3& operator/=(float t) { return *this *= 1 / t; }

  float length_squared() const { return e[0] * e[0] + e[1] * e[1] + e[2] * e[2]; }
  float length() const { return std::sqrt((double)length_squared()); }
  // All three components within 1e-8 of zero.  (The reference's host-side
  // helper of this name, vec3.h:76-80, has misplaced parentheses and is not
  // used anywhere; the shader's near_zero, testFragment.fs:400-403, is this.)
  bool near_zero() const {
    const double s = 1e-8;
    return std::fabs(e[0]) < s && std::fabs(e[1]) < s && std::fabs(e[2]) < s;
  }

  // Components are drawn first to last.
  static vec3 random() {
    float a = random_float();
    float b = random_float();
    float c = random_float();
    return vec3(a, b, c);
  }
  static vec3 random(float lo, float hi) {
    float a = random_float(lo, hi);
    float b = random_float(lo, hi);
    float c = random_float(lo, hi);
    return vec3(a, b, c);
  }
};

using point3 = vec3;
using colour = vec3;

inline std::ostream& operator<<(std::ostream& o, const vec3& v) { return o << v.e[0] << ' ' << v.e[1] << ' ' << v.e[2]; }
inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline vec3 operator-(const vec3& a, const vec3& b) { return vec3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline vec3 operator*(const vec3& a, const vec3& b) { return vec3(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }
inline vec3 operator*(const vec3& a, float t) { return vec3(t * a[0], t * a[1], t * a[2]); }
inline vec3 operator*(float t, const vec3& a) { return a * t; }
inline vec3 operator/(const vec3& a, float t) { return (1 / t) * a; }
inline float dot(const vec3& a, const vec3& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline vec3 cross(const vec3& a, const vec3& b) {
  return vec3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
inline vec3 unit_vector(const vec3& v) { return v / v.length(); }

// ---- uniform-buffer writes ------------------------------------------------

namespace rt_scene_detail {
// One field of a record: bind the buffer, write, unbind.
inline void put(unsigned int buffer, int offset, size_t bytes, const void* src) {
  rt_gl_bind_buffer(RT_GL_UNIFORM_BUFFER, buffer);
  rt_gl_buffer_sub_data(RT_GL_UNIFORM_BUFFER, (intptr_t)offset, (intptr_t)bytes, src);
  rt_gl_bind_buffer(RT_GL_UNIFORM_BUFFER, 0);
}
}  // namespace rt_scene_detail

// ---- materials ------------------------------------------------------------

enum MATERIAL_TYPE { NOTHING, LAMBERTIAN, METALLIC, DIALECTRIC, PERLIN_LAMBERTIAN };

// Slot layout (std140, 32 bytes): id @0, type @4, param1 @8, albedo @16.
class material {
 public:
  int id = 0;  // assigned by material_list::add
  int type = NOTHING;
  vec3 albedo;
  float param1 = 0.0f;  // fuzz (metallic), refraction index (dialectric), noise scale (perlin)
  const int size = RT_RECORD_BYTES;

  material() {}
  material(int t, vec3 a) : type{t}, albedo{a} {}
  material(int t, float p) : type{t}, param1{p} {}
  material(int t, vec3 a, float p) : type{t}, albedo{a}, param1{p} {}

  void add(unsigned int glBuffer, int offset) {
    using rt_scene_detail::put;
    put(glBuffer, offset + 0, sizeof(int), &id);
    put(glBuffer, offset + 4, sizeof(int), &type);
    put(glBuffer, offset + 8, sizeof(float), &param1);
    put(glBuffer, offset + 16, 3 * sizeof(float), albedo.e);
  }
};

class lambertian : public material {
 public:
  explicit lambertian(vec3 a) : material(LAMBERTIAN, a) {}
};
class metallic : public material {
 public:
  metallic(vec3 a, float roughness) : material(METALLIC, a, roughness) {}
};
class dialectric : public material {
 public:
  explicit dialectric(float rel_refract_index) : material(DIALECTRIC, rel_refract_index) {}
};
// Extension: lambertian whose albedo is multiplied by perlin(k*x, k*z) of the
// hit point (perlin() as perlin.h:59-85), k = scale.
class perlin_lambertian : public material {
 public:
  perlin_lambertian(vec3 a, float scale) : material(PERLIN_LAMBERTIAN, a, scale) {}
};

class material_list {
 public:
  int num = 0;
  int offset = 0;
  // Ids follow insertion order, so a material must be added before the spheres
  // that point at it are added (sphere::add reads mat->id when called).
  void add(unsigned int ubo, material& m) {
    m.id = num;
    m.add(ubo, offset);
    offset += m.size;
    num += 1;
  }
};

// ---- hittables ------------------------------------------------------------

class hittable {
 public:
  int size;
  hittable() : size{0} {}
  explicit hittable(int bytes) : size{bytes} {}
  virtual ~hittable() = default;
  virtual void add(unsigned int glBuffer, int offset) const = 0;
};

class hittable_list {
 public:
  int num = 0;
  int offset = 0;
  void add(unsigned int ubo, hittable& h) {
    h.add(ubo, offset);
    offset += h.size;
    num += 1;
  }
};

// Slot layout (std140, 32 bytes): material id @0, radius @4, origin @16.
class sphere : public hittable {
 public:
  float radius = 0.0f;
  vec3 origin;
  material* mat = nullptr;

  sphere() {}
  sphere(float r, vec3 o, material* m) : hittable{RT_RECORD_BYTES}, radius{r}, origin{o}, mat{m} {}

  void add(unsigned int glBuffer, int offset) const override {
    using rt_scene_detail::put;
    put(glBuffer, offset + 0, sizeof(int), &mat->id);
    put(glBuffer, offset + 4, sizeof(float), &radius);
    put(glBuffer, offset + 16, 3 * sizeof(float), origin.e);
  }
};

// Extension: a parallelogram with corner Q and edges u and v, the second subclass of hittable.
// Record layout (64 bytes): material id @0, RT_OBJECT_QUAD @8, Q @16, u @32, v @48.  Lists that hold one
// are handed to rt_scene_upload_objects with hittable_list::offset as the byte count.
class quad : public hittable {
 public:
  vec3 Q, u, v;
  material* mat = nullptr;

  quad() {}
  quad(vec3 corner, vec3 edge_u, vec3 edge_v, material* m) : hittable{RT_QUAD_RECORD_BYTES}, Q{corner}, u{edge_u}, v{edge_v}, mat{m} {}

  void add(unsigned int glBuffer, int offset) const override {
    using rt_scene_detail::put;
    const int kind = RT_OBJECT_QUAD;
    put(glBuffer, offset + 0, sizeof(int), &mat->id);
    put(glBuffer, offset + 8, sizeof(int), &kind);
    put(glBuffer, offset + 16, 3 * sizeof(float), Q.e);
    put(glBuffer, offset + 32, 3 * sizeof(float), u.e);
    put(glBuffer, offset + 48, 3 * sizeof(float), v.e);
  }
};

#endif  // RT_SCENE_API_HPP
