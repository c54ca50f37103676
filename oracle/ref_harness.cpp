// ref_harness.cpp — runs the parts of the reference that compile in this
// container WITHOUT any stand-in header, and prints what they compute as JSON.
//
// TEST INFRASTRUCTURE ONLY (fixture generator).  It includes, unmodified and
// in place under /root/reference: vec3.h, perlin.h, hittable.h,
// hittable_list.h.  It does NOT include sphere.h, material.h or
// material_list.h: those include <SDL2/SDL.h>, which this image lacks, and a
// reference build must not be made by writing stand-ins for missing headers.
// Their 32-byte record layouts are therefore pinned from the source text
// (sphere.h:22-35, material.h:29-46) in tests/test_scene_records.py instead.
//
// Output (stdout, one JSON object): every float as its IEEE-754 bit pattern.
//   perlin        [[x_bits, y_bits, perlin(x,y)_bits], ...]      perlin.h:59-85
//   gradients     [[ix, iy, gx_bits, gy_bits], ...]              perlin.h:31-43
//   perlin_far    the same as perlin, for points whose lattice cell lies far from the origin
//                 or on either side of +-2048 (where a consumer may switch between a
//                 precomputed gradient table and gradients computed on the spot)
//   gradients_far the same as gradients, for such lattice points
//   camera_c1/c2  the uniform vectors of raytrace.cpp:341-366 computed with the
//                 reference's own vec3 operators
//   list          hittable_list bookkeeping after three add() calls
//   random_scene  the commented-out scene of raytrace.cpp:455-504, generated
//                 with the reference's random_float()/vec3::random() and the
//                 unseeded std::rand() it uses; records in std140 layout
//
// Build and run: make -C oracle ref   (outputs only under oracle/_ref/)

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "vec3.h"           // /root/reference/vec3.h
#include "perlin.h"         // /root/reference/perlin.h
#include "hittable.h"       // /root/reference/hittable.h
#include "hittable_list.h"  // /root/reference/hittable_list.h

static uint32_t bits(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}

static void print_vec(const char* name, const vec3& v, bool comma = true) {
  std::printf("    \"%s\": [%u, %u, %u]%s\n", name, bits(v[0]), bits(v[1]), bits(v[2]), comma ? "," : "");
}

// The camera block of raytrace.cpp:332-366, with the same declared types
// (auto deduces float for theta/h/viewport_height, double for viewport_width).
static void camera(const char* name, point3 lookfrom, point3 lookat, vec3 vup, float vfov,
                   float defocus_angle, float focus_dist, int W, int H, bool comma) {
  vec3 u, v, w;
  point3 camera_origin = lookfrom;
  auto theta = degrees_to_radians(vfov);
  auto h = std::tan(theta / 2);
  auto viewport_height = 2 * h * focus_dist;
  auto viewport_width = (double(W) / double(H)) * viewport_height;
  w = unit_vector(lookfrom - lookat);
  u = unit_vector(cross(vup, w));
  v = cross(w, u);
  vec3 viewport_u = viewport_width * u;
  vec3 viewport_v = viewport_height * -v;
  vec3 delta_u = viewport_u / W;
  vec3 delta_v = viewport_v / H;
  vec3 viewport_top_left = camera_origin - (focus_dist * w) - viewport_u / 2 - viewport_v / 2;
  float defocus_radius = focus_dist * std::tan(degrees_to_radians(defocus_angle / 2));
  vec3 defocus_disc_u = u * defocus_radius;
  vec3 defocus_disc_v = v * defocus_radius;
  std::printf("  \"%s\": {\n", name);
  std::printf("    \"width\": %d, \"height\": %d,\n", W, H);
  print_vec("lookfrom", lookfrom);
  print_vec("lookat", lookat);
  print_vec("delta_u", delta_u);
  print_vec("delta_v", delta_v);
  print_vec("viewport_top_left", viewport_top_left);
  print_vec("defocus_disk_u", defocus_disc_u);
  print_vec("defocus_disk_v", defocus_disc_v);
  std::printf("    \"defocus_angle\": %u\n", bits(defocus_angle));
  std::printf("  }%s\n", comma ? "," : "");
}

// A hittable that records where the list told it to go.
struct probe : public hittable {
  mutable std::vector<int> offsets;
  probe() : hittable{32} {}
  void add(unsigned int, int offset) const override { offsets.push_back(offset); }
};

struct record_sink {
  std::vector<uint8_t> spheres, materials;
  int n_materials = 0;
  int add_material(int type, vec3 albedo, float param1) {
    uint8_t r[32] = {0};
    int id = n_materials++;
    std::memcpy(r + 0, &id, 4);
    std::memcpy(r + 4, &type, 4);
    std::memcpy(r + 8, &param1, 4);
    std::memcpy(r + 16, &albedo, 12);
    materials.insert(materials.end(), r, r + 32);
    return id;
  }
  void add_sphere(float radius, vec3 origin, int mat) {
    uint8_t r[32] = {0};
    std::memcpy(r + 0, &mat, 4);
    std::memcpy(r + 4, &radius, 4);
    std::memcpy(r + 16, &origin, 12);
    spheres.insert(spheres.end(), r, r + 32);
  }
};

static void print_bytes(const char* name, const std::vector<uint8_t>& b, bool comma) {
  std::printf("    \"%s\": \"", name);
  for (uint8_t c : b) std::printf("%02x", c);
  std::printf("\"%s\n", comma ? "," : "");
}

int main() {
  std::printf("{\n");

  // perlin(x,y) over a lattice of inputs that crosses cell borders, negative
  // coordinates and large magnitudes.
  std::printf("  \"perlin\": [\n");
  bool first = true;
  for (int i = -40; i <= 40; i++) {
    for (int j = -6; j <= 6; j++) {
      float x = 0.37f * (float)i + 0.013f * (float)j;
      float y = 1.91f * (float)j - 0.29f * (float)i;
      if ((i + j) % 7 == 0) { x *= 97.0f; y *= 113.0f; }
      float p = perlin(x, y);
      std::printf("%s    [%u, %u, %u]", first ? "" : ",\n", bits(x), bits(y), bits(p));
      first = false;
    }
  }
  std::printf(",\n    [%u, %u, %u]", bits(1.3f), bits(2.7f), bits(perlin(1.3f, 2.7f)));
  std::printf(",\n    [%u, %u, %u]\n  ],\n", bits(0.5f), bits(0.5f), bits(perlin(0.5f, 0.5f)));

  std::printf("  \"gradients\": [\n");
  first = true;
  for (int ix = -9; ix <= 9; ix += 3) {
    for (int iy = -8; iy <= 8; iy += 4) {
      vector2 g = randomGradient(ix * 131, iy * 77 + 1);
      std::printf("%s    [%d, %d, %u, %u]", first ? "" : ",\n", ix * 131, iy * 77 + 1, bits(g.x), bits(g.y));
      first = false;
    }
  }
  std::printf("\n  ],\n");

  // Far from the origin, and around +-2048 in either coordinate.
  std::printf("  \"perlin_far\": [\n");
  first = true;
  {
    const float edge[] = {-2049.5f, -2048.5f, -2047.5f, -2046.25f, 2045.75f, 2046.5f, 2047.5f, 2048.5f, 2049.25f};
    const float mid[] = {-3.25f, 0.5f, 7.75f, -2048.0f, 2047.0f, 2048.0f};
    for (float a : edge) {
      for (float b : mid) {
        std::printf("%s    [%u, %u, %u]", first ? "" : ",\n", bits(a), bits(b), bits(perlin(a, b)));
        first = false;
        std::printf(",\n    [%u, %u, %u]", bits(b), bits(a), bits(perlin(b, a)));
      }
      for (float b : edge) std::printf(",\n    [%u, %u, %u]", bits(a), bits(b), bits(perlin(a, b)));
    }
    for (int i = 1; i <= 60; i++) {
      const float x = 2048.0f + 37.37f * (float)(i * i), y = -2048.0f - 1234.5f * (float)i + 0.125f;
      std::printf(",\n    [%u, %u, %u]", bits(x), bits(y), bits(perlin(x, y)));
      std::printf(",\n    [%u, %u, %u]", bits(-x), bits(0.3f * (float)i), bits(perlin(-x, 0.3f * (float)i)));
      std::printf(",\n    [%u, %u, %u]", bits(0.7f * (float)i), bits(-y), bits(perlin(0.7f * (float)i, -y)));
    }
    const float big[] = {1.0e6f, -1.0e6f, 16777215.0f, -16777215.0f, 1.0e9f, -1.0e9f};  // beyond 2^24 the fraction is zero
    for (float a : big) std::printf(",\n    [%u, %u, %u]", bits(a + 0.5f), bits(-a), bits(perlin(a + 0.5f, -a)));
  }
  std::printf("\n  ],\n");
  std::printf("  \"gradients_far\": [\n");
  first = true;
  {
    const int pts[] = {-2147483647 - 1, -2147483647, -1000000007, -65536, -2049, -2048, -2047, -1, 0, 1, 2046, 2047, 2048, 2049, 65535, 123456789, 2147483646, 2147483647};
    for (int ix : pts) {
      for (int iy : {-2049, -2048, 0, 2047, 2048, 99991, -2147483647 - 1, 2147483647}) {
        vector2 g = randomGradient(ix, iy);
        std::printf("%s    [%d, %d, %u, %u]", first ? "" : ",\n", ix, iy, bits(g.x), bits(g.y));
        first = false;
      }
    }
  }
  std::printf("\n  ],\n");

  // Default camera (raytrace.cpp:334-339) at the default render size
  // (raytrace.cpp:87,127) and the camera used for the random-sphere scene.
  camera("camera_c1", point3(0, 10, 0), point3(0, -1, 0), vec3(0, 0, 1), 90.0, 0.6, 1.0, 320, 180, true);
  camera("camera_c2", point3(13, 2, 3), point3(0, 0, 0), vec3(0, 1, 0), 20.0, 0.6, 10.0, 1920, 1080, true);
  camera("camera_c2_small", point3(13, 2, 3), point3(0, 0, 0), vec3(0, 1, 0), 20.0, 0.6, 10.0, 160, 90, true);

  // hittable_list bookkeeping (hittable_list.h:15-20).
  {
    hittable_list l;
    probe a, b, c;
    l.add(7, a);
    l.add(7, b);
    l.add(7, c);
    std::printf("  \"list\": {\"num\": %d, \"offset\": %d, \"offsets\": [%d, %d, %d]},\n", l.num, l.offset,
                a.offsets[0], b.offsets[0], c.offsets[0]);
  }

  // Random-sphere scene, raytrace.cpp:455-504 (same order of materials.add /
  // objects.add, same expressions where the order of random_float() calls
  // inside one expression is left to the compiler).
  {
    record_sink s;
    int m = s.add_material(1, colour(0.5, 0.5, 0.5), 0.0f);
    s.add_sphere(1000, point3(0, -1000, 0), m);
    int m1 = s.add_material(3, vec3(), 1.5);
    int m2 = s.add_material(1, colour(0.4, 0.2, 0.1), 0.0f);
    int m3 = s.add_material(2, colour(0.7, 0.6, 0.5), 0.0);
    s.add_sphere(1.0, point3(0, 1, 0), m1);
    s.add_sphere(1.0, point3(-4, 1, 0), m2);
    s.add_sphere(1.0, point3(4, 1, 0), m3);
    int n_l = 0, n_m = 0, n_d = 0;
    for (int a = -11; a < 11; a++) {
      for (int b = -11; b < 11; b++) {
        auto choose_mat = random_float();
        point3 centre(a + 0.9 * random_float(), 0.2, b + 0.9 * random_float());
        if ((centre - point3(4, 0.2, 0)).length() > 0.9) {
          if (choose_mat < 0.8) {
            auto albedo = colour::random() * colour::random();
            s.add_sphere(0.2, centre, s.add_material(1, albedo, 0.0f));
            n_l++;
          } else if (choose_mat < 0.95) {
            auto albedo = colour::random(0.5, 1);
            auto fuzz = random_float(0, 0.5);
            s.add_sphere(0.2, centre, s.add_material(2, albedo, fuzz));
            n_m++;
          } else {
            s.add_sphere(0.2, centre, s.add_material(3, vec3(), 1.5));
            n_d++;
          }
        }
      }
    }
    std::printf("  \"random_scene\": {\n");
    std::printf("    \"n_spheres\": %d, \"n_materials\": %d,\n", (int)(s.spheres.size() / 32), s.n_materials);
    std::printf("    \"small_lambertian\": %d, \"small_metallic\": %d, \"small_dialectric\": %d,\n", n_l, n_m, n_d);
    print_bytes("spheres", s.spheres, true);
    print_bytes("materials", s.materials, false);
    std::printf("  }\n");
  }
  std::printf("}\n");
  return 0;
}
