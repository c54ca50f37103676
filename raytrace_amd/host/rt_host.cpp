// rt_host.cpp — host-side harness: camera set-up and the benchmark scenes,
// written against the scene description API (rt_scene_api.hpp) exactly as an
// application would use it.  Plain C++, no GPU code.

#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/raytrace_amd_host.h"
#include "rt_scene_api.hpp"

namespace {

inline void st3(float* dst, const vec3& v) {
  dst[0] = v[0];
  dst[1] = v[1];
  dst[2] = v[2];
}
inline vec3 ld3(const float* p) { return vec3(p[0], p[1], p[2]); }

// The additive-feedback generator glibc's rand() runs by default (degree 31,
// separation 3, LCG-16807 warm-up, 310 discarded outputs), so that the
// random-sphere scene is the one an unseeded std::rand() produces on Linux
// without touching the process-wide rand() state.
class libc_rand {
  uint32_t r_[34];
  int k_ = 0;

 public:
  explicit libc_rand(uint32_t seed = 1) {
    int32_t word = (int32_t)seed;
    r_[0] = (uint32_t)word;
    for (int i = 1; i < 31; i++) {
      int64_t w = (16807LL * word) % 2147483647LL;
      if (w < 0) w += 2147483647LL;
      word = (int32_t)w;
      r_[i] = (uint32_t)word;
    }
    for (int i = 31; i < 34; i++) r_[i] = r_[i - 31];
    k_ = 0;
    for (int i = 34; i < 344; i++) next_raw();
  }
  uint32_t next_raw() {
    // position k_ holds x[n-34]; x[n] = x[n-31] + x[n-3]
    uint32_t v = r_[(k_ + 3) % 34] + r_[(k_ + 31) % 34];
    r_[k_] = v;
    k_ = (k_ + 1) % 34;
    return v;
  }
  int next() { return (int)(next_raw() >> 1); }
  // random_float(), vec3.h:18-21
  float unit() { return next() / (2147483647 + 1.0); }
  float range(float lo, float hi) { return lo + (hi - lo) * unit(); }
};

inline uint32_t pcg_hash(uint32_t x) {
  uint32_t state = x * 747796405u + 2891336453u;
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
// k-th uniform of element i of the stream `seed`, in [0,1) with 24 bits.
inline float hash01(uint32_t seed, uint64_t i, uint32_t k) {
  uint32_t h = pcg_hash(seed ^ pcg_hash((uint32_t)i));
  h = pcg_hash(h ^ pcg_hash((uint32_t)(i >> 32) * 0x9E3779B9u + k));
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}

struct counts {
  int spheres, materials;
  int object_bytes = 0;  // hittable_list::offset, for lists whose records differ in size
};

thread_local int g_object_bytes = 0;

// glGenBuffers + glBufferData of raytrace.cpp:381-386 / 416-421.
void allocate_buffers(size_t slots) {
  for (uint32_t id : {RT_HOST_MATERIAL_BUFFER, RT_HOST_SPHERE_BUFFER}) {
    rt_gl_bind_buffer(RT_GL_UNIFORM_BUFFER, id);
    rt_gl_buffer_data(RT_GL_UNIFORM_BUFFER, (intptr_t)(slots * RT_RECORD_BYTES), nullptr, 0x88E4 /* GL_STATIC_DRAW */);
    rt_gl_bind_buffer(RT_GL_UNIFORM_BUFFER, 0);
  }
}
constexpr size_t MAX_NUM_OBJECTS = 1024;  // raytrace.cpp:94

counts scene_default() {  // raytrace.cpp:396-448
  allocate_buffers(MAX_NUM_OBJECTS);
  material_list materials;
  lambertian mat_ground(vec3(0.8, 0.8, 0.0));
  lambertian mat_centre(vec3(0.1, 0.2, 0.5));
  dialectric mat_left(1.5);
  dialectric mat_left_bubble(1.0 / 1.5);
  metallic mat_right(vec3(0.8, 0.6, 0.2), 0.0);
  lambertian mat_left2(vec3(1.0, 0.0, 0.0));
  lambertian mat_right2(vec3(0.0, 0.0, 1.0));
  for (material* m : {(material*)&mat_ground, (material*)&mat_centre, (material*)&mat_left, (material*)&mat_right,
                      (material*)&mat_left_bubble, (material*)&mat_left2, (material*)&mat_right2})
    materials.add(RT_HOST_MATERIAL_BUFFER, *m);

  hittable_list objects;
  sphere ground(100.0, vec3(0.0, -100.5, -1.0), &mat_ground);
  sphere centre(0.5, vec3(0.0, 0.0, -1.2), &mat_centre);
  sphere centre_bubble(0.4, vec3(0.0, 0.0, -1.2), &mat_left_bubble);
  sphere left(0.5, vec3(-1.0, 0.0, -1.0), &mat_left);
  sphere left_bubble(0.4, vec3(-1.0, 0.0, -1.0), &mat_left_bubble);
  sphere right(0.5, vec3(1.0, 0.0, -1.0), &mat_right);
  for (sphere* s : {&ground, &centre, &centre_bubble, &left, &left_bubble, &right}) objects.add(RT_HOST_SPHERE_BUFFER, *s);
  return {objects.num, materials.num};
}

counts scene_random_spheres() {  // raytrace.cpp:455-504
  allocate_buffers(MAX_NUM_OBJECTS);
  libc_rand rng(1);
  material_list materials;
  hittable_list objects;
  // Materials must outlive nothing here: add() copies the bytes out at once.
  lambertian ground_material(colour(0.5, 0.5, 0.5));
  sphere ground(1000, point3(0, -1000, 0), &ground_material);
  materials.add(RT_HOST_MATERIAL_BUFFER, ground_material);
  objects.add(RT_HOST_SPHERE_BUFFER, ground);

  dialectric material1(1.5);
  lambertian material2(colour(0.4, 0.2, 0.1));
  metallic material3(colour(0.7, 0.6, 0.5), 0.0);
  materials.add(RT_HOST_MATERIAL_BUFFER, material1);
  materials.add(RT_HOST_MATERIAL_BUFFER, material2);
  materials.add(RT_HOST_MATERIAL_BUFFER, material3);
  sphere sphere1(1.0, point3(0, 1, 0), &material1);
  sphere sphere2(1.0, point3(-4, 1, 0), &material2);
  sphere sphere3(1.0, point3(4, 1, 0), &material3);
  objects.add(RT_HOST_SPHERE_BUFFER, sphere1);
  objects.add(RT_HOST_SPHERE_BUFFER, sphere2);
  objects.add(RT_HOST_SPHERE_BUFFER, sphere3);

  for (int a = -11; a < 11; a++) {
    for (int b = -11; b < 11; b++) {
      float choose_mat = rng.unit();
      // g++ evaluates the arguments of a parenthesised constructor call right
      // to left: the z jitter is drawn before the x jitter.
      float jz = rng.unit();
      float jx = rng.unit();
      point3 centre(a + 0.9 * jx, 0.2, b + 0.9 * jz);
      if ((centre - point3(4, 0.2, 0)).length() > 0.9) {
        if (choose_mat < 0.8) {
          // albedo = random() * random(): right operand first, and inside each
          // random() the components last to first.
          float q2 = rng.unit(), q1 = rng.unit(), q0 = rng.unit();
          float p2 = rng.unit(), p1 = rng.unit(), p0 = rng.unit();
          colour albedo = colour(p0, p1, p2) * colour(q0, q1, q2);
          lambertian m(albedo);
          materials.add(RT_HOST_MATERIAL_BUFFER, m);
          sphere s(0.2, centre, &m);
          objects.add(RT_HOST_SPHERE_BUFFER, s);
        } else if (choose_mat < 0.95) {
          float c2 = rng.range(0.5, 1), c1 = rng.range(0.5, 1), c0 = rng.range(0.5, 1);
          float fuzz = rng.range(0, 0.5);
          metallic m(colour(c0, c1, c2), fuzz);
          materials.add(RT_HOST_MATERIAL_BUFFER, m);
          sphere s(0.2, centre, &m);
          objects.add(RT_HOST_SPHERE_BUFFER, s);
        } else {
          dialectric m(1.5);
          materials.add(RT_HOST_MATERIAL_BUFFER, m);
          sphere s(0.2, centre, &m);
          objects.add(RT_HOST_SPHERE_BUFFER, s);
        }
      }
    }
  }
  return {objects.num, materials.num};
}

counts scene_perlin() {
  allocate_buffers(MAX_NUM_OBJECTS);
  material_list materials;
  hittable_list objects;
  perlin_lambertian noise(colour(0.9, 0.9, 0.9), 4.0f);
  materials.add(RT_HOST_MATERIAL_BUFFER, noise);
  sphere ground(1000, point3(0, -1000, 0), &noise);
  sphere ball(2, point3(0, 2, 0), &noise);
  objects.add(RT_HOST_SPHERE_BUFFER, ground);
  objects.add(RT_HOST_SPHERE_BUFFER, ball);
  return {objects.num, materials.num};
}

counts scene_cloud(int n, uint32_t seed) {
  allocate_buffers((size_t)n);
  material_list materials;
  hittable_list objects;
  lambertian grey(colour(0.5, 0.5, 0.5));
  materials.add(RT_HOST_MATERIAL_BUFFER, grey);
  for (int i = 0; i < n; i++) {
    float cx = -500.0f + 1000.0f * hash01(seed, (uint64_t)i, 0);
    float cy = -500.0f + 1000.0f * hash01(seed, (uint64_t)i, 1);
    float cz = -500.0f + 1000.0f * hash01(seed, (uint64_t)i, 2);
    float r = 0.1f + 0.9f * hash01(seed, (uint64_t)i, 3);
    sphere s(r, point3(cx, cy, cz), &grey);
    objects.add(RT_HOST_SPHERE_BUFFER, s);
  }
  return {objects.num, materials.num};
}

// A room of five quads (walls, floor, ceiling; open towards the camera) with three spheres in it.
counts scene_quad_room() {
  allocate_buffers(MAX_NUM_OBJECTS);
  material_list materials;
  hittable_list objects;
  lambertian red(colour(0.65, 0.05, 0.05)), white(colour(0.73, 0.73, 0.73)), green(colour(0.12, 0.45, 0.15));
  lambertian blue(colour(0.2, 0.2, 0.8));
  metallic mirror(colour(0.8, 0.85, 0.88), 0.0), brushed(colour(0.7, 0.6, 0.5), 0.3);
  dialectric glass(1.5);
  for (material* m : {(material*)&red, (material*)&white, (material*)&green, (material*)&blue, (material*)&mirror,
                      (material*)&brushed, (material*)&glass})
    materials.add(RT_HOST_MATERIAL_BUFFER, *m);
  quad left(point3(-3, -2, 1), vec3(0, 0, -6), vec3(0, 4, 0), &red);
  quad back(point3(-3, -2, -5), vec3(6, 0, 0), vec3(0, 4, 0), &mirror);
  quad right(point3(3, -2, -5), vec3(0, 0, 6), vec3(0, 4, 0), &green);
  quad ceiling(point3(-3, 2, -5), vec3(6, 0, 0), vec3(0, 0, 6), &white);
  quad floor_(point3(-3, -2, 1), vec3(6, 0, 0), vec3(0, 0, -6), &white);
  quad slanted(point3(-1.5, -2, -3.5), vec3(1.5, 0, 1.0), vec3(0.3, 2.5, -0.4), &blue);
  sphere ball(0.8, point3(1.2, -1.2, -2.5), &glass);
  sphere ball2(0.6, point3(-1.4, -1.4, -1.2), &brushed);
  sphere bubble(-0.7, point3(1.2, -1.2, -2.5), &glass);
  objects.add(RT_HOST_SPHERE_BUFFER, left);
  objects.add(RT_HOST_SPHERE_BUFFER, ball);
  objects.add(RT_HOST_SPHERE_BUFFER, back);
  objects.add(RT_HOST_SPHERE_BUFFER, right);
  objects.add(RT_HOST_SPHERE_BUFFER, ball2);
  objects.add(RT_HOST_SPHERE_BUFFER, ceiling);
  objects.add(RT_HOST_SPHERE_BUFFER, floor_);
  objects.add(RT_HOST_SPHERE_BUFFER, bubble);
  objects.add(RT_HOST_SPHERE_BUFFER, slanted);
  return {objects.num, materials.num, objects.offset};
}

// n objects uniform in [-50,50)^3: every third one a quad with edges of up to 4 units, the others spheres of
// radius [0.1,1); four materials.
counts scene_quad_cloud(int n, uint32_t seed) {
  allocate_buffers((size_t)n * 2);
  material_list materials;
  hittable_list objects;
  lambertian grey(colour(0.5, 0.5, 0.5)), warm(colour(0.8, 0.5, 0.3));
  metallic metal(colour(0.7, 0.7, 0.8), 0.1);
  dialectric glass(1.5);
  material* mats[4] = {&grey, &warm, &metal, &glass};
  for (material* m : mats) materials.add(RT_HOST_MATERIAL_BUFFER, *m);
  for (int i = 0; i < n; i++) {
    auto h = [&](uint32_t k) { return hash01(seed, (uint64_t)i, k); };
    const point3 c(-50.0f + 100.0f * h(0), -50.0f + 100.0f * h(1), -50.0f + 100.0f * h(2));
    material* m = mats[(int)(h(3) * 4.0f) & 3];
    if (i % 3 == 1) {
      quad q(c, vec3(-2.0f + 4.0f * h(4), -2.0f + 4.0f * h(5), -2.0f + 4.0f * h(6)),
             vec3(-2.0f + 4.0f * h(7), -2.0f + 4.0f * h(8), -2.0f + 4.0f * h(9)), m);
      objects.add(RT_HOST_SPHERE_BUFFER, q);
    } else {
      sphere s(0.1f + 0.9f * h(4), c, m);
      objects.add(RT_HOST_SPHERE_BUFFER, s);
    }
  }
  return {objects.num, materials.num, objects.offset};
}

}  // namespace

extern "C" {

int rt_host_camera(const rt_camera* cam, int width, int height, rt_params* out) {
  if (!cam || !out || width < 1 || height < 1) return RT_ERR_INVALID;
  // Types as in raytrace.cpp:341-366: theta, h, viewport_height are float,
  // viewport_width is double and narrows when it scales a vec3.
  const point3 lookfrom = ld3(cam->lookfrom), lookat = ld3(cam->lookat);
  const vec3 vup = ld3(cam->vup);
  point3 camera_origin = lookfrom;
  float theta = degrees_to_radians(cam->vfov);
  float h = std::tan(theta / 2);
  float viewport_height = 2 * h * cam->focus_dist;
  double viewport_width = (double(width) / double(height)) * viewport_height;
  vec3 w = unit_vector(lookfrom - lookat);
  vec3 u = unit_vector(cross(vup, w));
  vec3 v = cross(w, u);
  vec3 viewport_u = viewport_width * u;
  vec3 viewport_v = viewport_height * -v;
  vec3 delta_u = viewport_u / width;
  vec3 delta_v = viewport_v / height;
  vec3 viewport_top_left = camera_origin - (cam->focus_dist * w) - viewport_u / 2 - viewport_v / 2;
  float defocus_radius = cam->focus_dist * std::tan(degrees_to_radians(cam->defocus_angle / 2));
  vec3 defocus_disc_u = u * defocus_radius;
  vec3 defocus_disc_v = v * defocus_radius;

  out->width = width;
  out->height = height;
  out->tile_x0 = 0;
  out->tile_y0 = 0;
  out->tile_x1 = width;
  out->tile_y1 = height;
  st3(out->delta_u, delta_u);
  st3(out->delta_v, delta_v);
  st3(out->camera_origin, cam->origin_is_lookat ? lookat : lookfrom);  // raytrace.cpp:528
  st3(out->viewport_top_left, viewport_top_left);
  st3(out->defocus_disk_u, defocus_disc_u);
  st3(out->defocus_disk_v, defocus_disc_v);
  out->defocus_angle = cam->defocus_angle;
  return RT_OK;
}

int rt_host_scene_camera(int scene, rt_camera* out) {
  if (!out) return RT_ERR_INVALID;
  std::memset(out, 0, sizeof(*out));
  auto set = [&](vec3 from, vec3 at, vec3 up, float fov, float defocus, float focus, int is_lookat) {
    st3(out->lookfrom, from);
    st3(out->lookat, at);
    st3(out->vup, up);
    out->vfov = fov;
    out->defocus_angle = defocus;
    out->focus_dist = focus;
    out->origin_is_lookat = is_lookat;
  };
  switch (scene) {
    case RT_SCENE_DEFAULT:  // raytrace.cpp:334-339, origin quirk of :528 kept
      set(point3(0, 10, 0), point3(0, -1, 0), vec3(0, 0, 1), 90.0, 0.6, 1.0, 1);
      return RT_OK;
    case RT_SCENE_RANDOM_SPHERES:
      set(point3(13, 2, 3), point3(0, 0, 0), vec3(0, 1, 0), 20.0, 0.6, 10.0, 0);
      return RT_OK;
    case RT_SCENE_PERLIN:
      set(point3(13, 2, 3), point3(0, 2, 0), vec3(0, 1, 0), 20.0, 0.0, 10.0, 0);
      return RT_OK;
    case RT_SCENE_CLOUD:
      set(point3(0, 0, 900), point3(0, 0, 0), vec3(0, 1, 0), 40.0, 0.0, 10.0, 0);
      return RT_OK;
    case RT_SCENE_QUAD_ROOM:
      set(point3(0, 0, 6), point3(0, 0, -2), vec3(0, 1, 0), 50.0, 0.0, 8.0, 0);
      return RT_OK;
    case RT_SCENE_QUAD_CLOUD:
      set(point3(0, 0, 120), point3(0, 0, 0), vec3(0, 1, 0), 45.0, 0.0, 10.0, 0);
      return RT_OK;
  }
  return RT_ERR_INVALID;
}

int rt_host_build_scene(int scene, int n, uint32_t seed, int* n_materials) {
  rt_gl_reset();
  counts c{0, 0};
  switch (scene) {
    case RT_SCENE_DEFAULT: c = scene_default(); break;
    case RT_SCENE_RANDOM_SPHERES: c = scene_random_spheres(); break;
    case RT_SCENE_PERLIN: c = scene_perlin(); break;
    case RT_SCENE_CLOUD:
      if (n < 1) return RT_ERR_INVALID;
      c = scene_cloud(n, seed);
      break;
    case RT_SCENE_QUAD_ROOM: c = scene_quad_room(); break;
    case RT_SCENE_QUAD_CLOUD:
      if (n < 1) return RT_ERR_INVALID;
      c = scene_quad_cloud(n, seed);
      break;
    default: return RT_ERR_INVALID;
  }
  if (n_materials) *n_materials = c.materials;
  g_object_bytes = c.object_bytes ? c.object_bytes : c.spheres * RT_RECORD_BYTES;
  return c.spheres;
}

int rt_host_scene_object_bytes(void) { return g_object_bytes; }

int rt_host_gen_rays(uint32_t seed, uint64_t first, size_t n, float* o_xyz, float* d_xyz) {
  if (!o_xyz || !d_xyz) return RT_ERR_INVALID;
  unsigned hw = std::thread::hardware_concurrency();
  int threads = (int)(hw == 0 ? 1 : (hw > 16 ? 16 : hw));
  if (n < 65536) threads = 1;
  auto work = [&](int tid) {
    for (size_t j = (size_t)tid; j < n; j += (size_t)threads) {
      const uint64_t i = first + j;
      float* o = o_xyz + 3 * j;
      float* d = d_xyz + 3 * j;
      for (int k = 0; k < 3; k++) o[k] = -600.0f + 1200.0f * hash01(seed, i, (uint32_t)k);
      float z = 2.0f * hash01(seed, i, 3) - 1.0f;
      float phi = 6.2831853f * hash01(seed, i, 4);
      float len = 0.5f + 1.5f * hash01(seed, i, 5);
      float s = std::sqrt(std::fmax(0.0f, 1.0f - z * z));
      d[0] = len * s * std::cos(phi);
      d[1] = len * s * std::sin(phi);
      d[2] = len * z;
    }
  };
  if (threads == 1) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back(work, t);
    for (auto& t : pool) t.join();
  }
  return RT_OK;
}

int rt_host_blit_nearest(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
  if (!src || !dst || sw < 1 || sh < 1 || dw < 1 || dh < 1) return RT_ERR_INVALID;
  for (int jt = 0; jt < dh; jt++) {
    const int j = dh - 1 - jt;  // window row from the bottom
    const int r = (int)(((int64_t)(2 * j + 1) * sh) / (2 * (int64_t)dh));
    const uint8_t* row = src + (size_t)(sh - 1 - r) * sw * 3;
    uint8_t* out = dst + (size_t)jt * dw * 3;
    for (int i = 0; i < dw; i++) {
      const int c = (int)(((int64_t)(2 * i + 1) * sw) / (2 * (int64_t)dw));
      out[3 * i] = row[3 * c];
      out[3 * i + 1] = row[3 * c + 1];
      out[3 * i + 2] = row[3 * c + 2];
    }
  }
  return RT_OK;
}

int rt_host_height_for_width(int width) {
  const double aspect_ratio = 16.0 / 9.0;  // raytrace.cpp:83, a double there too
  const int h = (int)(width / aspect_ratio);
  return h < 1 ? 1 : h;
}

int rt_host_write_ppm(const char* path, const uint8_t* rgb8, int width, int height) {
  if (!path || !rgb8 || width < 1 || height < 1) return RT_ERR_INVALID;
  FILE* f = std::fopen(path, "wb");
  if (!f) return RT_ERR_INVALID;
  const size_t n = (size_t)width * height * 3;
  bool ok = std::fprintf(f, "P6\n%d %d\n255\n", width, height) > 0;
  ok = ok && std::fwrite(rgb8, 1, n, f) == n;
  ok = (std::fclose(f) == 0) && ok;
  return ok ? RT_OK : RT_ERR_INVALID;
}

}  // extern "C"
