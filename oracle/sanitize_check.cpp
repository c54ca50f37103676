// sanitize_check.cpp — the host-side C++ of this repository (scene API and harness, GL capture, CPU checker) linked
// into one executable with -fsanitize=address,undefined and driven through every entry point the CPU tests use:
// all scene builders, cameras, the ray generator, the presentation pass, the checker's renders, closest-hit batches,
// quads, the seed image and the noise function.  TEST INFRASTRUCTURE (make -C oracle asan): any report of the
// sanitizers aborts the run; the last line printed is "sanitize_check: ok".
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/raytrace_amd_host.h"

extern "C" {
int orc_render(const void*, int, const void*, int, const rt_params*, float*, rt_counters*, int);
int orc_render_objects(const void*, size_t, int, const void*, int, const rt_params*, float*, rt_counters*, int);
int orc_trace_batch(const void*, int, const float*, const float*, size_t, int32_t*, float*, float*, rt_counters*, int);
int orc_trace_batch_objects(const void*, size_t, int, const float*, const float*, size_t, int32_t*, float*, float*, rt_counters*, int);
void orc_resolve(const float*, size_t, int, float*, uint8_t*);
void orc_seed_texture(int, int, float, float, int, float, uint32_t*);
float orc_perlin(float, float);
void orc_perlin_gradient(int, int, float*);
}

static void need(bool ok, const char* what) {
  if (!ok) {
    std::fprintf(stderr, "sanitize_check: %s failed\n", what);
    std::exit(1);
  }
}

struct captured {
  std::vector<unsigned char> objects, materials;
  int n_objects = 0, n_materials = 0;
  bool tagged = false;
};

static captured build(int scene, int n, uint32_t seed) {
  captured c;
  c.n_objects = rt_host_build_scene(scene, n, seed, &c.n_materials);
  need(c.n_objects >= 0, "rt_host_build_scene");
  c.tagged = scene == RT_SCENE_QUAD_ROOM || scene == RT_SCENE_QUAD_CLOUD;
  size_t size = 0;
  const void* p = rt_gl_buffer_bytes(RT_HOST_SPHERE_BUFFER, &size);
  const size_t want = c.tagged ? (size_t)rt_host_scene_object_bytes() : (size_t)c.n_objects * RT_RECORD_BYTES;
  need(p != nullptr && size >= want, "captured object buffer");
  c.objects.assign((const unsigned char*)p, (const unsigned char*)p + want);
  p = rt_gl_buffer_bytes(RT_HOST_MATERIAL_BUFFER, &size);
  need(p != nullptr && size >= (size_t)c.n_materials * RT_RECORD_BYTES, "captured material buffer");
  c.materials.assign((const unsigned char*)p, (const unsigned char*)p + (size_t)c.n_materials * RT_RECORD_BYTES);
  rt_gl_reset();
  return c;
}

int main() {
  const int scenes[] = {RT_SCENE_DEFAULT, RT_SCENE_RANDOM_SPHERES, RT_SCENE_PERLIN, RT_SCENE_CLOUD, RT_SCENE_QUAD_ROOM, RT_SCENE_QUAD_CLOUD};
  for (int scene : scenes) {
    const captured sc = build(scene, scene == RT_SCENE_CLOUD ? 3000 : 600, 7u);
    rt_camera cam;
    need(rt_host_scene_camera(scene, &cam) == 0, "rt_host_scene_camera");
    for (int rng = 0; rng < 2; rng++) {
      rt_params p;
      const int w = 37, h = 21, spp = 3;   // ragged against every tile and strip size
      need(rt_host_camera(&cam, w, h, &p) == 0, "rt_host_camera");
      p.sample_begin = 0;
      p.sample_end = spp;
      p.bounce_limit = 50;
      p.rng_mode = rng;
      p.seed = 0x5EED0001u;
      p.seed_tex = nullptr;
      std::vector<uint32_t> seed_tex((size_t)w * h);
      if (rng == RT_RNG_XORSHIFT_REF) {
        orc_seed_texture(w, h, 400.0f, 1.5f, 12, 1.5f, seed_tex.data());
        p.seed_tex = seed_tex.data();
      }
      for (int strips = 0; strips < 4; strips += 3) {
        std::vector<float> accum((size_t)w * h * 3, 0.0f);
        rt_counters k;
        for (int s = 0; s < (strips ? strips : 1); s++) {
          p.strip_count = (uint16_t)strips;
          p.strip_index = (uint16_t)s;
          const int rc = sc.tagged ? orc_render_objects(sc.objects.data(), sc.objects.size(), sc.n_objects, sc.materials.data(),
                                                        sc.n_materials, &p, accum.data(), &k, 3)
                                   : orc_render(sc.objects.data(), sc.n_objects, sc.materials.data(), sc.n_materials, &p,
                                                accum.data(), &k, 3);
          need(rc == 0, "orc_render");
        }
        std::vector<float> lin(accum.size());
        std::vector<uint8_t> rgb8(accum.size());
        orc_resolve(accum.data(), (size_t)w * h, spp, lin.data(), rgb8.data());
        const int dw = 100, dh = rt_host_height_for_width(dw);
        std::vector<uint8_t> window((size_t)dw * dh * 3);
        need(rt_host_blit_nearest(rgb8.data(), w, h, window.data(), dw, dh) == 0, "rt_host_blit_nearest");
        const std::string path = "/tmp/sanitize_check.ppm";
        need(rt_host_write_ppm(path.c_str(), window.data(), dw, dh) == 0, "rt_host_write_ppm");
        std::remove(path.c_str());
      }
    }
    const size_t n_rays = 2000;
    std::vector<float> o(3 * n_rays), d(3 * n_rays), t(n_rays), nrm(3 * n_rays);
    std::vector<int32_t> id(n_rays);
    need(rt_host_gen_rays(99u, 5u, n_rays, o.data(), d.data()) == 0, "rt_host_gen_rays");
    rt_counters k;
    const int rc = sc.tagged ? orc_trace_batch_objects(sc.objects.data(), sc.objects.size(), sc.n_objects, o.data(), d.data(), n_rays,
                                                       id.data(), t.data(), nrm.data(), &k, 4)
                             : orc_trace_batch(sc.objects.data(), sc.n_objects, o.data(), d.data(), n_rays, id.data(), t.data(),
                                               nrm.data(), &k, 4);
    need(rc == 0, "orc_trace_batch");
  }
  // empty scene, zero rays, degenerate sizes
  {
    rt_camera cam;
    need(rt_host_scene_camera(RT_SCENE_DEFAULT, &cam) == 0, "camera");
    rt_params p;
    need(rt_host_camera(&cam, 5, 3, &p) == 0, "rt_host_camera");
    p.sample_begin = 0; p.sample_end = 2; p.bounce_limit = 0; p.rng_mode = RT_RNG_COUNTER; p.seed = 1; p.seed_tex = nullptr;
    std::vector<float> accum(5 * 3 * 3, 0.0f);
    rt_counters k;
    need(orc_render(nullptr, 0, nullptr, 0, &p, accum.data(), &k, 1) == 0, "empty scene");
    need(orc_trace_batch(nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, nullptr, &k, 1) == 0, "zero rays");
    need(rt_host_camera(&cam, 0, 3, &p) != 0, "zero-width image refused");
  }
  // the noise function far outside any table, and beyond int range
  float g[2];
  for (float x : {0.5f, -2049.25f, 3.0e9f, -3.0e9f, 1.0e30f}) (void)orc_perlin(x, -x);
  for (int i : {0, -1, 2147483647, (int)0x80000000}) orc_perlin_gradient(i, -i - 1, g);
  std::puts("sanitize_check: ok");
  return 0;
}
