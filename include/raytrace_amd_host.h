/*
 * raytrace_amd_host.h — C entry points of the host-side harness (plain g++
 * code, no GPU): camera set-up and the benchmark scenes, both expressed
 * through the scene description API of raytrace_amd/host/rt_scene_api.hpp.
 * They exist so that tests and benchmarks written in another language can
 * drive the same host code a C++ application would.
 */
#ifndef RAYTRACE_AMD_HOST_H
#define RAYTRACE_AMD_HOST_H

#include "raytrace_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Camera description, struct Camera of raytrace.cpp:23-39 (inputs only). */
typedef struct rt_camera {
  float lookfrom[3];
  float lookat[3];
  float vup[3];
  float vfov;          /* vertical field of view, degrees */
  float defocus_angle; /* degrees; <= 0 disables the lens  */
  float focus_dist;
  /* The reference passes cam.lookat as the uniform camera_origin
   * (raytrace.cpp:528) although the viewport is built around lookfrom
   * (raytrace.cpp:360).  1 = do the same; 0 = use lookfrom. */
  int32_t origin_is_lookat;
} rt_camera;

/*
 * Replaces: the camera block raytrace.cpp:341-366 and the uniform wiring
 * raytrace.cpp:522-533.  Fills width/height, the full-image tile and every
 * camera vector of *out; leaves sample range, bounce limit and RNG fields
 * untouched.
 */
RT_API int rt_host_camera(const rt_camera* cam, int width, int height, rt_params* out);

/* Buffer ids the scene builders write to (what glGenBuffers would have made,
 * raytrace.cpp:381-384, 416-419). */
#define RT_HOST_MATERIAL_BUFFER 1u
#define RT_HOST_SPHERE_BUFFER 2u

enum {
  RT_SCENE_DEFAULT = 0,        /* the live scene, raytrace.cpp:396-448: 6 spheres, 7 materials     */
  RT_SCENE_RANDOM_SPHERES = 1, /* the commented-out scene, raytrace.cpp:455-504: 487 spheres       */
  RT_SCENE_PERLIN = 2,         /* ground + one sphere, perlin-modulated lambertian, scale 4         */
  RT_SCENE_CLOUD = 3,          /* n spheres uniform in [-500,500)^3, radius [0.1,1), one material   */
  RT_SCENE_QUAD_ROOM = 4,      /* six quads and three spheres, seven materials (mixed record sizes)  */
  RT_SCENE_QUAD_CLOUD = 5      /* n objects uniform in [-50,50)^3, every third a quad, four materials */
};

/*
 * Builds a scene with the scene API into the capture buffers
 * RT_HOST_MATERIAL_BUFFER / RT_HOST_SPHERE_BUFFER of the calling thread
 * (rt_gl_reset first; each buffer is allocated with rt_gl_buffer_data as at
 * raytrace.cpp:384,419).  n and seed are used by the two cloud scenes only.
 * Returns the number of objects (hittable_list::num), or a negative
 * rt_status; *n_materials (nullable) receives material_list::num.
 * Read the bytes with rt_gl_buffer_bytes.
 */
RT_API int rt_host_build_scene(int scene, int n, uint32_t seed, int* n_materials);

/* Bytes the objects of the calling thread's last rt_host_build_scene take (hittable_list::offset,
 * hittable_list.h:13-18): 32 per sphere, 64 per quad.  Scenes with quads go to rt_scene_upload_objects. */
RT_API int rt_host_scene_object_bytes(void);

/* Default camera of each scene (rt_host_build_scene's ids). */
RT_API int rt_host_scene_camera(int scene, rt_camera* out);

/*
 * Synthetic hit-test batch: rays [first, first+n) of the stream keyed by
 * seed.  Origin uniform in [-600,600)^3, direction uniform on the unit sphere
 * scaled by a length uniform in [0.5,2).  3 floats per ray.
 */
RT_API int rt_host_gen_rays(uint32_t seed, uint64_t first, size_t n, float* o_xyz, float* d_xyz);

/*
 * The reference's presentation pass (raytrace.cpp:545-551, texFragment.fs:8-11):
 * the RENDER_WIDTH x RENDER_HEIGHT RGB8 texture drawn over the whole
 * SCREEN_WIDTH x SCREEN_HEIGHT window with GL_NEAREST (raytrace.cpp:576-577).
 * Images here are rows from the top, as rt_readback returns them.  Window
 * pixel (i, j) shows source texel (floor((2i+1) sw / (2 dw)),
 * floor((2j+1) sh / (2 dh))) counted from the lower left, the texel under the
 * pixel's centre, in exact integer arithmetic.  dst: dw*dh*3 bytes.
 */
RT_API int rt_host_blit_nearest(const uint8_t* src_rgb8, int sw, int sh, uint8_t* dst_rgb8, int dw, int dh);

/* The window size the reference derives from a width and 16:9
 * (raytrace.cpp:84-88,121-128): height = max(1, int(width / (16.0 / 9.0))). */
RT_API int rt_host_height_for_width(int width);

/* Binary PPM (P6) of an RGB8 image, rows from the top.  0 or RT_ERR_INVALID. */
RT_API int rt_host_write_ppm(const char* path, const uint8_t* rgb8, int width, int height);

#ifdef __cplusplus
}
#endif
#endif
