/*
 * raytrace_amd.h — C ABI of the MI355X (gfx950) sphere path tracer.
 *
 * Everything the reference project (not-bottle/raytrace) does per pixel in its
 * fragment shader shaders/testFragment.fs runs behind these entry points as
 * hand-written HIP kernels.  Scenes are still described through the
 * reference's scene headers: their add() methods write 32-byte std140 records
 * through two OpenGL function pointers, and this library supplies the two
 * functions those pointers are pointed at (rt_gl_*), then consumes the captured
 * bytes in rt_scene_upload.
 *
 * Plain C, plain pointers and sizes.  All functions return 0 on success and a
 * negative rt_status on failure; rt_last_error() gives the message of the last
 * failure on the calling thread.  No exception crosses this boundary.
 * A context is single-caller: calls on one rt_ctx must not overlap.
 */
#ifndef RAYTRACE_AMD_H
#define RAYTRACE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_API __attribute__((visibility("default")))

typedef enum rt_status {
  RT_OK = 0,
  RT_ERR_INVALID = -1,  /* bad argument (null, negative size, out-of-range id)  */
  RT_ERR_DEVICE = -2,   /* a HIP call failed; message holds hipGetErrorString   */
  RT_ERR_STATE = -3,    /* call order violated (render before upload, ...)      */
  RT_ERR_NO_GPU = -4    /* no HIP device visible: there is no CPU fallback      */
} rt_status;

/* Material type ids.  0..3 are the reference's enum MATERIAL_TYPE
 * (material.h:9-14).  4 is an extension of this library: a lambertian whose
 * albedo is multiplied by perlin(k*p.x, k*p.z) of the hit point p, with
 * perlin() the 2-D noise of the reference's perlin.h:59-85 and k = param1. */
enum {
  RT_MAT_NOTHING = 0,
  RT_MAT_LAMBERTIAN = 1,
  RT_MAT_METALLIC = 2,
  RT_MAT_DIALECTRIC = 3,
  RT_MAT_PERLIN_LAMBERTIAN = 4
};

/* Intersection structure chosen at rt_scene_upload. */
enum {
  RT_ACCEL_LINEAR = 0, /* the reference's scan over all spheres, hit_any, testFragment.fs:165-199 */
  RT_ACCEL_LBVH = 1    /* Morton-code LBVH built on the GPU; same results as the scan            */
};

/* Random number source of rt_render. */
enum {
  /* xorshift32 seeded once per pixel and carried across that pixel's samples
   * and bounces (testFragment.fs:105-110, 326-345).  Needs sample_begin == 0
   * (RT_ERR_INVALID otherwise: a later range would repeat the first one's draws). */
  RT_RNG_XORSHIFT_REF = 0,
  /* Counter-based: the stream is a function of (seed, pixel, sample) only, so
   * an image does not depend on how it is split into tiles or sample ranges. */
  RT_RNG_COUNTER = 1
};

/* Byte size of one std140 sphere or material slot (sphere.h:19, material.h:22). */
#define RT_RECORD_BYTES 32

/*
 * The render call's arguments: the fragment shader's uniform block
 * (testFragment.fs:8-24, set at raytrace.cpp:522-533) plus what the GL draw
 * call supplied implicitly (viewport size) and this library's partitioning
 * (tile rectangle, sample range).  Vectors are x,y,z.
 */
typedef struct rt_params {
  int32_t width, height;           /* full image size (glViewport, raytrace.cpp:511)        */
  int32_t tile_x0, tile_y0;        /* pixels [x0,x1) x [y0,y1) are rendered by this call;   */
  int32_t tile_x1, tile_y1;        /*   the other pixels of the accumulator are untouched   */
  int32_t sample_begin, sample_end;/* samples [begin,end) of each pixel (num_samples = end
                                      when begin == 0)                                      */
  uint32_t bounce_limit;           /* uniform bounce_limit                                  */
  uint32_t time_u32t;              /* uniform time_u32t: accepted, never read (as in the
                                      shader, testFragment.fs:8)                            */
  float delta_u[3];
  float delta_v[3];
  float camera_origin[3];
  float viewport_top_left[3];
  float defocus_disk_u[3];
  float defocus_disk_v[3];
  float defocus_angle;
  int32_t rng_mode;                /* RT_RNG_*                                              */
  uint32_t seed;                   /* RT_RNG_COUNTER key                                    */
  /* Interleaved shares of the tile (no counterpart in the reference; 0, 0 = the whole tile): the
   * tile's rows are taken in strips of eight, and this call renders the strips whose number
   * (from the top of the tile) is strip_index modulo strip_count.  The shares of strip_count
   * calls (or GPUs) are disjoint, cover the tile, and cost about the same each, which row bands
   * of a frame with sky above and spheres below do not.                                       */
  uint16_t strip_count, strip_index;
  /* RT_RNG_XORSHIFT_REF only: width*height words, row y of the image at
   * [y*width, (y+1)*width): the bits of (tex.r + tex.g + tex.b) that the shader
   * reads at testFragment.fs:105-106.  Host pointer; NULL means all zero.  It is
   * copied to the device asynchronously: keep it valid and unchanged until the
   * next rt_synchronize / rt_readback on the context.                           */
  const uint32_t* seed_tex;
} rt_params;

/* Work counters of one instrumented render or trace (rt_get_counters). */
typedef struct rt_counters {
  uint64_t paths;          /* camera samples followed to termination                    */
  uint64_t segments;       /* hit_any calls (one per ray segment)                       */
  uint64_t sphere_tests;   /* hit_sphere calls                                          */
  uint64_t roots;          /* quadratic roots evaluated inside hit_sphere               */
  uint64_t node_tests;     /* LBVH box tests (0 for the linear scan)                    */
  uint64_t hits;           /* segments that recorded a hit                              */
  uint64_t sky;            /* segments that missed everything                           */
  uint64_t lambertian;     /* scatter calls by kind                                     */
  uint64_t metallic;
  uint64_t dialectric;
  uint64_t perlin;         /* perlin-lambertian scatters                                */
  uint64_t unit_tries;     /* tries of the random_unit_vector rejection loop            */
  uint64_t disk_tries;     /* tries of the random_unit_disk rejection loop              */
  uint64_t limit_hits;     /* paths ended by the bounce limit                           */
  /* SIMD occupancy of the three phases of the render / trace loops (node
   * step, path phase, leaf step): how many times a wave issued the phase,
   * and how many lanes took part in total (lanes / (64 * waves) = fraction
   * of the wave doing useful work).                                           */
  uint64_t trav_wave_steps;
  uint64_t trav_lane_steps;
  uint64_t path_wave_phases;
  uint64_t path_lane_phases;
  uint64_t leaf_wave_steps;
  uint64_t leaf_lane_steps;
} rt_counters;

typedef struct rt_ctx rt_ctx;

/* Message of the last failure on this thread ("" if none). */
RT_API const char* rt_last_error(void);

/*
 * Replaces: SDL/GL start-up, init() raytrace.cpp:121-184 (context creation is
 * all that is kept of it).  Creates a context on HIP device `device`.
 * Fails with RT_ERR_NO_GPU when no device is visible.
 */
RT_API int rt_init(int device, rt_ctx** out);

/* Replaces: close() raytrace.cpp:214-219.  Frees device memory and the stream. */
RT_API void rt_destroy(rt_ctx* ctx);

/*
 * Replaces: glBindBuffer, the function behind the glad pointer
 * glad_glBindBuffer (include/glad/glad.h:1812-1814) that sphere::add
 * (sphere.h:24-34) and material::add (material.h:30-44) call.  Remembers the
 * buffer bound to `target`; buffer 0 unbinds.
 */
RT_API void rt_gl_bind_buffer(uint32_t target, uint32_t buffer);

/*
 * Replaces: glBufferSubData, behind glad_glBufferSubData
 * (include/glad/glad.h:1827-1829).  Copies `size` bytes at `offset` into the
 * host image of the buffer currently bound to `target`, growing it as needed.
 * A write with nothing bound is dropped and counted (rt_gl_dropped_writes).
 */
RT_API void rt_gl_buffer_sub_data(uint32_t target, intptr_t offset, intptr_t size, const void* data);

/*
 * Replaces: glBufferData, behind glad_glBufferData (called at
 * raytrace.cpp:384,419 to allocate each uniform buffer).  (Re)creates the host
 * image of the bound buffer with `size` bytes, zero-filled when data is NULL.
 */
RT_API void rt_gl_buffer_data(uint32_t target, intptr_t size, const void* data, uint32_t usage);

/* Captured bytes of a buffer id: pointer valid until the next capture call on
 * this thread; *size receives the byte count (0 and NULL if never written). */
RT_API const void* rt_gl_buffer_bytes(uint32_t buffer, size_t* size);
/* Forget every captured buffer of this thread. */
RT_API void rt_gl_reset(void);
RT_API uint64_t rt_gl_dropped_writes(void);

/*
 * Replaces: the two uniform blocks "Spheres" and "Materials" the shader reads
 * (testFragment.fs:65-73; allocated and bound at raytrace.cpp:381-391,
 * 416-426).  Takes the std140 bytes the scene headers produced (32-byte slots:
 * sphere = int mat @0, float radius @4, vec3 origin @16, sphere.h:22-35;
 * material = int id @0, int type @4, float param1 @8, vec3 albedo @16,
 * material.h:29-46), flattens them to structure-of-arrays device buffers and,
 * for RT_ACCEL_LBVH, builds the hierarchy (on the host for a few thousand
 * spheres, on the GPU beyond).  A scene of at most 16 spheres gets no
 * hierarchy whatever `accel` says: the scan is the faster search there, with
 * the same answers (rt_get_scene_info then reports builder RT_BUILDER_NONE).
 * The input is copied; the caller keeps ownership.  There is no 64-object limit.
 * Fails with RT_ERR_INVALID if a sphere names a material slot >= n_materials.
 */
RT_API int rt_scene_upload(rt_ctx* ctx, const void* spheres_std140, int n_spheres,
                           const void* materials_std140, int n_materials, int accel);

/*
 * Scenes with a second kind of object.  The reference's hittable is a
 * polymorphic base whose records carry their own byte size (hittable.h:9-14,
 * hittable_list.h:13-18: offset += h.size), and sphere is its only subclass;
 * this entry point takes such a list of records of mixed size.  Each record
 * holds its kind as an int at byte 8 (a word sphere::add leaves untouched,
 * sphere.h:22-35, so sphere records written into a zeroed buffer are kind 0):
 *   RT_OBJECT_SPHERE  32 bytes, as in rt_scene_upload
 *   RT_OBJECT_QUAD    64 bytes: int mat @0, int kind @8, vec3 Q @16 (a corner),
 *                     vec3 u @32, vec3 v @48 (the two edges from Q)
 * `objects` holds n_objects records back to back in n_bytes bytes.  A quad is
 * the parallelogram Q + a u + b v, 0 <= a, b <= 1, hit from either side; its
 * normal faces the ray, as a sphere's does.  The intersection arithmetic, in
 * binary32 with one IEEE operation per step, is stated where hit_quad is
 * defined (csrc/rt_device.cuh) and restated by the oracle.  Objects are
 * numbered in upload order, spheres and quads alike; of two hits at the same
 * distance the lower number wins, as in the scan over spheres.
 * A scene with quads is searched by the scan or by the hierarchy read from
 * device memory, whose leaves then carry either kind.
 */
enum { RT_OBJECT_SPHERE = 0, RT_OBJECT_QUAD = 1 };
#define RT_QUAD_RECORD_BYTES 64
RT_API int rt_scene_upload_objects(rt_ctx* ctx, const void* objects, size_t n_bytes, int n_objects,
                                   const void* materials_std140, int n_materials, int accel);

/*
 * Replaces: the draw that runs the fragment shader over the viewport,
 * glDrawElements at raytrace.cpp:536 (and the never-called tiled variant
 * shader_chunk_pass, raytrace.cpp:598-630, whose c_min/c_max this call's tile
 * rectangle implements).  Adds the radiance of samples [sample_begin,
 * sample_end) of every pixel of the tile to the context's FP32 accumulator
 * (allocated and zeroed when the image size changes or rt_clear is called).
 * Asynchronous on the context's stream.
 * The sample count rt_readback divides by is end - begin of the union of the
 * ranges rendered since the last rt_clear: render each sample of a pixel once,
 * in ranges without gaps, and the same ranges over every tile (or state the
 * count with rt_set_accumulated_samples).
 */
RT_API int rt_render(rt_ctx* ctx, const rt_params* p);

/* Same render with the work counters switched on (slower; not for timing). */
RT_API int rt_render_counted(rt_ctx* ctx, const rt_params* p);

/*
 * Device memory rt_render may take for its schedule (no counterpart in the
 * reference, whose draw call needs none): split_bytes for the per-sample
 * radiance of pixels cut into several jobs (default 32 GiB), probe_bytes for
 * that of the cost-probe pass (default 4 GiB); 0 keeps a default.  Whatever the
 * budgets, a render takes at most half / an eighth of the memory the device
 * can still give, falls back to schedules that need less when an allocation
 * fails, and gives the image's bits regardless: the budgets trade speed only.
 * Buffers held beyond a new budget are freed by this call.
 */
RT_API int rt_set_scratch_budget(rt_ctx* ctx, size_t split_bytes, size_t probe_bytes);

/* Zero the accumulator and the accumulated-sample count. */
RT_API int rt_clear(rt_ctx* ctx);

/*
 * Use caller-owned device memory (width*height*3 floats, RGB interleaved) as
 * the accumulator, e.g. a tensor a collective library will reduce across GPUs.
 * NULL returns to the context-owned buffer.  The memory is not cleared here.
 */
RT_API int rt_set_accumulator(rt_ctx* ctx, void* device_rgb, int width, int height);

/* Set how many samples per pixel the accumulator holds (after an external
 * reduction added other GPUs' sample ranges into it). */
RT_API int rt_set_accumulated_samples(rt_ctx* ctx, int samples);

/*
 * Checkpoint and resume of a progressive render.  rt_accumulator_read copies
 * out the raw per-pixel radiance sums (width*height*3 floats, host memory;
 * sums may be NULL to query the size) and how many samples per pixel they
 * hold; rt_accumulator_write puts such sums back (into the caller's
 * accumulator if one is set, else the context's own) so that rendering goes on
 * with sample_begin = samples.  With RT_RNG_COUNTER the finished image equals,
 * bit for bit, the one an uninterrupted render gives: a pixel's sum is formed
 * in sample order either way.  No counterpart in the reference, which renders
 * once and keeps nothing.
 */
RT_API int rt_accumulator_read(rt_ctx* ctx, float* sums, int* width, int* height, int* samples);
RT_API int rt_accumulator_write(rt_ctx* ctx, const float* sums, int width, int height, int samples);

/*
 * Replaces: shader_chunk_pass(vec2 c_min, vec2 c_max, Shader, Camera, fb_help
 * fb, fb_help tex, hittable_list), raytrace.cpp:55,598-630 — the reference's
 * (unfinished, never called) pass over one chunk of the frame.  Renders the
 * sample range of p over pixels [c_min_x, c_max_x) x [c_min_y, c_max_y) and
 * adds them to the accumulator; p's own tile rectangle is ignored.  Chunks and
 * sample ranges may be issued in any order (RT_RNG_COUNTER) and compose to the
 * full frame.
 */
RT_API int rt_render_chunk(rt_ctx* ctx, const rt_params* p, int c_min_x, int c_min_y, int c_max_x, int c_max_y);

/*
 * Replaces: the end of the shader's main (colour / num_samples and the
 * 1/2.2 gamma, testFragment.fs:135-140) and the RGB8 colour attachment the
 * result lands in (raytrace.cpp:574).  Waits for the stream.  rgb_linear
 * (nullable) receives width*height*3 floats, the mean radiance before gamma;
 * rgb8 (nullable) receives width*height*3 bytes, gamma-encoded, clamped to
 * [0,1] and rounded to nearest.  Row 0 is the top row (origin_upper_left,
 * testFragment.fs:3).  Host pointers.
 */
RT_API int rt_readback(rt_ctx* ctx, float* rgb_linear, uint8_t* rgb8);

/*
 * Replaces: hit_any, testFragment.fs:165-199, called on a batch of rays
 * instead of from inside a path.  o_xyz, d_xyz: n rays, 3 floats each
 * (direction need not be normalised).  id: index of the closest sphere in
 * upload order, or -1; t: ray parameter of the hit (-1 on a miss); nrm: the
 * hit record's normal, 3 floats (flipped against the ray as at :190-193;
 * 0,0,0 on a miss).  Host pointers; copies in and out.  Uses the accel chosen
 * at upload; both give the same answers.
 */
RT_API int rt_trace_batch(rt_ctx* ctx, const float* o_xyz, const float* d_xyz, size_t n,
                          int32_t* id, float* t, float* nrm);

/* Same, on device pointers already resident in HBM; asynchronous on the
 * context's stream.  `counted` != 0 switches the work counters on. */
RT_API int rt_trace_batch_device(rt_ctx* ctx, const float* o_xyz, const float* d_xyz, size_t n,
                                 int32_t* id, float* t, float* nrm, int counted);

/*
 * rt_trace_batch_device for callers that cannot allocate: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  It has no
 * counterpart in the reference.  Same arguments (device pointers), same
 * results bit for bit, asynchronous on the context's stream, no work counters;
 * it calls no hipMalloc / hipFree and no allocating library routine.  The
 * kernel's own working memory (the head of its ray queue and, for hierarchies
 * read from HBM, the part of the lanes' traversal stacks that does not fit
 * LDS) comes from a fixed array of RT_NOALLOC_SCRATCH_BYTES that every GPU the
 * library runs on holds until the process ends.  Limits: at most
 * RT_NOALLOC_MAX_RAYS rays per call (RT_ERR_INVALID beyond), and a hierarchy
 * whose depth lets the stack overflow of at least one workgroup per compute
 * unit fit the array (RT_ERR_INVALID otherwise: about depth 150 for the
 * HBM-resident trees; scenes held in LDS and the linear scan always fit).
 * A refused call queues nothing and leaves rt_last_kernel_ms as it was.  When
 * the context's stream is being captured (rt_get_stream) the call records the
 * memset of the queue head and the kernel into the graph and no timing events:
 * rt_last_kernel_ms then fails with RT_ERR_STATE until an uncaptured launch.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU
 * (two contexts, two streams, two replays of a captured graph) race; issue them
 * one after another.
 */
#define RT_NOALLOC_SCRATCH_BYTES (64u << 20)
#define RT_NOALLOC_MAX_RAYS 0x80000000ull
RT_API int rt_trace_batch_noalloc(rt_ctx* ctx, const float* o_xyz, const float* d_xyz, size_t n,
                                  int32_t* id, float* t, float* nrm);

/* Wait for everything queued on the context's stream. */
RT_API int rt_synchronize(rt_ctx* ctx);

/* Counters of the last *_counted call (waits for the stream). */
RT_API int rt_get_counters(rt_ctx* ctx, rt_counters* out);

/* Device time in milliseconds of the last render or trace kernel, measured
 * with HIP events recorded on the context's stream around the launch (waits
 * for the stream).  *kernel_name (nullable) receives a static string. */
RT_API int rt_last_kernel_ms(rt_ctx* ctx, float* ms, const char** kernel_name);

/* Device time of the last LBVH build (all of its kernels), in milliseconds. */
RT_API int rt_last_build_ms(rt_ctx* ctx, float* ms);

/* What rt_scene_upload made of the scene (no counterpart in the reference, which has only its scan). */
enum { RT_BUILDER_NONE = 0, RT_BUILDER_HOST_SAH = 1, RT_BUILDER_DEVICE_LBVH = 2 };
typedef struct rt_scene_info {
  int32_t n_spheres, n_materials;
  int32_t accel;        /* RT_ACCEL_* asked for                                                     */
  int32_t builder;      /* RT_BUILDER_*: surface-area tree built on the host (scenes that fit LDS),
                           Morton-order LBVH built on the device, or none (linear scan)             */
  int32_t n_big;        /* spheres kept out of the tree and tested exactly for every ray            */
  int32_t n_leaf;       /* spheres in the tree                                                      */
  int32_t n_nodes;      /* 4-wide nodes                                                             */
  int32_t depth;        /* binary levels (rt_lbvh_depth)                                            */
  int32_t tree_in_lds;  /* 1: the kernels keep tree, hit-record and material arrays in LDS          */
  int32_t pad_per_ray;  /* 1: one box pad per ray, 0: per child box                                 */
  int32_t node_bytes;   /* bytes of one node as the kernels read it                                 */
  int32_t reserved[5];
} rt_scene_info;
RT_API int rt_get_scene_info(rt_ctx* ctx, rt_scene_info* out);

/*
 * The HIP stream (a hipStream_t) every asynchronous call of the context is
 * queued on, for callers that order their own device work against it without
 * a host wait: record an event on it and make another stream wait, e.g. before
 * a collective sums the accumulator (no counterpart in the reference).
 */
RT_API int rt_get_stream(rt_ctx* ctx, void** hip_stream);

/*
 * The sum of the ranks' accumulators as a library call (no counterpart in the
 * reference, which renders on one device): one context per GPU, one process
 * per context, RCCL over xGMI underneath, queued on the context's own stream
 * behind the renders it follows — no host wait, no other runtime involved.
 *   rt_comm_unique_id   on one rank: 128 bytes (RT_COMM_ID_BYTES) that the caller
 *                       hands to every other rank by whatever means it has
 *   rt_comm_init        every rank, with the same id: joins the group of `world`
 *                       ranks as `rank` on the context's device
 *   rt_comm_reduce      every rank, after its renders: adds the accumulators of
 *                       all ranks (same image size everywhere) onto rank `root`'s;
 *                       the other ranks' accumulators are left as they were.
 *                       Sample counts are not exchanged: the root calls
 *                       rt_set_accumulated_samples if the ranks rendered
 *                       different sample ranges.
 *   rt_comm_destroy     leaves the group (rt_destroy does it too)
 * RCCL is loaded when rt_comm_unique_id or rt_comm_init is first called
 * (librccl.so.1; a process that already holds one, PyTorch's for instance,
 * shares it); RT_ERR_DEVICE if it cannot be had.
 */
#define RT_COMM_ID_BYTES 128
RT_API int rt_comm_unique_id(void* id_out);
RT_API int rt_comm_init(rt_ctx* ctx, const void* id, int rank, int world);
RT_API int rt_comm_reduce(rt_ctx* ctx, int root);
RT_API int rt_comm_destroy(rt_ctx* ctx);

/* Depth of the LBVH of the uploaded scene (largest number of ancestors of a
 * leaf; it sizes the traversal stack).  0 for the linear scan. */
RT_API int rt_lbvh_depth(rt_ctx* ctx);

/* The seed image of the reference, ready for rt_params.seed_tex: what
 * shaders/perlin.fs draws into an RGB8 texture (raytrace.cpp:309-328; the
 * reference calls it with GRID_SIZE 400, CONTRAST 1.5, LAYERS 12 and, through
 * the slip at :326, LACURNITY 1.5) as the ray shader reads it back
 * (testFragment.fs:105-106: the bits of r+g+b of the texel under the pixel,
 * rows flipped because the ray shader counts rows from the top).  GLSL leaves
 * the precision of sin() to the driver, so no two GL devices agree on this
 * image; here sin is the double-precision sine rounded to binary32.
 * seed_tex: width*height words, host memory.                                 */
RT_API int rt_seed_texture(rt_ctx* ctx, int width, int height, float grid_size, float contrast, int layers,
                           float lacunarity, uint32_t* seed_tex);

/*
 * Replaces: perlin(float x, float y), perlin.h:59-85 — the reference's 2-D
 * gradient noise (linear interpolation, result 0.5 * v + 0.5), evaluated by
 * the same device function the noise-textured material (RT_MAT_PERLIN_
 * LAMBERTIAN) calls.  xy: n points, 2 floats each; value: n floats; host
 * pointers.  use_table != 0 reads the gradients of lattice points inside
 * [-2048, 2048)^2 from the context's precomputed table (built on first use),
 * as the render kernels do; 0 computes every gradient on the spot.  Both give
 * the bits of the reference's function.
 */
RT_API int rt_eval_perlin(rt_ctx* ctx, const float* xy, size_t n, int use_table, float* value);

/*
 * Replaces: randomGradient(int ix, int iy), perlin.h:31-43.  ixy: n lattice
 * points, 2 ints each; gxy: n gradients, 2 floats each (x = cos, y = sin of
 * the hashed angle); host pointers.
 */
RT_API int rt_eval_perlin_gradient(rt_ctx* ctx, const int32_t* ixy, size_t n, float* gxy);

/* Number of HIP devices visible (0 if none or the runtime cannot start). */
RT_API int rt_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* RAYTRACE_AMD_H */
