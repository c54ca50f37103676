// rt_kernels.cuh — the path-tracing megakernel, the batched closest-hit kernel
// and the resolve (mean + gamma + RGB8) kernel.
#pragma once

#include "rt_scene_dev.cuh"

struct render_args {
  rt_params p;                 // seed_tex inside is a host pointer and is not used on the device
  scene_dev sc;
  float* accum;                // width*height*3 radiance sums
  const uint32_t* seed_tex;    // device copy of p.seed_tex, or null
  unsigned long long* counters;  // sizeof(rt_counters)/8 words, or null
};

template <bool COUNTED>
RT_DEV void flush_counters(unsigned long long* g, const lane_counters& k) {
  if (!COUNTED || g == nullptr) return;
  const uint32_t v[14] = {k.paths,      k.segments, k.sphere_tests, k.roots,  k.node_tests, k.hits,       k.sky,
                          k.lambertian, k.metallic, k.dialectric,   k.perlin, k.unit_tries, k.disk_tries, k.limit_hits};
#pragma unroll
  for (int i = 0; i < 14; i++) {
    // sum over the wave first: one atomic per wave and counter
    unsigned long long s = v[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0 && s != 0) atomicAdd(g + i, s);
  }
}

// LDS traversal stack: STACK entries x 64 lanes x (ref, tn) per wave.
template <int ACCEL, int STACK>
struct wave_stack {
  static constexpr int words = (ACCEL == RT_ACCEL_LBVH) ? STACK * 64 : 1;
};

// One lane = one pixel.  The lane walks through its samples; whenever its path
// ends it starts the next sample at once (regeneration), so a wave keeps all
// lanes busy until the pixels' sample budgets run out, and the per-pixel sum
// is formed in sample order in a register — no atomics, and bit-identical to
// a sequential evaluation.  A wave covers an 8x8 pixel tile (coherent primary
// rays), a 256-thread workgroup a 16x16 tile.
template <int ACCEL, int RNG, bool COUNTED, int STACK>
__global__ __launch_bounds__(256) void rt_render_kernel(const render_args A) {
  __shared__ int s_ref[4 * wave_stack<ACCEL, STACK>::words];
  __shared__ float s_tn[4 * wave_stack<ACCEL, STACK>::words];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int* stack_ref = s_ref + wave * wave_stack<ACCEL, STACK>::words + (ACCEL == RT_ACCEL_LBVH ? lane : 0);
  float* stack_tn = s_tn + wave * wave_stack<ACCEL, STACK>::words + (ACCEL == RT_ACCEL_LBVH ? lane : 0);

  const rt_params& P = A.p;
  const scene_dev& sc = A.sc;
  const int x = P.tile_x0 + (int)blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
  const int y = P.tile_y0 + (int)blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
  const bool inside = x < P.tile_x1 && y < P.tile_y1;
  const uint32_t pixel = (uint32_t)y * (uint32_t)P.width + (uint32_t)x;

  lane_counters k = {};
  const f3 delta_u = mk3(P.delta_u[0], P.delta_u[1], P.delta_u[2]);
  const f3 delta_v = mk3(P.delta_v[0], P.delta_v[1], P.delta_v[2]);
  const f3 cam = mk3(P.camera_origin[0], P.camera_origin[1], P.camera_origin[2]);
  const f3 vtl = mk3(P.viewport_top_left[0], P.viewport_top_left[1], P.viewport_top_left[2]);
  const f3 ddu = mk3(P.defocus_disk_u[0], P.defocus_disk_u[1], P.defocus_disk_u[2]);
  const f3 ddv = mk3(P.defocus_disk_v[0], P.defocus_disk_v[1], P.defocus_disk_v[2]);

  uint32_t rng = 0;
  if (RNG == RT_RNG_XORSHIFT_REF && inside) {
    uint32_t seed = A.seed_tex ? A.seed_tex[pixel] : 0u;  // fs:105-106
    rng = seed ^ cantor_pair((uint32_t)x, (uint32_t)y);   // fs:107
  }

  int s = P.sample_begin;
  bool alive = false;
  bool done = !inside || s >= P.sample_end;
  f3 colour = mk3(0.0f, 0.0f, 0.0f);
  f3 o = colour, d = colour, alb = colour;
  uint32_t count = 0;

  while (!done) {
    if (!alive) {  // fs:117-131: next camera sample of this pixel
      if (RNG == RT_RNG_COUNTER) rng = counter_key(P.seed, pixel, (uint32_t)s);
      const float jx = rng_float<RNG>(rng) - 0.5f;
      const float jy = rng_float<RNG>(rng) - 0.5f;
      const f3 frag = vtl + ((float)x + jx) * delta_u + ((float)y + jy) * delta_v;
      f3 ro = cam;
      if (!(P.defocus_angle <= 0.0f)) {
        const f3 rd = random_unit_disk<RNG, COUNTED>(rng, k);
        ro = cam + rd.x * ddu + rd.y * ddv;
      }
      o = ro;
      d = frag - ro;
      alb = mk3(1.0f, 1.0f, 1.0f);
      count = 0;
      alive = true;
      RT_COUNT(k, paths, 1);
    }
    bool cont;
    if (count >= P.bounce_limit) {  // fs:203-205
      alb = mk3(0.0f, 0.0f, 0.0f);
      cont = false;
      RT_COUNT(k, limit_hits, 1);
    } else {
      RT_COUNT(k, segments, 1);
      float t;
      int idx;
      if (ACCEL == RT_ACCEL_LBVH)
        hit_bvh<COUNTED>(sc, o, d, stack_ref, stack_tn, t, idx, k);
      else
        hit_linear<COUNTED>(sc, o, d, t, idx, k);
      if (t > 0.001f) {  // fs:186-196
        RT_COUNT(k, hits, 1);
        const float4 sp = sc.sph[idx];
        hit_rec h;
        h.point = o + d * t;
        h.normal = sc.inv_r[idx] * (h.point - mk3(sp.x, sp.y, sp.z));
        h.interior = false;
        if (dot3(h.normal, d) >= 0.0f) {
          h.normal = -h.normal;
          h.interior = true;
        }
        h.mat = sc.sph_mat[idx];
        const float4 ma = sc.mat_a[h.mat];
        const int type = sc.mat_type[h.mat];
        o = h.point;  // fs:211-212
        count = count + 1u;
        cont = shade_hit<RNG, COUNTED>(h, type, ma.w, mk3(ma.x, ma.y, ma.z), d, alb, rng, k);
      } else {  // fs:217-218
        alb = sky_radiance(d, alb);
        cont = false;
        RT_COUNT(k, sky, 1);
      }
    }
    if (!cont) {  // fs:132: colour += raycast(...)
      colour = colour + alb;
      alive = false;
      s++;
      done = s >= P.sample_end;
    }
  }

  if (inside) {
    float* dst = A.accum + 3 * (size_t)pixel;
    dst[0] += colour.x;
    dst[1] += colour.y;
    dst[2] += colour.z;
  }
  flush_counters<COUNTED>(A.counters, k);
}

struct trace_args {
  scene_dev sc;
  const float* o_xyz;
  const float* d_xyz;
  size_t n;
  int32_t* id;
  float* t;
  float* nrm;
  unsigned long long* counters;
};

// hit_any (fs:165-199) for a batch of rays, one ray per lane.
template <int ACCEL, bool COUNTED, int STACK>
__global__ __launch_bounds__(256) void rt_trace_kernel(const trace_args A) {
  __shared__ int s_ref[4 * wave_stack<ACCEL, STACK>::words];
  __shared__ float s_tn[4 * wave_stack<ACCEL, STACK>::words];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int* stack_ref = s_ref + wave * wave_stack<ACCEL, STACK>::words + (ACCEL == RT_ACCEL_LBVH ? lane : 0);
  float* stack_tn = s_tn + wave * wave_stack<ACCEL, STACK>::words + (ACCEL == RT_ACCEL_LBVH ? lane : 0);
  const scene_dev& sc = A.sc;
  lane_counters k = {};
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < A.n) {
    const f3 o = mk3(A.o_xyz[3 * i], A.o_xyz[3 * i + 1], A.o_xyz[3 * i + 2]);
    const f3 d = mk3(A.d_xyz[3 * i], A.d_xyz[3 * i + 1], A.d_xyz[3 * i + 2]);
    RT_COUNT(k, segments, 1);
    float t;
    int idx;
    if (ACCEL == RT_ACCEL_LBVH)
      hit_bvh<COUNTED>(sc, o, d, stack_ref, stack_tn, t, idx, k);
    else
      hit_linear<COUNTED>(sc, o, d, t, idx, k);
    f3 n = mk3(0.0f, 0.0f, 0.0f);
    int id = -1;
    float tt = -1.0f;
    if (t > 0.001f) {
      RT_COUNT(k, hits, 1);
      const float4 sp = sc.sph[idx];
      const f3 p = o + d * t;
      n = sc.inv_r[idx] * (p - mk3(sp.x, sp.y, sp.z));
      if (dot3(n, d) >= 0.0f) n = -n;
      id = idx;
      tt = t;
    }
    A.id[i] = id;
    A.t[i] = tt;
    A.nrm[3 * i] = n.x;
    A.nrm[3 * i + 1] = n.y;
    A.nrm[3 * i + 2] = n.z;
  }
  flush_counters<COUNTED>(A.counters, k);
}

// End of the shader's main (fs:135-140) and the RGB8 store: mean over the
// samples, gamma 1/2.2, clamp, round to nearest.
__global__ __launch_bounds__(256) void rt_resolve_kernel(const float* accum, size_t n_values, float inv_samples,
                                                         float* rgb_linear, uint8_t* rgb8) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_values) return;
  const float c = accum[i] * inv_samples;
  if (rgb_linear) rgb_linear[i] = c;
  if (rgb8) {
    float g = powf(c, 1.0f / 2.2f);
    if (!(g > 0.0f)) g = 0.0f;
    if (g > 1.0f) g = 1.0f;
    rgb8[i] = (uint8_t)floorf(g * 255.0f + 0.5f);
  }
}
