// rt_kernels.cuh — the path-tracing megakernel, the batched closest-hit kernel
// and the resolve (mean + gamma + RGB8) kernel.
#pragma once

#include "rt_scene_dev.cuh"

// How a kernel finds the closest hit.
//   SEARCH_LINEAR      the reference's scan over every sphere
//   SEARCH_TREE_LDS_*  4-wide LBVH staged in LDS (small scenes): one copy per
//                      1024-thread workgroup, one-word stack entries
//   SEARCH_TREE_GLB_*  4-wide LBVH read from HBM / caches (large scenes)
//   *_RAY / *_NODE     the box-growing rule (rt_scene_dev.cuh)
enum {
  SEARCH_LINEAR = 0,
  SEARCH_TREE_LDS_RAY = 1,
  SEARCH_TREE_LDS_NODE = 2,
  SEARCH_TREE_GLB_RAY = 3,
  SEARCH_TREE_GLB_NODE = 4
};

template <int SEARCH>
struct search_traits {
  static constexpr bool tree = SEARCH != SEARCH_LINEAR;
  static constexpr bool lds = SEARCH == SEARCH_TREE_LDS_RAY || SEARCH == SEARCH_TREE_LDS_NODE;
  static constexpr bool pad_ray = SEARCH == SEARCH_TREE_LDS_RAY || SEARCH == SEARCH_TREE_GLB_RAY;
  static constexpr int waves = lds ? 16 : 4;       // waves per workgroup
  static constexpr int threads = waves * 64;
  static constexpr int waves_x = lds ? 4 : 2;      // 8x8-pixel wave tiles across a workgroup tile
  static constexpr int tile = waves_x * 8;         // workgroup tile edge in pixels
  static constexpr int stack = lds ? 16 : 32;      // entries per lane
  static constexpr int stack_words = !tree ? 0 : (lds ? stack * 64 : stack * 128);  // per wave
};

// Bytes of dynamic LDS a launch needs.
template <int SEARCH>
inline size_t search_lds_bytes(int n_nodes, int n_leaf) {
  using T = search_traits<SEARCH>;
  size_t b = (size_t)T::waves * T::stack_words * 4;
  if (T::lds) b += (size_t)n_nodes * sizeof(bvh4_node) + (size_t)n_leaf * 16 + (((size_t)n_leaf * 4 + 15) & ~(size_t)15);
  return b;
}

struct render_args {
  rt_params p;                 // seed_tex inside is a host pointer and is not used on the device
  scene_dev sc;
  float* accum;                // width*height*3 radiance sums
  const uint32_t* seed_tex;    // device copy of p.seed_tex, or null
  unsigned long long* counters;  // sizeof(rt_counters)/8 words, or null
  unsigned int* queue;         // head of the pixel queue, zero at launch
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

extern __shared__ __attribute__((aligned(16))) unsigned char rt_lds[];

// Carve the dynamic LDS: [nodes | leaf spheres | leaf indices] (LDS searches
// only) then the traversal stacks; copy the tree in.  Every thread of the
// workgroup must call this before the first search.
template <int SEARCH>
RT_DEV void search_setup(const scene_dev& sc, tree_view& tv, uint32_t*& stack_col) {
  using T = search_traits<SEARCH>;
  tv.nodes = sc.nodes;
  tv.leaf_sph = sc.leaf_sph;
  tv.leaf_idx = sc.leaf_idx;
  stack_col = nullptr;
  if (!T::tree) return;
  unsigned char* base = rt_lds;
  if (T::lds) {
    const size_t node_bytes = (size_t)sc.n_nodes * sizeof(bvh4_node);
    const size_t leaf_bytes = (size_t)sc.n_leaf * 16;
    const size_t idx_bytes = ((size_t)sc.n_leaf * 4 + 15) & ~(size_t)15;
    float4* dn = reinterpret_cast<float4*>(base);
    float4* dl = reinterpret_cast<float4*>(base + node_bytes);
    int32_t* di = reinterpret_cast<int32_t*>(base + node_bytes + leaf_bytes);
    const float4* sn = reinterpret_cast<const float4*>(sc.nodes);
    for (int i = threadIdx.x; i < sc.n_nodes * 8; i += T::threads) dn[i] = sn[i];
    for (int i = threadIdx.x; i < sc.n_leaf; i += T::threads) {
      dl[i] = sc.leaf_sph[i];
      di[i] = sc.leaf_idx[i];
    }
    tv.nodes = reinterpret_cast<const bvh4_node*>(dn);
    tv.leaf_sph = dl;
    tv.leaf_idx = di;
    base += node_bytes + leaf_bytes + idx_bytes;
    __syncthreads();
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  stack_col = reinterpret_cast<uint32_t*>(base) + wave * T::stack_words + lane;
}

template <int SEARCH, bool COUNTED>
RT_DEV void closest_hit(const scene_dev& sc, const tree_view& tv, uint32_t* stack_col, f3 o, f3 d, float& t, int& idx,
                        lane_counters& k) {
  using T = search_traits<SEARCH>;
  if (T::tree) {
    trav_stack<T::lds> stk;
    stk.col = stack_col;
    hit_bvh4<T::pad_ray, T::lds, T::stack, COUNTED>(sc, tv, o, d, stk, t, idx, k);
  } else {
    hit_linear<COUNTED>(sc, o, d, t, idx, k);
  }
}

// Persistent lanes.  The launch holds as many workgroups as the chip keeps
// resident; every lane pulls pixels from one global queue (the compiler folds
// the per-lane atomicAdd into one atomic per wave, ranked by ballot / mbcnt)
// and keeps pulling until the queue is dry, so no wave waits for the slowest
// pixel of a fixed tile.  Queue order: 8x8-pixel tiles, a tile's 64 pixels
// consecutive, so a wave starts on one coherent tile.  A lane owns its pixel
// for the whole sample range: it regenerates a camera ray the moment a path
// ends and forms the pixel's sum in sample order in registers — no atomics on
// the image, and bit-identical to a sequential evaluation.
template <int SEARCH, int RNG, bool COUNTED>
__global__ __launch_bounds__(search_traits<SEARCH>::threads) void rt_render_kernel(const render_args A) {
  const rt_params& P = A.p;
  const scene_dev& sc = A.sc;
  tree_view tv;
  uint32_t* stack_col;
  search_setup<SEARCH>(sc, tv, stack_col);

  const int tw = P.tile_x1 - P.tile_x0, th = P.tile_y1 - P.tile_y0;
  const int tiles_x = (tw + 7) >> 3, tiles_y = (th + 7) >> 3;
  const uint32_t total_items = (uint32_t)tiles_x * (uint32_t)tiles_y * 64u;

  lane_counters k = {};
  const f3 delta_u = mk3(P.delta_u[0], P.delta_u[1], P.delta_u[2]);
  const f3 delta_v = mk3(P.delta_v[0], P.delta_v[1], P.delta_v[2]);
  const f3 cam = mk3(P.camera_origin[0], P.camera_origin[1], P.camera_origin[2]);
  const f3 vtl = mk3(P.viewport_top_left[0], P.viewport_top_left[1], P.viewport_top_left[2]);
  const f3 ddu = mk3(P.defocus_disk_u[0], P.defocus_disk_u[1], P.defocus_disk_u[2]);
  const f3 ddv = mk3(P.defocus_disk_v[0], P.defocus_disk_v[1], P.defocus_disk_v[2]);

  bool have = false;       // this lane owns a pixel
  bool exhausted = P.sample_end <= P.sample_begin;
  int x = 0, y = 0, s = 0;
  uint32_t pixel = 0, rng = 0, count = 0;
  bool alive = false;
  f3 colour = mk3(0.0f, 0.0f, 0.0f);
  f3 o = colour, d = colour, alb = colour;

  while (!exhausted) {
    if (!have) {  // pull the next pixel
      const uint32_t item = atomicAdd(A.queue, 1u);
      if (item >= total_items) {
        exhausted = true;
      } else {
        const uint32_t tile = item >> 6, within = item & 63u;
        x = P.tile_x0 + (int)(tile % (uint32_t)tiles_x) * 8 + (int)(within & 7u);
        y = P.tile_y0 + (int)(tile / (uint32_t)tiles_x) * 8 + (int)(within >> 3);
        if (x < P.tile_x1 && y < P.tile_y1) {
          have = true;
          pixel = (uint32_t)y * (uint32_t)P.width + (uint32_t)x;
          s = P.sample_begin;
          alive = false;
          colour = mk3(0.0f, 0.0f, 0.0f);
          if (RNG == RT_RNG_XORSHIFT_REF) {
            const uint32_t seed = A.seed_tex ? A.seed_tex[pixel] : 0u;  // fs:105-106
            rng = seed ^ cantor_pair((uint32_t)x, (uint32_t)y);         // fs:107
          }
        }
      }
    }
    if (have) {
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
        closest_hit<SEARCH, COUNTED>(sc, tv, stack_col, o, d, t, idx, k);
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
        if (s >= P.sample_end) {  // the pixel is finished: one read-modify-write of its accumulator
          float* dst = A.accum + 3 * (size_t)pixel;
          dst[0] += colour.x;
          dst[1] += colour.y;
          dst[2] += colour.z;
          have = false;
        }
      }
    }
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
template <int SEARCH, bool COUNTED>
__global__ __launch_bounds__(search_traits<SEARCH>::threads) void rt_trace_kernel(const trace_args A) {
  using T = search_traits<SEARCH>;
  const scene_dev& sc = A.sc;
  tree_view tv;
  uint32_t* stack_col;
  search_setup<SEARCH>(sc, tv, stack_col);
  lane_counters k = {};
  const size_t i = (size_t)blockIdx.x * T::threads + threadIdx.x;
  if (i < A.n) {
    const f3 o = mk3(A.o_xyz[3 * i], A.o_xyz[3 * i + 1], A.o_xyz[3 * i + 2]);
    const f3 d = mk3(A.d_xyz[3 * i], A.d_xyz[3 * i + 1], A.d_xyz[3 * i + 2]);
    RT_COUNT(k, segments, 1);
    float t;
    int idx;
    closest_hit<SEARCH, COUNTED>(sc, tv, stack_col, o, d, t, idx, k);
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
