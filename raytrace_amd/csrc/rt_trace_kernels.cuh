// rt_trace_kernels.cuh — the batched closest-hit kernel (hit_any, testFragment.fs:165-199, on caller-given rays) and the
// resolve kernel (mean + gamma + RGB8, testFragment.fs:135-140).
#pragma once

#include "rt_kernels.cuh"

struct trace_args {
  scene_dev sc;
  const float* o_xyz;
  const float* d_xyz;
  size_t n;
  int32_t* id;
  float* t;
  float* nrm;
  unsigned long long* counters;
  unsigned long long* queue;  // next ray index, zero at launch
  int vote;
  int leaf_vote;
};

// hit_any (fs:165-199) for a batch of rays: persistent lanes pull ray indices
// from a queue, with the same phase voting as the renderer (a lane that has
// finished its ray fetches the next one while others keep traversing).
template <int SEARCH, bool COUNTED>
__global__ __launch_bounds__(search_traits<SEARCH>::threads, search_traits<SEARCH>::trace_waves_per_simd) void rt_trace_kernel(const trace_args A) {
  const scene_dev& sc = A.sc;
  scene_view sv;
  uint32_t* stack_col;
  uint32_t stage;  // this wave's node staging area (byte offset into LDS)
  search_setup<SEARCH>(sc, sv, stack_col, stage);
  lane_counters k = {};
  searcher<SEARCH, COUNTED> srch;
  srch.ref_mask = sc.ref_mask;
  bool pending = false, exhausted = false;
  srch.reset();
  size_t ray = 0;
  unsigned long long pool_next = 0, pool_end = 0;  // the wave's share of the ray queue

  while (true) {
    const unsigned long long m_live = __ballot(!exhausted);
    if (m_live == 0ull) break;
    while (search_traits<SEARCH>::tree) {
      const unsigned long long m_trav = __ballot(srch.active());
      if (m_trav == 0ull || __popcll(m_live & ~m_trav) >= A.vote) break;
      traversal_action(srch, sc, sv, stack_col, stage, A.leaf_vote, k);
    }
    RT_COUNT(k, path_wave_phases, (threadIdx.x & 63) == 0 ? 1u : 0u);
    const bool ready = !(srch.active() || exhausted);
    const unsigned long long next_ray = take_items<unsigned long long, 256u>(ready, A.queue, pool_next, pool_end);
    if (!ready) continue;
    RT_COUNT(k, path_lane_phases, 1);
    if (pending) {
      pending = false;
      float t;
      int idx;
      srch.finish(sc, sv, t, idx, k);
      const f3 o = srch.ts.o, d = srch.ts.d;
      f3 n = mk3(0.0f, 0.0f, 0.0f);
      int id = -1;
      float tt = -1.0f;
      if (t > 0.001f) {
        RT_COUNT(k, hits, 1);
        float4 sp;
        float inv_r;
        if (search_traits<SEARCH>::staged && srch.ts.best_pos >= 0) {  // the leaf's own record holds both (one request, not two)
          sp = sv.leaf_rec[2 * (size_t)srch.ts.best_pos];
          inv_r = sv.leaf_rec[2 * (size_t)srch.ts.best_pos + 1].y;
        } else {
          sp = sv.sph[idx];
          inv_r = sv.inv_r[idx];
        }
        const f3 p = o + d * t;
        n = inv_r * (p - mk3(sp.x, sp.y, sp.z));
        if (search_traits<SEARCH>::quads && sp.w == -__builtin_inff()) {
          const float4 qn = sc.quad_n[sc.quad_of[idx]];
          n = mk3(qn.x, qn.y, qn.z);
        }
        if (dot3(n, d) >= 0.0f) n = -n;
        id = idx;
        tt = t;
      }
      A.id[ray] = id;
      A.t[ray] = tt;
      A.nrm[3 * ray] = n.x;
      A.nrm[3 * ray + 1] = n.y;
      A.nrm[3 * ray + 2] = n.z;
    }
    ray = (size_t)next_ray;
    if (ray >= A.n) {
      exhausted = true;
    } else {
      const f3 o = mk3(A.o_xyz[3 * ray], A.o_xyz[3 * ray + 1], A.o_xyz[3 * ray + 2]);
      const f3 d = mk3(A.d_xyz[3 * ray], A.d_xyz[3 * ray + 1], A.d_xyz[3 * ray + 2]);
      RT_COUNT(k, segments, 1);
      srch.begin(sc, sv, o, d, k);
      pending = true;
    }
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
