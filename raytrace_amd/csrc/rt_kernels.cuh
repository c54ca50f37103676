// rt_kernels.cuh — what the render kernels (rt_render_kernels.cuh) and the closest-hit kernel (rt_trace_kernels.cuh)
// share: the kinds of search and their LDS budgets, the set-up of a workgroup's LDS, the search object, the work queue.
#pragma once

#include "rt_scene_dev.cuh"

// How a kernel finds the closest hit.
//   SEARCH_LINEAR      the reference's scan over every sphere, records by scalar loads
//   SEARCH_LINEAR_LDS  the same scan with the records staged in LDS (scenes up to 4096 spheres)
//   SEARCH_TREE_LDS_*  4-wide LBVH, hit-record and material arrays staged in LDS (small
//                      scenes): one copy per 1024-thread workgroup, one-word stack entries
//   SEARCH_TREE_GLB_*  4-wide LBVH read from HBM / caches (large scenes)
//   *_RAY / *_NODE     the box-growing rule (rt_scene_dev.cuh)
//   *_Q                the two searches of scenes that hold quads beside spheres: the scan, and the tree read from HBM
enum {
  SEARCH_LINEAR = 0,
  SEARCH_TREE_LDS_RAY = 1,
  SEARCH_TREE_LDS_NODE = 2,
  SEARCH_TREE_GLB_RAY = 3,
  SEARCH_TREE_GLB_NODE = 4,
  SEARCH_LINEAR_LDS = 5,
  SEARCH_LINEAR_Q = 6,
  SEARCH_TREE_GLB_NODE_Q = 7
};

#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 5
#endif
#ifndef RT_LDS_STACK
#define RT_LDS_STACK 12    // traversal-stack entries per lane for trees held in LDS (a deeper chain sends the ray to the scan)
#endif
#ifndef RT_JOB_CHUNK
#define RT_JOB_CHUNK 64u   // pixels / jobs a wave takes from the queue per atomic (more when more lanes ask at once)
#endif
#ifndef RT_LINEAR_WAVES
#define RT_LINEAR_WAVES 4
#endif

template <int SEARCH>
struct search_traits {
  static constexpr bool quads = SEARCH == SEARCH_LINEAR_Q || SEARCH == SEARCH_TREE_GLB_NODE_Q;
  static constexpr bool tree = (SEARCH >= SEARCH_TREE_LDS_RAY && SEARCH <= SEARCH_TREE_GLB_NODE) || SEARCH == SEARCH_TREE_GLB_NODE_Q;
  static constexpr bool lds = SEARCH == SEARCH_TREE_LDS_RAY || SEARCH == SEARCH_TREE_LDS_NODE;
  static constexpr bool scan_lds = SEARCH == SEARCH_LINEAR_LDS;
  static constexpr bool pad_ray = SEARCH == SEARCH_TREE_LDS_RAY || SEARCH == SEARCH_TREE_GLB_RAY;
  static constexpr int waves = lds ? 16 : 4;       // waves per workgroup
  static constexpr int threads = waves * 64;
  static constexpr int stack = lds ? RT_LDS_STACK : 8;  // entries per lane kept in LDS
  // per wave: one word per entry and lane; above the last row, three rows of slack for trees in LDS (a node step
  // stores three keys above the top whatever it pushes), one row for the others (the sink of stores that do not fit)
  // (trees read from HBM: above the sink row, the rows of the lane's leaf queue)
  static constexpr int stack_words = !tree ? 0 : (stack + (lds ? 3 : 1 + RT_LEAF_ROWS)) * 64;
  // trees read from HBM pass their nodes through a per-wave staging area in LDS (stage_nodes)
  static constexpr bool staged = tree && !lds;
  static constexpr int stage_bytes = staged ? RT_STAGE_BYTES : 0;
  // Waves per SIMD the register allocation of the render kernel aims for.
  static constexpr int render_waves_per_simd = tree ? 4 : RT_LINEAR_WAVES;
  // The closest-hit kernel on an HBM-resident tree waits on node fetches: more waves hide more of them.
  static constexpr int trace_waves_per_simd = (tree && !lds) ? RT_TRACE_WAVES : 4;
};

struct scene_counts {
  int n_nodes, n_leaf, n_spheres, n_materials;
};

inline size_t pad16(size_t b) { return (b + 15) & ~(size_t)15; }

// Bytes of the scene copy a search keeps in LDS.
template <int SEARCH>
inline size_t search_scene_lds_bytes(const scene_counts& n) {
  using T = search_traits<SEARCH>;
  if (T::scan_lds) return (size_t)n.n_spheres * 16;
  if (!T::lds) return 0;
  return (size_t)n.n_nodes * RT_LDS_NODE_STRIDE + (size_t)n.n_leaf * 16 + pad16((size_t)n.n_leaf * 4) +
         (size_t)n.n_spheres * 16 + 2 * pad16((size_t)n.n_spheres * 4) + (size_t)n.n_materials * 16 +
         pad16((size_t)n.n_materials * 4);
}

// Bytes of dynamic LDS a launch needs (scene copy + traversal stacks).
template <int SEARCH>
inline size_t search_lds_bytes(const scene_counts& n) {
  using T = search_traits<SEARCH>;
  return search_scene_lds_bytes<SEARCH>(n) + (size_t)T::waves * ((size_t)T::stack_words * 4 + T::stage_bytes);
}

// Rows of a call's share of its tile.  Without interleaving (strip_count 0) these are the tile's rows; with it, the
// rows of the share's strips of eight one after the other ("virtual" rows): per-share arrays are indexed by virtual
// row, and real_row gives the image row.
RT_DEV int real_row(int vy, int tile_y0, int strip_count, int strip_index) {
  if (strip_count == 0) return tile_y0 + vy;
  return tile_y0 + (((vy >> 3) * strip_count + strip_index) << 3) + (vy & 7);
}

template <bool COUNTED>
RT_DEV void flush_counters(unsigned long long* g, const lane_counters& k) {
  if (!COUNTED || g == nullptr) return;
  const uint32_t v[20] = {k.paths,           k.segments,        k.sphere_tests,     k.roots,           k.node_tests,
                          k.hits,            k.sky,             k.lambertian,       k.metallic,        k.dialectric,
                          k.perlin,          k.unit_tries,      k.disk_tries,       k.limit_hits,      k.trav_wave_steps,
                          k.trav_lane_steps, k.path_wave_phases, k.path_lane_phases, k.leaf_wave_steps, k.leaf_lane_steps};
#pragma unroll
  for (int i = 0; i < 20; i++) {
    // sum over the wave first: one atomic per wave and counter
    unsigned long long s = v[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0 && s != 0) atomicAdd(g + i, s);
  }
}


RT_DEV size_t pad16_dev(size_t b) { return (b + 15) & ~(size_t)15; }

// Carve the dynamic LDS — the scene copy of the search (tree, hit-record and
// material arrays; or the scan's sphere records), then the traversal stacks —
// and copy the scene in.  Every thread of the workgroup must call this before
// the first search.
template <int SEARCH>
RT_DEV void search_setup(const scene_dev& sc, scene_view& sv, uint32_t*& stack_col, uint32_t& stage) {
  using T = search_traits<SEARCH>;
  sv.nodes = sc.nodes;
  sv.qnodes = sc.qnodes;
  sv.leaf_sph = sc.leaf_sph;
  sv.leaf_idx = sc.leaf_idx;
  sv.leaf_rec = sc.leaf_rec;
  sv.scan_sph = sc.sph;
  sv.sph = sc.sph;
  sv.inv_r = sc.inv_r;
  sv.sph_mat = sc.sph_mat;
  sv.mat_a = sc.mat_a;
  sv.mat_type = sc.mat_type;
  sv.ovf_col = nullptr;
  sv.ovf_cap = 0;
  stack_col = nullptr;
  stage = 0u;
  unsigned char* base = rt_lds;
  if (T::scan_lds) {
    float4* d = reinterpret_cast<float4*>(base);
    for (int i = threadIdx.x; i < sc.n_spheres; i += T::threads) d[i] = sc.sph[i];
    sv.scan_sph = d;
    __syncthreads();
    return;
  }
  if (!T::tree) return;
  if (T::lds) {
    {
      float4* d = reinterpret_cast<float4*>(base);
      const float4* src = reinterpret_cast<const float4*>(sc.nodes);
      for (int i = threadIdx.x; i < sc.n_nodes * 8; i += T::threads) d[(i >> 3) * (RT_LDS_NODE_STRIDE / 16) + (i & 7)] = src[i];
      sv.nodes = reinterpret_cast<const bvh4_node*>(d);
      base += (size_t)sc.n_nodes * RT_LDS_NODE_STRIDE;
    }
#define RT_LDS_COPY(field, type, count)                                       \
  {                                                                           \
    type* d = reinterpret_cast<type*>(base);                                  \
    for (int i = threadIdx.x; i < (count); i += T::threads) d[i] = sc.field[i]; \
    sv.field = d;                                                             \
    base += pad16_dev((size_t)(count) * sizeof(type));                        \
  }
    RT_LDS_COPY(leaf_sph, float4, sc.n_leaf)
    RT_LDS_COPY(leaf_idx, int32_t, sc.n_leaf)
    RT_LDS_COPY(sph, float4, sc.n_spheres)
    RT_LDS_COPY(inv_r, float, sc.n_spheres)
    RT_LDS_COPY(sph_mat, int32_t, sc.n_spheres)
    RT_LDS_COPY(mat_a, float4, sc.n_materials)
    RT_LDS_COPY(mat_type, int32_t, sc.n_materials)
#undef RT_LDS_COPY
    __syncthreads();
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  stack_col = reinterpret_cast<uint32_t*>(base) + wave * T::stack_words + lane;
  // (no scene copy precedes the stacks; the offset is the same for the whole wave, and saying so keeps the LDS base of
  // the staging loads in a scalar register)
  if (T::staged) stage = (uint32_t)__builtin_amdgcn_readfirstlane((int)(T::waves * T::stack_words * 4 + wave * T::stage_bytes));
  if (!T::lds && sc.ovf != nullptr) {
    sv.ovf_col = sc.ovf + ((size_t)blockIdx.x * T::waves + wave) * (size_t)sc.ovf_cap * 64 + lane;
    sv.ovf_cap = sc.ovf_cap;
  }
}

// The search object of a kernel: the stepwise tree search, or for the linear
// scan an object whose finish() does the whole scan.
template <int SEARCH, bool COUNTED>
struct searcher {
  using T = search_traits<SEARCH>;
  tree_search<T::pad_ray, T::lds, T::stack, COUNTED, T::quads> ts;
  uint32_t ref_mask;  // wave-uniform (scene_dev::ref_mask), set by the kernel before the first step
  RT_DEV bool begin(const scene_dev& sc, const scene_view& sv, f3 o, f3 d, lane_counters& k) {
    if (T::tree) return ts.begin(sc, sv, o, d, k);
    ts.o = o;
    ts.d = d;
    ts.cur = -1;
    ts.c.give_up();  // finish() runs the scan
    return false;
  }
  static constexpr bool leaf_steps = T::tree && (!T::lds || RT_LDS_LEAF_STEPS);  // leaves wait for a step of their own
  RT_DEV void reset() {  // no search under way
    ts.cur = -1;
    if (!ts.LEAF_INLINE) ts.lq.clear();
  }
  RT_DEV bool node_ready() const { return T::tree && ts.node_ready(); }
  RT_DEV bool has_leaf() const { return T::tree && ts.has_leaf(); }
  RT_DEV bool active() const { return T::tree && ts.active(); }
  RT_DEV void step_node(const scene_view& sv, uint32_t* stack_col, uint32_t staged, lane_counters& k) {
    trav_stack<T::lds> stk;
    stk.col = stack_col;
    stk.mask = T::lds ? 0xFFFu : ref_mask;
    ts.step_node(sv, stk, staged, k);
  }
  RT_DEV int node_wanted() const { return ts.cur; }
  RT_DEV void step_leaf(const scene_dev& sc, const scene_view& sv, const uint32_t* stack_col, lane_counters& k) {
    ts.step_leaf(sc, sv, stack_col + (T::stack + 1) * 64, k);
  }
  RT_DEV void finish(const scene_dev& sc, const scene_view& sv, float& t, int& idx, lane_counters& k) {
    if (!T::tree) {  // the scan is the search
      hit_linear<COUNTED>(sv.scan_sph, sc.n_spheres, ts.o, ts.d, t, idx, k);
      if (T::quads) hit_quads_linear(sc, ts.o, ts.d, t, idx);
      return;
    }
    ts.finish(sc, sv, t, idx, k);
  }
};

// Lanes of a wave that are ready for their path phase wait until this many of
// them are (or until no lane is traversing); until then the wave keeps taking
// traversal steps for the others.
#define RT_PHASE_VOTE 52
#define RT_TRACE_VOTE 8
#define RT_LEAF_VOTE 8

// One traversal action of a wave, for the lanes whose search is under way: a node step, or — trees read from HBM,
// whose leaves wait in the lane's queue — a leaf step (one leaf per lane) once `leaf_vote` lanes hold one or no lane
// can take a node step.  Every lane of the wave must call this together.
template <int SEARCH, bool COUNTED>
RT_DEV void traversal_action(searcher<SEARCH, COUNTED>& srch, const scene_dev& sc, const scene_view& sv, uint32_t* stack_col,
                             uint32_t stage, int leaf_vote, lane_counters& k) {
  using T = search_traits<SEARCH>;
  const bool node_lane = srch.node_ready(), leaf_lane = srch.has_leaf();
  if (srch.leaf_steps && (__ballot(node_lane) == 0ull || __popcll(__ballot(leaf_lane)) >= leaf_vote)) {
    RT_COUNT(k, leaf_wave_steps, (threadIdx.x & 63) == 0 ? 1u : 0u);
    RT_COUNT(k, leaf_lane_steps, leaf_lane ? 1u : 0u);
    if (leaf_lane) srch.step_leaf(sc, sv, stack_col, k);
    return;
  }
  RT_COUNT(k, trav_wave_steps, (threadIdx.x & 63) == 0 ? 1u : 0u);
  RT_COUNT(k, trav_lane_steps, node_lane ? 1u : 0u);
  uint32_t staged = 0u;  // trees read from HBM: the whole wave fetches the nodes its lanes want
  if (T::staged) staged = stage_nodes(sv.qnodes, node_lane ? srch.node_wanted() : -1, stage);
  if (node_lane) srch.step_node(sv, stack_col, staged, k);
}

// Work items (pixels, rays) come from one global counter.  Same-address atomics
// from the whole chip complete at roughly 80 million per second, so a wave takes
// a chunk per atomic and hands the items of its chunk to its lanes by ballot
// rank.  `need` marks the lanes that want an item; every lane of the wave must
// call this together.  Returns the lane's item (any value for lanes that did
// not ask); items at or beyond the total mean the queue is dry.
template <typename Q, uint32_t CHUNK>
RT_DEV Q take_items(bool need, Q* queue, Q& pool_next, Q& pool_end) {
  const unsigned long long m_need = __ballot(need);
  Q item = 0;
  if (m_need != 0ull) {
    const uint32_t n_need = (uint32_t)__popcll(m_need);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_need, 0u));
    const Q avail = pool_end - pool_next;
    if (avail < (Q)n_need) {
      // a chunk, or as much as the lanes ask for beyond what is left of the old one if that is more
      const Q amount = (Q)n_need - avail > (Q)CHUNK ? (Q)n_need - avail : (Q)CHUNK;
      Q fresh = 0;
      if ((threadIdx.x & 63) == 0) fresh = atomicAdd(queue, amount);
      fresh = sizeof(Q) == 8 ? (Q)(((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)((unsigned long long)fresh >> 32)) << 32) |
                                   __builtin_amdgcn_readfirstlane((uint32_t)fresh))
                             : (Q)__builtin_amdgcn_readfirstlane((uint32_t)fresh);
      item = (Q)rank < avail ? pool_next + rank : fresh + ((Q)rank - avail);
      pool_next = fresh + ((Q)n_need - avail);
      pool_end = fresh + amount;
    } else {
      item = pool_next + rank;
      pool_next += (Q)n_need;
    }
  }
  return item;
}

// Tuning builds (-DRT_SECTION_CLOCKS) time the sections of the render loop with the
// shader clock, per wave, and the counted launch reports the sums in place of the
// hits .. unit_tries counters: traversal step, hit branch, sky branch, sample / pixel
// turnover, camera ray, segment set-up, whole loop.
#ifdef RT_SECTION_CLOCKS
#define RT_CLK_DECL unsigned long long clk_[7] = {0, 0, 0, 0, 0, 0, 0}; unsigned long long clk_t_ = 0, clk_0_ = __builtin_amdgcn_s_memtime();
#define RT_CLK_START() clk_t_ = __builtin_amdgcn_s_memtime()
#define RT_CLK_STOP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); clk_[i] += n_ - clk_t_; clk_t_ = n_; } while (0)
#else
#define RT_CLK_DECL
#define RT_CLK_START() do {} while (0)
#define RT_CLK_STOP(i) do {} while (0)
#endif

