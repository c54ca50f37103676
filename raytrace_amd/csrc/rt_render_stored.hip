// rt_render_stored.hip — launches of rt_render_stored_kernel, instantiated for every kind of search, with and without
// work counters, the noise material and lanes that sum whole pixels.

#include "rt_render_launch.hpp"

namespace {

// rt_render_stored_kernel: the search's LDS plus one camera ray in reserve per lane.
template <int SEARCH>
size_t stored_lds_bytes(const rt_ctx* c) {
  return pad16(search_lds_bytes<SEARCH>(counts_of(c))) + (size_t)search_traits<SEARCH>::waves * RT_RESERVE_BYTES;
}

template <int SEARCH, bool COUNTED, bool NOISE, bool OWNED>
hipError_t launch_stored_noise(rt_ctx* c, hipStream_t st, render_args a) {
  using T = search_traits<SEARCH>;
  const int tw = a.p.tile_x1 - a.p.tile_x0, th = a.vrows;
  const size_t lds = stored_lds_bytes<SEARCH>(c);
  a.reserve_offset = (uint32_t)(lds - (size_t)T::waves * RT_RESERVE_BYTES);
  hipError_t e = allow_lds(rt_render_stored_kernel<SEARCH, COUNTED, NOISE, OWNED>, lds);
  if (e != hipSuccess) return e;
  int per_cu = 0;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rt_render_stored_kernel<SEARCH, COUNTED, NOISE, OWNED>, T::threads, lds);
  if (e != hipSuccess) return e;
  if (per_cu < 1) per_cu = 1;
  const long long blocks = blocks_for<T::waves>(c, per_cu, tw, th, a);
  e = attach_overflow<SEARCH>(c, blocks, a.sc);
  if (e != hipSuccess) return e;
  rt_render_stored_kernel<SEARCH, COUNTED, NOISE, OWNED><<<dim3((unsigned)blocks), T::threads, lds, st>>>(a);
  return hipGetLastError();
}

template <int SEARCH, bool COUNTED, bool OWNED>
hipError_t launch_stored_one(rt_ctx* c, hipStream_t st, render_args a) {
  return c->have_noise ? launch_stored_noise<SEARCH, COUNTED, true, OWNED>(c, st, a) : launch_stored_noise<SEARCH, COUNTED, false, OWNED>(c, st, a);
}

// owned: the form of the kernel whose lanes sum whole pixels in registers (a main pass with jobs that have no row in the store)
template <int SEARCH>
hipError_t launch_stored(rt_ctx* c, bool counted, bool owned, hipStream_t st, const render_args& a) {
  if (owned) return counted ? launch_stored_one<SEARCH, true, true>(c, st, a) : launch_stored_one<SEARCH, false, true>(c, st, a);
  return counted ? launch_stored_one<SEARCH, true, false>(c, st, a) : launch_stored_one<SEARCH, false, false>(c, st, a);
}

}  // namespace

namespace rt_impl {

bool stored_fits(const rt_ctx* c) {
  switch (c->search) {
    case SEARCH_TREE_LDS_RAY: return stored_lds_bytes<SEARCH_TREE_LDS_RAY>(c) <= kMaxLdsBytes;
    case SEARCH_TREE_LDS_NODE: return stored_lds_bytes<SEARCH_TREE_LDS_NODE>(c) <= kMaxLdsBytes;
    case SEARCH_TREE_GLB_RAY: return stored_lds_bytes<SEARCH_TREE_GLB_RAY>(c) <= kMaxLdsBytes;
    case SEARCH_TREE_GLB_NODE: return stored_lds_bytes<SEARCH_TREE_GLB_NODE>(c) <= kMaxLdsBytes;
    case SEARCH_TREE_GLB_NODE_Q: return stored_lds_bytes<SEARCH_TREE_GLB_NODE_Q>(c) <= kMaxLdsBytes;
    case SEARCH_LINEAR_LDS: return stored_lds_bytes<SEARCH_LINEAR_LDS>(c) <= kMaxLdsBytes;
    default: return true;
  }
}

hipError_t launch_stored_pass(rt_ctx* c, bool counted, bool owned, const render_args& ra) {
  switch (c->search) {
    case SEARCH_TREE_LDS_RAY: return launch_stored<SEARCH_TREE_LDS_RAY>(c, counted, owned, c->stream, ra);
    case SEARCH_TREE_LDS_NODE: return launch_stored<SEARCH_TREE_LDS_NODE>(c, counted, owned, c->stream, ra);
    case SEARCH_TREE_GLB_RAY: return launch_stored<SEARCH_TREE_GLB_RAY>(c, counted, owned, c->stream, ra);
    case SEARCH_TREE_GLB_NODE: return launch_stored<SEARCH_TREE_GLB_NODE>(c, counted, owned, c->stream, ra);
    case SEARCH_TREE_GLB_NODE_Q: return launch_stored<SEARCH_TREE_GLB_NODE_Q>(c, counted, owned, c->stream, ra);
    case SEARCH_LINEAR_Q: return launch_stored<SEARCH_LINEAR_Q>(c, counted, owned, c->stream, ra);
    case SEARCH_LINEAR_LDS: return launch_stored<SEARCH_LINEAR_LDS>(c, counted, owned, c->stream, ra);
    default: return launch_stored<SEARCH_LINEAR>(c, counted, owned, c->stream, ra);
  }
}

}  // namespace rt_impl
