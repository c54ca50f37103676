// rt_render_pixels.hip — launches of rt_render_kernel, instantiated for every kind of search, both generators, with and
// without work counters and the noise material.

#include "rt_render_launch.hpp"

namespace {

template <int SEARCH, int RNG, bool COUNTED, bool NOISE>
hipError_t launch_render_noise(rt_ctx* c, const rt_params* p, hipStream_t st, render_args a) {
  using T = search_traits<SEARCH>;
  const int tw = p->tile_x1 - p->tile_x0, th = a.vrows;
  const size_t lds = search_lds_bytes<SEARCH>(counts_of(c));
  hipError_t e = allow_lds(rt_render_kernel<SEARCH, RNG, COUNTED, NOISE>, lds);
  if (e != hipSuccess) return e;
  int per_cu = 0;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rt_render_kernel<SEARCH, RNG, COUNTED, NOISE>, T::threads, lds);
  if (e != hipSuccess) return e;
  if (per_cu < 1) per_cu = 1;
  const long long blocks = blocks_for<T::waves>(c, per_cu, tw, th, a);
  e = attach_overflow<SEARCH>(c, blocks, a.sc);
  if (e != hipSuccess) return e;
  rt_render_kernel<SEARCH, RNG, COUNTED, NOISE><<<dim3((unsigned)blocks), T::threads, lds, st>>>(a);
  return hipGetLastError();
}

// (the kernels come in two forms, for scenes with and without the noise-textured material: see shade_hit)
template <int SEARCH, int RNG, bool COUNTED>
hipError_t launch_render_one(rt_ctx* c, const rt_params* p, hipStream_t st, render_args a) {
  return c->have_noise ? launch_render_noise<SEARCH, RNG, COUNTED, true>(c, p, st, a) : launch_render_noise<SEARCH, RNG, COUNTED, false>(c, p, st, a);
}

template <int SEARCH>
hipError_t launch_render(rt_ctx* c, const rt_params* p, bool counted, hipStream_t st, const render_args& a) {
  if (p->rng_mode == RT_RNG_XORSHIFT_REF)
    return counted ? launch_render_one<SEARCH, RT_RNG_XORSHIFT_REF, true>(c, p, st, a)
                   : launch_render_one<SEARCH, RT_RNG_XORSHIFT_REF, false>(c, p, st, a);
  return counted ? launch_render_one<SEARCH, RT_RNG_COUNTER, true>(c, p, st, a)
                 : launch_render_one<SEARCH, RT_RNG_COUNTER, false>(c, p, st, a);
}

}  // namespace

namespace rt_impl {

hipError_t launch_pixel_pass(rt_ctx* c, const rt_params* rp, bool counted, const render_args& ra) {
  switch (c->search) {
    case SEARCH_TREE_LDS_RAY: return launch_render<SEARCH_TREE_LDS_RAY>(c, rp, counted, c->stream, ra);
    case SEARCH_TREE_LDS_NODE: return launch_render<SEARCH_TREE_LDS_NODE>(c, rp, counted, c->stream, ra);
    case SEARCH_TREE_GLB_RAY: return launch_render<SEARCH_TREE_GLB_RAY>(c, rp, counted, c->stream, ra);
    case SEARCH_TREE_GLB_NODE: return launch_render<SEARCH_TREE_GLB_NODE>(c, rp, counted, c->stream, ra);
    case SEARCH_TREE_GLB_NODE_Q: return launch_render<SEARCH_TREE_GLB_NODE_Q>(c, rp, counted, c->stream, ra);
    case SEARCH_LINEAR_Q: return launch_render<SEARCH_LINEAR_Q>(c, rp, counted, c->stream, ra);
    case SEARCH_LINEAR_LDS: return launch_render<SEARCH_LINEAR_LDS>(c, rp, counted, c->stream, ra);
    default: return launch_render<SEARCH_LINEAR>(c, rp, counted, c->stream, ra);
  }
}

}  // namespace rt_impl
