// rt_render_launch.hpp — what the translation units that launch render kernels share.
#pragma once

#include "rt_ctx.hpp"
#include "rt_render_kernels.cuh"

namespace rt_impl {

// rt_render_kernel for the uploaded scene's search (rt_render_pixels.hip) and rt_render_stored_kernel (rt_render_stored.hip;
// `owned`: its form whose lanes sum whole pixels in registers); both queue on the context's stream.
hipError_t launch_pixel_pass(rt_ctx* c, const rt_params* p, bool counted, const render_args& a);
hipError_t launch_stored_pass(rt_ctx* c, bool counted, bool owned, const render_args& a);
// Whether the uploaded scene's search leaves room in LDS for the reserves of rt_render_stored_kernel.
bool stored_fits(const rt_ctx* c);

// Workgroups of a render launch: as many as stay resident, and no more than the work can occupy.  A wave takes 64 work
// items at a time: the pixels of an 8x8 tile, or 64 jobs — a pixel's sweeps in a pass in sweeps, and in a main pass
// with a job list more jobs than resident lanes whatever the tile's size (the plan cuts the pixels until there are),
// so that a small tile still fills the device.
template <int WAVES>
long long blocks_for(const rt_ctx* c, int per_cu, int tw, int th, const render_args& a) {
  long long blocks = (long long)per_cu * c->n_cu;
  if (a.jobs == nullptr) {
    const long long tiles = (long long)((tw + 7) / 8) * ((th + 7) / 8) * (a.sweeps > 1 ? a.sweeps : 1);
    const long long needed = (tiles + WAVES - 1) / WAVES;
    if (blocks > needed) blocks = needed;
  }
  return blocks < 1 ? 1 : blocks;
}

}  // namespace rt_impl
