// rt_ctx.hpp — what the translation units of the library share: error reporting, device buffers, the context, the
// tuning switches, and the launch helpers that depend on the kind of search.
//   rt_api.hip     contexts, accumulator, readback, closest-hit batches, queries
//   rt_build.hip   scene upload and flattening, the two hierarchy builders
//   rt_render.hip  the frame scheduler and the render launches
//   rt_comm.hip    the sum across ranks (RCCL, loaded on first use)
//   rt_eval.hip    device functions on caller-given values, the seed image
// There is no CPU fallback anywhere: without a HIP device every compute entry point fails with RT_ERR_NO_GPU.
#pragma once

#include <string.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/raytrace_amd.h"
#include "rt_kernels.cuh"

namespace rt_impl {

// Records the message rt_last_error() returns on this thread; returns `code`.  (rt_api.hip)
int fail(int code, const char* fmt, ...);

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) return fail(RT_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

template <typename T>
struct dev_buf {
  T* p = nullptr;
  size_t n = 0;
  hipError_t resize(size_t count) {
    if (count <= n && p) return hipSuccess;
    release();
    if (count == 0) return hipSuccess;
    hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
    if (e == hipSuccess) n = count; else p = nullptr;
    return e;
  }
  // resize() for the large per-frame buffers: one held from a much larger frame is given back first, and running out
  // of memory is reported without leaving the error pending
  hipError_t fit(size_t count) {
    if (p && n > 4 * count && n * sizeof(T) > ((size_t)64 << 20)) release();
    const hipError_t e = resize(count);
    if (e == hipErrorOutOfMemory) (void)hipGetLastError();
    return e;
  }
  hipError_t upload(const std::vector<T>& v, hipStream_t st) {
    hipError_t e = resize(v.empty() ? 1 : v.size());
    if (e != hipSuccess || v.empty()) return e;
    return hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st);
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};

// Memory for the per-sample radiance of pixels whose main pass is cut into several jobs.
constexpr size_t kSplitBudgetBytes = (size_t)32 << 30;
// The same for the probe pass in sweeps (every pixel's probe samples): beyond this the probe runs unsplit.
constexpr size_t kProbeBudgetBytes = (size_t)4 << 30;
// A stored sample is one aligned 16-byte record (r, g, b, unused): one store request per lane, whole 64-byte lines per
// group of four neighbouring rows, where three separate words per sample were written as partial lines (measured:
// 3.1 bytes written per byte of payload).
constexpr size_t kSampleFloats = 4;
inline size_t stored_samples(int n) { return ((size_t)n + 1) & ~(size_t)1; }  // (whole pairs of records: pair_slot)
// Largest dynamic LDS a launch may ask for (the CU has 160 KiB).
constexpr size_t kMaxLdsBytes = 160 * 1024;
// The linear scan stages its sphere records in LDS up to this many bytes.
constexpr size_t kLdsTreeBudget = 64 * 1024;
// Spheres with a radius above kBigFactor x the median radius stay out of the
// tree (at most kMaxBig of them) and are tested exactly for every ray.
// The noise texture's gradient table covers the lattice points [-kPerlinHalf, kPerlinHalf)^2 (134 MB).
constexpr int kPerlinHalf = 2048;
constexpr float kBigFactor = 8.0f;
constexpr int kMaxBig = 8;

constexpr int kMaxDevices = 64;

// Tuning switches (RT_PHASE_VOTE, RT_BUILDER, RT_SCAN_UP_TO, ...) are read from the environment only when RT_TUNING=1
// is set there as well: a variable that happens to be set in a caller's environment must not change what the library
// does.  With RT_TUNING=1 they are for measurements and tests; no switch changes a result.
inline const char* tuning_env(const char* name) {
  const char* on = getenv("RT_TUNING");
  if (!on || on[0] != '1' || on[1] != 0) return nullptr;
  return getenv(name);
}

}  // namespace rt_impl
using namespace rt_impl;

// Work arrays of the hierarchy build.  They stay with the context between uploads (a scene that is
// rebuilt every frame should not pay fifteen allocations per build) unless they are large.
struct lbvh_scratch {
  dev_buf<uint32_t> bounds, codes, codes_sorted, index, index_sorted;
  dev_buf<int> child0, child1, node_parent, leaf_parent, scalars, slot_of;
  dev_buf<unsigned int> arrivals;
  dev_buf<float> node_box, node_rmin;
  dev_buf<unsigned char> temp;
  void release() {
    bounds.release(); codes.release(); codes_sorted.release(); index.release(); index_sorted.release();
    child0.release(); child1.release(); node_parent.release(); leaf_parent.release(); scalars.release();
    slot_of.release(); arrivals.release(); node_box.release(); node_rmin.release(); temp.release();
  }
};
constexpr size_t kKeepScratchSpheres = (size_t)1 << 18;  // builds up to this size keep their work arrays

struct rt_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr, ev_b0 = nullptr, ev_b1 = nullptr;
  bool have_kernel_time = false, have_build_time = false;
  const char* last_kernel = "";

  // scene (upload order)
  bool have_scene = false;
  int n_spheres = 0, n_materials = 0, accel = RT_ACCEL_LINEAR;
  dev_buf<float4> sph;
  dev_buf<float> inv_r;
  dev_buf<int32_t> sph_mat, mat_type;
  dev_buf<float4> mat_a;
  void* comm = nullptr;  // the RCCL communicator of rt_comm_init
  int comm_rank = 0, comm_world = 0;
  int n_quads = 0;
  dev_buf<float4> quad_qd, quad_u, quad_v, quad_n, quad_w;
  dev_buf<int32_t> quad_idx, quad_of;
  dev_buf<float2> perlin_grad;  // gradient table of the noise texture, built with the first scene that uses it
  bool have_noise = false;
  // hierarchy
  int search = SEARCH_LINEAR;
  int n_big = 0, n_leaf = 0, n_nodes = 0, root = 0, depth = 0, builder = RT_BUILDER_NONE;
  bool scanned_for_size = false;  // a tree was asked for, but the scene is small enough for the scan to be faster
  dev_buf<float4> big_sph;
  dev_buf<int32_t> big_idx;
  dev_buf<float4> tree_sph;   // spheres of the tree, upload order among themselves
  dev_buf<float> tree_r;
  dev_buf<int32_t> tree_orig; // -> upload index
  dev_buf<bvh4_node> nodes;
  dev_buf<bvh4q_node> qnodes;  // the 64-byte form of the nodes, for trees the kernels read from HBM
  dev_buf<float4> leaf_sph;
  dev_buf<float> leaf_r;
  dev_buf<int32_t> leaf_idx;
  dev_buf<float4> leaf_rec;   // trees read from HBM: two float4 per leaf (scene_dev::leaf_rec)
  float root_lo[3] = {0, 0, 0}, root_hi[3] = {0, 0, 0};
  float r_max = 0.0f, r_min = 0.0f, r_median = 0.0f, abs_max = 0.0f;

  // accumulator
  dev_buf<float> accum_own;
  float* accum_ext = nullptr;
  int acc_w = 0, acc_h = 0;
  int samples_lo = 0, samples_hi = 0, samples_override = -1;

  dev_buf<uint32_t> seed_tex;
  dev_buf<unsigned long long> queue;  // work-queue head of the running launch
  lbvh_scratch build;
  // rt_trace_batch (host buffers): two sets of device chunk buffers, copy streams and events, made on first use
  hipStream_t copy_in = nullptr, copy_out = nullptr;
  hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_run[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
  dev_buf<float> tb_o[2], tb_d[2], tb_t[2], tb_n[2];
  dev_buf<int32_t> tb_id[2];
  dev_buf<uint32_t> ovf;              // traversal-stack overflow of the running launch (HBM-resident trees)
  // two-pass rendering: per-pixel carry of the probe pass and the sorted pixel order
  dev_buf<float> carry_colour;
  dev_buf<uint32_t> carry_rng, carry_cost, carry_key, carry_idx, order_key, order_idx;
  dev_buf<unsigned char> sort_temp;
  // main-pass jobs: parts per sorted pixel, their prefix sum, the job table, totals, per-sample store of split pixels
  dev_buf<uint32_t> plan_parts, plan_first, plan_counts;
  dev_buf<unsigned long long> plan_total;
  dev_buf<uint4> plan_jobs;
  dev_buf<float> split_samples, probe_samples;
  dev_buf<unsigned char> scan_temp;
  int n_cu = 0;
  size_t split_budget = 0, probe_budget = 0;  // rt_set_scratch_budget; 0: the defaults
  dev_buf<unsigned long long> counters;
  dev_buf<float> out_linear;
  dev_buf<uint8_t> out_rgb8;

  float* accum() { return accum_ext ? accum_ext : accum_own.p; }
  scene_dev view() const {
    scene_dev s;
    s.n_spheres = n_spheres;
    s.n_materials = n_materials;
    s.sph = sph.p;
    s.inv_r = inv_r.p;
    s.sph_mat = sph_mat.p;
    s.mat_a = mat_a.p;
    s.mat_type = mat_type.p;
    s.noise.grad = have_noise ? perlin_grad.p : nullptr;
    s.noise.half = kPerlinHalf;
    s.n_quads = n_quads;
    s.quad_qd = quad_qd.p;
    s.quad_u = quad_u.p;
    s.quad_v = quad_v.p;
    s.quad_n = quad_n.p;
    s.quad_w = quad_w.p;
    s.quad_idx = quad_idx.p;
    s.quad_of = quad_of.p;
    s.n_big = n_big;
    s.big_sph = big_sph.p;
    s.big_idx = big_idx.p;
    s.n_leaf = n_leaf;
    s.n_nodes = n_nodes;
    s.nodes = nodes.p;
    s.qnodes = qnodes.p;
    s.leaf_sph = leaf_sph.p;
    s.leaf_idx = leaf_idx.p;
    s.leaf_rec = leaf_rec.p;
    s.root = root;
    for (int a = 0; a < 3; a++) {
      s.root_lo[a] = root_lo[a];
      s.root_hi[a] = root_hi[a];
    }
    s.r_max = r_max;
    s.r_min = r_min;
    s.ref_mask = 0xFFFu;
    while (s.ref_mask + 1u < (uint32_t)(n_nodes > 0 ? n_nodes : 1)) s.ref_mask = s.ref_mask * 2u + 1u;
    s.abs_max = abs_max;
    s.ovf = nullptr;
    s.ovf_cap = 0;
    return s;
  }
};


namespace rt_impl {

inline int check_ctx(rt_ctx* c) {
  if (!c) return fail(RT_ERR_INVALID, "null context");
  HIP_TRY(hipSetDevice(c->device));
  return RT_OK;
}

// Which search the kernels use for the uploaded scene.
inline scene_counts counts_of(const rt_ctx* c) {
  scene_counts n;
  n.n_nodes = c->n_nodes;
  n.n_leaf = c->n_leaf;
  n.n_spheres = c->n_spheres;
  n.n_materials = c->n_materials;
  return n;
}

// The gradient table of the noise texture (built once per context).  (rt_eval.hip)
int ensure_noise_table(rt_ctx* c);

inline const char* stored_name(int search) {
  switch (search) {
    case SEARCH_LINEAR: return "rt_render_stored_kernel<linear>";
    case SEARCH_LINEAR_LDS: return "rt_render_stored_kernel<linear,lds>";
    case SEARCH_TREE_LDS_RAY: return "rt_render_stored_kernel<lbvh4,lds,pad_ray>";
    case SEARCH_TREE_LDS_NODE: return "rt_render_stored_kernel<lbvh4,lds,pad_node>";
    case SEARCH_TREE_GLB_RAY: return "rt_render_stored_kernel<lbvh4,hbm,pad_ray>";
    case SEARCH_LINEAR_Q: return "rt_render_stored_kernel<linear,quads>";
    case SEARCH_TREE_GLB_NODE_Q: return "rt_render_stored_kernel<lbvh4,hbm,pad_node,quads>";
    default: return "rt_render_stored_kernel<lbvh4,hbm,pad_node>";
  }
}

inline const char* search_name(int search, bool render) {
  switch (search) {
    case SEARCH_LINEAR: return render ? "rt_render_kernel<linear>" : "rt_trace_kernel<linear>";
    case SEARCH_LINEAR_LDS: return render ? "rt_render_kernel<linear,lds>" : "rt_trace_kernel<linear,lds>";
    case SEARCH_TREE_LDS_RAY: return render ? "rt_render_kernel<lbvh4,lds,pad_ray>" : "rt_trace_kernel<lbvh4,lds,pad_ray>";
    case SEARCH_TREE_LDS_NODE: return render ? "rt_render_kernel<lbvh4,lds,pad_node>" : "rt_trace_kernel<lbvh4,lds,pad_node>";
    case SEARCH_TREE_GLB_RAY: return render ? "rt_render_kernel<lbvh4,hbm,pad_ray>" : "rt_trace_kernel<lbvh4,hbm,pad_ray>";
    case SEARCH_LINEAR_Q: return render ? "rt_render_kernel<linear,quads>" : "rt_trace_kernel<linear,quads>";
    case SEARCH_TREE_GLB_NODE_Q: return render ? "rt_render_kernel<lbvh4,hbm,pad_node,quads>" : "rt_trace_kernel<lbvh4,hbm,pad_node,quads>";
    default: return render ? "rt_render_kernel<lbvh4,hbm,pad_node>" : "rt_trace_kernel<lbvh4,hbm,pad_node>";
  }
}

// Ready lanes a wave waits for before it interrupts traversal for the other
// phase: the path phase of the render kernels (long, so worth waiting for a
// nearly full wave) and the fetch / store phase of the closest-hit kernel
// (short).  RT_PHASE_VOTE / RT_TRACE_VOTE in the environment override the
// defaults, for tuning.
inline int vote_from_env(const char* name, int fallback) {
  const char* e = tuning_env(name);
  int x = e ? atoi(e) : fallback;
  return x < 1 ? 1 : (x > 64 ? 64 : x);
}
inline int phase_vote() {
  static int v = vote_from_env("RT_PHASE_VOTE", RT_PHASE_VOTE);
  return v;
}
inline int trace_vote() {
  static int v = vote_from_env("RT_TRACE_VOTE", RT_TRACE_VOTE);
  return v;
}
// Lanes with queued leaves a wave waits for before it takes a leaf step.
inline int leaf_vote() {
  static int v = vote_from_env("RT_LEAF_VOTE", RT_LEAF_VOTE);
  return v;
}


template <typename K>
hipError_t allow_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// Persistent launch: as many workgroups as the chip keeps resident (fewer when
// the image has fewer 8x8 tiles than that).
// Stack entries a lane of an HBM-resident tree may need beyond its LDS part:
// a node step pushes at most three children and the 4-wide tree has
// depth / 2 + 1 levels.
template <int SEARCH>
hipError_t attach_overflow(rt_ctx* c, long long blocks, scene_dev& sc) {
  using T = search_traits<SEARCH>;
  if (!T::tree || T::lds) return hipSuccess;
  const int cap = 3 * (c->depth / 2 + 2) - T::stack;
  if (cap <= 0) return hipSuccess;
  hipError_t e = c->ovf.resize((size_t)blocks * T::waves * (size_t)cap * 64);
  if (e != hipSuccess) return e;
  sc.ovf = c->ovf.p;
  sc.ovf_cap = cap;
  return hipSuccess;
}

// The same from a fixed piece of memory: the launch is cut down to the workgroups whose slices fit (the kernel is
// persistent, any number of workgroups drains the queue).  Returns false when not even one workgroup per CU fits.
template <int SEARCH>
bool attach_overflow_fixed(rt_ctx* c, long long& blocks, scene_dev& sc, unsigned char* base, size_t bytes) {
  using T = search_traits<SEARCH>;
  if (!T::tree || T::lds) return true;
  const int cap = 3 * (c->depth / 2 + 2) - T::stack;
  if (cap <= 0) return true;
  const size_t per_block = (size_t)T::waves * (size_t)cap * 64 * sizeof(uint32_t);
  const long long fit = (long long)(bytes / per_block);
  if (fit < (long long)c->n_cu && fit < blocks) return false;
  if (blocks > fit) blocks = fit;
  sc.ovf = reinterpret_cast<uint32_t*>(base);
  sc.ovf_cap = cap;
  return true;
}


}  // namespace rt_impl
