// rt_api.hip — the C ABI of raytrace_amd.h: contexts, scene upload and
// flattening, LBVH construction, render / trace launches, readback.
// There is no CPU fallback anywhere in this file: without a HIP device every
// compute entry point fails with RT_ERR_NO_GPU.

#include <string.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <dlfcn.h>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is loaded on first use (rccl_api)
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "../../include/raytrace_amd.h"
#include "rt_kernels.cuh"
#include "rt_seedtex.cuh"
#include "rt_lbvh.cuh"

namespace {

thread_local std::string g_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}

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

}  // namespace

// ---- fixed scratch of rt_trace_batch_noalloc ---------------------------------
// The closest-hit kernel needs device memory beyond its inputs and outputs: the head of its ray queue and, for trees
// read from HBM, the part of the lanes' traversal stacks that does not fit LDS.  rt_trace_batch_device takes both from
// buffers the context allocates on demand; the entry point that must not allocate takes them from this array, which
// every GPU the library runs on holds (at its own address) until the process ends.
//   [0, kNoAllocHeaderBytes)       queue head (8 bytes used)
//   [kNoAllocHeaderBytes, end)     traversal-stack overflow, one slice per workgroup of the launch
static __device__ __align__(256) unsigned char rt_noalloc_scratch[RT_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr size_t kNoAllocHeaderBytes = 256;
constexpr int kMaxDevices = 64;
// Device address of rt_noalloc_scratch on each device, looked up by rt_init with that device current (the array's
// name is only a host-side handle: kernels and memsets get these addresses).
unsigned char* g_noalloc_scratch[kMaxDevices] = {};
}  // namespace

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

namespace {

// Builds the 4-wide LBVH over tree_sph / tree_r / tree_orig (n_leaf spheres).
int build_lbvh(rt_ctx* c) {
  const int n = c->n_leaf;
  c->depth = 0;
  c->n_nodes = 0;
  c->root = n == 1 ? ~0 : 0;
  if (n == 0) return RT_OK;
  HIP_TRY(c->leaf_sph.resize((size_t)n));
  HIP_TRY(c->leaf_r.resize((size_t)n));
  HIP_TRY(c->leaf_idx.resize((size_t)n));
  const size_t max_nodes = n > 1 ? (size_t)(n - 1) : 1;  // (a 4-wide node is made from at least one binary node)
  HIP_TRY(c->nodes.resize(max_nodes));

  lbvh_scratch& w = c->build;
  dev_buf<uint32_t>&bounds = w.bounds, &codes = w.codes, &codes_sorted = w.codes_sorted, &index = w.index, &index_sorted = w.index_sorted;
  dev_buf<int>&child0 = w.child0, &child1 = w.child1, &node_parent = w.node_parent, &leaf_parent = w.leaf_parent,
              &scalars = w.scalars, &slot_of = w.slot_of;
  dev_buf<unsigned int>& arrivals = w.arrivals;
  dev_buf<float>&node_box = w.node_box, &node_rmin = w.node_rmin;
  dev_buf<unsigned char>& temp = w.temp;
  auto cleanup = [&]() {
    if ((size_t)n > kKeepScratchSpheres) w.release();
  };
#define BUILD_TRY(expr)                                                                       \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      cleanup();                                                                              \
      return fail(RT_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_));                     \
    }                                                                                         \
  } while (0)

  const size_t nn = (size_t)n, ni = (size_t)(n > 1 ? n - 1 : 1);
  BUILD_TRY(bounds.resize(6));
  BUILD_TRY(codes.resize(nn));
  BUILD_TRY(codes_sorted.resize(nn));
  BUILD_TRY(index.resize(nn));
  BUILD_TRY(index_sorted.resize(nn));
  BUILD_TRY(child0.resize(ni));
  BUILD_TRY(child1.resize(ni));
  BUILD_TRY(node_parent.resize(ni));
  BUILD_TRY(leaf_parent.resize(nn));
  BUILD_TRY(scalars.resize(2));  // [0] max depth of the binary tree, [1] 4-wide node counter
  BUILD_TRY(slot_of.resize(max_nodes));  // the binary node each 4-wide node is made from
  BUILD_TRY(arrivals.resize(ni));
  BUILD_TRY(node_box.resize(6 * ni));
  BUILD_TRY(node_rmin.resize(ni));
  size_t temp_bytes = 0;
  BUILD_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, codes.p, codes_sorted.p, index.p, index_sorted.p, nn, 0, 30,
                                      c->stream));
  BUILD_TRY(temp.resize(temp_bytes > 0 ? temp_bytes : 1));

  hipStream_t st = c->stream;
  const unsigned blocks_n = (unsigned)((nn + 255) / 256);
  const unsigned blocks_i = (unsigned)((ni + 255) / 256);
  const int first_free = 1;  // the root keeps slot 0
  int wide_levels = 0;
  BUILD_TRY(hipEventRecord(c->ev_b0, st));
  BUILD_TRY(hipMemsetAsync(bounds.p, 0xFF, 3 * sizeof(uint32_t), st));
  BUILD_TRY(hipMemsetAsync(bounds.p + 3, 0x00, 3 * sizeof(uint32_t), st));
  BUILD_TRY(hipMemsetAsync(arrivals.p, 0, ni * sizeof(unsigned int), st));
  BUILD_TRY(hipMemsetAsync(scalars.p, 0, sizeof(int), st));
  BUILD_TRY(hipMemcpyAsync(scalars.p + 1, &first_free, sizeof(int), hipMemcpyHostToDevice, st));
  BUILD_TRY(hipMemsetAsync(leaf_parent.p, 0xFF, nn * sizeof(int), st));  // -1: a lone leaf has no parent
  lbvh_bounds_kernel<<<blocks_n < 1024u ? blocks_n : 1024u, 256, 0, st>>>(c->tree_sph.p, n, bounds.p);
  lbvh_morton_kernel<<<blocks_n, 256, 0, st>>>(c->tree_sph.p, n, bounds.p, codes.p, index.p);
  BUILD_TRY(rocprim::radix_sort_pairs(temp.p, temp_bytes, codes.p, codes_sorted.p, index.p, index_sorted.p, nn, 0, 30, st));
  lbvh_gather_kernel<<<blocks_n, 256, 0, st>>>(c->tree_sph.p, c->tree_r.p, c->tree_orig.p, index_sorted.p, n, c->leaf_sph.p,
                                               c->leaf_r.p, c->leaf_idx.p);
  if (n > 1) {
    lbvh_hierarchy_kernel<<<blocks_i, 256, 0, st>>>(codes_sorted.p, n, child0.p, child1.p, node_parent.p, leaf_parent.p);
    lbvh_fit_kernel<<<blocks_n, 256, 0, st>>>(c->leaf_sph.p, c->leaf_r.p, n, child0.p, child1.p, node_parent.p,
                                              leaf_parent.p, node_box.p, node_rmin.p, arrivals.p, scalars.p);
    // the 4-wide tree, a level per launch: level L is the slots the launches before it handed out
    BUILD_TRY(hipMemsetAsync(slot_of.p, 0, sizeof(int), st));  // the root is made from binary node 0
    int begin = 0, end = 1;
    while (begin < end) {
      lbvh_collapse_level_kernel<<<(unsigned)((end - begin + 255) / 256), 256, 0, st>>>(
          c->leaf_sph.p, c->leaf_r.p, child0.p, child1.p, node_box.p, node_rmin.p, n - 1, slot_of.p, begin, end, scalars.p + 1,
          c->nodes.p);
      int handed_out = 0;
      BUILD_TRY(hipMemcpyAsync(&handed_out, scalars.p + 1, sizeof(int), hipMemcpyDeviceToHost, st));
      BUILD_TRY(hipStreamSynchronize(st));
      if (handed_out < end || (size_t)handed_out > max_nodes) {
        cleanup();
        return fail(RT_ERR_DEVICE, "hierarchy build: %d node slots handed out of %zu", handed_out, max_nodes);
      }
      begin = end;
      end = handed_out;
      wide_levels++;
    }
  }
  BUILD_TRY(hipGetLastError());
  BUILD_TRY(hipEventRecord(c->ev_b1, st));
  int sc[2] = {0, 0};
  float rb[6] = {0, 0, 0, 0, 0, 0};
  BUILD_TRY(hipMemcpyAsync(sc, scalars.p, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
  if (n > 1) BUILD_TRY(hipMemcpyAsync(rb, node_box.p, 6 * sizeof(float), hipMemcpyDeviceToHost, st));
  float4 only;
  float only_r = 0.0f;
  if (n == 1) {
    BUILD_TRY(hipMemcpyAsync(&only, c->leaf_sph.p, sizeof(float4), hipMemcpyDeviceToHost, st));
    BUILD_TRY(hipMemcpyAsync(&only_r, c->leaf_r.p, sizeof(float), hipMemcpyDeviceToHost, st));
  }
  BUILD_TRY(hipStreamSynchronize(st));
  cleanup();
#undef BUILD_TRY
  if (n == 1) {
    const float r = std::fabs(only_r);
    rb[0] = only.x - r; rb[1] = only.y - r; rb[2] = only.z - r;
    rb[3] = only.x + r; rb[4] = only.y + r; rb[5] = only.z + r;
  }
  c->depth = 2 * wide_levels;  // (attach_overflow and rt_lbvh_depth count binary levels, two to a 4-wide one)
  c->n_nodes = n > 1 ? sc[1] : 0;
  c->abs_max = 0.0f;
  for (int a = 0; a < 3; a++) {
    c->root_lo[a] = rb[a];
    c->root_hi[a] = rb[3 + a];
    c->abs_max += std::fmax(std::fabs(rb[a]), std::fabs(rb[3 + a]));
  }
  c->have_build_time = true;
  return RT_OK;
}

// ---- hierarchy of a small scene, built on the host ---------------------------
// A scene whose tree fits LDS has at most a few thousand spheres: a top-down build by the surface
// area heuristic (every split position along the three axes is priced, a leaf holds one sphere)
// takes a fraction of a millisecond on one core (0.33 ms for 487 spheres) and gives the render kernels fewer nodes to visit than
// the Morton-order tree.  The binary tree is collapsed into 4-wide nodes along the cuts that make
// the summed surface of the 4-wide nodes smallest (sah_builder::plan_collapse).  Boxes are the
// same centre -+ |r| boxes the device build makes, so the traversal's box-growing rules apply as
// they are.  RT_BUILDER=lbvh keeps the device build for every scene.
constexpr int kSahMaxLeaves = 2048;
constexpr int kSahMaxWideDepth = 20;

struct sah_bin_node {
  float lo[3], hi[3], rmin;
  int left, right;  // children, or -1 / sphere (index into the tree arrays) for a leaf
};

struct sah_builder {
  const std::vector<float4>& sph;
  const std::vector<float>& radius;
  std::vector<sah_bin_node> bin;
  std::vector<int> order;          // spheres of the range being split
  std::vector<float> area_right;   // sweep scratch
  // outputs
  std::vector<bvh4_node> nodes;
  std::vector<int> leaf_order;     // leaf position -> index into the tree arrays
  int wide_depth = 0;

  sah_builder(const std::vector<float4>& s, const std::vector<float>& r) : sph(s), radius(r) {}

  static float half_area(const float* lo, const float* hi) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
  }
  void leaf_box(int i, float* lo, float* hi) const {
    const float r = std::fabs(radius[i]);
    const float c[3] = {sph[i].x, sph[i].y, sph[i].z};
    for (int a = 0; a < 3; a++) {
      lo[a] = c[a] - r;
      hi[a] = c[a] + r;
    }
  }
  static void grow(float* lo, float* hi, const float* l2, const float* h2) {
    for (int a = 0; a < 3; a++) {
      lo[a] = std::fmin(lo[a], l2[a]);
      hi[a] = std::fmax(hi[a], h2[a]);
    }
  }
  float centre(int i, int axis) const { return axis == 0 ? sph[i].x : axis == 1 ? sph[i].y : sph[i].z; }

  // Builds the subtree over order[b, e) and returns its node.
  int build(int b, int e) {
    const int id = (int)bin.size();
    bin.push_back(sah_bin_node());
    if (e - b == 1) {
      sah_bin_node& n = bin[id];
      leaf_box(order[b], n.lo, n.hi);
      n.rmin = std::fabs(radius[order[b]]);
      n.left = -1;
      n.right = order[b];
      return id;
    }
    float best_cost = 0.0f;
    int best_axis = -1, best_split = -1;
    std::vector<int> sorted(order.begin() + b, order.begin() + e), best_sorted;
    const int m = e - b;
    for (int axis = 0; axis < 3; axis++) {
      std::stable_sort(sorted.begin(), sorted.end(), [&](int x, int y) { return centre(x, axis) < centre(y, axis); });
      float lo[3], hi[3], l2[3], h2[3];
      leaf_box(sorted[m - 1], lo, hi);
      area_right[m - 1] = half_area(lo, hi);
      for (int i = m - 2; i >= 1; i--) {
        leaf_box(sorted[i], l2, h2);
        grow(lo, hi, l2, h2);
        area_right[i] = half_area(lo, hi);
      }
      leaf_box(sorted[0], lo, hi);
      for (int i = 1; i < m; i++) {  // the left part holds sorted[0, i)
        const float cost = half_area(lo, hi) * (float)i + area_right[i] * (float)(m - i);
        // (equal costs — coincident spheres — go to the split nearest the middle: a chain would be rejected later)
        if (best_axis < 0 || cost < best_cost || (cost == best_cost && std::abs(2 * i - m) < std::abs(2 * best_split - m))) {
          best_cost = cost;
          best_axis = axis;
          best_split = i;
        }
        leaf_box(sorted[i], l2, h2);
        grow(lo, hi, l2, h2);
      }
      if (best_axis == axis) best_sorted = sorted;
    }
    std::copy(best_sorted.begin(), best_sorted.end(), order.begin() + b);
    const int l = build(b, b + best_split);
    const int r = build(b + best_split, e);
    sah_bin_node& n = bin[id];
    for (int a = 0; a < 3; a++) {
      n.lo[a] = std::fmin(bin[l].lo[a], bin[r].lo[a]);
      n.hi[a] = std::fmax(bin[l].hi[a], bin[r].hi[a]);
    }
    n.rmin = std::fmin(bin[l].rmin, bin[r].rmin);
    n.left = l;
    n.right = r;
    return id;
  }

  // Which binary nodes become the children of a 4-wide node: the cut that makes the summed surface of all
  // 4-wide nodes below smallest (a ray visits a node with a probability proportional to its surface, and a
  // visit costs the same whether two or four slots are in use).  slots[n][k] = smallest sum for the subtree
  // of n handed k slots of its parent: either n itself takes one slot and becomes a node (its surface plus
  // its own best cut into four), or its two children share the k slots; a leaf costs nothing.
  std::vector<float> slots;   // 4 per binary node, k = 1 .. 4
  std::vector<unsigned char> cut;  // slots given to the left child, 0 = n takes one slot itself
  void plan_collapse() {
    slots.assign(4 * bin.size(), 0.0f);
    cut.assign(4 * bin.size(), 0);
    for (int n = (int)bin.size() - 1; n >= 0; n--) {  // children were created after their parents
      if (bin[n].left < 0) continue;
      const int l = bin[n].left, r = bin[n].right;
      auto shared = [&](int k, int& best_i) {  // the children of n share k >= 2 slots
        float best = 0.0f;
        best_i = 0;
        for (int i = 1; i < k; i++) {
          const float v = slots[4 * l + i - 1] + slots[4 * r + k - i - 1];
          if (best_i == 0 || v < best) {
            best = v;
            best_i = i;
          }
        }
        return best;
      };
      int i4;
      const float own = half_area(bin[n].lo, bin[n].hi) + shared(4, i4);  // n as a node of its own
      slots[4 * n] = own;
      for (int k = 2; k <= 4; k++) {
        int i;
        const float v = shared(k, i);
        if (v < own) {
          slots[4 * n + k - 1] = v;
          cut[4 * n + k - 1] = (unsigned char)i;
        } else {
          slots[4 * n + k - 1] = own;
        }
      }
      cut[4 * n] = (unsigned char)i4;  // for k = 1 the entry holds the node's own cut into four
    }
  }
  void gather(int n, int k, int* kids, int& nk) const {
    if (bin[n].left < 0 || k == 1 || cut[4 * n + k - 1] == 0) {
      kids[nk++] = n;
      return;
    }
    const int i = cut[4 * n + k - 1];
    gather(bin[n].left, i, kids, nk);
    gather(bin[n].right, k - i, kids, nk);
  }

  // The 4-wide node of binary node `id` (not a leaf); returns its slot.
  int emit(int id, int level) {
    wide_depth = std::max(wide_depth, level);
    int kids[4] = {-1, -1, -1, -1};
    int nk = 0;
    const int i4 = cut[4 * id];
    gather(bin[id].left, i4, kids, nk);
    gather(bin[id].right, 4 - i4, kids, nk);
    const int slot = (int)nodes.size();
    nodes.push_back(bvh4_node());
    bvh4_node nd;
    for (int s = 0; s < 4; s++) {
      if (s < nk) {
        const sah_bin_node& k = bin[kids[s]];
        nd.lox[s] = k.lo[0]; nd.loy[s] = k.lo[1]; nd.loz[s] = k.lo[2];
        nd.hix[s] = k.hi[0]; nd.hiy[s] = k.hi[1]; nd.hiz[s] = k.hi[2];
        nd.rmin[s] = k.rmin;
        if (k.left < 0) {
          nd.ref[s] = ~(int32_t)leaf_order.size();
          leaf_order.push_back(k.right);
        } else {
          nd.ref[s] = emit(kids[s], level + 1);
        }
      } else {  // unused slot: the inverted box of the device build (lbvh_collapse_kernel)
        nd.lox[s] = nd.loy[s] = nd.loz[s] = __builtin_inff();
        nd.hix[s] = nd.hiy[s] = nd.hiz[s] = -__builtin_inff();
        nd.rmin[s] = 0.0f;
        nd.ref[s] = RT_REF_EMPTY;
      }
    }
    nodes[slot] = nd;
    return slot;
  }
};

bool host_builder_enabled() {  // read at every upload, so that one process can compare the two builds
  const char* e = getenv("RT_BUILDER");
  return !(e && std::strcmp(e, "lbvh") == 0);
}

// Host build for a tree of 2 .. kSahMaxLeaves spheres with finite centres and radii.  Fills what
// build_lbvh fills.
int build_sah_host(rt_ctx* c, const std::vector<float4>& tree_sph, const std::vector<float>& tree_r,
                   const std::vector<int32_t>& tree_orig) {
  const int n = (int)tree_sph.size();
  HIP_TRY(hipEventRecord(c->ev_b0, c->stream));
  sah_builder b(tree_sph, tree_r);
  b.bin.reserve(2 * (size_t)n);
  b.order.resize((size_t)n);
  b.area_right.resize((size_t)n);
  for (int i = 0; i < n; i++) b.order[i] = i;
  const int root = b.build(0, n);
  b.nodes.reserve((size_t)n);
  b.plan_collapse();
  b.emit(root, 1);
  std::vector<float4> leaf_sph((size_t)n);
  std::vector<float> leaf_r((size_t)n);
  std::vector<int32_t> leaf_idx((size_t)n);
  for (int pos = 0; pos < n; pos++) {
    const int i = b.leaf_order[pos];
    leaf_sph[pos] = tree_sph[i];
    leaf_r[pos] = tree_r[i];
    leaf_idx[pos] = tree_orig[i];
  }
  HIP_TRY(c->leaf_sph.upload(leaf_sph, c->stream));
  HIP_TRY(c->leaf_r.upload(leaf_r, c->stream));
  HIP_TRY(c->leaf_idx.upload(leaf_idx, c->stream));
  HIP_TRY(c->nodes.upload(b.nodes, c->stream));
  HIP_TRY(hipEventRecord(c->ev_b1, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));  // the host vectors go out of scope
  c->n_nodes = (int)b.nodes.size();
  c->root = 0;
  c->depth = 2 * b.wide_depth;  // attach_overflow and rt_lbvh_depth count binary levels
  c->abs_max = 0.0f;
  for (int a = 0; a < 3; a++) {
    c->root_lo[a] = b.bin[root].lo[a];
    c->root_hi[a] = b.bin[root].hi[a];
    c->abs_max += std::fmax(std::fabs(c->root_lo[a]), std::fabs(c->root_hi[a]));
  }
  c->have_build_time = true;
  return RT_OK;
}

// Which search the kernels use for the uploaded scene.
scene_counts counts_of(const rt_ctx* c) {
  scene_counts n;
  n.n_nodes = c->n_nodes;
  n.n_leaf = c->n_leaf;
  n.n_spheres = c->n_spheres;
  n.n_materials = c->n_materials;
  return n;
}

// A scene of a few spheres is scanned whatever the caller asked for: a node step costs as much as a dozen sphere
// tests, and the answers are the same bit for bit.  Measured on subsets of the 487-sphere scene (tools/tiny_scene_rate.py):
// the scan is 10 % faster up to 16 spheres, level at 24, slower from 32; the 2-sphere noise scene renders 25 % faster
// and the 6-sphere default scene twice as fast.  RT_SCAN_UP_TO overrides the limit (0: a tree whenever one was asked
// for); read at every upload, so that one process can compare.
constexpr int kScanUpToSpheres = 16;
bool few_enough_to_scan(int n_spheres) {
  const char* e = getenv("RT_SCAN_UP_TO");
  return n_spheres <= (e ? atoi(e) : kScanUpToSpheres);
}

int choose_search(const rt_ctx* c) {
  const scene_counts n = counts_of(c);
  if (c->n_quads > 0)  // scenes with quads: the scan from HBM, or the tree read from HBM with leaf steps
    return c->accel == RT_ACCEL_LBVH && c->n_spheres >= 1 ? SEARCH_TREE_GLB_NODE_Q : SEARCH_LINEAR_Q;
  if (c->accel != RT_ACCEL_LBVH || c->n_spheres < 1 || c->scanned_for_size)
    return search_scene_lds_bytes<SEARCH_LINEAR_LDS>(n) <= kLdsTreeBudget && c->n_spheres >= 1 ? SEARCH_LINEAR_LDS
                                                                                               : SEARCH_LINEAR;
  bool pad_ray = true;
  if (c->n_leaf > 0) {
    // One pad per ray is worth it when the worst pad (ray origin about one
    // scene extent away, smallest sphere of the tree) stays below the typical radius.
    const float extent = (c->root_hi[0] - c->root_lo[0]) + (c->root_hi[1] - c->root_lo[1]) + (c->root_hi[2] - c->root_lo[2]);
    const float worst = std::sqrt(c->r_min * c->r_min + RT_PAD_K * extent * extent) - c->r_min;
    pad_ray = worst < c->r_median;
  }
  // tree + hit-record + material arrays must leave room for the stacks of 16 waves
  const bool lds = c->n_nodes <= 4096 && search_lds_bytes<SEARCH_TREE_LDS_RAY>(n) <= kMaxLdsBytes - 4096;
  if (lds) return pad_ray ? SEARCH_TREE_LDS_RAY : SEARCH_TREE_LDS_NODE;
  return pad_ray ? SEARCH_TREE_GLB_RAY : SEARCH_TREE_GLB_NODE;
}

const char* stored_name(int search) {
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

const char* search_name(int search, bool render) {
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
int vote_from_env(const char* name, int fallback) {
  const char* e = getenv(name);
  int x = e ? atoi(e) : fallback;
  return x < 1 ? 1 : (x > 64 ? 64 : x);
}
int phase_vote() {
  static int v = vote_from_env("RT_PHASE_VOTE", RT_PHASE_VOTE);
  return v;
}
int trace_vote() {
  static int v = vote_from_env("RT_TRACE_VOTE", RT_TRACE_VOTE);
  return v;
}
// Lanes with queued leaves a wave waits for before it takes a leaf step.
int leaf_vote() {
  static int v = vote_from_env("RT_LEAF_VOTE", RT_LEAF_VOTE);
  return v;
}

// Cost levels per power of two in the main pass's sort key, as a power of two (RT_COST_STEPS_LOG2 overrides, 0..6).
int cost_steps_log2() {
  static int v = [] {
    const char* e = getenv("RT_COST_STEPS_LOG2");
    int x = e ? atoi(e) : 3;
    return x < 0 ? 0 : (x > 6 ? 6 : x);
  }();
  return v;
}

// How many of the most expensive pixels the main pass may cut into several jobs (RT_SPLIT_ROWS overrides; the
// memory budget for their per-sample radiance limits it further).
int split_rows() {
  static int v = [] {
    const char* e = getenv("RT_SPLIT_ROWS");
    return e ? atoi(e) : -1;  // negative: chosen from the frame size
  }();
  return v;
}

// Fewest jobs a pixel that may be cut is cut into.  The jobs of a pixel are neighbours in the main pass's order, so a
// wave that holds k jobs each of 64 / k pixels follows rays through a k times smaller part of the image than one
// that holds 64 pixels — what a frame of k times the pixels gets for nothing (measured: the same scene runs 14 %
// faster per path at 3840x2160 than at 1920x1080 with one job per pixel).  Chosen so that a lane sees about 48 jobs:
// 6 at 1920x1080 (main pass 61.8 -> 54 ms), 3 at 2560x1440, 1 from about 3500x2000 on (there more jobs only add
// traffic); RT_MIN_PARTS overrides.
int min_parts(size_t n_rect, size_t lanes) {
  static int forced = [] {
    const char* e = getenv("RT_MIN_PARTS");
    const int x = e ? atoi(e) : 0;
    return x < 0 ? 0 : (x > 64 ? 64 : x);
  }();
  if (forced > 0) return forced;
  const size_t k = 48 * lanes / (n_rect > 0 ? n_rect : 1);
  return (int)(k < 1 ? 1 : (k > 8 ? 8 : k));
}
// Bytes the per-sample radiance of cut pixels may take (RT_SPLIT_BUDGET_MB overrides kSplitBudgetBytes).
size_t split_budget_bytes() {
  static size_t v = [] {
    const char* e = getenv("RT_SPLIT_BUDGET_MB");
    return e && atoll(e) > 0 ? (size_t)atoll(e) << 20 : kSplitBudgetBytes;
  }();
  return v;
}

// What the two per-sample stores may take in this render: the context's budgets (rt_set_scratch_budget, else the
// defaults), and never more than a share of what the device can still give — free memory plus what the two buffers
// already hold.
void scratch_budgets(rt_ctx* c, size_t& split, size_t& probe) {
  split = c->split_budget ? c->split_budget : split_budget_bytes();
  probe = c->probe_budget ? c->probe_budget : kProbeBudgetBytes;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  const size_t avail = free_b + (c->split_samples.n + c->probe_samples.n) * sizeof(float);
  split = std::min(split, avail / 2);
  probe = std::min(probe, avail / 8);
}

// Most sweeps of a probe pass (RT_PROBE_SWEEPS overrides, 1 = unsplit probe).
int probe_sweeps() {
  static int v = [] {
    const char* e = getenv("RT_PROBE_SWEEPS");
    int x = e ? atoi(e) : 16;
    return x < 1 ? 1 : (x > 64 ? 64 : x);
  }();
  return v;
}

// How finely the main pass of two-pass rendering cuts expensive pixels: no job should exceed
// 1/fineness of what a lane does on average (RT_SPLIT_FINENESS overrides, for tuning).  A 1920x1080 frame, whose
// every pixel is cut into six parts anyway (min_parts), runs the same from 16 to 128; frames with fewer pixels per lane
// balance better with finer jobs: 960x540 +4 %, an eighth of the 1080p frame (one rank's strips of eight) +3 %, and
// 3840x2160x4096 +1 % at 64 against 16.
int split_fineness() {
  static int v = [] {
    const char* e = getenv("RT_SPLIT_FINENESS");
    int x = e ? atoi(e) : 64;
    return x < 1 ? 1 : (x > 256 ? 256 : x);
  }();
  return v;
}

// Samples per pixel of the cost-probe pass, and the switch for two-pass
// rendering (RT_TWO_PASS=0 in the environment turns it off, for comparison).
// The probe takes an eighth of the samples, between 8 and a cap: 12 when it runs in sweeps (counter RNG; a sample of
// the probe costs about a quarter more than one of the sorted main pass, and 12 measure a pixel's cost well enough:
// 1920x1080x256 on 487 spheres ran 0.9 % faster than with 32), 16 when every pixel's probe is one job.  RT_PROBE_SAMPLES sets the length itself, for tuning (the split of a frame into probe
// and main pass never changes the image).  Two-pass rendering needs at least 64 samples (16 with the reference's generator).
constexpr int kMinTwoPassSamples = 64, kMinTwoPassSamplesSequential = 16;
int probe_samples(int n_samples, bool swept) {
  static int cap = [] {
    const char* e = getenv("RT_PROBE_SAMPLES");
    int x = e ? atoi(e) : 0;
    return x < 0 ? 0 : (x > 256 ? 256 : x);
  }();
  if (cap > 0) return cap < n_samples - 8 ? cap : n_samples - 8;  // at least 8 samples stay for the main pass
  const int most = swept ? 12 : 16;
  int n = n_samples / 8;
  n = n < 8 ? 8 : n;
  return n > most ? most : n;
}
// Sweeps for a pass of n samples per pixel: jobs of two samples (one sample per job wastes the lanes' reserve of
// camera rays, four and more leave too few jobs in a tile's neighbourhood running together), within the most allowed.
int sweeps_for(int n) {
  const bool fixed = getenv("RT_PROBE_SWEEPS") != nullptr;
  int s = fixed ? probe_sweeps() : (n + 1) / 2;
  s = s > probe_sweeps() ? probe_sweeps() : s;
  s = s > n ? n : s;
  return s < 1 ? 1 : s;
}
bool stored_kernel_enabled() {  // RT_STORED_KERNEL=0: every pass in the kernel whose lanes own their pixels (comparison)
  static bool v = [] {
    const char* e = getenv("RT_STORED_KERNEL");
    return !(e && e[0] == '0');
  }();
  return v;
}
bool two_pass_enabled() {
  static bool v = [] {
    const char* e = getenv("RT_TWO_PASS");
    return !(e && e[0] == '0');
  }();
  return v;
}

int check_ctx(rt_ctx* c) {
  if (!c) return fail(RT_ERR_INVALID, "null context");
  HIP_TRY(hipSetDevice(c->device));
  return RT_OK;
}

// The gradient table of the noise texture does not depend on the scene: built once per context.
int ensure_noise_table(rt_ctx* c) {
  if (c->perlin_grad.p) return RT_OK;
  const size_t side = 2 * (size_t)kPerlinHalf;
  HIP_TRY(c->perlin_grad.resize(side * side));
  rt_perlin_table_kernel<<<(unsigned)((side * side + 255) / 256), 256, 0, c->stream>>>(c->perlin_grad.p, kPerlinHalf);
  HIP_TRY(hipGetLastError());
  return RT_OK;
}

int check_params(rt_ctx* c, const rt_params* p) {
  if (!p) return fail(RT_ERR_INVALID, "null params");
  if (!c->have_scene) return fail(RT_ERR_STATE, "rt_render before rt_scene_upload");
  if (p->width < 1 || p->height < 1 || (int64_t)p->width * p->height > (int64_t)1 << 30)
    return fail(RT_ERR_INVALID, "bad image size %d x %d", p->width, p->height);
  if (p->tile_x0 < 0 || p->tile_y0 < 0 || p->tile_x1 > p->width || p->tile_y1 > p->height || p->tile_x0 > p->tile_x1 ||
      p->tile_y0 > p->tile_y1)
    return fail(RT_ERR_INVALID, "tile [%d,%d)x[%d,%d) outside %d x %d", p->tile_x0, p->tile_x1, p->tile_y0, p->tile_y1,
                p->width, p->height);
  if (p->sample_begin < 0 || p->sample_end < p->sample_begin) return fail(RT_ERR_INVALID, "bad sample range");
  if (p->rng_mode != RT_RNG_XORSHIFT_REF && p->rng_mode != RT_RNG_COUNTER) return fail(RT_ERR_INVALID, "bad rng_mode");
  if (p->strip_count != 0 && p->strip_index >= p->strip_count)
    return fail(RT_ERR_INVALID, "strip_index %u must be below strip_count %u", (unsigned)p->strip_index, (unsigned)p->strip_count);
  // the reference's generator is seeded per pixel and runs through the pixel's samples from the first one
  // (fs:105-110): a range that starts later would draw the first range's numbers again
  if (p->rng_mode == RT_RNG_XORSHIFT_REF && p->sample_begin != 0)
    return fail(RT_ERR_INVALID, "RT_RNG_XORSHIFT_REF renders samples from 0 only (sample_begin = %d): use RT_RNG_COUNTER for sample ranges",
                p->sample_begin);
  return RT_OK;
}

// Rows of the call's share of its tile: the tile's rows, or with interleaving those of the share's strips of eight
// (the tile's last strip may be short).
int share_rows(const rt_params* p) {
  const int th = p->tile_y1 - p->tile_y0;
  if (p->strip_count == 0) return th;
  const int strips = (th + 7) / 8, n = p->strip_count, r = p->strip_index;
  if (r >= strips) return 0;
  const int mine = (strips - r + n - 1) / n;
  int rows = mine * 8;
  if ((strips - 1) % n == r && th % 8 != 0) rows -= 8 - th % 8;  // the short last strip is mine
  return rows;
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

// Whether the uploaded scene's search leaves room in LDS for the reserves of rt_render_stored_kernel.
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

// Empty reserves that make a refill worth a wave (RT_REFILL_VOTE overrides, for tuning).
int refill_vote() {
  static int v = vote_from_env("RT_REFILL_VOTE", RT_REFILL_VOTE);
  return v;
}

int refill_min() {
  static int v = vote_from_env("RT_REFILL_MIN", RT_REFILL_MIN);
  return v;
}

template <int SEARCH>
hipError_t launch_render(rt_ctx* c, const rt_params* p, bool counted, hipStream_t st, const render_args& a) {
  if (p->rng_mode == RT_RNG_XORSHIFT_REF)
    return counted ? launch_render_one<SEARCH, RT_RNG_XORSHIFT_REF, true>(c, p, st, a)
                   : launch_render_one<SEARCH, RT_RNG_XORSHIFT_REF, false>(c, p, st, a);
  return counted ? launch_render_one<SEARCH, RT_RNG_COUNTER, true>(c, p, st, a)
                 : launch_render_one<SEARCH, RT_RNG_COUNTER, false>(c, p, st, a);
}

// `fixed` (nullable): memory the traversal-stack overflow must come from instead of a buffer of the context
// (fixed_bytes of it); hipErrorOutOfMemory when the tree's depth needs more than it holds.
template <int SEARCH, bool COUNTED>
hipError_t launch_trace_one(rt_ctx* c, hipStream_t st, trace_args a, unsigned char* fixed, size_t fixed_bytes) {
  using T = search_traits<SEARCH>;
  const size_t lds = search_lds_bytes<SEARCH>(counts_of(c));
  hipError_t e = allow_lds(rt_trace_kernel<SEARCH, COUNTED>, lds);
  if (e != hipSuccess) return e;
  int per_cu = 0;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rt_trace_kernel<SEARCH, COUNTED>, T::threads, lds);
  if (e != hipSuccess) return e;
  if (per_cu < 1) per_cu = 1;
  long long blocks = (long long)per_cu * c->n_cu;
  const long long needed = (long long)((a.n + T::threads - 1) / T::threads);
  if (blocks > needed) blocks = needed;
  if (blocks < 1) blocks = 1;
  if (fixed) {
    if (!attach_overflow_fixed<SEARCH>(c, blocks, a.sc, fixed, fixed_bytes)) return hipErrorOutOfMemory;
  } else {
    e = attach_overflow<SEARCH>(c, blocks, a.sc);
    if (e != hipSuccess) return e;
  }
  rt_trace_kernel<SEARCH, COUNTED><<<dim3((unsigned)blocks), T::threads, lds, st>>>(a);
  return hipGetLastError();
}

template <int SEARCH>
hipError_t launch_trace(rt_ctx* c, bool counted, hipStream_t st, const trace_args& a, unsigned char* fixed = nullptr,
                        size_t fixed_bytes = 0) {
  return counted ? launch_trace_one<SEARCH, true>(c, st, a, fixed, fixed_bytes)
                 : launch_trace_one<SEARCH, false>(c, st, a, fixed, fixed_bytes);
}

int render_impl(rt_ctx* c, const rt_params* p, bool counted) {
  int rc = check_ctx(c);
  if (rc) return rc;
  rc = check_params(c, p);
  if (rc) return rc;
  const size_t n_px = (size_t)p->width * (size_t)p->height;
  if (c->accum_ext) {
    if (p->width != c->acc_w || p->height != c->acc_h)
      return fail(RT_ERR_STATE, "image %d x %d does not match the caller's accumulator %d x %d", p->width, p->height,
                  c->acc_w, c->acc_h);
  } else if (p->width != c->acc_w || p->height != c->acc_h || !c->accum_own.p) {
    HIP_TRY(c->accum_own.resize(3 * n_px));
    HIP_TRY(hipMemsetAsync(c->accum_own.p, 0, 3 * n_px * sizeof(float), c->stream));
    c->acc_w = p->width;
    c->acc_h = p->height;
    c->samples_lo = c->samples_hi = 0;
    c->samples_override = -1;
  }
  if (c->samples_hi == c->samples_lo) {
    c->samples_lo = p->sample_begin;
    c->samples_hi = p->sample_end;
  } else {
    if (p->sample_begin < c->samples_lo) c->samples_lo = p->sample_begin;
    if (p->sample_end > c->samples_hi) c->samples_hi = p->sample_end;
  }

  render_args a;
  a.p = *p;
  a.p.seed_tex = nullptr;
  a.sc = c->view();
  a.accum = c->accum();
  a.seed_tex = nullptr;
  a.counters = nullptr;
  HIP_TRY(c->queue.resize(1));
  a.queue = reinterpret_cast<unsigned int*>(c->queue.p);
  a.vote = phase_vote();
  a.leaf_vote = leaf_vote();
  if (p->rng_mode == RT_RNG_XORSHIFT_REF && p->seed_tex) {
    HIP_TRY(c->seed_tex.resize(n_px));
    HIP_TRY(hipMemcpyAsync(c->seed_tex.p, p->seed_tex, n_px * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    a.seed_tex = c->seed_tex.p;
  }
  if (counted) {
    HIP_TRY(c->counters.resize(sizeof(rt_counters) / sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(c->counters.p, 0, sizeof(rt_counters), c->stream));
    a.counters = c->counters.p;
  }
  const int tw = p->tile_x1 - p->tile_x0, th = share_rows(p);
  a.vrows = th;
  if (tw == 0 || th == 0 || p->sample_end == p->sample_begin) return RT_OK;
  bool used_stored = false;  // the kernel the frame's last (and longest) pass ran in
  auto launch = [&](const render_args& ra, const rt_params* rp) -> hipError_t {
    used_stored = false;
    hipError_t qe = hipMemsetAsync(c->queue.p, 0, sizeof(unsigned long long), c->stream);
    if (qe != hipSuccess) return qe;
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
  };
  auto launch_stored_pass = [&](const render_args& ra, bool owned = false) -> hipError_t {
    used_stored = true;
    hipError_t qe = hipMemsetAsync(c->queue.p, 0, sizeof(unsigned long long), c->stream);
    if (qe != hipSuccess) return qe;
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
  };
  // Passes whose every sample is stored run in the kernel that keeps camera rays in reserve (counter generator; a
  // bounce limit of zero has no rays to trace; the reserves must fit LDS beside the scene).
  const bool stored_ok = p->rng_mode == RT_RNG_COUNTER && p->bounce_limit > 0u && stored_fits(c) && stored_kernel_enabled();
  a.reserve_offset = 0;
  a.pairs = 0;
  a.refill_vote = refill_vote();
  a.refill_min = refill_min();
  a.pass = 0;
  a.sweeps = 1;
  a.jobs = nullptr;
  a.n_jobs = nullptr;
  a.split_samples = nullptr;
  a.main_samples = 0;
  a.split_rows = 0;
  a.carry_colour = nullptr;
  a.carry_rng = a.carry_cost = nullptr;
  // A pixel is the unit of work (its sum must be formed in sample order), so
  // the launch ends with the lanes that drew the most expensive pixels last.
  // With enough samples a short probe pass measures every pixel's cost, the
  // pixels are sorted most-expensive-first, and the main pass takes them in
  // that order: the tail then consists of the cheapest pixels.
  const size_t n_rect = (size_t)tw * (size_t)th;
  const int n_samples = p->sample_end - p->sample_begin;
  // (with the counter RNG shorter frames are rendered in sweeps as a whole, see below; the reference's generator
  // cannot start in the middle of a pixel, so there a probe pays from 16 samples on: 1080p x 32 spp 3 360 -> 4 340 Mpaths/s)
  const int min_two_pass = p->rng_mode == RT_RNG_COUNTER ? kMinTwoPassSamples : kMinTwoPassSamplesSequential;
  const bool two_pass = two_pass_enabled() && n_samples >= min_two_pass && n_rect >= 16384;
  HIP_TRY(hipEventRecord(c->ev_k0, c->stream));
  hipError_t e;
  // Few samples, counter RNG: the frame as a whole is rendered in sweeps (short jobs in tile order, per-sample
  // radiance summed in order afterwards) — a pixel per job would end with the expensive pixels' lanes working alone.
  size_t split_budget, probe_budget;
  scratch_budgets(c, split_budget, probe_budget);
  bool whole_in_sweeps = !two_pass && two_pass_enabled() && stored_ok && probe_sweeps() > 1 &&
                         n_samples >= 2 && n_rect >= 16384 &&
                         n_rect * stored_samples(n_samples) * kSampleFloats * sizeof(float) <= probe_budget;
  // (a store that cannot be had costs speed, never the frame: the plain schedules below need none)
  if (whole_in_sweeps && c->probe_samples.fit(n_rect * stored_samples(n_samples) * kSampleFloats) != hipSuccess) whole_in_sweeps = false;
  if (whole_in_sweeps) {
    render_args sw = a;
    sw.pass = 3;
    sw.sweeps = sweeps_for(n_samples);
    sw.split_samples = reinterpret_cast<float4*>(c->probe_samples.p);
    sw.main_samples = n_samples;
    e = launch_stored_pass(sw);
    if (e == hipSuccess) {
      rt_probe_sum_kernel<<<(unsigned)((n_rect + 255) / 256), 256, 0, c->stream>>>(
          (uint32_t)n_rect, reinterpret_cast<const float4*>(c->probe_samples.p), n_samples, p->tile_x0, p->tile_y0, tw, p->width,
          p->strip_count, p->strip_index, c->accum(), nullptr, nullptr, nullptr);
      e = hipGetLastError();
    }
  } else if (!two_pass) {
    e = launch(a, p);
  } else {
    HIP_TRY(c->carry_colour.resize(3 * n_rect));
    HIP_TRY(c->carry_rng.resize(n_rect));
    HIP_TRY(c->carry_cost.resize(n_rect));
    HIP_TRY(c->carry_key.resize(n_rect));
    HIP_TRY(c->carry_idx.resize(n_rect));
    HIP_TRY(c->order_key.resize(n_rect));
    HIP_TRY(c->order_idx.resize(n_rect));
    size_t temp_bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, c->carry_key.p, c->order_key.p, c->carry_idx.p, c->order_idx.p,
                                      n_rect, 0, 31, c->stream));
    HIP_TRY(c->sort_temp.resize(temp_bytes > 0 ? temp_bytes : 1));
    render_args probe = a;
    probe.pass = 1;
    probe.carry_colour = c->carry_colour.p;
    probe.carry_rng = c->carry_rng.p;
    probe.carry_cost = c->carry_cost.p;
    // Counter RNG: the probe runs in sweeps (see render_args), its samples in a buffer of their own.
    bool swept = stored_ok && probe_sweeps() > 1;
    int n_probe = probe_samples(n_samples, swept);
    if (swept && (n_rect * stored_samples(n_probe) * kSampleFloats * sizeof(float) > probe_budget ||
                  c->probe_samples.fit(n_rect * stored_samples(n_probe) * kSampleFloats) != hipSuccess)) {
      swept = false;
      n_probe = probe_samples(n_samples, false);
    }
    const int sweeps = sweeps_for(n_probe);
    probe.p.sample_end = p->sample_begin + n_probe;
    if (swept) {
      probe.pass = 3;
      probe.sweeps = sweeps;
      probe.split_samples = reinterpret_cast<float4*>(c->probe_samples.p);
      probe.main_samples = n_probe;
    }
    e = swept ? launch_stored_pass(probe) : launch(probe, p);
    if (e == hipSuccess && swept) {
      rt_probe_sum_kernel<<<(unsigned)((n_rect + 255) / 256), 256, 0, c->stream>>>(
          (uint32_t)n_rect, reinterpret_cast<const float4*>(c->probe_samples.p), n_probe, p->tile_x0, p->tile_y0, tw, p->width,
          p->strip_count, p->strip_index, c->accum(), c->carry_colour.p, c->carry_rng.p, c->carry_cost.p);
      e = hipGetLastError();
      probe.pass = 1;  // what follows copies `probe` for the main pass
      probe.sweeps = 1;
    }
    if (e == hipSuccess) {
      rt_sort_key_kernel<<<(unsigned)((n_rect + 255) / 256), 256, 0, c->stream>>>((uint32_t)n_rect, tw, cost_steps_log2(), /*8x8 tiles*/ 3, c->carry_cost.p, c->carry_key.p,
                                                                                 c->carry_idx.p);
      e = hipGetLastError();
    }
    if (e == hipSuccess)
      e = rocprim::radix_sort_pairs(c->sort_temp.p, temp_bytes, c->carry_key.p, c->order_key.p, c->carry_idx.p,
                                    c->order_idx.p, n_rect, 0, 31, c->stream);
    if (e == hipSuccess) {
      // Plan the main pass: one job per pixel, except that the most expensive pixels are cut into
      // several sample ranges (counter RNG only: the reference's generator runs through a pixel's
      // samples and cannot start in the middle).
      const int main_begin = p->sample_begin + n_probe, main_samples = p->sample_end - main_begin;
      const uint32_t n = (uint32_t)n_rect;
      // a job should be a small fraction (1 / split_fineness()) of a lane's fair share of the frame
      const uint32_t lanes = (uint32_t)c->n_cu * 1024u * (uint32_t)split_fineness();
      size_t want = split_budget / (stored_samples(main_samples) * kSampleFloats * sizeof(float));
      // (only where rays walk a tree worth the name: the scan and a tree of a few spheres gain nothing from
      // neighbouring rays, and every sample stored and read back costs them 24 bytes of traffic — the two-sphere
      // noise scene ran 18 % slower with six jobs per pixel)
      const bool scans = c->search == SEARCH_LINEAR || c->search == SEARCH_LINEAR_LDS;
      const bool walks_tree = !scans && c->n_nodes >= 32;
      const uint32_t fewest = walks_tree ? (uint32_t)min_parts(n_rect, (size_t)c->n_cu * 1024) : 1u;
      // Where a sample costs far more than the 24 bytes of traffic its storing adds (a tree walk, a scan over many
      // spheres), every pixel may be cut, as far as the budget goes, the most expensive ones first.  Elsewhere
      // only the few most expensive ones — or, in a frame with fewer than four pixels per lane, which balances
      // only if most pixels may be cut (960x540: +15 %), all of them.
      const bool heavy = walks_tree || (scans && c->n_spheres >= 64);
      const size_t row_cap = split_rows() >= 0 ? (size_t)split_rows()
                                                : (heavy || n_rect < (size_t)c->n_cu * 4096 ? n_rect : (size_t)65536);
      uint32_t max_split = (uint32_t)(want > row_cap ? row_cap : want);
      if (p->rng_mode != RT_RNG_COUNTER) max_split = 0;
      // the store of the cut pixels' samples: when the device cannot give it, fewer pixels are cut, in the end none
      while (max_split > 0 && c->split_samples.fit((size_t)max_split * stored_samples(main_samples) * kSampleFloats) != hipSuccess) max_split /= 2;
      const int splittable = max_split > 0 ? 1 : 0;
      // every pixel has a row in the store: the main pass too runs in the kernel with rays in reserve
      const int all_rows = splittable && max_split >= n && stored_ok ? 1 : 0;
      // ... whose records go out in pairs, as long as the store stays below 2^31 records (a slot number's top bit
      // marks the record that waits); RT_SAMPLE_PAIRS=0: every record on its own (comparison)
      static const bool pairs_enabled = [] {
        const char* e = getenv("RT_SAMPLE_PAIRS");
        return !(e && e[0] == '0');
      }();
      const int pairs = all_rows && pairs_enabled && (unsigned long long)n * (unsigned long long)main_samples < (1ull << 31) ? 1 : 0;
      HIP_TRY(c->plan_parts.resize(n));
      HIP_TRY(c->plan_first.resize(n));
      HIP_TRY(c->plan_counts.resize(1));
      HIP_TRY(c->plan_total.resize(1));
      HIP_TRY(c->plan_jobs.resize((size_t)n * (size_t)fewest + (size_t)lanes + 64));
      size_t scan_bytes = 0;
      HIP_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, c->plan_parts.p, c->plan_first.p, 0u, n, rocprim::plus<uint32_t>(), c->stream));
      HIP_TRY(c->scan_temp.resize(scan_bytes > 0 ? scan_bytes : 1));
      HIP_TRY(hipMemsetAsync(c->plan_total.p, 0, sizeof(unsigned long long), c->stream));
      const unsigned grid = (n + 255) / 256;
      rt_plan_cost_kernel<<<grid < 256 ? grid : 256, 256, 0, c->stream>>>(c->carry_cost.p, n, c->plan_total.p);
      rt_plan_count_kernel<<<grid, 256, 0, c->stream>>>(c->carry_cost.p, c->order_idx.p, n, c->plan_total.p, lanes, max_split, main_samples,
                                                       splittable, fewest, c->plan_parts.p);
      HIP_TRY(hipGetLastError());
      HIP_TRY(rocprim::exclusive_scan(c->scan_temp.p, scan_bytes, c->plan_parts.p, c->plan_first.p, 0u, n, rocprim::plus<uint32_t>(),
                                      c->stream));
      rt_plan_jobs_kernel<<<grid, 256, 0, c->stream>>>(c->order_idx.p, c->plan_parts.p, c->plan_first.p, n, main_begin,
                                                      p->sample_end, all_rows, pairs, c->plan_jobs.p, c->plan_counts.p);
      HIP_TRY(hipGetLastError());
      render_args mainp = probe;
      mainp.pass = 2;
      mainp.p.sample_begin = main_begin;
      mainp.p.sample_end = p->sample_end;
      mainp.jobs = c->plan_jobs.p;
      mainp.n_jobs = c->plan_counts.p;
      mainp.split_samples = reinterpret_cast<float4*>(c->split_samples.p);
      mainp.main_samples = main_samples;
      mainp.split_rows = n < max_split ? n : max_split;
      mainp.pairs = pairs;
      // Not every pixel has a row: the same kernel in the form whose lanes sum whole pixels in registers (counter
      // generator; RT_OWNED_RESERVE=0 keeps the kernel without reserves, for comparison)
      static const bool owned_enabled = [] {
        const char* e = getenv("RT_OWNED_RESERVE");
        return !(e && e[0] == '0');
      }();
      // (a slot number's top bit marks a whole pixel's sample there, and the pixel's number takes 29 bits)
      const bool owned = !all_rows && stored_ok && owned_enabled && n_rect <= (size_t)RT_TAG_PIXEL &&
                         (unsigned long long)mainp.split_rows * (unsigned long long)main_samples < (1ull << 31);
      if (owned) mainp.pairs = 0;
      e = all_rows ? launch_stored_pass(mainp) : (owned ? launch_stored_pass(mainp, true) : launch(mainp, p));
      if (e == hipSuccess && splittable) {
        const uint32_t cand = n < max_split ? n : max_split;
        rt_sum_split_kernel<<<(cand + 255) / 256, 256, 0, c->stream>>>(c->order_idx.p, c->plan_parts.p, cand, c->carry_colour.p,
                                                                      reinterpret_cast<const float4*>(c->split_samples.p), main_samples, cand, p->tile_x0, p->tile_y0,
                                                                      tw, p->width, p->strip_count, p->strip_index, all_rows, pairs, c->accum());
        e = hipGetLastError();
      }
    }
  }
  if (e != hipSuccess) return fail(RT_ERR_DEVICE, "render launch: %s", hipGetErrorString(e));
  c->last_kernel = used_stored ? stored_name(c->search) : search_name(c->search, true);
  HIP_TRY(hipEventRecord(c->ev_k1, c->stream));
  c->have_kernel_time = true;
  return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_error.c_str(); }

int rt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int rt_init(int device, rt_ctx** out) {
  if (!out) return fail(RT_ERR_INVALID, "null out pointer");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n < 1)
    return fail(RT_ERR_NO_GPU, "no HIP device visible (%s); this library has no CPU path",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  if (device < 0 || device >= n) return fail(RT_ERR_INVALID, "device %d out of range [0,%d)", device, n);
  HIP_TRY(hipSetDevice(device));
  if (device < kMaxDevices && !g_noalloc_scratch[device]) {  // once per device, with that device current
    void* ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(rt_noalloc_scratch)));
    g_noalloc_scratch[device] = static_cast<unsigned char*>(ptr);
  }
  rt_ctx* c = new rt_ctx();
  c->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
  if (c->n_cu < 1) c->n_cu = 256;
  hipError_t err = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (err == hipSuccess) err = hipEventCreate(&c->ev_k0);
  if (err == hipSuccess) err = hipEventCreate(&c->ev_k1);
  if (err == hipSuccess) err = hipEventCreate(&c->ev_b0);
  if (err == hipSuccess) err = hipEventCreate(&c->ev_b1);
  if (err != hipSuccess) {
    rt_destroy(c);
    return fail(RT_ERR_DEVICE, "context set-up: %s", hipGetErrorString(err));
  }
  *out = c;
  return RT_OK;
}

void rt_destroy(rt_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  (void)rt_comm_destroy(c);
  c->sph.release(); c->inv_r.release(); c->sph_mat.release(); c->mat_type.release(); c->mat_a.release();
  c->quad_qd.release(); c->quad_u.release(); c->quad_v.release(); c->quad_n.release(); c->quad_w.release(); c->quad_idx.release(); c->quad_of.release();
  c->big_sph.release(); c->big_idx.release(); c->tree_sph.release(); c->tree_r.release(); c->tree_orig.release();
  c->nodes.release(); c->qnodes.release(); c->leaf_sph.release(); c->leaf_r.release(); c->leaf_idx.release(); c->leaf_rec.release();
  c->accum_own.release(); c->seed_tex.release(); c->queue.release(); c->counters.release(); c->ovf.release(); c->build.release(); c->perlin_grad.release();
  c->carry_colour.release(); c->carry_rng.release(); c->carry_cost.release(); c->carry_key.release(); c->carry_idx.release();
  c->plan_parts.release(); c->plan_first.release(); c->plan_counts.release(); c->plan_total.release();
  c->plan_jobs.release(); c->split_samples.release(); c->probe_samples.release(); c->scan_temp.release();
  c->order_key.release(); c->order_idx.release(); c->sort_temp.release(); c->out_linear.release(); c->out_rgb8.release();
  if (c->ev_k0) (void)hipEventDestroy(c->ev_k0);
  if (c->ev_k1) (void)hipEventDestroy(c->ev_k1);
  if (c->ev_b0) (void)hipEventDestroy(c->ev_b0);
  if (c->ev_b1) (void)hipEventDestroy(c->ev_b1);
  for (int b = 0; b < 2; b++) {
    if (c->ev_in[b]) (void)hipEventDestroy(c->ev_in[b]);
    if (c->ev_run[b]) (void)hipEventDestroy(c->ev_run[b]);
    if (c->ev_out[b]) (void)hipEventDestroy(c->ev_out[b]);
    c->tb_o[b].release(); c->tb_d[b].release(); c->tb_t[b].release(); c->tb_n[b].release(); c->tb_id[b].release();
  }
  if (c->copy_in) (void)hipStreamDestroy(c->copy_in);
  if (c->copy_out) (void)hipStreamDestroy(c->copy_out);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

}  // extern "C"

namespace {

// rt_scene_upload and rt_scene_upload_objects.  `tagged`: the records carry their kind at byte 8 and are 32 (sphere)
// or 64 (quad) bytes long, `n_bytes` in all; otherwise n_spheres 32-byte sphere slots whose bytes 8..15 mean nothing.
int upload_impl(rt_ctx* c, const void* spheres_std140, size_t n_bytes, int n_spheres, bool tagged, const void* materials_std140,
                int n_materials, int accel) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (n_spheres < 0 || n_materials < 0 || (n_spheres > 0 && !spheres_std140) || (n_materials > 0 && !materials_std140))
    return fail(RT_ERR_INVALID, "bad scene arguments");
  if (accel != RT_ACCEL_LINEAR && accel != RT_ACCEL_LBVH) return fail(RT_ERR_INVALID, "bad accel %d", accel);
  c->have_scene = false;

  // Flatten the 32-byte slots (sphere.h:22-35, material.h:29-46) into arrays.
  const size_t ns = (size_t)n_spheres, nm = (size_t)n_materials;
  std::vector<float4> sph(ns);
  std::vector<float> radius(ns), inv_r(ns);
  std::vector<int32_t> sph_mat(ns), mat_type(nm), quad_of(ns, -1), quad_idx;
  std::vector<float4> mat_a(nm), quad_qd, quad_u, quad_v, quad_n, quad_w;
  std::vector<char> is_quad(ns, 0);
  const uint8_t* sp = static_cast<const uint8_t*>(spheres_std140);
  size_t off = 0;
  for (size_t i = 0; i < ns; i++) {
    int32_t kind = RT_OBJECT_SPHERE;
    if (tagged) {
      if (off + RT_RECORD_BYTES > n_bytes) return fail(RT_ERR_INVALID, "object %zu begins beyond the %zu bytes given", i, n_bytes);
      std::memcpy(&kind, sp + off + 8, 4);
      if (kind != RT_OBJECT_SPHERE && kind != RT_OBJECT_QUAD) return fail(RT_ERR_INVALID, "object %zu has unknown kind %d", i, kind);
      if (kind == RT_OBJECT_QUAD && off + RT_QUAD_RECORD_BYTES > n_bytes) return fail(RT_ERR_INVALID, "quad %zu ends beyond the %zu bytes given", i, n_bytes);
    }
    const uint8_t* r = sp + off;
    int32_t m;
    std::memcpy(&m, r + 0, 4);
    if (m < 0 || m >= n_materials)
      return fail(RT_ERR_INVALID, "sphere %zu names material %d, but only %d were uploaded", i, m, n_materials);
    sph_mat[i] = m;
    if (kind == RT_OBJECT_QUAD) {
      // corner and edges; the derived quantities in binary32, one IEEE operation each, in the order quad_dev states
      float Q[3], u[3], v[3];
      std::memcpy(Q, r + 16, 12);
      std::memcpy(u, r + 32, 12);
      std::memcpy(v, r + 48, 12);
      const float n[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
      const float nn = (n[0] * n[0] + n[1] * n[1]) + n[2] * n[2];
      const float inv_len = 1.0f / std::sqrt(nn), inv_nn = 1.0f / nn;
      const float normal[3] = {inv_len * n[0], inv_len * n[1], inv_len * n[2]};
      const float D = (normal[0] * Q[0] + normal[1] * Q[1]) + normal[2] * Q[2];
      quad_of[i] = (int32_t)quad_idx.size();
      quad_idx.push_back((int32_t)i);
      quad_qd.push_back(make_float4(Q[0], Q[1], Q[2], D));
      quad_u.push_back(make_float4(u[0], u[1], u[2], 0.0f));
      quad_v.push_back(make_float4(v[0], v[1], v[2], 0.0f));
      quad_n.push_back(make_float4(normal[0], normal[1], normal[2], 0.0f));
      quad_w.push_back(make_float4(inv_nn * n[0], inv_nn * n[1], inv_nn * n[2], 0.0f));
      // among the spheres: a record no ray hits (r * r = -infinity makes every discriminant negative)
      is_quad[i] = 1;
      sph[i] = make_float4(0.0f, 0.0f, 0.0f, -__builtin_inff());
      radius[i] = 0.0f;
      inv_r[i] = 0.0f;
      off += RT_QUAD_RECORD_BYTES;
      continue;
    }
    float rad, o[3];
    std::memcpy(&rad, r + 4, 4);
    std::memcpy(o, r + 16, 12);
    sph[i] = make_float4(o[0], o[1], o[2], rad * rad);
    radius[i] = rad;
    inv_r[i] = 1.0f / rad;
    off += RT_RECORD_BYTES;
  }
  const uint8_t* mp = static_cast<const uint8_t*>(materials_std140);
  for (size_t i = 0; i < nm; i++) {
    const uint8_t* r = mp + RT_RECORD_BYTES * i;
    int32_t type;
    float param1, alb[3];
    std::memcpy(&type, r + 4, 4);
    std::memcpy(&param1, r + 8, 4);
    std::memcpy(alb, r + 16, 12);
    mat_type[i] = type;
    mat_a[i] = make_float4(alb[0], alb[1], alb[2], param1);
  }
  HIP_TRY(c->sph.upload(sph, c->stream));
  HIP_TRY(c->inv_r.upload(inv_r, c->stream));
  HIP_TRY(c->sph_mat.upload(sph_mat, c->stream));
  HIP_TRY(c->mat_type.upload(mat_type, c->stream));
  HIP_TRY(c->mat_a.upload(mat_a, c->stream));
  c->n_quads = (int)quad_idx.size();
  HIP_TRY(c->quad_qd.upload(quad_qd, c->stream));
  HIP_TRY(c->quad_u.upload(quad_u, c->stream));
  HIP_TRY(c->quad_v.upload(quad_v, c->stream));
  HIP_TRY(c->quad_n.upload(quad_n, c->stream));
  HIP_TRY(c->quad_w.upload(quad_w, c->stream));
  HIP_TRY(c->quad_idx.upload(quad_idx, c->stream));
  if (c->n_quads > 0) {
    HIP_TRY(c->quad_of.upload(quad_of, c->stream));
  } else {
    c->quad_of.release();
  }
  c->have_noise = false;
  for (size_t i = 0; i < nm; i++) c->have_noise = c->have_noise || mat_type[i] == RT_MAT_PERLIN_LAMBERTIAN;
  if (c->have_noise) {
    rc = ensure_noise_table(c);
    if (rc) return rc;
  }

  // Hierarchy input: the few spheres far larger than the rest go to an exact
  // pre-test list, the others into the tree.
  std::vector<float4> big_sph, tree_sph;
  std::vector<int32_t> big_idx, tree_orig;
  std::vector<float> tree_r;
  c->r_max = c->r_median = c->r_min = 0.0f;
  float r_min = 3.0e38f;
  if (accel == RT_ACCEL_LBVH && ns > 0) {
    std::vector<float> absr(ns), sorted;
    for (size_t i = 0; i < ns; i++) {
      absr[i] = std::fabs(radius[i]);
      if (!is_quad[i]) sorted.push_back(absr[i]);
    }
    float median = 0.0f;
    if (!sorted.empty()) {
      std::nth_element(sorted.begin(), sorted.begin() + sorted.size() / 2, sorted.end());
      median = sorted[sorted.size() / 2];
    }
    std::vector<size_t> cand;  // quads always go into the tree
    for (size_t i = 0; i < ns; i++)
      if (!is_quad[i] && (absr[i] > kBigFactor * median || !(absr[i] < 3.0e37f))) cand.push_back(i);
    if ((int)cand.size() > kMaxBig) {  // keep only the largest ones out of the tree
      std::partial_sort(cand.begin(), cand.begin() + kMaxBig, cand.end(), [&](size_t x, size_t y) { return absr[x] > absr[y]; });
      cand.resize(kMaxBig);
      std::sort(cand.begin(), cand.end());
    }
    std::vector<char> is_big(ns, 0);
    for (size_t i : cand) is_big[i] = 1;
    std::vector<float> tree_abs;
    for (size_t i = 0; i < ns; i++) {
      if (is_big[i]) {
        big_sph.push_back(sph[i]);
        big_idx.push_back((int32_t)i);
      } else if (is_quad[i]) {
        // leaf box: the cube around the sphere through the four corners, centred between them (radius rounded up);
        // w = -infinity tells the leaf test to look the quad up
        const float4 Q = quad_qd[quad_of[i]], u = quad_u[quad_of[i]], v = quad_v[quad_of[i]];
        const double sx = (double)u.x + v.x, sy = (double)u.y + v.y, sz = (double)u.z + v.z;
        const double dx = (double)u.x - v.x, dy = (double)u.y - v.y, dz = (double)u.z - v.z;
        const double half = 0.5 * std::sqrt(std::fmax(sx * sx + sy * sy + sz * sz, dx * dx + dy * dy + dz * dz));
        const float cx = (float)(Q.x + 0.5 * sx), cy = (float)(Q.y + 0.5 * sy), cz = (float)(Q.z + 0.5 * sz);
        const float slack = 2.4e-7f * (std::fabs(cx) + std::fabs(cy) + std::fabs(cz));  // the rounding of the centre
        tree_sph.push_back(make_float4(cx, cy, cz, -__builtin_inff()));
        tree_r.push_back((float)(half * 1.000001) + slack);
        tree_orig.push_back((int32_t)i);
      } else {
        tree_sph.push_back(sph[i]);
        tree_r.push_back(radius[i]);
        tree_orig.push_back((int32_t)i);
        tree_abs.push_back(absr[i]);
        c->r_max = std::fmax(c->r_max, absr[i]);
        r_min = std::fmin(r_min, absr[i]);
      }
    }
    if (!tree_abs.empty()) {
      std::nth_element(tree_abs.begin(), tree_abs.begin() + tree_abs.size() / 2, tree_abs.end());
      c->r_median = tree_abs[tree_abs.size() / 2];
      c->r_min = r_min;
    }
  }
  HIP_TRY(c->big_sph.upload(big_sph, c->stream));
  HIP_TRY(c->big_idx.upload(big_idx, c->stream));
  HIP_TRY(c->tree_sph.upload(tree_sph, c->stream));
  HIP_TRY(c->tree_r.upload(tree_r, c->stream));
  HIP_TRY(c->tree_orig.upload(tree_orig, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));  // the host vectors go out of scope
  c->n_spheres = n_spheres;
  c->n_materials = n_materials;
  c->accel = accel;
  c->n_big = (int)big_sph.size();
  c->n_leaf = (int)tree_sph.size();
  c->n_nodes = 0;
  c->depth = 0;
  c->have_build_time = false;
  c->builder = RT_BUILDER_NONE;
  c->scanned_for_size = accel == RT_ACCEL_LBVH && c->n_quads == 0 && few_enough_to_scan(n_spheres);
  if (accel == RT_ACCEL_LBVH && c->n_leaf > 0 && !c->scanned_for_size) {
    bool built = false;
    if (host_builder_enabled() && c->n_quads == 0 && c->n_leaf >= 2 && c->n_leaf <= kSahMaxLeaves) {
      bool finite = true;
      for (size_t i = 0; i < tree_sph.size(); i++)
        finite = finite && std::isfinite(tree_sph[i].x) && std::isfinite(tree_sph[i].y) && std::isfinite(tree_sph[i].z) &&
                 std::isfinite(tree_r[i]);
      if (finite) {
        rc = build_sah_host(c, tree_sph, tree_r, tree_orig);
        if (rc) return rc;
        const int search = choose_search(c);  // kept only when the render kernels can hold it in LDS
        // (and when it is not degenerate: a chain as deep as coincident centres can make would overflow the
        // traversal stacks kept in LDS and send its rays to the scan)
        built = (search == SEARCH_TREE_LDS_RAY || search == SEARCH_TREE_LDS_NODE) && c->depth <= 2 * kSahMaxWideDepth;
      }
    }
    c->builder = RT_BUILDER_HOST_SAH;
    if (!built) {
      rc = build_lbvh(c);
      if (rc) return rc;
      c->builder = RT_BUILDER_DEVICE_LBVH;
    }
  }
  c->search = choose_search(c);
  if ((c->search == SEARCH_TREE_GLB_RAY || c->search == SEARCH_TREE_GLB_NODE || c->search == SEARCH_TREE_GLB_NODE_Q) &&
      c->n_nodes > 0) {
    HIP_TRY(c->qnodes.resize((size_t)c->n_nodes));
    lbvh_quantize_kernel<<<(unsigned)((c->n_nodes + 255) / 256), 256, 0, c->stream>>>(c->nodes.p, c->n_nodes, c->qnodes.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  if ((c->search == SEARCH_TREE_GLB_RAY || c->search == SEARCH_TREE_GLB_NODE || c->search == SEARCH_TREE_GLB_NODE_Q) && c->n_leaf > 0) {
    HIP_TRY(c->leaf_rec.resize(2 * (size_t)c->n_leaf));
    lbvh_pack_leaves_kernel<<<(unsigned)((c->n_leaf + 255) / 256), 256, 0, c->stream>>>(c->leaf_sph.p, c->leaf_idx.p, c->inv_r.p, c->n_leaf,
                                                                                        c->leaf_rec.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
  } else {
    c->leaf_rec.release();
  }
  c->have_scene = true;
  return RT_OK;
}

}  // namespace

extern "C" {

int rt_scene_upload(rt_ctx* c, const void* spheres_std140, int n_spheres, const void* materials_std140, int n_materials,
                    int accel) {
  return upload_impl(c, spheres_std140, (size_t)(n_spheres > 0 ? n_spheres : 0) * RT_RECORD_BYTES, n_spheres, false,
                     materials_std140, n_materials, accel);
}

int rt_scene_upload_objects(rt_ctx* c, const void* objects, size_t n_bytes, int n_objects, const void* materials_std140,
                            int n_materials, int accel) {
  return upload_impl(c, objects, n_bytes, n_objects, true, materials_std140, n_materials, accel);
}

int rt_render(rt_ctx* c, const rt_params* p) { return render_impl(c, p, false); }
int rt_render_counted(rt_ctx* c, const rt_params* p) { return render_impl(c, p, true); }

int rt_clear(rt_ctx* c) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (c->accum() && c->acc_w > 0)
    HIP_TRY(hipMemsetAsync(c->accum(), 0, 3 * (size_t)c->acc_w * (size_t)c->acc_h * sizeof(float), c->stream));
  c->samples_lo = c->samples_hi = 0;
  c->samples_override = -1;
  return RT_OK;
}

int rt_set_accumulator(rt_ctx* c, void* device_rgb, int width, int height) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (device_rgb && (width < 1 || height < 1)) return fail(RT_ERR_INVALID, "bad accumulator size");
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->accum_ext = static_cast<float*>(device_rgb);
  if (device_rgb) {
    c->acc_w = width;
    c->acc_h = height;
  } else {
    c->acc_w = c->acc_h = 0;  // the own buffer is re-created by the next render
  }
  c->samples_lo = c->samples_hi = 0;
  c->samples_override = -1;
  return RT_OK;
}

int rt_set_scratch_budget(rt_ctx* c, size_t split_bytes, size_t probe_bytes) {
  int rc = check_ctx(c);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->split_budget = split_bytes;
  c->probe_budget = probe_bytes;
  // what is held beyond the new budgets goes back to the device now
  if (split_bytes && c->split_samples.n * sizeof(float) > split_bytes) c->split_samples.release();
  if (probe_bytes && c->probe_samples.n * sizeof(float) > probe_bytes) c->probe_samples.release();
  return RT_OK;
}

int rt_set_accumulated_samples(rt_ctx* c, int samples) {
  if (!c) return fail(RT_ERR_INVALID, "null context");
  if (samples < 1) return fail(RT_ERR_INVALID, "samples must be positive");
  c->samples_override = samples;
  return RT_OK;
}

int rt_accumulator_read(rt_ctx* c, float* sums, int* width, int* height, int* samples) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->accum() || c->acc_w < 1) return fail(RT_ERR_STATE, "nothing rendered yet");
  if (width) *width = c->acc_w;
  if (height) *height = c->acc_h;
  if (samples) *samples = c->samples_override > 0 ? c->samples_override : c->samples_hi - c->samples_lo;
  if (sums) {
    const size_t nv = 3 * (size_t)c->acc_w * (size_t)c->acc_h;
    HIP_TRY(hipMemcpyAsync(sums, c->accum(), nv * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RT_OK;
}

int rt_accumulator_write(rt_ctx* c, const float* sums, int width, int height, int samples) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!sums || width < 1 || height < 1 || (int64_t)width * height > (int64_t)1 << 30) return fail(RT_ERR_INVALID, "bad accumulator image");
  if (samples < 0) return fail(RT_ERR_INVALID, "negative sample count");
  const size_t nv = 3 * (size_t)width * (size_t)height;
  if (c->accum_ext) {
    if (width != c->acc_w || height != c->acc_h)
      return fail(RT_ERR_STATE, "image %d x %d does not match the caller's accumulator %d x %d", width, height, c->acc_w, c->acc_h);
  } else {
    HIP_TRY(c->accum_own.resize(nv));
    c->acc_w = width;
    c->acc_h = height;
  }
  HIP_TRY(hipMemcpyAsync(c->accum(), sums, nv * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->samples_lo = 0;
  c->samples_hi = samples;
  c->samples_override = -1;
  return RT_OK;
}

int rt_render_chunk(rt_ctx* c, const rt_params* p, int c_min_x, int c_min_y, int c_max_x, int c_max_y) {
  if (!p) return fail(RT_ERR_INVALID, "null params");
  rt_params q = *p;
  q.tile_x0 = c_min_x;
  q.tile_y0 = c_min_y;
  q.tile_x1 = c_max_x;
  q.tile_y1 = c_max_y;
  return render_impl(c, &q, false);
}

int rt_readback(rt_ctx* c, float* rgb_linear, uint8_t* rgb8) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->accum() || c->acc_w < 1) return fail(RT_ERR_STATE, "nothing rendered yet");
  const int samples = c->samples_override > 0 ? c->samples_override : c->samples_hi - c->samples_lo;
  if (samples < 1) return fail(RT_ERR_STATE, "no samples accumulated");
  const size_t nv = 3 * (size_t)c->acc_w * (size_t)c->acc_h;
  if (rgb_linear) HIP_TRY(c->out_linear.resize(nv));
  if (rgb8) HIP_TRY(c->out_rgb8.resize(nv));
  if (rgb_linear || rgb8) {
    rt_resolve_kernel<<<(unsigned)((nv + 255) / 256), 256, 0, c->stream>>>(
        c->accum(), nv, 1.0f / (float)samples, rgb_linear ? c->out_linear.p : nullptr, rgb8 ? c->out_rgb8.p : nullptr);
    HIP_TRY(hipGetLastError());
    if (rgb_linear)
      HIP_TRY(hipMemcpyAsync(rgb_linear, c->out_linear.p, nv * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (rgb8) HIP_TRY(hipMemcpyAsync(rgb8, c->out_rgb8.p, nv, hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RT_OK;
}

int rt_trace_batch_device(rt_ctx* c, const float* o_xyz, const float* d_xyz, size_t n, int32_t* id, float* t, float* nrm,
                          int counted) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->have_scene) return fail(RT_ERR_STATE, "rt_trace_batch before rt_scene_upload");
  if (n == 0) return RT_OK;
  if (!o_xyz || !d_xyz || !id || !t || !nrm) return fail(RT_ERR_INVALID, "null ray or result pointer");
  if (n > ((size_t)1 << 31) * 255) return fail(RT_ERR_INVALID, "batch too large");
  trace_args a;
  a.sc = c->view();
  a.o_xyz = o_xyz;
  a.d_xyz = d_xyz;
  a.n = n;
  a.id = id;
  a.t = t;
  a.nrm = nrm;
  a.counters = nullptr;
  HIP_TRY(c->queue.resize(1));
  a.queue = c->queue.p;
  a.vote = trace_vote();
  a.leaf_vote = leaf_vote();
  if (counted) {
    HIP_TRY(c->counters.resize(sizeof(rt_counters) / sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(c->counters.p, 0, sizeof(rt_counters), c->stream));
    a.counters = c->counters.p;
  }
  HIP_TRY(hipMemsetAsync(c->queue.p, 0, sizeof(unsigned long long), c->stream));
  HIP_TRY(hipEventRecord(c->ev_k0, c->stream));
  hipError_t e;
  switch (c->search) {
    case SEARCH_TREE_LDS_RAY: e = launch_trace<SEARCH_TREE_LDS_RAY>(c, counted != 0, c->stream, a); break;
    case SEARCH_TREE_LDS_NODE: e = launch_trace<SEARCH_TREE_LDS_NODE>(c, counted != 0, c->stream, a); break;
    case SEARCH_TREE_GLB_RAY: e = launch_trace<SEARCH_TREE_GLB_RAY>(c, counted != 0, c->stream, a); break;
    case SEARCH_TREE_GLB_NODE: e = launch_trace<SEARCH_TREE_GLB_NODE>(c, counted != 0, c->stream, a); break;
    case SEARCH_TREE_GLB_NODE_Q: e = launch_trace<SEARCH_TREE_GLB_NODE_Q>(c, counted != 0, c->stream, a); break;
    case SEARCH_LINEAR_Q: e = launch_trace<SEARCH_LINEAR_Q>(c, counted != 0, c->stream, a); break;
    case SEARCH_LINEAR_LDS: e = launch_trace<SEARCH_LINEAR_LDS>(c, counted != 0, c->stream, a); break;
    default: e = launch_trace<SEARCH_LINEAR>(c, counted != 0, c->stream, a); break;
  }
  if (e != hipSuccess) return fail(RT_ERR_DEVICE, "trace launch: %s", hipGetErrorString(e));
  c->last_kernel = search_name(c->search, false);
  HIP_TRY(hipEventRecord(c->ev_k1, c->stream));
  c->have_kernel_time = true;
  return RT_OK;
}

// rt_trace_batch_device without a single allocation: queue head and traversal-stack overflow live in the device's
// copy of rt_noalloc_scratch (address looked up once by rt_init).
int rt_trace_batch_noalloc(rt_ctx* c, const float* o_xyz, const float* d_xyz, size_t n, int32_t* id, float* t, float* nrm) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->have_scene) return fail(RT_ERR_STATE, "rt_trace_batch_noalloc before rt_scene_upload");
  if (n == 0) return RT_OK;
  if (!o_xyz || !d_xyz || !id || !t || !nrm) return fail(RT_ERR_INVALID, "null ray or result pointer");
  if (n > (size_t)RT_NOALLOC_MAX_RAYS)
    return fail(RT_ERR_INVALID, "rt_trace_batch_noalloc takes at most %llu rays per call, got %zu",
                (unsigned long long)RT_NOALLOC_MAX_RAYS, n);
  unsigned char* scratch = c->device < kMaxDevices ? g_noalloc_scratch[c->device] : nullptr;
  if (!scratch) return fail(RT_ERR_STATE, "device %d has no fixed scratch (rt_init looks it up for devices below %d)", c->device, kMaxDevices);
  trace_args a;
  a.sc = c->view();
  a.o_xyz = o_xyz;
  a.d_xyz = d_xyz;
  a.n = n;
  a.id = id;
  a.t = t;
  a.nrm = nrm;
  a.counters = nullptr;
  a.queue = reinterpret_cast<unsigned long long*>(scratch);
  a.vote = trace_vote();
  a.leaf_vote = leaf_vote();
  // the array keeps what the last call left: the queue head must start at zero (the overflow slices are written
  // before they are read)
  HIP_TRY(hipMemsetAsync(scratch, 0, sizeof(unsigned long long), c->stream));
  HIP_TRY(hipEventRecord(c->ev_k0, c->stream));
  unsigned char* ovf = scratch + kNoAllocHeaderBytes;
  const size_t ovf_bytes = (size_t)RT_NOALLOC_SCRATCH_BYTES - kNoAllocHeaderBytes;
  hipError_t e;
  switch (c->search) {
    case SEARCH_TREE_LDS_RAY: e = launch_trace<SEARCH_TREE_LDS_RAY>(c, false, c->stream, a, ovf, ovf_bytes); break;
    case SEARCH_TREE_LDS_NODE: e = launch_trace<SEARCH_TREE_LDS_NODE>(c, false, c->stream, a, ovf, ovf_bytes); break;
    case SEARCH_TREE_GLB_RAY: e = launch_trace<SEARCH_TREE_GLB_RAY>(c, false, c->stream, a, ovf, ovf_bytes); break;
    case SEARCH_TREE_GLB_NODE: e = launch_trace<SEARCH_TREE_GLB_NODE>(c, false, c->stream, a, ovf, ovf_bytes); break;
    case SEARCH_TREE_GLB_NODE_Q: e = launch_trace<SEARCH_TREE_GLB_NODE_Q>(c, false, c->stream, a, ovf, ovf_bytes); break;
    case SEARCH_LINEAR_Q: e = launch_trace<SEARCH_LINEAR_Q>(c, false, c->stream, a, ovf, ovf_bytes); break;
    case SEARCH_LINEAR_LDS: e = launch_trace<SEARCH_LINEAR_LDS>(c, false, c->stream, a, ovf, ovf_bytes); break;
    default: e = launch_trace<SEARCH_LINEAR>(c, false, c->stream, a, ovf, ovf_bytes); break;
  }
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    return fail(RT_ERR_INVALID, "the hierarchy (depth %d) needs more traversal-stack overflow than the %u-byte fixed scratch "
                "holds for one workgroup per CU; use rt_trace_batch_device", c->depth, (unsigned)RT_NOALLOC_SCRATCH_BYTES);
  }
  if (e != hipSuccess) return fail(RT_ERR_DEVICE, "trace launch: %s", hipGetErrorString(e));
  c->last_kernel = search_name(c->search, false);
  HIP_TRY(hipEventRecord(c->ev_k1, c->stream));
  c->have_kernel_time = true;
  return RT_OK;
}

// Host buffers: the batch goes through the GPU in chunks, with the copy in of chunk k+1 and the copy out of chunk
// k-1 running beside the kernel of chunk k (two sets of device buffers, one stream per direction).
int rt_trace_batch(rt_ctx* c, const float* o_xyz, const float* d_xyz, size_t n, int32_t* id, float* t, float* nrm) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->have_scene) return fail(RT_ERR_STATE, "rt_trace_batch before rt_scene_upload");
  if (n == 0) return RT_OK;
  if (!o_xyz || !d_xyz || !id || !t || !nrm) return fail(RT_ERR_INVALID, "null ray or result pointer");
  constexpr size_t kChunk = (size_t)1 << 22;  // rays per chunk: 96 MB in, 80 MB out
  const size_t chunk = n < kChunk ? n : kChunk;
  if (!c->copy_in) HIP_TRY(hipStreamCreateWithFlags(&c->copy_in, hipStreamNonBlocking));
  if (!c->copy_out) HIP_TRY(hipStreamCreateWithFlags(&c->copy_out, hipStreamNonBlocking));
  for (int b = 0; b < 2; b++) {
    if (!c->ev_in[b]) HIP_TRY(hipEventCreateWithFlags(&c->ev_in[b], hipEventDisableTiming));
    if (!c->ev_run[b]) HIP_TRY(hipEventCreateWithFlags(&c->ev_run[b], hipEventDisableTiming));
    if (!c->ev_out[b]) HIP_TRY(hipEventCreateWithFlags(&c->ev_out[b], hipEventDisableTiming));
    HIP_TRY(c->tb_o[b].resize(3 * chunk));
    HIP_TRY(c->tb_d[b].resize(3 * chunk));
    HIP_TRY(c->tb_t[b].resize(chunk));
    HIP_TRY(c->tb_n[b].resize(3 * chunk));
    HIP_TRY(c->tb_id[b].resize(chunk));
  }
  const size_t n_chunks = (n + chunk - 1) / chunk;
  auto copy_out = [&](size_t k) -> hipError_t {  // results of chunk k to the caller, after its kernel
    const int b = (int)(k & 1);
    const size_t first = k * chunk, m = n - first < chunk ? n - first : chunk;
    hipError_t e = hipStreamWaitEvent(c->copy_out, c->ev_run[b], 0);
    if (e == hipSuccess) e = hipMemcpyAsync(id + first, c->tb_id[b].p, m * sizeof(int32_t), hipMemcpyDeviceToHost, c->copy_out);
    if (e == hipSuccess) e = hipMemcpyAsync(t + first, c->tb_t[b].p, m * sizeof(float), hipMemcpyDeviceToHost, c->copy_out);
    if (e == hipSuccess) e = hipMemcpyAsync(nrm + 3 * first, c->tb_n[b].p, 3 * m * sizeof(float), hipMemcpyDeviceToHost, c->copy_out);
    if (e == hipSuccess) e = hipEventRecord(c->ev_out[b], c->copy_out);
    return e;
  };
  hipError_t e = hipSuccess;
  for (size_t k = 0; k < n_chunks && e == hipSuccess && rc == RT_OK; k++) {
    const int b = (int)(k & 1);
    const size_t first = k * chunk, m = n - first < chunk ? n - first : chunk;
    // the rays of chunk k-2 were read by its kernel; its results have left the result buffers
    if (k >= 2) e = hipStreamWaitEvent(c->copy_in, c->ev_run[b], 0);
    if (e == hipSuccess) e = hipMemcpyAsync(c->tb_o[b].p, o_xyz + 3 * first, 3 * m * sizeof(float), hipMemcpyHostToDevice, c->copy_in);
    if (e == hipSuccess) e = hipMemcpyAsync(c->tb_d[b].p, d_xyz + 3 * first, 3 * m * sizeof(float), hipMemcpyHostToDevice, c->copy_in);
    if (e == hipSuccess) e = hipEventRecord(c->ev_in[b], c->copy_in);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_in[b], 0);
    if (e == hipSuccess && k >= 2) e = hipStreamWaitEvent(c->stream, c->ev_out[b], 0);
    if (e != hipSuccess) break;
    rc = rt_trace_batch_device(c, c->tb_o[b].p, c->tb_d[b].p, m, c->tb_id[b].p, c->tb_t[b].p, c->tb_n[b].p, 0);
    if (rc != RT_OK) break;
    e = hipEventRecord(c->ev_run[b], c->stream);
    // copies from pageable memory hold the calling thread: the previous chunk's results go out while this chunk's kernel runs
    if (e == hipSuccess && k >= 1) e = copy_out(k - 1);
  }
  if (e == hipSuccess && rc == RT_OK) e = copy_out(n_chunks - 1);
  const hipError_t s0 = hipStreamSynchronize(c->copy_in), s1 = hipStreamSynchronize(c->stream), s2 = hipStreamSynchronize(c->copy_out);
  if (rc != RT_OK) return rc;
  if (e == hipSuccess) e = s0 != hipSuccess ? s0 : (s1 != hipSuccess ? s1 : s2);
  if (e != hipSuccess) return fail(RT_ERR_DEVICE, "ray batch transfer: %s", hipGetErrorString(e));
  return RT_OK;
}

int rt_eval_perlin(rt_ctx* c, const float* xy, size_t n, int use_table, float* value) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (n == 0) return RT_OK;
  if (!xy || !value) return fail(RT_ERR_INVALID, "null point or result pointer");
  if (n > ((size_t)1 << 28)) return fail(RT_ERR_INVALID, "too many points");
  if (use_table) {
    rc = ensure_noise_table(c);
    if (rc) return rc;
  }
  dev_buf<float2> in;
  dev_buf<float> out;
  hipError_t e = in.resize(n);
  if (e == hipSuccess) e = out.resize(n);
  if (e == hipSuccess) e = hipMemcpyAsync(in.p, xy, n * sizeof(float2), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    perlin_table tab;
    tab.grad = use_table ? c->perlin_grad.p : nullptr;
    tab.half = kPerlinHalf;
    rt_eval_perlin_kernel<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(in.p, n, tab, out.p);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(value, out.p, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
  const hipError_t s = hipStreamSynchronize(c->stream);
  in.release();
  out.release();
  if (e == hipSuccess) e = s;
  if (e != hipSuccess) return fail(RT_ERR_DEVICE, "rt_eval_perlin: %s", hipGetErrorString(e));
  return RT_OK;
}

int rt_eval_perlin_gradient(rt_ctx* c, const int32_t* ixy, size_t n, float* gxy) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (n == 0) return RT_OK;
  if (!ixy || !gxy) return fail(RT_ERR_INVALID, "null point or result pointer");
  if (n > ((size_t)1 << 28)) return fail(RT_ERR_INVALID, "too many points");
  dev_buf<int2> in;
  dev_buf<float2> out;
  hipError_t e = in.resize(n);
  if (e == hipSuccess) e = out.resize(n);
  if (e == hipSuccess) e = hipMemcpyAsync(in.p, ixy, n * sizeof(int2), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    rt_eval_gradient_kernel<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(in.p, n, out.p);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(gxy, out.p, n * sizeof(float2), hipMemcpyDeviceToHost, c->stream);
  const hipError_t s = hipStreamSynchronize(c->stream);
  in.release();
  out.release();
  if (e == hipSuccess) e = s;
  if (e != hipSuccess) return fail(RT_ERR_DEVICE, "rt_eval_perlin_gradient: %s", hipGetErrorString(e));
  return RT_OK;
}

int rt_eval_sqrt(rt_ctx* c, const float* x, size_t n, float* y) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (n == 0) return RT_OK;
  if (!x || !y) return fail(RT_ERR_INVALID, "null argument or result pointer");
  if (n > ((size_t)1 << 28)) return fail(RT_ERR_INVALID, "too many values");
  dev_buf<float> in, out;
  hipError_t e = in.resize(n);
  if (e == hipSuccess) e = out.resize(n);
  if (e == hipSuccess) e = hipMemcpyAsync(in.p, x, n * sizeof(float), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    rt_eval_sqrt_kernel<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(in.p, n, out.p);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(y, out.p, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
  const hipError_t s = hipStreamSynchronize(c->stream);
  in.release();
  out.release();
  if (e == hipSuccess) e = s;
  if (e != hipSuccess) return fail(RT_ERR_DEVICE, "rt_eval_sqrt: %s", hipGetErrorString(e));
  return RT_OK;
}

int rt_check_sqrt(rt_ctx* c, uint64_t first_bits, uint64_t count, uint64_t* mismatches, uint32_t* example_bits) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!mismatches) return fail(RT_ERR_INVALID, "null result pointer");
  if (first_bits > 0xFFFFFFFFull || count > 0x100000000ull || first_bits + count > 0x100000000ull)
    return fail(RT_ERR_INVALID, "bit patterns beyond 32 bits");
  *mismatches = 0;
  if (example_bits) *example_bits = 0;
  if (count == 0) return RT_OK;
  dev_buf<unsigned long long> bad;
  dev_buf<uint32_t> ex;
  hipError_t e = bad.resize(1);
  if (e == hipSuccess) e = ex.resize(1);
  if (e == hipSuccess) e = hipMemsetAsync(bad.p, 0, sizeof(unsigned long long), c->stream);
  if (e == hipSuccess) e = hipMemsetAsync(ex.p, 0, sizeof(uint32_t), c->stream);
  if (e == hipSuccess) {
    const unsigned long long blocks = (count + 255) / 256;
    rt_check_sqrt_kernel<<<(unsigned)(blocks < 16384ull ? blocks : 16384ull), 256, 0, c->stream>>>(first_bits, count, bad.p, ex.p);
    e = hipGetLastError();
  }
  unsigned long long n_bad = 0;
  uint32_t one = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&n_bad, bad.p, sizeof(n_bad), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(&one, ex.p, sizeof(one), hipMemcpyDeviceToHost, c->stream);
  const hipError_t s = hipStreamSynchronize(c->stream);
  bad.release();
  ex.release();
  if (e == hipSuccess) e = s;
  if (e != hipSuccess) return fail(RT_ERR_DEVICE, "rt_check_sqrt: %s", hipGetErrorString(e));
  *mismatches = n_bad;
  if (example_bits) *example_bits = one;
  return RT_OK;
}

int rt_synchronize(rt_ctx* c) {
  int rc = check_ctx(c);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RT_OK;
}

int rt_get_counters(rt_ctx* c, rt_counters* out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!out) return fail(RT_ERR_INVALID, "null out pointer");
  if (!c->counters.p) return fail(RT_ERR_STATE, "no counted call was made");
  HIP_TRY(hipMemcpyAsync(out, c->counters.p, sizeof(rt_counters), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RT_OK;
}

int rt_last_kernel_ms(rt_ctx* c, float* ms, const char** kernel_name) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!ms) return fail(RT_ERR_INVALID, "null out pointer");
  if (!c->have_kernel_time) return fail(RT_ERR_STATE, "no kernel was launched");
  HIP_TRY(hipEventSynchronize(c->ev_k1));
  HIP_TRY(hipEventElapsedTime(ms, c->ev_k0, c->ev_k1));
  if (kernel_name) *kernel_name = c->last_kernel;
  return RT_OK;
}

int rt_last_build_ms(rt_ctx* c, float* ms) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!ms) return fail(RT_ERR_INVALID, "null out pointer");
  if (!c->have_build_time) return fail(RT_ERR_STATE, "no LBVH was built");
  HIP_TRY(hipEventSynchronize(c->ev_b1));
  HIP_TRY(hipEventElapsedTime(ms, c->ev_b0, c->ev_b1));
  return RT_OK;
}

int rt_lbvh_depth(rt_ctx* c) { return c ? c->depth : 0; }

int rt_get_scene_info(rt_ctx* c, rt_scene_info* out) {
  if (!c || !out) return fail(RT_ERR_INVALID, "null argument");
  if (!c->have_scene) return fail(RT_ERR_STATE, "no scene uploaded");
  std::memset(out, 0, sizeof(*out));
  out->n_spheres = c->n_spheres;
  out->n_materials = c->n_materials;
  out->accel = c->accel;
  out->builder = c->builder;
  out->n_big = c->n_big;
  out->n_leaf = c->n_leaf;
  out->n_nodes = c->n_nodes;
  out->depth = c->depth;
  out->tree_in_lds = c->search == SEARCH_TREE_LDS_RAY || c->search == SEARCH_TREE_LDS_NODE;
  out->pad_per_ray = c->search == SEARCH_TREE_LDS_RAY || c->search == SEARCH_TREE_GLB_RAY;
  out->node_bytes = c->n_nodes > 0 ? (out->tree_in_lds ? (int)sizeof(bvh4_node) : (int)sizeof(bvh4q_node)) : 0;
  return RT_OK;
}

}  // extern "C"

namespace {

// RCCL's entry points, looked up at first use: a caller that never sums across ranks needs no RCCL at all, and a
// process that already holds a copy (PyTorch ships its own) must not get a second one.
struct rccl_api {
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclReduce) reduce = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  bool ok = false;
};

const rccl_api& rccl() {
  static const rccl_api api = [] {
    rccl_api a;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
    if (!h) return a;
    a.get_unique_id = reinterpret_cast<decltype(a.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
    a.comm_init_rank = reinterpret_cast<decltype(a.comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
    a.reduce = reinterpret_cast<decltype(a.reduce)>(dlsym(h, "ncclReduce"));
    a.comm_destroy = reinterpret_cast<decltype(a.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
    a.error_string = reinterpret_cast<decltype(a.error_string)>(dlsym(h, "ncclGetErrorString"));
    a.ok = a.get_unique_id && a.comm_init_rank && a.reduce && a.comm_destroy && a.error_string;
    return a;
  }();
  return api;
}

int rccl_fail(const char* what, ncclResult_t r) { return fail(RT_ERR_DEVICE, "%s: %s", what, rccl().error_string(r)); }

}  // namespace

extern "C" {

int rt_comm_unique_id(void* id_out) {
  if (!id_out) return fail(RT_ERR_INVALID, "null id");
  if (!rccl().ok) return fail(RT_ERR_DEVICE, "RCCL (librccl.so.1) cannot be loaded: %s", dlerror() ? dlerror() : "symbols missing");
  ncclUniqueId id;
  const ncclResult_t r = rccl().get_unique_id(&id);
  if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
  static_assert(sizeof(id) == RT_COMM_ID_BYTES, "the id is RT_COMM_ID_BYTES long");
  std::memcpy(id_out, &id, sizeof(id));
  return RT_OK;
}

int rt_comm_init(rt_ctx* c, const void* id_bytes, int rank, int world) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!id_bytes || world < 1 || rank < 0 || rank >= world) return fail(RT_ERR_INVALID, "bad group arguments (rank %d of %d)", rank, world);
  if (c->comm) return fail(RT_ERR_STATE, "the context is in a group already");
  if (!rccl().ok) return fail(RT_ERR_DEVICE, "RCCL (librccl.so.1) cannot be loaded");
  ncclUniqueId id;
  std::memcpy(&id, id_bytes, sizeof(id));
  ncclComm_t comm = nullptr;
  const ncclResult_t r = rccl().comm_init_rank(&comm, world, id, rank);  // on the current device: check_ctx has set it
  if (r != ncclSuccess) return rccl_fail("ncclCommInitRank", r);
  c->comm = comm;
  c->comm_rank = rank;
  c->comm_world = world;
  return RT_OK;
}

int rt_comm_reduce(rt_ctx* c, int root) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->comm) return fail(RT_ERR_STATE, "rt_comm_reduce before rt_comm_init");
  if (root < 0 || root >= c->comm_world) return fail(RT_ERR_INVALID, "root %d out of range [0,%d)", root, c->comm_world);
  if (!c->accum() || c->acc_w < 1) return fail(RT_ERR_STATE, "nothing rendered yet");
  const size_t nv = 3 * (size_t)c->acc_w * (size_t)c->acc_h;
  // in place on the root; the others only send
  const ncclResult_t r = rccl().reduce(c->accum(), c->accum(), nv, ncclFloat32, ncclSum, root, static_cast<ncclComm_t>(c->comm), c->stream);
  if (r != ncclSuccess) return rccl_fail("ncclReduce", r);
  return RT_OK;
}

int rt_comm_destroy(rt_ctx* c) {
  if (!c) return fail(RT_ERR_INVALID, "null context");
  if (c->comm) {
    (void)hipStreamSynchronize(c->stream);
    (void)rccl().comm_destroy(static_cast<ncclComm_t>(c->comm));
    c->comm = nullptr;
  }
  return RT_OK;
}

int rt_get_stream(rt_ctx* c, void** hip_stream) {
  if (!c || !hip_stream) return fail(RT_ERR_INVALID, "null argument");
  *hip_stream = reinterpret_cast<void*>(c->stream);
  return RT_OK;
}

int rt_seed_texture(rt_ctx* c, int width, int height, float grid_size, float contrast, int layers, float lacunarity,
                    uint32_t* seed_tex) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (width < 1 || height < 1 || (int64_t)width * height > (int64_t)1 << 30) return fail(RT_ERR_INVALID, "bad image size %d x %d", width, height);
  if (layers < 0 || layers > 64) return fail(RT_ERR_INVALID, "bad layer count %d", layers);
  if (!(grid_size != 0.0f)) return fail(RT_ERR_INVALID, "grid size must not be zero");
  if (!seed_tex) return fail(RT_ERR_INVALID, "null output");
  const size_t n = (size_t)width * height;
  HIP_TRY(c->seed_tex.resize(n));
  rt_seed_texture_kernel<<<dim3((width + 63) / 64, (height + 3) / 4), 256, 0, c->stream>>>(width, height, grid_size, contrast, layers,
                                                                                           lacunarity, c->seed_tex.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(seed_tex, c->seed_tex.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RT_OK;
}

}  // extern "C"
