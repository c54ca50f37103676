// rt_build.hip — scene upload: the std140 records flattened into device arrays, and the two builders of the 4-wide
// hierarchy (surface-area build on the host for trees that fit LDS, Morton-order LBVH on the device beyond).

#include <rocprim/device/device_radix_sort.hpp>

#include "rt_ctx.hpp"
#include "rt_lbvh.cuh"

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
  const char* e = tuning_env("RT_BUILDER");
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

// A scene of a few spheres is scanned whatever the caller asked for: a node step costs as much as a dozen sphere
// tests, and the answers are the same bit for bit.  Measured on subsets of the 487-sphere scene (tools/tiny_scene_rate.py):
// the scan is 10 % faster up to 16 spheres, level at 24, slower from 32; the 2-sphere noise scene renders 25 % faster
// and the 6-sphere default scene twice as fast.  RT_SCAN_UP_TO overrides the limit (0: a tree whenever one was asked
// for); read at every upload, so that one process can compare.
constexpr int kScanUpToSpheres = 16;
bool few_enough_to_scan(int n_spheres) {
  const char* e = tuning_env("RT_SCAN_UP_TO");
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


}  // extern "C"
