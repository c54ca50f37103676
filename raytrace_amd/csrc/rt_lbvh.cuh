// rt_lbvh.cuh — GPU construction of the linear bounding volume hierarchy:
// scene bounds -> 30-bit Morton codes of the sphere centres -> radix sort ->
// Karras-style radix tree over the sorted codes -> bottom-up box fit -> top-down
// collapse into 128-byte 4-wide traversal nodes (largest child opened first).
// The reference has no counterpart (it scans every sphere,
// testFragment.fs:176-184); the hierarchy is checked against that scan.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_scene_dev.cuh"

// Order-preserving map float -> uint32 (for atomicMin / atomicMax on floats).
__device__ __forceinline__ uint32_t float_key(float f) {
  uint32_t u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k) {
  uint32_t u = k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu);
  return __uint_as_float(u);
}

// bounds[0..2] = min keys (init 0xFFFFFFFF), bounds[3..5] = max keys (init 0).
__global__ __launch_bounds__(256) void lbvh_bounds_kernel(const float4* sph, int n, uint32_t* bounds) {
  uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float4 s = sph[i];
    const uint32_t kx = float_key(s.x), ky = float_key(s.y), kz = float_key(s.z);
    mn[0] = min(mn[0], kx); mx[0] = max(mx[0], kx);
    mn[1] = min(mn[1], ky); mx[1] = max(mx[1], ky);
    mn[2] = min(mn[2], kz); mx[2] = max(mx[2], kz);
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
    for (int off = 32; off > 0; off >>= 1) {
      mn[a] = min(mn[a], (uint32_t)__shfl_xor((int)mn[a], off, 64));
      mx[a] = max(mx[a], (uint32_t)__shfl_xor((int)mx[a], off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(bounds + a, mn[a]);
      atomicMax(bounds + 3 + a, mx[a]);
    }
  }
}

__device__ __forceinline__ uint32_t expand_bits10(uint32_t v) {  // 10 bits -> every third bit
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

__global__ __launch_bounds__(256) void lbvh_morton_kernel(const float4* sph, int n, const uint32_t* bounds, uint32_t* codes,
                                                          uint32_t* index) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 s = sph[i];
  const float c[3] = {s.x, s.y, s.z};
  uint32_t q[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float lo = key_float(bounds[a]), hi = key_float(bounds[3 + a]);
    const float ext = hi - lo;
    float u = ext > 0.0f ? (c[a] - lo) / ext : 0.0f;
    int v = (int)(u * 1024.0f);
    q[a] = (uint32_t)min(max(v, 0), 1023);
  }
  codes[i] = (expand_bits10(q[0]) << 2) | (expand_bits10(q[1]) << 1) | expand_bits10(q[2]);
  index[i] = (uint32_t)i;
}

// Length of the common prefix of the keys at sorted positions i and j, the key
// being (code, position) so that equal codes still differ; -1 outside [0,n).
__device__ __forceinline__ int lbvh_delta(const uint32_t* codes, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint32_t ci = codes[i], cj = codes[j];
  if (ci != cj) return __clz((int)(ci ^ cj));
  return 32 + __clz(i ^ j);
}

// One thread per internal node i in [0, n-2]: its key range and split.
// child refs: >= 0 internal node, < 0 leaf (~ref = sorted position).
__global__ __launch_bounds__(256) void lbvh_hierarchy_kernel(const uint32_t* codes, int n, int* child0, int* child1,
                                                             int* node_parent, int* leaf_parent) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n - 1) return;
  const int dir = (lbvh_delta(codes, n, i, i + 1) - lbvh_delta(codes, n, i, i - 1)) < 0 ? -1 : 1;
  const int dmin = lbvh_delta(codes, n, i, i - dir);
  int lmax = 2;
  while (lbvh_delta(codes, n, i, i + lmax * dir) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (lbvh_delta(codes, n, i, i + (l + t) * dir) > dmin) l += t;
  const int j = i + l * dir;
  const int dnode = lbvh_delta(codes, n, i, j);
  int s = 0;
  int t = l;
  do {
    t = (t + 1) >> 1;
    if (lbvh_delta(codes, n, i, i + (s + t) * dir) > dnode) s += t;
  } while (t > 1);
  const int gamma = i + s * dir + min(dir, 0);
  const int first = min(i, j), last = max(i, j);
  const int c0 = (first == gamma) ? ~gamma : gamma;
  const int c1 = (last == gamma + 1) ? ~(gamma + 1) : gamma + 1;
  child0[i] = c0;
  child1[i] = c1;
  if (c0 < 0) leaf_parent[~c0] = i; else node_parent[c0] = i;
  if (c1 < 0) leaf_parent[~c1] = i; else node_parent[c1] = i;
  if (i == 0) node_parent[0] = -1;
}

struct box3 {
  float lo[3], hi[3];
};

__device__ __forceinline__ box3 sphere_box(const float4* leaf_sph, const float* leaf_r, int kpos) {
  const float4 s = leaf_sph[kpos];
  const float r = fabsf(leaf_r[kpos]);
  box3 b;
  b.lo[0] = s.x - r; b.lo[1] = s.y - r; b.lo[2] = s.z - r;
  b.hi[0] = s.x + r; b.hi[1] = s.y + r; b.hi[2] = s.z + r;
  return b;
}

// The radius bound a leaf brings to the box-growing rule: a sphere's radius; 0 for a quad (a leaf whose record's
// fourth word is -infinity, see quad_dev), whose box is the cube round its bounding sphere.
__device__ __forceinline__ float leaf_rmin(const float4* leaf_sph, const float* leaf_r, int kpos) {
  return leaf_sph[kpos].w == -__builtin_inff() ? 0.0f : fabsf(leaf_r[kpos]);
}

// Gather the spheres into Morton order.
__global__ __launch_bounds__(256) void lbvh_gather_kernel(const float4* sph, const float* radius, const int32_t* upload_index,
                                                          const uint32_t* sorted_index, int n, float4* leaf_sph,
                                                          float* leaf_r, int32_t* leaf_idx) {
  const int kpos = blockIdx.x * 256 + threadIdx.x;
  if (kpos >= n) return;
  const uint32_t src = sorted_index[kpos];
  leaf_sph[kpos] = sph[src];
  leaf_r[kpos] = radius[src];
  leaf_idx[kpos] = upload_index[src];
}

// One thread per leaf climbs towards the root; at each node the first arrival
// stops and the second one — which knows both children are finished — forms
// the node's box and the smallest radius below it.  node_box: 6 floats per
// internal node; node_rmin: one.  max_depth receives the largest number of
// ancestors of any leaf.
__global__ __launch_bounds__(256) void lbvh_fit_kernel(const float4* leaf_sph, const float* leaf_r, int n, const int* child0,
                                                       const int* child1, const int* node_parent, const int* leaf_parent,
                                                       float* node_box, float* node_rmin, unsigned int* arrivals,
                                                       int* max_depth) {
  const int kpos = blockIdx.x * 256 + threadIdx.x;
  if (kpos >= n) return;
  // depth of this leaf (independent of the fitting order)
  int depth = 0;
  for (int p = leaf_parent[kpos]; p >= 0; p = node_parent[p]) depth++;
  atomicMax(max_depth, depth);

  int cur = leaf_parent[kpos];
  while (cur >= 0) {
    const unsigned int old = atomicAdd(arrivals + cur, 1u);
    if (old == 0u) return;  // the sibling subtree is not finished yet
    __threadfence();        // see the boxes the sibling's thread published
    box3 b;
    float rmin = __builtin_inff();
#pragma unroll
    for (int side = 0; side < 2; side++) {
      const int c = side == 0 ? child0[cur] : child1[cur];
      box3 cb;
      if (c < 0) {
        cb = sphere_box(leaf_sph, leaf_r, ~c);
        rmin = fminf(rmin, leaf_rmin(leaf_sph, leaf_r, ~c));
      } else {
        rmin = fminf(rmin, __hip_atomic_load(node_rmin + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const float* src = node_box + 6 * (size_t)c;
#pragma unroll
        for (int a = 0; a < 3; a++) {
          cb.lo[a] = __hip_atomic_load(src + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          cb.hi[a] = __hip_atomic_load(src + 3 + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (side == 0) {
        b = cb;
      } else {
#pragma unroll
        for (int a = 0; a < 3; a++) {
          b.lo[a] = fminf(b.lo[a], cb.lo[a]);
          b.hi[a] = fmaxf(b.hi[a], cb.hi[a]);
        }
      }
    }
    float* dst = node_box + 6 * (size_t)cur;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      __hip_atomic_store(dst + a, b.lo[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(dst + 3 + a, b.hi[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __hip_atomic_store(node_rmin + cur, rmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();  // publish before announcing arrival at the parent
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    cur = node_parent[cur];
  }
}

// ---- collapse to a 4-wide tree ----------------------------------------------
// Top down, one launch per level of the 4-wide tree.  A 4-wide node starts from the two children of its binary node
// and, while it has fewer than four, replaces the internal child with the largest surface area by that child's two
// children: the boxes a ray is most likely to enter are opened at once, and nearly every node ends up with four
// children.  The 4-wide nodes made from a node's internal children get consecutive slots (one atomic per node), so
// that siblings — a ray that visits one often visits the next — share 128-byte lines, and the slots of a level
// are one range [begin, end): the host reads the counter after each launch and launches the next level over what was
// added.  (Blocks padded to even lengths, so that no pair of siblings straddles two lines, measured the same rate.)
//
// bin_of_slot[s] = the binary node that 4-wide node s is made from.
__device__ __forceinline__ float box_area(const float* b) {
  const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
  return dx * dy + dy * dz + dz * dx;
}

__device__ __forceinline__ bvh4_node empty_wide_node() {
  bvh4_node nd;
#pragma unroll
  for (int s = 0; s < 4; s++) {  // an inverted box no ray enters (tree_search::step_node), and a reference no test accepts
    nd.lox[s] = nd.loy[s] = nd.loz[s] = __builtin_inff();
    nd.hix[s] = nd.hiy[s] = nd.hiz[s] = -__builtin_inff();
    nd.rmin[s] = 0.0f;
    nd.ref[s] = RT_REF_EMPTY;
  }
  return nd;
}

__global__ __launch_bounds__(256) void lbvh_collapse_level_kernel(const float4* leaf_sph, const float* leaf_r, const int* child0,
                                                                  const int* child1, const float* node_box, const float* node_rmin,
                                                                  int n_internal, int* bin_of_slot, int begin, int end, int* counter,
                                                                  bvh4_node* nodes) {
  const int slot = begin + (int)(blockIdx.x * 256 + threadIdx.x);
  if (slot >= end) return;
  const int i = bin_of_slot[slot];
  if (i < 0 || i >= n_internal) return;  // (never: the slots of a level were all filled by the level before)
  int k0 = child0[i], k1 = child1[i], k2 = 0, k3 = 0;
  int nk = 2;
  while (nk < 4) {
    int best = -1;
    float best_area = 0.0f;
#pragma unroll
    for (int s = 0; s < 3; s++) {
      const int c = s == 0 ? k0 : s == 1 ? k1 : k2;
      if (s < nk && c >= 0) {
        const float area = box_area(node_box + 6 * (size_t)c);
        if (best < 0 || area > best_area) {
          best = s;
          best_area = area;
        }
      }
    }
    if (best < 0) break;  // leaves only
    const int c = best == 0 ? k0 : best == 1 ? k1 : k2;
    const int a = child0[c], b = child1[c];
    if (best == 0) k0 = a; else if (best == 1) k1 = a; else k2 = a;
    if (nk == 2) k2 = b; else k3 = b;
    nk++;
  }
  int m = 0;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int c = s == 0 ? k0 : s == 1 ? k1 : s == 2 ? k2 : k3;
    if (s < nk && c >= 0) m++;
  }
  int next = m > 0 ? atomicAdd(counter, m) : 0;
  bvh4_node nd = empty_wide_node();
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int c = s == 0 ? k0 : s == 1 ? k1 : s == 2 ? k2 : k3;
    if (s < nk) {
      box3 b;
      if (c < 0) {
        b = sphere_box(leaf_sph, leaf_r, ~c);
        nd.ref[s] = c;
        nd.rmin[s] = leaf_rmin(leaf_sph, leaf_r, ~c);
      } else {
        nd.rmin[s] = node_rmin[c];
#pragma unroll
        for (int a = 0; a < 3; a++) {
          b.lo[a] = node_box[6 * (size_t)c + a];
          b.hi[a] = node_box[6 * (size_t)c + 3 + a];
        }
        bin_of_slot[next] = c;
        nd.ref[s] = next++;
      }
      nd.lox[s] = b.lo[0]; nd.loy[s] = b.lo[1]; nd.loz[s] = b.lo[2];
      nd.hix[s] = b.hi[0]; nd.hiy[s] = b.hi[1]; nd.hiz[s] = b.hi[2];
    }
  }
  nodes[slot] = nd;
}

// Leaf records of trees read from HBM (scene_dev::leaf_rec): the sphere, its upload index and 1 / r side by side.
__global__ __launch_bounds__(256) void lbvh_pack_leaves_kernel(const float4* leaf_sph, const int32_t* leaf_idx, const float* inv_r, int n,
                                                              float4* leaf_rec) {
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= n) return;
  const int idx = leaf_idx[pos];
  leaf_rec[2 * (size_t)pos] = leaf_sph[pos];
  leaf_rec[2 * (size_t)pos + 1] = make_float4(__uint_as_float((uint32_t)idx), inv_r[idx], 0.0f, 0.0f);
}

// ---- 64-byte nodes for trees read from HBM ------------------------------------
// One thread per 4-wide node: the children's boxes onto a byte grid over their union (bvh4q_node).  Low planes are
// rounded down and high planes up, checked in double against the decoded value, so that the decoded box holds the
// child's box.
__global__ __launch_bounds__(256) void lbvh_quantize_kernel(const bvh4_node* nodes, int n_nodes, bvh4q_node* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_nodes) return;
  const bvh4_node nd = nodes[i];
  bvh4q_node q;
  q.exps = 0u;
  float rmin = __builtin_inff();
#pragma unroll
  for (int s = 0; s < 4; s++) {
    q.ref[s] = nd.ref[s];
    if (nd.ref[s] != RT_REF_EMPTY) rmin = fminf(rmin, fmaxf(nd.rmin[s], 0.0f));  // (a NaN radius counts as 0)
  }
  q.rmin = rmin < __builtin_inff() ? rmin : 0.0f;
  q.open = 0u;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float* lo = a == 0 ? nd.lox : a == 1 ? nd.loy : nd.loz;
    const float* hi = a == 0 ? nd.hix : a == 1 ? nd.hiy : nd.hiz;
    float ulo = __builtin_inff(), uhi = -__builtin_inff();
    bool any_nan = false;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      if (nd.ref[s] == RT_REF_EMPTY) continue;
      any_nan = any_nan || lo[s] != lo[s] || hi[s] != hi[s];
      ulo = fminf(ulo, lo[s]);
      uhi = fmaxf(uhi, hi[s]);
    }
    const double ext = (double)uhi - (double)ulo;
    int e = 0;
    bool grid = !any_nan && ulo > -3.0e38f && uhi < 3.0e38f && ext >= 0.0;
    if (grid) {
      if (ext > 0.0) {
        (void)frexp(ext / 254.0, &e);  // ext / 254 = m * 2^e with m in [0.5, 1): a step of 2^e covers the extent in 254 steps
      } else {
        e = -126;
      }
      e = e < -126 ? -126 : e;
      grid = e <= 127;
    }
    uint32_t wlo = 0u, whi = 0u;
    float org;
    if (grid) {
      org = ulo;
      const double step = ldexp(1.0, e);
#pragma unroll
      for (int s = 0; s < 4; s++) {
        int ql = 255, qh = 0;  // unused slot: inverted
        if (nd.ref[s] != RT_REF_EMPTY) {
          ql = (int)floor(((double)lo[s] - (double)org) / step);
          ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql);
          while (ql > 0 && (double)org + ql * step > (double)lo[s]) ql--;
          qh = (int)ceil(((double)hi[s] - (double)org) / step);
          qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
          while (qh < 255 && (double)org + qh * step < (double)hi[s]) qh++;
        }
        wlo |= (uint32_t)ql << (8 * s);
        whi |= (uint32_t)qh << (8 * s);
      }
    } else {  // an extent that is not finite: no grid, every child of the node passes
      org = 0.0f;
      e = 0;
      q.open = 1u;
    }
    q.org[a] = org;
    q.exps |= (uint32_t)(e + 127) << (8 * a);
    q.qlo[a] = wlo;
    q.qhi[a] = whi;
  }
  out[i] = q;
}
