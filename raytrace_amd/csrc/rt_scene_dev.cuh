// rt_scene_dev.cuh — device views of an uploaded scene and the two closest-hit
// searches over it: the reference's linear scan and the LBVH traversal.
#pragma once

#include "rt_device.cuh"

// One LBVH node: the boxes of its two children and their references.
// A reference >= 0 is an internal node index; < 0 is a leaf, ~ref = position of
// the sphere in Morton order.  64 bytes, fetched as four 16-byte loads.
struct __attribute__((aligned(64))) bvh_node {
  float lo0[3], hi0[3];
  float lo1[3], hi1[3];
  int32_t c0, c1;
  int32_t pad[2];
};

struct scene_dev {
  int32_t n_spheres;
  int32_t n_materials;
  // upload order
  const float4* sph;       // (cx, cy, cz, r*r)
  const float* inv_r;      // 1 / r
  const int32_t* sph_mat;  // material slot
  const float4* mat_a;     // (albedo.r, albedo.g, albedo.b, param1)
  const int32_t* mat_type;
  // LBVH (null for the linear scan)
  const bvh_node* nodes;
  const float4* leaf_sph;   // Morton order: (cx, cy, cz, r*r)
  const int32_t* leaf_idx;  // Morton position -> upload index
  int32_t root;             // reference of the root
};

// ---------------------------------------------------------------------------
// Linear scan: hit_any of testFragment.fs:165-199.  The loop index is the same
// in every lane, so the sphere record is fetched once per wave through the
// scalar cache and enters the VALU as scalar operands.
// Returns t (-1 or a stuck NaN mean "no hit": the caller tests t > 0.001) and
// the index that produced it.
// ---------------------------------------------------------------------------
template <bool COUNTED>
RT_DEV void hit_linear(const scene_dev& sc, f3 o, f3 d, float& t_out, int& i_out, lane_counters& k) {
  const float a = dot3(d, d);
  float best_t = -1.0f;
  int best_i = -1;
  const int n = sc.n_spheres;
  RT_COUNT(k, sphere_tests, (uint32_t)n);
  for (int i = 0; i < n; i++) {
    const float4 s = sc.sph[i];
    float h, disc;
    sphere_disc(s.x, s.y, s.z, s.w, o, d, a, h, disc);
    if (disc >= 0.0f) {
      float t = sphere_roots<COUNTED>(h, disc, a, k);
      if (best_t < 0.0f || (t < best_t && t > 0.001f)) {  // fs:180
        best_t = t;
        best_i = i;
      }
    }
  }
  t_out = best_t;
  i_out = best_i;
}

// ---------------------------------------------------------------------------
// LBVH traversal.
//
// The search must return exactly what the linear scan returns, including hits
// that exist only because of binary32 rounding in the sphere test: the computed
// discriminant differs from the exact one by at most a*(19u*|oc|^2 + 6u*r^2),
// u = 2^-24, so a ray can "hit" a sphere whose true distance from the ray is up
// to sqrt(r^2 + 19u(|oc|^2 + r^2)).  A child box is therefore grown, per ray,
// by  pad = S * m,  m = L1 distance from the ray origin to the box's farthest
// corner (>= |oc| + r for every sphere inside), S = 1.25e-3 > sqrt(19u) =
// 1.064e-3; the slack also covers the rounding of the slab arithmetic and of
// the computed t (a few u relative).  A box is skipped only when the ray misses
// the grown box, when the grown box lies entirely before t = 0.001, or when it
// starts beyond the closest hit so far.
//
// Closest hit = minimum over (t, upload index) lexicographically, which is what
// the scan's strict "new_t < t" keeps.  Two corner cases of the scan cannot be
// decided from a pruned search — a sphere whose accepted root is exactly
// 0.001f (the scan's "t > 0.001" gate then depends on index order) and a NaN
// root — so they set `redo` and the caller repeats that ray with the scan.
// ---------------------------------------------------------------------------
#define RT_PAD_SCALE 1.25e-3f

RT_DEV bool box_test(const float* lo, const float* hi, f3 o, f3 inv_d, float best_t, float& tn_out) {
  float ax = lo[0] - o.x, bx = hi[0] - o.x;
  float ay = lo[1] - o.y, by = hi[1] - o.y;
  float az = lo[2] - o.z, bz = hi[2] - o.z;
  float m = (fmaxf(-ax, bx) + fmaxf(-ay, by)) + fmaxf(-az, bz);
  float pad = m * RT_PAD_SCALE;
  float t0x = (ax - pad) * inv_d.x, t1x = (bx + pad) * inv_d.x;
  float t0y = (ay - pad) * inv_d.y, t1y = (by + pad) * inv_d.y;
  float t0z = (az - pad) * inv_d.z, t1z = (bz + pad) * inv_d.z;
  float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
  float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
  tn_out = tn;
  return tn <= tf && tf >= 0.000999f && tn <= best_t;
}

template <bool COUNTED>
RT_DEV void leaf_test(const scene_dev& sc, int kpos, f3 o, f3 d, float a, float& best_t, int& best_i, bool& redo,
                      lane_counters& k) {
  RT_COUNT(k, sphere_tests, 1);
  const float4 s = sc.leaf_sph[kpos];
  float h, disc;
  sphere_disc(s.x, s.y, s.z, s.w, o, d, a, h, disc);
  if (disc >= 0.0f) {
    float t = sphere_roots<COUNTED>(h, disc, a, k);
    if (t != t || t == 0.001f) redo = true;
    if (t > 0.001f && t <= best_t) {
      int idx = sc.leaf_idx[kpos];
      if (t < best_t || idx < best_i || best_i < 0) {
        best_t = t;
        best_i = idx;
      }
    }
  }
}

// stack_ref / stack_tn: this lane's column of the wave's LDS stack,
// element e at [e * 64].
template <bool COUNTED>
RT_DEV void hit_bvh(const scene_dev& sc, f3 o, f3 d, int* stack_ref, float* stack_tn, float& t_out, int& i_out,
                    lane_counters& k) {
  const float a = dot3(d, d);
  if (!(a > 0.0f) || !(a < __builtin_inff())) {  // zero, NaN or overflowing direction: the scan decides
    hit_linear<COUNTED>(sc, o, d, t_out, i_out, k);
    return;
  }
  const f3 inv_d = mk3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
  float best_t = __builtin_inff();
  int best_i = -1;
  bool redo = false;
  int sp = 0;
  int cur = sc.root;
  if (cur < 0) {  // a single sphere
    leaf_test<COUNTED>(sc, ~cur, o, d, a, best_t, best_i, redo, k);
  } else {
    while (true) {
      const bvh_node* nd = sc.nodes + cur;
      const float4 q0 = reinterpret_cast<const float4*>(nd)[0];
      const float4 q1 = reinterpret_cast<const float4*>(nd)[1];
      const float4 q2 = reinterpret_cast<const float4*>(nd)[2];
      const int2 ch = reinterpret_cast<const int2*>(nd)[6];
      const float lo0[3] = {q0.x, q0.y, q0.z}, hi0[3] = {q0.w, q1.x, q1.y};
      const float lo1[3] = {q1.z, q1.w, q2.x}, hi1[3] = {q2.y, q2.z, q2.w};
      float tn0 = 0.0f, tn1 = 0.0f;
      bool go0 = false, go1 = false;
      // A leaf child is one sphere: its exact test costs no more than its box.
      if (ch.x < 0) {
        leaf_test<COUNTED>(sc, ~ch.x, o, d, a, best_t, best_i, redo, k);
      } else {
        RT_COUNT(k, node_tests, 1);
        go0 = box_test(lo0, hi0, o, inv_d, best_t, tn0);
      }
      if (ch.y < 0) {
        leaf_test<COUNTED>(sc, ~ch.y, o, d, a, best_t, best_i, redo, k);
      } else {
        RT_COUNT(k, node_tests, 1);
        go1 = box_test(lo1, hi1, o, inv_d, best_t, tn1);
      }
      // a leaf hit found just now may already rule out the sibling subtree
      go0 = go0 && tn0 <= best_t;
      go1 = go1 && tn1 <= best_t;
      if (go0 && go1) {
        const bool first0 = tn0 <= tn1;
        const int far = first0 ? ch.y : ch.x;
        const float far_tn = first0 ? tn1 : tn0;
        stack_ref[sp * 64] = far;
        stack_tn[sp * 64] = far_tn;
        sp++;
        cur = first0 ? ch.x : ch.y;
        continue;
      }
      if (go0) {
        cur = ch.x;
        continue;
      }
      if (go1) {
        cur = ch.y;
        continue;
      }
      bool found = false;
      while (sp > 0) {
        sp--;
        const float tn = stack_tn[sp * 64];
        if (tn <= best_t) {
          cur = stack_ref[sp * 64];
          found = true;
          break;
        }
      }
      if (!found) break;
    }
  }
  if (redo) {
    hit_linear<COUNTED>(sc, o, d, t_out, i_out, k);
    return;
  }
  t_out = best_i >= 0 ? best_t : -1.0f;
  i_out = best_i;
}
