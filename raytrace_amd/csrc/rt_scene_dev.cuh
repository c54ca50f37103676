// rt_scene_dev.cuh — device views of an uploaded scene and the two closest-hit
// searches over it: the reference's linear scan and the 4-wide LBVH traversal.
#pragma once

#include "rt_device.cuh"

// One node of the 4-wide hierarchy (two levels of the binary LBVH collapsed):
// the boxes of up to four children, component-wise so that one 16-byte load
// brings the same plane of all four, and their references.
// ref >= 0: child node index; RT_REF_EMPTY: unused slot; otherwise a leaf,
// ~ref = position of the sphere in Morton order.  rmin: the smallest radius among
// the spheres below each child (the box-growing rule needs it).  128 bytes.
#define RT_REF_EMPTY ((int32_t)0x80000000)
struct __attribute__((aligned(16))) bvh4_node {
  float lox[4], loy[4], loz[4];
  float hix[4], hiy[4], hiz[4];
  int32_t ref[4];
  float rmin[4];
};

// The same node as trees read from HBM carry it: 64 bytes, what one group of four lanes fetches with one 64-byte
// access.  The children's planes are bytes on a grid laid over the box of all four: plane = org + q * 2^(e - 127)
// per axis, low planes rounded down and high planes up at build time, so that a decoded box holds the child's box
// (whose growing rule then applies unchanged: the decoded box holds every sphere below).  One rmin for the node, the
// smallest radius below any child: the PAD_NODE rule grows the four boxes by one pad, that of the grid's whole
// extent (farther corner, smaller radius than any child's own: never less than a child's own pad).  A node with
// an extent that is not finite on some axis has no grid: `open` is set and every child passes.
//   piece 0: org.x org.y org.z | exps (byte a: biased exponent of the step on axis a)
//   piece 1: qlo.x[4] | qlo.y[4] | qlo.z[4] | qhi.x[4]      (byte ch of a word: child ch)
//   piece 2: qhi.y[4] | qhi.z[4] | rmin | open
//   piece 3: ref[4]
struct __attribute__((aligned(16))) bvh4q_node {
  float org[3];
  uint32_t exps;
  uint32_t qlo[3];
  uint32_t qhi[3];
  float rmin;
  uint32_t open;
  int32_t ref[4];
};
static_assert(sizeof(bvh4q_node) == 64, "one 64-byte access per node");

struct scene_dev {
  int32_t n_spheres;    // all spheres, upload order
  int32_t n_materials;
  const float4* sph;        // (cx, cy, cz, r*r), upload order: linear scan and hit record
  const float* inv_r;       // 1 / r
  const int32_t* sph_mat;   // material slot
  const float4* mat_a;      // (albedo.r, albedo.g, albedo.b, param1)
  const int32_t* mat_type;
  perlin_table noise;       // gradient table of the noise-textured material (null without one)
  // --- quads (scenes uploaded with rt_scene_upload_objects; none otherwise) ---
  // An object that is a quad has, in the arrays above, the record (0, 0, 0, -infinity) — a sphere no ray hits —
  // 1/r = 0, its material, and its number among the quads in quad_of.
  int32_t n_quads;
  const float4* quad_qd;    // per quad: (Q, D)
  const float4* quad_u;     // (u, -)
  const float4* quad_v;     // (v, -)
  const float4* quad_n;     // (unit normal, -)
  const float4* quad_w;     // (w, -)
  const int32_t* quad_idx;  // quad -> upload index, ascending
  const int32_t* quad_of;   // upload index -> quad, or -1
  // --- hierarchy (unused by the linear scan) ---
  // Spheres much larger than the rest are kept out of the tree and tested
  // exactly for every ray: their boxes would swallow the whole tree.
  int32_t n_big;
  const float4* big_sph;    // (cx, cy, cz, r*r)
  const int32_t* big_idx;   // upload index, ascending
  int32_t n_leaf;           // spheres in the tree
  int32_t n_nodes;          // 4-wide nodes
  const bvh4_node* nodes;   // trees the kernels hold in LDS
  const bvh4q_node* qnodes; // trees read from HBM
  const float4* leaf_sph;   // Morton order
  const int32_t* leaf_idx;  // Morton position -> upload index
  // Trees read from HBM: everything a leaf test and the hit record need of a sphere in 32 bytes of one 64-byte half
  // line — (cx, cy, cz, r*r), (upload index as bits, 1/r, -, -) at [2 pos], [2 pos + 1] — where leaf_sph, leaf_idx, sph
  // and inv_r are four arrays and four fabric requests.  Null for trees held in LDS.
  const float4* leaf_rec;
  int32_t root;             // reference of the root (a leaf reference when n_leaf == 1)
  float root_lo[3], root_hi[3];  // bounds of the tree's spheres
  float r_max;              // largest radius in the tree
  float r_min;              // smallest radius in the tree
  uint32_t ref_mask;        // 2^k - 1 >= every 4-wide node index: the index bits of a traversal-stack entry
  float abs_max;            // largest |coordinate| sum of the root box corners
  // Trees read from HBM keep only the top of each lane's traversal stack in
  // LDS; deeper entries go to this per-launch buffer (ovf_cap entries per lane,
  // sized from the tree's depth so that a push can never fail).
  uint32_t* ovf;
  int32_t ovf_cap;
};

// The arrays a kernel actually reads: the HBM originals or their LDS copies.
struct scene_view {
  const bvh4_node* nodes;
  const bvh4q_node* qnodes;
  const float4* leaf_sph;
  const int32_t* leaf_idx;
  const float4* leaf_rec;
  const float4* scan_sph;   // upload-order spheres for the linear scan
  // hit record and shading, upload order
  const float4* sph;
  const float* inv_r;
  const int32_t* sph_mat;
  const float4* mat_a;
  const int32_t* mat_type;
  uint32_t* ovf_col;        // this lane's column of the stack overflow buffer: entry e at ovf_col[e * 64]
  int32_t ovf_cap;
};

// ---------------------------------------------------------------------------
// Linear scan: hit_any of testFragment.fs:165-199.  The loop index is the same
// in every lane, so a sphere record is one broadcast LDS read (or one scalar
// load when the scene is too large for LDS) per wave.
//
// Only a sphere whose discriminant is not negative can change the running hit
// (hit_sphere returns -1 otherwise, fs:161, and -1 is never accepted over a hit
// nor different from "no hit yet").  The scan therefore runs in blocks of 32
// spheres of straight-line code — discriminants only, each lane shifting the
// outcome into a bit mask — and after each block every lane evaluates the roots
// (correctly rounded sqrt and divides) of its own candidates, lowest index
// first, which applies the accept rule of fs:180 to the same spheres in the
// same order as the shader's loop.  There a sphere whose near root — estimated
// with the hardware's 1-ulp sqrt / rcp and a slack of 8u times the terms that
// were subtracted — already lies beyond the closest hit cannot satisfy
// "new_t < t" and is skipped.
// Returns t (-1 or a stuck NaN mean "no hit": the caller tests t > 0.001) and
// the index that produced it.
// ---------------------------------------------------------------------------
// The same scan one sphere at a time, for the rare ray a tree search hands back (finish()): the same accept rule
// on the same values in the same order, without the block structure and the prefetch registers of hit_linear, which
// would set the register budget of the tree kernels.
template <bool COUNTED>
RT_DEV void hit_linear_lean(const float4* sph, int n, f3 o, f3 d, float& t_out, int& i_out, lane_counters& k) {
  const float a = dot3(d, d);
  float best_t = -1.0f;
  int best_i = -1;
  RT_COUNT(k, sphere_tests, (uint32_t)n);
#pragma unroll 1
  for (int i = 0; i < n; i++) {
    const float4 s = sph[i];
    float h, disc;
    sphere_disc(s.x, s.y, s.z, s.w, o, d, a, h, disc);
    if (disc >= 0.0f) {
      const float t = sphere_roots<COUNTED>(h, disc, a, k);
      if (best_t < 0.0f || (t < best_t && t > 0.001f)) {  // fs:180
        best_t = t;
        best_i = i;
      }
    }
  }
  t_out = best_t;
  i_out = best_i;
}

template <bool COUNTED>
RT_DEV void hit_linear(const float4* sph, int n, f3 o, f3 d, float& t_out, int& i_out, lane_counters& k) {
  const float a = dot3(d, d);
  const bool can_estimate = a > 1e-30f && a < 1e30f;
  const float rcp_a = __builtin_amdgcn_rcpf(a);
  float best_t = -1.0f;
  int best_i = -1;
  RT_COUNT(k, sphere_tests, (uint32_t)n);
  uint32_t mask = 0;
  auto test = [&](const float4 s) {
    float h, disc;
    sphere_disc(s.x, s.y, s.z, s.w, o, d, a, h, disc);
    mask = (mask << 1) | (disc >= 0.0f ? 1u : 0u);
  };
  auto evaluate = [&](int i) {
    const float4 s = sph[i];
    float h, disc;
    sphere_disc(s.x, s.y, s.z, s.w, o, d, a, h, disc);  // the same operations on the same inputs: the same bits
    bool need_exact = true;
    if (can_estimate && !(best_t < 0.0f)) {
      const float sq = __builtin_amdgcn_sqrtf(disc);
      const float near_t = (h - sq) * rcp_a;
      const float slack = 1.0e-6f * ((fabsf(h) + sq) * rcp_a);
      need_exact = near_t - slack < best_t;
    }
    if (need_exact) {
      float t = sphere_roots<COUNTED>(h, disc, a, k);
      if (best_t < 0.0f || (t < best_t && t > 0.001f)) {  // fs:180
        best_t = t;
        best_i = i;
      }
    }
  };
  // Records are read four at a time, the next four before the current four are
  // tested, so that their latency hides behind the arithmetic; two register
  // sets take turns, so that nothing is copied from "next" to "current".
  const int last = n - 1;
  float4 n0, n1, n2, n3, m0, m1, m2, m3;
  if (n > 0) {
    n0 = sph[0];
    n1 = sph[min(1, last)];
    n2 = sph[min(2, last)];
    n3 = sph[min(3, last)];
  }
  for (int base = 0; base < n; base += 32) {
    const int cnt = min(32, n - base);
    mask = 0;
    int j = 0;
    for (; j + 8 <= cnt; j += 8) {
      const int nx = base + j + 4;
      m0 = sph[min(nx, last)];
      m1 = sph[min(nx + 1, last)];
      m2 = sph[min(nx + 2, last)];
      m3 = sph[min(nx + 3, last)];
      test(n0);
      test(n1);
      test(n2);
      test(n3);
      n0 = sph[min(nx + 4, last)];
      n1 = sph[min(nx + 5, last)];
      n2 = sph[min(nx + 6, last)];
      n3 = sph[min(nx + 7, last)];
      test(m0);
      test(m1);
      test(m2);
      test(m3);
    }
    if (j + 4 <= cnt) {
      const int nx = base + j + 4;
      m0 = sph[min(nx, last)];
      m1 = sph[min(nx + 1, last)];
      m2 = sph[min(nx + 2, last)];
      m3 = sph[min(nx + 3, last)];
      test(n0);
      test(n1);
      test(n2);
      test(n3);
      n0 = m0;
      n1 = m1;
      n2 = m2;
      n3 = m3;
      j += 4;
    }
    if (j < cnt) test(n0);  // the last block's tail: fewer than four left
    if (j + 1 < cnt) test(n1);
    if (j + 2 < cnt) test(n2);
    mask <<= 32 - cnt;  // sphere base + b sits at bit 31 - b
    while (mask != 0u) {
      const int b = __clz(mask);
      mask &= ~(0x80000000u >> b);
      evaluate(base + b);
    }
  }
  t_out = best_t;
  i_out = best_i;
}

RT_DEV quad_dev load_quad(const scene_dev& sc, int qi) {
  quad_dev q;
  const float4 a = sc.quad_qd[qi], u = sc.quad_u[qi], v = sc.quad_v[qi], n = sc.quad_n[qi], w = sc.quad_w[qi];
  q.Q = mk3(a.x, a.y, a.z);
  q.D = a.w;
  q.u = mk3(u.x, u.y, u.z);
  q.v = mk3(v.x, v.y, v.z);
  q.normal = mk3(n.x, n.y, n.z);
  q.w = mk3(w.x, w.y, w.z);
  return q;
}

// The quads of a scene against a result of the scan over its spheres (best_t > 0.001: a hit at sphere best_i).
// Quads are tested after all spheres and a quad replaces the result if it is nearer, or as near with a lower upload
// index: the closest object, ties to the one uploaded first, as among spheres.
RT_DEV void hit_quads_linear(const scene_dev& sc, f3 o, f3 d, float& best_t, int& best_i) {
  for (int qi = 0; qi < sc.n_quads; qi++) {
    const float t = hit_quad(load_quad(sc, qi), o, d);
    const int idx = sc.quad_idx[qi];
    if (t > 0.001f && (!(best_t > 0.001f) || t < best_t || (t == best_t && idx < best_i))) {
      best_t = t;
      best_i = idx;
    }
  }
}

// ---------------------------------------------------------------------------
// Hierarchy traversal.
//
// The search must return exactly what the linear scan returns, including hits
// that exist only because of binary32 rounding in the sphere test: the computed
// discriminant differs from the exact one by at most a*(19u*|oc|^2 + 6u*r^2),
// u = 2^-24, so a ray can "hit" a sphere whose true distance from the ray is up
// to sqrt(r^2 + E0), E0 = 19u*(|oc|^2 + r^2), and the point at the computed
// parameter lies within that same distance of the centre.  Boxes (which hold
// centre +- r of every sphere inside) are therefore grown by
//     pad = sqrt(rmin^2 + K*D^2) - rmin + L*D1,   K = 1.5625e-6 (1.38 * 19u),  L = 1e-6
// where D is the Euclidean distance from the ray origin to the farthest corner
// of the box (>= |oc| + r of every sphere the ray could still accept), D1 the
// same in the L1 norm, and rmin a lower bound of the radii inside: sqrt(r^2 + E)
// - r decreases with r, so a box of larger spheres needs far less than the
// worst case sqrt(K)*D = 1.25e-3*D of a vanishing one.  The slack K - 19u
// covers the rounding of h and of the computed t (a few u*D in position, i.e. a
// few u*D*(r + sqrt(E)) in squared distance); the linear term covers the 1-ulp
// hardware square root, the cancellation in sqrt(..) - rmin and the slab
// arithmetic.  Two rules:
//
//   PAD_RAY   one pad per ray: D from the tree's bounds, cut down to
//             1.003*(best_t*|d| + 2*r_max) once a hit at best_t is known; rmin
//             of the whole tree.  The slab test is six FMAs on per-ray
//             constants; their cancellation error (u * |coordinate|) is
//             covered by an extra 2^-22 * (|o|_1 + abs_max) in the pad.  Good
//             when the tree's extent is small against its spheres.
//   PAD_NODE  one pad per box: D to that box's farthest corner, rmin of the
//             spheres below it.  More arithmetic per box, but tight for near
//             boxes in a sparse, wide scene.
//
// A box is skipped only when the ray misses the grown box, when the grown box
// lies entirely before t = 0.001, or when it starts beyond the closest hit.
//
// Closest hit = minimum over (t, upload index) lexicographically, which is what
// the scan's strict "new_t < t" keeps.  Cases a pruned search cannot decide —
// an accepted root exactly 0.001f (the scan's "t > 0.001" gate then depends on
// index order), a NaN root, a traversal stack overflow — set `redo`, and the
// ray is repeated with the scan.
// ---------------------------------------------------------------------------
// Nodes copied into LDS sit 144 bytes apart, not 128: every lane reads its own
// node, and at 128 bytes the same field of any two nodes falls on one of only
// two bank groups (up to 8-way conflicts per 16-lane group); 144 = 36 banks
// spreads sixteen consecutive node numbers over sixteen different groups.
#define RT_LDS_NODE_STRIDE 144
#define RT_PAD_SCALE 1.25e-3f   // sqrt(RT_PAD_K): the pad per unit distance of a vanishing sphere
#define RT_PAD_K 1.5625e-6f
#define RT_PAD_LIN 1.0e-6f
#define RT_T_FLOOR 0.000999f

// The running closest hit.  best_i: upload index, -1 while there is none, RT_REDO once the search has met a case only
// the scan can decide; best_t is then -infinity, so that no box is worth a visit and no root is accepted any more and
// the search winds down by itself.  (Lane state kept in these two registers: a flag of its own would be a lane mask
// the compiler merges at every join of the wave's loop.)
#define RT_REDO (-2)
struct trav_candidate {
  float best_t;
  int best_i;
  RT_DEV void give_up() {
    best_t = -__builtin_inff();
    best_i = RT_REDO;
  }
  RT_DEV bool redo() const { return best_i == RT_REDO; }
};

// Exact test of one sphere against the running closest hit.
// Returns whether the sphere became the closest hit.
template <bool COUNTED>
RT_DEV bool consider_sphere(float4 s, int upload_index, f3 o, f3 d, float a, trav_candidate& c, lane_counters& k) {
  RT_COUNT(k, sphere_tests, 1);
  float h, disc;
  sphere_disc(s.x, s.y, s.z, s.w, o, d, a, h, disc);
  bool taken = false;
  if (disc >= 0.0f) {
    float t = sphere_roots<COUNTED>(h, disc, a, k);
    if (t != t || t == 0.001f) c.give_up();
    if (t > 0.001f && (t < c.best_t || (t == c.best_t && upload_index < c.best_i))) {
      c.best_t = t;
      c.best_i = upload_index;
      taken = true;
    }
  }
  return taken;
}

// The same for a quad leaf (trees of scenes with quads).
RT_DEV bool consider_quad(const scene_dev& sc, int upload_index, f3 o, f3 d, trav_candidate& c) {
  const float t = hit_quad(load_quad(sc, sc.quad_of[upload_index]), o, d);
  if (t > 0.001f && (t < c.best_t || (t == c.best_t && upload_index < c.best_i))) {
    c.best_t = t;
    c.best_i = upload_index;
    return true;
  }
  return false;
}

// Per-ray constants of the PAD_RAY slab test: t_lo = lo*inv_d + k_lo, t_hi = hi*inv_d + k_hi.
// With SELECT (nodes in LDS) the planes are picked by the sign of the direction instead of by
// min / max of the two parameters: k_lo then holds the constant of the plane the ray enters by,
// k_hi that of the plane it leaves by, and `sel` holds, one byte per axis, 48 where the direction
// is negative (the distance in bytes from a node's lo plane to its hi plane) and 0 elsewhere.
struct slab_consts {
  f3 inv_d, k_lo, k_hi;
  uint32_t sel;
};

RT_DEV float slab_dir(float v) { return fabsf(v) < 1e-20f ? copysignf(1e-20f, v) : v; }

// Growth of a box whose farthest corner lies mx, my, mz away along the axes and whose spheres
// have radii of at least rmin.  FMAs are fine here: box tests decide what is visited, never a value.
RT_DEV float box_pad(float mx, float my, float mz, float rmin) {
  const float d2 = __builtin_fmaf(mz, mz, __builtin_fmaf(my, my, mx * mx));
  const float s = __builtin_amdgcn_sqrtf(__builtin_fmaf(RT_PAD_K, d2, rmin * rmin));
  return __builtin_fmaf(RT_PAD_LIN, (mx + my) + mz, s - rmin);
}

template <bool SELECT>
RT_DEV void make_slab_consts(const scene_dev& sc, f3 o, f3 inv_d, float a, const trav_candidate& c, slab_consts& s) {
  const float mx = fmaxf(o.x - sc.root_lo[0], sc.root_hi[0] - o.x), my = fmaxf(o.y - sc.root_lo[1], sc.root_hi[1] - o.y),
              mz = fmaxf(o.z - sc.root_lo[2], sc.root_hi[2] - o.z);
  float d2 = __builtin_fmaf(mz, mz, __builtin_fmaf(my, my, mx * mx));
  if (c.best_i >= 0) {
    // nothing farther than the hit can still be accepted; |d| from the 1-ulp hardware sqrt: the
    // 1.003 factor absorbs it
    const float reach = 1.003f * (c.best_t * __builtin_amdgcn_sqrtf(a) + 2.0f * sc.r_max);
    d2 = fminf(d2, reach * reach);
  }
  // the L1 distance of the linear term is at most sqrt(3) times the Euclidean one
  const float grow = (__builtin_amdgcn_sqrtf(__builtin_fmaf(RT_PAD_K, d2, sc.r_min * sc.r_min)) - sc.r_min) +
                     1.7321f * RT_PAD_LIN * __builtin_amdgcn_sqrtf(d2);
  const float pad = grow + 2.384186e-7f * ((fabsf(o.x) + fabsf(o.y)) + fabsf(o.z) + sc.abs_max);
  s.inv_d = inv_d;
  s.k_lo = mk3(-(o.x + pad) * inv_d.x, -(o.y + pad) * inv_d.y, -(o.z + pad) * inv_d.z);
  s.k_hi = mk3(-(o.x - pad) * inv_d.x, -(o.y - pad) * inv_d.y, -(o.z - pad) * inv_d.z);
  s.sel = 0u;
  if (SELECT) {
    // fma is monotonic in each argument and lo <= hi, k_lo <= k_hi for a positive inv_d (the
    // reverse for a negative one): the selected parameters are bit for bit the min and the max
    const bool nx = inv_d.x < 0.0f, ny = inv_d.y < 0.0f, nz = inv_d.z < 0.0f;
    const f3 lo = s.k_lo, hi = s.k_hi;
    s.k_lo = mk3(nx ? hi.x : lo.x, ny ? hi.y : lo.y, nz ? hi.z : lo.z);
    s.k_hi = mk3(nx ? lo.x : hi.x, ny ? lo.y : hi.y, nz ? lo.z : hi.z);
    s.sel = (nx ? 48u : 0u) | (ny ? 48u << 8 : 0u) | (nz ? 48u << 16 : 0u);
  }
}

// The same interval from the entry and exit planes of a child (SELECT, see slab_consts).
RT_DEV void child_interval_selected(float nx, float ny, float nz, float fx, float fy, float fz, const slab_consts& s, float limit,
                                    float& tn, float& tf) {
  const float t0x = __builtin_fmaf(nx, s.inv_d.x, s.k_lo.x), t1x = __builtin_fmaf(fx, s.inv_d.x, s.k_hi.x);
  const float t0y = __builtin_fmaf(ny, s.inv_d.y, s.k_lo.y), t1y = __builtin_fmaf(fy, s.inv_d.y, s.k_hi.y);
  const float t0z = __builtin_fmaf(nz, s.inv_d.z, s.k_lo.z), t1z = __builtin_fmaf(fz, s.inv_d.z, s.k_hi.z);
  tn = fmaxf(fmaxf(fmaxf(t0x, t0y), t0z), RT_T_FLOOR);
  tf = fminf(fminf(fminf(t1x, t1y), t1z), limit);
}

// Parameter interval of the ray inside the grown box of one child, clipped to
// [RT_T_FLOOR, limit]: the child is worth visiting iff tn <= tf.
template <bool PAD_RAY>
RT_DEV void child_interval(float lx, float ly, float lz, float hx, float hy, float hz, float rmin, f3 o, const slab_consts& s,
                           float limit, float& tn, float& tf) {
  float t0x, t1x, t0y, t1y, t0z, t1z;
  if (PAD_RAY) {
    t0x = __builtin_fmaf(lx, s.inv_d.x, s.k_lo.x); t1x = __builtin_fmaf(hx, s.inv_d.x, s.k_hi.x);
    t0y = __builtin_fmaf(ly, s.inv_d.y, s.k_lo.y); t1y = __builtin_fmaf(hy, s.inv_d.y, s.k_hi.y);
    t0z = __builtin_fmaf(lz, s.inv_d.z, s.k_lo.z); t1z = __builtin_fmaf(hz, s.inv_d.z, s.k_hi.z);
  } else {
    const float ax = lx - o.x, bx = hx - o.x, ay = ly - o.y, by = hy - o.y, az = lz - o.z, bz = hz - o.z;
    const float pad = box_pad(fmaxf(-ax, bx), fmaxf(-ay, by), fmaxf(-az, bz), rmin);
    t0x = (ax - pad) * s.inv_d.x; t1x = (bx + pad) * s.inv_d.x;
    t0y = (ay - pad) * s.inv_d.y; t1y = (by + pad) * s.inv_d.y;
    t0z = (az - pad) * s.inv_d.z; t1z = (bz + pad) * s.inv_d.z;
  }
  tn = fmaxf(fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z)), RT_T_FLOOR);
  tf = fminf(fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z)), limit);
}

RT_DEV float byte_of(uint32_t w, int ch) { return (float)((w >> (8 * ch)) & 255u); }

// Node fetch for trees read from HBM.  A lane that loads its own 128-byte node with eight 16-byte
// loads costs the CU's vector cache eight tag look-ups per lane, 512 per wave and step, and that
// look-up rate (about one per clock) is what bounds an incoherent closest-hit batch.  So the first
// RT_STAGE_PIECES 16-byte pieces of every node are fetched by a group of as many lanes together —
// in pass p the lanes of a group read consecutive pieces of the node wanted by the group's lane p,
// one request per node instead of one per piece — with loads that write LDS directly: row p of the
// wave's staging area then holds, group after group, what the lanes p of the groups want.  The
// remaining pieces each lane loads itself (staging whole nodes would need 8 KB of LDS per wave
// and cost more in occupancy than it saves in look-ups).  Every lane of the wave must call this
// (my_node < 0: nothing wanted); returns where the staged part of the lane's node lies.
// The staging area and the returned place are byte offsets into the kernel's dynamic LDS.
#ifndef RT_LDS_LEAF_STEPS
#define RT_LDS_LEAF_STEPS 0   // 1: trees in LDS queue their leaves for voted leaf steps too (tuning builds)
#endif
#ifndef RT_STAGE_PIECES
#define RT_STAGE_PIECES 4
#endif
#ifndef RT_STAGE_DMA
#define RT_STAGE_DMA 1
#endif
#define RT_STAGE_ROW 1040   // 64 lanes x 16 bytes + 16: the rows start in different banks
#define RT_STAGE_BYTES (RT_STAGE_PIECES * RT_STAGE_ROW + (RT_STAGE_PIECES == 4 ? 0 : 256))  // (+ the wanted list of groups of eight)
extern __shared__ __attribute__((aligned(16))) unsigned char rt_lds[];
RT_DEV uint32_t stage_nodes(const bvh4q_node* nodes, int my_node, uint32_t stage_offset) {
  constexpr int G = RT_STAGE_PIECES;  // lanes of a group = passes = staged pieces
  static_assert(G == 4 || G == 8, "a group reads 64 or 128 consecutive bytes");
  unsigned char* stage = rt_lds + stage_offset;
  const int lane = threadIdx.x & 63, sub = lane & (G - 1);
  // what the lanes of this lane's group want: for groups of four a quad broadcast per pass (no LDS round trip)
  int w[G];
  if (G == 4) {
    w[0] = __builtin_amdgcn_mov_dpp(my_node, 0x00, 0xF, 0xF, true);  // quad_perm [0,0,0,0]
    w[1] = __builtin_amdgcn_mov_dpp(my_node, 0x55, 0xF, 0xF, true);  // [1,1,1,1]
    w[2] = __builtin_amdgcn_mov_dpp(my_node, 0xAA, 0xF, 0xF, true);  // [2,2,2,2]
    w[3] = __builtin_amdgcn_mov_dpp(my_node, 0xFF, 0xF, 0xF, true);  // [3,3,3,3]
  } else {
    int* wanted = reinterpret_cast<int*>(stage + G * RT_STAGE_ROW);
    wanted[lane] = my_node;
    __builtin_amdgcn_wave_barrier();  // LDS operations of one wave execute in order
    const int4 w0 = *reinterpret_cast<const int4*>(wanted + (lane & ~(G - 1)));
    w[0] = w0.x; w[1] = w0.y; w[2] = w0.z; w[3] = w0.w;
    const int4 w1 = *reinterpret_cast<const int4*>(wanted + (lane & ~(G - 1)) + 4);
    w[G - 4] = w1.x; w[G - 3] = w1.y; w[G - 2] = w1.z; w[G - 1] = w1.w;
  }
  const unsigned char* base = reinterpret_cast<const unsigned char*>(nodes) + sub * 16;
#if RT_STAGE_DMA
  // loads that write LDS directly: each lane gives its own source address; the 64 x 16 bytes land
  // side by side from the row start
#pragma unroll
  for (int p = 0; p < G; p++)  // (a pass nobody wants reads node 0: cheaper than a branch per pass)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)max(w[p], 0) * sizeof(bvh4q_node)),
                                     (__attribute__((address_space(3))) void*)(stage + p * RT_STAGE_ROW), 16, 0, 0);
#else
  // through registers: all loads first (a pass nobody wants reads node 0), then the stores
  float4 piece[G];
#pragma unroll
  for (int p = 0; p < G; p++) piece[p] = *reinterpret_cast<const float4*>(base + (size_t)max(w[p], 0) * sizeof(bvh4q_node));
#pragma unroll
  for (int p = 0; p < G; p++) *reinterpret_cast<float4*>(stage + p * RT_STAGE_ROW + lane * 16) = piece[p];
#endif
  return stage_offset + (uint32_t)(sub * RT_STAGE_ROW + (lane / G) * (16 * G));
}
// Before the staged part is read.
RT_DEV void stage_wait() {
#if RT_STAGE_DMA
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  __builtin_amdgcn_wave_barrier();
}

// Traversal stack entry: one word, the entry parameter (a positive float, so its bits order as
// an unsigned integer) with its low bits replaced by the node index.  `mask` covers the index:
// 12 bits for trees in LDS (at most 4096 nodes), as many as the node count needs otherwise.
// The truncated parameter is never above the true one, which keeps "entry <= closest hit" a
// conservative test; with fewer parameter bits the visiting order gets coarser, not the result.
template <bool PACKED>
struct trav_stack {
  uint32_t* col;  // this lane's column: element e at col[e * 64]
  uint32_t mask;
  RT_DEV uint32_t key(float tn, int ref) const { return (__float_as_uint(tn) & ~mask) | (uint32_t)ref; }
  RT_DEV float key_tn(uint32_t k) const { return __uint_as_float(k & ~mask); }
  RT_DEV int key_ref(uint32_t k) const { return (int)(k & mask); }
};

// Leaves whose box a node step found worth testing wait here until the wave
// takes a leaf step: Morton positions, six of 10 bits each (stored +1, zero = empty slot) in one
// word when the tree is the LDS-resident kind (at most 1022 leaves), RT_LEAF_ROWS words in LDS otherwise.
#define RT_LEAF_ROWS 5
template <bool PACKED>
struct leaf_queue;

template <>
struct leaf_queue<true> {
  static constexpr int max_leaves = 1022;
  unsigned long long q;
  RT_DEV void clear() { q = 0ull; }
  RT_DEV bool empty() const { return q == 0ull; }
  RT_DEV bool room_for_node() const { return (q >> 20) == 0ull; }  // at most two waiting: four more fit
  RT_DEV void push_if(uint32_t*, bool leaf, int pos) {
    if (leaf) q = (q << 10) | (unsigned long long)(pos + 1);
  }
  RT_DEV int pop(const uint32_t*) {
    const int pos = (int)(q & 1023ull) - 1;
    q >>= 10;
    return pos;
  }
};

template <>
struct leaf_queue<false> {
  // Trees read from HBM: the waiting leaves lie in rows of the lane's LDS column above its stack rows (`rows`), and
  // only their number lives in a register.  A push is a store and an add with carry whatever the lane's case — the
  // slot above the last leaf takes what a lane has no leaf for — where a queue shifted through registers cost a
  // branch and eight moves per child of every node.
  static constexpr int cap = RT_LEAF_ROWS;
  int n;
  RT_DEV void clear() { n = 0; }
  RT_DEV bool empty() const { return n == 0; }
  RT_DEV bool room_for_node() const { return n <= cap - 4; }
  RT_DEV void push_if(uint32_t* rows, bool leaf, int pos) {
    rows[n * 64] = (uint32_t)pos;
    n += leaf ? 1 : 0;
  }
  RT_DEV int pop(const uint32_t* rows) {
    n--;
    return (int)rows[n * 64];
  }
};

// The reference of the lowest child named in a mask of children (three selects, not a branch per child).
RT_DEV int pick_ref(int4 ref, unsigned mask) {
  const int ch = __builtin_ctz(mask | 8u);
  int r = ref.x;
  r = ch == 1 ? ref.y : r;
  r = ch == 2 ? ref.z : r;
  r = ch == 3 ? ref.w : r;
  return r;
}

// A closest-hit search that can be advanced one node at a time, so that a wave
// can interleave the traversal steps of some lanes with other work of others.
//   begin()  exact tests of the spheres kept out of the tree, per-ray constants;
//            returns true when step() must be called
//   step_node()  one 4-wide node: four box tests, leaves among the children
//                queued, ordered descent
//   step_leaf()  exact test of one queued leaf
//   active()     false when the search is over
//   finish()     the scan's answer (repeats the ray with the scan when flagged)
template <bool PAD_RAY, bool PACKED, int STACK, bool COUNTED, bool QUADS = false>
struct tree_search {
  f3 o, d;
  float a;
  slab_consts sl;
  trav_candidate c;
  int cur, sp;  // cur: node to test next, -1 when the stack has run dry
  int best_pos; // trees read from HBM: the closest hit's place in leaf_rec, -1 while it is none of the tree's spheres
  static constexpr bool LEAF_INLINE = PACKED && !RT_LDS_LEAF_STEPS;
  static constexpr bool SELECT = PAD_RAY && PACKED;  // planes picked by direction sign (slab_consts)
  leaf_queue<PACKED> lq;

  RT_DEV bool has_node() const { return cur >= 0; }
  RT_DEV bool node_ready() const { return cur >= 0 && (LEAF_INLINE || lq.room_for_node()); }
  RT_DEV bool has_leaf() const { return !LEAF_INLINE && !lq.empty(); }
  RT_DEV bool active() const { return cur >= 0 || has_leaf(); }

  RT_DEV bool begin(const scene_dev& sc, const scene_view& sv, f3 origin, f3 dir, lane_counters& k) {
    o = origin;
    d = dir;
    a = dot3(d, d);
    c.best_t = __builtin_inff();
    c.best_i = -1;
    sp = 0;
    cur = -1;
    best_pos = -1;
    if (!LEAF_INLINE) lq.clear();
    if (!(a > 0.0f) || !(a < __builtin_inff())) {  // zero, NaN or overflowing direction: the scan decides
      c.give_up();
      return false;
    }
    // spheres kept out of the tree: exact tests, wave-uniform index
    for (int j = 0; j < sc.n_big; j++) consider_sphere<COUNTED>(sc.big_sph[j], sc.big_idx[j], o, d, a, c, k);
    if (sc.n_leaf <= 0) return false;
    // Box tests only: a direction component closer to zero than 1e-20 is
    // treated as +-1e-20, which keeps every slab parameter finite (a ray
    // that flat is inside or outside the slab for its whole length either way).
    sl.inv_d = mk3(__builtin_amdgcn_rcpf(slab_dir(d.x)), __builtin_amdgcn_rcpf(slab_dir(d.y)),
                   __builtin_amdgcn_rcpf(slab_dir(d.z)));
    if (PAD_RAY) make_slab_consts<SELECT>(sc, o, sl.inv_d, a, c, sl);
    if (SELECT) {
      // the inverted boxes of unused slots stay inverted only under finite slab constants (an origin some 1e18
      // away along a nearly flat direction overflows them): the scan decides such a ray
      const float sum = ((fabsf(sl.k_lo.x) + fabsf(sl.k_lo.y)) + fabsf(sl.k_lo.z)) + ((fabsf(sl.k_hi.x) + fabsf(sl.k_hi.y)) + fabsf(sl.k_hi.z));
      if (!(sum < __builtin_inff())) {
        c.give_up();
        return false;
      }
    }
    if (sc.root < 0) {  // a single leaf in the tree
      consider_leaf(sc, sv, ~sc.root, k);
      return false;
    }
    cur = sc.root;
    return true;
  }

  // Exact test of the leaf at a position of the tree's order: a sphere, or (scenes with quads) a quad.
  RT_DEV void consider_leaf(const scene_dev& sc, const scene_view& sv, int pos, lane_counters& k) {
    if (!PACKED) {  // one 32-byte record: sphere, upload index, 1 / r
      const float4 rec = sv.leaf_rec[2 * (size_t)pos], aux = sv.leaf_rec[2 * (size_t)pos + 1];
      const int idx = (int)__float_as_uint(aux.x);
      bool taken;
      if (QUADS && rec.w == -__builtin_inff()) {
        RT_COUNT(k, sphere_tests, 1);
        taken = consider_quad(sc, idx, o, d, c);
      } else {
        taken = consider_sphere<COUNTED>(rec, idx, o, d, a, c, k);
      }
      if (taken) best_pos = pos;
      return;
    }
    const float4 rec = sv.leaf_sph[pos];
    if (QUADS && rec.w == -__builtin_inff()) {
      RT_COUNT(k, sphere_tests, 1);
      consider_quad(sc, sv.leaf_idx[pos], o, d, c);
    } else {
      consider_sphere<COUNTED>(rec, sv.leaf_idx[pos], o, d, a, c, k);
    }
  }
  // `leaf_rows`: the lane's queue rows (leaf_queue; trees read from HBM)
  RT_DEV void step_leaf(const scene_dev& sc, const scene_view& sv, const uint32_t* leaf_rows, lane_counters& k) {
    consider_leaf(sc, sv, lq.pop(leaf_rows), k);
  }

  // `staged`: where stage_nodes() put this lane's node (trees read from HBM; unused when the
  // whole tree is in LDS).
  RT_DEV void step_node(const scene_view& sv, trav_stack<PACKED> stk, uint32_t staged, lane_counters& k) {
    const unsigned char* nd = rt_lds + staged;
    if (PACKED) nd = reinterpret_cast<const unsigned char*>(sv.nodes) + __umul24((unsigned)cur, (unsigned)RT_LDS_NODE_STRIDE);  // (full-rate multiply)
    int4 ref;
    float tn[4], tf[4];
    if (!PACKED) {
      // trees read from HBM: the whole 64-byte node lies in the wave's staging area (stage_nodes)
      stage_wait();
      const uint4 p0 = *reinterpret_cast<const uint4*>(nd), p1 = *reinterpret_cast<const uint4*>(nd + 16),
                  p2 = *reinterpret_cast<const uint4*>(nd + 32);
      ref = *reinterpret_cast<const int4*>(nd + 48);
      const float sx = __uint_as_float((p0.w & 255u) << 23), sy = __uint_as_float(((p0.w >> 8) & 255u) << 23),
                  sz = __uint_as_float(((p0.w >> 16) & 255u) << 23);
      const float gx = __uint_as_float(p0.x), gy = __uint_as_float(p0.y), gz = __uint_as_float(p0.z);
      if (PAD_RAY) {
#pragma unroll
        for (int ch = 0; ch < 4; ch++)
          child_interval<true>(__builtin_fmaf(byte_of(p1.x, ch), sx, gx), __builtin_fmaf(byte_of(p1.y, ch), sy, gy),
                               __builtin_fmaf(byte_of(p1.z, ch), sz, gz), __builtin_fmaf(byte_of(p1.w, ch), sx, gx),
                               __builtin_fmaf(byte_of(p2.x, ch), sy, gy), __builtin_fmaf(byte_of(p2.y, ch), sz, gz), 0.0f, o, sl, c.best_t,
                               tn[ch], tf[ch]);
      } else {
        // One pad for the node: farthest corner of the grid's extent (org .. org + 255 steps), the node's rmin.  Folded
        // with the slab arithmetic into per-node constants, a child's plane parameter is one FMA on its byte:
        // t = q * (step / d) + (org - o -+ pad) / d.  (The rounding of these products, a few 1e-7 of the distances
        // involved, is what the pad's linear term is for.)
        const float rx = gx - o.x, ry = gy - o.y, rz = gz - o.z;
        const float pad = box_pad(fmaxf(-rx, __builtin_fmaf(255.0f, sx, rx)), fmaxf(-ry, __builtin_fmaf(255.0f, sy, ry)),
                                  fmaxf(-rz, __builtin_fmaf(255.0f, sz, rz)), __uint_as_float(p2.z));
        const float kx = sx * sl.inv_d.x, ky = sy * sl.inv_d.y, kz = sz * sl.inv_d.z;
        const float ax = (rx - pad) * sl.inv_d.x, ay = (ry - pad) * sl.inv_d.y, az = (rz - pad) * sl.inv_d.z;
        const float bx = (rx + pad) * sl.inv_d.x, by = (ry + pad) * sl.inv_d.y, bz = (rz + pad) * sl.inv_d.z;
        // Planes picked by the sign of the direction: with a negative 1/d the high plane is the one the ray enters by,
        // and (fma being monotonic in each argument, low bytes <= high bytes, a <= b exactly when 1/d is positive) the
        // picked parameters are bit for bit the min and the max of the two — twelve selects per node for 24 min / max.
        const bool ngx = sl.inv_d.x < 0.0f, ngy = sl.inv_d.y < 0.0f, ngz = sl.inv_d.z < 0.0f;
        const uint32_t wnx = ngx ? p1.w : p1.x, wfx = ngx ? p1.x : p1.w, wny = ngy ? p2.x : p1.y, wfy = ngy ? p1.y : p2.x,
                       wnz = ngz ? p2.y : p1.z, wfz = ngz ? p1.z : p2.y;
        const float cnx = ngx ? bx : ax, cfx = ngx ? ax : bx, cny = ngy ? by : ay, cfy = ngy ? ay : by, cnz = ngz ? bz : az,
                    cfz = ngz ? az : bz;
#pragma unroll
        for (int ch = 0; ch < 4; ch++) {
          const float t0x = __builtin_fmaf(byte_of(wnx, ch), kx, cnx), t1x = __builtin_fmaf(byte_of(wfx, ch), kx, cfx);
          const float t0y = __builtin_fmaf(byte_of(wny, ch), ky, cny), t1y = __builtin_fmaf(byte_of(wfy, ch), ky, cfy);
          const float t0z = __builtin_fmaf(byte_of(wnz, ch), kz, cnz), t1z = __builtin_fmaf(byte_of(wfz, ch), kz, cfz);
          tn[ch] = fmaxf(fmaxf(fmaxf(t0x, t0y), t0z), RT_T_FLOOR);
          tf[ch] = fminf(fminf(fminf(t1x, t1y), t1z), c.best_t);
        }
      }
    } else {
      ref = reinterpret_cast<const int4*>(nd)[6];
    }
    const int refs[4] = {ref.x, ref.y, ref.z, ref.w};
    if (!PACKED) {
      // a node with an extent that is not finite has no grid (bvh4q_node::open): all its children pass
      if (reinterpret_cast<const uint32_t*>(nd)[11] != 0u) {
#pragma unroll
        for (int ch = 0; ch < 4; ch++) {
          tn[ch] = RT_T_FLOOR;
          tf[ch] = c.best_t;
        }
      }
    } else if (SELECT) {
      const uint32_t sx = sl.sel & 0xFFu, sy = (sl.sel >> 8) & 0xFFu, sz = (sl.sel >> 16) & 0xFFu;
      const float4 nx = *reinterpret_cast<const float4*>(nd + sx);
      const float4 ny = *reinterpret_cast<const float4*>(nd + 16 + sy);
      const float4 nz = *reinterpret_cast<const float4*>(nd + 32 + sz);
      const float4 fx = *reinterpret_cast<const float4*>(nd + 48 - sx);
      const float4 fy = *reinterpret_cast<const float4*>(nd + 64 - sy);
      const float4 fz = *reinterpret_cast<const float4*>(nd + 80 - sz);
      child_interval_selected(nx.x, ny.x, nz.x, fx.x, fy.x, fz.x, sl, c.best_t, tn[0], tf[0]);
      child_interval_selected(nx.y, ny.y, nz.y, fx.y, fy.y, fz.y, sl, c.best_t, tn[1], tf[1]);
      child_interval_selected(nx.z, ny.z, nz.z, fx.z, fy.z, fz.z, sl, c.best_t, tn[2], tf[2]);
      child_interval_selected(nx.w, ny.w, nz.w, fx.w, fy.w, fz.w, sl, c.best_t, tn[3], tf[3]);
    } else {  // trees in LDS with one pad per child box
      const float4 lox = reinterpret_cast<const float4*>(nd)[0], loy = reinterpret_cast<const float4*>(nd)[1],
                   loz = reinterpret_cast<const float4*>(nd)[2], hix = reinterpret_cast<const float4*>(nd)[3],
                   hiy = reinterpret_cast<const float4*>(nd)[4], hiz = reinterpret_cast<const float4*>(nd)[5];
      float4 rmin = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (!PAD_RAY) rmin = reinterpret_cast<const float4*>(nd)[7];
      child_interval<PAD_RAY>(lox.x, loy.x, loz.x, hix.x, hiy.x, hiz.x, rmin.x, o, sl, c.best_t, tn[0], tf[0]);
      child_interval<PAD_RAY>(lox.y, loy.y, loz.y, hix.y, hiy.y, hiz.y, rmin.y, o, sl, c.best_t, tn[1], tf[1]);
      child_interval<PAD_RAY>(lox.z, loy.z, loz.z, hix.z, hiy.z, hiz.z, rmin.z, o, sl, c.best_t, tn[2], tf[2]);
      child_interval<PAD_RAY>(lox.w, loy.w, loz.w, hix.w, hiy.w, hiz.w, rmin.w, o, sl, c.best_t, tn[3], tf[3]);
    }
    bool ok[4];
    unsigned leaf_mask = 0;
#pragma unroll
    for (int ch = 0; ch < 4; ch++) {
      // (an unused slot holds an inverted box, lo = +inf and hi = -inf: the entry plane picked by the direction's sign
      // then gives tn = +inf and the exit plane tf = -inf on every axis, so the test by selected planes needs no look at
      // the reference; min / max of the two parameters would undo the inversion)
      ok[ch] = SELECT ? tn[ch] <= tf[ch] : bool((tn[ch] <= tf[ch]) & (refs[ch] != RT_REF_EMPTY));
      RT_COUNT(k, node_tests, refs[ch] != RT_REF_EMPTY ? 1u : 0u);
      if (LEAF_INLINE) {
        leaf_mask |= (ok[ch] & (refs[ch] < 0)) ? (1u << ch) : 0u;  // a select on combined lane masks, not a branch per child
      } else {
        lq.push_if(stk.col + (STACK + 1) * 64, ok[ch] & (refs[ch] < 0), ~refs[ch]);
      }
    }
    // LDS-resident scenes test the leaves here and now (an exact hit shortens the ray for what is popped later, and a
    // leaf costs less than another trip round the wave's loop); scenes read from HBM queue them for a leaf step, so
    // that the wave waits for leaf records once per step, not once per node.  The record of the first leaf is
    // asked for before the children are sorted and pushed, so that the read is under way while that happens.
#define leaf_ref(m) pick_ref(ref, (m))
    int pos0 = 0;
    float4 sph0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (LEAF_INLINE) {
      int first = -1;  // the reference of the lowest leaf among the children worth a visit: selects on the masks just formed
#pragma unroll
      for (int ch = 3; ch >= 0; ch--) first = (ok[ch] & (refs[ch] < 0)) ? refs[ch] : first;
      pos0 = ~first;  // (no such leaf: position 0, read and not used)
      sph0 = sv.leaf_sph[pos0];
    }
    // Children worth a visit, as stack keys; zero = none.  Sorted descending, the valid keys come first, farthest
    // first, and the nearest valid one is last.
    uint32_t key[4];
#pragma unroll
    for (int ch = 0; ch < 4; ch++) key[ch] = (ok[ch] & (refs[ch] >= 0)) ? stk.key(tn[ch], refs[ch]) : 0u;
#define RT_CE(i, j)                               \
  do {                                            \
    const uint32_t hi_ = max(key[i], key[j]);     \
    key[j] = min(key[i], key[j]);                 \
    key[i] = hi_;                                 \
  } while (0)
    RT_CE(0, 1); RT_CE(2, 3); RT_CE(0, 2); RT_CE(1, 3); RT_CE(1, 2);
#undef RT_CE
    // The nearest child is next; the others go on the stack, farthest first, so that the nearest of them is popped
    // first.  key[e] of cnt valid keys belongs at row sp + e for e < cnt - 1: key[0..2] are stored at rows sp .. sp + 2
    // whatever cnt is — what lands beyond the new top is never read — so the stores need no index arithmetic (trees
    // in LDS keep three rows of slack above the last row).  Trees read from HBM keep the entries beyond the LDS rows in
    // their overflow buffer; for trees in LDS an overflow sends the ray to the scan.
    const uint32_t pushed = min(key[1], 1u) + min(key[2], 1u) + min(key[3], 1u);  // valid keys but one (every valid key is > 1)
    if (PACKED) {
      uint32_t* top = stk.col + sp * 64;
      top[0] = key[0];
      top[64] = key[1];
      top[128] = key[2];
      sp += (int)pushed;
      if (sp > STACK) {  // (rare: a chain deeper than the rows)
        c.give_up();
        sp = 0;
        key[0] = key[1] = key[2] = key[3] = 0u;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 3; e++) {
        const int row = sp + e;
        stk.col[min(row, STACK) * 64] = key[e];  // row STACK: the sink of what does not fit the LDS rows
        if (e < (int)pushed && row >= STACK) {
          if (row - STACK < sv.ovf_cap) sv.ovf_col[(size_t)(row - STACK) * 64] = key[e];
          else c.give_up();
        }
      }
      const int grown = sp + (int)pushed, room = STACK + sv.ovf_cap;
      sp = min(grown, room);
    }
    uint32_t next = key[0];  // the last valid key
    next = key[1] ? key[1] : next;
    next = key[2] ? key[2] : next;
    next = key[3] ? key[3] : next;
    if (LEAF_INLINE && leaf_mask) {
      consider_sphere<COUNTED>(sph0, sv.leaf_idx[pos0], o, d, a, c, k);
      leaf_mask &= leaf_mask - 1;
      while (leaf_mask) {
        const int pos = ~leaf_ref(leaf_mask);
        leaf_mask &= leaf_mask - 1;
        consider_sphere<COUNTED>(sv.leaf_sph[pos], sv.leaf_idx[pos], o, d, a, c, k);
      }
      if (!(stk.key_tn(next) <= c.best_t)) next = 0u;  // a leaf hit nearer than the nearest child: the stack decides
    }
#undef leaf_ref
    if (next != 0u) {
      cur = stk.key_ref(next);
      return;
    }
    cur = -1;
    while (sp > 0) {
      sp--;
      // The LDS row is read unconditionally (row clamped) and replaced by the overflow row where
      // the entry lives there: a pointer chosen between the two memories would need flat loads.
      uint32_t e = stk.col[min(sp, STACK - 1) * 64];
      // (the LDS read stays an LDS read: left to the compiler, the two reads become one flat load from a selected address,
      // which waits on both memories' counters)
      if (!PACKED) asm volatile("" : "+v"(e));
      if (!PACKED && sp >= STACK) e = sv.ovf_col[(size_t)(sp - STACK) * 64];
      if (stk.key_tn(e) <= c.best_t) {
        cur = stk.key_ref(e);
        return;
      }
    }
  }

  RT_DEV void finish(const scene_dev& sc, const scene_view& sv, float& t_out, int& i_out, lane_counters& k) {
    if (c.redo()) {
      best_pos = -1;
      hit_linear_lean<COUNTED>(sv.scan_sph, sc.n_spheres, o, d, t_out, i_out, k);
      if (QUADS) hit_quads_linear(sc, o, d, t_out, i_out);
      return;
    }
    t_out = c.best_i >= 0 ? c.best_t : -1.0f;
    i_out = c.best_i;
  }
};
