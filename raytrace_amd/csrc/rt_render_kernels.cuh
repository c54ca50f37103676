// rt_render_kernels.cuh — the path-tracing megakernels: rt_render_kernel (a lane owns a pixel and sums its samples in
// registers) and rt_render_stored_kernel (every sample is stored; lanes keep a camera ray in reserve).
#pragma once

#include "rt_kernels.cuh"

struct render_args {
  rt_params p;                 // seed_tex inside is a host pointer and is not used on the device
  int vrows;                   // rows of this call's share of the tile (virtual rows, see real_row)
  uint32_t reserve_offset;     // rt_render_stored_kernel: where the lanes' camera rays in reserve lie in the dynamic LDS
  int pairs;                   // main pass of rt_render_stored_kernel: records are written in pairs (pair_slot)
  int refill_vote;             // ... and how many empty reserves make a refill worth a wave,
  int refill_min;              // ... or at least warranted when a lane has run out of rays
  scene_dev sc;
  float* accum;                // width*height*3 radiance sums
  const uint32_t* seed_tex;    // device copy of p.seed_tex, or null
  unsigned long long* counters;  // sizeof(rt_counters)/8 words, or null
  unsigned int* queue;         // head of the pixel queue, zero at launch
  int vote;                    // ready lanes needed before a wave leaves traversal for the path phase
  int leaf_vote;               // lanes with queued leaves needed before a wave takes a leaf step
  // Two-pass rendering (pixel queue ordered by measured cost):
  //   pass 0  everything in one launch, queue in tile order
  //   pass 1  probe: the first few samples of every pixel in tile order; leaves the pixel's
  //           running sum, RNG state and segment count in the carry arrays, not in accum
  //   pass 2  main: the remaining samples, taken job by job (most expensive pixels first),
  //           each pixel continuing from its carried sum and RNG state
  //   pass 3  probe in `sweeps` sweeps over the image (counter RNG only): every pixel's probe samples
  //           are cut into that many ranges taken by different lanes, so that the pixels that turn
  //           out expensive hold the end of the launch up for a fraction of their cost only; every
  //           sample's radiance goes to split_samples (row = rect pixel), the segment counts are
  //           added up in carry_cost, and rt_probe_sum_kernel forms what pass 1 would have left
  int pass;
  int sweeps;
  float* carry_colour;         // 3 floats per rect pixel
  uint32_t* carry_rng;
  uint32_t* carry_cost;        // segments the probe spent on the rect pixel (pass 1 stores, pass 3 adds)
  // Pass 2 takes jobs, not pixels: a pixel so expensive that it alone would outlast the rest of
  // the launch is cut into several sample ranges.  The lanes working on such a pixel store every
  // sample's radiance, and rt_sum_split_kernel adds them up in sample order afterwards, which
  // keeps the pixel's sum bit-identical to a sequential evaluation.
  const uint4* jobs;           // (rect pixel, first sample, end sample, split row or RT_JOB_WHOLE)
  const uint32_t* n_jobs;      // how many there are (computed on the device)
  float4* split_samples;       // split row r, sample s of the main pass: one 16-byte record (r, g, b, -) at sample_slot(s, r, rows)
  int main_samples;            // samples per pixel of the main pass
  uint32_t split_rows;         // rows of split_samples
  // Long paths of the probe.  A probe in sweeps ends with the lanes that drew a path of dozens of segments working
  // alone.  With a threshold in the upper half of `sweeps` (pass 3), a path about to begin one more segment than that
  // is given up: the sample's record says so (RT_COST_ABORTED in its cost word), rt_probe_sum_kernel turns it into
  // job k of the jobs at the head of the main pass's table (fourth word RT_JOB_REDO | k), and pass 2 renders that
  // sample again — same counter key, so the same path, to its end — into record 2 k behind the rows of its own store.
  // (Nothing else of the kernel knows: such a record is written like any other.)
};

#define RT_JOB_WHOLE 0xFFFFFFFFu
#define RT_JOB_REDO 0x80000000u      // in a job's fourth word (kernels whose every pixel has a row): probe sample to render again, see above
#define RT_COST_ABORTED 0x80000000u  // in a probe record's cost word: the path was given up after that many segments

// Where the record of sample s (counted from the first sample of the pass) of row r lies in a per-sample store of
// `rows` rows: sample-major, plane s holds that sample of every row, so that lanes holding neighbouring rows write next
// to each other and the kernels that add the samples up read whole planes.  (Measured and not kept: groups of four
// samples of a row in one 64-byte line, written by one lane over time — the L2 does not hold the line until it is
// complete, 22 GB written per C2 frame against 17 GB.)
RT_DEV size_t sample_slot(uint32_t s, size_t r, size_t rows) { return (size_t)s * rows + r; }
// Records in pairs (main pass of rt_render_stored_kernel): samples 2q and 2q + 1 of a row lie next to each other and
// one lane writes them at once; plane q holds pair q of every row.  A slot number counts records as if they were 16
// bytes each; the pair slot / 2 itself is 24 bytes, two radiances, at 24 * (slot / 2).  RT_SLOT_HOLD in a slot number: the record
// waits in the lane's registers for its partner (the store holds fewer than 2^31 records).
#define RT_SLOT_HOLD 0x80000000u
RT_DEV size_t pair_slot(uint32_t s, size_t r, size_t rows) { return ((size_t)(s >> 1) * rows + r) * 2 + (s & 1u); }

// bounce() for a lane whose search is over (fs:201-223): hit record and material scatter, or the sky.  Returns
// whether the path goes on; if so, srch.ts.o / srch.ts.d hold the next segment's ray.  alb is the path's throughput
// and, when the path ends, its radiance.
template <int SEARCH, int RNG, bool COUNTED, bool NOISE>
RT_DEV bool shade_finished(searcher<SEARCH, COUNTED>& srch, const scene_dev& sc, const scene_view& sv, uint32_t bounce_limit, f3& alb,
                           uint32_t& rng, uint32_t& count, lane_counters& k) {
  float t;
  int idx;
  srch.finish(sc, sv, t, idx, k);
  const f3 o = srch.ts.o;
  f3 d = srch.ts.d;
  bool cont;
  f3 next_o = o;
  if (t > 0.001f) {  // fs:186-196
    RT_COUNT(k, hits, 1);
    const float4 sp = sv.sph[idx];
    hit_rec h;
    h.point = o + d * t;
    h.normal = sv.inv_r[idx] * (h.point - mk3(sp.x, sp.y, sp.z));
    if (search_traits<SEARCH>::quads && sp.w == -__builtin_inff()) {  // a quad: its plane's unit normal
      const float4 qn = sc.quad_n[sc.quad_of[idx]];
      h.normal = mk3(qn.x, qn.y, qn.z);
    }
    h.interior = false;
    if (dot3(h.normal, d) >= 0.0f) {
      h.normal = -h.normal;
      h.interior = true;
    }
    h.mat = sv.sph_mat[idx];
    const float4 ma = sv.mat_a[h.mat];
    const int type = sv.mat_type[h.mat];
    next_o = h.point;  // fs:211-212
    count = count + 1u;
    cont = shade_hit<RNG, COUNTED, NOISE>(h, type, ma.w, mk3(ma.x, ma.y, ma.z), sc.noise, d, alb, rng, k);
  } else {  // fs:217-218
    alb = sky_radiance(d, alb);
    cont = false;
    RT_COUNT(k, sky, 1);
  }
  if (cont && count >= bounce_limit) {  // the next bounce() would stop here, fs:203-205
    alb = mk3(0.0f, 0.0f, 0.0f);
    cont = false;
    RT_COUNT(k, limit_hits, 1);
  }
  if (cont) {
    srch.ts.o = next_o;
    srch.ts.d = d;
  }
  return cont;
}

// The camera block of the shader's main (fs:117-131) for sample `rng` is keyed for: jitter, point on the viewport,
// origin on the defocus disk.
struct camera_consts {
  f3 delta_u, delta_v, cam, vtl, ddu, ddv;
  float defocus_angle;
};
RT_DEV camera_consts load_camera(const rt_params& P) {
  camera_consts c;
  c.delta_u = mk3(P.delta_u[0], P.delta_u[1], P.delta_u[2]);
  c.delta_v = mk3(P.delta_v[0], P.delta_v[1], P.delta_v[2]);
  c.cam = mk3(P.camera_origin[0], P.camera_origin[1], P.camera_origin[2]);
  c.vtl = mk3(P.viewport_top_left[0], P.viewport_top_left[1], P.viewport_top_left[2]);
  c.ddu = mk3(P.defocus_disk_u[0], P.defocus_disk_u[1], P.defocus_disk_u[2]);
  c.ddv = mk3(P.defocus_disk_v[0], P.defocus_disk_v[1], P.defocus_disk_v[2]);
  c.defocus_angle = P.defocus_angle;
  return c;
}
template <int RNG, bool COUNTED>
RT_DEV void camera_ray(const camera_consts& C, int x, int y, uint32_t& rng, f3& ro, f3& rd, lane_counters& k) {
  const float jx = rng_float<RNG>(rng) - 0.5f;
  const float jy = rng_float<RNG>(rng) - 0.5f;
  const f3 frag = C.vtl + ((float)x + jx) * C.delta_u + ((float)y + jy) * C.delta_v;
  ro = C.cam;
  if (!(C.defocus_angle <= 0.0f)) {
    const f3 disk = random_unit_disk<RNG, COUNTED>(rng, k);
    ro = C.cam + disk.x * C.ddu + disk.y * C.ddv;
  }
  rd = frag - ro;
}

// Persistent lanes with phase voting.
//
// The launch holds as many workgroups as the chip keeps resident.  Every lane
// pulls pixels from one global queue (the compiler folds the per-lane atomicAdd
// into one atomic per wave, ranked by ballot / mbcnt) and keeps pulling until
// the queue is dry.  Queue order: 8x8-pixel tiles, a tile's 64 pixels
// consecutive, so a wave starts on one coherent tile.  A lane owns its pixel
// for the whole sample range and forms the pixel's sum in sample order in
// registers — no atomics on the image, bit-identical to a sequential evaluation.
//
// A lane is either traversing the tree or ready for its path phase (shade the
// hit or the sky, close the sample, fetch a pixel, start a camera ray, set up
// the next segment).  Each iteration the wave takes ONE of the two actions,
// chosen by ballot: a traversal step for the traversing lanes, or the path
// phase for the ready ones.  Rays that finish their traversal early therefore
// do not hold the wave until the slowest ray of the segment is done; they wait
// only until enough lanes are ready to make the path phase worth issuing.
template <int SEARCH, int RNG, bool COUNTED, bool NOISE>
__global__ __launch_bounds__(search_traits<SEARCH>::threads, search_traits<SEARCH>::render_waves_per_simd) void rt_render_kernel(const render_args A) {
  const rt_params& P = A.p;
  const scene_dev& sc = A.sc;
  scene_view sv;
  uint32_t* stack_col;
  uint32_t stage;  // this wave's node staging area (byte offset into LDS)
  search_setup<SEARCH>(sc, sv, stack_col, stage);

  const int tw = P.tile_x1 - P.tile_x0, th = A.vrows;
  const int tiles_x = (tw + 7) >> 3, tiles_y = (th + 7) >> 3;
  const uint32_t tile_items = (uint32_t)tiles_x * (uint32_t)tiles_y * 64u;
  const uint32_t total_items = A.pass == 2 ? *A.n_jobs : (A.pass == 3 ? tile_items * (uint32_t)A.sweeps : tile_items);

  lane_counters k = {};
  const camera_consts C = load_camera(P);

  searcher<SEARCH, COUNTED> srch;
  srch.ref_mask = sc.ref_mask;
  // The lane's state, as bits of one vector register (see rt_render_stored_kernel): HAVE this lane owns a pixel, ALIVE
  // a path of that pixel is in flight, PENDING a finished search waits to be shaded, EXHAUSTED the queue is dry.
  enum : uint32_t { HAVE = 1u, ALIVE = 2u, PENDING = 4u, EXHAUSTED = 8u };
  uint32_t st = P.sample_end <= P.sample_begin ? EXHAUSTED : 0u;
  auto has = [&](uint32_t bits) { return (st & bits) != 0u; };
  auto none = [&](uint32_t bits) { return (st & bits) == 0u; };
  srch.reset();             // (a lane is traversing while its search is active: no flag of its own)
  int x = 0, y = 0, s = 0, s_end = P.sample_end;
  uint32_t pixel = 0, rect_pixel = 0, rng = 0, count = 0, pixel_segments = 0, split_row = RT_JOB_WHOLE;
  f3 colour = mk3(0.0f, 0.0f, 0.0f);
  f3 alb = colour;
  uint32_t pool_next = 0, pool_end = 0;  // the wave's share of the pixel queue

  // The pixel's sample range is over: pass 0 / 2 add its sum to the image, the
  // probe pass parks it (with the RNG state and a cost key) for pass 2.
  auto close_pixel = [&]() {
    if (A.pass == 3) {
      if (A.carry_cost) atomicAdd(A.carry_cost + rect_pixel, pixel_segments);
    } else if (A.pass == 1) {
      float* dst = A.carry_colour + 3 * (size_t)rect_pixel;
      dst[0] = colour.x;
      dst[1] = colour.y;
      dst[2] = colour.z;
      A.carry_rng[rect_pixel] = rng;
      A.carry_cost[rect_pixel] = pixel_segments;
    } else if (split_row == RT_JOB_WHOLE) {
      float* dst = A.accum + 3 * (size_t)pixel;
      dst[0] = colour.x;
      dst[1] = colour.y;
      dst[2] = colour.z;
    }
    st &= ~HAVE;
  };
  // The finished path's radiance joins the pixel's running sum, or, for a pixel cut into
  // several jobs, goes to its own slot to be summed in order later.
  auto add_sample = [&]() {
    if (split_row == RT_JOB_WHOLE) {
      colour = colour + alb;
    } else {
      // sample-major: plane s holds that sample of every row (split pixels of the main pass; every pixel of a pass
      // in sweeps), so that lanes holding neighbouring rows write next to each other
      const size_t rows = A.pass == 3 ? (size_t)tw * (size_t)th : (size_t)A.split_rows;
      A.split_samples[sample_slot((uint32_t)(s - P.sample_begin), (size_t)split_row, rows)] = make_float4(alb.x, alb.y, alb.z, 0.0f);
    }
  };

  RT_CLK_DECL
  while (true) {
    // Traversal steps for the lanes that are traversing, until enough lanes are ready for the path phase to be worth
    // a full wave (or no lane is traversing).  Which lanes have run out of work changes in the path phase only.
    const unsigned long long m_live = __ballot(none(EXHAUSTED));
    if (m_live == 0ull) break;
    while (search_traits<SEARCH>::tree) {  // `pending` (set at begin) marks the result as unread
      const unsigned long long m_trav = __ballot(srch.active());
      if (m_trav == 0ull || __popcll(m_live & ~m_trav) >= A.vote) break;
      RT_CLK_START();
      traversal_action(srch, sc, sv, stack_col, stage, A.leaf_vote, k);
      RT_CLK_STOP(0);
    }
    RT_CLK_START();
    RT_COUNT(k, path_wave_phases, (threadIdx.x & 63) == 0 ? 1u : 0u);
    const bool ready = !srch.active() && none(EXHAUSTED);  // only ready lanes take the path phase
    RT_COUNT(k, path_lane_phases, ready ? 1u : 0u);

    bool start_segment = false;
    if (ready && has(PENDING)) {  // the search for the current segment is over: bounce(), fs:207-219
      st &= ~PENDING;
      const bool cont = shade_finished<SEARCH, RNG, COUNTED, NOISE>(srch, sc, sv, P.bounce_limit, alb, rng, count, k);
      RT_CLK_STOP(1);
      if (cont) {
        start_segment = true;
      } else {  // fs:132: colour += raycast(...)
        add_sample();
        st &= ~ALIVE;
        s++;
        if (s >= s_end) close_pixel();
      }
    }
    RT_CLK_STOP(3);
    const bool need_pixel = ready && !start_segment && none(HAVE);
    const uint32_t item = take_items<unsigned int, RT_JOB_CHUNK>(need_pixel, A.queue, pool_next, pool_end);
    if (ready && !start_segment) {
      if (none(HAVE)) {  // the next pixel
        if (item >= total_items) {
          st |= EXHAUSTED;
        } else {
          uint4 job = make_uint4(0u, (uint32_t)P.sample_begin, (uint32_t)P.sample_end, RT_JOB_WHOLE);
          int vy;
          if (A.pass == 2) {
            job = A.jobs[item];
            rect_pixel = job.x;
            x = P.tile_x0 + (int)(rect_pixel % (uint32_t)tw);
            vy = (int)(rect_pixel / (uint32_t)tw);
          } else {
            uint32_t it = item;
            if (A.pass == 3) {  // sweep q takes the q-th part of every pixel's probe samples
              const uint32_t sweep = item / tile_items;
              it = item - sweep * tile_items;
              // (32-bit arithmetic: the host runs a pass in sweeps only over fewer than 2^24 samples, in at most 64 sweeps)
              const uint32_t n_probe = (uint32_t)(P.sample_end - P.sample_begin), first = n_probe * sweep / (uint32_t)A.sweeps;
              job.y = (uint32_t)P.sample_begin + first;
              job.z = (uint32_t)P.sample_begin + (n_probe * sweep + n_probe) / (uint32_t)A.sweeps;
            }
            const uint32_t tile = it >> 6, within = it & 63u;
            x = P.tile_x0 + (int)(tile % (uint32_t)tiles_x) * 8 + (int)(within & 7u);
            vy = (int)(tile / (uint32_t)tiles_x) * 8 + (int)(within >> 3);
            rect_pixel = (uint32_t)vy * (uint32_t)tw + (uint32_t)(x - P.tile_x0);
          }
          y = real_row(vy, P.tile_y0, P.strip_count, P.strip_index);
          if (x < P.tile_x1 && vy < th) {
            st = (st | HAVE) & ~ALIVE;
            pixel = (uint32_t)y * (uint32_t)P.width + (uint32_t)x;
            s = (int)job.y;
            s_end = (int)job.z;
            split_row = A.pass == 3 ? rect_pixel : job.w;
            pixel_segments = 0;
            if (A.pass == 2) {  // continue the sum and the RNG stream the probe pass left
              const float* src = A.carry_colour + 3 * (size_t)rect_pixel;
              colour = mk3(src[0], src[1], src[2]);
              rng = A.carry_rng[rect_pixel];
            } else {
              // the sum goes on from what earlier sample ranges left in the image (zero on a
              // fresh frame, fs:115): ranges rendered in order add up exactly as one render
              if (A.pass != 3) {  // (a probe in sweeps stores samples; rt_probe_sum_kernel adds them to the image's sum)
                const float* src = A.accum + 3 * (size_t)pixel;
                colour = mk3(src[0], src[1], src[2]);
              }
              if (RNG == RT_RNG_XORSHIFT_REF) {
                const uint32_t seed = A.seed_tex ? A.seed_tex[pixel] : 0u;  // fs:105-106
                rng = seed ^ cantor_pair((uint32_t)x, (uint32_t)y);         // fs:107
              }
            }
          }
        }
      }
      RT_CLK_STOP(3);
      if ((st & (HAVE | ALIVE)) == HAVE) {  // fs:117-131: next camera sample of this pixel
        if (RNG == RT_RNG_COUNTER) rng = counter_key(P.seed, pixel, (uint32_t)s);
        f3 ro, rdir;
        camera_ray<RNG, COUNTED>(C, x, y, rng, ro, rdir, k);
        alb = mk3(1.0f, 1.0f, 1.0f);
        count = 0;
        st |= ALIVE;
        RT_COUNT(k, paths, 1);
        if (P.bounce_limit == 0u) {  // fs:203-205 on the first bounce()
          alb = mk3(0.0f, 0.0f, 0.0f);
          RT_COUNT(k, limit_hits, 1);
          add_sample();
          st &= ~ALIVE;
          s++;
          if (s >= s_end) close_pixel();
        } else {
          srch.ts.o = ro;
          srch.ts.d = rdir;
          start_segment = true;
        }
      }
    }
    RT_CLK_STOP(4);
    if (start_segment) {
      RT_COUNT(k, segments, 1);
      pixel_segments++;
      srch.begin(sc, sv, srch.ts.o, srch.ts.d, k);
      st |= PENDING;
    }
    RT_CLK_STOP(5);
  }
#ifdef RT_SECTION_CLOCKS
  if (COUNTED && A.counters != nullptr && (threadIdx.x & 63) == 0) {
    const unsigned long long end_ = __builtin_amdgcn_s_memtime();
    clk_[6] = end_ - clk_0_;
    for (int i = 0; i < 7; i++) atomicAdd(A.counters + 5 + i, clk_[i]);
    atomicMax(A.counters + 12, clk_[6]);  // disk_tries: the longest-running wave
    atomicAdd(A.counters + 13, 1ull);     // limit_hits: waves
  }
  k.hits = k.sky = k.lambertian = k.metallic = k.dialectric = k.perlin = k.unit_tries = k.disk_tries = k.limit_hits = 0;
#endif
  flush_counters<COUNTED>(A.counters, k);
}

// ---- the render kernel for passes whose every sample is stored -----------------
// When every sample's radiance goes to the per-sample store (a pass in sweeps; a main pass whose every pixel has a
// row), a path owes nothing to the lane that runs it: no pixel sum lives in registers.  That is used to take camera
// rays out of the path phase.  In the kernel above a path that ends is followed, in the same phase, by the next
// sample's camera ray — counter key, jitter, the defocus disk's rejection loop, some 200 instructions — issued for
// the two lanes in five whose path has just ended.  Here every lane keeps ONE camera ray in reserve in LDS (32 bytes:
// origin, direction, generator state, the slot its sample goes to).  A lane whose path ends takes its reserve with
// two LDS reads and goes on; the reserves are refilled, for all lanes that have used theirs, when `refill_vote` of
// them are empty (or a lane needs one and has none), i.e. by a nearly full wave.  A lane walks its job's samples in
// order either way, and a sample's value does not depend on when its ray was made (counter generator), so the
// image's bits are those of the kernel above.  The probe's cost measure travels in the record's fourth word.
#define RT_REFILL_VOTE 40
#define RT_REFILL_MIN 16
#define RT_RESERVE_BYTES 2048  // per wave: two rows of 64 float4
// OWNED: the main pass of a frame in which only some pixels (or none) are cut into jobs with stored samples.  A job
// that is a whole pixel's sample range (RT_JOB_WHOLE) has no row in the store: the lane that takes it forms the
// pixel's sum in registers, in sample order, as rt_render_kernel does — from the sum the probe carried, written to
// the image after the job's last sample.  A lane walks a job's samples in order and one path at a time, so one
// running sum per lane is enough; what travels with a ray in the reserve's slot word is, for such a job, the pixel
// and whether the sample is the job's first or last (the job variables may be a job ahead of the path in flight).
#define RT_TAG_WHOLE 0x80000000u
#define RT_TAG_FIRST 0x40000000u
#define RT_TAG_LAST 0x20000000u
#define RT_TAG_PIXEL 0x1FFFFFFFu  // the tile's pixel number: the host keeps such passes to tiles below 2^29 pixels

template <int SEARCH, bool COUNTED, bool NOISE, bool OWNED>
__global__ __launch_bounds__(search_traits<SEARCH>::threads, search_traits<SEARCH>::render_waves_per_simd) void rt_render_stored_kernel(const render_args A) {
  constexpr int RNG = RT_RNG_COUNTER;
  const rt_params& P = A.p;
  const scene_dev& sc = A.sc;
  scene_view sv;
  uint32_t* stack_col;
  uint32_t stage;
  search_setup<SEARCH>(sc, sv, stack_col, stage);
  float4* reserve = reinterpret_cast<float4*>(rt_lds + A.reserve_offset) + (threadIdx.x >> 6) * (RT_RESERVE_BYTES / 16) + (threadIdx.x & 63);

  const int tw = P.tile_x1 - P.tile_x0, th = A.vrows;
  const int tiles_x = (tw + 7) >> 3, tiles_y = (th + 7) >> 3;
  const uint32_t tile_items = (uint32_t)tiles_x * (uint32_t)tiles_y * 64u;
  const uint32_t n_sweeps = (uint32_t)A.sweeps & 0xFFFFu, abort_after = A.pass == 3 ? (uint32_t)A.sweeps >> 16 : 0u;
  const uint32_t total_items = A.pass == 2 ? *A.n_jobs : tile_items * n_sweeps;
  const size_t rows = A.pass == 3 ? (size_t)tw * (size_t)th : (size_t)A.split_rows;

  lane_counters k = {};
  const camera_consts C = load_camera(P);
  searcher<SEARCH, COUNTED> srch;
  srch.ref_mask = sc.ref_mask;
  srch.reset();
  // The lane's state, as bits of one vector register (flags of their own would be lane masks in scalar registers, of
  // which this kernel has none to spare, merged at every join of the wave's loop):
  //   ALIVE a path is in flight, PENDING its finished search waits to be shaded, RESERVE the lane's reserve holds a
  //   camera ray, JOB_LEFT the lane's job has samples left, JOBS_DONE the queue is dry, DONE the lane is through
  //   JOB_FIRST (OWNED) no ray of the lane's job has been made yet
  enum : uint32_t { ALIVE = 1u, PENDING = 2u, RESERVE = 4u, JOB_LEFT = 8u, JOBS_DONE = 16u, DONE = 32u, JOB_FIRST = 64u };
  uint32_t st = P.sample_end <= P.sample_begin ? DONE : 0u;
  // Records in pairs: a lane that finishes the first sample of a pair keeps its radiance in registers until the second
  // is done, then writes both records, one aligned 32-byte sector, in two stores back to back.  (A record written
  // alone leaves L2 as a sector of its own before its neighbour arrives: 32 bytes to HBM per 16-byte record.  Four
  // records at a time — a whole 64-byte half line per lane — cut the writes by another 8 % and the frame rate by
  // 0.8 %.)  Jobs begin at even samples, so a pair is one lane's.
  const bool pairs = !OWNED && A.pairs != 0 && A.pass == 2;
  f3 held = mk3(0.0f, 0.0f, 0.0f);
  f3 colour = mk3(0.0f, 0.0f, 0.0f);  // OWNED: the running sum of the whole pixel whose path is in flight
  auto has = [&](uint32_t bits) { return (st & bits) != 0u; };
  auto none = [&](uint32_t bits) { return (st & bits) == 0u; };
  int jx = 0, jy = 0, js = 0, js_end = 0;    // the job: pixel, next sample to make a ray for, end
  uint32_t jrow = 0, jpixel = 0, jrect = 0;  // its row in the store; its pixel index in the image and in the tile
  uint32_t rng = 0, count = 0, cur_slot = 0, path_segments = 0;
  f3 alb = mk3(0.0f, 0.0f, 0.0f);
  uint32_t pool_next = 0, pool_end = 0;

  while (true) {
    const unsigned long long m_live = __ballot(none(DONE));
    if (m_live == 0ull) break;
    while (search_traits<SEARCH>::tree) {
      const unsigned long long m_trav = __ballot(srch.active());
      if (m_trav == 0ull || __popcll(m_live & ~m_trav) >= A.vote) break;
      traversal_action(srch, sc, sv, stack_col, stage, A.leaf_vote, k);
    }
    RT_COUNT(k, path_wave_phases, (threadIdx.x & 63) == 0 ? 1u : 0u);
    const bool ready = !srch.active() && none(DONE);
    RT_COUNT(k, path_lane_phases, ready ? 1u : 0u);

    bool start_segment = false;
    if (ready && has(PENDING)) {
      st &= ~PENDING;
      if (shade_finished<SEARCH, RNG, COUNTED, NOISE>(srch, sc, sv, P.bounce_limit, alb, rng, count, k)) {
        start_segment = true;
      } else {  // fs:132: the sample's radiance, and what the path cost
        if (OWNED && (cur_slot & RT_TAG_WHOLE) != 0u) {
          colour = colour + alb;
          if ((cur_slot & RT_TAG_LAST) != 0u) {  // the pixel's sum is complete
            const uint32_t rp = cur_slot & RT_TAG_PIXEL;
            const int px = P.tile_x0 + (int)(rp % (uint32_t)tw);
            const int py = real_row((int)(rp / (uint32_t)tw), P.tile_y0, P.strip_count, P.strip_index);
            float* dst = A.accum + 3 * ((size_t)py * (size_t)P.width + (size_t)px);
            dst[0] = colour.x;
            dst[1] = colour.y;
            dst[2] = colour.z;
          }
        } else if (pairs && (cur_slot & RT_SLOT_HOLD) != 0u) {
          held = alb;
        } else if (pairs) {
          // Pair q of the store is 24 bytes — the radiance of its two samples, nothing else — at q * 24.  An odd
          // record always follows its partner in this lane (jobs begin at even samples): both go out together; an
          // even record that comes here is the only one of its pair (a job's last sample, a probe sample rendered again).
          float2* q = reinterpret_cast<float2*>(A.split_samples) + 3 * (size_t)(cur_slot >> 1);
          if ((cur_slot & 1u) != 0u) {
            q[0] = make_float2(held.x, held.y);
            q[1] = make_float2(held.z, alb.x);
            q[2] = make_float2(alb.y, alb.z);
          } else {
            q[0] = make_float2(alb.x, alb.y);
            reinterpret_cast<float*>(q)[2] = alb.z;
          }
        } else {
          A.split_samples[cur_slot] = make_float4(alb.x, alb.y, alb.z, __uint_as_float(path_segments));
        }
        st &= ~ALIVE;
      }
      if (!OWNED && start_segment && abort_after != 0u && path_segments >= abort_after) {
        // pass 3 only: the path is left to a job of the main pass (render_args, "long paths of the probe")
        A.split_samples[cur_slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(RT_COST_ABORTED | path_segments));
        start_segment = false;
        st &= ~ALIVE;
      }
    }
    // a lane without a path takes the ray it holds in reserve
    auto take_reserve = [&]() {
      const float4 r0 = reserve[0], r1 = reserve[64];
      srch.ts.o = mk3(r0.x, r0.y, r0.z);
      srch.ts.d = mk3(r1.x, r1.y, r1.z);
      rng = __float_as_uint(r0.w);
      cur_slot = __float_as_uint(r1.w);
      if (OWNED && (cur_slot & (RT_TAG_WHOLE | RT_TAG_FIRST)) == (RT_TAG_WHOLE | RT_TAG_FIRST)) {
        const float* src = A.carry_colour + 3 * (size_t)(cur_slot & RT_TAG_PIXEL);  // the sum the probe pass left
        colour = mk3(src[0], src[1], src[2]);
      }
      alb = mk3(1.0f, 1.0f, 1.0f);
      count = 0;
      path_segments = 0;
      st = (st | ALIVE) & ~RESERVE;
      start_segment = true;
      RT_COUNT(k, paths, 1);
    };
    if (ready && (st & (ALIVE | RESERVE)) == RESERVE) take_reserve();
    // Refill the reserves: worth a wave when `refill_vote` are empty; when a lane is out of rays, already at
    // `refill_min` empty ones (below that the lane waits for company) — or at once if no lane of the wave would
    // be traversing meanwhile.
    const bool fill = none(RESERVE | JOBS_DONE | DONE);
    const unsigned long long m_fill = __ballot(fill);
    const int n_fill = __popcll(m_fill);
    const bool starving = __ballot(fill && ready && none(ALIVE)) != 0ull;
    if (m_fill != 0ull && (n_fill >= A.refill_vote || (starving && (n_fill >= A.refill_min || __ballot(srch.active() || start_segment) == 0ull)))) {
      const uint32_t item = take_items<unsigned int, RT_JOB_CHUNK>(fill && none(JOB_LEFT), A.queue, pool_next, pool_end);
      if (fill) {
        if (none(JOB_LEFT)) {
          if (item >= total_items) {
            st |= JOBS_DONE;
          } else {
            int vy;
            uint32_t rect_pixel;
            if (A.pass == 2) {
              const uint4 job = A.jobs[item];
              rect_pixel = job.x;
              jx = P.tile_x0 + (int)(rect_pixel % (uint32_t)tw);
              vy = (int)(rect_pixel / (uint32_t)tw);
              js = (int)job.y;
              js_end = (int)job.z;
              jrow = job.w;
            } else {  // sweep q takes the q-th part of every pixel's samples
              const uint32_t sweep = item / tile_items, it = item - sweep * tile_items;
              // (32-bit arithmetic: the host runs a pass in sweeps only over fewer than 2^24 samples, in at most 64 sweeps)
              const uint32_t n_all = (uint32_t)(P.sample_end - P.sample_begin);
              js = P.sample_begin + (int)(n_all * sweep / n_sweeps);
              js_end = P.sample_begin + (int)((n_all * sweep + n_all) / n_sweeps);
              const uint32_t tile = it >> 6, within = it & 63u;
              jx = P.tile_x0 + (int)(tile % (uint32_t)tiles_x) * 8 + (int)(within & 7u);
              vy = (int)(tile / (uint32_t)tiles_x) * 8 + (int)(within >> 3);
              rect_pixel = (uint32_t)vy * (uint32_t)tw + (uint32_t)(jx - P.tile_x0);
              jrow = rect_pixel;
            }
            jy = real_row(vy, P.tile_y0, P.strip_count, P.strip_index);
            jpixel = (uint32_t)jy * (uint32_t)P.width + (uint32_t)jx;
            jrect = rect_pixel;
            if (jx < P.tile_x1 && vy < th && js < js_end) st |= JOB_LEFT | JOB_FIRST;
          }
        }
        if (has(JOB_LEFT)) {  // fs:117-131 for the job's next sample, into the reserve
          uint32_t r = counter_key(P.seed, jpixel, (uint32_t)js);
          f3 ro, rd;
          camera_ray<RNG, COUNTED>(C, jx, jy, r, ro, rd, k);
          const uint32_t s_rel = (uint32_t)(js - P.sample_begin);
          uint32_t slot = (uint32_t)sample_slot(s_rel, (size_t)jrow, rows);
          if (pairs) slot = (uint32_t)pair_slot(s_rel, (size_t)jrow, rows) | ((s_rel & 1u) == 0u && js + 1 < js_end ? RT_SLOT_HOLD : 0u);
          if (!OWNED && (jrow & RT_JOB_REDO) != 0u) slot = (uint32_t)rows * (((uint32_t)A.main_samples + 1u) & ~1u) + 2u * (jrow & ~RT_JOB_REDO);  // (an even slot: written alone)
          if (OWNED && jrow == RT_JOB_WHOLE)
            slot = RT_TAG_WHOLE | (has(JOB_FIRST) ? RT_TAG_FIRST : 0u) | (js + 1 < js_end ? 0u : RT_TAG_LAST) | jrect;
          st &= ~JOB_FIRST;
          reserve[0] = make_float4(ro.x, ro.y, ro.z, __uint_as_float(r));
          reserve[64] = make_float4(rd.x, rd.y, rd.z, __uint_as_float(slot));
          st |= RESERVE;
          js++;
          if (!(js < js_end)) st &= ~JOB_LEFT;
        }
      }
      if (ready && (st & (ALIVE | RESERVE)) == RESERVE) take_reserve();
    }
    if (ready && (st & (ALIVE | RESERVE | JOBS_DONE)) == JOBS_DONE) st |= DONE;  // no path, no reserve, no job: the lane is through
    if (start_segment) {
      RT_COUNT(k, segments, 1);
      path_segments++;
      srch.begin(sc, sv, srch.ts.o, srch.ts.d, k);
      st |= PENDING;
    }
  }
  flush_counters<COUNTED>(A.counters, k);
}

