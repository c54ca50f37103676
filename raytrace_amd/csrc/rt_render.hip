// rt_render.hip — rt_render: the frame scheduler (one pass, sweeps, probe + cost-sorted main pass).  The launches of the
// two render kernels are in rt_render_pixels.hip and rt_render_stored.hip.

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "rt_render_launch.hpp"
#include "rt_plan_kernels.cuh"

namespace {

// Cost levels per power of two in the main pass's sort key, as a power of two (RT_COST_STEPS_LOG2 overrides, 0..6).
int cost_steps_log2() {
  static int v = [] {
    const char* e = tuning_env("RT_COST_STEPS_LOG2");
    int x = e ? atoi(e) : 3;
    return x < 0 ? 0 : (x > 6 ? 6 : x);
  }();
  return v;
}

// How many of the most expensive pixels the main pass may cut into several jobs (RT_SPLIT_ROWS overrides; the
// memory budget for their per-sample radiance limits it further).
int split_rows() {
  static int v = [] {
    const char* e = tuning_env("RT_SPLIT_ROWS");
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
    const char* e = tuning_env("RT_MIN_PARTS");
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
    const char* e = tuning_env("RT_SPLIT_BUDGET_MB");
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
    const char* e = tuning_env("RT_PROBE_SWEEPS");
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
    const char* e = tuning_env("RT_SPLIT_FINENESS");
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
    const char* e = tuning_env("RT_PROBE_SAMPLES");
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
// Where a sample is cheap — the scan over a few spheres, a tree of a few nodes — jobs are eight samples, and a probe's
// half the probe: the two-sphere noise scene at 1920x1080x1024 takes 60.2 / 59.2 / 58.6 / 57.9 / 58.7 ms with 6 / 4 / 3 /
// 2 / 1 jobs per pixel of its twelve-sample probe, and at 32 samples, rendered in sweeps as a whole, 6.64 / 3.90 / 3.63 /
// 4.34 ms with 16 / 8 / 4 / 2 jobs per pixel.
int sweeps_for(int n, bool cheap_samples = false) {
  const bool fixed = tuning_env("RT_PROBE_SWEEPS") != nullptr;
  int s = fixed ? probe_sweeps() : (cheap_samples ? std::max(2, n / 8) : (n + 1) / 2);
  s = s > probe_sweeps() ? probe_sweeps() : s;
  s = s > n ? n : s;
  return s < 1 ? 1 : s;
}
// Segments after which the probe pass gives a path up and leaves its sample to the main pass (render_args, "long paths of the probe";
// RT_PROBE_ABORT overrides, 0: never).  One rank's eighth of a 1920x1080 frame: probe pass 1.03 ms with paths run to
// their end, 0.70 ms cut at 16 segments (3 % of the work lies beyond); the eighth as a whole 7.49 ms without,
// 7.16 / 7.18 / 7.22 / 7.25 / 7.38 ms giving up after 8 / 12 / 16 / 24 / 32 segments, the whole frame 0.3 ms less too.
int probe_abort_segments() {  // (read at every render, so that one process can compare)
  const char* e = tuning_env("RT_PROBE_ABORT");
  const int x = e ? atoi(e) : 12;
  return x < 0 ? 0 : (x > 1000 ? 1000 : x);
}
bool stored_kernel_enabled() {  // RT_STORED_KERNEL=0: every pass in the kernel whose lanes own their pixels (comparison)
  static bool v = [] {
    const char* e = tuning_env("RT_STORED_KERNEL");
    return !(e && e[0] == '0');
  }();
  return v;
}
bool two_pass_enabled() {
  static bool v = [] {
    const char* e = tuning_env("RT_TWO_PASS");
    return !(e && e[0] == '0');
  }();
  return v;
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

// Empty reserves that make a refill worth a wave (RT_REFILL_VOTE overrides, for tuning).
int refill_vote() {
  static int v = vote_from_env("RT_REFILL_VOTE", RT_REFILL_VOTE);
  return v;
}

int refill_min() {
  static int v = vote_from_env("RT_REFILL_MIN", RT_REFILL_MIN);
  return v;
}

int render_impl(rt_ctx* c, const rt_params* p, bool counted, bool nested = false) {
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
    if (!nested) HIP_TRY(hipMemsetAsync(c->counters.p, 0, sizeof(rt_counters), c->stream));
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
    return launch_pixel_pass(c, rp, counted, ra);
  };
  auto launch_stored_pass = [&](const render_args& ra, bool owned = false) -> hipError_t {
    used_stored = true;
    hipError_t qe = hipMemsetAsync(c->queue.p, 0, sizeof(unsigned long long), c->stream);
    if (qe != hipSuccess) return qe;
    return rt_impl::launch_stored_pass(c, counted, owned, ra);
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
  if (!nested) HIP_TRY(hipEventRecord(c->ev_k0, c->stream));
  hipError_t e;
  // Few samples, counter RNG: the frame as a whole is rendered in sweeps (short jobs in tile order, per-sample
  // radiance summed in order afterwards) — a pixel per job would end with the expensive pixels' lanes working alone.
  size_t split_budget, probe_budget;
  scratch_budgets(c, split_budget, probe_budget);
  // A long sample range on a middling number of pixels (six to twenty-four per lane), where the store cannot give
  // every pixel a row: whole pixels as jobs balance badly there (a quarter of a 3840x2160x4096 frame, eight pixels
  // per lane: 867 ms against 780 for a quarter of the whole frame's time; half of it, sixteen per lane: 1 600 to
  // 1 680 ms against 1 565), and chunks of samples (below) want fewer pixels.  Such a share is rendered as smaller ones
  // one after the other — interleaved strip sets of its own strips, four pixels per lane each — which the chunks
  // then take: 799 to 804 ms, and 1 592 to 1 598.  The pixels of the parts are disjoint, so the image is the same.
  // With still more pixels per lane whole-pixel jobs do well (the whole frame, 32 per lane: 3.13 s, in eight parts
  // 3.19 s: every part has a probe pass and tails of its own).  (RT_SAMPLE_CHUNKS=0: never.)
  if (!nested && two_pass && stored_ok && p->rng_mode == RT_RNG_COUNTER && split_rows() < 0 && c->n_nodes >= 32 &&
      !(c->search == SEARCH_LINEAR || c->search == SEARCH_LINEAR_LDS) && !(tuning_env("RT_SAMPLE_CHUNKS") && tuning_env("RT_SAMPLE_CHUNKS")[0] == '0')) {
    const size_t lanes = (size_t)c->n_cu * 1024, most = 4 * lanes;
    const size_t fit_samples = split_budget / (n_rect * kSampleFloats * sizeof(float));
    if (n_rect > 6 * lanes && n_rect < 24 * lanes && fit_samples < stored_samples(n_samples - probe_samples(n_samples, true))) {
      const int parts = (int)((n_rect + most - 1) / most), mine = p->strip_count ? p->strip_count : 1;
      if ((long long)mine * parts <= 65535) {
        for (int j = 0; j < parts; j++) {
          rt_params q = *p;
          q.strip_count = (uint16_t)(mine * parts);
          q.strip_index = (uint16_t)(p->strip_index + mine * j);
          const int rc = render_impl(c, &q, counted, true);
          if (rc) return rc;
        }
        HIP_TRY(hipEventRecord(c->ev_k1, c->stream));
        c->have_kernel_time = true;
        return RT_OK;
      }
    }
  }
  bool whole_in_sweeps = !two_pass && two_pass_enabled() && stored_ok && probe_sweeps() > 1 &&
                         n_samples >= 2 && n_rect >= 16384 &&
                         n_rect * stored_samples(n_samples) * kSampleFloats * sizeof(float) <= probe_budget;
  // (a store that cannot be had costs speed, never the frame: the plain schedules below need none)
  if (whole_in_sweeps && c->probe_samples.fit(n_rect * stored_samples(n_samples) * kSampleFloats) != hipSuccess) whole_in_sweeps = false;
  if (whole_in_sweeps) {
    render_args sw = a;
    sw.pass = 3;
    sw.sweeps = sweeps_for(n_samples, (c->search == SEARCH_LINEAR || c->search == SEARCH_LINEAR_LDS || c->n_nodes < 32) && c->n_spheres < 64);
    sw.split_samples = reinterpret_cast<float4*>(c->probe_samples.p);
    sw.main_samples = n_samples;
    e = launch_stored_pass(sw);
    if (e == hipSuccess) {
      rt_probe_sum_kernel<<<(unsigned)((n_rect + 255) / 256), 256, 0, c->stream>>>(
          (uint32_t)n_rect, reinterpret_cast<float4*>(c->probe_samples.p), n_samples, p->tile_x0, p->tile_y0, tw, p->width,
          p->strip_count, p->strip_index, c->accum(), nullptr, nullptr, nullptr, nullptr, nullptr, 0);
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
    const bool cheap_samples = c->search == SEARCH_LINEAR || c->search == SEARCH_LINEAR_LDS || c->n_nodes < 32;
    const int sweeps = sweeps_for(n_probe, cheap_samples && c->n_spheres < 64);
    probe.p.sample_end = p->sample_begin + n_probe;
    // Long probe paths are given up and rendered again by the main pass when that pass stores every sample of every
    // pixel (decided here from the budget, before the probe runs; not in counted runs, whose counters are the frame's
    // work and nothing twice).
    HIP_TRY(c->plan_counts.resize(2));  // [0] jobs of the main pass, [1] of them probe samples to render again
    HIP_TRY(hipMemsetAsync(c->plan_counts.p, 0, 2 * sizeof(uint32_t), c->stream));
    bool redo = false;
    {
      const int ms = n_samples - n_probe;
      const bool scans_ = c->search == SEARCH_LINEAR || c->search == SEARCH_LINEAR_LDS;
      const size_t rows_fit = split_budget / (stored_samples(ms) * kSampleFloats * sizeof(float));
      // (the samples rendered again lie behind the rows of the main pass's store, two records apart: slot numbers stay
      // below 2^31, whose bit marks a record that waits for its partner)
      const unsigned long long records = (unsigned long long)n_rect * stored_samples(ms) + 2ull * n_rect * (unsigned long long)n_probe;
      // (where the main pass will give every pixel a row: a real tree, or a scan over many spheres — `heavy` below)
      const bool heavy_ = (!scans_ && c->n_nodes >= 32) || (scans_ && c->n_spheres >= 64);
      redo = swept && !counted && probe_abort_segments() > 0 && n_probe <= 32 && heavy_ && split_rows() < 0 &&
             rows_fit >= n_rect && records < (1ull << 31);
      // (and only if the store is really to be had: without it the main pass could not take those samples)
      if (redo && c->split_samples.fit((size_t)records * kSampleFloats) != hipSuccess) redo = false;
    }
    if (swept) {
      probe.pass = 3;
      probe.sweeps = sweeps;
      probe.split_samples = reinterpret_cast<float4*>(c->probe_samples.p);
      probe.main_samples = n_probe;
      if (redo) probe.sweeps |= probe_abort_segments() << 16;  // (sweeps <= 64; see render_args)
    }
    if (redo) HIP_TRY(c->plan_jobs.resize((size_t)n_rect * 8 + (size_t)c->n_cu * 1024 * (size_t)split_fineness() + 64 + n_rect * (size_t)n_probe));
    e = swept ? launch_stored_pass(probe) : launch(probe, p);
    if (e == hipSuccess && swept) {
      rt_probe_sum_kernel<<<(unsigned)((n_rect + 255) / 256), 256, 0, c->stream>>>(
          (uint32_t)n_rect, reinterpret_cast<float4*>(c->probe_samples.p), n_probe, p->tile_x0, p->tile_y0, tw, p->width,
          p->strip_count, p->strip_index, c->accum(), c->carry_colour.p, c->carry_rng.p, c->carry_cost.p,
          redo ? c->plan_jobs.p : nullptr, c->plan_counts.p + 1, p->sample_begin);
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
      const uint32_t n = (uint32_t)n_rect;
      // a job should be a small fraction (1 / split_fineness()) of a lane's fair share of the frame
      const uint32_t lanes = (uint32_t)c->n_cu * 1024u * (uint32_t)split_fineness();
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
      // So many samples that the budget cannot give every pixel a row for all of them (3840x2160x4096: 540 GB): the
      // range is rendered in chunks of samples, each with a row for every pixel, all in the order and from the costs the
      // one probe found.  The sums continue from chunk to chunk in sample order, so the image is that of one pass.
      // Only where a lane has few pixels: whole pixels as jobs balance badly there (an eighth of that frame, 4 pixels
      // per lane: 386 ms in two chunks against 456 to 489 ms with rows for the 525 000 most expensive pixels only),
      // while a lane with many pixels is better off summing them in registers than storing every sample (the whole
      // frame, 32 per lane: 3.16 s against 3.50 s in sixteen chunks; at 8 per lane the two tie).
      // RT_SAMPLE_CHUNKS=0: never (comparison).
      static const bool chunks_enabled = [] {
        const char* e = tuning_env("RT_SAMPLE_CHUNKS");
        return !(e && e[0] == '0');
      }();
      constexpr size_t kRecordBytes = kSampleFloats * sizeof(float), kFewestChunkSamples = 64, kChunkPixelsPerLane = 6;
      int chunk_len = n_samples;
      if (walks_tree && stored_ok && split_rows() < 0 && chunks_enabled && n_rect <= kChunkPixelsPerLane * (size_t)c->n_cu * 1024) {
        const size_t fit_samples = split_budget / (n_rect * kRecordBytes);
        if (fit_samples < stored_samples(n_samples - n_probe)) {
          // (whole pairs of records, and fewer than 2^31 records in the store: see `pairs` below)
          const size_t cap = std::min(fit_samples, (size_t)(((1ull << 31) - 1ull) / n_rect)) & ~(size_t)1;
          if (cap >= kFewestChunkSamples) {
            const size_t chunks = ((size_t)n_samples + cap - 1) / cap;
            chunk_len = (int)((((size_t)n_samples + chunks - 1) / chunks + 1) & ~(size_t)1);
            if (c->split_samples.fit(n_rect * stored_samples(chunk_len) * kSampleFloats) != hipSuccess) chunk_len = n_samples;
          }
        }
      }
      HIP_TRY(c->plan_parts.resize(n));
      HIP_TRY(c->plan_first.resize(n));
      HIP_TRY(c->plan_total.resize(1));
      if (!redo) HIP_TRY(c->plan_jobs.resize((size_t)n * (size_t)fewest + (size_t)lanes + 64));  // (with redo: sized before the probe, see there)
      size_t scan_bytes = 0;
      HIP_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, c->plan_parts.p, c->plan_first.p, 0u, n, rocprim::plus<uint32_t>(), c->stream));
      HIP_TRY(c->scan_temp.resize(scan_bytes > 0 ? scan_bytes : 1));
      HIP_TRY(hipMemsetAsync(c->plan_total.p, 0, sizeof(unsigned long long), c->stream));
      const unsigned grid = (n + 255) / 256;
      rt_plan_cost_kernel<<<grid < 256 ? grid : 256, 256, 0, c->stream>>>(c->carry_cost.p, n, c->plan_total.p);
      HIP_TRY(hipGetLastError());
      for (int chunk_begin = p->sample_begin; chunk_begin < p->sample_end && e == hipSuccess; chunk_begin += chunk_len) {
        const bool first_chunk = chunk_begin == p->sample_begin;
        const int main_begin = first_chunk ? p->sample_begin + n_probe : chunk_begin;
        const int main_end = std::min(chunk_begin + chunk_len, p->sample_end), main_samples = main_end - main_begin;
        const size_t want = split_budget / (stored_samples(main_samples) * kRecordBytes);
        uint32_t max_split = (uint32_t)(want > row_cap ? row_cap : want);
        if (p->rng_mode != RT_RNG_COUNTER) max_split = 0;
        // the store of the cut pixels' samples: when the device cannot give it, fewer pixels are cut, in the end none
        while (max_split > 0 && c->split_samples.fit((size_t)max_split * stored_samples(main_samples) * kSampleFloats) != hipSuccess) max_split /= 2;
        const int splittable = max_split > 0 ? 1 : 0;
        // every pixel has a row in the store: the main pass too runs in the kernel with rays in reserve
        // (the kernel's slot numbers are 32-bit: a budget beyond 64 GiB must not make a store of more records)
        const int all_rows = splittable && max_split >= n && stored_ok &&
                             (unsigned long long)n * stored_samples(main_samples) <= 0xFFFFFFFFull ? 1 : 0;
        if (chunk_len < n_samples && !all_rows) return fail(RT_ERR_DEVICE, "a chunk of %d samples has no row for every pixel", main_samples);
        if (redo && first_chunk && !all_rows) return fail(RT_ERR_DEVICE, "probe samples were left to a main pass that cannot store them");
        // ... whose records go out in pairs, as long as the store stays below 2^31 records (a slot number's top bit
        // marks the record that waits); RT_SAMPLE_PAIRS=0: every record on its own (comparison)
        static const bool pairs_enabled = [] {
          const char* e = tuning_env("RT_SAMPLE_PAIRS");
          return !(e && e[0] == '0');
        }();
        const int pairs = all_rows && pairs_enabled && (unsigned long long)n * (unsigned long long)main_samples < (1ull << 31) ? 1 : 0;
        rt_plan_count_kernel<<<grid, 256, 0, c->stream>>>(c->carry_cost.p, c->order_idx.p, n, c->plan_total.p, lanes, max_split, main_samples,
                                                         splittable, fewest, c->plan_parts.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(rocprim::exclusive_scan(c->scan_temp.p, scan_bytes, c->plan_parts.p, c->plan_first.p, 0u, n, rocprim::plus<uint32_t>(),
                                        c->stream));
        rt_plan_jobs_kernel<<<grid, 256, 0, c->stream>>>(c->order_idx.p, c->plan_parts.p, c->plan_first.p, n, main_begin,
                                                        main_end, all_rows, pairs, c->plan_jobs.p, c->plan_counts.p,
                                                        redo && first_chunk ? c->plan_counts.p + 1 : nullptr);
        HIP_TRY(hipGetLastError());
        render_args mainp = probe;
        mainp.pass = 2;
        mainp.p.sample_begin = main_begin;
        mainp.p.sample_end = main_end;
        mainp.jobs = c->plan_jobs.p;
        mainp.n_jobs = c->plan_counts.p;
        mainp.split_samples = reinterpret_cast<float4*>(c->split_samples.p);
        mainp.main_samples = main_samples;
        mainp.split_rows = n < max_split ? n : max_split;
        mainp.pairs = pairs;
        // Not every pixel has a row: the same kernel in the form whose lanes sum whole pixels in registers (counter
        // generator; RT_OWNED_RESERVE=0 keeps the kernel without reserves, for comparison)
        static const bool owned_enabled = [] {
          const char* e = tuning_env("RT_OWNED_RESERVE");
          return !(e && e[0] == '0');
        }();
        // (a slot number's top bit marks a whole pixel's sample there, and the pixel's number takes 29 bits)
        const bool owned = !all_rows && stored_ok && owned_enabled && n_rect <= (size_t)RT_TAG_PIXEL &&
                           (unsigned long long)mainp.split_rows * (unsigned long long)main_samples < (1ull << 31);
        if (owned) mainp.pairs = 0;
        e = all_rows ? launch_stored_pass(mainp) : (owned ? launch_stored_pass(mainp, true) : launch(mainp, p));
        if (e == hipSuccess && splittable) {
          const uint32_t cand = n < max_split ? n : max_split;
          // (the first chunk continues the probe's sums, a later one what the chunk before it left in the image)
          rt_sum_split_kernel<<<(cand + 255) / 256, 256, 0, c->stream>>>(c->order_idx.p, c->plan_parts.p, cand, first_chunk ? c->carry_colour.p : nullptr,
                                                                        reinterpret_cast<const float4*>(c->split_samples.p), main_samples, cand, p->tile_x0, p->tile_y0,
                                                                        tw, p->width, p->strip_count, p->strip_index, all_rows, pairs, c->accum(),
                                                                        redo && first_chunk ? c->carry_rng.p : nullptr,
                                                                        reinterpret_cast<const float4*>(c->probe_samples.p), n_probe, n,
                                                                        (size_t)cand * stored_samples(main_samples));
          e = hipGetLastError();
        }
      }
    }
  }
  if (e != hipSuccess) return fail(RT_ERR_DEVICE, "render launch: %s", hipGetErrorString(e));
  c->last_kernel = used_stored ? stored_name(c->search) : search_name(c->search, true);
  if (nested) return RT_OK;
  HIP_TRY(hipEventRecord(c->ev_k1, c->stream));
  c->have_kernel_time = true;
  return RT_OK;
}

}  // namespace

extern "C" {

int rt_render(rt_ctx* c, const rt_params* p) { return render_impl(c, p, false); }
int rt_render_counted(rt_ctx* c, const rt_params* p) { return render_impl(c, p, true); }

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

int rt_render_chunk(rt_ctx* c, const rt_params* p, int c_min_x, int c_min_y, int c_max_x, int c_max_y) {
  if (!p) return fail(RT_ERR_INVALID, "null params");
  rt_params q = *p;
  q.tile_x0 = c_min_x;
  q.tile_y0 = c_min_y;
  q.tile_x1 = c_max_x;
  q.tile_y1 = c_max_y;
  return render_impl(c, &q, false);
}

}  // extern "C"
