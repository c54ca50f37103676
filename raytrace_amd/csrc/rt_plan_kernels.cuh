// rt_plan_kernels.cuh — the small kernels of two-pass rendering: the cost sort key, the job plan of the main pass, and
// the in-order sums of stored samples.
#pragma once

#include "rt_render_kernels.cuh"

// ---- job planning for the main pass of two-pass rendering --------------------
// cost[p] = segments the probe pass spent on rect pixel p; order[i] = the pixel at sorted position i.

__global__ __launch_bounds__(256) void rt_plan_cost_kernel(const uint32_t* cost, uint32_t n, unsigned long long* total) {
  __shared__ unsigned long long part[4];
  unsigned long long sum = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) sum += cost[i];
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
  __syncthreads();
  // one atomic per workgroup: thousands of them on one address take longer than reading the costs
  if (threadIdx.x == 0) {
    sum = (part[0] + part[1]) + (part[2] + part[3]);
    if (sum != 0) atomicAdd(total, sum);
  }
}

// Jobs of the pixel at sorted position i: its share of the measured cost in units of 1/lanes of the
// total (`lanes` arrives multiplied by the fineness factor), so that no job is longer than that
// fraction of a lane's fair share; at least 8 samples each, at most 64 parts, only the first
// max_split positions (the most expensive pixels) and only when splittable.
RT_DEV uint32_t plan_parts(unsigned long long cost, unsigned long long total, uint32_t lanes, uint32_t i, uint32_t max_split,
                           int samples, int splittable, uint32_t min_parts) {
  if (!splittable || i >= max_split || total == 0ull) return 1u;
  unsigned long long parts = (cost * lanes + total - 1ull) / total;
  const unsigned long long most = (unsigned long long)max(samples / 8, 1);
  parts = parts < (unsigned long long)min_parts ? (unsigned long long)min_parts : parts;
  parts = parts > 64ull ? 64ull : parts;
  parts = parts > most ? most : parts;
  return (uint32_t)parts;
}

__global__ __launch_bounds__(256) void rt_plan_count_kernel(const uint32_t* cost, const uint32_t* order, uint32_t n,
                                                           const unsigned long long* total, uint32_t lanes, uint32_t max_split,
                                                           int samples, int splittable, uint32_t min_parts, uint32_t* parts) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) parts[i] = plan_parts(cost[order[i]], *total, lanes, i, max_split, samples, splittable, min_parts);
}

// first[i] = exclusive prefix sum of parts.  Writes the jobs of the pixels at sorted positions [256 b, 256 b + 256) — a
// run of consecutive jobs — and, from the last workgroup, the job count.  Thread t writes jobs t, t + 256, ... of the
// run (whole lines, side by side), finding each job's pixel by bisection of the workgroup's 257 prefix sums.
__global__ __launch_bounds__(256) void rt_plan_jobs_kernel(const uint32_t* order, const uint32_t* parts, const uint32_t* first,
                                                          uint32_t n, int sample_begin, int sample_end, int all_rows, int pairs,
                                                          uint4* jobs, uint32_t* n_jobs, const uint32_t* n_before) {
  // (the jobs rt_probe_sum_kernel has made of given-up probe samples stand first in the table)
  const uint32_t before = n_before ? *n_before : 0u;
  jobs += before;
  __shared__ uint32_t start[257];
  const uint32_t i0 = blockIdx.x * 256, i = i0 + threadIdx.x;
  const uint32_t last = min(i0 + 256u, n) - 1u;  // the workgroup's last pixel
  start[threadIdx.x] = i < n ? first[i] : 0xFFFFFFFFu;
  if (threadIdx.x == 0) start[256] = first[last] + parts[last];
  __syncthreads();
  const uint32_t run_begin = start[0], run_end = start[256], count = last - i0 + 1u;
  const int samples = sample_end - sample_begin;
  // `pairs`: jobs begin at even samples of the pass, so that the two records of a pair are one job's (parts <= samples / 8)
  const int unit = pairs ? 2 : 1, units = (samples + unit - 1) / unit;
  for (uint32_t j = run_begin + threadIdx.x; j < run_end; j += 256u) {
    uint32_t lo = 0, hi = count;  // the pixel lo with start[lo] <= j < start[lo + 1]
    while (hi - lo > 1u) {
      const uint32_t mid = (lo + hi) >> 1;
      if (start[mid] <= j) lo = mid; else hi = mid;
    }
    const uint32_t pos = i0 + lo, k = parts[pos], part = j - start[lo];
    const int s0 = sample_begin + min(samples, unit * (int)(((long long)units * part) / k));
    const int s1 = sample_begin + min(samples, unit * (int)(((long long)units * (part + 1)) / k));
    jobs[j] = make_uint4(order[pos], (uint32_t)s0, (uint32_t)s1, (k > 1 || all_rows) ? pos : RT_JOB_WHOLE);
  }
  if (last == n - 1u && threadIdx.x == 0) *n_jobs = before + run_end;
}

// Pixels that were cut into several jobs: the samples are added in sample order onto the sum the
// probe pass carried (carry_colour null: onto the pixel's sum in the image), exactly as one lane would have added them.
__global__ __launch_bounds__(256) void rt_sum_split_kernel(const uint32_t* order, const uint32_t* parts, uint32_t n_candidates,
                                                          const float* carry_colour, const float4* split_samples, int samples,
                                                          uint32_t n_rows, int tile_x0, int tile_y0, int tile_w, int width,
                                                          int strip_count, int strip_index, int all_rows, int pairs, float* accum,
                                                          const uint32_t* redone, const float4* probe_samples, int n_probe, uint32_t probe_rows,
                                                          size_t redo_base) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_candidates || (!all_rows && parts[i] <= 1u)) return;
  const uint32_t rect_pixel = order[i];
  const int x = tile_x0 + (int)(rect_pixel % (uint32_t)tile_w), y = real_row((int)(rect_pixel / (uint32_t)tile_w), tile_y0, strip_count, strip_index);
  float* dst = accum + 3 * ((size_t)y * (size_t)width + (size_t)x);
  const float* from = carry_colour ? carry_colour + 3 * (size_t)rect_pixel : dst;  // (no carried sums: the image's own)
  f3 colour = mk3(from[0], from[1], from[2]);
  if (redone && redone[rect_pixel] != 0u) {
    // some of the pixel's probe samples were rendered again by the main pass: the probe's sum anew, from the image
    colour = mk3(dst[0], dst[1], dst[2]);
    for (int s = 0; s < n_probe; s++) {
      float4 v = probe_samples[sample_slot((uint32_t)s, rect_pixel, probe_rows)];
      // (a sample given up: its record names the job that rendered it again, and that job wrote behind the rows)
      if ((__float_as_uint(v.w) & RT_COST_ABORTED) != 0u) {
        const size_t slot = redo_base + 2 * (size_t)__float_as_uint(v.x);
        if (pairs) {  // (24-byte pairs: the sample is the first of pair slot / 2)
          const float* q = reinterpret_cast<const float*>(split_samples) + 6 * (slot >> 1);
          v = make_float4(q[0], q[1], q[2], 0.0f);
        } else {
          v = split_samples[slot];
        }
      }
      colour = colour + mk3(v.x, v.y, v.z);
    }
  }
  if (pairs) {  // consecutive threads read consecutive 24-byte pairs of a plane (rt_render_stored_kernel writes them)
    for (int s = 0; s < samples; s += 2) {
      const float2* q = reinterpret_cast<const float2*>(split_samples) + 3 * (pair_slot((uint32_t)s, i, n_rows) >> 1);
      const float2 a = q[0], b = q[1], c = q[2];
      colour = colour + mk3(a.x, a.y, b.x);
      if (s + 1 < samples) colour = colour + mk3(b.y, c.x, c.y);
    }
  } else {
    for (int s = 0; s < samples; s++) {  // consecutive threads read consecutive lines of a group's plane
      const float4 v = split_samples[sample_slot((uint32_t)s, i, n_rows)];
      colour = colour + mk3(v.x, v.y, v.z);
    }
  }
  dst[0] = colour.x;
  dst[1] = colour.y;
  dst[2] = colour.z;
}

// After a pass in sweeps (pass 3): the pixel's samples are added in sample order onto what earlier
// sample ranges left in the image, exactly as one lane would have added them.  The sum goes to the
// carry arrays (a probe: the main pass continues from it) or, with carry_colour null, back into the
// image (a frame rendered in sweeps as a whole).
// Samples whose path the probe gave up (RT_COST_ABORTED) become jobs of one sample each at the head of the main pass's
// job table (redo_jobs, counted in n_redo: one atomic per workgroup; the job's number goes into the sample's record),
// and their pixel is marked in carry_rng, for rt_sum_split_kernel to form the probe's sum again once those samples
// are there.
__global__ __launch_bounds__(256) void rt_probe_sum_kernel(uint32_t n_rect, float4* probe_samples, int n_probe, int tile_x0,
                                                          int tile_y0, int tile_w, int width, int strip_count, int strip_index,
                                                          float* accum, float* carry_colour, uint32_t* carry_rng, uint32_t* record_cost,
                                                          uint4* redo_jobs, uint32_t* n_redo, int sample_begin) {
  __shared__ uint32_t wave_total[4];
  __shared__ uint32_t block_base;
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const bool mine = i < n_rect;
  f3 colour = mk3(0.0f, 0.0f, 0.0f);
  uint32_t cost = 0u, given_up = 0u;  // bit s: sample s was given up
  float* base = nullptr;
  if (mine) {
    const int x = tile_x0 + (int)(i % (uint32_t)tile_w), y = real_row((int)(i / (uint32_t)tile_w), tile_y0, strip_count, strip_index);
    base = accum + 3 * ((size_t)y * (size_t)width + (size_t)x);
    colour = mk3(base[0], base[1], base[2]);
    for (int s = 0; s < n_probe; s++) {  // consecutive threads read consecutive lines of a group's plane
      const float4 v = probe_samples[sample_slot((uint32_t)s, i, n_rect)];
      colour = colour + mk3(v.x, v.y, v.z);
      const uint32_t w = __float_as_uint(v.w);
      cost += w & ~RT_COST_ABORTED;
      if (redo_jobs != nullptr && (w & RT_COST_ABORTED) != 0u && s < 32) given_up |= 1u << s;
    }
    if (record_cost) record_cost[i] = cost;  // (records written by rt_render_stored_kernel carry the segments their path took)
    float* dst = carry_colour ? carry_colour + 3 * (size_t)i : base;
    dst[0] = colour.x;
    dst[1] = colour.y;
    dst[2] = colour.z;
    if (carry_rng) carry_rng[i] = given_up != 0u ? 1u : 0u;  // (the counter RNG re-keys every sample: the word is free for the mark)
  }
  if (redo_jobs == nullptr) return;
  // places for the workgroup's jobs: prefix sums over the wave, the waves, and one atomic for the workgroup
  const uint32_t mine_n = (uint32_t)__popc(given_up);
  uint32_t incl = mine_n;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(incl, off, 64);
    if ((int)(threadIdx.x & 63) >= off) incl += up;
  }
  if ((threadIdx.x & 63) == 63) wave_total[threadIdx.x >> 6] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t total = (wave_total[0] + wave_total[1]) + (wave_total[2] + wave_total[3]);
    block_base = total != 0u ? atomicAdd(n_redo, total) : 0u;
  }
  __syncthreads();
  uint32_t at = block_base + (incl - mine_n);
  for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) at += wave_total[w];
  while (given_up != 0u) {
    const int s = __builtin_ctz(given_up);
    given_up &= given_up - 1u;
    redo_jobs[at] = make_uint4(i, (uint32_t)(sample_begin + s), (uint32_t)(sample_begin + s + 1), RT_JOB_REDO | at);
    probe_samples[sample_slot((uint32_t)s, i, n_rect)].x = __uint_as_float(at);  // where rt_sum_split_kernel finds the sample
    at++;
  }
}

// Sort key of the main pass: the probe's cost in a few steps per power of two, most
// expensive first, and below it the pixel's 8x8 tile.  Pixels of about the same cost then stay
// together tile by tile, so that the lanes of a wave — which take consecutive jobs — follow
// neighbouring rays through the same part of the scene, where a sort by cost alone would scatter
// them over the image.
__global__ __launch_bounds__(256) void rt_sort_key_kernel(uint32_t n_rect, int tile_w, int log2_steps, int log2_tile, const uint32_t* carry_cost,
                                                         uint32_t* key, uint32_t* idx) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rect) return;
  const uint32_t x = i % (uint32_t)tile_w, y = i / (uint32_t)tile_w, tiles_x = (((uint32_t)tile_w - 1u) >> log2_tile) + 1u;
  uint32_t tile = (y >> log2_tile) * tiles_x + (x >> log2_tile);
  while (tile >> 20) tile >>= 1;  // (images beyond a million tiles: neighbouring tiles share a value)
  // 2^log2_steps levels per octave (log2_steps <= 6: fewer than 2048 levels over the 23 octaves)
  const uint32_t level = (__float_as_uint((float)(min(carry_cost[i], 0x7FFFFFu) + 1u)) >> (23 - log2_steps)) - (127u << log2_steps);
  key[i] = ((2047u - min(level, 2047u)) << 20) | tile;
  idx[i] = i;
}

