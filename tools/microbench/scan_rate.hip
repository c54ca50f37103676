// Sphere-scan rate microbenchmark: every lane holds one ray and tests it against N sphere
// records (centre, r^2) staged in LDS, the inner loop of the linear closest-hit search in
// isolation.  Variants differ only in control flow around the discriminant:
//   count     no branch: counts discriminants >= 0
//   branch1   one branch per sphere around a (rare) root evaluation
//   branch4   discriminants of four spheres, one branch per group
//   mask4     as branch4, group test on the wave-wide ballot held in scalar registers
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o scan_rate scan_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define DEV __device__ __forceinline__

struct ray {
  float ox, oy, oz, dx, dy, dz;
};

DEV float disc_of(const float4 s, const ray& r, float a, float& h) {
  const float x = s.x - r.ox, y = s.y - r.oy, z = s.z - r.oz;
  h = (r.dx * x + r.dy * y) + r.dz * z;
  const float c = ((x * x + y * y) + z * z) - s.w;
  return h * h - a * c;
}

DEV ray make_ray(unsigned id) {
  unsigned s = id * 747796405u + 2891336453u;
  auto next = [&]() {
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return (s >> 9) * (1.0f / 8388608.0f);
  };
  ray r;
  r.ox = 13.f + next() * 0.1f;
  r.oy = 2.f + next() * 0.1f;
  r.oz = 3.f + next() * 0.1f;
  r.dx = -13.f + (next() - 0.5f) * 8.f;
  r.dy = -2.f + (next() - 0.5f) * 3.f;
  r.dz = -3.f + (next() - 0.5f) * 8.f;
  return r;
}

extern __shared__ float4 lds_sph[];

DEV void stage(const float4* sph, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) lds_sph[i] = sph[i];
  __syncthreads();
}

DEV void root(float disc, float h, float inv_a, int i, float& best, int& idx) {
  const float sq = sqrtf(disc);
  float t = (h - sq) * inv_a;
  if (t < 0.001f) t = (h + sq) * inv_a;
  if (t > 0.001f && t < best) {
    best = t;
    idx = i;
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void scan(const float4* sph, int n, int reps, float* out_t, int* out_i) {
  stage(sph, n);
  const unsigned id = blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  int acci = 0;
  for (int rep = 0; rep < reps; rep++) {
    const ray r = make_ray(id * 31u + rep);
    const float a = (r.dx * r.dx + r.dy * r.dy) + r.dz * r.dz;
    const float inv_a = 1.0f / a;
    float best = 3.0e38f;
    int idx = -1;
    if (MODE == 0) {
      int cnt = 0;
      for (int i = 0; i < n; i++) {
        float h;
        const float d = disc_of(lds_sph[i], r, a, h);
        cnt += d >= 0.f ? 1 : 0;
      }
      idx = cnt;
    } else if (MODE == 1) {
      for (int i = 0; i < n; i++) {
        float h;
        const float d = disc_of(lds_sph[i], r, a, h);
        if (d >= 0.f) root(d, h, inv_a, i, best, idx);
      }
    } else if (MODE == 2) {
      int i = 0;
      for (; i + 4 <= n; i += 4) {
        float h0, h1, h2, h3;
        const float d0 = disc_of(lds_sph[i], r, a, h0), d1 = disc_of(lds_sph[i + 1], r, a, h1);
        const float d2 = disc_of(lds_sph[i + 2], r, a, h2), d3 = disc_of(lds_sph[i + 3], r, a, h3);
        if (fmaxf(fmaxf(d0, d1), fmaxf(d2, d3)) >= 0.f) {
          if (d0 >= 0.f) root(d0, h0, inv_a, i, best, idx);
          if (d1 >= 0.f) root(d1, h1, inv_a, i + 1, best, idx);
          if (d2 >= 0.f) root(d2, h2, inv_a, i + 2, best, idx);
          if (d3 >= 0.f) root(d3, h3, inv_a, i + 3, best, idx);
        }
      }
      for (; i < n; i++) {
        float h;
        const float d = disc_of(lds_sph[i], r, a, h);
        if (d >= 0.f) root(d, h, inv_a, i, best, idx);
      }
    } else {
      int i = 0;
      for (; i + 4 <= n; i += 4) {
        float h0, h1, h2, h3;
        const float d0 = disc_of(lds_sph[i], r, a, h0), d1 = disc_of(lds_sph[i + 1], r, a, h1);
        const float d2 = disc_of(lds_sph[i + 2], r, a, h2), d3 = disc_of(lds_sph[i + 3], r, a, h3);
        const float m = fmaxf(fmaxf(d0, d1), fmaxf(d2, d3));
        if (__ballot(m >= 0.f) != 0ull) {  // wave-uniform: a scalar branch, exec untouched on the common path
          if (d0 >= 0.f) root(d0, h0, inv_a, i, best, idx);
          if (d1 >= 0.f) root(d1, h1, inv_a, i + 1, best, idx);
          if (d2 >= 0.f) root(d2, h2, inv_a, i + 2, best, idx);
          if (d3 >= 0.f) root(d3, h3, inv_a, i + 3, best, idx);
        }
      }
      for (; i < n; i++) {
        float h;
        const float d = disc_of(lds_sph[i], r, a, h);
        if (d >= 0.f) root(d, h, inv_a, i, best, idx);
      }
    }
    acc += best;
    acci += idx;
  }
  out_t[id] = acc;
  out_i[id] = acci;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 487;
  const int reps = argc > 2 ? atoi(argv[2]) : 64;
  std::vector<float4> h(n);
  srand(1);
  auto u = [] { return rand() / (float)RAND_MAX; };
  h[0] = float4{0.f, -1000.f, 0.f, 1000.f * 1000.f};
  for (int i = 1; i < n; i++) {
    const int a = (i % 22) - 11, b = (i / 22) - 11;
    h[i] = float4{a + 0.9f * u(), 0.2f, b + 0.9f * u(), 0.04f};
  }
  float4* d_sph;
  float* d_t;
  int* d_i;
  const int max_blocks = 256 * 8;
  hipMalloc(&d_sph, n * sizeof(float4));
  hipMalloc(&d_t, max_blocks * 256 * 4);
  hipMalloc(&d_i, max_blocks * 256 * 4);
  hipMemcpy(d_sph, h.data(), n * sizeof(float4), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[] = {"count", "branch1", "branch4", "mask4"};
  for (int bpc : {2, 4, 8}) {
    const int blocks = 256 * bpc;
    for (int mode = 0; mode < 4; mode++) {
      auto launch = [&] {
        const size_t lds = n * sizeof(float4);
        switch (mode) {
          case 0: scan<0><<<blocks, 256, lds>>>(d_sph, n, reps, d_t, d_i); break;
          case 1: scan<1><<<blocks, 256, lds>>>(d_sph, n, reps, d_t, d_i); break;
          case 2: scan<2><<<blocks, 256, lds>>>(d_sph, n, reps, d_t, d_i); break;
          default: scan<3><<<blocks, 256, lds>>>(d_sph, n, reps, d_t, d_i); break;
        }
      };
      launch();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double tests = (double)blocks * 256 * reps * n;
      const double cyc_per_wave_test = ms * 1e-3 * 2.4e9 * 1024 / (tests / 64);
      printf("waves/SIMD=%d %-8s %8.3f ms  %7.1f Gtests/s  %5.1f TFLOP/s (23/test)  %5.1f SIMD-cycles per wave-test\n", bpc,
             names[mode], ms, tests / ms * 1e-6, tests * 23 / ms * 1e-9, cyc_per_wave_test);
    }
  }
  return 0;
}
