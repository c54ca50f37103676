// rt_eval_kernels.cuh — device functions of the path evaluated on caller-given values (test and tool entry points):
// the kernels' square root, the noise function and its gradient; and the fill of the noise texture's gradient table.
#pragma once

#include "rt_device.cuh"

// sqrt_rn against the compiler's correctly rounded sqrtf over a range of bit patterns (test entry point): counts the
// arguments whose results differ in any bit (two NaNs count as equal) and keeps one of them.
__global__ __launch_bounds__(256) void rt_check_sqrt_kernel(unsigned long long first, unsigned long long count, unsigned long long* mismatches,
                                                           uint32_t* example) {
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * 256) {
    const uint32_t bits = (uint32_t)(first + i);
    const float x = __uint_as_float(bits), a = sqrt_rn(x), b = sqrtf(x);
    if (__float_as_uint(a) != __float_as_uint(b) && !(a != a && b != b)) {
      atomicAdd(mismatches, 1ull);
      *example = bits;
    }
  }
}
__global__ __launch_bounds__(256) void rt_eval_sqrt_kernel(const float* x, size_t n, float* y) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = sqrt_rn(x[i]);
}

// Fills the gradient table of the noise texture: one lattice point per thread.
__global__ __launch_bounds__(256) void rt_perlin_table_kernel(float2* grad, int half) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, side = (size_t)(2 * half);
  if (i >= side * side) return;
  float gx, gy;
  perlin_gradient((int)(i % side) - half, (int)(i / side) - half, gx, gy);
  grad[i] = make_float2(gx, gy);
}

// perlin(x, y) of perlin.h:59-85 for a list of points, and randomGradient of perlin.h:31-43 for a list of
// lattice points: the device functions the noise-textured material uses, callable on their own.
__global__ __launch_bounds__(256) void rt_eval_perlin_kernel(const float2* xy, size_t n, perlin_table tab, float* value) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) value[i] = perlin2(xy[i].x, xy[i].y, tab);
}
__global__ __launch_bounds__(256) void rt_eval_gradient_kernel(const int2* ixy, size_t n, float2* g) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float gx, gy;
  perlin_gradient(ixy[i].x, ixy[i].y, gx, gy);
  g[i] = make_float2(gx, gy);
}

