// rt_seedtex.cuh — the seed image of the reference: shaders/perlin.fs rendered
// into an RGB8 texture (raytrace.cpp:309-328) and sampled by the ray shader
// (testFragment.fs:105-106).  One thread per ray-shader pixel; the output word
// is what floatBitsToUint(tex.r + tex.g + tex.b) yields for that pixel.
#pragma once

#include "rt_device.cuh"

// sin() of GLSL is driver-defined; here it is the double-precision sine
// rounded to binary32 (the same choice the CPU checker makes).
RT_DEV float seed_sin(float x) { return (float)sin((double)x); }
RT_DEV float seed_fract(float x) { return x - floorf(x); }

RT_DEV float seed_mix(float a0, float a1, float w) {  // perlin.fs:44-46
  return (a1 - a0) * (3.0f - w * 2.0f) * w * w + a0;
}

RT_DEV float seed_corner(int ix, int iy, float x, float y) {  // perlin.fs:48-62
  const float ux = (float)ix, uy = (float)iy;
  const float gx = -1.0f + 2.0f * seed_fract(seed_sin(ux * 127.1f + uy * 311.7f) * 43758.5453123f);
  const float gy = -1.0f + 2.0f * seed_fract(seed_sin(ux * 269.5f + uy * 183.3f) * 43758.5453123f);
  return (x - ux) * gx + (y - uy) * gy;
}

RT_DEV float seed_noise(float x, float y) {  // perlin.fs:64-87
  const float fx = floorf(x), fy = floorf(y);
  const int x0 = (int)fx, y0 = (int)fy;
  const float sx = x - fx, sy = y - fy;
  const float top = seed_mix(seed_corner(x0, y0, x, y), seed_corner(x0 + 1, y0, x, y), sx);
  const float bot = seed_mix(seed_corner(x0, y0 + 1, x, y), seed_corner(x0 + 1, y0 + 1, x, y), sx);
  return seed_mix(top, bot, sy);
}

__global__ __launch_bounds__(256) void rt_seed_texture_kernel(int width, int height, float grid, float contrast, int layers,
                                                             float lacunarity, uint32_t* out) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= width || y >= height) return;
  // the ray shader's origin is the upper left (fs:3), the texture's the lower left
  const float px = (float)x + 0.5f, py = (float)(height - 1 - y) + 0.5f;
  float value = 0.0f, freq = 1.0f, amp = 1.0f;
  for (int i = 0; i < layers; i++) {  // perlin.fs:21-26
    value += seed_noise(px * freq / grid, py * freq / grid) * amp;
    freq *= lacunarity;
    amp /= 2.0f;
  }
  value = value * contrast;  // perlin.fs:28-37
  value = value > 1.0f ? 1.0f : value;
  value = value < -1.0f ? -1.0f : value;
  value = value * 0.5f + 0.5f;
  const float c = floorf(value * 255.0f + 0.5f) / 255.0f;  // RGB8 store, normalised fetch
  out[(size_t)y * width + x] = __float_as_uint((c + c) + c);
}
