// rt_eval.hip — entry points that evaluate device functions of the path on caller-given values (tests and tools),
// the gradient table of the noise texture, and the reference's seed image.

#include "rt_ctx.hpp"
#include "../../include/raytrace_amd_test.h"
#include "rt_eval_kernels.cuh"
#include "rt_seedtex.cuh"

namespace rt_impl {

// The gradient table of the noise texture does not depend on the scene: built once per context.
int ensure_noise_table(rt_ctx* c) {
  if (c->perlin_grad.p) return RT_OK;
  const size_t side = 2 * (size_t)kPerlinHalf;
  HIP_TRY(c->perlin_grad.resize(side * side));
  rt_perlin_table_kernel<<<(unsigned)((side * side + 255) / 256), 256, 0, c->stream>>>(c->perlin_grad.p, kPerlinHalf);
  HIP_TRY(hipGetLastError());
  return RT_OK;
}


}  // namespace rt_impl

extern "C" {

int rt_eval_perlin(rt_ctx* c, const float* xy, size_t n, int use_table, float* value) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (n == 0) return RT_OK;
  if (!xy || !value) return fail(RT_ERR_INVALID, "null point or result pointer");
  if (n > ((size_t)1 << 28)) return fail(RT_ERR_INVALID, "too many points");
  if (use_table) {
    rc = ensure_noise_table(c);
    if (rc) return rc;
  }
  dev_buf<float2> in;
  dev_buf<float> out;
  hipError_t e = in.resize(n);
  if (e == hipSuccess) e = out.resize(n);
  if (e == hipSuccess) e = hipMemcpyAsync(in.p, xy, n * sizeof(float2), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    perlin_table tab;
    tab.grad = use_table ? c->perlin_grad.p : nullptr;
    tab.half = kPerlinHalf;
    rt_eval_perlin_kernel<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(in.p, n, tab, out.p);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(value, out.p, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
  const hipError_t s = hipStreamSynchronize(c->stream);
  in.release();
  out.release();
  if (e == hipSuccess) e = s;
  if (e != hipSuccess) return fail(RT_ERR_DEVICE, "rt_eval_perlin: %s", hipGetErrorString(e));
  return RT_OK;
}

int rt_eval_perlin_gradient(rt_ctx* c, const int32_t* ixy, size_t n, float* gxy) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (n == 0) return RT_OK;
  if (!ixy || !gxy) return fail(RT_ERR_INVALID, "null point or result pointer");
  if (n > ((size_t)1 << 28)) return fail(RT_ERR_INVALID, "too many points");
  dev_buf<int2> in;
  dev_buf<float2> out;
  hipError_t e = in.resize(n);
  if (e == hipSuccess) e = out.resize(n);
  if (e == hipSuccess) e = hipMemcpyAsync(in.p, ixy, n * sizeof(int2), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    rt_eval_gradient_kernel<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(in.p, n, out.p);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(gxy, out.p, n * sizeof(float2), hipMemcpyDeviceToHost, c->stream);
  const hipError_t s = hipStreamSynchronize(c->stream);
  in.release();
  out.release();
  if (e == hipSuccess) e = s;
  if (e != hipSuccess) return fail(RT_ERR_DEVICE, "rt_eval_perlin_gradient: %s", hipGetErrorString(e));
  return RT_OK;
}

int rt_eval_sqrt(rt_ctx* c, const float* x, size_t n, float* y) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (n == 0) return RT_OK;
  if (!x || !y) return fail(RT_ERR_INVALID, "null argument or result pointer");
  if (n > ((size_t)1 << 28)) return fail(RT_ERR_INVALID, "too many values");
  dev_buf<float> in, out;
  hipError_t e = in.resize(n);
  if (e == hipSuccess) e = out.resize(n);
  if (e == hipSuccess) e = hipMemcpyAsync(in.p, x, n * sizeof(float), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    rt_eval_sqrt_kernel<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(in.p, n, out.p);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(y, out.p, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
  const hipError_t s = hipStreamSynchronize(c->stream);
  in.release();
  out.release();
  if (e == hipSuccess) e = s;
  if (e != hipSuccess) return fail(RT_ERR_DEVICE, "rt_eval_sqrt: %s", hipGetErrorString(e));
  return RT_OK;
}

int rt_check_sqrt(rt_ctx* c, uint64_t first_bits, uint64_t count, uint64_t* mismatches, uint32_t* example_bits) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!mismatches) return fail(RT_ERR_INVALID, "null result pointer");
  if (first_bits > 0xFFFFFFFFull || count > 0x100000000ull || first_bits + count > 0x100000000ull)
    return fail(RT_ERR_INVALID, "bit patterns beyond 32 bits");
  *mismatches = 0;
  if (example_bits) *example_bits = 0;
  if (count == 0) return RT_OK;
  dev_buf<unsigned long long> bad;
  dev_buf<uint32_t> ex;
  hipError_t e = bad.resize(1);
  if (e == hipSuccess) e = ex.resize(1);
  if (e == hipSuccess) e = hipMemsetAsync(bad.p, 0, sizeof(unsigned long long), c->stream);
  if (e == hipSuccess) e = hipMemsetAsync(ex.p, 0, sizeof(uint32_t), c->stream);
  if (e == hipSuccess) {
    const unsigned long long blocks = (count + 255) / 256;
    rt_check_sqrt_kernel<<<(unsigned)(blocks < 16384ull ? blocks : 16384ull), 256, 0, c->stream>>>(first_bits, count, bad.p, ex.p);
    e = hipGetLastError();
  }
  unsigned long long n_bad = 0;
  uint32_t one = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&n_bad, bad.p, sizeof(n_bad), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(&one, ex.p, sizeof(one), hipMemcpyDeviceToHost, c->stream);
  const hipError_t s = hipStreamSynchronize(c->stream);
  bad.release();
  ex.release();
  if (e == hipSuccess) e = s;
  if (e != hipSuccess) return fail(RT_ERR_DEVICE, "rt_check_sqrt: %s", hipGetErrorString(e));
  *mismatches = n_bad;
  if (example_bits) *example_bits = one;
  return RT_OK;
}

int rt_seed_texture(rt_ctx* c, int width, int height, float grid_size, float contrast, int layers, float lacunarity,
                    uint32_t* seed_tex) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (width < 1 || height < 1 || (int64_t)width * height > (int64_t)1 << 30) return fail(RT_ERR_INVALID, "bad image size %d x %d", width, height);
  if (layers < 0 || layers > 64) return fail(RT_ERR_INVALID, "bad layer count %d", layers);
  if (!(grid_size != 0.0f)) return fail(RT_ERR_INVALID, "grid size must not be zero");
  if (!seed_tex) return fail(RT_ERR_INVALID, "null output");
  const size_t n = (size_t)width * height;
  HIP_TRY(c->seed_tex.resize(n));
  rt_seed_texture_kernel<<<dim3((width + 63) / 64, (height + 3) / 4), 256, 0, c->stream>>>(width, height, grid_size, contrast, layers,
                                                                                           lacunarity, c->seed_tex.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(seed_tex, c->seed_tex.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RT_OK;
}

}  // extern "C"
