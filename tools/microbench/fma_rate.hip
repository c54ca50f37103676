// FP32 VALU rate microbenchmark: independent v_fma_f32 chains (scalar and packed), all CUs busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int CHAINS>
__global__ __launch_bounds__(256) void fma_scalar(float* out, float a, float b, int iters) {
  float x[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; i++) x[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < CHAINS; i++) x[i] = __builtin_fmaf(x[i], a, b);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; i++) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CHAINS>
__global__ __launch_bounds__(256) void fma_packed(float* out, float a, float b, int iters) {
  float2v x[CHAINS];
  const float2v av = {a, a * 1.0001f}, bv = {b, b * 0.999f};
#pragma unroll
  for (int i = 0; i < CHAINS; i++) x[i] = float2v{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < CHAINS; i++) x[i] = __builtin_elementwise_fma(x[i], av, bv);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; i++) s += x[i].x + x[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CHAINS>
__global__ __launch_bounds__(256) void mul_add_scalar(float* out, float a, float b, int iters) {  // no FMA: mul then add
  float x[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; i++) x[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < CHAINS; i++) {
      float m = x[i] * a;
      asm volatile("" : "+v"(m));
      x[i] = m + b;
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; i++) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  const int blocks_per_cu[] = {1, 2, 4, 8};
  const int iters = 20000;
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * 4 * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int bpc : blocks_per_cu) {
    const int blocks = 256 * bpc;
    float ms;
    auto time = [&](auto launch) {
      launch();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      return ms;
    };
    double lanes = (double)blocks * 256;
    float t1 = time([&] { fma_scalar<8><<<blocks, 256>>>(out, 1.0001f, 0.5f, iters); });
    float t2 = time([&] { fma_packed<8><<<blocks, 256>>>(out, 1.0001f, 0.5f, iters); });
    float t3 = time([&] { mul_add_scalar<8><<<blocks, 256>>>(out, 1.0001f, 0.5f, iters); });
    printf("waves/SIMD=%d  v_fma_f32: %.1f TFLOP/s (%.2f ms)  v_pk_fma_f32: %.1f TFLOP/s (%.2f ms)  mul+add: %.1f Tinstr/s-lanes as TFLOP/s %.1f (%.2f ms)\n",
           bpc, lanes * iters * 8 * 2 / (t1 * 1e-3) / 1e12, t1, lanes * iters * 8 * 4 / (t2 * 1e-3) / 1e12, t2,
           lanes * iters * 8 * 2 / (t3 * 1e-3) / 1e12, lanes * iters * 8 * 2 / (t3 * 1e-3) / 1e12, t3);
  }
  return 0;
}
