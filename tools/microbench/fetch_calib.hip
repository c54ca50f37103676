// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of this library's kernels.
// Every kernel moves a KNOWN number of bytes that no cache can hold back (buffers of 1 - 2 GiB, each byte touched
// once per launch), so counter / known bytes is the factor to apply to that pattern:
//   calib_stream16      16 bytes per lane, consecutive lanes consecutive: the wide coalesced read (rt_probe_sum_kernel,
//                       the unpaired sample store)
//   calib_pair32        every lane reads the two 16-byte halves of its own 32-byte slot, slots consecutive
//                       (rt_sum_split_kernel on paired records)
//   calib_gather64      groups of four lanes read one 64-byte node each, nodes chosen at random in a 1 GiB table
//                       (stage_nodes: the closest-hit kernel's node fetch), with plain loads
//   calib_gather64_cached   the same gathers over a 64 MiB table, which stays in the Infinity Cache (the closest-hit
//                       kernel's tree does): the rate at which L2 misses are answered from there
//   calib_gather128     groups of eight lanes read one 128-byte node (a whole line) each, chosen at random: what a tree
//                       of 128-byte nodes would ask of the fabric (1 GiB table, and 64 MiB as calib_gather128_cached)
//   calib_gather64_x4   four gathers per lane in flight (1 GiB table, and 64 MiB as calib_gather64_x4_cached)
//   calib_gather64_lds  the same through loads that write LDS directly, as stage_nodes issues them
//   calib_halfline      every workgroup reads the even 64-byte halves of a private 128 KiB region, then the odd halves:
//                       fabric read requests = lines if a miss of half a line fills the whole 128-byte line,
//                       2 x lines if it fills 64 bytes
//   calib_write16       16 bytes per lane streamed out
//   calib_write_pair    two 16-byte stores back to back into the lane's own 32-byte slot, slots consecutive
//   calib_write_pair_scattered   the same into slots chosen at random (a sample pair written alone)
// Run plain for the rates (HIP events), and under `rocprofv3 --pmc ...` for the counters (tools/fetch_calib.sh).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                         \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                           \
    }                                                                                    \
  } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(256) void calib_stream16(const float4* src, size_t n16, float* sink) {
  float s = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const float4 v = src[i];
    s += (v.x + v.y) + (v.z + v.w);
  }
  if (s == 12345.678f) sink[0] = s;
}

__global__ __launch_bounds__(256) void calib_pair32(const float4* src, size_t n_pairs, float* sink) {
  float s = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_pairs; i += (size_t)gridDim.x * 256) {
    const float4* q = src + 2 * i;
    const float4 v0 = q[0];
    s += (v0.x + v0.y) + v0.z;
    const float4 v1 = q[1];
    s += (v1.x + v1.y) + v1.z;
  }
  if (s == 12345.678f) sink[0] = s;
}

// node_mask + 1 nodes of 64 bytes; gather g of the launch reads node mix(g) & node_mask
__global__ __launch_bounds__(256) void calib_gather64(const float4* table, uint32_t node_mask, size_t n_gathers, float* sink) {
  float s = 0.0f;
  const uint32_t sub = threadIdx.x & 3u;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < 4 * n_gathers; t += (size_t)gridDim.x * 256) {
    const uint32_t node = mix((uint32_t)(t >> 2) * 2654435761u + 12345u) & node_mask;
    const float4 v = table[(size_t)node * 4 + sub];
    s += v.x + v.w;
  }
  if (s == 12345.678f) sink[0] = s;
}

// node_mask + 1 nodes of 128 bytes
__global__ __launch_bounds__(256) void calib_gather128(const float4* table, uint32_t node_mask, size_t n_gathers, float* sink) {
  float s = 0.0f;
  const uint32_t sub = threadIdx.x & 7u;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < 8 * n_gathers; t += (size_t)gridDim.x * 256) {
    const uint32_t node = mix((uint32_t)(t >> 3) * 2654435761u + 12345u) & node_mask;
    const float4 v = table[(size_t)node * 8 + sub];
    s += v.x + v.w;
  }
  if (s == 12345.678f) sink[0] = s;
}

// four gathers per lane in flight at once (the request rate with more memory-level parallelism than one load per lane)
__global__ __launch_bounds__(256) void calib_gather64_x4(const float4* table, uint32_t node_mask, size_t n_gathers, float* sink) {
  float s = 0.0f;
  const uint32_t sub = threadIdx.x & 3u;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t + 3 * stride < 4 * n_gathers; t += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t node = mix((uint32_t)((t + u * stride) >> 2) * 2654435761u + 12345u) & node_mask;
      v[u] = table[(size_t)node * 4 + sub];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) s += v[u].x + v[u].w;
  }
  if (s == 12345.678f) sink[0] = s;
}

__global__ __launch_bounds__(256) void calib_gather64_cached(const float4* table, uint32_t node_mask, size_t n_gathers, float* sink) {
  float s = 0.0f;
  const uint32_t sub = threadIdx.x & 3u;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < 4 * n_gathers; t += (size_t)gridDim.x * 256) {
    const uint32_t node = mix((uint32_t)(t >> 2) * 2654435761u + 12345u) & node_mask;
    const float4 v = table[(size_t)node * 4 + sub];
    s += v.x + v.w;
  }
  if (s == 12345.678f) sink[0] = s;
}

__global__ __launch_bounds__(256) void calib_gather64_lds(const float4* table, uint32_t node_mask, size_t n_gathers, float* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char stage[4][1040];
  float s = 0.0f;
  const uint32_t sub = threadIdx.x & 3u, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < 4 * n_gathers; t += (size_t)gridDim.x * 256) {
    const uint32_t node = mix((uint32_t)(t >> 2) * 2654435761u + 12345u) & node_mask;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(table + (size_t)node * 4 + sub),
                                     (__attribute__((address_space(3))) void*)(&stage[wave][0]), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const float4 v = *reinterpret_cast<const float4*>(&stage[wave][lane * 16]);
    s += v.x + v.w;
    __builtin_amdgcn_wave_barrier();
  }
  if (s == 12345.678f) sink[0] = s;
}

// region of workgroup b: lines [1024 b, 1024 b + 1024) of 128 bytes; 256 threads read 16 bytes each = 32 halves per trip
__global__ __launch_bounds__(256) void calib_halfline(const float4* table, float* sink) {
  const float4* region = table + (size_t)blockIdx.x * 1024 * 8;  // 8 float4 per line
  const uint32_t sub = threadIdx.x & 3u, group = threadIdx.x >> 2;  // 64 groups of four lanes
  float s = 0.0f;
  for (int half = 0; half < 2; half++) {
    for (uint32_t line = group; line < 1024u; line += 64u) {
      const float4 v = region[(size_t)line * 8 + half * 4 + sub];
      s += v.x + v.w;
    }
    if (s == 12345.678f) sink[0] = s;  // the loads are consumed before the barrier
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void calib_write16(float4* dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
    dst[i] = make_float4((float)i, 1.0f, 2.0f, 3.0f);
}

__global__ __launch_bounds__(256) void calib_write_pair(float4* dst, size_t n_pairs) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_pairs; i += (size_t)gridDim.x * 256) {
    dst[2 * i] = make_float4((float)i, 1.0f, 2.0f, 0.0f);
    dst[2 * i + 1] = make_float4((float)i, 4.0f, 5.0f, 6.0f);
  }
}

// every slot of the buffer exactly once, in an order that is a bijection of the index (odd multiplier modulo 2^k)
__global__ __launch_bounds__(256) void calib_write_pair_scattered(float4* dst, uint32_t pair_mask, size_t n_pairs) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_pairs; i += (size_t)gridDim.x * 256) {
    const size_t slot = ((uint32_t)i * 2654435761u + 977u) & pair_mask;
    dst[2 * slot] = make_float4((float)i, 1.0f, 2.0f, 0.0f);
    dst[2 * slot + 1] = make_float4((float)i, 4.0f, 5.0f, 6.0f);
  }
}

int main() {
  const size_t bytes = (size_t)2 << 30;           // streamed buffers
  const size_t table_bytes = (size_t)1 << 30;     // gather table: four times the Infinity Cache
  const uint32_t node_mask = (uint32_t)(table_bytes / 64 - 1);
  const size_t n_gathers = (size_t)1 << 26;       // 4 GiB asked for in 64-byte nodes
  const int halfline_blocks = 2048;               // 2 097 152 lines, 256 MiB
  float4 *buf, *table;
  float* sink;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&table, table_bytes));
  CHECK(hipMalloc(&sink, 256));
  CHECK(hipMemset(buf, 0, bytes));
  CHECK(hipMemset(table, 0, table_bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int grid = 256 * 16;
  auto timed = [&](const char* name, double known_bytes, auto launch) {
    launch();  // untimed first launch (the profiler's per-dispatch rows keep both)
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s known_bytes %.0f  ms %.3f  GB/s %.1f\n", name, known_bytes, ms, known_bytes / ms / 1e6);
  };
  timed("calib_stream16", (double)bytes, [&] { calib_stream16<<<grid, 256>>>(buf, bytes / 16, sink); });
  timed("calib_pair32", (double)bytes, [&] { calib_pair32<<<grid, 256>>>(buf, bytes / 32, sink); });
  timed("calib_gather64", (double)n_gathers * 64.0, [&] { calib_gather64<<<grid, 256>>>(table, node_mask, n_gathers, sink); });
  timed("calib_gather128", (double)n_gathers * 64.0, [&] { calib_gather128<<<grid, 256>>>(table, (uint32_t)(table_bytes / 128 - 1), n_gathers / 2, sink); });
  timed("calib_gather128_cached", (double)n_gathers * 64.0, [&] { calib_gather128<<<grid, 256>>>(table, (uint32_t)(((size_t)64 << 20) / 128 - 1), n_gathers / 2, sink); });
  timed("calib_gather64_x4", (double)n_gathers * 64.0, [&] { calib_gather64_x4<<<grid, 256>>>(table, node_mask, n_gathers, sink); });
  timed("calib_gather64_x4_cached", (double)n_gathers * 64.0, [&] { calib_gather64_x4<<<grid, 256>>>(table, (uint32_t)(((size_t)64 << 20) / 64 - 1), n_gathers, sink); });
  // the same gathers over 64 MiB, which the Infinity Cache holds: known_bytes is what is asked for, not what HBM gives
  timed("calib_gather64_cached", (double)n_gathers * 64.0, [&] { calib_gather64_cached<<<grid, 256>>>(table, (uint32_t)(((size_t)64 << 20) / 64 - 1), n_gathers, sink); });
  timed("calib_gather64_lds", (double)n_gathers * 64.0, [&] { calib_gather64_lds<<<grid, 256>>>(table, node_mask, n_gathers, sink); });
  timed("calib_halfline", (double)halfline_blocks * 1024.0 * 128.0, [&] { calib_halfline<<<halfline_blocks, 256>>>(table, sink); });
  timed("calib_write16", (double)bytes, [&] { calib_write16<<<grid, 256>>>(buf, bytes / 16); });
  timed("calib_write_pair", (double)bytes, [&] { calib_write_pair<<<grid, 256>>>(buf, bytes / 32); });
  timed("calib_write_pair_scattered", (double)bytes, [&] { calib_write_pair_scattered<<<grid, 256>>>(buf, (uint32_t)(bytes / 32 - 1), bytes / 32); });
  CHECK(hipFree(buf));
  CHECK(hipFree(table));
  CHECK(hipFree(sink));
  return 0;
}
