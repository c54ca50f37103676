// rt_api.hip — contexts, the accumulator and its readback, closest-hit batches and the queries of the C ABI of
// raytrace_amd.h.  (Scene upload: rt_build.hip; rt_render: rt_render.hip.)

#include "rt_ctx.hpp"
#include "rt_trace_kernels.cuh"

namespace {
thread_local std::string g_error;
}

namespace rt_impl {
int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}
}  // namespace rt_impl

// ---- fixed scratch of rt_trace_batch_noalloc ---------------------------------
// The closest-hit kernel needs device memory beyond its inputs and outputs: the head of its ray queue and, for trees
// read from HBM, the part of the lanes' traversal stacks that does not fit LDS.  rt_trace_batch_device takes both from
// buffers the context allocates on demand; the entry point that must not allocate takes them from this array, which
// every GPU the library runs on holds (at its own address) until the process ends.
//   [0, kNoAllocHeaderBytes)       queue head (8 bytes used)
//   [kNoAllocHeaderBytes, end)     traversal-stack overflow, one slice per workgroup of the launch
static __device__ __align__(256) unsigned char rt_noalloc_scratch[RT_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr size_t kNoAllocHeaderBytes = 256;
// Device address of rt_noalloc_scratch on each device, looked up by rt_init with that device current (the array's
// name is only a host-side handle: kernels and memsets get these addresses).
unsigned char* g_noalloc_scratch[kMaxDevices] = {};
}  // namespace


namespace {

// `fixed` (nullable): memory the traversal-stack overflow must come from instead of a buffer of the context
// (fixed_bytes of it); hipErrorOutOfMemory when the tree's depth needs more than it holds.
template <int SEARCH, bool COUNTED>
hipError_t launch_trace_one(rt_ctx* c, hipStream_t st, trace_args a, unsigned char* fixed, size_t fixed_bytes) {
  using T = search_traits<SEARCH>;
  const size_t lds = search_lds_bytes<SEARCH>(counts_of(c));
  hipError_t e = allow_lds(rt_trace_kernel<SEARCH, COUNTED>, lds);
  if (e != hipSuccess) return e;
  int per_cu = 0;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rt_trace_kernel<SEARCH, COUNTED>, T::threads, lds);
  if (e != hipSuccess) return e;
  if (per_cu < 1) per_cu = 1;
  long long blocks = (long long)per_cu * c->n_cu;
  const long long needed = (long long)((a.n + T::threads - 1) / T::threads);
  if (blocks > needed) blocks = needed;
  if (blocks < 1) blocks = 1;
  if (fixed) {
    if (!attach_overflow_fixed<SEARCH>(c, blocks, a.sc, fixed, fixed_bytes)) return hipErrorOutOfMemory;
  } else {
    e = attach_overflow<SEARCH>(c, blocks, a.sc);
    if (e != hipSuccess) return e;
  }
  // Nothing has been queued so far: a refusal above leaves the stream and the timing events as they were.  The queue
  // head must start at zero (the fixed scratch keeps what the last call left; the overflow slices are written before
  // they are read).  Under stream capture the timing events are left alone: an event recorded into a graph cannot be
  // queried afterwards.
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &capturing) != hipSuccess) {
    (void)hipGetLastError();
    capturing = hipStreamCaptureStatusNone;
  }
  const bool timed = capturing == hipStreamCaptureStatusNone;
  e = hipMemsetAsync(a.queue, 0, sizeof(unsigned long long), st);
  if (e == hipSuccess && timed) e = hipEventRecord(c->ev_k0, st);
  if (e != hipSuccess) return e;
  rt_trace_kernel<SEARCH, COUNTED><<<dim3((unsigned)blocks), T::threads, lds, st>>>(a);
  e = hipGetLastError();
  if (e == hipSuccess && timed) e = hipEventRecord(c->ev_k1, st);
  if (e == hipSuccess) {
    c->last_kernel = search_name(SEARCH, false);
    c->have_kernel_time = timed;
  }
  return e;
}

template <int SEARCH>
hipError_t launch_trace(rt_ctx* c, bool counted, hipStream_t st, const trace_args& a, unsigned char* fixed = nullptr,
                        size_t fixed_bytes = 0) {
  return counted ? launch_trace_one<SEARCH, true>(c, st, a, fixed, fixed_bytes)
                 : launch_trace_one<SEARCH, false>(c, st, a, fixed, fixed_bytes);
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_error.c_str(); }

int rt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int rt_init(int device, rt_ctx** out) {
  if (!out) return fail(RT_ERR_INVALID, "null out pointer");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n < 1)
    return fail(RT_ERR_NO_GPU, "no HIP device visible (%s); this library has no CPU path",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  if (device < 0 || device >= n) return fail(RT_ERR_INVALID, "device %d out of range [0,%d)", device, n);
  HIP_TRY(hipSetDevice(device));
  if (device < kMaxDevices && !g_noalloc_scratch[device]) {  // once per device, with that device current
    void* ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(rt_noalloc_scratch)));
    g_noalloc_scratch[device] = static_cast<unsigned char*>(ptr);
  }
  rt_ctx* c = new rt_ctx();
  c->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
  if (c->n_cu < 1) c->n_cu = 256;
  hipError_t err = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (err == hipSuccess) err = hipEventCreate(&c->ev_k0);
  if (err == hipSuccess) err = hipEventCreate(&c->ev_k1);
  if (err == hipSuccess) err = hipEventCreate(&c->ev_b0);
  if (err == hipSuccess) err = hipEventCreate(&c->ev_b1);
  if (err != hipSuccess) {
    rt_destroy(c);
    return fail(RT_ERR_DEVICE, "context set-up: %s", hipGetErrorString(err));
  }
  *out = c;
  return RT_OK;
}

void rt_destroy(rt_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  (void)rt_comm_destroy(c);
  c->sph.release(); c->inv_r.release(); c->sph_mat.release(); c->mat_type.release(); c->mat_a.release();
  c->quad_qd.release(); c->quad_u.release(); c->quad_v.release(); c->quad_n.release(); c->quad_w.release(); c->quad_idx.release(); c->quad_of.release();
  c->big_sph.release(); c->big_idx.release(); c->tree_sph.release(); c->tree_r.release(); c->tree_orig.release();
  c->nodes.release(); c->qnodes.release(); c->leaf_sph.release(); c->leaf_r.release(); c->leaf_idx.release(); c->leaf_rec.release();
  c->accum_own.release(); c->seed_tex.release(); c->queue.release(); c->counters.release(); c->ovf.release(); c->build.release(); c->perlin_grad.release();
  c->carry_colour.release(); c->carry_rng.release(); c->carry_cost.release(); c->carry_key.release(); c->carry_idx.release();
  c->plan_parts.release(); c->plan_first.release(); c->plan_counts.release(); c->plan_total.release();
  c->plan_jobs.release(); c->split_samples.release(); c->probe_samples.release(); c->scan_temp.release();
  c->order_key.release(); c->order_idx.release(); c->sort_temp.release(); c->out_linear.release(); c->out_rgb8.release();
  if (c->ev_k0) (void)hipEventDestroy(c->ev_k0);
  if (c->ev_k1) (void)hipEventDestroy(c->ev_k1);
  if (c->ev_b0) (void)hipEventDestroy(c->ev_b0);
  if (c->ev_b1) (void)hipEventDestroy(c->ev_b1);
  for (int b = 0; b < 2; b++) {
    if (c->ev_in[b]) (void)hipEventDestroy(c->ev_in[b]);
    if (c->ev_run[b]) (void)hipEventDestroy(c->ev_run[b]);
    if (c->ev_out[b]) (void)hipEventDestroy(c->ev_out[b]);
    c->tb_o[b].release(); c->tb_d[b].release(); c->tb_t[b].release(); c->tb_n[b].release(); c->tb_id[b].release();
  }
  if (c->copy_in) (void)hipStreamDestroy(c->copy_in);
  if (c->copy_out) (void)hipStreamDestroy(c->copy_out);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

}  // extern "C"

extern "C" {

int rt_clear(rt_ctx* c) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (c->accum() && c->acc_w > 0)
    HIP_TRY(hipMemsetAsync(c->accum(), 0, 3 * (size_t)c->acc_w * (size_t)c->acc_h * sizeof(float), c->stream));
  c->samples_lo = c->samples_hi = 0;
  c->samples_override = -1;
  return RT_OK;
}

int rt_set_accumulator(rt_ctx* c, void* device_rgb, int width, int height) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (device_rgb && (width < 1 || height < 1)) return fail(RT_ERR_INVALID, "bad accumulator size");
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->accum_ext = static_cast<float*>(device_rgb);
  if (device_rgb) {
    c->acc_w = width;
    c->acc_h = height;
  } else {
    c->acc_w = c->acc_h = 0;  // the own buffer is re-created by the next render
  }
  c->samples_lo = c->samples_hi = 0;
  c->samples_override = -1;
  return RT_OK;
}

int rt_set_accumulated_samples(rt_ctx* c, int samples) {
  if (!c) return fail(RT_ERR_INVALID, "null context");
  if (samples < 1) return fail(RT_ERR_INVALID, "samples must be positive");
  c->samples_override = samples;
  return RT_OK;
}

int rt_accumulator_read(rt_ctx* c, float* sums, int* width, int* height, int* samples) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->accum() || c->acc_w < 1) return fail(RT_ERR_STATE, "nothing rendered yet");
  if (width) *width = c->acc_w;
  if (height) *height = c->acc_h;
  if (samples) *samples = c->samples_override > 0 ? c->samples_override : c->samples_hi - c->samples_lo;
  if (sums) {
    const size_t nv = 3 * (size_t)c->acc_w * (size_t)c->acc_h;
    HIP_TRY(hipMemcpyAsync(sums, c->accum(), nv * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RT_OK;
}

int rt_accumulator_write(rt_ctx* c, const float* sums, int width, int height, int samples) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!sums || width < 1 || height < 1 || (int64_t)width * height > (int64_t)1 << 30) return fail(RT_ERR_INVALID, "bad accumulator image");
  if (samples < 0) return fail(RT_ERR_INVALID, "negative sample count");
  const size_t nv = 3 * (size_t)width * (size_t)height;
  if (c->accum_ext) {
    if (width != c->acc_w || height != c->acc_h)
      return fail(RT_ERR_STATE, "image %d x %d does not match the caller's accumulator %d x %d", width, height, c->acc_w, c->acc_h);
  } else {
    HIP_TRY(c->accum_own.resize(nv));
    c->acc_w = width;
    c->acc_h = height;
  }
  HIP_TRY(hipMemcpyAsync(c->accum(), sums, nv * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->samples_lo = 0;
  c->samples_hi = samples;
  c->samples_override = -1;
  return RT_OK;
}

int rt_readback(rt_ctx* c, float* rgb_linear, uint8_t* rgb8) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->accum() || c->acc_w < 1) return fail(RT_ERR_STATE, "nothing rendered yet");
  const int samples = c->samples_override > 0 ? c->samples_override : c->samples_hi - c->samples_lo;
  if (samples < 1) return fail(RT_ERR_STATE, "no samples accumulated");
  const size_t nv = 3 * (size_t)c->acc_w * (size_t)c->acc_h;
  if (rgb_linear) HIP_TRY(c->out_linear.resize(nv));
  if (rgb8) HIP_TRY(c->out_rgb8.resize(nv));
  if (rgb_linear || rgb8) {
    rt_resolve_kernel<<<(unsigned)((nv + 255) / 256), 256, 0, c->stream>>>(
        c->accum(), nv, 1.0f / (float)samples, rgb_linear ? c->out_linear.p : nullptr, rgb8 ? c->out_rgb8.p : nullptr);
    HIP_TRY(hipGetLastError());
    if (rgb_linear)
      HIP_TRY(hipMemcpyAsync(rgb_linear, c->out_linear.p, nv * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (rgb8) HIP_TRY(hipMemcpyAsync(rgb8, c->out_rgb8.p, nv, hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RT_OK;
}

int rt_trace_batch_device(rt_ctx* c, const float* o_xyz, const float* d_xyz, size_t n, int32_t* id, float* t, float* nrm,
                          int counted) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->have_scene) return fail(RT_ERR_STATE, "rt_trace_batch before rt_scene_upload");
  if (n == 0) return RT_OK;
  if (!o_xyz || !d_xyz || !id || !t || !nrm) return fail(RT_ERR_INVALID, "null ray or result pointer");
  if (n > ((size_t)1 << 31) * 255) return fail(RT_ERR_INVALID, "batch too large");
  trace_args a;
  a.sc = c->view();
  a.o_xyz = o_xyz;
  a.d_xyz = d_xyz;
  a.n = n;
  a.id = id;
  a.t = t;
  a.nrm = nrm;
  a.counters = nullptr;
  HIP_TRY(c->queue.resize(1));
  a.queue = c->queue.p;
  a.vote = trace_vote();
  a.leaf_vote = leaf_vote();
  if (counted) {
    HIP_TRY(c->counters.resize(sizeof(rt_counters) / sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(c->counters.p, 0, sizeof(rt_counters), c->stream));
    a.counters = c->counters.p;
  }
  hipError_t e;  // (launch_trace zeroes the queue head and records the timing events around the kernel)
  switch (c->search) {
    case SEARCH_TREE_LDS_RAY: e = launch_trace<SEARCH_TREE_LDS_RAY>(c, counted != 0, c->stream, a); break;
    case SEARCH_TREE_LDS_NODE: e = launch_trace<SEARCH_TREE_LDS_NODE>(c, counted != 0, c->stream, a); break;
    case SEARCH_TREE_GLB_RAY: e = launch_trace<SEARCH_TREE_GLB_RAY>(c, counted != 0, c->stream, a); break;
    case SEARCH_TREE_GLB_NODE: e = launch_trace<SEARCH_TREE_GLB_NODE>(c, counted != 0, c->stream, a); break;
    case SEARCH_TREE_GLB_NODE_Q: e = launch_trace<SEARCH_TREE_GLB_NODE_Q>(c, counted != 0, c->stream, a); break;
    case SEARCH_LINEAR_Q: e = launch_trace<SEARCH_LINEAR_Q>(c, counted != 0, c->stream, a); break;
    case SEARCH_LINEAR_LDS: e = launch_trace<SEARCH_LINEAR_LDS>(c, counted != 0, c->stream, a); break;
    default: e = launch_trace<SEARCH_LINEAR>(c, counted != 0, c->stream, a); break;
  }
  if (e != hipSuccess) return fail(RT_ERR_DEVICE, "trace launch: %s", hipGetErrorString(e));
  return RT_OK;
}

// rt_trace_batch_device without a single allocation: queue head and traversal-stack overflow live in the device's
// copy of rt_noalloc_scratch (address looked up once by rt_init).
int rt_trace_batch_noalloc(rt_ctx* c, const float* o_xyz, const float* d_xyz, size_t n, int32_t* id, float* t, float* nrm) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->have_scene) return fail(RT_ERR_STATE, "rt_trace_batch_noalloc before rt_scene_upload");
  if (n == 0) return RT_OK;
  if (!o_xyz || !d_xyz || !id || !t || !nrm) return fail(RT_ERR_INVALID, "null ray or result pointer");
  if (n > (size_t)RT_NOALLOC_MAX_RAYS)
    return fail(RT_ERR_INVALID, "rt_trace_batch_noalloc takes at most %llu rays per call, got %zu",
                (unsigned long long)RT_NOALLOC_MAX_RAYS, n);
  unsigned char* scratch = c->device < kMaxDevices ? g_noalloc_scratch[c->device] : nullptr;
  if (!scratch) return fail(RT_ERR_STATE, "device %d has no fixed scratch (rt_init looks it up for devices below %d)", c->device, kMaxDevices);
  trace_args a;
  a.sc = c->view();
  a.o_xyz = o_xyz;
  a.d_xyz = d_xyz;
  a.n = n;
  a.id = id;
  a.t = t;
  a.nrm = nrm;
  a.counters = nullptr;
  a.queue = reinterpret_cast<unsigned long long*>(scratch);
  a.vote = trace_vote();
  a.leaf_vote = leaf_vote();
  unsigned char* ovf = scratch + kNoAllocHeaderBytes;
  const size_t ovf_bytes = (size_t)RT_NOALLOC_SCRATCH_BYTES - kNoAllocHeaderBytes;
  hipError_t e;
  switch (c->search) {
    case SEARCH_TREE_LDS_RAY: e = launch_trace<SEARCH_TREE_LDS_RAY>(c, false, c->stream, a, ovf, ovf_bytes); break;
    case SEARCH_TREE_LDS_NODE: e = launch_trace<SEARCH_TREE_LDS_NODE>(c, false, c->stream, a, ovf, ovf_bytes); break;
    case SEARCH_TREE_GLB_RAY: e = launch_trace<SEARCH_TREE_GLB_RAY>(c, false, c->stream, a, ovf, ovf_bytes); break;
    case SEARCH_TREE_GLB_NODE: e = launch_trace<SEARCH_TREE_GLB_NODE>(c, false, c->stream, a, ovf, ovf_bytes); break;
    case SEARCH_TREE_GLB_NODE_Q: e = launch_trace<SEARCH_TREE_GLB_NODE_Q>(c, false, c->stream, a, ovf, ovf_bytes); break;
    case SEARCH_LINEAR_Q: e = launch_trace<SEARCH_LINEAR_Q>(c, false, c->stream, a, ovf, ovf_bytes); break;
    case SEARCH_LINEAR_LDS: e = launch_trace<SEARCH_LINEAR_LDS>(c, false, c->stream, a, ovf, ovf_bytes); break;
    default: e = launch_trace<SEARCH_LINEAR>(c, false, c->stream, a, ovf, ovf_bytes); break;
  }
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    return fail(RT_ERR_INVALID, "the hierarchy (depth %d) needs more traversal-stack overflow than the %u-byte fixed scratch "
                "holds for one workgroup per CU; use rt_trace_batch_device", c->depth, (unsigned)RT_NOALLOC_SCRATCH_BYTES);
  }
  if (e != hipSuccess) return fail(RT_ERR_DEVICE, "trace launch: %s", hipGetErrorString(e));
  return RT_OK;
}

// Host buffers: the batch goes through the GPU in chunks, with the copy in of chunk k+1 and the copy out of chunk
// k-1 running beside the kernel of chunk k (two sets of device buffers, one stream per direction).
int rt_trace_batch(rt_ctx* c, const float* o_xyz, const float* d_xyz, size_t n, int32_t* id, float* t, float* nrm) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->have_scene) return fail(RT_ERR_STATE, "rt_trace_batch before rt_scene_upload");
  if (n == 0) return RT_OK;
  if (!o_xyz || !d_xyz || !id || !t || !nrm) return fail(RT_ERR_INVALID, "null ray or result pointer");
  constexpr size_t kChunk = (size_t)1 << 22;  // rays per chunk: 96 MB in, 80 MB out
  const size_t chunk = n < kChunk ? n : kChunk;
  if (!c->copy_in) HIP_TRY(hipStreamCreateWithFlags(&c->copy_in, hipStreamNonBlocking));
  if (!c->copy_out) HIP_TRY(hipStreamCreateWithFlags(&c->copy_out, hipStreamNonBlocking));
  for (int b = 0; b < 2; b++) {
    if (!c->ev_in[b]) HIP_TRY(hipEventCreateWithFlags(&c->ev_in[b], hipEventDisableTiming));
    if (!c->ev_run[b]) HIP_TRY(hipEventCreateWithFlags(&c->ev_run[b], hipEventDisableTiming));
    if (!c->ev_out[b]) HIP_TRY(hipEventCreateWithFlags(&c->ev_out[b], hipEventDisableTiming));
    HIP_TRY(c->tb_o[b].resize(3 * chunk));
    HIP_TRY(c->tb_d[b].resize(3 * chunk));
    HIP_TRY(c->tb_t[b].resize(chunk));
    HIP_TRY(c->tb_n[b].resize(3 * chunk));
    HIP_TRY(c->tb_id[b].resize(chunk));
  }
  const size_t n_chunks = (n + chunk - 1) / chunk;
  auto copy_out = [&](size_t k) -> hipError_t {  // results of chunk k to the caller, after its kernel
    const int b = (int)(k & 1);
    const size_t first = k * chunk, m = n - first < chunk ? n - first : chunk;
    hipError_t e = hipStreamWaitEvent(c->copy_out, c->ev_run[b], 0);
    if (e == hipSuccess) e = hipMemcpyAsync(id + first, c->tb_id[b].p, m * sizeof(int32_t), hipMemcpyDeviceToHost, c->copy_out);
    if (e == hipSuccess) e = hipMemcpyAsync(t + first, c->tb_t[b].p, m * sizeof(float), hipMemcpyDeviceToHost, c->copy_out);
    if (e == hipSuccess) e = hipMemcpyAsync(nrm + 3 * first, c->tb_n[b].p, 3 * m * sizeof(float), hipMemcpyDeviceToHost, c->copy_out);
    if (e == hipSuccess) e = hipEventRecord(c->ev_out[b], c->copy_out);
    return e;
  };
  hipError_t e = hipSuccess;
  for (size_t k = 0; k < n_chunks && e == hipSuccess && rc == RT_OK; k++) {
    const int b = (int)(k & 1);
    const size_t first = k * chunk, m = n - first < chunk ? n - first : chunk;
    // the rays of chunk k-2 were read by its kernel; its results have left the result buffers
    if (k >= 2) e = hipStreamWaitEvent(c->copy_in, c->ev_run[b], 0);
    if (e == hipSuccess) e = hipMemcpyAsync(c->tb_o[b].p, o_xyz + 3 * first, 3 * m * sizeof(float), hipMemcpyHostToDevice, c->copy_in);
    if (e == hipSuccess) e = hipMemcpyAsync(c->tb_d[b].p, d_xyz + 3 * first, 3 * m * sizeof(float), hipMemcpyHostToDevice, c->copy_in);
    if (e == hipSuccess) e = hipEventRecord(c->ev_in[b], c->copy_in);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_in[b], 0);
    if (e == hipSuccess && k >= 2) e = hipStreamWaitEvent(c->stream, c->ev_out[b], 0);
    if (e != hipSuccess) break;
    rc = rt_trace_batch_device(c, c->tb_o[b].p, c->tb_d[b].p, m, c->tb_id[b].p, c->tb_t[b].p, c->tb_n[b].p, 0);
    if (rc != RT_OK) break;
    e = hipEventRecord(c->ev_run[b], c->stream);
    // copies from pageable memory hold the calling thread: the previous chunk's results go out while this chunk's kernel runs
    if (e == hipSuccess && k >= 1) e = copy_out(k - 1);
  }
  if (e == hipSuccess && rc == RT_OK) e = copy_out(n_chunks - 1);
  const hipError_t s0 = hipStreamSynchronize(c->copy_in), s1 = hipStreamSynchronize(c->stream), s2 = hipStreamSynchronize(c->copy_out);
  if (rc != RT_OK) return rc;
  if (e == hipSuccess) e = s0 != hipSuccess ? s0 : (s1 != hipSuccess ? s1 : s2);
  if (e != hipSuccess) return fail(RT_ERR_DEVICE, "ray batch transfer: %s", hipGetErrorString(e));
  return RT_OK;
}

int rt_synchronize(rt_ctx* c) {
  int rc = check_ctx(c);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RT_OK;
}

int rt_get_counters(rt_ctx* c, rt_counters* out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!out) return fail(RT_ERR_INVALID, "null out pointer");
  if (!c->counters.p) return fail(RT_ERR_STATE, "no counted call was made");
  HIP_TRY(hipMemcpyAsync(out, c->counters.p, sizeof(rt_counters), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RT_OK;
}

int rt_last_kernel_ms(rt_ctx* c, float* ms, const char** kernel_name) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!ms) return fail(RT_ERR_INVALID, "null out pointer");
  if (!c->have_kernel_time) return fail(RT_ERR_STATE, "no kernel was launched");
  HIP_TRY(hipEventSynchronize(c->ev_k1));
  HIP_TRY(hipEventElapsedTime(ms, c->ev_k0, c->ev_k1));
  if (kernel_name) *kernel_name = c->last_kernel;
  return RT_OK;
}

int rt_last_build_ms(rt_ctx* c, float* ms) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!ms) return fail(RT_ERR_INVALID, "null out pointer");
  if (!c->have_build_time) return fail(RT_ERR_STATE, "no LBVH was built");
  HIP_TRY(hipEventSynchronize(c->ev_b1));
  HIP_TRY(hipEventElapsedTime(ms, c->ev_b0, c->ev_b1));
  return RT_OK;
}

int rt_lbvh_depth(rt_ctx* c) { return c ? c->depth : 0; }

int rt_get_scene_info(rt_ctx* c, rt_scene_info* out) {
  if (!c || !out) return fail(RT_ERR_INVALID, "null argument");
  if (!c->have_scene) return fail(RT_ERR_STATE, "no scene uploaded");
  std::memset(out, 0, sizeof(*out));
  out->n_spheres = c->n_spheres;
  out->n_materials = c->n_materials;
  out->accel = c->accel;
  out->builder = c->builder;
  out->n_big = c->n_big;
  out->n_leaf = c->n_leaf;
  out->n_nodes = c->n_nodes;
  out->depth = c->depth;
  out->tree_in_lds = c->search == SEARCH_TREE_LDS_RAY || c->search == SEARCH_TREE_LDS_NODE;
  out->pad_per_ray = c->search == SEARCH_TREE_LDS_RAY || c->search == SEARCH_TREE_GLB_RAY;
  out->node_bytes = c->n_nodes > 0 ? (out->tree_in_lds ? (int)sizeof(bvh4_node) : (int)sizeof(bvh4q_node)) : 0;
  return RT_OK;
}

}  // extern "C"

extern "C" {

int rt_get_stream(rt_ctx* c, void** hip_stream) {
  if (!c || !hip_stream) return fail(RT_ERR_INVALID, "null argument");
  *hip_stream = reinterpret_cast<void*>(c->stream);
  return RT_OK;
}

}  // extern "C"
