// rt_comm.hip — the sum of the ranks' accumulators behind the C ABI: RCCL over xGMI, one communicator per context,
// the library loaded on first use.

#include <dlfcn.h>

#include <rccl/rccl.h>  // types and prototypes only: the library is loaded on first use (rccl_api)

#include "rt_ctx.hpp"

namespace {

// RCCL's entry points, looked up at first use: a caller that never sums across ranks needs no RCCL at all, and a
// process that already holds a copy (PyTorch ships its own) must not get a second one.
struct rccl_api {
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclReduce) reduce = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  bool ok = false;
  std::string why;  // when !ok: what the loader said about each name tried, or which symbols are missing
};

const rccl_api& rccl() {
  static const rccl_api api = [] {
    rccl_api a;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
      const char* msg = dlerror();  // (one call: it clears the pending message)
      a.why += std::string(a.why.empty() ? "" : "; ") + (msg ? msg : name);
    }
    if (!h) return a;
    a.get_unique_id = reinterpret_cast<decltype(a.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
    a.comm_init_rank = reinterpret_cast<decltype(a.comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
    a.reduce = reinterpret_cast<decltype(a.reduce)>(dlsym(h, "ncclReduce"));
    a.comm_destroy = reinterpret_cast<decltype(a.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
    a.error_string = reinterpret_cast<decltype(a.error_string)>(dlsym(h, "ncclGetErrorString"));
    a.ok = a.get_unique_id && a.comm_init_rank && a.reduce && a.comm_destroy && a.error_string;
    a.why = a.ok ? "" : "the library lacks ncclGetUniqueId / ncclCommInitRank / ncclReduce / ncclCommDestroy / ncclGetErrorString";
    return a;
  }();
  return api;
}

int rccl_fail(const char* what, ncclResult_t r) { return fail(RT_ERR_DEVICE, "%s: %s", what, rccl().error_string(r)); }

}  // namespace

extern "C" {

int rt_comm_unique_id(void* id_out) {
  if (!id_out) return fail(RT_ERR_INVALID, "null id");
  if (!rccl().ok) return fail(RT_ERR_DEVICE, "RCCL (librccl.so.1) cannot be loaded: %s", rccl().why.c_str());
  ncclUniqueId id;
  const ncclResult_t r = rccl().get_unique_id(&id);
  if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
  static_assert(sizeof(id) == RT_COMM_ID_BYTES, "the id is RT_COMM_ID_BYTES long");
  std::memcpy(id_out, &id, sizeof(id));
  return RT_OK;
}

int rt_comm_init(rt_ctx* c, const void* id_bytes, int rank, int world) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!id_bytes || world < 1 || rank < 0 || rank >= world) return fail(RT_ERR_INVALID, "bad group arguments (rank %d of %d)", rank, world);
  if (c->comm) return fail(RT_ERR_STATE, "the context is in a group already");
  if (!rccl().ok) return fail(RT_ERR_DEVICE, "RCCL (librccl.so.1) cannot be loaded: %s", rccl().why.c_str());
  ncclUniqueId id;
  std::memcpy(&id, id_bytes, sizeof(id));
  ncclComm_t comm = nullptr;
  const ncclResult_t r = rccl().comm_init_rank(&comm, world, id, rank);  // on the current device: check_ctx has set it
  if (r != ncclSuccess) return rccl_fail("ncclCommInitRank", r);
  c->comm = comm;
  c->comm_rank = rank;
  c->comm_world = world;
  return RT_OK;
}

int rt_comm_reduce(rt_ctx* c, int root) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->comm) return fail(RT_ERR_STATE, "rt_comm_reduce before rt_comm_init");
  if (root < 0 || root >= c->comm_world) return fail(RT_ERR_INVALID, "root %d out of range [0,%d)", root, c->comm_world);
  if (!c->accum() || c->acc_w < 1) return fail(RT_ERR_STATE, "nothing rendered yet");
  const size_t nv = 3 * (size_t)c->acc_w * (size_t)c->acc_h;
  // in place on the root; the others only send
  const ncclResult_t r = rccl().reduce(c->accum(), c->accum(), nv, ncclFloat32, ncclSum, root, static_cast<ncclComm_t>(c->comm), c->stream);
  if (r != ncclSuccess) return rccl_fail("ncclReduce", r);
  return RT_OK;
}

int rt_comm_destroy(rt_ctx* c) {
  if (!c) return fail(RT_ERR_INVALID, "null context");
  if (c->comm) {
    (void)hipStreamSynchronize(c->stream);
    (void)rccl().comm_destroy(static_cast<ncclComm_t>(c->comm));
    c->comm = nullptr;
  }
  return RT_OK;
}

}  // extern "C"
