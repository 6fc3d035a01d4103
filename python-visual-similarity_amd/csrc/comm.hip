// The one exchange step of the multi-GPU path behind the C-ABI: RCCL over xGMI, one rank per GPU.
//
// The reference has no multi-GPU path; the step replaces nothing in it.  It exists because the path shards by image
// (pyvisim/encoders/vlad.py:87-113: images are independent; index order = dict insertion order, pyvisim/eval.py:28) and the
// pairwise step needs every rank's encoding block: one all-gather of the blocks, and -- in the symmetric block-pair scheme
// (pvsim/distributed.py) -- one all-to-all of k-candidate lists.
//
// librccl is NOT a link dependency of libpvsim_hip.so: it is resolved at pvs_comm_init, preferring an image of the library
// that is already mapped into the process (e.g. the one a PyTorch wheel loaded), then the copy that sits next to the mapped HIP
// runtime, then the system one -- so that the collective library and the runtime come from the same installation.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <string>

#include "common.hpp"

struct pvs_comm {
  pvs_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int nranks = 1, rank = 0;
};

namespace pvs {

struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string path;
};

static RcclApi g_rccl;   // process-wide by nature: one image of the collective library per process

static std::string dir_of_mapped(const char* needle) {
  FILE* f = fopen("/proc/self/maps", "r");
  if (!f) return "";
  char line[4096];
  std::string dir;
  while (fgets(line, sizeof(line), f)) {
    char* p = strchr(line, '/');
    if (!p) continue;
    char* nl = strchr(p, '\n');
    if (nl) *nl = 0;
    const char* base = strrchr(p, '/');
    if (base && strstr(base, needle)) {
      dir.assign(p, base - p);
      break;
    }
  }
  fclose(f);
  return dir;
}

static int load_rccl() {
  if (g_rccl.handle) return PVS_OK;
  void* h = nullptr;
  std::string tried;
  for (const char* name : {"librccl.so.1", "librccl.so"}) {     // an image that is already mapped
    h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    if (h) { g_rccl.path = std::string(name) + " (already mapped)"; break; }
  }
  if (!h) {
    const std::string mapped = dir_of_mapped("librccl");
    if (!mapped.empty())
      for (const char* name : {"/librccl.so", "/librccl.so.1"}) {
        h = dlopen((mapped + name).c_str(), RTLD_NOW | RTLD_NOLOAD);
        if (h) { g_rccl.path = mapped + name; break; }
      }
  }
  if (!h) {                                                       // next to the HIP runtime this process uses
    const std::string d = dir_of_mapped("libamdhip64");
    if (!d.empty())
      for (const char* name : {"/librccl.so.1", "/librccl.so"}) {
        h = dlopen((d + name).c_str(), RTLD_NOW | RTLD_GLOBAL);
        tried += d + name + " ";
        if (h) { g_rccl.path = d + name; break; }
      }
  }
  if (!h)
    for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      tried += std::string(name) + " ";
      if (h) { g_rccl.path = name; break; }
    }
  if (!h) PVS_FAIL(PVS_ERR_UNSUPPORTED, "RCCL not found (tried %s): %s", tried.c_str(), dlerror());
#define PVS_RCCL_SYM(field, sym)                                                            \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, #sym));                  \
  if (!g_rccl.field) PVS_FAIL(PVS_ERR_UNSUPPORTED, "RCCL (%s) lacks %s", g_rccl.path.c_str(), #sym)
  PVS_RCCL_SYM(GetUniqueId, ncclGetUniqueId);
  PVS_RCCL_SYM(CommInitRank, ncclCommInitRank);
  PVS_RCCL_SYM(CommDestroy, ncclCommDestroy);
  PVS_RCCL_SYM(AllGather, ncclAllGather);
  PVS_RCCL_SYM(AllReduce, ncclAllReduce);
  PVS_RCCL_SYM(Send, ncclSend);
  PVS_RCCL_SYM(Recv, ncclRecv);
  PVS_RCCL_SYM(GroupStart, ncclGroupStart);
  PVS_RCCL_SYM(GroupEnd, ncclGroupEnd);
  PVS_RCCL_SYM(GetErrorString, ncclGetErrorString);
#undef PVS_RCCL_SYM
  g_rccl.handle = h;
  return PVS_OK;
}

#define PVS_NCCL(expr)                                                                                   \
  do {                                                                                                   \
    ncclResult_t r__ = (expr);                                                                           \
    if (r__ != ncclSuccess) PVS_FAIL(PVS_ERR_NO_DEVICE, "%s failed: %s", #expr, g_rccl.GetErrorString(r__)); \
  } while (0)

// inside ncclGroupStart / ncclGroupEnd: remember the first failure and keep going, so that the group is always closed
// (a Send / Recv that fails must not leave the group open: the next collective would be queued into it or hang)
#define PVS_NCCL_IN_GROUP(first, expr)                                                            \
  do {                                                                                            \
    ncclResult_t r__ = (expr);                                                                    \
    if (r__ != ncclSuccess && (first) == ncclSuccess) {                                           \
      (first) = r__;                                                                              \
      ::pvs::set_error("%s failed: %s", #expr, g_rccl.GetErrorString(r__));                       \
    }                                                                                             \
  } while (0)

}  // namespace pvs

using namespace pvs;

#define PVS_NEEDC(p, what) \
  if (!(p)) PVS_FAIL(PVS_ERR_INVALID, "%s: null %s", __func__, what)

PVS_EXPORT int pvs_comm_unique_id(void* out_id) {
  PVS_NEEDC(out_id, "id");
  PVS_TRY(load_rccl());
  ncclUniqueId id;
  PVS_NCCL(g_rccl.GetUniqueId(&id));
  static_assert(sizeof(id) == PVS_UNIQUE_ID_BYTES, "unique id size");
  memcpy(out_id, &id, sizeof(id));
  return PVS_OK;
}

PVS_EXPORT int pvs_comm_init(pvs_ctx* ctx, int nranks, int rank, const void* unique_id, pvs_comm** out) {
  PVS_NEEDC(ctx, "ctx");
  PVS_NEEDC(unique_id, "unique id");
  PVS_NEEDC(out, "out");
  if (nranks < 1 || rank < 0 || rank >= nranks) PVS_FAIL(PVS_ERR_INVALID, "rank %d of %d", rank, nranks);
  PVS_TRY(load_rccl());
  PVS_HIP(hipSetDevice(ctx->device));
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  pvs_comm* c = new pvs_comm();
  c->ctx = ctx;
  c->nranks = nranks;
  c->rank = rank;
  const ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
  if (r != ncclSuccess) {
    delete c;
    PVS_FAIL(PVS_ERR_NO_DEVICE, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
  }
  *out = c;
  return PVS_OK;
}

PVS_EXPORT int pvs_comm_destroy(pvs_comm* c) {
  if (!c) return PVS_OK;
  if (c->comm && g_rccl.CommDestroy) {
    hipSetDevice(c->ctx->device);
    hipStreamSynchronize(c->ctx->stream);
    g_rccl.CommDestroy(c->comm);
  }
  delete c;
  return PVS_OK;
}

PVS_EXPORT const char* pvs_comm_library(void) { return g_rccl.path.c_str(); }

PVS_EXPORT int pvs_allgather_dev(pvs_comm* c, const void* d_send, void* d_recv, size_t bytes_per_rank) {
  PVS_NEEDC(c, "comm");
  if (bytes_per_rank == 0) return PVS_OK;
  PVS_NEEDC(d_send, "send");
  PVS_NEEDC(d_recv, "recv");
  PVS_HIP(hipSetDevice(c->ctx->device));
  // whole 16-B words when the size allows: fewer, wider elements for the ring kernels; bytes otherwise
  if (bytes_per_rank % 4 == 0) PVS_NCCL(g_rccl.AllGather(d_send, d_recv, bytes_per_rank / 4, ncclUint32, c->comm, c->ctx->stream));
  else PVS_NCCL(g_rccl.AllGather(d_send, d_recv, bytes_per_rank, ncclUint8, c->comm, c->ctx->stream));
  return PVS_OK;
}

PVS_EXPORT int pvs_alltoall_dev(pvs_comm* c, const void* d_send, void* d_recv, size_t bytes_per_rank) {
  PVS_NEEDC(c, "comm");
  if (bytes_per_rank == 0) return PVS_OK;
  PVS_NEEDC(d_send, "send");
  PVS_NEEDC(d_recv, "recv");
  PVS_HIP(hipSetDevice(c->ctx->device));
  const char* s = static_cast<const char*>(d_send);
  char* r = static_cast<char*>(d_recv);
  PVS_NCCL(g_rccl.GroupStart());
  ncclResult_t first = ncclSuccess;
  for (int p = 0; p < c->nranks && first == ncclSuccess; ++p) {
    PVS_NCCL_IN_GROUP(first, g_rccl.Send(s + (size_t)p * bytes_per_rank, bytes_per_rank, ncclUint8, p, c->comm, c->ctx->stream));
    PVS_NCCL_IN_GROUP(first, g_rccl.Recv(r + (size_t)p * bytes_per_rank, bytes_per_rank, ncclUint8, p, c->comm, c->ctx->stream));
  }
  PVS_NCCL_IN_GROUP(first, g_rccl.GroupEnd());
  return first == ncclSuccess ? PVS_OK : PVS_ERR_NO_DEVICE;
}

PVS_EXPORT int pvs_sendrecv_dev(pvs_comm* c, int n_ops, const int* peers, const void* const* d_send, const size_t* send_bytes,
                                void* const* d_recv, const size_t* recv_bytes) {
  PVS_NEEDC(c, "comm");
  if (n_ops <= 0) return PVS_OK;
  PVS_NEEDC(peers, "peers");
  PVS_HIP(hipSetDevice(c->ctx->device));
  for (int i = 0; i < n_ops; ++i)
    if (peers[i] < 0 || peers[i] >= c->nranks) PVS_FAIL(PVS_ERR_INVALID, "peer %d out of range", peers[i]);
  PVS_NCCL(g_rccl.GroupStart());
  ncclResult_t first = ncclSuccess;
  for (int i = 0; i < n_ops && first == ncclSuccess; ++i) {
    if (d_send && send_bytes && send_bytes[i] > 0)
      PVS_NCCL_IN_GROUP(first, g_rccl.Send(d_send[i], send_bytes[i], ncclUint8, peers[i], c->comm, c->ctx->stream));
    if (d_recv && recv_bytes && recv_bytes[i] > 0)
      PVS_NCCL_IN_GROUP(first, g_rccl.Recv(d_recv[i], recv_bytes[i], ncclUint8, peers[i], c->comm, c->ctx->stream));
  }
  PVS_NCCL_IN_GROUP(first, g_rccl.GroupEnd());
  return first == ncclSuccess ? PVS_OK : PVS_ERR_NO_DEVICE;
}

PVS_EXPORT int pvs_allreduce_max_f64(pvs_comm* c, double* h_inout, int count) {
  PVS_NEEDC(c, "comm");
  PVS_NEEDC(h_inout, "values");
  if (count < 1 || count > 64) PVS_FAIL(PVS_ERR_INVALID, "1..64 values");
  pvs_ctx* ctx = c->ctx;
  PVS_HIP(hipSetDevice(ctx->device));
  if (ctx->d_queue == nullptr) PVS_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_queue), 1024));
  double* d = reinterpret_cast<double*>(reinterpret_cast<char*>(ctx->d_queue) + 256);   // 64 doubles behind the queue word
  PVS_HIP(hipMemcpyAsync(d, h_inout, (size_t)count * 8, hipMemcpyHostToDevice, ctx->stream));
  PVS_NCCL(g_rccl.AllReduce(d, d, (size_t)count, ncclFloat64, ncclMax, c->comm, ctx->stream));
  PVS_HIP(hipMemcpyAsync(h_inout, d, (size_t)count * 8, hipMemcpyDeviceToHost, ctx->stream));
  PVS_HIP(hipStreamSynchronize(ctx->stream));
  return PVS_OK;
}

// a barrier over the ranks that also drains this context's stream: all-reduce of one value, then a stream synchronise
PVS_EXPORT int pvs_comm_barrier(pvs_comm* c) {
  double one = 1.0;
  return pvs_allreduce_max_f64(c, &one, 1);
}
