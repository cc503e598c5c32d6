// b3w_comm.cpp — C-ABI part 6: the multi-GPU exchange: b3w_comm over RCCL (loaded at run time), the host shared-memory transport
// (b3w_hostcomm.cpp) or the caller's collective.
#include "b3w_internal.h"


#include <dlfcn.h>

#include "b3w_hostcomm.h"

struct b3w_comm {
  b3w_ctx *ctx = nullptr;
  int32_t rank = 0, nranks = 1;
  enum Kind { RCCL, HOST, EXTERNAL } kind = RCCL;
  void *comm = nullptr;          // RCCL: ncclComm_t
  B3wHostComm *host = nullptr;   // HOST: the shared-memory segment, and two pinned staging buffers that grow with the messages
  uint8_t *h_send = nullptr, *h_recv = nullptr;
  uint64_t h_cap = 0;
  b3w_allgather_fn fn = nullptr; // EXTERNAL: the caller's collective
  void *user = nullptr;
};

namespace {
struct RcclId { char b[128]; };  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128, passed by value)
struct Rccl {                    // the five entry points used, with rccl.h's signatures
  void *so = nullptr;
  int (*GetUniqueId)(void *id) = nullptr;
  int (*CommInitRank)(void **comm, int nranks, RcclId id, int rank) = nullptr;
  int (*AllGather)(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t stream) = nullptr;
  int (*CommDestroy)(void *comm) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string err;
} rccl;

void load_rccl_once() {
  void *so = nullptr;
  for (const char *name : {"librccl.so", "librccl.so.1"}) if (!so) so = dlopen(name, RTLD_NOW | RTLD_NOLOAD);   // one already in the process
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if (!so) so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
  if (!so) { rccl.err = std::string("cannot load librccl: ") + dlerror(); return; }
  rccl.GetUniqueId = (decltype(rccl.GetUniqueId))dlsym(so, "ncclGetUniqueId");
  rccl.CommInitRank = (decltype(rccl.CommInitRank))dlsym(so, "ncclCommInitRank");
  rccl.AllGather = (decltype(rccl.AllGather))dlsym(so, "ncclAllGather");
  rccl.CommDestroy = (decltype(rccl.CommDestroy))dlsym(so, "ncclCommDestroy");
  rccl.GetErrorString = (decltype(rccl.GetErrorString))dlsym(so, "ncclGetErrorString");
  if (!rccl.GetUniqueId || !rccl.CommInitRank || !rccl.AllGather || !rccl.CommDestroy || !rccl.GetErrorString) { rccl.err = "librccl lacks an ncclAllGather entry point"; return; }
  rccl.so = so;
}
std::once_flag rccl_once;
bool load_rccl() {                 // thread-safe: distinct contexts may create communicators from different threads
  std::call_once(rccl_once, load_rccl_once);
  return rccl.so != nullptr;
}
}  // namespace

extern "C" {

int32_t b3w_comm_unique_id(uint8_t id[B3W_COMM_ID_BYTES]) {
  if (!id) return B3W_E_BAD_ARGUMENT;
  if (!load_rccl()) return B3W_E_RCCL;
  return rccl.GetUniqueId(id) == 0 ? B3W_OK : B3W_E_RCCL;
}

int32_t b3w_comm_create(b3w_ctx *ctx, const uint8_t id[B3W_COMM_ID_BYTES], int32_t rank, int32_t nranks, b3w_comm **out) {
  if (!ctx || !id || !out || nranks < 1 || rank < 0 || rank >= nranks) return B3W_E_BAD_ARGUMENT;
  *out = nullptr;
  if (!load_rccl()) { ctx->last_error = rccl.err; return B3W_E_RCCL; }
  ON_DEVICE(ctx);
  RcclId uid;
  memcpy(uid.b, id, 128);
  void *comm = nullptr;
  const int rc = rccl.CommInitRank(&comm, nranks, uid, rank);
  if (rc != 0) { ctx->last_error = std::string("ncclCommInitRank: ") + rccl.GetErrorString(rc); return B3W_E_RCCL; }
  b3w_comm *c = new b3w_comm;
  c->ctx = ctx; c->comm = comm; c->rank = rank; c->nranks = nranks;
  *out = c;
  return B3W_OK;
}

int32_t b3w_comm_create_host(b3w_ctx *ctx, const char *name, int32_t rank, int32_t nranks, b3w_comm **out) {
  if (!ctx || !name || !out || nranks < 1 || rank < 0 || rank >= nranks) return B3W_E_BAD_ARGUMENT;
  *out = nullptr;
  const char *t = getenv("B3W_HOSTCOMM_TIMEOUT_S");
  char err[256] = "";
  B3wHostComm *hc = nullptr;
  // 4 MiB per rank at a time: config 4's exchanges (256 KiB of h_out per rank at two ranks) go through in one piece
  if (b3w_hostcomm_open(name, rank, nranks, 4u << 20, t ? atof(t) : 120.0, &hc, err, sizeof err) != 0) {
    ctx->last_error = err;
    return B3W_E_RCCL;
  }
  b3w_comm *c = new b3w_comm;
  c->ctx = ctx; c->rank = rank; c->nranks = nranks; c->kind = b3w_comm::HOST; c->host = hc;
  *out = c;
  return B3W_OK;
}

int32_t b3w_comm_create_external(b3w_ctx *ctx, int32_t rank, int32_t nranks, b3w_allgather_fn allgather, void *user, b3w_comm **out) {
  if (!ctx || !allgather || !out || nranks < 1 || rank < 0 || rank >= nranks) return B3W_E_BAD_ARGUMENT;
  b3w_comm *c = new b3w_comm;
  c->ctx = ctx; c->rank = rank; c->nranks = nranks; c->kind = b3w_comm::EXTERNAL; c->fn = allgather; c->user = user;
  *out = c;
  return B3W_OK;
}

int32_t b3w_comm_rank(const b3w_comm *c) { return c ? c->rank : -1; }
int32_t b3w_comm_size(const b3w_comm *c) { return c ? c->nranks : 0; }

void b3w_comm_destroy(b3w_comm *c) {
  if (!c) return;
  B3wCaptureRelaxed relaxed;                                 // (b3w_capture.h)
  DeviceGuard guard(c->ctx->device);
  if (c->comm && rccl.CommDestroy) (void)rccl.CommDestroy(c->comm);
  if (c->h_send) (void)hipHostFree(c->h_send);
  if (c->h_recv) (void)hipHostFree(c->h_recv);
  b3w_hostcomm_close(c->host);
  delete c;
}

namespace {
// HOST transport: device -> pinned host -> shared-memory all-gather -> device, ordered on `stream` by waiting for it (twice)
int32_t host_allgather(b3w_comm *c, const void *d_send, void *d_recv, uint64_t bytes, hipStream_t st) {
  b3w_ctx *ctx = c->ctx;
  if (bytes > c->h_cap) {
    if (c->h_send) (void)hipHostFree(c->h_send);
    if (c->h_recv) (void)hipHostFree(c->h_recv);
    c->h_send = c->h_recv = nullptr; c->h_cap = 0;
    HIP_TRY(ctx, hipHostMalloc((void **)&c->h_send, bytes, hipHostMallocDefault));
    HIP_TRY(ctx, hipHostMalloc((void **)&c->h_recv, bytes * (uint64_t)c->nranks, hipHostMallocDefault));
    c->h_cap = bytes;
  }
  HIP_TRY(ctx, hipMemcpyAsync(c->h_send, d_send, bytes, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  char err[256] = "";
  if (b3w_hostcomm_allgather(c->host, c->h_send, c->h_recv, bytes, err, sizeof err) != 0) { ctx->last_error = err; return B3W_E_RCCL; }
  HIP_TRY(ctx, hipMemcpyAsync(d_recv, c->h_recv, bytes * (uint64_t)c->nranks, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));                     // the staging buffer is free for the next call, whatever its stream
  return B3W_OK;
}
}  // namespace

int32_t b3w_comm_allgather(b3w_comm *c, const void *d_send, void *d_recv, uint64_t bytes_per_rank, void *stream) {
  if (!c || !d_send || !d_recv || !bytes_per_rank) return B3W_E_BAD_ARGUMENT;
  ON_DEVICE(c->ctx);
  if (c->kind == b3w_comm::HOST) return host_allgather(c, d_send, d_recv, bytes_per_rank, (hipStream_t)stream);
  if (c->kind == b3w_comm::EXTERNAL) {
    const int32_t rc = c->fn(c->user, d_send, d_recv, bytes_per_rank, stream);
    if (rc != 0) {
      c->ctx->last_error = "the caller's all-gather (b3w_comm_create_external) returned " + std::to_string(rc);
      return B3W_E_RCCL;
    }
    return B3W_OK;
  }
  const int rc = rccl.AllGather(d_send, d_recv, (size_t)bytes_per_rank, /* ncclInt8 */ 0, c->comm, (hipStream_t)stream);
  if (rc != 0) { c->ctx->last_error = std::string("ncclAllGather: ") + rccl.GetErrorString(rc); return B3W_E_RCCL; }
  return B3W_OK;
}

int32_t b3w_batch_allgather_public(b3w_batch *b, b3w_comm *c, uint32_t *host_all) {
  if (!b || !c || !host_all || !b->n) return B3W_E_BAD_ARGUMENT;
  b3w_ctx *ctx = b->ctx;
  ON_DEVICE(ctx);
  const uint64_t per = (uint64_t)b->n * ctx->desc.npub * 4;
  void *d_all = nullptr;
  HIP_TRY(ctx, hipMalloc(&d_all, per * c->nranks));
  int32_t rc = b3w_comm_allgather(c, b->d_pub, d_all, per, nullptr);
  hipError_t e = rc == B3W_OK ? hipMemcpy(host_all, d_all, per * c->nranks, hipMemcpyDeviceToHost) : hipSuccess;
  (void)hipFree(d_all);
  if (rc) return rc;
  return e == hipSuccess ? B3W_OK : hip_fail(ctx, e, "hipMemcpy(gathered public outputs)");
}

}  // extern "C"
