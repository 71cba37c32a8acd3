// t1k_amd/csrc/t1k_comm.hip -- the exchange steps of a genotyper job that is sharded over several GPUs.
//
// One rank per GPU; a rank owns a contiguous slice of the fragments in file order.  Read-end assignment and mate pairing need no
// collective.  The exchanges are
//   * per-base coverage          all-reduce (sum, int32) of the difference + hole arrays           -- exact, integers commute
//   * fragment rows              all-to-all by pattern owner (hash word 1 mod nRanks): every read group is coalesced by exactly one
//                                rank over ALL its fragments in global fragment order (t1k_coalesce.hip), so group contents are
//                                bit-identical to a single-GPU run; the owners' group tables are all-gathered and merged by first
//                                fragment on the host (host/genotype.cpp), which restores the reference's group numbering
//   * EM (Genotyper.hpp:372-421) each rank runs the row pass of an EMupdate on its slice of the read groups; every element of the
//                                contribution array has exactly one writer, so its all-reduce (sum, f64) is exact in any order; the
//                                column pass then adds each class's contributions in group order on every rank -- the same doubles
//                                as one GPU computes.
// Two transports behind one interface: RCCL (ncclAllReduce / ncclSend+ncclRecv groups / ncclBroadcast over xGMI; librccl.so.1 is
// bound lazily with dlopen so that single-GPU runs never load its 570 MB of code objects), and an in-process one for ranks that are
// threads of one process and share a device (T1K_GPUS=0,0: the multi-rank logic runs -- and is tested -- on a single GPU).
#include <dlfcn.h>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <rccl/rccl.h>
#include "t1k_dev.h"
#include "t1k_launch.h"

namespace {

struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

RcclApi *rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // a process that already holds an RCCL (PyTorch's) gets that one: same soname
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (!api.lib) { api.err = std::string("cannot load librccl.so.1: ") + dlerror(); return; }
    auto sym = [&](const char *n) { void *p = dlsym(api.lib, n); if (!p && api.err.empty()) api.err = std::string("librccl: missing symbol ") + n; return p; };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.CommAbort = (decltype(api.CommAbort))sym("ncclCommAbort");
    api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
    api.Broadcast = (decltype(api.Broadcast))sym("ncclBroadcast");
    api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  });
  return &api;
}

// meeting point of the ranks of one process (in-process transport, and the unique-id hand-over of the RCCL one)
struct Hub {
  int n = 0;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t gen = 0;
  std::vector<const void *> ptr;
  std::vector<const uint64_t *> off;
  ncclUniqueId id;
  bool failed = false;
  bool aborted = false;  // a rank gave up (t1k_comm_abort): nobody waits for it any more, every collective returns an error
  // false: the job was aborted (before or while waiting)
  bool barrier() {
    std::unique_lock<std::mutex> lk(m);
    if (aborted) return false;
    const uint64_t g = gen;
    if (++arrived == n) { arrived = 0; ++gen; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != g || aborted; });
    return !aborted;
  }
  void abort() {
    { std::lock_guard<std::mutex> g(m); aborted = true; }
    cv.notify_all();
  }
};

__global__ void k_sum_ranks_i32(int32_t *dst, const int32_t *const *src, int n, uint64_t count) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  int32_t s = 0;
  for (int r = 0; r < n; ++r) s += src[r][i];
  dst[i] = s;
}
__global__ void k_sum_ranks_f64(double *dst, const double *const *src, int n, uint64_t count) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double s = 0;
  for (int r = 0; r < n; ++r) s += src[r][i];  // rank order
  dst[i] = s;
}

}  // namespace

struct t1k_comm {
  int nRanks = 1, rank = 0;
  t1k_ctx *ctx = nullptr;
  bool useRccl = false;
  ncclComm_t nccl = nullptr;
  Hub *hub = nullptr;       // shared by the ranks of one process (owned by the t1k_comm_group)
  T1kDevBuf tmp, ptrs;
  bool aborted = false;
  std::string err;
};

struct t1k_comm_group {
  Hub hub;
};

// (a collective that fails on one rank leaves the others at its next meeting point: the whole job is abandoned with it)
static int commFail(t1k_comm *c, int code, const std::string &m) {
  if (c) { c->err = m; if (c->hub && c->nRanks > 1) c->hub->abort(); }
  return code;
}
#define CM_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return commFail(c, T1K_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
#define CM_NCCL(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return commFail(c, T1K_ERR_DEVICE, std::string(#call) + ": " + rccl()->GetErrorString(r_)); } while (0)

extern "C" {

int t1k_comm_unique_id(void *id128) {
  if (!id128) return T1K_ERR_ARG;
  RcclApi *api = rccl();
  if (!api->err.empty()) return T1K_ERR_DEVICE;
  ncclUniqueId id;
  if (api->GetUniqueId(&id) != ncclSuccess) return T1K_ERR_DEVICE;
  memcpy(id128, &id, sizeof(id));
  return T1K_OK;
}

t1k_comm_group *t1k_comm_group_create(int nRanks) {
  if (nRanks < 1) return nullptr;
  t1k_comm_group *g = new t1k_comm_group();
  g->hub.n = nRanks;
  g->hub.ptr.assign(nRanks, nullptr);
  g->hub.off.assign(nRanks, nullptr);
  return g;
}
void t1k_comm_group_destroy(t1k_comm_group *g) { delete g; }

// id128 != NULL: ranks are processes (or threads) that were handed one ncclUniqueId: RCCL.  group != NULL: ranks are threads of this
// process; RCCL if their devices differ pairwise (transport < 0: decide; 0: in-process; 1: RCCL), the in-process transport otherwise.
int t1k_comm_init(t1k_ctx *ctx, int nRanks, int rank, const void *id128, t1k_comm_group *group, int transport, t1k_comm **out) {
  if (!ctx || !out || nRanks < 1 || rank < 0 || rank >= nRanks || (!id128 && !group && nRanks > 1)) return T1K_ERR_ARG;
  *out = nullptr;
  t1k_comm *c = new t1k_comm();
  c->nRanks = nRanks; c->rank = rank; c->ctx = ctx;
  c->hub = group ? &group->hub : nullptr;
  *out = c;
  if (nRanks == 1 && !id128) return T1K_OK;
  CM_HIP(hipSetDevice(ctx->device));
  bool wantRccl = id128 != nullptr || transport == 1;
  ncclUniqueId id;
  if (id128) memcpy(&id, id128, sizeof(id));
  if (group && !id128) {
    Hub &h = group->hub;
    // devices of all ranks: distinct -> RCCL
    static_assert(sizeof(void *) >= sizeof(int), "");
    h.ptr[rank] = (const void *)(intptr_t)(ctx->device + 1);
    (void)h.barrier();
    bool distinct = true;
    for (int a = 0; a < nRanks; ++a) for (int b = a + 1; b < nRanks; ++b) if (h.ptr[a] == h.ptr[b]) distinct = false;
    (void)h.barrier();
    if (transport < 0) wantRccl = distinct;
    if (wantRccl && !distinct) return commFail(c, T1K_ERR_ARG, "t1k_comm_init: RCCL needs one device per rank");
    if (wantRccl) {
      if (rank == 0) { if (!rccl()->err.empty() || rccl()->GetUniqueId(&h.id) != ncclSuccess) h.failed = true; }
      (void)h.barrier();
      if (h.failed) return commFail(c, T1K_ERR_DEVICE, "t1k_comm_init: " + (rccl()->err.empty() ? std::string("ncclGetUniqueId failed") : rccl()->err));
      id = h.id;
    }
  }
  if (!wantRccl && group) {  // in-process transport across devices: the ranks read each other's buffers directly
    int nDev = 0;
    (void)hipGetDeviceCount(&nDev);
    for (int d = 0; d < nDev; ++d)
      if (d != ctx->device) { hipError_t e = hipDeviceEnablePeerAccess(d, 0); (void)e; (void)hipGetLastError(); }
  }
  if (wantRccl) {
    RcclApi *api = rccl();
    if (!api->err.empty()) return commFail(c, T1K_ERR_DEVICE, api->err);
    CM_NCCL(api->CommInitRank(&c->nccl, nRanks, id, rank));
    c->useRccl = true;
    (void)hipGetLastError();  // RCCL probes devices and peers while it initialises; its leftovers are not errors of this library
  }
  return T1K_OK;
}

// a communicator outlives the contexts it is used with (a benchmark creates one job per step): point it at another context of the
// same device
int t1k_comm_bind(t1k_comm *c, t1k_ctx *ctx) {
  if (!c || !ctx || (c->ctx && c->ctx->device != ctx->device)) return T1K_ERR_ARG;
  if (c->tmp.p) { (void)t1k_dev_free(c->tmp.p); c->tmp.p = nullptr; c->tmp.bytes = 0; }
  if (c->ptrs.p) { (void)t1k_dev_free(c->ptrs.p); c->ptrs.p = nullptr; c->ptrs.bytes = 0; }
  c->ctx = ctx;
  return T1K_OK;
}

void t1k_comm_destroy(t1k_comm *c) {
  if (!c) return;
  if (c->nccl) { (void)rccl()->CommDestroy(c->nccl); (void)hipGetLastError(); }  // (an aborted communicator is gone already)
  if (c->ctx) (void)hipSetDevice(c->ctx->device);
  if (c->tmp.p) (void)t1k_dev_free(c->tmp.p);
  if (c->ptrs.p) (void)t1k_dev_free(c->ptrs.p);
  delete c;
}
const char *t1k_comm_last_error(const t1k_comm *c) { return c ? c->err.c_str() : "no communicator"; }
int t1k_comm_rank(const t1k_comm *c) { return c ? c->rank : 0; }
int t1k_comm_size(const t1k_comm *c) { return c ? c->nRanks : 1; }
int t1k_comm_is_rccl(const t1k_comm *c) { return c && c->useRccl ? 1 : 0; }

void t1k_comm_barrier_local(t1k_comm *c) { if (c && c->hub) (void)c->hub->barrier(); }

// A rank that cannot go on (a failed stage, out of memory, ...) says so before it returns: the in-process ranks waiting for it at a
// meeting point are released, RCCL operations in flight are torn down, and every later collective of the job returns T1K_ERR_STATE
// instead of waiting for a rank that will never come.
int t1k_comm_abort(t1k_comm *c) {
  if (!c) return T1K_ERR_ARG;
  c->aborted = true;
  if (c->hub) c->hub->abort();
  if (c->nccl && rccl()->CommAbort) { (void)rccl()->CommAbort(c->nccl); c->nccl = nullptr; (void)hipGetLastError(); }
  return T1K_OK;
}

// in place sum over the ranks of `count` elements at dev (kind 0: int32, 1: f64); blocking
int t1k_comm_allreduce(t1k_comm *c, void *dev, uint64_t count, int kind) {
  if (!c || !dev) return T1K_ERR_ARG;
  if (c->nRanks == 1 || count == 0) return T1K_OK;
  if (c->aborted) return commFail(c, T1K_ERR_STATE, "the job was aborted");
  t1k_ctx *ctx = c->ctx;
  CM_HIP(hipSetDevice(ctx->device));
  if (c->useRccl) {
    CM_NCCL(rccl()->AllReduce(dev, dev, count, kind == 0 ? ncclInt32 : ncclFloat64, ncclSum, c->nccl, ctx->stream));
    CM_HIP(hipStreamSynchronize(ctx->stream));
    return T1K_OK;
  }
  Hub &h = *c->hub;
  const size_t esz = kind == 0 ? 4 : 8;
  int rc;
  if ((rc = t1k_ensure(ctx, c->tmp, count * esz)) || (rc = t1k_ensure(ctx, c->ptrs, (size_t)c->nRanks * 8))) return commFail(c, rc, ctx->err);
  CM_HIP(hipStreamSynchronize(ctx->stream));
  h.ptr[c->rank] = dev;
  if (!h.barrier()) return commFail(c, T1K_ERR_STATE, "another rank of the job failed: the exchange was abandoned");
  std::vector<const void *> all(h.ptr.begin(), h.ptr.end());
  CM_HIP(hipMemcpyAsync(c->ptrs.p, all.data(), (size_t)c->nRanks * 8, hipMemcpyHostToDevice, ctx->stream));
  const unsigned nb = (unsigned)((count + 255) / 256);
  if (kind == 0) hipLaunchKernelGGL(k_sum_ranks_i32, dim3(nb), dim3(256), 0, ctx->stream, (int32_t *)c->tmp.p, (const int32_t *const *)c->ptrs.p, c->nRanks, count);
  else hipLaunchKernelGGL(k_sum_ranks_f64, dim3(nb), dim3(256), 0, ctx->stream, (double *)c->tmp.p, (const double *const *)c->ptrs.p, c->nRanks, count);
  CM_HIP(hipStreamSynchronize(ctx->stream));
  if (!h.barrier()) return commFail(c, T1K_ERR_STATE, "another rank of the job failed: the exchange was abandoned");  // everybody has read everybody's input
  CM_HIP(hipMemcpyAsync(dev, c->tmp.p, count * esz, hipMemcpyDeviceToDevice, ctx->stream));
  CM_HIP(hipStreamSynchronize(ctx->stream));
  if (!h.barrier()) return commFail(c, T1K_ERR_STATE, "another rank of the job failed: the exchange was abandoned");
  return T1K_OK;
}

// every rank contributes k 64-bit words (host), all[nRanks * k] receives them in rank order
int t1k_comm_allgather_u64(t1k_comm *c, const uint64_t *mine, uint32_t k, uint64_t *all) {
  if (!c || !mine || !all) return T1K_ERR_ARG;
  if (c->nRanks == 1) { memcpy(all, mine, (size_t)k * 8); return T1K_OK; }
  if (c->aborted) return commFail(c, T1K_ERR_STATE, "the job was aborted");
  t1k_ctx *ctx = c->ctx;
  CM_HIP(hipSetDevice(ctx->device));
  if (c->useRccl) {
    int rc;
    if ((rc = t1k_ensure(ctx, c->tmp, (size_t)(c->nRanks + 1) * k * 8))) return commFail(c, rc, ctx->err);
    uint64_t *dAll = (uint64_t *)c->tmp.p, *dMine = dAll + (size_t)c->nRanks * k;
    CM_HIP(hipMemcpyAsync(dMine, mine, (size_t)k * 8, hipMemcpyHostToDevice, ctx->stream));
    CM_NCCL(rccl()->AllGather(dMine, dAll, (size_t)k * 8, ncclUint8, c->nccl, ctx->stream));
    CM_HIP(hipMemcpyAsync(all, dAll, (size_t)c->nRanks * k * 8, hipMemcpyDeviceToHost, ctx->stream));
    CM_HIP(hipStreamSynchronize(ctx->stream));
    return T1K_OK;
  }
  Hub &h = *c->hub;
  h.off[c->rank] = mine;
  if (!h.barrier()) return commFail(c, T1K_ERR_STATE, "another rank of the job failed: the exchange was abandoned");
  for (int r = 0; r < c->nRanks; ++r) memcpy(all + (size_t)r * k, h.off[r], (size_t)k * 8);
  if (!h.barrier()) return commFail(c, T1K_ERR_STATE, "another rank of the job failed: the exchange was abandoned");
  return T1K_OK;
}

// all-to-all of device bytes: rank r's bytes [sendOff[p], sendOff[p+1]) of sendbuf go to rank p, where they land at recvOff[r]
// (recvOff[nRanks + 1] is this rank's view: what it gets from each peer; sizes must have been agreed on beforehand)
int t1k_comm_alltoallv(t1k_comm *c, const void *sendbuf, const uint64_t *sendOff, void *recvbuf, const uint64_t *recvOff) {
  if (!c || !sendOff || !recvOff) return T1K_ERR_ARG;
  if (c->aborted) return commFail(c, T1K_ERR_STATE, "the job was aborted");
  t1k_ctx *ctx = c->ctx;
  CM_HIP(hipSetDevice(ctx->device));
  const int N = c->nRanks;
  if (c->useRccl) {
    RcclApi *api = rccl();
    CM_NCCL(api->GroupStart());
    for (int p = 0; p < N; ++p) {
      const uint64_t sb = sendOff[p + 1] - sendOff[p], rb = recvOff[p + 1] - recvOff[p];
      if (sb) CM_NCCL(api->Send((const char *)sendbuf + sendOff[p], sb, ncclUint8, p, c->nccl, ctx->stream));
      if (rb) CM_NCCL(api->Recv((char *)recvbuf + recvOff[p], rb, ncclUint8, p, c->nccl, ctx->stream));
    }
    CM_NCCL(api->GroupEnd());
    CM_HIP(hipStreamSynchronize(ctx->stream));
    return T1K_OK;
  }
  if (N == 1) {
    const uint64_t b = sendOff[1] - sendOff[0];
    if (b) CM_HIP(hipMemcpyAsync((char *)recvbuf + recvOff[0], (const char *)sendbuf + sendOff[0], b, hipMemcpyDeviceToDevice, ctx->stream));
    CM_HIP(hipStreamSynchronize(ctx->stream));
    return T1K_OK;
  }
  Hub &h = *c->hub;
  CM_HIP(hipStreamSynchronize(ctx->stream));
  h.ptr[c->rank] = sendbuf; h.off[c->rank] = sendOff;
  if (!h.barrier()) return commFail(c, T1K_ERR_STATE, "another rank of the job failed: the exchange was abandoned");
  for (int p = 0; p < N; ++p) {  // pull what peer p holds for this rank
    const uint64_t *po = h.off[p];
    const uint64_t b = po[c->rank + 1] - po[c->rank];
    if (b) CM_HIP(hipMemcpyAsync((char *)recvbuf + recvOff[p], (const char *)h.ptr[p] + po[c->rank], b, hipMemcpyDefault, ctx->stream));
  }
  CM_HIP(hipStreamSynchronize(ctx->stream));
  if (!h.barrier()) return commFail(c, T1K_ERR_STATE, "another rank of the job failed: the exchange was abandoned");
  return T1K_OK;
}

// variable-length all-gather of device bytes: rank r's `bytes[r]` bytes land at out + displ[r] on every rank (mine == out + displ[rank]
// is allowed: in place)
int t1k_comm_allgatherv(t1k_comm *c, const void *mine, const uint64_t *bytes, const uint64_t *displ, void *out) {
  if (!c || !bytes || !displ) return T1K_ERR_ARG;
  if (c->aborted) return commFail(c, T1K_ERR_STATE, "the job was aborted");
  t1k_ctx *ctx = c->ctx;
  CM_HIP(hipSetDevice(ctx->device));
  const int N = c->nRanks;
  if (c->useRccl) {
    RcclApi *api = rccl();
    CM_NCCL(api->GroupStart());
    for (int r = 0; r < N; ++r)
      if (bytes[r]) CM_NCCL(api->Broadcast(r == c->rank ? mine : (const void *)((char *)out + displ[r]), (char *)out + displ[r], bytes[r], ncclUint8, r, c->nccl, ctx->stream));
    CM_NCCL(api->GroupEnd());
    CM_HIP(hipStreamSynchronize(ctx->stream));
    return T1K_OK;
  }
  if (N == 1) {
    if (bytes[0]) CM_HIP(hipMemcpyAsync((char *)out + displ[0], mine, bytes[0], hipMemcpyDeviceToDevice, ctx->stream));
    CM_HIP(hipStreamSynchronize(ctx->stream));
    return T1K_OK;
  }
  Hub &h = *c->hub;
  CM_HIP(hipStreamSynchronize(ctx->stream));
  h.ptr[c->rank] = mine;
  if (!h.barrier()) return commFail(c, T1K_ERR_STATE, "another rank of the job failed: the exchange was abandoned");
  for (int r = 0; r < N; ++r)
    if (bytes[r] && (const char *)h.ptr[r] != (const char *)out + displ[r])  // (in place: this rank's piece is where it belongs already)
      CM_HIP(hipMemcpyAsync((char *)out + displ[r], h.ptr[r], bytes[r], hipMemcpyDefault, ctx->stream));
  CM_HIP(hipStreamSynchronize(ctx->stream));
  if (!h.barrier()) return commFail(c, T1K_ERR_STATE, "another rank of the job failed: the exchange was abandoned");
  return T1K_OK;
}

// the same for host memory (every rank's slice of one host array of `total` bytes), staged through the device
int t1k_comm_allgatherv_host(t1k_comm *c, void *host, const uint64_t *bytes, const uint64_t *displ, uint64_t total) {
  if (!c) return T1K_ERR_ARG;
  if (c->nRanks == 1 || total == 0) return T1K_OK;  // (an empty array may come with a null pointer)
  if (!host || !bytes || !displ) return commFail(c, T1K_ERR_ARG, "t1k_comm_allgatherv_host: bad arguments");
  t1k_ctx *ctx = c->ctx;
  CM_HIP(hipSetDevice(ctx->device));
  T1kDevBuf all, mine;
  int rc;
  if ((rc = t1k_ensure(ctx, all, total + 16)) || (rc = t1k_ensure(ctx, mine, bytes[c->rank] + 16))) { if (all.p) (void)t1k_dev_free(all.p); return commFail(c, rc, ctx->err); }
  hipError_t e = bytes[c->rank] ? hipMemcpyAsync(mine.p, (const char *)host + displ[c->rank], bytes[c->rank], hipMemcpyHostToDevice, ctx->stream) : hipSuccess;
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) {
    rc = t1k_comm_allgatherv(c, mine.p, bytes, displ, all.p);
    if (rc == T1K_OK) e = hipMemcpy(host, all.p, total, hipMemcpyDeviceToHost);
  }
  (void)t1k_dev_free(all.p); (void)t1k_dev_free(mine.p);
  if (e != hipSuccess) return commFail(c, T1K_ERR_DEVICE, hipGetErrorString(e));
  return rc;
}

}  // extern "C"
