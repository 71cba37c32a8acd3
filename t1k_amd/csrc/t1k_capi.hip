// t1k_amd/csrc/t1k_capi.hip -- C ABI (include/t1k_gpu.h), device stage layer: context, reference upload + index build,
// read upload, the AssignRead batch pipeline, downloads.  Kernels live in t1k_assign.hip / t1k_pair.hip / t1k_em.hip.
#include <sys/mman.h>
#include <map>
#include <mutex>
#include <unordered_map>
#include <algorithm>
#include <memory>
#include <thread>
#include <chrono>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include "t1k_dev.h"
#include "t1k_launch.h"

// ------------------------------------------------------------------------------------------------------------------
// Device memory of the library goes through a process-wide pool: a freed block is kept and handed out again for a request of
// similar size.  Fresh VRAM is zeroed by the driver at ~35 ms per GB (and memory a process gave back is zeroed again for its next
// owner), so a process that runs job after job -- a service, the benchmark's steps -- would otherwise pay seconds per job for
// memory it had a moment ago.  A one-job process (the executables) sees no difference.  T1K_POOL_GB bounds what is kept (default 192).
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct DevPool {
  std::mutex m;
  std::map<int, std::multimap<size_t, void *>> freeBlocks;  // per device, by size
  std::unordered_map<void *, std::pair<size_t, int>> live;   // block -> (size, device)
  size_t pooled = 0;
};
DevPool &devPool() { static DevPool *p = new DevPool(); return *p; }  // never destroyed: outlives every static destructor that frees through it
}  // namespace

hipError_t t1k_dev_malloc(void **out, size_t bytes) {
  DevPool &P = devPool();
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (bytes == 0) bytes = 16;
  {
    std::lock_guard<std::mutex> g(P.m);
    auto &fb = P.freeBlocks[dev];
    auto it = fb.lower_bound(bytes);
    if (it != fb.end() && it->first <= bytes + bytes / 2 + (64u << 20)) {
      *out = it->second;
      P.live[*out] = {it->first, dev};
      P.pooled -= it->first;
      fb.erase(it);
      return hipSuccess;
    }
  }
  // A fresh block.  The runtime takes device memory of its own while kernels run (queue scratch) and ABORTS the process when there is none
  // (DESIGN 9.0 item 10); what this pool holds back is invisible to it.  So when the new block would leave the driver with less than the
  // memory rule's reserve, the cached blocks go back first (a job's stale blocks after its buffers have grown; a previous job's blocks are
  // normally taken again by the requests above and never get here).
  if (bytes >= ((size_t)64 << 20)) {
    bool drop = false;
    { std::lock_guard<std::mutex> g(P.m); drop = !P.freeBlocks[dev].empty(); }
    size_t f = 0, t = 0;
    if (drop && hipMemGetInfo(&f, &t) == hipSuccess && f < bytes + std::max<size_t>((size_t)8 << 30, t / 25)) {
      std::vector<void *> back;
      {
        std::lock_guard<std::mutex> g(P.m);
        for (auto &kv : P.freeBlocks[dev]) { back.push_back(kv.second); P.pooled -= kv.first; }
        P.freeBlocks[dev].clear();
      }
      for (void *q : back) (void)hipFree(q);
    }
  }
  const auto tA = std::chrono::steady_clock::now();
  hipError_t e = hipMalloc(out, bytes);
  if (getenv("T1K_DEBUG_ALLOC") && bytes >= (64u << 20))
    fprintf(stderr, "[t1k alloc] hipMalloc %.2f GB: %.1f ms\n", bytes / 1073741824.0, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tA).count());
  if (e != hipSuccess) {  // out of memory: give the pool back to the driver and try once more
    (void)hipGetLastError();
    std::vector<void *> drop;
    {
      std::lock_guard<std::mutex> g(P.m);
      for (auto &kv : P.freeBlocks[dev]) drop.push_back(kv.second);
      for (auto &kv : P.freeBlocks[dev]) P.pooled -= kv.first;
      P.freeBlocks[dev].clear();
    }
    for (void *q : drop) (void)hipFree(q);
    e = hipMalloc(out, bytes);
  }
  if (e == hipSuccess) { std::lock_guard<std::mutex> g(P.m); P.live[*out] = {bytes, dev}; }
  return e;
}

hipError_t t1k_dev_free(void *p) {
  if (!p) return hipSuccess;
  DevPool &P = devPool();
  static const size_t limit = [] { const char *e = getenv("T1K_POOL_GB"); return (size_t)(e ? atof(e) : 192.0) << 30; }();
  {
    std::lock_guard<std::mutex> g(P.m);
    auto it = P.live.find(p);
    if (it != P.live.end()) {
      const size_t bytes = it->second.first;
      const int dev = it->second.second;
      P.live.erase(it);
      if (P.pooled + bytes <= limit) { P.freeBlocks[dev].emplace(bytes, p); P.pooled += bytes; return hipSuccess; }
    }
  }
  return hipFree(p);
}

// what the pool holds for `device` and would hand out again (or give back to the driver when a fresh allocation fails)
uint64_t t1k_pool_cached_bytes(int device) {
  DevPool &P = devPool();
  std::lock_guard<std::mutex> g(P.m);
  uint64_t bytes = 0;
  auto it = P.freeBlocks.find(device);
  if (it != P.freeBlocks.end()) for (auto &kv : it->second) bytes += kv.first;
  return bytes;
}

static uint64_t pinnedRelease();
// gives every cached block back to the driver (a long-lived process that is done with its jobs for now)
extern "C" uint64_t t1k_pool_release(void) {
  DevPool &P = devPool();
  std::vector<std::pair<int, void *>> drop;
  uint64_t bytes = 0;
  {
    std::lock_guard<std::mutex> g(P.m);
    for (auto &dev : P.freeBlocks) {
      for (auto &kv : dev.second) { drop.push_back({dev.first, kv.second}); bytes += kv.first; }
      dev.second.clear();
    }
    P.pooled = 0;
  }
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (auto &d : drop) { (void)hipSetDevice(d.first); (void)hipFree(d.second); }
  (void)hipSetDevice(cur);
  (void)pinnedRelease();  // the cached page-locked host buffers go with it (not counted: the result is device memory)
  return bytes;
}

// ------------------------------------------------------------------------------------------------------------------
// Page-locked host buffers, cached per process like the device blocks.  A window's read text (2.5 GB at 10 M pairs) handed to
// hipMemcpyAsync from pageable memory is staged by the runtime on the calling thread -- 0.29 s, during which the pipelines' counter
// fetches on other streams queue behind its pieces (ranges that overlapped the upload took 80 - 100 ms instead of 48).  From a
// pinned buffer the copy is one DMA.  Pinning costs about as much as the staged copy did, once per process.
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct PinPool {
  std::mutex m;
  std::multimap<size_t, void *> freeBlocks;
  std::unordered_map<void *, size_t> live;
};
PinPool &pinPool() { static PinPool *p = new PinPool(); return *p; }
}  // namespace
extern "C" void *t1k_pinned_alloc(uint64_t bytes) {
  PinPool &P = pinPool();
  if (bytes == 0) bytes = 16;
  {
    std::lock_guard<std::mutex> g(P.m);
    auto it = P.freeBlocks.lower_bound((size_t)bytes);
    if (it != P.freeBlocks.end()) {
      void *p = it->second;
      P.live[p] = it->first;
      P.freeBlocks.erase(it);
      return p;
    }
  }
  const size_t want = ((size_t)bytes + (size_t)bytes / 8 + (64u << 20)) & ~(size_t)((1u << 20) - 1);  // room to grow: the next window of this size class fits too
  void *p = nullptr;
  if (hipHostMalloc(&p, want, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  std::lock_guard<std::mutex> g(P.m);
  P.live[p] = want;
  return p;
}
extern "C" void t1k_pinned_free(void *p) {
  if (!p) return;
  PinPool &P = pinPool();
  std::lock_guard<std::mutex> g(P.m);
  auto it = P.live.find(p);
  if (it == P.live.end()) return;
  P.freeBlocks.insert({it->second, p});
  P.live.erase(it);
}
static uint64_t pinnedRelease() {
  PinPool &P = pinPool();
  std::vector<void *> drop;
  uint64_t bytes = 0;
  {
    std::lock_guard<std::mutex> g(P.m);
    for (auto &kv : P.freeBlocks) { drop.push_back(kv.second); bytes += kv.first; }
    P.freeBlocks.clear();
  }
  for (void *q : drop) (void)hipHostFree(q);
  return bytes;
}

int t1k_fail(t1k_ctx *ctx, int code, const std::string &msg) {
  if (ctx) ctx->err = msg;
  return code;
}

int t1k_ensure(t1k_ctx *ctx, T1kDevBuf &b, size_t bytes) {
  if (bytes == 0) bytes = 16;
  if (b.bytes >= bytes) return 0;
  // a buffer that has to grow grows by at least half (the sizes of several arenas follow the data of each range and creep up: every
  // reallocation is fresh VRAM, which the driver zeroes at ~35 ms per GB)
  size_t want = std::max(bytes + bytes / 8, b.p ? b.bytes + b.bytes / 2 : (size_t)0) + 256;
  if (b.p) {
    // Kernels launched earlier on this context's stream may still be reading the old block, and the pool hands a freed block to the
    // next caller (another pipeline's thread) at once -- unlike hipFree, which waits for the device: drain the stream first.
    if (ctx && ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    (void)t1k_dev_free(b.p); b.p = nullptr; b.bytes = 0;
  }
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t e = t1k_dev_malloc(&b.p, want);
  if (ctx) { ctx->msAlloc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); ctx->bytesAlloc += want; }
  if (e != hipSuccess) { b.p = nullptr; return t1k_fail(ctx, T1K_ERR_DEVICE, std::string("t1k_dev_malloc(") + std::to_string(want) + "): " + hipGetErrorString(e)); }
  b.bytes = want;
  return 0;
}

static double nowMs() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// listPtr[i] = device address of read-end i's first PACKED overlap record in the store, listCount[i] = its length (after k_truncate)
__global__ void k_publish_lists(unsigned long long *listPtr, uint32_t *listCount, const T1kOvlP *store, const uint32_t *ovlStart, const uint32_t *ovlCount, uint32_t n,
                                const uint8_t *skip) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (skip && skip[i]) return;  // its entry comes from an earlier window (t1k_xwin_resolve)
  listPtr[i] = (unsigned long long)(store + ovlStart[i]);
  listCount[i] = ovlCount[i];
}
// the range's working records -> the store's packed form, same index (the gaps k_truncate left are copied along, harmlessly)
__global__ void k_pack_overlaps(const T1kOvl *work, T1kOvlP *store, uint64_t nOvl, unsigned long long *counters) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nOvl) return;
  T1kOvlP p;
  if (!t1k_ovl_pack(work[g], p)) atomicOr(&counters[2], 1024ull);
  store[g] = p;
}
static void t1k_launch_publish_lists(t1k_ctx *ctx, unsigned long long *listPtr, uint32_t *listCount, const T1kOvl *work, T1kOvlP *store, uint64_t nOvl,
                                     const uint32_t *ovlStart, const uint32_t *ovlCount, uint32_t n, unsigned long long *counters, const uint8_t *skip) {
  if (nOvl) hipLaunchKernelGGL(k_pack_overlaps, dim3((unsigned)((nOvl + 255) / 256)), dim3(256), 0, ctx->stream, work, store, nOvl, counters);
  if (n) hipLaunchKernelGGL(k_publish_lists, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, listPtr, listCount, store, ovlStart, ovlCount, n, skip);
}


extern "C" {

void t1k_params_default(t1k_params *p) {
  memset(p, 0, sizeof(*p));
  p->kmer_length = 11;
  p->radius = 10;
  p->hit_len_required = 31;
  p->ref_seq_similarity = 0.8;
  p->relax_intron_align = 0;
  p->max_assign_cnt = 2000;
  p->max_read_len = T1K_LONG_READ_LEN;
  p->workgroups = 2048;
  p->n_base_code = 3;
  p->store_chunk_mb = 1536;
  p->group_cap = 160ll << 20;
  p->cand_cap = 128ll << 20;
  p->ovl_cap = 96ll << 20;
  p->row_cap = 48ll << 20;
}

int t1k_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int t1k_ctx_create(int device, const t1k_params *params, t1k_ctx **out) {
  if (!out) return T1K_ERR_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return T1K_ERR_DEVICE;  // no GPU: fail loudly, there is no CPU path
  if (hipSetDevice(device) != hipSuccess) return T1K_ERR_DEVICE;
  t1k_ctx *ctx = new t1k_ctx();
  ctx->device = device;
  if (params) ctx->prm = *params; else t1k_params_default(&ctx->prm);
  t1k_params d;
  t1k_params_default(&d);
  if (ctx->prm.kmer_length <= 0) ctx->prm.kmer_length = d.kmer_length;
  if (ctx->prm.radius <= 0) ctx->prm.radius = d.radius;
  if (ctx->prm.hit_len_required <= 0) ctx->prm.hit_len_required = d.hit_len_required;
  if (ctx->prm.max_assign_cnt == 0) ctx->prm.max_assign_cnt = d.max_assign_cnt;
  if (ctx->prm.max_read_len <= 0) ctx->prm.max_read_len = d.max_read_len;
  if (ctx->prm.workgroups <= 0) ctx->prm.workgroups = d.workgroups;
  if (ctx->prm.group_cap <= 0) ctx->prm.group_cap = d.group_cap;

  if (ctx->prm.cand_cap <= 0) ctx->prm.cand_cap = d.cand_cap;
  if (ctx->prm.ovl_cap <= 0) ctx->prm.ovl_cap = d.ovl_cap;
  if (ctx->prm.row_cap <= 0) ctx->prm.row_cap = d.row_cap;
  if (ctx->prm.store_chunk_mb <= 0) ctx->prm.store_chunk_mb = d.store_chunk_mb;
  if (ctx->prm.kmer_length > 15 || ctx->prm.max_read_len > T1K_LONG_READ_LEN) { delete ctx; return T1K_ERR_ARG; }  // read coordinates < 1024 in the packed overlap records (t1k_dev.h)
  if (hipStreamCreate(&ctx->stream) != hipSuccess) { delete ctx; return T1K_ERR_DEVICE; }
  for (auto &e : ctx->ev) if (hipEventCreate(&e) != hipSuccess) { delete ctx; return T1K_ERR_DEVICE; }
  *out = ctx;
  return T1K_OK;
}

static void freeBuf(T1kDevBuf &b) {
  if (b.p) (void)t1k_dev_free(b.p);
  b.p = nullptr; b.bytes = 0;
}

void t1k_ctx_destroy(t1k_ctx *ctx) {
  if (ctx && ctx->emPinned) { (void)hipHostFree(ctx->emPinned); ctx->emPinned = nullptr; }
  if (ctx && ctx->countersPinned) { (void)hipHostFree(ctx->countersPinned); ctx->countersPinned = nullptr; }
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);  // nothing of this context may be in flight when its blocks go back to the pool
  for (auto &b : ctx->refBufs) freeBuf(b);
  T1kDevBuf *all[] = {&ctx->bReadAscii, &ctx->bReadOffs, &ctx->bReadBases, &ctx->bReadN, &ctx->bReadLen, &ctx->bReadWeight, &ctx->bWgHits, &ctx->bWgGroups,
                      &ctx->bWgStage, &ctx->bWgBig, &ctx->bWgCache, &ctx->bLists, &ctx->bCand, &ctx->bExt, &ctx->bCandStart, &ctx->bCandCount, &ctx->bListPtr, &ctx->bListCount,
                      &ctx->bOvlWork, &ctx->bDedupScratch, &ctx->bDedupBases, &ctx->bDedupN, &ctx->bDedupLen, &ctx->bDedupWeight, &ctx->bOvlStart, &ctx->bOvlCount, &ctx->bCounters, &ctx->bSlowQueue, &ctx->bSlowScratch, &ctx->bSortScratch, &ctx->bEqTrace, &ctx->bSortTmp, &ctx->bSlowKeys, &ctx->bJobSort, &ctx->bEnd1, &ctx->bEnd2,
                      &ctx->bHasN, &ctx->bRows, &ctx->bRowStart, &ctx->bRowCount, &ctx->bFragAssigned, &ctx->bPairScratch, &ctx->bPairOverflow, &ctx->bPairBig, &ctx->bExtractHuge, &ctx->bEmRowPtr, &ctx->bEmEc,
                      &ctx->bEmCount, &ctx->bEmLen, &ctx->bEmX0, &ctx->bEmN, &ctx->bEmPsum, &ctx->bEmColPtr, &ctx->bEmRowOf, &ctx->bExtract};
  for (auto *b : all) freeBuf(*b);
  for (auto &slot : ctx->storeChunks)
    for (auto &b : slot) freeBuf(b);
  for (auto &b : ctx->bAlign) freeBuf(b);
  for (auto &e : ctx->ev) if (e) (void)hipEventDestroy(e);
  for (auto &e : ctx->upEv) if (e) (void)hipEventDestroy(e);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char *t1k_last_error(const t1k_ctx *ctx) { return ctx ? ctx->err.c_str() : "no context"; }

// T1K_DEBUG_MEM: the device blocks a context holds right now, by name (what bounds the job's device memory: VERDICT round 5 item 7)
uint64_t t1k_ctx_mem_report(t1k_ctx *ctx, const char *tag, int print) {
  if (!ctx) return 0;
#define B(x) {#x, &ctx->x}
  const struct { const char *name; T1kDevBuf *b; } all[] = {B(bReadAscii), B(bReadOffs), B(bReadBases), B(bReadN), B(bReadLen), B(bReadWeight), B(bWgHits), B(bWgGroups),
    B(bWgStage), B(bWgBig), B(bWgCache), B(bLists), B(bCand), B(bExt), B(bCandStart), B(bCandCount), B(bListPtr), B(bListCount), B(bOvlWork), B(bDedupScratch), B(bDedupBases), B(bDedupN),
    B(bDedupLen), B(bDedupWeight), B(bOvlStart), B(bOvlCount), B(bCounters), B(bSlowQueue), B(bSlowScratch), B(bSortScratch), B(bEqTrace), B(bSortTmp), B(bSlowKeys), B(bJobSort), B(bEnd1),
    B(bEnd2), B(bHasN), B(bRows), B(bRowStart), B(bRowCount), B(bFragAssigned), B(bPairScratch), B(bPairOverflow), B(bPairBig), B(bExtractHuge), B(bEmRowPtr), B(bEmEc), B(bEmCount),
    B(bEmLen), B(bEmX0), B(bEmN), B(bEmPsum), B(bEmColPtr), B(bEmRowOf), B(bExtract)};
#undef B
  uint64_t tot = 0, ref = 0, store = 0, align = 0;
  std::string line;
  for (auto &e : all) if (e.b->p) { tot += e.b->bytes; if (e.b->bytes >= (64u << 20)) { char t[96]; snprintf(t, sizeof t, " %s %.0f", e.name + 1, e.b->bytes / 1048576.0); line += t; } }
  for (auto &b : ctx->refBufs) if (b.p) ref += b.bytes;
  for (auto &slot : ctx->storeChunks) for (auto &b : slot) if (b.p) store += b.bytes;
  for (auto &b : ctx->bAlign) if (b.p) align += b.bytes;
  if (print)
    fprintf(stderr, "[t1k mem] %s: %.2f GB in named blocks + reference %.2f + overlap store %.2f + alignment phase %.2f GB; blocks of 64 MB and more (MB):%s\n", tag ? tag : "context",
            tot / 1073741824.0, ref / 1073741824.0, store / 1073741824.0, align / 1073741824.0, line.c_str());
  return tot + ref + store + align;
}

// ------------------------------------------------------------------------------------------------------------------
// reference (t1k_ref_upload: t1k_refindex.hip)
// ------------------------------------------------------------------------------------------------------------------
int t1k_missing_coverage(t1k_ctx *ctx, int32_t *missing) {
  if (!ctx || !ctx->ref.covDiff || !missing) return t1k_fail(ctx, T1K_ERR_STATE, "no reference");
  T1K_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t A = ctx->ref.nAlleles;
  T1kDevBuf dScratch, dMiss;
  int rc;
  const double t0 = nowMs();
  if ((rc = t1k_ensure(ctx, dScratch, (ctx->ref.totalBases + 2) * 4))) return rc;
  if ((rc = t1k_ensure(ctx, dMiss, (size_t)A * 4))) { freeBuf(dScratch); return rc; }
  const double t1 = nowMs();
  if ((rc = t1k_coverage_fold(ctx))) { freeBuf(dScratch); freeBuf(dMiss); return rc; }
  t1k_launch_missing_coverage(ctx, ctx->ref, (int32_t *)dScratch.p, (int32_t *)dMiss.p);
  hipMemcpyAsync(missing, dMiss.p, (size_t)A * 4, hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (getenv("T1K_DEBUG_PHASES")) fprintf(stderr, "[t1k] missing coverage: buffers %.1f ms, kernel + copy %.1f ms\n", t1 - t0, nowMs() - t1);
  freeBuf(dScratch); freeBuf(dMiss);
  if (e != hipSuccess) return t1k_fail(ctx, T1K_ERR_DEVICE, hipGetErrorString(e));
  return T1K_OK;
}
int t1k_coverage_device(t1k_ctx *ctx, void **devPtr, uint64_t *count) {
  if (!ctx || !ctx->ref.covDiff || !devPtr || !count) return t1k_fail(ctx, T1K_ERR_STATE, "no reference");
  T1K_HIP(ctx, hipSetDevice(ctx->device));
  if (int rc = t1k_coverage_fold(ctx)) return rc;
  T1K_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *devPtr = ctx->ref.covDiff;
  *count = 2 * ctx->ref.covStride;  // difference array and hole array, contiguous: both are additive over GPUs
  return T1K_OK;
}

int t1k_ref_share(t1k_ctx *dst, const t1k_ctx *src) {
  if (!dst || !src || dst == src || !src->ref.bases || dst->device != src->device) return t1k_fail(dst, T1K_ERR_ARG, "t1k_ref_share: contexts do not match");
  if (dst->prm.kmer_length != src->prm.kmer_length) return t1k_fail(dst, T1K_ERR_ARG, "t1k_ref_share: different k-mer length");
  T1K_HIP(dst, hipSetDevice(dst->device));
  for (auto &b : dst->refBufs) freeBuf(b);
  dst->refBufs.clear();
  dst->ref = src->ref;  // read-only arrays are aliased; they stay owned by src, which must outlive dst
  dst->hAlleleOff = src->hAlleleOff;
  dst->hAlleleLen = src->hAlleleLen;
  T1kDevBuf cov;  // the coverage difference array is per context
  int rc;
  if ((rc = t1k_ensure(dst, cov, 3 * src->ref.covStride * sizeof(int32_t)))) return rc;
  T1K_HIP(dst, hipMemsetAsync(cov.p, 0, 3 * src->ref.covStride * sizeof(int32_t), dst->stream));
  T1K_HIP(dst, hipStreamSynchronize(dst->stream));
  dst->refBufs.push_back(cov);
  dst->ref.covDiff = (int32_t *)cov.p;
  dst->covFullLen = 0; dst->covFullDirty = false;
  return T1K_OK;
}
int t1k_reads_attach(t1k_ctx *dst, const t1k_ctx *src, int storeSlot, int resetStore) {
  if (!dst || !src || dst == src || dst->device != src->device || storeSlot < 0 || storeSlot > 1) return t1k_fail(dst, T1K_ERR_ARG, "t1k_reads_attach: contexts do not match");
  dst->reads = src->reads;  // packed read-ends are read-only for the stages and the list table is shared; both are owned by src
  dst->readsShared = true;
  dst->batchMaxLen = src->batchMaxLen; dst->batchFastMaxLen = src->batchFastMaxLen;
  dst->nCand = dst->nOvl = 0;
  dst->rangeCount = 0;
  dst->storeSlot = storeSlot;
  if (resetStore) { dst->storeChunk[storeSlot] = 0; dst->storeUsed[storeSlot] = 0; }
  return T1K_OK;
}
int t1k_reads_share(t1k_ctx *dst, const t1k_ctx *src) { return t1k_reads_attach(dst, src, 0, 1); }
int t1k_coverage_absorb(t1k_ctx *dst, t1k_ctx *src) {
  if (!dst || !src || !dst->ref.covDiff || !src->ref.covDiff || dst->device != src->device || dst->ref.totalBases != src->ref.totalBases)
    return t1k_fail(dst, T1K_ERR_ARG, "t1k_coverage_absorb: contexts do not match");
  T1K_HIP(dst, hipSetDevice(dst->device));
  if (int rc = t1k_coverage_fold(src)) return t1k_fail(dst, rc, "t1k_coverage_absorb: fold");
  T1K_HIP(dst, hipStreamSynchronize(src->stream));
  t1k_launch_coverage_add(dst, dst->ref.covDiff, src->ref.covDiff, 2 * dst->ref.covStride);
  T1K_HIP(dst, hipStreamSynchronize(dst->stream));
  return T1K_OK;
}
int t1k_coverage_reset(t1k_ctx *ctx) {
  if (!ctx || !ctx->ref.covDiff) return t1k_fail(ctx, T1K_ERR_STATE, "no reference");
  T1K_HIP(ctx, hipSetDevice(ctx->device));
  T1K_HIP(ctx, hipMemsetAsync(ctx->ref.covDiff, 0, 3 * ctx->ref.covStride * sizeof(int32_t), ctx->stream));
  T1K_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->covFullDirty = false;
  return T1K_OK;
}

int t1k_coverage_get(t1k_ctx *ctx, int32_t *out, uint64_t cap) {
  if (!ctx || !ctx->ref.covDiff || !out) return t1k_fail(ctx, T1K_ERR_STATE, "no reference");
  T1K_HIP(ctx, hipSetDevice(ctx->device));
  uint32_t A = ctx->ref.nAlleles;
  std::vector<uint64_t> outOff(A);
  uint64_t tot = 0;
  for (uint32_t a = 0; a < A; ++a) { outOff[a] = tot; tot += ctx->hAlleleLen[a]; }
  if (cap < tot) return t1k_fail(ctx, T1K_ERR_ARG, "coverage buffer too small");
  T1kDevBuf dOut, dOff;
  int rc;
  if ((rc = t1k_ensure(ctx, dOut, tot * 4))) return rc;
  if ((rc = t1k_ensure(ctx, dOff, A * 8))) { freeBuf(dOut); return rc; }
  hipMemcpyAsync(dOff.p, outOff.data(), A * 8, hipMemcpyHostToDevice, ctx->stream);
  if ((rc = t1k_coverage_fold(ctx))) { freeBuf(dOut); freeBuf(dOff); return rc; }
  t1k_launch_coverage_scan(ctx, ctx->ref, (int32_t *)dOut.p, (const uint64_t *)dOff.p);
  hipMemcpyAsync(out, dOut.p, tot * 4, hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  freeBuf(dOut); freeBuf(dOff);
  if (e != hipSuccess) return t1k_fail(ctx, T1K_ERR_DEVICE, hipGetErrorString(e));
  return T1K_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// reads
// ------------------------------------------------------------------------------------------------------------------
// The upload in three steps, so that a caller can send the text in pieces through a small page-locked staging buffer while it is
// still gathering the rest (host/job.cpp: a window's text is 2.5 GB at 10 M pairs; t1k_reads_upload is begin + two pieces + end).
static int readsUploadBegin(t1k_ctx *ctx, uint32_t n, uint64_t bytes, int maxLen) {
  T1K_HIP(ctx, hipSetDevice(ctx->device));
  if (maxLen > ctx->prm.max_read_len) return t1k_fail(ctx, T1K_ERR_ARG, "read longer than max_read_len");
  int S = (maxLen + 31) / 32 + 1;
  if (S < 2) S = 2;
  int rc;
  if ((rc = t1k_ensure(ctx, ctx->bReadAscii, bytes + 16))) return rc;
  if ((rc = t1k_ensure(ctx, ctx->bReadOffs, (size_t)(n + 1) * 8))) return rc;
  if ((rc = t1k_ensure(ctx, ctx->bReadBases, (size_t)n * 2 * S * 8 + 64))) return rc;
  if ((rc = t1k_ensure(ctx, ctx->bReadN, (size_t)n * 2 * S * 8 + 64))) return rc;
  if ((rc = t1k_ensure(ctx, ctx->bReadLen, (size_t)n * 2 + 16))) return rc;
  if ((rc = t1k_ensure(ctx, ctx->bReadWeight, (size_t)n * 4 + 16))) return rc;
  if ((rc = t1k_ensure(ctx, ctx->bListPtr, (size_t)n * 8 + 16))) return rc;
  if ((rc = t1k_ensure(ctx, ctx->bListCount, (size_t)n * 4 + 16))) return rc;
  T1K_HIP(ctx, hipMemsetAsync(ctx->bListCount.p, 0, (size_t)n * 4 + 16, ctx->stream));  // a read-end that was never assigned has an empty list
  T1K_HIP(ctx, hipMemsetAsync(ctx->bListPtr.p, 0, (size_t)n * 8 + 16, ctx->stream));
  ctx->upN = n; ctx->upS = S; ctx->upMaxLen = maxLen; ctx->upBytes = bytes; ctx->upOpen = true;
  return T1K_OK;
}
// longest read among those the fast kernels take (windows that hold a read beyond T1K_MAX_READ_LEN only)
__global__ void k_fast_max_len(const uint16_t *lens, uint32_t n, uint32_t *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t v = i < n && lens[i] <= T1K_MAX_READ_LEN ? lens[i] : 0u;
  for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
  if ((threadIdx.x & 63) == 0 && v) atomicMax(out, v);
}
static int readsUploadEnd(t1k_ctx *ctx, const uint32_t *weights) {
  const uint32_t n = ctx->upN;
  const int S = ctx->upS;
  uint32_t fastMax = 0;
  if (n) {
    if (weights) T1K_HIP(ctx, hipMemcpyAsync(ctx->bReadWeight.p, weights, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    else T1K_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)ctx->bReadWeight.p, 1, (size_t)n, ctx->stream));  // every read-end counts once
    // the spare word after each strand must be defined (funnel shifts read it)
    T1K_HIP(ctx, hipMemsetAsync(ctx->bReadBases.p, 0, (size_t)n * 2 * S * 8 + 64, ctx->stream));
    T1K_HIP(ctx, hipMemsetAsync(ctx->bReadN.p, 0, (size_t)n * 2 * S * 8 + 64, ctx->stream));
    t1k_launch_pack(ctx, (const char *)ctx->bReadAscii.p, (const uint64_t *)ctx->bReadOffs.p, n, S, (uint64_t *)ctx->bReadBases.p, (uint64_t *)ctx->bReadN.p,
                    (uint16_t *)ctx->bReadLen.p, ctx->prm.n_base_code & 3);
    if (ctx->upMaxLen > T1K_MAX_READ_LEN) {
      uint32_t *d = (uint32_t *)ctx->bListCount.p + n;  // (spare words behind the table)
      T1K_HIP(ctx, hipMemsetAsync(d, 0, 4, ctx->stream));
      hipLaunchKernelGGL(k_fast_max_len, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (const uint16_t *)ctx->bReadLen.p, n, d);
      T1K_HIP(ctx, hipMemcpyAsync(&fastMax, d, 4, hipMemcpyDeviceToHost, ctx->stream));
    }
  }
  T1K_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->upOpen = false;
  ctx->reads.nReadEnds = n;
  ctx->reads.S = S;
  ctx->reads.bases = (const uint64_t *)ctx->bReadBases.p;
  ctx->reads.nmask = (const uint64_t *)ctx->bReadN.p;
  ctx->reads.len = (const uint16_t *)ctx->bReadLen.p;
  ctx->reads.weight = (const uint32_t *)ctx->bReadWeight.p;
  ctx->reads.listPtr = (unsigned long long *)ctx->bListPtr.p;
  ctx->reads.listCount = (uint32_t *)ctx->bListCount.p;
  ctx->reads.skip = nullptr;
  ctx->readsShared = false;
  ctx->storeSlot = 0; ctx->storeChunk[0] = 0; ctx->storeUsed[0] = 0;  // the lists of the previous read set are dead
  ctx->batchMaxLen = ctx->upMaxLen;
  ctx->batchFastMaxLen = ctx->upMaxLen > T1K_MAX_READ_LEN ? (int)fastMax : ctx->upMaxLen;
  ctx->nCand = ctx->nOvl = 0;
  ctx->rangeCount = 0;
  return T1K_OK;
}
int t1k_reads_upload(t1k_ctx *ctx, const char *seqs, const uint64_t *offsets, const uint32_t *weights, uint32_t n) {
  if (!ctx || (!seqs && n) || !offsets) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_reads_upload: bad arguments");
  int maxLen = 0;
  for (uint32_t i = 0; i < n; ++i) maxLen = std::max<int>(maxLen, (int)(offsets[i + 1] - offsets[i]));
  const uint64_t bytes = n ? offsets[n] : 0;
  int rc;
  if ((rc = readsUploadBegin(ctx, n, bytes, maxLen))) return rc;
  if (n) {
    T1K_HIP(ctx, hipMemcpyAsync(ctx->bReadAscii.p, seqs, bytes, hipMemcpyHostToDevice, ctx->stream));
    T1K_HIP(ctx, hipMemcpyAsync(ctx->bReadOffs.p, offsets, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  }
  return readsUploadEnd(ctx, weights);
}
extern "C" int t1k_reads_upload_begin(t1k_ctx *ctx, uint32_t nReadEnds, uint64_t textBytes, int maxReadLen) {
  if (!ctx || maxReadLen < 0) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_reads_upload_begin: bad arguments");
  return readsUploadBegin(ctx, nReadEnds, textBytes, maxReadLen);
}
extern "C" int t1k_reads_upload_piece(t1k_ctx *ctx, int what, const void *src, uint64_t byteOffset, uint64_t bytes, int slot) {
  if (!ctx || !ctx->upOpen) return t1k_fail(ctx, T1K_ERR_STATE, "t1k_reads_upload_piece: no upload begun");
  if ((what != 0 && what != 1) || slot < 0 || slot >= 4 || (!src && bytes)) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_reads_upload_piece: bad arguments");
  const uint64_t limit = what == 0 ? ctx->upBytes : ((uint64_t)ctx->upN + 1) * 8;
  if (byteOffset + bytes > limit) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_reads_upload_piece: piece outside the announced size");
  T1K_HIP(ctx, hipSetDevice(ctx->device));
  char *dst = (char *)(what == 0 ? ctx->bReadAscii.p : ctx->bReadOffs.p) + byteOffset;
  if (bytes) T1K_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  if (!ctx->upEv[slot]) T1K_HIP(ctx, hipEventCreateWithFlags(&ctx->upEv[slot], hipEventDisableTiming));
  T1K_HIP(ctx, hipEventRecord(ctx->upEv[slot], ctx->stream));
  ctx->upEvSet[slot] = true;
  return T1K_OK;
}
extern "C" int t1k_reads_upload_wait(t1k_ctx *ctx, int slot) {
  if (!ctx || slot < 0 || slot >= 4) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_reads_upload_wait: bad arguments");
  if (ctx->upEvSet[slot]) { T1K_HIP(ctx, hipEventSynchronize(ctx->upEv[slot])); ctx->upEvSet[slot] = false; }
  return T1K_OK;
}
extern "C" int t1k_reads_upload_end(t1k_ctx *ctx) {
  if (!ctx || !ctx->upOpen) return t1k_fail(ctx, T1K_ERR_STATE, "t1k_reads_upload_end: no upload begun");
  for (int i = 0; i < 4; ++i) ctx->upEvSet[i] = false;  // the end drains the stream
  return readsUploadEnd(ctx, nullptr);
}

// ------------------------------------------------------------------------------------------------------------------
// candidate extraction over the batch (FastqExtractor.cpp:113-118 IsGoodCandidate, SeqSet.hpp:1915-1990 HasHitInSet)
// ------------------------------------------------------------------------------------------------------------------
int t1k_extract_batch(t1k_ctx *ctx, uint32_t endsPerFragment, uint8_t *good, uint64_t *stats) {
  if (!ctx || !good || (endsPerFragment != 1 && endsPerFragment != 2)) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_extract_batch: bad arguments");
  if (!ctx->ref.kStart) return t1k_fail(ctx, T1K_ERR_STATE, "t1k_extract_batch: no reference uploaded");
  const uint32_t nEnds = ctx->reads.nReadEnds;
  if (nEnds % endsPerFragment) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_extract_batch: odd number of read-ends in a paired batch");
  const uint32_t nFrag = nEnds / endsPerFragment;
  if (stats) memset(stats, 0, 8 * sizeof(uint64_t));
  if (!nFrag) return T1K_OK;
  T1K_HIP(ctx, hipSetDevice(ctx->device));
  const size_t flagBytes = ((size_t)nFrag + 63) / 64 * 64;
  int rc;
  const size_t stateBytes = ((size_t)nEnds + 63) / 64 * 64;
  if ((rc = t1k_ensure(ctx, ctx->bExtract, flagBytes + 128 + stateBytes))) return rc;
  uint8_t *dGood = (uint8_t *)ctx->bExtract.p;
  unsigned long long *dCtl = (unsigned long long *)(dGood + flagBytes);  // [0] error flags, [1..5] statistics
  uint8_t *dState = dGood + flagBytes + 128;                            // per read-end verdict of the screen
  T1K_HIP(ctx, hipMemsetAsync(dCtl, 0, 128, ctx->stream));
  const uint32_t maxK = (uint32_t)((2 * std::max(1, ctx->batchMaxLen - ctx->prm.kmer_length + 1) + 3) / 4 * 4);
  t1k_launch_extract(ctx, ctx->ref, ctx->reads, ctx->prm.kmer_length, ctx->prm.radius, ctx->prm.hit_len_required, 1 - ctx->prm.ref_seq_similarity, nFrag,
                     endsPerFragment, maxK, dGood, dState, dCtl, dCtl + 1, ctx->prm.workgroups * 4);
  T1K_HIP(ctx, hipGetLastError());
  unsigned long long ctl[16];
  T1K_HIP(ctx, hipMemcpyAsync(ctl, dCtl, 128, hipMemcpyDeviceToHost, ctx->stream));
  T1K_HIP(ctx, hipStreamSynchronize(ctx->stream));
  bool bigShape = false, hugeShape = false;
  if (ctl[0] || getenv("T1K_EXTRACT_FORCE_BIG") || getenv("T1K_EXTRACT_FORCE_HUGE")) {
    bigShape = true;
    // some (strand, sequence) bucket holds more hits than the production kernel keeps in LDS: the whole batch again in the large shape
    T1K_HIP(ctx, hipMemsetAsync(dCtl, 0, 128, ctx->stream));
    t1k_launch_extract_big(ctx, ctx->ref, ctx->reads, ctx->prm.kmer_length, ctx->prm.radius, ctx->prm.hit_len_required, 1 - ctx->prm.ref_seq_similarity, nFrag,
                           endsPerFragment, maxK, dGood, dState, dCtl, dCtl + 1, ctx->prm.workgroups * 4);
    T1K_HIP(ctx, hipGetLastError());
    T1K_HIP(ctx, hipMemcpyAsync(ctl, dCtl, 128, hipMemcpyDeviceToHost, ctx->stream));
    T1K_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctl[0] || getenv("T1K_EXTRACT_FORCE_HUGE")) {
      // ... and a bucket beyond the large LDS shape too (a read inside a long tandem repeat of one sequence): third attempt with the hit
      // arrays in HBM, 64 workgroups x 4 x 65 535 words (the LIS links are 16-bit: that many hits a bucket; more is still T1K_ERR_CAPACITY)
      const uint32_t cap = 65535u;
      const int hugeWg = 64;
      if ((rc = t1k_ensure(ctx, ctx->bExtractHuge, (size_t)hugeWg * 4 * cap * 4))) return rc;
      T1K_HIP(ctx, hipMemsetAsync(dCtl, 0, 128, ctx->stream));
      t1k_launch_extract_huge(ctx, ctx->ref, ctx->reads, ctx->prm.kmer_length, ctx->prm.radius, ctx->prm.hit_len_required, 1 - ctx->prm.ref_seq_similarity, nFrag,
                              endsPerFragment, maxK, dGood, dState, dCtl, dCtl + 1, hugeWg, (uint32_t *)ctx->bExtractHuge.p, cap);
      T1K_HIP(ctx, hipGetLastError());
      T1K_HIP(ctx, hipMemcpyAsync(ctl, dCtl, 128, hipMemcpyDeviceToHost, ctx->stream));
      hugeShape = true;
    }
  }
  T1K_HIP(ctx, hipMemcpyAsync(good, dGood, nFrag, hipMemcpyDeviceToHost, ctx->stream));
  T1K_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (getenv("T1K_XPROF")) {  // phase clocks of k_extract (builds with -DT1K_XPROF only): list, look-ups, replay, vote, gather, diagonal test, chain
    fprintf(stderr, "[t1k xprof]");
    for (int i = 0; i < 7; ++i) fprintf(stderr, " %.3g", (double)ctl[8 + i]);
    fprintf(stderr, "\n");
  }
  if (ctl[0]) return t1k_fail(ctx, T1K_ERR_CAPACITY, "t1k_extract_batch: a read has more than 65 535 hits on one reference sequence");
  if (stats) {
    for (int i = 0; i < 5; ++i) stats[i] = ctl[1 + i];
    stats[0] = nEnds;
    float msScreen = 0, msMain = 0;  // HIP events on the context's stream around each launch
    (void)hipEventElapsedTime(&msScreen, ctx->ev[0], ctx->ev[1]);
    (void)hipEventElapsedTime(&msMain, ctx->ev[1], ctx->ev[2]);
    stats[5] = (uint64_t)(msScreen * 1e6); stats[6] = (uint64_t)(msMain * 1e6); stats[7] = hugeShape ? 2 : bigShape ? 1 : 0;
  }
  return T1K_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// AssignRead over the batch
// ------------------------------------------------------------------------------------------------------------------
// control counters + the striped statistics folded into their historical slots (7 dp, 11 fast, 12 general, 14 extend dp, 10 near-best)
extern "C++" int t1k_fetch_counters(t1k_ctx *ctx, unsigned long long *h) {
  std::vector<unsigned long long> &raw = ctx->hRaw;
  raw.resize(T1K_COUNTER_WORDS);
  if (getenv("T1K_DEBUG_TRACE")) fprintf(stderr, "[t1k trace] counter fetch %d of this batch\n", ++ctx->traceFetch);
  // a dozen of these per batch sit on the pipeline's critical path: page-locked landing buffer (a plain DMA, no staging)
  if (!ctx->countersPinned) T1K_HIP(ctx, hipHostMalloc((void **)&ctx->countersPinned, (size_t)T1K_COUNTER_WORDS * 8, hipHostMallocDefault));
  T1K_HIP(ctx, hipMemcpyAsync(ctx->countersPinned, ctx->bCounters.p, (size_t)T1K_COUNTER_WORDS * 8, hipMemcpyDeviceToHost, ctx->stream));
  T1K_HIP(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(raw.data(), ctx->countersPinned, (size_t)T1K_COUNTER_WORDS * 8);
  memcpy(h, raw.data(), 64 * 8);
  h[6] = 0;  // group records: sum of the arena's segment cursors
  for (int s = 0; s < T1K_NSTRIPE; ++s) h[6] += raw[T1K_ARENA_BASE + ((size_t)T1K_AR_GROUPS * T1K_NSTRIPE + s) * T1K_STRIPE_WORDS];
  for (int s = 56; s < 64; ++s) h[6] += raw[s];  // ... + the groups the fused seeding kernel ended without a record
  static const int slot[8] = {7, 11, 12, 14, 10, 3, 4, 5};
  for (int s = 0; s < T1K_STAT_STRIPES; ++s)
    for (int k = 0; k < 8; ++k) h[slot[k]] += raw[64 + s * 8 + k];
  if (getenv("T1K_SEED_PROFILE") && raw[48]) fprintf(stderr, "[t1k] seed phases (ticks, thread 0 of every workgroup): lookup %llu rule %llu setup+chunk-selection %llu [chunks visited %llu] slice %llu scan %llu walk %llu emit-scan %llu (record writes + reset are charged to slice)\n", raw[48], raw[49], raw[50], raw[51], raw[52], raw[53], raw[54], raw[55]);
  return 0;
}
static int fetchCounters(t1k_ctx *ctx, unsigned long long *h) { return t1k_fetch_counters(ctx, h); }

// An arena overflowed.  The cursors keep counting past their capacity, so the last fetched counter block holds the demand of the
// stages that ran: remember it, t1k_assign_range sizes its working capacities from it and runs the range again.
static int capacityError(t1k_ctx *ctx, unsigned long long flags) {
  // (a full job list releases memo claims other lanes may already wait on; with the launches behind it no longer held back by a counter
  // fetch, one of them can meet a slot left pending: a consequence of the overflow, cured by the rerun with lists that fit)
  if (flags & 256) flags &= ~512ull;
  ctx->lastCapFlags = flags;
  if (flags & 512) return t1k_fail(ctx, T1K_ERR_INTERNAL, "alignment memo entry left pending");
  if (flags & 1024) return t1k_fail(ctx, T1K_ERR_INTERNAL, "an overlap record does not fit the packed form of the overlap store");
  if (ctx->covCommitted) {
    // k_fullalign has already added this range's ungapped alignments to the coverage arrays: running the range again (larger arenas,
    // or split in two) would count them twice.  The context's coverage is void; the caller must not retry on it.
    return t1k_fail(ctx, T1K_ERR_COMMITTED, "an alignment queue or sort buffer overflowed after part of the range's coverage was added (flags " + std::to_string(flags) +
                                               "): rerun with smaller ranges (T1K_BATCH) -- this context's coverage is no longer valid");
  }
  if (ctx->hRaw.size() >= T1K_COUNTER_WORDS) {
    auto maxSeg = [&](int arena) {
      unsigned long long m = 0;
      for (int st = 0; st < T1K_NSTRIPE; ++st) m = std::max(m, ctx->hRaw[T1K_ARENA_BASE + ((size_t)arena * T1K_NSTRIPE + st) * T1K_STRIPE_WORDS]);
      return m;
    };
    unsigned long long lists = 0, rare = 0;
    for (int ar : {T1K_AR_SLOW, T1K_AR_RETRY, T1K_AR_FINISH, T1K_AR_EXTRETRY}) lists = std::max(lists, maxSeg(ar));
    for (int ar : {T1K_AR_GENERAL, T1K_AR_WAVE, T1K_AR_BIG}) rare = std::max(rare, maxSeg(ar));
    ctx->needGroup = maxSeg(T1K_AR_GROUPS) * T1K_NSTRIPE;
    ctx->needList = lists * T1K_NSTRIPE;
    ctx->needRare = rare * T1K_NSTRIPE;
    ctx->needJob = std::max(maxSeg(T1K_AR_JOBS), maxSeg(T1K_AR_EXTJOBS)) * T1K_NSTRIPE;
    ctx->needGenJob = maxSeg(T1K_AR_GENJOBS) * T1K_NSTRIPE;
    ctx->needGenHit = maxSeg(T1K_AR_GENHITS) * T1K_NSTRIPE;
    ctx->needCand = ctx->hRaw[0];
    ctx->needOvl = ctx->hRaw[1];
  }
  std::string m = "device arena overflow:";
  if (flags & 1) m += " hit_cap";
  if (flags & 2) m += " candidate staging";
  if (flags & 4) m += " cand_cap";
  if (flags & 8) m += " group too large";
  if (flags & 16) m += " ovl_cap";
  if (flags & 32) m += " sort capacity";
  if (flags & 64) m += " slow-alignment queue";
  if (flags & 128) m += " row_cap";
  if (flags & 256) m += " group_cap";
  if (flags & 512) return t1k_fail(ctx, T1K_ERR_INTERNAL, "alignment memo entry left pending");
  if (flags & 1024) return t1k_fail(ctx, T1K_ERR_INTERNAL, "an overlap record does not fit the packed form of the overlap store");
  return t1k_fail(ctx, T1K_ERR_CAPACITY, m);
}


// The near-best full alignments of `nOvl` working records (relaxed match counts + per-base coverage, SeqSet.hpp:2188-2285): k_fullalign
// (ungapped in closed form), then the queued DP alignments -- equal spans / spans differing by <= 4 sorted so that identical jobs share
// one traced fill, wider differences through the general DP.  noCov: the coverage updates are left out.  exactQueues: every stripe of
// the alignment queues can hold all records (no overflow possible; for callers that cannot run the records again).
extern "C++" int t1k_fullalign_phase(t1k_ctx *ctx, const T1kReadsDev &rd, T1kOvl *ovl, uint64_t nOvl, int relaxFlag, int noCov, int maxLen, bool exactQueues,
                                     unsigned long long *hc) {
  int rc;
  unsigned long long *counters = (unsigned long long *)ctx->bCounters.p;
  const int maxCells = std::max(340, maxLen + 24) * std::max(340, maxLen + 24);
  const int slowBlocks = 64;
  // alignment queues (equal spans | spans differing by <= 4 | wider): striped arenas appended to by k_fullalign, then made dense
  // (T1K_TEST_SMALL_QUEUES: the tests' way of making a stripe overflow)
  static const bool tinyQueues = getenv("T1K_TEST_SMALL_QUEUES") != nullptr;
  const uint32_t qSegCap = exactQueues ? (uint32_t)(nOvl + 64)
                                       : (uint32_t)std::min<uint64_t>(nOvl + 64, tinyQueues ? (uint64_t)8 * ctx->queueBoost : (nOvl / T1K_NSTRIPE * 2 + 1024) * ctx->queueBoost);
  const size_t qDense = (size_t)nOvl + 1, qStr = (size_t)qSegCap * T1K_NSTRIPE;
  // The group records, the work lists and the candidates are dead from here on: the alignment queues, their sort keys and the trace
  // rows live in that memory when it is large enough (every arena of its own is more fresh VRAM for the driver to zero).
  const size_t queueBytes = ((qDense + qStr) * 3 * 4 + 255) & ~(size_t)255, keyBytes = (qStr * 2 + qDense * 2) * 8 + qDense * 4;
  char *queueMem = nullptr;
  if (ctx->bWgGroups.bytes >= queueBytes + keyBytes) queueMem = (char *)ctx->bWgGroups.p;
  else {
    if ((rc = t1k_ensure(ctx, ctx->bSlowQueue, queueBytes + keyBytes))) return rc;
    queueMem = (char *)ctx->bSlowQueue.p;
  }
  FullArgs f{};
  f.ref = ctx->ref; f.reads = rd; f.relax = relaxFlag; f.noCov = noCov; f.ovl = ovl; f.nOvl = nOvl;
  uint32_t *qEq = (uint32_t *)queueMem, *qBand = qEq + qDense, *qWide = qBand + qDense;
  f.eqStr = qWide + qDense; f.bandStr = f.eqStr + qStr; f.wideStr = f.bandStr + qStr; f.segCap = qSegCap; f.counters = counters;
  // sort keys of the equal / band queues: striped | dense | sorted (scratch of the radix sort), and the unsorted dense gid lists
  f.eqKeyStr = (unsigned long long *)(queueMem + queueBytes); f.bandKeyStr = f.eqKeyStr + qStr;
  unsigned long long *kDense = f.bandKeyStr + qStr, *kSorted = kDense + qDense;
  uint32_t *vDense = (uint32_t *)(kSorted + qDense);
  if (!noCov) {
    if (!ctx->covFullLen) ctx->covFullLen = std::max(1, maxLen);
    ctx->covFullDirty = true;
  }
  f.fullLen = noCov ? -1 : ctx->covFullLen;
  t1k_launch_fullalign(ctx, f);
  T1K_HIP(ctx, hipEventRecord(ctx->ev[7], ctx->stream));
  if ((rc = fetchCounters(ctx, hc))) return rc;
  if (hc[2]) return capacityError(ctx, hc[2]);
  {
    const T1kArenaCounts ce = t1k_arena_counts(ctx, T1K_AR_EQ, qSegCap), cb = t1k_arena_counts(ctx, T1K_AR_BAND, qSegCap), cw = t1k_arena_counts(ctx, T1K_AR_WIDE, qSegCap);
    if (ce.overflow || cb.overflow || cw.overflow) {
      // A stripe of an alignment queue is full (the records are spread over the stripes by workgroup, a skewed range can put more than
      // twice its share into one).  k_fullalign has already added the coverage of the ungapped alignments: a second pass over the same
      // records takes exactly that back, the queued alignments had not touched the arrays yet, so nothing of the range is committed and
      // it runs again with larger stripes (t1k_assign_range) instead of ending the job with T1K_ERR_COMMITTED.
      if (!noCov) {
        f.undo = 1;
        t1k_launch_fullalign(ctx, f);
        T1K_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->covCommitted = false;
      }
      ctx->queueBoost = std::min<uint32_t>(ctx->queueBoost * 4, 1u << 20);
      if (getenv("T1K_DEBUG_PHASES")) fprintf(stderr, "[t1k] a stripe of an alignment queue is full (%llu records in stripes of %u): %s, the range runs again with stripes x%u\n",
                                              (unsigned long long)nOvl, qSegCap, noCov ? "nothing was committed" : "the range's coverage was taken back", ctx->queueBoost);
      return capacityError(ctx, 64);
    }
    // equal / band queues: dense, then ordered by (read-end, strand, read window, allele-window hash)
    t1k_arena_compact(ctx, T1K_AR_EQ, f.eqStr, qSegCap, vDense, ce.maxSeg);
    t1k_arena_compact64(ctx, T1K_AR_EQ, f.eqKeyStr, qSegCap, kDense, ce.maxSeg);
    if ((rc = t1k_sort_pairs(ctx, kDense, kSorted, vDense, qEq, (uint32_t)ce.total))) return rc;
    t1k_arena_compact(ctx, T1K_AR_BAND, f.bandStr, qSegCap, vDense, cb.maxSeg);
    t1k_arena_compact64(ctx, T1K_AR_BAND, f.bandKeyStr, qSegCap, kDense, cb.maxSeg);
    if ((rc = t1k_sort_pairs(ctx, kDense, kSorted, vDense, qBand, (uint32_t)cb.total))) return rc;
    t1k_arena_compact(ctx, T1K_AR_WIDE, f.wideStr, qSegCap, qWide, cw.maxSeg);
    hc[8] = ce.total; hc[15] = cb.total; hc[20] = cw.total;
  }
  // equal spans (register-band traced DP) and spans differing by 1..4 (wider register band): flags -> runs -> fill -> apply
  for (int kind = 0; kind < 2; ++kind) {
    const uint32_t nJobs = (uint32_t)(kind == 0 ? hc[8] : hc[15]);
    if (!nJobs) continue;
    SlowArgs sl{};
    sl.ref = ctx->ref; sl.reads = rd; sl.relax = relaxFlag; sl.noCov = noCov; sl.ovl = ovl; sl.slowQueue = kind == 0 ? qEq : qBand; sl.nSlow = nJobs;
    sl.perThread = 0; sl.maxCells = 0; sl.counters = counters;
    uint32_t *flags = vDense, *runOf = (uint32_t *)kDense, *rep = (uint32_t *)kSorted;  // the sort's buffers are free again
    t1k_launch_align_flags(ctx, sl, flags);
    if ((rc = t1k_inclusive_sum(ctx, flags, runOf, nJobs))) return rc;
    uint32_t nRuns = 0;
    T1K_HIP(ctx, hipMemcpyAsync(&nRuns, runOf + (nJobs - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    T1K_HIP(ctx, hipStreamSynchronize(ctx->stream));
    t1k_launch_align_reps(ctx, flags, runOf, rep, nJobs);
    const uint64_t stride = ((uint64_t)nRuns + 63) / 64 * 64;
    const size_t traceBytes = stride * (size_t)(maxLen + 2) * 8;
    if (ctx->bLists.bytes >= traceBytes) sl.scratch = (uint8_t *)ctx->bLists.p;
    else if (ctx->bCand.bytes >= traceBytes) sl.scratch = (uint8_t *)ctx->bCand.p;
    else { if ((rc = t1k_ensure(ctx, ctx->bEqTrace, traceBytes))) return rc; sl.scratch = (uint8_t *)ctx->bEqTrace.p; }
    sl.runOf = runOf; sl.rep = rep; sl.nRuns = nRuns; sl.traceStride = stride;
    t1k_launch_align_fill_apply(ctx, sl, kind == 0);
    if (getenv("T1K_DEBUG_PHASES")) fprintf(stderr, "[t1k] %s alignments: %u jobs in %u runs of identical windows\n", kind == 0 ? "equal-span" : "band", nJobs, nRuns);
  }
  if (hc[20]) {  // wide length difference: general DP with row arrays in HBM
    if ((rc = t1k_ensure(ctx, ctx->bSlowScratch, (size_t)slowBlocks * 64 * t1k_slow_per_thread(maxCells)))) return rc;
    SlowArgs sl{};
    sl.ref = ctx->ref; sl.reads = rd; sl.relax = relaxFlag; sl.noCov = noCov; sl.ovl = ovl; sl.slowQueue = qWide; sl.nSlow = (uint32_t)hc[20];
    sl.scratch = (uint8_t *)ctx->bSlowScratch.p; sl.perThread = t1k_slow_per_thread(maxCells); sl.maxCells = maxCells; sl.counters = counters;
    t1k_launch_fullalign_slow(ctx, sl, slowBlocks);
  }
  return T1K_OK;
}

int t1k_assign_batch(t1k_ctx *ctx) {
  if (!ctx) return T1K_ERR_ARG;
  ctx->storeChunk[ctx->storeSlot] = 0; ctx->storeUsed[ctx->storeSlot] = 0;  // every list is recomputed
  return t1k_assign_range(ctx, 0, ctx->reads.nReadEnds);
}

static int assignOnce(t1k_ctx *ctx, uint64_t first, uint32_t count);

// Working capacities.  t1k_params holds the LIMITS of the batch arenas; what is allocated follows the demand: the first range
// starts from a floor per read-end, an overflow (always detected before anything of the range is committed) raises the working
// capacity to the demand the device counted and the range runs again.  Fresh VRAM costs about 35 ms per GB on this platform
// (hipMalloc of 12 GB: 0.3-0.6 s, measured), so a context that allocated its limits up front (40 GB) paid more for memory than
// for kernels on a 1 M-pair job.
int t1k_assign_range(t1k_ctx *ctx, uint64_t first, uint32_t count) {
  if (!ctx || !ctx->ref.bases) return t1k_fail(ctx, T1K_ERR_STATE, "t1k_assign_range: no reference uploaded");
  if (first + count > ctx->reads.nReadEnds) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_assign_range: range outside the uploaded reads");
  T1K_HIP(ctx, hipSetDevice(ctx->device));
  auto clampCap = [](uint64_t want, uint64_t floor, int64_t limit) { return std::min<uint64_t>((uint64_t)limit, std::max<uint64_t>(want, floor)); };
  // floors per read-end: what a 2x150 bp read-end needs against an HLA-sized reference (about 30 000 alleles: 1 900 groups, 1 000 of
  // them on the gap-walk list, 800 candidates, 610 overlaps), with room to spare; smaller references just leave part of it unused
  // (the group arena is striped: its fullest stripe decides, and it runs about 1.3 - 1.45 times the average -- 3 350 per read-end
  // keeps the first ranges of a job from running twice)
  const uint64_t perRe = std::min<uint64_t>(3400, std::max<uint64_t>(256, (uint64_t)ctx->ref.nAlleles * 23 / 200));
  ctx->wGroup = clampCap(std::max<uint64_t>(ctx->wGroup, (uint64_t)count * perRe), 4u << 20, ctx->prm.group_cap);
  ctx->wCand = clampCap(std::max<uint64_t>(ctx->wCand, (uint64_t)count * perRe * 5 / 12), 2u << 20, ctx->prm.cand_cap);
  ctx->wOvl = clampCap(std::max<uint64_t>(ctx->wOvl, (uint64_t)count * perRe / 3), 2u << 20, ctx->prm.ovl_cap);
  ctx->wList = clampCap(std::max<uint64_t>(ctx->wList, (uint64_t)count * perRe * 7 / 12), 1u << 20, ctx->prm.group_cap);
  ctx->wRare = clampCap(std::max<uint64_t>(ctx->wRare, (uint64_t)count * perRe / 24), 1u << 18, ctx->prm.group_cap);
  // alignment job lists (memo slots waiting for k_dp_dense): a read-end's memo has 2048 slots, so count * 2048 is the most there can be.
  // A list that overflows must not be survived by releasing the claim (another lane may already wait on that memo slot): the range
  // runs again with a list that fits.
  const int64_t jobLimit = (int64_t)count * 2048 + (1 << 20);
  ctx->wJob = clampCap(std::max<uint64_t>(ctx->wJob, (uint64_t)count * 512), 1u << 20, std::max<int64_t>(jobLimit, (int64_t)ctx->wJob));
  ctx->wGenJob = clampCap(std::max<uint64_t>(ctx->wGenJob, (uint64_t)count * 128), 1u << 20, std::max<int64_t>(jobLimit, (int64_t)ctx->wGenJob));
  // hit lists of the multi-diagonal groups: 64 M entries (256 MB) serve every 2 x 150 bp range; a window with reads beyond the hit masks sends
  // every group of THOSE read-ends through explicit lists (~1000 hits a group) and grows the arena to what the device counted -- round 3
  // reserved 1 G entries (4 GB per pipeline) as soon as one such read was in the window
  const int64_t genHitLimit = 1024ll << 20;
  ctx->wGenHit = clampCap(ctx->wGenHit, 64u << 20, genHitLimit);
  for (int attempt = 0;; ++attempt) {
    ctx->lastCapFlags = 0; ctx->needGroup = ctx->needCand = ctx->needOvl = ctx->needList = ctx->needRare = ctx->needJob = ctx->needGenJob = ctx->needGenHit = 0;
    const int rc = assignOnce(ctx, first, count);
    if (rc != T1K_ERR_CAPACITY || attempt >= 10) return rc;
    bool grew = false;
    auto grow = [&](uint64_t &w, uint64_t need, int64_t limit) {  // to the demand the device counted, with a quarter to spare
      if (w >= (uint64_t)limit || need <= w) return;
      w = std::min<uint64_t>((uint64_t)limit, need + need / 4);
      grew = true;
    };
    if (ctx->lastCapFlags & 256) {
      const uint64_t before = ctx->wGroup;
      grow(ctx->wGroup, ctx->needGroup, ctx->prm.group_cap);
      if (ctx->wGroup > before && !ctx->scaledOnce) {
        // the seeding kernel is the first to overflow and the later arenas' demand goes with the number of groups: scale them once
        // by the same factor instead of discovering each of them with a failed pass of its own
        const double f = std::min(4.0, (double)ctx->wGroup / (double)before);
        ctx->wList = std::min<uint64_t>((uint64_t)ctx->prm.group_cap, (uint64_t)(ctx->wList * f));
        ctx->wCand = std::min<uint64_t>((uint64_t)ctx->prm.cand_cap, (uint64_t)(ctx->wCand * f));
        ctx->wOvl = std::min<uint64_t>((uint64_t)ctx->prm.ovl_cap, (uint64_t)(ctx->wOvl * f));
        ctx->scaledOnce = true;
      }
      grow(ctx->wList, ctx->needList, ctx->prm.group_cap);
      grow(ctx->wRare, ctx->needRare, ctx->prm.group_cap);
      grow(ctx->wJob, ctx->needJob, jobLimit * 2);
      grow(ctx->wGenJob, ctx->needGenJob, jobLimit * 2);
    }
    if (ctx->lastCapFlags & 2) grow(ctx->wGenHit, ctx->needGenHit, genHitLimit);  // (candidate staging of the multi-diagonal groups: only the hit arena grows)
    if (ctx->lastCapFlags & 4) grow(ctx->wCand, ctx->needCand, ctx->prm.cand_cap);
    if (ctx->lastCapFlags & 16) grow(ctx->wOvl, ctx->needOvl, ctx->prm.ovl_cap);
    if (ctx->lastCapFlags & 64) grew = true;  // an alignment queue's stripe: t1k_fullalign_phase has raised queueBoost (and taken the range's coverage back)
    if (!grew || (ctx->lastCapFlags & ~(256ull | 4ull | 16ull | 64ull | 2ull))) return rc;  // at the limits (or another arena): the caller splits the range
    if (getenv("T1K_DEBUG_PHASES")) fprintf(stderr, "[t1k] range of %u read-ends again with capacities: groups %llu lists %llu / %llu jobs %llu / %llu candidates %llu overlaps %llu\n", count,
                                            (unsigned long long)ctx->wGroup, (unsigned long long)ctx->wList, (unsigned long long)ctx->wRare, (unsigned long long)ctx->wJob, (unsigned long long)ctx->wGenJob,
                                            (unsigned long long)ctx->wCand, (unsigned long long)ctx->wOvl);
  }
}

static int assignOnce(t1k_ctx *ctx, uint64_t first, uint32_t count) {
  const uint32_t n = count;
  ctx->covCommitted = false;
  T1kReadsDev rd = ctx->reads;  // view of the sub-range; read-end ids inside the batch are relative to `first`
  rd.nReadEnds = count;
  rd.bases += first * 2 * rd.S; rd.nmask += first * 2 * rd.S; rd.len += first; rd.weight += first;
  if (rd.skip) rd.skip += first;
  ctx->rangeCount = count;
  int rc;
  const int nWg = (int)std::min<uint32_t>((uint32_t)ctx->prm.workgroups, std::max<uint32_t>(n, 1));
  const uint32_t sortCap = 1u << 15;
  const int maxChunks = t1k_chain_max_chunks(ctx->ref.nAlleles), memoN = t1k_chain_memo_entries();
  const bool longReads = ctx->batchFastMaxLen > 160;  // (read-ends beyond T1K_MAX_READ_LEN are k_seed_long's: they do not widen the masks)
  const int recStride = t1k_chain_rec_stride(ctx->batchFastMaxLen);
  const uint64_t groupCap = ctx->wGroup;
  // (a window with reads beyond T1K_MAX_READ_LEN sends every group of those read-ends through the explicit hit lists: up to ~1000 hits a group)
  const uint32_t jobCap = (uint32_t)ctx->wJob, genCandCap = 16u << 20, genHitCap = (uint32_t)std::min<uint64_t>(ctx->wGenHit ? ctx->wGenHit : (64u << 20), 1024u << 20), genJobCap = (uint32_t)ctx->wGenJob;
  const int bigBlocks = 32;
  if ((rc = t1k_ensure(ctx, ctx->bCounters, (size_t)T1K_COUNTER_WORDS * 8))) return rc;
  if ((rc = t1k_ensure(ctx, ctx->bWgGroups, groupCap * recStride * 4))) return rc;                        // batch group records
  if ((rc = t1k_ensure(ctx, ctx->bWgStage, (size_t)n * maxChunks * 8 + 64))) return rc;                   // chunkStart | chunkCount
  const int maxK = t1k_chain_max_kmers(ctx->batchMaxLen, ctx->prm.kmer_length);
  if ((rc = t1k_ensure(ctx, ctx->bWgHits, (size_t)n * (t1k_chain_used_u32(maxK) + 2 + 2 * T1K_USED_MASK_WORDS) * 4 + 64))) return rc;  // used k-mer lists | counts | used-offset masks
  if ((rc = t1k_ensure(ctx, ctx->bWgCache, (size_t)n * memoN * 8 + 64))) return rc;                       // per-read-end memo
  if ((rc = t1k_ensure(ctx, ctx->bWgBig, (size_t)bigBlocks * 64 * t1k_chain_big_scratch_u32() * 4))) return rc;
  // work lists: every list exists twice, as a striped arena the kernels append to and as the dense list its consumer reads
  const uint32_t groupSegCap = (uint32_t)std::min<uint64_t>(groupCap / T1K_NSTRIPE, 0xFFFFFFFFull / T1K_NSTRIPE);
  // group-id lists: the frequent kinds (gap walk, retry, finish; extension retry) and the rare ones (several diagonals, wave, big scratch)
  const uint32_t listSegCap = (uint32_t)std::max<uint64_t>(ctx->wList / T1K_NSTRIPE, 1024u), rareSegCap = (uint32_t)std::max<uint64_t>(ctx->wRare / T1K_NSTRIPE, 1024u);
  const uint32_t jobSegCap = jobCap / T1K_NSTRIPE, genCandSegCap = genCandCap / T1K_NSTRIPE;
  const size_t listWords = (size_t)listSegCap * T1K_NSTRIPE, rareWords = (size_t)rareSegCap * T1K_NSTRIPE;
  if ((rc = t1k_ensure(ctx, ctx->bLists, ((size_t)jobCap * 2 + listWords * 6 + rareWords * 6 + (size_t)genCandCap * 6 + genHitCap + (size_t)genJobCap * 2) * 4 + 64))) return rc;
  if ((rc = t1k_ensure(ctx, ctx->bCand, (size_t)ctx->wCand * sizeof(T1kCand)))) return rc;
  if ((rc = t1k_ensure(ctx, ctx->bExt, (size_t)ctx->wCand * sizeof(T1kExt)))) return rc;
  // this range's lists go to the end of the overlap store: the current chunk if the working overlap capacity still fits, else the next one
  const uint64_t chunkEntries = std::max<uint64_t>(ctx->wOvl, (uint64_t)std::max(1, ctx->prm.store_chunk_mb) * ((1ull << 20) / sizeof(T1kOvlP)));
  {
    const int sl = ctx->storeSlot;
    std::vector<T1kDevBuf> &chunks = ctx->storeChunks[sl];
    for (;;) {
      if (ctx->storeChunk[sl] >= chunks.size()) chunks.resize(ctx->storeChunk[sl] + 1);
      T1kDevBuf &ch = chunks[ctx->storeChunk[sl]];
      if (ch.p && ctx->storeUsed[sl] + ctx->wOvl <= ch.bytes / sizeof(T1kOvlP)) break;   // fits behind what the chunk already holds
      if (ch.p && ctx->storeUsed[sl] > 0) { ++ctx->storeChunk[sl]; ctx->storeUsed[sl] = 0; continue; }
      if (ch.p) { (void)t1k_dev_free(ch.p); ch.p = nullptr; ch.bytes = 0; }  // an empty chunk that is too small for this range
      const auto t0 = std::chrono::steady_clock::now();
      hipError_t e = t1k_dev_malloc(&ch.p, chunkEntries * sizeof(T1kOvlP));
      ctx->msAlloc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); ctx->bytesAlloc += chunkEntries * sizeof(T1kOvlP);
      if (e != hipSuccess) { ch.p = nullptr; return t1k_fail(ctx, T1K_ERR_DEVICE, std::string("overlap store: hipMalloc of another ") + std::to_string(chunkEntries * sizeof(T1kOvlP) >> 20) + " MB chunk failed (" + hipGetErrorString(e) + "); fewer fragments per window (T1K_WINDOW) need less"); }
      ch.bytes = chunkEntries * sizeof(T1kOvlP);
    }
    ctx->storeBase = (T1kOvlP *)chunks[ctx->storeChunk[sl]].p + ctx->storeUsed[sl];
  }
  if ((rc = t1k_ensure(ctx, ctx->bOvlWork, (size_t)ctx->wOvl * sizeof(T1kOvl)))) return rc;
  ctx->ovlBase = (T1kOvl *)ctx->bOvlWork.p;
  // (n + 1 entries each: the two counts are cleared one entry past the range.  Since round 1 the blocks were asked for n entries and cleared for n + 1 --
  // inside the block's slack of an eighth + 256 bytes except when a LATER range of the context had exactly that many more read-ends than its first:
  // 29 read-ends first, 96 later -> 386 bytes held, 388 cleared, "hipMemsetAsync: invalid argument".  Found by the fuzz file under 96-read-end ranges, round 6)
  if ((rc = t1k_ensure(ctx, ctx->bCandStart, (size_t)(n + 1) * 4))) return rc;
  if ((rc = t1k_ensure(ctx, ctx->bCandCount, (size_t)(n + 1) * 4))) return rc;
  if ((rc = t1k_ensure(ctx, ctx->bOvlStart, (size_t)(n + 1) * 4))) return rc;
  if ((rc = t1k_ensure(ctx, ctx->bOvlCount, (size_t)(n + 1) * 4))) return rc;
  if ((rc = t1k_ensure(ctx, ctx->bSortScratch, (size_t)std::min(nWg, 512) * sortCap * 48))) return rc;  // only the kernels launched with <= 512 workgroups use it
  T1K_HIP(ctx, hipMemsetAsync(ctx->bCounters.p, 0, (size_t)T1K_COUNTER_WORDS * 8, ctx->stream));
  T1K_HIP(ctx, hipMemsetAsync(ctx->bOvlCount.p, 0, (size_t)n * 4 + 4, ctx->stream));
  T1K_HIP(ctx, hipMemsetAsync(ctx->bCandCount.p, 0, (size_t)n * 4 + 4, ctx->stream));
  T1K_HIP(ctx, hipMemsetAsync(ctx->bWgCache.p, 0, (size_t)n * memoN * 8, ctx->stream));
  ctx->nCand = ctx->nOvl = 0;
  ctx->traceFetch = 0;
  if (getenv("T1K_DEBUG_TRACE")) fprintf(stderr, "[t1k trace] assign_range first %llu count %u\n", (unsigned long long)first, n);
  memset(&ctx->stats, 0, sizeof(ctx->stats));
  if (n == 0) return T1K_OK;
  unsigned long long hc[64];
  double t0 = nowMs();
  ChainArgs a{};
  a.ref = ctx->ref; a.reads = rd;
  a.k = ctx->prm.kmer_length; a.radius = ctx->prm.radius; a.hitLenRequired = ctx->prm.hit_len_required;
  a.sim = ctx->prm.ref_seq_similarity;
  const int relaxFlag = ctx->prm.relax_intron_align;
  a.recs = (uint32_t *)ctx->bWgGroups.p; a.recStride = (uint32_t)recStride; a.groupCap = groupCap;
  a.chunkStart = (uint32_t *)ctx->bWgStage.p; a.chunkCount = a.chunkStart + (size_t)n * maxChunks; a.maxChunks = maxChunks;
  a.usedOut = (uint32_t *)ctx->bWgHits.p; a.usedCount = a.usedOut + (size_t)n * t1k_chain_used_u32(maxK); a.usedMask = a.usedCount + (size_t)n * 2; a.maxK = (uint32_t)maxK;
  a.maxKFast = (uint32_t)t1k_chain_max_kmers(ctx->batchFastMaxLen, ctx->prm.kmer_length);
  a.memo = (unsigned long long *)ctx->bWgCache.p;
  a.jobList = (uint32_t *)ctx->bLists.p; a.jobCap = jobCap;
  a.jobStr = a.jobList + jobCap;
  a.retryList = a.jobStr + jobCap; a.finishList = a.retryList + listWords; a.slowList = a.finishList + listWords;
  a.retryStr = a.slowList + listWords; a.finishStr = a.retryStr + listWords; a.slowStr = a.finishStr + listWords;
  a.generalList = a.slowStr + listWords; a.bigList = a.generalList + rareWords; a.waveList = a.bigList + rareWords;
  a.generalStr = a.waveList + rareWords; a.bigStr = a.generalStr + rareWords; a.waveStr = a.bigStr + rareWords;
  a.genCand = a.waveStr + rareWords; a.genCandCap = genCandCap;
  a.genHits = a.genCand + (size_t)genCandCap * 6; a.genHitSegCap = genHitCap / T1K_NSTRIPE;
  a.genJobStr = a.genHits + genHitCap; a.genJobList = a.genJobStr + genJobCap; a.genJobSegCap = genJobCap / T1K_NSTRIPE;
  a.groupSegCap = groupSegCap; a.jobSegCap = jobSegCap; a.listSegCap = listSegCap; a.rareSegCap = rareSegCap; a.genCandSegCap = genCandSegCap;
  a.bigScratch = (uint32_t *)ctx->bWgBig.p;
  { static const int ns = getenv("T1K_NO_SIMPLE_CHAIN") ? 0 : 1; a.nearSimple = ns; }
  { static const int fz = getenv("T1K_FUSE_SEED") ? atoi(getenv("T1K_FUSE_SEED")) : 0; a.fuse = fz; }  // measured in round 5: 20 % MORE kernel time than the two launches; opt-in
  { static const int ep = getenv("T1K_NO_EARLY_PRUNE") ? 0 : getenv("T1K_WALK_IN_CLOSED") ? atoi(getenv("T1K_WALK_IN_CLOSED")) + 1 : 2; a.earlyPrune = ep; }  // 0: none, 1: the closed-form pass prunes with the gap-count bound, 2: ... and runs the gap walk's first pass
  a.cand = (T1kCand *)ctx->bCand.p; a.candCap = ctx->wCand;
  a.candStart = (uint32_t *)ctx->bCandStart.p; a.candCount = (uint32_t *)ctx->bCandCount.p;
  a.counters = (unsigned long long *)ctx->bCounters.p;
  if ((rc = t1k_run_chain(ctx, a, nWg, bigBlocks, longReads, hc))) return rc;
  double t1 = nowMs();
  if (hc[2]) return capacityError(ctx, hc[2]);
  ctx->nCand = hc[0];
  ExtendArgs e{};
  e.ref = ctx->ref; e.reads = rd; e.k = a.k; e.sim = a.sim;
  e.cand = a.cand; e.ext = (T1kExt *)ctx->bExt.p; e.nCand = ctx->nCand; e.counters = a.counters;
  // the extension alignments go through the memo like the chain's (the chain's work lists are dead by now: reuse their memory)
  e.memo = a.memo; e.jobStr = a.jobStr; e.jobSegCap = a.jobSegCap; e.retryStr = a.retryStr; e.retrySegCap = a.listSegCap;
  t1k_launch_extend(ctx, e);
  const bool hostDriven = t1k_chain_host_driven();
  if (hostDriven) {
    if ((rc = fetchCounters(ctx, hc))) return rc;
    const T1kArenaCounts ej = t1k_arena_counts(ctx, T1K_AR_EXTJOBS, e.jobSegCap), er = t1k_arena_counts(ctx, T1K_AR_EXTRETRY, e.retrySegCap);
    if (er.overflow || ej.overflow) return capacityError(ctx, 256);
    t1k_arena_compact(ctx, T1K_AR_EXTJOBS, e.jobStr, e.jobSegCap, a.jobList, ej.maxSeg);
    t1k_arena_compact(ctx, T1K_AR_EXTRETRY, e.retryStr, e.retrySegCap, a.retryList, er.maxSeg);
    t1k_launch_dp_dense(ctx, a, a.jobList, (uint32_t)ej.total);
    t1k_launch_extend_retry(ctx, e, a.retryList, (uint32_t)er.total);
  } else if (ctx->nCand) {
    // the same launches without the counter fetch: item counts read on the device, grids from the previous range's counts; a full stripe
    // raises the group-capacity flag on the device and shows in the fetch behind the selection (nothing of the range is committed before it)
    const uint64_t jobCapAll = (uint64_t)e.jobSegCap * T1K_NSTRIPE, retryCapAll = (uint64_t)e.retrySegCap * T1K_NSTRIPE;
    const uint64_t eJ = t1k_arena_estimate(ctx, T1K_AR_EXTJOBS, jobCapAll, n), eR = t1k_arena_estimate(ctx, T1K_AR_EXTRETRY, retryCapAll, n);
    t1k_arena_compact_dev(ctx, T1K_AR_EXTJOBS, e.jobStr, e.jobSegCap, a.jobList, eJ);
    t1k_arena_compact_dev(ctx, T1K_AR_EXTRETRY, e.retryStr, e.retrySegCap, a.retryList, eR);
    t1k_launch_dp_dense_dev(ctx, a, a.jobList, T1K_AR_EXTJOBS, (uint32_t)jobCapAll, eJ);
    t1k_launch_extend_retry_dev(ctx, e, a.retryList, T1K_AR_EXTRETRY, eR);
  }
  T1K_HIP(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
  if (hostDriven) T1K_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double t2 = nowMs();
  SelectArgs s{};
  s.reads = rd; s.cand = a.cand; s.ext = e.ext; s.candStart = a.candStart; s.candCount = a.candCount;
  s.ovl = ctx->ovlBase; s.ovlCap = ctx->wOvl;
  s.ovlStart = (uint32_t *)ctx->bOvlStart.p; s.ovlCount = (uint32_t *)ctx->bOvlCount.p;
  s.sortScratch = (uint64_t *)ctx->bSortScratch.p; s.sortCap = sortCap; s.counters = a.counters;
  s.alleleBits = 1; s.relax = relaxFlag; s.xl = ctx->batchMaxLen > T1K_MAX_READ_LEN ? 1 : 0;
  while ((1u << s.alleleBits) < ctx->ref.nAlleles) ++s.alleleBits;
  t1k_launch_select(ctx, s, nWg);
  T1K_HIP(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
  if ((rc = fetchCounters(ctx, hc))) return rc;
  double t3 = nowMs();
  if (hc[2]) return capacityError(ctx, hc[2]);
  if (!hostDriven) {
    t1k_arena_estimate_set(ctx, T1K_AR_EXTJOBS, t1k_arena_counts(ctx, T1K_AR_EXTJOBS, e.jobSegCap).total, n);
    t1k_arena_estimate_set(ctx, T1K_AR_EXTRETRY, t1k_arena_counts(ctx, T1K_AR_EXTRETRY, e.retrySegCap).total, n);
  }
  ctx->nOvl = hc[1];
  // Near-best full alignments (SeqSet.hpp:2188-2285).  Without --relaxIntronAlign they only feed the per-base coverage, so a context
  // whose coverage is deferred (t1k_ctx_set_coverage_mode) skips them here altogether: k_select has written the relaxed counts, and
  // t1k_coverage_selected later aligns the records of the alleles whose coverage is actually read.  With it they run for the relaxed
  // counts, and the deferred mode only leaves the coverage updates out.
  const bool deferCov = ctx->covMode == 1;
  if (relaxFlag || !deferCov) {
    if (!deferCov) ctx->covCommitted = true;  // from here on the range's coverage is in the context's arrays
    if ((rc = t1k_fullalign_phase(ctx, rd, s.ovl, ctx->nOvl, relaxFlag, deferCov ? 1 : 0, ctx->batchMaxLen, false, hc))) return rc;
  } else {
    T1K_HIP(ctx, hipEventRecord(ctx->ev[7], ctx->stream));
    hc[8] = hc[15] = hc[20] = 0;
  }
  hipEvent_t evSlow = ctx->ev[9];
  T1K_HIP(ctx, hipEventRecord(evSlow, ctx->stream));
  TruncArgs tr{};
  tr.reads = rd; tr.ovl = s.ovl; tr.ovlStart = s.ovlStart; tr.ovlCount = s.ovlCount; tr.sortScratch = s.sortScratch; tr.sortCap = sortCap; tr.alleleBits = s.alleleBits;
  tr.counters = a.counters; tr.xl = s.xl;
  t1k_launch_truncate(ctx, tr, nWg);
  T1K_HIP(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
  // the lists are final: publish them in the read set's table (absolute read-end index) and keep their records in the store
  t1k_launch_publish_lists(ctx, rd.listPtr + first, rd.listCount + first, ctx->ovlBase, ctx->storeBase, ctx->nOvl, s.ovlStart, s.ovlCount, n, a.counters, rd.skip);
  if ((rc = fetchCounters(ctx, hc))) return rc;
  double t4 = nowMs();
  if (hc[2]) return capacityError(ctx, hc[2]);
  ctx->storeUsed[ctx->storeSlot] += ctx->nOvl;
  if (getenv("T1K_DEBUG_PHASES")) {
    float a1 = 0, a2 = 0, a3 = 0;
    (void)hipEventElapsedTime(&a1, ctx->ev[3], ctx->ev[7]); (void)hipEventElapsedTime(&a2, ctx->ev[7], evSlow); (void)hipEventElapsedTime(&a3, evSlow, ctx->ev[4]);
    fprintf(stderr, "[t1k] fullalign %.2f ms, eq-DP (%llu jobs) + general-DP (%llu jobs) %.2f ms, truncate %.2f ms; cand %llu ovl %llu dp %llu; groups %llu (gap walk %llu) fast %llu general %llu big %llu; memo jobs %llu parked %llu wide-queue %llu\n",
            a1, hc[8], hc[15], a2, a3, hc[0], hc[1], hc[7], hc[6], (unsigned long long)ctx->lastSlowGroups, hc[11], hc[12], hc[13], hc[16], hc[17], hc[20]);
#ifdef T1K_NEAR_STATS
    fprintf(stderr, "[t1k] multi-diagonal groups: far strays %llu, near count saturated %llu, other fallback %llu, masks ok but not a chain (<= 32 hits) %llu, (> 32 hits) %llu, chain (<= 96) %llu, chain (> 96) %llu\n", hc[40], hc[41], hc[42], hc[43], hc[47], hc[45], hc[46]);
#endif
#ifdef T1K_WALK_STATS
    fprintf(stderr, "[t1k] gap walk classes: exact without a DP %llu, pruned by the per-gap bound %llu, memo / DP %llu\n", hc[40], hc[41], hc[42]);
#endif
  }
  t1k_stats &st = ctx->stats;
  st.read_ends = n; st.lookups = hc[3]; st.postings = hc[4]; st.hits = hc[5]; st.groups = hc[6]; st.candidates = hc[0]; st.extended = hc[1];
  st.dp_calls = hc[7] + hc[14]; st.near_best = hc[10]; st.dp_cells = hc[24];
  float ms[4] = {0, 0, 0, 0}, msSeed = 0;  // kernel durations from HIP events on the launch stream
  for (int i = 0; i < 4; ++i) (void)hipEventElapsedTime(&ms[i], ctx->ev[i], ctx->ev[i + 1]);
  (void)hipEventElapsedTime(&msSeed, ctx->ev[0], ctx->ev[8]);
  (void)t1; (void)t2; (void)t3;
  st.ms_seed = msSeed; st.ms_chain = ms[0] - msSeed; st.ms_extend = ms[1]; st.ms_select = ms[2]; st.ms_fullalign = ms[3]; st.ms_total = t4 - t0;
  st.batches = 1;
  return T1K_OK;
}

int t1k_overlaps_download(t1k_ctx *ctx, uint32_t *counts, t1k_overlap *out, uint64_t cap, uint64_t *total) {
  if (!ctx) return T1K_ERR_ARG;
  T1K_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t n = ctx->rangeCount;
  std::vector<uint32_t> start(n), cnt(n);
  if (n) {
    T1K_HIP(ctx, hipMemcpy(start.data(), ctx->bOvlStart.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    T1K_HIP(ctx, hipMemcpy(cnt.data(), ctx->bOvlCount.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  }
  uint64_t tot = 0;
  for (uint32_t i = 0; i < n; ++i) tot += cnt[i];
  if (total) *total = tot;
  if (counts) memcpy(counts, cnt.data(), (size_t)n * 4);
  if (!out) return T1K_OK;
  if (cap < tot) return t1k_fail(ctx, T1K_ERR_ARG, "overlap buffer too small");
  std::vector<T1kOvl> h(ctx->nOvl);
  if (ctx->nOvl) T1K_HIP(ctx, hipMemcpy(h.data(), ctx->ovlBase, ctx->nOvl * sizeof(T1kOvl), hipMemcpyDeviceToHost));
  uint64_t w = 0;
  for (uint32_t i = 0; i < n; ++i) {
    for (uint32_t j = 0; j < cnt[i]; ++j) {
      const T1kOvl &o = h[(uint64_t)start[i] + j];
      t1k_overlap &r = out[w++];
      r.seq_idx = (int32_t)o.allele; r.read_start = o.readStart; r.read_end = o.readEnd; r.seq_start = o.seqStart; r.seq_end = o.seqEnd;
      r.strand = (o.flags & 2) ? -1 : 1;
      r.match_cnt = o.matchCnt; r.left_clip = o.leftClip; r.right_clip = o.rightClip; r.relaxed_match_cnt = o.relaxed;
      r.similarity = (double)o.matchCnt / (double)(o.readEnd - o.readStart + 1 + o.seqEnd - o.seqStart + 1 + 2 * o.leftClip + 2 * o.rightClip);
    }
  }
  return T1K_OK;
}

double t1k_alloc_ms(t1k_ctx *ctx, uint64_t *bytes) {  // time this context has spent in hipMalloc, and how much it asked for
  if (!ctx) return 0;
  if (bytes) *bytes = ctx->bytesAlloc;
  return ctx->msAlloc;
}

int t1k_stats_get(t1k_ctx *ctx, t1k_stats *out) {
  if (!ctx || !out) return T1K_ERR_ARG;
  *out = ctx->stats;
  return T1K_OK;
}

}  // extern "C"
