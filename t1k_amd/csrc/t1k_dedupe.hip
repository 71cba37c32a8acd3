// t1k_amd/csrc/t1k_dedupe.hip -- identical read-ends collapse onto one representative with the run length as its weight.
//
// Reference: Genotyper.cpp:451-480 sorts all read-ends (strcmp order) and calls SeqSet::AssignRead once per distinct sequence
// with weight = multiplicity; the per-base coverage is the only thing the weight feeds (SeqSet.hpp:2253-2274).  Here the
// packed read-ends of the uploaded batch are hashed (64 bits over length + forward-strand words + N mask), ordered by a stable
// radix sort of (hash, index), and a run of equal hashes is split wherever two neighbours differ in content, so that a hash
// collision can only cost a missed merge, never a wrong one.  The representative of a run is its first read-end in upload order, and
// the distinct read-ends are numbered in the upload order of their representatives: the first m fragments of a window use exactly
// the distinct read-ends [0, D_m), so a pairing range can start as soon as a prefix of the assignment ranges is done (host/job.cpp)
// -- (T1K_DISTINCT_ORDER=hash numbers them in hash order, as rounds 1-2 did).
// Everything is integer / HBM-bound (a few hundred MB per window of reads); no MFMA.
#include <cstring>
#include "t1k_dev.h"
#include "t1k_launch.h"

__device__ __forceinline__ unsigned long long mix64(unsigned long long h, unsigned long long v) {
  h ^= v;
  h *= 0x9E3779B97F4A7C15ull;
  h ^= h >> 32;
  h *= 0xD6E8FEB86659FD93ull;
  h ^= h >> 29;
  return h;
}

__global__ void k_dedupe_hash(T1kReadsDev R, unsigned long long *keys, uint32_t *idx) {
  const uint32_t re = blockIdx.x * blockDim.x + threadIdx.x;
  if (re >= R.nReadEnds) return;
  const int S = R.S;
  const uint64_t *b = R.bases + (uint64_t)re * 2 * S, *m = R.nmask + (uint64_t)re * 2 * S;
  unsigned long long h = mix64(0x243F6A8885A308D3ull, R.len[re]);
  for (int w = 0; w < S; ++w) { h = mix64(h, b[w]); h = mix64(h, m[w] + 0x9E3779B97F4A7C15ull * (unsigned)(w + 1)); }
  keys[re] = h;
  idx[re] = re;
}

// flag[j] = 1 where sorted position j starts a new distinct sequence
__global__ void k_dedupe_mark(T1kReadsDev R, const unsigned long long *keys, const uint32_t *idx, uint32_t *flag) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= R.nReadEnds) return;
  uint32_t f = 1;
  if (j > 0 && keys[j] == keys[j - 1]) {
    const uint32_t a = idx[j], p = idx[j - 1];
    const int S = R.S;
    bool same = R.len[a] == R.len[p];
    const uint64_t *ba = R.bases + (uint64_t)a * 2 * S, *bp = R.bases + (uint64_t)p * 2 * S;
    const uint64_t *ma = R.nmask + (uint64_t)a * 2 * S, *mp = R.nmask + (uint64_t)p * 2 * S;
    for (int w = 0; w < S && same; ++w) same = ba[w] == bp[w] && ma[w] == mp[w];
    f = same ? 0 : 1;
  }
  flag[j] = f;
}

// isRep[re] = 1 for the representative of a run (idx is a permutation: every entry is written); runRep[run] = that read-end
__global__ void k_dedupe_reps(uint32_t n, const uint32_t *idx, const uint32_t *flag, const uint32_t *runOf, uint32_t *isRep, uint32_t *runRep) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  isRep[idx[j]] = flag[j];
  if (flag[j]) runRep[runOf[j] - 1] = idx[j];
}
// runOf = inclusive scan of flag (1-based run of each sorted position); repPos (or NULL) = inclusive scan of isRep (1-based number of a
// representative in upload order)
__global__ void k_dedupe_scatter(uint32_t n, const uint32_t *idx, const uint32_t *flag, const uint32_t *runOf, const uint32_t *runRep, const uint32_t *repPos,
                                 const uint32_t *wIn, uint32_t *distinctOf, uint32_t *repr, uint32_t *wOut) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint32_t run = runOf[j] - 1, re = idx[j];
  const uint32_t d = repPos ? repPos[runRep[run]] - 1 : run;
  distinctOf[re] = d;
  if (flag[j]) repr[d] = re;
  atomicAdd(&wOut[d], wIn[re]);
}

__global__ void k_dedupe_gather(T1kReadsDev R, uint32_t nDistinct, const uint32_t *repr, uint64_t *bases, uint64_t *nmask, uint16_t *len) {
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int W = 2 * R.S;
  if (gid >= (uint64_t)nDistinct * W) return;
  const uint32_t d = (uint32_t)(gid / W);
  const int w = (int)(gid % W);
  const uint32_t re = repr[d];
  bases[gid] = R.bases[(uint64_t)re * W + w];
  nmask[gid] = R.nmask[(uint64_t)re * W + w];
  if (w == 0) len[d] = R.len[re];
}

extern "C" int t1k_reads_dedupe(t1k_ctx *ctx, uint32_t *distinctOf, uint32_t *nDistinct) {
  if (!ctx || !nDistinct) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_reads_dedupe: bad arguments");
  if (ctx->readsShared) return t1k_fail(ctx, T1K_ERR_STATE, "t1k_reads_dedupe: the context aliases another context's reads");
  T1K_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t n = ctx->reads.nReadEnds;
  *nDistinct = 0;
  if (n == 0) return T1K_OK;
  if (!distinctOf) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_reads_dedupe: distinctOf is NULL");
  const int S = ctx->reads.S;
  int rc;
  // scratch: keys | sorted keys | idx | sorted idx | flag | runOf | repr | distinctOf | isRep | repPos | runRep
  const size_t n4 = ((size_t)n * 4 + 255) & ~(size_t)255, n8 = ((size_t)n * 8 + 255) & ~(size_t)255;
  if ((rc = t1k_ensure(ctx, ctx->bDedupScratch, 2 * n8 + 9 * n4))) return rc;
  char *sp = (char *)ctx->bDedupScratch.p;
  unsigned long long *keys = (unsigned long long *)sp, *keysSorted = (unsigned long long *)(sp + n8);
  uint32_t *idx = (uint32_t *)(sp + 2 * n8), *idxSorted = idx + n4 / 4, *flag = idxSorted + n4 / 4, *runOf = flag + n4 / 4, *repr = runOf + n4 / 4,
           *dDistinctOf = repr + n4 / 4, *isRep = dDistinctOf + n4 / 4, *repPos = isRep + n4 / 4, *runRep = repPos + n4 / 4;
  static const bool hashOrder = [] { const char *e = getenv("T1K_DISTINCT_ORDER"); return e && !strcmp(e, "hash"); }();
  const unsigned nb = (n + 255) / 256;
  hipLaunchKernelGGL(k_dedupe_hash, dim3(nb), dim3(256), 0, ctx->stream, ctx->reads, keys, idx);
  if ((rc = t1k_sort_pairs(ctx, keys, keysSorted, idx, idxSorted, n))) return rc;
  hipLaunchKernelGGL(k_dedupe_mark, dim3(nb), dim3(256), 0, ctx->stream, ctx->reads, keysSorted, idxSorted, flag);
  if ((rc = t1k_inclusive_sum(ctx, flag, runOf, n))) return rc;
  uint32_t D = 0;
  T1K_HIP(ctx, hipMemcpyAsync(&D, runOf + (n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
  T1K_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // the distinct set lives in its own buffers; the uploaded (full) set stays where it is until the next upload overwrites it
  if ((rc = t1k_ensure(ctx, ctx->bDedupBases, (size_t)D * 2 * S * 8 + 64))) return rc;
  if ((rc = t1k_ensure(ctx, ctx->bDedupN, (size_t)D * 2 * S * 8 + 64))) return rc;
  if ((rc = t1k_ensure(ctx, ctx->bDedupLen, (size_t)D * 2 + 16))) return rc;
  if ((rc = t1k_ensure(ctx, ctx->bDedupWeight, (size_t)D * 4 + 16))) return rc;
  T1K_HIP(ctx, hipMemsetAsync(ctx->bDedupWeight.p, 0, (size_t)D * 4, ctx->stream));
  if (!hashOrder) {
    hipLaunchKernelGGL(k_dedupe_reps, dim3(nb), dim3(256), 0, ctx->stream, n, idxSorted, flag, runOf, isRep, runRep);
    if ((rc = t1k_inclusive_sum(ctx, isRep, repPos, n))) return rc;
  }
  hipLaunchKernelGGL(k_dedupe_scatter, dim3(nb), dim3(256), 0, ctx->stream, n, idxSorted, flag, runOf, runRep, hashOrder ? (const uint32_t *)nullptr : repPos, ctx->reads.weight,
                     dDistinctOf, repr, (uint32_t *)ctx->bDedupWeight.p);
  const uint64_t words = (uint64_t)D * 2 * S;
  hipLaunchKernelGGL(k_dedupe_gather, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, ctx->stream, ctx->reads, D, repr, (uint64_t *)ctx->bDedupBases.p,
                     (uint64_t *)ctx->bDedupN.p, (uint16_t *)ctx->bDedupLen.p);
  T1K_HIP(ctx, hipMemcpyAsync(distinctOf, dDistinctOf, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  T1K_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->reads.nReadEnds = D;
  ctx->reads.bases = (const uint64_t *)ctx->bDedupBases.p;
  ctx->reads.nmask = (const uint64_t *)ctx->bDedupN.p;
  ctx->reads.len = (const uint16_t *)ctx->bDedupLen.p;
  ctx->reads.weight = (const uint32_t *)ctx->bDedupWeight.p;
  ctx->reads.skip = nullptr;
  ctx->rangeCount = 0;
  *nDistinct = D;
  return T1K_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Identical read-ends ACROSS the windows of a job.
//
// t1k_reads_dedupe collapses a window; error-free reads of a locus still come back in every window (10 M benchmark pairs in three
// windows: 5.28 M distinct read-ends, 4.94 M if they were one window).  The overlap list of a read-end is a function of its
// sequence, and the lists of a window whose coverage is deferred stay resident for the whole job (t1k_readset): so a later window
// looks its distinct read-ends up in a table of the sequences those windows assigned (open addressing over a 64-bit hash of length +
// bases + N mask, the sequence itself verified word by word against the earlier window's packed copy: a collision costs a probe,
// never a wrong match), marks the ones it finds (T1kReadsDev::skip: t1k_assign_range does not seed them and does not publish a
// list for them) and, once the earlier windows' assignment ranges are done, copies their table entries (t1k_xwin_resolve):
// pairing then reads the earlier window's records.  The deferred coverage pass scans each window's table with the window's own
// multiplicities, so a linked read-end's records are counted once per window, with that window's weight -- as if it had been
// assigned there.
// ------------------------------------------------------------------------------------------------------------------
struct XSrc {  // what a kept window exposes to later ones
  const uint64_t *bases, *nmask;
  const uint16_t *len;
  unsigned long long *listPtr;
  uint32_t *listCount;
  int S, pad;
};
struct t1k_xwin {
  int device = 0;
  uint64_t cap = 0;  // slots (a power of two)
  T1kDevBuf bKeys, bVals, bSrc, bCount;
  uint32_t maxWindows = 0;
  struct Per { T1kDevBuf ext, skip; uint32_t n = 0; unsigned long long *listPtr = nullptr; uint32_t *listCount = nullptr; };
  std::vector<Per> per;
  std::string err;
};

__device__ __forceinline__ unsigned long long xwinHash(const T1kReadsDev &R, uint32_t re) {
  const int S = R.S, len = R.len[re], nw = (len + 31) >> 5;
  const uint64_t *b = R.bases + (uint64_t)re * 2 * S, *m = R.nmask + (uint64_t)re * 2 * S;
  unsigned long long h = mix64(0x13198A2E03707344ull, (unsigned long long)len);
  for (int w = 0; w < nw; ++w) { h = mix64(h, b[w]); h = mix64(h, m[w] + 0x9E3779B97F4A7C15ull * (unsigned)(w + 1)); }  // (words of the forward strand: S does not enter)
  return h | 1ull;  // 0 = empty slot
}
__device__ __forceinline__ bool xwinSame(const T1kReadsDev &R, uint32_t re, const XSrc &s, uint32_t idx) {
  const int len = R.len[re];
  if ((int)s.len[idx] != len) return false;
  const int nw = (len + 31) >> 5;
  const uint64_t *b = R.bases + (uint64_t)re * 2 * R.S, *m = R.nmask + (uint64_t)re * 2 * R.S;
  const uint64_t *sb = s.bases + (uint64_t)idx * 2 * s.S, *sm = s.nmask + (uint64_t)idx * 2 * s.S;
  for (int w = 0; w < nw; ++w)
    if (b[w] != sb[w] || m[w] != sm[w]) return false;
  return true;
}
// ext[d] = 0, or 1 + (window << 32 | index) of the kept read-end with the same sequence; skip[d] = ext[d] != 0
__global__ void k_xwin_lookup(T1kReadsDev R, const unsigned long long *keys, const unsigned long long *vals, uint64_t mask, const XSrc *src, unsigned long long *ext,
                              uint8_t *skip, uint32_t *found) {
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= R.nReadEnds) return;
  const unsigned long long h = xwinHash(R, d);
  unsigned long long hit = 0;
  uint64_t slot = (h >> 1) & mask;
  for (uint64_t probe = 0; probe <= mask; ++probe, slot = (slot + 1) & mask) {
    const unsigned long long k = keys[slot];
    if (k == 0) break;
    if (k != h) continue;
    const unsigned long long v = vals[slot];
    if (xwinSame(R, d, src[v >> 32], (uint32_t)v)) { hit = v + 1; break; }
  }
  ext[d] = hit;
  skip[d] = hit ? 1 : 0;
  if (hit) atomicAdd(found, 1u);
}
__global__ void k_xwin_insert(T1kReadsDev R, unsigned long long *keys, unsigned long long *vals, uint64_t mask, uint32_t window, const uint8_t *skip) {
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= R.nReadEnds || skip[d]) return;
  const unsigned long long h = xwinHash(R, d);
  uint64_t slot = (h >> 1) & mask;
  for (uint64_t probe = 0; probe < 4096; ++probe, slot = (slot + 1) & mask) {  // (a table this full is not worth more probes: the read-end just stays unknown)
    if (atomicCAS(&keys[slot], 0ull, h) == 0ull) { vals[slot] = ((unsigned long long)window << 32) | d; return; }
  }
}
__global__ void k_xwin_resolve(const unsigned long long *ext, const XSrc *src, unsigned long long *listPtr, uint32_t *listCount, uint32_t n) {
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= n || !ext[d]) return;
  const unsigned long long v = ext[d] - 1;
  const XSrc s = src[v >> 32];
  listPtr[d] = s.listPtr[(uint32_t)v];
  listCount[d] = s.listCount[(uint32_t)v];
}

static int xFail(t1k_xwin *x, int code, const std::string &m) { if (x) x->err = m; return code; }
#define X_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return xFail(x, T1K_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

extern "C" int t1k_xwin_create(t1k_ctx *owner, uint64_t maxReadEnds, uint32_t maxWindows, t1k_xwin **out) {
  if (!owner || !out || !maxWindows) return T1K_ERR_ARG;
  *out = nullptr;
  t1k_xwin *x = new t1k_xwin();
  x->device = owner->device;
  x->maxWindows = maxWindows;
  uint64_t cap = 1024;
  while (cap < maxReadEnds + maxReadEnds / 3) cap <<= 1;
  x->cap = cap;
  int rc;
  if ((rc = t1k_ensure(owner, x->bKeys, cap * 8)) || (rc = t1k_ensure(owner, x->bVals, cap * 8)) || (rc = t1k_ensure(owner, x->bSrc, (size_t)maxWindows * sizeof(XSrc))) ||
      (rc = t1k_ensure(owner, x->bCount, 64))) { t1k_xwin_destroy(x); return rc; }
  if (hipSetDevice(owner->device) != hipSuccess || hipMemsetAsync(x->bKeys.p, 0, cap * 8, owner->stream) != hipSuccess || hipStreamSynchronize(owner->stream) != hipSuccess) {
    t1k_xwin_destroy(x);
    return t1k_fail(owner, T1K_ERR_DEVICE, "t1k_xwin_create: clearing the table");
  }
  x->per.resize(maxWindows);
  *out = x;
  return T1K_OK;
}
extern "C" void t1k_xwin_destroy(t1k_xwin *x) {
  if (!x) return;
  (void)hipSetDevice(x->device);
  auto drop = [](T1kDevBuf &b) { if (b.p) (void)t1k_dev_free(b.p); b = T1kDevBuf(); };
  drop(x->bKeys); drop(x->bVals); drop(x->bSrc); drop(x->bCount);
  for (auto &p : x->per) { drop(p.ext); drop(p.skip); }
  delete x;
}
extern "C" const char *t1k_xwin_last_error(const t1k_xwin *x) { return x ? x->err.c_str() : "no table"; }

// reader: the context whose read set t1k_reads_dedupe has just collapsed (window `window` of the job, a window whose lists will be kept)
extern "C" int t1k_xwin_link(t1k_xwin *x, t1k_ctx *reader, uint32_t window, uint32_t *nExternal) {
  if (!x || !reader || !nExternal || window >= x->maxWindows || reader->device != x->device || reader->readsShared) return xFail(x, T1K_ERR_ARG, "t1k_xwin_link: bad arguments");
  *nExternal = 0;
  const uint32_t n = reader->reads.nReadEnds;
  t1k_xwin::Per &P = x->per[window];
  P.n = n; P.listPtr = reader->reads.listPtr; P.listCount = reader->reads.listCount;
  if (n == 0) return T1K_OK;
  X_HIP(hipSetDevice(x->device));
  int rc;
  if ((rc = t1k_ensure(reader, P.ext, (size_t)n * 8 + 64)) || (rc = t1k_ensure(reader, P.skip, (size_t)n + 64))) return xFail(x, rc, t1k_last_error(reader));
  XSrc s{reader->reads.bases, reader->reads.nmask, reader->reads.len, reader->reads.listPtr, reader->reads.listCount, reader->reads.S, 0};
  X_HIP(hipMemcpyAsync((XSrc *)x->bSrc.p + window, &s, sizeof(XSrc), hipMemcpyHostToDevice, reader->stream));
  X_HIP(hipMemsetAsync(x->bCount.p, 0, 4, reader->stream));
  const unsigned nb = (n + 255) / 256;
  hipLaunchKernelGGL(k_xwin_lookup, dim3(nb), dim3(256), 0, reader->stream, reader->reads, (const unsigned long long *)x->bKeys.p, (const unsigned long long *)x->bVals.p, x->cap - 1,
                     (const XSrc *)x->bSrc.p, (unsigned long long *)P.ext.p, (uint8_t *)P.skip.p, (uint32_t *)x->bCount.p);
  hipLaunchKernelGGL(k_xwin_insert, dim3(nb), dim3(256), 0, reader->stream, reader->reads, (unsigned long long *)x->bKeys.p, (unsigned long long *)x->bVals.p, x->cap - 1, window,
                     (const uint8_t *)P.skip.p);
  uint32_t found = 0;
  X_HIP(hipMemcpyAsync(&found, x->bCount.p, 4, hipMemcpyDeviceToHost, reader->stream));
  X_HIP(hipStreamSynchronize(reader->stream));
  reader->reads.skip = found ? (const uint8_t *)P.skip.p : nullptr;
  *nExternal = found;
  return T1K_OK;
}
// the linked read-ends of `window` take the table entries of the windows that assigned their sequences (whose assignment ranges must
// be done); runs on ctx's stream and waits for it
extern "C" int t1k_xwin_resolve(t1k_xwin *x, t1k_ctx *ctx, uint32_t window) {
  if (!x || !ctx || window >= x->maxWindows || ctx->device != x->device) return xFail(x, T1K_ERR_ARG, "t1k_xwin_resolve: bad arguments");
  const t1k_xwin::Per &P = x->per[window];
  if (!P.n || !P.ext.p) return T1K_OK;
  X_HIP(hipSetDevice(x->device));
  hipLaunchKernelGGL(k_xwin_resolve, dim3((P.n + 255) / 256), dim3(256), 0, ctx->stream, (const unsigned long long *)P.ext.p, (const XSrc *)x->bSrc.p, P.listPtr, P.listCount, P.n);
  X_HIP(hipStreamSynchronize(ctx->stream));
  return T1K_OK;
}
