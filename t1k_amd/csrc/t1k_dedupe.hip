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
  ctx->rangeCount = 0;
  *nDistinct = D;
  return T1K_OK;
}
