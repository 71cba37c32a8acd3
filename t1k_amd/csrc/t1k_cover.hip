// t1k_amd/csrc/t1k_cover.hip -- per-base coverage for the alleles whose coverage is actually read.
//
// SeqSet::AssignRead adds the read-end's weight to posWeight of every near-best overlap's allele (SeqSet.hpp:2253-2274); the only
// reader of posWeight is GetSeqMissingBaseCoverage (2717-2755) through Genotyper::FinalizeReadAssignments (Genotyper.hpp:935), and the
// resulting alleleInfo[].missingCoverage is consumed for the SELECTED alleles of a gene only (Genotyper.hpp:1754, 1870-1878; its use
// in EMupdate is overwritten by `adjust = 1`, 389-390 / 400-401).  A job therefore keeps the final overlap lists of its windows (a
// "read set": the packed distinct read-ends + their lists in the overlap store) resident, runs allele selection up to the point
// where missingCoverage is first read, and then adds coverage here for the records of the selected alleles alone: a scan of the
// kept lists, and the same alignment kernels t1k_assign_range uses (k_fullalign + the traced DP family) over the ~1 % that match.
// Integer sums: the result for those alleles is identical to adding every record's coverage eagerly.
#include <algorithm>
#include <vector>
#include "t1k_dev.h"
#include "t1k_launch.h"

struct t1k_readset {
  int device = 0;
  T1kReadsDev reads{};            // the distinct read-ends and the table of their final lists
  int maxLen = 0;
  std::vector<T1kDevBuf> bufs;    // everything the views above point into: read buffers, list table, overlap-store chunks
  uint64_t bytes = 0;
  std::string err;
};

// one wavefront per read-end: how many records of its list are near-best overlaps on a selected allele
__global__ __launch_bounds__(256) void k_cover_count(const unsigned long long *listPtr, const uint32_t *listCount, const uint8_t *sel, uint32_t n, uint32_t *cnt) {
  const uint32_t re = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (re >= n) return;
  const T1kOvlP *list = (const T1kOvlP *)listPtr[re];
  const uint32_t m = listCount[re];
  uint32_t c = 0;
  for (uint32_t j = lane; j < m; j += 64) {
    const unsigned long long lo = list[j].lo;
    if (((lo >> 57) & 1ull) && sel[lo & 0xFFFFFFull]) ++c;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if (lane == 0) cnt[re] = c;
}

// ... and those records as working records, read-end indices relative to re0, at out[off[re] - off0 ...] in list order
__global__ __launch_bounds__(256) void k_cover_gather(const unsigned long long *listPtr, const uint32_t *listCount, const uint8_t *sel, uint32_t re0, uint32_t n,
                                                      const unsigned long long *off, T1kOvl *out) {
  const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= n) return;
  const uint32_t re = re0 + r;
  const T1kOvlP *list = (const T1kOvlP *)listPtr[re];
  const uint32_t m = listCount[re];
  unsigned long long base = off[re] - off[re0];
  if (off[re + 1] == off[re]) return;
  for (uint32_t j0 = 0; j0 < m; j0 += 64) {
    const uint32_t j = j0 + lane;
    T1kOvlP p{0, 0};
    bool hit = false;
    if (j < m) {
      p = list[j];
      hit = ((p.lo >> 57) & 1ull) && sel[p.lo & 0xFFFFFFull];
    }
    const unsigned long long mask = __ballot(hit);
    if (hit) {
      T1kOvl o = t1k_ovl_unpack(p);
      o.re = r;
      out[base + __popcll(mask & ((1ull << lane) - 1ull))] = o;
    }
    base += __popcll(mask);
  }
}

static void rsFree(T1kDevBuf &b) {
  if (b.p) (void)t1k_dev_free(b.p);
  b.p = nullptr; b.bytes = 0;
}

extern "C" {

int t1k_device_memory(int device, uint64_t *freeBytes, uint64_t *totalBytes) {
  size_t f = 0, t = 0;
  if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(&f, &t) != hipSuccess) return T1K_ERR_DEVICE;
  // (blocks the library's pool keeps for reuse are free to the library: a process that ran a job before would otherwise look full to the
  // next job's memory rules -- a warm benchmark step with four windows fell back to per-range coverage for two of them, 3.8 s for 2.7)
  if (freeBytes) *freeBytes = std::min<uint64_t>(t, (uint64_t)f + t1k_pool_cached_bytes(device));
  if (totalBytes) *totalBytes = t;
  return T1K_OK;
}

int t1k_ctx_set_coverage_mode(t1k_ctx *ctx, int deferred) {
  if (!ctx) return T1K_ERR_ARG;
  ctx->covMode = deferred ? 1 : 0;
  return T1K_OK;
}

int t1k_readset_detach(t1k_ctx *reader, t1k_readset **out) {
  if (!reader || !out) return t1k_fail(reader, T1K_ERR_ARG, "t1k_readset_detach: bad arguments");
  *out = nullptr;
  if (reader->readsShared) return t1k_fail(reader, T1K_ERR_STATE, "t1k_readset_detach: the context aliases another context's reads");
  t1k_readset *rs = new t1k_readset();
  rs->device = reader->device;
  rs->reads = reader->reads;
  rs->maxLen = reader->batchMaxLen;
  auto steal = [&](T1kDevBuf &b) {
    if (!b.p) return;
    rs->bufs.push_back(b); rs->bytes += b.bytes;
    b = T1kDevBuf();
  };
  if (reader->reads.bases == (const uint64_t *)reader->bDedupBases.p && reader->bDedupBases.p) {
    steal(reader->bDedupBases); steal(reader->bDedupN); steal(reader->bDedupLen); steal(reader->bDedupWeight);
  } else {
    steal(reader->bReadBases); steal(reader->bReadN); steal(reader->bReadLen); steal(reader->bReadWeight);
  }
  steal(reader->bListPtr); steal(reader->bListCount);
  reader->reads = T1kReadsDev{};
  reader->rangeCount = 0;
  *out = rs;
  return T1K_OK;
}

int t1k_readset_take_store(t1k_readset *rs, t1k_ctx *pipe, int slot) {
  if (!rs || !pipe || slot < 0 || slot > 1 || pipe->device != rs->device) return T1K_ERR_ARG;
  for (T1kDevBuf &b : pipe->storeChunks[slot])
    if (b.p) { rs->bufs.push_back(b); rs->bytes += b.bytes; b = T1kDevBuf(); }
  pipe->storeChunks[slot].clear();
  pipe->storeChunk[slot] = 0; pipe->storeUsed[slot] = 0;
  return T1K_OK;
}

uint64_t t1k_readset_bytes(const t1k_readset *rs) { return rs ? rs->bytes : 0; }
uint32_t t1k_readset_size(const t1k_readset *rs) { return rs ? rs->reads.nReadEnds : 0; }
const char *t1k_readset_last_error(const t1k_readset *rs) { return rs ? rs->err.c_str() : "no read set"; }

void t1k_readset_destroy(t1k_readset *rs) {
  if (!rs) return;
  (void)hipSetDevice(rs->device);
  for (T1kDevBuf &b : rs->bufs) rsFree(b);
  delete rs;
}

int t1k_coverage_selected(t1k_ctx *ctx, t1k_readset *rs, const uint8_t *selected, uint64_t *nRecords) {
  if (nRecords) *nRecords = 0;
  if (!ctx || !rs || !selected) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_coverage_selected: bad arguments");
  if (!ctx->ref.covDiff || ctx->device != rs->device) return t1k_fail(ctx, T1K_ERR_STATE, "t1k_coverage_selected: no reference on this context / another device");
  T1K_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t D = rs->reads.nReadEnds, A = ctx->ref.nAlleles;
  if (!D) return T1K_OK;
  int rc;
  T1kDevBuf dSel, dCnt, dOff;
  auto done = [&](int code) { rsFree(dSel); rsFree(dCnt); rsFree(dOff); return code; };
  if ((rc = t1k_ensure(ctx, ctx->bCounters, (size_t)T1K_COUNTER_WORDS * 8))) return rc;
  if ((rc = t1k_ensure(ctx, dSel, (size_t)A + 16)) || (rc = t1k_ensure(ctx, dCnt, ((size_t)D + 1) * 4)) || (rc = t1k_ensure(ctx, dOff, ((size_t)D + 1) * 8))) return done(rc);
  hipError_t e = hipMemcpyAsync(dSel.p, selected, A, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync((char *)dCnt.p + (size_t)D * 4, 0, 4, ctx->stream);
  if (e != hipSuccess) return done(t1k_fail(ctx, T1K_ERR_DEVICE, hipGetErrorString(e)));
  hipLaunchKernelGGL(k_cover_count, dim3((D + 3) / 4), dim3(256), 0, ctx->stream, rs->reads.listPtr, rs->reads.listCount, (const uint8_t *)dSel.p, D, (uint32_t *)dCnt.p);
  if ((rc = t1k_exclusive_sum64(ctx, (const uint32_t *)dCnt.p, (unsigned long long *)dOff.p, D + 1))) return done(rc);
  std::vector<unsigned long long> off((size_t)D + 1);
  e = hipMemcpyAsync(off.data(), dOff.p, ((size_t)D + 1) * 8, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return done(t1k_fail(ctx, T1K_ERR_DEVICE, hipGetErrorString(e)));
  if (nRecords) *nRecords = off[D];
  // batches of consecutive read-ends holding at most `cap` records (and at most 2^20 read-ends: the DP queues' sort key keeps 20 bits of
  // the batch-relative read-end); every queue stripe can hold the whole batch, so nothing here can overflow and nothing is added twice
  uint64_t cap = 1u << 20;
  if (const char *ev = getenv("T1K_COVER_BATCH")) cap = (uint64_t)std::max(64, atoi(ev));
  unsigned long long hc[64];
  for (uint32_t i0 = 0; i0 < D;) {
    const uint32_t lim = (uint32_t)std::min<uint64_t>(D, (uint64_t)i0 + (1u << 20));
    uint32_t i1 = (uint32_t)(std::upper_bound(off.begin() + i0 + 1, off.begin() + lim + 1, off[i0] + cap) - off.begin()) - 1;
    if (i1 <= i0) i1 = i0 + 1;
    const uint64_t n = off[i1] - off[i0];
    if (n) {
      if ((rc = t1k_ensure(ctx, ctx->bOvlWork, (size_t)n * sizeof(T1kOvl)))) return done(rc);
      e = hipMemsetAsync(ctx->bCounters.p, 0, (size_t)T1K_COUNTER_WORDS * 8, ctx->stream);
      if (e != hipSuccess) return done(t1k_fail(ctx, T1K_ERR_DEVICE, hipGetErrorString(e)));
      const uint32_t cnt = i1 - i0;
      hipLaunchKernelGGL(k_cover_gather, dim3((cnt + 3) / 4), dim3(256), 0, ctx->stream, rs->reads.listPtr, rs->reads.listCount, (const uint8_t *)dSel.p, i0, cnt,
                         (const unsigned long long *)dOff.p, (T1kOvl *)ctx->bOvlWork.p);
      T1kReadsDev rd = rs->reads;
      rd.nReadEnds = cnt;
      rd.bases += (uint64_t)i0 * 2 * rd.S; rd.nmask += (uint64_t)i0 * 2 * rd.S; rd.len += i0; rd.weight += i0;
      ctx->covCommitted = true;
      if ((rc = t1k_fullalign_phase(ctx, rd, (T1kOvl *)ctx->bOvlWork.p, n, ctx->prm.relax_intron_align, 0, rs->maxLen, true, hc))) return done(rc);
    }
    i0 = i1;
  }
  e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return done(t1k_fail(ctx, T1K_ERR_DEVICE, hipGetErrorString(e)));
  return done(T1K_OK);
}

}  // extern "C"
