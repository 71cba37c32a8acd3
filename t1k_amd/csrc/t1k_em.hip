// t1k_amd/csrc/t1k_em.hip -- EM read-normalise / class-accumulate kernels (Genotyper::EMupdate, Genotyper.hpp:372-421)
// and the batched AlignAlgo::GlobalAlignment entry points used by the unit tests.
//
// E-step, bit-identical to the reference's sequential doubles:
//   k_em_psum : one wavefront per read group g: psum[g] = sum_j x[ec_j] in row order (psum == 0 -> 1)
//   k_em_cols : one wavefront per class: n[ec] = sum over its entries, in group order -- the order the reference's row-major loop adds
//               them to ecReadCount[ec] -- of count_g * (x[ec] / psum[g]): the contribution is formed where it is added (the class-major
//               list holds each entry's read group; count and psum are G doubles, cache-resident), so every class total is the same
//               rounded double as the reference's.  Rounds 1-4 had the row pass write the nnz contributions to their class-major
//               slots (23.8 M scattered 8-byte stores per update at 10 M pairs, 0.86 ms) for the class pass to read back.
// then the M-step (normalise by length, sum|diff|) runs on the host in the reference's order: it is O(#classes) and needs the
// values on the host anyway.  Sharded over GPUs (t1k_em_shard): a rank runs k_em_psum on its slice of the read groups only; the slices
// of psum are all-gathered (every element has exactly one writer: no reduction, G doubles in all -- 1.8 MB at 10 M pairs, where the
// contributions were 190 MB) and k_em_cols, on every rank, adds each class in group order: the same doubles as on one GPU, whatever
// the number of ranks.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include "t1k_dev.h"
#include "t1k_launch.h"

// acc + v(lane 0) + v(lane 1) + ... + v(lane cnt-1), added strictly in that order (the floating-point sums of the EM must follow
// the reference's order).  The operands go through 512 bytes of LDS that belong to the wavefront: every lane reads them back at the same
// addresses (a broadcast, two doubles a read) and runs the same chain of additions.  Taking them out of the lanes with v_readlane, as
// rounds 1-4 did, put two readlanes into one scalar register pair and two wait states in front of EVERY addition (14 issue cycles an
// operand); now a dependent addition follows the previous one as soon as its result is there.
__device__ __forceinline__ double waveOrderedSum(double v, int cnt, double acc, double *slot, int lane) {
  slot[lane] = v;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (cnt == 64) {
    // a full wavefront of operands (all but the last piece of a long list): no loop control between the dependent additions -- the
    // chain of one long class is the critical path of the whole E-step
#pragma unroll
    for (int j = 0; j < 64; ++j) acc += slot[j];
  } else {
    for (int j = 0; j < cnt; ++j) acc += slot[j];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();  // the reads are done before the slot is written again
  return acc;
}

// one wavefront per read group: the lanes gather the row's class abundances together, the sum is taken in row order
__global__ __launch_bounds__(256) void k_em_psum(const uint64_t *rowPtr, const uint32_t *ecIdx, const double *x, double *psumOut, uint32_t nGroups) {
  const uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  __shared__ __attribute__((aligned(16))) double sOrd[4][64];
  if (g >= nGroups) return;
  double *slot = sOrd[threadIdx.x >> 6];
  const uint64_t b = rowPtr[g], e = rowPtr[g + 1];
  double psum = 0;
  for (uint64_t base = b; base < e; base += 64) {
    const uint64_t p = base + lane;
    const double v = p < e ? x[ecIdx[p]] : 0.0;
    psum = waveOrderedSum(v, (int)(e - base < 64 ? e - base : 64), psum, slot, lane);
  }
  if (psum == 0) psum = 1;
  if (lane == 0) psumOut[g] = psum;
}

// one wavefront per class: its entries' contributions are formed and added in group order.  rowOf[j] = the read group of class-major
// slot j; groups outside [rowLo, rowHi) contribute 0.0 (T1K_EM_COLLECTIVE=allreduce: a rank adds its own groups only; x + 0.0 == x).
// The 64 dependent additions of a piece are the critical path of the E-step (one class holds ~10^5 entries), so nothing else may sit on
// it: the entries are taken 512 at a time, the read groups of the piece after next and the count / psum of the next piece are requested
// before the additions of this one start.
__global__ __launch_bounds__(256) void k_em_cols(const uint64_t *colPtr, const uint32_t *rowOf, const double *count, const double *psum, const double *x, double *n, uint32_t nEc,
                                                 uint32_t rowLo, uint32_t rowHi) {
  constexpr int K = 8;
  constexpr uint64_t S = 64 * K;
  const uint32_t ec = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  __shared__ __attribute__((aligned(16))) double sOrd[4][64];
  if (ec >= nEc) return;
  double *slot = sOrd[threadIdx.x >> 6];
  const uint64_t b = colPtr[ec], e = colPtr[ec + 1];
  const double xe = x[ec];
  uint32_t g[K];       // read groups of the piece whose values are requested next (~0u: no entry, or not this rank's group)
  double c[K], ps[K];  // count / psum of the piece that is added next
  bool ok[K];
  auto rows = [&](uint64_t start) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint64_t p = start + 64 * k + lane;
      const uint32_t r = p < e ? rowOf[p] : ~0u;
      g[k] = r != ~0u && r - rowLo < rowHi - rowLo ? r : ~0u;
    }
  };
  auto values = [&]() {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      ok[k] = g[k] != ~0u;
      const uint32_t gi = ok[k] ? g[k] : rowLo;  // (any valid index: the value is not used)
      c[k] = ok[k] ? count[gi] : 0.0;
      ps[k] = ok[k] ? psum[gi] : 1.0;
    }
  };
  double s = 0;
  if (b < e) {
    rows(b);
    values();
    rows(b + S);
  }
  for (uint64_t base = b; base < e; base += S) {
    double v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = ok[k] ? c[k] * (xe / ps[k]) : 0.0;
    values();            // the next piece's count / psum (its read groups arrived during the previous piece's additions)
    rows(base + 2 * S);  // the read groups of the piece after next
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint64_t cb = base + 64 * k;
      if (cb < e) s = waveOrderedSum(v[k], (int)(e - cb < 64 ? e - cb : 64), s, slot, lane);
    }
  }
  if (lane == 0) n[ec] = s;
}

// t1k_em_setup: sort keys (class, entry) + class sizes
__global__ void k_em_keys(const uint32_t *ecIdx, uint32_t nnz, uint32_t nEc, unsigned long long *key, uint32_t *val, unsigned long long *colCount, unsigned long long *bad) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nnz) return;
  const uint32_t ec = ecIdx[p];
  if (ec >= nEc) { atomicOr(bad, 1ull); key[p] = 0; val[p] = p; return; }
  key[p] = ec; val[p] = p;
  atomicAdd(&colCount[ec], 1ull);
}
// the read group of every entry (row-major), then of every class-major slot
__global__ __launch_bounds__(256) void k_em_entry_rows(const uint64_t *rowPtr, uint32_t nGroups, uint32_t *rowOfEntry) {
  const uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (g >= nGroups) return;
  for (uint64_t p = rowPtr[g] + lane; p < rowPtr[g + 1]; p += 64) rowOfEntry[p] = g;
}
__global__ void k_em_slot_rows(const uint32_t *sortedEntry, const uint32_t *rowOfEntry, uint32_t nnz, uint32_t *rowOf) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < nnz) rowOf[j] = rowOfEntry[sortedEntry[j]];
}

// ------------------------------------------------------------------------------------------------------------------
// batched GlobalAlignment for tests
// ------------------------------------------------------------------------------------------------------------------
struct AlignJobArgs {
  const uint64_t *tb, *tn, *pb, *pn;   // packed text / pattern streams
  const uint64_t *tPos, *pPos;         // start position of each job in the packed streams
  const uint32_t *tLen, *pLen;
  uint32_t nJobs;
  int32_t *score, *nMatch, *nMismatch, *nIndel;
  int8_t *ops; const uint32_t *opsOff; uint32_t *nOps;
  uint8_t *scratch; uint64_t perThread; int maxCells;
  unsigned long long *err;
};

__global__ __launch_bounds__(64) void k_align_batch(AlignJobArgs P) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, nT = gridDim.x * blockDim.x;
  uint8_t *mine = P.scratch + (uint64_t)t * P.perThread;
  int *rows = (int *)mine;
  int8_t *ops = (int8_t *)(mine + 6 * (2048 + 4) * 4);
  uint8_t *trace = mine + 6 * (2048 + 4) * 4 + 4224;
  for (uint32_t q = t; q < P.nJobs; q += nT) {
    int lt = (int)P.tLen[q], lp = (int)P.pLen[q];
    if (lt > 2048 || lp > 2048 || (lt + 1) * (lp + 1) > P.maxCells) { atomicOr(P.err, 1ull); continue; }
    T1kSeqView T{P.tb, P.tn, (int64_t)P.tPos[q]}, Pv{P.pb, P.pn, (int64_t)P.pPos[q]};
    int nm = 0;
    int sc = t1k_ga_general(T, lt, Pv, lp, rows, trace, &nm);
    int n = t1k_ga_traceback(trace, lt, lp, ops);
    int c0 = 0, c1 = 0, c2 = 0;
    for (int i = 0; i < n; ++i) { if (ops[i] == 0) ++c0; else if (ops[i] == 1) ++c1; else ++c2; }
    P.score[q] = sc;
    P.nMatch[q] = c0; P.nMismatch[q] = c1; P.nIndel[q] = c2;
    if (nm != c0) atomicOr(P.err, 2ull);  // the forward-sweep count must agree with the traceback
    P.nOps[q] = (uint32_t)n;
    if (P.ops) for (int i = 0; i < n; ++i) P.ops[P.opsOff[q] + i] = ops[i];
  }
}

__global__ void k_align_count(const uint64_t *tb, const uint64_t *tn, const uint64_t *pb, const uint64_t *pn, const uint64_t *tPos, const uint64_t *pPos,
                              const uint32_t *len, uint32_t nJobs, int32_t *nMatch) {
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nJobs) return;
  nMatch[q] = t1k_ga_matches_window(pb, pn, (int64_t)pPos[q], tb, tn, (int64_t)tPos[q], (int)len[q], nullptr);
}

static void packStream(const char *s, const uint32_t *off, const uint32_t *len, uint32_t n, std::vector<uint64_t> &b, std::vector<uint64_t> &nm,
                       std::vector<uint64_t> &pos) {
  uint64_t total = 0;
  pos.resize(n);
  for (uint32_t i = 0; i < n; ++i) { pos[i] = total; total += ((uint64_t)len[i] + 31) / 32 * 32; }
  b.assign(total / 32 + 4, 0); nm.assign(total / 32 + 4, 0);
  for (uint32_t i = 0; i < n; ++i)
    for (uint32_t j = 0; j < len[i]; ++j) {
      char c = s[off[i] + j];
      int code = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;
      uint64_t p = pos[i] + j;
      if (code == 4) nm[p >> 5] |= 1ull << ((p & 31) * 2); else b[p >> 5] |= (uint64_t)code << ((p & 31) * 2);
    }
}

template <typename T>
static int up(t1k_ctx *ctx, T1kDevBuf &buf, const std::vector<T> &v) {
  int rc = t1k_ensure(ctx, buf, v.size() * sizeof(T) + 16);
  if (rc) return rc;
  if (!v.empty()) T1K_HIP(ctx, hipMemcpyAsync(buf.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  return 0;
}

extern "C" {

int t1k_align_batch(t1k_ctx *ctx, const char *text, const uint32_t *tOff, const uint32_t *tLen, const char *pat, const uint32_t *pOff,
                    const uint32_t *pLen, uint32_t nJobs, int32_t *score, int32_t *nMatch, int32_t *nMismatch, int32_t *nIndel, int8_t *ops,
                    const uint32_t *opsOff, uint32_t *nOps) {
  if (!ctx || !text || !pat || !tOff || !pOff || !tLen || !pLen) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_align_batch: bad arguments");
  if (nJobs == 0) return T1K_OK;
  T1K_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<uint64_t> tb, tn, tp, pb, pn, pp;
  packStream(text, tOff, tLen, nJobs, tb, tn, tp);
  packStream(pat, pOff, pLen, nJobs, pb, pn, pp);
  int maxCells = 0;
  uint64_t opsTotal = 0;
  for (uint32_t i = 0; i < nJobs; ++i) {
    maxCells = std::max<int>(maxCells, (int)((tLen[i] + 1) * (pLen[i] + 1)));
    if (ops) opsTotal = std::max<uint64_t>(opsTotal, (uint64_t)opsOff[i] + tLen[i] + pLen[i] + 2);
  }
  const size_t perThread = (size_t)6 * (2048 + 4) * 4 + 4224 + (size_t)maxCells + 64;
  // one lane per alignment: 64 workgroups of one wavefront for a handful of jobs (the vectors of the tests), up to 512 when the analyzer's
  // variant pass sends 10^5 at a time -- as long as the lanes' trace scratch stays under 4 GB
  int blocks = (int)std::min<uint64_t>(512, std::max<uint64_t>(64, ((uint64_t)nJobs + 127) / 128));
  blocks = (int)std::max<uint64_t>(64, std::min<uint64_t>((uint64_t)blocks, ((uint64_t)4 << 30) / (64 * perThread)));
  T1kDevBuf *B = ctx->bAlign;
  int rc;
  if ((rc = up(ctx, B[0], tb)) || (rc = up(ctx, B[1], tn)) || (rc = up(ctx, B[2], pb)) || (rc = up(ctx, B[3], pn)) || (rc = up(ctx, B[4], tp)) ||
      (rc = up(ctx, B[5], pp)))
    return rc;
  std::vector<uint32_t> tl(tLen, tLen + nJobs), pl(pLen, pLen + nJobs), oo;
  if (ops) oo.assign(opsOff, opsOff + nJobs); else oo.assign(nJobs, 0);
  if ((rc = up(ctx, B[6], tl)) || (rc = up(ctx, B[7], pl)) || (rc = up(ctx, B[8], oo))) return rc;
  if ((rc = t1k_ensure(ctx, B[9], (size_t)nJobs * 5 * 4 + 64))) return rc;           // score, nMatch, nMismatch, nIndel, nOps
  if ((rc = t1k_ensure(ctx, B[10], opsTotal + 64))) return rc;
  if ((rc = t1k_ensure(ctx, B[11], (size_t)blocks * 64 * perThread + 64))) return rc;
  if ((rc = t1k_ensure(ctx, ctx->bCounters, 64 * 8))) return rc;
  T1K_HIP(ctx, hipMemsetAsync(ctx->bCounters.p, 0, 64 * 8, ctx->stream));
  AlignJobArgs a{};
  a.tb = (uint64_t *)B[0].p; a.tn = (uint64_t *)B[1].p; a.pb = (uint64_t *)B[2].p; a.pn = (uint64_t *)B[3].p;
  a.tPos = (uint64_t *)B[4].p; a.pPos = (uint64_t *)B[5].p; a.tLen = (uint32_t *)B[6].p; a.pLen = (uint32_t *)B[7].p; a.nJobs = nJobs;
  int32_t *res = (int32_t *)B[9].p;
  a.score = res; a.nMatch = res + nJobs; a.nMismatch = res + 2 * (size_t)nJobs; a.nIndel = res + 3 * (size_t)nJobs; a.nOps = (uint32_t *)(res + 4 * (size_t)nJobs);
  a.ops = ops ? (int8_t *)B[10].p : nullptr; a.opsOff = (uint32_t *)B[8].p;
  a.scratch = (uint8_t *)B[11].p; a.perThread = perThread; a.maxCells = maxCells;
  a.err = (unsigned long long *)ctx->bCounters.p;
  hipLaunchKernelGGL(k_align_batch, dim3(blocks), dim3(64), 0, ctx->stream, a);
  std::vector<int32_t> h((size_t)nJobs * 5);
  unsigned long long err = 0;
  T1K_HIP(ctx, hipMemcpyAsync(h.data(), res, h.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
  T1K_HIP(ctx, hipMemcpyAsync(&err, ctx->bCounters.p, 8, hipMemcpyDeviceToHost, ctx->stream));
  if (ops) T1K_HIP(ctx, hipMemcpyAsync(ops, B[10].p, opsTotal, hipMemcpyDeviceToHost, ctx->stream));
  T1K_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (err & 1) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_align_batch: sequence longer than 2048");
  if (err & 2) return t1k_fail(ctx, T1K_ERR_DEVICE, "t1k_align_batch: forward-sweep match count disagrees with traceback");
  if (score) memcpy(score, h.data(), (size_t)nJobs * 4);
  if (nMatch) memcpy(nMatch, h.data() + nJobs, (size_t)nJobs * 4);
  if (nMismatch) memcpy(nMismatch, h.data() + 2 * (size_t)nJobs, (size_t)nJobs * 4);
  if (nIndel) memcpy(nIndel, h.data() + 3 * (size_t)nJobs, (size_t)nJobs * 4);
  if (nOps) memcpy(nOps, h.data() + 4 * (size_t)nJobs, (size_t)nJobs * 4);
  return T1K_OK;
}

int t1k_align_count_batch(t1k_ctx *ctx, const char *text, const uint32_t *tOff, const char *pat, const uint32_t *pOff, const uint32_t *len, uint32_t nJobs,
                          int32_t *nMatch) {
  if (!ctx || !text || !pat || !tOff || !pOff || !len || !nMatch) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_align_count_batch: bad arguments");
  if (nJobs == 0) return T1K_OK;
  T1K_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<uint64_t> tb, tn, tp, pb, pn, pp;
  packStream(text, tOff, len, nJobs, tb, tn, tp);
  packStream(pat, pOff, len, nJobs, pb, pn, pp);
  T1kDevBuf *B = ctx->bAlign;
  int rc;
  if ((rc = up(ctx, B[0], tb)) || (rc = up(ctx, B[1], tn)) || (rc = up(ctx, B[2], pb)) || (rc = up(ctx, B[3], pn)) || (rc = up(ctx, B[4], tp)) ||
      (rc = up(ctx, B[5], pp)))
    return rc;
  std::vector<uint32_t> l(len, len + nJobs);
  if ((rc = up(ctx, B[6], l))) return rc;
  if ((rc = t1k_ensure(ctx, B[9], (size_t)nJobs * 4 + 64))) return rc;
  hipLaunchKernelGGL(k_align_count, dim3((nJobs + 255) / 256), dim3(256), 0, ctx->stream, (uint64_t *)B[0].p, (uint64_t *)B[1].p, (uint64_t *)B[2].p,
                     (uint64_t *)B[3].p, (uint64_t *)B[4].p, (uint64_t *)B[5].p, (uint32_t *)B[6].p, nJobs, (int32_t *)B[9].p);
  T1K_HIP(ctx, hipMemcpyAsync(nMatch, B[9].p, (size_t)nJobs * 4, hipMemcpyDeviceToHost, ctx->stream));
  T1K_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return T1K_OK;
}

int t1k_em_setup(t1k_ctx *ctx, const uint64_t *rowPtr, const uint32_t *ecIdx, const double *count, const int32_t *ecLen, uint32_t nGroups, uint32_t nEc,
                 t1k_allreduce_fn allreduce, void *user) {
  if (!ctx || !rowPtr || (!ecIdx && nGroups && rowPtr[nGroups]) || (!count && nGroups) || (!ecLen && nEc)) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_em_setup: bad arguments");
  T1K_HIP(ctx, hipSetDevice(ctx->device));
  const uint64_t nnz = rowPtr[nGroups];
  int rc;
  if ((rc = t1k_ensure(ctx, ctx->bEmRowPtr, (size_t)(nGroups + 1) * 8)) || (rc = t1k_ensure(ctx, ctx->bEmEc, (size_t)nnz * 4 + 16)) ||
      (rc = t1k_ensure(ctx, ctx->bEmCount, (size_t)nGroups * 8 + 16)) || (rc = t1k_ensure(ctx, ctx->bEmColPtr, (size_t)(nEc + 2) * 8)) ||
      (rc = t1k_ensure(ctx, ctx->bEmPsum, (size_t)nGroups * 8 + 16)) || (rc = t1k_ensure(ctx, ctx->bEmRowOf, (size_t)nnz * 4 + 16)) ||
      (rc = t1k_ensure(ctx, ctx->bEmX0, (size_t)nEc * 8 + 16)) || (rc = t1k_ensure(ctx, ctx->bEmN, (size_t)nEc * 8 + 16)))
    return rc;
  T1K_HIP(ctx, hipMemcpyAsync(ctx->bEmRowPtr.p, rowPtr, (size_t)(nGroups + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  if (nnz) T1K_HIP(ctx, hipMemcpyAsync(ctx->bEmEc.p, ecIdx, (size_t)nnz * 4, hipMemcpyHostToDevice, ctx->stream));
  if (nGroups) T1K_HIP(ctx, hipMemcpyAsync(ctx->bEmCount.p, count, (size_t)nGroups * 8, hipMemcpyHostToDevice, ctx->stream));
  // Class-major order: a STABLE sort of the entries by class keeps them in group order within a class (the order in which the
  // reference's row-major loop adds them to ecReadCount[ec]).  On the device: radix sort of (class, entry); slot j of the sorted
  // list keeps the read group of its entry (rowOf[j]); class starts from the sorted keys.
  T1K_HIP(ctx, hipMemsetAsync(ctx->bEmColPtr.p, 0, (size_t)(nEc + 2) * 8, ctx->stream));
  if (nnz) {
    if (nnz >= 0xFFFFFFFFull) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_em_setup: more than 2^32 entries");
    T1kDevBuf tmp;
    if ((rc = t1k_ensure(ctx, tmp, (size_t)nnz * 28 + 64))) return rc;
    unsigned long long *k0 = (unsigned long long *)tmp.p, *k1 = k0 + nnz;
    uint32_t *v0 = (uint32_t *)(k1 + nnz), *v1 = v0 + nnz, *rowOfEntry = v1 + nnz;  // v1[j] = the entry in class-major slot j
    unsigned long long *bad = (unsigned long long *)ctx->bEmN.p;  // a word that is free until the first update
    T1K_HIP(ctx, hipMemsetAsync(bad, 0, 8, ctx->stream));
    const unsigned nb = (unsigned)((nnz + 255) / 256);
    hipLaunchKernelGGL(k_em_keys, dim3(nb), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->bEmEc.p, (uint32_t)nnz, nEc, k0, v0, (unsigned long long *)ctx->bEmColPtr.p, bad);
    int bits = 1;
    while ((1ull << bits) < nEc) ++bits;
    rc = t1k_sort_pairs(ctx, k0, k1, v0, v1, (uint32_t)nnz, bits);
    if (rc == T1K_OK) {
      hipLaunchKernelGGL(k_em_entry_rows, dim3((nGroups + 3) / 4), dim3(256), 0, ctx->stream, (const uint64_t *)ctx->bEmRowPtr.p, nGroups, rowOfEntry);
      hipLaunchKernelGGL(k_em_slot_rows, dim3(nb), dim3(256), 0, ctx->stream, (const uint32_t *)v1, (const uint32_t *)rowOfEntry, (uint32_t)nnz, (uint32_t *)ctx->bEmRowOf.p);
      unsigned long long isBad = 0;
      hipError_t e = hipMemcpyAsync(&isBad, bad, 8, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      (void)t1k_dev_free(tmp.p);
      if (e != hipSuccess) return t1k_fail(ctx, T1K_ERR_DEVICE, hipGetErrorString(e));
      if (isBad) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_em_setup: class index out of range");
    } else { (void)hipStreamSynchronize(ctx->stream); (void)t1k_dev_free(tmp.p); return rc; }
  }
  // class sizes -> class starts (colPtr[nEc] = nnz)
  if ((rc = t1k_exclusive_sum_u64(ctx, (const unsigned long long *)ctx->bEmColPtr.p, (unsigned long long *)ctx->bEmColPtr.p, (uint64_t)nEc + 1))) return rc;
  T1K_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->emGroups = nGroups; ctx->emEc = nEc; ctx->emNnz = nnz;
  ctx->emAllreduce = allreduce; ctx->emUser = user;
  ctx->emRowBegin = 0; ctx->emRowEnd = nGroups; ctx->emComm = nullptr; ctx->emReduceMode = false;
  // class lengths stay on the host (M-step)
  if (nEc) ctx->hEmLen.assign(ecLen, ecLen + nEc); else ctx->hEmLen.clear();
  return T1K_OK;
}

int t1k_em_shard(t1k_ctx *ctx, uint32_t rowBegin, uint32_t rowEnd, t1k_comm *comm) {
  if (!ctx || rowBegin > rowEnd || rowEnd > ctx->emGroups) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_em_shard: bad row range");
  ctx->emRowBegin = rowBegin; ctx->emRowEnd = rowEnd; ctx->emComm = comm;
  // every rank's piece of the per-group sums: the ranks' row ranges partition [0, G) in rank order
  ctx->emPieceBytes.clear(); ctx->emPieceDispl.clear();
  const int N = comm ? t1k_comm_size(comm) : 1;
  if (N > 1) {
    std::vector<uint64_t> mine{rowBegin, rowEnd}, all((size_t)2 * N);
    const int rc = t1k_comm_allgather_u64(comm, mine.data(), 2, all.data());
    if (rc != T1K_OK) return t1k_fail(ctx, rc, std::string("t1k_em_shard: ") + t1k_comm_last_error(comm));
    uint64_t next = 0;
    for (int r = 0; r < N; ++r) {
      if (all[2 * r] != next || all[2 * r + 1] < all[2 * r] || all[2 * r + 1] > ctx->emGroups) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_em_shard: the ranks' row ranges do not partition the read groups in rank order");
      next = all[2 * r + 1];
      ctx->emPieceDispl.push_back(all[2 * r] * 8);
      ctx->emPieceBytes.push_back((all[2 * r + 1] - all[2 * r]) * 8);
    }
    if (next != ctx->emGroups) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_em_shard: the ranks' row ranges do not cover the read groups");
  }
  // T1K_EM_COLLECTIVE=allreduce: the collective north_star names -- every rank adds up its OWN rows' contributions per class and the E
  // partial sums are all-reduced (E doubles per update instead of the gather of the G per-group sums).  The class totals then differ from
  // the one-GPU run's by the re-association of the sum (a few ulp), so this mode is opt-in; the default gathers psum and stays bit-exact.
  const char *mode = getenv("T1K_EM_COLLECTIVE");
  ctx->emReduceMode = N > 1 && mode && !strcmp(mode, "allreduce");
  return T1K_OK;
}

int t1k_em_update(t1k_ctx *ctx, const double *x0, double *x1, double *ecReadCount, double *diff) {
  if (!ctx || !x0 || !x1 || !ecReadCount) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_em_update: bad arguments");
  T1K_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t E = ctx->emEc, G = ctx->emGroups;
  if (E == 0) { if (diff) *diff = 0; return T1K_OK; }
  // the vectors are small (8 B per class) but there are ~40 updates per job: page-locked staging keeps each copy a plain DMA
  if (ctx->emPinnedN < E) {
    if (ctx->emPinned) (void)hipHostFree(ctx->emPinned);
    ctx->emPinned = nullptr; ctx->emPinnedN = 0;
    T1K_HIP(ctx, hipHostMalloc((void **)&ctx->emPinned, (size_t)E * 16, hipHostMallocDefault));
    ctx->emPinnedN = E;
  }
  double *px = ctx->emPinned, *pn = ctx->emPinned + ctx->emPinnedN;
  const auto tu0 = std::chrono::steady_clock::now();
  memcpy(px, x0, (size_t)E * 8);
  T1K_HIP(ctx, hipMemcpyAsync(ctx->bEmX0.p, px, (size_t)E * 8, hipMemcpyHostToDevice, ctx->stream));
  const bool reduce = ctx->emComm && t1k_comm_size(ctx->emComm) > 1 && ctx->emReduceMode;
  const bool sharded = ctx->emComm && t1k_comm_size(ctx->emComm) > 1 && !reduce;
  const uint32_t g0 = (sharded || reduce) ? ctx->emRowBegin : 0, gn = ((sharded || reduce) ? ctx->emRowEnd : G) - g0;
  if (gn) hipLaunchKernelGGL(k_em_psum, dim3((gn + 3) / 4), dim3(256), 0, ctx->stream, (const uint64_t *)ctx->bEmRowPtr.p + g0, (const uint32_t *)ctx->bEmEc.p,
                             (const double *)ctx->bEmX0.p, (double *)ctx->bEmPsum.p + g0, gn);
  if (sharded) {
    // this rank's groups are one contiguous piece of psum: gather everybody's piece in place (G doubles in all)
    T1K_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int me = t1k_comm_rank(ctx->emComm);
    const int rc = t1k_comm_allgatherv(ctx->emComm, (const char *)ctx->bEmPsum.p + ctx->emPieceDispl[me], ctx->emPieceBytes.data(), ctx->emPieceDispl.data(), ctx->bEmPsum.p);
    if (rc != T1K_OK) return t1k_fail(ctx, rc, std::string("t1k_em_update: ") + t1k_comm_last_error(ctx->emComm));
  }
  hipLaunchKernelGGL(k_em_cols, dim3((E + 3) / 4), dim3(256), 0, ctx->stream, (const uint64_t *)ctx->bEmColPtr.p, (const uint32_t *)ctx->bEmRowOf.p, (const double *)ctx->bEmCount.p,
                     (const double *)ctx->bEmPsum.p, (const double *)ctx->bEmX0.p, (double *)ctx->bEmN.p, E, reduce ? g0 : 0u, reduce ? g0 + gn : G);
  if (reduce) {  // partial class totals of this rank's rows -> totals of all rows (E doubles over RCCL / the in-process transport)
    T1K_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int rc = t1k_comm_allreduce(ctx->emComm, ctx->bEmN.p, E, 1);
    if (rc != T1K_OK) return t1k_fail(ctx, rc, std::string("t1k_em_update: ") + t1k_comm_last_error(ctx->emComm));
  }
  if (ctx->emAllreduce) {
    T1K_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->emAllreduce(ctx->bEmN.p, E, ctx->emUser);  // RCCL all-reduce of the per-class expected read counts
  }
  T1K_HIP(ctx, hipMemcpyAsync(pn, ctx->bEmN.p, (size_t)E * 8, hipMemcpyDeviceToHost, ctx->stream));
  const auto tu1 = std::chrono::steady_clock::now();
  T1K_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const auto tu2 = std::chrono::steady_clock::now();
  memcpy(ecReadCount, pn, (size_t)E * 8);
  // M-step (Genotyper.hpp:406-420), same summation order as the reference
  double norm = 0, d = 0;
  for (uint32_t i = 0; i < E; ++i) norm += ecReadCount[i] / ctx->hEmLen[i];
  for (uint32_t i = 0; i < E; ++i) {
    double t = ecReadCount[i] / ctx->hEmLen[i] / norm;
    d += std::fabs(t - x0[i]);
    x1[i] = t;
  }
  if (diff) *diff = d;
  const auto tu3 = std::chrono::steady_clock::now();
  ctx->emMs[0] += std::chrono::duration<double, std::milli>(tu1 - tu0).count();
  ctx->emMs[1] += std::chrono::duration<double, std::milli>(tu2 - tu1).count();
  ctx->emMs[2] += std::chrono::duration<double, std::milli>(tu3 - tu2).count();
  ctx->emMs[3] += 1;
  return T1K_OK;
}
void t1k_em_times(const t1k_ctx *ctx, double *ms4) { for (int i = 0; i < 4; ++i) ms4[i] = ctx ? ctx->emMs[i] : 0; }

}  // extern "C"
