// t1k_amd/csrc/t1k_refindex.hip -- the allele reference in HBM: 2-bit packing and the k-mer index, built ON THE DEVICE.
//
// Reference: SeqSet::InputRefSeq (SeqSet.hpp:906-982) stores every allele as chars and feeds it to KmerIndex::BuildIndexFromRead
// (KmerIndex.hpp:107-130): a map from k-mer code to the list of (allele, offset) in insertion order, with the insert rule of line 121
// (a window is inserted if it holds no N and -- it is the window ending at position k, or its code differs from the previous
// window's; the first window is compared with code 0, so an all-A first window is left out).
// Here (round 1 did this on 16 host threads in 0.6-1.2 s, the largest fixed cost of a run):
//   k_ref_pack      ASCII -> 2-bit bases / N mask / exon mask, one thread per 32-base word (every allele starts on a word boundary)
//   k_ref_codes     one thread per position: the code of the window ending there, straight from the packed words (first base in the
//                   low bits), validity from the N mask, the insert rule from the neighbouring window's code
//   rocPRIM         stable radix sort of (code, global position): postings of a code come out in (allele, offset) order, which is the
//                   reference's insertion order
//   k_ref_postings  global position -> (allele, offset) through a per-word allele table; bucket counts by atomics, bucket starts by
//                   an exclusive scan over the 4^k codes
//   k_ref_flags     per code: presence / multiplicity / canonical-prefix bitmaps of the extractor, "needs a chunk directory"
//   k_ref_dir       per (directory row, allele chunk): first posting of the list at or beyond the chunk, by bisection
// Everything is integer, HBM-bound work on ~26 M positions (HLA-sized reference): milliseconds.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include "t1k_dev.h"
#include "t1k_launch.h"

namespace {

__global__ void k_ref_pack(const char *ascii, const uint8_t *exon, const uint64_t *srcOff, const uint64_t *alleleOff, const uint32_t *alleleLen, const uint32_t *wordAllele,
                           uint64_t nWords, uint64_t *bases, uint64_t *nmask, uint64_t *exonm, int nCode) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nWords) return;
  const uint32_t a = wordAllele[w];
  uint64_t b = 0, n = 0, e = 0;
  if (a != 0xFFFFFFFFu) {
    const uint64_t first = w * 32 - alleleOff[a];  // offset of the word's first base inside the allele
    const uint32_t len = alleleLen[a];
    const char *s = ascii + srcOff[a];
    const uint8_t *x = exon ? exon + srcOff[a] : nullptr;
    for (int q = 0; q < 32; ++q) {
      const uint64_t i = first + q;
      if (i >= len) break;
      const char c = s[i];
      const int code = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;
      if (code == 4) { n |= 1ull << (2 * q); b |= (uint64_t)nCode << (2 * q); } else b |= (uint64_t)code << (2 * q);
      if (x && x[i]) e |= 1ull << (2 * q);
    }
  }
  bases[w] = b; nmask[w] = n; exonm[w] = e;
}

// the transposed copy of the bases (T1kRefDev::basesT): one thread per word of the copy; block b = alleles [64 b, 64 b + 64), rows[b] rows of 64 words
__global__ void k_ref_transpose(const uint64_t *bases, const uint64_t *alleleOff, const uint32_t *alleleLen, const uint32_t *blockT, uint32_t nBlocks, uint32_t nAlleles, uint64_t totalT,
                                uint64_t *basesT) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= totalT) return;
  uint32_t lo = 0, hi = nBlocks - 1;   // the last block whose first word is <= t
  while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if ((uint64_t)blockT[mid] <= t) lo = mid; else hi = mid - 1; }
  const uint64_t rel = t - blockT[lo];
  const uint32_t a = lo * 64 + (uint32_t)(rel & 63);
  const uint64_t w = rel >> 6;
  uint64_t v = 0;
  if (a < nAlleles && w < ((uint64_t)alleleLen[a] + 31) / 32) v = bases[(alleleOff[a] >> 5) + w];
  basesT[t] = v;
}

// 2k bits starting at global position p (k <= 15)
__device__ __forceinline__ uint32_t windowBits(const uint64_t *w, uint64_t p, int k) {
  const uint64_t wi = p >> 5;
  const int sh = (int)(p & 31) * 2;
  uint64_t v = w[wi] >> sh;
  if (sh) v |= w[wi + 1] << (64 - sh);
  return (uint32_t)(v & ((1ull << (2 * k)) - 1));
}

// flag[g] = 1 if the window ENDING at global position g is inserted; key[g] = its code
__global__ void k_ref_codes(const uint64_t *bases, const uint64_t *nmask, const uint64_t *alleleOff, const uint32_t *alleleLen, const uint32_t *wordAllele, uint64_t total,
                            int k, uint32_t *flag, uint32_t *code) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total) return;
  uint32_t f = 0, c = 0;
  const uint32_t a = wordAllele[g >> 5];
  if (a != 0xFFFFFFFFu) {
    const uint64_t i = g - alleleOff[a];
    if (i < alleleLen[a] && i + 1 >= (uint64_t)k) {
      const uint64_t start = g + 1 - k;
      c = windowBits(bases, start, k);
      const bool valid = windowBits(nmask, start, k) == 0;
      // KmerIndex.hpp:121: prev is the code of the window ending at i - 1 (whatever its validity), 0 before the first window
      const uint32_t prev = i + 1 == (uint64_t)k ? 0u : windowBits(bases, start - 1, k);
      f = valid && (i == (uint64_t)k || c != prev) ? 1u : 0u;
    }
  }
  flag[g] = f; code[g] = c;
}
// posted bitmap (nmask geometry) from the per-END-position insert flags: the window starting at s ends at s + k - 1
__global__ void k_ref_posted(const uint32_t *flag, uint64_t total, int k, uint64_t nWords, uint64_t *posted) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nWords) return;
  uint64_t v = 0;
  for (int q = 0; q < 32; ++q) {
    const uint64_t e = w * 32 + q + (uint64_t)(k - 1);
    if (e < total && flag[e]) v |= 1ull << (2 * q);
  }
  posted[w] = v;
}
__global__ void k_ref_compact(const uint32_t *flag, const uint32_t *pos, const uint32_t *code, unsigned long long *key, uint32_t *val, uint64_t total) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < total && flag[g]) { key[pos[g] - 1] = code[g]; val[pos[g] - 1] = (uint32_t)g; }
}
// sorted (code, end position) -> posting (allele, offset of the window's first base); bucket counts
__global__ void k_ref_postings(const unsigned long long *key, const uint32_t *val, const uint64_t *alleleOff, const uint32_t *wordAllele, int k, T1kPosting *post,
                               uint32_t *postAllele, uint32_t *count, uint32_t m) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const uint32_t g = val[j];
  const uint32_t a = wordAllele[g >> 5];
  post[j] = T1kPosting{a, (uint32_t)(g - alleleOff[a]) - (uint32_t)(k - 1)};
  postAllele[j] = a;
  atomicAdd(&count[(uint32_t)key[j]], 1u);
}
// kStart[c] .. kStart[c + 1] is list c.  Bitmaps and "this list gets a directory row" (longer than T1K_DIR_MINLEN).
__global__ void k_ref_flags(const uint32_t *kStart, const uint32_t *postAllele, uint64_t nKeys, int k, uint32_t *has, uint32_t *multi, uint32_t *hasPre, uint32_t *needDir) {
  const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nKeys) return;
  const uint32_t st = kStart[c], ln = kStart[c + 1] - st;
  needDir[c] = ln > T1K_DIR_MINLEN ? 1u : 0u;
  if (!ln) return;
  atomicOr(&has[c >> 5], 1u << (c & 31));
  const int kp = k - 2 > 1 ? k - 2 : 1;
  const uint32_t rc = t1k_code_revcomp((uint32_t)c, k);
  const uint32_t pc = ((uint32_t)c < rc ? (uint32_t)c : rc) & (uint32_t)((1ull << (2 * kp)) - 1);
  atomicOr(&hasPre[pc >> 5], 1u << (pc & 31));
  for (uint32_t i = 1; i < ln; ++i)
    if (postAllele[st + i] == postAllele[st + i - 1]) { atomicOr(&multi[c >> 5], 1u << (c & 31)); break; }
}
__global__ void k_ref_diridx(const uint32_t *needDir, const uint32_t *rowOf, uint32_t *dirIdx, uint32_t *rowCode, uint64_t nKeys) {
  const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nKeys) return;
  if (needDir[c]) { dirIdx[c] = rowOf[c] - 1; rowCode[rowOf[c] - 1] = (uint32_t)c; }
  else dirIdx[c] = T1K_NO_DIR;
}
// dir[row][cidx] = first posting of the row's list whose allele is >= cidx * T1K_SEED_CHUNK (relative to the list start)
__global__ void k_ref_dir(const uint32_t *rowCode, const uint32_t *kStart, const uint32_t *postAllele, uint32_t stride, uint64_t nCells, uint32_t *dir) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nCells) return;
  const uint32_t row = (uint32_t)(t / stride), cidx = (uint32_t)(t % stride);
  const uint32_t c = rowCode[row];
  const uint32_t st = kStart[c], ln = kStart[c + 1] - st;
  const uint32_t bound = cidx * T1K_SEED_CHUNK;
  uint32_t lo = 0, hi = ln;
  while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (postAllele[st + mid] < bound) lo = mid + 1; else hi = mid; }
  dir[t] = lo;
}

// mask[row][w] bit b = chunk 64 * w + b of the row's list is not empty
__global__ void k_ref_dirmask(const uint32_t *dir, uint32_t stride, uint32_t rows, uint32_t words, unsigned long long *mask) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (uint64_t)rows * words) return;
  const uint32_t row = (uint32_t)(t / words), w = (uint32_t)(t % words);
  const uint32_t *d = dir + (uint64_t)row * stride;
  unsigned long long m = 0;
  for (uint32_t b = 0; b < 64; ++b) {
    const uint32_t c = 64 * w + b;
    if (c + 1 < stride && d[c + 1] > d[c]) m |= 1ull << b;
  }
  mask[t] = m;
}

static inline int asciiCode(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; }

int keep(t1k_ctx *ctx, size_t bytes, void **out) {
  T1kDevBuf b;
  int rc = t1k_ensure(ctx, b, bytes);
  if (rc) return rc;
  ctx->refBufs.push_back(b);
  *out = b.p;
  return 0;
}

}  // namespace

extern "C" int t1k_ref_upload(t1k_ctx *ctx, const char *seqs, const uint64_t *offsets, const uint8_t *exon, uint32_t nAlleles) {
  if (!ctx || !seqs || !offsets || nAlleles == 0 || nAlleles >= (1u << 24)) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_ref_upload: bad arguments");
  if (offsets[nAlleles] - offsets[0] >= (1ull << 29)) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_ref_upload: reference larger than 512 Mbases");
  T1K_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  (void)hipStreamSynchronize(st);
  for (auto &b : ctx->refBufs) if (b.p) (void)t1k_dev_free(b.p);
  ctx->refBufs.clear();
  const int k = ctx->prm.kmer_length;
  const int nCode = ctx->prm.n_base_code & 3;
  const bool dbgPhases = getenv("T1K_DEBUG_PHASES") != nullptr;
  auto tLap = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!dbgPhases) return;
    (void)hipStreamSynchronize(st);
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[t1k] ref_upload %s: %.3f s\n", what, std::chrono::duration<double>(now - tLap).count());
    tLap = now;
  };
  // layout: every allele starts on a 32-base boundary with at least one spare position behind it (the coverage difference array
  // writes its end marker at seqEnd + 1)
  std::vector<uint64_t> alleleOff(nAlleles), srcOff(nAlleles);
  std::vector<uint32_t> alleleLen(nAlleles);
  uint64_t total = 0;
  for (uint32_t a = 0; a < nAlleles; ++a) {
    const uint64_t len = offsets[a + 1] - offsets[a];
    if (len >= (1u << 20)) return t1k_fail(ctx, T1K_ERR_ARG, "allele longer than 2^20 bases");
    alleleOff[a] = total; alleleLen[a] = (uint32_t)len; srcOff[a] = offsets[a] - offsets[0];
    total += (len + 32) / 32 * 32;
  }
  total += 64;
  const uint64_t nWords = total / 32 + 2;
  std::vector<uint32_t> wordAllele(nWords, 0xFFFFFFFFu);
  for (uint32_t a = 0; a < nAlleles; ++a)
    for (uint64_t w = alleleOff[a] / 32; w < (alleleOff[a] + alleleLen[a] + 31) / 32; ++w) wordAllele[w] = a;
  // interior N positions of every allele (SeqSet.hpp:924-928) and "the allele holds an N": a scan of the text on the host threads
  std::vector<uint32_t> sepStart(nAlleles + 1, 0);
  std::vector<uint8_t> alleleHasN(nAlleles, 0);
  std::vector<int32_t> sepPos;
  {
    const unsigned T = std::max(1u, std::min({16u, std::thread::hardware_concurrency(), nAlleles}));
    std::vector<std::vector<int32_t>> part(T);
    std::vector<uint32_t> cnt(nAlleles, 0);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        const uint32_t a0 = (uint32_t)((uint64_t)nAlleles * t / T), a1 = (uint32_t)((uint64_t)nAlleles * (t + 1) / T);
        for (uint32_t a = a0; a < a1; ++a) {
          const char *s = seqs + offsets[a];
          for (uint32_t i = 0; i < alleleLen[a]; ++i)
            if (asciiCode(s[i]) == 4) { part[t].push_back((int32_t)i); ++cnt[a]; alleleHasN[a] = 1; }
        }
      });
    for (auto &x : th) x.join();
    for (uint32_t a = 0; a < nAlleles; ++a) sepStart[a + 1] = sepStart[a] + cnt[a];
    for (auto &p : part) sepPos.insert(sepPos.end(), p.begin(), p.end());
    if (sepPos.empty()) sepPos.push_back(0);
  }
  lap("layout + separator scan (host)");
  ctx->hAlleleOff = alleleOff;
  ctx->hAlleleLen = alleleLen;
  T1kRefDev r{};
  r.nAlleles = nAlleles;
  r.totalBases = total;
  int rc;
  // resident arrays
  void *dBases, *dN, *dExon, *dAlleleOff, *dAlleleLen, *dHasN, *dSepStart, *dSepPos, *dPosted;
  if ((rc = keep(ctx, (nWords + 8) * 8, &dPosted)) || (rc = keep(ctx, (nWords + 8) * 8, &dBases)) || (rc = keep(ctx, (nWords + 8) * 8, &dN)) || (rc = keep(ctx, (nWords + 8) * 8, &dExon))  // (+8: kernels fetch a window as six whole words)
      || (rc = keep(ctx, (size_t)nAlleles * 8, &dAlleleOff)) ||
      (rc = keep(ctx, (size_t)nAlleles * 4, &dAlleleLen)) || (rc = keep(ctx, nAlleles, &dHasN)) || (rc = keep(ctx, (size_t)(nAlleles + 1) * 4, &dSepStart)) ||
      (rc = keep(ctx, sepPos.size() * 4, &dSepPos)))
    return rc;
  // build-time scratch (freed at the end)
  const uint64_t textBytes = offsets[nAlleles] - offsets[0];
  T1kDevBuf bText, bExonB, bSrcOff, bWordAllele, bFlag, bPos, bCode;
  auto freeScratch = [&] { (void)hipStreamSynchronize(st); for (T1kDevBuf *b : {&bText, &bExonB, &bSrcOff, &bWordAllele, &bFlag, &bPos, &bCode}) if (b->p) { (void)t1k_dev_free(b->p); b->p = nullptr; } };
  if ((rc = t1k_ensure(ctx, bText, textBytes + 16)) || (exon && (rc = t1k_ensure(ctx, bExonB, textBytes + 16))) || (rc = t1k_ensure(ctx, bSrcOff, (size_t)nAlleles * 8)) ||
      (rc = t1k_ensure(ctx, bWordAllele, nWords * 4)) || (rc = t1k_ensure(ctx, bFlag, (total + 1) * 4)) || (rc = t1k_ensure(ctx, bPos, (total + 1) * 4)) ||
      (rc = t1k_ensure(ctx, bCode, (total + 1) * 4))) { freeScratch(); return rc; }
#define RU_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { freeScratch(); return t1k_fail(ctx, T1K_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); } } while (0)
  RU_HIP(hipMemcpyAsync(bText.p, seqs + offsets[0], textBytes, hipMemcpyHostToDevice, st));
  if (exon) RU_HIP(hipMemcpyAsync(bExonB.p, exon + offsets[0], textBytes, hipMemcpyHostToDevice, st));
  RU_HIP(hipMemcpyAsync(bSrcOff.p, srcOff.data(), (size_t)nAlleles * 8, hipMemcpyHostToDevice, st));
  RU_HIP(hipMemcpyAsync(bWordAllele.p, wordAllele.data(), nWords * 4, hipMemcpyHostToDevice, st));
  RU_HIP(hipMemcpyAsync(dAlleleOff, alleleOff.data(), (size_t)nAlleles * 8, hipMemcpyHostToDevice, st));
  RU_HIP(hipMemcpyAsync(dAlleleLen, alleleLen.data(), (size_t)nAlleles * 4, hipMemcpyHostToDevice, st));
  RU_HIP(hipMemcpyAsync(dHasN, alleleHasN.data(), nAlleles, hipMemcpyHostToDevice, st));
  RU_HIP(hipMemcpyAsync(dSepStart, sepStart.data(), (size_t)(nAlleles + 1) * 4, hipMemcpyHostToDevice, st));
  RU_HIP(hipMemcpyAsync(dSepPos, sepPos.data(), sepPos.size() * 4, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_ref_pack, dim3((unsigned)((nWords + 255) / 256)), dim3(256), 0, st, (const char *)bText.p, exon ? (const uint8_t *)bExonB.p : nullptr,
                     (const uint64_t *)bSrcOff.p, (const uint64_t *)dAlleleOff, (const uint32_t *)dAlleleLen, (const uint32_t *)bWordAllele.p, nWords, (uint64_t *)dBases,
                     (uint64_t *)dN, (uint64_t *)dExon, nCode);
  // the transposed copy of the bases for the closed-form pass of the chain (T1kRefDev::basesT): blocks of 64 consecutive alleles, as many rows as the
  // longest of them has words + 2 (a window is fetched as whole words up to two behind its last)
  void *dBasesT = nullptr, *dBlockT = nullptr;
  {
    static const bool on = [] { const char *e = getenv("T1K_REF_TRANSPOSE"); return !e || atoi(e) != 0; }();
    const uint32_t nBlocks = (nAlleles + 63) / 64;
    std::vector<uint32_t> blockT(nBlocks);
    uint64_t totalT = 0;
    for (uint32_t b = 0; b < nBlocks; ++b) {
      uint32_t words = 0;
      for (uint32_t a = b * 64; a < std::min(nAlleles, b * 64 + 64); ++a) words = std::max(words, (alleleLen[a] + 31) / 32);
      blockT[b] = (uint32_t)totalT;
      totalT += (uint64_t)(words + 2) * 64;
    }
    if (on && nBlocks && totalT + 512 < (1ull << 32)) {
      if ((rc = keep(ctx, (totalT + 512) * 8, &dBasesT)) || (rc = keep(ctx, (size_t)nBlocks * 4, &dBlockT))) { freeScratch(); return rc; }
      RU_HIP(hipMemcpyAsync(dBlockT, blockT.data(), (size_t)nBlocks * 4, hipMemcpyHostToDevice, st));
      RU_HIP(hipMemsetAsync((char *)dBasesT + totalT * 8, 0, 512 * 8, st));
      hipLaunchKernelGGL(k_ref_transpose, dim3((unsigned)((totalT + 255) / 256)), dim3(256), 0, st, (const uint64_t *)dBases, (const uint64_t *)dAlleleOff, (const uint32_t *)dAlleleLen,
                         (const uint32_t *)dBlockT, nBlocks, nAlleles, totalT, (uint64_t *)dBasesT);
      RU_HIP(hipStreamSynchronize(st));  // (blockT is a local: the copy must have left it)
    }
  }
  lap("text upload + pack");
  // inserted windows
  hipLaunchKernelGGL(k_ref_codes, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const uint64_t *)dBases, (const uint64_t *)dN, (const uint64_t *)dAlleleOff,
                     (const uint32_t *)dAlleleLen, (const uint32_t *)bWordAllele.p, total, k, (uint32_t *)bFlag.p, (uint32_t *)bCode.p);
  RU_HIP(hipMemsetAsync(dPosted, 0, (nWords + 8) * 8, st));
  hipLaunchKernelGGL(k_ref_posted, dim3((unsigned)((nWords + 255) / 256)), dim3(256), 0, st, (const uint32_t *)bFlag.p, total, k, nWords, (uint64_t *)dPosted);
  if ((rc = t1k_inclusive_sum(ctx, (const uint32_t *)bFlag.p, (uint32_t *)bPos.p, (uint32_t)total))) { freeScratch(); return rc; }
  uint32_t M = 0;
  RU_HIP(hipMemcpyAsync(&M, (uint32_t *)bPos.p + (total - 1), 4, hipMemcpyDeviceToHost, st));
  RU_HIP(hipStreamSynchronize(st));
  const uint64_t nKeys = 1ull << (2 * k);
  const uint32_t Mx = std::max(M, 1u);
  T1kDevBuf bKey, bKeyS, bVal, bValS, bNeed, bRowOf, bRowCode;
  auto freeScratch2 = [&] { freeScratch(); for (T1kDevBuf *b : {&bKey, &bKeyS, &bVal, &bValS, &bNeed, &bRowOf, &bRowCode}) if (b->p) { (void)t1k_dev_free(b->p); b->p = nullptr; } };
#undef RU_HIP
#define RU_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { freeScratch2(); return t1k_fail(ctx, T1K_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); } } while (0)
  if ((rc = t1k_ensure(ctx, bKey, (size_t)Mx * 8)) || (rc = t1k_ensure(ctx, bKeyS, (size_t)Mx * 8)) || (rc = t1k_ensure(ctx, bVal, (size_t)Mx * 4)) ||
      (rc = t1k_ensure(ctx, bValS, (size_t)Mx * 4))) { freeScratch2(); return rc; }
  void *dPost, *dPostAllele, *dKStart, *dHas, *dMulti, *dHasPre, *dDirIdx;
  const int kp = std::max(1, k - 2);
  const size_t nPre = (size_t)1 << (2 * kp);
  if ((rc = keep(ctx, (size_t)Mx * sizeof(T1kPosting), &dPost)) || (rc = keep(ctx, (size_t)Mx * 4, &dPostAllele)) || (rc = keep(ctx, (nKeys + 2) * 4, &dKStart)) ||
      (rc = keep(ctx, (nKeys + 31) / 32 * 4, &dHas)) || (rc = keep(ctx, (nKeys + 31) / 32 * 4, &dMulti)) || (rc = keep(ctx, (nPre + 31) / 32 * 4, &dHasPre)) ||
      (rc = keep(ctx, nKeys * 4, &dDirIdx))) { freeScratch2(); return rc; }
  RU_HIP(hipMemsetAsync(dKStart, 0, (nKeys + 2) * 4, st));
  RU_HIP(hipMemsetAsync(dHas, 0, (nKeys + 31) / 32 * 4, st));
  RU_HIP(hipMemsetAsync(dMulti, 0, (nKeys + 31) / 32 * 4, st));
  RU_HIP(hipMemsetAsync(dHasPre, 0, (nPre + 31) / 32 * 4, st));
  if (M) {
    hipLaunchKernelGGL(k_ref_compact, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const uint32_t *)bFlag.p, (const uint32_t *)bPos.p, (const uint32_t *)bCode.p,
                       (unsigned long long *)bKey.p, (uint32_t *)bVal.p, total);
    if ((rc = t1k_sort_pairs(ctx, (const unsigned long long *)bKey.p, (unsigned long long *)bKeyS.p, (const uint32_t *)bVal.p, (uint32_t *)bValS.p, M, 2 * k))) { freeScratch2(); return rc; }
    // bucket counts first; the exclusive scan below turns them into kStart[c] = start of list c, kStart[4^k] = M
    hipLaunchKernelGGL(k_ref_postings, dim3((M + 255) / 256), dim3(256), 0, st, (const unsigned long long *)bKeyS.p, (const uint32_t *)bValS.p, (const uint64_t *)dAlleleOff,
                       (const uint32_t *)bWordAllele.p, k, (T1kPosting *)dPost, (uint32_t *)dPostAllele, (uint32_t *)dKStart, M);
  }
  lap("window codes + sort + postings");
  if ((rc = t1k_exclusive_sum32(ctx, (const uint32_t *)dKStart, (uint32_t *)dKStart, nKeys + 1))) { freeScratch2(); return rc; }
  // bitmaps + chunk directory
  if ((rc = t1k_ensure(ctx, bNeed, (nKeys + 1) * 4)) || (rc = t1k_ensure(ctx, bRowOf, (nKeys + 1) * 4))) { freeScratch2(); return rc; }
  hipLaunchKernelGGL(k_ref_flags, dim3((unsigned)((nKeys + 255) / 256)), dim3(256), 0, st, (const uint32_t *)dKStart, (const uint32_t *)dPostAllele, nKeys, k, (uint32_t *)dHas,
                     (uint32_t *)dMulti, (uint32_t *)dHasPre, (uint32_t *)bNeed.p);
  if ((rc = t1k_inclusive_sum_n(ctx, (const uint32_t *)bNeed.p, (uint32_t *)bRowOf.p, nKeys))) { freeScratch2(); return rc; }
  uint32_t rows = 0;
  RU_HIP(hipMemcpyAsync(&rows, (uint32_t *)bRowOf.p + (nKeys - 1), 4, hipMemcpyDeviceToHost, st));
  RU_HIP(hipStreamSynchronize(st));
  const uint32_t stride = (nAlleles + T1K_SEED_CHUNK - 1) / T1K_SEED_CHUNK + 1;
  void *dDir;
  if ((rc = keep(ctx, (size_t)std::max(rows, 1u) * stride * 4, &dDir)) || (rc = t1k_ensure(ctx, bRowCode, (size_t)std::max(rows, 1u) * 4))) { freeScratch2(); return rc; }
  hipLaunchKernelGGL(k_ref_diridx, dim3((unsigned)((nKeys + 255) / 256)), dim3(256), 0, st, (const uint32_t *)bNeed.p, (const uint32_t *)bRowOf.p, (uint32_t *)dDirIdx,
                     (uint32_t *)bRowCode.p, nKeys);
  if (rows) {
    const uint64_t cells = (uint64_t)rows * stride;
    hipLaunchKernelGGL(k_ref_dir, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, (const uint32_t *)bRowCode.p, (const uint32_t *)dKStart, (const uint32_t *)dPostAllele, stride,
                       cells, (uint32_t *)dDir);
  } else RU_HIP(hipMemsetAsync(dDir, 0, (size_t)stride * 4, st));
  const uint32_t maskWords = std::max(1u, (stride - 1 + 63) / 64);
  void *dDirMask;
  if ((rc = keep(ctx, (size_t)std::max(rows, 1u) * maskWords * 8, &dDirMask))) { freeScratch2(); return rc; }
  if (rows) hipLaunchKernelGGL(k_ref_dirmask, dim3((unsigned)(((uint64_t)rows * maskWords + 255) / 256)), dim3(256), 0, st, (const uint32_t *)dDir, stride, rows, maskWords, (unsigned long long *)dDirMask);
  else RU_HIP(hipMemsetAsync(dDirMask, 0, (size_t)maskWords * 8, st));
  // coverage arrays
  void *dCov;
  r.covStride = total + 2;
  if ((rc = keep(ctx, 3 * r.covStride * sizeof(int32_t), &dCov))) { freeScratch2(); return rc; }
  RU_HIP(hipMemsetAsync(dCov, 0, 3 * r.covStride * sizeof(int32_t), st));
  RU_HIP(hipStreamSynchronize(st));
  freeScratch2();
#undef RU_HIP
  r.basesT = (const uint64_t *)dBasesT; r.blockT = (const uint32_t *)dBlockT;
  r.bases = (const uint64_t *)dBases; r.nmask = (const uint64_t *)dN; r.exon = (const uint64_t *)dExon; r.posted = (const uint64_t *)dPosted;
  r.alleleOff = (const uint64_t *)dAlleleOff; r.alleleLen = (const uint32_t *)dAlleleLen; r.alleleHasN = (const uint8_t *)dHasN;
  r.anyN = 0;
  for (uint8_t h : alleleHasN) r.anyN |= h;
  r.sepStart = (const uint32_t *)dSepStart; r.sepPos = (const int32_t *)dSepPos;
  r.kStart = (const uint32_t *)dKStart; r.kHas = (const uint32_t *)dHas; r.kMulti = (const uint32_t *)dMulti; r.kHasPre = (const uint32_t *)dHasPre;
  r.kDirIdx = (const uint32_t *)dDirIdx; r.kDir = (const uint32_t *)dDir; r.kDirStride = stride;
  r.kDirMask = (const unsigned long long *)dDirMask; r.kDirMaskWords = maskWords;
  r.kPost = (const T1kPosting *)dPost; r.kPostAllele = (const uint32_t *)dPostAllele;
  r.covDiff = (int32_t *)dCov;
  ctx->ref = r;
  ctx->covFullLen = 0; ctx->covFullDirty = false;
  lap("bucket starts + bitmaps + directory + coverage arrays");
  return T1K_OK;
}
