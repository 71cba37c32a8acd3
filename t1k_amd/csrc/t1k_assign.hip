// t1k_amd/csrc/t1k_assign.hip -- gfx950 kernels for SeqSet::AssignRead (reference SeqSet.hpp:2119-2303) over a batch
// of read-ends resident in HBM.  Stages (one kernel each, all integer / HBM- and LDS-bound, no MFMA):
//
//   k_pack_reads   ASCII -> 2-bit words + N masks, forward and reverse complement           (SeqSet.hpp:2103-2114)
//   k_seed_chain   one 256-thread workgroup per read-end: rolling 11-mers + direct-address look-up with the >=100
//                  skip rule (GetHitsFromRead 1071-1229), LDS histogram over alleles -> counting-sort of the hits by
//                  (strand, allele) (SortHits 1558-1590), one lane per (strand, allele) group: diagonal runs, dominant
//                  diagonal, LIS, seed-chain match count (GetOverlapsFromHits 1232-1556, GetOverlapsFromRead
//                  1665-1848), strand vote (1619-1648)
//   k_extend       one lane per candidate: similarity / low-complexity filter (1838-1845, 458-485, 1894-1908),
//                  separator tests (2163-2169), ExtendOverlap (1994-2100)
//   k_select       one workgroup per read-end: sort by _overlap::operator< (103-127), the onlyConsiderClip latch
//                  (2156-2186) evaluated in parallel, near-best flags (2192-2200)
//   k_fullalign    one lane per kept overlap: near-best full alignment -> relaxedMatchCnt + base coverage as a
//                  difference array (2188-2285); alignments that need a real DP traceback go to a queue (k_fullalign_slow)
//   k_truncate     >1000 overlaps: re-sort and cut at similarity < best - 0.1 (2290-2298)
#include "t1k_dev.h"
#include "t1k_launch.h"

#define WG 256
#define TILE_ALLELES 16384          // LDS histogram tile (u32 per allele)
#define GROUP_FAST_MAXLEN 320       // read-offset bitmask width of the single-diagonal fast path
#define THREAD_CAP 192              // hits per group handled with per-thread scratch; larger groups go to lane 0
#define BIG_CAP 16384
#define GA_BIG_MAX 2048
#define GA_SCRATCH_INTS (6 * (GA_BIG_MAX + 4))  // row arrays of t1k_ga_general for sequences up to 2048 bases
#define GA_T_MAX 512
#define GA_THREAD_INTS (6 * (GA_T_MAX + 4))

#define THREAD_SCRATCH_U32 (3 * THREAD_CAP + GA_THREAD_INTS)

enum { ERR_HITCAP = 1, ERR_STAGECAP = 2, ERR_CANDCAP = 4, ERR_BIGGROUP = 8, ERR_OVLCAP = 16, ERR_SORTCAP = 32, ERR_SLOWCAP = 64, ERR_ROWCAP = 128 };

// ------------------------------------------------------------------------------------------------------------------
// pack
// ------------------------------------------------------------------------------------------------------------------
__global__ void k_pack_reads(const char *ascii, const uint64_t *offs, uint32_t n, int S, uint64_t *bases, uint64_t *nmask, uint16_t *lens) {
  uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t total = (uint64_t)n * S;
  if (gid >= total) return;
  uint32_t re = (uint32_t)(gid / S);
  int w = (int)(gid % S);
  uint64_t o = offs[re];
  int len = (int)(offs[re + 1] - o);
  if (w == 0) lens[re] = (uint16_t)len;
  uint64_t fb = 0, fn = 0, rb = 0, rn = 0;
  for (int q = 0; q < 32; ++q) {
    int i = w * 32 + q;
    if (i >= len) break;
    char c = ascii[o + i];
    int code = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;
    if (code == 4) fn |= 1ull << (2 * q); else fb |= (uint64_t)code << (2 * q);
    char d = ascii[o + len - 1 - i];
    int dc = d == 'A' ? 3 : d == 'C' ? 2 : d == 'G' ? 1 : d == 'T' ? 0 : 4;
    if (dc == 4) rn |= 1ull << (2 * q); else rb |= (uint64_t)dc << (2 * q);
  }
  uint64_t base = (uint64_t)re * 2 * S;
  bases[base + w] = fb; nmask[base + w] = fn;
  bases[base + S + w] = rb; nmask[base + S + w] = rn;
}

// ------------------------------------------------------------------------------------------------------------------
// group -> candidate overlaps
// ------------------------------------------------------------------------------------------------------------------
struct ReadCtx {
  const uint64_t *rb, *rn;   // strand-specific read words
  int len;
  const uint64_t *gb, *gn;   // reference words
  int64_t goff;              // allele global base offset
  int alleleLen;
};

struct CandOut {  // packed into the group's own hit segment: 3 u32 per candidate
  uint32_t *dst;
  int n;
  __device__ void push(int rs, int re, int ss, int se, int m0, int m) {
    dst[3 * n + 0] = (uint32_t)rs | ((uint32_t)re << 12);
    dst[3 * n + 1] = (uint32_t)ss | ((uint32_t)m0 << 20);
    dst[3 * n + 2] = (uint32_t)se | ((uint32_t)m << 20);
    ++n;
  }
};

// seed-chain match count of one gap (SeqSet.hpp:1710-1752 / 1794-1824)
__device__ inline int gapMatches(const ReadCtx &c, int ra, int ga, int lp, int lt, int *gaScratch, int gaMax, unsigned long long *dpCounter, unsigned long long *errFlags) {
  if (lp == lt) return t1k_ga_matches_window(c.rb, c.rn, ra, c.gb, c.gn, c.goff + ga, lp, dpCounter);
  if (lt == 0 || lp == 0) return 0;
  if (dpCounter) atomicAdd(dpCounter, 1ull);
  T1kSeqView T{c.gb, c.gn, c.goff + ga}, P{c.rb, c.rn, ra};
  int nm = 0;
  if (lt > gaMax) { atomicOr(errFlags, (unsigned long long)ERR_BIGGROUP); return 0; }
  t1k_ga_general(T, lt, P, lp, gaScratch, nullptr, &nm);
  return nm;
}

// Single-diagonal group: every hit has the same (readOffset - alleleOffset).  The LIS is the identity, both hit
// lengths are equal, and the chain's match count is 2*(covered) + 2*sum of per-gap alignment matches.
__device__ inline bool groupFastPath(const uint32_t *h, int n, const ReadCtx &c, int k, int hitLenRequired, CandOut &out, unsigned long long *dpCounter) {
  uint64_t M[GROUP_FAST_MAXLEN / 64];
#pragma unroll
  for (int i = 0; i < GROUP_FAST_MAXLEN / 64; ++i) M[i] = 0;
  int diag = 0;
  for (int i = 0; i < n; ++i) {
    uint32_t x = h[i];
    int a = (int)(x & 0xFFF), b = (int)(x >> 12);
    if (i == 0) diag = a - b; else if (a - b != diag) return false;
    if (a >= GROUP_FAST_MAXLEN) return false;
#pragma unroll
    for (int w = 0; w < GROUP_FAST_MAXLEN / 64; ++w)
      if ((a >> 6) == w) M[w] |= 1ull << (a & 63);
  }
  // walk the set bits in ascending read offset
  int first = -1, prev = -1, cov = 0, gapMatch = 0;
#pragma unroll
  for (int w = 0; w < GROUP_FAST_MAXLEN / 64; ++w) {
    uint64_t m = M[w];
    while (m) {
      int a = w * 64 + __ffsll((long long)m) - 1;
      m &= m - 1;
      if (first < 0) { first = a; cov = k; }
      else {
        int d = a - prev;
        if (d < k) cov += d;  // k-mers overlap on the read (SeqSet.hpp:1704-1707)
        else {
          cov += k;
          int g = d - k;
          if (g > 0) gapMatch += t1k_ga_matches_window(c.rb, c.rn, prev + k, c.gb, c.gn, c.goff + (prev + k - diag), g, dpCounter);
        }
      }
      prev = a;
    }
  }
  if (n * k < hitLenRequired) return true;  // cannot happen for n >= 3, k = 11
  if (cov < hitLenRequired) return true;    // GetTotalHitLengthOnRead/OnSeq (1512-1522)
  out.push(first, prev + k - 1, first - diag, prev - diag + k - 1, 2 * cov, 2 * cov + 2 * gapMatch);
  return true;
}

__device__ __forceinline__ bool hitKeyLess(uint32_t x, uint32_t y) {  // (diag, alleleOff, readOff): CompSortHitCoordDiff (266-274)
  int cx = (int)(x & 0xFFF) - (int)(x >> 12), cy = (int)(y & 0xFFF) - (int)(y >> 12);
  if (cx != cy) return cx < cy;
  return x < y;
}

// General group (several diagonals): restates GetOverlapsFromHits 1338-1551 and the chain walk 1697-1833.
// A[n] sorted copy of the hits, B[n] concordant hits, C[n] packs top (low 16) / link (high 16) of the LIS.
__device__ inline void groupGeneral(const uint32_t *h, int n, const ReadCtx &c, int k, int radius, int hitLenRequired, uint32_t *A, uint32_t *B,
                                     uint32_t *C, int *gaScratch, int gaMax, CandOut &out, unsigned long long *dpCounter, unsigned long long *errFlags) {
  // insertion sort into A
  for (int i = 0; i < n; ++i) {
    uint32_t x = h[i];
    int j = i - 1;
    while (j >= 0 && hitKeyLess(x, A[j])) { A[j + 1] = A[j]; --j; }
    A[j + 1] = x;
  }
  for (int s = 0; s < n;) {
    auto diagOf = [](uint32_t x) { return (int)(x & 0xFFF) - (int)(x >> 12); };
    int curDiff = diagOf(A[s]), curCnt = 1, domCnt = 0, dominant = 0;
    int e = s + 1;
    for (; e < n; ++e) {
      int d = diagOf(A[e]) - diagOf(A[e - 1]);
      if (d < 0) d = -d;
      if (d > radius) break;
      if (d == 0) ++curCnt;
      else {
        if (curCnt > domCnt) { dominant = curDiff; domCnt = curCnt; }
        curDiff = diagOf(A[e]); curCnt = 1;
      }
    }
    if (curCnt > domCnt) dominant = curDiff;
    if (e - s < 3 || (e - s) * k < hitLenRequired) { s = e; continue; }
    // nearest-to-dominant filter per read offset (1437-1456)
    int m = 0;
    for (int q = s; q < e; ++q) {
      int a = (int)(A[q] & 0xFFF);
      int dq = diagOf(A[q]) - dominant; if (dq < 0) dq = -dq;
      bool keep = true;
      for (int r = s; r < e; ++r) {
        if ((int)(A[r] & 0xFFF) != a) continue;
        int dr = diagOf(A[r]) - dominant; if (dr < 0) dr = -dr;
        if (dr < dq) { keep = false; break; }
      }
      if (keep) {  // insertion by (alleleOff, readOff) == packed value order (CompSortPairBInc)
        uint32_t x = A[q];
        int j = m - 1;
        while (j >= 0 && x < B[j]) { B[j + 1] = B[j]; --j; }
        B[j + 1] = x;
        ++m;
      }
    }
    // LIS over read offsets (352-436); C[i] = top | link << 16, link 0xFFFF = none
    int ret = 1;
    C[0] = 0 | (0xFFFFu << 16);
    auto topOf = [&](int i) { return (int)(C[i] & 0xFFFF); };
    auto setTop = [&](int i, int v) { C[i] = (C[i] & 0xFFFF0000u) | (uint32_t)v; };
    auto setLink = [&](int i, int v) { C[i] = (C[i] & 0xFFFFu) | ((uint32_t)(v & 0xFFFF) << 16); };
    auto linkOf = [&](int i) { return (int)(C[i] >> 16); };
    auto aOf = [&](int i) { return (int)(B[i] & 0xFFF); };
    for (int i = 1; i < m; ++i) C[i] = 0xFFFFu << 16;
    for (int i = 1; i < m; ++i) {
      int tag;
      if (aOf(topOf(ret - 1)) <= aOf(i)) tag = ret - 1;
      else {
        int l = 0, r = ret - 1;
        tag = -2;
        while (l <= r) {
          int mid = (l + r) / 2;
          if (aOf(i) == aOf(topOf(mid))) { tag = mid; break; }
          if (aOf(i) < aOf(topOf(mid))) r = mid - 1; else l = mid + 1;
        }
        if (tag == -2) tag = l - 1;
      }
      if (tag == -1) { setTop(0, i); setLink(i, 0xFFFF); }
      else if (aOf(i) > aOf(topOf(tag))) {
        if (tag == ret - 1) { setTop(ret, i); ++ret; setLink(i, topOf(tag)); }
        else if (aOf(i) < aOf(topOf(tag + 1))) { setTop(tag + 1, i); setLink(i, topOf(tag)); }
      }
    }
    // retrieve the chain into A[s .. s+ret) (the run's slice of A is dead now), then drop repeated allele offsets
    {
      int kx = topOf(ret - 1);
      for (int i = ret - 1; i >= 0; --i) { A[s + i] = B[kx]; kx = linkOf(kx); }
      int w = 1;
      for (int i = 1; i < ret; ++i) {
        if ((A[s + i] >> 12) == (A[s + w - 1] >> 12)) continue;
        A[s + w] = A[s + i];
        ++w;
      }
      ret = w;
    }
    if (ret * k < hitLenRequired) { s = e; continue; }
    // hit lengths on read and on allele (1032-1069)
    int lenR = 0, lenS = 0;
    for (int i = 0; i < ret;) {
      int j = i + 1;
      for (; j < ret; ++j) if ((int)(A[s + j] & 0xFFF) > (int)(A[s + j - 1] & 0xFFF) + k - 1) break;
      lenR += (int)(A[s + j - 1] & 0xFFF) - (int)(A[s + i] & 0xFFF) + k;
      i = j;
    }
    for (int i = 0; i < ret;) {
      int j = i + 1;
      for (; j < ret; ++j) if ((int)(A[s + j] >> 12) > (int)(A[s + j - 1] >> 12) + k - 1) break;
      lenS += (int)(A[s + j - 1] >> 12) - (int)(A[s + i] >> 12) + k;
      i = j;
    }
    if (lenR < hitLenRequired || lenS < hitLenRequired) { s = e; continue; }
    // seed-chain match count (1697-1833)
    int matchCnt = 2 * k;
    for (int i = 1; i < ret; ++i) {
      int pa = (int)(A[s + i - 1] & 0xFFF), pb = (int)(A[s + i - 1] >> 12), qa = (int)(A[s + i] & 0xFFF), qb = (int)(A[s + i] >> 12);
      bool sameDiag = (pb - pa) == (qb - qa);
      bool readOv = pa + k - 1 >= qa, seqOv = pb + k - 1 >= qb;
      if (sameDiag) {
        if (readOv) matchCnt += 2 * (qa - pa);
        else matchCnt += 2 * k + 2 * gapMatches(c, pa + k, pb + k, qa - (pa + k), qb - (pb + k), gaScratch, gaMax, dpCounter, errFlags);
      } else {
        if (readOv && !seqOv) matchCnt += 2 * (qa - pa);
        else if (!readOv && seqOv) matchCnt += 2 * (qb - pb);
        else if (readOv && seqOv) matchCnt += 2 * ((qa - pa) < (qb - pb) ? (qa - pa) : (qb - pb));
        else matchCnt += 2 * k + 2 * gapMatches(c, pa + k, pb + k, qa - (pa + k), qb - (pb + k), gaScratch, gaMax, dpCounter, errFlags);
      }
    }
    int rs = (int)(A[s] & 0xFFF), ss = (int)(A[s] >> 12);
    int re = (int)(A[s + ret - 1] & 0xFFF) + k - 1, se = (int)(A[s + ret - 1] >> 12) + k - 1;
    out.push(rs, re, ss, se, 2 * lenR, matchCnt);
    s = e;
  }
}

// key of the strand vote: _overlap::operator< with similarity == 0 (SeqSet.hpp:103-127, 1623-1627); smaller = better
struct VoteKey {
  uint64_t hi, lo;
  __device__ bool operator<(const VoteKey &o) const { return hi != o.hi ? hi < o.hi : lo < o.lo; }
};
__device__ __forceinline__ VoteKey voteKey(int matchCnt0, int rs, int re, uint32_t allele, int strandPlus, int ss, int se) {
  VoteKey k;
  k.hi = ((uint64_t)(4095 - matchCnt0) << 40) | ((uint64_t)(4095 - (re - rs)) << 26) | ((uint64_t)allele << 1) | (uint64_t)strandPlus;
  k.lo = ((uint64_t)rs << 52) | ((uint64_t)re << 40) | ((uint64_t)ss << 20) | (uint64_t)se;
  return k;
}

__device__ __forceinline__ uint32_t blockScanExclusive(uint32_t v, uint32_t *warpSums, uint32_t *total) {
  // 256 threads = 4 wavefronts of 64
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) warpSums[wave] = x;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wave; ++w) base += warpSums[w];
  uint32_t tot = warpSums[0] + warpSums[1] + warpSums[2] + warpSums[3];
  __syncthreads();
  *total = tot;
  return base + x - v;
}

__global__ __launch_bounds__(WG) void k_seed_chain(AssignArgs P) {
  extern __shared__ uint32_t lds[];
  const int k = P.k;
  const int maxK = 2 * (P.reads.S * 32);  // >= 2 * (len - k + 1)
  uint32_t *hist = lds;                             // [TILE_ALLELES]
  uint32_t *ukCode = hist + TILE_ALLELES;           // [maxK]  code | valid << 31
  uint32_t *ukStart = ukCode + maxK;                // [maxK]
  uint32_t *ukLen = ukStart + maxK;                 // [maxK]
  uint16_t *usedQ = (uint16_t *)(ukLen + maxK);     // [maxK]
  __shared__ uint32_t warpSums[4];
  __shared__ uint32_t sUsed[2];      // used k-mers of pass 0 (+) and pass 1 (-)
  __shared__ uint32_t sStageCount, sBase;
  __shared__ uint64_t sVoteHi[WG], sVoteLo[WG];

  const int tid = threadIdx.x;
  const uint32_t kmask = (1u << (2 * k)) - 1;
  uint32_t *myHits = P.wgHits + (uint64_t)blockIdx.x * P.hitCap;
  uint32_t *myGroups = P.wgGroups + (uint64_t)blockIdx.x * TILE_ALLELES * 3;
  T1kCand *myStage = P.wgStage + (uint64_t)blockIdx.x * P.stageCap;
  uint32_t *myThread = P.wgThread + ((uint64_t)blockIdx.x * WG + tid) * THREAD_SCRATCH_U32;
  uint32_t *myBig = P.wgBig + (uint64_t)blockIdx.x * (3 * BIG_CAP + GA_SCRATCH_INTS);
  const int nTiles = (int)((P.ref.nAlleles + TILE_ALLELES - 1) / TILE_ALLELES);

  for (uint32_t re = blockIdx.x; re < P.reads.nReadEnds; re += gridDim.x) {
    const int len = P.reads.len[re];
    const int S = P.reads.S;
    const uint64_t *rbase = P.reads.bases + (uint64_t)re * 2 * S;
    const uint64_t *rnm = P.reads.nmask + (uint64_t)re * 2 * S;
    if (tid == 0) { sStageCount = 0; }
    __syncthreads();
    if (len < k) {  // GetOverlapsFromRead returns -1 (SeqSet.hpp:1598-1599)
      if (tid == 0) { P.candStart[re] = 0; P.candCount[re] = 0; }
      __syncthreads();
      continue;
    }
    const int nk = len - k + 1;
    // ---- 1. k-mer codes and posting-list bounds for both strands -------------------------------------------------
    for (int q = tid; q < 2 * nk; q += WG) {
      int pass = q / nk, p = q - pass * nk;
      const uint64_t *b = rbase + pass * S, *nm = rnm + pass * S;
      uint32_t code = (uint32_t)t1k_get32(b, p) & kmask;
      bool valid = ((uint32_t)t1k_get32(nm, p) & kmask) == 0;
      uint32_t st = 0, ln = 0;
      if (valid) { st = P.ref.kStart[code]; ln = P.ref.kStart[code + 1] - st; }
      ukCode[q] = code | (valid ? 0x80000000u : 0);
      ukStart[q] = st; ukLen[q] = ln;
    }
    __syncthreads();
    // ---- 2. the sequential look-up rule (SeqSet.hpp:1098-1153, 1165-1226) -----------------------------
    if (tid == 0) {
      uint32_t prev = 0;  // prevKmerCode starts at code 0 and is carried from the + strand into the - strand
      uint32_t nUsed = 0;
      unsigned long long lookups = 0, postings = 0;
      for (int pass = 0; pass < 2; ++pass) {
        int skipCnt = 0;
        uint32_t begin = nUsed;
        for (int p = 0; p < nk; ++p) {
          int q = pass * nk + p;
          uint32_t code = ukCode[q] & 0x7FFFFFFFu;
          if (p == 0 || code != prev) {
            uint32_t size = ukLen[q];
            ++lookups;
            if (size >= 100 && p != 0 && p != nk - 1 && skipCnt < k / 2) { ++skipCnt; continue; }
            skipCnt = 0;
            if (size) { usedQ[nUsed++] = (uint16_t)q; postings += size; }
          }
          prev = code;
        }
        sUsed[pass] = nUsed - begin;
      }
      atomicAdd(&P.counters[3], lookups);
      atomicAdd(&P.counters[4], postings);
    }
    __syncthreads();
    const uint32_t nUsedPlus = sUsed[0], nUsedMinus = sUsed[1];
    VoteKey best; best.hi = ~0ull; best.lo = ~0ull;
    // ---- 3. per strand ('-' first, SortHits 1577-1583) and allele tile ---------------------------------------------
    for (int sp = 0; sp < 2; ++sp) {
      const int pass = sp == 0 ? 1 : 0;  // pass 1 = reverse complement = strand -1
      const uint32_t uBegin = pass == 0 ? 0 : nUsedPlus;
      const uint32_t uCount = pass == 0 ? nUsedPlus : nUsedMinus;
      if (uCount == 0) continue;
      for (int tile = 0; tile < nTiles; ++tile) {
        const uint32_t a0 = (uint32_t)tile * TILE_ALLELES;
        const uint32_t a1 = min(a0 + TILE_ALLELES, P.ref.nAlleles);
        for (uint32_t i = tid; i < TILE_ALLELES; i += WG) hist[i] = 0;
        __syncthreads();
        // count
        for (uint32_t u = 0; u < uCount; ++u) {
          int q = usedQ[uBegin + u];
          uint32_t st = ukStart[q], ln = ukLen[q];
          for (uint32_t x = tid; x < ln; x += WG) {
            uint32_t al = P.ref.kPost[st + x].allele;
            if (al >= a0 && al < a1) atomicAdd(&hist[al - a0], 1u);
          }
        }
        __syncthreads();
        // scan: groups with >= 3 hits (refMinHitRequired, SeqSet.hpp:1253, 1314) get a slice of the hit arena
        const int EPT = TILE_ALLELES / WG;
        uint32_t hSum = 0, gSum = 0;
        for (int i = 0; i < EPT; ++i) {
          uint32_t c = hist[tid * EPT + i];
          if (c >= 3) { hSum += c; ++gSum; }
        }
        uint32_t hTot, gTot;
        uint32_t hOff = blockScanExclusive(hSum, warpSums, &hTot);
        uint32_t gOff = blockScanExclusive(gSum, warpSums, &gTot);
        if (hTot > P.hitCap) {
          if (tid == 0) atomicOr(&P.counters[2], (unsigned long long)ERR_HITCAP);
          __syncthreads();
          continue;
        }
        for (int i = 0; i < EPT; ++i) {
          uint32_t idx = tid * EPT + i;
          uint32_t c = hist[idx];
          if (c >= 3) {
            myGroups[gOff * 3 + 0] = a0 + idx;
            myGroups[gOff * 3 + 1] = hOff;
            myGroups[gOff * 3 + 2] = c;
            hist[idx] = hOff;
            hOff += c; ++gOff;
          } else hist[idx] = 0xFFFFFFFFu;
        }
        if (tid == 0) { atomicAdd(&P.counters[5], (unsigned long long)hTot); atomicAdd(&P.counters[6], (unsigned long long)gTot); }
        __syncthreads();
        // scatter the hits of surviving groups: packed (alleleOffset << 12 | readOffset)
        for (uint32_t u = 0; u < uCount; ++u) {
          int q = usedQ[uBegin + u];
          uint32_t st = ukStart[q], ln = ukLen[q];
          uint32_t rOff = (uint32_t)(q - pass * nk);
          for (uint32_t x = tid; x < ln; x += WG) {
            T1kPosting pst = P.ref.kPost[st + x];
            if (pst.allele >= a0 && pst.allele < a1 && hist[pst.allele - a0] != 0xFFFFFFFFu) {
              uint32_t pos = atomicAdd(&hist[pst.allele - a0], 1u);
              myHits[pos] = (pst.offset << 12) | rOff;
            }
          }
        }
        __threadfence_block();
        __syncthreads();
        // chain: one lane per (strand, allele) group; candidates are packed back into the group's hit slice
        for (uint32_t g = tid; g < gTot; g += WG) {
          uint32_t allele = myGroups[g * 3 + 0], hs = myGroups[g * 3 + 1], n = myGroups[g * 3 + 2];
          ReadCtx c{rbase + pass * S, rnm + pass * S, len, P.ref.bases, P.ref.nmask, (int64_t)P.ref.alleleOff[allele], (int)P.ref.alleleLen[allele]};
          CandOut out{myHits + hs, 0};
          bool done = false;
          if (len <= GROUP_FAST_MAXLEN) done = groupFastPath(myHits + hs, (int)n, c, k, P.hitLenRequired, out, &P.counters[7]);
          if (!done) {
            if (n <= THREAD_CAP) {
              groupGeneral(myHits + hs, (int)n, c, k, P.radius, P.hitLenRequired, myThread, myThread + THREAD_CAP, myThread + 2 * THREAD_CAP,
                           (int *)(myThread + 3 * THREAD_CAP), GA_T_MAX, out, &P.counters[7], &P.counters[2]);
            } else { myGroups[g * 3 + 2] = n | 0x80000000u; continue; }  // deferred to lane 0 below
          }
          myGroups[g * 3 + 2] = (uint32_t)out.n;
        }
        __syncthreads();
        if (tid == 0) {
          for (uint32_t g = 0; g < gTot; ++g) {
            uint32_t n = myGroups[g * 3 + 2];
            if (!(n & 0x80000000u)) continue;
            n &= 0x7FFFFFFFu;
            uint32_t allele = myGroups[g * 3 + 0], hs = myGroups[g * 3 + 1];
            if (n > BIG_CAP) { atomicOr(&P.counters[2], (unsigned long long)ERR_BIGGROUP); myGroups[g * 3 + 2] = 0; continue; }
            ReadCtx c{rbase + pass * S, rnm + pass * S, len, P.ref.bases, P.ref.nmask, (int64_t)P.ref.alleleOff[allele], (int)P.ref.alleleLen[allele]};
            CandOut out{myHits + hs, 0};
            groupGeneral(myHits + hs, (int)n, c, k, P.radius, P.hitLenRequired, myBig, myBig + BIG_CAP, myBig + 2 * BIG_CAP, (int *)(myBig + 3 * BIG_CAP), GA_BIG_MAX, out,
                         &P.counters[7], &P.counters[2]);
            myGroups[g * 3 + 2] = (uint32_t)out.n;
          }
        }
        __syncthreads();
        // compact the candidates of this (strand, tile) into the per-read-end staging list, in group order
        for (uint32_t g0 = 0; g0 < gTot; g0 += WG) {
          uint32_t g = g0 + tid;
          uint32_t nc = g < gTot ? myGroups[g * 3 + 2] : 0;
          uint32_t tot;
          uint32_t off = blockScanExclusive(nc, warpSums, &tot);
          uint32_t base = sStageCount;
          if (base + tot > P.stageCap) {
            if (tid == 0) atomicOr(&P.counters[2], (unsigned long long)ERR_STAGECAP);
            __syncthreads();
            break;
          }
          if (nc) {
            uint32_t allele = myGroups[g * 3 + 0], hs = myGroups[g * 3 + 1];
            for (uint32_t i = 0; i < nc; ++i) {
              uint32_t w0 = myHits[hs + 3 * i], w1 = myHits[hs + 3 * i + 1], w2 = myHits[hs + 3 * i + 2];
              T1kCand cd;
              int rs = (int)(w0 & 0xFFF), rend = (int)((w0 >> 12) & 0xFFF);
              cd.allele = allele | (pass == 0 ? 0x80000000u : 0);  // bit31: '+' strand
              cd.readSE = (uint32_t)rs | ((uint32_t)rend << 16);
              cd.seqStart = (int)(w1 & 0xFFFFF); cd.seqEnd = (int)(w2 & 0xFFFFF);
              int m0 = (int)(w1 >> 20), m = (int)(w2 >> 20);
              cd.match = (uint32_t)m0 | ((uint32_t)m << 16);
              cd.re = re;
              myStage[base + off + i] = cd;
              VoteKey vk = voteKey(m0, rs, rend, allele, pass == 0 ? 1 : 0, cd.seqStart, cd.seqEnd);
              if (vk < best) best = vk;
            }
          }
          __syncthreads();
          if (tid == 0) sStageCount = base + tot;
          __syncthreads();
        }
        __syncthreads();
      }  // tile
    }    // strand
    // ---- 4. strand vote and copy-out of the winning strand's candidates -------------------------------------------
    sVoteHi[tid] = best.hi; sVoteLo[tid] = best.lo;
    __syncthreads();
    for (int o = WG / 2; o > 0; o >>= 1) {
      if (tid < o) {
        VoteKey a{sVoteHi[tid], sVoteLo[tid]}, b{sVoteHi[tid + o], sVoteLo[tid + o]};
        if (b < a) { sVoteHi[tid] = b.hi; sVoteLo[tid] = b.lo; }
      }
      __syncthreads();
    }
    const uint32_t winPlus = (uint32_t)(sVoteHi[0] & 1);
    const uint32_t nStage = sStageCount;
    __syncthreads();
    // count winners
    uint32_t mine = 0;
    for (uint32_t i = tid; i < nStage; i += WG) mine += ((myStage[i].allele >> 31) == winPlus) ? 1u : 0u;
    uint32_t totWin;
    blockScanExclusive(mine, warpSums, &totWin);
    if (tid == 0) {
      unsigned long long b = atomicAdd(&P.counters[0], (unsigned long long)totWin);
      if (b + totWin > P.candCap) { atomicOr(&P.counters[2], (unsigned long long)ERR_CANDCAP); sBase = 0xFFFFFFFFu; P.candStart[re] = 0; P.candCount[re] = 0; }
      else { sBase = (uint32_t)b; P.candStart[re] = (uint32_t)b; P.candCount[re] = totWin; }
    }
    __syncthreads();
    if (sBase != 0xFFFFFFFFu) {
      // order-preserving copy in chunks of WG
      uint32_t written = 0;
      for (uint32_t i0 = 0; i0 < nStage; i0 += WG) {
        uint32_t i = i0 + tid;
        uint32_t flag = (i < nStage && (myStage[i].allele >> 31) == winPlus) ? 1u : 0u;
        uint32_t tot;
        uint32_t off = blockScanExclusive(flag, warpSums, &tot);
        if (flag) P.cand[(uint64_t)sBase + written + off] = myStage[i];
        written += tot;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// extension: one lane per candidate
// ------------------------------------------------------------------------------------------------------------------

// SeqSet::IsSeparatorInRange (SeqSet.hpp:487-498) with the -1 / len sentinels of InputRefSeq (924-928)
__device__ __forceinline__ bool sepInRange(const T1kRefDev &ref, uint32_t allele, int s, int e) {
  int len = (int)ref.alleleLen[allele];
  if (s <= -1 && e >= -1) return true;
  if (s <= len && e >= len) return true;
  uint32_t b = ref.sepStart[allele], en = ref.sepStart[allele + 1];
  for (uint32_t i = b; i < en; ++i) {
    int p = ref.sepPos[i];
    if (p >= s && p <= e) return true;
  }
  return false;
}

__device__ inline bool lowComplexity(const uint64_t *rb, const uint64_t *rn, int rs, int re) {  // SeqSet.hpp:458-485
  int cnt[4] = {0, 0, 0, 0};
  int L = re - rs + 1;
  for (int o = 0; o < L; o += 32) {
    uint64_t x = t1k_get32(rb, rs + o), nn = t1k_get32(rn, rs + o);
    uint64_t valid = T1K_EVEN & ~nn & t1k_lowmask(L - o);
    uint64_t lo = x & T1K_EVEN, hi = (x >> 1) & T1K_EVEN;
    cnt[0] += __popcll(~lo & ~hi & valid);
    cnt[1] += __popcll(lo & ~hi & valid);
    cnt[2] += __popcll(~lo & hi & valid);
    cnt[3] += __popcll(lo & hi & valid);
  }
  int low = 0, lowTotal = 0;
  for (int i = 0; i < 4; ++i)
    if (cnt[i] <= 2) { ++low; lowTotal += cnt[i]; }
  if (lowTotal * 7 >= L) return false;
  return low >= 2;
}

__global__ __launch_bounds__(WG) void k_extend(ExtendArgs P) {
  uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= P.nCand) return;
  T1kCand c = P.cand[gid];
  T1kExt x{};
  const uint32_t allele = c.allele & 0x7FFFFFFFu;
  const int pass = (c.allele >> 31) ? 0 : 1;
  const int S = P.reads.S;
  const uint64_t *rb = P.reads.bases + ((uint64_t)c.re * 2 + pass) * S;
  const uint64_t *rn = P.reads.nmask + ((uint64_t)c.re * 2 + pass) * S;
  const int len = P.reads.len[c.re];
  const int rs = (int)(c.readSE & 0xFFFF), re = (int)(c.readSE >> 16);
  const int ss = c.seqStart, se = c.seqEnd;
  const int matchCnt = (int)(c.match >> 16);
  double sim = (double)matchCnt / (double)(se - ss + 1 + re - rs + 1);  // SeqSet.hpp:1838-1840
  if (lowComplexity(rb, rn, rs, re)) sim = 0;                           // 1844-1845
  if (sim < P.sim) { x.flags = T1K_F_DROP; P.ext[gid] = x; return; }    // 1894-1908
  uint16_t flags = 0;
  if (sepInRange(P.ref, allele, ss, se)) flags |= T1K_F_SEPSEED;                                  // 2163
  if (sepInRange(P.ref, allele, ss - rs, se + (len - re - 1))) flags |= T1K_F_NEEDCLIP;           // 2167-2169
  if (flags & T1K_F_SEPSEED) { x.flags = flags; P.ext[gid] = x; return; }
  // ExtendOverlap (1994-2100)
  const int alleleLen = (int)P.ref.alleleLen[allele];
  const int64_t goff = (int64_t)P.ref.alleleOff[allele];
  int lo = rs < ss ? rs : ss;
  int leftClip = rs > ss ? rs - ss : 0, rightClip = 0;
  {
    // nearest N to the left of ss within lo bases
    uint32_t b = P.ref.sepStart[allele], en = P.ref.sepStart[allele + 1];
    int bestP = -1;
    for (uint32_t i = b; i < en; ++i) {
      int p = P.ref.sepPos[i];
      if (p < ss && p >= ss - lo && p > bestP) bestP = p;
    }
    if (bestP >= 0) { int i = ss - 1 - bestP; leftClip = lo - i; lo = i; }
  }
  int match = t1k_ga_matches_window(rb, rn, rs - lo, P.ref.bases, P.ref.nmask, goff + ss - lo, lo, &P.counters[7]);
  int ro = (len - 1 - re) < (alleleLen - 1 - se) ? (len - 1 - re) : (alleleLen - 1 - se);
  if (len - 1 - re > alleleLen - 1 - se) rightClip = len - 1 - re - (alleleLen - 1 - se);
  {
    uint32_t b = P.ref.sepStart[allele], en = P.ref.sepStart[allele + 1];
    int bestP = 0x7FFFFFFF;
    for (uint32_t i = b; i < en; ++i) {
      int p = P.ref.sepPos[i];
      if (p > se && p <= se + ro && p < bestP) bestP = p;
    }
    if (bestP != 0x7FFFFFFF) { int i = bestP - se - 1; rightClip = ro - i; ro = i; }
  }
  match += t1k_ga_matches_window(rb, rn, re + 1, P.ref.bases, P.ref.nmask, goff + se + 1, ro, &P.counters[7]);
  int eMatch = 2 * match + matchCnt;
  int ers = rs - lo, ere = re + ro, ess = ss - lo, ese = se + ro;
  double esim = (double)eMatch / (double)(ere - ers + 1 + ese - ess + 1);
  if (!(esim < P.sim)) flags |= T1K_F_EXTOK;                          // 2074 (before clip credit)
  if (leftClip > 0 || rightClip > 0) eMatch += 2 * leftClip + 2 * rightClip;  // 2078-2087
  x.seqStart = ess; x.seqEnd = ese; x.readStart = (uint16_t)ers; x.readEnd = (uint16_t)ere;
  x.matchCnt = (uint16_t)eMatch; x.leftClip = (uint16_t)leftClip; x.rightClip = (uint16_t)rightClip; x.flags = flags;
  P.ext[gid] = x;
}

// ------------------------------------------------------------------------------------------------------------------
// selection: sort + latch, one workgroup per read-end
// ------------------------------------------------------------------------------------------------------------------

// bitonic sort of n (key, idx) pairs, n padded to a power of two by the caller with key = ~0
__device__ inline void bitonicSort(uint64_t *key, uint32_t *idx, uint32_t np2) {
  for (uint32_t size = 2; size <= np2; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      for (uint32_t t = threadIdx.x; t < np2 / 2; t += blockDim.x) {
        uint32_t lo = 2 * t - (t & (stride - 1));
        uint32_t hi = lo + stride;
        bool up = (lo & size) == 0;
        uint64_t a = key[lo], b = key[hi];
        bool sw = up ? (a > b) : (a < b);
        if (sw) { key[lo] = b; key[hi] = a; uint32_t x = idx[lo]; idx[lo] = idx[hi]; idx[hi] = x; }
      }
      __syncthreads();
    }
  }
}

// full comparator on the seed coordinates for the (rare) ties of the 64-bit key
__device__ inline bool candBeforeFull(const T1kCand &a, const T1kCand &b) {
  int ars = a.readSE & 0xFFFF, are = a.readSE >> 16, brs = b.readSE & 0xFFFF, bre = b.readSE >> 16;
  if (ars != brs) return ars < brs;
  if (are != bre) return are < bre;
  if (a.seqStart != b.seqStart) return a.seqStart < b.seqStart;
  return a.seqEnd < b.seqEnd;
}

#define SELECT_LDS_CAP 8192

__global__ __launch_bounds__(WG) void k_select(SelectArgs P) {
  extern __shared__ uint64_t dynLds[];
  uint64_t *sKey = dynLds;
  uint32_t *sIdx = (uint32_t *)(dynLds + SELECT_LDS_CAP);
  __shared__ uint32_t warpSums[4];
  __shared__ int sLatch, sGood, sBest;
  __shared__ uint32_t sBase;
  const int tid = threadIdx.x;
  for (uint32_t re = blockIdx.x; re < P.reads.nReadEnds; re += gridDim.x) {
    const uint32_t n = P.candCount[re], c0 = P.candStart[re];
    if (n == 0) {
      if (tid == 0) { P.ovlStart[re] = 0; P.ovlCount[re] = 0; }
      continue;
    }
    uint32_t np2 = 1;
    while (np2 < n) np2 <<= 1;
    uint64_t *key; uint32_t *idx;
    if (np2 <= SELECT_LDS_CAP) { key = sKey; idx = sIdx; }
    else if (np2 <= P.sortCap) { key = P.sortScratch + (uint64_t)blockIdx.x * P.sortCap * 2; idx = (uint32_t *)(key + P.sortCap); }
    else {
      if (tid == 0) { atomicOr(&P.counters[2], (unsigned long long)ERR_SORTCAP); P.ovlStart[re] = 0; P.ovlCount[re] = 0; }
      continue;
    }
    // key: matchCnt desc, similarity desc (== readSpan+seqSpan asc at equal matchCnt), readSpan desc, allele asc;
    // dropped candidates sort last
    for (uint32_t i = tid; i < np2; i += WG) {
      uint64_t kk = ~0ull;
      if (i < n) {
        const T1kCand c = P.cand[c0 + i];
        const uint16_t fl = P.ext[c0 + i].flags;
        if (!(fl & T1K_F_DROP)) {
          int rs = c.readSE & 0xFFFF, rend = c.readSE >> 16;
          int m = (int)(c.match >> 16);
          int rspan = rend - rs, d = rspan + 1 + c.seqEnd - c.seqStart + 1;
          kk = ((uint64_t)(4095 - m) << 50) | ((uint64_t)(d & 0x1FFF) << 37) | ((uint64_t)(4095 - rspan) << 24) | (uint64_t)(c.allele & 0xFFFFFF);
        }
      }
      key[i] = kk; idx[i] = i;
    }
    __syncthreads();
    bitonicSort(key, idx, np2);
    // resolve ties of the packed key with the remaining comparator fields (same allele, same spans)
    if (tid == 0) {
      for (uint32_t i = 1; i < n; ++i) {
        if (key[i] == ~0ull) break;
        if (key[i] != key[i - 1]) continue;
        uint32_t j = i;
        while (j > 0 && key[j - 1] == key[j] && candBeforeFull(P.cand[c0 + idx[j]], P.cand[c0 + idx[j - 1]])) {
          uint32_t t = idx[j]; idx[j] = idx[j - 1]; idx[j - 1] = t;
          --j;
        }
      }
      sLatch = 0x7FFFFFFF; sGood = -1; sBest = -1;
    }
    __syncthreads();
    // latch position: first tried candidate whose extension fails (all candidates before the latch are tried)
    int myLatch = 0x7FFFFFFF;
    for (uint32_t i = tid; i < n; i += WG) {
      if (key[i] == ~0ull) continue;
      uint16_t fl = P.ext[c0 + idx[i]].flags;
      if (fl & T1K_F_SEPSEED) continue;
      if (!(fl & T1K_F_EXTOK)) { if ((int)i < myLatch) myLatch = (int)i; }
    }
    atomicMin(&sLatch, myLatch);
    __syncthreads();
    const int latch = sLatch;
    // goodMatchCnt = seed matchCnt of the first emitted candidate before the latch (the list is sorted by it)
    int myGood = 0x7FFFFFFF;
    for (uint32_t i = tid; i < n && (int)i < latch; i += WG) {
      if (key[i] == ~0ull) continue;
      uint16_t fl = P.ext[c0 + idx[i]].flags;
      if ((fl & T1K_F_SEPSEED) || !(fl & T1K_F_EXTOK)) continue;
      if ((int)i < myGood) myGood = (int)i;
    }
    __shared__ int sFirst;
    if (tid == 0) sFirst = 0x7FFFFFFF;
    __syncthreads();
    atomicMin(&sFirst, myGood);
    __syncthreads();
    if (tid == 0) sGood = sFirst == 0x7FFFFFFF ? -1 : (int)(P.cand[c0 + idx[sFirst]].match >> 16);
    __syncthreads();
    const int good = sGood;
    // emit flags + best extended matchCnt
    uint32_t written = 0;
    int myBest = -1;
    // first pass: count and best
    uint32_t mine = 0;
    for (uint32_t i = tid; i < n; i += WG) {
      bool emit = false;
      if (key[i] != ~0ull) {
        const T1kExt x = P.ext[c0 + idx[i]];
        if (!(x.flags & T1K_F_SEPSEED)) {
          bool tried = true;
          if ((int)i > latch) {
            const T1kCand c = P.cand[c0 + idx[i]];
            int m = (int)(c.match >> 16);
            int rs = c.readSE & 0xFFFF, rend = c.readSE >> 16;
            double sim = (double)m / (double)(c.seqEnd - c.seqStart + 1 + rend - rs + 1);
            if (m < good && (!(x.flags & T1K_F_NEEDCLIP) || sim < 0.95)) tried = false;  // SeqSet.hpp:2170-2172
          }
          emit = tried && (x.flags & T1K_F_EXTOK);
          if (emit && (int)x.matchCnt > myBest) myBest = x.matchCnt;
        }
      }
      if (emit) { ++mine; idx[i] |= 0x80000000u; }
    }
    atomicMax(&sBest, myBest);
    uint32_t tot;
    blockScanExclusive(mine, warpSums, &tot);
    if (tid == 0) {
      unsigned long long b = atomicAdd(&P.counters[1], (unsigned long long)tot);
      if (b + tot > P.ovlCap) { atomicOr(&P.counters[2], (unsigned long long)ERR_OVLCAP); sBase = 0xFFFFFFFFu; P.ovlStart[re] = 0; P.ovlCount[re] = 0; }
      else { sBase = (uint32_t)b; P.ovlStart[re] = (uint32_t)b; P.ovlCount[re] = tot; }
    }
    __syncthreads();
    const int bestMatch = sBest;
    if (sBase != 0xFFFFFFFFu) {
      for (uint32_t i0 = 0; i0 < n; i0 += WG) {
        uint32_t i = i0 + tid;
        uint32_t flag = (i < n && (idx[i] & 0x80000000u)) ? 1u : 0u;
        uint32_t t2;
        uint32_t off = blockScanExclusive(flag, warpSums, &t2);
        if (flag) {
          uint32_t ci = c0 + (idx[i] & 0x7FFFFFFFu);
          const T1kCand c = P.cand[ci];
          const T1kExt x = P.ext[ci];
          T1kOvl o;
          o.allele = c.allele & 0x7FFFFFFFu;
          o.seqStart = x.seqStart; o.seqEnd = x.seqEnd; o.readStart = x.readStart; o.readEnd = x.readEnd;
          o.matchCnt = x.matchCnt; o.relaxed = 0; o.leftClip = x.leftClip; o.rightClip = x.rightClip; o.re = re;
          o.flags = ((int)x.matchCnt >= bestMatch - 10 ? 1u : 0u) | ((c.allele >> 31) ? 0u : 2u);  // SeqSet.hpp:2200
          P.ovl[(uint64_t)sBase + written + off] = o;
        }
        written += t2;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// near-best full alignment: relaxedMatchCnt + coverage (SeqSet.hpp:2188-2285)
// ------------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(WG) void k_fullalign(FullArgs P) {
  uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= P.nOvl) return;
  T1kOvl o = P.ovl[gid];
  if (!(o.flags & 1)) { P.ovl[gid].relaxed = 0; return; }  // 2282
  const int pass = (o.flags & 2) ? 1 : 0;
  const int S = P.reads.S;
  const uint64_t *rb = P.reads.bases + ((uint64_t)o.re * 2 + pass) * S;
  const uint64_t *rn = P.reads.nmask + ((uint64_t)o.re * 2 + pass) * S;
  const int64_t goff = (int64_t)P.ref.alleleOff[o.allele];
  const int L = o.readEnd - o.readStart + 1, Ls = o.seqEnd - o.seqStart + 1;
  const int w = (int)P.reads.weight[o.re];
  bool slow = (L != Ls);
  int x = 0;
  if (!slow) {
    x = t1k_hamming(rb, rn, o.readStart, P.ref.bases, P.ref.nmask, goff + o.seqStart, L);
    if (x > 3) slow = true;
  }
  if (slow) {
    unsigned long long q = atomicAdd(&P.counters[8], 1ull);
    if (q < P.slowCap) P.slowQueue[q] = (uint32_t)gid; else atomicOr(&P.counters[2], (unsigned long long)ERR_SLOWCAP);
    return;
  }
  // ungapped alignment: columns are MATCH except at the x mismatching positions
  int exonMis = 0;
  uint64_t carry = 0;  // coverage state of the previous position
  int32_t *diff = P.ref.covDiff + goff + o.seqStart;
  for (int off = 0; off < L; off += 32) {
    uint64_t lm = t1k_lowmask(L - off);
    uint64_t rnn = t1k_get32(rn, o.readStart + off), gnn = t1k_get32(P.ref.nmask, goff + o.seqStart + off);
    uint64_t xo = t1k_get32(rb, o.readStart + off) ^ t1k_get32(P.ref.bases, goff + o.seqStart + off);
    uint64_t mm = (xo | (xo >> 1)) & T1K_EVEN & ~(rnn | gnn) & lm;
    if (P.relax) exonMis += __popcll(mm & t1k_get32(P.ref.exon, goff + o.seqStart + off));
    // covered: MATCH column, read base not N (2261-2265); an N allele base never feeds GetSeqMissingBaseCoverage's
    // counter of the allele's own base, so it is left out
    uint64_t cov = T1K_EVEN & lm & ~mm & ~rnn & ~gnn;
    uint64_t tr = cov ^ ((cov << 2) | carry);  // positions whose covered state differs from the previous position
    while (tr) {
      int b = __ffsll((long long)tr) - 1;
      tr &= tr - 1;
      bool on = (cov >> b) & 1;
      if (w) atomicAdd(&diff[off + (b >> 1)], on ? w : -w);
    }
    carry = (cov >> 62) & 1;
  }
  if (carry && w) atomicAdd(&diff[L], -w);  // a run reaching the last position of a full final word closes at L
  int relaxed = P.relax ? 2 * (L - exonMis) : (int)o.matchCnt;  // 2215-2250
  P.ovl[gid].relaxed = (uint16_t)relaxed;
}


__global__ __launch_bounds__(64) void k_fullalign_slow(SlowArgs P) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t nThreads = gridDim.x * blockDim.x;
  uint8_t *mine = P.scratch + (uint64_t)t * P.perThread;
  int *rows = (int *)mine;
  int8_t *ops = (int8_t *)(mine + GA_SCRATCH_INTS * 4);
  uint8_t *trace = mine + GA_SCRATCH_INTS * 4 + 4224;
  for (uint32_t q = t; q < P.nSlow; q += nThreads) {
    uint32_t gid = P.slowQueue[q];
    T1kOvl o = P.ovl[gid];
    const int pass = (o.flags & 2) ? 1 : 0;
    const int S = P.reads.S;
    const uint64_t *rb = P.reads.bases + ((uint64_t)o.re * 2 + pass) * S;
    const uint64_t *rn = P.reads.nmask + ((uint64_t)o.re * 2 + pass) * S;
    const int64_t goff = (int64_t)P.ref.alleleOff[o.allele];
    const int lp = o.readEnd - o.readStart + 1, lt = o.seqEnd - o.seqStart + 1;
    const int w = (int)P.reads.weight[o.re];
    if ((lp + 1) * (lt + 1) > P.maxCells || lt > GA_BIG_MAX) { atomicOr(&P.counters[2], (unsigned long long)ERR_SLOWCAP); continue; }
    atomicAdd(&P.counters[7], 1ull);
    T1kSeqView T{P.ref.bases, P.ref.nmask, goff + o.seqStart}, Pv{rb, rn, (int64_t)o.readStart};
    t1k_ga_general(T, lt, Pv, lp, rows, trace, nullptr);
    int n = t1k_ga_traceback(trace, lt, lp, ops);
    int m = 0, refPos = o.seqStart, readPos = o.readStart;
    int32_t *cov = P.ref.covDiff + goff;
    for (int i = 0; i < n; ++i) {
      int op = ops[i];
      bool ex = refPos < (int)P.ref.alleleLen[o.allele] ? t1k_bit(P.ref.exon, goff + refPos) != 0 : false;
      if (P.relax) { if (ex) { if (op == 0) ++m; } else ++m; }
      if (op == 0 && w) {
        // MATCH with a real read base on a real allele base: +w at refPos (difference array: +w here, -w next)
        if (!t1k_bit(rn, readPos) && !t1k_bit(P.ref.nmask, goff + refPos)) { atomicAdd(&cov[refPos], w); atomicAdd(&cov[refPos + 1], -w); }
      }
      if (op != 2) ++refPos;
      if (op != 3) ++readPos;
    }
    P.ovl[gid].relaxed = (uint16_t)(P.relax ? 2 * m : (int)o.matchCnt);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// > 1000 overlaps: sort by _overlap::operator< on the extended records and cut (SeqSet.hpp:2290-2298)
// ------------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ double ovlSimilarity(const T1kOvl &o) {
  // ExtendOverlap: matchCnt / (spans) without clips, (matchCnt incl. credit) / (spans + 2 clips) with (2066, 2085-2086)
  int spans = o.readEnd - o.readStart + 1 + o.seqEnd - o.seqStart + 1 + 2 * o.leftClip + 2 * o.rightClip;
  return (double)o.matchCnt / (double)spans;
}

__device__ inline bool ovlBeforeFull(const T1kOvl &a, const T1kOvl &b) {
  if (a.readStart != b.readStart) return a.readStart < b.readStart;
  if (a.readEnd != b.readEnd) return a.readEnd < b.readEnd;
  if (a.seqStart != b.seqStart) return a.seqStart < b.seqStart;
  return a.seqEnd < b.seqEnd;
}

__global__ __launch_bounds__(WG) void k_truncate(TruncArgs P) {
  extern __shared__ uint64_t dynLds[];
  uint64_t *sKey = dynLds;
  uint32_t *sIdx = (uint32_t *)(dynLds + SELECT_LDS_CAP);
  __shared__ uint32_t sCut;
  const int tid = threadIdx.x;
  for (uint32_t re = blockIdx.x; re < P.reads.nReadEnds; re += gridDim.x) {
    const uint32_t n = P.ovlCount[re], o0 = P.ovlStart[re];
    if (n <= 1000) continue;
    uint32_t np2 = 1;
    while (np2 < n) np2 <<= 1;
    uint64_t *key; uint32_t *idx;
    uint64_t *wgScratch = P.sortScratch + (uint64_t)blockIdx.x * (P.sortCap * 2 + (uint64_t)P.sortCap * 4);
    if (np2 <= SELECT_LDS_CAP) { key = sKey; idx = sIdx; }
    else if (np2 <= P.sortCap) { key = wgScratch; idx = (uint32_t *)(key + P.sortCap); }
    else { if (tid == 0) atomicOr(&P.counters[2], (unsigned long long)ERR_SORTCAP); continue; }
    T1kOvl *stage = (T1kOvl *)(wgScratch + P.sortCap * 2);
    if (n > P.sortCap) { if (tid == 0) atomicOr(&P.counters[2], (unsigned long long)ERR_SORTCAP); continue; }
    for (uint32_t i = tid; i < np2; i += WG) {
      uint64_t kk = ~0ull;
      if (i < n) {
        const T1kOvl o = P.ovl[o0 + i];
        int rspan = o.readEnd - o.readStart;
        int d = rspan + 1 + o.seqEnd - o.seqStart + 1 + 2 * o.leftClip + 2 * o.rightClip;  // similarity desc == d asc at equal matchCnt
        kk = ((uint64_t)(4095 - o.matchCnt) << 50) | ((uint64_t)(d & 0x1FFF) << 37) | ((uint64_t)(4095 - rspan) << 24) | (uint64_t)(o.allele & 0xFFFFFF);
        stage[i] = o;
      }
      key[i] = kk; idx[i] = i;
    }
    __syncthreads();
    bitonicSort(key, idx, np2);
    if (tid == 0) {
      for (uint32_t i = 1; i < n; ++i) {
        if (key[i] != key[i - 1]) continue;
        uint32_t j = i;
        while (j > 0 && key[j - 1] == key[j] && ovlBeforeFull(stage[idx[j]], stage[idx[j - 1]])) {
          uint32_t t = idx[j]; idx[j] = idx[j - 1]; idx[j - 1] = t;
          --j;
        }
      }
      double s0 = ovlSimilarity(stage[idx[0]]);
      uint32_t j = 1;
      for (; j < n; ++j)
        if (ovlSimilarity(stage[idx[j]]) < s0 - 0.1) break;
      sCut = j;
    }
    __syncthreads();
    const uint32_t cut = sCut;
    for (uint32_t i = tid; i < cut; i += WG) P.ovl[o0 + i] = stage[idx[i]];
    if (tid == 0) P.ovlCount[re] = cut;
    __syncthreads();
  }
}

// prefix-sum of the coverage difference array, one workgroup per allele, and read-out of the allele's coverage
__global__ __launch_bounds__(WG) void k_coverage_scan(T1kRefDev ref, int32_t *out, const uint64_t *outOff) {
  uint32_t a = blockIdx.x;
  if (a >= ref.nAlleles) return;
  __shared__ uint32_t warpSums[4];
  __shared__ int32_t sCarry;
  const int len = (int)ref.alleleLen[a];
  const int32_t *d = ref.covDiff + ref.alleleOff[a];
  int32_t *o = out + outOff[a];
  if (threadIdx.x == 0) sCarry = 0;
  __syncthreads();
  for (int base = 0; base < len; base += WG) {
    int i = base + threadIdx.x;
    int32_t v = i < len ? d[i] : 0;
    uint32_t tot;
    uint32_t ex = blockScanExclusive((uint32_t)v, warpSums, &tot);
    int32_t incl = (int32_t)ex + v + sCarry;
    if (i < len) o[i] = incl;
    __syncthreads();
    if (threadIdx.x == 0) sCarry += (int32_t)tot;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------------------------
int t1k_launch_pack(t1k_ctx *ctx, const char *dAscii, const uint64_t *dOffs, uint32_t n, int S, uint64_t *bases, uint64_t *nmask, uint16_t *lens) {
  uint64_t total = (uint64_t)n * S;
  if (!total) return 0;
  hipLaunchKernelGGL(k_pack_reads, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, dAscii, dOffs, n, S, bases, nmask, lens);
  return 0;
}

size_t t1k_seed_chain_lds(int S) { return (size_t)TILE_ALLELES * 4 + (size_t)(2 * S * 32) * (4 + 4 + 4 + 2); }
size_t t1k_wg_groups_u32() { return (size_t)TILE_ALLELES * 3; }
size_t t1k_wg_thread_u32() { return (size_t)WG * THREAD_SCRATCH_U32; }
size_t t1k_wg_big_u32() { return (size_t)3 * BIG_CAP + GA_SCRATCH_INTS; }
size_t t1k_slow_per_thread(int maxCells) { return (size_t)GA_SCRATCH_INTS * 4 + 4224 + (size_t)maxCells + 64; }

void t1k_launch_seed_chain(t1k_ctx *ctx, const AssignArgs &a, int nWg) {
  size_t lds = t1k_seed_chain_lds(a.reads.S);
  hipFuncSetAttribute((const void *)k_seed_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_seed_chain, dim3(nWg), dim3(WG), lds, ctx->stream, a);
}
void t1k_launch_extend(t1k_ctx *ctx, const ExtendArgs &a) {
  if (!a.nCand) return;
  hipLaunchKernelGGL(k_extend, dim3((unsigned)((a.nCand + WG - 1) / WG)), dim3(WG), 0, ctx->stream, a);
}
#define SELECT_LDS_BYTES (SELECT_LDS_CAP * 12)
void t1k_launch_select(t1k_ctx *ctx, const SelectArgs &a, int nWg) {
  hipFuncSetAttribute((const void *)k_select, hipFuncAttributeMaxDynamicSharedMemorySize, SELECT_LDS_BYTES);
  hipLaunchKernelGGL(k_select, dim3(nWg), dim3(WG), SELECT_LDS_BYTES, ctx->stream, a);
}
void t1k_launch_fullalign(t1k_ctx *ctx, const FullArgs &a) {
  if (!a.nOvl) return;
  hipLaunchKernelGGL(k_fullalign, dim3((unsigned)((a.nOvl + WG - 1) / WG)), dim3(WG), 0, ctx->stream, a);
}
void t1k_launch_fullalign_slow(t1k_ctx *ctx, const SlowArgs &a, int nBlocks) { hipLaunchKernelGGL(k_fullalign_slow, dim3(nBlocks), dim3(64), 0, ctx->stream, a); }
void t1k_launch_truncate(t1k_ctx *ctx, const TruncArgs &a, int nWg) {
  hipFuncSetAttribute((const void *)k_truncate, hipFuncAttributeMaxDynamicSharedMemorySize, SELECT_LDS_BYTES);
  hipLaunchKernelGGL(k_truncate, dim3(nWg), dim3(WG), SELECT_LDS_BYTES, ctx->stream, a);
}
void t1k_launch_coverage_scan(t1k_ctx *ctx, const T1kRefDev &ref, int32_t *out, const uint64_t *outOff) {
  hipLaunchKernelGGL(k_coverage_scan, dim3(ref.nAlleles), dim3(WG), 0, ctx->stream, ref, out, outOff);
}
