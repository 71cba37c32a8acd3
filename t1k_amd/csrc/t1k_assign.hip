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
#ifdef T1K_PHASE_TIMERS
#define PHASE(k) do { if (threadIdx.x == 0) { long long now_ = wall_clock64(); atomicAdd(&P.counters[((k) >= 9 ? 32 + (k) : 16 + (k))], (unsigned long long)(now_ - tPhase)); tPhase = now_; } } while (0)
#else
#define PHASE(k) do {} while (0)
#endif
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
__device__ inline int gapMatches(const ReadCtx &c, int ra, int ga, int lp, int lt, int *gaScratch, int gaMax, unsigned int *dpCounter, unsigned long long *errFlags) {
  if (lp == lt) return t1k_ga_matches_window(c.rb, c.rn, ra, c.gb, c.gn, c.goff + ga, lp, dpCounter);
  if (lt == 0 || lp == 0) return 0;
  if (dpCounter) ++*dpCounter;
  T1kSeqView T{c.gb, c.gn, c.goff + ga}, P{c.rb, c.rn, ra};
  int nm = 0;
  if (lt > gaMax) { atomicOr(errFlags, (unsigned long long)ERR_BIGGROUP); return 0; }
  t1k_ga_general(T, lt, P, lp, gaScratch, nullptr, &nm);
  return nm;
}

// ------------------------------------------------------------------------------------------------------------------
// Exact memo of gap alignments within one read-end.  Thousands of alleles of a gene carry the same bases under a given
// read window, so the same banded DP would be recomputed for each of them.  One 64-bit entry identifies a job completely:
//   [gpos:34 | matches:9 | readPos:11 | len:9 | strand:1]
// A probe whose (strand, readPos, len) agree verifies that the allele window at the entry's gpos holds exactly the same
// bases and N-mask as its own window before it reuses the stored match count, so a hit is bit-exact by construction.
// The table lives in per-workgroup HBM scratch (L2-resident, 32 KB) and is cleared per read-end.
// ------------------------------------------------------------------------------------------------------------------
#define GAP_CACHE 4096
__device__ inline bool sameWindow(const uint64_t *gb, const uint64_t *gn, int64_t a, int64_t b, int L) {
  for (int o = 0; o < L; o += 32) {
    uint64_t lm = t1k_lowmask(L - o);
    if (((t1k_get32(gb, a + o) ^ t1k_get32(gb, b + o)) & lm) | ((t1k_get32(gn, a + o) ^ t1k_get32(gn, b + o)) & lm)) return false;
  }
  return true;
}

#define GAP_PENDING 0x1FFull
// DEFER = true : never run a DP here.  A miss claims a memo slot (CAS) with the PENDING marker and appends the slot to the
//                workgroup's job list; the caller parks its group (return -1) until the dense DP phase has filled the memo.
// DEFER = false: a miss is computed inline (used after the dense phase; only slot-collision leftovers get here).
template <bool DEFER>
__device__ inline int gapMatchesCached(const ReadCtx &c, int readPos, int64_t gpos, int L, int strandBit, unsigned long long *cache, unsigned int *dpCounter,
                                       uint32_t *jobList, uint32_t *jobCount) {
  if (L <= 0) return 0;
  // mismatch count and a content hash of the allele window in one sweep
  int x = 0;
  uint64_t hsh = 0x9E3779B97F4A7C15ull ^ ((uint64_t)readPos << 20) ^ ((uint64_t)L << 1) ^ (uint64_t)strandBit;
  for (int o = 0; o < L; o += 32) {
    uint64_t lm = t1k_lowmask(L - o);
    uint64_t gw = t1k_get32(c.gb, gpos + o) & lm, gnw = t1k_get32(c.gn, gpos + o) & lm;
    uint64_t xo = t1k_get32(c.rb, readPos + o) ^ gw;
    uint64_t mm = (xo | (xo >> 1)) & T1K_EVEN & ~(t1k_get32(c.rn, readPos + o) | gnw) & lm;
    x += __popcll(mm);
    hsh = (hsh ^ gw ^ (gnw << 1)) * 0xD6E8FEB86659FD93ull;
    hsh ^= hsh >> 32;
  }
  if (x <= 3) return L - x;  // exact fast path (see t1k_ga_matches_window)
  if (L > 510 || readPos > 2047) {
    if (DEFER) return -2;  // not memoisable: the retry phase computes it inline
    return t1k_ga_matches_window(c.rb, c.rn, readPos, c.gb, c.gn, gpos, L, dpCounter);
  }
  const uint64_t idBits = ((uint64_t)readPos << 10) | ((uint64_t)L << 1) | (uint64_t)strandBit;  // low 21 bits of an entry
  const uint32_t slot = (uint32_t)hsh & (GAP_CACHE - 1);
  bool pendingSeen = false;
#pragma unroll
  for (int probe = 0; probe < 2; ++probe) {
    unsigned long long e = __hip_atomic_load(&cache[slot ^ probe], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (e != 0 && (e & 0x1FFFFFull) == idBits) {
      int64_t eg = (int64_t)(e >> 30);
      if (eg == gpos || sameWindow(c.gb, c.gn, eg, gpos, L)) {
        unsigned long long v = (e >> 21) & 0x1FF;
        if (v != GAP_PENDING) return (int)v;
        pendingSeen = true;
      }
    }
  }
  if (DEFER) {
    if (pendingSeen) return -1;
    const unsigned long long pe = ((unsigned long long)gpos << 30) | (GAP_PENDING << 21) | idBits;
#pragma unroll
    for (int probe = 0; probe < 2; ++probe) {
      unsigned long long old = atomicCAS(&cache[slot ^ probe], 0ull, pe);
      if (old == 0ull) {
        uint32_t q = atomicAdd(jobCount, 1u);
        jobList[q] = slot ^ probe;
        return -1;
      }
      if ((old & 0x1FFFFFull) == idBits && ((int64_t)(old >> 30) == gpos || sameWindow(c.gb, c.gn, (int64_t)(old >> 30), gpos, L))) return -1;  // somebody else just claimed it
    }
    return -2;  // both slots taken by other jobs: inline in the retry phase
  }
  if (dpCounter) ++*dpCounter;
  T1kSeqView T{c.gb, c.gn, gpos}, P{c.rb, c.rn, (int64_t)readPos};
  int m = t1k_ga_matches_equal(T, P, L, nullptr);
  if (!pendingSeen) {
    unsigned long long ne = ((unsigned long long)gpos << 30) | ((unsigned long long)m << 21) | idBits;
    unsigned long long cur = __hip_atomic_load(&cache[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == 0) __hip_atomic_store(&cache[slot], ne, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return m;
}

// Single-diagonal group (the common case: the read differs from the allele by substitutions only).
// NW = number of 32-position words covering a read (5: reads <= 160 bp, 10: reads <= 320 bp).
//   * the group's hits are fetched in one burst of independent loads (<= 32 hits; larger groups loop)
//   * majority diagonal by Boyer-Moore vote; every other hit must lie more than `radius` diagonals away (it cannot join the
//     main run, SeqSet.hpp:1360-1392) and there may be at most two such strays (they cannot form a run of >= 3 hits,
//     1400-1405); anything else goes to the general path
//   * on one diagonal the LIS is the identity and both hit lengths are equal; with M = bitmask of hit read-offsets,
//     covered = popcount(dilate(M, k)) and matchCnt = 2*covered + 2*sum over gaps of GlobalAlignment matches (1697-1760)
//   * a gap with x <= 3 mismatches aligns ungapped (g - x matches, exact, see t1k_ga_matches_window); if the whole span
//     has <= 3 mismatches every gap does and matchCnt = 2*(span - mismatches) in closed form
//   * exact prune: a gap with x > 3 yields at most g - 1 matches, so U = 2*(span - sum_{x<=3} x - #{x>3}) bounds matchCnt;
//     if U / (2*span) < -s the candidate is certain to fail the similarity filter (1838-1840, 1894-1908) and is emitted
//     with matchCnt = U (it is dropped by k_extend either way, and the strand vote only reads matchCnt0)
template <int NW, bool DEFER>
__device__ inline int groupFastPath(const uint32_t *h, int n, const ReadCtx &c, int k, int radius, int hitLenRequired, double simThreshold, CandOut &out,
                                    unsigned int *dpCounter, int strandBit, unsigned long long *cache, uint32_t *jobList, uint32_t *jobCount) {
  constexpr int MW = (NW + 1) / 2;  // 64-bit words of the read-offset bitmask
#ifdef T1K_PHASE_TIMERS
  long long tq_ = clock64();
#define SECT(i) do { long long n_ = clock64(); dpCounter[3 + (i)] += (unsigned int)((n_ - tq_) >> 4); tq_ = n_; } while (0)
#else
#define SECT(i) do {} while (0)
#endif
  uint64_t M[MW];
#pragma unroll
  for (int i = 0; i < MW; ++i) M[i] = 0;
  int diag = 0, votes = 0, strays = 0, onDiag = 0;
  if (n <= 32) {
    uint32_t hr[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) hr[i] = i < n ? h[i] : 0u;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if (i < n) {
        int d = (int)(hr[i] & 0xFFF) - (int)(hr[i] >> 12);
        if (votes == 0) { diag = d; votes = 1; }
        else if (d == diag) ++votes;
        else --votes;
      }
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if (i < n) {
        int a = (int)(hr[i] & 0xFFF);
        int d = a - (int)(hr[i] >> 12) - diag;
        if (d != 0) {
          if (d < 0) d = -d;
          if (d <= radius) return 0;
          ++strays;
        } else {
          if (a >= NW * 32) return 0;
          ++onDiag;
#pragma unroll
          for (int w = 0; w < MW; ++w)
            if ((a >> 6) == w) M[w] |= 1ull << (a & 63);
        }
      }
    }
  } else {
    for (int i = 0; i < n; ++i) {
      uint32_t x = h[i];
      int d = (int)(x & 0xFFF) - (int)(x >> 12);
      if (votes == 0) { diag = d; votes = 1; }
      else if (d == diag) ++votes;
      else --votes;
    }
    for (int i = 0; i < n; ++i) {
      uint32_t x = h[i];
      int a = (int)(x & 0xFFF);
      int d = a - (int)(x >> 12) - diag;
      if (d != 0) {
        if (d < 0) d = -d;
        if (d <= radius) return 0;
        ++strays;
      } else {
        if (a >= NW * 32) return 0;
        ++onDiag;
#pragma unroll
        for (int w = 0; w < MW; ++w)
          if ((a >> 6) == w) M[w] |= 1ull << (a & 63);
      }
    }
  }
  SECT(0);
  if (strays > 2) return 0;
  if (onDiag < 3) return 1;  // minHitRequired (1314, 1400)
  if (onDiag * k < hitLenRequired) return 1;
  int first = -1, last = -1;
#pragma unroll
  for (int w = 0; w < MW; ++w) {
    if (M[w]) {
      if (first < 0) first = w * 64 + __ffsll((long long)M[w]) - 1;
      last = w * 64 + 63 - __clzll((long long)M[w]);
    }
  }
  // covered positions: dilate M by k (bit p set iff some hit offset a has a <= p < a + k)
  uint64_t C[MW];
#pragma unroll
  for (int w = 0; w < MW; ++w) C[w] = M[w];
  for (int sft = 1; sft < k; ++sft) {
#pragma unroll
    for (int w = MW - 1; w >= 0; --w) {
      uint64_t carry = w > 0 ? (M[w - 1] >> (64 - sft)) : 0ull;
      C[w] |= (M[w] << sft) | carry;
    }
  }
  int cov = 0;
#pragma unroll
  for (int w = 0; w < MW; ++w) cov += __popcll(C[w]);
  // (the dilation may spill past NW*32 only for offsets that cannot occur: a + k <= len <= NW*32)
  SECT(1);
  if (cov < hitLenRequired) return 1;  // GetTotalHitLengthOnRead/OnSeq (1512-1522)
  const int spanEnd = last + k, span = spanEnd - first;
  // mismatch bits of the span on this diagonal, one burst of independent loads
  uint64_t mmw[NW];
  int mmT = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    mmw[w] = 0;
    const int p0 = w * 32;
    if (p0 < spanEnd && p0 + 32 > first) {
      const int lo = p0 < first ? first : p0;  // never read the allele before its first base
      uint64_t xo = t1k_get32(c.rb, lo) ^ t1k_get32(c.gb, c.goff + lo - diag);
      uint64_t mm = (xo | (xo >> 1)) & T1K_EVEN & ~(t1k_get32(c.rn, lo) | t1k_get32(c.gn, c.goff + lo - diag));
      mm <<= 2 * (lo - p0);
      const int hiN = spanEnd - p0;
      if (hiN < 32) mm &= t1k_lowmask(hiN);
      mmw[w] = mm;
      mmT += __popcll(mm);
    }
  }
  SECT(2);
  int matchCnt;
  if (mmT <= 3) matchCnt = 2 * (span - mmT);
  else {
    // walk the gaps (maximal uncovered runs inside the span)
    int sumSmall = 0, nBig = 0;
    // pass 1: per-gap mismatch counts -> upper bound
    int pos = first;
    // gap iteration helper: next uncovered position >= pos is the lowest clear bit of C at or above pos
    auto nextClear = [&](int from) -> int {
#pragma unroll
      for (int w = 0; w < MW; ++w) {
        if (from < (w + 1) * 64) {
          uint64_t inv = ~C[w];
          if (from > w * 64) inv &= ~0ull << (from - w * 64);
          if (inv) return w * 64 + __ffsll((long long)inv) - 1;
        }
      }
      return MW * 64;
    };
    auto nextSet = [&](int from) -> int {
#pragma unroll
      for (int w = 0; w < MW; ++w) {
        if (from < (w + 1) * 64) {
          uint64_t v = C[w];
          if (from > w * 64) v &= ~0ull << (from - w * 64);
          if (v) return w * 64 + __ffsll((long long)v) - 1;
        }
      }
      return MW * 64;
    };
    auto mmIn = [&](int gs, int ge) -> int {  // mismatches in [gs, ge)
      int x = 0;
#pragma unroll
      for (int q = 0; q < NW; ++q) {
        const int p0 = q * 32;
        if (p0 < ge && p0 + 32 > gs) {
          uint64_t msk = ~0ull;
          if (gs > p0) msk &= ~t1k_lowmask(gs - p0);
          if (ge < p0 + 32) msk &= t1k_lowmask(ge - p0);
          x += __popcll(mmw[q] & msk);
        }
      }
      return x;
    };
    while (true) {
      int gs = nextClear(pos);
      if (gs >= spanEnd) break;
      int ge = nextSet(gs);
      if (ge > spanEnd) ge = spanEnd;
      int x = mmIn(gs, ge);
      if (x <= 3) sumSmall += x; else ++nBig;
      pos = ge;
    }
    const int upper = 2 * (span - sumSmall - nBig);
    if (nBig == 0) matchCnt = upper;  // exact
    else if ((double)upper / (double)(2 * span) < simThreshold) matchCnt = upper;  // certain to be dropped; no DP needed
    else {
      int gapMatch = 0;
      bool parked = false;
      pos = first;
      while (true) {
        int gs = nextClear(pos);
        if (gs >= spanEnd) break;
        int ge = nextSet(gs);
        if (ge > spanEnd) ge = spanEnd;
        int x = mmIn(gs, ge);
        if (x <= 3) gapMatch += (ge - gs) - x;
        else {
          int r = gapMatchesCached<DEFER>(c, gs, c.goff + (gs - diag), ge - gs, strandBit, cache, dpCounter, jobList, jobCount);
          if (r < 0) parked = true; else gapMatch += r;  // keep walking: later gaps register their jobs too
        }
        pos = ge;
      }
      if (parked) return 2;
      matchCnt = 2 * cov + 2 * gapMatch;
    }
  }
  SECT(3);
  out.push(first, last + k - 1, first - diag, last - diag + k - 1, 2 * cov, matchCnt);
  return 1;
}

__device__ __forceinline__ bool hitKeyLess(uint32_t x, uint32_t y) {  // (diag, alleleOff, readOff): CompSortHitCoordDiff (266-274)
  int cx = (int)(x & 0xFFF) - (int)(x >> 12), cy = (int)(y & 0xFFF) - (int)(y >> 12);
  if (cx != cy) return cx < cy;
  return x < y;
}

// General group (several diagonals): restates GetOverlapsFromHits 1338-1551 and the chain walk 1697-1833.
// A[n] sorted copy of the hits, B[n] concordant hits, C[n] packs top (low 16) / link (high 16) of the LIS.
__device__ inline void groupGeneral(const uint32_t *h, int n, const ReadCtx &c, int k, int radius, int hitLenRequired, uint32_t *A, uint32_t *B,
                                     uint32_t *C, int *gaScratch, int gaMax, CandOut &out, unsigned int *dpCounter, unsigned long long *errFlags) {
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
  __shared__ uint32_t sStageCount, sBase, sAnyDeferred, sGenCount, sRetryCount, sJobCount;
  __shared__ uint64_t sVoteHi[WG], sVoteLo[WG];

  const int tid = threadIdx.x;
  const uint32_t kmask = (1u << (2 * k)) - 1;
  uint32_t *myHits = P.wgHits + (uint64_t)blockIdx.x * P.hitCap;
  uint32_t *myGroups = P.wgGroups + (uint64_t)blockIdx.x * TILE_ALLELES * 4;
  uint32_t *genList = myGroups + TILE_ALLELES * 3;
  T1kCand *myStage = P.wgStage + (uint64_t)blockIdx.x * P.stageCap;
  uint32_t *myThread = P.wgThread + ((uint64_t)blockIdx.x * WG + tid) * THREAD_SCRATCH_U32;
  uint32_t *myBig = P.wgBig + (uint64_t)blockIdx.x * (3 * BIG_CAP + GA_SCRATCH_INTS);
  unsigned long long *myCache = P.wgCache + (uint64_t)blockIdx.x * (GAP_CACHE + GAP_CACHE / 2);
  uint32_t *jobList = (uint32_t *)(myCache + GAP_CACHE);
  const int nTiles = (int)((P.ref.nAlleles + TILE_ALLELES - 1) / TILE_ALLELES);

  unsigned int dpLocalArr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned int &dpLocal = dpLocalArr[0];
  unsigned int fastLocal = 0, generalLocal = 0, deferredLocal = 0;
  long long tChain = 0;
  (void)tChain;
  long long tPhase = 0;
  (void)tPhase;
#ifdef T1K_PHASE_TIMERS
  tPhase = wall_clock64();
#endif
  for (uint32_t re = blockIdx.x; re < P.reads.nReadEnds; re += gridDim.x) {
    const int len = P.reads.len[re];
    const int S = P.reads.S;
    const uint64_t *rbase = P.reads.bases + (uint64_t)re * 2 * S;
    const uint64_t *rnm = P.reads.nmask + (uint64_t)re * 2 * S;
    if (tid == 0) { sStageCount = 0; sAnyDeferred = 0; sGenCount = 0; sRetryCount = 0; sJobCount = 0; }
    __syncthreads();
    if (len < k) {  // GetOverlapsFromRead returns -1 (SeqSet.hpp:1598-1599)
      if (tid == 0) { P.candStart[re] = 0; P.candCount[re] = 0; }
      __syncthreads();
      continue;
    }
    const int nk = len - k + 1;
    for (int i = tid; i < GAP_CACHE; i += WG) __hip_atomic_store(&myCache[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    PHASE(0);
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
    PHASE(1);
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
        PHASE(2);
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
        PHASE(3);
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
        PHASE(4);
        // chain: one lane per (strand, allele) group; candidates are packed back into the group's hit slice
#ifdef T1K_PHASE_TIMERS
        long long tc0 = clock64();
#endif
        for (uint32_t g = tid; g < gTot; g += WG) {
          uint32_t allele = myGroups[g * 3 + 0], hs = myGroups[g * 3 + 1], n = myGroups[g * 3 + 2];
          ReadCtx c{rbase + pass * S, rnm + pass * S, len, P.ref.bases, P.ref.nmask, (int64_t)P.ref.alleleOff[allele], (int)P.ref.alleleLen[allele]};
          CandOut out{myHits + hs, 0};
          int done = 0;
          if (len <= 160) done = groupFastPath<5, true>(myHits + hs, (int)n, c, k, P.radius, P.hitLenRequired, P.sim, out, dpLocalArr, pass, myCache, jobList, &sJobCount);
          else if (len <= GROUP_FAST_MAXLEN)
            done = groupFastPath<10, true>(myHits + hs, (int)n, c, k, P.radius, P.hitLenRequired, P.sim, out, dpLocalArr, pass, myCache, jobList, &sJobCount);
          if (done == 1) { ++fastLocal; myGroups[g * 3 + 2] = (uint32_t)out.n; }
          else if (done == 2) {  // parked until the dense DP phase has filled the memo
            uint32_t q = atomicAdd(&sRetryCount, 1u);
            genList[TILE_ALLELES - 1 - q] = g;
          } else {
            // several diagonals: handled after the lock-step loop so that one slow lane does not stall its wavefront
            uint32_t q = atomicAdd(&sGenCount, 1u);
            genList[q] = g;
          }
        }
#ifdef T1K_PHASE_TIMERS
        tChain += clock64() - tc0;
#endif
        __syncthreads();
        PHASE(9);
        {
          // dense DP phase: one lane per distinct (strand, read window, allele window) job registered above
          const uint32_t nJobs = sJobCount;
          for (uint32_t q = tid; q < nJobs; q += WG) {
            const uint32_t slot = jobList[q];
            const unsigned long long e = __hip_atomic_load(&myCache[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int L = (int)((e >> 1) & 0x1FF), readPos = (int)((e >> 10) & 0x7FF), sb = (int)(e & 1);
            const int64_t gpos = (int64_t)(e >> 30);
            T1kSeqView T{P.ref.bases, P.ref.nmask, gpos}, Pv{rbase + sb * S, rnm + sb * S, (int64_t)readPos};
            const int m = t1k_ga_matches_equal(T, Pv, L, nullptr);
            ++dpLocal;
            __hip_atomic_store(&myCache[slot], (e & ~(GAP_PENDING << 21)) | ((unsigned long long)m << 21), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        __threadfence_block();
        __syncthreads();
        PHASE(10);
        {
          const uint32_t nRetry = sRetryCount;
          for (uint32_t q = tid; q < nRetry; q += WG) {
            const uint32_t g = genList[TILE_ALLELES - 1 - q];
            uint32_t allele = myGroups[g * 3 + 0], hs = myGroups[g * 3 + 1], n = myGroups[g * 3 + 2];
            ReadCtx c{rbase + pass * S, rnm + pass * S, len, P.ref.bases, P.ref.nmask, (int64_t)P.ref.alleleOff[allele], (int)P.ref.alleleLen[allele]};
            CandOut out{myHits + hs, 0};
            if (len <= 160) groupFastPath<5, false>(myHits + hs, (int)n, c, k, P.radius, P.hitLenRequired, P.sim, out, dpLocalArr, pass, myCache, jobList, &sJobCount);
            else groupFastPath<10, false>(myHits + hs, (int)n, c, k, P.radius, P.hitLenRequired, P.sim, out, dpLocalArr, pass, myCache, jobList, &sJobCount);
            ++fastLocal;
            myGroups[g * 3 + 2] = (uint32_t)out.n;
          }
        }
        __syncthreads();
        PHASE(11);
        {
          const uint32_t nGen = sGenCount;
          for (uint32_t q = tid; q < nGen; q += WG) {
            const uint32_t g = genList[q];
            uint32_t allele = myGroups[g * 3 + 0], hs = myGroups[g * 3 + 1], n = myGroups[g * 3 + 2];
            if (n > THREAD_CAP) { myGroups[g * 3 + 2] = n | 0x80000000u; ++deferredLocal; sAnyDeferred = 1; continue; }  // lane 0, below
            ReadCtx c{rbase + pass * S, rnm + pass * S, len, P.ref.bases, P.ref.nmask, (int64_t)P.ref.alleleOff[allele], (int)P.ref.alleleLen[allele]};
            CandOut out{myHits + hs, 0};
            if (n <= 48) {
              // small group: work arrays in private (scratch) memory, which is lane-interleaved and therefore coalesced when
              // the lanes of this dense phase walk their arrays in step
              uint32_t wa[48], wb[48], wc[48];
              groupGeneral(myHits + hs, (int)n, c, k, P.radius, P.hitLenRequired, wa, wb, wc, (int *)(myThread + 3 * THREAD_CAP), GA_T_MAX, out, &dpLocal,
                           &P.counters[2]);
            } else
              groupGeneral(myHits + hs, (int)n, c, k, P.radius, P.hitLenRequired, myThread, myThread + THREAD_CAP, myThread + 2 * THREAD_CAP,
                           (int *)(myThread + 3 * THREAD_CAP), GA_T_MAX, out, &dpLocal, &P.counters[2]);
            ++generalLocal;
            myGroups[g * 3 + 2] = (uint32_t)out.n;
          }
        }
        __syncthreads();
        if (tid == 0) { sGenCount = 0; sRetryCount = 0; sJobCount = 0; }
        PHASE(5);
        if (tid == 0 && sAnyDeferred) {
          sAnyDeferred = 0;
          for (uint32_t g = 0; g < gTot; ++g) {
            uint32_t n = myGroups[g * 3 + 2];
            if (!(n & 0x80000000u)) continue;
            n &= 0x7FFFFFFFu;
            uint32_t allele = myGroups[g * 3 + 0], hs = myGroups[g * 3 + 1];
            if (n > BIG_CAP) { atomicOr(&P.counters[2], (unsigned long long)ERR_BIGGROUP); myGroups[g * 3 + 2] = 0; continue; }
            ReadCtx c{rbase + pass * S, rnm + pass * S, len, P.ref.bases, P.ref.nmask, (int64_t)P.ref.alleleOff[allele], (int)P.ref.alleleLen[allele]};
            CandOut out{myHits + hs, 0};
            groupGeneral(myHits + hs, (int)n, c, k, P.radius, P.hitLenRequired, myBig, myBig + BIG_CAP, myBig + 2 * BIG_CAP, (int *)(myBig + 3 * BIG_CAP), GA_BIG_MAX, out,
                         &dpLocal, &P.counters[2]);
            myGroups[g * 3 + 2] = (uint32_t)out.n;
          }
        }
        __syncthreads();
        PHASE(6);
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
        PHASE(7);
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
    PHASE(8);
  }
  // flush the thread-local tallies: one atomic per counter per wavefront
  {
#ifdef T1K_PHASE_TIMERS
    atomicAdd(&P.counters[25], (unsigned long long)(tChain >> 6));
    atomicAdd(&P.counters[26], (unsigned long long)dpLocalArr[1]);
    atomicAdd(&P.counters[27], (unsigned long long)dpLocalArr[2]);
    atomicAdd(&P.counters[28], (unsigned long long)dpLocalArr[3]); atomicAdd(&P.counters[29], (unsigned long long)dpLocalArr[4]);
    atomicAdd(&P.counters[30], (unsigned long long)dpLocalArr[5]); atomicAdd(&P.counters[31], (unsigned long long)dpLocalArr[6]);
#endif
    unsigned int v[4] = {dpLocal, fastLocal, generalLocal, deferredLocal};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned int x = v[q];
      for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
      if ((threadIdx.x & 63) == 0 && x) atomicAdd(&P.counters[q == 0 ? 7 : 10 + q], (unsigned long long)x);
    }
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
  unsigned int dpLocal = 0;
  int match = t1k_ga_matches_window(rb, rn, rs - lo, P.ref.bases, P.ref.nmask, goff + ss - lo, lo, &dpLocal);
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
  match += t1k_ga_matches_window(rb, rn, re + 1, P.ref.bases, P.ref.nmask, goff + se + 1, ro, &dpLocal);
  if (dpLocal) atomicAdd(&P.counters[14], (unsigned long long)dpLocal);
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
    // candidates that failed the similarity filter never reach the sort: count the survivors first
    __shared__ uint32_t sLive;
    if (tid == 0) sLive = 0;
    __syncthreads();
    {
      uint32_t mine = 0;
      for (uint32_t i = tid; i < n; i += WG) mine += (P.ext[c0 + i].flags & T1K_F_DROP) ? 0u : 1u;
      if (mine) atomicAdd(&sLive, mine);
    }
    __syncthreads();
    const uint32_t live = sLive;
    __syncthreads();
    if (live == 0) {
      if (tid == 0) { P.ovlStart[re] = 0; P.ovlCount[re] = 0; }
      continue;
    }
    uint32_t np2 = 1;
    while (np2 < live) np2 <<= 1;
    uint64_t *key; uint32_t *idx;
    if (np2 <= SELECT_LDS_CAP) { key = sKey; idx = sIdx; }
    else if (np2 <= P.sortCap) { key = P.sortScratch + (uint64_t)blockIdx.x * P.sortCap * 2; idx = (uint32_t *)(key + P.sortCap); }
    else {
      if (tid == 0) { atomicOr(&P.counters[2], (unsigned long long)ERR_SORTCAP); P.ovlStart[re] = 0; P.ovlCount[re] = 0; }
      continue;
    }
    // key: matchCnt desc, similarity desc (== readSpan+seqSpan asc at equal matchCnt), readSpan desc, allele asc
    if (tid == 0) sLive = 0;
    for (uint32_t i = tid; i < np2; i += WG) key[i] = ~0ull;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += WG) {
      const uint16_t fl = P.ext[c0 + i].flags;
      if (fl & T1K_F_DROP) continue;
      const T1kCand c = P.cand[c0 + i];
      int rs = c.readSE & 0xFFFF, rend = c.readSE >> 16;
      int m = (int)(c.match >> 16);
      int rspan = rend - rs, d = rspan + 1 + c.seqEnd - c.seqStart + 1;
      uint32_t slot = atomicAdd(&sLive, 1u);  // any order: the sort follows
      key[slot] = ((uint64_t)(4095 - m) << 50) | ((uint64_t)(d & 0x1FFF) << 37) | ((uint64_t)(4095 - rspan) << 24) | (uint64_t)(c.allele & 0xFFFFFF);
      idx[slot] = i;
    }
    for (uint32_t i = live + tid; i < np2; i += WG) idx[i] = 0;
    __syncthreads();
    bitonicSort(key, idx, np2);
    // resolve ties of the packed key with the remaining comparator fields (same allele, same spans)
    const uint32_t nAll = n;
    (void)nAll;
    if (tid == 0) {
      for (uint32_t i = 1; i < live; ++i) {
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
    for (uint32_t i = tid; i < live; i += WG) {
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
    for (uint32_t i = tid; i < live && (int)i < latch; i += WG) {
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
    unsigned int nbLocal = 0;
    int myBest = -1;
    // first pass: count and best
    uint32_t mine = 0;
    for (uint32_t i = tid; i < live; i += WG) {
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
      for (uint32_t i0 = 0; i0 < live; i0 += WG) {
        uint32_t i = i0 + tid;
        uint32_t flag = (i < live && (idx[i] & 0x80000000u)) ? 1u : 0u;
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
          if (o.flags & 1) ++nbLocal;
          P.ovl[(uint64_t)sBase + written + off] = o;
        }
        written += t2;
      }
    }
    for (int o = 32; o > 0; o >>= 1) nbLocal += __shfl_down(nbLocal, o, 64);
    if ((tid & 63) == 0 && nbLocal) atomicAdd(&P.counters[10], (unsigned long long)nbLocal);
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
    // equal spans: register-band traced DP (queue A, from the front); unequal spans: general DP (queue B, from the back)
    if (L == Ls) {
      unsigned long long q = atomicAdd(&P.counters[8], 1ull);
      P.slowQueue[q] = (uint32_t)gid;
    } else {
      unsigned long long q = atomicAdd(&P.counters[15], 1ull);
      P.slowQueue[P.slowCap - 1 - q] = (uint32_t)gid;
    }
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

// equal-span near-best alignments with more than 3 mismatches: banded DP with the band in registers, decision words in a
// coalesced global trace, then the reference's traceback walked backwards (AlignAlgo.hpp:323-408) accumulating the relaxed
// match count and the coverage runs directly (no edit string is materialised).
__global__ __launch_bounds__(WG) void k_fullalign_eq(SlowArgs P) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, nThreads = gridDim.x * blockDim.x;
  uint64_t *trace = (uint64_t *)P.scratch + t;
  for (uint32_t q = t; q < P.nSlow; q += nThreads) {
    const uint32_t gid = P.slowQueue[q];
    const T1kOvl o = P.ovl[gid];
    const int pass = (o.flags & 2) ? 1 : 0;
    const int S = P.reads.S;
    const uint64_t *rb = P.reads.bases + ((uint64_t)o.re * 2 + pass) * S;
    const uint64_t *rn = P.reads.nmask + ((uint64_t)o.re * 2 + pass) * S;
    const int64_t goff = (int64_t)P.ref.alleleOff[o.allele];
    const int L = o.readEnd - o.readStart + 1;
    const int w = (int)P.reads.weight[o.re];
    T1kSeqView T{P.ref.bases, P.ref.nmask, goff + o.seqStart}, Pv{rb, rn, (int64_t)o.readStart};
    t1k_ga_equal_traced(T, Pv, L, trace, nThreads);
    int32_t *cov = P.ref.covDiff + goff;
    const int alleleLen = (int)P.ref.alleleLen[o.allele];
    int relaxed = 0;
    int runLo = -1, runHi = -1;  // current run of covered reference positions, extended downwards
    int ti = L, tj = L, mat = 0;
    while (ti > 0 || tj > 0) {
      int bits;
      if (ti > 0 && tj > 0) bits = (int)((trace[(size_t)ti * nThreads] >> (5 * (tj - ti + 5))) & 31);
      else if (ti == 0) bits = 2 | (tj == 1 ? 8 : 0);   // row 0: f >= e always; f opens from m only at column 1
      else bits = (ti == 1 ? 4 : 0);                     // column 0: e > f; e opens from m only at row 1
      int op;  // 0 match 1 mismatch 2 insert 3 delete
      int refPos;
      if (mat == 0) {
        if (ti > 0 && tj > 0 && (bits & 1)) { op = (bits & 16) ? 0 : 1; refPos = o.seqStart + tj - 1; --ti; --tj; }
        else { mat = (bits & 2) ? 2 : 1; continue; }
      } else if (mat == 1) {
        op = 2; refPos = o.seqStart + tj;
        if (ti > 0) { if (bits & 4) mat = 0; --ti; } else mat = 2;
      } else {
        op = 3; refPos = o.seqStart + tj - 1;
        if (tj > 0) { if (bits & 8) mat = 0; --tj; } else mat = 1;
      }
      if (P.relax) {
        bool ex = refPos < alleleLen ? t1k_bit(P.ref.exon, goff + refPos) != 0 : false;
        if (!ex || op == 0) ++relaxed;
      }
      if (op == 0 && w) {
        const int readPos = o.readStart + ti;  // ti was already decremented: this column consumed read base ti
        if (!t1k_bit(rn, readPos) && !t1k_bit(P.ref.nmask, goff + refPos)) {
          if (refPos == runLo - 1) runLo = refPos;
          else {
            if (runLo >= 0) { atomicAdd(&cov[runLo], w); atomicAdd(&cov[runHi + 1], -w); }
            runLo = runHi = refPos;
          }
        }
      }
    }
    if (runLo >= 0) { atomicAdd(&cov[runLo], w); atomicAdd(&cov[runHi + 1], -w); }
    P.ovl[gid].relaxed = (uint16_t)(P.relax ? 2 * relaxed : (int)o.matchCnt);
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
size_t t1k_wg_groups_u32() { return (size_t)TILE_ALLELES * 4; }
size_t t1k_wg_thread_u32() { return (size_t)WG * THREAD_SCRATCH_U32; }
size_t t1k_wg_big_u32() { return (size_t)3 * BIG_CAP + GA_SCRATCH_INTS; }
size_t t1k_wg_cache_u64() { return (size_t)GAP_CACHE + GAP_CACHE / 2; }
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
void t1k_launch_fullalign_eq(t1k_ctx *ctx, const SlowArgs &a, int nBlocks) { hipLaunchKernelGGL(k_fullalign_eq, dim3(nBlocks), dim3(WG), 0, ctx->stream, a); }
void t1k_launch_truncate(t1k_ctx *ctx, const TruncArgs &a, int nWg) {
  hipFuncSetAttribute((const void *)k_truncate, hipFuncAttributeMaxDynamicSharedMemorySize, SELECT_LDS_BYTES);
  hipLaunchKernelGGL(k_truncate, dim3(nWg), dim3(WG), SELECT_LDS_BYTES, ctx->stream, a);
}
void t1k_launch_coverage_scan(t1k_ctx *ctx, const T1kRefDev &ref, int32_t *out, const uint64_t *outOff) {
  hipLaunchKernelGGL(k_coverage_scan, dim3(ref.nAlleles), dim3(WG), 0, ctx->stream, ref, out, outOff);
}
