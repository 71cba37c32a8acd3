// t1k_amd/csrc/t1k_chain.hip -- seeding, hit grouping and chaining of SeqSet::AssignRead on gfx950, as a sequence of flat,
// batch-wide kernels (all integer, HBM/LDS-bound; no MFMA):
//
//   k_seed_groups   one 256-thread workgroup per read-end: rolling 11-mers + direct-address look-up with the >=100 skip
//                   rule (GetHitsFromRead, SeqSet.hpp:1071-1229); per (strand, 1024-allele chunk) the hits are folded into
//                   per-allele LDS accumulators (reference diagonal, bitmask of hit offsets, stray counts), i.e. grouped by
//                   (strand, allele) as SortHits 1558-1590 does, and one record per group leaves the chip; groups with fewer
//                   than 3 hits are never written (refMinHitRequired, 1253/1314)
//   k_chain_fast    one lane per (read-end, strand, allele) group over the whole batch: single-diagonal fast path
//                   (GetOverlapsFromHits 1232-1556 + seed-chain match count 1697-1848); alignments that need a DP are
//                   registered in the read-end's memo table and the group is parked
//   k_dp_dense      one lane per distinct registered alignment (banded forward sweep, band in registers)
//   k_chain_retry   parked groups finish from the memo
//   k_chain_general groups with hits on several nearby diagonals: the reference's diagonal-run / LIS logic verbatim
//   k_collect       one workgroup per read-end: strand vote (1619-1648) and copy-out of the winning strand's candidates
#include <algorithm>
#include "t1k_dev.h"
#include "t1k_launch.h"
#include "t1k_memo.h"
#include "t1k_group.h"

#define WG 256
#define GROUP_FAST_MAXLEN 320
#define GENERAL_CAP 128             // hits per group handled by k_chain_general in private memory; larger groups: k_chain_big
#define BIG_CAP 16384
#define GA_BIG_MAX 2048
#define GA_SCRATCH_INTS (6 * (GA_BIG_MAX + 4))

enum { ERR_HITCAP = 1, ERR_STAGECAP = 2, ERR_CANDCAP = 4, ERR_BIGGROUP = 8, ERR_OVLCAP = 16, ERR_SORTCAP = 32, ERR_SLOWCAP = 64, ERR_ROWCAP = 128, ERR_GROUPCAP = 256, ERR_MEMO = 512 };

// ------------------------------------------------------------------------------------------------------------------
// group -> candidate overlaps
// ------------------------------------------------------------------------------------------------------------------
struct CandOut {  // packed candidates: 3 u32 each (+ 3 u32 of memo references when stride == 6, multi-diagonal groups)
  uint32_t *dst;
  int n;
  int stride = 3;
  int cap = 0x7FFFFFFF;
  bool overflow = false;
  __device__ void push(int rs, int re, int ss, int se, int m0, int m) {
    if (n >= cap) { overflow = true; return; }
    dst[stride * n + 0] = (uint32_t)rs | ((uint32_t)re << 12);
    dst[stride * n + 1] = (uint32_t)ss | ((uint32_t)m0 << 20);
    dst[stride * n + 2] = (uint32_t)se | ((uint32_t)m << 20);
    if (stride == 6) dst[6 * n + 3] = dst[6 * n + 4] = dst[6 * n + 5] = 0;
    ++n;
  }
  // memo slots whose match counts are still missing from the last pushed candidate's matchCnt (count in bits 24..27 of word 0)
  __device__ void setRefs(const uint32_t *refs, int nref) {
    if (overflow || stride != 6 || n == 0) return;
    uint32_t *d = dst + 6 * (n - 1);
    d[0] |= (uint32_t)nref << 24; d[3] = refs[0]; d[4] = refs[1]; d[5] = refs[2];
  }
};

// seed-chain match count of one gap (SeqSet.hpp:1710-1752 / 1794-1824)
__device__ inline int gapMatches(const ReadCtx &c, int ra, int ga, int lp, int lt, int *gaScratch, int gaMax, unsigned int *dpCounter, unsigned long long *errFlags,
                                 bool *needScratch) {
  if (lp == lt) return t1k_ga_matches_window(c.rb, c.rn, ra, c.gb, c.gn, c.goff + ga, lp, dpCounter);
  if (lt == 0 || lp == 0) return 0;
  if (dpCounter) ++*dpCounter;
  T1kSeqView T{c.gb, c.gn, c.goff + ga}, P{c.rb, c.rn, ra};
  const int diff = lt > lp ? lt - lp : lp - lt;
  if (diff <= 4) return t1k_ga_band<4, false>(T, lt, P, lp, nullptr, 0);  // band in registers
  if (!gaScratch) { *needScratch = true; return 0; }
  int nm = 0;
  if (lt > gaMax) { atomicOr(errFlags, (unsigned long long)ERR_BIGGROUP); return 0; }
  t1k_ga_general(T, lt, P, lp, gaScratch, nullptr, &nm);
  return nm;
}

// Single-diagonal group (the common case: the read differs from the allele by substitutions only).
// NW = number of 32-position words covering a read (5: reads <= 160 bp, 10: reads <= 320 bp).
//   * input: the group's reference diagonal and the bitmask M of read offsets hitting it, accumulated by k_seed_groups; the
//     caller has checked that every other hit lies more than `radius` diagonals away (it cannot join this run,
//     SeqSet.hpp:1360-1392) and that there are at most two such strays (they cannot form a run of >= 3 hits, 1400-1405)
//   * on one diagonal the LIS is the identity and both hit lengths are equal; with M = bitmask of hit read-offsets,
//     covered = popcount(dilate(M, k)) and matchCnt = 2*covered + 2*sum over gaps of GlobalAlignment matches (1697-1760)
//   * a gap with x <= 3 mismatches aligns ungapped (g - x matches, exact, see t1k_ga_matches_window); if the whole span
//     has <= 3 mismatches every gap does and matchCnt = 2*(span - mismatches) in closed form
//   * exact prune: a gap with x > 3 yields at most g - 1 matches, so U = 2*(span - sum_{x<=3} x - #{x>3}) bounds matchCnt;
//     if U / (2*span) < -s the candidate is certain to fail the similarity filter (1838-1840, 1894-1908) and is emitted
//     with matchCnt = U (it is dropped by k_extend either way, and the strand vote only reads matchCnt0)
// multi-word shift-left-or: C |= C << sft (sft < 64), MW 64-bit words, low word first
template <int MW>
__device__ __forceinline__ void shlOr(uint64_t *C, int sft) {
#pragma unroll
  for (int w = MW - 1; w >= 0; --w) {
    uint64_t carry = w > 0 ? (C[w - 1] >> (64 - sft)) : 0ull;
    C[w] |= (C[w] << sft) | carry;
  }
}

// return value: 5 = needs the gap walk (CLOSED only), 1 = finished (out holds 0 or 1 candidate), 2 = run again after the dense DP phase, 3 = candidate pushed with a
// partial matchCnt, refs[] name the memo slots whose match counts are still to be added (DEFER only)
#define GROUP_MAX_REFS 4   // two 32-bit words of 16-bit memo slots in a group record
template <int NW, bool DEFER, bool CLOSED, bool ONDEMAND = false>
__device__ inline int groupFastPath(const uint32_t *Mw, int diag, const ReadCtx &c, bool hasN, int k, int hitLenRequired, double simThreshold, CandOut &out,
                                    unsigned int *dpCounter, int strandBit, const GapSink &sink, uint32_t *refs, int *nRefs, int earlyPrune = 1) {
  constexpr int MW = (NW + 1) / 2;  // 64-bit words of the read-offset bitmask
  uint64_t M[MW];
  int onDiag = 0;
#pragma unroll
  for (int i = 0; i < MW; ++i) {
    M[i] = (uint64_t)Mw[2 * i] | ((2 * i + 1 < NW) ? ((uint64_t)Mw[2 * i + 1] << 32) : 0ull);
    onDiag += __popcll(M[i]);
  }
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
  // covered positions: dilate M by k (bit p set iff some hit offset a has a <= p < a + k), by doubling
  uint64_t C[MW];
#pragma unroll
  for (int w = 0; w < MW; ++w) C[w] = M[w];
  {
    int have = 1;
    while (2 * have <= k) { shlOr<MW>(C, have); have *= 2; }
    if (have < k) shlOr<MW>(C, k - have);
  }
  int cov = 0;
#pragma unroll
  for (int w = 0; w < MW; ++w) cov += __popcll(C[w]);
  // (the dilation may spill past NW*32 only for offsets that cannot occur: a + k <= len <= NW*32)
  if (cov < hitLenRequired) return 1;  // GetTotalHitLengthOnRead/OnSeq (1512-1522)
  const int spanEnd = last + k, span = spanEnd - first;
  // mismatch bits of the span on this diagonal, relative to `first`: bit 2*j of mmw[i] <-> read position first + 32*i + j.
  // Both windows are fetched as NW + 1 consecutive words and funnel-shifted in registers (one burst of independent loads).
  uint64_t mmw[NW];
  int mmT = 0;
  {
    const int64_t g0 = c.goff + (first - diag);
    const int64_t gw = g0 >> 5;
    const int gsh = (int)(g0 & 31) * 2, rsh = (first & 31) * 2;
    const int rw = first >> 5;
    const int nWin = (span + 31) >> 5;
    constexpr int NP = (NW + 2) / 2;  // 16-byte pieces covering NW + 1 words
    if (ONDEMAND) {
      // (the seeding kernel's call: the read's words are in LDS -- c.rb / c.rn point there -- and are fetched where they are used; only the
      // allele's words come as one burst; the allele's N-mask words, rarely needed, are loaded where they are used as well.  Half the
      // registers of the form below, the same words)
      uint64_t G[2 * NP];
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        t1k_u64x2 g = {0ull, 0ull};
        if (2 * j <= nWin) g = *(const t1k_u64x2 *)(c.gb + gw + 2 * j);
        G[2 * j] = g.x; G[2 * j + 1] = g.y;
      }
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        mmw[i] = 0;
        if (i < nWin) {
          const uint64_t r0 = c.rb[rw + i], r1 = c.rb[rw + i + 1], n0 = c.rn[rw + i], n1 = c.rn[rw + i + 1];
          const uint64_t g = (G[i] >> gsh) | ((G[i + 1] << 1) << (63 - gsh)), r = (r0 >> rsh) | ((r1 << 1) << (63 - rsh));
          uint64_t nn = (n0 >> rsh) | ((n1 << 1) << (63 - rsh));
          if (hasN) { const uint64_t q0 = c.gn[gw + i], q1 = c.gn[gw + i + 1]; nn |= (q0 >> gsh) | ((q1 << 1) << (63 - gsh)); }
          const uint64_t xo = g ^ r;
          uint64_t mm = (xo | (xo >> 1)) & T1K_EVEN & ~nn;
          const int rem = span - 32 * i;
          if (rem < 32) mm &= t1k_lowmask(rem);
          mmw[i] = mm;
          mmT += __popcll(mm);
        }
      }
    } else {
    uint64_t G[2 * NP], R[2 * NP], GN[2 * NP], RN[2 * NP];
    // (the allele's words from the transposed copy where there is one: lanes = consecutive alleles, the same word of each, one row -- see T1kRefDev::basesT)
    const uint64_t *gt = c.gT ? c.gT + (gw - (c.goff >> 5)) * 64 : nullptr;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const bool need = 2 * j <= nWin;  // the arrays carry spare words, but do not stream what is never used
      t1k_u64x2 g = {0ull, 0ull}, r = {0ull, 0ull}, rn2 = {0ull, 0ull};
      if (need) {
        if (gt) { g.x = gt[(2 * j) * 64]; g.y = gt[(2 * j + 1) * 64]; } else g = *(const t1k_u64x2 *)(c.gb + gw + 2 * j);
        r = *(const t1k_u64x2 *)(c.rb + rw + 2 * j); rn2 = *(const t1k_u64x2 *)(c.rn + rw + 2 * j);
      }
      G[2 * j] = g.x; G[2 * j + 1] = g.y; R[2 * j] = r.x; R[2 * j + 1] = r.y; RN[2 * j] = rn2.x; RN[2 * j + 1] = rn2.y;
      GN[2 * j] = GN[2 * j + 1] = 0ull;
    }
    if (hasN) {
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        if (2 * j <= nWin) { const t1k_u64x2 g = *(const t1k_u64x2 *)(c.gn + gw + 2 * j); GN[2 * j] = g.x; GN[2 * j + 1] = g.y; }
      }
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      mmw[i] = 0;
      if (i < nWin) {
        const uint64_t g = (G[i] >> gsh) | ((G[i + 1] << 1) << (63 - gsh)), r = (R[i] >> rsh) | ((R[i + 1] << 1) << (63 - rsh));
        const uint64_t nn = ((GN[i] >> gsh) | ((GN[i + 1] << 1) << (63 - gsh))) | ((RN[i] >> rsh) | ((RN[i + 1] << 1) << (63 - rsh)));
        const uint64_t xo = g ^ r;
        uint64_t mm = (xo | (xo >> 1)) & T1K_EVEN & ~nn;
        const int rem = span - 32 * i;
        if (rem < 32) mm &= t1k_lowmask(rem);
        mmw[i] = mm;
        mmT += __popcll(mm);
      }
    }
    }
  }
  int matchCnt;
  int result = 1;
  // next uncovered / covered position >= from (lowest clear / set bit of C at or above it)
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
  auto mmIn = [&](int gsAbs, int geAbs) -> int {  // mismatches in read positions [gsAbs, geAbs)
    const int gs = gsAbs - first, ge = geAbs - first;
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
  if (mmT <= 3) matchCnt = 2 * (span - mmT);
  else if (CLOSED) {
    // first pass: the closed form ... and the groups that are certain to fail the similarity filter whatever their gaps hold.  Every gap
    // (maximal uncovered run inside the span) with a mismatch costs at least one match -- x of them if x <= 3, one or more if x > 3 --
    // so U' = 2 * (span - #gaps holding a mismatch) bounds the exact prune bound U of the gap walk from above; U' / (2 * span) < s
    // means the candidate is dropped (k_collect), and like the walk's pruned candidates it leaves with the bound as its matchCnt
    // (nothing reads the matchCnt of a dropped candidate; the strand vote reads matchCnt0 = 2 * cov).  The gaps are counted without a
    // walk: relative to `first`, X = uncovered positions, E = mismatching positions (a subset of X: covered positions are exact
    // k-mer hits); in X + E every run of X that holds a bit of E carries out exactly once into the covered position above it.
    uint64_t X[MW], E[MW];
    {
      const int ws = first >> 6, bs = first & 63;
#pragma unroll
      for (int w = 0; w < MW; ++w) {
        uint64_t lo = 0, hi = 0;
#pragma unroll
        for (int v = 0; v < MW; ++v) {
          if (v == w + ws) lo = C[v];
          if (v == w + ws + 1) hi = C[v];
        }
        const uint64_t crel = bs ? ((lo >> bs) | (hi << (64 - bs))) : lo;
        const int rem = span - 64 * w;
        X[w] = rem <= 0 ? 0ull : (~crel & (rem >= 64 ? ~0ull : ((1ull << rem) - 1ull)));
        uint64_t e = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (2 * w + h < NW) {
            uint64_t x = mmw[2 * w + h] & 0x5555555555555555ull;
            x = (x | (x >> 1)) & 0x3333333333333333ull; x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full; x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
            x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull; x = (x | (x >> 16)) & 0xFFFFFFFFull;
            e |= x << (32 * h);
          }
        }
        E[w] = e & X[w];
      }
    }
    int nGaps = 0;
    uint64_t carry = 0;
#pragma unroll
    for (int w = 0; w < MW; ++w) {
      const uint64_t s1 = X[w] + E[w], s2 = s1 + carry;
      carry = ((s1 < X[w]) | (s2 < s1)) ? 1ull : 0ull;
      nGaps += __popcll(s2 & ~X[w]);
    }
    const int bound = 2 * (span - nGaps);
    if (earlyPrune && (double)bound / (double)(2 * span) < simThreshold) matchCnt = bound;
    else if (earlyPrune < 2) return 5;  // the gap walk decides (compacted list of these groups)
    else {
      // ... unless its first pass already does: per gap the mismatch count (all in the registers this pass holds).  No gap with more than
      // three mismatches => every gap aligns ungapped and the count is exact; or the walk's own bound U = 2 * (span - sum of the small gaps'
      // mismatches - #big gaps) fails the similarity filter => certain to be dropped.  Round 4 counted 21 % + 39 % of the groups that used to
      // go to k_chain_fast<*, 1> in these two kinds (profiles/r04_walk_classes.txt); only the rest needs the memo and is listed for it.
      int sumSmall = 0, nBig = 0;
      int pos = first;
      while (true) {
        int gs = nextClear(pos);
        if (gs >= spanEnd) break;
        int ge = nextSet(gs);
        if (ge > spanEnd) ge = spanEnd;
        const int x = mmIn(gs, ge);
        if (x <= 3) sumSmall += x; else ++nBig;
        pos = ge;
      }
      const int upper = 2 * (span - sumSmall - nBig);
      if (nBig != 0 && !((double)upper / (double)(2 * span) < simThreshold)) return 5;
      matchCnt = upper;
    }
  }
  else {
    // walk the gaps (maximal uncovered runs inside the span)
    int sumSmall = 0, nBig = 0;
    int pos = first;
    // pass 1: per-gap mismatch counts -> upper bound
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
#ifdef T1K_WALK_STATS
    { const int cls = nBig == 0 ? 0 : ((double)upper / (double)(2 * span) < simThreshold) ? 1 : 2;
      for (int c = 0; c < 3; ++c) { const unsigned long long mk = __ballot(cls == c); if (mk && (int)(threadIdx.x & 63) == __ffsll((long long)mk) - 1) atomicAdd(&sink.counters[40 + c], (unsigned long long)__popcll(mk)); } }
#endif
    if (nBig == 0) matchCnt = upper;  // exact
    else if ((double)upper / (double)(2 * span) < simThreshold) matchCnt = upper;  // certain to be dropped; no DP needed
    else {
      int gapMatch = 0, nref = 0;
      bool again = false;
      pos = first;
      while (true) {
        int gs = nextClear(pos);
        if (gs >= spanEnd) break;
        int ge = nextSet(gs);
        if (ge > spanEnd) ge = spanEnd;
        int x = mmIn(gs, ge);
        if (x <= 3) gapMatch += (ge - gs) - x;
        else {
          uint32_t slot = 0;
          int r = gapMatchesCached<DEFER>(c, gs, c.goff + (gs - diag), ge - gs, ge - gs, strandBit, sink, dpCounter, &slot);
          if (r >= 0) gapMatch += r;
          else if (r == -1 && nref < GROUP_MAX_REFS) { refs[nref >> 1] |= slot << (16 * (nref & 1)); ++nref; }
          else again = true;  // keep walking: later gaps register their jobs too
        }
        pos = ge;
      }
      if (again) return 2;
      matchCnt = 2 * cov + 2 * gapMatch;
      if (nref) { *nRefs = nref; result = 3; }
    }
  }
  out.push(first, last + k - 1, first - diag, last - diag + k - 1, 2 * cov, matchCnt);
  return result;
}

// General group (several diagonals): restates GetOverlapsFromHits 1338-1551 and the chain walk 1697-1833.
// A[n] sorted copy of the hits, B[n] concordant hits, C[n] packs top (low 16) / link (high 16) of the LIS.
// Arr: work-array accessor (plain pointer, or LaneArr = LDS arrays interleaved over the lanes of a wavefront)
// second half of one diagonal run: B[0..m) = the run's hits nearest to the dominant diagonal, sorted by (allele offset, read offset);
// LIS over the read offsets, chain -> A[s..), hit lengths, seed-chain match count, candidate (SeqSet.hpp:352-436, 1512-1551, 1697-1833)
template <class Arr>
__device__ inline void chainRun(const ReadCtx &c, int k, int hitLenRequired, Arr A, Arr B, Arr C, int s, int m, int *gaScratch, int gaMax, CandOut &out,
                                unsigned int *dpCounter, unsigned long long *errFlags, bool *needScratch, const GapSink *sink, int strandBit, bool chainReady = false) {
    int ret, lenR, lenS;
    if (!t1k_run_lis(A, B, C, s, m, k, hitLenRequired, &ret, &lenR, &lenS, chainReady)) return;
    // seed-chain match count (1697-1833).  With a sink the alignments are registered in the read-end's memo instead of being
    // run here; the candidate then carries the memo slots (k_general_finish adds their match counts).
    uint32_t refs[3] = {0, 0, 0};
    int nref = 0;
    auto gapM = [&](int ra, int ga, int lp, int lt) -> int {
      if (!sink) return gapMatches(c, ra, ga, lp, lt, gaScratch, gaMax, dpCounter, errFlags, needScratch);
      if (lp <= 0 || lt <= 0) return 0;
      const int diff = lt > lp ? lt - lp : lp - lt;
      if (diff > 4) { *needScratch = true; return 0; }
      uint32_t slot = 0;
      const int r = gapMatchesCached<true>(c, ra, c.goff + ga, lp, lt, strandBit, *sink, dpCounter, &slot);
      if (r >= 0) return r;
      if (r == -1 && nref < GROUP_MAX_REFS) { refs[nref >> 1] |= slot << (16 * (nref & 1)); ++nref; return 0; }
      if (dpCounter) ++*dpCounter;
      return gapAlign(c, ra, c.goff + ga, lp, lt);
    };
    int matchCnt = 2 * k;
    for (int i = 1; i < ret; ++i) {
      int pa = (int)(A[s + i - 1] & 0xFFF), pb = (int)(A[s + i - 1] >> 12), qa = (int)(A[s + i] & 0xFFF), qb = (int)(A[s + i] >> 12);
      bool sameDiag = (pb - pa) == (qb - qa);
      bool readOv = pa + k - 1 >= qa, seqOv = pb + k - 1 >= qb;
      if (sameDiag) {
        if (readOv) matchCnt += 2 * (qa - pa);
        else matchCnt += 2 * k + 2 * gapM(pa + k, pb + k, qa - (pa + k), qb - (pb + k));
      } else {
        if (readOv && !seqOv) matchCnt += 2 * (qa - pa);
        else if (!readOv && seqOv) matchCnt += 2 * (qb - pb);
        else if (readOv && seqOv) matchCnt += 2 * ((qa - pa) < (qb - pb) ? (qa - pa) : (qb - pb));
        else matchCnt += 2 * k + 2 * gapM(pa + k, pb + k, qa - (pa + k), qb - (pb + k));
      }
    }
    int rs = (int)(A[s] & 0xFFF), ss = (int)(A[s] >> 12);
    int re = (int)(A[s + ret - 1] & 0xFFF) + k - 1, se = (int)(A[s + ret - 1] >> 12) + k - 1;
    out.push(rs, re, ss, se, 2 * lenR, matchCnt);
    out.setRefs(refs, nref);
}

template <class Arr>
__device__ inline void groupGeneral(const uint32_t *h, int n, const ReadCtx &c, int k, int radius, int hitLenRequired, Arr A, Arr B,
                                     Arr C, int *gaScratch, int gaMax, CandOut &out, unsigned int *dpCounter, unsigned long long *errFlags, bool *needScratch,
                                     const GapSink *sink = nullptr, int strandBit = 0) {
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
    chainRun(c, k, hitLenRequired, A, B, C, s, m, gaScratch, gaMax, out, dpCounter, errFlags, needScratch, sink, strandBit);
    s = e;
  }
}

// A multi-diagonal group whose hit list k_near_hits found to be a chain already (two diagonals, the hits of one entirely before the
// other's on the read AND on the allele; written in that order): one diagonal run (the diagonals lie within `radius`), every read offset
// occurs once so the nearest-to-dominant filter keeps everything, the order by allele offset is the order given, its read offsets ascend
// strictly so the LIS is the whole list, no allele offset repeats.  What is left of GetOverlapsFromHits are the length tests and the
// chain walk -- the same code as the general path runs after its sorts (SeqSet.hpp:1400-1405, 1468, 1512-1551, 1697-1833).
template <class Arr>
__device__ inline void groupSimple(const uint32_t *h, int n, const ReadCtx &c, int k, int hitLenRequired, Arr A, CandOut &out, unsigned int *dpCounter,
                                   unsigned long long *errFlags, bool *needScratch, const GapSink *sink, int strandBit) {
  for (int i = 0; i < n; ++i) A[i] = h[i];
  if (n < 3 || n * k < hitLenRequired) return;
  chainRun(c, k, hitLenRequired, A, A, A, 0, n, nullptr, 0, out, dpCounter, errFlags, needScratch, sink, strandBit, true);
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


// ------------------------------------------------------------------------------------------------------------------
// K1: seeds -> one record per (strand, allele) group, built in LDS in a single pass over the posting lists
//
// Alleles are processed in chunks of CHUNK_A; each allele of the chunk owns an accumulator in LDS:
//   diag  reference diagonal = diagonal of the first hit that arrived (any choice is valid, see groupFastPath)
//   M     bitmask of the read offsets whose k-mer hits the allele on that diagonal
//   meta  number of hits on other diagonals (low 16 bits) and how many of those lie within `radius` (high 16 bits)
// Posting lists are sorted by allele, so the chunk's slice of every used list is found by binary search from a cursor.
// A group record leaves the chip once, coalesced; no hit list is ever written.
// ------------------------------------------------------------------------------------------------------------------
#define CHUNK_A T1K_SEED_CHUNK
// record word 2 before chaining: reference diagonal (22 bits, biased; alleles are shorter than 2^20 bases) and the counts of hits off
// it: FAR = beyond `radius` diagonals (bits 22..24, saturating at 7) and NEAR = within `radius` (bits 25..29, saturating at 31) -- the
// chain asks "near > 0" and "far > 2" (several diagonals: the general path), k_near_hits asks for "far == 0" and the exact near count
// (31 = unknown).  Bits 30 and 31 stay clear: after chaining the word holds the state, whose REC_DONE is bit 31.
__device__ __forceinline__ uint32_t packDiagMeta(int diag, uint32_t meta) {
  const uint32_t strays = meta & 0xFFFFu, nearCnt = meta >> 16;  // (strays counts every hit off the reference diagonal, near ones included)
  return (uint32_t)(diag + (1 << 21)) | (min(strays - nearCnt, 7u) << 22) | (min(nearCnt, 31u) << 25);
}
__device__ __forceinline__ int recDiag(uint32_t w2) { return (int)(w2 & 0x3FFFFFu) - (1 << 21); }
__device__ __forceinline__ uint32_t recFar(uint32_t w2) { return (w2 >> 22) & 7u; }
__device__ __forceinline__ uint32_t recNear(uint32_t w2) { return (w2 >> 25) & 31u; }
__device__ __forceinline__ bool recIsGeneral(uint32_t w2) { return recNear(w2) > 0 || recFar(w2) > 2; }
#define REC_NEAR_DONE 0x4E454152u  // record word 5 of a multi-diagonal group whose hit list k_near_hits wrote (k_gather_general leaves it alone)
enum { REC_DONE = 0x80000000u };  // word 3 after chaining: REC_DONE | number of candidates (general groups: candidates in the side arena)

__device__ __forceinline__ ReadCtx makeCtx(const ChainArgs &P, uint32_t re, int pass, uint32_t allele) {
  const int S = P.reads.S;
  ReadCtx c{P.reads.bases + ((uint64_t)re * 2 + pass) * S, P.reads.nmask + ((uint64_t)re * 2 + pass) * S, (int)P.reads.len[re], P.ref.bases, P.ref.nmask,
            (int64_t)P.ref.alleleOff[allele], (int)P.ref.alleleLen[allele], P.ref.anyN != 0};
  if (P.ref.basesT) c.gT = P.ref.basesT + (uint64_t)P.ref.blockT[allele >> 6] + (allele & 63u);
  return c;
}

#define DIAG_EMPTY 0x7FFFFFFF

// the closed-form pass on one accumulator (fused seeding): `a` = the accumulator row in LDS (diag, meta, M[NW]); returns groupFastPath's verdict, 6 instead of
// 1 when the group ended WITH a candidate, which is left in a[2..4]
template <int NW>
__device__ __forceinline__ uint32_t closedFormGroup(uint32_t *a, const uint64_t *rb, const uint64_t *rn, const uint64_t *gb, const uint64_t *gn, int64_t goff, bool anyN, bool hasN, int k,
                                                    int hitLenRequired, double sim, int pass, int earlyPrune) {
  uint32_t Mw[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) Mw[w] = a[2 + w];
  const ReadCtx c{rb, rn, 0, gb, gn, goff, 0, anyN};
  uint32_t cbuf[3];
  CandOut out{cbuf, 0};
  const GapSink sink{nullptr, nullptr, nullptr, 0u, 0u, T1K_AR_JOBS};  // (the closed-form pass registers nothing)
  uint32_t refs[2] = {0, 0};
  int nRefs = 0;
  const uint32_t kind = (uint32_t)groupFastPath<NW, true, true, true>(Mw, (int)a[0], c, hasN, k, hitLenRequired, sim, out, nullptr, pass, sink, refs, &nRefs, earlyPrune);
  if (kind == 1u && out.n) { a[2] = cbuf[0]; a[3] = cbuf[1]; a[4] = cbuf[2]; return 6u; }
  return kind;
}

#ifndef T1K_SEED_WAVES
#define T1K_SEED_WAVES 7   // round 6: the kernel's 20.7 KB of LDS admit SEVEN workgroups a compute unit; held to 64 VGPRs for eight it spilled 41 registers for an occupancy it never had (72 VGPRs: 24 spilled; 3.11 -> 2.87 ms per range alone, profiles/r06_callE_seed_waves_suite.log)
#endif
#ifndef T1K_SEED_PACK_Q
#define T1K_SEED_PACK_Q 0
#endif
#ifndef T1K_FUSE_WAVES
#define T1K_FUSE_WAVES 5
#endif
// FUSE: the closed-form pass of the chain (groupFastPath<NW, true, true>, what k_chain_fast<NW, 0> does per record) runs here, on the
// accumulators while they are in LDS: a group that ends there leaves as a finished record (state + candidate) or -- no candidate -- not
// at all, a group for the gap walk / the multi-diagonal path leaves as before AND is put on its work list.  No launch reads the
// records back just to classify them.
template <int NW, bool FUSE>
__device__ __forceinline__ void seedGroupsBody(const ChainArgs &P) {
  constexpr int AW = NW == 5 ? 7 : 13;  // u32 per accumulator: diag, meta, M[NW]; odd stride = no LDS bank conflicts
  extern __shared__ uint32_t lds[];
  const int k = P.k;
  const int maxK = (int)P.maxKFast;                 // >= k-mers of a read-end this kernel seeds, both strands (LDS layout; the used-list table's stride is P.maxK)
  uint32_t *acc = lds;                              // [CHUNK_A][AW] per-allele accumulators of the current chunk
  // look-up phase only (overlaid on the accumulators, which are re-initialised afterwards):
  uint32_t *ukCode = acc;                           // [maxK]  code | valid << 31
  uint32_t *ukStart = ukCode + maxK;                // [maxK]
  uint32_t *ukLen = ukStart + maxK;                 // [maxK]
  uint32_t *ukDir = ukLen + maxK;                   // [maxK]  chunk-directory row of the list
  uint16_t *usedQ = (uint16_t *)(ukDir + maxK);     // [maxK]  k-mers whose lists are used, + strand first
  // chunk loop:
  uint32_t *sLo = acc + CHUNK_A * AW;               // [maxK]  slice of the current chunk
  uint32_t *pre = sLo + maxK;                       // [maxK + 1] exclusive prefix of the slice lengths
  uint32_t *lstStart = pre + maxK + 1;              // [maxK]  posting-list start / length of the used lists (both strands)
  uint32_t *lstLen = lstStart + maxK;               // [maxK]
  uint32_t *lstDir = lstLen + maxK;                 // [maxK]
  // read offset of the used lists: T1K_SEED_PACK_Q (round 6) keeps it in the upper nine bits of lstLen -- a list's length is only asked for "is it
  // empty" (lists with a directory row) or is below T1K_DIR_MINLEN (the others), so 23 bits hold all that is read -- and the 560 bytes of its own array
  // go: 20 756 -> 20 196 bytes of LDS per workgroup for 2 x 150 bp reads, under the 20 480 at which EIGHT workgroups fit a compute unit's 160 KB
  // (the register allocation has been held to 64 VGPRs for eight wavefronts per SIMD all along; the LDS admitted seven)
#if T1K_SEED_PACK_Q
  uint16_t *qOf = (uint16_t *)(lstDir + maxK);      // (no array: the pointer only marks where the bitmaps may start)
#define LST_LEN(x) ((x) & 0x7FFFFFu)
#else
  uint16_t *qOf = (uint16_t *)(lstDir + maxK);      // [maxK]  read offset of the used lists
#define LST_LEN(x) (x)
#endif
  // two bitmaps over all alleles (chunk selection, before the chunk loop of each strand): over the accumulators when they fit there
  // (references of up to 57 344 / 106 496 sequences), else behind the list arrays (the launcher sizes the dynamic LDS for it)
  const uint32_t A = P.ref.nAlleles;
  const uint32_t nChunks = P.ref.kDirStride - 1;    // (the launcher refuses more than 256 chunks: sHot)
  uint32_t *bitmaps = 2 * ((A + 31) >> 5) <= (uint32_t)(CHUNK_A * AW) ? acc : (uint32_t *)(qOf + (T1K_SEED_PACK_Q ? 0 : ((maxK + 1) & ~1)));
  __shared__ uint32_t warpSums[4];
  __shared__ uint32_t sHot[8];                      // bit c: chunk c can hold an allele with three hits (this strand)
  __shared__ uint32_t sUsed[2], sGroupBase, sFallback, sPost;
  __shared__ int sWaveMax[4];
  const int tid = threadIdx.x;
  const uint32_t kmask = (1u << (2 * k)) - 1;
  const uint32_t stride = P.recStride;
  for (uint32_t i = tid; i < CHUNK_A * AW; i += WG) acc[i] = (i % AW) == 0 ? (uint32_t)DIAG_EMPTY : 0u;
  __syncthreads();
#ifdef T1K_SEED_PROFILE
  uint64_t tp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_ = __builtin_amdgcn_s_memtime();
#endif
  __shared__ uint32_t sUMask[2 * T1K_USED_MASK_WORDS];  // read offsets whose lists are used, per strand
  constexpr int RDW = 14;                             // >= S + 2 (S <= 11: reads of at most T1K_MAX_READ_LEN bases)
  __shared__ uint64_t sRd[2][2][FUSE ? RDW : 1];      // [strand][bases | N mask][word] of the read-end (FUSE)
  __shared__ unsigned long long sStat[3];             // lookups, postings, hits: thread 0 tallies them in LDS (three 64-bit counters in registers would be held by every lane), flushed once per workgroup
  if (tid == 0) { sStat[0] = 0; sStat[1] = 0; sStat[2] = 0; }
  unsigned int fastLocal = 0, groupsLocal = 0, recsLocal = 0;  // (FUSE)
  for (uint32_t re = blockIdx.x; re < P.reads.nReadEnds; re += gridDim.x) {
    const int len = P.reads.len[re];
    const int S = P.reads.S;
    const uint64_t *rbase = P.reads.bases + (uint64_t)re * 2 * S;
    const uint64_t *rnm = P.reads.nmask + (uint64_t)re * 2 * S;
    for (int c = tid; c < P.maxChunks; c += WG) P.chunkCount[(uint64_t)re * P.maxChunks + c] = 0;
    if (tid == 0) { P.usedCount[2 * re] = 0; P.usedCount[2 * re + 1] = 0; }
    // (a read-end beyond the hit masks' span is seeded by k_seed_long, launched behind this kernel)
    // (... and one whose lists an earlier window of the job holds is not seeded at all: t1k_xwin_link)
    if (len < k || len > T1K_MAX_READ_LEN || (P.reads.skip && P.reads.skip[re])) { __syncthreads(); continue; }  // GetOverlapsFromRead returns -1 (SeqSet.hpp:1598-1599)
    const int nk = len - k + 1;
    if (tid < 2 * T1K_USED_MASK_WORDS) sUMask[tid] = 0;  // (read by the previous read-end before its chunk loop's barriers)
    if (FUSE && tid >= 64 && tid < 64 + 2 * 2 * RDW) {  // the read's packed words for the closed-form pass (used behind the barriers of the look-up phase)
      const int x = tid - 64, strand = x / (2 * RDW), kind = (x / RDW) & 1, w = x % RDW;
      sRd[strand][kind][w] = w < S ? (kind ? rnm : rbase)[strand * S + w] : 0ull;
    }
    for (int q = tid; q < 2 * nk; q += WG) {
      int pass = q / nk, p = q - pass * nk;
      const uint64_t *b = rbase + pass * S, *nm = rnm + pass * S;
      uint32_t code = (uint32_t)t1k_get32(b, p) & kmask;
      bool valid = ((uint32_t)t1k_get32(nm, p) & kmask) == 0;
      uint32_t st = 0, ln = 0, dr = T1K_NO_DIR;
      if (valid) { st = P.ref.kStart[code]; ln = P.ref.kStart[code + 1] - st; dr = P.ref.kDirIdx[code]; }
      ukCode[q] = code | (valid ? 0x80000000u : 0);
      ukStart[q] = st; ukLen[q] = ln; ukDir[q] = dr;
    }
    __syncthreads();

#ifdef T1K_SEED_PROFILE
    { const uint64_t tn_ = __builtin_amdgcn_s_memtime(); tp_[0] += tn_ - tl_; tl_ = tn_; }
#endif
    // The look-up rule (SeqSet.hpp:1098-1153, 1165-1226) is a sequential state machine (prevKmerCode, skipCnt).
    // Parallel form: if no two k-mers within k/2 + 1 consecutive positions of a strand are equal, `code != prev` holds at every
    // position (prev is the code of one of the previous k/2 + 1 positions), every position is a look-up, and only skipCnt is
    // left: in a maximal run of "big" positions (list >= 100, not the first / last k-mer) exactly every (k/2 + 1)-th one is
    // used, any other position resets the count.  Reads with such short repeats take the sequential replay below.
    const int W1 = k / 2 + 1;
    if (tid == 0) { sFallback = 0; sUsed[0] = 0; sUsed[1] = 0; sPost = 0; }
    __syncthreads();
    int qv[3], lastNonBig[3];
    uint32_t szv[3];
    bool bigv[3];
    {
      int localMax = -1;
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        const int q = 3 * tid + x;
        qv[x] = q; szv[x] = 0; bigv[x] = false; lastNonBig[x] = -1;
        if (q < 2 * nk) {
          const int pass = q >= nk ? 1 : 0, p = q - pass * nk;
          const uint32_t code = ukCode[q] & 0x7FFFFFFFu;
          for (int d = 1; d <= W1 && d <= p; ++d)
            if ((ukCode[q - d] & 0x7FFFFFFFu) == code) sFallback = 1;
          szv[x] = ukLen[q];
          bigv[x] = szv[x] >= 100 && p != 0 && p != nk - 1;
          if (!bigv[x]) localMax = q;
        }
        lastNonBig[x] = localMax;  // within this lane so far; the lanes before are merged in below
      }
      // inclusive max-scan of localMax over the lanes (three consecutive positions per lane, lanes in position order)
      int incl = localMax;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl = max(incl, y); }
      if ((tid & 63) == 63) sWaveMax[tid >> 6] = incl;
      __syncthreads();
      int before = __shfl_up(incl, 1, 64);
      if ((tid & 63) == 0) before = -1;
      for (int w = 0; w < (tid >> 6); ++w) before = max(before, sWaveMax[w]);
#pragma unroll
      for (int x = 0; x < 3; ++x) lastNonBig[x] = max(lastNonBig[x], before);
    }
    const bool fallback = sFallback != 0;  // (sWaveMax's barrier also published sFallback)
    if (!fallback) {
      uint32_t mine = 0, minePlus = 0, minePost = 0;
      bool usedv[3];
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        usedv[x] = false;
        if (qv[x] < 2 * nk && szv[x]) usedv[x] = !bigv[x] || ((qv[x] - lastNonBig[x]) % W1 == 0);
        if (usedv[x]) { ++mine; minePost += szv[x]; if (qv[x] < nk) ++minePlus; }
      }
      uint32_t tot;
      uint32_t slot = t1k_block_scan_exclusive(mine, warpSums, &tot);
#pragma unroll
      for (int x = 0; x < 3; ++x)
        if (usedv[x]) usedQ[slot++] = (uint16_t)qv[x];
      for (int o = 32; o > 0; o >>= 1) { minePlus += __shfl_xor(minePlus, o, 64); minePost += __shfl_xor(minePost, o, 64); }
      if ((tid & 63) == 0) { atomicAdd(&sUsed[0], minePlus); atomicAdd(&sPost, minePost); }
      __syncthreads();
      if (tid == 0) {
        sUsed[1] = tot - sUsed[0];
        sStat[0] += 2 * nk; sStat[1] += sPost;
        P.usedCount[2 * re] = sUsed[0]; P.usedCount[2 * re + 1] = sUsed[1];
      }
    }
    // sequential replay (reads with short repeats): the first wavefront runs it as uniform (scalar) code: each lane holds one
    // k-mer's code and list length, the loop reads them with v_readlane.
    if (fallback && tid < 64) {
      uint32_t prev = 0;  // prevKmerCode starts at code 0 and is carried from the + strand into the - strand
      uint32_t nUsed = 0;
      uint32_t lookups = 0, postings = 0;
      for (int pass = 0; pass < 2; ++pass) {
        int skipCnt = 0;
        const uint32_t begin = nUsed;
        for (int seg = 0; seg < nk; seg += 64) {
          const int pl = seg + tid;
          const uint32_t vc = pl < nk ? (ukCode[pass * nk + pl] & 0x7FFFFFFFu) : 0u;
          const uint32_t vl = pl < nk ? ukLen[pass * nk + pl] : 0u;
          const int cnt = min(64, nk - seg);
          for (int j = 0; j < cnt; ++j) {
            const uint32_t code = (uint32_t)__builtin_amdgcn_readlane((int)vc, j);
            const uint32_t size = (uint32_t)__builtin_amdgcn_readlane((int)vl, j);
            const int p = seg + j;
            if (p == 0 || code != prev) {
              ++lookups;
              if (size >= 100 && p != 0 && p != nk - 1 && skipCnt < k / 2) { ++skipCnt; continue; }
              skipCnt = 0;
              if (size) {
                if (tid == 0) usedQ[nUsed] = (uint16_t)(pass * nk + p);
                ++nUsed;
                postings += size;
              }
            }
            prev = code;
          }
        }
        if (tid == 0) sUsed[pass] = nUsed - begin;
      }
      if (tid == 0) {
        sStat[0] += lookups; sStat[1] += postings;
        P.usedCount[2 * re] = sUsed[0]; P.usedCount[2 * re + 1] = sUsed[1];
      }
    }
    __syncthreads();
    const uint32_t nUsedPlus = sUsed[0], nUsedMinus = sUsed[1];
#ifdef T1K_SEED_PROFILE
    { const uint64_t tn_ = __builtin_amdgcn_s_memtime(); tp_[1] += tn_ - tl_; tl_ = tn_; }
#endif

    // the used lists are kept for k_chain_general, which re-derives the hits of the few multi-diagonal groups; which read offsets have
    // their lists used goes out per strand as bit masks as well (k_near_hits rebuilds the hits on near diagonals from them; sUMask was
    // cleared at the top of this read-end, before the barriers of the look-up phase)
    {
      uint32_t *uo = P.usedOut + (uint64_t)re * P.maxK * 4;
      for (uint32_t u = tid; u < nUsedPlus + nUsedMinus; u += WG) {
        int q = usedQ[u];
        int pass = u < nUsedPlus ? 0 : 1;
        uo[4 * u] = (uint32_t)(q - pass * nk); uo[4 * u + 1] = ukStart[q]; uo[4 * u + 2] = ukLen[q]; uo[4 * u + 3] = ukDir[q];
        atomicOr(&sUMask[pass * T1K_USED_MASK_WORDS + ((q - pass * nk) >> 5)], 1u << ((q - pass * nk) & 31));
      }
    }
    // what the chunk loop needs of the used lists moves out of the overlay, then the accumulators under it are made clean again
    __syncthreads();
    if (tid < 2 * T1K_USED_MASK_WORDS) P.usedMask[(uint64_t)re * 2 * T1K_USED_MASK_WORDS + tid] = sUMask[tid];
    for (uint32_t u = tid; u < nUsedPlus + nUsedMinus; u += WG) {
      const int q = usedQ[u];
      const int pass = u < nUsedPlus ? 0 : 1;
      const uint32_t st = ukStart[q], ln = ukLen[q];
#if T1K_SEED_PACK_Q
      lstStart[u] = st; lstLen[u] = min(ln, 0x7FFFFFu) | ((uint32_t)(q - pass * nk) << 23); lstDir[u] = ukDir[q];
#else
      lstStart[u] = st; lstLen[u] = ln; lstDir[u] = ukDir[q]; qOf[u] = (uint16_t)(q - pass * nk);
#endif
    }
    __syncthreads();
    for (uint32_t i = tid; i < (uint32_t)((9 * maxK + 1) / 2); i += WG) acc[i] = (i % AW) == 0 ? (uint32_t)DIAG_EMPTY : 0u;
    __syncthreads();

#ifdef T1K_SEED_PROFILE
    { const uint64_t tn_ = __builtin_amdgcn_s_memtime(); tp_[2] += tn_ - tl_; tl_ = tn_; }
#endif
    int chunk = 0;
    for (int sp = 0; sp < 2; ++sp) {  // '-' strand first (SortHits 1577-1583)
      const int pass = sp == 0 ? 1 : 0;
      const uint32_t uBegin = pass == 0 ? 0 : nUsedPlus;
      const uint32_t uCount = pass == 0 ? nUsedPlus : nUsedMinus;
      if (uCount == 0) continue;
      // thread t owns the lists t and t + WG (uCount <= 2 * WG: reads are at most 320 bp)
      const bool has0 = (uint32_t)tid < uCount, has1 = (uint32_t)tid + WG < uCount;
      uint32_t cur0 = 0, cur1 = 0;
      // ---- which chunks can hold a group at all.  A group needs >= 3 hits on its allele.  The k-mers of a read that are not part of
      // a gene's conserved sequence have short lists (a handful of chance postings anywhere in the reference), and there are enough of
      // them to put a posting or two into EVERY chunk: stepping through all chunks for them was most of this kernel's time.  So: a
      // chunk is visited if a long list (one with a directory row: its per-chunk occupancy mask is part of the index) has a posting
      // in it, or if some allele of it collects three postings from the short lists alone -- counted exactly with two bitmaps over
      // all alleles (seen once / seen twice; the third sighting marks the chunk).  Every other chunk holds no allele with three hits
      // and cannot emit a record.  mk0 / mk1: chunk occupancy of this lane's own lists (chunks < 64; beyond that: "maybe").
      unsigned long long mk0 = 0, mk1 = 0;
      {
        const uint32_t BW = (A + 31) >> 5;
        uint32_t *b1 = bitmaps, *b2 = bitmaps + BW;
        for (uint32_t i = tid; i < 2 * BW; i += WG) bitmaps[i] = 0;
        if (tid < 8) sHot[tid] = 0;
        __syncthreads();
        auto mark = [&](bool has, uint32_t st, uint32_t ln, uint32_t row, unsigned long long &mk) {
          if (!has || !ln) return;
          if (row != T1K_NO_DIR) {
            const unsigned long long *m = P.ref.kDirMask + (uint64_t)row * P.ref.kDirMaskWords;
            for (uint32_t w = 0; w < P.ref.kDirMaskWords; ++w) {
              const unsigned long long v = m[w];
              if (w == 0) mk = v;
              if ((uint32_t)v) atomicOr(&sHot[2 * w], (uint32_t)v);
              if ((uint32_t)(v >> 32)) atomicOr(&sHot[2 * w + 1], (uint32_t)(v >> 32));
            }
            if (P.ref.kDirMaskWords > 1) mk = ~0ull;  // (more than 64 chunks: the directory itself answers)
          } else {
            for (uint32_t j0 = 0; j0 < ln; j0 += 8) {  // <= T1K_DIR_MINLEN postings; eight loads in flight
              uint32_t al[8];
#pragma unroll
              for (int x = 0; x < 8; ++x) al[x] = j0 + x < ln ? P.ref.kPostAllele[st + j0 + x] : 0xFFFFFFFFu;
#pragma unroll
              for (int x = 0; x < 8; ++x) {
                if (al[x] == 0xFFFFFFFFu) continue;
                const uint32_t ci = al[x] / CHUNK_A, bit = 1u << (al[x] & 31);
                mk |= ci < 64 ? 1ull << ci : 0ull;
                if (atomicOr(&b1[al[x] >> 5], bit) & bit)
                  if (atomicOr(&b2[al[x] >> 5], bit) & bit) atomicOr(&sHot[ci >> 5], 1u << (ci & 31));
              }
            }
            if (nChunks > 64) mk = ~0ull;
          }
        };
        mark(has0, has0 ? lstStart[uBegin + tid] : 0u, has0 ? LST_LEN(lstLen[uBegin + tid]) : 0u, has0 ? lstDir[uBegin + tid] : T1K_NO_DIR, mk0);
        mark(has1, has1 ? lstStart[uBegin + tid + WG] : 0u, has1 ? LST_LEN(lstLen[uBegin + tid + WG]) : 0u, has1 ? lstDir[uBegin + tid + WG] : T1K_NO_DIR, mk1);
        __syncthreads();
        if (bitmaps == acc) {  // the bitmaps lay over the accumulators: make those clean again
          for (uint32_t i = tid; i < 2 * BW; i += WG) acc[i] = (i % AW) == 0 ? (uint32_t)DIAG_EMPTY : 0u;
          __syncthreads();
        }
      }
      // first posting with allele >= bound in [lo, ln) of a short list (bisection over the allele column)
      auto lowerBound = [&](uint32_t st, uint32_t lo, uint32_t ln, uint32_t bound) -> uint32_t {
        uint32_t hi = ln;
        while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (P.ref.kPostAllele[st + m] < bound) lo = m + 1; else hi = m; }
        return lo;
      };
      for (uint32_t hw = 0; hw < 8; ++hw) {
       uint32_t hotBits = sHot[hw];
       while (hotBits) {
        const uint32_t ci = hw * 32 + (uint32_t)__ffs((int)hotBits) - 1;
        hotBits &= hotBits - 1;
        const uint32_t c0 = ci * CHUNK_A;
        const uint32_t c1 = min(c0 + CHUNK_A, A);
#ifdef T1K_SEED_PROFILE
        tp_[3] += 1;  // chunks visited (not a clock)
#endif
        // slice of every used list inside [c0, c1): long lists from their directory row, short ones by bisection from their cursor
        // (the chunks come in ascending order, so the cursor only moves forward); lists without a posting here are not touched
        // (a lane's list start / length / directory row are read back from LDS where they are needed: held in registers across the
        // chunk loop they were spilled to scratch under the 64-VGPR budget)
        uint32_t n0 = 0, n1 = 0;
        const bool may = ci >= 64;
        if (has0) {
          uint32_t lo = cur0, hi = cur0;
          if (may || ((mk0 >> ci) & 1ull)) {
            const uint32_t row = lstDir[uBegin + tid];
            if (row != T1K_NO_DIR) { const uint32_t *dir = P.ref.kDir + (uint64_t)row * P.ref.kDirStride; lo = dir[ci]; hi = dir[ci + 1]; }
            else { const uint32_t st = lstStart[uBegin + tid], ln = LST_LEN(lstLen[uBegin + tid]); lo = lowerBound(st, cur0, ln, c0); hi = lowerBound(st, lo, ln, c1); }
          }
          n0 = hi - lo; sLo[tid] = lo; cur0 = hi;
        }
        if (has1) {
          uint32_t lo = cur1, hi = cur1;
          if (may || ((mk1 >> ci) & 1ull)) {
            const uint32_t row = lstDir[uBegin + tid + WG];
            if (row != T1K_NO_DIR) { const uint32_t *dir = P.ref.kDir + (uint64_t)row * P.ref.kDirStride; lo = dir[ci]; hi = dir[ci + 1]; }
            else { const uint32_t st = lstStart[uBegin + tid + WG], ln = LST_LEN(lstLen[uBegin + tid + WG]); lo = lowerBound(st, cur1, ln, c0); hi = lowerBound(st, lo, ln, c1); }
          }
          n1 = hi - lo; sLo[tid + WG] = lo; cur1 = hi;
        }

#ifdef T1K_SEED_PROFILE
    { const uint64_t tn_ = __builtin_amdgcn_s_memtime(); tp_[4] += tn_ - tl_; tl_ = tn_; }
#endif
        uint32_t tot0, tot1 = 0;
        const uint32_t e0 = t1k_block_scan_exclusive(n0, warpSums, &tot0);
        if (has0) pre[tid] = e0;
        if (uCount > WG) {
          const uint32_t e1 = t1k_block_scan_exclusive(n1, warpSums, &tot1);
          if (has1) pre[tid + WG] = tot0 + e1;
        }
        const uint32_t T = tot0 + tot1;
        if (tid == 0) { pre[uCount] = T; sStat[2] += T; }
        __syncthreads();

#ifdef T1K_SEED_PROFILE
    { const uint64_t tn_ = __builtin_amdgcn_s_memtime(); tp_[5] += tn_ - tl_; tl_ = tn_; }
#endif
        // walk the chunk's postings.  The flat posting index [0, T) is cut into one contiguous range per wavefront; a lane finds the
        // list of its first posting by bisection over the prefix ONCE, and from there its list index only moves forward (its
        // positions grow by 64 a step), so the later postings cost a look at one or two prefix entries instead of a bisection each.
        // Four postings in flight per lane; 64 consecutive postings per wavefront load.
        {
          const uint32_t wv = (uint32_t)tid >> 6, ln = (uint32_t)tid & 63u;
          const uint32_t Rw = (((T + 3) >> 2) + 63u) & ~63u;
          const uint32_t jBeg = wv * Rw, jEnd = min(T, jBeg + Rw);
          uint32_t lo = 0;
          if (jBeg + ln < jEnd) {
            const uint32_t j = jBeg + ln;
            uint32_t hi = uCount;
            while (hi - lo > 1) { uint32_t m = (lo + hi) >> 1; if (pre[m] <= j) lo = m; else hi = m; }
          }
#ifndef T1K_SEED_INFLIGHT
#define T1K_SEED_INFLIGHT 4
#endif
          for (uint32_t j0 = jBeg + ln; j0 < jEnd; j0 += T1K_SEED_INFLIGHT * 64) {
            T1kPosting pst[T1K_SEED_INFLIGHT];
            int rr[T1K_SEED_INFLIGHT];
#pragma unroll
            for (int x = 0; x < T1K_SEED_INFLIGHT; ++x) {
              const uint32_t j = j0 + x * 64;
              if (j < jEnd) {
                while (pre[lo + 1] <= j) ++lo;  // pre[uCount] = T > j ends it
                pst[x] = P.ref.kPost[lstStart[uBegin + lo] + sLo[lo] + (j - pre[lo])];
                rr[x] = T1K_SEED_PACK_Q ? (int)(lstLen[uBegin + lo] >> 23) : (int)qOf[uBegin + lo];
              }
            }
#pragma unroll
            for (int x = 0; x < T1K_SEED_INFLIGHT; ++x) {
              const uint32_t j = j0 + x * 64;
              if (j < jEnd) {
                const int r = rr[x];
                const int d = r - (int)pst[x].offset;
                uint32_t *a = acc + (pst[x].allele - c0) * AW;
                const uint32_t old = atomicCAS(&a[0], (uint32_t)DIAG_EMPTY, (uint32_t)d);
                if (old == (uint32_t)DIAG_EMPTY || old == (uint32_t)d) atomicOr(&a[2 + (r >> 5)], 1u << (r & 31));
                else {
                  int dd = d - (int)old; if (dd < 0) dd = -dd;
                  atomicAdd(&a[1], dd <= P.radius ? 0x10001u : 1u);
                }
              }
            }
          }
        }
        __syncthreads();

#ifdef T1K_SEED_PROFILE
    { const uint64_t tn_ = __builtin_amdgcn_s_memtime(); tp_[6] += tn_ - tl_; tl_ = tn_; }
#endif
        // emit the groups that can still produce a candidate: >= 3 hits in total, and either >= 3 on the reference diagonal
        // or some hit close enough to chain with it, or > 2 strays (which could form their own run).  Lane t looks at the
        // accumulators t, t + WG, ... (conflict-free with the odd accumulator stride); the records leave in allele order.
        constexpr int EPT = CHUNK_A / WG;
        uint32_t flags = 0, kinds = 0;  // kinds (FUSE): 4 bits per accumulator row, groupFastPath's verdict (1 finished with a candidate, 5 gap walk) or 4 = several diagonals
        uint64_t packed = 0;  // EPT counters of 16 bits
        if (FUSE) {
#pragma unroll 1
          for (int i = 0; i < EPT; ++i) {
            uint32_t *a = acc + (i * WG + tid) * AW;
            if (a[0] == (uint32_t)DIAG_EMPTY) continue;
            int onDiag = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) onDiag += __popc(a[2 + w]);
            const uint32_t strays = a[1] & 0xFFFFu, nearCnt = a[1] >> 16;
            const bool general = nearCnt > 0 || strays > 2;
            flags |= 2u << (2 * i);  // occupied
            if (!(onDiag + (int)strays >= 3 && (general || onDiag >= 3))) continue;
            ++groupsLocal;
            uint32_t kind = 4;
            if (!general) {
              // the closed-form pass (what k_chain_fast<NW, 0> does with the record): the candidate of a group that ends here waits in the
              // accumulator's mask words (dead now) for the record write below; a group without a candidate leaves no record
              const uint32_t allele = c0 + i * WG + tid;
              kind = closedFormGroup<NW>(a, &sRd[pass][0][0], &sRd[pass][1][0], P.ref.bases, P.ref.nmask, (int64_t)P.ref.alleleOff[allele], P.ref.anyN != 0, P.ref.alleleHasN[allele] != 0, k,
                                         P.hitLenRequired, P.sim, pass, P.earlyPrune);
              if (kind == 1u) { ++fastLocal; continue; }  // no candidate
              if (kind == 6u) { ++fastLocal; kind = 1u; }
            }
            flags |= 1u << (2 * i); packed += 1ull << (16 * i); kinds |= kind << (4 * i); ++recsLocal;
          }
        } else {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
          const uint32_t *a = acc + (i * WG + tid) * AW;
          if (a[0] == (uint32_t)DIAG_EMPTY) continue;
          int onDiag = 0;
#pragma unroll
          for (int w = 0; w < NW; ++w) onDiag += __popc(a[2 + w]);
          const uint32_t strays = a[1] & 0xFFFFu, nearCnt = a[1] >> 16;
          const bool general = nearCnt > 0 || strays > 2;
          flags |= 2u << (2 * i);  // occupied
          if (onDiag + (int)strays >= 3 && (general || onDiag >= 3)) { flags |= 1u << (2 * i); packed += 1ull << (16 * i); }
        }
        }
        uint32_t totLo, totHi = 0, exHi = 0;
        const uint32_t exLo = t1k_block_scan_exclusive((uint32_t)packed, warpSums, &totLo);
        exHi = t1k_block_scan_exclusive((uint32_t)(packed >> 32), warpSums, &totHi);  // (skipping it when CHUNK_A <= 512 -- the high word is empty then -- measured 4 % SLOWER)
        const uint64_t ex = (uint64_t)exLo | ((uint64_t)exHi << 32), tt = (uint64_t)totLo | ((uint64_t)totHi << 32);
        const uint32_t gTot = (uint32_t)((tt & 0xFFFF) + ((tt >> 16) & 0xFFFF) + ((tt >> 32) & 0xFFFF) + (tt >> 48));
        if (tid == 0) {
          const uint32_t gb = gTot ? t1k_arena_alloc(P.counters, T1K_AR_GROUPS, gTot, P.groupSegCap) : 0u;
          const bool ok = gb != T1K_ARENA_FULL && chunk < P.maxChunks;
          if (!ok) atomicOr(&P.counters[2], (unsigned long long)ERR_GROUPCAP);
          sGroupBase = ok ? gb : 0xFFFFFFFFu;
          if (ok && gTot) { P.chunkStart[(uint64_t)re * P.maxChunks + chunk] = gb; P.chunkCount[(uint64_t)re * P.maxChunks + chunk] = gTot; }
        }
        __syncthreads();
        const uint32_t groupBase = sGroupBase;

#ifdef T1K_SEED_PROFILE
    { const uint64_t tn_ = __builtin_amdgcn_s_memtime(); tp_[7] += tn_ - tl_; tl_ = tn_; }
#endif
        if (gTot) ++chunk;
        uint32_t before = 0;  // records of the lower accumulator rows
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
          if ((flags >> (2 * i)) & 2u) {
            uint32_t *a = acc + (i * WG + tid) * AW;
            if (((flags >> (2 * i)) & 1u) && groupBase != 0xFFFFFFFFu) {
              const uint32_t slot = before + (uint32_t)((ex >> (16 * i)) & 0xFFFF);
              uint4 *rec = (uint4 *)(P.recs + (uint64_t)(groupBase + slot) * stride);
              constexpr int RW = NW == 5 ? 8 : 16;  // record words: re|strand, allele, diagonal + stray counts, M[NW]
              uint32_t v[RW];
              const uint32_t kind = FUSE ? (kinds >> (4 * i)) & 15u : 0u;
              v[0] = re | (pass == 0 ? 0x80000000u : 0);  // bit31: '+' strand
              v[1] = c0 + i * WG + tid;
              if (FUSE && kind == 1u) {  // finished: state, candidate (the record k_chain_fast<NW, 0> leaves behind)
                v[2] = REC_DONE | 1u; v[3] = a[2]; v[4] = a[3]; v[5] = a[4];
#pragma unroll
                for (int w = 6; w < RW; ++w) v[w] = 0;
              } else {
                v[2] = packDiagMeta((int)a[0], a[1]);
#pragma unroll
                for (int w = 0; w < NW; ++w) v[3 + w] = a[2 + w];
#pragma unroll
                for (int w = 3 + NW; w < RW; ++w) v[w] = 0;
              }
#pragma unroll
              for (int w = 0; w < RW / 4; ++w) rec[w] = make_uint4(v[4 * w], v[4 * w + 1], v[4 * w + 2], v[4 * w + 3]);
              if (FUSE) {  // the work lists k_chain_fast<NW, 0> used to fill
                if (kind == 5u) { const uint32_t q = t1k_arena_append(P.counters, T1K_AR_SLOW, P.listSegCap); if (q != T1K_ARENA_FULL) P.slowStr[q] = groupBase + slot; }
                else if (kind == 4u) { const uint32_t q = t1k_arena_append(P.counters, T1K_AR_GENERAL, P.rareSegCap); if (q != T1K_ARENA_FULL) P.generalStr[q] = groupBase + slot; }
              }
            }
            a[0] = (uint32_t)DIAG_EMPTY; a[1] = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) a[2 + w] = 0;
          }
          before += (uint32_t)((tt >> (16 * i)) & 0xFFFF);
        }
        __syncthreads();
       }
      }
    }
  }
#ifdef T1K_SEED_PROFILE
  if (tid == 0) for (int i = 0; i < 8; ++i) atomicAdd(&P.counters[48 + i], (unsigned long long)tp_[i]);
#endif
  if (tid == 0) {  // statistics: one striped atomic per workgroup and counter
    unsigned long long *st = P.counters + 64 + (blockIdx.x & (T1K_STAT_STRIPES - 1)) * 8;
    atomicAdd(&st[T1K_STAT_LOOKUPS], sStat[0]); atomicAdd(&st[T1K_STAT_POSTINGS], sStat[1]); atomicAdd(&st[T1K_STAT_HITS], sStat[2]);
  }
  if (FUSE) {
    t1k_stat_add(P.counters, T1K_STAT_FAST, fastLocal);
    // groups seeded = records written + groups that ended without a candidate: the latter are counted here (control counters 56..63, striped)
    unsigned int noRec = groupsLocal - recsLocal;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) noRec += __shfl_down(noRec, o, 64);
    if ((tid & 63) == 0 && noRec) atomicAdd(&P.counters[56 + ((blockIdx.x * 4 + (tid >> 6)) & 7)], (unsigned long long)noRec);
  }
}
template <int NW>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(T1K_SEED_WAVES, T1K_SEED_WAVES))) void k_seed_groups(ChainArgs P) { seedGroupsBody<NW, false>(P); }
template <int NW>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(T1K_FUSE_WAVES, T1K_FUSE_WAVES))) void k_seed_chain(ChainArgs P) { seedGroupsBody<NW, true>(P); }

// ------------------------------------------------------------------------------------------------------------------
// K1L: seeding of the read-ends beyond the hit masks' span (T1K_MAX_READ_LEN < len <= T1K_LONG_READ_LEN), one workgroup per such
// read-end; the others are k_seed_groups' and are skipped here.  No masks, no diagonals: the look-up rule (GetHitsFromRead,
// SeqSet.hpp:1071-1229) is replayed sequentially by the first wavefront, the used lists are kept for gatherHits as usual, and a
// (strand, allele) pair that collects >= 3 postings (minHitRequired, 1253 / 1314) becomes a group record flagged "several diagonals":
// k_chain_fast<*, 0> hands such records to k_gather_general -> k_chain_general / k_chain_wave / k_chain_big, which rebuild the hit
// list and run the reference's diagonal-run / LIS logic on it whatever the read's length.  Counts live in LDS for LONG_CH alleles at
// a time; a pass emits its records in allele order as one entry of the read-end's chunk table ('-' strand first, as SortHits
// 1577-1583 orders the groups).  Slow by design: such reads are the odd ones among millions.
// ------------------------------------------------------------------------------------------------------------------
#define LONG_CH 16384
__global__ __launch_bounds__(WG) void k_seed_long(ChainArgs P) {
  extern __shared__ uint32_t lds[];
  const int k = P.k;
  const int maxK = (int)P.maxK;
  uint32_t *ukCode = lds;                           // [maxK]  code | valid << 31
  uint32_t *ukStart = ukCode + maxK;                // [maxK]
  uint32_t *ukLen = ukStart + maxK;                 // [maxK]
  uint32_t *ukDir = ukLen + maxK;                   // [maxK]
  uint16_t *usedQ = (uint16_t *)(ukDir + maxK);     // [maxK]  used k-mers, + strand first
  uint32_t *cnt = (uint32_t *)(usedQ + ((maxK + 1) & ~1));  // [LONG_CH] postings per allele of the current pass
  __shared__ uint32_t warpSums[4];
  __shared__ uint32_t sUsed[2], sGroupBase;
  const int tid = threadIdx.x;
  const uint32_t kmask = (1u << (2 * k)) - 1;
  const uint32_t stride = P.recStride;
  const uint32_t A = P.ref.nAlleles;
  unsigned int hitsLocal = 0;
  for (uint32_t re = blockIdx.x; re < P.reads.nReadEnds; re += gridDim.x) {
    const int len = P.reads.len[re];
    if (len <= T1K_MAX_READ_LEN || (P.reads.skip && P.reads.skip[re])) continue;  // (uniform over the workgroup)
    const int S = P.reads.S;
    const uint64_t *rbase = P.reads.bases + (uint64_t)re * 2 * S;
    const uint64_t *rnm = P.reads.nmask + (uint64_t)re * 2 * S;
    const int nk = len - k + 1;
    __syncthreads();  // the previous read-end's tables are dead
    for (int q = tid; q < 2 * nk; q += WG) {
      const int pass = q / nk, p = q - pass * nk;
      const uint64_t *b = rbase + pass * S, *nm = rnm + pass * S;
      const uint32_t code = (uint32_t)t1k_get32(b, p) & kmask;
      const bool valid = ((uint32_t)t1k_get32(nm, p) & kmask) == 0;
      uint32_t st = 0, ln = 0, dr = T1K_NO_DIR;
      if (valid) { st = P.ref.kStart[code]; ln = P.ref.kStart[code + 1] - st; dr = P.ref.kDirIdx[code]; }
      ukCode[q] = code | (valid ? 0x80000000u : 0);
      ukStart[q] = st; ukLen[q] = ln; ukDir[q] = dr;
    }
    __syncthreads();
    // the look-up rule, sequentially (SeqSet.hpp:1098-1153, 1165-1226): lane j of the first wavefront holds one k-mer's code
    // and list length, the loop reads them with v_readlane (the same replay k_seed_groups runs for reads with short repeats)
    if (tid < 64) {
      uint32_t prev = 0, nUsed = 0, lookups = 0, postings = 0;
      for (int pass = 0; pass < 2; ++pass) {
        int skipCnt = 0;
        const uint32_t begin = nUsed;
        for (int seg = 0; seg < nk; seg += 64) {
          const int pl = seg + tid;
          const uint32_t vc = pl < nk ? (ukCode[pass * nk + pl] & 0x7FFFFFFFu) : 0u;
          const uint32_t vl = pl < nk ? ukLen[pass * nk + pl] : 0u;
          const int cntj = min(64, nk - seg);
          for (int j = 0; j < cntj; ++j) {
            const uint32_t code = (uint32_t)__builtin_amdgcn_readlane((int)vc, j);
            const uint32_t size = (uint32_t)__builtin_amdgcn_readlane((int)vl, j);
            const int p = seg + j;
            if (p == 0 || code != prev) {
              ++lookups;
              if (size >= 100 && p != 0 && p != nk - 1 && skipCnt < k / 2) { ++skipCnt; continue; }
              skipCnt = 0;
              if (size) {
                if (tid == 0) usedQ[nUsed] = (uint16_t)(pass * nk + p);
                ++nUsed;
                postings += size;
              }
            }
            prev = code;
          }
        }
        if (tid == 0) sUsed[pass] = nUsed - begin;
      }
      if (tid == 0) {
        P.usedCount[2 * re] = sUsed[0]; P.usedCount[2 * re + 1] = sUsed[1];
        unsigned long long *st = P.counters + 64 + (blockIdx.x & (T1K_STAT_STRIPES - 1)) * 8;
        atomicAdd(&st[T1K_STAT_LOOKUPS], (unsigned long long)lookups); atomicAdd(&st[T1K_STAT_POSTINGS], (unsigned long long)postings);
      }
    }
    __syncthreads();
    const uint32_t nUsedPlus = sUsed[0], nUsedMinus = sUsed[1];
    {
      uint32_t *uo = P.usedOut + (uint64_t)re * maxK * 4;
      for (uint32_t u = tid; u < nUsedPlus + nUsedMinus; u += WG) {
        const int q = usedQ[u];
        const int pass = u < nUsedPlus ? 0 : 1;
        uo[4 * u] = (uint32_t)(q - pass * nk); uo[4 * u + 1] = ukStart[q]; uo[4 * u + 2] = ukLen[q]; uo[4 * u + 3] = ukDir[q];
      }
    }
    int chunk = 0;
    for (int sp = 0; sp < 2; ++sp) {  // '-' strand first
      const int pass = sp == 0 ? 1 : 0;
      const uint32_t uBegin = pass == 0 ? 0 : nUsedPlus;
      const uint32_t uCount = pass == 0 ? nUsedPlus : nUsedMinus;
      if (uCount == 0) continue;
      for (uint32_t c0 = 0; c0 < A; c0 += LONG_CH) {
        const uint32_t c1 = min(c0 + (uint32_t)LONG_CH, A);
        for (uint32_t i = tid; i < LONG_CH; i += WG) cnt[i] = 0;
        __syncthreads();
        for (uint32_t u = tid; u < uCount; u += WG) {
          const int q = usedQ[uBegin + u];
          const uint32_t st = ukStart[q], ln = ukLen[q];
          uint32_t lo = 0, hi = ln;  // first posting with allele >= c0
          while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (P.ref.kPostAllele[st + m] < c0) lo = m + 1; else hi = m; }
          for (uint32_t j = lo; j < ln; ++j) {
            const uint32_t al = P.ref.kPostAllele[st + j];
            if (al >= c1) break;
            atomicAdd(&cnt[al - c0], 1u);
            ++hitsLocal;
          }
        }
        __syncthreads();
        constexpr uint32_t PER = LONG_CH / WG;  // thread t looks at the alleles [c0 + t * PER, + PER): the records leave in allele order
        uint32_t mine = 0;
        for (uint32_t i = 0; i < PER; ++i) mine += cnt[tid * PER + i] >= 3u ? 1u : 0u;
        uint32_t gTot;
        const uint32_t ex = t1k_block_scan_exclusive(mine, warpSums, &gTot);
        if (tid == 0) {
          const uint32_t gb = gTot ? t1k_arena_alloc(P.counters, T1K_AR_GROUPS, gTot, P.groupSegCap) : 0u;
          const bool ok = gb != T1K_ARENA_FULL && chunk < P.maxChunks;
          if (!ok) atomicOr(&P.counters[2], (unsigned long long)ERR_GROUPCAP);
          sGroupBase = ok ? gb : 0xFFFFFFFFu;
          if (ok && gTot) { P.chunkStart[(uint64_t)re * P.maxChunks + chunk] = gb; P.chunkCount[(uint64_t)re * P.maxChunks + chunk] = gTot; }
        }
        __syncthreads();
        const uint32_t groupBase = sGroupBase;
        if (gTot) ++chunk;
        if (mine && groupBase != 0xFFFFFFFFu) {
          uint32_t slot = ex;
          for (uint32_t i = 0; i < PER; ++i) {
            if (cnt[tid * PER + i] < 3u) continue;
            uint4 *rec = (uint4 *)(P.recs + (uint64_t)(groupBase + slot) * stride);
            // words 0..2: read-end | '+' strand, allele, "several diagonals" (recIsGeneral: near count 1, diagonal 0); the rest is the chain's
            rec[0] = make_uint4(re | (pass == 0 ? 0x80000000u : 0u), c0 + tid * PER + i, (1u << 21) | (31u << 25), 0u);  // near = 31: "count unknown", the hits come from the used lists
            for (uint32_t w = 1; w < stride / 4; ++w) rec[w] = make_uint4(0u, 0u, 0u, 0u);
            if (P.fuse) { const uint32_t gq = t1k_arena_append(P.counters, T1K_AR_GENERAL, P.rareSegCap); if (gq != T1K_ARENA_FULL) P.generalStr[gq] = groupBase + slot; }  // (no k_chain_fast<*, 0> lists it)
            ++slot;
          }
        }
        __syncthreads();
      }
    }
  }
  t1k_stat_add(P.counters, T1K_STAT_HITS, hitsLocal);
}

// ------------------------------------------------------------------------------------------------------------------
// K2 / K4: single-diagonal chain, one lane per group record
// record words: 0 re|strand, 1 allele, 2 diagonal + stray counts, 3.. M   -> after chaining: 2 = state, 3..5 packed candidate (3 = side-arena base for
// multi-diagonal groups), 6..7 memo slots still to be added
// ------------------------------------------------------------------------------------------------------------------
// MODE 0: all records, closed form only (the rest -> slow list; multi-diagonal groups -> general list)
// MODE 1: slow list: gap walk, alignments registered in the memo (-> finish list / retry list)
// MODE 2: retry list after k_dp_dense: alignments from the memo or inline
// nDev != nullptr: the list's length is read there (the total word k_arena_compact left), nItems is ignored.  The workgroups stride over the
// items, so a grid sized from an estimate (t1k_run_chain) covers whatever the device counted.
// LOOP = false (the host-driven launches: one item per thread, the grid covers the list) keeps the straight-line kernel: the loop costs the
// closed-form pass 16 VGPRs and a wavefront per SIMD (78 -> 94).
#if T1K_CF0_WAVES > 0
#define T1K_CF_ATTR __attribute__((amdgpu_waves_per_eu(MODE == 0 && NW == 5 ? T1K_CF0_WAVES : 1)))
#else
#define T1K_CF_ATTR
#endif
template <int NW, int MODE, bool LOOP>
__global__ __launch_bounds__(WG) T1K_CF_ATTR void k_chain_fast(ChainArgs P, const uint32_t *list, uint32_t nItems, const unsigned long long *nDev) {
  constexpr bool DEFER = MODE != 2;
  unsigned int dpLocal = 0, fastLocal = 0;
  if (nDev && P.counters[2]) return;  // an arena overflowed earlier in this submission: the range runs again, nothing of this pass is kept
  // list mode: 1-D grid over a dense list of record indices; all records: blockIdx.y = arena segment, blockIdx.x strides over the segment
  const uint32_t limit = list ? (nDev ? (uint32_t)*nDev : nItems)
                              : (uint32_t)min(*t1k_arena_cursor(P.counters, T1K_AR_GROUPS, blockIdx.y), (unsigned long long)P.groupSegCap);
  const uint32_t step = gridDim.x * WG;
#pragma unroll 1
  for (uint32_t base = blockIdx.x * WG; base < limit; base += step) {  // (lists hold fewer than 2^32 - step entries)
  int kind = 0;
  uint32_t gi = 0;
  const uint32_t gid = base + threadIdx.x;
  const bool valid = gid < limit;
  if (valid) gi = list ? list[gid] : blockIdx.y * P.groupSegCap + gid;
  if (valid) {
    uint32_t *rec = P.recs + (uint64_t)gi * P.recStride;
    constexpr int RW = NW == 5 ? 8 : 16;
    uint32_t rv[RW];  // the record, fetched as 16-byte pieces
#pragma unroll
    for (int w = 0; w < RW / 4; ++w) {
      const uint4 q4 = ((const uint4 *)rec)[w];
      rv[4 * w] = q4.x; rv[4 * w + 1] = q4.y; rv[4 * w + 2] = q4.z; rv[4 * w + 3] = q4.w;
    }
    const uint32_t re = rv[0] & 0x7FFFFFFFu, allele = rv[1];
    const int pass = (rv[0] >> 31) ? 0 : 1;
    const bool general = recIsGeneral(rv[2]);
    if (general) kind = 4;
    else {
      uint32_t Mw[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) Mw[w] = rv[3 + w];
      ReadCtx c = makeCtx(P, re, pass, allele);
      // (the transposed copy pays where the lanes hold consecutive alleles: the pass over ALL records.  The gap walk's list is a scattered 40 % of them --
      // measured with the copy: its fabric traffic went UP by a third, six lines a lane instead of one or two; profiles/r06_callC_*.log)
      if (MODE != 0) c.gT = nullptr;
      uint32_t cbuf[3];
      CandOut out{cbuf, 0};
      const GapSink sink{P.memo + (uint64_t)re * GAP_CACHE, P.jobStr, P.counters, re * GAP_CACHE, P.jobSegCap, T1K_AR_JOBS};
      uint32_t refs[2] = {0, 0};
      int nRefs = 0;
      kind = groupFastPath<NW, DEFER, MODE == 0>(Mw, recDiag(rv[2]), c, P.ref.alleleHasN[allele] != 0, P.k, P.hitLenRequired, P.sim, out, &dpLocal, pass, sink, refs, &nRefs, P.earlyPrune);
      if (kind == 3) {  // matchCnt lacks the registered alignments: k_collect adds them from the memo
        // words 2..7: state (= number of memo slots to add), candidate, slots
        ((uint2 *)rec)[1] = make_uint2((uint32_t)nRefs, cbuf[0]);
        ((uint2 *)rec)[2] = make_uint2(cbuf[1], cbuf[2]);
        ((uint2 *)rec)[3] = make_uint2(refs[0], refs[1]);
      } else if (kind == 1) {
        ++fastLocal;
        if (out.n) { ((uint2 *)rec)[1] = make_uint2(REC_DONE | 1u, cbuf[0]); ((uint2 *)rec)[2] = make_uint2(cbuf[1], cbuf[2]); }
        else if (MODE != 0) rec[2] = REC_DONE;
        // (MODE 0, no candidate -- six in ten groups: the record is left as seeding wrote it.  Word 2 still holds the diagonal word: neither REC_DONE nor a
        // pending count (it is at least 2^20), which k_collect reads as "no candidate".  Round 6: a 4-byte store into a record nobody reads again was a
        // 32-byte write on the fabric, 0.19 TB per 10 M-pair step)
      }
    }
  }
  // list appends: one atomic per wavefront, list and arena segment (a full segment shows in its cursor; the host fails the batch)
  if (kind == 2) { const uint32_t q = t1k_arena_append(P.counters, T1K_AR_RETRY, P.listSegCap); if (q != T1K_ARENA_FULL) P.retryStr[q] = gi; }
  if (kind == 3) { const uint32_t q = t1k_arena_append(P.counters, T1K_AR_FINISH, P.listSegCap); if (q != T1K_ARENA_FULL) P.finishStr[q] = gi; }
  if (kind == 4) { const uint32_t q = t1k_arena_append(P.counters, T1K_AR_GENERAL, P.rareSegCap); if (q != T1K_ARENA_FULL) P.generalStr[q] = gi; }
  if (kind == 5) { const uint32_t q = t1k_arena_append(P.counters, T1K_AR_SLOW, P.listSegCap); if (q != T1K_ARENA_FULL) P.slowStr[q] = gi; }
  if (!LOOP) break;
  }
  t1k_stat_add(P.counters, T1K_STAT_DP, dpLocal);
  t1k_stat_add(P.counters, T1K_STAT_FAST, fastLocal);
}

// K3: one lane per registered alignment
__global__ __launch_bounds__(WG) void k_dp_dense(ChainArgs P, const uint32_t *jobs, uint32_t nJobs, const unsigned long long *nDev) {
  if (nDev) { if (P.counters[2]) return; nJobs = (uint32_t)*nDev; }
  unsigned int dpLocal = 0;
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nJobs; q += gridDim.x * blockDim.x) {
    const uint32_t tag = jobs[q];
    const uint32_t re = tag / GAP_CACHE;
    unsigned long long *slot = P.memo + tag;
    const unsigned long long e = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int sb = (int)(e & 1), d = (int)((e >> 1) & 0xF) - 4, lp = (int)((e >> 5) & 0x1FF), readPos = (int)((e >> 14) & 0x7FF);
    const int S = P.reads.S;
    ReadCtx c{P.reads.bases + ((uint64_t)re * 2 + sb) * S, P.reads.nmask + ((uint64_t)re * 2 + sb) * S, 0, P.ref.bases, P.ref.nmask, 0, 0, true};
    const int m = gapAlign(c, readPos, GAP_GPOS(e), lp, lp + d);
    ++dpLocal;
    __hip_atomic_store(slot, (e & ~(GAP_PENDING << 25)) | ((unsigned long long)m << 25), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  t1k_stat_add(P.counters, T1K_STAT_DP, dpLocal);
}

// K3b: groups whose candidate waits for registered alignments: add the memo's match counts (SeqSet.hpp:1736-1741)
__global__ __launch_bounds__(WG) void k_chain_finish(ChainArgs P, uint32_t nItems) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned int fastLocal = 0;
  if (q < nItems) {
    uint32_t *rec = P.recs + (uint64_t)P.finishList[q] * P.recStride;
    const uint32_t re = rec[0] & 0x7FFFFFFFu;
    const int nRefs = (int)rec[2];
    const unsigned long long *memo = P.memo + (uint64_t)re * GAP_CACHE;
    uint32_t sum = 0;
    for (int i = 0; i < nRefs; ++i) {
      const uint32_t slot = (rec[6 + (i >> 1)] >> (16 * (i & 1))) & 0xFFFFu;
      const unsigned long long e = __hip_atomic_load(&memo[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t v = GAP_VAL(e);
      if (v == GAP_PENDING) atomicOr(&P.counters[2], (unsigned long long)ERR_MEMO);  // cannot happen: every registered job is run by k_dp_dense
      sum += v;
    }
    rec[5] += (2u * sum) << 20;
    rec[2] = REC_DONE | 1u;
    ++fastLocal;
  }
  t1k_stat_add(P.counters, T1K_STAT_FAST, fastLocal);
}

// re-derive the hit list of one (read-end, strand, allele) group from the used posting lists kept by k_seed_groups:
// the allele's postings are a contiguous run of each list (lists are sorted by allele, then offset)
__device__ inline int gatherHits(const ChainArgs &P, uint32_t re, int pass, uint32_t allele, uint32_t *h, int cap) {
  const int maxK = (int)P.maxK;
  const uint32_t nPlus = P.usedCount[2 * re], nMinus = P.usedCount[2 * re + 1];
  const uint32_t b = pass == 0 ? 0 : nPlus, e = pass == 0 ? nPlus : nPlus + nMinus;
  const uint32_t *uo = P.usedOut + (uint64_t)re * maxK * 4;
  int n = 0;
  for (uint32_t u = b; u < e; ++u) {
    const uint32_t rOff = uo[4 * u], st = uo[4 * u + 1], ln = uo[4 * u + 2];
    uint32_t l = 0, r = ln;
    while (l < r) { uint32_t m = (l + r) >> 1; if (P.ref.kPost[st + m].allele < allele) l = m + 1; else r = m; }
    for (; l < ln; ++l) {
      const T1kPosting pst = P.ref.kPost[st + l];
      if (pst.allele != allele) break;
      if (n < cap) h[n] = (pst.offset << 12) | rOff;
      ++n;
    }
  }
  return n;
}

#define GENERAL_SMALL 32   // hits per group handled one lane per group (k_chain_general)
#define WAVE_CAP 4096      // ... one wavefront per group with the work arrays in LDS (k_chain_wave); beyond: k_chain_big
// K5a: hit lists of the multi-diagonal groups.  One wavefront per group: the lanes share the read-end's used posting lists,
// each finds the allele's run in its lists (sorted by allele, then offset) and the hits are written to the hit arena;
// record word 4 = arena offset, word 5 = hit count (0xFFFFFFFF: handed to k_chain_big)
// ROUNDS: used lists of one strand over the 64 lanes (5: reads <= 320 bp; 16: up to T1K_LONG_READ_LEN)
// ------------------------------------------------------------------------------------------------------------------
// K5a': hit lists of the multi-diagonal groups WITHOUT the posting lists.  Most such groups have all their hits within `radius` diagonals
// of the reference diagonal (a read with an indel: every one of its ~2000 groups; an allele with a small indel under the read): far == 0
// in the record.  A hit is a posting (allele, b) of a USED list of read offset a: the read's k-mer at a equals the allele's at b, that
// k-mer of the allele was inserted (the `posted` bitmap of the index: valid + the reference's insert rule), and a's list is used (the
// read-end's used-offset mask).  All three are bit masks over the read offsets, per diagonal d = a - b -- so one lane rebuilds the hit
// list of a group from the packed read, ~7 words of the allele's text and of `posted`, with shifts and ANDs, for the 2 * radius + 1
// diagonals, instead of a wavefront bisecting ~130 posting lists (k_gather_general: 9 dependent loads a lane, 2.8 KB fetched a group).
// Checked, not assumed: the mask of the reference diagonal must equal the record's (which came from the postings themselves), and the
// hits on the other diagonals must number exactly the record's near count; anything else (far strays, a saturated count, windows at
// the ends of the packed text) leaves the group to k_gather_general.  Record word 5 = REC_NEAR_DONE marks a group done here.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t evenBits32(uint64_t x) {  // bit 2j of x -> bit j
  x &= 0x5555555555555555ull;
  x = (x | (x >> 1)) & 0x3333333333333333ull; x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full; x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
  x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull; x = (x | (x >> 16)) & 0xFFFFFFFFull;
  return (uint32_t)x;
}
template <int NW>
__global__ __launch_bounds__(WG) void k_near_hits(ChainArgs P, uint32_t nItems, const unsigned long long *nDev) {
  constexpr int NQ = NW + 1;  // 64-bit words (32 positions each) of the per-position masks: one spare word for the k-mer's reach
  if (nDev) { if (P.counters[2]) return; nItems = (uint32_t)*nDev; }
  for (uint32_t qb = blockIdx.x * WG; qb < nItems; qb += gridDim.x * WG) {
  const uint32_t q = qb + threadIdx.x;
  bool toWave = false;
  uint32_t gi = 0;
  if (q < nItems) {
    gi = P.generalList[q];
    uint32_t *rec = P.recs + (uint64_t)gi * P.recStride;
    constexpr int RW = NW == 5 ? 8 : 16;
    uint32_t rv[RW];
#pragma unroll
    for (int w = 0; w < RW / 4; ++w) { const uint4 q4 = ((const uint4 *)rec)[w]; rv[4 * w] = q4.x; rv[4 * w + 1] = q4.y; rv[4 * w + 2] = q4.z; rv[4 * w + 3] = q4.w; }
    const uint32_t re = rv[0] & 0x7FFFFFFFu, allele = rv[1];
    const int pass = (rv[0] >> 31) ? 0 : 1;
    const int d0 = recDiag(rv[2]), k = P.k, R = P.radius;
    const uint32_t nearCnt = recNear(rv[2]);
    uint32_t onDiag = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) onDiag += __popc(rv[3 + w]);
    const int64_t goff = (int64_t)P.ref.alleleOff[allele];
    const int alleleLen = (int)P.ref.alleleLen[allele];
    const int64_t W0 = goff - d0 - R;  // global position under read offset 0 on the diagonal d0 + R
    const int64_t lastWord = (W0 >> 5) + NQ + 2;
    bool ok = recFar(rv[2]) == 0 && nearCnt > 0 && nearCnt < 31 && R <= 15 && W0 >= 0 && (uint64_t)lastWord < (P.ref.totalBases >> 5) + 7;
    uint32_t base = T1K_ARENA_FULL;
    const uint32_t n = onDiag + nearCnt;
    if (ok) { base = t1k_arena_alloc(P.counters, T1K_AR_GENHITS, n, P.genHitSegCap); ok = base != T1K_ARENA_FULL; }
    if (ok) {
      const int S = P.reads.S;
      const uint64_t *rb = P.reads.bases + ((uint64_t)re * 2 + pass) * S;
      const uint32_t *um = P.usedMask + ((uint64_t)re * 2 + pass) * T1K_USED_MASK_WORDS;
      uint64_t Rr[NQ], G[NQ + 2], PG[NQ + 2];
#pragma unroll
      for (int j = 0; j < NQ; ++j) Rr[j] = j < S ? rb[j] : 0ull;
      const uint64_t *gb = P.ref.bases + (W0 >> 5), *pb = P.ref.posted + (W0 >> 5);
#pragma unroll
      for (int j = 0; j < NQ + 2; ++j) { G[j] = gb[j]; PG[j] = pb[j]; }
      const int sh0 = (int)(W0 & 31);
      uint32_t w = base, cntNear = 0;
      bool same0 = false;
      int nearDiags = 0, d1 = 0;    // near diagonals that hold hits; the first one and its mask
      uint32_t M1[NW];
#pragma unroll
      for (int j = 0; j < NW; ++j) M1[j] = 0;
      for (int dl = -R; dl <= R; ++dl) {
        const int d = d0 + dl;
        const int o = sh0 + (R - dl);          // base offset of read position 0 inside G (0 .. 31 + 2R < 64)
        const bool up = o >= 32;
        const int bsh = (o & 31) * 2;
        uint64_t acc[NQ], pst[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
          const uint64_t lo = up ? G[j + 1] : G[j], hi = up ? G[j + 2] : G[j + 1];
          const uint64_t g = bsh ? ((lo >> bsh) | (hi << (64 - bsh))) : lo;
          const uint64_t x = g ^ Rr[j];
          acc[j] = ~(x | (x >> 1)) & 0x5555555555555555ull;   // bit 2p: the bases at position 32 j + p are equal
          const uint64_t pl = up ? PG[j + 1] : PG[j], ph = up ? PG[j + 2] : PG[j + 1];
          pst[j] = bsh ? ((pl >> bsh) | (ph << (64 - bsh))) : pl;
        }
        // acc(a) = AND over t < k of eq(a + t), by doubling (a shift by m positions = 2 m bits towards lower positions)
        auto andShifted = [&](int m) {
          const int b = 2 * m;
#pragma unroll
          for (int j = 0; j < NQ; ++j) { const uint64_t nx = j + 1 < NQ ? acc[j + 1] : 0ull; acc[j] &= (acc[j] >> b) | (nx << (64 - b)); }
        };
        {
          int have = 1;
          while (2 * have <= k) { andShifted(have); have *= 2; }
          if (have < k) andShifted(k - have);
        }
        // hits of this diagonal: equal k-mers, posted in the index, the read offset's list used, the k-mer inside THIS allele
        const int aLo = d > 0 ? d : 0, aHi = d + alleleLen - k;  // b = a - d in [0, alleleLen - k]
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          uint32_t m = evenBits32(acc[j] & pst[j]) & um[j];
          const int p0 = 32 * j;
          if (aLo > p0) m &= aLo - p0 >= 32 ? 0u : ~0u << (aLo - p0);
          if (aHi < p0 + 31) m &= aHi < p0 ? 0u : ~0u >> (31 - (aHi - p0));
          if (dl == 0) { if (j == 0) same0 = true; same0 = same0 && m == rv[3 + j]; continue; }
          if (m) {
            if (nearDiags == 0 || d1 != d) { ++nearDiags; if (nearDiags == 1) d1 = d; }
            if (nearDiags == 1) M1[j] = m;
          }
          cntNear += __popc(m);
          while (m) {
            const int p = __ffs((int)m) - 1;
            m &= m - 1;
            const uint32_t a = (uint32_t)(p0 + p);
            if (w < base + n) P.genHits[w] = ((uint32_t)((int)a - d) << 12) | a;
            ++w;
          }
        }
      }
      ok = same0 && cntNear == nearCnt;
      // two diagonals whose hits follow one another on the read and on the allele: the list IS its chain (groupSimple)
      bool simple = false;
      if (ok && P.nearSimple && nearDiags == 1 && n >= 3 && (int)n * k >= P.hitLenRequired) {
        int lo0 = -1, hi0 = -1, lo1 = -1, hi1 = -1;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          if (rv[3 + j]) { if (lo0 < 0) lo0 = 32 * j + __ffs((int)rv[3 + j]) - 1; hi0 = 32 * j + 31 - __clz((int)rv[3 + j]); }
          if (M1[j]) { if (lo1 < 0) lo1 = 32 * j + __ffs((int)M1[j]) - 1; hi1 = 32 * j + 31 - __clz((int)M1[j]); }
        }
        const bool first0 = hi0 < lo1 && hi0 - d0 < lo1 - d1, first1 = hi1 < lo0 && hi1 - d1 < lo0 - d0;
        if (first0 || first1) {
          simple = true;
          w = base;
          for (int part = 0; part < 2; ++part) {
            const bool zero = (part == 0) == first0;  // this part is the reference diagonal's
            const int dd = zero ? d0 : d1;
#pragma unroll
            for (int j = 0; j < NW; ++j) {
              uint32_t m = zero ? rv[3 + j] : M1[j];
              while (m) {
                const int p = __ffs((int)m) - 1;
                m &= m - 1;
                const uint32_t a = (uint32_t)(32 * j + p);
                P.genHits[w++] = ((uint32_t)((int)a - dd) << 12) | a;
              }
            }
          }
        }
      }
      if (ok && simple) { rec[3] = base; rec[4] = n; rec[6] = 1u; toWave = n > 3 * GENERAL_SMALL; }
      else if (ok) {
        rec[6] = 0u;
        // ... and the reference diagonal's own hits, from the record's mask
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          uint32_t m = rv[3 + j];
          while (m) {
            const int p = __ffs((int)m) - 1;
            m &= m - 1;
            const uint32_t a = (uint32_t)(32 * j + p);
            P.genHits[w++] = ((uint32_t)((int)a - d0) << 12) | a;
          }
        }
        rec[3] = base; rec[4] = n;
        toWave = n > GENERAL_SMALL;
      }
    }
    rec[5] = ok ? REC_NEAR_DONE : 0u;
    if (!ok) rec[6] = 0u;
#ifdef T1K_NEAR_STATS
    {  // what the multi-diagonal groups are (profiles/r04_multidiag_classes.txt)
      int cls = 42;
      if (recFar(rv[2]) != 0) cls = 40; else if (nearCnt == 31) cls = 41; else if (ok) cls = rec[6] == 1u ? (n > 96 ? 46 : 45) : (n > 32 ? 47 : 43);
      atomicAdd(&P.counters[cls], 1ull);
    }
#endif
  }
  if (toWave) { const uint32_t wq = t1k_arena_append(P.counters, T1K_AR_WAVE, P.rareSegCap); if (wq != T1K_ARENA_FULL) P.waveStr[wq] = gi; }
  }
}

template <int ROUNDS>
__global__ __launch_bounds__(WG) void k_gather_general(ChainArgs P, uint32_t nItems, int skipDone, const unsigned long long *nDev) {
  if (nDev) { if (P.counters[2]) return; nItems = (uint32_t)*nDev; }
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * WG + threadIdx.x) >> 6, nWaves = gridDim.x * (WG / 64);
  const int maxK = (int)P.maxK;
  static_assert(ROUNDS * 64 >= GROUP_FAST_MAXLEN, "lists of one strand");
  for (uint32_t q = wave; q < nItems; q += nWaves) {
    const uint32_t gi = P.generalList[q];
    uint32_t *rec = P.recs + (uint64_t)gi * P.recStride;
    if (skipDone && rec[5] == REC_NEAR_DONE) continue;  // k_near_hits wrote this group's hits (uniform over the wavefront)
    const uint32_t re = rec[0] & 0x7FFFFFFFu, allele = rec[1];
    const int pass = (rec[0] >> 31) ? 0 : 1;
    const uint32_t nPlus = P.usedCount[2 * re], nMinus = P.usedCount[2 * re + 1];
    const uint32_t b = pass == 0 ? 0 : nPlus, e = pass == 0 ? nPlus : nPlus + nMinus;
    const uint32_t *uo = P.usedOut + (uint64_t)re * maxK * 4;
    uint32_t first[ROUNDS], cnt[ROUNDS], mine = 0;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const uint32_t u = b + lane + 64 * r;
      first[r] = 0; cnt[r] = 0;
      if (u < e) {
        const uint32_t st = uo[4 * u + 1], ln = uo[4 * u + 2], dr = uo[4 * u + 3];
        uint32_t l = 0, rr = ln;
        if (dr != T1K_NO_DIR) {  // a long list: its chunk directory narrows the search to the allele's chunk of T1K_SEED_CHUNK sequences
          const uint32_t *dir = P.ref.kDir + (uint64_t)dr * P.ref.kDirStride;
          const uint32_t ci = allele / T1K_SEED_CHUNK;
          l = dir[ci]; rr = dir[ci + 1];
        }
        while (l < rr) { uint32_t m = (l + rr) >> 1; if (P.ref.kPost[st + m].allele < allele) l = m + 1; else rr = m; }
        uint32_t c = 0;
        while (l + c < ln && P.ref.kPost[st + l + c].allele == allele) ++c;
        first[r] = st + l; cnt[r] = c; mine += c;
      }
    }
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
    const uint32_t n = __shfl(incl, 63, 64);
    uint32_t base = 0;
    if (lane == 0) {
      if (n > WAVE_CAP) {  // k_chain_big gathers for itself
        const uint32_t bq = t1k_arena_append(P.counters, T1K_AR_BIG, P.rareSegCap);
        if (bq != T1K_ARENA_FULL) P.bigStr[bq] = gi;
        rec[4] = 0xFFFFFFFFu;
      } else {
        base = n ? t1k_arena_alloc(P.counters, T1K_AR_GENHITS, n, P.genHitSegCap) : 0u;
        rec[3] = base; rec[4] = n;
        if (n > GENERAL_SMALL) {  // one wavefront per group (k_chain_wave)
          const uint32_t wq = t1k_arena_append(P.counters, T1K_AR_WAVE, P.rareSegCap);
          if (wq != T1K_ARENA_FULL) P.waveStr[wq] = gi;
        }
      }
    }
    base = __shfl(base, 0, 64);
    if (n <= WAVE_CAP && base != T1K_ARENA_FULL) {
      uint32_t w = base + incl - mine;
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        const uint32_t u = b + lane + 64 * r;
        if (cnt[r]) {
          const uint32_t rOff = uo[4 * u];
          for (uint32_t j = 0; j < cnt[r]; ++j) P.genHits[w++] = (P.ref.kPost[first[r] + j].offset << 12) | rOff;
        }
      }
    }
  }
}

// K5b: multi-diagonal groups with at most GENERAL_SMALL hits, one lane per group, 64-thread workgroups; the three work arrays
// live in LDS, interleaved over the lanes (24 KB).  Alignments go through the read-end's memo (registered now, run by
// k_dp_dense, added by k_general_finish); groups that need an alignment wider than the register band go to k_chain_big.
// useSimple: record word 6 = 1 marks a group whose hit list k_near_hits wrote as its chain (groupSimple: no sorts, no LIS, one array -- up
// to 3 * GENERAL_SMALL hits in the same LDS)
__global__ __launch_bounds__(64) void k_chain_general(ChainArgs P, uint32_t nItems, int useSimple, const unsigned long long *nDev) {
  __shared__ uint32_t sArr[3 * GENERAL_SMALL * 64];
  if (nDev) { if (P.counters[2]) return; nItems = (uint32_t)*nDev; }
  unsigned int dpLocal = 0, genLocal = 0;
  for (uint32_t qb = blockIdx.x * 64; qb < nItems; qb += gridDim.x * 64) {  // (a lane's work arrays are its own: no barrier between the rounds)
  const uint32_t q = qb + threadIdx.x;
  uint32_t nHits = 0xFFFFFFFFu;
  bool simple = false;
  if (q < nItems) {
    const uint32_t *r0 = P.recs + (uint64_t)P.generalList[q] * P.recStride;
    nHits = r0[4];
    simple = useSimple && r0[5] == REC_NEAR_DONE && r0[6] == 1u;
  }
  if (nHits <= GENERAL_SMALL || (simple && nHits <= 3 * GENERAL_SMALL)) {
    const uint32_t gi = P.generalList[q];
    uint32_t *rec = P.recs + (uint64_t)gi * P.recStride;
    const uint32_t re = rec[0] & 0x7FFFFFFFu, allele = rec[1];
    const int pass = (rec[0] >> 31) ? 0 : 1;
    ReadCtx c = makeCtx(P, re, pass, allele);
    const int n = (int)nHits;
    if (rec[3] == T1K_ARENA_FULL) atomicOr(&P.counters[2], (unsigned long long)ERR_STAGECAP);
    const uint32_t *hh = P.genHits + (rec[3] == T1K_ARENA_FULL ? 0u : rec[3]);
    bool needScratch = false;
    uint32_t cbuf[2 * GENERAL_SMALL + 6];  // a candidate needs >= 3 hits
    CandOut out{cbuf, 0, 6, GENERAL_SMALL / 3 + 1};
    const GapSink sink{P.memo + (uint64_t)re * GAP_CACHE, P.genJobStr, P.counters, re * GAP_CACHE, P.genJobSegCap, T1K_AR_GENJOBS};
    LaneArr A{sArr + threadIdx.x}, B{sArr + GENERAL_SMALL * 64 + threadIdx.x}, C{sArr + 2 * GENERAL_SMALL * 64 + threadIdx.x};
    if (simple) groupSimple(hh, n, c, P.k, P.hitLenRequired, A, out, &dpLocal, &P.counters[2], &needScratch, &sink, pass);
    else groupGeneral(hh, n, c, P.k, P.radius, P.hitLenRequired, A, B, C, nullptr, 0, out, &dpLocal, &P.counters[2], &needScratch, &sink, pass);
    if (needScratch) { const uint32_t b = t1k_arena_append(P.counters, T1K_AR_BIG, P.rareSegCap); if (b != T1K_ARENA_FULL) P.bigStr[b] = gi; }
    else {
      ++genLocal;
      uint32_t base = 0;
      if (out.n) {
        base = t1k_arena_alloc(P.counters, T1K_AR_GENCAND, (uint32_t)out.n, P.genCandSegCap);
        if (base == T1K_ARENA_FULL) { atomicOr(&P.counters[2], (unsigned long long)ERR_STAGECAP); out.n = 0; base = 0; }
        for (int i = 0; i < 6 * out.n; ++i) P.genCand[(uint64_t)base * 6 + i] = cbuf[i];
      }
      rec[3] = base;
      rec[2] = REC_DONE | 0x40000000u | (uint32_t)out.n;  // bit30: candidates live in the side arena
    }
  }
  }
  t1k_stat_add(P.counters, T1K_STAT_DP, dpLocal);
  t1k_stat_add(P.counters, T1K_STAT_GENERAL, genLocal);
}

// K5b': multi-diagonal groups with GENERAL_SMALL < hits <= WAVE_CAP, one wavefront per group.  The quadratic steps of
// GetOverlapsFromHits (sort by diagonal, nearest-to-dominant filter, sort by allele offset; SeqSet.hpp:1338-1456) are spread
// over the lanes as rank sorts; the LIS and the chain walk of each diagonal run are done by lane 0 (chainRun).
// cap / above: this launch takes the groups with above < hits <= cap and has 3 * cap words of LDS (round 5: the list is run twice -- most of
// these groups hold a few dozen hits, and with room for WAVE_CAP of them a workgroup's 48 KB left three wavefronts on a compute unit)
__global__ __launch_bounds__(64) void k_chain_wave(ChainArgs P, uint32_t nItems, const unsigned long long *nDev, uint32_t cap, uint32_t above) {
  extern __shared__ uint32_t sW[];  // A | B | C, cap words each
  if (nDev) { if (P.counters[2]) return; nItems = (uint32_t)*nDev; }
  uint32_t *A = sW, *B = sW + cap, *C = sW + 2 * cap;
  const int lane = threadIdx.x;
  unsigned int dpLocal = 0, genLocal = 0;
  auto diagOf = [](uint32_t x) { return (int)(x & 0xFFF) - (int)(x >> 12); };
  for (uint32_t q = blockIdx.x; q < nItems; q += gridDim.x) {
    const uint32_t gi = P.waveList[q];
    uint32_t *rec = P.recs + (uint64_t)gi * P.recStride;
    const uint32_t re = rec[0] & 0x7FFFFFFFu, allele = rec[1];
    const int pass = (rec[0] >> 31) ? 0 : 1;
    const int n = (int)rec[4];
    if ((uint32_t)n <= above || (uint32_t)n > cap) continue;  // (the other launch's; uniform over the wavefront)
    const bool lost = rec[3] == T1K_ARENA_FULL;
    if (lost && lane == 0) atomicOr(&P.counters[2], (unsigned long long)ERR_STAGECAP);
    const uint32_t *hh = P.genHits + (lost ? 0u : rec[3]);
    ReadCtx c = makeCtx(P, re, pass, allele);
    const int k = P.k;
    __syncthreads();  // the previous group's arrays are dead
    for (int i = lane; i < n; i += 64) C[i] = hh[i];
    __syncthreads();
    for (int i = lane; i < n; i += 64) {  // rank sort by (diagonal, allele offset, read offset): CompSortHitCoordDiff (266-274)
      const uint32_t x = C[i];
      int r = 0;
      for (int j = 0; j < n; ++j) r += hitKeyLess(C[j], x) ? 1 : 0;
      A[r] = x;
    }
    __syncthreads();
    bool needScratch = false;
    uint32_t cbuf[192];  // lane 0: at most 32 candidates are kept per group
    CandOut out{cbuf, 0, 6, 32};
    const GapSink sink{P.memo + (uint64_t)re * GAP_CACHE, P.genJobStr, P.counters, re * GAP_CACHE, P.genJobSegCap, T1K_AR_GENJOBS};
    for (int s = 0; s < n;) {  // wave-uniform: every lane tracks the run boundaries
      int curDiff = diagOf(A[s]), curCnt = 1, domCnt = 0, dominant = 0;
      int e = s + 1;
      for (; e < n; ++e) {
        int d = diagOf(A[e]) - diagOf(A[e - 1]);
        if (d < 0) d = -d;
        if (d > P.radius) break;
        if (d == 0) ++curCnt;
        else {
          if (curCnt > domCnt) { dominant = curDiff; domCnt = curCnt; }
          curDiff = diagOf(A[e]); curCnt = 1;
        }
      }
      if (curCnt > domCnt) dominant = curDiff;
      if (e - s < 3 || (e - s) * k < P.hitLenRequired) { s = e; continue; }
      // nearest-to-dominant filter per read offset (1437-1456): keep flags in C
      int mine = 0;
      for (int qq = s + lane; qq < e; qq += 64) {
        const int a = (int)(A[qq] & 0xFFF);
        int dq = diagOf(A[qq]) - dominant; if (dq < 0) dq = -dq;
        bool keep = true;
        for (int r = s; r < e; ++r) {
          if ((int)(A[r] & 0xFFF) != a) continue;
          int dr = diagOf(A[r]) - dominant; if (dr < 0) dr = -dr;
          if (dr < dq) { keep = false; break; }
        }
        C[qq] = keep ? 1u : 0u;
        mine += keep ? 1 : 0;
      }
      for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
      const int m = mine;
      __syncthreads();
      for (int qq = s + lane; qq < e; qq += 64) {  // kept hits by (allele offset, read offset) == packed value (CompSortPairBInc)
        if (!C[qq]) continue;
        const uint32_t x = A[qq];
        int r = 0;
        for (int t = s; t < e; ++t) r += (C[t] && A[t] < x) ? 1 : 0;
        B[r] = x;
      }
      __syncthreads();
      if (lane == 0) chainRun(c, k, P.hitLenRequired, A, B, C, s, m, nullptr, 0, out, &dpLocal, &P.counters[2], &needScratch, &sink, pass);
      __syncthreads();
      s = e;
    }
    if (lane == 0) {
      if (out.overflow) atomicOr(&P.counters[2], (unsigned long long)ERR_BIGGROUP);
      if (needScratch) { const uint32_t b = t1k_arena_append(P.counters, T1K_AR_BIG, P.rareSegCap); if (b != T1K_ARENA_FULL) P.bigStr[b] = gi; }
      else {
        ++genLocal;
        uint32_t base = 0;
        if (out.n) {
          base = t1k_arena_alloc(P.counters, T1K_AR_GENCAND, (uint32_t)out.n, P.genCandSegCap);
          if (base == T1K_ARENA_FULL) { atomicOr(&P.counters[2], (unsigned long long)ERR_STAGECAP); out.n = 0; base = 0; }
          for (int i = 0; i < 6 * out.n; ++i) P.genCand[(uint64_t)base * 6 + i] = cbuf[i];
        }
        rec[3] = base;
        rec[2] = REC_DONE | 0x40000000u | (uint32_t)out.n;
      }
    }
  }
  t1k_stat_add(P.counters, T1K_STAT_DP, dpLocal);
  t1k_stat_add(P.counters, T1K_STAT_GENERAL, genLocal);
}

// K5c: add the memo's match counts to the candidates of the multi-diagonal groups
__global__ __launch_bounds__(WG) void k_general_finish(ChainArgs P, uint32_t nItems, const unsigned long long *nDev) {
  if (nDev) { if (P.counters[2]) return; nItems = (uint32_t)*nDev; }
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nItems; q += gridDim.x * blockDim.x) {
  const uint32_t *rec = P.recs + (uint64_t)P.generalList[q] * P.recStride;
  if ((rec[2] & (REC_DONE | 0x40000000u)) != (REC_DONE | 0x40000000u)) continue;  // not chained yet (waits for k_chain_big)
  const uint32_t re = rec[0] & 0x7FFFFFFFu, nc = rec[2] & 0x3FFFFFFFu;
  const unsigned long long *memo = P.memo + (uint64_t)re * GAP_CACHE;
  for (uint32_t j = 0; j < nc; ++j) {
    uint32_t *g = P.genCand + ((uint64_t)rec[3] + j) * 6;
    const int nref = (int)((g[0] >> 24) & 0xF);
    if (!nref) continue;
    uint32_t sum = 0;
    for (int i = 0; i < nref; ++i) {
      const uint32_t slot = (g[3 + (i >> 1)] >> (16 * (i & 1))) & 0xFFFFu;
      const unsigned long long e = __hip_atomic_load(&memo[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t v = GAP_VAL(e);
      if (v == GAP_PENDING) atomicOr(&P.counters[2], (unsigned long long)ERR_MEMO);
      sum += v;
    }
    g[2] += (2u * sum) << 20;
    g[0] &= 0x00FFFFFFu;
  }
  }
}

// what is left: groups with more than WAVE_CAP hits, or a gap whose two sides differ by more than the register band covers.
// One working lane per 64-thread workgroup; sort arrays in HBM scratch, the rows of the general alignment in LDS.
__global__ __launch_bounds__(64) void k_chain_big(ChainArgs P, uint32_t nItems, const unsigned long long *nDev) {
  __shared__ int sGa[GA_SCRATCH_INTS];
  if (threadIdx.x != 0) return;
  if (nDev) { if (P.counters[2]) return; nItems = (uint32_t)*nDev; }
  const uint32_t t = blockIdx.x, nT = gridDim.x;
  uint32_t *mine = P.bigScratch + (uint64_t)t * (4 * BIG_CAP);
  unsigned int dpLocal = 0;
  for (uint32_t q = t; q < nItems; q += nT) {
    const uint32_t gi = P.bigList[q];
    uint32_t *rec = P.recs + (uint64_t)gi * P.recStride;
    const uint32_t re = rec[0] & 0x7FFFFFFFu, allele = rec[1];
    const int pass = (rec[0] >> 31) ? 0 : 1;
    ReadCtx c = makeCtx(P, re, pass, allele);
    uint32_t *hh = mine + 3 * BIG_CAP;
    const int n = gatherHits(P, re, pass, allele, hh, BIG_CAP);
    if (n > BIG_CAP) { atomicOr(&P.counters[2], (unsigned long long)ERR_BIGGROUP); rec[2] = REC_DONE; continue; }
    bool dummy = false;
    uint32_t cbuf[192];  // at most 32 candidates are kept per group
    CandOut o2{cbuf, 0, 6, 32};
    groupGeneral(hh, n, c, P.k, P.radius, P.hitLenRequired, mine, mine + BIG_CAP, mine + 2 * BIG_CAP, sGa, GA_BIG_MAX, o2, &dpLocal,
                 &P.counters[2], &dummy);
    if (o2.overflow) atomicOr(&P.counters[2], (unsigned long long)ERR_BIGGROUP);
    uint32_t base = 0;
    if (o2.n) {
      base = t1k_arena_alloc(P.counters, T1K_AR_GENCAND, (uint32_t)o2.n, P.genCandSegCap);
      if (base == T1K_ARENA_FULL) { atomicOr(&P.counters[2], (unsigned long long)ERR_STAGECAP); o2.n = 0; base = 0; }
      for (int i = 0; i < 6 * o2.n; ++i) P.genCand[(uint64_t)base * 6 + i] = cbuf[i];
    }
    rec[3] = base;
    rec[2] = REC_DONE | 0x40000000u | (uint32_t)o2.n;
    atomicAdd(&P.counters[13], 1ull);
  }
  if (dpLocal) atomicAdd(&P.counters[7], (unsigned long long)dpLocal);
}

// ------------------------------------------------------------------------------------------------------------------
// K6: strand vote + copy-out, one workgroup per read-end
// ------------------------------------------------------------------------------------------------------------------

// the similarity / low-complexity filter on a seed candidate (SeqSet.hpp:1838-1845, 1894-1908): candidates that fail it take no
// further part (they only counted for the strand vote), so they are never copied out.  The low-complexity test (SeqSet.hpp:458-485)
// asks for the number of A / C / G / T among the non-N bases of a read span; a workgroup works on one read-end, so it keeps prefix
// counts of the four bases for both strands in LDS (sBaseCnt[strand][base][p] = occurrences in [0, p)) and a span costs eight reads.
template <int MAXLEN>
__device__ __forceinline__ bool lowComplexityFromCounts(const uint16_t (*cnt)[MAXLEN + 1], int rs, int re) {
  const int L = re - rs + 1;
  int low = 0, lowTotal = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int c = (int)cnt[b][re + 1] - (int)cnt[b][rs];
    if (c <= 2) { ++low; lowTotal += c; }
  }
  if (lowTotal * 7 >= L) return false;
  return low >= 2;
}
template <int MAXLEN>
__device__ __forceinline__ bool keepCandidate(const ChainArgs &P, const uint16_t (*baseCnt)[4][MAXLEN + 1], int plus, uint32_t w0, uint32_t w1, uint32_t w2) {
  const int rs = (int)(w0 & 0xFFF), rend = (int)((w0 >> 12) & 0xFFF), ss = (int)(w1 & 0xFFFFF), se = (int)(w2 & 0xFFFFF);
  const int matchCnt = (int)(w2 >> 20);
  const double sim = (double)matchCnt / (double)(se - ss + 1 + rend - rs + 1);
  if (sim < P.sim) return false;
  if (lowComplexityFromCounts<MAXLEN>(baseCnt[plus ? 0 : 1], rs, rend)) return !(0.0 < P.sim);  // similarity becomes 0
  return true;
}

// A single-diagonal group whose candidate still waits for registered alignments (record word 2 = their number, words 6..7 their memo
// slots; k_dp_dense has run them): the match counts are added here, where the record is read anyway -- k_chain_finish used to fetch and
// rewrite those records in a launch of its own (1.9 GB fetched per range for scattered 32-byte records).  Returns the candidate's word 2.
__device__ __forceinline__ uint32_t pendingMatchWord(const ChainArgs &P, uint32_t re, uint32_t nRefs, uint32_t w2, uint32_t s0, uint32_t s1) {
  const unsigned long long *memo = P.memo + (uint64_t)re * GAP_CACHE;
  uint32_t sum = 0;
  for (uint32_t i = 0; i < nRefs; ++i) {
    const uint32_t slot = ((i >> 1 ? s1 : s0) >> (16 * (i & 1))) & 0xFFFFu;
    const unsigned long long e = __hip_atomic_load(&memo[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t v = GAP_VAL(e);
    if (v == GAP_PENDING) atomicOr(&P.counters[2], (unsigned long long)ERR_MEMO);  // cannot happen: every registered job is run by k_dp_dense
    sum += v;
  }
  return w2 + ((2u * sum) << 20);
}
__device__ __forceinline__ bool recPending(uint32_t state) { return !(state & REC_DONE) && state >= 1u && state <= (uint32_t)GROUP_MAX_REFS; }

// MAXLEN: longest read of the launch (T1K_MAX_READ_LEN, or T1K_LONG_READ_LEN for a window with longer reads: 16 KB of prefix counts)
#if T1K_COLLECT_WAVES > 0
#define T1K_COLLECT_ATTR T1K_WAVES_ATTR_(T1K_COLLECT_WAVES)
#else
#define T1K_COLLECT_ATTR
#endif
template <int MAXLEN>
__global__ __launch_bounds__(WG) T1K_COLLECT_ATTR void k_collect(ChainArgs P) {
  __shared__ uint32_t warpSums[4];
  __shared__ uint16_t sBaseCnt[2][4][MAXLEN + 1];
  __shared__ uint64_t sVote[2 * WG];
  uint64_t *sVoteHi = sVote, *sVoteLo = sVote + WG;
  __shared__ uint32_t sBase;
  // the read-end's chunk table (at most 2 x 256 + 2 chunks: the launcher refuses references with more seeding chunks): record index of
  // each chunk's first record, prefix of the chunk lengths, the chunks' strands; the copy-out pass's own prefix lives in the vote arrays,
  // which are dead by then
  constexpr int MAXCH = 2 * 256 + 2;
  __shared__ uint32_t sCs[MAXCH], sPre[MAXCH + 1], sNCh;
  __shared__ uint8_t sChPlus[MAXCH];
  uint32_t *sPreW = (uint32_t *)sVote;
  static_assert(sizeof(uint64_t) * 2 * WG >= sizeof(uint32_t) * (MAXCH + 1), "the copy-out prefix does not fit the vote arrays");
  // one bit per group record of the read-end (in chunk order): "the copy-out pass has to look at this record" -- its only candidate passed
  // the filter, or it holds several (multi-diagonal groups).  The vote pass reads every record anyway; the copy-out pass then loads the
  // 40 % that hold something instead of all of them again (records beyond the bitmap's reach are simply looked at).
  constexpr uint32_t KEEP_BITS = 65536;
  __shared__ uint32_t sKeepBits[KEEP_BITS / 32];
  const int tid = threadIdx.x;
  const uint32_t stride = P.recStride;
  if (P.devDriven && P.counters[2]) return;  // an arena overflowed earlier in this submission: the range runs again
  __shared__ uint32_t sNextRe;  // read-ends handed out one at a time (device counter): their group counts differ by orders of magnitude
  uint32_t hoNext = 0, hoLeft = 0;
  for (;;) {
    __syncthreads();
    if (tid == 0) {  // (T1K_RE_HANDOUT read-ends per atomic on the hand-out word: round 6)
      if (hoLeft == 0) { hoNext = (uint32_t)atomicAdd(&P.counters[27], (unsigned long long)T1K_COLLECT_HANDOUT); hoLeft = T1K_COLLECT_HANDOUT; }
      sNextRe = hoNext++; --hoLeft;
    }
    __syncthreads();
    const uint32_t re = sNextRe;
    if (re >= P.reads.nReadEnds) break;
    const uint32_t *cs = P.chunkStart + (uint64_t)re * P.maxChunks, *cc = P.chunkCount + (uint64_t)re * P.maxChunks;
    for (uint32_t i = tid; i < KEEP_BITS / 32; i += WG) sKeepBits[i] = 0;
    {  // prefix counts of the four bases (non-N positions only) of both strands of this read-end
      const int len = (int)P.reads.len[re], S = P.reads.S;
      for (int idx = tid; idx < 2 * (len + 1); idx += WG) {
        const int strand = idx / (len + 1), p = idx - strand * (len + 1);
        const uint64_t *rb = P.reads.bases + ((uint64_t)re * 2 + strand) * S, *rn = P.reads.nmask + ((uint64_t)re * 2 + strand) * S;
        int c[4] = {0, 0, 0, 0};
        for (int o = 0; o < p; o += 32) {
          const uint64_t x = rb[o >> 5], nn = rn[o >> 5];
          const uint64_t valid = T1K_EVEN & ~nn & t1k_lowmask(p - o);
          const uint64_t lo = x & T1K_EVEN, hi = (x >> 1) & T1K_EVEN;
          c[0] += __popcll(~lo & ~hi & valid); c[1] += __popcll(lo & ~hi & valid); c[2] += __popcll(~lo & hi & valid); c[3] += __popcll(lo & hi & valid);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) sBaseCnt[strand][b][p] = (uint16_t)c[b];
      }
      __syncthreads();
    }
    // The chunk table of the read-end (runs of records per strand and allele chunk, recorded densely) as a prefix over ONE record index r:
    // both passes below walk r in strides of the workgroup, so a pass issues ceil(records / 256) rounds of loads -- chunk by chunk it was
    // one or two rounds per chunk, ten to twenty chunks a read-end, each behind a dependent load of the chunk's start and count.
    if (tid < 64) {
      uint32_t carry = 0, nDense = 0;
      bool open = true;
      if (tid == 0) sPre[0] = 0;
      for (int base = 0; base < P.maxChunks && open; base += 64) {
        const int ch = base + tid;
        const uint32_t n = ch < P.maxChunks ? cc[ch] : 0u;
        const uint32_t st = ch < P.maxChunks ? cs[ch] : 0u;  // (requested with the count: used only for the recorded chunks)
        const unsigned long long zero = __ballot(n == 0);
        const int live = zero ? __ffsll((long long)zero) - 1 : 64;  // chunks of this strip in front of the first empty one
        uint32_t x = tid < live ? n : 0u;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o, 64); if (tid >= o) x += y; }
        if (tid < live) { sCs[ch] = st; sPre[ch + 1] = carry + x; }
        carry += __shfl(x, 63, 64);
        nDense += (uint32_t)live;
        open = live == 64;
      }
      if (tid == 0) sNCh = nDense;
    }
    __syncthreads();
    const int nCh = (int)sNCh;
    const uint32_t nRec = sPre[nCh];
    auto chunkOf = [&](uint32_t r) {  // the chunk that holds record index r: the last one that starts at or before it
      int lo = 0, hi = nCh;
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sPre[mid] <= r) lo = mid; else hi = mid; }
      return lo;
    };
    VoteKey best; best.hi = ~0ull; best.lo = ~0ull;
    uint32_t nCand[2] = {0, 0};
    for (uint32_t r0 = tid; r0 < nRec; r0 += 2 * WG) {
      // two records per lane and round, both requested before either is looked at
      uint4 hdv[2], cwv[2];
      uint32_t rv[2];
      bool first[2];
      int chv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        rv[u] = r0 + u * WG;
        hdv[u] = make_uint4(0u, 0u, 0u, 0u); cwv[u] = hdv[u]; first[u] = false; chv[u] = 0;
        if (rv[u] < nRec) {
          chv[u] = chunkOf(rv[u]);
          first[u] = rv[u] == sPre[chv[u]];
          const uint32_t *rec = P.recs + (uint64_t)(sCs[chv[u]] + (rv[u] - sPre[chv[u]])) * stride;
          hdv[u] = ((const uint4 *)rec)[0]; cwv[u] = ((const uint4 *)rec)[1];  // words 0..3: re|strand, allele, state, candidate word 0 (or side-arena base); 4..5: candidate words 1, 2
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (rv[u] >= nRec) continue;
        const uint4 hd = hdv[u], cw = cwv[u];
        const bool pend = recPending(hd.z);
        const uint32_t nc = pend ? 1u : ((hd.z & REC_DONE) ? (hd.z & 0x3FFFFFFFu) : 0u);   // (neither: a group the closed-form pass ended without a candidate, left as seeded)
        const uint32_t w2own = pend ? pendingMatchWord(P, re, hd.z, cw.y, cw.z, cw.w) : cw.y;
        const int plus = (int)(hd.x >> 31);
        if (first[u]) sChPlus[chv[u]] = (uint8_t)plus;  // a chunk holds one strand
        uint32_t kept = 0;
        for (uint32_t j = 0; j < nc; ++j) {
          uint32_t w0 = hd.w, w1 = cw.x, w2 = w2own;
          if (hd.z & 0x40000000u) { const uint32_t *g = P.genCand + ((uint64_t)hd.w + j) * 6; w0 = g[0]; w1 = g[1]; w2 = g[2]; }
          VoteKey vk = voteKey((int)(w1 >> 20), (int)(w0 & 0xFFF), (int)((w0 >> 12) & 0xFFF), hd.y, plus, (int)(w1 & 0xFFFFF), (int)(w2 & 0xFFFFF));
          if (vk < best) best = vk;
          const uint32_t kc = keepCandidate<MAXLEN>(P, sBaseCnt, plus, w0, w1, w2) ? 1u : 0u;
          nCand[plus] += kc; kept += kc;
        }
        if (kept && rv[u] < KEEP_BITS) atomicOr(&sKeepBits[rv[u] >> 5], 1u << (rv[u] & 31));
      }
    }
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
    __syncthreads();
    uint32_t totWin;
    t1k_block_scan_exclusive(nCand[winPlus], warpSums, &totWin);
    if (tid == 0) {
      unsigned long long b = totWin ? atomicAdd(&P.counters[0], (unsigned long long)totWin) : 0ull;
      if (b + totWin > P.candCap) { atomicOr(&P.counters[2], (unsigned long long)ERR_CANDCAP); sBase = 0xFFFFFFFFu; P.candStart[re] = 0; P.candCount[re] = 0; }
      else { sBase = (uint32_t)b; P.candStart[re] = (uint32_t)b; P.candCount[re] = totWin; }
    }
    __syncthreads();
    if (sBase != 0xFFFFFFFFu && totWin) {
      // chunks are in the reference's order ('-' strand first, alleles ascending); copy the winning strand's candidates in order:
      // a second prefix, over the winning strand's chunks alone (the others are empty in it), gives the pass its own dense index
      if (tid < 64) {
        uint32_t carry = 0;
        if (tid == 0) sPreW[0] = 0;
        for (int base = 0; base < nCh; base += 64) {
          const int ch = base + tid;
          uint32_t x = (ch < nCh && sChPlus[ch] == (uint8_t)winPlus) ? sPre[ch + 1] - sPre[ch] : 0u;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o, 64); if (tid >= o) x += y; }
          if (ch < nCh) sPreW[ch + 1] = carry + x;
          carry += __shfl(x, 63, 64);
        }
      }
      __syncthreads();
      const uint32_t nWin = sPreW[nCh];
      uint32_t written = 0;
      for (uint32_t i0 = 0; i0 < nWin; i0 += WG) {
        const uint32_t q = i0 + tid;
        uint32_t ridx = 0, g = 0;
        if (q < nWin) {
          int lo = 0, hi = nCh;  // the last chunk that starts at or before q in the winning strand's index (an empty one never is the last)
          while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sPreW[mid] <= q) lo = mid; else hi = mid; }
          ridx = sPre[lo] + (q - sPreW[lo]);
          g = sCs[lo] + (q - sPreW[lo]);
        }
        const bool look = q < nWin && (ridx >= KEEP_BITS || ((sKeepBits[ridx >> 5] >> (ridx & 31)) & 1u));
        uint4 hd = make_uint4(0u, 0u, 0u, 0u), cw = make_uint4(0u, 0u, 0u, 0u);
        if (look) { const uint32_t *rec = P.recs + (uint64_t)g * stride; hd = ((const uint4 *)rec)[0]; cw = ((const uint4 *)rec)[1]; }
        const bool pend = look && recPending(hd.z);
        const uint32_t nc = look ? (pend ? 1u : ((hd.z & REC_DONE) ? (hd.z & 0x3FFFFFFFu) : 0u)) : 0;
        const uint32_t w2own = pend ? pendingMatchWord(P, re, hd.z, cw.y, cw.z, cw.w) : cw.y;
        uint32_t keepMask = 0, nk = 0;  // a group holds at most 32 candidates
        for (uint32_t j = 0; j < nc; ++j) {
          uint32_t w0 = hd.w, w1 = cw.x, w2 = w2own;
          if (hd.z & 0x40000000u) { const uint32_t *gc = P.genCand + ((uint64_t)hd.w + j) * 6; w0 = gc[0]; w1 = gc[1]; w2 = gc[2]; }
          if (keepCandidate<MAXLEN>(P, sBaseCnt, (int)winPlus, w0, w1, w2)) { keepMask |= 1u << j; ++nk; }
        }
        uint32_t tot;
        uint32_t off = t1k_block_scan_exclusive(nk, warpSums, &tot);
        for (uint32_t j = 0; j < nc; ++j) {
          if (!((keepMask >> j) & 1u)) continue;
          uint32_t w0 = hd.w, w1 = cw.x, w2 = w2own;
          if (hd.z & 0x40000000u) { const uint32_t *gc = P.genCand + ((uint64_t)hd.w + j) * 6; w0 = gc[0]; w1 = gc[1]; w2 = gc[2]; }
          T1kCand cd;
          cd.allele = hd.y | (winPlus ? 0x80000000u : 0);
          cd.readSE = (w0 & 0xFFF) | (((w0 >> 12) & 0xFFF) << 16);
          cd.seqStart = (int)(w1 & 0xFFFFF); cd.seqEnd = (int)(w2 & 0xFFFFF);
          cd.match = (w1 >> 20) | ((w2 >> 20) << 16);
          cd.re = re;
          P.cand[(uint64_t)sBase + written + off] = cd;
          ++off;
        }
        written += tot;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
size_t t1k_chain_big_scratch_u32() { return (size_t)4 * BIG_CAP + GA_SCRATCH_INTS; }
int t1k_chain_max_chunks(uint32_t nAlleles) { return 2 * (int)((nAlleles + CHUNK_A - 1) / CHUNK_A) + 2; }
int t1k_chain_memo_entries() { return GAP_CACHE; }
int t1k_chain_rec_stride(int maxLen) { return maxLen <= 160 ? 8 : 16; }  // u32 per record (32 / 64 bytes)
int t1k_chain_max_kmers(int maxLen, int k) { return (2 * std::max(1, maxLen - k + 1) + 3) / 4 * 4; }
int t1k_chain_used_u32(int maxK) { return maxK * 4; }  // per used list: read offset, list start, list length, directory row

static int readCounters(t1k_ctx *ctx, unsigned long long *h) { return t1k_fetch_counters(ctx, h); }

// striped list -> dense list; grid (blocks, T1K_NSTRIPE)
// The first workgroup also leaves the dense list's length in the arena's total word (T1K_TOTAL_BASE) for the consumers that take their item
// count from the device, and raises `overflowFlag` in the control word when a stripe is full (the range then runs again with larger lists).
template <class V>
__global__ __launch_bounds__(WG) void k_arena_compact(const V *src, uint32_t segCap, unsigned long long *counters, int arena, V *dst, unsigned long long overflowFlag) {
  const unsigned long long *cursors = counters + T1K_ARENA_BASE + (size_t)arena * T1K_NSTRIPE * T1K_STRIPE_WORDS;
  const uint32_t seg = blockIdx.y;
  uint32_t prefix = 0;
  for (uint32_t s = 0; s < seg; ++s) prefix += (uint32_t)min(cursors[s * T1K_STRIPE_WORDS], (unsigned long long)segCap);
  const uint32_t cnt = (uint32_t)min(cursors[seg * T1K_STRIPE_WORDS], (unsigned long long)segCap);
  if (blockIdx.x == 0 && seg == T1K_NSTRIPE - 1 && threadIdx.x == 0) {
    counters[T1K_TOTAL_BASE + arena] = (unsigned long long)prefix + cnt;
    bool over = false;
    for (uint32_t s = 0; s < T1K_NSTRIPE; ++s) over = over || cursors[s * T1K_STRIPE_WORDS] > (unsigned long long)segCap;
    if (over && overflowFlag) atomicOr(&counters[2], overflowFlag);
  }
  for (uint32_t i = blockIdx.x * WG + threadIdx.x; i < cnt; i += gridDim.x * WG) dst[prefix + i] = src[(uint64_t)seg * segCap + i];
}

T1kArenaCounts t1k_arena_counts(const t1k_ctx *ctx, int arena, uint32_t segCap) {
  T1kArenaCounts r{0, 0, false};
  for (int s = 0; s < T1K_NSTRIPE; ++s) {
    unsigned long long c = ctx->hRaw[T1K_ARENA_BASE + ((size_t)arena * T1K_NSTRIPE + s) * T1K_STRIPE_WORDS];
    if (c > segCap) { r.overflow = true; c = segCap; }
    r.total += c;
    r.maxSeg = std::max<uint32_t>(r.maxSeg, (uint32_t)c);
  }
  return r;
}

void t1k_arena_compact(t1k_ctx *ctx, int arena, const uint32_t *src, uint32_t segCap, uint32_t *dst, uint32_t maxSeg) {
  if (!maxSeg) return;
  hipLaunchKernelGGL(k_arena_compact<uint32_t>, dim3(std::min<uint32_t>((maxSeg + WG - 1) / WG, 256u), T1K_NSTRIPE), dim3(WG), 0, ctx->stream, src, segCap,
                     (unsigned long long *)ctx->bCounters.p, arena, dst, 0ull);
}
void t1k_arena_compact64(t1k_ctx *ctx, int arena, const unsigned long long *src, uint32_t segCap, unsigned long long *dst, uint32_t maxSeg) {
  if (!maxSeg) return;
  hipLaunchKernelGGL(k_arena_compact<unsigned long long>, dim3(std::min<uint32_t>((maxSeg + WG - 1) / WG, 256u), T1K_NSTRIPE), dim3(WG), 0, ctx->stream, src, segCap,
                     (unsigned long long *)ctx->bCounters.p, arena, dst, 0ull);
}
// the device-driven form: the grid is sized from an estimate of the fullest stripe (the kernel strides over what is there), the list's length goes to
// the arena's total word, a full stripe raises ERR_GROUPCAP on the device
void t1k_arena_compact_dev(t1k_ctx *ctx, int arena, const uint32_t *src, uint32_t segCap, uint32_t *dst, uint64_t estTotal) {
  const uint32_t perSeg = (uint32_t)std::min<uint64_t>(segCap, estTotal / T1K_NSTRIPE * 2 + WG);
  hipLaunchKernelGGL(k_arena_compact<uint32_t>, dim3(std::max(1u, std::min<uint32_t>((perSeg + WG - 1) / WG, 256u)), T1K_NSTRIPE), dim3(WG), 0, ctx->stream, src, segCap,
                     (unsigned long long *)ctx->bCounters.p, arena, dst, (unsigned long long)ERR_GROUPCAP);
}

// sort key of a registered alignment: its read-window length, so that the lanes of a wavefront sweep DPs of equal height
__global__ __launch_bounds__(WG) void k_job_keys(const unsigned long long *memo, const uint32_t *jobs, unsigned long long *keys, uint32_t n) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) keys[q] = (memo[jobs[q]] >> 5) & 0x1FF;
}
// sort key of a multi-diagonal group: its hit count (record word 4 after k_gather_general), so that the lanes of k_chain_general's
// wavefronts work on groups of similar size
__global__ __launch_bounds__(WG) void k_group_size_keys(const uint32_t *recs, uint32_t stride, const uint32_t *list, unsigned long long *keys, uint32_t n, int useSimple) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) {
    const uint32_t *r = recs + (uint64_t)list[q] * stride;
    const bool simple = useSimple && r[5] == REC_NEAR_DONE && r[6] == 1u;  // chains first (they skip the sorts: their own wavefronts), each kind by size
    keys[q] = simple ? min(r[4], 127u) : 128u + min(r[4], 63u);
  }
}
// ------------------------------------------------------------------------------------------------------------------
// Counting sort of a dense work list by a small key (<= CS_BINS values), fused with the key's computation: the two in-loop orderings --
// registered alignments by read-window length (wavefronts of k_dp_dense sweep DPs of equal height: mixed lengths ran at 40 % lane
// utilisation), multi-diagonal groups by kind and hit count (k_chain_general) -- are 9- and 8-bit keys over ~10^6 items, for which a
// radix-sort library call (key kernel + three to four launches + its scan state) is out of proportion.  Two launches:
//   k_csort_count    LDS histogram per workgroup, its non-empty bins added to the global histogram
//   k_csort_scatter  every workgroup prefix-sums the global histogram itself (512 values), then tile by tile: LDS histogram with each
//                    item's rank inside the tile, one global atomic per non-empty bin reserves the tile's run inside the bin, items go
//                    out.  The order inside a bin is not defined (nothing downstream depends on it: the consumers' results are per
//                    item).  The last workgroup to finish clears the two global arrays for the next call (ticket counter).
// scratch: [hist CS_BINS | cursor CS_BINS | ticket] u32, zero when the first call starts (allocated zeroed, re-zeroed by every call).
// ------------------------------------------------------------------------------------------------------------------
#define CS_BINS 512
#define CS_TILE (WG * 8)
struct JobLenKey {  // read-window length of a registered alignment (memo entry bits 5..13)
  const unsigned long long *memo;
  __device__ __forceinline__ uint32_t operator()(uint32_t item) const { return (uint32_t)((memo[item] >> 5) & 0x1FF); }
};
struct GroupSizeKey {  // chains first (they skip the sorts: their own wavefronts), each kind by hit count (record word 4 after k_near_hits / k_gather_general)
  const uint32_t *recs; uint32_t stride; int useSimple;
  __device__ __forceinline__ uint32_t operator()(uint32_t item) const {
    const uint32_t *r = recs + (uint64_t)item * stride;
    const bool simple = useSimple && r[5] == REC_NEAR_DONE && r[6] == 1u;
    return simple ? min(r[4], 127u) : 128u + min(r[4], 63u);
  }
};
template <class KeyFn>
__global__ __launch_bounds__(WG) void k_csort_count(const uint32_t *list, uint32_t n, uint32_t *scratch, KeyFn key, const unsigned long long *nDev) {
  __shared__ uint32_t h[CS_BINS];
  if (nDev) n = (uint32_t)*nDev;
  for (int b = threadIdx.x; b < CS_BINS; b += WG) h[b] = 0;
  __syncthreads();
  for (uint32_t i = blockIdx.x * WG + threadIdx.x; i < n; i += gridDim.x * WG) atomicAdd(&h[key(list[i])], 1u);
  __syncthreads();
  for (int b = threadIdx.x; b < CS_BINS; b += WG)
    if (h[b]) atomicAdd(&scratch[b], h[b]);
}
template <class KeyFn>
__global__ __launch_bounds__(WG) void k_csort_scatter(const uint32_t *list, uint32_t n, uint32_t *scratch, uint32_t *out, KeyFn key, const unsigned long long *nDev) {
  __shared__ uint32_t base[CS_BINS], h[CS_BINS], off[CS_BINS], warpSums[4];
  __shared__ uint32_t sLast;
  if (nDev) n = (uint32_t)*nDev;
  uint32_t *hist = scratch, *cursor = scratch + CS_BINS, *ticket = scratch + 2 * CS_BINS;
  {  // exclusive prefix of the global histogram (two bins per thread)
    const uint32_t a = hist[2 * threadIdx.x], b = hist[2 * threadIdx.x + 1];
    uint32_t tot;
    const uint32_t ex = t1k_block_scan_exclusive(a + b, warpSums, &tot);
    base[2 * threadIdx.x] = ex; base[2 * threadIdx.x + 1] = ex + a;
  }
  for (uint32_t t0 = blockIdx.x * CS_TILE; t0 < n; t0 += gridDim.x * CS_TILE) {
    for (int b = threadIdx.x; b < CS_BINS; b += WG) h[b] = 0;
    __syncthreads();
    uint32_t item[8], kk[8], rk[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t i = t0 + j * WG + threadIdx.x;
      kk[j] = 0xFFFFFFFFu;
      if (i < n) { item[j] = list[i]; kk[j] = key(item[j]); rk[j] = atomicAdd(&h[kk[j]], 1u); }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < CS_BINS; b += WG)
      if (h[b]) off[b] = base[b] + atomicAdd(&cursor[b], h[b]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (kk[j] != 0xFFFFFFFFu) { const uint32_t pos = off[kk[j]] + rk[j]; if (pos < n) out[pos] = item[j]; }  // (pos < n always; the test keeps a scratch left dirty by a killed launch from writing outside the list)
    __syncthreads();
  }
  // the last workgroup out clears the arrays for the next call
  __threadfence();
  if (threadIdx.x == 0) sLast = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (sLast) {
    for (int b = threadIdx.x; b < 2 * CS_BINS; b += WG) scratch[b] = 0;
    if (threadIdx.x == 0) *ticket = 0;
  }
}
// list -> sorted (both n entries); returns `sorted`, or `list` when the scratch cannot be had
// nDev: the list's length is read on the device (n is then the most it can be: the sorted list's room; nEst sizes the grid)
template <class KeyFn>
static const uint32_t *countingSort(t1k_ctx *ctx, const uint32_t *list, uint32_t n, KeyFn key, const unsigned long long *nDev = nullptr, uint64_t nEst = 0) {
  static_assert(CS_BINS == 2 * WG, "two bins per thread in the prefix");
  const size_t head = (2 * CS_BINS + 16) * 4;
  const bool fresh = ctx->bJobSort.bytes < head + (size_t)n * 4;
  if (t1k_ensure(ctx, ctx->bJobSort, head + (size_t)n * 4) != T1K_OK) return list;
  uint32_t *scratch = (uint32_t *)ctx->bJobSort.p, *sorted = scratch + 2 * CS_BINS + 16;
  if (fresh && hipMemsetAsync(scratch, 0, head, ctx->stream) != hipSuccess) return list;  // (a new block: not zero yet; afterwards every call leaves it zero)
  const uint64_t forGrid = nDev ? std::min<uint64_t>(n, nEst) : n;
  const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((forGrid + CS_TILE - 1) / CS_TILE, 1024u));
  hipLaunchKernelGGL(k_csort_count<KeyFn>, dim3(grid), dim3(WG), 0, ctx->stream, list, n, scratch, key, nDev);
  hipLaunchKernelGGL(k_csort_scatter<KeyFn>, dim3(grid), dim3(WG), 0, ctx->stream, list, n, scratch, sorted, key, nDev);
  return sorted;
}
void t1k_launch_dp_dense(t1k_ctx *ctx, const ChainArgs &a, const uint32_t *jobs, uint32_t n) {
  if (!n) return;
  // order the jobs by length first: wavefronts of mixed lengths ran at 40 % lane utilisation (T1K_RADIX_SORTS=1: the rocPRIM radix sort of rounds 2-4)
  static const bool radix = getenv("T1K_RADIX_SORTS") != nullptr;
  if (n >= 4096 && !radix) jobs = countingSort(ctx, jobs, n, JobLenKey{(const unsigned long long *)a.memo});
  else if (n >= 4096 && t1k_ensure(ctx, ctx->bSlowKeys, (size_t)n * 20 + 64) == T1K_OK) {
    unsigned long long *k0 = (unsigned long long *)ctx->bSlowKeys.p, *k1 = k0 + n;
    uint32_t *sorted = (uint32_t *)(k1 + n);
    hipLaunchKernelGGL(k_job_keys, dim3((n + WG - 1) / WG), dim3(WG), 0, ctx->stream, (const unsigned long long *)a.memo, jobs, k0, n);
    if (t1k_sort_pairs(ctx, k0, k1, jobs, sorted, n, 9) == T1K_OK) jobs = sorted;
  }
  hipLaunchKernelGGL(k_dp_dense, dim3((n + WG - 1) / WG), dim3(WG), 0, ctx->stream, a, jobs, n, (const unsigned long long *)nullptr);
}
// the same with the number of jobs read on the device (arena's total word); cap = the most there can be, est sizes the grids
void t1k_launch_dp_dense_dev(t1k_ctx *ctx, const ChainArgs &a, const uint32_t *jobs, int arena, uint32_t cap, uint64_t est) {
  const unsigned long long *nDev = (const unsigned long long *)ctx->bCounters.p + T1K_TOTAL_BASE + arena;
  jobs = countingSort(ctx, jobs, cap, JobLenKey{(const unsigned long long *)a.memo}, nDev, est);
  const uint32_t grid = (uint32_t)std::max<uint64_t>(1, (std::min<uint64_t>(cap, est) + WG - 1) / WG);
  hipLaunchKernelGGL(k_dp_dense, dim3(grid), dim3(WG), 0, ctx->stream, a, jobs, cap, nDev);
}

// ------------------------------------------------------------------------------------------------------------------
// The chain of one range as ONE submission (round 5).  Rounds 1-4 fetched the counter block six times per range -- after seeding, after
// the closed-form pass, after the gap walk, twice inside the multi-diagonal path, at the end -- because every consumer's grid and item
// count were launch arguments.  Here every consumer reads its item count on the device (the total word k_arena_compact leaves where it
// makes a striped list dense, or the group arena's cursors) and strides over the items, so its grid only has to be roughly right: it is
// sized from what the context's previous range counted (per read-end; the first range of a context: from the arenas' capacities).  A full
// stripe raises ERR_GROUPCAP on the device; the kernels behind it work on the clamped lists (their results are thrown away: the caller
// runs the range again with the capacities the cursors ask for, as before).  One counter fetch is left, at the end.
// ------------------------------------------------------------------------------------------------------------------
// expected entries of an arena in a range of nRe read-ends (a quarter to spare), never more than it can hold
uint64_t t1k_arena_estimate(const t1k_ctx *ctx, int arena, uint64_t cap, uint32_t nRe) {
  if (!ctx->estValid || !ctx->estTotal[arena]) return cap;
  return std::min<uint64_t>(cap, (ctx->estTotal[arena] * std::max(1u, nRe) >> 16) * 5 / 4 + 4096);
}
void t1k_arena_estimate_set(t1k_ctx *ctx, int arena, uint64_t total, uint32_t nRe) { ctx->estTotal[arena] = (total << 16) / std::max(1u, nRe) + 1; }
bool t1k_chain_host_driven() { static const bool h = getenv("T1K_HOST_CHAIN") != nullptr; return h; }
static int runChainDevice(t1k_ctx *ctx, const ChainArgs &aIn, int nWg, int bigBlocks, bool longReads, bool xlong, unsigned long long *hc) {
  ChainArgs a = aIn;
  a.devDriven = 1;
  const unsigned long long *tot = (const unsigned long long *)ctx->bCounters.p + T1K_TOTAL_BASE;
  const uint32_t nRe = std::max<uint32_t>(1u, a.reads.nReadEnds);
  auto est = [&](int arena, uint64_t cap) -> uint64_t { return t1k_arena_estimate(ctx, arena, cap, nRe); };
  auto blocks = [](uint64_t items, uint32_t per) { return (uint32_t)std::max<uint64_t>(1, (items + per - 1) / per); };
  const uint64_t listCap = (uint64_t)a.listSegCap * T1K_NSTRIPE, rareCap = (uint64_t)a.rareSegCap * T1K_NSTRIPE, jobCap = (uint64_t)a.jobSegCap * T1K_NSTRIPE,
                 genJobCap = (uint64_t)a.genJobSegCap * T1K_NSTRIPE, groupCap = (uint64_t)a.groupSegCap * T1K_NSTRIPE;
  {  // closed-form pass over all records: blockIdx.y = stripe.  The straight-line kernel (the striding form costs it a wavefront per SIMD) over a
     // grid that covers what a stripe CAN hold: the workgroups beyond the stripe's cursor end at once (a few hundred thousand empty
     // workgroups are tens of microseconds of dispatch)
    const dim3 grid(blocks(a.groupSegCap, WG), T1K_NSTRIPE);
    (void)groupCap;
    if (longReads) hipLaunchKernelGGL((k_chain_fast<10, 0, false>), grid, dim3(WG), 0, ctx->stream, a, (const uint32_t *)nullptr, 0u, tot);
    else hipLaunchKernelGGL((k_chain_fast<5, 0, false>), grid, dim3(WG), 0, ctx->stream, a, (const uint32_t *)nullptr, 0u, tot);
  }
  {  // gap walk over the groups the closed form left
    const uint64_t e = est(T1K_AR_SLOW, listCap);
    t1k_arena_compact_dev(ctx, T1K_AR_SLOW, a.slowStr, a.listSegCap, a.slowList, e);
    if (longReads) hipLaunchKernelGGL((k_chain_fast<10, 1, true>), dim3(blocks(e, WG)), dim3(WG), 0, ctx->stream, a, (const uint32_t *)a.slowList, 0u, tot + T1K_AR_SLOW);
    else hipLaunchKernelGGL((k_chain_fast<5, 1, true>), dim3(blocks(e, WG)), dim3(WG), 0, ctx->stream, a, (const uint32_t *)a.slowList, 0u, tot + T1K_AR_SLOW);
  }
  const uint64_t eJobs = est(T1K_AR_JOBS, jobCap), eRetry = est(T1K_AR_RETRY, listCap), eGen = est(T1K_AR_GENERAL, rareCap);
  t1k_arena_compact_dev(ctx, T1K_AR_JOBS, a.jobStr, a.jobSegCap, a.jobList, eJobs);
  t1k_arena_compact_dev(ctx, T1K_AR_RETRY, a.retryStr, a.listSegCap, a.retryList, eRetry);
  t1k_arena_compact_dev(ctx, T1K_AR_GENERAL, a.generalStr, a.rareSegCap, a.generalList, eGen);
  // (the finish list -- groups whose candidate waits for registered alignments -- is only a count here: k_collect adds the match counts itself;
  // a full stripe of it loses nothing, its entries are never read)
  t1k_launch_dp_dense_dev(ctx, a, a.jobList, T1K_AR_JOBS, (uint32_t)jobCap, eJobs);
  if (longReads) hipLaunchKernelGGL((k_chain_fast<10, 2, true>), dim3(blocks(eRetry, WG)), dim3(WG), 0, ctx->stream, a, (const uint32_t *)a.retryList, 0u, tot + T1K_AR_RETRY);
  else hipLaunchKernelGGL((k_chain_fast<5, 2, true>), dim3(blocks(eRetry, WG)), dim3(WG), 0, ctx->stream, a, (const uint32_t *)a.retryList, 0u, tot + T1K_AR_RETRY);
  {  // groups with hits on several diagonals
    static const bool nearHits = getenv("T1K_NO_NEAR_HITS") == nullptr;
    const int skipDone = nearHits && !xlong ? 1 : 0;
    const unsigned long long *nGen = tot + T1K_AR_GENERAL;
    if (skipDone) {
      if (longReads) hipLaunchKernelGGL(k_near_hits<10>, dim3(blocks(eGen, WG)), dim3(WG), 0, ctx->stream, a, 0u, nGen);
      else hipLaunchKernelGGL(k_near_hits<5>, dim3(blocks(eGen, WG)), dim3(WG), 0, ctx->stream, a, 0u, nGen);
    }
    const uint32_t gatherGrid = std::min<uint32_t>(blocks(eGen, 4), 8192u);
    if (xlong) hipLaunchKernelGGL(k_gather_general<(T1K_LONG_READ_LEN + 63) / 64>, dim3(gatherGrid), dim3(WG), 0, ctx->stream, a, 0u, skipDone, nGen);
    else hipLaunchKernelGGL(k_gather_general<GROUP_FAST_MAXLEN / 64>, dim3(gatherGrid), dim3(WG), 0, ctx->stream, a, 0u, skipDone, nGen);
    ChainArgs g = a;
    g.generalList = (uint32_t *)countingSort(ctx, a.generalList, (uint32_t)rareCap, GroupSizeKey{(const uint32_t *)a.recs, a.recStride, skipDone}, nGen, eGen);  // groups of similar size side by side
    hipLaunchKernelGGL(k_chain_general, dim3(blocks(eGen, 64)), dim3(64), 0, ctx->stream, g, 0u, skipDone, nGen);
    const uint64_t eWave = est(T1K_AR_WAVE, rareCap);
    t1k_arena_compact_dev(ctx, T1K_AR_WAVE, a.waveStr, a.rareSegCap, a.waveList, eWave);
    T1K_HIP(ctx, hipFuncSetAttribute((const void *)k_chain_wave, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * WAVE_CAP * 4));
    {
      const uint32_t wgrid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(eWave, 2048u));
      static const uint32_t smallCap = getenv("T1K_WAVE_SMALL") ? (uint32_t)atoi(getenv("T1K_WAVE_SMALL")) : 512u;  // 0: one launch as in rounds 2-4
      if (smallCap) hipLaunchKernelGGL(k_chain_wave, dim3(wgrid), dim3(64), 3 * smallCap * 4, ctx->stream, a, 0u, tot + T1K_AR_WAVE, smallCap, 0u);
      hipLaunchKernelGGL(k_chain_wave, dim3(wgrid), dim3(64), 3 * WAVE_CAP * 4, ctx->stream, a, 0u, tot + T1K_AR_WAVE, (uint32_t)WAVE_CAP, smallCap);
    }
    const uint64_t eGenJobs = est(T1K_AR_GENJOBS, genJobCap);
    t1k_arena_compact_dev(ctx, T1K_AR_GENJOBS, a.genJobStr, a.genJobSegCap, a.genJobList, eGenJobs);
    t1k_launch_dp_dense_dev(ctx, a, a.genJobList, T1K_AR_GENJOBS, (uint32_t)genJobCap, eGenJobs);
    hipLaunchKernelGGL(k_general_finish, dim3(blocks(eGen, WG)), dim3(WG), 0, ctx->stream, a, 0u, nGen);
    t1k_arena_compact_dev(ctx, T1K_AR_BIG, a.bigStr, a.rareSegCap, a.bigList, est(T1K_AR_BIG, rareCap));
    hipLaunchKernelGGL(k_chain_big, dim3(bigBlocks * 64), dim3(64), 0, ctx->stream, a, 0u, tot + T1K_AR_BIG);
  }
  if (xlong) hipLaunchKernelGGL(k_collect<T1K_LONG_READ_LEN>, dim3(nWg), dim3(WG), 0, ctx->stream, a);
  else hipLaunchKernelGGL(k_collect<T1K_MAX_READ_LEN>, dim3(nWg), dim3(WG), 0, ctx->stream, a);
  T1K_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
  const int rc = readCounters(ctx, hc);
  if (rc) return rc;
  // what the cursors say: overflow of a list whose compaction is the only place that could have seen it, the statistics, the next range's estimates
  const T1kArenaCounts groups = t1k_arena_counts(ctx, T1K_AR_GROUPS, a.groupSegCap), slow = t1k_arena_counts(ctx, T1K_AR_SLOW, a.listSegCap), jobs = t1k_arena_counts(ctx, T1K_AR_JOBS, a.jobSegCap),
                       retry = t1k_arena_counts(ctx, T1K_AR_RETRY, a.listSegCap), fin = t1k_arena_counts(ctx, T1K_AR_FINISH, a.listSegCap), gen = t1k_arena_counts(ctx, T1K_AR_GENERAL, a.rareSegCap),
                       wv = t1k_arena_counts(ctx, T1K_AR_WAVE, a.rareSegCap), gj = t1k_arena_counts(ctx, T1K_AR_GENJOBS, a.genJobSegCap), big = t1k_arena_counts(ctx, T1K_AR_BIG, a.rareSegCap);
  if (groups.overflow || slow.overflow || jobs.overflow || retry.overflow || gen.overflow || wv.overflow || gj.overflow || big.overflow) hc[2] |= ERR_GROUPCAP;
  // (the finish list: word 22 is a statistic; the list itself is not read -- see above -- so its overflow is not an error)
  // A full job list means memo claims were released while other lanes already waited on them: what ran behind it may have met a slot still
  // pending (ERR_MEMO).  That is a consequence of the overflow, not an internal error: the range runs again with lists that fit.
  if (hc[2] & ERR_GROUPCAP) hc[2] &= ~(unsigned long long)ERR_MEMO;
  ctx->lastSlowGroups = slow.total;
  hc[16] = jobs.total; hc[17] = retry.total; hc[18] = gen.total; hc[19] = big.total; hc[22] = fin.total;
  if (!hc[2]) {
    const T1kArenaCounts *cs[] = {&groups, &slow, &jobs, &retry, &gen, &wv, &gj, &big};
    const int ar[] = {T1K_AR_GROUPS, T1K_AR_SLOW, T1K_AR_JOBS, T1K_AR_RETRY, T1K_AR_GENERAL, T1K_AR_WAVE, T1K_AR_GENJOBS, T1K_AR_BIG};
    for (int i = 0; i < 8; ++i) t1k_arena_estimate_set(ctx, ar[i], cs[i]->total, nRe);
    ctx->estValid = true;
  }
  return 0;
}

// runs K1..K6; on return counters[0] = number of candidates, counters[2] = error flags
int t1k_run_chain(t1k_ctx *ctx, const ChainArgs &a, int nWg, int bigBlocks, bool longReads, unsigned long long *hc) {
  const int AW = longReads ? 13 : 7;
  const size_t maxK = a.maxKFast;
  const bool xlong = a.maxK > a.maxKFast;  // the window holds read-ends beyond T1K_MAX_READ_LEN: k_seed_long takes those
  size_t lds = (size_t)CHUNK_A * AW * 4 + maxK * (5 * 4 + (T1K_SEED_PACK_Q ? 0 : 2)) + 4 + (T1K_SEED_PACK_Q ? 8 : 64);  // accumulators | sLo, pre, lstStart, lstLen, lstDir, qOf
  const size_t bitmapWords = 2 * (((size_t)a.ref.nAlleles + 31) / 32);        // chunk selection: two bitmaps over all alleles ...
  if (bitmapWords > (size_t)CHUNK_A * AW) lds += bitmapWords * 4 + 8;         // ... behind the list arrays when the accumulators cannot hold them
  if (a.ref.kDirStride - 1 > 256) return t1k_fail(ctx, T1K_ERR_ARG, "the reference holds more than 131 072 distinct sequences (256 seeding chunks)");
  if (lds > 160 * 1024) return t1k_fail(ctx, T1K_ERR_ARG, "the reference holds too many sequences for the seeding kernel's LDS bitmaps");
  // the seeding kernel keeps no per-workgroup HBM scratch: one workgroup per read-end (up to 32768) balances their uneven cost best
  const char *esw = getenv("T1K_SEED_WG");
  const int seedWg = (int)std::min<uint32_t>(a.reads.nReadEnds, esw ? (uint32_t)atoi(esw) : 32768u);
  T1K_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
  if (a.fuse) {
    if (longReads) {
      T1K_HIP(ctx, hipFuncSetAttribute((const void *)k_seed_chain<10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k_seed_chain<10>, dim3(seedWg), dim3(WG), lds, ctx->stream, a);
    } else {
      T1K_HIP(ctx, hipFuncSetAttribute((const void *)k_seed_chain<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k_seed_chain<5>, dim3(seedWg), dim3(WG), lds, ctx->stream, a);
    }
  } else if (longReads) {
    T1K_HIP(ctx, hipFuncSetAttribute((const void *)k_seed_groups<10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_seed_groups<10>, dim3(seedWg), dim3(WG), lds, ctx->stream, a);
  } else {
    T1K_HIP(ctx, hipFuncSetAttribute((const void *)k_seed_groups<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_seed_groups<5>, dim3(seedWg), dim3(WG), lds, ctx->stream, a);
  }
  if (xlong) {
    const size_t ldsLong = (size_t)a.maxK * (4 * 4 + 2) + 8 + (size_t)LONG_CH * 4;
    T1K_HIP(ctx, hipFuncSetAttribute((const void *)k_seed_long, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsLong));
    hipLaunchKernelGGL(k_seed_long, dim3(std::min<uint32_t>(a.reads.nReadEnds, 32768u)), dim3(WG), ldsLong, ctx->stream, a);
  }
  T1K_HIP(ctx, hipEventRecord(ctx->ev[8], ctx->stream));
  // T1K_HOST_CHAIN: the launch sequence of rounds 1-4 -- the host fetches the counters between a producer and its consumers (A/B, fallback)
  if (!t1k_chain_host_driven() && !a.fuse) return runChainDevice(ctx, a, nWg, bigBlocks, longReads, xlong, hc);
  const unsigned long long *NODEV = nullptr;
  int rc = readCounters(ctx, hc);
  if (rc) return rc;
  if (hc[2]) return 0;
  const T1kArenaCounts groups = t1k_arena_counts(ctx, T1K_AR_GROUPS, a.groupSegCap);
  const unsigned long long groupsSeeded = hc[6];  // records + (fused seeding) the groups that ended without one
  if (!a.fuse) {  // the closed-form pass as a launch of its own over all records (T1K_FUSE_SEED=0)
    if (groups.maxSeg) {
      const dim3 grid((groups.maxSeg + WG - 1) / WG, T1K_NSTRIPE);
      if (longReads) hipLaunchKernelGGL((k_chain_fast<10, 0, false>), grid, dim3(WG), 0, ctx->stream, a, (const uint32_t *)nullptr, 0u, NODEV);
      else hipLaunchKernelGGL((k_chain_fast<5, 0, false>), grid, dim3(WG), 0, ctx->stream, a, (const uint32_t *)nullptr, 0u, NODEV);
    }
    if ((rc = readCounters(ctx, hc))) return rc;
  }
  const T1kArenaCounts slow = t1k_arena_counts(ctx, T1K_AR_SLOW, a.listSegCap);
  if (slow.overflow) { hc[2] |= ERR_GROUPCAP; return 0; }
  ctx->lastSlowGroups = slow.total;
  if (slow.total) {
    t1k_arena_compact(ctx, T1K_AR_SLOW, a.slowStr, a.listSegCap, a.slowList, slow.maxSeg);
    const uint32_t nSlowGroups = (uint32_t)slow.total;
    if (longReads) hipLaunchKernelGGL((k_chain_fast<10, 1, false>), dim3((nSlowGroups + WG - 1) / WG), dim3(WG), 0, ctx->stream, a, (const uint32_t *)a.slowList, nSlowGroups, NODEV);
    else hipLaunchKernelGGL((k_chain_fast<5, 1, false>), dim3((nSlowGroups + WG - 1) / WG), dim3(WG), 0, ctx->stream, a, (const uint32_t *)a.slowList, nSlowGroups, NODEV);
  }
  if ((rc = readCounters(ctx, hc))) return rc;
  hc[6] = groupsSeeded;
  const T1kArenaCounts jobs = t1k_arena_counts(ctx, T1K_AR_JOBS, a.jobSegCap), retry = t1k_arena_counts(ctx, T1K_AR_RETRY, a.listSegCap),
                       fin = t1k_arena_counts(ctx, T1K_AR_FINISH, a.listSegCap), gen = t1k_arena_counts(ctx, T1K_AR_GENERAL, a.rareSegCap);
  // A full job segment is NOT benign: the lane that could not list its job released the memo claim, but another lane may already be
  // waiting on that slot (it saw the claim) and would find it empty or re-claimed at finish time.  The range runs again with a larger list.
  if (retry.overflow || fin.overflow || gen.overflow || jobs.overflow) { hc[2] |= ERR_GROUPCAP; return 0; }
  hc[16] = jobs.total; hc[17] = retry.total; hc[18] = gen.total; hc[22] = fin.total;
  t1k_arena_compact(ctx, T1K_AR_JOBS, a.jobStr, a.jobSegCap, a.jobList, jobs.maxSeg);
  t1k_arena_compact(ctx, T1K_AR_FINISH, a.finishStr, a.listSegCap, a.finishList, fin.maxSeg);
  t1k_arena_compact(ctx, T1K_AR_RETRY, a.retryStr, a.listSegCap, a.retryList, retry.maxSeg);
  t1k_arena_compact(ctx, T1K_AR_GENERAL, a.generalStr, a.rareSegCap, a.generalList, gen.maxSeg);
  const uint32_t nJobs = (uint32_t)jobs.total, nRetry = (uint32_t)retry.total, nGen = (uint32_t)gen.total, nFinish = (uint32_t)fin.total;
  t1k_launch_dp_dense(ctx, a, a.jobList, nJobs);
  // (k_collect adds the registered alignments' match counts itself; T1K_FINISH_KERNEL=1 runs the separate pass of rounds 1-3 first)
  static const bool finishKernel = getenv("T1K_FINISH_KERNEL") != nullptr;
  if (nFinish && finishKernel) hipLaunchKernelGGL(k_chain_finish, dim3((nFinish + WG - 1) / WG), dim3(WG), 0, ctx->stream, a, nFinish);
  if (nRetry) {
    if (longReads) hipLaunchKernelGGL((k_chain_fast<10, 2, false>), dim3((nRetry + WG - 1) / WG), dim3(WG), 0, ctx->stream, a, (const uint32_t *)a.retryList, nRetry, NODEV);
    else hipLaunchKernelGGL((k_chain_fast<5, 2, false>), dim3((nRetry + WG - 1) / WG), dim3(WG), 0, ctx->stream, a, (const uint32_t *)a.retryList, nRetry, NODEV);
  }
  uint32_t nBig = 0;
  if (nGen) {
    static const bool nearHits = getenv("T1K_NO_NEAR_HITS") == nullptr;
    const int skipDone = nearHits && !xlong ? 1 : 0;
    if (skipDone) {
      if (longReads) hipLaunchKernelGGL(k_near_hits<10>, dim3((nGen + WG - 1) / WG), dim3(WG), 0, ctx->stream, a, nGen, NODEV);
      else hipLaunchKernelGGL(k_near_hits<5>, dim3((nGen + WG - 1) / WG), dim3(WG), 0, ctx->stream, a, nGen, NODEV);
    }
    if (xlong) hipLaunchKernelGGL(k_gather_general<(T1K_LONG_READ_LEN + 63) / 64>, dim3(std::min<uint32_t>((nGen + 3) / 4, 8192u)), dim3(WG), 0, ctx->stream, a, nGen, skipDone, NODEV);
    else hipLaunchKernelGGL(k_gather_general<GROUP_FAST_MAXLEN / 64>, dim3(std::min<uint32_t>((nGen + 3) / 4, 8192u)), dim3(WG), 0, ctx->stream, a, nGen, skipDone, NODEV);
    ChainArgs g = a;
    static const bool radix = getenv("T1K_RADIX_SORTS") != nullptr;
    if (nGen >= 4096 && !radix)  // groups of similar size side by side
      g.generalList = (uint32_t *)countingSort(ctx, a.generalList, nGen, GroupSizeKey{(const uint32_t *)a.recs, a.recStride, skipDone});
    else if (nGen >= 4096 && t1k_ensure(ctx, ctx->bSlowKeys, (size_t)nGen * 20 + 64) == T1K_OK) {
      unsigned long long *k0 = (unsigned long long *)ctx->bSlowKeys.p, *k1 = k0 + nGen;
      uint32_t *sorted = (uint32_t *)(k1 + nGen);
      hipLaunchKernelGGL(k_group_size_keys, dim3((nGen + WG - 1) / WG), dim3(WG), 0, ctx->stream, (const uint32_t *)a.recs, a.recStride, (const uint32_t *)a.generalList, k0, nGen, skipDone);
      if (t1k_sort_pairs(ctx, k0, k1, a.generalList, sorted, nGen, 8) == T1K_OK) g.generalList = sorted;
    }
    hipLaunchKernelGGL(k_chain_general, dim3((nGen + 63) / 64), dim3(64), 0, ctx->stream, g, nGen, skipDone, NODEV);
    if ((rc = readCounters(ctx, hc))) return rc;
    const T1kArenaCounts wv = t1k_arena_counts(ctx, T1K_AR_WAVE, a.rareSegCap);
    if (wv.overflow) { hc[2] |= ERR_GROUPCAP; return 0; }
    if (wv.total) {
      t1k_arena_compact(ctx, T1K_AR_WAVE, a.waveStr, a.rareSegCap, a.waveList, wv.maxSeg);
      T1K_HIP(ctx, hipFuncSetAttribute((const void *)k_chain_wave, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * WAVE_CAP * 4));
      hipLaunchKernelGGL(k_chain_wave, dim3(std::min<uint32_t>((uint32_t)wv.total, 2048u)), dim3(64), 3 * WAVE_CAP * 4, ctx->stream, a, (uint32_t)wv.total, NODEV, (uint32_t)WAVE_CAP, 0u);
    }
    if ((rc = readCounters(ctx, hc))) return rc;  // the general kernels register alignments and may hand groups over to the big-scratch kernel
    const T1kArenaCounts gj = t1k_arena_counts(ctx, T1K_AR_GENJOBS, a.genJobSegCap);
    if (gj.overflow) { hc[2] |= ERR_GROUPCAP; return 0; }
    t1k_arena_compact(ctx, T1K_AR_GENJOBS, a.genJobStr, a.genJobSegCap, a.genJobList, gj.maxSeg);
    t1k_launch_dp_dense(ctx, a, a.genJobList, (uint32_t)gj.total);
    hipLaunchKernelGGL(k_general_finish, dim3((nGen + WG - 1) / WG), dim3(WG), 0, ctx->stream, a, nGen, NODEV);
    const T1kArenaCounts big = t1k_arena_counts(ctx, T1K_AR_BIG, a.rareSegCap);
    if (big.overflow) { hc[2] |= ERR_GROUPCAP; return 0; }
    t1k_arena_compact(ctx, T1K_AR_BIG, a.bigStr, a.rareSegCap, a.bigList, big.maxSeg);
    nBig = (uint32_t)big.total;
  }
  if (nBig) hipLaunchKernelGGL(k_chain_big, dim3(bigBlocks * 64), dim3(64), 0, ctx->stream, a, nBig, NODEV);
  if (xlong) hipLaunchKernelGGL(k_collect<T1K_LONG_READ_LEN>, dim3(nWg), dim3(WG), 0, ctx->stream, a);
  else hipLaunchKernelGGL(k_collect<T1K_MAX_READ_LEN>, dim3(nWg), dim3(WG), 0, ctx->stream, a);
  T1K_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
  rc = readCounters(ctx, hc);
  hc[6] = groupsSeeded; hc[16] = jobs.total; hc[17] = retry.total; hc[18] = gen.total; hc[19] = nBig; hc[22] = fin.total;
  return rc;
}
