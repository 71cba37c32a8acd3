// t1k_amd/csrc/t1k_memo.h -- exact per-read-end memo of gap alignments, shared by the chain stage (t1k_chain.hip) and
// the extension stage (k_extend, t1k_assign.hip)
#pragma once
#include "t1k_dev.h"

struct ReadCtx {
  const uint64_t *rb, *rn;   // strand-specific read words
  int len;
  const uint64_t *gb, *gn;   // reference words
  int64_t goff;              // allele global base offset
  int alleleLen;
  bool refN;                 // the reference holds an N somewhere (T1kRefDev::anyN): otherwise its N-mask words are all zero and not loaded here
  const uint64_t *gT = nullptr;  // this allele's column of the transposed copy of the bases (T1kRefDev::basesT): word w of the allele at gT[w * 64]; NULL = linear only
};


// ------------------------------------------------------------------------------------------------------------------
// Exact memo of gap alignments within one read-end.  Thousands of alleles of a gene carry the same bases under a given
// read window, so the same banded DP would be recomputed for each of them.  One 64-bit entry identifies a job completely:
//   [gpos:30 | matches:9 | readPos:11 | readLen:9 | (alleleLen - readLen + 4):4 | strand:1]
// A probe whose (strand, readPos, lengths) agree verifies that the allele window at the entry's gpos holds exactly the same
// bases and N-mask as its own window before it reuses the stored match count, so a hit is bit-exact by construction
// (GlobalAlignment only sees the two windows).  The table lives in HBM (16 KB per read-end) and is cleared per batch.
// ------------------------------------------------------------------------------------------------------------------
#define GAP_CACHE 2048
#define GAP_PROBES 4
#define GAP_ID_MASK 0x1FFFFFFull
#define GAP_VAL(e) ((uint32_t)(((e) >> 25) & 0x1FF))
#define GAP_GPOS(e) ((int64_t)((e) >> 34))
#define GAP_PENDING 0x1FFull
struct GapSink {  // where deferred alignments are registered
  unsigned long long *cache;     // the read-end's memo table
  uint32_t *jobStr;              // striped job list
  unsigned long long *counters;
  uint32_t jobTag, jobSegCap;
  int arena;
};

// the alignment itself: lp read positions from readPos against lt allele positions from gpos, |lt - lp| <= 4
__device__ __forceinline__ int gapAlign(const ReadCtx &c, int readPos, int64_t gpos, int lp, int lt) {
  T1kSeqView T{c.gb, c.gn, gpos}, P{c.rb, c.rn, (int64_t)readPos};
  if (lp == lt) return t1k_ga_matches_equal(T, P, lp, nullptr);
  return t1k_ga_band<4, false>(T, lt, P, lp, nullptr, 0);
}

// DEFER = true : never run a DP here.  A miss claims a memo slot (CAS) with the PENDING marker and appends the slot to the
//                job list; the caller keeps the slot (return -1) and adds the match count once the dense DP phase has
//                filled the memo.  -2: not memoisable / table or list full, the caller aligns it inline later.
// DEFER = false: a miss is computed inline.
template <bool DEFER>
__device__ inline int gapMatchesCached(const ReadCtx &c, int readPos, int64_t gpos, int lp, int lt, int strandBit, const GapSink &sink, unsigned int *dpCounter,
                                       uint32_t *slotOut) {
  if (lp <= 0 || lt <= 0) return 0;
  const int d = lt - lp;
  // content hash of the allele window (and, for equal lengths, the mismatch count) in one sweep
  int x = 0;
  uint64_t hsh = 0x9E3779B97F4A7C15ull ^ ((uint64_t)readPos << 20) ^ ((uint64_t)lp << 1) ^ ((uint64_t)(d + 4) << 40) ^ (uint64_t)strandBit;
  for (int o = 0; o < lt; o += 32) {
    uint64_t lm = t1k_lowmask(lt - o);
    uint64_t gw = t1k_get32(c.gb, gpos + o) & lm, gnw = c.refN ? t1k_get32(c.gn, gpos + o) & lm : 0;
    if (d == 0) {
      uint64_t xo = t1k_get32(c.rb, readPos + o) ^ gw;
      uint64_t mm = (xo | (xo >> 1)) & T1K_EVEN & ~(t1k_get32(c.rn, readPos + o) | gnw) & lm;
      x += __popcll(mm);
    }
    hsh = (hsh ^ gw ^ (gnw << 1)) * 0xD6E8FEB86659FD93ull;
    hsh ^= hsh >> 32;
  }
  if (d == 0 && x <= 3) return lp - x;  // exact fast path (see t1k_ga_matches_window)
  if (lp > 510 || lt > 510 || readPos > 2047 || gpos >= (1ll << 30)) {
    if (DEFER) return -2;  // not memoisable
    if (dpCounter) ++*dpCounter;
    return gapAlign(c, readPos, gpos, lp, lt);
  }
  const uint64_t idBits = ((uint64_t)readPos << 14) | ((uint64_t)lp << 5) | ((uint64_t)(d + 4) << 1) | (uint64_t)strandBit;  // low 25 bits of an entry
  unsigned long long *cache = sink.cache;
  const uint32_t slot = (uint32_t)hsh & (GAP_CACHE - 1);
  bool pendingSeen = false;
  unsigned long long probed[GAP_PROBES];  // the probe slots share one 32-byte block: fetch them together, then look
#pragma unroll
  for (int probe = 0; probe < GAP_PROBES; ++probe) probed[probe] = __hip_atomic_load(&cache[slot ^ probe], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int probe = 0; probe < GAP_PROBES; ++probe) {
    const unsigned long long e = probed[probe];
    if (e != 0 && (e & GAP_ID_MASK) == idBits) {
      int64_t eg = GAP_GPOS(e);
      if (eg == gpos || t1k_same_window(c.gb, c.gn, eg, gpos, lt, c.refN)) {
        const uint32_t v = GAP_VAL(e);
        if (v != GAP_PENDING) return (int)v;
        pendingSeen = true;
        *slotOut = slot ^ probe;
      }
    }
  }
  if (DEFER) {
    if (pendingSeen) return -1;
    const unsigned long long pe = ((unsigned long long)gpos << 34) | (GAP_PENDING << 25) | idBits;
#pragma unroll
    for (int probe = 0; probe < GAP_PROBES; ++probe) {
      unsigned long long old = atomicCAS(&cache[slot ^ probe], 0ull, pe);
      if (old == 0ull) {
        const uint32_t q = t1k_arena_append(sink.counters, sink.arena, sink.jobSegCap);
        if (q == T1K_ARENA_FULL) {  // job list full: release the claim
          __hip_atomic_store(&cache[slot ^ probe], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          return -2;
        }
        sink.jobStr[q] = sink.jobTag + (slot ^ probe);
        *slotOut = slot ^ probe;
        return -1;
      }
      if ((old & GAP_ID_MASK) == idBits && (GAP_GPOS(old) == gpos || t1k_same_window(c.gb, c.gn, GAP_GPOS(old), gpos, lt, c.refN))) {
        *slotOut = slot ^ probe;
        return -1;  // somebody else just claimed it
      }
    }
    return -2;  // all probe slots taken by other jobs
  }
  if (dpCounter) ++*dpCounter;
  const int m = gapAlign(c, readPos, gpos, lp, lt);
  if (!pendingSeen) {
    unsigned long long ne = ((unsigned long long)gpos << 34) | ((unsigned long long)m << 25) | idBits;
    unsigned long long cur = __hip_atomic_load(&cache[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == 0) __hip_atomic_store(&cache[slot], ne, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return m;
}

