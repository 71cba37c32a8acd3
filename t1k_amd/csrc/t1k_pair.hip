// t1k_amd/csrc/t1k_pair.hip -- mate pairing and fragment rows on the GPU.
//   k_pair: one 256-thread workgroup per fragment restates SeqSet::ReadAssignmentToFragmentAssignment
//           (SeqSet.hpp:2310-2655) and Genotyper::SetReadAssignments / ReadAssignmentWeight (Genotyper.hpp:778-832,
//           205-230) on the two overlap lists left in HBM by t1k_assign_batch.
// The two overlap lists are found through the read set's list table (T1kReadsDev::listPtr / listCount): identical read-ends are
// assigned once (t1k_reads_dedupe), so the mates of a fragment may have been assigned in different batches, by different contexts.
// Two output forms: t1k_pair_batch keeps the rows of one call in the context (fragment order, list order inside a row);
// t1k_pair_into appends them to a t1k_rowset -- all rows of a job resident in HBM, each row ordered by allele index with a
// 128-bit pattern hash per fragment -- which t1k_rowset_coalesce (t1k_coalesce.hip) turns into read groups.
// Mates are joined through a per-workgroup direct-address table (allele -> index in the mate's list, stamped with a
// fragment epoch so it never needs clearing).  If an allele occurs twice in a list (several diagonal runs on one
// allele -- rare) lane 0 replays the reference's sequential algorithm instead.
#include <algorithm>
#include <chrono>
#include <cstring>
#include "t1k_dev.h"
#include "t1k_launch.h"

#ifdef T1K_PAIR_PROFILE
#define PP(i) do { if (tid == 0) { const uint64_t tn_ = __builtin_amdgcn_s_memtime(); tp_[i] += tn_ - tl_; tl_ = tn_; } } while (0)
#else
#define PP(i) do { } while (0)
#endif
#define SORT_TILE 2048
// Join table of one fragment in LDS: allele -> (index in list 1 + 1) | membership bit 15 | (index in list 2 + 1) << 16.  Open addressing,
// LJ_SLOTS slots (4096 or 2048: a template parameter of the kernel); fragments whose two lists hold more than 68 % of that (or a list
// with a repeated allele) use the per-workgroup direct-address tables in HBM instead.
template <int LJ_SLOTS>
__device__ __forceinline__ uint32_t ljHash(uint32_t allele) { return (allele * 2654435761u) >> (LJ_SLOTS == 4096 ? 20 : 21); }  // 12 / 11 bits
template <int LJ_SLOTS>
__device__ __forceinline__ uint32_t ljInsert(uint32_t *hKey, uint32_t allele) {
  const uint32_t key = allele + 1;
  uint32_t h = ljHash<LJ_SLOTS>(allele);
  for (;;) {
    const uint32_t k = hKey[h];
    if (k == key) return h;
    if (k == 0) {
      const uint32_t old = atomicCAS(&hKey[h], 0u, key);
      if (old == 0 || old == key) return h;
    }
    h = (h + 1) & (LJ_SLOTS - 1);
  }
}
template <int LJ_SLOTS>
__device__ __forceinline__ int ljFind(const uint32_t *hKey, uint32_t allele) {  // slot of an allele, -1 if absent
  const uint32_t key = allele + 1;
  uint32_t h = ljHash<LJ_SLOTS>(allele);
  for (;;) {
    const uint32_t k = hKey[h];
    if (k == key) return (int)h;
    if (k == 0) return -1;
    h = (h + 1) & (LJ_SLOTS - 1);
  }
}

// contribution of the allele at sorted position r to a fragment's pattern hash (summed over the row: order of evaluation is free)
__device__ __forceinline__ unsigned long long t1k_pattern_mix(uint32_t allele, uint32_t r, unsigned long long k) {
  unsigned long long x = ((unsigned long long)allele << 32 | r) * k;
  x ^= x >> 31; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 29;
  return x;
}

#ifndef T1K_PAIR_UNROLL
#define T1K_PAIR_UNROLL 2
#endif
#ifndef T1K_PAIR_HANDOUT
#define T1K_PAIR_HANDOUT 4   // fragments a workgroup takes per hand-out atomic (1 / 2 / 4 / 8 alone: 1.296 / 1.299 / 1.320 / 1.43 ms; under three pipelines 4 + the row reservation: -0.6 % of the step)
#endif
#ifndef T1K_ROW_RESERVE
#define T1K_ROW_RESERVE 32   // row entries a workgroup reserves per atomic on the rowset's cursor (0: every row its own)
#endif
#ifndef T1K_PAIR_WAVES
#define T1K_PAIR_WAVES 4   // wavefronts per SIMD the register allocation of k_pair is held to (the join table's LDS admits four workgroups per compute unit)
#endif
struct Frag {
  uint32_t allele;
  int32_t i, j;            // index in list 1 / list 2 (-1 = none)
  int32_t matchCnt, relaxed, seqStart, seqEnd;
  int32_t slot;            // slow path: position in `assign`
  double sim;
};

struct PairArgs {
  T1kRefDev ref;
  const unsigned long long *listPtr;  // per read-end of the read set: address of its first overlap record
  const uint32_t *listCount;
  const uint32_t *end1, *end2;     // end2 == NULL: single-end run
  const uint8_t *hasN;
  uint32_t nFragments;
  double sim; int relax; int maxAssign; int hitLenRequired;
  t1k_row_entry *rows; uint64_t rowCap;
  uint32_t *rowStart, *rowCount; uint8_t *fragAssigned;
  uint64_t *tab2, *tabSlot;        // [wg][nAlleles]  (epoch << 32 | value)
  uint32_t epochBase;              // epoch of fragment f = epochBase + f + 1 (never 0, never reused between clears)
  Frag *frags; uint32_t fragCap;   // [wg][fragCap]
  uint32_t *keep;                  // [wg][fragCap]
  unsigned long long *counters;    // [2] error flags, [9] row total
  // second pass (same launch sequence, no host step between the two): the fragments whose two lists did not fit the first pass's per-workgroup
  // scratch.  `only` = {fragment, offset of its scratch in the big arena} pairs written by the first pass, their number in counters[23]
  // (read by the device: the launch is unconditional and sized to fill the chip), the arena cursor in counters[24] (entries, counted past
  // the capacity so that the host knows what to allocate when it has to run the call again)
  const uint32_t *only;
  Frag *bigFrags; uint32_t *bigKeep; uint64_t bigCap;
  uint32_t *overflowList;          // first pass: fragments with more than fragCap overlaps are listed here instead of failing
  const uint8_t *whitelist;        // [nAlleles] or NULL: alleles outside it are left out of the rows (Genotyper.hpp:822-823)
  int listForm;                    // T1K_PAIR_LIST=1: the joined fragments are materialised as in round 3 (A/B and parity aid)
  int rawKept;                     // rows = the fragment assignment list itself (what ReadAssignmentToFragmentAssignment returns), without
                                   // the -n / separator / whitelist drops of SetReadAssignments: the analyzer's per-barcode summary reads that
  // rowset form (rsRowPtr != NULL): rows go to rsRows[*rsCursor ...), ordered by allele; per-fragment records at fragBase + f
  unsigned long long *rsRowPtr; uint32_t *rsRowCount; unsigned long long *rsH1, *rsH2; uint8_t *rsAssigned;
  t1k_row_entry *rsRows; uint64_t rsCap; unsigned long long *rsCursor; uint64_t fragBase;
};

// a read-end's list in the overlap store: packed records, unpacked on access
struct OvlList {
  const T1kOvlP *p;
  __device__ __forceinline__ T1kOvl operator[](uint32_t i) const { return t1k_ovl_unpack(p[i]); }
};

__device__ __forceinline__ double ovlSim(const T1kOvl &o) {
  return (double)o.matchCnt / (double)(o.readEnd - o.readStart + 1 + o.seqEnd - o.seqStart + 1 + 2 * o.leftClip + 2 * o.rightClip);
}
__device__ __forceinline__ int ovlStrand(const T1kOvl &o) { return (o.flags & 2) ? -1 : 1; }

__device__ __forceinline__ bool sepInRangeP(const T1kRefDev &ref, uint32_t allele, int s, int e) {  // SeqSet.hpp:487-498
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

__device__ inline void makeFrag(Frag &f, const T1kOvl *o1, int i, const T1kOvl *o2, int j) {
  // SeqSet.hpp:2391-2440
  const T1kOvl &o = o1 ? *o1 : *o2;
  f.allele = o.allele; f.i = o1 ? i : -1; f.j = o2 ? j : -1; f.slot = -1;
  f.matchCnt = o.matchCnt; f.relaxed = o.relaxed; f.seqStart = o.seqStart; f.seqEnd = o.seqEnd; f.sim = ovlSim(o);
  if (o1 && o2) {
    f.matchCnt += o2->matchCnt;
    f.relaxed += o2->relaxed;
    if (ovlStrand(o) == 1) f.seqEnd = o2->seqEnd; else f.seqStart = o2->seqStart;
    f.sim = (double)f.matchCnt / (double)(o.readEnd - o.readStart + 1 + o2->readEnd - o2->readStart + 1 + o.seqEnd - o.seqStart + 1 + o2->seqEnd -
                                          o2->seqStart + 1 + 2 * o.leftClip + 2 * o.rightClip + 2 * o2->leftClip + 2 * o2->rightClip);
  }
}

// _fragmentOverlap::operator< (SeqSet.hpp:164-171) then _overlap::operator< on overlap1 (103-127)
__device__ inline bool fragBefore(const Frag &a, const T1kOvl &a1, const Frag &b, const T1kOvl &b1) {
  if (a.matchCnt != b.matchCnt) return a.matchCnt > b.matchCnt;
  if (a.sim != b.sim) return a.sim > b.sim;
  if (a1.matchCnt != b1.matchCnt) return a1.matchCnt > b1.matchCnt;
  double sa = ovlSim(a1), sb = ovlSim(b1);
  if (sa != sb) return sa > sb;
  if (a1.readEnd - a1.readStart != b1.readEnd - b1.readStart) return a1.readEnd - a1.readStart > b1.readEnd - b1.readStart;
  if (a1.allele != b1.allele) return a1.allele < b1.allele;
  if (ovlStrand(a1) != ovlStrand(b1)) return ovlStrand(a1) < ovlStrand(b1);
  if (a1.readStart != b1.readStart) return a1.readStart < b1.readStart;
  if (a1.readEnd != b1.readEnd) return a1.readEnd < b1.readEnd;
  if (a1.seqStart != b1.seqStart) return a1.seqStart < b1.seqStart;
  return a1.seqEnd < b1.seqEnd;
}

__device__ inline bool truncatedMate(const T1kRefDev &ref, const T1kOvl &o, const T1kOvl &c1, const T1kOvl &c2) {  // SeqSet.hpp:502-523
  if (ovlStrand(o) == 1) {
    if ((int)ref.alleleLen[o.allele] - 1 < o.seqEnd + c2.seqEnd - c1.seqEnd || sepInRangeP(ref, o.allele, o.seqEnd, o.seqEnd + c2.seqEnd - c1.seqEnd + 1))
      return true;
  } else {
    if (o.seqStart - (c1.seqStart - c2.seqStart) < 0 || sepInRangeP(ref, o.allele, o.seqStart - (c1.seqStart - c2.seqStart) - 1, o.seqStart)) return true;
  }
  return false;
}

__device__ __forceinline__ float rowWeight(double sim, double s, bool hasN) {  // Genotyper::ReadAssignmentWeight (205-230)
  double ret = 1;
  double segment = (1 - s) / 4.0;
  if (segment < 0.01) segment = 0.01;
  if (sim < 1 - 3 * segment) ret = 0.01;
  else if (sim < 1 - 2 * segment) ret = 0.1;
  else if (sim < 1 - segment) ret = 0.5;
  if (hasN) ret /= 10.0;
  return (float)ret;
}

template <int NWAVE>
__device__ __forceinline__ uint32_t scanExcl(uint32_t v, uint32_t *warpSums, uint32_t *total) {
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (NWAVE == 1) {  // one wavefront per fragment: no workgroup barrier anywhere in the scan
    *total = __shfl(x, 63, 64);
    return x - v;
  }
  if (lane == 63) warpSums[wave] = x;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NWAVE; ++w) { if (w < wave) base += warpSums[w]; tot += warpSums[w]; }
  __syncthreads();
  *total = tot;
  return base + x - v;
}

// WG threads per fragment (256: four wavefronts meeting at ~20 barriers per fragment; 64: one wavefront, whose barriers cost nothing)
template <int WG, int LJ_SLOTS>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(T1K_PAIR_WAVES, T1K_PAIR_WAVES))) void k_pair(PairArgs P) {
  constexpr int NWAVE = WG / 64;
  constexpr int PU = T1K_PAIR_UNROLL;  // records of a list a lane requests before it looks at the first (the streamed passes)
  constexpr uint32_t LJ_CAP = LJ_SLOTS * 17 / 25;
  __shared__ uint32_t warpSums[NWAVE];
  __shared__ int sDup, sFail, sBestM, sBestIdx, sAnySep, sNotOne;
  __shared__ double sBestSim;
  __shared__ uint32_t sN, sBase;
  __shared__ uint32_t hKey[LJ_SLOTS], hVal[LJ_SLOTS];
  const int tid = threadIdx.x;
  const uint32_t A = P.ref.nAlleles;
  uint64_t *tab2 = P.tab2 + (uint64_t)blockIdx.x * A;
  uint64_t *tabSlot = P.tabSlot + (uint64_t)blockIdx.x * A;
  Frag *frags = P.frags + (uint64_t)blockIdx.x * P.fragCap;
  uint32_t *keep = P.keep + (uint64_t)blockIdx.x * P.fragCap;
  uint32_t fragCap = P.fragCap;
#ifdef T1K_PAIR_PROFILE
  uint64_t tp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_ = __builtin_amdgcn_s_memtime();
#endif
  const uint32_t nItems = P.only ? (uint32_t)P.counters[23] : P.nFragments;
  // fragments are handed out one at a time (a device counter per launch): their cost spans two orders of magnitude (0 .. 8192 and more
  // overlaps), and with a fixed stride the kernel lasted as long as its unluckiest workgroup
  // Round 6: a fragment cost FOUR atomics on three hot words of the counter block -- the hand-out and the row cursor (both returning), the
  // "overlap records read" and "row entries" statistics -- 260 k atomics per launch of 65 536 fragments into two cache lines, on a part whose
  // returning atomics on one word saturate near 88 M/s ("striped allocation"): at 1.4 ms a launch the kernel sat at half of that rate on
  // each word.  The hand-out now takes T1K_PAIR_HANDOUT fragments at a time (their cost is uneven: no more than a few), the two statistics are
  // summed per workgroup (thread 0's registers) and flushed once.
  __shared__ uint32_t sItem;
  uint32_t hoNext = 0, hoLeft = 0;                         // thread 0: the fragments it holds from its last hand-out
  unsigned long long statOvl = 0, statRows = 0;            // thread 0
  unsigned long long rowNext = 0; uint32_t rowLeft = 0;    // thread 0: what is left of its last reservation in the rowset's chunk
  for (;;) {
    if (tid == 0) {
      if (hoLeft == 0) { hoNext = (uint32_t)atomicAdd(&P.counters[P.only ? 26 : 25], (unsigned long long)T1K_PAIR_HANDOUT); hoLeft = T1K_PAIR_HANDOUT; }
      sItem = hoNext++; --hoLeft;
      sDup = 0; sFail = 0; sBestM = -1; sBestIdx = 0x7FFFFFFF; sAnySep = 0; sNotOne = 0; sN = 0;
    }
    __syncthreads();
    const uint32_t it = sItem;
    if (it >= nItems) break;
    const uint32_t f = P.only ? P.only[2 * it] : it;
    const uint64_t epoch = (uint64_t)(P.epochBase + f + 1) << 32;
    const bool paired = P.end2 != nullptr;
    // (the two read-end numbers and the flag are requested together, then the four list words: two round trips, not four)
    const uint32_t e1 = P.end1[f];
    const uint32_t e2 = paired ? P.end2[f] : 0u;
    const bool hasN = P.hasN ? P.hasN[f] != 0 : false;
    const uint32_t n1 = P.listCount[e1];
    const OvlList L1{(const T1kOvlP *)P.listPtr[e1]};
    uint32_t n2 = 0;
    OvlList L2{nullptr};
    if (paired) { n2 = P.listCount[e2]; L2.p = (const T1kOvlP *)P.listPtr[e2]; }
    const bool dangling = paired && (n1 == 0 || n2 == 0);
    const bool both = paired && !dangling;
    uint32_t nFrag = 0;
    if (P.only) {  // this fragment's piece of the big arena
      const uint64_t off = P.only[2 * it + 1];
      fragCap = off + n1 + n2 <= P.bigCap ? n1 + n2 : 0u;
      frags = P.bigFrags + off; keep = P.bigKeep + off;
    }
    if (n1 + n2 > fragCap) {
      if (P.overflowList) {  // the second launch takes it, with a scratch of its own size
        if (tid == 0) {
          const unsigned long long q = atomicAdd(&P.counters[23], 1ull);
          const unsigned long long off = atomicAdd(&P.counters[24], (unsigned long long)(n1 + n2));
          P.overflowList[2 * q] = f; P.overflowList[2 * q + 1] = (uint32_t)min(off, 0xFFFFFFFFull);
        }
        __syncthreads();
        continue;
      }
      if (tid == 0) {
        atomicOr(&P.counters[2], P.only ? 1024ull : 128ull);  // 1024: the big arena is too small (the host grows it to counters[24] and runs the call again)
        if (P.rsRowPtr) { P.rsRowCount[P.fragBase + f] = 0; P.rsAssigned[P.fragBase + f] = 0; }
        else { P.rowStart[f] = 0; P.rowCount[f] = 0; P.fragAssigned[f] = 0; }
      }
      __syncthreads();
      continue;
    }
    if (tid == 0) statOvl += n1 + n2;  // statistics: overlap records read (by the launch that pairs the fragment)
    // ---- duplicate detection + join table --------------------------------------------------------------------------
    bool lds = n1 + n2 <= LJ_CAP;
    if (lds) {
      for (uint32_t q = tid; q < LJ_SLOTS; q += WG) { hKey[q] = 0; hVal[q] = 0; }
      __syncthreads();
      // (this is the first touch of the two lists: four records' allele words are requested together before the dependent LDS inserts --
      // one after the other, each insert's atomic kept the next record's load from being issued)
      for (uint32_t i0 = tid; i0 < n1; i0 += 4 * WG) {
        uint32_t al[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) al[r] = i0 + r * WG < n1 ? (uint32_t)(L1.p[i0 + r * WG].lo & 0xFFFFFFu) : 0u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t i = i0 + r * WG;
          if (i < n1) { const uint32_t old = atomicOr(&hVal[ljInsert<LJ_SLOTS>(hKey, al[r])], i + 1); if (old & 0x7FFFu) sDup = 1; }
        }
      }
      for (uint32_t j0 = tid; j0 < n2; j0 += 4 * WG) {
        uint32_t al[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) al[r] = j0 + r * WG < n2 ? (uint32_t)(L2.p[j0 + r * WG].lo & 0xFFFFFFu) : 0u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t j = j0 + r * WG;
          if (j < n2) { const uint32_t old = atomicOr(&hVal[ljInsert<LJ_SLOTS>(hKey, al[r])], (j + 1) << 16); if (old >> 16) sDup = 1; }
        }
      }
      __syncthreads();
      if (sDup) lds = false;  // a repeated allele: the sequential replay below works on the HBM tables
    }
    if (!lds) {
      for (uint32_t i = tid; i < n1; i += WG) {
        unsigned long long old = atomicExch((unsigned long long *)&tabSlot[L1[i].allele], (unsigned long long)(epoch | i));
        if ((old >> 32) == (epoch >> 32)) sDup = 1;
      }
      for (uint32_t j = tid; j < n2; j += WG) {
        unsigned long long old = atomicExch((unsigned long long *)&tab2[L2[j].allele], (unsigned long long)(epoch | j));
        if ((old >> 32) == (epoch >> 32)) sDup = 1;
      }
      __syncthreads();
    }
    const bool dup = sDup != 0;
    int tbM = -1, tbI = 0x7FFFFFFF;  // best fragment seen by this lane while it built the list (paired fast path only)
    double tbS = 0;
    bool tracked = false;
    PP(0);
    // ---- both mates have a list, no allele twice in a list (nearly every fragment): NO fragment list is materialised.  Round 3 wrote a
    // 48-byte record per joined allele into the workgroup's HBM scratch (about 700 a fragment) and read them back for the best / keep
    // passes: 51 KB written and twice that fetched per fragment (PMC, profiles/r04_before_traffic_work) for a row of ~14 entries.  The
    // best fragment only needs (matchCnt, similarity, position) of each joined pair -- computed from the two 16-byte records on the fly --
    // and the keep filter is a second sweep over list 1 that joins again through the LDS table and writes the FEW kept fragments, in
    // order, to frags[0 .. nKept).  Same decisions in the same order as the list form (SeqSet.hpp:2369-2545).
    const bool stream = both && !dup && !P.listForm;
    uint32_t nKeptStream = 0;
    if (stream) {
      tracked = true;
      const int s1 = ovlStrand(L1[0]), s2 = ovlStrand(L2[0]);
      uint32_t nPairs = 0;
      // PU records of list 1 per lane and round, all requested before the first is looked at, then their mates' records likewise: a list of
      // ~700 overlaps is one round of two dependent loads instead of three (the kernel waits for loads 85 % of its time, profiles/r05_pmc_sq.md)
      for (uint32_t i0 = tid; i0 < n1; i0 += PU * WG) {
        T1kOvlP pa[PU], pb[PU];
        int jv[PU], slotv[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) { const uint32_t i = i0 + u * WG; pa[u] = i < n1 ? L1.p[i] : T1kOvlP{0ull, 0ull}; }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
          const uint32_t i = i0 + u * WG;
          jv[u] = -1; slotv[u] = -1;
          if (i < n1 && s1 != s2) {
            const uint32_t al = (uint32_t)(pa[u].lo & 0xFFFFFFu);
            if (lds) { slotv[u] = ljFind<LJ_SLOTS>(hKey, al); jv[u] = slotv[u] >= 0 ? (int)(hVal[slotv[u]] >> 16) - 1 : -1; }
            else { const uint64_t e = tab2[al]; if ((e >> 32) == (epoch >> 32)) jv[u] = (int)(e & 0x3FFFFFFFu); }
          }
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) pb[u] = jv[u] >= 0 ? L2.p[jv[u]] : T1kOvlP{0ull, 0ull};
#pragma unroll
        for (int u = 0; u < PU; ++u) {
          const uint32_t i = i0 + u * WG;
          if (jv[u] < 0) continue;
          const T1kOvl oa = t1k_ovl_unpack(pa[u]), ob = t1k_ovl_unpack(pb[u]);
          if (!((s1 == 1 && oa.seqStart < ob.seqStart) || (s1 == -1 && oa.seqStart > ob.seqStart))) continue;  // 2369-2380
          Frag fr;
          makeFrag(fr, &oa, (int)i, &ob, jv[u]);
          ++nPairs;
          if (fr.matchCnt > tbM || (fr.matchCnt == tbM && fr.sim > tbS)) { tbM = fr.matchCnt; tbS = fr.sim; tbI = (int)i; }  // (a lane's i ascend: its first maximum stays)
          if (lds) atomicOr(&hVal[slotv[u]], 0x8000u); else tab2[oa.allele] |= 0x80000000ull;   // the mate's allele has a fragment (seqIdxToOverlapIdx membership)
        }
      }
      nFrag = nPairs;  // (this lane's share: only "any at all" is asked below, through sBestM)
    } else
    if (!dup) {
      // ---- fast path: every allele at most once per list -> `assign` == fragment list, in list order ----------------
      if (!both) {
        nFrag = n1 + n2;
        for (uint32_t q = tid; q < nFrag; q += WG) {
          if (q < n1) { const T1kOvl o = L1[q]; makeFrag(frags[q], &o, (int)q, nullptr, -1); } else { const T1kOvl o = L2[q - n1]; makeFrag(frags[q], nullptr, -1, &o, (int)(q - n1)); }
        }
      } else {
        tracked = true;
        const int s1 = ovlStrand(L1[0]), s2 = ovlStrand(L2[0]);
        for (uint32_t i0 = 0; i0 < n1; i0 += WG) {
          uint32_t i = i0 + tid;
          int j = -1;
          int slot = -1;
          if (i < n1 && s1 != s2) {
            int jj = -1;
            if (lds) {
              slot = ljFind<LJ_SLOTS>(hKey, L1[i].allele);
              jj = (int)(hVal[slot] >> 16) - 1;
            } else {
              uint64_t e = tab2[L1[i].allele];
              if ((e >> 32) == (epoch >> 32)) jj = (int)(e & 0x7FFFFFFFu);
            }
            if (jj >= 0 && ((s1 == 1 && L1[i].seqStart < L2[jj].seqStart) || (s1 == -1 && L1[i].seqStart > L2[jj].seqStart))) j = jj;  // 2369-2380
          }
          uint32_t tot;
          uint32_t off = scanExcl<NWAVE>(j >= 0 ? 1u : 0u, warpSums, &tot);
          if (j >= 0) {
            const T1kOvl oa = L1[i], ob = L2[j];
            Frag fr;
            makeFrag(fr, &oa, (int)i, &ob, j);
            frags[nFrag + off] = fr;
            // this lane's best fragment so far (its fragments come in list order: the first maximal one stays)
            if (fr.matchCnt > tbM || (fr.matchCnt == tbM && fr.sim > tbS)) { tbM = fr.matchCnt; tbS = fr.sim; tbI = (int)(nFrag + off); }
            // the mate's allele has a fragment (seqIdxToOverlapIdx membership)
            if (lds) atomicOr(&hVal[slot], 0x8000u); else tab2[L1[i].allele] |= 0x80000000ull;
          }
          nFrag += tot;
        }
      }
      __syncthreads();
    } else {
      // ---- slow path: lane 0 replays SeqSet.hpp:2320-2455 ----------------------------------------------------------------
      if (tid == 0) {
        uint32_t nA = 0;
        const uint64_t ep2 = epoch | 0x40000000ull;  // second stamp space for the slot table
        auto add = [&](const Frag &fr, const T1kOvl &o1) {
          uint64_t e = tabSlot[fr.allele];
          if ((e >> 32) == (epoch >> 32) && (e & 0x40000000ull)) {
            uint32_t s = (uint32_t)(e & 0x3FFFFFFFu);
            const T1kOvl &b1 = frags[s].i >= 0 ? L1[frags[s].i] : L2[frags[s].j];
            if (fragBefore(fr, o1, frags[s], b1)) frags[s] = fr;
          } else {
            tabSlot[fr.allele] = ep2 | nA;
            frags[nA++] = fr;
          }
        };
        Frag fr;
        if (!both) {
          for (uint32_t i = 0; i < n1; ++i) { const T1kOvl o = L1[i]; makeFrag(fr, &o, (int)i, nullptr, -1); add(fr, o); }
          for (uint32_t j = 0; j < n2; ++j) { const T1kOvl o = L2[j]; makeFrag(fr, nullptr, -1, &o, (int)j); add(fr, o); }
        } else {
          for (uint32_t i = 0; i < n1; ++i) {
            uint64_t e = tab2[L1[i].allele];
            if ((e >> 32) != (epoch >> 32)) continue;
            for (uint32_t j = 0; j < n2; ++j) {
              if (L2[j].allele != L1[i].allele) continue;
              int s1 = ovlStrand(L1[i]), s2 = ovlStrand(L2[j]);
              if (s1 == s2) continue;
              if ((s1 == 1 && L1[i].seqStart < L2[j].seqStart) || (s1 == -1 && L1[i].seqStart > L2[j].seqStart)) {
                const T1kOvl oa = L1[i], ob = L2[j];
                makeFrag(fr, &oa, (int)i, &ob, (int)j);
                add(fr, oa);
                tab2[L1[i].allele] |= 0x80000000ull;
              }
            }
          }
        }
        sN = nA;
      }
      __syncthreads();
      nFrag = sN;
    }
    PP(1);
    // ---- best fragment: max matchCnt, then max similarity, first in order (2474-2487) --------------------------------
    if (tracked) {
      // max matchCnt, then max similarity, then the smallest index: reduced over the lanes' own bests with shuffles
      int m = tbM, ix = tbI; double sm = tbS;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const int om = __shfl_xor(m, o, 64), oi = __shfl_xor(ix, o, 64);
        const double os = __shfl_xor(sm, o, 64);
        if (om > m || (om == m && (os > sm || (os == sm && oi < ix)))) { m = om; sm = os; ix = oi; }
      }
      __shared__ int sWM[NWAVE], sWI[NWAVE];
      __shared__ double sWS[NWAVE];
      if ((tid & 63) == 0) { sWM[tid >> 6] = m; sWS[tid >> 6] = sm; sWI[tid >> 6] = ix; }
      __syncthreads();
      if (tid == 0) {
        int bm = sWM[0], bi = sWI[0]; double bs = sWS[0];
        for (int w = 1; w < NWAVE; ++w)
          if (sWM[w] > bm || (sWM[w] == bm && (sWS[w] > bs || (sWS[w] == bs && sWI[w] < bi)))) { bm = sWM[w]; bs = sWS[w]; bi = sWI[w]; }
        sBestM = bm; sBestSim = bm >= 0 ? bs : -1.0; sBestIdx = bi;
      }
      __syncthreads();
    } else {
      {
        int bm = -1, bi = 0x7FFFFFFF; double bs = 0;
        for (uint32_t q = tid; q < nFrag; q += WG) {
          const Frag &fr = frags[q];
          if (fr.matchCnt > bm || (fr.matchCnt == bm && fr.sim > bs)) { bm = fr.matchCnt; bs = fr.sim; bi = (int)q; }
        }
        atomicMax(&sBestM, bm);
        __syncthreads();
        if (tid == 0) sBestSim = -1;
        __syncthreads();
        // among the lanes holding the best matchCnt: max similarity (serialised through a CAS-free two-step)
        __shared__ double sSim[WG];
        __shared__ int sIdx2[WG];
        sSim[tid] = (bm == sBestM && bm >= 0) ? bs : -1.0;
        sIdx2[tid] = (bm == sBestM && bm >= 0) ? bi : 0x7FFFFFFF;
        __syncthreads();
        for (int o = WG / 2; o > 0; o >>= 1) {
          if (tid < o) {
            double a = sSim[tid], b = sSim[tid + o];
            int ia = sIdx2[tid], ib = sIdx2[tid + o];
            if (b > a || (b == a && ib < ia)) { sSim[tid] = b; sIdx2[tid] = ib; }
          }
          __syncthreads();
        }
        if (tid == 0) { sBestSim = sSim[0]; sBestIdx = sIdx2[0]; }
        __syncthreads();
      }
      // a lane's local best is its first maximal element, but a later element of the same lane could tie the global best
      // with a smaller index than another lane's: resolve "first in order" exactly
      {
        int mine = 0x7FFFFFFF;
        for (uint32_t q = tid; q < nFrag; q += WG) {
          const Frag &fr = frags[q];
          if (fr.matchCnt == sBestM && fr.sim == sBestSim) { mine = (int)q; break; }
        }
        if (tid == 0) sBestIdx = 0x7FFFFFFF;
        __syncthreads();
        if (mine != 0x7FFFFFFF) atomicMin(&sBestIdx, mine);
        __syncthreads();
      }
    }
    const int bestM = sBestM;
    const double bestSim = sBestSim;
    int bestRelaxed = 0;
    if (stream) {
      if (bestM >= 0) {  // the best fragment again, from its two records (every lane computes the same)
        const T1kOvl oa = L1[sBestIdx];
        int jj;
        if (lds) jj = (int)(hVal[ljFind<LJ_SLOTS>(hKey, oa.allele)] >> 16) - 1; else jj = (int)(tab2[oa.allele] & 0x3FFFFFFFu);
        bestRelaxed = oa.relaxed + L2[jj].relaxed;
      }
    } else bestRelaxed = nFrag ? frags[sBestIdx].relaxed : 0;
    PP(2);
    // ---- keep filter (2488-2545), order-preserving -----------------------------------------------------------------------
    uint32_t nKept = 0;
    if (stream) {
      if (bestM >= 0) {
        const int s1 = ovlStrand(L1[0]), s2 = ovlStrand(L2[0]);
        for (uint32_t i0 = 0; i0 < n1; i0 += PU * WG) {  // (PU rounds' loads in flight together, as in the pass above; the rounds' scans keep list order)
          T1kOvlP pa[PU], pb[PU];
          int jv[PU];
#pragma unroll
          for (int u = 0; u < PU; ++u) { const uint32_t i = i0 + u * WG + tid; pa[u] = (i < n1 && s1 != s2) ? L1.p[i] : T1kOvlP{0ull, 0ull}; }
#pragma unroll
          for (int u = 0; u < PU; ++u) {
            const uint32_t i = i0 + u * WG + tid;
            jv[u] = -1;
            if (i < n1 && s1 != s2) {
              const uint32_t al = (uint32_t)(pa[u].lo & 0xFFFFFFu);
              if (lds) { const int sl = ljFind<LJ_SLOTS>(hKey, al); jv[u] = sl >= 0 ? (int)((hVal[sl] >> 16) & 0xFFFFu) - 1 : -1; }
              else { const uint64_t e = tab2[al]; if ((e >> 32) == (epoch >> 32)) jv[u] = (int)(e & 0x3FFFFFFFu); }
            }
          }
#pragma unroll
          for (int u = 0; u < PU; ++u) pb[u] = jv[u] >= 0 ? L2.p[jv[u]] : T1kOvlP{0ull, 0ull};
#pragma unroll
          for (int u = 0; u < PU; ++u) {
            if (i0 + u * WG >= n1) break;  // (uniform: the round lies behind the list)
            const uint32_t i = i0 + u * WG + tid;
            bool kp = false;
            Frag fr;
            if (jv[u] >= 0) {
              const T1kOvl oa = t1k_ovl_unpack(pa[u]), ob = t1k_ovl_unpack(pb[u]);
              if ((s1 == 1 && oa.seqStart < ob.seqStart) || (s1 == -1 && oa.seqStart > ob.seqStart)) {
                makeFrag(fr, &oa, (int)i, &ob, jv[u]);
                int relaxBy = 2;
                if (P.relax) {
                  const bool inter = (oa.seqStart <= ob.seqStart && oa.seqEnd >= ob.seqStart) || (ob.seqStart <= oa.seqStart && ob.seqEnd >= oa.seqStart);  // 317-324
                  if (inter && oa.matchCnt < oa.relaxed && ob.matchCnt < ob.relaxed) relaxBy = 4;
                }
                kp = (fr.matchCnt == bestM && fr.sim == bestSim) || (P.relax && fr.matchCnt >= bestM - relaxBy && fr.relaxed == bestRelaxed);
              }
            }
            uint32_t tot;
            const uint32_t off = scanExcl<NWAVE>(kp ? 1u : 0u, warpSums, &tot);
            if (kp) {
              if (nKept + off < fragCap) { frags[nKept + off] = fr; keep[nKept + off] = nKept + off; }
              else sFail = 2;  // (more kept fragments than the scratch holds: cannot happen -- kept <= joined <= n1 <= fragCap; guarded anyway)
            }
            nKept += tot;
          }
        }
      }
      nKeptStream = nKept;
    } else
    for (uint32_t q0 = 0; q0 < nFrag; q0 += WG) {
      uint32_t q = q0 + tid;
      bool kp = false;
      if (q < nFrag) {
        const Frag &fr = frags[q];
        int relaxBy = 2;
        if (P.relax && fr.i >= 0 && fr.j >= 0) {
          const T1kOvl &a = L1[fr.i], &b = L2[fr.j];
          bool inter = (a.seqStart <= b.seqStart && a.seqEnd >= b.seqStart) || (b.seqStart <= a.seqStart && b.seqEnd >= a.seqStart);  // 317-324
          if (inter && a.matchCnt < a.relaxed && b.matchCnt < b.relaxed) relaxBy = 4;
        }
        kp = (fr.matchCnt == bestM && fr.sim == bestSim) || (P.relax && fr.matchCnt >= bestM - relaxBy && fr.relaxed == bestRelaxed);
      }
      uint32_t tot;
      uint32_t off = scanExcl<NWAVE>(kp ? 1u : 0u, warpSums, &tot);
      if (kp) keep[nKept + off] = q;
      nKept += tot;
    }
    __syncthreads();
    PP(3);
    // ---- dangling-mate rule (2553-2578) ----------------------------------------------------------------------------------
    bool cleared = false;
    if (nKept > 0 && paired && !(frags[keep[0]].i >= 0 && frags[keep[0]].j >= 0)) {
      for (uint32_t q = tid; q < nKept; q += WG) {
        const Frag &fr = frags[keep[q]];
        const T1kOvl &o1 = fr.i >= 0 ? L1[fr.i] : L2[fr.j];
        bool bad = fr.sim < 1 || sepInRangeP(P.ref, fr.allele, fr.seqStart, fr.seqEnd) ||
                   (fr.seqEnd - fr.seqStart + 1 + o1.readEnd - o1.readStart + 1 < 3 * P.hitLenRequired);
        if (!bad) {
          const int spanRange = 100;
          if ((ovlStrand(o1) == 1 && fr.seqEnd + spanRange < (int)P.ref.alleleLen[fr.allele]) || (ovlStrand(o1) == -1 && fr.seqStart - spanRange >= 0)) bad = true;
        }
        if (bad) sFail = 1;
      }
      __syncthreads();
      cleared = sFail != 0;
    }
    // ---- truncated-reference rule (2580-2653) -----------------------------------------------------------------------------
    if (!cleared && nKept > 0 && paired && frags[keep[0]].i >= 0 && frags[keep[0]].j >= 0) {
      const Frag rep = frags[keep[0]];
      const T1kOvl r1 = L1[rep.i], r2 = L2[rep.j];
      const double r1s = ovlSim(r1), r2s = ovlSim(r2);
      for (uint32_t i0 = tid; i0 < n1; i0 += PU * WG) {
       T1kOvlP pv[PU];
#pragma unroll
       for (int u = 0; u < PU; ++u) pv[u] = i0 + u * WG < n1 ? L1.p[i0 + u * WG] : T1kOvlP{0ull, 0ull};
#pragma unroll
       for (int u = 0; u < PU; ++u) {
        if (i0 + u * WG >= n1) break;
        const T1kOvl o = t1k_ovl_unpack(pv[u]);
        double os = ovlSim(o);
        const bool tie = o.matchCnt == r1.matchCnt && os > r1s;
        bool inSlot = false;
        if (tie) {
          if (lds) inSlot = (hVal[ljFind<LJ_SLOTS>(hKey, o.allele)] & 0x8000u) != 0;
          else if (!dup) { uint64_t e = tab2[o.allele]; inSlot = (e >> 32) == (epoch >> 32) && (e & 0x80000000ull); }
          else { uint64_t e = tabSlot[o.allele]; inSlot = (e >> 32) == (epoch >> 32) && (e & 0x40000000ull); }
        }
        if (o.matchCnt > r1.matchCnt || (tie && !inSlot)) {
          if (truncatedMate(P.ref, o, r1, r2)) sFail = 1;
          else if (os > r2s + 0.1) sFail = 1;
        }
       }
      }
      for (uint32_t j0 = tid; j0 < n2; j0 += PU * WG) {
       T1kOvlP pv[PU];
#pragma unroll
       for (int u = 0; u < PU; ++u) pv[u] = j0 + u * WG < n2 ? L2.p[j0 + u * WG] : T1kOvlP{0ull, 0ull};
#pragma unroll
       for (int u = 0; u < PU; ++u) {
        if (j0 + u * WG >= n2) break;
        const T1kOvl o = t1k_ovl_unpack(pv[u]);
        double os = ovlSim(o);
        const bool tie = o.matchCnt == r2.matchCnt && os > r2s;
        bool inSlot = false;
        if (tie) {
          if (lds) inSlot = (hVal[ljFind<LJ_SLOTS>(hKey, o.allele)] & 0x8000u) != 0;
          else if (!dup) { uint64_t e = tab2[o.allele]; inSlot = (e >> 32) == (epoch >> 32) && (e & 0x80000000ull); }
          else { uint64_t e = tabSlot[o.allele]; inSlot = (e >> 32) == (epoch >> 32) && (e & 0x40000000ull); }
        }
        if (o.matchCnt > r2.matchCnt || (tie && !inSlot)) {
          if (truncatedMate(P.ref, o, r2, r1)) sFail = 1;
          else if (os > r1s + 0.1) sFail = 1;
        }
       }
      }
      __syncthreads();
      cleared = sFail != 0;
    }
    PP(4);
    if (cleared) nKept = 0;
    // ---- Genotyper::SetReadAssignments (778-832) ------------------------------------------------------------------------------
    bool emptyRow = nKept == 0 || (!P.rawKept && P.maxAssign > 0 && (int)nKept > P.maxAssign);
    if (!emptyRow) {
      for (uint32_t q = tid; q < nKept; q += WG) {
        const Frag &fr = frags[keep[q]];
        if (sepInRangeP(P.ref, fr.allele, fr.seqStart, fr.seqEnd)) sAnySep = 1;
        if (fr.sim >= 1) sNotOne = 1;  // maxSimilarity >= 1 somewhere
      }
      __syncthreads();
      if (sAnySep && !P.rawKept) emptyRow = true;
    }
    uint32_t nRow = emptyRow ? 0 : nKept;
    const bool anyKept = nKept > 0;  // fragmentAssigned is set before the -n / separator / whitelist drops
    if (nRow && P.whitelist && !P.rawKept) {      // order-preserving compaction of `keep`
      uint32_t w = 0;
      for (uint32_t q0 = 0; q0 < nRow; q0 += WG) {
        const uint32_t q = q0 + tid;
        const uint32_t kq = q < nRow ? keep[q] : 0;
        const bool ok = q < nRow && P.whitelist[frags[kq].allele] != 0;
        uint32_t tot;
        const uint32_t off = scanExcl<NWAVE>(ok ? 1u : 0u, warpSums, &tot);  // (its barriers separate the reads above from the writes below)
        if (ok) keep[w + off] = kq;
        w += tot;
      }
      __syncthreads();
      nRow = w;
    }
    const double adjust = sNotOne ? 1.0 : 0.25;  // 804-817
    if (!P.rsRowPtr) {
      if (tid == 0) {
        unsigned long long b = nRow ? atomicAdd(&P.counters[9], (unsigned long long)nRow) : 0ull;
        if (b + nRow > P.rowCap) { atomicOr(&P.counters[2], 128ull); sBase = 0xFFFFFFFFu; P.rowStart[f] = 0; P.rowCount[f] = 0; }
        else { sBase = (uint32_t)b; P.rowStart[f] = (uint32_t)b; P.rowCount[f] = nRow; }
        P.fragAssigned[f] = anyKept ? 1 : 0;
      }
      __syncthreads();
      if (nRow && sBase != 0xFFFFFFFFu) {
        for (uint32_t q = tid; q < nRow; q += WG) {
          const Frag &fr = frags[keep[q]];
          t1k_row_entry r;
          r.allele_idx = (int32_t)fr.allele; r.start = fr.seqStart; r.end = fr.seqEnd;
          r.weight = rowWeight(fr.sim, P.sim, hasN);
          r.qual = 1.0f;
          r.adjust_weight = (float)(adjust * r.weight);
          P.rows[(uint64_t)sBase + q] = r;
        }
      }
      __syncthreads();
      continue;
    }
    PP(5);
    // ---- rowset form: the row ordered by allele index (the order of a coalesced group's entries, Genotyper.hpp:847-853), its place
    // in the list kept in the `qual` slot (all assignment qualities are 1), and a 128-bit hash of the allele pattern ---------------
    __shared__ unsigned long long sRowBase;
    uint32_t *sAllele = hKey;  // [SORT_TILE]: the join table is done with by now (the barriers of the phases above lie between); 8 KB less
                               // LDS lets a fourth workgroup onto the CU
    static_assert(SORT_TILE <= LJ_SLOTS, "the rank-sort tile lives in the join table's keys");
    __shared__ unsigned long long sHash[2][NWAVE];
    if (tid == 0) {
      // (the row cursor: T1K_ROW_RESERVE > 0 takes that many entries at a time -- a row holds ~14 -- and serves the workgroup's next rows from what is
      // left; what a workgroup holds when the launch ends stays empty in the chunk: rows are found through rsRowPtr, never by position)
      unsigned long long b = 0;
      bool fitsChunk = true;
      if (nRow) {
        if (T1K_ROW_RESERVE > 0 && nRow <= rowLeft) { b = rowNext; rowNext += nRow; rowLeft -= nRow; }
        else {
          const unsigned long long take = (unsigned long long)(T1K_ROW_RESERVE > 0 && nRow < (uint32_t)T1K_ROW_RESERVE ? (uint32_t)T1K_ROW_RESERVE : nRow);
          b = atomicAdd(P.rsCursor, take);
          if (b + take > P.rsCap) fitsChunk = false; else { rowNext = b + nRow; rowLeft = (uint32_t)(take - nRow); }
        }
      }
      if (!fitsChunk) { atomicOr(&P.counters[2], 128ull); sRowBase = ~0ull; }
      else { sRowBase = b; statRows += nRow; }
    }
    // rank of every entry among the row's (distinct) alleles
    for (uint32_t q = tid; q < nRow; q += WG) frags[keep[q]].slot = 0;
    for (uint32_t t0 = 0; t0 < nRow; t0 += SORT_TILE) {
      const uint32_t tn = min(nRow - t0, (uint32_t)SORT_TILE);
      __syncthreads();
      for (uint32_t i = tid; i < tn; i += WG) sAllele[i] = frags[keep[t0 + i]].allele;
      __syncthreads();
      for (uint32_t q = tid; q < nRow; q += WG) {
        Frag &fr = frags[keep[q]];
        const uint32_t a = fr.allele;
        int c = 0;
        for (uint32_t i = 0; i < tn; ++i) c += sAllele[i] < a ? 1 : 0;
        fr.slot += c;
      }
    }
    __syncthreads();
    PP(6);
    unsigned long long h1 = 0, h2 = 0;
    const bool fits = sRowBase != ~0ull;
    for (uint32_t q = tid; q < nRow; q += WG) {
      const Frag &fr = frags[keep[q]];
      const uint32_t r = (uint32_t)fr.slot;
      h1 += t1k_pattern_mix(fr.allele, r, 0x9E3779B97F4A7C15ull);
      h2 += t1k_pattern_mix(fr.allele, r, 0xC2B2AE3D27D4EB4Full);
      if (fits) {
        t1k_row_entry e;
        e.allele_idx = (int32_t)fr.allele; e.start = fr.seqStart; e.end = fr.seqEnd;
        e.weight = rowWeight(fr.sim, P.sim, hasN);
        e.qual = __uint_as_float(q);  // position in the reference's row order (read back by t1k_rowset_rows_download)
        e.adjust_weight = (float)(adjust * e.weight);
        P.rsRows[sRowBase + r] = e;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { h1 += __shfl_down(h1, o, 64); h2 += __shfl_down(h2, o, 64); }
    if ((tid & 63) == 0) { sHash[0][tid >> 6] = h1; sHash[1][tid >> 6] = h2; }
    __syncthreads();
    if (tid == 0) {
      const uint64_t g = P.fragBase + f;
      P.rsRowPtr[g] = fits ? (unsigned long long)(P.rsRows + sRowBase) : 0ull;
      P.rsRowCount[g] = fits ? nRow : 0;
      unsigned long long t1 = 0, t2 = 0;
#pragma unroll
      for (int w = 0; w < NWAVE; ++w) { t1 += sHash[0][w]; t2 += sHash[1][w]; }
      P.rsH1[g] = t1 + 0x632BE59BD9B4E019ull * nRow;
      P.rsH2[g] = t2 + 0xA0761D6478BD642Full * nRow;
      P.rsAssigned[g] = anyKept ? 1 : 0;
    }
    __syncthreads();
    PP(7);
  }
  if (tid == 0) {  // the workgroup's statistics, once
    if (statOvl) atomicAdd(&P.counters[22], statOvl);
    if (statRows) atomicAdd(&P.counters[9], statRows);
  }
#ifdef T1K_PAIR_PROFILE
  if (tid == 0) for (int i = 0; i < 8; ++i) atomicAdd(&P.counters[32 + i], (unsigned long long)tp_[i]);
#endif
}

// shared by t1k_pair_batch (rs == NULL) and t1k_pair_into
static int pairLaunch(t1k_ctx *ctx, t1k_rowset *rs, const uint32_t *end1, const uint32_t *end2, const uint8_t *hasN, uint32_t nFragments, uint64_t fragBase,
                      const uint8_t *dWhitelist) {
  if (!ctx->ref.bases) return t1k_fail(ctx, T1K_ERR_STATE, "mate pairing: no reference");
  if (!ctx->reads.listPtr) return t1k_fail(ctx, T1K_ERR_STATE, "mate pairing: no reads uploaded");
  T1K_HIP(ctx, hipSetDevice(ctx->device));
  int rc;
  const uint32_t n = nFragments;
  // threads per fragment x slots of the LDS join table: 256 x 4096 (the other shapes of the template -- 128 x 4096, 128 x 2048, 64 x 2048, and
  // 256 x 4096 / 2048 with the lists copied into LDS -- were measured in round 3 and are slower); 1024 workgroups keep
  // the wave slots of the chip filled (the kernel is latency-bound)
  const int wgThreads = 256;
  // (T1K_PAIR_WGS: workgroups = fragments in flight, for A/B -- round 6 measured the kernel's fabric traffic at 127 KB per fragment for 22 KB of
  // records: each of its four sweeps over a fragment's two lists misses the L2 that 128 fragments per XCD share)
  static const int maxWg = [] { const char *e = getenv("T1K_PAIR_WGS"); return e ? std::max(64, std::min(4096, atoi(e))) : 1024; }();
  const int nWg = (int)std::min<uint32_t>(maxWg, std::max<uint32_t>(n, 1));
  auto launch = [&](unsigned grid, const PairArgs &args) { hipLaunchKernelGGL((k_pair<256, 4096>), dim3(grid), dim3(256), 0, ctx->stream, args); };
  static const uint32_t envFragCap = [] { const char *e = getenv("T1K_PAIR_FRAGCAP"); return e ? (uint32_t)std::max(8, atoi(e)) : 8192u; }();  // (tests: force the second launch)
  static const uint64_t envBigCap = [] { const char *e = getenv("T1K_PAIR_BIGCAP"); return e ? (uint64_t)std::max(64, atoi(e)) : (uint64_t)(8u << 20); }();
  const uint32_t fragCap = envFragCap;   // overlaps of both mates a workgroup's own scratch holds; longer fragments: the second launch, scratch from the big arena
  const uint32_t A = ctx->ref.nAlleles;
  if (!rs) { ctx->nFragments = n; ctx->nRows = 0; }
  if ((rc = t1k_ensure(ctx, ctx->bEnd1, (size_t)n * 4))) return rc;
  if ((rc = t1k_ensure(ctx, ctx->bEnd2, (size_t)n * 4))) return rc;
  if ((rc = t1k_ensure(ctx, ctx->bHasN, (size_t)n))) return rc;
  if (!rs) {
    if ((rc = t1k_ensure(ctx, ctx->bRows, (size_t)ctx->prm.row_cap * sizeof(t1k_row_entry)))) return rc;
    if ((rc = t1k_ensure(ctx, ctx->bRowStart, (size_t)n * 4))) return rc;
    if ((rc = t1k_ensure(ctx, ctx->bRowCount, (size_t)n * 4))) return rc;
    if ((rc = t1k_ensure(ctx, ctx->bFragAssigned, (size_t)n))) return rc;
  }
  size_t perWg = (size_t)A * 16 + (size_t)fragCap * (sizeof(Frag) + 4);
  bool fresh = ctx->bPairScratch.bytes < (size_t)maxWg * perWg;
  if ((rc = t1k_ensure(ctx, ctx->bPairScratch, (size_t)maxWg * perWg))) return rc;
  if ((rc = t1k_ensure(ctx, ctx->bPairOverflow, (size_t)n * 8 + 16))) return rc;
  // the big arena: 52 bytes per overlap of the fragments that go to the second launch; starts at 8 M entries, grows to the demand the device counted
  if (ctx->pairBigCap == 0) ctx->pairBigCap = envBigCap;
  if ((rc = t1k_ensure(ctx, ctx->bPairBig, (size_t)ctx->pairBigCap * (sizeof(Frag) + 4)))) return rc;
  if ((rc = t1k_ensure(ctx, ctx->bCounters, (size_t)T1K_COUNTER_WORDS * 8))) return rc;
  if (n == 0) return T1K_OK;
  T1K_HIP(ctx, hipMemcpyAsync(ctx->bEnd1.p, end1, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  if (end2) T1K_HIP(ctx, hipMemcpyAsync(ctx->bEnd2.p, end2, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  if (hasN) T1K_HIP(ctx, hipMemcpyAsync(ctx->bHasN.p, hasN, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  else T1K_HIP(ctx, hipMemsetAsync(ctx->bHasN.p, 0, (size_t)n, ctx->stream));
  // the epoch of fragment f is epochBase + f + 1; the base moves on with every call, so the tables only need clearing when the
  // 32-bit epoch space is about to wrap (or the scratch was just allocated)
  if (fresh || ctx->pairEpoch > 0xFFFFFFFFull - 12ull * n - 2) {
    T1K_HIP(ctx, hipMemsetAsync(ctx->bPairScratch.p, 0, (size_t)maxWg * (size_t)A * 16, ctx->stream));
    ctx->pairEpoch = 0;
  }
  PairArgs p{};
  p.ref = ctx->ref;
  p.listPtr = ctx->reads.listPtr; p.listCount = ctx->reads.listCount;
  p.end1 = (const uint32_t *)ctx->bEnd1.p; p.end2 = end2 ? (const uint32_t *)ctx->bEnd2.p : nullptr; p.hasN = (const uint8_t *)ctx->bHasN.p;
  p.nFragments = n; p.sim = ctx->prm.ref_seq_similarity; p.relax = ctx->prm.relax_intron_align; p.maxAssign = ctx->prm.max_assign_cnt;
  p.hitLenRequired = ctx->prm.hit_len_required;
  p.rows = (t1k_row_entry *)ctx->bRows.p; p.rowCap = (uint64_t)ctx->prm.row_cap;
  p.rowStart = (uint32_t *)ctx->bRowStart.p; p.rowCount = (uint32_t *)ctx->bRowCount.p; p.fragAssigned = (uint8_t *)ctx->bFragAssigned.p;
  uint8_t *sc = (uint8_t *)ctx->bPairScratch.p;
  p.tab2 = (uint64_t *)sc;
  p.tabSlot = (uint64_t *)(sc + (size_t)maxWg * A * 8);
  p.frags = (Frag *)(sc + (size_t)maxWg * A * 16);
  p.fragCap = fragCap;
  p.keep = (uint32_t *)(sc + (size_t)maxWg * A * 16 + (size_t)maxWg * fragCap * sizeof(Frag));
  p.overflowList = (uint32_t *)ctx->bPairOverflow.p;
  p.counters = (unsigned long long *)ctx->bCounters.p;
  p.whitelist = dWhitelist;
  { static const int lf = getenv("T1K_PAIR_LIST") ? 1 : 0; p.listForm = lf; }
  if (getenv("T1K_DEBUG_TRACE")) fprintf(stderr, "[t1k trace] pair %u fragments%s\n", n, rs ? " into the rowset" : "");
  for (int attempt = 0;; ++attempt) {
    T1K_HIP(ctx, hipMemsetAsync((char *)ctx->bCounters.p + 2 * 8, 0, 8, ctx->stream));
    T1K_HIP(ctx, hipMemsetAsync((char *)ctx->bCounters.p + 9 * 8, 0, 8, ctx->stream));
    T1K_HIP(ctx, hipMemsetAsync((char *)ctx->bCounters.p + 22 * 8, 0, 40, ctx->stream));  // [22] overlaps read, [23] long fragments, [24] their arena cursor, [25] / [26] next fragment of the two launches
    size_t chunk = 0;
    if (rs) {
      if ((rc = t1k_rowset_chunk(rs, ctx, &chunk, &p.rsRows, &p.rsCap, &p.rsCursor))) return rc;
      p.rawKept = rs->rawKept ? 1 : 0;
      p.rsRowPtr = rs->rowPtr; p.rsRowCount = rs->rowCount; p.rsH1 = rs->h1; p.rsH2 = rs->h2; p.rsAssigned = rs->assigned; p.fragBase = fragBase;
    }
    p.epochBase = (uint32_t)ctx->pairEpoch; ctx->pairEpoch += n;
    p.only = nullptr; p.overflowList = (uint32_t *)ctx->bPairOverflow.p;
    T1K_HIP(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
    launch((unsigned)nWg, p);
    {
      // second launch, always: the fragments whose lists exceed the first pass's scratch (with 29 k alleles the large genes put whole classes
      // of fragments here).  Their number is on the device (counters[23]); the workgroups beyond it leave at once.  Same per-workgroup
      // allele tables as the first launch (the kernels run one after the other on this stream), fresh epochs.
      PairArgs q = p;
      q.only = (const uint32_t *)ctx->bPairOverflow.p; q.overflowList = nullptr;
      q.bigFrags = (Frag *)ctx->bPairBig.p; q.bigKeep = (uint32_t *)((char *)ctx->bPairBig.p + (size_t)ctx->pairBigCap * sizeof(Frag)); q.bigCap = ctx->pairBigCap;
      q.epochBase = (uint32_t)ctx->pairEpoch; ctx->pairEpoch += n;
      launch((unsigned)nWg, q);
    }
    T1K_HIP(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
    if (!ctx->countersPinned) T1K_HIP(ctx, hipHostMalloc((void **)&ctx->countersPinned, (size_t)T1K_COUNTER_WORDS * 8, hipHostMallocDefault));
    T1K_HIP(ctx, hipMemcpyAsync(ctx->countersPinned, ctx->bCounters.p, 64 * 8, hipMemcpyDeviceToHost, ctx->stream));
    T1K_HIP(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long *hc = ctx->countersPinned;
    { float ms = 0; (void)hipEventElapsedTime(&ms, ctx->ev[5], ctx->ev[6]); ctx->stats.ms_pair = (attempt ? ctx->stats.ms_pair : 0) + ms; }
    if ((hc[2] & 1024ull) && attempt < 4) {
      // the big arena was too small for this call's long fragments: the device counted the demand (counters[24]); fragments are written
      // by index, so the call simply runs again (the rows of the first attempt are left behind in the rowset's chunk)
      ctx->pairBigCap = (uint64_t)((double)hc[24] * 1.25) + (getenv("T1K_PAIR_BIGCAP") ? 16 : (1u << 20));
      if ((rc = t1k_ensure(ctx, ctx->bPairBig, (size_t)ctx->pairBigCap * (sizeof(Frag) + 4)))) return rc;
      if (getenv("T1K_DEBUG_TRACE")) fprintf(stderr, "[t1k trace] pair: big arena grows to %llu entries\n", (unsigned long long)ctx->pairBigCap);
      if (!rs || !(hc[2] & ~1024ull)) continue;
    }
    if (hc[2] && rs && attempt < 4) {
      // the rowset's current chunk is full (or a fragment has more overlaps than the scratch holds: that repeats and fails below):
      // close the chunk and run the call again; fragments are written by index, the first attempt's rows are simply left behind
      if ((rc = t1k_rowset_chunk_full(rs, ctx, chunk))) return rc;
      continue;
    }
    if (hc[2]) return t1k_fail(ctx, T1K_ERR_CAPACITY, "device arena overflow: row_cap / fragment scratch");
    if (!rs) ctx->nRows = hc[9];
    ctx->stats.rows = hc[9]; ctx->stats.pair_overlaps = hc[22];
#ifdef T1K_PAIR_PROFILE
    fprintf(stderr, "[t1k] pair phases (ticks, thread 0 of every workgroup): tables %llu join %llu best %llu keep %llu rules %llu setreads %llu rank %llu rows+hash %llu\n", hc[32], hc[33], hc[34], hc[35], hc[36], hc[37], hc[38], hc[39]);
#endif
    return T1K_OK;
  }
}

extern "C" {

int t1k_pair_batch(t1k_ctx *ctx, const uint32_t *end1, const uint32_t *end2, const uint8_t *hasN, uint32_t nFragments) {
  if (!ctx || !end1) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_pair_batch: bad arguments");
  return pairLaunch(ctx, nullptr, end1, end2, hasN, nFragments, 0, nullptr);
}

int t1k_pair_into(t1k_ctx *ctx, t1k_rowset *rs, const uint32_t *end1, const uint32_t *end2, const uint8_t *hasN, uint32_t nFragments, uint64_t fragBase) {
  if (!ctx || !rs || !end1) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_pair_into: bad arguments");
  if (fragBase + nFragments > rs->nFrag) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_pair_into: fragments outside the rowset");
  if (rs->device != ctx->device) return t1k_fail(ctx, T1K_ERR_ARG, "t1k_pair_into: rowset lives on another device");
  return pairLaunch(ctx, rs, end1, end2, hasN, nFragments, fragBase, rs->whitelist);
}

int t1k_rows_download(t1k_ctx *ctx, uint32_t *rowCounts, uint8_t *fragAssigned, t1k_row_entry *rows, uint64_t cap, uint64_t *total) {
  if (!ctx) return T1K_ERR_ARG;
  T1K_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t n = ctx->nFragments;
  std::vector<uint32_t> start(n), cnt(n);
  if (n) {
    T1K_HIP(ctx, hipMemcpy(start.data(), ctx->bRowStart.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    T1K_HIP(ctx, hipMemcpy(cnt.data(), ctx->bRowCount.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (fragAssigned) T1K_HIP(ctx, hipMemcpy(fragAssigned, ctx->bFragAssigned.p, (size_t)n, hipMemcpyDeviceToHost));
  }
  uint64_t tot = 0;
  for (uint32_t i = 0; i < n; ++i) tot += cnt[i];
  if (total) *total = tot;
  if (rowCounts) memcpy(rowCounts, cnt.data(), (size_t)n * 4);
  if (!rows) return T1K_OK;
  if (cap < tot) return t1k_fail(ctx, T1K_ERR_ARG, "row buffer too small");
  std::vector<t1k_row_entry> h(ctx->nRows);
  if (ctx->nRows) T1K_HIP(ctx, hipMemcpy(h.data(), ctx->bRows.p, ctx->nRows * sizeof(t1k_row_entry), hipMemcpyDeviceToHost));
  uint64_t w = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (cnt[i]) memcpy(rows + w, h.data() + start[i], (size_t)cnt[i] * sizeof(t1k_row_entry));
    w += cnt[i];
  }
  return T1K_OK;
}

}  // extern "C"
