// t1k_amd/csrc/t1k_extract.hip -- the candidate-read test of the reference's fastq-extractor on gfx950:
// IsGoodCandidate = !IsLowComplexity(read) && SeqSet::HasHitInSet(read) (FastqExtractor.cpp:89-118, SeqSet.hpp:1915-1990), for a batch
// of fragments resident in HBM.  All integer, HBM/LDS-bound work; no MFMA.
//
//   k_extract_screen  one wavefront per read-end: base counts (IsLowComplexity, FastqExtractor.cpp:89-111) and one look-up per k-mer
//               position, in one burst of independent loads, into presence bitmaps of the index (a position and its reverse complement
//               share the first look-up); a read-end whose strands cannot collect hitLenRequired / k hits in one bucket
//               (SeqSet.hpp:1925-1927, 1959) is finished here -- that is nearly every read of a sequencing run.  No LDS.
//   k_extract   the remaining read-ends, one 256-thread workgroup per fragment (workgroups walk blocks of 256 fragments and compact
//               the ones with work left); the mate is only looked at when the first end fails, as in the reference
//               (FastqExtractor.cpp:459-464).  Per read-end:
//     1. every k-mer of both strands is looked up in the direct-address index at once (one load round); the look-up rule
//        (prevKmerCode / skipCnt, GetHitsFromRead SeqSet.hpp:1071-1229) is applied in its parallel form, reads with short repeats are
//        replayed sequentially by the first wavefront over registers
//     2. hits per (strand, sequence) counted in an LDS histogram, 4096 sequences at a time and only over the span of sequences the
//        strand's lists name; minus strand first, first maximum wins (SeqSet.hpp:1934-1957); k * max < hitLenRequired ends the read (1959)
//     3. the winning bucket's hits are gathered (bisection in each used posting list), rank-sorted by (diagonal, sequence offset,
//        read offset) by the whole workgroup, and thread 0 walks the diagonal runs: nearest-to-dominant filter, LIS, hit lengths
//        (GetOverlapsFromHits with filter 0, SeqSet.hpp:1232-1556); the read is a candidate if some overlap has
//        len - hitLen <= int(len * (1 - similarity)) * k  (1974-1979).  A bucket whose hits all lie on one diagonal (the read differs
//        from the sequence by substitutions only) needs none of that: one run, every hit kept, LIS = all hits, and both hit lengths are
//        the union of the k-mer intervals -- a parallel sum
#include <algorithm>
#include "t1k_dev.h"
#include "t1k_launch.h"

// Shapes.  The production kernels give a lane / thread a fixed number of k-mer positions and used lists, sized for reads of up to
// T1K_MAX_READ_LEN bases; a batch that holds a longer read (up to T1K_LONG_READ_LEN) runs the same code with larger counts (template
// parameters PS: positions per lane of the screen, PT: positions per thread, LT: used lists per thread) -- more registers, same results.
#define X_PS_FAST 5
#define X_PT_FAST 3
#define X_LT_FAST 2
#define X_PS_LONG 16
#define X_PT_LONG 8
#define X_LT_LONG 4
static_assert(T1K_MAX_READ_LEN <= X_PS_FAST * 64 && T1K_LONG_READ_LEN <= X_PS_LONG * 64, "k_extract_screen: a lane takes PS consecutive k-mer positions of a 64-lane wavefront");
static_assert(X_PS_LONG + 15 - 1 <= 32, "k_extract_screen: a lane cuts its codes (k <= 15) out of one 32-position window");
static_assert(2 * T1K_MAX_READ_LEN <= X_PT_FAST * 256 && 2 * T1K_LONG_READ_LEN <= X_PT_LONG * 256, "k_extract: a thread takes PT k-mer positions (both strands) of a 256-thread workgroup");
static_assert(T1K_MAX_READ_LEN <= X_LT_FAST * 256 && T1K_LONG_READ_LEN <= X_LT_LONG * 256, "k_extract: a thread takes LT used lists of a strand");

#include "t1k_group.h"

#define XWG 256
// k_extract<X_RANGE>: sequences per histogram pass (u32 counters in LDS); the same words later hold the bucket's hits and the three
// work arrays of the chain, so a bucket may have X_RANGE / 4 hits.  4096 (16 KB, 1024 hits) is the production shape (measured: 8192 -> 8.5 ms,
// 4096 -> 7.4 ms, 2048 -> 8.1 ms per 4 M pairs: occupancy against the number of passes; 6.6 ms with the kernel held to 80 VGPRs); a batch in which some bucket is larger is run
// again with 32768 (128 KB, one workgroup per CU, 8192 hits).
#define X_RANGE_SMALL 4096
#define X_RANGE_BIG 32768
#define X_RANGE_BIG_LONG 24576   // the large shape of a batch with reads beyond T1K_MAX_READ_LEN: its k-mer tables take 48 KB of the 160
enum { XERR_HITCAP = 1 };
#ifdef T1K_XPROF
#define XP(i) { const uint64_t tn_ = __builtin_amdgcn_s_memtime(); xp_[i] += tn_ - xl_; xl_ = tn_; }
#else
#define XP(i)
#endif

struct ExtractArgs {
  T1kRefDev ref;
  T1kReadsDev reads;
  int k, radius, hitLenRequired;
  double oneMinusSim;
  uint32_t nFragments, epf, maxK;
  uint8_t *good;
  uint8_t *state;             // [read-end] written by k_extract_screen: 1 = the read-end may have a hit (k_extract decides)
  unsigned long long *err;
  unsigned long long *stats;  // [0] read-ends screened, [1] look-ups, [2] postings streamed, [3] read-ends reaching the histogram, [4] reaching the chain
  uint32_t *ghScratch;        // k_extract<..., true>: the bucket's hit arrays H | A | B | C live here, ghCap words each per workgroup (HBM instead of LDS)
  uint32_t ghCap;
};

// slice [lo, hi) of a posting list (sorted by sequence) holding the sequences [a0, a1)
__device__ __forceinline__ uint32_t listLowerBound(const uint32_t *p, uint32_t n, uint32_t allele) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (p[mid] < allele) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// popcounts of C, G, T and N in one packed word (A is what is left of the length)
__device__ __forceinline__ void baseCounts(uint64_t b, uint64_t n, int &c, int &g, int &t, int &nn) {
  const uint64_t nm = n & T1K_EVEN, lo = b & T1K_EVEN & ~nm, hi = (b >> 1) & T1K_EVEN & ~nm;  // whatever bits sit under an N do not count
  c = __popcll(lo & ~hi); g = __popcll(hi & ~lo); t = __popcll(lo & hi); nn = __popcll(nm);
}
__device__ __forceinline__ bool lowComplexity(int len, int cC, int cG, int cT, int cN) {  // IsLowComplexity (FastqExtractor.cpp:89-111)
  const int cA = len - cC - cG - cT - cN, half = len / 2;
  if (cA >= half || cC >= half || cG >= half || cT >= half || cN >= len / 10) return true;
  return (cA <= 2) + (cC <= 2) + (cG <= 2) + (cT <= 2) >= 2;
}

template <int PS>
__global__ __launch_bounds__(XWG) void k_extract_screen(ExtractArgs P) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = blockIdx.x * (XWG / 64) + (threadIdx.x >> 6), nWaves = gridDim.x * (XWG / 64);
  const int k = P.k;
  const uint32_t kmask = (1u << (2 * k)) - 1;
  const int S = P.reads.S;
  // a wavefront takes 64 consecutive read-ends at a time, lane i keeps the verdict of the i-th: one 64-byte store instead of 64 one-byte ones
  for (uint32_t base = wave * 64u; base < P.reads.nReadEnds; base += nWaves * 64u) {
   uint32_t mine = 0;
   const uint32_t cnt = min(64u, P.reads.nReadEnds - base);
   for (uint32_t i = 0; i < cnt; ++i) {
    const uint32_t re = base + i;
    const int len = P.reads.len[re];
    const uint64_t *rbase = P.reads.bases + (uint64_t)re * 2 * S;
    const uint64_t *rnm = P.reads.nmask + (uint64_t)re * 2 * S;
    bool live = len >= k, hasN = false;  // HasHitInSet 1919-1920
    if (live) {
      int cC = 0, cG = 0, cT = 0, cN = 0;
      if (lane < (len + 31) / 32) baseCounts(rbase[lane], rnm[lane], cC, cG, cT, cN);
      for (int o = 32; o > 0; o >>= 1) { cC += __shfl_xor(cC, o, 64); cG += __shfl_xor(cG, o, 64); cT += __shfl_xor(cT, o, 64); cN += __shfl_xor(cN, o, 64); }
      live = !lowComplexity(len, cC, cG, cT, cN);
      hasN = cN > 0;
    }
    // Per strand: how many positions have a non-empty list, and whether one of those lists names a sequence twice.  Without such a list
    // a (strand, sequence) bucket holds at most one hit per position, and a bucket with fewer than ceil(hitLenRequired / k) hits fails
    // SeqSet.hpp:1959 whichever bucket wins the vote: the read-end is finished here unless some strand can reach that many hits.
    int nz0 = 0, nz1 = 0;
    bool multi = false;
    if (live) {
      const int nk = len - k + 1;
      // Two presence bitmaps.  A window and its reverse complement are the same k-mer position seen from the two strands, so the first
      // one is keyed by the smaller of the two codes (its first k - 2 bases: 2 MB, stays in L2) and one look-up per forward position
      // screens both strands; only the positions it lets through ask the full bitmap, once per strand.  A lane owns `per` consecutive
      // positions and cuts their codes out of two packed words it loads once (per + k - 1 <= 32 positions).
      const uint32_t pmask = k > 2 ? (1u << (2 * (k - 2))) - 1 : kmask;
      const int per = (nk + 63) / 64;  // <= PS
      const int p0 = lane * per;
      uint32_t code[PS], rcode[PS], w[PS];
      uint64_t bits = 0, nbits = 0;
      if (p0 < nk) {
        bits = t1k_get32(rbase, p0);
        if (hasN) nbits = t1k_get32(rnm, p0);
      }
#pragma unroll
      for (int j = 0; j < PS; ++j) {
        code[j] = (uint32_t)(bits >> (2 * j)) & kmask;
        rcode[j] = t1k_code_revcomp(code[j], k);
        w[j] = 0;
        const bool valid = j < per && p0 + j < nk && ((uint32_t)(nbits >> (2 * j)) & kmask) == 0;
        if (valid) w[j] = P.ref.kHasPre[(min(code[j], rcode[j]) & pmask) >> 5];
      }
#pragma unroll
      for (int j = 0; j < PS; ++j) {
        if (j >= per) break;  // uniform
        bool hit0 = false, hit1 = false;
        if ((w[j] >> (min(code[j], rcode[j]) & 31u)) & 1u) {
          const uint32_t c = code[j], r = rcode[j];  // r is the code of the minus strand's position nk - 1 - (p0 + j)
          hit0 = ((P.ref.kHas[c >> 5] >> (c & 31u)) & 1u) != 0;
          hit1 = ((P.ref.kHas[r >> 5] >> (r & 31u)) & 1u) != 0;
          if (hit0) multi |= ((P.ref.kMulti[c >> 5] >> (c & 31u)) & 1u) != 0;
          if (hit1) multi |= ((P.ref.kMulti[r >> 5] >> (r & 31u)) & 1u) != 0;
        }
        nz0 += __popcll(__ballot(hit0));
        nz1 += __popcll(__ballot(hit1));
      }
    }
    const int needHits = (P.hitLenRequired + k - 1) / k;
    const bool anyWave = __ballot(multi) != 0ull ? (nz0 + nz1 > 0) : (nz0 >= needHits || nz1 >= needHits);
    if (lane == (int)i) mine = (live && anyWave) ? 1u : 0u;
   }
   if ((uint32_t)lane < cnt) P.state[base + lane] = (uint8_t)mine;
  }
}

// GH: third attempt of a batch -- a bucket beyond the large LDS shape (tandem repeats: thousands of hits of one read on one sequence).  The hit
// arrays move to a per-workgroup piece of HBM (up to 65 535 hits: the LIS links are 16-bit); same code, slow by design, never the common path.
template <int X_RANGE, int PT, int LT, bool GH = false>
__global__ __launch_bounds__(XWG) __attribute__((amdgpu_waves_per_eu(6))) void k_extract(ExtractArgs P) {
  const uint32_t X_HCAP = GH ? P.ghCap : (uint32_t)(X_RANGE / 4);
  extern __shared__ uint32_t lds[];
  const int maxK = (int)P.maxK;
  uint32_t *ukCode = lds;                        // [maxK] code | valid << 31, both strands
  uint32_t *ukStart = ukCode + maxK;             // [maxK]
  uint32_t *ukLen = ukStart + maxK;              // [maxK]
  uint32_t *ukDir = ukLen + maxK;                // [maxK] chunk-directory row of the list (lists longer than T1K_DIR_MINLEN)
  uint32_t *sliceLo = ukDir + maxK;              // [maxK] slice of the current pass inside each used list (later: minDist of the chain)
  uint32_t *pre = sliceLo + maxK;                // [maxK + 1] exclusive prefix of the slice lengths
  uint16_t *usedQ = (uint16_t *)(pre + maxK + 2);  // [maxK] used look-ups, + strand first
  uint32_t *hist = (uint32_t *)(usedQ + maxK + (maxK & 1));  // [X_RANGE]; after the vote: H | A | B | C, X_HCAP words each
  __shared__ uint32_t warpSums[4];
  __shared__ uint32_t sUsed[2];
  __shared__ unsigned long long sKey[4];
  __shared__ int sRes, sMulti, sLen;
  __shared__ int sWaveMax[4];
  __shared__ uint32_t sRed[12];
  __shared__ uint32_t sPost;
  __shared__ uint16_t sList[XWG];
  const int tid = threadIdx.x;
  const int k = P.k;
  const uint32_t kmask = (1u << (2 * k)) - 1;
  const uint32_t A = P.ref.nAlleles;
#ifdef T1K_XPROF
  uint64_t xp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xl_ = __builtin_amdgcn_s_memtime();
#endif
  unsigned long long stLook = 0, stPost = 0, stHist = 0, stChain = 0;  // thread 0 tallies

  for (uint32_t blk = blockIdx.x; (uint64_t)blk * XWG < P.nFragments; blk += gridDim.x) {
   // fragments of this block with a read-end the screen could not finish, in order
   const uint32_t f0 = blk * XWG;
   uint32_t nList;
   {
     uint32_t need = 0;
     if (f0 + tid < P.nFragments) {
       need = P.state[(uint64_t)(f0 + tid) * P.epf];
       if (P.epf == 2) need |= (uint32_t)P.state[(uint64_t)(f0 + tid) * 2 + 1] << 1;
       if (!need) P.good[f0 + tid] = 0;
     }
     __syncthreads();  // the previous block is done with sList
     const uint32_t slot = t1k_block_scan_exclusive(need ? 1u : 0u, warpSums, &nList);
     if (need) sList[slot] = (uint16_t)(tid | (need << 8));
     __syncthreads();
   }
   XP(0)
   for (uint32_t li = 0; li < nList; ++li) {
    const uint32_t f = f0 + (sList[li] & 255u), need = sList[li] >> 8;
    bool fragGood = false;
    for (uint32_t j = 0; j < P.epf && !fragGood; ++j) {
      if (!((need >> j) & 1u)) continue;
      const uint32_t re = f * P.epf + j;
      const int len = P.reads.len[re];
      const int S = P.reads.S;
      const uint64_t *rbase = P.reads.bases + (uint64_t)re * 2 * S;
      const uint64_t *rnm = P.reads.nmask + (uint64_t)re * 2 * S;
      __syncthreads();  // the previous read-end is done with the shared words
      if (tid == 0) { sUsed[0] = 0; sUsed[1] = 0; sRes = 0; sMulti = 0; sLen = 0; sPost = 0; }
      __syncthreads();
      if (len < k) continue;  // HasHitInSet 1919-1920 (uniform)
      // (1. IsLowComplexity and len < k were decided by the screen: such read-ends never get here)
      // ---- 2. look-ups of both strands
      const int nk = len - k + 1;
      for (int q = tid; q < 2 * nk; q += XWG) {
        const int pass = q >= nk ? 1 : 0, p = q - pass * nk;
        const uint64_t *b = rbase + pass * S, *nm = rnm + pass * S;
        const uint32_t code = (uint32_t)t1k_get32(b, p) & kmask;
        const bool valid = ((uint32_t)t1k_get32(nm, p) & kmask) == 0;
        uint32_t st = 0, ln = 0, dr = T1K_NO_DIR;
        if (valid) { st = P.ref.kStart[code]; ln = P.ref.kStart[code + 1] - st; }
        if (ln > T1K_DIR_MINLEN && A > X_RANGE) dr = P.ref.kDirIdx[code];
        ukCode[q] = code; ukStart[q] = st; ukLen[q] = ln; ukDir[q] = dr;
      }
      __syncthreads();
      XP(1)
      // The look-up rule (SeqSet.hpp:1098-1153, 1165-1226) is a sequential state machine (prevKmerCode, skipCnt).  Parallel form: if no two
      // k-mers within k/2 + 1 consecutive positions of a strand are equal, `code != prev` holds at every position (prev is the code of one
      // of the previous k/2 + 1 positions), every position is a look-up, and only skipCnt is left: in a maximal run of "big" positions
      // (list >= 100, not the first / last k-mer) exactly every (k/2 + 1)-th one is used, any other position resets the count.  Reads
      // with such short repeats take the sequential replay below.
      {
        const int W1 = k / 2 + 1;
        int lastNonBig[PT];
        uint32_t szv[PT];
        bool bigv[PT], dup = false;
        int localMax = -1;
#pragma unroll
        for (int x = 0; x < PT; ++x) {
          const int q = PT * tid + x;
          szv[x] = 0; bigv[x] = false;
          if (q < 2 * nk) {
            const int pass = q >= nk ? 1 : 0, p = q - pass * nk;
            const uint32_t code = ukCode[q];
            for (int d = 1; d <= W1 && d <= p; ++d) dup |= ukCode[q - d] == code;
            szv[x] = ukLen[q];
            bigv[x] = szv[x] >= 100 && p != 0 && p != nk - 1;
            if (!bigv[x]) localMax = q;
          }
          lastNonBig[x] = localMax;  // within this thread so far; the threads before are merged in below
        }
        if (dup) sMulti = 1;  // (borrowed as the "needs the replay" flag until the chain)
        int incl = localMax;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl = max(incl, y); }
        if ((tid & 63) == 63) sWaveMax[tid >> 6] = incl;
        __syncthreads();
        int before = __shfl_up(incl, 1, 64);
        if ((tid & 63) == 0) before = -1;
        for (int w = 0; w < (tid >> 6); ++w) before = max(before, sWaveMax[w]);
        const bool fallback = sMulti != 0;
        uint32_t mine = 0, minePlus = 0, minePost = 0;
        bool usedv[PT];
#pragma unroll
        for (int x = 0; x < PT; ++x) {
          const int q = PT * tid + x;
          usedv[x] = false;
          if (!fallback && q < 2 * nk && szv[x]) usedv[x] = !bigv[x] || ((q - max(lastNonBig[x], before)) % W1 == 0);
          if (usedv[x]) { ++mine; minePost += szv[x]; if (q < nk) ++minePlus; }
        }
        uint32_t tot;
        uint32_t slot = t1k_block_scan_exclusive(mine, warpSums, &tot);
#pragma unroll
        for (int x = 0; x < PT; ++x)
          if (usedv[x]) usedQ[slot++] = (uint16_t)(PT * tid + x);
        for (int o = 32; o > 0; o >>= 1) { minePlus += __shfl_xor(minePlus, o, 64); minePost += __shfl_xor(minePost, o, 64); }
        if ((tid & 63) == 0 && !fallback) { atomicAdd(&sUsed[0], minePlus); atomicAdd(&sPost, minePost); }
        __syncthreads();
        if (tid == 0 && !fallback) { sUsed[1] = tot - sUsed[0]; stLook += 2 * nk; stPost += sPost; }
      }
      // sequential replay by the first wavefront (uniform code over v_readlane)
      if (sMulti && tid < 64) {
        uint32_t prev = 0;  // prevKmerCode starts at code 0 and is carried from the + strand into the - strand
        uint32_t nUsed = 0, lookups = 0, postings = 0;
        for (int pass = 0; pass < 2; ++pass) {
          int skipCnt = 0;
          const uint32_t begin = nUsed;
          for (int seg = 0; seg < nk; seg += 64) {
            const int pl = seg + tid;
            const uint32_t vc = pl < nk ? ukCode[pass * nk + pl] : 0u;
            const uint32_t vl = pl < nk ? ukLen[pass * nk + pl] : 0u;
            const int cnt = min(64, nk - seg);
            for (int jx = 0; jx < cnt; ++jx) {
              const uint32_t code = (uint32_t)__builtin_amdgcn_readlane((int)vc, jx);
              const uint32_t size = (uint32_t)__builtin_amdgcn_readlane((int)vl, jx);
              const int p = seg + jx;
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
        stLook += lookups; stPost += postings;
      }
      __syncthreads();
      if (tid == 0) sMulti = 0;  // back to its own meaning (set again before it is read: a barrier follows in every path)
      const uint32_t nUsed0 = sUsed[0], nUsed1 = sUsed[1];
      XP(2)
      if (nUsed0 + nUsed1 == 0) continue;  // no hit (1925-1927)
      ++stHist;
      // ---- 3. the fullest (strand, sequence) bucket
      int bestCnt = -1, bestPass = 0;
      uint32_t bestAllele = 0;
      const int needHits = (P.hitLenRequired + k - 1) / k;
      for (int tag = 0; tag < 2; ++tag) {
        const int pass = tag == 0 ? 1 : 0;  // bucket tag 0 = minus strand
        const uint32_t uBeg = pass == 0 ? 0 : nUsed0, uCnt = pass == 0 ? nUsed0 : nUsed1;
        if (!uCnt) continue;
        // sequences the strand's used lists span, and how many hits they hold: a strand with fewer than needHits hits cannot matter
        // (see below), and only the histogram ranges inside the span are walked (the sequences of one gene are neighbours)
        uint32_t rBeg, rEnd;
        {
          uint32_t mn = 0xFFFFFFFFu, mx = 0, tsum = 0;
#pragma unroll
          for (int x = 0; x < LT; ++x) {
            const uint32_t u = LT * tid + x;
            if (u < uCnt) {
              const int q = usedQ[uBeg + u];
              const uint32_t st = ukStart[q], ln = ukLen[q];
              mn = min(mn, P.ref.kPostAllele[st]); mx = max(mx, P.ref.kPostAllele[st + ln - 1]); tsum += ln;
            }
          }
          for (int o = 32; o > 0; o >>= 1) { mn = min(mn, (uint32_t)__shfl_xor(mn, o, 64)); mx = max(mx, (uint32_t)__shfl_xor(mx, o, 64)); tsum += __shfl_xor(tsum, o, 64); }
          if ((tid & 63) == 0) { sRed[tid >> 6] = mn; sRed[4 + (tid >> 6)] = mx; sRed[8 + (tid >> 6)] = tsum; }
          __syncthreads();
          mn = min(min(sRed[0], sRed[1]), min(sRed[2], sRed[3])); mx = max(max(sRed[4], sRed[5]), max(sRed[6], sRed[7]));
          tsum = sRed[8] + sRed[9] + sRed[10] + sRed[11];
          __syncthreads();
          if ((int)tsum < needHits) continue;
          rBeg = A > X_RANGE ? mn / T1K_SEED_CHUNK * T1K_SEED_CHUNK : 0;
          rEnd = mx + 1;
        }
        for (uint32_t r0 = rBeg; r0 < rEnd; r0 += X_RANGE) {
          const uint32_t r1 = min(A, r0 + X_RANGE);
          // slices of the used lists (LT consecutive lists per thread keep the prefix in list order)
          uint32_t myLen[LT];
#pragma unroll
          for (int x = 0; x < LT; ++x) myLen[x] = 0;
#pragma unroll
          for (int x = 0; x < LT; ++x) {
            const uint32_t u = LT * tid + x;
            if (u < uCnt) {
              const int q = usedQ[uBeg + u];
              const uint32_t st = ukStart[q], ln = ukLen[q];
              uint32_t lo = 0, hi = ln;
              if (A > X_RANGE) {
                const uint32_t dr = ukDir[q];
                if (dr != T1K_NO_DIR) {
                  const uint32_t *row = P.ref.kDir + (uint64_t)dr * P.ref.kDirStride;
                  lo = row[r0 / T1K_SEED_CHUNK];
                  hi = r1 >= A ? ln : row[r1 / T1K_SEED_CHUNK];
                } else {
                  lo = listLowerBound(P.ref.kPostAllele + st, ln, r0);
                  hi = r1 >= A ? ln : listLowerBound(P.ref.kPostAllele + st, ln, r1);
                }
              }
              sliceLo[u] = lo;
              myLen[x] = hi - lo;
            }
          }
          uint32_t total, mySum = 0;
#pragma unroll
          for (int x = 0; x < LT; ++x) mySum += myLen[x];
          const uint32_t base = t1k_block_scan_exclusive(mySum, warpSums, &total);
          {
            uint32_t run = base;
#pragma unroll
            for (int x = 0; x < LT; ++x) {
              if (LT * tid + x < (int)uCnt) pre[LT * tid + x] = run;
              run += myLen[x];
            }
          }
          if (tid == 0) pre[uCnt] = total;
          // a bucket holds at most all `total` hits of this range: fewer than ceil(hitLenRequired / k) cannot pass 1959, and cannot outvote
          // a bucket that does (the vote is by count alone)
          if ((int)total < needHits) { __syncthreads(); continue; }
          for (int i = tid; i < X_RANGE; i += XWG) hist[i] = 0;
          __syncthreads();
          {
            uint32_t cur = 0;  // list holding flattened posting x: pre[cur] <= x < pre[cur + 1]; x only grows, so the cursor only advances
            for (uint32_t x0 = tid; x0 < total; x0 += 8 * XWG) {
              uint32_t al[8];
#pragma unroll
              for (int v = 0; v < 8; ++v) {  // eight independent loads in flight
                const uint32_t x = x0 + v * XWG;
                al[v] = 0xFFFFFFFFu;
                if (x < total) {
                  while (pre[cur + 1] <= x) ++cur;
                  const int q = usedQ[uBeg + cur];
                  al[v] = P.ref.kPostAllele[ukStart[q] + sliceLo[cur] + (x - pre[cur])];
                }
              }
#pragma unroll
              for (int v = 0; v < 8; ++v)
                if (al[v] != 0xFFFFFFFFu) atomicAdd(&hist[al[v] - r0], 1u);
            }
          }
          __syncthreads();
          // first maximum of the range: largest count, then smallest sequence index
          unsigned long long key = 0;
          for (int i = tid; i < X_RANGE; i += XWG) {
            const uint32_t c = hist[i];
            if (c) { const unsigned long long kx = ((unsigned long long)c << 32) | (0xFFFFFFFFu - (r0 + (uint32_t)i)); if (kx > key) key = kx; }
          }
          for (int o = 32; o > 0; o >>= 1) { const unsigned long long y = __shfl_xor(key, o, 64); if (y > key) key = y; }
          if ((tid & 63) == 0) sKey[tid >> 6] = key;
          __syncthreads();
          key = max(max(sKey[0], sKey[1]), max(sKey[2], sKey[3]));
          const int c = (int)(key >> 32);
          if (c > 0 && c > bestCnt) { bestCnt = c; bestPass = pass; bestAllele = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu); }
          __syncthreads();  // hist / pre / sKey are rewritten by the next pass
        }
      }
      XP(3)
      if (bestCnt < 0 || k * bestCnt < P.hitLenRequired) continue;  // 1959-1964
      ++stChain;
      // ---- 4. the bucket's hits -> H (hist is dead)
      uint32_t *H = GH ? P.ghScratch + (size_t)blockIdx.x * 4 * X_HCAP : hist, *SA = H + X_HCAP, *SB = H + 2 * (size_t)X_HCAP, *SC = H + 3 * (size_t)X_HCAP;
      const uint32_t uBeg = bestPass == 0 ? 0 : nUsed0, uCnt = bestPass == 0 ? nUsed0 : nUsed1;
      uint32_t n;
      {
        uint32_t myLo[LT], myLen[LT], mySum = 0;
#pragma unroll
        for (int x = 0; x < LT; ++x) { myLo[x] = 0; myLen[x] = 0; }
#pragma unroll
        for (int x = 0; x < LT; ++x) {
          const uint32_t u = LT * tid + x;
          if (u < uCnt) {
            const int q = usedQ[uBeg + u];
            const uint32_t *pl = P.ref.kPostAllele + ukStart[q];
            const uint32_t ln = ukLen[q];
            const uint32_t lo = listLowerBound(pl, ln, bestAllele);
            uint32_t hi = lo;
            while (hi < ln && pl[hi] == bestAllele) ++hi;
            myLo[x] = lo; myLen[x] = hi - lo;
          }
        }
#pragma unroll
        for (int x = 0; x < LT; ++x) mySum += myLen[x];
        const uint32_t base = t1k_block_scan_exclusive(mySum, warpSums, &n);
        if (n <= X_HCAP) {
          uint32_t w = base;
#pragma unroll
          for (int x = 0; x < LT; ++x) {
            const uint32_t u = LT * tid + x;
            if (u < uCnt) {
              const int q = usedQ[uBeg + u];
              const uint32_t a = (uint32_t)(q - bestPass * nk);
              const T1kPosting *pl = P.ref.kPost + ukStart[q] + myLo[x];
              for (uint32_t i = 0; i < myLen[x]; ++i) H[w++] = a | (pl[i].offset << 12);
            }
          }
        }
      }
      if (n > X_HCAP) { if (tid == 0) atomicOr(P.err, (unsigned long long)XERR_HITCAP); continue; }
      __syncthreads();
      XP(4)
      // one diagonal, read offsets ascending (the gather order): one run, nothing filtered, LIS = every hit (see the header)
      {
        const int d0 = (int)(H[0] & 0xFFF) - (int)(H[0] >> 12);
        int part = 0;
        bool multi = false;
        for (uint32_t i = tid; i < n; i += XWG) {
          const uint32_t x = H[i];
          if ((int)(x & 0xFFF) - (int)(x >> 12) != d0) multi = true;
          if (i + 1 < n) {
            const int gap = (int)(H[i + 1] & 0xFFF) - (int)(x & 0xFFF);
            if (gap <= 0) multi = true;
            part += gap < k ? gap : k;
          } else part += k;
        }
        if (multi) sMulti = 1;
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if ((tid & 63) == 0) atomicAdd(&sLen, part);
      }
      __syncthreads();
      XP(5)
      if (!sMulti) {
        const int hitLen = sLen;  // on the read and on the sequence alike
        const int thr = (int)(len * P.oneMinusSim) * k;
        fragGood = n >= 3 && (int)n * k >= P.hitLenRequired && hitLen >= P.hitLenRequired && len - hitLen <= thr;
        continue;
      }
      // rank sort by (diagonal, sequence offset, read offset): the packed words are distinct
      for (uint32_t i = tid; i < n; i += XWG) {
        const uint32_t x = H[i];
        uint32_t r = 0;
        for (uint32_t jx = 0; jx < n; ++jx) r += hitKeyLess(H[jx], x) ? 1u : 0u;
        SA[r] = x;
      }
      __syncthreads();
      if (tid == 0) {
        int *minDist = (int *)sliceLo;  // per read offset (readOffsetUsed, 1246, 1437-1456)
        const int thr = (int)(len * P.oneMinusSim) * k;  // mismatchThreshold (1974)
        bool ok = false;
        auto diagOf = [](uint32_t x) { return (int)(x & 0xFFF) - (int)(x >> 12); };
        for (int s = 0; s < (int)n && !ok;) {
          int curDiff = diagOf(SA[s]), curCnt = 1, domCnt = 0, dominant = 0;
          int e = s + 1;
          for (; e < (int)n; ++e) {
            int d = diagOf(SA[e]) - diagOf(SA[e - 1]);
            if (d < 0) d = -d;
            if (d > P.radius) break;
            if (d == 0) ++curCnt;
            else {
              if (curCnt > domCnt) { dominant = curDiff; domCnt = curCnt; }
              curDiff = diagOf(SA[e]); curCnt = 1;
            }
          }
          if (curCnt > domCnt) dominant = curDiff;
          if (e - s < 3 || (e - s) * k < P.hitLenRequired) { s = e; continue; }
          for (int q = s; q < e; ++q) minDist[SA[q] & 0xFFF] = 0x7FFFFFFF;
          for (int q = s; q < e; ++q) {
            int dq = diagOf(SA[q]) - dominant; if (dq < 0) dq = -dq;
            int *md = &minDist[SA[q] & 0xFFF];
            if (dq < *md) *md = dq;
          }
          int m = 0;
          for (int q = s; q < e; ++q) {
            const uint32_t x = SA[q];
            int dq = diagOf(x) - dominant; if (dq < 0) dq = -dq;
            if (dq != minDist[x & 0xFFF]) continue;
            int jx = m - 1;  // insertion by (sequence offset, read offset) == packed value order (CompSortPairBInc)
            while (jx >= 0 && x < SB[jx]) { SB[jx + 1] = SB[jx]; --jx; }
            SB[jx + 1] = x;
            ++m;
          }
          int ret, lenR, lenS;
          if (t1k_run_lis(SA, SB, SC, s, m, k, P.hitLenRequired, &ret, &lenR, &lenS)) {
            if (len - lenR <= thr) ok = true;  // matchCnt = 2 * hitLen (1536); len - matchCnt / 2 <= mismatchThreshold (1978)
          }
          s = e;
        }
        sRes = ok ? 1 : 0;
      }
      __syncthreads();
      XP(6)
      fragGood = sRes != 0;
    }
    if (tid == 0) P.good[f] = fragGood ? 1 : 0;
   }
  }
  if (tid == 0 && P.stats) {
    atomicAdd(&P.stats[1], stLook); atomicAdd(&P.stats[2], stPost);
    atomicAdd(&P.stats[3], stHist); atomicAdd(&P.stats[4], stChain);
#ifdef T1K_XPROF
    for (int i = 0; i < 7; ++i) atomicAdd(&P.stats[7 + i], (unsigned long long)xp_[i]);
#endif
  }
}

size_t t1k_extract_lds_bytes(int maxK, int range) { return ((size_t)maxK * 6 + 3) * 4 + ((size_t)maxK + 1) / 2 * 2 * 2 + (size_t)range * 4 + 16; }

static ExtractArgs extractArgs(const T1kRefDev &ref, const T1kReadsDev &reads, int k, int radius, int hitLenRequired, double oneMinusSim, uint32_t nFragments,
                               uint32_t epf, uint32_t maxK, uint8_t *good, uint8_t *state, unsigned long long *err, unsigned long long *stats) {
  ExtractArgs a{};
  a.ref = ref; a.reads = reads; a.k = k; a.radius = radius; a.hitLenRequired = hitLenRequired; a.oneMinusSim = oneMinusSim;
  a.nFragments = nFragments; a.epf = epf; a.maxK = maxK; a.good = good; a.state = state; a.err = err; a.stats = stats;
  return a;
}

void t1k_launch_extract(t1k_ctx *ctx, const T1kRefDev &ref, const T1kReadsDev &reads, int k, int radius, int hitLenRequired, double oneMinusSim, uint32_t nFragments,
                        uint32_t epf, uint32_t maxK, uint8_t *good, uint8_t *state, unsigned long long *err, unsigned long long *stats, int nWg) {
  const ExtractArgs a = extractArgs(ref, reads, k, radius, hitLenRequired, oneMinusSim, nFragments, epf, maxK, good, state, err, stats);
  const size_t ldsBytes = t1k_extract_lds_bytes((int)maxK, X_RANGE_SMALL);
  const bool xl = ctx->batchMaxLen > T1K_MAX_READ_LEN;  // the batch holds a read beyond the production shape
  if (xl) hipFuncSetAttribute((const void *)k_extract<X_RANGE_SMALL, X_PT_LONG, X_LT_LONG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
  else hipFuncSetAttribute((const void *)k_extract<X_RANGE_SMALL, X_PT_FAST, X_LT_FAST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
  const unsigned gridS = (unsigned)std::min<uint64_t>(((uint64_t)reads.nReadEnds + 3) / 4, (uint64_t)nWg);
  (void)hipEventRecord(ctx->ev[0], ctx->stream);
  if (xl) hipLaunchKernelGGL(k_extract_screen<X_PS_LONG>, dim3(gridS), dim3(XWG), 0, ctx->stream, a);
  else hipLaunchKernelGGL(k_extract_screen<X_PS_FAST>, dim3(gridS), dim3(XWG), 0, ctx->stream, a);
  (void)hipEventRecord(ctx->ev[1], ctx->stream);
  const unsigned grid = (unsigned)std::min<uint64_t>(((uint64_t)nFragments + XWG - 1) / XWG, (uint64_t)nWg);
  if (xl) hipLaunchKernelGGL((k_extract<X_RANGE_SMALL, X_PT_LONG, X_LT_LONG>), dim3(grid), dim3(XWG), ldsBytes, ctx->stream, a);
  else hipLaunchKernelGGL((k_extract<X_RANGE_SMALL, X_PT_FAST, X_LT_FAST>), dim3(grid), dim3(XWG), ldsBytes, ctx->stream, a);
  (void)hipEventRecord(ctx->ev[2], ctx->stream);
}

// third attempt: a bucket beyond the large LDS shape; hit arrays in HBM (scratch: grid x 4 x cap words)
void t1k_launch_extract_huge(t1k_ctx *ctx, const T1kRefDev &ref, const T1kReadsDev &reads, int k, int radius, int hitLenRequired, double oneMinusSim,
                             uint32_t nFragments, uint32_t epf, uint32_t maxK, uint8_t *good, uint8_t *state, unsigned long long *err, unsigned long long *stats, int nWg,
                             uint32_t *scratch, uint32_t cap) {
  ExtractArgs a = extractArgs(ref, reads, k, radius, hitLenRequired, oneMinusSim, nFragments, epf, maxK, good, state, err, stats);
  a.ghScratch = scratch; a.ghCap = cap;
  const bool xl = ctx->batchMaxLen > T1K_MAX_READ_LEN;
  const size_t ldsBytes = t1k_extract_lds_bytes((int)maxK, X_RANGE_SMALL);
  const unsigned grid = (unsigned)std::min<uint64_t>(((uint64_t)nFragments + XWG - 1) / XWG, (uint64_t)nWg);
  if (xl) {
    hipFuncSetAttribute((const void *)k_extract<X_RANGE_SMALL, X_PT_LONG, X_LT_LONG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
    hipLaunchKernelGGL((k_extract<X_RANGE_SMALL, X_PT_LONG, X_LT_LONG, true>), dim3(grid), dim3(XWG), ldsBytes, ctx->stream, a);
  } else {
    hipFuncSetAttribute((const void *)k_extract<X_RANGE_SMALL, X_PT_FAST, X_LT_FAST, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
    hipLaunchKernelGGL((k_extract<X_RANGE_SMALL, X_PT_FAST, X_LT_FAST, true>), dim3(grid), dim3(XWG), ldsBytes, ctx->stream, a);
  }
  (void)hipEventRecord(ctx->ev[2], ctx->stream);
}

// second attempt of a batch in which a bucket overflowed the production shape (the screen's verdicts are still in `state`)
void t1k_launch_extract_big(t1k_ctx *ctx, const T1kRefDev &ref, const T1kReadsDev &reads, int k, int radius, int hitLenRequired, double oneMinusSim,
                            uint32_t nFragments, uint32_t epf, uint32_t maxK, uint8_t *good, uint8_t *state, unsigned long long *err, unsigned long long *stats, int nWg) {
  const ExtractArgs a = extractArgs(ref, reads, k, radius, hitLenRequired, oneMinusSim, nFragments, epf, maxK, good, state, err, stats);
  const bool xl = ctx->batchMaxLen > T1K_MAX_READ_LEN;
  const size_t ldsBytes = t1k_extract_lds_bytes((int)maxK, xl ? X_RANGE_BIG_LONG : X_RANGE_BIG);
  const unsigned grid = (unsigned)std::min<uint64_t>(((uint64_t)nFragments + XWG - 1) / XWG, (uint64_t)nWg);
  if (xl) {
    hipFuncSetAttribute((const void *)k_extract<X_RANGE_BIG_LONG, X_PT_LONG, X_LT_LONG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
    hipLaunchKernelGGL((k_extract<X_RANGE_BIG_LONG, X_PT_LONG, X_LT_LONG>), dim3(grid), dim3(XWG), ldsBytes, ctx->stream, a);
  } else {
    hipFuncSetAttribute((const void *)k_extract<X_RANGE_BIG, X_PT_FAST, X_LT_FAST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
    hipLaunchKernelGGL((k_extract<X_RANGE_BIG, X_PT_FAST, X_LT_FAST>), dim3(grid), dim3(XWG), ldsBytes, ctx->stream, a);
  }
  (void)hipEventRecord(ctx->ev[2], ctx->stream);
}
