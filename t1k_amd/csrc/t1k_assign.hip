// t1k_amd/csrc/t1k_assign.hip -- gfx950 kernels for SeqSet::AssignRead (reference SeqSet.hpp:2119-2303) over a batch
// of read-ends resident in HBM.  Stages (one kernel each, all integer / HBM- and LDS-bound, no MFMA):
//
//   k_pack_reads   ASCII -> 2-bit words + N masks, forward and reverse complement           (SeqSet.hpp:2103-2114)
//   (seeding / hit grouping / chaining / strand vote: t1k_chain.hip)
//   k_extend       one lane per candidate: similarity / low-complexity filter (1838-1845, 458-485, 1894-1908),
//                  separator tests (2163-2169), ExtendOverlap (1994-2100)
//   k_select       one workgroup per read-end: sort by _overlap::operator< (103-127), the onlyConsiderClip latch
//                  (2156-2186) evaluated in parallel, near-best flags (2192-2200)
//   k_fullalign    one lane per kept overlap: near-best full alignment -> relaxedMatchCnt + base coverage as a
//                  difference array (2188-2285); alignments that need a real DP traceback go to a queue (k_fullalign_slow)
//   k_truncate     >1000 overlaps: re-sort and cut at similarity < best - 0.1 (2290-2298)
#include "t1k_dev.h"
#include "t1k_launch.h"
#include "t1k_memo.h"

#define WG 256
#define GA_BIG_MAX 2048
#define GA_SCRATCH_INTS (6 * (GA_BIG_MAX + 4))  // row arrays of t1k_ga_general for sequences up to 2048 bases

enum { ERR_HITCAP = 1, ERR_STAGECAP = 2, ERR_CANDCAP = 4, ERR_BIGGROUP = 8, ERR_OVLCAP = 16, ERR_SORTCAP = 32, ERR_SLOWCAP = 64, ERR_ROWCAP = 128, ERR_GROUPCAP = 256 };

// ------------------------------------------------------------------------------------------------------------------
// pack
// ------------------------------------------------------------------------------------------------------------------
// nCode: the base bits stored under an N (t1k_params::n_base_code); every consumer of the bases other than the k-mer codes masks N positions out
__global__ void k_pack_reads(const char *ascii, const uint64_t *offs, uint32_t n, int S, uint64_t *bases, uint64_t *nmask, uint16_t *lens, int nCode) {
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
    if (code == 4) { fn |= 1ull << (2 * q); fb |= (uint64_t)nCode << (2 * q); } else fb |= (uint64_t)code << (2 * q);
    char d = ascii[o + len - 1 - i];
    int dc = d == 'A' ? 3 : d == 'C' ? 2 : d == 'G' ? 1 : d == 'T' ? 0 : 4;
    if (dc == 4) { rn |= 1ull << (2 * q); rb |= (uint64_t)nCode << (2 * q); } else rb |= (uint64_t)dc << (2 * q);
  }
  uint64_t base = (uint64_t)re * 2 * S;
  bases[base + w] = fb; nmask[base + w] = fn;
  bases[base + S + w] = rb; nmask[base + S + w] = rn;
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


// DEFER = true: the two extension alignments that need a DP are registered in the read-end's memo (the same table the chain
// stage filled: thousands of alleles share a window) and the candidate is put on the retry list; returns false.
// DEFER = false: second pass, after k_dp_dense: the memo answers (or, for the few unregistered ones, the DP runs here).
template <bool DEFER>
__device__ __forceinline__ bool extendOne(const ExtendArgs &P, uint64_t gid, unsigned int &extendDp) {
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
  // (the similarity / low-complexity filter of SeqSet.hpp:1838-1845, 1894-1908 has been applied by k_collect: only survivors are here)
  uint16_t flags = 0;
  if (sepInRange(P.ref, allele, ss, se)) flags |= T1K_F_SEPSEED;                                  // 2163
  if (sepInRange(P.ref, allele, ss - rs, se + (len - re - 1))) flags |= T1K_F_NEEDCLIP;           // 2167-2169
  if (flags & T1K_F_SEPSEED) { x.flags = flags; P.ext[gid] = x; return true; }
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
  const ReadCtx rc{rb, rn, len, P.ref.bases, P.ref.nmask, goff, alleleLen, P.ref.anyN != 0};
  const GapSink sink{P.memo + (uint64_t)c.re * GAP_CACHE, P.jobStr, P.counters, c.re * GAP_CACHE, P.jobSegCap, T1K_AR_EXTJOBS};
  uint32_t slot = 0;
  bool pending = false;
  int match = gapMatchesCached<DEFER>(rc, rs - lo, goff + ss - lo, lo, lo, pass, sink, &dpLocal, &slot);
  if (match < 0) { pending = true; match = 0; }
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
  {
    const int mr = gapMatchesCached<DEFER>(rc, re + 1, goff + se + 1, ro, ro, pass, sink, &dpLocal, &slot);
    if (mr < 0) pending = true; else match += mr;
  }
  extendDp = dpLocal;
  if (pending) return false;
  int eMatch = 2 * match + matchCnt;
  int ers = rs - lo, ere = re + ro, ess = ss - lo, ese = se + ro;
  double esim = (double)eMatch / (double)(ere - ers + 1 + ese - ess + 1);
  if (!(esim < P.sim)) flags |= T1K_F_EXTOK;                          // 2074 (before clip credit)
  if (leftClip > 0 || rightClip > 0) eMatch += 2 * leftClip + 2 * rightClip;  // 2078-2087
  x.seqStart = ess; x.seqEnd = ese; x.readStart = (uint16_t)ers; x.readEnd = (uint16_t)ere;
  x.matchCnt = (uint16_t)eMatch; x.leftClip = (uint16_t)leftClip; x.rightClip = (uint16_t)rightClip; x.flags = flags;
  P.ext[gid] = x;
  return true;
}

#if T1K_EXTEND_WAVES > 0
#define T1K_EXTEND_ATTR T1K_WAVES_ATTR_(T1K_EXTEND_WAVES)
#else
#define T1K_EXTEND_ATTR
#endif
__global__ __launch_bounds__(WG) T1K_EXTEND_ATTR void k_extend(ExtendArgs P) {
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned int dp = 0;
  bool done = true;
  if (gid < P.nCand) done = extendOne<true>(P, gid, dp);
  if (!done) {
    const uint32_t q = t1k_arena_append(P.counters, T1K_AR_EXTRETRY, P.retrySegCap);
    if (q != T1K_ARENA_FULL) P.retryStr[q] = (uint32_t)gid;
  }
  t1k_stat_add(P.counters, T1K_STAT_EXTEND_DP, dp);
}

// second pass over the candidates whose extension waited for registered alignments
__global__ __launch_bounds__(WG) void k_extend_retry(ExtendArgs P, const uint32_t *list, uint32_t nItems, const unsigned long long *nDev) {
  if (nDev) nItems = (uint32_t)*nDev;  // (the total word k_arena_compact left: no counter fetch between k_extend and this launch)
  unsigned int dp = 0;
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nItems; q += gridDim.x * blockDim.x) {
    unsigned int d1 = 0;
    extendOne<false>(P, list[q], d1);
    dp += d1;
  }
  t1k_stat_add(P.counters, T1K_STAT_EXTEND_DP, dp);
}

// ------------------------------------------------------------------------------------------------------------------
// selection: sort + latch, one workgroup per read-end
// ------------------------------------------------------------------------------------------------------------------

// bitonic sort of np2 (a power of two >= 8) packed 64-bit entries (sort key in the high bits, payload index in the low bits), padded
// by the caller with ~0.  Blocked layout: lane t of the first np2/8 threads keeps entries [8t, 8t + 8) in registers, so the
// compare-exchange partners are in the same lane for strides < 8, in the same wavefront for strides < 512 (exchanged with
// ds_bpermute shuffles, no barrier) and only the strides >= 512 go through LDS with workgroup barriers: 10 of the 91 steps at
// 8192 entries.  All threads of the workgroup must call it.
__device__ __forceinline__ uint64_t shflXor64(uint64_t v, int laneMask) {
  const int lo = __shfl_xor((int)(uint32_t)v, laneMask, 64), hi = __shfl_xor((int)(uint32_t)(v >> 32), laneMask, 64);
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
// plain version (any size, entries in LDS or HBM): one barrier per step
__device__ inline void bitonicSortSimple(uint64_t *key, uint32_t np2) {
  __syncthreads();
  for (uint32_t size = 2; size <= np2; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      for (uint32_t t = threadIdx.x; t < np2 / 2; t += blockDim.x) {
        uint32_t lo = 2 * t - (t & (stride - 1));
        uint32_t hi = lo + stride;
        bool up = (lo & size) == 0;
        uint64_t a = key[lo], b = key[hi];
        bool sw = up ? (a > b) : (a < b);
        if (sw) { key[lo] = b; key[hi] = a; }
      }
      __syncthreads();
    }
  }
}
__device__ inline void bitonicSort(uint64_t *key, uint32_t np2) {
  constexpr int E = 8;
  if (np2 < E || np2 > blockDim.x * E) { bitonicSortSimple(key, np2); return; }
  const uint32_t t = threadIdx.x;
  const bool act = t * E < np2;
  uint64_t a[E];
  __syncthreads();
  if (act) {
#pragma unroll
    for (int e = 0; e < E; ++e) a[e] = key[t * E + e];
  }
  for (uint32_t size = 2; size <= np2; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      if (stride >= 64 * E) {  // partner in another wavefront: through LDS
        if (act) {
#pragma unroll
          for (int e = 0; e < E; ++e) key[t * E + e] = a[e];
        }
        __syncthreads();
        if (act) {
          const uint32_t pt = t ^ (stride / E);
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const uint32_t i = t * E + e;
            const uint64_t b = key[pt * E + e];
            const bool up = (i & size) == 0, lower = (i & stride) == 0;
            const bool keepMin = up == lower;
            a[e] = keepMin ? (a[e] < b ? a[e] : b) : (a[e] > b ? a[e] : b);
          }
        }
        __syncthreads();
      } else if (stride >= E) {  // partner lane in the same wavefront
        if (act) {
          const int lm = (int)(stride / E);
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const uint32_t i = t * E + e;
            const uint64_t b = shflXor64(a[e], lm);
            const bool up = (i & size) == 0, lower = (i & stride) == 0;
            const bool keepMin = up == lower;
            a[e] = keepMin ? (a[e] < b ? a[e] : b) : (a[e] > b ? a[e] : b);
          }
        }
      } else if (act) {  // both entries in this lane
#pragma unroll
        for (int s = E / 2; s > 0; s >>= 1) {
          if (stride == (uint32_t)s) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
              if ((e & s) == 0) {
                const uint32_t i = t * E + e;
                const bool up = (i & size) == 0;
                const uint64_t x = a[e], y = a[e + s];
                const bool sw = up ? (x > y) : (x < y);
                a[e] = sw ? y : x; a[e + s] = sw ? x : y;
              }
            }
          }
        }
      }
    }
  }
  if (act) {
#pragma unroll
    for (int e = 0; e < E; ++e) key[t * E + e] = a[e];
  }
  __syncthreads();
}
// ------------------------------------------------------------------------------------------------------------------
// Stable counting sort in LDS (replaces the bitonic network for the common case).
//
// The lists k_select and k_truncate order hold a few thousand entries but only a handful of distinct (matchCnt, spans) classes,
// and k_select's input already arrives in allele order -- so ordering it is ONE stable pass over the dense rank of the class;
// k_truncate's input needs the allele order first: two (three) more stable passes over the allele's bytes.
//   ldsClassRanks     distinct classes of the list -> dense ranks 0 .. K - 1 in class order (LDS hash table, K <= 256)
//   ldsCountingPass   one stable pass by an 8-bit digit: a wavefront takes a contiguous piece of the list, 64 entries a round;
//                     lanes with the same digit find each other with ballots (their order inside the round is their lane
//                     order), the first of them advances the piece's counter of that digit.  The counters, laid out
//                     [digit][wavefront], are then prefix-summed: entry -> position.  Four workgroup barriers a pass.
// Both must be called by all NT threads.  Entries stay 64-bit packed keys; the caller falls back to the bitonic network when a
// list has more than 256 classes.
// ------------------------------------------------------------------------------------------------------------------
#define CS_SLOTS 512
struct CsTables {
  uint32_t hKey[CS_SLOTS];   // class + 1 (0 = empty)
  uint8_t hRank[CS_SLOTS];
  uint32_t dList[256];
  uint32_t nDistinct, overflow;
};
__device__ __forceinline__ uint32_t csHash(uint32_t c) { return (c * 2654435761u) >> 23; }  // 9 bits
__device__ __forceinline__ uint32_t csRankOf(const CsTables &T, uint32_t cls) {
  const uint32_t want = cls + 1;
  uint32_t h = csHash(cls);
  while (T.hKey[h] != want) h = (h + 1) & (CS_SLOTS - 1);
  return T.hRank[h];
}
template <int NT>
__device__ inline bool ldsClassRanks(const uint64_t *key, uint32_t n, int classShift, CsTables &T) {
  const int tid = threadIdx.x;
  for (int q = tid; q < CS_SLOTS; q += NT) T.hKey[q] = 0;
  if (tid == 0) { T.nDistinct = 0; T.overflow = 0; }
  __syncthreads();
  for (uint32_t i = tid; i < n; i += NT) {
    const uint32_t cls = (uint32_t)(key[i] >> classShift), want = cls + 1;
    uint32_t h = csHash(cls);
    for (int probe = 0; probe < CS_SLOTS; ++probe) {
      const uint32_t cur = T.hKey[h];
      if (cur == want) break;
      if (cur == 0) {
        if (T.overflow) break;
        const uint32_t old = atomicCAS(&T.hKey[h], 0u, want);
        if (old == 0) { if (atomicAdd(&T.nDistinct, 1u) >= 256u) T.overflow = 1; break; }
        if (old == want) break;
      }
      h = (h + 1) & (CS_SLOTS - 1);
    }
  }
  __syncthreads();
  if (T.overflow || T.nDistinct > 256u) { __syncthreads(); return false; }
  if (tid == 0) T.nDistinct = 0;
  __syncthreads();
  for (int q = tid; q < CS_SLOTS; q += NT)
    if (T.hKey[q]) T.dList[atomicAdd(&T.nDistinct, 1u)] = T.hKey[q];
  __syncthreads();
  const uint32_t K = T.nDistinct;
  for (int q = tid; q < CS_SLOTS; q += NT) {
    const uint32_t mine = T.hKey[q];
    if (!mine) continue;
    uint32_t r = 0;
    for (uint32_t j = 0; j < K; ++j) r += T.dList[j] < mine ? 1u : 0u;
    T.hRank[q] = (uint8_t)r;
  }
  __syncthreads();
  return true;
}

// cnt: 256 * (NT / 64) u16 of LDS.  n <= 512 * (NT / 64).
template <int NT, class DigitFn>
__device__ inline void ldsCountingPass(uint64_t *key, uint32_t n, uint16_t *cnt, uint32_t *warpSums, DigitFn digit) {
  constexpr int NW = NT / 64, ROUNDS = 8;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  {
    uint32_t *c32 = (uint32_t *)cnt;  // 256 * NW / 2 words, two per thread
    c32[2 * tid] = 0; c32[2 * tid + 1] = 0;
  }
  __syncthreads();
  const uint32_t seg = (((n + NW - 1) / NW) + 63u) & ~63u;  // entries of one wavefront's piece (<= 512)
  uint64_t kk[ROUNDS];
  uint32_t at[ROUNDS];  // digit << 16 | position inside the piece's run of that digit (0xFFFFFFFF: no entry)
  const uint64_t ltMask = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
  for (int q = 0; q < ROUNDS; ++q) {
    kk[q] = 0; at[q] = 0xFFFFFFFFu;
    if ((uint32_t)q * 64u < seg) {
      const uint32_t i = (uint32_t)w * seg + (uint32_t)q * 64u + (uint32_t)lane;
      const bool valid = i < n;
      uint32_t r = 0;
      if (valid) { kk[q] = key[i]; r = digit(kk[q]) & 0xFFu; }
      uint64_t m = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const uint64_t bal = __ballot(valid && ((r >> b) & 1u));
        m &= ((r >> b) & 1u) ? bal : ~bal;
      }
      // m: the valid lanes of this round that hold digit r (for a valid lane it contains the lane itself)
      const int leader = valid ? (__ffsll((long long)m) - 1) : lane;
      uint32_t base = 0;
      if (valid && lane == leader) {
        uint16_t *c = cnt + r * NW + w;
        base = *c;
        *c = (uint16_t)(base + (uint32_t)__popcll(m));
      }
      base = (uint32_t)__shfl((int)base, leader, 64);
      if (valid) at[q] = (r << 16) | (base + (uint32_t)__popcll(m & ltMask));
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  {  // exclusive prefix over cnt in [digit][wavefront] order: four consecutive counters per thread
    uint16_t *c = cnt + 4 * tid;
    const uint32_t a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
    uint32_t tot;
    const uint32_t ex = t1k_block_scan_exclusive_n<NW>(a0 + a1 + a2 + a3, warpSums, &tot);
    c[0] = (uint16_t)ex; c[1] = (uint16_t)(ex + a0); c[2] = (uint16_t)(ex + a0 + a1); c[3] = (uint16_t)(ex + a0 + a1 + a2);
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < ROUNDS; ++q) {
    if ((uint32_t)q * 64u < seg) {
      if (at[q] != 0xFFFFFFFFu) key[(uint32_t)cnt[(at[q] >> 16) * NW + w] + (at[q] & 0xFFFFu)] = kk[q];
    }
  }
  __syncthreads();
}

// Field widths of the sort keys' class (match count, span sum, read span).  A read-end's list is sorted on its own, so the layout may
// differ from read-end to read-end: the fast windows use fixed widths (compile-time constants: reads of at most T1K_MAX_READ_LEN bases);
// in a window that holds longer reads (XL) the widths follow the read-end's length -- a 1000-base read needs 11 / 11 / 10 bits, which
// leaves the allele and the index 28 bits (8192 candidates against a reference of 32 768 sequences; beyond: ERR_SORTCAP, loudly), while
// the short read-ends of the same window keep the room they have elsewhere (a 150-base read-end of the HLA-like benchmark reference
// can hold more than 8192 candidates).
struct KeyW { int MB, DB, RB; };
__device__ __forceinline__ int bitLen(uint32_t v) { return 32 - __clz((int)v); }  // v >= 1
template <bool XL>
__device__ __forceinline__ KeyW selectWidths(int len) {  // k_select: seed match counts <= 2 len, span sums <= 2 len + indel slack
  if (!XL || len <= T1K_MAX_READ_LEN) return KeyW{10, 10, 9};  // (the fast windows' layout for every read-end they could hold)
  return KeyW{bitLen((uint32_t)(2 * len)), bitLen((uint32_t)(2 * len + 40)), bitLen((uint32_t)len)};
}
template <bool XL>
__device__ __forceinline__ KeyW truncWidths(int len) {   // k_truncate: extended match counts incl. clip credit, span sums incl. clips
  if (!XL || len <= T1K_MAX_READ_LEN) return KeyW{11, 11, 9};
  return KeyW{bitLen((uint32_t)(2 * len)), bitLen((uint32_t)(2 * len + 40)), bitLen((uint32_t)len)};
}

// k_truncate's entry: [0.. | MMAX - matchCnt : MB | span sum : DB | RMAX - read span : RB | allele : aBits | index : iBits] (fast windows:
// 11 / 11 / 9 and bit 63 clear).  Returns false if a field does not fit.
__device__ __forceinline__ bool packSortKey(const KeyW &W, int m, int d, int rspan, uint32_t allele, uint32_t index, int aBits, int iBits, uint64_t *out) {
  const int MMAX = (1 << W.MB) - 1, DMAX = (1 << W.DB) - 1, RMAX = (1 << W.RB) - 1;
  if (m < 0 || m > MMAX || d < 0 || d > DMAX || rspan < 0 || rspan > RMAX || 1 + W.MB + W.DB + W.RB + aBits + iBits > 64) return false;
  *out = (((((uint64_t)(MMAX - m) << W.DB | (uint64_t)d) << W.RB | (uint64_t)(RMAX - rspan)) << aBits | (uint64_t)allele) << iBits) | (uint64_t)index;
  return true;
}

// k_select's entry: [emit mark : 1 | MMAX - seed matchCnt : MB | span sum : DB | RMAX - read span : RB | allele : aBits | index : iBits |
// flags : 3] (fast windows: 10 / 10 / 9); the flags (separator in the seed, extension passed, "needs clipping and similarity >= 0.95")
// are all the later phases need to know about a candidate, so they never go back to HBM for it.  The index is unique, so the flags never order.
#define SEL_F_SEPSEED 1u
#define SEL_F_EXTOK 2u
#define SEL_F_KEEPCLIP 4u
__device__ __forceinline__ bool packSelectKey(const KeyW &W, int m, int d, int rspan, uint32_t allele, uint32_t index, uint32_t flags, int aBits, int iBits, uint64_t *out) {
  const int MMAX = (1 << W.MB) - 1, DMAX = (1 << W.DB) - 1, RMAX = (1 << W.RB) - 1;
  if (m < 0 || m > MMAX || d < 0 || d > DMAX || rspan < 0 || rspan > RMAX || 1 + W.MB + W.DB + W.RB + 3 + aBits + iBits > 64) return false;
  *out = ((((((uint64_t)(MMAX - m) << W.DB | (uint64_t)d) << W.RB | (uint64_t)(RMAX - rspan)) << aBits | (uint64_t)allele) << iBits) | (uint64_t)index) << 3 | flags;
  return true;
}

// full comparator on the seed coordinates for the (rare) ties of the 64-bit key
__device__ inline bool candBeforeFull(const T1kCand &a, const T1kCand &b) {
  int ars = a.readSE & 0xFFFF, are = a.readSE >> 16, brs = b.readSE & 0xFFFF, bre = b.readSE >> 16;
  if (ars != brs) return ars < brs;
  if (are != bre) return are < bre;
  if (a.seqStart != b.seqStart) return a.seqStart < b.seqStart;
  return a.seqEnd < b.seqEnd;
}

// The per-read-end sorts run in LDS.  Two instantiations share the read-ends by size so that the common small lists do not
// pay the occupancy of the rare large ones: CAP 2048 (16 KB of LDS) takes lists of up to 2048 entries, CAP 8192 (64 KB,
// 1024 threads) the rest; beyond 8192 the sort falls back to per-workgroup HBM scratch.
#define SELECT_SMALL 2048
#define SELECT_LARGE 8192

// (registers: the 1024-thread shape's 76 KB of LDS admit two workgroups a compute unit = eight wavefronts per SIMD, the 256-thread shape's 22 KB seven
// workgroups = SEVEN wavefronts: held to eight wavefronts' worth of registers it spilled 13 of them for an occupancy its LDS does not admit -- round 6,
// as for k_seed_groups; T1K_SELECT_SMALL_WAVES)
#ifndef T1K_SELECT_SMALL_WAVES
#define T1K_SELECT_SMALL_WAVES 7
#endif
template <int SELECT_LDS_CAP, int NT, bool XL>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(SELECT_LDS_CAP == SELECT_SMALL ? T1K_SELECT_SMALL_WAVES : 8))) void k_select(SelectArgs P) {
  extern __shared__ uint64_t dynLds[];
  uint64_t *sKey = dynLds;
  __shared__ uint32_t warpSums[NT / 64];
  __shared__ int sLatch, sGood, sBest, sTie;
  __shared__ uint32_t sBase;
  __shared__ CsTables sCs;
  __shared__ uint16_t sCnt[256 * (NT / 64)];
  const int tid = threadIdx.x;
  unsigned int nbTotal = 0;
  // read-ends are handed out one at a time (a device counter per launch): their cost is uneven, and a workgroup of this shape that waits
  // for a free CU would otherwise hold up its whole fixed share of them.  (Alone the kernels are 20 - 35 % faster for it; under three
  // pipelines the step is not: what one kernel gives back the others take.  64 at a time with the counts pre-read was slower.)
  __shared__ uint32_t sNextRe;
  uint32_t hoNext = 0, hoLeft = 0;  // thread 0: what it holds from its last hand-out
  for (;;) {
    __syncthreads();
    if (tid == 0) {  // (T1K_RE_HANDOUT read-ends per atomic on the hand-out word: round 6)
      constexpr uint32_t HO = SELECT_LDS_CAP == SELECT_SMALL ? T1K_SELECT_SMALL_HANDOUT : T1K_SELECT_LARGE_HANDOUT;
      if (hoLeft == 0) { hoNext = (uint32_t)atomicAdd(&P.counters[SELECT_LDS_CAP == SELECT_SMALL ? 28 : 29], (unsigned long long)HO); hoLeft = HO; }
      sNextRe = hoNext++; --hoLeft;
    }
    __syncthreads();
    const uint32_t re = sNextRe;
    if (re >= P.reads.nReadEnds) break;
    const uint32_t n = P.candCount[re], c0 = P.candStart[re];
    if (n == 0) {
      if (tid == 0 && SELECT_LDS_CAP == SELECT_SMALL) { P.ovlStart[re] = 0; P.ovlCount[re] = 0; }
      continue;
    }
    // every candidate of the list is sorted (the ones that failed the similarity filter were left out by k_collect; until round 4 this
    // kernel -- both instantiations -- swept the extension records once just to count survivors that all survive, each thread over its own
    // contiguous piece).  Candidate i goes to position i: the list keeps the order in which k_collect wrote it (allele order), which the
    // counting sort below relies on, and neighbouring lanes read neighbouring records
    if ((n > SELECT_SMALL) != (SELECT_LDS_CAP == SELECT_LARGE)) continue;  // the other instantiation's read-end
    const uint32_t live = n;
    if (live == 0) {
      if (tid == 0) { P.ovlStart[re] = 0; P.ovlCount[re] = 0; }
      continue;
    }
    uint32_t np2 = 8;
    while (np2 < live) np2 <<= 1;
    uint64_t *key;
    if (np2 <= SELECT_LDS_CAP) key = sKey;
    else if (np2 <= P.sortCap) key = P.sortScratch + (uint64_t)blockIdx.x * P.sortCap * 6;
    else {
      if (tid == 0) { atomicOr(&P.counters[2], (unsigned long long)ERR_SORTCAP); P.ovlStart[re] = 0; P.ovlCount[re] = 0; }
      continue;
    }
    // key: matchCnt desc, similarity desc (== readSpan+seqSpan asc at equal matchCnt), readSpan desc, allele asc; payload: candidate
    int iBits = 1;
    while ((1u << iBits) < n) ++iBits;
    const uint64_t iMask = (1ull << iBits) - 1;
    const KeyW W = selectWidths<XL>((int)P.reads.len[re]);
    const int mShift = W.DB + W.RB + P.alleleBits + iBits + 3;  // position of the (MMAX - matchCnt) field
    const int mMax = (1 << W.MB) - 1;
    auto idxOf = [&](uint64_t kk) { return (uint32_t)((kk >> 3) & iMask); };
    auto seedOf = [&](uint64_t kk) { return mMax - (int)((kk >> mShift) & (uint64_t)mMax); };
    for (uint32_t i = live + tid; i < np2; i += NT) key[i] = ~0ull;
    for (uint32_t i = tid; i < n; i += NT) {
      const uint16_t fl = P.ext[c0 + i].flags;
      const T1kCand c = P.cand[c0 + i];
      int rs = c.readSE & 0xFFFF, rend = c.readSE >> 16;
      int m = (int)(c.match >> 16);
      int rspan = rend - rs, d = rspan + 1 + c.seqEnd - c.seqStart + 1;
      const double sim = (double)m / (double)(c.seqEnd - c.seqStart + 1 + rend - rs + 1);
      const uint32_t kf = ((fl & T1K_F_SEPSEED) ? SEL_F_SEPSEED : 0u) | ((fl & T1K_F_EXTOK) ? SEL_F_EXTOK : 0u) |
                          (((fl & T1K_F_NEEDCLIP) && !(sim < 0.95)) ? SEL_F_KEEPCLIP : 0u);  // SeqSet.hpp:2170-2172
      const uint32_t slot = i;
      uint64_t kk;
      if (!packSelectKey(W, m, d, rspan, c.allele & 0x7FFFFFFFu, i, kf, P.alleleBits, iBits, &kk)) { atomicOr(&P.counters[2], (unsigned long long)ERR_SORTCAP); kk = (uint64_t)i << 3; }
      key[slot] = kk;
    }
    __syncthreads();
    // the list is in allele order: one stable pass by the (seed matchCnt, span sum, read span) class orders it; the bitonic network
    // on the whole key takes the lists with more than 256 classes and the ones that live in HBM scratch
    if (key != sKey || !ldsClassRanks<NT>(key, live, P.alleleBits + iBits + 3, sCs)) bitonicSort(key, np2);
    else ldsCountingPass<NT>(key, live, sCnt, warpSums, [&](uint64_t kk) { return csRankOf(sCs, (uint32_t)(kk >> (P.alleleBits + iBits + 3))); });
    // resolve ties of the packed key with the remaining comparator fields (same allele, same spans)
    const uint32_t nAll = n;
    (void)nAll;
    // equal packed keys (same allele, same spans and match count; rare) are ordered by the remaining comparator fields
    if (tid == 0) { sTie = 0; sLatch = 0x7FFFFFFF; sGood = -1; sBest = -1; }
    __syncthreads();
    for (uint32_t i = 1 + tid; i < live; i += NT)
      if ((key[i] >> (iBits + 3)) == (key[i - 1] >> (iBits + 3))) sTie = 1;
    __syncthreads();
    if (sTie && tid == 0) {
      for (uint32_t i = 1; i < live; ++i) {
        if ((key[i] >> (iBits + 3)) != (key[i - 1] >> (iBits + 3))) continue;
        uint32_t j = i;
        while (j > 0 && (key[j - 1] >> (iBits + 3)) == (key[j] >> (iBits + 3)) && candBeforeFull(P.cand[c0 + idxOf(key[j])], P.cand[c0 + idxOf(key[j - 1])])) {
          uint64_t t = key[j]; key[j] = key[j - 1]; key[j - 1] = t;
          --j;
        }
      }
    }
    __syncthreads();
    // latch position: first tried candidate whose extension fails (all candidates before the latch are tried)
    int myLatch = 0x7FFFFFFF;
    for (uint32_t i = tid; i < live; i += NT) {
      const uint32_t fl = (uint32_t)key[i] & 7u;
      if (fl & SEL_F_SEPSEED) continue;
      if (!(fl & SEL_F_EXTOK)) { if ((int)i < myLatch) myLatch = (int)i; }
    }
    atomicMin(&sLatch, myLatch);
    __syncthreads();
    const int latch = sLatch;
    // goodMatchCnt = seed matchCnt of the first emitted candidate before the latch (the list is sorted by it)
    int myGood = 0x7FFFFFFF;
    for (uint32_t i = tid; i < live && (int)i < latch; i += NT) {
      const uint32_t fl = (uint32_t)key[i] & 7u;
      if ((fl & SEL_F_SEPSEED) || !(fl & SEL_F_EXTOK)) continue;
      if ((int)i < myGood) myGood = (int)i;
    }
    __shared__ int sFirst;
    if (tid == 0) sFirst = 0x7FFFFFFF;
    __syncthreads();
    atomicMin(&sFirst, myGood);
    __syncthreads();
    if (tid == 0) sGood = sFirst == 0x7FFFFFFF ? -1 : seedOf(key[sFirst]);
    __syncthreads();
    const int good = sGood;
    // emit flags + best extended matchCnt
    uint32_t written = 0;
    unsigned int nbLocal = 0;
    int myBest = -1;
    // first pass: count and best
    uint32_t mine = 0;
    for (uint32_t i = tid; i < live; i += NT) {
      const uint64_t kk = key[i];
      const uint32_t fl = (uint32_t)kk & 7u;
      bool emit = false;
      if (!(fl & SEL_F_SEPSEED)) {
        bool tried = true;
        if ((int)i > latch && seedOf(kk) < good && !(fl & SEL_F_KEEPCLIP)) tried = false;  // SeqSet.hpp:2170-2172
        emit = tried && (fl & SEL_F_EXTOK);
        if (emit) { const int xm = (int)P.ext[c0 + idxOf(kk)].matchCnt; if (xm > myBest) myBest = xm; }
      }
      if (emit) { ++mine; key[i] = kk | (1ull << 63); }
    }
    atomicMax(&sBest, myBest);
    uint32_t tot;
    t1k_block_scan_exclusive_n<NT / 64>(mine, warpSums, &tot);
    if (tid == 0) {
      unsigned long long b = atomicAdd(&P.counters[1], (unsigned long long)tot);
      if (b + tot > P.ovlCap) { atomicOr(&P.counters[2], (unsigned long long)ERR_OVLCAP); sBase = 0xFFFFFFFFu; P.ovlStart[re] = 0; P.ovlCount[re] = 0; }
      else { sBase = (uint32_t)b; P.ovlStart[re] = (uint32_t)b; P.ovlCount[re] = tot; }
    }
    __syncthreads();
    const int bestMatch = sBest;
    if (sBase != 0xFFFFFFFFu) {
      for (uint32_t i0 = 0; i0 < live; i0 += NT) {
        uint32_t i = i0 + tid;
        uint32_t flag = (i < live && (key[i] >> 63)) ? 1u : 0u;
        uint32_t t2;
        uint32_t off = t1k_block_scan_exclusive_n<NT / 64>(flag, warpSums, &t2);
        if (flag) {
          uint32_t ci = c0 + idxOf(key[i]);
          const T1kCand c = P.cand[ci];
          const T1kExt x = P.ext[ci];
          T1kOvl o;
          o.allele = c.allele & 0x7FFFFFFFu;
          o.seqStart = x.seqStart; o.seqEnd = x.seqEnd; o.readStart = x.readStart; o.readEnd = x.readEnd;
          o.matchCnt = x.matchCnt; o.leftClip = x.leftClip; o.rightClip = x.rightClip; o.re = re;
          o.flags = ((int)x.matchCnt >= bestMatch - 10 ? 1u : 0u) | ((c.allele >> 31) ? 0u : 2u);  // SeqSet.hpp:2200
          // without --relaxIntronAlign the relaxed count is the match count of a near-best overlap and 0 otherwise (SeqSet.hpp:2247-2250, 2282):
          // known here, whether or not the full alignments run; with it the alignment kernels fill it in
          o.relaxed = (!P.relax && (o.flags & 1)) ? x.matchCnt : (uint16_t)0;
          if (o.flags & 1) ++nbLocal;
          P.ovl[(uint64_t)sBase + written + off] = o;
        }
        written += t2;
      }
    }
    nbTotal += nbLocal;
    __syncthreads();
  }
  t1k_stat_add(P.counters, T1K_STAT_NEARBEST, nbTotal);
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
  const int w = P.noCov ? 0 : (P.undo ? -(int)P.reads.weight[o.re] : (int)P.reads.weight[o.re]);
  bool slow = (L != Ls) || L > T1K_MAX_READ_LEN;  // (the closed form below keeps 320 covered-column bits in registers; longer spans take the traced DP)
  int x = 0, exonMis = 0;
  // one sweep over the windows: mismatch count, and -- kept in registers for the coverage updates below -- the covered-column
  // words (MATCH column, read base not N (2261-2265); an N allele base never feeds GetSeqMissingBaseCoverage's counter of the
  // allele's own base, so it is left out)
  uint64_t covw[10];  // spans of at most T1K_MAX_READ_LEN = 320 positions
  if (!slow && L <= 160) {
    // the usual read length: the allele window (6 words) comes in with three 16-byte loads per array instead of one load pair per
    // 32-base piece -- every lane's window is in another cache line, so the number of load instructions is what the kernel pays for --
    // and the N mask of an allele without N is not fetched at all
    typedef uint64_t u64x2 __attribute__((ext_vector_type(2), aligned(8)));
    uint64_t gB[6], gN[6], rB[6], rN[6], gE[6];
    const int64_t gpos = goff + o.seqStart;
    const int64_t gw = gpos >> 5, rw = o.readStart >> 5;
    const int gsh = (int)(gpos & 31) * 2, rsh = (o.readStart & 31) * 2;
    const bool hasN = P.ref.alleleHasN[o.allele] != 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const u64x2 a = *(const u64x2 *)(P.ref.bases + gw + 2 * q), b = *(const u64x2 *)(rb + rw + 2 * q), c = *(const u64x2 *)(rn + rw + 2 * q);
      gB[2 * q] = a.x; gB[2 * q + 1] = a.y; rB[2 * q] = b.x; rB[2 * q + 1] = b.y; rN[2 * q] = c.x; rN[2 * q + 1] = c.y;
      gN[2 * q] = 0; gN[2 * q + 1] = 0; gE[2 * q] = 0; gE[2 * q + 1] = 0;
    }
    if (hasN) {
#pragma unroll
      for (int q = 0; q < 3; ++q) { const u64x2 a = *(const u64x2 *)(P.ref.nmask + gw + 2 * q); gN[2 * q] = a.x; gN[2 * q + 1] = a.y; }
    }
    if (P.relax) {
#pragma unroll
      for (int q = 0; q < 3; ++q) { const u64x2 a = *(const u64x2 *)(P.ref.exon + gw + 2 * q); gE[2 * q] = a.x; gE[2 * q + 1] = a.y; }
    }
    auto piece = [](const uint64_t *W, int wi, int sh) { return (W[wi] >> sh) | ((W[wi + 1] << 1) << (63 - sh)); };
#pragma unroll
    for (int wi = 0; wi < 10; ++wi) {
      const int off = wi * 32;
      covw[wi] = 0;
      if (wi < 5 && off < L) {
        const uint64_t lm = t1k_lowmask(L - off);
        const uint64_t rnn = piece(rN, wi, rsh), gnn = piece(gN, wi, gsh);
        const uint64_t xo = piece(rB, wi, rsh) ^ piece(gB, wi, gsh);
        const uint64_t mm = (xo | (xo >> 1)) & T1K_EVEN & ~(rnn | gnn) & lm;
        x += __popcll(mm);
        if (P.relax) exonMis += __popcll(mm & piece(gE, wi, gsh));
        covw[wi] = T1K_EVEN & lm & ~mm & ~rnn & ~gnn;
      }
    }
    if (x > 3) slow = true;
  } else if (!slow) {
#pragma unroll
    for (int wi = 0; wi < 10; ++wi) {
      const int off = wi * 32;
      covw[wi] = 0;
      if (off < L) {
        const uint64_t lm = t1k_lowmask(L - off);
        const uint64_t rnn = t1k_get32(rn, o.readStart + off), gnn = t1k_get32(P.ref.nmask, goff + o.seqStart + off);
        const uint64_t xo = t1k_get32(rb, o.readStart + off) ^ t1k_get32(P.ref.bases, goff + o.seqStart + off);
        const uint64_t mm = (xo | (xo >> 1)) & T1K_EVEN & ~(rnn | gnn) & lm;
        x += __popcll(mm);
        if (P.relax) exonMis += __popcll(mm & t1k_get32(P.ref.exon, goff + o.seqStart + off));
        covw[wi] = T1K_EVEN & lm & ~mm & ~rnn & ~gnn;
      }
    }
    if (x > 3) slow = true;
  }
  if (slow) {
    if (P.undo) return;  // (queued alignments had not added anything yet)
    // equal spans: register-band traced DP (queue A, from the front); unequal spans: general DP (queue B, from the back)
    const int dl = L > Ls ? L - Ls : Ls - L;
    // sort key of the queue entry: alignments of one read window against identical allele windows become neighbours, and the
    // traced kernels fill the DP only once per run of them (content hash; equality is verified there, the hash only orders)
    unsigned long long key = 0;
    if (dl <= 4) {
      uint64_t hsh = 0x9E3779B97F4A7C15ull;
      for (int off = 0; off < Ls; off += 32) {
        const uint64_t lm = t1k_lowmask(Ls - off);
        hsh = (hsh ^ (t1k_get32(P.ref.bases, goff + o.seqStart + off) & lm) ^ ((t1k_get32(P.ref.nmask, goff + o.seqStart + off) & lm) << 1)) * 0xD6E8FEB86659FD93ull;
        hsh ^= hsh >> 32;
      }
      // (9-bit fields: a span or start beyond 511 -- reads longer than T1K_MAX_READ_LEN -- spills into its neighbour, which only makes
      // the order coarser: k_align_flags compares the jobs themselves)
      key = ((unsigned long long)o.re << 44) | ((unsigned long long)pass << 43) | ((unsigned long long)o.readStart << 34) | ((unsigned long long)L << 25) |
            ((unsigned long long)(Ls - L + 4) << 21) | (hsh & 0x1FFFFFull);
    }
    if (L == Ls) {
      const uint32_t q = t1k_arena_append(P.counters, T1K_AR_EQ, P.segCap);
      if (q != T1K_ARENA_FULL) { P.eqStr[q] = (uint32_t)gid; P.eqKeyStr[q] = key; }
    } else if (dl <= 4) {  // register-band DP (k_fullalign_band)
      const uint32_t q = t1k_arena_append(P.counters, T1K_AR_BAND, P.segCap);
      if (q != T1K_ARENA_FULL) { P.bandStr[q] = (uint32_t)gid; P.bandKeyStr[q] = key; }
    } else {               // wide band: general DP with row arrays in HBM (k_fullalign_slow)
      const uint32_t q = t1k_arena_append(P.counters, T1K_AR_WIDE, P.segCap);
      if (q != T1K_ARENA_FULL) P.wideStr[q] = (uint32_t)gid;
    }
    return;
  }
  // ungapped alignment: columns are MATCH except at the x mismatching positions.  Coverage: the whole span as one run in the
  // difference array (2 atomics) and one "hole" per position of the span that is not covered (mismatch or N)
  if (w) {
    int32_t *diff = P.ref.covDiff + goff + o.seqStart, *hole = diff + P.ref.covStride;
    if (L == P.fullLen) atomicAdd(&diff[2 * P.ref.covStride], w);  // a run of the usual length: counted by its start (t1k_coverage_fold)
    else { atomicAdd(&diff[0], w); atomicAdd(&diff[L], -w); }
#pragma unroll
    for (int wi = 0; wi < 10; ++wi) {
      const int off = wi * 32;
      if (off < L) {
        uint64_t un = T1K_EVEN & t1k_lowmask(L - off) & ~covw[wi];
        while (un) {
          const int b = __ffsll((long long)un) - 1;
          un &= un - 1;
          atomicAdd(&hole[off + (b >> 1)], w);
        }
      }
    }
  }
  int relaxed = P.relax ? 2 * (L - exonMis) : (int)o.matchCnt;  // 2215-2250
  P.ovl[gid].relaxed = (uint16_t)relaxed;
}


// (rare: a handful per batch.)  One working lane per 64-thread workgroup: the row arrays of the general DP live in LDS -- with
// them in HBM every cell paid a memory round trip and the kernel sat on the batch's critical path for a millisecond.
__global__ __launch_bounds__(64) void k_fullalign_slow(SlowArgs P) {
  __shared__ int rows[GA_SCRATCH_INTS];
  if (threadIdx.x != 0) return;
  uint32_t t = blockIdx.x;
  uint32_t nThreads = gridDim.x;
  uint8_t *mine = P.scratch + (uint64_t)t * P.perThread;
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
    const int w = P.noCov ? 0 : (int)P.reads.weight[o.re];
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
// one bit per position of a 2-bit-geometry mask stream, fetched 32 positions at a time (the tracebacks walk downwards)
struct T1kWinBits {
  const uint64_t *p;
  int64_t off;
  int base;
  uint64_t w;
  __device__ T1kWinBits(const uint64_t *p_, int64_t off_) : p(p_), off(off_), base(1 << 30), w(0) {}
  __device__ __forceinline__ int bit(int pos) {
    if (pos < base || pos >= base + 32) { base = pos > 31 ? pos - 31 : 0; w = t1k_get32(p, off + base); }
    return (int)((w >> (2 * (pos - base))) & 1);
  }
};

// the same for the 2-bit base codes of a packed stream: 32 positions in a register, refilled when the walk leaves them
struct T1kWinCodes {
  const uint64_t *p;
  int64_t off;
  int base;
  uint64_t w;
  __device__ T1kWinCodes(const uint64_t *p_, int64_t off_) : p(p_), off(off_), base(1 << 30), w(0) {}
  __device__ __forceinline__ int code(int pos) {
    if (pos < base || pos >= base + 32) { base = pos > 31 ? pos - 31 : 0; w = t1k_get32(p, off + base); }
    return (int)((w >> (2 * (pos - base))) & 3);
  }
};

// The traced alignments run in three steps over the SORTED queue (identical jobs are neighbours):
//   k_align_flags  marks the first job of every run of identical (read window, allele window) jobs; an inclusive scan numbers the runs
//   k_align_fill   one lane per run: DP sweep, decision words -> trace[row * stride + run]
//   k_align_apply  one lane per job: traceback over its run's decision words -> relaxed match count + coverage of its own allele
// (neighbouring lanes read the same or adjacent trace columns)
__global__ __launch_bounds__(WG) void k_align_flags(SlowArgs P, uint32_t *flags) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= P.nSlow) return;
  uint32_t fl = 1;
  if (q > 0) {
    const T1kOvl o = P.ovl[P.slowQueue[q]], p = P.ovl[P.slowQueue[q - 1]];
    const int lt = o.seqEnd - o.seqStart + 1;
    if (o.re == p.re && ((o.flags ^ p.flags) & 2) == 0 && o.readStart == p.readStart && o.readEnd == p.readEnd && lt == p.seqEnd - p.seqStart + 1 &&
        t1k_same_window(P.ref.bases, P.ref.nmask, (int64_t)P.ref.alleleOff[p.allele] + p.seqStart, (int64_t)P.ref.alleleOff[o.allele] + o.seqStart, lt, P.ref.anyN != 0))
      fl = 0;
  }
  flags[q] = fl;
}
__global__ __launch_bounds__(WG) void k_align_reps(const uint32_t *flags, const uint32_t *runOf, uint32_t *rep, uint32_t n) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n && flags[q]) rep[runOf[q] - 1] = q;
}

template <bool EQ>
__global__ __launch_bounds__(WG) void k_align_fill(SlowArgs P) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long cells = 0;
  if (u < P.nRuns) {
    const uint32_t q = P.rep[u];
    const uint32_t gid = P.slowQueue[q];
    const T1kOvl o = P.ovl[gid];
    const int pass = (o.flags & 2) ? 1 : 0;
    const int S = P.reads.S;
    const uint64_t *rb = P.reads.bases + ((uint64_t)o.re * 2 + pass) * S;
    const uint64_t *rn = P.reads.nmask + ((uint64_t)o.re * 2 + pass) * S;
    const int64_t goff = (int64_t)P.ref.alleleOff[o.allele];
    const int lp = o.readEnd - o.readStart + 1, lt = o.seqEnd - o.seqStart + 1;
    T1kSeqView T{P.ref.bases, P.ref.nmask, goff + o.seqStart}, Pv{rb, rn, (int64_t)o.readStart};
    uint64_t *trace = (uint64_t *)P.scratch + u;
    if (EQ) t1k_ga_equal_traced(T, Pv, lp, trace, P.traceStride);
    else t1k_ga_band<4, true>(T, lt, Pv, lp, trace, P.traceStride);
    cells = (unsigned long long)lp * (EQ ? 11u : 9u);  // in-band cells per row: 11 in the equal-span sweep, the 9 diagonals of the +-4 band
  }
  // statistics: DP cells filled, one atomic per wavefront
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cells += __shfl_xor(cells, o, 64);
  if ((threadIdx.x & 63) == 0 && cells) atomicAdd(&P.counters[24], cells);
}

__global__ __launch_bounds__(WG) void k_align_apply_eq(SlowArgs P) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= P.nSlow) return;
  const size_t nThreads = P.traceStride;
  const uint64_t *trace = (const uint64_t *)P.scratch + (P.runOf[q] - 1);
  {
    const uint32_t gid = P.slowQueue[q];
    const T1kOvl o = P.ovl[gid];
    const int pass = (o.flags & 2) ? 1 : 0;
    const int S = P.reads.S;
    const uint64_t *rb = P.reads.bases + ((uint64_t)o.re * 2 + pass) * S;
    const uint64_t *rn = P.reads.nmask + ((uint64_t)o.re * 2 + pass) * S;
    const int64_t goff = (int64_t)P.ref.alleleOff[o.allele];
    const int L = o.readEnd - o.readStart + 1;
    const int w = P.noCov ? 0 : (int)P.reads.weight[o.re];
    int32_t *cov = P.ref.covDiff + goff;
    T1kWinBits exW(P.ref.exon, goff), gnW(P.ref.nmask, goff), rnW(rn, 0);
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
        bool ex = refPos < alleleLen ? exW.bit(refPos) != 0 : false;
        if (!ex || op == 0) ++relaxed;
      }
      if (op == 0 && w) {
        const int readPos = o.readStart + ti;  // ti was already decremented: this column consumed read base ti
        if (!rnW.bit(readPos) && !gnW.bit(refPos)) {
          if (refPos == runLo - 1) runLo = refPos;
          else if (refPos == runLo - 2 && runLo >= 0) { atomicAdd(&cov[P.ref.covStride + runLo - 1], w); runLo = refPos; }  // one uncovered position: a hole in the run (1 atomic, not 2)
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

__global__ __launch_bounds__(WG) void k_align_apply_band(SlowArgs P) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= P.nSlow) return;
  const size_t nThreads = P.traceStride;
  const uint64_t *trace = (const uint64_t *)P.scratch + (P.runOf[q] - 1);
  {
    const uint32_t gid = P.slowQueue[q];
    const T1kOvl o = P.ovl[gid];
    const int pass = (o.flags & 2) ? 1 : 0;
    const int S = P.reads.S;
    const uint64_t *rb = P.reads.bases + ((uint64_t)o.re * 2 + pass) * S;
    const uint64_t *rn = P.reads.nmask + ((uint64_t)o.re * 2 + pass) * S;
    const int64_t goff = (int64_t)P.ref.alleleOff[o.allele];
    const int lp = o.readEnd - o.readStart + 1, lt = o.seqEnd - o.seqStart + 1;
    const int LB = 5 + (lp > lt ? lp - lt : 0);
    const int w = P.noCov ? 0 : (int)P.reads.weight[o.re];
    T1kSeqView T{P.ref.bases, P.ref.nmask, goff + o.seqStart}, Pv{rb, rn, (int64_t)o.readStart};
    int32_t *cov = P.ref.covDiff + goff;
    T1kWinBits exW(P.ref.exon, goff), gnW(P.ref.nmask, goff), rnW(rn, 0);
    T1kWinCodes gbW(P.ref.bases, goff), rbW(rb, 0);
    const int alleleLen = (int)P.ref.alleleLen[o.allele];
    int relaxed = 0, runLo = -1, runHi = -1;
    int ti = lp, tj = lt, mat = 0;
    while (ti > 0 || tj > 0) {
      int bits;
      if (ti > 0 && tj > 0) bits = (int)((trace[(size_t)ti * nThreads] >> (4 * (tj - ti + LB))) & 15);
      else if (ti == 0) bits = 2 | (tj == 1 ? 8 : 0);
      else bits = (ti == 1 ? 4 : 0);
      int op, refPos;
      if (mat == 0) {
        if (ti > 0 && tj > 0 && (bits & 1)) {
          // the two bases of the column, from 32-position register windows that follow the walk (a load pair per 32 columns, not per column)
          const int ct = gnW.bit(o.seqStart + tj - 1) ? 4 : gbW.code(o.seqStart + tj - 1);
          const int cp = rnW.bit(o.readStart + ti - 1) ? 4 : rbW.code(o.readStart + ti - 1);
          op = t1k_eq(ct, cp) ? 0 : 1; refPos = o.seqStart + tj - 1; --ti; --tj;
        }
        else { mat = (bits & 2) ? 2 : 1; continue; }
      } else if (mat == 1) {
        op = 2; refPos = o.seqStart + tj;
        if (ti > 0) { if (bits & 4) mat = 0; --ti; } else mat = 2;
      } else {
        op = 3; refPos = o.seqStart + tj - 1;
        if (tj > 0) { if (bits & 8) mat = 0; --tj; } else mat = 1;
      }
      if (P.relax) {
        bool ex = refPos < alleleLen ? exW.bit(refPos) != 0 : false;
        if (!ex || op == 0) ++relaxed;
      }
      if (op == 0 && w) {
        const int readPos = o.readStart + ti;
        if (!rnW.bit(readPos) && !gnW.bit(refPos)) {
          if (refPos == runLo - 1) runLo = refPos;
          else if (refPos == runLo - 2 && runLo >= 0) { atomicAdd(&cov[P.ref.covStride + runLo - 1], w); runLo = refPos; }  // one uncovered position: a hole in the run (1 atomic, not 2)
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

#if T1K_TRUNC_SMALL_WAVES > 0 || T1K_TRUNC_LARGE_WAVES > 0
#define T1K_TRUNC_ATTR __attribute__((amdgpu_waves_per_eu(SELECT_LDS_CAP == SELECT_SMALL ? (T1K_TRUNC_SMALL_WAVES > 0 ? T1K_TRUNC_SMALL_WAVES : 1) : (T1K_TRUNC_LARGE_WAVES > 0 ? T1K_TRUNC_LARGE_WAVES : 1))))
#else
#define T1K_TRUNC_ATTR
#endif
template <int SELECT_LDS_CAP, int NT, bool XL>
__global__ __launch_bounds__(NT) T1K_TRUNC_ATTR void k_truncate(TruncArgs P) {
  extern __shared__ uint64_t dynLds[];
  uint64_t *sKey = dynLds;
  __shared__ uint32_t sCut, sTie2;
  __shared__ uint32_t warpSums[NT / 64];
  __shared__ CsTables sCs;
  __shared__ uint16_t sCnt[256 * (NT / 64)];
  const int tid = threadIdx.x;
  __shared__ uint32_t sNextRe;  // read-ends handed out one at a time, as in k_select
  uint32_t hoNext = 0, hoLeft = 0;
  for (;;) {
    __syncthreads();
    if (tid == 0) {  // (T1K_RE_HANDOUT read-ends per atomic on the hand-out word: round 6)
      constexpr uint32_t HO = SELECT_LDS_CAP == SELECT_SMALL ? T1K_TRUNC_SMALL_HANDOUT : T1K_TRUNC_LARGE_HANDOUT;
      if (hoLeft == 0) { hoNext = (uint32_t)atomicAdd(&P.counters[SELECT_LDS_CAP == SELECT_SMALL ? 30 : 31], (unsigned long long)HO); hoLeft = HO; }
      sNextRe = hoNext++; --hoLeft;
    }
    __syncthreads();
    const uint32_t re = sNextRe;
    if (re >= P.reads.nReadEnds) break;
    const uint32_t n = P.ovlCount[re], o0 = P.ovlStart[re];
    if (n <= 1000) continue;
    if ((n > SELECT_SMALL) != (SELECT_LDS_CAP == SELECT_LARGE)) continue;  // the other instantiation's read-end
    uint32_t np2 = 8;
    while (np2 < n) np2 <<= 1;
    uint64_t *key;
    uint64_t *wgScratch = P.sortScratch + (uint64_t)blockIdx.x * ((uint64_t)P.sortCap * 6);
    if (np2 <= SELECT_LDS_CAP) key = sKey;
    else if (np2 <= P.sortCap) key = wgScratch;
    else { if (tid == 0) atomicOr(&P.counters[2], (unsigned long long)ERR_SORTCAP); continue; }
    T1kOvl *stage = (T1kOvl *)(wgScratch + P.sortCap * 2);
    if (n > P.sortCap) { if (tid == 0) atomicOr(&P.counters[2], (unsigned long long)ERR_SORTCAP); continue; }
    int iBits = 1;
    while ((1u << iBits) < n) ++iBits;
    const uint64_t iMask = (1ull << iBits) - 1;
    const KeyW W = truncWidths<XL>((int)P.reads.len[re]);
    for (uint32_t i = tid; i < np2; i += NT) {
      uint64_t kk = ~0ull;
      if (i < n) {
        const T1kOvl o = P.ovl[o0 + i];
        int rspan = o.readEnd - o.readStart;
        int d = rspan + 1 + o.seqEnd - o.seqStart + 1 + 2 * o.leftClip + 2 * o.rightClip;  // similarity desc == d asc at equal matchCnt
        if (!packSortKey(W, (int)o.matchCnt, d, rspan, o.allele, i, P.alleleBits, iBits, &kk)) { atomicOr(&P.counters[2], (unsigned long long)ERR_SORTCAP); kk = i; }
        stage[i] = o;
      }
      key[i] = kk;
    }
    __syncthreads();
    // the list arrives in k_select's order, not in allele order: stable passes over the allele's bytes first, then one over the
    // class (matchCnt, span sum, read span); lists with more than 256 classes and the ones in HBM scratch: the bitonic network
    if (key != sKey || !ldsClassRanks<NT>(key, n, P.alleleBits + iBits, sCs)) bitonicSort(key, np2);
    else {
      for (int sh = 0; sh < P.alleleBits; sh += 8)
        ldsCountingPass<NT>(key, n, sCnt, warpSums, [&](uint64_t kk) { return (uint32_t)(kk >> (iBits + sh)) & (P.alleleBits - sh >= 8 ? 0xFFu : ((1u << (P.alleleBits - sh)) - 1u)); });
      ldsCountingPass<NT>(key, n, sCnt, warpSums, [&](uint64_t kk) { return csRankOf(sCs, (uint32_t)(kk >> (P.alleleBits + iBits))); });
    }
    if (tid == 0) { sCut = n; sTie2 = 0; }
    __syncthreads();
    for (uint32_t i = 1 + tid; i < n; i += NT)
      if ((key[i] >> iBits) == (key[i - 1] >> iBits)) sTie2 = 1;
    __syncthreads();
    if (sTie2 && tid == 0) {
      for (uint32_t i = 1; i < n; ++i) {
        if ((key[i] >> iBits) != (key[i - 1] >> iBits)) continue;
        uint32_t j = i;
        while (j > 0 && (key[j - 1] >> iBits) == (key[j] >> iBits) && ovlBeforeFull(stage[key[j] & iMask], stage[key[j - 1] & iMask])) {
          uint64_t t = key[j]; key[j] = key[j - 1]; key[j - 1] = t;
          --j;
        }
      }
    }
    __syncthreads();
    {
      // first j >= 1 whose similarity falls more than 0.1 below the best one (SeqSet.hpp:2294-2297)
      const double s0 = ovlSimilarity(stage[key[0] & iMask]);
      uint32_t mine = n;
      for (uint32_t j = 1 + tid; j < n; j += NT)
        if (ovlSimilarity(stage[key[j] & iMask]) < s0 - 0.1) { mine = j; break; }
      if (mine < n) atomicMin(&sCut, mine);
    }
    __syncthreads();
    const uint32_t cut = sCut;
    for (uint32_t i = tid; i < cut; i += NT) P.ovl[o0 + i] = stage[key[i] & iMask];
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
    uint32_t ex = t1k_block_scan_exclusive((uint32_t)v, warpSums, &tot);
    int32_t incl = (int32_t)ex + v + sCarry;
    if (i < len) o[i] = incl - d[ref.covStride + i];  // minus the holes
    __syncthreads();
    if (threadIdx.x == 0) sCarry += (int32_t)tot;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------------------------
int t1k_launch_pack(t1k_ctx *ctx, const char *dAscii, const uint64_t *dOffs, uint32_t n, int S, uint64_t *bases, uint64_t *nmask, uint16_t *lens, int nCode) {
  uint64_t total = (uint64_t)n * S;
  if (!total) return 0;
  hipLaunchKernelGGL(k_pack_reads, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, dAscii, dOffs, n, S, bases, nmask, lens, nCode);
  return 0;
}

size_t t1k_slow_per_thread(int maxCells) { return (size_t)GA_SCRATCH_INTS * 4 + 4224 + (size_t)maxCells + 64; }

void t1k_launch_extend(t1k_ctx *ctx, const ExtendArgs &a) {
  if (!a.nCand) return;
  hipLaunchKernelGGL(k_extend, dim3((unsigned)((a.nCand + WG - 1) / WG)), dim3(WG), 0, ctx->stream, a);
}
void t1k_launch_extend_retry(t1k_ctx *ctx, const ExtendArgs &a, const uint32_t *list, uint32_t n) {
  if (!n) return;
  hipLaunchKernelGGL(k_extend_retry, dim3((n + WG - 1) / WG), dim3(WG), 0, ctx->stream, a, list, n, (const unsigned long long *)nullptr);
}
void t1k_launch_extend_retry_dev(t1k_ctx *ctx, const ExtendArgs &a, const uint32_t *list, int arena, uint64_t est) {
  const unsigned long long *nDev = (const unsigned long long *)ctx->bCounters.p + T1K_TOTAL_BASE + arena;
  hipLaunchKernelGGL(k_extend_retry, dim3((unsigned)std::max<uint64_t>(1, (est + WG - 1) / WG)), dim3(WG), 0, ctx->stream, a, list, 0u, nDev);
}
void t1k_launch_select(t1k_ctx *ctx, const SelectArgs &a, int nWg) {
  if (a.xl) {  // a window with reads beyond T1K_MAX_READ_LEN: wider key fields
    hipFuncSetAttribute((const void *)k_select<SELECT_LARGE, 1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SELECT_LARGE * 8);
    hipLaunchKernelGGL((k_select<SELECT_SMALL, 256, true>), dim3(nWg), dim3(256), SELECT_SMALL * 8, ctx->stream, a);
    hipLaunchKernelGGL((k_select<SELECT_LARGE, 1024, true>), dim3(std::min(nWg, 512)), dim3(1024), SELECT_LARGE * 8, ctx->stream, a);
    return;
  }
  hipFuncSetAttribute((const void *)k_select<SELECT_LARGE, 1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SELECT_LARGE * 8);
  hipLaunchKernelGGL((k_select<SELECT_SMALL, 256, false>), dim3(nWg), dim3(256), SELECT_SMALL * 8, ctx->stream, a);
  hipLaunchKernelGGL((k_select<SELECT_LARGE, 1024, false>), dim3(std::min(nWg, 512)), dim3(1024), SELECT_LARGE * 8, ctx->stream, a);
}
void t1k_launch_fullalign(t1k_ctx *ctx, const FullArgs &a) {
  if (!a.nOvl) return;
  hipLaunchKernelGGL(k_fullalign, dim3((unsigned)((a.nOvl + WG - 1) / WG)), dim3(WG), 0, ctx->stream, a);
}
void t1k_launch_fullalign_slow(t1k_ctx *ctx, const SlowArgs &a, int nBlocks) { hipLaunchKernelGGL(k_fullalign_slow, dim3(nBlocks * 64), dim3(64), 0, ctx->stream, a); }
void t1k_launch_align_flags(t1k_ctx *ctx, const SlowArgs &a, uint32_t *flags) {
  hipLaunchKernelGGL(k_align_flags, dim3((a.nSlow + WG - 1) / WG), dim3(WG), 0, ctx->stream, a, flags);
}
void t1k_launch_align_reps(t1k_ctx *ctx, const uint32_t *flags, const uint32_t *runOf, uint32_t *rep, uint32_t n) {
  hipLaunchKernelGGL(k_align_reps, dim3((n + WG - 1) / WG), dim3(WG), 0, ctx->stream, flags, runOf, rep, n);
}
void t1k_launch_align_fill_apply(t1k_ctx *ctx, const SlowArgs &a, bool eq) {
  if (eq) {
    hipLaunchKernelGGL(k_align_fill<true>, dim3((a.nRuns + WG - 1) / WG), dim3(WG), 0, ctx->stream, a);
    hipLaunchKernelGGL(k_align_apply_eq, dim3((a.nSlow + WG - 1) / WG), dim3(WG), 0, ctx->stream, a);
  } else {
    hipLaunchKernelGGL(k_align_fill<false>, dim3((a.nRuns + WG - 1) / WG), dim3(WG), 0, ctx->stream, a);
    hipLaunchKernelGGL(k_align_apply_band, dim3((a.nSlow + WG - 1) / WG), dim3(WG), 0, ctx->stream, a);
  }
}
void t1k_launch_truncate(t1k_ctx *ctx, const TruncArgs &a, int nWg) {
  if (a.xl) {
    hipFuncSetAttribute((const void *)k_truncate<SELECT_LARGE, 1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SELECT_LARGE * 8);
    hipLaunchKernelGGL((k_truncate<SELECT_SMALL, 256, true>), dim3(std::min(nWg, 512)), dim3(256), SELECT_SMALL * 8, ctx->stream, a);
    hipLaunchKernelGGL((k_truncate<SELECT_LARGE, 1024, true>), dim3(std::min(nWg, 512)), dim3(1024), SELECT_LARGE * 8, ctx->stream, a);
    return;
  }
  hipFuncSetAttribute((const void *)k_truncate<SELECT_LARGE, 1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SELECT_LARGE * 8);
  hipLaunchKernelGGL((k_truncate<SELECT_SMALL, 256, false>), dim3(std::min(nWg, 512)), dim3(256), SELECT_SMALL * 8, ctx->stream, a);  // (its staging area: 512 workgroups' worth)
  hipLaunchKernelGGL((k_truncate<SELECT_LARGE, 1024, false>), dim3(std::min(nWg, 512)), dim3(1024), SELECT_LARGE * 8, ctx->stream, a);
}
__global__ __launch_bounds__(WG) void k_coverage_fold(int32_t *diff, const int32_t *full, uint64_t n, uint64_t len) {
  const uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x;
  if (i >= n) return;
  const int32_t v = full[i] - (i >= len ? full[i - len] : 0);  // runs starting here, minus runs that ended just before
  if (v) diff[i] += v;
}
int t1k_coverage_fold(t1k_ctx *ctx) {
  if (!ctx->covFullDirty || !ctx->ref.covDiff) return T1K_OK;
  const uint64_t n = ctx->ref.covStride;
  int32_t *full = ctx->ref.covDiff + 2 * n;
  hipLaunchKernelGGL(k_coverage_fold, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, ctx->stream, ctx->ref.covDiff, (const int32_t *)full, n, (uint64_t)ctx->covFullLen);
  if (hipMemsetAsync(full, 0, n * sizeof(int32_t), ctx->stream) != hipSuccess) return T1K_ERR_DEVICE;
  ctx->covFullDirty = false;
  return T1K_OK;
}
__global__ __launch_bounds__(WG) void k_coverage_add(int32_t *dst, int32_t *src, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x;
  if (i < n) { dst[i] += src[i]; src[i] = 0; }
}
void t1k_launch_coverage_add(t1k_ctx *ctx, int32_t *dst, int32_t *src, uint64_t n) {
  hipLaunchKernelGGL(k_coverage_add, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, ctx->stream, dst, src, n);
}
// GetSeqMissingBaseCoverage on the device (SeqSet.hpp:2717-2755): per allele, the coverage of its exon positions; median by radix
// selection (the reference sorts and takes element size/2), cutoff = max(1, 1 % of the median), missing = positions below it.
// scratch: one int per reference position (the prefix-summed coverage of the allele under construction)
__global__ __launch_bounds__(WG) void k_missing_coverage(T1kRefDev ref, int32_t *scratch, int32_t *missing) {
  const uint32_t a = blockIdx.x;
  if (a >= ref.nAlleles) return;
  __shared__ uint32_t warpSums[4];
  __shared__ int32_t sCarry;
  __shared__ uint32_t sCnt;
  const int tid = threadIdx.x;
  const int len = (int)ref.alleleLen[a];
  const int64_t g0 = (int64_t)ref.alleleOff[a];
  const int32_t *d = ref.covDiff + g0;
  int32_t *cv = scratch + g0;
  if (tid == 0) sCarry = 0;
  __syncthreads();
  for (int base = 0; base < len; base += WG) {  // coverage = prefix sum of the difference array
    const int i = base + tid;
    const int32_t v = i < len ? d[i] : 0;
    uint32_t tot;
    const uint32_t ex = t1k_block_scan_exclusive((uint32_t)v, warpSums, &tot);
    if (i < len) cv[i] = (int32_t)ex + v + sCarry - d[ref.covStride + i];  // minus the holes
    __syncthreads();
    if (tid == 0) sCarry += (int32_t)tot;
    __syncthreads();
  }
  auto blockCount = [&](uint32_t mine) -> uint32_t {  // sum over the workgroup, returned to every thread
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if (tid == 0) sCnt = 0;
    __syncthreads();
    if ((tid & 63) == 0 && mine) atomicAdd(&sCnt, mine);
    __syncthreads();
    const uint32_t r = sCnt;
    __syncthreads();
    return r;
  };
  uint32_t mine = 0;
  for (int i = tid; i < len; i += WG) mine += (uint32_t)t1k_bit(ref.exon, g0 + i);
  const uint32_t nEx = blockCount(mine);
  if (nEx == 0) { if (tid == 0) missing[a] = 0; return; }
  // element nEx / 2 of the sorted exon coverages: fix the bits from the top (values mapped order-preservingly to unsigned)
  uint32_t k = nEx / 2, prefix = 0;
  for (int bit = 31; bit >= 0; --bit) {
    const uint32_t hiMask = bit == 31 ? 0u : ~0u << (bit + 1);
    uint32_t zeros = 0;
    for (int i = tid; i < len; i += WG) {
      if (!t1k_bit(ref.exon, g0 + i)) continue;
      const uint32_t u = (uint32_t)cv[i] ^ 0x80000000u;
      if ((u & hiMask) == (prefix & hiMask) && !((u >> bit) & 1u)) ++zeros;
    }
    zeros = blockCount(zeros);
    if (k >= zeros) { k -= zeros; prefix |= 1u << bit; }
  }
  const int32_t median = (int32_t)(prefix ^ 0x80000000u);
  double cutoff = median * 0.01;
  if (cutoff < 1) cutoff = 1;
  mine = 0;
  for (int i = tid; i < len; i += WG)
    if (t1k_bit(ref.exon, g0 + i) && !((double)cv[i] >= cutoff)) ++mine;
  const uint32_t miss = blockCount(mine);
  if (tid == 0) missing[a] = (int32_t)miss;
}
void t1k_launch_missing_coverage(t1k_ctx *ctx, const T1kRefDev &ref, int32_t *scratch, int32_t *missing) {
  hipLaunchKernelGGL(k_missing_coverage, dim3(ref.nAlleles), dim3(WG), 0, ctx->stream, ref, scratch, missing);
}
void t1k_launch_coverage_scan(t1k_ctx *ctx, const T1kRefDev &ref, int32_t *out, const uint64_t *outOff) {
  hipLaunchKernelGGL(k_coverage_scan, dim3(ref.nAlleles), dim3(WG), 0, ctx->stream, ref, out, outOff);
}
