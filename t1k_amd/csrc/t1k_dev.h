// t1k_amd/csrc/t1k_dev.h -- internal: device context, HBM layouts, bit helpers and the banded-alignment device
// routines shared by the gfx950 kernels.  wave64 / CDNA4 only; no CUDA-compat paths.
//
// HBM layouts (all little-endian within a word):
//   bases : uint64 words, 32 bases per word, base i of a sequence at bits [2*(i&31), +2) of word i>>5 (A0 C1 G2 T3)
//   masks : uint64 words with the SAME geometry, bit 2*(i&31) set if the property holds at base i (N mask, exon mask);
//           sharing the geometry lets one funnel shift serve bases and masks, and mask & mismatch-bits is a plain AND.
//   reference: all alleles concatenated, every allele starts on a 32-base boundary; alleleOff[a] = global base offset.
//   reads: per read-end a fixed stride of S words: [fwd bases | rc bases] and [fwd N | rc N].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <type_traits>
#include <vector>
#include "../../include/t1k_gpu.h"

// Longest read of the FAST kernels: the hit-offset masks of the chain kernels span 320 positions (10 words), k_extract_screen gives a
// lane five consecutive k-mers and k_extract a thread three positions of both strands.  The genotyper / analyzer take reads of up to
// T1K_LONG_READ_LEN bases: a window (upload) that holds a read beyond T1K_MAX_READ_LEN seeds those read-ends with k_seed_long (hit
// counts instead of masks; every group takes the explicit-hit-list path of the multi-diagonal groups) and runs the selection kernels
// with wider sort-key fields; the alignment routines themselves do not depend on the length.  1000: read coordinates < 1024 and match
// counts < 4096 in the packed overlap record (T1kOvlP), 11-bit match counts and span sums in the sort keys.
#define T1K_MAX_READ_LEN 320
#define T1K_LONG_READ_LEN 1000
#define T1K_EVEN 0x5555555555555555ull
#define T1K_NEG_BIG (-(1 << 28))

// ------------------------------------------------------------------------------------------------------------------
// records living in HBM
// ------------------------------------------------------------------------------------------------------------------
struct T1kPosting { uint32_t allele, offset; };  // KmerIndex.hpp:12-17

// candidate overlap after chaining + seed-chain match counting (SeqSet.hpp:1524-1548, 1697-1848); 24 bytes
struct T1kCand {
  uint32_t allele;     // seqIdx
  uint32_t readSE;     // readStart | readEnd << 16
  int32_t seqStart, seqEnd;
  uint32_t match;      // matchCnt0 (= 2*hitLen, used for the strand vote) | matchCnt << 16
  uint32_t re;         // read-end id within the batch
};

// extension result per candidate (SeqSet::ExtendOverlap, SeqSet.hpp:1994-2100); 24 bytes
struct T1kExt {
  int32_t seqStart, seqEnd;
  uint16_t readStart, readEnd;
  uint16_t matchCnt;   // after extension, including clip credit
  uint16_t leftClip, rightClip;
  uint16_t flags;      // T1K_F_*
  uint32_t pad;
};
enum { T1K_F_DROP = 1, T1K_F_SEPSEED = 2, T1K_F_NEEDCLIP = 4, T1K_F_EXTOK = 8 };

// final overlap record kept per read-end (what SeqSet::AssignRead returns); 32 bytes
struct T1kOvl {
  uint32_t allele;
  int32_t seqStart, seqEnd;
  uint16_t readStart, readEnd;
  uint16_t matchCnt, relaxed;
  uint16_t leftClip, rightClip;
  uint32_t re;
  uint32_t flags;      // bit0 near-best, bit1 strand is '-', bit2 duplicate-allele list marker
};

// The same record as the overlap store keeps it (16 bytes: the store holds the lists of a whole window of fragments, tens of GB at
// 32 bytes each, and mate pairing streams every list it joins):
//   lo = allele:24 | seqStart:20 | (seqEnd - seqStart):12 | strand is '-':1 | near-best:1        hi = readStart:10 | readEnd:10 | matchCnt:12 | relaxed:12 | leftClip:10 | rightClip:10
// Ranges: alleles < 2^24 and allele lengths < 2^20 are enforced by t1k_ref_upload, reads are at most T1K_MAX_READ_LEN (320) long;
// t1k_ovl_pack reports anything that does not fit (an internal error, never silent).
struct T1kOvlP { unsigned long long lo, hi; };
__host__ __device__ __forceinline__ bool t1k_ovl_pack(const T1kOvl &o, T1kOvlP &p) {
  const int span = o.seqEnd - o.seqStart;
  const bool ok = o.allele < (1u << 24) && o.seqStart >= 0 && o.seqStart < (1 << 20) && span >= 0 && span < 4096 && o.readStart < 1024 && o.readEnd < 1024 &&
                  o.matchCnt < 4096 && o.relaxed < 4096 && o.leftClip < 1024 && o.rightClip < 1024;
  p.lo = (unsigned long long)o.allele | ((unsigned long long)(uint32_t)o.seqStart << 24) | ((unsigned long long)(uint32_t)span << 44) | ((unsigned long long)((o.flags >> 1) & 1u) << 56) |
         ((unsigned long long)(o.flags & 1u) << 57);
  p.hi = (unsigned long long)o.readStart | ((unsigned long long)o.readEnd << 10) | ((unsigned long long)o.matchCnt << 20) | ((unsigned long long)o.relaxed << 32) |
         ((unsigned long long)o.leftClip << 44) | ((unsigned long long)o.rightClip << 54);
  return ok;
}
__host__ __device__ __forceinline__ T1kOvl t1k_ovl_unpack(const T1kOvlP &p) {
  T1kOvl o;
  o.allele = (uint32_t)(p.lo & 0xFFFFFFu);
  o.seqStart = (int32_t)((p.lo >> 24) & 0xFFFFFu);
  o.seqEnd = o.seqStart + (int32_t)((p.lo >> 44) & 0xFFFu);
  o.flags = ((uint32_t)((p.lo >> 56) & 1u) << 1) | (uint32_t)((p.lo >> 57) & 1u);
  o.readStart = (uint16_t)(p.hi & 0x3FFu); o.readEnd = (uint16_t)((p.hi >> 10) & 0x3FFu);
  o.matchCnt = (uint16_t)((p.hi >> 20) & 0xFFFu); o.relaxed = (uint16_t)((p.hi >> 32) & 0xFFFu);
  o.leftClip = (uint16_t)((p.hi >> 44) & 0x3FFu); o.rightClip = (uint16_t)((p.hi >> 54) & 0x3FFu);
  o.re = 0;
  return o;
}

#ifndef T1K_SEED_CHUNK
#define T1K_SEED_CHUNK 512    // alleles per seeding chunk (k_seed_groups): accumulators of one chunk live in LDS
#endif
#define T1K_DIR_MINLEN 32     // posting lists longer than this get a chunk directory
#define T1K_NO_DIR 0xFFFFFFFFu
struct T1kRefDev {
  uint32_t nAlleles;
  uint64_t totalBases;          // padded global base count
  const uint64_t *bases, *nmask, *exon;
  // Round 6: a second copy of `bases`, TRANSPOSED in blocks of 64 consecutive alleles: word w of allele a lives at basesT[blockT[a >> 6] + w * 64 + (a & 63)].
  // The lanes of the closed-form pass hold consecutive alleles of a gene and read the same few words of each (the window under a read): in the
  // linear layout every lane's 48 bytes sit in a cache line of their own (an allele is ~290 bytes on), 3 - 4 x the bytes come over the fabric
  // and the L2 keeps lines of which a third is ever used; here a wavefront's 64 windows are six contiguous 512-byte rows.  NULL: not built
  // (T1K_REF_TRANSPOSE=0).  Rows behind an allele's last word hold zeros (the linear layout has the next allele there: neither is ever looked at).
  const uint64_t *basesT;
  const uint32_t *blockT;       // [ceil(A / 64)] first word of the block in basesT
  const uint64_t *posted;       // same geometry as nmask: bit 2(i&31) of word i>>5 = the k-mer STARTING at global position i has a posting (valid + the reference's insert rule, KmerIndex.hpp:121)
  const uint64_t *alleleOff;    // [A]
  const uint32_t *alleleLen;    // [A]
  const uint8_t *alleleHasN;    // [A] 1 if the allele holds an N anywhere (its N-mask words can be skipped otherwise)
  uint32_t anyN;                // some allele holds an N (0: nobody needs to load nmask words of the reference)
  const uint32_t *sepStart;     // [A+1] into sepPos: interior N positions only (the -1 / len sentinels are implicit)
  const int32_t *sepPos;
  const uint32_t *kStart;       // [4^k + 1]
  const uint32_t *kHas;         // [4^k / 32] bit c = posting list of code c is not empty (32 MB at k = 14: lives in the Infinity Cache)
  const uint32_t *kMulti;       // [4^k / 32] bit c = list c names some sequence more than once (the k-mer repeats inside a sequence)
  const uint32_t *kHasPre;      // [4^(k-2) / 32] bit p = some non-empty code c has p as the first k - 2 bases of min(c, revcomp(c)): one look-up
                                // screens a k-mer position for both strands (2 MB at k = 14: lives in L2)
  // chunk directory of the long posting lists: kDirIdx[code] = row or T1K_NO_DIR; row r, entry c = first posting of the list
  // whose allele is >= c * T1K_SEED_CHUNK (relative to the list start), c = 0 .. kDirStride - 1
  const uint32_t *kDirIdx;      // [4^k]
  const uint32_t *kDir;         // [rows][kDirStride]
  uint32_t kDirStride;
  // per directory row: bit c = the list holds a posting of chunk c (kDir[row][c + 1] > kDir[row][c]); kDirMaskWords 64-bit words a row
  const unsigned long long *kDirMask;
  uint32_t kDirMaskWords;
  const T1kPosting *kPost;
  const uint32_t *kPostAllele;  // the allele column of kPost on its own: the extractor's vote streams only this
  // per-base coverage = prefix sum of covDiff (+w where a covered run starts, -w behind its end) minus covHole (w at a position
  // inside an ungapped alignment's span that is not covered: a mismatch or an N).  covHole = covDiff + covStride.
  // covFull = covDiff + 2 * covStride counts the covered runs of exactly the context's covFullLen positions by their start (one
  // atomic instead of the +w / -w pair -- most alignments span the whole read); t1k_coverage_fold adds them into covDiff before
  // anything reads the arrays.
  int32_t *covDiff;             // [3][covStride]
  uint64_t covStride;           // totalBases + 2
};

struct T1kReadsDev {
  uint32_t nReadEnds;
  int S;                        // words per strand per read-end
  const uint64_t *bases;        // [re][2][S]  (0 = forward, 1 = reverse complement)
  const uint64_t *nmask;        // [re][2][S]
  const uint16_t *len;
  const uint32_t *weight;
  // the final overlap list of every read-end assigned since the read set was uploaded (published by t1k_assign_range, read by
  // k_pair): device address of its first T1kOvl record and its length.  Lists live in the overlap stores of whichever context
  // (pipeline) assigned the read-end; all contexts that alias one read set share this table.
  unsigned long long *listPtr;  // [nReadEnds]
  uint32_t *listCount;          // [nReadEnds]
  // [nReadEnds] or NULL: 1 = the read-end's sequence was assigned in an earlier window of the job that is still resident; it is not
  // seeded again and its table entry is filled from that window's (t1k_xwin_link / t1k_xwin_resolve, t1k_dedupe.hip)
  const uint8_t *skip;
};

// ------------------------------------------------------------------------------------------------------------------
// bit helpers
// ------------------------------------------------------------------------------------------------------------------
// 32 consecutive positions starting at position pos (pos >= 0) of a 2-bit-per-position stream; the arrays carry
// one spare word so w[wi+1] is always addressable.
// two consecutive stream words in one 16-byte access (the streams are only 8-byte aligned; gfx950 global loads allow that)
typedef uint64_t t1k_u64x2 __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ uint64_t t1k_get32(const uint64_t *w, int64_t pos) {
  int64_t wi = pos >> 5;
  int sh = (int)(pos & 31) * 2;
  const t1k_u64x2 v = *(const t1k_u64x2 *)(w + wi);  // one 16-byte request for both words (a lane's address is its own: the request count is what these kernels pay for); branch-free
  return (v.x >> sh) | ((v.y << 1) << (63 - sh));
}
// reverse complement of a k-mer code (first base in the low bits): the code of the same window read on the other strand
__host__ __device__ __forceinline__ uint32_t t1k_code_revcomp(uint32_t code, int k) {
  uint32_t x = code;
  x = ((x >> 16) | (x << 16));
  x = ((x & 0xFF00FF00u) >> 8) | ((x & 0x00FF00FFu) << 8);
  x = ((x & 0xF0F0F0F0u) >> 4) | ((x & 0x0F0F0F0Fu) << 4);
  x = ((x & 0xCCCCCCCCu) >> 2) | ((x & 0x33333333u) << 2);  // 2-bit groups reversed, bits inside a group kept
  return (~x) >> (32 - 2 * k);
}
__device__ __forceinline__ uint64_t t1k_lowmask(int nPos) {  // mask of the first nPos positions (0..32)
  return nPos >= 32 ? ~0ull : ((1ull << (2 * nPos)) - 1);
}
__device__ __forceinline__ int t1k_base(const uint64_t *w, int64_t pos) { return (int)((w[pos >> 5] >> ((pos & 31) * 2)) & 3); }
__device__ __forceinline__ int t1k_bit(const uint64_t *w, int64_t pos) { return (int)((w[pos >> 5] >> ((pos & 31) * 2)) & 1); }

// mismatch bits (even positions) between 32 read positions and 32 reference positions; an N on either side matches
// (AlignAlgo.hpp:304-305)
__device__ __forceinline__ uint64_t t1k_mm32(const uint64_t *rb, const uint64_t *rn, int64_t rpos, const uint64_t *gb, const uint64_t *gn,
                                               int64_t gpos) {
  uint64_t x = t1k_get32(rb, rpos) ^ t1k_get32(gb, gpos);
  uint64_t m = (x | (x >> 1)) & T1K_EVEN;
  return m & ~(t1k_get32(rn, rpos) | t1k_get32(gn, gpos));
}

// number of mismatching columns of the ungapped comparison of L positions
__device__ __forceinline__ int t1k_hamming(const uint64_t *rb, const uint64_t *rn, int64_t rpos, const uint64_t *gb, const uint64_t *gn,
                                            int64_t gpos, int L) {
  int x = 0;
  for (int o = 0; o < L; o += 32) {
    uint64_t m = t1k_mm32(rb, rn, rpos + o, gb, gn, gpos + o) & t1k_lowmask(L - o);
    x += __popcll(m);
  }
  return x;
}

// SeqSet::IsLowComplexity over read positions [rs, re] (SeqSet.hpp:458-485)
__device__ inline bool t1k_low_complexity(const uint64_t *rb, const uint64_t *rn, int rs, int re) {  // SeqSet.hpp:458-485
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

// do two L-position windows of the reference hold the same bases and N marks?
__device__ inline bool t1k_same_window(const uint64_t *gb, const uint64_t *gn, int64_t a, int64_t b, int L, bool anyN = true) {
  if (a == b) return true;
  for (int o = 0; o < L; o += 32) {
    uint64_t lm = t1k_lowmask(L - o);
    uint64_t x = t1k_get32(gb, a + o) ^ t1k_get32(gb, b + o);
    if (anyN) x |= t1k_get32(gn, a + o) ^ t1k_get32(gn, b + o);
    if (x & lm) return false;
  }
  return true;
}

// accessor for one sequence operand of the alignment routines
struct T1kSeqView {
  const uint64_t *b, *n;
  int64_t pos;
  __device__ __forceinline__ int code(int i) const {  // 0..3, or 4 for N
    int64_t p = pos + i;
    return t1k_bit(n, p) ? 4 : t1k_base(b, p);
  }
};
__device__ __forceinline__ bool t1k_eq(int a, int b) { return a == b || a == 4 || b == 4; }

// ------------------------------------------------------------------------------------------------------------------
// AlignAlgo::GlobalAlignment (AlignAlgo.hpp:215-421), equal lengths L >= 2, band +-5: number of MATCH columns on the
// reference's traceback path, computed in one forward sweep.  cnt*[cell] = matches on the traceback path from
// (cell, state) to the origin; the traceback's choice at a cell is local (diagonal if it reproduces m, else delete
// if f >= e, else insert; "gap opened here" iff m(prev)-5 == e/f), so the counts obey the same recurrences as the
// scores.  Band columns live in registers (d = j - i in [-6, 6], the outer two being the negInf fence cells).
// ------------------------------------------------------------------------------------------------------------------
__device__ inline int t1k_ga_matches_equal(const T1kSeqView &T, const T1kSeqView &P, int L, int *scoreOut) {
  const int negInf = (L + 1) * (L + 1) * -4;
  int m[13], e[13], cm[13], ce[13];
  // row 0: boundary values for every column (AlignAlgo.hpp:255-270); e[0][j] uses the stale i == lenp+1
#pragma unroll
  for (int s = 0; s < 13; ++s) {
    int j = s - 6;
    if (j == 0) { m[s] = 0; e[s] = 0; }
    else { m[s] = -4 - 4 * j; e[s] = -4 - 4 * (L + 1); }
    cm[s] = 0; ce[s] = 0;
  }
  uint64_t pwB = 0, pwN = 0;
  // one row of the band.  INTERIOR rows (7 <= i <= L - 5) have all eleven columns inside the matrix: no boundary tests, the
  // text window starts exactly at column i - 5, and the eleven base comparisons are done at once on the packed window
  auto row = [&](int i, auto interiorTag) {
    constexpr bool INTERIOR = decltype(interiorTag)::value;
    // pattern base i-1: one 32-base window per 32 rows; text bases i-6 .. i+4: one window per row (2 loads, L1-resident)
    if (((i - 1) & 31) == 0) { pwB = t1k_get32(P.b, P.pos + i - 1); pwN = t1k_get32(P.n, P.pos + i - 1); }
    const int pq = ((i - 1) & 31) * 2;
    const bool pIsN = (pwN >> pq) & 1;
    const int pc = pIsN ? 4 : (int)((pwB >> pq) & 3);
    const int t0 = i - 6 > 0 ? i - 6 : 0;  // first text index covered by the window
    const uint64_t twB = t1k_get32(T.b, T.pos + t0), twN = t1k_get32(T.n, T.pos + t0);
    uint64_t eqm = 0;  // INTERIOR: bit 2(s-1) = text base of slot s compares equal to the pattern base (N matches anything)
    if (INTERIOR) {
      const uint64_t x = twB ^ ((uint64_t)(pc & 3) * T1K_EVEN);
      eqm = pIsN ? ~0ull : (~((x | (x >> 1))) | twN);
    }
    int fLeft = negInf, mLeft = negInf, cfLeft = 0, cmLeft = 0;  // cell (i, j-1) of the current row
#pragma unroll
    for (int s = 0; s < 13; ++s) {
      int j = i + s - 6;
      int nm, ne, nf, ncm, nce, ncf;
      if (!INTERIOR && j < 0) { nm = ne = nf = negInf; ncm = nce = ncf = 0; }
      else if (!INTERIOR && j == 0) { nm = -4 - 4 * i; ne = -4 - i; nf = -4 - 4 * i; ncm = nce = ncf = 0; }  // column 0 (also in slot 0, row 6)
      else if (s == 0 || s == 12 || (!INTERIOR && j > L)) { nm = ne = nf = negInf; ncm = nce = ncf = 0; }
      else {
        // e: from (i-1, j) = previous row at slot s+1
        int eu = e[s + 1] - 1, mu = m[s + 1] - 5;
        ne = eu > mu ? eu : mu;
        nce = (mu == ne) ? cm[s + 1] : ce[s + 1];
        // f: from (i, j-1)
        int fl = fLeft - 1, ml = mLeft - 5;
        nf = fl > ml ? fl : ml;
        ncf = (ml == nf) ? cmLeft : cfLeft;
        // m: diagonal (i-1, j-1) = previous row at slot s
        bool eq;
        if (INTERIOR) eq = (eqm >> (2 * (s - 1))) & 1;
        else {
          const int tq = (j - 1 - t0) * 2;
          const int tc = ((twN >> tq) & 1) ? 4 : (int)((twB >> tq) & 3);
          eq = t1k_eq(tc, pc);
        }
        int dg = m[s] + (eq ? 2 : -2);
        nm = dg;
        if (ne > nm) nm = ne;
        if (nf > nm) nm = nf;
        if (dg == nm) ncm = cm[s] + (eq ? 1 : 0);
        else if (nf >= ne) ncm = ncf;
        else ncm = nce;
      }
      // slot s of the previous row is dead now (slot s+1 is still needed by the next iteration)
      m[s] = nm; e[s] = ne; cm[s] = ncm; ce[s] = nce;
      fLeft = nf; mLeft = nm; cfLeft = ncf; cmLeft = ncm;
    }
  };
  int i = 1;
  for (; i <= L && i < 7; ++i) row(i, std::false_type{});
  for (; i <= L - 5; ++i) row(i, std::true_type{});
  for (; i <= L; ++i) row(i, std::false_type{});
  if (scoreOut) *scoreOut = m[6];
  return cm[6];
}

// Same sweep as t1k_ga_matches_equal, additionally recording the traceback decisions: one 64-bit word per row, 5 bits per
// in-band cell (slot s = j - i + 6 in 1..11 at bits 5*(s-1)): b0 diagonal reproduces m, b1 f >= e, b2 e opened from m,
// b3 f opened from m, b4 the two bases compare equal.  Words go to trace[i * stride] (i = 1..L) so that a wavefront's
// stores coalesce.  Boundary cells (row 0 / column 0) need no storage: their decisions are closed-form (see the walker).
__device__ inline void t1k_ga_equal_traced(const T1kSeqView &T, const T1kSeqView &P, int L, uint64_t *trace, size_t stride) {
  const int negInf = (L + 1) * (L + 1) * -4;
  int m[13], e[13];
#pragma unroll
  for (int s = 0; s < 13; ++s) {
    int j = s - 6;
    if (j == 0) { m[s] = 0; e[s] = 0; }
    else { m[s] = -4 - 4 * j; e[s] = -4 - 4 * (L + 1); }
  }
  uint64_t pwB = 0, pwN = 0;
  auto row = [&](int i, auto interiorTag) {  // see t1k_ga_matches_equal
    constexpr bool INTERIOR = decltype(interiorTag)::value;
    if (((i - 1) & 31) == 0) { pwB = t1k_get32(P.b, P.pos + i - 1); pwN = t1k_get32(P.n, P.pos + i - 1); }
    const int pq = ((i - 1) & 31) * 2;
    const bool pIsN = (pwN >> pq) & 1;
    const int pc = pIsN ? 4 : (int)((pwB >> pq) & 3);
    const int t0 = i - 6 > 0 ? i - 6 : 0;
    const uint64_t twB = t1k_get32(T.b, T.pos + t0), twN = t1k_get32(T.n, T.pos + t0);
    uint64_t eqm = 0;
    if (INTERIOR) {
      const uint64_t x = twB ^ ((uint64_t)(pc & 3) * T1K_EVEN);
      eqm = pIsN ? ~0ull : (~((x | (x >> 1))) | twN);
    }
    int fLeft = negInf, mLeft = negInf;
    uint64_t word = 0;
#pragma unroll
    for (int s = 0; s < 13; ++s) {
      int j = i + s - 6;
      int nm, ne, nf;
      if (!INTERIOR && j < 0) { nm = ne = nf = negInf; }
      else if (!INTERIOR && j == 0) { nm = -4 - 4 * i; ne = -4 - i; nf = -4 - 4 * i; }
      else if (s == 0 || s == 12 || (!INTERIOR && j > L)) { nm = ne = nf = negInf; }
      else {
        int eu = e[s + 1] - 1, mu = m[s + 1] - 5;
        ne = eu > mu ? eu : mu;
        int fl = fLeft - 1, ml = mLeft - 5;
        nf = fl > ml ? fl : ml;
        bool eq;
        if (INTERIOR) eq = (eqm >> (2 * (s - 1))) & 1;
        else {
          const int tq = (j - 1 - t0) * 2;
          const int tc = ((twN >> tq) & 1) ? 4 : (int)((twB >> tq) & 3);
          eq = t1k_eq(tc, pc);
        }
        int dg = m[s] + (eq ? 2 : -2);
        nm = dg;
        if (ne > nm) nm = ne;
        if (nf > nm) nm = nf;
        uint64_t bits = (dg == nm ? 1u : 0u) | (nf >= ne ? 2u : 0u) | (mu == ne ? 4u : 0u) | (ml == nf ? 8u : 0u) | (eq ? 16u : 0u);
        word |= bits << (5 * (s - 1));
      }
      m[s] = nm; e[s] = ne;
      fLeft = nf; mLeft = nm;
    }
    trace[(size_t)i * stride] = word;
  };
  int i = 1;
  for (; i <= L && i < 7; ++i) row(i, std::false_type{});
  for (; i <= L - 5; ++i) row(i, std::true_type{});
  for (; i <= L; ++i) row(i, std::false_type{});
}

// Banded GlobalAlignment for |lent - lenp| <= DMAX with the whole band in registers (generalises the two routines above:
// leftBand = 5 + max(0, lenp - lent), rightBand = 5 + max(0, lent - lenp), AlignAlgo.hpp:240-245).  Slot s of a row holds
// column j = i + s - (LB + 1); slots 0 and LB + RB + 2 are the negInf fence cells.
//   TRACE = false: returns the number of MATCH columns of the reference's traceback (forward-carried counts).
//   TRACE = true : stores one 64-bit word of 4-bit decisions per row (b0 diagonal reproduces m, b1 f >= e, b2 e opened from
//                  m, b3 f opened from m) at trace[i * stride]; whether two bases compare equal is re-derived by the walker.
template <int DMAX, bool TRACE>
__device__ inline int t1k_ga_band(const T1kSeqView &T, int lent, const T1kSeqView &P, int lenp, uint64_t *trace, size_t stride) {
  constexpr int NS = 13 + DMAX;
  const int LB = 5 + (lenp > lent ? lenp - lent : 0), RB = 5 + (lent > lenp ? lent - lenp : 0);
  const int last = LB + RB + 2;  // right fence slot
  const int negInf = (lent + 1) * (lenp + 1) * -4;
  int m[NS], e[NS], cm[NS], ce[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {  // row 0: boundary values for every column (AlignAlgo.hpp:255-270)
    int j = s - (LB + 1);
    if (j == 0) { m[s] = 0; e[s] = 0; }
    else { m[s] = -4 - 4 * j; e[s] = -4 - 4 * (lenp + 1); }
    cm[s] = 0; ce[s] = 0;
  }
  uint64_t pwB = 0, pwN = 0;
  for (int i = 1; i <= lenp; ++i) {
    if (((i - 1) & 31) == 0) { pwB = t1k_get32(P.b, P.pos + i - 1); pwN = t1k_get32(P.n, P.pos + i - 1); }
    const int pq = ((i - 1) & 31) * 2;
    const int pc = ((pwN >> pq) & 1) ? 4 : (int)((pwB >> pq) & 3);
    const int t0 = i - LB - 1 > 0 ? i - LB - 1 : 0;
    const uint64_t twB = t1k_get32(T.b, T.pos + t0), twN = t1k_get32(T.n, T.pos + t0);
    int fLeft = 0, mLeft = 0, cfLeft = 0, cmLeft = 0;
    uint64_t word = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      int j = i + s - (LB + 1);
      int nm, ne, nf, ncm = 0, nce = 0, ncf = 0;
      if (s > last || j < 0) { nm = ne = nf = negInf; }
      else if (j == 0) { nm = -4 - 4 * i; ne = -4 - i; nf = -4 - 4 * i; }
      else if (j > lent || s == 0 || s == last) { nm = ne = nf = negInf; }
      else {
        const int up = s + 1 < NS ? s + 1 : s;  // (i-1, j); s + 1 <= last < NS whenever this branch is live
        int eu = e[up] - 1, mu = m[up] - 5;
        ne = eu > mu ? eu : mu;
        int fl = fLeft - 1, ml = mLeft - 5;
        nf = fl > ml ? fl : ml;
        const int tq = (j - 1 - t0) * 2;
        const int tc = ((twN >> tq) & 1) ? 4 : (int)((twB >> tq) & 3);
        bool eq = t1k_eq(tc, pc);
        int dg = m[s] + (eq ? 2 : -2);
        nm = dg;
        if (ne > nm) nm = ne;
        if (nf > nm) nm = nf;
        if (TRACE) {
          uint64_t bits = (dg == nm ? 1u : 0u) | (nf >= ne ? 2u : 0u) | (mu == ne ? 4u : 0u) | (ml == nf ? 8u : 0u);
          word |= bits << (4 * (s - 1));
        } else {
          nce = (mu == ne) ? cm[up] : ce[up];
          ncf = (ml == nf) ? cmLeft : cfLeft;
          if (dg == nm) ncm = cm[s] + (eq ? 1 : 0);
          else if (nf >= ne) ncm = ncf;
          else ncm = nce;
        }
      }
      m[s] = nm; e[s] = ne;
      if (!TRACE) { cm[s] = ncm; ce[s] = nce; }
      fLeft = nf; mLeft = nm; cfLeft = ncf; cmLeft = ncm;
    }
    if (TRACE) trace[(size_t)i * stride] = word;
  }
  // final cell (lenp, lent): slot lent - lenp + LB + 1
  int res = 0;
#pragma unroll
  for (int s = 0; s < NS; ++s)
    if (s == lent - lenp + LB + 1) res = cm[s];
  return res;
}

// Exact fast path for equal lengths: with x <= 3 mismatches the ungapped alignment is optimal and is the one the
// traceback returns (any gapped alignment of equal-length strings scores <= 2L-12 <= 2L-4x, ties go to the diagonal).
__device__ __forceinline__ int t1k_ga_matches_window(const uint64_t *rb, const uint64_t *rn, int64_t rpos, const uint64_t *gb, const uint64_t *gn,
                                                      int64_t gpos, int L, unsigned int *dpCounter) {
  if (L <= 0) return 0;
  int x = t1k_hamming(rb, rn, rpos, gb, gn, gpos, L);
  if (x <= 3) return L - x;
  if (dpCounter) ++*dpCounter;  // thread-local tally, flushed once per workgroup by the caller
  T1kSeqView T{gb, gn, gpos}, P{rb, rn, rpos};
#ifdef T1K_PHASE_TIMERS
  long long t0_ = clock64();
  int r_ = t1k_ga_matches_equal(T, P, L, nullptr);
  if (dpCounter) { dpCounter[1] += (unsigned int)((clock64() - t0_) >> 6); dpCounter[2] += (unsigned int)L; }
  return r_;
#else
  return t1k_ga_matches_equal(T, P, L, nullptr);
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// General GlobalAlignment (any lent, lenp) with row arrays in caller-provided scratch (6 * (lent + 2) ints).
// Returns the score; *nMatch etc. the column counts of the traceback path.  If trace != NULL, one decision byte per
// cell is stored at trace[i * (lent + 1) + j] (bit0 diagonal reproduces m, bit1 f >= e, bit2 e opened from m,
// bit3 f opened from m, bit4 columns compare equal) for t1k_ga_traceback().
// ------------------------------------------------------------------------------------------------------------------
__device__ inline int t1k_ga_general(const T1kSeqView &T, int lent, const T1kSeqView &P, int lenp, int *scratch, uint8_t *trace, int *nMatch) {
  if (lent == 0 || lenp == 0) { if (nMatch) *nMatch = 0; return 0; }
  if (lent == 1 && lenp == 1) {
    bool eq = t1k_eq(T.code(0), P.code(0));
    if (nMatch) *nMatch = eq ? 1 : 0;
    if (trace) { trace[0] = 0; trace[1 * 2 + 1] = (uint8_t)(1 | (eq ? 16 : 0)); }
    return eq ? 2 : -2;
  }
  int leftBand = 5, rightBand = 5;
  if (lent > lenp) rightBand += lent - lenp; else if (lent < lenp) leftBand += lenp - lent;
  const int W = lent + 1;
  const int negInf = (lent + 1) * (lenp + 1) * -4;
  int *m = scratch, *e = scratch + (W + 1), *f = scratch + 2 * (W + 1);
  int *cm = scratch + 3 * (W + 1), *ce = scratch + 4 * (W + 1), *cf = scratch + 5 * (W + 1);
  m[0] = e[0] = f[0] = 0; cm[0] = ce[0] = cf[0] = 0;
  for (int j = 1; j <= lent; ++j) { f[j] = -4 - j; e[j] = -4 - 4 * (lenp + 1); m[j] = -4 - 4 * j; cm[j] = ce[j] = cf[j] = 0; }
  if (trace) {
    for (int j = 0; j <= lent; ++j) {
      // row 0: no diagonal; bit1 = f >= e; bit3 = m[0][j-1]-5 == f[0][j]
      uint8_t t = 0;
      if (j > 0) {
        if (f[j] >= e[j]) t |= 2;
        int mprev = (j - 1 == 0) ? 0 : -4 - 4 * (j - 1);
        if (mprev - 5 == f[j]) t |= 8;
      }
      trace[j] = t;
    }
  }
  for (int i = 1; i <= lenp; ++i) {
    int start = (i - leftBand < 1) ? 1 : (i - leftBand);
    int end = (i + rightBand > lent) ? lent : (i + rightBand);
    int pc = P.code(i - 1);
    // previous-row values are consumed left to right; keep the diagonal predecessor before overwriting
    int mDiag, cmDiag;      // (i-1, j-1)
    int mL, fL, cmL, cfL;   // (i, j-1)
    if (start == 1) {
      mDiag = (i - 1 == 0) ? 0 : -4 - 4 * (i - 1);  // m[i-1][0]
      cmDiag = 0;
      mL = -4 - 4 * i; fL = -4 - 4 * i; cmL = 0; cfL = 0;  // (i, 0)
      // column 0 of this row also becomes visible to row i+1 as (i, 0): write it
    } else {
      mDiag = m[start - 1]; cmDiag = cm[start - 1];
      mL = negInf; fL = negInf; cmL = 0; cfL = 0;  // the fence cell (i, start-1)
    }
    int saveM0 = m[0], saveE0 = e[0];
    (void)saveM0; (void)saveE0;
    // column 0 / fence bookkeeping for the next row
    if (start == 1) { /* (i,0) written after the sweep */ }
    for (int j = start; j <= end; ++j) {
      int eu = e[j] - 1, mu = m[j] - 5;  // (i-1, j): still the previous row
      int ne = eu > mu ? eu : mu;
      int nce = (mu == ne) ? cm[j] : ce[j];
      int fl = fL - 1, ml = mL - 5;
      int nf = fl > ml ? fl : ml;
      int ncf = (ml == nf) ? cmL : cfL;
      bool eq = t1k_eq(T.code(j - 1), pc);
      int dg = mDiag + (eq ? 2 : -2);
      int nm = dg;
      if (ne > nm) nm = ne;
      if (nf > nm) nm = nf;
      int ncm;
      if (dg == nm) ncm = cmDiag + (eq ? 1 : 0);
      else if (nf >= ne) ncm = ncf;
      else ncm = nce;
      if (trace) {
        uint8_t t = 0;
        if (dg == nm) t |= 1;
        if (nf >= ne) t |= 2;
        if (mu == ne) t |= 4;
        if (ml == nf) t |= 8;
        if (eq) t |= 16;
        trace[i * W + j] = t;
      }
      mDiag = m[j]; cmDiag = cm[j];  // becomes (i-1, j) -> diagonal for j+1
      m[j] = nm; e[j] = ne; f[j] = nf; cm[j] = ncm; ce[j] = nce; cf[j] = ncf;
      mL = nm; fL = nf; cmL = ncm; cfL = ncf;
    }
    if (end < lent) { m[end + 1] = e[end + 1] = f[end + 1] = negInf; cm[end + 1] = ce[end + 1] = cf[end + 1] = 0; }
    if (start > 1) { m[start - 1] = e[start - 1] = f[start - 1] = negInf; cm[start - 1] = ce[start - 1] = cf[start - 1] = 0; }
    // column 0 of row i (boundary values, AlignAlgo.hpp:256-262) for the next row's j == 1
    m[0] = -4 - 4 * i; e[0] = -4 - i; f[0] = -4 - 4 * i; cm[0] = ce[0] = cf[0] = 0;
    if (trace) {
      // (i, 0): no diagonal; f >= e only when i <= 0; bit2 = m[i-1][0]-5 == e[i][0]
      uint8_t t = 0;
      if (f[0] >= e[0]) t |= 2;
      int mprev = (i - 1 == 0) ? 0 : -4 - 4 * (i - 1);
      if (mprev - 5 == e[0]) t |= 4;
      trace[i * W] = t;
    }
  }
  if (nMatch) *nMatch = cm[lent];
  return m[lent];
}

// Replays AlignAlgo.hpp:323-415 on the decision bytes written by t1k_ga_general.  ops (capacity lent+lenp+2) receives
// the edit string in forward order; returns its length.
__device__ inline int t1k_ga_traceback(const uint8_t *trace, int lent, int lenp, int8_t *ops) {
  if (lent == 0 || lenp == 0) return 0;
  const int W = lent + 1;
  int ti = lenp, tj = lent, mat = 0, n = 0;
  if (lent == 1 && lenp == 1) { ops[0] = (trace[1 * 2 + 1] & 16) ? 0 : 1; return 1; }
  while (ti > 0 || tj > 0) {
    uint8_t t = trace[ti * W + tj];
    if (mat == 0) {
      if (ti > 0 && tj > 0 && (t & 1)) { ops[n++] = (t & 16) ? 0 : 1; --ti; --tj; }
      else if (t & 2) mat = 2;
      else mat = 1;
    } else if (mat == 1) {
      ops[n++] = 2;
      if (ti > 0) { if (t & 4) mat = 0; --ti; }
      else mat = 2;
    } else {
      ops[n++] = 3;
      if (tj > 0) { if (t & 8) mat = 0; --tj; }
      else mat = 1;
    }
  }
  for (int a = 0, b = n - 1; a < b; ++a, --b) { int8_t x = ops[a]; ops[a] = ops[b]; ops[b] = x; }
  return n;
}

// Statistics counters (never read by device code).  Millions of atomics on one address serialise in L2, so every
// statistic is striped over T1K_STAT_STRIPES cache lines behind the 64 control counters; the host adds the stripes up.
#define T1K_STAT_STRIPES 256
enum { T1K_STAT_DP = 0, T1K_STAT_FAST = 1, T1K_STAT_GENERAL = 2, T1K_STAT_EXTEND_DP = 3, T1K_STAT_NEARBEST = 4, T1K_STAT_LOOKUPS = 5, T1K_STAT_POSTINGS = 6, T1K_STAT_HITS = 7 };
// Allocation cursors.  A returning atomic on ONE word tops out near 88 M/s on this part, far below what the chain kernels
// ask for, so every device arena (group records, work lists, queues) is cut into T1K_NSTRIPE independent segments with
// their own cursor (64 bytes apart); a workgroup allocates from segment blockIdx.x % T1K_NSTRIPE.  Lists are made dense
// again by k_arena_compact before their consumer runs; group records are consumed segment by segment.
// latency-bound kernels: let the register allocator aim for 8 wavefronts per SIMD (<= 64 VGPRs; a few spills are cheaper than
// half the occupancy -- measured per kernel)
#ifndef T1K_OCC8
#define T1K_OCC8 __attribute__((amdgpu_waves_per_eu(8)))
#endif
// (round 6) T1K_<KERNEL>_WAVES: the wavefronts per SIMD the register allocation of a kernel aims for; 0 = the compiler's own choice, which is what
// ships for these four.  Measured (profiles/r06_callG_occupancy_variants.log, one pipeline, per range): k_collect at 7: 0.98 -> 0.94 ms, k_extend at 7:
// 0.85 -> 0.79 ms, k_truncate small at 6 / 7: no change, large at 8: 0.61 -> 0.65 ms -- and k_chain_fast<5, 0> at 7 (72 VGPRs, 32 of its registers
// spilled): 1.90 -> 2.42 ms AND WRONG, VARYING RESULTS in every configuration (one pipeline, host-driven chain, kernels serialised:
// profiles/r06_callH_occupancy_bisect.log, r06_callI_closed_form_spill_variant.log).  The kernel was re-read for values used before they are set
// and for out-of-range local indices without a finding; the build that ships keeps it in registers (78 VGPRs, no scratch) and is the one every test
// and reference hash covers.  Budgets are therefore only ever RAISED towards what a kernel's LDS admits (k_seed_groups, k_select's small shape:
// fewer spills than before), never lowered.
#define T1K_WAVES_ATTR_(n) __attribute__((amdgpu_waves_per_eu(n)))
// read-ends a workgroup takes per atomic on its kernel's hand-out word (round 6, profiles/r06_callL_hot_word_atomics.log, one pipeline, per range of 32 768
// read-ends).  A returning atomic on one word saturates near 88 M/s on this part: 32 768 hand-outs are 0.37 ms, which is what a kernel that SKIPS most read-ends
// lasts -- k_truncate's 256-thread shape (lists of 1001 .. 2048 overlaps): 0.450 ms one at a time, 0.267 two, 0.223 four, 0.234 eight.  Kernels that work on
// every read-end trade it against balance: k_select's 256-thread shape 0.509 / 0.478 / 0.503 / 0.550 ms, its 1024-thread shape 0.957 / 0.94 / 0.97 / 1.00,
// k_truncate's 1024-thread shape 0.615 / 0.604 / 0.62 / 0.64, k_collect 0.98 / 1.19 / 1.38 / 1.56 (its read-ends' cost spans orders of magnitude).
#ifndef T1K_COLLECT_HANDOUT
#define T1K_COLLECT_HANDOUT 1
#endif
#ifndef T1K_SELECT_SMALL_HANDOUT
#define T1K_SELECT_SMALL_HANDOUT 2
#endif
#ifndef T1K_SELECT_LARGE_HANDOUT
#define T1K_SELECT_LARGE_HANDOUT 1
#endif
#ifndef T1K_TRUNC_SMALL_HANDOUT
#define T1K_TRUNC_SMALL_HANDOUT 4
#endif
#ifndef T1K_TRUNC_LARGE_HANDOUT
#define T1K_TRUNC_LARGE_HANDOUT 1
#endif
#ifndef T1K_COLLECT_WAVES
#define T1K_COLLECT_WAVES 0
#endif
#ifndef T1K_EXTEND_WAVES
#define T1K_EXTEND_WAVES 0
#endif
#ifndef T1K_CF0_WAVES
#define T1K_CF0_WAVES 0
#endif
#ifndef T1K_TRUNC_SMALL_WAVES
#define T1K_TRUNC_SMALL_WAVES 0
#endif
#ifndef T1K_TRUNC_LARGE_WAVES
#define T1K_TRUNC_LARGE_WAVES 0
#endif
#define T1K_NSTRIPE 32
enum { T1K_AR_GROUPS = 0, T1K_AR_JOBS, T1K_AR_RETRY, T1K_AR_FINISH, T1K_AR_GENERAL, T1K_AR_BIG, T1K_AR_GENCAND, T1K_AR_EQ, T1K_AR_BAND, T1K_AR_WIDE, T1K_AR_GENHITS, T1K_AR_GENJOBS, T1K_AR_WAVE, T1K_AR_EXTJOBS, T1K_AR_EXTRETRY, T1K_AR_SLOW, T1K_NARENA };
#define T1K_ARENA_BASE (64 + T1K_STAT_STRIPES * 8)
// behind the cursors: one word per arena = the number of entries of its DENSE list (sum over the stripes of min(cursor, segCap)), written by
// k_arena_compact where it makes the list dense -- the list's consumers read their item count there instead of from a launch argument, so
// the host does not have to fetch the cursors between a producer and its consumers (t1k_run_chain: one counter fetch per range)
#ifndef T1K_STRIPE_WORDS
#define T1K_STRIPE_WORDS 8   // 64-bit words between the cursors of two stripes of an arena (8: 64 bytes apart; 16: a 128-byte cache line each)
#endif
#define T1K_TOTAL_BASE (T1K_ARENA_BASE + T1K_NARENA * T1K_NSTRIPE * T1K_STRIPE_WORDS)
#define T1K_COUNTER_WORDS (T1K_TOTAL_BASE + ((T1K_NARENA + 7) & ~7))
#define T1K_ARENA_FULL 0xFFFFFFFFu
__device__ __forceinline__ unsigned long long *t1k_arena_cursor(unsigned long long *counters, int arena, uint32_t stripe) {
  return counters + T1K_ARENA_BASE + ((uint32_t)arena * T1K_NSTRIPE + stripe) * T1K_STRIPE_WORDS;
}
// n consecutive slots in this workgroup's segment -> global slot index (segment * segCap + offset), or T1K_ARENA_FULL.
// The cursor keeps counting past segCap, which is how the host sees the overflow.
__device__ __forceinline__ uint32_t t1k_arena_alloc(unsigned long long *counters, int arena, uint32_t n, uint32_t segCap) {
  const uint32_t stripe = blockIdx.x & (T1K_NSTRIPE - 1);
  const unsigned long long off = atomicAdd(t1k_arena_cursor(counters, arena, stripe), (unsigned long long)n);
  return off + n <= segCap ? stripe * segCap + (uint32_t)off : T1K_ARENA_FULL;
}
// one slot per calling lane; the active lanes of the wavefront share one atomic (works in divergent code)
__device__ __forceinline__ uint32_t t1k_arena_append(unsigned long long *counters, int arena, uint32_t segCap) {
  const uint64_t m = __ballot(1);
  const int lane = threadIdx.x & 63;
  const uint32_t stripe = blockIdx.x & (T1K_NSTRIPE - 1);
  uint32_t base = 0;
  if (lane == __ffsll((long long)m) - 1) {
    const unsigned long long off = atomicAdd(t1k_arena_cursor(counters, arena, stripe), (unsigned long long)__popcll(m));
    base = off > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)off;
  }
  base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
  const uint64_t off = (uint64_t)base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
  return off < segCap ? stripe * segCap + (uint32_t)off : T1K_ARENA_FULL;
}
__device__ __forceinline__ void t1k_stat_add(unsigned long long *counters, int kind, unsigned int v) {  // call from converged code
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0 && v) {
    const unsigned int stripe = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (T1K_STAT_STRIPES - 1);
    atomicAdd(&counters[64 + stripe * 8 + kind], (unsigned long long)v);
  }
}

// Workgroup barrier for data that is exchanged through LDS only: it waits for this wavefront's LDS operations, not for its global
// loads and stores.  __syncthreads() fences global memory as well -- s_waitcnt vmcnt(0) before every s_barrier, and on this ISA stores
// count in vmcnt too -- so a barrier behind a burst of record writes, or with the next step's loads already requested, waits a full
// memory round trip for data nobody in the workgroup reads (measured: k_co_reduce_long 211 -> 31 ms, t1k_coalesce.hip).  Only for
// barriers with no global-memory communication between the threads of the workgroup across them.
__device__ __forceinline__ void t1k_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// exclusive prefix sum over a workgroup of NWAVES wavefronts; all threads must call it (warpSums: NWAVES words of LDS).  LDS_ONLY: its
// two barriers are t1k_lds_barrier()
template <int NWAVES, bool LDS_ONLY = false>
__device__ __forceinline__ uint32_t t1k_block_scan_exclusive_n(uint32_t v, uint32_t *warpSums, uint32_t *total) {
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) warpSums[wave] = x;
  if (LDS_ONLY) t1k_lds_barrier(); else __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NWAVES; ++w) { const uint32_t s = warpSums[w]; if (w < wave) base += s; tot += s; }
  if (LDS_ONLY) t1k_lds_barrier(); else __syncthreads();
  *total = tot;
  return base + x - v;
}
__device__ __forceinline__ uint32_t t1k_block_scan_exclusive(uint32_t v, uint32_t *warpSums, uint32_t *total) {  // 256 threads
  return t1k_block_scan_exclusive_n<4>(v, warpSums, total);
}

// ------------------------------------------------------------------------------------------------------------------
// host-side context
// ------------------------------------------------------------------------------------------------------------------
struct T1kDevBuf {
  void *p = nullptr;
  size_t bytes = 0;
};

struct t1k_ctx {
  int device = 0;
  std::vector<unsigned long long> hRaw;  // last fetched counter block (control | statistics stripes | arena cursors)
  hipStream_t stream = nullptr;
  hipEvent_t ev[12] = {};
  t1k_params prm;
  std::string err;
  // reference
  T1kRefDev ref{};
  std::vector<uint64_t> hAlleleOff;
  std::vector<uint32_t> hAlleleLen;
  std::vector<T1kDevBuf> refBufs;
  // reads
  T1kReadsDev reads{};
  T1kDevBuf bReadAscii, bReadOffs, bReadBases, bReadN, bReadLen, bReadWeight, bListPtr, bListCount;
  T1kDevBuf bDedupScratch, bDedupBases, bDedupN, bDedupLen, bDedupWeight;  // t1k_reads_dedupe
  bool readsShared = false;      // the read set belongs to another context (t1k_reads_share)
  int batchMaxLen = 0;
  int batchFastMaxLen = 0;       // longest read among those of at most T1K_MAX_READ_LEN bases (== batchMaxLen unless the window holds longer ones):
                                 // decides the mask width of the fast kernels, so that an odd long read does not widen them for everybody
  int covFullLen = 0;            // run length counted in covFull (fixed at the context's first range)
  bool covFullDirty = false;     // covFull holds runs that are not in covDiff yet
  uint32_t rangeCount = 0;       // read-ends of the last t1k_assign_range
  // assignment arenas
  T1kDevBuf bWgHits, bWgGroups, bWgStage, bWgBig, bWgCache, bLists;
  // Overlap store: the final overlap lists stay resident from the upload of a read set until the next one (mate pairing reads the
  // lists of both mates, which identical-read-end collapse scatters over different batches).  Two slots (consecutive windows of a
  // job overlap in time), each a list of chunks of storeChunkEntries records; a range is written contiguously into one chunk.
  std::vector<T1kDevBuf> storeChunks[2];
  int storeSlot = 0;
  size_t storeChunk[2] = {0, 0};   // chunk being filled
  uint64_t storeUsed[2] = {0, 0};  // records used in it
  T1kOvl *ovlBase = nullptr;       // where the last t1k_assign_range wrote its lists (32-byte working records: t1k_overlaps_download)
  T1kDevBuf bOvlWork;              // the working records of one range (the store keeps the packed form)
  T1kOvlP *storeBase = nullptr;    // where the running range's packed lists go
  // working capacities of the batch arenas (t1k_assign_range grows them on demand up to the limits in prm) and the demand the last
  // overflow reported
  uint64_t wGroup = 0, wList = 0, wRare = 0, wCand = 0, wOvl = 0, wJob = 0, wGenJob = 0, wGenHit = 0;
  uint64_t needGroup = 0, needList = 0, needRare = 0, needCand = 0, needOvl = 0, needJob = 0, needGenJob = 0, needGenHit = 0;
  unsigned long long lastCapFlags = 0;
  bool scaledOnce = false;
  bool covCommitted = false;     // the running range has started adding to the coverage arrays (no retry after that)
  uint32_t queueBoost = 1;       // capacity factor of the alignment queues' stripes (t1k_fullalign_phase): raised when a stripe overflowed
  int covMode = 0;               // t1k_ctx_set_coverage_mode: 0 = t1k_assign_range adds per-base coverage (eager), 1 = it does not (deferred to
                                 // t1k_coverage_selected over the kept lists, or not wanted at all)
  double msAlloc = 0;            // wall time spent in hipMalloc (fresh VRAM is zeroed by the driver: ~35 ms per GB)
  uint64_t bytesAlloc = 0;
  T1kDevBuf bCand, bExt, bCandStart, bCandCount, bOvlStart, bOvlCount, bCounters, bSlowQueue, bSlowScratch, bSortScratch, bEqTrace, bSortTmp, bSlowKeys, bJobSort;
  uint64_t nCand = 0, nOvl = 0;
  // pairing
  T1kDevBuf bEnd1, bEnd2, bHasN, bRows, bRowStart, bRowCount, bFragAssigned, bPairScratch, bPairOverflow, bPairBig, bExtractHuge;
  uint32_t nFragments = 0;
  uint64_t nRows = 0;
  // EM
  T1kDevBuf bEmRowPtr, bEmEc, bEmCount, bEmLen, bEmX0, bEmN, bEmPsum, bEmColPtr, bEmRowOf;
  std::vector<uint64_t> emPieceBytes, emPieceDispl;  // sharded E-step: the ranks' pieces of the per-group sums (psum)
  uint32_t emGroups = 0, emEc = 0, emRowBegin = 0, emRowEnd = 0;
  struct t1k_comm *emComm = nullptr;
  bool emReduceMode = false;   // T1K_EM_COLLECTIVE=allreduce: partial class totals per rank, all-reduce of E doubles (opt-in: re-associates the sums)
  uint64_t emNnz = 0;
  std::vector<int32_t> hEmLen;
  int traceFetch = 0;          // T1K_DEBUG_TRACE
  // an upload in pieces (t1k_reads_upload_begin .. _end)
  uint32_t upN = 0; int upS = 0, upMaxLen = 0; uint64_t upBytes = 0; bool upOpen = false;
  hipEvent_t upEv[4] = {nullptr, nullptr, nullptr, nullptr}; bool upEvSet[4] = {false, false, false, false};
  uint64_t lastSlowGroups = 0;             // groups the last range left to the gap walk (T1K_DEBUG_PHASES)
  uint64_t estTotal[T1K_NARENA] = {};      // entries per arena the last range of this context ended with, per read-end x 2^16: the next range's launch grids (t1k_run_chain)
  bool estValid = false;
  uint64_t pairEpoch = 0;      // epochs handed out to k_pair's allele tables since they were last cleared
  uint64_t pairBigCap = 0;     // entries of k_pair's big arena (scratch of the fragments whose lists exceed a workgroup's own)
  unsigned long long *countersPinned = nullptr;  // page-locked landing buffer of t1k_fetch_counters
  double *emPinned = nullptr;  // page-locked staging for the per-update vectors: [x | n], emPinnedN doubles each
  size_t emPinnedN = 0;
  double emMs[4] = {0, 0, 0, 0};  // T1K_DEBUG_PHASES: per job, time of the updates' staging + enqueue | wait for the device | M-step; [3] = updates
  t1k_allreduce_fn emAllreduce = nullptr;
  void *emUser = nullptr;
  // align batch scratch
  T1kDevBuf bAlign[12];
  // candidate extraction (t1k_extract.hip): bAlign-independent scratch [good flags | error word | statistics]
  T1kDevBuf bExtract;
  t1k_stats stats{};
};

// device memory through the library's process-wide pool (t1k_capi.hip)
hipError_t t1k_dev_malloc(void **out, size_t bytes);
hipError_t t1k_dev_free(void *p);
uint64_t t1k_pool_cached_bytes(int device);  // free blocks the pool keeps for `device`
int t1k_fail(t1k_ctx *ctx, int code, const std::string &msg);
int t1k_ensure(t1k_ctx *ctx, T1kDevBuf &b, size_t bytes);
#define T1K_HIP(ctx, call)                                                                              \
  do {                                                                                                  \
    hipError_t e_ = (call);                                                                             \
    if (e_ != hipSuccess) return t1k_fail(ctx, T1K_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)
