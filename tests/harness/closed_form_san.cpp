// tests/harness/closed_form_san.cpp -- the closed-form pass of the chain (groupFastPath<5, DEFER, CLOSED>: what k_chain_fast<5, 0> runs per group record,
// SeqSet.hpp:1232-1556 + 1697-1848 for single-diagonal groups) compiled for the HOST under the address / undefined-behaviour (and, where clang has it,
// memory) sanitizers.  tests/test_closed_form_cpu.py pastes the routine's text -- taken from t1k_amd/csrc/t1k_chain.hip at test time, between the marker
// lines -- at the marker line below: device intrinsics are shimmed, everything else is the product's code.  Round 6: a build of the kernel under a tighter
// register budget (spilled registers) gave wrong results; this harness is the check that the routine itself reads nothing before it is written, indexes no
// local array out of range and shifts by no out-of-range count on random groups (reads of 60 .. 160 bases, 0 .. 8 substitutions, N on either side,
// hits dropped as the look-up rule drops them, every early-prune mode).

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#define __device__
#define __forceinline__ inline
#define T1K_EVEN 0x5555555555555555ull
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __clzll(long long x) { return __builtin_clzll((unsigned long long)x); }
struct __attribute__((aligned(8))) t1k_u64x2 { uint64_t x, y; };
static inline uint64_t t1k_lowmask(int nPos) { return nPos >= 32 ? ~0ull : ((1ull << (2 * nPos)) - 1); }
struct ReadCtx { const uint64_t *rb, *rn; int len; const uint64_t *gb, *gn; int64_t goff; int alleleLen; bool refN; const uint64_t *gT = nullptr; };
struct GapSink { unsigned long long *cache; uint32_t *jobStr; unsigned long long *counters; uint32_t jobTag, jobSegCap; int arena; };
struct CandOut {
  uint32_t *dst; int n; int stride = 3; int cap = 0x7FFFFFFF; bool overflow = false;
  void push(int rs, int re, int ss, int se, int m0, int m) {
    if (n >= cap) { overflow = true; return; }
    dst[stride * n + 0] = (uint32_t)rs | ((uint32_t)re << 12); dst[stride * n + 1] = (uint32_t)ss | ((uint32_t)m0 << 20); dst[stride * n + 2] = (uint32_t)se | ((uint32_t)m << 20); ++n;
  }
};
template <bool DEFER> static int gapMatchesCached(const ReadCtx &, int, int64_t, int, int, int, const GapSink &, unsigned int *, uint32_t *) { abort(); }
@@ROUTINE@@

static void put(std::vector<uint64_t> &w, int64_t pos, int code) { w[pos >> 5] |= (uint64_t)code << ((pos & 31) * 2); }
int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200000;
  std::mt19937_64 rng(argc > 2 ? atoll(argv[2]) : 7);
  const int k = 11;
  unsigned long long kinds[8] = {0};
  for (int it = 0; it < iters; ++it) {
    const int len = 60 + (int)(rng() % 101);            // 60 .. 160
    const int aLen = len + 40 + (int)(rng() % 400);
    const int apos = (int)(rng() % (aLen - len + 1));   // where the read sits on the allele
    const int64_t goff = 32 * (int64_t)(rng() % 50);
    const int64_t total = goff + aLen + 64 + 320;
    std::vector<uint64_t> gb(total / 32 + 8, 0), gn(total / 32 + 8, 0), rb(16, 0), rn(16, 0);
    std::vector<int> al(aLen), rd(len);
    for (int i = 0; i < aLen; ++i) { al[i] = (int)(rng() & 3); put(gb, goff + i, al[i]); }
    for (int64_t i = goff + aLen; i < total; ++i) put(gb, i, (int)(rng() & 3));    // the next allele's bases behind it
    const int nSub = (int)(rng() % 9);
    for (int i = 0; i < len; ++i) rd[i] = al[apos + i];
    for (int s = 0; s < nSub; ++s) rd[rng() % len] = (int)(rng() & 3);
    const bool hasN = (rng() % 8) == 0;
    std::vector<char> rN(len, 0), gN(aLen, 0);
    if (hasN) { gN[apos + rng() % len] = 1; if (rng() & 1) rN[rng() % len] = 1; }
    for (int i = 0; i < len; ++i) { put(rb, i, rd[i]); if (rN[i]) rn[i >> 5] |= 1ull << ((i & 31) * 2); }
    for (int i = 0; i < aLen; ++i) if (gN[i]) gn[(goff + i) >> 5] |= 1ull << (((goff + i) & 31) * 2);
    // hits: read offsets whose k-mer equals the allele's on this diagonal (no N inside), some of them dropped as the look-up rule does
    uint32_t Mw[5] = {0, 0, 0, 0, 0};
    for (int a = 0; a + k <= len; ++a) {
      bool eq = true;
      for (int j = 0; j < k && eq; ++j) eq = rd[a + j] == al[apos + a + j] && !rN[a + j] && !gN[apos + a + j];
      if (eq && (rng() % 4) != 0) Mw[a >> 5] |= 1u << (a & 31);
    }
    const int diag = -apos;   // read offset - allele offset (the record's convention: allele position = read offset - diag)
    ReadCtx c{rb.data(), rn.data(), len, gb.data(), gn.data(), goff, aLen, true};
    uint32_t cbuf[3]; CandOut out{cbuf, 0};
    const GapSink sink{nullptr, nullptr, nullptr, 0u, 0u, 0};
    uint32_t refs[2] = {0, 0}; int nRefs = 0;
    const int earlyPrune = (int)(rng() % 3);
    const int kind = groupFastPath<5, true, true>(Mw, diag, c, hasN, k, 27, 0.8 + 0.19 * (double)(rng() % 100) / 100.0, out, nullptr, 0, sink, refs, &nRefs, earlyPrune);
    ++kinds[kind & 7];
  }
  printf("ok kinds: finished %llu, gap walk %llu\n", kinds[1], kinds[5]);
  return 0;
}
