// tools/scratch/dp16_check.cpp -- CPU experiment for DESIGN 9 item 2 (not part of the product): the equal-span banded sweep of
// t1k_ga_equal_traced (csrc/t1k_dev.h: band +-5, 11 cells per row, 5 decision bits per cell) evaluated once with the kernel's int32
// scores and fence -(L+1)^2*4 and once with int16 scores and a fixed fence of -16000, on random read / allele windows with
// substitutions, N's and shifted copies.  Question: are the decision words identical, and do the int16 scores stay in range?
//   g++ -O2 -o /tmp/dp16 tools/scratch/dp16_check.cpp && /tmp/dp16 [cases] [seed]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

static inline bool eqBase(char t, char p) { return t == 'N' || p == 'N' || t == p; }  // t1k_eq: an N compares equal to anything

template <class S>
static void sweep(const std::string &T, const std::string &P, S negInf, std::vector<uint64_t> &words, long &lo, long &hi) {
  const int L = (int)P.size();
  S m[13], e[13];
  for (int s = 0; s < 13; ++s) {
    const int j = s - 6;
    if (j == 0) { m[s] = 0; e[s] = 0; }
    else { m[s] = (S)(-4 - 4 * j); e[s] = (S)(-4 - 4 * (L + 1)); }
  }
  words.assign(L + 1, 0);
  auto note = [&](long v) { if (v > negInf / 2) { if (v < lo) lo = v; if (v > hi) hi = v; } };
  for (int i = 1; i <= L; ++i) {
    S fLeft = negInf, mLeft = negInf;
    uint64_t word = 0;
    for (int s = 0; s < 13; ++s) {
      const int j = i + s - 6;
      S nm, ne, nf;
      if (j < 0) nm = ne = nf = negInf;
      else if (j == 0) { nm = (S)(-4 - 4 * i); ne = (S)(-4 - i); nf = (S)(-4 - 4 * i); }
      else if (s == 0 || s == 12 || j > L) nm = ne = nf = negInf;
      else {
        const S eu = (S)(e[s + 1] - 1), mu = (S)(m[s + 1] - 5);
        ne = eu > mu ? eu : mu;
        const S fl = (S)(fLeft - 1), ml = (S)(mLeft - 5);
        nf = fl > ml ? fl : ml;
        const bool eq = eqBase(T[j - 1], P[i - 1]);
        const S dg = (S)(m[s] + (eq ? 2 : -2));
        nm = dg;
        if (ne > nm) nm = ne;
        if (nf > nm) nm = nf;
        const uint64_t bits = (dg == nm ? 1u : 0u) | (nf >= ne ? 2u : 0u) | (mu == ne ? 4u : 0u) | (ml == nf ? 8u : 0u) | (eq ? 16u : 0u);
        word |= bits << (5 * (s - 1));
        note(nm); note(ne); note(nf);
      }
      m[s] = nm; e[s] = ne;
      fLeft = nf; mLeft = nm;
    }
    words[i] = word;
  }
}

int main(int argc, char **argv) {
  const long cases = argc > 1 ? atol(argv[1]) : 200000;
  std::mt19937_64 rng(argc > 2 ? atoll(argv[2]) : 1);
  auto rnd = [&](int n) { return (int)(rng() % (uint64_t)n); };
  long differ = 0, lo = 0, hi = 0, lo32 = 0, hi32 = 0;
  for (long c = 0; c < cases; ++c) {
    const int L = 2 + rnd(319);  // 2 .. 320
    std::string T(L, 'A'), P;
    for (auto &ch : T) ch = "ACGT"[rnd(4)];
    P = T;
    const int kind = rnd(5);
    if (kind == 1 || kind == 3) {  // shifted copy: the band has to bend
      const int sh = 1 + rnd(L > 5 ? 5 : L - 1);
      if (rnd(2)) P = T.substr(sh) + std::string(sh, 'C'); else P = std::string(sh, 'G') + T.substr(0, L - sh);
    }
    const int nSub = kind == 4 ? L / 2 : rnd(1 + L / 8);
    for (int k = 0; k < nSub; ++k) P[rnd(L)] = "ACGT"[rnd(4)];
    if (rnd(4) == 0) for (int k = 0; k < 1 + rnd(4); ++k) P[rnd(L)] = 'N';
    if (rnd(6) == 0) for (int k = 0; k < 1 + rnd(4); ++k) T[rnd(L)] = 'N';
    std::vector<uint64_t> w32, w16;
    sweep<int32_t>(T, P, (int32_t)((L + 1) * (L + 1) * -4), w32, lo32, hi32);
    sweep<int16_t>(T, P, (int16_t)-16000, w16, lo, hi);
    if (w32 != w16) { if (++differ <= 3) fprintf(stderr, "case %ld (L %d, kind %d): decision words differ\n", c, L, kind); }
  }
  printf("%ld cases, %ld with different decision words; int16 scores off the fence within [%ld, %ld] (int32: [%ld, %ld])\n", cases, differ, lo, hi, lo32, hi32);
  return differ != 0;
}
