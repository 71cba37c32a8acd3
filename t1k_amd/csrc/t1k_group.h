// t1k_amd/csrc/t1k_group.h -- the part of SeqSet::GetOverlapsFromHits (SeqSet.hpp:1232-1556) shared by the genotyper's multi-diagonal
// groups (t1k_chain.hip) and the candidate-read test of the extractor (t1k_extract.hip): hit order, LIS over one diagonal run, hit lengths.
// A hit is one packed word: read offset (low 12 bits) | allele offset << 12.
#pragma once
#include "t1k_dev.h"

__device__ __forceinline__ bool hitKeyLess(uint32_t x, uint32_t y) {  // (diag, alleleOff, readOff): CompSortHitCoordDiff (266-274)
  int cx = (int)(x & 0xFFF) - (int)(x >> 12), cy = (int)(y & 0xFFF) - (int)(y >> 12);
  if (cx != cy) return cx < cy;
  return x < y;
}

// work-array accessor: element i of a lane's private array lives at p[i * 64] (LDS arrays interleaved over the lanes of a wavefront)
struct LaneArr {
  uint32_t *p;  // element i of this lane's array lives at p[i * 64]
  __device__ __forceinline__ uint32_t &operator[](int i) const { return p[i * 64]; }
};

// B[0..m) = one diagonal run's hits nearest to the dominant diagonal, sorted by (allele offset, read offset).  LIS over the read offsets
// (SeqSet.hpp:352-436), chain -> A[s .. s + ret) with repeated allele offsets dropped, then the total hit length on the read and on the
// allele (1032-1069).  Returns false if the run fails the length tests of 1468, 1512-1522.  C[m]: scratch (top | link << 16).
template <class Arr>
// chainReady: A[s .. s + m) already IS the chain (hits in ascending allele offset whose read offsets ascend strictly as well and whose
// allele offsets are distinct: the LIS of such a run is the run itself and nothing repeats) -- only the length tests are left.
__device__ inline bool t1k_run_lis(Arr A, Arr B, Arr C, int s, int m, int k, int hitLenRequired, int *retOut, int *lenROut, int *lenSOut, bool chainReady = false) {
    // LIS over read offsets (352-436); C[i] = top | link << 16, link 0xFFFF = none
    int ret = 1;
    if (chainReady) ret = m;
    else {
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
    }
    if (ret * k < hitLenRequired) return false;
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
    if (lenR < hitLenRequired || lenS < hitLenRequired) return false;
    *retOut = ret; *lenROut = lenR; *lenSOut = lenS;
    return true;
}
