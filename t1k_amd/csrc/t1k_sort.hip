// t1k_amd/csrc/t1k_sort.hip -- device radix sort of (64-bit key, 32-bit value) pairs.  Used to ORDER work queues (so that
// identical alignment jobs are neighbours); it is not part of the genotyper's arithmetic.  rocPRIM's radix sort via hipCUB.
#include <hipcub/hipcub.hpp>
#include "t1k_dev.h"
#include "t1k_launch.h"

int t1k_sort_pairs(t1k_ctx *ctx, const unsigned long long *keysIn, unsigned long long *keysOut, const uint32_t *valsIn, uint32_t *valsOut, uint32_t n, int endBit) {
  if (!n) return T1K_OK;
  size_t bytes = 0;
  T1K_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, keysIn, keysOut, valsIn, valsOut, n, 0, endBit, ctx->stream));
  int rc = t1k_ensure(ctx, ctx->bSortTmp, bytes + 256);
  if (rc) return rc;
  T1K_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(ctx->bSortTmp.p, bytes, keysIn, keysOut, valsIn, valsOut, n, 0, endBit, ctx->stream));
  return T1K_OK;
}

int t1k_inclusive_sum(t1k_ctx *ctx, const uint32_t *in, uint32_t *out, uint32_t n) {
  if (!n) return T1K_OK;
  size_t bytes = 0;
  T1K_HIP(ctx, hipcub::DeviceScan::InclusiveSum(nullptr, bytes, in, out, n, ctx->stream));
  int rc = t1k_ensure(ctx, ctx->bSortTmp, bytes + 256);
  if (rc) return rc;
  T1K_HIP(ctx, hipcub::DeviceScan::InclusiveSum(ctx->bSortTmp.p, bytes, in, out, n, ctx->stream));
  return T1K_OK;
}
