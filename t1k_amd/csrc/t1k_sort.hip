// t1k_amd/csrc/t1k_sort.hip -- device-wide ordering primitives: stable radix sort of (64-bit key, 32-bit value) pairs and
// prefix sums over 32-/64-bit counters.  They ORDER and NUMBER work items (alignment queues, read-end hashes, fragment pattern
// hashes, group offsets); they are not part of the genotyper's arithmetic.  rocPRIM (the ROCm-native device primitives) directly.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>
#include "t1k_dev.h"
#include "t1k_launch.h"

int t1k_sort_pairs(t1k_ctx *ctx, const unsigned long long *keysIn, unsigned long long *keysOut, const uint32_t *valsIn, uint32_t *valsOut, uint32_t n, int endBit) {
  if (!n) return T1K_OK;
  (void)hipGetLastError();  // rocPRIM reports the thread's last HIP error after its launches: a stale one (left by another library, e.g. RCCL's device probing) is not ours
  size_t bytes = 0;
  T1K_HIP(ctx, rocprim::radix_sort_pairs(nullptr, bytes, keysIn, keysOut, valsIn, valsOut, (size_t)n, 0u, (unsigned)endBit, ctx->stream));
  int rc = t1k_ensure(ctx, ctx->bSortTmp, bytes + 256);
  if (rc) return rc;
  T1K_HIP(ctx, rocprim::radix_sort_pairs(ctx->bSortTmp.p, bytes, keysIn, keysOut, valsIn, valsOut, (size_t)n, 0u, (unsigned)endBit, ctx->stream));
  return T1K_OK;
}

int t1k_inclusive_sum(t1k_ctx *ctx, const uint32_t *in, uint32_t *out, uint32_t n) {
  if (!n) return T1K_OK;
  (void)hipGetLastError();  // rocPRIM reports the thread's last HIP error after its launches: a stale one (left by another library, e.g. RCCL's device probing) is not ours
  size_t bytes = 0;
  T1K_HIP(ctx, rocprim::inclusive_scan(nullptr, bytes, in, out, (size_t)n, rocprim::plus<uint32_t>(), ctx->stream));
  int rc = t1k_ensure(ctx, ctx->bSortTmp, bytes + 256);
  if (rc) return rc;
  T1K_HIP(ctx, rocprim::inclusive_scan(ctx->bSortTmp.p, bytes, in, out, (size_t)n, rocprim::plus<uint32_t>(), ctx->stream));
  return T1K_OK;
}

int t1k_exclusive_sum64(t1k_ctx *ctx, const uint32_t *in, unsigned long long *out, uint32_t n) {
  if (!n) return T1K_OK;
  (void)hipGetLastError();  // rocPRIM reports the thread's last HIP error after its launches: a stale one (left by another library, e.g. RCCL's device probing) is not ours
  size_t bytes = 0;
  T1K_HIP(ctx, rocprim::exclusive_scan(nullptr, bytes, in, out, 0ull, (size_t)n, rocprim::plus<unsigned long long>(), ctx->stream));
  int rc = t1k_ensure(ctx, ctx->bSortTmp, bytes + 256);
  if (rc) return rc;
  T1K_HIP(ctx, rocprim::exclusive_scan(ctx->bSortTmp.p, bytes, in, out, 0ull, (size_t)n, rocprim::plus<unsigned long long>(), ctx->stream));
  return T1K_OK;
}

int t1k_inclusive_sum_n(t1k_ctx *ctx, const uint32_t *in, uint32_t *out, uint64_t n) {
  if (!n) return T1K_OK;
  (void)hipGetLastError();  // rocPRIM reports the thread's last HIP error after its launches: a stale one (left by another library, e.g. RCCL's device probing) is not ours
  size_t bytes = 0;
  T1K_HIP(ctx, rocprim::inclusive_scan(nullptr, bytes, in, out, (size_t)n, rocprim::plus<uint32_t>(), ctx->stream));
  int rc = t1k_ensure(ctx, ctx->bSortTmp, bytes + 256);
  if (rc) return rc;
  T1K_HIP(ctx, rocprim::inclusive_scan(ctx->bSortTmp.p, bytes, in, out, (size_t)n, rocprim::plus<uint32_t>(), ctx->stream));
  return T1K_OK;
}

// in == out is allowed (every tile is read before it is written)
int t1k_exclusive_sum32(t1k_ctx *ctx, const uint32_t *in, uint32_t *out, uint64_t n) {
  if (!n) return T1K_OK;
  (void)hipGetLastError();  // rocPRIM reports the thread's last HIP error after its launches: a stale one (left by another library, e.g. RCCL's device probing) is not ours
  size_t bytes = 0;
  T1K_HIP(ctx, rocprim::exclusive_scan(nullptr, bytes, in, out, 0u, (size_t)n, rocprim::plus<uint32_t>(), ctx->stream));
  int rc = t1k_ensure(ctx, ctx->bSortTmp, bytes + 256);
  if (rc) return rc;
  T1K_HIP(ctx, rocprim::exclusive_scan(ctx->bSortTmp.p, bytes, in, out, 0u, (size_t)n, rocprim::plus<uint32_t>(), ctx->stream));
  return T1K_OK;
}

int t1k_exclusive_sum_u64(t1k_ctx *ctx, const unsigned long long *in, unsigned long long *out, uint64_t n) {
  if (!n) return T1K_OK;
  (void)hipGetLastError();  // rocPRIM reports the thread's last HIP error after its launches: a stale one (left by another library, e.g. RCCL's device probing) is not ours
  size_t bytes = 0;
  T1K_HIP(ctx, rocprim::exclusive_scan(nullptr, bytes, in, out, 0ull, (size_t)n, rocprim::plus<unsigned long long>(), ctx->stream));
  int rc = t1k_ensure(ctx, ctx->bSortTmp, bytes + 256);
  if (rc) return rc;
  T1K_HIP(ctx, rocprim::exclusive_scan(ctx->bSortTmp.p, bytes, in, out, 0ull, (size_t)n, rocprim::plus<unsigned long long>(), ctx->stream));
  return T1K_OK;
}
