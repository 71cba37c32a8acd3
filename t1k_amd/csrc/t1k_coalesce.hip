// t1k_amd/csrc/t1k_coalesce.hip -- Genotyper::CoalesceReadAssignments (Genotyper.hpp:841-908) on the GPU, over all fragment
// rows of a job at once.
//
// The reference walks the fragments in file order: a fragment whose (allele-sorted) row pattern was seen before is folded into
// that read group -- per entry start = min, the `end` rule of 893-894 (sic: `if (new.end < end) end = new.start`), float weights
// added in fragment order -- otherwise it opens the next group (group ids = first-appearance order, H10).
// Here every fragment's row stays in HBM (t1k_pair_into: ordered by allele, with a 128-bit pattern hash), and one pass at the
// end of the job does the same thing data-parallel:
//   1. the fragments with a non-empty row, in fragment order (scan + compaction);
//   2. two stable 64-bit radix sorts (hash word 2, then hash word 1): equal patterns become neighbours, still in fragment order;
//   3. a run of equal hashes is split wherever two neighbours differ in length or in any allele (a hash collision costs
//      nothing but that comparison);
//   4. runs are numbered by their first fragment (radix sort of the run heads) = the reference's group ids;
//   5. one wavefront per (group, 64 slots): lane q folds slot q of the group's fragments in fragment order -- the same sequence
//      of float additions and of the order-dependent `end` updates as the reference's loop, so the result is bit-identical and
//      does not depend on how the fragments were batched, on the number of pipelines or (with the exchange of t1k_comm.hip) of GPUs.
// Integer / float-add work bound by HBM latency; no MFMA.
#include <algorithm>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "t1k_dev.h"
#include "t1k_launch.h"

struct T1kGroupEnt { int32_t allele, start, end; float weight, adjustWeight; };  // == t1k_group_entry == host GroupEntry, 20 bytes

__global__ void k_co_flag(const uint32_t *rowCount, uint32_t *flag, uint32_t n) {
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f < n) flag[f] = rowCount[f] ? 1u : 0u;
}
__global__ void k_co_compact(const uint32_t *flag, const uint32_t *pos, const unsigned long long *h2, uint32_t *idx, unsigned long long *key, uint32_t n) {
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f < n && flag[f]) { idx[pos[f] - 1] = f; key[pos[f] - 1] = h2[f]; }
}
__global__ void k_co_gather_key(const uint32_t *idx, const unsigned long long *h, unsigned long long *key, uint32_t m) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) key[j] = h[idx[j]];
}
__global__ void k_co_mark(const uint32_t *idx, const unsigned long long *k1, const unsigned long long *h2, const unsigned long long *rowPtr, const uint32_t *rowCount,
                          uint32_t *flag, uint32_t m) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  uint32_t fl = 1;
  if (j > 0 && k1[j] == k1[j - 1]) {
    const uint32_t a = idx[j], b = idx[j - 1];
    if (h2[a] == h2[b] && rowCount[a] == rowCount[b]) {
      const t1k_row_entry *ra = (const t1k_row_entry *)rowPtr[a], *rb = (const t1k_row_entry *)rowPtr[b];
      const uint32_t n = rowCount[a];
      bool same = true;
      for (uint32_t q = 0; q < n && same; ++q) same = ra[q].allele_idx == rb[q].allele_idx;
      if (same) fl = 0;
    }
  }
  flag[j] = fl;
}
__global__ void k_co_heads(const uint32_t *idx, const uint32_t *flag, const uint32_t *runOf, uint32_t *runStart, unsigned long long *headKey, uint32_t *headVal, uint32_t m) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  if (flag[j]) { const uint32_t r = runOf[j] - 1; runStart[r] = j; headKey[r] = idx[j]; headVal[r] = r; }
  if (j == m - 1) runStart[runOf[j]] = m;
}
// group g (first-appearance order) is run order[g]; its size, its 64-slot tiles
__global__ void k_co_sizes(const uint32_t *order, const uint32_t *runStart, const uint32_t *idx, const uint32_t *rowCount, const uint32_t *gid, uint32_t *gSize,
                           uint32_t *gTiles, uint32_t *gFirst, uint32_t G) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const uint32_t f = idx[runStart[order[g]]];
  const uint32_t n = rowCount[f];
  gSize[g] = n; gTiles[g] = (n + 63) / 64; gFirst[g] = gid ? gid[f] : f;
}
__global__ void k_co_tilemap(const unsigned long long *tilePtr, const uint32_t *gTiles, uint32_t *tileGroup, uint32_t G) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const unsigned long long b = tilePtr[g];
  for (uint32_t t = 0; t < gTiles[g]; ++t) tileGroup[b + t] = g;
}

__device__ __forceinline__ void foldEntry(T1kGroupEnt &g, const t1k_row_entry &e) {  // Genotyper.hpp:887-897 (qual == 1 always)
  if (e.start < g.start) g.start = e.start;
  if (e.end < g.end) g.end = e.start;  // sic
  g.weight += e.weight;
  g.adjustWeight += e.adjust_weight;
}

// address of every sorted fragment's row (one dependent load less in the sequential fold below)
__global__ void k_co_ptrs(const uint32_t *idx, const unsigned long long *rowPtr, unsigned long long *ptrSorted, uint32_t m) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) ptrSorted[j] = rowPtr[idx[j]];
}

// One wavefront per (group, 64 slots): lane q folds slot q of the group's fragments in fragment order.  A group's chain is sequential
// (float sums and the `end` rule), and the largest groups hold 10^6 fragments, so the loop is software-pipelined: while batch b is
// folded, the rows of batch b + 1 are in flight and the row addresses of batch b + 2 are being fetched (they are uniform over the
// wavefront and consecutive in memory).
#define CO_B 16
static_assert(sizeof(t1k_row_entry) == 24 && offsetof(t1k_row_entry, start) == 4 && offsetof(t1k_row_entry, weight) == 12 && offsetof(t1k_row_entry, adjust_weight) == 20, "k_co_reduce_long reads the row entries as words");
#define CO_LONG_RUN 4096   // groups of at least this many fragments are folded by k_co_reduce_long (four wavefronts a tile)
__global__ __launch_bounds__(256) void k_co_reduce(const uint32_t *tileGroup, const unsigned long long *tilePtr, const uint32_t *order, const uint32_t *runStart,
                                                   const unsigned long long *ptrSorted, const uint32_t *gSize, const unsigned long long *groupPtr,
                                                   T1kGroupEnt *out, uint64_t nTiles, uint32_t longRun) {
  const uint64_t tile = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= nTiles) return;
  const int lane = threadIdx.x & 63;
  const uint32_t g = tileGroup[tile];
  const uint32_t q = (uint32_t)(tile - tilePtr[g]) * 64 + lane;
  const uint32_t n = gSize[g];
  const uint32_t run = order[g];
  const uint32_t j0 = runStart[run], j1 = runStart[run + 1];
  if (j1 - j0 >= longRun) return;  // k_co_reduce_long's
  if (q >= n) return;
  const t1k_row_entry first = ((const t1k_row_entry *)ptrSorted[j0])[q];
  T1kGroupEnt acc{first.allele_idx, first.start, first.end, first.weight, first.adjust_weight};
  auto fold = [&](const int4 &e) {  // Genotyper.hpp:887-897 (qual == 1 always): x = start, y = end, z = weight, w = adjust weight
    if (e.x < acc.start) acc.start = e.x;
    if (e.y < acc.end) acc.end = e.x;  // sic
    acc.weight += __int_as_float(e.z);
    acc.adjustWeight += __int_as_float(e.w);
  };
  auto rowOf = [&](unsigned long long p) {
    const t1k_row_entry *r = (const t1k_row_entry *)p + q;
    return int4{r->start, r->end, __float_as_int(r->weight), __float_as_int(r->adjust_weight)};
  };
  uint32_t j = j0 + 1;
  if (j + 2 * CO_B <= j1) {
    unsigned long long pNext[CO_B];
    int4 eCur[CO_B];
#pragma unroll
    for (int u = 0; u < CO_B; ++u) eCur[u] = rowOf(ptrSorted[j + u]);
#pragma unroll
    for (int u = 0; u < CO_B; ++u) pNext[u] = ptrSorted[j + CO_B + u];
    j += CO_B;  // eCur holds batch [j - CO_B, j), pNext the addresses of [j, j + CO_B)
    while (j + CO_B <= j1) {
      int4 eNext[CO_B];
#pragma unroll
      for (int u = 0; u < CO_B; ++u) eNext[u] = rowOf(pNext[u]);
      const bool more = j + 2 * CO_B <= j1;
      if (more) {
#pragma unroll
        for (int u = 0; u < CO_B; ++u) pNext[u] = ptrSorted[j + CO_B + u];
      }
#pragma unroll
      for (int u = 0; u < CO_B; ++u) fold(eCur[u]);
#pragma unroll
      for (int u = 0; u < CO_B; ++u) eCur[u] = eNext[u];
      j += CO_B;
      if (!more) break;
    }
#pragma unroll
    for (int u = 0; u < CO_B; ++u) fold(eCur[u]);  // the batch [j - CO_B, j)
  }
  for (; j < j1; ++j) fold(rowOf(ptrSorted[j]));
  out[groupPtr[g] + q] = acc;
}

// The same fold for the groups of many fragments (10^6 in the largest group of a 10 M-pair job: its chain IS the kernel's time, and a
// wavefront that folds 16 rows while the next 16 are in flight spends most of a step waiting for memory: 57 ns a fragment).  Here a tile
// (group, 64 slots) belongs to a workgroup of four wavefronts: wavefront w loads the batches w, w + 4, ... of 16 rows, and the batches
// are folded strictly in order -- the accumulators of the 64 slots pass from wavefront to wavefront through LDS, one turn (and one
// barrier) a batch -- so every addition still happens in fragment order, while a wavefront's next rows have the other three turns to
// arrive: four times the rows in flight per chain.
__global__ __launch_bounds__(256) void k_co_reduce_long(const uint32_t *tileGroup, const unsigned long long *tilePtr, const uint32_t *order, const uint32_t *runStart,
                                                        const unsigned long long *ptrSorted, const uint32_t *gSize, const unsigned long long *groupPtr,
                                                        T1kGroupEnt *out, uint32_t longRun, uint32_t laneStride) {
  __shared__ int4 sAcc[64];
  const uint64_t tile = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t g = tileGroup[tile];
  const uint32_t run = order[g];
  const uint32_t j0 = runStart[run], j1 = runStart[run + 1];
  if (j1 - j0 < longRun) return;  // (uniform over the workgroup) k_co_reduce's
  const uint32_t q = (uint32_t)(tile - tilePtr[g]) * 64 + lane;
  const uint32_t nSlots = gSize[g];
  const bool act = q < nSlots;
  const uint32_t qc = act ? q : nSlots - 1;  // idle lanes of the group's last tile read a valid entry and drop it: no branch around any load
  int32_t allele = 0;
  if (w == 0) {
    const t1k_row_entry first = ((const t1k_row_entry *)ptrSorted[j0])[qc];
    allele = first.allele_idx;
    sAcc[lane] = int4{first.start, first.end, __float_as_int(first.weight), __float_as_int(first.adjust_weight)};
  }
  __syncthreads();
  const uint32_t total = j1 - (j0 + 1);                   // rows to fold
  const uint32_t nBatches = (total + CO_B - 1) / CO_B;
  const uint32_t rounds = (nBatches + 3) / 4;
  unsigned long long pNext[CO_B];
  int4 e[CO_B];
  // (indices past the run are clamped to its last row: every load is unconditional, the fold counts)
  // (The row addresses are the same for every lane, and a uniform load is a SCALAR load: it counts with the LDS operations, out of order,
  // so the wait before every meeting point -- lgkmcnt(0) -- waited for the addresses just requested to come back, a memory latency on
  // every turn.  laneStride is 0: the index only looks lane-dependent to the compiler, and the addresses come through the vector
  // memory path, whose counter the meeting points do not wait for.)
  auto loadPtrs = [&](uint32_t b) {
    typedef const __attribute__((address_space(1))) unsigned long long GPtr;
    GPtr *ps = (GPtr *)ptrSorted + (uint32_t)lane * laneStride;
#pragma unroll
    for (int u = 0; u < CO_B; ++u) pNext[u] = ps[min(j0 + 1 + b * CO_B + (uint32_t)u, j1 - 1)];
  };
  auto loadRows = [&]() {
#pragma unroll
    for (int u = 0; u < CO_B; ++u) {
      // (a GLOBAL load: through a generic pointer it would be a flat load, which also counts as an LDS operation -- and the wait for the
      // LDS write before the meeting point would then wait for the rows as well)
      typedef const __attribute__((address_space(1))) int GInt;
      GInt *r = (GInt *)(pNext[u] + (unsigned long long)qc * sizeof(t1k_row_entry));
      e[u] = int4{r[1], r[2], r[3], r[5]};  // start, end, weight, adjust weight (t1k_row_entry: allele, start, end, weight, qual, adjust_weight)
    }
  };
  // (the accumulators travel through LDS only, so a turn ends with "wait for the LDS write and meet" -- NOT __syncthreads(), whose fence
  // also waits for the row loads just issued (vmcnt(0)) and puts a full memory latency back into every turn: 211 ms instead of 53.  And
  // the loads sit in straight-line code: inside an `if (w == turn)` the compiler copies the loaded registers at the join and waits there)
  auto meet = [] { t1k_lds_barrier(); };
  const int ws = __builtin_amdgcn_readfirstlane(w);
  loadPtrs(min((uint32_t)ws, nBatches - 1));
  loadRows();
  loadPtrs(min((uint32_t)ws + 4, nBatches - 1));
  for (uint32_t r = 0; r < rounds; ++r) {
    const uint32_t b = 4 * r + (uint32_t)ws;
    for (int t = 0; t < ws; ++t) meet();  // the wavefronts before this one fold their batches
    {
      int4 a = sAcc[lane];
      float wt = __int_as_float(a.z), aw = __int_as_float(a.w);
      const int cnt = b < nBatches ? (int)min((uint32_t)CO_B, total - b * CO_B) : 0;  // (past the run: nothing to fold, the turn is still taken)
      // Genotyper.hpp:887-897 (qual == 1 always): x = start, y = end, z = weight, w = adjust weight
      if (cnt == CO_B) {
#pragma unroll
        for (int u = 0; u < CO_B; ++u) {
          if (e[u].x < a.x) a.x = e[u].x;
          if (e[u].y < a.y) a.y = e[u].x;  // sic
          wt += __int_as_float(e[u].z);
          aw += __int_as_float(e[u].w);
        }
      } else {
#pragma unroll
        for (int u = 0; u < CO_B; ++u) {
          if (u < cnt) {
            if (e[u].x < a.x) a.x = e[u].x;
            if (e[u].y < a.y) a.y = e[u].x;
            wt += __int_as_float(e[u].z);
            aw += __int_as_float(e[u].w);
          }
        }
      }
      a.z = __float_as_int(wt); a.w = __float_as_int(aw);
      sAcc[lane] = a;
    }
    // The accumulators are handed over FIRST: the next wavefront folds while this one requests its next batch (clamped past the run:
    // loaded and never folded; its addresses are here, its rows have the other wavefronts' turns to arrive).  Requesting before the
    // hand-over, as round 3 did, put the issue of 32 loads and their addresses on every turn's critical path: it was as long as the fold.
    meet();
    loadRows();
    loadPtrs(min(b + 8, nBatches - 1));
    for (int t = ws + 1; t < 4; ++t) meet();
  }
  if (w == 0 && act) {
    const int4 a = sAcc[lane];
    out[groupPtr[g] + q] = T1kGroupEnt{allele, a.x, a.y, __int_as_float(a.z), __int_as_float(a.w)};
  }
}

// rows of fragments [f0, f0 + n) back in the reference's row order (the `qual` slot holds the position), one wave per fragment
__global__ void k_co_unsort(const unsigned long long *rowPtr, const uint32_t *rowCount, const unsigned long long *outOff, t1k_row_entry *out, uint64_t f0, uint32_t n) {
  const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  const t1k_row_entry *r = (const t1k_row_entry *)rowPtr[f0 + i];
  const uint32_t c = rowCount[f0 + i];
  for (uint32_t q = lane; q < c; q += 64) {
    t1k_row_entry e = r[q];
    const uint32_t at = __float_as_uint(e.qual);
    e.qual = 1.0f;
    out[outOff[i] + at] = e;
  }
}

static int rsFail(t1k_rowset *rs, int code, const std::string &m) { if (rs) rs->err = m; return code; }
#define RS_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return rsFail(rs, T1K_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

int t1k_rowset_chunk(t1k_rowset *rs, t1k_ctx *ctx, size_t *chunk, t1k_row_entry **rows, uint64_t *cap, unsigned long long **cursor) {
  std::lock_guard<std::mutex> g(rs->m);
  if (rs->cur >= 1024) return t1k_fail(ctx, T1K_ERR_CAPACITY, "rowset: more than 1024 row chunks");
  if (rs->cur >= rs->chunks.size()) rs->chunks.resize(rs->cur + 1);
  T1kDevBuf &b = rs->chunks[rs->cur];
  if (!b.p) {
    hipError_t e = t1k_dev_malloc(&b.p, rs->chunkEntries * sizeof(t1k_row_entry));
    if (e != hipSuccess) { b.p = nullptr; return t1k_fail(ctx, T1K_ERR_DEVICE, std::string("rowset: hipMalloc of a row chunk failed: ") + hipGetErrorString(e)); }
    b.bytes = rs->chunkEntries * sizeof(t1k_row_entry);
  }
  *chunk = rs->cur; *rows = (t1k_row_entry *)b.p; *cap = rs->chunkEntries; *cursor = (unsigned long long *)rs->bCursors.p + rs->cur;
  return T1K_OK;
}
int t1k_rowset_chunk_full(t1k_rowset *rs, t1k_ctx *, size_t chunk) {
  std::lock_guard<std::mutex> g(rs->m);
  if (rs->cur == chunk) ++rs->cur;
  return T1K_OK;
}

extern "C" {

int t1k_rowset_create(t1k_ctx *owner, uint64_t nFragments, const uint8_t *whitelist, t1k_rowset **out) {
  if (!owner || !out) return T1K_ERR_ARG;
  *out = nullptr;
  if (nFragments >= 0xFFFFFFF0ull) return t1k_fail(owner, T1K_ERR_ARG, "t1k_rowset_create: more than 2^32 fragments on one GPU");
  T1K_HIP(owner, hipSetDevice(owner->device));
  t1k_rowset *rs = new t1k_rowset();
  rs->device = owner->device; rs->owner = owner; rs->nFrag = nFragments;
  const uint64_t F = std::max<uint64_t>(nFragments, 1);
  int rc;
  if ((rc = t1k_ensure(owner, rs->bFrag, F * (8 * 3 + 4 + 1) + 256)) || (rc = t1k_ensure(owner, rs->bCursors, 1024 * 8))) { delete rs; return rc; }
  rs->rowPtr = (unsigned long long *)rs->bFrag.p; rs->h1 = rs->rowPtr + F; rs->h2 = rs->h1 + F;
  rs->rowCount = (uint32_t *)(rs->h2 + F); rs->assigned = (uint8_t *)(rs->rowCount + F);
  T1K_HIP(owner, hipMemsetAsync(rs->bFrag.p, 0, rs->bFrag.bytes, owner->stream));  // a fragment that is never paired has an empty row
  T1K_HIP(owner, hipMemsetAsync(rs->bCursors.p, 0, 1024 * 8, owner->stream));
  const char *ce = getenv("T1K_ROW_CHUNK");
  rs->chunkEntries = ce ? std::max<uint64_t>(1u << 16, strtoull(ce, nullptr, 10)) : std::max<uint64_t>((uint64_t)owner->prm.row_cap, 64ull << 20);
  if (whitelist) {
    const uint32_t A = owner->ref.nAlleles;
    if ((rc = t1k_ensure(owner, rs->bWhitelist, A))) { delete rs; return rc; }
    T1K_HIP(owner, hipMemcpyAsync(rs->bWhitelist.p, whitelist, A, hipMemcpyHostToDevice, owner->stream));
    rs->whitelist = (const uint8_t *)rs->bWhitelist.p;
  }
  T1K_HIP(owner, hipStreamSynchronize(owner->stream));
  *out = rs;
  return T1K_OK;
}

// a rowset created for an upper bound of the fragment count (a streamed input): the fragments that exist.  The per-fragment arrays keep
// their places; the stages behind the loop walk the first n entries only.
int t1k_rowset_trim(t1k_rowset *rs, uint64_t nFragments) {
  if (!rs || rs->exchanged || nFragments > rs->nFrag) return T1K_ERR_ARG;
  rs->nFrag = nFragments;
  return T1K_OK;
}

void t1k_rowset_destroy(t1k_rowset *rs) {
  if (!rs) return;
  (void)hipSetDevice(rs->device);
  T1kDevBuf *all[] = {&rs->bFrag, &rs->bCursors, &rs->bWhitelist, &rs->bWork, &rs->bGroupPtr, &rs->bGroupEnt, &rs->bGroupFirst, &rs->bSend, &rs->bRecv, &rs->bFrag2, &rs->bAll};
  for (auto *b : all) if (b->p) (void)t1k_dev_free(b->p);
  for (auto &b : rs->chunks) if (b.p) (void)t1k_dev_free(b.p);
  if (rs->copyStream) (void)hipStreamDestroy(rs->copyStream);
  delete rs;
}

int t1k_rowset_set_raw(t1k_rowset *rs, int raw) { if (!rs) return T1K_ERR_ARG; rs->rawKept = raw != 0; return T1K_OK; }
int t1k_rowset_device_bytes(t1k_rowset *rs, uint64_t *bytes, uint64_t *rowEntries) {
  if (!rs || !bytes) return T1K_ERR_ARG;
  size_t nChunks;
  {
    std::lock_guard<std::mutex> g(rs->m);
    uint64_t b = rs->bFrag.bytes;
    for (const T1kDevBuf &c : rs->chunks) if (c.p) b += c.bytes;
    *bytes = b;
    nChunks = std::min<size_t>(rs->chunks.size(), 1024);
  }
  if (rowEntries) {  // the chunks' cursors (they count past a full chunk's capacity: clamped), read on the rowset's own stream
    *rowEntries = 0;
    if (nChunks) {
      RS_HIP(hipSetDevice(rs->device));
      std::vector<unsigned long long> cur(nChunks);
      hipStream_t st = nullptr;  // (a stream of its own: the rowset's copy stream belongs to the thread that writes the read files beside the loop)
      RS_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      hipError_t e = hipMemcpyAsync(cur.data(), rs->bCursors.p, nChunks * 8, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      (void)hipStreamDestroy(st);
      RS_HIP(e);
      for (unsigned long long c : cur) *rowEntries += std::min<unsigned long long>(c, rs->chunkEntries);
    }
  }
  return T1K_OK;
}
const char *t1k_rowset_last_error(const t1k_rowset *rs) { return rs ? rs->err.c_str() : "no rowset"; }

int t1k_rowset_coalesce(t1k_rowset *rs, uint64_t *nGroups, uint64_t *nEntries, uint64_t *assignedFragments) {
  return t1k_rowset_coalesce_sized(rs, nGroups, nEntries, assignedFragments, nullptr, nullptr);
}

int t1k_rowset_coalesce_sized(t1k_rowset *rs, uint64_t *nGroups, uint64_t *nEntries, uint64_t *assignedFragments, void (*sized)(uint64_t, uint64_t, void *), void *user) {
  if (!rs) return T1K_ERR_ARG;
  t1k_ctx *ctx = rs->owner;
  RS_HIP(hipSetDevice(rs->device));
  hipStream_t st = ctx->stream;
  const uint32_t F = (uint32_t)rs->nFrag;
  rs->nGroups = rs->nEntries = 0;
  if (!rs->exchanged) rs->nAssigned = 0;
  rs->coalesced = true;
  if (nGroups) *nGroups = 0;
  if (nEntries) *nEntries = 0;
  if (assignedFragments) *assignedFragments = rs->exchanged ? rs->nAssigned : 0;
  if (F == 0) return T1K_OK;
  int rc;
  auto fail = [&](int code) { rs->err = ctx->err; return code; };
  const bool dbg = getenv("T1K_DEBUG_PHASES") != nullptr;
  auto tLap = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!dbg) return;
    (void)hipStreamSynchronize(st);
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[t1k] coalesce %s: %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tLap).count());
    tLap = now;
  };
  // work area: u32 arrays a0..a5 [F+2], u64 arrays k0, k1 [F+1]
  const size_t n4 = ((size_t)(F + 2) * 4 + 255) & ~(size_t)255, n8 = ((size_t)(F + 2) * 8 + 255) & ~(size_t)255;
  if ((rc = t1k_ensure(ctx, rs->bWork, 8 * n4 + 3 * n8))) return fail(rc);
  char *wp = (char *)rs->bWork.p;
  uint32_t *a0 = (uint32_t *)wp, *a1 = (uint32_t *)(wp + n4), *a2 = (uint32_t *)(wp + 2 * n4), *a3 = (uint32_t *)(wp + 3 * n4), *a4 = (uint32_t *)(wp + 4 * n4),
           *a5 = (uint32_t *)(wp + 5 * n4), *a6 = (uint32_t *)(wp + 6 * n4), *a7 = (uint32_t *)(wp + 7 * n4);
  unsigned long long *k0 = (unsigned long long *)(wp + 8 * n4), *k1 = (unsigned long long *)(wp + 8 * n4 + n8), *k2 = (unsigned long long *)(wp + 8 * n4 + 2 * n8);
  const unsigned nbF = (F + 255) / 256;
  // 1. fragments with a row, in fragment order
  hipLaunchKernelGGL(k_co_flag, dim3(nbF), dim3(256), 0, st, rs->rowCount, a0, F);
  if ((rc = t1k_inclusive_sum(ctx, a0, a1, F))) return fail(rc);
  uint32_t M = 0;
  RS_HIP(hipMemcpyAsync(&M, a1 + (F - 1), 4, hipMemcpyDeviceToHost, st));
  RS_HIP(hipStreamSynchronize(st));
  if (!rs->exchanged) rs->nAssigned = M;  // (after an exchange: the count of this rank's own fragments, taken before it)
  if (assignedFragments) *assignedFragments = rs->nAssigned;
  if (M == 0) return T1K_OK;
  hipLaunchKernelGGL(k_co_compact, dim3(nbF), dim3(256), 0, st, a0, a1, rs->h2, a2, k0, F);  // a2 = fragment ids, k0 = hash word 2
  const unsigned nbM = (M + 255) / 256;
  // 2. stable sorts: by hash word 2, then by hash word 1
  if ((rc = t1k_sort_pairs(ctx, k0, k1, a2, a3, M))) return fail(rc);                       // a3 = ids ordered by h2
  hipLaunchKernelGGL(k_co_gather_key, dim3(nbM), dim3(256), 0, st, a3, rs->h1, k0, M);
  if ((rc = t1k_sort_pairs(ctx, k0, k1, a3, a2, M))) return fail(rc);                       // a2 = ids ordered by (h1, h2, fragment), k1 = sorted h1
  uint32_t *idx = a2;
  lap("compaction + two radix sorts");
  // 3. run heads
  hipLaunchKernelGGL(k_co_mark, dim3(nbM), dim3(256), 0, st, idx, k1, rs->h2, rs->rowPtr, rs->rowCount, a0, M);
  if ((rc = t1k_inclusive_sum(ctx, a0, a1, M))) return fail(rc);                            // a0 = head flags, a1 = 1-based run of each position
  uint32_t G = 0;
  RS_HIP(hipMemcpyAsync(&G, a1 + (M - 1), 4, hipMemcpyDeviceToHost, st));
  RS_HIP(hipStreamSynchronize(st));
  // 4. runs numbered by their first fragment
  uint32_t *runStart = a3, *headVal = a4, *order = a5;
  hipLaunchKernelGGL(k_co_heads, dim3(nbM), dim3(256), 0, st, idx, a0, a1, runStart, k0, headVal, M);
  if ((rc = t1k_sort_pairs(ctx, k0, k2, headVal, order, G, 32))) return fail(rc);
  uint32_t *gSize = a0, *gTiles = a1, *gFirst = a6;
  const unsigned nbG = (G + 255) / 256;
  hipLaunchKernelGGL(k_co_sizes, dim3(nbG), dim3(256), 0, st, order, runStart, idx, rs->rowCount, rs->gid, gSize, gTiles, gFirst, G);
  if ((rc = t1k_ensure(ctx, rs->bGroupPtr, ((size_t)G + 2) * 8 * 2))) return fail(rc);
  unsigned long long *groupPtr = (unsigned long long *)rs->bGroupPtr.p, *tilePtr = groupPtr + (G + 2);
  // exclusive sums over G + 1 elements (the element behind the last group is read but never used: make it defined)
  RS_HIP(hipMemsetAsync(gSize + G, 0, 4, st));
  RS_HIP(hipMemsetAsync(gTiles + G, 0, 4, st));
  if ((rc = t1k_exclusive_sum64(ctx, gSize, groupPtr, G + 1))) return fail(rc);
  if ((rc = t1k_exclusive_sum64(ctx, gTiles, tilePtr, G + 1))) return fail(rc);
  unsigned long long tot[2] = {0, 0};
  RS_HIP(hipMemcpyAsync(&tot[0], groupPtr + G, 8, hipMemcpyDeviceToHost, st));
  RS_HIP(hipMemcpyAsync(&tot[1], tilePtr + G, 8, hipMemcpyDeviceToHost, st));
  RS_HIP(hipStreamSynchronize(st));
  const uint64_t N = tot[0], nTiles = tot[1];
  if (nTiles >= 0xFFFFFFFFull) return rsFail(rs, T1K_ERR_CAPACITY, "t1k_rowset_coalesce: too many group tiles");
  if ((rc = t1k_ensure(ctx, rs->bGroupEnt, N * sizeof(T1kGroupEnt) + 64))) return fail(rc);
  if ((rc = t1k_ensure(ctx, rs->bGroupFirst, ((size_t)G + 1) * 4 + nTiles * 4 + 64))) return fail(rc);
  uint32_t *tileGroup = (uint32_t *)rs->bGroupFirst.p + (G + 1);
  RS_HIP(hipMemcpyAsync(rs->bGroupFirst.p, gFirst, (size_t)G * 4, hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(k_co_tilemap, dim3(nbG), dim3(256), 0, st, tilePtr, gTiles, tileGroup, G);
  lap("run heads, group order, offsets");
  // 5. fold every (group, slot) in fragment order
  unsigned long long *ptrSorted = k1;  // the sorted hash words are no longer needed
  hipLaunchKernelGGL(k_co_ptrs, dim3(nbM), dim3(256), 0, st, idx, rs->rowPtr, ptrSorted, M);
  {
    static const uint32_t longRun = getenv("T1K_CO_LONG_RUN") ? (uint32_t)std::max(2 * CO_B + 2, atoi(getenv("T1K_CO_LONG_RUN"))) : (uint32_t)CO_LONG_RUN;  // (0xFFFFFFFF-like values: every group through k_co_reduce)
    hipLaunchKernelGGL(k_co_reduce_long, dim3((unsigned)nTiles), dim3(256), 0, st, tileGroup, tilePtr, order, runStart, ptrSorted, gSize, groupPtr,
                       (T1kGroupEnt *)rs->bGroupEnt.p, longRun, 0u);
    hipLaunchKernelGGL(k_co_reduce, dim3((unsigned)((nTiles + 3) / 4)), dim3(256), 0, st, tileGroup, tilePtr, order, runStart, ptrSorted, gSize, groupPtr,
                       (T1kGroupEnt *)rs->bGroupEnt.p, nTiles, longRun);
  }
  if (sized) sized(G, N, user);  // the fold is on its way: the caller sizes (and first-touches) its host tables meanwhile
  RS_HIP(hipStreamSynchronize(st));
  lap("k_co_reduce");
  rs->nGroups = G; rs->nEntries = N;
  if (nGroups) *nGroups = G;
  if (nEntries) *nEntries = N;
  return T1K_OK;
}

int t1k_rowset_groups_download(t1k_rowset *rs, uint64_t *groupPtr, t1k_group_entry *entries, uint32_t *firstFragment) {
  if (!rs || !rs->coalesced) return rsFail(rs, T1K_ERR_STATE, "t1k_rowset_groups_download: not coalesced");
  RS_HIP(hipSetDevice(rs->device));
  const uint64_t G = rs->nGroups;
  if (groupPtr) {
    if (G) RS_HIP(hipMemcpy(groupPtr, rs->bGroupPtr.p, (G + 1) * 8, hipMemcpyDeviceToHost));
    else groupPtr[0] = 0;
  }
  if (entries && rs->nEntries) RS_HIP(hipMemcpy(entries, rs->bGroupEnt.p, rs->nEntries * sizeof(T1kGroupEnt), hipMemcpyDeviceToHost));
  if (firstFragment && G) RS_HIP(hipMemcpy(firstFragment, rs->bGroupFirst.p, G * 4, hipMemcpyDeviceToHost));
  return T1K_OK;
}

int t1k_rowset_assigned_download(t1k_rowset *rs, uint8_t *fragAssigned) {
  if (!rs) return T1K_ERR_ARG;
  const uint64_t nLocal = rs->exchanged ? rs->nFragLocal : rs->nFrag;  // the flags belong to this rank's own fragments
  if (nLocal == 0) return T1K_OK;  // (an empty input: the caller's array is empty too)
  if (!fragAssigned) return rsFail(rs, T1K_ERR_ARG, "t1k_rowset_assigned_download: no output array");
  RS_HIP(hipSetDevice(rs->device));
  if (nLocal) RS_HIP(hipMemcpy(fragAssigned, rs->assigned, nLocal, hipMemcpyDeviceToHost));
  return T1K_OK;
}

// flags of fragments [first, first + count) while the pipelines are still at work on later ones: a copy on the rowset's own
// non-blocking stream, which neither waits for nor holds up the pipelines' streams (one caller at a time)
int t1k_rowset_assigned_range(t1k_rowset *rs, uint64_t first, uint64_t count, uint8_t *fragAssigned) {
  if (!rs || !fragAssigned || rs->exchanged || first + count > rs->nFrag) return rsFail(rs, T1K_ERR_ARG, "t1k_rowset_assigned_range: bad arguments");
  RS_HIP(hipSetDevice(rs->device));
  if (!count) return T1K_OK;
  if (!rs->copyStream) RS_HIP(hipStreamCreateWithFlags(&rs->copyStream, hipStreamNonBlocking));
  RS_HIP(hipMemcpyAsync(fragAssigned, rs->assigned + first, count, hipMemcpyDeviceToHost, rs->copyStream));
  RS_HIP(hipStreamSynchronize(rs->copyStream));
  return T1K_OK;
}

int t1k_rowset_rows_download(t1k_rowset *rs, uint64_t first, uint32_t count, uint32_t *rowCounts, t1k_row_entry *rows, uint64_t cap, uint64_t *total) {
  if (!rs || !rowCounts || first + count > rs->nFrag) return rsFail(rs, T1K_ERR_ARG, "t1k_rowset_rows_download: bad arguments");
  RS_HIP(hipSetDevice(rs->device));
  if (total) *total = 0;
  if (!count) return T1K_OK;
  RS_HIP(hipMemcpy(rowCounts, rs->rowCount + first, (size_t)count * 4, hipMemcpyDeviceToHost));
  std::vector<unsigned long long> off(count);
  uint64_t tot = 0;
  for (uint32_t i = 0; i < count; ++i) { off[i] = tot; tot += rowCounts[i]; }
  if (total) *total = tot;
  if (!rows || !tot) return T1K_OK;
  if (cap < tot) return rsFail(rs, T1K_ERR_ARG, "t1k_rowset_rows_download: row buffer too small");
  void *dOff = nullptr, *dOut = nullptr;
  RS_HIP(t1k_dev_malloc(&dOff, (size_t)count * 8));
  if (t1k_dev_malloc(&dOut, tot * sizeof(t1k_row_entry)) != hipSuccess) { (void)t1k_dev_free(dOff); return rsFail(rs, T1K_ERR_DEVICE, "t1k_rowset_rows_download: out of device memory"); }
  hipStream_t st = rs->owner->stream;
  hipError_t e = hipMemcpyAsync(dOff, off.data(), (size_t)count * 8, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_co_unsort, dim3((count + 3) / 4), dim3(256), 0, st, rs->rowPtr, rs->rowCount, (const unsigned long long *)dOff, (t1k_row_entry *)dOut, first, count);
    e = hipMemcpyAsync(rows, dOut, tot * sizeof(t1k_row_entry), hipMemcpyDeviceToHost, st);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)t1k_dev_free(dOff); (void)t1k_dev_free(dOut);
  if (e != hipSuccess) return rsFail(rs, T1K_ERR_DEVICE, hipGetErrorString(e));
  return T1K_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// multi-GPU: fragments travel to the owner of their pattern (hash word 1 mod nRanks)
// ------------------------------------------------------------------------------------------------------------------
struct T1kFragMeta { uint32_t gid, n; unsigned long long h1, h2; };  // 24 bytes

__global__ void k_ex_dest(const uint32_t *idx, const unsigned long long *h1, unsigned long long *key, uint32_t nRanks, uint32_t m) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) key[j] = h1[idx[j]] % nRanks;
}
__global__ void k_ex_counts(const uint32_t *idxS, const uint32_t *rowCount, uint32_t *cnt, uint32_t m) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) cnt[j] = rowCount[idxS[j]];
}
// start[p] = first sorted position whose destination is >= p  (p = 0 .. nRanks)
__global__ void k_ex_bounds(const unsigned long long *keyS, uint32_t m, uint32_t nRanks, uint32_t *start) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > nRanks) return;
  uint32_t lo = 0, hi = m;
  while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (keyS[mid] < p) lo = mid + 1; else hi = mid; }
  start[p] = lo;
}
__global__ void k_ex_pack(const uint32_t *idxS, const unsigned long long *eOff, const unsigned long long *rowPtr, const uint32_t *rowCount, const unsigned long long *h1,
                          const unsigned long long *h2, uint64_t fragBase, T1kFragMeta *meta, t1k_row_entry *rows, uint32_t m) {
  const uint32_t j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= m) return;
  const int lane = threadIdx.x & 63;
  const uint32_t f = idxS[j], n = rowCount[f];
  if (lane == 0) meta[j] = T1kFragMeta{(uint32_t)(fragBase + f), n, h1[f], h2[f]};
  const t1k_row_entry *src = (const t1k_row_entry *)rowPtr[f];
  t1k_row_entry *dst = rows + eOff[j];
  for (uint32_t q = lane; q < n; q += 64) dst[q] = src[q];
}
__global__ void k_ex_unpack_counts(const T1kFragMeta *meta, uint32_t *cnt, uint32_t m) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) cnt[j] = meta[j].n;
}
__global__ void k_ex_unpack(const T1kFragMeta *meta, const unsigned long long *eOff, const t1k_row_entry *rows, unsigned long long *rowPtr, uint32_t *rowCount,
                            unsigned long long *h1, unsigned long long *h2, uint32_t *gid, uint32_t m) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const T1kFragMeta t = meta[j];
  rowPtr[j] = (unsigned long long)(rows + eOff[j]); rowCount[j] = t.n; h1[j] = t.h1; h2[j] = t.h2; gid[j] = t.gid;
}
__global__ void k_ex_sizes(const unsigned long long *groupPtr, uint32_t *size, uint32_t g) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < g) size[i] = (uint32_t)(groupPtr[i + 1] - groupPtr[i]);
}

extern "C" {

// Every fragment row of every rank moves to the rank that owns its pattern.  Afterwards the rowset holds the fragments this rank
// owns, in global fragment order (ranks hold contiguous slices of the fragments in rank order), and t1k_rowset_coalesce folds them.
// fragBase = global index of this rank's first fragment.
int t1k_rowset_exchange(t1k_rowset *rs, t1k_comm *comm, uint64_t fragBase) {
  if (!rs || !comm) return T1K_ERR_ARG;
  if (rs->exchanged) return rsFail(rs, T1K_ERR_STATE, "t1k_rowset_exchange: already exchanged");
  t1k_ctx *ctx = rs->owner;
  RS_HIP(hipSetDevice(rs->device));
  hipStream_t st = ctx->stream;
  const int N = t1k_comm_size(comm), me = t1k_comm_rank(comm);
  const uint32_t F = (uint32_t)rs->nFrag;
  int rc;
  auto fail = [&](int code) { rs->err = ctx->err; return code; };
  auto cfail = [&](int code) { rs->err = t1k_comm_last_error(comm); return code; };
  const size_t n4 = ((size_t)(F + 2) * 4 + 255) & ~(size_t)255, n8 = ((size_t)(F + 2) * 8 + 255) & ~(size_t)255;
  if ((rc = t1k_ensure(ctx, rs->bWork, 8 * n4 + 3 * n8))) return fail(rc);
  char *wp = (char *)rs->bWork.p;
  uint32_t *a0 = (uint32_t *)wp, *a1 = (uint32_t *)(wp + n4), *a2 = (uint32_t *)(wp + 2 * n4), *a3 = (uint32_t *)(wp + 3 * n4), *a4 = (uint32_t *)(wp + 4 * n4);
  unsigned long long *k0 = (unsigned long long *)(wp + 8 * n4), *k1 = (unsigned long long *)(wp + 8 * n4 + n8), *k2 = (unsigned long long *)(wp + 8 * n4 + 2 * n8);
  uint32_t M = 0;
  if (F) {
    const unsigned nbF = (F + 255) / 256;
    hipLaunchKernelGGL(k_co_flag, dim3(nbF), dim3(256), 0, st, rs->rowCount, a0, F);
    if ((rc = t1k_inclusive_sum(ctx, a0, a1, F))) return fail(rc);
    RS_HIP(hipMemcpyAsync(&M, a1 + (F - 1), 4, hipMemcpyDeviceToHost, st));
    RS_HIP(hipStreamSynchronize(st));
    if (M) hipLaunchKernelGGL(k_co_compact, dim3(nbF), dim3(256), 0, st, a0, a1, rs->h2, a2, k0, F);  // a2 = fragments with a row, in order
  }
  rs->nAssigned = M;
  // stable order by destination
  std::vector<uint32_t> start(N + 1, 0);
  std::vector<unsigned long long> eAt(N + 1, 0);
  uint32_t *idxS = a3;
  if (M) {
    const unsigned nbM = (M + 255) / 256;
    hipLaunchKernelGGL(k_ex_dest, dim3(nbM), dim3(256), 0, st, a2, rs->h1, k0, (uint32_t)N, M);
    if ((rc = t1k_sort_pairs(ctx, k0, k1, a2, idxS, M, 16))) return fail(rc);
    hipLaunchKernelGGL(k_ex_counts, dim3(nbM), dim3(256), 0, st, idxS, rs->rowCount, a0, M);
    RS_HIP(hipMemsetAsync(a0 + M, 0, 4, st));
    if ((rc = t1k_exclusive_sum64(ctx, a0, k2, M + 1))) return fail(rc);  // k2 = entry offset of each sorted fragment, k2[M] = total
    hipLaunchKernelGGL(k_ex_bounds, dim3(1), dim3(256), 0, st, k1, M, (uint32_t)N, a4);
    if (N + 1 > 256) return rsFail(rs, T1K_ERR_ARG, "t1k_rowset_exchange: more than 255 ranks");
    RS_HIP(hipMemcpyAsync(start.data(), a4, (size_t)(N + 1) * 4, hipMemcpyDeviceToHost, st));
    RS_HIP(hipStreamSynchronize(st));
    for (int p = 0; p <= N; ++p) RS_HIP(hipMemcpyAsync(&eAt[p], k2 + start[p], 8, hipMemcpyDeviceToHost, st));
    RS_HIP(hipStreamSynchronize(st));
  }
  const uint64_t Etot = eAt[N];
  // what everybody sends to everybody: [rank][2 * dest] fragments, [rank][2 * dest + 1] entries
  std::vector<uint64_t> mine(2 * (size_t)N), all(2 * (size_t)N * N);
  for (int p = 0; p < N; ++p) { mine[2 * p] = start[p + 1] - start[p]; mine[2 * p + 1] = eAt[p + 1] - eAt[p]; }
  if ((rc = t1k_comm_allgather_u64(comm, mine.data(), (uint32_t)(2 * N), all.data()))) return cfail(rc);
  std::vector<uint64_t> sOffM(N + 1), sOffE(N + 1), rOffM(N + 1, 0), rOffE(N + 1, 0);
  for (int p = 0; p <= N; ++p) { sOffM[p] = (uint64_t)start[p] * sizeof(T1kFragMeta); sOffE[p] = eAt[p] * sizeof(t1k_row_entry); }
  uint64_t recvF = 0, recvE = 0, totalF = 0;
  for (int r = 0; r < N; ++r) {
    rOffM[r] = recvF * sizeof(T1kFragMeta); rOffE[r] = recvE * sizeof(t1k_row_entry);
    recvF += all[(size_t)r * 2 * N + 2 * me]; recvE += all[(size_t)r * 2 * N + 2 * me + 1];
    for (int p = 0; p < N; ++p) totalF += all[(size_t)r * 2 * N + 2 * p];
  }
  rOffM[N] = recvF * sizeof(T1kFragMeta); rOffE[N] = recvE * sizeof(t1k_row_entry);
  if (recvF >= 0xFFFFFFF0ull || totalF >= 0xFFFFFFF0ull) return rsFail(rs, T1K_ERR_CAPACITY, "t1k_rowset_exchange: more than 2^32 fragments");
  if ((rc = t1k_ensure(ctx, rs->bSend, (size_t)M * sizeof(T1kFragMeta) + Etot * sizeof(t1k_row_entry) + 512))) return fail(rc);
  if ((rc = t1k_ensure(ctx, rs->bRecv, recvF * sizeof(T1kFragMeta) + recvE * sizeof(t1k_row_entry) + 512))) return fail(rc);
  T1kFragMeta *sMeta = (T1kFragMeta *)rs->bSend.p;
  t1k_row_entry *sRows = (t1k_row_entry *)((char *)rs->bSend.p + (((size_t)M * sizeof(T1kFragMeta) + 255) & ~(size_t)255));
  T1kFragMeta *rMeta = (T1kFragMeta *)rs->bRecv.p;
  t1k_row_entry *rRows = (t1k_row_entry *)((char *)rs->bRecv.p + ((recvF * sizeof(T1kFragMeta) + 255) & ~(size_t)255));
  if (M) hipLaunchKernelGGL(k_ex_pack, dim3((M + 3) / 4), dim3(256), 0, st, idxS, k2, rs->rowPtr, rs->rowCount, rs->h1, rs->h2, fragBase, sMeta, sRows, M);
  RS_HIP(hipStreamSynchronize(st));
  if ((rc = t1k_comm_alltoallv(comm, sMeta, sOffM.data(), rMeta, rOffM.data()))) return cfail(rc);
  if ((rc = t1k_comm_alltoallv(comm, sRows, sOffE.data(), rRows, rOffE.data()))) return cfail(rc);
  // the rows this rank gave away are dead: the chunks and the send buffer go back
  for (auto &b : rs->chunks) if (b.p) { (void)t1k_dev_free(b.p); b.p = nullptr; b.bytes = 0; }
  (void)t1k_dev_free(rs->bSend.p); rs->bSend.p = nullptr; rs->bSend.bytes = 0;
  // the received fragments, in (source rank, source order) = global fragment order
  const uint64_t R = std::max<uint64_t>(recvF, 1);
  if ((rc = t1k_ensure(ctx, rs->bFrag2, R * (8 * 3 + 4 + 4) + 256))) return fail(rc);
  rs->rowPtr = (unsigned long long *)rs->bFrag2.p; rs->h1 = rs->rowPtr + R; rs->h2 = rs->h1 + R;
  rs->rowCount = (uint32_t *)(rs->h2 + R); rs->gid = rs->rowCount + R;
  if (recvF) {
    const uint32_t m = (uint32_t)recvF;
    const size_t m4 = ((size_t)(m + 2) * 4 + 255) & ~(size_t)255, m8 = ((size_t)(m + 2) * 8 + 255) & ~(size_t)255;
    if ((rc = t1k_ensure(ctx, rs->bWork, 8 * m4 + 3 * m8))) return fail(rc);
    uint32_t *c0 = (uint32_t *)rs->bWork.p;
    unsigned long long *e0 = (unsigned long long *)((char *)rs->bWork.p + 8 * m4);
    hipLaunchKernelGGL(k_ex_unpack_counts, dim3((m + 255) / 256), dim3(256), 0, st, rMeta, c0, m);
    RS_HIP(hipMemsetAsync(c0 + m, 0, 4, st));
    if ((rc = t1k_exclusive_sum64(ctx, c0, e0, m + 1))) return fail(rc);
    hipLaunchKernelGGL(k_ex_unpack, dim3((m + 255) / 256), dim3(256), 0, st, rMeta, e0, rRows, rs->rowPtr, rs->rowCount, rs->h1, rs->h2, rs->gid, m);
    RS_HIP(hipStreamSynchronize(st));
  }
  rs->nFragLocal = rs->nFrag;
  rs->nFrag = recvF;
  rs->exchanged = true;
  return T1K_OK;
}

// after t1k_rowset_coalesce on every rank: all ranks' group tables on every rank, concatenated in rank order
// (sizes[totalGroups], entries[totalEntries], firstFragment[totalGroups] -- global fragment indices)
int t1k_rowset_groups_gather(t1k_rowset *rs, t1k_comm *comm, uint64_t *totalGroups, uint64_t *totalEntries, uint64_t *totalAssigned) {
  if (!rs || !comm || !rs->coalesced) return rsFail(rs, T1K_ERR_STATE, "t1k_rowset_groups_gather: not coalesced");
  t1k_ctx *ctx = rs->owner;
  RS_HIP(hipSetDevice(rs->device));
  hipStream_t st = ctx->stream;
  const int N = t1k_comm_size(comm);
  int rc;
  std::vector<uint64_t> mine{rs->nGroups, rs->nEntries, rs->nAssigned}, all(3 * (size_t)N);
  if ((rc = t1k_comm_allgather_u64(comm, mine.data(), 3, all.data()))) { rs->err = t1k_comm_last_error(comm); return rc; }
  std::vector<uint64_t> bG(N), dG(N), bE(N), dE(N);
  uint64_t G = 0, E = 0, A = 0;
  for (int r = 0; r < N; ++r) { dG[r] = G * 4; dE[r] = E * sizeof(T1kGroupEnt); bG[r] = all[3 * r] * 4; bE[r] = all[3 * r + 1] * sizeof(T1kGroupEnt); G += all[3 * r]; E += all[3 * r + 1]; A += all[3 * r + 2]; }
  if ((rc = t1k_ensure(ctx, rs->bAll, (G + 1) * 8 + E * sizeof(T1kGroupEnt) + 1024))) { rs->err = ctx->err; return rc; }
  uint32_t *allSize = (uint32_t *)rs->bAll.p, *allFirst = allSize + (G + 1);
  T1kGroupEnt *allEnt = (T1kGroupEnt *)((char *)rs->bAll.p + (((G + 1) * 8 + 255) & ~(size_t)255));
  // own sizes from groupPtr
  T1kDevBuf tmp;
  if ((rc = t1k_ensure(ctx, tmp, (rs->nGroups + 1) * 4))) { rs->err = ctx->err; return rc; }
  if (rs->nGroups) hipLaunchKernelGGL(k_ex_sizes, dim3((unsigned)((rs->nGroups + 255) / 256)), dim3(256), 0, st, (const unsigned long long *)rs->bGroupPtr.p, (uint32_t *)tmp.p, (uint32_t)rs->nGroups);
  RS_HIP(hipStreamSynchronize(st));
  rc = t1k_comm_allgatherv(comm, tmp.p, bG.data(), dG.data(), allSize);
  if (rc == T1K_OK) rc = t1k_comm_allgatherv(comm, rs->bGroupFirst.p, bG.data(), dG.data(), allFirst);
  if (rc == T1K_OK) rc = t1k_comm_allgatherv(comm, rs->bGroupEnt.p, bE.data(), dE.data(), allEnt);
  (void)t1k_dev_free(tmp.p);
  if (rc != T1K_OK) { rs->err = t1k_comm_last_error(comm); return rc; }
  rs->allGroups = G; rs->allEntries = E;
  if (totalGroups) *totalGroups = G;
  if (totalEntries) *totalEntries = E;
  if (totalAssigned) *totalAssigned = A;
  return T1K_OK;
}

int t1k_rowset_groups_download_all(t1k_rowset *rs, uint32_t *sizes, t1k_group_entry *entries, uint32_t *firstFragment) {
  if (!rs || !rs->bAll.p) return rsFail(rs, T1K_ERR_STATE, "t1k_rowset_groups_download_all: nothing gathered");
  RS_HIP(hipSetDevice(rs->device));
  const uint64_t G = rs->allGroups, E = rs->allEntries;
  const uint32_t *allSize = (const uint32_t *)rs->bAll.p, *allFirst = allSize + (G + 1);
  const T1kGroupEnt *allEnt = (const T1kGroupEnt *)((const char *)rs->bAll.p + (((G + 1) * 8 + 255) & ~(size_t)255));
  if (sizes && G) RS_HIP(hipMemcpy(sizes, allSize, G * 4, hipMemcpyDeviceToHost));
  if (firstFragment && G) RS_HIP(hipMemcpy(firstFragment, allFirst, G * 4, hipMemcpyDeviceToHost));
  if (entries && E) RS_HIP(hipMemcpy(entries, allEnt, E * sizeof(T1kGroupEnt), hipMemcpyDeviceToHost));
  return T1K_OK;
}

}  // extern "C"
