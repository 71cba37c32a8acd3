// t1k_amd/csrc/t1k_launch.h -- internal: kernel argument blocks and host launchers shared by the .hip files
#pragma once
#include "t1k_dev.h"

struct AssignArgs {
  T1kRefDev ref;
  T1kReadsDev reads;
  int k, radius, hitLenRequired;
  double sim;
  int relax;
  // per-workgroup scratch
  uint32_t *wgHits; uint64_t hitCap;
  uint32_t *wgGroups;          // [wg][TILE_ALLELES][3]
  T1kCand *wgStage; uint32_t stageCap;
  uint32_t *wgThread;          // [wg][WG][THREAD_SCRATCH_U32]
  uint32_t *wgBig;             // [wg][3 * BIG_CAP + GA_SCRATCH_INTS]
  unsigned long long *wgCache; // [wg][GAP_CACHE] memo of gap alignments of the current read-end
  // outputs
  T1kCand *cand; uint64_t candCap;
  uint32_t *candStart, *candCount;
  unsigned long long *counters;  // [0] cand total, [1] ovl total, [2] error flags, [3] lookups, [4] postings, [5] hits, [6] groups, [7] dp, [8] slow queue len
};

struct ExtendArgs {
  T1kRefDev ref;
  T1kReadsDev reads;
  int k;
  double sim;
  const T1kCand *cand;
  T1kExt *ext;
  uint64_t nCand;
  unsigned long long *counters;
};

struct SelectArgs {
  T1kReadsDev reads;
  const T1kCand *cand;
  const T1kExt *ext;
  const uint32_t *candStart, *candCount;
  T1kOvl *ovl; uint64_t ovlCap;
  uint32_t *ovlStart, *ovlCount;
  uint64_t *sortScratch;   // [wg][sortCap] keys, then [wg][sortCap] u32 idx
  uint32_t sortCap;
  unsigned long long *counters;
};

struct FullArgs {
  T1kRefDev ref;
  T1kReadsDev reads;
  int relax;
  T1kOvl *ovl;
  uint64_t nOvl;
  uint32_t *slowQueue; uint32_t slowCap;
  unsigned long long *counters;
};

struct SlowArgs {
  T1kRefDev ref;
  T1kReadsDev reads;
  int relax;
  T1kOvl *ovl;
  const uint32_t *slowQueue;
  uint32_t nSlow;
  uint8_t *scratch; uint64_t perThread;  // per thread: int rows[GA_SCRATCH_INTS] | int8 ops[] | trace bytes
  int maxCells;
  unsigned long long *counters;
};

struct TruncArgs {
  T1kReadsDev reads;
  T1kOvl *ovl;
  const uint32_t *ovlStart;
  uint32_t *ovlCount;
  uint64_t *sortScratch; uint32_t sortCap;   // keys + idx, then a T1kOvl staging area of sortCap records
  unsigned long long *counters;
};

int t1k_launch_pack(t1k_ctx *ctx, const char *dAscii, const uint64_t *dOffs, uint32_t n, int S, uint64_t *bases, uint64_t *nmask, uint16_t *lens);
size_t t1k_seed_chain_lds(int S);
size_t t1k_wg_groups_u32();
size_t t1k_wg_thread_u32();
size_t t1k_wg_big_u32();
size_t t1k_wg_cache_u64();
size_t t1k_slow_per_thread(int maxCells);
void t1k_launch_seed_chain(t1k_ctx *ctx, const AssignArgs &a, int nWg);
void t1k_launch_extend(t1k_ctx *ctx, const ExtendArgs &a);
void t1k_launch_select(t1k_ctx *ctx, const SelectArgs &a, int nWg);
void t1k_launch_fullalign(t1k_ctx *ctx, const FullArgs &a);
void t1k_launch_fullalign_slow(t1k_ctx *ctx, const SlowArgs &a, int nBlocks);
void t1k_launch_fullalign_eq(t1k_ctx *ctx, const SlowArgs &a, int nBlocks);
void t1k_launch_truncate(t1k_ctx *ctx, const TruncArgs &a, int nWg);
void t1k_launch_coverage_scan(t1k_ctx *ctx, const T1kRefDev &ref, int32_t *out, const uint64_t *outOff);
