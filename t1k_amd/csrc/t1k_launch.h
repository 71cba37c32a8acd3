// t1k_amd/csrc/t1k_launch.h -- internal: kernel argument blocks and host launchers shared by the .hip files
#pragma once
#include "t1k_dev.h"

// one (read-end, strand, allele) hit group of the batch; 16 bytes
struct T1kGroup {
  uint32_t reStrand;   // read-end id | '+' strand << 31
  uint32_t allele;
  uint32_t hitStart;   // index into the batch hit arena; after chaining the slice holds the group's packed candidates
  uint32_t n;          // hit count; after chaining: candidate count | 1 << 30
};

struct ChainArgs {
  T1kRefDev ref;
  T1kReadsDev reads;
  int k, radius, hitLenRequired;
  double sim;
  uint32_t *hits; uint64_t hitCap;
  T1kGroup *groups; uint64_t groupCap;
  uint32_t *chunkStart, *chunkCount;          // [re][MAX_CHUNKS] runs of groups per (strand, allele tile), reference order
  unsigned long long *memo;                   // [re][GAP_CACHE] memo of gap alignments
  uint32_t *jobList; uint32_t jobCap;
  uint32_t *retryList, *generalList, *bigList;
  uint32_t *threadScratch, *bigScratch;
  T1kCand *cand; uint64_t candCap;
  uint32_t *candStart, *candCount;
  unsigned long long *counters;
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
size_t t1k_slow_per_thread(int maxCells);
size_t t1k_chain_thread_scratch_u32();
size_t t1k_chain_big_scratch_u32();
int t1k_chain_max_chunks();
int t1k_chain_memo_entries();
int t1k_run_chain(t1k_ctx *ctx, const ChainArgs &a, int nWg, int generalBlocks, int bigBlocks, unsigned long long *hc);
void t1k_launch_extend(t1k_ctx *ctx, const ExtendArgs &a);
void t1k_launch_select(t1k_ctx *ctx, const SelectArgs &a, int nWg);
void t1k_launch_fullalign(t1k_ctx *ctx, const FullArgs &a);
void t1k_launch_fullalign_slow(t1k_ctx *ctx, const SlowArgs &a, int nBlocks);
void t1k_launch_fullalign_eq(t1k_ctx *ctx, const SlowArgs &a, int nBlocks);
void t1k_launch_fullalign_band(t1k_ctx *ctx, const SlowArgs &a, int nBlocks);
void t1k_launch_truncate(t1k_ctx *ctx, const TruncArgs &a, int nWg);
void t1k_launch_coverage_scan(t1k_ctx *ctx, const T1kRefDev &ref, int32_t *out, const uint64_t *outOff);
