// t1k_amd/csrc/t1k_launch.h -- internal: kernel argument blocks and host launchers shared by the .hip files
#pragma once
#include <mutex>
#include "t1k_dev.h"

#define T1K_USED_MASK_WORDS 10
struct ChainArgs {
  T1kRefDev ref;
  T1kReadsDev reads;
  int k, radius, hitLenRequired;
  double sim;
  uint32_t *recs; uint32_t recStride; uint64_t groupCap;   // group records: re|strand, allele, diag, meta, M[...]
  uint32_t *chunkStart, *chunkCount; int maxChunks;        // [re][maxChunks] runs of groups per (strand, allele chunk), reference order
  uint32_t *usedMask;                                      // [re][2][T1K_USED_MASK_WORDS] bit r = the list of the k-mer at read offset r (that strand) is used (k_near_hits rebuilds hits on near diagonals from it)
  uint32_t *usedOut, *usedCount;                           // [re][maxK][4] used k-mers (readOff, listStart, listLen, directory row); [re][2] counts per strand
  unsigned long long *memo;                                // [re][GAP_CACHE] memo of gap alignments
  uint32_t *jobList; uint32_t jobCap;
  uint32_t *retryList, *generalList, *bigList, *finishList;  // dense lists (filled by k_arena_compact)
  uint32_t *jobStr, *retryStr, *generalStr, *bigStr, *finishStr, *waveStr, *waveList, *slowStr, *slowList;  // striped arenas the kernels append to
  uint32_t groupSegCap, jobSegCap, listSegCap, rareSegCap, genCandSegCap, genHitSegCap;  // listSegCap: slow / retry / finish lists; rareSegCap: general / wave / big
  uint32_t maxK;  // upper bound of the k-mers of a read-end (both strands): stride of the used-list table
  uint32_t maxKFast;  // the same for the read-ends k_seed_groups takes (<= T1K_MAX_READ_LEN bases): its LDS layout; < maxK in a window with longer reads
  uint32_t *genJobStr, *genJobList; uint32_t genJobSegCap;  // alignments registered by the multi-diagonal groups
  uint32_t *genHits;  // hit lists of the multi-diagonal groups (k_gather_general)
  uint32_t *genCand; uint32_t genCandCap;                  // packed candidates of multi-diagonal groups (3 u32 each)
  uint32_t *bigScratch;
  int fuse;                // the seeding kernel runs the closed-form pass itself (k_seed_chain) and fills the gap-walk / multi-diagonal lists; 0 (T1K_FUSE_SEED=0): k_seed_groups + k_chain_fast<*, 0>
  int devDriven;             // this submission holds no counter fetch between its launches (runChainDevice): a launch behind an overflow ends at once
  int nearSimple;          // k_near_hits marks the two-diagonal groups whose hit list is its chain (T1K_NO_SIMPLE_CHAIN=1: leaves them to the general path)
  int earlyPrune;          // k_chain_fast<*, 0>: 0 = closed form only (T1K_NO_EARLY_PRUNE=1), 1 = + the groups that cannot pass the similarity filter by the gap-count bound
                           // (T1K_WALK_IN_CLOSED=0), 2 = + the gap walk's first pass: groups without a gap of more than three mismatches and groups its bound prunes (default)
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
  unsigned long long *memo;                 // per-read-end alignment memo (t1k_memo.h)
  uint32_t *jobStr; uint32_t jobSegCap;     // striped list of registered alignments
  uint32_t *retryStr; uint32_t retrySegCap; // striped list of candidates waiting for them
};

struct SelectArgs {
  T1kReadsDev reads;
  const T1kCand *cand;
  const T1kExt *ext;
  const uint32_t *candStart, *candCount;
  T1kOvl *ovl; uint64_t ovlCap;
  uint32_t *ovlStart, *ovlCount;
  uint64_t *sortScratch;   // [wg][sortCap] packed keys (per-workgroup stride: sortCap * 6 words, shared with k_truncate)
  uint32_t sortCap;
  int alleleBits;          // bits of an allele index
  int relax;               // --relaxIntronAlign: the relaxed match counts come from the full alignments (else k_select writes them)
  int xl;                  // the window holds reads beyond T1K_MAX_READ_LEN: wider sort-key fields
  unsigned long long *counters;
};

struct FullArgs {
  T1kRefDev ref;
  T1kReadsDev reads;
  int relax;
  int noCov;               // the alignments only feed the relaxed counts: no per-base coverage is added (t1k_ctx_set_coverage_mode 1)
  int fullLen;             // span counted in ref.covFull
  int undo;                // second pass over the same records after an alignment queue overflowed: the coverage the first pass added for the
                           // ungapped alignments is taken back (weights negated, integer atomics: exact) and nothing is queued -- the range can run again
  T1kOvl *ovl;
  uint64_t nOvl;
  uint32_t *eqStr, *bandStr, *wideStr; uint32_t segCap;  // striped alignment queues
  unsigned long long *eqKeyStr, *bandKeyStr;              // their sort keys (see k_fullalign)
  unsigned long long *counters;
};

struct SlowArgs {
  T1kRefDev ref;
  T1kReadsDev reads;
  int relax;
  int noCov;
  T1kOvl *ovl;
  const uint32_t *slowQueue;  // (sorted) queue of overlap indices
  uint32_t nSlow;
  const uint32_t *runOf, *rep; uint32_t nRuns; uint64_t traceStride;  // runs of identical jobs: 1-based run of each job, first job of each run
  uint8_t *scratch; uint64_t perThread;  // per thread: int rows[GA_SCRATCH_INTS] | int8 ops[] | trace bytes
  int maxCells;
  unsigned long long *counters;
};

struct TruncArgs {
  T1kReadsDev reads;
  T1kOvl *ovl;
  const uint32_t *ovlStart;
  uint32_t *ovlCount;
  uint64_t *sortScratch; uint32_t sortCap;   // keys, then a T1kOvl staging area of sortCap records
  int alleleBits;
  int xl;                  // as SelectArgs::xl
  unsigned long long *counters;
};

// All fragment rows of a job on one GPU (t1k_pair_into appends, several contexts concurrently) and their coalescing into read
// groups (t1k_coalesce.hip).  Rows live in chunks; a fragment's row is contiguous, ordered by allele index.
struct t1k_rowset {
  int device = 0;
  t1k_ctx *owner = nullptr;         // its stream runs the coalescing kernels
  uint64_t nFrag = 0;
  T1kDevBuf bFrag;                  // rowPtr | h1 | h2 | rowCount | assigned
  unsigned long long *rowPtr = nullptr, *h1 = nullptr, *h2 = nullptr;
  uint32_t *rowCount = nullptr;
  uint8_t *assigned = nullptr;
  // after t1k_rowset_exchange: the fragments this rank owns (bFrag2 / bRecv), gid = their global fragment index
  uint32_t *gid = nullptr;
  bool exchanged = false;
  uint64_t nFragLocal = 0;
  T1kDevBuf bSend, bRecv, bFrag2, bAll;
  uint64_t allGroups = 0, allEntries = 0;
  T1kDevBuf bCursors;               // one append cursor per chunk
  std::vector<T1kDevBuf> chunks;
  size_t cur = 0;
  uint64_t chunkEntries = 0;
  std::mutex m;
  T1kDevBuf bWhitelist;
  const uint8_t *whitelist = nullptr;
  bool rawKept = false;             // rows = the raw fragment assignment lists (analyzer)
  // coalescing results (device)
  T1kDevBuf bWork, bGroupPtr, bGroupEnt, bGroupFirst;
  uint64_t nGroups = 0, nEntries = 0, nAssigned = 0;
  bool coalesced = false;
  hipStream_t copyStream = nullptr; // t1k_rowset_assigned_range: copies beside the pipelines (non-blocking stream)
  std::string err;
};
int t1k_rowset_chunk(t1k_rowset *rs, t1k_ctx *ctx, size_t *chunk, t1k_row_entry **rows, uint64_t *cap, unsigned long long **cursor);
int t1k_rowset_chunk_full(t1k_rowset *rs, t1k_ctx *ctx, size_t chunk);
int t1k_exclusive_sum64(t1k_ctx *ctx, const uint32_t *in, unsigned long long *out, uint32_t n);
int t1k_inclusive_sum_n(t1k_ctx *ctx, const uint32_t *in, uint32_t *out, uint64_t n);
int t1k_exclusive_sum32(t1k_ctx *ctx, const uint32_t *in, uint32_t *out, uint64_t n);
int t1k_exclusive_sum_u64(t1k_ctx *ctx, const unsigned long long *in, unsigned long long *out, uint64_t n);

int t1k_launch_pack(t1k_ctx *ctx, const char *dAscii, const uint64_t *dOffs, uint32_t n, int S, uint64_t *bases, uint64_t *nmask, uint16_t *lens, int nCode);
size_t t1k_slow_per_thread(int maxCells);
size_t t1k_chain_big_scratch_u32();
int t1k_chain_max_chunks(uint32_t nAlleles);
int t1k_chain_memo_entries();
int t1k_chain_rec_stride(int maxLen);
int t1k_chain_max_kmers(int maxLen, int k);
int t1k_chain_used_u32(int maxK);
int t1k_run_chain(t1k_ctx *ctx, const ChainArgs &a, int nWg, int bigBlocks, bool longReads, unsigned long long *hc);
void t1k_launch_extend(t1k_ctx *ctx, const ExtendArgs &a);
void t1k_launch_select(t1k_ctx *ctx, const SelectArgs &a, int nWg);
void t1k_launch_fullalign(t1k_ctx *ctx, const FullArgs &a);
void t1k_launch_fullalign_slow(t1k_ctx *ctx, const SlowArgs &a, int nBlocks);
void t1k_launch_align_flags(t1k_ctx *ctx, const SlowArgs &a, uint32_t *flags);
void t1k_launch_align_reps(t1k_ctx *ctx, const uint32_t *flags, const uint32_t *runOf, uint32_t *rep, uint32_t n);
void t1k_launch_align_fill_apply(t1k_ctx *ctx, const SlowArgs &a, bool eq);
int t1k_inclusive_sum(t1k_ctx *ctx, const uint32_t *in, uint32_t *out, uint32_t n);
void t1k_launch_truncate(t1k_ctx *ctx, const TruncArgs &a, int nWg);
void t1k_launch_coverage_scan(t1k_ctx *ctx, const T1kRefDev &ref, int32_t *out, const uint64_t *outOff);
int t1k_fetch_counters(t1k_ctx *ctx, unsigned long long *h);
// near-best full alignments of nOvl working records (t1k_capi.hip): relaxed counts + (unless noCov) per-base coverage
int t1k_fullalign_phase(t1k_ctx *ctx, const T1kReadsDev &rd, T1kOvl *ovl, uint64_t nOvl, int relaxFlag, int noCov, int maxLen, bool exactQueues, unsigned long long *hc);
struct T1kArenaCounts { uint64_t total; uint32_t maxSeg; bool overflow; };
T1kArenaCounts t1k_arena_counts(const t1k_ctx *ctx, int arena, uint32_t segCap);  // from the last t1k_fetch_counters
void t1k_launch_coverage_add(t1k_ctx *ctx, int32_t *dst, int32_t *src, uint64_t n);
int t1k_coverage_fold(t1k_ctx *ctx);  // covFull -> covDiff (on the context's stream, not synchronised)
void t1k_launch_missing_coverage(t1k_ctx *ctx, const T1kRefDev &ref, int32_t *scratch, int32_t *missing);
void t1k_launch_extend_retry(t1k_ctx *ctx, const ExtendArgs &a, const uint32_t *list, uint32_t n);
void t1k_launch_dp_dense(t1k_ctx *ctx, const ChainArgs &a, const uint32_t *jobs, uint32_t n);
// device-driven forms (t1k_chain.hip, runChainDevice): item counts read on the device from the arena's total word, grids from estimates
void t1k_arena_compact_dev(t1k_ctx *ctx, int arena, const uint32_t *src, uint32_t segCap, uint32_t *dst, uint64_t estTotal);
void t1k_launch_dp_dense_dev(t1k_ctx *ctx, const ChainArgs &a, const uint32_t *jobs, int arena, uint32_t cap, uint64_t est);
uint64_t t1k_arena_estimate(const t1k_ctx *ctx, int arena, uint64_t cap, uint32_t nRe);
void t1k_arena_estimate_set(t1k_ctx *ctx, int arena, uint64_t total, uint32_t nRe);
bool t1k_chain_host_driven();
void t1k_launch_extend_retry_dev(t1k_ctx *ctx, const ExtendArgs &a, const uint32_t *list, int arena, uint64_t est);
void t1k_arena_compact64(t1k_ctx *ctx, int arena, const unsigned long long *src, uint32_t segCap, unsigned long long *dst, uint32_t maxSeg);
int t1k_sort_pairs(t1k_ctx *ctx, const unsigned long long *keysIn, unsigned long long *keysOut, const uint32_t *valsIn, uint32_t *valsOut, uint32_t n, int endBit = 64);
void t1k_arena_compact(t1k_ctx *ctx, int arena, const uint32_t *src, uint32_t segCap, uint32_t *dst, uint32_t maxSeg);

// t1k_extract.hip
void t1k_launch_extract(t1k_ctx *ctx, const T1kRefDev &ref, const T1kReadsDev &reads, int k, int radius, int hitLenRequired, double oneMinusSim, uint32_t nFragments,
                        uint32_t epf, uint32_t maxK, uint8_t *good, uint8_t *state, unsigned long long *err, unsigned long long *stats, int nWg);
void t1k_launch_extract_huge(t1k_ctx *ctx, const T1kRefDev &ref, const T1kReadsDev &reads, int k, int radius, int hitLenRequired, double oneMinusSim,
                             uint32_t nFragments, uint32_t epf, uint32_t maxK, uint8_t *good, uint8_t *state, unsigned long long *err, unsigned long long *stats, int nWg,
                             uint32_t *scratch, uint32_t cap);
void t1k_launch_extract_big(t1k_ctx *ctx, const T1kRefDev &ref, const T1kReadsDev &reads, int k, int radius, int hitLenRequired, double oneMinusSim,
                            uint32_t nFragments, uint32_t epf, uint32_t maxK, uint8_t *good, uint8_t *state, unsigned long long *err, unsigned long long *stats, int nWg);
