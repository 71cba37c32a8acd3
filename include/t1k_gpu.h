/* include/t1k_gpu.h -- C ABI of libt1k_gpu.so: the MI355X (gfx950) genotyper hot path of T1K.
 *
 * The reference (mourisl/T1K) has no library/FFI seam for this path: its seam is the `genotyper` process
 * (run-t1k:430,434) and, inside it, the C++ methods listed below.  Each entry point names the reference routine it
 * replaces (paths relative to the reference repository root).  All functions are extern "C", take plain pointers
 * and sizes, return 0 on success or a negative t1k_status, never throw and never exit().  A t1k_ctx owns one GPU
 * (one HIP stream); it is driven by one host thread at a time; different contexts are independent.
 *
 * Layers:
 *   (1) device stage API      t1k_ctx_* / t1k_ref_upload / t1k_reads_upload / t1k_assign_batch / t1k_pair_batch /
 *                             t1k_align_batch / t1k_em_*          -- what a cgo/JNI/ctypes binding would call
 *   (2) whole-stage job API   t1k_job_*  (host C++ around (1): FASTA/FASTQ parsing, packing, group coalescing,
 *                             equivalence classes, allele selection, TSV writers) and t1k_genotyper_main(), the
 *                             argv-compatible replacement of the reference's genotyper main() (Genotyper.cpp:194-738).
 */
#ifndef T1K_GPU_H
#define T1K_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  T1K_OK = 0,
  T1K_ERR_ARG = -1,       /* bad argument */
  T1K_ERR_DEVICE = -2,    /* HIP error (message in t1k_last_error) */
  T1K_ERR_CAPACITY = -3,  /* a device arena overflowed; raise the matching t1k_params cap and retry */
  T1K_ERR_IO = -4,        /* file could not be opened / parsed */
  T1K_ERR_STATE = -5,     /* call made in the wrong order */
  T1K_ERR_INTERNAL = -6,  /* an internal invariant did not hold (a bug; results of the call are void) */
  T1K_ERR_COMMITTED = -7  /* a capacity limit was hit after part of the range's per-base coverage had been added: do not retry the range on this
                             context (its coverage is void); run again with smaller ranges */
} t1k_status;

typedef struct t1k_ctx t1k_ctx;
typedef struct t1k_job t1k_job;
typedef struct t1k_rowset t1k_rowset;
typedef struct t1k_comm t1k_comm;

/* Genotyper.cpp:218-229 defaults; SeqSet.hpp:760-772 constants */
typedef struct {
  int32_t kmer_length;          /* 11 (Genotyper.cpp:207) */
  int32_t radius;               /* 10 (SeqSet.hpp:763) */
  int32_t hit_len_required;     /* 31 (SeqSet.hpp:764) */
  double ref_seq_similarity;    /* -s, default 0.8 */
  int32_t relax_intron_align;   /* --relaxIntronAlign */
  int32_t max_assign_cnt;       /* -n, default 2000 */
  /* device arena sizing (0 = defaults).  The *_cap values are LIMITS: what a context allocates follows the demand of its ranges */
  int32_t max_read_len;         /* longest read accepted, default (and most) 1000 */
  int32_t workgroups;           /* persistent workgroups of the per-read-end kernels, default 2048 */
  int64_t group_cap;            /* (read-end, strand, allele) hit groups per batch */
  int64_t cand_cap;             /* candidate records per batch */
  int64_t ovl_cap;              /* overlap records of one t1k_assign_range */
  int64_t row_cap;              /* fragment-row entries per batch */
  int32_t n_base_code;          /* the two bits a non-ACGT base contributes to a k-mer code (the code of a window holding one still decides
                                   whether its neighbour repeats the previous k-mer): 3 in the genotyper (nucToNum, Genotyper.cpp:37-40:
                                   -1 & 3), 0 in fastq-extractor (FastqExtractor.cpp:51-54: 'N' -> 0).  t1k_params_default sets 3. */
  int32_t store_chunk_mb;       /* the overlap store (final overlap lists, resident until the next read upload) grows in chunks of
                                   this many MB (at least one range's worth); default 1536 */
} t1k_params;

void t1k_params_default(t1k_params *p);

/* ---- context ------------------------------------------------------------------------------------------------ */
int t1k_ctx_create(int device, const t1k_params *params, t1k_ctx **out);
void t1k_ctx_destroy(t1k_ctx *ctx);
const char *t1k_last_error(const t1k_ctx *ctx);
int t1k_device_count(void);
int t1k_device_memory(int device, uint64_t *freeBytes, uint64_t *totalBytes); /* free = what the driver reports + the blocks this library keeps cached for reuse (t1k_pool_release hands those back) */

/* ---- reference: SeqSet::InputRefSeq + KmerIndex::BuildIndexFromRead (SeqSet.hpp:906-982, KmerIndex.hpp:107-130) --
 * seqs: nAlleles sequences concatenated as ASCII (ACGT, anything else is treated as N); offsets[nAlleles+1] byte offsets;
 * exon: one byte per base, non-zero = exon position (_validDiff::exon, SeqSet.hpp:638-723).  Packs to 2-bit + masks,
 * builds the direct-address 4^k index (postings in (allele, offset) order with the reference's insert rule). */
int t1k_ref_upload(t1k_ctx *ctx, const char *seqs, const uint64_t *offsets, const uint8_t *exon, uint32_t nAlleles);

/* ---- reads: one batch of read-ends resident in HBM -------------------------------------------------------------
 * seqs: nReadEnds ASCII reads concatenated; offsets[nReadEnds+1]; weights[nReadEnds] = multiplicity of the read-end
 * (the run length of identical sequences, Genotyper.cpp:463-480), NULL = all 1. */
int t1k_reads_upload(t1k_ctx *ctx, const char *seqs, const uint64_t *offsets, const uint32_t *weights, uint32_t nReadEnds);
/* The same upload in pieces, for a caller that gathers the text through a small page-locked staging buffer (t1k_pinned_alloc) while
 * earlier pieces are on their way: begin (read-ends, total text bytes, longest read), then pieces of the text (what = 0) and of the
 * offset array (what = 1, nReadEnds + 1 uint64; byteOffset / bytes in bytes of that array) in any order -- each an asynchronous copy
 * on the context's stream, tagged with the staging slot (0..3) it was sent from; t1k_reads_upload_wait(slot) returns once the last
 * piece sent from that slot has left the host buffer, which may then be refilled -- then end (packs the reads; every read-end gets
 * weight 1).  What Genotyper.cpp:451-480 does when it collects the read-ends of a run. */
int t1k_reads_upload_begin(t1k_ctx *ctx, uint32_t nReadEnds, uint64_t textBytes, int maxReadLen);
int t1k_reads_upload_piece(t1k_ctx *ctx, int what, const void *src, uint64_t byteOffset, uint64_t bytes, int slot);
int t1k_reads_upload_wait(t1k_ctx *ctx, int slot);
int t1k_reads_upload_end(t1k_ctx *ctx);

/* Identical read-ends collapse onto one representative whose weight is the sum of theirs: the sort + run-length loop of
 * Genotyper.cpp:451-480 (AssignRead is called once per distinct sequence with weight = multiplicity; the weight only feeds the
 * per-base coverage, SeqSet.hpp:2253-2274).  After the call the context's read set is the nDistinct distinct sequences (all
 * later read-end indices -- t1k_assign_range, t1k_pair_batch, t1k_pair_into -- refer to it) and distinctOf[i] is the index of
 * uploaded read-end i's sequence in it.  distinctOf: host array of the uploaded size. */
int t1k_reads_dedupe(t1k_ctx *ctx, uint32_t *distinctOf, uint32_t *nDistinct);

/* One overlap of a read-end on an allele: SeqSet::_overlap (SeqSet.hpp:89-144) after AssignRead. 48 bytes. */
typedef struct {
  int32_t seq_idx, read_start, read_end, seq_start, seq_end, strand;
  int32_t match_cnt, left_clip, right_clip, relaxed_match_cnt;
  double similarity;
} t1k_overlap;

/* SeqSet::AssignRead for every read-end of the uploaded batch (SeqSet.hpp:2119-2303: GetHitsFromRead 1071,
 * GetOverlapsFromHits 1232, GetOverlapsFromRead 1594, ExtendOverlap 1994, near-best GlobalAlignment + base coverage
 * 2188-2285).  Results stay on the device; per-base coverage accumulates in the context. */
int t1k_assign_batch(t1k_ctx *ctx);
/* the same for the sub-range [first, first+count) of the read set.  The final overlap lists of every read-end assigned since
 * the read set was uploaded stay resident (the "overlap store", grown in chunks of store_chunk_mb) and are
 * published in a table shared by all contexts that alias the read set (t1k_reads_share / t1k_reads_attach): mate pairing finds
 * both mates' lists whichever call, on whichever context, produced them.  t1k_overlaps_download returns the lists of the last
 * call only (its read-end indices are relative to `first`). */
int t1k_assign_range(t1k_ctx *ctx, uint64_t first, uint32_t count);
/* copy the overlap lists out (tests / --outputReadAssignment): counts[nReadEnds]; ovl may be NULL to query the total */
int t1k_overlaps_download(t1k_ctx *ctx, uint32_t *counts, t1k_overlap *ovl, uint64_t cap, uint64_t *total);

/* One kept (fragment, allele) assignment: Genotyper::_readAssignment (Genotyper.hpp:44-56). 24 bytes. */
typedef struct {
  int32_t allele_idx, start, end;
  float weight, qual, adjust_weight;
} t1k_row_entry;

/* SeqSet::ReadAssignmentToFragmentAssignment + Genotyper::SetReadAssignments for nFragments fragments
 * (SeqSet.hpp:2310-2655, Genotyper.hpp:778-832).  end1[i]/end2[i] index the context's read set (after t1k_reads_dedupe: the
 * distinct sequences; end2 NULL = single-end run, "-u"); both read-ends must have been assigned since the upload;
 * hasN[i] != 0 if either mate contains an N (Genotyper.cpp:537-539).  Rows stay on the device, in the reference's row order. */
int t1k_pair_batch(t1k_ctx *ctx, const uint32_t *end1, const uint32_t *end2, const uint8_t *hasN, uint32_t nFragments);
/* rowCounts[nFragments]; fragAssigned[nFragments] = fragmentAssigned flag (Genotyper.cpp:564-565) */
int t1k_rows_download(t1k_ctx *ctx, uint32_t *rowCounts, uint8_t *fragAssigned, t1k_row_entry *rows, uint64_t cap, uint64_t *total);

/* ---- all fragment rows of a job, resident on one GPU, and Genotyper::CoalesceReadAssignments over them ------------------------
 * (Genotyper.hpp:841-908).  t1k_pair_into is t1k_pair_batch writing fragment i's row as fragment fragBase + i of the rowset, ordered
 * by allele index (the order of a coalesced group's entries, 847-853), alleles outside the whitelist left out (822-823); several
 * contexts of one GPU may append concurrently.  t1k_rowset_coalesce then folds all fragments into read groups in fragment order:
 * groups numbered by first appearance, start = min, the `end` rule of 893-894 and the float weight sums evaluated in exactly the
 * reference's order (bit-identical, whatever the batching).  Results: groupPtr[nGroups + 1] into entries[nEntries];
 * firstFragment[nGroups] = the fragment that opened each group. */
typedef struct {
  int32_t allele, start, end;
  float weight, adjust_weight;
} t1k_group_entry;
int t1k_rowset_create(t1k_ctx *owner, uint64_t nFragments, const uint8_t *whitelist /* [nAlleles] or NULL */, t1k_rowset **out);
/* A rowset made for an upper bound of the fragment count (an input that is still being inflated when the job starts, t1k_reads_open_stream):
 * the fragments that turned out to exist.  No counterpart in the reference, whose vectors grow (Genotyper.cpp:365-440). */
int t1k_rowset_trim(t1k_rowset *rs, uint64_t n_fragments);
void t1k_rowset_destroy(t1k_rowset *rs);
/* raw != 0: t1k_pair_into stores the fragment assignment list itself (SeqSet::ReadAssignmentToFragmentAssignment's result) without the
 * -n / separator / whitelist drops of SetReadAssignments -- what the analyzer's BarcodeSummary::AddFragment reads (BarcodeSummary.hpp:24-57) */
int t1k_rowset_set_raw(t1k_rowset *rs, int raw);
/* device memory the row chunks hold so far and (rowEntries, may be NULL) the row entries written so far: the job layer projects the whole
 * job's rows from them when it decides which windows keep their read sets */
int t1k_rowset_device_bytes(t1k_rowset *rs, uint64_t *bytes, uint64_t *rowEntries);
const char *t1k_rowset_last_error(const t1k_rowset *rs);
int t1k_pair_into(t1k_ctx *ctx, t1k_rowset *rs, const uint32_t *end1, const uint32_t *end2, const uint8_t *hasN, uint32_t nFragments, uint64_t fragBase);
int t1k_rowset_coalesce(t1k_rowset *rs, uint64_t *nGroups, uint64_t *nEntries, uint64_t *assignedFragments);
/* the same; `sized` (may be NULL) is called once with the number of groups and entries as soon as they are known -- the fold of the
 * groups (the longest step) is then still running on the device, so a caller can size its host tables beside it.  Not called when no
 * fragment has a row. */
int t1k_rowset_coalesce_sized(t1k_rowset *rs, uint64_t *nGroups, uint64_t *nEntries, uint64_t *assignedFragments, void (*sized)(uint64_t nGroups, uint64_t nEntries, void *user), void *user);
int t1k_rowset_groups_download(t1k_rowset *rs, uint64_t *groupPtr, t1k_group_entry *entries, uint32_t *firstFragment);
/* fragAssigned[nFragments]: Genotyper.cpp:564-565 */
int t1k_rowset_assigned_download(t1k_rowset *rs, uint8_t *fragAssigned);
/* the flags of fragments [first, first + count), callable while t1k_pair_into calls for later fragments are in flight (the job's
 * output writer follows the windows of the device loop with it) */
int t1k_rowset_assigned_range(t1k_rowset *rs, uint64_t first, uint64_t count, uint8_t *fragAssigned);
/* rows of fragments [first, first + count) in the reference's row order (--outputReadAssignment, tests) */
int t1k_rowset_rows_download(t1k_rowset *rs, uint64_t first, uint32_t count, uint32_t *rowCounts, t1k_row_entry *rows, uint64_t cap, uint64_t *total);

/* ---- multi-GPU: one rank per GPU, each owning a contiguous slice of the fragments in file order -------------------
 * Ranks are processes (one per GPU, e.g. under torch.distributed.run: rank 0 calls t1k_comm_unique_id and hands the 128 bytes to
 * the others) or threads of one process (t1k_comm_group_create: `genotyper --gpus N`).  Transport: RCCL over xGMI (librccl.so.1,
 * bound lazily); ranks of one process that share a device use an in-process transport (transport = 0, or automatically): that is
 * how the multi-rank path is tested on one GPU.
 *   t1k_comm_allreduce            in-place sum over the ranks (kind 0: int32 -- the coverage arrays; 1: f64 -- the EM contributions)
 *   t1k_rowset_exchange           every fragment row goes to the rank that owns its pattern (hash mod nRanks); that rank coalesces the
 *                                 group over ALL its fragments in global order, so group contents do not depend on the sharding
 *   t1k_rowset_groups_gather      every rank's group table on every rank; merged by first fragment on the host (the job layer)
 *   t1k_em_shard                  t1k_em_update runs the row pass (one sum per read group) on [rowBegin, rowEnd) only (collective: the
 *                                 ranks' ranges must partition the read groups in rank order), all-gathers the ranks' pieces of the
 *                                 per-group sums (every element has one writer: nothing is added, 8 B per read group in all), then the
 *                                 class pass everywhere in group order: same doubles as on one GPU */
typedef struct t1k_comm_group t1k_comm_group;
int t1k_comm_unique_id(void *id128);
t1k_comm_group *t1k_comm_group_create(int nRanks);
void t1k_comm_group_destroy(t1k_comm_group *g);
int t1k_comm_init(t1k_ctx *ctx, int nRanks, int rank, const void *id128, t1k_comm_group *group, int transport /* -1 auto, 0 in-process, 1 RCCL */, t1k_comm **out);
int t1k_comm_bind(t1k_comm *c, t1k_ctx *ctx);  /* use the communicator with another context of the same device */
/* a rank that cannot go on says so before it returns: ranks waiting for it are released and every later collective of the job fails
 * with T1K_ERR_STATE instead of waiting (in-process transport: the meeting points; RCCL: ncclCommAbort) */
int t1k_comm_abort(t1k_comm *c);
void t1k_comm_destroy(t1k_comm *c);
const char *t1k_comm_last_error(const t1k_comm *c);
int t1k_comm_rank(const t1k_comm *c);
int t1k_comm_size(const t1k_comm *c);
int t1k_comm_is_rccl(const t1k_comm *c);
int t1k_comm_allreduce(t1k_comm *c, void *dev, uint64_t count, int kind);
int t1k_comm_allgather_u64(t1k_comm *c, const uint64_t *mine, uint32_t k, uint64_t *all);
int t1k_comm_alltoallv(t1k_comm *c, const void *sendbuf, const uint64_t *sendOff, void *recvbuf, const uint64_t *recvOff);
int t1k_comm_allgatherv(t1k_comm *c, const void *mine, const uint64_t *bytes, const uint64_t *displ, void *out);
int t1k_comm_allgatherv_host(t1k_comm *c, void *host, const uint64_t *bytes, const uint64_t *displ, uint64_t total);
int t1k_rowset_exchange(t1k_rowset *rs, t1k_comm *comm, uint64_t fragBase);
int t1k_rowset_groups_gather(t1k_rowset *rs, t1k_comm *comm, uint64_t *totalGroups, uint64_t *totalEntries, uint64_t *totalAssigned);
int t1k_rowset_groups_download_all(t1k_rowset *rs, uint32_t *sizes, t1k_group_entry *entries, uint32_t *firstFragment);
int t1k_em_shard(t1k_ctx *ctx, uint32_t rowBegin, uint32_t rowEnd, t1k_comm *comm);

/* per-base coverage of each allele's own base (posWeight[pos].count[base], SeqSet.hpp:2253-2274, read back by
 * GetSeqMissingBaseCoverage 2717-2755).  out[sum of allele lengths], alleles concatenated in upload order. */
int t1k_coverage_get(t1k_ctx *ctx, int32_t *out, uint64_t cap);
int t1k_coverage_reset(t1k_ctx *ctx);
/* per allele: number of exon positions whose coverage is below max(1, 1 % of the allele's median exon coverage)
 * (SeqSet::GetSeqMissingBaseCoverage, SeqSet.hpp:2717-2755), computed on the device; missing[nAlleles] */
int t1k_missing_coverage(t1k_ctx *ctx, int32_t *missing);
/* ---- per-base coverage only where it is read ------------------------------------------------------------------------------------
 * posWeight (SeqSet.hpp:2253-2274) has one reader, GetSeqMissingBaseCoverage (2717-2755) via Genotyper::FinalizeReadAssignments
 * (Genotyper.hpp:935), and alleleInfo[].missingCoverage is consumed for the SELECTED alleles of a gene only (Genotyper.hpp:1754,
 * 1870-1878; its use in EMupdate is overwritten by `adjust = 1`, 389-390 / 400-401).  A context in deferred mode
 * (t1k_ctx_set_coverage_mode(ctx, 1)) therefore adds no coverage in t1k_assign_range -- without --relaxIntronAlign it skips the
 * near-best full alignments altogether (relaxedMatchCnt = matchCnt there, SeqSet.hpp:2247-2250), with it they run for the relaxed
 * counts alone.  The caller keeps the finished read sets instead of letting the next upload overwrite them:
 *   t1k_readset_detach(reader)        the distinct read-ends of the reader context and the table of their final lists change owner
 *   t1k_readset_take_store(rs, pipe)  ... and so do the overlap-store chunks of slot `slot` of every context that assigned ranges of it
 * and, once allele selection has its candidates (Genotyper.hpp:1462-1695),
 *   t1k_coverage_selected(ctx, rs, selected[nAlleles])  adds to ctx's coverage arrays what t1k_assign_range would have added for the
 *                                     overlaps on the alleles with selected[a] != 0: identical per-base coverage for those alleles
 * (integer sums), so t1k_missing_coverage returns the same numbers for them.  *nRecords = records aligned (may be NULL). */
typedef struct t1k_readset t1k_readset;
int t1k_ctx_set_coverage_mode(t1k_ctx *ctx, int deferred);
int t1k_readset_detach(t1k_ctx *reader, t1k_readset **out);
int t1k_readset_take_store(t1k_readset *rs, t1k_ctx *pipe, int slot);
uint64_t t1k_readset_bytes(const t1k_readset *rs);   /* device memory the read set holds */
uint32_t t1k_readset_size(const t1k_readset *rs);    /* distinct read-ends */
const char *t1k_readset_last_error(const t1k_readset *rs);
void t1k_readset_destroy(t1k_readset *rs);
int t1k_coverage_selected(t1k_ctx *ctx, t1k_readset *rs, const uint8_t *selected, uint64_t *nRecords);
/* Identical read-ends across the windows of a job (the reference sorts ALL read-ends and assigns each distinct sequence once,
 * Genotyper.cpp:451-480; a window only sees its own).  A table of the sequences assigned in the windows whose lists stay resident:
 *   t1k_xwin_link(x, reader, w, &n)  after t1k_reads_dedupe of window w: its read-ends found in the table are marked (t1k_assign_range
 *                                    skips them), the others are entered; n = how many were found
 *   t1k_xwin_resolve(x, ctx, w)      once every earlier window's assignment ranges are done: the marked read-ends' list-table entries are
 *                                    copied from the windows that hold their lists (before any fragment of w is paired) */
typedef struct t1k_xwin t1k_xwin;
int t1k_xwin_create(t1k_ctx *owner, uint64_t maxReadEnds, uint32_t maxWindows, t1k_xwin **out);
void t1k_xwin_destroy(t1k_xwin *x);
const char *t1k_xwin_last_error(const t1k_xwin *x);
int t1k_xwin_link(t1k_xwin *x, t1k_ctx *reader, uint32_t window, uint32_t *nExternal);
int t1k_xwin_resolve(t1k_xwin *x, t1k_ctx *ctx, uint32_t window);

/* Several contexts on one GPU (pipelines of one job) can share the read-only device data of one of them: dst aliases src's
 * reference (and gets its own, zeroed coverage array) / src's packed reads.  src must outlive dst. */
int t1k_ref_share(t1k_ctx *dst, const t1k_ctx *src);
/* Device memory of the library is pooled per process (a freed block is reused by the next job: fresh VRAM costs ~35 ms per GB of
 * driver zeroing; T1K_POOL_GB bounds what is kept).  This returns every cached block to the driver; result = bytes released. */
uint64_t t1k_pool_release(void);
/* Page-locked host memory from a per-process cache (t1k_pool_release empties it too): what a caller should gather read text into
 * before t1k_reads_upload -- from pageable memory a multi-GB upload is staged by the runtime on the calling thread and other
 * streams' small copies wait behind its pieces.  NULL when the memory cannot be pinned (the caller falls back to malloc).  No
 * counterpart in the reference (host-only program). */
void *t1k_pinned_alloc(uint64_t bytes);
void t1k_pinned_free(void *p);
int t1k_reads_share(t1k_ctx *dst, const t1k_ctx *src);
/* the same with a choice of the overlap-store slot (0 or 1) dst writes its lists to, emptied first if resetStore != 0: a job
 * keeps two read sets (windows of fragments) in flight, so the lists of one stay valid while the next is being assigned */
int t1k_reads_attach(t1k_ctx *dst, const t1k_ctx *src, int storeSlot, int resetStore);
/* adds src's coverage into dst's and clears src's; both contexts must live on the same device and hold the same reference */
int t1k_coverage_absorb(t1k_ctx *dst, t1k_ctx *src);

/* ---- candidate extraction: IsGoodCandidate (FastqExtractor.cpp:113-118) = !IsLowComplexity (FastqExtractor.cpp:89-111) &&
 * SeqSet::HasHitInSet (SeqSet.hpp:1915-1990: GetHitsFromRead 1071, fullest (strand, sequence) bucket 1929-1964, GetOverlapsFromHits 1232
 * with filter 0, mismatch threshold 1974-1979) for every fragment of the uploaded batch.  The context is created with the
 * extractor's parameters (kmer_length = SeqSet::InferKmerLength, hit_len_required as FastqExtractor.cpp:383-416 computes it,
 * ref_seq_similarity = -s) and its reference uploaded with t1k_ref_upload (exon = NULL; SeqSet::InputRefFa 872-904 keeps one
 * sequence per FASTA record).  Read-ends of a fragment are consecutive in the batch (endsPerFragment 1 or 2); the second end is
 * only tested when the first fails (FastqExtractor.cpp:459-464).  good[nReadEnds / endsPerFragment] receives 0 / 1.
 * stats (may be NULL, else 8 words) receives {read-ends screened, index look-ups of the read-ends the screen let through, postings of their
 * used lists, read-ends with a hit, read-ends chained, k_extract_screen ns, k_extract ns (HIP events on the context's stream), 1 if the
 * batch had to be run again in the large LDS shape (a bucket of more than 1024 hits; more than 8192 is T1K_ERR_CAPACITY)}. */
int t1k_extract_batch(t1k_ctx *ctx, uint32_t endsPerFragment, uint8_t *good, uint64_t *stats);

/* ---- AlignAlgo::GlobalAlignment (AlignAlgo.hpp:215-421) as a batch --------------------------------------------
 * job i aligns t = text[tOff[i] .. tOff[i]+tLen[i]) against p = pat[pOff[i] .. +pLen[i]) (ASCII, N = wildcard).
 * Outputs per job: score, number of MATCH / MISMATCH / indel columns (SeqSet::GetAlignStats, SeqSet.hpp:438-455) and,
 * if ops != NULL, the edit string (0 match,1 mismatch,2 insert,3 delete) at ops[opsOff[i]..], length in nOps[i]. */
/* production match-count routine (exact <=3-mismatch fast path + banded forward sweep) on equal-length jobs: nMatch only */
int t1k_align_count_batch(t1k_ctx *ctx, const char *text, const uint32_t *tOff, const char *pat, const uint32_t *pOff, const uint32_t *len, uint32_t nJobs,
                          int32_t *nMatch);
int t1k_align_batch(t1k_ctx *ctx, const char *text, const uint32_t *tOff, const uint32_t *tLen, const char *pat, const uint32_t *pOff,
                    const uint32_t *pLen, uint32_t nJobs, int32_t *score, int32_t *nMatch, int32_t *nMismatch, int32_t *nIndel, int8_t *ops,
                    const uint32_t *opsOff, uint32_t *nOps);

/* ---- EM: Genotyper::EMupdate / QuantifyAlleleEquivalentClass inner loop (Genotyper.hpp:372-421, 1234-1314) -------
 * CSR over read groups: rowPtr[nGroups+1], ecIdx[nnz] (distinct classes of each group in first-appearance order),
 * count[nGroups] (group read count), ecLen[nEc] (class effective length).  t1k_em_update performs one EMupdate:
 * x1 = EM(x0); returns sum|x1-x0| in *diff and the expected read counts in ecReadCount (all host pointers).
 * The summation order is the reference's (bit-identical doubles).  allreduce (may be NULL) is called on the
 * device-resident partial read-count vector between the E and M steps when the groups are sharded over GPUs. */
typedef void (*t1k_allreduce_fn)(void *dev_f64, uint64_t n, void *user);
int t1k_em_setup(t1k_ctx *ctx, const uint64_t *rowPtr, const uint32_t *ecIdx, const double *count, const int32_t *ecLen, uint32_t nGroups,
                 uint32_t nEc, t1k_allreduce_fn allreduce, void *user);
int t1k_em_update(t1k_ctx *ctx, const double *x0, double *x1, double *ecReadCount, double *diff);

/* ---- profiling counters of the last t1k_assign_batch (algorithmic-traffic terms of SURVEY.md 8d) --------------- */
typedef struct {
  uint64_t read_ends, lookups, postings, hits, groups, candidates, extended, near_best, dp_calls, rows, batches;
  /* kernel time (HIP events on the launch stream), summed over the batches of the last run:
     ms_seed = k_seed_groups, ms_chain = the remaining chain kernels, ms_fullalign = k_fullalign + DP kernels + k_truncate */
  double ms_seed, ms_chain, ms_extend, ms_select, ms_fullalign, ms_pair, ms_em, ms_total;
  /* job level (t1k_job_stats): read_ends above counts the DISTINCT read-ends the kernels ran on, read_ends_total all of them;
     pair_overlaps = overlap records read by mate pairing; dp_cells = cell updates of the traced alignment kernels;
     wall-clock phases of the last run: mapping + indexing the read files, the window loop on the device, coalescing + download,
     writing the output files */
  uint64_t read_ends_total, pair_overlaps, dp_cells;
  double ms_load, ms_device, ms_coalesce, ms_write;
} t1k_stats;
int t1k_stats_get(t1k_ctx *ctx, t1k_stats *out);
/* wall time (ms) the context has spent allocating device memory and the bytes it asked for (diagnostics) */
double t1k_alloc_ms(t1k_ctx *ctx, uint64_t *bytes);
/* the device blocks the context holds right now, in bytes; print != 0: one line on stderr naming every block of 64 MB and more (diagnostics, T1K_DEBUG_MEM) */
uint64_t t1k_ctx_mem_report(t1k_ctx *ctx, const char *tag, int print);

/* ---- whole-stage job API (host C++ + the device stages above) ------------------------------------------------- */
/* argv-compatible replacement of the reference's genotyper executable (Genotyper.cpp:194-738). Returns the exit code. */
int t1k_genotyper_main(int argc, char **argv);
/* argv-compatible replacement of the reference's fastq-extractor main() (FastqExtractor.cpp:260-626; run-t1k:377-403) */
int t1k_extractor_main(int argc, char **argv);
/* argv-compatible replacement of the reference's bam-extractor main() (BamExtractor.cpp:463-949; run-t1k:350): candidate reads from a
 * coordinate-sorted BAM file -- reads over the gene intervals, reads on alternative contigs and unaligned reads that pass
 * IsLowComplexity + SeqSet::HasHitInSet (t1k_extract_batch).  The BAM container is read natively (BGZF blocks inflated in parallel). */
int t1k_bam_extractor_main(int argc, char **argv);
/* argv-compatible replacement of the reference's analyzer main() (Analyzer.cpp:236-733; run-t1k:438-449): re-assignment of the aligned reads
 * to the selected alleles, novel-variant calling (VariantCaller.hpp: host code over the GPU's overlap lists and alignments, see
 * t1k_variants_call below) and the per-barcode expression table */
int t1k_analyzer_main(int argc, char **argv);

typedef struct {
  t1k_params dev;
  double filter_frac, filter_cov, cross_gene_rate, squarem_min_alpha; /* --frac --cov --crossGeneRate --squaremMinAlpha */
  int32_t allele_digit_units;                                         /* --alleleDigitUnits, -1 = automatic */
  char allele_delimiter;                                              /* --alleleDelimiter, 0 = automatic */
  int32_t threads;                                                    /* -t: host parser/packer threads */
  int32_t device;                                                     /* GPU ordinal; -1 = host-only job (group bookkeeping only, cannot run) */
  int32_t output_read_assignment;                                     /* --outputReadAssignment */
  int32_t batch_fragments;                                            /* fragments per device batch (0 = default) */
} t1k_job_params;
void t1k_job_params_default(t1k_job_params *p);

int t1k_job_create(const t1k_job_params *p, const char *refFasta, t1k_job **out);
void t1k_job_destroy(t1k_job *job);
const char *t1k_job_last_error(const t1k_job *job);
/* load + keep reads on the host (FASTA/FASTQ, optionally gz); file2 NULL = single-end; barcodeFile may be NULL */
int t1k_job_load_reads(t1k_job *job, const char *file1, const char *file2, const char *barcodeFile);
/* several files per mate, read back to back (ReadFiles::AddReadFile, Genotyper.cpp: every -u / -1 / -2 adds one); files2 NULL = single-end */
int t1k_job_load_reads_multi(t1k_job *job, const char *const *files1, uint32_t n1, const char *const *files2, uint32_t n2, const char *barcodeFile);
/* The same input opened on its own, before or beside t1k_job_create (both calls block; a caller with two threads -- the genotyper
 * executable, bench.py -- maps and indexes the read files while the reference is parsed and the contexts come up; the reference's main
 * does the two one after the other, Genotyper.cpp:226-232 then 365-454).  `threads` as -t (0 = the default of a job).
 * t1k_job_attach_reads hands the input to an UNSHARDED job exactly as t1k_job_load_reads_multi would have left it and consumes the handle
 * (also on failure); t1k_reads_close is for a handle that was never attached.  A rank of a sharded job (t1k_job_set_shard with a
 * communicator) gets T1K_ERR_STATE whatever the input: its open is the collective t1k_job_load_reads, including the cases in which that
 * call ends up indexing the whole input on every rank (a barcode file, gz, non-strict layouts). */
typedef struct t1k_reads t1k_reads;
int t1k_reads_open(const char *const *files1, uint32_t n1, const char *const *files2, uint32_t n2, const char *barcodeFile, int threads, t1k_reads **out);
/* The same for a job of ONE rank, with ordinary .gz inputs not waited for: the .gz files of every mate (four-line FASTQ, not bgzip-framed; the same
 * number of files for each mate, read back to back) -- and the barcode file with them when it is a .gz file too -- are inflated by a decoder that publishes its progress while a second thread indexes the records behind it, and t1k_job_run
 * takes the fragments as they arrive -- as the reference's record-at-a-time reader does (ReadFiles.hpp:13, 95, 155-204; kseq.h:94-150).
 * t1k_reads_fragments / t1k_job_fragments wait for the end of the stream.  Inputs that are not eligible (plain files, mates with different
 * numbers of files, small files, text whose first 4 MB are not strict four-line records, T1K_STREAM_GZ=0) are opened exactly as by
 * t1k_reads_open.  Text that leaves the strict layout further on (a blank line between records, a last record without its quality line:
 * things the whole-file reader takes as the reference's kseq does) ends the stream; t1k_job_run then opens the files whole by itself and
 * starts over (one line on stderr says so), so the caller sees the result of t1k_reads_open at the price of the time lost. */
int t1k_reads_open_stream(const char *const *files1, uint32_t n1, const char *const *files2, uint32_t n2, const char *barcodeFile, int threads, t1k_reads **out);
const char *t1k_reads_last_error(const t1k_reads *reads);
int t1k_reads_fragments(const t1k_reads *reads, uint64_t *nFragments);
void t1k_reads_close(t1k_reads *reads);
int t1k_job_attach_reads(t1k_job *job, t1k_reads *reads);
/* or hand reads over from memory: concatenated ASCII + offsets, mates parallel; ids may be NULL ("r<i>") */
int t1k_job_set_reads(t1k_job *job, const char *seq1, const uint64_t *off1, const char *seq2, const uint64_t *off2, uint32_t nFragments);
/* read-end assignment, pairing, coalescing, EC build, EM, allele selection (Genotyper.cpp:451-650) */
int t1k_job_run(t1k_job *job);
/* optional, before t1k_job_run: the *_aligned*.fa files (which only need the fragmentAssigned flags) are then written by background
 * threads while the classes are built and the EM runs; t1k_job_write_outputs with the same prefix waits for them */
int t1k_job_set_output_prefix(t1k_job *job, const char *prefix);
/* <prefix>_genotype.tsv, _allele.tsv, _aligned*.fa, (_assign.tsv) (Genotyper.cpp:653-718) */
int t1k_job_write_outputs(t1k_job *job, const char *prefix);
/* results in memory: one line per gene, same text as _genotype.tsv */
int t1k_job_genotype_text(t1k_job *job, char *buf, uint64_t cap, uint64_t *needed);
int t1k_job_counts(t1k_job *job, uint64_t *fragments, uint64_t *assignedFragments, uint64_t *groups, uint64_t *ecs, int32_t *emIterations);
int t1k_job_stats(t1k_job *job, t1k_stats *out);
t1k_ctx *t1k_job_ctx(t1k_job *job);
/* ---- multi-GPU jobs: one job per GPU (rank), every rank loads the same inputs and owns the fragments
 * [F * rank / nRanks, F * (rank + 1) / nRanks) in file order.  t1k_job_run then does, per rank: read-end assignment and pairing of its
 * slice (no collective), the coverage all-reduce, the row exchange + coalescing + group gather (t1k_rowset_exchange ...), the
 * replicated class build, the EM with a sharded row pass, and the replicated selection; every rank ends with the same result and the
 * complete fragmentAssigned flags, rank 0 writes the files.  The result is identical to a single-GPU run for any nRanks. */
int t1k_job_set_shard(t1k_job *job, int rank, int nRanks, t1k_comm *comm);
/* rank threads of one process: dst uses the read files src has mapped (t1k_job_load_reads on src only) */
int t1k_job_share_reads(t1k_job *dst, t1k_job *src);
int t1k_job_run_local(t1k_job *job);
/* the two halves of t1k_job_run: t1k_job_run_local = windows of fragments through the GPU up to the coalesced read groups (with the
 * exchanges of a sharded job), t1k_job_finish = classes, EM, pruning, selection */
int t1k_job_finish(t1k_job *job);
/* the job's group table as a byte string: [u64 nGroups][u64 nEntries][u64 assignedFragments][u64 groupPtr[nGroups+1]]
 * [u32 firstFragment[nGroups]][t1k_group_entry entries[nEntries]] */
int t1k_job_groups_serialize(t1k_job *job, void *buf, uint64_t cap, uint64_t *needed);
/* host half of the multi-GPU merge: the tables of all pattern owners become this job's table, ordered by first fragment */
int t1k_job_groups_merge(t1k_job *job, const void *const *bufs, const uint64_t *lens, uint32_t n);
/* host-side CoalesceReadAssignments (Genotyper.hpp:841-908) on caller-provided fragment rows, in order (tests; device = -1 jobs);
 * fragments[i] = global index of fragment i (NULL: 0, 1, ...) */
int t1k_job_coalesce_rows(t1k_job *job, const t1k_row_entry *rows, const uint32_t *rowCounts, const uint32_t *fragments, uint32_t nFragments);
/* device pointer + element count of the int32 coverage difference array (for an in-place all-reduce) */
int t1k_coverage_device(t1k_ctx *ctx, void **devPtr, uint64_t *count);

/* ---- novel-variant calling of the analyzer stage (VariantCaller.hpp:92-1311), host code ----------------------------------------------
 * One assignment of a fragment as SeqSet::ReadAssignmentToFragmentAssignment leaves it (_fragmentOverlap, SeqSet.hpp:146-173) with the
 * edit strings SeqSet::AddFragmentAlignmentInfo adds (SeqSet.hpp:2657-2681, 2758-2778: AlignAlgo::GlobalAlignment of the allele window
 * [seq_start, seq_end] against the read window [read_start, read_end] of the strand-corrected read -- t1k_align_batch): ops1 / ops2 are
 * where the strings of o1 / o2 start in the caller's `ops` bytes (0 match, 1 mismatch, 2 insert, 3 delete; no terminator). */
typedef struct {
  int32_t allele_idx;                 /* _fragmentOverlap::seqIdx */
  int32_t has_mate_pair, o1_from_r2;  /* SeqSet.hpp:155-156 */
  t1k_overlap o1, o2;                 /* o2 is read only when has_mate_pair */
  uint64_t ops1, ops2;
  uint32_t n_ops1, n_ops2;
} t1k_frag_assignment;
/* The overlaps behind a fragment's kept assignments: for every allele of the fragment's row (alleles[nAlleles], the row's order) the choice
 * SeqSet::ReadAssignmentToFragmentAssignment makes among the mates' overlaps on that allele (SeqSet.hpp:2310-2458: compatible pairs, or single
 * overlaps when one list is empty / the run is single-end, ranked by _fragmentOverlap::operator<), taken from the read-ends' final overlap
 * lists (t1k_overlaps_download).  paired = 0: a -u run (l2 ignored).  Fills allele_idx, has_mate_pair, o1_from_r2, o1, o2 of out[nAlleles]
 * (the ops fields are left 0).  T1K_ERR_ARG: an allele has no candidate in the lists.  Host code. */
int t1k_fragment_details(const t1k_overlap *l1, uint32_t n1, const t1k_overlap *l2, uint32_t n2, int paired, const int32_t *alleles, uint32_t nAlleles,
                         t1k_frag_assignment *out);
typedef struct {  /* _variant (VariantCaller.hpp:7-20) + the exonic coordinate OutputAlleleVCF prints */
  int32_t allele_idx, ref_pos, exon_pos;
  char ref, var;
  int32_t qual, group, output_group;
  double var_support, all_support, var_uniq_support;
} t1k_variant;
typedef struct t1k_variants t1k_variants;
/* VariantCaller::SetSeqAbundance + SetMaxVarGroupToResolve + ComputeVariant (249-271, 978-1140) on the alleles of `job` (a host-only job
 * will do: nothing here runs on the device).  abundance[nAlleles] = Genotyper::GetAlleleAbundance after the analyzer's EM
 * (Analyzer.cpp:609, 674); fragments in file order: fragment f has assignments asg[asgPtr[f] .. asgPtr[f + 1]) in the reference's list
 * order and reads read1[f][0 .. len1[f]) / read2[f][0 .. len2[f]) (read2 / len2 NULL: single-end).  var_max_group = --varMaxGroup.
 * The job must outlive the result. */
int t1k_variants_call(t1k_job *job, const double *abundance, int32_t var_max_group, uint32_t nFragments, const uint64_t *asgPtr, const t1k_frag_assignment *asg,
                      const int8_t *ops, const char *const *read1, const uint32_t *len1, const char *const *read2, const uint32_t *len2, t1k_variants **out);
uint32_t t1k_variants_count(const t1k_variants *v);
int t1k_variants_get(const t1k_variants *v, t1k_variant *out /* [t1k_variants_count] */);
/* the text of <prefix>_allele.vcf (VariantCaller::OutputAlleleVCF, 1202-1227); buf may be NULL to query the size */
int t1k_variants_vcf(const t1k_variants *v, char *buf, uint64_t cap, uint64_t *needed);
/* VariantCaller::AdjustFragmentAssignment (1229-1311) for one fragment: keep[i] = 1 for the assignments BarcodeSummary::AddFragment counts */
int t1k_variants_adjust(const t1k_variants *v, const t1k_frag_assignment *asg, uint32_t n, const int8_t *ops, const char *read1, uint32_t len1, const char *read2, uint32_t len2,
                        uint8_t *keep);
void t1k_variants_destroy(t1k_variants *v);

#ifdef __cplusplus
}
#endif
#endif /* T1K_GPU_H */
