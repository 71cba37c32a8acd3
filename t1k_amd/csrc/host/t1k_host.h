// t1k_amd/csrc/host/t1k_host.h -- host C++ around the device stages: the parts of the genotyper stage that stay on
// the CPU: FASTA/FASTQ input, allele naming, read-group coalescing,
// equivalence classes, the SQUAREM control loop (the E-step runs on the GPU), allele selection, TSV writers.
#pragma once
#include <cstdint>
#include <functional>
#include <atomic>
#include <list>
#include <map>
#include <mutex>
#include <set>
#include <memory>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>
#include "../../../include/t1k_gpu.h"

namespace t1k {

struct SeqRec {
  std::string id, comment, seq;
  bool hasComment = false;
};
// FASTA / FASTQ, plain or gz; id = header up to the first blank with a trailing /1 or /2 removed (ReadFiles.hpp:185-189)
bool readSeqFile(const std::string &path, std::vector<SeqRec> &out, std::string &err);
// the allele reference's records as RefSet::load reads them (a FASTA of plain '>' records is parsed by all host threads, with the same result)
bool readReferenceRecords(const std::string &path, std::vector<SeqRec> &out, std::string &err);

// Large host blocks of a job ask for transparent huge pages (host/refset.cpp).  Measured with tools/hip_hello on the MI355X hosts (round 5): a
// first touch costs ~150 ms per GB in 4 KB pages, and the kernel takes a process's resident pages apart at ~75 ms per GB, single-threaded,
// before the parent sees the exit; the hosts run huge pages in `madvise` mode.  With the advice the 10 M-pair step is 2.5 % shorter (group
// tables, record index, EM arrays: 0.9 GB) -- profiles/r05_ab_thp.log.  T1K_NO_THP=1: no advice (A/B).
void bigBlockAdvise(void *p, size_t bytes);
constexpr size_t kBigBlock = (size_t)4 << 20;

// vectors that do not zero what they are about to receive (record index of the read files, group tables: hundreds of MB each)
template <class T>
struct NoInitAlloc : std::allocator<T> {
  template <class U> struct rebind { using other = NoInitAlloc<U>; };
  T *allocate(size_t n) {
    T *p = std::allocator<T>::allocate(n);
    if (n * sizeof(T) >= kBigBlock) bigBlockAdvise(p, n * sizeof(T));
    return p;
  }
  template <class U> void construct(U *p) noexcept { ::new ((void *)p) U; }
  template <class U, class A0, class... A> void construct(U *p, A0 &&a0, A &&...a) { ::new ((void *)p) U(std::forward<A0>(a0), std::forward<A>(a)...); }
};

// The read files of a job, whole in memory (mmap / inflated once) with a record index built in place by the host threads
// (host/reads.cpp).  Record i of mate m is side[m].seqP[i][0 .. seqL[i]) with name idP[i][0 .. idL[i]); nothing is copied.
struct ReadInput {
  struct Side {
    std::vector<const char *, NoInitAlloc<const char *>> seqP, idP;
    std::vector<uint32_t, NoInitAlloc<uint32_t>> seqL;
    std::vector<uint16_t, NoInitAlloc<uint16_t>> idL;
  };
  Side side[2], bc;            // mates; barcode records (sequence = the barcode)
  bool paired = false, hasBarcode = false, noIds = false;
  std::atomic<bool> inPlace{true};  // every file was indexed in place (strict layouts): records point into the file's own text, qualities included
  std::vector<uint32_t, NoInitAlloc<uint32_t>> frag;  // fragment f = record frag[f] (records with a missing barcode are dropped with their mates)
  int maxLen = 0;
  ReadInput();
  ReadInput(const ReadInput &) = delete;
  ReadInput &operator=(const ReadInput &) = delete;
  ~ReadInput();
  bool open(const std::vector<std::string> &files1, const std::vector<std::string> &files2, const std::string &barcodeFile, int threads, std::string &err);
  void setMemory(const char *seq1, const uint64_t *off1, const char *seq2, const uint64_t *off2, uint32_t n);
  size_t nFrag() const { return frag.size(); }   // fragments held here
  // One process per GPU (host/reads.cpp, openSharded): only the fragments [base, base + nFrag()) of nAll() are indexed here
  struct ShardComm {
    int rank = 0, nRanks = 1;
    std::function<bool(void *buf, const uint64_t *bytes, const uint64_t *displ, uint64_t total)> allgatherv;  // host buffers, in place
  };
  // 1 = indexed this rank's slice, 0 = not eligible (the caller opens the files whole, as a single process does), -1 = error
  int openSharded(const std::vector<std::string> &files1, const std::vector<std::string> &files2, int threads, const ShardComm &c, std::string &err);
  bool sharded = false;
  uint32_t base = 0;
  int shardRank = 0, shardRanks = 1;
  size_t nAll() const { return sharded ? nAll_ : frag.size(); }
  // Streaming open (host/reads.cpp): ordinary .gz files (the same number for every mate, read back to back as ReadFiles does), four-line FASTQ.  Each file is inflated by the decoder of host/inflate.cpp on
  // a thread of its own, a second thread indexes the records behind it and a third runs the CRC over the text; the tables are sized to an
  // upper bound of the record count (nFrag() while the stream runs) and the job's window loop takes records as they are published
  // (streamAvail).  streamFinish trims the tables to what was found.  false with err empty = not eligible: the caller opens the files whole.
  bool openStreaming(const std::vector<std::string> &files1, const std::vector<std::string> &files2, const std::string &barcodeFile, std::string &err);
  std::atomic<bool> streaming{false};  // opened by openStreaming and not finished yet: nFrag() / nAll() are the upper bound
  size_t streamAvail() const;  // records indexed in every mate so far (their table entries may be read)
  int streamState() const;     // 0 running, 1 every thread finished, -1 failed
  void streamWait(size_t records) const;  // until that many records are there or the stream has ended
  bool streamFinish(std::string &err);    // joins the threads, checks the mates against one another and the CRCs, trims the tables
  std::atomic<int> streamMaxLen{0};       // longest read indexed so far
  std::vector<std::string> streamFiles1, streamFiles2; std::string streamBarcodeFile;  // what openStreaming was given (the job opens them whole when the stream gives up)
  std::atomic<bool> streamGaveUp{false};  // the stream failed on text the whole-file reader takes (a record outside the strict layout behind the checked head, more records than the tables were sized for): open the files whole
  struct Stream;
  std::unique_ptr<Stream> stream_;
  // the mapped bytes of records [recLo, recHi) are not needed any more (their text went to the GPU and their output is written): drop
  // the page-table entries now, piece by piece beside the device loop, instead of all at once when the job is destroyed
  void release(size_t recLo, size_t recHi);
  // inflated text (anonymous memory) is dropped as well: only for a caller that runs the job ONCE (the executables) -- dropped pages read as
  // zeros, a second run over the same input would see empty reads.  A 10 M-pair .gz job otherwise leaves with 10 GB resident (0.4 s of exit).
  bool dropInflatedText = false;

 private:
  struct Blob {
    void *map = nullptr;
    size_t len = 0;
    bool anon = false;  // map is anonymous memory (inflated text), not a file: its pages cannot be dropped and read again
    std::unique_ptr<std::vector<char>> owned;
  };
  std::list<Blob> blobs_;  // the mates are read by concurrent threads: nodes never move, additions are serialised
  std::mutex blobLock_;
  Blob &newBlob() { std::lock_guard<std::mutex> g(blobLock_); blobs_.emplace_back(); return blobs_.back(); }
  void dropBlob(Blob &b) { std::lock_guard<std::mutex> g(blobLock_); for (auto it = blobs_.begin(); it != blobs_.end(); ++it) if (&*it == &b) { blobs_.erase(it); return; } }  // (a failed attempt's empty node)
  bool addFile(const std::string &path, int threads, Side &dst, std::string &err);
  static bool bgzfInflate(int fd, size_t fileSize, int threads, Blob &blob, const char *&data, size_t &size);
  static bool gzipInflate(int fd, size_t fileSize, Blob &blob, const char *&data, size_t &size);
  bool addBuffer(const char *p, size_t n, int threads, Side &dst, std::string &err, const std::string &what);
  bool addRange(const char *b, const char *stop, const char *end, bool fastq, int threads, Side &dst);
  uint32_t nAll_ = 0;
  bool addGeneral(const std::string &path, Side &dst, std::string &err);
  void finish();
};

// host/inflate.cpp: the gzip decoder that publishes how far it has got (bytes of text that are final), for readers that follow it
struct GzProgress {
  std::atomic<uint64_t> produced{0};
  std::atomic<int> state{0};  // 0 running, 1 finished, -1 failed
  // members that are complete: (offset of the member's end in the text, CRC-32 of its trailer), for the reader that checks the text behind the decoder
  std::mutex m;
  std::vector<std::pair<uint64_t, uint32_t>> members;
};
int gzInflateAll(const uint8_t *src, size_t srcLen, uint8_t *dst, size_t cap, GzProgress *pg, size_t *outLen, uint32_t *lastCrc, size_t *members, std::string &err, uint64_t progressBase = 0,
                 bool finishes = true);

struct AlleleMeta {
  std::string name;
  int seqLen = 0, effLen = 0, weight = 1;
  int gene = -1, major = -1;
  int missingCov = 0, ec = -1;
  int rank = -1, quality = -1;
  double abundance = 0, ecAbundance = 0;
};

struct RefSet {
  std::vector<AlleleMeta> al;
  std::vector<std::string> seqs;
  std::vector<std::vector<uint8_t>> exon;
  std::vector<std::string> geneName, majorName;
  std::vector<std::vector<double>> geneSim;  // geneSim[i][j] = how similar gene i's k-mers are to gene j's (KmerCount.hpp:196-216)
  bool rnaData = true;
  // load + merge identical sequences + exon masks + weights + naming (Genotyper::InitRefSet, Genotyper.hpp:707-730)
  // selected != NULL: only the records named in it (Genotyper::InitRefSet(filename, selectedAlleles), Genotyper.hpp:732-757)
  bool load(const std::string &fasta, int digitUnits, char delimiter, std::string &err, const std::set<std::string> *selected = nullptr);
  void splitName(const std::string &allele, std::string &gene, std::string &major, int fieldsType) const;
  int digitUnits = -1;
  char delimiter = 0;
};

struct GroupEntry {
  int allele, start, end;
  float weight, adjustWeight;
};
typedef std::vector<GroupEntry, NoInitAlloc<GroupEntry>> GroupVec;

struct Genotyper {
  RefSet *ref = nullptr;
  t1k_job_params prm;
  // coalesced read groups (Genotyper::readAssignments, Genotyper.hpp:443), CSR
  std::vector<uint64_t> groupPtr{0};
  GroupVec groupEnt;
  std::vector<uint32_t> groupFirst;   // the fragment that opened each group (host coalescing and the multi-GPU merge keep it)
  std::unordered_map<uint64_t, std::vector<uint32_t>> groupOfHash;
  uint64_t assignedFragments = 0;
  double sumAssign = 0;
  int readLength = 0;
  // allele -> (group, slot) lists, equivalence classes
  std::vector<std::vector<std::pair<int, int>>> inAllele;
  std::vector<std::vector<int>> ecAlleles;
  std::vector<std::vector<std::pair<int, int>>> selected;  // per gene: (allele, rank)
  std::vector<double> geneAbund, majorAbund, geneMaxMajor;
  int emIterations = 0;

  size_t nGroups() const { return groupPtr.size() - 1; }
  void coalesce(t1k_row_entry *row, uint32_t n, uint32_t fragment = 0);  // CoalesceReadAssignments (841-908), one fragment
  // group tables of several owners (every pattern belongs to exactly one of them) -> one table in first-fragment order = the
  // reference's first-appearance numbering
  void setGroupsMerged(const std::vector<uint32_t> &sizes, const GroupVec &entries, const std::vector<uint32_t> &first);
  void finalize(const std::vector<int32_t> &missing);        // FinalizeReadAssignments (912-939); missing[a] from t1k_missing_coverage
  // QuantifyAlleleEquivalentClass (1142-1328); the E-step covers groups [gBegin, gEnd) (the whole table on one GPU)
  // the E-step's row pass covers the slice `rank` of `nRanks` of the read groups when a communicator is given (t1k_em_shard)
  int quantify(t1k_ctx *ctx, t1k_comm *comm, std::string &err);
  void dropUnlikely();                                       // RemoveLowLikelihoodAlleleInEquivalentClass (1371-1460)
  void select();                                             // SelectAllelesForGenes (1462-2090)
  // Optional: called once inside select(), after the per-gene candidate lists exist (1462-1695) and before the first read of an
  // allele's missingCoverage (1733-1770, 1870-1878), with the alleles on those lists; it fills ref->al[a].missingCov for them (the only
  // alleles whose value is ever read).  A job whose per-base coverage is deferred computes it here (t1k_coverage_selected).
  std::function<bool(const std::vector<int> &)> missingCoverageHook;
  bool hookFailed = false;
  std::string geneLine(int gene) const;                      // GetAlleleDescription (2103-2178) + Genotyper.cpp:660-670
  std::string alleleLines() const;                           // OutputRepresentativeAlleles (2180-2229)
  void setAbundance(const double *ecReadCount, const std::vector<int> &ecLen);  // SetAlleleAbundance (957-1014)
  int geneTypes(int gene) const;                             // GetGeneAlleleTypes (1053-1069)
};

// host/variants.cpp: novel-variant calling of the analyzer stage (VariantCaller.hpp), host code over the fragments' assignment lists and
// the edit strings of their read-ends
struct VariantRec {  // _variant (VariantCaller.hpp:7-20)
  int allele = 0, refPos = 0;
  char ref = 0, var = 0;
  double allSupport = 0, varSupport = 0, varUniqSupport = 0;
  int group = 0, outputGroup = 0, qual = 0;
};
class VariantCaller {
 public:
  struct Fragment {  // one fragment: its assignment list (SeqSet::ReadAssignmentToFragmentAssignment's order) and its read(s)
    const t1k_frag_assignment *asg = nullptr;
    uint32_t n = 0;
    const char *r1 = nullptr, *r2 = nullptr;
    uint32_t l1 = 0, l2 = 0;
  };
  // abundance[a] = Genotyper::GetAlleleAbundance(a) after the analyzer's EM (VariantCaller::SetSeqAbundance 249-265); maxGroup = --varMaxGroup
  VariantCaller(const RefSet &ref, const std::vector<double> &abundance, int maxGroup);
  ~VariantCaller();
  void compute(const std::vector<Fragment> &frags, const int8_t *ops);          // ComputeVariant (978-1140)
  std::string vcfText() const;                                                   // OutputAlleleVCF (1202-1227)
  void adjust(const Fragment &f, const int8_t *ops, uint8_t *keep) const;        // AdjustFragmentAssignment (1229-1311)
  std::vector<VariantRec> variants;                                              // finalVariants

 private:
  struct Cell;
  const RefSet &ref_;
  std::vector<double> abundance_;
  int maxGroup_;
  std::vector<size_t> base_;                 // allele -> first cell
  std::unique_ptr<std::vector<Cell>> cells_; // one per allele base
  std::vector<int> copies_;                  // seqCopy
  std::vector<std::pair<int, int>> cand_;    // candidateVariants: (allele, position)
  std::vector<std::vector<int>> candAt_;     // allele -> its candidates' positions, ascending
  std::vector<char> root_;                   // found from the counts (rootCandidate), not by expansion
  std::vector<int> group_;                   // candidateVariantGroupId
  std::vector<std::vector<std::pair<int, double>>> edges_;     // candidate -> (candidate, weight)
  std::vector<std::vector<std::pair<uint32_t, char>>> seen_;   // candidate -> (fragment, nucleotide shown)
  std::unordered_map<size_t, std::vector<int>> calledAt_;      // cell -> called variants (finalVariantIds)
  Cell &cell(int allele, int pos) const;
  void bookOverlap(const char *read, uint32_t len, const t1k_overlap &o, const int8_t *ops, uint32_t nOps, double weight, bool filter, Cell *table, const Cell *best);
  void bookFragment(const Fragment &f, const int8_t *ops, bool first, int part, int parts, Cell *table, const Cell *best);
  bool candidateIn(int allele, int from, int to) const;
  int newCandidate(int allele, int pos, bool root);
  void findRoots();
  void expandFragment(const Fragment &f, const int8_t *ops);
  void linkFragment(const Fragment &f, uint32_t fragIdx, const int8_t *ops, int part, int parts);
  void solveGroup(const std::vector<int> &vars, int groupId);
};

// the overlaps behind the assignments a fragment kept (SeqSet::ReadAssignmentToFragmentAssignment's choice per allele, SeqSet.hpp:2310-2458) from
// its read-ends' overlap lists; alleles = the fragment's row, in order.  false: an allele without a candidate in the lists
bool fragmentDetails(const t1k_overlap *l1, uint32_t n1, const t1k_overlap *l2, uint32_t n2, bool paired, const int32_t *alleles, uint32_t nAlleles, t1k_frag_assignment *out);

}  // namespace t1k
