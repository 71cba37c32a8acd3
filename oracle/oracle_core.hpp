// oracle/oracle_core.hpp
//
// TEST INFRASTRUCTURE ONLY.  CPU restatement of the T1K genotyper hot path (assignment through the EM,
// likelihood pruning, allele selection, genotype quality and the two result tables),
// written from the reference's behaviour, one function per reference routine, each citing the reference
// file:line it follows (paths relative to /root/reference).  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may use this code -- and only as the checker.  The product
// (t1k_amd/csrc) never includes, links or executes anything under oracle/.
//
// Parity status: PINNED.  The restatement is checked (tests/test_oracle_golden.py, tests/golden/) against
// outputs of the reference itself: oracle/_ref/genotyper built by oracle/Makefile from the reference's own
// sources (--outputReadAssignment rows, -DDEBUG EM trajectories, *_genotype.tsv and *_allele.tsv byte for byte,
// fixtures and live runs with the selection options), and against GlobalAlignment
// I/O vectors captured from the reference.  The reference repository has no tests of its own.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <map>

namespace t1k_oracle {

enum { OP_MATCH = 0, OP_MISMATCH = 1, OP_INSERT = 2, OP_DELETE = 3 };  // AlignAlgo.hpp:7-10

// AlignAlgo::GlobalAlignment (AlignAlgo.hpp:215-421).  ops receives the edit string (no terminator).
int globalAlignment(const char *t, int lent, const char *p, int lenp, std::vector<int8_t> &ops);

struct Posting { uint32_t allele, offset; };  // KmerIndex.hpp:12-17

struct Overlap {  // SeqSet.hpp:89-144
  int seqIdx = -1, readStart = 0, readEnd = 0, seqStart = 0, seqEnd = 0, strand = 0;
  int matchCnt = 0;
  double similarity = 0;
  int leftClip = 0, rightClip = 0, relaxedMatchCnt = 0;
};
bool overlapBefore(const Overlap &a, const Overlap &b);  // SeqSet.hpp:103-127

struct FragmentOverlap {  // SeqSet.hpp:146-173 (only the fields the reference sets and reads)
  int seqIdx = -1, seqStart = 0, seqEnd = 0, matchCnt = 0, relaxedMatchCnt = 0;
  double similarity = 0;
  bool hasMatePair = false, o1FromR2 = false, hasN = false;
  Overlap o1, o2;
  double qual = 0;
};

struct RowEntry {  // Genotyper.hpp:44-56
  int alleleIdx, start, end;
  float weight, qual, adjustWeight;
};

struct AlleleRec {
  std::string name, seq;
  int effectiveLen = 0, weight = 1;
  std::vector<int> separators;       // SeqSet.hpp:924-928 (with the -1 and len sentinels)
  std::vector<uint8_t> exon;         // _validDiff::exon, SeqSet.hpp:638-723
  std::vector<int32_t> cov;          // posWeight[pos].count[b], 4 ints per base (SeqSet.hpp:2253-2274)
  int gene = -1, majorAllele = -1;
  int missingCoverage = 0;
  int ec = -1;
  double abundance = 0, ecAbundance = 0;
  int genotypeQuality = -1, alleleRank = -1;  // Genotyper.hpp:590-593
};

struct Params {
  int k = 11, radius = 10, hitLenRequired = 31;
  double refSeqSimilarity = 0.8;  // -s
  bool relaxIntronAlign = false;
  int maxAssignCnt = 2000;        // -n
  double filterFrac = 0.15;
  int alleleDigitUnits = -1;
  char alleleDelimiter = 0;
  double minSquaremAlpha = 0;
  double filterCov = 1.0, crossGeneRate = 0.04;  // --cov, --crossGeneRate (Genotyper.cpp:223-224)
  int nBaseCode = 3;  // bits an N contributes to a k-mer code: nucToNum['N'] & 3 = 3 in Genotyper.cpp:37-40, 0 in FastqExtractor.cpp:51-54
};

struct Stats {  // algorithmic-traffic counters for SURVEY 8d's roofline formula
  uint64_t readEnds = 0, lookups = 0, postings = 0, candidates = 0, extended = 0, nearBest = 0, gaCalls = 0, gaCells = 0;
};

class Oracle {
 public:
  Params prm;
  std::vector<AlleleRec> alleles;
  bool rnaData = true;
  std::vector<std::string> geneNames, majorNames;
  Stats stats;

  // ---- reference side (Genotyper::InitRefSet Genotyper.hpp:707-730, SeqSet::InputRefSeq SeqSet.hpp:906-982) ----
  int loadReference(const std::string &fasta);
  void addAllele(const std::string &name, const std::string &comment, const std::string &seq, bool hasComment);
  void finishReference();  // UpdateDnaSeqWeight + InitAlleleInfo (gene ids, effective-length fix) + index

  // ---- read-end side ----
  // SeqSet::AssignRead (SeqSet.hpp:2119-2303) with barcode=-1
  int assignRead(const std::string &read, int weight, std::vector<Overlap> &out);
  // SeqSet::ReadAssignmentToFragmentAssignment (SeqSet.hpp:2310-2655); o2 may be NULL (single-end)
  int pairFragments(const std::vector<Overlap> &o1, const std::vector<Overlap> *o2, bool hasN, std::vector<FragmentOverlap> &out);
  // Genotyper::SetReadAssignments (Genotyper.hpp:778-832)
  void fragmentToRow(const std::vector<FragmentOverlap> &frag, std::vector<RowEntry> &row);

  // ---- group side ----
  std::vector<std::vector<RowEntry>> groups;  // coalesced read groups ("readAssignments", Genotyper.hpp:443)
  std::map<std::vector<int>, int> groupOfPattern;
  void coalesceRow(std::vector<RowEntry> &row);  // Genotyper::CoalesceReadAssignments body (841-908)
  std::vector<std::vector<int>> ecAlleles;       // equivalentClassToAlleles
  void finalizeGroups();                         // FinalizeReadAssignments (912-939)
  int quantify(std::vector<double> *trajectory = nullptr);  // QuantifyAlleleEquivalentClass (1142-1328); returns #iterations
  std::vector<double> ecReadCountFinal, ecAbundanceFinal;
  std::vector<int> ecLength;
  // ---- after the EM: pruning, selection, the two tables ----
  std::vector<double> majorAlleleAbundance, geneMaxMajorAlleleAbundance;  // as SetAlleleAbundance left them (1012-1050)
  std::vector<std::vector<double>> geneSimilarity;                        // InitAlleleInfo (597-638)
  int readLength = 0;                                                     // SetReadLength(maxReadLength), Genotyper.cpp:443
  std::vector<std::vector<std::pair<int, int>>> selectedAlleles;          // per gene: (allele, rank)
  void computeGeneSimilarity();
  void removeLowLikelihoodAlleles();  // RemoveLowLikelihoodAlleleInEquivalentClass (1371-1460)
  void selectAllelesForGenes();       // SelectAllelesForGenes (1462-2090)
  int geneAlleleTypes(int gene) const;  // GetGeneAlleleTypes (1053-1068)
  std::string genotypeText() const;   // GetAlleleDescription (2103-2178) + the loop of Genotyper.cpp:660-670
  std::string alleleText() const;     // OutputRepresentativeAlleles (2180-2229)
  void parseAlleleNameExon(const std::string &allele, std::string &gene, std::string &major) const;  // ParseAlleleName with fieldsType = 1

  // ---- candidate extraction (oracle_extract.cpp) ----
  int loadReferenceFa(const std::string &fasta);   // SeqSet::InputRefFa (SeqSet.hpp:872-904): one sequence per record, no merging
  int inferKmerLength() const;                     // SeqSet::InferKmerLength (SeqSet.hpp:2830-2845)
  void setKmerLength(int k);                       // SeqSet::UpdateKmerLength (SeqSet.hpp:2847-2858)
  bool hasHitInSet(const std::string &read);       // SeqSet::HasHitInSet (SeqSet.hpp:1915-1990)
  static bool isLowComplexityRead(const std::string &read);  // IsLowComplexity (FastqExtractor.cpp:89-111)
  bool isGoodCandidate(const std::string &read) { return !isLowComplexityRead(read) && hasHitInSet(read); }  // FastqExtractor.cpp:113-118

  // helpers exposed for unit tests
  void seedHits(const std::string &read, std::vector<int> &strand, std::vector<int> &readOff, std::vector<Posting> &post);
  bool separatorInRange(int s, int e, int seqIdx) const;  // SeqSet.hpp:487-498
  int missingBaseCoverage(int seqIdx, double ratio) const;  // SeqSet.hpp:2717-2755
  void parseAlleleName(const std::string &allele, std::string &gene, std::string &major) const;  // Genotyper.hpp:63-131

 private:
  std::vector<uint32_t> idxStart;   // 4^k + 1
  std::vector<Posting> idxPost;
  std::vector<std::vector<int>> groupsInAllele;  // readsInAllele (group ids), with slot
  std::vector<std::vector<int>> slotInAllele;
  struct Cand { Overlap o; std::vector<std::pair<int, int>> chain; };
  void buildIndex();
  void candidatesFromHits(const std::vector<int> &strand, const std::vector<int> &readOff, const std::vector<Posting> &post,
                          std::vector<Cand> &cands);
  int overlapsFromRead(const std::string &read, const std::string &rc, std::vector<Overlap> &out);
  bool extendOverlap(const std::string &r, const Overlap &o, Overlap &e);
  double emUpdate(const std::vector<double> &x0, std::vector<double> &x1, std::vector<double> &n,
                  const std::vector<std::vector<int>> &rows, const std::vector<double> &count);
  void setAlleleAbundance(const std::vector<double> &n, std::vector<double> &majorAbund, std::vector<double> &geneMax);
};

// FASTA/FASTQ(+gz) record reader with the reference's id/comment conventions (ReadFiles.hpp:155-204, kseq.h)
struct SeqRecord { std::string id, rawId, comment, seq, qual; bool hasComment = false; };  // rawId: before the /1 /2 strip
bool readAllRecords(const std::string &path, std::vector<SeqRecord> &out);

std::string reverseComplement(const std::string &s);  // SeqSet.hpp:2103-2114

}  // namespace t1k_oracle
