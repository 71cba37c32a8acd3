// oracle/oracle_capi.cpp -- TEST INFRASTRUCTURE ONLY.  Thin C ABI over the CPU restatement for ctypes-based tests.
#include <cstring>
#include "oracle_core.hpp"

using namespace t1k_oracle;

extern "C" {

// AlignAlgo::GlobalAlignment restatement.  ops must hold lent+lenp+2 bytes.  Returns the score, *nops the edit-string length.
int orc_global_alignment(const char *t, int lent, const char *p, int lenp, signed char *ops, int *nops) {
  std::vector<int8_t> v;
  int s = globalAlignment(t, lent, p, lenp, v);
  memcpy(ops, v.data(), v.size());
  *nops = (int)v.size();
  return s;
}

void *orc_create(double similarity, int relaxIntron, int maxAssign, int digitUnits, char delimiter) {
  Oracle *o = new Oracle();
  o->prm.refSeqSimilarity = similarity;
  o->prm.relaxIntronAlign = relaxIntron != 0;
  o->prm.maxAssignCnt = maxAssign;
  o->prm.alleleDigitUnits = digitUnits;
  o->prm.alleleDelimiter = delimiter;
  return o;
}
void orc_destroy(void *h) { delete (Oracle *)h; }
int orc_load_reference(void *h, const char *fasta) { return ((Oracle *)h)->loadReference(fasta); }
int orc_allele_count(void *h) { return (int)((Oracle *)h)->alleles.size(); }
const char *orc_allele_name(void *h, int i) { return ((Oracle *)h)->alleles[i].name.c_str(); }

// SeqSet::AssignRead restatement.  out: 12 int32 per overlap
// (seqIdx, readStart, readEnd, seqStart, seqEnd, strand, matchCnt, leftClip, rightClip, relaxedMatchCnt, simNumer, simDenom);
// sim: similarity doubles.  Returns the number of overlaps (<= cap written).
int orc_assign_read(void *h, const char *read, int weight, int *out, double *sim, int cap) {
  std::vector<Overlap> ov;
  ((Oracle *)h)->assignRead(read, weight, ov);
  int n = 0;
  for (auto &o : ov) {
    if (n >= cap) break;
    int *r = out + 12 * n;
    r[0] = o.seqIdx; r[1] = o.readStart; r[2] = o.readEnd; r[3] = o.seqStart; r[4] = o.seqEnd; r[5] = o.strand;
    r[6] = o.matchCnt; r[7] = o.leftClip; r[8] = o.rightClip; r[9] = o.relaxedMatchCnt; r[10] = 0; r[11] = 0;
    sim[n] = o.similarity;
    ++n;
  }
  return (int)ov.size();
}

// per-base coverage of the allele's own base (the only counter GetSeqMissingBaseCoverage reads, SeqSet.hpp:2729)
int orc_coverage(void *h, int allele, int *out, int cap) {
  Oracle *o = (Oracle *)h;
  const AlleleRec &a = o->alleles[allele];
  int L = (int)a.seq.size();
  for (int i = 0; i < L && i < cap; ++i) {
    int b = a.seq[i] == 'A' ? 0 : a.seq[i] == 'C' ? 1 : a.seq[i] == 'G' ? 2 : a.seq[i] == 'T' ? 3 : -1;
    out[i] = b >= 0 ? a.cov[(size_t)i * 4 + b] : 0;
  }
  return L;
}

// ---- candidate extraction (oracle_extract.cpp) ----
int orc_load_reference_fa(void *h, const char *fasta) { return ((Oracle *)h)->loadReferenceFa(fasta); }
int orc_infer_kmer_length(void *h) { return ((Oracle *)h)->inferKmerLength(); }
void orc_set_extract_params(void *h, int k, int hitLenRequired) {
  Oracle *o = (Oracle *)h;
  if (k != o->prm.k) o->setKmerLength(k);
  o->prm.hitLenRequired = hitLenRequired;
}
int orc_is_low_complexity(const char *read) { return Oracle::isLowComplexityRead(read) ? 1 : 0; }
int orc_has_hit_in_set(void *h, const char *read) { return ((Oracle *)h)->hasHitInSet(read) ? 1 : 0; }
int orc_is_good_candidate(void *h, const char *read) { return ((Oracle *)h)->isGoodCandidate(read) ? 1 : 0; }
}
