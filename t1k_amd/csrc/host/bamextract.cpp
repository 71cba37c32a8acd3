// t1k_amd/csrc/host/bamextract.cpp -- candidate-read extraction from a coordinate-sorted BAM file: the argv-compatible
// replacement of the reference's bam-extractor (BamExtractor.cpp:463-949, started by run-t1k:350).
//
// What the reference does: it reads the gene coordinates from the headers of the "coordinate FASTA" (">gene chrom start end strand"), walks
// the BAM once and keeps (a) every aligned read that overlaps a gene interval and is not low-complexity (BamExtractor.cpp:836-880),
// (b) reads aligned to alternative contigs and unaligned reads that pass IsLowComplexity + SeqSet::HasHitInSet against the gene
// sequences (646-775; unaligned pairs come as two consecutive records and are written at once), then -- paired data -- walks the BAM a
// second time to collect both mates of every kept template name (895-938).  Output: <prefix>_1.fq / _2.fq (or <prefix>.fq), optionally
// _bc.fa / _umi.fa from BAM tags.
//
// Here: the BAM container is read natively (BGZF blocks inflated in parallel by the host threads, records parsed in place -- no
// samtools), and every HasHitInSet question of a batch of records is answered by the GPU stage the fastq-extractor uses
// (t1k_extract_batch, csrc/t1k_extract.hip); the decisions are then replayed in file order, so the output equals the reference's
// single-thread run byte for byte (its -t > 1 mode writes the unaligned pairs in thread-completion order).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include "t1k_host.h"

#define T1K_BAM_MAX_READ 1000  // t1k_params_default's max_read_len (T1K_LONG_READ_LEN: a batch with reads beyond 320 bases takes the wide kernel shapes)

namespace {

void logLine(const char *msg) {  // PrintLog (BamExtractor.cpp:130-142)
  time_t t = time(nullptr);
  char stime[500];
  strftime(stime, sizeof(stime), "%c", localtime(&t));
  fprintf(stderr, "[%s] %s\n", stime, msg);
}

const char kUsage[] =
    "./bam-extractor [OPTIONS]:\n"
    "Required:\n"
    "\t-f STRING: fasta file containing the reference sequence\n"
    "\t-b STRING: path to BAM file\n"
    "Optional:\n"
    "\t-o STRING: prefix to the output file\n"
    "\t-t INT: number of threads (default: 1)\n"
    "\t-u: the flag or order of unaligned read-pair is not ordinary (default: not used)\n"
    "\t--barcode STRING: the barcode field in the bam file (default: not used)\n"
    "\t--UMI STRING: the UMI field in the bam file (default: not used)\n"
    "\t--mateIdSuffixLen INT: the suffix length in read id for mate. (default: not used)\n"
    "(this build: reads of up to 1000 bases; a longer read on an alternative contig / unaligned is not tested and not kept, with a warning)\n";

// ---- BGZF: the file is a series of gzip members of at most 64 KiB of data each, their compressed size in a "BC" extra field ----------
struct BamFile {
  int fd = -1;
  const uint8_t *map = nullptr;
  size_t size = 0, pos = 0;          // next compressed byte
  std::vector<uint8_t> buf;          // inflated bytes not consumed yet: [bufPos, buf.size())
  size_t bufPos = 0;
  int threads = 8;
  std::string err;
  // header
  std::vector<std::string> refName;
  std::unordered_map<std::string, int> refId;
  bool open(const std::string &path) {
    fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) { err = "Can not open " + path + "."; return false; }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 28) { err = "Can not open " + path + "."; return false; }
    size = (size_t)st.st_size;
    map = (const uint8_t *)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map == MAP_FAILED) { map = nullptr; err = "Can not map " + path + "."; return false; }
    (void)madvise((void *)map, size, MADV_SEQUENTIAL);
    return rewind();
  }
  ~BamFile() { close(); }
  void close() {
    if (map) munmap((void *)map, size);
    if (fd >= 0) ::close(fd);
    map = nullptr; fd = -1;
  }
  // inflate the next blocks (about `want` bytes of output) behind the unconsumed tail of the buffer; false at the end of the file
  bool refill(size_t want = 32u << 20) {
    if (bufPos) { buf.erase(buf.begin(), buf.begin() + (long)bufPos); bufPos = 0; }
    struct Blk { size_t in, inLen, out, outLen; };
    std::vector<Blk> blks;
    size_t out = buf.size(), got = 0;
    while (pos < size && got < want) {
      if (size - pos < 28 || map[pos] != 31 || map[pos + 1] != 139 || map[pos + 2] != 8 || !(map[pos + 3] & 4)) { err = "not a BGZF (BAM) file"; return false; }
      const size_t xlen = map[pos + 10] | (map[pos + 11] << 8);
      size_t bsize = 0;
      if (pos + 12 + xlen > size) { err = "damaged BGZF block"; return false; }  // (the extra field itself must lie inside the file)
      for (size_t x = pos + 12; x + 4 <= pos + 12 + xlen;) {
        const size_t slen = map[x + 2] | (map[x + 3] << 8);
        if (x + 4 + slen > pos + 12 + xlen) break;  // a subfield that runs past the extra field: damaged, no BC found -> reported below
        if (map[x] == 'B' && map[x + 1] == 'C' && slen == 2) bsize = (size_t)(map[x + 4] | (map[x + 5] << 8)) + 1;
        x += 4 + slen;
      }
      if (!bsize || pos + bsize > size || bsize < xlen + 20) { err = "damaged BGZF block"; return false; }
      const uint8_t *tail = map + pos + bsize - 4;
      const size_t isize = (size_t)tail[0] | ((size_t)tail[1] << 8) | ((size_t)tail[2] << 16) | ((size_t)tail[3] << 24);
      if (isize > 65536) { err = "damaged BGZF block"; return false; }  // (a BGZF block holds at most 64 KiB of data; the field is 32 bits wide)
      blks.push_back({pos + 12 + xlen, bsize - xlen - 20, out, isize});
      out += isize; got += isize; pos += bsize;
    }
    if (blks.empty()) return false;
    buf.resize(out);
    std::atomic<size_t> next{0};
    std::atomic<bool> bad{false};
    auto work = [&] {
      for (size_t i = next.fetch_add(1); i < blks.size(); i = next.fetch_add(1)) {
        if (!blks[i].outLen) continue;
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -15) != Z_OK) { bad = true; continue; }
        zs.next_in = (Bytef *)(map + blks[i].in); zs.avail_in = (uInt)blks[i].inLen;
        zs.next_out = buf.data() + blks[i].out; zs.avail_out = (uInt)blks[i].outLen;
        const int r = inflate(&zs, Z_FINISH);
        if (r != Z_STREAM_END || zs.avail_out != 0) bad = true;
        inflateEnd(&zs);
      }
    };
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, blks.size() / 4 + 1));
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
    if (bad) { err = "damaged BGZF block (inflate failed)"; return false; }
    return true;
  }
  // n bytes of the inflated stream, contiguous; nullptr at the end of the file (or on error: err is set)
  const uint8_t *take(size_t n) {
    while (buf.size() - bufPos < n)
      if (!refill()) return nullptr;
    const uint8_t *p = buf.data() + bufPos;
    bufPos += n;
    return p;
  }
  static uint32_t u32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
  bool rewind() {  // (Alignments::Rewind reopens the file, alignments.hpp:129-136)
    pos = 0; buf.clear(); bufPos = 0; err.clear();
    const uint8_t *p = take(8);
    if (!p || memcmp(p, "BAM\1", 4) != 0) { if (err.empty()) err = "not a BAM file"; return false; }
    const uint32_t lText = u32(p + 4);
    if (!take(lText)) return false;
    p = take(4);
    if (!p) return false;
    const uint32_t nRef = u32(p);
    refName.clear(); refId.clear();
    for (uint32_t i = 0; i < nRef; ++i) {
      p = take(4);
      if (!p) return false;
      const uint32_t l = u32(p);
      p = take((size_t)l + 4);
      if (!p) return false;
      std::string nm((const char *)p, l ? l - 1 : 0);
      refId[nm] = (int)i;  // (the last of equal names wins, as in the reference's map)
      refName.push_back(nm);
    }
    return true;
  }
};

// one alignment record, viewed in the reader's buffer (valid until the next record is taken)
struct Rec {
  const uint8_t *d = nullptr;
  uint32_t len = 0;
  int32_t tid() const { return (int32_t)BamFile::u32(d); }
  int32_t pos() const { return (int32_t)BamFile::u32(d + 4); }
  uint32_t lName() const { return d[8]; }
  uint32_t nCigar() const { return d[12] | (d[13] << 8); }
  uint32_t flag() const { return d[14] | (d[15] << 8); }
  int32_t lSeq() const { return (int32_t)BamFile::u32(d + 16); }
  int32_t mtid() const { return (int32_t)BamFile::u32(d + 20); }
  int32_t mpos() const { return (int32_t)BamFile::u32(d + 24); }
  const char *name() const { return (const char *)(d + 32); }
  const uint8_t *cigar() const { return d + 32 + lName(); }
  const uint8_t *seq() const { return cigar() + 4 * nCigar(); }
  const uint8_t *qual() const { return seq() + (lSeq() + 1) / 2; }
  const uint8_t *aux() const { return qual() + lSeq(); }
  bool reverse() const { return flag() & 0x10; }
  bool mateReverse() const { return flag() & 0x20; }
  bool firstMate() const { return flag() & 0x40; }
  bool primary() const { return (flag() & 0x900) == 0; }
  bool aligned() const { return !(flag() & 0x4) && tid() >= 0; }                                                       // alignments.hpp:434-439
  bool templateAligned() const { const uint32_t f = flag(); return !((f & 0xd) == 0xd || (f & 0x5) == 0x4 || tid() < 0); }  // 426-432
  // GetReadSeq / GetQual (alignments.hpp:521-575): the read as sequenced (reverse-complemented back when it aligned to the minus strand)
  void readSeq(std::string &s) const {
    const int n = lSeq();
    s.resize((size_t)n);
    const uint8_t *q = seq();
    if (!reverse()) {
      for (int i = 0; i < n; ++i) { const int b = (q[i >> 1] >> ((~i & 1) << 2)) & 0xf; s[i] = b == 1 ? 'A' : b == 2 ? 'C' : b == 4 ? 'G' : b == 8 ? 'T' : 'N'; }
    } else {
      for (int i = 0, j = n - 1; j >= 0; ++i, --j) { const int b = (q[j >> 1] >> ((~j & 1) << 2)) & 0xf; s[i] = b == 1 ? 'T' : b == 2 ? 'G' : b == 4 ? 'C' : b == 8 ? 'A' : 'N'; }
    }
  }
  void readQual(std::string &s) const {
    const int n = lSeq();
    s.resize((size_t)n);
    const uint8_t *q = qual();
    if (!reverse()) for (int i = 0; i < n; ++i) s[i] = (char)(q[i] + 33);
    else for (int i = 0, j = n - 1; j >= 0; ++i, --j) s[i] = (char)(q[j] + 33);
  }
  // first / last reference position of the aligned segments (Alignments::Next, alignments.hpp:226-283: M, D, =, X extend a segment, N closes it)
  // (64-bit sums: the 28-bit lengths of a damaged record must not overflow; a valid file's values fit an int as in the reference)
  void span(int64_t &start, int64_t &end) const {
    int64_t st = pos(), ln = 0, first = 0, last = -1;
    bool any = false;
    const uint8_t *c = cigar();
    for (uint32_t i = 0; i < nCigar(); ++i) {
      const uint32_t v = BamFile::u32(c + 4 * i);
      const int op = (int)(v & 0xf);
      const int64_t num = (int64_t)(v >> 4);
      if (op == 0 || op == 2 || op == 7 || op == 8) ln += num;
      else if (op == 3) { if (!any) first = st; any = true; last = st + ln - 1; st = st + ln + num; ln = 0; }
    }
    if (ln > 0) { if (!any) first = st; any = true; last = st + ln - 1; }
    if (!any) { first = pos(); last = (int64_t)pos() - 1; }
    start = first; end = last;
  }
  // bam_aux_get + bam_aux2Z (Alignments::GetFieldZ, alignments.hpp:490-497): the value of a Z (or H) tag, NULL if absent / of another type
  const char *fieldZ(const char *tag) const {
    const uint8_t *p = aux(), *e = d + len;
    while (p + 3 <= e) {
      const bool hit = p[0] == (uint8_t)tag[0] && p[1] == (uint8_t)tag[1];
      const char ty = (char)p[2];
      p += 3;
      if (ty == 'Z' || ty == 'H') {
        const char *s = (const char *)p;
        while (p < e && *p) ++p;
        if (p >= e) return nullptr;  // no terminator inside the record: damaged, the tag counts as absent (never read past the record)
        ++p;
        if (hit) return s;
        continue;
      }
      if (hit) return nullptr;
      const size_t left = (size_t)(e - p);
      if (ty == 'A' || ty == 'c' || ty == 'C') p += std::min<size_t>(1, left);
      else if (ty == 's' || ty == 'S') p += std::min<size_t>(2, left);
      else if (ty == 'i' || ty == 'I' || ty == 'f') p += std::min<size_t>(4, left);
      else if (ty == 'd') p += std::min<size_t>(8, left);
      else if (ty == 'B') {
        if (p + 5 > e) return nullptr;
        const char sub = (char)p[0];
        const uint32_t n = BamFile::u32(p + 1);
        const size_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
        if ((uint64_t)w * n > (uint64_t)(e - p) - 5) return nullptr;  // the array runs past the record
        p += 5 + w * n;
      } else return nullptr;
    }
    return nullptr;
  }
};

bool nextRecord(BamFile &bam, Rec &r) {
  const uint8_t *p = bam.take(4);
  if (!p) return false;
  const uint32_t n = BamFile::u32(p);
  if (n < 32 || n > (1u << 30)) { bam.err = "damaged BAM record"; return false; }  // (a length word of a damaged file must not make the reader buffer the rest of the file)
  p = bam.take(n);
  if (!p) { if (bam.err.empty()) bam.err = "truncated BAM file"; return false; }
  r.d = p; r.len = n;
  // the lengths inside the record must fit the record: name, CIGAR, 4-bit sequence and qualities are read through them
  const int64_t lSeq = r.lSeq();
  if (lSeq < 0 || r.lName() < 1 || 32 + (uint64_t)r.lName() + 4ull * r.nCigar() + ((uint64_t)lSeq + 1) / 2 + (uint64_t)lSeq > (uint64_t)n) { bam.err = "damaged BAM record"; return false; }
  // the name is read as a C string and the contig number indexes the header's table
  if (r.d[32 + r.lName() - 1] != 0 || r.tid() < -1 || r.tid() >= (int64_t)bam.refName.size()) { bam.err = "damaged BAM record"; return false; }
  return true;
}

bool isLowComplexity(const std::string &s) {  // BamExtractor.cpp:144-168
  int cnt[5] = {0, 0, 0, 0, 0};
  for (char c : s) ++cnt[c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4];
  const int n = (int)s.size();
  if (cnt[0] >= n / 2 || cnt[1] >= n / 2 || cnt[2] >= n / 2 || cnt[3] >= n / 2 || cnt[4] >= n / 10) return true;
  int low = 0;
  for (int i = 0; i < 4; ++i) if (cnt[i] <= 2) ++low;
  return low >= 2;
}

bool validAlternativeChrom(const std::string &c) { return c.find('_') != std::string::npos || c.find('.') != std::string::npos || c.find('*') != std::string::npos; }  // 120-128

void trimName(std::string &name, int trimLen) {  // 170-185
  const size_t len = name.size();
  if (trimLen == -1) {
    if (len >= 2 && (name[len - 1] == '1' || name[len - 1] == '2') && name[len - 2] == '/') name.erase(len - 2, 2);
  } else if ((size_t)trimLen <= len) name.erase(len - (size_t)trimLen, (size_t)trimLen);
}

struct Interval { int chr, start, end; bool operator<(const Interval &o) const { return chr != o.chr ? chr < o.chr : start != o.start ? start < o.start : end < o.end; } };

}  // namespace

extern "C" int t1k_bam_extractor_main(int argc, char **argv) {
  if (argc <= 1) { fprintf(stderr, "%s", kUsage); return 0; }
  std::string refPath, bamPath, prefix = "toassemble", bcField, umiField;
  bool abnormalUnaligned = false;
  int threadCnt = 1, mateIdLen = -1;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto val = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
    if (a == "-f") refPath = val();
    else if (a == "-b") bamPath = val();
    else if (a == "-o") prefix = val();
    else if (a == "-u") abnormalUnaligned = true;
    else if (a == "-t") threadCnt = atoi(val());
    else if (a == "--barcode") bcField = val();
    else if (a == "--UMI") umiField = val();
    else if (a == "--mateIdSuffixLen") mateIdLen = atoi(val());
    else { fprintf(stderr, "Unknown parameter %s\n", a.c_str()); return EXIT_FAILURE; }
  }
  if (refPath.empty()) { fprintf(stderr, "Need to use -f to specify the reference sequence.\n"); return EXIT_FAILURE; }
  if (bamPath.empty()) { fprintf(stderr, "Need to use -b to specify the BAM file.\n"); return EXIT_FAILURE; }
  (void)threadCnt;  // the order of the output is that of the reference's single-thread run whatever -t says
  BamFile bam;
  {
    const int hw = (int)std::thread::hardware_concurrency();
    bam.threads = std::max(1, std::min(hw > 0 ? hw : 8, 32));
    if (const char *e = getenv("T1K_HOST_THREADS")) bam.threads = std::max(1, atoi(e));
  }
  if (!bam.open(bamPath)) { fprintf(stderr, "%s\n", bam.err.empty() ? "Can not read the BAM header." : bam.err.c_str()); return EXIT_FAILURE; }
  // gene sequences (SeqSet::InputRefFa) and their coordinates from the header lines: ">gene chrom start end strand" (BamExtractor.cpp:548-562)
  std::vector<t1k::SeqRec> ref;
  std::string err;
  if (!t1k::readSeqFile(refPath, ref, err) || ref.empty()) { fprintf(stderr, "%s\n", err.empty() ? "empty reference" : err.c_str()); return EXIT_FAILURE; }
  logLine("Start to extract candidate reads from bam file.");
  auto chromId = [&](const std::string &s, int &id) {  // Alignments::GetChromIdFromName (alignments.hpp:321-341)
    auto it = bam.refId.find(s);
    if (it != bam.refId.end()) { id = it->second; return true; }
    if (s.size() >= 4 && (it = bam.refId.find(s.substr(3))) != bam.refId.end()) { id = it->second; return true; }
    if ((it = bam.refId.find("chr" + s)) != bam.refId.end()) { id = it->second; return true; }
    return false;
  };
  std::vector<Interval> genes;
  for (auto &r : ref) {
    char chrom[1024];
    int start = 0, end = 0;
    if (sscanf(r.comment.c_str(), "%1023s %d %d", chrom, &start, &end) != 3) { fprintf(stderr, "bam-extractor: the header of %s does not hold \"chrom start end strand\".\n", r.id.c_str()); return EXIT_FAILURE; }
    Interval iv;
    if (!chromId(chrom, iv.chr)) { printf("Unknown genome name: %s\n", chrom); return 1; }
    iv.start = start; iv.end = end;
    genes.push_back(iv);
  }
  std::sort(genes.begin(), genes.end());
  const int geneCnt = (int)genes.size();

  // Alignments::GetGeneralInfo(true) (alignments.hpp:594-690): read length and "is this paired data" from the first 100 000 primary records
  int readLen = 0;
  bool paired = false;
  {
    Rec r;
    long total = 0, hasMate = 0;
    while (nextRecord(bam, r)) {
      if (!r.primary()) continue;
      readLen = std::max(readLen, r.lSeq());
      if (r.flag() & 0x1) ++hasMate;
      if (++total >= 100000) break;
    }
    if (!bam.err.empty()) { fprintf(stderr, "bam-extractor: %s\n", bam.err.c_str()); return EXIT_FAILURE; }
    if (total == 0) { fprintf(stderr, "bam-extractor: the BAM file holds no reads.\n"); return EXIT_FAILURE; }
    paired = hasMate >= total / 2;  // (fragStdev != 0 in the reference: it is forced to >= 1 for paired data)
    if (!bam.rewind()) { fprintf(stderr, "bam-extractor: %s\n", bam.err.c_str()); return EXIT_FAILURE; }
  }
  int hitLenRequired = paired ? 21 : 17;  // BamExtractor.cpp:570-575
  if (readLen / 5 > hitLenRequired) hitLenRequired = readLen / 5;
  int kmerLength = 9;
  {
    int total = 0;  // SeqSet::InferKmerLength (SeqSet.hpp:2830-2845)
    for (auto &r : ref) total += (int)r.seq.size();
    int ret = 0;
    while (total) { ++ret; total /= 4; }
    ++ret;
    if (ret > kmerLength) { kmerLength = ret; if (kmerLength > hitLenRequired) hitLenRequired = kmerLength; }
  }
  // The device context (gene sequences packed and indexed on the GPU) is made when the first batch of reads needs HasHitInSet: a BAM
  // file whose reads all align to the primary assembly is handled by the host alone.  There is no CPU path for the test itself.
  t1k_ctx *ctx = nullptr;
  if (readLen > T1K_BAM_MAX_READ) { fprintf(stderr, "bam-extractor: a read of %d bases is longer than this build handles (%d)\n", readLen, T1K_BAM_MAX_READ); return EXIT_FAILURE; }
  auto ensureCtx = [&]() -> bool {
    if (ctx) return true;
    if (t1k_device_count() <= 0) { fprintf(stderr, "bam-extractor: no HIP device (this build has no CPU path)\n"); return false; }
    t1k_params prm;
    t1k_params_default(&prm);
    prm.kmer_length = kmerLength;
    prm.hit_len_required = hitLenRequired;
    prm.ref_seq_similarity = 0.8;  // (SeqSet's default: the program has no -s)
    prm.n_base_code = 0;           // nucToNum maps 'N' to 0 here (BamExtractor.cpp:42-45)
    const int device = getenv("T1K_DEVICE") ? atoi(getenv("T1K_DEVICE")) : 0;
    if (t1k_ctx_create(device, &prm, &ctx) != T1K_OK) { fprintf(stderr, "bam-extractor: cannot create the device context (k = %d)\n", kmerLength); ctx = nullptr; return false; }
    std::string cat;
    std::vector<uint64_t> off(ref.size() + 1, 0);
    for (size_t i = 0; i < ref.size(); ++i) { cat += ref[i].seq; off[i + 1] = cat.size(); }
    if (t1k_ref_upload(ctx, cat.data(), off.data(), nullptr, (uint32_t)ref.size()) != T1K_OK) { fprintf(stderr, "bam-extractor: %s\n", t1k_last_error(ctx)); return false; }
    return true;
  };
  const std::string p1 = prefix + (paired ? "_1.fq" : ".fq"), p2 = prefix + "_2.fq", pBc = prefix + "_bc.fa", pUmi = prefix + "_umi.fa";
  FILE *fp1 = fopen(p1.c_str(), "w"), *fp2 = paired ? fopen(p2.c_str(), "w") : nullptr;
  FILE *fpBc = bcField.empty() ? nullptr : fopen(pBc.c_str(), "w"), *fpUmi = umiField.empty() ? nullptr : fopen(pUmi.c_str(), "w");
  auto closeAll = [&](bool removeFiles) {
    if (fp1) fclose(fp1);
    if (fp2) fclose(fp2);
    if (fpBc) fclose(fpBc);
    if (fpUmi) fclose(fpUmi);
    if (removeFiles) { remove(p1.c_str()); if (paired) remove(p2.c_str()); if (!bcField.empty()) remove(pBc.c_str()); if (!umiField.empty()) remove(pUmi.c_str()); }
    if (ctx) t1k_ctx_destroy(ctx);
    bam.close();
  };
  if (!fp1 || (paired && !fp2) || (!bcField.empty() && !fpBc) || (!umiField.empty() && !fpUmi)) { fprintf(stderr, "Cannot open the output files.\n"); closeAll(true); return EXIT_FAILURE; }
  auto outSeq = [](FILE *fp, const std::string &name, const std::string &seq, const std::string &qual) { fprintf(fp, "@%s\n%s\n+\n%s\n", name.c_str(), seq.c_str(), qual.c_str()); };
  auto outTag = [](FILE *fp, const std::string &name, bool has, const std::string &v) { if (has) fprintf(fp, ">%s\n%s\n", name.c_str(), v.c_str()); else fprintf(fp, ">%s\nmissing_barcode\n", name.c_str()); };

  // ---- first pass.  Records are turned into events in file order; the reads whose fate hangs on HasHitInSet go to the GPU in batches,
  // then the events of the batch are replayed in order with the answers in hand.
  enum Kind { UnalignedPair, TestToCandidates, SingleTest, SingleGene };
  struct Event {
    Kind kind;
    std::string name, seq, qual, seq2, qual2, bc, umi;
    bool hasBc = false, hasUmi = false, aligned = false;
    int test = -1, test2 = -1;  // read-ends of the GPU batch
  };
  std::vector<Event> events;
  std::string batchSeq;
  std::vector<uint64_t> batchOff{0};
  std::unordered_set<std::string> candidates;          // paired data: template names to collect in the second pass
  std::unordered_set<std::string> usedName;            // single-end data: aligned reads already written
  uint64_t nTested = 0, nKept = 0;
  // a read beyond the device stage's length limit is asked about as an EMPTY sequence (never a candidate) and counted: the run goes on
  // (the reference has no limit and would test it; a read-length estimate above the limit was refused before the first record)
  uint64_t nOverLong = 0;
  auto addTest = [&](const std::string &s) {
    if (s.size() > (size_t)T1K_BAM_MAX_READ) { ++nOverLong; batchOff.push_back(batchSeq.size()); return (int)batchOff.size() - 2; }
    batchSeq += s; batchOff.push_back(batchSeq.size()); return (int)batchOff.size() - 2;
  };
  auto tags = [&](const Rec &r, Event &e) {
    if (!bcField.empty()) { const char *v = r.fieldZ(bcField.c_str()); e.hasBc = v != nullptr; if (v) e.bc = v; }
    if (!umiField.empty()) { const char *v = r.fieldZ(umiField.c_str()); e.hasUmi = v != nullptr; if (v) e.umi = v; }
  };
  std::vector<uint8_t> good;
  auto flush = [&]() -> bool {
    const uint32_t n = (uint32_t)batchOff.size() - 1;
    good.assign(n, 0);
    if (n) {
      if (!ensureCtx()) return false;
      if (t1k_reads_upload(ctx, batchSeq.data(), batchOff.data(), nullptr, n) != T1K_OK || t1k_extract_batch(ctx, 1, good.data(), nullptr) != T1K_OK) {
        fprintf(stderr, "bam-extractor: %s\n", t1k_last_error(ctx));
        return false;
      }
      nTested += n;
    }
    for (Event &e : events) {
      switch (e.kind) {
        case UnalignedPair:  // BamExtractor.cpp:686-708: neither mate low-complexity, and one of them hits (good = not low-complexity and a hit)
          if (!isLowComplexity(e.seq) && !isLowComplexity(e.seq2) && (good[e.test] || good[e.test2])) {
            outSeq(fp1, e.name, e.seq, e.qual);
            outSeq(fp2, e.name, e.seq2, e.qual2);
            if (fpBc) outTag(fpBc, e.name, e.hasBc, e.bc);
            if (fpUmi) outTag(fpUmi, e.name, e.hasUmi, e.umi);
            ++nKept;
          }
          break;
        case TestToCandidates:  // 752-769
          if (good[e.test]) candidates.insert(e.name);
          break;
        case SingleTest:  // 771-801
          if (e.aligned && usedName.count(e.name)) break;
          if (good[e.test]) {
            if (e.aligned) usedName.insert(e.name);
            outSeq(fp1, e.name, e.seq, e.qual);
            if (fpBc) outTag(fpBc, e.name, e.hasBc, e.bc);
            if (fpUmi) outTag(fpUmi, e.name, e.hasUmi, e.umi);
            ++nKept;
          }
          break;
        case SingleGene:  // 868-880
          if (usedName.count(e.name)) break;
          usedName.insert(e.name);
          outSeq(fp1, e.name, e.seq, e.qual);
          if (fpBc) outTag(fpBc, e.name, e.hasBc, e.bc);
          if (fpUmi) outTag(fpUmi, e.name, e.hasUmi, e.umi);
          ++nKept;
          break;
      }
    }
    events.clear(); batchSeq.clear(); batchOff.assign(1, 0);
    return true;
  };
  size_t batchReads = 1u << 17;
  if (const char *e = getenv("T1K_EXTRACT_CHUNK")) batchReads = (size_t)std::max(1, atoi(e));
  {
    Rec r;
    int tag = 0;
    std::string seq, qual;
    while (nextRecord(bam, r)) {
      const bool tAligned = r.templateAligned();
      if (!tAligned || (r.aligned() && validAlternativeChrom(bam.refName[(size_t)r.tid()]))) {
        if (!tAligned && paired && !abnormalUnaligned) {  // the two reads of an unaligned template come together (646-741)
          Event e;
          e.kind = UnalignedPair;
          std::string s1, q1, name(r.name());
          r.readSeq(s1); r.readQual(q1);
          if (!nextRecord(bam, r)) {
            fprintf(stderr, "Two reads from the unaligned fragment are not showing up together. Please use -u(--abnormalUnmapFlag from wrapper) option.\n");
            closeAll(true);
            return EXIT_FAILURE;
          }
          std::string mateName(r.name()), s2, q2;
          r.readSeq(s2); r.readQual(q2);
          trimName(name, mateIdLen); trimName(mateName, mateIdLen);
          if (name != mateName) {
            fprintf(stderr, "%s\n%s\n", name.c_str(), mateName.c_str());
            fprintf(stderr, "Two reads from the unaligned fragment are not showing up together. Please use -u(--abnormalUnmapFlag from wrapper) option.\n");
            closeAll(true);
            return EXIT_FAILURE;
          }
          e.name = name;
          if (!r.firstMate()) { e.seq = s1; e.qual = q1; e.seq2 = s2; e.qual2 = q2; }
          else { e.seq = s2; e.qual = q2; e.seq2 = s1; e.qual2 = q1; }
          tags(r, e);
          e.test = addTest(e.seq); e.test2 = addTest(e.seq2);
          events.push_back(std::move(e));
        } else if (paired) {  // a read on an alternative contig, or an unaligned one under -u: kept by name if it hits (752-769)
          Event e;
          e.kind = TestToCandidates;
          r.readSeq(e.seq);
          e.name = r.name();
          trimName(e.name, mateIdLen);
          if (!candidates.count(e.name)) { e.test = addTest(e.seq); events.push_back(std::move(e)); }
        } else {  // single-end (771-822)
          Event e;
          e.kind = SingleTest;
          e.aligned = r.aligned();
          r.readSeq(e.seq); r.readQual(e.qual);
          e.name = r.name();
          tags(r, e);
          e.test = addTest(e.seq);
          events.push_back(std::move(e));
        }
        if (batchOff.size() - 1 >= batchReads && !flush()) { closeAll(true); return EXIT_FAILURE; }
        continue;
      }
      if (!r.aligned()) continue;  // (paired data: this mate is unaligned, the other one is not)
      int64_t start, end;
      r.span(start, end);
      const int chr = r.tid();
      while (tag < geneCnt && (chr > genes[tag].chr || (chr == genes[tag].chr && start > genes[tag].end))) ++tag;  // the input is sorted by coordinate
      if (tag >= geneCnt) continue;
      if (chr < genes[tag].chr || (chr == genes[tag].chr && end <= genes[tag].start)) continue;
      r.readSeq(seq);
      if (isLowComplexity(seq)) continue;
      if (paired) {
        std::string name(r.name());
        trimName(name, mateIdLen);
        candidates.insert(name);
      } else {
        Event e;
        e.kind = SingleGene;
        e.name = r.name();
        e.seq = seq;
        r.readQual(e.qual);
        tags(r, e);
        events.push_back(std::move(e));
        if (events.size() >= 4 * batchReads && !flush()) { closeAll(true); return EXIT_FAILURE; }
      }
    }
    if (!bam.err.empty()) { fprintf(stderr, "bam-extractor: %s\n", bam.err.c_str()); closeAll(true); return EXIT_FAILURE; }
    if (!flush()) { closeAll(true); return EXIT_FAILURE; }
  }
  if (!paired) {
    if (nOverLong) fprintf(stderr, "bam-extractor: WARNING: %llu read(s) longer than %d bases were not tested against the reference sequences and not kept (the reference bam-extractor has no such limit)\n", (unsigned long long)nOverLong, T1K_BAM_MAX_READ);
    if (getenv("T1K_DEBUG_PHASES")) fprintf(stderr, "[t1k] bam-extractor: k=%d hitLenRequired=%d reads tested on the GPU %llu, kept %llu\n", kmerLength, hitLenRequired, (unsigned long long)nTested, (unsigned long long)nKept);
    closeAll(false);
    logLine("Finish extracting reads.");
    return 0;
  }
  // ---- second pass (895-938): both mates of every kept template, written when the second one shows up
  logLine("Finish obtaining the candidate read ids.");
  if (!bam.rewind()) { fprintf(stderr, "bam-extractor: %s\n", bam.err.c_str()); closeAll(true); return EXIT_FAILURE; }
  {
    struct Mates { std::string s1, q1, s2, q2; bool has1 = false, has2 = false; };
    std::unordered_map<std::string, Mates> open;
    open.reserve(candidates.size() * 2 + 16);
    for (auto &n : candidates) open.emplace(n, Mates());
    const size_t candidateCnt = open.size();
    size_t outputCnt = 0;
    Rec r;
    std::string name;
    while (outputCnt < candidateCnt && nextRecord(bam, r)) {
      if (!r.primary()) continue;
      if (!r.templateAligned() && !abnormalUnaligned) continue;
      name = r.name();
      trimName(name, mateIdLen);
      auto it = open.find(name);
      if (it == open.end()) continue;
      Mates &m = it->second;
      if (r.firstMate()) { r.readSeq(m.s1); r.readQual(m.q1); m.has1 = true; }
      else { r.readSeq(m.s2); r.readQual(m.q2); m.has2 = true; }
      if (m.has1 && m.has2) {
        outSeq(fp1, name, m.s1, m.q1);
        outSeq(fp2, name, m.s2, m.q2);
        if (fpBc) { const char *v = r.fieldZ(bcField.c_str()); outTag(fpBc, name, v != nullptr, v ? v : ""); }
        if (fpUmi) { const char *v = r.fieldZ(umiField.c_str()); outTag(fpUmi, name, v != nullptr, v ? v : ""); }
        m = Mates();
        ++outputCnt; ++nKept;
      }
    }
    if (!bam.err.empty()) { fprintf(stderr, "bam-extractor: %s\n", bam.err.c_str()); closeAll(true); return EXIT_FAILURE; }
  }
  if (nOverLong) fprintf(stderr, "bam-extractor: WARNING: %llu read(s) longer than %d bases were not tested against the reference sequences and not kept (the reference bam-extractor has no such limit)\n", (unsigned long long)nOverLong, T1K_BAM_MAX_READ);
  if (getenv("T1K_DEBUG_PHASES")) fprintf(stderr, "[t1k] bam-extractor: k=%d hitLenRequired=%d reads tested on the GPU %llu, templates kept %llu\n", kmerLength, hitLenRequired, (unsigned long long)nTested, (unsigned long long)nKept);
  closeAll(false);
  logLine("Finish extracting reads.");
  return 0;
}
