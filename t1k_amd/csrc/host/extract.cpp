// t1k_amd/csrc/host/extract.cpp -- host side of the candidate-read extractor: the argv-compatible replacement of the reference's
// fastq-extractor main() (FastqExtractor.cpp:260-626, called by run-t1k:377-403).  Input parsing, parameter inference and the
// output writers stay on the CPU; the per-read test IsGoodCandidate runs on the GPU (t1k_extract_batch), one chunk of fragments at a
// time, with the readers running ahead of the GPU on their own threads.  There is no CPU fallback: without a GPU the run fails.
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unistd.h>
#include <unordered_map>
#include <vector>

#include "t1k_host.h"

namespace {

// FASTA/FASTQ records with the rules of kseq.h:185-224 (the reader behind ReadFiles.hpp), as a state machine over this reader's own buffer
// -- the same rules as the general reader of host/refset.cpp, here beside an in-place path for the common case: the name ends at the first
// isspace() character; sequence lines are joined up to a line that starts with '+', '>' or '@' (empty lines skipped); behind a '+' line
// quality lines are joined until they are at least as long as the sequence, and another length ends the FILE (kseq_read returns -2,
// ReadFiles::Next goes on with the next file); behind a FASTQ record the next record starts at the next '@' or '>' wherever it stands; a
// trailing CR is dropped from a line only when what has been gathered is longer than one character (kseq.h:142).
struct RecordReader {
  gzFile fp = nullptr;
  std::vector<char> buf;
  size_t pos = 0, end = 0;
  bool eof = false;
  int last = 0;  // the header character of the next record, already taken from the buffer (0: none)
  explicit RecordReader(const std::string &path, size_t bufBytes = (size_t)1 << 22) : buf(std::max<size_t>(bufBytes, 16)) {
    fp = gzopen(path.c_str(), "r");
    if (fp) gzbuffer(fp, 1 << 20);
  }
  ~RecordReader() { if (fp) gzclose(fp); }
  bool fill() {  // false at the end of the file
    if (eof) return false;
    const int n = gzread(fp, buf.data(), (unsigned)buf.size());
    if (n <= 0) { eof = true; pos = end = 0; return false; }
    pos = 0; end = (size_t)n;
    return true;
  }
  int getc() {
    if (pos >= end && !fill()) return -1;
    return (unsigned char)buf[pos++];
  }
  // gathers up to the end of the line (line) or the first isspace() character; the delimiter is consumed and reported in *delim (0: none);
  // -1 = nothing left in the file
  long until(bool line, std::string &str, int *delim, bool append) {
    if (delim) *delim = 0;
    if (!append) str.clear();
    if (pos >= end && !fill()) return -1;
    for (;;) {
      size_t i = pos;
      if (line) { const char *nl = (const char *)memchr(buf.data() + pos, '\n', end - pos); i = nl ? (size_t)(nl - buf.data()) : end; }
      else while (i < end && !isspace((unsigned char)buf[i])) ++i;
      str.append(buf.data() + pos, i - pos);
      if (i < end) { if (delim) *delim = (unsigned char)buf[i]; pos = i + 1; break; }
      pos = end;
      if (!fill()) break;
    }
    if (line && str.size() > 1 && str.back() == '\r') str.pop_back();
    return (long)str.size();
  }
  // The common case without copies: a four-line FASTQ record that lies entirely in the buffer.  The pointers stay valid until the next
  // call.  Returns false if the record is anything else (FASTA, wrapped lines, CR line ends, a record cut by the buffer end, a header
  // character taken already): the caller then takes next().
  bool nextInPlace(const char *&name, size_t &nameLen, const char *&seq, size_t &seqLen, const char *&qual) {
    if (last != 0 || pos >= end) return false;
    const char *p = buf.data() + pos, *e = buf.data() + end;
    if (*p != '@') return false;
    const char *l1 = (const char *)memchr(p, '\n', e - p);
    if (!l1 || l1 + 1 >= e) return false;
    const char *l2 = (const char *)memchr(l1 + 1, '\n', e - (l1 + 1));
    if (!l2 || l2 + 1 >= e || l2[1] != '+') return false;
    const char *l3 = (const char *)memchr(l2 + 1, '\n', e - (l2 + 1));
    if (!l3 || l3 + 1 >= e) return false;
    const char *l4 = (const char *)memchr(l3 + 1, '\n', e - (l3 + 1));
    if (!l4) return false;
    if (l2 - l1 != l4 - l3 || l2 - l1 < 2 || l1[-1] == '\r' || l2[-1] == '\r' || l4[-1] == '\r') return false;
    if (l1[1] == '>' || l1[1] == '@' || l1[1] == '+') return false;  // (a sequence line that looks like a header: the general path's case)
    const char *sp = p + 1;
    while (sp < l1 && !isspace((unsigned char)*sp)) ++sp;
    name = p + 1; nameLen = (size_t)(sp - (p + 1));
    seq = l1 + 1; seqLen = (size_t)(l2 - l1 - 1);
    qual = l3 + 1;
    pos = (size_t)(l4 - buf.data()) + 1;
    return true;
  }
  // name (up to the first blank), sequence, quality ("" for FASTA); false at the end of the file (or at a record that ends it)
  bool next(std::string &name, std::string &seq, std::string &qual) {
    int c;
    if (last == 0) {
      while ((c = getc()) != -1 && c != '>' && c != '@') {}
      if (c == -1) return false;
      last = c;
    }
    seq.clear(); qual.clear();
    if (until(false, name, &c, false) < 0) return false;
    if (c != '\n') { std::string comment; until(true, comment, nullptr, false); }
    while ((c = getc()) != -1 && c != '>' && c != '+' && c != '@') {
      if (c == '\n') continue;
      seq.push_back((char)c);
      until(true, seq, nullptr, true);
    }
    if (c == '>' || c == '@') last = c;
    if (c != '+') return true;  // FASTA
    while ((c = getc()) != -1 && c != '\n') {}
    if (c == -1) return false;  // no quality string: the end of this file for ReadFiles::Next
    while (until(true, qual, nullptr, true) >= 0 && qual.size() < seq.size()) {}
    last = 0;
    return qual.size() == seq.size();  // another length: kseq_read returns -2, the file ends here
  }
};

struct EndChunk {  // one chunk of one input stream
  std::string arena;  // name \0 seq \0 qual \0 per record
  std::vector<uint64_t> off;      // start of each record in the arena
  std::vector<uint32_t> nameLen, seqLen;
  std::vector<uint8_t> hasQual;
  size_t n() const { return nameLen.size(); }
  const char *name(size_t i) const { return arena.data() + off[i]; }
  const char *seq(size_t i) const { return arena.data() + off[i] + nameLen[i] + 1; }
  const char *qual(size_t i) const { return hasQual[i] ? seq(i) + seqLen[i] + 1 : nullptr; }
};

// One input stream = the files given with repeated -1 / -2 / -u / --barcode, read back to back on its own thread.  mod / rem select
// the records of an interleaved file (-i): record r is kept if r % mod == rem.
struct Stream {
  std::vector<std::string> files;
  int mod = 1, rem = 0;
  size_t chunkRecords = 1 << 20;
  size_t bufBytes = (size_t)1 << 22;  // the reader's buffer (small in tests: records cut by the buffer end)
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::unique_ptr<EndChunk>> q;
  bool done = false, failed = false;
  std::thread th;
  void start() { th = std::thread([this] { run(); }); }
  void run() {
    auto cur = std::make_unique<EndChunk>();
    auto flush = [&](bool last) {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return q.size() < 2; });
      if (cur->n() || last) q.push_back(std::move(cur));
      if (last) done = true;
      cv.notify_all();
      cur = std::make_unique<EndChunk>();
    };
    std::string name, seq, qual;
    for (auto &f : files) {
      RecordReader rd(f, bufBytes);
      if (!rd.fp) { failed = true; break; }
      uint64_t r = 0;
      while (true) {
        const char *pn, *ps, *pq;
        size_t nl, sl;
        if (rd.nextInPlace(pn, nl, ps, sl, pq)) {
          if ((int)(r++ % mod) != rem) continue;
          cur->off.push_back(cur->arena.size());
          cur->nameLen.push_back((uint32_t)nl);
          cur->seqLen.push_back((uint32_t)sl);
          cur->hasQual.push_back(1);
          cur->arena.append(pn, nl); cur->arena.push_back('\0');
          cur->arena.append(ps, sl); cur->arena.push_back('\0');
          cur->arena.append(pq, sl); cur->arena.push_back('\0');
        } else {
          if (!rd.next(name, seq, qual)) break;
          if ((int)(r++ % mod) != rem) continue;
          cur->off.push_back(cur->arena.size());
          cur->nameLen.push_back((uint32_t)name.size());
          cur->seqLen.push_back((uint32_t)seq.size());
          cur->hasQual.push_back(qual.empty() ? 0 : 1);
          cur->arena.append(name); cur->arena.push_back('\0');
          cur->arena.append(seq); cur->arena.push_back('\0');
          if (!qual.empty()) { cur->arena.append(qual); cur->arena.push_back('\0'); }
        }
        if (cur->n() >= chunkRecords) flush(false);
      }
    }
    flush(true);
  }
  // next chunk (possibly empty at the very end); nullptr when the stream is exhausted
  std::unique_ptr<EndChunk> pop() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return !q.empty() || done; });
    if (q.empty()) return nullptr;
    auto c = std::move(q.front());
    q.pop_front();
    cv.notify_all();
    return c;
  }
  ~Stream() { if (th.joinable()) th.join(); }
};

void printLog(const char *msg) {  // PrintLog (FastqExtractor.cpp:76-87)
  time_t t = time(nullptr);
  char stime[500];
  strftime(stime, sizeof(stime), "%c", localtime(&t));
  fprintf(stderr, "[%s] %s\n", stime, msg);
}

const char kUsage[] =
    "./fastq-extractor [OPTIONS]:\n"
    "Required:\n"
    "\t-f STRING: fasta file containing the reference sequence\n"
    "\t-u STRING: path to single-end read file\n"
    "\t\tor\n"
    "\t-1 STRING -2 STRING: path to paired-end read files\n"
    "\t\tor\n"
    "\t-i STRING: path to interleaved read file\n"
    "Optional:\n"
    "\t-o STRING: prefix to the output file (default: toassemble)\n"
    "\t-t INT: number of threads (default: 1)\n"
    "\t-s FLOAT: filter alignments with alignment similarity less than specified value (defalut: 0.8)\n"
    "\t--barcode STRING: path to the raw barcode file (default: not used)\n"
    "\t--barcodeStart INT: the start position of barcode in the barcode sequence (default: 0)\n"
    "\t--barcodeEnd INT: the end position of barcode in the barcode sequence (default: length-1)\n"
    "\t--barcodeRevComp: whether the barcode need to be reverse complemented (default: not used)\n"
    "\t--barcodeWhitelist STRING: path to the barcode whitelist (default: not used)\n"
    "\t--read1Start INT: the start position of sequence in read 1 (default: 0)\n"
    "\t--read1End INT: the end position of sequence in read 1 (default: length-1)\n"
    "\t--read2Start INT: the start position of sequence in read 2 (default: 0)\n"
    "\t--read2End INT: the end position of sequence in read 2 (default: length-1)\n";

// OutputSeq (FastqExtractor.cpp:120-154)
void outputSeq(std::string &out, const char *name, size_t nameLen, const char *seq, const char *qual, size_t len, int start, int end) {
  size_t s = 0, n = len;
  if (!(start == 0 && end == -1)) {
    const long e = end == -1 ? (long)len - 1 : end;
    s = (size_t)start;
    n = e >= start ? (size_t)(e - start + 1) : 0;
    if (s > len) { s = len; n = 0; }
    if (s + n > len) n = len - s;
  }
  out.push_back(qual ? '@' : '>');
  out.append(name, nameLen); out.push_back('\n');
  out.append(seq + s, n); out.push_back('\n');
  if (qual) { out.append("+\n"); out.append(qual + s, n); out.push_back('\n'); }
}

// BarcodeCorrector::FormatBarcode (BarcodeCorrector.hpp:119-139) / the sub-range and reverse complement of OutputBarcode
std::string formatBarcode(const char *bc, size_t bl, int start, int end, bool revcomp) {
  if (start == 0 && end == -1 && !revcomp) return std::string(bc, bl);
  std::string out;
  const long s = start, e = end == -1 ? (long)bl - 1 : end;
  if (!revcomp) { for (long x = s; x <= e && x < (long)bl; ++x) out.push_back(bc[x]); }
  else {
    for (long x = std::min(e, (long)bl - 1); x >= s; --x) {  // SeqSet::ReverseComplement (SeqSet.hpp:2103-2114)
      const char c = bc[x];
      out.push_back(c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N');
    }
  }
  return out;
}

// BarcodeCorrector (BarcodeCorrector.hpp:104-239): whitelist with background counts; a barcode that is not on the list is replaced by
// the one-substitution neighbour on the list with the highest count (first in (position, base) order on ties, or -- with qualities --
// the one whose changed position has the lowest quality)
struct BarcodeCorrector {
  // the reference's Trie (BarcodeCorrector.hpp:17-102): a look-up succeeds for any path that exists, whole barcode or prefix of one, and
  // returns (and updates) the count stored at the node it ends on
  std::vector<int> next{-1, -1, -1, -1}, cnt{0};
  static int code(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }
  static bool acgt(const std::string &s) { for (char c : s) if (code(c) < 0) return false; return true; }
  void insert(const std::string &s, int weight) {
    if (!acgt(s)) return;
    int p = 0;
    for (char c : s) {
      const int t = code(c);
      if (next[4 * p + t] < 0) { next[4 * p + t] = (int)cnt.size(); cnt.push_back(0); next.insert(next.end(), 4, -1); }
      p = next[4 * p + t];
    }
    cnt[p] += weight;
  }
  int searchAndUpdate(const std::string &s, int weight) {  // the count after the update, -1 if there is no such path
    if (!acgt(s)) return -1;
    int p = 0;
    for (char c : s) { p = next[4 * p + code(c)]; if (p < 0) return -1; }
    cnt[p] += weight;
    return cnt[p];
  }
  bool load(const std::string &path) {  // SetWhitelist (145-152)
    FILE *fp = fopen(path.c_str(), "r");
    if (!fp) return false;
    char buffer[256];
    while (fscanf(fp, "%255s", buffer) != EOF) insert(buffer, 1);
    fclose(fp);
    return true;
  }
  int count(const std::string &b) { return searchAndUpdate(b, 0); }
  void observe(const std::string &b) { searchAndUpdate(b, 1); }  // CollectBackgroundDistribution (154-168)
  // Correct (170-237): 0 = on the list, 1 = corrected in place, -1 = no candidate
  int correct(std::string &barcode, const char *qual) {
    if (count(barcode) != -1) return 0;
    static const char testChr[5] = "ACGT";
    int bestCnt = -1, bestPos = -1, bestBase = -1, bestLowQual = 255;
    std::string buffer = barcode;
    for (size_t i = 0; i < barcode.size(); ++i)
      for (int j = 0; j < 4; ++j) {
        if (testChr[j] == barcode[i]) continue;
        buffer[i] = testChr[j];
        const int cnt = count(buffer);
        buffer[i] = barcode[i];
        if (cnt == -1) continue;
        if (cnt > bestCnt) { bestCnt = cnt; bestPos = (int)i; bestBase = j; if (qual) bestLowQual = qual[i]; }
        else if (cnt == bestCnt && qual && qual[i] < bestLowQual) { bestLowQual = qual[i]; bestPos = (int)i; bestBase = j; }
      }
    if (bestPos < 0) return -1;
    barcode[bestPos] = testChr[bestBase];
    return 1;
  }
};

// ------------------------------------------------------------------------------------------------------------------
// The fast way through ordinary read files (SURVEY 8f row 1 end to end).  The streaming loop below parses records on one thread per
// file into owned chunks, interleaves the mates on the main thread, uploads from pageable memory and writes -- 3.4 M pairs/s whatever
// the kernels do (they screen 300 M pairs/s).  When the input is what a sequencer / the pipeline delivers -- regular files (plain or
// gz) in the strict four-line FASTQ / two-line FASTA layout, no barcode stream, not interleaved -- the files are mapped and indexed in
// place by the host threads exactly as the genotyper does (ReadInput, host/reads.cpp), and batches of fragments alternate between TWO
// device contexts that share the index: while one context's kernels test batch b, the other's worker gathers the read text of batch
// b + 1 from the mapping into page-locked staging slots (all host threads) and sends it piece by piece.  The kept records are
// formatted by the host threads straight from the mapping and written in batch order.  Same records, same order, same bytes as the
// streaming loop (and the reference); anything else about the input falls back to that loop.
// ------------------------------------------------------------------------------------------------------------------
template <class F>
void parallelPieces(size_t n, int T, F fn) {  // fn(t, begin, end) over contiguous pieces of [0, n)
  T = (int)std::max<size_t>(1, std::min<size_t>((size_t)T, n / 2048 + 1));
  if (T == 1) { fn(0, (size_t)0, n); return; }
  std::vector<std::thread> th;
  const size_t per = (n + T - 1) / T;
  for (int t = 0; t < T; ++t) th.emplace_back([=] { fn(t, std::min(n, t * per), std::min(n, (t + 1) * per)); });
  for (auto &x : th) x.join();
}

struct MappedRun {
  int r1s = 0, r1e = -1, r2s = 0, r2e = -1, threadCnt = 1;
  bool hasMate = false, dbg = false;
  std::string prefix;
};

// The input of the mapped path, opened on its own thread while the device context and the reference index are being made (the index
// of 4 M pairs takes the host threads 60 - 90 ms: as long as the HIP runtime needs to come up).
struct MappedInput {
  t1k::ReadInput in;
  std::thread th;
  int state = 0;  // 1 = indexed in place, 0 = not eligible (the streaming loop takes the input), -1 = the mates hold different numbers of reads
  void start(const std::vector<std::string> &files1, const std::vector<std::string> &files2, bool hasMate) {
    th = std::thread([this, &files1, &files2, hasMate] {
      if (getenv("T1K_EXTRACT_STREAM")) return;
      const int hw = (int)std::thread::hardware_concurrency();
      const int T = std::max(2, std::min(hw > 0 ? hw : 2, 32));
      // the whole input is mapped and indexed (22 bytes of index per record): beyond this much TEXT the streaming loop's bounded memory
      // wins.  A gzip file is inflated whole into anonymous memory, so what counts for it is its uncompressed size: the ISIZE field of the
      // last member where the file has one member (exact below 4 GB), never less than 5 x the file (FASTQ deflates 4 - 5 x; ISIZE wraps at
      // 4 GB and names only the last member of a multi-member / bgzip file).  Inflated text is resident, unlike a mapping of the page
      // cache: it must also fit the memory the host has free right now.
      uint64_t bytes = 0, resident = 0;
      for (const auto *fs : {&files1, &files2})
        for (const auto &f : *fs) {
          FILE *fp = fopen(f.c_str(), "rb");
          if (!fp) return;  // (the streaming loop reports it the reference's way)
          uint64_t sz = 0;
          unsigned char magic[2] = {0, 0}, tail[4] = {0, 0, 0, 0};
          const bool gz = fread(magic, 1, 2, fp) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
          if (fseeko(fp, 0, SEEK_END) == 0) sz = (uint64_t)ftello(fp);
          if (gz) {
            uint64_t text = sz * 5;
            if (sz >= 18 && fseeko(fp, -4, SEEK_END) == 0 && fread(tail, 1, 4, fp) == 4)
              text = std::max<uint64_t>(text, (uint64_t)tail[0] | ((uint64_t)tail[1] << 8) | ((uint64_t)tail[2] << 16) | ((uint64_t)tail[3] << 24));
            bytes += text; resident += text;
          } else bytes += sz;
          fclose(fp);
        }
      const char *e = getenv("T1K_EXTRACT_MAP_GB");
      if ((double)bytes > (e ? atof(e) : 256.0) * 1073741824.0) return;
      if (resident) {
        const long pages = sysconf(_SC_AVPHYS_PAGES), psz = sysconf(_SC_PAGESIZE);
        // inflated text + its record index (22 B per ~300 B record) must leave half of what is free to everybody else
        if (pages > 0 && psz > 0 && (double)resident * 1.08 > 0.5 * (double)pages * (double)psz) return;
      }
      std::string err;
      if (!in.open(files1, hasMate ? files2 : std::vector<std::string>(), "", T, err)) {
        if (err.find("different numbers of reads") != std::string::npos) state = -1;
        return;
      }
      if (in.inPlace) state = 1;  // (wrapped FASTA, blank lines ...: records in owned storage without their qualities -> streaming loop)
    });
  }
  int wait() { if (th.joinable()) th.join(); return state; }
  ~MappedInput() { if (th.joinable()) th.join(); }
};

// 1 = done (rc holds the exit code), 0 = not eligible: the caller runs the streaming loop
int extractMapped(t1k_ctx *ctx0, const t1k_params &prm, int device, MappedInput &mi, const MappedRun &o, int &rc, uint64_t &nFragments, uint64_t &nGood) {
  const int st = mi.wait();
  if (st == 0) return 0;
  if (st < 0) { fprintf(stderr, "The two mate-pair read files have different number of reads.\n"); rc = 1; return 1; }
  t1k::ReadInput &in = mi.in;
  const int hw = (int)std::thread::hardware_concurrency();
  const int T = std::max(2, std::min(hw > 0 ? hw : 2, 32));
  const size_t F = in.nFrag();
  const uint32_t per = o.hasMate ? 2 : 1;
  size_t B = 1u << 20;
  if (const char *e = getenv("T1K_EXTRACT_CHUNK")) B = (size_t)std::max(1, atoi(e));
  const size_t nBatches = (F + B - 1) / B;
  // two contexts on the one index
  t1k_ctx *ctx[2] = {ctx0, nullptr};
  if (nBatches > 1) {
    if (t1k_ctx_create(device, &prm, &ctx[1]) != T1K_OK || t1k_ref_share(ctx[1], ctx0) != T1K_OK) { if (ctx[1]) t1k_ctx_destroy(ctx[1]); ctx[1] = nullptr; }
  }
  const int nWorkers = ctx[1] ? 2 : 1;
  struct Done { std::string o1, o2; uint64_t good = 0; bool ready = false; };
  std::vector<Done> done(nBatches);
  std::mutex mu;
  std::condition_variable cv;
  bool failed = false;
  std::string failMsg;
  size_t written = 0;  // batches the writer has taken (a worker runs at most 3 batches ahead of it: bounded output memory)
  const size_t slotBytes = 64u << 20;
  auto rawNameLen = [](const char *id, size_t il) {  // the name as kseq cuts it (up to the first blank): ReadInput dropped a trailing /1 or /2
    if (id[il] == '/' && (id[il + 1] == '1' || id[il + 1] == '2') && (id[il + 2] == ' ' || id[il + 2] == '\t' || id[il + 2] == '\r' || id[il + 2] == '\n')) return il + 2;
    return il;
  };
  auto qualOf = [](const char *id, const char *seq, size_t sl) -> const char * {  // four-line FASTQ: the line after the '+' line (the indexer checked both exist); FASTA: none
    if (id[-1] != '@') return nullptr;
    const char *p = seq + sl;
    if (*p == '\r') ++p;
    ++p;  // the newline
    if (*p != '+') return nullptr;
    while (*p != '\n') ++p;
    return p + 1;
  };
  auto worker = [&](int w) {
    t1k_ctx *c = ctx[w];
    char *ring = (char *)t1k_pinned_alloc(3 * slotBytes);
    const bool pinned = ring != nullptr;
    if (!ring) ring = (char *)malloc(3 * slotBytes);
    std::vector<uint64_t> off;
    std::vector<uint8_t> good;
    const int Tw = std::max(1, T / nWorkers);
    for (size_t b = (size_t)w; b < nBatches && ring; b += (size_t)nWorkers) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return failed || b < written + 4; });
        if (failed) break;
      }
      const size_t f0 = b * B, nf = std::min(B, F - f0), ne = nf * per;
      off.resize(ne + 1);
      std::vector<uint64_t> pieceBytes((size_t)Tw + 2, 0);
      std::vector<uint32_t> pieceMax((size_t)Tw + 1, 0);
      parallelPieces(nf, Tw, [&](int t, size_t lo, size_t hi) {
        uint64_t run = 0; uint32_t mx = 0;
        for (size_t i = lo; i < hi; ++i) {
          const uint32_t r = in.frag[f0 + i];
          for (uint32_t m = 0; m < per; ++m) { const uint32_t len = in.side[m].seqL[r]; off[i * per + m] = run; run += len; mx = std::max(mx, len); }
        }
        pieceBytes[(size_t)t + 1] = run; pieceMax[(size_t)t] = mx;
      });
      const int usedT = (int)std::max<size_t>(1, std::min<size_t>((size_t)Tw, nf / 2048 + 1));
      for (int t = 0; t < usedT; ++t) pieceBytes[(size_t)t + 1] += pieceBytes[(size_t)t];
      const uint64_t total = pieceBytes[(size_t)usedT];
      uint32_t maxLen = 0;
      for (int t = 0; t < usedT; ++t) maxLen = std::max(maxLen, pieceMax[(size_t)t]);
      parallelPieces(nf, Tw, [&](int t, size_t lo, size_t hi) {
        const uint64_t carry = pieceBytes[(size_t)t];
        for (size_t i = lo; i < hi; ++i) for (uint32_t m = 0; m < per; ++m) off[i * per + m] += carry;
      });
      off[ne] = total;
      int r = t1k_reads_upload_begin(c, (uint32_t)ne, total, (int)maxLen);
      if (r == T1K_OK) r = t1k_reads_upload_piece(c, 1, off.data(), 0, ((uint64_t)ne + 1) * 8, 3);
      uint32_t nPieces = 0;
      for (size_t i0 = 0; i0 < nf && r == T1K_OK; ++nPieces) {
        const uint64_t base = off[i0 * per];
        size_t lo = i0 + 1, hi = nf;  // the last fragment boundary whose text still fits the slot (one fragment always does)
        while (lo < hi) { const size_t mid = lo + (hi - lo + 1) / 2; if ((mid < nf ? off[mid * per] : total) - base <= slotBytes) lo = mid; else hi = mid - 1; }
        const size_t i1 = lo;
        const int slot = (int)(nPieces % 3);
        if ((r = t1k_reads_upload_wait(c, slot)) != T1K_OK) break;
        char *dst = ring + (size_t)slot * slotBytes;
        parallelPieces(i1 - i0, Tw, [&](int, size_t a, size_t z) {
          for (size_t i = i0 + a; i < i0 + z; ++i) {
            const uint32_t rr = in.frag[f0 + i];
            for (uint32_t m = 0; m < per; ++m) memcpy(dst + (off[i * per + m] - base), in.side[m].seqP[rr], in.side[m].seqL[rr]);
          }
        });
        r = t1k_reads_upload_piece(c, 0, dst, base, (i1 < nf ? off[i1 * per] : total) - base, slot);
        i0 = i1;
      }
      if (r == T1K_OK) r = t1k_reads_upload_end(c);
      good.assign(nf, 0);
      if (r == T1K_OK) r = t1k_extract_batch(c, per, good.data(), nullptr);
      if (r != T1K_OK) {
        std::lock_guard<std::mutex> g(mu);
        if (!failed) { failed = true; failMsg = t1k_last_error(c); }
        cv.notify_all();
        break;
      }
      // the kept records, formatted from the mapping by the host threads (pieces in order)
      Done d;
      {
        const int Tf = (int)std::max<size_t>(1, std::min<size_t>((size_t)Tw, nf / 2048 + 1));
        std::vector<std::string> p1((size_t)Tf), p2((size_t)Tf);
        std::vector<uint64_t> cnt((size_t)Tf, 0);
        parallelPieces(nf, Tw, [&](int t, size_t lo, size_t hi) {
          std::string &a = p1[(size_t)t], &bb = p2[(size_t)t];
          for (size_t i = lo; i < hi; ++i) {
            if (!good[i]) continue;
            ++cnt[(size_t)t];
            const uint32_t rr = in.frag[f0 + i];
            const char *nm = in.side[0].idP[rr];
            // -t 1 prints ReadFiles::Next()'s id (a trailing /1 or /2 removed), -t > 1 the raw name (FastqExtractor.cpp:446-476 vs 529-545)
            const size_t nl = o.threadCnt == 1 ? (size_t)in.side[0].idL[rr] : rawNameLen(nm, in.side[0].idL[rr]);
            outputSeq(a, nm, nl, in.side[0].seqP[rr], qualOf(nm, in.side[0].seqP[rr], in.side[0].seqL[rr]), in.side[0].seqL[rr], o.r1s, o.r1e);
            if (o.hasMate) outputSeq(bb, nm, nl, in.side[1].seqP[rr], qualOf(in.side[1].idP[rr], in.side[1].seqP[rr], in.side[1].seqL[rr]), in.side[1].seqL[rr], o.r2s, o.r2e);
          }
        });
        for (int t = 0; t < Tf; ++t) { d.o1 += p1[(size_t)t]; d.o2 += p2[(size_t)t]; d.good += cnt[(size_t)t]; }
      }
      d.ready = true;
      {
        std::lock_guard<std::mutex> g(mu);
        done[b] = std::move(d);
      }
      cv.notify_all();
    }
    if (pinned) t1k_pinned_free(ring); else free(ring);
    if (!ring) { std::lock_guard<std::mutex> g(mu); if (!failed) { failed = true; failMsg = "out of host memory"; } cv.notify_all(); }
  };
  FILE *fp1 = fopen((o.prefix + (o.hasMate ? "_1.fq" : ".fq")).c_str(), "w");
  FILE *fp2 = o.hasMate ? fopen((o.prefix + "_2.fq").c_str(), "w") : nullptr;
  if (!fp1 || (o.hasMate && !fp2)) {
    fprintf(stderr, "Cannot open the output files.\n");
    if (fp1) fclose(fp1);
    if (fp2) fclose(fp2);
    if (ctx[1]) t1k_ctx_destroy(ctx[1]);
    rc = EXIT_FAILURE;
    return 1;
  }
  std::vector<std::thread> th;
  for (int w = 0; w < nWorkers; ++w) th.emplace_back(worker, w);
  for (size_t b = 0; b < nBatches; ++b) {
    Done d;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return failed || done[b].ready; });
      if (failed) break;
      d = std::move(done[b]);
      written = b + 1;
    }
    cv.notify_all();
    fwrite(d.o1.data(), 1, d.o1.size(), fp1);
    if (fp2) fwrite(d.o2.data(), 1, d.o2.size(), fp2);
    nGood += d.good;
    nFragments += std::min(B, F - b * B);
  }
  for (auto &t : th) t.join();
  fclose(fp1);
  if (fp2) fclose(fp2);
  if (ctx[1]) t1k_ctx_destroy(ctx[1]);
  rc = 0;
  if (failed) {
    fprintf(stderr, "fastq-extractor: %s\n", failMsg.c_str());
    rc = 1;
    remove((o.prefix + (o.hasMate ? "_1.fq" : ".fq")).c_str());  // no truncated candidate files for the next stage
    if (o.hasMate) remove((o.prefix + "_2.fq").c_str());
  }
  return 1;
}

}  // namespace

extern "C" int t1k_extractor_main(int argc, char **argv) {
  if (argc <= 1) { fprintf(stderr, "%s", kUsage); return 0; }
  std::string refPath, prefix = "toassemble";
  Stream reads, mates, barcodes;
  bool hasMate = false, hasBarcode = false, barcodeRevComp = false, hasWhitelist = false;
  BarcodeCorrector corrector;
  double similarity = 0.8;
  int threadCnt = 1, barcodeStart = 0, barcodeEnd = -1, r1s = 0, r1e = -1, r2s = 0, r2e = -1;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto val = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
    if (a == "-f") refPath = val();
    else if (a == "-o") prefix = val();
    else if (a == "-1") { reads.files.push_back(val()); hasMate = true; }
    else if (a == "-2") { mates.files.push_back(val()); hasMate = true; }
    else if (a == "-u") reads.files.push_back(val());
    else if (a == "-i") {
      const char *f = val();
      if (!reads.files.empty() || !mates.files.empty()) { fprintf(stderr, "-i takes the place of every other read file option.\n"); return EXIT_FAILURE; }
      reads.files.push_back(f); mates.files.push_back(f);
      reads.mod = mates.mod = 2; reads.rem = 0; mates.rem = 1;
      hasMate = true;
    }
    else if (a == "-t") threadCnt = atoi(val());
    else if (a == "-s") similarity = atof(val());
    else if (a == "--barcode") { hasBarcode = true; barcodes.files.push_back(val()); }
    else if (a == "--barcodeStart") barcodeStart = atoi(val());
    else if (a == "--barcodeEnd") barcodeEnd = atoi(val());
    else if (a == "--barcodeRevComp") barcodeRevComp = true;
    else if (a == "--barcodeWhitelist") {
      hasWhitelist = true;
      if (!corrector.load(val())) { fprintf(stderr, "Cannot open the barcode whitelist.\n"); return EXIT_FAILURE; }
    }
    else if (a == "--read1Start") r1s = atoi(val());
    else if (a == "--read1End") r1e = atoi(val());
    else if (a == "--read2Start") r2s = atoi(val());
    else if (a == "--read2End") r2e = atoi(val());
    else { fprintf(stderr, "Unknown parameter %s\n", a.c_str()); return EXIT_FAILURE; }
  }
  if (refPath.empty()) { fprintf(stderr, "Need to use -f to specify the reference sequence.\n"); return EXIT_FAILURE; }
  if (reads.files.empty()) { fprintf(stderr, "Need to use -u/-1/-2/-i to specify the read files.\n"); return EXIT_FAILURE; }
  printLog("Start to extract candidate reads from read files.");
  // the HIP runtime comes up (0.1 s) while the reference and the first reads are parsed
  int nDevices = 0;
  struct Joiner { std::thread t; ~Joiner() { if (t.joinable()) t.join(); } } hipUp;
  hipUp.t = std::thread([&nDevices] { nDevices = t1k_device_count(); });
  const bool dbg = getenv("T1K_DEBUG_PHASES") != nullptr;
  auto tStart = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!dbg) return;
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[t1k] extractor %s: %.3f s\n", what, std::chrono::duration<double>(now - tStart).count());
    tStart = now;
  };

  // reference: one sequence per FASTA record (SeqSet::InputRefFa, SeqSet.hpp:872-904)
  std::vector<t1k::SeqRec> ref;
  std::string err;
  if (!t1k::readSeqFile(refPath, ref, err) || ref.empty()) { fprintf(stderr, "%s\n", err.empty() ? "empty reference" : err.c_str()); return EXIT_FAILURE; }
  lap("reference read");
  // hitLenRequired from the first 1000 reads (FastqExtractor.cpp:383-401)
  int hitLenRequired = hasMate ? 27 : 23;
  {
    int len = 0, n = 0;
    std::string name, seq, qual;
    uint64_t r = 0;
    for (auto &f : reads.files) {
      RecordReader rd(f);
      if (!rd.fp) { fprintf(stderr, "Cannot open %s\n", f.c_str()); return EXIT_FAILURE; }
      while (n < 1000 && rd.next(name, seq, qual)) {
        if ((int)(r++ % reads.mod) != reads.rem) continue;
        len += (int)seq.size(); ++n;
      }
      if (n >= 1000) break;
    }
    if (n == 0) { fprintf(stderr, "Read file is empty.\n"); return EXIT_FAILURE; }
    if (len / (n * 5) > hitLenRequired) hitLenRequired = len / (n * 5);
  }
  // k-mer length (SeqSet::InferKmerLength 2830-2845, FastqExtractor.cpp:409-416); the total is an int in the reference as well
  int kmerLength = 9;
  {
    int total = 0;
    for (auto &r : ref) total += (int)r.seq.size();
    int ret = 0;
    while (total) { ++ret; total /= 4; }
    ++ret;
    if (ret > kmerLength) { kmerLength = ret; if (kmerLength > hitLenRequired) hitLenRequired = kmerLength; }
  }

  if (hasBarcode && hasWhitelist) {  // CollectBackgroundDistribution over the first 2,000,000 barcodes (FastqExtractor.cpp:420-423)
    std::string name, seq, qual;
    int seen = 0;
    for (auto &f : barcodes.files) {
      RecordReader rd(f);
      if (!rd.fp) { fprintf(stderr, "Cannot open %s\n", f.c_str()); return EXIT_FAILURE; }
      while (seen < 2000000 && rd.next(name, seq, qual)) { corrector.observe(formatBarcode(seq.data(), seq.size(), barcodeStart, barcodeEnd, barcodeRevComp)); ++seen; }
      if (seen >= 2000000) break;
    }
  }
  lap("parameters from the first reads");
  hipUp.t.join();
  if (nDevices <= 0) { fprintf(stderr, "fastq-extractor: no HIP device (this build has no CPU path)\n"); return EXIT_FAILURE; }
  // the readers start now and parse their first chunks while the index is built and uploaded
  if (const char *e = getenv("T1K_EXTRACT_CHUNK")) reads.chunkRecords = mates.chunkRecords = barcodes.chunkRecords = (size_t)std::max(1, atoi(e));
  // (input the mapped path below can take is not parsed by the stream threads at all: it is mapped and indexed while the context comes up)
  bool started = false;
  MappedInput mapped;
  if (!(hasBarcode || reads.mod != 1)) mapped.start(reads.files, mates.files, hasMate);
  if (hasBarcode || reads.mod != 1 || getenv("T1K_EXTRACT_STREAM")) { reads.start(); if (hasMate) mates.start(); if (hasBarcode) barcodes.start(); started = true; }
  auto drain = [&]() { if (!started) return; while (reads.pop()) {} if (hasMate) while (mates.pop()) {} if (hasBarcode) while (barcodes.pop()) {} };  // lets blocked readers finish
  t1k_params prm;
  t1k_params_default(&prm);
  prm.kmer_length = kmerLength;
  prm.hit_len_required = hitLenRequired;
  prm.ref_seq_similarity = similarity;
  prm.n_base_code = 0;  // this program's nucToNum maps 'N' to 0 (FastqExtractor.cpp:51-54), the genotyper's to -1
  t1k_ctx *ctx = nullptr;
  const int device = getenv("T1K_DEVICE") ? atoi(getenv("T1K_DEVICE")) : 0;  // as the genotyper executable picks its GPU
  if (t1k_ctx_create(device, &prm, &ctx) != T1K_OK) { fprintf(stderr, "fastq-extractor: cannot create the device context (k = %d)\n", kmerLength); drain(); return EXIT_FAILURE; }
  {
    std::string cat;
    std::vector<uint64_t> off(ref.size() + 1, 0);
    for (size_t i = 0; i < ref.size(); ++i) { cat += ref[i].seq; off[i + 1] = cat.size(); }
    if (t1k_ref_upload(ctx, cat.data(), off.data(), nullptr, (uint32_t)ref.size()) != T1K_OK) {
      fprintf(stderr, "fastq-extractor: %s\n", t1k_last_error(ctx));
      t1k_ctx_destroy(ctx);
      drain();
      return EXIT_FAILURE;
    }
  }

  lap("context + reference index upload");
  if (!hasBarcode && reads.mod == 1 && !started) {  // ordinary files: mapped, indexed in place, two device contexts (extractMapped)
    MappedRun mo;
    mo.r1s = r1s; mo.r1e = r1e; mo.r2s = r2s; mo.r2e = r2e; mo.threadCnt = threadCnt; mo.hasMate = hasMate; mo.dbg = dbg; mo.prefix = prefix;
    int rcFast = 0;
    uint64_t nf = 0, ng = 0;
    if (extractMapped(ctx, prm, device, mapped, mo, rcFast, nf, ng) == 1) {
      lap("read loop (mapped input, two contexts: index / gather / upload / test / write)");
      t1k_ctx_destroy(ctx);
      lap("context released");
      if (rcFast) return rcFast;
      if (dbg) fprintf(stderr, "[t1k] extractor: k=%d hitLenRequired=%d fragments=%llu kept=%llu\n", kmerLength, hitLenRequired, (unsigned long long)nf, (unsigned long long)ng);
      printLog("Finish extracting reads.");
      return 0;
    }
  }
  if (!started) { reads.start(); if (hasMate) mates.start(); if (hasBarcode) barcodes.start(); started = true; }
  FILE *fp1 = fopen((prefix + (hasMate ? "_1.fq" : ".fq")).c_str(), "w");
  FILE *fp2 = hasMate ? fopen((prefix + "_2.fq").c_str(), "w") : nullptr;
  FILE *fpBc = hasBarcode ? fopen((prefix + "_bc.fa").c_str(), "w") : nullptr;
  if (!fp1 || (hasMate && !fp2) || (hasBarcode && !fpBc)) { fprintf(stderr, "Cannot open the output files.\n"); t1k_ctx_destroy(ctx); drain(); return EXIT_FAILURE; }

  int rc = 0;
  std::string seqCat, out1, out2, outBc;
  std::vector<uint64_t> offs;
  std::vector<uint8_t> good;
  uint64_t nFragments = 0, nGood = 0;
  while (true) {
    auto c1 = reads.pop();
    std::unique_ptr<EndChunk> c2, cb;
    if (hasMate) c2 = mates.pop();
    if (hasBarcode) cb = barcodes.pop();
    const size_t n1 = c1 ? c1->n() : 0;
    if (hasMate && (c2 ? c2->n() : 0) != n1) { fprintf(stderr, "The two mate-pair read files have different number of reads.\n"); rc = 1; break; }
    if (hasBarcode && (cb ? cb->n() : 0) != n1) { fprintf(stderr, "Read file and barcode have different number of reads.\n"); rc = 1; break; }
    if (!c1) break;
    if (n1 == 0) continue;
    const uint32_t epf = hasMate ? 2 : 1;
    seqCat.clear(); offs.clear(); offs.push_back(0);
    for (size_t i = 0; i < n1; ++i) {
      seqCat.append(c1->seq(i), c1->seqLen[i]); offs.push_back(seqCat.size());
      if (hasMate) { seqCat.append(c2->seq(i), c2->seqLen[i]); offs.push_back(seqCat.size()); }
    }
    good.assign(n1, 0);
    if (t1k_reads_upload(ctx, seqCat.data(), offs.data(), nullptr, (uint32_t)(n1 * epf)) != T1K_OK || t1k_extract_batch(ctx, epf, good.data(), nullptr) != T1K_OK) {
      fprintf(stderr, "fastq-extractor: %s\n", t1k_last_error(ctx));
      rc = 1;
      break;
    }
    out1.clear(); out2.clear(); outBc.clear();
    for (size_t i = 0; i < n1; ++i) {
      if (!good[i]) continue;
      ++nGood;
      // the single-thread loop prints ReadFiles::Next()'s id (a trailing /1 or /2 removed, ReadFiles.hpp:185-189), the batch loop of
      // -t > 1 prints NextWithBuffer()'s raw name (FastqExtractor.cpp:446-476 vs 529-545)
      size_t nl = c1->nameLen[i];
      const char *nm = c1->name(i);
      if (threadCnt == 1 && nl >= 2 && nm[nl - 2] == '/' && (nm[nl - 1] == '1' || nm[nl - 1] == '2')) nl -= 2;
      outputSeq(out1, nm, nl, c1->seq(i), c1->qual(i), c1->seqLen[i], r1s, r1e);
      if (hasMate) outputSeq(out2, nm, nl, c2->seq(i), c2->qual(i), c2->seqLen[i], r2s, r2e);
      if (hasBarcode) {  // OutputBarcode (FastqExtractor.cpp:157-204)
        outBc.push_back('>'); outBc.append(nm, nl); outBc.push_back('\n');
        const size_t bl = cb->seqLen[i];
        if (bl == 0) outBc.append("missing_barcode\n");
        else {
          std::string b = formatBarcode(cb->seq(i), bl, barcodeStart, barcodeEnd, barcodeRevComp);
          // the corrector looks the changed position up in the quality string of the raw barcode read, unshifted and unreversed
          if (hasWhitelist && corrector.correct(b, cb->qual(i)) < 0) outBc.append("missing_barcode\n");
          else { outBc.append(b); outBc.push_back('\n'); }
        }
      }
    }
    fwrite(out1.data(), 1, out1.size(), fp1);
    if (fp2) fwrite(out2.data(), 1, out2.size(), fp2);
    if (fpBc) fwrite(outBc.data(), 1, outBc.size(), fpBc);
    nFragments += n1;
  }
  lap("read loop (parse / upload / test / write)");
  if (reads.failed || mates.failed || barcodes.failed) { fprintf(stderr, "Cannot open a read file.\n"); rc = 1; }
  // on an error the reader threads may still be blocked on a full queue: drain them
  if (rc) drain();
  fclose(fp1);
  if (fp2) fclose(fp2);
  if (fpBc) fclose(fpBc);
  if (rc != 0) {
    // a failed run (e.g. a read longer than this build's max_read_len) must not leave truncated candidate files behind for the next stage
    remove((prefix + (hasMate ? "_1.fq" : ".fq")).c_str());
    if (hasMate) remove((prefix + "_2.fq").c_str());
    if (hasBarcode) remove((prefix + "_bc.fa").c_str());
  }
  t1k_ctx_destroy(ctx);
  if (rc) return rc;
  if (dbg) fprintf(stderr, "[t1k] extractor: k=%d hitLenRequired=%d fragments=%llu kept=%llu\n", kmerLength, hitLenRequired, (unsigned long long)nFragments, (unsigned long long)nGood);
  printLog("Finish extracting reads.");
  return 0;
}
