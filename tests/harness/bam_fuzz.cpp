// tests/harness/bam_fuzz.cpp -- the product's BAM reader and the whole of bam-extractor's host side (t1k_amd/csrc/host/bamextract.cpp) on
// damaged input, for a build with -fsanitize=address,undefined:   bam_fuzz in.bam coord.fa workdir variants seed [args of bam-extractor]
// A BAM file is untrusted input.  Every variant is the file with (a) fields of its records / header changed in the INFLATED stream (length
// words, name lengths, CIGAR counts and lengths, sequence lengths, contig numbers, flags, tag types, string terminators, random bytes), the
// stream cut or a piece removed, wrapped again in BGZF blocks of a random size, or (b) bytes of the COMPRESSED container changed (block
// headers, the BC field, ISIZE).  t1k_bam_extractor_main runs on each: it may refuse the file or write its outputs, it may not read outside
// its buffers (the sanitizers end the process) and it must return.  The device stage is a stand-in that calls every other read a hit (the
// GPU kernels are not what is tested here): the second pass over the file runs too.  Prints "<variants> <refused> <accepted>".
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "../../t1k_amd/csrc/host/bamextract.cpp"

// ---- stand-ins for the device stage of the C ABI (include/t1k_gpu.h) ---------------------------------------------------------------
struct t1k_ctx { uint32_t n = 0; };
extern "C" {
void t1k_params_default(t1k_params *p) { memset(p, 0, sizeof(*p)); }
int t1k_device_count(void) { return 1; }
int t1k_ctx_create(int, const t1k_params *, t1k_ctx **out) { *out = new t1k_ctx; return T1K_OK; }
void t1k_ctx_destroy(t1k_ctx *c) { delete c; }
const char *t1k_last_error(const t1k_ctx *) { return "stand-in"; }
int t1k_ref_upload(t1k_ctx *, const char *, const uint64_t *, const uint8_t *, uint32_t) { return T1K_OK; }
int t1k_reads_upload(t1k_ctx *c, const char *seqs, const uint64_t *off, const uint32_t *, uint32_t n) {
  unsigned long sum = 0;
  for (uint64_t i = 0; i < off[n]; ++i) sum += (unsigned char)seqs[i];  // every byte the caller says it hands over is read
  c->n = n + (uint32_t)(sum & 0);
  return T1K_OK;
}
int t1k_extract_batch(t1k_ctx *c, uint32_t, uint8_t *good, uint64_t *) { for (uint32_t i = 0; i < c->n; ++i) good[i] = (uint8_t)(i & 1); return T1K_OK; }
}

static std::vector<uint8_t> slurp(const char *path) {
  std::vector<uint8_t> v;
  FILE *fp = fopen(path, "rb");
  if (!fp) return v;
  uint8_t buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, fp)) > 0) v.insert(v.end(), buf, buf + n);
  fclose(fp);
  return v;
}
static std::vector<uint8_t> unbgzf(const std::vector<uint8_t> &f) {  // (the undamaged file: plain zlib, member by member)
  std::vector<uint8_t> out;
  size_t pos = 0;
  while (pos + 18 <= f.size()) {
    const size_t bsize = (size_t)(f[pos + 16] | (f[pos + 17] << 8)) + 1;
    const size_t isize = (size_t)f[pos + bsize - 4] | ((size_t)f[pos + bsize - 3] << 8) | ((size_t)f[pos + bsize - 2] << 16) | ((size_t)f[pos + bsize - 1] << 24);
    const size_t at = out.size();
    out.resize(at + isize);
    z_stream zs; memset(&zs, 0, sizeof zs);
    inflateInit2(&zs, -15);
    zs.next_in = (Bytef *)f.data() + pos + 18; zs.avail_in = (uInt)(bsize - 26);
    zs.next_out = out.data() + at; zs.avail_out = (uInt)isize;
    inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    pos += bsize;
  }
  return out;
}
static std::vector<uint8_t> bgzf(const std::vector<uint8_t> &raw, size_t block, bool eofBlock) {
  std::vector<uint8_t> out;
  auto one = [&](const uint8_t *p, size_t n) {
    std::vector<uint8_t> comp(n + n / 8 + 64);
    z_stream zs; memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    zs.next_in = (Bytef *)p; zs.avail_in = (uInt)n; zs.next_out = comp.data(); zs.avail_out = (uInt)comp.size();
    deflate(&zs, Z_FINISH);
    const size_t c = comp.size() - zs.avail_out;
    deflateEnd(&zs);
    const size_t bsize = c + 26;
    const uint8_t head[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, (uint8_t)((bsize - 1) & 255), (uint8_t)((bsize - 1) >> 8)};
    out.insert(out.end(), head, head + 18);
    out.insert(out.end(), comp.begin(), comp.begin() + (long)c);
    const uint32_t crc = (uint32_t)crc32(crc32(0, nullptr, 0), p, (uInt)n);
    for (int i = 0; i < 4; ++i) out.push_back((uint8_t)(crc >> (8 * i)));
    for (int i = 0; i < 4; ++i) out.push_back((uint8_t)((uint32_t)n >> (8 * i)));
  };
  for (size_t i = 0; i < raw.size(); i += block) one(raw.data() + i, std::min(block, raw.size() - i));
  if (eofBlock) one(raw.data(), 0);
  return out;
}

int main(int argc, char **argv) {
  if (argc < 6) return 2;
  const std::vector<uint8_t> file = slurp(argv[1]);
  const std::string fa = argv[2], dir = argv[3];
  const int variants = atoi(argv[4]);
  std::mt19937_64 rnd((uint64_t)atoll(argv[5]));
  const std::vector<uint8_t> raw = unbgzf(file);
  if (raw.size() < 100 || memcmp(raw.data(), "BAM\1", 4) != 0) { printf("ERROR the undamaged file\n"); return 1; }
  // where the records start, and every record's offset
  auto u32 = [&](size_t at) { return (uint32_t)raw[at] | ((uint32_t)raw[at + 1] << 8) | ((uint32_t)raw[at + 2] << 16) | ((uint32_t)raw[at + 3] << 24); };
  size_t at = 8 + u32(4);
  const uint32_t nRef = u32(at);
  at += 4;
  for (uint32_t i = 0; i < nRef; ++i) at += 4 + u32(at) + 4;
  const size_t firstRec = at;
  std::vector<size_t> recs;
  while (at + 4 <= raw.size()) { recs.push_back(at); at += 4 + u32(at); }
  if (recs.size() < 10 || at != raw.size()) { printf("ERROR the undamaged file's records\n"); return 1; }
  const std::string bam = dir + "/v.bam", prefix = dir + "/out";
  int refused = 0, accepted = 0;
  for (int v = 0; v < variants; ++v) {
    std::vector<uint8_t> out;
    const int kind = (int)(rnd() % 10);
    if (kind < 8) {
      std::vector<uint8_t> x = raw;
      auto put32 = [&](size_t a, uint32_t val) { for (int i = 0; i < 4 && a + i < x.size(); ++i) x[a + i] = (uint8_t)(val >> (8 * i)); };
      const uint32_t odd[10] = {0, 1, 31, 32, 33, 0x7fffffffu, 0x80000000u, 0xffffffffu, 0xfffffff0u, (uint32_t)rnd()};
      const int edits = 1 + (int)(rnd() % 3);
      for (int e = 0; e < edits; ++e) {
        const size_t r = recs[rnd() % recs.size()];
        const uint32_t len = u32(r), lName = raw[r + 12], nCig = raw[r + 16] | (raw[r + 17] << 8);
        const size_t aux = r + 36 + lName + 4 * nCig + (u32(r + 20) + 1) / 2 + u32(r + 20);
        switch ((int)(rnd() % 14)) {
          case 0: put32(r, odd[rnd() % 10]); break;                                               // the record's length word
          case 1: x[r + 12] = (uint8_t)(rnd() % 3 ? rnd() : 0); break;                            // l_read_name
          case 2: x[r + 16] = (uint8_t)rnd(); x[r + 17] = (uint8_t)rnd(); break;                  // n_cigar_op
          case 3: put32(r + 20, odd[rnd() % 10]); break;                                          // l_seq
          case 4: put32(r + 4, rnd() % 2 ? (uint32_t)(rnd() % 16) : odd[rnd() % 10]); break;      // contig number
          case 5: put32(r + 8, odd[rnd() % 10]); break;                                           // position
          case 6: x[r + 18] = (uint8_t)rnd(); x[r + 19] = (uint8_t)rnd(); break;                  // flag
          case 7: if (nCig) put32(r + 36 + lName + 4 * (rnd() % nCig), (uint32_t)rnd()); break;   // a CIGAR operation: any length, any code
          case 8: x[r + 36 + lName - 1] = 'x'; break;                                             // the name's terminator
          case 9: for (size_t i = aux; i < r + 4 + len; ++i) if (x[i] == 0 && rnd() % 2) x[i] = 'Z'; break;     // the tags' string terminators
          case 10: if (aux + 3 <= r + 4 + len) x[aux + 2] = (uint8_t)"ZHBidsSAcCIfq"[rnd() % 13]; break;       // a tag's type
          case 11: if (aux + 8 <= r + 4 + len) { x[aux + 2] = 'B'; x[aux + 3] = (uint8_t)"cCsSiIfz"[rnd() % 8]; put32(aux + 4, odd[rnd() % 10]); } break;  // an array tag of any length
          case 12: { const size_t a = r + rnd() % (4 + len); const size_t k = 1 + rnd() % 8; for (size_t i = a; i < a + k && i < x.size(); ++i) x[i] = (uint8_t)rnd(); } break;
          default: {                                                                               // the header: text length, contig count, a name's length
            const size_t w = rnd() % 3;
            if (w == 0) put32(4, odd[rnd() % 10]); else if (w == 1) put32(8 + u32(4), odd[rnd() % 10]); else put32(8 + u32(4) + 4, odd[rnd() % 10]);
          }
        }
      }
      const int cut = (int)(rnd() % 8);
      if (cut == 0) x.resize(firstRec + rnd() % (x.size() - firstRec));                                            // the stream ends inside a record
      else if (cut == 1) { const size_t a = firstRec + rnd() % (x.size() - firstRec), b = a + rnd() % std::min<size_t>(x.size() - a, 300); x.erase(x.begin() + (long)a, x.begin() + (long)b); }
      else if (cut == 2) x.resize(rnd() % firstRec);                                                               // ... inside the header
      out = bgzf(x, 200 + rnd() % 65000, rnd() % 2);
    } else {
      out = file;
      const size_t k = 1 + rnd() % 6;
      for (size_t i = 0; i < k; ++i) {
        const size_t a = rnd() % out.size();
        if (kind == 8) out[a] = (uint8_t)rnd();
        else {  // aim at the block headers: the first 18 bytes and the last 8 of some block
          size_t pos = 0, hops = rnd() % 8;
          while (hops-- && pos + 18 < out.size()) { const size_t b = (size_t)(out[pos + 16] | (out[pos + 17] << 8)) + 1; if (pos + b + 18 >= out.size()) break; pos += b; }
          const size_t b = (size_t)(out[pos + 16] | (out[pos + 17] << 8)) + 1;
          const size_t t = rnd() % 26;
          const size_t where = t < 18 ? pos + t : pos + b - 8 + (t - 18);
          if (where < out.size()) out[where] = (uint8_t)rnd();
        }
      }
      if (rnd() % 4 == 0) out.resize(1 + rnd() % out.size());
    }
    FILE *fp = fopen(bam.c_str(), "wb");
    if (!fp) return 2;
    if (!out.empty()) fwrite(out.data(), 1, out.size(), fp);
    fclose(fp);
    std::vector<std::string> a = {"bam-extractor", "-b", bam, "-f", fa, "-o", prefix};
    for (int i = 6; i < argc; ++i) a.push_back(argv[i]);
    std::vector<char *> av;
    for (auto &s : a) av.push_back((char *)s.c_str());
    const int rc = t1k_bam_extractor_main((int)av.size(), av.data());
    if (rc) ++refused; else ++accepted;
  }
  printf("%d %d %d\n", variants, refused, accepted);
  return 0;
}
