// tests/harness/inflate_fuzz.cpp -- the product's gzip decoder (t1k_amd/csrc/host/inflate.cpp) on damaged input, for a build with
// -fsanitize=address,undefined:   inflate_fuzz in.gz variants seed
// Every variant is a copy of the file with a few bytes changed, a piece cut out or the end cut off, in a heap block of exactly its size,
// decoded into a heap block of a capacity drawn around the true size: the decoder may refuse it or decode it, it may not touch a byte
// outside the two blocks.  Prints "<variants> <refused> <decoded>".
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "../../t1k_amd/csrc/host/t1k_host.h"
int main(int argc, char **argv) {
  if (argc < 4) return 2;
  FILE *fp = fopen(argv[1], "rb");
  if (!fp) return 2;
  std::vector<uint8_t> file;
  uint8_t buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, fp)) > 0) file.insert(file.end(), buf, buf + n);
  fclose(fp);
  const int variants = atoi(argv[2]);
  std::mt19937_64 rnd((uint64_t)atoll(argv[3]));
  size_t trueLen = 0;
  {
    std::vector<uint8_t> dst(file.size() * 1200 + (1 << 20));
    t1k::GzProgress pg; std::string err; uint32_t crc = 0; size_t members = 0;
    if (t1k::gzInflateAll(file.data(), file.size(), dst.data(), dst.size(), &pg, &trueLen, &crc, &members, err)) { printf("ERROR the undamaged file: %s\n", err.c_str()); return 1; }
  }
  int refused = 0, decoded = 0;
  for (int v = 0; v < variants; ++v) {
    std::vector<uint8_t> x = file;
    const int kind = (int)(rnd() % 5);
    if (kind == 0) { const int k = 1 + (int)(rnd() % 4); for (int i = 0; i < k; ++i) x[rnd() % x.size()] ^= (uint8_t)(1u << (rnd() % 8)); }
    else if (kind == 1) { const int k = 1 + (int)(rnd() % 16); for (int i = 0; i < k; ++i) x[rnd() % x.size()] = (uint8_t)rnd(); }
    else if (kind == 2) x.resize(1 + rnd() % x.size());                                                   // cut off
    else if (kind == 3) { const size_t a = rnd() % x.size(), b = a + rnd() % (x.size() - a); x.erase(x.begin() + a, x.begin() + b); if (x.empty()) x.push_back(0x1f); }  // a piece cut out
    else { const size_t a = 10 + rnd() % 64; for (size_t i = a; i < x.size() && i < a + 8; ++i) x[i] = (uint8_t)rnd(); }   // the first block's header and code lengths
    uint8_t *src = (uint8_t *)malloc(x.size());
    memcpy(src, x.data(), x.size());
    const size_t caps[4] = {trueLen, trueLen / 2 + rnd() % (trueLen / 2 + 1), trueLen + rnd() % 4096, (size_t)(rnd() % 1024)};
    const size_t cap = caps[rnd() % 4];
    uint8_t *dst = (uint8_t *)malloc(cap ? cap : 1);
    t1k::GzProgress pg; std::string err; uint32_t crc = 0; size_t members = 0, outLen = 0;
    const int rc = t1k::gzInflateAll(src, x.size(), dst, cap, &pg, &outLen, &crc, &members, err);
    if (rc) ++refused; else { ++decoded; if (outLen > cap) { printf("ERROR %zu bytes reported in a block of %zu\n", outLen, cap); return 1; } }
    if (pg.produced.load() > cap) { printf("ERROR progress beyond the block\n"); return 1; }
    free(src); free(dst);
  }
  printf("%d %d %d\n", variants, refused, decoded);
  return 0;
}
