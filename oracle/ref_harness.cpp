// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.  Tiny driver around the REFERENCE's own AlignAlgo::GlobalAlignment
// (compiled with -I/root/reference by oracle/Makefile `ref`, output oracle/_ref/ga_harness; no reference source is copied):
// reads "T P" pairs from stdin and prints "score ops" so golden vectors can be captured from the reference itself.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
char nucToNum[26] = {0, -1, 1, -1, -1, -1, 2, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 3, -1, -1, -1, -1, -1, -1};
char numToNuc[4] = {'A', 'C', 'G', 'T'};
#include "AlignAlgo.hpp"

int main() {
  static char t[1 << 16], p[1 << 16];
  std::vector<char> align(1 << 17);
  while (scanf("%65535s %65535s", t, p) == 2) {
    int lt = (int)strlen(t), lp = (int)strlen(p);
    if (!strcmp(t, "-")) lt = 0;
    if (!strcmp(p, "-")) lp = 0;
    int s = AlignAlgo::GlobalAlignment(t, lt, p, lp, align.data());
    std::string ops;
    for (int i = 0; align[i] != -1; ++i) ops += (char)('0' + align[i]);
    printf("%d %s\n", s, ops.empty() ? "-" : ops.c_str());
  }
  return 0;
}
