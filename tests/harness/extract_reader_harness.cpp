// tests/harness/extract_reader_harness.cpp -- TEST INFRASTRUCTURE (CPU): the streaming record reader of fastq-extractor's host side
// (t1k_amd/csrc/host/extract.cpp: RecordReader behind a Stream, the in-place four-line path and the general path taking turns over one
// buffer), alone:   extract_reader_harness <buffer bytes> files...   prints "id<TAB>seq" per record, the id as ReadFiles::Next leaves it
// (a trailing /1 or /2 dropped), so that the output can be compared with oracle/_ref/reads_harness on odd and damaged files.  The device
// stage of the C ABI is not used; the stand-ins below only satisfy the linker.
#include "../../t1k_amd/csrc/host/extract.cpp"

struct t1k_ctx {};
extern "C" {
void t1k_params_default(t1k_params *p) { memset(p, 0, sizeof(*p)); }
int t1k_device_count(void) { return 0; }
int t1k_ctx_create(int, const t1k_params *, t1k_ctx **) { return T1K_ERR_DEVICE; }
void t1k_ctx_destroy(t1k_ctx *) {}
const char *t1k_last_error(const t1k_ctx *) { return "stand-in"; }
int t1k_ref_upload(t1k_ctx *, const char *, const uint64_t *, const uint8_t *, uint32_t) { return T1K_ERR_DEVICE; }
int t1k_ref_share(t1k_ctx *, const t1k_ctx *) { return T1K_ERR_DEVICE; }
int t1k_reads_upload(t1k_ctx *, const char *, const uint64_t *, const uint32_t *, uint32_t) { return T1K_ERR_DEVICE; }
int t1k_reads_upload_begin(t1k_ctx *, uint32_t, uint64_t, int) { return T1K_ERR_DEVICE; }
int t1k_reads_upload_piece(t1k_ctx *, int, const void *, uint64_t, uint64_t, int) { return T1K_ERR_DEVICE; }
int t1k_reads_upload_wait(t1k_ctx *, int) { return T1K_ERR_DEVICE; }
int t1k_reads_upload_end(t1k_ctx *) { return T1K_ERR_DEVICE; }
int t1k_extract_batch(t1k_ctx *, uint32_t, uint8_t *, uint64_t *) { return T1K_ERR_DEVICE; }
void *t1k_pinned_alloc(uint64_t n) { return malloc(n); }
void t1k_pinned_free(void *p) { free(p); }
}

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  Stream s;
  s.bufBytes = (size_t)atoll(argv[1]);
  for (int i = 2; i < argc; ++i) s.files.push_back(argv[i]);
  s.chunkRecords = 7;
  s.start();
  while (auto c = s.pop()) {
    for (size_t i = 0; i < c->n(); ++i) {
      size_t nl = c->nameLen[i];
      const char *nm = c->name(i);
      nl = strnlen(nm, nl);  // (the reference copies the name as a C string)
      if (nl >= 2 && nm[nl - 2] == '/' && (nm[nl - 1] == '1' || nm[nl - 1] == '2')) nl -= 2;
      fwrite(nm, 1, nl, stdout);
      fputc('\t', stdout);
      fwrite(c->seq(i), 1, strnlen(c->seq(i), c->seqLen[i]), stdout);
      fputc('\n', stdout);
    }
  }
  return s.failed ? 1 : 0;
}
