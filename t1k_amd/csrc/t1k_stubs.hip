// temporary: entry points not implemented yet
#include "t1k_dev.h"
extern "C" {
int t1k_pair_batch(t1k_ctx *ctx, const uint32_t *, const uint32_t *, const uint8_t *, uint32_t) { return t1k_fail(ctx, T1K_ERR_STATE, "not implemented"); }
int t1k_rows_download(t1k_ctx *ctx, uint32_t *, uint8_t *, t1k_row_entry *, uint64_t, uint64_t *) { return t1k_fail(ctx, T1K_ERR_STATE, "not implemented"); }
int t1k_align_batch(t1k_ctx *ctx, const char *, const uint32_t *, const uint32_t *, const char *, const uint32_t *, const uint32_t *, uint32_t, int32_t *, int32_t *, int32_t *, int32_t *, int8_t *, const uint32_t *, uint32_t *) { return t1k_fail(ctx, T1K_ERR_STATE, "not implemented"); }
int t1k_align_count_batch(t1k_ctx *ctx, const char *, const uint32_t *, const char *, const uint32_t *, const uint32_t *, uint32_t, int32_t *) { return t1k_fail(ctx, T1K_ERR_STATE, "not implemented"); }
int t1k_em_setup(t1k_ctx *ctx, const uint64_t *, const uint32_t *, const double *, const int32_t *, uint32_t, uint32_t, t1k_allreduce_fn, void *) { return t1k_fail(ctx, T1K_ERR_STATE, "not implemented"); }
int t1k_em_update(t1k_ctx *ctx, const double *, double *, double *, double *) { return t1k_fail(ctx, T1K_ERR_STATE, "not implemented"); }
int t1k_genotyper_main(int, char **) { return 1; }
void t1k_job_params_default(t1k_job_params *) {}
int t1k_job_create(const t1k_job_params *, const char *, t1k_job **) { return T1K_ERR_STATE; }
void t1k_job_destroy(t1k_job *) {}
const char *t1k_job_last_error(const t1k_job *) { return "not implemented"; }
int t1k_job_load_reads(t1k_job *, const char *, const char *, const char *) { return T1K_ERR_STATE; }
int t1k_job_set_reads(t1k_job *, const char *, const uint64_t *, const char *, const uint64_t *, uint32_t) { return T1K_ERR_STATE; }
int t1k_job_stage_reads(t1k_job *) { return T1K_ERR_STATE; }
int t1k_job_run(t1k_job *) { return T1K_ERR_STATE; }
int t1k_job_write_outputs(t1k_job *, const char *) { return T1K_ERR_STATE; }
int t1k_job_genotype_text(t1k_job *, char *, uint64_t, uint64_t *) { return T1K_ERR_STATE; }
int t1k_job_counts(t1k_job *, uint64_t *, uint64_t *, uint64_t *, uint64_t *, int32_t *) { return T1K_ERR_STATE; }
int t1k_job_stats(t1k_job *, t1k_stats *) { return T1K_ERR_STATE; }
t1k_ctx *t1k_job_ctx(t1k_job *) { return nullptr; }
int t1k_job_set_allreduce(t1k_job *, t1k_allreduce_fn, void *) { return T1K_ERR_STATE; }
}
