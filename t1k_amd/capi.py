"""ctypes binding of include/t1k_gpu.h (libt1k_gpu.so, built in-tree by __graft_entry__.build()).

There is deliberately no fallback: if the HIP library is missing, or no GPU is visible when a context is created,
an exception is raised.  Nothing here touches oracle/.
"""
import ctypes as C
import gzip
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    return os.environ.get("T1K_GPU_LIB") or os.path.join(_HERE, "lib", "libt1k_gpu.so")


class T1kError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [("kmer_length", C.c_int32), ("radius", C.c_int32), ("hit_len_required", C.c_int32),
                ("ref_seq_similarity", C.c_double), ("relax_intron_align", C.c_int32), ("max_assign_cnt", C.c_int32),
                ("max_read_len", C.c_int32), ("workgroups", C.c_int32), ("group_cap", C.c_int64),
                ("cand_cap", C.c_int64), ("ovl_cap", C.c_int64), ("row_cap", C.c_int64), ("n_base_code", C.c_int32), ("store_chunk_mb", C.c_int32)]


class JobParams(C.Structure):
    _fields_ = [("dev", Params), ("filter_frac", C.c_double), ("filter_cov", C.c_double),
                ("cross_gene_rate", C.c_double), ("squarem_min_alpha", C.c_double), ("allele_digit_units", C.c_int32),
                ("allele_delimiter", C.c_char), ("threads", C.c_int32), ("device", C.c_int32),
                ("output_read_assignment", C.c_int32), ("batch_fragments", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("read_ends", "lookups", "postings", "hits", "groups", "candidates", "extended",
                                          "near_best", "dp_calls", "rows", "batches")] + \
               [(n, C.c_double) for n in ("ms_seed", "ms_chain", "ms_extend", "ms_select", "ms_fullalign", "ms_pair", "ms_em", "ms_total")] + \
               [(n, C.c_uint64) for n in ("read_ends_total", "pair_overlaps", "dp_cells")] + \
               [(n, C.c_double) for n in ("ms_load", "ms_device", "ms_coalesce", "ms_write")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


OVERLAP_DTYPE = np.dtype([("seq_idx", "<i4"), ("read_start", "<i4"), ("read_end", "<i4"), ("seq_start", "<i4"),
                          ("seq_end", "<i4"), ("strand", "<i4"), ("match_cnt", "<i4"), ("left_clip", "<i4"),
                          ("right_clip", "<i4"), ("relaxed_match_cnt", "<i4"), ("similarity", "<f8")])
ROW_DTYPE = np.dtype([("allele_idx", "<i4"), ("start", "<i4"), ("end", "<i4"), ("weight", "<f4"), ("qual", "<f4"),
                      ("adjust_weight", "<f4")])
assert OVERLAP_DTYPE.itemsize == 48 and ROW_DTYPE.itemsize == 24
# t1k_frag_assignment / t1k_variant (novel-variant calling of the analyzer stage, host code)
FRAG_ASG_DTYPE = np.dtype([("allele_idx", "<i4"), ("has_mate_pair", "<i4"), ("o1_from_r2", "<i4"), ("_pad", "<i4"), ("o1", OVERLAP_DTYPE), ("o2", OVERLAP_DTYPE),
                           ("ops1", "<u8"), ("ops2", "<u8"), ("n_ops1", "<u4"), ("n_ops2", "<u4")])
VARIANT_DTYPE = np.dtype([("allele_idx", "<i4"), ("ref_pos", "<i4"), ("exon_pos", "<i4"), ("ref", "S1"), ("var", "S1"), ("_pad", "S2"), ("qual", "<i4"), ("group", "<i4"),
                          ("output_group", "<i4"), ("_pad2", "<i4"), ("var_support", "<f8"), ("all_support", "<f8"), ("var_uniq_support", "<f8")])
assert FRAG_ASG_DTYPE.itemsize == 136 and VARIANT_DTYPE.itemsize == 56

ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_void_p)

_lib = None


def lib():
    """Load libt1k_gpu.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise T1kError("%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` first" % p)
    L = C.CDLL(p)
    vp, u64p, u32p, u8p, i32p = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
    L.t1k_params_default.argtypes = [C.POINTER(Params)]
    L.t1k_ctx_create.argtypes = [C.c_int, C.POINTER(Params), C.POINTER(vp)]
    L.t1k_ctx_destroy.argtypes = [vp]
    L.t1k_last_error.argtypes = [vp]
    L.t1k_last_error.restype = C.c_char_p
    L.t1k_device_count.restype = C.c_int
    L.t1k_ref_upload.argtypes = [vp, C.c_char_p, vp, vp, C.c_uint32]
    L.t1k_reads_upload.argtypes = [vp, C.c_char_p, vp, vp, C.c_uint32]
    L.t1k_assign_batch.argtypes = [vp]
    L.t1k_assign_range.argtypes = [vp, C.c_uint64, C.c_uint32]
    L.t1k_overlaps_download.argtypes = [vp, vp, vp, C.c_uint64, u64p]
    L.t1k_pair_batch.argtypes = [vp, vp, vp, vp, C.c_uint32]
    L.t1k_rows_download.argtypes = [vp, vp, vp, vp, C.c_uint64, u64p]
    L.t1k_coverage_get.argtypes = [vp, vp, C.c_uint64]
    L.t1k_coverage_reset.argtypes = [vp]
    L.t1k_missing_coverage.argtypes = [vp, vp]
    L.t1k_coverage_absorb.argtypes = [vp, vp]
    L.t1k_align_batch.argtypes = [vp, C.c_char_p, vp, vp, C.c_char_p, vp, vp, C.c_uint32, vp, vp, vp, vp, vp, vp, vp]
    L.t1k_align_count_batch.argtypes = [vp, C.c_char_p, vp, C.c_char_p, vp, vp, C.c_uint32, vp]
    L.t1k_em_setup.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, C.c_uint32, ALLREDUCE_FN, vp]
    L.t1k_em_update.argtypes = [vp, vp, vp, vp, vp]
    L.t1k_extract_batch.argtypes = [vp, C.c_uint32, vp, vp]
    L.t1k_extractor_main.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
    L.t1k_stats_get.argtypes = [vp, C.POINTER(Stats)]
    L.t1k_genotyper_main.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
    L.t1k_job_params_default.argtypes = [C.POINTER(JobParams)]
    L.t1k_job_create.argtypes = [C.POINTER(JobParams), C.c_char_p, C.POINTER(vp)]
    L.t1k_job_destroy.argtypes = [vp]
    L.t1k_job_last_error.argtypes = [vp]
    L.t1k_job_last_error.restype = C.c_char_p
    L.t1k_job_load_reads.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p]
    L.t1k_job_set_reads.argtypes = [vp, C.c_char_p, vp, C.c_char_p, vp, C.c_uint32]
    L.t1k_job_run.argtypes = [vp]
    L.t1k_job_write_outputs.argtypes = [vp, C.c_char_p]
    L.t1k_pool_release.restype = C.c_uint64
    L.t1k_pool_release.argtypes = []
    L.t1k_job_load_reads_multi.argtypes = [vp, C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_char_p), C.c_uint32, C.c_char_p]
    L.t1k_reads_open.argtypes = [C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_char_p), C.c_uint32, C.c_char_p, C.c_int, C.POINTER(vp)]
    L.t1k_reads_open_stream.argtypes = [C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_char_p), C.c_uint32, C.c_char_p, C.c_int, C.POINTER(vp)]
    L.t1k_reads_last_error.argtypes = [vp]
    L.t1k_reads_last_error.restype = C.c_char_p
    L.t1k_reads_fragments.argtypes = [vp, u64p]
    L.t1k_reads_close.argtypes = [vp]
    L.t1k_reads_close.restype = None
    L.t1k_job_attach_reads.argtypes = [vp, vp]
    L.t1k_job_genotype_text.argtypes = [vp, C.c_char_p, C.c_uint64, u64p]
    L.t1k_job_counts.argtypes = [vp, u64p, u64p, u64p, u64p, i32p]
    L.t1k_job_stats.argtypes = [vp, C.POINTER(Stats)]
    L.t1k_job_ctx.argtypes = [vp]
    L.t1k_job_ctx.restype = vp
    L.t1k_job_run_local.argtypes = [vp]
    L.t1k_job_finish.argtypes = [vp]
    L.t1k_job_groups_serialize.argtypes = [vp, vp, C.c_uint64, u64p]
    L.t1k_reads_dedupe.argtypes = [vp, vp, u32p]
    L.t1k_job_groups_merge.argtypes = [vp, vp, vp, C.c_uint32]
    L.t1k_job_coalesce_rows.argtypes = [vp, vp, vp, vp, C.c_uint32]
    L.t1k_variants_call.argtypes = [vp, vp, C.c_int32, C.c_uint32, vp, vp, vp, vp, vp, vp, vp, C.POINTER(vp)]
    L.t1k_fragment_details.argtypes = [vp, C.c_uint32, vp, C.c_uint32, C.c_int, vp, C.c_uint32, vp]
    L.t1k_variants_count.argtypes = [vp]
    L.t1k_variants_count.restype = C.c_uint32
    L.t1k_variants_get.argtypes = [vp, vp]
    L.t1k_variants_vcf.argtypes = [vp, vp, C.c_uint64, u64p]
    L.t1k_variants_adjust.argtypes = [vp, vp, C.c_uint32, vp, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, vp]
    L.t1k_variants_destroy.argtypes = [vp]
    L.t1k_job_set_shard.argtypes = [vp, C.c_int, C.c_int, vp]
    L.t1k_job_share_reads.argtypes = [vp, vp]
    L.t1k_job_ctx.restype = vp
    L.t1k_job_ctx.argtypes = [vp]
    L.t1k_comm_unique_id.argtypes = [vp]
    L.t1k_comm_init.argtypes = [vp, C.c_int, C.c_int, vp, vp, C.c_int, C.POINTER(vp)]
    L.t1k_comm_bind.argtypes = [vp, vp]
    L.t1k_comm_destroy.argtypes = [vp]
    L.t1k_comm_last_error.restype = C.c_char_p
    L.t1k_comm_last_error.argtypes = [vp]
    L.t1k_comm_is_rccl.argtypes = [vp]
    L.t1k_comm_group_create.restype = vp
    L.t1k_comm_group_create.argtypes = [C.c_int]
    L.t1k_comm_group_destroy.argtypes = [vp]
    L.t1k_coverage_device.argtypes = [vp, C.POINTER(vp), u64p]
    L.t1k_device_memory.argtypes = [C.c_int, u64p, u64p]
    L.t1k_ctx_set_coverage_mode.argtypes = [vp, C.c_int]
    L.t1k_readset_detach.argtypes = [vp, C.POINTER(vp)]
    L.t1k_readset_take_store.argtypes = [vp, vp, C.c_int]
    L.t1k_readset_bytes.argtypes = [vp]
    L.t1k_readset_bytes.restype = C.c_uint64
    L.t1k_readset_size.argtypes = [vp]
    L.t1k_readset_size.restype = C.c_uint32
    L.t1k_readset_destroy.argtypes = [vp]
    L.t1k_coverage_selected.argtypes = [vp, vp, vp, u64p]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _concat(seqs):
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if len(seqs):
        offs[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    blob = "".join(seqs).encode("ascii")
    return blob, offs


class Context:
    """One GPU context (t1k_ctx)."""

    def __init__(self, device=0, **kw):
        L = lib()
        p = Params()
        L.t1k_params_default(C.byref(p))
        for k, v in kw.items():
            setattr(p, k, v)
        self.params = p
        h = C.c_void_p()
        rc = L.t1k_ctx_create(device, C.byref(p), C.byref(h))
        if rc != 0:
            raise T1kError("t1k_ctx_create failed (%d): no usable GPU / bad parameters" % rc)
        self.h = h
        self.n_read_ends = 0
        self.allele_len = None

    def _check(self, rc, what):
        if rc != 0:
            raise T1kError("%s failed (%d): %s" % (what, rc, lib().t1k_last_error(self.h).decode()))

    def close(self):
        if self.h:
            lib().t1k_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ref_upload(self, seqs, exon_masks=None):
        blob, offs = _concat(seqs)
        ex = None
        if exon_masks is not None:
            ex = np.concatenate([np.asarray(m, dtype=np.uint8) for m in exon_masks]) if len(seqs) else np.zeros(0, np.uint8)
        self.allele_len = np.array([len(s) for s in seqs], dtype=np.int64)
        self._check(lib().t1k_ref_upload(self.h, blob, _ptr(offs), _ptr(ex), len(seqs)), "t1k_ref_upload")

    def reads_upload(self, seqs, weights=None):
        blob, offs = _concat(seqs)
        w = None if weights is None else np.asarray(weights, dtype=np.uint32)
        self.n_read_ends = len(seqs)
        self._check(lib().t1k_reads_upload(self.h, blob, _ptr(offs), _ptr(w), len(seqs)), "t1k_reads_upload")

    def reads_dedupe(self):
        """identical read-ends collapse (t1k_reads_dedupe): the context's read set becomes the distinct sequences; returns distinct_of[n_uploaded]"""
        out = np.zeros(self.n_read_ends, dtype=np.uint32)
        nd = C.c_uint32()
        self._check(lib().t1k_reads_dedupe(self.h, _ptr(out), C.byref(nd)), "t1k_reads_dedupe")
        self.n_uploaded, self.n_read_ends = self.n_read_ends, nd.value
        return out

    def assign(self):
        self._check(lib().t1k_assign_batch(self.h), "t1k_assign_batch")

    def extract(self, ends_per_fragment=1):
        """IsGoodCandidate of every fragment of the uploaded batch (t1k_extract_batch): (good[nFragments] uint8, stats dict)."""
        good = np.zeros(self.n_read_ends // ends_per_fragment, dtype=np.uint8)
        st = np.zeros(8, dtype=np.uint64)
        self._check(lib().t1k_extract_batch(self.h, ends_per_fragment, _ptr(good), _ptr(st)), "t1k_extract_batch")
        return good, dict(zip(("read_ends", "lookups", "postings", "read_ends_with_hits", "read_ends_chained", "screen_ns", "main_ns", "big_shape"), (int(x) for x in st)))

    def overlaps(self):
        n = self.n_read_ends
        counts = np.zeros(n, dtype=np.uint32)
        tot = C.c_uint64()
        self._check(lib().t1k_overlaps_download(self.h, _ptr(counts), None, 0, C.byref(tot)), "t1k_overlaps_download")
        out = np.zeros(tot.value, dtype=OVERLAP_DTYPE)
        self._check(lib().t1k_overlaps_download(self.h, _ptr(counts), _ptr(out), tot.value, C.byref(tot)), "t1k_overlaps_download")
        return counts, out

    def pair(self, end1, end2, has_n):
        e1 = np.asarray(end1, dtype=np.uint32)
        e2 = None if end2 is None else np.asarray(end2, dtype=np.uint32)
        hn = np.asarray(has_n, dtype=np.uint8)
        self.n_fragments = len(e1)
        self._check(lib().t1k_pair_batch(self.h, _ptr(e1), _ptr(e2), _ptr(hn), len(e1)), "t1k_pair_batch")

    def rows(self):
        n = self.n_fragments
        counts = np.zeros(n, dtype=np.uint32)
        assigned = np.zeros(n, dtype=np.uint8)
        tot = C.c_uint64()
        self._check(lib().t1k_rows_download(self.h, _ptr(counts), _ptr(assigned), None, 0, C.byref(tot)), "t1k_rows_download")
        out = np.zeros(tot.value, dtype=ROW_DTYPE)
        self._check(lib().t1k_rows_download(self.h, _ptr(counts), _ptr(assigned), _ptr(out), tot.value, C.byref(tot)), "t1k_rows_download")
        return counts, assigned, out

    def coverage(self):
        tot = int(self.allele_len.sum())
        out = np.zeros(tot, dtype=np.int32)
        self._check(lib().t1k_coverage_get(self.h, _ptr(out), tot), "t1k_coverage_get")
        return out

    def set_coverage_mode(self, deferred):
        """deferred: t1k_assign_range adds no per-base coverage (t1k_coverage_selected adds it later for the alleles that need it)"""
        self._check(lib().t1k_ctx_set_coverage_mode(self.h, 1 if deferred else 0), "t1k_ctx_set_coverage_mode")

    def detach_readset(self, slot=0):
        """the context's read set and the final overlap lists it holds change owner (t1k_readset_detach + t1k_readset_take_store)"""
        h = C.c_void_p()
        self._check(lib().t1k_readset_detach(self.h, C.byref(h)), "t1k_readset_detach")
        rs = Readset(h)
        self._check(lib().t1k_readset_take_store(rs.h, self.h, slot), "t1k_readset_take_store")
        return rs

    def coverage_selected(self, readset, selected):
        """adds the coverage of the read set's near-best overlaps on the alleles with selected[a] != 0; returns the records aligned"""
        sel = np.ascontiguousarray(selected, dtype=np.uint8)
        n = C.c_uint64()
        self._check(lib().t1k_coverage_selected(self.h, readset.h, _ptr(sel), C.byref(n)), "t1k_coverage_selected")
        return n.value

    def coverage_reset(self):
        self._check(lib().t1k_coverage_reset(self.h), "t1k_coverage_reset")

    def missing_coverage(self):
        """per allele: exon positions with coverage below max(1, 1 % of the allele's median exon coverage)"""
        out = np.zeros(len(self.allele_len), dtype=np.int32)
        self._check(lib().t1k_missing_coverage(self.h, _ptr(out)), "t1k_missing_coverage")
        return out

    def stats(self):
        s = Stats()
        lib().t1k_stats_get(self.h, C.byref(s))
        return s.as_dict()

    def align_batch(self, texts, pats, want_ops=True):
        n = len(texts)
        tb, toff = _concat(texts)
        pb, poff = _concat(pats)
        tlen = np.array([len(t) for t in texts], dtype=np.uint32)
        plen = np.array([len(p) for p in pats], dtype=np.uint32)
        toff32, poff32 = toff[:-1].astype(np.uint32), poff[:-1].astype(np.uint32)
        score = np.zeros(n, np.int32); nm = np.zeros(n, np.int32); nx = np.zeros(n, np.int32); ni = np.zeros(n, np.int32)
        ops_off = np.zeros(n, np.uint32)
        if n:
            ops_off[1:] = np.cumsum((tlen + plen + 2)[:-1], dtype=np.uint64).astype(np.uint32)
        ops = np.zeros(int((tlen + plen + 2).sum()) + 8, np.int8)
        nops = np.zeros(n, np.uint32)
        self._check(lib().t1k_align_batch(self.h, tb, _ptr(toff32), _ptr(tlen), pb, _ptr(poff32), _ptr(plen), n, _ptr(score), _ptr(nm),
                                          _ptr(nx), _ptr(ni), _ptr(ops) if want_ops else None, _ptr(ops_off), _ptr(nops)), "t1k_align_batch")
        op_list = [ops[ops_off[i]:ops_off[i] + nops[i]].copy() for i in range(n)] if want_ops else None
        return score, nm, nx, ni, op_list

    def align_count_batch(self, texts, pats):
        """production match-count routine (exact fast path + banded forward sweep), equal-length jobs"""
        n = len(texts)
        tb, toff = _concat(texts)
        pb, poff = _concat(pats)
        ln = np.array([len(t) for t in texts], dtype=np.uint32)
        out = np.zeros(n, np.int32)
        self._check(lib().t1k_align_count_batch(self.h, tb, _ptr(toff[:-1].astype(np.uint32)), pb, _ptr(poff[:-1].astype(np.uint32)), _ptr(ln), n,
                                                _ptr(out)), "t1k_align_count_batch")
        return out

    def em_setup(self, row_ptr, ec_idx, count, ec_len, allreduce=None):
        self._em_keep = (np.asarray(row_ptr, np.uint64), np.asarray(ec_idx, np.uint32), np.asarray(count, np.float64), np.asarray(ec_len, np.int32))
        rp, ei, ct, el = self._em_keep
        self._em_cb = ALLREDUCE_FN(allreduce) if allreduce else C.cast(None, ALLREDUCE_FN)
        self._em_nec = len(el)
        self._check(lib().t1k_em_setup(self.h, _ptr(rp), _ptr(ei), _ptr(ct), _ptr(el), len(ct), len(el), self._em_cb, None), "t1k_em_setup")

    def em_update(self, x0):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        x1 = np.zeros(self._em_nec, np.float64)
        n = np.zeros(self._em_nec, np.float64)
        diff = C.c_double()
        self._check(lib().t1k_em_update(self.h, _ptr(x0), _ptr(x1), _ptr(n), C.byref(diff)), "t1k_em_update")
        return x1, n, diff.value


class Readset:
    """A finished read set kept resident (t1k_readset): distinct read-ends + their final overlap lists."""

    def __init__(self, h):
        self.h = h

    def bytes(self):
        return int(lib().t1k_readset_bytes(self.h))

    def size(self):
        return int(lib().t1k_readset_size(self.h))

    def close(self):
        if self.h:
            lib().t1k_readset_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Reads:
    """The read files of a job, mapped and indexed on their own (t1k_reads_open): needs no GPU and no job, so a second thread can open
    them while Job(...) parses the reference and brings the contexts up (ctypes releases the GIL for both calls); Job.attach_reads
    takes the input over.  f1 / f2 / barcode as in Job.load_reads."""

    def __init__(self, f1, f2=None, barcode=None, threads=0, stream=False):
        """stream=True: t1k_reads_open_stream -- ordinary .gz files are handed to Job.run while they are still being inflated"""
        l1 = [f1] if isinstance(f1, str) else list(f1)
        l2 = [] if not f2 else ([f2] if isinstance(f2, str) else list(f2))
        a1 = (C.c_char_p * len(l1))(*[x.encode() for x in l1])
        a2 = (C.c_char_p * len(l2))(*[x.encode() for x in l2]) if l2 else None
        h = C.c_void_p()
        opener = lib().t1k_reads_open_stream if stream else lib().t1k_reads_open
        rc = opener(a1, len(l1), a2, len(l2), barcode.encode() if barcode else None, threads, C.byref(h))
        if rc != 0:
            msg = lib().t1k_reads_last_error(h).decode() if h else "bad arguments"
            if h:
                lib().t1k_reads_close(h)
            raise T1kError("t1k_reads_open failed (%d): %s" % (rc, msg))
        self.h = h

    def fragments(self):
        n = C.c_uint64()
        if lib().t1k_reads_fragments(self.h, C.byref(n)) != 0:
            raise T1kError("t1k_reads_fragments failed")
        return n.value

    def close(self):
        if self.h:
            lib().t1k_reads_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Job:
    """Whole-stage job (t1k_job): host C++ around the device stages; same behaviour as the `genotyper` executable."""

    def __init__(self, ref_fasta, **kw):
        L = lib()
        p = JobParams()
        L.t1k_job_params_default(C.byref(p))
        for k, v in kw.items():
            if hasattr(p.dev, k):
                setattr(p.dev, k, v)
            elif k == "allele_delimiter":
                p.allele_delimiter = v.encode() if isinstance(v, str) else v
            else:
                setattr(p, k, v)
        self.params = p
        h = C.c_void_p()
        rc = L.t1k_job_create(C.byref(p), ref_fasta.encode(), C.byref(h))
        if rc != 0:
            msg = L.t1k_job_last_error(h).decode() if h else "no GPU / cannot read reference"
            raise T1kError("t1k_job_create failed (%d): %s" % (rc, msg))
        self.h = h

    def _check(self, rc, what):
        if rc != 0:
            raise T1kError("%s failed (%d): %s" % (what, rc, lib().t1k_job_last_error(self.h).decode()))

    def close(self):
        if self.h:
            lib().t1k_job_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_reads(self, f1, f2=None, barcode=None):
        """f1 / f2: a path, or a list of paths read back to back (every -1 / -2 of the executable).  On a rank of a sharded job
        (set_shard came first) the call is collective and indexes this rank's fragments only."""
        l1 = [f1] if isinstance(f1, str) else list(f1)
        l2 = [] if not f2 else ([f2] if isinstance(f2, str) else list(f2))
        a1 = (C.c_char_p * len(l1))(*[x.encode() for x in l1])
        a2 = (C.c_char_p * len(l2))(*[x.encode() for x in l2]) if l2 else None
        self._check(lib().t1k_job_load_reads_multi(self.h, a1, len(l1), a2, len(l2), barcode.encode() if barcode else None), "t1k_job_load_reads")

    def attach_reads(self, reads):
        """take over a Reads object (consumed, also on failure): the same state load_reads leaves"""
        h, reads.h = reads.h, None
        self._check(lib().t1k_job_attach_reads(self.h, h), "t1k_job_attach_reads")

    def set_reads(self, seqs1, seqs2=None):
        b1, o1 = _concat(seqs1)
        b2, o2 = _concat(seqs2) if seqs2 is not None else (None, None)
        self._check(lib().t1k_job_set_reads(self.h, b1, _ptr(o1), b2, _ptr(o2), len(seqs1)), "t1k_job_set_reads")

    def run(self):
        self._check(lib().t1k_job_run(self.h), "t1k_job_run")

    def set_output_prefix(self, prefix):
        lib().t1k_job_set_output_prefix.argtypes = [C.c_void_p, C.c_char_p]
        self._check(lib().t1k_job_set_output_prefix(self.h, prefix.encode()), "t1k_job_set_output_prefix")

    def write_outputs(self, prefix):
        self._check(lib().t1k_job_write_outputs(self.h, prefix.encode()), "t1k_job_write_outputs")

    def genotype_text(self):
        need = C.c_uint64()
        lib().t1k_job_genotype_text(self.h, None, 0, C.byref(need))
        buf = C.create_string_buffer(need.value + 1)
        self._check(lib().t1k_job_genotype_text(self.h, buf, need.value + 1, C.byref(need)), "t1k_job_genotype_text")
        return buf.value.decode()

    def counts(self):
        a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        e = C.c_int32()
        self._check(lib().t1k_job_counts(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e)), "t1k_job_counts")
        return dict(fragments=a.value, assigned_fragments=b.value, groups=c.value, ecs=d.value, em_iterations=e.value)

    def stats(self):
        s = Stats()
        lib().t1k_job_stats(self.h, C.byref(s))
        return s.as_dict()

    # ---- multi-GPU building blocks (include/t1k_gpu.h, "multi-GPU jobs") ----
    def set_shard(self, rank, n_ranks, comm):
        """this job is rank `rank` of `n_ranks`: it owns the fragments [F*rank/n, F*(rank+1)/n) of the loaded input"""
        self._comm = comm
        self._check(lib().t1k_job_set_shard(self.h, rank, n_ranks, comm.h if comm is not None else None), "t1k_job_set_shard")

    def share_reads(self, other):
        self._check(lib().t1k_job_share_reads(self.h, other.h), "t1k_job_share_reads")

    def ctx(self):
        return lib().t1k_job_ctx(self.h)

    def run_local(self):
        self._check(lib().t1k_job_run_local(self.h), "t1k_job_run_local")

    def finish(self):
        self._check(lib().t1k_job_finish(self.h), "t1k_job_finish")

    def groups_serialize(self):
        need = C.c_uint64()
        self._check(lib().t1k_job_groups_serialize(self.h, None, 0, C.byref(need)), "t1k_job_groups_serialize")
        buf = np.zeros(need.value, dtype=np.uint8)
        self._check(lib().t1k_job_groups_serialize(self.h, _ptr(buf), need.value, C.byref(need)), "t1k_job_groups_serialize")
        return buf

    def groups_merge(self, bufs):
        """the group tables of all pattern owners (byte strings of groups_serialize) become this job's table, ordered by first fragment"""
        bufs = [np.ascontiguousarray(b, dtype=np.uint8) for b in bufs]
        ptrs = (C.c_void_p * len(bufs))(*[b.ctypes.data for b in bufs])
        lens = np.array([b.size for b in bufs], dtype=np.uint64)
        self._check(lib().t1k_job_groups_merge(self.h, ptrs, _ptr(lens), len(bufs)), "t1k_job_groups_merge")

    def coalesce_rows(self, rows, row_counts, fragments=None):
        rows = np.ascontiguousarray(rows, dtype=ROW_DTYPE)
        rc = np.ascontiguousarray(row_counts, dtype=np.uint32)
        fr = None if fragments is None else np.ascontiguousarray(fragments, dtype=np.uint32)
        self._check(lib().t1k_job_coalesce_rows(self.h, _ptr(rows), _ptr(rc), _ptr(fr), len(rc)), "t1k_job_coalesce_rows")

    def call_variants(self, abundance, var_max_group, asg_ptr, asg, ops, reads1, reads2=None):
        """VariantCaller::ComputeVariant on this job's alleles (t1k_variants_call; host code, a device=-1 job will do): asg[asg_ptr[f] :
        asg_ptr[f + 1]] = fragment f's assignment list (FRAG_ASG_DTYPE), ops = the edit strings its ops1 / ops2 offsets point into"""
        ab = np.ascontiguousarray(abundance, dtype=np.float64)
        ap = np.ascontiguousarray(asg_ptr, dtype=np.uint64)
        asg = np.ascontiguousarray(asg, dtype=FRAG_ASG_DTYPE)
        ops = np.ascontiguousarray(ops, dtype=np.int8)
        n = len(ap) - 1

        def side(reads):
            b = [r.encode() if isinstance(r, str) else r for r in reads]
            return b, (C.c_char_p * n)(*b), np.array([len(x) for x in b], dtype=np.uint32)
        b1, p1, l1 = side(reads1)
        b2, p2, l2 = side(reads2) if reads2 is not None else (None, None, None)
        h = C.c_void_p()
        self._check(lib().t1k_variants_call(self.h, _ptr(ab), var_max_group, n, _ptr(ap), _ptr(asg), _ptr(ops), p1, _ptr(l1), p2, _ptr(l2), C.byref(h)), "t1k_variants_call")
        return Variants(h, self)

    def coverage_device(self):
        """(device pointer, element count) of the int32 coverage difference array"""
        p, n = C.c_void_p(), C.c_uint64()
        ctx = lib().t1k_job_ctx(self.h)
        if lib().t1k_coverage_device(ctx, C.byref(p), C.byref(n)) != 0:
            raise T1kError("t1k_coverage_device failed")
        return p.value, n.value


def fragment_details(l1, l2, alleles, paired=True):
    """t1k_fragment_details: the overlaps behind a fragment's kept assignments (OVERLAP_DTYPE lists of its read-ends -> FRAG_ASG_DTYPE)"""
    l1 = np.ascontiguousarray(l1, dtype=OVERLAP_DTYPE)
    l2 = np.ascontiguousarray(l2 if l2 is not None else [], dtype=OVERLAP_DTYPE)
    al = np.ascontiguousarray(alleles, dtype=np.int32)
    out = np.zeros(len(al), dtype=FRAG_ASG_DTYPE)
    rc = lib().t1k_fragment_details(_ptr(l1), len(l1), _ptr(l2), len(l2), 1 if paired else 0, _ptr(al), len(al), _ptr(out))
    if rc != 0:
        raise T1kError("t1k_fragment_details failed (%d): an allele of the row has no candidate in the lists" % rc)
    return out


class Variants:
    """called variants of an analyzer run (t1k_variants): VariantCaller's finalVariants and what reads them"""

    def __init__(self, h, job):
        self.h, self.job = h, job  # (the job must outlive the result)

    def close(self):
        if self.h:
            lib().t1k_variants_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def records(self):
        out = np.zeros(lib().t1k_variants_count(self.h), dtype=VARIANT_DTYPE)
        if len(out) and lib().t1k_variants_get(self.h, _ptr(out)) != 0:
            raise T1kError("t1k_variants_get failed")
        return out

    def vcf(self):
        """the text of <prefix>_allele.vcf"""
        need = C.c_uint64()
        if lib().t1k_variants_vcf(self.h, None, 0, C.byref(need)) != 0:
            raise T1kError("t1k_variants_vcf failed")
        buf = C.create_string_buffer(need.value + 1)
        if lib().t1k_variants_vcf(self.h, buf, need.value + 1, C.byref(need)) != 0:
            raise T1kError("t1k_variants_vcf failed")
        return buf.value.decode()

    def adjust(self, asg, ops, read1, read2=None):
        """VariantCaller::AdjustFragmentAssignment for one fragment: a 0/1 flag per assignment"""
        asg = np.ascontiguousarray(asg, dtype=FRAG_ASG_DTYPE)
        ops = np.ascontiguousarray(ops, dtype=np.int8)
        keep = np.zeros(len(asg), dtype=np.uint8)
        b1 = read1.encode() if isinstance(read1, str) else read1
        b2 = None if read2 is None else (read2.encode() if isinstance(read2, str) else read2)
        if lib().t1k_variants_adjust(self.h, _ptr(asg), len(asg), _ptr(ops), b1, len(b1), b2, len(b2) if b2 else 0, _ptr(keep)) != 0:
            raise T1kError("t1k_variants_adjust failed")
        return keep


class CommGroup:
    """meeting point of the rank threads of one process (t1k_comm_group)"""

    def __init__(self, n_ranks):
        self.h = lib().t1k_comm_group_create(n_ranks)

    def close(self):
        if self.h:
            lib().t1k_comm_group_destroy(self.h)
            self.h = None


def pool_release():
    """give the process-wide device memory pool back to the driver; returns the bytes released"""
    return int(lib().t1k_pool_release())


def comm_unique_id():
    """128 bytes from ncclGetUniqueId (rank 0 calls this and hands them to the other ranks)"""
    buf = np.zeros(128, dtype=np.uint8)
    if lib().t1k_comm_unique_id(_ptr(buf)) != 0:
        raise T1kError("t1k_comm_unique_id failed: RCCL not available")
    return buf


class Comm:
    """One rank's communicator (t1k_comm): RCCL over xGMI between processes / GPUs, in-process transport between threads that share a GPU."""

    def __init__(self, owner, n_ranks, rank, unique_id=None, group=None, transport=-1):
        """owner: a Job or a Context on this rank's device (the communicator uses its stream until bind() moves it)"""
        h = C.c_void_p()
        uid = None if unique_id is None else np.ascontiguousarray(unique_id, dtype=np.uint8)
        ctx = owner.ctx() if hasattr(owner, "ctx") else owner.h
        rc = lib().t1k_comm_init(ctx, n_ranks, rank, _ptr(uid), group.h if group is not None else None, transport, C.byref(h))
        self.h = h
        if rc != 0:
            raise T1kError("t1k_comm_init failed (%d): %s" % (rc, lib().t1k_comm_last_error(h).decode() if h else ""))

    def bind(self, owner):
        """owner: a Job or a Context, as in __init__"""
        ctx = owner.ctx() if hasattr(owner, "ctx") else owner.h
        if lib().t1k_comm_bind(self.h, ctx) != 0:
            raise T1kError("t1k_comm_bind failed")

    def is_rccl(self):
        return bool(lib().t1k_comm_is_rccl(self.h))

    def close(self):
        if self.h:
            lib().t1k_comm_destroy(self.h)
            self.h = None


# ---------------------------------------------------------------------------------------------------------------------
# small host-side helpers for tests / bench (format conventions of the reference's ReadFiles.hpp / SeqSet::InputRefSeq)
# ---------------------------------------------------------------------------------------------------------------------
def _open(path):
    return gzip.open(path, "rt") if path.endswith(".gz") else open(path, "rt")


def read_fastx(path):
    """-> list of (id, comment, seq).  id loses a trailing /1 or /2 (ReadFiles.hpp:185-189)."""
    out = []
    with _open(path) as f:
        lines = [l.rstrip("\r\n") for l in f]
    i = 0
    while i < len(lines):
        l = lines[i]
        if not l or l[0] not in ">@":
            i += 1
            continue
        fq = l[0] == "@"
        head = l[1:].split(None, 1)
        rid = head[0] if head else ""
        comment = head[1] if len(head) > 1 else ""
        if len(rid) >= 2 and rid[-2] == "/" and rid[-1] in "12":
            rid = rid[:-2]
        i += 1
        seq = []
        while i < len(lines) and not (lines[i][:1] in (">", "@", "+") and lines[i][:1] != ""):
            seq.append(lines[i])
            i += 1
        seq = "".join(seq)
        if fq and i < len(lines) and lines[i][:1] == "+":
            i += 1
            q = 0
            while i < len(lines) and q < len(seq):
                q += len(lines[i])
                i += 1
        out.append((rid, comment, seq))
    return out


def load_reference_fasta(path):
    """Reference loading as Genotyper::InitRefSet does it (Genotyper.hpp:707-730): identical sequences collapse onto the
    first name; exon mask from the header comment (SeqSet.hpp:933-976).  -> names, seqs, exon masks (uint8 arrays), weights"""
    names, seqs, masks, weights, seen = [], [], [], [], {}
    for rid, comment, seq in read_fastx(path):
        if seq in seen:
            weights[seen[seq]] += 1
            continue
        seen[seq] = len(seqs)
        L = len(seq)
        exons = []
        if comment:
            nums, n = [], 0
            for ch in comment:
                if ch.isdigit():
                    n = n * 10 + int(ch)
                else:
                    nums.append(n)
                    n = 0
            if n:
                nums.append(n)
            if nums:
                for i in range(1, len(nums), 2):
                    exons.append((nums[i], nums[i + 1] if i + 1 < len(nums) else 0))
            else:
                exons.append((0, L - 1))
        else:
            exons.append((0, L - 1))
        m = np.zeros(L, dtype=np.uint8)
        for a, b in exons:
            m[max(a, 0):min(b, L - 1) + 1] = 1
        names.append(rid); seqs.append(seq); masks.append(m); weights.append(1)
    return names, seqs, masks, weights
