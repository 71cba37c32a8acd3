#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X genotyper hot path.

Metric (BASELINE.json / SURVEY 8d): genotyped reads/sec END TO END -- fragments divided by the wall time of the whole genotyper
stage, from opening the reference FASTA and the FASTQ files to the closed *_genotype.tsv / *_allele.tsv / *_aligned_{1,2}.fa.
A "step" is one complete pass of that stage inside this process: reference parse + pack + index build + upload, FASTQ mapping and
record indexing, the windows of fragments streamed through the GPU (upload, 2-bit pack, identical-read-end collapse, read-end
assignment, mate pairing), coalescing, equivalence classes, SQUAREM EM, allele selection, and all output files written.  Nothing is
kept between steps except the HIP runtime itself (the process), the OS page cache of the input files and the library's device-memory
pool (t1k_capi.hip, T1K_POOL_GB: blocks a finished job frees are handed to the next one instead of going back to the driver, which
would zero them again at ~35 ms/GB; what is left of a finished job's host memory is released by a detached thread) -- so a step is a
WARM-process figure.  The cold figure -- a fresh `genotyper` process on the same files, stopwatch from exec to exit -- is measured
beside it on rank 0 and printed as config.executable_cold_run (--no-executable-check skips it).

Workload at N=1: 10 M synthetic 2x150 bp pairs against the HLA-like rna reference (north_star's "10M synthetic 2x150 bp HLA reads";
the real hlaidx_rna_seq.fa cannot be downloaded: tools/t1k_synth generates a reference of the same shape, seed 20250614);
--pairs 1000000 gives BASELINE.json configs[1].  With N ranks ONE sample of N x --pairs fragments is genotyped by all of them
(weak scaling): rank r owns the r-th contiguous slice of the fragments; the exchanges (coverage all-reduce, row exchange to the
pattern owners, group gather, the all-gather of every EM update's contribution slices) run inside libt1k_gpu.so over RCCL.  torch.distributed only
launches the ranks, hands rank 0's ncclUniqueId to the others and provides the barriers of the timing contract.

  python bench.py --gpus 1 --steps 3 --warmup 1            (single GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line with the contract fields plus "roofline" (dominant kernel: algorithmic bytes / HIP-event time vs the
8 TB/s HBM peak, and the whole pipeline's algorithmic bytes over the step time) and "cpu_baseline" (the reference genotyper built
from /root/reference by oracle/Makefile, timed here on a bounded sample of the same workload with -t = all host cores).
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
READ_LEN = 150


def sh(cmd, **kw):
    return subprocess.run(cmd, check=True, **kw)


def ensure_inputs(workdir, pairs, genes, scale, seed, barcodes=0):
    """synthetic HLA-like reference + reads (deterministic); cached per parameter set"""
    synth = os.path.join(ROOT, "tools", "t1k_synth")
    os.makedirs(workdir, exist_ok=True)
    ref = os.path.join(workdir, "hla_g%d_s%s.fa" % (genes, scale))
    if not os.path.exists(ref):
        with open(ref + ".tmp", "w") as f:
            sh([synth, "ref-rna", "--genes", str(genes), "--scale", str(scale), "--seed", "20250614"], stdout=f)
        os.replace(ref + ".tmp", ref)
    pfx = os.path.join(workdir, "reads_g%d_s%s_p%d_seed%d" % (genes, scale, pairs, seed)) + ("_bc%d" % barcodes if barcodes else "")
    if not os.path.exists(pfx + "_2.fq"):
        sh([synth, "reads", "--ref", ref, "--pairs", str(pairs), "--len", str(READ_LEN), "--seed", str(seed), "--out", pfx + ".tmp"] +
           (["--barcodes", str(barcodes)] if barcodes else []))
        for s in ("_1.fq", "_2.fq", "_truth.tsv") + (("_bc.fa",) if barcodes else ()):
            os.replace(pfx + ".tmp" + s, pfx + s)
    return ref, pfx


def distinct_input_files(workdir, pairs, world, genes=24, scale=1.0, free_bytes=None):
    """how many of the `world` read sets fit the scratch disk beside the output files (all of them, normally)"""
    if world <= 1:
        return 1
    try:
        per_set = pairs * (2 * (2 * READ_LEN + 32) + 48)          # two FASTQ files + the generator's truth table
        outputs = world * pairs * 2 * (READ_LEN + 20)             # the two aligned-read FASTA files of the whole job
        have = 0
        for i in range(world):                                    # sets that exist already cost nothing more
            pfx = os.path.join(workdir, "reads_g%d_s%s_p%d_seed%d" % (genes, scale, pairs, 2 + i))
            have += os.path.exists(pfx + "_2.fq")
        if free_bytes is None:
            os.makedirs(workdir, exist_ok=True)
            import shutil
            free_bytes = shutil.disk_usage(workdir).free
        room = free_bytes - outputs - (4 << 30)
        k = have + max(0, int(room // per_set)) if room > 0 else have
        return max(1, min(world, k))
    except Exception:
        return world


def head_fastq(src, dst, n):
    with open(src) as f, open(dst, "w") as g:
        for i, line in enumerate(f):
            if i >= 4 * n:
                break
            g.write(line)


def cpu_baseline(ref, pfx, workdir, pairs_total):
    """the reference's genotyper (oracle/_ref/genotyper, built by oracle/Makefile from /root/reference) on the host cores of this box,
    on a bounded sample (first n pairs) of the same workload, at -t 32, -t 64 and -t nproc: `value` is the BEST of them (the reference
    slices the sorted read-ends by thread and stops scaling well before 256 threads); the reference-load time (the same command on an
    empty read file) is reported separately.  Without the reference binary the baseline is reported as unmeasured."""
    cores = os.cpu_count() or 1
    refbin = os.path.join(ROOT, "oracle", "_ref", "genotyper")
    if not os.path.exists(refbin):
        return dict(value=None, unit="read pairs/s", cores=0, kind="unmeasured",
                    sample="oracle/_ref/genotyper is not built here (oracle/Makefile builds it where /root/reference exists)")
    n = min(pairs_total, 30000)
    s1, s2 = os.path.join(workdir, "cpu_1.fq"), os.path.join(workdir, "cpu_2.fq")
    out = os.path.join(workdir, "cpu_out")

    def run(k, threads):
        head_fastq(pfx + "_1.fq", s1, k)
        head_fastq(pfx + "_2.fq", s2, k)
        t0 = time.time()
        sh([refbin, "-f", ref, "-1", s1, "-2", s2, "-s", "0.97", "-t", str(threads), "-o", out], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
        return time.time() - t0

    load = run(0, min(64, cores))
    wall = {t: run(n, t) for t in sorted({min(32, cores), min(64, cores), cores})}
    best = min(wall, key=wall.get)
    dt = wall[best]
    full = None  # the reference's own run over the WHOLE workload on a box of this pool (committed: it takes most of an hour)
    rec = reference_hashes(pairs_total, 24, 1.0, 0)
    if rec and rec.get("reference_run"):
        full = dict(rec["reference_run"], pairs=pairs_total)
    return dict(value=n / dt, unit="read pairs/s", cores=best, kind="reference", wall_s=dt, reference_load_s=load,
                value_without_reference_load=n / max(dt - load, 1e-9),
                by_threads={str(t): n / w for t, w in wall.items()},
                full_workload_reference_run=full,
                sample="first %d of %d pairs, same reference, -s 0.97; best of -t %s (read pairs/s by thread count in by_threads); wall %.1f s at -t %d of which %.1f s is the "
                       "reference load (same command, no reads)" % (n, pairs_total, " / ".join(str(t) for t in wall), dt, best, load))


def md5_file(path):
    import hashlib
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def reference_hashes(pairs, genes, scale, barcodes):
    """md5 sums of the files the REFERENCE genotyper wrote for this very input on an MI355X host (tools/full_size_parity_r03.sh; the
    10 M-pair run takes the reference most of an hour at -t 64), committed as tests/golden/full_size_md5.json"""
    path = os.path.join(ROOT, "tests", "golden", "full_size_md5.json")
    if not os.path.exists(path) or genes != 24 or scale != 1.0:
        return None
    for rec in json.load(open(path)).values():
        if rec.get("pairs") == pairs and rec.get("seed") == 2 and int(rec.get("barcodes") or 0) == int(barcodes or 0) and "sub" not in rec:
            return rec
    return None


FAMILIES = [("k_seed_groups", r"k_seed_groups|k_seed_chain|k_seed_long"), ("k_pair", r"k_pair"), ("k_extend", r"k_extend"), ("k_select", r"k_select"),
            ("fullalign kernels", r"k_truncate|k_pack_overlaps|k_publish_lists|k_fullalign|k_align_"),
            ("chain kernels", r"k_chain_|k_dp_dense|k_gather_general|k_near_hits|k_collect|k_general_finish|k_arena_compact|k_group_size_keys|k_job_keys|k_csort|k_widen_keys")]


def kernel_bytes_8d(st):
    """ALGORITHMIC bytes of one step by SURVEY.md 8(d) AS WRITTEN, every term once, every count measured by the device itself: per distinct
    read-end 3l/8 (packed read) + sum over looked-up k-mers of (8 + 8 L_j) (bucket header + postings) + C x 60 (allele window of every
    candidate reaching extension) + C' x 32 (emitted overlap record) + C'' x 12 (coverage updates of the near-best overlaps); per fragment
    (C'_1 + C'_2) x 32 read back for pairing + R_f x 16 (kept row entries).  Attributed to the kernel family that first needs the bytes: the
    window term to the chain family (it reads the allele window to score the candidate), the record term to selection, the coverage term to
    the near-best alignments.  This is the figure `pipeline_frac_8d` is computed from."""
    re, L = st["read_ends"], READ_LEN
    return {
        "k_seed_groups": re * (3 * L / 8.0) + st["lookups"] * 8 + st["postings"] * 8,
        "chain kernels": st["candidates"] * 60,
        "k_extend": 0.0,
        "k_select": st["extended"] * 32,
        "fullalign kernels": st["near_best"] * 12,
        "k_pair": st["pair_overlaps"] * 32 + st["rows"] * 16,
    }


def kernel_bytes(st):
    """PER-KERNEL RE-COUNT (not 8d's formula): what each kernel family has to touch at least once given the stage split of this build --
    the chain family looks at an allele window per GROUP (8d counts one per candidate), extension and the near-best alignments read the
    candidate's window again, records are re-read by the stage after the one that wrote them.  Terms appear several times on purpose: it
    is the lower bound of each family's OWN traffic and is what the per-family `frac` figures use; the pipeline total by this count is
    printed as pipeline_frac_recount and must not be read as 8d's."""
    re, L = st["read_ends"], READ_LEN
    return {
        "k_seed_groups": re * (3 * L / 8.0) + st["lookups"] * 8 + st["postings"] * 8,
        "chain kernels": st["groups"] * 60 + st["candidates"] * 24,
        "k_extend": st["candidates"] * (24 + 60 + 24),
        "k_select": st["candidates"] * (24 + 24) + st["extended"] * 32,
        "fullalign kernels": st["extended"] * 32 + st["near_best"] * (60 + 12),
        "k_pair": st["pair_overlaps"] * 32 + st["rows"] * 24,
    }


def alone_ms_per_step(pairs):
    """kernel time per family with ONE pipeline (no overlap between streams) at the bench size, from the committed rocprofv3 --kernel-trace
    --stats summary of `T1K_PIPELINES=1 python bench.py --pairs <pairs>` (profiles/r*_kernel_stats_10M_1pipeline.csv + .json naming the
    steps profiled); None if no such profile is committed for this size"""
    import csv, glob, re
    metas = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats_*_1pipeline.json")))
    for mpath in reversed(metas):
        meta = json.load(open(mpath))
        if meta.get("pairs") != pairs:
            continue
        out = {n: 0.0 for n, _ in FAMILIES}
        other = 0.0
        for r in csv.DictReader(open(os.path.join(ROOT, "profiles", meta["csv"]))):
            ms = float(r["TotalDurationNs"]) / 1e6 / max(1, meta["passes"])
            for n, rx in FAMILIES:
                if re.search(rx, r["Name"]):
                    out[n] += ms
                    break
            else:
                other += ms
        return out, other, os.path.relpath(mpath, ROOT)
    return None, None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=10000000, help="read pairs per GPU (weak scaling)")
    ap.add_argument("--genes", type=int, default=24)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--workdir", default=os.environ.get("T1K_BENCH_DIR", "/tmp/t1k_bench"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-executable-check", action="store_true", help="skip the cold run of t1k_amd/bin/genotyper on the same files (stopwatch around the process) that is timed beside the steps")
    ap.add_argument("--executable-check", action="store_true", help="(default since round 3; kept for old command lines)")
    ap.add_argument("--no-roofline-step", action="store_true", help="skip the untimed one-pipeline step the roofline figures are measured on")
    ap.add_argument("--cold-runs", type=int, default=5, help="fresh `genotyper` processes timed for value_cold (median; all of them listed)")
    ap.add_argument("--barcodes", type=int, default=0, help="BASELINE configs[4]: the reads carry this many 10x-style barcodes (--barcode file; log-uniform usage)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # RCCL on ROCm
    import numpy as np
    import t1k_amd

    # inputs: ONE sample of world x --pairs fragments, given as `world` files per mate read back to back (every -1 / -2 of the
    # executable counts, ReadFiles::AddReadFile); every rank maps the same files and owns a contiguous slice of the fragments.
    # File i is the same for every world size (seed 2 + i), so the N = 1 input is the first file of the N = 8 input; each rank
    # generates one of the files.
    total_pairs = a.pairs * world
    # The scratch disk has to hold `world` read sets and the job's aligned-read files (80 GB at 8 x 10 M pairs).  Where it cannot,
    # the first k read sets are given several times (rank r reads set r mod k) and the line says so in config.distinct_input_files.
    distinct = distinct_input_files(a.workdir, a.pairs, world, a.genes, a.scale)
    if rank == 0:
        ensure_inputs(a.workdir, a.pairs, a.genes, a.scale, seed=2, barcodes=a.barcodes)   # (also writes the reference)
    if dist is not None:
        dist.barrier()
        if rank < distinct:
            ensure_inputs(a.workdir, a.pairs, a.genes, a.scale, seed=2 + rank, barcodes=a.barcodes)
        dist.barrier()
    parts = [ensure_inputs(a.workdir, a.pairs, a.genes, a.scale, seed=2 + i % distinct, barcodes=a.barcodes) for i in range(world)]
    ref, pfx = parts[0]
    files1, files2 = [p_[1] + "_1.fq" for p_ in parts], [p_[1] + "_2.fq" for p_ in parts]
    barcode_file = None
    if a.barcodes:
        # one barcode file for the whole sample (the genotyper takes one --barcode): the ranks' files back to back
        barcode_file = parts[0][1] + "_bc.fa" if world == 1 else os.path.join(a.workdir, "barcodes_x%d_p%d_bc%d.fa" % (world, a.pairs, a.barcodes))
        if world > 1 and rank == 0 and not os.path.exists(barcode_file):
            with open(barcode_file + ".tmp", "w") as o:
                for p_ in parts:  # (record names are unused by the genotyper: Genotyper.cpp:372-392 reads the sequences in step with the reads)
                    o.write(open(p_[1] + "_bc.fa").read())
            os.replace(barcode_file + ".tmp", barcode_file)
        if dist is not None:
            dist.barrier()
    out_prefix = os.path.join(a.workdir, "out")
    want = reference_hashes(a.pairs, a.genes, a.scale, a.barcodes) if world == 1 else None
    hashes_ok = []
    last = {}
    comm = anchor = None
    if dist is not None:
        # the job's own communicator (RCCL inside libt1k_gpu.so), created once: rank 0's ncclUniqueId travels over torch.distributed
        uid = torch.from_numpy(t1k_amd.comm_unique_id() if rank == 0 else np.zeros(128, dtype=np.uint8)).cuda()
        dist.broadcast(uid, 0)
        anchor = t1k_amd.Context(device=local_rank)
        comm = t1k_amd.Comm(anchor, world, rank, unique_id=uid.cpu().numpy())

    seg = {}  # wall time of the step's calls, summed over the timed steps (config.calls_ms)

    def step():
        t = [time.perf_counter()]
        if comm is None and not os.environ.get("T1K_SERIAL_OPEN"):
            # as the executable does: the read files are mapped and indexed by a second thread while the reference is parsed and the
            # contexts come up (t1k_reads_open / t1k_job_attach_reads; both C calls release the GIL).  calls_ms: "job_create_reference"
            # is then the wall time of the two together, "load_reads" what is left after it (the join + attach)
            box = {}

            def opener():
                try:
                    box["reads"] = t1k_amd.Reads(files1, files2, barcode=barcode_file)
                except Exception as e:  # noqa: BLE001 -- re-raised on the main thread below
                    box["error"] = e
            th = threading.Thread(target=opener)
            th.start()
            try:
                job = t1k_amd.Job(ref, ref_seq_similarity=0.97, device=local_rank)
            finally:
                th.join()
            t.append(time.perf_counter())
            if "error" in box:
                raise box["error"]
            job.attach_reads(box["reads"])
        else:
            job = t1k_amd.Job(ref, ref_seq_similarity=0.97, device=local_rank)
            if comm is not None:
                comm.bind(job)
                job.set_shard(rank, world, comm)    # before the reads: a rank indexes only its own fragments of the files (host/reads.cpp)
            t.append(time.perf_counter())
            job.load_reads(files1, files2, barcode=barcode_file)
        job.set_output_prefix(out_prefix)       # as the executable does: the aligned-read files are written while the EM runs
        t.append(time.perf_counter())
        job.run()
        t.append(time.perf_counter())
        job.write_outputs(out_prefix)           # rank 0: the two tables; every rank: its own part of the aligned-read files
        t.append(time.perf_counter())
        last["stats"] = job.stats()
        last["counts"] = job.counts()
        last["text"] = job.genotype_text()
        if want is not None:  # every step's calls against the reference's own output for this input (the big files: after the last step)
            import hashlib
            hashes_ok.append(hashlib.md5(last["text"].encode()).hexdigest() == want["_genotype.tsv"] and md5_file(out_prefix + "_allele.tsv") == want["_allele.tsv"])
        if comm is not None:
            comm.bind(anchor)
        t.append(time.perf_counter())
        job.close()
        t.append(time.perf_counter())
        for i, k in enumerate(("job_create_reference", "load_reads", "run", "write_outputs", "stats_text_reference_check", "job_close")):
            seg[k] = seg.get(k, 0.0) + (t[i + 1] - t[i]) * 1e3

    # peak device memory of the process (hipMemGetInfo sampled 20 x a second on a side thread: total - least free seen, pooled blocks included)
    mem = {"free_min": None, "total": None, "stop": False}

    def mem_sampler():
        while not mem["stop"]:
            try:
                free, total = torch.cuda.mem_get_info(local_rank)
                mem["total"] = total
                mem["free_min"] = free if mem["free_min"] is None else min(mem["free_min"], free)
            except Exception:  # noqa: BLE001
                pass
            time.sleep(0.05)
    mem_thread = threading.Thread(target=mem_sampler, daemon=True)
    mem_thread.start()

    for _ in range(a.warmup):
        step()
    seg.clear()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        comm.close()
        anchor.close()
        dist.destroy_process_group()
    st, counts, text = last["stats"], last["counts"], last["text"]
    mem["stop"] = True
    mem_thread.join()
    # roofline step (untimed, after the K timed steps): the same job once more with ONE pipeline, so that every family's HIP-event time is
    # the time of kernels that had the device to themselves -- the overlapped streams of the timed steps stretch each other by 2 - 2.5 x and
    # by a different amount from run to run (frac swung 0.021 <-> 0.034 on identical code).  Checked against the reference's
    # hashes like every other step.
    st1 = None
    if world == 1 and not a.no_roofline_step:
        keep_env, keep_seg = os.environ.get("T1K_PIPELINES"), dict(seg)
        os.environ["T1K_PIPELINES"] = "1"
        try:
            step()
            st1 = last["stats"]
        finally:
            if keep_env is None:
                del os.environ["T1K_PIPELINES"]
            else:
                os.environ["T1K_PIPELINES"] = keep_env
            seg.clear()
            seg.update(keep_seg)
    import ctypes
    ctypes.CDLL(None).fflush(None)  # RCCL prints its banner through C stdio: get it out before the JSON line
    if rank == 0:
        ms = {"k_seed_groups": st["ms_seed"], "chain kernels": st["ms_chain"], "k_extend": st["ms_extend"], "k_select": st["ms_select"],
              "fullalign kernels": st["ms_fullalign"], "k_pair": st["ms_pair"]}
        kb = kernel_bytes(st)       # per-kernel re-count (a family's own lower bound; terms repeat across families)
        kb8 = kernel_bytes_8d(st)   # SURVEY 8d as written, every term once
        # the dominant kernel (family) is the one with the most MEASURED time (HIP events on the launch streams, summed over the timed
        # step's ranges); a "launch" of a family is one pass over one range of distinct read-ends (k_pair: one range of fragments).
        # With several pipelines per GPU those event times are STREAM times: the streams overlap, so their sum exceeds the step's wall
        # time and `frac` is a lower bound; `frac_alone` prices the same bytes with the family's kernel time in a one-pipeline run.
        # roofline.frac / achieved are measured on the untimed ONE-pipeline step behind the timed ones (kernels alone on the device: event
        # time = kernel time); the overlapped-stream figure of the timed steps is kept as frac_overlapped_streams.
        fam = lambda x: {"k_seed_groups": x["ms_seed"], "chain kernels": x["ms_chain"], "k_extend": x["ms_extend"], "k_select": x["ms_select"],  # noqa: E731
                         "fullalign kernels": x["ms_fullalign"], "k_pair": x["ms_pair"]}
        ms1 = fam(st1) if st1 is not None else None
        dom = max(ms1, key=lambda k: ms1[k]) if ms1 else max(ms, key=lambda k: ms[k])
        launches = max(1, st["batches"])
        launches1 = max(1, st1["batches"]) if st1 is not None else launches
        kb1 = kernel_bytes(st1) if st1 is not None else kb
        achieved_overlapped = kb[dom] / (ms[dom] * 1e-3) / 1e9 if ms[dom] > 0 else 0.0
        achieved = (kb1[dom] / (ms1[dom] * 1e-3) / 1e9 if ms1[dom] > 0 else 0.0) if ms1 else achieved_overlapped
        step_s = dt / a.steps
        alone, alone_other, alone_src = alone_ms_per_step(a.pairs) if world == 1 else (None, None, None)
        # HBM traffic by the PMC counters: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of THIS command line (tools/profile_r04.sh
        # writes profiles/r04_traffic.json: per kernel family, bytes per step, FETCH x 2 calibrated + WRITE); not measured in this run
        traffic = traffic_src = None
        traffic_all = {}
        import glob
        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
        tfiles = [f for f in tfiles if "seed_traffic" not in f]
        if tfiles:
            tj = json.load(open(tfiles[-1]))
            if tj.get("pairs") == a.pairs and world == 1:
                traffic_all = tj.get("bytes_per_step", {})
                if dom in traffic_all:
                    traffic = traffic_all[dom] / launches
                traffic_src = "%s: %s" % (os.path.relpath(tfiles[-1], ROOT), tj.get("note", ""))
        out = {
            "metric": "genotyped reads/sec (end-to-end genotyper stage, 2x150 bp HLA)",
            "value": total_pairs * a.steps / dt,
            "unit": "read pairs/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": step_s * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "%d synthetic 2x150 bp pairs per GPU vs synthetic HLA-like rna reference (%d genes, scale %s: %d alleles), -s 0.97%s; "
                                   "END TO END per step: reference FASTA -> index -> FASTQ parse -> GPU -> genotype.tsv + allele.tsv + aligned_{1,2}.fa written"
                                   % (a.pairs, a.genes, a.scale, sum(1 for l in open(ref) if l.startswith(">")),
                                      ", --barcode with %d barcodes (BASELINE configs[4]; + aligned_bc.fa written)" % a.barcodes if a.barcodes else ""),
                       "process_state": "warm: device-memory pool and HIP runtime kept across steps (see executable_cold_run for a fresh process)",
                       "parallelism": ("one sample of %d pairs sharded over %d GPUs by contiguous fragment slices (%d per GPU); RCCL inside libt1k_gpu.so: int32 all-reduce of the coverage arrays, "
                                       "all-to-all of fragment rows to their pattern owners, all-gather of group tables, all-gather of the ranks' contribution slices in every EM update; "
                                       "communicator created once outside the steps" % (total_pairs, world, a.pairs)) if world > 1 else "1 GPU",
                       "distinct_input_files": distinct,  # of n_gpus read sets; fewer only where the scratch disk cannot hold them all
                       "arithmetic": "2-bit packed bases in u64 words, int32 alignment scores, f32 read-group weights, f64 EM",
                       "read_ends": st["read_ends_total"], "distinct_read_ends": st["read_ends"],
                       "groups": counts["groups"], "equivalence_classes": counts["ecs"], "em_iterations": counts["em_iterations"],
                       "assigned_fragments": counts["assigned_fragments"],
                       "phases_ms": {"read_files_map_index": st["ms_load"], "device_loop": st["ms_device"], "coalesce": st["ms_coalesce"], "em": st["ms_em"],
                                     "write_outputs": st["ms_write"]},
                       "calls_ms": {k: v / a.steps for k, v in seg.items()},
                       "calls_note": ("job_create_reference = t1k_job_create with t1k_reads_open on a second thread (the executable does the same); load_reads = the join + t1k_job_attach_reads"
                                      if world == 1 and not os.environ.get("T1K_SERIAL_OPEN") else "t1k_job_create, then t1k_job_load_reads")},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "frac_source": ("live: HIP-event time of the family's kernels in an untimed ONE-pipeline step of the same job behind the timed steps (kernels alone on the device)"
                                         if ms1 else "overlapped streams of the timed steps (no one-pipeline step in this run)"),
                         "frac_overlapped_streams": achieved_overlapped / HBM_PEAK_GBS, "achieved_overlapped_streams": achieved_overlapped,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_choice": "the kernel family with the most measured time (HIP events) in the one-pipeline step",
                         "launches_per_step": launches1, "pipelines_per_gpu": int(os.environ.get("T1K_PIPELINES", "3")), "algorithmic_bytes_per_launch": kb1[dom] / launches1,
                         "avg_launch_ms": (ms1[dom] / launches1) if ms1 else ms[dom] / launches,
                         "one_pipeline_step": ({"ms_device_loop": st1["ms_device"], "all_kernels_ms": ms1, "sum_ms": sum(ms1.values()),
                                                "all_kernels_frac": {k: (kb1[k] / (ms1[k] * 1e-3) / 1e9 / HBM_PEAK_GBS if ms1[k] > 0 else 0.0) for k in ms1}} if ms1 else None),
                         "all_kernels_ms_per_step": ms,
                         "all_kernels_algorithmic_GBs": {k: (kb[k] / (ms[k] * 1e-3) / 1e9 if ms[k] > 0 else 0.0) for k in ms},
                         "all_kernels_frac_overlapped_streams": {k: (kb[k] / (ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS if ms[k] > 0 else 0.0) for k in ms},
                         "all_kernels_traffic_bytes_per_step": traffic_all or None,
                         "bytes_model": "achieved / frac / all_kernels_*: PER-KERNEL RE-COUNT (each family's own lower bound: the allele window is counted per group in the chain family, "
                                        "again per candidate in extension and per near-best alignment; not 8d's formula).  pipeline_frac_8d: SURVEY 8d as written, every term once",
                         "time_model": "frac: family's HIP-event time in the one-pipeline step (no other stream on the device).  *_overlapped_streams: event time on the launch streams of the timed "
                                       "steps; with %d pipelines the streams overlap (sum over families %.0f ms for a %.0f ms step) and stretch each other by a factor that differs from run to run.  "
                                       "frac_alone_rocprof: the same family's kernel time in the committed rocprofv3 one-pipeline summary at this size (%s) -- must agree with frac"
                                       % (int(os.environ.get("T1K_PIPELINES", "3")), sum(ms.values()), step_s * 1e3, alone_src or "none committed for this size"),
                         "alone_ms_per_step": alone, "alone_source": alone_src,
                         "frac_alone_rocprof": (kb[dom] / (alone[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS) if alone and alone.get(dom) else None,
                         "all_kernels_frac_alone_rocprof": ({k: (kb[k] / (alone[k] * 1e-3) / 1e9 / HBM_PEAK_GBS if alone[k] > 0 else 0.0) for k in ms} if alone else None),
                         "pipeline_8d_bytes_per_step": sum(kb8.values()), "pipeline_8d_bytes_by_term": kb8,
                         "pipeline_frac_8d": sum(kb8.values()) / step_s / 1e9 / HBM_PEAK_GBS,
                         "pipeline_frac_8d_device_loop": sum(kb8.values()) / max(st["ms_device"] * 1e-3, 1e-9) / 1e9 / HBM_PEAK_GBS,
                         "pipeline_recount_bytes_per_step": sum(kb.values()),
                         "pipeline_frac_recount": sum(kb.values()) / step_s / 1e9 / HBM_PEAK_GBS,
                         "pipeline_frac_recount_device_loop": sum(kb.values()) / max(st["ms_device"] * 1e-3, 1e-9) / 1e9 / HBM_PEAK_GBS,
                         "dp_cell_updates_per_s": st["dp_cells"] / max(st["ms_fullalign"] * 1e-3, 1e-9) if st.get("dp_cells") else None,
                         "em_ms": st["ms_em"], "job_ms_total": st["ms_total"]},
        }
        if want is not None:
            big = {suf: md5_file(out_prefix + suf) == want[suf] for suf in ("_aligned_1.fa", "_aligned_2.fa", "_aligned_bc.fa") if suf in want}
            out["config"]["reference_output_check"] = {
                "what": "md5 of this run's files vs the files the REFERENCE genotyper wrote for this input (tests/golden/full_size_md5.json, made by tools/full_size_parity_r03.sh)",
                "genotype_and_allele_tsv_identical_every_step": bool(hashes_ok) and all(hashes_ok), "steps_checked": len(hashes_ok),
                "aligned_1_fa_identical": big["_aligned_1.fa"], "aligned_2_fa_identical": big["_aligned_2.fa"], "aligned_bc_fa_identical": big.get("_aligned_bc.fa")}
            if not (all(hashes_ok) and all(big.values())):
                out["config"]["reference_output_check"]["FAILED"] = True
        cold_same = None
        if world == 1 and not a.no_executable_check:
            exe = os.path.join(ROOT, "t1k_amd", "bin", "genotyper")
            t1k_amd.pool_release()   # this process's cached device memory would otherwise be fresh (to-be-zeroed) VRAM for the other one
            walls = []
            cold_same = True
            for i in range(max(1, a.cold_runs)):
                time.sleep(10 if i == 0 else 7)  # the driver wipes what a process just returned (> 100 GB) in the background: a process started into that
                                                 # waits for the wipe (8 - 9 s measured instead of 4.8 s for the same command on an idle GPU)
                t1 = time.time()
                sh([exe, "-f", ref, "-1", pfx + "_1.fq", "-2", pfx + "_2.fq", "-s", "0.97", "-o", os.path.join(a.workdir, "exe_out")] +
                   (["--barcode", barcode_file] if barcode_file else []), stderr=subprocess.DEVNULL)
                walls.append(time.time() - t1)
                cold_same = cold_same and open(os.path.join(a.workdir, "exe_out_genotype.tsv")).read() == text
            if want is not None:  # the fresh process's files against the reference's too (the last run's)
                cold_same = cold_same and all(md5_file(os.path.join(a.workdir, "exe_out" + suf)) == want[suf]
                                              for suf in ("_genotype.tsv", "_allele.tsv", "_aligned_1.fa", "_aligned_2.fa", "_aligned_bc.fa") if suf in want)
            wall = sorted(walls)[len(walls) // 2]
            out["config"]["executable_cold_run"] = {"wall_s": wall, "wall_s_all_runs": walls, "wall_s_min": min(walls), "wall_s_max": max(walls), "runs": len(walls),
                                                    "read_pairs_per_s": a.pairs / wall, "genotype_tsv_identical_to_bench": cold_same,
                                                    "files_identical_to_reference": cold_same if want is not None else None}
            # SURVEY 8d reads the metric as process start -> TSV closed: that is the cold executable.  `value` stays the warm in-process
            # step (the contract's "K timed steps"); value_cold is the same workload as one fresh process, exec to exit (median of the runs).
            out["value_cold"] = a.pairs / wall
            out["value_cold_spread"] = {"runs": len(walls), "min": a.pairs / max(walls), "median": a.pairs / wall, "max": a.pairs / min(walls)}
            out["metric_note"] = ("SURVEY 8d defines the metric as process start -> genotype.tsv closed: that is value_cold (a fresh `genotyper` process, exec to exit, same files; median of "
                                  "%d runs). `value` is the bench contract's K timed in-process steps (HIP runtime and the library's device-memory pool kept between steps)." % len(walls))
            rec = reference_hashes(a.pairs, a.genes, a.scale, a.barcodes)
            if rec and rec.get("reference_run", {}).get("read_pairs_per_s"):
                full = rec["reference_run"]["read_pairs_per_s"]
                out["vs_reference_cpu_full_workload"] = {
                    "cold_executable_over_reference": a.pairs / wall / full, "warm_step_over_reference": out["value"] / full,
                    "reference_read_pairs_per_s": full, "reference_threads": rec["reference_run"].get("threads"),
                    "what": "the reference genotyper's own run over this WHOLE workload on a host of this pool (%s); vs_baseline stays null: BASELINE.md "
                            "holds no published number for the metric" % rec["reference_run"].get("log")}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ref, pfx, a.workdir, a.pairs)
        else:
            out["cpu_baseline"] = None
        with open(os.path.join(a.workdir, "last_genotype.tsv"), "w") as f:
            f.write(text)
        if mem["free_min"] is not None:
            out["peak_device_gb"] = (mem["total"] - mem["free_min"]) / 1e9
        # the verdict the credit hangs on, as the LAST key of the line (a tail-truncated record still shows it) and as the exit status:
        # True = every checked step's tables and the aligned-read files (and the fresh processes' files) carry the md5 sums of the files the
        # REFERENCE genotyper wrote for this input; None = no reference hashes are committed for this configuration; False = a difference.
        ok = None
        if want is not None:
            ok = not out["config"]["reference_output_check"].get("FAILED", False) and cold_same is not False
        out["reference_md5_ok"] = ok
        print(json.dumps(out))
        sys.stdout.flush()
        if ok is False:
            sys.stderr.write("bench.py: OUTPUT DIFFERS FROM THE REFERENCE'S (reference_md5_ok false)\n")
            sys.exit(3)


if __name__ == "__main__":
    main()
