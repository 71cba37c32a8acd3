#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X genotyper hot path.

Metric (BASELINE.json): genotyped reads/sec, 2x150 bp HLA, at N MI355X.  A "step" is one full pass of the genotyper stage
over the synthetic read set resident in HBM: read-end assignment (seed/chain/extend/select/full-align kernels), mate
pairing, read-group coalescing, equivalence classes, SQUAREM EM (E-step on the GPU), allele selection.  The workload at
N=1 is BASELINE.json configs[1]: 1 M synthetic 2x150 bp pairs against the HLA-like rna reference (the real
hlaidx_rna_seq.fa cannot be downloaded; tools/t1k_synth generates a reference of the same shape, seed 20250614).

  python bench.py --gpus 1 --steps 3 --warmup 1            (single GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line with the contract fields plus "roofline" (dominant kernel: algorithmic bytes / HIP-event time
vs the 8 TB/s HBM peak) and "cpu_baseline" (the reference genotyper built from /root/reference, timed here on a bounded
sample of the same workload).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
READ_LEN = 150
# HBM traffic of one full-batch k_seed_groups launch (16384 fragments) from the rocprofv3 PMC passes of profiles/r01_pmc_hbm.md:
# FETCH_SIZE 5.37e6 KB (x2: the counter tallies 128-B requests at 64 B on gfx950, MI355X_MICROARCH.md) + WRITE_SIZE 3.12e6 KB
TRAFFIC_BYTES_PER_LAUNCH = 2 * 5.37e9 + 3.12e9


def sh(cmd, **kw):
    return subprocess.run(cmd, check=True, **kw)


def ensure_inputs(workdir, pairs, genes, scale, seed):
    """synthetic HLA-like reference + reads (deterministic); cached per parameter set"""
    synth = os.path.join(ROOT, "tools", "t1k_synth")
    os.makedirs(workdir, exist_ok=True)
    ref = os.path.join(workdir, "hla_g%d_s%s.fa" % (genes, scale))
    if not os.path.exists(ref):
        with open(ref + ".tmp", "w") as f:
            sh([synth, "ref-rna", "--genes", str(genes), "--scale", str(scale), "--seed", "20250614"], stdout=f)
        os.replace(ref + ".tmp", ref)
    pfx = os.path.join(workdir, "reads_g%d_s%s_p%d_seed%d" % (genes, scale, pairs, seed))
    if not os.path.exists(pfx + "_2.fq"):
        sh([synth, "reads", "--ref", ref, "--pairs", str(pairs), "--len", str(READ_LEN), "--seed", str(seed), "--out", pfx + ".tmp"])
        for s in ("_1.fq", "_2.fq", "_truth.tsv"):
            os.replace(pfx + ".tmp" + s, pfx + s)
    return ref, pfx


def head_fastq(src, dst, n):
    with open(src) as f, open(dst, "w") as g:
        for i, line in enumerate(f):
            if i >= 4 * n:
                break
            g.write(line)


def cpu_baseline(ref, pfx, workdir, pairs_total):
    """the reference's genotyper (oracle/_ref/genotyper, built by oracle/Makefile from /root/reference) on the host cores of
    this box, on a bounded sample (first n pairs) of the same workload.  Falls back to the oracle restatement (1 thread)."""
    cores = os.cpu_count() or 1
    threads = min(cores, 32)
    refbin = os.path.join(ROOT, "oracle", "_ref", "genotyper")
    kind, binary = "reference", refbin
    if not os.path.exists(refbin):
        kind, binary, threads = "port", os.path.join(ROOT, "oracle", "t1k_oracle_cli"), 1
    n = min(pairs_total, 480 * threads if kind == "reference" else 400)  # ~20 s of CPU work on the HLA-like workload
    s1, s2 = os.path.join(workdir, "cpu_1.fq"), os.path.join(workdir, "cpu_2.fq")
    head_fastq(pfx + "_1.fq", s1, n)
    head_fastq(pfx + "_2.fq", s2, n)
    out = os.path.join(workdir, "cpu_out")
    t0 = time.time()
    sh([binary, "-f", ref, "-1", s1, "-2", s2, "-s", "0.97", "-t", str(threads), "-o", out], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
    dt = time.time() - t0
    geno = out + "_genotype.tsv"
    return dict(value=n / dt, unit="read pairs/s", cores=threads, kind=kind,
                sample="first %d of %d pairs, same reference, -s 0.97, wall %.1f s incl. reference load" % (n, pairs_total, dt)), (geno if os.path.exists(geno) else None)


def kernel_bytes(st, pairs):
    """ALGORITHMIC bytes per kernel (group) for one step: the terms of SURVEY.md 8d, every count measured by
    the device itself.  k_seed_groups covers the first two terms: the packed read (3l/8 B per read-end) + one 8 B bucket header per
    looked-up k-mer + 8 B per posting of the used lists (each posting counted once)."""
    re, L = st["read_ends"], READ_LEN
    return {
        "k_seed_groups": re * (3 * L / 8.0) + st["lookups"] * 8 + st["postings"] * 8,
        "chain kernels": st["groups"] * 60 + st["candidates"] * 24,
        "k_extend": st["candidates"] * (24 + 60 + 24),
        "k_select": st["candidates"] * (24 + 24) + st["extended"] * 32,
        "fullalign kernels": st["extended"] * 32 + st["near_best"] * (60 + 12),
        "k_pair": st["extended"] * 32 + st["rows"] * 24,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=1000000, help="read pairs per GPU (weak scaling)")
    ap.add_argument("--genes", type=int, default=24)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--workdir", default=os.environ.get("T1K_BENCH_DIR", "/tmp/t1k_bench"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="take the sharded (collective) code path even with one rank")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1 or a.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # RCCL on ROCm
    import t1k_amd
    import bench_dist

    # inputs: one sample of world*pairs fragments, rank r owns the contiguous slice r (fragments in file order)
    total_pairs = a.pairs * world
    if rank == 0:
        ref, pfx = ensure_inputs(a.workdir, total_pairs, a.genes, a.scale, seed=2)
    if dist is not None:
        dist.barrier()
    ref, pfx = ensure_inputs(a.workdir, total_pairs, a.genes, a.scale, seed=2)

    job = t1k_amd.Job(ref, ref_seq_similarity=0.97, device=local_rank)
    if dist is None:
        job.load_reads(pfx + "_1.fq", pfx + "_2.fq")
    else:
        bench_dist.load_shard(job, pfx, rank, world, a.pairs)
        bench_dist.install_allreduce(job, dist, torch)
    job.stage_reads()  # reads are packed and resident in HBM before the timed region

    def step():
        if dist is None:
            job.run()
        else:
            bench_dist.sharded_step(job, dist, torch, rank, world)

    for _ in range(a.warmup):
        step()
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
    st = job.stats()
    counts = job.counts()
    text = job.genotype_text()
    job.close()
    if dist is not None:
        dist.destroy_process_group()
    import ctypes
    ctypes.CDLL(None).fflush(None)  # RCCL prints its banner through C stdio: get it out before the JSON line
    if rank == 0:
        ms = {"k_seed_groups": st["ms_seed"], "chain kernels": st["ms_chain"], "k_extend": st["ms_extend"], "k_select": st["ms_select"],
              "fullalign kernels": st["ms_fullalign"], "k_pair": st["ms_pair"]}
        kb = kernel_bytes(st, a.pairs)
        dom = "k_seed_groups"  # the single largest kernel (profiles/): one launch per device batch
        launches = max(1, st["batches"])
        achieved = kb[dom] / (ms[dom] * 1e-3) / 1e9 if ms[dom] > 0 else 0.0
        out = {
            "metric": "genotyped reads/sec (end-to-end genotyper stage, 2x150 bp HLA)",
            "value": total_pairs * a.steps / dt,
            "unit": "read pairs/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "%d synthetic 2x150 bp pairs per GPU vs synthetic HLA-like rna reference (%d genes, scale %s: %d alleles), -s 0.97, reads packed and resident in HBM"
                                   % (a.pairs, a.genes, a.scale, sum(1 for l in open(ref) if l.startswith(">"))),
                       "parallelism": ("fragments sharded over %d GPUs; RCCL all-reduce of coverage and of the EM read-count vector, all-gather of group tables" % world)
                       if world > 1 else "1 GPU",
                       "arithmetic": "2-bit packed bases in u64 words, int32 alignment scores, f64 EM",
                       "groups": counts["groups"], "equivalence_classes": counts["ecs"], "em_iterations": counts["em_iterations"],
                       "assigned_fragments": counts["assigned_fragments"]},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": TRAFFIC_BYTES_PER_LAUNCH if a.pairs >= 16384 else None,
                         "launches_per_step": launches, "pipelines_per_gpu": int(os.environ.get("T1K_PIPELINES", "4")), "algorithmic_bytes_per_launch": kb[dom] / launches, "avg_launch_ms": ms[dom] / launches,
                         "all_kernels_ms_per_step": ms,
                         "all_kernels_algorithmic_GBs": {k: (kb[k] / (ms[k] * 1e-3) / 1e9 if ms[k] > 0 else 0.0) for k in ms},
                         "em_ms": st["ms_em"], "job_ms_total": st["ms_total"]},
        }
        if world == 1 and not a.no_cpu_baseline:
            cb, cpu_geno = cpu_baseline(ref, pfx, a.workdir, a.pairs)
            out["cpu_baseline"] = cb
        else:
            out["cpu_baseline"] = None
        with open(os.path.join(a.workdir, "last_genotype.tsv"), "w") as f:
            f.write(text)
        print(json.dumps(out))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
