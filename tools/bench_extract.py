#!/usr/bin/env python3
"""Candidate-read extraction on one MI355X: fragments screened per second by t1k_extract_batch with the batch
packed and resident in HBM, the algorithmic-byte roofline fraction of k_extract, and the reference's own fastq-extractor
(oracle/_ref/fastq-extractor) timed on the host cores beside it.  Prints ONE JSON line, same shape as bench.py.

  python tools/bench_extract.py [--pairs 4000000] [--steps 5] [--warmup 1] [--bg 0.97]

Workload: synthetic 2x150 bp pairs of which a fraction --bg are background (random sequence, the bulk of a real sequencing run) and the
rest come from the synthetic HLA-like rna reference bench.py uses (24 genes, scale 1)."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md
# HBM traffic of one launch per read pair of the default workload (profiles/r01_extract_pmc.md: FETCH_SIZE + WRITE_SIZE for 4 M pairs, as
# reported; on gfx950 FETCH_SIZE is calibrated only for wide coalesced loads, these kernels issue 4-byte-per-lane loads)
READ_LEN = 150
TRAFFIC_BYTES_PER_PAIR = {"k_extract_screen": (6.861e9 + 3.926e8) / 4e6, "k_extract": (1.177e10 + 1.163e7) / 4e6}


def records(path, n=None):
    out = []
    with open(path) as f:
        for i, line in enumerate(f):
            if i % 4 == 1:
                out.append(line.rstrip("\n"))
                if n and len(out) >= n:
                    break
    return out


def ref_seqs(path):
    out, cur = [], []
    for l in open(path):
        if l[0] == ">":
            if cur:
                out.append("".join(cur))
            cur = []
        else:
            cur.append(l.strip())
    if cur:
        out.append("".join(cur))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=4000000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bg", type=float, default=0.97)
    ap.add_argument("--genes", type=int, default=24)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--workdir", default=os.environ.get("T1K_BENCH_DIR", "/tmp/t1k_bench"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    import numpy as np
    import t1k_amd
    synth = os.path.join(ROOT, "tools", "t1k_synth")
    os.makedirs(a.workdir, exist_ok=True)
    ref = os.path.join(a.workdir, "hla_g%d_s%s.fa" % (a.genes, a.scale))
    if not os.path.exists(ref):
        with open(ref + ".tmp", "w") as f:
            subprocess.run([synth, "ref-rna", "--genes", str(a.genes), "--scale", str(a.scale), "--seed", "20250614"], stdout=f, check=True)
        os.replace(ref + ".tmp", ref)
    pfx = os.path.join(a.workdir, "xreads_p%d_bg%s" % (a.pairs, a.bg))
    if not os.path.exists(pfx + "_2.fq"):
        subprocess.run([synth, "reads", "--ref", ref, "--pairs", str(a.pairs), "--len", str(READ_LEN), "--seed", "77", "--bg", str(a.bg), "--out", pfx], check=True)
    rs = ref_seqs(ref)
    # parameters as FastqExtractor.cpp:383-416 derives them
    total = sum(len(s) for s in rs)
    k = 1
    while total:
        k += 1
        total //= 4
    k = max(9, k)
    hit_len = max(27, READ_LEN // 5, k)
    r1, r2 = records(pfx + "_1.fq"), records(pfx + "_2.fq")
    seqs = [s for pr in zip(r1, r2) for s in pr]
    ctx = t1k_amd.Context(kmer_length=k, hit_len_required=hit_len, ref_seq_similarity=0.8, n_base_code=0)
    ctx.ref_upload(rs)
    ctx.reads_upload(seqs)  # packed and resident in HBM before the timed region
    del seqs
    for _ in range(a.warmup):
        ctx.extract(2)
    t0 = time.time()
    for _ in range(a.steps):
        good, st = ctx.extract(2)
    dt = (time.time() - t0) / a.steps
    # ALGORITHMIC bytes per launch:
    #   k_extract_screen  every read-end: the packed words of its forward strand (bases + N mask: 2 * ceil(l/32) * 8 B) + one 4-byte presence
    #                     word per k-mer position (one look-up serves both strands)
    #   k_extract         the read-ends the screen let through: the packed words again + one 8-byte bucket header per k-mer position + 4 B per
    #                     posting of the used lists (the sequence index, which is all the vote reads; counted once)
    ends, words, npos = st["read_ends"], (READ_LEN + 31) // 32, 2 * (READ_LEN - k + 1)
    heavy = st["lookups"] // npos  # read-ends that reached k_extract's look-ups
    alg_screen = ends * (2 * words * 8 + (npos // 2) * 4)
    alg_main = heavy * (4 * words * 8 + npos * 8) + st["postings"] * 4
    t_screen, t_main = st["screen_ns"] * 1e-9, st["main_ns"] * 1e-9
    dom, alg, t_dom = ("k_extract", alg_main, t_main) if t_main >= t_screen else ("k_extract_screen", alg_screen, t_screen)
    out = {
        "metric": "read pairs screened per second (candidate extraction)", "value": a.pairs / dt, "unit": "read pairs/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "%d synthetic 2x150 bp pairs (%.0f %% background) vs synthetic HLA-like rna reference (%d sequences), k=%d, hitLenRequired=%d, -s 0.8; "
                               "reads packed and resident in HBM" % (a.pairs, 100 * a.bg, len(rs), k, hit_len), "kept_pairs": int(good.sum()), "device_stats": st},
        "roofline": {"bound": "hbm", "achieved": alg / t_dom / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / t_dom / 1e9 / HBM_PEAK_GBS,
                     "traffic": TRAFFIC_BYTES_PER_PAIR[dom] * a.pairs if a.bg == 0.97 else None,
                     "kernel": dom, "algorithmic_bytes_per_launch": alg, "kernel_ms": t_dom * 1e3,
                     "other": {"k_extract_screen" if dom == "k_extract" else "k_extract": {
                         "ms": (t_screen if dom == "k_extract" else t_main) * 1e3,
                         "achieved": (alg_screen / t_screen if dom == "k_extract" else alg_main / t_main) / 1e9}},
                     "note": "durations: HIP events on the context's stream around each launch (last timed step); rocprofv3 summary in profiles/r01_extract_kernel_stats.csv"},
    }
    if not a.no_cpu_baseline:
        refbin = os.path.join(ROOT, "oracle", "_ref", "fastq-extractor")
        kind, threads = "reference", min(os.cpu_count() or 1, 32)
        if not os.path.exists(refbin):
            refbin, kind, threads = os.path.join(ROOT, "oracle", "t1k_oracle_extract"), "port", 1
        n = min(a.pairs, 1000000 if kind == "reference" else 40000)
        for s in ("_1.fq", "_2.fq"):
            with open(pfx + s) as f, open(os.path.join(a.workdir, "xcpu" + s), "w") as g:
                for i, line in enumerate(f):
                    if i >= 4 * n:
                        break
                    g.write(line)
        t0 = time.time()
        subprocess.run([refbin, "-f", ref, "-1", os.path.join(a.workdir, "xcpu_1.fq"), "-2", os.path.join(a.workdir, "xcpu_2.fq"), "-t", str(threads), "-o",
                        os.path.join(a.workdir, "xcpu_out")], check=True, stderr=subprocess.DEVNULL)
        cdt = time.time() - t0
        out["cpu_baseline"] = {"value": n / cdt, "unit": "read pairs/s", "cores": threads, "kind": kind,
                               "sample": "first %d of %d pairs from FASTQ files, same reference, wall %.1f s incl. reference load and file I/O" % (n, a.pairs, cdt)}
        # the same sample through our executable end to end (files in, files out): the PCIe- and parse-inclusive rate
        t0 = time.time()
        subprocess.run([os.path.join(ROOT, "t1k_amd", "bin", "fastq-extractor"), "-f", ref, "-1", os.path.join(a.workdir, "xcpu_1.fq"), "-2",
                        os.path.join(a.workdir, "xcpu_2.fq"), "-t", str(threads), "-o", os.path.join(a.workdir, "xgpu_out")], check=True, stderr=subprocess.DEVNULL)
        gdt = time.time() - t0
        same = all(open(os.path.join(a.workdir, "xcpu_out" + s)).read() == open(os.path.join(a.workdir, "xgpu_out" + s)).read() for s in ("_1.fq", "_2.fq"))
        out["end_to_end"] = {"value": n / gdt, "unit": "read pairs/s", "sample": "same files through t1k_amd/bin/fastq-extractor, wall %.1f s incl. reference upload, parsing, PCIe, output" % gdt,
                             "identical_to_reference_output": same}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
