# 2 x 250 bp and 2 x 300 bp reads end to end: this build (1 GPU, 2 ranks, tiny windows) against the reference binary
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT + "/tests")
import util
W = "/tmp/t1k_long"; os.makedirs(W, exist_ok=True)
res = 0
for kind, L, flags in (("ref-rna", 250, ["-s", "0.9"]), ("ref-dna", 300, ["-s", "0.9", "--relaxIntronAlign"]), ("ref-rna", 320, ["-s", "0.97"])):
    ref = W + "/ref_%s.fa" % kind
    util.synth_ref(kind, ref, genes=6, scale=0.2, seed=77)
    pfx = W + "/r%d" % L
    util.synth_reads(ref, pfx, pairs=6000, len=L, seed=5, sub=0.008, fragmean=2 * L + 40)
    args = ["-f", ref, "-1", pfx + "_1.fq", "-2", pfx + "_2.fq"] + flags
    a = subprocess.run([ROOT + "/oracle/_ref/genotyper"] + args + ["-t", "32", "-o", W + "/ref"], stderr=subprocess.PIPE, text=True)
    for tag, env in (("1gpu", {}), ("2ranks", {"T1K_GPUS": "0,0"}), ("tiny-windows", {"T1K_FIRST_WINDOW": "64", "T1K_WINDOW": "1500", "T1K_BATCH": "128", "T1K_PAIR_BATCH": "256"})):
        b = subprocess.run([ROOT + "/t1k_amd/bin/genotyper"] + args + ["-o", W + "/gpu"], stderr=subprocess.PIPE, text=True, env=dict(os.environ, **env))
        print(kind, L, tag, "rc", a.returncode, b.returncode, end=" ")
        for s in ("_genotype.tsv", "_allele.tsv", "_aligned_1.fa", "_aligned_2.fa"):
            same = os.path.exists(W + "/ref" + s) and os.path.exists(W + "/gpu" + s) and open(W + "/ref" + s, "rb").read() == open(W + "/gpu" + s, "rb").read()
            print(s, "ok" if same else "DIFF", end=" "); res |= 0 if same else 1
        print(os.path.getsize(W + "/gpu_aligned_1.fa"))
        if b.returncode: print(b.stderr[-300:])
sys.exit(res)
