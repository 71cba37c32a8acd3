# one job, reads loaded from files once, run three times with the output writer behind the loop (which releases the mapped input):
# every run must write the same files; then --gpus beyond the device count must fail with a message
import os, sys, hashlib, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import t1k_amd, util, bench
os.environ["T1K_FIRST_WINDOW"] = "4096"; os.environ["T1K_WINDOW"] = "60000"
ref, pfx = bench.ensure_inputs("/tmp/t1k_bench", 200000, 24, 1.0, seed=2)
job = t1k_amd.Job(ref, ref_seq_similarity=0.97)
job.load_reads(pfx + "_1.fq", pfx + "_2.fq")
sums = []
for i in range(3):
    out = "/tmp/t1k_bench/rerun%d" % i
    job.set_output_prefix(out); job.run(); job.write_outputs(out)
    sums.append([hashlib.md5(open(out + s, "rb").read()).hexdigest()[:8] for s in ("_genotype.tsv", "_allele.tsv", "_aligned_1.fa", "_aligned_2.fa")])
    print(i, sums[-1], job.counts()["assigned_fragments"])
job.close()
assert sums[0] == sums[1] == sums[2]
r = subprocess.run([ROOT + "/t1k_amd/bin/genotyper", "-f", ref, "-1", pfx + "_1.fq", "-2", pfx + "_2.fq", "--gpus", "3", "-o", "/tmp/t1k_bench/toomany"], stderr=subprocess.PIPE, text=True)
print("--gpus 3 on one GPU: rc", r.returncode, r.stderr.strip().split("\n")[-1][:200])
assert r.returncode != 0
