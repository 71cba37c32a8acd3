"""Round 6: reproduce a failing case of tests/test_gpu_fuzz.py::test_adversarial_pairs_executable_vs_reference_binary outside pytest, with the
executable's stderr shown.  usage: T1K_FUZZ_SEED=644 python tools/repro_fuzz_r06.py ref-dna 8 -s 0.9 --relaxIntronAlign"""
import os, random, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import util
import test_gpu_fuzz as fz
kind, seed, flags = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
tmp = tempfile.mkdtemp(prefix="fz_")
ref = os.path.join(tmp, "ref.fa")
util.synth_ref(kind, ref, seed=seed + fz.SEED0, genes=5, scale=0.15)
pairs = fz.adversarial_pairs(fz.alleles(ref), random.Random(seed + fz.SEED0), 6000 * fz.SCALE)
for i, suffix in enumerate(("_1.fq", "_2.fq")):
    with open(os.path.join(tmp, "p") + suffix, "w") as f:
        for j, pr in enumerate(pairs):
            f.write("@f%d/%d\n%s\n+\n%s\n" % (j, i + 1, pr[i], "I" * len(pr[i])))
args = ["-f", ref, "-1", os.path.join(tmp, "p_1.fq"), "-2", os.path.join(tmp, "p_2.fq")] + flags + ["--outputReadAssignment"]
exe = os.path.join(util.ROOT, "t1k_amd", "bin", "genotyper")
r = subprocess.run([exe] + args + ["-o", os.path.join(tmp, "ours")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
print("rc", r.returncode)
print(r.stderr[-3000:])
