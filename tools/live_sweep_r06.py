"""Round 6: fresh random samples (reference kind, genes, pairs, read length, error rates, flags) through this build's genotyper and the REFERENCE binary
(oracle/_ref/genotyper), every output file byte for byte.  usage: python tools/live_sweep_r06.py [runs] [seed]   (on a GPU box; needs oracle/_ref)"""
import os, random, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import util
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 25
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
exe = os.path.join(util.ROOT, "t1k_amd", "bin", "genotyper")
bad = 0
for i in range(runs):
    tmp = tempfile.mkdtemp(prefix="live_")
    kind = rng.choice(["ref-rna", "ref-dna"])
    ref = os.path.join(tmp, "ref.fa")
    util.synth_ref(kind, ref, genes=rng.randint(2, 8), scale=rng.choice([0.03, 0.08, 0.2]), seed=rng.randint(1, 10**6))
    pairs = rng.randint(500, 9000)
    L = rng.choice([75, 100, 125, 150, 151])
    kw = dict(pairs=pairs, len=L, seed=rng.randint(1, 10**6), sub=rng.choice([0.0, 0.002, 0.01, 0.03]), indel=rng.choice([0.0, 0.00005, 0.001]))
    util.synth_reads(ref, os.path.join(tmp, "r"), **kw)
    flags = rng.choice([["-s", "0.97"], ["-s", "0.9"], ["-s", "0.8"], ["-s", "0.9", "--relaxIntronAlign"], ["-s", "0.97", "-n", "30"]])
    paired = rng.random() < 0.8
    files = ["-1", os.path.join(tmp, "r_1.fq"), "-2", os.path.join(tmp, "r_2.fq")] if paired else ["-u", os.path.join(tmp, "r_1.fq")]
    args = ["-f", ref] + files + flags + ["--outputReadAssignment"]
    a = subprocess.run([exe] + args + ["-o", os.path.join(tmp, "ours")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    b = subprocess.run([util.REF_BIN] + args + ["-t", "32", "-o", os.path.join(tmp, "ref")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    why = ""
    if a.returncode != 0 or b.returncode != 0: why = "rc %d / %d: %s" % (a.returncode, b.returncode, a.stderr.strip().splitlines()[-1][:160] if a.stderr.strip() else "")
    else:
        for fn in sorted(os.listdir(tmp)):
            if fn.startswith("ref_") and open(os.path.join(tmp, fn), "rb").read() != open(os.path.join(tmp, "ours_" + fn[4:]), "rb").read(): why = fn + " differs"; break
    if why:
        bad += 1
        print("FAIL %s %s pairs=%d len=%d %s %s -> %s" % (kind, kw, pairs, L, flags, "paired" if paired else "single", why), flush=True)
    subprocess.run(["rm", "-rf", tmp])
print("%d live runs against the reference binary, %d failed" % (runs, bad))
sys.exit(1 if bad else 0)
