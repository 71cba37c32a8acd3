#!/bin/bash
# BASELINE configs[2] at size against the committed md5 sums of the REFERENCE's files for this input (tests/golden/full_size_md5.json "kir_10M",
# made by tools/kir_10M_parity_r03.sh): 10 M 2x150 bp pairs, KIR-like dna reference, -s 0.9 --relaxIntronAlign.  Run on the GPU box.
cd "$(dirname "$0")/.."
W=/tmp/t1k_kir; mkdir -p $W gpurun_out
L=gpurun_out/r04_kir_10M_check.log; : > $L
tools/t1k_synth ref-dna --genes 17 --scale 1.0 --seed 20250614 > $W/kir.fa
tools/t1k_synth reads --ref $W/kir.fa --pairs 10000000 --len 150 --seed 3 --out $W/k
A="-f $W/kir.fa -1 $W/k_1.fq -2 $W/k_2.fq -s 0.9 --relaxIntronAlign"
for mode in deferred eager two_ranks; do
  E=""; [ $mode = eager ] && E="T1K_COVERAGE=eager"; [ $mode = two_ranks ] && E="T1K_GPUS=0,0"
  t0=$(date +%s%N); env $E t1k_amd/bin/genotyper $A -o $W/o_$mode 2> $W/o_$mode.log; rc=$?
  echo "$mode: rc $rc, $(( ($(date +%s%N) - t0) / 1000000 )) ms wall" >> $L
  python3 - $W/o_$mode >> $L <<'PY'
import hashlib, json, sys
want = json.load(open("tests/golden/full_size_md5.json"))["kir_10M"]
for suf in ("_genotype.tsv", "_allele.tsv", "_aligned_1.fa", "_aligned_2.fa"):
    h = hashlib.md5()
    with open(sys.argv[1] + suf, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""): h.update(blk)
    print("   %s %s %s" % (suf, h.hexdigest(), "identical to the reference's" if h.hexdigest() == want[suf] else "DIFFERS from the reference's " + want[suf]))
PY
done
cat $L
