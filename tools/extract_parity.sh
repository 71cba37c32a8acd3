#!/bin/bash
# Candidate extraction through the executable (t1k_amd/bin/fastq-extractor) against the reference's own fastq-extractor
# (oracle/_ref/fastq-extractor, built by oracle/Makefile from the reference's sources): every output file must be identical.
# Synthetic references / reads from tools/t1k_synth (seeded).  Usage: tools/extract_parity.sh [pairs]
set -e
cd "$(dirname "$0")/.."
W=${W:-/tmp/extract_parity}; mkdir -p $W
P=${1:-200000}
fail=0
run() { # name, args...
  n=$1; shift
  timeout 300 t1k_amd/bin/fastq-extractor "$@" -o $W/ours_$n > $W/ours_$n.log 2>&1 || { echo "ours rc=$? ($n)"; tail -2 $W/ours_$n.log; }
  timeout 300 oracle/_ref/fastq-extractor "$@" -o $W/ref_$n > $W/ref_$n.log 2>&1 || echo "ref rc=$? ($n)"
  ok=1; cnt=0
  for f in $W/ref_${n}[._]*; do case $f in *.log) continue;; esac
    o=$W/ours_${n}${f#$W/ref_${n}}; cnt=$((cnt+1)); cmp -s $o $f || { ok=0; fail=1; echo "DIFF $n ${f##*/}"; }; done
  if [ $ok = 1 ]; then echo "$n: $cnt files identical, $(grep -c '^[@>]' $(ls $W/ref_${n}[._]*f[qa] | head -1)) reads kept"; fi
}
tools/t1k_synth ref-rna --seed 31 --genes 8 --scale 0.3 > $W/rna.fa
tools/t1k_synth ref-dna --seed 32 --genes 6 --scale 0.2 > $W/dna.fa
tools/t1k_synth reads --ref $W/rna.fa --out $W/a --seed 33 --pairs $P --len 100 --bg 0.6 --barcodes 500
tools/t1k_synth reads --ref $W/rna.fa --out $W/b --seed 34 --pairs $P --len 150 --bg 0.3 --sub 0.12 --indel 0.01 --nrate 0.02
tools/t1k_synth reads --ref $W/dna.fa --out $W/c --seed 35 --pairs $P --len 250 --bg 0.5 --sub 0.03
tools/t1k_synth reads --ref $W/rna.fa --out $W/d --seed 36 --pairs $((P/4)) --len 75 --bg 0.5 --fasta
run rna100 -f $W/rna.fa -1 $W/a_1.fq -2 $W/a_2.fq
run rna100_t8 -f $W/rna.fa -1 $W/a_1.fq -2 $W/a_2.fq -t 8
run rna100_bc -f $W/rna.fa -1 $W/a_1.fq -2 $W/a_2.fq --barcode $W/a_bc.fa --barcodeStart 2 --barcodeEnd 13 -t 4
# (reverse-complementing the literal "missing_barcode" indexes the reference's nucToNum out of bounds: only the head of the file, which has none)
head -400 $W/a_1.fq > $W/h_1.fq; head -400 $W/a_2.fq > $W/h_2.fq; head -200 $W/a_bc.fa > $W/h_bc.fa
run rna100_bcrc -f $W/rna.fa -1 $W/h_1.fq -2 $W/h_2.fq --barcode $W/h_bc.fa --barcodeStart 1 --barcodeEnd 14 --barcodeRevComp
# barcode whitelist correction (BarcodeCorrector.hpp): whitelist = the barcodes in use + decoys one substitution apart; the barcode reads get
# substitutions, N and varying qualities (records of the synthetic file that say "missing_barcode" are replaced: the reference's trie
# indexes nucToNum out of bounds on lower-case letters)
python3 - $W <<'PYEOF'
import random, sys
W = sys.argv[1]; rng = random.Random(3)
recs = [l.strip() for l in open(W + "/a_bc.fa")]
ids, bcs = recs[0::2], recs[1::2]
rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
L = len(next(b for b in bcs if b != "missing_barcode"))
wl = sorted(set(b for b in bcs if b != "missing_barcode"))
decoys = []
for b in wl[:200]:
    i = rng.randrange(L); decoys.append(b[:i] + rng.choice([c for c in "ACGT" if c != b[i]]) + b[i + 1:])
comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
open(W + "/wl.txt", "w").write("\n".join(wl + decoys + [rnd(L) for _ in range(300)]) + "\n")
open(W + "/wl_rc.txt", "w").write("\n".join("".join(comp[c] for c in reversed(b[1:L - 1])) for b in wl + decoys) + "\n")
with open(W + "/a_bcq.fq", "w") as f:
    for i, b in zip(ids, bcs):
        if b == "missing_barcode": b = rnd(L)
        t = list(b); u = rng.random()
        for _ in range(1 if u < 0.3 else 2 if u < 0.36 else 0):
            j = rng.randrange(L); t[j] = rng.choice("ACGTN")
        q = "".join(chr(rng.randrange(35, 74)) for _ in range(L))
        f.write("@%s\n%s\n+\n%s\n" % (i[1:], "".join(t), q))
PYEOF
run rna100_wl -f $W/rna.fa -1 $W/a_1.fq -2 $W/a_2.fq --barcode $W/a_bcq.fq --barcodeWhitelist $W/wl.txt
run rna100_wl_t4 -f $W/rna.fa -1 $W/a_1.fq -2 $W/a_2.fq --barcode $W/a_bcq.fq --barcodeWhitelist $W/wl.txt -t 4
run rna100_wl_rc -f $W/rna.fa -1 $W/a_1.fq -2 $W/a_2.fq --barcode $W/a_bcq.fq --barcodeWhitelist $W/wl_rc.txt --barcodeStart 1 --barcodeEnd 14 --barcodeRevComp -t 2
run rna100_wl_prefix -f $W/rna.fa -1 $W/a_1.fq -2 $W/a_2.fq --barcode $W/a_bcq.fq --barcodeWhitelist $W/wl.txt --barcodeEnd 11
run rna150_noisy -f $W/rna.fa -1 $W/b_1.fq -2 $W/b_2.fq -t 4
run rna150_s95 -f $W/rna.fa -1 $W/b_1.fq -2 $W/b_2.fq -s 0.95
run rna150_single -f $W/rna.fa -u $W/b_1.fq -s 0.97 --read1Start 5 --read1End 120
run dna250 -f $W/dna.fa -1 $W/c_1.fq -2 $W/c_2.fq -t 4
run dna250_single -f $W/dna.fa -u $W/c_2.fq
run fasta75 -f $W/rna.fa -1 $W/d_1.fa -2 $W/d_2.fa -t 2
gzip -c $W/b_1.fq > $W/bz_1.fq.gz; gzip -c $W/b_2.fq > $W/bz_2.fq.gz
run gz -f $W/rna.fa -1 $W/bz_1.fq.gz -2 $W/bz_2.fq.gz -t 4
paste -d '\n' <(paste - - - - < $W/a_1.fq) <(paste - - - - < $W/a_2.fq) | tr '\t' '\n' > $W/il.fq
run interleaved -f $W/rna.fa -i $W/il.fq
run interleaved_t4 -f $W/rna.fa -i $W/il.fq -t 4
run two_files -f $W/rna.fa -1 $W/a_1.fq -1 $W/b_1.fq -2 $W/a_2.fq -2 $W/b_2.fq -t 2
run cross -f $W/dna.fa -1 $W/b_1.fq -2 $W/b_2.fq -t 4
if [ $fail = 0 ]; then echo "extract parity: all identical"; fi
exit $fail
