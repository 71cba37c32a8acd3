# phase split of k_pair (T1K_PAIR_PROFILE build on the box: s_memtime ticks of thread 0 of every workgroup, summed over the launches), 1 M pairs, one pipeline
mkdir -p gpurun_out /tmp/t1k_bench
W=/tmp/t1k_bench; LOG=gpurun_out/r05_pair_phases.log; : > $LOG
python -c "import bench; bench.ensure_inputs('$W', 1000000, 24, 1.0, seed=2)"
( cd t1k_amd/csrc && touch t1k_pair.hip && make -j8 EXTRA=-DT1K_PAIR_PROFILE > /tmp/make_pp.log 2>&1 ) || { tail -5 /tmp/make_pp.log >> $LOG; exit 1; }
T1K_PIPELINES=1 t1k_amd/bin/genotyper -f $W/hla_g24_s1.0.fa -1 $W/reads_g24_s1.0_p1000000_seed2_1.fq -2 $W/reads_g24_s1.0_p1000000_seed2_2.fq -s 0.97 -o $W/pp 2> $W/pp.err
python - $W/pp.err >> $LOG <<'PY'
import re, sys
names = ["tables", "join", "best", "keep", "rules", "setreads", "rank", "rows+hash"]
tot = [0] * 8; n = 0
for l in open(sys.argv[1]):
    if "pair phases" in l:
        v = [int(x) for x in re.findall(r"(?:tables|join|best|keep|rules|setreads|rank|rows\+hash) (\d+)", l)]
        if len(v) == 8:
            tot = [a + b for a, b in zip(tot, v)]; n += 1
s = sum(tot) or 1
print("k_pair phases over %d launches (share of thread-0 ticks):" % n)
for a, b in zip(names, tot): print("  %-10s %5.1f %%" % (a, 100.0 * b / s))
PY
cat $LOG
