#!/bin/bash
# Round 5: the sharded job with EIGHT ranks on ONE device (in-process transport), 8 x 0.5 M pairs = 4 M pairs in all (eight ranks' arenas and kept read sets have to share ONE device's 288 GB), next to the
# single-rank run of the same sample: every phase line of rank 0 (T1K_DEBUG_PHASES) -> the serial / replicated terms of the N = 8 path
# (row exchange, owner-side coalescing, group gather, replicated classes / EM / selection) measured instead of argued.  The device loop
# of this run is NOT a scaling figure (eight ranks share one GPU); the phases after it are what is read.  Run on the GPU box.
#   tools/ranks8_r05.sh [pairs per rank]      -> gpurun_out/r05_ranks8_inprocess.log
W=/tmp/t1k_bench; P=${1:-500000}; R=8
mkdir -p gpurun_out
for i in $(seq 0 $((R - 1))); do python -c "import bench; bench.ensure_inputs('$W', $P, 24, 1.0, seed=$((2 + i)))" || exit 1; done
ARGS="-f $W/hla_g24_s1.0.fa"
for i in $(seq 0 $((R - 1))); do ARGS="$ARGS -1 $W/reads_g24_s1.0_p${P}_seed$((2 + i))_1.fq -2 $W/reads_g24_s1.0_p${P}_seed$((2 + i))_2.fq"; done
LOG=gpurun_out/r05_ranks8_inprocess.log
: > $LOG
run() {  # label, env ...
  local label=$1; shift
  local t0=$(date +%s%N)
  env "$@" T1K_DEBUG_PHASES=1 t1k_amd/bin/genotyper $ARGS -s 0.97 -o $W/r8_$label 2> $W/r8_$label.err
  local rc=$?
  echo "== $label ($*): rc $rc, $(( ($(date +%s%N) - t0) / 1000000 )) ms wall, genotype md5 $(md5sum < $W/r8_${label}_genotype.tsv | cut -c1-8), allele md5 $(md5sum < $W/r8_${label}_allele.tsv | cut -c1-8), aligned_1 md5 $(md5sum < $W/r8_${label}_aligned_1.fa | cut -c1-8)" >> $LOG
  grep "^\[t1k job\]\|^\[t1k host\]\|^genotyper\|^\[t1k\] coalesce\|Finish\|error\|Abort" $W/r8_$label.err | cut -c1-420 >> $LOG
}
run one_rank T1K_PIPELINES=3
run eight_ranks_exact T1K_GPUS=0,0,0,0,0,0,0,0 T1K_PIPELINES=1 T1K_BATCH=8192
run eight_ranks_own_input T1K_GPUS=0,0,0,0,0,0,0,0 T1K_PIPELINES=1 T1K_BATCH=8192 T1K_SHARD_INPUT=1
run eight_ranks_allreduce T1K_GPUS=0,0,0,0,0,0,0,0 T1K_PIPELINES=1 T1K_BATCH=8192 T1K_EM_COLLECTIVE=allreduce
cat $LOG
