#!/bin/bash
# Copies the artefacts of one scripts/gpu_round.sh visit (gpurun_out/<tag>: scratch) to profiles/<prefix>_* (tracked) and refreshes
# the figures DESIGN.md quotes from them.  Usage: scripts/publish_round.sh <tag> [prefix]   e.g. scripts/publish_round.sh r06_final3 r06_final
set -e
cd "$(dirname "$0")/.."
TAG=$1; PRE=${2:-$1}; G=gpurun_out/$TAG; P=profiles
for f in bench bench_driver bench_c1 bench_c1_noprofile bench_c4 bench_c5 bench_mapping rccl_single_rank plain_single_rank two_ranks_one_gpu torchrun_two_ranks_one_gpu; do
  [ -f $G/$f.json ] && cp $G/$f.json $P/${PRE}_$f.json
done
[ -f $G/pmc_summary.txt ] && cp $G/pmc_summary.txt $P/${PRE}_pmc_summary.txt
[ -f $G/pmc_traffic.json ] && cp $G/pmc_traffic.json $P/${PRE}_pmc_traffic.json && cp $G/pmc_traffic.json $P/pmc_traffic.json
for c in c4 c5; do
  [ -f $G/${c}_pmc_summary.txt ] && cp $G/${c}_pmc_summary.txt $P/${PRE}_${c}_pmc_summary.txt
  [ -f $G/${c}_pmc_traffic.json ] && cp $G/${c}_pmc_traffic.json $P/${PRE}_${c}_pmc_traffic.json
done
[ -f $G/pytest_gpu.log ] && cp $G/pytest_gpu.log $P/${PRE}_pytest_gpu.log
[ -f $G/env.txt ] && cp $G/env.txt $P/${PRE}_env.txt
for t in prof:$PRE prof_c4:${PRE}_c4 prof_c5:${PRE}_c5; do
  d=${t%%:*}; n=${t##*:}; f=$(find $G/$d -name "*kernel_stats.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" $P/${n}_kernel_stats.csv
done
R=${PRE%_final}
[ -f $G/adapter_latency.json ] && cp $G/adapter_latency.json $P/${R}_adapter_latency.json
[ -f $G/adapter_latency_copy_frames.json ] && cp $G/adapter_latency_copy_frames.json $P/${R}_adapter_latency_copy_frames.json
[ -f $G/small_latency.json ] && cp $G/small_latency.json $P/${R}_small_latency.json
python scripts/refresh_design.py
