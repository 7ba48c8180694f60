#!/bin/bash
# Round 3: the lean matrix-core search (variant 4) against the round-2 kernel (variant 3): parity tests, bench lines,
# instruction counters.  Usage: scripts/r3_search_ab.sh [tag]
TAG=${1:-r03_search}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_slam.py -q -x --timeout=600 -k "search or near_ties or sequences_track or adversarial" > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
run() { timeout 300 python bench.py --cpu-sample 0 2>$OUT/bench_$1.err | tail -1 > $OUT/bench_$1.json; python -c "import sys,json; d=json.load(open('$OUT/bench_$1.json')); print('$1', round(d['value']), round(d['ms_per_step'],4), 'search', round(d['kernels']['k_search']['ms_per_step'],4), 'fallbacks', d['work_per_step']['search_fallbacks'], 'parity', d.get('parity'))"; }
SL2_SEARCH_VARIANT=3 run v3
SL2_SEARCH_VARIANT=4 run v4
SL2_SEARCH_VARIANT=3 run v3b
SL2_SEARCH_VARIANT=4 run v4b
i=0
for v in 3 4; do
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  ( cd /tmp && SL2_SEARCH_VARIANT=$v timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$i -o bench -- python $OLDPWD/bench.py --steps 3 --warmup 30 --cpu-sample 0 --no-profile > $OUT/pmc_$i.json 2> $OUT/pmc_$i.err ); echo "pmc group $i (variant $v) exit $?"
done
done
python scripts/summarize_pmc.py $OUT 2>&1 | grep -E "^==|k_search" | tee $OUT/pmc_search_summary.txt
