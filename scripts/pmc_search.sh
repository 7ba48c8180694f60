#!/bin/bash
# Instruction-mix counters of the search kernel (one rocprofv3 --pmc pass per group; --kernel-trace only).
# Usage: scripts/pmc_search.sh [tag]   -> gpurun_out/<tag>/pmc_*  + summary on stdout
TAG=${1:-pmc_search}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$i -o bench -- python $OLDPWD/bench.py --steps 3 --warmup 30 --cpu-sample 0 --no-profile > $OUT/pmc_$i.json 2> $OUT/pmc_$i.err ); echo "pmc group $i exit $?"
done
python scripts/summarize_pmc.py $OUT | grep -E "^==|k_search"
