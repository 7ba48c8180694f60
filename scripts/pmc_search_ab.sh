#!/bin/bash
# PMC counters of k_search_mfma for several builds of the library (one pass per counter group, --kernel-trace only).
# usage: scripts/pmc_search_ab.sh OUTDIR "libA.so libB.so"
cd "$(dirname "$0")/.."
OUT=$1; LIBS=$2
mkdir -p $OUT
export TMPDIR=/tmp
for lib in $LIBS; do
  tag=${lib%.so}
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    ( cd /tmp && SL2_LIB_PATH=$OLDPWD/scenelib2_amd/$lib timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OLDPWD/$OUT/$tag/pmc_$i -o bench -- python $OLDPWD/bench.py --steps 3 --warmup 30 --cpu-sample 0 --no-profile > /dev/null 2> $OLDPWD/$OUT/${tag}_pmc_$i.err )
  done
  echo "#### $lib"
  PMC_LAST=3 python scripts/summarize_pmc.py $OUT/$tag | grep -E "^==|k_search_mfma"
done
