#!/bin/bash
# More SQ counters of k_search_mfma (one pass per group, --kernel-trace only): what the wavefronts wait on and which pipes run together.
# usage: scripts/pmc_search_deep.sh OUTDIR
cd "$(dirname "$0")/.."
OUT=$1; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LEVEL_WAVES SQ_WAVES SQ_THREAD_CYCLES_VALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_IFETCH SQ_IFETCH_LEVEL" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OLDPWD/$OUT/pmc_$i -o bench -- python $OLDPWD/bench.py --steps 3 --warmup 30 --cpu-sample 0 --no-profile > /dev/null 2> $OLDPWD/$OUT/pmc_$i.err ); echo "group $i exit $?"
done
PMC_LAST=3 python scripts/summarize_pmc.py $OUT | grep -E "^==|k_search_mfma|k_syrk|k_fwdsub_lds"
