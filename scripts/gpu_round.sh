#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprofv3 summaries.  Everything
# lands under gpurun_out/ (merged back by gpurun).  Usage: scripts/gpu_round.sh [tag] [stages]
TAG=${1:-r01}
STAGES=${2:-"tests smoke bench prof"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
echo "== env" > $OUT/env.txt
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -12; nproc; free -g | head -2; python -c "import torch;print(torch.__version__, torch.cuda.is_available())") >> $OUT/env.txt 2>&1
for st in $STAGES; do
case $st in
tests)
  timeout 900 python -m pytest tests -q -m gpu -x --timeout=600 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
  tail -25 $OUT/pytest_gpu.log ;;
testsall)
  timeout 1200 python -m pytest tests -q -m gpu --timeout=600 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
  tail -40 $OUT/pytest_gpu.log ;;
smoke)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
  tail -5 $OUT/smoke.log ;;
benchsmall)
  timeout 600 python bench.py --batch 64 --steps 3 --warmup 1 --cpu-sample 2 > $OUT/bench_small.json 2> $OUT/bench_small.err; echo "bench small exit $?"
  tail -c 3000 $OUT/bench_small.json; tail -5 $OUT/bench_small.err ;;
bench)
  timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
  tail -c 4000 $OUT/bench.json; tail -5 $OUT/bench.err ;;
prof)
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o bench -- python $OLDPWD/bench.py --cpu-sample 0 --no-profile > $OLDPWD/$OUT/prof_bench.json 2> $OLDPWD/$OUT/prof.err ); echo "prof exit $?"
  find $OUT/prof -name "*stats*" | head; for f in $(find $OUT/prof -name "*kernel_stats.csv" | head -1); do head -30 $f; done ;;
pmc)
  # separate passes (PMC only with --kernel-trace; never with sys/runtime traces); 30 warm-up steps so that the last three
  # dispatches of every kernel are steady-state steps (PMC_LAST=3)
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    ( cd /tmp && timeout 900 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OLDPWD/$OUT/pmc_$i -o bench -- python $OLDPWD/bench.py --steps 3 --warmup 30 --cpu-sample 0 --no-profile > $OLDPWD/$OUT/pmc_$i.json 2> $OLDPWD/$OUT/pmc_$i.err ); echo "pmc group $i ($grp) exit $?"
  done
  PMC_LAST=3 PMC_GIT=${PMC_GIT:-$(cat .build_git 2>/dev/null || echo unknown)} python scripts/summarize_pmc.py $OUT $OUT/pmc_traffic.json > $OUT/pmc_summary.txt 2>&1; tail -80 $OUT/pmc_summary.txt ;;
c1)
  # BASELINE configs[1]: one sequence; parity over every stepped frame and the CPU oracle on one host thread in the same line
  timeout 900 python bench.py --batch 1 --steps 200 --warmup 50 > $OUT/bench_c1.json 2> $OUT/bench_c1.err; echo "bench c1 exit $?"; tail -c 1500 $OUT/bench_c1.json
  timeout 600 python bench.py --batch 1 --steps 200 --warmup 50 --no-profile --cpu-sample 0 > $OUT/bench_c1_noprofile.json 2> /dev/null; tail -c 300 $OUT/bench_c1_noprofile.json ;;
mapping)
  timeout 1200 python bench.py --mapping > $OUT/bench_mapping.json 2> $OUT/bench_mapping.err; echo "bench mapping exit $?"; tail -c 2500 $OUT/bench_mapping.json; tail -3 $OUT/bench_mapping.err ;;
adapter)
  make -s -C examples > /dev/null 2>&1
  timeout 900 python scripts/adapter_latency.py $OUT/adapter_latency.json > $OUT/adapter_latency.log 2>&1; echo "adapter exit $?"; tail -5 $OUT/adapter_latency.log ;;
rccl1)
  SL2_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 > $OUT/rccl_single_rank.json 2> $OUT/rccl_single_rank.err; echo "rccl1 exit $?"
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 > $OUT/plain_single_rank.json 2> /dev/null
  python -c "
import json
a=json.load(open('$OUT/rccl_single_rank.json')); b=json.load(open('$OUT/plain_single_rank.json'))
print('rccl', round(a['value']), a['ms_per_step'], 'plain', round(b['value']), b['ms_per_step'])" ;;
tworanks)
  # two ranks sharing the one GPU of the box (gloo in place of RCCL: the test hook), self-spawned and under the driver's own launch line
  SL2_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --batch 512 --steps 40 --warmup 10 > $OUT/two_ranks_one_gpu.json 2> $OUT/two_ranks_one_gpu.err; echo "two ranks (self-spawned) exit $?"
  SL2_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --batch 512 --steps 20 --warmup 5 > $OUT/torchrun_two_ranks_one_gpu.json 2> $OUT/torchrun_two_ranks_one_gpu.err; echo "two ranks (torch.distributed.run) exit $?"
  for f in two_ranks_one_gpu torchrun_two_ranks_one_gpu; do python -c "
import json
d=json.loads([l for l in open('$OUT/$f.json') if l.strip().startswith('{')][0]); print('$f', round(d['value']), d['ms_per_step'], d['n_gpus'], d['parity'].get('ranks_checked'), d['parity']['traj_rmse_vs_oracle'])"; done ;;
driver)
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench driver-protocol exit $?"; tail -c 1200 $OUT/bench_driver.json ;;
c4|c5)
  # BASELINE configs[3] / configs[4] per GPU: bench line, rocprofv3 kernel stats, FETCH / WRITE passes
  if [ $st = c4 ]; then SHAPE="--width 640 --height 480 --features 200 --batch 1024 --steps 20 --warmup 10"; else SHAPE="--width 1280 --height 720 --features 500 --batch 512 --steps 10 --warmup 6"; fi
  timeout 1200 python bench.py $SHAPE > $OUT/bench_$st.json 2> $OUT/bench_$st.err; echo "bench $st exit $?"; tail -c 600 $OUT/bench_$st.json
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_$st -o bench -- python $OLDPWD/bench.py $SHAPE --cpu-sample 0 --no-profile > $OLDPWD/$OUT/prof_$st.json 2> $OLDPWD/$OUT/prof_$st.err ); echo "prof $st exit $?"
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    ( cd /tmp && timeout 900 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OLDPWD/$OUT/${st}/pmc_$i -o bench -- python $OLDPWD/bench.py $SHAPE --steps 2 --warmup 1 --cpu-sample 0 --no-profile > $OLDPWD/$OUT/${st}_pmc_$i.json 2> $OLDPWD/$OUT/${st}_pmc_$i.err ); echo "pmc $st group $i exit $?"
  done
  python scripts/summarize_pmc.py $OUT/$st $OUT/${st}_pmc_traffic.json > $OUT/${st}_pmc_summary.txt 2>&1; grep -E "k_syrk|k_build|k_fwd|k_chol|k_search_mfma" $OUT/${st}_pmc_summary.txt | head -20 ;;
esac
done
ls -la $OUT | head -30
