#!/bin/bash
# Same-box A/B of kernel variants through the development switches of the TEST build (libscenelib2_amd_test.so).
# usage: scripts/ab_variants.sh <tag> "<bench flags>" "VAR=val VAR2=val" "VAR=val" ...   ("-" = no switch: the defaults)
TAG=$1; FLAGS=$2; shift 2
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export SL2_LIB_PATH=$PWD/scenelib2_amd/libscenelib2_amd_test.so
i=0
for rep in 1 2; do
for sw in "$@"; do
  i=$((i+1))
  [ "$sw" = "-" ] && sw=""
  env $sw timeout 600 python bench.py $FLAGS --cpu-sample 0 2>$OUT/ab_$i.err | tail -1 > $OUT/ab_$i.json
  python - <<PY
import json
d=json.load(open("$OUT/ab_$i.json"))
print("[%s]" % "$sw", round(d['value']), round(d['ms_per_step'],4), {k:round(v['ms_per_step'],4) for k,v in list(d["kernels"].items())[:int(__import__("os").environ.get("NK","6"))]})
PY
done; done | tee -a $OUT/ab_summary.txt
