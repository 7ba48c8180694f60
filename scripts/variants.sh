cd /root/repo
# A/B of kernel variants through the env overrides read by sl2_create of the TEST build (development aid; the product
# library reads no environment).
export SL2_LIB_PATH=/root/repo/scenelib2_amd/libscenelib2_amd_test.so
# usage: VAR=SL2_CHOL_VARIANT VALUES="1 0" bash scripts/variants.sh
VAR=${VAR:-SL2_FWD_VARIANT}
for v in ${VALUES:-0 1}; do
  echo "$VAR=$v"; env $VAR=$v python bench.py --cpu-sample ${CPU_SAMPLE:-4} 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity']); print({k:round(v['ms_per_step'],3) for k,v in d['kernels'].items()})"
done
