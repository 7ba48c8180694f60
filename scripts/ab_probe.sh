# A/B of development builds: scripts/ab_probe.sh "<bench flags>" lib1.so lib2.so ...   (run on the GPU box)
FLAGS=$1; shift
run() { SL2_LIB_PATH=$2 timeout 300 python bench.py $FLAGS --cpu-sample 0 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],4), {k:round(v['ms_per_step'],3) for k,v in list(d['kernels'].items())[:${NK:-6}]})"; }
for l in "$@"; do run $(basename $l) $l; done
