run() { timeout 300 python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],4), {k:round(v['ms_per_step'],3) for k,v in list(d['kernels'].items())[:5]})"; }
run whole
for c in 64 128 192 256 512; do SL2_UPDATE_CHUNK=$c run chunk$c; done
