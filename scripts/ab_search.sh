run() { timeout 300 python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],4), 'search', round(d['kernels']['k_search']['ms_per_step'],4), d['work_per_step']['search_fallbacks'])"; }
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "search or near_ties" 2>&1 | tail -1
SL2_SEARCH_VARIANT=2 run packed
SL2_SEARCH_VARIANT=3 run mfma
for w in scenelib2_amd/libscenelib2_amd_w*.so; do SL2_SEARCH_VARIANT=3 SL2_LIB_PATH=$PWD/$w run mfma_$(basename $w .so); done
