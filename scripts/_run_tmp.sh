timeout 900 python -m pytest tests -q -x --timeout=600 -m gpu > gpurun_out/r03_e_pytest.log 2>&1; tail -4 gpurun_out/r03_e_pytest.log
timeout 600 python bench.py --cpu-sample 64 > gpurun_out/r03_e_bench.json 2> gpurun_out/r03_e_bench.err; python -c "
import json;d=json.load(open('gpurun_out/r03_e_bench.json'));print(d['value'],d['ms_per_step'],d['parity'],{k:round(v['ms_per_step'],4) for k,v in d['kernels'].items()})"
