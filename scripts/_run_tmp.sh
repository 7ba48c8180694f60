timeout 1200 python -m pytest tests/test_gpu_slam.py -q -x --timeout=900 -m gpu -k "larger or ragged or variants or configs3" > gpurun_out/r03_j_pytest.log 2>&1; tail -5 gpurun_out/r03_j_pytest.log
