timeout 1200 python -m pytest tests -q -x --timeout=900 -m gpu > gpurun_out/r03_g_pytest.log 2>&1; tail -12 gpurun_out/r03_g_pytest.log
