mkdir -p gpurun_out/r06b
timeout 1200 python -m pytest tests/test_gpu_unfused.py -q 2>&1 | grep -E "^E  |Error|assert|^tests|passed|failed" | head -60 > gpurun_out/r06b/tests.txt
