mkdir -p gpurun_out/r06b
timeout 1200 python -m pytest tests/test_gpu_configs.py -q -s -k "config5_full_size" 2>&1 | grep -E "^E  |Error|assert|config 5|passed|failed" | head -40 > gpurun_out/r06b/tests.txt
