mkdir -p gpurun_out/r06b
timeout 1200 python -m pytest tests/test_gpu_dynamic_fused.py tests/test_gpu_step.py tests/test_abi.py tests/test_gpu_bench_launch.py -q 2>&1 | grep -E "^E  |Error|assert|^tests|passed|failed" | head -40 > gpurun_out/r06b/tests.txt
