mkdir -p gpurun_out/r06b
timeout 1800 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity_f64.py tests/test_gpu_fullsize_parity.py tests/test_gpu_dynamic_fused.py tests/test_gpu_unfused.py tests/test_gpu_c_adapter.py -q 2>&1 | grep -E "^E  |Error|assert|^tests|passed|failed|full size" | head -60 > gpurun_out/r06b/tests.txt
python bench.py --no-dp-projection --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/r06b/bench_expr.json
