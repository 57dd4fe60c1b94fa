mkdir -p gpurun_out/r06b
timeout 1200 python -m pytest tests/test_gpu_quantize.py tests/test_gpu_configs.py tests/test_gpu_dynamic_fused.py tests/test_abi.py -q 2>&1 | grep -E "^E  |Error|assert|^tests|passed|failed" | head -40 > gpurun_out/r06b/tests.txt
for f in reference full; do python bench.py --dynamic --dynamic-form $f 2>/dev/null | tail -1 > gpurun_out/r06b/dyn2_$f.json; done
