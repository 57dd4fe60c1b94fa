set -u
mkdir -p gpurun_out/r06b
timeout 900 python -m pytest tests/test_gpu_dynamic_fused.py tests/test_gpu_dynamic.py tests/test_gpu_step.py tests/test_abi.py -x -q 2>&1 | tail -30 > gpurun_out/r06b/tests.txt
for f in reference activate fused full; do for d in 3 9; do
  timeout 300 python bench.py --dynamic --dynamic-form $f --dynamic-channels $d --steps 20 --warmup 5 2>gpurun_out/r06b/err_${f}_$d.txt | tail -1 > gpurun_out/r06b/dyn_${f}_$d.json
done; done
bash tools/prof.sh r06dyn_full --dynamic --dynamic-form full > gpurun_out/r06b/prof_full.log 2>&1
echo done
