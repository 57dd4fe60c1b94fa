# The whole GPU test suite + one headline bench line on the GPU box:  gpurun -- bash tools/gpu_suite.sh  -> gpurun_out/r06c/
mkdir -p gpurun_out/r06c
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -30 > gpurun_out/r06c/tests.txt
python bench.py --no-dp-projection --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06c/bench.json
