python -m pytest tests/test_gpu_presort.py tests/test_gpu_ops.py tests/test_gpu_step.py tests/test_gpu_rasterization.py tests/test_gpu_fullsize_parity.py tests/test_gpu_fuzz.py tests/test_gpu_distributed.py -x -q 2>&1 | tail -3
for r in 1 2 3; do
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras --no-dp-projection 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['roofline']['binning']['ms'],4))"
done
bash tools/prof.sh pk > /dev/null 2>&1; cut -c1-120 gpurun_out/prof_pk_last_step.txt | sed -n 1,16p
