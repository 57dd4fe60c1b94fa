# Gaussian-sharded step at world 1 with the collectives forced through RCCL: tests, bench (plain / gaussian / operator path),
# kernel timeline, host-bound step.  Run on the GPU box: bash tools/dist_profile.sh
mkdir -p gpurun_out/dist
python -m pytest tests/test_gpu_step.py tests/test_gpu_distributed.py -x -q 2>&1 | tail -5
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-dp-projection 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'])"
export GS_BENCH_PG=1 GS_DIST_FORCE_COLLECTIVES=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655
for i in 1 2 3; do python bench.py --gpus 1 --dp-mode gaussian --steps 20 --warmup 5 --min-timed-s 0.3 --no-cpu-baseline --no-extras --no-dp-projection 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gaussian', d['ms_per_step'])"; done
GS_STEP_DRIVER=0 python bench.py --gpus 1 --dp-mode gaussian --steps 20 --warmup 5 --min-timed-s 0.3 --no-cpu-baseline --no-extras --no-dp-projection 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gaussian, operator path', d['ms_per_step'])"
bash tools/prof.sh distg --gpus 1 --dp-mode gaussian > gpurun_out/dist/prof.log 2>&1; tail -3 gpurun_out/dist/prof.log
DIST=1 timeout 300 python tools/cpu_overhead.py 2>&1 | grep "host-bound"
