#!/bin/bash
# Everything profiles/rNN_* is made from, on the GPU box:  tools/final_profiles.sh r06
set -u
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/final
mkdir -p "$out"
cd "$root"
# (the counter passes first: bench.py reads roofline.traffic / roofline.valu from the file they produce)
bash tools/pmc.sh "$out/${tag}_pmc_traffic.json" > "$out/pmc.log" 2>&1
cp "$out/${tag}_pmc_traffic.json" profiles/${tag}_pmc_traffic.json
python bench.py > "$out/${tag}_bench.json" 2> "$out/bench.err" < /dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/${tag}_bench_driver_cmd.json" 2>> "$out/bench.err" < /dev/null
{
  for f in "--quantize" "--quantize --quantize-reference-calls" "--quantize --ada-mask" "--quantize --ada-mask --quantize-reference-calls"; do
    python bench.py --no-cpu-baseline --no-extras --no-dp-projection $f 2>/dev/null < /dev/null | tail -1
  done
} > "$out/${tag}_bench_quantize.jsonl"
bash tools/prof.sh $tag > "$out/prof.log" 2>&1
cp gpurun_out/prof_${tag}_kernel_stats.csv "$out/${tag}_kernel_stats.csv"
cp gpurun_out/prof_${tag}_last_step.txt "$out/${tag}_last_step_timeline.txt"
bash tools/prof.sh ${tag}q --quantize --ada-mask > "$out/profq.log" 2>&1
cp gpurun_out/prof_${tag}q_kernel_stats.csv "$out/${tag}_quantize_kernel_stats.csv"
cp gpurun_out/prof_${tag}q_last_step.txt "$out/${tag}_quantize_last_step_timeline.txt"
{
  python tools/bench_multicam.py 2>/dev/null
  python tools/bench_inference.py 2>/dev/null
  AB=1 python tools/cpu_overhead.py 2>/dev/null | head -2
  AB=1 python tools/cpu_overhead.py 2>/dev/null | head -2
  for cfg in "1 1" "0 1" "1 0" "0 0"; do set -- $cfg
    GS_PRESORT=$1 GS_STEP_DRIVER=$2 python bench.py --steps 100 --min-timed-s 2 --no-cpu-baseline --no-extras --no-dp-projection 2>/dev/null < /dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('A/B bucketed pre-sort', sys.argv[1], 'step driver', sys.argv[2], ': ms/step', round(d['ms_per_step'],4))" $1 $2
  done
} > "$out/${tag}_secondary.txt" 2>&1
python tools/bench_profile_protocol.py 5 > "$out/${tag}_profile_protocol.txt" 2>&1
python -m pytest tests/test_gpu_fullsize_parity.py -q -s 2>&1 | grep -E "full size|config 4|passed|failed" > "$out/${tag}_fullsize_parity.txt"
./build_abl/mfma_reduce_ab > "$out/${tag}_mfma_reduce_ab.txt" 2>&1
./build_abl/presort_bench > "$out/${tag}_presort_breakdown.txt" 2>&1
{
  python tools/host_phases.py 2>&1 | grep " us"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-dp-projection 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; print('plain ms/step', round(json.loads(sys.stdin.read())['ms_per_step'], 4))"
  for i in 1 2 3; do
    GS_BENCH_PG=1 GS_DIST_FORCE_COLLECTIVES=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2965$i python bench.py --gpus 1 --dp-mode gaussian --steps 20 --warmup 5 --min-timed-s 0.3 --no-cpu-baseline --no-extras --no-dp-projection 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; print('gaussian-sharded, world 1, forced collectives: ms/step', round(json.loads(sys.stdin.read())['ms_per_step'], 4))"
  done
} > "$out/${tag}_gaussian_mode.txt" 2>&1
# 5..32 colour channels (round 5): the step at config 2's size, the reference's 32-channel protocol row, kernel stats
{
  python tools/bench_channels.py 3 9 16 32 2>&1 | grep -v amdgpu.ids
  python tools/bench_profile_protocol.py --channels 32 1 2>&1 | grep -v amdgpu.ids
} > "$out/${tag}_channels.txt" 2>&1
bash tools/prof_cmd.sh ${tag}_ch9 python "$root/tools/bench_channels.py" 9 > "$out/prof_ch9.log" 2>&1
cp gpurun_out/prof_${tag}_ch9_kernel_stats.csv "$out/${tag}_channels9_kernel_stats.csv"
bash tools/prof_cmd.sh ${tag}_ch32 python "$root/tools/bench_channels.py" 32 > "$out/prof_ch32.log" 2>&1
cp gpurun_out/prof_${tag}_ch32_kernel_stats.csv "$out/${tag}_channels32_kernel_stats.csv"
[ -x ./build_abl/grid_barrier ] && timeout 120 ./build_abl/grid_barrier > "$out/${tag}_grid_barrier_raw.txt" 2>&1
# extended fuzz sessions on the final kernels (shifted seeds; every kernel route)
{
  for k in ${FUZZ_OFFSETS:-6100 6200 6300 6400 6500 6600 6700 6800}; do
    echo "GS_FUZZ_SEED_OFFSET=$k: $(GS_FUZZ_SEED_OFFSET=$k python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -1)"
    echo "GS_FUZZ_SEED_OFFSET=$k (dynamic, fused vs chain): $(GS_FUZZ_SEED_OFFSET=$k python -m pytest tests/test_gpu_dynamic_fused.py -q -k fuzz 2>&1 | tail -1)"
  done
} > "$out/${tag}_fuzz_extended.txt" 2>&1
# ---- round 6: BASELINE config 5's step (bench.py --dynamic): counters, bench lines per form, kernel tables, timelines
PMC_BENCH_ARGS="--dynamic" PMC_WORKLOAD_KEY=dynamic_2000000_1920x1080_ch3_full bash tools/pmc.sh "$out/${tag}_pmc_traffic_dynamic.json" > "$out/pmc_dyn.log" 2>&1
cp "$out/${tag}_pmc_traffic_dynamic.json" profiles/${tag}_pmc_traffic_dynamic.json
{
  for f in reference activate fused full; do for d in 3 9; do
    python bench.py --dynamic --dynamic-form $f --dynamic-channels $d 2>/dev/null < /dev/null | tail -1
  done; done
  # the same splats in Z-order (config.splat_order = "morton"): what a spatially sorted array buys the streaming stages
  for f in reference full; do python bench.py --dynamic --dynamic-form $f --dynamic-order morton 2>/dev/null < /dev/null | tail -1; done
} > "$out/${tag}_bench_dynamic.jsonl"
for f in reference full; do
  bash tools/prof.sh ${tag}dyn_$f --dynamic --dynamic-form $f > "$out/prof_dyn_$f.log" 2>&1
  cp gpurun_out/prof_${tag}dyn_${f}_kernel_stats.csv "$out/${tag}_dynamic_${f}_kernel_stats.csv"
  cp gpurun_out/prof_${tag}dyn_${f}_last_step.txt "$out/${tag}_dynamic_${f}_last_step_timeline.txt"
done
bash tools/prof.sh ${tag}dyn_full9 --dynamic --dynamic-form full --dynamic-channels 9 > "$out/prof_dyn_full9.log" 2>&1
cp gpurun_out/prof_${tag}dyn_full9_kernel_stats.csv "$out/${tag}_dynamic_full_ch9_kernel_stats.csv"
# counters of one wide instance (9 channels): HBM traffic + VALU instructions per launch of the tile forward / wide backward
PMC_CMD="python $root/tools/bench_channels.py 9" bash tools/pmc_any.sh ${tag}_ch9 FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU > "$out/pmc_ch9.log" 2>&1
cp gpurun_out/pmc_${tag}_ch9.txt "$out/${tag}_channels9_pmc.txt"
python tools/bench_order.py 2>&1 | grep -v amdgpu.ids > "$out/${tag}_splat_order.txt"
python tools/bench_modes.py 2>&1 | grep -v amdgpu.ids > "$out/${tag}_modes.txt"
timeout 600 python tools/probe_big.py 2>&1 | grep -v amdgpu.ids > "$out/${tag}_big_sizes.txt"
timeout 300 python tools/probe_two_streams.py > "$out/${tag}_probe_two_streams.txt" 2>&1
echo done
