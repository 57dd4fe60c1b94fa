set -u
mkdir -p gpurun_out/r06a
for f in reference activate; do for d in 3 9; do
  timeout 300 python bench.py --dynamic --dynamic-form $f --dynamic-channels $d --steps 20 --warmup 5 2>gpurun_out/r06a/err_${f}_$d.txt | tail -1 > gpurun_out/r06a/dyn_${f}_$d.json
done; done
bash tools/prof.sh r06dyn_ref --dynamic --dynamic-form reference > gpurun_out/r06a/prof_ref.log 2>&1
bash tools/prof.sh r06dyn_ref9 --dynamic --dynamic-form reference --dynamic-channels 9 > gpurun_out/r06a/prof_ref9.log 2>&1
timeout 300 python tools/probe_two_streams.py > gpurun_out/r06a/probe_two_streams.txt 2>&1
echo done
