mkdir -p gpurun_out/r06d
for p in 1 0; do GS_GRAD_PREFILL=$p python bench.py --dynamic --dynamic-form full 2>/dev/null | tail -1 > gpurun_out/r06d/prefill_$p.json; done
python bench.py --dynamic --dynamic-form full --dynamic-splats 1000000 2>/dev/null | tail -1 > gpurun_out/r06d/n1m.json
