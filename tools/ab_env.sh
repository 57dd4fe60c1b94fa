#!/bin/bash
# A/B of an environment switch on the GPU box: tools/ab_env.sh VAR rounds [bench args] -- bench.py with VAR=0 and VAR=1 interleaved
var=$1; rounds=${2:-3}; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
for r in $(seq 1 "$rounds"); do for v in 0 1; do
    ms=$(env $var=$v timeout 300 python "$root/bench.py" --no-cpu-baseline --no-extras --no-dp-projection "$@" 2>/dev/null < /dev/null |
         python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$var=$v $ms"
done; done
