#!/bin/bash
# A/B of environment knobs on the GPU box:  tools/ab_env.sh "<bench args>" "NAME=VAL [NAME=VAL...]" ...
set -u
ARGS=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  env $v timeout -k 10 300 python3 -u bench.py --no-cpu-baseline --no-rr --no-reference-count --others= $ARGS 2> gpurun_out/abenv.err | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('$v', 'ms', j['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'nodes/ray', r['nodes_per_ray'], 'prims/ray', r['prim_tests_per_ray'], 'setup_s', j['config']['setup_s'], flush=True)
" || { echo "$v failed"; tail -3 gpurun_out/abenv.err; exit 1; }
done
