#!/bin/bash
# A/B of kernel build variants on the GPU box, one process per variant, same bench line:
#   tools/ab.sh "<bench args>" name1:path1.so name2:path2.so ...
# (variants are built beforehand on the CPU side: hipcc ... -D<flags> -o raytracer_amd/lib/variants/<name>.so)
set -u
ARGS=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  name=${v%%:*}; lib=${v#*:}
  for rep in 1 2; do
    RTCORE_LIB=$lib timeout -k 10 300 python3 -u bench.py --no-cpu-baseline --no-rr --no-reference-count --others= $ARGS 2> gpurun_out/ab_$name.err | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('$name rep$rep', 'Mrays/s', j['value'], 'ms', j['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'nodes/ray', r['nodes_per_ray'], 'prims/ray', r['prim_tests_per_ray'], flush=True)
" || { echo "$name failed"; tail -3 gpurun_out/ab_$name.err; exit 1; }
  done
done
