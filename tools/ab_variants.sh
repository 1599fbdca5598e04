#!/bin/bash
# A/B of kernel builds: every raytracer_amd/lib/variants/*.so (and the product library) on the bench workload, kernel ms.
# usage (on the GPU box): tools/ab_variants.sh [bench.py args]   -> gpurun_out/ab_variants.txt
mkdir -p gpurun_out
out=gpurun_out/ab_variants.txt; : > $out
for lib in raytracer_amd/lib/librtcore.so raytracer_amd/lib/variants/*.so; do
  [ -f "$lib" ] || continue
  RTCORE_LIB=$lib timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rr --no-reference-count --others= "$@" > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$lib FAILED" >> $out; exit 1; }
  echo "$lib $(grep -o '"kernel_ms": [0-9.]*' gpurun_out/ab_tmp.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_tmp.json)" >> $out
done
cat $out
