#!/bin/bash
# sky texel boundaries: the probe at three sky sizes, the eight seeds of the 9,602-seed extreme campaign that differed, a short bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u tools/sky_boundary_probe.py 64 32 > gpurun_out/sky_probe_64.log 2>&1; echo "probe 64x32 rc $?"; tail -1 gpurun_out/sky_probe_64.log
timeout -k 10 300 python -u tools/sky_boundary_probe.py 100 50 > gpurun_out/sky_probe_100.log 2>&1; echo "probe 100x50 rc $?"; tail -1 gpurun_out/sky_probe_100.log
timeout -k 10 300 python -u tools/sky_boundary_probe.py 2048 1024 > gpurun_out/sky_probe_2048.log 2>&1; echo "probe 2048x1024 rc $?"; tail -1 gpurun_out/sky_probe_2048.log
timeout -k 10 300 python -u tools/fuzz_oracle.py --extreme --seeds 5524,5914,10096,10841,11277,11471,12394,12407 > gpurun_out/sky_seeds.log 2>&1; echo "seeds rc $?"; tail -1 gpurun_out/sky_seeds.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-rr --no-reference-count --others= > gpurun_out/sky_bench.json 2> gpurun_out/sky_bench.err; echo "bench rc $?"; cut -c1-220 gpurun_out/sky_bench.json
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/sky_gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -3 gpurun_out/sky_gpu_tests.log
