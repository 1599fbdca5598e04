set -o pipefail
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz_oracle.py tests/test_gpu_planes.py tests/test_gpu_spheres.py tests/test_gpu_ties.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r03_gpu_h.log 2>&1 || { tail -30 gpurun_out/r03_gpu_h.log; exit 1; }
tail -1 gpurun_out/r03_gpu_h.log
for w in c4 c5 c3 c2; do bash tools/ab_variants.sh --workload $w | sed "s/^/$w /" | tee -a gpurun_out/r03_ab_shaderec.txt; done
